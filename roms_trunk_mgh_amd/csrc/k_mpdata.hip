// k_mpdata.hip -- the MPDATA branch of step3d_t_tile (ROMS/Nonlinear/step3d_t.F:363-1318,
// :1431-1501) with mpdata_adiff_tile (ROMS/Nonlinear/mpdata_adiff.F:38-1105) for one tracer:
//
//   K1 k_mp_ta     first-order upstream H + V advection on the extended range
//                  (IstrUm2:Iendp2i, JstrVm2:Jendp2i) -> intermediate tracer Ta (Tunits),
//                  wall rows copied (mpdata_adiff.F:170-240)
//   K2 k_mp_adiff  anti-diffusive velocities Ua, Va, Wa incl. the third-order terms
//                  (MPDATA_HOT) and the |.| <= |Um| clamp (mpdata_adiff.F:258-840)
//   K3 k_mp_beta   flux-corrected-transport factors beta_up / beta_dn (:842-1030)
//   K4 k_mp_update limited transports (:1032-1100) -> corrected H and V advection of Ta
//                  (step3d_t.F:1235-1316) -> classic tridiagonal vertical diffusion
//                  (:1431-1501), one thread per column, Thomas arrays in VGPRs
//
// Ta, Ua, Va, Wa, beta_up, beta_dn live in the library's 3-D scratch arrays with the
// module extents (LBi:UBi,LBj:UBj) -- a superset of the reference's private IminS:ImaxS
// extents with three ghost points.  oHz and odz are recomputed where they are used (the
// same IEEE division, so the same bits).  Expressions keep the reference's association.
#include <cstdlib>
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

#define EPS_MP  1.0E-18
#define EPS2_MP 1.0E-10

struct MpArgs {
  double *Ta, *Ua, *Va, *Wa, *bup, *bdn;   // scratch, module horizontal extents
  const double *oHz, *odz;                 // 1/Hz and 1/(z_r(k+1)-z_r(k)), k_mp_metrics (once per step3d_t call)
  int nnew, itrc;
  // the wall rule of the anti-diffusive velocities on a physical edge (mpdata_adiff.F:577-640, :1031-1100): zero where
  // the 3-D momentum's condition is closed (LBC(side, isBu3d = isUvel / isBv3d = isVvel)%closed), the neighbouring
  // face's value otherwise; [LBS_WEST .. LBS_NORTH]
  int closed[4];
};

__device__ __forceinline__ double upstream(double flx, double a, double b)
{
  return fmax(flx, 0.0) * a + fmin(flx, 0.0) * b;
}

// ------------------------------------------------------- K0: oHz, odz ----
// mpdata_adiff.F:150-168 keeps oHz and odz in private arrays; here once per step3d_t call for all MPDATA
// tracers (Hz and z_r do not change in between).  k_mp_adiff is bound by FP64 division throughput: with
// the reciprocals recomputed at every use it spent half of its divisions on these two.
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_mp_metrics(const RomsDev *__restrict__ c, double *__restrict__ oHz, double *__restrict__ odz)
{
  DEV_PROLOGUE(c)
  const int i = b.LBi + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.LBj + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.UBi || j > b.UBj) return;
  const gcd_t Hz = (gcd_t)c->F.Hz, z_r = (gcd_t)c->F.z_r;
  const long a2 = I2(i, j);
  double zr = z_r[a2];
  for (int k = 1; k <= N; k++) {
    const long a = a2 + (long)(k - 1) * nij;
    oHz[a] = 1.0 / Hz[a];
    if (k < N) {
      const double zu = z_r[a + nij];
      odz[a] = 1.0 / (zu - zr);
      zr = zu;
    }
  }
}

// ---------------------------------------------------------------- K1: Ta ----
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_mp_ta(const RomsDev *__restrict__ c, MpArgs m)
{
  DEV_PROLOGUE(c)
  // level fastest inside an XCD: the planes k-1, k, k+1 of a tile meet in one L2
  const TileLv XB = decode_tile_level(b.Iendp2i - b.IstrUm2 + 1, b.Jendp2i - b.JstrVm2 + 1, N);
  if (!XB.valid) return;
  const int i = b.IstrUm2 + XB.bx * BLK_X + threadIdx.x;
  const int j = b.JstrVm2 + XB.by * BLK_Y + threadIdx.y;
  if (i > b.Iendp2i || j > b.Jendp2i) return;
  const double dt = c->p.dt;
  const gcd_t t3 = (gcd_t)(c->F.t + (2L + 3L * (m.itrc - 1)) * n3r);
  const gcd_t tn = (gcd_t)(c->F.t + ((long)(m.nnew - 1) + 3L * (m.itrc - 1)) * n3r);
  const gcd_t Huon = (gcd_t)c->F.Huon, Hvom = (gcd_t)c->F.Hvom, Wv = (gcd_t)c->F.W, Hz = (gcd_t)c->F.Hz;
  const gd_t Ta = (gd_t)m.Ta;
  const long c0 = I2(i, j);
  const double cff = dt * GF(pm)[c0] * GF(pn)[c0];
  const bool s_wall = b.south_edge && !b.NSperiodic && j == b.Jstr;
  const bool n_wall = b.north_edge && !b.NSperiodic && j == b.Jend;
  const bool w_wall = b.west_edge && !b.EWperiodic && i == b.Istr;
  const bool e_wall = b.east_edge && !b.EWperiodic && i == b.Iend;
  {
    const int k = XB.k0 + 1;                                      // one thread per (i,j,k)
    const long a = c0 + (long)(k - 1) * nij;
    const double t0 = t3[a];
    // FC(k-1): the expression the level below evaluates as its FC(k)
    const double FCm1 = (k > 1) ? upstream(Wv[a], t3[a - nij], t0) : 0.0;
    const double FXi = upstream(Huon[a], t3[a - 1], t0);
    const double FXip1 = upstream(Huon[a + 1], t0, t3[a + 1]);
    const double FEj = upstream(Hvom[a], t3[a - ni], t0);
    const double FEjp1 = upstream(Hvom[a + ni], t0, t3[a + ni]);
    const double cff1 = cff * (FXip1 - FXi);
    const double cff2 = cff * (FEjp1 - FEj);
    const double cff3 = cff1 + cff2;
    double ta = tn[a] - cff3;                                   // step3d_t.F:838
    const double FCk = (k < N) ? upstream(Wv[a + nij], t0, t3[a + nij]) : 0.0;   // :1006-1018
    const double c1 = cff * (FCk - FCm1);
    ta = (ta - c1) * (1.0 / Hz[a]);                             // :1175
    Ta[a] = ta;
    if (w_wall) Ta[a - 1] = ta;                                 // mpdata_adiff.F:160-176
    if (e_wall) Ta[a + 1] = ta;
    if (s_wall) Ta[a - ni] = ta;                                // mpdata_adiff.F:177-199
    if (n_wall) Ta[a + ni] = ta;
    // corners, :201-240: the mean of the two neighbouring boundary values, both copies of this cell
    if (s_wall && w_wall) Ta[a - ni - 1] = 0.5 * (ta + ta);
    if (s_wall && e_wall) Ta[a - ni + 1] = 0.5 * (ta + ta);
    if (n_wall && w_wall) Ta[a + ni - 1] = 0.5 * (ta + ta);
    if (n_wall && e_wall) Ta[a + ni + 1] = 0.5 * (ta + ta);
  }
}

// Quotient n/d.  Exact variant: the IEEE division of the reference (bit-identical results).  FAST variant
// (roms_params_t.mpdata_fast): n times a reciprocal from v_rcp_f64 refined by two Newton steps (explicit FMAs,
// within ~1 ulp of 1/d) -- one reciprocal per distinct denominator, the compiler merges the three quotients of
// a face that share the tracer sum.  The IEEE division costs ~30 VALU instructions on gfx950, the refined
// reciprocal 5, and this kernel evaluates 49 quotients per cell; validated against the oracle at the north-star
// bound (1e-10 relative RMS after 100 steps, tests/test_gpu_mpdata.py).
template <bool FAST>
__device__ __forceinline__ double mp_div(double n, double d)
{
  if constexpr (FAST) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return n * r;
  } else {
    return n / d;
  }
}
#define DV(n, d) mp_div<FAST>((n), (d))

// ------------------------------------------------- K2: Ua, Va, Wa (raw) ----
// vertical-gradient factor C and mean vertical Courant number Wm of a face between
// column p (offset 0) and column q (offset dq), mpdata_adiff.F:262-310 / :456-504
template <bool FAST>
__device__ __forceinline__ void face_CW(const gcd_t Ta, const gcd_t z_r, const gcd_t odzA, const gcd_t Wv, const gcd_t pm,
                                        const gcd_t pn, long a2, long a, long dq, long nij, int k, int N,
                                        double dt, double &C, double &Wm)
{
  // a = index of (p,k) in rho arrays, a2 = 2-D index of p; q = p + dq
  const long aq = a + dq, a2q = a2 + dq;
  auto odz = [&](long x) { return odzA[x]; };                             // odz at level of x
  const long w = a + nij, wq = aq + nij;      // W(.,.,k) of a K_3DW array = rho index + nij
  if (k == 1) {
    C = DV(0.25 * ((Ta[a + nij] - Ta[a]) * odz(a) + (Ta[aq + nij] - Ta[aq]) * odz(aq)) *
               (z_r[a + nij] - z_r[a] + z_r[aq + nij] - z_r[aq]), Ta[aq] + Ta[a] + EPS_MP);
    Wm = 0.25 * dt * (Wv[wq] * odz(aq) * pm[a2q] * pn[a2q] + Wv[w] * odz(a) * pm[a2] * pn[a2]);
  } else if (k < N) {
    C = DV(0.0625 *
               ((Ta[a + nij] - Ta[a]) * odz(a) + (Ta[a] - Ta[a - nij]) * odz(a - nij) +
                (Ta[aq + nij] - Ta[aq]) * odz(aq) + (Ta[aq] - Ta[aq - nij]) * odz(aq - nij)) *
               (z_r[a + nij] - z_r[a - nij] + z_r[aq + nij] - z_r[aq - nij]), Ta[aq] + Ta[a] + EPS_MP);
    Wm = 0.25 * dt *
         ((Wv[wq - nij] * odz(aq - nij) + Wv[wq] * odz(aq)) * pm[a2q] * pn[a2q] +
          (Wv[w] * odz(a) + Wv[w - nij] * odz(a - nij)) * pm[a2] * pn[a2]);
  } else {
    C = DV(0.25 * ((Ta[a] - Ta[a - nij]) * odz(a - nij) + (Ta[aq] - Ta[aq - nij]) * odz(aq - nij)) *
               (z_r[a] - z_r[a - nij] + z_r[aq] - z_r[aq - nij]), Ta[aq] + Ta[a] + EPS_MP);
    Wm = 0.25 * dt * (Wv[wq - nij] * odz(aq - nij) * pm[a2q] * pn[a2q] + Wv[w - nij] * odz(a - nij) * pm[a2] * pn[a2]);
  }
}

// One launch for the three faces (they share the Ta, Huon, Hvom, oHz loads): 208 VGPRs, two waves per SIMD.
// The kernel is FP64-issue bound (per cell and level ~1800 VALU instructions, 49 divisions among them); one
// launch per face raises the occupancy to 3-4 waves but repeats the shared loads and was 20 % slower.
// MASK (MASKING applications, separate instantiations so that the unmasked kernel keeps its code): the horizontal
// differences of the cross terms times the mask of their face (mpdata_adiff.F:288-297, :470-480, :640-660) and
// the clamped velocities times umask / vmask / rmask (:395, :568, :801).
template <bool FAST, bool MASK>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_mp_adiff(const RomsDev *__restrict__ c, MpArgs m)
{
  DEV_PROLOGUE(c)
  // union of the three ranges: i = IstrU-1 : Iendp2, j = JstrV-1 : Jendp2
  // level fastest inside an XCD (see k_mp_ta)
  const TileLv XB = decode_tile_level(b.Iendp2 - (b.IstrU - 1) + 1, b.Jendp2 - (b.JstrV - 1) + 1, N);
  if (!XB.valid) return;
  const int i = b.IstrU - 1 + XB.bx * BLK_X + threadIdx.x;
  const int j = b.JstrV - 1 + XB.by * BLK_Y + threadIdx.y;
  if (i > b.Iendp2 || j > b.Jendp2) return;
  const bool do_u = j <= b.Jendp1;                                  // Ua: j = JstrV-1:Jendp1, i = IstrU-1:Iendp2
  const bool do_v = j >= b.JstrVm1 && i <= b.Iendp1;                // Va: j = JstrVm1:Jendp2, i = IstrU-1:Iendp1
  const bool do_w = j <= b.Jendp1 && i <= b.Iendp1;                 // Wa: j = JstrV-1:Jendp1, i = IstrU-1:Iendp1
  const double dt = c->p.dt;
  const gcd_t Ta = (gcd_t)m.Ta;
  const gcd_t z_r = (gcd_t)c->F.z_r, Wv = (gcd_t)c->F.W, Hz = (gcd_t)c->F.Hz;
  const gcd_t Huon = (gcd_t)c->F.Huon, Hvom = (gcd_t)c->F.Hvom;
  const gcd_t pm = (gcd_t)c->F.pm, pn = (gcd_t)c->F.pn, on_v = (gcd_t)c->F.on_v, om_u = (gcd_t)c->F.om_u;
  const gd_t Ua = (gd_t)m.Ua, Va = (gd_t)m.Va, Wa = (gd_t)m.Wa;
  const long a2 = I2(i, j);
  // faces on a physical edge, mpdata_adiff.F:577-640: zero (closed) or the value of the next face inside, which the
  // thread of that face stores
  const bool v_wall_n = b.north_edge && !b.NSperiodic && j == b.Jend + 1;
  const bool u_wall_w = b.west_edge && !b.EWperiodic && i == b.Istr;
  const bool u_wall_e = b.east_edge && !b.EWperiodic && i == b.Iend + 1;
  const gcd_t oHzA = (gcd_t)m.oHz, odzA = (gcd_t)m.odz;
  auto oHz = [&](long x) { return oHzA[x]; };
  const gcd_t umk = (gcd_t)c->F.umask, vmk = (gcd_t)c->F.vmask, rmk = (gcd_t)c->F.rmask;
  auto UM = [&](double x, long q) { if constexpr (MASK) return x * umk[q]; else return x; };   // x * umask(q)
  auto VM = [&](double x, long q) { if constexpr (MASK) return x * vmk[q]; else return x; };
  // one thread per (i,j,k): nothing is carried from level to level, and with ~1200 FP64 instructions per
  // cell the kernel needs every wave it can get (a k-loop per column ran at two waves per SIMD)
  {
    const int k = XB.k0 + 1;
    const long a = a2 + (long)(k - 1) * nij;
    const double T0 = Ta[a];
    // ---------------- XI face between (i-1,j) and (i,j) ----------------
    if (do_u) {
      const double Tw = Ta[a - 1];
      double ua = 0.0;
      if (!u_wall_w && !u_wall_e && !((Tw <= 0.0) || (T0 <= 0.0) || (fabs(Tw - T0) <= EPS2_MP))) {
        double Ck, Wk;
        face_CW<FAST>(Ta, z_r, odzA, Wv, pm, pn, a2, a, -1, nij, k, N, dt, Ck, Wk);
        const double A = DV(T0 - Tw, T0 + Tw + EPS_MP);
        double B = 0.03125 *
                   (VM((Ta[a + ni] - T0) * (pn[a2] + pn[a2 + ni]), a2 + ni) + VM((T0 - Ta[a - ni]) * (pn[a2 - ni] + pn[a2]), a2) +
                    VM((Ta[a - 1 + ni] - Tw) * (pn[a2 - 1] + pn[a2 - 1 + ni]), a2 - 1 + ni) +
                    VM((Tw - Ta[a - 1 - ni]) * (pn[a2 - 1 - ni] + pn[a2 - 1]), a2 - 1));
        B = DV(B * (on_v[a2] + on_v[a2 + ni] + on_v[a2 - 1] + on_v[a2 - 1 + ni]), Tw + T0 + EPS_MP);
        const double Um = 0.125 * Huon[a] * dt * (pm[a2] + pm[a2 - 1]) * (pn[a2] + pn[a2 - 1]) * (oHz(a - 1) + oHz(a));
        const double Vm = 0.03125 * dt *
                          (Hvom[a - 1] * (pm[a2 - 1] + pm[a2 - 1 - ni]) * (pn[a2 - 1] + pn[a2 - 1 - ni]) *
                               (oHz(a - 1) + oHz(a - 1 - ni)) +
                           Hvom[a - 1 + ni] * (pm[a2 - 1 + ni] + pm[a2 - 1]) * (pn[a2 - 1 + ni] + pn[a2 - 1]) *
                               (oHz(a - 1 + ni) + oHz(a - 1)) +
                           Hvom[a] * (pm[a2] + pm[a2 - ni]) * (pn[a2] + pn[a2 - ni]) * (oHz(a) + oHz(a - ni)) +
                           Hvom[a + ni] * (pm[a2 + ni] + pm[a2]) * (pn[a2 + ni] + pn[a2]) * (oHz(a + ni) + oHz(a)));
        const double X = (fabs(Um) - Um * Um) * A - B * Um * Vm - Ck * Um * Wk;
        const double Y = (fabs(Vm) - Vm * Vm) * B - A * Um * Vm - Ck * Vm * Wk;
        const double Z = (fabs(Wk) - Wk * Wk) * Ck - A * Um * Wk - B * Vm * Wk;
        const double AA = A * A, BB = B * B, CC = Ck * Ck, AB = A * B, AC = A * Ck;
        const double XX = X * X, YY = Y * Y, ZZ = Z * Z, XY = X * Y, XZ = X * Z;
        const double sig_alfa = DV(1.0, 1.0 - fabs(A) + EPS_MP);
        const double sig_beta = DV(-A, (1.0 - fabs(A)) * (1.0 - AA) + EPS_MP);
        const double sig_gama = DV(2.0 * fabs(AA * A), (1.0 - fabs(A)) * (1.0 - AA) * (1.0 - fabs(AA * A)) + EPS_MP);
        const double sig_a = DV(-B, (1.0 - fabs(A)) * (1.0 - fabs(AB)) + EPS_MP);
        const double sig_b = DV(AB, (1.0 - fabs(A)) * (1.0 - AA * fabs(B)) + EPS_MP) *
                             (DV(fabs(B), 1.0 - fabs(AB) + EPS_MP) + DV(2.0 * A, 1.0 - AA + EPS_MP));
        const double sig_c = DV(fabs(A) * BB, (1.0 - fabs(A)) * (1.0 - BB * fabs(A)) * (1.0 - fabs(AB)) + EPS_MP);
        const double sig_d = DV(-Ck, (1.0 - fabs(A)) * (1.0 - fabs(AC)) + EPS_MP);
        const double sig_e = DV(AC, (1.0 - fabs(A)) * (1.0 - AA * fabs(Ck)) + EPS_MP) *
                             (DV(fabs(Ck), 1.0 - fabs(AC) + EPS_MP) + DV(2.0 * A, 1.0 - AA + EPS_MP));
        const double sig_f = DV(fabs(A) * CC, (1.0 - fabs(A)) * (1.0 - CC * fabs(A)) * (1.0 - fabs(AC)) + EPS_MP);
        const double u0 = sig_alfa * X + sig_beta * XX + sig_gama * XX * X + sig_a * XY + sig_b * XX * Y +
                          sig_c * X * YY + sig_d * XZ + sig_e * XX * Z + sig_f * X * ZZ;
        ua = fmin(fabs(u0), 1.0 * fabs(Um)) * copysign(1.0, u0);
        ua = UM(ua, a2);
      }
      if (u_wall_w) { if (m.closed[LBS_WEST]) Ua[a] = 0.0; }
      else if (u_wall_e) { if (m.closed[LBS_EAST]) Ua[a] = 0.0; }
      else {
        Ua[a] = ua;
        if (b.west_edge && !b.EWperiodic && i == b.Istr + 1 && !m.closed[LBS_WEST]) Ua[a - 1] = ua;
        if (b.east_edge && !b.EWperiodic && i == b.Iend && !m.closed[LBS_EAST]) Ua[a + 1] = ua;
      }
    }
    // ---------------- ETA face between (i,j-1) and (i,j) ----------------
    if (do_v) {
      const double Ts = Ta[a - ni];
      double va = 0.0;
      if (!v_wall_n && !((Ts <= 0.0) || (T0 <= 0.0) || (fabs(Ts - T0) <= EPS2_MP))) {
        double Ck, Wk;
        face_CW<FAST>(Ta, z_r, odzA, Wv, pm, pn, a2, a, -ni, nij, k, N, dt, Ck, Wk);
        double A = 0.03125 *
                   (UM((Ta[a + 1] - T0) * (pm[a2 + 1] + pm[a2]), a2 + 1) + UM((T0 - Ta[a - 1]) * (pm[a2 - 1] + pm[a2]), a2) +
                    UM((Ta[a + 1 - ni] - Ts) * (pm[a2 + 1 - ni] + pm[a2 - ni]), a2 + 1 - ni) +
                    UM((Ts - Ta[a - 1 - ni]) * (pm[a2 - 1 - ni] + pm[a2 - ni]), a2 - ni));
        A = DV(A * (om_u[a2] + om_u[a2 + 1] + om_u[a2 - ni] + om_u[a2 + 1 - ni]), Ts + T0 + EPS_MP);
        const double B = DV(T0 - Ts, T0 + Ts + EPS_MP);
        const double Um = 0.03125 * dt *
                          (Huon[a + 1] * (pm[a2 + 1] + pm[a2]) * (pn[a2 + 1] + pn[a2]) * (oHz(a + 1) + oHz(a)) +
                           Huon[a + 1 - ni] * (pm[a2 + 1 - ni] + pm[a2 - ni]) * (pn[a2 + 1 - ni] + pn[a2 - ni]) *
                               (oHz(a + 1 - ni) + oHz(a - ni)) +
                           Huon[a] * (pm[a2 - 1] + pm[a2]) * (pn[a2 - 1] + pn[a2]) * (oHz(a - 1) + oHz(a)) +
                           Huon[a - ni] * (pm[a2 - 1 - ni] + pm[a2 - ni]) * (pn[a2 - 1 - ni] + pn[a2 - ni]) *
                               (oHz(a - 1 - ni) + oHz(a - ni)));
        const double Vm = 0.125 * Hvom[a] * dt * (pn[a2 - ni] + pn[a2]) * (pm[a2 - ni] + pm[a2]) * (oHz(a - ni) + oHz(a));
        const double X = (fabs(Um) - Um * Um) * A - B * Um * Vm - Ck * Um * Wk;
        const double Y = (fabs(Vm) - Vm * Vm) * B - A * Um * Vm - Ck * Vm * Wk;
        const double Z = (fabs(Wk) - Wk * Wk) * Ck - A * Um * Wk - B * Vm * Wk;
        const double AA = A * A, BB = B * B, CC = Ck * Ck, AB = A * B, BC = B * Ck;
        const double XX = X * X, YY = Y * Y, ZZ = Z * Z, XY = X * Y, YZ = Y * Z;
        const double sig_alfa = DV(1.0, 1.0 - fabs(B) + EPS_MP);
        const double sig_beta = DV(-B, (1.0 - fabs(B)) * (1.0 - BB) + EPS_MP);
        const double sig_gama = DV(2.0 * fabs(BB * B), (1.0 - fabs(B)) * (1.0 - BB) * (1.0 - fabs(BB * B)) + EPS_MP);
        const double sig_a = DV(-A, (1.0 - fabs(B)) * (1.0 - fabs(AB)) + EPS_MP);
        const double sig_b = DV(AB, (1.0 - fabs(B)) * (1.0 - BB * fabs(A)) + EPS_MP) *
                             (DV(fabs(A), 1.0 - fabs(AB) + EPS_MP) + DV(2.0 * B, 1.0 - BB + EPS_MP));
        const double sig_c = DV(fabs(B) * AA, (1.0 - fabs(B)) * (1.0 - AA * fabs(B)) * (1.0 - fabs(AB)) + EPS_MP);
        const double sig_d = DV(-Ck, (1.0 - fabs(B)) * (1.0 - fabs(BC)) + EPS_MP);
        const double sig_e = DV(BC, (1.0 - fabs(B)) * (1.0 - BB * fabs(Ck)) + EPS_MP) *
                             (DV(fabs(Ck), 1.0 - fabs(BC) + EPS_MP) + DV(2.0 * B, 1.0 - BB + EPS_MP));
        const double sig_f = DV(fabs(B) * CC, (1.0 - fabs(B)) * (1.0 - CC * fabs(B)) * (1.0 - fabs(BC)) + EPS_MP);
        const double v0 = sig_alfa * Y + sig_beta * YY + sig_gama * YY * Y + sig_a * XY + sig_b * Y * XX +
                          sig_c * YY * X + sig_d * YZ + sig_e * YY * Z + sig_f * Y * ZZ;
        va = fmin(fabs(v0), 1.0 * fabs(Vm)) * copysign(1.0, v0);
        va = VM(va, a2);
      }
      if (v_wall_n) { if (m.closed[LBS_NORTH]) Va[a] = 0.0; }
      else {
        Va[a] = va;
        // southern edge: Va(i,Jstr) (:612-625); row Jstr is below this kernel's Va range
        if (b.south_edge && !b.NSperiodic && j == b.Jstr + 1) Va[a - ni] = m.closed[LBS_SOUTH] ? 0.0 : va;
        if (b.north_edge && !b.NSperiodic && j == b.Jend && !m.closed[LBS_NORTH]) Va[a + ni] = va;
      }
    }
    // ---------------- W face between levels k and k+1 ----------------
    if (do_w) {
      const long aw = a + nij;                 // Wa(i,j,k) in a (0:N) array
      if (k == 1) Wa[a2] = 0.0;                // Wa(i,j,0)
      if (k == N) { Wa[aw] = 0.0; return; }    // Wa(i,j,N)
      const double Tu = Ta[a + nij];
      double wa = 0.0;
      if (!((T0 <= 0.0) || (Tu <= 0.0) || (fabs(T0 - Tu) <= EPS2_MP))) {
        const double Ck = DV(Tu - T0, Tu + T0 + EPS_MP);
        double A = 0.0625 *
                   (UM((Ta[a + 1 + nij] - Tu) * (pm[a2 + 1] + pm[a2]), a2 + 1) + UM((Tu - Ta[a - 1 + nij]) * (pm[a2] + pm[a2 - 1]), a2) +
                    UM((Ta[a + 1] - T0) * (pm[a2 + 1] + pm[a2]), a2 + 1) + UM((T0 - Ta[a - 1]) * (pm[a2] + pm[a2 - 1]), a2));
        double B = 0.0625 *
                   (VM((Ta[a + ni + nij] - Tu) * (pn[a2 + ni] + pn[a2]), a2 + ni) + VM((Tu - Ta[a - ni + nij]) * (pn[a2] + pn[a2 - ni]), a2) +
                    VM((Ta[a + ni] - T0) * (pn[a2 + ni] + pn[a2]), a2 + ni) + VM((T0 - Ta[a - ni]) * (pn[a2] + pn[a2 - ni]), a2));
        A = DV(A * (om_u[a2 + 1] + om_u[a2]), Tu + T0 + EPS_MP);
        B = DV(B * (on_v[a2 + ni] + on_v[a2]), Tu + T0 + EPS_MP);
        const double Um = 0.03125 * dt *
                          (Huon[a] * (pm[a2] + pm[a2 - 1]) * (pn[a2] + pn[a2 - 1]) * (oHz(a) + oHz(a - 1)) +
                           Huon[a + nij] * (pm[a2] + pm[a2 - 1]) * (pn[a2] + pn[a2 - 1]) * (oHz(a + nij) + oHz(a - 1 + nij)) +
                           Huon[a + 1] * (pm[a2] + pm[a2 + 1]) * (pn[a2] + pn[a2 + 1]) * (oHz(a) + oHz(a + 1)) +
                           Huon[a + 1 + nij] * (pm[a2] + pm[a2 + 1]) * (pn[a2] + pn[a2 + 1]) *
                               (oHz(a + nij) + oHz(a + 1 + nij)));
        const double Vm = 0.03125 * dt *
                          (Hvom[a] * (pm[a2] + pm[a2 - ni]) * (pn[a2] + pn[a2 - ni]) * (oHz(a) + oHz(a - ni)) +
                           Hvom[a + nij] * (pm[a2] + pm[a2 - ni]) * (pn[a2] + pn[a2 - ni]) * (oHz(a + nij) + oHz(a - ni + nij)) +
                           Hvom[a + ni] * (pm[a2] + pm[a2 + ni]) * (pn[a2] + pn[a2 + ni]) * (oHz(a) + oHz(a + ni)) +
                           Hvom[a + ni + nij] * (pm[a2] + pm[a2 + ni]) * (pn[a2] + pn[a2 + ni]) *
                               (oHz(a + nij) + oHz(a + ni + nij)));
        const double Wk = Wv[aw] * odzA[a] * pm[a2] * pn[a2] * dt;
        const double X = (fabs(Um) - Um * Um) * A - B * Um * Vm - Ck * Um * Wk;
        const double Y = (fabs(Vm) - Vm * Vm) * B - A * Um * Vm - Ck * Vm * Wk;
        const double Z = (fabs(Wk) - Wk * Wk) * Ck - A * Um * Wk - B * Vm * Wk;
        const double AA = A * A, BB = B * B, CC = Ck * Ck, AC = A * Ck, BC = B * Ck;
        const double XX = X * X, YY = Y * Y, ZZ = Z * Z, XZ = X * Z, YZ = Y * Z;
        const double sig_alfa = DV(1.0, 1.0 - fabs(Ck) + EPS_MP);
        const double sig_beta = DV(-Ck, (1.0 - fabs(Ck)) * (1.0 - CC) + EPS_MP);
        const double sig_gama = DV(2.0 * fabs(CC * Ck), (1.0 - fabs(Ck)) * (1.0 - CC) * (1.0 - fabs(CC * Ck)) + EPS_MP);
        const double sig_a = DV(-B, (1.0 - fabs(Ck)) * (1.0 - fabs(BC)) + EPS_MP);
        const double sig_b = DV(BC, (1.0 - fabs(Ck)) * (1.0 - CC * fabs(B)) + EPS_MP) *
                             (DV(fabs(B), 1.0 - fabs(BC) + EPS_MP) + DV(2.0 * Ck, 1.0 - CC + EPS_MP));
        const double sig_c = DV(fabs(Ck) * BB, (1.0 - fabs(Ck)) * (1.0 - B * B * fabs(Ck)) * (1.0 - fabs(BC)) + EPS_MP);
        const double sig_d = DV(-A, (1.0 - fabs(Ck)) * (1.0 - fabs(AC)) + EPS_MP);
        const double sig_e = DV(AC, (1.0 - fabs(Ck)) * (1.0 - CC * fabs(A)) + EPS_MP) *
                             (DV(fabs(A), 1.0 - fabs(AC) + EPS_MP) + DV(2.0 * Ck, 1.0 - CC + EPS_MP));
        const double sig_f = DV(fabs(Ck) * AA, (1.0 - fabs(Ck)) * (1.0 - AA * fabs(Ck)) * (1.0 - fabs(AC)) + EPS_MP);
        const double w0 = sig_alfa * Z + sig_beta * ZZ + sig_gama * ZZ * Z + sig_a * YZ + sig_b * ZZ * Y +
                          sig_c * Z * YY + sig_d * XZ + sig_e * ZZ * X + sig_f * Z * XX;
        wa = fmin(fabs(w0), 1.0 * fabs(Wk)) * copysign(1.0, w0);
        if constexpr (MASK) wa = wa * rmk[a2];
      }
      Wa[aw] = wa;
    }
  }
}

#undef DV

// ------------------------------------------------------ K3: beta_up/dn ----
// MASK: every term of the extrema times mask_up = rmask (land values out of Tmax) / mask_dn = 1 on water, 1e20 on
// land (out of Tmin) of its column, mpdata_adiff.F:826-835.
template <bool MASK>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_mp_beta(const RomsDev *__restrict__ c, MpArgs m)
{
  DEV_PROLOGUE(c)
  const TileLv XB = decode_tile_level(b.Iendp1 - (b.IstrU - 1) + 1, b.Jendp1 - (b.JstrV - 1) + 1, N);
  if (!XB.valid) return;
  const int i = b.IstrU - 1 + XB.bx * BLK_X + threadIdx.x;
  const int j = b.JstrV - 1 + XB.by * BLK_Y + threadIdx.y;
  if (i > b.Iendp1 || j > b.Jendp1) return;
  const gcd_t Ta = (gcd_t)m.Ta, Ua = (gcd_t)m.Ua, Va = (gcd_t)m.Va, Wa = (gcd_t)m.Wa;
  const gcd_t t3 = (gcd_t)(c->F.t + (2L + 3L * (m.itrc - 1)) * n3r);
  const gd_t bup = (gd_t)m.bup, bdn = (gd_t)m.bdn;
  const long a2 = I2(i, j);
  {
    const int k = XB.k0 + 1;                                     // one thread per (i,j,k)
    const long a = a2 + (long)(k - 1) * nij;
    const long aw = a + nij;                                     // Wa(i,j,k); Wa(i,j,k-1) = Wa[a]
    const double T0 = Ta[a], Tw = Ta[a - 1], Te = Ta[a + 1], Ts = Ta[a - ni], Tn = Ta[a + ni];
    double Tmax, Tmin;
    if constexpr (!MASK) {
      Tmax = fmax(fmax(fmax(fmax(fmax(fmax(fmax(fmax(fmax(Tw, t3[a - 1]), T0), t3[a]), Te), t3[a + 1]), Ts),
                            t3[a - ni]), Tn), t3[a + ni]);
      Tmin = fmin(fmin(fmin(fmin(fmin(fmin(fmin(fmin(fmin(Tw, t3[a - 1]), T0), t3[a]), Te), t3[a + 1]), Ts),
                            t3[a - ni]), Tn), t3[a + ni]);
      if (k > 1) {
        Tmax = fmax(fmax(Tmax, Ta[a - nij]), t3[a - nij]);
        Tmin = fmin(fmin(Tmin, Ta[a - nij]), t3[a - nij]);
      }
      if (k < N) {
        Tmax = fmax(fmax(Tmax, Ta[a + nij]), t3[a + nij]);
        Tmin = fmin(fmin(Tmin, Ta[a + nij]), t3[a + nij]);
      }
    } else {
      const gcd_t rmk = (gcd_t)c->F.rmask;
      const double Large = 1.0E+20;
      auto mdn = [&](double r) { return fmax(1.0, fmin(Large, (1.0 - r) * Large)); };
      const double mu0 = rmk[a2], md0 = mdn(mu0);
      Tmax = T0 * mu0;
      Tmin = T0 * md0;
      auto take = [&](double mu, double md, double x) { Tmax = fmax(Tmax, x * mu); Tmin = fmin(Tmin, x * md); };
      take(mu0, md0, t3[a]);
      const long dq[4] = {-1, 1, -(long)ni, (long)ni};
      const double Tq[4] = {Tw, Te, Ts, Tn};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double mu = rmk[a2 + dq[q]], md = mdn(mu);
        take(mu, md, Tq[q]);
        take(mu, md, t3[a + dq[q]]);
      }
      if (k > 1) { take(mu0, md0, Ta[a - nij]); take(mu0, md0, t3[a - nij]); }
      if (k < N) { take(mu0, md0, Ta[a + nij]); take(mu0, md0, t3[a + nij]); }
    }
    const double ua0 = Ua[a], ua1 = Ua[a + 1], va0 = Va[a], va1 = Va[a + ni];
    double cff1 = Tw * fmax(0.0, ua0) - Te * fmin(0.0, ua1) + Ts * fmax(0.0, va0) - Tn * fmin(0.0, va1);
    double cff2 = T0 * fmax(0.0, ua1) - T0 * fmin(0.0, ua0) + T0 * fmax(0.0, va1) - T0 * fmin(0.0, va0);
    if (k == 1) {
      cff1 = cff1 - Ta[a + nij] * fmin(0.0, Wa[aw]);
      cff2 = cff2 + T0 * fmax(0.0, Wa[aw]);
    } else if (k < N) {
      cff1 = cff1 + Ta[a - nij] * fmax(0.0, Wa[a]) - Ta[a + nij] * fmin(0.0, Wa[aw]);
      cff2 = cff2 + T0 * fmax(0.0, Wa[aw]) - T0 * fmin(0.0, Wa[a]);
    } else {
      cff1 = cff1 + Ta[a - nij] * fmax(0.0, Wa[a]);
      cff2 = cff2 - T0 * fmin(0.0, Wa[a]);
    }
    bup[a] = (Tmax - T0) / (cff1 + EPS_MP);
    bdn[a] = (T0 - Tmin) / (cff2 + EPS_MP);
  }
}

// -------------------------- K4: limited transports, update, tridiagonal ----
// MASK: the limited transports times umask / vmask / rmask (mpdata_adiff.F:991, :1006, :1022) and the new tracer
// times rmask (step3d_t.F:1586-1596).
template <int NMAX, bool MASK>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_mp_update(const RomsDev *__restrict__ c, MpArgs m)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const double dt = c->p.dt;
  const gcd_t Ta = (gcd_t)m.Ta, Ua = (gcd_t)m.Ua, Va = (gcd_t)m.Va, Wa = (gcd_t)m.Wa;
  const gcd_t bup = (gcd_t)m.bup, bdn = (gcd_t)m.bdn;
  const gcd_t Hz = (gcd_t)c->F.Hz, z_r = (gcd_t)c->F.z_r;
  const int ltrc = m.itrc < b.NAT ? m.itrc : b.NAT;
  const gcd_t Akt = (gcd_t)(c->F.Akt + (long)(ltrc - 1) * n3w);
  const gd_t tn = (gd_t)(c->F.t + ((long)(m.nnew - 1) + 3L * (m.itrc - 1)) * n3r);
  const long a2 = I2(i, j);
  const double cffa = 1.0 / dt;                                   // mpdata_adiff.F:254
  const double cpp = dt * GF(pm)[a2] * GF(pn)[a2];
  const double omu0 = GF(om_u)[a2], omu1 = GF(om_u)[a2 + 1], onv0 = GF(on_v)[a2], onv1 = GF(on_v)[a2 + ni];
  const double onu0 = GF(on_u)[a2], onu1 = GF(on_u)[a2 + 1], omv0 = GF(om_v)[a2], omv1 = GF(om_v)[a2 + ni];
  const double omn = GF(omn)[a2];
  // physical edges of the limited transports, mpdata_adiff.F:1031-1100: zero (closed) or the limited transport of the
  // next face inside -- the other face of this cell
  const bool v0_wall = b.south_edge && !b.NSperiodic && j == b.Jstr;       // Va(i,Jstr)
  const bool v1_wall = b.north_edge && !b.NSperiodic && j == b.Jend;       // Va(i,Jend+1)
  const bool u0_wall = b.west_edge && !b.EWperiodic && i == b.Istr;        // Ua(Istr,j)
  const bool u1_wall = b.east_edge && !b.EWperiodic && i == b.Iend;        // Ua(Iend+1,j)
  auto lim_u = [&](long x, double om) {       // limited Ua at index x (face between x-1 and x), :1034-1040
    const double cff1 = fmin(fmin(bdn[x - 1], bup[x]), 1.0);
    const double cff2 = fmin(fmin(bup[x - 1], bdn[x]), 1.0);
    return (cff1 * fmax(0.0, Ua[x]) + cff2 * fmin(0.0, Ua[x])) * cffa * om;
  };
  auto lim_v = [&](long x, double on) {       // :1042-1049
    const double cff1 = fmin(fmin(bdn[x - ni], bup[x]), 1.0);
    const double cff2 = fmin(fmin(bup[x - ni], bdn[x]), 1.0);
    return (cff1 * fmax(0.0, Va[x]) + cff2 * fmin(0.0, Va[x])) * cffa * on;
  };
  double um0 = 1.0, um1 = 1.0, vm0 = 1.0, vm1 = 1.0, rm0 = 1.0;
  if constexpr (MASK) {
    um0 = GF(umask)[a2]; um1 = GF(umask)[a2 + 1]; vm0 = GF(vmask)[a2]; vm1 = GF(vmask)[a2 + ni]; rm0 = GF(rmask)[a2];
  }
  double DCm[NMAX + 1];      // right-hand side / solution
  double CFm[NMAX + 1];
  double FCm1 = 0.0;         // corrected vertical flux through the bottom face
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const long a = a2 + (long)(k - 1) * nij;
      const double T0 = Ta[a], hz = Hz[a];
      double u0 = lim_u(a, omu0), u1 = lim_u(a + 1, omu1);
      double v0 = lim_v(a, onv0), v1 = lim_v(a + ni, onv1);
      if constexpr (MASK) { u0 = u0 * um0; u1 = u1 * um1; v0 = v0 * vm0; v1 = v1 * vm1; }
      {
        const double u0i = u0, u1i = u1, v0i = v0, v1i = v1;
        if (u0_wall) u0 = m.closed[LBS_WEST] ? 0.0 : u1i;
        if (u1_wall) u1 = m.closed[LBS_EAST] ? 0.0 : u0i;
        if (v0_wall) v0 = m.closed[LBS_SOUTH] ? 0.0 : v1i;
        if (v1_wall) v1 = m.closed[LBS_NORTH] ? 0.0 : v0i;
      }
      // corrected horizontal fluxes, step3d_t.F:1238-1255
      const double FXi = (fmax(u0, 0.0) * Ta[a - 1] + fmin(u0, 0.0) * T0) * 0.5 * (hz + Hz[a - 1]) * onu0;
      const double FXip1 = (fmax(u1, 0.0) * T0 + fmin(u1, 0.0) * Ta[a + 1]) * 0.5 * (Hz[a + 1] + hz) * onu1;
      const double FEj = (fmax(v0, 0.0) * Ta[a - ni] + fmin(v0, 0.0) * T0) * 0.5 * (hz + Hz[a - ni]) * omv0;
      const double FEjp1 = (fmax(v1, 0.0) * T0 + fmin(v1, 0.0) * Ta[a + ni]) * 0.5 * (Hz[a + ni] + hz) * omv1;
      const double cff1 = cpp * (FXip1 - FXi);
      const double cff2 = cpp * (FEjp1 - FEj);
      const double cff3 = cff1 + cff2;
      double tv = T0 * hz - cff3;                                 // :1265
      // corrected vertical flux through the top face, :1281-1290 with the limited Wa (:1051-1060)
      double FCk = 0.0;
      if (k < N) {
        const long aw = a + nij;
        const double c1 = fmin(fmin(bdn[a], bup[a + nij]), 1.0);
        const double c2 = fmin(fmin(bup[a], bdn[a + nij]), 1.0);
        double w = (c1 * fmax(0.0, Wa[aw]) + c2 * fmin(0.0, Wa[aw])) * cffa * omn * (z_r[a + nij] - z_r[a]);
        if constexpr (MASK) w = w * rm0;
        FCk = fmax(w, 0.0) * T0 + fmin(w, 0.0) * Ta[a + nij];
      }
      tv = tv - cpp * (FCk - FCm1);                               // :1305 (m Tunits)
      FCm1 = FCk;
      DCm[k] = tv;
    }
  }
  // classic tridiagonal, step3d_t.F:1431-1501
  const double cfl = -dt * c->p.lambda;
  double FCprev = 0.0;       // FC(k-1)
  double CFprev = 0.0, DCprev = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const long a = a2 + (long)(k - 1) * nij;
      double FCk = 0.0;
      if (k < N) {
        const double cff1 = 1.0 / (z_r[a + nij] - z_r[a]);
        FCk = cfl * cff1 * Akt[a + nij];                          // Akt(i,j,k)
      }
      const double BCk = Hz[a] - FCk - FCprev;
      if (k == 1) {
        const double cff = 1.0 / BCk;
        CFm[1] = cff * FCk;
        DCm[1] = cff * DCm[1];
      } else if (k < N) {
        const double cff = 1.0 / (BCk - FCprev * CFprev);
        CFm[k] = cff * FCk;
        DCm[k] = cff * (DCm[k] - FCprev * DCprev);
      } else {
        DCm[k] = (DCm[k] - FCprev * DCprev) / (BCk - FCprev * CFprev);
      }
      CFprev = CFm[k < N ? k : 1];
      DCprev = DCm[k];
      FCprev = FCk;
    }
  }
  double up = 0.0;
#pragma unroll
  for (int k = NMAX; k >= 1; k--) {
    if (k <= N) {
      const long a = a2 + (long)(k - 1) * nij;
      double v;
      if (k == N) v = DCm[k];
      else v = DCm[k] - CFm[k] * up;
      up = v;
      if constexpr (MASK) v = v * rm0;
      tn[a] = v;
    }
  }
}

}  // namespace

// One MPDATA tracer of step3d_t; called by roms_hip_step3d_t (k_step3d_t.hip).
int roms_launch_step3d_t_mpdata(int nnew, int itrc, int first)
{
  const roms_bounds_t &b = g_ctx.b;
  if (b.NghostPoints != 3) return roms_fail("roms_hip_step3d_t", "MPDATA needs NghostPoints = 3 (inp_par.F:266-278)");
  if (b.N > ROMS_MAXN) return roms_fail("roms_hip_step3d_t", "N > 64 not instantiated");
  int rc;
  const long n3r = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * b.N;
  // three-point footprint: refresh the ghost points of t(nnew) first, step3d_t.F:369-386
  if ((rc = halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_t] + ((long)(nnew - 1) + 3L * (itrc - 1)) * n3r))) return rc;
  MpArgs m;
  m.Ta = g_ctx.hostc.ws3[1]; m.Ua = g_ctx.hostc.ws3[2]; m.Va = g_ctx.hostc.ws3[3]; m.Wa = g_ctx.hostc.ws3[4];
  m.bup = g_ctx.hostc.ws3[5]; m.bdn = g_ctx.hostc.ws3[6];
  m.oHz = g_ctx.hostc.ws3[0]; m.odz = g_ctx.hostc.ws3[7];
  m.nnew = nnew; m.itrc = itrc;
  m.closed[LBS_WEST] = lbc_code(g_ctx.p, LBS_WEST, LBV_U) == LBC_CLOSED;
  m.closed[LBS_EAST] = lbc_code(g_ctx.p, LBS_EAST, LBV_U) == LBC_CLOSED;
  m.closed[LBS_SOUTH] = lbc_code(g_ctx.p, LBS_SOUTH, LBV_V) == LBC_CLOSED;
  m.closed[LBS_NORTH] = lbc_code(g_ctx.p, LBS_NORTH, LBV_V) == LBC_CLOSED;
  if (first) {
    hipLaunchKernelGGL(k_mp_metrics, grid2d(b.UBi - b.LBi + 1, b.UBj - b.LBj + 1), block2d(), 0, g_ctx.stream, g_ctx.devc,
                       g_ctx.hostc.ws3[0], g_ctx.hostc.ws3[7]);
    KERNEL_CHECK("k_mp_metrics");
  }
  {
    dim3 g3 = grid_tile_level(b.Iendp2i - b.IstrUm2 + 1, b.Jendp2i - b.JstrVm2 + 1, b.N);
    hipLaunchKernelGGL(k_mp_ta, g3, block2d(), 0, g_ctx.stream, g_ctx.devc, m);
  }
  KERNEL_CHECK("k_mp_ta");
  {
    dim3 g3 = grid_tile_level(b.Iendp2 - (b.IstrU - 1) + 1, b.Jendp2 - (b.JstrV - 1) + 1, b.N);
    void (*kern)(const RomsDev *, MpArgs) = g_ctx.p.masking ? (g_ctx.p.mpdata_fast ? k_mp_adiff<true, true> : k_mp_adiff<false, true>)
                                : (g_ctx.p.mpdata_fast ? k_mp_adiff<true, false> : k_mp_adiff<false, false>);
    hipLaunchKernelGGL(kern, g3, block2d(), 0, g_ctx.stream, g_ctx.devc, m);
  }
  KERNEL_CHECK("k_mp_adiff");
  {
    dim3 g3 = grid_tile_level(b.Iendp1 - (b.IstrU - 1) + 1, b.Jendp1 - (b.JstrV - 1) + 1, b.N);
    void (*beta)(const RomsDev *, MpArgs) = g_ctx.p.masking ? k_mp_beta<true> : k_mp_beta<false>;
    hipLaunchKernelGGL(beta, g3, block2d(), 0, g_ctx.stream, g_ctx.devc, m);
  }
  KERNEL_CHECK("k_mp_beta");
  const dim3 g = grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1);
  const bool mk = g_ctx.p.masking != 0;
  void (*upd)(const RomsDev *, MpArgs);
  if (b.N <= 16) upd = mk ? k_mp_update<16, true> : k_mp_update<16, false>;
  else if (b.N <= 32) upd = mk ? k_mp_update<32, true> : k_mp_update<32, false>;
  else upd = mk ? k_mp_update<ROMS_MAXN, true> : k_mp_update<ROMS_MAXN, false>;
  hipLaunchKernelGGL(upd, g, block2d(), 0, g_ctx.stream, g_ctx.devc, m);
  KERNEL_CHECK("k_mp_update");
  return 0;
}
