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
  // k_mp_ta steps nb consecutive tracers (itrc, itrc+1 ..) in one launch, Ta of tracer itrc+q -> TaB[q]: the mass fluxes
  // and Hz (4 of the 7 arrays a tracer's upstream step reads or writes) cross the memory bus once per batch
  double *TaB[3];
  int nb;
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
  const gcd_t Huon = (gcd_t)c->F.Huon, Hvom = (gcd_t)c->F.Hvom, Wv = (gcd_t)c->F.W, Hz = (gcd_t)c->F.Hz;
  const long c0 = I2(i, j);
  const double cff = dt * GF(pm)[c0] * GF(pn)[c0];
  const bool s_wall = b.south_edge && !b.NSperiodic && j == b.Jstr;
  const bool n_wall = b.north_edge && !b.NSperiodic && j == b.Jend;
  const bool w_wall = b.west_edge && !b.EWperiodic && i == b.Istr;
  const bool e_wall = b.east_edge && !b.EWperiodic && i == b.Iend;
  const int k = XB.k0 + 1;                                        // one thread per (i,j,k)
  const long a = c0 + (long)(k - 1) * nij;
  const bool src_cell = c->src.n > 0 && src_cell_any(c, c0, ni);
  // the transports and the thickness of the cell: the same for every tracer of the batch
  const double hu0 = Huon[a], hu1 = Huon[a + 1], hv0 = Hvom[a], hv1 = Hvom[a + ni];
  const double w0 = (k > 1) ? (double)Wv[a] : 0.0, w1 = (k < N) ? (double)Wv[a + nij] : 0.0, ohz = 1.0 / Hz[a];
  for (int q = 0; q < m.nb; q++) {
    const int itrc = m.itrc + q;
    const gcd_t t3 = (gcd_t)(c->F.t + (2L + 3L * (itrc - 1)) * n3r);
    const gcd_t tn = (gcd_t)(c->F.t + ((long)(m.nnew - 1) + 3L * (itrc - 1)) * n3r);
    const gd_t Ta = (gd_t)m.TaB[q];
    const double t0 = t3[a];
    // FC(k-1): the expression the level below evaluates as its FC(k)
    const double FCm1 = (k > 1) ? upstream(w0, t3[a - nij], t0) : 0.0;
    double FXi = upstream(hu0, t3[a - 1], t0);
    double FXip1 = upstream(hu1, t0, t3[a + 1]);
    double FEj = upstream(hv0, t3[a - ni], t0);
    double FEjp1 = upstream(hv1, t0, t3[a + ni]);
    if (src_cell)                                    // LuvSrc, step3d_t.F:734-799 (on the extended range of MPDATA)
      src_cell_fluxes<false>(c, c0, a, ni, k, itrc, c->F.t + (2L + 3L * (itrc - 1)) * n3r, FXi, FXip1, FEj, FEjp1);
    const double cff1 = cff * (FXip1 - FXi);
    const double cff2 = cff * (FEjp1 - FEj);
    const double cff3 = cff1 + cff2;
    double ta = tn[a] - cff3;                                   // step3d_t.F:838
    // LwSrc, :1136-1158: on Istr:Iend+1, Jstr:Jend+1 only, not on the rest of MPDATA's extended range
    if (src_cell && i >= b.Istr && i <= b.Iend + 1 && j >= b.Jstr && j <= b.Jend + 1)
      ta = src_w_tracer(c, c0, k, itrc, cff, t0, ta);
    const double FCk = (k < N) ? upstream(w1, t0, t3[a + nij]) : 0.0;   // :1006-1018
    const double c1 = cff * (FCk - FCm1);
    ta = (ta - c1) * ohz;                                       // :1175
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

// Reciprocal of the FAST variant (roms_params_t.mpdata_fast): v_rcp_f64 refined by one Newton step (explicit FMAs) --
// one reciprocal per distinct denominator, the compiler merges the quotients of a face that share one.  The IEEE
// division costs ~11 FP64 instructions on gfx950 (two of them quarter rate) and this kernel evaluates 45 quotients
// per cell; validated against the oracle at the north-star bound (1e-10 relative RMS after 100 steps,
// tests/test_gpu_mpdata.py).  The exact variant divides as the reference does (bit-identical results).
__device__ __forceinline__ double mp_rcp(double d)
{
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}

// ------------------------------------------------- K2: Ua, Va, Wa (raw) ----
// LDS-tiled and level-marching.  A workgroup of 64 x 4 threads owns the faces of 64 x 4 columns and walks from the
// bottom level to the top.  Everything the three faces of a cell need from the cells around it is a handful of *cell
// quantities* that the reference evaluates again and again inside every face (mpdata_adiff.F:262-560):
//
//   GU(i,j,k) = (Ta(i) - Ta(i-1)) * (pm(i-1) + pm(i)) [* umask(i)]      xi-gradient on the u-face of the cell
//   GV(i,j,k) = (Ta(j) - Ta(j-1)) * (pn(j-1) + pn(j)) [* vmask(j)]      eta-gradient on its v-face
//   HU(i,j,k) = Huon * (pm(i-1)+pm(i)) * (pn(i-1)+pn(i)) * (oHz(i-1)+oHz(i))    the Courant-number term of its u-face
//   HV(i,j,k) = Hvom * (pm(j-1)+pm(j)) * (pn(j-1)+pn(j)) * (oHz(j-1)+oHz(j))
//   DZ(i,j,k) = (Ta(k+1) - Ta(k)) * odz(k)                              vertical gradient on its w-face
//   WC(i,j,k) = (W(k-1)*odz(k-1) + W(k)*odz(k)) * pm * pn               (one term at k = 1 and k = N)
//
// -- the same operations in the same order wherever the reference uses them, so sharing them keeps every bit.  Per
// level each thread evaluates them once for its own column (and the first 140 threads for one column of the one-point
// ring around the tile as well), publishes them in LDS, and after a barrier every thread combines its own values with
// those of its W, S, E, N, NW and SE neighbours into Ua, Va and Wa.  Against the one-thread-per-cell kernel of round 2
// (which re-derived all of this from ~150 global loads per cell) that is ~15 global loads + 28 LDS reads per cell and
// roughly a third fewer FP64 instructions.
//
// LDS plane A_k (published in iteration k; two planes alive: A_k and A_k-1):
//   at level k+1: Ta, GU, GV, HU, HV, ZU = z_r(min(k+1,N));   at level k: DZ, WC;   ZL = z_r(max(k-1,1))
#define MPX 66
#ifndef MP_ATY
#define MP_ATY 4          // rows per workgroup of k_mp_adiff (64 x 8 was measured 10 % slower: fewer, larger workgroups)
#endif
#define MPY (MP_ATY + 2)
#define MPC (MPX * MPY)
enum { Q_TA = 0, Q_GU, Q_GV, Q_HU, Q_HV, Q_ZU, Q_DZ, Q_WC, Q_ZL, Q_N };

struct MpCol {          // level-independent part of a column
  long a2;              // index in a 2-D array
  long dW, dS;          // distance to the western / southern neighbour (0 where that would leave the array: value unused)
  double PMU, PNU, PMV, PNV, pm, pn, um, vm, umw, vmw;
  bool ok;              // the column lies inside LBi:UBi, LBj:UBj
};

struct MpLev {          // what a thread keeps of its own column for one level
  double Ta, GU, GV, HU, HV;        // cell quantities
  double Hu, Hv, OHU, OHV;          // raw Huon, Hvom and the oHz sums (the face's own Courant number keeps the reference's
                                    // association 0.125*Huon*dt*..., which is not HU)
};

template <bool MASK>
__device__ __forceinline__ MpCol mp_column(const RomsDev *__restrict__ c, int i, int j)
{
  DEV_PROLOGUE(c)
  MpCol cc;
  cc.ok = i >= b.LBi && i <= b.UBi && j >= b.LBj && j <= b.UBj;
  const int ic = cc.ok ? i : b.LBi, jc = cc.ok ? j : b.LBj;
  cc.a2 = I2(ic, jc);
  cc.dW = ic > b.LBi ? 1 : 0;
  cc.dS = jc > b.LBj ? ni : 0;
  const gcd_t pm = (gcd_t)c->F.pm, pn = (gcd_t)c->F.pn;
  cc.pm = pm[cc.a2]; cc.pn = pn[cc.a2];
  const double pmW = pm[cc.a2 - cc.dW], pnW = pn[cc.a2 - cc.dW], pmS = pm[cc.a2 - cc.dS], pnS = pn[cc.a2 - cc.dS];
  cc.PMU = pmW + cc.pm; cc.PNU = pnW + cc.pn;
  cc.PMV = cc.pm + pmS; cc.PNV = cc.pn + pnS;
  cc.um = 1.0; cc.vm = 1.0; cc.umw = 1.0; cc.vmw = 1.0;
  // umw / vmw: the face masks of the limited velocities, times the wet/dry mask under WET_DRY (mpdata_adiff.F:394-399,
  // :567-572); um / vm: the land/sea masks alone, as the gradients of the cross terms take them
  if constexpr (MASK) { cc.um = GF(umask)[cc.a2]; cc.vm = GF(vmask)[cc.a2]; cc.umw = umaskw(c, cc.a2); cc.vmw = vmaskw(c, cc.a2); }
  return cc;
}

struct MpF { gcd_t Ta, oHz, odz, Huon, Hvom, W, z_r; };

// what plane A_k of one column needs from memory: level k+1 of Ta (own, west, south), oHz (own, west, south), Huon,
// Hvom, z_r and level k of odz and W.  Loaded one iteration ahead (the loads are in flight while the faces of the
// level below are evaluated: the level loop is bound by memory latency, not bytes, like k_step3d_t_pipe).
struct MpRaw { double t1, tW, tS, o, oW, oS, hu, hv, zu, odz, w; };

__device__ __forceinline__ MpRaw mp_load_raw(const MpF &f, const MpCol &cc, int k, int N, long nij)
{
  MpRaw r;
  // k = N: nothing new is needed (plane A_N has no level N+1); re-read level N, the values are not used
  const long a1 = cc.a2 + (long)(k < N ? k : N - 1) * nij;          // level k+1
  const long a0 = cc.a2 + (long)(k >= 1 ? (k < N ? k - 1 : N - 2) : 0) * nij;   // level k (odz has levels 1..N-1)
  r.t1 = f.Ta[a1]; r.tW = f.Ta[a1 - cc.dW]; r.tS = f.Ta[a1 - cc.dS];
  r.o = f.oHz[a1]; r.oW = f.oHz[a1 - cc.dW]; r.oS = f.oHz[a1 - cc.dS];
  r.hu = f.Huon[a1]; r.hv = f.Hvom[a1];
  r.zu = f.z_r[a1];
  r.odz = f.odz[a0];
  r.w = f.W[a1];                                        // W(i,j,k) of a (0:N) array = rho index of level k+1
  return r;
}

struct MpSlide { double Tk, WZm, zk, zkm1; };    // Ta(k), W(k-1)*odz(k-1), z_r(k), z_r(k-1) of a column

// plane A_k of one column from its raw loads and what it carries from the level below: q[] for the LDS, L = level
// k+1 of the column for its owner, WZk = W(k)*odz(k)
template <bool MASK>
__device__ __forceinline__ void mp_cellq(const MpRaw &r, const MpCol &cc, MpSlide &s, int k, int N, double q[Q_N], MpLev &L,
                                         double &WZk)
{
  // level k+1 (at k = N the raw loads repeat level N: those values are never used -- there is no W face above level N)
  L.Hu = r.hu; L.Hv = r.hv;
  L.Ta = r.t1;
  L.GU = (r.t1 - r.tW) * cc.PMU;
  L.GV = (r.t1 - r.tS) * cc.PNV;
  if constexpr (MASK) { L.GU = L.GU * cc.um; L.GV = L.GV * cc.vm; }
  L.OHU = r.oW + r.o;
  L.OHV = r.oS + r.o;
  L.HU = L.Hu * cc.PMU * cc.PNU * L.OHU;
  L.HV = L.Hv * cc.PMV * cc.PNV * L.OHV;
  q[Q_ZU] = k < N ? r.zu : s.zk;
  q[Q_TA] = L.Ta; q[Q_GU] = L.GU; q[Q_GV] = L.GV; q[Q_HU] = L.HU; q[Q_HV] = L.HV;
  // level k (DZ and WZk are not used at k = N, nothing of this at k = 0)
  q[Q_DZ] = (r.t1 - s.Tk) * r.odz;
  WZk = r.w * r.odz;
  q[Q_WC] = (k == 1 ? WZk : (k < N ? (s.WZm + WZk) : s.WZm)) * cc.pm * cc.pn;
  q[Q_ZL] = k > 1 ? s.zkm1 : s.zk;
  // slide to the next level
  s.Tk = r.t1; s.WZm = WZk;
  if (k >= 1) s.zkm1 = s.zk;
  if (k < N) s.zk = r.zu;
}

// the nine third-order coefficients and the polynomial of one face; P = the gradient factor along the face's own
// direction, Q, R the other two, Xp, Xq, Xr the matching first-order velocities.  SWAPBC: the eta face of the
// reference multiplies sig_b with Xp*Xq^2 and sig_c with Xp^2*Xq (mpdata_adiff.F:556-560), the other two the
// other way round (:376-380, :786-790).
#define MP_SIGMA_EXACT(P, Q, R, Xp, Xq, Xr, OUT, SWAPBC)                                                                    \
  {                                                                                                                   \
    const double PP = P * P, QQ = Q * Q, RR = R * R, PQ = P * Q, PR = P * R;                                          \
    const double XpXp = Xp * Xp, XqXq = Xq * Xq, XrXr = Xr * Xr, XpXq = Xp * Xq, XpXr = Xp * Xr;                      \
    const double sig_alfa = DV(1.0, 1.0 - fabs(P) + EPS_MP);                                                          \
    const double sig_beta = DV(-P, (1.0 - fabs(P)) * (1.0 - PP) + EPS_MP);                                            \
    const double sig_gama = DV(2.0 * fabs(PP * P), (1.0 - fabs(P)) * (1.0 - PP) * (1.0 - fabs(PP * P)) + EPS_MP);     \
    const double sig_a = DV(-Q, (1.0 - fabs(P)) * (1.0 - fabs(PQ)) + EPS_MP);                                         \
    const double sig_b = DV(PQ, (1.0 - fabs(P)) * (1.0 - PP * fabs(Q)) + EPS_MP) *                                    \
                         (DV(fabs(Q), 1.0 - fabs(PQ) + EPS_MP) + DV(2.0 * P, 1.0 - PP + EPS_MP));                     \
    const double sig_c = DV(fabs(P) * QQ, (1.0 - fabs(P)) * (1.0 - QQ * fabs(P)) * (1.0 - fabs(PQ)) + EPS_MP);        \
    const double sig_d = DV(-R, (1.0 - fabs(P)) * (1.0 - fabs(PR)) + EPS_MP);                                         \
    const double sig_e = DV(PR, (1.0 - fabs(P)) * (1.0 - PP * fabs(R)) + EPS_MP) *                                    \
                         (DV(fabs(R), 1.0 - fabs(PR) + EPS_MP) + DV(2.0 * P, 1.0 - PP + EPS_MP));                     \
    const double sig_f = DV(fabs(P) * RR, (1.0 - fabs(P)) * (1.0 - RR * fabs(P)) * (1.0 - fabs(PR)) + EPS_MP);        \
    if (SWAPBC)                                                                                                       \
      OUT = sig_alfa * Xp + sig_beta * XpXp + sig_gama * XpXp * Xp + sig_a * XpXq + sig_b * Xp * XqXq +               \
            sig_c * XpXp * Xq + sig_d * XpXr + sig_e * XpXp * Xr + sig_f * Xp * XrXr;                                 \
    else                                                                                                              \
      OUT = sig_alfa * Xp + sig_beta * XpXp + sig_gama * XpXp * Xp + sig_a * XpXq + sig_b * XpXp * Xq +               \
            sig_c * Xp * XqXq + sig_d * XpXr + sig_e * XpXp * Xr + sig_f * Xp * XrXr;                                 \
  }

// FAST variant: the twelve denominators of a face are inverted together (prefix products, ONE refined reciprocal,
// then the individual inverses by back-multiplication: 33 multiplications instead of eleven more reciprocals).
// Every denominator carries the reference's +eps and lies in [1e-18, ~2], so the product stays in range.
template <int NQ>
__device__ __forceinline__ void mp_batch_rcp(const double (&d)[NQ], double (&inv)[NQ])
{
  double pre[NQ];
  pre[0] = d[0];
#pragma unroll
  for (int q = 1; q < NQ; q++) pre[q] = pre[q - 1] * d[q];
  double r = mp_rcp(pre[NQ - 1]);
#pragma unroll
  for (int q = NQ - 1; q >= 1; q--) {
    inv[q] = r * pre[q - 1];
    r = r * d[q];
  }
  inv[0] = r;
}
#define MP_SIGMA_FAST(P, Q, R, Xp, Xq, Xr, OUT, SWAPBC)                                                               \
  {                                                                                                                   \
    const double PP = P * P, QQ = Q * Q, RR = R * R, PQ = P * Q, PR = P * R;                                          \
    const double XpXp = Xp * Xp, XqXq = Xq * Xq, XrXr = Xr * Xr, XpXq = Xp * Xq, XpXr = Xp * Xr;                      \
    const double omP = 1.0 - fabs(P), omPP = 1.0 - PP, omPQ = 1.0 - fabs(PQ), omPR = 1.0 - fabs(PR);                  \
    const double dd[12] = {omP + EPS_MP, omP * omPP + EPS_MP, omP * omPP * (1.0 - fabs(PP * P)) + EPS_MP,             \
                           omP * omPQ + EPS_MP, omP * (1.0 - PP * fabs(Q)) + EPS_MP, omPQ + EPS_MP, omPP + EPS_MP,    \
                           omP * (1.0 - QQ * fabs(P)) * omPQ + EPS_MP, omP * omPR + EPS_MP,                           \
                           omP * (1.0 - PP * fabs(R)) + EPS_MP, omPR + EPS_MP,                                        \
                           omP * (1.0 - RR * fabs(P)) * omPR + EPS_MP};                                               \
    double iv[12];                                                                                                    \
    mp_batch_rcp<12>(dd, iv);                                                                                         \
    const double sig_alfa = iv[0];                                                                                    \
    const double sig_beta = -P * iv[1];                                                                               \
    const double sig_gama = 2.0 * fabs(PP * P) * iv[2];                                                               \
    const double sig_a = -Q * iv[3];                                                                                  \
    const double sig_b = PQ * iv[4] * (fabs(Q) * iv[5] + 2.0 * P * iv[6]);                                            \
    const double sig_c = fabs(P) * QQ * iv[7];                                                                        \
    const double sig_d = -R * iv[8];                                                                                  \
    const double sig_e = PR * iv[9] * (fabs(R) * iv[10] + 2.0 * P * iv[6]);                                           \
    const double sig_f = fabs(P) * RR * iv[11];                                                                       \
    if (SWAPBC)                                                                                                       \
      OUT = sig_alfa * Xp + sig_beta * XpXp + sig_gama * XpXp * Xp + sig_a * XpXq + sig_b * Xp * XqXq +               \
            sig_c * XpXp * Xq + sig_d * XpXr + sig_e * XpXp * Xr + sig_f * Xp * XrXr;                                 \
    else                                                                                                              \
      OUT = sig_alfa * Xp + sig_beta * XpXp + sig_gama * XpXp * Xp + sig_a * XpXq + sig_b * XpXp * Xq +               \
            sig_c * Xp * XqXq + sig_d * XpXr + sig_e * XpXp * Xr + sig_f * Xp * XrXr;                                 \
  }

template <bool FAST, bool MASK>
__global__ void __launch_bounds__(BLK_X *MP_ATY, 2)
k_mp_adiff(const RomsDev *__restrict__ c, MpArgs m)
{
  DEV_PROLOGUE(c)
  __shared__ double lds[2][Q_N][MPC];
  // union of the three ranges: i = IstrU-1 : Iendp2, j = JstrV-1 : Jendp2
  const Blk XB = xcd_block();
  const int i0 = b.IstrU - 1 + XB.x * BLK_X, j0 = b.JstrV - 1 + XB.y * MP_ATY;
  const int i = i0 + threadIdx.x, j = j0 + threadIdx.y;
  const int tid = threadIdx.y * BLK_X + threadIdx.x;
  const bool inr = i <= b.Iendp2 && j <= b.Jendp2;
  const bool do_u = inr && j <= b.Jendp1;                                   // Ua: j = JstrV-1:Jendp1, i = IstrU-1:Iendp2
  const bool do_v = inr && j >= b.JstrVm1 && i <= b.Iendp1;                 // Va: j = JstrVm1:Jendp2, i = IstrU-1:Iendp1
  const bool do_w = inr && j <= b.Jendp1 && i <= b.Iendp1;                  // Wa: j = JstrV-1:Jendp1, i = IstrU-1:Iendp1
  const double dt = c->p.dt;
  MpF f;
  f.Ta = (gcd_t)m.Ta; f.oHz = (gcd_t)m.oHz; f.odz = (gcd_t)m.odz;
  f.Huon = (gcd_t)c->F.Huon; f.Hvom = (gcd_t)c->F.Hvom; f.W = (gcd_t)c->F.W; f.z_r = (gcd_t)c->F.z_r;
  const gd_t Ua = (gd_t)m.Ua, Va = (gd_t)m.Va, Wa = (gd_t)m.Wa;
  // own column: slot (tx+1, ty+1) of the 66 x 6 plane
  const int so = (threadIdx.y + 1) * MPX + threadIdx.x + 1;
  const MpCol co = mp_column<MASK>(c, i, j);
  // ring column of this thread (the first 140 threads): rows 0 and 5, then columns 0 and 65 of rows 1..4
  int sh = -1, ih = 0, jh = 0;
  if (tid < 2 * MPX) { const int r = tid / MPX, x = tid - r * MPX; sh = (r ? (MPY - 1) * MPX : 0) + x; ih = i0 - 1 + x; jh = j0 - 1 + (r ? MPY - 1 : 0); }
  else if (tid < 2 * MPX + 2 * MP_ATY) { const int q = tid - 2 * MPX, y = 1 + (q >> 1), x = (q & 1) ? MPX - 1 : 0; sh = y * MPX + x; ih = i0 - 1 + x; jh = j0 - 1 + y; }
  MpCol ch = co;
  if (sh >= 0) ch = mp_column<MASK>(c, ih, jh);
  const bool halo = sh >= 0 && ch.ok;
  // level-independent sums of the faces (mpdata_adiff.F:300, :478, :656-657)
  const gcd_t on_v = (gcd_t)c->F.on_v, om_u = (gcd_t)c->F.om_u;
  const long a2 = co.a2;
  double ONV4 = 0.0, OMU4 = 0.0, OMU2 = 0.0, ONV2 = 0.0, rm = 1.0;
  if (do_u) ONV4 = on_v[a2] + on_v[a2 + ni] + on_v[a2 - 1] + on_v[a2 - 1 + ni];
  if (do_v) OMU4 = om_u[a2] + om_u[a2 + 1] + om_u[a2 - ni] + om_u[a2 + 1 - ni];
  if (do_w) { OMU2 = om_u[a2 + 1] + om_u[a2]; ONV2 = on_v[a2 + ni] + on_v[a2]; }
  if constexpr (MASK) { if (inr) rm = rmaskw(c, a2); }           // Wa's mask (+ WET_DRY, mpdata_adiff.F:800-805)
  // faces on a physical edge, mpdata_adiff.F:577-640: zero (closed) or the value of the next face inside, which the
  // thread of that face stores
  const bool v_wall_n = b.north_edge && !b.NSperiodic && j == b.Jend + 1;
  const bool u_wall_w = b.west_edge && !b.EWperiodic && i == b.Istr;
  const bool u_wall_e = b.east_edge && !b.EWperiodic && i == b.Iend + 1;
  const bool u_copy_w = b.west_edge && !b.EWperiodic && i == b.Istr + 1 && !m.closed[LBS_WEST];
  const bool u_copy_e = b.east_edge && !b.EWperiodic && i == b.Iend && !m.closed[LBS_EAST];
  const bool v_copy_s = b.south_edge && !b.NSperiodic && j == b.Jstr + 1;
  const bool v_copy_n = b.north_edge && !b.NSperiodic && j == b.Jend && !m.closed[LBS_NORTH];
  // a thread without a ring column repeats its own one there (valid addresses, nothing stored): no divergent loads
  const MpCol chh = halo ? ch : co;
  // plane A_0: level 1
  MpLev Lc, Ln;              // own column at level k and k+1
  double DZm1 = 0.0;         // own DZ(k-1)
  MpSlide so_s{0.0, 0.0, 0.0, 0.0}, sh_s{0.0, 0.0, 0.0, 0.0};
  {
    const MpRaw r0 = mp_load_raw(f, co, 0, N, nij), r0h = mp_load_raw(f, chh, 0, N, nij);
    double q[Q_N], wz;
    MpLev Lh;
    mp_cellq<MASK>(r0, co, so_s, 0, N, q, Lc, wz);
#pragma unroll
    for (int e = 0; e < Q_N; e++) lds[0][e][so] = q[e];
    mp_cellq<MASK>(r0h, chh, sh_s, 0, N, q, Lh, wz);
    if (sh >= 0) {
#pragma unroll
      for (int e = 0; e < Q_N; e++) lds[0][e][sh] = q[e];
    }
  }
  MpRaw rn = mp_load_raw(f, co, 1, N, nij), rnh = mp_load_raw(f, chh, 1, N, nij);
  __syncthreads();
  for (int k = 1; k <= N; k++) {
    const int cu = k & 1, pv = cu ^ 1;
    double qo[Q_N], WZk;
    // ---- phase 1: plane A_k from the loads issued one iteration ago; then the loads of plane A_k+1 go out
    {
      double q[Q_N], wz;
      MpLev Lh;
      mp_cellq<MASK>(rn, co, so_s, k, N, qo, Ln, WZk);
#pragma unroll
      for (int e = 0; e < Q_N; e++) lds[cu][e][so] = qo[e];
      mp_cellq<MASK>(rnh, chh, sh_s, k, N, q, Lh, wz);
      if (sh >= 0) {
#pragma unroll
        for (int e = 0; e < Q_N; e++) lds[cu][e][sh] = q[e];
      }
    }
    if (k < N) { rn = mp_load_raw(f, co, k + 1, N, nij); rnh = mp_load_raw(f, chh, k + 1, N, nij); }
    __syncthreads();
    // ---- phase 2: the three faces of (i,j,k)
    const long a = a2 + (long)(k - 1) * nij;
    const double T0 = Lc.Ta;
    const double DZk = qo[Q_DZ], WCk = qo[Q_WC], ZU = qo[Q_ZU], ZL = qo[Q_ZL];
    if constexpr (FAST) {
#pragma clang fp contract(fast)
#define DV(n, d) ((n) * mp_rcp(d))
#define MP_SIGMA MP_SIGMA_FAST
#include "k_mpdata_faces.inc"
#undef MP_SIGMA
#undef DV
    } else {
#define DV(n, d) ((n) / (d))
#define MP_SIGMA MP_SIGMA_EXACT
#include "k_mpdata_faces.inc"
#undef MP_SIGMA
#undef DV
    }
    Lc = Ln;
    DZm1 = DZk;
    __syncthreads();        // the plane of iteration k-1 is overwritten next
  }
}

// --------------- K3: FCT limiter, limited transports, update, tridiagonal ----
// mpdata_adiff.F:842-1100 (beta_up / beta_dn, the limited Ua, Va, Wa) + step3d_t.F:1235-1316 (corrected advection of
// Ta) + :1431-1501 (classic tridiagonal vertical diffusion) in ONE level-marching kernel: the limiter factors of a
// level live in LDS only (66 x (TY+2) cells: the tile and a one-cell ring, evaluated by the tile's threads), so
// beta_up / beta_dn are never written to memory, and Ta, Ua, Va, Wa are read once instead of twice.
//   iteration k:  beta(k) of own + ring cell -> LDS | barrier | limited horizontal fluxes of level k (neighbours'
//                 beta from LDS), limited vertical flux through the face below (own beta(k-1), beta(k)), which
//                 completes level k-1: forward elimination of the tridiagonal for k-1.
// MASK: every term of the extrema times mask_up = rmask (land values out of Tmax) / mask_dn = 1 on water, 1e20 on
// land (out of Tmin) of its column (mpdata_adiff.F:826-835); the limited transports times umask / vmask / rmask
// (:991, :1006, :1022) and the new tracer times rmask (step3d_t.F:1586-1596).
struct MpBeta { double up, dn; };

// what beta(k) of a column needs from memory beyond what the column carries from the levels below: level k+1 of Ta
// and t3, the four horizontal neighbours of Ta and t3 at level k, the anti-diffusive velocities of the cell's faces
struct MpBRaw { double Tup, t3up, Tw, Te, Ts, Tn, t3w, t3e, t3s, t3n, ua0, ua1, va0, va1, wa; };
struct MpBSlide { double T0, Tdn, t30, t3dn, wadn; };     // Ta(k), Ta(k-1), t3(k), t3(k-1), Wa(k-1)

__device__ __forceinline__ MpBRaw mp_load_braw(gcd_t Ta, gcd_t t3, gcd_t Ua, gcd_t Va, gcd_t Wa, long a2, int k, int N, long ni,
                                               long nij)
{
  MpBRaw r;
  const long a = a2 + (long)(k <= N ? k - 1 : N - 1) * nij;      // level k (k = N+1: nothing new is needed, values unused)
  const long au = k < N ? a + nij : a;
  r.Tup = Ta[au]; r.t3up = t3[au];
  r.Tw = Ta[a - 1]; r.Te = Ta[a + 1]; r.Ts = Ta[a - ni]; r.Tn = Ta[a + ni];
  r.t3w = t3[a - 1]; r.t3e = t3[a + 1]; r.t3s = t3[a - ni]; r.t3n = t3[a + ni];
  r.ua0 = Ua[a]; r.ua1 = Ua[a + 1]; r.va0 = Va[a]; r.va1 = Va[a + ni];
  r.wa = Wa[a + nij];                                              // Wa(i,j,k)
  return r;
}

// beta_up, beta_dn of level k of a column (mpdata_adiff.F:842-990), then slide the column to level k+1.
// mu0, muq[]: mask_up = rmask of the column and of its W, E, S, N neighbours (MASK only); mask_dn follows from it
template <bool MASK>
__device__ __forceinline__ MpBeta mp_beta(const MpBRaw &r, MpBSlide &s, int k, int N, double mu0, const double muq[4])
{
  const double T0 = s.T0, Tw = r.Tw, Te = r.Te, Ts = r.Ts, Tn = r.Tn;
  double Tmax, Tmin;
  if constexpr (!MASK) {
    Tmax = fmax(fmax(fmax(fmax(fmax(fmax(fmax(fmax(fmax(Tw, r.t3w), T0), s.t30), Te), r.t3e), Ts), r.t3s), Tn), r.t3n);
    Tmin = fmin(fmin(fmin(fmin(fmin(fmin(fmin(fmin(fmin(Tw, r.t3w), T0), s.t30), Te), r.t3e), Ts), r.t3s), Tn), r.t3n);
    if (k > 1) {
      Tmax = fmax(fmax(Tmax, s.Tdn), s.t3dn);
      Tmin = fmin(fmin(Tmin, s.Tdn), s.t3dn);
    }
    if (k < N) {
      Tmax = fmax(fmax(Tmax, r.Tup), r.t3up);
      Tmin = fmin(fmin(Tmin, r.Tup), r.t3up);
    }
  } else {
    const double Large = 1.0E+20;
    auto mdn = [&](double rr) { return fmax(1.0, fmin(Large, (1.0 - rr) * Large)); };
    const double md0 = mdn(mu0);
    Tmax = T0 * mu0;
    Tmin = T0 * md0;
    auto take = [&](double mu, double md, double x) { Tmax = fmax(Tmax, x * mu); Tmin = fmin(Tmin, x * md); };
    take(mu0, md0, s.t30);
    const double Tq[4] = {Tw, Te, Ts, Tn}, t3q[4] = {r.t3w, r.t3e, r.t3s, r.t3n};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double md = mdn(muq[q]);
      take(muq[q], md, Tq[q]);
      take(muq[q], md, t3q[q]);
    }
    if (k > 1) { take(mu0, md0, s.Tdn); take(mu0, md0, s.t3dn); }
    if (k < N) { take(mu0, md0, r.Tup); take(mu0, md0, r.t3up); }
  }
  double cff1 = Tw * fmax(0.0, r.ua0) - Te * fmin(0.0, r.ua1) + Ts * fmax(0.0, r.va0) - Tn * fmin(0.0, r.va1);
  double cff2 = T0 * fmax(0.0, r.ua1) - T0 * fmin(0.0, r.ua0) + T0 * fmax(0.0, r.va1) - T0 * fmin(0.0, r.va0);
  if (k == 1) {
    cff1 = cff1 - r.Tup * fmin(0.0, r.wa);
    cff2 = cff2 + T0 * fmax(0.0, r.wa);
  } else if (k < N) {
    cff1 = cff1 + s.Tdn * fmax(0.0, s.wadn) - r.Tup * fmin(0.0, r.wa);
    cff2 = cff2 + T0 * fmax(0.0, r.wa) - T0 * fmin(0.0, s.wadn);
  } else {
    cff1 = cff1 + s.Tdn * fmax(0.0, s.wadn);
    cff2 = cff2 - T0 * fmin(0.0, s.wadn);
  }
  MpBeta o;
  o.up = (Tmax - T0) / (cff1 + EPS_MP);
  o.dn = (T0 - Tmin) / (cff2 + EPS_MP);
  s.Tdn = s.T0; s.T0 = r.Tup; s.t3dn = s.t30; s.t30 = r.t3up; s.wadn = r.wa;
  return o;
}

// LDS slots of the update kernel: the level's Ta, t3, Hz, Ua, Va of every cell of the 66 x (TY+2) tile (two
// planes, alternating) and its beta_up / beta_dn (one plane)
enum { S_T = 0, S_T3, S_HZ, S_UA, S_VA, S_N };

template <int TY, bool MASK>
__global__ void __launch_bounds__(BLK_X *TY)
k_mp_update(const RomsDev *__restrict__ c, MpArgs m)
{
  DEV_PROLOGUE(c)
  constexpr int PY = TY + 2, PC = MPX * PY;
  __shared__ double lr[2][S_N][PC];
  __shared__ double lb[2][PC];              // beta_up / beta_dn
  const Blk XB = xcd_block();
  const int i0 = b.Istr + XB.x * BLK_X, j0 = b.Jstr + XB.y * TY;
  const int i = i0 + threadIdx.x, j = j0 + threadIdx.y;
  const int tid = threadIdx.y * BLK_X + threadIdx.x;
  const bool own = i <= b.Iend && j <= b.Jend;
  const double dt = c->p.dt;
  const gcd_t Ta = (gcd_t)m.Ta, Ua = (gcd_t)m.Ua, Va = (gcd_t)m.Va, Wa = (gcd_t)m.Wa;
  const gcd_t Hz = (gcd_t)c->F.Hz, z_r = (gcd_t)c->F.z_r;
  const gcd_t t3 = (gcd_t)(c->F.t + (2L + 3L * (m.itrc - 1)) * n3r);
  const int ltrc = m.itrc < b.NAT ? m.itrc : b.NAT;
  const gcd_t Akt = (gcd_t)(c->F.Akt + (long)(ltrc - 1) * n3w);
  const gd_t tn = (gd_t)(c->F.t + ((long)(m.nnew - 1) + 3L * (m.itrc - 1)) * n3r);
  const int so = (threadIdx.y + 1) * MPX + threadIdx.x + 1;
  // ring cell of this thread: rows 0 and TY+1, then columns 0 and 65 of rows 1..TY; od = the side on which its
  // neighbour lies outside the tile (0 W, 1 E, 2 S, 3 N).  The four corners are nobody's W/E/S/N neighbour.
  // beta exists on IstrU-1:Iendp1 x JstrV-1:Jendp1 (mpdata_adiff.F:842); cells outside it carry zeros, their faces
  // take the wall rule below.
  int sh = -1, ih = 0, jh = 0, od = 0;
  bool corner = false;
  if (tid < 2 * MPX) {
    const int r = tid / MPX, x = tid - r * MPX;
    sh = (r ? (PY - 1) * MPX : 0) + x; ih = i0 - 1 + x; jh = j0 - 1 + (r ? PY - 1 : 0);
    od = r ? 3 : 2; corner = x == 0 || x == MPX - 1;
  } else if (tid < 2 * MPX + 2 * TY) {
    const int q = tid - 2 * MPX, y = 1 + (q >> 1), x = (q & 1) ? MPX - 1 : 0;
    sh = y * MPX + x; ih = i0 - 1 + x; jh = j0 - 1 + y;
    od = (q & 1) ? 1 : 0;
  }
  const bool ring = sh >= 0;
  const bool inarr = i <= b.UBi && j <= b.UBj;                                   // own slot inside the arrays
  const bool cellb = i <= b.Iendp1 && j <= b.Jendp1;                             // ... and a cell with a beta
  const bool harr = ring && ih >= b.LBi && ih <= b.UBi && jh >= b.LBj && jh <= b.UBj;
  const bool halo = harr && !corner && ih >= b.IstrU - 1 && ih <= b.Iendp1 && jh >= b.JstrV - 1 && jh <= b.Jendp1;
  const long a2 = inarr ? I2(i, j) : I2(b.Istr, b.Jstr);
  const long a2h = harr ? I2(ih, jh) : a2;
  // ring cell: its neighbour outside the tile, and the faces Ua(i+1) / Va(j+1) (memory; 0 offset where there is no beta)
  const long oq[4] = {-1, 1, -(long)ni, (long)ni};
  const long aout = halo ? oq[od] : 0, au1 = halo ? 1 : 0, av1 = halo ? ni : 0;
  const double cffa = 1.0 / dt;                                   // mpdata_adiff.F:254
  const long a2c = cellb ? a2 : I2(b.Istr, b.Jstr);               // 2-D constants of the own cell
  const double cpp = dt * GF(pm)[a2c] * GF(pn)[a2c];
  const double omu0 = GF(om_u)[a2c], omu1 = GF(om_u)[a2c + 1], onv0 = GF(on_v)[a2c], onv1 = GF(on_v)[a2c + ni];
  const double onu0 = GF(on_u)[a2c], onu1 = GF(on_u)[a2c + 1], omv0 = GF(om_v)[a2c], omv1 = GF(om_v)[a2c + ni];
  const double omn = GF(omn)[a2c];
  // physical edges of the limited transports, mpdata_adiff.F:1031-1100: zero (closed) or the limited transport of the
  // next face inside -- the other face of this cell
  const bool v0_wall = b.south_edge && !b.NSperiodic && j == b.Jstr;       // Va(i,Jstr)
  const bool v1_wall = b.north_edge && !b.NSperiodic && j == b.Jend;       // Va(i,Jend+1)
  const bool u0_wall = b.west_edge && !b.EWperiodic && i == b.Istr;        // Ua(Istr,j)
  const bool u1_wall = b.east_edge && !b.EWperiodic && i == b.Iend;        // Ua(Iend+1,j)
  double um0 = 1.0, um1 = 1.0, vm0 = 1.0, vm1 = 1.0, rm0 = 1.0, rw0 = 1.0;
  double muq[4] = {1.0, 1.0, 1.0, 1.0}, muh[4] = {1.0, 1.0, 1.0, 1.0}, mu0h = 1.0;
  if constexpr (MASK) {
    // the masks of the limited transports (:991, :1006, :1022), times the wet/dry masks under WET_DRY (:993, :1008, :1024);
    // rm0 alone for the extrema and for the new tracer
    um0 = umaskw(c, a2c); um1 = umaskw(c, a2c + 1); vm0 = vmaskw(c, a2c); vm1 = vmaskw(c, a2c + ni); rm0 = GF(rmask)[a2c];
    rw0 = rmaskw(c, a2c);
    const gcd_t rmk = (gcd_t)c->F.rmask;
    const long a2hc = halo ? a2h : a2c;
    mu0h = rmk[a2hc];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      muq[q] = rmk[a2c + oq[q]];
      muh[q] = rmk[a2hc + oq[q]];
    }
  }
  const double cfl = -dt * c->p.lambda;
  // The eliminated right-hand side DC(k) is parked in t(nnew) itself and CF(k) in a scratch array (the round-2
  // beta_up storage): a rolled level loop with a few dozen live registers instead of two N-long register arrays
  // under full unrolling.  The back substitution reads them again in reverse order, a workgroup's own 2 x 30 x 4 KB
  // straight after writing them.
  const gd_t CFg = (gd_t)m.bup;
  // per level from memory -- own column: the level above of Ta and t3, the cell's own faces, Hz, z_r, Akt;
  // ring column: the same without z_r / Akt, plus the neighbour outside the tile and the far faces
  struct ORaw { double Tup, t3up, ua0, va0, wa, hz, zr, akt; };
  struct RRaw { double Tup, t3up, ua0, ua1, va0, va1, wa, hz, Tout, t3out; };
  auto load_own = [&](int k) {
    ORaw r;
    const long a = a2 + (long)(k <= N ? k - 1 : N - 1) * nij;      // k = N+1: nothing new is needed
    const long au = k < N ? a + nij : a;
    r.Tup = Ta[au]; r.t3up = t3[au];
    r.ua0 = Ua[a]; r.va0 = Va[a]; r.wa = Wa[a + nij];              // Wa(i,j,k)
    r.hz = Hz[a]; r.zr = z_r[a]; r.akt = Akt[a];                   // Akt(i,j,k-1)
    return r;
  };
  auto load_ring = [&](int k) {
    RRaw r;
    const long a = a2h + (long)(k <= N ? k - 1 : N - 1) * nij;
    const long au = k < N ? a + nij : a;
    r.Tup = Ta[au]; r.t3up = t3[au];
    r.ua0 = Ua[a]; r.ua1 = Ua[a + au1]; r.va0 = Va[a]; r.va1 = Va[a + av1]; r.wa = Wa[a + nij];
    r.hz = Hz[a];
    r.Tout = Ta[a + aout]; r.t3out = t3[a + aout];
    return r;
  };
  // carried from level to level
  MpBSlide so_s{Ta[a2], 0.0, t3[a2], 0.0, 0.0}, sh_s{Ta[a2h], 0.0, t3[a2h], 0.0, 0.0};
  double bup_p = 0.0, bdn_p = 0.0;   // beta(k-1)
  double T_p = 0.0;                  // Ta(k-1)
  double tvh_p = 0.0;                // Ta*Hz - horizontal divergence of level k-1
  double hz_p = 0.0, zr_p = 0.0;     // Hz(k-1), z_r(k-1)
  double FCadv_pp = 0.0;             // limited advective flux through the face below level k-1
  double FCd_pp = 0.0;               // diffusion coefficient FC(k-2) of the tridiagonal
  double CF_pp = 0.0, DC_pp = 0.0;   // CF(k-2), DC(k-2) after elimination
  // loads of level 1; from then on the loads of level k+1 are in flight while level k is evaluated
  ORaw po = load_own(1);
  RRaw pr;
  if (ring) pr = load_ring(1);
#pragma unroll 1
  for (int k = 1; k <= N + 1; k++) {
    const long a = a2 + (long)(k - 1) * nij;
    double bup = 0.0, bdn = 0.0, T0 = 0.0, tvh = 0.0, hz = 0.0, zr = po.zr, FCadv_p = 0.0;
    const double akt = po.akt;
    if (k <= N) {
      const ORaw co = po;
      const int pl = k & 1;
      T0 = so_s.T0;
      hz = co.hz;
      const double wadn = so_s.wadn;
      // ---- publish level k of the own and the ring cell
      lr[pl][S_T][so] = T0; lr[pl][S_T3][so] = so_s.t30; lr[pl][S_HZ][so] = hz; lr[pl][S_UA][so] = co.ua0; lr[pl][S_VA][so] = co.va0;
      RRaw cr;
      if (ring) {
        cr = pr;
        lr[pl][S_T][sh] = sh_s.T0; lr[pl][S_T3][sh] = sh_s.t30; lr[pl][S_HZ][sh] = cr.hz; lr[pl][S_UA][sh] = cr.ua0; lr[pl][S_VA][sh] = cr.va0;
      }
      // the loads of level k+1 go out before the barrier
      po = load_own(k + 1);
      if (ring) pr = load_ring(k + 1);
      __syncthreads();
      // ---- beta(k): own cell from the tile in LDS, ring cell with its outer neighbour from memory
      MpBRaw rb;
      rb.Tup = co.Tup; rb.t3up = co.t3up; rb.wa = co.wa;
      rb.Tw = lr[pl][S_T][so - 1]; rb.Te = lr[pl][S_T][so + 1]; rb.Ts = lr[pl][S_T][so - MPX]; rb.Tn = lr[pl][S_T][so + MPX];
      rb.t3w = lr[pl][S_T3][so - 1]; rb.t3e = lr[pl][S_T3][so + 1]; rb.t3s = lr[pl][S_T3][so - MPX]; rb.t3n = lr[pl][S_T3][so + MPX];
      rb.ua0 = co.ua0; rb.ua1 = lr[pl][S_UA][so + 1]; rb.va0 = co.va0; rb.va1 = lr[pl][S_VA][so + MPX];
      const double hzW = lr[pl][S_HZ][so - 1], hzE = lr[pl][S_HZ][so + 1], hzS = lr[pl][S_HZ][so - MPX], hzN = lr[pl][S_HZ][so + MPX];
      MpBeta bo = mp_beta<MASK>(rb, so_s, k, N, rm0, muq);
      if (!cellb) bo.up = bo.dn = 0.0;
      bup = bo.up; bdn = bo.dn;
      lb[0][so] = bup; lb[1][so] = bdn;
      if (ring) {
        MpBRaw rh;
        const int sW = od == 0 ? sh : sh - 1, sE = od == 1 ? sh : sh + 1, sS = od == 2 ? sh : sh - MPX, sN = od == 3 ? sh : sh + MPX;
        rh.Tup = cr.Tup; rh.t3up = cr.t3up; rh.wa = cr.wa;
        rh.Tw = lr[pl][S_T][sW]; rh.Te = lr[pl][S_T][sE]; rh.Ts = lr[pl][S_T][sS]; rh.Tn = lr[pl][S_T][sN];
        rh.t3w = lr[pl][S_T3][sW]; rh.t3e = lr[pl][S_T3][sE]; rh.t3s = lr[pl][S_T3][sS]; rh.t3n = lr[pl][S_T3][sN];
        if (od == 0) { rh.Tw = cr.Tout; rh.t3w = cr.t3out; }
        else if (od == 1) { rh.Te = cr.Tout; rh.t3e = cr.t3out; }
        else if (od == 2) { rh.Ts = cr.Tout; rh.t3s = cr.t3out; }
        else { rh.Tn = cr.Tout; rh.t3n = cr.t3out; }
        rh.ua0 = cr.ua0; rh.ua1 = cr.ua1; rh.va0 = cr.va0; rh.va1 = cr.va1;
        const MpBeta bh = mp_beta<MASK>(rh, sh_s, k, N, mu0h, muh);
        lb[0][sh] = halo ? bh.up : 0.0; lb[1][sh] = halo ? bh.dn : 0.0;
      }
      __syncthreads();
      {
        // ---- limited horizontal transports of level k (mpdata_adiff.F:1034-1049) and corrected fluxes
        const double bupW = lb[0][so - 1], bdnW = lb[1][so - 1], bupE = lb[0][so + 1], bdnE = lb[1][so + 1];
        const double bupS = lb[0][so - MPX], bdnS = lb[1][so - MPX], bupN = lb[0][so + MPX], bdnN = lb[1][so + MPX];
        const double ua0 = rb.ua0, ua1 = rb.ua1, va0 = rb.va0, va1 = rb.va1;
        double u0 = (fmin(fmin(bdnW, bup), 1.0) * fmax(0.0, ua0) + fmin(fmin(bupW, bdn), 1.0) * fmin(0.0, ua0)) * cffa * omu0;
        double u1 = (fmin(fmin(bdn, bupE), 1.0) * fmax(0.0, ua1) + fmin(fmin(bup, bdnE), 1.0) * fmin(0.0, ua1)) * cffa * omu1;
        double v0 = (fmin(fmin(bdnS, bup), 1.0) * fmax(0.0, va0) + fmin(fmin(bupS, bdn), 1.0) * fmin(0.0, va0)) * cffa * onv0;
        double v1 = (fmin(fmin(bdn, bupN), 1.0) * fmax(0.0, va1) + fmin(fmin(bup, bdnN), 1.0) * fmin(0.0, va1)) * cffa * onv1;
        if constexpr (MASK) { u0 = u0 * um0; u1 = u1 * um1; v0 = v0 * vm0; v1 = v1 * vm1; }
        {
          const double u0i = u0, u1i = u1, v0i = v0, v1i = v1;
          if (u0_wall) u0 = m.closed[LBS_WEST] ? 0.0 : u1i;
          if (u1_wall) u1 = m.closed[LBS_EAST] ? 0.0 : u0i;
          if (v0_wall) v0 = m.closed[LBS_SOUTH] ? 0.0 : v1i;
          if (v1_wall) v1 = m.closed[LBS_NORTH] ? 0.0 : v0i;
        }
        // corrected horizontal fluxes, step3d_t.F:1238-1255
        const double FXi = (fmax(u0, 0.0) * rb.Tw + fmin(u0, 0.0) * T0) * 0.5 * (hz + hzW) * onu0;
        const double FXip1 = (fmax(u1, 0.0) * T0 + fmin(u1, 0.0) * rb.Te) * 0.5 * (hzE + hz) * onu1;
        const double FEj = (fmax(v0, 0.0) * rb.Ts + fmin(v0, 0.0) * T0) * 0.5 * (hz + hzS) * omv0;
        const double FEjp1 = (fmax(v1, 0.0) * T0 + fmin(v1, 0.0) * rb.Tn) * 0.5 * (hzN + hz) * omv1;
        const double cff1 = cpp * (FXip1 - FXi);
        const double cff2 = cpp * (FEjp1 - FEj);
        const double cff3 = cff1 + cff2;
        tvh = T0 * hz - cff3;                                     // :1265
        // limited vertical transport through the face between k-1 and k (:1051-1060), corrected flux :1281-1290
        if (k > 1) {
          const double c1 = fmin(fmin(bdn_p, bup), 1.0);
          const double c2 = fmin(fmin(bup_p, bdn), 1.0);
          double w = (c1 * fmax(0.0, wadn) + c2 * fmin(0.0, wadn)) * cffa * omn * (zr - zr_p);
          if constexpr (MASK) w = w * rw0;
          FCadv_p = fmax(w, 0.0) * T_p + fmin(w, 0.0) * T0;
        }
      }
    }
    // ---- level kk = k-1 is complete: vertical advection (:1305), forward elimination of the classic
    //      tridiagonal (step3d_t.F:1431-1501)
    if (own && k > 1) {
      const int kk = k - 1;
      const double tv = tvh_p - cpp * (FCadv_p - FCadv_pp);
      double FCd = 0.0;
      if (kk < N) {
        const double cff1 = 1.0 / (zr - zr_p);
        FCd = cfl * cff1 * akt;                                   // Akt(i,j,kk)
      }
      const double BC = hz_p - FCd - FCd_pp;
      double CFk = 0.0, DCk;
      if (kk == 1) {
        const double cff = 1.0 / BC;
        CFk = cff * FCd;
        DCk = cff * tv;
      } else if (kk < N) {
        const double cff = 1.0 / (BC - FCd_pp * CF_pp);
        CFk = cff * FCd;
        DCk = cff * (tv - FCd_pp * DC_pp);
      } else {
        DCk = (tv - FCd_pp * DC_pp) / (BC - FCd_pp * CF_pp);
      }
      tn[a - nij] = DCk;
      CFg[a - nij] = CFk;
      CF_pp = CFk;
      DC_pp = DCk;
      FCd_pp = FCd;
      FCadv_pp = FCadv_p;
    }
    bup_p = bup; bdn_p = bdn; T_p = T0; tvh_p = tvh; hz_p = hz; zr_p = zr;
  }
  if (!own) return;
  // back substitution; DC_pp = DC(N) is the top value.  A lane reads back only what it stored itself.
  double up = DC_pp;
  {
    double v = up;
    if constexpr (MASK) v = v * rm0;
    tn[a2 + (long)(N - 1) * nij] = v;
  }
#pragma unroll 8
  for (int k = N - 1; k >= 1; k--) {       // the loads do not depend on `up`: eight levels in flight
    const long a = a2 + (long)(k - 1) * nij;
    double v = tn[a] - CFg[a] * up;
    up = v;
    if constexpr (MASK) v = v * rm0;
    tn[a] = v;
  }
}

}  // namespace

// The MPDATA tracers itrc0 .. itrc0+n-1 of step3d_t; called by roms_hip_step3d_t (k_step3d_t.hip).  In batches of up
// to three: one exchange of their t(nnew), ONE upstream step for the batch (k_mp_ta; Ta -> ws3[1], [8], [9]), then per
// tracer the anti-diffusive velocities and the limited, corrected step.
int roms_launch_step3d_t_mpdata(int nnew, int itrc0, int n)
{
  const roms_bounds_t &b = g_ctx.b;
  if (b.NghostPoints != 3) return roms_fail("roms_hip_step3d_t", "MPDATA needs NghostPoints = 3 (inp_par.F:266-278)");
  int rc;
  const long n3r = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * b.N;
  MpArgs m{};
  m.Ua = g_ctx.hostc.ws3[2]; m.Va = g_ctx.hostc.ws3[3]; m.Wa = g_ctx.hostc.ws3[4];
  m.bup = g_ctx.hostc.ws3[5]; m.bdn = g_ctx.hostc.ws3[6];
  m.oHz = g_ctx.hostc.ws3[0]; m.odz = g_ctx.hostc.ws3[7];
  m.TaB[0] = g_ctx.hostc.ws3[1]; m.TaB[1] = g_ctx.hostc.ws3[8]; m.TaB[2] = g_ctx.hostc.ws3[9];
  m.nnew = nnew;
  m.closed[LBS_WEST] = lbc_code(g_ctx.p, LBS_WEST, LBV_U) == LBC_CLOSED;
  m.closed[LBS_EAST] = lbc_code(g_ctx.p, LBS_EAST, LBV_U) == LBC_CLOSED;
  m.closed[LBS_SOUTH] = lbc_code(g_ctx.p, LBS_SOUTH, LBV_V) == LBC_CLOSED;
  m.closed[LBS_NORTH] = lbc_code(g_ctx.p, LBS_NORTH, LBV_V) == LBC_CLOSED;
  hipLaunchKernelGGL(k_mp_metrics, grid2d(b.UBi - b.LBi + 1, b.UBj - b.LBj + 1), block2d(), 0, g_ctx.stream, g_ctx.devc,
                     g_ctx.hostc.ws3[0], g_ctx.hostc.ws3[7]);
  KERNEL_CHECK("k_mp_metrics");
#ifndef MP_UTY
#define MP_UTY 8
#endif
  constexpr int UTY = MP_UTY;            // rows per workgroup of the fused limiter + update kernel
  const bool mk = g_ctx.p.masking != 0;
  for (int q0 = 0; q0 < n; q0 += 3) {
    const int nb = n - q0 < 3 ? n - q0 : 3;
    // three-point footprint: refresh the ghost points of t(nnew) first, step3d_t.F:369-386
    halo_batch_begin();
    for (int q = 0; q < nb; q++)
      halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_t] + ((long)(nnew - 1) + 3L * (itrc0 + q0 + q - 1)) * n3r);
    if ((rc = halo_batch_end())) return rc;
    m.itrc = itrc0 + q0;
    m.nb = nb;
    m.Ta = m.TaB[0];
    {
      dim3 g3 = grid_tile_level(b.Iendp2i - b.IstrUm2 + 1, b.Jendp2i - b.JstrVm2 + 1, b.N);
      hipLaunchKernelGGL(k_mp_ta, g3, block2d(), 0, g_ctx.stream, g_ctx.devc, m);
    }
    KERNEL_CHECK("k_mp_ta");
    for (int q = 0; q < nb; q++) {
      m.itrc = itrc0 + q0 + q;
      m.Ta = m.TaB[q];
      {
        const dim3 g3((unsigned)((b.Iendp2 - (b.IstrU - 1) + 1 + BLK_X - 1) / BLK_X),
                      (unsigned)((b.Jendp2 - (b.JstrV - 1) + 1 + MP_ATY - 1) / MP_ATY), 1);
        void (*kern)(const RomsDev *, MpArgs) = mk ? (g_ctx.p.mpdata_fast ? k_mp_adiff<true, true> : k_mp_adiff<false, true>)
                                                  : (g_ctx.p.mpdata_fast ? k_mp_adiff<true, false> : k_mp_adiff<false, false>);
        hipLaunchKernelGGL(kern, g3, dim3(BLK_X, MP_ATY, 1), 0, g_ctx.stream, g_ctx.devc, m);
      }
      KERNEL_CHECK("k_mp_adiff");
      const dim3 g((unsigned)((b.Iend - b.Istr + 1 + BLK_X - 1) / BLK_X), (unsigned)((b.Jend - b.Jstr + 1 + UTY - 1) / UTY), 1);
      void (*upd)(const RomsDev *, MpArgs) = mk ? k_mp_update<UTY, true> : k_mp_update<UTY, false>;
      hipLaunchKernelGGL(upd, g, dim3(BLK_X, UTY, 1), 0, g_ctx.stream, g_ctx.devc, m);
      KERNEL_CHECK("k_mp_update");
    }
  }
  return 0;
}
