// k_gls.hip -- the generic length-scale turbulence closure (GLS_MIXING) of the reference's step:
//
//   roms_hip_gls_prestep   gls_prestep_tile, ROMS/Nonlinear/gls_prestep.F:66-420 (main3d.F:567): half-step predictor of
//                          tke and gls (fourth-order centred advection, LF-AM3 weights) and the Hz-weighted start
//                          values of the corrector
//   roms_hip_gls_corstep   gls_corstep_tile, ROMS/Nonlinear/gls_corstep.F:101-1218 (main3d.F:793): shear (RI_SPLINES or
//                          plain) and stratification, their horizontal smoothing (N2S2_HORAVG), third-order upstream
//                          advection of tke / gls, production and dissipation, the two implicit vertical diffusion
//                          solves, length-scale limitation, the stability functions (Galperin, Kantha-Clayson,
//                          Canuto A / B) and Akv, Akt, Akk, Akp, Lscale
//                          With roms_params_t.gls_mixing = 2 (MY25_MIXING) the same two entries are my25_prestep.F (the same
//                          text as gls_prestep.F) and my25_corstep.F:96-894: no floors in the advection, the level-2.5
//                          production / dissipation / wall-proximity terms and the Galperin stability functions
//   tkebc_tile             ROMS/Nonlinear/tkebc_im.F:50-698, closed and gradient edges and the corners
//
// One thread per water column (64 x 4 columns per workgroup, i fastest), levels in a loop: the closure is a per-column
// algorithm (two tridiagonal solves) with a five-point horizontal advection stencil; the Thomas coefficients and the
// per-level shear / stratification go through four 3-D scratch arrays.  Every expression keeps the reference's order
// of operations (no FMA contraction); the real powers call the device pow().  Not one of the five BASELINE
// configurations: written for parity first.  Pinned: the oracle against the reference's GLS builds bit for bit, this
// file against the oracle (tests/test_gpu_gls.py).
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

struct GlsConst {            // initialize_scalars, mod_scalars.F:1686-1712, :1756-1768, :4450-4490
  double Gh0, Ghcri, Ghmin, E2;
  double s0, s1, s2, s4, s5, s6, b0, b1, b2, b3, b4, b5;
  double my_Sh1, my_Sh2, my_Sm2, my_Sm3, my_Sm4, my_B1pm1o3, my_B1p2o3;
};

GlsConst gls_constants(int stab)
{
  GlsConst c;
  memset(&c, 0, sizeof c);
  c.Ghmin = -0.28;
  c.E2 = 1.33;
  if (stab == GLS_CANUTO_A || stab == GLS_CANUTO_B) {
    double L1, L2, L3, L4, L5, L6, L7, L8;
    if (stab == GLS_CANUTO_A) { c.Gh0 = 0.0329; c.Ghcri = 0.03; L1 = 0.107; L2 = 0.0032; L3 = 0.0864; L4 = 0.12; L5 = 11.9; L6 = 0.4; L7 = 0.0; L8 = 0.48; }
    else { c.Gh0 = 0.0444; c.Ghcri = 0.0414; L1 = 0.127; L2 = 0.00336; L3 = 0.0906; L4 = 0.101; L5 = 11.2; L6 = 0.4; L7 = 0.0; L8 = 0.318; }
    c.s0 = 3.0 / 2.0 * L1 * (L5 * L5);
    c.s1 = -L4 * (L6 + L7) + 2.0 * L4 * L5 * (L1 - 1.0 / 3.0 * L2 - L3) + 3.0 / 2.0 * L1 * L5 * L8;
    c.s2 = -3.0 / 8.0 * L1 * ((L6 * L6) - (L7 * L7));
    c.s4 = 2.0 * L5;
    c.s5 = 2.0 * L4;
    c.s6 = 2.0 / 3.0 * L5 * (3.0 * (L3 * L3) - (L2 * L2)) - 1.0 / 2.0 * L5 * L1 * (3.0 * L3 - L2) + 3.0 / 4.0 * L1 * (L6 - L7);
    c.b0 = 3.0 * (L5 * L5);
    c.b1 = L5 * (7.0 * L4 + 3.0 * L8);
    c.b2 = (L5 * L5) * (3.0 * (L3 * L3) - (L2 * L2)) - 3.0 / 4.0 * ((L6 * L6) - (L7 * L7));
    c.b3 = L4 * (4.0 * L4 + 3.0 * L8);
    c.b5 = 1.0 / 4.0 * ((L2 * L2) - 3.0 * (L3 * L3)) * ((L6 * L6) - (L7 * L7));
    c.b4 = L4 * (L2 * L6 - 3.0 * L3 * L7 - L5 * ((L2 * L2) - (L3 * L3))) + L5 * L8 * (3.0 * (L3 * L3) - (L2 * L2));
  } else {
    c.Gh0 = 0.028;
    c.Ghcri = 0.02;
  }
  const double A1 = 0.92, A2 = 0.74, B1 = 16.6, B2 = 10.1, C1 = 0.08, C2 = 0.7, C3 = 0.2;
  c.my_B1pm1o3 = 1.0 / pow(B1, 1.0 / 3.0);
  c.my_B1p2o3 = pow(B1, 2.0 / 3.0);
  c.my_Sm2 = 9.0 * A1 * A2;
  c.my_Sh1 = A2 * (1.0 - 6.0 * A1 / B1);
  if (stab == GLS_KANTHA_CLAYSON) {
    c.my_Sh2 = 3.0 * A2 * (6.0 * A1 + B2 * (1.0 - C3));
    c.my_Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2 * (1.0 - C2);
  } else {
    c.my_Sh2 = 3.0 * A2 * (6.0 * A1 + B2);
    c.my_Sm3 = A1 * (1.0 - 3.0 * C1 - 6.0 * A1 / B1);
    c.my_Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2;
  }
  return c;
}

// constants of gls_corstep.F:262-312, evaluated on the host (pow of the host's math library, as the oracle does)
struct GlsFac {
  double Zos_min, L_sft, ogls_sigp, sqrt2, cmu_fac1, cmu_fac2, cmu_fac3, gls_fac2, gls_fac3, gls_fac4, gls_fac5, gls_fac6;
  double gls_exp1, tke_exp1, tke_exp2, tke_exp4, cmu0p;      // cmu0p = gls_cmu0 ** gls_p
  int Lmy25;
};

GlsFac gls_factors(const roms_params_t &p)
{
  GlsFac f;
  const double vonKar = 0.41;
  f.Zos_min = p.Zos > 0.0001 ? p.Zos : 0.0001;
  f.Lmy25 = (p.gls_p == 0.0) && (p.gls_n == 1.0) && (p.gls_m == 1.0);
  f.L_sft = vonKar;
  f.ogls_sigp = 1.0 / p.gls_sigp;
  f.sqrt2 = sqrt(2.0);
  f.cmu_fac1 = pow(p.gls_cmu0, -p.gls_p / p.gls_n);
  f.cmu_fac2 = pow(p.gls_cmu0, 3.0 + p.gls_p / p.gls_n);
  f.cmu_fac3 = 1.0 / pow(p.gls_cmu0, 2.0);
  f.gls_fac2 = pow(p.gls_cmu0, p.gls_p) * p.gls_n * pow(vonKar, p.gls_n);
  f.gls_fac3 = pow(p.gls_cmu0, p.gls_p) * p.gls_n;
  f.gls_fac4 = pow(p.gls_cmu0, p.gls_p);
  f.gls_fac5 = pow(0.56, 0.5 * p.gls_n) * pow(p.gls_cmu0, p.gls_p);
  f.gls_fac6 = 8.0 / pow(p.gls_cmu0, 6.0);
  f.gls_exp1 = 1.0 / p.gls_n;
  f.tke_exp1 = p.gls_m / p.gls_n;
  f.tke_exp2 = 0.5 + p.gls_m / p.gls_n;
  f.tke_exp4 = p.gls_m + 0.5 * p.gls_n;
  f.cmu0p = pow(p.gls_cmu0, p.gls_p);
  return f;
}

// ------------------------------------------------------------------ tkebc --
// One thread per boundary point and level; phase 0 = the four edges, phase 1 = the corners (they read edge values).
__global__ void k_tke_bc(const RomsDev *__restrict__ c, int nout, int phase)
{
  DEV_PROLOGUE(c)
  const long lev = (long)(nout - 1) * n3w;
  double *tke = c->F.tke + lev, *gls = c->F.gls + lev;
  const bool mk = c->p.masking != 0;
  const double *rmask = c->F.rmask;
  const int k = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int Istr = b.Istr, Iend = b.Iend, Jstr = b.Jstr, Jend = b.Jend;
  const bool wE = b.west_edge && !b.EWperiodic, eE = b.east_edge && !b.EWperiodic;
  const bool sE = b.south_edge && !b.NSperiodic, nE = b.north_edge && !b.NSperiodic;
  auto copy = [&](int ig, int jg, int ii, int ji) {       // ghost (ig,jg) <- inside (ii,ji), times rmask(ghost)
    const long g = I3W(ig, jg, k), s = I3W(ii, ji, k);
    double a = tke[s], p2 = gls[s];
    if (mk) { a = a * rmask[I2(ig, jg)]; p2 = p2 * rmask[I2(ig, jg)]; }
    tke[g] = a; gls[g] = p2;
  };
  if (phase == 0) {
    const int nI = Iend - Istr + 1, nJ = Jend - Jstr + 1;
    if (q < nJ) {
      if (wE) copy(Istr - 1, Jstr + q, Istr, Jstr + q);
      if (eE) copy(Iend + 1, Jstr + q, Iend, Jstr + q);
    }
    if (q < nI) {
      if (sE) copy(Istr + q, Jstr - 1, Istr + q, Jstr);
      if (nE) copy(Istr + q, Jend + 1, Istr + q, Jend);
    }
  } else if (q == 0 && !(b.EWperiodic || b.NSperiodic)) {
    auto corner = [&](int ic, int jc, int ia, int ja, int ib, int jb) {
      tke[I3W(ic, jc, k)] = 0.5 * (tke[I3W(ia, ja, k)] + tke[I3W(ib, jb, k)]);
      gls[I3W(ic, jc, k)] = 0.5 * (gls[I3W(ia, ja, k)] + gls[I3W(ib, jb, k)]);
    };
    if (b.south_edge && b.west_edge) corner(Istr - 1, Jstr - 1, Istr, Jstr - 1, Istr - 1, Jstr);
    if (b.south_edge && b.east_edge) corner(Iend + 1, Jstr - 1, Iend, Jstr - 1, Iend + 1, Jstr);
    if (b.north_edge && b.west_edge) corner(Istr - 1, Jend + 1, Istr, Jend + 1, Istr - 1, Jend);
    if (b.north_edge && b.east_edge) corner(Iend + 1, Jend + 1, Iend, Jend + 1, Iend + 1, Jend);
  }
}

int tke_bc(int nout)
{
  const roms_bounds_t &b = g_ctx.b;
  const int nI = b.Iend - b.Istr + 1, nJ = b.Jend - b.Jstr + 1, n = nI > nJ ? nI : nJ;
  hipLaunchKernelGGL(k_tke_bc, dim3((n + 63) / 64, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, nout, 0);
  if (!(b.EWperiodic || b.NSperiodic))
    hipLaunchKernelGGL(k_tke_bc, dim3(1, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, nout, 1);
  KERNEL_CHECK("k_tke_bc");
  return 0;
}

// the conditions built: periodic, closed, gradient on the tracers' table (LBC(:,isMtke,ng) is not carried separately)
int gls_check(const char *where)
{
  if (!g_ctx.p.gls_mixing) return roms_fail(where, "gls_mixing is not set in roms_params_t");
  for (int sd = 0; sd < 4; sd++) {
    const int code = lbc_code(g_ctx.p, sd, LBV_T);
    if (code != LBC_PERIODIC && code != LBC_CLOSED && code != LBC_GRADIENT)
      return roms_fail(where, "tkebc: only periodic, closed and gradient edges are implemented for tke / gls");
  }
  return 0;
}

// gradient of field A at the u-face i (v-face j) of level offset `lk`, with the MASKING multiply and the reference's
// rule for the face outside a physical edge (the next face's value)
struct GradX {
  const roms_bounds_t &b; const double *um; bool mk; long ni;
  __device__ double operator()(const double *A, long base, int i, int j, int LBi, int LBj) const
  {
    int ii = i;
    if (!b.EWperiodic) {
      if (b.west_edge && ii == b.Istr - 1) ii = b.Istr;
      if (b.east_edge && ii == b.Iend + 2) ii = b.Iend + 1;
    }
    const long a = base + (long)(ii - LBi) + (long)(j - LBj) * ni;
    double g = (A[a] - A[a - 1]);
    if (mk) g = g * um[(long)(ii - LBi) + (long)(j - LBj) * ni];
    return g;
  }
};
struct GradY {
  const roms_bounds_t &b; const double *vm; bool mk; long ni;
  __device__ double operator()(const double *A, long base, int i, int j, int LBi, int LBj) const
  {
    int jj = j;
    if (!b.NSperiodic) {
      if (b.south_edge && jj == b.Jstr - 1) jj = b.Jstr;
      if (b.north_edge && jj == b.Jend + 2) jj = b.Jend + 1;
    }
    const long a = base + (long)(i - LBi) + (long)(jj - LBj) * ni;
    double g = (A[a] - A[a - ni]);
    if (mk) g = g * vm[(long)(i - LBi) + (long)(jj - LBj) * ni];
    return g;
  }
};

// ------------------------------------------------------------ gls_prestep --
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_gls_prestep(const RomsDev *__restrict__ c, roms_step_idx_t s)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const roms_params_t &p = c->p;
  const int nstp = s.nstp, nnew = s.nnew;
  const double dt = p.dt, Gamma = 1.0 / 6.0;
  const bool mk = p.masking != 0;
  const double *Huon = c->F.Huon, *Hvom = c->F.Hvom, *Hz = c->F.Hz, *Wv = c->F.W;
  double *tke = c->F.tke, *gls = c->F.gls;
  const long Ls = (long)(nstp - 1) * n3w, Ln = (long)(nnew - 1) * n3w, L3 = 2L * n3w;
  const GradX gx{b, c->F.umask, mk, ni};
  const GradY gy{b, c->F.vmask, mk, ni};
  const long a2 = I2(i, j);
  const double pmn = 0.0;
  (void)pmn;
  double cff1, cff2, cff3;
  int indx;
  if (s.iic == s.ntfirst) { cff1 = 1.0; cff2 = 0.0; cff3 = 0.5 * dt; indx = nstp; }
  else { cff1 = 0.5 + Gamma; cff2 = 0.5 - Gamma; cff3 = (1.0 - Gamma) * dt; indx = 3 - nstp; }
  const long Li = (long)(indx - 1) * n3w;
  const double c6 = 1.0 / 6.0;
  // vertical flux through the rho-level kk between W-levels kk-1 and kk (gls_prestep.F:322-352): CF and FC, FCL
  auto vflux = [&](int kk, double &CFk, double &FCk, double &FCLk) {
    const long w = a2 + (long)kk * nij;                   // W-level kk
    CFk = 0.5 * (Wv[w] + Wv[w - nij]);
    const double *T = tke + Ls, *G = gls + Ls;
    if (kk == 1) {
      FCk = CFk * (1.0 / 3.0 * T[w - nij] + 5.0 / 6.0 * T[w] - 1.0 / 6.0 * T[w + nij]);
      FCLk = CFk * (1.0 / 3.0 * G[w - nij] + 5.0 / 6.0 * G[w] - 1.0 / 6.0 * G[w + nij]);
    } else if (kk == N) {
      FCk = CFk * (1.0 / 3.0 * T[w] + 5.0 / 6.0 * T[w - nij] - 1.0 / 6.0 * T[w - 2 * nij]);
      FCLk = CFk * (1.0 / 3.0 * G[w] + 5.0 / 6.0 * G[w - nij] - 1.0 / 6.0 * G[w - 2 * nij]);
    } else {
      FCk = CFk * (7.0 / 12.0 * (T[w - nij] + T[w]) - 1.0 / 12.0 * (T[w - 2 * nij] + T[w + nij]));
      FCLk = CFk * (7.0 / 12.0 * (G[w - nij] + G[w]) - 1.0 / 12.0 * (G[w - 2 * nij] + G[w + nij]));
    }
  };
  double CFlo, FClo, FCLlo;
  vflux(1, CFlo, FClo, FCLlo);
  for (int k = 1; k <= N - 1; k++) {
    const long wk = (long)k * nij;                                  // plane of W-level k
    const long r = a2 + (long)(k - 1) * nij;                        // rho-level k
    const double *T = tke + Ls + wk, *G = gls + Ls + wk;
    // horizontal fluxes at the faces i, i+1, j, j+1 (fourth-order centred, :176-263)
    auto fx = [&](int ii, double &XF, double &FX, double &FXL) {
      const long q = (long)(ii - LBi) + (long)(j - LBj) * ni;
      XF = 0.5 * (Huon[q + (long)(k - 1) * nij] + Huon[q + (long)k * nij]);
      FX = XF * 0.5 * (T[q - 1] + T[q] - c6 * (gx(T, 0, ii + 1, j, LBi, LBj) - gx(T, 0, ii - 1, j, LBi, LBj)));
      FXL = XF * 0.5 * (G[q - 1] + G[q] - c6 * (gx(G, 0, ii + 1, j, LBi, LBj) - gx(G, 0, ii - 1, j, LBi, LBj)));
    };
    auto fe = [&](int jj, double &EF, double &FE, double &FEL) {
      const long q = (long)(i - LBi) + (long)(jj - LBj) * ni;
      EF = 0.5 * (Hvom[q + (long)(k - 1) * nij] + Hvom[q + (long)k * nij]);
      FE = EF * 0.5 * (T[q - ni] + T[q] - c6 * (gy(T, 0, i, jj + 1, LBi, LBj) - gy(T, 0, i, jj - 1, LBi, LBj)));
      FEL = EF * 0.5 * (G[q - ni] + G[q] - c6 * (gy(G, 0, i, jj + 1, LBi, LBj) - gy(G, 0, i, jj - 1, LBi, LBj)));
    };
    double XF0, FX0, FXL0, XF1, FX1, FXL1, EF0, FE0, FEL0, EF1, FE1, FEL1;
    fx(i, XF0, FX0, FXL0); fx(i + 1, XF1, FX1, FXL1);
    fe(j, EF0, FE0, FEL0); fe(j + 1, EF1, FE1, FEL1);
    const double cff = 0.5 * (Hz[r] + Hz[r + nij]);
    const double cff4 = cff3 * c->F.pm[a2] * c->F.pn[a2];
    double Hz_half = cff - cff4 * (XF1 - XF0 + EF1 - EF0);
    const double tk = tke[Ls + a2 + wk], gk = gls[Ls + a2 + wk];
    double t3 = cff * (cff1 * tk + cff2 * tke[Li + a2 + wk]) - cff4 * (FX1 - FX0 + FE1 - FE0);
    double g3 = cff * (cff1 * gk + cff2 * gls[Li + a2 + wk]) - cff4 * (FXL1 - FXL0 + FEL1 - FEL0);
    tke[Ln + a2 + wk] = cff * tk;
    gls[Ln + a2 + wk] = cff * gk;
    // vertical advection (:300-375)
    double CFhi, FChi, FCLhi;
    vflux(k + 1, CFhi, FChi, FCLhi);
    Hz_half = Hz_half - cff4 * (CFhi - CFlo);
    const double o = 1.0 / Hz_half;
    t3 = o * (t3 - cff4 * (FChi - FClo));
    g3 = o * (g3 - cff4 * (FCLhi - FCLlo));
    tke[L3 + a2 + wk] = t3;
    gls[L3 + a2 + wk] = g3;
    CFlo = CFhi; FClo = FChi; FCLlo = FCLhi;
  }
}

// ------------------------------------------------------- gls_corstep: shear --
// shear2 at W-levels 1..N-1 on Istrm1:Iendp1 x Jstrm1:Jendp1 (gls_corstep.F:316-372) -> S (W-type scratch array)
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_gls_shear(const RomsDev *__restrict__ c, int nstp, double *__restrict__ S, double *__restrict__ CFs, double *__restrict__ dUs,
            double *__restrict__ dVs)
{
  DEV_PROLOGUE(c)
  const int i = b.Istrm1 + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.Jstrm1 + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.Iendp1 || j > b.Jendp1) return;
  const double *Hz = c->F.Hz, *z_r = c->F.z_r;
  const double *u = c->F.u + (long)(nstp - 1) * n3r, *v = c->F.v + (long)(nstp - 1) * n3r;
  const long a2 = I2(i, j);
  if (c->p.gls_ri_splines) {
    double CFm = 0.0, dUm = 0.0, dVm = 0.0;
    for (int k = 1; k <= N - 1; k++) {
      const long r = a2 + (long)(k - 1) * nij;
      const double cff = 1.0 / (2.0 * Hz[r + nij] + Hz[r] * (2.0 - CFm));
      const double CFk = cff * Hz[r + nij];
      const double dUk = cff * (3.0 * (u[r + nij] - u[r] + u[r + nij + 1] - u[r + 1]) - Hz[r] * dUm);
      const double dVk = cff * (3.0 * (v[r + nij] - v[r] + v[r + nij + ni] - v[r + ni]) - Hz[r] * dVm);
      CFs[a2 + (long)k * nij] = CFk; dUs[a2 + (long)k * nij] = dUk; dVs[a2 + (long)k * nij] = dVk;
      CFm = CFk; dUm = dUk; dVm = dVk;
    }
    double dUp = 0.0, dVp = 0.0;
    for (int k = N - 1; k >= 1; k--) {
      const long w = a2 + (long)k * nij;
      const double cf = CFs[w];
      const double dUk = dUs[w] - cf * dUp, dVk = dVs[w] - cf * dVp;
      S[w] = dUk * dUk + dVk * dVk;
      dUp = dUk; dVp = dVk;
    }
  } else {
    for (int k = 1; k <= N - 1; k++) {
      const long r = a2 + (long)(k - 1) * nij;
      const double cff = 0.5 / (z_r[r + nij] - z_r[r]);
      const double a1 = cff * (u[r + nij] - u[r] + u[r + nij + 1] - u[r + 1]);
      const double a2v = cff * (v[r + nij] - v[r] + v[r + nij + ni] - v[r + ni]);
      S[a2 + (long)k * nij] = a1 * a1 + a2v * a2v;
    }
  }
}

struct GlsArgs {
  roms_step_idx_t s;
  GlsConst K;
  GlsFac f;
  double *S;                 // raw shear2 (k_gls_shear)
  double *SH, *BU;           // shear2 / buoy2 as the column uses them (after N2S2_HORAVG), W-type scratch
  double *FCK, *FCP, *BCK, *BCP, *CF;
};

// --------------------------------------------------------- gls_corstep: column --
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_gls_corstep(const RomsDev *__restrict__ c, GlsArgs A)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const roms_params_t &p = c->p;
  const GlsConst &K = A.K;
  const GlsFac &f = A.f;
  const int nstp = A.s.nstp, nnew = A.s.nnew;
  const double dt = p.dt;
  const double vonKar = 0.41, Gadv = 1.0 / 3.0, eps = 1.0E-10;
  const bool mk = p.masking != 0;
  const bool my25 = p.gls_mixing == 2;       // MY25_MIXING: my25_corstep.F, the same routine up to the vertical terms
  const double gls_p = p.gls_p, gls_m = p.gls_m, gls_n = p.gls_n, gls_cmu0 = p.gls_cmu0;
  const double gls_c1 = p.gls_c1, gls_c2 = p.gls_c2, gls_sigk = p.gls_sigk, gls_sigp = p.gls_sigp;
  const double gls_Kmin = p.gls_Kmin, gls_Pmin = p.gls_Pmin;
  const double Akv_bak = p.Akv_bak, Akk_bak = p.Akk_bak, Akp_bak = p.Akp_bak, AktT_bak = p.Akt_bak[0];
  (void)gls_p;
  const double *Huon = c->F.Huon, *Hvom = c->F.Hvom, *Hz = c->F.Hz, *Wv = c->F.W, *z_w = c->F.z_w, *bvf = c->F.bvf;
  double *tke = c->F.tke, *gls = c->F.gls, *Akv = c->F.Akv, *Akt = c->F.Akt, *Akk = c->F.Akk, *Akp = c->F.Akp, *Lscale = c->F.Lscale;
  const long Ls = (long)(nstp - 1) * n3w, Ln = (long)(nnew - 1) * n3w, L3 = 2L * n3w;
  const GradX gx{b, c->F.umask, mk, ni};
  const GradY gy{b, c->F.vmask, mk, ni};
  const long a2 = I2(i, j);
  const double cdt = dt * c->F.pm[a2] * c->F.pn[a2];
  double *Tn = tke + Ln, *Gn = gls + Ln;
  // ---- shear2 / buoy2 of the column, with N2S2_HORAVG (:384-440) evaluated on the fly: the reference copies shear2
  //      (not buoy2) across the tile's domain edges before averaging, which is an index clamp here
  {
    const int Istr = b.Istr, Iend = b.Iend, Jstr = b.Jstr, Jend = b.Jend;
    auto Sfix = [&](int ii, int jj, long wk) {
      if (b.west_edge && ii == Istr - 1) ii = Istr;
      if (b.east_edge && ii == Iend + 1) ii = Iend;
      if (b.south_edge && jj == Jstr - 1) jj = Jstr;
      if (b.north_edge && jj == Jend + 1) jj = Jend;
      return A.S[I2(ii, jj) + wk];
    };
    for (int k = 1; k <= N - 1; k++) {
      const long wk = (long)k * nij;
      double sh, bu;
      if (p.gls_n2s2_horavg) {
        auto avgS = [&](int ii, int jj) { return 0.25 * (Sfix(ii, jj, wk) + Sfix(ii + 1, jj, wk) + Sfix(ii, jj + 1, wk) + Sfix(ii + 1, jj + 1, wk)); };
        auto avgB = [&](int ii, int jj) {
          const long q = I2(ii, jj) + wk;
          return 0.25 * (bvf[q] + bvf[q + 1] + bvf[q + ni] + bvf[q + ni + 1]);
        };
        bu = 0.25 * (avgB(i, j) + avgB(i - 1, j) + avgB(i, j - 1) + avgB(i - 1, j - 1));
        sh = 0.25 * (avgS(i, j) + avgS(i - 1, j) + avgS(i, j - 1) + avgS(i - 1, j - 1));
      } else {
        sh = A.S[a2 + wk];
        bu = bvf[a2 + wk];
      }
      A.SH[a2 + wk] = sh;
      A.BU[a2 + wk] = bu;
    }
  }
  // ---- horizontal advection, third-order upstream bias (:444-640)
  for (int k = 1; k <= N - 1; k++) {
    const long wk = (long)k * nij;
    const double *T = tke + L3 + wk, *G = gls + L3 + wk;
    auto fx = [&](int ii, double &FXK, double &FXP) {
      const long q = (long)(ii - LBi) + (long)(j - LBj) * ni;
      const double cff = 0.5 * (Huon[q + (long)(k - 1) * nij] + Huon[q + (long)k * nij]);
      double c1, c2;
      if (cff > 0.0) {
        c1 = gx(T, 0, ii, j, LBi, LBj) - gx(T, 0, ii - 1, j, LBi, LBj);          // curvK(ii-1)
        c2 = gx(G, 0, ii, j, LBi, LBj) - gx(G, 0, ii - 1, j, LBi, LBj);
      } else {
        c1 = gx(T, 0, ii + 1, j, LBi, LBj) - gx(T, 0, ii, j, LBi, LBj);          // curvK(ii)
        c2 = gx(G, 0, ii + 1, j, LBi, LBj) - gx(G, 0, ii, j, LBi, LBj);
      }
      FXK = cff * 0.5 * (T[q - 1] + T[q] - Gadv * c1);
      FXP = cff * 0.5 * (G[q - 1] + G[q] - Gadv * c2);
    };
    auto fe = [&](int jj, double &FEK, double &FEP) {
      const long q = (long)(i - LBi) + (long)(jj - LBj) * ni;
      const double cff = 0.5 * (Hvom[q + (long)(k - 1) * nij] + Hvom[q + (long)k * nij]);
      double c1, c2;
      if (cff > 0.0) {
        c1 = gy(T, 0, i, jj, LBi, LBj) - gy(T, 0, i, jj - 1, LBi, LBj);
        c2 = gy(G, 0, i, jj, LBi, LBj) - gy(G, 0, i, jj - 1, LBi, LBj);
      } else {
        c1 = gy(T, 0, i, jj + 1, LBi, LBj) - gy(T, 0, i, jj, LBi, LBj);
        c2 = gy(G, 0, i, jj + 1, LBi, LBj) - gy(G, 0, i, jj, LBi, LBj);
      }
      FEK = cff * 0.5 * (T[q - ni] + T[q] - Gadv * c1);
      FEP = cff * 0.5 * (G[q - ni] + G[q] - Gadv * c2);
    };
    double FXK0, FXP0, FXK1, FXP1, FEK0, FEP0, FEK1, FEP1;
    fx(i, FXK0, FXP0); fx(i + 1, FXK1, FXP1);
    fe(j, FEK0, FEP0); fe(j + 1, FEK1, FEP1);
    double tv = Tn[a2 + wk] - cdt * (FXK1 - FXK0 + FEK1 - FEK0);
    if (!my25) tv = fmax(tv, gls_Kmin);                  // my25_corstep.F:511-514 has no floor
    double gv = Gn[a2 + wk] - cdt * (FXP1 - FXP0 + FEP1 - FEP0);
    if (!my25) gv = fmax(gv, gls_Pmin);
    Tn[a2 + wk] = tv;
    Gn[a2 + wk] = gv;
  }
  // ---- vertical advection (:644-700), fourth-order centred with the one-sided end formulas
  {
    const double *T = tke + L3, *G = gls + L3;
    auto vflux = [&](int kk, double &FCK, double &FCP) {
      const long w = a2 + (long)kk * nij;
      const double cff = 0.5 * (Wv[w] + Wv[w - nij]);
      if (kk == 1) {
        FCK = cff * (1.0 / 3.0 * T[w - nij] + 5.0 / 6.0 * T[w] - 1.0 / 6.0 * T[w + nij]);
        FCP = cff * (1.0 / 3.0 * G[w - nij] + 5.0 / 6.0 * G[w] - 1.0 / 6.0 * G[w + nij]);
      } else if (kk == N) {
        FCK = cff * (1.0 / 3.0 * T[w] + 5.0 / 6.0 * T[w - nij] - 1.0 / 6.0 * T[w - 2 * nij]);
        FCP = cff * (1.0 / 3.0 * G[w] + 5.0 / 6.0 * G[w - nij] - 1.0 / 6.0 * G[w - 2 * nij]);
      } else {
        FCK = cff * (7.0 / 12.0 * (T[w - nij] + T[w]) - 1.0 / 12.0 * (T[w - 2 * nij] + T[w + nij]));
        FCP = cff * (7.0 / 12.0 * (G[w - nij] + G[w]) - 1.0 / 12.0 * (G[w - 2 * nij] + G[w + nij]));
      }
    };
    double FKlo, FPlo;
    vflux(1, FKlo, FPlo);
    for (int k = 1; k <= N - 1; k++) {
      const long wk = (long)k * nij;
      double FKhi, FPhi;
      vflux(k + 1, FKhi, FPhi);
      double tv = Tn[a2 + wk] - cdt * (FKhi - FKlo);
      if (!my25) tv = fmax(tv, gls_Kmin);                // my25_corstep.F:569-576
      double gv = Gn[a2 + wk] - cdt * (FPhi - FPlo);
      if (!my25) gv = fmax(gv, gls_Pmin);
      Tn[a2 + wk] = tv;
      Gn[a2 + wk] = gv;
      FKlo = FKhi; FPlo = FPhi;
    }
  }
  if (my25) {
    // ---- MY25_MIXING, my25_corstep.F:580-770: Mellor and Yamada (1982) level 2.5 with the Galperin et al. (1988)
    //      stability functions (Kantha and Clayson's Sm under KANTHA_CLAYSON); tke = q2, gls = q2l
    const double my_B1 = 16.6, my_E1 = 1.8, my_E2 = 1.33, my_Gh0 = 0.0233, my_Sq = 0.2, my_lmax = 0.53, my_qmin = 1.0E-8;
    const double *Ts = tke + Ls, *Gs = gls + Ls;
    const long wN = a2 + (long)N * nij, w0 = a2;
    {
      const double cff = -0.5 * dt;
      for (int k = 1; k <= N; k++) {
        const long w = a2 + (long)k * nij, r = a2 + (long)(k - 1) * nij;
        A.FCK[w] = cff * (Akk[w] + Akk[w - nij]) / Hz[r];
      }
    }
    const double cff3 = my_E2 / (vonKar * vonKar);
    const double zN = z_w[wN], z0 = z_w[w0];
    for (int k = 1; k <= N - 1; k++) {
      const long w = a2 + (long)k * nij, r = a2 + (long)(k - 1) * nij;
      const double bu = A.BU[w];
      const double strat2 = ((bu > -5.0E-5) && (bu < 0.0)) ? 0.0 : bu;
      const double Qprod = A.SH[w] * (Akv[w] - Akv_bak) - strat2 * (Akt[w] - AktT_bak);
      const double Ls_unlmt = fmax(eps, Gs[w] / (fmax(Ts[w], eps)));
      const double cff1 = 0.5 * (Hz[r] + Hz[r + nij]);
      Tn[w] = Tn[w] + dt * cff1 * Qprod * 2.0;
      Gn[w] = Gn[w] + dt * cff1 * Qprod * my_E1 * Ls_unlmt;
      const double Qdiss = dt * sqrt(Ts[w]) / (my_B1 * Ls_unlmt);
      const double zk = z_w[w];
      const double cff = Ls_unlmt * (1.0 / (zN - zk) + 1.0 / (zk - z0));
      const double Wscale = 1.0 + cff3 * cff * cff;
      const double FCKk = A.FCK[w], FCKk1 = A.FCK[w + nij];
      A.BCK[w] = cff1 * (1.0 + 2.0 * Qdiss) - FCKk - FCKk1;
      A.BCP[w] = cff1 * (1.0 + Wscale * Qdiss) - FCKk - FCKk1;
    }
    {
      const double sx = c->F.sustr[a2] + c->F.sustr[a2 + 1], sy = c->F.svstr[a2] + c->F.svstr[a2 + ni];
      const double bx = c->F.bustr[a2] + c->F.bustr[a2 + 1], by = c->F.bvstr[a2] + c->F.bvstr[a2 + ni];
      Tn[wN] = K.my_B1p2o3 * 0.5 * sqrt(sx * sx + sy * sy);
      Gn[wN] = 0.0;
      Tn[w0] = K.my_B1p2o3 * 0.5 * sqrt(bx * bx + by * by);
      Gn[w0] = 0.0;
    }
    // the two tridiagonal systems (:649-692): elimination from the top, substitution from the bottom
    for (int sys = 0; sys < 2; sys++) {
      double *X = sys == 0 ? Tn : Gn;
      const double *BC = sys == 0 ? A.BCK : A.BCP;
      const long wt = a2 + (long)(N - 1) * nij;
      double cff = 1.0 / BC[wt];
      double CFp = cff * A.FCK[wt];
      A.CF[wt] = CFp;
      double Xp = cff * (X[wt] - A.FCK[wN] * X[wN]);
      X[wt] = Xp;
      for (int k = N - 2; k >= 1; k--) {
        const long w = a2 + (long)k * nij;
        const double FCK1 = A.FCK[w + nij];
        cff = 1.0 / (BC[w] - CFp * FCK1);
        CFp = cff * A.FCK[w];
        A.CF[w] = CFp;
        Xp = cff * (X[w] - FCK1 * Xp);
        X[w] = Xp;
      }
      double Xm = X[w0];
      for (int k = 1; k <= N - 1; k++) {
        const long w = a2 + (long)k * nij;
        Xm = X[w] - A.CF[w] * Xm;
        X[w] = Xm;
      }
    }
    // mixing coefficients (:699-770)
    for (int k = 1; k <= N - 1; k++) {
      const long w = a2 + (long)k * nij;
      const double tk = fmax(Tn[w], my_qmin), gk = fmax(Gn[w], my_qmin);
      const double buoy2 = A.BU[w];
      const double Ls_unlmt = gk / tk;
      const double Ls_lmt = fmin(Ls_unlmt, my_lmax * sqrt(tk / (fmax(0.0, buoy2) + eps)));
      const double Gh = fmin(my_Gh0, -buoy2 * Ls_lmt * Ls_lmt / tk);
      const double cff = 1.0 - K.my_Sh2 * Gh;
      const double Sh = K.my_Sh1 / cff;
      double Sm;
      if (p.gls_stability == GLS_KANTHA_CLAYSON) Sm = (K.my_B1pm1o3 + Sh * Gh * K.my_Sm4) / (1.0 - K.my_Sm2 * Gh);
      else Sm = (K.my_Sm3 + Sh * Gh * K.my_Sm4) / (1.0 - K.my_Sm2 * Gh);
      const double ql = 0.5 * (Ls_lmt * sqrt(tk) + Lscale[w] * sqrt(Ts[w]));
      Tn[w] = tk;
      Gn[w] = gk;
      Akv[w] = Akv_bak + ql * Sm;
      for (int it = 0; it < b.NAT; it++) Akt[w + (long)it * n3w] = p.Akt_bak[it] + ql * Sh;
      Akk[w] = Akk_bak + ql * my_Sq;
      Lscale[w] = Ls_lmt;
    }
    return;
  }
  // ---- vertical mixing terms, production, dissipation (:706-800)
  const double *Ts = tke + Ls, *Gs = gls + Ls;
  {
    const double cff = -0.5 * dt;
    for (int k = 2; k <= N - 1; k++) {
      const long w = a2 + (long)k * nij, r = a2 + (long)(k - 1) * nij;
      A.FCK[w] = cff * (Akk[w] + Akk[w - nij]) / Hz[r];
      A.FCP[w] = cff * (Akp[w] + Akp[w - nij]) / Hz[r];
    }
    A.FCP[a2 + nij] = 0.0; A.FCP[a2 + (long)N * nij] = 0.0;
    A.FCK[a2 + nij] = 0.0; A.FCK[a2 + (long)N * nij] = 0.0;
  }
  for (int k = 1; k <= N - 1; k++) {
    const long w = a2 + (long)k * nij, r = a2 + (long)(k - 1) * nij;
    const double strat2 = A.BU[w], shear2 = A.SH[w];
    const double gls_c3 = (strat2 > 0.0) ? p.gls_c3m : p.gls_c3p;
    const double akt = Akt[w];
    double Kprod = shear2 * (Akv[w] - Akv_bak) - strat2 * (akt - AktT_bak);
    double Pprod = gls_c1 * shear2 * (Akv[w] - Akv_bak) - gls_c3 * strat2 * (akt - AktT_bak);
    double cff1 = 1.0;
    if (Kprod < 0.0) { Kprod = Kprod + strat2 * (akt - AktT_bak); cff1 = 0.0; }
    double cff2 = 1.0;
    if (Pprod < 0.0) { Pprod = Pprod + gls_c3 * strat2 * (akt - AktT_bak); cff2 = 0.0; }
    const double cff = 0.5 * (Hz[r] + Hz[r + nij]);
    const double tks = Ts[w], gss = Gs[w];
    Tn[w] = Tn[w] + dt * cff * Kprod;
    Gn[w] = Gn[w] + dt * cff * Pprod * gss / fmax(tks, gls_Kmin);
    double wall_fac = 1.0;
    if (f.Lmy25) {
      const double zN = z_w[a2 + (long)N * nij], z0 = z_w[a2], zk = z_w[w];
      const double q1 = pow(gss, f.gls_exp1) * f.cmu_fac1 * pow(tks, -f.tke_exp1) * (1.0 / (zk - z0));
      const double q2 = pow(gss, f.gls_exp1) * f.cmu_fac1 * pow(tks, -f.tke_exp1) * (1.0 / (zN - zk));
      wall_fac = 1.0 + K.E2 / (vonKar * vonKar) * (q1 * q1) + 0.25 / (vonKar * vonKar) * (q2 * q2);
    }
    const double FCKk = A.FCK[w], FCKk1 = A.FCK[w + nij], FCPk = A.FCP[w], FCPk1 = A.FCP[w + nij];
    A.BCK[w] = cff * (1.0 + dt * pow(gss, -f.gls_exp1) * f.cmu_fac2 * pow(tks, f.tke_exp2) +
                      dt * (1.0 - cff1) * strat2 * (akt - AktT_bak) / tks) - FCKk - FCKk1;
    A.BCP[w] = cff * (1.0 + dt * gls_c2 * wall_fac * pow(gss, -f.gls_exp1) * f.cmu_fac2 * pow(tks, f.tke_exp2) +
                      dt * (1.0 - cff2) * gls_c3 * strat2 * (akt - AktT_bak) / tks) - FCPk - FCPk1;
  }
  // ---- Dirichlet surface and bottom values (:806-860)
  const double sus = (c->F.sustr[a2] + c->F.sustr[a2 + 1]), svs = (c->F.svstr[a2] + c->F.svstr[a2 + ni]);
  const double bus = (c->F.bustr[a2] + c->F.bustr[a2 + 1]), bvs = (c->F.bvstr[a2] + c->F.bvstr[a2 + ni]);
  const long wN = a2 + (long)N * nij, w0 = a2;
  const double tkeN = fmax(f.cmu_fac3 * 0.5 * sqrt(sus * sus + svs * svs), gls_Kmin);
  const double tke0 = fmax(f.cmu_fac3 * 0.5 * sqrt(bus * bus + bvs * bvs), gls_Kmin);
  Tn[wN] = tkeN;
  Tn[w0] = tke0;
  const double Zos_eff = f.Zos_min;
  const double Zob_min = fmax(c->F.ZoBot[a2], 0.0001);
  Gn[wN] = fmax(f.cmu0p * pow(tkeN, gls_m) * pow(f.L_sft * Zos_eff, gls_n), gls_Pmin);
  {
    const double cff = f.gls_fac4 * pow(vonKar * Zob_min, gls_n);
    Gn[w0] = fmax(cff * pow(tke0, gls_m), gls_Pmin);
  }
  // ---- tri-diagonal system for tke (:864-895): elimination from the top, substitution from the bottom
  {
    const double tke_fluxt = 0.0, tke_fluxb = 0.0;
    const long wt = a2 + (long)(N - 1) * nij;
    double cff = 1.0 / A.BCK[wt];
    double CFp = cff * A.FCK[wt];
    A.CF[wt] = CFp;
    double Tp = cff * (Tn[wt] + tke_fluxt);
    Tn[wt] = Tp;
    for (int k = N - 2; k >= 1; k--) {
      const long w = a2 + (long)k * nij;
      const double FCK1 = A.FCK[w + nij];
      cff = 1.0 / (A.BCK[w] - CFp * FCK1);
      CFp = cff * A.FCK[w];
      A.CF[w] = CFp;
      Tp = cff * (Tn[w] - FCK1 * Tp);
      Tn[w] = Tp;
    }
    Tn[a2 + nij] = Tn[a2 + nij] - cff * tke_fluxb;
    double Tm = Tn[a2 + nij];
    for (int k = 2; k <= N - 1; k++) {
      const long w = a2 + (long)k * nij;
      Tm = Tn[w] - A.CF[w] * Tm;
      Tn[w] = Tm;
    }
  }
  // ---- tri-diagonal system for gls (:899-960)
  {
    const long wt = a2 + (long)(N - 1) * nij;
    double cffa = 0.5 * (Tn[wN] + Tn[wt]);
    const double gls_fluxt = dt * f.gls_fac3 * pow(cffa, gls_m) * pow(f.L_sft, gls_n) *
                             pow(Zos_eff + 0.5 * Hz[a2 + (long)(N - 1) * nij], gls_n - 1.0) * 0.5 * (Akp[wN] + Akp[wt]);
    cffa = 0.5 * (Tn[w0] + Tn[a2 + nij]);
    const double gls_fluxb = dt * f.gls_fac2 * (pow(cffa, gls_m)) * pow(0.5 * Hz[a2] + Zob_min, gls_n - 1.0) * 0.5 *
                             (Akp[w0] + Akp[a2 + nij]);
    double cff = 1.0 / A.BCP[wt];
    double CFp = cff * A.FCP[wt];
    A.CF[wt] = CFp;
    double Gp = cff * (Gn[wt] - gls_fluxt);
    Gn[wt] = Gp;
    for (int k = N - 2; k >= 1; k--) {
      const long w = a2 + (long)k * nij;
      const double FCP1 = A.FCP[w + nij];
      cff = 1.0 / (A.BCP[w] - CFp * FCP1);
      CFp = cff * A.FCP[w];
      A.CF[w] = CFp;
      Gp = cff * (Gn[w] - FCP1 * Gp);
      Gn[w] = Gp;
    }
    Gn[a2 + nij] = Gn[a2 + nij] - cff * gls_fluxb;
    double Gm = Gn[a2 + nij];
    for (int k = 2; k <= N - 1; k++) {
      const long w = a2 + (long)k * nij;
      Gm = Gn[w] - A.CF[w] * Gm;
      Gn[w] = Gm;
    }
  }
  // ---- vertical mixing coefficients (:964-1095)
  const int NAT = b.NAT;
  for (int k = 1; k <= N - 1; k++) {
    const long w = a2 + (long)k * nij;
    const double buoy2 = A.BU[w], shear2 = A.SH[w];
    double tk = fmax(Tn[w], gls_Kmin);
    double gk = fmax(Gn[w], gls_Pmin);
    const double lim = f.gls_fac5 * pow(tk, f.tke_exp4) * pow(sqrt(fmax(0.0, buoy2)) + eps, -gls_n);
    if (gls_n >= 0.0) gk = fmin(gk, lim);
    else gk = fmax(gk, lim);
    const double Ls_unlmt = fmax(eps, pow(gk, f.gls_exp1) * f.cmu_fac1 * pow(tk, -f.tke_exp1));
    double Ls_lmt;
    if (buoy2 > 0.0) Ls_lmt = fmin(Ls_unlmt, sqrt(0.56 * tk / (fmax(0.0, buoy2) + eps)));
    else Ls_lmt = Ls_unlmt;
    gk = fmax(f.cmu0p * pow(tk, gls_m) * pow(Ls_lmt, gls_n), gls_Pmin);
    double Gh = fmin(K.Gh0, -buoy2 * Ls_lmt * Ls_lmt / (2.0 * tk));
    Gh = fmin(Gh, Gh - ((Gh - K.Ghcri) * (Gh - K.Ghcri)) / (Gh + K.Gh0 - 2.0 * K.Ghcri));
    Gh = fmax(Gh, K.Ghmin);
    double Sm, Sh;
    if (p.gls_stability == GLS_CANUTO_A || p.gls_stability == GLS_CANUTO_B) {
      const double f6 = f.gls_fac6;
      double Gm = (K.b0 / f6 - K.b1 * Gh + K.b3 * f6 * (Gh * Gh)) / (K.b2 - K.b4 * f6 * Gh);
      Gm = fmin(Gm, shear2 * Ls_lmt * Ls_lmt / (2.0 * tk));
      const double cff = K.b0 - K.b1 * f6 * Gh + K.b2 * f6 * Gm + K.b3 * (f6 * f6) * (Gh * Gh) - K.b4 * (f6 * f6) * Gh * Gm +
                         K.b5 * (f6 * f6) * Gm * Gm;
      Sm = (K.s0 - K.s1 * f6 * Gh + K.s2 * f6 * Gm) / cff;
      Sh = (K.s4 - K.s5 * f6 * Gh + K.s6 * f6 * Gm) / cff;
      Sm = fmax(Sm, 0.0);
      Sh = fmax(Sh, 0.0);
      Sm = Sm * f.sqrt2 / (gls_cmu0 * gls_cmu0 * gls_cmu0);
      Sh = Sh * f.sqrt2 / (gls_cmu0 * gls_cmu0 * gls_cmu0);
    } else if (p.gls_stability == GLS_KANTHA_CLAYSON) {
      const double cff = 1.0 - K.my_Sh2 * Gh;
      Sh = K.my_Sh1 / cff;
      Sm = (K.my_B1pm1o3 + K.my_Sm4 * Sh * Gh) / (1.0 - K.my_Sm2 * Gh);
    } else {
      const double cff = 1.0 - K.my_Sh2 * Gh;
      Sh = K.my_Sh1 / cff;
      Sm = (K.my_Sm3 + Sh * Gh * K.my_Sm4) / (1.0 - K.my_Sm2 * Gh);
    }
    const double ql = f.sqrt2 * 0.5 * (Ls_lmt * sqrt(tk) + Lscale[w] * sqrt(Ts[w]));
    Tn[w] = tk;
    Gn[w] = gk;
    Akv[w] = Akv_bak + Sm * ql;
    for (int it = 0; it < NAT; it++) Akt[w + (long)it * n3w] = p.Akt_bak[it] + Sh * ql;
    Akk[w] = Akk_bak + Sm * ql / gls_sigk;
    Akp[w] = Akp_bak + Sm * ql * f.ogls_sigp;
    Lscale[w] = Ls_lmt;
  }
  Akv[wN] = Akv_bak + f.L_sft * Zos_eff * gls_cmu0 * sqrt(Tn[wN]);
  Akv[w0] = Akv_bak + vonKar * Zob_min * gls_cmu0 * sqrt(Tn[w0]);
  Akk[wN] = Akk_bak + Akv[wN] / gls_sigk;
  Akk[w0] = Akk_bak + Akv[w0] / gls_sigk;
  Akp[wN] = Akp_bak + Akv[wN] * f.ogls_sigp;
  Akp[w0] = Akp_bak + Akv[w0] / gls_sigp;
  for (int it = 0; it < NAT; it++) {
    Akt[wN + (long)it * n3w] = p.Akt_bak[it];
    Akt[w0 + (long)it * n3w] = p.Akt_bak[it];
  }
}

// lateral conditions of Akv and Akt as gls_corstep.F:1100-1185 writes them: on every tile that holds a domain edge
// (periodic or not), west: (Istr-1) <- (Istr); east: (Iend-1) <- (Iend) -- an interior column, as written; south,
// north; then the corners.  phase 0 W, 1 E, 2 S and N, 3 corners: each reads what the one before left.
__global__ void k_gls_akbc(const RomsDev *__restrict__ c, int phase)
{
  DEV_PROLOGUE(c)
  const int k = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int Istr = b.Istr, Iend = b.Iend, Jstr = b.Jstr, Jend = b.Jend;
  const int nI = Iend - Istr + 1, nJ = Jend - Jstr + 1;
  const int nf = 1 + b.NAT;
  for (int fq = 0; fq < nf; fq++) {
    double *F = fq == 0 ? c->F.Akv : c->F.Akt + (long)(fq - 1) * n3w;
    if (phase == 0 && b.west_edge && q < nJ) F[I3W(Istr - 1, Jstr + q, k)] = F[I3W(Istr, Jstr + q, k)];
    if (phase == 1 && b.east_edge && q < nJ) F[I3W(Iend - 1, Jstr + q, k)] = F[I3W(Iend, Jstr + q, k)];
    if (phase == 2 && q < nI) {
      if (b.south_edge) F[I3W(Istr + q, Jstr - 1, k)] = F[I3W(Istr + q, Jstr, k)];
      if (b.north_edge) F[I3W(Istr + q, Jend + 1, k)] = F[I3W(Istr + q, Jend, k)];
    }
    if (phase == 3 && q == 0) {
      if (b.south_edge && b.west_edge) F[I3W(Istr - 1, Jstr - 1, k)] = 0.5 * (F[I3W(Istr, Jstr - 1, k)] + F[I3W(Istr - 1, Jstr, k)]);
      if (b.south_edge && b.east_edge) F[I3W(Iend + 1, Jstr - 1, k)] = 0.5 * (F[I3W(Iend, Jstr - 1, k)] + F[I3W(Iend + 1, Jstr, k)]);
      if (b.north_edge && b.west_edge) F[I3W(Istr - 1, Jend + 1, k)] = 0.5 * (F[I3W(Istr, Jend + 1, k)] + F[I3W(Istr - 1, Jend, k)]);
      if (b.north_edge && b.east_edge) F[I3W(Iend + 1, Jend + 1, k)] = 0.5 * (F[I3W(Iend, Jend + 1, k)] + F[I3W(Iend + 1, Jend, k)]);
    }
  }
}

}  // namespace

extern "C" int roms_hip_gls_prestep(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_gls_prestep");
  if (rc) return rc;
  if ((rc = gls_check("roms_hip_gls_prestep"))) return rc;
  ScopedTimer tm("gls_prestep");
  const roms_bounds_t &b = g_ctx.b;
  if (b.N < 3) return roms_fail("roms_hip_gls_prestep", "N < 3");
  const long n3w = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * (b.N + 1);
  hipLaunchKernelGGL(k_gls_prestep, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream, g_ctx.devc, *s);
  KERNEL_CHECK("k_gls_prestep");
  if ((rc = tke_bc(3))) return rc;
  halo_batch_begin();
  halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_tke] + 2L * n3w);
  halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_gls] + 2L * n3w);
  return halo_batch_end();
}

extern "C" int roms_hip_gls_corstep(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_gls_corstep");
  if (rc) return rc;
  if ((rc = gls_check("roms_hip_gls_corstep"))) return rc;
  ScopedTimer tm("gls_corstep");
  const roms_bounds_t &b = g_ctx.b;
  if (b.N < 3) return roms_fail("roms_hip_gls_corstep", "N < 3");
  const long n3w = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * (b.N + 1);
  GlsArgs A;
  A.s = *s;
  A.K = gls_constants(g_ctx.p.gls_stability);
  A.f = gls_factors(g_ctx.p);
  double **ws = g_ctx.hostc.ws3;
  A.S = ws[0]; A.SH = ws[1]; A.BU = ws[2]; A.FCK = ws[3]; A.FCP = ws[4]; A.BCK = ws[5]; A.BCP = ws[6]; A.CF = ws[7];
  // the spline sweeps of the shear use three of the arrays the column kernel fills later
  hipLaunchKernelGGL(k_gls_shear, grid2d(b.Iendp1 - b.Istrm1 + 1, b.Jendp1 - b.Jstrm1 + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, s->nstp, A.S, ws[3], ws[4], ws[5]);
  KERNEL_CHECK("k_gls_shear");
  hipLaunchKernelGGL(k_gls_corstep, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream, g_ctx.devc, A);
  KERNEL_CHECK("k_gls_corstep");
  {
    const int nI = b.Iend - b.Istr + 1, nJ = b.Jend - b.Jstr + 1;
    if (b.west_edge) hipLaunchKernelGGL(k_gls_akbc, dim3((nJ + 63) / 64, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, 0);
    if (b.east_edge) hipLaunchKernelGGL(k_gls_akbc, dim3((nJ + 63) / 64, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, 1);
    if (b.south_edge || b.north_edge)
      hipLaunchKernelGGL(k_gls_akbc, dim3((nI + 63) / 64, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, 2);
    if ((b.south_edge || b.north_edge) && (b.west_edge || b.east_edge))
      hipLaunchKernelGGL(k_gls_akbc, dim3(1, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, 3);
    KERNEL_CHECK("k_gls_akbc");
  }
  if ((rc = tke_bc(s->nnew))) return rc;
  halo_batch_begin();
  halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_tke] + (long)(s->nnew - 1) * n3w);
  halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_gls] + (long)(s->nnew - 1) * n3w);
  halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_Akv]);
  for (int it = 0; it < b.NAT; it++) halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_Akt] + (long)it * n3w);
  return halo_batch_end();
}
