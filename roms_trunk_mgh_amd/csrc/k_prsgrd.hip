// k_prsgrd.hip -- baroclinic pressure gradient, prsgrd32_tile
// (ROMS/Nonlinear/prsgrd32.h:106-423; DJ_GRADPS: density Jacobian with
// harmonic-mean limited cubic reconstruction, Shchepetkin & McWilliams 2003).
// First writer of ru,rv(:,:,1:N,nrhs).
//
// Two kernels:
//   k_prsgrd_P   one thread per water column, top-down recursion for the
//                dynamic pressure P with 3-level sliding windows in registers
//                (reads rho, z_r, z_w(N); writes P to device scratch);
//   k_prsgrd_uv  one thread per (i,j,k): horizontal harmonic-mean slopes and
//                the XI/ETA pressure-gradient terms (reads P, rho, z_r, Hz with
//                a 2-point i/j halo served by L1/L2; writes ru, rv).
// Algorithmic traffic: rho, z_r, Hz read, P written+read, ru, rv written
// = 7 field passes (56 B per cell).
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

__device__ __forceinline__ double harm(double a, double bq, double eps)
{
  // cff=2*a*b; cff>eps ? cff/(a+b) : 0  (prsgrd32.h:243-249)
  const double cff = 2.0 * a * bq;
  return (cff > eps) ? cff / (a + bq) : 0.0;
}
__device__ __forceinline__ double harm_inv(double a, double bq, double eps)
{
  // the horizontal form multiplies by a reciprocal (prsgrd32.h:306-320)
  const double cff = 2.0 * a * bq;
  if (cff > eps) { const double cff1 = 1.0 / (a + bq); return cff * cff1; }
  return 0.0;
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_prsgrd_P(const RomsDev *__restrict__ c, double *__restrict__ P)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrU - 1 + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrV - 1 + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const gcd_t rho = (gcd_t)(c->F.rho);
  const gcd_t z_r = (gcd_t)(c->F.z_r);
  const double eps = 1.0E-10, OneFifth = 0.2, OneTwelfth = 1.0 / 12.0;
  const double g = c->p.g, GRho = g / c->p.rho0, HalfGRho = 0.5 * GRho;
  const long c0 = I2(i, j);
  // raw differences: dRr(k)=rho(k+1)-rho(k), k=1..N-1; dRr(N)=dRr(N-1); dRr(0)=dRr(1)
  double rho_k1 = rho[c0 + (long)(N - 1) * nij];      // rho(N)
  double zr_k1 = z_r[c0 + (long)(N - 1) * nij];
  double rho_k = rho[c0 + (long)(N - 2) * nij];       // rho(N-1)
  double zr_k = z_r[c0 + (long)(N - 2) * nij];
  const double zwN = GF(z_w)[c0 + (long)N * nij];
  // surface level
  {
    const double cff1 = 1.0 / (zr_k1 - zr_k);
    const double cff2 = 0.5 * (rho_k1 - rho_k) * (zwN - zr_k1) * cff1;
    // ATM_PRESS (prsgrd32.h:229-232, :264-266): + (100 / rho0) (Pair - 1 atm), Pair in mb
    if (c->p.atm_press)
      P[c0 + (long)(N - 1) * nij] = g * zwN + (100.0 / c->p.rho0) * (GF(Pair)[c0] - 1013.25) + GRho * (rho_k1 + cff2) * (zwN - zr_k1);
    else
      P[c0 + (long)(N - 1) * nij] = g * zwN + GRho * (rho_k1 + cff2) * (zwN - zr_k1);
  }
  double Pk1 = P[c0 + (long)(N - 1) * nij];
  // limited slopes at level N: harm(raw(N), raw(N-1)) with raw(N)=raw(N-1)
  double rawR_k = rho_k1 - rho_k, rawZ_k = zr_k1 - zr_k;       // raw(N-1)
  double dR_k1 = harm(rawR_k, rawR_k, eps);                     // dR(N)
  double dZ_k1 = 2.0 * rawZ_k * rawZ_k / (rawZ_k + rawZ_k);     // dZ(N)
  for (int k = N - 1; k >= 1; k--) {
    // raw(k-1): k-1 >= 1 -> rho(k)-rho(k-1); k-1 == 0 -> raw(1)
    double rawR_km1, rawZ_km1, rho_km1 = 0.0, zr_km1 = 0.0;
    if (k >= 2) {
      rho_km1 = rho[c0 + (long)(k - 2) * nij];
      zr_km1 = z_r[c0 + (long)(k - 2) * nij];
      rawR_km1 = rho_k - rho_km1;
      rawZ_km1 = zr_k - zr_km1;
    } else { rawR_km1 = rawR_k; rawZ_km1 = rawZ_k; }
    const double dR_k = harm(rawR_k, rawR_km1, eps);
    const double dZ_k = 2.0 * rawZ_k * rawZ_km1 / (rawZ_k + rawZ_km1);
    const double Pk = Pk1 +
        HalfGRho * ((rho_k1 + rho_k) * (zr_k1 - zr_k) -
                    OneFifth *
                    ((dR_k1 - dR_k) * (zr_k1 - zr_k - OneTwelfth * (dZ_k1 + dZ_k)) -
                     (dZ_k1 - dZ_k) * (rho_k1 - rho_k - OneTwelfth * (dR_k1 + dR_k))));
    P[c0 + (long)(k - 1) * nij] = Pk;
    Pk1 = Pk;
    rho_k1 = rho_k; zr_k1 = zr_k; rho_k = rho_km1; zr_k = zr_km1;
    rawR_k = rawR_km1; rawZ_k = rawZ_km1;
    dR_k1 = dR_k; dZ_k1 = dZ_k;
  }
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_prsgrd_uv(const RomsDev *__restrict__ c, const double *__restrict__ P, int nrhs)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  const int k = XB.z + 1;
  if (i > b.Iend || j > b.Jend) return;
  const gcd_t rho = (gcd_t)(c->F.rho);
  const gcd_t z_r = (gcd_t)(c->F.z_r);
  const gcd_t Hz = (gcd_t)(c->F.Hz);
  const double eps = 1.0E-10, OneFifth = 0.2, OneTwelfth = 1.0 / 12.0;
  const double HalfGRho = 0.5 * (c->p.g / c->p.rho0);
  const bool masking = c->p.masking != 0;
  const long ck = I3(i, j, k);
  const double r0 = rho[ck], z0 = z_r[ck], hz0 = Hz[ck], P0 = P[ck];
  if (i >= b.IstrU) {
    const double rm2 = rho[ck - 2], rm1 = rho[ck - 1], rp1 = rho[ck + 1];
    const double zm2 = z_r[ck - 2], zm1 = z_r[ck - 1], zp1 = z_r[ck + 1];
    double aux_m1 = zm1 - zm2, aux_0 = z0 - zm1, aux_p1 = zp1 - z0;
    double FC_m1 = rm1 - rm2, FC_0 = r0 - rm1, FC_p1 = rp1 - r0;
    if (masking) {                                          // MASKING, prsgrd32.h:300-306
      const gcd_t um = (gcd_t)c->F.umask;
      const long a2 = I2(i, j);
      const double m_m1 = um[a2 - 1], m_0 = um[a2], m_p1 = um[a2 + 1];
      aux_m1 = aux_m1 * m_m1; aux_0 = aux_0 * m_0; aux_p1 = aux_p1 * m_p1;
      FC_m1 = FC_m1 * m_m1; FC_0 = FC_0 * m_0; FC_p1 = FC_p1 * m_p1;
    }
    const double dZx0 = harm_inv(aux_0, aux_p1, eps), dZxm = harm_inv(aux_m1, aux_0, eps);
    const double dRx0 = harm_inv(FC_0, FC_p1, eps), dRxm = harm_inv(FC_m1, FC_0, eps);
    const gd_t ru = (gd_t)(c->F.ru + (long)(nrhs - 1) * n3w);
    double r = GF(on_u)[I2(i, j)] * 0.5 * (hz0 + Hz[ck - 1]) *
        (P[ck - 1] - P0 -
         HalfGRho * ((r0 + rm1) * (z0 - zm1) -
                     OneFifth * ((dRx0 - dRxm) * (z0 - zm1 - OneTwelfth * (dZx0 + dZxm)) -
                                 (dZx0 - dZxm) * (r0 - rm1 - OneTwelfth * (dRx0 + dRxm)))));
    if (c->p.wet_dry) r = r * GF(umask_wet)[I2(i, j)];      // WET_DRY, prsgrd32.h:346-348
    ru[I3W(i, j, k)] = r;
  }
  if (j >= b.JstrV) {
    const double rm1 = rho[ck - ni], rp1 = rho[ck + ni];
    const double zm1 = z_r[ck - ni], zp1 = z_r[ck + ni];
    // aux(j-1) needs row j-2: exists for j-1 >= JstrV-1, i.e. always here
    const double rm2 = rho[ck - 2 * ni], zm2 = z_r[ck - 2 * ni];
    double aux_m1 = zm1 - zm2, aux_0 = z0 - zm1, aux_p1 = zp1 - z0;
    double FC_m1 = rm1 - rm2, FC_0 = r0 - rm1, FC_p1 = rp1 - r0;
    if (masking) {                                          // MASKING, prsgrd32.h:364-370
      const gcd_t vm = (gcd_t)c->F.vmask;
      const long a2 = I2(i, j);
      const double m_m1 = vm[a2 - ni], m_0 = vm[a2], m_p1 = vm[a2 + ni];
      aux_m1 = aux_m1 * m_m1; aux_0 = aux_0 * m_0; aux_p1 = aux_p1 * m_p1;
      FC_m1 = FC_m1 * m_m1; FC_0 = FC_0 * m_0; FC_p1 = FC_p1 * m_p1;
    }
    const double dZx0 = harm_inv(aux_0, aux_p1, eps), dZxm = harm_inv(aux_m1, aux_0, eps);
    const double dRx0 = harm_inv(FC_0, FC_p1, eps), dRxm = harm_inv(FC_m1, FC_0, eps);
    const gd_t rv = (gd_t)(c->F.rv + (long)(nrhs - 1) * n3w);
    double r = GF(om_v)[I2(i, j)] * 0.5 * (hz0 + Hz[ck - ni]) *
        (P[ck - ni] - P0 -
         HalfGRho * ((r0 + rm1) * (z0 - zm1) -
                     OneFifth * ((dRx0 - dRxm) * (z0 - zm1 - OneTwelfth * (dZx0 + dZxm)) -
                                 (dZx0 - dZxm) * (r0 - rm1 - OneTwelfth * (dRx0 + dRxm)))));
    if (c->p.wet_dry) r = r * GF(vmask_wet)[I2(i, j)];      // WET_DRY, prsgrd32.h:410-412
    rv[I3W(i, j, k)] = r;
  }
}

// prsgrd31_tile (ROMS/Nonlinear/prsgrd31.h:97-364): the standard density Jacobian -- the reference's default when no
// pressure-gradient option is defined (prsgrd.F:24-25) -- and, WJ = true, the weighted Jacobian of Song 1998
// (WJ_GRADP).  One thread per column marches down from the surface with the two running sums phix, phie in
// registers and the (k+1) values of rho, z_r of its three columns kept from the previous level: every field is
// read once per point it is needed at, nothing goes through scratch (5 field passes: rho, z_r, Hz in; ru, rv out).
// RHO_SURF is always on (globaldefs.h:130); the routine has no MASKING blocks.
template <bool WJ>
__global__ void __launch_bounds__(BLK_X *BLK_Y) k_prsgrd31(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x, j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const bool do_u = i >= b.IstrU, do_v = j >= b.JstrV;
  const gcd_t rho = (gcd_t)(c->F.rho), z_r = (gcd_t)(c->F.z_r), Hz = (gcd_t)(c->F.Hz);
  const gd_t ru = (gd_t)(c->F.ru + (long)(nrhs - 1) * n3w), rv = (gd_t)(c->F.rv + (long)(nrhs - 1) * n3w);
  const double g = c->p.g, rho0 = c->p.rho0;
  const double fac1 = 0.5 * g / rho0, fac2 = 1000.0 * g / rho0, fac3 = 0.25 * g / rho0;
  const long a = I2(i, j), aw = do_u ? a - 1 : a, as = do_v ? a - ni : a;   // neighbours only where they are used
  const double onu = GF(on_u)[a], omv = GF(om_v)[a];
  // WET_DRY: each stored term times the wet/dry mask of its face (prsgrd31.h:223, :269, :304, :350)
  const bool wet = c->p.wet_dry != 0;
  const double uw = wet ? (double)GF(umask_wet)[a] : 1.0, vw = wet ? (double)GF(vmask_wet)[a] : 1.0;
  // surface level, :196-226 and :276-306
  long q = a + (long)(N - 1) * nij, qw = aw + (long)(N - 1) * nij, qs = as + (long)(N - 1) * nij;
  double r1 = rho[q], z1 = z_r[q], r1w = rho[qw], z1w = z_r[qw], r1s = rho[qs], z1s = z_r[qs];
  const gcd_t zw = (gcd_t)(c->F.z_w);
  const double zw0 = zw[a + (long)N * nij], zww = zw[aw + (long)N * nij], zws = zw[as + (long)N * nij];
  double phix, phie;
  {
    const double cff1 = zw0 - z1 + zww - z1w;
    phix = fac1 * (r1 - r1w) * cff1;
    const bool atm = c->p.atm_press != 0;                       // ATM_PRESS, prsgrd31.h:213-215, :294-296
    const double fpa = 100.0 / rho0, pa0 = atm ? (double)GF(Pair)[a] : 0.0;
    if (atm) phix = phix + fpa * (pa0 - GF(Pair)[aw]);
    phix = phix + (fac2 + fac1 * (r1 + r1w)) * (zw0 - zww);
    const double cff1e = zw0 - z1 + zws - z1s;
    phie = fac1 * (r1 - r1s) * cff1e;
    if (atm) phie = phie + fpa * (pa0 - GF(Pair)[as]);
    phie = phie + (fac2 + fac1 * (r1 + r1s)) * (zw0 - zws);
    const double hz = Hz[q];
    if (do_u) { const double r = -0.5 * (hz + Hz[qw]) * phix * onu; ru[I3W(i, j, N)] = wet ? r * uw : r; }
    if (do_v) { const double r = -0.5 * (hz + Hz[qs]) * phie * omv; rv[I3W(i, j, N)] = wet ? r * vw : r; }
  }
  // interior: differentiate, then integrate downwards, :232-268 and :312-352
  auto jac = [&](double rk1, double rk1m, double rk, double rkm, double zk1, double zk1m, double zk, double zkm) {
    double cff1, cff2, cff3, cff4;
    if (WJ) {
      cff1 = 1.0 / ((zk1 - zk) * (zk1m - zkm));
      cff2 = zk - zkm + zk1 - zk1m;
      cff3 = zk1 - zk - zk1m + zkm;
      const double gamma = 0.125 * cff1 * cff2 * cff3;
      cff1 = (1.0 + gamma) * (rk1 - rk1m) + (1.0 - gamma) * (rk - rkm);
      cff2 = rk1 + rk1m - rk - rkm;
      cff3 = zk1 + zk1m - zk - zkm;
      cff4 = (1.0 + gamma) * (zk1 - zk1m) + (1.0 - gamma) * (zk - zkm);
    } else {
      cff1 = rk1 - rk1m + rk - rkm;
      cff2 = rk1 + rk1m - rk - rkm;
      cff3 = zk1 + zk1m - zk - zkm;
      cff4 = zk1 - zk1m + zk - zkm;
    }
    return fac3 * (cff1 * cff3 - cff2 * cff4);
  };
#pragma unroll 2
  for (int k = N - 1; k >= 1; k--) {
    q -= nij; qw -= nij; qs -= nij;
    const double r0 = rho[q], z0 = z_r[q], r0w = rho[qw], z0w = z_r[qw], r0s = rho[qs], z0s = z_r[qs];
    const double hz = Hz[q], hzw = Hz[qw], hzs = Hz[qs];
    phix = phix + jac(r1, r1w, r0, r0w, z1, z1w, z0, z0w);
    phie = phie + jac(r1, r1s, r0, r0s, z1, z1s, z0, z0s);
    if (do_u) { const double r = -0.5 * (hz + hzw) * phix * onu; ru[I3W(i, j, k)] = wet ? r * uw : r; }
    if (do_v) { const double r = -0.5 * (hz + hzs) * phie * omv; rv[I3W(i, j, k)] = wet ? r * vw : r; }
    r1 = r0; z1 = z0; r1w = r0w; z1w = z0w; r1s = r0s; z1s = z0s;
  }
}

// prsgrd40_tile (prsgrd40.h:170-268, PJ_GRADP): the hydrostatic pressure integral P of the own column and of the
// western / southern neighbour march downwards together in registers -- one launch, no scratch
__global__ void __launch_bounds__(BLK_X *BLK_Y) k_prsgrd40(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x, j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const bool do_u = i >= b.IstrU, do_v = j >= b.JstrV;
  const gcd_t rho = (gcd_t)(c->F.rho), zw = (gcd_t)(c->F.z_w), Hz = (gcd_t)(c->F.Hz);
  const gd_t ru = (gd_t)(c->F.ru + (long)(nrhs - 1) * n3w), rv = (gd_t)(c->F.rv + (long)(nrhs - 1) * n3w);
  const double cff = 0.5 * c->p.g, cff1 = c->p.g / c->p.rho0;
  const long a = I2(i, j), aw = do_u ? a - 1 : a, as = do_v ? a - ni : a;
  const double onu = GF(on_u)[a], omv = GF(om_v)[a];
  const double dzx = zw[aw + (long)N * nij] - zw[a + (long)N * nij];      // z_w(i-1,j,N) - z_w(i,j,N)
  const double dze = zw[as + (long)N * nij] - zw[a + (long)N * nij];
  double P0 = 0.0, Pw = 0.0, Ps = 0.0;        // P(.,.,k) of the three columns
  if (c->p.atm_press) {                       // ATM_PRESS, prsgrd40.h:187-196: P(N) = (100 / g) (Pair - 1 atm)
    const double fpa = 100.0 / c->p.g;
    P0 = P0 + fpa * (GF(Pair)[a] - 1013.25); Pw = Pw + fpa * (GF(Pair)[aw] - 1013.25); Ps = Ps + fpa * (GF(Pair)[as] - 1013.25);
  }
  double FCx = 0.0, FCe = 0.0;                // FC(i,k) of the two components
  for (int k = N; k >= 1; k--) {
    const long q = a + (long)(k - 1) * nij, qw = aw + (long)(k - 1) * nij, qs = as + (long)(k - 1) * nij;
    const double h0 = Hz[q], hw = Hz[qw], hs = Hz[qs];
    const double P0m = P0 + h0 * rho[q], Pwm = Pw + hw * rho[qw], Psm = Ps + hs * rho[qs];    // P(k-1)
    const double FX0 = 0.5 * h0 * (P0 + P0m);
    if (do_u) {
      const double FXw = 0.5 * hw * (Pw + Pwm);
      const double dh = zw[q] - zw[qw];                                   // z_w(i,j,k-1) - z_w(i-1,j,k-1)
      const double FCm = 0.5 * dh * (P0m + Pwm);
      ru[I3W(i, j, k)] = (cff * (hw + h0) * dzx + cff1 * (FXw - FX0 + FCx - FCm)) * onu;
      FCx = FCm;
    }
    if (do_v) {
      const double FXs = 0.5 * hs * (Ps + Psm);
      const double dh = zw[q] - zw[qs];
      const double FCm = 0.5 * dh * (P0m + Psm);
      rv[I3W(i, j, k)] = (cff * (hs + h0) * dze + cff1 * (FXs - FX0 + FCe - FCm)) * omv;
      FCe = FCm;
    }
    P0 = P0m; Pw = Pwm; Ps = Psm;
  }
}

}  // namespace

extern "C" int roms_hip_prsgrd(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_prsgrd");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  ScopedTimer tm("prsgrd");
  const roms_bounds_t &b = g_ctx.b;
  if (g_ctx.p.pgf == PGF_PJ_GRADP) {
    // the reference does not compile PJ_GRADP with WET_DRY (prsgrd40.h:98-100 passes umask_wet, vmask_wet to a
    // routine that does not declare them): refused
    if (g_ctx.p.wet_dry) return roms_fail("roms_hip_prsgrd", "PJ_GRADP with WET_DRY does not exist in the reference");
    hipLaunchKernelGGL(k_prsgrd40, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s->nrhs);
    KERNEL_CHECK("k_prsgrd40");
    return 0;
  }
  if (g_ctx.p.pgf != PGF_DJ_GRADPS) {                   // prsgrd31.h: one launch, no scratch
    if (g_ctx.p.pgf != PGF_STANDARD && g_ctx.p.pgf != PGF_WJ_GRADP)
      return roms_fail("roms_hip_prsgrd", "unknown pressure-gradient algorithm (enum roms_pgf)");
    const dim3 grid = grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1);
    if (g_ctx.p.pgf == PGF_WJ_GRADP)
      hipLaunchKernelGGL(k_prsgrd31<true>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs);
    else
      hipLaunchKernelGGL(k_prsgrd31<false>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs);
    KERNEL_CHECK("k_prsgrd31");
    return 0;
  }
  if (b.N < 3) return roms_fail("roms_hip_prsgrd", "N < 3");
  double *P = g_ctx.hostc.ws3[0];
  hipLaunchKernelGGL(k_prsgrd_P, grid2d(b.Iend - (b.IstrU - 1) + 1, b.Jend - (b.JstrV - 1) + 1), block2d(), 0,
                     g_ctx.stream, g_ctx.devc, P);
  KERNEL_CHECK("k_prsgrd_P");
  dim3 grid = grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1);
  grid.z = b.N;
  hipLaunchKernelGGL(k_prsgrd_uv, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, (const double *)P, s->nrhs);
  KERNEL_CHECK("k_prsgrd_uv");
  return 0;
}
