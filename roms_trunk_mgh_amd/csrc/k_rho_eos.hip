// k_rho_eos.hip -- equation of state, rho_eos_tile
// (ROMS/Nonlinear/rho_eos.F:111-575 nonlinear Jackett & McDougall 1995;
// :576-889 linear), with the VAR_RHO_2D vertical integrals rhoA/rhoS, the
// Brunt-Vaisala frequency bvf and the surface expansion coefficients.
//
// One thread per water column, swept top-down once: the level-(k+1)
// polynomial values needed by bvf(k) are carried in registers, so each 3-D
// input (t, s, z_r, z_w, Hz) is read once and each output (rho, pden, bvf)
// written once: 8 field passes, ~150 flops + 1 sqrt + 3 divides per cell.
#include "roms_dev.h"
#include "roms_eoscoef.h"

int roms_entry_check(const char *name);

namespace {

struct EosPt {
  double den1, bulk0, bulk1, bulk2, bulk, den;
  double Dden1DS, Dden1DT, DbulkDS, DbulkDT;
};

template <bool DERIV>
__device__ __forceinline__ EosPt eos_point(double tt, double ts, double Tp)
{
  EosPt o;
  const double Tt = (-2.0 > tt) ? -2.0 : tt;        // MAX(-2,t)
  const double Ts = (0.0 > ts) ? 0.0 : ts;          // MAX(0,s)
  const double sqrtTs = sqrt(Ts);
  const double Tpr10 = 0.1 * Tp;
  const double C0 = EOS_Q00 + Tt * (EOS_Q01 + Tt * (EOS_Q02 + Tt * (EOS_Q03 + Tt * (EOS_Q04 + Tt * EOS_Q05))));
  const double C1 = EOS_U00 + Tt * (EOS_U01 + Tt * (EOS_U02 + Tt * (EOS_U03 + Tt * EOS_U04)));
  const double C2 = EOS_V00 + Tt * (EOS_V01 + Tt * EOS_V02);
  o.den1 = C0 + Ts * (C1 + sqrtTs * C2 + Ts * EOS_W00);
  const double C3 = EOS_A00 + Tt * (EOS_A01 + Tt * (EOS_A02 + Tt * (EOS_A03 + Tt * EOS_A04)));
  const double C4 = EOS_B00 + Tt * (EOS_B01 + Tt * (EOS_B02 + Tt * EOS_B03));
  const double C5 = EOS_D00 + Tt * (EOS_D01 + Tt * EOS_D02);
  const double C6 = EOS_E00 + Tt * (EOS_E01 + Tt * (EOS_E02 + Tt * EOS_E03));
  const double C7 = EOS_F00 + Tt * (EOS_F01 + Tt * EOS_F02);
  const double C8 = EOS_G01 + Tt * (EOS_G02 + Tt * EOS_G03);
  const double C9 = EOS_H00 + Tt * (EOS_H01 + Tt * EOS_H02);
  o.bulk0 = C3 + Ts * (C4 + sqrtTs * C5);
  o.bulk1 = C6 + Ts * (C7 + sqrtTs * EOS_G00);
  o.bulk2 = C8 + Ts * C9;
  o.bulk = o.bulk0 - Tp * (o.bulk1 - Tp * o.bulk2);
  const double cff = 1.0 / (o.bulk + Tpr10);
  double den = o.den1 * o.bulk * cff;
  o.den = den - 1000.0;
  if constexpr (DERIV) {
    const double d0 = EOS_Q01 + Tt * (2.0 * EOS_Q02 + Tt * (3.0 * EOS_Q03 + Tt * (4.0 * EOS_Q04 + Tt * 5.0 * EOS_Q05)));
    const double d1 = EOS_U01 + Tt * (2.0 * EOS_U02 + Tt * (3.0 * EOS_U03 + Tt * 4.0 * EOS_U04));
    const double d2 = EOS_V01 + Tt * 2.0 * EOS_V02;
    o.Dden1DS = C1 + 1.5 * C2 * sqrtTs + 2.0 * EOS_W00 * Ts;
    o.Dden1DT = d0 + Ts * (d1 + sqrtTs * d2);
    const double d3 = EOS_A01 + Tt * (2.0 * EOS_A02 + Tt * (3.0 * EOS_A03 + Tt * 4.0 * EOS_A04));
    const double d4 = EOS_B01 + Tt * (2.0 * EOS_B02 + Tt * 3.0 * EOS_B03);
    const double d5 = EOS_D01 + Tt * 2.0 * EOS_D02;
    const double d6 = EOS_E01 + Tt * (2.0 * EOS_E02 + Tt * 3.0 * EOS_E03);
    const double d7 = EOS_F01 + Tt * 2.0 * EOS_F02;
    const double d8 = EOS_G02 + Tt * 2.0 * EOS_G03;
    const double d9 = EOS_H01 + Tt * 2.0 * EOS_H02;
    o.DbulkDS = C4 + sqrtTs * 1.5 * C5 - Tp * (C7 + sqrtTs * 1.5 * EOS_G00 - Tp * C9);
    o.DbulkDT = d3 + Ts * (d4 + sqrtTs * d5) - Tp * (d6 + Ts * d7 - Tp * (d8 + Ts * d9));
  }
  return o;
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_rho_eos_nonlin(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrT + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrT + XB.y * BLK_Y + threadIdx.y;
  if (i > b.IendT || j > b.JendT) return;
  const long c0 = I2(i, j);
  const gcd_t T = (gcd_t)(c->F.t + (long)(nrhs - 1) * n3r);                 // itemp = 1
  const gcd_t S = (gcd_t)(c->F.t + ((long)(nrhs - 1) + 3L) * n3r);          // isalt = 2
  const bool salt = c->p.salinity != 0 && b.NT >= 2;
  const gcd_t z_r = (gcd_t)(c->F.z_r);
  const gcd_t z_w = (gcd_t)(c->F.z_w);
  const gcd_t Hz = (gcd_t)(c->F.Hz);
  const double g = c->p.g;
  double rhoA = 0.0, rhoS = 0.0;
  EosPt up{};
  double zr_up = 0.0;
  const bool masking = c->p.masking != 0;
  const double rm = masking ? GF(rmask)[c0] : 1.0;
  for (int k = N; k >= 1; k--) {
    const long ck = c0 + (long)(k - 1) * nij;
    const double zr = z_r[ck], hz = Hz[ck];
    const double tt = T[ck], ts = salt ? S[ck] : 0.0;
    EosPt e;
    if (k == N) {
      e = eos_point<true>(tt, ts, zr);
      if (masking) e.den = e.den * rm;                      // MASKING, rho_eos.F:356
      // surface thermal expansion / saline contraction, rho_eos.F:480-520
      const double Tpr10 = 0.1 * zr;
      const double cff = e.bulk + Tpr10;
      const double cff1 = Tpr10 * e.den1;
      const double cff2 = e.bulk * cff;
      const double wrk = (e.den + 1000.0) * cff * cff;
      const double Tcof = -(e.DbulkDT * cff1 + e.Dden1DT * cff2);
      const double Scof = (e.DbulkDS * cff1 + e.Dden1DS * cff2);
      const double r = 1.0 / wrk;
      GF(alpha)[c0] = r * Tcof;
      GF(beta)[c0] = r * Scof;
      const double cf1 = e.den * hz;
      rhoS = 0.5 * cf1 * hz;
      rhoA = cf1;
      GF(bvf)[c0 + (long)N * nij] = 0.0;
    } else {
      e = eos_point<false>(tt, ts, zr);
      if (masking) e.den = e.den * rm;
      const double cf1 = e.den * hz;
      rhoS = rhoS + hz * (rhoA + 0.5 * cf1);
      rhoA = rhoA + cf1;
      // bvf(k) between levels k and k+1, rho_eos.F:440-470
      const double zw = z_w[c0 + (long)k * nij];
      const double bulk_up = up.bulk0 - zw * (up.bulk1 - up.bulk2 * zw);
      const double bulk_dn = e.bulk0 - zw * (e.bulk1 - e.bulk2 * zw);
      const double c1 = 1.0 / (bulk_up + 0.1 * zw);
      const double c2 = 1.0 / (bulk_dn + 0.1 * zw);
      const double den_up = c1 * (up.den1 * bulk_up);
      const double den_dn = c2 * (e.den1 * bulk_dn);
      GF(bvf)[c0 + (long)k * nij] = -g * (den_up - den_dn) / (0.5 * (den_up + den_dn) * (zr_up - zr));
    }
    GF(rho)[ck] = e.den;
    GF(pden)[ck] = masking ? (e.den1 - 1000.0) * rm : e.den1 - 1000.0;      // rho_eos.F:478
    up = e;
    zr_up = zr;
  }
  GF(bvf)[c0] = 0.0;
  const double cff2 = 1.0 / c->p.rho0;
  const double cff1 = 1.0 / (z_w[c0 + (long)N * nij] - z_w[c0]);
  GF(rhoA)[c0] = cff2 * cff1 * rhoA;
  GF(rhoS)[c0] = 2.0 * cff1 * cff1 * cff2 * rhoS;
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_rho_eos_lin(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrT + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrT + XB.y * BLK_Y + threadIdx.y;
  if (i > b.IendT || j > b.JendT) return;
  const long c0 = I2(i, j);
  const gcd_t T = (gcd_t)(c->F.t + (long)(nrhs - 1) * n3r);
  const gcd_t S = (gcd_t)(c->F.t + ((long)(nrhs - 1) + 3L) * n3r);
  const bool salt = c->p.salinity != 0 && b.NT >= 2;
  const double R0 = c->p.R0, T0 = c->p.T0, S0 = c->p.S0, Tcoef = c->p.Tcoef, Scoef = c->p.Scoef;
  double rhoA = 0.0, rhoS = 0.0;
  for (int k = N; k >= 1; k--) {
    const long ck = c0 + (long)(k - 1) * nij;
    double r = R0 - R0 * Tcoef * (T[ck] - T0);
    if (salt) r = r + R0 * Scoef * (S[ck] - S0);
    r = r - 1000.0;
    if (c->p.masking) r = r * GF(rmask)[c0];                // MASKING, rho_eos.F:717
    GF(rho)[ck] = r;
    GF(pden)[ck] = r;
    const double hz = GF(Hz)[ck];
    const double cf1 = r * hz;
    if (k == N) { rhoS = 0.5 * cf1 * hz; rhoA = cf1; }
    else { rhoS = rhoS + hz * (rhoA + 0.5 * cf1); rhoA = rhoA + cf1; }
  }
  const double cff2 = 1.0 / c->p.rho0;
  const double cff1 = 1.0 / (GF(z_w)[c0 + (long)N * nij] - GF(z_w)[c0]);
  GF(rhoA)[c0] = cff2 * cff1 * rhoA;
  GF(rhoS)[c0] = 2.0 * cff1 * cff1 * cff2 * rhoS;
}

}  // namespace

extern "C" int roms_hip_rho_eos(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_rho_eos");
  if (rc) return rc;
  ScopedTimer tm("rho_eos");
  const roms_bounds_t &b = g_ctx.b;
  dim3 grid = grid2d(b.IendT - b.IstrT + 1, b.JendT - b.JstrT + 1);
  if (g_ctx.p.nonlin_eos)
    hipLaunchKernelGGL(k_rho_eos_nonlin, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs);
  else
    hipLaunchKernelGGL(k_rho_eos_lin, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs);
  KERNEL_CHECK("k_rho_eos");
  halo_batch_begin();
  halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_rho]);
  halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_pden]);
  halo_exchange2d(GT_R, g_ctx.dev[FID_rhoA]);
  halo_exchange2d(GT_R, g_ctx.dev[FID_rhoS]);
  if (g_ctx.p.nonlin_eos) {
    halo_exchange2d(GT_R, g_ctx.dev[FID_alpha]);
    halo_exchange2d(GT_R, g_ctx.dev[FID_beta]);
    halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_bvf]);
  }
  return halo_batch_end();
}
