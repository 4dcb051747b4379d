// k_uv3dmix2.hip -- harmonic horizontal viscosity along s-surfaces,
// uv3dmix2_s_tile (ROMS/Nonlinear/uv3dmix2_s.h:114-335); also accumulates
// rufrc, rvfrc.
//
// One thread per (i,j) column sweeping k upward.  Each thread needs the stress
// at three rho-points (own, west, south) and three psi-points (own, north,
// east).  Their metric coefficients -- pmon/pnom and the four (pm+pm), (pn+pn)
// pairs per point -- do not depend on k, so they are formed once per column
// (36 doubles in VGPRs) instead of ~50 L2 requests per level.  Per level the
// kernel then reads u, v (7 points each), Hz (8 points) and read-modify-writes
// u,v(nnew): the 7 algorithmic field passes plus cache-served neighbours.
#include "roms_dev.h"

namespace {

struct RhoC { double pmon, pnom, e1, e0, n1, n0, k_x, k_e; };   // stress point at rho
struct PsiC { double pmon, pnom, a, b, c, d, k_e, k_x, mask; };  // stress point at psi (mask: pmask, 1 without MASKING)

__device__ __forceinline__ RhoC rho_coef(const RomsDev *__restrict__ c, long r, long ni)
{
  const double *pm = c->F.pm, *pn = c->F.pn;
  RhoC o;
  o.pmon = c->F.pmon_r[r];
  o.pnom = c->F.pnom_r[r];
  o.e1 = pn[r] + pn[r + 1];
  o.e0 = pn[r - 1] + pn[r];
  o.n1 = pm[r] + pm[r + ni];
  o.n0 = pm[r - ni] + pm[r];
  o.k_x = c->F.on_r[r] * c->F.on_r[r] * c->F.visc2_r[r];
  o.k_e = c->F.om_r[r] * c->F.om_r[r] * c->F.visc2_r[r];
  return o;
}
__device__ __forceinline__ PsiC psi_coef(const RomsDev *__restrict__ c, long q, long ni)
{
  const double *pm = c->F.pm, *pn = c->F.pn;
  PsiC o;
  o.pmon = c->F.pmon_p[q];
  o.pnom = c->F.pnom_p[q];
  o.a = pn[q - ni] + pn[q];
  o.b = pn[q - 1 - ni] + pn[q - 1];
  o.c = pm[q - 1] + pm[q];
  o.d = pm[q - 1 - ni] + pm[q - ni];
  o.k_e = c->F.om_p[q] * c->F.om_p[q] * c->F.visc2_p[q];
  o.k_x = c->F.on_p[q] * c->F.on_p[q] * c->F.visc2_p[q];
  o.mask = c->p.masking ? c->F.pmask[q] : 1.0;
  return o;
}
__device__ __forceinline__ double stress_r(const RhoC &m, const double *__restrict__ u, const double *__restrict__ v,
                                           const double *__restrict__ Hz, long rk, long ni)
{
  return Hz[rk] * 0.5 * (m.pmon * (m.e1 * u[rk + 1] - m.e0 * u[rk]) - m.pnom * (m.n1 * v[rk + ni] - m.n0 * v[rk]));
}
__device__ __forceinline__ double stress_p(const PsiC &m, const double *__restrict__ u, const double *__restrict__ v,
                                           const double *__restrict__ Hz, long qk, long ni, bool msk)
{
  const double cff = 0.125 * (Hz[qk - 1] + Hz[qk] + Hz[qk - 1 - ni] + Hz[qk - ni]) *
                     (m.pmon * (m.a * v[qk] - m.b * v[qk - 1]) + m.pnom * (m.c * u[qk] - m.d * u[qk - ni]));
  return msk ? cff * m.mask : cff;                        // MASKING, uv3dmix2_s.h:272
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_uv3dmix2_v2(const RomsDev *__restrict__ c, int nrhs, int nnew)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const bool do_u = i >= b.IstrU, do_v = j >= b.JstrV;
  const bool msk = c->p.masking != 0;
  const double dt = c->p.dt;
  const double *__restrict__ u = c->F.u + (long)(nrhs - 1) * n3r;
  const double *__restrict__ v = c->F.v + (long)(nrhs - 1) * n3r;
  const double *__restrict__ Hz = c->F.Hz;
  double *__restrict__ un = c->F.u + (long)(nnew - 1) * n3r;
  double *__restrict__ vn = c->F.v + (long)(nnew - 1) * n3r;
  const long a = I2(i, j);
  const double *pm = c->F.pm, *pn = c->F.pn;
  const double cu = dt * 0.25 * (pm[a - 1] + pm[a]) * (pn[a - 1] + pn[a]);
  const double cv = dt * 0.25 * (pm[a] + pm[a - ni]) * (pn[a] + pn[a - ni]);
  const double hn_u = 0.5 * (pn[a - 1] + pn[a]), hm_u = 0.5 * (pm[a - 1] + pm[a]);
  const double hn_v = 0.5 * (pn[a - ni] + pn[a]), hm_v = 0.5 * (pm[a - ni] + pm[a]);
  // the west / south stress points exist only where u / v is stepped: next to a closed wall
  // their stencil would reach below LBi / LBj (LBj = 0 on a closed southern edge)
  const RhoC r0 = rho_coef(c, a, ni);
  const RhoC rw = do_u ? rho_coef(c, a - 1, ni) : r0;
  const RhoC rs = do_v ? rho_coef(c, a - ni, ni) : r0;
  const PsiC p0 = psi_coef(c, a, ni), pN = psi_coef(c, a + ni, ni), pE = psi_coef(c, a + 1, ni);
  double ruf = do_u ? c->F.rufrc[a] : 0.0, rvf = do_v ? c->F.rvfrc[a] : 0.0;
  // levels are independent apart from the two running sums: two at a time, so that the loads of the second
  // are in flight while the first is computed
#pragma unroll 2
  for (int k = 1; k <= N; k++) {
    const long ak = a + (long)(k - 1) * nij;
    const double sr0 = stress_r(r0, u, v, Hz, ak, ni);
    const double sp0 = stress_p(p0, u, v, Hz, ak, ni, msk);
    if (do_u) {
      const double srm = stress_r(rw, u, v, Hz, ak - 1, ni);
      const double spn = stress_p(pN, u, v, Hz, ak + ni, ni, msk);
      const double cff1 = hn_u * (r0.k_x * sr0 - rw.k_x * srm);
      const double cff2 = hm_u * (pN.k_e * spn - p0.k_e * sp0);
      const double cff3 = cu * (cff1 + cff2);
      ruf = ruf + cff1 + cff2;
      un[ak] = un[ak] + cff3;
    }
    if (do_v) {
      const double srs = stress_r(rs, u, v, Hz, ak - ni, ni);
      const double spe = stress_p(pE, u, v, Hz, ak + 1, ni, msk);
      const double cff1 = hn_v * (pE.k_x * spe - p0.k_x * sp0);
      const double cff2 = hm_v * (r0.k_e * sr0 - rs.k_e * srs);
      const double cff3 = cv * (cff1 - cff2);
      rvf = rvf + cff1 - cff2;
      vn[ak] = vn[ak] + cff3;
    }
  }
  if (do_u) c->F.rufrc[a] = ruf;
  if (do_v) c->F.rvfrc[a] = rvf;
}

}  // namespace

int roms_entry_check(const char *name);

extern "C" int roms_hip_uv3dmix2(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_uv3dmix2");
  if (rc) return rc;
  ScopedTimer tm("uv3dmix2");
  const roms_bounds_t &b = g_ctx.b;
  hipLaunchKernelGGL(k_uv3dmix2_v2, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, s->nrhs, s->nnew);
  KERNEL_CHECK("k_uv3dmix2_v2");
  return 0;
}
