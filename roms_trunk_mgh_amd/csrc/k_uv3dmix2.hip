// k_uv3dmix2.hip -- harmonic horizontal viscosity along s-surfaces,
// uv3dmix2_s_tile (ROMS/Nonlinear/uv3dmix2_s.h:114-335); also accumulates
// rufrc, rvfrc.  Rotated to geopotentials (MIX_GEO_UV): k_uv3dmix2_geo below.  And the biharmonic one, uv3dmix4_s_tile (uv3dmix4_s.h:120-629): the first operator without the
// layer thickness (k_uv4_first -> LapU, LapV on a range one point wider), its rule outside physical edges and at the
// corners (k_uv4_edges, k_uv4_corners), then the harmonic kernel itself on LapU, LapV with visc4 and the sign reversed.
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

__device__ __forceinline__ RhoC rho_coef(const RomsDev *__restrict__ c, long r, long ni, const double *__restrict__ visc_r)
{
  const double *pm = c->F.pm, *pn = c->F.pn;
  RhoC o;
  o.pmon = c->F.pmon_r[r];
  o.pnom = c->F.pnom_r[r];
  o.e1 = pn[r] + pn[r + 1];
  o.e0 = pn[r - 1] + pn[r];
  o.n1 = pm[r] + pm[r + ni];
  o.n0 = pm[r - ni] + pm[r];
  o.k_x = c->F.on_r[r] * c->F.on_r[r] * visc_r[r];
  o.k_e = c->F.om_r[r] * c->F.om_r[r] * visc_r[r];
  return o;
}
__device__ __forceinline__ PsiC psi_coef(const RomsDev *__restrict__ c, long q, long ni, const double *__restrict__ visc_p)
{
  const double *pm = c->F.pm, *pn = c->F.pn;
  PsiC o;
  o.pmon = c->F.pmon_p[q];
  o.pnom = c->F.pnom_p[q];
  o.a = pn[q - ni] + pn[q];
  o.b = pn[q - 1 - ni] + pn[q - 1];
  o.c = pm[q - 1] + pm[q];
  o.d = pm[q - 1 - ni] + pm[q - ni];
  o.k_e = c->F.om_p[q] * c->F.om_p[q] * visc_p[q];
  o.k_x = c->F.on_p[q] * c->F.on_p[q] * visc_p[q];
  o.mask = c->p.masking ? pmaskw(c, q) : 1.0;               // (+ WET_DRY, uv3dmix2_s.h:275, uv3dmix4_s.h:334, :560)
  return o;
}
// (u, v, Hz through the global address space: generic pointers made these flat loads, which tie up the LDS / scalar
// counter as well and cannot be issued past it)
__device__ __forceinline__ double stress_r(const RhoC &m, gcd_t u, gcd_t v, gcd_t Hz, long rk, long ni)
{
  return Hz[rk] * 0.5 * (m.pmon * (m.e1 * u[rk + 1] - m.e0 * u[rk]) - m.pnom * (m.n1 * v[rk + ni] - m.n0 * v[rk]));
}
__device__ __forceinline__ double stress_p(const PsiC &m, gcd_t u, gcd_t v, gcd_t Hz, long qk, long ni, bool msk)
{
  const double cff = 0.125 * (Hz[qk - 1] + Hz[qk] + Hz[qk - 1 - ni] + Hz[qk - ni]) *
                     (m.pmon * (m.a * v[qk] - m.b * v[qk - 1]) + m.pnom * (m.c * u[qk] - m.d * u[qk - ni]));
  return msk ? cff * m.mask : cff;                        // MASKING, uv3dmix2_s.h:272
}

// BIH = false: uv3dmix2_s.  BIH = true: the second operator of uv3dmix4_s (uv3dmix4_s.h:522-620) -- the same
// expressions on LapU, LapV (module extents, N levels) with visc4, subtracted.
template <bool BIH>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_uv3dmix2_v2(const RomsDev *__restrict__ c, int nrhs, int nnew, const double *__restrict__ lapU,
              const double *__restrict__ lapV)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const bool do_u = i >= b.IstrU, do_v = j >= b.JstrV;
  const bool msk = c->p.masking != 0;
  const double dt = c->p.dt;
  const gcd_t u = (gcd_t)(BIH ? lapU : c->F.u + (long)(nrhs - 1) * n3r);
  const gcd_t v = (gcd_t)(BIH ? lapV : c->F.v + (long)(nrhs - 1) * n3r);
  const double *__restrict__ visc_r = BIH ? c->F.visc4_r : c->F.visc2_r;
  const double *__restrict__ visc_p = BIH ? c->F.visc4_p : c->F.visc2_p;
  const gcd_t Hz = (gcd_t)c->F.Hz;
  const gd_t un = (gd_t)(c->F.u + (long)(nnew - 1) * n3r);
  const gd_t vn = (gd_t)(c->F.v + (long)(nnew - 1) * n3r);
  const long a = I2(i, j);
  const double *pm = c->F.pm, *pn = c->F.pn;
  const double cu = dt * 0.25 * (pm[a - 1] + pm[a]) * (pn[a - 1] + pn[a]);
  const double cv = dt * 0.25 * (pm[a] + pm[a - ni]) * (pn[a] + pn[a - ni]);
  const double hn_u = 0.5 * (pn[a - 1] + pn[a]), hm_u = 0.5 * (pm[a - 1] + pm[a]);
  const double hn_v = 0.5 * (pn[a - ni] + pn[a]), hm_v = 0.5 * (pm[a - ni] + pm[a]);
  // the west / south stress points exist only where u / v is stepped: next to a closed wall
  // their stencil would reach below LBi / LBj (LBj = 0 on a closed southern edge)
  const RhoC r0 = rho_coef(c, a, ni, visc_r);
  const RhoC rw = do_u ? rho_coef(c, a - 1, ni, visc_r) : r0;
  const RhoC rs = do_v ? rho_coef(c, a - ni, ni, visc_r) : r0;
  const PsiC p0 = psi_coef(c, a, ni, visc_p), pN = psi_coef(c, a + ni, ni, visc_p), pE = psi_coef(c, a + 1, ni, visc_p);
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
      if constexpr (BIH) { ruf = ruf - cff1 - cff2; un[ak] = un[ak] - cff3; }
      else { ruf = ruf + cff1 + cff2; un[ak] = un[ak] + cff3; }
    }
    if (do_v) {
      const double srs = stress_r(rs, u, v, Hz, ak - ni, ni);
      const double spe = stress_p(pE, u, v, Hz, ak + 1, ni, msk);
      const double cff1 = hn_v * (pE.k_x * spe - p0.k_x * sp0);
      const double cff2 = hm_v * (r0.k_e * sr0 - rs.k_e * srs);
      const double cff3 = cv * (cff1 - cff2);
      if constexpr (BIH) { rvf = rvf - cff1 + cff2; vn[ak] = vn[ak] - cff3; }
      else { rvf = rvf + cff1 - cff2; vn[ak] = vn[ak] + cff3; }
    }
  }
  if (do_u) c->F.rufrc[a] = ruf;
  if (do_v) c->F.rvfrc[a] = rvf;
}

// ---- uv3dmix4_s: first operator (m s^-3/2), uv3dmix4_s.h:283-355 ----
struct Uv4 {
  double *lapU, *lapV;
  int iUa, iUb, jUa, jUb;        // IminU:ImaxU, JminU:JmaxU
  int iVa, iVb, jVa, jVb;        // IminV:ImaxV, JminV:JmaxV
  int cu[4], cv[4];              // the u / v condition on [LBS_WEST..LBS_NORTH] is "closed"
  double gamma2;
};

__device__ __forceinline__ double stress_r1(const RhoC &m, const double *__restrict__ u, const double *__restrict__ v, long rk, long ni)
{
  return 0.5 * (m.pmon * (m.e1 * u[rk + 1] - m.e0 * u[rk]) - m.pnom * (m.n1 * v[rk + ni] - m.n0 * v[rk]));
}
__device__ __forceinline__ double stress_p1(const PsiC &m, const double *__restrict__ u, const double *__restrict__ v, long qk, long ni, bool msk)
{
  const double cff = 0.5 * (m.pmon * (m.a * v[qk] - m.b * v[qk - 1]) + m.pnom * (m.c * u[qk] - m.d * u[qk - ni]));
  return msk ? cff * m.mask : cff;
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_uv4_first(const RomsDev *__restrict__ c, int nrhs, Uv4 A)
{
  DEV_PROLOGUE(c)
  const int ilo = A.iUa < A.iVa ? A.iUa : A.iVa, ihi = A.iUb > A.iVb ? A.iUb : A.iVb;
  const int jlo = A.jUa < A.jVa ? A.jUa : A.jVa, jhi = A.jUb > A.jVb ? A.jUb : A.jVb;
  const int i = ilo + blockIdx.x * BLK_X + threadIdx.x;
  const int j = jlo + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > ihi || j > jhi) return;
  const bool do_u = i >= A.iUa && i <= A.iUb && j >= A.jUa && j <= A.jUb;
  const bool do_v = i >= A.iVa && i <= A.iVb && j >= A.jVa && j <= A.jVb;
  if (!do_u && !do_v) return;
  const bool msk = c->p.masking != 0;
  const double *__restrict__ u = c->F.u + (long)(nrhs - 1) * n3r;
  const double *__restrict__ v = c->F.v + (long)(nrhs - 1) * n3r;
  const long a = I2(i, j);
  const double *pm = c->F.pm, *pn = c->F.pn;
  const RhoC r0 = rho_coef(c, a, ni, c->F.visc4_r);
  const RhoC rw = do_u ? rho_coef(c, a - 1, ni, c->F.visc4_r) : r0;
  const RhoC rs = do_v ? rho_coef(c, a - ni, ni, c->F.visc4_r) : r0;
  const PsiC p0 = psi_coef(c, a, ni, c->F.visc4_p);
  const PsiC pN = do_u ? psi_coef(c, a + ni, ni, c->F.visc4_p) : p0;
  const PsiC pE = do_v ? psi_coef(c, a + 1, ni, c->F.visc4_p) : p0;
  const double mu = pm[a - 1] + pm[a], nu = pn[a - 1] + pn[a];          // :340-347
  const double mv = pm[a] + pm[a - ni], nv = pn[a] + pn[a - ni];
  for (int k = 1; k <= N; k++) {
    const long ak = a + (long)(k - 1) * nij;
    const double sr0 = stress_r1(r0, u, v, ak, ni);
    const double sp0 = stress_p1(p0, u, v, ak, ni, msk);
    if (do_u) {
      const double srm = stress_r1(rw, u, v, ak - 1, ni);
      const double spn = stress_p1(pN, u, v, ak + ni, ni, msk);
      A.lapU[ak] = 0.125 * mu * nu * (nu * (r0.k_x * sr0 - rw.k_x * srm) + mu * (pN.k_e * spn - p0.k_e * sp0));
    }
    if (do_v) {
      const double srs = stress_r1(rs, u, v, ak - ni, ni);
      const double spe = stress_p1(pE, u, v, ak + 1, ni, msk);
      A.lapV[ak] = 0.125 * mv * nv * (nv * (pE.k_x * spe - p0.k_x * sp0) - mv * (r0.k_e * sr0 - rs.k_e * srs));
    }
  }
}

// the rule for LapU, LapV outside a physical edge, uv3dmix4_s.h:357-470: the normal component zero (closed) or a copy
// of the next one inside, the tangential one gamma2 times the first inside value (closed: the slipperiness) or zero.
// grid: x = positions along the edge, y = level, z = edge
__global__ void k_uv4_edges(const RomsDev *__restrict__ c, Uv4 A)      // gridDim.y = number of levels
{
  DEV_PROLOGUE(c)
  const int e = blockIdx.z, k = blockIdx.y + 1, q = blockIdx.x * blockDim.x + threadIdx.x;
  const long lev = (long)(k - 1) * nij;
  double *LU = A.lapU + lev, *LV = A.lapV + lev;
  if (e == LBS_WEST && b.west_edge && !b.EWperiodic) {
    const int ju = A.jUa + q, jv = A.jVa + q;
    if (ju <= A.jUb) LU[I2(b.Istr, ju)] = A.cu[e] ? 0.0 : LU[I2(b.Istr + 1, ju)];
    if (jv <= A.jVb) LV[I2(b.Istr - 1, jv)] = A.cv[e] ? A.gamma2 * LV[I2(b.Istr, jv)] : 0.0;
  } else if (e == LBS_EAST && b.east_edge && !b.EWperiodic) {
    const int ju = A.jUa + q, jv = A.jVa + q;
    if (ju <= A.jUb) LU[I2(b.Iend + 1, ju)] = A.cu[e] ? 0.0 : LU[I2(b.Iend, ju)];
    if (jv <= A.jVb) LV[I2(b.Iend + 1, jv)] = A.cv[e] ? A.gamma2 * LV[I2(b.Iend, jv)] : 0.0;
  } else if (e == LBS_SOUTH && b.south_edge && !b.NSperiodic) {
    const int iu = A.iUa + q, iv = A.iVa + q;
    if (iu <= A.iUb) LU[I2(iu, b.Jstr - 1)] = A.cu[e] ? A.gamma2 * LU[I2(iu, b.Jstr)] : 0.0;
    if (iv <= A.iVb) LV[I2(iv, b.Jstr)] = A.cv[e] ? 0.0 : LV[I2(iv, b.Jstr + 1)];
  } else if (e == LBS_NORTH && b.north_edge && !b.NSperiodic) {
    const int iu = A.iUa + q, iv = A.iVa + q;
    if (iu <= A.iUb) LU[I2(iu, b.Jend + 1)] = A.cu[e] ? A.gamma2 * LU[I2(iu, b.Jend)] : 0.0;
    if (iv <= A.iVb) LV[I2(iv, b.Jend + 1)] = A.cv[e] ? 0.0 : LV[I2(iv, b.Jend)];
  }
}

// corners, uv3dmix4_s.h:472-520 (after the edges); one thread per level
__global__ void k_uv4_corners(const RomsDev *__restrict__ c, Uv4 A, int nk)
{
  DEV_PROLOGUE(c)
  const int k = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (k > nk || b.EWperiodic || b.NSperiodic) return;
  const long lev = (long)(k - 1) * nij;
  double *LU = A.lapU + lev, *LV = A.lapV + lev;
  const int Is = b.Istr, Ie = b.Iend, Js = b.Jstr, Je = b.Jend;
  if (b.south_edge && b.west_edge) {
    LU[I2(Is, Js - 1)] = 0.5 * (LU[I2(Is + 1, Js - 1)] + LU[I2(Is, Js)]);
    LV[I2(Is - 1, Js)] = 0.5 * (LV[I2(Is - 1, Js + 1)] + LV[I2(Is, Js)]);
  }
  if (b.south_edge && b.east_edge) {
    LU[I2(Ie + 1, Js - 1)] = 0.5 * (LU[I2(Ie, Js - 1)] + LU[I2(Ie + 1, Js)]);
    LV[I2(Ie + 1, Js)] = 0.5 * (LV[I2(Ie, Js)] + LV[I2(Ie + 1, Js + 1)]);
  }
  if (b.north_edge && b.west_edge) {
    LU[I2(Is, Je + 1)] = 0.5 * (LU[I2(Is + 1, Je + 1)] + LU[I2(Is, Je)]);
    LV[I2(Is - 1, Je + 1)] = 0.5 * (LV[I2(Is, Je + 1)] + LV[I2(Is - 1, Je)]);
  }
  if (b.north_edge && b.east_edge) {
    LU[I2(Ie + 1, Je + 1)] = 0.5 * (LU[I2(Ie, Je + 1)] + LU[I2(Ie + 1, Je)]);
    LV[I2(Ie + 1, Je + 1)] = 0.5 * (LV[I2(Ie, Je + 1)] + LV[I2(Ie + 1, Je)]);
  }
}

// ---- the 2-D biharmonic viscosity of step2d, step2d_LF_AM3.h:1474-1740, as a pass of its own in front of the momentum
// kernel: the first operator on ubar, vbar(krhs) (no depth), the same edge / corner rule (with the conditions of ubar,
// vbar), the second operator with the total depth -> fac_u, fac_v, which k2d_mom_lds subtracts from rhs_ubar, rhs_vbar
// where the reference does.  The association of the products differs from the 3-D operator (visc4 multiplies first).
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_visc4_first(const RomsDev *__restrict__ c, int krhs, Uv4 A)
{
  DEV_PROLOGUE(c)
  const int ilo = A.iUa < A.iVa ? A.iUa : A.iVa, ihi = A.iUb > A.iVb ? A.iUb : A.iVb;
  const int jlo = A.jUa < A.jVa ? A.jUa : A.jVa, jhi = A.jUb > A.jVb ? A.jUb : A.jVb;
  const int i = ilo + blockIdx.x * BLK_X + threadIdx.x;
  const int j = jlo + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > ihi || j > jhi) return;
  const bool do_u = i >= A.iUa && i <= A.iUb && j >= A.jUa && j <= A.jUb;
  const bool do_v = i >= A.iVa && i <= A.iVb && j >= A.jVa && j <= A.jVb;
  if (!do_u && !do_v) return;
  const bool msk = c->p.masking != 0;
  const double *__restrict__ u = c->F.ubar + (long)(krhs - 1) * nij;
  const double *__restrict__ v = c->F.vbar + (long)(krhs - 1) * nij;
  const double *pm = c->F.pm, *pn = c->F.pn;
  const long a = I2(i, j);
  auto str_r = [&](long q) {       // UFx, VFe without their on_r^2 / om_r^2, :1474-1486
    return c->F.visc4_r[q] * 0.5 *
           (c->F.pmon_r[q] * ((pn[q] + pn[q + 1]) * u[q + 1] - (pn[q - 1] + pn[q]) * u[q]) -
            c->F.pnom_r[q] * ((pm[q] + pm[q + ni]) * v[q + ni] - (pm[q - ni] + pm[q]) * v[q]));
  };
  auto str_p = [&](long q) {       // :1487-1505
    const double cff = c->F.visc4_p[q] * 0.5 *
           (c->F.pmon_p[q] * ((pn[q - ni] + pn[q]) * v[q] - (pn[q - 1 - ni] + pn[q - 1]) * v[q - 1]) +
            c->F.pnom_p[q] * ((pm[q - 1] + pm[q]) * u[q] - (pm[q - 1 - ni] + pm[q - ni]) * u[q - ni]));
    return msk ? cff * pmaskw(c, q) : cff;                  // (+ WET_DRY, step2d_LF_AM3.h:1512, :1707)
  };
  const double sr0 = str_r(a), sp0 = str_p(a);
  if (do_u) {
    const double srw = str_r(a - 1), spn = str_p(a + ni);
    const double onr0 = c->F.on_r[a], onrw = c->F.on_r[a - 1], omp0 = c->F.om_p[a], ompn = c->F.om_p[a + ni];
    const double mu = pm[a - 1] + pm[a], nu = pn[a - 1] + pn[a];
    A.lapU[a] = 0.125 * mu * nu * (nu * (onr0 * onr0 * sr0 - onrw * onrw * srw) + mu * (ompn * ompn * spn - omp0 * omp0 * sp0));
  }
  if (do_v) {
    const double srs = str_r(a - ni), spe = str_p(a + 1);
    const double onp0 = c->F.on_p[a], onpe = c->F.on_p[a + 1], omr0 = c->F.om_r[a], omrs = c->F.om_r[a - ni];
    const double mv = pm[a] + pm[a - ni], nv = pn[a] + pn[a - ni];
    A.lapV[a] = 0.125 * mv * nv * (nv * (onpe * onpe * spe - onp0 * onp0 * sp0) - mv * (omr0 * omr0 * sr0 - omrs * omrs * srs));
  }
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_visc4_second(const RomsDev *__restrict__ c, int krhs, Uv4 A, double *__restrict__ facu, double *__restrict__ facv)
{
  DEV_PROLOGUE(c)
  const int i = b.Istr + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const bool do_u = i >= b.IstrU, do_v = j >= b.JstrV;
  const bool msk = c->p.masking != 0;
  const double *__restrict__ u = A.lapU, *__restrict__ v = A.lapV;
  const double *__restrict__ zeta = c->F.zeta + (long)(krhs - 1) * nij, *__restrict__ h = c->F.h;
  const double *pm = c->F.pm, *pn = c->F.pn;
  const long a = I2(i, j);
  auto D = [&](long q) { return zeta[q] + h[q]; };                       // Drhs, :700
  auto str_r = [&](long q) {       // :1660-1672
    return c->F.visc4_r[q] * D(q) * 0.5 *
           (c->F.pmon_r[q] * ((pn[q] + pn[q + 1]) * u[q + 1] - (pn[q - 1] + pn[q]) * u[q]) -
            c->F.pnom_r[q] * ((pm[q] + pm[q + ni]) * v[q + ni] - (pm[q - ni] + pm[q]) * v[q]));
  };
  auto str_p = [&](long q) {       // :1673-1692
    const double Dp = 0.25 * (D(q) + D(q - 1) + D(q - ni) + D(q - 1 - ni));
    const double cff = c->F.visc4_p[q] * Dp * 0.5 *
           (c->F.pmon_p[q] * ((pn[q - ni] + pn[q]) * v[q] - (pn[q - 1 - ni] + pn[q - 1]) * v[q - 1]) +
            c->F.pnom_p[q] * ((pm[q - 1] + pm[q]) * u[q] - (pm[q - 1 - ni] + pm[q - ni]) * u[q - ni]));
    return msk ? cff * pmaskw(c, q) : cff;                  // (+ WET_DRY, step2d_LF_AM3.h:1512, :1707)
  };
  const double sr0 = str_r(a), sp0 = str_p(a);
  if (do_u) {
    const double srw = str_r(a - 1), spn = str_p(a + ni);
    const double onr0 = c->F.on_r[a], onrw = c->F.on_r[a - 1], omp0 = c->F.om_p[a], ompn = c->F.om_p[a + ni];
    const double cff1 = 0.5 * (pn[a - 1] + pn[a]) * (onr0 * onr0 * sr0 - onrw * onrw * srw);
    const double cff2 = 0.5 * (pm[a - 1] + pm[a]) * (ompn * ompn * spn - omp0 * omp0 * sp0);
    facu[a] = cff1 + cff2;
  }
  if (do_v) {
    const double srs = str_r(a - ni), spe = str_p(a + 1);
    const double onp0 = c->F.on_p[a], onpe = c->F.on_p[a + 1], omr0 = c->F.om_r[a], omrs = c->F.om_r[a - ni];
    const double cff1 = 0.5 * (pn[a - ni] + pn[a]) * (onpe * onpe * spe - onp0 * onp0 * sp0);
    const double cff2 = 0.5 * (pm[a - ni] + pm[a]) * (omr0 * omr0 * sr0 - omrs * omrs * srs);
    facv[a] = cff1 - cff2;
  }
}

// ---- uv3dmix2_geo_tile (ROMS/Nonlinear/uv3dmix2_geo.h:116-756; UV_VIS2 with MIX_GEO_UV, roms_params_t.uv_vis2 = 2):
// harmonic viscosity rotated to geopotential surfaces.  The reference marches k over two-level slabs of fourteen
// private arrays; here a workgroup owns BLK_X x BLK_Y columns and keeps the slabs of its patch (own points plus one
// ring) in LDS, marching k upward: the slopes of z_r at u- and v-faces (their averages to rho- and psi-points, the
// reference's dZdx_r/_p, dZde_r/_p, are formed where they are used: the same 0.5*(a + b)), the four horizontal
// gradients, the two vertical ones, and the four horizontal fluxes of the level.  The vertical fluxes UFsx, UFse,
// VFsx, VFse are only ever read at the point that made them and wait in registers.  Every value is the reference's
// expression in the reference's order, evaluated on the reference's ranges for the patch taken as a tile, so the
// result does not depend on the partition (workgroups or tiles).  Two barriers per level.
__device__ __forceinline__ double gmin0(double a) { return a < 0.0 ? a : 0.0; }   // MIN(a, 0)
__device__ __forceinline__ double gmax0(double a) { return a > 0.0 ? a : 0.0; }   // MAX(a, 0)
#define GEO_W (BLK_X + 2)
#define GEO_H (BLK_Y + 2)
#define GEO_P (GEO_W * GEO_H)
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_uv3dmix2_geo(const RomsDev *__restrict__ c, int nrhs, int nnew)
{
  DEV_PROLOGUE(c)
  __shared__ double sUFx[GEO_P], sVFe[GEO_P], sUFe[GEO_P], sVFx[GEO_P];            // fluxes of the level
  __shared__ double sSx[2][GEO_P], sSe[2][GEO_P];                                  // slopes at u- / v-faces
  __shared__ double sdnUdx[2][GEO_P], sdmUde[2][GEO_P], sdnVdx[2][GEO_P], sdmVde[2][GEO_P];
  __shared__ double sdUdz[2][GEO_P], sdVdz[2][GEO_P];
  const Blk XB = xcd_block();
  const int i0 = b.Istr + XB.x * BLK_X, j0 = b.Jstr + XB.y * BLK_Y;
  // the patch as a tile
  const int Istr = i0, Iend = (i0 + BLK_X - 1 < b.Iend) ? i0 + BLK_X - 1 : b.Iend;
  const int Jstr = j0, Jend = (j0 + BLK_Y - 1 < b.Jend) ? j0 + BLK_Y - 1 : b.Jend;
  const int IstrU = i0 > b.IstrU ? i0 : b.IstrU, JstrV = j0 > b.JstrV ? j0 : b.JstrV;
  const int tid = threadIdx.y * BLK_X + threadIdx.x;
  const bool msk = c->p.masking != 0;
  const double dt = c->p.dt;
  const gcd_t u = (gcd_t)(c->F.u + (long)(nrhs - 1) * n3r), v = (gcd_t)(c->F.v + (long)(nrhs - 1) * n3r);
  const gd_t un = (gd_t)(c->F.u + (long)(nnew - 1) * n3r), vn = (gd_t)(c->F.v + (long)(nnew - 1) * n3r);
  const gcd_t z_r = (gcd_t)c->F.z_r, Hz = (gcd_t)c->F.Hz, pm = (gcd_t)c->F.pm, pn = (gcd_t)c->F.pn;
  const gcd_t on_r = (gcd_t)c->F.on_r, om_r = (gcd_t)c->F.om_r, on_p = (gcd_t)c->F.on_p, om_p = (gcd_t)c->F.om_p;
  const gcd_t visc2_r = (gcd_t)c->F.visc2_r, visc2_p = (gcd_t)c->F.visc2_p;
#define GL(i, j) (((j) - (j0 - 1)) * GEO_W + ((i) - (i0 - 1)))
#define dZdx_p(i, j, s) (0.5 * (sSx[s][GL(i, (j) - 1)] + sSx[s][GL(i, j)]))
#define dZde_p(i, j, s) (0.5 * (sSe[s][GL((i) - 1, j)] + sSe[s][GL(i, j)]))
#define dZdx_r(i, j, s) (0.5 * (sSx[s][GL(i, j)] + sSx[s][GL((i) + 1, j)]))
#define dZde_r(i, j, s) (0.5 * (sSe[s][GL(i, j)] + sSe[s][GL(i, (j) + 1)]))
#define dnUdx(i, j, s) sdnUdx[s][GL(i, j)]
#define dmUde(i, j, s) sdmUde[s][GL(i, j)]
#define dnVdx(i, j, s) sdnVdx[s][GL(i, j)]
#define dmVde(i, j, s) sdmVde[s][GL(i, j)]
#define dUdz(i, j, s) sdUdz[s][GL(i, j)]
#define dVdz(i, j, s) sdVdz[s][GL(i, j)]
#define UFx(i, j) sUFx[GL(i, j)]
#define VFe(i, j) sVFe[GL(i, j)]
#define UFe(i, j) sUFe[GL(i, j)]
#define VFx(i, j) sVFx[GL(i, j)]
  // the thread's own column
  const int io = i0 + (int)threadIdx.x, jo = j0 + (int)threadIdx.y;
  const bool own = io <= Iend && jo <= Jend;
  const bool do_u = own && io >= IstrU, do_v = own && jo >= JstrV;
  double UFsx1 = 0.0, UFse1 = 0.0, VFsx1 = 0.0, VFse1 = 0.0;        // the k1 level of the vertical fluxes
  double ruf = 0.0, rvf = 0.0;
  const long ao = own ? I2(io, jo) : 0;
  if (do_u) ruf = c->F.rufrc[ao];
  if (do_v) rvf = c->F.rvfrc[ao];
  // every thread looks after at most two points of the patch (q = tid, tid + 256); what those points need from the
  // k-independent fields for the gradients of a level -- the metric factors of the reference's expressions with their
  // masks, formed exactly as the reference forms them -- is kept in registers across the levels.  (The factors of the
  // flux phase as well: 282 registers, one workgroup per CU, 2.09 ms against 1.88 -- measured and left out.)
  constexpr int NPT = (GEO_P + BLK_X * BLK_Y - 1) / (BLK_X * BLK_Y);
  static_assert(NPT == 2, "patch points per thread");
  long pa[NPT];
  int flag[NPT];                                   // 1: u-face range, 2: v-face, 4: rho, 8: psi
  double cU[NPT], cV[NPT], cR1[NPT], cR2[NPT], pnE[NPT], pnW[NPT], pmN[NPT], pmS[NPT];
  double cP1[NPT], cP2[NPT], pmA[NPT], pmB[NPT], pnA[NPT], pnB[NPT];
#pragma unroll
  for (int r = 0; r < NPT; r++) {
    const int q = tid + r * BLK_X * BLK_Y;
    flag[r] = 0; pa[r] = 0;
    cU[r] = cV[r] = cR1[r] = cR2[r] = pnE[r] = pnW[r] = pmN[r] = pmS[r] = 0.0;
    cP1[r] = cP2[r] = pmA[r] = pmB[r] = pnA[r] = pnB[r] = 0.0;
    if (q < GEO_P) {
      const int i = i0 - 1 + q % GEO_W, j = j0 - 1 + q / GEO_W;
      const long a = I2(i, j);
      pa[r] = a;
      if (i >= IstrU - 1 && i <= Iend + 1 && j >= Jstr - 1 && j <= Jend + 1) flag[r] |= 1;
      if (i >= Istr - 1 && i <= Iend + 1 && j >= JstrV - 1 && j <= Jend + 1) flag[r] |= 2;
      if (i >= IstrU - 1 && i <= Iend && j >= JstrV - 1 && j <= Jend) flag[r] |= 4;
      if (i >= Istr && i <= Iend + 1 && j >= Jstr && j <= Jend + 1) flag[r] |= 8;
      if (flag[r] & 1) {
        double cff = 0.5 * (pm[a - 1] + pm[a]);
        if (msk) cff = cff * umaskw(c, a);
        cU[r] = cff;
      }
      if (flag[r] & 2) {
        double cff = 0.5 * (pn[a - ni] + pn[a]);
        if (msk) cff = cff * vmaskw(c, a);
        cV[r] = cff;
      }
      if (flag[r] & 4) {
        const double m = msk ? rmaskw(c, a) : 1.0;
        double cff = 0.5 * pm[a];
        if (msk) cff = cff * m;
        cR1[r] = cff;
        cff = 0.5 * pn[a];
        if (msk) cff = cff * m;
        cR2[r] = cff;
        pnE[r] = pn[a] + pn[a + 1]; pnW[r] = pn[a - 1] + pn[a];
        pmN[r] = pm[a] + pm[a + ni]; pmS[r] = pm[a - ni] + pm[a];
      }
      if (flag[r] & 8) {
        const double m = msk ? pmaskw(c, a) : 1.0;
        double cff = 0.125 * (pn[a - 1] + pn[a] + pn[a - 1 - ni] + pn[a - ni]);
        if (msk) cff = cff * m;
        cP1[r] = cff;
        cff = 0.125 * (pm[a - 1] + pm[a] + pm[a - 1 - ni] + pm[a - ni]);
        if (msk) cff = cff * m;
        cP2[r] = cff;
        pmA[r] = pm[a - 1] + pm[a]; pmB[r] = pm[a - 1 - ni] + pm[a - ni];
        pnA[r] = pn[a - ni] + pn[a]; pnB[r] = pn[a - 1 - ni] + pn[a - 1];
      }
    }
  }
  int k1, k2 = 1;
  for (int k = 0; k <= N; k++) {
    k1 = k2;
    k2 = 1 - k1;
    // ---- phase 1: everything of level k+1 that depends on the fields alone -> slot k2
#pragma unroll
    for (int r = 0; r < NPT; r++) {
      const int q = tid + r * BLK_X * BLK_Y;
      const long a = pa[r];
      const bool inU = flag[r] & 1, inV = flag[r] & 2, inR = flag[r] & 4, inP = flag[r] & 8;
      if (k < N) {
        const long ak = a + (long)k * nij;                                                    // level k+1
        if (inU) sSx[k2][q] = cU[r] * (z_r[ak] - z_r[ak - 1]);                                // :303-340
        if (inV) sSe[k2][q] = cV[r] * (z_r[ak] - z_r[ak - ni]);
        if (inR) {                                                                            // :345-412
          sdnUdx[k2][q] = cR1[r] * (pnE[r] * u[ak + 1] - pnW[r] * u[ak]);
          sdmVde[k2][q] = cR2[r] * (pmN[r] * v[ak + ni] - pmS[r] * v[ak]);
        }
        if (inP) {
          sdmUde[k2][q] = cP1[r] * (pmA[r] * u[ak] - pmB[r] * u[ak - ni]);
          sdnVdx[k2][q] = cP2[r] * (pnA[r] * v[ak] - pnB[r] * v[ak - 1]);
        }
      }
      if (k == 0 || k == N) {                                                                 // :414-440
        if (inU) sdUdz[k2][q] = 0.0;
        if (inV) sdVdz[k2][q] = 0.0;
      } else {
        const long ak = a + (long)k * nij, al = ak - nij;                                     // levels k+1, k
        if (inU) {
          const double cff = 1.0 / (0.5 * (z_r[ak - 1] - z_r[al - 1] + z_r[ak] - z_r[al]));
          sdUdz[k2][q] = cff * (u[ak] - u[al]);
        }
        if (inV) {
          const double cff = 1.0 / (0.5 * (z_r[ak - ni] - z_r[al - ni] + z_r[ak] - z_r[al]));
          sdVdz[k2][q] = cff * (v[ak] - v[al]);
        }
      }
    }
    __syncthreads();
    double UFsx2 = 0.0, UFse2 = 0.0, VFsx2 = 0.0, VFse2 = 0.0;       // k == N: zero (:433-440)
    if (k > 0) {
      // ---- phase 2: the rotated horizontal fluxes of level k (:463-541)
      for (int q = tid; q < GEO_P; q += BLK_X * BLK_Y) {
        const int i = i0 - 1 + q % GEO_W, j = j0 - 1 + q / GEO_W;
        const long a = I2(i, j), ak = a + (long)(k - 1) * nij;
        if (i >= IstrU - 1 && i <= Iend && j >= JstrV - 1 && j <= Jend) {
          const double zx = dZdx_r(i, j, k1), ze = dZde_r(i, j, k1);
          const double cff1 = gmin0(zx), cff2 = gmax0(zx), cff3 = gmin0(ze), cff4 = gmax0(ze);
          double cff = Hz[ak] *
                (on_r[a] * (dnUdx(i, j, k1) - 0.5 * pn[a] * (cff1 * (dUdz(i, j, k1) + dUdz(i + 1, j, k2)) + cff2 * (dUdz(i, j, k2) + dUdz(i + 1, j, k1)))) -
                 om_r[a] * (dmVde(i, j, k1) - 0.5 * pm[a] * (cff3 * (dVdz(i, j, k1) + dVdz(i, j + 1, k2)) + cff4 * (dVdz(i, j, k2) + dVdz(i, j + 1, k1)))));
          if (msk) cff = cff * rmaskw(c, a);
          sUFx[q] = on_r[a] * on_r[a] * visc2_r[a] * cff;
          sVFe[q] = om_r[a] * om_r[a] * visc2_r[a] * cff;
        }
        if (i >= Istr && i <= Iend + 1 && j >= Jstr && j <= Jend + 1) {
          const double pm_p = 0.25 * (pm[a - 1 - ni] + pm[a - 1] + pm[a - ni] + pm[a]);
          const double pn_p = 0.25 * (pn[a - 1 - ni] + pn[a - 1] + pn[a - ni] + pn[a]);
          const double zx = dZdx_p(i, j, k1), ze = dZde_p(i, j, k1);
          const double cff1 = gmin0(zx), cff2 = gmax0(zx), cff3 = gmin0(ze), cff4 = gmax0(ze);
          double cff = 0.25 * (Hz[ak - 1] + Hz[ak] + Hz[ak - 1 - ni] + Hz[ak - ni]) *
                (on_p[a] * (dnVdx(i, j, k1) - 0.5 * pn_p * (cff1 * (dVdz(i - 1, j, k1) + dVdz(i, j, k2)) + cff2 * (dVdz(i - 1, j, k2) + dVdz(i, j, k1)))) +
                 om_p[a] * (dmUde(i, j, k1) - 0.5 * pm_p * (cff3 * (dUdz(i, j - 1, k1) + dUdz(i, j, k2)) + cff4 * (dUdz(i, j - 1, k2) + dUdz(i, j, k1)))));
          if (msk) cff = cff * pmaskw(c, a);
          sUFe[q] = om_p[a] * om_p[a] * visc2_p[a] * cff;
          sVFx[q] = on_p[a] * on_p[a] * visc2_p[a] * cff;
        }
      }
      // ---- the vertical fluxes through the top of level k, at the thread's own faces (:546-700)
      if (k < N) {
        const int i = io, j = jo;
        if (do_u) {
          const long a = ao;
          double cff = 0.25 * (visc2_r[a - 1] + visc2_r[a]);
          const double fac1 = cff * c->F.on_u[a], fac2 = cff * c->F.om_u[a];
          cff = 0.5 * (pn[a - 1] + pn[a]);
          const double dnUdz = cff * dUdz(i, j, k2);
          const double dnVdz = cff * 0.25 * (dVdz(i - 1, j + 1, k2) + dVdz(i, j + 1, k2) + dVdz(i - 1, j, k2) + dVdz(i, j, k2));
          cff = 0.5 * (pm[a - 1] + pm[a]);
          const double dmUdz = cff * dUdz(i, j, k2);
          const double dmVdz = cff * 0.25 * (dVdz(i - 1, j + 1, k2) + dVdz(i, j + 1, k2) + dVdz(i - 1, j, k2) + dVdz(i, j, k2));
          const double xr1 = gmin0(dZdx_r(i - 1, j, k1)), xr2 = gmin0(dZdx_r(i, j, k2));
          const double xr3 = gmax0(dZdx_r(i - 1, j, k2)), xr4 = gmax0(dZdx_r(i, j, k1));
          const double ep1 = gmin0(dZde_p(i, j, k1)), ep2 = gmin0(dZde_p(i, j + 1, k2));
          const double ep3 = gmax0(dZde_p(i, j, k2)), ep4 = gmax0(dZde_p(i, j + 1, k1));
          const double xp5 = gmin0(dZdx_p(i, j, k1)), xp6 = gmin0(dZdx_p(i, j + 1, k2));
          const double xp7 = gmax0(dZdx_p(i, j, k2)), xp8 = gmax0(dZdx_p(i, j + 1, k1));
          const double er5 = gmin0(dZde_r(i - 1, j, k1)), er6 = gmin0(dZde_r(i, j, k2));
          const double er7 = gmax0(dZde_r(i - 1, j, k2)), er8 = gmax0(dZde_r(i, j, k1));
          UFsx2 = fac1 * (xr1 * (xr1 * dnUdz - dnUdx(i - 1, j, k1)) + xr2 * (xr2 * dnUdz - dnUdx(i, j, k2)) +
                          xr3 * (xr3 * dnUdz - dnUdx(i - 1, j, k2)) + xr4 * (xr4 * dnUdz - dnUdx(i, j, k1)));
          UFse2 = fac2 * (ep1 * (ep1 * dmUdz - dmUde(i, j, k1)) + ep2 * (ep2 * dmUdz - dmUde(i, j + 1, k2)) +
                          ep3 * (ep3 * dmUdz - dmUde(i, j, k2)) + ep4 * (ep4 * dmUdz - dmUde(i, j + 1, k1)));
          UFsx2 = UFsx2 +
                  fac1 * (ep1 * (xp5 * dnVdz - dnVdx(i, j, k1)) + ep2 * (xp6 * dnVdz - dnVdx(i, j + 1, k2)) +
                          ep3 * (xp7 * dnVdz - dnVdx(i, j, k2)) + ep4 * (xp8 * dnVdz - dnVdx(i, j + 1, k1)));
          UFse2 = UFse2 -
                  fac2 * (xr1 * (er5 * dmVdz - dmVde(i - 1, j, k1)) + xr2 * (er6 * dmVdz - dmVde(i, j, k2)) +
                          xr3 * (er7 * dmVdz - dmVde(i - 1, j, k2)) + xr4 * (er8 * dmVdz - dmVde(i, j, k1)));
        }
        if (do_v) {
          const long a = ao;
          double cff = 0.25 * (visc2_r[a - ni] + visc2_r[a]);
          const double fac1 = cff * c->F.on_v[a], fac2 = cff * c->F.om_v[a];
          cff = 0.5 * (pn[a - ni] + pn[a]);
          const double dnUdz = cff * 0.25 * (dUdz(i, j, k2) + dUdz(i + 1, j, k2) + dUdz(i, j - 1, k2) + dUdz(i + 1, j - 1, k2));
          const double dnVdz = cff * dVdz(i, j, k2);
          cff = 0.5 * (pm[a - ni] + pm[a]);
          const double dmUdz = cff * 0.25 * (dUdz(i, j, k2) + dUdz(i + 1, j, k2) + dUdz(i, j - 1, k2) + dUdz(i + 1, j - 1, k2));
          const double dmVdz = cff * dVdz(i, j, k2);
          const double xp1 = gmin0(dZdx_p(i, j, k1)), xp2 = gmin0(dZdx_p(i + 1, j, k2));
          const double xp3 = gmax0(dZdx_p(i, j, k2)), xp4 = gmax0(dZdx_p(i + 1, j, k1));
          const double er1 = gmin0(dZde_r(i, j - 1, k1)), er2 = gmin0(dZde_r(i, j, k2));
          const double er3 = gmax0(dZde_r(i, j - 1, k2)), er4 = gmax0(dZde_r(i, j, k1));
          const double xr5 = gmin0(dZdx_r(i, j - 1, k1)), xr6 = gmin0(dZdx_r(i, j, k2));
          const double xr7 = gmax0(dZdx_r(i, j - 1, k2)), xr8 = gmax0(dZdx_r(i, j, k1));
          const double ep5 = gmin0(dZde_p(i, j, k1)), ep6 = gmin0(dZde_p(i + 1, j, k2));
          const double ep7 = gmax0(dZde_p(i, j, k2)), ep8 = gmax0(dZde_p(i + 1, j, k1));
          VFsx2 = fac1 * (xp1 * (xp1 * dnVdz - dnVdx(i, j, k1)) + xp2 * (xp2 * dnVdz - dnVdx(i + 1, j, k2)) +
                          xp3 * (xp3 * dnVdz - dnVdx(i, j, k2)) + xp4 * (xp4 * dnVdz - dnVdx(i + 1, j, k1)));
          VFse2 = fac2 * (er1 * (er1 * dmVdz - dmVde(i, j - 1, k1)) + er2 * (er2 * dmVdz - dmVde(i, j, k2)) +
                          er3 * (er3 * dmVdz - dmVde(i, j - 1, k2)) + er4 * (er4 * dmVdz - dmVde(i, j, k1)));
          VFsx2 = VFsx2 -
                  fac1 * (er1 * (xr5 * dnUdz - dnUdx(i, j - 1, k1)) + er2 * (xr6 * dnUdz - dnUdx(i, j, k2)) +
                          er3 * (xr7 * dnUdz - dnUdx(i, j - 1, k2)) + er4 * (xr8 * dnUdz - dnUdx(i, j, k1)));
          VFse2 = VFse2 +
                  fac2 * (xp1 * (ep5 * dmUdz - dmUde(i, j, k1)) + xp2 * (ep6 * dmUdz - dmUde(i + 1, j, k2)) +
                          xp3 * (ep7 * dmUdz - dmUde(i, j, k2)) + xp4 * (ep8 * dmUdz - dmUde(i + 1, j, k1)));
        }
      }
    }
    __syncthreads();
    if (k > 0) {
      // ---- phase 3: the time step of level k; momentum is Hz*u, Hz*v here (:710-752)
      const int i = io, j = jo;
      const long ak = ao + (long)(k - 1) * nij;
      if (do_u) {
        const long a = ao;
        const double cff = dt * 0.25 * (pm[a - 1] + pm[a]) * (pn[a - 1] + pn[a]);
        const double cff1 = 0.5 * (pn[a - 1] + pn[a]) * (UFx(i, j) - UFx(i - 1, j));
        const double cff2 = 0.5 * (pm[a - 1] + pm[a]) * (UFe(i, j + 1) - UFe(i, j));
        const double cff3 = UFsx2 - UFsx1;
        const double cff4 = UFse2 - UFse1;
        const double cff5 = cff * (cff1 + cff2);
        const double cff6 = dt * (cff3 + cff4);
        ruf = ruf + cff1 + cff2 + cff3 + cff4;
        un[ak] = un[ak] + cff5 + cff6;
      }
      if (do_v) {
        const long a = ao;
        const double cff = dt * 0.25 * (pm[a] + pm[a - ni]) * (pn[a] + pn[a - ni]);
        const double cff1 = 0.5 * (pn[a - ni] + pn[a]) * (VFx(i + 1, j) - VFx(i, j));
        const double cff2 = 0.5 * (pm[a - ni] + pm[a]) * (VFe(i, j) - VFe(i, j - 1));
        const double cff3 = VFsx2 - VFsx1;
        const double cff4 = VFse2 - VFse1;
        const double cff5 = cff * (cff1 - cff2);
        const double cff6 = dt * (cff3 + cff4);
        rvf = rvf + cff1 - cff2 + cff3 + cff4;
        vn[ak] = vn[ak] + cff5 + cff6;
      }
    }
    UFsx1 = UFsx2; UFse1 = UFse2; VFsx1 = VFsx2; VFse1 = VFse2;
  }
  if (do_u) c->F.rufrc[ao] = ruf;
  if (do_v) c->F.rvfrc[ao] = rvf;
#undef GL
#undef dZdx_p
#undef dZde_p
#undef dZdx_r
#undef dZde_r
#undef dnUdx
#undef dmUde
#undef dnVdx
#undef dmVde
#undef dUdz
#undef dVdz
#undef UFx
#undef VFe
#undef UFe
#undef VFx
}

}  // namespace

// called by step2d_impl (k_step2d.hip) in front of the momentum kernel when UV_VIS4 is set
int roms_launch_step2d_visc4(int krhs)
{
  const roms_bounds_t &b = g_ctx.b;
  const roms_params_t &p = g_ctx.p;
  if (b.NghostPoints != 3) return roms_fail("roms_hip_step2d", "UV_VIS4 needs NghostPoints = 3 (inp_par.F:268-270)");
  Uv4 A;
  A.lapU = g_ctx.hostc.ws2[20];
  A.lapV = g_ctx.hostc.ws2[21];
  A.iUa = b.IstrUm1; A.iUb = b.Iendp1; A.jUa = b.Jstrm1; A.jUb = b.Jendp1;      // :1506-1529
  A.iVa = b.Istrm1; A.iVb = b.Iendp1; A.jVa = b.JstrVm1; A.jVb = b.Jendp1;
  for (int sd = 0; sd < 4; sd++) {
    A.cu[sd] = lbc_code(p, sd, LBV_UBAR) == LBC_CLOSED;
    A.cv[sd] = lbc_code(p, sd, LBV_VBAR) == LBC_CLOSED;
  }
  A.gamma2 = p.gamma2;
  const int ilo = A.iUa < A.iVa ? A.iUa : A.iVa, jlo = A.jUa < A.jVa ? A.jUa : A.jVa;
  const int nx = b.Iendp1 - ilo + 1, ny = b.Jendp1 - jlo + 1;
  hipLaunchKernelGGL(k2d_visc4_first, grid2d(nx, ny), block2d(), 0, g_ctx.stream, g_ctx.devc, krhs, A);
  KERNEL_CHECK("k2d_visc4_first");
  if (!b.EWperiodic || !b.NSperiodic) {
    const int len = nx > ny ? nx : ny;
    hipLaunchKernelGGL(k_uv4_edges, dim3((len + 63) / 64, 1, 4), dim3(64), 0, g_ctx.stream, g_ctx.devc, A);
    KERNEL_CHECK("k_uv4_edges");
  }
  if (!b.EWperiodic && !b.NSperiodic) {
    hipLaunchKernelGGL(k_uv4_corners, dim3(1), dim3(64), 0, g_ctx.stream, g_ctx.devc, A, 1);
    KERNEL_CHECK("k_uv4_corners");
  }
  hipLaunchKernelGGL(k2d_visc4_second, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, krhs, A, g_ctx.hostc.ws2[22], g_ctx.hostc.ws2[23]);
  KERNEL_CHECK("k2d_visc4_second");
  return 0;
}

int roms_entry_check(const char *name);

extern "C" int roms_hip_uv3dmix2(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_uv3dmix2");
  if (rc) return rc;
  ScopedTimer tm("uv3dmix2");
  const roms_bounds_t &b = g_ctx.b;
  if (g_ctx.p.uv_vis2 == 2) {                                     // MIX_GEO_UV
    hipLaunchKernelGGL(k_uv3dmix2_geo, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s->nrhs, s->nnew);
    KERNEL_CHECK("k_uv3dmix2_geo");
    return 0;
  }
  hipLaunchKernelGGL(k_uv3dmix2_v2<false>, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, s->nrhs, s->nnew, (const double *)nullptr, (const double *)nullptr);
  KERNEL_CHECK("k_uv3dmix2_v2");
  return 0;
}

extern "C" int roms_hip_uv3dmix4(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_uv3dmix4");
  if (rc) return rc;
  const roms_bounds_t &b = g_ctx.b;
  const roms_params_t &p = g_ctx.p;
  if (!p.uv_vis4) return roms_fail("roms_hip_uv3dmix4", "UV_VIS4 is not set (roms_params_t.uv_vis4)");
  if (b.NghostPoints != 3) return roms_fail("roms_hip_uv3dmix4", "UV_VIS4 needs NghostPoints = 3 (inp_par.F:268-270)");
  ScopedTimer tm("uv3dmix4");
  Uv4 A;
  A.lapU = g_ctx.hostc.ws3[1];
  A.lapV = g_ctx.hostc.ws3[2];
  auto mx = [](int x, int y) { return x > y ? x : y; };
  auto mn = [](int x, int y) { return x < y ? x : y; };
  if (b.EWperiodic) { A.iUa = b.Istr - 1; A.iUb = b.Iend + 1; A.iVa = b.Istr - 1; A.iVb = b.Iend + 1; }
  else { A.iUa = mx(2, b.IstrU - 1); A.iUb = mn(b.Iend + 1, b.Lm); A.iVa = mx(1, b.Istr - 1); A.iVb = mn(b.Iend + 1, b.Lm); }
  if (b.NSperiodic) { A.jUa = b.Jstr - 1; A.jUb = b.Jend + 1; A.jVa = b.Jstr - 1; A.jVb = b.Jend + 1; }
  else { A.jUa = mx(1, b.Jstr - 1); A.jUb = mn(b.Jend + 1, b.Mm); A.jVa = mx(2, b.JstrV - 1); A.jVb = mn(b.Jend + 1, b.Mm); }
  for (int sd = 0; sd < 4; sd++) {
    A.cu[sd] = lbc_code(p, sd, LBV_U) == LBC_CLOSED;
    A.cv[sd] = lbc_code(p, sd, LBV_V) == LBC_CLOSED;
  }
  A.gamma2 = p.gamma2;
  const int nx = mx(A.iUb, A.iVb) - mn(A.iUa, A.iVa) + 1, ny = mx(A.jUb, A.jVb) - mn(A.jUa, A.jVa) + 1;
  hipLaunchKernelGGL(k_uv4_first, grid2d(nx, ny), block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, A);
  KERNEL_CHECK("k_uv4_first");
  if (!b.EWperiodic || !b.NSperiodic) {
    const int len = mx(nx, ny);
    hipLaunchKernelGGL(k_uv4_edges, dim3((len + 63) / 64, b.N, 4), dim3(64), 0, g_ctx.stream, g_ctx.devc, A);
    KERNEL_CHECK("k_uv4_edges");
  }
  if (!b.EWperiodic && !b.NSperiodic) {
    hipLaunchKernelGGL(k_uv4_corners, dim3((b.N + 63) / 64), dim3(64), 0, g_ctx.stream, g_ctx.devc, A, b.N);
    KERNEL_CHECK("k_uv4_corners");
  }
  hipLaunchKernelGGL(k_uv3dmix2_v2<true>, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, s->nrhs, s->nnew, (const double *)A.lapU, (const double *)A.lapV);
  KERNEL_CHECK("k_uv3dmix2_v2<bih>");
  return 0;
}
