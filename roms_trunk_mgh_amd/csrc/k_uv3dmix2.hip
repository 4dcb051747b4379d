// k_uv3dmix2.hip -- harmonic horizontal viscosity along s-surfaces,
// uv3dmix2_s_tile (ROMS/Nonlinear/uv3dmix2_s.h:114-335); also accumulates
// rufrc, rvfrc.  And the biharmonic one, uv3dmix4_s_tile (uv3dmix4_s.h:120-629): the first operator without the
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
