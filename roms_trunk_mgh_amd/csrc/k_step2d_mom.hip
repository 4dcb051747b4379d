// k_step2d_mom.hip -- LDS-tiled momentum kernel of step2d_tile
// (ROMS/Nonlinear/step2d_LF_AM3.h:939-2255): pressure gradient with VAR_RHO_2D
// (:939-1019), 4th-order centred advection (:1079-1283), Coriolis (:1291-1325),
// curvilinear terms (:1333-1382), harmonic viscosity (:1394-1471), 2D<->3D coupling
// (:1884-2065) and the ubar/vbar step (:2098-2255).  The five fields the wide
// stencils read -- ubar(krhs),
// vbar(krhs), DUon, DVom and the total depth Drhs = zeta(krhs)+h -- are staged
// once per workgroup into LDS with their 2-point C-grid halo (68 x 8 doubles
// per field for a 64 x 4 workgroup, 21.8 KB).  The 4th-order advection,
// Coriolis, curvilinear and viscous terms then read LDS instead of issuing
// ~100 L2 requests per point.
//
// Source mapping (single tile, E-W periodic, closed N-S walls): threads cover
// i = LBi:UBi; ghost columns evaluate at their periodic image (the tile is
// loaded through the same wrap, so neighbour relations are preserved), and the
// wall-adjacent rows also store the u2dbc / v2dbc closed-wall values, so no
// boundary-condition or periodic-copy launch follows.
#include "roms_dev.h"

namespace {

struct S2 {
  int krhs, kstp, knew, nstp, nnew, iif, iic, ntfirst, predictor, sm;
};

// Row-uniform metrics.  On a zonally uniform grid (a Cartesian channel, a longitude-latitude grid: every
// configuration of BASELINE.json) the fifteen metric arrays below do not depend on i.  roms_rowm_prepare() checks
// that bit for bit on the device whenever one of them has been uploaded and, if it holds for all of them, leaves
// one value per row in a small table; the one-tile kernel then takes its metrics from the table (ROWM = true):
// all lanes of a wave read the same address, fifteen 2-D fields (65 MB of 183 MB per call on BENCHMARK3) no longer
// stream through the fabric and ~60 of the ~100 vector loads per point become broadcasts.  Same values, same
// arithmetic, same results; any i-dependence in any of the arrays selects the general kernel.
enum { RM_pm = 0, RM_pn, RM_on_u, RM_om_v, RM_fomn, RM_dndx, RM_dmde, RM_pmon_r, RM_pnom_r, RM_pmon_p, RM_pnom_p,
       RM_om_r, RM_on_r, RM_om_p, RM_on_p,
       // second group, examined on its own: the resting depth and the viscosity coefficients (flat or zonally uniform
       // bathymetry, uniform or grid-scaled viscosity) -- used from the table only when the first group is
       RM_h, RM_visc2_r, RM_visc2_p, RM_COUNT };
#define RM_FIRST_GROUP RM_h

// the row table is read-only for every kernel that uses it (k_rowm_build wrote it in an earlier launch): constant
// address space, so that a load with a wave-uniform row index is a scalar load (s_load_dwordx2 through the scalar
// cache) instead of a vector load that occupies the memory pipeline and a VGPR pair per value
typedef const double __attribute__((address_space(4))) *ccd_t;
template <bool ROWM>
struct Met {
  ccd_t tab;      // nj rows of RM_COUNT doubles
  int nj, LBj;
  // element q (flat index, row jr) of metric array A
  __device__ __forceinline__ double get(gcd_t A, int f, long q, int jr) const
  {
    if constexpr (ROWM) return tab[(jr - LBj) * RM_COUNT + f];   // [row][field]: one row's values lie together
    else return A[q];
  }
};
#define MT(name, q, jr) met.get((gcd_t)c->F.name, RM_##name, (q), (jr))
#define HT(name, q, jr) meth.get((gcd_t)c->F.name, RM_##name, (q), (jr))

#define TP (BLK_X + 4)       // tile pitch (i)
#define TJ (BLK_Y + 4)       // tile rows  (j)
#define C6 (1.0 / 6.0)

struct T2 {                  // LDS tiles, addressed with TARGET coordinates
  const double *ub, *vb, *DU, *DV, *D;
  int i0, j0;                // target coordinates of tile element (0,0)
  int Istr, Iend, Jstr, Jend;
  bool s_edge, n_edge, w_edge, e_edge;
  __device__ __forceinline__ int at(int i, int j) const { return (i - i0) + (j - j0) * TP; }
};

__device__ __forceinline__ double d2x(const double *f, int a) { return f[a - 1] - 2.0 * f[a] + f[a + 1]; }
__device__ __forceinline__ double d2y(const double *f, int a) { return f[a - TP] - 2.0 * f[a] + f[a + TP]; }

__device__ __forceinline__ double UFx2(const T2 &m, int i, int j)      // :1079-1125
{
  int ia = i, ib = i + 1;
  if (m.w_edge) { if (ia == m.Istr) ia = m.Istr + 1; if (ib == m.Istr) ib = m.Istr + 1; }
  if (m.e_edge) { if (ia == m.Iend + 1) ia = m.Iend; if (ib == m.Iend + 1) ib = m.Iend; }
  const int a = m.at(i, j);
  return 0.25 * (m.ub[a] + m.ub[a + 1] - C6 * (d2x(m.ub, m.at(ia, j)) + d2x(m.ub, m.at(ib, j)))) *
         (m.DU[a] + m.DU[a + 1] - C6 * (d2x(m.DU, m.at(ia, j)) + d2x(m.DU, m.at(ib, j))));
}
__device__ __forceinline__ double UFe2(const T2 &m, int i, int j)      // :1127-1165
{
  int ja = j, jb = j - 1;
  if (m.s_edge) { if (ja == m.Jstr - 1) ja = m.Jstr; if (jb == m.Jstr - 1) jb = m.Jstr; }
  if (m.n_edge) { if (ja == m.Jend + 1) ja = m.Jend; if (jb == m.Jend + 1) jb = m.Jend; }
  const int a = m.at(i, j);
  return 0.25 * (m.ub[a] + m.ub[a - TP] - C6 * (d2y(m.ub, m.at(i, ja)) + d2y(m.ub, m.at(i, jb)))) *
         (m.DV[a] + m.DV[a - 1] - C6 * (d2x(m.DV, a) + d2x(m.DV, a - 1)));
}
__device__ __forceinline__ double VFx2(const T2 &m, int i, int j)      // :1167-1205
{
  int ia = i, ib = i - 1;
  if (m.w_edge) { if (ia == m.Istr - 1) ia = m.Istr; if (ib == m.Istr - 1) ib = m.Istr; }
  if (m.e_edge) { if (ia == m.Iend + 1) ia = m.Iend; if (ib == m.Iend + 1) ib = m.Iend; }
  const int a = m.at(i, j);
  return 0.25 * (m.vb[a] + m.vb[a - 1] - C6 * (d2x(m.vb, m.at(ia, j)) + d2x(m.vb, m.at(ib, j)))) *
         (m.DU[a] + m.DU[a - TP] - C6 * (d2y(m.DU, a) + d2y(m.DU, a - TP)));
}
__device__ __forceinline__ double VFe2(const T2 &m, int i, int j)      // :1207-1256
{
  int ja = j, jb = j + 1;
  if (m.s_edge) { if (ja == m.Jstr) ja = m.Jstr + 1; if (jb == m.Jstr) jb = m.Jstr + 1; }
  if (m.n_edge) { if (ja == m.Jend + 1) ja = m.Jend; if (jb == m.Jend + 1) jb = m.Jend; }
  const int a = m.at(i, j);
  return 0.25 * (m.vb[a] + m.vb[a + TP] - C6 * (d2y(m.vb, m.at(i, ja)) + d2y(m.vb, m.at(i, jb)))) *
         (m.DV[a] + m.DV[a + TP] - C6 * (d2y(m.DV, m.at(i, ja)) + d2y(m.DV, m.at(i, jb))));
}

__device__ __forceinline__ int wrap_i(const roms_bounds_t &b, int i)
{
  return (i < 1) ? i + b.Lm : ((i > b.Lm) ? i - b.Lm : i);
}

// One free-surface point (step2d_LF_AM3.h:770-868) evaluated at source index a: the new free surface zn
// and the time-weighted zw the pressure gradient uses.  Same expressions as zeta_point (k_step2d.hip).
__device__ __forceinline__ void zeta_eval(const RomsDev *__restrict__ c, const S2 &s, const double rhs, long a, long nij,
                                          const double pmn_a, const double pn_a, double &zn, double &zw)
{
  const roms_params_t &p = c->p;
  const gcd_t zk = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
  const gcd_t zs = (gcd_t)(c->F.zeta + (long)(s.kstp - 1) * nij);
  const double dtfast = p.dtfast;
  if (s.iif == 1) {
    const double cff1 = dtfast;
    zn = zs[a] + pmn_a * pn_a * cff1 * rhs;
    if (p.masking) zn = zn * GF(rmask)[a];                      // MASKING, step2d_LF_AM3.h:778
    zw = 0.5 * (zs[a] + zn);
  } else if (s.predictor) {
    const double cff1 = 2.0 * dtfast;
    const double cff4 = 4.0 / 25.0;
    const double cff5 = 1.0 - 2.0 * cff4;
    zn = zs[a] + pmn_a * pn_a * cff1 * rhs;
    if (p.masking) zn = zn * GF(rmask)[a];                      // :804
    zw = cff5 * zk[a] + cff4 * (zs[a] + zn);
  } else {
    const int ptsk = 3 - s.kstp;
    const double cff1 = dtfast * 5.0 / 12.0;
    const double cff2 = dtfast * 8.0 / 12.0;
    const double cff3 = dtfast * 1.0 / 12.0;
    const double cff4 = 2.0 / 5.0;
    const double cff5 = 1.0 - cff4;
    const double cff = cff1 * rhs;
    zn = zs[a] + pmn_a * pn_a * (cff + cff2 * GF(rzeta)[a + (long)(s.kstp - 1) * nij] -
                                 cff3 * GF(rzeta)[a + (long)(ptsk - 1) * nij]);
    if (p.masking) zn = zn * GF(rmask)[a];                      // :835
    zw = cff5 * zn + cff4 * zk[a];
  }
}

// The same with the point's inputs already in registers (the fused kernel issues these loads before its first barrier):
// zs_a = zeta(kstp), zk_a = zeta(krhs), rz_k / rz_p = rzeta(kstp) / rzeta(ptsk), rm = rmask (1 without MASKING)
__device__ __forceinline__ void zeta_eval_pre(const roms_params_t &p, const S2 &s, const double rhs, const double pmn_a,
                                              const double pn_a, const double zs_a, const double zk_a, const double rz_k,
                                              const double rz_p, const double rm, const bool masking, double &zn,
                                              double &zw)
{
  const double dtfast = p.dtfast;
  if (s.iif == 1) {
    const double cff1 = dtfast;
    zn = zs_a + pmn_a * pn_a * cff1 * rhs;
    if (masking) zn = zn * rm;
    zw = 0.5 * (zs_a + zn);
  } else if (s.predictor) {
    const double cff1 = 2.0 * dtfast;
    const double cff4 = 4.0 / 25.0;
    const double cff5 = 1.0 - 2.0 * cff4;
    zn = zs_a + pmn_a * pn_a * cff1 * rhs;
    if (masking) zn = zn * rm;
    zw = cff5 * zk_a + cff4 * (zs_a + zn);
  } else {
    const double cff1 = dtfast * 5.0 / 12.0;
    const double cff2 = dtfast * 8.0 / 12.0;
    const double cff3 = dtfast * 1.0 / 12.0;
    const double cff4 = 2.0 / 5.0;
    const double cff5 = 1.0 - cff4;
    const double cff = cff1 * rhs;
    zn = zs_a + pmn_a * pn_a * (cff + cff2 * rz_k - cff3 * rz_p);
    if (masking) zn = zn * rm;
    zw = cff5 * zn + cff4 * zk_a;
  }
}

// FUSED (single tile, source-mapped calls only): the free-surface step and the fast-time averaging
// of k2d_zeta_sm are done here as well -- zeta_new and zwrk are evaluated from the staged DUon/DVom
// tiles for the (65 x 5) points this workgroup's momentum stencil touches and kept in LDS, so one
// step2d call is ONE launch and the zeta_new/zwrk scratch round trip disappears.
// WET (general path only): the WET_DRY blocks -- pmask_wet in the viscous stress (:1436-1438), the wet/dry factor of
// the new velocity, of the right-hand side and, in the first predictor, of rufrc / ru(:,:,0,nstp) (:2123-2135 ...).
template <bool FUSED, bool ROWM = false, bool ROWH = false, bool WET = false>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_mom_lds(const RomsDev *__restrict__ c, S2 s, const double *__restrict__ DUon, const double *__restrict__ DVom,
            const double *__restrict__ zeta_new, const double *__restrict__ zwrk, double *__restrict__ DUnext,
            double *__restrict__ DVnext)
{
  DEV_PROLOGUE(c)
  const roms_params_t &p = c->p;
  const Met<ROWM> met{(ccd_t)c->rowm, (int)nj, LBj};
  const Met<ROWH> meth{(ccd_t)c->rowm, (int)nj, LBj};
  // DUnext != nullptr (FUSED on several tiles, inside LOOP_2D): the closed-wall conditions are applied here
  // and DUon/DVom of the NEXT call (level knew) are left in DUnext/DVnext on the points this tile owns,
  // so that one exchange per call moves everything the next call needs
  const bool inline_bc = s.sm || DUnext != nullptr;
  __shared__ double sU[TJ * TP], sV[TJ * TP], sDU[TJ * TP], sDV[TJ * TP], sD[TJ * TP];
  __shared__ double sZn[FUSED ? TJ * TP : 1], sZw[FUSED ? TJ * TP : 1];
  // FUSED on one tile: threads cover the interior only and the owner of a column also stores its
  // periodic images (columns Lm+1.. and ..0), instead of ghost threads repeating the work of their
  // source column -- no nearly empty 33rd workgroup column, and the first predictor of a step (it
  // read-modify-writes rufrc and ru(:,:,0,nstp)) has no reader/writer race any more
  const bool img = FUSED && s.sm;
  const bool ghost_threads = s.sm && !img;
  const int ibase = ghost_threads ? b.LBi : b.Istr;
  const int ilast = ghost_threads ? (b.Lm + b.NghostPoints) : b.Iend;
  const Blk XB = xcd_block();
  const int it0 = ibase + XB.x * BLK_X, j0 = b.Jstr + XB.y * BLK_Y;
  // a wave is one row of the tile (BLK_X = 64 = the wave size): its row index is a scalar, so that the row-table
  // metrics of the momentum phase (MT / HT with row j, j-1, j+1) are scalar loads, not 64 lanes reading one address
  static_assert(BLK_X == 64, "one wave per tile row");
  const int it = it0 + threadIdx.x, j = j0 + __builtin_amdgcn_readfirstlane(threadIdx.y);
  const gcd_t ubk = (gcd_t)(c->F.ubar + (long)(s.krhs - 1) * nij);
  const gcd_t vbk = (gcd_t)(c->F.vbar + (long)(s.krhs - 1) * nij);
  const gcd_t zk = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
  // ---- FUSED: every global value the later phases need is requested now, so that the kernel waits for memory once
  // instead of once per phase (staging, free surface, momentum): the free-surface inputs of this thread's (up to two)
  // tile points and the momentum inputs of its own point ----
  constexpr int ZW = BLK_X + 1, ZH = BLK_Y + 1;
  constexpr int ZIT = (ZW * ZH + BLK_X * BLK_Y - 1) / (BLK_X * BLK_Y);
  double z_zs[ZIT], z_zk[ZIT], z_rk[ZIT], z_rp[ZIT], z_rm[ZIT];
  double q_rhoA0 = 0.0, q_rhoS0 = 0.0, q_rhoAw = 0.0, q_rhoSw = 0.0, q_rhoAs = 0.0, q_rhoSs = 0.0;
  double q_zs0 = 0.0, q_zsw = 0.0, q_zss = 0.0, q_ruf = 0.0, q_rvf = 0.0, q_us = 0.0, q_vs = 0.0;
  double q_rubk = 0.0, q_rubp = 0.0, q_rvbk = 0.0, q_rvbp = 0.0;
  double q_Zt = 0.0, q_DU1 = 0.0, q_DU2 = 0.0, q_DV1 = 0.0, q_DV2 = 0.0, q_zkr = 0.0;
  if constexpr (FUSED) {
    const int tid = threadIdx.y * BLK_X + threadIdx.x;
    const gcd_t zsq = (gcd_t)(c->F.zeta + (long)(s.kstp - 1) * nij);
    const bool am3z = !(s.iif == 1 || s.predictor);
#pragma unroll
    for (int r = 0; r < ZIT; r++) {
      const int q = tid + r * BLK_X * BLK_Y;
      z_zs[r] = 0.0; z_zk[r] = 0.0; z_rk[r] = 0.0; z_rp[r] = 0.0; z_rm[r] = 1.0;
      if (q < ZW * ZH) {
        const int li = 1 + q % ZW, lj = 1 + q / ZW;
        int gi = it0 - 2 + li, gj = j0 - 2 + lj;
        if (s.sm) gi = wrap_i(b, gi);
        else gi = gi < b.LBi ? b.LBi : (gi > b.UBi ? b.UBi : gi);
        gj = gj < b.LBj ? b.LBj : (gj > b.UBj ? b.UBj : gj);
        const long gq = I2(gi, gj);
        z_zs[r] = zsq[gq];
        z_zk[r] = zk[gq];
        if (am3z) {
          z_rk[r] = GF(rzeta)[gq + (long)(s.kstp - 1) * nij];
          z_rp[r] = GF(rzeta)[gq + (long)(3 - s.kstp - 1) * nij];
        }
        if (p.masking) z_rm[r] = GF(rmask)[gq];
      }
    }
    if (it <= ilast && j <= b.Jend) {
      const int isrc = ghost_threads ? wrap_i(b, it) : it;
      const long a = I2(isrc, j), o = I2(it, j);
      const gcd_t rhoA = (gcd_t)(c->F.rhoA), rhoS = (gcd_t)(c->F.rhoS);
      q_rhoA0 = rhoA[a]; q_rhoS0 = rhoS[a]; q_rhoAw = rhoA[a - 1]; q_rhoSw = rhoS[a - 1];
      q_rhoAs = rhoA[a - ni]; q_rhoSs = rhoS[a - ni];
      q_zs0 = zsq[a]; q_zsw = zsq[a - 1]; q_zss = zsq[a - ni];
      q_ruf = GF(rufrc)[a]; q_rvf = GF(rvfrc)[a];
      q_us = GF(ubar)[a + (long)(s.kstp - 1) * nij]; q_vs = GF(vbar)[a + (long)(s.kstp - 1) * nij];
      if (am3z) {
        q_rubk = GF(rubar)[a + (long)(s.kstp - 1) * nij]; q_rubp = GF(rubar)[a + (long)(3 - s.kstp - 1) * nij];
        q_rvbk = GF(rvbar)[a + (long)(s.kstp - 1) * nij]; q_rvbp = GF(rvbar)[a + (long)(3 - s.kstp - 1) * nij];
      }
      if (!(s.predictor && s.iif == 1)) {              // the running sums of the fast-time averages (:614-682)
        q_DU2 = GF(DU_avg2)[o]; q_DV2 = GF(DV_avg2)[o];
        if (s.predictor) { q_Zt = GF(Zt_avg1)[o]; q_DU1 = GF(DU_avg1)[o]; q_DV1 = GF(DV_avg1)[o]; q_zkr = zk[o]; }
      }
    }
  }
  // ---- stage the stencil fields (target coordinates it0-2.., j0-2..) ----
  {
    const int tid = threadIdx.y * BLK_X + threadIdx.x;
    for (int e = tid; e < TJ * TP; e += BLK_X * BLK_Y) {
      const int li = e % TP, lj = e / TP;
      int gi = it0 - 2 + li, gj = j0 - 2 + lj;
      if (s.sm) gi = wrap_i(b, gi);
      else gi = gi < b.LBi ? b.LBi : (gi > b.UBi ? b.UBi : gi);
      gj = gj < b.LBj ? b.LBj : (gj > b.UBj ? b.UBj : gj);
      const long g = I2(gi, gj);
      const double ug = ubk[g], vg = vbk[g], Dg = zk[g] + HT(h, g, gj);
      sU[e] = ug;
      sV[e] = vg;
      sD[e] = Dg;
      if (DUon) {
        sDU[e] = DUon[g];
        sDV[e] = DVom[g];
      } else {
        // DUon, DVom evaluated in place (:509-544), identical expression to k2d_flux; the
        // ghost columns/rows they reach hold exact copies, so no separate flux pass is needed
        const double cu = 0.5 * MT(on_u, g, gj);
        sDU[e] = ug * (cu * (Dg + (zk[g - 1] + HT(h, g - 1, gj))));
        if (gj >= b.LBj + 1) {
          const double cv = 0.5 * MT(om_v, g, gj);
          sDV[e] = vg * (cv * (Dg + (zk[g - ni] + HT(h, g - ni, gj - 1))));
        } else sDV[e] = 0.0;
      }
    }
  }
  __syncthreads();
  if constexpr (FUSED) {
    // free surface at tile points li = 1..BLK_X+1, lj = 1..BLK_Y+1 (targets it0-1.., j0-1..)
    const int tid = threadIdx.y * BLK_X + threadIdx.x;
#pragma unroll
    for (int r = 0; r < ZIT; r++) {
      const int q = tid + r * BLK_X * BLK_Y;
      if (q < ZW * ZH) {
        const int li = 1 + q % ZW, lj = 1 + q / ZW;
        const int e = lj * TP + li;
        int gi = it0 - 2 + li, gj = j0 - 2 + lj;
        if (s.sm) gi = wrap_i(b, gi);
        else gi = gi < b.LBi ? b.LBi : (gi > b.UBi ? b.UBi : gi);
        gj = gj < b.LBj ? b.LBj : (gj > b.UBj ? b.UBj : gj);
        const double rhs = (sDU[e] - sDU[e + 1]) + (sDV[e] - sDV[e + TP]);
        double zn, zw;
        const long gq = I2(gi, gj);
        zeta_eval_pre(p, s, rhs, MT(pm, gq, gj), MT(pn, gq, gj), z_zs[r], z_zk[r], z_rk[r], z_rp[r], z_rm[r],
                      p.masking != 0, zn, zw);
        sZn[e] = zn;
        sZw[e] = zw;
      }
    }
    __syncthreads();
  }
  if (it > ilast || j > b.Jend) return;
  const int i = ghost_threads ? wrap_i(b, it) : it; // source column
  // store at the target and, for an owner next to the periodic seam, at its image column(s)
  auto put = [&](gd_t A, long idx, double val) {
    A[idx] = val;
    if (img) {
      if (i <= b.NghostPoints) A[idx + b.Lm] = val;
      if (i >= b.Lm - 2) A[idx - b.Lm] = val;
    }
  };
  const bool owner = (i == it);
  const bool do_u = s.sm ? true : (i >= b.IstrU);
  const bool do_v = j >= b.JstrV;
  const long a = I2(i, j);                          // source index in global arrays
  const long o = I2(it, j);                         // target index
  const gcd_t rhoA = (gcd_t)(c->F.rhoA);
  const gcd_t rhoS = (gcd_t)(c->F.rhoS);
  const gcd_t zs = (gcd_t)(c->F.zeta + (long)(s.kstp - 1) * nij);
  T2 m;
  m.ub = sU; m.vb = sV; m.DU = sDU; m.DV = sDV; m.D = sD;
  m.i0 = it0 - 2; m.j0 = j0 - 2;
  m.Istr = b.Istr; m.Iend = b.Iend; m.Jstr = b.Jstr; m.Jend = b.Jend;
  m.s_edge = b.south_edge && !b.NSperiodic; m.n_edge = b.north_edge && !b.NSperiodic;
  m.w_edge = b.west_edge && !b.EWperiodic;  m.e_edge = b.east_edge && !b.EWperiodic;
  const int t = m.at(it, j);                        // this point in the tiles
  const double fac = 1000.0 / p.rho0;
  const bool masking = p.masking != 0;
  // closed-wall rows under MASKING: the boundary value times the mask of the boundary point (zetabc.F:540,
  // u2dbc_im.F:975); mk(M, q) = 1 without masks
  auto mk = [&](gd_t M, long q) { return masking ? (double)M[q] : 1.0; };
  // ---- pressure gradient, :939-1019 ----
  if constexpr (FUSED) {
    // ---- what k2d_zeta_sm did: fast-time averages on the owned ranges and zeta(knew), rzeta(krhs) ----
    const int iif = s.iif;
    const gcd_t zkr = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
    auto average = [&](long oo, int tt, bool inU, bool inV) {       // step2d_LF_AM3.h:614-682 at target oo
      if (s.predictor && iif == 1) {
        const double cff2 = (-1.0 / 12.0) * p.weight2[iif];
        GF(Zt_avg1)[oo] = 0.0;
        if (inU) { GF(DU_avg1)[oo] = 0.0; GF(DU_avg2)[oo] = cff2 * sDU[tt]; }
        if (inV) { GF(DV_avg1)[oo] = 0.0; GF(DV_avg2)[oo] = cff2 * sDV[tt]; }
      } else if (s.predictor) {
        const double cff1 = p.weight1[iif - 2];
        const double cff2 = (8.0 / 12.0) * p.weight2[iif - 1] - (1.0 / 12.0) * p.weight2[iif];
        GF(Zt_avg1)[oo] = GF(Zt_avg1)[oo] + cff1 * zkr[oo];
        if (inU) {
          GF(DU_avg1)[oo] = GF(DU_avg1)[oo] + cff1 * sDU[tt];
          GF(DU_avg2)[oo] = GF(DU_avg2)[oo] + cff2 * sDU[tt];
        }
        if (inV) {
          GF(DV_avg1)[oo] = GF(DV_avg1)[oo] + cff1 * sDV[tt];
          GF(DV_avg2)[oo] = GF(DV_avg2)[oo] + cff2 * sDV[tt];
        }
      } else {
        const double cff2 = (iif == 1) ? p.weight2[iif - 1] : (5.0 / 12.0) * p.weight2[iif - 1];
        if (inU) GF(DU_avg2)[oo] = GF(DU_avg2)[oo] + cff2 * sDU[tt];
        if (inV) GF(DV_avg2)[oo] = GF(DV_avg2)[oo] + cff2 * sDV[tt];
      }
    };
    const bool in_i = it >= b.IstrR && it <= b.IendR;
    if (in_i) {
      const bool inU = it >= b.Istr;
      // the thread's own point, with the running sums requested at the top of the kernel (same expressions as `average`)
      if (s.predictor && iif == 1) {
        const double cff2 = (-1.0 / 12.0) * p.weight2[iif];
        GF(Zt_avg1)[o] = 0.0;
        if (inU) { GF(DU_avg1)[o] = 0.0; GF(DU_avg2)[o] = cff2 * sDU[t]; }
        GF(DV_avg1)[o] = 0.0; GF(DV_avg2)[o] = cff2 * sDV[t];
      } else if (s.predictor) {
        const double cff1 = p.weight1[iif - 2];
        const double cff2 = (8.0 / 12.0) * p.weight2[iif - 1] - (1.0 / 12.0) * p.weight2[iif];
        GF(Zt_avg1)[o] = q_Zt + cff1 * q_zkr;
        if (inU) {
          GF(DU_avg1)[o] = q_DU1 + cff1 * sDU[t];
          GF(DU_avg2)[o] = q_DU2 + cff2 * sDU[t];
        }
        GF(DV_avg1)[o] = q_DV1 + cff1 * sDV[t];
        GF(DV_avg2)[o] = q_DV2 + cff2 * sDV[t];
      } else {
        const double cff2 = (iif == 1) ? p.weight2[iif - 1] : (5.0 / 12.0) * p.weight2[iif - 1];
        if (inU) GF(DU_avg2)[o] = q_DU2 + cff2 * sDU[t];
        GF(DV_avg2)[o] = q_DV2 + cff2 * sDV[t];
      }
      if (j == b.Jstr && b.JstrR < b.Jstr) average(o - ni, t - TP, inU, false);   // row JstrR = Jstr-1
      if (j == b.Jend && b.JendR > b.Jend) average(o + ni, t + TP, inU, true);    // row JendR = Jend+1
    }
    const double zn = sZn[t];
    const gd_t zout = (gd_t)(c->F.zeta + (long)(s.knew - 1) * nij);
    put(zout, o, zn);
    if (b.south_edge && j == b.Jstr) put(zout, o - ni, masking ? zn * GF(rmask)[a - ni] : zn);   // zetabc closed
    if (b.north_edge && j == b.Jend) put(zout, o + ni, masking ? zn * GF(rmask)[a + ni] : zn);
    if (s.predictor)      // at the target: ghost columns hold the periodic copy, as after the exchange
      put((gd_t)(c->F.rzeta + (long)(s.krhs - 1) * nij), o, (sDU[t] - sDU[t + 1]) + (sDV[t] - sDV[t + TP]));
  }
  // resting depth at the point, its western and its southern neighbour (rows j, j, j-1: always inside the array)
  const double h0 = HT(h, a, j), hw = HT(h, a - 1, j), hs = HT(h, a - ni, j - 1);
  const double zw0 = FUSED ? sZw[t] : zwrk[a];
  const double rS0 = FUSED ? q_rhoS0 : (double)rhoS[a], rA0 = FUSED ? q_rhoA0 : (double)rhoA[a];
  const double gz0 = (fac + rS0) * zw0, gz20 = gz0 * zw0, gsa0 = zw0 * (rS0 - rA0);
  const double cg = 0.5 * p.g, c3 = 1.0 / 3.0;
  double rhs_u = 0.0, rhs_v = 0.0;
  if (do_u) {
    const long q = a - 1;
    const double zw = FUSED ? sZw[t - 1] : zwrk[q];
    const double rSq = FUSED ? q_rhoSw : (double)rhoS[q], rAq = FUSED ? q_rhoAw : (double)rhoA[q];
    const double gz = (fac + rSq) * zw, gz2 = gz * zw, gsa = zw * (rSq - rAq);
    rhs_u = cg * MT(on_u, a, j) *
            ((hw + h0) * (gz - gz0) +
             (hw - h0) * (gsa + gsa0 + c3 * (rAq - rA0) * (zw - zw0)) +
             (gz2 - gz20));
  }
  if (do_v) {
    const long q = a - ni;
    const double zw = FUSED ? sZw[t - TP] : zwrk[q];
    const double rSq = FUSED ? q_rhoSs : (double)rhoS[q], rAq = FUSED ? q_rhoAs : (double)rhoA[q];
    const double gz = (fac + rSq) * zw, gz2 = gz * zw, gsa = zw * (rSq - rAq);
    rhs_v = cg * MT(om_v, a, j) *
            ((hs + h0) * (gz - gz0) +
             (hs - h0) * (gsa + gsa0 + c3 * (rAq - rA0) * (zw - zw0)) +
             (gz2 - gz20));
  }
  // ---- advection, :1079-1283 ----
  if (p.uv_adv) {
    if (do_u) {
      const double cff1 = UFx2(m, it, j) - UFx2(m, it - 1, j);
      const double cff2 = UFe2(m, it, j + 1) - UFe2(m, it, j);
      rhs_u = rhs_u - (cff1 + cff2);
    }
    if (do_v) {
      const double cff1 = VFx2(m, it + 1, j) - VFx2(m, it, j);
      const double cff2 = VFe2(m, it, j) - VFe2(m, it, j - 1);
      rhs_v = rhs_v - (cff1 + cff2);
    }
  }
  const double D0 = sD[t], Dw = sD[t - 1], Ds = sD[t - TP];
  // ---- Coriolis, :1291-1325 ----
  if (p.uv_cor) {
    const double cf0 = 0.5 * D0 * MT(fomn, a, j);
    const double UFx0 = cf0 * (sV[t] + sV[t + TP]);
    const double VFe0 = cf0 * (sU[t] + sU[t + 1]);
    if (do_u) {
      const double cfw = 0.5 * Dw * MT(fomn, a - 1, j);
      const double UFxw = cfw * (sV[t - 1] + sV[t - 1 + TP]);
      rhs_u = rhs_u + 0.5 * (UFx0 + UFxw);
    }
    if (do_v) {
      const double cfs = 0.5 * Ds * MT(fomn, a - ni, j - 1);
      const double VFes = cfs * (sU[t - TP] + sU[t - TP + 1]);
      rhs_v = rhs_v - 0.5 * (VFe0 + VFes);
    }
  }
  // ---- curvilinear terms, :1333-1382 ----
  if (p.curvgrid && p.uv_adv) {
    auto cell = [&](long q, int jq, int tq, double D, double &ufx, double &vfe) {
      const double cff1 = 0.5 * (sV[tq] + sV[tq + TP]);
      const double cff2 = 0.5 * (sU[tq] + sU[tq + 1]);
      const double cff3 = cff1 * MT(dndx, q, jq);
      const double cff4 = cff2 * MT(dmde, q, jq);
      const double cff = D * (cff3 - cff4);
      ufx = cff * cff1;
      vfe = cff * cff2;
    };
    double u0, v0, u1, v1;
    cell(a, j, t, D0, u0, v0);
    if (do_u) { cell(a - 1, j, t - 1, Dw, u1, v1); rhs_u = rhs_u + 0.5 * (u0 + u1); }
    if (do_v) { cell(a - ni, j - 1, t - TP, Ds, u1, v1); rhs_v = rhs_v - 0.5 * (v0 + v1); }
  }
  // ---- harmonic viscosity, :1394-1471 ----
  if (p.uv_vis2) {
    // q = flat index of the stress point, jq = its row
    auto str_r = [&](long q, int jq, int tq) {
      return HT(visc2_r, q, jq) * sD[tq] * 0.5 *
             (MT(pmon_r, q, jq) * ((MT(pn, q, jq) + MT(pn, q + 1, jq)) * sU[tq + 1] - (MT(pn, q - 1, jq) + MT(pn, q, jq)) * sU[tq]) -
              MT(pnom_r, q, jq) * ((MT(pm, q, jq) + MT(pm, q + ni, jq + 1)) * sV[tq + TP] - (MT(pm, q - ni, jq - 1) + MT(pm, q, jq)) * sV[tq]));
    };
    auto str_p = [&](long q, int jq, int tq) {
      const double Dp = 0.25 * (sD[tq] + sD[tq - 1] + sD[tq - TP] + sD[tq - 1 - TP]);
      const double cffp = HT(visc2_p, q, jq) * Dp * 0.5 *
             (MT(pmon_p, q, jq) * ((MT(pn, q - ni, jq - 1) + MT(pn, q, jq)) * sV[tq] - (MT(pn, q - 1 - ni, jq - 1) + MT(pn, q - 1, jq)) * sV[tq - 1]) +
              MT(pnom_p, q, jq) * ((MT(pm, q - 1, jq) + MT(pm, q, jq)) * sU[tq] - (MT(pm, q - 1 - ni, jq - 1) + MT(pm, q - ni, jq - 1)) * sU[tq - TP]));
      if constexpr (WET) return (masking ? cffp * GF(pmask)[q] : cffp) * GF(pmask_wet)[q];   // WET_DRY, :1436
      return masking ? cffp * GF(pmask)[q] : cffp;              // MASKING, :1433
    };
    const double sr0 = str_r(a, j, t), sp0 = str_p(a, j, t);
    if (do_u) {
      const double srw = str_r(a - 1, j, t - 1), spn = str_p(a + ni, j + 1, t + TP);
      const double onr0 = MT(on_r, a, j), onrw = MT(on_r, a - 1, j), omp0 = MT(om_p, a, j), ompn = MT(om_p, a + ni, j + 1);
      const double UFx0 = onr0 * onr0 * sr0, UFxw = onrw * onrw * srw;
      const double UFe0 = omp0 * omp0 * sp0, UFen = ompn * ompn * spn;
      const double cff1 = 0.5 * (MT(pn, a - 1, j) + MT(pn, a, j)) * (UFx0 - UFxw);
      const double cff2 = 0.5 * (MT(pm, a - 1, j) + MT(pm, a, j)) * (UFen - UFe0);
      rhs_u = rhs_u + (cff1 + cff2);
    }
    if (do_v) {
      const double srs = str_r(a - ni, j - 1, t - TP), spe = str_p(a + 1, j, t + 1);
      const double onp0 = MT(on_p, a, j), onpe = MT(on_p, a + 1, j), omr0 = MT(om_r, a, j), omrs = MT(om_r, a - ni, j - 1);
      const double VFx0 = onp0 * onp0 * sp0, VFxe = onpe * onpe * spe;
      const double VFe0 = omr0 * omr0 * sr0, VFes = omrs * omrs * srs;
      const double cff1 = 0.5 * (MT(pn, a - ni, j - 1) + MT(pn, a, j)) * (VFxe - VFx0);
      const double cff2 = 0.5 * (MT(pm, a - ni, j - 1) + MT(pm, a, j)) * (VFe0 - VFes);
      rhs_v = rhs_v + (cff1 - cff2);
    }
  }
  // ---- biharmonic viscosity, :1474-1740: evaluated by the pass in front of this kernel (roms_launch_step2d_visc4,
  // k_uv3dmix2.hip); UV_VIS4 runs take the general path, so `a` is this thread's own point ----
  if (p.uv_vis4) {
    if (do_u) rhs_u = rhs_u - c->ws2[22][a];
    if (do_v) rhs_v = rhs_v - c->ws2[23][a];
  }
  // ---- coupling between 2-D and 3-D equations, :1884-2065 ----
  double rf_u = 0.0, rf_v = 0.0;                        // rufrc / rvfrc of the first predictor (WET: scaled below)
  if (s.iif == 1 && s.predictor) {
    // never source-mapped (step2d_impl), so owner is always true here
    const gd_t ru_s = (gd_t)(c->F.ru + (long)(s.nstp - 1) * n3w);
    const gd_t rv_s = (gd_t)(c->F.rv + (long)(s.nstp - 1) * n3w);
    const gcd_t ru_n = (gcd_t)(c->F.ru + (long)(s.nnew - 1) * n3w);
    const gcd_t rv_n = (gcd_t)(c->F.rv + (long)(s.nnew - 1) * n3w);
    if (do_u) {
      const double rf = (FUSED ? q_ruf : (double)GF(rufrc)[a]) - rhs_u;
      if (s.iic == s.ntfirst) rhs_u = rhs_u + rf;
      else if (s.iic == s.ntfirst + 1) rhs_u = rhs_u + 1.5 * rf - 0.5 * ru_n[a];
      else rhs_u = rhs_u + (23.0 / 12.0) * rf - (16.0 / 12.0) * ru_n[a] + (5.0 / 12.0) * ru_s[a];
      GF(rufrc)[a] = rf;
      ru_s[a] = rf;
      rf_u = rf;
    }
    if (do_v) {
      const double rf = (FUSED ? q_rvf : (double)GF(rvfrc)[a]) - rhs_v;
      if (s.iic == s.ntfirst) rhs_v = rhs_v + rf;
      else if (s.iic == s.ntfirst + 1) rhs_v = rhs_v + 1.5 * rf - 0.5 * rv_n[a];
      else rhs_v = rhs_v + (23.0 / 12.0) * rf - (16.0 / 12.0) * rv_n[a] + (5.0 / 12.0) * rv_s[a];
      GF(rvfrc)[a] = rf;
      rv_s[a] = rf;
      rf_v = rf;
    }
  } else {
    if (do_u) rhs_u = rhs_u + (FUSED ? q_ruf : (double)GF(rufrc)[a]);
    if (do_v) rhs_v = rhs_v + (FUSED ? q_rvf : (double)GF(rvfrc)[a]);
  }
  // ---- time step, :2098-2255 ----
  const double dtfast = p.dtfast;
  const double Dn0 = (FUSED ? sZn[t] : zeta_new[a]) + h0, Dst0 = (FUSED ? q_zs0 : (double)zs[a]) + h0;
  const int ptsk = 3 - s.kstp;
  const bool am3 = !(s.iif == 1 || s.predictor);
  const double c1 = (s.iif == 1) ? 0.5 * dtfast : dtfast;
  const double a1 = 0.5 * dtfast * 5.0 / 12.0, a2 = 0.5 * dtfast * 8.0 / 12.0, a3 = 0.5 * dtfast * 1.0 / 12.0;
  const gd_t ubn = (gd_t)(c->F.ubar + (long)(s.knew - 1) * nij);
  const gd_t vbn = (gd_t)(c->F.vbar + (long)(s.knew - 1) * nij);
  if (do_u) {
    const long q = a - 1;
    const double cff = (MT(pm, a, j) + MT(pm, q, j)) * (MT(pn, a, j) + MT(pn, q, j));
    const double fc = 1.0 / (Dn0 + ((FUSED ? sZn[t - 1] : zeta_new[q]) + hw));
    const double us = FUSED ? q_us : (double)GF(ubar)[a + (long)(s.kstp - 1) * nij];
    const double zsq = FUSED ? q_zsw : (double)zs[q];
    double un;
    if (!am3) un = (us * (Dst0 + (zsq + hw)) + cff * c1 * rhs_u) * fc;
    else un = (us * (Dst0 + (zsq + hw)) +
               cff * (a1 * rhs_u + a2 * (FUSED ? q_rubk : (double)GF(rubar)[a + (long)(s.kstp - 1) * nij]) -
                      a3 * (FUSED ? q_rubp : (double)GF(rubar)[a + (long)(ptsk - 1) * nij]))) * fc;
    if (masking) un = un * GF(umask)[a];               // MASKING, :2120 / :2175
    if constexpr (WET) {                               // WET_DRY, :2123-2135 / :2178-2184 / :2225-2231
      const double cff7 = wet_factor(GF(umask_wet)[a], un);
      un = un * cff7;
      rhs_u = rhs_u * cff7;
      if (s.iif == 1 && s.predictor) {                 // FIRST_2D_STEP and predictor, :2129-2133
        const double rf = rf_u * cff7;
        GF(rufrc)[a] = rf;
        ((gd_t)(c->F.ru + (long)(s.nstp - 1) * n3w))[a] = rf;
      }
    }
    put(ubn, o, un);
    if (s.predictor && owner) GF(rubar)[a + (long)(s.krhs - 1) * nij] = rhs_u;
    const double un_s = masking ? (p.gamma2 * un) * GF(umask)[a - ni] : p.gamma2 * un;   // wall rows (u2dbc)
    const double un_n = masking ? (p.gamma2 * un) * GF(umask)[a + ni] : p.gamma2 * un;
    if (inline_bc) {                                   // u2dbc closed walls, u2dbc_im.F:51
      if (b.south_edge && j == b.Jstr) put(ubn, o - ni, un_s);
      if (b.north_edge && j == b.Jend) put(ubn, o + ni, un_n);
    }
    if constexpr (FUSED) {
      if (DUnext) {                                    // DUon of level knew, :509-525 (as k2d_flux)
        const double znw = sZn[t - 1];
        DUnext[a] = un * ((0.5 * MT(on_u, a, j)) * (Dn0 + (znw + hw)));
        // wall rows: u = gamma2*u(adjacent row), zeta = zero-gradient copy (u2dbc_im.F:51, zetabc.F:48)
        if (b.south_edge && j == b.Jstr)
          DUnext[a - ni] = un_s * ((0.5 * MT(on_u, a - ni, j - 1)) *
                                   ((sZn[t] * mk(GF(rmask), a - ni) + hs) + (znw * mk(GF(rmask), q - ni) + HT(h, q - ni, j - 1))));
        if (b.north_edge && j == b.Jend)
          DUnext[a + ni] = un_n * ((0.5 * MT(on_u, a + ni, j + 1)) *
                                   ((sZn[t] * mk(GF(rmask), a + ni) + HT(h, a + ni, j + 1)) + (znw * mk(GF(rmask), q + ni) + HT(h, q + ni, j + 1))));
      }
    }
  }
  if (do_v) {
    const long q = a - ni;
    const double cff = (MT(pm, a, j) + MT(pm, q, j - 1)) * (MT(pn, a, j) + MT(pn, q, j - 1));
    const double fc = 1.0 / (Dn0 + ((FUSED ? sZn[t - TP] : zeta_new[q]) + hs));
    const double vs = FUSED ? q_vs : (double)GF(vbar)[a + (long)(s.kstp - 1) * nij];
    const double zsq = FUSED ? q_zss : (double)zs[q];
    double vn;
    if (!am3) vn = (vs * (Dst0 + (zsq + hs)) + cff * c1 * rhs_v) * fc;
    else vn = (vs * (Dst0 + (zsq + hs)) +
               cff * (a1 * rhs_v + a2 * (FUSED ? q_rvbk : (double)GF(rvbar)[a + (long)(s.kstp - 1) * nij]) -
                      a3 * (FUSED ? q_rvbp : (double)GF(rvbar)[a + (long)(ptsk - 1) * nij]))) * fc;
    if (masking) vn = vn * GF(vmask)[a];               // MASKING, :2145 / :2194
    if constexpr (WET) {                               // WET_DRY, :2148-2160 / :2197-2203 / :2246-2252
      const double cff7 = wet_factor(GF(vmask_wet)[a], vn);
      vn = vn * cff7;
      rhs_v = rhs_v * cff7;
      if (s.iif == 1 && s.predictor) {
        const double rf = rf_v * cff7;
        GF(rvfrc)[a] = rf;
        ((gd_t)(c->F.rv + (long)(s.nstp - 1) * n3w))[a] = rf;
      }
    }
    put(vbn, o, vn);
    if (s.predictor && owner) GF(rvbar)[a + (long)(s.krhs - 1) * nij] = rhs_v;
    if constexpr (FUSED) {
      if (DVnext) DVnext[a] = vn * ((0.5 * MT(om_v, a, j)) * (Dn0 + (sZn[t - TP] + hs)));   // :527-544
    }
  }
  if (inline_bc) {                                     // v2dbc closed walls, v2dbc_im.F:52
    if (b.south_edge && j == b.Jstr) put(vbn, o, 0.0);
    if (b.north_edge && j == b.Jend) put(vbn, o + ni, 0.0);
  }
  if constexpr (FUSED) {
    if (DVnext) {                                      // wall rows: v = 0 there
      const double zn0 = sZn[t];
      if (b.south_edge && j == b.Jstr)
        DVnext[a] = 0.0 * ((0.5 * MT(om_v, a, j)) * ((zn0 + h0) + (zn0 * mk(GF(rmask), a - ni) + hs)));
      if (b.north_edge && j == b.Jend)
        DVnext[a + ni] = 0.0 * ((0.5 * MT(om_v, a + ni, j + 1)) * ((zn0 * mk(GF(rmask), a + ni) + HT(h, a + ni, j + 1)) + (zn0 + h0)));
    }
  }
}

// One workgroup per (row, metric array): is the row constant in i (bit for bit) over LBi:UBi?  Lane 0 stores the
// row's value in the table; any difference raises the flag.
__global__ void k_rowm_build(const RomsDev *__restrict__ c, double *__restrict__ tab, int *__restrict__ flag)
{
  DEV_PROLOGUE(c)
  const int jr = blockIdx.x, f = blockIdx.y;
  const double *const A[RM_COUNT] = {c->F.pm, c->F.pn, c->F.on_u, c->F.om_v, c->F.fomn, c->F.dndx, c->F.dmde,
                                     c->F.pmon_r, c->F.pnom_r, c->F.pmon_p, c->F.pnom_p, c->F.om_r, c->F.on_r,
                                     c->F.om_p, c->F.on_p, c->F.h, c->F.visc2_r, c->F.visc2_p};
  const double *a = A[f];
  if (!a) {                           // array not registered: never read by a kernel either
    if (threadIdx.x == 0) tab[(long)jr * RM_COUNT + f] = 0.0;
    return;
  }
  // the columns a kernel can read: the tile's own and two on either side (ghost columns; the reference's padding
  // column beyond them is never read and need not be filled)
  const int i0 = (b.Istr - 2 > b.LBi) ? b.Istr - 2 : b.LBi, i1 = (b.Iend + 2 < b.UBi) ? b.Iend + 2 : b.UBi;
  const unsigned long long *row = reinterpret_cast<const unsigned long long *>(a + (long)jr * ni) - LBi;
  const unsigned long long v0 = row[b.Istr];
  bool differs = false;
  for (int i = i0 + (int)threadIdx.x; i <= i1; i += (int)blockDim.x) differs |= row[i] != v0;
  if (differs) atomicOr(flag + (f >= RM_FIRST_GROUP ? 1 : 0), 1);     // one flag per group
  if (threadIdx.x == 0) tab[(long)jr * RM_COUNT + f] = a[(long)jr * ni + (b.Istr - LBi)];
}

}  // namespace

// 0 = not examined yet, 1 = the metric arrays are independent of i (row table in use), 2 = they are not,
// 3 = as 1 and h, visc2_r, visc2_p are independent of i too
extern "C" int roms_hip_row_metrics_state(void) { return g_ctx.rowm_state == 1 && g_ctx.rowh ? 3 : g_ctx.rowm_state; }

bool roms_rowm_is_table_field(int id)
{
  static const int ids[RM_COUNT] = {FID_pm, FID_pn, FID_on_u, FID_om_v, FID_fomn, FID_dndx, FID_dmde, FID_pmon_r,
                                    FID_pnom_r, FID_pmon_p, FID_pnom_p, FID_om_r, FID_on_r, FID_om_p, FID_on_p,
                                    FID_h, FID_visc2_r, FID_visc2_p};
  for (int q = 0; q < RM_COUNT; q++)
    if (ids[q] == id) return true;
  return false;
}

void roms_rowm_invalidate()
{
  if (g_ctx.rowm_state) step2d_graphs_release();     // the captured launches chose their kernel by the old state
  g_ctx.rowm_state = 0;
}

void roms_rowm_release()
{
  if (g_ctx.rowm_dev) (void)hipFree(g_ctx.rowm_dev);
  g_ctx.rowm_dev = nullptr;
  g_ctx.rowm_nj = 0;
  g_ctx.rowm_state = 0;
  g_ctx.hostc.rowm = nullptr;
}

int roms_rowm_prepare()
{
  if (g_ctx.rowm_state) return 0;
  const roms_bounds_t &b = g_ctx.b;
  const long nj = b.UBj - b.LBj + 1;
  if (g_ctx.rowm_nj != nj) {
    roms_rowm_release();
    HIP_TRY(hipMalloc(&g_ctx.rowm_dev, sizeof(double) * (RM_COUNT * nj + 1)));   // + one word for the flag
    g_ctx.rowm_nj = nj;
  }
  int *flag = reinterpret_cast<int *>(g_ctx.rowm_dev + RM_COUNT * nj);
  int rc = roms_flush_consts();
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(flag, 0, sizeof(double), g_ctx.stream));
  hipLaunchKernelGGL(k_rowm_build, dim3((unsigned)nj, RM_COUNT), dim3(256), 0, g_ctx.stream, g_ctx.devc,
                     g_ctx.rowm_dev, flag);
  KERNEL_CHECK("k_rowm_build");
  int differs[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(differs, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  g_ctx.rowm_state = differs[0] ? 2 : 1;
  g_ctx.rowh = !differs[0] && !differs[1];
  const double *want = differs[0] ? nullptr : g_ctx.rowm_dev;
  if (g_ctx.hostc.rowm != want) {
    g_ctx.hostc.rowm = want;
    g_ctx.devc_dirty = true;
    if ((rc = roms_flush_consts())) return rc;
  }
  return 0;
}

// Launcher used by step2d_impl (k_step2d.hip); s10 = the ten ints of its S2.
int roms_launch_k2d_mom_lds(const int *s10, const double *DUon, const double *DVom, const double *zeta_new,
                            const double *zwrk, double *DUnext, double *DVnext)
{
  const roms_bounds_t &b = g_ctx.b;
  S2 s{s10[0], s10[1], s10[2], s10[3], s10[4], s10[5], s10[6], s10[7], s10[8], s10[9]};
  const int nx = (s.sm == 1) ? (b.UBi - b.LBi + 1) : (b.Iend - b.Istr + 1);
  if (s.sm == 2) {      // fused free-surface + momentum call (source-mapped, fluxes in place)
    s.sm = 1;
    if (g_ctx.rowm_state == 1 && g_ctx.rowh)   // ... and depth and viscosity coefficients as well
      hipLaunchKernelGGL((k2d_mom_lds<true, true, true>), grid2d(nx, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                         g_ctx.devc, s, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr,
                         (const double *)nullptr, (double *)nullptr, (double *)nullptr);
    else if (g_ctx.rowm_state == 1)   // metrics independent of i: row table instead of fifteen 2-D arrays
      hipLaunchKernelGGL((k2d_mom_lds<true, true>), grid2d(nx, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                         g_ctx.devc, s, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr,
                         (const double *)nullptr, (double *)nullptr, (double *)nullptr);
    else
      hipLaunchKernelGGL((k2d_mom_lds<true, false>), grid2d(nx, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                         g_ctx.devc, s, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr,
                         (const double *)nullptr, (double *)nullptr, (double *)nullptr);
  } else if (s.sm == 3) {   // fused call on several tiles: exchanged DUon/DVom in, next call's fluxes out
    s.sm = 0;
    if (g_ctx.rowm_state == 1 && g_ctx.rowh)
      hipLaunchKernelGGL((k2d_mom_lds<true, true, true>), grid2d(nx, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                         g_ctx.devc, s, DUon, DVom, (const double *)nullptr, (const double *)nullptr, DUnext, DVnext);
    else if (g_ctx.rowm_state == 1)
      hipLaunchKernelGGL((k2d_mom_lds<true, true>), grid2d(nx, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                         g_ctx.devc, s, DUon, DVom, (const double *)nullptr, (const double *)nullptr, DUnext, DVnext);
    else
      hipLaunchKernelGGL((k2d_mom_lds<true, false>), grid2d(nx, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                         g_ctx.devc, s, DUon, DVom, (const double *)nullptr, (const double *)nullptr, DUnext, DVnext);
  } else if (g_ctx.p.wet_dry)
    hipLaunchKernelGGL((k2d_mom_lds<false, false, false, true>), grid2d(nx, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s, DUon, DVom, zeta_new, zwrk, (double *)nullptr, (double *)nullptr);
  else
    hipLaunchKernelGGL((k2d_mom_lds<false, false>), grid2d(nx, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream, g_ctx.devc, s,
                       DUon, DVom, zeta_new, zwrk, (double *)nullptr, (double *)nullptr);
  KERNEL_CHECK("k2d_mom_lds");
  return 0;
}
