// k_step3d_uv.hip -- corrector step for 3-D momentum, step3d_uv_tile
// (ROMS/Nonlinear/step3d_uv.F:111-1482).
//
//   k_uv_column  one thread per water column (u-column at (i,j) and v-column at
//                (i,j)): AB3 step u(nnew) += 23/12 dt ru(nrhs) (:303-315),
//                divide by Hz, implicit vertical viscosity in spline form --
//                a wavefront-serial Thomas solve per column with the column
//                state in VGPRs (:346-400, :679-735) -- and replacement of the
//                vertical mean by DU_avg1/DV_avg1 (:466-520).
//                The coupling of the same column (:997-1460: ubar,vbar(:,:,1:2)
//                and the corrected mass fluxes Huon,Hvom with DU_avg2/DV_avg2)
//                needs nothing but the column's own final velocity, so it is
//                done in the same thread from the registers that hold it: the
//                Thomas arrays are dead by then and keep the layer factors and
//                the intermediate fluxes.
//   k_uv_couple  the columns the first kernel does not step -- boundary rows
//                (after u3dbc/v3dbc) with their mean correction (:1087-1110) --
//                one thread per column over IstrT:IendT, JstrT:JendT.
// Algorithmic traffic: read Akv,Hz,ru,rv, read+write u,v(nnew), read+write
// Huon,Hvom = 12 passes.
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

// One momentum column.  off = 1 (u: neighbour i-1) or ni (v: neighbour j-1).
template <int NMAX>
__device__ __forceinline__ void uv_column(const RomsDev *__restrict__ c, long c0, long off, long nij, int N,
                                          gd_t vel, gd_t rhs,
                                          double dc0, double metric, double Davg1,
                                          gd_t Hflx, gd_t bar, double Davg2, bool masking, double msk, bool wet, double wmsk,
                                          double *__restrict__ un)
{
  // un[k * UN_S]: the velocity column, in LDS ([level][thread]) -- the registers it occupied hold hs[k] = Hz(k) +
  // Hz(k) of the neighbour, the one input all four sweeps need: the sweeps after the first read Hz no more (15 -> 9
  // loads per level; the mean and the final sweep have no load to wait for at all)
  constexpr int UN_S = BLK_X * BLK_Y;
  const gcd_t Akv = (gcd_t)(c->F.Akv);
  const gcd_t Hz = (gcd_t)(c->F.Hz);
  const double dt = c->p.dt;
  double hs[NMAX + 1], CF[NMAX + 1], DC[NMAX + 1];
  CF[0] = 0.0;
  DC[0] = 0.0;
  double hzk_m1 = 0.0, ohz_m1 = 0.0, ak_m2 = 0.0;
  double ak_m1 = 0.5 * (Akv[c0 - off] + Akv[c0]);           // AK(0)
  double sumH = 0.0, un_prev = 0.0;
  // the six inputs of level k+1 are requested before level k is computed (the kernel waits for memory on 85 % of its
  // wave-cycles otherwise: a level's loads sit behind the `k <= N` test of its own block)
  double n_aA = Akv[c0 + nij - off], n_aB = Akv[c0 + nij], n_hA = Hz[c0 - off], n_hB = Hz[c0], n_v = vel[c0], n_r = rhs[c0 + nij];
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const double aA = n_aA, aB = n_aB, hA = n_hA, hB = n_hB, v0 = n_v, r0 = n_r;
      if (k + 1 <= N) {
        const long cn = c0 + (long)k * nij;
        n_aA = Akv[cn + nij - off]; n_aB = Akv[cn + nij]; n_hA = Hz[cn - off]; n_hB = Hz[cn]; n_v = vel[cn]; n_r = rhs[cn + nij];
      }
      const double ak_0 = 0.5 * (aA + aB);   // AK(k)
      hs[k] = hA + hB;
      const double hzk = 0.5 * hs[k];
      const double ohz = 1.0 / hzk;
      double uv = v0;
      uv = uv + dc0 * r0;
      uv = uv * ohz;
      un[k * UN_S] = uv;
      const double un_k = uv;
      if (k >= 2) {
        const double c6 = 1.0 / 6.0, c3 = 1.0 / 3.0;
        const double fc = c6 * hzk_m1 - dt * ak_m2 * ohz_m1;
        const double cf = c6 * hzk - dt * ak_0 * ohz;
        const double bc = c3 * (hzk_m1 + hzk) + dt * ak_m1 * (ohz_m1 + ohz);
        const double cff = 1.0 / (bc - fc * CF[k - 2]);
        CF[k - 1] = cff * cf;
        DC[k - 1] = cff * (un_k - un_prev - fc * DC[k - 2]);
      }
      sumH = (k == 1) ? hzk : sumH + hzk;
      un_prev = un_k;
      hzk_m1 = hzk; ohz_m1 = ohz; ak_m2 = ak_m1; ak_m1 = ak_0;
    }
  }
  // back substitution (descending) with the viscous update
  double dcA_up = 0.0, dc_up = 0.0;
#pragma unroll
  for (int kk = NMAX - 1; kk >= 0; kk--) {
    if (kk <= N - 1) {
      double dcA = 0.0;
      if (kk >= 1) {
        const double dc = DC[kk] - CF[kk] * dc_up;
        dc_up = dc;
        const double ak = 0.5 * (Akv[c0 + (long)kk * nij - off] + Akv[c0 + (long)kk * nij]);
        dcA = dc * ak;
      }
      const double hzk = 0.5 * hs[kk + 1];           // level kk+1
      const double ohz = 1.0 / hzk;
      un[(kk + 1) * UN_S] = un[(kk + 1) * UN_S] + dt * ohz * (dcA_up - dcA);
      dcA_up = dcA;
    }
  }
  // vertical mean replacement, step3d_uv.F:466-520 (sums ascend in k)
  double sumU = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const double hzk = 0.5 * hs[k];
      sumU = (k == 1) ? un[k * UN_S] * hzk : sumU + un[k * UN_S] * hzk;
    }
  }
  const double cff1 = 1.0 / (sumH * metric);
  const double corr = (sumU * metric - Davg1) * cff1;
  // final velocity; then the coupling of this column (couple_column below, fix_mean = false) from registers:
  // CF[] now holds DC(i,k) of step3d_uv.F:1010, DC[] the intermediate flux
  const double cffm = 0.5 * metric;
  double DC0 = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const long ck = c0 + (long)(k - 1) * nij;
      double uv = un[k * UN_S] - corr;
      if (masking) uv = uv * msk;                 // MASKING, step3d_uv.F:558 / :891 (msk: times the wet/dry mask, :561 / :894)
      if (wet) rhs[ck + nij] = rhs[ck + nij] * wmsk;   // WET_DRY, step3d_uv.F:563 / :896: ru / rv(nrhs) as well
      vel[ck] = uv;
      un[k * UN_S] = uv;
      const double dck = cffm * hs[k];
      CF[k] = dck;
      DC0 = DC0 + dck;
    }
  }
  DC0 = 1.0 / DC0;
  double bv = DC0 * Davg1;
  if (wet) bv = bv * wmsk;                        // WET_DRY, step3d_uv.F:1042-1044 / :1255-1257
  bar[c0] = bv;
  bar[c0 + nij] = bv;
  double FC0 = 0.0;
#pragma unroll
  for (int k = NMAX; k >= 1; k--) {
    if (k <= N) {
      const double h = 0.5 * (Hflx[c0 + (long)(k - 1) * nij] + un[k * UN_S] * CF[k]);
      DC[k] = h;
      FC0 = FC0 + h;
    }
  }
  FC0 = DC0 * (FC0 - Davg2);
#pragma unroll
  for (int k = 1; k <= NMAX; k++)
    if (k <= N) Hflx[c0 + (long)(k - 1) * nij] = DC[k] - CF[k] * FC0;
}

template <int NMAX>
__global__ void __launch_bounds__(BLK_X *BLK_Y, (NMAX <= 32 ? 2 : 1))      // two waves per SIMD where the level arrays allow it
k_uv_column(const RomsDev *__restrict__ c, int nrhs, int nnew, double cff)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const long c0 = I2(i, j);
  const gcd_t pm = (gcd_t)c->F.pm, pn = (gcd_t)c->F.pn;
  const bool masking = c->p.masking != 0, wet = c->p.wet_dry != 0;
  __shared__ double s_un[(NMAX + 1) * BLK_X * BLK_Y];
  double *const un = s_un + threadIdx.y * BLK_X + threadIdx.x;
  // blockIdx.z selects the component so that both columns do not share VGPRs
  if (XB.z == 0) {
    if (i < b.IstrU) return;
    const double dc0 = cff * (pm[c0] + pm[c0 - 1]) * (pn[c0] + pn[c0 - 1]);
    uv_column<NMAX>(c, c0, 1, nij, N, (gd_t)(c->F.u + (long)(nnew - 1) * n3r), (gd_t)(c->F.ru + (long)(nrhs - 1) * n3w), dc0,
                    GF(on_u)[c0], GF(DU_avg1)[c0], GF(Huon), GF(ubar), GF(DU_avg2)[c0], masking,
                    masking ? umaskw(c, c0) : 1.0, wet, wet ? (double)GF(umask_wet)[c0] : 1.0, un);
  } else {
    if (j < b.JstrV) return;
    const double dc0 = cff * (pm[c0] + pm[c0 - ni]) * (pn[c0] + pn[c0 - ni]);
    uv_column<NMAX>(c, c0, ni, nij, N, (gd_t)(c->F.v + (long)(nnew - 1) * n3r), (gd_t)(c->F.rv + (long)(nrhs - 1) * n3w), dc0,
                    GF(om_v)[c0], GF(DV_avg1)[c0], GF(Hvom), GF(vbar), GF(DV_avg2)[c0], masking,
                    masking ? vmaskw(c, c0) : 1.0, wet, wet ? (double)GF(vmask_wet)[c0] : 1.0, un);
  }
}

// LuvSrc, step3d_uv.F:971-995: u, v(nnew) of a source face = the source's transport of the level over the face's area
// (the layer thickness as the reference writes it, from z_w).  One thread per (source, level); the face map settles two
// sources on one face the way the sequential loop of the reference does.
// The reference sets these velocities after the boundary conditions and couples afterwards; k_uv_column has coupled its
// columns already.  So the mass fluxes of the source faces are set aside before it runs (save = 1) and put back here
// (save = 0), and k_uv_couple couples those columns again, from the source's velocity.
__global__ void k_src_uv(const RomsDev *__restrict__ c, int nnew, int save)
{
  DEV_PROLOGUE(c)
  const int is = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = 1 + (int)blockIdx.y;
  if (is >= c->src.n) return;
  const int i = c->src.I[is], j = c->src.J[is];
  if (!(b.IstrR <= i && i <= b.IendR && b.JstrR <= j && j <= b.JendR)) return;
  const long c0 = I2(i, j);
  const double *__restrict__ z_w = c->F.z_w;
  const double q = c->src.Qsrc[is + (long)c->src.n * (k - 1)];
  const long w1 = c0 + (long)k * nij, w0 = c0 + (long)(k - 1) * nij;
  double *__restrict__ keep = c->src.save + is + (long)c->src.n * (k - 1);
  if (c->src.D[is] == 0) {
    if (c->src.umap[c0] != is + 1) return;
    if (save) { *keep = c->F.Huon[w0]; return; }
    c->F.Huon[w0] = *keep;
    const double cff1 = 1.0 / (c->F.on_u[c0] * 0.5 * (z_w[w1 - 1] - z_w[w0 - 1] + z_w[w1] - z_w[w0]));
    c->F.u[(long)(nnew - 1) * n3r + c0 + (long)(k - 1) * nij] = q * cff1;
  } else if (c->src.D[is] == 1) {
    if (c->src.vmap[c0] != is + 1) return;
    if (save) { *keep = c->F.Hvom[w0]; return; }
    c->F.Hvom[w0] = *keep;
    const double cff1 = 1.0 / (c->F.om_v[c0] * 0.5 * (z_w[w1 - ni] - z_w[w0 - ni] + z_w[w1] - z_w[w0]));
    c->F.v[(long)(nnew - 1) * n3r + c0 + (long)(k - 1) * nij] = q * cff1;
  }
}

// Coupling of one column; comp 0 = u (neighbour i-1), 1 = v (neighbour j-1).
template <int NMAX>
__device__ __forceinline__ void couple_column(const RomsDev *__restrict__ c, long c0, long off, long nij, int N,
                                              gd_t vel, gd_t Hflx,
                                              gd_t bar, double metric, double Davg1, double Davg2,
                                              bool fix_mean, bool masking, double msk, bool wet, double wmsk)
{
  const gcd_t Hz = (gcd_t)(c->F.Hz);
  const double cff = 0.5 * metric;
  double DC0 = 0.0, CF0 = 0.0;
  // dc[k] = DC(i,k) of the reference: kept in registers for the two later passes (it was re-read from Hz twice
  // per component: 4 of the kernel's 13 field passes) -- the same product, so the same bits
  double hu[NMAX + 1], dc[NMAX + 1];
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const long ck = c0 + (long)(k - 1) * nij;
      const double dck = cff * (Hz[ck] + Hz[ck - off]);
      dc[k] = dck;
      DC0 = DC0 + dck;
      CF0 = CF0 + dck * vel[ck];
    }
  }
  DC0 = 1.0 / DC0;
  CF0 = DC0 * (CF0 - Davg1);
  double bv = DC0 * Davg1;
  if (wet) bv = bv * wmsk;                                // WET_DRY, step3d_uv.F:1042-1044 / :1255-1257
  bar[c0] = bv;
  bar[c0 + nij] = bv;
  double FC0 = 0.0;
#pragma unroll
  for (int k = NMAX; k >= 1; k--) {
    if (k <= N) {
      const long ck = c0 + (long)(k - 1) * nij;
      const double dck = dc[k];
      double uv = vel[ck];
      if (fix_mean) {                                     // boundary rows, :1087-1110
        uv = uv - CF0;
        if (masking) uv = uv * msk;                       // MASKING, :1137 / :1166 / :1355 / :1384 (msk: times the wet/dry mask)
        vel[ck] = uv;
      }
      const double h = 0.5 * (Hflx[ck] + uv * dck);
      hu[k] = h;
      FC0 = FC0 + h;
    }
  }
  FC0 = DC0 * (FC0 - Davg2);
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const long ck = c0 + (long)(k - 1) * nij;
      Hflx[ck] = hu[k] - dc[k] * FC0;
    }
  }
}

// Without SPLINES_VVISC (step3d_uv.F:400-464, :733-797; 3 of the reference's 31 three-dimensional applications): the
// AB3 step leaves u(nnew) thickness-weighted, and the implicit vertical viscosity is a tridiagonal system for the velocity
// itself with the layer distances taken from z_r.  A plain column kernel (level arrays in registers / scratch, not
// tuned like k_uv_column): solve, replace the vertical mean (:466-520), then couple_column from memory.
template <int NMAX>
__device__ __forceinline__ void uv_column_classic(const RomsDev *__restrict__ c, long c0, long off, long nij, int N, gd_t vel,
                                                  gd_t rhs, double dc0, double metric, double Davg1, gd_t Hflx, gd_t bar,
                                                  double Davg2, bool masking, double msk, bool wet, double wmsk)
{
  const gcd_t Akv = (gcd_t)(c->F.Akv), Hz = (gcd_t)(c->F.Hz), z_r = (gcd_t)(c->F.z_r);
  const double dt = c->p.dt;
  double hzk[NMAX + 1], FC[NMAX + 1], CF[NMAX + 1], DC[NMAX + 1];
  FC[0] = 0.0;
  const double cffv = -c->p.lambda * dt / 0.5;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const long ck = c0 + (long)(k - 1) * nij;
      hzk[k] = 0.5 * (Hz[ck - off] + Hz[ck]);
      DC[k] = vel[ck] + dc0 * rhs[ck + nij];                    // :316-318 (no division by the thickness)
      if (k <= N - 1) {
        const double ak = 0.5 * (Akv[ck + nij - off] + Akv[ck + nij]);      // AK(i,k)
        const double cff1 = 1.0 / (z_r[ck + nij] + z_r[ck + nij - off] - z_r[ck] - z_r[ck - off]);
        FC[k] = cffv * cff1 * ak;
      } else FC[k] = 0.0;
    }
  }
  // BC(k) = Hzk(k) - FC(k) - FC(k-1); forward elimination
  {
    const double cff = 1.0 / (hzk[1] - FC[1] - FC[0]);
    CF[1] = cff * FC[1];
    DC[1] = cff * DC[1];
  }
#pragma unroll
  for (int k = 2; k <= NMAX - 1; k++) {
    if (k <= N - 1) {
      const double bc = hzk[k] - FC[k] - FC[k - 1];
      const double cff = 1.0 / (bc - FC[k - 1] * CF[k - 1]);
      CF[k] = cff * FC[k];
      DC[k] = cff * (DC[k] - FC[k - 1] * DC[k - 1]);
    }
  }
#pragma unroll
  for (int k = 2; k <= NMAX; k++)
    if (k == N) {
      const double bc = hzk[k] - FC[k] - FC[k - 1];
      DC[k] = (DC[k] - FC[k - 1] * DC[k - 1]) / (bc - FC[k - 1] * CF[k - 1]);
    }
#pragma unroll
  for (int k = NMAX - 1; k >= 1; k--)
    if (k <= N - 1) DC[k] = DC[k] - CF[k] * DC[k + 1];
  // vertical mean replacement, :466-520 (sums ascend in k)
  double sumH = 0.0, sumU = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      sumH = (k == 1) ? hzk[k] : sumH + hzk[k];
      sumU = (k == 1) ? DC[k] * hzk[k] : sumU + DC[k] * hzk[k];
    }
  }
  const double cff1 = 1.0 / (sumH * metric);
  const double corr = (sumU * metric - Davg1) * cff1;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const long ck = c0 + (long)(k - 1) * nij;
      double uv = DC[k] - corr;
      if (masking) uv = uv * msk;
      if (wet) rhs[ck + nij] = rhs[ck + nij] * wmsk;
      vel[ck] = uv;
    }
  }
  couple_column<NMAX>(c, c0, off, nij, N, vel, Hflx, bar, metric, Davg1, Davg2, false, masking, msk, wet, wmsk);
}

template <int NMAX>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_uv_column_classic(const RomsDev *__restrict__ c, int nrhs, int nnew, double cff)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const long c0 = I2(i, j);
  const gcd_t pm = (gcd_t)c->F.pm, pn = (gcd_t)c->F.pn;
  const bool masking = c->p.masking != 0, wet = c->p.wet_dry != 0;
  if (XB.z == 0) {
    if (i < b.IstrU) return;
    const double dc0 = cff * (pm[c0] + pm[c0 - 1]) * (pn[c0] + pn[c0 - 1]);
    uv_column_classic<NMAX>(c, c0, 1, nij, N, (gd_t)(c->F.u + (long)(nnew - 1) * n3r), (gd_t)(c->F.ru + (long)(nrhs - 1) * n3w),
                            dc0, GF(on_u)[c0], GF(DU_avg1)[c0], GF(Huon), GF(ubar), GF(DU_avg2)[c0], masking,
                            masking ? umaskw(c, c0) : 1.0, wet, wet ? (double)GF(umask_wet)[c0] : 1.0);
  } else {
    if (j < b.JstrV) return;
    const double dc0 = cff * (pm[c0] + pm[c0 - ni]) * (pn[c0] + pn[c0 - ni]);
    uv_column_classic<NMAX>(c, c0, ni, nij, N, (gd_t)(c->F.v + (long)(nnew - 1) * n3r), (gd_t)(c->F.rv + (long)(nrhs - 1) * n3w),
                            dc0, GF(om_v)[c0], GF(DV_avg1)[c0], GF(Hvom), GF(vbar), GF(DV_avg2)[c0], masking,
                            masking ? vmaskw(c, c0) : 1.0, wet, wet ? (double)GF(vmask_wet)[c0] : 1.0);
  }
}

template <int NMAX>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_uv_couple(const RomsDev *__restrict__ c, int nnew)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrT + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrT + XB.y * BLK_Y + threadIdx.y;
  if (i > b.IendT || j > b.JendT) return;
  const long c0 = I2(i, j);
  const bool ns_wall = !b.NSperiodic;
  const bool masking = c->p.masking != 0, wet = c->p.wet_dry != 0;
  // columns stepped by k_uv_column were coupled there (but for the source faces, which k_src_uv has set since)
  const bool inner = i <= b.Iend && j >= b.Jstr && j <= b.Jend;
  const bool srcs = c->src.n > 0;
  if (XB.z == 0) {
    if (inner && i >= b.IstrU && !(srcs && c->src.umap[c0] != 0)) return;
    if (i < b.IstrP) return;
    // boundary points whose vertical mean is replaced after the boundary conditions: the wall rows (:1131-1190) and,
    // without E-W periodicity, the western / eastern boundary columns (:1075-1125)
    const bool fix = (ns_wall && (j == 0 || j == b.Mm + 1) && i >= b.IstrU && i <= b.Iend) ||
                     (!b.EWperiodic && ((b.west_edge && i == b.Istr) || (b.east_edge && i == b.Iend + 1)));
    couple_column<NMAX>(c, c0, 1, nij, N, (gd_t)(c->F.u + (long)(nnew - 1) * n3r), GF(Huon), GF(ubar), GF(on_u)[c0],
                        GF(DU_avg1)[c0], GF(DU_avg2)[c0], fix, masking, masking ? umaskw(c, c0) : 1.0, wet,
                        wet ? (double)GF(umask_wet)[c0] : 1.0);
  } else {
    if (inner && i >= b.Istr && j >= b.JstrV && !(srcs && c->src.vmap[c0] != 0)) return;
    if (j < b.Jstr) return;
    const bool fix = (ns_wall && (j == 1 || j == b.Mm + 1) && i >= b.Istr && i <= b.Iend) ||
                     (!b.EWperiodic && ((b.west_edge && i == b.Istr - 1) || (b.east_edge && i == b.Iend + 1)));
    couple_column<NMAX>(c, c0, ni, nij, N, (gd_t)(c->F.v + (long)(nnew - 1) * n3r), GF(Hvom), GF(vbar), GF(om_v)[c0],
                        GF(DV_avg1)[c0], GF(DV_avg2)[c0], fix, masking, masking ? vmaskw(c, c0) : 1.0, wet,
                        wet ? (double)GF(vmask_wet)[c0] : 1.0);
  }
}

}  // namespace

extern "C" int roms_hip_step3d_uv(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_step3d_uv");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  const roms_bounds_t &b = g_ctx.b;
  const double dt = g_ctx.p.dt;
  double cff;
  if (s->iic == s->ntfirst) cff = 0.25 * dt;
  else if (s->iic == s->ntfirst + 1) cff = 0.25 * dt * 3.0 / 2.0;
  else cff = 0.25 * dt * 23.0 / 12.0;
  {
    ScopedTimer tm("step3d_uv");
    if ((g_ctx.p.point_sources & 1) && g_ctx.hostc.src.n > 0) {      // LuvSrc: set the source faces' mass fluxes aside
      hipLaunchKernelGGL(k_src_uv, dim3((g_ctx.hostc.src.n + 63) / 64, b.N), dim3(64), 0, g_ctx.stream, g_ctx.devc, s->nnew, 1);
      KERNEL_CHECK("k_src_uv (save)");
    }
    dim3 grid = grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1);
    grid.z = 2;
    if (!g_ctx.p.splines_vvisc) {           // without SPLINES_VVISC: the plain column kernel
      if (b.N <= 32) hipLaunchKernelGGL(k_uv_column_classic<32>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, cff);
      else if (b.N <= ROMS_MAXN) hipLaunchKernelGGL(k_uv_column_classic<ROMS_MAXN>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, cff);
      else return roms_fail("roms_hip_step3d_uv", "N > 64 not instantiated");
    } else if (b.N <= 16) hipLaunchKernelGGL(k_uv_column<16>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, cff);
    else if (b.N <= 32) hipLaunchKernelGGL(k_uv_column<32>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, cff);
    else if (b.N <= 48) hipLaunchKernelGGL(k_uv_column<48>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, cff);
    else if (b.N <= ROMS_MAXN) hipLaunchKernelGGL(k_uv_column<ROMS_MAXN>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, cff);
    else return roms_fail("roms_hip_step3d_uv", "N > 64 not instantiated");
    KERNEL_CHECK("k_uv_column");
    if ((rc = bc_u3d(s->nnew, s->nstp))) return rc;
    if ((rc = bc_v3d(s->nnew, s->nstp))) return rc;
    if ((g_ctx.p.point_sources & 1) && g_ctx.hostc.src.n > 0) {
      hipLaunchKernelGGL(k_src_uv, dim3((g_ctx.hostc.src.n + 63) / 64, b.N), dim3(64), 0, g_ctx.stream, g_ctx.devc, s->nnew, 0);
      KERNEL_CHECK("k_src_uv");
    }
    dim3 grid2 = grid2d(b.IendT - b.IstrT + 1, b.JendT - b.JstrT + 1);
    grid2.z = 2;
    if (b.N <= 16) hipLaunchKernelGGL(k_uv_couple<16>, grid2, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nnew);
    else if (b.N <= 32) hipLaunchKernelGGL(k_uv_couple<32>, grid2, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nnew);
    else hipLaunchKernelGGL(k_uv_couple<ROMS_MAXN>, grid2, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nnew);
    KERNEL_CHECK("k_uv_couple");
  }
  const long nij = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1);
  const long n3r = nij * b.N;
  halo_batch_begin();
  halo_exchange3d(GT_U, b.N, g_ctx.dev[FID_u] + (long)(s->nnew - 1) * n3r);
  halo_exchange3d(GT_V, b.N, g_ctx.dev[FID_v] + (long)(s->nnew - 1) * n3r);
  halo_exchange3d(GT_U, b.N, g_ctx.dev[FID_Huon]);
  halo_exchange3d(GT_V, b.N, g_ctx.dev[FID_Hvom]);
  halo_exchange3d(GT_U, 2, g_ctx.dev[FID_ubar]);
  halo_exchange3d(GT_V, 2, g_ctx.dev[FID_vbar]);
  return halo_batch_end();
}
