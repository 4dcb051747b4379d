// k_step2d.hip -- split-explicit barotropic engine, step2d_tile
// (ROMS/Nonlinear/step2d_LF_AM3.h:137-2528: leap-frog predictor / Adams-Moulton
// corrector) and the LOOP_2D sequencing of main3d.F:592-700.
//
// Inside LOOP_2D a step2d call is ONE launch of k2d_mom_lds<true> (k_step2d_mom.hip:
// free surface, fast-time averages and momentum together) plus, on several tiles, one
// batched halo exchange.  The kernels of this file serve the calls that cannot take that
// route -- a stand-alone roms_hip_step2d on several tiles, and the last predictor of the
// loop, which only finishes the averages:
//   k2d_flux   Drhs, DUon, DVom two points into the halo (:509-544)
//   k2d_zeta   fast-time averaging (:614-682) and the free-surface step
//              (:770-868): zeta(knew), rzeta(krhs); zeta_new and zwrk go to
//              device scratch on the extended range so the momentum kernel can
//              use them at i-1 / j-1 without another exchange
// followed by k2d_mom_lds<false> with the reference's boundary-condition and halo calls
// in between.  The reference's ~25 private (IminS:ImaxS,JminS:JmaxS) work arrays become
// registers / LDS; only DUon, DVom, zeta_new, zwrk live in device scratch.
#include "roms_dev.h"
#include <cstdlib>
#include <map>

int roms_entry_check(const char *name);
int roms_launch_step2d_visc4(int krhs);      // k_uv3dmix2.hip
int roms_launch_k2d_mom_lds(const int *s10, const double *DUon, const double *DVom, const double *zeta_new,
                            const double *zwrk, double *DUnext = nullptr, double *DVnext = nullptr);   // k_step2d_mom.hip

namespace {

struct S2 {
  int krhs, kstp, knew, nstp, nnew, iif, iic, ntfirst, predictor;
  int sm;   // 1 = single tile, E-W periodic: kernels cover the whole allocated tile and every
            // ghost point is computed from its SOURCE point (periodic image in i, wall mirror in
            // j), which reproduces "compute interior, apply zetabc/u2dbc/v2dbc, periodic copy"
            // bit for bit without the extra launches of exchange_*2d_tile and the 2-D BCs.
};

__device__ __forceinline__ int wrap_i(const roms_bounds_t &b, int i)
{
  return (i < 1) ? i + b.Lm : ((i > b.Lm) ? i - b.Lm : i);
}

__device__ __forceinline__ void zeta_point(const RomsDev *__restrict__ c, const S2 &s, const double rhs,
                                           double *__restrict__ zeta_new, double *__restrict__ zwrk, long a, long o,
                                           bool write_scratch, bool write_zeta, bool write_rzeta, long nij, long ni);

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_flux(const RomsDev *__restrict__ c, S2 s, double *__restrict__ DUon, double *__restrict__ DVom)
{
  DEV_PROLOGUE(c)
  const int i0 = s.sm ? b.LBi : b.IstrU - 2, i1 = s.sm ? (b.Lm + b.NghostPoints) : b.Iendp2;
  const int i = i0 + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.JstrV - 2 + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > i1 || j > b.Jendp2) return;
  const int is = s.sm ? wrap_i(b, i) : i;
  const gcd_t zeta = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
  const gcd_t h = (gcd_t)(c->F.h);
  const long a = I2(is, j), o = I2(i, j);
  const double Drhs = zeta[a] + h[a];
  if (s.sm || i >= b.IstrU - 1) {
    const double cff = 0.5 * GF(on_u)[a];
    const double cff1 = cff * (Drhs + (zeta[a - 1] + h[a - 1]));
    DUon[o] = GF(ubar)[a + (long)(s.krhs - 1) * nij] * cff1;
  }
  if (j >= b.JstrV - 1) {
    const double cff = 0.5 * GF(om_v)[a];
    const double cff1 = cff * (Drhs + (zeta[a - ni] + h[a - ni]));
    DVom[o] = GF(vbar)[a + (long)(s.krhs - 1) * nij] * cff1;
  }
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_zeta(const RomsDev *__restrict__ c, S2 s, const double *__restrict__ DUon, const double *__restrict__ DVom,
         double *__restrict__ zeta_new, double *__restrict__ zwrk)
{
  DEV_PROLOGUE(c)
  // thread range covers both the averaging range (IstrR:IendR,JstrR:JendR) and
  // the extended zeta range (IstrU-1:Iend, JstrV-1:Jend)
  const int i0 = min(b.IstrR, b.IstrU - 1), j0 = min(b.JstrR, b.JstrV - 1);
  const int i = i0 + blockIdx.x * BLK_X + threadIdx.x;
  const int j = j0 + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendR || j > b.JendR) return;
  const roms_params_t &p = c->p;
  const long a = I2(i, j);
  const int iif = s.iif, nfast = p.nfast;
  const gcd_t zk = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
  // ---- fast-time averaging, :614-682 ----
  const bool inR = i >= b.IstrR && j >= b.JstrR;
  if (inR) {
    const bool inU = i >= b.Istr, inV = j >= b.Jstr;
    if (s.predictor) {
      if (iif == 1) {
        const double cff2 = (-1.0 / 12.0) * p.weight2[iif];
        GF(Zt_avg1)[a] = 0.0;
        if (inU) { GF(DU_avg1)[a] = 0.0; GF(DU_avg2)[a] = cff2 * DUon[a]; }
        if (inV) { GF(DV_avg1)[a] = 0.0; GF(DV_avg2)[a] = cff2 * DVom[a]; }
      } else {
        const double cff1 = p.weight1[iif - 2];
        const double cff2 = (8.0 / 12.0) * p.weight2[iif - 1] - (1.0 / 12.0) * p.weight2[iif];
        GF(Zt_avg1)[a] = GF(Zt_avg1)[a] + cff1 * zk[a];
        if (inU) {
          GF(DU_avg1)[a] = GF(DU_avg1)[a] + cff1 * DUon[a];
          GF(DU_avg2)[a] = GF(DU_avg2)[a] + cff2 * DUon[a];
        }
        if (inV) {
          GF(DV_avg1)[a] = GF(DV_avg1)[a] + cff1 * DVom[a];
          GF(DV_avg2)[a] = GF(DV_avg2)[a] + cff2 * DVom[a];
        }
      }
    } else {
      const double cff2 = (iif == 1) ? p.weight2[iif - 1] : (5.0 / 12.0) * p.weight2[iif - 1];
      if (inU) GF(DU_avg2)[a] = GF(DU_avg2)[a] + cff2 * DUon[a];
      if (inV) GF(DV_avg2)[a] = GF(DV_avg2)[a] + cff2 * DVom[a];
    }
  }
  if (iif > nfast) return;
  // ---- free surface, :770-868 ----
  if (i < b.IstrU - 1 || i > b.Iend || j < b.JstrV - 1 || j > b.Jend) return;
  const bool own = i >= b.Istr && j >= b.Jstr;
  const double rhs = (DUon[a] - DUon[a + 1]) + (DVom[a] - DVom[a + ni]);
  // zeta(knew) is stored on the extended range too: the low-side ghost value computed here is, bit
  // for bit, what the exchange delivers later
  zeta_point(c, s, rhs, zeta_new, zwrk, a, a, true, true, own, nij, ni);
}

// Source-mapped variant (single tile, E-W periodic, closed N-S walls): one
// thread per ALLOCATED point.  Thread (i,j) evaluates the free-surface step at
// its source point (periodic image in i; wall row -> adjacent interior row =
// the zero-gradient zetabc) and stores at (i,j): zeta(knew), rzeta(krhs),
// zeta_new and zwrk come out with their ghost points already filled.
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_zeta_sm(const RomsDev *__restrict__ c, S2 s, const double *__restrict__ DUon, const double *__restrict__ DVom,
            double *__restrict__ zeta_new, double *__restrict__ zwrk)
{
  DEV_PROLOGUE(c)
  const int i = b.LBi + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.LBj + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.Lm + b.NghostPoints || j > b.UBj) return;
  const roms_params_t &p = c->p;
  const int iif = s.iif, nfast = p.nfast;
  const long o = I2(i, j);
  // DUon / DVom (:509-544): from scratch, or -- when k2d_flux was not launched (DUon == nullptr) --
  // evaluated in place with the same expression, so no separate flux kernel is needed
  const gcd_t zkq = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
  const gcd_t hq = (gcd_t)(c->F.h);
  const gcd_t ubq = (gcd_t)(c->F.ubar + (long)(s.krhs - 1) * nij);
  const gcd_t vbq = (gcd_t)(c->F.vbar + (long)(s.krhs - 1) * nij);
  auto du = [&](long q) -> double {
    if (DUon) return DUon[q];
    const double cff = 0.5 * GF(on_u)[q];
    const double cff1 = cff * ((zkq[q] + hq[q]) + (zkq[q - 1] + hq[q - 1]));
    return ubq[q] * cff1;
  };
  auto dv = [&](long q) -> double {
    if (DVom) return DVom[q];
    const double cff = 0.5 * GF(om_v)[q];
    const double cff1 = cff * ((zkq[q] + hq[q]) + (zkq[q - ni] + hq[q - ni]));
    return vbq[q] * cff1;
  };
  // ---- fast-time averaging on the owned ranges only, :614-682 ----
  if (i >= b.IstrR && i <= b.IendR && j >= b.JstrR && j <= b.JendR) {
    const gcd_t zk = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
    const bool inU = i >= b.Istr, inV = j >= b.Jstr;
    if (s.predictor) {
      if (iif == 1) {
        const double cff2 = (-1.0 / 12.0) * p.weight2[iif];
        GF(Zt_avg1)[o] = 0.0;
        if (inU) { GF(DU_avg1)[o] = 0.0; GF(DU_avg2)[o] = cff2 * du(o); }
        if (inV) { GF(DV_avg1)[o] = 0.0; GF(DV_avg2)[o] = cff2 * dv(o); }
      } else {
        const double cff1 = p.weight1[iif - 2];
        const double cff2 = (8.0 / 12.0) * p.weight2[iif - 1] - (1.0 / 12.0) * p.weight2[iif];
        GF(Zt_avg1)[o] = GF(Zt_avg1)[o] + cff1 * zk[o];
        if (inU) {
          GF(DU_avg1)[o] = GF(DU_avg1)[o] + cff1 * du(o);
          GF(DU_avg2)[o] = GF(DU_avg2)[o] + cff2 * du(o);
        }
        if (inV) {
          GF(DV_avg1)[o] = GF(DV_avg1)[o] + cff1 * dv(o);
          GF(DV_avg2)[o] = GF(DV_avg2)[o] + cff2 * dv(o);
        }
      }
    } else {
      const double cff2 = (iif == 1) ? p.weight2[iif - 1] : (5.0 / 12.0) * p.weight2[iif - 1];
      if (inU) GF(DU_avg2)[o] = GF(DU_avg2)[o] + cff2 * du(o);
      if (inV) GF(DV_avg2)[o] = GF(DV_avg2)[o] + cff2 * dv(o);
    }
  }
  if (iif > nfast) return;
  // ---- free surface at the source point ----
  const int is = wrap_i(b, i);
  int js = j;
  if (b.south_edge && j == b.Jstr - 1) js = b.Jstr;       // zetabc closed: zero gradient
  if (b.north_edge && j == b.Jend + 1) js = b.Jend;
  if (js < b.Jstr || js > b.Jend) return;
  const long a = I2(is, js);
  const bool own_row = (js == j);
  const double rhs = (du(a) - du(a + 1)) + (dv(a) - dv(a + ni));
  zeta_point(c, s, rhs, zeta_new, zwrk, a, o, own_row, true, own_row, nij, ni);
}

// One free-surface point: evaluate at index a, store at index o.
__device__ __forceinline__ void zeta_point(const RomsDev *__restrict__ c, const S2 &s, const double rhs,
                                           double *__restrict__ zeta_new, double *__restrict__ zwrk, long a, long o,
                                           bool write_scratch, bool write_zeta, bool write_rzeta, long nij, long ni)
{
  const roms_params_t &p = c->p;
  const int iif = s.iif;
  const gcd_t zk = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
  const double dtfast = p.dtfast;
  const gcd_t zs = (gcd_t)(c->F.zeta + (long)(s.kstp - 1) * nij);
  const double pmn_a = GF(pm)[a], pn_a = GF(pn)[a];
  double zn, zw;
  if (iif == 1) {
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
  if (write_scratch) { zeta_new[o] = zn; zwrk[o] = zw; }
  // WET_DRY && MASKING, :863-866: the shared array (not zeta_new) keeps the total depth of a land cell at Dcrit
  if (write_zeta)
    GF(zeta)[o + (long)(s.knew - 1) * nij] = (p.wet_dry && p.masking) ? zn + (p.Dcrit - GF(h)[a]) * (1.0 - GF(rmask)[a]) : zn;
  if (write_rzeta && s.predictor) GF(rzeta)[o + (long)(s.krhs - 1) * nij] = rhs;
}

// ---------------------------------------------------------------------------
// WET_DRY: wetdry_tile (wetdry.F:93-393), called by every step2d with zeta(:,:,kstp) after the fast-time averaging
// (step2d_LF_AM3.h:729-749), and wetdry_ini_tile (:395-561).  One thread per point of (IstrR:IendR,JstrR:JendR); the
// rho-point flag "sea and total depth above Dcrit" (:190-200) of the point and of its three lower neighbours is
// evaluated in registers (the reference's private array `wetdry`), so the masks need no second pass:
//   mode 0  a fast step (iif <= nfast): wetdry_mask_tile (:563-716) -- u / v masks 2 / 0 / +-1, and the running sum
//           rmask_wet_avg (:214-227; `first` = the first predictor of the loop: the sum starts)
//   mode 1  after the loop (iif = nfast+1): the flag is AINT(rmask_wet_avg / (2 nfast)), wetdry_avg_mask_tile
//           (:734-917) with DU_avg1, DV_avg1, and the full masks (:325-345)
//   mode 2  wetdry_ini_tile: the flag from zeta(Tindex), wetdry_avg_mask_tile with ubar, vbar(Tindex), full masks
// The caller exchanges the masks afterwards, as the reference does.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_wetdry(const RomsDev *__restrict__ c, int mode, int first, int kstp)
{
  DEV_PROLOGUE(c)
  const int i = b.IstrR + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.JstrR + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendR || j > b.JendR) return;
  const roms_params_t &p = c->p;
  const long a = I2(i, j);
  const gcd_t zk = (gcd_t)(c->F.zeta + (long)(kstp - 1) * nij);
  const gcd_t h = (gcd_t)c->F.h, rmask = (gcd_t)c->F.rmask, avg = (gcd_t)c->F.rmask_wet_avg;
  const double eps = 1.0E-10;
  const double cffa = 1.0 / (double)(2 * p.nfast);
  auto flag = [&](long q) -> double {
    if (mode == 1) return trunc(avg[q] * cffa);
    double w = 1.0;
    if (p.masking) w = w * rmask[q];
    if ((zk[q] + h[q]) <= (p.Dcrit + eps)) w = 0.0;
    return w;
  };
  const bool inU = i >= b.Istr, inV = j >= b.Jstr;       // i-1 / j-1 exist in the flag's range (Istr-1:IendR,Jstr-1:JendR)
  const double w0 = flag(a);
  const double wW = inU ? flag(a - 1) : 0.0, wS = inV ? flag(a - ni) : 0.0, wSW = (inU && inV) ? flag(a - 1 - ni) : 0.0;
  GF(rmask_wet)[a] = w0;
  if (mode == 0) GF(rmask_wet_avg)[a] = first ? w0 : GF(rmask_wet_avg)[a] + w0;
  double um = 0.0, vm = 0.0;
  if (inU) {
    double cff1 = wW + w0;
    if (cff1 == 1.0) cff1 = wW - w0;
    um = cff1;
    if (mode != 0) {
      const double DU = mode == 1 ? (double)GF(DU_avg1)[a] : (double)GF(ubar)[a + (long)(kstp - 1) * nij];
      const double cff5 = fabs(fabs(cff1) - 1.0);
      const double cff6 = 0.5 + copysign(0.5, DU) * cff1;
      um = 0.5 * cff1 * cff5 + cff6 * (1.0 - cff5);
      if (DU == 0.0 && (wW + w0) <= 1.0) um = 0.0;       // catch lone ponds
    }
    GF(umask_wet)[a] = um;
  }
  if (inV) {
    double cff1 = wS + w0;
    if (cff1 == 1.0) cff1 = wS - w0;
    vm = cff1;
    if (mode != 0) {
      const double DV = mode == 1 ? (double)GF(DV_avg1)[a] : (double)GF(vbar)[a + (long)(kstp - 1) * nij];
      const double cff5 = fabs(fabs(cff1) - 1.0);
      const double cff6 = 0.5 + copysign(0.5, DV) * cff1;
      vm = 0.5 * cff1 * cff5 + cff6 * (1.0 - cff5);
      if (DV == 0.0 && (wS + w0) <= 1.0) vm = 0.0;
    }
    GF(vmask_wet)[a] = vm;
  }
  double pw = 0.0;
  if (inU && inV) {
    // :631-683: 1 with four or three wet neighbours, 2 with two on the same side, 0 otherwise
    const bool A = wW > 0.5, B = w0 > 0.5, C = wSW > 0.5, D = wS > 0.5;      // (i-1,j) (i,j) (i-1,j-1) (i,j-1)
    const int n = (int)A + (int)B + (int)C + (int)D;
    if (n >= 3) pw = 1.0;
    else if (n == 2 && !((A && D) || (B && C))) pw = 2.0;
    GF(pmask_wet)[a] = pw;
  }
  if (mode != 0) {                                       // :325-345 / :478-498 (as written: pmask_full is never below 2)
    GF(rmask_full)[a] = w0 * rmask[a];
    // (LuvSrc, wetdry.F:307-320 / :511-524: the output masks count a source face as water; the source loop of the
    // reference runs over IstrR:IendR, JstrR:JendR, which holds every inU / inV point)
    const bool srcs = c->src.n > 0;
    if (inU) GF(umask_full)[a] = (srcs && c->src.umap[a] != 0) ? 1.0 : um * GF(umask)[a];
    if (inV) GF(vmask_full)[a] = (srcs && c->src.vmap[a] != 0) ? 1.0 : vm * GF(vmask)[a];
    if (inU && inV) GF(pmask_full)[a] = fmax(pw * GF(pmask)[a], 2.0);
  }
}

static int wetdry_launch(int mode, int first, int kstp)
{
  const roms_bounds_t &b = g_ctx.b;
  hipLaunchKernelGGL(k2d_wetdry, grid2d(b.IendR - b.IstrR + 1, b.JendR - b.JstrR + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, mode, first, kstp);
  KERNEL_CHECK("k2d_wetdry");
  halo_batch_begin();
  halo_exchange2d(GT_P, g_ctx.dev[FID_pmask_wet]);
  halo_exchange2d(GT_R, g_ctx.dev[FID_rmask_wet]);
  halo_exchange2d(GT_U, g_ctx.dev[FID_umask_wet]);
  halo_exchange2d(GT_V, g_ctx.dev[FID_vmask_wet]);
  if (mode == 0) halo_exchange2d(GT_R, g_ctx.dev[FID_rmask_wet_avg]);
  else {
    halo_exchange2d(GT_P, g_ctx.dev[FID_pmask_full]);
    halo_exchange2d(GT_R, g_ctx.dev[FID_rmask_full]);
    halo_exchange2d(GT_U, g_ctx.dev[FID_umask_full]);
    halo_exchange2d(GT_V, g_ctx.dev[FID_vmask_full]);
  }
  return halo_batch_end();
}

// DUon/DVom scratch already holds the exchanged fluxes of barotropic level g_flux_lev (left there by the
// previous call of the same LOOP_2D)
bool g_flux_ready = false;
int g_flux_lev = 0;
int g_flux_buf = 0;       // which scratch pair holds them

// LuvSrc: ubar, vbar(knew) at the source faces = the source's transport over the face's new water column
// (step2d_LF_AM3.h:2484-2502; after the boundary conditions, before the exchange).  One thread per source; two sources
// on one face would race where the reference lets the later one win: the face map decides (its entry is that later one).
__global__ void k2d_src_bar(const RomsDev *__restrict__ c, int knew)
{
  DEV_PROLOGUE(c)
  const int is = blockIdx.x * blockDim.x + threadIdx.x;
  if (is >= c->src.n) return;
  const int i = c->src.I[is], j = c->src.J[is];
  if (!(b.IstrR <= i && i <= b.IendR && b.JstrR <= j && j <= b.JendR)) return;
  const long c0 = I2(i, j);
  const double *__restrict__ zeta = c->F.zeta + (long)(knew - 1) * nij;
  const double *__restrict__ h = c->F.h;
  if (c->src.D[is] == 0) {
    if (c->src.umap[c0] != is + 1) return;
    const double cff = 1.0 / (c->F.on_u[c0] * 0.5 * (zeta[c0 - 1] + h[c0 - 1] + zeta[c0] + h[c0]));
    c->F.ubar[c0 + (long)(knew - 1) * nij] = c->src.Qbar[is] * cff;
  } else if (c->src.D[is] == 1) {
    if (c->src.vmap[c0] != is + 1) return;
    const double cff = 1.0 / (c->F.om_v[c0] * 0.5 * (zeta[c0 - ni] + h[c0 - ni] + zeta[c0] + h[c0]));
    c->F.vbar[c0 + (long)(knew - 1) * nij] = c->src.Qbar[is] * cff;
  }
}

// LwSrc, step2d_LF_AM3.h:890-908: the new free surface of the cells with a cell-centred source rises by the source's
// volume over the cell's area (before zetabc).  ONE thread walks the table: sources sharing a cell add up in the
// table's order, as in the reference.
__global__ void k2d_src_zeta(const RomsDev *__restrict__ c, int knew)
{
  DEV_PROLOGUE(c)
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double *__restrict__ zeta = c->F.zeta + (long)(knew - 1) * nij;
  for (int is = 0; is < c->src.n; is++) {
    if (c->src.D[is] != 2) continue;
    const int i = c->src.I[is], j = c->src.J[is];
    if (!(b.IstrR <= i && i <= b.IendR && b.JstrR <= j && j <= b.JendR)) continue;
    const long c0 = I2(i, j);
    zeta[c0] = zeta[c0] + c->src.Qbar[is] * c->F.pm[c0] * c->F.pn[c0] * c->p.dtfast;
  }
}

int step2d_impl(const roms_step_idx_t *si, bool in_loop)
{
  const roms_bounds_t &b = g_ctx.b;
  const roms_params_t &p = g_ctx.p;
  int rc;
  S2 s{si->krhs, si->kstp, si->knew, si->nstp, si->nnew, si->iif, si->iic, si->ntfirst, si->predictor_2d_step, 0};
  double *DUon = g_ctx.hostc.ws2[0], *DVom = g_ctx.hostc.ws2[1];
  double *zeta_new = g_ctx.hostc.ws2[2], *zwrk = g_ctx.hostc.ws2[3];
  const long nij = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1);
  // Source-mapped fast path: one tile, E-W periodic, closed N-S walls.  Not used on
  // the first predictor of a step: there the momentum kernel read-modify-writes
  // rufrc and ru(:,:,0,nstp) at the source points (:1884-2037), which ghost-point
  // threads would race with.
  // (the fused kernels apply the closed-wall conditions themselves; open S/N edges take the general path below,
  // whose boundary conditions are separate launches)
  const bool walls = lbc2d_all_closed();
  // (UV_VIS4: the biharmonic term is a pass of its own in front of the momentum kernel -- general path only)
  // (WET_DRY: the masks are a pass of their own between the averages and the free surface -- general path only)
  // (LuvSrc / LwSrc: the source faces and cells are launches of their own -- general path only)
  const bool srcs = (p.point_sources & 3) != 0;
  const bool sm = b.ntileI * b.ntileJ == 1 && b.EWperiodic && !b.NSperiodic && !g_ctx.loopback && walls && !p.uv_vis4 && !p.wet_dry && !srcs;
  if (sm) {
    if (s.iif <= p.nfast) {
      // ONE launch: free surface, fast-time averages and momentum (k2d_mom_lds<true>)
      s.sm = 2;
      return roms_launch_k2d_mom_lds((const int *)&s, nullptr, nullptr, nullptr, nullptr);
    }
    // the last predictor of the loop (iif = nfast+1) only finishes the fast-time averages (:614-682);
    // DUon/DVom are evaluated in place
    s.sm = 1;
    hipLaunchKernelGGL(k2d_zeta_sm, grid2d(b.UBi - b.LBi + 1, b.UBj - b.LBj + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s, (const double *)nullptr, (const double *)nullptr, zeta_new, zwrk);
    KERNEL_CHECK("k2d_zeta_sm");
    if (s.predictor) {
      if ((rc = halo_exchange2d(GT_R, g_ctx.dev[FID_Zt_avg1]))) return rc;
      if ((rc = halo_exchange2d(GT_U, g_ctx.dev[FID_DU_avg1]))) return rc;
      if ((rc = halo_exchange2d(GT_V, g_ctx.dev[FID_DV_avg1]))) return rc;
    }
    return 0;
  }
  // General path (several tiles; first predictor of a step on one tile).  Messages per call: inside
  // LOOP_2D ONE fused exchange at the end (rzeta, zeta, ubar, vbar of this call + DUon, DVom of the
  // next one, whose krhs is this call's knew); a stand-alone call exchanges its own fluxes first.
  const bool multi = b.ntileI * b.ntileJ > 1 || g_ctx.loopback;
  // DUon/DVom live in two scratch pairs: the fused kernel reads one (exchanged fluxes of this level)
  // while it writes the other (own-point fluxes of the next level)
  if (g_flux_buf) { DUon = g_ctx.hostc.ws2[4]; DVom = g_ctx.hostc.ws2[5]; }
  double *DUnext = g_flux_buf ? g_ctx.hostc.ws2[0] : g_ctx.hostc.ws2[4];
  double *DVnext = g_flux_buf ? g_ctx.hostc.ws2[1] : g_ctx.hostc.ws2[5];
  if (!(g_flux_ready && g_flux_lev == s.krhs)) {
    hipLaunchKernelGGL(k2d_flux, grid2d(b.Iendp2 - (b.IstrU - 2) + 1, b.Jendp2 - (b.JstrV - 2) + 1), block2d(), 0,
                       g_ctx.stream, g_ctx.devc, s, DUon, DVom);
    KERNEL_CHECK("k2d_flux");
    halo_batch_begin();
    halo_exchange2d(GT_U, DUon);
    halo_exchange2d(GT_V, DVom);
    if ((rc = halo_batch_end())) return rc;
  }
  g_flux_ready = false;
  if (in_loop && multi && walls && s.iif <= p.nfast && !p.uv_vis4 && !p.wet_dry && !srcs) {
    // ONE compute launch + ONE exchange per call
    s.sm = 3;
    if ((rc = roms_launch_k2d_mom_lds((const int *)&s, DUon, DVom, nullptr, nullptr, DUnext, DVnext))) return rc;
    halo_batch_begin();
    if (s.predictor) halo_exchange2d(GT_R, g_ctx.dev[FID_rzeta] + (long)(s.krhs - 1) * nij);
    halo_exchange2d(GT_R, g_ctx.dev[FID_zeta] + (long)(s.knew - 1) * nij);
    halo_exchange2d(GT_U, g_ctx.dev[FID_ubar] + (long)(s.knew - 1) * nij);
    halo_exchange2d(GT_V, g_ctx.dev[FID_vbar] + (long)(s.knew - 1) * nij);
    halo_exchange2d(GT_U, DUnext);
    halo_exchange2d(GT_V, DVnext);
    if ((rc = halo_batch_end())) return rc;
    g_flux_ready = true;
    g_flux_lev = s.knew;
    g_flux_buf ^= 1;
    return 0;
  }
  const int i0 = b.IstrR < b.IstrU - 1 ? b.IstrR : b.IstrU - 1, j0 = b.JstrR < b.JstrV - 1 ? b.JstrR : b.JstrV - 1;
  hipLaunchKernelGGL(k2d_zeta, grid2d(b.IendR - i0 + 1, b.JendR - j0 + 1), block2d(), 0, g_ctx.stream, g_ctx.devc, s,
                     (const double *)DUon, (const double *)DVom, zeta_new, zwrk);
  KERNEL_CHECK("k2d_zeta");
  if (s.iif == p.nfast + 1 && s.predictor) {
    halo_batch_begin();
    halo_exchange2d(GT_R, g_ctx.dev[FID_Zt_avg1]);
    halo_exchange2d(GT_U, g_ctx.dev[FID_DU_avg1]);
    halo_exchange2d(GT_V, g_ctx.dev[FID_DV_avg1]);
    if ((rc = halo_batch_end())) return rc;
  }
  // WET_DRY: the new wet/dry masks, :729-749 (after the averages and their exchange, before the return below)
  if (p.wet_dry && (rc = wetdry_launch(s.iif <= p.nfast ? 0 : 1, s.iif == 1 && s.predictor, s.kstp))) return rc;
  if (s.iif > p.nfast) return 0;
  if ((p.point_sources & 2) && g_ctx.hostc.src.n > 0) {     // LwSrc, :890-908
    hipLaunchKernelGGL(k2d_src_zeta, dim3(1), dim3(64), 0, g_ctx.stream, g_ctx.devc, s.knew);
    KERNEL_CHECK("k2d_src_zeta");
  }
  if ((rc = bc_zeta(s.knew, si))) return rc;
  if (p.uv_vis4 && (rc = roms_launch_step2d_visc4(s.krhs))) return rc;
  if ((rc = roms_launch_k2d_mom_lds((const int *)&s, DUon, DVom, zeta_new, zwrk))) return rc;
  if ((rc = bc_u2d(s.knew, si))) return rc;
  if ((rc = bc_v2d(s.knew, si))) return rc;
  if ((p.point_sources & 1) && g_ctx.hostc.src.n > 0) {        // LuvSrc, step2d_LF_AM3.h:2484-2502
    hipLaunchKernelGGL(k2d_src_bar, dim3((g_ctx.hostc.src.n + 63) / 64), dim3(64), 0, g_ctx.stream, g_ctx.devc, s.knew);
    KERNEL_CHECK("k2d_src_bar");
  }
  halo_batch_begin();
  if (s.predictor) halo_exchange2d(GT_R, g_ctx.dev[FID_rzeta] + (long)(s.krhs - 1) * nij);
  halo_exchange2d(GT_R, g_ctx.dev[FID_zeta] + (long)(s.knew - 1) * nij);
  halo_exchange2d(GT_U, g_ctx.dev[FID_ubar] + (long)(s.knew - 1) * nij);
  halo_exchange2d(GT_V, g_ctx.dev[FID_vbar] + (long)(s.knew - 1) * nij);
  return halo_batch_end();
}

}  // namespace

// wetdry(ng, tile, Tindex, .TRUE.): the masks of the initial state (initial.F:438-466), Tindex = s->kstp
extern "C" int roms_hip_wetdry(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_wetdry");
  if (rc) return rc;
  if (!g_ctx.p.wet_dry) return 0;
  if (s->kstp < 1 || s->kstp > 3) return roms_fail("roms_hip_wetdry", "kstp (the time index of the initial state) outside 1..3");
  ScopedTimer tm("wetdry");
  return wetdry_launch(2, 0, s->kstp);
}

extern "C" int roms_hip_step2d(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_step2d");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  if ((rc = roms_rowm_prepare())) return rc;
  ScopedTimer tm("step2d");
  g_flux_ready = false;
  return step2d_impl(s, false);
}

// LOOP_2D of main3d.F:592-700: the predictor/corrector sequencing, launches queued on the library's stream
static int step2d_loop_body(roms_step_idx_t *s, int *indx1)
{
  int rc;
  const int nfast = g_ctx.p.nfast;
  int predictor = 0;
  g_flux_ready = false;
  for (int my_iif = 1; my_iif <= nfast + 1; my_iif++) {
    const int next_indx1 = 3 - *indx1;
    if (!predictor && my_iif <= nfast + 1) {
      predictor = 1;
      s->iif = my_iif;
      s->kstp = (s->iif == 1) ? *indx1 : 3 - *indx1;
      s->knew = 3;
      s->krhs = *indx1;
    }
    s->predictor_2d_step = predictor;
    if ((rc = step2d_impl(s, true))) return rc;
    if (predictor) {
      predictor = 0;
      s->knew = next_indx1;
      s->kstp = 3 - s->knew;
      s->krhs = 3;
      if (s->iif < nfast + 1) *indx1 = next_indx1;
    }
    s->predictor_2d_step = predictor;
    if (s->iif < nfast + 1)
      if ((rc = step2d_impl(s, true))) return rc;
  }
  g_flux_ready = false;
  return 0;
}

// One tile: the 2*nfast+1 launches of the loop are captured once per (indx1, nstp, start-up phase) into a
// hipGraph and replayed -- the launch arguments of a replay are those of the capture, so the graphs are
// dropped whenever bounds, parameters or a field registration change (step2d_graphs_release).  On several
// tiles the loop contains host-side transport calls and runs eagerly.
struct LoopGraph { hipGraphExec_t exec; int indx1_out; roms_step_idx_t s_out; };
static std::map<int, LoopGraph> g_loop_graphs;
static int g_graph_exchanges = 0;            // roms_hip_graph_exchanges: 0 (default) never, 1 always, 2 in loopback only
static bool g_rccl_graph_failed = false;     // the capture of the transport failed once on this stack: stay eager
bool halo_rccl_capturable();                  // halo.hip: RCCL transport in use, message plan and buffers can be fixed
int halo_reserve_buffers();                   // halo.hip: allocate the message buffers for the largest exchange now

extern "C" int roms_hip_graph_exchanges(int on)
{
  g_graph_exchanges = on ? 1 : 0;
  step2d_graphs_release();
  return 0;
}
// 0 = LOOP_2D runs eagerly on several tiles, 1 = replayed as one hipGraph with the exchanges inside, -1 = the capture
// was tried and refused by this stack
extern "C" int roms_hip_graph_exchanges_state(void)
{
  if (g_rccl_graph_failed) return -1;
  return g_loop_graphs.empty() ? 0 : 1;
}

void step2d_graphs_release()
{
  g_rccl_graph_failed = false;
  for (auto &kv : g_loop_graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  g_loop_graphs.clear();
}

extern "C" int roms_hip_step2d_loop(roms_step_idx_t *s, int *indx1)
{
  int rc = roms_entry_check("roms_hip_step2d_loop");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  if ((rc = roms_rowm_prepare())) return rc;          // before any capture: it synchronises when it has work
  ScopedTimer tm("step2d_loop");
  const roms_bounds_t &b = g_ctx.b;
  const bool one_tile = b.ntileI * b.ntileJ == 1 && !g_ctx.loopback;
  // Several tiles over RCCL: the loop's launches AND its ncclSend / ncclRecv groups can be captured too (RCCL enqueues
  // its kernels on the capturing stream), so that a replay is one hipGraphLaunch for 59 compute launches, 59 packs,
  // 59 transport kernels and 59 unpacks.  Opt-in (roms_hip_graph_exchanges): measured in loopback on BENCHMARK1
  // (ROCm 7.2, RCCL of this image) the loop's device time drops from 1.85 to 1.65 ms but the step's wall time RISES
  // from 3.05 to 3.64 ms -- replaying what RCCL records besides its kernels costs more on the host than the eager
  // calls.  Every rank must capture and replay the same sequence.  If the capture fails the loop runs eagerly.
  const bool rccl_graph = !one_tile && halo_rccl_capturable() && !g_rccl_graph_failed &&
                          (g_graph_exchanges == 1 || (g_graph_exchanges == 2 && g_ctx.loopback));
  if (!one_tile && !rccl_graph) return step2d_loop_body(s, indx1);
  const int phase = s->iic == s->ntfirst ? 0 : (s->iic == s->ntfirst + 1 ? 1 : 2);
  const int key = ((*indx1 * 4 + s->nstp) * 4 + s->nnew) * 4 + phase;
  auto it = g_loop_graphs.find(key);
  if (it == g_loop_graphs.end()) {
    hipGraph_t graph = nullptr;
    const roms_step_idx_t s_in = *s;
    const int indx1_in = *indx1;
    if (rccl_graph && (rc = halo_reserve_buffers())) return rc;       // no allocation inside a capture
    HIP_TRY(hipStreamBeginCapture(g_ctx.stream, hipStreamCaptureModeThreadLocal));
    rc = step2d_loop_body(s, indx1);
    const hipError_t e = hipStreamEndCapture(g_ctx.stream, &graph);
    if (rccl_graph && (rc || e != hipSuccess)) {
      // this stack does not capture the transport: remember it and run the loop the ordinary way
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      g_rccl_graph_failed = true;
      *s = s_in;
      *indx1 = indx1_in;
      return step2d_loop_body(s, indx1);
    }
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return roms_fail("roms_hip_step2d_loop: hipStreamEndCapture", hipGetErrorString(e));
    LoopGraph lg{nullptr, *indx1, *s};
    const hipError_t e2 = hipGraphInstantiate(&lg.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) return roms_fail("roms_hip_step2d_loop: hipGraphInstantiate", hipGetErrorString(e2));
    it = g_loop_graphs.emplace(key, lg).first;
  } else {
    // the sequencing of the loop does not depend on the data: indices as the capture left them
    const int iic = s->iic, ntfirst = s->ntfirst;
    *s = it->second.s_out;
    s->iic = iic;
    s->ntfirst = ntfirst;
    *indx1 = it->second.indx1_out;
  }
  HIP_TRY(hipGraphLaunch(it->second.exec, g_ctx.stream));
  return 0;
}
