// k_step3d_t.hip -- corrector time-step for tracers, step3d_t_tile
// (ROMS/Nonlinear/step3d_t.F:108-1682), as ONE fused kernel per tracer:
//
//   horizontal advection (C2/U3/A4/C4, :363-880) -> vertical advection
//   (C2/A4/C4/splines, :883-1210) -> t*oHz -> implicit vertical diffusion in
//   spline form (:1363-1455), per water column.
//
// Mapping: one thread per (i,j) water column, 64 consecutive i per wavefront,
// so every level of every field is read as contiguous 512-byte lines.  The
// column sweeps upward once (fluxes, advective update, Thomas forward
// elimination fused level by level) and downward once (back substitution +
// final update).  The column state the Thomas algorithm needs (post-advection
// tracer, modified super-diagonal CF, right-hand side DC) stays in VGPRs
// (fully unrolled, template on the maximum N), so t(:,:,:,nnew,itrc) is read
// once and written once.
//
// Algorithmic HBM traffic per cell and tracer: read t(..,3,itrc), read+write
// t(..,nnew,itrc), read Akt = 32 B, plus Huon,Hvom,W,Hz = 32 B shared by the
// tracers (8*(4*NT+4) B per cell).
//
// The arithmetic order inside every expression is the reference's (the build
// uses -ffp-contract=off), so results agree with the CPU restatement to the
// last bit except where a library function differs.
#include "roms_dev.h"
#include <cstdlib>

int roms_entry_check(const char *name);
int roms_launch_step3d_t_mpdata(int nnew, int itrc0, int n);         // k_mpdata.hip

#include "advect.h"

namespace {

// k_step3d_t<> below: one thread per column reading straight from global memory; instantiated for the
// HSIMT pair only (every other scheme pair runs k_step3d_t_pipe).
//
// Tuning notes (MI355X, BENCHMARK3, profiles/r01c): the kernel is bound by memory LATENCY
// per level, not by bytes -- FETCH_SIZE barely moves the time.  Issuing every load of a
// level before the first use took 0.59 -> 0.53 ms.  Taller workgroups (64x8) changed
// neither traffic nor time (j-halo rows are L2 hits).  A compile-time N (no guards in the
// unrolled loop) let the scheduler hoist loads of many levels at once and was 2-3x SLOWER
// (occupancy 1 or scratch spills), so the run-time guards stay.  Keeping 1/Hz in LDS for the
// downward sweep saved 0.15 GB of fetch and 0.03 ms; the pipelined kernel uses the LDS for
// tn() instead.  XCD strips (roms_dev.h) cut the fetch from 2.1 to 1.8 GB.
// ---------------------------------------------------------------------------
// HSIMT (Wu and Zhu, 2010) with the TVD limiter, step3d_t.F:430-590 (horizontal) and :1022-1090
// (vertical).  Evaluated straight from global memory by the classic kernel: an optional scheme that
// none of the five configurations uses, kept simple.  A face is named by the cell on its high side.
// ---------------------------------------------------------------------------
#define HS_EPS1 1.0E-12
__device__ __forceinline__ double hsimt_limited(double g0, double gn, double k0, double kn, double ok0)
{
  // cff of :473-489 for the upwind neighbour face (gn, kn): 0.5*MAX(0,MIN(2, 2 r rka, beta))*grad*Ka
  const double cc1 = 0.25, cc2 = 0.5, cc3 = 1.0 / 12.0;
  double r, rka;
  if (fabs(g0) <= HS_EPS1) { r = 0.0; rka = 0.0; }
  else { r = gn / g0; rka = kn * ok0; }
  const double a1 = cc1 * k0 + cc2 - cc3 * ok0;
  const double b1 = -cc1 * k0 + cc2 + cc3 * ok0;
  const double beta = a1 + b1 * r;
  return 0.5 * fmax(0.0, fmin(fmin(2.0, 2.0 * r * rka), beta)) * g0 * k0;
}
// horizontal face between cell (a - off) and cell a; off = 1 (xi, Huon) or ni (eta, Hvom).
// lo_zero / hi_zero: the wall rule of :451-464 / :527-540 applies to the neighbour face used.
// MASKING (fmask = umask / vmask of the direction, null without the option): the differences and Ka times the
// mask of their face (:448, :523) and the limited correction times rmask two cells upstream of the face,
// rmask(MAX(i-2,0)) or rmask(MIN(i+1,Lm+1)) (:487, :506, :562, :581); idx = the face's global i (j), idxmax = Lm+1 (Mm+1).
template <int DIR>
__device__ __forceinline__ double hsimt_hface(gcd_t t3, gcd_t H, gcd_t Hz, gcd_t pm, gcd_t pn, long a, long a2, long off,
                                              double dt, bool lo_zero, bool hi_zero, gcd_t fmask, gcd_t rmask, int idx,
                                              int idxmax)
{
  auto grad = [&](long x, long x2) {
    const double g = t3[x] - t3[x - off];
    return fmask ? g * fmask[x2] : g;
  };
  auto Ka = [&](long x, long x2) {
    double cff;
    if constexpr (DIR == 0) cff = 0.125 * (pm[x2 - off] + pm[x2]) * (pn[x2 - off] + pn[x2]) * dt;
    else cff = 0.125 * (pn[x2] + pn[x2 - off]) * (pm[x2] + pm[x2 - off]) * dt;
    double cff1;
    if constexpr (DIR == 0) cff1 = cff * (1.0 / Hz[x - off] + 1.0 / Hz[x]);
    else cff1 = cff * (1.0 / Hz[x] + 1.0 / Hz[x - off]);
    const double ka = 1.0 - fabs(H[x] * cff1);
    return fmask ? ka * fmask[x2] : ka;
  };
  const double Hf = H[a];
  const double g0 = grad(a, a2), k0 = Ka(a, a2);
  const double ok0 = (k0 <= HS_EPS1) ? 0.0 : 1.0 / fmax(k0, HS_EPS1);
  double sw;
  if (Hf >= 0.0) {
    const double gn = lo_zero ? 0.0 : grad(a - off, a2 - off), kn = lo_zero ? 0.0 : Ka(a - off, a2 - off);
    double cff = hsimt_limited(g0, gn, k0, kn, ok0);
    if (fmask) cff = cff * rmask[a2 + (long)((idx - 2 > 0 ? idx - 2 : 0) - idx) * off];
    sw = t3[a - off] + cff;
  } else {
    const double gn = hi_zero ? 0.0 : grad(a + off, a2 + off), kn = hi_zero ? 0.0 : Ka(a + off, a2 + off);
    double cff = hsimt_limited(g0, gn, k0, kn, ok0);
    if (fmask) cff = cff * rmask[a2 + (long)((idx + 1 < idxmax ? idx + 1 : idxmax) - idx) * off];
    sw = t3[a] - cff;
  }
  return sw * Hf;
}
// vertical flux through the top face of level k (W-level k), k = 1..N-1; a = index of (i,j,k) in rho arrays
__device__ __forceinline__ double hsimt_vface(gcd_t t3, gcd_t Wv, gcd_t z_r, long a, long nij, int k, int N, double cff)
{
  const double Wk = Wv[a + nij];
  if (k == 1 && Wk >= 0.0) return Wk * t3[a];
  if (k == N - 1 && Wk < 0.0) return Wk * t3[a + nij];
  // KaZ, gradZ at W-level q (1..N-1; zero at 0 and N), x = rho index of level q
  auto KaZ = [&](int q, long x) { return (q < 1 || q > N - 1) ? 0.0 : 1.0 - fabs(cff * Wv[x + nij] / (z_r[x + nij] - z_r[x])); };
  auto gradZ = [&](int q, long x) { return (q < 1 || q > N - 1) ? 0.0 : t3[x + nij] - t3[x]; };
  const double k0 = KaZ(k, a), g0 = gradZ(k, a);
  const double ok0 = 1.0 / k0;
  double sw;
  if (Wk >= 0) sw = t3[a] + hsimt_limited(g0, gradZ(k - 1, a - nij), k0, KaZ(k - 1, a - nij), ok0);
  else sw = t3[a + nij] - hsimt_limited(g0, gradZ(k + 1, a + nij), k0, KaZ(k + 1, a + nij), ok0);
  return Wk * sw;
}

template <int HADV, int VADV, int NMAX>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_step3d_t(const RomsDev *__restrict__ c, int nnew, int itrc0, int ntr)
{
  DEV_PROLOGUE(c)
  const TileTr tt = decode_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, ntr);
  if (!tt.valid) return;
  const int i = b.Istr + tt.bx * BLK_X + threadIdx.x;
  const int j = b.Jstr + tt.by * BLK_Y + threadIdx.y;
  const int itrc = itrc0 + tt.itr;              // 1-based tracer index
  if (i > b.Iend || j > b.Jend) return;
  const int ltrc = itrc < b.NAT ? itrc : b.NAT;
  const double dt = c->p.dt;
  const gcd_t t3 = (gcd_t)(c->F.t + (2L + 3L * (itrc - 1)) * n3r);
  const gd_t tn_g = (gd_t)(c->F.t + ((long)(nnew - 1) + 3L * (itrc - 1)) * n3r);
  const gcd_t Huon = (gcd_t)c->F.Huon;
  const gcd_t Hvom = (gcd_t)c->F.Hvom;
  const gcd_t Wv = (gcd_t)c->F.W;
  const gcd_t Hz = (gcd_t)c->F.Hz;
  const gcd_t Akt = (gcd_t)(c->F.Akt + (long)(ltrc - 1) * n3w);
  const double cffdt = dt * GF(pm)[I2(i, j)] * GF(pn)[I2(i, j)];
  const bool s_wall = b.south_edge && !b.NSperiodic && j == b.Jstr;       // FE(Jstr-1)=FE(Jstr)
  const bool n_wall = b.north_edge && !b.NSperiodic && j == b.Jend;       // FE(Jend+2)=FE(Jend+1)
  const bool n_wall1 = b.north_edge && !b.NSperiodic && j == b.Jend - 1;
  (void)n_wall1;
  const long c0 = I2(i, j);
  // wall rows: the outer stencil point is not used (:741-760); read a valid address instead
  const long oym2 = s_wall ? 0 : -2 * ni, oyp2 = n_wall ? 0 : 2 * ni;
  // physical western / eastern edges (:700-715: FX(Istr-1) = FX(Istr), FX(Iend+2) = FX(Iend+1))
  const bool w_wall = b.west_edge && !b.EWperiodic && i == b.Istr;
  const bool e_wall = b.east_edge && !b.EWperiodic && i == b.Iend;
  const long oxm2 = w_wall ? 0 : -2, oxp2 = e_wall ? 0 : 2;

  const bool src_cell = c->src.n > 0 && src_cell_any(c, c0, ni);      // LuvSrc: a face of this cell is a source face

  double tn[NMAX + 1], CF[NMAX + 1], DC[NMAX + 1];
  CF[0] = 0.0;
  DC[0] = 0.0;

  // A4 vertical: harmonic-mean slopes need the whole column of differences
  double a4cf[(VADV == ADV_A4) ? NMAX + 2 : 1];
  double spl[(VADV == ADV_SPLINES) ? NMAX + 1 : 1];
  if constexpr (VADV == ADV_A4) {
    const double eps = 1.0E-16;
    double dprev = 0.0, tk = t3[c0];
#pragma unroll
    for (int k = 1; k <= NMAX; k++) {
      if (k <= N) {
        double dk;                                 // FC(k) = t(k+1)-t(k), FC(N)=FC(N-1), FC(0)=FC(1)
        if (k < N) { const double tk1 = t3[c0 + (long)k * nij]; dk = tk1 - tk; tk = tk1; }
        else dk = dprev;
        if (k == 1) dprev = dk;
        const double cff = 2.0 * dk * dprev;
        a4cf[k] = (cff > eps) ? cff / (dk + dprev) : 0.0;
        dprev = dk;
      }
    }
  }
  if constexpr (VADV == ADV_SPLINES) {
    // parabolic-spline reconstruction of t at W-points, step3d_t.F:894-930
    double cfs[NMAX + 1];
    spl[0] = 2.0 * t3[c0];
    cfs[1] = 1.0;
#pragma unroll
    for (int k = 1; k < NMAX; k++) {
      if (k <= N - 1) {
        const double hk = Hz[c0 + (long)(k - 1) * nij], hk1 = Hz[c0 + (long)k * nij];
        const double cff = 1.0 / (2.0 * hk + hk1 * (2.0 - cfs[k]));
        cfs[k + 1] = cff * hk;
        spl[k] = cff * (3.0 * (hk * t3[c0 + (long)k * nij] + hk1 * t3[c0 + (long)(k - 1) * nij]) - hk1 * spl[k - 1]);
      }
    }
    double top = 0.0;
#pragma unroll
    for (int k = 1; k <= NMAX; k++)
      if (k == N) { top = (2.0 * t3[c0 + (long)(N - 1) * nij] - spl[k - 1]) / (1.0 - cfs[k]); spl[k] = top; }
#pragma unroll
    for (int k = NMAX - 1; k >= 0; k--) {
      if (k <= N - 1) {
        spl[k] = spl[k] - cfs[k + 1] * spl[k + 1];
        spl[k + 1] = Wv[c0 + (long)(k + 1) * nij] * spl[k + 1];
      }
    }
#pragma unroll
    for (int k = 0; k <= NMAX; k++) if (k == 0 || k == N) spl[k] = 0.0;
  }

  // sliding windows along k
  double tkm1 = 0.0, tk = t3[c0], tkp1 = (N >= 2) ? t3[c0 + nij] : 0.0, tkp2;
  double FCprev = 0.0;
  double hz_m1 = 0.0, ohz_m1 = 0.0, akt_m2 = 0.0, akt_m1 = Akt[c0];   // Akt(k-1) for k=1 is Akt(0)

#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      const long ck = c0 + (long)(k - 1) * nij;
      // every load of the level is issued up front (one memory round trip per level)
      const double hz = Hz[ck];
      double tv = tn_g[ck];
      const double akt_0 = Akt[ck + nij];          // Akt(i,j,k)
      const double wtop = Wv[ck + nij];
      tkp2 = (k + 2 <= N) ? t3[ck + 2 * nij] : 0.0;
      // ---- horizontal fluxes, step3d_t.F:596-828 ----
      const double xm2 = t3[ck + oxm2], xm1 = t3[ck - 1], xp1 = t3[ck + 1], xp2 = t3[ck + oxp2];
      const double ym1 = t3[ck - ni], yp1 = t3[ck + ni];
      const double ym2 = t3[ck + oym2];
      const double yp2 = t3[ck + oyp2];
      const double hu0 = Huon[ck], hu1 = Huon[ck + 1];
      const double hv0 = Hvom[ck], hv1 = Hvom[ck + ni];
      const double dx0 = tk - xm1, dxp1 = xp1 - tk;
      const double dxm1 = w_wall ? dx0 : (xm1 - xm2), dxp2 = e_wall ? dxp1 : (xp2 - xp1);
      const double dy0 = tk - ym1, dyp1 = yp1 - tk;
      const double dym1 = s_wall ? dy0 : (ym1 - ym2);
      const double dyp2 = n_wall ? dyp1 : (yp2 - yp1);
      double FXi, FXip1, FEj, FEjp1;
      if constexpr (HADV == ADV_HSIMT) {
        const gcd_t pmg = (gcd_t)c->F.pm, png = (gcd_t)c->F.pn;
        const bool mk = c->p.masking != 0;
        const gcd_t um = mk ? (gcd_t)c->F.umask : (gcd_t) nullptr, vm = mk ? (gcd_t)c->F.vmask : (gcd_t) nullptr;
        const gcd_t rm = (gcd_t)c->F.rmask;
        // physical edges: the face outside Istr / Jstr (Iend+1 / Jend+1) enters only through its zeroed gradient,
        // :451-464, :527-540
        FXi = hsimt_hface<0>(t3, Huon, Hz, pmg, png, ck, c0, 1, dt, w_wall, false, um, rm, i, b.Lm + 1);
        FXip1 = hsimt_hface<0>(t3, Huon, Hz, pmg, png, ck + 1, c0 + 1, 1, dt, false, e_wall, um, rm, i + 1, b.Lm + 1);
        FEj = hsimt_hface<1>(t3, Hvom, Hz, pmg, png, ck, c0, ni, dt, s_wall, false, vm, rm, j, b.Mm + 1);
        FEjp1 = hsimt_hface<1>(t3, Hvom, Hz, pmg, png, ck + ni, c0 + ni, ni, dt, false, n_wall, vm, rm, j + 1, b.Mm + 1);
      } else {
        FXi = hflux<HADV>(hu0, xm1, tk, dxm1, dx0, dxp1);
        FXip1 = hflux<HADV>(hu1, tk, xp1, dx0, dxp1, dxp2);
        FEj = hflux<HADV>(hv0, ym1, tk, dym1, dy0, dyp1);
        FEjp1 = hflux<HADV>(hv1, tk, yp1, dy0, dyp1, dyp2);
      }
      if (src_cell)                                  // LuvSrc, step3d_t.F:734-799
        src_cell_fluxes<false>(c, c0, ck, ni, k, itrc, c->F.t + (2L + 3L * (itrc - 1)) * n3r, FXi, FXip1, FEj, FEjp1);
      // ---- vertical flux through the top face of level k ----
      double FCk;
      if (k == N) FCk = 0.0;
      else if constexpr (VADV == ADV_SPLINES) FCk = spl[k];
      else if constexpr (VADV == ADV_HSIMT)      // cff = pm*pn*dt in this order, :1032
        FCk = hsimt_vface(t3, Wv, (gcd_t)c->F.z_r, ck, nij, k, N, GF(pm)[c0] * GF(pn)[c0] * dt);
      else {
        double cfk = 0.0, cfk1 = 0.0;
        if constexpr (VADV == ADV_A4) { cfk = a4cf[k]; cfk1 = a4cf[k + 1]; }
        FCk = vflux<VADV>(k, N, wtop, tkm1, tk, tkp1, tkp2, cfk, cfk1);
      }
      // ---- advective update, step3d_t.F:857-875 and :1168-1208 ----
      const double ohz = 1.0 / hz;
      {
        const double cff1 = cffdt * (FXip1 - FXi);
        const double cff2 = cffdt * (FEjp1 - FEj);
        const double cff3 = cff1 + cff2;
        tv = tv - cff3;
      }
      tv = tv - cffdt * (FCk - FCprev);
      tv = tv * ohz;
      if (src_cell) tv = src_w_tracer(c, c0, k, itrc, cffdt * ohz, tk, tv);       // LwSrc, step3d_t.F:1331-1360
      tn[k] = tv;
      FCprev = FCk;
      // ---- Thomas forward elimination for row kk = k-1, step3d_t.F:1376-1410 ----
      if (k >= 2) {
        const double cff6 = 1.0 / 6.0, cff3r = 1.0 / 3.0;
        const double fc = cff6 * hz_m1 - dt * akt_m2 * ohz_m1;
        const double cf = cff6 * hz - dt * akt_0 * ohz;
        const double bc = cff3r * (hz_m1 + hz) + dt * akt_m1 * (ohz_m1 + ohz);
        const double cff = 1.0 / (bc - fc * CF[k - 2]);
        CF[k - 1] = cff * cf;
        DC[k - 1] = cff * (tn[k] - tn[k - 1] - fc * DC[k - 2]);
      }
      hz_m1 = hz; ohz_m1 = ohz; akt_m2 = akt_m1; akt_m1 = akt_0;
      tkm1 = tk; tk = tkp1; tkp1 = tkp2;
    }
  }

  // ---- back substitution + final update, step3d_t.F:1411-1455 ----
  double dcA_up = 0.0;      // DC(N)*Akt(N) = 0
  double dc_up = 0.0;       // DC(N) = 0
#pragma unroll
  for (int kk = NMAX - 1; kk >= 0; kk--) {
    if (kk <= N - 1) {
      double dcA = 0.0;
      if (kk >= 1) {
        const double dc = DC[kk] - CF[kk] * dc_up;
        dc_up = dc;
        dcA = dc * Akt[c0 + (long)kk * nij];
      }
      const long ck = c0 + (long)kk * nij;        // level kk+1
      const double ohz = 1.0 / Hz[ck];
      const double cff1 = dt * ohz * (dcA_up - dcA);
      double tv = tn[kk + 1] + cff1;
      if constexpr (HADV == ADV_HSIMT)          // (the only pair this kernel is instantiated for) step3d_t.F:1586-1596
        if (c->p.masking) tv = tv * GF(rmask)[c0];
      tn_g[ck] = tv;
      dcA_up = dcA;
    }
  }
}

// ---------------------------------------------------------------------------
// Software-pipelined variant.  The post-advection tracer column tn() lives in LDS
// ([level][thread], conflict-free), which frees ~60 VGPRs; they hold the 17 loads of level
// k+1, issued BEFORE level k is computed, so one memory round trip overlaps a whole level
// of arithmetic (and the other wave of the SIMD).  The downward sweep prefetches Akt and
// Hz two levels ahead.  Arithmetic is identical to k_step3d_t.
// ---------------------------------------------------------------------------
struct LevelIn {
  double tkp2, xm2, xm1, xp1, xp2, ym2, ym1, yp1, yp2, hu0, hu1, hv0, hv1, w, hz, tv, akt;
};

// MASK (MASKING applications, a second instantiation so that the unmasked kernel keeps its registers): the
// first differences are multiplied by umask / vmask of their face (step3d_t.F:603, :667) and the result by
// rmask (:1586-1596); the masks are read per level (cache hits) rather than held in registers.
// SRC (LuvSrc, point sources): the cells with a source face are a handful; the plain instantiation skips them (one
// look at the face maps per column, and only when there is a table) and the SRC instantiation -- the same text plus the
// replacement of the source faces' fluxes, step3d_t.F:734-799 -- runs over the list of those cells, grid =
// (cells / 256, tracers of the launch).  The threads of the kernel do not cooperate, so any cell-to-thread map will do.
// SPL = false (without SPLINES_VDIFF, step3d_t.F:1431-1501; 3 of the reference's 31 three-dimensional applications): the
// advected tracer stays thickness-weighted (:1196-1198 is not compiled) and the implicit vertical diffusion is a
// tridiagonal system for the tracer itself with the layer distances from z_r -- solved after the level loop from the
// column in LDS.  Instantiated at NMAX = 64 only (one kernel serves every N, not tuned).
template <int HADV, int VADV, int NMAX, bool MASK, bool SRC = false, bool SPL = true>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_step3d_t_pipe(const RomsDev *__restrict__ c, int nnew, int itrc0, int ntr)
{
  DEV_PROLOGUE(c)
  constexpr int NTH = BLK_X * BLK_Y;
  __shared__ double s_tn[NMAX * NTH];
  const int tid = threadIdx.y * BLK_X + threadIdx.x;
  int i, j, itrc;
  if constexpr (SRC) {
    const int q = blockIdx.x * NTH + tid;
    if (q >= c->src.ncell) return;
    const int cell = c->src.cells[q];
    i = LBi + (int)(cell % ni);
    j = LBj + (int)(cell / ni);
    itrc = itrc0 + (int)blockIdx.y;
  } else {
    const TileTr tt = decode_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, ntr);
    if (!tt.valid) return;
    i = b.Istr + tt.bx * BLK_X + threadIdx.x;
    j = b.Jstr + tt.by * BLK_Y + threadIdx.y;
    itrc = itrc0 + tt.itr;
    if (i > b.Iend || j > b.Jend) return;
    if (c->src.n > 0 && src_cell_any(c, I2(i, j), ni)) return;      // the SRC launch steps this column
  }
  const int ltrc = itrc < b.NAT ? itrc : b.NAT;
  const double dt = c->p.dt;
  const gcd_t t3 = (gcd_t)(c->F.t + (2L + 3L * (itrc - 1)) * n3r);
  const gd_t tn_g = (gd_t)(c->F.t + ((long)(nnew - 1) + 3L * (itrc - 1)) * n3r);
  const gcd_t Huon = (gcd_t)c->F.Huon;
  const gcd_t Hvom = (gcd_t)c->F.Hvom;
  const gcd_t Wv = (gcd_t)c->F.W;
  const gcd_t Hz = (gcd_t)c->F.Hz;
  const gcd_t Akt = (gcd_t)(c->F.Akt + (long)(ltrc - 1) * n3w);
  const double cffdt = dt * GF(pm)[I2(i, j)] * GF(pn)[I2(i, j)];
  const bool s_wall = b.south_edge && !b.NSperiodic && j == b.Jstr;
  const bool n_wall = b.north_edge && !b.NSperiodic && j == b.Jend;
  const long c0 = I2(i, j);
  const long oym2 = s_wall ? 0 : -2 * ni, oyp2 = n_wall ? 0 : 2 * ni;
  // physical western / eastern edges (step3d_t.F:700-715: FX(Istr-1) = FX(Istr), FX(Iend+2) = FX(Iend+1))
  const bool w_wall = b.west_edge && !b.EWperiodic && i == b.Istr;
  const bool e_wall = b.east_edge && !b.EWperiodic && i == b.Iend;
  const long oxm2 = w_wall ? 0 : -2, oxp2 = e_wall ? 0 : 2;

  double CF[NMAX + 1], DC[NMAX + 1];
  CF[0] = 0.0;
  DC[0] = 0.0;

  double a4cf[(VADV == ADV_A4) ? NMAX + 2 : 1];
  double spl[(VADV == ADV_SPLINES) ? NMAX + 1 : 1];
  if constexpr (VADV == ADV_A4) {
    const double eps = 1.0E-16;
    double dprev = 0.0, tk = t3[c0];
#pragma unroll
    for (int k = 1; k <= NMAX; k++) {
      if (k <= N) {
        double dk;
        if (k < N) { const double tk1 = t3[c0 + (long)k * nij]; dk = tk1 - tk; tk = tk1; }
        else dk = dprev;
        if (k == 1) dprev = dk;
        const double cff = 2.0 * dk * dprev;
        a4cf[k] = (cff > eps) ? cff / (dk + dprev) : 0.0;
        dprev = dk;
      }
    }
  }
  if constexpr (VADV == ADV_SPLINES) {
    double cfs[NMAX + 1];
    spl[0] = 2.0 * t3[c0];
    cfs[1] = 1.0;
#pragma unroll
    for (int k = 1; k < NMAX; k++) {
      if (k <= N - 1) {
        const double hk = Hz[c0 + (long)(k - 1) * nij], hk1 = Hz[c0 + (long)k * nij];
        const double cff = 1.0 / (2.0 * hk + hk1 * (2.0 - cfs[k]));
        cfs[k + 1] = cff * hk;
        spl[k] = cff * (3.0 * (hk * t3[c0 + (long)k * nij] + hk1 * t3[c0 + (long)(k - 1) * nij]) - hk1 * spl[k - 1]);
      }
    }
    double top = 0.0;
#pragma unroll
    for (int k = 1; k <= NMAX; k++)
      if (k == N) { top = (2.0 * t3[c0 + (long)(N - 1) * nij] - spl[k - 1]) / (1.0 - cfs[k]); spl[k] = top; }
#pragma unroll
    for (int k = NMAX - 1; k >= 0; k--) {
      if (k <= N - 1) {
        spl[k] = spl[k] - cfs[k + 1] * spl[k + 1];
        spl[k + 1] = Wv[c0 + (long)(k + 1) * nij] * spl[k + 1];
      }
    }
#pragma unroll
    for (int k = 0; k <= NMAX; k++) if (k == 0 || k == N) spl[k] = 0.0;
  }

  auto load_level = [&](int k) {
    LevelIn L;
    const long ck = c0 + (long)(k - 1) * nij;
    L.hz = Hz[ck];
    L.tv = tn_g[ck];
    L.akt = Akt[ck + nij];
    L.w = Wv[ck + nij];
    L.tkp2 = (k + 2 <= N) ? t3[ck + 2 * nij] : 0.0;
    L.xm2 = t3[ck + oxm2]; L.xm1 = t3[ck - 1]; L.xp1 = t3[ck + 1]; L.xp2 = t3[ck + oxp2];
    L.ym1 = t3[ck - ni]; L.yp1 = t3[ck + ni];
    L.ym2 = t3[ck + oym2];
    L.yp2 = t3[ck + oyp2];
    L.hu0 = Huon[ck]; L.hu1 = Huon[ck + 1];
    L.hv0 = Hvom[ck]; L.hv1 = Hvom[ck + ni];
    return L;
  };

  double tkm1 = 0.0, tk = t3[c0], tkp1 = (N >= 2) ? t3[c0 + nij] : 0.0;
  double FCprev = 0.0, tn_prev = 0.0;
  double hz_m1 = 0.0, ohz_m1 = 0.0, akt_m2 = 0.0, akt_m1 = Akt[c0];
  LevelIn cur = load_level(1);

#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      LevelIn nxt;
      if (k + 1 <= N) nxt = load_level(k + 1);       // in flight while level k is computed
      const double tkp2 = cur.tkp2;
      double dxm1 = cur.xm1 - cur.xm2, dx0 = tk - cur.xm1, dxp1 = cur.xp1 - tk, dxp2 = cur.xp2 - cur.xp1;
      double dy0 = tk - cur.ym1, dyp1 = cur.yp1 - tk;
      double dym1 = cur.ym1 - cur.ym2, dyp2 = cur.yp2 - cur.yp1;
      if constexpr (MASK) {
        const gcd_t um = (gcd_t)c->F.umask, vm = (gcd_t)c->F.vmask;
        dxm1 = dxm1 * um[c0 + (w_wall ? 0 : -1)]; dx0 = dx0 * um[c0]; dxp1 = dxp1 * um[c0 + 1];
        dxp2 = dxp2 * um[c0 + (e_wall ? 1 : 2)];
        dy0 = dy0 * vm[c0]; dyp1 = dyp1 * vm[c0 + ni];
        dym1 = dym1 * vm[c0 + (s_wall ? 0 : -ni)]; dyp2 = dyp2 * vm[c0 + (n_wall ? ni : 2 * ni)];
      }
      if (s_wall) dym1 = dy0;
      if (n_wall) dyp2 = dyp1;
      if (w_wall) dxm1 = dx0;
      if (e_wall) dxp2 = dxp1;
      double FXi = hflux<HADV>(cur.hu0, cur.xm1, tk, dxm1, dx0, dxp1);
      double FXip1 = hflux<HADV>(cur.hu1, tk, cur.xp1, dx0, dxp1, dxp2);
      double FEj = hflux<HADV>(cur.hv0, cur.ym1, tk, dym1, dy0, dyp1);
      double FEjp1 = hflux<HADV>(cur.hv1, tk, cur.yp1, dy0, dyp1, dyp2);
      if constexpr (SRC)                             // LuvSrc, step3d_t.F:734-799
        src_cell_fluxes<false>(c, c0, c0 + (long)(k - 1) * nij, ni, k, itrc, c->F.t + (2L + 3L * (itrc - 1)) * n3r, FXi,
                               FXip1, FEj, FEjp1);
      double FCk;
      if (k == N) FCk = 0.0;
      else if constexpr (VADV == ADV_SPLINES) FCk = spl[k];
      else {
        double cfk = 0.0, cfk1 = 0.0;
        if constexpr (VADV == ADV_A4) { cfk = a4cf[k]; cfk1 = a4cf[k + 1]; }
        FCk = vflux<VADV>(k, N, cur.w, tkm1, tk, tkp1, tkp2, cfk, cfk1);
      }
      const double hz = cur.hz;
      const double ohz = 1.0 / hz;
      double tv = cur.tv;
      {
        const double cff1 = cffdt * (FXip1 - FXi);
        const double cff2 = cffdt * (FEjp1 - FEj);
        const double cff3 = cff1 + cff2;
        tv = tv - cff3;
      }
      tv = tv - cffdt * (FCk - FCprev);
      if constexpr (SPL) tv = tv * ohz;
      if constexpr (SRC) tv = src_w_tracer(c, c0, k, itrc, SPL ? cffdt * ohz : cffdt, tk, tv);   // LwSrc, step3d_t.F:1331-1360
      s_tn[(k - 1) * NTH + tid] = tv;
      FCprev = FCk;
      const double akt_0 = cur.akt;
      if (SPL && k >= 2) {
        const double cff6 = 1.0 / 6.0, cff3r = 1.0 / 3.0;
        const double fc = cff6 * hz_m1 - dt * akt_m2 * ohz_m1;
        const double cf = cff6 * hz - dt * akt_0 * ohz;
        const double bc = cff3r * (hz_m1 + hz) + dt * akt_m1 * (ohz_m1 + ohz);
        const double cff = 1.0 / (bc - fc * CF[k - 2]);
        CF[k - 1] = cff * cf;
        DC[k - 1] = cff * (tv - tn_prev - fc * DC[k - 2]);
      }
      tn_prev = tv;
      hz_m1 = hz; ohz_m1 = ohz; akt_m2 = akt_m1; akt_m1 = akt_0;
      tkm1 = tk; tk = tkp1; tkp1 = tkp2;
      cur = nxt;
    }
  }

  if constexpr (!SPL) {
    // the tridiagonal system of step3d_t.F:1431-1501: FC(k) = -dt lambda Akt(k) / (z_r(k+1) - z_r(k)),
    // BC(k) = Hz(k) - FC(k) - FC(k-1), right-hand side = the advected, thickness-weighted tracer
    const gcd_t z_r = (gcd_t)c->F.z_r;
    const double cfl = -dt * c->p.lambda;
    double fc_prev = 0.0;
#pragma unroll
    for (int k = 1; k <= NMAX; k++) {
      if (k <= N) {
        const long ck = c0 + (long)(k - 1) * nij;
        double fc = 0.0;
        if (k < N) {
          const double cff1 = 1.0 / (z_r[ck + nij] - z_r[ck]);
          fc = cfl * cff1 * Akt[ck + nij];
        }
        const double bc = Hz[ck] - fc - fc_prev;
        const double d = s_tn[(k - 1) * NTH + tid];
        if (k == 1) {
          const double cff = 1.0 / bc;
          CF[1] = cff * fc;
          DC[1] = cff * d;
        } else if (k < N) {
          const double cff = 1.0 / (bc - fc_prev * CF[k - 1]);
          CF[k] = cff * fc;
          DC[k] = cff * (d - fc_prev * DC[k - 1]);
        } else {
          DC[k] = (d - fc_prev * DC[k - 1]) / (bc - fc_prev * CF[k - 1]);
        }
        fc_prev = fc;
      }
    }
    double up = 0.0;
#pragma unroll
    for (int k = NMAX; k >= 1; k--) {
      if (k <= N) {
        const double v = (k == N) ? DC[k] : DC[k] - CF[k] * up;
        up = v;
        const long ck = c0 + (long)(k - 1) * nij;
        if constexpr (MASK) tn_g[ck] = v * GF(rmask)[c0];
        else tn_g[ck] = v;
      }
    }
    return;
  }
  // ---- back substitution + final update; Akt(kk), Hz(kk+1) prefetched two levels ahead ----
  double dcA_up = 0.0, dc_up = 0.0;
  double akA = Akt[c0 + (long)(N - 1) * nij], hzA = Hz[c0 + (long)(N - 1) * nij];
  double akB = 0.0, hzB = 0.0;
  if (N >= 2) { akB = Akt[c0 + (long)(N - 2) * nij]; hzB = Hz[c0 + (long)(N - 2) * nij]; }
#pragma unroll
  for (int kk = NMAX - 1; kk >= 0; kk--) {
    if (kk <= N - 1) {
      double akC = 0.0, hzC = 0.0;
      if (kk >= 2) { akC = Akt[c0 + (long)(kk - 2) * nij]; hzC = Hz[c0 + (long)(kk - 2) * nij]; }
      double dcA = 0.0;
      if (kk >= 1) {
        const double dc = DC[kk] - CF[kk] * dc_up;
        dc_up = dc;
        dcA = dc * akA;
      }
      const long ck = c0 + (long)kk * nij;
      const double ohz = 1.0 / hzA;
      const double cff1 = dt * ohz * (dcA_up - dcA);
      if constexpr (MASK) tn_g[ck] = (s_tn[kk * NTH + tid] + cff1) * GF(rmask)[c0];
      else tn_g[ck] = s_tn[kk * NTH + tid] + cff1;
      dcA_up = dcA;
      akA = akB; hzA = hzB; akB = akC; hzB = hzC;
    }
  }
}

// the straight-from-memory kernel for a pair (HSIMT in the vertical with another horizontal scheme)
template <int HADV, int VADV>
int launch_classic(int nnew, int itrc0, int ntr)
{
  const roms_bounds_t &b = g_ctx.b;
  const dim3 grid = grid_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, ntr);
  if (b.N > ROMS_MAXN) return roms_fail("roms_hip_step3d_t", "N > 64 not instantiated");
  if (g_ctx.p.masking) return roms_fail("roms_hip_step3d_t", "MASKING is not built for the pair (other scheme, HSIMT)");
  if (b.N <= 16)
    hipLaunchKernelGGL((k_step3d_t<HADV, VADV, 16>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
  else if (b.N <= 32)
    hipLaunchKernelGGL((k_step3d_t<HADV, VADV, 32>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
  else
    hipLaunchKernelGGL((k_step3d_t<HADV, VADV, 64>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
  KERNEL_CHECK("k_step3d_t");
  return 0;
}

template <int HADV, int VADV>
int launch_nmax(int nnew, int itrc0, int ntr)
{
  const roms_bounds_t &b = g_ctx.b;
  const dim3 grid = grid_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, ntr);
  if (b.N > ROMS_MAXN) return roms_fail("roms_hip_step3d_t", "N > 64 not instantiated");
  if constexpr (HADV == ADV_HSIMT) {      // straight-from-memory kernel
    if (b.N <= 16)
      hipLaunchKernelGGL((k_step3d_t<HADV, VADV, 16>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
    else if (b.N <= 32)
      hipLaunchKernelGGL((k_step3d_t<HADV, VADV, 32>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
    else
      hipLaunchKernelGGL((k_step3d_t<HADV, VADV, 64>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
  } else if (!g_ctx.p.splines_vdiff) {    // without SPLINES_VDIFF: one instantiation for every N
    if (g_ctx.p.masking)
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, ROMS_MAXN, true, false, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
    else
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, ROMS_MAXN, false, false, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
  } else if (g_ctx.p.masking) {           // software-pipelined kernel, land/sea masks applied
    if (b.N <= 16)
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, 16, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
    else if (b.N <= 32)
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, 32, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
    else
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, 64, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
  } else {                                // software-pipelined kernel
    if (b.N <= 16)
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, 16, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
    else if (b.N <= 32)
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, 32, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
    else if (b.N <= 48)     // the column arrays then spill into AGPRs (one wave per SIMD); slower per cell, same results
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, 48, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
    else
      hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, 64, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
  }
  KERNEL_CHECK("k_step3d_t");
  if constexpr (HADV != ADV_HSIMT) {
    // LuvSrc: the cells with a source face (one instantiation serves every N: a handful of columns)
    const int ncell = g_ctx.hostc.src.ncell;
    if (ncell > 0) {
      const dim3 gs((ncell + BLK_X * BLK_Y - 1) / (BLK_X * BLK_Y), ntr);
      const bool spl = g_ctx.p.splines_vdiff != 0;
      if (g_ctx.p.masking) {
        if (spl) hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, ROMS_MAXN, true, true>), gs, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
        else hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, ROMS_MAXN, true, true, false>), gs, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
      } else {
        if (spl) hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, ROMS_MAXN, false, true>), gs, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
        else hipLaunchKernelGGL((k_step3d_t_pipe<HADV, VADV, ROMS_MAXN, false, true, false>), gs, block2d(), 0, g_ctx.stream, g_ctx.devc, nnew, itrc0, ntr);
      }
      KERNEL_CHECK("k_step3d_t (source cells)");
    }
  }
  return 0;
}

}  // namespace

extern "C" int roms_hip_step3d_t(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_step3d_t");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  const roms_bounds_t &b = g_ctx.b;
  const roms_params_t &p = g_ctx.p;
  if (b.N < 4) return roms_fail("roms_hip_step3d_t", "N < 4");
  if (!p.splines_vdiff)
    for (int it = 1; it <= b.NT; it++)
      if (p.Hadv[it - 1] == ADV_HSIMT || p.Vadv[it - 1] == ADV_HSIMT)
        return roms_fail("roms_hip_step3d_t", "HSIMT without SPLINES_VDIFF is not built (the straight-from-memory kernel "
                                              "carries the spline form of the vertical diffusion only)");
  {
    ScopedTimer tm("step3d_t");
    // one launch per run of consecutive tracers sharing a scheme pair (run-time
    // selection per tracer, step3d_t.F:596+: one kernel per scheme, host dispatch)
    int it = 1;
    while (it <= b.NT) {
      const int ha = p.Hadv[it - 1], va = p.Vadv[it - 1];
      int n = 1;
      while (it + n <= b.NT && p.Hadv[it + n - 1] == ha && p.Vadv[it + n - 1] == va) n++;
      const int hv = ha * 16 + va;
      switch (hv) {
      case ADV_U3 * 16 + ADV_C4:  rc = launch_nmax<ADV_U3, ADV_C4>(s->nnew, it, n); break;
      case ADV_U3 * 16 + ADV_SU3: rc = launch_nmax<ADV_U3, ADV_C4>(s->nnew, it, n); break;
      case ADV_A4 * 16 + ADV_A4:  rc = launch_nmax<ADV_A4, ADV_A4>(s->nnew, it, n); break;
      case ADV_C4 * 16 + ADV_C4:  rc = launch_nmax<ADV_C4, ADV_C4>(s->nnew, it, n); break;
      case ADV_SU3 * 16 + ADV_SU3: rc = launch_nmax<ADV_C4, ADV_C4>(s->nnew, it, n); break;
      case ADV_C2 * 16 + ADV_C2:  rc = launch_nmax<ADV_C2, ADV_C2>(s->nnew, it, n); break;
      case ADV_U3 * 16 + ADV_SPLINES: rc = launch_nmax<ADV_U3, ADV_SPLINES>(s->nnew, it, n); break;
      case ADV_C4 * 16 + ADV_SPLINES: rc = launch_nmax<ADV_C4, ADV_SPLINES>(s->nnew, it, n); break;
      case ADV_A4 * 16 + ADV_SPLINES: rc = launch_nmax<ADV_A4, ADV_SPLINES>(s->nnew, it, n); break;
      case ADV_HSIMT * 16 + ADV_HSIMT: {
        // three-point footprint: refresh the ghost points of t(nnew) first (step3d_t.F:369-386); classic kernel
        if (b.NghostPoints != 3) return roms_fail("roms_hip_step3d_t", "HSIMT needs NghostPoints = 3 (inp_par.F:266-278)");
        const long n3r_ = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * b.N;
        halo_batch_begin();
        for (int q = 0; q < n; q++)
          halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_t] + ((long)(s->nnew - 1) + 3L * (it + q - 1)) * n3r_);
        if ((rc = halo_batch_end())) return rc;
        rc = launch_nmax<ADV_HSIMT, ADV_HSIMT>(s->nnew, it, n);
        break;
      }
      // HSIMT vertically with another scheme horizontally: the straight-from-memory kernel (two ghost points suffice)
      case ADV_U3 * 16 + ADV_HSIMT: rc = launch_classic<ADV_U3, ADV_HSIMT>(s->nnew, it, n); break;
      case ADV_C4 * 16 + ADV_HSIMT: rc = launch_classic<ADV_C4, ADV_HSIMT>(s->nnew, it, n); break;
      case ADV_SU3 * 16 + ADV_HSIMT: rc = launch_classic<ADV_C4, ADV_HSIMT>(s->nnew, it, n); break;
      case ADV_A4 * 16 + ADV_HSIMT: rc = launch_classic<ADV_A4, ADV_HSIMT>(s->nnew, it, n); break;
      case ADV_C2 * 16 + ADV_HSIMT: rc = launch_classic<ADV_C2, ADV_HSIMT>(s->nnew, it, n); break;
      case ADV_MPDATA * 16 + ADV_MPDATA:
        // multi-pass: upstream step, anti-diffusive velocities, FCT limiter, corrected step (k_mpdata.hip)
        rc = roms_launch_step3d_t_mpdata(s->nnew, it, n);
        break;
      default:
        return roms_fail("roms_hip_step3d_t", "advection scheme pair not implemented (MPDATA and HSIMT only as H+V pairs)");
      }
      if (rc) return rc;
      it += n;
    }
  }
  // t3dbc_tile + periodic wrap / mp_exchange4d, step3d_t.F:1564-1626
  for (int it = 1; it <= b.NT; it++)
    if ((rc = bc_t3d(s->nnew, it, s->nstp))) return rc;
  const long n3r = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * b.N;
  halo_batch_begin();
  for (int it = 1; it <= b.NT; it++)
    halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_t] + ((long)(s->nnew - 1) + 3L * (it - 1)) * n3r);
  return halo_batch_end();
}
