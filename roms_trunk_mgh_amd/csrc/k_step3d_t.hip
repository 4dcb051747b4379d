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

#include "advect.h"

namespace {

// TY = rows of columns per workgroup (64 x TY threads); LDSO = 1 keeps 1/Hz of the
// column in LDS between the upward and the downward sweep (one HBM pass less per tracer).
template <int HADV, int VADV, int NMAX, int TY, int LDSO>
__global__ void __launch_bounds__(BLK_X *TY)
k_step3d_t(const RomsDev *__restrict__ c, int nnew, int itrc0, int ntr)
{
  DEV_PROLOGUE(c)
  constexpr int NTH = BLK_X * TY;
  __shared__ double s_ohz[LDSO ? NMAX * NTH : 1];
  const int tid = threadIdx.y * BLK_X + threadIdx.x;
  const TileTr tt = decode_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, ntr, TY);
  if (!tt.valid) return;
  const int i = b.Istr + tt.bx * BLK_X + threadIdx.x;
  const int j = b.Jstr + tt.by * TY + threadIdx.y;
  const int itrc = itrc0 + tt.itr;              // 1-based tracer index
  if (i > b.Iend || j > b.Jend) return;
  const int ltrc = itrc < b.NAT ? itrc : b.NAT;
  const double dt = c->p.dt;
  const double *__restrict__ t3 = c->F.t + (2L + 3L * (itrc - 1)) * n3r;
  double *__restrict__ tn_g = c->F.t + ((long)(nnew - 1) + 3L * (itrc - 1)) * n3r;
  const double *__restrict__ Huon = c->F.Huon;
  const double *__restrict__ Hvom = c->F.Hvom;
  const double *__restrict__ Wv = c->F.W;
  const double *__restrict__ Hz = c->F.Hz;
  const double *__restrict__ Akt = c->F.Akt + (long)(ltrc - 1) * n3w;
  const double cffdt = dt * c->F.pm[I2(i, j)] * c->F.pn[I2(i, j)];
  const bool s_wall = b.south_edge && !b.NSperiodic && j == b.Jstr;       // FE(Jstr-1)=FE(Jstr)
  const bool n_wall = b.north_edge && !b.NSperiodic && j == b.Jend;       // FE(Jend+2)=FE(Jend+1)
  const bool n_wall1 = b.north_edge && !b.NSperiodic && j == b.Jend - 1;
  (void)n_wall1;
  const long c0 = I2(i, j);

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
      tkp2 = (k + 2 <= N) ? t3[ck + 2 * nij] : 0.0;
      // ---- horizontal fluxes, step3d_t.F:596-828 ----
      const double xm2 = t3[ck - 2], xm1 = t3[ck - 1], xp1 = t3[ck + 1], xp2 = t3[ck + 2];
      const double ym1 = t3[ck - ni], yp1 = t3[ck + ni];
      const double ym2 = s_wall ? 0.0 : t3[ck - 2 * ni];
      const double yp2 = n_wall ? 0.0 : t3[ck + 2 * ni];
      const double hu0 = Huon[ck], hu1 = Huon[ck + 1];
      const double hv0 = Hvom[ck], hv1 = Hvom[ck + ni];
      const double dxm1 = xm1 - xm2, dx0 = tk - xm1, dxp1 = xp1 - tk, dxp2 = xp2 - xp1;
      const double dy0 = tk - ym1, dyp1 = yp1 - tk;
      const double dym1 = s_wall ? dy0 : (ym1 - ym2);
      const double dyp2 = n_wall ? dyp1 : (yp2 - yp1);
      const double FXi = hflux<HADV>(hu0, xm1, tk, dxm1, dx0, dxp1);
      const double FXip1 = hflux<HADV>(hu1, tk, xp1, dx0, dxp1, dxp2);
      const double FEj = hflux<HADV>(hv0, ym1, tk, dym1, dy0, dyp1);
      const double FEjp1 = hflux<HADV>(hv1, tk, yp1, dy0, dyp1, dyp2);
      // ---- vertical flux through the top face of level k ----
      double FCk;
      if (k == N) FCk = 0.0;
      else if constexpr (VADV == ADV_SPLINES) FCk = spl[k];
      else {
        double cfk = 0.0, cfk1 = 0.0;
        if constexpr (VADV == ADV_A4) { cfk = a4cf[k]; cfk1 = a4cf[k + 1]; }
        FCk = vflux<VADV>(k, N, Wv[ck + nij], tkm1, tk, tkp1, tkp2, cfk, cfk1);
      }
      // ---- advective update, step3d_t.F:857-875 and :1168-1208 ----
      const double hz = Hz[ck];
      const double ohz = 1.0 / hz;
      double tv = tn_g[ck];
      {
        const double cff1 = cffdt * (FXip1 - FXi);
        const double cff2 = cffdt * (FEjp1 - FEj);
        const double cff3 = cff1 + cff2;
        tv = tv - cff3;
      }
      tv = tv - cffdt * (FCk - FCprev);
      tv = tv * ohz;
      tn[k] = tv;
      if constexpr (LDSO) s_ohz[(k - 1) * NTH + tid] = ohz;
      FCprev = FCk;
      // ---- Thomas forward elimination for row kk = k-1, step3d_t.F:1376-1410 ----
      const double akt_0 = Akt[ck + nij];          // Akt(i,j,k)
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
      double ohz;
      if constexpr (LDSO) ohz = s_ohz[kk * NTH + tid];
      else ohz = 1.0 / Hz[ck];
      const double cff1 = dt * ohz * (dcA_up - dcA);
      tn_g[ck] = tn[kk + 1] + cff1;
      dcA_up = dcA;
    }
  }
}

template <int HADV, int VADV, int TY, int LDSO>
int launch_var(int nnew, int itrc0, int ntr)
{
  const roms_bounds_t &b = g_ctx.b;
  const dim3 grid = grid_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, ntr, TY);
  const dim3 block(BLK_X, TY, 1);
  if (b.N <= 16)
    hipLaunchKernelGGL((k_step3d_t<HADV, VADV, 16, TY, LDSO>), grid, block, 0, g_ctx.stream, g_ctx.devc, nnew, itrc0,
                       ntr);
  else if (b.N <= 32)
    hipLaunchKernelGGL((k_step3d_t<HADV, VADV, 32, TY, LDSO>), grid, block, 0, g_ctx.stream, g_ctx.devc, nnew, itrc0,
                       ntr);
  else
    return roms_fail("roms_hip_step3d_t", "N > 32 not instantiated");
  KERNEL_CHECK("k_step3d_t");
  return 0;
}

template <int HADV, int VADV>
int launch_nmax(int nnew, int itrc0, int ntr)
{
  if constexpr (HADV == ADV_U3 && VADV == ADV_C4) {
    // A/B variants (ROMS_HIP_S3T_VARIANT) while the workgroup shape is being tuned
    static const int variant = getenv("ROMS_HIP_S3T_VARIANT") ? atoi(getenv("ROMS_HIP_S3T_VARIANT")) : 0;
    switch (variant) {
    case 1: return launch_var<HADV, VADV, 4, 1>(nnew, itrc0, ntr);
    case 2: return launch_var<HADV, VADV, 8, 0>(nnew, itrc0, ntr);
    case 3: return launch_var<HADV, VADV, 8, 1>(nnew, itrc0, ntr);
    default: break;
    }
  }
  return launch_var<HADV, VADV, 4, 0>(nnew, itrc0, ntr);
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
      default:
        return roms_fail("roms_hip_step3d_t", "advection scheme pair not implemented (MPDATA/HSIMT pending)");
      }
      if (rc) return rc;
      it += n;
    }
  }
  // t3dbc_tile + periodic wrap / mp_exchange4d, step3d_t.F:1564-1626
  for (int it = 1; it <= b.NT; it++)
    if ((rc = bc_t3d(s->nnew, it))) return rc;
  const long n3r = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * b.N;
  for (int it = 1; it <= b.NT; it++)
    if ((rc = halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_t] + ((long)(s->nnew - 1) + 3L * (it - 1)) * n3r))) return rc;
  return 0;
}
