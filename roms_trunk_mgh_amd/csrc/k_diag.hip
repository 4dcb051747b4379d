// k_diag.hip -- the two diagnostics main3d runs every step next to the hot path:
//   roms_hip_wvelocity  wvelocity_tile  ROMS/Nonlinear/wvelocity.F:61   true vertical velocity wvel(0:N)
//   roms_hip_diag       diag_tile       ROMS/Nonlinear/diag.F:80        tile-local part: volume, kinetic and
//                                       potential energy sums, Courant-number maximum with its location,
//                                       maximum speed and density (SOLVE3D branch)
// diag: the reference sums j first and then i (diag.F:262-290) and finds the Courant maximum in loop order
// (j ascending, k descending, i ascending; strict ">"): three kernels reproduce both bit for bit --
// per column, per i (serial in j), one thread (serial in i).  The global reduction over tiles (mp_reduce,
// mp_reduce2 MAXLOC; diag.F:398-420) stays with the caller.
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

// ---------------------------------------------------------------------------------------------------
// wvelocity.F:137-236: vert(k) = horizontal advection of z_r by (u,v) averaged to rho-points; wvel(k)
// interpolates it to W-levels (parabolic extrapolation at bottom and surface) and adds the omega part.
// One thread per column; vert(k-1..k+2) live in a rolling register window.
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_wvelocity(const RomsDev *__restrict__ c, int ninp)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const long a = I2(i, j);
  const gcd_t u = (gcd_t)(c->F.u + (long)(ninp - 1) * n3r), v = (gcd_t)(c->F.v + (long)(ninp - 1) * n3r);
  const gcd_t z_r = (gcd_t)c->F.z_r, z_w = (gcd_t)c->F.z_w, W = (gcd_t)c->F.W;
  const gd_t wvel = (gd_t)c->F.wvel;
  const gcd_t pm = (gcd_t)c->F.pm, pn = (gcd_t)c->F.pn;
  const double pmw = pm[a - 1] + pm[a], pme = pm[a] + pm[a + 1];
  const double pns = pn[a - ni] + pn[a], pnn = pn[a] + pn[a + ni];
  const double pmn = pm[a] * pn[a];
  auto vert_at = [&](int k) {                                    // :150-172
    const long q = a + (long)(k - 1) * nij;
    const double zr0 = z_r[q];
    const double wu0 = u[q] * (zr0 - z_r[q - 1]) * pmw;
    const double wu1 = u[q + 1] * (z_r[q + 1] - zr0) * pme;
    double ve = 0.25 * (wu0 + wu1);
    const double wv0 = v[q] * (zr0 - z_r[q - ni]) * pns;
    const double wv1 = v[q + ni] * (z_r[q + ni] - zr0) * pnn;
    ve = ve + 0.25 * (wv0 + wv1);
    return ve;
  };
  auto w3i = [&](int k) { return a + (long)k * nij; };
  const double cff1 = 3.0 / 8.0, cff2 = 3.0 / 4.0, cff3 = 1.0 / 8.0, cff4 = 9.0 / 16.0, cff5 = 1.0 / 16.0;
  const double zw0 = z_w[w3i(0)], zwN = z_w[w3i(N)];
  const double wrk = (GF(DU_avg1)[a] - GF(DU_avg1)[a + 1] + GF(DV_avg1)[a] - GF(DV_avg1)[a + ni]) / (zwN - zw0);
  double v1 = vert_at(1), v2 = vert_at(2), v3 = vert_at(3);
  {
    const double slope = (z_r[a] - zw0) / (z_r[a + nij] - z_r[a]);           // extrapolation slope
    wvel[w3i(0)] = cff1 * (v1 - slope * (v2 - v1)) + cff2 * v1 - cff3 * v2;
    wvel[w3i(1)] = pmn * (W[w3i(1)] + wrk * (z_w[w3i(1)] - zw0)) + cff1 * v1 + cff2 * v2 - cff3 * v3;
  }
  // window: vm = vert(k-1), v0 = vert(k), vp = vert(k+1), vq = vert(k+2)
  double vm = v1, v0 = v2, vp = v3;
  for (int k = 2; k <= N - 2; k++) {
    const double vq = vert_at(k + 2);
    wvel[w3i(k)] = pmn * (W[w3i(k)] + wrk * (z_w[w3i(k)] - zw0)) + cff4 * (v0 + vp) - cff5 * (vm + vq);
    vm = v0; v0 = vp; vp = vq;
  }
  // now vm = vert(N-2), v0 = vert(N-1), vp = vert(N)
  {
    const long qN = a + (long)(N - 1) * nij;
    const double slope = (zwN - z_r[qN]) / (z_r[qN] - z_r[qN - nij]);
    wvel[w3i(N)] = pmn * wrk * (zwN - zw0) + cff1 * (vp + slope * (vp - v0)) + cff2 * vp - cff3 * v0;
    wvel[w3i(N - 1)] = pmn * (W[w3i(N - 1)] + wrk * (z_w[w3i(N - 1)] - zw0)) + cff1 * vp + cff2 * v0 - cff3 * vm;
  }
}

// ---------------------------------------------------------------------------------------------------
struct DiagScratch { double *ke, *pe, *C, *Cu, *Cv, *Cw, *Ck, *spd, *rho, *vol; };

// diag.F:199-240, one column
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_diag_col(const RomsDev *__restrict__ c, int idia, DiagScratch w)
{
  DEV_PROLOGUE(c)
  const roms_params_t &p = c->p;
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const long a = I2(i, j);
  const gcd_t u = (gcd_t)(c->F.u + (long)(idia - 1) * n3r), v = (gcd_t)(c->F.v + (long)(idia - 1) * n3r);
  const gcd_t Hz = (gcd_t)c->F.Hz, z_r = (gcd_t)c->F.z_r, z_w = (gcd_t)c->F.z_w, rho = (gcd_t)c->F.rho;
  const gcd_t wvel = (gcd_t)c->F.wvel;
  const double zwN = z_w[a + (long)N * nij], zw0 = z_w[a];
  const double pm = GF(pm)[a], pn = GF(pn)[a], dt = p.dt;
  double ke = 0.0, pe = 0.5 * p.g * zwN * zwN;
  const double cff = p.g / p.rho0;
  double mC = 0.0, mCu = 0.0, mCv = 0.0, mCw = 0.0, mspd = 0.0, mrho = -1.0E+37;
  int mk = 0;
  double wv_up = wvel[a + (long)N * nij];
  for (int k = N; k >= 1; k--) {
    const long q = a + (long)(k - 1) * nij;
    const double u0 = u[q], u1 = u[q + 1], v0 = v[q], v1 = v[q + ni], hz = Hz[q], rh = rho[q];
    const double wv_dn = wvel[q];                                // W-level k-1
    const double u2v2 = u0 * u0 + u1 * u1 + v0 * v0 + v1 * v1;
    ke = ke + hz * 0.25 * u2v2;
    pe = pe + cff * hz * (rh + 1000.0) * (z_r[q] - zw0);
    const double Cu = 0.5 * fabs(u0 + u1) * dt * pm;
    const double Cv = 0.5 * fabs(v0 + v1) * dt * pn;
    const double Cw = 0.5 * fabs(wv_dn + wv_up) * dt / hz;
    const double C = Cu + Cv + Cw;
    if (C > mC) { mC = C; mCu = Cu; mCv = Cv; mCw = Cw; mk = k; }
    mspd = fmax(mspd, sqrt(0.5 * u2v2));
    mrho = fmax(mrho, rh);
    wv_up = wv_dn;
  }
  // the terms of the j-sums (diag.F:268-278), so that the serial pass reads three values per point
  const double om = GF(omn)[a];
  w.vol[a] = om * (zwN - zw0);
  ke = om * ke;
  pe = om * pe;
  w.ke[a] = ke; w.pe[a] = pe; w.C[a] = mC; w.Cu[a] = mCu; w.Cv[a] = mCv; w.Cw[a] = mCw; w.Ck[a] = (double)mk;
  w.spd[a] = mspd; w.rho[a] = mrho;
}

// per-i partial results (j collapsed serially, diag.F:262-280); slot q of row r lives at R[r*nI + (i-Istr)]
enum { R_VOL = 0, R_PE, R_KE, R_C, R_CU, R_CV, R_CW, R_CJ, R_CK, R_SPD, R_RHO, R_COUNT };

// serial in j, one thread per i; DIAG_UNR rows are fetched at once so that the (independent) loads of a
// group overlap instead of paying one memory latency per row
#define DIAG_UNR 16
__global__ void __launch_bounds__(64)
k_diag_rows(const RomsDev *__restrict__ c, DiagScratch w, double *__restrict__ R)
{
  DEV_PROLOGUE(c)
  const int i = b.Istr + blockIdx.x * blockDim.x + threadIdx.x;
  if (i > b.Iend) return;
  const int nI = b.Iend - b.Istr + 1, o = i - b.Istr;
  const gcd_t gvol = (gcd_t)w.vol, gpe = (gcd_t)w.pe, gke = (gcd_t)w.ke, gC = (gcd_t)w.C, gspd = (gcd_t)w.spd,
              grho = (gcd_t)w.rho;
  double vol = 0.0, pe = 0.0, ke = 0.0, mC = 0.0, mCu = 0.0, mCv = 0.0, mCw = 0.0, mspd = 0.0, mrho = -1.0E+37;
  int mj = 0, mk = 0;
  for (int j0 = b.Jstr; j0 <= b.Jend; j0 += DIAG_UNR) {
    double tv[DIAG_UNR], tp[DIAG_UNR], tk[DIAG_UNR], tc[DIAG_UNR], ts[DIAG_UNR], tr[DIAG_UNR];
#pragma unroll
    for (int q = 0; q < DIAG_UNR; q++) {
      const int j = (j0 + q <= b.Jend) ? j0 + q : b.Jend;          // clamped: unused beyond Jend
      const long a = I2(i, j);
      tv[q] = gvol[a]; tp[q] = gpe[a]; tk[q] = gke[a]; tc[q] = gC[a]; ts[q] = gspd[a]; tr[q] = grho[a];
    }
#pragma unroll
    for (int q = 0; q < DIAG_UNR; q++) {
      const int j = j0 + q;
      if (j <= b.Jend) {
        vol = vol + tv[q];
        pe = pe + tp[q];
        ke = ke + tk[q];
        if (tc[q] > mC) {
          const long a = I2(i, j);
          mC = tc[q]; mCu = w.Cu[a]; mCv = w.Cv[a]; mCw = w.Cw[a]; mj = j; mk = (int)w.Ck[a];
        }
        mspd = fmax(mspd, ts[q]);
        mrho = fmax(mrho, tr[q]);
      }
    }
  }
  R[R_VOL * nI + o] = vol; R[R_PE * nI + o] = pe; R[R_KE * nI + o] = ke;
  R[R_C * nI + o] = mC; R[R_CU * nI + o] = mCu; R[R_CV * nI + o] = mCv; R[R_CW * nI + o] = mCw;
  R[R_CJ * nI + o] = (double)mj; R[R_CK * nI + o] = (double)mk;
  R[R_SPD * nI + o] = mspd; R[R_RHO * nI + o] = mrho;
}

// diag.F:281-290, one workgroup.  The three sums are serial in i (the reference's order): the partial rows are
// staged in LDS chunk by chunk and three wavefronts run one chain each, 16 values per round.  The maxima need no
// order: "first strict maximum in loop order" = the largest C, ties to the smallest j, then the largest k, then
// the smallest i -- a total order, so all threads scan a share and an LDS tree finishes.
#define DIAG_CH 512
struct DiagBest { double C, Cu, Cv, Cw; int i, j, k; };
__device__ __forceinline__ bool diag_better(const DiagBest &x, const DiagBest &y)     // x ahead of y?
{
  if (x.C != y.C) return x.C > y.C;
  if (!(x.C > 0.0)) return false;                     // nothing exceeds zero: keep (0,0,0)
  if (x.j != y.j) return x.j < y.j;
  if (x.k != y.k) return x.k > y.k;
  return x.i < y.i;
}
__global__ void __launch_bounds__(256)
k_diag_final(const RomsDev *__restrict__ c, const double *__restrict__ R, double *__restrict__ out)
{
  const roms_bounds_t &b = c->b;
  __shared__ double sR[3 * DIAG_CH];
  __shared__ DiagBest sB[256];
  __shared__ double sSpd[256], sRho[256];
  const int nI = b.Iend - b.Istr + 1;
  const int tid = threadIdx.x;
  // ---- maxima ----
  DiagBest best{0.0, 0.0, 0.0, 0.0, 0, 0, 0};
  double mspd = 0.0, mrho = -1.0E+37;
  for (int o = tid; o < nI; o += 256) {
    DiagBest x{R[R_C * nI + o], R[R_CU * nI + o], R[R_CV * nI + o], R[R_CW * nI + o], b.Istr + o,
               (int)R[R_CJ * nI + o], (int)R[R_CK * nI + o]};
    if (diag_better(x, best)) best = x;
    mspd = fmax(mspd, R[R_SPD * nI + o]);
    mrho = fmax(mrho, R[R_RHO * nI + o]);
  }
  sB[tid] = best; sSpd[tid] = mspd; sRho[tid] = mrho;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st) {
      if (diag_better(sB[tid + st], sB[tid])) sB[tid] = sB[tid + st];
      sSpd[tid] = fmax(sSpd[tid], sSpd[tid + st]);
      sRho[tid] = fmax(sRho[tid], sRho[tid + st]);
    }
    __syncthreads();
  }
  // ---- sums, serial in i ----
  double acc = 0.0;
  const int chain = tid / 64;                         // wavefront 0: volume, 1: potential, 2: kinetic energy
  for (int o0 = 0; o0 < nI; o0 += DIAG_CH) {
    const int n = (nI - o0 < DIAG_CH) ? nI - o0 : DIAG_CH;
    __syncthreads();
    for (int e = tid; e < 3 * DIAG_CH; e += 256) {
      const int r = e / DIAG_CH, q = e % DIAG_CH;
      sR[e] = (q < n) ? R[(r == 0 ? R_VOL : (r == 1 ? R_PE : R_KE)) * nI + o0 + q] : 0.0;
    }
    __syncthreads();
    if ((tid & 63) == 0 && chain < 3) {
      const double *src = sR + chain * DIAG_CH;
      for (int q0 = 0; q0 < n; q0 += 16) {
        double t[16];
#pragma unroll
        for (int q = 0; q < 16; q++) t[q] = src[q0 + q < DIAG_CH ? q0 + q : DIAG_CH - 1];
#pragma unroll
        for (int q = 0; q < 16; q++)
          if (q0 + q < n) acc = acc + t[q];
      }
    }
  }
  if (tid == 0) out[0] = acc;
  if (tid == 64) out[2] = acc;
  if (tid == 128) out[1] = acc;
  if (tid == 192) {
    out[3] = sSpd[0]; out[4] = sRho[0];
    out[5] = sB[0].C; out[6] = sB[0].Cu; out[7] = sB[0].Cv; out[8] = sB[0].Cw;
    out[9] = (double)sB[0].i; out[10] = (double)sB[0].j; out[11] = (double)sB[0].k;
  }
}

double *g_diag_dev = nullptr;       // 12 results
double *g_diag_rows = nullptr;      // R_COUNT x (Iend-Istr+1)
long g_diag_rows_n = 0;

}  // namespace

void diag_release()
{
  if (g_diag_dev) (void)hipFree(g_diag_dev);
  if (g_diag_rows) (void)hipFree(g_diag_rows);
  g_diag_dev = g_diag_rows = nullptr;
  g_diag_rows_n = 0;
}

extern "C" int roms_hip_wvelocity(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_wvelocity");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  const roms_bounds_t &b = g_ctx.b;
  if (b.N < 3) return roms_fail("roms_hip_wvelocity", "needs N >= 3");
  // wvelocity.F:119-135: the exchanges of DU_avg1, DV_avg1 come first (intent inout there as well)
  halo_batch_begin();
  halo_exchange2d(GT_U, g_ctx.dev[FID_DU_avg1]);
  halo_exchange2d(GT_V, g_ctx.dev[FID_DV_avg1]);
  if ((rc = halo_batch_end())) return rc;
  {
    ScopedTimer tm("wvelocity");
    hipLaunchKernelGGL(k_wvelocity, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s->nstp);
    KERNEL_CHECK("k_wvelocity");
  }
  return bc_w3d(g_ctx.dev[FID_wvel]);       // bc_w3d_tile + exchange, :237-250
}

extern "C" int roms_hip_diag(const roms_step_idx_t *s, double *out12)
{
  int rc = roms_entry_check("roms_hip_diag");
  if (rc) return rc;
  if (!out12) return roms_fail("roms_hip_diag", "null output");
  const roms_bounds_t &b = g_ctx.b;
  const int nI = b.Iend - b.Istr + 1, nJ = b.Jend - b.Jstr + 1;
  if (!g_diag_dev) HIP_TRY(hipMalloc(&g_diag_dev, 12 * sizeof(double)));
  if (g_diag_rows_n < (long)R_COUNT * nI) {
    if (g_diag_rows) HIP_TRY(hipFree(g_diag_rows));
    g_diag_rows = nullptr;
    HIP_TRY(hipMalloc(&g_diag_rows, sizeof(double) * R_COUNT * nI));
    g_diag_rows_n = (long)R_COUNT * nI;
  }
  double **ws = g_ctx.hostc.ws2;
  DiagScratch w{ws[8], ws[9], ws[10], ws[11], ws[12], ws[13], ws[14], ws[15], ws[16], ws[17]};
  {
    ScopedTimer tm("diag");
    hipLaunchKernelGGL(k_diag_col, grid2d(nI, nJ), block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, w);
    KERNEL_CHECK("k_diag_col");
    hipLaunchKernelGGL(k_diag_rows, dim3((nI + 63) / 64), dim3(64), 0, g_ctx.stream, g_ctx.devc, w, g_diag_rows);
    KERNEL_CHECK("k_diag_rows");
    hipLaunchKernelGGL(k_diag_final, dim3(1), dim3(256), 0, g_ctx.stream, g_ctx.devc, (const double *)g_diag_rows,
                       g_diag_dev);
    KERNEL_CHECK("k_diag_final");
  }
  HIP_TRY(hipMemcpyAsync(out12, g_diag_dev, 12 * sizeof(double), hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return 0;
}
