// k_base.hip -- closed-wall boundary conditions and the small glue kernels of
// the 3-D step: set_massflux, omega, set_zeta, set_depth.
//
// All of them are HBM-bound single-pass kernels: one thread per (i,j) water
// column, consecutive lanes = consecutive i, so every k-level access of a
// wavefront is one contiguous 512-byte segment.
#include "roms_dev.h"

int roms_entry_check(const char *name);

int check_lbc()
{
  const roms_bounds_t &b = g_ctx.b;
  const roms_params_t &p = g_ctx.p;
  if (!b.EWperiodic || b.NSperiodic || p.lbc_south != LBC_CLOSED || p.lbc_north != LBC_CLOSED)
    return roms_fail("check_lbc", "only LBC == Per Clo Per Clo is implemented on this path");
  return 0;
}

// ---------------------------------------------------------------------------
// Closed S/N walls.  mode: 0 zero-gradient rho-type (zetabc.F:48, t3dbc_im.F:50,
// bc_3d.F:588), 1 tangential u (gamma2 slip; u2dbc_im.F:51, u3dbc_im.F:50),
// 2 normal v = 0 (v2dbc_im.F:52, v3dbc_im.F:50).  A points at the (i,j,k=first)
// plane of the wanted time level; nk planes are processed.  masked (MASKING applications): the boundary
// value is multiplied by the land/sea mask of the boundary point -- rmask for zeta and tracers
// (zetabc.F:540, t3dbc_im.F:483), umask for the tangential velocity (u2dbc_im.F:975, u3dbc_im.F:520);
// bc_w3d_tile has no mask.
// ---------------------------------------------------------------------------
__global__ void k_wall_bc(const RomsDev *__restrict__ c, double *__restrict__ A, int nk, int mode, int masked)
{
  DEV_PROLOGUE(c)
  const int k = blockIdx.y;
  if (k >= nk) return;
  int i0, i1;
  if (mode == 1) { i0 = b.EWperiodic ? b.IstrU : b.Istr; i1 = b.EWperiodic ? b.Iend : b.IendR; }
  else { i0 = b.Istr; i1 = b.Iend; }
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i > i1) return;
  double *P = A + (long)k * nij;
  const double g2 = c->p.gamma2;
  const double *M = mode == 0 ? c->F.rmask : c->F.umask;
  if (b.south_edge) {
    const long q = I2(i, b.Jstr - 1);
    if (mode == 0) { double x = P[I2(i, b.Jstr)]; if (masked) x = x * M[q]; P[q] = x; }
    else if (mode == 1) { double x = g2 * P[I2(i, b.Jstr)]; if (masked) x = x * M[q]; P[q] = x; }
    else P[I2(i, b.Jstr)] = 0.0;
  }
  if (b.north_edge) {
    const long q = I2(i, b.Jend + 1);
    if (mode == 0) { double x = P[I2(i, b.Jend)]; if (masked) x = x * M[q]; P[q] = x; }
    else if (mode == 1) { double x = g2 * P[I2(i, b.Jend)]; if (masked) x = x * M[q]; P[q] = x; }
    else P[q] = 0.0;
  }
}

static int wall_bc(double *A, int nk, int mode, bool maskable = true)
{
  const roms_bounds_t &b = g_ctx.b;
  if (!b.south_edge && !b.north_edge) return 0;
  const int nx = b.Iend - b.Istr + 2;
  dim3 grid((nx + 255) / 256, nk);
  hipLaunchKernelGGL(k_wall_bc, grid, dim3(256), 0, g_ctx.stream, g_ctx.devc, A, nk, mode,
                     (int)(maskable && g_ctx.p.masking));
  KERNEL_CHECK("k_wall_bc");
  return 0;
}

static inline long nij_host()
{
  const roms_bounds_t &b = g_ctx.b;
  return (long)(b.UBi - b.LBi + 1) * (long)(b.UBj - b.LBj + 1);
}

int bc_zeta(int kout) { return wall_bc(g_ctx.dev[FID_zeta] + (long)(kout - 1) * nij_host(), 1, 0); }
int bc_u2d(int kout)  { return wall_bc(g_ctx.dev[FID_ubar] + (long)(kout - 1) * nij_host(), 1, 1); }
int bc_v2d(int kout)  { return wall_bc(g_ctx.dev[FID_vbar] + (long)(kout - 1) * nij_host(), 1, 2); }
int bc_u3d(int nout)  { return wall_bc(g_ctx.dev[FID_u] + (long)(nout - 1) * nij_host() * g_ctx.b.N, g_ctx.b.N, 1); }
int bc_v3d(int nout)  { return wall_bc(g_ctx.dev[FID_v] + (long)(nout - 1) * nij_host() * g_ctx.b.N, g_ctx.b.N, 2); }
int bc_t3d(int nout, int itrc)
{
  const long n3r = nij_host() * g_ctx.b.N;
  return wall_bc(g_ctx.dev[FID_t] + ((long)(nout - 1) + 3L * (itrc - 1)) * n3r, g_ctx.b.N, 0);
}
int bc_w3d(double *A)
{
  int rc = wall_bc(A, g_ctx.b.N + 1, 0, false);
  if (rc) return rc;
  return halo_exchange3d(GT_R, g_ctx.b.N + 1, A);
}

// ---------------------------------------------------------------------------
// set_massflux_tile -- ROMS/Nonlinear/set_massflux.F:73-188
//   Huon = 0.5 (Hz(i)+Hz(i-1)) u on_u ,  Hvom = 0.5 (Hz(j)+Hz(j-1)) v om_v
// 5 field passes (read Hz,u,v; write Huon,Hvom): 40 B per cell.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_set_massflux(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrT + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrT + XB.y * BLK_Y + threadIdx.y;
  const int k = XB.z + 1;
  if (i > b.IendT || j > b.JendT) return;
  const gcd_t Hz = (gcd_t)(c->F.Hz);
  const gcd_t u = (gcd_t)(c->F.u + (long)(nrhs - 1) * n3r);
  const gcd_t v = (gcd_t)(c->F.v + (long)(nrhs - 1) * n3r);
  const double hz = Hz[I3(i, j, k)];
  if (i >= b.IstrP)
    GF(Huon)[I3(i, j, k)] = 0.5 * (hz + Hz[I3(i - 1, j, k)]) * u[I3(i, j, k)] * GF(on_u)[I2(i, j)];
  if (j >= b.JstrP)
    GF(Hvom)[I3(i, j, k)] = 0.5 * (hz + Hz[I3(i, j - 1, k)]) * v[I3(i, j, k)] * GF(om_v)[I2(i, j)];
}

extern "C" int roms_hip_set_massflux(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_set_massflux");
  if (rc) return rc;
  ScopedTimer tm("set_massflux");
  const roms_bounds_t &b = g_ctx.b;
  dim3 grid = grid2d(b.IendT - b.IstrT + 1, b.JendT - b.JstrT + 1);
  grid.z = b.N;
  hipLaunchKernelGGL(k_set_massflux, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs);
  KERNEL_CHECK("k_set_massflux");
  halo_batch_begin();
  halo_exchange3d(GT_U, b.N, g_ctx.dev[FID_Huon]);
  halo_exchange3d(GT_V, b.N, g_ctx.dev[FID_Hvom]);
  return halo_batch_end();
}

// ---------------------------------------------------------------------------
// omega_tile -- ROMS/Nonlinear/omega.F:73-229.  One thread per column:
// bottom-up integration of the flux divergence, then removal of the
// moving-surface part.  The running W column stays in registers (template on
// N) so that W is written exactly once: read Huon,Hvom,z_w + write W.
// ---------------------------------------------------------------------------
template <int NMAX>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_omega(const RomsDev *__restrict__ c)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const gcd_t Huon = (gcd_t)(c->F.Huon);
  const gcd_t Hvom = (gcd_t)(c->F.Hvom);
  const gcd_t z_w = (gcd_t)(c->F.z_w);
  const gd_t W = (gd_t)(c->F.W);
  double w[NMAX + 1];
  w[0] = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      w[k] = w[k - 1] - (Huon[I3(i + 1, j, k)] - Huon[I3(i, j, k)] +
                         Hvom[I3(i, j + 1, k)] - Hvom[I3(i, j, k)]);
    }
  }
  const double zw0 = z_w[I3W(i, j, 0)];
  double wN = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) if (k == N) wN = w[k];
  const double wrk = wN / (z_w[I3W(i, j, N)] - zw0);
  W[I3W(i, j, 0)] = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k < N) W[I3W(i, j, k)] = w[k] - wrk * (z_w[I3W(i, j, k)] - zw0);
  }
  W[I3W(i, j, N)] = 0.0;
}

extern "C" int roms_hip_omega(const roms_step_idx_t *s)
{
  (void)s;
  int rc = roms_entry_check("roms_hip_omega");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  ScopedTimer tm("omega");
  const roms_bounds_t &b = g_ctx.b;
  dim3 grid = grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1);
  if (b.N <= 16) hipLaunchKernelGGL(k_omega<16>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc);
  else if (b.N <= 32) hipLaunchKernelGGL(k_omega<32>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc);
  else hipLaunchKernelGGL(k_omega<ROMS_MAXN>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc);
  KERNEL_CHECK("k_omega");
  return bc_w3d(g_ctx.dev[FID_W]);
}

// ---------------------------------------------------------------------------
// set_zeta_tile -- ROMS/Nonlinear/set_zeta.F:59-129
// ---------------------------------------------------------------------------
__global__ void k_set_zeta(const RomsDev *__restrict__ c)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrR + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrR + XB.y * BLK_Y + threadIdx.y;
  if (i > b.IendR || j > b.JendR) return;
  const double z = GF(Zt_avg1)[I2(i, j)];
  GF(zeta)[I2(i, j)] = z;
  GF(zeta)[I2(i, j) + nij] = z;
}

extern "C" int roms_hip_set_zeta(const roms_step_idx_t *s)
{
  (void)s;
  int rc = roms_entry_check("roms_hip_set_zeta");
  if (rc) return rc;
  ScopedTimer tm("set_zeta");
  const roms_bounds_t &b = g_ctx.b;
  hipLaunchKernelGGL(k_set_zeta, grid2d(b.IendR - b.IstrR + 1, b.JendR - b.JstrR + 1), block2d(), 0,
                     g_ctx.stream, g_ctx.devc);
  KERNEL_CHECK("k_set_zeta");
  // both time levels travel in one exchange (mp_exchange2d Nvar=2, set_zeta.F:118)
  return halo_exchange3d(GT_R, 2, g_ctx.dev[FID_zeta]);
}

// ---------------------------------------------------------------------------
// set_depth_tile -- ROMS/Nonlinear/set_depth.F:82-300.  One thread per column;
// z_w(k-1) is carried in a register.  Writes z_w, z_r, Hz: 24 B per cell.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_set_depth(const RomsDev *__restrict__ c)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrT + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrT + XB.y * BLK_Y + threadIdx.y;
  if (i > b.IendT || j > b.JendT) return;
  const roms_params_t &p = c->p;
  const double hc = p.hc;
  const double hwater = GF(h)[I2(i, j)];
  const double zt = GF(Zt_avg1)[I2(i, j)];
  const gd_t z_w = (gd_t)(c->F.z_w);
  const gd_t z_r = (gd_t)(c->F.z_r);
  const gd_t Hz = (gd_t)(c->F.Hz);
  double zw_prev = -hwater;
  z_w[I3W(i, j, 0)] = zw_prev;
  if (p.Vtransform == 1) {
    const double hinv = 1.0 / hwater;
    for (int k = 1; k <= N; k++) {
      const double cff_r = hc * (p.sc_r[k] - p.Cs_r[k]);
      const double cff_w = hc * (p.sc_w[k] - p.Cs_w[k]);
      const double z_w0 = cff_w + p.Cs_w[k] * hwater;
      const double zw = z_w0 + zt * (1.0 + z_w0 * hinv);
      const double z_r0 = cff_r + p.Cs_r[k] * hwater;
      z_w[I3W(i, j, k)] = zw;
      z_r[I3(i, j, k)] = z_r0 + zt * (1.0 + z_r0 * hinv);
      Hz[I3(i, j, k)] = zw - zw_prev;
      zw_prev = zw;
    }
  } else {
    const double hinv = 1.0 / (hc + hwater);
    for (int k = 1; k <= N; k++) {
      const double cff_r = hc * p.sc_r[k];
      const double cff_w = hc * p.sc_w[k];
      const double cff2_r = (cff_r + p.Cs_r[k] * hwater) * hinv;
      const double cff2_w = (cff_w + p.Cs_w[k] * hwater) * hinv;
      const double zw = zt + (zt + hwater) * cff2_w;
      z_w[I3W(i, j, k)] = zw;
      z_r[I3(i, j, k)] = zt + (zt + hwater) * cff2_r;
      Hz[I3(i, j, k)] = zw - zw_prev;
      zw_prev = zw;
    }
  }
}

extern "C" int roms_hip_set_depth(const roms_step_idx_t *s)
{
  (void)s;
  int rc = roms_entry_check("roms_hip_set_depth");
  if (rc) return rc;
  ScopedTimer tm("set_depth");
  const roms_bounds_t &b = g_ctx.b;
  hipLaunchKernelGGL(k_set_depth, grid2d(b.IendT - b.IstrT + 1, b.JendT - b.JstrT + 1), block2d(), 0,
                     g_ctx.stream, g_ctx.devc);
  KERNEL_CHECK("k_set_depth");
  halo_batch_begin();
  halo_exchange2d(GT_R, g_ctx.dev[FID_h]);
  halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_z_w]);
  halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_z_r]);
  halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_Hz]);
  return halo_batch_end();
}
