// k_ini.hip -- ini_zeta_tile and ini_fields_tile (ROMS/Nonlinear/ini_fields.F:836-1137, :106-777): what main3d
// does on the first time step before the first set_massflux (main3d.F:269-283) -- the other time levels are
// loaded from the initial state with the MASKING multiplies, the lateral boundary conditions are applied, ubar and
// vbar become the vertical means of u and v, Zt_avg1 the initial free surface.  SOLVE3D, no PERFECT_RESTART, no
// WET_DRY.  Run once per simulation: plain kernels, one thread per column.
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

struct IniArgs { int kstp, knew, nstp, nnew, i0, i1, j0, j1; };

// ini_zeta_tile, :945-968 and :1062-1068
__global__ void k_ini_zeta(const RomsDev *__restrict__ c, IniArgs a)
{
  DEV_PROLOGUE(c)
  const int i = b.IstrT + blockIdx.x * BLK_X + threadIdx.x, j = b.JstrT + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendT || j > b.JendT) return;
  const long q = I2(i, j);
  if (i >= a.i0 && i <= a.i1 && j >= a.j0 && j <= a.j1) {
    double cff1 = c->F.zeta[q + (long)(a.kstp - 1) * nij];
    if (c->p.masking) cff1 = cff1 * c->F.rmask[q];
    if (c->p.wet_dry && cff1 <= (c->p.Dcrit - c->F.h[q])) cff1 = c->p.Dcrit - c->F.h[q];   // WET_DRY, ini_fields.F:951-957
    c->F.zeta[q + (long)(a.kstp - 1) * nij] = cff1;
    c->F.zeta[q + (long)(a.knew - 1) * nij] = cff1;
  }
}
__global__ void k_ini_zavg(const RomsDev *__restrict__ c, IniArgs a)
{
  DEV_PROLOGUE(c)
  const int i = b.IstrT + blockIdx.x * BLK_X + threadIdx.x, j = b.JstrT + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendT || j > b.JendT) return;
  c->F.Zt_avg1[I2(i, j)] = c->F.zeta[I2(i, j) + (long)(a.kstp - 1) * nij];
}

// ini_fields_tile, :286-318 (3-D momentum) and :604-622 (tracers): mask, other time level
__global__ void k_ini_3d(const RomsDev *__restrict__ c, IniArgs a)
{
  DEV_PROLOGUE(c)
  const int i = b.IstrB + blockIdx.x * BLK_X + threadIdx.x, j = b.JstrB + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendB || j > b.JendB) return;
  const bool mk = c->p.masking != 0;
  const long q = I2(i, j);
  const double mu = mk ? umaskw(c, q) : 1.0, mv = mk ? vmaskw(c, q) : 1.0, mr = mk ? c->F.rmask[q] : 1.0;   // (+ WET_DRY, ini_fields.F:292, :307)
  const bool do_u = i >= b.IstrM, do_v = j >= b.JstrM;
  for (int k = 1; k <= N; k++) {
    const long q3 = I3(i, j, k);
    if (do_u) {
      double cff1 = c->F.u[q3 + (long)(a.nstp - 1) * n3r];
      if (mk) cff1 = cff1 * mu;
      c->F.u[q3 + (long)(a.nstp - 1) * n3r] = cff1;
      c->F.u[q3 + (long)(a.nnew - 1) * n3r] = cff1;
    }
    if (do_v) {
      double cff2 = c->F.v[q3 + (long)(a.nstp - 1) * n3r];
      if (mk) cff2 = cff2 * mv;
      c->F.v[q3 + (long)(a.nstp - 1) * n3r] = cff2;
      c->F.v[q3 + (long)(a.nnew - 1) * n3r] = cff2;
    }
    for (int itrc = 0; itrc < b.NT; itrc++) {
      double *T = c->F.t + 3L * itrc * n3r;
      double cff1 = T[q3 + (long)(a.nstp - 1) * n3r];
      if (mk) cff1 = cff1 * mr;
      T[q3 + (long)(a.nstp - 1) * n3r] = cff1;
      T[q3 + (long)(a.nnew - 1) * n3r] = cff1;
    }
  }
}

// ini_fields_tile, :380-430: ubar, vbar = vertical means of u, v (sums upwards from k = 1, as the reference)
__global__ void k_ini_bar(const RomsDev *__restrict__ c, IniArgs a)
{
  DEV_PROLOGUE(c)
  const int i = b.IstrB + blockIdx.x * BLK_X + threadIdx.x, j = b.JstrB + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendB || j > b.JendB) return;
  const bool mk = c->p.masking != 0;
  const long q = I2(i, j);
  if (i >= b.IstrM) {
    double DC0 = 0.0, CF0 = 0.0;
    for (int k = 1; k <= N; k++) {
      const double DC = 0.5 * (c->F.Hz[I3(i, j, k)] + c->F.Hz[I3(i - 1, j, k)]);
      DC0 = DC0 + DC;
      CF0 = CF0 + DC * c->F.u[I3(i, j, k) + (long)(a.nstp - 1) * n3r];
    }
    const double cff1 = 1.0 / DC0;
    double cff2 = CF0 * cff1;
    if (mk) cff2 = cff2 * umaskw(c, q);               // (+ WET_DRY, ini_fields.F:398)
    c->F.ubar[q + (long)(a.kstp - 1) * nij] = cff2;
    c->F.ubar[q + (long)(a.knew - 1) * nij] = cff2;
  }
  if (j >= b.JstrM) {
    double DC0 = 0.0, CF0 = 0.0;
    for (int k = 1; k <= N; k++) {
      const double DC = 0.5 * (c->F.Hz[I3(i, j, k)] + c->F.Hz[I3(i, j - 1, k)]);
      DC0 = DC0 + DC;
      CF0 = CF0 + DC * c->F.v[I3(i, j, k) + (long)(a.nstp - 1) * n3r];
    }
    const double cff1 = 1.0 / DC0;
    double cff2 = CF0 * cff1;
    if (mk) cff2 = cff2 * vmaskw(c, q);               // (+ WET_DRY, ini_fields.F:423)
    c->F.vbar[q + (long)(a.kstp - 1) * nij] = cff2;
    c->F.vbar[q + (long)(a.knew - 1) * nij] = cff2;
  }
}

bool any_lbc(int v, int c1, int c2)
{
  for (int sd = 0; sd < 4; sd++) {
    const int c = lbc_code(g_ctx.p, sd, v);
    if (c == c1 || c == c2) return true;
    if (c1 == LBC_RADIATION && c == LBC_RADIATION_NUDGING) return true;     // RadNud sets LBC%radiation too
    if (c2 == LBC_CHAPMAN_IMPLICIT && c == LBC_CHAPMAN_EXPLICIT) return true;   // ini_fields.F:932-934 names both
  }
  return false;
}

IniArgs ini_args(const roms_step_idx_t *s)
{
  IniArgs a{};
  a.kstp = s->kstp; a.knew = s->knew; a.nstp = s->nstp; a.nnew = s->nnew;
  return a;
}

int check_levels(const char *where, const roms_step_idx_t *s)
{
  if (!s || s->kstp < 1 || s->kstp > 3 || s->knew < 1 || s->knew > 3 || s->nstp < 1 || s->nstp > 2 || s->nnew < 1 ||
      s->nnew > 2)
    return roms_fail(where, "time indices out of range");
  return 0;
}

}  // namespace

extern "C" int roms_hip_ini_zeta(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_ini_zeta");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  if ((rc = check_levels("roms_hip_ini_zeta", s))) return rc;
  ScopedTimer tm("ini_zeta");
  const roms_bounds_t &b = g_ctx.b;
  const long nij = (long)(b.UBi - b.LBi + 1) * (long)(b.UBj - b.LBj + 1);
  // radiation / Chapman edges keep their initial boundary values: the whole array is loaded and zetabc is not
  // applied (ini_fields.F:932-944, :971-974)
  const bool open = any_lbc(LBV_ZETA, LBC_RADIATION, LBC_CHAPMAN_IMPLICIT);
  IniArgs a = ini_args(s);
  a.i0 = open ? b.IstrT : b.IstrB; a.i1 = open ? b.IendT : b.IendB;
  a.j0 = open ? b.JstrT : b.JstrB; a.j1 = open ? b.JendT : b.JendB;
  const dim3 grid = grid2d(b.IendT - b.IstrT + 1, b.JendT - b.JstrT + 1);
  hipLaunchKernelGGL(k_ini_zeta, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, a);
  KERNEL_CHECK("k_ini_zeta");
  if (!open) {
    if ((rc = bc_zeta(s->kstp, s))) return rc;
    if ((rc = bc_zeta(s->knew, s))) return rc;
  }
  halo_batch_begin();
  halo_exchange2d(GT_R, g_ctx.dev[FID_zeta] + (long)(s->kstp - 1) * nij);
  if (s->knew != s->kstp) halo_exchange2d(GT_R, g_ctx.dev[FID_zeta] + (long)(s->knew - 1) * nij);
  if ((rc = halo_batch_end())) return rc;
  hipLaunchKernelGGL(k_ini_zavg, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, a);
  KERNEL_CHECK("k_ini_zavg");
  return halo_exchange2d(GT_R, g_ctx.dev[FID_Zt_avg1]);
}

extern "C" int roms_hip_ini_fields(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_ini_fields");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  if ((rc = check_levels("roms_hip_ini_fields", s))) return rc;
  ScopedTimer tm("ini_fields");
  const roms_bounds_t &b = g_ctx.b;
  const long nij = (long)(b.UBi - b.LBi + 1) * (long)(b.UBj - b.LBj + 1), n3r = nij * b.N;
  const IniArgs a = ini_args(s);
  const dim3 grid = grid2d(b.IendB - b.IstrB + 1, b.JendB - b.JstrB + 1);
  double *u = g_ctx.dev[FID_u], *v = g_ctx.dev[FID_v], *t = g_ctx.dev[FID_t];
  hipLaunchKernelGGL(k_ini_3d, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, a);
  KERNEL_CHECK("k_ini_3d");
  // u3dbc / v3dbc on both levels (:322-343), t3dbc per tracer (:626-637); the conditions of different variables
  // touch different arrays, so their order among variables is free
  if ((rc = bc_u3d(s->nstp, s->nstp)) || (rc = bc_v3d(s->nstp, s->nstp))) return rc;
  if ((rc = bc_u3d(s->nnew, s->nstp)) || (rc = bc_v3d(s->nnew, s->nstp))) return rc;
  for (int itrc = 1; itrc <= b.NT; itrc++) {
    if ((rc = bc_t3d(s->nstp, itrc, s->nstp))) return rc;
    if ((rc = bc_t3d(s->nnew, itrc, s->nstp))) return rc;
  }
  halo_batch_begin();
  halo_exchange3d(GT_U, b.N, u + (long)(s->nstp - 1) * n3r);
  halo_exchange3d(GT_V, b.N, v + (long)(s->nstp - 1) * n3r);
  halo_exchange3d(GT_U, b.N, u + (long)(s->nnew - 1) * n3r);
  halo_exchange3d(GT_V, b.N, v + (long)(s->nnew - 1) * n3r);
  for (int itrc = 0; itrc < b.NT; itrc++) {
    halo_exchange3d(GT_R, b.N, t + (3L * itrc + (s->nstp - 1)) * n3r);
    halo_exchange3d(GT_R, b.N, t + (3L * itrc + (s->nnew - 1)) * n3r);
  }
  if ((rc = halo_batch_end())) return rc;
  // vertically integrated momentum (:380-430) from the u, v just completed on the tile's own columns
  hipLaunchKernelGGL(k_ini_bar, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, a);
  KERNEL_CHECK("k_ini_bar");
  if (!any_lbc(LBV_UBAR, LBC_RADIATION, LBC_FLATHER) && !any_lbc(LBV_VBAR, LBC_RADIATION, LBC_FLATHER)) {   // :434-460
    // (with the step's barotropic indices, as the reference's u2dbc_tile reads them from mod_stepping: a Shchepetkin
    // edge is not in the exclusion list above)
    if ((rc = bc_u2d(s->kstp, s)) || (rc = bc_v2d(s->kstp, s))) return rc;
    if ((rc = bc_u2d(s->knew, s)) || (rc = bc_v2d(s->knew, s))) return rc;
  }
  halo_batch_begin();
  halo_exchange2d(GT_U, g_ctx.dev[FID_ubar] + (long)(s->kstp - 1) * nij);
  halo_exchange2d(GT_V, g_ctx.dev[FID_vbar] + (long)(s->kstp - 1) * nij);
  if (s->knew != s->kstp) {
    halo_exchange2d(GT_U, g_ctx.dev[FID_ubar] + (long)(s->knew - 1) * nij);
    halo_exchange2d(GT_V, g_ctx.dev[FID_vbar] + (long)(s->knew - 1) * nij);
  }
  return halo_batch_end();
}
