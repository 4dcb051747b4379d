// k_rhs3d.hip -- right-hand side of the 3-D momentum equations, rhs3d_tile
// (ROMS/Nonlinear/rhs3d.F:174-1673): Coriolis (:467-505), curvilinear metric
// terms (:509-560), third-order upstream-biased horizontal advection
// (UV_U3HADVECTION, :596-982), fourth-order centred vertical advection
// (:1009-1260) and the vertical integral rufrc/rvfrc (:1560-1660); plus the
// rhs3d(ng,tile) driver (rhs3d.F:25-170).
//
// One thread per (i,j) column sweeps k upward once.  ru and rv of the column
// are read-modify-written once per level, the k-window of u, v for the vertical
// stencil is carried in registers, and the vertical sums accumulate in
// registers, so rufrc/rvfrc cost no extra pass.  Horizontal neighbours
// (2-point halo in i and j) are re-read by adjacent lanes/rows and are served
// by L1/L2 (round 1; LDS tiling of these 8 planes is the next optimisation).
// Algorithmic traffic: read u,v,Hz,Huon,Hvom,W, read+write ru,rv = 10 passes.
#include "roms_dev.h"

int roms_entry_check(const char *name);
int roms_launch_rhs3d_lds(int nrhs);

namespace {

struct Lvl {           // pointers to one k-level of the inputs
  const double *u, *v, *Huon, *Hvom, *Hz;
  long ni;
  bool s_edge, n_edge, w_edge, e_edge;   // physical, non-periodic edges
  int Istr, Iend, Jstr, Jend, LBi, LBj;
  __device__ __forceinline__ long at(int i, int j) const { return (long)(i - LBi) + (long)(j - LBj) * ni; }
};

#define Gadv (-0.25)

__device__ __forceinline__ double uxx_at(const Lvl &L, int i, int j)
{
  if (L.w_edge && i == L.Istr) i = L.Istr + 1;           // rhs3d.F:668-676
  if (L.e_edge && i == L.Iend + 1) i = L.Iend;
  const long a = L.at(i, j);
  return L.u[a - 1] - 2.0 * L.u[a] + L.u[a + 1];
}
__device__ __forceinline__ double Huxx_at(const Lvl &L, int i, int j)
{
  if (L.w_edge && i == L.Istr) i = L.Istr + 1;
  if (L.e_edge && i == L.Iend + 1) i = L.Iend;
  const long a = L.at(i, j);
  return L.Huon[a - 1] - 2.0 * L.Huon[a] + L.Huon[a + 1];
}
__device__ __forceinline__ double uee_at(const Lvl &L, int i, int j)
{
  if (L.s_edge && j == L.Jstr - 1) j = L.Jstr;           // rhs3d.F:717-733
  if (L.n_edge && j == L.Jend + 1) j = L.Jend;
  const long a = L.at(i, j);
  return L.u[a - L.ni] - 2.0 * L.u[a] + L.u[a + L.ni];
}
__device__ __forceinline__ double vxx_at(const Lvl &L, int i, int j)
{
  if (L.w_edge && i == L.Istr - 1) i = L.Istr;           // rhs3d.F:770-786
  if (L.e_edge && i == L.Iend + 1) i = L.Iend;
  const long a = L.at(i, j);
  return L.v[a - 1] - 2.0 * L.v[a] + L.v[a + 1];
}
__device__ __forceinline__ double vee_at(const Lvl &L, int i, int j)
{
  if (L.s_edge && j == L.Jstr) j = L.Jstr + 1;           // rhs3d.F:820-838
  if (L.n_edge && j == L.Jend + 1) j = L.Jend;
  const long a = L.at(i, j);
  return L.v[a - L.ni] - 2.0 * L.v[a] + L.v[a + L.ni];
}
__device__ __forceinline__ double Hvee_at(const Lvl &L, int i, int j)
{
  if (L.s_edge && j == L.Jstr) j = L.Jstr + 1;
  if (L.n_edge && j == L.Jend + 1) j = L.Jend;
  const long a = L.at(i, j);
  return L.Hvom[a - L.ni] - 2.0 * L.Hvom[a] + L.Hvom[a + L.ni];
}
__device__ __forceinline__ double Hvxx_at(const Lvl &L, int i, int j)
{
  const long a = L.at(i, j);
  return L.Hvom[a - 1] - 2.0 * L.Hvom[a] + L.Hvom[a + 1];
}
__device__ __forceinline__ double Huee_at(const Lvl &L, int i, int j)
{
  const long a = L.at(i, j);
  return L.Huon[a - L.ni] - 2.0 * L.Huon[a] + L.Huon[a + L.ni];
}

// UFx(i,j): xi-flux of u-momentum at rho-point (i,j), rhs3d.F:688-704
__device__ __forceinline__ double UFx_at(const Lvl &L, int i, int j)
{
  const long a = L.at(i, j);
  const double cff1 = L.u[a] + L.u[a + 1];
  const double cff = (cff1 > 0.0) ? uxx_at(L, i, j) : uxx_at(L, i + 1, j);
  return 0.25 * (cff1 + Gadv * cff) *
         (L.Huon[a] + L.Huon[a + 1] + Gadv * 0.5 * (Huxx_at(L, i, j) + Huxx_at(L, i + 1, j)));
}
// UFe(i,j): eta-flux of u-momentum at psi-point (i,j), rhs3d.F:741-757
__device__ __forceinline__ double UFe_at(const Lvl &L, int i, int j)
{
  const long a = L.at(i, j);
  const double cff1 = L.u[a] + L.u[a - L.ni];
  const double cff2 = L.Hvom[a] + L.Hvom[a - 1];
  const double cff = (cff2 > 0.0) ? uee_at(L, i, j - 1) : uee_at(L, i, j);
  return 0.25 * (cff1 + Gadv * cff) * (cff2 + Gadv * 0.5 * (Hvxx_at(L, i, j) + Hvxx_at(L, i - 1, j)));
}
// VFx(i,j): xi-flux of v-momentum at psi-point (i,j), rhs3d.F:794-810
__device__ __forceinline__ double VFx_at(const Lvl &L, int i, int j)
{
  const long a = L.at(i, j);
  const double cff1 = L.v[a] + L.v[a - 1];
  const double cff2 = L.Huon[a] + L.Huon[a - L.ni];
  const double cff = (cff2 > 0.0) ? vxx_at(L, i - 1, j) : vxx_at(L, i, j);
  return 0.25 * (cff1 + Gadv * cff) * (cff2 + Gadv * 0.5 * (Huee_at(L, i, j) + Huee_at(L, i, j - 1)));
}
// VFe(i,j): eta-flux of v-momentum at rho-point (i,j), rhs3d.F:846-862
__device__ __forceinline__ double VFe_at(const Lvl &L, int i, int j)
{
  const long a = L.at(i, j);
  const double cff1 = L.v[a] + L.v[a + L.ni];
  const double cff = (cff1 > 0.0) ? vee_at(L, i, j) : vee_at(L, i, j + 1);
  return 0.25 * (cff1 + Gadv * cff) *
         (L.Hvom[a] + L.Hvom[a + L.ni] + Gadv * 0.5 * (Hvee_at(L, i, j) + Hvee_at(L, i, j + 1)));
}
// Coriolis / curvilinear cell terms at rho-point (i,j): returns {UFx, VFe}
__device__ __forceinline__ void cor_at(const Lvl &L, const double *fomn, int i, int j, double &ufx, double &vfe)
{
  const long a = L.at(i, j);
  const double cff = 0.5 * L.Hz[a] * fomn[a];
  ufx = cff * (L.v[a] + L.v[a + L.ni]);
  vfe = cff * (L.u[a] + L.u[a + 1]);
}
__device__ __forceinline__ void curv_at(const Lvl &L, const double *dndx, const double *dmde, int i, int j,
                                        double &ufx, double &vfe)
{
  const long a = L.at(i, j);
  const double cff1 = 0.5 * (L.v[a] + L.v[a + L.ni]);
  const double cff2 = 0.5 * (L.u[a] + L.u[a + 1]);
  const double cff3 = cff1 * dndx[a];
  const double cff4 = cff2 * dmde[a];
  const double cff = L.Hz[a] * (cff3 - cff4);
  ufx = cff * cff1;
  vfe = cff * cff2;
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_rhs3d(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const int i = b.Istr + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const roms_params_t &p = c->p;
  const bool do_u = i >= b.IstrU, do_v = j >= b.JstrV;
  const double *__restrict__ ug = c->F.u + (long)(nrhs - 1) * n3r;
  const double *__restrict__ vg = c->F.v + (long)(nrhs - 1) * n3r;
  const double *__restrict__ Wg = c->F.W;
  double *__restrict__ ru = c->F.ru + (long)(nrhs - 1) * n3w;
  double *__restrict__ rv = c->F.rv + (long)(nrhs - 1) * n3w;
  const long c0 = I2(i, j);
  Lvl L;
  L.ni = ni; L.LBi = LBi; L.LBj = LBj;
  L.Istr = b.Istr; L.Iend = b.Iend; L.Jstr = b.Jstr; L.Jend = b.Jend;
  L.s_edge = b.south_edge && !b.NSperiodic; L.n_edge = b.north_edge && !b.NSperiodic;
  L.w_edge = b.west_edge && !b.EWperiodic;  L.e_edge = b.east_edge && !b.EWperiodic;
  const bool cor = p.uv_cor != 0, curv = p.curvgrid != 0 && p.uv_adv != 0, adv = p.uv_adv != 0;

  // k-windows for the vertical stencil
  double u_m1 = 0.0, u_0 = ug[c0], u_p1 = ug[c0 + nij], u_p2;
  double v_m1 = 0.0, v_0 = vg[c0], v_p1 = vg[c0 + nij], v_p2;
  double FCu_prev = 0.0, FCv_prev = 0.0;
  double sum_u = 0.0, sum_v = 0.0;
  const double c9 = 9.0 / 16.0, c1 = 1.0 / 16.0;

  for (int k = 1; k <= N; k++) {
    const long koff = (long)(k - 1) * nij;
    L.u = ug + koff; L.v = vg + koff;
    L.Huon = c->F.Huon + koff; L.Hvom = c->F.Hvom + koff; L.Hz = c->F.Hz + koff;
    u_p2 = (k + 2 <= N) ? ug[c0 + koff + 2 * nij] : 0.0;
    v_p2 = (k + 2 <= N) ? vg[c0 + koff + 2 * nij] : 0.0;
    const long cw = c0 + (long)k * nij;          // W(i,j,k), ru(i,j,k)
    double ruv = do_u ? ru[cw] : 0.0;
    double rvv = do_v ? rv[cw] : 0.0;
    if (cor) {
      double a0, b0, a1, b1, a2, b2;
      cor_at(L, c->F.fomn, i, j, a0, b0);
      if (do_u) { cor_at(L, c->F.fomn, i - 1, j, a1, b1); ruv = ruv + 0.5 * (a0 + a1); }
      if (do_v) { cor_at(L, c->F.fomn, i, j - 1, a2, b2); rvv = rvv - 0.5 * (b0 + b2); }
    }
    if (curv) {
      double a0, b0, a1, b1, a2, b2;
      curv_at(L, c->F.dndx, c->F.dmde, i, j, a0, b0);
      if (do_u) { curv_at(L, c->F.dndx, c->F.dmde, i - 1, j, a1, b1); ruv = ruv + 0.5 * (a0 + a1); }
      if (do_v) { curv_at(L, c->F.dndx, c->F.dmde, i, j - 1, a2, b2); rvv = rvv - 0.5 * (b0 + b2); }
    }
    if (adv) {
      if (do_u) {
        const double cff1 = UFx_at(L, i, j) - UFx_at(L, i - 1, j);
        const double cff2 = UFe_at(L, i, j + 1) - UFe_at(L, i, j);
        ruv = ruv - (cff1 + cff2);
      }
      if (do_v) {
        const double cff1 = VFx_at(L, i + 1, j) - VFx_at(L, i, j);
        const double cff2 = VFe_at(L, i, j) - VFe_at(L, i, j - 1);
        rvv = rvv - (cff1 + cff2);
      }
      // vertical advection, rhs3d.F:1177-1330
      double FCu = 0.0, FCv = 0.0;
      if (k < N) {
        if (do_u) {
          const double um = (k == 1) ? u_0 : u_m1;
          const double up = (k == N - 1) ? u_p1 : u_p2;
          FCu = (c9 * (u_0 + u_p1) - c1 * (um + up)) *
                (c9 * (Wg[cw] + Wg[cw - 1]) - c1 * (Wg[cw + 1] + Wg[cw - 2]));
        }
        if (do_v) {
          const double vm = (k == 1) ? v_0 : v_m1;
          const double vp = (k == N - 1) ? v_p1 : v_p2;
          FCv = (c9 * (v_0 + v_p1) - c1 * (vm + vp)) *
                (c9 * (Wg[cw] + Wg[cw - ni]) - c1 * (Wg[cw + ni] + Wg[cw - 2 * ni]));
        }
      }
      ruv = ruv - (FCu - FCu_prev);
      rvv = rvv - (FCv - FCv_prev);
      FCu_prev = FCu; FCv_prev = FCv;
    }
    if (do_u) { ru[cw] = ruv; sum_u = (k == 1) ? ruv : sum_u + ruv; }
    if (do_v) { rv[cw] = rvv; sum_v = (k == 1) ? rvv : sum_v + rvv; }
    u_m1 = u_0; u_0 = u_p1; u_p1 = u_p2;
    v_m1 = v_0; v_0 = v_p1; v_p1 = v_p2;
  }
  if (do_u) {
    const double cff = c->F.om_u[c0] * c->F.on_u[c0];
    const double cff1 = c->F.sustr[c0] * cff;
    const double cff2 = -c->F.bustr[c0] * cff;
    c->F.rufrc[c0] = sum_u + cff1 + cff2;
  }
  if (do_v) {
    const double cff = c->F.om_v[c0] * c->F.on_v[c0];
    const double cff1 = c->F.svstr[c0] * cff;
    const double cff2 = -c->F.bvstr[c0] * cff;
    c->F.rvfrc[c0] = sum_v + cff1 + cff2;
  }
}

}  // namespace

extern "C" int roms_hip_rhs3d_tile(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_rhs3d_tile");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  ScopedTimer tm("rhs3d_tile");
  const roms_bounds_t &b = g_ctx.b;
  if (b.N < 4) return roms_fail("roms_hip_rhs3d_tile", "N < 4");
  if (!g_ctx.no_lds_3d) return roms_launch_rhs3d_lds(s->nrhs);      // LDS-staged version (k_rhs3d_lds.hip)
  hipLaunchKernelGGL(k_rhs3d, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, s->nrhs);
  KERNEL_CHECK("k_rhs3d");
  return 0;
}

// rhs3d(ng,tile) -- rhs3d.F:25-170: pre_step3d, prsgrd, t3dmix2, rhs3d_tile, uv3dmix2
extern "C" int roms_hip_rhs3d(const roms_step_idx_t *s)
{
  int rc;
  if ((rc = roms_hip_pre_step3d(s))) return rc;
  if ((rc = roms_hip_prsgrd(s))) return rc;
  if (g_ctx.p.ts_dif2 && (rc = roms_hip_t3dmix2(s))) return rc;
  if ((rc = roms_hip_rhs3d_tile(s))) return rc;
  if (g_ctx.p.uv_vis2 && (rc = roms_hip_uv3dmix2(s))) return rc;
  return 0;
}
