// k_rhs3d.hip -- right-hand side of the 3-D momentum equations, rhs3d_tile
// (ROMS/Nonlinear/rhs3d.F:174-1673): Coriolis (:467-505), curvilinear metric
// terms (:509-560), third-order upstream-biased horizontal advection
// (UV_U3HADVECTION, :596-982), fourth-order centred vertical advection
// (:1009-1260) and the vertical integral rufrc/rvfrc (:1560-1660); plus the
// rhs3d(ng,tile) driver (rhs3d.F:25-170).
//
// A 64 x 4 workgroup sweeps its columns upward level by level.  For every
// level the five fields the horizontal operators read -- u, v, Huon, Hvom
// (2-point C-grid halo) and Hz -- are staged into LDS once (68 x 8 doubles per
// field, 21.8 KB per workgroup) and the third-order upstream fluxes, Coriolis
// and curvilinear terms are evaluated from LDS.  HBM/L2 sees each plane ~1.5x
// (halo rows) instead of the ~12 re-reads per point of the direct version.
// The k-window of u, v for the 4th-order vertical advection stays in VGPRs and
// the vertical sums rufrc/rvfrc accumulate in registers.
#include "roms_dev.h"

namespace {

#define TP (BLK_X + 4)
#define TJ (BLK_Y + 4)
#define Gadv (-0.25)

struct T3 {
  const double *u, *v, *Hu, *Hv, *Hz;      // LDS tiles
  int i0, j0;
  int Istr, Iend, Jstr, Jend;
  bool s_edge, n_edge, w_edge, e_edge;
  __device__ __forceinline__ int at(int i, int j) const { return (i - i0) + (j - j0) * TP; }
};

__device__ __forceinline__ double d2x(const double *f, int a) { return f[a - 1] - 2.0 * f[a] + f[a + 1]; }
__device__ __forceinline__ double d2y(const double *f, int a) { return f[a - TP] - 2.0 * f[a] + f[a + TP]; }

__device__ __forceinline__ int ex_uxx(const T3 &L, int i) {           // rhs3d.F:668-685
  if (L.w_edge && i == L.Istr) return L.Istr + 1;
  if (L.e_edge && i == L.Iend + 1) return L.Iend;
  return i;
}
__device__ __forceinline__ int ey_uee(const T3 &L, int j) {           // :717-733
  if (L.s_edge && j == L.Jstr - 1) return L.Jstr;
  if (L.n_edge && j == L.Jend + 1) return L.Jend;
  return j;
}
__device__ __forceinline__ int ex_vxx(const T3 &L, int i) {           // :770-786
  if (L.w_edge && i == L.Istr - 1) return L.Istr;
  if (L.e_edge && i == L.Iend + 1) return L.Iend;
  return i;
}
__device__ __forceinline__ int ey_vee(const T3 &L, int j) {           // :820-838
  if (L.s_edge && j == L.Jstr) return L.Jstr + 1;
  if (L.n_edge && j == L.Jend + 1) return L.Jend;
  return j;
}

__device__ __forceinline__ double UFx_at(const T3 &L, int i, int j)   // :688-704
{
  const int a = L.at(i, j);
  const double cff1 = L.u[a] + L.u[a + 1];
  const int ia = ex_uxx(L, i), ib = ex_uxx(L, i + 1);
  const double cff = (cff1 > 0.0) ? d2x(L.u, L.at(ia, j)) : d2x(L.u, L.at(ib, j));
  return 0.25 * (cff1 + Gadv * cff) *
         (L.Hu[a] + L.Hu[a + 1] + Gadv * 0.5 * (d2x(L.Hu, L.at(ia, j)) + d2x(L.Hu, L.at(ib, j))));
}
__device__ __forceinline__ double UFe_at(const T3 &L, int i, int j)   // :741-757
{
  const int a = L.at(i, j);
  const double cff1 = L.u[a] + L.u[a - TP];
  const double cff2 = L.Hv[a] + L.Hv[a - 1];
  const double cff = (cff2 > 0.0) ? d2y(L.u, L.at(i, ey_uee(L, j - 1))) : d2y(L.u, L.at(i, ey_uee(L, j)));
  return 0.25 * (cff1 + Gadv * cff) * (cff2 + Gadv * 0.5 * (d2x(L.Hv, a) + d2x(L.Hv, a - 1)));
}
__device__ __forceinline__ double VFx_at(const T3 &L, int i, int j)   // :794-810
{
  const int a = L.at(i, j);
  const double cff1 = L.v[a] + L.v[a - 1];
  const double cff2 = L.Hu[a] + L.Hu[a - TP];
  const double cff = (cff2 > 0.0) ? d2x(L.v, L.at(ex_vxx(L, i - 1), j)) : d2x(L.v, L.at(ex_vxx(L, i), j));
  return 0.25 * (cff1 + Gadv * cff) * (cff2 + Gadv * 0.5 * (d2y(L.Hu, a) + d2y(L.Hu, a - TP)));
}
__device__ __forceinline__ double VFe_at(const T3 &L, int i, int j)   // :846-862
{
  const int a = L.at(i, j);
  const double cff1 = L.v[a] + L.v[a + TP];
  const int ja = ey_vee(L, j), jb = ey_vee(L, j + 1);
  const double cff = (cff1 > 0.0) ? d2y(L.v, L.at(i, ja)) : d2y(L.v, L.at(i, jb));
  return 0.25 * (cff1 + Gadv * cff) *
         (L.Hv[a] + L.Hv[a + TP] + Gadv * 0.5 * (d2y(L.Hv, L.at(i, ja)) + d2y(L.Hv, L.at(i, jb))));
}

__device__ __forceinline__ void rhs3d_lds_body(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  // two LDS buffers: level k+1 is fetched into registers while level k is computed from LDS, and is
  // stored into the other buffer at the top of the next iteration -- one barrier per level, and the
  // global-load latency of a level overlaps the arithmetic of the previous one
  constexpr int TT = TJ * TP;
  __shared__ double sU[2 * TT], sV[2 * TT], sHu[2 * TT], sHv[2 * TT], sHz[2 * TT];
  const roms_params_t &p = c->p;
  const int i0 = b.Istr + XB.x * BLK_X, j0 = b.Jstr + XB.y * BLK_Y;
  const int i = i0 + threadIdx.x, j = j0 + threadIdx.y;
  const bool active = i <= b.Iend && j <= b.Jend;
  const bool do_u = active && i >= b.IstrU, do_v = active && j >= b.JstrV;
  const gcd_t ug = (gcd_t)(c->F.u + (long)(nrhs - 1) * n3r);
  const gcd_t vg = (gcd_t)(c->F.v + (long)(nrhs - 1) * n3r);
  const gcd_t Wg = (gcd_t)(c->F.W);
  const gd_t ru = (gd_t)(c->F.ru + (long)(nrhs - 1) * n3w);
  const gd_t rv = (gd_t)(c->F.rv + (long)(nrhs - 1) * n3w);
  const int ic = active ? i : b.Iend, jc = active ? j : b.Jend;     // clamped for address formation only
  const long c0 = I2(ic, jc);
  T3 L;
  L.i0 = i0 - 2; L.j0 = j0 - 2;
  L.Istr = b.Istr; L.Iend = b.Iend; L.Jstr = b.Jstr; L.Jend = b.Jend;
  L.s_edge = b.south_edge && !b.NSperiodic; L.n_edge = b.north_edge && !b.NSperiodic;
  L.w_edge = b.west_edge && !b.EWperiodic;  L.e_edge = b.east_edge && !b.EWperiodic;
  const bool cor = p.uv_cor != 0, curv = p.curvgrid != 0 && p.uv_adv != 0, adv = p.uv_adv != 0;
  const int t = L.at(i, j);
  const int tid = threadIdx.y * BLK_X + threadIdx.x;
  // per-thread constants of the cell terms
  const double fomn0 = GF(fomn)[c0], fomnw = GF(fomn)[c0 - 1], fomns = GF(fomn)[c0 - ni];
  double dndx0 = 0, dndxw = 0, dndxs = 0, dmde0 = 0, dmdew = 0, dmdes = 0;
  if (curv) {
    dndx0 = GF(dndx)[c0]; dndxw = GF(dndx)[c0 - 1]; dndxs = GF(dndx)[c0 - ni];
    dmde0 = GF(dmde)[c0]; dmdew = GF(dmde)[c0 - 1]; dmdes = GF(dmde)[c0 - ni];
  }
  double u_m1 = 0.0, u_0 = ug[c0], u_p1 = ug[c0 + nij], u_p2;
  double v_m1 = 0.0, v_0 = vg[c0], v_p1 = vg[c0 + nij], v_p2;
  double FCu_prev = 0.0, FCv_prev = 0.0, sum_u = 0.0, sum_v = 0.0;
  const double c9 = 9.0 / 16.0, c1 = 1.0 / 16.0;

  // staging slots of this thread: tile elements tid, tid+256, tid+512 (the tile has 544)
  constexpr int NSLOT = (TT + BLK_X * BLK_Y - 1) / (BLK_X * BLK_Y);
  long gof[NSLOT];
#pragma unroll
  for (int q = 0; q < NSLOT; q++) {
    const int e = tid + q * BLK_X * BLK_Y;
    const int li = e % TP, lj = (e / TP) % TJ;
    int gi = i0 - 2 + li, gj = j0 - 2 + lj;
    gi = gi < b.LBi ? b.LBi : (gi > b.UBi ? b.UBi : gi);
    gj = gj < b.LBj ? b.LBj : (gj > b.UBj ? b.UBj : gj);
    gof[q] = I2(gi, gj);
  }
  const gcd_t gU = (gcd_t)ug, gV = (gcd_t)vg, gHu = (gcd_t)c->F.Huon, gHv = (gcd_t)c->F.Hvom, gHz = (gcd_t)c->F.Hz;
  const gcd_t gW = (gcd_t)Wg;
  const gd_t gru = (gd_t)ru, grv = (gd_t)rv;
  // staged values of the next level (plain arrays and scalars, constant indices only: a struct returned from a
  // conditional load ended up in scratch memory, with a vmcnt(0) wait right behind the loads)
  double Ru[NSLOT], Rv[NSLOT], Rhu[NSLOT], Rhv[NSLOT], Rhz[NSLOT];
  struct Own { double up2, vp2, ru, rv, w0, wm1, wp1, wm2, wmn, wpn, wm2n; };
  auto gload = [&](int k) {
    const long koff = (long)(k - 1) * nij;
#pragma unroll
    for (int q = 0; q < NSLOT; q++) {
      const long g = gof[q] + koff;
      // slots wholly inside the tile load unconditionally; the last one only where it has an element
      if ((q + 1) * BLK_X * BLK_Y <= TT || tid < TT - q * BLK_X * BLK_Y) {
        Ru[q] = gU[g]; Rv[q] = gV[g]; Rhu[q] = gHu[g]; Rhv[q] = gHv[g]; Rhz[q] = gHz[g];
      }
    }
  };
  auto oload = [&](int k, Own &P) {
    const long koff = (long)(k - 1) * nij;
    const long cw = c0 + (long)k * nij;
    const long c2 = c0 + ((k + 2 <= N) ? koff + 2 * nij : koff);      // clamped: unused above N-2
    P.up2 = gU[c2];
    P.vp2 = gV[c2];
    P.ru = do_u ? gru[cw] : 0.0;
    P.rv = do_v ? grv[cw] : 0.0;
    P.w0 = gW[cw];
    // the W stencil of the vertical flux (k < N, u: i-2..i+1, v: j-2..j+1); inactive lanes read their own point
    P.wm1 = do_u ? gW[cw - 1] : P.w0; P.wp1 = do_u ? gW[cw + 1] : P.w0; P.wm2 = do_u ? gW[cw - 2] : P.w0;
    P.wmn = do_v ? gW[cw - ni] : P.w0; P.wpn = do_v ? gW[cw + ni] : P.w0; P.wm2n = do_v ? gW[cw - 2 * ni] : P.w0;
  };
#pragma unroll
  for (int q = 0; q < NSLOT; q++) Ru[q] = Rv[q] = Rhu[q] = Rhv[q] = Rhz[q] = 0.0;
  gload(1);
  Own P;
  oload(1, P);

  for (int k = 1; k <= N; k++) {
    const int buf = (k & 1) * TT;
    double *bU = sU + buf, *bV = sV + buf, *bHu = sHu + buf, *bHv = sHv + buf, *bHz = sHz + buf;
#pragma unroll
    for (int q = 0; q < NSLOT; q++) {
      const int e = tid + q * BLK_X * BLK_Y;
      if (e < TT) { bU[e] = Ru[q]; bV[e] = Rv[q]; bHu[e] = Rhu[q]; bHv[e] = Rhv[q]; bHz[e] = Rhz[q]; }
    }
    __syncthreads();
    const Own Pk = P;
    { const int kn = k < N ? k + 1 : N; gload(kn); oload(kn, P); }   // in flight while level k is computed
    L.u = bU; L.v = bV; L.Hu = bHu; L.Hv = bHv; L.Hz = bHz;
    u_p2 = Pk.up2;
    v_p2 = Pk.vp2;
    const long cw = c0 + (long)k * nij;
    double ruv = Pk.ru;
    double rvv = Pk.rv;
    if (active) {
      if (cor) {
        const double cf0 = 0.5 * bHz[t] * fomn0;
        const double a0 = cf0 * (bV[t] + bV[t + TP]), b0 = cf0 * (bU[t] + bU[t + 1]);
        if (do_u) {
          const double cf1 = 0.5 * bHz[t - 1] * fomnw;
          const double a1 = cf1 * (bV[t - 1] + bV[t - 1 + TP]);
          ruv = ruv + 0.5 * (a0 + a1);
        }
        if (do_v) {
          const double cf2 = 0.5 * bHz[t - TP] * fomns;
          const double b2 = cf2 * (bU[t - TP] + bU[t - TP + 1]);
          rvv = rvv - 0.5 * (b0 + b2);
        }
      }
      if (curv) {
        auto cell = [&](int q, double dn, double dm, double &ufx, double &vfe) {
          const double cff1 = 0.5 * (bV[q] + bV[q + TP]);
          const double cff2 = 0.5 * (bU[q] + bU[q + 1]);
          const double cff3 = cff1 * dn;
          const double cff4 = cff2 * dm;
          const double cff = bHz[q] * (cff3 - cff4);
          ufx = cff * cff1;
          vfe = cff * cff2;
        };
        double a0, b0, a1, b1;
        cell(t, dndx0, dmde0, a0, b0);
        if (do_u) { cell(t - 1, dndxw, dmdew, a1, b1); ruv = ruv + 0.5 * (a0 + a1); }
        if (do_v) { cell(t - TP, dndxs, dmdes, a1, b1); rvv = rvv - 0.5 * (b0 + b1); }
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
        double FCu = 0.0, FCv = 0.0;
        if (k < N) {
          if (do_u) {
            const double um = (k == 1) ? u_0 : u_m1;
            const double up = (k == N - 1) ? u_p1 : u_p2;
            FCu = (c9 * (u_0 + u_p1) - c1 * (um + up)) *
                  (c9 * (Pk.w0 + Pk.wm1) - c1 * (Pk.wp1 + Pk.wm2));
          }
          if (do_v) {
            const double vm = (k == 1) ? v_0 : v_m1;
            const double vp = (k == N - 1) ? v_p1 : v_p2;
            FCv = (c9 * (v_0 + v_p1) - c1 * (vm + vp)) *
                  (c9 * (Pk.w0 + Pk.wmn) - c1 * (Pk.wpn + Pk.wm2n));
          }
        }
        ruv = ruv - (FCu - FCu_prev);
        rvv = rvv - (FCv - FCv_prev);
        FCu_prev = FCu; FCv_prev = FCv;
      }
      if (do_u) { gru[cw] = ruv; sum_u = (k == 1) ? ruv : sum_u + ruv; }
      if (do_v) { grv[cw] = rvv; sum_v = (k == 1) ? rvv : sum_v + rvv; }
    }
    u_m1 = u_0; u_0 = u_p1; u_p1 = u_p2;
    v_m1 = v_0; v_0 = v_p1; v_p1 = v_p2;
  }
  if (do_u) {
    const double cff = GF(om_u)[c0] * GF(on_u)[c0];
    const double cff1 = GF(sustr)[c0] * cff;
    const double cff2 = -GF(bustr)[c0] * cff;
    GF(rufrc)[c0] = sum_u + cff1 + cff2;
  }
  if (do_v) {
    const double cff = GF(om_v)[c0] * GF(on_v)[c0];
    const double cff1 = GF(svstr)[c0] * cff;
    const double cff2 = -GF(bvstr)[c0] * cff;
    GF(rvfrc)[c0] = sum_v + cff1 + cff2;
  }
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_rhs3d_lds(const RomsDev *__restrict__ c, int nrhs) { rhs3d_lds_body(c, nrhs); }

}  // namespace

int roms_entry_check(const char *name);

extern "C" int roms_hip_rhs3d_tile(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_rhs3d_tile");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  ScopedTimer tm("rhs3d_tile");
  const roms_bounds_t &b = g_ctx.b;
  if (b.N < 4) return roms_fail("roms_hip_rhs3d_tile", "N < 4");
  hipLaunchKernelGGL(k_rhs3d_lds, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, s->nrhs);
  KERNEL_CHECK("k_rhs3d_lds");
  return 0;
}

// rhs3d(ng,tile) -- rhs3d.F:25-170: pre_step3d, prsgrd, t3dmix2, t3dmix4, rhs3d_tile, uv3dmix2, uv3dmix4
extern "C" int roms_hip_rhs3d(const roms_step_idx_t *s)
{
  int rc;
  if ((rc = roms_hip_pre_step3d(s))) return rc;
  if ((rc = roms_hip_prsgrd(s))) return rc;
  if (g_ctx.p.ts_dif2 && (rc = roms_hip_t3dmix2(s))) return rc;
  if (g_ctx.p.ts_dif4 && (rc = roms_hip_t3dmix4(s))) return rc;
  if ((rc = roms_hip_rhs3d_tile(s))) return rc;
  if (g_ctx.p.uv_vis2 && (rc = roms_hip_uv3dmix2(s))) return rc;
  if (g_ctx.p.uv_vis4 && (rc = roms_hip_uv3dmix4(s))) return rc;
  return 0;
}
