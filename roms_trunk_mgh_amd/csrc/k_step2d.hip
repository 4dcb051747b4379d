// k_step2d.hip -- split-explicit barotropic engine, step2d_tile
// (ROMS/Nonlinear/step2d_LF_AM3.h:137-2528: leap-frog predictor / Adams-Moulton
// corrector) and the LOOP_2D sequencing of main3d.F:592-700.
//
// One step2d call = three kernels over the 2-D tile (all fields together are a
// few MB, i.e. L2/Infinity-Cache resident; the loop is launch-latency bound,
// not HBM bound):
//   k2d_flux   Drhs, DUon, DVom two points into the halo (:509-544)
//   k2d_zeta   fast-time averaging (:614-682) and the free-surface step
//              (:770-868): zeta(knew), rzeta(krhs); zeta_new and zwrk go to
//              device scratch on the extended range so the momentum kernel can
//              use them at i-1 / j-1 without another exchange
//   k2d_mom    pressure gradient with VAR_RHO_2D (:939-1019), 4th-order centred
//              advection (:1079-1283), Coriolis (:1291-1325), curvilinear
//              (:1333-1382), harmonic viscosity (:1394-1471), 2D<->3D coupling
//              (:1884-2065) and the ubar/vbar step (:2098-2255)
// with the reference's boundary-condition and halo calls in between.  The
// reference's ~25 private (IminS:ImaxS,JminS:JmaxS) work arrays become
// registers; only DUon, DVom, zeta_new, zwrk live in device scratch.
#include "roms_dev.h"
#include <cstdlib>

int roms_entry_check(const char *name);
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

// DUon, DVom of level `lev` on the points this tile owns (interior + physical boundary rows); the
// ghost points then come with the end-of-call exchange.  Used inside LOOP_2D on several tiles: the
// fluxes of the NEXT call travel in the same message as zeta, ubar, vbar of this one.
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_flux_own(const RomsDev *__restrict__ c, int lev, double *__restrict__ DUon, double *__restrict__ DVom)
{
  DEV_PROLOGUE(c)
  const int i = b.Istr + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.JstrR + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.JendR) return;
  const gcd_t zeta = (gcd_t)(c->F.zeta + (long)(lev - 1) * nij);
  const gcd_t h = (gcd_t)(c->F.h);
  const long a = I2(i, j);
  // zeta on a closed-wall row is the zero-gradient copy of the adjacent interior row (zetabc.F:48);
  // it is read from that row directly because bc_zeta fills the wall rows of the OWN columns only and
  // the ghost column i-1 has not been exchanged yet
  auto zrow = [&](int jj) {
    if (b.south_edge && !b.NSperiodic && jj < b.Jstr) return b.Jstr;
    if (b.north_edge && !b.NSperiodic && jj > b.Jend) return b.Jend;
    return jj;
  };
  const double Drhs = zeta[I2(i, zrow(j))] + h[a];
  {
    const double cff = 0.5 * GF(on_u)[a];
    const double cff1 = cff * (Drhs + (zeta[I2(i - 1, zrow(j))] + h[a - 1]));
    DUon[a] = GF(ubar)[a + (long)(lev - 1) * nij] * cff1;
  }
  if (j >= b.JstrV - 1 && j >= b.LBj + 1) {
    const double cff = 0.5 * GF(om_v)[a];
    const double cff1 = cff * (Drhs + (zeta[I2(i, zrow(j - 1))] + h[a - ni]));
    DVom[a] = GF(vbar)[a + (long)(lev - 1) * nij] * cff1;
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
  // for bit, what the exchange delivers later, and k2d_flux_own (next call's fluxes) needs it now
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
    zw = 0.5 * (zs[a] + zn);
  } else if (s.predictor) {
    const double cff1 = 2.0 * dtfast;
    const double cff4 = 4.0 / 25.0;
    const double cff5 = 1.0 - 2.0 * cff4;
    zn = zs[a] + pmn_a * pn_a * cff1 * rhs;
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
    zw = cff5 * zn + cff4 * zk[a];
  }
  if (write_scratch) { zeta_new[o] = zn; zwrk[o] = zw; }
  if (write_zeta) GF(zeta)[o + (long)(s.knew - 1) * nij] = zn;
  if (write_rzeta && s.predictor) GF(rzeta)[o + (long)(s.krhs - 1) * nij] = rhs;
}

// ---------------------------------------------------------------- momentum --
struct M2 {
  const double *ub, *vb, *DU, *DV;     // ubar,vbar(krhs), DUon, DVom
  long ni;
  int LBi, LBj, Istr, Iend, Jstr, Jend;
  bool s_edge, n_edge, w_edge, e_edge;
  __device__ __forceinline__ long at(int i, int j) const { return (long)(i - LBi) + (long)(j - LBj) * ni; }
};
#define C6 (1.0 / 6.0)

__device__ __forceinline__ double d2x(const double *f, long a) { return f[a - 1] - 2.0 * f[a] + f[a + 1]; }
__device__ __forceinline__ double d2y(const double *f, long a, long ni) { return f[a - ni] - 2.0 * f[a] + f[a + ni]; }

__device__ __forceinline__ double UFx2(const M2 &m, int i, int j)
{
  int ia = i, ib = i + 1;     // grad/Dgrad indices with the physical-edge rule (:1092-1107)
  if (m.w_edge) { if (ia == m.Istr) ia = m.Istr + 1; if (ib == m.Istr) ib = m.Istr + 1; }
  if (m.e_edge) { if (ia == m.Iend + 1) ia = m.Iend; if (ib == m.Iend + 1) ib = m.Iend; }
  const long a = m.at(i, j);
  return 0.25 * (m.ub[a] + m.ub[a + 1] - C6 * (d2x(m.ub, m.at(ia, j)) + d2x(m.ub, m.at(ib, j)))) *
         (m.DU[a] + m.DU[a + 1] - C6 * (d2x(m.DU, m.at(ia, j)) + d2x(m.DU, m.at(ib, j))));
}
__device__ __forceinline__ double UFe2(const M2 &m, int i, int j)
{
  int ja = j, jb = j - 1;     // grad(i,Jstr-1)=grad(i,Jstr), grad(i,Jend+1)=grad(i,Jend)
  if (m.s_edge) { if (ja == m.Jstr - 1) ja = m.Jstr; if (jb == m.Jstr - 1) jb = m.Jstr; }
  if (m.n_edge) { if (ja == m.Jend + 1) ja = m.Jend; if (jb == m.Jend + 1) jb = m.Jend; }
  const long a = m.at(i, j);
  return 0.25 * (m.ub[a] + m.ub[a - m.ni] - C6 * (d2y(m.ub, m.at(i, ja), m.ni) + d2y(m.ub, m.at(i, jb), m.ni))) *
         (m.DV[a] + m.DV[a - 1] - C6 * (d2x(m.DV, a) + d2x(m.DV, a - 1)));
}
__device__ __forceinline__ double VFx2(const M2 &m, int i, int j)
{
  int ia = i, ib = i - 1;     // grad(Istr-1)=grad(Istr), grad(Iend+1)=grad(Iend)
  if (m.w_edge) { if (ia == m.Istr - 1) ia = m.Istr; if (ib == m.Istr - 1) ib = m.Istr; }
  if (m.e_edge) { if (ia == m.Iend + 1) ia = m.Iend; if (ib == m.Iend + 1) ib = m.Iend; }
  const long a = m.at(i, j);
  return 0.25 * (m.vb[a] + m.vb[a - 1] - C6 * (d2x(m.vb, m.at(ia, j)) + d2x(m.vb, m.at(ib, j)))) *
         (m.DU[a] + m.DU[a - m.ni] - C6 * (d2y(m.DU, a, m.ni) + d2y(m.DU, a - m.ni, m.ni)));
}
__device__ __forceinline__ double VFe2(const M2 &m, int i, int j)
{
  int ja = j, jb = j + 1;     // (Jstr)=(Jstr+1), (Jend+1)=(Jend) for grad and Dgrad
  if (m.s_edge) { if (ja == m.Jstr) ja = m.Jstr + 1; if (jb == m.Jstr) jb = m.Jstr + 1; }
  if (m.n_edge) { if (ja == m.Jend + 1) ja = m.Jend; if (jb == m.Jend + 1) jb = m.Jend; }
  const long a = m.at(i, j);
  return 0.25 * (m.vb[a] + m.vb[a + m.ni] - C6 * (d2y(m.vb, m.at(i, ja), m.ni) + d2y(m.vb, m.at(i, jb), m.ni))) *
         (m.DV[a] + m.DV[a + m.ni] - C6 * (d2y(m.DV, m.at(i, ja), m.ni) + d2y(m.DV, m.at(i, jb), m.ni)));
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k2d_mom(const RomsDev *__restrict__ c, S2 s, const double *__restrict__ DUon, const double *__restrict__ DVom,
        const double *__restrict__ zeta_new, const double *__restrict__ zwrk)
{
  DEV_PROLOGUE(c)
  const roms_params_t &p = c->p;
  // (it,jt) = target point this thread stores to; (i,j) = source point it evaluates.
  // Without source mapping they coincide and the range is the tile interior.
  int it, jt, i, j;
  double fu = 1.0;                 // u2dbc closed-wall factor gamma2 on ghost rows
  bool do_u, do_v, owner = true, v_wall = false;
  if (!s.sm) {
    it = i = b.Istr + blockIdx.x * BLK_X + threadIdx.x;
    jt = j = b.Jstr + blockIdx.y * BLK_Y + threadIdx.y;
    if (i > b.Iend || j > b.Jend) return;
    do_u = i >= b.IstrU;
    do_v = j >= b.JstrV;
  } else {
    it = b.LBi + blockIdx.x * BLK_X + threadIdx.x;
    jt = b.LBj + blockIdx.y * BLK_Y + threadIdx.y;
    if (it > b.Lm + b.NghostPoints || jt > b.UBj) return;
    i = wrap_i(b, it);
    j = jt;
    if (b.south_edge && jt == b.Jstr - 1) { j = b.Jstr; fu = p.gamma2; }
    if (b.north_edge && jt == b.Jend + 1) { j = b.Jend; fu = p.gamma2; }
    if (j < b.Jstr || j > b.Jend) return;
    owner = (it == i) && (jt == j);
    do_u = true;
    do_v = (jt == j) && j >= b.JstrV;
    // v2dbc closed: vbar = 0 on the wall rows Jstr (south) and Jend+1 (north)
    v_wall = (b.south_edge && jt == b.Jstr) || (b.north_edge && jt == b.Jend + 1);
  }
  const long a = I2(i, j);
  const long o = I2(it, jt);
  const gcd_t h = (gcd_t)(c->F.h);
  const gcd_t rhoA = (gcd_t)(c->F.rhoA);
  const gcd_t rhoS = (gcd_t)(c->F.rhoS);
  const gcd_t pm = (gcd_t)(c->F.pm);
  const gcd_t pn = (gcd_t)(c->F.pn);
  const gcd_t zk = (gcd_t)(c->F.zeta + (long)(s.krhs - 1) * nij);
  const gcd_t zs = (gcd_t)(c->F.zeta + (long)(s.kstp - 1) * nij);
  M2 m;
  m.ub = c->F.ubar + (long)(s.krhs - 1) * nij;
  m.vb = c->F.vbar + (long)(s.krhs - 1) * nij;
  m.DU = DUon; m.DV = DVom; m.ni = ni; m.LBi = LBi; m.LBj = LBj;
  m.Istr = b.Istr; m.Iend = b.Iend; m.Jstr = b.Jstr; m.Jend = b.Jend;
  m.s_edge = b.south_edge && !b.NSperiodic; m.n_edge = b.north_edge && !b.NSperiodic;
  m.w_edge = b.west_edge && !b.EWperiodic;  m.e_edge = b.east_edge && !b.EWperiodic;
  const double fac = 1000.0 / p.rho0;
  // ---- pressure gradient, :939-1019 ----
  const double zw0 = zwrk[a];
  const double gz0 = (fac + rhoS[a]) * zw0, gz20 = gz0 * zw0, gsa0 = zw0 * (rhoS[a] - rhoA[a]);
  const double cg = 0.5 * p.g, c3 = 1.0 / 3.0;
  double rhs_u = 0.0, rhs_v = 0.0;
  if (do_u) {
    const long q = a - 1;
    const double zw = zwrk[q];
    const double gz = (fac + rhoS[q]) * zw, gz2 = gz * zw, gsa = zw * (rhoS[q] - rhoA[q]);
    rhs_u = cg * GF(on_u)[a] *
            ((h[q] + h[a]) * (gz - gz0) +
             (h[q] - h[a]) * (gsa + gsa0 + c3 * (rhoA[q] - rhoA[a]) * (zw - zw0)) +
             (gz2 - gz20));
  }
  if (do_v) {
    const long q = a - ni;
    const double zw = zwrk[q];
    const double gz = (fac + rhoS[q]) * zw, gz2 = gz * zw, gsa = zw * (rhoS[q] - rhoA[q]);
    rhs_v = cg * GF(om_v)[a] *
            ((h[q] + h[a]) * (gz - gz0) +
             (h[q] - h[a]) * (gsa + gsa0 + c3 * (rhoA[q] - rhoA[a]) * (zw - zw0)) +
             (gz2 - gz20));
  }
  // ---- advection, :1079-1283 ----
  if (p.uv_adv) {
    if (do_u) {
      const double cff1 = UFx2(m, i, j) - UFx2(m, i - 1, j);
      const double cff2 = UFe2(m, i, j + 1) - UFe2(m, i, j);
      rhs_u = rhs_u - (cff1 + cff2);
    }
    if (do_v) {
      const double cff1 = VFx2(m, i + 1, j) - VFx2(m, i, j);
      const double cff2 = VFe2(m, i, j) - VFe2(m, i, j - 1);
      rhs_v = rhs_v - (cff1 + cff2);
    }
  }
  // total depth at the rho points this thread needs
  const double D0 = zk[a] + h[a], Dw = zk[a - 1] + h[a - 1], Ds = zk[a - ni] + h[a - ni];
  // ---- Coriolis, :1291-1325 ----
  if (p.uv_cor) {
    const gcd_t fomn = (gcd_t)(c->F.fomn);
    const double cf0 = 0.5 * D0 * fomn[a];
    const double UFx0 = cf0 * (m.vb[a] + m.vb[a + ni]);
    const double VFe0 = cf0 * (m.ub[a] + m.ub[a + 1]);
    if (do_u) {
      const double cfw = 0.5 * Dw * fomn[a - 1];
      const double UFxw = cfw * (m.vb[a - 1] + m.vb[a - 1 + ni]);
      rhs_u = rhs_u + 0.5 * (UFx0 + UFxw);
    }
    if (do_v) {
      const double cfs = 0.5 * Ds * fomn[a - ni];
      const double VFes = cfs * (m.ub[a - ni] + m.ub[a - ni + 1]);
      rhs_v = rhs_v - 0.5 * (VFe0 + VFes);
    }
  }
  // ---- curvilinear terms, :1333-1382 ----
  if (p.curvgrid && p.uv_adv) {
    const gcd_t dndx = (gcd_t)c->F.dndx, dmde = (gcd_t)c->F.dmde;
    auto cell = [&](long q, double D, double &ufx, double &vfe) {
      const double cff1 = 0.5 * (m.vb[q] + m.vb[q + ni]);
      const double cff2 = 0.5 * (m.ub[q] + m.ub[q + 1]);
      const double cff3 = cff1 * dndx[q];
      const double cff4 = cff2 * dmde[q];
      const double cff = D * (cff3 - cff4);
      ufx = cff * cff1;
      vfe = cff * cff2;
    };
    double u0, v0, u1, v1;
    cell(a, D0, u0, v0);
    if (do_u) { cell(a - 1, Dw, u1, v1); rhs_u = rhs_u + 0.5 * (u0 + u1); }
    if (do_v) { cell(a - ni, Ds, u1, v1); rhs_v = rhs_v - 0.5 * (v0 + v1); }
  }
  // ---- harmonic viscosity, :1394-1471 ----
  if (p.uv_vis2) {
    const gcd_t visc2_r = (gcd_t)c->F.visc2_r, visc2_p = (gcd_t)c->F.visc2_p;
    const gcd_t pmon_r = (gcd_t)c->F.pmon_r, pnom_r = (gcd_t)c->F.pnom_r, pmon_p = (gcd_t)c->F.pmon_p, pnom_p = (gcd_t)c->F.pnom_p;
    const gcd_t om_r = (gcd_t)c->F.om_r, on_r = (gcd_t)c->F.on_r, om_p = (gcd_t)c->F.om_p, on_p = (gcd_t)c->F.on_p;
    auto Dat = [&](long q) { return zk[q] + h[q]; };
    auto str_r = [&](long q) {       // cff at rho-point q
      return visc2_r[q] * Dat(q) * 0.5 *
             (pmon_r[q] * ((pn[q] + pn[q + 1]) * m.ub[q + 1] - (pn[q - 1] + pn[q]) * m.ub[q]) -
              pnom_r[q] * ((pm[q] + pm[q + ni]) * m.vb[q + ni] - (pm[q - ni] + pm[q]) * m.vb[q]));
    };
    auto str_p = [&](long q) {       // cff at psi-point q
      const double Dp = 0.25 * (Dat(q) + Dat(q - 1) + Dat(q - ni) + Dat(q - 1 - ni));
      return visc2_p[q] * Dp * 0.5 *
             (pmon_p[q] * ((pn[q - ni] + pn[q]) * m.vb[q] - (pn[q - 1 - ni] + pn[q - 1]) * m.vb[q - 1]) +
              pnom_p[q] * ((pm[q - 1] + pm[q]) * m.ub[q] - (pm[q - 1 - ni] + pm[q - ni]) * m.ub[q - ni]));
    };
    const double sr0 = str_r(a), sp0 = str_p(a);
    if (do_u) {
      const double srw = str_r(a - 1), spn = str_p(a + ni);
      const double UFx0 = on_r[a] * on_r[a] * sr0, UFxw = on_r[a - 1] * on_r[a - 1] * srw;
      const double UFe0 = om_p[a] * om_p[a] * sp0, UFen = om_p[a + ni] * om_p[a + ni] * spn;
      const double cff1 = 0.5 * (pn[a - 1] + pn[a]) * (UFx0 - UFxw);
      const double cff2 = 0.5 * (pm[a - 1] + pm[a]) * (UFen - UFe0);
      rhs_u = rhs_u + (cff1 + cff2);
    }
    if (do_v) {
      const double srs = str_r(a - ni), spe = str_p(a + 1);
      const double VFx0 = on_p[a] * on_p[a] * sp0, VFxe = on_p[a + 1] * on_p[a + 1] * spe;
      const double VFe0 = om_r[a] * om_r[a] * sr0, VFes = om_r[a - ni] * om_r[a - ni] * srs;
      const double cff1 = 0.5 * (pn[a - ni] + pn[a]) * (VFxe - VFx0);
      const double cff2 = 0.5 * (pm[a - ni] + pm[a]) * (VFe0 - VFes);
      rhs_v = rhs_v + (cff1 - cff2);
    }
  }
  // ---- coupling between 2-D and 3-D equations, :1884-2065 ----
  if (s.iif == 1 && s.predictor) {
    const gd_t ru_s = (gd_t)(c->F.ru + (long)(s.nstp - 1) * n3w);      // k = 0 plane
    const gd_t rv_s = (gd_t)(c->F.rv + (long)(s.nstp - 1) * n3w);
    const gcd_t ru_n = (gcd_t)(c->F.ru + (long)(s.nnew - 1) * n3w);
    const gcd_t rv_n = (gcd_t)(c->F.rv + (long)(s.nnew - 1) * n3w);
    if (do_u) {
      const double rf = GF(rufrc)[a] - rhs_u;
      if (s.iic == s.ntfirst) rhs_u = rhs_u + rf;
      else if (s.iic == s.ntfirst + 1) rhs_u = rhs_u + 1.5 * rf - 0.5 * ru_n[a];
      else rhs_u = rhs_u + (23.0 / 12.0) * rf - (16.0 / 12.0) * ru_n[a] + (5.0 / 12.0) * ru_s[a];
      if (owner) GF(rufrc)[a] = rf;
      if (owner) ru_s[a] = rf;
    }
    if (do_v) {
      const double rf = GF(rvfrc)[a] - rhs_v;
      if (s.iic == s.ntfirst) rhs_v = rhs_v + rf;
      else if (s.iic == s.ntfirst + 1) rhs_v = rhs_v + 1.5 * rf - 0.5 * rv_n[a];
      else rhs_v = rhs_v + (23.0 / 12.0) * rf - (16.0 / 12.0) * rv_n[a] + (5.0 / 12.0) * rv_s[a];
      if (owner) GF(rvfrc)[a] = rf;
      if (owner) rv_s[a] = rf;
    }
  } else {
    if (do_u) rhs_u = rhs_u + GF(rufrc)[a];
    if (do_v) rhs_v = rhs_v + GF(rvfrc)[a];
  }
  // ---- time step, :2098-2255 ----
  const double dtfast = p.dtfast;
  const double Dn0 = zeta_new[a] + h[a], Dst0 = zs[a] + h[a];
  const int ptsk = 3 - s.kstp;
  const bool am3 = !(s.iif == 1 || s.predictor);
  const double c1 = (s.iif == 1) ? 0.5 * dtfast : dtfast;
  const double a1 = 0.5 * dtfast * 5.0 / 12.0, a2 = 0.5 * dtfast * 8.0 / 12.0, a3 = 0.5 * dtfast * 1.0 / 12.0;
  if (do_u) {
    const long q = a - 1;
    const double cff = (pm[a] + pm[q]) * (pn[a] + pn[q]);
    const double fc = 1.0 / (Dn0 + (zeta_new[q] + h[q]));
    const double us = GF(ubar)[a + (long)(s.kstp - 1) * nij];
    double un;
    if (!am3) un = (us * (Dst0 + (zs[q] + h[q])) + cff * c1 * rhs_u) * fc;
    else un = (us * (Dst0 + (zs[q] + h[q])) +
               cff * (a1 * rhs_u + a2 * GF(rubar)[a + (long)(s.kstp - 1) * nij] -
                      a3 * GF(rubar)[a + (long)(ptsk - 1) * nij])) * fc;
    GF(ubar)[o + (long)(s.knew - 1) * nij] = (fu == 1.0) ? un : fu * un;
    if (s.predictor && owner) GF(rubar)[a + (long)(s.krhs - 1) * nij] = rhs_u;
  }
  if (do_v) {
    const long q = a - ni;
    const double cff = (pm[a] + pm[q]) * (pn[a] + pn[q]);
    const double fc = 1.0 / (Dn0 + (zeta_new[q] + h[q]));
    const double vs = GF(vbar)[a + (long)(s.kstp - 1) * nij];
    double vn;
    if (!am3) vn = (vs * (Dst0 + (zs[q] + h[q])) + cff * c1 * rhs_v) * fc;
    else vn = (vs * (Dst0 + (zs[q] + h[q])) +
               cff * (a1 * rhs_v + a2 * GF(rvbar)[a + (long)(s.kstp - 1) * nij] -
                      a3 * GF(rvbar)[a + (long)(ptsk - 1) * nij])) * fc;
    GF(vbar)[o + (long)(s.knew - 1) * nij] = vn;
    if (s.predictor && owner) GF(rvbar)[a + (long)(s.krhs - 1) * nij] = rhs_v;
  }
  if (v_wall) GF(vbar)[o + (long)(s.knew - 1) * nij] = 0.0;
}

// DUon/DVom scratch already holds the exchanged fluxes of barotropic level g_flux_lev (left there by the
// previous call of the same LOOP_2D)
bool g_flux_ready = false;
int g_flux_lev = 0;
int g_flux_buf = 0;       // which scratch pair holds them

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
  static const bool no_fuse_zeta = getenv("ROMS_HIP_NO_FUSED_ZETA") != nullptr;     // A/B switch
  const bool one_launch = !g_ctx.no_lds_2d && !no_fuse_zeta && s.iif <= p.nfast;     // k2d_mom_lds<true>
  // (the kernels that use ghost threads cannot take the first predictor of a step: there the momentum
  // part read-modify-writes rufrc and ru(:,:,0,nstp) at the source points; the one-launch kernel can)
  const bool sm = b.ntileI * b.ntileJ == 1 && b.EWperiodic && !b.NSperiodic && !g_ctx.no_fused_2d &&
                  (one_launch || !(s.iif == 1 && s.predictor));
  if (sm) {
    s.sm = 1;
    const dim3 full = grid2d(b.UBi - b.LBi + 1, b.UBj - b.LBj + 1);
    if (one_launch) {
      // ONE launch: free surface, fast-time averages and momentum (k2d_mom_lds<true>)
      s.sm = 2;
      return roms_launch_k2d_mom_lds((const int *)&s, nullptr, nullptr, nullptr, nullptr);
    }
    if (!g_ctx.no_lds_2d) {
      // DUon/DVom are evaluated in place by the two remaining kernels: 2 launches per call
      DUon = nullptr;
      DVom = nullptr;
    } else {
      hipLaunchKernelGGL(k2d_flux, grid2d(b.UBi - b.LBi + 1, b.Jendp2 - (b.JstrV - 2) + 1), block2d(), 0,
                         g_ctx.stream, g_ctx.devc, s, DUon, DVom);
      KERNEL_CHECK("k2d_flux");
    }
    hipLaunchKernelGGL(k2d_zeta_sm, full, block2d(), 0, g_ctx.stream, g_ctx.devc, s, (const double *)DUon,
                       (const double *)DVom, zeta_new, zwrk);
    KERNEL_CHECK("k2d_zeta_sm");
    if (s.iif == p.nfast + 1 && s.predictor) {
      if ((rc = halo_exchange2d(GT_R, g_ctx.dev[FID_Zt_avg1]))) return rc;
      if ((rc = halo_exchange2d(GT_U, g_ctx.dev[FID_DU_avg1]))) return rc;
      if ((rc = halo_exchange2d(GT_V, g_ctx.dev[FID_DV_avg1]))) return rc;
    }
    if (s.iif > p.nfast) return 0;
    if (!g_ctx.no_lds_2d) return roms_launch_k2d_mom_lds((const int *)&s, DUon, DVom, zeta_new, zwrk);
    hipLaunchKernelGGL(k2d_mom, full, block2d(), 0, g_ctx.stream, g_ctx.devc, s, (const double *)DUon,
                       (const double *)DVom, (const double *)zeta_new, (const double *)zwrk);
    KERNEL_CHECK("k2d_mom");
    return 0;
  }
  // General path (several tiles; first predictor of a step on one tile).  Messages per call: inside
  // LOOP_2D ONE fused exchange at the end (rzeta, zeta, ubar, vbar of this call + DUon, DVom of the
  // next one, whose krhs is this call's knew); a stand-alone call exchanges its own fluxes first.
  const bool multi = b.ntileI * b.ntileJ > 1;
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
  if (in_loop && multi && !g_ctx.no_lds_2d && s.iif <= p.nfast) {
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
  if (s.iif > p.nfast) return 0;
  if ((rc = bc_zeta(s.knew))) return rc;
  if (!g_ctx.no_lds_2d) {
    if ((rc = roms_launch_k2d_mom_lds((const int *)&s, DUon, DVom, zeta_new, zwrk))) return rc;
  } else {
    hipLaunchKernelGGL(k2d_mom, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s, (const double *)DUon, (const double *)DVom, (const double *)zeta_new,
                       (const double *)zwrk);
    KERNEL_CHECK("k2d_mom");
  }
  if ((rc = bc_u2d(s.knew))) return rc;
  if ((rc = bc_v2d(s.knew))) return rc;
  const bool defer = in_loop && multi;
  if (defer) {
    hipLaunchKernelGGL(k2d_flux_own, grid2d(b.Iend - b.Istr + 1, b.JendR - b.JstrR + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s.knew, DUon, DVom);
    KERNEL_CHECK("k2d_flux_own");
  }
  halo_batch_begin();
  if (s.predictor) halo_exchange2d(GT_R, g_ctx.dev[FID_rzeta] + (long)(s.krhs - 1) * nij);
  halo_exchange2d(GT_R, g_ctx.dev[FID_zeta] + (long)(s.knew - 1) * nij);
  halo_exchange2d(GT_U, g_ctx.dev[FID_ubar] + (long)(s.knew - 1) * nij);
  halo_exchange2d(GT_V, g_ctx.dev[FID_vbar] + (long)(s.knew - 1) * nij);
  if (defer) {
    halo_exchange2d(GT_U, DUon);
    halo_exchange2d(GT_V, DVom);
  }
  if ((rc = halo_batch_end())) return rc;
  if (defer) { g_flux_ready = true; g_flux_lev = s.knew; }
  return 0;
}

}  // namespace

extern "C" int roms_hip_step2d(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_step2d");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  ScopedTimer tm("step2d");
  g_flux_ready = false;
  return step2d_impl(s, false);
}

// LOOP_2D of main3d.F:592-700
extern "C" int roms_hip_step2d_loop(roms_step_idx_t *s, int *indx1)
{
  int rc = roms_entry_check("roms_hip_step2d_loop");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  ScopedTimer tm("step2d_loop");
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
