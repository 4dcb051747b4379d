// k_mix.hip -- harmonic and biharmonic lateral mixing of tracers on the 3-D step:
//   t3dmix2_geo_tile  ROMS/Nonlinear/t3dmix2_geo.h:90-424  (rotated to
//                     geopotentials, two-slab k-recursion)
//   t3dmix2_s_tile    ROMS/Nonlinear/t3dmix2_s.h:89-306    (along s-surfaces)
//   t3dmix4_geo_tile  ROMS/Nonlinear/t3dmix4_geo.h:104-784 \ the same operators applied twice (MODE 1: tracer ->
//   t3dmix4_s_tile    ROMS/Nonlinear/t3dmix4_s.h:100-480   /  LapT on a range one point wider, with the reference's
//                     rule for LapT outside a physical edge; MODE 2: LapT -> tracer), coefficient diff4
// (uv3dmix2_s_tile / uv3dmix4_s_tile are in k_uv3dmix2.hip)
//
// One thread per (i,j) column sweeping k upward.  The reference's two-slab
// buffers (level k and k+1 of dZdx,dTdx,dZde,dTde, W-levels k-1 and k of dTdz,
// FS) become register sliding windows, so the rotated operator needs one read
// of t(nrhs), z_r, Hz and one read-modify-write of t(nnew) per tracer.
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

__device__ __forceinline__ double dmin0(double a) { return a < 0.0 ? a : 0.0; }   // MIN(a,0)
__device__ __forceinline__ double dmax0(double a) { return a > 0.0 ? a : 0.0; }   // MAX(a,0)

// The tracer difference a - b of the operators' first application; STAB (TS_MIX_STABILITY, t3dmix2_s.h:212-218,
// t3dmix2_geo.h:236 / :268 / :301 and the like sites of t3dmix2_iso.h and of the first operator of t3dmix4_*.h): 3/4 of
// it plus 1/4 of the same difference a2 - b2 of t(nstp)
template <bool STAB>
__device__ __forceinline__ double tdiff(double a, double b, double a2, double b2)
{
  if constexpr (STAB) return 0.75 * (a - b) + 0.25 * (a2 - b2);
  else return a - b;
}

// The range and the edge rule of the first biharmonic operator (t3dmix4_s.h:262-275, :347-405): one point beyond the
// tile inside the grid; outside a physical edge LapT is zero where the tracer's condition is closed, a copy of the
// first inside value otherwise.  (The corner values the reference also sets are never read by the second operator.)
struct Lap4 {
  double *lap;                  // LapT of one tracer, module extents
  int i0, i1, j0, j1;           // Imin:Imax, Jmin:Jmax
  int closed[4];                // [LBS_WEST .. LBS_NORTH]
  int itrc;
  int nstp;                     // STAB kernels: the time level of the 1/4 part
};

template <int MODE>
__device__ __forceinline__ void lap4_store(const roms_bounds_t &b, const Lap4 &L, long ni, int i, int j, long a, double val)
{
  const gd_t lap = (gd_t)L.lap;
  lap[a] = val;
  if (!b.EWperiodic) {
    if (b.west_edge && i == b.Istr) lap[a - 1] = L.closed[LBS_WEST] ? 0.0 : val;
    if (b.east_edge && i == b.Iend) lap[a + 1] = L.closed[LBS_EAST] ? 0.0 : val;
  }
  if (!b.NSperiodic) {
    if (b.south_edge && j == b.Jstr) lap[a - ni] = L.closed[LBS_SOUTH] ? 0.0 : val;
    if (b.north_edge && j == b.Jend) lap[a + ni] = L.closed[LBS_NORTH] ? 0.0 : val;
  }
}

// MODE 0: t3dmix2_geo (all tracers of the launch).  MODE 1 / 2: first / second operator of t3dmix4_geo for tracer
// L.itrc -- the arithmetic of the three blocks of the reference is the same.
// ISO (MIX_ISO_TS): t3dmix2_iso.h:193-437 / t3dmix4_iso.h:262-805 -- the same sweep with the horizontal differences
// of the potential density in the place of those of z_r, the vertical tracer difference scaled by
// -1 / MAX(pden(k) - pden(k+1), eps), MIN and MAX exchanged, and the vertical flux times that scale times dz.
// STAB (MODE 0 / 1): TS_MIX_STABILITY -- t(nstp) is read beside t(nrhs) and every tracer difference is tdiff<true>.
template <int MODE, bool ISO, bool STAB = false>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_t3dmix_geo(const RomsDev *__restrict__ c, int nrhs, int nnew, Lap4 L)
{
  DEV_PROLOGUE(c)
  const int ilo = MODE == 1 ? L.i0 : b.Istr, ihi = MODE == 1 ? L.i1 : b.Iend;
  const int jlo = MODE == 1 ? L.j0 : b.Jstr, jhi = MODE == 1 ? L.j1 : b.Jend;
  const TileTr tt = decode_tile_tracer(ihi - ilo + 1, jhi - jlo + 1, MODE == 0 ? b.NT : 1);
  if (!tt.valid) return;
  const int i = ilo + tt.bx * BLK_X + threadIdx.x;
  const int j = jlo + tt.by * BLK_Y + threadIdx.y;
  const int itrc = MODE == 0 ? 1 + tt.itr : L.itrc;
  if (i > ihi || j > jhi) return;
  const double dt = c->p.dt;
  const double *__restrict__ T = MODE == 2 ? L.lap : c->F.t + ((long)(nrhs - 1) + 3L * (itrc - 1)) * n3r;
  static_assert(!(STAB && MODE == 2), "the second biharmonic operator acts on LapT alone");
  const double *__restrict__ S = STAB ? c->F.t + ((long)(L.nstp - 1) + 3L * (itrc - 1)) * n3r : T;
  double *__restrict__ tn = c->F.t + ((long)(nnew - 1) + 3L * (itrc - 1)) * n3r;
  const double *__restrict__ z_r = ISO ? c->F.pden : c->F.z_r;      // the field whose horizontal differences give the slopes
  const double *__restrict__ zz = c->F.z_r;                          // ISO: depths of the own column
  const double *__restrict__ Hz = c->F.Hz;
  const double *__restrict__ pm = c->F.pm;
  const double *__restrict__ pn = c->F.pn;
  const double *__restrict__ d2 = (MODE == 0 ? c->F.diff2 : c->F.diff4) + (long)(itrc - 1) * nij;
  // the vertical scale of a column between levels k (a) and k+1 (b): 1/dz, or -1/MAX(drho, eps) (t3dmix2_iso.h:299-301)
  // (ISO with TS_MIX_MIN_STRAT, t3dmix2_iso.h:313-316: the bound of the density difference is strat_min = 0.1 times the
  // distance dz of the two levels in that column instead of eps = 0.5)
  const bool minstrat = ISO && c->p.ts_mix_min_strat != 0;
  auto vscale = [minstrat](double a, double b, double dz) {
    if constexpr (ISO) return -1.0 / fmax(a - b, minstrat ? 0.1 * dz : 0.5);
    else return 1.0 / (b - a);
  };
  auto f1 = [](double a) { if constexpr (ISO) return dmax0(a); else return dmin0(a); };
  auto f2 = [](double a) { if constexpr (ISO) return dmin0(a); else return dmax0(a); };
  const long c0 = I2(i, j);
  // face metrics: xi faces i and i+1, eta faces j and j+1
  double mx0 = 0.5 * (pm[c0] + pm[c0 - 1]), mx1 = 0.5 * (pm[c0 + 1] + pm[c0]);
  double my0 = 0.5 * (pn[c0] + pn[c0 - ni]), my1 = 0.5 * (pn[c0 + ni] + pn[c0]);
  if (c->p.masking) {                                     // MASKING, t3dmix2_geo.h:228, :260
    mx0 = mx0 * umaskw(c, c0); mx1 = mx1 * umaskw(c, c0 + 1);          // (+ WET_DRY, :231, :263)
    my0 = my0 * vmaskw(c, c0); my1 = my1 * vmaskw(c, c0 + ni);
  }
  const double cfx0 = 0.25 * (d2[c0] + d2[c0 - 1]) * c->F.on_u[c0];
  const double cfx1 = 0.25 * (d2[c0 + 1] + d2[c0]) * c->F.on_u[c0 + 1];
  const double cfe0 = 0.25 * (d2[c0] + d2[c0 - ni]) * c->F.om_v[c0];
  const double cfe1 = 0.25 * (d2[c0 + ni] + d2[c0]) * c->F.om_v[c0 + ni];
  const double cfs = 0.5 * d2[c0];
  const double cdt = dt * pm[c0] * pn[c0];
  // level-k ("k1") and level-(k+1) ("k2") slabs at faces i, i+1, j, j+1
  double zx0_a, zx1_a, tx0_a, tx1_a, ze0_a, ze1_a, te0_a, te1_a;     // level k
  double zx0_b, zx1_b, tx0_b, tx1_b, ze0_b, ze1_b, te0_b, te1_b;     // level k+1
  // dTdz at W-level k-1 ("k1") and k ("k2") in columns (i-1,i,i+1,j-1,j+1)
  double dzm_a = 0.0, dz0_a = 0.0, dzp_a = 0.0, dzs_a = 0.0, dzn_a = 0.0;
  double dzm_b, dz0_b, dzp_b, dzs_b, dzn_b;
  double FS_a = 0.0, FS_b;
  double fsf_b = 0.0;                      // ISO: FS(i,j,k2) = scale * dz before the flux is formed (:303)
  // Software pipeline (cf. k_step3d_t_pipe): the 16 loads an iteration consumes -- T and z_r of level
  // k+1 at the five columns, Hz of level k at the five columns, t(nnew) of level k -- are issued one
  // iteration ahead, so their latency overlaps the arithmetic of the previous level.
  struct LvIn { double Tm1, T01, Tp1, Ts1, Tn1, Zm1, Z01, Zp1, Zs1, Zn1, hz0, hzm, hzp, hzs, hzn, tn, zz1;
                double Sm1, S01, Sp1, Ss1, Sn1;
                double zzm1, zzp1, zzs1, zzn1; };          // TS_MIX_MIN_STRAT: z_r of the four neighbour columns
  const gcd_t gT = (gcd_t)T, gS = (gcd_t)S, gZ = (gcd_t)z_r, gHz = (gcd_t)Hz;
  const gd_t gtn = (gd_t)tn;
  auto load_level = [&](int k) {
    LvIn L;
    const long ck = c0 + (long)(k - 1) * nij;
    const long cu = (k < N) ? ck + nij : ck;               // level k+1 (clamped at the top; unused there)
    L.Tm1 = gT[cu - 1]; L.T01 = gT[cu]; L.Tp1 = gT[cu + 1]; L.Ts1 = gT[cu - ni]; L.Tn1 = gT[cu + ni];
    if constexpr (STAB) { L.Sm1 = gS[cu - 1]; L.S01 = gS[cu]; L.Sp1 = gS[cu + 1]; L.Ss1 = gS[cu - ni]; L.Sn1 = gS[cu + ni]; }
    else L.Sm1 = L.S01 = L.Sp1 = L.Ss1 = L.Sn1 = 0.0;
    L.Zm1 = gZ[cu - 1]; L.Z01 = gZ[cu]; L.Zp1 = gZ[cu + 1]; L.Zs1 = gZ[cu - ni]; L.Zn1 = gZ[cu + ni];
    L.hz0 = gHz[ck]; L.hzm = gHz[ck - 1]; L.hzp = gHz[ck + 1]; L.hzs = gHz[ck - ni]; L.hzn = gHz[ck + ni];
    L.tn = gtn[ck];
    if constexpr (ISO) L.zz1 = ((gcd_t)zz)[cu]; else L.zz1 = 0.0;
    L.zzm1 = L.zzp1 = L.zzs1 = L.zzn1 = 0.0;
    if (minstrat) {
      const gcd_t gz = (gcd_t)zz;
      L.zzm1 = gz[cu - 1]; L.zzp1 = gz[cu + 1]; L.zzs1 = gz[cu - ni]; L.zzn1 = gz[cu + ni];
    }
    return L;
  };
  // values of level k carried for the vertical differences
  double Tm = T[c0 - 1], T0 = T[c0], Tp = T[c0 + 1], Ts = T[c0 - ni], Tn = T[c0 + ni];
  double Sm = 0.0, S0 = 0.0, Sp = 0.0, Ss = 0.0, Sn = 0.0;
  if constexpr (STAB) { Sm = S[c0 - 1]; S0 = S[c0]; Sp = S[c0 + 1]; Ss = S[c0 - ni]; Sn = S[c0 + ni]; }
  double Zm = z_r[c0 - 1], Z0 = z_r[c0], Zp = z_r[c0 + 1], Zs = z_r[c0 - ni], Zn = z_r[c0 + ni];
  double zzc = ISO ? zz[c0] : 0.0;
  double zzm = 0.0, zzp = 0.0, zzs = 0.0, zzn = 0.0;
  if (minstrat) { zzm = zz[c0 - 1]; zzp = zz[c0 + 1]; zzs = zz[c0 - ni]; zzn = zz[c0 + ni]; }
  // level 1 slabs (iteration k=0 of the reference)
  zx0_b = mx0 * (Z0 - Zm); tx0_b = mx0 * tdiff<STAB>(T0, Tm, S0, Sm);
  zx1_b = mx1 * (Zp - Z0); tx1_b = mx1 * tdiff<STAB>(Tp, T0, Sp, S0);
  ze0_b = my0 * (Z0 - Zs); te0_b = my0 * tdiff<STAB>(T0, Ts, S0, Ss);
  ze1_b = my1 * (Zn - Z0); te1_b = my1 * tdiff<STAB>(Tn, T0, Sn, S0);
  LvIn cur = load_level(1);
  for (int k = 1; k <= N; k++) {
    const long ck = c0 + (long)(k - 1) * nij;
    LvIn nxt = cur;
    if (k < N) nxt = load_level(k + 1);
    zx0_a = zx0_b; zx1_a = zx1_b; tx0_a = tx0_b; tx1_a = tx1_b;
    ze0_a = ze0_b; ze1_a = ze1_b; te0_a = te0_b; te1_a = te1_b;
    if (k < N) {
      const double Tm1 = cur.Tm1, T01 = cur.T01, Tp1 = cur.Tp1, Ts1 = cur.Ts1, Tn1 = cur.Tn1;
      const double Zm1 = cur.Zm1, Z01 = cur.Z01, Zp1 = cur.Zp1, Zs1 = cur.Zs1, Zn1 = cur.Zn1;
      const double Sm1 = cur.Sm1, S01 = cur.S01, Sp1 = cur.Sp1, Ss1 = cur.Ss1, Sn1 = cur.Sn1;
      zx0_b = mx0 * (Z01 - Zm1); tx0_b = mx0 * tdiff<STAB>(T01, Tm1, S01, Sm1);
      zx1_b = mx1 * (Zp1 - Z01); tx1_b = mx1 * tdiff<STAB>(Tp1, T01, Sp1, S01);
      ze0_b = my0 * (Z01 - Zs1); te0_b = my0 * tdiff<STAB>(T01, Ts1, S01, Ss1);
      ze1_b = my1 * (Zn1 - Z01); te1_b = my1 * tdiff<STAB>(Tn1, T01, Sn1, S01);
      { const double q = vscale(Zm, Zm1, cur.zzm1 - zzm); dzm_b = q * tdiff<STAB>(Tm1, Tm, Sm1, Sm); }
      { const double q = vscale(Z0, Z01, cur.zz1 - zzc); dz0_b = q * tdiff<STAB>(T01, T0, S01, S0);
        if constexpr (ISO) { fsf_b = q * (cur.zz1 - zzc); zzc = cur.zz1; } }
      { const double q = vscale(Zp, Zp1, cur.zzp1 - zzp); dzp_b = q * tdiff<STAB>(Tp1, Tp, Sp1, Sp); }
      { const double q = vscale(Zs, Zs1, cur.zzs1 - zzs); dzs_b = q * tdiff<STAB>(Ts1, Ts, Ss1, Ss); }
      { const double q = vscale(Zn, Zn1, cur.zzn1 - zzn); dzn_b = q * tdiff<STAB>(Tn1, Tn, Sn1, Sn); }
      zzm = cur.zzm1; zzp = cur.zzp1; zzs = cur.zzs1; zzn = cur.zzn1;
      Tm = Tm1; T0 = T01; Tp = Tp1; Ts = Ts1; Tn = Tn1;
      Sm = Sm1; S0 = S01; Sp = Sp1; Ss = Ss1; Sn = Sn1;
      Zm = Zm1; Z0 = Z01; Zp = Zp1; Zs = Zs1; Zn = Zn1;
    } else {
      dzm_b = dz0_b = dzp_b = dzs_b = dzn_b = 0.0;
    }
    const double hz0 = cur.hz0, hzm = cur.hzm, hzp = cur.hzp, hzs = cur.hzs, hzn = cur.hzn;
    // FX(i), FX(i+1), FE(j), FE(j+1): t3dmix2_geo.h:268-310
    const double FX0 = cfx0 * (hz0 + hzm) *
        (tx0_a - 0.5 * (f1(zx0_a) * (dzm_a + dz0_b) + f2(zx0_a) * (dzm_b + dz0_a)));
    const double FX1 = cfx1 * (hzp + hz0) *
        (tx1_a - 0.5 * (f1(zx1_a) * (dz0_a + dzp_b) + f2(zx1_a) * (dz0_b + dzp_a)));
    const double FE0 = cfe0 * (hz0 + hzs) *
        (te0_a - 0.5 * (f1(ze0_a) * (dzs_a + dz0_b) + f2(ze0_a) * (dzs_b + dz0_a)));
    const double FE1 = cfe1 * (hzn + hz0) *
        (te1_a - 0.5 * (f1(ze1_a) * (dz0_a + dzn_b) + f2(ze1_a) * (dz0_b + dzn_a)));
    if (k < N) {
      double c1 = f1(zx0_a), c2 = f1(zx1_b), c3 = f2(zx0_b), c4 = f2(zx1_a);
      if constexpr (!ISO) {
        FS_b = cfs * (c1 * (c1 * dz0_b - tx0_a) + c2 * (c2 * dz0_b - tx1_b) +
                      c3 * (c3 * dz0_b - tx0_b) + c4 * (c4 * dz0_b - tx1_a));
        c1 = f1(ze0_a); c2 = f1(ze1_b); c3 = f2(ze0_b); c4 = f2(ze1_a);
        FS_b = FS_b + cfs * (c1 * (c1 * dz0_b - te0_a) + c2 * (c2 * dz0_b - te1_b) +
                             c3 * (c3 * dz0_b - te0_b) + c4 * (c4 * dz0_b - te1_a));
      } else if constexpr (MODE == 0) {          // t3dmix2_iso.h:395-417: one running sum, then 0.5 * cff * diff2 * FS
        double cff = c1 * (c1 * dz0_b - tx0_a) + c2 * (c2 * dz0_b - tx1_b) +
                     c3 * (c3 * dz0_b - tx0_b) + c4 * (c4 * dz0_b - tx1_a);
        c1 = f1(ze0_a); c2 = f1(ze1_b); c3 = f2(ze0_b); c4 = f2(ze1_a);
        cff = cff + c1 * (c1 * dz0_b - te0_a) + c2 * (c2 * dz0_b - te1_b) +
              c3 * (c3 * dz0_b - te0_b) + c4 * (c4 * dz0_b - te1_a);
        FS_b = 0.5 * cff * d2[c0] * fsf_b;
      } else {                                   // t3dmix4_iso.h:443-480: each direction times 0.5 * diff4
        double cff = cfs * (c1 * (c1 * dz0_b - tx0_a) + c2 * (c2 * dz0_b - tx1_b) +
                            c3 * (c3 * dz0_b - tx0_b) + c4 * (c4 * dz0_b - tx1_a));
        c1 = f1(ze0_a); c2 = f1(ze1_b); c3 = f2(ze0_b); c4 = f2(ze1_a);
        cff = cff + cfs * (c1 * (c1 * dz0_b - te0_a) + c2 * (c2 * dz0_b - te1_b) +
                           c3 * (c3 * dz0_b - te0_b) + c4 * (c4 * dz0_b - te1_a));
        FS_b = cff * fsf_b;
      }
    } else FS_b = 0.0;
    if constexpr (MODE == 1) {                 // t3dmix4_geo.h:445-455
      const double cff = pm[c0] * pn[c0];
      const double cff1 = 1.0 / hz0;
      lap4_store<MODE>(b, L, ni, i, j, ck, cff1 * (cff * (FX1 - FX0 + FE1 - FE0) + (FS_b - FS_a)));
    } else {
      const double cff1 = cdt * (FX1 - FX0);
      const double cff2 = cdt * (FE1 - FE0);
      const double cff3 = dt * (FS_b - FS_a);
      const double cff4 = cff1 + cff2 + cff3;
      if constexpr (MODE == 0) gtn[ck] = cur.tn + cff4;
      else gtn[ck] = cur.tn - cff4;            // t3dmix4_geo.h:767
    }
    cur = nxt;
    dzm_a = dzm_b; dz0_a = dz0_b; dzp_a = dzp_b; dzs_a = dzs_b; dzn_a = dzn_b;
    FS_a = FS_b;
  }
}

template <bool STAB>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_t3dmix2_s(const RomsDev *__restrict__ c, int nrhs, int nnew, int nstp)
{
  DEV_PROLOGUE(c)
  const TileTr tt = decode_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, b.NT);
  if (!tt.valid) return;
  const int i = b.Istr + tt.bx * BLK_X + threadIdx.x;
  const int j = b.Jstr + tt.by * BLK_Y + threadIdx.y;
  const int itrc = 1 + tt.itr;
  if (i > b.Iend || j > b.Jend) return;
  const double *__restrict__ T = c->F.t + ((long)(nrhs - 1) + 3L * (itrc - 1)) * n3r;
  const double *__restrict__ S = STAB ? c->F.t + ((long)(nstp - 1) + 3L * (itrc - 1)) * n3r : T;   // TS_MIX_STABILITY
  double *__restrict__ tn = c->F.t + ((long)(nnew - 1) + 3L * (itrc - 1)) * n3r;
  const double *__restrict__ Hz = c->F.Hz;
  const double *__restrict__ d2 = c->F.diff2 + (long)(itrc - 1) * nij;
  const long c0 = I2(i, j);
  const double cfx0 = 0.25 * (d2[c0] + d2[c0 - 1]) * c->F.pmon_u[c0];
  const double cfx1 = 0.25 * (d2[c0 + 1] + d2[c0]) * c->F.pmon_u[c0 + 1];
  const double cfe0 = 0.25 * (d2[c0] + d2[c0 - ni]) * c->F.pnom_v[c0];
  const double cfe1 = 0.25 * (d2[c0 + ni] + d2[c0]) * c->F.pnom_v[c0 + ni];
  const double cdt = c->p.dt * c->F.pm[c0] * c->F.pn[c0];
  const bool masking = c->p.masking != 0;                 // MASKING, t3dmix2_s.h:235, :275
  // (WET_DRY: times the wet/dry mask of the face, the block after each MASKING block of t3dmix2_s.h / t3dmix4_s.h)
  const double um0 = masking ? umaskw(c, c0) : 1.0, um1 = masking ? umaskw(c, c0 + 1) : 1.0;
  const double vm0 = masking ? vmaskw(c, c0) : 1.0, vm1 = masking ? vmaskw(c, c0 + ni) : 1.0;
  for (int k = 1; k <= N; k++) {
    const long ck = c0 + (long)(k - 1) * nij;
    const double t0 = T[ck], h0 = Hz[ck];
    double s0 = 0.0, sm = 0.0, sp = 0.0, ss = 0.0, sn = 0.0;
    if constexpr (STAB) { s0 = S[ck]; sm = S[ck - 1]; sp = S[ck + 1]; ss = S[ck - ni]; sn = S[ck + ni]; }
    double FX0 = cfx0 * (h0 + Hz[ck - 1]) * tdiff<STAB>(t0, T[ck - 1], s0, sm);
    double FX1 = cfx1 * (Hz[ck + 1] + h0) * tdiff<STAB>(T[ck + 1], t0, sp, s0);
    double FE0 = cfe0 * (h0 + Hz[ck - ni]) * tdiff<STAB>(t0, T[ck - ni], s0, ss);
    double FE1 = cfe1 * (Hz[ck + ni] + h0) * tdiff<STAB>(T[ck + ni], t0, sn, s0);
    if (masking) { FX0 = FX0 * um0; FX1 = FX1 * um1; FE0 = FE0 * vm0; FE1 = FE1 * vm1; }
    const double cff1 = cdt * (FX1 - FX0);
    const double cff2 = cdt * (FE1 - FE0);
    tn[ck] = tn[ck] + (cff1 + cff2);
  }
}

// t3dmix4_s_tile, first operator (MODE 1, :281-345) and second operator with the time step (MODE 2, :407-475)
template <int MODE, bool STAB = false>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_t3dmix4_s(const RomsDev *__restrict__ c, int nrhs, int nnew, Lap4 L)
{
  DEV_PROLOGUE(c)
  const int ilo = MODE == 1 ? L.i0 : b.Istr, ihi = MODE == 1 ? L.i1 : b.Iend;
  const int jlo = MODE == 1 ? L.j0 : b.Jstr, jhi = MODE == 1 ? L.j1 : b.Jend;
  const TileTr tt = decode_tile_tracer(ihi - ilo + 1, jhi - jlo + 1, 1);
  if (!tt.valid) return;
  const int i = ilo + tt.bx * BLK_X + threadIdx.x;
  const int j = jlo + tt.by * BLK_Y + threadIdx.y;
  const int itrc = L.itrc;
  if (i > ihi || j > jhi) return;
  const double *__restrict__ T = MODE == 2 ? L.lap : c->F.t + ((long)(nrhs - 1) + 3L * (itrc - 1)) * n3r;
  static_assert(!(STAB && MODE == 2), "the second biharmonic operator acts on LapT alone");
  const double *__restrict__ S = STAB ? c->F.t + ((long)(L.nstp - 1) + 3L * (itrc - 1)) * n3r : T;   // TS_MIX_STABILITY
  double *__restrict__ tn = c->F.t + ((long)(nnew - 1) + 3L * (itrc - 1)) * n3r;
  const double *__restrict__ Hz = c->F.Hz;
  const double *__restrict__ d4 = c->F.diff4 + (long)(itrc - 1) * nij;
  const long c0 = I2(i, j);
  double cfx0 = 0.25 * (d4[c0] + d4[c0 - 1]) * c->F.pmon_u[c0];
  double cfx1 = 0.25 * (d4[c0 + 1] + d4[c0]) * c->F.pmon_u[c0 + 1];
  double cfe0 = 0.25 * (d4[c0] + d4[c0 - ni]) * c->F.pnom_v[c0];
  double cfe1 = 0.25 * (d4[c0 + ni] + d4[c0]) * c->F.pnom_v[c0 + ni];
  const bool masking = c->p.masking != 0;
  // (WET_DRY: times the wet/dry mask of the face, the block after each MASKING block of t3dmix2_s.h / t3dmix4_s.h)
  const double um0 = masking ? umaskw(c, c0) : 1.0, um1 = masking ? umaskw(c, c0 + 1) : 1.0;
  const double vm0 = masking ? vmaskw(c, c0) : 1.0, vm1 = masking ? vmaskw(c, c0 + ni) : 1.0;
  if (MODE == 1 && masking) {                            // the first operator masks the coefficient, :296, :326
    cfx0 = cfx0 * um0; cfx1 = cfx1 * um1; cfe0 = cfe0 * vm0; cfe1 = cfe1 * vm1;
  }
  const double pmn = c->F.pm[c0] * c->F.pn[c0];
  const double cdt = c->p.dt * c->F.pm[c0] * c->F.pn[c0];
  for (int k = 1; k <= N; k++) {
    const long ck = c0 + (long)(k - 1) * nij;
    const double t0 = T[ck], h0 = Hz[ck];
    double s0 = 0.0, sm = 0.0, sp = 0.0, ss = 0.0, sn = 0.0;
    if constexpr (STAB) { s0 = S[ck]; sm = S[ck - 1]; sp = S[ck + 1]; ss = S[ck - ni]; sn = S[ck + ni]; }
    double FX0 = cfx0 * (h0 + Hz[ck - 1]) * tdiff<STAB>(t0, T[ck - 1], s0, sm);
    double FX1 = cfx1 * (Hz[ck + 1] + h0) * tdiff<STAB>(T[ck + 1], t0, sp, s0);
    double FE0 = cfe0 * (h0 + Hz[ck - ni]) * tdiff<STAB>(t0, T[ck - ni], s0, ss);
    double FE1 = cfe1 * (Hz[ck + ni] + h0) * tdiff<STAB>(T[ck + ni], t0, sn, s0);
    if constexpr (MODE == 1) {
      const double cff = 1.0 / h0;
      lap4_store<MODE>(b, L, ni, i, j, ck, pmn * cff * (FX1 - FX0 + FE1 - FE0));
    } else {
      if (masking) { FX0 = FX0 * um0; FX1 = FX1 * um1; FE0 = FE0 * vm0; FE1 = FE1 * vm1; }   // the second the flux, :425, :446
      const double cff1 = cdt * (FX1 - FX0);
      const double cff2 = cdt * (FE1 - FE0);
      tn[ck] = tn[ck] - (cff1 + cff2);
    }
  }
}

}  // namespace

static int t3dmix2_launch(const roms_step_idx_t *s)
{
  const roms_bounds_t &b = g_ctx.b;
  const dim3 grid = grid_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, b.NT);
  const bool stab = g_ctx.p.ts_mix_stability != 0;       // TS_MIX_STABILITY
  Lap4 L{};
  L.nstp = s->nstp;
  if (g_ctx.p.mix_iso_ts) {
    if (stab) hipLaunchKernelGGL((k_t3dmix_geo<0, true, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
    else hipLaunchKernelGGL((k_t3dmix_geo<0, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
  } else if (g_ctx.p.mix_geo_ts) {
    if (stab) hipLaunchKernelGGL((k_t3dmix_geo<0, false, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
    else hipLaunchKernelGGL((k_t3dmix_geo<0, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
  } else if (g_ctx.p.mix_s_ts) {
    if (stab) hipLaunchKernelGGL(k_t3dmix2_s<true>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, s->nstp);
    else hipLaunchKernelGGL(k_t3dmix2_s<false>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, s->nstp);
  } else
    return roms_fail("roms_hip_t3dmix2", "no tracer mixing option (MIX_ISO_TS / MIX_GEO_TS / MIX_S_TS) selected");
  KERNEL_CHECK("k_t3dmix2");
  return 0;
}

extern "C" int roms_hip_t3dmix2(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_t3dmix2");
  if (rc) return rc;
  ScopedTimer tm("t3dmix2");
  return t3dmix2_launch(s);
}

extern "C" int roms_hip_t3dmix4(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_t3dmix4");
  if (rc) return rc;
  const roms_bounds_t &b = g_ctx.b;
  const roms_params_t &p = g_ctx.p;
  if (!p.ts_dif4) return roms_fail("roms_hip_t3dmix4", "TS_DIF4 is not set (roms_params_t.ts_dif4)");
  if (!p.mix_iso_ts && !p.mix_geo_ts && !p.mix_s_ts)
    return roms_fail("roms_hip_t3dmix4", "no tracer mixing option (MIX_ISO_TS / MIX_GEO_TS / MIX_S_TS) selected");
  ScopedTimer tm("t3dmix4");
  const bool stab = p.ts_mix_stability != 0;             // TS_MIX_STABILITY: in the first operator only
  Lap4 L;
  L.lap = g_ctx.hostc.ws3[1];
  L.nstp = s->nstp;
  if (b.EWperiodic) { L.i0 = b.Istr - 1; L.i1 = b.Iend + 1; }
  else { L.i0 = b.Istr - 1 > 1 ? b.Istr - 1 : 1; L.i1 = b.Iend + 1 < b.Lm ? b.Iend + 1 : b.Lm; }
  if (b.NSperiodic) { L.j0 = b.Jstr - 1; L.j1 = b.Jend + 1; }
  else { L.j0 = b.Jstr - 1 > 1 ? b.Jstr - 1 : 1; L.j1 = b.Jend + 1 < b.Mm ? b.Jend + 1 : b.Mm; }
  for (int sd = 0; sd < 4; sd++) L.closed[sd] = lbc_code(p, sd, LBV_T) == LBC_CLOSED;
  const dim3 g1 = grid_tile_tracer(L.i1 - L.i0 + 1, L.j1 - L.j0 + 1, 1);
  const dim3 g2 = grid_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, 1);
  for (int itrc = 1; itrc <= b.NT; itrc++) {
    L.itrc = itrc;
    if (p.mix_iso_ts) {
      if (stab) hipLaunchKernelGGL((k_t3dmix_geo<1, true, true>), g1, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
      else hipLaunchKernelGGL((k_t3dmix_geo<1, true>), g1, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
      hipLaunchKernelGGL((k_t3dmix_geo<2, true>), g2, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
    } else if (p.mix_geo_ts) {
      if (stab) hipLaunchKernelGGL((k_t3dmix_geo<1, false, true>), g1, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
      else hipLaunchKernelGGL((k_t3dmix_geo<1, false>), g1, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
      hipLaunchKernelGGL((k_t3dmix_geo<2, false>), g2, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
    } else {
      if (stab) hipLaunchKernelGGL((k_t3dmix4_s<1, true>), g1, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
      else hipLaunchKernelGGL(k_t3dmix4_s<1>, g1, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
      hipLaunchKernelGGL(k_t3dmix4_s<2>, g2, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs, s->nnew, L);
    }
    KERNEL_CHECK("k_t3dmix4");
  }
  return 0;
}
