// k_pre_step3d.hip -- predictor step, pre_step3d_tile
// (ROMS/Nonlinear/pre_step3d.F:123-1156).
//
//   k_pre_t   one thread per water column and tracer, one upward sweep:
//             horizontal + vertical advective fluxes of t(nstp), artificial
//             continuity, t(n+1/2) -> t(:,:,:,3,itrc)  (:342-915), and the
//             start of the corrector t(:,:,:,nnew,itrc) = Hz*t(nstp) + explicit
//             vertical flux divergence incl. KPP non-local and solar terms
//             (:917-1010, lmd_swfrac.F:6).  A thread reads t(nnew) only in its
//             own column before overwriting it, so the fusion is hazard-free.
//   k_pre_uv  one thread per column: start of u,v(nnew) with the AB3 terms
//             5/12 ru(nrhs) - 16/12 ru(3-nrhs) and surface/bottom stress
//             (:1012-1120).
// Algorithmic traffic (BENCHMARK, NT=2): per tracer read t(nstp), t(nnew), Akt,
// ghats; write t(3), t(nnew) = 48 B; shared Huon,Hvom,W,Hz,z_r,z_w = 48 B; momentum
// part reads u,v(nstp), ru,rv x2, Akv, writes u,v(nnew) = 72 B.
#include "roms_dev.h"
#include "advect.h"

int roms_entry_check(const char *name);

namespace {

// MASK: MASKING applications, first differences times umask / vmask of their face (pre_step3d.F:398, :463)
template <int HADV, int VADV, int NMAX, bool MASK>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_pre_t(const RomsDev *__restrict__ c, int nstp, int nnew, int first_step, int itrc0, int ntr)
{
  DEV_PROLOGUE(c)
  const TileTr tt = decode_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, ntr);
  if (!tt.valid) return;
  const int i = b.Istr + tt.bx * BLK_X + threadIdx.x;
  const int j = b.Jstr + tt.by * BLK_Y + threadIdx.y;
  const int itrc = itrc0 + tt.itr;
  if (i > b.Iend || j > b.Jend) return;
  const roms_params_t &p = c->p;
  const int ltrc = itrc < b.NAT ? itrc : b.NAT;
  const double dt = p.dt;
  const gcd_t ts = (gcd_t)(c->F.t + ((long)(nstp - 1) + 3L * (itrc - 1)) * n3r);
  const gd_t t3 = (gd_t)(c->F.t + (2L + 3L * (itrc - 1)) * n3r);
  const gd_t tn = (gd_t)(c->F.t + ((long)(nnew - 1) + 3L * (itrc - 1)) * n3r);
  const gcd_t Huon = (gcd_t)(c->F.Huon);
  const gcd_t Hvom = (gcd_t)(c->F.Hvom);
  const gcd_t Wv = (gcd_t)(c->F.W);
  const gcd_t Hz = (gcd_t)(c->F.Hz);
  const gcd_t z_r = (gcd_t)(c->F.z_r);
  const gcd_t z_w = (gcd_t)(c->F.z_w);
  const gcd_t Akt = (gcd_t)(c->F.Akt + (long)(ltrc - 1) * n3w);
  const long c0 = I2(i, j);
  const bool s_wall = b.south_edge && !b.NSperiodic && j == b.Jstr;
  const bool n_wall = b.north_edge && !b.NSperiodic && j == b.Jend;
  // physical western / eastern edges (pre_step3d.F:401-412: FX(Istr-1) = FX(Istr), FX(Iend+2) = FX(Iend+1))
  const bool w_wall = b.west_edge && !b.EWperiodic && i == b.Istr;
  const bool e_wall = b.east_edge && !b.EWperiodic && i == b.Iend;
  // time-stepping weights, pre_step3d.F:586-600
  // (upstream predictors -- MPDATA, HSIMT -- use Gamma = 1/2; the horizontal part takes the weight of the horizontal
  // scheme, :557-563, the vertical part that of the vertical scheme, :793-799: they differ for "HSIMT vertically with
  // another scheme horizontally")
  const double Gamma = (HADV == ADV_MPDATA) ? 0.5 : 1.0 / 6.0;
  const double GammaV = (VADV == ADV_MPDATA) ? 0.5 : 1.0 / 6.0;
  double cff, cff1, cff2;
  if (first_step) { cff = 0.5 * dt; cff1 = 1.0; cff2 = 0.0; }
  else { cff = (1.0 - Gamma) * dt; cff1 = 0.5 + Gamma; cff2 = 0.5 - Gamma; }
  const double cpp = cff * GF(pm)[c0] * GF(pn)[c0];
  const double cppv = (first_step ? 0.5 * dt : (1.0 - GammaV) * dt) * GF(pm)[c0] * GF(pn)[c0];
  // explicit vertical flux pieces
  const double cff3 = dt * (1.0 - p.lambda);
  const bool nonlocal = p.lmd_nonlocal && itrc <= b.NAT;
  const bool solar = p.solar_source && itrc == 1;
  const gcd_t ghats = (gcd_t)(c->F.ghats + (long)((itrc <= b.NAT ? itrc : 1) - 1) * n3w);
  const gcd_t AktN = (gcd_t)(c->F.Akt + (long)((itrc <= b.NAT ? itrc : 1) - 1) * n3w);   // Akt(..,itrc) of the non-local term
  const double srf = GF(srflx)[c0];
  // dt*srflx [*rmask_wet under WET_DRY, pre_step3d.F:873-878] -- the product the reference forms first, left to right
  const double dsrf = c->p.wet_dry ? (dt * srf) * GF(rmask_wet)[c0] : dt * srf;
  const double zwN = z_w[c0 + (long)N * nij];
  const double fac1 = -1.0 / p.swfrac_mu1, fac2 = -1.0 / p.swfrac_mu2, fac3 = p.swfrac_r1;

  // vertical schemes that need the whole column first
  double a4cf[(VADV == ADV_A4) ? NMAX + 2 : 1];
  double spl[(VADV == ADV_SPLINES) ? NMAX + 1 : 1];
  if constexpr (VADV == ADV_A4) {
    const double eps = 1.0E-16;
    double dprev = 0.0, tk = ts[c0];
#pragma unroll
    for (int k = 1; k <= NMAX; k++) {
      if (k <= N) {
        double dk;
        if (k < N) { const double tk1 = ts[c0 + (long)k * nij]; dk = tk1 - tk; tk = tk1; }
        else dk = dprev;
        if (k == 1) dprev = dk;
        const double cf = 2.0 * dk * dprev;
        a4cf[k] = (cf > eps) ? cf / (dk + dprev) : 0.0;
        dprev = dk;
      }
    }
  }
  if constexpr (VADV == ADV_SPLINES) {
    // pre_step3d.F:622-650 (note 1.5/0.5/3/2 instead of step3d_t's 2/1/2/1)
    double cfs[NMAX + 1];
    spl[0] = 1.5 * ts[c0];
    cfs[1] = 0.5;
#pragma unroll
    for (int k = 1; k < NMAX; k++) {
      if (k <= N - 1) {
        const double hk = Hz[c0 + (long)(k - 1) * nij], hk1 = Hz[c0 + (long)k * nij];
        const double cf = 1.0 / (2.0 * hk + hk1 * (2.0 - cfs[k]));
        cfs[k + 1] = cf * hk;
        spl[k] = cf * (3.0 * (hk * ts[c0 + (long)k * nij] + hk1 * ts[c0 + (long)(k - 1) * nij]) - hk1 * spl[k - 1]);
      }
    }
#pragma unroll
    for (int k = 1; k <= NMAX; k++)
      if (k == N) spl[k] = (3.0 * ts[c0 + (long)(N - 1) * nij] - spl[k - 1]) / (2.0 - cfs[k]);
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

  const bool src_cell = c->src.n > 0 && src_cell_any(c, c0, ni);      // LuvSrc: a face of this cell is a source face
  double tkm1 = 0.0, tk = ts[c0], tkp1 = (N >= 2) ? ts[c0 + nij] : 0.0, tkp2;
  double FCprev = 0.0;                                        // advective FC(k-1)
  double FDprev = dt * GF(btflx)[c0 + (long)(itrc - 1) * nij]; // diffusive FC(0)
  double w_km1 = Wv[c0];
  double zr_k = (cff3 != 0.0) ? z_r[c0] : 0.0;
  for (int k = 1; k <= N; k++) {
    const long ck = c0 + (long)(k - 1) * nij;
    tkp2 = (k + 2 <= N) ? ts[ck + 2 * nij] : 0.0;
    // ---- horizontal fluxes of t(nstp) ----
    const double xm1 = ts[ck - 1], xp1 = ts[ck + 1];
    const double xm2 = w_wall ? 0.0 : ts[ck - 2];
    const double xp2 = e_wall ? 0.0 : ts[ck + 2];
    const double ym1 = ts[ck - ni], yp1 = ts[ck + ni];
    const double ym2 = s_wall ? 0.0 : ts[ck - 2 * ni];
    const double yp2 = n_wall ? 0.0 : ts[ck + 2 * ni];
    const double hu0 = Huon[ck], hu1 = Huon[ck + 1];
    const double hv0 = Hvom[ck], hv1 = Hvom[ck + ni];
    double dxm1 = xm1 - xm2, dx0 = tk - xm1, dxp1 = xp1 - tk, dxp2 = xp2 - xp1;
    double dy0 = tk - ym1, dyp1 = yp1 - tk;
    double dym1 = ym1 - ym2, dyp2 = yp2 - yp1;
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
    double FXi = hflux<HADV>(hu0, xm1, tk, dxm1, dx0, dxp1);
    double FXip1 = hflux<HADV>(hu1, tk, xp1, dx0, dxp1, dxp2);
    double FEj = hflux<HADV>(hv0, ym1, tk, dym1, dy0, dyp1);
    double FEjp1 = hflux<HADV>(hv1, tk, yp1, dy0, dyp1, dyp2);
    if (src_cell)                                    // LuvSrc, pre_step3d.F:530-553
      src_cell_fluxes<true>(c, c0, ck, ni, k, itrc, nullptr, FXi, FXip1, FEj, FEjp1);
    const double hz = Hz[ck];
    const double tnn = tn[ck];
    double t3v = hz * (cff1 * tk + cff2 * tnn) - cpp * (FXip1 - FXi + FEjp1 - FEj);
    // ---- vertical advective flux through the top face ----
    const double w_k = Wv[ck + nij];
    double FCk;
    if (k == N) FCk = 0.0;
    else if constexpr (VADV == ADV_SPLINES) FCk = spl[k];
    else {
      double cfk = 0.0, cfk1 = 0.0;
      if constexpr (VADV == ADV_A4) { cfk = a4cf[k]; cfk1 = a4cf[k + 1]; }
      FCk = vflux<VADV>(k, N, w_k, tkm1, tk, tkp1, tkp2, cfk, cfk1);
    }
    // ---- artificial continuity, pre_step3d.F:893-915 ----
    const double DCk = 1.0 / (hz - cppv * (hu1 - hu0 + hv1 - hv0 + (w_k - w_km1)));
    t3v = DCk * (t3v - cppv * (FCk - FCprev));
    t3[ck] = t3v;
    // ---- start of the corrector: explicit vertical flux, pre_step3d.F:917-1010 ----
    double FDk;
    if (k == N) FDk = dt * GF(stflx)[c0 + (long)(itrc - 1) * nij];
    else {
      if (cff3 != 0.0) {
        const double zr_k1 = z_r[ck + nij];
        const double cz = 1.0 / (zr_k1 - zr_k);
        FDk = cff3 * cz * Akt[ck + nij] * (tkp1 - tk);
        zr_k = zr_k1;
      } else {
        // lambda = 1 (fully implicit vertical diffusion, every shipped application): the explicit flux is
        // cff3 * (...) = +-0; z_r, the division and (without the non-local term) Akt are not needed
        FDk = 0.0;
      }
      if (nonlocal) FDk = FDk - dt * AktN[ck + nij] * ghats[ck + nij];
      if (solar) {
        const double Z = zwN - z_w[ck + nij];
        const double swdk = exp(Z * fac1) * fac3 + exp(Z * fac2) * (1.0 - fac3);
        FDk = FDk + dsrf * swdk;
      }
    }
    tn[ck] = hz * tk + (FDk - FDprev);
    FDprev = FDk;
    FCprev = FCk;
    w_km1 = w_k;
    tkm1 = tk; tk = tkp1; tkp1 = tkp2;
  }
}

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_pre_uv(const RomsDev *__restrict__ c, int nstp, int nnew, int nrhs, int stage)
{
  // stage: 0 = first step (forward Euler), 1 = second step (AB2), 2 = AB3
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const roms_params_t &p = c->p;
  const double dt = p.dt;
  const double cff3 = dt * (1.0 - p.lambda);
  const gcd_t z_r = (gcd_t)(c->F.z_r);
  const gcd_t Hz = (gcd_t)(c->F.Hz);
  const gcd_t Akv = (gcd_t)(c->F.Akv);
  const gcd_t pm = (gcd_t)(c->F.pm);
  const gcd_t pn = (gcd_t)(c->F.pn);
  const long c0 = I2(i, j);
  const int indx = 3 - nrhs;
#pragma unroll
  for (int comp = 0; comp < 2; comp++) {
    if (comp == 0 && i < b.IstrU) continue;
    if (comp == 1 && j < b.JstrV) continue;
    const long off = comp == 0 ? 1 : ni;          // neighbour at i-1 or j-1
    const gcd_t vel = (gcd_t)((comp == 0 ? c->F.u : c->F.v) + (long)(nstp - 1) * n3r);
    const gd_t vnew = (gd_t)((comp == 0 ? c->F.u : c->F.v) + (long)(nnew - 1) * n3r);
    const gcd_t r1 = (gcd_t)((comp == 0 ? c->F.ru : c->F.rv) + (long)(nrhs - 1) * n3w);
    const gcd_t r2 = (gcd_t)((comp == 0 ? c->F.ru : c->F.rv) + (long)(indx - 1) * n3w);
    const double bstr = (comp == 0 ? c->F.bustr : c->F.bvstr)[c0];
    const double sstr = (comp == 0 ? c->F.sustr : c->F.svstr)[c0];
    const double cq = dt * 0.25;
    const double DC0 = cq * (pm[c0] + pm[c0 - off]) * (pn[c0] + pn[c0 - off]);
    double FCprev = dt * bstr;
    double vk = vel[c0];
    double zsum_k = 0.0;   // unused placeholder for clarity
    (void)zsum_k;
    for (int k = 1; k <= N; k++) {
      const long ck = c0 + (long)(k - 1) * nij;
      double FCk, vk1 = 0.0;
      if (k == N) FCk = dt * sstr;
      else {
        vk1 = vel[ck + nij];
        if (cff3 != 0.0) {
          const double cz = 1.0 / (z_r[ck + nij] + z_r[ck + nij - off] - z_r[ck] - z_r[ck - off]);
          FCk = cff3 * cz * (vk1 - vk) * (Akv[ck + nij] + Akv[ck + nij - off]);
        } else {
          // lambda = 1 (fully implicit vertical viscosity, every shipped application): the explicit flux is
          // cff3 * (...) = +-0; z_r and Akv need not be read (a zero of either sign gives the same sums)
          FCk = 0.0;
        }
      }
      const double hzs = Hz[ck] + Hz[ck - off];
      const double d = FCk - FCprev;
      double out;
      if (stage == 0) {
        const double a = vk * 0.5 * hzs;
        out = a + d;
      } else if (stage == 1) {
        const double a = vk * 0.5 * hzs;
        const double c3 = 0.5 * DC0;
        out = a - c3 * r2[ck + nij] + d;
      } else {
        const double a = vk * 0.5 * hzs;
        out = a + DC0 * ((5.0 / 12.0) * r1[ck + nij] - (16.0 / 12.0) * r2[ck + nij]) + d;
      }
      vnew[ck] = out;
      FCprev = FCk;
      vk = vk1;
    }
  }
}

template <int HADV, int VADV>
int launch_pre_t(const roms_step_idx_t *s, int itrc0, int ntr)
{
  const roms_bounds_t &b = g_ctx.b;
  const dim3 grid = grid_tile_tracer(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1, ntr);
  const int first = s->iic == s->ntfirst;
  if (b.N > ROMS_MAXN) return roms_fail("roms_hip_pre_step3d", "N > 64 not instantiated");
  constexpr bool maskable = HADV != ADV_MPDATA;        // the upstream predictor of MPDATA / HSIMT has no mask
  if constexpr (maskable) {
    if (g_ctx.p.masking) {
      if (b.N <= 16)
        hipLaunchKernelGGL((k_pre_t<HADV, VADV, 16, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, s->nnew, first, itrc0, ntr);
      else if (b.N <= 32)
        hipLaunchKernelGGL((k_pre_t<HADV, VADV, 32, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, s->nnew, first, itrc0, ntr);
      else
        hipLaunchKernelGGL((k_pre_t<HADV, VADV, ROMS_MAXN, true>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, s->nnew, first, itrc0, ntr);
      KERNEL_CHECK("k_pre_t");
      return 0;
    }
  }
  if (b.N <= 16)
    hipLaunchKernelGGL((k_pre_t<HADV, VADV, 16, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, s->nnew, first, itrc0, ntr);
  else if (b.N <= 32)
    hipLaunchKernelGGL((k_pre_t<HADV, VADV, 32, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, s->nnew, first, itrc0, ntr);
  else
    hipLaunchKernelGGL((k_pre_t<HADV, VADV, ROMS_MAXN, false>), grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, s->nnew, first, itrc0, ntr);
  KERNEL_CHECK("k_pre_t");
  return 0;
}

}  // namespace

// The tracer half of pre_step3d_tile (pre_step3d.F:342-915 and the tracer part of :917-1145): t(3), t(nnew),
// then t3dbc_tile(nout = 3) and the periodic wrap / mp_exchange4d of t(3).
static int pre_step3d_tracers(const roms_step_idx_t *s)
{
  const roms_bounds_t &b = g_ctx.b;
  const roms_params_t &p = g_ctx.p;
  int rc = 0;
  {
    int it = 1;
    while (it <= b.NT) {
      const int ha = p.Hadv[it - 1], va = p.Vadv[it - 1];
      int n = 1;
      while (it + n <= b.NT && p.Hadv[it + n - 1] == ha && p.Vadv[it + n - 1] == va) n++;
      switch (ha * 16 + va) {
      case ADV_U3 * 16 + ADV_C4:
      case ADV_U3 * 16 + ADV_SU3:  rc = launch_pre_t<ADV_U3, ADV_C4>(s, it, n); break;
      case ADV_A4 * 16 + ADV_A4:   rc = launch_pre_t<ADV_A4, ADV_A4>(s, it, n); break;
      case ADV_C4 * 16 + ADV_C4:
      case ADV_SU3 * 16 + ADV_SU3: rc = launch_pre_t<ADV_C4, ADV_C4>(s, it, n); break;
      case ADV_C2 * 16 + ADV_C2:   rc = launch_pre_t<ADV_C2, ADV_C2>(s, it, n); break;
      case ADV_U3 * 16 + ADV_SPLINES: rc = launch_pre_t<ADV_U3, ADV_SPLINES>(s, it, n); break;
      case ADV_C4 * 16 + ADV_SPLINES: rc = launch_pre_t<ADV_C4, ADV_SPLINES>(s, it, n); break;
      case ADV_A4 * 16 + ADV_SPLINES: rc = launch_pre_t<ADV_A4, ADV_SPLINES>(s, it, n); break;
      case ADV_MPDATA * 16 + ADV_MPDATA: rc = launch_pre_t<ADV_MPDATA, ADV_MPDATA>(s, it, n); break;
      // HSIMT: the predictor is the same first-order upstream step with Gamma = 1/2 (pre_step3d.F:364, :558, :730, :794)
      case ADV_HSIMT * 16 + ADV_HSIMT: rc = launch_pre_t<ADV_MPDATA, ADV_MPDATA>(s, it, n); break;
      // HSIMT vertically with another scheme horizontally (the one working one-sided pair of the reference): the
      // vertical predictor is the upstream step (:729-748, Gamma = 1/2 :793-799), the horizontal one the scheme's own
      case ADV_U3 * 16 + ADV_HSIMT: rc = launch_pre_t<ADV_U3, ADV_MPDATA>(s, it, n); break;
      case ADV_C4 * 16 + ADV_HSIMT: rc = launch_pre_t<ADV_C4, ADV_MPDATA>(s, it, n); break;
      case ADV_SU3 * 16 + ADV_HSIMT: rc = launch_pre_t<ADV_C4, ADV_MPDATA>(s, it, n); break;
      case ADV_A4 * 16 + ADV_HSIMT: rc = launch_pre_t<ADV_A4, ADV_MPDATA>(s, it, n); break;
      case ADV_C2 * 16 + ADV_HSIMT: rc = launch_pre_t<ADV_C2, ADV_MPDATA>(s, it, n); break;
      default:
        return roms_fail("roms_hip_pre_step3d", "advection scheme pair not implemented (MPDATA and HSIMT only as H+V pairs)");
      }
      if (rc) return rc;
      it += n;
    }
  }
  // t3dbc_tile(nout=3) + periodic wrap / mp_exchange4d, pre_step3d.F:1131-1145
  const long n3r = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * b.N;
  for (int it = 1; it <= b.NT; it++)
    if ((rc = bc_t3d(3, it, s->nstp))) return rc;
  halo_batch_begin();
  for (int it = 1; it <= b.NT; it++) halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_t] + (2L + 3L * (it - 1)) * n3r);
  return halo_batch_end();
}

// The momentum half: u, v(nnew) of the predictor (pre_step3d.F:1012-1120)
static int pre_step3d_uv(const roms_step_idx_t *s)
{
  const roms_bounds_t &b = g_ctx.b;
  const int stage = (s->iic == s->ntfirst) ? 0 : (s->iic == s->ntfirst + 1 ? 1 : 2);
  hipLaunchKernelGGL(k_pre_uv, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, s->nstp, s->nnew, s->nrhs, stage);
  KERNEL_CHECK("k_pre_uv");
  return 0;
}

extern "C" int roms_hip_pre_step3d(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_pre_step3d");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  if (g_ctx.b.N < 4) return roms_fail("roms_hip_pre_step3d", "N < 4");
  ScopedTimer tm("pre_step3d");
  // (the reference does the tracers first; the two halves share no output)
  if ((rc = pre_step3d_uv(s))) return rc;
  return pre_step3d_tracers(s);
}
