// k_physics.hip -- per-step physics between the hot kernels, kept on the
// device so that the state never crosses PCIe inside a step:
//   roms_hip_set_vbc    set_vbc_tile   ROMS/Nonlinear/set_vbc.F:104   (UV_QDRAG / UV_LDRAG, SALINITY)
//   roms_hip_bulk_flux  bulk_flux_tile ROMS/Nonlinear/bulk_flux.F:146 (COARE 3.0 with the Berliand
//                       longwave formula; no COOL_SKIN / EMINUSP / WIND_MINUS_CURRENT / masking)
// Both are point-local (bulk_flux: three fixed iterations per point); one thread per (i,j).
// bulk_flux uses device log/exp/pow/atan: results agree with the host libraries to a few ulp, not
// bit for bit (the tests state the tolerance).
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_set_vbc(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const roms_params_t &p = c->p;
  const int i = b.IstrR + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.JstrR + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendR || j > b.JendR) return;
  const long a = I2(i, j);
  const int NT = b.NT;
  const double *__restrict__ u = c->F.u + (long)(nrhs - 1) * n3r;
  const double *__restrict__ v = c->F.v + (long)(nrhs - 1) * n3r;
  // kinematic surface / bottom tracer fluxes, set_vbc.F:262-312
  c->F.stflx[a] = c->F.stflux[a];
  c->F.btflx[a] = c->F.btflux[a];
  if (p.salinity && NT >= 2) {
    const double *__restrict__ S = c->F.t + ((long)(nrhs - 1) + 3L * 1) * n3r;      // t(:,:,:,nrhs,isalt)
    const double EmP = c->F.stflux[a + nij];
    c->F.stflx[a + nij] = EmP * S[a + (long)(N - 1) * nij];
    c->F.btflx[a + nij] = c->F.btflx[a + nij] * S[a];
  }
  // bottom stress, :380-470 (k = 1 is the first plane of u, v)
  const bool on_u = i >= b.IstrU && i <= b.Iend && j >= b.Jstr && j <= b.Jend;
  const bool on_v = i >= b.Istr && i <= b.Iend && j >= b.JstrV && j <= b.Jend;
  if (p.uv_drag == 2) {
    const double *__restrict__ r2 = c->F.rdrag2;
    if (on_u) {
      const double cff1 = 0.25 * (v[a] + v[a + ni] + v[a - 1] + v[a - 1 + ni]);
      const double cff2 = sqrt(u[a] * u[a] + cff1 * cff1);
      const double bu = 0.5 * (r2[a - 1] + r2[a]) * u[a] * cff2;
      c->F.bustr[a] = bu;
      if (b.south_edge && !b.NSperiodic && j == b.Jstr) c->F.bustr[a - ni] = p.gamma2 * bu;   // bc_u2d_tile
      if (b.north_edge && !b.NSperiodic && j == b.Jend) c->F.bustr[a + ni] = p.gamma2 * bu;
    }
    if (on_v) {
      const double cff1 = 0.25 * (u[a] + u[a + 1] + u[a - ni] + u[a + 1 - ni]);
      const double cff2 = sqrt(cff1 * cff1 + v[a] * v[a]);
      c->F.bvstr[a] = 0.5 * (r2[a - ni] + r2[a]) * v[a] * cff2;
    }
  } else {
    const double *__restrict__ r1 = c->F.rdrag;
    if (on_u) {
      const double bu = 0.5 * (r1[a - 1] + r1[a]) * u[a];
      c->F.bustr[a] = bu;
      if (b.south_edge && !b.NSperiodic && j == b.Jstr) c->F.bustr[a - ni] = p.gamma2 * bu;
      if (b.north_edge && !b.NSperiodic && j == b.Jend) c->F.bustr[a + ni] = p.gamma2 * bu;
    }
    if (on_v) c->F.bvstr[a] = 0.5 * (r1[a - ni] + r1[a]) * v[a];
  }
  // bc_v2d_tile, closed walls: normal component zero on the wall rows
  if (i >= b.Istr && i <= b.Iend && !b.NSperiodic) {
    if (b.south_edge && j == b.Jstr) c->F.bvstr[a] = 0.0;
    if (b.north_edge && j == b.Jend) c->F.bvstr[a + ni] = 0.0;
  }
}

// stability functions, bulk_flux.F:1020-1108
__device__ __forceinline__ double bulk_psiu(double ZoL, double pi)
{
  const double r3 = 1.0 / 3.0;
  if (ZoL < 0.0) {
    const double x = pow(1.0 - 15.0 * ZoL, 0.25);
    const double psik = 2.0 * log(0.5 * (1.0 + x)) + log(0.5 * (1.0 + x * x)) - 2.0 * atan(x) + 0.5 * pi;
    double cff = sqrt(3.0);
    const double y = pow(1.0 - 10.15 * ZoL, r3);
    const double psic = 1.5 * log(r3 * (1.0 + y + y * y)) - cff * atan((1.0 + 2.0 * y) / cff) + pi / cff;
    cff = ZoL * ZoL;
    const double Fw = cff / (1.0 + cff);
    return (1.0 - Fw) * psik + Fw * psic;
  }
  const double cff = fmin(50.0, 0.35 * ZoL);
  return -((1.0 + ZoL) + 0.6667 * (ZoL - 14.28) / exp(cff) + 8.525);
}
__device__ __forceinline__ double bulk_psit(double ZoL, double pi)
{
  const double r3 = 1.0 / 3.0;
  if (ZoL < 0.0) {
    const double x = pow(1.0 - 15.0 * ZoL, 0.5);
    const double psik = 2.0 * log(0.5 * (1.0 + x));
    double cff = sqrt(3.0);
    const double y = pow(1.0 - 34.15 * ZoL, r3);
    const double psic = 1.5 * log(r3 * (1.0 + y + y * y)) - cff * atan((1.0 + 2.0 * y) / cff) + pi / cff;
    cff = ZoL * ZoL;
    const double Fw = cff / (1.0 + cff);
    return (1.0 - Fw) * psik + Fw * psic;
  }
  const double cff = fmin(50.0, 0.35 * ZoL);
  return -(pow(1.0 + 2.0 * ZoL, 1.5) + 0.6667 * (ZoL - 14.28) / exp(cff) + 8.525);
}

// one (i,j): wind stress components Taux, Tauy (N/m2) and the heat fluxes LRad, LHeat, SHeat (W/m2)
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_bulk_flux(const RomsDev *__restrict__ c, int nrhs, double *__restrict__ Taux, double *__restrict__ Tauy)
{
  DEV_PROLOGUE(c)
  const roms_params_t &p = c->p;
  const int i = b.Istr - 1 + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.Jstr - 1 + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendR || j > b.JendR) return;
  const long a = I2(i, j);
  const double Cp = 3985.0, StefBo = 5.67E-8, emmiss = 0.97, vonKar = 0.41;       // mod_scalars.F:431-444
  const double blk_Cpa = 1004.67, blk_Cpw = 4000.0, blk_Rgas = 287.1, blk_Zabl = 600.0, blk_beta = 1.2;
  const double pi = 3.14159265358979323846;
  const double g = p.g, rho0 = p.rho0;
  const double blk_ZQ = p.blk_ZQ, blk_ZT = p.blk_ZT, blk_ZW = p.blk_ZW;
  const double eps = 1.0E-20, r3 = 1.0 / 3.0;
  const double Ua = c->F.Uwind[a], Va = c->F.Vwind[a];
  const double Wmag = sqrt(Ua * Ua + Va * Va);
  const double PairM = c->F.Pair[a];
  const double TairC = c->F.Tair[a], TairK = TairC + 273.16;
  const double TseaC = c->F.t[a + (long)(N - 1) * nij + (long)(nrhs - 1) * n3r], TseaK = TseaC + 273.16;
  const double RH = c->F.Hair[a];
  const double cl = c->F.cloud[a], rn = c->F.rain[a];
  const double delTc = 0.0, delQc = 0.0;
  double cff = (0.7859 + 0.03477 * TairC) / (1.0 + 0.00412 * TairC);
  const double e_sat = pow(10.0, cff);
  const double vap_p = e_sat * RH;
  double cff2 = TairK * TairK * TairK;
  double cff1 = cff2 * TairK;
  const double LRad = -emmiss * StefBo *
                      (cff1 * (0.39 - 0.05 * sqrt(vap_p)) * (1.0 - 0.6823 * cl * cl) + cff2 * 4.0 * (TseaK - TairK));
  cff = (1.0007 + 3.46E-6 * PairM) * 6.1121 * exp(17.502 * TairC / (240.97 + TairC));
  const double Qair = 0.62197 * (cff / (PairM - 0.378 * cff));
  double Q;
  if (RH < 2.0) {
    cff = cff * RH;
    Q = 0.62197 * (cff / (PairM - 0.378 * cff));
  } else {
    Q = RH / 1000.0;
  }
  cff = (1.0007 + 3.46E-6 * PairM) * 6.1121 * exp(17.502 * TseaC / (240.97 + TseaC));
  cff = cff * 0.98;
  const double Qsea = 0.62197 * (cff / (PairM - 0.378 * cff));
  const double rhoAir = PairM * 100.0 / (blk_Rgas * TairK * (1.0 + 0.61 * Q));
  const double VisAir = 1.326E-5 * (1.0 + TairC * (6.542E-3 + TairC * (8.301E-6 - 4.84E-9 * TairC)));
  const double Hlv = (2.501 - 0.00237 * TseaC) * 1.0E+6;
  double Wgus = 0.5;
  double delW = sqrt(Wmag * Wmag + Wgus * Wgus);
  const double delQ = Qsea - Q;
  const double delT = TseaC - TairC;
  double ZoW = 0.0001;
  const double u10 = delW * log(10.0 / ZoW) / log(blk_ZW / ZoW);
  double Wstar = 0.035 * u10;
  const double Zo10 = 0.011 * Wstar * Wstar / g + 0.11 * VisAir / Wstar;
  const double Cd10 = (vonKar / log(10.0 / Zo10)) * (vonKar / log(10.0 / Zo10));
  const double Ch10 = 0.00115;
  const double Ct10 = Ch10 / sqrt(Cd10);
  const double ZoT10 = 10.0 / exp(vonKar / Ct10);
  double Cd = (vonKar / log(blk_ZW / Zo10)) * (vonKar / log(blk_ZW / Zo10));
  const double Ct = vonKar / log(blk_ZT / ZoT10);
  const double CC = vonKar * Ct / Cd;
  const double Ribcu = -blk_ZW / (blk_Zabl * 0.004 * (blk_beta * blk_beta * blk_beta));
  const double Ri = -g * blk_ZW * ((delT - delTc) + 0.61 * TairK * delQ) / (TairK * delW * delW);
  double Zetu;
  if (Ri < 0.0) Zetu = CC * Ri / (1.0 + Ri / Ribcu);
  else Zetu = CC * Ri / (1.0 + 3.0 * Ri / CC);
  const double L10 = blk_ZW / Zetu;
  Wstar = delW * vonKar / (log(blk_ZW / Zo10) - bulk_psiu(blk_ZW / L10, pi));
  double Tstar = -(delT - delTc) * vonKar / (log(blk_ZT / ZoT10) - bulk_psit(blk_ZT / L10, pi));
  double Qstar = -(delQ - delQc) * vonKar / (log(blk_ZQ / ZoT10) - bulk_psit(blk_ZQ / L10, pi));
  double charn;
  if (delW > 18.0) charn = 0.018;
  else if ((10.0 < delW) && (delW <= 18.0)) charn = 0.011 + 0.125 * (0.018 - 0.011) * (delW - 10.);
  else charn = 0.011;
  for (int Iter = 1; Iter <= 3; Iter++) {
    ZoW = charn * Wstar * Wstar / g + 0.11 * VisAir / (Wstar + eps);
    const double Rr = ZoW * Wstar / VisAir;
    const double ZoQ = fmin(1.15e-4, 5.5e-5 / pow(Rr, 0.6));
    const double ZoT = ZoQ;
    const double ZoL = vonKar * g * blk_ZW * (Tstar * (1.0 + 0.61 * Q) + 0.61 * TairK * Qstar) /
                       (TairK * Wstar * Wstar * (1.0 + 0.61 * Q) + eps);
    const double L = blk_ZW / (ZoL + eps);
    const double Wpsi = bulk_psiu(ZoL, pi);
    const double Tpsi = bulk_psit(blk_ZT / L, pi);
    const double Qpsi = bulk_psit(blk_ZQ / L, pi);
    Wstar = fmax(eps, delW * vonKar / (log(blk_ZW / ZoW) - Wpsi));
    Tstar = -(delT - delTc) * vonKar / (log(blk_ZT / ZoT) - Tpsi);
    Qstar = -(delQ - delQc) * vonKar / (log(blk_ZQ / ZoQ) - Qpsi);
    const double Bf = -g / TairK * Wstar * (Tstar + 0.61 * TairK * Qstar);
    if (Bf > 0.0) Wgus = blk_beta * pow(Bf * blk_Zabl, r3);
    else Wgus = 0.2;
    delW = sqrt(Wmag * Wmag + Wgus * Wgus);
  }
  const double Wspeed = sqrt(Wmag * Wmag + Wgus * Wgus);
  Cd = Wstar * Wstar / (Wspeed * Wspeed + eps);
  const double Hs = -blk_Cpa * rhoAir * Wstar * Tstar;
  const double diffw = 2.11E-5 * pow(TairK / 273.16, 1.94);
  const double diffh = 0.02411 * (1.0 + TairC * (3.309E-3 - 1.44E-6 * TairC)) / (rhoAir * blk_Cpa);
  cff = Qair * Hlv / (blk_Rgas * TairK * TairK);
  const double wet_bulb = 1.0 / (1.0 + 0.622 * (cff * Hlv * diffw) / (blk_Cpa * diffh));
  const double Hsr = rn * wet_bulb * blk_Cpw * ((TseaC - TairC) + (Qsea - Q) * Hlv / blk_Cpa);
  const double SHeat = (Hs + Hsr);
  const double Hl = -Hlv * rhoAir * Wstar * Qstar;
  const double upvel = -1.61 * Wstar * Qstar - (1.0 + 1.61 * Q) * Wstar * Tstar / TairK;
  const double Hlw = rhoAir * Hlv * upvel * Q;
  const double LHeat = (Hl + Hlw);
  const double Taur = 0.85 * rn * Wmag;
  cff = rhoAir * Cd * Wspeed;
  Taux[a] = (cff * Ua + Taur * copysign(1.0, Ua));
  Tauy[a] = (cff * Va + Taur * copysign(1.0, Va));
  // kinematic heat fluxes on the owned range, bulk_flux.F:790-812
  if (i >= b.IstrR && j >= b.JstrR) {
    const double Hscale = 1.0 / (rho0 * Cp);
    const double lr = LRad * Hscale, lh = -LHeat * Hscale, sh = -SHeat * Hscale;
    c->F.lrflx[a] = lr;
    c->F.lhflx[a] = lh;
    c->F.shflx[a] = sh;
    c->F.stflux[a] = (c->F.srflx[a] + lr + lh + sh);
  }
}

// kinematic wind stress at u- and v-points, bulk_flux.F:846-858
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_bulk_stress(const RomsDev *__restrict__ c, const double *__restrict__ Taux, const double *__restrict__ Tauy)
{
  DEV_PROLOGUE(c)
  const int i = b.IstrR + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.JstrR + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendR || j > b.JendR) return;
  const long a = I2(i, j);
  const double cff = 0.5 / c->p.rho0;
  if (i >= b.Istr) c->F.sustr[a] = cff * (Taux[a - 1] + Taux[a]);
  if (j >= b.Jstr) c->F.svstr[a] = cff * (Tauy[a - ni] + Tauy[a]);
}

}  // namespace

extern "C" int roms_hip_set_vbc(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_set_vbc");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  const roms_bounds_t &b = g_ctx.b;
  if (g_ctx.p.uv_drag != 1 && g_ctx.p.uv_drag != 2)
    return roms_fail("roms_hip_set_vbc", "bottom drag law not implemented (UV_LDRAG / UV_QDRAG only)");
  {
    ScopedTimer tm("set_vbc");
    hipLaunchKernelGGL(k_set_vbc, grid2d(b.IendR - b.IstrR + 1, b.JendR - b.JstrR + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s->nrhs);
    KERNEL_CHECK("k_set_vbc");
  }
  halo_batch_begin();
  halo_exchange2d(GT_U, g_ctx.dev[FID_bustr]);
  halo_exchange2d(GT_V, g_ctx.dev[FID_bvstr]);
  return halo_batch_end();
}

extern "C" int roms_hip_bulk_flux(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_bulk_flux");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  const roms_bounds_t &b = g_ctx.b;
  double *Taux = g_ctx.hostc.ws2[6], *Tauy = g_ctx.hostc.ws2[7];
  {
    ScopedTimer tm("bulk_flux");
    hipLaunchKernelGGL(k_bulk_flux, grid2d(b.IendR - (b.Istr - 1) + 1, b.JendR - (b.Jstr - 1) + 1), block2d(), 0,
                       g_ctx.stream, g_ctx.devc, s->nrhs, Taux, Tauy);
    KERNEL_CHECK("k_bulk_flux");
    hipLaunchKernelGGL(k_bulk_stress, grid2d(b.IendR - b.IstrR + 1, b.JendR - b.JstrR + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, (const double *)Taux, (const double *)Tauy);
    KERNEL_CHECK("k_bulk_stress");
  }
  halo_batch_begin();
  halo_exchange2d(GT_R, g_ctx.dev[FID_lrflx]);
  halo_exchange2d(GT_R, g_ctx.dev[FID_lhflx]);
  halo_exchange2d(GT_R, g_ctx.dev[FID_shflx]);
  halo_exchange2d(GT_R, g_ctx.dev[FID_stflux]);        // first plane = itemp
  halo_exchange2d(GT_U, g_ctx.dev[FID_sustr]);
  halo_exchange2d(GT_V, g_ctx.dev[FID_svstr]);
  return halo_batch_end();
}
