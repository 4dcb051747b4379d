// k_physics.hip -- per-step physics between the hot kernels, kept on the
// device so that the state never crosses PCIe inside a step:
//   roms_hip_set_vbc    set_vbc_tile   ROMS/Nonlinear/set_vbc.F:104   (UV_QDRAG / UV_LDRAG, SALINITY)
//   roms_hip_bulk_flux  bulk_flux_tile ROMS/Nonlinear/bulk_flux.F:146 (COARE 3.0 with the Berliand
//                       longwave formula; no COOL_SKIN / EMINUSP / WIND_MINUS_CURRENT; MASKING multiplies)
// Both are point-local (bulk_flux: three fixed iterations per point); one thread per (i,j).
// bulk_flux uses device log/exp/pow/atan: results agree with the host libraries to a few ulp, not
// bit for bit (the tests state the tolerance).
#include "roms_dev.h"
#include <cmath>

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
  const gcd_t u = (gcd_t)(c->F.u + (long)(nrhs - 1) * n3r);
  const gcd_t v = (gcd_t)(c->F.v + (long)(nrhs - 1) * n3r);
  // kinematic surface / bottom tracer fluxes, set_vbc.F:262-312
  GF(stflx)[a] = GF(stflux)[a];
  GF(btflx)[a] = GF(btflux)[a];
  if (p.salinity && NT >= 2) {
    const gcd_t S = (gcd_t)(c->F.t + ((long)(nrhs - 1) + 3L * 1) * n3r);      // t(:,:,:,nrhs,isalt)
    const double EmP = GF(stflux)[a + nij];
    GF(stflx)[a + nij] = EmP * S[a + (long)(N - 1) * nij];
    GF(btflx)[a + nij] = GF(btflx)[a + nij] * S[a];
  }
  // bottom stress, :380-470 (k = 1 is the first plane of u, v)
  const bool on_u = i >= b.IstrU && i <= b.Iend && j >= b.Jstr && j <= b.Jend;
  const bool on_v = i >= b.Istr && i <= b.Iend && j >= b.JstrV && j <= b.Jend;
  // LIMIT_BSTRESS, set_vbc.F:533-540, :562-567: the stress may only slow the bottom velocity down to zero within a step
  const gcd_t Hz1 = (gcd_t)c->F.Hz;
  auto lim_u = [&](double bs) {
    if (!p.limit_bstress) return bs;
    const double cff3 = (0.75 / p.dt) * 0.5 * (Hz1[a - 1] + Hz1[a]);
    return copysign(1.0, bs) * fmin(fabs(bs), fabs(u[a]) * cff3);
  };
  auto lim_v = [&](double bs) {
    if (!p.limit_bstress) return bs;
    const double cff3 = (0.75 / p.dt) * 0.5 * (Hz1[a - ni] + Hz1[a]);
    return copysign(1.0, bs) * fmin(fabs(bs), fabs(v[a]) * cff3);
  };
  if (p.uv_drag == 2) {
    const gcd_t r2 = (gcd_t)(c->F.rdrag2);
    if (on_u) {
      const double cff1 = 0.25 * (v[a] + v[a + ni] + v[a - 1] + v[a - 1 + ni]);
      const double cff2 = sqrt(u[a] * u[a] + cff1 * cff1);
      const double bu = lim_u(0.5 * (r2[a - 1] + r2[a]) * u[a] * cff2);
      GF(bustr)[a] = bu;
      if (b.south_edge && !b.NSperiodic && j == b.Jstr) GF(bustr)[a - ni] = p.gamma2 * bu;   // bc_u2d_tile
      if (b.north_edge && !b.NSperiodic && j == b.Jend) GF(bustr)[a + ni] = p.gamma2 * bu;
    }
    if (on_v) {
      const double cff1 = 0.25 * (u[a] + u[a + 1] + u[a - ni] + u[a + 1 - ni]);
      const double cff2 = sqrt(cff1 * cff1 + v[a] * v[a]);
      GF(bvstr)[a] = lim_v(0.5 * (r2[a - ni] + r2[a]) * v[a] * cff2);
    }
  } else if (p.uv_drag == 3) {      // UV_LOGDRAG, set_vbc.F:542-580
    const gcd_t zr = (gcd_t)c->F.z_r, zw = (gcd_t)c->F.z_w, zo = (gcd_t)c->F.ZoBot;
    auto cd = [&](long q) {          // wrk(i,j): (vonKar / LOG(dz/ZoBot))^2 within [Cdb_min, Cdb_max]
      const double vonKar = 0.41;
      const double cff1 = 1.0 / log((zr[q] - zw[q]) / zo[q]);
      const double cff2 = vonKar * vonKar * cff1 * cff1;
      return fmin(p.Cdb_max, fmax(p.Cdb_min, cff2));
    };
    if (on_u) {
      const double cff1 = 0.25 * (v[a] + v[a + ni] + v[a - 1] + v[a - 1 + ni]);
      const double cff2 = sqrt(u[a] * u[a] + cff1 * cff1);
      const double bu = lim_u(0.5 * (cd(a - 1) + cd(a)) * u[a] * cff2);
      GF(bustr)[a] = bu;
      if (b.south_edge && !b.NSperiodic && j == b.Jstr) GF(bustr)[a - ni] = p.gamma2 * bu;   // bc_u2d_tile
      if (b.north_edge && !b.NSperiodic && j == b.Jend) GF(bustr)[a + ni] = p.gamma2 * bu;
    }
    if (on_v) {
      const double cff1 = 0.25 * (u[a] + u[a + 1] + u[a - ni] + u[a + 1 - ni]);
      const double cff2 = sqrt(cff1 * cff1 + v[a] * v[a]);
      GF(bvstr)[a] = lim_v(0.5 * (cd(a - ni) + cd(a)) * v[a] * cff2);
    }
  } else {
    const gcd_t r1 = (gcd_t)(c->F.rdrag);
    if (on_u) {
      const double bu = lim_u(0.5 * (r1[a - 1] + r1[a]) * u[a]);
      GF(bustr)[a] = bu;
      if (b.south_edge && !b.NSperiodic && j == b.Jstr) GF(bustr)[a - ni] = p.gamma2 * bu;
      if (b.north_edge && !b.NSperiodic && j == b.Jend) GF(bustr)[a + ni] = p.gamma2 * bu;
    }
    if (on_v) GF(bvstr)[a] = lim_v(0.5 * (r1[a - ni] + r1[a]) * v[a]);
  }
  // bc_v2d_tile, closed walls: normal component zero on the wall rows
  if (i >= b.Istr && i <= b.Iend && !b.NSperiodic) {
    if (b.south_edge && j == b.Jstr) GF(bvstr)[a] = 0.0;
    if (b.north_edge && j == b.Jend) GF(bvstr)[a + ni] = 0.0;
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
  const double Ua = GF(Uwind)[a], Va = GF(Vwind)[a];
  const double Wmag = sqrt(Ua * Ua + Va * Va);
  const double PairM = GF(Pair)[a];
  const double TairC = GF(Tair)[a], TairK = TairC + 273.16;
  const double TseaC = GF(t)[a + (long)(N - 1) * nij + (long)(nrhs - 1) * n3r], TseaK = TseaC + 273.16;
  const double RH = GF(Hair)[a];
  const double cl = GF(cloud)[a], rn = GF(rain)[a];
  const double delTc = 0.0, delQc = 0.0;
  double cff = (0.7859 + 0.03477 * TairC) / (1.0 + 0.00412 * TairC);
  const double e_sat = pow(10.0, cff);
  const double vap_p = e_sat * RH;
  double cff2 = TairK * TairK * TairK;
  double cff1 = cff2 * TairK;
  // MASKING (bulk_flux.F:486, :790, :809, :824, :831, :877): the five fluxes and stflux times rmask; mr = 1 without
  const bool masking = c->p.masking != 0;
  const double mr = masking ? rmaskw(c, a) : 1.0;               // (+ WET_DRY: the block after each of them)
  double LRad = -emmiss * StefBo *
                (cff1 * (0.39 - 0.05 * sqrt(vap_p)) * (1.0 - 0.6823 * cl * cl) + cff2 * 4.0 * (TseaK - TairK));
  if (masking) LRad = LRad * mr;
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
  double SHeat = (Hs + Hsr);
  if (masking) SHeat = SHeat * mr;
  const double Hl = -Hlv * rhoAir * Wstar * Qstar;
  const double upvel = -1.61 * Wstar * Qstar - (1.0 + 1.61 * Q) * Wstar * Tstar / TairK;
  const double Hlw = rhoAir * Hlv * upvel * Q;
  double LHeat = (Hl + Hlw);
  if (masking) LHeat = LHeat * mr;
  const double Taur = 0.85 * rn * Wmag;
  cff = rhoAir * Cd * Wspeed;
  double tx = (cff * Ua + Taur * copysign(1.0, Ua)), ty = (cff * Va + Taur * copysign(1.0, Va));
  if (masking) { tx = tx * mr; ty = ty * mr; }
  Taux[a] = tx;
  Tauy[a] = ty;
  // kinematic heat fluxes on the owned range, bulk_flux.F:790-812
  if (i >= b.IstrR && j >= b.JstrR) {
    const double Hscale = 1.0 / (rho0 * Cp);
    const double lr = LRad * Hscale, lh = -LHeat * Hscale, sh = -SHeat * Hscale;
    GF(lrflx)[a] = lr;
    GF(lhflx)[a] = lh;
    GF(shflx)[a] = sh;
    double stf = (GF(srflx)[a] + lr + lh + sh);
    if (masking) stf = stf * mr;
    GF(stflux)[a] = stf;
    if (p.eminusp) {                               // EMINUSP, bulk_flux.F:883-899 (rhow = 1000, mod_scalars.F:437)
      const double cffw = 1.0 / 1000.0;
      double ev = LHeat / Hlv;
      if (masking) ev = ev * mr;
      GF(evap)[a] = ev;
      double sf = cffw * (ev - rn);
      if (masking) sf = sf * mr;
      GF(stflux)[a + nij] = sf;
    }
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
  const bool masking = c->p.masking != 0;                          // bulk_flux.F:908, :919
  if (i >= b.Istr) {
    const double su = cff * (Taux[a - 1] + Taux[a]);
    GF(sustr)[a] = masking ? su * umaskw(c, a) : su;          // (+ WET_DRY, :911)
  }
  if (j >= b.Jstr) {
    const double sv = cff * (Tauy[a - ni] + Tauy[a]);
    GF(svstr)[a] = masking ? sv * vmaskw(c, a) : sv;          // (+ WET_DRY, :922)
  }
}

}  // namespace

// ana_srflux_tile, ALBEDO branch (ROMS/Functionals/ana_srflux.h:120-150); point-local
namespace {
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_ana_srflux(const RomsDev *__restrict__ c, double Dangle, double Hangle, double Rsolar)
{
  DEV_PROLOGUE(c)
  const int i = b.IstrT + blockIdx.x * BLK_X + threadIdx.x;
  const int j = b.JstrT + blockIdx.y * BLK_Y + threadIdx.y;
  if (i > b.IendT || j > b.JendT) return;
  const long a = I2(i, j);
  const double pi = 3.14159265358979323846, deg2rad = pi / 180.0, alb_w = 0.06;
  const double LatRad = GF(latr)[a] * deg2rad;
  const double cff1 = sin(LatRad) * sin(Dangle);
  const double cff2 = cos(LatRad) * cos(Dangle);
  double sr = 0.0;
  const double zenith = cff1 + cff2 * cos(Hangle - GF(lonr)[a] * deg2rad);
  if (zenith > 0.0) {
    const double Ta = GF(Tair)[a];
    const double cff = (0.7859 + 0.03477 * Ta) / (1.0 + 0.00412 * Ta);
    const double e_sat = pow(10.0, cff);
    const double vap_p = e_sat * GF(Hair)[a];
    const double cl = GF(cloud)[a];
    sr = Rsolar * zenith * zenith * (1.0 - 0.6 * (cl * cl * cl)) / ((zenith + 2.7) * vap_p * 1.0E-3 + 1.085 * zenith + 0.1);
  }
  GF(srflx)[a] = (1.0 - alb_w) * sr;
}
}  // namespace

extern "C" int roms_hip_ana_srflux(double yday, double hour)
{
  int rc = roms_entry_check("roms_hip_ana_srflux");
  if (rc) return rc;
  const roms_bounds_t &b = g_ctx.b;
  // the scalar part of ana_srflux.h:120-127 on the host, as the reference does once per call
  const double pi = 3.14159265358979323846, deg2rad = pi / 180.0, Csolar = 1353.0, Cp = 3985.0;
  double Dangle = 23.44 * cos((172.0 - yday) * 2.0 * pi / 365.2425);
  Dangle = Dangle * deg2rad;
  const double Hangle = (12.0 - hour) * pi / 12.0;
  const double Rsolar = Csolar / (g_ctx.p.rho0 * Cp);
  ScopedTimer tm("ana_srflux");
  hipLaunchKernelGGL(k_ana_srflux, grid2d(b.IendT - b.IstrT + 1, b.JendT - b.JstrT + 1), block2d(), 0, g_ctx.stream,
                     g_ctx.devc, Dangle, Hangle, Rsolar);
  KERNEL_CHECK("k_ana_srflux");
  return 0;
}

extern "C" int roms_hip_set_vbc(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_set_vbc");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  const roms_bounds_t &b = g_ctx.b;
  if (g_ctx.p.uv_drag < 1 || g_ctx.p.uv_drag > 3)
    return roms_fail("roms_hip_set_vbc", "bottom drag law not implemented (UV_LDRAG / UV_QDRAG only)");
  {
    ScopedTimer tm("set_vbc");
    hipLaunchKernelGGL(k_set_vbc, grid2d(b.IendR - b.IstrR + 1, b.JendR - b.JstrR + 1), block2d(), 0, g_ctx.stream,
                       g_ctx.devc, s->nrhs);
    KERNEL_CHECK("k_set_vbc");
    // bc_u2d_tile / bc_v2d_tile (bc_2d.F:184, :386; isBu2d = isUbar, isBv2d = isVbar): the kernel applied the
    // closed-wall rule of an E-W periodic channel on its own rows; everything else -- western / eastern edges,
    // corners, open edges (zero gradient), the mask of the wall points -- through the generic edge kernel
    if (!b.EWperiodic || g_ctx.p.masking || !lbc2d_all_closed()) {
      if ((rc = bc_generic(LBV_UBAR, LBV_UBAR, g_ctx.dev[FID_bustr], 1))) return rc;
      if ((rc = bc_generic(LBV_VBAR, LBV_VBAR, g_ctx.dev[FID_bvstr], 1))) return rc;
    }
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
  if (g_ctx.p.eminusp) {                               // bulk_flux.F:945-952
    halo_exchange2d(GT_R, g_ctx.dev[FID_evap]);
    halo_exchange2d(GT_R, g_ctx.dev[FID_stflux] + (long)(g_ctx.b.UBi - g_ctx.b.LBi + 1) * (g_ctx.b.UBj - g_ctx.b.LBj + 1));
  }
  halo_exchange2d(GT_U, g_ctx.dev[FID_sustr]);
  halo_exchange2d(GT_V, g_ctx.dev[FID_svstr]);
  return halo_batch_end();
}

// =================================================================================================
// lmd_vmix = lmd_vmix_tile + lmd_skpp_tile + lmd_finish_tile (ROMS/Nonlinear/lmd_vmix.F:99/465,
// lmd_skpp.F:98, lmd_swfrac.F:6): K-profile vertical mixing with the BENCHMARK option set (LMD_RIMIX
// + RI_SPLINES, LMD_CONVEC, LMD_SKPP, LMD_NONLOCAL, SALINITY; uniform Jerlov water type).
// Column-local: one thread per (i,j), three sweeps.  Upward: the forward recurrences of the three splines
// (u, v, pden), of which only every sixth level is kept (registers).  Downward, segment by segment: the
// recurrences of a segment are re-run from its checkpoint (registers) and their back-substitution is fused with
// the Richardson-number mixing, the bulk Richardson function and the boundary-layer depth search, so no profile
// goes to device memory at all: the shear function nu_sx waits in LDS for the last sweep; the buoyancy-flux profile
// is recomputed (two exp) where used.  Upward again: the boundary-layer profiles, the convective adjustment, the
// final Akv, Akt.  Per column and level 4 + 4 + 2 + 2 array reads from memory (4 more from cache) and 5 writes -- the
// first version (forward values of all levels in scratch) read 18 and wrote 11.
// =================================================================================================
namespace {

__device__ __forceinline__ double swfrac(const roms_params_t &p, double Z)     // lmd_swfrac.F:60-75, Zscale = -1
{
  const double fac1 = -1.0 / p.swfrac_mu1, fac2 = -1.0 / p.swfrac_mu2, fac3 = p.swfrac_r1;
  return exp(Z * fac1) * fac3 + exp(Z * fac2) * (1.0 - fac3);
}

__device__ __forceinline__ void wscale(double Ustar, double sigma, double Bf, double &wm, double &ws)
{
  const double vonKar = 0.41, small = 1.0E-20;
  const double lmd_am = 1.257, lmd_as = -28.86, lmd_cm = 8.36, lmd_cs = 98.96, lmd_zetam = -0.2, lmd_zetas = -1.0;
  const double Ustar3 = Ustar * Ustar * Ustar;
  const double zetahat = vonKar * sigma * Bf;
  const double zetapar = zetahat / (Ustar3 + small);
  if (zetahat >= 0.0) {
    wm = vonKar * Ustar / (1.0 + 5.0 * zetapar);
    ws = wm;
  } else {
    // x**0.25, x**0.5, x**(1/3) of lmd_skpp.F:441-452 as sqrt(sqrt(x)), sqrt(x), cbrt(x): each within an
    // ulp of the reference's pow (whose device version is not bit-identical to the host's either) at a
    // fraction of the cost -- the kernel is bound by these calls, ~4 per level and sweep
    if (zetapar > lmd_zetam) wm = vonKar * Ustar * sqrt(sqrt(1.0 - 16.0 * zetapar));
    else wm = vonKar * cbrt(lmd_am * Ustar3 - lmd_cm * zetahat);
    if (zetapar > lmd_zetas) ws = vonKar * Ustar * sqrt(1.0 - 16.0 * zetapar);
    else ws = vonKar * cbrt(lmd_as * Ustar3 - lmd_cs * zetahat);
  }
}

// The forward recurrences of the three splines are NOT stored per level: the upward sweep keeps only their values at
// every LMD_SEG-th level (checkpoints, in registers); the downward sweep takes the column segment by segment from the
// top, re-runs the recurrence of a segment from its checkpoint and consumes it in descending order.  Same operations
// in the same order as a single upward sweep, so the same bits -- without the 4 x (N-1) scratch stores and loads per
// column of the earlier version.  All three sweeps load a whole segment (LMD_SEG levels of every input) before they
// compute it: a wave has a dozen memory round trips per column instead of one per level and sweep (the kernel was
// waiting on memory for 72 % of its wave-cycles, SQ_WAIT_ANY, with one level's loads in flight).
#define LMD_SEG 5

template <int NMAX>
__global__ void __launch_bounds__(BLK_X *BLK_Y, 2)
k_lmd_vmix(const RomsDev *__restrict__ c, int nstp, double lmd_Cg, double Vtc)
{
  DEV_PROLOGUE(c)
  const roms_params_t &p = c->p;
  constexpr int MS = (NMAX - 1 + LMD_SEG - 1) / LMD_SEG;        // segments of W-levels 1..NMAX-1
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const long a = I2(i, j);
  const double g = p.g, vonKar = 0.41;
  const double lmd_Ri0 = 0.7, lmd_bvfcon = -2.0E-5, lmd_nu0c = 0.01, lmd_nu0m = 10.0E-4, lmd_nu0s = 10.0E-4;
  const double lmd_Ric = 0.3, lmd_cekman = 0.7, lmd_cmonob = 1.0, lmd_epsilon = 0.1;
  const double gorho0 = g / p.rho0;
  const gcd_t Hz = (gcd_t)c->F.Hz, pden = (gcd_t)c->F.pden, bvf = (gcd_t)c->F.bvf;
  const gcd_t z_w = (gcd_t)c->F.z_w;
  const gcd_t u = (gcd_t)(c->F.u + (long)(nstp - 1) * n3r), v = (gcd_t)(c->F.v + (long)(nstp - 1) * n3r);
  const gd_t Akv = (gd_t)c->F.Akv, AkT = (gd_t)c->F.Akt, AkS = (gd_t)(c->F.Akt + n3w);
  const gd_t ghT = (gd_t)c->F.ghats, ghS = (gd_t)(c->F.ghats + n3w);
  auto r3i = [&](int k) { return a + (long)(k - 1) * nij; };     // rho-type level k = 1..N
  auto w3i = [&](int k) { return a + (long)k * nij; };           // W-type level k = 0..N
  // the shear function nu_sx of W-levels 1..N-1 waits in LDS ([level][thread]) for the last sweep -- it used to be
  // parked in Akv: one field written and read back for nothing (0.13 GB each way on BENCHMARK3)
  constexpr int NTH = BLK_X * BLK_Y;
  __shared__ double s_nu[(NMAX - 1) * NTH];
  double *const nu_col = s_nu + threadIdx.y * BLK_X + threadIdx.x;        // nu_col[(k - 1) * NTH], k = 1..N-1

  // the inputs of rho-levels k0+1 .. k0+SEG+1 (clamped to N: the clamped ones are loaded but not used)
  struct Lev { double hz, ua, ub, va, vb, pd; };
  auto load_seg = [&](int k0, Lev (&L)[LMD_SEG + 1]) {
#pragma unroll
    for (int t = 0; t <= LMD_SEG; t++) {
      const long q = r3i(min(k0 + 1 + t, N));
      L[t].hz = Hz[q]; L[t].ua = u[q]; L[t].ub = u[q + 1]; L[t].va = v[q]; L[t].vb = v[q + ni]; L[t].pd = pden[q];
    }
  };
  // one level of the spline recurrences of u, v (lmd_vmix_tile :205-225 = lmd_skpp_tile :345-365) and of pden:
  // from the values of level k-1 (FCm ...) and the inputs of rho-levels k (A) and k+1 (B)
#define LMD_FWD(A, B, fck, duk, dvk, drk)                                             \
      const double cff = 1.0 / (2.0 * B.hz + A.hz * (2.0 - FCm));                     \
      const double fck = cff * B.hz;                                                  \
      const double duk = cff * (3.0 * (B.ua - A.ua + B.ub - A.ub) - A.hz * dUm);      \
      const double dvk = cff * (3.0 * (B.va - A.va + B.vb - A.vb) - A.hz * dVm);      \
      const double drk = cff * (6.0 * (B.pd - A.pd) - A.hz * dRm);

  // ---------- upward sweep: the recurrences, keeping their values at levels 0, SEG, 2 SEG ... only ----------
  double ckF[MS], ckU[MS], ckV[MS], ckR[MS];
  {
    double FCm = 0.0, dUm = 0.0, dVm = 0.0, dRm = 0.0;
#pragma unroll
    for (int m = 0; m < MS; m++) {
      ckF[m] = FCm; ckU[m] = dUm; ckV[m] = dVm; ckR[m] = dRm;
      if (m * LMD_SEG + 1 <= N - 1) {
        Lev L[LMD_SEG + 1];
        load_seg(m * LMD_SEG, L);
#pragma unroll
        for (int t = 0; t < LMD_SEG; t++) {
          if (m * LMD_SEG + 1 + t <= N - 1) {
            LMD_FWD(L[t], L[t + 1], fck, duk, dvk, drk)
            FCm = fck; dUm = duk; dVm = dvk; dRm = drk;
          }
        }
      }
    }
  }

  // ---------- surface forcing of the boundary layer, lmd_skpp.F:300-340 ----------
  const double eps = 1.0E-10;
  const double zwN = z_w[w3i(N)];
  double hsbl = GF(hsbl)[a];
  double sl_dpth = lmd_epsilon * (zwN - hsbl);
  const double s1 = 0.5 * (GF(sustr)[a] + GF(sustr)[a + 1]), s2 = 0.5 * (GF(svstr)[a] + GF(svstr)[a + ni]);
  // MASKING (lmd_skpp.F:272-866): mr = rmask, applied where the reference applies it
  const bool masking = p.masking != 0;
  const double mr = masking ? (double)GF(rmask)[a] : 1.0;
  double Ustar = sqrt(sqrt(s1 * s1 + s2 * s2));
  if (masking) Ustar = Ustar * mr;                               // :272
  const double alpha = GF(alpha)[a], beta = GF(beta)[a], srflx = GF(srflx)[a];
  const double stT = GF(stflx)[a], stS = GF(stflx)[a + nij];
  const double Bo = g * (alpha * (stT - srflx) - beta * stS);
  const double Bosol = g * alpha * srflx;
  // buoyancy flux and the non-local flux shape at W-level k (:320-338); recomputed where needed
  // (two exp per call) instead of being stored and re-read
  auto bflux_z = [&](double zwk, double &gT, double &gS) {
    const double swdk = swfrac(p, zwN - zwk);
    double bf = (Bo + Bosol * (1.0 - swdk));
    if (masking) bf = bf * mr;                                   // :316
    const double cff = 1.0 - (0.5 + copysign(0.5, bf));
    gT = -cff * (stT - srflx + srflx * (1.0 - swdk));
    gS = cff * stS;
    return bf;
  };
  auto bflux = [&](int k, double &gT, double &gS) { return bflux_z(z_w[w3i(k)], gT, gS); };
  { double gT, gS; bflux(N, gT, gS); ghT[w3i(N)] = gT; ghS[w3i(N)] = gS; }

  // ---------- downward sweep, segment by segment: back-substitution of the three splines fused with (a) the shear /
  // Richardson-number mixing of lmd_vmix_tile (:240-300) and (b) the bulk Richardson function and the
  // boundary-layer depth search of lmd_skpp_tile (:380-470) ----------
  const double cff1 = 1.0 / 3.0, cff2 = 1.0 / 6.0;
  int ksbl = 1;
  {
    double Rref = 0.0, Uref = 0.0, Vref = 0.0;
    double dRk = 0.0, dUk = 0.0, dVk = 0.0;                      // final values at level k (k = N: 0)
    double FCk = 0.0;                                            // FC(i,N) = 0
    double bvk = 0.0, zwk = zwN;                                 // W-type level k (bvf(N) is not used)
    hsbl = z_w[w3i(1)];
    const int mtop = (N - 2) / LMD_SEG;                          // the segment of W-level N-1
    for (int m = mtop; m >= 0; m--) {
      const int k0 = m * LMD_SEG;                                // this segment: W-levels k0+1 .. min(k0+SEG, N-1)
      // everything the segment reads, in flight together: rho-levels k0+1 .. k0+SEG+1, bvf and z_w of W-levels
      // k0 .. k0+SEG (iteration k works on W-level k-1)
      double bvW[LMD_SEG + 1], zwW[LMD_SEG + 1];
#pragma unroll
      for (int t = 0; t <= LMD_SEG; t++) {
        const long qz = w3i(min(k0 + t, N - 1));
        bvW[t] = bvf[qz]; zwW[t] = z_w[qz];
      }
      // re-run the recurrences of the segment from its checkpoint
      double fF[LMD_SEG], fU[LMD_SEG], fV[LMD_SEG], fR[LMD_SEG];
      double hzA[LMD_SEG + 1], pdA[LMD_SEG + 1], umA[LMD_SEG + 1], vmA[LMD_SEG + 1];     // what the iterations need of L
      {
        Lev L[LMD_SEG + 1];
        load_seg(k0, L);
        double FCm = ckF[0], dUm = ckU[0], dVm = ckV[0], dRm = ckR[0];
#pragma unroll
        for (int t = 1; t < MS; t++)
          if (m == t) { FCm = ckF[t]; dUm = ckU[t]; dVm = ckV[t]; dRm = ckR[t]; }
#pragma unroll
        for (int t = 0; t < LMD_SEG; t++) {
          fF[t] = 0.0; fU[t] = 0.0; fV[t] = 0.0; fR[t] = 0.0;
          if (k0 + 1 + t <= N - 1) {
            LMD_FWD(L[t], L[t + 1], fck, duk, dvk, drk)
            fF[t] = fck; fU[t] = duk; fV[t] = dvk; fR[t] = drk;
            FCm = fck; dUm = duk; dVm = dvk; dRm = drk;
          }
        }
#pragma unroll
        for (int t = 0; t <= LMD_SEG; t++) {
          hzA[t] = L[t].hz; pdA[t] = L[t].pd; umA[t] = 0.5 * (L[t].ua + L[t].ub); vmA[t] = 0.5 * (L[t].va + L[t].vb);
        }
      }
      // iteration k (rho-level k = L[k-k0-1], forward values and bvf / z_w of W-level k-1) for k = k0+SEG+1 .. k0+2;
      // the lowest segment also runs k = 1, which has no level below
#pragma unroll
      for (int t = LMD_SEG; t >= 0; t--) {
        const int k = k0 + 1 + t;
        if (k <= N && (t >= 1 || m == 0)) {
          const double hz = hzA[t], pd = pdA[t], um = umA[t], vm = vmA[t];
          const int tw = t >= 1 ? t - 1 : 0;                     // slot of the forward values of W-level k-1
          const double fc = fF[tw], du = fU[tw], dv = fV[tw], dr = fR[tw];
          const double bvm = bvW[t], zwm = zwW[t];
          if (k == N) {                                          // the reference values at the surface: x(N) = 0
            Rref = pd + hz * (cff1 * 0.0 + cff2 * dr);
            Uref = um + hz * (cff1 * 0.0 + cff2 * du);
            Vref = vm + hz * (cff1 * 0.0 + cff2 * dv);
          }
          // final spline derivatives at level k-1
          double dRm = 0.0, dUm = 0.0, dVm = 0.0;
          if (k - 1 >= 1) {
            dRm = dr - fc * dRk;
            dUm = du - fc * dUk;
            dVm = dv - fc * dVk;
          }
          // (a) interior mixing at W-level k: only the shear function nu_sx is kept (in LDS); the last sweep forms
          // Akv and Akt from it and from bvf, which it reads anyway (lmd_vmix.F:286-297)
          if (k <= N - 1) {
            const double epsv = 1.0E-14;
            double shear2 = dUk * dUk + dVk * dVk;
            const double bv = bvk;
            const double Rig = bv / (shear2 + epsv);
            double cff = fmin(1.0, fmax(0.0, Rig) / lmd_Ri0);
            double nu_sx = 1.0 - cff * cff;
            nu_sx = nu_sx * nu_sx * nu_sx;
            shear2 = bv / (Rig + epsv);
            cff = shear2 * shear2 / (shear2 * shear2 + 16.0E-10);
            nu_sx = cff * nu_sx;
            nu_col[(k - 1) * NTH] = nu_sx;
          }
          // (b) bulk Richardson function at W-level k-1 -- until the boundary-layer depth is found: the levels below
          // it cannot change hsbl / ksbl any more (the search keeps the first crossing from the top)
          if (ksbl == 1) {
            const double depth = zwN - zwm;
            double gT, gS;
            const double bf = bflux_z(zwm, gT, gS);
            if (k - 1 == 0) { ghT[w3i(0)] = gT; ghS[w3i(0)] = gS; }
            const double sigma = (bf < 0.0) ? fmin(sl_dpth, depth) : depth;
            double wmk, wsk;
            wscale(Ustar, sigma, bf, wmk, wsk);
            const double Rk = pd - hz * (cff1 * dRm + cff2 * dRk);
            const double Uk = um - hz * (cff1 * dUm + cff2 * dUk);
            const double Vk = vm - hz * (cff1 * dVm + cff2 * dVk);
            const double Ritop = -gorho0 * (Rref - Rk) * depth;
            const double Ribot = (Uref - Uk) * (Uref - Uk) + (Vref - Vk) * (Vref - Vk) +
                                 Vtc * depth * wsk * sqrt(fabs(bvm));
            const double FCkm1 = Ritop - lmd_Ric * Ribot;
            // boundary-layer depth: first level (from the top, k = N..2) where the function turns positive
            if (k >= 2 && FCkm1 > 0.0) {
              hsbl = (zwk * FCkm1 - zwm * FCk) / (FCkm1 - FCk);
              ksbl = k;
            }
            FCk = FCkm1;
          } else if (k - 1 == 0) {                               // the non-local flux shape at the bottom (:320-338)
            double gT, gS;
            bflux_z(zwm, gT, gS);
            ghT[w3i(0)] = gT; ghS[w3i(0)] = gS;
          }
          dRk = dRm; dUk = dUm; dVk = dVm;
          bvk = bvm; zwk = zwm;
        }
      }
    }
  }
#undef LMD_FWD
  // the interior mixing coefficients of (a) at W-level k from the stored shear function: the same expressions
  // as lmd_vmix.F:286-297
  auto akv_a = [&](double nu_sx, double bv) { return 1.0E-6 * (1.0 / sqrt(fmax(bv, 1.0E-7))) + lmd_nu0m * nu_sx; };
  auto akt_a = [&](double nu_sx, double bv) { return 1.0E-7 * (1.0 / sqrt(fmax(bv, 1.0E-7))) + lmd_nu0s * nu_sx; };
  // buoyancy flux at the boundary-layer depth; MASKING: depth and flux times rmask (:562, :574 and :669, :681)
  auto bfsfc_at = [&](double hs) {
    double zgrid = zwN - hs;
    if (masking) zgrid = zgrid * mr;
    double bf = (Bo + Bosol * (1.0 - swfrac(p, zgrid)));
    if (masking) bf = bf * mr;
    return bf;
  };
  double Bfsfc = bfsfc_at(hsbl);
  if ((Ustar > 0.0) && (Bfsfc > 0.0)) {
    const double hekman = lmd_cekman * Ustar / fmax(fabs(GF(f)[a]), eps);
    const double hmonob = lmd_cmonob * Ustar * Ustar * Ustar / fmax(vonKar * Bfsfc, eps);
    hsbl = (zwN - fmin(fmin(hekman, hmonob), zwN - hsbl));
  }
  hsbl = fmin(hsbl, zwN);
  hsbl = fmax(hsbl, z_w[w3i(0)]);
  if (masking) hsbl = hsbl * mr;                                 // :595
  GF(hsbl)[a] = hsbl;
  if (!b.NSperiodic) {                                           // bc_r2d_tile: zero gradient at closed walls
    if (b.south_edge && j == b.Jstr) GF(hsbl)[a - ni] = hsbl;
    if (b.north_edge && j == b.Jend) GF(hsbl)[a + ni] = hsbl;
  }
  // ksbl = the first k from the top (N..2) with z_w(k-1) < hsbl (lmd_skpp.F:640-650); LMD_SEG levels per round trip
  ksbl = 1;
  for (int kt = N; kt >= 2 && ksbl == 1; kt -= LMD_SEG) {
    double zc[LMD_SEG];
#pragma unroll
    for (int t = 0; t < LMD_SEG; t++) zc[t] = z_w[w3i(max(kt - t - 1, 0))];
#pragma unroll
    for (int t = 0; t < LMD_SEG; t++)
      if ((ksbl == 1) && (kt - t >= 2) && (zc[t] < hsbl)) ksbl = kt - t;
  }
  Bfsfc = bfsfc_at(hsbl);
  sl_dpth = lmd_epsilon * (zwN - hsbl);
  double wm, ws;
  {
    const double cff = (Bfsfc > 0.0) ? 1.0 : lmd_epsilon;
    wscale(Ustar, cff * (zwN - hsbl), Bfsfc, wm, ws);
  }
  const double f1 = 5.0 * fmax(0.0, Bfsfc) * vonKar / (Ustar * Ustar * Ustar * Ustar + eps);
  const double zbl = zwN - hsbl;
  double Gm1, Gt1, Gs1, dGm1dS, dGt1dS, dGs1dS;
  if (hsbl > z_w[w3i(1)]) {
    const int k = ksbl;
    const double cff = 1.0 / (z_w[w3i(k)] - z_w[w3i(k - 1)]);
    const double cff_dn = cff * (hsbl - z_w[w3i(k - 1)]);
    const double cff_up = cff * (z_w[w3i(k)] - hsbl);
    // Akv / Akt of (a) at the two levels around the boundary-layer depth (level N keeps the array's own values)
    const double bv_k = bvf[w3i(k)], bv_m = bvf[w3i(k - 1)];
    const double nu_k = (k <= N - 1) ? nu_col[(k - 1) * NTH] : (double)Akv[w3i(k)], nu_m = nu_col[(k - 2) * NTH];
    const double akv_k = (k <= N - 1) ? akv_a(nu_k, bv_k) : nu_k, akv_m = akv_a(nu_m, bv_m);
    const double akt_k = (k <= N - 1) ? akt_a(nu_k, bv_k) : (double)AkT[w3i(k)], akt_m = akt_a(nu_m, bv_m);
    double K_bl = cff_dn * akv_k + cff_up * akv_m;
    double dK_bl = cff * (akv_k - akv_m);
    Gm1 = K_bl / (zbl * wm + eps);
    if (masking) Gm1 = Gm1 * mr;                                 // :754
    dGm1dS = fmin(0.0, -dK_bl / (wm + eps) - K_bl * f1);
    K_bl = cff_dn * akt_k + cff_up * akt_m;
    dK_bl = cff * (akt_k - akt_m);
    Gt1 = K_bl / (zbl * ws + eps);
    if (masking) Gt1 = Gt1 * mr;                                 // :765
    dGt1dS = fmin(0.0, -dK_bl / (ws + eps) - K_bl * f1);
    // salinity: Akt(isalt) = Akt(itemp) at the interior levels (lmd_vmix.F:296-297); level N was not touched
    const double aks_k = (k <= N - 1) ? akt_k : (double)AkS[w3i(k)], aks_km1 = akt_m;
    K_bl = cff_dn * aks_k + cff_up * aks_km1;
    dK_bl = cff * (aks_k - aks_km1);
    Gs1 = K_bl / (zbl * ws + eps);
    if (masking) Gs1 = Gs1 * mr;                                 // :777
    dGs1dS = fmin(0.0, -dK_bl / (ws + eps) - K_bl * f1);
  } else {
    ksbl = 0;
    const double b1 = 0.5 * (GF(bustr)[a] + GF(bustr)[a + 1]), b2 = 0.5 * (GF(bvstr)[a] + GF(bvstr)[a + ni]);
    double Ustarb = sqrt(sqrt(b1 * b1 + b2 * b2));
    if (masking) Ustarb = Ustarb * mr;                           // :793
    const double dK_bl = vonKar * Ustarb;
    const double K_bl = dK_bl * (hsbl - z_w[w3i(0)]);
    Gm1 = K_bl / (zbl * wm + eps);
    if (masking) Gm1 = Gm1 * mr;                                 // :799
    dGm1dS = fmin(0.0, -dK_bl / (wm + eps) - K_bl * f1);
    Gt1 = K_bl / (zbl * ws + eps);
    if (masking) Gt1 = Gt1 * mr;                                 // :808
    dGt1dS = fmin(0.0, -dK_bl / (ws + eps) - K_bl * f1);
    Gs1 = Gt1;
    dGs1dS = dGt1dS;
  }
  // ---------- upward again, LMD_SEG levels at a time: boundary-layer profiles, convective adjustment, final values ----------
  for (int kb = 1; kb <= N - 1; kb += LMD_SEG) {
    double nuA[LMD_SEG], zwA[LMD_SEG], bvA[LMD_SEG];
#pragma unroll
    for (int t = 0; t < LMD_SEG; t++) {
      const long q3 = w3i(min(kb + t, N - 1));
      nuA[t] = nu_col[(min(kb + t, N - 1) - 1) * NTH]; zwA[t] = z_w[q3]; bvA[t] = bvf[q3];
    }
#pragma unroll
    for (int t = 0; t < LMD_SEG; t++) {
      const int k = kb + t;
      if (k <= N - 1) {
        const double zwk = zwA[t], bvk = bvA[t];
        double akv = akv_a(nuA[t], bvk), akt = akt_a(nuA[t], bvk), aks = akt;
        if (k > ksbl) {
          const double depth = zwN - zwk;
          double gT, gS;
          const double bf = bflux_z(zwk, gT, gS);
          double sigma = (bf < 0.0) ? fmin(sl_dpth, depth) : depth;
          double wmk, wsk;
          wscale(Ustar, sigma, bf, wmk, wsk);
          sigma = depth / (zbl + eps);
          if (masking) sigma = sigma * mr;                         // :866
          const double a1 = sigma - 2.0, a2 = 3.0 - 2.0 * sigma, a3 = sigma - 1.0;
          const double Gm = a1 + a2 * Gm1 + a3 * dGm1dS;
          const double Gt = a1 + a2 * Gt1 + a3 * dGt1dS;
          const double Gs = a1 + a2 * Gs1 + a3 * dGs1dS;
          akv = depth * wmk * (1.0 + sigma * Gm);
          akt = depth * wsk * (1.0 + sigma * Gt);
          aks = depth * wsk * (1.0 + sigma * Gs);
          const double cff = lmd_Cg * (1.0 - (0.5 + copysign(0.5, bf))) / (zbl * wsk + eps);
          ghT[w3i(k)] = cff * gT;
          ghS[w3i(k)] = cff * gS;
        } else {
          ghT[w3i(k)] = 0.0;
          ghS[w3i(k)] = 0.0;
        }
        // lmd_finish_tile: convective mixing where the stratification is unstable (lmd_vmix.F:520-540)
        double cff = fmax(bvk, lmd_bvfcon);
        cff = fmin(1.0, (lmd_bvfcon - cff) / lmd_bvfcon);
        double nu_sxc = 1.0 - cff * cff;
        nu_sxc = nu_sxc * nu_sxc * nu_sxc;
        Akv[w3i(k)] = akv + lmd_nu0c * nu_sxc;
        AkT[w3i(k)] = akt + lmd_nu0c * nu_sxc;
        AkS[w3i(k)] = aks + lmd_nu0c * nu_sxc;
      }
    }
  }
}

// lmd_finish_tile boundary values exactly as written at lmd_vmix.F:545-640: W/E columns (note Iend-1 on
// the eastern edge, and no periodicity guard), then S/N rows, then the four corners.
__global__ void k_lmd_edges(const RomsDev *__restrict__ c, int phase)
{
  DEV_PROLOGUE(c)
  const int NAT = b.NAT;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;                                      // 0..N
  const gd_t Akv = (gd_t)(c->F.Akv + (long)k * nij);
  auto copy = [&](long dst, long src) {
    for (int it = 0; it < NAT; it++) GF(Akt)[dst + (long)k * nij + (long)it * n3w] = GF(Akt)[src + (long)k * nij + (long)it * n3w];
    Akv[dst] = Akv[src];
  };
  auto corner = [&](long dst, long s1, long s2) {
    for (int it = 0; it < NAT; it++) {
      const gd_t A = (gd_t)(c->F.Akt + (long)k * nij + (long)it * n3w);
      A[dst] = 0.5 * (A[s1] + A[s2]);
    }
    Akv[dst] = 0.5 * (Akv[s1] + Akv[s2]);
  };
  if (phase == 0) {
    const int j = b.Jstr + q;
    if (j > b.Jend) return;
    if (b.west_edge) copy(I2(b.Istr - 1, j), I2(b.Istr, j));
    if (b.east_edge) copy(I2(b.Iend - 1, j), I2(b.Iend, j));
  } else if (phase == 1) {
    const int i = b.Istr + q;
    if (i > b.Iend) return;
    if (b.south_edge) copy(I2(i, b.Jstr - 1), I2(i, b.Jstr));
    if (b.north_edge) copy(I2(i, b.Jend + 1), I2(i, b.Jend));
  } else if (q == 0) {
    if (b.south_edge && b.west_edge) corner(I2(b.Istr - 1, b.Jstr - 1), I2(b.Istr, b.Jstr - 1), I2(b.Istr - 1, b.Jstr));
    if (b.south_edge && b.east_edge) corner(I2(b.Iend + 1, b.Jstr - 1), I2(b.Iend, b.Jstr - 1), I2(b.Iend + 1, b.Jstr));
    if (b.north_edge && b.west_edge) corner(I2(b.Istr - 1, b.Jend + 1), I2(b.Istr, b.Jend + 1), I2(b.Istr - 1, b.Jend));
    if (b.north_edge && b.east_edge) corner(I2(b.Iend + 1, b.Jend + 1), I2(b.Iend, b.Jend + 1), I2(b.Iend + 1, b.Jend));
  }
}

}  // namespace

extern "C" int roms_hip_lmd_vmix(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_lmd_vmix");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  const roms_bounds_t &b = g_ctx.b;
  if (b.NAT < 2 || !g_ctx.p.salinity) return roms_fail("roms_hip_lmd_vmix", "built for the SALINITY set-up (NAT = 2)");
  const long nij = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1);
  const long n3w = nij * (b.N + 1);
  {
    ScopedTimer tm("lmd_vmix");
    // run-time constants of mod_scalars.F:4330 (lmd_Cg) and lmd_skpp.F:300 (Vtc), evaluated on the host
    const double vonKar = 0.41, lmd_Cstar = 10.0, lmd_Cv = 1.25, lmd_Ric = 0.3, lmd_betaT = -0.2, lmd_cs = 98.96,
                 lmd_epsilon = 0.1;
    const double lmd_Cg = lmd_Cstar * vonKar * pow(lmd_cs * vonKar * lmd_epsilon, 1.0 / 3.0);
    const double Vtc = lmd_Cv * sqrt(-lmd_betaT) / (sqrt(lmd_cs * lmd_epsilon) * lmd_Ric * vonKar * vonKar);
    const dim3 grid = grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1);
    if (b.N <= 32) hipLaunchKernelGGL(k_lmd_vmix<32>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, lmd_Cg, Vtc);
    else if (b.N <= ROMS_MAXN) hipLaunchKernelGGL(k_lmd_vmix<ROMS_MAXN>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nstp, lmd_Cg, Vtc);
    else return roms_fail("roms_hip_lmd_vmix", "N > 64 not instantiated");
    KERNEL_CHECK("k_lmd_vmix");
    const int nj = b.Jend - b.Jstr + 1, ni_ = b.Iend - b.Istr + 1;
    if (b.west_edge || b.east_edge)
      hipLaunchKernelGGL(k_lmd_edges, dim3((nj + 63) / 64, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, 0);
    if (b.south_edge || b.north_edge)
      hipLaunchKernelGGL(k_lmd_edges, dim3((ni_ + 63) / 64, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, 1);
    if ((b.south_edge || b.north_edge) && (b.west_edge || b.east_edge))
      hipLaunchKernelGGL(k_lmd_edges, dim3(1, b.N + 1), dim3(64), 0, g_ctx.stream, g_ctx.devc, 2);
    KERNEL_CHECK("k_lmd_edges");
  }
  // bc_r2d_tile(hsbl) (the kernel wrote the wall rows of a channel; western / eastern edges and corners through the
  // generic edge kernel) + exchange; bc_w3d_tile(Akv), bc_w3d_tile(Akt(:,:,:,itrc)): wall rows + exchange
  if (!b.EWperiodic && (rc = bc_generic(LBV_ZETA, LBV_ZETA, g_ctx.dev[FID_hsbl], 1))) return rc;
  if ((rc = halo_exchange2d(GT_R, g_ctx.dev[FID_hsbl]))) return rc;
  if ((rc = bc_w3d(g_ctx.dev[FID_Akv]))) return rc;
  for (int it = 0; it < b.NAT; it++)
    if ((rc = bc_w3d(g_ctx.dev[FID_Akt] + (long)it * n3w))) return rc;
  return 0;
}
