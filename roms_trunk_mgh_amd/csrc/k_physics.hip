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
// Column-local: one thread per (i,j), two sweeps.  Upward: the forward recurrences of the three splines
// (u, v, pden) go to the 3-D device scratch ([level][i,j], coalesced).  Downward: their back-substitution
// is fused with the Richardson-number mixing, the bulk Richardson function and the boundary-layer depth
// search, so no other profile is stored; the buoyancy-flux profile is recomputed (two exp) where used.
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

struct LmdScratch { double *FC, *dU, *dV, *dR, *Bf; };

__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_lmd_vmix(const RomsDev *__restrict__ c, int nstp, LmdScratch w, double lmd_Cg, double Vtc)
{
  DEV_PROLOGUE(c)
  const roms_params_t &p = c->p;
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const long a = I2(i, j);
  const double g = p.g, vonKar = 0.41;
  const double lmd_Ri0 = 0.7, lmd_bvfcon = -2.0E-5, lmd_nu0c = 0.01, lmd_nu0m = 10.0E-4, lmd_nu0s = 10.0E-4;
  const double lmd_Ric = 0.3, lmd_cekman = 0.7, lmd_cmonob = 1.0, lmd_epsilon = 0.1;
  const double gorho0 = g / p.rho0;
  const gcd_t Hz = (gcd_t)c->F.Hz, rho = (gcd_t)c->F.rho, pden = (gcd_t)c->F.pden, bvf = (gcd_t)c->F.bvf;
  const gcd_t z_w = (gcd_t)c->F.z_w;
  const gcd_t u = (gcd_t)(c->F.u + (long)(nstp - 1) * n3r), v = (gcd_t)(c->F.v + (long)(nstp - 1) * n3r);
  const gd_t Akv = (gd_t)c->F.Akv, AkT = (gd_t)c->F.Akt, AkS = (gd_t)(c->F.Akt + n3w);
  const gd_t ghT = (gd_t)c->F.ghats, ghS = (gd_t)(c->F.ghats + n3w);
  const gd_t FC = (gd_t)w.FC, dU = (gd_t)w.dU, dV = (gd_t)w.dV, dR = (gd_t)w.dR;
  auto r3i = [&](int k) { return a + (long)(k - 1) * nij; };     // rho-type level k = 1..N
  auto w3i = [&](int k) { return a + (long)k * nij; };           // W-type level k = 0..N

  // ---------- one upward sweep: spline recurrences of u, v (lmd_vmix_tile :205-225 = lmd_skpp_tile
  // :345-365) and of pden; only these forward values go to scratch ----------
  // (software-pipelined: the six loads of level k+2 are issued before level k is computed)
  {
    double FCm = 0.0, dUm = 0.0, dVm = 0.0, dRm = 0.0;
    long q = r3i(1);
    double hz = Hz[q], ua = u[q], ub = u[q + 1], va = v[q], vb = v[q + ni], pd = pden[q];
    q = r3i(N >= 2 ? 2 : 1);
    double hz1 = Hz[q], ua1 = u[q], ub1 = u[q + 1], va1 = v[q], vb1 = v[q + ni], pd1 = pden[q];
    for (int k = 1; k <= N - 1; k++) {
      q = r3i(k + 2 <= N ? k + 2 : N);
      const double hz2 = Hz[q], ua2 = u[q], ub2 = u[q + 1], va2 = v[q], vb2 = v[q + ni], pd2 = pden[q];
      const double cff = 1.0 / (2.0 * hz1 + hz * (2.0 - FCm));
      const double fck = cff * hz1;
      const double duk = cff * (3.0 * (ua1 - ua + ub1 - ub) - hz * dUm);
      const double dvk = cff * (3.0 * (va1 - va + vb1 - vb) - hz * dVm);
      const double drk = cff * (6.0 * (pd1 - pd) - hz * dRm);
      FC[w3i(k)] = fck; dU[w3i(k)] = duk; dV[w3i(k)] = dvk; dR[w3i(k)] = drk;
      FCm = fck; dUm = duk; dVm = dvk; dRm = drk;
      hz = hz1; ua = ua1; ub = ub1; va = va1; vb = vb1; pd = pd1;
      hz1 = hz2; ua1 = ua2; ub1 = ub2; va1 = va2; vb1 = vb2; pd1 = pd2;
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

  // ---------- one downward sweep: back-substitution of the three splines fused with (a) the shear /
  // Richardson-number mixing of lmd_vmix_tile (:240-300) and (b) the bulk Richardson function and the
  // boundary-layer depth search of lmd_skpp_tile (:380-470) ----------
  const double cff1 = 1.0 / 3.0, cff2 = 1.0 / 6.0;
  int ksbl = 1;
  {
    // inputs of iteration k: rho-type level k (hz, pd, ua/ub = u(i), u(i+1), va/vb = v(j), v(j+1)) and
    // W-type level k-1 (fc, dr, du, dv: forward spline values; bvm, zwm); software-pipelined, the twelve
    // loads of iteration k-1 are issued before iteration k is computed
    long q = r3i(N), qw = w3i(N > 1 ? N - 1 : 1), qz = w3i(N - 1);
    double hz = Hz[q], pd = pden[q], ua = u[q], ub = u[q + 1], va = v[q], vb = v[q + ni];
    double fc = FC[qw], dr = dR[qw], du = dU[qw], dv = dV[qw], bvm = bvf[qz], zwm = z_w[qz];
    double bvk = 0.0, zwk = zwN;                                 // W-type level k (bvf(N) is not used)
    const double Rref = pd + hz * (cff1 * 0.0 + cff2 * dr);      // final: x(N) = 0
    const double Uref = 0.5 * (ua + ub) + hz * (cff1 * 0.0 + cff2 * du);
    const double Vref = 0.5 * (va + vb) + hz * (cff1 * 0.0 + cff2 * dv);
    double dRk = 0.0, dUk = 0.0, dVk = 0.0;                      // final values at level k (k = N: 0)
    double FCk = 0.0;                                            // FC(i,N) = 0
    hsbl = z_w[w3i(1)];
    for (int k = N; k >= 1; k--) {
      q = r3i(k > 1 ? k - 1 : 1); qw = w3i(k > 2 ? k - 2 : 1); qz = w3i(k > 2 ? k - 2 : 0);
      const double n_hz = Hz[q], n_pd = pden[q], n_ua = u[q], n_ub = u[q + 1], n_va = v[q], n_vb = v[q + ni];
      const double n_fc = FC[qw], n_dr = dR[qw], n_du = dU[qw], n_dv = dV[qw], n_bvm = bvf[qz], n_zwm = z_w[qz];
      // final spline derivatives at level k-1
      double dRm = 0.0, dUm = 0.0, dVm = 0.0;
      if (k - 1 >= 1) {
        dRm = dr - fc * dRk;
        dUm = du - fc * dUk;
        dVm = dv - fc * dVk;
      }
      // (a) interior mixing at W-level k
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
        cff = 1.0 / sqrt(fmax(bv, 1.0E-7));
        const double lmd_iwm = 1.0E-6 * cff, lmd_iws = 1.0E-7 * cff;
        Akv[w3i(k)] = lmd_iwm + lmd_nu0m * nu_sx;
        // Akt(itemp) = Akt(isalt) here (lmd_vmix.F:296-297): one store; the sweeps below read AkT for both and the
        // last one writes the final AkS of every interior level
        AkT[w3i(k)] = lmd_iws + lmd_nu0s * nu_sx;
      }
      // (b) bulk Richardson function at W-level k-1
      const double depth = zwN - zwm;
      double gT, gS;
      const double bf = bflux_z(zwm, gT, gS);
      if (k - 1 == 0) { ghT[w3i(0)] = gT; ghS[w3i(0)] = gS; }
      const double sigma = (bf < 0.0) ? fmin(sl_dpth, depth) : depth;
      double wmk, wsk;
      wscale(Ustar, sigma, bf, wmk, wsk);
      const double Rk = pd - hz * (cff1 * dRm + cff2 * dRk);
      const double Uk = 0.5 * (ua + ub) - hz * (cff1 * dUm + cff2 * dUk);
      const double Vk = 0.5 * (va + vb) - hz * (cff1 * dVm + cff2 * dVk);
      const double Ritop = -gorho0 * (Rref - Rk) * depth;
      const double Ribot = (Uref - Uk) * (Uref - Uk) + (Vref - Vk) * (Vref - Vk) +
                           Vtc * depth * wsk * sqrt(fabs(bvm));
      const double FCkm1 = Ritop - lmd_Ric * Ribot;
      // boundary-layer depth: first level (from the top, k = N..2) where the function turns positive
      if (k >= 2 && ksbl == 1 && FCkm1 > 0.0) {
        hsbl = (zwk * FCkm1 - zwm * FCk) / (FCkm1 - FCk);
        ksbl = k;
      }
      FCk = FCkm1;
      dRk = dRm; dUk = dUm; dVk = dVm;
      bvk = bvm; zwk = zwm;
      hz = n_hz; pd = n_pd; ua = n_ua; ub = n_ub; va = n_va; vb = n_vb;
      fc = n_fc; dr = n_dr; du = n_du; dv = n_dv; bvm = n_bvm; zwm = n_zwm;
    }
  }
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
  ksbl = 1;
  for (int k = N; k >= 2; k--)
    if ((ksbl == 1) && (z_w[w3i(k - 1)] < hsbl)) ksbl = k;
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
    double K_bl = cff_dn * Akv[w3i(k)] + cff_up * Akv[w3i(k - 1)];
    double dK_bl = cff * (Akv[w3i(k)] - Akv[w3i(k - 1)]);
    Gm1 = K_bl / (zbl * wm + eps);
    if (masking) Gm1 = Gm1 * mr;                                 // :754
    dGm1dS = fmin(0.0, -dK_bl / (wm + eps) - K_bl * f1);
    K_bl = cff_dn * AkT[w3i(k)] + cff_up * AkT[w3i(k - 1)];
    dK_bl = cff * (AkT[w3i(k)] - AkT[w3i(k - 1)]);
    Gt1 = K_bl / (zbl * ws + eps);
    if (masking) Gt1 = Gt1 * mr;                                 // :765
    dGt1dS = fmin(0.0, -dK_bl / (ws + eps) - K_bl * f1);
    // salinity: interior levels hold the temperature values (see above); level N was not touched
    const double aks_k = (k <= N - 1) ? AkT[w3i(k)] : AkS[w3i(k)], aks_km1 = AkT[w3i(k - 1)];
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
  long q3 = w3i(1);
  double n_akv = Akv[q3], n_akt = AkT[q3], n_zw = z_w[q3], n_bv = bvf[q3];
  for (int k = 1; k <= N - 1; k++) {
    double akv = n_akv, akt = n_akt, aks = n_akt;
    const double zwk = n_zw, bvk = n_bv;
    q3 = w3i(k + 1 <= N - 1 ? k + 1 : k);                      // next level, in flight during this one
    n_akv = Akv[q3]; n_akt = AkT[q3]; n_zw = z_w[q3]; n_bv = bvf[q3];
    if (k > ksbl) {
      const double depth = zwN - zwk;
      double gT, gS;
      const double bf = bflux_z(zwk, gT, gS);
      double sigma = (bf < 0.0) ? fmin(sl_dpth, depth) : depth;
      double wmk, wsk;
      wscale(Ustar, sigma, bf, wmk, wsk);
      sigma = depth / (zbl + eps);
      if (masking) sigma = sigma * mr;                           // :866
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
    LmdScratch w{g_ctx.hostc.ws3[1], g_ctx.hostc.ws3[2], g_ctx.hostc.ws3[3], g_ctx.hostc.ws3[4], g_ctx.hostc.ws3[5]};
    // run-time constants of mod_scalars.F:4330 (lmd_Cg) and lmd_skpp.F:300 (Vtc), evaluated on the host
    const double vonKar = 0.41, lmd_Cstar = 10.0, lmd_Cv = 1.25, lmd_Ric = 0.3, lmd_betaT = -0.2, lmd_cs = 98.96,
                 lmd_epsilon = 0.1;
    const double lmd_Cg = lmd_Cstar * vonKar * pow(lmd_cs * vonKar * lmd_epsilon, 1.0 / 3.0);
    const double Vtc = lmd_Cv * sqrt(-lmd_betaT) / (sqrt(lmd_cs * lmd_epsilon) * lmd_Ric * vonKar * vonKar);
    hipLaunchKernelGGL(k_lmd_vmix, grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1), block2d(), 0, g_ctx.stream, g_ctx.devc,
                       s->nstp, w, lmd_Cg, Vtc);
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
