/*
 * oracle_physics.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Per-step physics between the hot kernels:
 *   set_vbc_tile    ROMS/Nonlinear/set_vbc.F:104   (SOLVE3D; UV_QDRAG or UV_LDRAG; SALINITY, no EMINUSP)
 *   bulk_flux_tile  ROMS/Nonlinear/bulk_flux.F:146 (COARE 3.0; LONGWAVE = Berliand formula; no COOL_SKIN,
 *                                                   no EMINUSP, no WIND_MINUS_CURRENT; MASKING: the
 *                                                   multiplies of bulk_flux.F:486-920)
 * Both reference files compile stand-alone and are compared with this restatement in
 * tests/test_ref_pinning.py (bulk_flux to a few ulp: LOG/EXP/ATAN/pow come from different math
 * libraries).
 */
#include "oracle.h"

int oracle_set_vbc(OARGS)
{
  ORACLE_PROLOGUE
  if (o_check_lbc(b, p)) return 8;
  const int nrhs = s->nrhs, itemp = 1, isalt = 2;
  /* set_vbc.F:262-268 */
  for (int j = JstrR; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) {
      stflx(i, j, itemp) = stflux(i, j, itemp);
      btflx(i, j, itemp) = btflux(i, j, itemp);
    }
  /* SALINITY without EMINUSP, :292-312: kinematic salt flux = (E-P) * S */
  if (p->salinity && NT >= 2)
    for (int j = JstrR; j <= JendR; j++)
      for (int i = IstrR; i <= IendR; i++) {
        const double EmP = stflux(i, j, isalt);
        stflx(i, j, isalt) = EmP * t(i, j, N, nrhs, isalt);
        btflx(i, j, isalt) = btflx(i, j, isalt) * t(i, j, 1, nrhs, isalt);
      }
  /* bottom stress, :380-470 */
  if (p->uv_drag == 2) {            /* UV_QDRAG */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        const double cff1 = 0.25 * (v(i, j, 1, nrhs) + v(i, j + 1, 1, nrhs) + v(i - 1, j, 1, nrhs) + v(i - 1, j + 1, 1, nrhs));
        const double cff2 = sqrt(u(i, j, 1, nrhs) * u(i, j, 1, nrhs) + cff1 * cff1);
        F->bustr[I2(i, j)] = 0.5 * (rdrag2(i - 1, j) + rdrag2(i, j)) * u(i, j, 1, nrhs) * cff2;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        const double cff1 = 0.25 * (u(i, j, 1, nrhs) + u(i + 1, j, 1, nrhs) + u(i, j - 1, 1, nrhs) + u(i + 1, j - 1, 1, nrhs));
        const double cff2 = sqrt(cff1 * cff1 + v(i, j, 1, nrhs) * v(i, j, 1, nrhs));
        F->bvstr[I2(i, j)] = 0.5 * (rdrag2(i, j - 1) + rdrag2(i, j)) * v(i, j, 1, nrhs) * cff2;
      }
  } else if (p->uv_drag == 1) {     /* UV_LDRAG */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++)
        F->bustr[I2(i, j)] = 0.5 * (rdrag(i - 1, j) + rdrag(i, j)) * u(i, j, 1, nrhs);
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++)
        F->bvstr[I2(i, j)] = 0.5 * (rdrag(i, j - 1) + rdrag(i, j)) * v(i, j, 1, nrhs);
  } else if (p->uv_drag == 3) {     /* UV_LOGDRAG, :542-580: Cd = (vonKar / LOG(dz/ZoBot))^2 within [Cdb_min, Cdb_max] */
    const double vonKar = 0.41;      /* mod_scalars.F:444 */
    double *wrk_ = walloc(nis * njs);
#define wrk(i,j) wrk_[WS2(i,j)]
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        const double cff1 = 1.0 / log((z_r(i, j, 1) - z_w(i, j, 0)) / F->ZoBot[I2(i, j)]);
        const double cff2 = vonKar * vonKar * cff1 * cff1;
        wrk(i, j) = MIN(p->Cdb_max, MAX(p->Cdb_min, cff2));
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        const double cff1 = 0.25 * (v(i, j, 1, nrhs) + v(i, j + 1, 1, nrhs) + v(i - 1, j, 1, nrhs) + v(i - 1, j + 1, 1, nrhs));
        const double cff2 = sqrt(u(i, j, 1, nrhs) * u(i, j, 1, nrhs) + cff1 * cff1);
        F->bustr[I2(i, j)] = 0.5 * (wrk(i - 1, j) + wrk(i, j)) * u(i, j, 1, nrhs) * cff2;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        const double cff1 = 0.25 * (u(i, j, 1, nrhs) + u(i + 1, j, 1, nrhs) + u(i, j - 1, 1, nrhs) + u(i + 1, j - 1, 1, nrhs));
        const double cff2 = sqrt(cff1 * cff1 + v(i, j, 1, nrhs) * v(i, j, 1, nrhs));
        F->bvstr[I2(i, j)] = 0.5 * (wrk(i, j - 1) + wrk(i, j)) * v(i, j, 1, nrhs) * cff2;
      }
    free(wrk_);
#undef wrk
  } else return 8;
  if (p->limit_bstress) {           /* LIMIT_BSTRESS, :533-540 and the same four lines after each law (:562-567 ...) */
    const double cff = 0.75 / p->dt;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        const double cff3 = cff * 0.5 * (Hz(i - 1, j, 1) + Hz(i, j, 1));
        const double bs = F->bustr[I2(i, j)];
        F->bustr[I2(i, j)] = copysign(1.0, bs) * MIN(fabs(bs), fabs(u(i, j, 1, nrhs)) * cff3);
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        const double cff3 = cff * 0.5 * (Hz(i, j - 1, 1) + Hz(i, j, 1));
        const double bs = F->bvstr[I2(i, j)];
        F->bvstr[I2(i, j)] = copysign(1.0, bs) * MIN(fabs(bs), fabs(v(i, j, 1, nrhs)) * cff3);
      }
  }
  /* boundary conditions + periodic / tile exchange, :472-500 */
  /* bc_u2d_tile / bc_v2d_tile with isBu2d = isUbar, isBv2d = isVbar (bc_2d.F:184, :386; mod_ncparam.F:1229) */
  o_bc_generic(b, p, F, GT_U, LBV_UBAR, F->bustr, 1);
  o_bc_generic(b, p, F, GT_V, LBV_VBAR, F->bvstr, 1);
  o_exchange2d(b, GT_U, F->bustr);
  o_exchange2d(b, GT_V, F->bvstr);
  return 0;
}

/* stability functions of Fairall et al., bulk_flux.F:1020-1108 */
static double bulk_psiu(double ZoL, double pi)
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
  const double cff = MIN(50.0, 0.35 * ZoL);
  return -((1.0 + ZoL) + 0.6667 * (ZoL - 14.28) / exp(cff) + 8.525);
}
static double bulk_psit(double ZoL, double pi)
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
  const double cff = MIN(50.0, 0.35 * ZoL);
  return -(pow(1.0 + 2.0 * ZoL, 1.5) + 0.6667 * (ZoL - 14.28) / exp(cff) + 8.525);
}

int oracle_bulk_flux(OARGS)
{
  ORACLE_PROLOGUE
  if (o_check_lbc(b, p)) return 8;
  const int nrhs = s->nrhs, itemp = 1, isalt = 2, IterMax = 3, mk = p->masking;
  /* mod_scalars.F:431-444, :1415-1421 */
  const double Cp = 3985.0, StefBo = 5.67E-8, emmiss = 0.97, rhow = 1000.0, vonKar = 0.41;
  const double blk_Cpa = 1004.67, blk_Cpw = 4000.0, blk_Rgas = 287.1, blk_Zabl = 600.0, blk_beta = 1.2;
  const double pi = 3.14159265358979323846;
  const double g = p->g, rho0 = p->rho0;
  const double blk_ZQ = p->blk_ZQ, blk_ZT = p->blk_ZT, blk_ZW = p->blk_ZW;
  const double eps = 1.0E-20, r3 = 1.0 / 3.0;
  (void)rhow;
  double *Taux_ = walloc(nis * njs), *Tauy_ = walloc(nis * njs), *LHeat_ = walloc(nis * njs);
  double *Hlv_ = walloc(nis * njs);                                    /* (the reference keeps Hlv(i,j) too, :175) */
  double *SHeat_ = walloc(nis * njs), *LRad_ = walloc(nis * njs);
#define Taux(i,j)  Taux_[WS2(i,j)]
#define Tauy(i,j)  Tauy_[WS2(i,j)]
#define LHeat(i,j) LHeat_[WS2(i,j)]
#define SHeat(i,j) SHeat_[WS2(i,j)]
#define LRad(i,j)  LRad_[WS2(i,j)]
  double Hscale = rho0 * Cp;
  for (int j = Jstr - 1; j <= JendR; j++)
    for (int i = Istr - 1; i <= IendR; i++) {
      const double Ua = Uwind(i, j), Va = Vwind(i, j);
      const double Wmag = sqrt(Ua * Ua + Va * Va);
      const double PairM = Pair(i, j);
      const double TairC = Tair(i, j), TairK = TairC + 273.16;
      const double TseaC = t(i, j, N, nrhs, itemp), TseaK = TseaC + 273.16;
      const double RH = Hair(i, j);
      const double delTc = 0.0, delQc = 0.0;
      /* LONGWAVE (Berliand), bulk_flux.F:440-462 */
      double cff = (0.7859 + 0.03477 * TairC) / (1.0 + 0.00412 * TairC);
      const double e_sat = pow(10.0, cff);
      const double vap_p = e_sat * RH;
      double cff2 = TairK * TairK * TairK;
      double cff1 = cff2 * TairK;
      LRad(i, j) = -emmiss * StefBo *
                   (cff1 * (0.39 - 0.05 * sqrt(vap_p)) * (1.0 - 0.6823 * cloud(i, j) * cloud(i, j)) +
                    cff2 * 4.0 * (TseaK - TairK));
      if (mk) LRad(i, j) = LRad(i, j) * rmask(i, j);                          /* MASKING, bulk_flux.F:486 */
      if (p->wet_dry) LRad(i, j) = LRad(i, j) * rmask_wet(i, j);           /* WET_DRY: the next block */
      /* specific humidities, :486-520 */
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
      Hlv_[WS2(i, j)] = Hlv;
      /* first guesses, :536-600 */
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
      /* iterate, :612-672 */
      for (int Iter = 1; Iter <= IterMax; Iter++) {
        ZoW = charn * Wstar * Wstar / g + 0.11 * VisAir / (Wstar + eps);
        const double Rr = ZoW * Wstar / VisAir;
        const double ZoQ = MIN(1.15e-4, 5.5e-5 / pow(Rr, 0.6));
        const double ZoT = ZoQ;
        const double ZoL = vonKar * g * blk_ZW * (Tstar * (1.0 + 0.61 * Q) + 0.61 * TairK * Qstar) /
                           (TairK * Wstar * Wstar * (1.0 + 0.61 * Q) + eps);
        const double L = blk_ZW / (ZoL + eps);
        const double Wpsi = bulk_psiu(ZoL, pi);
        const double Tpsi = bulk_psit(blk_ZT / L, pi);
        const double Qpsi = bulk_psit(blk_ZQ / L, pi);
        Wstar = MAX(eps, delW * vonKar / (log(blk_ZW / ZoW) - Wpsi));
        Tstar = -(delT - delTc) * vonKar / (log(blk_ZT / ZoT) - Tpsi);
        Qstar = -(delQ - delQc) * vonKar / (log(blk_ZQ / ZoQ) - Qpsi);
        const double Bf = -g / TairK * Wstar * (Tstar + 0.61 * TairK * Qstar);
        if (Bf > 0.0) Wgus = blk_beta * pow(Bf * blk_Zabl, r3);
        else Wgus = 0.2;
        delW = sqrt(Wmag * Wmag + Wgus * Wgus);
      }
      /* fluxes, :680-760 */
      const double Wspeed = sqrt(Wmag * Wmag + Wgus * Wgus);
      Cd = Wstar * Wstar / (Wspeed * Wspeed + eps);
      const double Hs = -blk_Cpa * rhoAir * Wstar * Tstar;
      const double diffw = 2.11E-5 * pow(TairK / 273.16, 1.94);
      const double diffh = 0.02411 * (1.0 + TairC * (3.309E-3 - 1.44E-6 * TairC)) / (rhoAir * blk_Cpa);
      cff = Qair * Hlv / (blk_Rgas * TairK * TairK);
      const double wet_bulb = 1.0 / (1.0 + 0.622 * (cff * Hlv * diffw) / (blk_Cpa * diffh));
      const double Hsr = rain(i, j) * wet_bulb * blk_Cpw * ((TseaC - TairC) + (Qsea - Q) * Hlv / blk_Cpa);
      SHeat(i, j) = (Hs + Hsr);
      if (mk) SHeat(i, j) = SHeat(i, j) * rmask(i, j);                        /* :790 */
      if (p->wet_dry) SHeat(i, j) = SHeat(i, j) * rmask_wet(i, j);           /* WET_DRY: the next block */
      const double Hl = -Hlv * rhoAir * Wstar * Qstar;
      const double upvel = -1.61 * Wstar * Qstar - (1.0 + 1.61 * Q) * Wstar * Tstar / TairK;
      const double Hlw = rhoAir * Hlv * upvel * Q;
      LHeat(i, j) = (Hl + Hlw);
      if (mk) LHeat(i, j) = LHeat(i, j) * rmask(i, j);                        /* :809 */
      if (p->wet_dry) LHeat(i, j) = LHeat(i, j) * rmask_wet(i, j);           /* WET_DRY: the next block */
      const double Taur = 0.85 * rain(i, j) * Wmag;
      cff = rhoAir * Cd * Wspeed;
      Taux(i, j) = (cff * Ua + Taur * copysign(1.0, Ua));
      if (mk) Taux(i, j) = Taux(i, j) * rmask(i, j);                          /* :824 */
      if (p->wet_dry) Taux(i, j) = Taux(i, j) * rmask_wet(i, j);           /* WET_DRY: the next block */
      Tauy(i, j) = (cff * Va + Taur * copysign(1.0, Va));
      if (mk) Tauy(i, j) = Tauy(i, j) * rmask(i, j);                          /* :831 */
      if (p->wet_dry) Tauy(i, j) = Tauy(i, j) * rmask_wet(i, j);           /* WET_DRY: the next block */
    }
  /* kinematic fluxes, :790-860 */
  Hscale = 1.0 / (rho0 * Cp);
  for (int j = JstrR; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) {
      lrflx(i, j) = LRad(i, j) * Hscale;
      lhflx(i, j) = -LHeat(i, j) * Hscale;
      shflx(i, j) = -SHeat(i, j) * Hscale;
      stflux(i, j, itemp) = (srflx(i, j) + lrflx(i, j) + lhflx(i, j) + shflx(i, j));
      if (mk) stflux(i, j, itemp) = stflux(i, j, itemp) * rmask(i, j);        /* :877 */
      if (p->wet_dry) stflux(i, j, itemp) = stflux(i, j, itemp) * rmask_wet(i, j);           /* WET_DRY: the next block */
      if (p->eminusp) {                                                       /* EMINUSP, :883-899 */
        const double cffw = 1.0 / rhow;
        F->evap[I2(i, j)] = LHeat(i, j) / Hlv_[WS2(i, j)];
        if (mk) F->evap[I2(i, j)] = F->evap[I2(i, j)] * rmask(i, j);
        if (p->wet_dry) F->evap[I2(i, j)] = F->evap[I2(i, j)] * rmask_wet(i, j);           /* WET_DRY: the next block */
        stflux(i, j, isalt) = cffw * (F->evap[I2(i, j)] - rain(i, j));
        if (mk) stflux(i, j, isalt) = stflux(i, j, isalt) * rmask(i, j);
        if (p->wet_dry) stflux(i, j, isalt) = stflux(i, j, isalt) * rmask_wet(i, j);           /* WET_DRY: the next block */
      }
    }
  const double cffs = 0.5 / rho0;
  for (int j = JstrR; j <= JendR; j++)
    for (int i = Istr; i <= IendR; i++) {
      F->sustr[I2(i, j)] = cffs * (Taux(i - 1, j) + Taux(i, j));
      if (mk) F->sustr[I2(i, j)] = F->sustr[I2(i, j)] * umask(i, j);          /* :908 */
      if (p->wet_dry) F->sustr[I2(i, j)] = F->sustr[I2(i, j)] * umask_wet(i, j);           /* WET_DRY: the next block */
    }
  for (int j = Jstr; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) {
      F->svstr[I2(i, j)] = cffs * (Tauy(i, j - 1) + Tauy(i, j));
      if (mk) F->svstr[I2(i, j)] = F->svstr[I2(i, j)] * vmask(i, j);          /* :919 */
      if (p->wet_dry) F->svstr[I2(i, j)] = F->svstr[I2(i, j)] * vmask_wet(i, j);           /* WET_DRY: the next block */
    }
  o_exchange2d(b, GT_R, F->lrflx);
  o_exchange2d(b, GT_R, F->lhflx);
  o_exchange2d(b, GT_R, F->shflx);
  o_exchange2d(b, GT_R, &stflux(LBi, LBj, itemp));
  if (p->eminusp) {                                                     /* bulk_flux.F:945-952 */
    o_exchange2d(b, GT_R, F->evap);
    o_exchange2d(b, GT_R, &stflux(LBi, LBj, isalt));
  }
  o_exchange2d(b, GT_U, F->sustr);
  o_exchange2d(b, GT_V, F->svstr);
  free(Taux_); free(Tauy_); free(LHeat_); free(SHeat_); free(LRad_); free(Hlv_);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * lmd_vmix = lmd_vmix_tile + lmd_skpp_tile + lmd_finish_tile: Large/McWilliams/Doney K-profile
 * vertical mixing (ROMS/Nonlinear/lmd_vmix.F:99/465, lmd_skpp.F:98, lmd_swfrac.F:6) with the
 * BENCHMARK option set: LMD_RIMIX + RI_SPLINES, LMD_CONVEC, LMD_SKPP, LMD_NONLOCAL, SALINITY;
 * no LMD_DDMIX, LMD_SHAPIRO, LMD_BKPP, WET_DRY; MASKING: the multiplies of lmd_skpp.F:272-866.  Uniform Jerlov water type.
 * --------------------------------------------------------------------------------------------- */
static double o_swfrac(const roms_params_t *p, double Zscale, double Z)   /* lmd_swfrac.F:60-75 */
{
  const double fac1 = Zscale / p->swfrac_mu1, fac2 = Zscale / p->swfrac_mu2, fac3 = p->swfrac_r1;
  return exp(Z * fac1) * fac3 + exp(Z * fac2) * (1.0 - fac3);
}

/* turbulent velocity scales wm, ws (lmd_skpp.F:430-455, and twice more below) */
static void o_wscale(double Ustar, double sigma, double Bf, double *wm, double *ws)
{
  const double vonKar = 0.41, small = 1.0E-20, r3 = 1.0 / 3.0;
  const double lmd_am = 1.257, lmd_as = -28.86, lmd_cm = 8.36, lmd_cs = 98.96, lmd_zetam = -0.2, lmd_zetas = -1.0;
  const double Ustar3 = Ustar * Ustar * Ustar;
  const double zetahat = vonKar * sigma * Bf;
  const double zetapar = zetahat / (Ustar3 + small);
  if (zetahat >= 0.0) {
    *wm = vonKar * Ustar / (1.0 + 5.0 * zetapar);
    *ws = *wm;
  } else {
    if (zetapar > lmd_zetam) *wm = vonKar * Ustar * pow(1.0 - 16.0 * zetapar, 0.25);
    else *wm = vonKar * pow(lmd_am * Ustar3 - lmd_cm * zetahat, r3);
    if (zetapar > lmd_zetas) *ws = vonKar * Ustar * pow(1.0 - 16.0 * zetapar, 0.5);
    else *ws = vonKar * pow(lmd_as * Ustar3 - lmd_cs * zetahat, r3);
  }
}

int oracle_lmd_vmix(OARGS)
{
  ORACLE_PROLOGUE
  if (o_check_lbc(b, p)) return 8;
  if (NAT < 2 || !p->salinity) return 8;          /* restated for the SALINITY set-up only */
  const int nstp = s->nstp, itemp = 1, isalt = 2, mk = p->masking;   /* MASKING: the multiplies of lmd_skpp.F:272-866 */
  const double g = p->g, vonKar = 0.41;
  /* mod_scalars.F:1552-1629 */
  const double lmd_Ri0 = 0.7, lmd_bvfcon = -2.0E-5, lmd_nu0c = 0.01, lmd_nu0m = 10.0E-4, lmd_nu0s = 10.0E-4;
  const double lmd_Cstar = 10.0, lmd_Cv = 1.25, lmd_Ric = 0.3, lmd_betaT = -0.2, lmd_cekman = 0.7, lmd_cmonob = 1.0;
  const double lmd_cs = 98.96, lmd_epsilon = 0.1;
  const double lmd_Cg = lmd_Cstar * vonKar * pow(lmd_cs * vonKar * lmd_epsilon, 1.0 / 3.0);   /* :4330 */
  const double gorho0 = g / p->rho0;                                                          /* :4176 */
  double *Rig_ = walloc(nis * njs * (N + 1)), *Bflux_ = walloc(nis * njs * (N + 1));
  double *FC_ = walloc(nis * (N + 1)), *dR_ = walloc(nis * (N + 1)), *dU_ = walloc(nis * (N + 1)), *dV_ = walloc(nis * (N + 1));
  double *Bo_ = walloc(nis * njs), *Bosol_ = walloc(nis * njs), *Bfsfc_ = walloc(nis * njs), *Ustar_ = walloc(nis * njs);
  double *sl_dpth_ = walloc(nis * njs), *wm_ = walloc(nis * njs), *ws_ = walloc(nis * njs), *f1_ = walloc(nis * njs);
  double *Gm1_ = walloc(nis * njs), *Gt1_ = walloc(nis * njs), *Gs1_ = walloc(nis * njs);
  double *dGm1dS_ = walloc(nis * njs), *dGt1dS_ = walloc(nis * njs), *dGs1dS_ = walloc(nis * njs);
  int *ksbl_ = (int *)calloc((size_t)(nis * njs), sizeof(int));
#define Rig(i,j,k)   Rig_[WS2(i,j) + (long)(k) * nis * njs]
#define Bflux(i,j,k) Bflux_[WS2(i,j) + (long)(k) * nis * njs]
#define FC(i,k) FC_[WSK(i,k)]
#define dR(i,k) dR_[WSK(i,k)]
#define dU(i,k) dU_[WSK(i,k)]
#define dV(i,k) dV_[WSK(i,k)]
#define Bo(i,j) Bo_[WS2(i,j)]
#define Bosol(i,j) Bosol_[WS2(i,j)]
#define Bfsfc(i,j) Bfsfc_[WS2(i,j)]
#define Ustar(i,j) Ustar_[WS2(i,j)]
#define sl_dpth(i,j) sl_dpth_[WS2(i,j)]
#define wm(i,j) wm_[WS2(i,j)]
#define ws(i,j) ws_[WS2(i,j)]
#define f1(i,j) f1_[WS2(i,j)]
#define Gm1(i,j) Gm1_[WS2(i,j)]
#define Gt1(i,j) Gt1_[WS2(i,j)]
#define Gs1(i,j) Gs1_[WS2(i,j)]
#define dGm1dS(i,j) dGm1dS_[WS2(i,j)]
#define dGt1dS(i,j) dGt1dS_[WS2(i,j)]
#define dGs1dS(i,j) dGs1dS_[WS2(i,j)]
#define ksbl(i,j) ksbl_[WS2(i,j)]
#define hsbl(i,j) F->hsbl[I2(i,j)]
#define sustr(i,j) F->sustr[I2(i,j)]
#define svstr(i,j) F->svstr[I2(i,j)]
#define bustr(i,j) F->bustr[I2(i,j)]
#define bvstr(i,j) F->bvstr[I2(i,j)]

  /* ================= lmd_vmix_tile, lmd_vmix.F:190-330 ================= */
  {
    const double eps = 1.0E-14;
    for (int j = MAX(1, Jstr - 1); j <= MIN(Jend + 1, Mm); j++) {
      const int ia = MAX(1, Istr - 1), ib = MIN(Iend + 1, Lm);
      for (int i = ia; i <= ib; i++) { FC(i, 0) = 0.0; dR(i, 0) = 0.0; dU(i, 0) = 0.0; dV(i, 0) = 0.0; }
      for (int k = 1; k <= N - 1; k++)
        for (int i = ia; i <= ib; i++) {
          const double cff = 1.0 / (2.0 * Hz(i, j, k + 1) + Hz(i, j, k) * (2.0 - FC(i, k - 1)));
          FC(i, k) = cff * Hz(i, j, k + 1);
          dR(i, k) = cff * (6.0 * (rho(i, j, k + 1) - rho(i, j, k)) - Hz(i, j, k) * dR(i, k - 1));
          dU(i, k) = cff * (3.0 * (u(i, j, k + 1, nstp) - u(i, j, k, nstp) + u(i + 1, j, k + 1, nstp) - u(i + 1, j, k, nstp)) -
                            Hz(i, j, k) * dU(i, k - 1));
          dV(i, k) = cff * (3.0 * (v(i, j, k + 1, nstp) - v(i, j, k, nstp) + v(i, j + 1, k + 1, nstp) - v(i, j + 1, k, nstp)) -
                            Hz(i, j, k) * dV(i, k - 1));
        }
      for (int i = ia; i <= ib; i++) { dR(i, N) = 0.0; dU(i, N) = 0.0; dV(i, N) = 0.0; }
      for (int k = N - 1; k >= 1; k--)
        for (int i = ia; i <= ib; i++) {
          dR(i, k) = dR(i, k) - FC(i, k) * dR(i, k + 1);
          dU(i, k) = dU(i, k) - FC(i, k) * dU(i, k + 1);
          dV(i, k) = dV(i, k) - FC(i, k) * dV(i, k + 1);
        }
      for (int k = 1; k <= N - 1; k++)
        for (int i = ia; i <= ib; i++) {
          const double shear2 = dU(i, k) * dU(i, k) + dV(i, k) * dV(i, k);
          Rig(i, j, k) = bvf(i, j, k) / (shear2 + eps);
        }
    }
    for (int k = 1; k <= N - 1; k++)
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          double cff = MIN(1.0, MAX(0.0, Rig(i, j, k)) / lmd_Ri0);
          double nu_sx = 1.0 - cff * cff;
          nu_sx = nu_sx * nu_sx * nu_sx;
          const double shear2 = bvf(i, j, k) / (Rig(i, j, k) + eps);
          cff = shear2 * shear2 / (shear2 * shear2 + 16.0E-10);
          nu_sx = cff * nu_sx;
          cff = 1.0 / sqrt(MAX(bvf(i, j, k), 1.0E-7));
          const double lmd_iwm = 1.0E-6 * cff, lmd_iws = 1.0E-7 * cff;
          Akv(i, j, k) = lmd_iwm + lmd_nu0m * nu_sx;
          Akt(i, j, k, itemp) = lmd_iws + lmd_nu0s * nu_sx;
          Akt(i, j, k, isalt) = Akt(i, j, k, itemp);
        }
  }

  /* ================= lmd_skpp_tile, lmd_skpp.F:300-930 ================= */
  {
    const double eps = 1.0E-10;
    const double Vtc = lmd_Cv * sqrt(-lmd_betaT) / (sqrt(lmd_cs * lmd_epsilon) * lmd_Ric * vonKar * vonKar);
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) sl_dpth(i, j) = lmd_epsilon * (z_w(i, j, N) - hsbl(i, j));
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        const double a1 = 0.5 * (sustr(i, j) + sustr(i + 1, j)), a2 = 0.5 * (svstr(i, j) + svstr(i, j + 1));
        Ustar(i, j) = sqrt(sqrt(a1 * a1 + a2 * a2));
        if (mk) Ustar(i, j) = Ustar(i, j) * rmask(i, j);                       /* :272 */
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        Bo(i, j) = g * (alpha(i, j) * (stflx(i, j, itemp) - srflx(i, j)) - beta(i, j) * stflx(i, j, isalt));
        Bosol(i, j) = g * alpha(i, j) * srflx(i, j);
      }
    for (int k = 0; k <= N; k++)
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          const double swdk = o_swfrac(p, -1.0, z_w(i, j, N) - z_w(i, j, k));
          Bflux(i, j, k) = (Bo(i, j) + Bosol(i, j) * (1.0 - swdk));
          if (mk) Bflux(i, j, k) = Bflux(i, j, k) * rmask(i, j);               /* :316 */
          const double cff = 1.0 - (0.5 + copysign(0.5, Bflux(i, j, k)));
          ghats(i, j, k, itemp) = -cff * (stflx(i, j, itemp) - srflx(i, j) + srflx(i, j) * (1.0 - swdk));
          ghats(i, j, k, isalt) = cff * stflx(i, j, isalt);
        }
    for (int j = Jstr; j <= Jend; j++) {
      for (int i = Istr; i <= Iend; i++) { FC(i, 0) = 0.0; dR(i, 0) = 0.0; dU(i, 0) = 0.0; dV(i, 0) = 0.0; }
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          const double cff = 1.0 / (2.0 * Hz(i, j, k + 1) + Hz(i, j, k) * (2.0 - FC(i, k - 1)));
          FC(i, k) = cff * Hz(i, j, k + 1);
          dR(i, k) = cff * (6.0 * (pden(i, j, k + 1) - pden(i, j, k)) - Hz(i, j, k) * dR(i, k - 1));
          dU(i, k) = cff * (3.0 * (u(i, j, k + 1, nstp) - u(i, j, k, nstp) + u(i + 1, j, k + 1, nstp) - u(i + 1, j, k, nstp)) -
                            Hz(i, j, k) * dU(i, k - 1));
          dV(i, k) = cff * (3.0 * (v(i, j, k + 1, nstp) - v(i, j, k, nstp) + v(i, j + 1, k + 1, nstp) - v(i, j + 1, k, nstp)) -
                            Hz(i, j, k) * dV(i, k - 1));
        }
      for (int i = Istr; i <= Iend; i++) { dR(i, N) = 0.0; dU(i, N) = 0.0; dV(i, N) = 0.0; }
      for (int k = N - 1; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) {
          dR(i, k) = dR(i, k) - FC(i, k) * dR(i, k + 1);
          dU(i, k) = dU(i, k) - FC(i, k) * dU(i, k + 1);
          dV(i, k) = dV(i, k) - FC(i, k) * dV(i, k + 1);
        }
      const double cff1 = 1.0 / 3.0, cff2 = 1.0 / 6.0;
      for (int i = Istr; i <= Iend; i++) {
        const double Rref = pden(i, j, N) + Hz(i, j, N) * (cff1 * dR(i, N) + cff2 * dR(i, N - 1));
        const double Uref = 0.5 * (u(i, j, N, nstp) + u(i + 1, j, N, nstp)) + Hz(i, j, N) * (cff1 * dU(i, N) + cff2 * dU(i, N - 1));
        const double Vref = 0.5 * (v(i, j, N, nstp) + v(i, j + 1, N, nstp)) + Hz(i, j, N) * (cff1 * dV(i, N) + cff2 * dV(i, N - 1));
        FC(i, N) = 0.0;
        for (int k = N; k >= 1; k--) {
          const double depth = z_w(i, j, N) - z_w(i, j, k - 1);
          double sigma;
          if (Bflux(i, j, k - 1) < 0.0) sigma = MIN(sl_dpth(i, j), depth);
          else sigma = depth;
          double wmk, wsk;
          o_wscale(Ustar(i, j), sigma, Bflux(i, j, k - 1), &wmk, &wsk);
          wm(i, j) = wmk; ws(i, j) = wsk;
          const double Rk = pden(i, j, k) - Hz(i, j, k) * (cff1 * dR(i, k - 1) + cff2 * dR(i, k));
          const double Uk = 0.5 * (u(i, j, k, nstp) + u(i + 1, j, k, nstp)) - Hz(i, j, k) * (cff1 * dU(i, k - 1) + cff2 * dU(i, k));
          const double Vk = 0.5 * (v(i, j, k, nstp) + v(i, j + 1, k, nstp)) - Hz(i, j, k) * (cff1 * dV(i, k - 1) + cff2 * dV(i, k));
          const double Ritop = -gorho0 * (Rref - Rk) * depth;
          const double Ribot = (Uref - Uk) * (Uref - Uk) + (Vref - Vk) * (Vref - Vk) +
                               Vtc * depth * wsk * sqrt(fabs(bvf(i, j, k - 1)));
          FC(i, k - 1) = Ritop - lmd_Ric * Ribot;
        }
      }
      for (int i = Istr; i <= Iend; i++) { ksbl(i, j) = 1; hsbl(i, j) = z_w(i, j, 1); }
      for (int k = N; k >= 2; k--)
        for (int i = Istr; i <= Iend; i++)
          if ((ksbl(i, j) == 1) && (FC(i, k - 1) > 0.0)) {
            hsbl(i, j) = (z_w(i, j, k) * FC(i, k - 1) - z_w(i, j, k - 1) * FC(i, k)) / (FC(i, k - 1) - FC(i, k));
            ksbl(i, j) = k;
          }
    }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        double zgrid = z_w(i, j, N) - hsbl(i, j);
        if (mk) zgrid = zgrid * rmask(i, j);                                   /* :562 */
        const double swdk = o_swfrac(p, -1.0, zgrid);
        Bfsfc(i, j) = (Bo(i, j) + Bosol(i, j) * (1.0 - swdk));
        if (mk) Bfsfc(i, j) = Bfsfc(i, j) * rmask(i, j);                       /* :574 */
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        if ((Ustar(i, j) > 0.0) && (Bfsfc(i, j) > 0.0)) {
          const double hekman = lmd_cekman * Ustar(i, j) / MAX(fabs(f(i, j)), eps);
          const double hmonob = lmd_cmonob * Ustar(i, j) * Ustar(i, j) * Ustar(i, j) / MAX(vonKar * Bfsfc(i, j), eps);
          hsbl(i, j) = (z_w(i, j, N) - MIN(MIN(hekman, hmonob), z_w(i, j, N) - hsbl(i, j)));
        }
        hsbl(i, j) = MIN(hsbl(i, j), z_w(i, j, N));
        hsbl(i, j) = MAX(hsbl(i, j), z_w(i, j, 0));
        if (mk) hsbl(i, j) = hsbl(i, j) * rmask(i, j);                         /* :595 */
      }
    /* bc_r2d_tile (closed walls: zero gradient) + periodic / tile exchange, :640-652 */
    o_bc_generic(b, p, F, GT_R, LBV_ZETA, F->hsbl, 1);
    o_exchange2d(b, GT_R, F->hsbl);
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        ksbl(i, j) = 1;
        for (int k = N; k >= 2; k--)
          if ((ksbl(i, j) == 1) && (z_w(i, j, k - 1) < hsbl(i, j))) ksbl(i, j) = k;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        double zgrid = z_w(i, j, N) - hsbl(i, j);
        if (mk) zgrid = zgrid * rmask(i, j);                                   /* :669 */
        const double swdk = o_swfrac(p, -1.0, zgrid);
        Bfsfc(i, j) = (Bo(i, j) + Bosol(i, j) * (1.0 - swdk));
        if (mk) Bfsfc(i, j) = Bfsfc(i, j) * rmask(i, j);                       /* :681 */
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        sl_dpth(i, j) = lmd_epsilon * (z_w(i, j, N) - hsbl(i, j));
        const double cff = (Bfsfc(i, j) > 0.0) ? 1.0 : lmd_epsilon;
        const double sigma = cff * (z_w(i, j, N) - hsbl(i, j));
        double wmk, wsk;
        o_wscale(Ustar(i, j), sigma, Bfsfc(i, j), &wmk, &wsk);
        wm(i, j) = wmk; ws(i, j) = wsk;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++)
        f1(i, j) = 5.0 * MAX(0.0, Bfsfc(i, j)) * vonKar / (Ustar(i, j) * Ustar(i, j) * Ustar(i, j) * Ustar(i, j) + eps);
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        const double zbl = z_w(i, j, N) - hsbl(i, j);
        if (hsbl(i, j) > z_w(i, j, 1)) {
          const int k = ksbl(i, j);
          const double cff = 1.0 / (z_w(i, j, k) - z_w(i, j, k - 1));
          const double cff_dn = cff * (hsbl(i, j) - z_w(i, j, k - 1));
          const double cff_up = cff * (z_w(i, j, k) - hsbl(i, j));
          double K_bl = cff_dn * Akv(i, j, k) + cff_up * Akv(i, j, k - 1);
          double dK_bl = cff * (Akv(i, j, k) - Akv(i, j, k - 1));
          Gm1(i, j) = K_bl / (zbl * wm(i, j) + eps);
          if (mk) Gm1(i, j) = Gm1(i, j) * rmask(i, j);                         /* :754 */
          dGm1dS(i, j) = MIN(0.0, -dK_bl / (wm(i, j) + eps) - K_bl * f1(i, j));
          K_bl = cff_dn * Akt(i, j, k, itemp) + cff_up * Akt(i, j, k - 1, itemp);
          dK_bl = cff * (Akt(i, j, k, itemp) - Akt(i, j, k - 1, itemp));
          Gt1(i, j) = K_bl / (zbl * ws(i, j) + eps);
          if (mk) Gt1(i, j) = Gt1(i, j) * rmask(i, j);                         /* :765 */
          dGt1dS(i, j) = MIN(0.0, -dK_bl / (ws(i, j) + eps) - K_bl * f1(i, j));
          K_bl = cff_dn * Akt(i, j, k, isalt) + cff_up * Akt(i, j, k - 1, isalt);
          dK_bl = cff * (Akt(i, j, k, isalt) - Akt(i, j, k - 1, isalt));
          Gs1(i, j) = K_bl / (zbl * ws(i, j) + eps);
          if (mk) Gs1(i, j) = Gs1(i, j) * rmask(i, j);                         /* :777 */
          dGs1dS(i, j) = MIN(0.0, -dK_bl / (ws(i, j) + eps) - K_bl * f1(i, j));
        } else {
          ksbl(i, j) = 0;
          const double b1 = 0.5 * (bustr(i, j) + bustr(i + 1, j)), b2 = 0.5 * (bvstr(i, j) + bvstr(i, j + 1));
          double Ustarb = sqrt(sqrt(b1 * b1 + b2 * b2));
          if (mk) Ustarb = Ustarb * rmask(i, j);                               /* :793 */
          const double dK_bl = vonKar * Ustarb;
          const double K_bl = dK_bl * (hsbl(i, j) - z_w(i, j, 0));
          Gm1(i, j) = K_bl / (zbl * wm(i, j) + eps);
          if (mk) Gm1(i, j) = Gm1(i, j) * rmask(i, j);                         /* :799 */
          dGm1dS(i, j) = MIN(0.0, -dK_bl / (wm(i, j) + eps) - K_bl * f1(i, j));
          Gt1(i, j) = K_bl / (zbl * ws(i, j) + eps);
          if (mk) Gt1(i, j) = Gt1(i, j) * rmask(i, j);                         /* :808 */
          dGt1dS(i, j) = MIN(0.0, -dK_bl / (ws(i, j) + eps) - K_bl * f1(i, j));
          Gs1(i, j) = Gt1(i, j);
          dGs1dS(i, j) = dGt1dS(i, j);
        }
      }
    for (int k = 1; k <= N - 1; k++)
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          const double zbl = z_w(i, j, N) - hsbl(i, j);
          if (k > ksbl(i, j)) {
            const double depth = z_w(i, j, N) - z_w(i, j, k);
            double sigma;
            if (Bflux(i, j, k) < 0.0) sigma = MIN(sl_dpth(i, j), depth);
            else sigma = depth;
            double wmk, wsk;
            o_wscale(Ustar(i, j), sigma, Bflux(i, j, k), &wmk, &wsk);
            wm(i, j) = wmk; ws(i, j) = wsk;
            sigma = depth / (zbl + eps);
            if (mk) sigma = sigma * rmask(i, j);                               /* :866 */
            const double a1 = sigma - 2.0, a2 = 3.0 - 2.0 * sigma, a3 = sigma - 1.0;
            const double Gm = a1 + a2 * Gm1(i, j) + a3 * dGm1dS(i, j);
            const double Gt = a1 + a2 * Gt1(i, j) + a3 * dGt1dS(i, j);
            const double Gs = a1 + a2 * Gs1(i, j) + a3 * dGs1dS(i, j);
            Akv(i, j, k) = depth * wmk * (1.0 + sigma * Gm);
            Akt(i, j, k, itemp) = depth * wsk * (1.0 + sigma * Gt);
            Akt(i, j, k, isalt) = depth * wsk * (1.0 + sigma * Gs);
            const double cff = lmd_Cg * (1.0 - (0.5 + copysign(0.5, Bflux(i, j, k)))) / (zbl * wsk + eps);
            ghats(i, j, k, itemp) = cff * ghats(i, j, k, itemp);
            ghats(i, j, k, isalt) = cff * ghats(i, j, k, isalt);
          } else {
            ghats(i, j, k, itemp) = 0.0;
            ghats(i, j, k, isalt) = 0.0;
          }
        }
  }

  /* ================= lmd_finish_tile, lmd_vmix.F:520-700 ================= */
  for (int k = 1; k <= N - 1; k++)
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        double cff = MAX(bvf(i, j, k), lmd_bvfcon);
        cff = MIN(1.0, (lmd_bvfcon - cff) / lmd_bvfcon);
        double nu_sxc = 1.0 - cff * cff;
        nu_sxc = nu_sxc * nu_sxc * nu_sxc;
        Akv(i, j, k) = Akv(i, j, k) + lmd_nu0c * nu_sxc;
        Akt(i, j, k, itemp) = Akt(i, j, k, itemp) + lmd_nu0c * nu_sxc;
        Akt(i, j, k, isalt) = Akt(i, j, k, isalt) + lmd_nu0c * nu_sxc;
      }
  /* boundary values exactly as written there, :620-700 (note Iend-1 on the eastern edge and that
   * the edge copies are not guarded by periodicity; the periodic exchange afterwards restores the
   * ghost columns but the copy INTO column Iend-1 stays) */
  for (int k = 0; k <= N; k++) {
    if (west_edge)
      for (int j = Jstr; j <= Jend; j++) {
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(Istr - 1, j, k, itrc) = Akt(Istr, j, k, itrc);
        Akv(Istr - 1, j, k) = Akv(Istr, j, k);
      }
    if (east_edge)
      for (int j = Jstr; j <= Jend; j++) {
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(Iend - 1, j, k, itrc) = Akt(Iend, j, k, itrc);
        Akv(Iend - 1, j, k) = Akv(Iend, j, k);
      }
    if (south_edge)
      for (int i = Istr; i <= Iend; i++) {
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(i, Jstr - 1, k, itrc) = Akt(i, Jstr, k, itrc);
        Akv(i, Jstr - 1, k) = Akv(i, Jstr, k);
      }
    if (north_edge)
      for (int i = Istr; i <= Iend; i++) {
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(i, Jend + 1, k, itrc) = Akt(i, Jend, k, itrc);
        Akv(i, Jend + 1, k) = Akv(i, Jend, k);
      }
    if (south_edge && west_edge) {
      for (int itrc = 1; itrc <= NAT; itrc++)
        Akt(Istr - 1, Jstr - 1, k, itrc) = 0.5 * (Akt(Istr, Jstr - 1, k, itrc) + Akt(Istr - 1, Jstr, k, itrc));
      Akv(Istr - 1, Jstr - 1, k) = 0.5 * (Akv(Istr, Jstr - 1, k) + Akv(Istr - 1, Jstr, k));
    }
    if (south_edge && east_edge) {
      for (int itrc = 1; itrc <= NAT; itrc++)
        Akt(Iend + 1, Jstr - 1, k, itrc) = 0.5 * (Akt(Iend, Jstr - 1, k, itrc) + Akt(Iend + 1, Jstr, k, itrc));
      Akv(Iend + 1, Jstr - 1, k) = 0.5 * (Akv(Iend, Jstr - 1, k) + Akv(Iend + 1, Jstr, k));
    }
    if (north_edge && west_edge) {
      for (int itrc = 1; itrc <= NAT; itrc++)
        Akt(Istr - 1, Jend + 1, k, itrc) = 0.5 * (Akt(Istr, Jend + 1, k, itrc) + Akt(Istr - 1, Jend, k, itrc));
      Akv(Istr - 1, Jend + 1, k) = 0.5 * (Akv(Istr, Jend + 1, k) + Akv(Istr - 1, Jend, k));
    }
    if (north_edge && east_edge) {
      for (int itrc = 1; itrc <= NAT; itrc++)
        Akt(Iend + 1, Jend + 1, k, itrc) = 0.5 * (Akt(Iend, Jend + 1, k, itrc) + Akt(Iend + 1, Jend, k, itrc));
      Akv(Iend + 1, Jend + 1, k) = 0.5 * (Akv(Iend, Jend + 1, k) + Akv(Iend + 1, Jend, k));
    }
  }
  o_bc_w3d(b, F->Akv);
  for (int itrc = 1; itrc <= NAT; itrc++) o_bc_w3d(b, &Akt(LBi, LBj, 0, itrc));
  free(Rig_); free(Bflux_); free(FC_); free(dR_); free(dU_); free(dV_); free(Bo_); free(Bosol_); free(Bfsfc_);
  free(Ustar_); free(sl_dpth_); free(wm_); free(ws_); free(f1_); free(Gm1_); free(Gt1_); free(Gs1_);
  free(dGm1dS_); free(dGt1dS_); free(dGs1dS_); free(ksbl_);
  return 0;
}

/* ana_srflux_tile, ALBEDO branch (ROMS/Functionals/ana_srflux.h:120-150): yday, hour = what caldate returns
 * for tdays(ng) (dateclock.F:73).  Csolar = 1353 W/m2, Cp = 3985 J/kg/K (mod_scalars.F). */
int oracle_ana_srflux(OARGS, double yday, double hour)
{
  ORACLE_PROLOGUE
  const double pi = 3.14159265358979323846, deg2rad = pi / 180.0;
  const double Csolar = 1353.0, Cp = 3985.0, alb_w = 0.06;
  double Dangle = 23.44 * cos((172.0 - yday) * 2.0 * pi / 365.2425);
  Dangle = Dangle * deg2rad;
  const double Hangle = (12.0 - hour) * pi / 12.0;
  const double Rsolar = Csolar / (p->rho0 * Cp);
  for (int j = JstrT; j <= JendT; j++)
    for (int i = IstrT; i <= IendT; i++) {
      const double LatRad = latr(i, j) * deg2rad;
      const double cff1 = sin(LatRad) * sin(Dangle);
      const double cff2 = cos(LatRad) * cos(Dangle);
      double sr = 0.0;
      const double zenith = cff1 + cff2 * cos(Hangle - lonr(i, j) * deg2rad);
      if (zenith > 0.0) {
        const double cff = (0.7859 + 0.03477 * F->Tair[I2(i, j)]) / (1.0 + 0.00412 * F->Tair[I2(i, j)]);
        const double e_sat = pow(10.0, cff);
        const double vap_p = e_sat * F->Hair[I2(i, j)];
        const double cl = F->cloud[I2(i, j)];
        sr = Rsolar * zenith * zenith * (1.0 - 0.6 * (cl * cl * cl)) /
             ((zenith + 2.7) * vap_p * 1.0E-3 + 1.085 * zenith + 0.1);
      }
      F->srflx[I2(i, j)] = (1.0 - alb_w) * sr;
    }
  return 0;
}
