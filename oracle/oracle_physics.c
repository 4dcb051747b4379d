/*
 * oracle_physics.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Per-step physics between the hot kernels:
 *   set_vbc_tile    ROMS/Nonlinear/set_vbc.F:104   (SOLVE3D; UV_QDRAG or UV_LDRAG; SALINITY, no EMINUSP)
 *   bulk_flux_tile  ROMS/Nonlinear/bulk_flux.F:146 (COARE 3.0; LONGWAVE = Berliand formula; no COOL_SKIN,
 *                                                   no EMINUSP, no WIND_MINUS_CURRENT, no masking)
 * Both reference files compile stand-alone and are compared with this restatement in
 * tests/test_ref_pinning.py (bulk_flux to a few ulp: LOG/EXP/ATAN/pow come from different math
 * libraries).
 */
#include "oracle.h"

/* generic closed-wall conditions of bc_2d.F (bc_u2d_tile :205, bc_v2d_tile :400) for a 2-D array */
static void o_bc_u2d_generic(const roms_bounds_t *b, const roms_params_t *p, double *A)
{
  const int LBi = b->LBi, LBj = b->LBj;
  const long ni = b->UBi - b->LBi + 1;
  const int Imin = b->EWperiodic ? b->IstrU : b->Istr, Imax = b->EWperiodic ? b->Iend : b->IendR;
  if (!b->NSperiodic) {
    if (b->north_edge) for (int i = Imin; i <= Imax; i++) A[I2(i, b->Jend + 1)] = p->gamma2 * A[I2(i, b->Jend)];
    if (b->south_edge) for (int i = Imin; i <= Imax; i++) A[I2(i, b->Jstr - 1)] = p->gamma2 * A[I2(i, b->Jstr)];
  }
}
static void o_bc_v2d_generic(const roms_bounds_t *b, double *A)
{
  const int LBi = b->LBi, LBj = b->LBj;
  const long ni = b->UBi - b->LBi + 1;
  if (!b->NSperiodic) {
    if (b->north_edge) for (int i = b->Istr; i <= b->Iend; i++) A[I2(i, b->Jend + 1)] = 0.0;
    if (b->south_edge) for (int i = b->Istr; i <= b->Iend; i++) A[I2(i, b->Jstr)] = 0.0;
  }
}

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
  } else return 8;
  /* boundary conditions + periodic / tile exchange, :472-500 */
  o_bc_u2d_generic(b, p, F->bustr);
  o_bc_v2d_generic(b, F->bvstr);
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
  const int nrhs = s->nrhs, itemp = 1, IterMax = 3;
  /* mod_scalars.F:431-444, :1415-1421 */
  const double Cp = 3985.0, StefBo = 5.67E-8, emmiss = 0.97, rhow = 1000.0, vonKar = 0.41;
  const double blk_Cpa = 1004.67, blk_Cpw = 4000.0, blk_Rgas = 287.1, blk_Zabl = 600.0, blk_beta = 1.2;
  const double pi = 3.14159265358979323846;
  const double g = p->g, rho0 = p->rho0;
  const double blk_ZQ = p->blk_ZQ, blk_ZT = p->blk_ZT, blk_ZW = p->blk_ZW;
  const double eps = 1.0E-20, r3 = 1.0 / 3.0;
  (void)rhow;
  double *Taux_ = walloc(nis * njs), *Tauy_ = walloc(nis * njs), *LHeat_ = walloc(nis * njs);
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
      const double Hl = -Hlv * rhoAir * Wstar * Qstar;
      const double upvel = -1.61 * Wstar * Qstar - (1.0 + 1.61 * Q) * Wstar * Tstar / TairK;
      const double Hlw = rhoAir * Hlv * upvel * Q;
      LHeat(i, j) = (Hl + Hlw);
      const double Taur = 0.85 * rain(i, j) * Wmag;
      cff = rhoAir * Cd * Wspeed;
      Taux(i, j) = (cff * Ua + Taur * copysign(1.0, Ua));
      Tauy(i, j) = (cff * Va + Taur * copysign(1.0, Va));
    }
  /* kinematic fluxes, :790-860 */
  Hscale = 1.0 / (rho0 * Cp);
  for (int j = JstrR; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) {
      lrflx(i, j) = LRad(i, j) * Hscale;
      lhflx(i, j) = -LHeat(i, j) * Hscale;
      shflx(i, j) = -SHeat(i, j) * Hscale;
      stflux(i, j, itemp) = (srflx(i, j) + lrflx(i, j) + lhflx(i, j) + shflx(i, j));
    }
  const double cffs = 0.5 / rho0;
  for (int j = JstrR; j <= JendR; j++)
    for (int i = Istr; i <= IendR; i++) F->sustr[I2(i, j)] = cffs * (Taux(i - 1, j) + Taux(i, j));
  for (int j = Jstr; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) F->svstr[I2(i, j)] = cffs * (Tauy(i, j - 1) + Tauy(i, j));
  o_exchange2d(b, GT_R, F->lrflx);
  o_exchange2d(b, GT_R, F->lhflx);
  o_exchange2d(b, GT_R, F->shflx);
  o_exchange2d(b, GT_R, &stflux(LBi, LBj, itemp));
  o_exchange2d(b, GT_U, F->sustr);
  o_exchange2d(b, GT_V, F->svstr);
  free(Taux_); free(Tauy_); free(LHeat_); free(SHeat_); free(LRad_);
  return 0;
}
