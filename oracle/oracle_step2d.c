/*
 * oracle_step2d.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * step2d_tile: one predictor (leap-frog) or corrector (Adams-Moulton 3)
 * barotropic sub-step with fast-time averaging and 2D<->3D coupling
 * (ROMS/Nonlinear/step2d_LF_AM3.h:137-2528), and the LOOP_2D sequencing of
 * main3d.F:592-700.  Parity unpinned (mod_sources).
 */
#include "oracle.h"

int oracle_step2d(OARGS)
{
  ORACLE_PROLOGUE
  if (o_src_check(p)) return 8;
  if (o_check_lbc(b, p)) return 8;
  const int krhs = s->krhs, kstp = s->kstp, knew = s->knew, nstp = s->nstp, nnew = s->nnew;
  const int iif = s->iif, iic = s->iic, ntfirst = s->ntfirst, nfast = p->nfast;
  const int PREDICTOR = s->predictor_2d_step, CORRECTOR = !PREDICTOR;
  const double dtfast = p->dtfast, g = p->g, rho0 = p->rho0;
  const int ptsk = 3 - kstp;
  double cff, cff1, cff2, cff3, cff4, cff5, fac, fac1;
  /* private arrays allocated with the FIELD extents (DUon/DVom are exchanged) */
  double *Dgrad_ = walloc(nij), *Dnew_ = walloc(nij), *Drhs_ = walloc(nij), *Drhs_p_ = walloc(nij), *Dstp_ = walloc(nij);
  double *DUon_ = walloc(nij), *DVom_ = walloc(nij), *UFe_ = walloc(nij), *UFx_ = walloc(nij), *VFe_ = walloc(nij), *VFx_ = walloc(nij);
  double *grad_ = walloc(nij), *gzeta_ = walloc(nij), *gzeta2_ = walloc(nij), *gzetaSA_ = walloc(nij);
  double *rhs_ubar_ = walloc(nij), *rhs_vbar_ = walloc(nij), *rhs_zeta_ = walloc(nij), *zeta_new_ = walloc(nij), *zwrk_ = walloc(nij);
#define Dgrad(i,j) Dgrad_[I2(i,j)]
#define Dnew(i,j) Dnew_[I2(i,j)]
#define Drhs(i,j) Drhs_[I2(i,j)]
#define Drhs_p(i,j) Drhs_p_[I2(i,j)]
#define Dstp(i,j) Dstp_[I2(i,j)]
#define DUon(i,j) DUon_[I2(i,j)]
#define DVom(i,j) DVom_[I2(i,j)]
#define UFe(i,j) UFe_[I2(i,j)]
#define UFx(i,j) UFx_[I2(i,j)]
#define VFe(i,j) VFe_[I2(i,j)]
#define VFx(i,j) VFx_[I2(i,j)]
#define grad(i,j) grad_[I2(i,j)]
#define gzeta(i,j) gzeta_[I2(i,j)]
#define gzeta2(i,j) gzeta2_[I2(i,j)]
#define gzetaSA(i,j) gzetaSA_[I2(i,j)]
#define rhs_ubar(i,j) rhs_ubar_[I2(i,j)]
#define rhs_vbar(i,j) rhs_vbar_[I2(i,j)]
#define rhs_zeta(i,j) rhs_zeta_[I2(i,j)]
#define zeta_new(i,j) zeta_new_[I2(i,j)]
#define zwrk(i,j) zwrk_[I2(i,j)]
#define FREE_ALL free(Dgrad_); free(Dnew_); free(Drhs_); free(Drhs_p_); free(Dstp_); free(DUon_); free(DVom_); \
  free(UFe_); free(UFx_); free(VFe_); free(VFx_); free(grad_); free(gzeta_); free(gzeta2_); free(gzetaSA_); \
  free(rhs_ubar_); free(rhs_vbar_); free(rhs_zeta_); free(zeta_new_); free(zwrk_);

  /* total depth and transports on the extended range, :509-590 */
  for (int j = JstrV - 2; j <= Jendp2; j++)
    for (int i = IstrU - 2; i <= Iendp2; i++) Drhs(i, j) = zeta(i, j, krhs) + h(i, j);
  for (int j = JstrV - 2; j <= Jendp2; j++)
    for (int i = IstrU - 1; i <= Iendp2; i++) {
      cff = 0.5 * on_u(i, j);
      cff1 = cff * (Drhs(i, j) + Drhs(i - 1, j));
      DUon(i, j) = ubar(i, j, krhs) * cff1;
    }
  for (int j = JstrV - 1; j <= Jendp2; j++)
    for (int i = IstrU - 2; i <= Iendp2; i++) {
      cff = 0.5 * om_v(i, j);
      cff1 = cff * (Drhs(i, j) + Drhs(i, j - 1));
      DVom(i, j) = vbar(i, j, krhs) * cff1;
    }
  o_exchange2d(b, GT_U, DUon_);
  o_exchange2d(b, GT_V, DVom_);

  /* fast-time averaging, :614-727 */
  if (PREDICTOR) {
    if (iif == 1) {
      cff2 = (-1.0 / 12.0) * p->weight2[iif + 1 - 1];
      for (int j = JstrR; j <= JendR; j++) {
        for (int i = IstrR; i <= IendR; i++) Zt_avg1(i, j) = 0.0;
        for (int i = Istr; i <= IendR; i++) { DU_avg1(i, j) = 0.0; DU_avg2(i, j) = cff2 * DUon(i, j); }
      }
      for (int j = Jstr; j <= JendR; j++)
        for (int i = IstrR; i <= IendR; i++) { DV_avg1(i, j) = 0.0; DV_avg2(i, j) = cff2 * DVom(i, j); }
    } else {
      cff1 = p->weight1[iif - 1 - 1];
      cff2 = (8.0 / 12.0) * p->weight2[iif - 1] - (1.0 / 12.0) * p->weight2[iif + 1 - 1];
      for (int j = JstrR; j <= JendR; j++) {
        for (int i = IstrR; i <= IendR; i++) Zt_avg1(i, j) = Zt_avg1(i, j) + cff1 * zeta(i, j, krhs);
        for (int i = Istr; i <= IendR; i++) {
          DU_avg1(i, j) = DU_avg1(i, j) + cff1 * DUon(i, j);
          DU_avg2(i, j) = DU_avg2(i, j) + cff2 * DUon(i, j);
        }
      }
      for (int j = Jstr; j <= JendR; j++)
        for (int i = IstrR; i <= IendR; i++) {
          DV_avg1(i, j) = DV_avg1(i, j) + cff1 * DVom(i, j);
          DV_avg2(i, j) = DV_avg2(i, j) + cff2 * DVom(i, j);
        }
    }
  } else {
    if (iif == 1) cff2 = p->weight2[iif - 1];
    else cff2 = (5.0 / 12.0) * p->weight2[iif - 1];
    for (int j = JstrR; j <= JendR; j++)
      for (int i = Istr; i <= IendR; i++) DU_avg2(i, j) = DU_avg2(i, j) + cff2 * DUon(i, j);
    for (int j = Jstr; j <= JendR; j++)
      for (int i = IstrR; i <= IendR; i++) DV_avg2(i, j) = DV_avg2(i, j) + cff2 * DVom(i, j);
  }
  if (iif == nfast + 1 && PREDICTOR) {
    o_exchange2d(b, GT_R, F->Zt_avg1);
    o_exchange2d(b, GT_U, F->DU_avg1);
    o_exchange2d(b, GT_V, F->DV_avg1);
  }
  if (p->wet_dry) o_wetdry(b, p, s, F);                 /* WET_DRY: the new wet/dry masks, :729-749 */
  if (iif > nfast) { FREE_ALL return 0; }

  /* free-surface step, :770-929 */
  fac = 1000.0 / rho0;
  if (iif == 1) {
    cff1 = dtfast;
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        rhs_zeta(i, j) = (DUon(i, j) - DUon(i + 1, j)) + (DVom(i, j) - DVom(i, j + 1));
        zeta_new(i, j) = zeta(i, j, kstp) + pm(i, j) * pn(i, j) * cff1 * rhs_zeta(i, j);
        if (p->masking) zeta_new(i, j) = zeta_new(i, j) * rmask(i, j);      /* MASKING, :778/:804/:835 */
        Dnew(i, j) = zeta_new(i, j) + h(i, j);
        zwrk(i, j) = 0.5 * (zeta(i, j, kstp) + zeta_new(i, j));
        gzeta(i, j) = (fac + rhoS(i, j)) * zwrk(i, j);
        gzeta2(i, j) = gzeta(i, j) * zwrk(i, j);
        gzetaSA(i, j) = zwrk(i, j) * (rhoS(i, j) - rhoA(i, j));
      }
  } else if (PREDICTOR) {
    cff1 = 2.0 * dtfast;
    cff4 = 4.0 / 25.0;
    cff5 = 1.0 - 2.0 * cff4;
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        rhs_zeta(i, j) = (DUon(i, j) - DUon(i + 1, j)) + (DVom(i, j) - DVom(i, j + 1));
        zeta_new(i, j) = zeta(i, j, kstp) + pm(i, j) * pn(i, j) * cff1 * rhs_zeta(i, j);
        if (p->masking) zeta_new(i, j) = zeta_new(i, j) * rmask(i, j);      /* MASKING, :778/:804/:835 */
        Dnew(i, j) = zeta_new(i, j) + h(i, j);
        zwrk(i, j) = cff5 * zeta(i, j, krhs) + cff4 * (zeta(i, j, kstp) + zeta_new(i, j));
        gzeta(i, j) = (fac + rhoS(i, j)) * zwrk(i, j);
        gzeta2(i, j) = gzeta(i, j) * zwrk(i, j);
        gzetaSA(i, j) = zwrk(i, j) * (rhoS(i, j) - rhoA(i, j));
      }
  } else {
    cff1 = dtfast * 5.0 / 12.0;
    cff2 = dtfast * 8.0 / 12.0;
    cff3 = dtfast * 1.0 / 12.0;
    cff4 = 2.0 / 5.0;
    cff5 = 1.0 - cff4;
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = cff1 * ((DUon(i, j) - DUon(i + 1, j)) + (DVom(i, j) - DVom(i, j + 1)));
        zeta_new(i, j) = zeta(i, j, kstp) + pm(i, j) * pn(i, j) * (cff + cff2 * rzeta(i, j, kstp) - cff3 * rzeta(i, j, ptsk));
        if (p->masking) zeta_new(i, j) = zeta_new(i, j) * rmask(i, j);      /* MASKING, :778/:804/:835 */
        Dnew(i, j) = zeta_new(i, j) + h(i, j);
        zwrk(i, j) = cff5 * zeta_new(i, j) + cff4 * zeta(i, j, krhs);
        gzeta(i, j) = (fac + rhoS(i, j)) * zwrk(i, j);
        gzeta2(i, j) = gzeta(i, j) * zwrk(i, j);
        gzetaSA(i, j) = zwrk(i, j) * (rhoS(i, j) - rhoA(i, j));
      }
  }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      zeta(i, j, knew) = zeta_new(i, j);
      /* WET_DRY && MASKING, :863-866: the total depth of a land cell stays at Dcrit */
      if (p->wet_dry && p->masking) zeta(i, j, knew) = zeta(i, j, knew) + (p->Dcrit - h(i, j)) * (1.0 - rmask(i, j));
    }
  if (PREDICTOR) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) rzeta(i, j, krhs) = rhs_zeta(i, j);
    o_exchange2d(b, GT_R, &rzeta(LBi, LBj, krhs));
  }
  o_src_zeta(b, p, s, F, knew);                      /* LwSrc, :890-908 */
  o_zetabc(b, p, s, F, knew);
  o_exchange2d(b, GT_R, &zeta(LBi, LBj, knew));

  /* pressure gradient with VAR_RHO_2D, :939-1019 */
  cff1 = 0.5 * g;
  cff2 = 1.0 / 3.0;
  for (int j = Jstr; j <= Jend; j++) {
    for (int i = IstrU; i <= Iend; i++)
      rhs_ubar(i, j) = cff1 * on_u(i, j) *
                       ((h(i - 1, j) + h(i, j)) * (gzeta(i - 1, j) - gzeta(i, j)) +
                        (h(i - 1, j) - h(i, j)) * (gzetaSA(i - 1, j) + gzetaSA(i, j) +
                                                   cff2 * (rhoA(i - 1, j) - rhoA(i, j)) * (zwrk(i - 1, j) - zwrk(i, j))) +
                        (gzeta2(i - 1, j) - gzeta2(i, j)));
    if (j >= JstrV)
      for (int i = Istr; i <= Iend; i++)
        rhs_vbar(i, j) = cff1 * om_v(i, j) *
                         ((h(i, j - 1) + h(i, j)) * (gzeta(i, j - 1) - gzeta(i, j)) +
                          (h(i, j - 1) - h(i, j)) * (gzetaSA(i, j - 1) + gzetaSA(i, j) +
                                                     cff2 * (rhoA(i, j - 1) - rhoA(i, j)) * (zwrk(i, j - 1) - zwrk(i, j))) +
                          (gzeta2(i, j - 1) - gzeta2(i, j)));
  }
  if (p->uv_adv) {
    /* 4th-order centred advection, :1079-1283 */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrUm1; i <= Iendp1; i++) {
        grad(i, j) = ubar(i - 1, j, krhs) - 2.0 * ubar(i, j, krhs) + ubar(i + 1, j, krhs);
        Dgrad(i, j) = DUon(i - 1, j) - 2.0 * DUon(i, j) + DUon(i + 1, j);
      }
    if (!EWperiodic) {
      if (west_edge) for (int j = Jstr; j <= Jend; j++) { grad(Istr, j) = grad(Istr + 1, j); Dgrad(Istr, j) = Dgrad(Istr + 1, j); }
      if (east_edge) for (int j = Jstr; j <= Jend; j++) { grad(Iend + 1, j) = grad(Iend, j); Dgrad(Iend + 1, j) = Dgrad(Iend, j); }
    }
    cff = 1.0 / 6.0;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++)
        UFx(i, j) = 0.25 * (ubar(i, j, krhs) + ubar(i + 1, j, krhs) - cff * (grad(i, j) + grad(i + 1, j))) *
                    (DUon(i, j) + DUon(i + 1, j) - cff * (Dgrad(i, j) + Dgrad(i + 1, j)));
    for (int j = Jstrm1; j <= Jendp1; j++)
      for (int i = IstrU; i <= Iend; i++)
        grad(i, j) = ubar(i, j - 1, krhs) - 2.0 * ubar(i, j, krhs) + ubar(i, j + 1, krhs);
    if (!NSperiodic) {
      if (south_edge) for (int i = IstrU; i <= Iend; i++) grad(i, Jstr - 1) = grad(i, Jstr);
      if (north_edge) for (int i = IstrU; i <= Iend; i++) grad(i, Jend + 1) = grad(i, Jend);
    }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = IstrU - 1; i <= Iend; i++) Dgrad(i, j) = DVom(i - 1, j) - 2.0 * DVom(i, j) + DVom(i + 1, j);
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = IstrU; i <= Iend; i++)
        UFe(i, j) = 0.25 * (ubar(i, j, krhs) + ubar(i, j - 1, krhs) - cff * (grad(i, j) + grad(i, j - 1))) *
                    (DVom(i, j) + DVom(i - 1, j) - cff * (Dgrad(i, j) + Dgrad(i - 1, j)));
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istrm1; i <= Iendp1; i++)
        grad(i, j) = vbar(i - 1, j, krhs) - 2.0 * vbar(i, j, krhs) + vbar(i + 1, j, krhs);
    if (!EWperiodic) {
      if (west_edge) for (int j = JstrV; j <= Jend; j++) grad(Istr - 1, j) = grad(Istr, j);
      if (east_edge) for (int j = JstrV; j <= Jend; j++) grad(Iend + 1, j) = grad(Iend, j);
    }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++) Dgrad(i, j) = DUon(i, j - 1) - 2.0 * DUon(i, j) + DUon(i, j + 1);
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++)
        VFx(i, j) = 0.25 * (vbar(i, j, krhs) + vbar(i - 1, j, krhs) - cff * (grad(i, j) + grad(i - 1, j))) *
                    (DUon(i, j) + DUon(i, j - 1) - cff * (Dgrad(i, j) + Dgrad(i, j - 1)));
    for (int j = JstrVm1; j <= Jendp1; j++)
      for (int i = Istr; i <= Iend; i++) {
        grad(i, j) = vbar(i, j - 1, krhs) - 2.0 * vbar(i, j, krhs) + vbar(i, j + 1, krhs);
        Dgrad(i, j) = DVom(i, j - 1) - 2.0 * DVom(i, j) + DVom(i, j + 1);
      }
    if (!NSperiodic) {
      if (south_edge) for (int i = Istr; i <= Iend; i++) { grad(i, Jstr) = grad(i, Jstr + 1); Dgrad(i, Jstr) = Dgrad(i, Jstr + 1); }
      if (north_edge) for (int i = Istr; i <= Iend; i++) { grad(i, Jend + 1) = grad(i, Jend); Dgrad(i, Jend + 1) = Dgrad(i, Jend); }
    }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++)
        VFe(i, j) = 0.25 * (vbar(i, j, krhs) + vbar(i, j + 1, krhs) - cff * (grad(i, j) + grad(i, j + 1))) *
                    (DVom(i, j) + DVom(i, j + 1) - cff * (Dgrad(i, j) + Dgrad(i, j + 1)));
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff1 = UFx(i, j) - UFx(i - 1, j);
        cff2 = UFe(i, j + 1) - UFe(i, j);
        fac = cff1 + cff2;
        rhs_ubar(i, j) = rhs_ubar(i, j) - fac;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = VFx(i + 1, j) - VFx(i, j);
        cff2 = VFe(i, j) - VFe(i, j - 1);
        fac = cff1 + cff2;
        rhs_vbar(i, j) = rhs_vbar(i, j) - fac;
      }
  }
  if (p->uv_cor) {
    /* Coriolis, :1291-1325 */
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = 0.5 * Drhs(i, j) * fomn(i, j);
        UFx(i, j) = cff * (vbar(i, j, krhs) + vbar(i, j + 1, krhs));
        VFe(i, j) = cff * (ubar(i, j, krhs) + ubar(i + 1, j, krhs));
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) { fac1 = 0.5 * (UFx(i, j) + UFx(i - 1, j)); rhs_ubar(i, j) = rhs_ubar(i, j) + fac1; }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) { fac1 = 0.5 * (VFe(i, j) + VFe(i, j - 1)); rhs_vbar(i, j) = rhs_vbar(i, j) - fac1; }
  }
  if (p->curvgrid && p->uv_adv) {
    /* curvilinear terms, :1333-1382 */
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff1 = 0.5 * (vbar(i, j, krhs) + vbar(i, j + 1, krhs));
        cff2 = 0.5 * (ubar(i, j, krhs) + ubar(i + 1, j, krhs));
        cff3 = cff1 * dndx(i, j);
        cff4 = cff2 * dmde(i, j);
        cff = Drhs(i, j) * (cff3 - cff4);
        UFx(i, j) = cff * cff1;
        VFe(i, j) = cff * cff2;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) { fac1 = 0.5 * (UFx(i, j) + UFx(i - 1, j)); rhs_ubar(i, j) = rhs_ubar(i, j) + fac1; }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) { fac1 = 0.5 * (VFe(i, j) + VFe(i, j - 1)); rhs_vbar(i, j) = rhs_vbar(i, j) - fac1; }
  }
  if (p->uv_vis2 || p->uv_vis4)          /* total depth at psi-points, :1380-1390 */
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++)
        Drhs_p(i, j) = 0.25 * (Drhs(i, j) + Drhs(i - 1, j) + Drhs(i, j - 1) + Drhs(i - 1, j - 1));
  if (p->uv_vis2) {
    /* harmonic viscosity, :1394-1471 */
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = visc2_r(i, j) * Drhs(i, j) * 0.5 *
              (pmon_r(i, j) * ((pn(i, j) + pn(i + 1, j)) * ubar(i + 1, j, krhs) - (pn(i - 1, j) + pn(i, j)) * ubar(i, j, krhs)) -
               pnom_r(i, j) * ((pm(i, j) + pm(i, j + 1)) * vbar(i, j + 1, krhs) - (pm(i, j - 1) + pm(i, j)) * vbar(i, j, krhs)));
        UFx(i, j) = on_r(i, j) * on_r(i, j) * cff;
        VFe(i, j) = om_r(i, j) * om_r(i, j) * cff;
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff = visc2_p(i, j) * Drhs_p(i, j) * 0.5 *
              (pmon_p(i, j) * ((pn(i, j - 1) + pn(i, j)) * vbar(i, j, krhs) - (pn(i - 1, j - 1) + pn(i - 1, j)) * vbar(i - 1, j, krhs)) +
               pnom_p(i, j) * ((pm(i - 1, j) + pm(i, j)) * ubar(i, j, krhs) - (pm(i - 1, j - 1) + pm(i, j - 1)) * ubar(i, j - 1, krhs)));
        if (p->masking) cff = cff * pmask(i, j);                                        /* MASKING, :1433 */
        if (p->wet_dry) cff = cff * pmask_wet(i, j);                                    /* WET_DRY, :1436/:1512/:1707 */
        UFe(i, j) = om_p(i, j) * om_p(i, j) * cff;
        VFx(i, j) = on_p(i, j) * on_p(i, j) * cff;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff1 = 0.5 * (pn(i - 1, j) + pn(i, j)) * (UFx(i, j) - UFx(i - 1, j));
        cff2 = 0.5 * (pm(i - 1, j) + pm(i, j)) * (UFe(i, j + 1) - UFe(i, j));
        fac = cff1 + cff2;
        rhs_ubar(i, j) = rhs_ubar(i, j) + fac;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = 0.5 * (pn(i, j - 1) + pn(i, j)) * (VFx(i + 1, j) - VFx(i, j));
        cff2 = 0.5 * (pm(i, j - 1) + pm(i, j)) * (VFe(i, j) - VFe(i, j - 1));
        fac = cff1 - cff2;
        rhs_vbar(i, j) = rhs_vbar(i, j) + fac;
      }
  }

  if (p->uv_vis4) {
    /* biharmonic viscosity, :1474-1740: the harmonic operator without the depth (m s^-3/2) on a range one point
     * wider, its rule on physical edges and corners, then the operator with the depth */
    const double gamma2 = p->gamma2;
    double *LapU_ = walloc(nij), *LapV_ = walloc(nij);
#define LapU(i,j) LapU_[I2(i,j)]
#define LapV(i,j) LapV_[I2(i,j)]
#define visc4_p(i,j) F->visc4_p[I2(i,j)]
#define visc4_r(i,j) F->visc4_r[I2(i,j)]
    for (int j = JstrVm2; j <= Jendp1; j++)
      for (int i = IstrUm2; i <= Iendp1; i++) {
        cff = visc4_r(i, j) * 0.5 *
              (pmon_r(i, j) * ((pn(i, j) + pn(i + 1, j)) * ubar(i + 1, j, krhs) - (pn(i - 1, j) + pn(i, j)) * ubar(i, j, krhs)) -
               pnom_r(i, j) * ((pm(i, j) + pm(i, j + 1)) * vbar(i, j + 1, krhs) - (pm(i, j - 1) + pm(i, j)) * vbar(i, j, krhs)));
        UFx(i, j) = on_r(i, j) * on_r(i, j) * cff;
        VFe(i, j) = om_r(i, j) * om_r(i, j) * cff;
      }
    for (int j = Jstrm1; j <= Jendp2; j++)
      for (int i = Istrm1; i <= Iendp2; i++) {
        cff = visc4_p(i, j) * 0.5 *
              (pmon_p(i, j) * ((pn(i, j - 1) + pn(i, j)) * vbar(i, j, krhs) - (pn(i - 1, j - 1) + pn(i - 1, j)) * vbar(i - 1, j, krhs)) +
               pnom_p(i, j) * ((pm(i - 1, j) + pm(i, j)) * ubar(i, j, krhs) - (pm(i - 1, j - 1) + pm(i, j - 1)) * ubar(i, j - 1, krhs)));
        if (p->masking) cff = cff * pmask(i, j);
        if (p->wet_dry) cff = cff * pmask_wet(i, j);                                    /* WET_DRY, :1436/:1512/:1707 */
        UFe(i, j) = om_p(i, j) * om_p(i, j) * cff;
        VFx(i, j) = on_p(i, j) * on_p(i, j) * cff;
      }
    for (int j = Jstrm1; j <= Jendp1; j++)
      for (int i = IstrUm1; i <= Iendp1; i++)
        LapU(i, j) = 0.125 * (pm(i - 1, j) + pm(i, j)) * (pn(i - 1, j) + pn(i, j)) *
                     ((pn(i - 1, j) + pn(i, j)) * (UFx(i, j) - UFx(i - 1, j)) +
                      (pm(i - 1, j) + pm(i, j)) * (UFe(i, j + 1) - UFe(i, j)));
    for (int j = JstrVm1; j <= Jendp1; j++)
      for (int i = Istrm1; i <= Iendp1; i++)
        LapV(i, j) = 0.125 * (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1)) *
                     ((pn(i, j - 1) + pn(i, j)) * (VFx(i + 1, j) - VFx(i, j)) -
                      (pm(i, j - 1) + pm(i, j)) * (VFe(i, j) - VFe(i, j - 1)));
    if (!EWperiodic) {
      if (west_edge) {
        const int cu = o_lbc(p, LBS_WEST, LBV_UBAR) == LBC_CLOSED, cv = o_lbc(p, LBS_WEST, LBV_VBAR) == LBC_CLOSED;
        for (int j = Jstrm1; j <= Jendp1; j++) LapU(IstrU - 1, j) = cu ? 0.0 : LapU(IstrU, j);
        for (int j = JstrVm1; j <= Jendp1; j++) LapV(Istr - 1, j) = cv ? gamma2 * LapV(Istr, j) : 0.0;
      }
      if (east_edge) {
        const int cu = o_lbc(p, LBS_EAST, LBV_UBAR) == LBC_CLOSED, cv = o_lbc(p, LBS_EAST, LBV_VBAR) == LBC_CLOSED;
        for (int j = Jstrm1; j <= Jendp1; j++) LapU(Iend + 1, j) = cu ? 0.0 : LapU(Iend, j);
        for (int j = JstrVm1; j <= Jendp1; j++) LapV(Iend + 1, j) = cv ? gamma2 * LapV(Iend, j) : 0.0;
      }
    }
    if (!NSperiodic) {
      if (south_edge) {
        const int cu = o_lbc(p, LBS_SOUTH, LBV_UBAR) == LBC_CLOSED, cv = o_lbc(p, LBS_SOUTH, LBV_VBAR) == LBC_CLOSED;
        for (int i = IstrUm1; i <= Iendp1; i++) LapU(i, Jstr - 1) = cu ? gamma2 * LapU(i, Jstr) : 0.0;
        for (int i = Istrm1; i <= Iendp1; i++) LapV(i, JstrV - 1) = cv ? 0.0 : LapV(i, JstrV);
      }
      if (north_edge) {
        const int cu = o_lbc(p, LBS_NORTH, LBV_UBAR) == LBC_CLOSED, cv = o_lbc(p, LBS_NORTH, LBV_VBAR) == LBC_CLOSED;
        for (int i = IstrUm1; i <= Iendp1; i++) LapU(i, Jend + 1) = cu ? gamma2 * LapU(i, Jend) : 0.0;
        for (int i = Istrm1; i <= Iendp1; i++) LapV(i, Jend + 1) = cv ? 0.0 : LapV(i, Jend);
      }
    }
    if (!(NSperiodic || EWperiodic)) {
      if (south_edge && west_edge) {
        LapU(Istr, Jstr - 1) = 0.5 * (LapU(Istr + 1, Jstr - 1) + LapU(Istr, Jstr));
        LapV(Istr - 1, Jstr) = 0.5 * (LapV(Istr - 1, Jstr + 1) + LapV(Istr, Jstr));
      }
      if (south_edge && east_edge) {
        LapU(Iend + 1, Jstr - 1) = 0.5 * (LapU(Iend, Jstr - 1) + LapU(Iend + 1, Jstr));
        LapV(Iend + 1, Jstr) = 0.5 * (LapV(Iend, Jstr) + LapV(Iend + 1, Jstr + 1));
      }
      if (north_edge && west_edge) {
        LapU(Istr, Jend + 1) = 0.5 * (LapU(Istr + 1, Jend + 1) + LapU(Istr, Jend));
        LapV(Istr - 1, Jend + 1) = 0.5 * (LapV(Istr, Jend + 1) + LapV(Istr - 1, Jend));
      }
      if (north_edge && east_edge) {
        LapU(Iend + 1, Jend + 1) = 0.5 * (LapU(Iend, Jend + 1) + LapU(Iend + 1, Jend));
        LapV(Iend + 1, Jend + 1) = 0.5 * (LapV(Iend, Jend + 1) + LapV(Iend + 1, Jend));
      }
    }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = visc4_r(i, j) * Drhs(i, j) * 0.5 *
              (pmon_r(i, j) * ((pn(i, j) + pn(i + 1, j)) * LapU(i + 1, j) - (pn(i - 1, j) + pn(i, j)) * LapU(i, j)) -
               pnom_r(i, j) * ((pm(i, j) + pm(i, j + 1)) * LapV(i, j + 1) - (pm(i, j - 1) + pm(i, j)) * LapV(i, j)));
        UFx(i, j) = on_r(i, j) * on_r(i, j) * cff;
        VFe(i, j) = om_r(i, j) * om_r(i, j) * cff;
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff = visc4_p(i, j) * Drhs_p(i, j) * 0.5 *
              (pmon_p(i, j) * ((pn(i, j - 1) + pn(i, j)) * LapV(i, j) - (pn(i - 1, j - 1) + pn(i - 1, j)) * LapV(i - 1, j)) +
               pnom_p(i, j) * ((pm(i - 1, j) + pm(i, j)) * LapU(i, j) - (pm(i - 1, j - 1) + pm(i, j - 1)) * LapU(i, j - 1)));
        if (p->masking) cff = cff * pmask(i, j);
        if (p->wet_dry) cff = cff * pmask_wet(i, j);                                    /* WET_DRY, :1436/:1512/:1707 */
        UFe(i, j) = om_p(i, j) * om_p(i, j) * cff;
        VFx(i, j) = on_p(i, j) * on_p(i, j) * cff;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff1 = 0.5 * (pn(i - 1, j) + pn(i, j)) * (UFx(i, j) - UFx(i - 1, j));
        cff2 = 0.5 * (pm(i - 1, j) + pm(i, j)) * (UFe(i, j + 1) - UFe(i, j));
        fac = cff1 + cff2;
        rhs_ubar(i, j) = rhs_ubar(i, j) - fac;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = 0.5 * (pn(i, j - 1) + pn(i, j)) * (VFx(i + 1, j) - VFx(i, j));
        cff2 = 0.5 * (pm(i, j - 1) + pm(i, j)) * (VFe(i, j) - VFe(i, j - 1));
        fac = cff1 - cff2;
        rhs_vbar(i, j) = rhs_vbar(i, j) - fac;
      }
    free(LapU_); free(LapV_);
#undef LapU
#undef LapV
  }

  /* coupling between 2-D and 3-D equations, :1884-2065 */
  if (iif == 1 && PREDICTOR) {
    if (iic == ntfirst) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          rufrc(i, j) = rufrc(i, j) - rhs_ubar(i, j);
          rhs_ubar(i, j) = rhs_ubar(i, j) + rufrc(i, j);
          ru(i, j, 0, nstp) = rufrc(i, j);
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          rvfrc(i, j) = rvfrc(i, j) - rhs_vbar(i, j);
          rhs_vbar(i, j) = rhs_vbar(i, j) + rvfrc(i, j);
          rv(i, j, 0, nstp) = rvfrc(i, j);
        }
    } else if (iic == ntfirst + 1) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          rufrc(i, j) = rufrc(i, j) - rhs_ubar(i, j);
          rhs_ubar(i, j) = rhs_ubar(i, j) + 1.5 * rufrc(i, j) - 0.5 * ru(i, j, 0, nnew);
          ru(i, j, 0, nstp) = rufrc(i, j);
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          rvfrc(i, j) = rvfrc(i, j) - rhs_vbar(i, j);
          rhs_vbar(i, j) = rhs_vbar(i, j) + 1.5 * rvfrc(i, j) - 0.5 * rv(i, j, 0, nnew);
          rv(i, j, 0, nstp) = rvfrc(i, j);
        }
    } else {
      cff1 = 23.0 / 12.0;
      cff2 = 16.0 / 12.0;
      cff3 = 5.0 / 12.0;
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          rufrc(i, j) = rufrc(i, j) - rhs_ubar(i, j);
          rhs_ubar(i, j) = rhs_ubar(i, j) + cff1 * rufrc(i, j) - cff2 * ru(i, j, 0, nnew) + cff3 * ru(i, j, 0, nstp);
          ru(i, j, 0, nstp) = rufrc(i, j);
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          rvfrc(i, j) = rvfrc(i, j) - rhs_vbar(i, j);
          rhs_vbar(i, j) = rhs_vbar(i, j) + cff1 * rvfrc(i, j) - cff2 * rv(i, j, 0, nnew) + cff3 * rv(i, j, 0, nstp);
          rv(i, j, 0, nstp) = rvfrc(i, j);
        }
    }
  } else {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) rhs_ubar(i, j) = rhs_ubar(i, j) + rufrc(i, j);
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) rhs_vbar(i, j) = rhs_vbar(i, j) + rvfrc(i, j);
  }

  /* time-step the 2-D momentum equations, :2098-2255 */
  for (int j = JstrV - 1; j <= Jend; j++)
    for (int i = IstrU - 1; i <= Iend; i++) Dstp(i, j) = zeta(i, j, kstp) + h(i, j);
  if (iif == 1 || PREDICTOR) {
    cff1 = (iif == 1) ? 0.5 * dtfast : dtfast;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = (pm(i, j) + pm(i - 1, j)) * (pn(i, j) + pn(i - 1, j));
        fac = 1.0 / (Dnew(i, j) + Dnew(i - 1, j));
        ubar(i, j, knew) = (ubar(i, j, kstp) * (Dstp(i, j) + Dstp(i - 1, j)) + cff * cff1 * rhs_ubar(i, j)) * fac;
        if (p->masking) ubar(i, j, knew) = ubar(i, j, knew) * umask(i, j);      /* MASKING, :2120/:2175 */
        if (p->wet_dry) {                                                        /* WET_DRY, :2123-2135 ... */
          const double cff7 = o_wet_factor(umask_wet(i, j), ubar(i, j, knew));
          ubar(i, j, knew) = ubar(i, j, knew) * cff7;
          rhs_ubar(i, j) = rhs_ubar(i, j) * cff7;
          if (iif == 1 && PREDICTOR) {                                           /* FIRST_2D_STEP only, :2129-2133 */
            rufrc(i, j) = rufrc(i, j) * cff7;
            ru(i, j, 0, nstp) = rufrc(i, j);
          }
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1));
        fac = 1.0 / (Dnew(i, j) + Dnew(i, j - 1));
        vbar(i, j, knew) = (vbar(i, j, kstp) * (Dstp(i, j) + Dstp(i, j - 1)) + cff * cff1 * rhs_vbar(i, j)) * fac;
        if (p->masking) vbar(i, j, knew) = vbar(i, j, knew) * vmask(i, j);      /* MASKING, :2145/:2194 */
        if (p->wet_dry) {                                                        /* WET_DRY, :2123-2135 ... */
          const double cff7 = o_wet_factor(vmask_wet(i, j), vbar(i, j, knew));
          vbar(i, j, knew) = vbar(i, j, knew) * cff7;
          rhs_vbar(i, j) = rhs_vbar(i, j) * cff7;
          if (iif == 1 && PREDICTOR) {                                           /* FIRST_2D_STEP only, :2129-2133 */
            rvfrc(i, j) = rvfrc(i, j) * cff7;
            rv(i, j, 0, nstp) = rvfrc(i, j);
          }
        }
      }
  } else {
    cff1 = 0.5 * dtfast * 5.0 / 12.0;
    cff2 = 0.5 * dtfast * 8.0 / 12.0;
    cff3 = 0.5 * dtfast * 1.0 / 12.0;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = (pm(i, j) + pm(i - 1, j)) * (pn(i, j) + pn(i - 1, j));
        fac = 1.0 / (Dnew(i, j) + Dnew(i - 1, j));
        ubar(i, j, knew) = (ubar(i, j, kstp) * (Dstp(i, j) + Dstp(i - 1, j)) +
                            cff * (cff1 * rhs_ubar(i, j) + cff2 * rubar(i, j, kstp) - cff3 * rubar(i, j, ptsk))) * fac;
        if (p->masking) ubar(i, j, knew) = ubar(i, j, knew) * umask(i, j);      /* MASKING, :2120/:2175 */
        if (p->wet_dry) {                                                        /* WET_DRY, :2123-2135 ... */
          const double cff7 = o_wet_factor(umask_wet(i, j), ubar(i, j, knew));
          ubar(i, j, knew) = ubar(i, j, knew) * cff7;
          rhs_ubar(i, j) = rhs_ubar(i, j) * cff7;
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1));
        fac = 1.0 / (Dnew(i, j) + Dnew(i, j - 1));
        vbar(i, j, knew) = (vbar(i, j, kstp) * (Dstp(i, j) + Dstp(i, j - 1)) +
                            cff * (cff1 * rhs_vbar(i, j) + cff2 * rvbar(i, j, kstp) - cff3 * rvbar(i, j, ptsk))) * fac;
        if (p->masking) vbar(i, j, knew) = vbar(i, j, knew) * vmask(i, j);      /* MASKING, :2145/:2194 */
        if (p->wet_dry) {                                                        /* WET_DRY, :2123-2135 ... */
          const double cff7 = o_wet_factor(vmask_wet(i, j), vbar(i, j, knew));
          vbar(i, j, knew) = vbar(i, j, knew) * cff7;
          rhs_vbar(i, j) = rhs_vbar(i, j) * cff7;
        }
      }
  }
  if (PREDICTOR) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) rubar(i, j, krhs) = rhs_ubar(i, j);
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) rvbar(i, j, krhs) = rhs_vbar(i, j);
  }
  o_u2dbc(b, p, s, F, knew);
  o_v2dbc(b, p, s, F, knew);
  o_src_ubar(b, p, s, F, knew);                      /* LuvSrc, step2d_LF_AM3.h:2484-2502 */
  o_exchange2d(b, GT_U, &ubar(LBi, LBj, knew));
  o_exchange2d(b, GT_V, &vbar(LBi, LBj, knew));
  FREE_ALL
  return 0;
}

/* LOOP_2D of main3d.F:592-700.  s->iif/kstp/krhs/knew/predictor are driven here. */
int oracle_step2d_loop(const roms_bounds_t *b, const roms_params_t *p, roms_step_idx_t *s,
                       roms_fields_t *F, int *indx1)
{
  const int nfast = p->nfast;
  int predictor = 0, rc;
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
    if ((rc = oracle_step2d(b, p, s, F))) return rc;
    if (predictor) {
      predictor = 0;
      s->knew = next_indx1;
      s->kstp = 3 - s->knew;
      s->krhs = 3;
      if (s->iif < nfast + 1) *indx1 = next_indx1;
    }
    s->predictor_2d_step = predictor;
    if (s->iif < nfast + 1)
      if ((rc = oracle_step2d(b, p, s, F))) return rc;
  }
  return 0;
}
