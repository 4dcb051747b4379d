/*
 * oracle_wetdry.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * WET_DRY: the time-dependent wet/dry masks, ROMS/Nonlinear/wetdry.F, restated loop for loop.
 *   o_wetdry        wetdry_tile           wetdry.F:93    every step2d call (step2d_LF_AM3.h:729-749)
 *   oracle_wetdry   wetdry_ini_tile       wetdry.F:395   the initial masks (initial.F:438-466)
 *   wd_mask         wetdry_mask_tile      wetdry.F:563   masks from a rho-point wet/dry flag
 *   wd_avg_mask     wetdry_avg_mask_tile  wetdry.F:734   masks of the baroclinic step: a one-sided face is open
 *                                                        only to flow out of its wet cell
 * LuvSrc: the output masks count source faces as water (o_src_masks, oracle_sources.c).  PARITY UNPINNED: wetdry.F
 * USEs mod_sources (netCDF) and cannot be built here; known-answer and property tests in tests/test_wetdry.py.
 */
#include "oracle.h"

#define wetdry(i,j) wd[WS2(i,j)]

/* the PSI-point rule shared by wetdry_mask_tile (:631-683) and wetdry_avg_mask_tile (:834-886): 1 with four or three
 * wet neighbours, 2 with two wet neighbours on the same side, 0 otherwise (diagonal pairs included) */
static void wd_pmask(const roms_bounds_t *b, roms_fields_t *F, const double *wd)
{
  ORACLE_PROLOGUE
  const double cff1 = 1.0, cff2 = 2.0;
  for (int j = Jstr; j <= JendR; j++)
    for (int i = Istr; i <= IendR; i++) {
      const int a = wetdry(i - 1, j) > 0.5, bb = wetdry(i, j) > 0.5, c = wetdry(i - 1, j - 1) > 0.5, d = wetdry(i, j - 1) > 0.5;
      const int al = wetdry(i - 1, j) < 0.5, bl = wetdry(i, j) < 0.5, cl = wetdry(i - 1, j - 1) < 0.5, dl = wetdry(i, j - 1) < 0.5;
      if (a && bb && c && d) pmask_wet(i, j) = 1.0;
      else if (al && bb && c && d) pmask_wet(i, j) = cff1;
      else if (a && bl && c && d) pmask_wet(i, j) = cff1;
      else if (a && bb && cl && d) pmask_wet(i, j) = cff1;
      else if (a && bb && c && dl) pmask_wet(i, j) = cff1;
      else if (a && bl && c && dl) pmask_wet(i, j) = cff2;
      else if (al && bb && cl && d) pmask_wet(i, j) = cff2;
      else if (a && bb && cl && dl) pmask_wet(i, j) = cff2;
      else if (al && bl && c && d) pmask_wet(i, j) = cff2;
      else pmask_wet(i, j) = 0.0;
    }
}

static void wd_exchange(const roms_bounds_t *b, roms_fields_t *F)
{
  o_exchange2d(b, GT_P, F->pmask_wet);
  o_exchange2d(b, GT_R, F->rmask_wet);
  o_exchange2d(b, GT_U, F->umask_wet);
  o_exchange2d(b, GT_V, F->vmask_wet);
}

/* wetdry_mask_tile, wetdry.F:563-716 */
static void wd_mask(const roms_bounds_t *b, roms_fields_t *F, const double *wd)
{
  ORACLE_PROLOGUE
  for (int j = JstrR; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) rmask_wet(i, j) = wetdry(i, j);
  for (int j = JstrR; j <= JendR; j++)
    for (int i = Istr; i <= IendR; i++) {
      umask_wet(i, j) = wetdry(i - 1, j) + wetdry(i, j);
      if (umask_wet(i, j) == 1.0) umask_wet(i, j) = wetdry(i - 1, j) - wetdry(i, j);
    }
  for (int j = Jstr; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) {
      vmask_wet(i, j) = wetdry(i, j - 1) + wetdry(i, j);
      if (vmask_wet(i, j) == 1.0) vmask_wet(i, j) = wetdry(i, j - 1) - wetdry(i, j);
    }
  wd_pmask(b, F, wd);
  wd_exchange(b, F);
}

/* wetdry_avg_mask_tile, wetdry.F:734-917: DU, DV = DU_avg1, DV_avg1 (or, from wetdry_ini_tile, ubar, vbar) */
static void wd_avg_mask(const roms_bounds_t *b, roms_fields_t *F, const double *wd, const double *DU, const double *DV)
{
  ORACLE_PROLOGUE
  double cff1, cff5, cff6;
  for (int j = JstrR; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) rmask_wet(i, j) = wetdry(i, j);
  for (int j = JstrR; j <= JendR; j++)
    for (int i = Istr; i <= IendR; i++) {
      cff1 = wetdry(i - 1, j) + wetdry(i, j);
      if (cff1 == 1.0) cff1 = wetdry(i - 1, j) - wetdry(i, j);
      cff5 = fabs(fabs(cff1) - 1.0);
      cff6 = 0.5 + copysign(0.5, DU[I2(i, j)]) * cff1;
      umask_wet(i, j) = 0.5 * cff1 * cff5 + cff6 * (1.0 - cff5);
      if (DU[I2(i, j)] == 0.0)                                   /* catch lone ponds */
        if ((wetdry(i - 1, j) + wetdry(i, j)) <= 1.0) umask_wet(i, j) = 0.0;
    }
  for (int j = Jstr; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) {
      cff1 = wetdry(i, j - 1) + wetdry(i, j);
      if (cff1 == 1.0) cff1 = wetdry(i, j - 1) - wetdry(i, j);
      cff5 = fabs(fabs(cff1) - 1.0);
      cff6 = 0.5 + copysign(0.5, DV[I2(i, j)]) * cff1;
      vmask_wet(i, j) = 0.5 * cff1 * cff5 + cff6 * (1.0 - cff5);
      if (DV[I2(i, j)] == 0.0)
        if ((wetdry(i, j - 1) + wetdry(i, j)) <= 1.0) vmask_wet(i, j) = 0.0;
    }
  wd_pmask(b, F, wd);
  wd_exchange(b, F);
}

/* the "full" masks, wetdry.F:325-345 / :478-498 (as written: pmask_full is never below 2) */
static void wd_full(OARGS)
{
  ORACLE_PROLOGUE
  for (int j = JstrR; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) rmask_full(i, j) = rmask_wet(i, j) * rmask(i, j);
  for (int j = Jstr; j <= JendR; j++)
    for (int i = Istr; i <= IendR; i++) pmask_full(i, j) = MAX(pmask_wet(i, j) * pmask(i, j), 2.0);
  for (int j = JstrR; j <= JendR; j++)
    for (int i = Istr; i <= IendR; i++) umask_full(i, j) = umask_wet(i, j) * umask(i, j);
  for (int j = Jstr; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) vmask_full(i, j) = vmask_wet(i, j) * vmask(i, j);
  o_src_masks(b, p, s, F);                           /* LuvSrc, wetdry.F:307-320 / :511-524 */
  o_exchange2d(b, GT_P, F->pmask_full);
  o_exchange2d(b, GT_R, F->rmask_full);
  o_exchange2d(b, GT_U, F->umask_full);
  o_exchange2d(b, GT_V, F->vmask_full);
}

/* the rho-point flag of both routines, wetdry.F:190-200 / :451-461: wet = sea and a total depth above Dcrit */
static void wd_flag(const roms_bounds_t *b, const roms_params_t *p, roms_fields_t *F, const double *Z, double *wd)
{
  ORACLE_PROLOGUE
  const double eps = 1.0E-10;
  for (int j = Jstr - 1; j <= JendR; j++)
    for (int i = Istr - 1; i <= IendR; i++) {
      wetdry(i, j) = 1.0;
      if (p->masking) wetdry(i, j) = wetdry(i, j) * rmask(i, j);
      if ((Z[I2(i, j)] + h(i, j)) <= (p->Dcrit + eps)) wetdry(i, j) = 0.0;
    }
}

/* wetdry_tile, wetdry.F:93-393, with zeta(:,:,kstp) as step2d passes it (step2d_LF_AM3.h:740) */
void o_wetdry(OARGS)
{
  ORACLE_PROLOGUE
  const int iif = s->iif, nfast = p->nfast;
  double *wd = walloc(nis * njs);
  wd_flag(b, p, F, &zeta(LBi, LBj, s->kstp), wd);
  if (iif <= nfast) wd_mask(b, F, wd);
  if (iif <= nfast) {
    if (s->predictor_2d_step && iif == 1) {
      for (int j = JstrR; j <= JendR; j++)
        for (int i = IstrR; i <= IendR; i++) rmask_wet_avg(i, j) = wetdry(i, j);
    } else {
      for (int j = JstrR; j <= JendR; j++)
        for (int i = IstrR; i <= IendR; i++) rmask_wet_avg(i, j) = rmask_wet_avg(i, j) + wetdry(i, j);
    }
    o_exchange2d(b, GT_R, F->rmask_wet_avg);
  } else {
    /* after the last fast step: wet only where every one of the 2*nfast calls found the cell wet */
    const double cff = 1.0 / (double)(2 * nfast);
    for (int j = Jstr - 1; j <= JendR; j++)
      for (int i = Istr - 1; i <= IendR; i++) wetdry(i, j) = trunc(rmask_wet_avg(i, j) * cff);
    wd_avg_mask(b, F, wd, F->DU_avg1, F->DV_avg1);
  }
  if (iif > nfast) wd_full(b, p, s, F);
  free(wd);
}

/* wetdry(ng, tile, Tindex, .TRUE.) -> wetdry_ini_tile, wetdry.F:395-561, with Tindex = s->kstp */
int oracle_wetdry(OARGS)
{
  ORACLE_PROLOGUE
  if (!p->wet_dry) return 0;
  double *wd = walloc(nis * njs);
  wd_flag(b, p, F, &zeta(LBi, LBj, s->kstp), wd);
  wd_avg_mask(b, F, wd, &ubar(LBi, LBj, s->kstp), &vbar(LBi, LBj, s->kstp));
  wd_full(b, p, s, F);
  free(wd);
  return 0;
}
