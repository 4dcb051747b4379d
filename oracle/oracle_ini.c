/* oracle_ini.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product).
 *
 * ini_zeta_tile and ini_fields_tile (ROMS/Nonlinear/ini_fields.F:836-1137, :106-777), the two initialisation
 * routines main3d calls on the first time step (main3d.F:269-283) before the first set_massflux: they load the
 * other time levels from the initial state, apply the MASKING multiplies and the lateral boundary conditions, and
 * derive ubar, vbar from the vertical integral of u, v.  SOLVE3D, no PERFECT_RESTART, no sediment; WET_DRY: the wet-mask products and the Dcrit floor of zeta.
 * Pinned against the reference's own routines (tests/test_ref_pinning.py, oracle/ref_wrap.F90 ref_bc kinds 7, 8).
 */
#include "oracle.h"
#include <stdlib.h>

static int any_lbc(const roms_params_t *p, int v, int c1, int c2)
{
  for (int sd = 0; sd < 4; sd++) {
    const int c = o_lbc(p, sd, v);
    if (c == c1 || c == c2) return 1;
    if (c1 == LBC_RADIATION && c == LBC_RADIATION_NUDGING) return 1;      /* RadNud sets LBC%radiation too */
    if (c2 == LBC_CHAPMAN_IMPLICIT && c == LBC_CHAPMAN_EXPLICIT) return 1; /* ini_fields.F:932-934 names both */
  }
  return 0;
}

/* ini_zeta_tile -- ini_fields.F:930-1080 */
int oracle_ini_zeta(OARGS)
{
  ORACLE_PROLOGUE
  if (o_check_lbc(b, p)) return 8;
  const int kstp = s->kstp, knew = s->knew;
  /* radiation / Chapman edges keep their initial boundary values: the whole array is loaded and zetabc is not
   * applied (:932-944, :971-974) */
  const int open = any_lbc(p, LBV_ZETA, LBC_RADIATION, LBC_CHAPMAN_IMPLICIT);
  const int Imin = open ? IstrT : b->IstrB, Imax = open ? IendT : b->IendB;
  const int Jmin = open ? JstrT : b->JstrB, Jmax = open ? JendT : b->JendB;
  for (int j = Jmin; j <= Jmax; j++)
    for (int i = Imin; i <= Imax; i++) {
      double cff1 = zeta(i, j, kstp);
      if (p->masking) cff1 = cff1 * rmask(i, j);
      if (p->wet_dry && cff1 <= (p->Dcrit - h(i, j))) cff1 = p->Dcrit - h(i, j);   /* WET_DRY, ini_fields.F:951-957 */
      zeta(i, j, kstp) = cff1;
      zeta(i, j, knew) = cff1;
    }
  if (!open) {
    o_zetabc(b, p, s, F, kstp);
    o_zetabc(b, p, s, F, knew);
  }
  o_exchange2d(b, GT_R, &zeta(LBi, LBj, kstp));
  if (knew != kstp) o_exchange2d(b, GT_R, &zeta(LBi, LBj, knew));
  if (p->wet_dry) o_exchange2d(b, GT_R, F->rmask_wet);                             /* :1020-1024, :1060-1066 */
  /* fast-time averaged free surface, :1062-1080 */
  for (int j = JstrT; j <= JendT; j++)
    for (int i = IstrT; i <= IendT; i++) Zt_avg1(i, j) = zeta(i, j, kstp);
  o_exchange2d(b, GT_R, F->Zt_avg1);
  return 0;
}

/* ini_fields_tile -- ini_fields.F:280-660 */
int oracle_ini_fields(OARGS)
{
  ORACLE_PROLOGUE
  if (o_check_lbc(b, p)) return 8;
  const int kstp = s->kstp, knew = s->knew, nstp = s->nstp, nnew = s->nnew;
  const int IstrB = b->IstrB, IendB = b->IendB, JstrB = b->JstrB, JendB = b->JendB, IstrM = b->IstrM, JstrM = b->JstrM;
  const int mk = p->masking;
  /* 3-D momentum: the other time level, :286-318 */
  for (int j = JstrB; j <= JendB; j++)
    for (int k = 1; k <= N; k++) {
      for (int i = IstrM; i <= IendB; i++) {
        double cff1 = u(i, j, k, nstp);
        if (mk) cff1 = cff1 * umask(i, j);
        if (p->wet_dry) cff1 = cff1 * umask_wet(i, j);            /* WET_DRY, ini_fields.F:292 / :307 / :398 / :423 */
        u(i, j, k, nstp) = cff1;
        u(i, j, k, nnew) = cff1;
      }
      if (j >= JstrM)
        for (int i = IstrB; i <= IendB; i++) {
          double cff2 = v(i, j, k, nstp);
          if (mk) cff2 = cff2 * vmask(i, j);
          if (p->wet_dry) cff2 = cff2 * vmask_wet(i, j);            /* WET_DRY, ini_fields.F:292 / :307 / :398 / :423 */
          v(i, j, k, nstp) = cff2;
          v(i, j, k, nnew) = cff2;
        }
    }
  o_u3dbc(b, p, s, F, nstp);
  o_v3dbc(b, p, s, F, nstp);
  o_u3dbc(b, p, s, F, nnew);
  o_v3dbc(b, p, s, F, nnew);
  o_exchange3d(b, GT_U, N, &u(LBi, LBj, 1, nstp));
  o_exchange3d(b, GT_V, N, &v(LBi, LBj, 1, nstp));
  o_exchange3d(b, GT_U, N, &u(LBi, LBj, 1, nnew));
  o_exchange3d(b, GT_V, N, &v(LBi, LBj, 1, nnew));
  /* vertically integrated momentum, :380-430: DC(i,0) the depth of the column at the velocity point, CF(i,0) the
   * integral, summed upwards from k = 1 */
  for (int j = JstrB; j <= JendB; j++) {
    for (int i = IstrM; i <= IendB; i++) {
      double DC0 = 0.0, CF0 = 0.0;
      for (int k = 1; k <= N; k++) {
        const double DC = 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k));
        DC0 = DC0 + DC;
        CF0 = CF0 + DC * u(i, j, k, nstp);
      }
      const double cff1 = 1.0 / DC0;
      double cff2 = CF0 * cff1;
      if (mk) cff2 = cff2 * umask(i, j);
      if (p->wet_dry) cff2 = cff2 * umask_wet(i, j);            /* WET_DRY, ini_fields.F:292 / :307 / :398 / :423 */
      ubar(i, j, kstp) = cff2;
      ubar(i, j, knew) = cff2;
    }
    if (j >= JstrM)
      for (int i = IstrB; i <= IendB; i++) {
        double DC0 = 0.0, CF0 = 0.0;
        for (int k = 1; k <= N; k++) {
          const double DC = 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k));
          DC0 = DC0 + DC;
          CF0 = CF0 + DC * v(i, j, k, nstp);
        }
        const double cff1 = 1.0 / DC0;
        double cff2 = CF0 * cff1;
        if (mk) cff2 = cff2 * vmask(i, j);
        if (p->wet_dry) cff2 = cff2 * vmask_wet(i, j);            /* WET_DRY, ini_fields.F:292 / :307 / :398 / :423 */
        vbar(i, j, kstp) = cff2;
        vbar(i, j, knew) = cff2;
      }
  }
  /* the 2-D conditions are applied unless an edge radiates or is a Flather edge, :434-460 */
  if (!any_lbc(p, LBV_UBAR, LBC_RADIATION, LBC_FLATHER) && !any_lbc(p, LBV_VBAR, LBC_RADIATION, LBC_FLATHER)) {
    o_u2dbc(b, p, s, F, kstp);
    o_v2dbc(b, p, s, F, kstp);
    o_u2dbc(b, p, s, F, knew);
    o_v2dbc(b, p, s, F, knew);
  }
  o_exchange2d(b, GT_U, &ubar(LBi, LBj, kstp));
  o_exchange2d(b, GT_V, &vbar(LBi, LBj, kstp));
  if (knew != kstp) {
    o_exchange2d(b, GT_U, &ubar(LBi, LBj, knew));
    o_exchange2d(b, GT_V, &vbar(LBi, LBj, knew));
  }
  /* tracers, :604-650 */
  for (int itrc = 1; itrc <= NT; itrc++) {
    for (int k = 1; k <= N; k++)
      for (int j = JstrB; j <= JendB; j++)
        for (int i = IstrB; i <= IendB; i++) {
          double cff1 = t(i, j, k, nstp, itrc);
          if (mk) cff1 = cff1 * rmask(i, j);
          t(i, j, k, nstp, itrc) = cff1;
          t(i, j, k, nnew, itrc) = cff1;
        }
    o_t3dbc(b, p, s, F, nstp, itrc);
    o_t3dbc(b, p, s, F, nnew, itrc);
  }
  for (int itrc = 1; itrc <= NT; itrc++) {
    o_exchange3d(b, GT_R, N, &t(LBi, LBj, 1, nstp, itrc));
    o_exchange3d(b, GT_R, N, &t(LBi, LBj, 1, nnew, itrc));
  }
  return 0;
}
