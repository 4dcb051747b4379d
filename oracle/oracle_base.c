/*
 * oracle_base.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Periodic exchanges, closed lateral boundary conditions and the small glue
 * kernels (set_massflux, omega, set_zeta, set_depth).
 */
#include "oracle.h"

int roms_abi_sizeof(int which)
{
  switch (which) {
  case 0: return (int)sizeof(roms_bounds_t);
  case 1: return (int)sizeof(roms_params_t);
  case 2: return (int)sizeof(roms_step_idx_t);
  case 3: return (int)sizeof(roms_fields_t);
  case 4: return (int)FID_COUNT;
  }
  return -1;
}

/* Effective boundary-condition code (enum roms_lbc) of variable v (enum roms_lbc_var) on side sd. */
int o_lbc(const roms_params_t *p, int sd, int v)
{
  if (p->lbc[sd][v]) return p->lbc[sd][v];
  return sd == LBS_WEST ? p->lbc_west : sd == LBS_EAST ? p->lbc_east : sd == LBS_SOUTH ? p->lbc_south : p->lbc_north;
}

/* West/east periodic; south/north per variable: closed, gradient, clamped, radiation (all six), Chapman implicit
 * (zeta), Flather (vbar: the normal component; ubar: the reference applies a Chapman-type condition to the
 * tangential component of a Flather edge) -- the conditions restated below. */
int o_check_lbc(const roms_bounds_t *b, const roms_params_t *p)
{
  if (!b->EWperiodic || b->NSperiodic) return 1;
  for (int sd = LBS_SOUTH; sd <= LBS_NORTH; sd++)
    for (int v = 0; v < LBV_COUNT; v++) {
      const int c = o_lbc(p, sd, v);
      int ok = c == LBC_CLOSED || c == LBC_GRADIENT || c == LBC_CLAMPED;
      ok = ok || c == LBC_RADIATION;
      if (v == LBV_ZETA) ok = ok || c == LBC_CHAPMAN_IMPLICIT;
      if (v == LBV_VBAR || v == LBV_UBAR) ok = ok || c == LBC_FLATHER;
      if (!ok) return 1;
    }
  return 0;
}

/* exchange_{r,u,v,p}2d_tile -- ROMS/Nonlinear/exchange_2d.F:43/229/416/603.
 * Periodic ghost copy, only when the direction is not partitioned. */
/* Multi-tile runs (CPU tests of the N>1 protocol): the test harness installs a
 * hook that performs mp_exchange2d/3d (mp_exchange.F:290/1413) between ranks. */
static o_exchange_hook_t g_hook = 0;
void oracle_set_exchange_hook(o_exchange_hook_t fn) { g_hook = fn; }

static void o_periodic2d(const roms_bounds_t *b, int gtype, double *A);

void o_exchange2d(const roms_bounds_t *b, int gtype, double *A)
{
  o_periodic2d(b, gtype, A);
  if (g_hook && b->ntileI * b->ntileJ > 1) g_hook(A, 1, gtype);
}

static void o_periodic2d(const roms_bounds_t *b, int gtype, double *A)
{
  const int Lm = b->Lm, LBi = b->LBi, LBj = b->LBj;
  const long ni = b->UBi - b->LBi + 1;
  if (!(b->EWperiodic && b->ntileI == 1)) return;
  int Jmin, Jmax;
  if (b->NSperiodic) { Jmin = b->Jstr; Jmax = b->Jend; }
  else {
    Jmin = (gtype == GT_R || gtype == GT_U) ? b->JstrR : b->Jstr;
    Jmax = b->JendR;
  }
#define A2(i,j) A[(long)((i) - LBi) + (long)((j) - LBj) * ni]
  if (b->west_edge) {
    for (int j = Jmin; j <= Jmax; j++) {
      A2(Lm + 1, j) = A2(1, j);
      A2(Lm + 2, j) = A2(2, j);
    }
    if (b->NghostPoints == 3)
      for (int j = Jmin; j <= Jmax; j++) A2(Lm + 3, j) = A2(3, j);
  }
  if (b->east_edge) {
    for (int j = Jmin; j <= Jmax; j++) {
      A2(-2, j) = A2(Lm - 2, j);
      A2(-1, j) = A2(Lm - 1, j);
      A2(0, j) = A2(Lm, j);
    }
  }
#undef A2
}

/* exchange_{r,u,v,w}3d_tile -- ROMS/Nonlinear/exchange_3d.F:259/471/683/896 */
void o_exchange3d(const roms_bounds_t *b, int gtype, int nk, double *A)
{
  const long nij = (long)(b->UBi - b->LBi + 1) * (b->UBj - b->LBj + 1);
  for (int k = 0; k < nk; k++) o_periodic2d(b, gtype, A + (long)k * nij);
  if (g_hook && b->ntileI * b->ntileJ > 1) g_hook(A, nk, gtype);
}

/* The time level `know` and the step dt2d of the 2-D boundary conditions (zetabc.F:96-106, v2dbc_im.F:116-126) */
static void o_know(const roms_params_t *p, const roms_step_idx_t *s, int *know, double *dt2d)
{
  if (s->iif == 1) { *know = s->krhs; *dt2d = p->dtfast; }
  else if (s->predictor_2d_step) { *know = s->krhs; *dt2d = 2.0 * p->dtfast; }
  else { *know = s->kstp; *dt2d = p->dtfast; }
}

/* Implicit upstream radiation condition on a southern / northern edge, the form shared by u3dbc_im.F:381-463 /
 * :539-621, v3dbc_im.F:97-180 / :239-322 and t3dbc_im.F:364-443 / :498-577 without nudging and without
 * RADIATION_2D (Cx = 0); the 2-D conditions (zetabc.F:408-470, u2dbc_im.F:833-908, v2dbc_im.F:138-214) have the
 * same form with the levels know / kout in place of nstp / nout.  xb_old = X(i,jb,nstp); x1_old, x1 = X(i,j1,nstp), X(i,j1,nout); x2 = X(i,j2,nout);
 * gL, gR = the two along-boundary differences of X(:,j1,nstp) on either side of point i. */
static double o_radiate(double xb_old, double x1_old, double x1, double x2, double gL, double gR)
{
  const double eps = 1.0E-20;
  double dXdt = x1_old - x1;
  const double dXde = x1 - x2;
  if ((dXdt * dXde) < 0.0) dXdt = 0.0;
  const double dXdx = ((dXdt * (gL + gR)) > 0.0) ? gL : gR;
  const double cff = MAX(dXdx * dXdx + dXde * dXde, eps);
  const double Ce = dXdt * dXde;
  return (cff * xb_old + Ce * x1) / (cff + Ce);
}

/* zetabc_tile, S/N edges -- ROMS/Nonlinear/zetabc.F:404-700: radiation (:408, :557), Chapman implicit (:489, :638),
 * clamped (:508, :657), gradient (:521, :670), closed (:534, :683); every branch ends with the MASKING multiply.
 * Radiation on the SOUTHERN edge takes its normal difference as zeta(i,Jstr,kout)-zeta(i,Jstr-1,kout), i.e. towards
 * the boundary row (:424; the northern edge, :573, and the other variables look into the interior) -- restated as
 * written. */
void o_zetabc(OARGS, int kout)
{
  ORACLE_PROLOGUE
  const int mk = p->masking;
  int know; double dt2d;
  o_know(p, s, &know, &dt2d);
  for (int side = 0; side < 2; side++) {
    if (!(side ? north_edge : south_edge)) continue;
    const int code = o_lbc(p, side ? LBS_NORTH : LBS_SOUTH, LBV_ZETA);
    const int jb = side ? Jend + 1 : Jstr - 1, j1 = side ? Jend : Jstr;
    for (int i = Istr; i <= Iend; i++) {
      if (code == LBC_RADIATION) {
        const int j2 = side ? Jend - 1 : Jstr - 1;
        double gL = zeta(i, j1, know) - zeta(i - 1, j1, know), gR = zeta(i + 1, j1, know) - zeta(i, j1, know);
        if (mk) { gL = gL * umask(i, j1); gR = gR * umask(i + 1, j1); }
        zeta(i, jb, kout) = o_radiate(zeta(i, jb, know), zeta(i, j1, know), zeta(i, j1, kout), zeta(i, j2, kout), gL, gR);
      } else if (code == LBC_CHAPMAN_IMPLICIT) {
        const double cff = dt2d * pn(i, j1);
        const double cff1 = sqrt(p->g * (h(i, j1) + zeta(i, j1, know)));
        const double Ce = cff * cff1;
        const double cff2 = 1.0 / (1.0 + Ce);
        zeta(i, jb, kout) = cff2 * (zeta(i, jb, know) + Ce * zeta(i, j1, kout));
      } else if (code == LBC_CLAMPED) zeta(i, jb, kout) = zeta_bry(i, jb);
      else zeta(i, jb, kout) = zeta(i, j1, kout);                   /* gradient, closed */
      if (mk) zeta(i, jb, kout) = zeta(i, jb, kout) * rmask(i, jb);
    }
  }
}

/* u2dbc_tile, S/N edges (tangential component) -- ROMS/Nonlinear/u2dbc_im.F:829-1140: radiation (:833, :991),
 * the Chapman-type condition of a Flather edge (:912, :1070), clamped (:934, :1092), gradient (:947, :1105),
 * closed = slipperiness gamma2 (:960, :1118) */
void o_u2dbc(OARGS, int kout)
{
  ORACLE_PROLOGUE
  const int mk = p->masking;
  int know; double dt2d;
  o_know(p, s, &know, &dt2d);
  for (int side = 0; side < 2; side++) {
    if (!(side ? north_edge : south_edge)) continue;
    const int code = o_lbc(p, side ? LBS_NORTH : LBS_SOUTH, LBV_UBAR);
    const int jb = side ? Jend + 1 : Jstr - 1, j1 = side ? Jend : Jstr;
    int Imin = IstrU, Imax = Iend;
    if (code == LBC_CLOSED) { Imin = EWperiodic ? IstrU : Istr; Imax = EWperiodic ? Iend : IendR; }
    for (int i = Imin; i <= Imax; i++) {
      if (code == LBC_RADIATION) {
        const int j2 = side ? Jend - 1 : Jstr + 1;
        const double gL = ubar(i, j1, know) - ubar(i - 1, j1, know), gR = ubar(i + 1, j1, know) - ubar(i, j1, know);
        ubar(i, jb, kout) = o_radiate(ubar(i, jb, know), ubar(i, j1, know), ubar(i, j1, kout), ubar(i, j2, kout), gL, gR);
      } else if (code == LBC_FLATHER) {
        const double cff = dt2d * 0.5 * (pn(i - 1, j1) + pn(i, j1));
        const double cff1 = sqrt(p->g * 0.5 * (h(i - 1, j1) + zeta(i - 1, j1, know) + h(i, j1) + zeta(i, j1, know)));
        const double Ce = cff * cff1;
        const double cff2 = 1.0 / (1.0 + Ce);
        ubar(i, jb, kout) = cff2 * (ubar(i, jb, know) + Ce * ubar(i, j1, kout));
      } else if (code == LBC_CLAMPED) ubar(i, jb, kout) = ubar_bry(i, jb);
      else if (code == LBC_GRADIENT) ubar(i, jb, kout) = ubar(i, j1, kout);
      else ubar(i, jb, kout) = p->gamma2 * ubar(i, j1, kout);
      if (mk) ubar(i, jb, kout) = ubar(i, jb, kout) * umask(i, jb);
    }
  }
}

/* v2dbc_tile, S/N edges (normal component) -- ROMS/Nonlinear/v2dbc_im.F:134-830: radiation (:138, :487), Flather (:216, :565) with
 * bry_val = BOUNDARY%vbar_south/north (no SSH_TIDES), clamped (:366, :715), gradient (:379, :728), closed
 * (:434, :783) */
void o_v2dbc(OARGS, int kout)
{
  ORACLE_PROLOGUE
  const int mk = p->masking;
  int know; double dt2d;
  o_know(p, s, &know, &dt2d);
  for (int side = 0; side < 2; side++) {
    if (!(side ? north_edge : south_edge)) continue;
    const int code = o_lbc(p, side ? LBS_NORTH : LBS_SOUTH, LBV_VBAR);
    const int jb = side ? Jend + 1 : Jstr, j1 = side ? Jend : Jstr + 1;     /* boundary v-row, first interior v-row */
    const int ja = side ? Jend : Jstr - 1, jc = side ? Jend + 1 : Jstr;     /* the two rho-rows around row jb */
    for (int i = Istr; i <= Iend; i++) {
      if (code == LBC_RADIATION) {
        const int j2 = side ? Jend - 1 : Jstr + 2;
        const double gL = vbar(i, j1, know) - vbar(i - 1, j1, know), gR = vbar(i + 1, j1, know) - vbar(i, j1, know);
        vbar(i, jb, kout) = o_radiate(vbar(i, jb, know), vbar(i, j1, know), vbar(i, j1, kout), vbar(i, j2, kout), gL, gR);
      } else if (code == LBC_FLATHER) {
        const double bry_val = vbar_bry(i, jb);
        const double cff = 1.0 / (0.5 * (h(i, ja) + zeta(i, ja, know) + h(i, jc) + zeta(i, jc, know)));
        const double Ce = sqrt(p->g * cff);
        if (side) vbar(i, jb, kout) = bry_val + Ce * (0.5 * (zeta(i, ja, know) + zeta(i, jc, know)) - zeta_bry(i, Jend + 1));
        else vbar(i, jb, kout) = bry_val - Ce * (0.5 * (zeta(i, ja, know) + zeta(i, jc, know)) - zeta_bry(i, Jstr - 1));
      } else if (code == LBC_CLAMPED) vbar(i, jb, kout) = vbar_bry(i, jb);
      else if (code == LBC_GRADIENT) vbar(i, jb, kout) = vbar(i, j1, kout);
      else vbar(i, jb, kout) = 0.0;
      if (mk && code != LBC_CLOSED) vbar(i, jb, kout) = vbar(i, jb, kout) * vmask(i, jb);
    }
  }
}

/* u3dbc_tile, S/N edges -- ROMS/Nonlinear/u3dbc_im.F:379-700 */
void o_u3dbc(OARGS, int nout)
{
  ORACLE_PROLOGUE
  const int mk = p->masking, nstp = s->nstp;
  for (int side = 0; side < 2; side++) {
    if (!(side ? north_edge : south_edge)) continue;
    const int code = o_lbc(p, side ? LBS_NORTH : LBS_SOUTH, LBV_U);
    const int jb = side ? Jend + 1 : Jstr - 1, j1 = side ? Jend : Jstr, j2 = side ? Jend - 1 : Jstr + 1;
    int Imin = IstrU, Imax = Iend;
    if (code == LBC_CLOSED) { Imin = EWperiodic ? IstrU : Istr; Imax = EWperiodic ? Iend : IendR; }
    for (int k = 1; k <= N; k++)
      for (int i = Imin; i <= Imax; i++) {
        if (code == LBC_RADIATION)
          u(i, jb, k, nout) = o_radiate(u(i, jb, k, nstp), u(i, j1, k, nstp), u(i, j1, k, nout), u(i, j2, k, nout),
                                        u(i, j1, k, nstp) - u(i - 1, j1, k, nstp), u(i + 1, j1, k, nstp) - u(i, j1, k, nstp));
        else if (code == LBC_CLAMPED) u(i, jb, k, nout) = u_bry(i, jb, k);
        else if (code == LBC_GRADIENT) u(i, jb, k, nout) = u(i, j1, k, nout);
        else u(i, jb, k, nout) = p->gamma2 * u(i, j1, k, nout);
        if (mk) u(i, jb, k, nout) = u(i, jb, k, nout) * umask(i, jb);
      }
  }
}

/* v3dbc_tile, S/N edges -- ROMS/Nonlinear/v3dbc_im.F:95-380 */
void o_v3dbc(OARGS, int nout)
{
  ORACLE_PROLOGUE
  const int mk = p->masking, nstp = s->nstp;
  for (int side = 0; side < 2; side++) {
    if (!(side ? north_edge : south_edge)) continue;
    const int code = o_lbc(p, side ? LBS_NORTH : LBS_SOUTH, LBV_V);
    const int jb = side ? Jend + 1 : Jstr, j1 = side ? Jend : Jstr + 1, j2 = side ? Jend - 1 : Jstr + 2;
    for (int k = 1; k <= N; k++)
      for (int i = Istr; i <= Iend; i++) {
        if (code == LBC_RADIATION)
          v(i, jb, k, nout) = o_radiate(v(i, jb, k, nstp), v(i, j1, k, nstp), v(i, j1, k, nout), v(i, j2, k, nout),
                                        v(i, j1, k, nstp) - v(i - 1, j1, k, nstp), v(i + 1, j1, k, nstp) - v(i, j1, k, nstp));
        else if (code == LBC_CLAMPED) v(i, jb, k, nout) = v_bry(i, jb, k);
        else if (code == LBC_GRADIENT) v(i, jb, k, nout) = v(i, j1, k, nout);
        else v(i, jb, k, nout) = 0.0;
        if (mk && code != LBC_CLOSED) v(i, jb, k, nout) = v(i, jb, k, nout) * vmask(i, jb);
      }
  }
}

/* t3dbc_tile, S/N edges -- ROMS/Nonlinear/t3dbc_im.F:362-630 (MASKING: the along-boundary differences of the
 * radiation condition are multiplied by umask, :370-379) */
void o_t3dbc(OARGS, int nout, int itrc)
{
  ORACLE_PROLOGUE
  const int mk = p->masking, nstp = s->nstp;
  for (int side = 0; side < 2; side++) {
    if (!(side ? north_edge : south_edge)) continue;
    const int code = o_lbc(p, side ? LBS_NORTH : LBS_SOUTH, LBV_T);
    const int jb = side ? Jend + 1 : Jstr - 1, j1 = side ? Jend : Jstr, j2 = side ? Jend - 1 : Jstr + 1;
    for (int k = 1; k <= N; k++)
      for (int i = Istr; i <= Iend; i++) {
        if (code == LBC_RADIATION) {
          double gL = t(i, j1, k, nstp, itrc) - t(i - 1, j1, k, nstp, itrc);
          double gR = t(i + 1, j1, k, nstp, itrc) - t(i, j1, k, nstp, itrc);
          if (mk) { gL = gL * umask(i, j1); gR = gR * umask(i + 1, j1); }
          t(i, jb, k, nout, itrc) = o_radiate(t(i, jb, k, nstp, itrc), t(i, j1, k, nstp, itrc), t(i, j1, k, nout, itrc),
                                              t(i, j2, k, nout, itrc), gL, gR);
        } else if (code == LBC_CLAMPED) t(i, jb, k, nout, itrc) = t_bry(i, jb, k, itrc);
        else t(i, jb, k, nout, itrc) = t(i, j1, k, nout, itrc);          /* gradient, closed */
        if (mk) t(i, jb, k, nout, itrc) = t(i, jb, k, nout, itrc) * rmask(i, jb);
      }
  }
}

/* one boundary-condition routine on its own (tests/test_ref_pinning.py pins each against the reference) */
int oracle_bc(OARGS, int kind, int nout, int itrc)
{
  if (o_check_lbc(b, p)) return 8;
  switch (kind) {
  case 1: o_zetabc(b, p, s, F, nout); break;
  case 2: o_u2dbc(b, p, s, F, nout); break;
  case 3: o_v2dbc(b, p, s, F, nout); break;
  case 4: o_u3dbc(b, p, s, F, nout); break;
  case 5: o_v3dbc(b, p, s, F, nout); break;
  case 6: o_t3dbc(b, p, s, F, nout, itrc); break;
  default: return 2;
  }
  return 0;
}

/* bc_w3d_tile (gradient walls + periodic wrap) -- ROMS/Nonlinear/bc_3d.F:588 */
void o_bc_w3d(const roms_bounds_t *b, double *A)
{
  const int LBi = b->LBi, LBj = b->LBj, N = b->N;
  const long ni = b->UBi - b->LBi + 1, nij = ni * (b->UBj - b->LBj + 1);
#define A3(i,j,k) A[(long)((i) - LBi) + (long)((j) - LBj) * ni + (long)(k) * nij]
  if (!b->NSperiodic) {
    if (b->north_edge)
      for (int k = 0; k <= N; k++)
        for (int i = b->Istr; i <= b->Iend; i++) A3(i, b->Jend + 1, k) = A3(i, b->Jend, k);
    if (b->south_edge)
      for (int k = 0; k <= N; k++)
        for (int i = b->Istr; i <= b->Iend; i++) A3(i, b->Jstr - 1, k) = A3(i, b->Jstr, k);
  }
#undef A3
  o_exchange3d(b, GT_R, N + 1, A);
}

/* set_massflux_tile -- ROMS/Nonlinear/set_massflux.F:73-188 */
int oracle_set_massflux(OARGS)
{
  ORACLE_PROLOGUE
  (void)p;
  const int nrhs = s->nrhs;
  for (int k = 1; k <= N; k++) {
    for (int j = JstrT; j <= JendT; j++)
      for (int i = IstrP; i <= IendT; i++)
        Huon(i, j, k) = 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k)) * u(i, j, k, nrhs) * on_u(i, j);
    for (int j = JstrP; j <= JendT; j++)
      for (int i = IstrT; i <= IendT; i++)
        Hvom(i, j, k) = 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k)) * v(i, j, k, nrhs) * om_v(i, j);
  }
  o_exchange3d(b, GT_U, N, F->Huon);
  o_exchange3d(b, GT_V, N, F->Hvom);
  return 0;
}

/* omega_tile -- ROMS/Nonlinear/omega.F:73-229 (no point sources) */
int oracle_omega(OARGS)
{
  ORACLE_PROLOGUE
  (void)p; (void)s;
  double *wrk = walloc(nis);
  for (int j = Jstr; j <= Jend; j++) {
    for (int i = Istr; i <= Iend; i++) W(i, j, 0) = 0.0;
    for (int k = 1; k <= N; k++)
      for (int i = Istr; i <= Iend; i++)
        W(i, j, k) = W(i, j, k - 1) -
                     (Huon(i + 1, j, k) - Huon(i, j, k) +
                      Hvom(i, j + 1, k) - Hvom(i, j, k));
    for (int i = Istr; i <= Iend; i++)
      wrk[i - IminS] = W(i, j, N) / (z_w(i, j, N) - z_w(i, j, 0));
    for (int k = N - 1; k >= 1; k--)
      for (int i = Istr; i <= Iend; i++)
        W(i, j, k) = W(i, j, k) - wrk[i - IminS] * (z_w(i, j, k) - z_w(i, j, 0));
    for (int i = Istr; i <= Iend; i++) W(i, j, N) = 0.0;
  }
  free(wrk);
  o_bc_w3d(b, F->W);
  return 0;
}

/* set_zeta_tile -- ROMS/Nonlinear/set_zeta.F:59-129 */
int oracle_set_zeta(OARGS)
{
  ORACLE_PROLOGUE
  (void)p; (void)s;
  for (int j = JstrR; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) {
      zeta(i, j, 1) = Zt_avg1(i, j);
      zeta(i, j, 2) = Zt_avg1(i, j);
    }
  o_exchange2d(b, GT_R, F->zeta);
  o_exchange2d(b, GT_R, F->zeta + nij);
  return 0;
}

/* set_depth_tile -- ROMS/Nonlinear/set_depth.F:82-300 */
int oracle_set_depth(OARGS)
{
  ORACLE_PROLOGUE
  (void)s;
  const double hc = p->hc;
  if (p->Vtransform == 1) {
    for (int j = JstrT; j <= JendT; j++) {
      for (int i = IstrT; i <= IendT; i++) z_w(i, j, 0) = -h(i, j);
      for (int k = 1; k <= N; k++) {
        double cff_r = hc * (p->sc_r[k] - p->Cs_r[k]);
        double cff_w = hc * (p->sc_w[k] - p->Cs_w[k]);
        double cff1_r = p->Cs_r[k], cff1_w = p->Cs_w[k];
        for (int i = IstrT; i <= IendT; i++) {
          double hwater = h(i, j);
          double hinv = 1.0 / hwater;
          double z_w0 = cff_w + cff1_w * hwater;
          z_w(i, j, k) = z_w0 + Zt_avg1(i, j) * (1.0 + z_w0 * hinv);
          double z_r0 = cff_r + cff1_r * hwater;
          z_r(i, j, k) = z_r0 + Zt_avg1(i, j) * (1.0 + z_r0 * hinv);
          Hz(i, j, k) = z_w(i, j, k) - z_w(i, j, k - 1);
        }
      }
    }
  } else {
    for (int j = JstrT; j <= JendT; j++) {
      for (int i = IstrT; i <= IendT; i++) z_w(i, j, 0) = -h(i, j);
      for (int k = 1; k <= N; k++) {
        double cff_r = hc * p->sc_r[k], cff_w = hc * p->sc_w[k];
        double cff1_r = p->Cs_r[k], cff1_w = p->Cs_w[k];
        for (int i = IstrT; i <= IendT; i++) {
          double hwater = h(i, j);
          double hinv = 1.0 / (hc + hwater);
          double cff2_r = (cff_r + cff1_r * hwater) * hinv;
          double cff2_w = (cff_w + cff1_w * hwater) * hinv;
          z_w(i, j, k) = Zt_avg1(i, j) + (Zt_avg1(i, j) + hwater) * cff2_w;
          z_r(i, j, k) = Zt_avg1(i, j) + (Zt_avg1(i, j) + hwater) * cff2_r;
          Hz(i, j, k) = z_w(i, j, k) - z_w(i, j, k - 1);
        }
      }
    }
  }
  o_exchange2d(b, GT_R, F->h);
  o_exchange3d(b, GT_R, N + 1, F->z_w);
  o_exchange3d(b, GT_R, N, F->z_r);
  o_exchange3d(b, GT_R, N, F->Hz);
  return 0;
}
