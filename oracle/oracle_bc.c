/*
 * oracle_bc.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The six lateral boundary-condition routines of the path on all four edges and the corners:
 *   zetabc_tile  ROMS/Nonlinear/zetabc.F:48     u2dbc_tile  u2dbc_im.F:51     v2dbc_tile  v2dbc_im.F:52
 *   u3dbc_tile   u3dbc_im.F:50                  v3dbc_tile  v3dbc_im.F:50     t3dbc_tile  t3dbc_im.F:50
 * The reference writes every edge out separately; its western / eastern blocks are the transposes of the southern /
 * northern ones (pm <-> pn, umask <-> vmask, i <-> j), so one routine per variable walks an edge in "edge
 * coordinates": B = the boundary point, P1 / P2 = the first / second point inward along the normal, +-T = the
 * neighbours along the edge.  Conditions: closed, gradient, clamped, implicit upstream radiation (no nudging, no
 * RADIATION_2D), Chapman implicit (zeta), Flather (normal barotropic velocity) with the Chapman-type rule for the
 * tangential one.  Pinned against the reference's own routines (tests/test_ref_pinning.py).
 */
#include "oracle.h"
#include <math.h>

/* The time level `know` and the step dt2d of the 2-D boundary conditions (zetabc.F:91-101, v2dbc_im.F:116-126) */
static void o_know(const roms_params_t *p, const roms_step_idx_t *s, int *know, double *dt2d)
{
  if (s->iif == 1) { *know = s->krhs; *dt2d = p->dtfast; }
  else if (s->predictor_2d_step) { *know = s->krhs; *dt2d = 2.0 * p->dtfast; }
  else { *know = s->kstp; *dt2d = p->dtfast; }
}

/* Implicit upstream radiation with Cx (or Ce) of the tangential direction = 0 (no RADIATION_2D), e.g.
 * zetabc.F:123-160: xb_old = X(B) at the old level; x1_old, x1 = X(P1) at the old / new level; x2 = X(P2) at the new
 * level; gL, gR = the two along-edge differences of X(old) at P1 on either side of the point. */
static double o_radiate(double xb_old, double x1_old, double x1, double x2, double gL, double gR, int *inward,
                        int rad2d, double gLb, double gRb)
{
  const double eps = 1.0E-20;
  double dXdt = x1_old - x1;
  const double dXdn = x1 - x2;
  *inward = (dXdt * dXdn) < 0.0;                 /* selects the nudging time scale, e.g. t3dbc_im.F:138-146 */
  if ((dXdt * dXdn) < 0.0) dXdt = 0.0;
  const double dXds = ((dXdt * (gL + gR)) > 0.0) ? gL : gR;
  const double cff = MAX(dXdn * dXdn + dXds * dXds, eps);
  const double Cn = dXdt * dXdn;
  /* RADIATION_2D: the tangential phase speed with the upstream along-edge difference of the boundary row itself
   * (gLb, gRb), e.g. zetabc.F:141-160 */
  const double Ct = rad2d ? MIN(cff, MAX(dXdt * dXds, -cff)) : 0.0;
  if (rad2d) return (cff * xb_old + Cn * x1 - MAX(Ct, 0.0) * gLb - MIN(Ct, 0.0) * gRb) / (cff + Cn);
  return (cff * xb_old + Cn * x1) / (cff + Cn);
}

typedef struct {
  int we;            /* 1: western / eastern edge (normal along i), 0: southern / northern */
  int hi;            /* 1: eastern / northern edge */
  long sn, st;       /* flat-index step inward along the normal, step along the edge */
  int a0, a1;        /* range of the along-edge index */
  int bi, bj;        /* boundary point of along-index a: (we ? bi : a, we ? a : bj) */
} Edge;

/* gtype = GT_R / GT_U / GT_V; closed = the closed condition (its tangential range differs, u2dbc_im.F:960-975) */
static int edge_of(const roms_bounds_t *b, int side, int gtype, int closed, Edge *e)
{
  const long ni = b->UBi - b->LBi + 1;
  e->we = side <= LBS_EAST;
  e->hi = side == LBS_EAST || side == LBS_NORTH;
  if (!(side == LBS_WEST ? b->west_edge : side == LBS_EAST ? b->east_edge : side == LBS_SOUTH ? b->south_edge : b->north_edge))
    return 0;
  if (e->we ? b->EWperiodic : b->NSperiodic) return 0;
  e->sn = (e->we ? 1 : ni) * (e->hi ? -1 : 1);
  e->st = e->we ? ni : 1;
  if (e->we) {
    e->bj = 0;
    e->bi = e->hi ? b->Iend + 1 : (gtype == GT_U ? b->Istr : b->Istr - 1);
    e->a0 = b->Jstr; e->a1 = b->Jend;
    if (gtype == GT_V) {                      /* tangential component, v2dbc_im.F:812-1120 */
      e->a0 = b->JstrV;
      if (closed) { e->a0 = b->NSperiodic ? b->JstrV : b->Jstr; e->a1 = b->NSperiodic ? b->Jend : b->JendR; }
    }
  } else {
    e->bi = 0;
    e->bj = e->hi ? b->Jend + 1 : (gtype == GT_V ? b->Jstr : b->Jstr - 1);
    e->a0 = b->Istr; e->a1 = b->Iend;
    if (gtype == GT_U) {                      /* tangential component, u2dbc_im.F:829-1140 */
      e->a0 = b->IstrU;
      if (closed) { e->a0 = b->EWperiodic ? b->IstrU : b->Istr; e->a1 = b->EWperiodic ? b->Iend : b->IendR; }
    }
  }
  return 1;
}

/* One variable on one edge.  X = plane(s) of the level written, O = the same variable at the level the condition
 * compares with (know / nstp), D = boundary data, Z = zeta(know), Zb = zeta_bry; nk planes of stride nij. */
static void bc_edge(const roms_bounds_t *b, const roms_params_t *p, const roms_fields_t *F, int side, int var, int code,
                    double *X, const double *O, const double *D, const double *Z, const double *Zb, const double *Zn, int nk,
                    double dt2d, const double *T)
{
  const int LBi = b->LBi, LBj = b->LBj;
  const long ni = b->UBi - b->LBi + 1, nij = ni * (b->UBj - b->LBj + 1);
  const int gtype = (var == LBV_UBAR || var == LBV_U) ? GT_U : (var == LBV_VBAR || var == LBV_V) ? GT_V : GT_R;
  Edge e;
  if (!edge_of(b, side, gtype, code == LBC_CLOSED, &e)) return;
  const int normal = (gtype == GT_U && e.we) || (gtype == GT_V && !e.we);
  const int mk = p->masking;
  const double *mask = gtype == GT_U ? F->umask : gtype == GT_V ? F->vmask : F->rmask;
  const double *gmask = e.we ? F->vmask : F->umask;          /* mask of the along-edge differences (rho-type) */
  const double *pmn = e.we ? F->pm : F->pn;
  for (int k = 0; k < nk; k++) {
    double *Xk = X + (long)k * nij;
    const double *Ok = O ? O + (long)k * nij : NULL;
    for (int a = e.a0; a <= e.a1; a++) {
      const long B = I2(e.we ? e.bi : a, e.we ? a : e.bj), P1 = B + e.sn, P2 = P1 + e.sn;
      double x;
      if (code == LBC_RADIATION || code == LBC_RADIATION_NUDGING) {
        double gL = Ok[P1] - Ok[P1 - e.st], gR = Ok[P1 + e.st] - Ok[P1];
        if (mk && gtype == GT_R) { gL = gL * gmask[P1]; gR = gR * gmask[P1 + e.st]; }       /* zetabc.F:112-120, t3dbc_im.F */
        /* zetabc.F:424 -- on the SOUTHERN edge the free surface takes its normal difference towards the boundary
         * row (the other edges, :126, :275, :573, and every other variable look into the interior) */
        const long Q2 = (var == LBV_ZETA && side == LBS_SOUTH) ? B : P2;
        int inward;
        double gLb = Ok[B] - Ok[B - e.st], gRb = Ok[B + e.st] - Ok[B];
        if (mk && gtype == GT_R) { gLb = gLb * gmask[B]; gRb = gRb * gmask[B + e.st]; }
        /* zetabc.F:455-456 -- and the same edge takes the along-edge differences of the first INSIDE row there */
        if (var == LBV_ZETA && side == LBS_SOUTH) { gLb = gL; gRb = gR; }
        x = o_radiate(Ok[B], Ok[P1], Xk[P1], Xk[Q2], gL, gR, &inward, p->radiation_2d, gLb, gRb);
        if (code == LBC_RADIATION_NUDGING) {                    /* explicit nudging, zetabc.F:128-135/:162-166 ... */
          double tau = inward ? p->obc_in[side][var] : p->obc_out[side][var];
          tau = tau * (nk == 1 && var <= LBV_VBAR ? dt2d : p->dt);
          x = x + tau * (D[B + (long)k * nij] - Ok[B]);
        }
      } else if (code == LBC_CLAMPED) {
        x = D[B + (long)k * nij];
      } else if (code == LBC_CHAPMAN_IMPLICIT) {                /* zetabc.F:193-220, :342, :491, :640 */
        const double cff = dt2d * pmn[P1];
        const double cff1 = sqrt(p->g * (F->h[P1] + O[P1]));
        const double Cn = cff * cff1;
        const double cff2 = 1.0 / (1.0 + Cn);
        x = cff2 * (O[B] + Cn * Xk[P1]);
      } else if (code == LBC_CHAPMAN_EXPLICIT) {                /* zetabc.F:175-190, :324, :473, :622 */
        const double cff = dt2d * pmn[P1];
        const double cff1 = sqrt(p->g * (F->h[P1] + O[P1]));
        const double Cn = cff * cff1;
        x = (1.0 - Cn) * O[B] + Cn * O[P1];
      } else if (code == LBC_SHCHEPETKIN && normal) {           /* u2dbc_im.F:288-362, :636-710; v2dbc_im.F:290-364, :639-713 */
        /* (Mason et al., 2010; no SSH_TIDES: bry_val = the boundary data.)  qi / qo = the rho-points inside / outside
         * the boundary velocity point; Zn = zeta at the level being written */
        const long lo = B - (e.we ? 1 : ni), qi = e.hi ? lo : B, qo = e.hi ? B : lo;
        const double Co = 1.0 / (2.0 + sqrt(2.0));              /* mod_scalars.F:4175 */
        const double bry_val = D[B];
        /* WET_DRY: the total depth instead of the resting one, u2dbc_im.F:331-340, :679-688; v2dbc_im.F:333, :682 */
        const double cff = p->wet_dry ? 0.5 * (F->h[lo] + Z[lo] + F->h[B] + Z[B]) : 0.5 * (F->h[lo] + F->h[B]);
        const double cff1 = sqrt(p->g / cff);
        const double Cn = dt2d * cff1 * cff * 0.5 * (pmn[lo] + pmn[B]);
        double Zx = (0.5 + Cn) * Z[qi] + (0.5 - Cn) * Z[qo];
        if (Cn > Co) {
          const double cff2 = (1.0 - Co / Cn) * (1.0 - Co / Cn);
          const double cff3 = Zn[qi] + Cn * Z[qo] - (1.0 + Cn) * Z[qi];
          Zx = Zx + cff2 * cff3;
        }
        x = e.hi ? 0.5 * ((1.0 - Cn) * O[B] + Cn * O[P1] + bry_val + cff1 * (Zx - Zb[qo]))
                 : 0.5 * ((1.0 - Cn) * O[B] + Cn * O[P1] + bry_val - cff1 * (Zx - Zb[qo]));
      } else if (code == LBC_REDUCED && normal) {               /* u2dbc_im.F:392-432, :740-780; v2dbc_im.F:394-436, :743-785 */
        /* T = the other barotropic component at know; acquire: boundary data of the free surface exist on this side
         * (inp_decode.F:1620-1655, no FSOBC_REDUCED) */
        const long un = e.we ? 1 : ni, lo = B - un, qi = e.hi ? lo : B, qo = e.hi ? B : lo;
        const int zc = o_lbc(p, side, LBV_ZETA), uc = o_lbc(p, side, LBV_UBAR), vc = o_lbc(p, side, LBV_VBAR);
        const int acquire = zc == LBC_CLAMPED || zc == LBC_RADIATION_NUDGING || uc == LBC_FLATHER || uc == LBC_SHCHEPETKIN ||
                            vc == LBC_FLATHER || vc == LBC_SHCHEPETKIN;
        double bry_pgr, bry_cor = 0.0;
        if (acquire) bry_pgr = e.hi ? -p->g * (Zb[qo] - Z[qi]) * 0.5 * pmn[qi] : -p->g * (Z[qi] - Zb[qo]) * 0.5 * pmn[qi];
        else bry_pgr = -p->g * (Z[B] - Z[lo]) * 0.5 * (pmn[lo] + pmn[B]);
        if (p->uv_cor) {
          const long ut = e.we ? ni : 1;                        /* step along the edge */
          bry_cor = 0.125 * (T[lo] + T[lo + ut] + T[B] + T[B + ut]) * (F->f[lo] + F->f[B]);
          if (!e.we) bry_cor = -bry_cor;
        }
        const double cff = 1.0 / (0.5 * (F->h[lo] + Z[lo] + F->h[B] + Z[B]));
        const double bry_str = cff * ((e.we ? F->sustr : F->svstr)[B] - (e.we ? F->bustr : F->bvstr)[B]);
        x = O[B] + dt2d * (bry_pgr + bry_cor + bry_str);
      } else if (code == LBC_FLATHER && normal) {               /* u2dbc_im.F:214-300, v2dbc_im.F:216-286 */
        /* the two rho-points around the boundary velocity point, lower index first: u(i,j) lies between
         * rho(i-1,j) and rho(i,j), v(i,j) between rho(i,j-1) and rho(i,j) */
        const long qa = B - (e.we ? 1 : ni), qc = B;
        const double bry_val = D[B];
        const double cff = 1.0 / (0.5 * (F->h[qa] + Z[qa] + F->h[qc] + Z[qc]));
        const double Cn = sqrt(p->g * cff);
        const double zb = Zb[e.hi ? qc : qa];                   /* zeta_west(j) = the rho boundary point */
        if (p->atm_press && p->press_compensate) {             /* ATM_PRESS && PRESS_COMPENSATE, u2dbc_im.F:264-272, :612-620 */
          const double OneAtm = 1013.25, fac = 100.0 / (p->g * p->rho0);
          const double zm = 0.5 * (Z[qa] + Z[qc] + fac * (F->Pair[qa] + F->Pair[qc] - 2.0 * OneAtm));
          x = e.hi ? bry_val + Cn * (zm - zb) : bry_val - Cn * (zm - zb);
        } else
        x = e.hi ? bry_val + Cn * (0.5 * (Z[qa] + Z[qc]) - zb) : bry_val - Cn * (0.5 * (Z[qa] + Z[qc]) - zb);
      } else if (code == LBC_FLATHER || code == LBC_SHCHEPETKIN || code == LBC_REDUCED) {   /* tangential: u2dbc_im.F:912-932, v2dbc_im.F:886-906 */
        const double cff = dt2d * 0.5 * (pmn[P1 - e.st] + pmn[P1]);
        const double cff1 = sqrt(p->g * 0.5 * (F->h[P1 - e.st] + Z[P1 - e.st] + F->h[P1] + Z[P1]));
        const double Cn = cff * cff1;
        const double cff2 = 1.0 / (1.0 + Cn);
        x = cff2 * (O[B] + Cn * Xk[P1]);
      } else if (code == LBC_GRADIENT) {
        x = Xk[P1];
      } else {                                                  /* closed */
        x = normal ? 0.0 : (gtype == GT_R ? Xk[P1] : p->gamma2 * Xk[P1]);
      }
      if (mk && !(normal && code == LBC_CLOSED)) x = x * mask[B];
      /* WET_DRY, 3-D momentum: the wet/dry mask after every land/sea-mask product (u3dbc_im.F:174 ... :681,
       * v3dbc_im.F:174 ... :681) -- but for u on a southern gradient edge, whose block tests a symbol that no header
       * defines (u3dbc_im.F:496) */
      if (p->wet_dry && (var == LBV_U || var == LBV_V) && !(normal && code == LBC_CLOSED) &&
          !(var == LBV_U && side == LBS_SOUTH && code == LBC_GRADIENT))
        x = x * (var == LBV_U ? F->umask_wet : F->vmask_wet)[B];
      Xk[B] = x;
    }
  }
}

/* corners, e.g. zetabc.F:699-731: the mean of the two neighbouring boundary points, when neither direction is
 * periodic and the tile holds the corner */
static void bc_corners(const roms_bounds_t *b, int gtype, double *X, int nk)
{
  if (b->EWperiodic || b->NSperiodic) return;
  const int LBi = b->LBi, LBj = b->LBj;
  const long ni = b->UBi - b->LBi + 1, nij = ni * (b->UBj - b->LBj + 1);
  const int iw = gtype == GT_U ? b->Istr : b->Istr - 1, ie = b->Iend + 1;
  const int js = gtype == GT_V ? b->Jstr : b->Jstr - 1, jn = b->Jend + 1;
  for (int k = 0; k < nk; k++) {
    double *Xk = X + (long)k * nij;
    if (b->south_edge && b->west_edge) Xk[I2(iw, js)] = 0.5 * (Xk[I2(iw + 1, js)] + Xk[I2(iw, js + 1)]);
    if (b->south_edge && b->east_edge) Xk[I2(ie, js)] = 0.5 * (Xk[I2(ie - 1, js)] + Xk[I2(ie, js + 1)]);
    if (b->north_edge && b->west_edge) Xk[I2(iw, jn)] = 0.5 * (Xk[I2(iw, jn - 1)] + Xk[I2(iw + 1, jn)]);
    if (b->north_edge && b->east_edge) Xk[I2(ie, jn)] = 0.5 * (Xk[I2(ie, jn - 1)] + Xk[I2(ie - 1, jn)]);
  }
}

/* bc_r2d_tile / bc_u2d_tile / bc_v2d_tile (bc_2d.F:45/184/386) and bc_r3d / bc_u3d / bc_v3d / bc_w3d_tile (bc_3d.F):
 * the generic conditions the reference applies to derived fields (stresses, boundary-layer depth, omega, mixing
 * coefficients ...) -- on an edge where variable `lbv` is closed: zero normal velocity, gamma2 * (inner value)
 * for the tangential one (times the mask under MASKING), zero gradient for rho-type fields; on any other physical
 * edge: zero gradient; then the corners.  No periodic wrap here (the caller exchanges). */
void o_bc_generic(const roms_bounds_t *b, const roms_params_t *p, const roms_fields_t *F, int gtype, int lbv, double *A,
                  int nk)
{
  const int LBi = b->LBi, LBj = b->LBj;
  const long ni = b->UBi - b->LBi + 1, nij = ni * (b->UBj - b->LBj + 1);
  const double *mask = gtype == GT_U ? F->umask : F->vmask;
  for (int side = LBS_WEST; side <= LBS_NORTH; side++) {
    const int closed = o_lbc(p, side, lbv) == LBC_CLOSED;
    Edge e;
    if (!edge_of(b, side, gtype, closed, &e)) continue;
    const int normal = (gtype == GT_U && e.we) || (gtype == GT_V && !e.we);
    for (int k = 0; k < nk; k++) {
      double *Ak = A + (long)k * nij;
      for (int a = e.a0; a <= e.a1; a++) {
        const long B = I2(e.we ? e.bi : a, e.we ? a : e.bj), P1 = B + e.sn;
        if (closed && normal) Ak[B] = 0.0;
        else if (closed && gtype != GT_R) {
          Ak[B] = p->gamma2 * Ak[P1];
          if (p->masking) Ak[B] = Ak[B] * mask[B];
        } else Ak[B] = Ak[P1];
      }
    }
  }
  bc_corners(b, gtype, A, nk);
}

/* order of the edges as in the reference: west, east, south, north, then the corners */
static const int SIDES[4] = {LBS_WEST, LBS_EAST, LBS_SOUTH, LBS_NORTH};

void o_zetabc(OARGS, int kout)
{
  ORACLE_PROLOGUE
  int know; double dt2d;
  o_know(p, s, &know, &dt2d);
  for (int q = 0; q < 4; q++)
    bc_edge(b, p, F, SIDES[q], LBV_ZETA, o_lbc(p, SIDES[q], LBV_ZETA), &zeta(LBi, LBj, kout), &zeta(LBi, LBj, know),
            F->zeta_bry, &zeta(LBi, LBj, know), F->zeta_bry, NULL, 1, dt2d, NULL);
  bc_corners(b, GT_R, &zeta(LBi, LBj, kout), 1);
  if (p->wet_dry) {
    /* WET_DRY, zetabc.F:733-827: the water level of every boundary point (and corner) stays above the bed */
    const double eps = 1.0E-20, cff = p->Dcrit - eps;
#define RAISE(i, j) if (zeta(i, j, kout) <= (p->Dcrit - h(i, j))) zeta(i, j, kout) = cff - h(i, j)
    if (!EWperiodic) {
      if (west_edge) for (int j = Jstr; j <= Jend; j++) { RAISE(Istr - 1, j); }
      if (east_edge) for (int j = Jstr; j <= Jend; j++) { RAISE(Iend + 1, j); }
    }
    if (!NSperiodic) {
      if (south_edge) for (int i = Istr; i <= Iend; i++) { RAISE(i, Jstr - 1); }
      if (north_edge) for (int i = Istr; i <= Iend; i++) { RAISE(i, Jend + 1); }
    }
    if (!(EWperiodic || NSperiodic)) {
      if (south_edge && west_edge) { RAISE(Istr - 1, Jstr - 1); }
      if (south_edge && east_edge) { RAISE(Iend + 1, Jstr - 1); }
      if (north_edge && west_edge) { RAISE(Istr - 1, Jend + 1); }
      if (north_edge && east_edge) { RAISE(Iend + 1, Jend + 1); }
    }
#undef RAISE
  }
}

void o_u2dbc(OARGS, int kout)
{
  ORACLE_PROLOGUE
  int know; double dt2d;
  o_know(p, s, &know, &dt2d);
  for (int q = 0; q < 4; q++)
    bc_edge(b, p, F, SIDES[q], LBV_UBAR, o_lbc(p, SIDES[q], LBV_UBAR), &ubar(LBi, LBj, kout), &ubar(LBi, LBj, know),
            F->ubar_bry, &zeta(LBi, LBj, know), F->zeta_bry, &zeta(LBi, LBj, kout), 1, dt2d, &vbar(LBi, LBj, know));
  bc_corners(b, GT_U, &ubar(LBi, LBj, kout), 1);
  if (p->wet_dry) {
    /* WET_DRY, u2dbc_im.F:1176-1293 (ranges as written: IstrU on the southern edge, Istr on the northern one) */
#define WETBC(i, j) ubar(i, j, kout) = ubar(i, j, kout) * o_wet_factor(umask_wet(i, j), ubar(i, j, kout))
    if (!EWperiodic) {
      if (west_edge) for (int j = Jstr; j <= Jend; j++) { WETBC(Istr, j); }
      if (east_edge) for (int j = Jstr; j <= Jend; j++) { WETBC(Iend + 1, j); }
    }
    if (!NSperiodic) {
      if (south_edge) for (int i = IstrU; i <= Iend; i++) { WETBC(i, Jstr - 1); }
      if (north_edge) for (int i = Istr; i <= Iend; i++) { WETBC(i, Jend + 1); }
    }
    if (!(EWperiodic || NSperiodic)) {
      if (south_edge && west_edge) { WETBC(Istr, Jstr - 1); }
      if (south_edge && east_edge) { WETBC(Iend + 1, Jstr - 1); }
      if (north_edge && west_edge) { WETBC(Istr, Jend + 1); }
      if (north_edge && east_edge) { WETBC(Iend + 1, Jend + 1); }
    }
#undef WETBC
  }
}

void o_v2dbc(OARGS, int kout)
{
  ORACLE_PROLOGUE
  int know; double dt2d;
  o_know(p, s, &know, &dt2d);
  for (int q = 0; q < 4; q++)
    bc_edge(b, p, F, SIDES[q], LBV_VBAR, o_lbc(p, SIDES[q], LBV_VBAR), &vbar(LBi, LBj, kout), &vbar(LBi, LBj, know),
            F->vbar_bry, &zeta(LBi, LBj, know), F->zeta_bry, &zeta(LBi, LBj, kout), 1, dt2d, &ubar(LBi, LBj, know));
  bc_corners(b, GT_V, &vbar(LBi, LBj, kout), 1);
  if (p->wet_dry) {
    /* WET_DRY, v2dbc_im.F:1169-1287.  As written, the western edge takes its factor from the boundary point
     * (Istr-1,j) and applies it to the first INTERIOR point (Istr,j), :1180-1185. */
#define WETBC(i, j) vbar(i, j, kout) = vbar(i, j, kout) * o_wet_factor(vmask_wet(i, j), vbar(i, j, kout))
    if (!EWperiodic) {
      if (west_edge)
        for (int j = JstrV; j <= Jend; j++)
          vbar(Istr, j, kout) = vbar(Istr, j, kout) * o_wet_factor(vmask_wet(Istr - 1, j), vbar(Istr - 1, j, kout));
      if (east_edge) for (int j = JstrV; j <= Jend; j++) { WETBC(Iend + 1, j); }
    }
    if (!NSperiodic) {
      if (south_edge) for (int i = Istr; i <= Iend; i++) { WETBC(i, Jstr); }
      if (north_edge) for (int i = Istr; i <= Iend; i++) { WETBC(i, Jend + 1); }
    }
    if (!(EWperiodic || NSperiodic)) {
      if (south_edge && west_edge) { WETBC(Istr - 1, Jstr); }
      if (south_edge && east_edge) { WETBC(Iend + 1, Jstr); }
      if (north_edge && west_edge) { WETBC(Istr - 1, Jend + 1); }
      if (north_edge && east_edge) { WETBC(Iend + 1, Jend + 1); }
    }
#undef WETBC
  }
}

void o_u3dbc(OARGS, int nout)
{
  ORACLE_PROLOGUE
  for (int q = 0; q < 4; q++)
    bc_edge(b, p, F, SIDES[q], LBV_U, o_lbc(p, SIDES[q], LBV_U), &u(LBi, LBj, 1, nout), &u(LBi, LBj, 1, s->nstp),
            F->u_bry, NULL, NULL, NULL, N, 0.0, NULL);
  bc_corners(b, GT_U, &u(LBi, LBj, 1, nout), N);
}

void o_v3dbc(OARGS, int nout)
{
  ORACLE_PROLOGUE
  for (int q = 0; q < 4; q++)
    bc_edge(b, p, F, SIDES[q], LBV_V, o_lbc(p, SIDES[q], LBV_V), &v(LBi, LBj, 1, nout), &v(LBi, LBj, 1, s->nstp),
            F->v_bry, NULL, NULL, NULL, N, 0.0, NULL);
  bc_corners(b, GT_V, &v(LBi, LBj, 1, nout), N);
}

void o_t3dbc(OARGS, int nout, int itrc)
{
  ORACLE_PROLOGUE
  for (int q = 0; q < 4; q++)
    bc_edge(b, p, F, SIDES[q], LBV_T, o_lbc(p, SIDES[q], LBV_T), &t(LBi, LBj, 1, nout, itrc), &t(LBi, LBj, 1, s->nstp, itrc),
            F->t_bry + (long)(itrc - 1) * n3r, NULL, NULL, NULL, N, 0.0, NULL);
  bc_corners(b, GT_R, &t(LBi, LBj, 1, nout, itrc), N);
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
