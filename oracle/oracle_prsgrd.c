/*
 * oracle_prsgrd.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * prsgrd32_tile: density-Jacobian pressure gradient with harmonic-mean
 * limited cubic reconstruction (ROMS/Nonlinear/prsgrd32.h:106-423, DJ_GRADPS).
 * Pinned against the flang build of the reference file (oracle/_ref).
 */
#include "oracle.h"

/* prsgrd31_tile -- ROMS/Nonlinear/prsgrd31.h:97-364: the standard density Jacobian (the reference's default when no
 * pressure-gradient option is defined, prsgrd.F:24-25), optionally the weighted Jacobian of Song 1998 (WJ_GRADP).
 * RHO_SURF is always defined (globaldefs.h:130); no ATM_PRESS, no TIDE_GENERATING_FORCES, no WET_DRY.  The routine
 * has no MASKING blocks.  Pinned against builds of the reference without DJ_GRADPS (oracle/_ref/UPWELLING_PG31,
 * UPWELLING_WJ). */
static int oracle_prsgrd31(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, wj = p->pgf == PGF_WJ_GRADP;
  const double g = p->g, rho0 = p->rho0;
  const double fac1 = 0.5 * g / rho0, fac2 = 1000.0 * g / rho0, fac3 = 0.25 * g / rho0;
  double *phi = walloc(nis);
  for (int j = Jstr; j <= Jend; j++) {
    for (int dir = 0; dir < 2; dir++) {               /* 0: XI-component (:196-268), 1: ETA-component (:276-352) */
      if (dir == 1 && j < JstrV) continue;
      const int di = dir ? 0 : 1, dj = dir ? 1 : 0;
      for (int i = (dir ? Istr : IstrU); i <= Iend; i++) {
        const int im = i - di, jm = j - dj;
        double cff1 = z_w(i, j, N) - z_r(i, j, N) + z_w(im, jm, N) - z_r(im, jm, N);
        double ph = fac1 * (rho(i, j, N) - rho(im, jm, N)) * cff1;
        if (p->atm_press) ph = ph + (100.0 / p->rho0) * (F->Pair[I2(i, j)] - F->Pair[I2(im, jm)]);   /* ATM_PRESS, prsgrd31.h:213-215, :294-296 */
        ph = ph + (fac2 + fac1 * (rho(i, j, N) + rho(im, jm, N))) * (z_w(i, j, N) - z_w(im, jm, N));
        phi[i - IminS] = ph;
        const double r = -0.5 * (Hz(i, j, N) + Hz(im, jm, N)) * ph * (dir ? om_v(i, j) : on_u(i, j));
        if (dir) rv(i, j, N, nrhs) = r; else ru(i, j, N, nrhs) = r;
      }
      for (int k = N - 1; k >= 1; k--)
        for (int i = (dir ? Istr : IstrU); i <= Iend; i++) {
          const int im = i - di, jm = j - dj;
          double cff1, cff2, cff3, cff4;
          if (wj) {
            cff1 = 1.0 / ((z_r(i, j, k + 1) - z_r(i, j, k)) * (z_r(im, jm, k + 1) - z_r(im, jm, k)));
            cff2 = z_r(i, j, k) - z_r(im, jm, k) + z_r(i, j, k + 1) - z_r(im, jm, k + 1);
            cff3 = z_r(i, j, k + 1) - z_r(i, j, k) - z_r(im, jm, k + 1) + z_r(im, jm, k);
            const double gamma = 0.125 * cff1 * cff2 * cff3;
            cff1 = (1.0 + gamma) * (rho(i, j, k + 1) - rho(im, jm, k + 1)) + (1.0 - gamma) * (rho(i, j, k) - rho(im, jm, k));
            cff2 = rho(i, j, k + 1) + rho(im, jm, k + 1) - rho(i, j, k) - rho(im, jm, k);
            cff3 = z_r(i, j, k + 1) + z_r(im, jm, k + 1) - z_r(i, j, k) - z_r(im, jm, k);
            cff4 = (1.0 + gamma) * (z_r(i, j, k + 1) - z_r(im, jm, k + 1)) + (1.0 - gamma) * (z_r(i, j, k) - z_r(im, jm, k));
          } else {
            cff1 = rho(i, j, k + 1) - rho(im, jm, k + 1) + rho(i, j, k) - rho(im, jm, k);
            cff2 = rho(i, j, k + 1) + rho(im, jm, k + 1) - rho(i, j, k) - rho(im, jm, k);
            cff3 = z_r(i, j, k + 1) + z_r(im, jm, k + 1) - z_r(i, j, k) - z_r(im, jm, k);
            cff4 = z_r(i, j, k + 1) - z_r(im, jm, k + 1) + z_r(i, j, k) - z_r(im, jm, k);
          }
          phi[i - IminS] = phi[i - IminS] + fac3 * (cff1 * cff3 - cff2 * cff4);
          const double r = -0.5 * (Hz(i, j, k) + Hz(im, jm, k)) * phi[i - IminS] * (dir ? om_v(i, j) : on_u(i, j));
          if (dir) rv(i, j, k, nrhs) = r; else ru(i, j, k, nrhs) = r;
        }
    }
  }
  free(phi);
  return 0;
}

/* prsgrd40_tile -- prsgrd40.h:170-268 (PJ_GRADP: finite-volume pressure Jacobian, Lin 1997).  No ATM_PRESS, no
 * TIDE_GENERATING_FORCES; the file has no MASKING blocks.  Pinned against oracle/_ref/<APP>_PJ. */
static int oracle_prsgrd40(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs;
  const double g = p->g, rho0 = p->rho0;
  double *FC_ = walloc(nis * (N + 1)), *FX_ = walloc(nis * njs * N), *P_ = walloc(nis * njs * (N + 1));
#define FCk(i,k) FC_[(long)((i) - IminS) + (long)(k) * nis]
#define FXk(i,j,k) FX_[WS3(i,j,k)]
#define Pk(i,j,k) P_[WS2(i,j) + (long)(k) * nis * njs]
  for (int j = JstrV - 1; j <= Jend; j++) {
    for (int i = IstrU - 1; i <= Iend; i++) {
      Pk(i, j, N) = 0.0;
      if (p->atm_press) Pk(i, j, N) = Pk(i, j, N) + (100.0 / g) * (F->Pair[I2(i, j)] - 1013.25);   /* ATM_PRESS, prsgrd40.h:187-196 */
    }
    for (int k = N; k >= 1; k--)
      for (int i = IstrU - 1; i <= Iend; i++) {
        Pk(i, j, k - 1) = Pk(i, j, k) + Hz(i, j, k) * rho(i, j, k);
        FXk(i, j, k) = 0.5 * Hz(i, j, k) * (Pk(i, j, k) + Pk(i, j, k - 1));
      }
    if (j >= Jstr) {
      for (int i = IstrU; i <= Iend; i++) FCk(i, N) = 0.0;
      const double cff = 0.5 * g, cff1 = g / rho0;
      for (int k = N; k >= 1; k--)
        for (int i = IstrU; i <= Iend; i++) {
          const double dh = z_w(i, j, k - 1) - z_w(i - 1, j, k - 1);
          FCk(i, k - 1) = 0.5 * dh * (Pk(i, j, k - 1) + Pk(i - 1, j, k - 1));
          ru(i, j, k, nrhs) = (cff * (Hz(i - 1, j, k) + Hz(i, j, k)) * (z_w(i - 1, j, N) - z_w(i, j, N)) +
                               cff1 * (FXk(i - 1, j, k) - FXk(i, j, k) + FCk(i, k) - FCk(i, k - 1))) * on_u(i, j);
        }
    }
    if (j >= JstrV) {
      for (int i = Istr; i <= Iend; i++) FCk(i, N) = 0.0;
      const double cff = 0.5 * g, cff1 = g / rho0;
      for (int k = N; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) {
          const double dh = z_w(i, j, k - 1) - z_w(i, j - 1, k - 1);
          FCk(i, k - 1) = 0.5 * dh * (Pk(i, j, k - 1) + Pk(i, j - 1, k - 1));
          rv(i, j, k, nrhs) = (cff * (Hz(i, j - 1, k) + Hz(i, j, k)) * (z_w(i, j - 1, N) - z_w(i, j, N)) +
                               cff1 * (FXk(i, j - 1, k) - FXk(i, j, k) + FCk(i, k) - FCk(i, k - 1))) * om_v(i, j);
        }
    }
  }
  free(FC_); free(FX_); free(P_);
#undef FCk
#undef FXk
#undef Pk
  return 0;
}

static int o_prsgrd_any(OARGS);

/* WET_DRY: every variant multiplies the term it has just stored by the wet/dry mask of the face (prsgrd32.h:346,
 * :410; prsgrd31.h:223 ... :350; prsgrd40.h:229, :259) -- one pass over the same ranges after the variant */
int oracle_prsgrd(OARGS)
{
  /* PJ_GRADP with WET_DRY does not compile in the reference (prsgrd40.h:98-100 passes umask_wet, vmask_wet to a
   * routine that never declares them): refused, as the library does */
  if (p->wet_dry && p->pgf == PGF_PJ_GRADP) return 8;
  const int rc = o_prsgrd_any(b, p, s, F);
  if (rc || !p->wet_dry) return rc;
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs;
  for (int k = 1; k <= N; k++) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) ru(i, j, k, nrhs) = ru(i, j, k, nrhs) * umask_wet(i, j);
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) rv(i, j, k, nrhs) = rv(i, j, k, nrhs) * vmask_wet(i, j);
  }
  return 0;
}

static int o_prsgrd_any(OARGS)
{
  if (p->pgf == PGF_PJ_GRADP) return oracle_prsgrd40(b, p, s, F);
  if (p->pgf != PGF_DJ_GRADPS) {
    if (p->pgf != PGF_STANDARD && p->pgf != PGF_WJ_GRADP) return 8;
    return oracle_prsgrd31(b, p, s, F);
  }
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs;
  const double OneFifth = 0.2, OneTwelfth = 1.0 / 12.0, eps = 1.0E-10;
  const double g = p->g, rho0 = p->rho0;
  const double GRho = g / rho0, HalfGRho = 0.5 * GRho;
  double cff, cff1, cff2;
  double *P_ = walloc(nis * njs * N);
  double *dR_ = walloc(nis * (N + 1)), *dZ_ = walloc(nis * (N + 1));
  double *FC_ = walloc(nis * njs), *aux_ = walloc(nis * njs);
  double *dRx_ = walloc(nis * njs), *dZx_ = walloc(nis * njs);
#define P(i,j,k) P_[WS3(i,j,k)]
#define dR(i,k)  dR_[WSK(i,k)]
#define dZ(i,k)  dZ_[WSK(i,k)]
#define FC(i,j)  FC_[WS2(i,j)]
#define aux(i,j) aux_[WS2(i,j)]
#define dRx(i,j) dRx_[WS2(i,j)]
#define dZx(i,j) dZx_[WS2(i,j)]

  /* prsgrd32.h:223-291 */
  for (int j = JstrV - 1; j <= Jend; j++) {
    for (int k = 1; k <= N - 1; k++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        dR(i, k) = rho(i, j, k + 1) - rho(i, j, k);
        dZ(i, k) = z_r(i, j, k + 1) - z_r(i, j, k);
      }
    for (int i = IstrU - 1; i <= Iend; i++) {
      dR(i, N) = dR(i, N - 1);
      dZ(i, N) = dZ(i, N - 1);
      dR(i, 0) = dR(i, 1);
      dZ(i, 0) = dZ(i, 1);
    }
    for (int k = N; k >= 1; k--)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = 2.0 * dR(i, k) * dR(i, k - 1);
        if (cff > eps) dR(i, k) = cff / (dR(i, k) + dR(i, k - 1));
        else dR(i, k) = 0.0;
        dZ(i, k) = 2.0 * dZ(i, k) * dZ(i, k - 1) / (dZ(i, k) + dZ(i, k - 1));
      }
    for (int i = IstrU - 1; i <= Iend; i++) {
      cff1 = 1.0 / (z_r(i, j, N) - z_r(i, j, N - 1));
      cff2 = 0.5 * (rho(i, j, N) - rho(i, j, N - 1)) * (z_w(i, j, N) - z_r(i, j, N)) * cff1;
      if (p->atm_press)                                                  /* ATM_PRESS, prsgrd32.h:229-232, :264-266 */
        P(i, j, N) = g * z_w(i, j, N) + (100.0 / rho0) * (F->Pair[I2(i, j)] - 1013.25) +
                     GRho * (rho(i, j, N) + cff2) * (z_w(i, j, N) - z_r(i, j, N));
      else
      P(i, j, N) = g * z_w(i, j, N) + GRho * (rho(i, j, N) + cff2) * (z_w(i, j, N) - z_r(i, j, N));
    }
    for (int k = N - 1; k >= 1; k--)
      for (int i = IstrU - 1; i <= Iend; i++) {
        P(i, j, k) = P(i, j, k + 1) +
                     HalfGRho * ((rho(i, j, k + 1) + rho(i, j, k)) * (z_r(i, j, k + 1) - z_r(i, j, k)) -
                                 OneFifth *
                                 ((dR(i, k + 1) - dR(i, k)) *
                                  (z_r(i, j, k + 1) - z_r(i, j, k) - OneTwelfth * (dZ(i, k + 1) + dZ(i, k))) -
                                  (dZ(i, k + 1) - dZ(i, k)) *
                                  (rho(i, j, k + 1) - rho(i, j, k) - OneTwelfth * (dR(i, k + 1) + dR(i, k)))));
      }
  }
  /* XI-component, prsgrd32.h:293-355 */
  for (int k = N; k >= 1; k--) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend + 1; i++) {
        aux(i, j) = z_r(i, j, k) - z_r(i - 1, j, k);
        if (p->masking) aux(i, j) = aux(i, j) * umask(i, j);                         /* MASKING, prsgrd32.h:300 */
        FC(i, j) = rho(i, j, k) - rho(i - 1, j, k);
        if (p->masking) FC(i, j) = FC(i, j) * umask(i, j);                           /* :304 */
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = 2.0 * aux(i, j) * aux(i + 1, j);
        if (cff > eps) { cff1 = 1.0 / (aux(i, j) + aux(i + 1, j)); dZx(i, j) = cff * cff1; }
        else dZx(i, j) = 0.0;
        cff1 = 2.0 * FC(i, j) * FC(i + 1, j);
        if (cff1 > eps) { cff2 = 1.0 / (FC(i, j) + FC(i + 1, j)); dRx(i, j) = cff1 * cff2; }
        else dRx(i, j) = 0.0;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        ru(i, j, k, nrhs) = on_u(i, j) * 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k)) *
                            (P(i - 1, j, k) - P(i, j, k) -
                             HalfGRho *
                             ((rho(i, j, k) + rho(i - 1, j, k)) * (z_r(i, j, k) - z_r(i - 1, j, k)) -
                              OneFifth *
                              ((dRx(i, j) - dRx(i - 1, j)) *
                               (z_r(i, j, k) - z_r(i - 1, j, k) - OneTwelfth * (dZx(i, j) + dZx(i - 1, j))) -
                               (dZx(i, j) - dZx(i - 1, j)) *
                               (rho(i, j, k) - rho(i - 1, j, k) - OneTwelfth * (dRx(i, j) + dRx(i - 1, j))))));
      }
  }
  /* ETA-component, prsgrd32.h:357-420 */
  for (int k = N; k >= 1; k--) {
    for (int j = JstrV - 1; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++) {
        aux(i, j) = z_r(i, j, k) - z_r(i, j - 1, k);
        if (p->masking) aux(i, j) = aux(i, j) * vmask(i, j);                         /* MASKING, prsgrd32.h:364 */
        FC(i, j) = rho(i, j, k) - rho(i, j - 1, k);
        if (p->masking) FC(i, j) = FC(i, j) * vmask(i, j);                           /* :368 */
      }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = 2.0 * aux(i, j) * aux(i, j + 1);
        if (cff > eps) { cff1 = 1.0 / (aux(i, j) + aux(i, j + 1)); dZx(i, j) = cff * cff1; }
        else dZx(i, j) = 0.0;
        cff1 = 2.0 * FC(i, j) * FC(i, j + 1);
        if (cff1 > eps) { cff2 = 1.0 / (FC(i, j) + FC(i, j + 1)); dRx(i, j) = cff1 * cff2; }
        else dRx(i, j) = 0.0;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        rv(i, j, k, nrhs) = om_v(i, j) * 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k)) *
                            (P(i, j - 1, k) - P(i, j, k) -
                             HalfGRho *
                             ((rho(i, j, k) + rho(i, j - 1, k)) * (z_r(i, j, k) - z_r(i, j - 1, k)) -
                              OneFifth *
                              ((dRx(i, j) - dRx(i, j - 1)) *
                               (z_r(i, j, k) - z_r(i, j - 1, k) - OneTwelfth * (dZx(i, j) + dZx(i, j - 1))) -
                               (dZx(i, j) - dZx(i, j - 1)) *
                               (rho(i, j, k) - rho(i, j - 1, k) - OneTwelfth * (dRx(i, j) + dRx(i, j - 1))))));
      }
  }
  free(P_); free(dR_); free(dZ_); free(FC_); free(aux_); free(dRx_); free(dZx_);
  return 0;
}
