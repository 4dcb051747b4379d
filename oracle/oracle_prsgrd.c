/*
 * oracle_prsgrd.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * prsgrd32_tile: density-Jacobian pressure gradient with harmonic-mean
 * limited cubic reconstruction (ROMS/Nonlinear/prsgrd32.h:106-423, DJ_GRADPS).
 * Pinned against the flang build of the reference file (oracle/_ref).
 */
#include "oracle.h"

int oracle_prsgrd(OARGS)
{
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
