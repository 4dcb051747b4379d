/*
 * oracle_rho_eos.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * rho_eos_tile: nonlinear (ROMS/Nonlinear/rho_eos.F:111-575) and linear
 * (:576-889) equation of state with VAR_RHO_2D, BV_FREQUENCY and
 * EOS_TDERIVATIVE outputs as compiled for BENCHMARK.  Pinned against the
 * flang build of the reference file (oracle/_ref).
 */
#include "oracle.h"
#include "roms_eoscoef.h"

int oracle_rho_eos(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs;
  const int itemp = 1, isalt = 2;
  const double g = p->g, rho0 = p->rho0;
  double cff, cff1, cff2;
  if (!p->nonlin_eos) {
    const double R0 = p->R0, T0 = p->T0, S0 = p->S0, Tcoef = p->Tcoef, Scoef = p->Scoef;
    for (int j = JstrT; j <= JendT; j++) {
      for (int k = 1; k <= N; k++)
        for (int i = IstrT; i <= IendT; i++) {
          rho(i, j, k) = R0 - R0 * Tcoef * (t(i, j, k, nrhs, itemp) - T0);
          if (p->salinity) rho(i, j, k) = rho(i, j, k) + R0 * Scoef * (t(i, j, k, nrhs, isalt) - S0);
          rho(i, j, k) = rho(i, j, k) - 1000.0;
          if (p->masking) rho(i, j, k) = rho(i, j, k) * rmask(i, j);                 /* MASKING, rho_eos.F:717 */
          pden(i, j, k) = rho(i, j, k);
        }
      for (int i = IstrT; i <= IendT; i++) {
        cff1 = rho(i, j, N) * Hz(i, j, N);
        rhoS(i, j) = 0.5 * cff1 * Hz(i, j, N);
        rhoA(i, j) = cff1;
      }
      for (int k = N - 1; k >= 1; k--)
        for (int i = IstrT; i <= IendT; i++) {
          cff1 = rho(i, j, k) * Hz(i, j, k);
          rhoS(i, j) = rhoS(i, j) + Hz(i, j, k) * (rhoA(i, j) + 0.5 * cff1);
          rhoA(i, j) = rhoA(i, j) + cff1;
        }
      cff2 = 1.0 / rho0;
      for (int i = IstrT; i <= IendT; i++) {
        cff1 = 1.0 / (z_w(i, j, N) - z_w(i, j, 0));
        rhoA(i, j) = cff2 * cff1 * rhoA(i, j);
        rhoS(i, j) = 2.0 * cff1 * cff1 * cff2 * rhoS(i, j);
      }
    }
    o_exchange3d(b, GT_R, N, F->rho);
    o_exchange3d(b, GT_R, N, F->pden);
    o_exchange2d(b, GT_R, F->rhoA);
    o_exchange2d(b, GT_R, F->rhoS);
    return 0;
  }
  const long nk = nis * (N + 1);
  double *DbulkDS_ = walloc(nk), *DbulkDT_ = walloc(nk), *Dden1DS_ = walloc(nk), *Dden1DT_ = walloc(nk);
  double *Scof_ = walloc(nk), *Tcof_ = walloc(nk), *wrk_ = walloc(nk), *bulk_ = walloc(nk);
  double *bulk0_ = walloc(nk), *bulk1_ = walloc(nk), *bulk2_ = walloc(nk), *den_ = walloc(nk), *den1_ = walloc(nk);
#define DbulkDS(i,k) DbulkDS_[WSK(i,k)]
#define DbulkDT(i,k) DbulkDT_[WSK(i,k)]
#define Dden1DS(i,k) Dden1DS_[WSK(i,k)]
#define Dden1DT(i,k) Dden1DT_[WSK(i,k)]
#define Scof(i,k) Scof_[WSK(i,k)]
#define Tcof(i,k) Tcof_[WSK(i,k)]
#define wrk(i,k) wrk_[WSK(i,k)]
#define bulk(i,k) bulk_[WSK(i,k)]
#define bulk0(i,k) bulk0_[WSK(i,k)]
#define bulk1(i,k) bulk1_[WSK(i,k)]
#define bulk2(i,k) bulk2_[WSK(i,k)]
#define den(i,k) den_[WSK(i,k)]
#define den1(i,k) den1_[WSK(i,k)]
  double C[10], dCdT[10];
  for (int j = JstrT; j <= JendT; j++) {
    for (int k = 1; k <= N; k++)
      for (int i = IstrT; i <= IendT; i++) {
        const double Tt = MAX(-2.0, t(i, j, k, nrhs, itemp));
        const double Ts = p->salinity ? MAX(0.0, t(i, j, k, nrhs, isalt)) : 0.0;
        const double sqrtTs = sqrt(Ts);
        const double Tp = z_r(i, j, k);
        const double Tpr10 = 0.1 * Tp;
        C[0] = EOS_Q00 + Tt * (EOS_Q01 + Tt * (EOS_Q02 + Tt * (EOS_Q03 + Tt * (EOS_Q04 + Tt * EOS_Q05))));
        C[1] = EOS_U00 + Tt * (EOS_U01 + Tt * (EOS_U02 + Tt * (EOS_U03 + Tt * EOS_U04)));
        C[2] = EOS_V00 + Tt * (EOS_V01 + Tt * EOS_V02);
        dCdT[0] = EOS_Q01 + Tt * (2.0 * EOS_Q02 + Tt * (3.0 * EOS_Q03 + Tt * (4.0 * EOS_Q04 + Tt * 5.0 * EOS_Q05)));
        dCdT[1] = EOS_U01 + Tt * (2.0 * EOS_U02 + Tt * (3.0 * EOS_U03 + Tt * 4.0 * EOS_U04));
        dCdT[2] = EOS_V01 + Tt * 2.0 * EOS_V02;
        den1(i, k) = C[0] + Ts * (C[1] + sqrtTs * C[2] + Ts * EOS_W00);
        Dden1DS(i, k) = C[1] + 1.5 * C[2] * sqrtTs + 2.0 * EOS_W00 * Ts;
        Dden1DT(i, k) = dCdT[0] + Ts * (dCdT[1] + sqrtTs * dCdT[2]);
        C[3] = EOS_A00 + Tt * (EOS_A01 + Tt * (EOS_A02 + Tt * (EOS_A03 + Tt * EOS_A04)));
        C[4] = EOS_B00 + Tt * (EOS_B01 + Tt * (EOS_B02 + Tt * EOS_B03));
        C[5] = EOS_D00 + Tt * (EOS_D01 + Tt * EOS_D02);
        C[6] = EOS_E00 + Tt * (EOS_E01 + Tt * (EOS_E02 + Tt * EOS_E03));
        C[7] = EOS_F00 + Tt * (EOS_F01 + Tt * EOS_F02);
        C[8] = EOS_G01 + Tt * (EOS_G02 + Tt * EOS_G03);
        C[9] = EOS_H00 + Tt * (EOS_H01 + Tt * EOS_H02);
        dCdT[3] = EOS_A01 + Tt * (2.0 * EOS_A02 + Tt * (3.0 * EOS_A03 + Tt * 4.0 * EOS_A04));
        dCdT[4] = EOS_B01 + Tt * (2.0 * EOS_B02 + Tt * 3.0 * EOS_B03);
        dCdT[5] = EOS_D01 + Tt * 2.0 * EOS_D02;
        dCdT[6] = EOS_E01 + Tt * (2.0 * EOS_E02 + Tt * 3.0 * EOS_E03);
        dCdT[7] = EOS_F01 + Tt * 2.0 * EOS_F02;
        dCdT[8] = EOS_G02 + Tt * 2.0 * EOS_G03;
        dCdT[9] = EOS_H01 + Tt * 2.0 * EOS_H02;
        bulk0(i, k) = C[3] + Ts * (C[4] + sqrtTs * C[5]);
        bulk1(i, k) = C[6] + Ts * (C[7] + sqrtTs * EOS_G00);
        bulk2(i, k) = C[8] + Ts * C[9];
        bulk(i, k) = bulk0(i, k) - Tp * (bulk1(i, k) - Tp * bulk2(i, k));
        DbulkDS(i, k) = C[4] + sqrtTs * 1.5 * C[5] - Tp * (C[7] + sqrtTs * 1.5 * EOS_G00 - Tp * C[9]);
        DbulkDT(i, k) = dCdT[3] + Ts * (dCdT[4] + sqrtTs * dCdT[5]) -
                        Tp * (dCdT[6] + Ts * dCdT[7] - Tp * (dCdT[8] + Ts * dCdT[9]));
        cff = 1.0 / (bulk(i, k) + Tpr10);
        den(i, k) = den1(i, k) * bulk(i, k) * cff;
        den(i, k) = den(i, k) - 1000.0;
        if (p->masking) den(i, k) = den(i, k) * rmask(i, j);                         /* MASKING, rho_eos.F:356 */
      }
    for (int i = IstrT; i <= IendT; i++) {
      cff1 = den(i, N) * Hz(i, j, N);
      rhoS(i, j) = 0.5 * cff1 * Hz(i, j, N);
      rhoA(i, j) = cff1;
    }
    for (int k = N - 1; k >= 1; k--)
      for (int i = IstrT; i <= IendT; i++) {
        cff1 = den(i, k) * Hz(i, j, k);
        rhoS(i, j) = rhoS(i, j) + Hz(i, j, k) * (rhoA(i, j) + 0.5 * cff1);
        rhoA(i, j) = rhoA(i, j) + cff1;
      }
    cff2 = 1.0 / rho0;
    for (int i = IstrT; i <= IendT; i++) {
      cff1 = 1.0 / (z_w(i, j, N) - z_w(i, j, 0));
      rhoA(i, j) = cff2 * cff1 * rhoA(i, j);
      rhoS(i, j) = 2.0 * cff1 * cff1 * cff2 * rhoS(i, j);
    }
    for (int k = 1; k <= N - 1; k++)
      for (int i = IstrT; i <= IendT; i++) {
        const double bulk_up = bulk0(i, k + 1) - z_w(i, j, k) * (bulk1(i, k + 1) - bulk2(i, k + 1) * z_w(i, j, k));
        const double bulk_dn = bulk0(i, k) - z_w(i, j, k) * (bulk1(i, k) - bulk2(i, k) * z_w(i, j, k));
        cff1 = 1.0 / (bulk_up + 0.1 * z_w(i, j, k));
        cff2 = 1.0 / (bulk_dn + 0.1 * z_w(i, j, k));
        const double den_up = cff1 * (den1(i, k + 1) * bulk_up);
        const double den_dn = cff2 * (den1(i, k) * bulk_dn);
        bvf(i, j, k) = -g * (den_up - den_dn) / (0.5 * (den_up + den_dn) * (z_r(i, j, k + 1) - z_r(i, j, k)));
      }
    for (int i = IstrT; i <= IendT; i++) {
      bvf(i, j, 0) = 0.0;
      bvf(i, j, N) = 0.0;
    }
    for (int k = N; k <= N; k++) {
      for (int i = IstrT; i <= IendT; i++) {
        const double Tpr10 = 0.1 * z_r(i, j, k);
        cff = bulk(i, k) + Tpr10;
        cff1 = Tpr10 * den1(i, k);
        cff2 = bulk(i, k) * cff;
        wrk(i, k) = (den(i, k) + 1000.0) * cff * cff;
        Tcof(i, k) = -(DbulkDT(i, k) * cff1 + Dden1DT(i, k) * cff2);
        Scof(i, k) = (DbulkDS(i, k) * cff1 + Dden1DS(i, k) * cff2);
      }
      for (int i = IstrT; i <= IendT; i++) {
        cff = 1.0 / wrk(i, N);
        alpha(i, j) = cff * Tcof(i, N);
        beta(i, j) = cff * Scof(i, N);
      }
    }
    for (int k = 1; k <= N; k++)
      for (int i = IstrT; i <= IendT; i++) {
        rho(i, j, k) = den(i, k);
        pden(i, j, k) = (den1(i, k) - 1000.0);
        if (p->masking) pden(i, j, k) = pden(i, j, k) * rmask(i, j);                 /* MASKING, rho_eos.F:478 */
      }
  }
  o_exchange3d(b, GT_R, N, F->rho);
  o_exchange3d(b, GT_R, N, F->pden);
  o_exchange2d(b, GT_R, F->alpha);
  o_exchange2d(b, GT_R, F->beta);
  o_exchange2d(b, GT_R, F->rhoA);
  o_exchange2d(b, GT_R, F->rhoS);
  o_exchange3d(b, GT_R, N + 1, F->bvf);
  free(DbulkDS_); free(DbulkDT_); free(Dden1DS_); free(Dden1DT_); free(Scof_); free(Tcof_);
  free(wrk_); free(bulk_); free(bulk0_); free(bulk1_); free(bulk2_); free(den_); free(den1_);
  return 0;
}
