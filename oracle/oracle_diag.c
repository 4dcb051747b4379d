/*
 * oracle_diag.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * The two diagnostics main3d runs every step next to the hot path:
 *   wvelocity_tile  ROMS/Nonlinear/wvelocity.F:61  (SOLVE3D, no masking)
 *   diag_tile       ROMS/Nonlinear/diag.F:80       (SOLVE3D branch; the tile-local sums and maxima of
 *                                                   :190-290 -- the reduction over tiles and the printing
 *                                                   are the caller's)
 * Loop order and operation order as in the reference.
 */
#include "oracle.h"

int oracle_wvelocity(OARGS)
{
  ORACLE_PROLOGUE
  const int Ninp = s->nstp;                       /* main3d.F:475: CALL wvelocity (ng, tile, nstp(ng)) */
  if (N < 3) return 8;
  /* :119-135 */
  o_exchange2d(b, GT_U, F->DU_avg1);
  o_exchange2d(b, GT_V, F->DV_avg1);
  double *vert = (double *)malloc(sizeof(double) * nis * njs * N);
  double *wrk = (double *)malloc(sizeof(double) * nis * njs);
  if (!vert || !wrk) { free(vert); free(wrk); return 8; }
#define S2(i,j)     ((long)((i) - IminS) + (long)((j) - JminS) * nis)
#define VERT(i,j,k) vert[S2(i,j) + (long)((k) - 1) * nis * njs]
  /* :137-172 */
  for (int k = 1; k <= N; k++) {
    for (int j = Jstr; j <= Jend; j++) {
      for (int i = Istr; i <= Iend + 1; i++)
        wrk[S2(i, j)] = u(i, j, k, Ninp) * (z_r(i, j, k) - z_r(i - 1, j, k)) * (pm(i - 1, j) + pm(i, j));
      for (int i = Istr; i <= Iend; i++)
        VERT(i, j, k) = 0.25 * (wrk[S2(i, j)] + wrk[S2(i + 1, j)]);
    }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++)
        wrk[S2(i, j)] = v(i, j, k, Ninp) * (z_r(i, j, k) - z_r(i, j - 1, k)) * (pn(i, j - 1) + pn(i, j));
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++)
        VERT(i, j, k) = VERT(i, j, k) + 0.25 * (wrk[S2(i, j)] + wrk[S2(i, j + 1)]);
  }
  /* :174-236 */
  const double cff1 = 3.0 / 8.0, cff2 = 3.0 / 4.0, cff3 = 1.0 / 8.0, cff4 = 9.0 / 16.0, cff5 = 1.0 / 16.0;
  for (int j = Jstr; j <= Jend; j++) {
    for (int i = Istr; i <= Iend; i++)
      wrk[S2(i, j)] = (DU_avg1(i, j) - DU_avg1(i + 1, j) + DV_avg1(i, j) - DV_avg1(i, j + 1)) /
                      (z_w(i, j, N) - z_w(i, j, 0));
    for (int i = Istr; i <= Iend; i++) {
      const double slope = (z_r(i, j, 1) - z_w(i, j, 0)) / (z_r(i, j, 2) - z_r(i, j, 1));
      wvel(i, j, 0) = cff1 * (VERT(i, j, 1) - slope * (VERT(i, j, 2) - VERT(i, j, 1))) +
                      cff2 * VERT(i, j, 1) - cff3 * VERT(i, j, 2);
      wvel(i, j, 1) = pm(i, j) * pn(i, j) * (W(i, j, 1) + wrk[S2(i, j)] * (z_w(i, j, 1) - z_w(i, j, 0))) +
                      cff1 * VERT(i, j, 1) + cff2 * VERT(i, j, 2) - cff3 * VERT(i, j, 3);
    }
    for (int k = 2; k <= N - 2; k++)
      for (int i = Istr; i <= Iend; i++)
        wvel(i, j, k) = pm(i, j) * pn(i, j) * (W(i, j, k) + wrk[S2(i, j)] * (z_w(i, j, k) - z_w(i, j, 0))) +
                        cff4 * (VERT(i, j, k) + VERT(i, j, k + 1)) - cff5 * (VERT(i, j, k - 1) + VERT(i, j, k + 2));
    for (int i = Istr; i <= Iend; i++) {
      const double slope = (z_w(i, j, N) - z_r(i, j, N)) / (z_r(i, j, N) - z_r(i, j, N - 1));
      wvel(i, j, N) = pm(i, j) * pn(i, j) * wrk[S2(i, j)] * (z_w(i, j, N) - z_w(i, j, 0)) +
                      cff1 * (VERT(i, j, N) + slope * (VERT(i, j, N) - VERT(i, j, N - 1))) +
                      cff2 * VERT(i, j, N) - cff3 * VERT(i, j, N - 1);
      wvel(i, j, N - 1) = pm(i, j) * pn(i, j) * (W(i, j, N - 1) + wrk[S2(i, j)] * (z_w(i, j, N - 1) - z_w(i, j, 0))) +
                          cff1 * VERT(i, j, N) + cff2 * VERT(i, j, N - 1) - cff3 * VERT(i, j, N - 2);
    }
  }
  free(vert);
  free(wrk);
  /* :237-250 */
  o_bc_w3d(b, F->wvel);
  return 0;
#undef VERT
}

int oracle_diag(OARGS, double *out)
{
  ORACLE_PROLOGUE
  const int idia = s->nstp;                       /* SOLVE3D: idia = nstp, diag.F:182 */
  const double g = p->g, rho0 = p->rho0, dt = p->dt, spval = 1.0E+37;
  double *ke2d = (double *)calloc((size_t)(nis * njs), sizeof(double));
  double *pe2d = (double *)calloc((size_t)(nis * njs), sizeof(double));
  if (!ke2d || !pe2d) { free(ke2d); free(pe2d); return 8; }
  double my_max_C = 0.0, my_max_Cu = 0.0, my_max_Cv = 0.0, my_max_Cw = 0.0, my_maxspeed = 0.0, my_maxrho = -spval;
  int my_max_Ci = 0, my_max_Cj = 0, my_max_Ck = 0;
  /* :199-240 */
  for (int j = Jstr; j <= Jend; j++) {
    for (int i = Istr; i <= Iend; i++) {
      ke2d[S2(i, j)] = 0.0;
      pe2d[S2(i, j)] = 0.5 * g * z_w(i, j, N) * z_w(i, j, N);
    }
    const double cff = g / rho0;
    for (int k = N; k >= 1; k--)
      for (int i = Istr; i <= Iend; i++) {
        const double u2v2 = u(i, j, k, idia) * u(i, j, k, idia) + u(i + 1, j, k, idia) * u(i + 1, j, k, idia) +
                            v(i, j, k, idia) * v(i, j, k, idia) + v(i, j + 1, k, idia) * v(i, j + 1, k, idia);
        ke2d[S2(i, j)] = ke2d[S2(i, j)] + Hz(i, j, k) * 0.25 * u2v2;
        pe2d[S2(i, j)] = pe2d[S2(i, j)] + cff * Hz(i, j, k) * (rho(i, j, k) + 1000.0) * (z_r(i, j, k) - z_w(i, j, 0));
        const double my_Cu = 0.5 * fabs(u(i, j, k, idia) + u(i + 1, j, k, idia)) * dt * pm(i, j);
        const double my_Cv = 0.5 * fabs(v(i, j, k, idia) + v(i, j + 1, k, idia)) * dt * pn(i, j);
        const double my_Cw = 0.5 * fabs(wvel(i, j, k - 1) + wvel(i, j, k)) * dt / Hz(i, j, k);
        const double my_C = my_Cu + my_Cv + my_Cw;
        if (my_C > my_max_C) {
          my_max_C = my_C; my_max_Cu = my_Cu; my_max_Cv = my_Cv; my_max_Cw = my_Cw;
          my_max_Ci = i; my_max_Cj = j; my_max_Ck = k;
        }
        my_maxspeed = MAX(my_maxspeed, sqrt(0.5 * u2v2));
        my_maxrho = MAX(my_maxrho, rho(i, j, k));
      }
  }
  /* :262-290: j collapsed first, then i */
  for (int i = Istr; i <= Iend; i++) {
    pe2d[S2(i, Jend + 1)] = 0.0;
    pe2d[S2(i, Jstr - 1)] = 0.0;
    ke2d[S2(i, Jstr - 1)] = 0.0;
  }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      pe2d[S2(i, Jend + 1)] = pe2d[S2(i, Jend + 1)] + omn(i, j) * (z_w(i, j, N) - z_w(i, j, 0));
      pe2d[S2(i, Jstr - 1)] = pe2d[S2(i, Jstr - 1)] + omn(i, j) * pe2d[S2(i, j)];
      ke2d[S2(i, Jstr - 1)] = ke2d[S2(i, Jstr - 1)] + omn(i, j) * ke2d[S2(i, j)];
    }
  double my_volume = 0.0, my_avgpe = 0.0, my_avgke = 0.0;
  for (int i = Istr; i <= Iend; i++) {
    my_volume = my_volume + pe2d[S2(i, Jend + 1)];
    my_avgpe = my_avgpe + pe2d[S2(i, Jstr - 1)];
    my_avgke = my_avgke + ke2d[S2(i, Jstr - 1)];
  }
  free(ke2d);
  free(pe2d);
  out[0] = my_volume; out[1] = my_avgke; out[2] = my_avgpe; out[3] = my_maxspeed; out[4] = my_maxrho;
  out[5] = my_max_C; out[6] = my_max_Cu; out[7] = my_max_Cv; out[8] = my_max_Cw;
  out[9] = (double)my_max_Ci; out[10] = (double)my_max_Cj; out[11] = (double)my_max_Ck;
  return 0;
#undef S2
}
