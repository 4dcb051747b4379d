/*
 * oracle_pre_step3d.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * pre_step3d_tile: predictor step for tracers (t(n+1/2) -> t(:,:,:,3,:)) and
 * start of the corrector for t, u, v (ROMS/Nonlinear/pre_step3d.F:123-1156).
 * Parity unpinned: the reference file cannot be compiled here (mod_sources).
 */
#include "oracle.h"

int oracle_pre_step3d(OARGS)
{
  ORACLE_PROLOGUE
  if (o_src_check(p)) return 8;
  if (o_check_lbc(b, p)) return 8;
  const int nrhs = s->nrhs, nstp = s->nstp, nnew = s->nnew;
  const int iic = s->iic, ntfirst = s->ntfirst;
  const int itemp = 1;
  const double dt = p->dt, lambda = p->lambda;
  const double eps = 1.0E-16;
  double cff, cff1, cff2, cff3, cff4, Gamma;
  for (int itrc = 1; itrc <= NT; itrc++) {
    int ha = p->Hadv[itrc - 1], va = p->Vadv[itrc - 1];
    if (ha == ADV_SPLINES) return 8;
    if (va == ADV_U3) return 8;
    /* the reference lets H and V differ; MPDATA and HSIMT are restated as H+V pairs only */
    if (ha == ADV_HSIMT && va != ADV_HSIMT) return 8;
    if ((ha == ADV_MPDATA) != (va == ADV_MPDATA)) return 8;
  }
  double *CF_ = walloc(nis * (N + 1)), *DC_ = walloc(nis * (N + 1)), *FC_ = walloc(nis * (N + 1));
  double *swdk_ = walloc(nis * njs * (N + 1));
  double *FE_ = walloc(nis * njs), *FX_ = walloc(nis * njs), *curv_ = walloc(nis * njs), *grad_ = walloc(nis * njs);
#define CF(i,k) CF_[WSK(i,k)]
#define DC(i,k) DC_[WSK(i,k)]
#define FC(i,k) FC_[WSK(i,k)]
#define swdk(i,j,k) swdk_[WS2(i,j) + (long)(k) * nis * njs]
#define FE(i,j) FE_[WS2(i,j)]
#define FX(i,j) FX_[WS2(i,j)]
#define curv(i,j) curv_[WS2(i,j)]
#define grad(i,j) grad_[WS2(i,j)]

  /* SOLAR_SOURCE: fraction of solar shortwave at W-levels, pre_step3d.F:313-335
   * + lmd_swfrac_tile (lmd_swfrac.F:6), Zscale = -1 */
  if (p->solar_source) {
    const double Zscale = -1.0;
    for (int k = 1; k <= N - 1; k++) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) FX(i, j) = z_w(i, j, N) - z_w(i, j, k);
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          const double fac1 = Zscale / p->swfrac_mu1, fac2 = Zscale / p->swfrac_mu2, fac3 = p->swfrac_r1;
          FE(i, j) = exp(FX(i, j) * fac1) * fac3 + exp(FX(i, j) * fac2) * (1.0 - fac3);
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) swdk(i, j, k) = FE(i, j);
    }
  }

  /* T_LOOP1: horizontal advection of t(nstp), pre_step3d.F:342-617 */
  for (int itrc = 1; itrc <= NT; itrc++) {
    const int ha = p->Hadv[itrc - 1];
    for (int k = 1; k <= N; k++) {
      if (ha == ADV_MPDATA || ha == ADV_HSIMT) {          /* pre_step3d.F:364-386: first-order upstream for both */
        /* first-order upstream fluxes, pre_step3d.F:364-386 */
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            cff1 = MAX(Huon(i, j, k), 0.0);
            cff2 = MIN(Huon(i, j, k), 0.0);
            FX(i, j) = cff1 * t(i - 1, j, k, nstp, itrc) + cff2 * t(i, j, k, nstp, itrc);
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff1 = MAX(Hvom(i, j, k), 0.0);
            cff2 = MIN(Hvom(i, j, k), 0.0);
            FE(i, j) = cff1 * t(i, j - 1, k, nstp, itrc) + cff2 * t(i, j, k, nstp, itrc);
          }
      } else if (ha == ADV_C2) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++)
            FX(i, j) = Huon(i, j, k) * 0.5 * (t(i - 1, j, k, nstp, itrc) + t(i, j, k, nstp, itrc));
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++)
            FE(i, j) = Hvom(i, j, k) * 0.5 * (t(i, j - 1, k, nstp, itrc) + t(i, j, k, nstp, itrc));
      } else {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istrm1; i <= Iendp2; i++) {
            FX(i, j) = t(i, j, k, nstp, itrc) - t(i - 1, j, k, nstp, itrc);
            if (p->masking) FX(i, j) = FX(i, j) * umask(i, j);                     /* MASKING, pre_step3d.F:398 */
          }
        if (!EWperiodic) {
          if (west_edge) for (int j = Jstr; j <= Jend; j++) FX(Istr - 1, j) = FX(Istr, j);
          if (east_edge) for (int j = Jstr; j <= Jend; j++) FX(Iend + 2, j) = FX(Iend + 1, j);
        }
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) {
            if (ha == ADV_U3) curv(i, j) = FX(i + 1, j) - FX(i, j);
            else if (ha == ADV_A4) {
              cff = 2.0 * FX(i + 1, j) * FX(i, j);
              if (cff > eps) grad(i, j) = cff / (FX(i + 1, j) + FX(i, j));
              else grad(i, j) = 0.0;
            } else grad(i, j) = 0.5 * (FX(i + 1, j) + FX(i, j));
          }
        cff1 = 1.0 / 6.0;
        cff2 = 1.0 / 3.0;
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            if (ha == ADV_U3)
              FX(i, j) = Huon(i, j, k) * 0.5 * (t(i - 1, j, k, nstp, itrc) + t(i, j, k, nstp, itrc)) -
                         cff1 * (curv(i - 1, j) * MAX(Huon(i, j, k), 0.0) + curv(i, j) * MIN(Huon(i, j, k), 0.0));
            else
              FX(i, j) = Huon(i, j, k) * 0.5 *
                         (t(i - 1, j, k, nstp, itrc) + t(i, j, k, nstp, itrc) - cff2 * (grad(i, j) - grad(i - 1, j)));
          }
        for (int j = Jstrm1; j <= Jendp2; j++)
          for (int i = Istr; i <= Iend; i++) {
            FE(i, j) = t(i, j, k, nstp, itrc) - t(i, j - 1, k, nstp, itrc);
            if (p->masking) FE(i, j) = FE(i, j) * vmask(i, j);                     /* MASKING, pre_step3d.F:463 */
          }
        if (!NSperiodic) {
          if (south_edge) for (int i = Istr; i <= Iend; i++) FE(i, Jstr - 1) = FE(i, Jstr);
          if (north_edge) for (int i = Istr; i <= Iend; i++) FE(i, Jend + 2) = FE(i, Jend + 1);
        }
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            if (ha == ADV_U3) curv(i, j) = FE(i, j + 1) - FE(i, j);
            else if (ha == ADV_A4) {
              cff = 2.0 * FE(i, j + 1) * FE(i, j);
              if (cff > eps) grad(i, j) = cff / (FE(i, j + 1) + FE(i, j));
              else grad(i, j) = 0.0;
            } else grad(i, j) = 0.5 * (FE(i, j + 1) + FE(i, j));
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            if (ha == ADV_U3)
              FE(i, j) = Hvom(i, j, k) * 0.5 * (t(i, j - 1, k, nstp, itrc) + t(i, j, k, nstp, itrc)) -
                         cff1 * (curv(i, j - 1) * MAX(Hvom(i, j, k), 0.0) + curv(i, j) * MIN(Hvom(i, j, k), 0.0));
            else
              FE(i, j) = Hvom(i, j, k) * 0.5 *
                         (t(i, j - 1, k, nstp, itrc) + t(i, j, k, nstp, itrc) - cff2 * (grad(i, j) - grad(i, j - 1)));
          }
      }
      o_src_tflux(b, p, s, F, itrc, k, FX_, FE_, 0, 1);                          /* LuvSrc, pre_step3d.F:530-553 */
      Gamma = (ha == ADV_MPDATA || ha == ADV_HSIMT) ? 0.5 : 1.0 / 6.0;          /* pre_step3d.F:557-563 */
      if (iic == ntfirst) { cff = 0.5 * dt; cff1 = 1.0; cff2 = 0.0; }
      else { cff = (1.0 - Gamma) * dt; cff1 = 0.5 + Gamma; cff2 = 0.5 - Gamma; }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++)
          t(i, j, k, 3, itrc) = Hz(i, j, k) * (cff1 * t(i, j, k, nstp, itrc) + cff2 * t(i, j, k, nnew, itrc)) -
                                cff * pm(i, j) * pn(i, j) *
                                (FX(i + 1, j) - FX(i, j) + FE(i, j + 1) - FE(i, j));
    }
  }

  /* J_LOOP1/T_LOOP2: vertical advection + artificial continuity, pre_step3d.F:619-915 */
  for (int j = Jstr; j <= Jend; j++) {
    for (int itrc = 1; itrc <= NT; itrc++) {
      const int va = p->Vadv[itrc - 1];
      if (va == ADV_MPDATA || va == ADV_HSIMT) {          /* pre_step3d.F:729-748 */
        /* first-order upstream vertical flux, pre_step3d.F:729-748 */
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff1 = MAX(W(i, j, k), 0.0);
            cff2 = MIN(W(i, j, k), 0.0);
            FC(i, k) = cff1 * t(i, j, k, nstp, itrc) + cff2 * t(i, j, k + 1, nstp, itrc);
          }
        for (int i = Istr; i <= Iend; i++) { FC(i, 0) = 0.0; FC(i, N) = 0.0; }
      } else if (va == ADV_SPLINES) {
        for (int i = Istr; i <= Iend; i++) { FC(i, 0) = 1.5 * t(i, j, 1, nstp, itrc); CF(i, 1) = 0.5; }
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 1.0 / (2.0 * Hz(i, j, k) + Hz(i, j, k + 1) * (2.0 - CF(i, k)));
            CF(i, k + 1) = cff * Hz(i, j, k);
            FC(i, k) = cff * (3.0 * (Hz(i, j, k) * t(i, j, k + 1, nstp, itrc) + Hz(i, j, k + 1) * t(i, j, k, nstp, itrc)) -
                              Hz(i, j, k + 1) * FC(i, k - 1));
          }
        for (int i = Istr; i <= Iend; i++)
          FC(i, N) = (3.0 * t(i, j, N, nstp, itrc) - FC(i, N - 1)) / (2.0 - CF(i, N));
        for (int k = N - 1; k >= 0; k--)
          for (int i = Istr; i <= Iend; i++) {
            FC(i, k) = FC(i, k) - CF(i, k + 1) * FC(i, k + 1);
            FC(i, k + 1) = W(i, j, k + 1) * FC(i, k + 1);
          }
        for (int i = Istr; i <= Iend; i++) { FC(i, N) = 0.0; FC(i, 0) = 0.0; }
      } else if (va == ADV_A4) {
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) FC(i, k) = t(i, j, k + 1, nstp, itrc) - t(i, j, k, nstp, itrc);
        for (int i = Istr; i <= Iend; i++) { FC(i, 0) = FC(i, 1); FC(i, N) = FC(i, N - 1); }
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 2.0 * FC(i, k) * FC(i, k - 1);
            if (cff > eps) CF(i, k) = cff / (FC(i, k) + FC(i, k - 1));
            else CF(i, k) = 0.0;
          }
        cff1 = 1.0 / 3.0;
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = W(i, j, k) * 0.5 *
                       (t(i, j, k, nstp, itrc) + t(i, j, k + 1, nstp, itrc) - cff1 * (CF(i, k + 1) - CF(i, k)));
        for (int i = Istr; i <= Iend; i++) { FC(i, 0) = 0.0; FC(i, N) = 0.0; }
      } else if (va == ADV_C2) {
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = W(i, j, k) * 0.5 * (t(i, j, k, nstp, itrc) + t(i, j, k + 1, nstp, itrc));
        for (int i = Istr; i <= Iend; i++) { FC(i, 0) = 0.0; FC(i, N) = 0.0; }
      } else {
        cff1 = 0.5; cff2 = 7.0 / 12.0; cff3 = 1.0 / 12.0;
        for (int k = 2; k <= N - 2; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = W(i, j, k) * (cff2 * (t(i, j, k, nstp, itrc) + t(i, j, k + 1, nstp, itrc)) -
                                     cff3 * (t(i, j, k - 1, nstp, itrc) + t(i, j, k + 2, nstp, itrc)));
        for (int i = Istr; i <= Iend; i++) {
          FC(i, 0) = 0.0;
          FC(i, 1) = W(i, j, 1) * (cff1 * t(i, j, 1, nstp, itrc) + cff2 * t(i, j, 2, nstp, itrc) - cff3 * t(i, j, 3, nstp, itrc));
          FC(i, N - 1) = W(i, j, N - 1) * (cff1 * t(i, j, N, nstp, itrc) + cff2 * t(i, j, N - 1, nstp, itrc) -
                                           cff3 * t(i, j, N - 2, nstp, itrc));
          FC(i, N) = 0.0;
        }
      }
      Gamma = (va == ADV_MPDATA || va == ADV_HSIMT) ? 0.5 : 1.0 / 6.0;          /* pre_step3d.F:793-799 */
      if (iic == ntfirst) cff = 0.5 * dt;
      else cff = (1.0 - Gamma) * dt;
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++)
          DC(i, k) = 1.0 / (Hz(i, j, k) - cff * pm(i, j) * pn(i, j) *
                                          (Huon(i + 1, j, k) - Huon(i, j, k) + Hvom(i, j + 1, k) - Hvom(i, j, k) +
                                           (W(i, j, k) - W(i, j, k - 1))));
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = cff * pm(i, j) * pn(i, j);
          t(i, j, k, 3, itrc) = DC(i, k) * (t(i, j, k, 3, itrc) - cff1 * (FC(i, k) - FC(i, k - 1)));
        }
    }
  }

  /* start computation of t(nnew): explicit vertical diffusion + fluxes, pre_step3d.F:917-1010 */
  for (int j = Jstr; j <= Jend; j++) {
    cff3 = dt * (1.0 - lambda);
    for (int itrc = 1; itrc <= NT; itrc++) {
      const int ltrc = MIN(NAT, itrc);
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 1.0 / (z_r(i, j, k + 1) - z_r(i, j, k));
          FC(i, k) = cff3 * cff * Akt(i, j, k, ltrc) * (t(i, j, k + 1, nstp, itrc) - t(i, j, k, nstp, itrc));
        }
      if (p->lmd_nonlocal && itrc <= NAT)
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = FC(i, k) - dt * Akt(i, j, k, itrc) * ghats(i, j, k, itrc);
      if (p->solar_source && itrc == itemp)
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) FC(i, k) = FC(i, k) + dt * srflx(i, j) * (p->wet_dry ? rmask_wet(i, j) : 1.0) * swdk(i, j, k);   /* WET_DRY, :876 */
      for (int i = Istr; i <= Iend; i++) {
        FC(i, 0) = dt * btflx(i, j, itrc);
        FC(i, N) = dt * stflx(i, j, itrc);
      }
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = Hz(i, j, k) * t(i, j, k, nstp, itrc);
          cff2 = FC(i, k) - FC(i, k - 1);
          t(i, j, k, nnew, itrc) = cff1 + cff2;
        }
    }
  }

  /* J_LOOP2: start of u,v(nnew), pre_step3d.F:1012-1120 */
  for (int j = Jstr; j <= Jend; j++) {
    cff3 = dt * (1.0 - lambda);
    for (int k = 1; k <= N - 1; k++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = 1.0 / (z_r(i, j, k + 1) + z_r(i - 1, j, k + 1) - z_r(i, j, k) - z_r(i - 1, j, k));
        FC(i, k) = cff3 * cff * (u(i, j, k + 1, nstp) - u(i, j, k, nstp)) * (Akv(i, j, k) + Akv(i - 1, j, k));
      }
    for (int i = IstrU; i <= Iend; i++) { FC(i, 0) = dt * bustr(i, j); FC(i, N) = dt * sustr(i, j); }
    cff = dt * 0.25;
    for (int i = IstrU; i <= Iend; i++) DC(i, 0) = cff * (pm(i, j) + pm(i - 1, j)) * (pn(i, j) + pn(i - 1, j));
    const int indx = 3 - nrhs;
    if (iic == ntfirst) {
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = u(i, j, k, nstp) * 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k));
          cff2 = FC(i, k) - FC(i, k - 1);
          u(i, j, k, nnew) = cff1 + cff2;
        }
    } else if (iic == ntfirst + 1) {
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = u(i, j, k, nstp) * 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k));
          cff2 = FC(i, k) - FC(i, k - 1);
          cff3 = 0.5 * DC(i, 0);
          u(i, j, k, nnew) = cff1 - cff3 * ru(i, j, k, indx) + cff2;
        }
    } else {
      cff1 = 5.0 / 12.0;
      cff2 = 16.0 / 12.0;
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff3 = u(i, j, k, nstp) * 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k));
          cff4 = FC(i, k) - FC(i, k - 1);
          u(i, j, k, nnew) = cff3 + DC(i, 0) * (cff1 * ru(i, j, k, nrhs) - cff2 * ru(i, j, k, indx)) + cff4;
        }
    }
    if (j >= JstrV) {
      cff3 = dt * (1.0 - lambda);
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 1.0 / (z_r(i, j, k + 1) + z_r(i, j - 1, k + 1) - z_r(i, j, k) - z_r(i, j - 1, k));
          FC(i, k) = cff3 * cff * (v(i, j, k + 1, nstp) - v(i, j, k, nstp)) * (Akv(i, j, k) + Akv(i, j - 1, k));
        }
      for (int i = Istr; i <= Iend; i++) { FC(i, 0) = dt * bvstr(i, j); FC(i, N) = dt * svstr(i, j); }
      cff = dt * 0.25;
      for (int i = Istr; i <= Iend; i++) DC(i, 0) = cff * (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1));
      if (iic == ntfirst) {
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff1 = v(i, j, k, nstp) * 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k));
            cff2 = FC(i, k) - FC(i, k - 1);
            v(i, j, k, nnew) = cff1 + cff2;
          }
      } else if (iic == ntfirst + 1) {
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff1 = v(i, j, k, nstp) * 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k));
            cff2 = FC(i, k) - FC(i, k - 1);
            cff3 = 0.5 * DC(i, 0);
            v(i, j, k, nnew) = cff1 - cff3 * rv(i, j, k, indx) + cff2;
          }
      } else {
        cff1 = 5.0 / 12.0;
        cff2 = 16.0 / 12.0;
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff3 = v(i, j, k, nstp) * 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k));
            cff4 = FC(i, k) - FC(i, k - 1);
            v(i, j, k, nnew) = cff3 + DC(i, 0) * (cff1 * rv(i, j, k, nrhs) - cff2 * rv(i, j, k, indx)) + cff4;
          }
      }
    }
  }

  /* tracer BCs on the predictor level + periodic wrap, pre_step3d.F:1131-1145 */
  for (int itrc = 1; itrc <= NT; itrc++) {
    o_t3dbc(b, p, s, F, 3, itrc);
    o_exchange3d(b, GT_R, N, &t(LBi, LBj, 1, 3, itrc));
  }
  free(CF_); free(DC_); free(FC_); free(swdk_); free(FE_); free(FX_); free(curv_); free(grad_);
  return 0;
}
