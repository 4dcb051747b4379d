/*
 * oracle_step3d_uv.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * step3d_uv_tile: corrector step for 3-D momentum with implicit vertical
 * viscosity (spline form), replacement of the vertical mean by the fast-time
 * averaged barotropic transport, and correction of the mass fluxes
 * (ROMS/Nonlinear/step3d_uv.F:111-1482).  Parity unpinned (mod_sources).
 */
#include "oracle.h"

int oracle_step3d_uv(OARGS)
{
  ORACLE_PROLOGUE
  if (o_src_check(p)) return 8;
  if (o_check_lbc(b, p)) return 8;
  const int nrhs = s->nrhs, nnew = s->nnew;
  const int iic = s->iic, ntfirst = s->ntfirst;
  const double dt = p->dt;
  double cff, cff1;
  const long nk = nis * (N + 1);
  double *AK_ = walloc(nk), *BC_ = walloc(nk), *CF_ = walloc(nk), *DC_ = walloc(nk), *FC_ = walloc(nk);
  double *Hzk_ = walloc(nk), *oHz_ = walloc(nk);
#define AK(i,k) AK_[WSK(i,k)]
#define BC(i,k) BC_[WSK(i,k)]
#define CF(i,k) CF_[WSK(i,k)]
#define DC(i,k) DC_[WSK(i,k)]
#define FC(i,k) FC_[WSK(i,k)]
#define Hzk(i,k) Hzk_[WSK(i,k)]
#define oHz(i,k) oHz_[WSK(i,k)]

  for (int j = Jstr; j <= Jend; j++) {
    /* ---- u, step3d_uv.F:282-520 ---- */
    for (int i = IstrU; i <= Iend; i++) {
      AK(i, 0) = 0.5 * (Akv(i - 1, j, 0) + Akv(i, j, 0));
      for (int k = 1; k <= N; k++) {
        AK(i, k) = 0.5 * (Akv(i - 1, j, k) + Akv(i, j, k));
        Hzk(i, k) = 0.5 * (Hz(i - 1, j, k) + Hz(i, j, k));
        oHz(i, k) = 1.0 / Hzk(i, k);
      }
    }
    if (iic == ntfirst) cff = 0.25 * dt;
    else if (iic == ntfirst + 1) cff = 0.25 * dt * 3.0 / 2.0;
    else cff = 0.25 * dt * 23.0 / 12.0;
    for (int i = IstrU; i <= Iend; i++) DC(i, 0) = cff * (pm(i, j) + pm(i - 1, j)) * (pn(i, j) + pn(i - 1, j));
    for (int k = 1; k <= N; k++)
      for (int i = IstrU; i <= Iend; i++) {
        u(i, j, k, nnew) = u(i, j, k, nnew) + DC(i, 0) * ru(i, j, k, nrhs);
        if (p->splines_vvisc) u(i, j, k, nnew) = u(i, j, k, nnew) * oHz(i, k);
      }
    if (!p->splines_vvisc) {
      /* the implicit vertical viscosity without SPLINES_VVISC, step3d_uv.F:400-464: a tridiagonal system for u itself */
      cff = -p->lambda * dt / 0.5;
      for (int k = 1; k <= N - 1; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = 1.0 / (z_r(i, j, k + 1) + z_r(i - 1, j, k + 1) - z_r(i, j, k) - z_r(i - 1, j, k));
          FC(i, k) = cff * cff1 * AK(i, k);
        }
      for (int i = IstrU; i <= Iend; i++) { FC(i, 0) = 0.0; FC(i, N) = 0.0; }
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          DC(i, k) = u(i, j, k, nnew);
          BC(i, k) = Hzk(i, k) - FC(i, k) - FC(i, k - 1);
        }
      for (int i = IstrU; i <= Iend; i++) {
        cff = 1.0 / BC(i, 1);
        CF(i, 1) = cff * FC(i, 1);
        DC(i, 1) = cff * DC(i, 1);
      }
      for (int k = 2; k <= N - 1; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff = 1.0 / (BC(i, k) - FC(i, k - 1) * CF(i, k - 1));
          CF(i, k) = cff * FC(i, k);
          DC(i, k) = cff * (DC(i, k) - FC(i, k - 1) * DC(i, k - 1));
        }
      for (int i = IstrU; i <= Iend; i++) {
        DC(i, N) = (DC(i, N) - FC(i, N - 1) * DC(i, N - 1)) / (BC(i, N) - FC(i, N - 1) * CF(i, N - 1));
        u(i, j, N, nnew) = DC(i, N);
      }
      for (int k = N - 1; k >= 1; k--)
        for (int i = IstrU; i <= Iend; i++) {
          DC(i, k) = DC(i, k) - CF(i, k) * DC(i, k + 1);
          u(i, j, k, nnew) = DC(i, k);
        }
    } else {
      cff1 = 1.0 / 6.0;
      for (int k = 1; k <= N - 1; k++)
        for (int i = IstrU; i <= Iend; i++) {
          FC(i, k) = cff1 * Hzk(i, k) - dt * AK(i, k - 1) * oHz(i, k);
          CF(i, k) = cff1 * Hzk(i, k + 1) - dt * AK(i, k + 1) * oHz(i, k + 1);
        }
      for (int i = IstrU; i <= Iend; i++) { CF(i, 0) = 0.0; DC(i, 0) = 0.0; }
      cff1 = 1.0 / 3.0;
      for (int k = 1; k <= N - 1; k++)
        for (int i = IstrU; i <= Iend; i++) {
          BC(i, k) = cff1 * (Hzk(i, k) + Hzk(i, k + 1)) + dt * AK(i, k) * (oHz(i, k) + oHz(i, k + 1));
          cff = 1.0 / (BC(i, k) - FC(i, k) * CF(i, k - 1));
          CF(i, k) = cff * CF(i, k);
          DC(i, k) = cff * (u(i, j, k + 1, nnew) - u(i, j, k, nnew) - FC(i, k) * DC(i, k - 1));
        }
      for (int i = IstrU; i <= Iend; i++) DC(i, N) = 0.0;
      for (int k = N - 1; k >= 1; k--)
        for (int i = IstrU; i <= Iend; i++) DC(i, k) = DC(i, k) - CF(i, k) * DC(i, k + 1);
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          DC(i, k) = DC(i, k) * AK(i, k);
          cff = dt * oHz(i, k) * (DC(i, k) - DC(i, k - 1));
          u(i, j, k, nnew) = u(i, j, k, nnew) + cff;
        }
    }
    /* replace the vertical mean with the barotropic one, step3d_uv.F:466-520 */
    for (int i = IstrU; i <= Iend; i++) { CF(i, 0) = Hzk(i, 1); DC(i, 0) = u(i, j, 1, nnew) * Hzk(i, 1); }
    for (int k = 2; k <= N; k++)
      for (int i = IstrU; i <= Iend; i++) {
        CF(i, 0) = CF(i, 0) + Hzk(i, k);
        DC(i, 0) = DC(i, 0) + u(i, j, k, nnew) * Hzk(i, k);
      }
    for (int i = IstrU; i <= Iend; i++) {
      cff1 = 1.0 / (CF(i, 0) * on_u(i, j));
      DC(i, 0) = (DC(i, 0) * on_u(i, j) - DU_avg1(i, j)) * cff1;
    }
    for (int k = 1; k <= N; k++)
      for (int i = IstrU; i <= Iend; i++) {
        u(i, j, k, nnew) = u(i, j, k, nnew) - DC(i, 0);
        if (p->masking) u(i, j, k, nnew) = u(i, j, k, nnew) * umask(i, j);      /* MASKING, step3d_uv.F:558/:891/:1137/:1166/:1355/:1384 */
        if (p->wet_dry) {                                                        /* WET_DRY, step3d_uv.F:561-564 / :894-897 */
          u(i, j, k, nnew) = u(i, j, k, nnew) * umask_wet(i, j);
          ru(i, j, k, nrhs) = ru(i, j, k, nrhs) * umask_wet(i, j);
        }
      }

    /* ---- v, step3d_uv.F:620-860 ---- */
    if (j >= JstrV) {
      for (int i = Istr; i <= Iend; i++) {
        AK(i, 0) = 0.5 * (Akv(i, j - 1, 0) + Akv(i, j, 0));
        for (int k = 1; k <= N; k++) {
          AK(i, k) = 0.5 * (Akv(i, j - 1, k) + Akv(i, j, k));
          Hzk(i, k) = 0.5 * (Hz(i, j - 1, k) + Hz(i, j, k));
          oHz(i, k) = 1.0 / Hzk(i, k);
        }
      }
      if (iic == ntfirst) cff = 0.25 * dt;
      else if (iic == ntfirst + 1) cff = 0.25 * dt * 3.0 / 2.0;
      else cff = 0.25 * dt * 23.0 / 12.0;
      for (int i = Istr; i <= Iend; i++) DC(i, 0) = cff * (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1));
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          v(i, j, k, nnew) = v(i, j, k, nnew) + DC(i, 0) * rv(i, j, k, nrhs);
          if (p->splines_vvisc) v(i, j, k, nnew) = v(i, j, k, nnew) * oHz(i, k);
        }
      if (!p->splines_vvisc) {
        /* without SPLINES_VVISC, step3d_uv.F:733-797 */
        cff = -p->lambda * dt / 0.5;
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff1 = 1.0 / (z_r(i, j, k + 1) + z_r(i, j - 1, k + 1) - z_r(i, j, k) - z_r(i, j - 1, k));
            FC(i, k) = cff * cff1 * AK(i, k);
          }
        for (int i = Istr; i <= Iend; i++) { FC(i, 0) = 0.0; FC(i, N) = 0.0; }
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            DC(i, k) = v(i, j, k, nnew);
            BC(i, k) = Hzk(i, k) - FC(i, k) - FC(i, k - 1);
          }
        for (int i = Istr; i <= Iend; i++) {
          cff = 1.0 / BC(i, 1);
          CF(i, 1) = cff * FC(i, 1);
          DC(i, 1) = cff * DC(i, 1);
        }
        for (int k = 2; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 1.0 / (BC(i, k) - FC(i, k - 1) * CF(i, k - 1));
            CF(i, k) = cff * FC(i, k);
            DC(i, k) = cff * (DC(i, k) - FC(i, k - 1) * DC(i, k - 1));
          }
        for (int i = Istr; i <= Iend; i++) {
          DC(i, N) = (DC(i, N) - FC(i, N - 1) * DC(i, N - 1)) / (BC(i, N) - FC(i, N - 1) * CF(i, N - 1));
          v(i, j, N, nnew) = DC(i, N);
        }
        for (int k = N - 1; k >= 1; k--)
          for (int i = Istr; i <= Iend; i++) {
            DC(i, k) = DC(i, k) - CF(i, k) * DC(i, k + 1);
            v(i, j, k, nnew) = DC(i, k);
          }
      } else {
        cff1 = 1.0 / 6.0;
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) {
            FC(i, k) = cff1 * Hzk(i, k) - dt * AK(i, k - 1) * oHz(i, k);
            CF(i, k) = cff1 * Hzk(i, k + 1) - dt * AK(i, k + 1) * oHz(i, k + 1);
          }
        for (int i = Istr; i <= Iend; i++) { CF(i, 0) = 0.0; DC(i, 0) = 0.0; }
        cff1 = 1.0 / 3.0;
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) {
            BC(i, k) = cff1 * (Hzk(i, k) + Hzk(i, k + 1)) + dt * AK(i, k) * (oHz(i, k) + oHz(i, k + 1));
            cff = 1.0 / (BC(i, k) - FC(i, k) * CF(i, k - 1));
            CF(i, k) = cff * CF(i, k);
            DC(i, k) = cff * (v(i, j, k + 1, nnew) - v(i, j, k, nnew) - FC(i, k) * DC(i, k - 1));
          }
        for (int i = Istr; i <= Iend; i++) DC(i, N) = 0.0;
        for (int k = N - 1; k >= 1; k--)
          for (int i = Istr; i <= Iend; i++) DC(i, k) = DC(i, k) - CF(i, k) * DC(i, k + 1);
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            DC(i, k) = DC(i, k) * AK(i, k);
            cff = dt * oHz(i, k) * (DC(i, k) - DC(i, k - 1));
            v(i, j, k, nnew) = v(i, j, k, nnew) + cff;
          }
      }
      for (int i = Istr; i <= Iend; i++) { CF(i, 0) = Hzk(i, 1); DC(i, 0) = v(i, j, 1, nnew) * Hzk(i, 1); }
      for (int k = 2; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          CF(i, 0) = CF(i, 0) + Hzk(i, k);
          DC(i, 0) = DC(i, 0) + v(i, j, k, nnew) * Hzk(i, k);
        }
      for (int i = Istr; i <= Iend; i++) {
        cff1 = 1.0 / (CF(i, 0) * om_v(i, j));
        DC(i, 0) = (DC(i, 0) * om_v(i, j) - DV_avg1(i, j)) * cff1;
      }
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          v(i, j, k, nnew) = v(i, j, k, nnew) - DC(i, 0);
          if (p->masking) v(i, j, k, nnew) = v(i, j, k, nnew) * vmask(i, j);      /* MASKING, step3d_uv.F:558/:891/:1137/:1166/:1355/:1384 */
          if (p->wet_dry) {                                                        /* WET_DRY, step3d_uv.F:561-564 / :894-897 */
            v(i, j, k, nnew) = v(i, j, k, nnew) * vmask_wet(i, j);
            rv(i, j, k, nrhs) = rv(i, j, k, nrhs) * vmask_wet(i, j);
          }
        }
    }
  }

  /* lateral BCs, step3d_uv.F:956-961 */
  o_u3dbc(b, p, s, F, nnew);
  o_v3dbc(b, p, s, F, nnew);
  o_src_uv(b, p, s, F, nnew);                        /* LuvSrc, step3d_uv.F:971-995 */

  /* coupling 2-D and 3-D momentum, corrected mass fluxes, step3d_uv.F:997-1460 */
  for (int j = JstrT; j <= JendT; j++) {
    for (int i = IstrP; i <= IendT; i++) { DC(i, 0) = 0.0; CF(i, 0) = 0.0; FC(i, 0) = 0.0; }
    for (int k = 1; k <= N; k++)
      for (int i = IstrP; i <= IendT; i++) {
        cff = 0.5 * on_u(i, j);
        DC(i, k) = cff * (Hz(i, j, k) + Hz(i - 1, j, k));
        DC(i, 0) = DC(i, 0) + DC(i, k);
        CF(i, 0) = CF(i, 0) + DC(i, k) * u(i, j, k, nnew);
      }
    for (int i = IstrP; i <= IendT; i++) {
      DC(i, 0) = 1.0 / DC(i, 0);
      CF(i, 0) = DC(i, 0) * (CF(i, 0) - DU_avg1(i, j));
      ubar(i, j, 1) = DC(i, 0) * DU_avg1(i, j);
      if (p->wet_dry) ubar(i, j, 1) = ubar(i, j, 1) * umask_wet(i, j);                /* WET_DRY, step3d_uv.F:1042-1044 */
      ubar(i, j, 2) = ubar(i, j, 1);
    }
    if (!EWperiodic) {
      if (west_edge) for (int k = 1; k <= N; k++) {
        u(Istr, j, k, nnew) = u(Istr, j, k, nnew) - CF(Istr, 0);
        if (p->masking) u(Istr, j, k, nnew) = u(Istr, j, k, nnew) * umask(Istr, j);               /* :1081-1084 */
        if (p->wet_dry) u(Istr, j, k, nnew) = u(Istr, j, k, nnew) * umask_wet(Istr, j);   /* WET_DRY, the next block */
      }
      if (east_edge) for (int k = 1; k <= N; k++) {
        u(Iend + 1, j, k, nnew) = u(Iend + 1, j, k, nnew) - CF(Iend + 1, 0);
        if (p->masking) u(Iend + 1, j, k, nnew) = u(Iend + 1, j, k, nnew) * umask(Iend + 1, j);   /* :1108-1111 */
        if (p->wet_dry) u(Iend + 1, j, k, nnew) = u(Iend + 1, j, k, nnew) * umask_wet(Iend + 1, j);   /* WET_DRY, the next block */
      }
    }
    if (!NSperiodic) {
      if (j == 0)
        for (int k = 1; k <= N; k++)
          for (int i = IstrU; i <= Iend; i++) {
            u(i, j, k, nnew) = u(i, j, k, nnew) - CF(i, 0);
            if (p->masking) u(i, j, k, nnew) = u(i, j, k, nnew) * umask(i, j);      /* MASKING, step3d_uv.F:558/:891/:1137/:1166/:1355/:1384 */
            if (p->wet_dry) u(i, j, k, nnew) = u(i, j, k, nnew) * umask_wet(i, j);   /* WET_DRY, the next block */
          }
      if (j == Mm + 1)
        for (int k = 1; k <= N; k++)
          for (int i = IstrU; i <= Iend; i++) {
            u(i, j, k, nnew) = u(i, j, k, nnew) - CF(i, 0);
            if (p->masking) u(i, j, k, nnew) = u(i, j, k, nnew) * umask(i, j);      /* MASKING, step3d_uv.F:558/:891/:1137/:1166/:1355/:1384 */
            if (p->wet_dry) u(i, j, k, nnew) = u(i, j, k, nnew) * umask_wet(i, j);   /* WET_DRY, the next block */
          }
    }
    for (int k = N; k >= 1; k--)
      for (int i = IstrP; i <= IendT; i++) {
        Huon(i, j, k) = 0.5 * (Huon(i, j, k) + u(i, j, k, nnew) * DC(i, k));
        FC(i, 0) = FC(i, 0) + Huon(i, j, k);
      }
    for (int i = IstrP; i <= IendT; i++) FC(i, 0) = DC(i, 0) * (FC(i, 0) - DU_avg2(i, j));
    for (int k = 1; k <= N; k++)
      for (int i = IstrP; i <= IendT; i++) Huon(i, j, k) = Huon(i, j, k) - DC(i, k) * FC(i, 0);

    if (j >= Jstr) {
      for (int i = IstrT; i <= IendT; i++) { DC(i, 0) = 0.0; CF(i, 0) = 0.0; FC(i, 0) = 0.0; }
      for (int k = 1; k <= N; k++)
        for (int i = IstrT; i <= IendT; i++) {
          cff = 0.5 * om_v(i, j);
          DC(i, k) = cff * (Hz(i, j, k) + Hz(i, j - 1, k));
          DC(i, 0) = DC(i, 0) + DC(i, k);
          CF(i, 0) = CF(i, 0) + DC(i, k) * v(i, j, k, nnew);
        }
      for (int i = IstrT; i <= IendT; i++) {
        DC(i, 0) = 1.0 / DC(i, 0);
        CF(i, 0) = DC(i, 0) * (CF(i, 0) - DV_avg1(i, j));
        vbar(i, j, 1) = DC(i, 0) * DV_avg1(i, j);
        if (p->wet_dry) vbar(i, j, 1) = vbar(i, j, 1) * vmask_wet(i, j);              /* WET_DRY, step3d_uv.F:1255-1257 */
        vbar(i, j, 2) = vbar(i, j, 1);
      }
      if (!EWperiodic) {
        if (west_edge) for (int k = 1; k <= N; k++) {
          v(Istr - 1, j, k, nnew) = v(Istr - 1, j, k, nnew) - CF(Istr - 1, 0);
          if (p->masking) v(Istr - 1, j, k, nnew) = v(Istr - 1, j, k, nnew) * vmask(Istr - 1, j);   /* :1297-1301 */
          if (p->wet_dry) v(Istr - 1, j, k, nnew) = v(Istr - 1, j, k, nnew) * vmask_wet(Istr - 1, j);   /* WET_DRY, the next block */
        }
        if (east_edge) for (int k = 1; k <= N; k++) {
          v(Iend + 1, j, k, nnew) = v(Iend + 1, j, k, nnew) - CF(Iend + 1, 0);
          if (p->masking) v(Iend + 1, j, k, nnew) = v(Iend + 1, j, k, nnew) * vmask(Iend + 1, j);
          if (p->wet_dry) v(Iend + 1, j, k, nnew) = v(Iend + 1, j, k, nnew) * vmask_wet(Iend + 1, j);   /* WET_DRY, the next block */
        }
      }
      if (!NSperiodic) {
        if (j == 1)
          for (int k = 1; k <= N; k++)
            for (int i = Istr; i <= Iend; i++) {
              v(i, j, k, nnew) = v(i, j, k, nnew) - CF(i, 0);
              if (p->masking) v(i, j, k, nnew) = v(i, j, k, nnew) * vmask(i, j);      /* MASKING, step3d_uv.F:558/:891/:1137/:1166/:1355/:1384 */
              if (p->wet_dry) v(i, j, k, nnew) = v(i, j, k, nnew) * vmask_wet(i, j);   /* WET_DRY, the next block */
            }
        if (j == Mm + 1)
          for (int k = 1; k <= N; k++)
            for (int i = Istr; i <= Iend; i++) {
              v(i, j, k, nnew) = v(i, j, k, nnew) - CF(i, 0);
              if (p->masking) v(i, j, k, nnew) = v(i, j, k, nnew) * vmask(i, j);      /* MASKING, step3d_uv.F:558/:891/:1137/:1166/:1355/:1384 */
              if (p->wet_dry) v(i, j, k, nnew) = v(i, j, k, nnew) * vmask_wet(i, j);   /* WET_DRY, the next block */
            }
      }
      for (int k = N; k >= 1; k--)
        for (int i = IstrT; i <= IendT; i++) {
          Hvom(i, j, k) = 0.5 * (Hvom(i, j, k) + v(i, j, k, nnew) * DC(i, k));
          FC(i, 0) = FC(i, 0) + Hvom(i, j, k);
        }
      for (int i = IstrT; i <= IendT; i++) FC(i, 0) = DC(i, 0) * (FC(i, 0) - DV_avg2(i, j));
      for (int k = 1; k <= N; k++)
        for (int i = IstrT; i <= IendT; i++) Hvom(i, j, k) = Hvom(i, j, k) - DC(i, k) * FC(i, 0);
    }
  }

  /* periodic wrap / mp_exchange, step3d_uv.F:1436-1475 */
  o_exchange3d(b, GT_U, N, &u(LBi, LBj, 1, nnew));
  o_exchange3d(b, GT_V, N, &v(LBi, LBj, 1, nnew));
  o_exchange3d(b, GT_U, N, F->Huon);
  o_exchange3d(b, GT_V, N, F->Hvom);
  for (int k = 1; k <= 2; k++) {
    o_exchange2d(b, GT_U, &ubar(LBi, LBj, k));
    o_exchange2d(b, GT_V, &vbar(LBi, LBj, k));
  }
  free(AK_); free(BC_); free(CF_); free(DC_); free(FC_); free(Hzk_); free(oHz_);
  return 0;
}
