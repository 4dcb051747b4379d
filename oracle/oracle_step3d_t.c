/*
 * oracle_step3d_t.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * step3d_t_tile: corrector time-step for tracers
 * (ROMS/Nonlinear/step3d_t.F:108-1682).  Parity unpinned: the reference file
 * cannot be compiled here (USE mod_sources -> mod_netcdf).
 */
#include "oracle.h"

int oracle_mpdata_adiff(const roms_bounds_t *b, const roms_params_t *p, const roms_step_idx_t *s,
                        roms_fields_t *F, const double *oHz_, const double *t3_,
                        double *Ta_, double *Ua_, double *Va_, double *Wa_);   /* oracle_mpdata_adiff.c */

int oracle_step3d_t(OARGS)
{
  ORACLE_PROLOGUE
  if (o_src_check(p)) return 8;
  if (o_check_lbc(b, p)) return 8;
  const int nnew = s->nnew;
  const double dt = p->dt;
  const double eps = 1.0E-16;
  int Lmpdata = 0, Lhsimt = 0;
  const double eps1 = 1.0E-12, cc1 = 0.25, cc2 = 0.5, cc3 = 1.0 / 12.0;      /* step3d_t.F:255, mod_scalars.F:376-378 */
  for (int itrc = 1; itrc <= NT; itrc++) {
    int ha = p->Hadv[itrc - 1], va = p->Vadv[itrc - 1];
    if (ha == ADV_SPLINES) return 8;
    if (va == ADV_U3) return 8;
    /* HSIMT horizontally alone reads oHz at halo points the reference has not computed (Lhsimt needs both directions,
     * step3d_t.F:304-305, :340-360, :445); HSIMT vertically with another horizontal scheme is a working pair */
    if (ha == ADV_HSIMT && va != ADV_HSIMT) return 8;
    if (ha == ADV_HSIMT) Lhsimt = 1;
    /* the reference lets H and V differ; only the pair MPDATA/MPDATA is restated */
    if ((ha == ADV_MPDATA) != (va == ADV_MPDATA)) return 8;
    if (ha == ADV_MPDATA) Lmpdata = 1;
  }
  if ((Lmpdata || Lhsimt) && b->NghostPoints != 3) return 8;          /* inp_par.F:266-278 */
  double *FX_ = walloc(nis * njs), *FE_ = walloc(nis * njs);
  double *curv_ = walloc(nis * njs), *grad_ = walloc(nis * njs);
  double *oHz_ = walloc(nis * njs * N);
  double *CF_ = walloc(nis * (N + 1)), *BC_ = walloc(nis * (N + 1));
  double *DC_ = walloc(nis * (N + 1)), *FC_ = walloc(nis * (N + 1));
  /* step3d_t.F:311-330: Ta(..,N,NT), Ua, Va, Wa allocated only when MPDATA is in use */
  double *Ta_ = Lmpdata ? walloc(nis * njs * N * NT) : NULL;
  double *Ua_ = Lmpdata ? walloc(nis * njs * N) : NULL, *Va_ = Lmpdata ? walloc(nis * njs * N) : NULL;
  double *Wa_ = Lmpdata ? walloc(nis * njs * (N + 1)) : NULL;
#define Ta(i,j,k,it) Ta_[WS3(i,j,k) + (long)((it) - 1) * nis * njs * N]
#define Ua(i,j,k)    Ua_[WS3(i,j,k)]
#define Va(i,j,k)    Va_[WS3(i,j,k)]
#define Wa(i,j,k)    Wa_[WS2(i,j) + (long)(k) * nis * njs]
#define FX(i,j)    FX_[WS2(i,j)]
#define FE(i,j)    FE_[WS2(i,j)]
#define curv(i,j)  curv_[WS2(i,j)]
#define grad(i,j)  grad_[WS2(i,j)]
#define oHz(i,j,k) oHz_[WS3(i,j,k)]
#define CF(i,k)    CF_[WSK(i,k)]
#define BC(i,k)    BC_[WSK(i,k)]
#define DC(i,k)    DC_[WSK(i,k)]
#define FC(i,k)    FC_[WSK(i,k)]

  /* step3d_t.F:340-360 */
  if (Lmpdata || Lhsimt) {
    for (int k = 1; k <= N; k++)
      for (int j = Jstrm2; j <= Jendp2; j++)
        for (int i = Istrm2; i <= Iendp2; i++) oHz(i, j, k) = 1.0 / Hz(i, j, k);
  } else {
    for (int k = 1; k <= N; k++)
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) oHz(i, j, k) = 1.0 / Hz(i, j, k);
  }

  /* T_LOOP1 / K_LOOP: horizontal advection, step3d_t.F:363-880 */
  for (int itrc = 1; itrc <= NT; itrc++) {
    const int ha = p->Hadv[itrc - 1];
    /* three-point footprint: refresh the ghost points of t(nnew) first, :369-386 */
    if (ha == ADV_MPDATA || ha == ADV_HSIMT) o_exchange3d(b, GT_R, N, &t(LBi, LBj, 1, nnew, itrc));
    for (int k = 1; k <= N; k++) {
      if (ha == ADV_MPDATA) {
        /* first-order upstream fluxes on the extended range, :409-428 */
        for (int j = JstrVm2; j <= Jendp2i; j++)
          for (int i = IstrUm2; i <= Iendp3; i++) {
            double cff1 = MAX(Huon(i, j, k), 0.0), cff2 = MIN(Huon(i, j, k), 0.0);
            FX(i, j) = cff1 * t(i - 1, j, k, 3, itrc) + cff2 * t(i, j, k, 3, itrc);
          }
        for (int j = JstrVm2; j <= Jendp3; j++)
          for (int i = IstrUm2; i <= Iendp2i; i++) {
            double cff1 = MAX(Hvom(i, j, k), 0.0), cff2 = MIN(Hvom(i, j, k), 0.0);
            FE(i, j) = cff1 * t(i, j - 1, k, 3, itrc) + cff2 * t(i, j, k, 3, itrc);
          }
        o_src_tflux(b, p, s, F, itrc, k, FX_, FE_, 1, 0);                       /* LuvSrc, :734-799 */
        /* intermediate diffusive tracer Ta (m Tunits), :831-840 */
        for (int j = JstrVm2; j <= Jendp2i; j++)
          for (int i = IstrUm2; i <= Iendp2i; i++) {
            double cff = dt * pm(i, j) * pn(i, j);
            double cff1 = cff * (FX(i + 1, j) - FX(i, j));
            double cff2 = cff * (FE(i, j + 1) - FE(i, j));
            double cff3 = cff1 + cff2;
            Ta(i, j, k, itrc) = t(i, j, k, nnew, itrc) - cff3;
          }
        continue;
      }
      if (ha == ADV_HSIMT) {
        /* HSIMT with TVD limiter (Wu and Zhu, 2010), horizontal fluxes, step3d_t.F:430-590 */
        double gX_[512 + 8], kX_[512 + 8], oX_[512 + 8];
        double *gradX = NULL, *KaX = NULL, *oKaX = NULL, *line = NULL;
        const long nline = (nis > njs ? nis : njs) + 8;
        if (nline <= 520) { gradX = gX_; KaX = kX_; oKaX = oX_; }
        else { line = (double *)malloc(sizeof(double) * 3 * nline); gradX = line; KaX = line + nline; oKaX = line + 2 * nline; }
#define GX(i) gradX[(i) - IminS]
#define KX(i) KaX[(i) - IminS]
#define OX(i) oKaX[(i) - IminS]
        for (int j = Jstr; j <= Jend; j++) {
          for (int i = IstrU - 1; i <= Iendp2; i++) {
            const double cff = 0.125 * (pm(i - 1, j) + pm(i, j)) * (pn(i - 1, j) + pn(i, j)) * dt;
            const double cff1 = cff * (oHz(i - 1, j, k) + oHz(i, j, k));
            GX(i) = t(i, j, k, 3, itrc) - t(i - 1, j, k, 3, itrc);
            KX(i) = 1.0 - fabs(Huon(i, j, k) * cff1);
            if (p->masking) { GX(i) = GX(i) * umask(i, j); KX(i) = KX(i) * umask(i, j); }   /* MASKING, :448 */
          }
          if (!EWperiodic) {
            if (west_edge && Huon(Istr, j, k) >= 0.0) { GX(Istr - 1) = 0.0; KX(Istr - 1) = 0.0; }
            if (east_edge && Huon(Iend + 1, j, k) < 0.0) { GX(Iend + 2) = 0.0; KX(Iend + 2) = 0.0; }
          }
          for (int i = Istr; i <= Iend + 1; i++) {
            double sw_xi, cff;
            if (KX(i) <= eps1) OX(i) = 0.0;
            else OX(i) = 1.0 / MAX(KX(i), eps1);
            if (Huon(i, j, k) >= 0.0) {
              double rL, rkaL;
              if (fabs(GX(i)) <= eps1) { rL = 0.0; rkaL = 0.0; }
              else { rL = GX(i - 1) / GX(i); rkaL = KX(i - 1) * OX(i); }
              const double a1 = cc1 * KX(i) + cc2 - cc3 * OX(i);
              const double b1 = -cc1 * KX(i) + cc2 + cc3 * OX(i);
              const double betaL = a1 + b1 * rL;
              cff = 0.5 * MAX(0.0, MIN(MIN(2.0, 2.0 * rL * rkaL), betaL)) * GX(i) * KX(i);
              if (p->masking) cff = cff * rmask(MAX(i - 2, 0), j);                           /* MASKING, :487 */
              sw_xi = t(i - 1, j, k, 3, itrc) + cff;
            } else {
              double rR, rkaR;
              if (fabs(GX(i)) <= eps1) { rR = 0.0; rkaR = 0.0; }
              else { rR = GX(i + 1) / GX(i); rkaR = KX(i + 1) * OX(i); }
              const double a1 = cc1 * KX(i) + cc2 - cc3 * OX(i);
              const double b1 = -cc1 * KX(i) + cc2 + cc3 * OX(i);
              const double betaR = a1 + b1 * rR;
              cff = 0.5 * MAX(0.0, MIN(MIN(2.0, 2.0 * rR * rkaR), betaR)) * GX(i) * KX(i);
              if (p->masking) cff = cff * rmask(MIN(i + 1, Lm + 1), j);                      /* MASKING, :506 */
              sw_xi = t(i, j, k, 3, itrc) - cff;
            }
            FX(i, j) = sw_xi * Huon(i, j, k);
          }
        }
#undef GX
#undef KX
#undef OX
#define GE(j) gradX[(j) - JminS]
#define KE(j) KaX[(j) - JminS]
#define OE(j) oKaX[(j) - JminS]
        for (int i = Istr; i <= Iend; i++) {
          for (int j = JstrV - 1; j <= Jendp2; j++) {
            const double cff = 0.125 * (pn(i, j) + pn(i, j - 1)) * (pm(i, j) + pm(i, j - 1)) * dt;
            const double cff1 = cff * (oHz(i, j, k) + oHz(i, j - 1, k));
            GE(j) = t(i, j, k, 3, itrc) - t(i, j - 1, k, 3, itrc);
            KE(j) = 1.0 - fabs(Hvom(i, j, k) * cff1);
            if (p->masking) { GE(j) = GE(j) * vmask(i, j); KE(j) = KE(j) * vmask(i, j); }   /* MASKING, :523 */
          }
          if (!NSperiodic) {
            if (south_edge && Hvom(i, Jstr, k) >= 0.0) { GE(Jstr - 1) = 0.0; KE(Jstr - 1) = 0.0; }
            if (north_edge && Hvom(i, Jend + 1, k) < 0.0) { GE(Jend + 2) = 0.0; KE(Jend + 2) = 0.0; }
          }
          for (int j = Jstr; j <= Jend + 1; j++) {
            double sw_eta, cff;
            if (KE(j) <= eps1) OE(j) = 0.0;
            else OE(j) = 1.0 / MAX(KE(j), eps1);
            if (Hvom(i, j, k) >= 0.0) {
              double rD, rkaD;
              if (fabs(GE(j)) <= eps1) { rD = 0.0; rkaD = 0.0; }
              else { rD = GE(j - 1) / GE(j); rkaD = KE(j - 1) * OE(j); }
              const double a1 = cc1 * KE(j) + cc2 - cc3 * OE(j);
              const double b1 = -cc1 * KE(j) + cc2 + cc3 * OE(j);
              const double betaD = a1 + b1 * rD;
              cff = 0.5 * MAX(0.0, MIN(MIN(2.0, 2.0 * rD * rkaD), betaD)) * GE(j) * KE(j);
              if (p->masking) cff = cff * rmask(i, MAX(j - 2, 0));                           /* MASKING, :562 */
              sw_eta = t(i, j - 1, k, 3, itrc) + cff;
            } else {
              double rU, rkaU;
              if (fabs(GE(j)) <= eps1) { rU = 0.0; rkaU = 0.0; }
              else { rU = GE(j + 1) / GE(j); rkaU = KE(j + 1) * OE(j); }
              const double a1 = cc1 * KE(j) + cc2 - cc3 * OE(j);
              const double b1 = -cc1 * KE(j) + cc2 + cc3 * OE(j);
              const double betaU = a1 + b1 * rU;
              cff = 0.5 * MAX(0.0, MIN(MIN(2.0, 2.0 * rU * rkaU), betaU)) * GE(j) * KE(j);
              if (p->masking) cff = cff * rmask(i, MIN(j + 1, Mm + 1));                      /* MASKING, :581 */
              sw_eta = t(i, j, k, 3, itrc) - cff;
            }
            FE(i, j) = sw_eta * Hvom(i, j, k);
          }
        }
#undef GE
#undef KE
#undef OE
        free(line);
      } else if (ha == ADV_C2) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++)
            FX(i, j) = Huon(i, j, k) * 0.5 * (t(i - 1, j, k, 3, itrc) + t(i, j, k, 3, itrc));
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++)
            FE(i, j) = Hvom(i, j, k) * 0.5 * (t(i, j - 1, k, 3, itrc) + t(i, j, k, 3, itrc));
      } else {
        /* A4 / C4 / SU3 / U3 -- step3d_t.F:596-828 */
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istrm1; i <= Iendp2; i++) {
            FX(i, j) = t(i, j, k, 3, itrc) - t(i - 1, j, k, 3, itrc);
            if (p->masking) FX(i, j) = FX(i, j) * umask(i, j);                     /* MASKING, step3d_t.F:603 */
          }
        if (!EWperiodic) {
          if (west_edge) for (int j = Jstr; j <= Jend; j++) FX(Istr - 1, j) = FX(Istr, j);
          if (east_edge) for (int j = Jstr; j <= Jend; j++) FX(Iend + 2, j) = FX(Iend + 1, j);
        }
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) {
            if (ha == ADV_U3) {
              curv(i, j) = FX(i + 1, j) - FX(i, j);
            } else if (ha == ADV_A4) {
              double cff = 2.0 * FX(i + 1, j) * FX(i, j);
              if (cff > eps) grad(i, j) = cff / (FX(i + 1, j) + FX(i, j));
              else grad(i, j) = 0.0;
            } else {
              grad(i, j) = 0.5 * (FX(i + 1, j) + FX(i, j));
            }
          }
        double cff1 = 1.0 / 6.0, cff2 = 1.0 / 3.0;
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            if (ha == ADV_U3) {
              FX(i, j) = Huon(i, j, k) * 0.5 * (t(i - 1, j, k, 3, itrc) + t(i, j, k, 3, itrc)) -
                         cff1 * (curv(i - 1, j) * MAX(Huon(i, j, k), 0.0) +
                                 curv(i, j) * MIN(Huon(i, j, k), 0.0));
            } else {
              FX(i, j) = Huon(i, j, k) * 0.5 *
                         (t(i - 1, j, k, 3, itrc) + t(i, j, k, 3, itrc) -
                          cff2 * (grad(i, j) - grad(i - 1, j)));
            }
          }
        for (int j = Jstrm1; j <= Jendp2; j++)
          for (int i = Istr; i <= Iend; i++) {
            FE(i, j) = t(i, j, k, 3, itrc) - t(i, j - 1, k, 3, itrc);
            if (p->masking) FE(i, j) = FE(i, j) * vmask(i, j);                     /* MASKING, step3d_t.F:667 */
          }
        if (!NSperiodic) {
          if (south_edge) for (int i = Istr; i <= Iend; i++) FE(i, Jstr - 1) = FE(i, Jstr);
          if (north_edge) for (int i = Istr; i <= Iend; i++) FE(i, Jend + 2) = FE(i, Jend + 1);
        }
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            if (ha == ADV_U3) {
              curv(i, j) = FE(i, j + 1) - FE(i, j);
            } else if (ha == ADV_A4) {
              double cff = 2.0 * FE(i, j + 1) * FE(i, j);
              if (cff > eps) grad(i, j) = cff / (FE(i, j + 1) + FE(i, j));
              else grad(i, j) = 0.0;
            } else {
              grad(i, j) = 0.5 * (FE(i, j + 1) + FE(i, j));
            }
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            if (ha == ADV_U3) {
              FE(i, j) = Hvom(i, j, k) * 0.5 * (t(i, j - 1, k, 3, itrc) + t(i, j, k, 3, itrc)) -
                         cff1 * (curv(i, j - 1) * MAX(Hvom(i, j, k), 0.0) +
                                 curv(i, j) * MIN(Hvom(i, j, k), 0.0));
            } else {
              FE(i, j) = Hvom(i, j, k) * 0.5 *
                         (t(i, j - 1, k, 3, itrc) + t(i, j, k, 3, itrc) -
                          cff2 * (grad(i, j) - grad(i, j - 1)));
            }
          }
      }
      o_src_tflux(b, p, s, F, itrc, k, FX_, FE_, ha == ADV_HSIMT, 0);           /* LuvSrc, :734-799 */
      /* HADV_STEPPING, step3d_t.F:831-875 */
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          double cff = dt * pm(i, j) * pn(i, j);
          double cff1 = cff * (FX(i + 1, j) - FX(i, j));
          double cff2 = cff * (FE(i, j + 1) - FE(i, j));
          double cff3 = cff1 + cff2;
          t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) - cff3;
        }
    }
  }

  /* T_LOOP2 / J_LOOP1: vertical advection, step3d_t.F:883-1210 */
  for (int itrc = 1; itrc <= NT; itrc++) {
    const int va = p->Vadv[itrc - 1];
    const int JminT = (va == ADV_MPDATA) ? JstrVm2 : Jstr, JmaxT = (va == ADV_MPDATA) ? Jendp2i : Jend;
    for (int j = JminT; j <= JmaxT; j++) {
      if (va == ADV_MPDATA) {
        /* first-order upstream vertical flux, :1002-1018; Ta in Tunits, :1168-1177 */
        for (int i = IstrUm2; i <= Iendp2i; i++) {
          for (int k = 1; k <= N - 1; k++) {
            double cff1 = MAX(W(i, j, k), 0.0), cff2 = MIN(W(i, j, k), 0.0);
            FC(i, k) = cff1 * t(i, j, k, 3, itrc) + cff2 * t(i, j, k + 1, 3, itrc);
          }
          FC(i, 0) = 0.0;
          FC(i, N) = 0.0;
        }
        o_src_wtracer(b, p, s, F, itrc, 1, j, Ta_, oHz_);                        /* LwSrc, :1136-1158 */
        for (int i = IstrUm2; i <= Iendp2i; i++) CF(i, 0) = dt * pm(i, j) * pn(i, j);
        for (int k = 1; k <= N; k++)
          for (int i = IstrUm2; i <= Iendp2i; i++) {
            double cff1 = CF(i, 0) * (FC(i, k) - FC(i, k - 1));
            Ta(i, j, k, itrc) = (Ta(i, j, k, itrc) - cff1) * oHz(i, j, k);
          }
        continue;
      }
      if (va == ADV_HSIMT) {
        /* HSIMT with TVD limiter, vertical flux, step3d_t.F:1022-1090 */
        double KaZ[512], oKaZ[512], gradZ[512];
        if (N + 1 > 512) return 8;
        for (int i = Istr; i <= Iend; i++) {
          KaZ[0] = 0.0; oKaZ[0] = 0.0; gradZ[0] = 0.0;
          for (int k = 1; k <= N - 1; k++) {
            const double cff = pm(i, j) * pn(i, j) * dt;
            KaZ[k] = 1.0 - fabs(cff * W(i, j, k) / (z_r(i, j, k + 1) - z_r(i, j, k)));
            oKaZ[k] = 1.0 / KaZ[k];
            gradZ[k] = t(i, j, k + 1, 3, itrc) - t(i, j, k, 3, itrc);
          }
          KaZ[N] = 0.0; oKaZ[N] = 0.0; gradZ[N] = 0.0;
          for (int k = 1; k <= N - 1; k++) {
            if ((k == 1) && (W(i, j, k) >= 0.0)) {
              FC(i, k) = W(i, j, k) * t(i, j, k, 3, itrc);
            } else if ((k == N - 1) && (W(i, j, k) < 0.0)) {
              FC(i, k) = W(i, j, k) * t(i, j, k + 1, 3, itrc);
            } else {
              double sw, cff;
              if (W(i, j, k) >= 0) {
                double rD, rkaD;
                if (fabs(gradZ[k]) <= eps1) { rD = 0.0; rkaD = 0.0; }
                else { rD = gradZ[k - 1] / gradZ[k]; rkaD = KaZ[k - 1] * oKaZ[k]; }
                const double a1 = cc1 * KaZ[k] + cc2 - cc3 * oKaZ[k];
                const double b1 = -cc1 * KaZ[k] + cc2 + cc3 * oKaZ[k];
                const double betaD = a1 + b1 * rD;
                cff = 0.5 * MAX(0.0, MIN(MIN(2.0, 2.0 * rD * rkaD), betaD)) * gradZ[k] * KaZ[k];
                sw = t(i, j, k, 3, itrc) + cff;
              } else {
                double rU, rkaU;
                if (fabs(gradZ[k]) <= eps1) { rU = 0.0; rkaU = 0.0; }
                else { rU = gradZ[k + 1] / gradZ[k]; rkaU = KaZ[k + 1] * oKaZ[k]; }
                const double a1 = cc1 * KaZ[k] + cc2 - cc3 * oKaZ[k];
                const double b1 = -cc1 * KaZ[k] + cc2 + cc3 * oKaZ[k];
                const double betaU = a1 + b1 * rU;
                cff = 0.5 * MAX(0.0, MIN(MIN(2.0, 2.0 * rU * rkaU), betaU)) * gradZ[k] * KaZ[k];
                sw = t(i, j, k + 1, 3, itrc) - cff;
              }
              FC(i, k) = W(i, j, k) * sw;
            }
          }
          FC(i, 0) = 0.0;
          FC(i, N) = 0.0;
        }
      } else if (va == ADV_SPLINES) {
        for (int i = Istr; i <= Iend; i++) {
          FC(i, 0) = 2.0 * t(i, j, 1, 3, itrc);
          CF(i, 1) = 1.0;
        }
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) {
            double cff = 1.0 / (2.0 * Hz(i, j, k) + Hz(i, j, k + 1) * (2.0 - CF(i, k)));
            CF(i, k + 1) = cff * Hz(i, j, k);
            FC(i, k) = cff * (3.0 * (Hz(i, j, k) * t(i, j, k + 1, 3, itrc) +
                                     Hz(i, j, k + 1) * t(i, j, k, 3, itrc)) -
                              Hz(i, j, k + 1) * FC(i, k - 1));
          }
        for (int i = Istr; i <= Iend; i++)
          FC(i, N) = (2.0 * t(i, j, N, 3, itrc) - FC(i, N - 1)) / (1.0 - CF(i, N));
        for (int k = N - 1; k >= 0; k--)
          for (int i = Istr; i <= Iend; i++) {
            FC(i, k) = FC(i, k) - CF(i, k + 1) * FC(i, k + 1);
            FC(i, k + 1) = W(i, j, k + 1) * FC(i, k + 1);
          }
        for (int i = Istr; i <= Iend; i++) {
          FC(i, N) = 0.0;
          FC(i, 0) = 0.0;
        }
      } else if (va == ADV_A4) {
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = t(i, j, k + 1, 3, itrc) - t(i, j, k, 3, itrc);
        for (int i = Istr; i <= Iend; i++) {
          FC(i, 0) = FC(i, 1);
          FC(i, N) = FC(i, N - 1);
        }
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            double cff = 2.0 * FC(i, k) * FC(i, k - 1);
            if (cff > eps) CF(i, k) = cff / (FC(i, k) + FC(i, k - 1));
            else CF(i, k) = 0.0;
          }
        double cff1 = 1.0 / 3.0;
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = W(i, j, k) * 0.5 *
                       (t(i, j, k, 3, itrc) + t(i, j, k + 1, 3, itrc) -
                        cff1 * (CF(i, k + 1) - CF(i, k)));
        for (int i = Istr; i <= Iend; i++) {
          FC(i, 0) = 0.0;
          FC(i, N) = 0.0;
        }
      } else if (va == ADV_C2) {
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = W(i, j, k) * 0.5 * (t(i, j, k, 3, itrc) + t(i, j, k + 1, 3, itrc));
        for (int i = Istr; i <= Iend; i++) {
          FC(i, 0) = 0.0;
          FC(i, N) = 0.0;
        }
      } else { /* C4 / SU3, step3d_t.F:1094+ */
        double cff1 = 0.5, cff2 = 7.0 / 12.0, cff3 = 1.0 / 12.0;
        for (int k = 2; k <= N - 2; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = W(i, j, k) *
                       (cff2 * (t(i, j, k, 3, itrc) + t(i, j, k + 1, 3, itrc)) -
                        cff3 * (t(i, j, k - 1, 3, itrc) + t(i, j, k + 2, 3, itrc)));
        for (int i = Istr; i <= Iend; i++) {
          FC(i, 0) = 0.0;
          FC(i, 1) = W(i, j, 1) *
                     (cff1 * t(i, j, 1, 3, itrc) + cff2 * t(i, j, 2, 3, itrc) -
                      cff3 * t(i, j, 3, 3, itrc));
          FC(i, N - 1) = W(i, j, N - 1) *
                         (cff1 * t(i, j, N, 3, itrc) + cff2 * t(i, j, N - 1, 3, itrc) -
                          cff3 * t(i, j, N - 2, 3, itrc));
          FC(i, N) = 0.0;
        }
      }
      /* VADV_STEPPING, step3d_t.F:1168-1208 */
      for (int i = Istr; i <= Iend; i++) CF(i, 0) = dt * pm(i, j) * pn(i, j);
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          double cff1 = CF(i, 0) * (FC(i, k) - FC(i, k - 1));
          t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) - cff1;
          if (p->splines_vdiff) t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) * oHz(i, j, k);      /* :1196-1198 */
        }
    }
  }

  /* T_LOOP3: MPDATA anti-diffusive correction, step3d_t.F:1217-1318 */
  for (int itrc = 1; itrc <= NT; itrc++) {
    if (p->Hadv[itrc - 1] != ADV_MPDATA) continue;
    int rc = oracle_mpdata_adiff(b, p, s, F, oHz_, &t(LBi, LBj, 1, 3, itrc), &Ta(IminS, JminS, 1, itrc),
                                 Ua_, Va_, Wa_);
    if (rc) return rc;
    for (int k = 1; k <= N; k++) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          double cff1 = MAX(Ua(i, j, k), 0.0), cff2 = MIN(Ua(i, j, k), 0.0);
          FX(i, j) = (cff1 * Ta(i - 1, j, k, itrc) + cff2 * Ta(i, j, k, itrc)) *
                     0.5 * (Hz(i, j, k) + Hz(i - 1, j, k)) * on_u(i, j);
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend; i++) {
          double cff1 = MAX(Va(i, j, k), 0.0), cff2 = MIN(Va(i, j, k), 0.0);
          FE(i, j) = (cff1 * Ta(i, j - 1, k, itrc) + cff2 * Ta(i, j, k, itrc)) *
                     0.5 * (Hz(i, j, k) + Hz(i, j - 1, k)) * om_v(i, j);
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          double cff = dt * pm(i, j) * pn(i, j);
          double cff1 = cff * (FX(i + 1, j) - FX(i, j));
          double cff2 = cff * (FE(i, j + 1) - FE(i, j));
          double cff3 = cff1 + cff2;
          t(i, j, k, nnew, itrc) = Ta(i, j, k, itrc) * Hz(i, j, k) - cff3;
        }
    }
    for (int j = Jstr; j <= Jend; j++) {
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          double cff1 = MAX(Wa(i, j, k), 0.0), cff2 = MIN(Wa(i, j, k), 0.0);
          FC(i, k) = cff1 * Ta(i, j, k, itrc) + cff2 * Ta(i, j, k + 1, itrc);
        }
      for (int i = Istr; i <= Iend; i++) {
        FC(i, 0) = 0.0;
        FC(i, N) = 0.0;
      }
      for (int i = Istr; i <= Iend; i++) CF(i, 0) = dt * pm(i, j) * pn(i, j);
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          double cff1 = CF(i, 0) * (FC(i, k) - FC(i, k - 1));
          t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) - cff1;      /* stays in m Tunits */
        }
    }
  }

  /* LwSrc: the tracer that comes with the volume of a cell-centred source, :1321-1360 */
  for (int itrc = 1; itrc <= NT; itrc++)
    if (!(p->Hadv[itrc - 1] == ADV_MPDATA && p->Vadv[itrc - 1] == ADV_MPDATA))
      o_src_wtracer(b, p, s, F, itrc, 0, 0, NULL, oHz_);

  /* J_LOOP2: implicit vertical diffusion, step3d_t.F:1363-1560 */
  for (int j = Jstr; j <= Jend; j++) {
    for (int itrc = 1; itrc <= NT; itrc++) {
      const int ltrc = MIN(NAT, itrc);
      if (p->Hadv[itrc - 1] == ADV_MPDATA || !p->splines_vdiff) {
        /* classic tridiagonal: without SPLINES_VDIFF, and for MPDATA tracers under it as well, :1431-1501 */
        double cff = -dt * p->lambda;
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++) {
            double cff1 = 1.0 / (z_r(i, j, k + 1) - z_r(i, j, k));
            FC(i, k) = cff * cff1 * Akt(i, j, k, ltrc);
          }
        for (int i = Istr; i <= Iend; i++) {
          FC(i, 0) = 0.0;
          FC(i, N) = 0.0;
        }
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            BC(i, k) = Hz(i, j, k) - FC(i, k) - FC(i, k - 1);
            DC(i, k) = t(i, j, k, nnew, itrc);
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
          t(i, j, N, nnew, itrc) = DC(i, N);
        }
        for (int k = N - 1; k >= 1; k--)
          for (int i = Istr; i <= Iend; i++) {
            DC(i, k) = DC(i, k) - CF(i, k) * DC(i, k + 1);
            t(i, j, k, nnew, itrc) = DC(i, k);
          }
        continue;
      }
      double cff1 = 1.0 / 6.0;
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          FC(i, k) = cff1 * Hz(i, j, k) - dt * Akt(i, j, k - 1, ltrc) * oHz(i, j, k);
          CF(i, k) = cff1 * Hz(i, j, k + 1) - dt * Akt(i, j, k + 1, ltrc) * oHz(i, j, k + 1);
        }
      for (int i = Istr; i <= Iend; i++) {
        CF(i, 0) = 0.0;
        DC(i, 0) = 0.0;
      }
      cff1 = 1.0 / 3.0;
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          BC(i, k) = cff1 * (Hz(i, j, k) + Hz(i, j, k + 1)) +
                     dt * Akt(i, j, k, ltrc) * (oHz(i, j, k) + oHz(i, j, k + 1));
          double cff = 1.0 / (BC(i, k) - FC(i, k) * CF(i, k - 1));
          CF(i, k) = cff * CF(i, k);
          DC(i, k) = cff * (t(i, j, k + 1, nnew, itrc) - t(i, j, k, nnew, itrc) -
                            FC(i, k) * DC(i, k - 1));
        }
      for (int i = Istr; i <= Iend; i++) DC(i, N) = 0.0;
      for (int k = N - 1; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) DC(i, k) = DC(i, k) - CF(i, k) * DC(i, k + 1);
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          DC(i, k) = DC(i, k) * Akt(i, j, k, ltrc);
          double c1 = dt * oHz(i, j, k) * (DC(i, k) - DC(i, k - 1));
          t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) + c1;
        }
    }
  }

  /* lateral BCs + periodic wrap, step3d_t.F:1564-1626 */
  for (int itrc = 1; itrc <= NT; itrc++) {
    o_t3dbc(b, p, s, F, nnew, itrc);
    if (p->masking)                                                       /* apply land/sea mask, step3d_t.F:1586-1596 */
      for (int k = 1; k <= N; k++)
        for (int j = JstrR; j <= JendR; j++)
          for (int i = IstrR; i <= IendR; i++) t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) * rmask(i, j);
    o_exchange3d(b, GT_R, N, &t(LBi, LBj, 1, nnew, itrc));
  }
  free(FX_); free(FE_); free(curv_); free(grad_); free(oHz_);
  free(CF_); free(BC_); free(DC_); free(FC_);
  free(Ta_); free(Ua_); free(Va_); free(Wa_);
  return 0;
}
