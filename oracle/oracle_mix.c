/*
 * oracle_mix.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Harmonic lateral mixing:
 *   t3dmix2_geo_tile  ROMS/Nonlinear/t3dmix2_geo.h:90-424 (MIX_GEO_TS)
 *   t3dmix2_s_tile    ROMS/Nonlinear/t3dmix2_s.h:89-306   (MIX_S_TS)
 *   uv3dmix2_s_tile   ROMS/Nonlinear/uv3dmix2_s.h:114-335 (MIX_S_UV)
 * All three reference files compile stand-alone; pinned against oracle/_ref.
 */
#include "oracle.h"

static int t3dmix2_geo(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, nnew = s->nnew, nstp = s->nstp, stab = p->ts_mix_stability;
  const double dt = p->dt;
  double cff, cff1, cff2, cff3, cff4;
  const long n2 = nis * njs;
  double *FE_ = walloc(n2), *FX_ = walloc(n2), *FS_ = walloc(2 * n2);
  double *dTdz_ = walloc(2 * n2), *dTdx_ = walloc(2 * n2), *dTde_ = walloc(2 * n2);
  double *dZdx_ = walloc(2 * n2), *dZde_ = walloc(2 * n2);
#define FE(i,j) FE_[WS2(i,j)]
#define FX(i,j) FX_[WS2(i,j)]
#define FS(i,j,k) FS_[WS2(i,j) + ((k)-1) * n2]
#define dTdz(i,j,k) dTdz_[WS2(i,j) + ((k)-1) * n2]
#define dTdx(i,j,k) dTdx_[WS2(i,j) + ((k)-1) * n2]
#define dTde(i,j,k) dTde_[WS2(i,j) + ((k)-1) * n2]
#define dZdx(i,j,k) dZdx_[WS2(i,j) + ((k)-1) * n2]
#define dZde(i,j,k) dZde_[WS2(i,j) + ((k)-1) * n2]
  for (int itrc = 1; itrc <= NT; itrc++) {
    int k1, k2 = 1;
    for (int k = 0; k <= N; k++) {
      k1 = k2;
      k2 = 3 - k1;
      if (k < N) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            cff = 0.5 * (pm(i, j) + pm(i - 1, j));
            if (p->masking) cff = cff * umask(i, j);                                  /* MASKING, t3dmix2_geo.h:228 */
            if (p->wet_dry) cff = cff * umask_wet(i, j);           /* WET_DRY: the next block of the same file */
            dZdx(i, j, k2) = cff * (z_r(i, j, k + 1) - z_r(i - 1, j, k + 1));
            dTdx(i, j, k2) = cff * o_tdiff(stab, t(i, j, k + 1, nrhs, itrc), t(i - 1, j, k + 1, nrhs, itrc),   /* TS_MIX_STABILITY, :236 */
                                     t(i, j, k + 1, nstp, itrc), t(i - 1, j, k + 1, nstp, itrc));
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 0.5 * (pn(i, j) + pn(i, j - 1));
            if (p->masking) cff = cff * vmask(i, j);                                  /* MASKING, t3dmix2_geo.h:260 */
            if (p->wet_dry) cff = cff * vmask_wet(i, j);           /* WET_DRY: the next block of the same file */
            dZde(i, j, k2) = cff * (z_r(i, j, k + 1) - z_r(i, j - 1, k + 1));
            dTde(i, j, k2) = cff * o_tdiff(stab, t(i, j, k + 1, nrhs, itrc), t(i, j - 1, k + 1, nrhs, itrc),   /* :268 */
                                     t(i, j, k + 1, nstp, itrc), t(i, j - 1, k + 1, nstp, itrc));
          }
      }
      if (k == 0 || k == N) {
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) { dTdz(i, j, k2) = 0.0; FS(i, j, k2) = 0.0; }
      } else {
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) {
            cff = 1.0 / (z_r(i, j, k + 1) - z_r(i, j, k));
            dTdz(i, j, k2) = cff * o_tdiff(stab, t(i, j, k + 1, nrhs, itrc), t(i, j, k, nrhs, itrc),           /* :301 */
                                     t(i, j, k + 1, nstp, itrc), t(i, j, k, nstp, itrc));
          }
      }
      if (k > 0) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            cff = 0.25 * (diff2(i, j, itrc) + diff2(i - 1, j, itrc)) * on_u(i, j);
            FX(i, j) = cff * (Hz(i, j, k) + Hz(i - 1, j, k)) *
                       (dTdx(i, j, k1) -
                        0.5 * (MIN(dZdx(i, j, k1), 0.0) * (dTdz(i - 1, j, k1) + dTdz(i, j, k2)) +
                               MAX(dZdx(i, j, k1), 0.0) * (dTdz(i - 1, j, k2) + dTdz(i, j, k1))));
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 0.25 * (diff2(i, j, itrc) + diff2(i, j - 1, itrc)) * om_v(i, j);
            FE(i, j) = cff * (Hz(i, j, k) + Hz(i, j - 1, k)) *
                       (dTde(i, j, k1) -
                        0.5 * (MIN(dZde(i, j, k1), 0.0) * (dTdz(i, j - 1, k1) + dTdz(i, j, k2)) +
                               MAX(dZde(i, j, k1), 0.0) * (dTdz(i, j - 1, k2) + dTdz(i, j, k1))));
          }
        if (k < N) {
          for (int j = Jstr; j <= Jend; j++)
            for (int i = Istr; i <= Iend; i++) {
              cff = 0.5 * diff2(i, j, itrc);
              cff1 = MIN(dZdx(i, j, k1), 0.0);
              cff2 = MIN(dZdx(i + 1, j, k2), 0.0);
              cff3 = MAX(dZdx(i, j, k2), 0.0);
              cff4 = MAX(dZdx(i + 1, j, k1), 0.0);
              FS(i, j, k2) = cff * (cff1 * (cff1 * dTdz(i, j, k2) - dTdx(i, j, k1)) +
                                    cff2 * (cff2 * dTdz(i, j, k2) - dTdx(i + 1, j, k2)) +
                                    cff3 * (cff3 * dTdz(i, j, k2) - dTdx(i, j, k2)) +
                                    cff4 * (cff4 * dTdz(i, j, k2) - dTdx(i + 1, j, k1)));
              cff1 = MIN(dZde(i, j, k1), 0.0);
              cff2 = MIN(dZde(i, j + 1, k2), 0.0);
              cff3 = MAX(dZde(i, j, k2), 0.0);
              cff4 = MAX(dZde(i, j + 1, k1), 0.0);
              FS(i, j, k2) = FS(i, j, k2) +
                             cff * (cff1 * (cff1 * dTdz(i, j, k2) - dTde(i, j, k1)) +
                                    cff2 * (cff2 * dTdz(i, j, k2) - dTde(i, j + 1, k2)) +
                                    cff3 * (cff3 * dTdz(i, j, k2) - dTde(i, j, k2)) +
                                    cff4 * (cff4 * dTdz(i, j, k2) - dTde(i, j + 1, k1)));
            }
        }
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = dt * pm(i, j) * pn(i, j);
            cff1 = cff * (FX(i + 1, j) - FX(i, j));
            cff2 = cff * (FE(i, j + 1) - FE(i, j));
            cff3 = dt * (FS(i, j, k2) - FS(i, j, k1));
            cff4 = cff1 + cff2 + cff3;
            t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) + cff4;
          }
      }
    }
  }
  free(FE_); free(FX_); free(FS_); free(dTdz_); free(dTdx_); free(dTde_); free(dZdx_); free(dZde_);
#undef FE
#undef FX
#undef FS
  return 0;
}

static int t3dmix2_s(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, nnew = s->nnew, nstp = s->nstp, stab = p->ts_mix_stability;
  const double dt = p->dt;
  double cff, cff1, cff2, cff3;
  double *FE_ = walloc(nis * njs), *FX_ = walloc(nis * njs);
#define FE(i,j) FE_[WS2(i,j)]
#define FX(i,j) FX_[WS2(i,j)]
  for (int itrc = 1; itrc <= NT; itrc++)
    for (int k = 1; k <= N; k++) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          cff = 0.25 * (diff2(i, j, itrc) + diff2(i - 1, j, itrc)) * pmon_u(i, j);
          FX(i, j) = cff * (Hz(i, j, k) + Hz(i - 1, j, k)) *
                     o_tdiff(stab, t(i, j, k, nrhs, itrc), t(i - 1, j, k, nrhs, itrc),         /* TS_MIX_STABILITY, :212 */
                             t(i, j, k, nstp, itrc), t(i - 1, j, k, nstp, itrc));
          if (p->masking) FX(i, j) = FX(i, j) * umask(i, j);                          /* MASKING, t3dmix2_s.h:235 */
          if (p->wet_dry) FX(i, j) = FX(i, j) * umask_wet(i, j);           /* WET_DRY: the next block of the same file */
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 0.25 * (diff2(i, j, itrc) + diff2(i, j - 1, itrc)) * pnom_v(i, j);
          FE(i, j) = cff * (Hz(i, j, k) + Hz(i, j - 1, k)) *
                     o_tdiff(stab, t(i, j, k, nrhs, itrc), t(i, j - 1, k, nrhs, itrc),         /* :252 */
                             t(i, j, k, nstp, itrc), t(i, j - 1, k, nstp, itrc));
          if (p->masking) FE(i, j) = FE(i, j) * vmask(i, j);                          /* MASKING, t3dmix2_s.h:275 */
          if (p->wet_dry) FE(i, j) = FE(i, j) * vmask_wet(i, j);           /* WET_DRY: the next block of the same file */
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = dt * pm(i, j) * pn(i, j);
          cff1 = cff * (FX(i + 1, j) - FX(i, j));
          cff2 = cff * (FE(i, j + 1) - FE(i, j));
          cff3 = cff1 + cff2;
          t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) + cff3;
        }
    }
  free(FE_); free(FX_);
#undef FE
#undef FX
  return 0;
}

int oracle_t3dmix2(OARGS)
{
  if (p->mix_iso_ts) return oracle_t3dmix2_iso(b, p, s, F);      /* oracle_mix4.c */
  if (p->mix_geo_ts) return t3dmix2_geo(b, p, s, F);
  if (p->mix_s_ts) return t3dmix2_s(b, p, s, F);
  return 8;
}

int oracle_uv3dmix2(OARGS)
{
  if (p->uv_vis2 == 2) return oracle_uv3dmix2_geo(b, p, s, F);     /* MIX_GEO_UV: oracle_uvmix_geo.c */
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, nnew = s->nnew;
  const double dt = p->dt;
  double cff, cff1, cff2, cff3;
  double *UFe_ = walloc(nis * njs), *VFe_ = walloc(nis * njs), *UFx_ = walloc(nis * njs), *VFx_ = walloc(nis * njs);
#define UFe(i,j) UFe_[WS2(i,j)]
#define VFe(i,j) VFe_[WS2(i,j)]
#define UFx(i,j) UFx_[WS2(i,j)]
#define VFx(i,j) VFx_[WS2(i,j)]
  for (int k = 1; k <= N; k++) {
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = Hz(i, j, k) * 0.5 *
              (pmon_r(i, j) * ((pn(i, j) + pn(i + 1, j)) * u(i + 1, j, k, nrhs) - (pn(i - 1, j) + pn(i, j)) * u(i, j, k, nrhs)) -
               pnom_r(i, j) * ((pm(i, j) + pm(i, j + 1)) * v(i, j + 1, k, nrhs) - (pm(i, j - 1) + pm(i, j)) * v(i, j, k, nrhs)));
        UFx(i, j) = on_r(i, j) * on_r(i, j) * visc2_r(i, j) * cff;
        VFe(i, j) = om_r(i, j) * om_r(i, j) * visc2_r(i, j) * cff;
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff = 0.125 * (Hz(i - 1, j, k) + Hz(i, j, k) + Hz(i - 1, j - 1, k) + Hz(i, j - 1, k)) *
              (pmon_p(i, j) * ((pn(i, j - 1) + pn(i, j)) * v(i, j, k, nrhs) - (pn(i - 1, j - 1) + pn(i - 1, j)) * v(i - 1, j, k, nrhs)) +
               pnom_p(i, j) * ((pm(i - 1, j) + pm(i, j)) * u(i, j, k, nrhs) - (pm(i - 1, j - 1) + pm(i, j - 1)) * u(i, j - 1, k, nrhs)));
        if (p->masking) cff = cff * pmask(i, j);                                      /* MASKING, uv3dmix2_s.h:272 */
        if (p->wet_dry) cff = cff * pmask_wet(i, j);           /* WET_DRY: the next block of the same file */
        UFe(i, j) = om_p(i, j) * om_p(i, j) * visc2_p(i, j) * cff;
        VFx(i, j) = on_p(i, j) * on_p(i, j) * visc2_p(i, j) * cff;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = dt * 0.25 * (pm(i - 1, j) + pm(i, j)) * (pn(i - 1, j) + pn(i, j));
        cff1 = 0.5 * (pn(i - 1, j) + pn(i, j)) * (UFx(i, j) - UFx(i - 1, j));
        cff2 = 0.5 * (pm(i - 1, j) + pm(i, j)) * (UFe(i, j + 1) - UFe(i, j));
        cff3 = cff * (cff1 + cff2);
        rufrc(i, j) = rufrc(i, j) + cff1 + cff2;
        u(i, j, k, nnew) = u(i, j, k, nnew) + cff3;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = dt * 0.25 * (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1));
        cff1 = 0.5 * (pn(i, j - 1) + pn(i, j)) * (VFx(i + 1, j) - VFx(i, j));
        cff2 = 0.5 * (pm(i, j - 1) + pm(i, j)) * (VFe(i, j) - VFe(i, j - 1));
        cff3 = cff * (cff1 - cff2);
        rvfrc(i, j) = rvfrc(i, j) + cff1 - cff2;
        v(i, j, k, nnew) = v(i, j, k, nnew) + cff3;
      }
  }
  free(UFe_); free(VFe_); free(UFx_); free(VFx_);
  return 0;
}
