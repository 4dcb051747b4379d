/*
 * oracle_mix4.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Biharmonic lateral mixing (TS_DIF4, UV_VIS4): the harmonic operator applied twice, with the reference's rule for
 * the first result on a physical edge in between.  The coefficients are the square roots the reference stores
 * (inp_par.F:986, read_phypar.F:6905).
 *   t3dmix4_s_tile    ROMS/Nonlinear/t3dmix4_s.h:100-480   (MIX_S_TS)
 *   t3dmix4_geo_tile  ROMS/Nonlinear/t3dmix4_geo.h:104-784 (MIX_GEO_TS)
 *   uv3dmix4_s_tile   ROMS/Nonlinear/uv3dmix4_s.h:120-629  (MIX_S_UV)
 * and the isopycnal variants of the tracer operators (MIX_ISO_TS; default slope treatment, i.e. none of
 * TS_MIX_MAX_SLOPE / TS_MIX_MIN_STRAT / TS_MIX_STABILITY / TS_MIX_CLIMA):
 *   t3dmix2_iso_tile  ROMS/Nonlinear/t3dmix2_iso.h:100-443
 *   t3dmix4_iso_tile  ROMS/Nonlinear/t3dmix4_iso.h:104-814
 * All reference files compile stand-alone; pinned against oracle/_ref/<APP>_DIF4 and <APP>_ISO.
 */
#include "oracle.h"
#include <stdlib.h>

#define diff4(i,j,it) F->diff4[I2(i,j) + (long)((it)-1) * nij]
#define visc4_p(i,j)  F->visc4_p[I2(i,j)]
#define visc4_r(i,j)  F->visc4_r[I2(i,j)]

/* the range of the first operator: one point beyond the tile, inside the grid (t3dmix4_s.h:262-275) */
#define FIRST_RANGE                                                                   \
  int Imin, Imax, Jmin, Jmax;                                                         \
  if (EWperiodic) { Imin = Istr - 1; Imax = Iend + 1; }                               \
  else { Imin = MAX(Istr - 1, 1); Imax = MIN(Iend + 1, Lm); }                         \
  if (NSperiodic) { Jmin = Jstr - 1; Jmax = Jend + 1; }                               \
  else { Jmin = MAX(Jstr - 1, 1); Jmax = MIN(Jend + 1, Mm); }

static int t3dmix4_s(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, nnew = s->nnew, nstp = s->nstp, stab = p->ts_mix_stability;
  const double dt = p->dt;
  double cff, cff1, cff2, cff3;
  double *FE_ = walloc(nis * njs), *FX_ = walloc(nis * njs), *LapT_ = walloc(nis * njs);
#define FE(i,j) FE_[WS2(i,j)]
#define FX(i,j) FX_[WS2(i,j)]
#define LapT(i,j) LapT_[WS2(i,j)]
  FIRST_RANGE
  for (int itrc = 1; itrc <= NT; itrc++)
    for (int k = 1; k <= N; k++) {
      /* first harmonic operator, :281-345 */
      for (int j = Jmin; j <= Jmax; j++)
        for (int i = Imin; i <= Imax + 1; i++) {
          cff = 0.25 * (diff4(i, j, itrc) + diff4(i - 1, j, itrc)) * pmon_u(i, j);
          if (p->masking) cff = cff * umask(i, j);
          if (p->wet_dry) cff = cff * umask_wet(i, j);           /* WET_DRY: the next block of the same file */
          FX(i, j) = cff * (Hz(i, j, k) + Hz(i - 1, j, k)) *
                     o_tdiff(stab, t(i, j, k, nrhs, itrc), t(i - 1, j, k, nrhs, itrc),         /* TS_MIX_STABILITY, :262 */
                             t(i, j, k, nstp, itrc), t(i - 1, j, k, nstp, itrc));
        }
      for (int j = Jmin; j <= Jmax + 1; j++)
        for (int i = Imin; i <= Imax; i++) {
          cff = 0.25 * (diff4(i, j, itrc) + diff4(i, j - 1, itrc)) * pnom_v(i, j);
          if (p->masking) cff = cff * vmask(i, j);
          if (p->wet_dry) cff = cff * vmask_wet(i, j);           /* WET_DRY: the next block of the same file */
          FE(i, j) = cff * (Hz(i, j, k) + Hz(i, j - 1, k)) *
                     o_tdiff(stab, t(i, j, k, nrhs, itrc), t(i, j - 1, k, nrhs, itrc),         /* :308 */
                             t(i, j, k, nstp, itrc), t(i, j - 1, k, nstp, itrc));
        }
      for (int j = Jmin; j <= Jmax; j++)
        for (int i = Imin; i <= Imax; i++) {
          cff = 1.0 / Hz(i, j, k);
          LapT(i, j) = pm(i, j) * pn(i, j) * cff * (FX(i + 1, j) - FX(i, j) + FE(i, j + 1) - FE(i, j));
        }
      /* physical edges: closed or gradient, :347-405 */
      if (!EWperiodic) {
        if (west_edge) {
          const int closed = o_lbc(p, LBS_WEST, LBV_T) == LBC_CLOSED;
          for (int j = Jmin; j <= Jmax; j++) LapT(Istr - 1, j) = closed ? 0.0 : LapT(Istr, j);
        }
        if (east_edge) {
          const int closed = o_lbc(p, LBS_EAST, LBV_T) == LBC_CLOSED;
          for (int j = Jmin; j <= Jmax; j++) LapT(Iend + 1, j) = closed ? 0.0 : LapT(Iend, j);
        }
      }
      if (!NSperiodic) {
        if (south_edge) {
          const int closed = o_lbc(p, LBS_SOUTH, LBV_T) == LBC_CLOSED;
          for (int i = Imin; i <= Imax; i++) LapT(i, Jstr - 1) = closed ? 0.0 : LapT(i, Jstr);
        }
        if (north_edge) {
          const int closed = o_lbc(p, LBS_NORTH, LBV_T) == LBC_CLOSED;
          for (int i = Imin; i <= Imax; i++) LapT(i, Jend + 1) = closed ? 0.0 : LapT(i, Jend);
        }
      }
      /* second operator and time step, :407-475 */
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          cff = 0.25 * (diff4(i, j, itrc) + diff4(i - 1, j, itrc)) * pmon_u(i, j);
          FX(i, j) = cff * (Hz(i, j, k) + Hz(i - 1, j, k)) * (LapT(i, j) - LapT(i - 1, j));
          if (p->masking) FX(i, j) = FX(i, j) * umask(i, j);
          if (p->wet_dry) FX(i, j) = FX(i, j) * umask_wet(i, j);           /* WET_DRY: the next block of the same file */
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 0.25 * (diff4(i, j, itrc) + diff4(i, j - 1, itrc)) * pnom_v(i, j);
          FE(i, j) = cff * (Hz(i, j, k) + Hz(i, j - 1, k)) * (LapT(i, j) - LapT(i, j - 1));
          if (p->masking) FE(i, j) = FE(i, j) * vmask(i, j);
          if (p->wet_dry) FE(i, j) = FE(i, j) * vmask_wet(i, j);           /* WET_DRY: the next block of the same file */
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = dt * pm(i, j) * pn(i, j);
          cff1 = cff * (FX(i + 1, j) - FX(i, j));
          cff2 = cff * (FE(i, j + 1) - FE(i, j));
          cff3 = cff1 + cff2;
          t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) - cff3;
        }
    }
  free(FE_); free(FX_); free(LapT_);
#undef FE
#undef FX
#undef LapT
  return 0;
}

/* One application of the rotated (geopotential) operator of t3dmix4_geo.h to the 3-D array S (private extents,
 * k = 1..N) on the range (i0:i1, j0:j1): :243-455 with S = t(nrhs) and out = LapT (first = 1), :577-775 with S = LapT
 * and t(nnew) = t(nnew) - dt * (...) (first = 0).  The two blocks of the reference differ in nothing else. */
static void rotated_pass(const roms_bounds_t *b, const roms_params_t *p, const roms_step_idx_t *s, roms_fields_t *F,
                         int itrc, const double *S_, const double *S2_, double *out_, int i0, int i1, int j0, int j1, int first)
{
  ORACLE_PROLOGUE
  const int nnew = s->nnew;
  const double dt = p->dt;
  double cff, cff1, cff2, cff3, cff4;
  const long n2 = nis * njs;
  double *FE_ = walloc(n2), *FX_ = walloc(n2), *FS_ = walloc(2 * n2);
  double *dTdz_ = walloc(2 * n2), *dTdx_ = walloc(2 * n2), *dTde_ = walloc(2 * n2);
  double *dZdx_ = walloc(2 * n2), *dZde_ = walloc(2 * n2);
#define S(i,j,k) S_[WS3(i,j,k)]
/* the difference S(a) - S(b); with TS_MIX_STABILITY (S2_ = t(nstp), first operator only) the weighted one */
#define SD(ia,ja,ka,ib,jb,kb) o_tdiff(S2_ != NULL, S(ia,ja,ka), S(ib,jb,kb), S2_ ? S2_[WS3(ia,ja,ka)] : 0.0, S2_ ? S2_[WS3(ib,jb,kb)] : 0.0)
#define OUT(i,j,k) out_[WS3(i,j,k)]
#define FE(i,j) FE_[WS2(i,j)]
#define FX(i,j) FX_[WS2(i,j)]
#define FS(i,j,k) FS_[WS2(i,j) + ((k)-1) * n2]
#define dTdz(i,j,k) dTdz_[WS2(i,j) + ((k)-1) * n2]
#define dTdx(i,j,k) dTdx_[WS2(i,j) + ((k)-1) * n2]
#define dTde(i,j,k) dTde_[WS2(i,j) + ((k)-1) * n2]
#define dZdx(i,j,k) dZdx_[WS2(i,j) + ((k)-1) * n2]
#define dZde(i,j,k) dZde_[WS2(i,j) + ((k)-1) * n2]
  int k1, k2 = 1;
  for (int k = 0; k <= N; k++) {
    k1 = k2;
    k2 = 3 - k1;
    if (k < N) {
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1 + 1; i++) {
          cff = 0.5 * (pm(i, j) + pm(i - 1, j));
          if (p->masking) cff = cff * umask(i, j);
          if (p->wet_dry) cff = cff * umask_wet(i, j);           /* WET_DRY: the next block of the same file */
          dZdx(i, j, k2) = cff * (z_r(i, j, k + 1) - z_r(i - 1, j, k + 1));
          dTdx(i, j, k2) = cff * SD(i, j, k + 1, i - 1, j, k + 1);
        }
      for (int j = j0; j <= j1 + 1; j++)
        for (int i = i0; i <= i1; i++) {
          cff = 0.5 * (pn(i, j) + pn(i, j - 1));
          if (p->masking) cff = cff * vmask(i, j);
          if (p->wet_dry) cff = cff * vmask_wet(i, j);           /* WET_DRY: the next block of the same file */
          dZde(i, j, k2) = cff * (z_r(i, j, k + 1) - z_r(i, j - 1, k + 1));
          dTde(i, j, k2) = cff * SD(i, j, k + 1, i, j - 1, k + 1);
        }
    }
    if (k == 0 || k == N) {
      for (int j = j0 - 1; j <= j1 + 1; j++)
        for (int i = i0 - 1; i <= i1 + 1; i++) { dTdz(i, j, k2) = 0.0; FS(i, j, k2) = 0.0; }
    } else {
      for (int j = j0 - 1; j <= j1 + 1; j++)
        for (int i = i0 - 1; i <= i1 + 1; i++) {
          cff = 1.0 / (z_r(i, j, k + 1) - z_r(i, j, k));
          dTdz(i, j, k2) = cff * SD(i, j, k + 1, i, j, k);
        }
    }
    if (k > 0) {
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1 + 1; i++) {
          cff = 0.25 * (diff4(i, j, itrc) + diff4(i - 1, j, itrc)) * on_u(i, j);
          FX(i, j) = cff * (Hz(i, j, k) + Hz(i - 1, j, k)) *
                     (dTdx(i, j, k1) -
                      0.5 * (MIN(dZdx(i, j, k1), 0.0) * (dTdz(i - 1, j, k1) + dTdz(i, j, k2)) +
                             MAX(dZdx(i, j, k1), 0.0) * (dTdz(i - 1, j, k2) + dTdz(i, j, k1))));
        }
      for (int j = j0; j <= j1 + 1; j++)
        for (int i = i0; i <= i1; i++) {
          cff = 0.25 * (diff4(i, j, itrc) + diff4(i, j - 1, itrc)) * om_v(i, j);
          FE(i, j) = cff * (Hz(i, j, k) + Hz(i, j - 1, k)) *
                     (dTde(i, j, k1) -
                      0.5 * (MIN(dZde(i, j, k1), 0.0) * (dTdz(i, j - 1, k1) + dTdz(i, j, k2)) +
                             MAX(dZde(i, j, k1), 0.0) * (dTdz(i, j - 1, k2) + dTdz(i, j, k1))));
        }
      if (k < N) {
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            const double difx = 0.5 * diff4(i, j, itrc), dife = difx;
            cff1 = MIN(dZdx(i, j, k1), 0.0);
            cff2 = MIN(dZdx(i + 1, j, k2), 0.0);
            cff3 = MAX(dZdx(i, j, k2), 0.0);
            cff4 = MAX(dZdx(i + 1, j, k1), 0.0);
            FS(i, j, k2) = difx * (cff1 * (cff1 * dTdz(i, j, k2) - dTdx(i, j, k1)) +
                                   cff2 * (cff2 * dTdz(i, j, k2) - dTdx(i + 1, j, k2)) +
                                   cff3 * (cff3 * dTdz(i, j, k2) - dTdx(i, j, k2)) +
                                   cff4 * (cff4 * dTdz(i, j, k2) - dTdx(i + 1, j, k1)));
            cff1 = MIN(dZde(i, j, k1), 0.0);
            cff2 = MIN(dZde(i, j + 1, k2), 0.0);
            cff3 = MAX(dZde(i, j, k2), 0.0);
            cff4 = MAX(dZde(i, j + 1, k1), 0.0);
            FS(i, j, k2) = FS(i, j, k2) +
                           dife * (cff1 * (cff1 * dTdz(i, j, k2) - dTde(i, j, k1)) +
                                   cff2 * (cff2 * dTdz(i, j, k2) - dTde(i, j + 1, k2)) +
                                   cff3 * (cff3 * dTdz(i, j, k2) - dTde(i, j, k2)) +
                                   cff4 * (cff4 * dTdz(i, j, k2) - dTde(i, j + 1, k1)));
          }
      }
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1; i++) {
          if (first) {
            cff = pm(i, j) * pn(i, j);
            cff1 = 1.0 / Hz(i, j, k);
            OUT(i, j, k) = cff1 * (cff * (FX(i + 1, j) - FX(i, j) + FE(i, j + 1) - FE(i, j)) + (FS(i, j, k2) - FS(i, j, k1)));
          } else {
            cff = dt * pm(i, j) * pn(i, j);
            cff1 = cff * (FX(i + 1, j) - FX(i, j));
            cff2 = cff * (FE(i, j + 1) - FE(i, j));
            cff3 = dt * (FS(i, j, k2) - FS(i, j, k1));
            cff4 = cff1 + cff2 + cff3;
            t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) - cff4;
          }
        }
    }
  }
  free(FE_); free(FX_); free(FS_); free(dTdz_); free(dTdx_); free(dTde_); free(dZdx_); free(dZde_);
#undef S
#undef SD
#undef OUT
#undef FE
#undef FX
#undef FS
#undef dTdz
#undef dTdx
#undef dTde
#undef dZdx
#undef dZde
}

/* The isopycnal operator on the 3-D array S (private extents) over (i0:i1, j0:j1).  mode 0: t3dmix2_iso.h:193-437
 * (coefficient diff2, t(nnew) = t(nnew) + ...); mode 1 / 2: the two blocks of t3dmix4_iso.h (:262-500 -> LapT,
 * :610-805 t(nnew) = t(nnew) - ...; coefficient diff4 folded into the vertical flux term by term).  Horizontal
 * differences of the potential density take the place of the geopotential operator's dZdx / dZde, the vertical tracer
 * difference is scaled by -1 / MAX(pden(k) - pden(k+1), eps), and MIN / MAX change places. */
static void iso_pass(const roms_bounds_t *b, const roms_params_t *p, const roms_step_idx_t *s, roms_fields_t *F,
                     int itrc, const double *S_, const double *S2_, double *out_, int i0, int i1, int j0, int j1, int mode)
{
  ORACLE_PROLOGUE
  const int nnew = s->nnew;
  const double dt = p->dt;
  const double eps = 0.5;
  double cff, cff1, cff2, cff3, cff4;
  const long n2 = nis * njs;
  double *FE_ = walloc(n2), *FX_ = walloc(n2), *FS_ = walloc(2 * n2);
  double *dTdr_ = walloc(2 * n2), *dTdx_ = walloc(2 * n2), *dTde_ = walloc(2 * n2);
  double *dRdx_ = walloc(2 * n2), *dRde_ = walloc(2 * n2);
#define S(i,j,k) S_[WS3(i,j,k)]
/* the difference S(a) - S(b); with TS_MIX_STABILITY (S2_ = t(nstp), first operator only) the weighted one */
#define SD(ia,ja,ka,ib,jb,kb) o_tdiff(S2_ != NULL, S(ia,ja,ka), S(ib,jb,kb), S2_ ? S2_[WS3(ia,ja,ka)] : 0.0, S2_ ? S2_[WS3(ib,jb,kb)] : 0.0)
#define OUT(i,j,k) out_[WS3(i,j,k)]
#define FE(i,j) FE_[WS2(i,j)]
#define FX(i,j) FX_[WS2(i,j)]
#define FS(i,j,k) FS_[WS2(i,j) + ((k)-1) * n2]
#define dTdr(i,j,k) dTdr_[WS2(i,j) + ((k)-1) * n2]
#define dTdx(i,j,k) dTdx_[WS2(i,j) + ((k)-1) * n2]
#define dTde(i,j,k) dTde_[WS2(i,j) + ((k)-1) * n2]
#define dRdx(i,j,k) dRdx_[WS2(i,j) + ((k)-1) * n2]
#define dRde(i,j,k) dRde_[WS2(i,j) + ((k)-1) * n2]
#define DIF(i,j) (mode == 0 ? diff2(i, j, itrc) : diff4(i, j, itrc))
  int k1, k2 = 1;
  for (int k = 0; k <= N; k++) {
    k1 = k2;
    k2 = 3 - k1;
    if (k < N) {
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1 + 1; i++) {
          cff = 0.5 * (pm(i, j) + pm(i - 1, j));
          if (p->masking) cff = cff * umask(i, j);
          if (p->wet_dry) cff = cff * umask_wet(i, j);           /* WET_DRY: the next block of the same file */
          dRdx(i, j, k2) = cff * (pden(i, j, k + 1) - pden(i - 1, j, k + 1));
          dTdx(i, j, k2) = cff * SD(i, j, k + 1, i - 1, j, k + 1);
        }
      for (int j = j0; j <= j1 + 1; j++)
        for (int i = i0; i <= i1; i++) {
          cff = 0.5 * (pn(i, j) + pn(i, j - 1));
          if (p->masking) cff = cff * vmask(i, j);
          if (p->wet_dry) cff = cff * vmask_wet(i, j);           /* WET_DRY: the next block of the same file */
          dRde(i, j, k2) = cff * (pden(i, j, k + 1) - pden(i, j - 1, k + 1));
          dTde(i, j, k2) = cff * SD(i, j, k + 1, i, j - 1, k + 1);
        }
    }
    if (k == 0 || k == N) {
      for (int j = j0 - 1; j <= j1 + 1; j++)
        for (int i = i0 - 1; i <= i1 + 1; i++) { dTdr(i, j, k2) = 0.0; FS(i, j, k2) = 0.0; }
    } else {
      for (int j = j0 - 1; j <= j1 + 1; j++)
        for (int i = i0 - 1; i <= i1 + 1; i++) {
          /* TS_MIX_MIN_STRAT, t3dmix2_iso.h:313-316 / t3dmix4_iso.h:361-364, :679-682 (strat_min = 0.1) */
          if (p->ts_mix_min_strat) cff1 = MAX(pden(i, j, k) - pden(i, j, k + 1), 0.1 * (z_r(i, j, k + 1) - z_r(i, j, k)));
          else cff1 = MAX(pden(i, j, k) - pden(i, j, k + 1), eps);
          cff = -1.0 / cff1;
          dTdr(i, j, k2) = cff * SD(i, j, k + 1, i, j, k);
          FS(i, j, k2) = cff * (z_r(i, j, k + 1) - z_r(i, j, k));
        }
    }
    if (k > 0) {
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1 + 1; i++) {
          cff = 0.25 * (DIF(i, j) + DIF(i - 1, j)) * on_u(i, j);
          FX(i, j) = cff * (Hz(i, j, k) + Hz(i - 1, j, k)) *
                     (dTdx(i, j, k1) -
                      0.5 * (MAX(dRdx(i, j, k1), 0.0) * (dTdr(i - 1, j, k1) + dTdr(i, j, k2)) +
                             MIN(dRdx(i, j, k1), 0.0) * (dTdr(i - 1, j, k2) + dTdr(i, j, k1))));
        }
      for (int j = j0; j <= j1 + 1; j++)
        for (int i = i0; i <= i1; i++) {
          cff = 0.25 * (DIF(i, j) + DIF(i, j - 1)) * om_v(i, j);
          FE(i, j) = cff * (Hz(i, j, k) + Hz(i, j - 1, k)) *
                     (dTde(i, j, k1) -
                      0.5 * (MAX(dRde(i, j, k1), 0.0) * (dTdr(i, j - 1, k1) + dTdr(i, j, k2)) +
                             MIN(dRde(i, j, k1), 0.0) * (dTdr(i, j - 1, k2) + dTdr(i, j, k1))));
        }
      if (k < N) {
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            const double r = dTdr(i, j, k2);
            cff1 = MAX(dRdx(i, j, k1), 0.0);
            cff2 = MAX(dRdx(i + 1, j, k2), 0.0);
            cff3 = MIN(dRdx(i, j, k2), 0.0);
            cff4 = MIN(dRdx(i + 1, j, k1), 0.0);
            if (mode == 0) {                     /* t3dmix2_iso.h:395-417: one running sum, the coefficient last */
              cff = cff1 * (cff1 * r - dTdx(i, j, k1)) + cff2 * (cff2 * r - dTdx(i + 1, j, k2)) +
                    cff3 * (cff3 * r - dTdx(i, j, k2)) + cff4 * (cff4 * r - dTdx(i + 1, j, k1));
              cff1 = MAX(dRde(i, j, k1), 0.0);
              cff2 = MAX(dRde(i, j + 1, k2), 0.0);
              cff3 = MIN(dRde(i, j, k2), 0.0);
              cff4 = MIN(dRde(i, j + 1, k1), 0.0);
              cff = cff + cff1 * (cff1 * r - dTde(i, j, k1)) + cff2 * (cff2 * r - dTde(i, j + 1, k2)) +
                    cff3 * (cff3 * r - dTde(i, j, k2)) + cff4 * (cff4 * r - dTde(i, j + 1, k1));
              FS(i, j, k2) = 0.5 * cff * diff2(i, j, itrc) * FS(i, j, k2);
            } else {                             /* t3dmix4_iso.h:443-480: each direction times its coefficient */
              const double difx = 0.5 * diff4(i, j, itrc), dife = difx;
              cff = difx * (cff1 * (cff1 * r - dTdx(i, j, k1)) + cff2 * (cff2 * r - dTdx(i + 1, j, k2)) +
                            cff3 * (cff3 * r - dTdx(i, j, k2)) + cff4 * (cff4 * r - dTdx(i + 1, j, k1)));
              cff1 = MAX(dRde(i, j, k1), 0.0);
              cff2 = MAX(dRde(i, j + 1, k2), 0.0);
              cff3 = MIN(dRde(i, j, k2), 0.0);
              cff4 = MIN(dRde(i, j + 1, k1), 0.0);
              cff = cff + dife * (cff1 * (cff1 * r - dTde(i, j, k1)) + cff2 * (cff2 * r - dTde(i, j + 1, k2)) +
                                  cff3 * (cff3 * r - dTde(i, j, k2)) + cff4 * (cff4 * r - dTde(i, j + 1, k1)));
              FS(i, j, k2) = cff * FS(i, j, k2);
            }
          }
      }
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1; i++) {
          if (mode == 1) {
            cff = pm(i, j) * pn(i, j);
            cff1 = 1.0 / Hz(i, j, k);
            OUT(i, j, k) = cff1 * (cff * (FX(i + 1, j) - FX(i, j) + FE(i, j + 1) - FE(i, j)) + (FS(i, j, k2) - FS(i, j, k1)));
          } else {
            cff = dt * pm(i, j) * pn(i, j);
            cff1 = cff * (FX(i + 1, j) - FX(i, j));
            cff2 = cff * (FE(i, j + 1) - FE(i, j));
            cff3 = dt * (FS(i, j, k2) - FS(i, j, k1));
            cff4 = cff1 + cff2 + cff3;
            if (mode == 0) t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) + cff4;
            else t(i, j, k, nnew, itrc) = t(i, j, k, nnew, itrc) - cff4;
          }
        }
    }
  }
  free(FE_); free(FX_); free(FS_); free(dTdr_); free(dTdx_); free(dTde_); free(dRdx_); free(dRde_);
#undef S
#undef SD
#undef OUT
#undef FE
#undef FX
#undef FS
#undef dTdr
#undef dTdx
#undef dTde
#undef dRdx
#undef dRde
#undef DIF
}

/* t3dmix2_iso_tile */
int oracle_t3dmix2_iso(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, nstp = s->nstp;
  double *T_ = walloc(nis * njs * N);
  double *T2_ = p->ts_mix_stability ? walloc(nis * njs * N) : NULL;      /* TS_MIX_STABILITY: t(nstp) beside t(nrhs) */
  for (int itrc = 1; itrc <= NT; itrc++) {
    for (int k = 1; k <= N; k++)
      for (int j = MAX(JminS, LBj); j <= MIN(JmaxS, UBj); j++)
        for (int i = MAX(IminS, LBi); i <= MIN(ImaxS, UBi); i++) {
          T_[WS3(i, j, k)] = t(i, j, k, nrhs, itrc);
          if (T2_) T2_[WS3(i, j, k)] = t(i, j, k, nstp, itrc);
        }
    iso_pass(b, p, s, F, itrc, T_, T2_, NULL, Istr, Iend, Jstr, Jend, 0);
  }
  free(T_); free(T2_);
  return 0;
}

/* t3dmix4_geo_tile (iso = 0) and t3dmix4_iso_tile (iso = 1): the two operators with the edge / corner rule of the
 * first result in between (t3dmix4_geo.h:457-575 = t3dmix4_iso.h:502-608) */
static int t3dmix4_rot(OARGS, int iso)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, nstp = s->nstp;
  const long n3s = nis * njs * N;
  double *T_ = walloc(n3s), *LapT_ = walloc(n3s);
  double *T2_ = p->ts_mix_stability ? walloc(n3s) : NULL;               /* TS_MIX_STABILITY: t(nstp) beside t(nrhs) */
#define LapT(i,j,k) LapT_[WS3(i,j,k)]
  FIRST_RANGE
  for (int itrc = 1; itrc <= NT; itrc++) {
    /* t(nrhs) on the private extents (the first pass reads it on Imin-1:Imax+1, Jmin-1:Jmax+1) */
    for (int k = 1; k <= N; k++)
      for (int j = MAX(JminS, LBj); j <= MIN(JmaxS, UBj); j++)
        for (int i = MAX(IminS, LBi); i <= MIN(ImaxS, UBi); i++) {
          T_[WS3(i, j, k)] = t(i, j, k, nrhs, itrc);
          if (T2_) T2_[WS3(i, j, k)] = t(i, j, k, nstp, itrc);
        }
    if (iso) iso_pass(b, p, s, F, itrc, T_, T2_, LapT_, Imin, Imax, Jmin, Jmax, 1);
    else rotated_pass(b, p, s, F, itrc, T_, T2_, LapT_, Imin, Imax, Jmin, Jmax, 1);
    /* physical edges and corners of the first result, :457-575 */
    if (!EWperiodic) {
      if (west_edge) {
        const int closed = o_lbc(p, LBS_WEST, LBV_T) == LBC_CLOSED;
        for (int k = 1; k <= N; k++)
          for (int j = Jmin; j <= Jmax; j++) LapT(Istr - 1, j, k) = closed ? 0.0 : LapT(Istr, j, k);
      }
      if (east_edge) {
        const int closed = o_lbc(p, LBS_EAST, LBV_T) == LBC_CLOSED;
        for (int k = 1; k <= N; k++)
          for (int j = Jmin; j <= Jmax; j++) LapT(Iend + 1, j, k) = closed ? 0.0 : LapT(Iend, j, k);
      }
    }
    if (!NSperiodic) {
      if (south_edge) {
        const int closed = o_lbc(p, LBS_SOUTH, LBV_T) == LBC_CLOSED;
        for (int k = 1; k <= N; k++)
          for (int i = Imin; i <= Imax; i++) LapT(i, Jstr - 1, k) = closed ? 0.0 : LapT(i, Jstr, k);
      }
      if (north_edge) {
        const int closed = o_lbc(p, LBS_NORTH, LBV_T) == LBC_CLOSED;
        for (int k = 1; k <= N; k++)
          for (int i = Imin; i <= Imax; i++) LapT(i, Jend + 1, k) = closed ? 0.0 : LapT(i, Jend, k);
      }
    }
    if (!(NSperiodic || EWperiodic)) {
      for (int k = 1; k <= N; k++) {
        if (south_edge && west_edge) LapT(Istr - 1, Jstr - 1, k) = 0.5 * (LapT(Istr, Jstr - 1, k) + LapT(Istr - 1, Jstr, k));
        if (south_edge && east_edge) LapT(Iend + 1, Jstr - 1, k) = 0.5 * (LapT(Iend, Jstr - 1, k) + LapT(Iend + 1, Jstr, k));
        if (north_edge && west_edge) LapT(Istr - 1, Jend + 1, k) = 0.5 * (LapT(Istr, Jend + 1, k) + LapT(Istr - 1, Jend, k));
        if (north_edge && east_edge) LapT(Iend + 1, Jend + 1, k) = 0.5 * (LapT(Iend, Jend + 1, k) + LapT(Iend + 1, Jend, k));
      }
    }
    if (iso) iso_pass(b, p, s, F, itrc, LapT_, NULL, NULL, Istr, Iend, Jstr, Jend, 2);
    else rotated_pass(b, p, s, F, itrc, LapT_, NULL, NULL, Istr, Iend, Jstr, Jend, 0);
  }
  free(T_); free(T2_); free(LapT_);
#undef LapT
  return 0;
}

int oracle_t3dmix4(OARGS)
{
  if (p->mix_iso_ts) return t3dmix4_rot(b, p, s, F, 1);
  if (p->mix_geo_ts) return t3dmix4_rot(b, p, s, F, 0);
  if (p->mix_s_ts) return t3dmix4_s(b, p, s, F);
  return 8;
}

/* uv3dmix4_s_tile -- uv3dmix4_s.h:255-625 */
int oracle_uv3dmix4(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, nnew = s->nnew;
  const double dt = p->dt, gamma2 = p->gamma2;
  double cff, cff1, cff2, cff3;
  const long n2 = nis * njs;
  double *UFe_ = walloc(n2), *VFe_ = walloc(n2), *UFx_ = walloc(n2), *VFx_ = walloc(n2), *LapU_ = walloc(n2), *LapV_ = walloc(n2);
#define UFe(i,j) UFe_[WS2(i,j)]
#define VFe(i,j) VFe_[WS2(i,j)]
#define UFx(i,j) UFx_[WS2(i,j)]
#define VFx(i,j) VFx_[WS2(i,j)]
#define LapU(i,j) LapU_[WS2(i,j)]
#define LapV(i,j) LapV_[WS2(i,j)]
  int IminU, ImaxU, IminV, ImaxV, JminU, JmaxU, JminV, JmaxV;
  if (EWperiodic) { IminU = Istr - 1; ImaxU = Iend + 1; IminV = Istr - 1; ImaxV = Iend + 1; }
  else { IminU = MAX(2, IstrU - 1); ImaxU = MIN(Iend + 1, Lm); IminV = MAX(1, Istr - 1); ImaxV = MIN(Iend + 1, Lm); }
  if (NSperiodic) { JminU = Jstr - 1; JmaxU = Jend + 1; JminV = Jstr - 1; JmaxV = Jend + 1; }
  else { JminU = MAX(1, Jstr - 1); JmaxU = MIN(Jend + 1, Mm); JminV = MAX(2, JstrV - 1); JmaxV = MIN(Jend + 1, Mm); }
  for (int k = 1; k <= N; k++) {
    /* first harmonic operator (m s^-3/2), :283-355: no Hz in the flux */
    for (int j = JminV - 1; j <= JmaxV; j++)
      for (int i = IminU - 1; i <= ImaxU; i++) {
        cff = 0.5 *
              (pmon_r(i, j) * ((pn(i, j) + pn(i + 1, j)) * u(i + 1, j, k, nrhs) - (pn(i - 1, j) + pn(i, j)) * u(i, j, k, nrhs)) -
               pnom_r(i, j) * ((pm(i, j) + pm(i, j + 1)) * v(i, j + 1, k, nrhs) - (pm(i, j - 1) + pm(i, j)) * v(i, j, k, nrhs)));
        UFx(i, j) = on_r(i, j) * on_r(i, j) * visc4_r(i, j) * cff;
        VFe(i, j) = om_r(i, j) * om_r(i, j) * visc4_r(i, j) * cff;
      }
    for (int j = JminU; j <= JmaxU + 1; j++)
      for (int i = IminV; i <= ImaxV + 1; i++) {
        cff = 0.5 *
              (pmon_p(i, j) * ((pn(i, j - 1) + pn(i, j)) * v(i, j, k, nrhs) - (pn(i - 1, j - 1) + pn(i - 1, j)) * v(i - 1, j, k, nrhs)) +
               pnom_p(i, j) * ((pm(i - 1, j) + pm(i, j)) * u(i, j, k, nrhs) - (pm(i - 1, j - 1) + pm(i, j - 1)) * u(i, j - 1, k, nrhs)));
        if (p->masking) cff = cff * pmask(i, j);
        if (p->wet_dry) cff = cff * pmask_wet(i, j);           /* WET_DRY: the next block of the same file */
        UFe(i, j) = om_p(i, j) * om_p(i, j) * visc4_p(i, j) * cff;
        VFx(i, j) = on_p(i, j) * on_p(i, j) * visc4_p(i, j) * cff;
      }
    for (int j = JminU; j <= JmaxU; j++)
      for (int i = IminU; i <= ImaxU; i++)
        LapU(i, j) = 0.125 * (pm(i - 1, j) + pm(i, j)) * (pn(i - 1, j) + pn(i, j)) *
                     ((pn(i - 1, j) + pn(i, j)) * (UFx(i, j) - UFx(i - 1, j)) +
                      (pm(i - 1, j) + pm(i, j)) * (UFe(i, j + 1) - UFe(i, j)));
    for (int j = JminV; j <= JmaxV; j++)
      for (int i = IminV; i <= ImaxV; i++)
        LapV(i, j) = 0.125 * (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1)) *
                     ((pn(i, j - 1) + pn(i, j)) * (VFx(i + 1, j) - VFx(i, j)) -
                      (pm(i, j - 1) + pm(i, j)) * (VFe(i, j) - VFe(i, j - 1)));
    /* physical edges, :357-470: the normal component zero (closed) or a copy; the tangential one by the slipperiness
     * gamma2 (closed) or zero */
    if (!EWperiodic) {
      if (west_edge) {
        const int cu = o_lbc(p, LBS_WEST, LBV_U) == LBC_CLOSED, cv = o_lbc(p, LBS_WEST, LBV_V) == LBC_CLOSED;
        for (int j = JminU; j <= JmaxU; j++) LapU(Istr, j) = cu ? 0.0 : LapU(Istr + 1, j);
        for (int j = JminV; j <= JmaxV; j++) LapV(Istr - 1, j) = cv ? gamma2 * LapV(Istr, j) : 0.0;
      }
      if (east_edge) {
        const int cu = o_lbc(p, LBS_EAST, LBV_U) == LBC_CLOSED, cv = o_lbc(p, LBS_EAST, LBV_V) == LBC_CLOSED;
        for (int j = JminU; j <= JmaxU; j++) LapU(Iend + 1, j) = cu ? 0.0 : LapU(Iend, j);
        for (int j = JminV; j <= JmaxV; j++) LapV(Iend + 1, j) = cv ? gamma2 * LapV(Iend, j) : 0.0;
      }
    }
    if (!NSperiodic) {
      if (south_edge) {
        const int cu = o_lbc(p, LBS_SOUTH, LBV_U) == LBC_CLOSED, cv = o_lbc(p, LBS_SOUTH, LBV_V) == LBC_CLOSED;
        for (int i = IminU; i <= ImaxU; i++) LapU(i, Jstr - 1) = cu ? gamma2 * LapU(i, Jstr) : 0.0;
        for (int i = IminV; i <= ImaxV; i++) LapV(i, Jstr) = cv ? 0.0 : LapV(i, Jstr + 1);
      }
      if (north_edge) {
        const int cu = o_lbc(p, LBS_NORTH, LBV_U) == LBC_CLOSED, cv = o_lbc(p, LBS_NORTH, LBV_V) == LBC_CLOSED;
        for (int i = IminU; i <= ImaxU; i++) LapU(i, Jend + 1) = cu ? gamma2 * LapU(i, Jend) : 0.0;
        for (int i = IminV; i <= ImaxV; i++) LapV(i, Jend + 1) = cv ? 0.0 : LapV(i, Jend);
      }
    }
    if (!(NSperiodic || EWperiodic)) {                 /* corners, :472-520 */
      if (south_edge && west_edge) {
        LapU(Istr, Jstr - 1) = 0.5 * (LapU(Istr + 1, Jstr - 1) + LapU(Istr, Jstr));
        LapV(Istr - 1, Jstr) = 0.5 * (LapV(Istr - 1, Jstr + 1) + LapV(Istr, Jstr));
      }
      if (south_edge && east_edge) {
        LapU(Iend + 1, Jstr - 1) = 0.5 * (LapU(Iend, Jstr - 1) + LapU(Iend + 1, Jstr));
        LapV(Iend + 1, Jstr) = 0.5 * (LapV(Iend, Jstr) + LapV(Iend + 1, Jstr + 1));
      }
      if (north_edge && west_edge) {
        LapU(Istr, Jend + 1) = 0.5 * (LapU(Istr + 1, Jend + 1) + LapU(Istr, Jend));
        LapV(Istr - 1, Jend + 1) = 0.5 * (LapV(Istr, Jend + 1) + LapV(Istr - 1, Jend));
      }
      if (north_edge && east_edge) {
        LapU(Iend + 1, Jend + 1) = 0.5 * (LapU(Iend, Jend + 1) + LapU(Iend + 1, Jend));
        LapV(Iend + 1, Jend + 1) = 0.5 * (LapV(Iend, Jend + 1) + LapV(Iend + 1, Jend));
      }
    }
    /* second operator (with Hz) and time step, :522-620 */
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = Hz(i, j, k) * 0.5 *
              (pmon_r(i, j) * ((pn(i, j) + pn(i + 1, j)) * LapU(i + 1, j) - (pn(i - 1, j) + pn(i, j)) * LapU(i, j)) -
               pnom_r(i, j) * ((pm(i, j) + pm(i, j + 1)) * LapV(i, j + 1) - (pm(i, j - 1) + pm(i, j)) * LapV(i, j)));
        UFx(i, j) = on_r(i, j) * on_r(i, j) * visc4_r(i, j) * cff;
        VFe(i, j) = om_r(i, j) * om_r(i, j) * visc4_r(i, j) * cff;
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff = 0.125 * (Hz(i - 1, j, k) + Hz(i, j, k) + Hz(i - 1, j - 1, k) + Hz(i, j - 1, k)) *
              (pmon_p(i, j) * ((pn(i, j - 1) + pn(i, j)) * LapV(i, j) - (pn(i - 1, j - 1) + pn(i - 1, j)) * LapV(i - 1, j)) +
               pnom_p(i, j) * ((pm(i - 1, j) + pm(i, j)) * LapU(i, j) - (pm(i - 1, j - 1) + pm(i, j - 1)) * LapU(i, j - 1)));
        if (p->masking) cff = cff * pmask(i, j);
        if (p->wet_dry) cff = cff * pmask_wet(i, j);           /* WET_DRY: the next block of the same file */
        UFe(i, j) = om_p(i, j) * om_p(i, j) * visc4_p(i, j) * cff;
        VFx(i, j) = on_p(i, j) * on_p(i, j) * visc4_p(i, j) * cff;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = dt * 0.25 * (pm(i - 1, j) + pm(i, j)) * (pn(i - 1, j) + pn(i, j));
        cff1 = 0.5 * (pn(i - 1, j) + pn(i, j)) * (UFx(i, j) - UFx(i - 1, j));
        cff2 = 0.5 * (pm(i - 1, j) + pm(i, j)) * (UFe(i, j + 1) - UFe(i, j));
        cff3 = cff * (cff1 + cff2);
        rufrc(i, j) = rufrc(i, j) - cff1 - cff2;
        u(i, j, k, nnew) = u(i, j, k, nnew) - cff3;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = dt * 0.25 * (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1));
        cff1 = 0.5 * (pn(i, j - 1) + pn(i, j)) * (VFx(i + 1, j) - VFx(i, j));
        cff2 = 0.5 * (pm(i, j - 1) + pm(i, j)) * (VFe(i, j) - VFe(i, j - 1));
        cff3 = cff * (cff1 - cff2);
        rvfrc(i, j) = rvfrc(i, j) - cff1 + cff2;
        v(i, j, k, nnew) = v(i, j, k, nnew) - cff3;
      }
  }
  free(UFe_); free(VFe_); free(UFx_); free(VFx_); free(LapU_); free(LapV_);
  return 0;
}
