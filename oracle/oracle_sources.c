/*
 * oracle_sources.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Point sources / sinks through u- and v-faces (LuvSrc, rivers): the blocks of
 *   step2d_tile     ROMS/Nonlinear/step2d_LF_AM3.h:2484-2502  (ubar, vbar at the source faces)
 *   step3d_uv_tile  ROMS/Nonlinear/step3d_uv.F:971-995        (u, v at the source faces)
 *   pre_step3d_tile ROMS/Nonlinear/pre_step3d.F:530-553       (the predictor's horizontal tracer flux)
 *   step3d_t_tile   ROMS/Nonlinear/step3d_t.F:734-799         (the corrector's)
 *   wetdry_tile     ROMS/Nonlinear/wetdry.F:307-320, :511-524 (the output masks at source faces)
 * PARITY UNPINNED: every one of these routines USEs mod_sources, which needs netCDF; the checks are known answers
 * (tests/test_sources.py: the volume and tracer budgets of a river, constancy of a uniform tracer).
 * The table (SOURCES(ng) of mod_sources.F:56-80) is a process-wide static here: the oracle serves one tile per process.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

o_src_t o_src;

int oracle_set_sources(int Nsrc, const int *Isrc, const int *Jsrc, const double *Dsrc, const double *Qbar,
                       const double *Qsrc, const double *Tsrc, const int *LtracerSrc, int N, int NT)
{
  free(o_src.I); free(o_src.J); free(o_src.D); free(o_src.Qbar); free(o_src.Qsrc); free(o_src.Tsrc);
  memset(&o_src, 0, sizeof o_src);
  o_src.given = Isrc != NULL || Nsrc > 0;     /* (0, NULL ...) forgets the table; (0, arrays) is an application without sources */
  if (Nsrc <= 0) return 0;
  for (int is = 0; is < Nsrc; is++)
    if ((int)Dsrc[is] < 0 || (int)Dsrc[is] > 2) return 8;
  o_src.n = Nsrc; o_src.N = N; o_src.NT = NT;
#define DUP(dst, src, cnt, T) do { dst = (T *)malloc(sizeof(T) * (size_t)(cnt)); memcpy(dst, src, sizeof(T) * (size_t)(cnt)); } while (0)
  DUP(o_src.I, Isrc, Nsrc, int); DUP(o_src.J, Jsrc, Nsrc, int); DUP(o_src.D, Dsrc, Nsrc, double);
  DUP(o_src.Qbar, Qbar, Nsrc, double); DUP(o_src.Qsrc, Qsrc, (long)Nsrc * N, double);
  DUP(o_src.Tsrc, Tsrc, (long)Nsrc * N * NT, double);
#undef DUP
  for (int it = 0; it < NT && it < ROMS_MAXNT; it++) o_src.ltr[it] = LtracerSrc[it];
  return 0;
}

#define QSRC(is,k)      o_src.Qsrc[(is) + (long)o_src.n * ((k) - 1)]
#define TSRC(is,k,itrc) o_src.Tsrc[(is) + (long)o_src.n * (((k) - 1) + (long)o_src.N * ((itrc) - 1))]

/* the library's rule (include/roms_hip.h): LuvSrc / LwSrc without a table is an error */
int o_src_check(const roms_params_t *p)
{
  if ((p->point_sources & 3) && !o_src.given) return 8;
  return 0;
}

/* ---- LwSrc: volume influx at cell centres (Dsrc = 2) ---- */

/* step2d_LF_AM3.h:890-908: the free surface of the source cells, before zetabc */
void o_src_zeta(OARGS, int knew)
{
  ORACLE_PROLOGUE
  if (!(p->point_sources & 2)) return;
  for (int is = 0; is < o_src.n; is++) {
    if ((int)o_src.D[is] != 2) continue;
    const int i = o_src.I[is], j = o_src.J[is];
    if (!(IstrR <= i && i <= IendR && JstrR <= j && j <= JendR)) continue;
    zeta(i, j, knew) = zeta(i, j, knew) + o_src.Qbar[is] * pm(i, j) * pn(i, j) * p->dtfast;
  }
}

/* omega.F:165-190: W of row j recomputed at the source columns with Qsrc added */
void o_src_omega(OARGS, int j)
{
  ORACLE_PROLOGUE
  if (!(p->point_sources & 2)) return;
  for (int is = 0; is < o_src.n; is++) {
    if ((int)o_src.D[is] != 2) continue;
    const int ii = o_src.I[is], jj = o_src.J[is];
    if (!(IstrR <= ii && ii <= IendR && JstrR <= jj && jj <= JendR && j == jj)) continue;
    for (int k = 1; k <= N; k++)
      W(ii, jj, k) = W(ii, jj, k - 1) -
                     (Huon(ii + 1, jj, k) - Huon(ii, jj, k) + Hvom(ii, jj + 1, k) - Hvom(ii, jj, k)) + QSRC(is, k);
  }
}

/* step3d_t.F:1136-1158 (mpdata = 1: Ta of row j, before its vertical advection) and :1331-1360 (mpdata = 0: t(nnew)
 * after the vertical advection, oHz the reciprocal thickness under SPLINES_VDIFF).  Without LtracerSrc the inflow
 * carries the cell's own value t(:,:,:,3,itrc). */
void o_src_wtracer(OARGS, int itrc, int mpdata, int j, double *Ta_, const double *oHz_)
{
  ORACLE_PROLOGUE
  if (!(p->point_sources & 2)) return;
  const int nnew = s->nnew;
  const long n3s = nis * njs * N;
  for (int is = 0; is < o_src.n; is++) {
    if ((int)o_src.D[is] != 2) continue;
    const int Isrc = o_src.I[is], Jsrc = o_src.J[is];
    if (!(Istr <= Isrc && Isrc <= Iend + 1 && Jstr <= Jsrc && Jsrc <= Jend + 1)) continue;
    if (mpdata && j != Jsrc) continue;
    for (int k = 1; k <= N; k++) {
      double cff = p->dt * pm(Isrc, Jsrc) * pn(Isrc, Jsrc);
      if (!mpdata && p->splines_vdiff) cff = cff * oHz_[WS3(Isrc, Jsrc, k)];  /* SPLINES_VDIFF, :1341-1343 */
      const double cff3 = o_src.ltr[itrc - 1] ? TSRC(is, k, itrc) : t(Isrc, Jsrc, k, 3, itrc);
      if (mpdata) Ta_[WS3(Isrc, Jsrc, k) + (long)(itrc - 1) * n3s] = Ta_[WS3(Isrc, Jsrc, k) + (long)(itrc - 1) * n3s] + cff * QSRC(is, k) * cff3;
      else t(Isrc, Jsrc, k, nnew, itrc) = t(Isrc, Jsrc, k, nnew, itrc) + cff * QSRC(is, k) * cff3;
    }
  }
}

/* step2d_LF_AM3.h:2484-2502 */
void o_src_ubar(OARGS, int knew)
{
  ORACLE_PROLOGUE
  if (!(p->point_sources & 1)) return;
  for (int is = 0; is < o_src.n; is++) {
    const int i = o_src.I[is], j = o_src.J[is];
    if (!(IstrR <= i && i <= IendR && JstrR <= j && j <= JendR)) continue;
    if ((int)o_src.D[is] == 0) {
      const double cff = 1.0 / (on_u(i, j) * 0.5 * (zeta(i - 1, j, knew) + h(i - 1, j) + zeta(i, j, knew) + h(i, j)));
      ubar(i, j, knew) = o_src.Qbar[is] * cff;
    } else if ((int)o_src.D[is] == 1) {
      const double cff = 1.0 / (om_v(i, j) * 0.5 * (zeta(i, j - 1, knew) + h(i, j - 1) + zeta(i, j, knew) + h(i, j)));
      vbar(i, j, knew) = o_src.Qbar[is] * cff;
    }
  }
}

/* step3d_uv.F:971-995 */
void o_src_uv(OARGS, int nnew)
{
  ORACLE_PROLOGUE
  if (!(p->point_sources & 1)) return;
  for (int is = 0; is < o_src.n; is++) {
    const int i = o_src.I[is], j = o_src.J[is];
    if (!(IstrR <= i && i <= IendR && JstrR <= j && j <= JendR)) continue;
    if ((int)o_src.D[is] == 0) {
      for (int k = 1; k <= N; k++) {
        const double cff1 = 1.0 / (on_u(i, j) * 0.5 * (z_w(i - 1, j, k) - z_w(i - 1, j, k - 1) + z_w(i, j, k) - z_w(i, j, k - 1)));
        u(i, j, k, nnew) = QSRC(is, k) * cff1;
      }
    } else if ((int)o_src.D[is] == 1) {
      for (int k = 1; k <= N; k++) {
        const double cff1 = 1.0 / (om_v(i, j) * 0.5 * (z_w(i, j - 1, k) - z_w(i, j - 1, k - 1) + z_w(i, j, k) - z_w(i, j, k - 1)));
        v(i, j, k, nnew) = QSRC(is, k) * cff1;
      }
    }
  }
}

/* The horizontal advective tracer flux of level k at the source faces.  pre = 1: pre_step3d.F:530-553 (Tsrc or, without
 * LtracerSrc, zero).  pre = 0: step3d_t.F:734-799 (Tsrc; without LtracerSrc and under MASKING the upstream value of the
 * wet side); wide = the ranges of MPDATA and HSIMT. */
void o_src_tflux(OARGS, int itrc, int k, double *FX_, double *FE_, int wide, int pre)
{
  ORACLE_PROLOGUE
  if (!(p->point_sources & 1)) return;
#define FX(i,j) FX_[WS2(i,j)]
#define FE(i,j) FE_[WS2(i,j)]
  const int ltr = o_src.ltr[itrc - 1];
  for (int is = 0; is < o_src.n; is++) {
    const int Isrc = o_src.I[is], Jsrc = o_src.J[is];
    const int d = (int)o_src.D[is];
    int apply;
    if (pre) apply = Istr <= Isrc && Isrc <= Iend + 1 && Jstr <= Jsrc && Jsrc <= Jend + 1;
    else if (d == 0)
      apply = wide ? (IstrUm2 <= Isrc && Isrc <= Iendp3 && JstrVm2 <= Jsrc && Jsrc <= Jendp2i)
                   : (Istr <= Isrc && Isrc <= Iend + 1 && Jstr <= Jsrc && Jsrc <= Jend);
    else
      apply = wide ? (IstrUm2 <= Isrc && Isrc <= Iendp2i && JstrVm2 <= Jsrc && Jsrc <= Jendp3)
                   : (Istr <= Isrc && Isrc <= Iend && Jstr <= Jsrc && Jsrc <= Jend + 1);
    if (!apply) continue;
    /* the private arrays span IminS:ImaxS x JminS:JmaxS */
    if (Isrc < IminS || Isrc > ImaxS || Jsrc < JminS || Jsrc > JmaxS) continue;
    if (d == 0) {
      if (ltr) FX(Isrc, Jsrc) = Huon(Isrc, Jsrc, k) * TSRC(is, k, itrc);
      else if (pre) FX(Isrc, Jsrc) = 0.0;
      else if (p->masking) {
        if (rmask(Isrc, Jsrc) == 0.0 && rmask(Isrc - 1, Jsrc) == 1.0) FX(Isrc, Jsrc) = Huon(Isrc, Jsrc, k) * t(Isrc - 1, Jsrc, k, 3, itrc);
        else if (rmask(Isrc, Jsrc) == 1.0 && rmask(Isrc - 1, Jsrc) == 0.0) FX(Isrc, Jsrc) = Huon(Isrc, Jsrc, k) * t(Isrc, Jsrc, k, 3, itrc);
      }
    } else if (d == 1) {
      if (ltr) FE(Isrc, Jsrc) = Hvom(Isrc, Jsrc, k) * TSRC(is, k, itrc);
      else if (pre) FE(Isrc, Jsrc) = 0.0;
      else if (p->masking) {
        if (rmask(Isrc, Jsrc) == 0.0 && rmask(Isrc, Jsrc - 1) == 1.0) FE(Isrc, Jsrc) = Hvom(Isrc, Jsrc, k) * t(Isrc, Jsrc - 1, k, 3, itrc);
        else if (rmask(Isrc, Jsrc) == 1.0 && rmask(Isrc, Jsrc - 1) == 0.0) FE(Isrc, Jsrc) = Hvom(Isrc, Jsrc, k) * t(Isrc, Jsrc, k, 3, itrc);
      }
    }
  }
#undef FX
#undef FE
}

/* wetdry.F:307-320, :511-524: the output masks count source faces as water */
void o_src_masks(OARGS)
{
  ORACLE_PROLOGUE
  if (!(p->point_sources & 1)) return;
  for (int is = 0; is < o_src.n; is++) {
    const int i = o_src.I[is], j = o_src.J[is];
    if (!(IstrR <= i && i <= IendR && JstrR <= j && j <= JendR)) continue;
    /* (as written, wetdry.F:313-317 takes every Dsrc other than 0 for a v-face: with LuvSrc AND LwSrc a cell-centred
     * source would open the v-face of its cell in the output mask -- not followed, here or in the library) */
    if ((int)o_src.D[is] == 0) umask_full(i, j) = 1.0;
    else if ((int)o_src.D[is] == 1) vmask_full(i, j) = 1.0;
  }
}
