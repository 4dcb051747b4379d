/*
 * oracle_rhs3d.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * rhs3d_tile: Coriolis, curvilinear terms, 3rd-order upstream horizontal and
 * 4th-order centred vertical advection of momentum, and the vertical integral
 * rufrc/rvfrc (ROMS/Nonlinear/rhs3d.F:174-1673), plus the rhs3d driver
 * (rhs3d.F:25-170).  Parity unpinned (mod_sources chain via pre_step3d_mod).
 */
#include "oracle.h"

int oracle_rhs3d_tile(OARGS)
{
  ORACLE_PROLOGUE
  if (o_check_lbc(b, p)) return 8;
  const int nrhs = s->nrhs;
  const double Gadv = -0.25;
  double cff, cff1, cff2, cff3, cff4;
  double *FC_ = walloc(nis * (N + 1));
  double *Huee_ = walloc(nis * njs), *Huxx_ = walloc(nis * njs), *Hvee_ = walloc(nis * njs), *Hvxx_ = walloc(nis * njs);
  double *UFx_ = walloc(nis * njs), *UFe_ = walloc(nis * njs), *VFx_ = walloc(nis * njs), *VFe_ = walloc(nis * njs);
  double *uee_ = walloc(nis * njs), *uxx_ = walloc(nis * njs), *vee_ = walloc(nis * njs), *vxx_ = walloc(nis * njs);
#define FC(i,k) FC_[WSK(i,k)]
#define Huee(i,j) Huee_[WS2(i,j)]
#define Huxx(i,j) Huxx_[WS2(i,j)]
#define Hvee(i,j) Hvee_[WS2(i,j)]
#define Hvxx(i,j) Hvxx_[WS2(i,j)]
#define UFx(i,j) UFx_[WS2(i,j)]
#define UFe(i,j) UFe_[WS2(i,j)]
#define VFx(i,j) VFx_[WS2(i,j)]
#define VFe(i,j) VFe_[WS2(i,j)]
#define uee(i,j) uee_[WS2(i,j)]
#define uxx(i,j) uxx_[WS2(i,j)]
#define vee(i,j) vee_[WS2(i,j)]
#define vxx(i,j) vxx_[WS2(i,j)]

  for (int k = 1; k <= N; k++) {
    if (p->uv_cor) {
      /* Coriolis, rhs3d.F:467-505 */
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          cff = 0.5 * Hz(i, j, k) * fomn(i, j);
          UFx(i, j) = cff * (v(i, j, k, nrhs) + v(i, j + 1, k, nrhs));
          VFe(i, j) = cff * (u(i, j, k, nrhs) + u(i + 1, j, k, nrhs));
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = 0.5 * (UFx(i, j) + UFx(i - 1, j));
          ru(i, j, k, nrhs) = ru(i, j, k, nrhs) + cff1;
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = 0.5 * (VFe(i, j) + VFe(i, j - 1));
          rv(i, j, k, nrhs) = rv(i, j, k, nrhs) - cff1;
        }
    }
    if (p->curvgrid && p->uv_adv) {
      /* curvilinear terms, rhs3d.F:509-560 */
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          cff1 = 0.5 * (v(i, j, k, nrhs) + v(i, j + 1, k, nrhs));
          cff2 = 0.5 * (u(i, j, k, nrhs) + u(i + 1, j, k, nrhs));
          cff3 = cff1 * dndx(i, j);
          cff4 = cff2 * dmde(i, j);
          cff = Hz(i, j, k) * (cff3 - cff4);
          UFx(i, j) = cff * cff1;
          VFe(i, j) = cff * cff2;
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = 0.5 * (UFx(i, j) + UFx(i - 1, j));
          ru(i, j, k, nrhs) = ru(i, j, k, nrhs) + cff1;
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = 0.5 * (VFe(i, j) + VFe(i, j - 1));
          rv(i, j, k, nrhs) = rv(i, j, k, nrhs) - cff1;
        }
    }
    if (p->uv_adv) {
      /* horizontal advection, UV_U3HADVECTION default, rhs3d.F:596-982 */
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrUm1; i <= Iendp1; i++) {
          uxx(i, j) = u(i - 1, j, k, nrhs) - 2.0 * u(i, j, k, nrhs) + u(i + 1, j, k, nrhs);
          Huxx(i, j) = Huon(i - 1, j, k) - 2.0 * Huon(i, j, k) + Huon(i + 1, j, k);
        }
      if (!EWperiodic) {
        if (west_edge) for (int j = Jstr; j <= Jend; j++) { uxx(Istr, j) = uxx(Istr + 1, j); Huxx(Istr, j) = Huxx(Istr + 1, j); }
        if (east_edge) for (int j = Jstr; j <= Jend; j++) { uxx(Iend + 1, j) = uxx(Iend, j); Huxx(Iend + 1, j) = Huxx(Iend, j); }
      }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          cff1 = u(i, j, k, nrhs) + u(i + 1, j, k, nrhs);
          if (cff1 > 0.0) cff = uxx(i, j);
          else cff = uxx(i + 1, j);
          UFx(i, j) = 0.25 * (cff1 + Gadv * cff) *
                      (Huon(i, j, k) + Huon(i + 1, j, k) + Gadv * 0.5 * (Huxx(i, j) + Huxx(i + 1, j)));
        }
      for (int j = Jstrm1; j <= Jendp1; j++)
        for (int i = IstrU; i <= Iend; i++)
          uee(i, j) = u(i, j - 1, k, nrhs) - 2.0 * u(i, j, k, nrhs) + u(i, j + 1, k, nrhs);
      if (!NSperiodic) {
        if (south_edge) for (int i = IstrU; i <= Iend; i++) uee(i, Jstr - 1) = uee(i, Jstr);
        if (north_edge) for (int i = IstrU; i <= Iend; i++) uee(i, Jend + 1) = uee(i, Jend);
      }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = IstrU - 1; i <= Iend; i++)
          Hvxx(i, j) = Hvom(i - 1, j, k) - 2.0 * Hvom(i, j, k) + Hvom(i + 1, j, k);
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = u(i, j, k, nrhs) + u(i, j - 1, k, nrhs);
          cff2 = Hvom(i, j, k) + Hvom(i - 1, j, k);
          if (cff2 > 0.0) cff = uee(i, j - 1);
          else cff = uee(i, j);
          UFe(i, j) = 0.25 * (cff1 + Gadv * cff) * (cff2 + Gadv * 0.5 * (Hvxx(i, j) + Hvxx(i - 1, j)));
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istrm1; i <= Iendp1; i++)
          vxx(i, j) = v(i - 1, j, k, nrhs) - 2.0 * v(i, j, k, nrhs) + v(i + 1, j, k, nrhs);
      if (!EWperiodic) {
        if (west_edge) for (int j = JstrV; j <= Jend; j++) vxx(Istr - 1, j) = vxx(Istr, j);
        if (east_edge) for (int j = JstrV; j <= Jend; j++) vxx(Iend + 1, j) = vxx(Iend, j);
      }
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++)
          Huee(i, j) = Huon(i, j - 1, k) - 2.0 * Huon(i, j, k) + Huon(i, j + 1, k);
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          cff1 = v(i, j, k, nrhs) + v(i - 1, j, k, nrhs);
          cff2 = Huon(i, j, k) + Huon(i, j - 1, k);
          if (cff2 > 0.0) cff = vxx(i - 1, j);
          else cff = vxx(i, j);
          VFx(i, j) = 0.25 * (cff1 + Gadv * cff) * (cff2 + Gadv * 0.5 * (Huee(i, j) + Huee(i, j - 1)));
        }
      for (int j = JstrVm1; j <= Jendp1; j++)
        for (int i = Istr; i <= Iend; i++) {
          vee(i, j) = v(i, j - 1, k, nrhs) - 2.0 * v(i, j, k, nrhs) + v(i, j + 1, k, nrhs);
          Hvee(i, j) = Hvom(i, j - 1, k) - 2.0 * Hvom(i, j, k) + Hvom(i, j + 1, k);
        }
      if (!NSperiodic) {
        if (south_edge) for (int i = Istr; i <= Iend; i++) { vee(i, Jstr) = vee(i, Jstr + 1); Hvee(i, Jstr) = Hvee(i, Jstr + 1); }
        if (north_edge) for (int i = Istr; i <= Iend; i++) { vee(i, Jend + 1) = vee(i, Jend); Hvee(i, Jend + 1) = Hvee(i, Jend); }
      }
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = v(i, j, k, nrhs) + v(i, j + 1, k, nrhs);
          if (cff1 > 0.0) cff = vee(i, j);
          else cff = vee(i, j + 1);
          VFe(i, j) = 0.25 * (cff1 + Gadv * cff) *
                      (Hvom(i, j, k) + Hvom(i, j + 1, k) + Gadv * 0.5 * (Hvee(i, j) + Hvee(i, j + 1)));
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = UFx(i, j) - UFx(i - 1, j);
          cff2 = UFe(i, j + 1) - UFe(i, j);
          cff = cff1 + cff2;
          ru(i, j, k, nrhs) = ru(i, j, k, nrhs) - cff;
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = VFx(i + 1, j) - VFx(i, j);
          cff2 = VFe(i, j) - VFe(i, j - 1);
          cff = cff1 + cff2;
          rv(i, j, k, nrhs) = rv(i, j, k, nrhs) - cff;
        }
    }
  }

  /* J_LOOP: vertical advection (C4) and vertical integral, rhs3d.F:1009-1660 */
  for (int j = Jstr; j <= Jend; j++) {
    if (p->uv_adv) {
      cff1 = 9.0 / 16.0;
      cff2 = 1.0 / 16.0;
      for (int k = 2; k <= N - 2; k++)
        for (int i = IstrU; i <= Iend; i++)
          FC(i, k) = (cff1 * (u(i, j, k, nrhs) + u(i, j, k + 1, nrhs)) - cff2 * (u(i, j, k - 1, nrhs) + u(i, j, k + 2, nrhs))) *
                     (cff1 * (W(i, j, k) + W(i - 1, j, k)) - cff2 * (W(i + 1, j, k) + W(i - 2, j, k)));
      for (int i = IstrU; i <= Iend; i++) {
        FC(i, N) = 0.0;
        FC(i, N - 1) = (cff1 * (u(i, j, N - 1, nrhs) + u(i, j, N, nrhs)) - cff2 * (u(i, j, N - 2, nrhs) + u(i, j, N, nrhs))) *
                       (cff1 * (W(i, j, N - 1) + W(i - 1, j, N - 1)) - cff2 * (W(i + 1, j, N - 1) + W(i - 2, j, N - 1)));
        FC(i, 1) = (cff1 * (u(i, j, 1, nrhs) + u(i, j, 2, nrhs)) - cff2 * (u(i, j, 1, nrhs) + u(i, j, 3, nrhs))) *
                   (cff1 * (W(i, j, 1) + W(i - 1, j, 1)) - cff2 * (W(i + 1, j, 1) + W(i - 2, j, 1)));
        FC(i, 0) = 0.0;
      }
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff = FC(i, k) - FC(i, k - 1);
          ru(i, j, k, nrhs) = ru(i, j, k, nrhs) - cff;
        }
      if (j >= JstrV) {
        for (int k = 2; k <= N - 2; k++)
          for (int i = Istr; i <= Iend; i++)
            FC(i, k) = (cff1 * (v(i, j, k, nrhs) + v(i, j, k + 1, nrhs)) - cff2 * (v(i, j, k - 1, nrhs) + v(i, j, k + 2, nrhs))) *
                       (cff1 * (W(i, j, k) + W(i, j - 1, k)) - cff2 * (W(i, j + 1, k) + W(i, j - 2, k)));
        for (int i = Istr; i <= Iend; i++) {
          FC(i, N) = 0.0;
          FC(i, N - 1) = (cff1 * (v(i, j, N - 1, nrhs) + v(i, j, N, nrhs)) - cff2 * (v(i, j, N - 2, nrhs) + v(i, j, N, nrhs))) *
                         (cff1 * (W(i, j, N - 1) + W(i, j - 1, N - 1)) - cff2 * (W(i, j + 1, N - 1) + W(i, j - 2, N - 1)));
          FC(i, 1) = (cff1 * (v(i, j, 1, nrhs) + v(i, j, 2, nrhs)) - cff2 * (v(i, j, 1, nrhs) + v(i, j, 3, nrhs))) *
                     (cff1 * (W(i, j, 1) + W(i, j - 1, 1)) - cff2 * (W(i, j + 1, 1) + W(i, j - 2, 1)));
          FC(i, 0) = 0.0;
        }
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff = FC(i, k) - FC(i, k - 1);
            rv(i, j, k, nrhs) = rv(i, j, k, nrhs) - cff;
          }
      }
    }
    /* vertical integral -> forcing of the barotropic mode, rhs3d.F:1560-1660 */
    for (int i = IstrU; i <= Iend; i++) rufrc(i, j) = ru(i, j, 1, nrhs);
    for (int k = 2; k <= N; k++)
      for (int i = IstrU; i <= Iend; i++) rufrc(i, j) = rufrc(i, j) + ru(i, j, k, nrhs);
    for (int i = IstrU; i <= Iend; i++) {
      cff = om_u(i, j) * on_u(i, j);
      cff1 = sustr(i, j) * cff;
      cff2 = -bustr(i, j) * cff;
      rufrc(i, j) = rufrc(i, j) + cff1 + cff2;
    }
    if (j >= JstrV) {
      for (int i = Istr; i <= Iend; i++) rvfrc(i, j) = rv(i, j, 1, nrhs);
      for (int k = 2; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) rvfrc(i, j) = rvfrc(i, j) + rv(i, j, k, nrhs);
      for (int i = Istr; i <= Iend; i++) {
        cff = om_v(i, j) * on_v(i, j);
        cff1 = svstr(i, j) * cff;
        cff2 = -bvstr(i, j) * cff;
        rvfrc(i, j) = rvfrc(i, j) + cff1 + cff2;
      }
    }
  }
  free(FC_); free(Huee_); free(Huxx_); free(Hvee_); free(Hvxx_); free(UFx_); free(UFe_); free(VFx_); free(VFe_);
  free(uee_); free(uxx_); free(vee_); free(vxx_);
  return 0;
}

/* rhs3d(ng,tile) driver -- rhs3d.F:25-170: pre_step3d, prsgrd, t3dmix2, t3dmix4,
 * rhs3d_tile, uv3dmix2, uv3dmix4 in this order. */
int oracle_rhs3d(OARGS)
{
  int rc;
  if ((rc = oracle_pre_step3d(b, p, s, F))) return rc;
  if ((rc = oracle_prsgrd(b, p, s, F))) return rc;
  if (p->ts_dif2 && (rc = oracle_t3dmix2(b, p, s, F))) return rc;
  if (p->ts_dif4 && (rc = oracle_t3dmix4(b, p, s, F))) return rc;
  if ((rc = oracle_rhs3d_tile(b, p, s, F))) return rc;
  if (p->uv_vis2 && (rc = oracle_uv3dmix2(b, p, s, F))) return rc;
  if (p->uv_vis4 && (rc = oracle_uv3dmix4(b, p, s, F))) return rc;
  return 0;
}
