/*
 * oracle_uvmix_geo.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of
 *   uv3dmix2_geo_tile   ROMS/Nonlinear/uv3dmix2_geo.h:116-756 (UV_VIS2 with MIX_GEO_UV: harmonic viscosity rotated to
 *                       geopotential surfaces; roms_params_t.uv_vis2 = 2)
 * loop for loop, with the reference's two-level slabs (k1, k2) and its order of operations.  Pinned bit for bit against
 * the reference built with MIX_GEO_UV instead of MIX_S_UV (oracle/_ref/UPWELLING_GEOUV, SEAMOUNT_GEOUV,
 * UPWELLING_MASK_GEOUV; tests/golden/ref_geouv.npz).  Never linked into the product.
 */
#include "oracle.h"

int oracle_uv3dmix2_geo(OARGS)
{
  ORACLE_PROLOGUE
  const int nrhs = s->nrhs, nnew = s->nnew;
  const double dt = p->dt;
  const long n2 = nis * njs;
  double cff, fac1, fac2, pm_p, pn_p, cff1, cff2, cff3, cff4, cff5, cff6, cff7, cff8, dmUdz, dnUdz, dmVdz, dnVdz;
  double *UFe_ = walloc(n2), *VFe_ = walloc(n2), *UFx_ = walloc(n2), *VFx_ = walloc(n2);
  double *UFse_ = walloc(2 * n2), *UFsx_ = walloc(2 * n2), *VFse_ = walloc(2 * n2), *VFsx_ = walloc(2 * n2);
  double *dmUde_ = walloc(2 * n2), *dmVde_ = walloc(2 * n2), *dnUdx_ = walloc(2 * n2), *dnVdx_ = walloc(2 * n2);
  double *dUdz_ = walloc(2 * n2), *dVdz_ = walloc(2 * n2);
  double *dZde_p_ = walloc(2 * n2), *dZde_r_ = walloc(2 * n2), *dZdx_p_ = walloc(2 * n2), *dZdx_r_ = walloc(2 * n2);
#define UFe(i,j) UFe_[WS2(i,j)]
#define VFe(i,j) VFe_[WS2(i,j)]
#define UFx(i,j) UFx_[WS2(i,j)]
#define VFx(i,j) VFx_[WS2(i,j)]
#define SL(A,i,j,k) A[WS2(i,j) + (long)((k) - 1) * n2]
#define UFse(i,j,k) SL(UFse_,i,j,k)
#define UFsx(i,j,k) SL(UFsx_,i,j,k)
#define VFse(i,j,k) SL(VFse_,i,j,k)
#define VFsx(i,j,k) SL(VFsx_,i,j,k)
#define dmUde(i,j,k) SL(dmUde_,i,j,k)
#define dmVde(i,j,k) SL(dmVde_,i,j,k)
#define dnUdx(i,j,k) SL(dnUdx_,i,j,k)
#define dnVdx(i,j,k) SL(dnVdx_,i,j,k)
#define dUdz(i,j,k) SL(dUdz_,i,j,k)
#define dVdz(i,j,k) SL(dVdz_,i,j,k)
#define dZde_p(i,j,k) SL(dZde_p_,i,j,k)
#define dZde_r(i,j,k) SL(dZde_r_,i,j,k)
#define dZdx_p(i,j,k) SL(dZdx_p_,i,j,k)
#define dZdx_r(i,j,k) SL(dZdx_r_,i,j,k)
  int k1, k2 = 1;
  for (int k = 0; k <= N; k++) {
    k1 = k2;
    k2 = 3 - k1;
    if (k < N) {
      /* slopes at RHO- and PSI-points (:303-340) */
      for (int j = Jstr - 1; j <= Jend + 1; j++)
        for (int i = IstrU - 1; i <= Iend + 1; i++) {
          cff = 0.5 * (pm(i - 1, j) + pm(i, j));
          if (p->masking) cff = cff * umask(i, j);
          if (p->wet_dry) cff = cff * umask_wet(i, j);
          UFx(i, j) = cff * (z_r(i, j, k + 1) - z_r(i - 1, j, k + 1));
        }
      for (int j = JstrV - 1; j <= Jend + 1; j++)
        for (int i = Istr - 1; i <= Iend + 1; i++) {
          cff = 0.5 * (pn(i, j - 1) + pn(i, j));
          if (p->masking) cff = cff * vmask(i, j);
          if (p->wet_dry) cff = cff * vmask_wet(i, j);
          VFe(i, j) = cff * (z_r(i, j, k + 1) - z_r(i, j - 1, k + 1));
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          dZdx_p(i, j, k2) = 0.5 * (UFx(i, j - 1) + UFx(i, j));
          dZde_p(i, j, k2) = 0.5 * (VFe(i - 1, j) + VFe(i, j));
        }
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          dZdx_r(i, j, k2) = 0.5 * (UFx(i, j) + UFx(i + 1, j));
          dZde_r(i, j, k2) = 0.5 * (VFe(i, j) + VFe(i, j + 1));
        }
      /* horizontal gradients of momentum (:345-412) */
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          cff = 0.5 * pm(i, j);
          if (p->masking) cff = cff * rmask(i, j);
          if (p->wet_dry) cff = cff * rmask_wet(i, j);
          dnUdx(i, j, k2) = cff * ((pn(i, j) + pn(i + 1, j)) * u(i + 1, j, k + 1, nrhs) - (pn(i - 1, j) + pn(i, j)) * u(i, j, k + 1, nrhs));
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          cff = 0.125 * (pn(i - 1, j) + pn(i, j) + pn(i - 1, j - 1) + pn(i, j - 1));
          if (p->masking) cff = cff * pmask(i, j);
          if (p->wet_dry) cff = cff * pmask_wet(i, j);
          dmUde(i, j, k2) = cff * ((pm(i - 1, j) + pm(i, j)) * u(i, j, k + 1, nrhs) - (pm(i - 1, j - 1) + pm(i, j - 1)) * u(i, j - 1, k + 1, nrhs));
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          cff = 0.125 * (pm(i - 1, j) + pm(i, j) + pm(i - 1, j - 1) + pm(i, j - 1));
          if (p->masking) cff = cff * pmask(i, j);
          if (p->wet_dry) cff = cff * pmask_wet(i, j);
          dnVdx(i, j, k2) = cff * ((pn(i, j - 1) + pn(i, j)) * v(i, j, k + 1, nrhs) - (pn(i - 1, j - 1) + pn(i - 1, j)) * v(i - 1, j, k + 1, nrhs));
        }
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          cff = 0.5 * pn(i, j);
          if (p->masking) cff = cff * rmask(i, j);
          if (p->wet_dry) cff = cff * rmask_wet(i, j);
          dmVde(i, j, k2) = cff * ((pm(i, j) + pm(i, j + 1)) * v(i, j + 1, k + 1, nrhs) - (pm(i, j - 1) + pm(i, j)) * v(i, j, k + 1, nrhs));
        }
    }
    if (k == 0 || k == N) {
      for (int j = Jstr - 1; j <= Jend + 1; j++)
        for (int i = IstrU - 1; i <= Iend + 1; i++) dUdz(i, j, k2) = 0.0;
      for (int j = JstrV - 1; j <= Jend + 1; j++)
        for (int i = Istr - 1; i <= Iend + 1; i++) dVdz(i, j, k2) = 0.0;
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) { UFsx(i, j, k2) = 0.0; UFse(i, j, k2) = 0.0; }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) { VFsx(i, j, k2) = 0.0; VFse(i, j, k2) = 0.0; }
    } else {
      for (int j = Jstr - 1; j <= Jend + 1; j++)
        for (int i = IstrU - 1; i <= Iend + 1; i++) {
          cff = 1.0 / (0.5 * (z_r(i - 1, j, k + 1) - z_r(i - 1, j, k) + z_r(i, j, k + 1) - z_r(i, j, k)));
          dUdz(i, j, k2) = cff * (u(i, j, k + 1, nrhs) - u(i, j, k, nrhs));
        }
      for (int j = JstrV - 1; j <= Jend + 1; j++)
        for (int i = Istr - 1; i <= Iend + 1; i++) {
          cff = 1.0 / (0.5 * (z_r(i, j - 1, k + 1) - z_r(i, j - 1, k) + z_r(i, j, k + 1) - z_r(i, j, k)));
          dVdz(i, j, k2) = cff * (v(i, j, k + 1, nrhs) - v(i, j, k, nrhs));
        }
    }
    if (k > 0) {
      /* rotated viscous flux along geopotentials, XI- and ETA-components (:463-541) */
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          cff1 = MIN(dZdx_r(i, j, k1), 0.0);
          cff2 = MAX(dZdx_r(i, j, k1), 0.0);
          cff3 = MIN(dZde_r(i, j, k1), 0.0);
          cff4 = MAX(dZde_r(i, j, k1), 0.0);
          cff = Hz(i, j, k) *
                (on_r(i, j) * (dnUdx(i, j, k1) - 0.5 * pn(i, j) * (cff1 * (dUdz(i, j, k1) + dUdz(i + 1, j, k2)) + cff2 * (dUdz(i, j, k2) + dUdz(i + 1, j, k1)))) -
                 om_r(i, j) * (dmVde(i, j, k1) - 0.5 * pm(i, j) * (cff3 * (dVdz(i, j, k1) + dVdz(i, j + 1, k2)) + cff4 * (dVdz(i, j, k2) + dVdz(i, j + 1, k1)))));
          if (p->masking) cff = cff * rmask(i, j);
          if (p->wet_dry) cff = cff * rmask_wet(i, j);
          UFx(i, j) = on_r(i, j) * on_r(i, j) * visc2_r(i, j) * cff;
          VFe(i, j) = om_r(i, j) * om_r(i, j) * visc2_r(i, j) * cff;
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          pm_p = 0.25 * (pm(i - 1, j - 1) + pm(i - 1, j) + pm(i, j - 1) + pm(i, j));
          pn_p = 0.25 * (pn(i - 1, j - 1) + pn(i - 1, j) + pn(i, j - 1) + pn(i, j));
          cff1 = MIN(dZdx_p(i, j, k1), 0.0);
          cff2 = MAX(dZdx_p(i, j, k1), 0.0);
          cff3 = MIN(dZde_p(i, j, k1), 0.0);
          cff4 = MAX(dZde_p(i, j, k1), 0.0);
          cff = 0.25 * (Hz(i - 1, j, k) + Hz(i, j, k) + Hz(i - 1, j - 1, k) + Hz(i, j - 1, k)) *
                (on_p(i, j) * (dnVdx(i, j, k1) - 0.5 * pn_p * (cff1 * (dVdz(i - 1, j, k1) + dVdz(i, j, k2)) + cff2 * (dVdz(i - 1, j, k2) + dVdz(i, j, k1)))) +
                 om_p(i, j) * (dmUde(i, j, k1) - 0.5 * pm_p * (cff3 * (dUdz(i, j - 1, k1) + dUdz(i, j, k2)) + cff4 * (dUdz(i, j - 1, k2) + dUdz(i, j, k1)))));
          if (p->masking) cff = cff * pmask(i, j);
          if (p->wet_dry) cff = cff * pmask_wet(i, j);
          UFe(i, j) = om_p(i, j) * om_p(i, j) * visc2_p(i, j) * cff;
          VFx(i, j) = on_p(i, j) * on_p(i, j) * visc2_p(i, j) * cff;
        }
      /* vertical flux due to the sloping coordinate surfaces (:546-700) */
      if (k < N) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = IstrU; i <= Iend; i++) {
            cff = 0.25 * (visc2_r(i - 1, j) + visc2_r(i, j));
            fac1 = cff * on_u(i, j);
            fac2 = cff * om_u(i, j);
            cff = 0.5 * (pn(i - 1, j) + pn(i, j));
            dnUdz = cff * dUdz(i, j, k2);
            dnVdz = cff * 0.25 * (dVdz(i - 1, j + 1, k2) + dVdz(i, j + 1, k2) + dVdz(i - 1, j, k2) + dVdz(i, j, k2));
            cff = 0.5 * (pm(i - 1, j) + pm(i, j));
            dmUdz = cff * dUdz(i, j, k2);
            dmVdz = cff * 0.25 * (dVdz(i - 1, j + 1, k2) + dVdz(i, j + 1, k2) + dVdz(i - 1, j, k2) + dVdz(i, j, k2));
            cff1 = MIN(dZdx_r(i - 1, j, k1), 0.0);
            cff2 = MIN(dZdx_r(i, j, k2), 0.0);
            cff3 = MAX(dZdx_r(i - 1, j, k2), 0.0);
            cff4 = MAX(dZdx_r(i, j, k1), 0.0);
            UFsx(i, j, k2) = fac1 * (cff1 * (cff1 * dnUdz - dnUdx(i - 1, j, k1)) + cff2 * (cff2 * dnUdz - dnUdx(i, j, k2)) +
                                     cff3 * (cff3 * dnUdz - dnUdx(i - 1, j, k2)) + cff4 * (cff4 * dnUdz - dnUdx(i, j, k1)));
            cff1 = MIN(dZde_p(i, j, k1), 0.0);
            cff2 = MIN(dZde_p(i, j + 1, k2), 0.0);
            cff3 = MAX(dZde_p(i, j, k2), 0.0);
            cff4 = MAX(dZde_p(i, j + 1, k1), 0.0);
            UFse(i, j, k2) = fac2 * (cff1 * (cff1 * dmUdz - dmUde(i, j, k1)) + cff2 * (cff2 * dmUdz - dmUde(i, j + 1, k2)) +
                                     cff3 * (cff3 * dmUdz - dmUde(i, j, k2)) + cff4 * (cff4 * dmUdz - dmUde(i, j + 1, k1)));
            cff1 = MIN(dZde_p(i, j, k1), 0.0);
            cff2 = MIN(dZde_p(i, j + 1, k2), 0.0);
            cff3 = MAX(dZde_p(i, j, k2), 0.0);
            cff4 = MAX(dZde_p(i, j + 1, k1), 0.0);
            cff5 = MIN(dZdx_p(i, j, k1), 0.0);
            cff6 = MIN(dZdx_p(i, j + 1, k2), 0.0);
            cff7 = MAX(dZdx_p(i, j, k2), 0.0);
            cff8 = MAX(dZdx_p(i, j + 1, k1), 0.0);
            UFsx(i, j, k2) = UFsx(i, j, k2) +
                             fac1 * (cff1 * (cff5 * dnVdz - dnVdx(i, j, k1)) + cff2 * (cff6 * dnVdz - dnVdx(i, j + 1, k2)) +
                                     cff3 * (cff7 * dnVdz - dnVdx(i, j, k2)) + cff4 * (cff8 * dnVdz - dnVdx(i, j + 1, k1)));
            cff1 = MIN(dZdx_r(i - 1, j, k1), 0.0);
            cff2 = MIN(dZdx_r(i, j, k2), 0.0);
            cff3 = MAX(dZdx_r(i - 1, j, k2), 0.0);
            cff4 = MAX(dZdx_r(i, j, k1), 0.0);
            cff5 = MIN(dZde_r(i - 1, j, k1), 0.0);
            cff6 = MIN(dZde_r(i, j, k2), 0.0);
            cff7 = MAX(dZde_r(i - 1, j, k2), 0.0);
            cff8 = MAX(dZde_r(i, j, k1), 0.0);
            UFse(i, j, k2) = UFse(i, j, k2) -
                             fac2 * (cff1 * (cff5 * dmVdz - dmVde(i - 1, j, k1)) + cff2 * (cff6 * dmVdz - dmVde(i, j, k2)) +
                                     cff3 * (cff7 * dmVdz - dmVde(i - 1, j, k2)) + cff4 * (cff8 * dmVdz - dmVde(i, j, k1)));
          }
        for (int j = JstrV; j <= Jend; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 0.25 * (visc2_r(i, j - 1) + visc2_r(i, j));
            fac1 = cff * on_v(i, j);
            fac2 = cff * om_v(i, j);
            cff = 0.5 * (pn(i, j - 1) + pn(i, j));
            dnUdz = cff * 0.25 * (dUdz(i, j, k2) + dUdz(i + 1, j, k2) + dUdz(i, j - 1, k2) + dUdz(i + 1, j - 1, k2));
            dnVdz = cff * dVdz(i, j, k2);
            cff = 0.5 * (pm(i, j - 1) + pm(i, j));
            dmUdz = cff * 0.25 * (dUdz(i, j, k2) + dUdz(i + 1, j, k2) + dUdz(i, j - 1, k2) + dUdz(i + 1, j - 1, k2));
            dmVdz = cff * dVdz(i, j, k2);
            cff1 = MIN(dZdx_p(i, j, k1), 0.0);
            cff2 = MIN(dZdx_p(i + 1, j, k2), 0.0);
            cff3 = MAX(dZdx_p(i, j, k2), 0.0);
            cff4 = MAX(dZdx_p(i + 1, j, k1), 0.0);
            VFsx(i, j, k2) = fac1 * (cff1 * (cff1 * dnVdz - dnVdx(i, j, k1)) + cff2 * (cff2 * dnVdz - dnVdx(i + 1, j, k2)) +
                                     cff3 * (cff3 * dnVdz - dnVdx(i, j, k2)) + cff4 * (cff4 * dnVdz - dnVdx(i + 1, j, k1)));
            cff1 = MIN(dZde_r(i, j - 1, k1), 0.0);
            cff2 = MIN(dZde_r(i, j, k2), 0.0);
            cff3 = MAX(dZde_r(i, j - 1, k2), 0.0);
            cff4 = MAX(dZde_r(i, j, k1), 0.0);
            VFse(i, j, k2) = fac2 * (cff1 * (cff1 * dmVdz - dmVde(i, j - 1, k1)) + cff2 * (cff2 * dmVdz - dmVde(i, j, k2)) +
                                     cff3 * (cff3 * dmVdz - dmVde(i, j - 1, k2)) + cff4 * (cff4 * dmVdz - dmVde(i, j, k1)));
            cff1 = MIN(dZde_r(i, j - 1, k1), 0.0);
            cff2 = MIN(dZde_r(i, j, k2), 0.0);
            cff3 = MAX(dZde_r(i, j - 1, k2), 0.0);
            cff4 = MAX(dZde_r(i, j, k1), 0.0);
            cff5 = MIN(dZdx_r(i, j - 1, k1), 0.0);
            cff6 = MIN(dZdx_r(i, j, k2), 0.0);
            cff7 = MAX(dZdx_r(i, j - 1, k2), 0.0);
            cff8 = MAX(dZdx_r(i, j, k1), 0.0);
            VFsx(i, j, k2) = VFsx(i, j, k2) -
                             fac1 * (cff1 * (cff5 * dnUdz - dnUdx(i, j - 1, k1)) + cff2 * (cff6 * dnUdz - dnUdx(i, j, k2)) +
                                     cff3 * (cff7 * dnUdz - dnUdx(i, j - 1, k2)) + cff4 * (cff8 * dnUdz - dnUdx(i, j, k1)));
            cff1 = MIN(dZdx_p(i, j, k1), 0.0);
            cff2 = MIN(dZdx_p(i + 1, j, k2), 0.0);
            cff3 = MAX(dZdx_p(i, j, k2), 0.0);
            cff4 = MAX(dZdx_p(i + 1, j, k1), 0.0);
            cff5 = MIN(dZde_p(i, j, k1), 0.0);
            cff6 = MIN(dZde_p(i + 1, j, k2), 0.0);
            cff7 = MAX(dZde_p(i, j, k2), 0.0);
            cff8 = MAX(dZde_p(i + 1, j, k1), 0.0);
            VFse(i, j, k2) = VFse(i, j, k2) +
                             fac2 * (cff1 * (cff5 * dmUdz - dmUde(i, j, k1)) + cff2 * (cff6 * dmUdz - dmUde(i + 1, j, k2)) +
                                     cff3 * (cff7 * dmUdz - dmUde(i, j, k2)) + cff4 * (cff8 * dmUdz - dmUde(i + 1, j, k1)));
          }
      }
      /* time-step the term; momentum is HzU, HzV here (:710-752) */
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          cff = dt * 0.25 * (pm(i - 1, j) + pm(i, j)) * (pn(i - 1, j) + pn(i, j));
          cff1 = 0.5 * (pn(i - 1, j) + pn(i, j)) * (UFx(i, j) - UFx(i - 1, j));
          cff2 = 0.5 * (pm(i - 1, j) + pm(i, j)) * (UFe(i, j + 1) - UFe(i, j));
          cff3 = UFsx(i, j, k2) - UFsx(i, j, k1);
          cff4 = UFse(i, j, k2) - UFse(i, j, k1);
          cff5 = cff * (cff1 + cff2);
          cff6 = dt * (cff3 + cff4);
          rufrc(i, j) = rufrc(i, j) + cff1 + cff2 + cff3 + cff4;
          u(i, j, k, nnew) = u(i, j, k, nnew) + cff5 + cff6;
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = dt * 0.25 * (pm(i, j) + pm(i, j - 1)) * (pn(i, j) + pn(i, j - 1));
          cff1 = 0.5 * (pn(i, j - 1) + pn(i, j)) * (VFx(i + 1, j) - VFx(i, j));
          cff2 = 0.5 * (pm(i, j - 1) + pm(i, j)) * (VFe(i, j) - VFe(i, j - 1));
          cff3 = VFsx(i, j, k2) - VFsx(i, j, k1);
          cff4 = VFse(i, j, k2) - VFse(i, j, k1);
          cff5 = cff * (cff1 - cff2);
          cff6 = dt * (cff3 + cff4);
          rvfrc(i, j) = rvfrc(i, j) + cff1 - cff2 + cff3 + cff4;
          v(i, j, k, nnew) = v(i, j, k, nnew) + cff5 + cff6;
        }
    }
  }
  free(UFe_); free(VFe_); free(UFx_); free(VFx_); free(UFse_); free(UFsx_); free(VFse_); free(VFsx_);
  free(dmUde_); free(dmVde_); free(dnUdx_); free(dnVdx_); free(dUdz_); free(dVdz_);
  free(dZde_p_); free(dZde_r_); free(dZdx_p_); free(dZdx_r_);
  return 0;
}
