// advect.h -- tracer advective flux stencils shared by the predictor
// (pre_step3d.F:342-915) and the corrector (step3d_t.F:363-1210): the
// reference writes the same C2 / U3 / A4 / C4 flux formulas in both files.
#pragma once
#include "roms_hip.h"

template <int HADV>
__device__ __forceinline__ double hflux(double Hflx, double tm1, double t0, double dm1, double d0, double dp1)
{
  // flux through the face between cell "m1" and cell "0"; d* are first
  // differences centred on faces (dm1 = face-1, d0 = this face, dp1 = face+1).
  if constexpr (HADV == ADV_C2) {
    return Hflx * 0.5 * (tm1 + t0);
  } else if constexpr (HADV == ADV_MPDATA) {
    // first-order upstream (the MPDATA predictor, pre_step3d.F:364-386 / step3d_t.F:409-428)
    const double cff1 = fmax(Hflx, 0.0), cff2 = fmin(Hflx, 0.0);
    return cff1 * tm1 + cff2 * t0;
  } else if constexpr (HADV == ADV_U3) {
    const double curv_m1 = d0 - dm1;     // curv at cell m1
    const double curv_0 = dp1 - d0;      // curv at cell 0
    const double cff1 = 1.0 / 6.0;
    return Hflx * 0.5 * (tm1 + t0) -
           cff1 * (curv_m1 * fmax(Hflx, 0.0) + curv_0 * fmin(Hflx, 0.0));
  } else if constexpr (HADV == ADV_A4) {
    const double eps = 1.0E-16;
    double g_m1, g_0;
    double cff = 2.0 * d0 * dm1;
    g_m1 = (cff > eps) ? cff / (d0 + dm1) : 0.0;
    cff = 2.0 * dp1 * d0;
    g_0 = (cff > eps) ? cff / (dp1 + d0) : 0.0;
    const double cff2 = 1.0 / 3.0;
    return Hflx * 0.5 * (tm1 + t0 - cff2 * (g_0 - g_m1));
  } else {  // C4 / SU3
    const double g_m1 = 0.5 * (d0 + dm1);
    const double g_0 = 0.5 * (dp1 + d0);
    const double cff2 = 1.0 / 3.0;
    return Hflx * 0.5 * (tm1 + t0 - cff2 * (g_0 - g_m1));
  }
}

// Vertical flux FC(k), k = 1..N-1, for the non-spline schemes.
template <int VADV>
__device__ __forceinline__ double vflux(int k, int N, double Wk, double tkm1, double tk, double tkp1, double tkp2,
                                        double a4_cf_k, double a4_cf_kp1)
{
  if constexpr (VADV == ADV_C2) {
    return Wk * 0.5 * (tk + tkp1);
  } else if constexpr (VADV == ADV_MPDATA) {
    // first-order upstream, pre_step3d.F:729-748
    const double cff1 = fmax(Wk, 0.0), cff2 = fmin(Wk, 0.0);
    return cff1 * tk + cff2 * tkp1;
  } else if constexpr (VADV == ADV_A4) {
    const double cff1 = 1.0 / 3.0;
    return Wk * 0.5 * (tk + tkp1 - cff1 * (a4_cf_kp1 - a4_cf_k));
  } else {  // C4 / SU3, step3d_t.F:1094+
    const double cff1 = 0.5, cff2 = 7.0 / 12.0, cff3 = 1.0 / 12.0;
    if (k == 1) return Wk * (cff1 * tk + cff2 * tkp1 - cff3 * tkp2);
    if (k == N - 1) return Wk * (cff1 * tkp1 + cff2 * tk - cff3 * tkm1);
    return Wk * (cff2 * (tk + tkp1) - cff3 * (tkm1 + tkp2));
  }
}

