/*
 * oracle_mpdata_adiff.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * mpdata_adiff_tile: MPDATA anti-diffusive velocities Ua, Va, Wa with the
 * third-order terms (MPDATA_HOT, defined locally at mpdata_adiff.F:2) and the
 * flux-corrected-transport limiter (ROMS/Nonlinear/mpdata_adiff.F:38-1105; no
 * MASKING, no WET_DRY).  Pinned: the reference file compiles stand-alone and this
 * restatement is compared with it bit for bit (tests/test_ref_pinning.py).
 *
 * Ta, Ua, Va, oHz are tile-private (IminS:ImaxS,JminS:JmaxS,N) arrays as in the
 * caller step3d_t_tile; Wa is (IminS:ImaxS,JminS:JmaxS,0:N); t3 is the module
 * array t(:,:,:,3,itrc).  Every expression keeps the reference's association
 * order (Fortran evaluates a*b*c and a+b+c left to right, as C does).
 */
#include "oracle.h"

#define WS3W(i,j,k) (WS2(i,j) + (long)(k) * nis * njs)   /* (..,..,0:N) */
#define MAX12(a,b,c,d,e,f,g,h,i_,j_,k_,l) MAX(MAX(MAX(MAX(MAX(MAX(MAX(MAX(MAX(MAX(MAX(a,b),c),d),e),f),g),h),i_),j_),k_),l)
#define MIN12(a,b,c,d,e,f,g,h,i_,j_,k_,l) MIN(MIN(MIN(MIN(MIN(MIN(MIN(MIN(MIN(MIN(MIN(a,b),c),d),e),f),g),h),i_),j_),k_),l)

int oracle_mpdata_adiff(const roms_bounds_t *b, const roms_params_t *p, const roms_step_idx_t *s,
                        roms_fields_t *F, const double *oHz_, const double *t3_,
                        double *Ta_, double *Ua_, double *Va_, double *Wa_)
{
  ORACLE_PROLOGUE
  (void)s;
  const double dt = p->dt;
  const double eps = 1.0E-18, eps2 = 1.0E-10, fac = 1.0;
  const int mk = p->masking;
  /* MASKING: face / point masks as multipliers (exactly 1 without the option: x * 1.0 = x bit for bit) */
#define UM(i,j) (mk ? umask(i,j) : 1.0)
#define VM(i,j) (mk ? vmask(i,j) : 1.0)
  /* limiter masks, mpdata_adiff.F:826-835: land values do not enter Tmax, and Tmin takes a huge value there */
  const double Large = 1.0E+20;
#define MUP(i,j) (mk ? rmask(i,j) : 1.0)
#define MDN(i,j) (mk ? MAX(1.0, MIN(Large, (1.0 - rmask(i,j)) * Large)) : 1.0)

  double *C_ = walloc(nis * (N + 1)), *Wm_ = walloc(nis * (N + 1));
  double *beta_dn_ = walloc(nis * njs * N), *beta_up_ = walloc(nis * njs * N), *odz_ = walloc(nis * njs * N);
#define Ta(i,j,k)  Ta_[WS3(i,j,k)]
#define Ua(i,j,k)  Ua_[WS3(i,j,k)]
#define Va(i,j,k)  Va_[WS3(i,j,k)]
#define Wa(i,j,k)  Wa_[WS3W(i,j,k)]
#define oHz(i,j,k) oHz_[WS3(i,j,k)]
#define t3(i,j,k)  t3_[I3(i,j,k)]
#define odz(i,j,k) odz_[WS3(i,j,k)]
#define beta_dn(i,j,k) beta_dn_[WS3(i,j,k)]
#define beta_up(i,j,k) beta_up_[WS3(i,j,k)]
#define C(i,k)     C_[WSK(i,k)]
#define Wm(i,k)    Wm_[WSK(i,k)]

  /* boundary values of Ta, mpdata_adiff.F:170-240 */
  if (!EWperiodic) {
    if (west_edge)
      for (int k = 1; k <= N; k++)
        for (int j = JstrVm2; j <= Jendp2i; j++) Ta(Istr - 1, j, k) = Ta(Istr, j, k);
    if (east_edge)
      for (int k = 1; k <= N; k++)
        for (int j = JstrVm2; j <= Jendp2i; j++) Ta(Iend + 1, j, k) = Ta(Iend, j, k);
  }
  if (!NSperiodic) {
    if (south_edge)
      for (int k = 1; k <= N; k++)
        for (int i = IstrUm2; i <= Iendp2i; i++) Ta(i, Jstr - 1, k) = Ta(i, Jstr, k);
    if (north_edge)
      for (int k = 1; k <= N; k++)
        for (int i = IstrUm2; i <= Iendp2i; i++) Ta(i, Jend + 1, k) = Ta(i, Jend, k);
  }
  if (!(EWperiodic || NSperiodic)) {
    if (south_edge && west_edge)
      for (int k = 1; k <= N; k++)
        Ta(Istr - 1, Jstr - 1, k) = 0.5 * (Ta(Istr, Jstr - 1, k) + Ta(Istr - 1, Jstr, k));
    if (south_edge && east_edge)
      for (int k = 1; k <= N; k++)
        Ta(Iend + 1, Jstr - 1, k) = 0.5 * (Ta(Iend + 1, Jstr, k) + Ta(Iend, Jstr - 1, k));
    if (north_edge && west_edge)
      for (int k = 1; k <= N; k++)
        Ta(Istr - 1, Jend + 1, k) = 0.5 * (Ta(Istr - 1, Jend, k) + Ta(Istr, Jend + 1, k));
    if (north_edge && east_edge)
      for (int k = 1; k <= N; k++)
        Ta(Iend + 1, Jend + 1, k) = 0.5 * (Ta(Iend + 1, Jend, k) + Ta(Iend, Jend + 1, k));
  }
  /* :244-252 */
  for (int k = 1; k <= N - 1; k++)
    for (int j = Jstrm2; j <= Jendp2; j++)
      for (int i = Istrm2; i <= Iendp2; i++) odz(i, j, k) = 1.0 / (z_r(i, j, k + 1) - z_r(i, j, k));
  const double cff = 1.0 / dt;

  /* ---- anti-diffusive velocity in the XI direction, :258-450 ---- */
  for (int j = JstrV - 1; j <= Jendp1; j++) {
    int k = 1;
    for (int i = IstrUm1; i <= Iendp2; i++) {
      C(i, k) = 0.25 *
                ((Ta(i, j, k + 1) - Ta(i, j, k)) * odz(i, j, k) +
                 (Ta(i - 1, j, k + 1) - Ta(i - 1, j, k)) * odz(i - 1, j, k)) *
                (z_r(i, j, k + 1) - z_r(i, j, k) +
                 z_r(i - 1, j, k + 1) - z_r(i - 1, j, k)) /
                (Ta(i - 1, j, k) + Ta(i, j, k) + eps);
      Wm(i, k) = 0.25 * dt *
                 (W(i - 1, j, k) * odz(i - 1, j, k) * pm(i - 1, j) * pn(i - 1, j) +
                  W(i, j, k) * odz(i, j, k) * pm(i, j) * pn(i, j));
    }
    for (k = 2; k <= N - 1; k++)
      for (int i = IstrU - 1; i <= Iendp2; i++) {
        C(i, k) = 0.0625 *
                  ((Ta(i, j, k + 1) - Ta(i, j, k)) * odz(i, j, k) +
                   (Ta(i, j, k) - Ta(i, j, k - 1)) * odz(i, j, k - 1) +
                   (Ta(i - 1, j, k + 1) - Ta(i - 1, j, k)) * odz(i - 1, j, k) +
                   (Ta(i - 1, j, k) - Ta(i - 1, j, k - 1)) * odz(i - 1, j, k - 1)) *
                  (z_r(i, j, k + 1) - z_r(i, j, k - 1) +
                   z_r(i - 1, j, k + 1) - z_r(i - 1, j, k - 1)) /
                  (Ta(i - 1, j, k) + Ta(i, j, k) + eps);
        Wm(i, k) = 0.25 * dt *
                   ((W(i - 1, j, k - 1) * odz(i - 1, j, k - 1) +
                     W(i - 1, j, k) * odz(i - 1, j, k)) * pm(i - 1, j) * pn(i - 1, j) +
                    (W(i, j, k) * odz(i, j, k) +
                     W(i, j, k - 1) * odz(i, j, k - 1)) * pm(i, j) * pn(i, j));
      }
    k = N;
    for (int i = IstrU - 1; i <= Iendp2; i++) {
      C(i, k) = 0.25 *
                ((Ta(i, j, k) - Ta(i, j, k - 1)) * odz(i, j, k - 1) +
                 (Ta(i - 1, j, k) - Ta(i - 1, j, k - 1)) * odz(i - 1, j, k - 1)) *
                (z_r(i, j, k) - z_r(i, j, k - 1) +
                 z_r(i - 1, j, k) - z_r(i - 1, j, k - 1)) /
                (Ta(i - 1, j, k) + Ta(i, j, k) + eps);
      Wm(i, k) = 0.25 * dt *
                 (W(i - 1, j, k - 1) * odz(i - 1, j, k - 1) * pm(i - 1, j) * pn(i - 1, j) +
                  W(i, j, k - 1) * odz(i, j, k - 1) * pm(i, j) * pn(i, j));
    }
    for (k = 1; k <= N; k++)
      for (int i = IstrU - 1; i <= Iendp2; i++) {
        if ((Ta(i - 1, j, k) <= 0.0) || (Ta(i, j, k) <= 0.0) ||
            (fabs(Ta(i - 1, j, k) - Ta(i, j, k)) <= eps2)) {
          Ua(i, j, k) = 0.0;
        } else {
          const double Ck = C(i, k), Wk = Wm(i, k);
          double A = (Ta(i, j, k) - Ta(i - 1, j, k)) / (Ta(i, j, k) + Ta(i - 1, j, k) + eps);
          /* MASKING (mpdata_adiff.F:288-297): each difference times the mask of its face; VM / UM = 1 without */
          double B = 0.03125 *
                     ((Ta(i, j + 1, k) - Ta(i, j, k)) * (pn(i, j) + pn(i, j + 1)) * VM(i, j + 1) +
                      (Ta(i, j, k) - Ta(i, j - 1, k)) * (pn(i, j - 1) + pn(i, j)) * VM(i, j) +
                      (Ta(i - 1, j + 1, k) - Ta(i - 1, j, k)) * (pn(i - 1, j) + pn(i - 1, j + 1)) * VM(i - 1, j + 1) +
                      (Ta(i - 1, j, k) - Ta(i - 1, j - 1, k)) * (pn(i - 1, j - 1) + pn(i - 1, j)) * VM(i - 1, j));
          B = B * (on_v(i, j) + on_v(i, j + 1) + on_v(i - 1, j) + on_v(i - 1, j + 1)) /
              (Ta(i - 1, j, k) + Ta(i, j, k) + eps);
          const double Um = 0.125 * Huon(i, j, k) * dt * (pm(i, j) + pm(i - 1, j)) * (pn(i, j) + pn(i - 1, j)) *
                            (oHz(i - 1, j, k) + oHz(i, j, k));
          const double Vm = 0.03125 * dt *
                            (Hvom(i - 1, j, k) * (pm(i - 1, j) + pm(i - 1, j - 1)) *
                                 (pn(i - 1, j) + pn(i - 1, j - 1)) * (oHz(i - 1, j, k) + oHz(i - 1, j - 1, k)) +
                             Hvom(i - 1, j + 1, k) * (pm(i - 1, j + 1) + pm(i - 1, j)) *
                                 (pn(i - 1, j + 1) + pn(i - 1, j)) * (oHz(i - 1, j + 1, k) + oHz(i - 1, j, k)) +
                             Hvom(i, j, k) * (pm(i, j) + pm(i, j - 1)) *
                                 (pn(i, j) + pn(i, j - 1)) * (oHz(i, j, k) + oHz(i, j - 1, k)) +
                             Hvom(i, j + 1, k) * (pm(i, j + 1) + pm(i, j)) *
                                 (pn(i, j + 1) + pn(i, j)) * (oHz(i, j + 1, k) + oHz(i, j, k)));
          const double X = (fabs(Um) - Um * Um) * A - B * Um * Vm - Ck * Um * Wk;
          const double Y = (fabs(Vm) - Vm * Vm) * B - A * Um * Vm - Ck * Vm * Wk;
          const double Z = (fabs(Wk) - Wk * Wk) * Ck - A * Um * Wk - B * Vm * Wk;
          const double AA = A * A, BB = B * B, CC = Ck * Ck, AB = A * B, AC = A * Ck;
          const double XX = X * X, YY = Y * Y, ZZ = Z * Z, XY = X * Y, XZ = X * Z;
          const double sig_alfa = 1.0 / (1.0 - fabs(A) + eps);
          const double sig_beta = -A / ((1.0 - fabs(A)) * (1.0 - AA) + eps);
          const double sig_gama = 2.0 * fabs(AA * A) / ((1.0 - fabs(A)) * (1.0 - AA) * (1.0 - fabs(AA * A)) + eps);
          const double sig_a = -B / ((1.0 - fabs(A)) * (1.0 - fabs(AB)) + eps);
          const double sig_b = AB / ((1.0 - fabs(A)) * (1.0 - AA * fabs(B)) + eps) *
                               (fabs(B) / (1.0 - fabs(AB) + eps) + 2.0 * A / (1.0 - AA + eps));
          const double sig_c = fabs(A) * BB / ((1.0 - fabs(A)) * (1.0 - BB * fabs(A)) * (1.0 - fabs(AB)) + eps);
          const double sig_d = -Ck / ((1.0 - fabs(A)) * (1.0 - fabs(AC)) + eps);
          const double sig_e = AC / ((1.0 - fabs(A)) * (1.0 - AA * fabs(Ck)) + eps) *
                               (fabs(Ck) / (1.0 - fabs(AC) + eps) + 2.0 * A / (1.0 - AA + eps));
          const double sig_f = fabs(A) * CC / ((1.0 - fabs(A)) * (1.0 - CC * fabs(A)) * (1.0 - fabs(AC)) + eps);
          double ua = sig_alfa * X + sig_beta * XX + sig_gama * XX * X + sig_a * XY + sig_b * XX * Y +
                      sig_c * X * YY + sig_d * XZ + sig_e * XX * Z + sig_f * X * ZZ;
          Ua(i, j, k) = MIN(fabs(ua), fac * fabs(Um)) * copysign(1.0, ua);
          if (mk) Ua(i, j, k) = Ua(i, j, k) * umask(i, j);                    /* :395 */
          if (p->wet_dry) Ua(i, j, k) = Ua(i, j, k) * umask_wet(i, j);           /* WET_DRY: the next block */
        }
      }
  }

  /* ---- anti-diffusive velocity in the ETA direction, :452-640 ---- */
  for (int j = JstrVm1; j <= Jendp2; j++) {
    int k = 1;
    for (int i = IstrU - 1; i <= Iendp1; i++) {
      C(i, k) = 0.25 *
                ((Ta(i, j, k + 1) - Ta(i, j, k)) * odz(i, j, k) +
                 (Ta(i, j - 1, k + 1) - Ta(i, j - 1, k)) * odz(i, j - 1, k)) *
                (z_r(i, j, k + 1) - z_r(i, j, k) +
                 z_r(i, j - 1, k + 1) - z_r(i, j - 1, k)) /
                (Ta(i, j - 1, k) + Ta(i, j, k) + eps);
      Wm(i, k) = 0.25 * dt *
                 (W(i, j - 1, k) * odz(i, j - 1, k) * pm(i, j - 1) * pn(i, j - 1) +
                  W(i, j, k) * odz(i, j, k) * pm(i, j) * pn(i, j));
    }
    for (k = 2; k <= N - 1; k++)
      for (int i = IstrU - 1; i <= Iendp1; i++) {
        C(i, k) = 0.0625 *
                  ((Ta(i, j, k + 1) - Ta(i, j, k)) * odz(i, j, k) +
                   (Ta(i, j, k) - Ta(i, j, k - 1)) * odz(i, j, k - 1) +
                   (Ta(i, j - 1, k + 1) - Ta(i, j - 1, k)) * odz(i, j - 1, k) +
                   (Ta(i, j - 1, k) - Ta(i, j - 1, k - 1)) * odz(i, j - 1, k - 1)) *
                  (z_r(i, j, k + 1) - z_r(i, j, k - 1) +
                   z_r(i, j - 1, k + 1) - z_r(i, j - 1, k - 1)) /
                  (Ta(i, j - 1, k) + Ta(i, j, k) + eps);
        Wm(i, k) = 0.25 * dt *
                   ((W(i, j - 1, k - 1) * odz(i, j - 1, k - 1) +
                     W(i, j - 1, k) * odz(i, j - 1, k)) * pm(i, j - 1) * pn(i, j - 1) +
                    (W(i, j, k) * odz(i, j, k) +
                     W(i, j, k - 1) * odz(i, j, k - 1)) * pm(i, j) * pn(i, j));
      }
    k = N;
    for (int i = IstrU - 1; i <= Iendp1; i++) {
      C(i, k) = 0.25 *
                ((Ta(i, j, k) - Ta(i, j, k - 1)) * odz(i, j, k - 1) +
                 (Ta(i, j - 1, k) - Ta(i, j - 1, k - 1)) * odz(i, j - 1, k - 1)) *
                (z_r(i, j, k) - z_r(i, j, k - 1) +
                 z_r(i, j - 1, k) - z_r(i, j - 1, k - 1)) /
                (Ta(i, j - 1, k) + Ta(i, j, k) + eps);
      Wm(i, k) = 0.25 * dt *
                 (W(i, j - 1, k - 1) * odz(i, j - 1, k - 1) * pm(i, j - 1) * pn(i, j - 1) +
                  W(i, j, k - 1) * odz(i, j, k - 1) * pm(i, j) * pn(i, j));
    }
    for (k = 1; k <= N; k++)
      for (int i = IstrU - 1; i <= Iendp1; i++) {
        if ((Ta(i, j - 1, k) <= 0.0) || (Ta(i, j, k) <= 0.0) ||
            (fabs(Ta(i, j - 1, k) - Ta(i, j, k)) <= eps2)) {
          Va(i, j, k) = 0.0;
        } else {
          const double Ck = C(i, k), Wk = Wm(i, k);
          double A = 0.03125 *                                                  /* MASKING, :458-467 */
                     ((Ta(i + 1, j, k) - Ta(i, j, k)) * (pm(i + 1, j) + pm(i, j)) * UM(i + 1, j) +
                      (Ta(i, j, k) - Ta(i - 1, j, k)) * (pm(i - 1, j) + pm(i, j)) * UM(i, j) +
                      (Ta(i + 1, j - 1, k) - Ta(i, j - 1, k)) * (pm(i + 1, j - 1) + pm(i, j - 1)) * UM(i + 1, j - 1) +
                      (Ta(i, j - 1, k) - Ta(i - 1, j - 1, k)) * (pm(i - 1, j - 1) + pm(i, j - 1)) * UM(i, j - 1));
          A = A * (om_u(i, j) + om_u(i + 1, j) + om_u(i, j - 1) + om_u(i + 1, j - 1)) /
              (Ta(i, j - 1, k) + Ta(i, j, k) + eps);
          const double B = (Ta(i, j, k) - Ta(i, j - 1, k)) / (Ta(i, j, k) + Ta(i, j - 1, k) + eps);
          const double Um = 0.03125 * dt *
                            (Huon(i + 1, j, k) * (pm(i + 1, j) + pm(i, j)) *
                                 (pn(i + 1, j) + pn(i, j)) * (oHz(i + 1, j, k) + oHz(i, j, k)) +
                             Huon(i + 1, j - 1, k) * (pm(i + 1, j - 1) + pm(i, j - 1)) *
                                 (pn(i + 1, j - 1) + pn(i, j - 1)) * (oHz(i + 1, j - 1, k) + oHz(i, j - 1, k)) +
                             Huon(i, j, k) * (pm(i - 1, j) + pm(i, j)) *
                                 (pn(i - 1, j) + pn(i, j)) * (oHz(i - 1, j, k) + oHz(i, j, k)) +
                             Huon(i, j - 1, k) * (pm(i - 1, j - 1) + pm(i, j - 1)) *
                                 (pn(i - 1, j - 1) + pn(i, j - 1)) * (oHz(i - 1, j - 1, k) + oHz(i, j - 1, k)));
          const double Vm = 0.125 * Hvom(i, j, k) * dt * (pn(i, j - 1) + pn(i, j)) * (pm(i, j - 1) + pm(i, j)) *
                            (oHz(i, j - 1, k) + oHz(i, j, k));
          const double X = (fabs(Um) - Um * Um) * A - B * Um * Vm - Ck * Um * Wk;
          const double Y = (fabs(Vm) - Vm * Vm) * B - A * Um * Vm - Ck * Vm * Wk;
          const double Z = (fabs(Wk) - Wk * Wk) * Ck - A * Um * Wk - B * Vm * Wk;
          const double AA = A * A, BB = B * B, CC = Ck * Ck, AB = A * B, BC = B * Ck;
          const double XX = X * X, YY = Y * Y, ZZ = Z * Z, XY = X * Y, YZ = Y * Z;
          const double sig_alfa = 1.0 / (1.0 - fabs(B) + eps);
          const double sig_beta = -B / ((1.0 - fabs(B)) * (1.0 - BB) + eps);
          const double sig_gama = 2.0 * fabs(BB * B) / ((1.0 - fabs(B)) * (1.0 - BB) * (1.0 - fabs(BB * B)) + eps);
          const double sig_a = -A / ((1.0 - fabs(B)) * (1.0 - fabs(AB)) + eps);
          const double sig_b = AB / ((1.0 - fabs(B)) * (1.0 - BB * fabs(A)) + eps) *
                               (fabs(A) / (1.0 - fabs(AB) + eps) + 2.0 * B / (1.0 - BB + eps));
          const double sig_c = fabs(B) * AA / ((1.0 - fabs(B)) * (1.0 - AA * fabs(B)) * (1.0 - fabs(AB)) + eps);
          const double sig_d = -Ck / ((1.0 - fabs(B)) * (1.0 - fabs(BC)) + eps);
          const double sig_e = BC / ((1.0 - fabs(B)) * (1.0 - BB * fabs(Ck)) + eps) *
                               (fabs(Ck) / (1.0 - fabs(BC) + eps) + 2.0 * B / (1.0 - BB + eps));
          const double sig_f = fabs(B) * CC / ((1.0 - fabs(B)) * (1.0 - CC * fabs(B)) * (1.0 - fabs(BC)) + eps);
          double va = sig_alfa * Y + sig_beta * YY + sig_gama * YY * Y + sig_a * XY + sig_b * Y * XX +
                      sig_c * YY * X + sig_d * YZ + sig_e * YY * Z + sig_f * Y * ZZ;
          Va(i, j, k) = MIN(fabs(va), fac * fabs(Vm)) * copysign(1.0, va);
          if (mk) Va(i, j, k) = Va(i, j, k) * vmask(i, j);                    /* :568 */
          if (p->wet_dry) Va(i, j, k) = Va(i, j, k) * vmask_wet(i, j);           /* WET_DRY: the next block */
        }
      }
  }

  /* closed / gradient walls for the horizontal pair, :577-640: the rule follows LBC(side, isBu3d = isUvel / isBv3d =
   * isVvel)%closed (mod_ncparam.F:1235-1236), the 3-D momentum's condition on that side */
  if (!EWperiodic) {
    if (west_edge)
      for (int k = 1; k <= N; k++)
        for (int j = Jstrm1; j <= Jendp1; j++)
          Ua(Istr, j, k) = (o_lbc(p, LBS_WEST, LBV_U) == LBC_CLOSED) ? 0.0 : Ua(Istr + 1, j, k);
    if (east_edge)
      for (int k = 1; k <= N; k++)
        for (int j = Jstrm1; j <= Jendp1; j++)
          Ua(Iend + 1, j, k) = (o_lbc(p, LBS_EAST, LBV_U) == LBC_CLOSED) ? 0.0 : Ua(Iend, j, k);
  }
  if (!NSperiodic) {
    if (south_edge)
      for (int k = 1; k <= N; k++)
        for (int i = Istrm1; i <= Iendp1; i++)
          Va(i, Jstr, k) = (o_lbc(p, LBS_SOUTH, LBV_V) == LBC_CLOSED) ? 0.0 : Va(i, Jstr + 1, k);
    if (north_edge)
      for (int k = 1; k <= N; k++)
        for (int i = Istrm1; i <= Iendp1; i++)
          Va(i, Jend + 1, k) = (o_lbc(p, LBS_NORTH, LBV_V) == LBC_CLOSED) ? 0.0 : Va(i, Jend, k);
  }

  /* ---- anti-diffusive velocity in the vertical, :714-840 ---- */
  for (int j = JstrV - 1; j <= Jendp1; j++) {
    for (int k = 1; k <= N - 1; k++)
      for (int i = IstrU - 1; i <= Iendp1; i++) {
        if ((Ta(i, j, k) <= 0.0) || (Ta(i, j, k + 1) <= 0.0) ||
            (fabs(Ta(i, j, k) - Ta(i, j, k + 1)) <= eps2)) {
          Wa(i, j, k) = 0.0;
        } else {
          C(i, k) = (Ta(i, j, k + 1) - Ta(i, j, k)) / (Ta(i, j, k + 1) + Ta(i, j, k) + eps);
          const double Ck = C(i, k);
          double A = 0.0625 *                                                   /* MASKING, :662-679 */
                     ((Ta(i + 1, j, k + 1) - Ta(i, j, k + 1)) * (pm(i + 1, j) + pm(i, j)) * UM(i + 1, j) +
                      (Ta(i, j, k + 1) - Ta(i - 1, j, k + 1)) * (pm(i, j) + pm(i - 1, j)) * UM(i, j) +
                      (Ta(i + 1, j, k) - Ta(i, j, k)) * (pm(i + 1, j) + pm(i, j)) * UM(i + 1, j) +
                      (Ta(i, j, k) - Ta(i - 1, j, k)) * (pm(i, j) + pm(i - 1, j)) * UM(i, j));
          double B = 0.0625 *
                     ((Ta(i, j + 1, k + 1) - Ta(i, j, k + 1)) * (pn(i, j + 1) + pn(i, j)) * VM(i, j + 1) +
                      (Ta(i, j, k + 1) - Ta(i, j - 1, k + 1)) * (pn(i, j) + pn(i, j - 1)) * VM(i, j) +
                      (Ta(i, j + 1, k) - Ta(i, j, k)) * (pn(i, j + 1) + pn(i, j)) * VM(i, j + 1) +
                      (Ta(i, j, k) - Ta(i, j - 1, k)) * (pn(i, j) + pn(i, j - 1)) * VM(i, j));
          A = A * (om_u(i + 1, j) + om_u(i, j)) / (Ta(i, j, k + 1) + Ta(i, j, k) + eps);
          B = B * (on_v(i, j + 1) + on_v(i, j)) / (Ta(i, j, k + 1) + Ta(i, j, k) + eps);
          const double Um = 0.03125 * dt *
                            (Huon(i, j, k) * (pm(i, j) + pm(i - 1, j)) *
                                 (pn(i, j) + pn(i - 1, j)) * (oHz(i, j, k) + oHz(i - 1, j, k)) +
                             Huon(i, j, k + 1) * (pm(i, j) + pm(i - 1, j)) *
                                 (pn(i, j) + pn(i - 1, j)) * (oHz(i, j, k + 1) + oHz(i - 1, j, k + 1)) +
                             Huon(i + 1, j, k) * (pm(i, j) + pm(i + 1, j)) *
                                 (pn(i, j) + pn(i + 1, j)) * (oHz(i, j, k) + oHz(i + 1, j, k)) +
                             Huon(i + 1, j, k + 1) * (pm(i, j) + pm(i + 1, j)) *
                                 (pn(i, j) + pn(i + 1, j)) * (oHz(i, j, k + 1) + oHz(i + 1, j, k + 1)));
          const double Vm = 0.03125 * dt *
                            (Hvom(i, j, k) * (pm(i, j) + pm(i, j - 1)) *
                                 (pn(i, j) + pn(i, j - 1)) * (oHz(i, j, k) + oHz(i, j - 1, k)) +
                             Hvom(i, j, k + 1) * (pm(i, j) + pm(i, j - 1)) *
                                 (pn(i, j) + pn(i, j - 1)) * (oHz(i, j, k + 1) + oHz(i, j - 1, k + 1)) +
                             Hvom(i, j + 1, k) * (pm(i, j) + pm(i, j + 1)) *
                                 (pn(i, j) + pn(i, j + 1)) * (oHz(i, j, k) + oHz(i, j + 1, k)) +
                             Hvom(i, j + 1, k + 1) * (pm(i, j) + pm(i, j + 1)) *
                                 (pn(i, j) + pn(i, j + 1)) * (oHz(i, j, k + 1) + oHz(i, j + 1, k + 1)));
          Wm(i, k) = W(i, j, k) * odz(i, j, k) * pm(i, j) * pn(i, j) * dt;
          const double Wk = Wm(i, k);
          const double X = (fabs(Um) - Um * Um) * A - B * Um * Vm - Ck * Um * Wk;
          const double Y = (fabs(Vm) - Vm * Vm) * B - A * Um * Vm - Ck * Vm * Wk;
          const double Z = (fabs(Wk) - Wk * Wk) * Ck - A * Um * Wk - B * Vm * Wk;
          const double AA = A * A, BB = B * B, CC = Ck * Ck, AC = A * Ck, BC = B * Ck;
          const double XX = X * X, YY = Y * Y, ZZ = Z * Z, XZ = X * Z, YZ = Y * Z;
          const double sig_alfa = 1.0 / (1.0 - fabs(Ck) + eps);
          const double sig_beta = -Ck / ((1.0 - fabs(Ck)) * (1.0 - CC) + eps);
          const double sig_gama = 2.0 * fabs(CC * Ck) /
                                  ((1.0 - fabs(Ck)) * (1.0 - CC) * (1.0 - fabs(CC * Ck)) + eps);
          const double sig_a = -B / ((1.0 - fabs(Ck)) * (1.0 - fabs(BC)) + eps);
          const double sig_b = BC / ((1.0 - fabs(Ck)) * (1.0 - CC * fabs(B)) + eps) *
                               (fabs(B) / (1.0 - fabs(BC) + eps) + 2.0 * Ck / (1.0 - CC + eps));
          const double sig_c = fabs(Ck) * BB / ((1.0 - fabs(Ck)) * (1.0 - B * B * fabs(Ck)) * (1.0 - fabs(BC)) + eps);
          const double sig_d = -A / ((1.0 - fabs(Ck)) * (1.0 - fabs(AC)) + eps);
          const double sig_e = AC / ((1.0 - fabs(Ck)) * (1.0 - CC * fabs(A)) + eps) *
                               (fabs(A) / (1.0 - fabs(AC) + eps) + 2.0 * Ck / (1.0 - CC + eps));
          const double sig_f = fabs(Ck) * AA / ((1.0 - fabs(Ck)) * (1.0 - AA * fabs(Ck)) * (1.0 - fabs(AC)) + eps);
          double wa = sig_alfa * Z + sig_beta * ZZ + sig_gama * ZZ * Z + sig_a * YZ + sig_b * ZZ * Y +
                      sig_c * Z * YY + sig_d * XZ + sig_e * ZZ * X + sig_f * Z * XX;
          Wa(i, j, k) = MIN(fabs(wa), fac * fabs(Wk)) * copysign(1.0, wa);
          if (mk) Wa(i, j, k) = Wa(i, j, k) * rmask(i, j);                    /* :801 */
          if (p->wet_dry) Wa(i, j, k) = Wa(i, j, k) * rmask_wet(i, j);           /* WET_DRY: the next block */
        }
      }
    for (int i = IstrU - 1; i <= Iendp1; i++) {
      Wa(i, j, 0) = 0.0;
      Wa(i, j, N) = 0.0;
    }
  }

  /* ---- flux-corrected-transport limiter, :842-1030 (mask_up = mask_dn = 1 without MASKING) ---- */
  for (int j = JstrV - 1; j <= Jendp1; j++) {
    for (int k = 1; k <= N; k++)
      for (int i = IstrU - 1; i <= Iendp1; i++) {
        /* extrema over the point, its four horizontal neighbours and the level(s) above / below, of Ta and of
         * t(:,:,:,3) -- :842-925; the list has 12 terms at k = 1 and k = N, 14 in between; the order of a MAX / MIN
         * list does not matter.  MASKING: every term times mask_up (land values out of Tmax) / mask_dn (huge on
         * land, out of Tmin), :826-835; both are exactly 1 without the option */
        double Tmax, Tmin;
        {
          const int pi[5] = {i - 1, i, i + 1, i, i}, pj[5] = {j, j, j, j - 1, j + 1};
          Tmax = Ta(pi[0], pj[0], k) * MUP(pi[0], pj[0]);
          Tmin = Ta(pi[0], pj[0], k) * MDN(pi[0], pj[0]);
          for (int q = 0; q < 5; q++) {
            const double mu = MUP(pi[q], pj[q]), md = MDN(pi[q], pj[q]);
            Tmax = MAX(Tmax, Ta(pi[q], pj[q], k) * mu); Tmax = MAX(Tmax, t3(pi[q], pj[q], k) * mu);
            Tmin = MIN(Tmin, Ta(pi[q], pj[q], k) * md); Tmin = MIN(Tmin, t3(pi[q], pj[q], k) * md);
          }
          const double mu = MUP(i, j), md = MDN(i, j);
          if (k > 1) {
            Tmax = MAX(Tmax, Ta(i, j, k - 1) * mu); Tmax = MAX(Tmax, t3(i, j, k - 1) * mu);
            Tmin = MIN(Tmin, Ta(i, j, k - 1) * md); Tmin = MIN(Tmin, t3(i, j, k - 1) * md);
          }
          if (k < N) {
            Tmax = MAX(Tmax, Ta(i, j, k + 1) * mu); Tmax = MAX(Tmax, t3(i, j, k + 1) * mu);
            Tmin = MIN(Tmin, Ta(i, j, k + 1) * md); Tmin = MIN(Tmin, t3(i, j, k + 1) * md);
          }
        }
        double cff1, cff2;
        if (k == 1) {
          cff1 = Ta(i - 1, j, k) * MAX(0.0, Ua(i, j, k)) - Ta(i + 1, j, k) * MIN(0.0, Ua(i + 1, j, k)) +
                 Ta(i, j - 1, k) * MAX(0.0, Va(i, j, k)) - Ta(i, j + 1, k) * MIN(0.0, Va(i, j + 1, k)) -
                 Ta(i, j, k + 1) * MIN(0.0, Wa(i, j, k));
          cff2 = Ta(i, j, k) * MAX(0.0, Ua(i + 1, j, k)) - Ta(i, j, k) * MIN(0.0, Ua(i, j, k)) +
                 Ta(i, j, k) * MAX(0.0, Va(i, j + 1, k)) - Ta(i, j, k) * MIN(0.0, Va(i, j, k)) +
                 Ta(i, j, k) * MAX(0.0, Wa(i, j, k));
        } else if (k < N) {
          cff1 = Ta(i - 1, j, k) * MAX(0.0, Ua(i, j, k)) - Ta(i + 1, j, k) * MIN(0.0, Ua(i + 1, j, k)) +
                 Ta(i, j - 1, k) * MAX(0.0, Va(i, j, k)) - Ta(i, j + 1, k) * MIN(0.0, Va(i, j + 1, k)) +
                 Ta(i, j, k - 1) * MAX(0.0, Wa(i, j, k - 1)) - Ta(i, j, k + 1) * MIN(0.0, Wa(i, j, k));
          cff2 = Ta(i, j, k) * MAX(0.0, Ua(i + 1, j, k)) - Ta(i, j, k) * MIN(0.0, Ua(i, j, k)) +
                 Ta(i, j, k) * MAX(0.0, Va(i, j + 1, k)) - Ta(i, j, k) * MIN(0.0, Va(i, j, k)) +
                 Ta(i, j, k) * MAX(0.0, Wa(i, j, k)) - Ta(i, j, k) * MIN(0.0, Wa(i, j, k - 1));
        } else {
          cff1 = Ta(i - 1, j, k) * MAX(0.0, Ua(i, j, k)) - Ta(i + 1, j, k) * MIN(0.0, Ua(i + 1, j, k)) +
                 Ta(i, j - 1, k) * MAX(0.0, Va(i, j, k)) - Ta(i, j + 1, k) * MIN(0.0, Va(i, j + 1, k)) +
                 Ta(i, j, k - 1) * MAX(0.0, Wa(i, j, k - 1));
          cff2 = Ta(i, j, k) * MAX(0.0, Ua(i + 1, j, k)) - Ta(i, j, k) * MIN(0.0, Ua(i, j, k)) +
                 Ta(i, j, k) * MAX(0.0, Va(i, j + 1, k)) - Ta(i, j, k) * MIN(0.0, Va(i, j, k)) -
                 Ta(i, j, k) * MIN(0.0, Wa(i, j, k - 1));
        }
        beta_up(i, j, k) = (Tmax - Ta(i, j, k)) / (cff1 + eps);
        beta_dn(i, j, k) = (Ta(i, j, k) - Tmin) / (cff2 + eps);
      }
  }

  /* ---- limited anti-diffusive transports, :1032-1066 ---- */
  for (int k = 1; k <= N; k++) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iendp1; i++) {
        const double cff1 = MIN(MIN(beta_dn(i - 1, j, k), beta_up(i, j, k)), 1.0);
        const double cff2 = MIN(MIN(beta_up(i - 1, j, k), beta_dn(i, j, k)), 1.0);
        Ua(i, j, k) = (cff1 * MAX(0.0, Ua(i, j, k)) + cff2 * MIN(0.0, Ua(i, j, k))) * cff * om_u(i, j);
        if (mk) Ua(i, j, k) = Ua(i, j, k) * umask(i, j);                      /* :991 */
        if (p->wet_dry) Ua(i, j, k) = Ua(i, j, k) * umask_wet(i, j);           /* WET_DRY: the next block */
      }
    for (int j = JstrV; j <= Jendp1; j++)
      for (int i = Istr; i <= Iend; i++) {
        const double cff1 = MIN(MIN(beta_dn(i, j - 1, k), beta_up(i, j, k)), 1.0);
        const double cff2 = MIN(MIN(beta_up(i, j - 1, k), beta_dn(i, j, k)), 1.0);
        Va(i, j, k) = (cff1 * MAX(0.0, Va(i, j, k)) + cff2 * MIN(0.0, Va(i, j, k))) * cff * on_v(i, j);
        if (mk) Va(i, j, k) = Va(i, j, k) * vmask(i, j);                      /* :1006 */
        if (p->wet_dry) Va(i, j, k) = Va(i, j, k) * vmask_wet(i, j);           /* WET_DRY: the next block */
      }
    if (k < N)
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          const double cff1 = MIN(MIN(beta_dn(i, j, k), beta_up(i, j, k + 1)), 1.0);
          const double cff2 = MIN(MIN(beta_up(i, j, k), beta_dn(i, j, k + 1)), 1.0);
          Wa(i, j, k) = (cff1 * MAX(0.0, Wa(i, j, k)) + cff2 * MIN(0.0, Wa(i, j, k))) * cff * omn(i, j) *
                        (z_r(i, j, k + 1) - z_r(i, j, k));
          if (mk) Wa(i, j, k) = Wa(i, j, k) * rmask(i, j);                    /* :1022 */
          if (p->wet_dry) Wa(i, j, k) = Wa(i, j, k) * rmask_wet(i, j);           /* WET_DRY: the next block */
        }
  }

  /* walls again, :1068-1100 */
  if (!EWperiodic) {
    if (west_edge)
      for (int k = 1; k <= N; k++)
        for (int j = Jstr; j <= Jend; j++)
          Ua(Istr, j, k) = (o_lbc(p, LBS_WEST, LBV_U) == LBC_CLOSED) ? 0.0 : Ua(Istr + 1, j, k);
    if (east_edge)
      for (int k = 1; k <= N; k++)
        for (int j = Jstr; j <= Jend; j++)
          Ua(Iend + 1, j, k) = (o_lbc(p, LBS_EAST, LBV_U) == LBC_CLOSED) ? 0.0 : Ua(Iend, j, k);
  }
  if (!NSperiodic) {
    if (south_edge)
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++)
          Va(i, Jstr, k) = (o_lbc(p, LBS_SOUTH, LBV_V) == LBC_CLOSED) ? 0.0 : Va(i, Jstr + 1, k);
    if (north_edge)
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++)
          Va(i, Jend + 1, k) = (o_lbc(p, LBS_NORTH, LBV_V) == LBC_CLOSED) ? 0.0 : Va(i, Jend, k);
  }
  free(C_); free(Wm_); free(beta_dn_); free(beta_up_); free(odz_);
  return 0;
#undef Ta
#undef Ua
#undef Va
#undef Wa
#undef oHz
#undef t3
#undef odz
#undef beta_dn
#undef beta_up
#undef C
#undef Wm
}
