/*
 * oracle_gls.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Generic length-scale turbulence closure (GLS_MIXING):
 *   gls_prestep_tile  ROMS/Nonlinear/gls_prestep.F:66-420   predictor (half-step) advection of tke and gls
 *   gls_corstep_tile  ROMS/Nonlinear/gls_corstep.F:101-1218 corrector: advection, production, dissipation,
 *                     implicit vertical diffusion, length-scale limitation, stability functions, Akv / Akt / Akk / Akp
 *   tkebc_tile        ROMS/Nonlinear/tkebc_im.F:50-698       closed and gradient edges, corners
 * Options restated: the default third-order upstream advection (neither K_C2ADVECTION nor K_C4ADVECTION), RI_SPLINES
 * and N2S2_HORAVG as run-time switches, the four stability-function choices (GALPERIN = none of the CPP options,
 * KANTHA_CLAYSON, CANUTO_A, CANUTO_B), MASKING.  Not restated: CRAIG_BANNER, CHARNOK, ZOS_HSIG, TKE_WAVEDISS,
 * LIMIT_VDIFF / LIMIT_VVISC, radiation edges.  Both reference files compile stand-alone (no mod_sources): pinned
 * against oracle/_ref/<APP>_GLS* (tests/test_ref_pinning.py).
 */
#include "oracle.h"

#define tke(i,j,k,n) F->tke[I3W(i,j,k) + (long)((n)-1) * n3w]
#define gls(i,j,k,n) F->gls[I3W(i,j,k) + (long)((n)-1) * n3w]
#define Lscale(i,j,k) F->Lscale[I3W(i,j,k)]
#define Akk(i,j,k)   F->Akk[I3W(i,j,k)]
#define Akp(i,j,k)   F->Akp[I3W(i,j,k)]
#define ZoBot(i,j)   F->ZoBot[I2(i,j)]

/* tkebc_tile for the conditions restated here: a closed and a gradient edge both copy the first inside value
 * (tkebc_im.F:181-214 and the other three edges), times rmask of the boundary point under MASKING; then the corners
 * (:644-695).  LBC(:,isMtke,ng) follows the tracers' table here (every variable without an entry of its own does in
 * this library). */
static void o_tkebc(const roms_bounds_t *b, const roms_params_t *p, const roms_fields_t *F, int nout)
{
  ORACLE_PROLOGUE
  const int mk = p->masking;
  if (west_edge && !EWperiodic)
    for (int k = 0; k <= N; k++)
      for (int j = Jstr; j <= Jend; j++) {
        tke(Istr - 1, j, k, nout) = tke(Istr, j, k, nout);
        if (mk) tke(Istr - 1, j, k, nout) = tke(Istr - 1, j, k, nout) * rmask(Istr - 1, j);
        gls(Istr - 1, j, k, nout) = gls(Istr, j, k, nout);
        if (mk) gls(Istr - 1, j, k, nout) = gls(Istr - 1, j, k, nout) * rmask(Istr - 1, j);
      }
  if (east_edge && !EWperiodic)
    for (int k = 0; k <= N; k++)
      for (int j = Jstr; j <= Jend; j++) {
        tke(Iend + 1, j, k, nout) = tke(Iend, j, k, nout);
        if (mk) tke(Iend + 1, j, k, nout) = tke(Iend + 1, j, k, nout) * rmask(Iend + 1, j);
        gls(Iend + 1, j, k, nout) = gls(Iend, j, k, nout);
        if (mk) gls(Iend + 1, j, k, nout) = gls(Iend + 1, j, k, nout) * rmask(Iend + 1, j);
      }
  if (south_edge && !NSperiodic)
    for (int k = 0; k <= N; k++)
      for (int i = Istr; i <= Iend; i++) {
        tke(i, Jstr - 1, k, nout) = tke(i, Jstr, k, nout);
        if (mk) tke(i, Jstr - 1, k, nout) = tke(i, Jstr - 1, k, nout) * rmask(i, Jstr - 1);
        gls(i, Jstr - 1, k, nout) = gls(i, Jstr, k, nout);
        if (mk) gls(i, Jstr - 1, k, nout) = gls(i, Jstr - 1, k, nout) * rmask(i, Jstr - 1);
      }
  if (north_edge && !NSperiodic)
    for (int k = 0; k <= N; k++)
      for (int i = Istr; i <= Iend; i++) {
        tke(i, Jend + 1, k, nout) = tke(i, Jend, k, nout);
        if (mk) tke(i, Jend + 1, k, nout) = tke(i, Jend + 1, k, nout) * rmask(i, Jend + 1);
        gls(i, Jend + 1, k, nout) = gls(i, Jend, k, nout);
        if (mk) gls(i, Jend + 1, k, nout) = gls(i, Jend + 1, k, nout) * rmask(i, Jend + 1);
      }
  if (!(EWperiodic || NSperiodic)) {
    for (int k = 0; k <= N; k++) {
      if (south_edge && west_edge) {
        tke(Istr - 1, Jstr - 1, k, nout) = 0.5 * (tke(Istr, Jstr - 1, k, nout) + tke(Istr - 1, Jstr, k, nout));
        gls(Istr - 1, Jstr - 1, k, nout) = 0.5 * (gls(Istr, Jstr - 1, k, nout) + gls(Istr - 1, Jstr, k, nout));
      }
      if (south_edge && east_edge) {
        tke(Iend + 1, Jstr - 1, k, nout) = 0.5 * (tke(Iend, Jstr - 1, k, nout) + tke(Iend + 1, Jstr, k, nout));
        gls(Iend + 1, Jstr - 1, k, nout) = 0.5 * (gls(Iend, Jstr - 1, k, nout) + gls(Iend + 1, Jstr, k, nout));
      }
      if (north_edge && west_edge) {
        tke(Istr - 1, Jend + 1, k, nout) = 0.5 * (tke(Istr, Jend + 1, k, nout) + tke(Istr - 1, Jend, k, nout));
        gls(Istr - 1, Jend + 1, k, nout) = 0.5 * (gls(Istr, Jend + 1, k, nout) + gls(Istr - 1, Jend, k, nout));
      }
      if (north_edge && east_edge) {
        tke(Iend + 1, Jend + 1, k, nout) = 0.5 * (tke(Iend, Jend + 1, k, nout) + tke(Iend + 1, Jend, k, nout));
        gls(Iend + 1, Jend + 1, k, nout) = 0.5 * (gls(Iend, Jend + 1, k, nout) + gls(Iend + 1, Jend, k, nout));
      }
    }
  }
}

int oracle_gls_check(const roms_bounds_t *b, const roms_params_t *p)
{
  if (!p->gls_mixing) return 8;
  /* the conditions restated: periodic, closed, gradient (the tracers' table stands in for LBC(:,isMtke,ng)) */
  for (int sd = 0; sd < 4; sd++) {
    const int c = o_lbc(p, sd, LBV_T);
    if (c != LBC_PERIODIC && c != LBC_CLOSED && c != LBC_GRADIENT) return 8;
  }
  (void)b;
  return 0;
}

/* ---------------------------------------------------------------- gls_prestep -- */
int oracle_gls_prestep(OARGS)
{
  ORACLE_PROLOGUE
  if (oracle_gls_check(b, p)) return 8;
  const int nstp = s->nstp, nnew = s->nnew;
  const double dt = p->dt;
  const double Gamma = 1.0 / 6.0;
  double cff, cff1, cff2, cff3, cff4;
  int indx;
  const long n2 = nis * njs;
  double *CF_ = walloc(nis * (N + 1)), *FC_ = walloc(nis * (N + 1)), *FCL_ = walloc(nis * (N + 1));
  double *Hz_half_ = walloc(n2 * N);
  double *EF_ = walloc(n2), *FE_ = walloc(n2), *FEL_ = walloc(n2), *FX_ = walloc(n2), *FXL_ = walloc(n2), *XF_ = walloc(n2);
  double *grad_ = walloc(n2), *gradL_ = walloc(n2);
#define CF(i,k) CF_[WSK(i,k)]
#define FC(i,k) FC_[WSK(i,k)]
#define FCL(i,k) FCL_[WSK(i,k)]
#define Hz_half(i,j,k) Hz_half_[WS3(i,j,k)]
#define EF(i,j) EF_[WS2(i,j)]
#define FE(i,j) FE_[WS2(i,j)]
#define FEL(i,j) FEL_[WS2(i,j)]
#define FX(i,j) FX_[WS2(i,j)]
#define FXL(i,j) FXL_[WS2(i,j)]
#define XF(i,j) XF_[WS2(i,j)]
#define grad(i,j) grad_[WS2(i,j)]
#define gradL(i,j) gradL_[WS2(i,j)]
  for (int k = 1; k <= N - 1; k++) {
    /* fourth-order, centered differences advection (gls_prestep.F:176-263) */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istrm1; i <= Iendp2; i++) {
        grad(i, j) = (tke(i, j, k, nstp) - tke(i - 1, j, k, nstp));
        if (p->masking) grad(i, j) = grad(i, j) * umask(i, j);
        gradL(i, j) = (gls(i, j, k, nstp) - gls(i - 1, j, k, nstp));
        if (p->masking) gradL(i, j) = gradL(i, j) * umask(i, j);
      }
    if (!EWperiodic) {
      if (west_edge)
        for (int j = Jstr; j <= Jend; j++) { grad(Istr - 1, j) = grad(Istr, j); gradL(Istr - 1, j) = gradL(Istr, j); }
      if (east_edge)
        for (int j = Jstr; j <= Jend; j++) { grad(Iend + 2, j) = grad(Iend + 1, j); gradL(Iend + 2, j) = gradL(Iend + 1, j); }
    }
    cff = 1.0 / 6.0;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        XF(i, j) = 0.5 * (Huon(i, j, k) + Huon(i, j, k + 1));
        FX(i, j) = XF(i, j) * 0.5 * (tke(i - 1, j, k, nstp) + tke(i, j, k, nstp) - cff * (grad(i + 1, j) - grad(i - 1, j)));
        FXL(i, j) = XF(i, j) * 0.5 * (gls(i - 1, j, k, nstp) + gls(i, j, k, nstp) - cff * (gradL(i + 1, j) - gradL(i - 1, j)));
      }
    for (int j = Jstrm1; j <= Jendp2; j++)
      for (int i = Istr; i <= Iend; i++) {
        grad(i, j) = (tke(i, j, k, nstp) - tke(i, j - 1, k, nstp));
        if (p->masking) grad(i, j) = grad(i, j) * vmask(i, j);
        gradL(i, j) = (gls(i, j, k, nstp) - gls(i, j - 1, k, nstp));
        if (p->masking) gradL(i, j) = gradL(i, j) * vmask(i, j);
      }
    if (!NSperiodic) {
      if (south_edge)
        for (int i = Istr; i <= Iend; i++) { grad(i, Jstr - 1) = grad(i, Jstr); gradL(i, Jstr - 1) = gradL(i, Jstr); }
      if (north_edge)
        for (int i = Istr; i <= Iend; i++) { grad(i, Jend + 2) = grad(i, Jend + 1); gradL(i, Jend + 2) = gradL(i, Jend + 1); }
    }
    cff = 1.0 / 6.0;
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++) {
        EF(i, j) = 0.5 * (Hvom(i, j, k) + Hvom(i, j, k + 1));
        FE(i, j) = EF(i, j) * 0.5 * (tke(i, j - 1, k, nstp) + tke(i, j, k, nstp) - cff * (grad(i, j + 1) - grad(i, j - 1)));
        FEL(i, j) = EF(i, j) * 0.5 * (gls(i, j - 1, k, nstp) + gls(i, j, k, nstp) - cff * (gradL(i, j + 1) - gradL(i, j - 1)));
      }
    /* time-step horizontal advection (:267-296) */
    if (s->iic == s->ntfirst) { cff1 = 1.0; cff2 = 0.0; cff3 = 0.5 * dt; indx = nstp; }
    else { cff1 = 0.5 + Gamma; cff2 = 0.5 - Gamma; cff3 = (1.0 - Gamma) * dt; indx = 3 - nstp; }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = 0.5 * (Hz(i, j, k) + Hz(i, j, k + 1));
        cff4 = cff3 * pm(i, j) * pn(i, j);
        Hz_half(i, j, k) = cff - cff4 * (XF(i + 1, j) - XF(i, j) + EF(i, j + 1) - EF(i, j));
        tke(i, j, k, 3) = cff * (cff1 * tke(i, j, k, nstp) + cff2 * tke(i, j, k, indx)) -
                          cff4 * (FX(i + 1, j) - FX(i, j) + FE(i, j + 1) - FE(i, j));
        gls(i, j, k, 3) = cff * (cff1 * gls(i, j, k, nstp) + cff2 * gls(i, j, k, indx)) -
                          cff4 * (FXL(i + 1, j) - FXL(i, j) + FEL(i, j + 1) - FEL(i, j));
        tke(i, j, k, nnew) = cff * tke(i, j, k, nstp);
        gls(i, j, k, nnew) = cff * gls(i, j, k, nstp);
      }
  }
  /* vertical advection (:300-375) */
  for (int j = Jstr; j <= Jend; j++) {
    cff1 = 7.0 / 12.0;
    cff2 = 1.0 / 12.0;
    for (int k = 2; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) {
        CF(i, k) = 0.5 * (W(i, j, k) + W(i, j, k - 1));
        FC(i, k) = CF(i, k) * (cff1 * (tke(i, j, k - 1, nstp) + tke(i, j, k, nstp)) -
                               cff2 * (tke(i, j, k - 2, nstp) + tke(i, j, k + 1, nstp)));
        FCL(i, k) = CF(i, k) * (cff1 * (gls(i, j, k - 1, nstp) + gls(i, j, k, nstp)) -
                                cff2 * (gls(i, j, k - 2, nstp) + gls(i, j, k + 1, nstp)));
      }
    cff1 = 1.0 / 3.0;
    cff2 = 5.0 / 6.0;
    cff3 = 1.0 / 6.0;
    for (int i = Istr; i <= Iend; i++) {
      CF(i, 1) = 0.5 * (W(i, j, 0) + W(i, j, 1));
      FC(i, 1) = CF(i, 1) * (cff1 * tke(i, j, 0, nstp) + cff2 * tke(i, j, 1, nstp) - cff3 * tke(i, j, 2, nstp));
      FCL(i, 1) = CF(i, 1) * (cff1 * gls(i, j, 0, nstp) + cff2 * gls(i, j, 1, nstp) - cff3 * gls(i, j, 2, nstp));
      CF(i, N) = 0.5 * (W(i, j, N) + W(i, j, N - 1));
      FC(i, N) = CF(i, N) * (cff1 * tke(i, j, N, nstp) + cff2 * tke(i, j, N - 1, nstp) - cff3 * tke(i, j, N - 2, nstp));
      FCL(i, N) = CF(i, N) * (cff1 * gls(i, j, N, nstp) + cff2 * gls(i, j, N - 1, nstp) - cff3 * gls(i, j, N - 2, nstp));
    }
    if (s->iic == s->ntfirst) cff3 = 0.5 * dt;
    else cff3 = (1.0 - Gamma) * dt;
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) {
        cff4 = cff3 * pm(i, j) * pn(i, j);
        Hz_half(i, j, k) = Hz_half(i, j, k) - cff4 * (CF(i, k + 1) - CF(i, k));
        cff1 = 1.0 / Hz_half(i, j, k);
        tke(i, j, k, 3) = cff1 * (tke(i, j, k, 3) - cff4 * (FC(i, k + 1) - FC(i, k)));
        gls(i, j, k, 3) = cff1 * (gls(i, j, k, 3) - cff4 * (FCL(i, k + 1) - FCL(i, k)));
      }
  }
  o_tkebc(b, p, F, 3);
  o_exchange3d(b, GT_R, N + 1, &tke(LBi, LBj, 0, 3));
  o_exchange3d(b, GT_R, N + 1, &gls(LBi, LBj, 0, 3));
  free(CF_); free(FC_); free(FCL_); free(Hz_half_); free(EF_); free(FE_); free(FEL_); free(FX_); free(FXL_); free(XF_);
  free(grad_); free(gradL_);
  return 0;
#undef CF
#undef FC
#undef FCL
#undef Hz_half
#undef EF
#undef FE
#undef FEL
#undef FX
#undef FXL
#undef XF
#undef grad
#undef gradL
}

/* stability-function constants as initialize_scalars sets them (mod_scalars.F:1686-1712, :1756-1768, :4450-4490) */
typedef struct {
  double Gh0, Ghcri, Ghmin, E2;
  double s0, s1, s2, s4, s5, s6, b0, b1, b2, b3, b4, b5;     /* Canuto */
  double my_Sh1, my_Sh2, my_Sm2, my_Sm3, my_Sm4, my_B1pm1o3;
} gls_const_t;

static gls_const_t gls_constants(int stab)
{
  gls_const_t c;
  memset(&c, 0, sizeof c);
  c.Ghmin = -0.28;
  c.E2 = 1.33;
  if (stab == GLS_CANUTO_A || stab == GLS_CANUTO_B) {
    double L1, L2, L3, L4, L5, L6, L7, L8;
    if (stab == GLS_CANUTO_A) { c.Gh0 = 0.0329; c.Ghcri = 0.03; L1 = 0.107; L2 = 0.0032; L3 = 0.0864; L4 = 0.12; L5 = 11.9; L6 = 0.4; L7 = 0.0; L8 = 0.48; }
    else { c.Gh0 = 0.0444; c.Ghcri = 0.0414; L1 = 0.127; L2 = 0.00336; L3 = 0.0906; L4 = 0.101; L5 = 11.2; L6 = 0.4; L7 = 0.0; L8 = 0.318; }
    c.s0 = 3.0 / 2.0 * L1 * (L5 * L5);
    c.s1 = -L4 * (L6 + L7) + 2.0 * L4 * L5 * (L1 - 1.0 / 3.0 * L2 - L3) + 3.0 / 2.0 * L1 * L5 * L8;
    c.s2 = -3.0 / 8.0 * L1 * ((L6 * L6) - (L7 * L7));
    c.s4 = 2.0 * L5;
    c.s5 = 2.0 * L4;
    c.s6 = 2.0 / 3.0 * L5 * (3.0 * (L3 * L3) - (L2 * L2)) - 1.0 / 2.0 * L5 * L1 * (3.0 * L3 - L2) + 3.0 / 4.0 * L1 * (L6 - L7);
    c.b0 = 3.0 * (L5 * L5);
    c.b1 = L5 * (7.0 * L4 + 3.0 * L8);
    c.b2 = (L5 * L5) * (3.0 * (L3 * L3) - (L2 * L2)) - 3.0 / 4.0 * ((L6 * L6) - (L7 * L7));
    c.b3 = L4 * (4.0 * L4 + 3.0 * L8);
    c.b5 = 1.0 / 4.0 * ((L2 * L2) - 3.0 * (L3 * L3)) * ((L6 * L6) - (L7 * L7));
    c.b4 = L4 * (L2 * L6 - 3.0 * L3 * L7 - L5 * ((L2 * L2) - (L3 * L3))) + L5 * L8 * (3.0 * (L3 * L3) - (L2 * L2));
  } else {
    c.Gh0 = 0.028;
    c.Ghcri = 0.02;
  }
  const double A1 = 0.92, A2 = 0.74, B1 = 16.6, B2 = 10.1, C1 = 0.08, C2 = 0.7, C3 = 0.2;
  c.my_B1pm1o3 = 1.0 / pow(B1, 1.0 / 3.0);
  c.my_Sm2 = 9.0 * A1 * A2;
  c.my_Sh1 = A2 * (1.0 - 6.0 * A1 / B1);
  if (stab == GLS_KANTHA_CLAYSON) {
    c.my_Sh2 = 3.0 * A2 * (6.0 * A1 + B2 * (1.0 - C3));
    c.my_Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2 * (1.0 - C2);
  } else {
    c.my_Sh2 = 3.0 * A2 * (6.0 * A1 + B2);
    c.my_Sm3 = A1 * (1.0 - 3.0 * C1 - 6.0 * A1 / B1);
    c.my_Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2;
  }
  return c;
}

/* ---------------------------------------------------------------- gls_corstep -- */
int oracle_gls_corstep(OARGS)
{
  ORACLE_PROLOGUE
  if (oracle_gls_check(b, p)) return 8;
  const int nstp = s->nstp, nnew = s->nnew;
  const double dt = p->dt, g = p->g;
  const double vonKar = 0.41;
  const double Gadv = 1.0 / 3.0, eps = 1.0E-10;
  const gls_const_t K = gls_constants(p->gls_stability);
  const int my25 = (p->gls_mixing == 2);      /* MY25_MIXING: my25_corstep.F (the same routine up to the vertical terms) */
  const double gls_p = p->gls_p, gls_m = p->gls_m, gls_n = p->gls_n, gls_cmu0 = p->gls_cmu0;
  const double gls_c1 = p->gls_c1, gls_c2 = p->gls_c2, gls_c3m = p->gls_c3m, gls_c3p = p->gls_c3p;
  const double gls_sigk = p->gls_sigk, gls_sigp = p->gls_sigp, gls_Kmin = p->gls_Kmin, gls_Pmin = p->gls_Pmin;
  const double Akv_bak = p->Akv_bak, Akk_bak = p->Akk_bak, Akp_bak = p->Akp_bak;
  const int itemp = 1;
  double cff, cff1, cff2, cff3;
  (void)g; (void)cff3;
  const long n2 = nis * njs;
  double *tke_fluxt = walloc(nis), *tke_fluxb = walloc(nis), *gls_fluxt = walloc(nis), *gls_fluxb = walloc(nis), *Zos_eff = walloc(nis);
  double *BCK_ = walloc(nis * (N + 1)), *BCP_ = walloc(nis * (N + 1)), *CF_ = walloc(nis * (N + 1));
  double *FCK_ = walloc(nis * (N + 1)), *FCP_ = walloc(nis * (N + 1)), *dU_ = walloc(nis * (N + 1)), *dV_ = walloc(nis * (N + 1));
  double *shear2_ = walloc(n2 * (N + 1)), *buoy2_ = walloc(n2 * (N + 1));
  double *FEK_ = walloc(n2), *FEP_ = walloc(n2), *FXK_ = walloc(n2), *FXP_ = walloc(n2), *Zob_min_ = walloc(n2);
  double *curvK_ = walloc(n2), *curvP_ = walloc(n2), *gradK_ = walloc(n2), *gradP_ = walloc(n2);
#define BCK(i,k) BCK_[WSK(i,k)]
#define BCP(i,k) BCP_[WSK(i,k)]
#define CF(i,k) CF_[WSK(i,k)]
#define FCK(i,k) FCK_[WSK(i,k)]
#define FCP(i,k) FCP_[WSK(i,k)]
#define dU(i,k) dU_[WSK(i,k)]
#define dV(i,k) dV_[WSK(i,k)]
#define shear2(i,j,k) shear2_[WS2(i,j) + (long)(k) * n2]
#define buoy2(i,j,k) buoy2_[WS2(i,j) + (long)(k) * n2]
#define FEK(i,j) FEK_[WS2(i,j)]
#define FEP(i,j) FEP_[WS2(i,j)]
#define FXK(i,j) FXK_[WS2(i,j)]
#define FXP(i,j) FXP_[WS2(i,j)]
#define Zob_min(i,j) Zob_min_[WS2(i,j)]
#define curvK(i,j) curvK_[WS2(i,j)]
#define curvP(i,j) curvP_[WS2(i,j)]
#define gradK(i,j) gradK_[WS2(i,j)]
#define gradP(i,j) gradP_[WS2(i,j)]
  /* constants (gls_corstep.F:262-312) */
  const double Zos_min = MAX(p->Zos, 0.0001);
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) Zob_min(i, j) = MAX(ZoBot(i, j), 0.0001);
  const int Lmy25 = (gls_p == 0.0) && (gls_n == 1.0) && (gls_m == 1.0);
  const double L_sft = vonKar;
  const double gls_sigp_cb = gls_sigp;
  const double ogls_sigp = 1.0 / gls_sigp_cb;
  const double sqrt2 = sqrt(2.0);
  const double cmu_fac1 = pow(gls_cmu0, -gls_p / gls_n);
  const double cmu_fac2 = pow(gls_cmu0, 3.0 + gls_p / gls_n);
  const double cmu_fac3 = 1.0 / pow(gls_cmu0, 2.0);
  const double gls_fac2 = pow(gls_cmu0, gls_p) * gls_n * pow(vonKar, gls_n);
  const double gls_fac3 = pow(gls_cmu0, gls_p) * gls_n;
  const double gls_fac4 = pow(gls_cmu0, gls_p);
  const double gls_fac5 = pow(0.56, 0.5 * gls_n) * pow(gls_cmu0, gls_p);
  const double gls_fac6 = 8.0 / pow(gls_cmu0, 6.0);
  const double gls_exp1 = 1.0 / gls_n;
  const double tke_exp1 = gls_m / gls_n;
  const double tke_exp2 = 0.5 + gls_m / gls_n;
  const double tke_exp4 = gls_m + 0.5 * gls_n;
  /* vertical shear at W-points (:316-372) */
  if (p->gls_ri_splines) {
    for (int j = Jstrm1; j <= Jendp1; j++) {
      for (int i = Istrm1; i <= Iendp1; i++) { CF(i, 0) = 0.0; dU(i, 0) = 0.0; dV(i, 0) = 0.0; }
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istrm1; i <= Iendp1; i++) {
          cff = 1.0 / (2.0 * Hz(i, j, k + 1) + Hz(i, j, k) * (2.0 - CF(i, k - 1)));
          CF(i, k) = cff * Hz(i, j, k + 1);
          dU(i, k) = cff * (3.0 * (u(i, j, k + 1, nstp) - u(i, j, k, nstp) + u(i + 1, j, k + 1, nstp) - u(i + 1, j, k, nstp)) -
                            Hz(i, j, k) * dU(i, k - 1));
          dV(i, k) = cff * (3.0 * (v(i, j, k + 1, nstp) - v(i, j, k, nstp) + v(i, j + 1, k + 1, nstp) - v(i, j + 1, k, nstp)) -
                            Hz(i, j, k) * dV(i, k - 1));
        }
      for (int i = Istrm1; i <= Iendp1; i++) { dU(i, N) = 0.0; dV(i, N) = 0.0; }
      for (int k = N - 1; k >= 1; k--)
        for (int i = Istrm1; i <= Iendp1; i++) {
          dU(i, k) = dU(i, k) - CF(i, k) * dU(i, k + 1);
          dV(i, k) = dV(i, k) - CF(i, k) * dV(i, k + 1);
        }
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istrm1; i <= Iendp1; i++) shear2(i, j, k) = dU(i, k) * dU(i, k) + dV(i, k) * dV(i, k);
    }
  } else {
    for (int k = 1; k <= N - 1; k++)
      for (int j = Jstrm1; j <= Jendp1; j++)
        for (int i = Istrm1; i <= Iendp1; i++) {
          cff = 0.5 / (z_r(i, j, k + 1) - z_r(i, j, k));
          const double a1 = cff * (u(i, j, k + 1, nstp) - u(i, j, k, nstp) + u(i + 1, j, k + 1, nstp) - u(i + 1, j, k, nstp));
          const double a2 = cff * (v(i, j, k + 1, nstp) - v(i, j, k, nstp) + v(i, j + 1, k + 1, nstp) - v(i, j + 1, k, nstp));
          shear2(i, j, k) = a1 * a1 + a2 * a2;
        }
  }
  for (int k = 1; k <= N - 1; k++)
    for (int j = Jstr - 1; j <= Jend + 1; j++)
      for (int i = Istr - 1; i <= Iend + 1; i++) buoy2(i, j, k) = bvf(i, j, k);
  if (p->gls_n2s2_horavg) {
    /* N2S2_HORAVG (:384-440): level 0 of the two arrays is the scratch plane */
    for (int k = 1; k <= N - 1; k++) {
      if (west_edge) for (int j = MAX(1, Jstr - 1); j <= MIN(Jend + 1, Mm); j++) shear2(Istr - 1, j, k) = shear2(Istr, j, k);
      if (east_edge) for (int j = MAX(1, Jstr - 1); j <= MIN(Jend + 1, Mm); j++) shear2(Iend + 1, j, k) = shear2(Iend, j, k);
      if (south_edge) for (int i = MAX(1, Istr - 1); i <= MIN(Iend + 1, Lm); i++) shear2(i, Jstr - 1, k) = shear2(i, Jstr, k);
      if (north_edge) for (int i = MAX(1, Istr - 1); i <= MIN(Iend + 1, Lm); i++) shear2(i, Jend + 1, k) = shear2(i, Jend, k);
      if (south_edge && west_edge) shear2(Istr - 1, Jstr - 1, k) = shear2(Istr, Jstr, k);
      if (north_edge && west_edge) shear2(Istr - 1, Jend + 1, k) = shear2(Istr, Jend, k);
      if (south_edge && east_edge) shear2(Iend + 1, Jstr - 1, k) = shear2(Iend, Jstr, k);
      if (north_edge && east_edge) shear2(Iend + 1, Jend + 1, k) = shear2(Iend, Jend, k);
      for (int j = Jstr - 1; j <= Jend; j++)
        for (int i = Istr - 1; i <= Iend; i++) {
          buoy2(i, j, 0) = 0.25 * (buoy2(i, j, k) + buoy2(i + 1, j, k) + buoy2(i, j + 1, k) + buoy2(i + 1, j + 1, k));
          shear2(i, j, 0) = 0.25 * (shear2(i, j, k) + shear2(i + 1, j, k) + shear2(i, j + 1, k) + shear2(i + 1, j + 1, k));
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          buoy2(i, j, k) = 0.25 * (buoy2(i, j, 0) + buoy2(i - 1, j, 0) + buoy2(i, j - 1, 0) + buoy2(i - 1, j - 1, 0));
          shear2(i, j, k) = 0.25 * (shear2(i, j, 0) + shear2(i - 1, j, 0) + shear2(i, j - 1, 0) + shear2(i - 1, j - 1, 0));
        }
    }
  }
  /* time-step advective terms (:444-640): third-order upstream bias */
  for (int k = 1; k <= N - 1; k++) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istrm1; i <= Iendp2; i++) {
        gradK(i, j) = (tke(i, j, k, 3) - tke(i - 1, j, k, 3));
        if (p->masking) gradK(i, j) = gradK(i, j) * umask(i, j);
        gradP(i, j) = (gls(i, j, k, 3) - gls(i - 1, j, k, 3));
        if (p->masking) gradP(i, j) = gradP(i, j) * umask(i, j);
      }
    if (!EWperiodic) {
      if (west_edge)
        for (int j = Jstr; j <= Jend; j++) { gradK(Istr - 1, j) = gradK(Istr, j); gradP(Istr - 1, j) = gradP(Istr, j); }
      if (east_edge)
        for (int j = Jstr; j <= Jend; j++) { gradK(Iend + 2, j) = gradK(Iend + 1, j); gradP(Iend + 2, j) = gradP(Iend + 1, j); }
    }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr - 1; i <= Iend + 1; i++) {
        curvK(i, j) = gradK(i + 1, j) - gradK(i, j);
        curvP(i, j) = gradP(i + 1, j) - gradP(i, j);
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff = 0.5 * (Huon(i, j, k) + Huon(i, j, k + 1));
        if (cff > 0.0) { cff1 = curvK(i - 1, j); cff2 = curvP(i - 1, j); }
        else { cff1 = curvK(i, j); cff2 = curvP(i, j); }
        FXK(i, j) = cff * 0.5 * (tke(i - 1, j, k, 3) + tke(i, j, k, 3) - Gadv * cff1);
        FXP(i, j) = cff * 0.5 * (gls(i - 1, j, k, 3) + gls(i, j, k, 3) - Gadv * cff2);
      }
    for (int j = Jstrm1; j <= Jendp2; j++)
      for (int i = Istr; i <= Iend; i++) {
        gradK(i, j) = (tke(i, j, k, 3) - tke(i, j - 1, k, 3));
        if (p->masking) gradK(i, j) = gradK(i, j) * vmask(i, j);
        gradP(i, j) = (gls(i, j, k, 3) - gls(i, j - 1, k, 3));
        if (p->masking) gradP(i, j) = gradP(i, j) * vmask(i, j);
      }
    if (!NSperiodic) {
      if (south_edge)
        for (int i = Istr; i <= Iend; i++) { gradK(i, Jstr - 1) = gradK(i, Jstr); gradP(i, Jstr - 1) = gradP(i, Jstr); }
      if (north_edge)
        for (int i = Istr; i <= Iend; i++) { gradK(i, Jend + 2) = gradK(i, Jend + 1); gradP(i, Jend + 2) = gradP(i, Jend + 1); }
    }
    for (int j = Jstr - 1; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++) {
        curvK(i, j) = gradK(i, j + 1) - gradK(i, j);
        curvP(i, j) = gradP(i, j + 1) - gradP(i, j);
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = 0.5 * (Hvom(i, j, k) + Hvom(i, j, k + 1));
        if (cff > 0.0) { cff1 = curvK(i, j - 1); cff2 = curvP(i, j - 1); }
        else { cff1 = curvK(i, j); cff2 = curvP(i, j); }
        FEK(i, j) = cff * 0.5 * (tke(i, j - 1, k, 3) + tke(i, j, k, 3) - Gadv * cff1);
        FEP(i, j) = cff * 0.5 * (gls(i, j - 1, k, 3) + gls(i, j, k, 3) - Gadv * cff2);
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = dt * pm(i, j) * pn(i, j);
        tke(i, j, k, nnew) = tke(i, j, k, nnew) - cff * (FXK(i + 1, j) - FXK(i, j) + FEK(i, j + 1) - FEK(i, j));
        if (!my25) tke(i, j, k, nnew) = MAX(tke(i, j, k, nnew), gls_Kmin);       /* my25_corstep.F:511-514 has no floor */
        gls(i, j, k, nnew) = gls(i, j, k, nnew) - cff * (FXP(i + 1, j) - FXP(i, j) + FEP(i, j + 1) - FEP(i, j));
        if (!my25) gls(i, j, k, nnew) = MAX(gls(i, j, k, nnew), gls_Pmin);
      }
  }
  for (int j = Jstr; j <= Jend; j++) {
    /* vertical advection (:644-700) */
    cff1 = 7.0 / 12.0;
    cff2 = 1.0 / 12.0;
    for (int k = 2; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) {
        cff = 0.5 * (W(i, j, k) + W(i, j, k - 1));
        FCK(i, k) = cff * (cff1 * (tke(i, j, k - 1, 3) + tke(i, j, k, 3)) - cff2 * (tke(i, j, k - 2, 3) + tke(i, j, k + 1, 3)));
        FCP(i, k) = cff * (cff1 * (gls(i, j, k - 1, 3) + gls(i, j, k, 3)) - cff2 * (gls(i, j, k - 2, 3) + gls(i, j, k + 1, 3)));
      }
    cff1 = 1.0 / 3.0;
    cff2 = 5.0 / 6.0;
    cff3 = 1.0 / 6.0;
    for (int i = Istr; i <= Iend; i++) {
      cff = 0.5 * (W(i, j, 0) + W(i, j, 1));
      FCK(i, 1) = cff * (cff1 * tke(i, j, 0, 3) + cff2 * tke(i, j, 1, 3) - cff3 * tke(i, j, 2, 3));
      FCP(i, 1) = cff * (cff1 * gls(i, j, 0, 3) + cff2 * gls(i, j, 1, 3) - cff3 * gls(i, j, 2, 3));
      cff = 0.5 * (W(i, j, N) + W(i, j, N - 1));
      FCK(i, N) = cff * (cff1 * tke(i, j, N, 3) + cff2 * tke(i, j, N - 1, 3) - cff3 * tke(i, j, N - 2, 3));
      FCP(i, N) = cff * (cff1 * gls(i, j, N, 3) + cff2 * gls(i, j, N - 1, 3) - cff3 * gls(i, j, N - 2, 3));
    }
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) {
        cff = dt * pm(i, j) * pn(i, j);
        tke(i, j, k, nnew) = tke(i, j, k, nnew) - cff * (FCK(i, k + 1) - FCK(i, k));
        if (!my25) tke(i, j, k, nnew) = MAX(tke(i, j, k, nnew), gls_Kmin);       /* my25_corstep.F:569-576 */
        gls(i, j, k, nnew) = gls(i, j, k, nnew) - cff * (FCP(i, k + 1) - FCP(i, k));
        if (!my25) gls(i, j, k, nnew) = MAX(gls(i, j, k, nnew), gls_Pmin);
      }
    if (my25) {
      /* MY25_MIXING: my25_corstep.F:580-770 (Mellor and Yamada 1982 level 2.5 with the Galperin et al. 1988 stability
       * functions; tke = q2, gls = q2l) */
      const double my_B1 = 16.6, my_E1 = 1.8, my_E2 = 1.33, my_Gh0 = 0.0233, my_Sq = 0.2, my_lmax = 0.53, my_qmin = 1.0E-8;
      const double my_B1p2o3 = pow(my_B1, 2.0 / 3.0);
      cff = -0.5 * dt;
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          FCK(i, k) = cff * (Akk(i, j, k) + Akk(i, j, k - 1)) / Hz(i, j, k);
          CF(i, k) = 0.0;
        }
      cff3 = my_E2 / (vonKar * vonKar);
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          double strat2;
          if ((buoy2(i, j, k) > -5.0E-5) && (buoy2(i, j, k) < 0.0)) strat2 = 0.0;
          else strat2 = buoy2(i, j, k);
          const double Qprod = shear2(i, j, k) * (Akv(i, j, k) - Akv_bak) - strat2 * (Akt(i, j, k, itemp) - p->Akt_bak[itemp - 1]);
          const double Ls_unlmt = MAX(eps, gls(i, j, k, nstp) / (MAX(tke(i, j, k, nstp), eps)));
          cff1 = 0.5 * (Hz(i, j, k) + Hz(i, j, k + 1));
          tke(i, j, k, nnew) = tke(i, j, k, nnew) + dt * cff1 * Qprod * 2.0;
          gls(i, j, k, nnew) = gls(i, j, k, nnew) + dt * cff1 * Qprod * my_E1 * Ls_unlmt;
          const double Qdiss = dt * sqrt(tke(i, j, k, nstp)) / (my_B1 * Ls_unlmt);
          cff = Ls_unlmt * (1.0 / (z_w(i, j, N) - z_w(i, j, k)) + 1.0 / (z_w(i, j, k) - z_w(i, j, 0)));
          const double Wscale = 1.0 + cff3 * cff * cff;
          BCK(i, k) = cff1 * (1.0 + 2.0 * Qdiss) - FCK(i, k) - FCK(i, k + 1);
          BCP(i, k) = cff1 * (1.0 + Wscale * Qdiss) - FCK(i, k) - FCK(i, k + 1);
        }
      for (int i = Istr; i <= Iend; i++) {
        const double sx = sustr(i, j) + sustr(i + 1, j), sy = svstr(i, j) + svstr(i, j + 1);
        tke(i, j, N, nnew) = my_B1p2o3 * 0.5 * sqrt(sx * sx + sy * sy);
        gls(i, j, N, nnew) = 0.0;
        const double bx = bustr(i, j) + bustr(i + 1, j), by = bvstr(i, j) + bvstr(i, j + 1);
        tke(i, j, 0, nnew) = my_B1p2o3 * 0.5 * sqrt(bx * bx + by * by);
        gls(i, j, 0, nnew) = 0.0;
      }
      /* the two tridiagonal systems, eliminated from the top (:649-692) */
      for (int i = Istr; i <= Iend; i++) {
        cff = 1.0 / BCK(i, N - 1);
        CF(i, N - 1) = cff * FCK(i, N - 1);
        tke(i, j, N - 1, nnew) = cff * (tke(i, j, N - 1, nnew) - FCK(i, N) * tke(i, j, N, nnew));
      }
      for (int k = N - 2; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) {
          cff = 1.0 / (BCK(i, k) - CF(i, k + 1) * FCK(i, k + 1));
          CF(i, k) = cff * FCK(i, k);
          tke(i, j, k, nnew) = cff * (tke(i, j, k, nnew) - FCK(i, k + 1) * tke(i, j, k + 1, nnew));
        }
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) tke(i, j, k, nnew) = tke(i, j, k, nnew) - CF(i, k) * tke(i, j, k - 1, nnew);
      for (int i = Istr; i <= Iend; i++) {
        cff = 1.0 / BCP(i, N - 1);
        CF(i, N - 1) = cff * FCK(i, N - 1);
        gls(i, j, N - 1, nnew) = cff * (gls(i, j, N - 1, nnew) - FCK(i, N) * gls(i, j, N, nnew));
      }
      for (int k = N - 2; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) {
          cff = 1.0 / (BCP(i, k) - CF(i, k + 1) * FCK(i, k + 1));
          CF(i, k) = cff * FCK(i, k);
          gls(i, j, k, nnew) = cff * (gls(i, j, k, nnew) - FCK(i, k + 1) * gls(i, j, k + 1, nnew));
        }
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) gls(i, j, k, nnew) = gls(i, j, k, nnew) - CF(i, k) * gls(i, j, k - 1, nnew);
      /* mixing coefficients (:699-770) */
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          tke(i, j, k, nnew) = MAX(tke(i, j, k, nnew), my_qmin);
          gls(i, j, k, nnew) = MAX(gls(i, j, k, nnew), my_qmin);
          const double Ls_unlmt = gls(i, j, k, nnew) / tke(i, j, k, nnew);
          const double Ls_lmt = MIN(Ls_unlmt, my_lmax * sqrt(tke(i, j, k, nnew) / (MAX(0.0, buoy2(i, j, k)) + eps)));
          const double Gh = MIN(my_Gh0, -buoy2(i, j, k) * Ls_lmt * Ls_lmt / tke(i, j, k, nnew));
          cff = 1.0 - K.my_Sh2 * Gh;
          const double Sh = K.my_Sh1 / cff;
          double Sm;
          if (p->gls_stability == GLS_KANTHA_CLAYSON) Sm = (K.my_B1pm1o3 + Sh * Gh * K.my_Sm4) / (1.0 - K.my_Sm2 * Gh);
          else Sm = (K.my_Sm3 + Sh * Gh * K.my_Sm4) / (1.0 - K.my_Sm2 * Gh);
          const double ql = 0.5 * (Ls_lmt * sqrt(tke(i, j, k, nnew)) + Lscale(i, j, k) * sqrt(tke(i, j, k, nstp)));
          Akv(i, j, k) = Akv_bak + ql * Sm;
          for (int itrc = 1; itrc <= NAT; itrc++) Akt(i, j, k, itrc) = p->Akt_bak[itrc - 1] + ql * Sh;
          Akk(i, j, k) = Akk_bak + ql * my_Sq;
          Lscale(i, j, k) = Ls_lmt;
        }
      continue;
    }
    /* vertical mixing, production, dissipation (:706-800) */
    cff = -0.5 * dt;
    for (int i = Istr; i <= Iend; i++) {
      for (int k = 2; k <= N - 1; k++) {
        FCK(i, k) = cff * (Akk(i, j, k) + Akk(i, j, k - 1)) / Hz(i, j, k);
        FCP(i, k) = cff * (Akp(i, j, k) + Akp(i, j, k - 1)) / Hz(i, j, k);
        CF(i, k) = 0.0;
      }
      FCP(i, 1) = 0.0;
      FCP(i, N) = 0.0;
      FCK(i, 1) = 0.0;
      FCK(i, N) = 0.0;
    }
    for (int i = Istr; i <= Iend; i++)
      for (int k = 1; k <= N - 1; k++) {
        const double strat2 = buoy2(i, j, k);
        const double gls_c3 = (strat2 > 0.0) ? gls_c3m : gls_c3p;
        double Kprod = shear2(i, j, k) * (Akv(i, j, k) - Akv_bak) - strat2 * (Akt(i, j, k, itemp) - p->Akt_bak[itemp - 1]);
        double Pprod = gls_c1 * shear2(i, j, k) * (Akv(i, j, k) - Akv_bak) -
                       gls_c3 * strat2 * (Akt(i, j, k, itemp) - p->Akt_bak[itemp - 1]);
        cff1 = 1.0;
        if (Kprod < 0.0) {
          Kprod = Kprod + strat2 * (Akt(i, j, k, itemp) - p->Akt_bak[itemp - 1]);
          cff1 = 0.0;
        }
        cff2 = 1.0;
        if (Pprod < 0.0) {
          Pprod = Pprod + gls_c3 * strat2 * (Akt(i, j, k, itemp) - p->Akt_bak[itemp - 1]);
          cff2 = 0.0;
        }
        cff = 0.5 * (Hz(i, j, k) + Hz(i, j, k + 1));
        tke(i, j, k, nnew) = tke(i, j, k, nnew) + dt * cff * Kprod;
        gls(i, j, k, nnew) = gls(i, j, k, nnew) + dt * cff * Pprod * gls(i, j, k, nstp) / MAX(tke(i, j, k, nstp), gls_Kmin);
        double wall_fac = 1.0;
        if (Lmy25) {
          const double q1 = pow(gls(i, j, k, nstp), gls_exp1) * cmu_fac1 * pow(tke(i, j, k, nstp), -tke_exp1) *
                            (1.0 / (z_w(i, j, k) - z_w(i, j, 0)));
          const double q2 = pow(gls(i, j, k, nstp), gls_exp1) * cmu_fac1 * pow(tke(i, j, k, nstp), -tke_exp1) *
                            (1.0 / (z_w(i, j, N) - z_w(i, j, k)));
          wall_fac = 1.0 + K.E2 / (vonKar * vonKar) * (q1 * q1) + 0.25 / (vonKar * vonKar) * (q2 * q2);
        }
        BCK(i, k) = cff * (1.0 + dt * pow(gls(i, j, k, nstp), -gls_exp1) * cmu_fac2 * pow(tke(i, j, k, nstp), tke_exp2) +
                           dt * (1.0 - cff1) * strat2 * (Akt(i, j, k, itemp) - p->Akt_bak[itemp - 1]) / tke(i, j, k, nstp)) -
                    FCK(i, k) - FCK(i, k + 1);
        BCP(i, k) = cff * (1.0 + dt * gls_c2 * wall_fac * pow(gls(i, j, k, nstp), -gls_exp1) * cmu_fac2 *
                                     pow(tke(i, j, k, nstp), tke_exp2) +
                           dt * (1.0 - cff2) * gls_c3 * strat2 * (Akt(i, j, k, itemp) - p->Akt_bak[itemp - 1]) /
                               tke(i, j, k, nstp)) -
                    FCP(i, k) - FCP(i, k + 1);
      }
    /* Dirichlet surface and bottom values (:806-860) */
    for (int i = Istr; i <= Iend; i++) {
      const double sus = (sustr(i, j) + sustr(i + 1, j)), svs = (svstr(i, j) + svstr(i, j + 1));
      tke(i, j, N, nnew) = MAX(cmu_fac3 * 0.5 * sqrt(sus * sus + svs * svs), gls_Kmin);
      const double bus = (bustr(i, j) + bustr(i + 1, j)), bvs = (bvstr(i, j) + bvstr(i, j + 1));
      tke(i, j, 0, nnew) = MAX(cmu_fac3 * 0.5 * sqrt(bus * bus + bvs * bvs), gls_Kmin);
      Zos_eff[i - IminS] = Zos_min;
      gls(i, j, N, nnew) = MAX(pow(gls_cmu0, gls_p) * pow(tke(i, j, N, nnew), gls_m) * pow(L_sft * Zos_eff[i - IminS], gls_n),
                               gls_Pmin);
      cff = gls_fac4 * pow(vonKar * Zob_min(i, j), gls_n);
      gls(i, j, 0, nnew) = MAX(cff * pow(tke(i, j, 0, nnew), gls_m), gls_Pmin);
    }
    /* tri-diagonal system for tke (:864-895) */
    for (int i = Istr; i <= Iend; i++) {
      tke_fluxt[i - IminS] = 0.0;
      tke_fluxb[i - IminS] = 0.0;
      cff = 1.0 / BCK(i, N - 1);
      CF(i, N - 1) = cff * FCK(i, N - 1);
      tke(i, j, N - 1, nnew) = cff * (tke(i, j, N - 1, nnew) + tke_fluxt[i - IminS]);
    }
    for (int i = Istr; i <= Iend; i++) {
      for (int k = N - 2; k >= 1; k--) {
        cff = 1.0 / (BCK(i, k) - CF(i, k + 1) * FCK(i, k + 1));
        CF(i, k) = cff * FCK(i, k);
        tke(i, j, k, nnew) = cff * (tke(i, j, k, nnew) - FCK(i, k + 1) * tke(i, j, k + 1, nnew));
      }
      tke(i, j, 1, nnew) = tke(i, j, 1, nnew) - cff * tke_fluxb[i - IminS];
    }
    for (int k = 2; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) tke(i, j, k, nnew) = tke(i, j, k, nnew) - CF(i, k) * tke(i, j, k - 1, nnew);
    /* tri-diagonal system for gls (:899-960) */
    for (int i = Istr; i <= Iend; i++) {
      cff = 0.5 * (tke(i, j, N, nnew) + tke(i, j, N - 1, nnew));
      gls_fluxt[i - IminS] = dt * gls_fac3 * pow(cff, gls_m) * pow(L_sft, gls_n) *
                             pow(Zos_eff[i - IminS] + 0.5 * Hz(i, j, N), gls_n - 1.0) * 0.5 * (Akp(i, j, N) + Akp(i, j, N - 1));
      cff = 0.5 * (tke(i, j, 0, nnew) + tke(i, j, 1, nnew));
      gls_fluxb[i - IminS] = dt * gls_fac2 * (pow(cff, gls_m)) * pow(0.5 * Hz(i, j, 1) + Zob_min(i, j), gls_n - 1.0) * 0.5 *
                             (Akp(i, j, 0) + Akp(i, j, 1));
      cff = 1.0 / BCP(i, N - 1);
      CF(i, N - 1) = cff * FCP(i, N - 1);
      gls(i, j, N - 1, nnew) = cff * (gls(i, j, N - 1, nnew) - gls_fluxt[i - IminS]);
    }
    for (int i = Istr; i <= Iend; i++) {
      for (int k = N - 2; k >= 1; k--) {
        cff = 1.0 / (BCP(i, k) - CF(i, k + 1) * FCP(i, k + 1));
        CF(i, k) = cff * FCP(i, k);
        gls(i, j, k, nnew) = cff * (gls(i, j, k, nnew) - FCP(i, k + 1) * gls(i, j, k + 1, nnew));
      }
      gls(i, j, 1, nnew) = gls(i, j, 1, nnew) - cff * gls_fluxb[i - IminS];
    }
    for (int k = 2; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) gls(i, j, k, nnew) = gls(i, j, k, nnew) - CF(i, k) * gls(i, j, k - 1, nnew);
    /* vertical mixing coefficients (:964-1095) */
    for (int i = Istr; i <= Iend; i++) {
      for (int k = 1; k <= N - 1; k++) {
        tke(i, j, k, nnew) = MAX(tke(i, j, k, nnew), gls_Kmin);
        gls(i, j, k, nnew) = MAX(gls(i, j, k, nnew), gls_Pmin);
        if (gls_n >= 0.0)
          gls(i, j, k, nnew) = MIN(gls(i, j, k, nnew), gls_fac5 * pow(tke(i, j, k, nnew), tke_exp4) *
                                                           pow(sqrt(MAX(0.0, buoy2(i, j, k))) + eps, -gls_n));
        else
          gls(i, j, k, nnew) = MAX(gls(i, j, k, nnew), gls_fac5 * pow(tke(i, j, k, nnew), tke_exp4) *
                                                           pow(sqrt(MAX(0.0, buoy2(i, j, k))) + eps, -gls_n));
        const double Ls_unlmt = MAX(eps, pow(gls(i, j, k, nnew), gls_exp1) * cmu_fac1 * pow(tke(i, j, k, nnew), -tke_exp1));
        double Ls_lmt;
        if (buoy2(i, j, k) > 0.0) Ls_lmt = MIN(Ls_unlmt, sqrt(0.56 * tke(i, j, k, nnew) / (MAX(0.0, buoy2(i, j, k)) + eps)));
        else Ls_lmt = Ls_unlmt;
        gls(i, j, k, nnew) = MAX(pow(gls_cmu0, gls_p) * pow(tke(i, j, k, nnew), gls_m) * pow(Ls_lmt, gls_n), gls_Pmin);
        double Gh = MIN(K.Gh0, -buoy2(i, j, k) * Ls_lmt * Ls_lmt / (2.0 * tke(i, j, k, nnew)));
        Gh = MIN(Gh, Gh - ((Gh - K.Ghcri) * (Gh - K.Ghcri)) / (Gh + K.Gh0 - 2.0 * K.Ghcri));
        Gh = MAX(Gh, K.Ghmin);
        double Sm, Sh;
        if (p->gls_stability == GLS_CANUTO_A || p->gls_stability == GLS_CANUTO_B) {
          double Gm = (K.b0 / gls_fac6 - K.b1 * Gh + K.b3 * gls_fac6 * (Gh * Gh)) / (K.b2 - K.b4 * gls_fac6 * Gh);
          Gm = MIN(Gm, shear2(i, j, k) * Ls_lmt * Ls_lmt / (2.0 * tke(i, j, k, nnew)));
          cff = K.b0 - K.b1 * gls_fac6 * Gh + K.b2 * gls_fac6 * Gm + K.b3 * (gls_fac6 * gls_fac6) * (Gh * Gh) -
                K.b4 * (gls_fac6 * gls_fac6) * Gh * Gm + K.b5 * (gls_fac6 * gls_fac6) * Gm * Gm;
          Sm = (K.s0 - K.s1 * gls_fac6 * Gh + K.s2 * gls_fac6 * Gm) / cff;
          Sh = (K.s4 - K.s5 * gls_fac6 * Gh + K.s6 * gls_fac6 * Gm) / cff;
          Sm = MAX(Sm, 0.0);
          Sh = MAX(Sh, 0.0);
          Sm = Sm * sqrt2 / (gls_cmu0 * gls_cmu0 * gls_cmu0);
          Sh = Sh * sqrt2 / (gls_cmu0 * gls_cmu0 * gls_cmu0);
        } else if (p->gls_stability == GLS_KANTHA_CLAYSON) {
          cff = 1.0 - K.my_Sh2 * Gh;
          Sh = K.my_Sh1 / cff;
          Sm = (K.my_B1pm1o3 + K.my_Sm4 * Sh * Gh) / (1.0 - K.my_Sm2 * Gh);
        } else {
          cff = 1.0 - K.my_Sh2 * Gh;
          Sh = K.my_Sh1 / cff;
          Sm = (K.my_Sm3 + Sh * Gh * K.my_Sm4) / (1.0 - K.my_Sm2 * Gh);
        }
        const double ql = sqrt2 * 0.5 * (Ls_lmt * sqrt(tke(i, j, k, nnew)) + Lscale(i, j, k) * sqrt(tke(i, j, k, nstp)));
        Akv(i, j, k) = Akv_bak + Sm * ql;
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(i, j, k, itrc) = p->Akt_bak[itrc - 1] + Sh * ql;
        Akk(i, j, k) = Akk_bak + Sm * ql / gls_sigk;
        Akp(i, j, k) = Akp_bak + Sm * ql * ogls_sigp;
        Lscale(i, j, k) = Ls_lmt;
      }
      Akv(i, j, N) = Akv_bak + L_sft * Zos_eff[i - IminS] * gls_cmu0 * sqrt(tke(i, j, N, nnew));
      Akv(i, j, 0) = Akv_bak + vonKar * Zob_min(i, j) * gls_cmu0 * sqrt(tke(i, j, 0, nnew));
      Akk(i, j, N) = Akk_bak + Akv(i, j, N) / gls_sigk;
      Akk(i, j, 0) = Akk_bak + Akv(i, j, 0) / gls_sigk;
      Akp(i, j, N) = Akp_bak + Akv(i, j, N) * ogls_sigp;
      Akp(i, j, 0) = Akp_bak + Akv(i, j, 0) / gls_sigp;
      for (int itrc = 1; itrc <= NAT; itrc++) {
        Akt(i, j, N, itrc) = p->Akt_bak[itrc - 1];
        Akt(i, j, 0, itrc) = p->Akt_bak[itrc - 1];
      }
    }
  }
  /* lateral boundary conditions of Akv, Akt as written (:1100-1185): note Iend-1 on the eastern edge */
  for (int k = 0; k <= N; k++) {
    if (west_edge)
      for (int j = Jstr; j <= Jend; j++) {
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(Istr - 1, j, k, itrc) = Akt(Istr, j, k, itrc);
        Akv(Istr - 1, j, k) = Akv(Istr, j, k);
      }
    if (east_edge)
      for (int j = Jstr; j <= Jend; j++) {
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(Iend - 1, j, k, itrc) = Akt(Iend, j, k, itrc);
        Akv(Iend - 1, j, k) = Akv(Iend, j, k);
      }
    if (south_edge)
      for (int i = Istr; i <= Iend; i++) {
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(i, Jstr - 1, k, itrc) = Akt(i, Jstr, k, itrc);
        Akv(i, Jstr - 1, k) = Akv(i, Jstr, k);
      }
    if (north_edge)
      for (int i = Istr; i <= Iend; i++) {
        for (int itrc = 1; itrc <= NAT; itrc++) Akt(i, Jend + 1, k, itrc) = Akt(i, Jend, k, itrc);
        Akv(i, Jend + 1, k) = Akv(i, Jend, k);
      }
    if (south_edge && west_edge) {
      for (int itrc = 1; itrc <= NAT; itrc++)
        Akt(Istr - 1, Jstr - 1, k, itrc) = 0.5 * (Akt(Istr, Jstr - 1, k, itrc) + Akt(Istr - 1, Jstr, k, itrc));
      Akv(Istr - 1, Jstr - 1, k) = 0.5 * (Akv(Istr, Jstr - 1, k) + Akv(Istr - 1, Jstr, k));
    }
    if (south_edge && east_edge) {
      for (int itrc = 1; itrc <= NAT; itrc++)
        Akt(Iend + 1, Jstr - 1, k, itrc) = 0.5 * (Akt(Iend, Jstr - 1, k, itrc) + Akt(Iend + 1, Jstr, k, itrc));
      Akv(Iend + 1, Jstr - 1, k) = 0.5 * (Akv(Iend, Jstr - 1, k) + Akv(Iend + 1, Jstr, k));
    }
    if (north_edge && west_edge) {
      for (int itrc = 1; itrc <= NAT; itrc++)
        Akt(Istr - 1, Jend + 1, k, itrc) = 0.5 * (Akt(Istr, Jend + 1, k, itrc) + Akt(Istr - 1, Jend, k, itrc));
      Akv(Istr - 1, Jend + 1, k) = 0.5 * (Akv(Istr, Jend + 1, k) + Akv(Istr - 1, Jend, k));
    }
    if (north_edge && east_edge) {
      for (int itrc = 1; itrc <= NAT; itrc++)
        Akt(Iend + 1, Jend + 1, k, itrc) = 0.5 * (Akt(Iend, Jend + 1, k, itrc) + Akt(Iend + 1, Jend, k, itrc));
      Akv(Iend + 1, Jend + 1, k) = 0.5 * (Akv(Iend, Jend + 1, k) + Akv(Iend + 1, Jend, k));
    }
  }
  o_tkebc(b, p, F, nnew);
  o_exchange3d(b, GT_R, N + 1, &tke(LBi, LBj, 0, nnew));
  o_exchange3d(b, GT_R, N + 1, &gls(LBi, LBj, 0, nnew));
  o_exchange3d(b, GT_R, N + 1, F->Akv);
  for (int itrc = 1; itrc <= NAT; itrc++) o_exchange3d(b, GT_R, N + 1, F->Akt + (long)(itrc - 1) * n3w);
  free(tke_fluxt); free(tke_fluxb); free(gls_fluxt); free(gls_fluxb); free(Zos_eff);
  free(BCK_); free(BCP_); free(CF_); free(FCK_); free(FCP_); free(dU_); free(dV_); free(shear2_); free(buoy2_);
  free(FEK_); free(FEP_); free(FXK_); free(FXP_); free(Zob_min_); free(curvK_); free(curvP_); free(gradK_); free(gradP_);
  return 0;
}
