// k_base.hip -- closed-wall boundary conditions and the small glue kernels of
// the 3-D step: set_massflux, omega, set_zeta, set_depth.
//
// All of them are HBM-bound single-pass kernels: one thread per (i,j) water
// column, consecutive lanes = consecutive i, so every k-level access of a
// wavefront is one contiguous 512-byte segment.
#include "roms_dev.h"

int roms_entry_check(const char *name);

// Effective boundary-condition code (enum roms_lbc) of variable v (enum roms_lbc_var) on side sd.
int lbc_code(const roms_params_t &p, int sd, int v)
{
  if (p.lbc[sd][v]) return p.lbc[sd][v];
  return sd == LBS_WEST ? p.lbc_west : sd == LBS_EAST ? p.lbc_east : sd == LBS_SOUTH ? p.lbc_south : p.lbc_north;
}

// Which conditions are implemented, per variable: closed, gradient, clamped, radiation (all six), Chapman implicit
// (zeta), Flather (ubar, vbar: the normal component; the tangential one gets the reference's Chapman-type rule).
// A periodic direction is periodic on both sides; a physical edge takes one of the conditions above.
int check_lbc()
{
  const roms_bounds_t &b = g_ctx.b;
  const roms_params_t &p = g_ctx.p;
  if (b.NSperiodic)
    return roms_fail("check_lbc", "N-S periodic grids are not implemented on this path");
  for (int sd = LBS_WEST; sd <= LBS_NORTH; sd++) {
    const bool periodic = (sd <= LBS_EAST) ? b.EWperiodic != 0 : b.NSperiodic != 0;
    for (int v = 0; v < LBV_COUNT; v++) {
      const int c = lbc_code(p, sd, v);
      if (periodic) {
        if (c != LBC_PERIODIC) return roms_fail("check_lbc", "a periodic direction needs LBC = Per on both of its sides");
        continue;
      }
      bool ok = c == LBC_CLOSED || c == LBC_GRADIENT || c == LBC_CLAMPED || c == LBC_RADIATION || c == LBC_RADIATION_NUDGING;
      if (v == LBV_ZETA) ok = ok || c == LBC_CHAPMAN_IMPLICIT || c == LBC_CHAPMAN_EXPLICIT;
      if (v == LBV_VBAR || v == LBV_UBAR) ok = ok || c == LBC_FLATHER || c == LBC_SHCHEPETKIN || c == LBC_REDUCED;
      if (!ok) return roms_fail("check_lbc", "lateral boundary condition not implemented for this variable");
    }
  }
  return 0;
}

// every 2-D variable closed on its physical S/N edges and the E-W direction periodic: step2d may apply the
// conditions inside its fused kernel
bool lbc2d_all_closed()
{
  const roms_params_t &p = g_ctx.p;
  if (!g_ctx.b.EWperiodic) return false;
  for (int sd = LBS_SOUTH; sd <= LBS_NORTH; sd++)
    for (int v = LBV_ZETA; v <= LBV_VBAR; v++)
      if (lbc_code(p, sd, v) != LBC_CLOSED) return false;
  return true;
}

// ---------------------------------------------------------------------------
// Lateral boundary conditions, one launch per variable: zetabc.F:48, u2dbc_im.F:51, v2dbc_im.F:52, u3dbc_im.F:50,
// v3dbc_im.F:50, t3dbc_im.F:50 on all four edges + the corner rule (e.g. zetabc.F:699-731); bc_w3d_tile
// (bc_3d.F:588) = gradient without mask.  The reference writes each edge out; its western / eastern blocks are the
// transposes of the southern / northern ones (pm <-> pn, umask <-> vmask), so a thread works in "edge
// coordinates": B = boundary point, P1 / P2 = first / second point inward along the normal, +-T = neighbours along
// the edge.  closed, gradient, clamped, Chapman implicit, Flather, radiation (implicit upstream, no nudging, no
// RADIATION_2D).
// X = the (i,j,k) array of the wanted time level; Xold = the same variable at the level the condition reads (know
// for the 2-D conditions, nstp for 3-D radiation); D = boundary data; Z, Zb = zeta(know), zeta_bry.
// ---------------------------------------------------------------------------
struct BcArgs {
  double *X;             // level written (kout / nout)
  const double *Xold;    // 2-D conditions: X(know); 3-D radiation: X(nstp)
  const double *D;       // boundary data of this variable (or nullptr)
  const double *Z, *Zb;  // Flather, Shchepetkin (ubar, vbar): zeta(know), zeta_bry
  const double *Zn;      // Shchepetkin: zeta at the level being written
  const double *T;       // reduced physics: the other barotropic component at know
  int acquire[4];        // reduced physics: boundary data of the free surface exist on the side (inp_decode.F:1620-1655)
  int var;               // enum roms_lbc_var; -1 = bc_w3d (gradient, no mask)
  int code[4];           // enum roms_lbc on the western / eastern / southern / northern edge
  int nk, masked;
  double dt2d;
};

__device__ __forceinline__ double bc_radiate(double xb_old, double x1_old, double x1, double x2, double gL, double gR,
                                             bool &inward, bool rad2d, double gLb, double gRb)
{
  const double eps = 1.0E-20;
  double dXdt = x1_old - x1;
  const double dXdn = x1 - x2;
  inward = (dXdt * dXdn) < 0.0;                    // selects the nudging time scale (RadNud), t3dbc_im.F:138-146
  if ((dXdt * dXdn) < 0.0) dXdt = 0.0;
  const double dXds = ((dXdt * (gL + gR)) > 0.0) ? gL : gR;
  const double cff = fmax(dXds * dXds + dXdn * dXdn, eps);
  const double Cn = dXdt * dXdn;
  if (rad2d) {                                     // RADIATION_2D: tangential phase speed, e.g. zetabc.F:141-160
    const double Ct = fmin(cff, fmax(dXdt * dXds, -cff));
    return (cff * xb_old + Cn * x1 - fmax(Ct, 0.0) * gLb - fmin(Ct, 0.0) * gRb) / (cff + Cn);
  }
  return (cff * xb_old + Cn * x1) / (cff + Cn);
}

// grid: x = position along the edge, y = level, z = edge (0 W, 1 E, 2 S, 3 N)
__global__ void k_edge_bc(const RomsDev *__restrict__ c, BcArgs a)
{
  DEV_PROLOGUE(c)
  const int k = blockIdx.y;
  if (k >= a.nk) return;
  const roms_params_t &p = c->p;
  const int side = blockIdx.z;
  const bool we = side <= LBS_EAST, hi = side == LBS_EAST || side == LBS_NORTH;
  if (!(side == LBS_WEST ? b.west_edge : side == LBS_EAST ? b.east_edge : side == LBS_SOUTH ? b.south_edge : b.north_edge)) return;
  if (we ? b.EWperiodic : b.NSperiodic) return;
  const int code = a.code[side];
  const bool utype = a.var == LBV_UBAR || a.var == LBV_U, vtype = a.var == LBV_VBAR || a.var == LBV_V;
  const bool normal = (utype && we) || (vtype && !we);
  // along-edge range (tangential components: u2dbc_im.F:829-1140, v2dbc_im.F:812-1120)
  int a0, a1, bi = 0, bj = 0;
  if (we) {
    bi = hi ? b.Iend + 1 : (utype ? b.Istr : b.Istr - 1);
    a0 = b.Jstr; a1 = b.Jend;
    if (vtype) {
      a0 = b.JstrV;
      if (code == LBC_CLOSED) { a0 = b.NSperiodic ? b.JstrV : b.Jstr; a1 = b.NSperiodic ? b.Jend : b.JendR; }
    }
  } else {
    bj = hi ? b.Jend + 1 : (vtype ? b.Jstr : b.Jstr - 1);
    a0 = b.Istr; a1 = b.Iend;
    if (utype) {
      a0 = b.IstrU;
      if (code == LBC_CLOSED) { a0 = b.EWperiodic ? b.IstrU : b.Istr; a1 = b.EWperiodic ? b.Iend : b.IendR; }
    }
  }
  const int al = a0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (al > a1) return;
  const long sn = (we ? 1 : ni) * (hi ? -1 : 1), st = we ? ni : 1;
  const long B = I2(we ? bi : al, we ? al : bj), P1 = B + sn, P2 = P1 + sn;
  const long kb = (long)k * nij;
  double *X = a.X + kb;
  double x;
  if (code == LBC_RADIATION || code == LBC_RADIATION_NUDGING) {
    const double *O = a.Xold + kb;
    double gL = O[P1] - O[P1 - st], gR = O[P1 + st] - O[P1];
    if (a.masked && (a.var == LBV_T || a.var == LBV_ZETA)) {      // zetabc.F:112-120, t3dbc_im.F:370-379
      const double *gm = we ? c->F.vmask : c->F.umask;
      gL = gL * gm[P1];
      gR = gR * gm[P1 + st];
    }
    // zetabc.F:424 -- on the southern edge the free surface takes its normal difference towards the boundary row
    const long Q2 = (a.var == LBV_ZETA && side == LBS_SOUTH) ? B : P2;
    bool inward;
    double gLb = O[B] - O[B - st], gRb = O[B + st] - O[B];        // along-edge differences of the boundary row
    if (a.masked && (a.var == LBV_T || a.var == LBV_ZETA)) {
      const double *gm = we ? c->F.vmask : c->F.umask;
      gLb = gLb * gm[B];
      gRb = gRb * gm[B + st];
    }
    // zetabc.F:455-456 -- ... and the along-edge differences of the first inside row in the tangential term
    if (a.var == LBV_ZETA && side == LBS_SOUTH) { gLb = gL; gRb = gR; }
    x = bc_radiate(O[B], O[P1], X[P1], X[Q2], gL, gR, inward, p.radiation_2d != 0, gLb, gRb);
    if (code == LBC_RADIATION_NUDGING) {            // explicit nudging towards the boundary data, zetabc.F:162-166 ...
      double tau = inward ? p.obc_in[side][a.var] : p.obc_out[side][a.var];
      tau = tau * (a.var <= LBV_VBAR ? a.dt2d : p.dt);
      x = x + tau * (a.D[B + kb] - O[B]);
    }
  } else if (code == LBC_CLAMPED) {
    x = a.D[B + kb];
  } else if (code == LBC_CHAPMAN_IMPLICIT) {        // zetabc.F:193-220, :342, :491, :640
    const double cff = a.dt2d * (we ? c->F.pm : c->F.pn)[P1];
    const double cff1 = sqrt(p.g * (c->F.h[P1] + a.Xold[P1]));
    const double Cn = cff * cff1;
    const double cff2 = 1.0 / (1.0 + Cn);
    x = cff2 * (a.Xold[B] + Cn * X[P1]);
  } else if (code == LBC_CHAPMAN_EXPLICIT) {        // zetabc.F:175-190, :324, :473, :622
    const double cff = a.dt2d * (we ? c->F.pm : c->F.pn)[P1];
    const double cff1 = sqrt(p.g * (c->F.h[P1] + a.Xold[P1]));
    const double Cn = cff * cff1;
    x = (1.0 - Cn) * a.Xold[B] + Cn * a.Xold[P1];
  } else if (code == LBC_SHCHEPETKIN && normal) {   // u2dbc_im.F:288-362, :636-710; v2dbc_im.F:290-364, :639-713 (no SSH_TIDES)
    const double *pmn = we ? c->F.pm : c->F.pn;
    const long lo = B - (we ? 1 : ni), qi = hi ? lo : B, qo = hi ? B : lo;     // rho-points inside / outside
    const double Co = 1.0 / (2.0 + sqrt(2.0));      // mod_scalars.F:4175
    const double bry_val = a.D[B];
    // WET_DRY: the total depth instead of the resting one (u2dbc_im.F:331-340, :679-688; v2dbc_im.F:333, :682)
    const double cff = p.wet_dry ? 0.5 * (c->F.h[lo] + a.Z[lo] + c->F.h[B] + a.Z[B]) : 0.5 * (c->F.h[lo] + c->F.h[B]);
    const double cff1 = sqrt(p.g / cff);
    const double Cn = a.dt2d * cff1 * cff * 0.5 * (pmn[lo] + pmn[B]);
    double Zx = (0.5 + Cn) * a.Z[qi] + (0.5 - Cn) * a.Z[qo];
    if (Cn > Co) {
      const double cff2 = (1.0 - Co / Cn) * (1.0 - Co / Cn);
      const double cff3 = a.Zn[qi] + Cn * a.Z[qo] - (1.0 + Cn) * a.Z[qi];
      Zx = Zx + cff2 * cff3;
    }
    x = hi ? 0.5 * ((1.0 - Cn) * a.Xold[B] + Cn * a.Xold[P1] + bry_val + cff1 * (Zx - a.Zb[qo]))
           : 0.5 * ((1.0 - Cn) * a.Xold[B] + Cn * a.Xold[P1] + bry_val - cff1 * (Zx - a.Zb[qo]));
  } else if (code == LBC_REDUCED && normal) {       // u2dbc_im.F:392-432, :740-780; v2dbc_im.F:394-436, :743-785
    const double *pmn = we ? c->F.pm : c->F.pn;
    const long un = we ? 1 : ni, lo = B - un, qi = hi ? lo : B, qo = hi ? B : lo;
    double bry_pgr, bry_cor = 0.0;
    if (a.acquire[side]) bry_pgr = hi ? -p.g * (a.Zb[qo] - a.Z[qi]) * 0.5 * pmn[qi] : -p.g * (a.Z[qi] - a.Zb[qo]) * 0.5 * pmn[qi];
    else bry_pgr = -p.g * (a.Z[B] - a.Z[lo]) * 0.5 * (pmn[lo] + pmn[B]);
    if (p.uv_cor) {
      bry_cor = 0.125 * (a.T[lo] + a.T[lo + st] + a.T[B] + a.T[B + st]) * (c->F.f[lo] + c->F.f[B]);
      if (!we) bry_cor = -bry_cor;
    }
    const double cff = 1.0 / (0.5 * (c->F.h[lo] + a.Z[lo] + c->F.h[B] + a.Z[B]));
    const double bry_str = cff * ((we ? c->F.sustr : c->F.svstr)[B] - (we ? c->F.bustr : c->F.bvstr)[B]);
    x = a.Xold[B] + a.dt2d * (bry_pgr + bry_cor + bry_str);
  } else if (code == LBC_FLATHER && normal) {       // u2dbc_im.F:214-300, v2dbc_im.F:216-286 (bry_val = boundary data)
    const long qa = B - (we ? 1 : ni), qc = B;      // the two rho-points around the velocity point, lower index first
    const double bry_val = a.D[B];
    const double cff = 1.0 / (0.5 * (c->F.h[qa] + a.Z[qa] + c->F.h[qc] + a.Z[qc]));
    const double Cn = sqrt(p.g * cff);
    const double zb = a.Zb[hi ? qc : qa];
    if (p.atm_press && p.press_compensate) {         // ATM_PRESS && PRESS_COMPENSATE, u2dbc_im.F:264-272, :612-620
      const double OneAtm = 1013.25, fac = 100.0 / (p.g * p.rho0);
      const double zm = 0.5 * (a.Z[qa] + a.Z[qc] + fac * (c->F.Pair[qa] + c->F.Pair[qc] - 2.0 * OneAtm));
      x = hi ? bry_val + Cn * (zm - zb) : bry_val - Cn * (zm - zb);
    } else
    x = hi ? bry_val + Cn * (0.5 * (a.Z[qa] + a.Z[qc]) - zb) : bry_val - Cn * (0.5 * (a.Z[qa] + a.Z[qc]) - zb);
  } else if (code == LBC_FLATHER || code == LBC_SHCHEPETKIN || code == LBC_REDUCED) {   // tangential component, Chapman type: u2dbc_im.F:912-932, v2dbc_im.F:886-906
    const double *pmn = we ? c->F.pm : c->F.pn;
    const double cff = a.dt2d * 0.5 * (pmn[P1 - st] + pmn[P1]);
    const double cff1 = sqrt(p.g * 0.5 * (c->F.h[P1 - st] + a.Z[P1 - st] + c->F.h[P1] + a.Z[P1]));
    const double Cn = cff * cff1;
    const double cff2 = 1.0 / (1.0 + Cn);
    x = cff2 * (a.Xold[B] + Cn * X[P1]);
  } else if (code == LBC_GRADIENT) {
    x = X[P1];
  } else {                                          // closed
    x = normal ? 0.0 : ((utype || vtype) ? p.gamma2 * X[P1] : X[P1]);
  }
  // masked = 1: every branch but the closed normal one ends with the mask of the boundary point (the six
  // boundary-condition routines); masked = 2: only the closed tangential branch does (bc_2d.F / bc_3d.F)
  const bool do_mask = a.masked == 1 ? !(normal && code == LBC_CLOSED)
                                     : (a.masked == 2 && code == LBC_CLOSED && !normal && (utype || vtype));
  if (do_mask) {
    const double *M = utype ? c->F.umask : (vtype ? c->F.vmask : c->F.rmask);
    x = x * M[B];
  }
  // WET_DRY, 3-D momentum: the wet/dry mask after every land/sea-mask product of u3dbc_im.F / v3dbc_im.F (:174 ...
  // :681) -- but for u on a southern gradient edge, whose block tests a symbol no header defines (u3dbc_im.F:496)
  if (p.wet_dry && a.masked == 1 && (a.var == LBV_U || a.var == LBV_V) && !(normal && code == LBC_CLOSED) &&
      !(a.var == LBV_U && side == LBS_SOUTH && code == LBC_GRADIENT))
    x = x * (a.var == LBV_U ? c->F.umask_wet : c->F.vmask_wet)[B];
  X[B] = x;
}

// WET_DRY: what zetabc.F:733-827, u2dbc_im.F:1176-1293 and v2dbc_im.F:1169-1287 do after their edges and corners.
// One thread per boundary point; blockIdx.y = 0..3 the edges (ranges as written in the reference), 4 the corners.
//   zeta: a boundary free surface at or below Dcrit - h is lifted to (Dcrit - 1e-20) - h
//   ubar / vbar: times the wet/dry factor of the boundary face.  As written, the northern edge of ubar starts at Istr
//   where the southern one starts at IstrU, and the western edge of vbar takes its factor from the boundary point
//   (Istr-1,j) and applies it to the first interior point (Istr,j).
__global__ void k_wet_bc2d(const RomsDev *__restrict__ c, double *__restrict__ X, int var)
{
  DEV_PROLOGUE(c)
  const roms_params_t &p = c->p;
  const int side = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ew = !b.EWperiodic, ns = !b.NSperiodic;
  int i, j, it, jt;                                    // (i,j): the point whose mask and value decide; (it,jt): the target
  if (side == 4) {
    if (!(ew && ns) || t >= 4) return;
    const bool south = t < 2, west = (t & 1) == 0;
    if (!((south ? b.south_edge : b.north_edge) && (west ? b.west_edge : b.east_edge))) return;
    if (var == LBV_UBAR && !south && west) return;     // (Istr,Jend+1): the northern edge's range holds it already
    i = west ? (var == LBV_UBAR ? b.Istr : b.Istr - 1) : b.Iend + 1;
    j = south ? (var == LBV_VBAR ? b.Jstr : b.Jstr - 1) : b.Jend + 1;
    it = i; jt = j;
  } else if (side <= LBS_EAST) {
    if (!ew || !(side == LBS_WEST ? b.west_edge : b.east_edge)) return;
    j = (var == LBV_VBAR ? b.JstrV : b.Jstr) + t;
    if (j > b.Jend) return;
    i = side == LBS_EAST ? b.Iend + 1 : (var == LBV_UBAR ? b.Istr : b.Istr - 1);
    it = (var == LBV_VBAR && side == LBS_WEST) ? b.Istr : i;       // v2dbc_im.F:1180-1185
    jt = j;
  } else {
    if (!ns || !(side == LBS_SOUTH ? b.south_edge : b.north_edge)) return;
    i = ((var == LBV_UBAR && side == LBS_SOUTH) ? b.IstrU : b.Istr) + t;
    if (i > b.Iend) return;
    j = side == LBS_NORTH ? b.Jend + 1 : (var == LBV_VBAR ? b.Jstr : b.Jstr - 1);
    it = i; jt = j;
  }
  const long q = I2(i, j), qt = I2(it, jt);
  if (var == LBV_ZETA) {
    const double eps = 1.0E-20, cff = p.Dcrit - eps;
    if (X[q] <= (p.Dcrit - c->F.h[q])) X[q] = cff - c->F.h[q];
  } else {
    const double *W = var == LBV_UBAR ? c->F.umask_wet : c->F.vmask_wet;
    X[qt] = X[qt] * wet_factor(W[q], X[q]);
  }
}

// corners when neither direction is periodic, e.g. zetabc.F:699-731: the mean of the two neighbouring boundary
// points; one thread per (corner, level), after the edges
__global__ void k_corner_bc(const RomsDev *__restrict__ c, double *A, int gtype, int nk)
{
  DEV_PROLOGUE(c)
  const int k = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (k >= nk) return;
  const int iw = gtype == GT_U ? b.Istr : b.Istr - 1, ie = b.Iend + 1;
  const int js = gtype == GT_V ? b.Jstr : b.Jstr - 1, jn = b.Jend + 1;
  double *X = A + (long)k * nij;
  if (q == 0 && b.south_edge && b.west_edge) X[I2(iw, js)] = 0.5 * (X[I2(iw + 1, js)] + X[I2(iw, js + 1)]);
  if (q == 1 && b.south_edge && b.east_edge) X[I2(ie, js)] = 0.5 * (X[I2(ie - 1, js)] + X[I2(ie, js + 1)]);
  if (q == 2 && b.north_edge && b.west_edge) X[I2(iw, jn)] = 0.5 * (X[I2(iw, jn - 1)] + X[I2(iw + 1, jn)]);
  if (q == 3 && b.north_edge && b.east_edge) X[I2(ie, jn)] = 0.5 * (X[I2(ie, jn - 1)] + X[I2(ie - 1, jn)]);
}

static int edge_bc(BcArgs a)
{
  const roms_bounds_t &b = g_ctx.b;
  const bool any_edge = (!b.EWperiodic && (b.west_edge || b.east_edge)) || (!b.NSperiodic && (b.south_edge || b.north_edge));
  if (!any_edge) return 0;
  const int nx = b.Iend - b.Istr + 3, ny = b.Jend - b.Jstr + 3;
  const int nmax = nx > ny ? nx : ny;
  dim3 grid((nmax + 255) / 256, a.nk, 4);
  hipLaunchKernelGGL(k_edge_bc, grid, dim3(256), 0, g_ctx.stream, g_ctx.devc, a);
  KERNEL_CHECK("k_edge_bc");
  if (!b.EWperiodic && !b.NSperiodic && (b.west_edge || b.east_edge) && (b.south_edge || b.north_edge)) {
    const int gtype = (a.var == LBV_UBAR || a.var == LBV_U) ? GT_U : (a.var == LBV_VBAR || a.var == LBV_V) ? GT_V : GT_R;
    hipLaunchKernelGGL(k_corner_bc, dim3((a.nk + 63) / 64, 4), dim3(64), 0, g_ctx.stream, g_ctx.devc, a.X, gtype, a.nk);
    KERNEL_CHECK("k_corner_bc");
  }
  if (g_ctx.p.wet_dry && a.masked == 1 && a.var >= LBV_ZETA && a.var <= LBV_VBAR) {
    hipLaunchKernelGGL(k_wet_bc2d, dim3((nmax + 255) / 256, 5), dim3(256), 0, g_ctx.stream, g_ctx.devc, a.X, a.var);
    KERNEL_CHECK("k_wet_bc2d");
  }
  return 0;
}

static inline long nij_host()
{
  const roms_bounds_t &b = g_ctx.b;
  return (long)(b.UBi - b.LBi + 1) * (long)(b.UBj - b.LBj + 1);
}

static BcArgs bc_args(int var, double *X, int nk)
{
  BcArgs a{};
  a.X = X; a.var = var; a.nk = nk;
  a.masked = g_ctx.p.masking != 0;
  for (int sd = 0; sd < 4; sd++) a.code[sd] = var >= 0 ? lbc_code(g_ctx.p, sd, var) : LBC_GRADIENT;
  if (var < 0) a.masked = 0;
  return a;
}

// bc_r2d / bc_u2d / bc_v2d / bc_w3d_tile ... (bc_2d.F, bc_3d.F): the generic conditions of derived fields --
// where variable `lbv` is closed: zero normal velocity, gamma2 * (inner value) for the tangential one (times the
// mask under MASKING), zero gradient for rho-type fields; on any other physical edge zero gradient without mask;
// then the corners.  Expressed through the edge kernel: closed stays closed, everything else becomes "gradient";
// gtype_var = LBV_ZETA / LBV_UBAR / LBV_VBAR selects the point type.
int bc_generic(int gtype_var, int lbv, double *A, int nk)
{
  BcArgs a = bc_args(gtype_var, A, nk);
  for (int sd = 0; sd < 4; sd++) a.code[sd] = lbc_code(g_ctx.p, sd, lbv) == LBC_CLOSED ? LBC_CLOSED : LBC_GRADIENT;
  // bc_2d.F multiplies by the mask only in the closed tangential branch; the edge kernel masks every branch but the
  // closed normal one: identical for closed edges, and a gradient edge copies an already masked inner value ...
  // except that the copy is not re-masked in the reference -- keep that: no mask on gradient edges
  a.masked = g_ctx.p.masking != 0 ? 2 : 0;          // 2 = "closed tangential branch only"
  return edge_bc(a);
}

// the time level `know` and the step dt2d of the 2-D conditions (zetabc.F:96-106)
static void bc_know(const roms_step_idx_t *s, int *know, double *dt2d)
{
  const double dtfast = g_ctx.p.dtfast;
  if (s->iif == 1) { *know = s->krhs; *dt2d = dtfast; }
  else if (s->predictor_2d_step) { *know = s->krhs; *dt2d = 2.0 * dtfast; }
  else { *know = s->kstp; *dt2d = dtfast; }
}

// boundary data of the free surface are "acquired" on a side whose free-surface condition is clamped or nudged or whose
// ubar / vbar condition is Flather or Shchepetkin (inp_decode.F:1620-1655, no FSOBC_REDUCED)
static void bc_acquire(BcArgs &a)
{
  for (int sd = 0; sd < 4; sd++) {
    const int zc = lbc_code(g_ctx.p, sd, LBV_ZETA), uc = lbc_code(g_ctx.p, sd, LBV_UBAR), vc = lbc_code(g_ctx.p, sd, LBV_VBAR);
    a.acquire[sd] = zc == LBC_CLAMPED || zc == LBC_RADIATION_NUDGING || uc == LBC_FLATHER || uc == LBC_SHCHEPETKIN ||
                    vc == LBC_FLATHER || vc == LBC_SHCHEPETKIN;
  }
}

static bool needs_know(const BcArgs &a)
{
  for (int sd = 0; sd < 4; sd++)
    if (a.code[sd] == LBC_RADIATION || a.code[sd] == LBC_RADIATION_NUDGING || a.code[sd] == LBC_FLATHER ||
        a.code[sd] == LBC_CHAPMAN_IMPLICIT || a.code[sd] == LBC_CHAPMAN_EXPLICIT || a.code[sd] == LBC_SHCHEPETKIN ||
        a.code[sd] == LBC_REDUCED)
      return true;
  return false;
}

int bc_zeta(int kout, const roms_step_idx_t *s)
{
  BcArgs a = bc_args(LBV_ZETA, g_ctx.dev[FID_zeta] + (long)(kout - 1) * nij_host(), 1);
  int know = kout; double dt2d = 0.0;
  if (s) bc_know(s, &know, &dt2d);
  else if (needs_know(a)) return roms_fail("bc_zeta", "this condition needs the barotropic time indices");
  a.Xold = g_ctx.dev[FID_zeta] + (long)(know - 1) * nij_host();
  a.D = g_ctx.dev[FID_zeta_bry];
  a.dt2d = dt2d;
  return edge_bc(a);
}
int bc_u2d(int kout, const roms_step_idx_t *s)
{
  BcArgs a = bc_args(LBV_UBAR, g_ctx.dev[FID_ubar] + (long)(kout - 1) * nij_host(), 1);
  int know = kout; double dt2d = 0.0;
  if (s) bc_know(s, &know, &dt2d);
  else if (needs_know(a)) return roms_fail("bc_u2d", "this condition needs the barotropic time indices");
  a.Xold = g_ctx.dev[FID_ubar] + (long)(know - 1) * nij_host();
  a.D = g_ctx.dev[FID_ubar_bry];
  a.Z = g_ctx.dev[FID_zeta] + (long)(know - 1) * nij_host();
  a.Zb = g_ctx.dev[FID_zeta_bry];
  a.Zn = g_ctx.dev[FID_zeta] + (long)(kout - 1) * nij_host();
  a.T = g_ctx.dev[FID_vbar] + (long)(know - 1) * nij_host();
  bc_acquire(a);
  a.dt2d = dt2d;
  return edge_bc(a);
}
int bc_v2d(int kout, const roms_step_idx_t *s)
{
  BcArgs a = bc_args(LBV_VBAR, g_ctx.dev[FID_vbar] + (long)(kout - 1) * nij_host(), 1);
  int know = kout; double dt2d = 0.0;
  if (s) bc_know(s, &know, &dt2d);
  else if (needs_know(a)) return roms_fail("bc_v2d", "this condition needs the barotropic time indices");
  a.Xold = g_ctx.dev[FID_vbar] + (long)(know - 1) * nij_host();
  a.D = g_ctx.dev[FID_vbar_bry];
  a.Zn = g_ctx.dev[FID_zeta] + (long)(kout - 1) * nij_host();
  a.Z = g_ctx.dev[FID_zeta] + (long)(know - 1) * nij_host();
  a.Zb = g_ctx.dev[FID_zeta_bry];
  a.T = g_ctx.dev[FID_ubar] + (long)(know - 1) * nij_host();
  bc_acquire(a);
  a.dt2d = dt2d;
  return edge_bc(a);
}
int bc_u3d(int nout, int nstp)
{
  const long n3r = nij_host() * g_ctx.b.N;
  BcArgs a = bc_args(LBV_U, g_ctx.dev[FID_u] + (long)(nout - 1) * n3r, g_ctx.b.N);
  a.Xold = g_ctx.dev[FID_u] + (long)(nstp - 1) * n3r;
  a.D = g_ctx.dev[FID_u_bry];
  return edge_bc(a);
}
int bc_v3d(int nout, int nstp)
{
  const long n3r = nij_host() * g_ctx.b.N;
  BcArgs a = bc_args(LBV_V, g_ctx.dev[FID_v] + (long)(nout - 1) * n3r, g_ctx.b.N);
  a.Xold = g_ctx.dev[FID_v] + (long)(nstp - 1) * n3r;
  a.D = g_ctx.dev[FID_v_bry];
  return edge_bc(a);
}
int bc_t3d(int nout, int itrc, int nstp)
{
  const long n3r = nij_host() * g_ctx.b.N;
  BcArgs a = bc_args(LBV_T, g_ctx.dev[FID_t] + ((long)(nout - 1) + 3L * (itrc - 1)) * n3r, g_ctx.b.N);
  a.Xold = g_ctx.dev[FID_t] + ((long)(nstp - 1) + 3L * (itrc - 1)) * n3r;
  a.D = g_ctx.dev[FID_t_bry] + (long)(itrc - 1) * n3r;
  return edge_bc(a);
}
int bc_w3d(double *A)
{
  int rc = edge_bc(bc_args(-1, A, g_ctx.b.N + 1));
  if (rc) return rc;
  return halo_exchange3d(GT_R, g_ctx.b.N + 1, A);
}

// ---------------------------------------------------------------------------
// set_massflux_tile -- ROMS/Nonlinear/set_massflux.F:73-188
//   Huon = 0.5 (Hz(i)+Hz(i-1)) u on_u ,  Hvom = 0.5 (Hz(j)+Hz(j-1)) v om_v
// 5 field passes (read Hz,u,v; write Huon,Hvom): 40 B per cell.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_set_massflux(const RomsDev *__restrict__ c, int nrhs)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrT + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrT + XB.y * BLK_Y + threadIdx.y;
  const int k = XB.z + 1;
  if (i > b.IendT || j > b.JendT) return;
  const gcd_t Hz = (gcd_t)(c->F.Hz);
  const gcd_t u = (gcd_t)(c->F.u + (long)(nrhs - 1) * n3r);
  const gcd_t v = (gcd_t)(c->F.v + (long)(nrhs - 1) * n3r);
  const double hz = Hz[I3(i, j, k)];
  if (i >= b.IstrP)
    GF(Huon)[I3(i, j, k)] = 0.5 * (hz + Hz[I3(i - 1, j, k)]) * u[I3(i, j, k)] * GF(on_u)[I2(i, j)];
  if (j >= b.JstrP)
    GF(Hvom)[I3(i, j, k)] = 0.5 * (hz + Hz[I3(i, j - 1, k)]) * v[I3(i, j, k)] * GF(om_v)[I2(i, j)];
}

extern "C" int roms_hip_set_massflux(const roms_step_idx_t *s)
{
  int rc = roms_entry_check("roms_hip_set_massflux");
  if (rc) return rc;
  ScopedTimer tm("set_massflux");
  const roms_bounds_t &b = g_ctx.b;
  dim3 grid = grid2d(b.IendT - b.IstrT + 1, b.JendT - b.JstrT + 1);
  grid.z = b.N;
  hipLaunchKernelGGL(k_set_massflux, grid, block2d(), 0, g_ctx.stream, g_ctx.devc, s->nrhs);
  KERNEL_CHECK("k_set_massflux");
  halo_batch_begin();
  halo_exchange3d(GT_U, b.N, g_ctx.dev[FID_Huon]);
  halo_exchange3d(GT_V, b.N, g_ctx.dev[FID_Hvom]);
  return halo_batch_end();
}

// ---------------------------------------------------------------------------
// omega_tile -- ROMS/Nonlinear/omega.F:73-229.  One thread per column:
// bottom-up integration of the flux divergence, then removal of the
// moving-surface part.  The running W column stays in registers (template on
// N) so that W is written exactly once: read Huon,Hvom,z_w + write W.
// ---------------------------------------------------------------------------
template <int NMAX>
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_omega(const RomsDev *__restrict__ c)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.Istr + XB.x * BLK_X + threadIdx.x;
  const int j = b.Jstr + XB.y * BLK_Y + threadIdx.y;
  if (i > b.Iend || j > b.Jend) return;
  const gcd_t Huon = (gcd_t)(c->F.Huon);
  const gcd_t Hvom = (gcd_t)(c->F.Hvom);
  const gcd_t z_w = (gcd_t)(c->F.z_w);
  const gd_t W = (gd_t)(c->F.W);
  double w[NMAX + 1];
  w[0] = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k <= N) {
      w[k] = w[k - 1] - (Huon[I3(i + 1, j, k)] - Huon[I3(i, j, k)] +
                         Hvom[I3(i, j + 1, k)] - Hvom[I3(i, j, k)]);
    }
  }
  const double zw0 = z_w[I3W(i, j, 0)];
  double wN = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) if (k == N) wN = w[k];
  const double wrk = wN / (z_w[I3W(i, j, N)] - zw0);
  W[I3W(i, j, 0)] = 0.0;
#pragma unroll
  for (int k = 1; k <= NMAX; k++) {
    if (k < N) W[I3W(i, j, k)] = w[k] - wrk * (z_w[I3W(i, j, k)] - zw0);
  }
  W[I3W(i, j, N)] = 0.0;
}

// LwSrc, omega.F:165-190: W of a column with a cell-centred source, recomputed with the source's Qsrc(k) added to the
// divergence and normalised like every other column.  One thread per source; of several sources in one cell the last
// one's Qsrc alone enters W (the loop of the reference overwrites) -- the cell map names it.
__global__ void k_src_omega(const RomsDev *__restrict__ c)
{
  DEV_PROLOGUE(c)
  const int is = blockIdx.x * blockDim.x + threadIdx.x;
  if (is >= c->src.n || c->src.D[is] != 2) return;
  const int i = c->src.I[is], j = c->src.J[is];
  if (i < b.Istr || i > b.Iend || j < b.Jstr || j > b.Jend) return;
  if (c->src.wmap[I2(i, j)] != is + 1) return;
  const double *__restrict__ Huon = c->F.Huon, *__restrict__ Hvom = c->F.Hvom, *__restrict__ z_w = c->F.z_w;
  double *__restrict__ W = c->F.W;
  double w = 0.0;
  for (int k = 1; k <= N; k++) {
    w = w - (Huon[I3(i + 1, j, k)] - Huon[I3(i, j, k)] + Hvom[I3(i, j + 1, k)] - Hvom[I3(i, j, k)]) +
        c->src.Qsrc[is + (long)c->src.n * (k - 1)];
    W[I3W(i, j, k)] = w;
  }
  const double zw0 = z_w[I3W(i, j, 0)];
  const double wrk = w / (z_w[I3W(i, j, N)] - zw0);
  for (int k = N - 1; k >= 1; k--) W[I3W(i, j, k)] = W[I3W(i, j, k)] - wrk * (z_w[I3W(i, j, k)] - zw0);
  W[I3W(i, j, N)] = 0.0;
}

extern "C" int roms_hip_omega(const roms_step_idx_t *s)
{
  (void)s;
  int rc = roms_entry_check("roms_hip_omega");
  if (rc) return rc;
  if ((rc = check_lbc())) return rc;
  ScopedTimer tm("omega");
  const roms_bounds_t &b = g_ctx.b;
  dim3 grid = grid2d(b.Iend - b.Istr + 1, b.Jend - b.Jstr + 1);
  if (b.N <= 16) hipLaunchKernelGGL(k_omega<16>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc);
  else if (b.N <= 32) hipLaunchKernelGGL(k_omega<32>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc);
  else hipLaunchKernelGGL(k_omega<ROMS_MAXN>, grid, block2d(), 0, g_ctx.stream, g_ctx.devc);
  KERNEL_CHECK("k_omega");
  if ((g_ctx.p.point_sources & 2) && g_ctx.hostc.src.n > 0) {
    hipLaunchKernelGGL(k_src_omega, dim3((g_ctx.hostc.src.n + 63) / 64), dim3(64), 0, g_ctx.stream, g_ctx.devc);
    KERNEL_CHECK("k_src_omega");
  }
  return bc_w3d(g_ctx.dev[FID_W]);
}

// ---------------------------------------------------------------------------
// set_zeta_tile -- ROMS/Nonlinear/set_zeta.F:59-129
// ---------------------------------------------------------------------------
__global__ void k_set_zeta(const RomsDev *__restrict__ c)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrR + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrR + XB.y * BLK_Y + threadIdx.y;
  if (i > b.IendR || j > b.JendR) return;
  const double z = GF(Zt_avg1)[I2(i, j)];
  GF(zeta)[I2(i, j)] = z;
  GF(zeta)[I2(i, j) + nij] = z;
}

extern "C" int roms_hip_set_zeta(const roms_step_idx_t *s)
{
  (void)s;
  int rc = roms_entry_check("roms_hip_set_zeta");
  if (rc) return rc;
  ScopedTimer tm("set_zeta");
  const roms_bounds_t &b = g_ctx.b;
  hipLaunchKernelGGL(k_set_zeta, grid2d(b.IendR - b.IstrR + 1, b.JendR - b.JstrR + 1), block2d(), 0,
                     g_ctx.stream, g_ctx.devc);
  KERNEL_CHECK("k_set_zeta");
  // both time levels travel in one exchange (mp_exchange2d Nvar=2, set_zeta.F:118)
  return halo_exchange3d(GT_R, 2, g_ctx.dev[FID_zeta]);
}

// ---------------------------------------------------------------------------
// set_depth_tile -- ROMS/Nonlinear/set_depth.F:82-300.  One thread per column;
// z_w(k-1) is carried in a register.  Writes z_w, z_r, Hz: 24 B per cell.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_X *BLK_Y)
k_set_depth(const RomsDev *__restrict__ c)
{
  DEV_PROLOGUE(c)
  const Blk XB = xcd_block();
  const int i = b.IstrT + XB.x * BLK_X + threadIdx.x;
  const int j = b.JstrT + XB.y * BLK_Y + threadIdx.y;
  if (i > b.IendT || j > b.JendT) return;
  const roms_params_t &p = c->p;
  const double hc = p.hc;
  double hwater = GF(h)[I2(i, j)];
  if (p.wet_dry && hwater == 0.0) {                 // WET_DRY, set_depth.F:168-172 / :216-220: h itself is changed
    hwater = 1.0E-14;
    GF(h)[I2(i, j)] = hwater;
  }
  const double zt = GF(Zt_avg1)[I2(i, j)];
  const gd_t z_w = (gd_t)(c->F.z_w);
  const gd_t z_r = (gd_t)(c->F.z_r);
  const gd_t Hz = (gd_t)(c->F.Hz);
  double zw_prev = -hwater;
  z_w[I3W(i, j, 0)] = zw_prev;
  if (p.Vtransform == 1) {
    const double hinv = 1.0 / hwater;
    for (int k = 1; k <= N; k++) {
      const double cff_r = hc * (p.sc_r[k] - p.Cs_r[k]);
      const double cff_w = hc * (p.sc_w[k] - p.Cs_w[k]);
      const double z_w0 = cff_w + p.Cs_w[k] * hwater;
      const double zw = z_w0 + zt * (1.0 + z_w0 * hinv);
      const double z_r0 = cff_r + p.Cs_r[k] * hwater;
      z_w[I3W(i, j, k)] = zw;
      z_r[I3(i, j, k)] = z_r0 + zt * (1.0 + z_r0 * hinv);
      Hz[I3(i, j, k)] = zw - zw_prev;
      zw_prev = zw;
    }
  } else {
    const double hinv = 1.0 / (hc + hwater);
    for (int k = 1; k <= N; k++) {
      const double cff_r = hc * p.sc_r[k];
      const double cff_w = hc * p.sc_w[k];
      const double cff2_r = (cff_r + p.Cs_r[k] * hwater) * hinv;
      const double cff2_w = (cff_w + p.Cs_w[k] * hwater) * hinv;
      const double zw = zt + (zt + hwater) * cff2_w;
      z_w[I3W(i, j, k)] = zw;
      z_r[I3(i, j, k)] = zt + (zt + hwater) * cff2_r;
      Hz[I3(i, j, k)] = zw - zw_prev;
      zw_prev = zw;
    }
  }
}

extern "C" int roms_hip_set_depth(const roms_step_idx_t *s)
{
  (void)s;
  int rc = roms_entry_check("roms_hip_set_depth");
  if (rc) return rc;
  ScopedTimer tm("set_depth");
  const roms_bounds_t &b = g_ctx.b;
  hipLaunchKernelGGL(k_set_depth, grid2d(b.IendT - b.IstrT + 1, b.JendT - b.JstrT + 1), block2d(), 0,
                     g_ctx.stream, g_ctx.devc);
  KERNEL_CHECK("k_set_depth");
  halo_batch_begin();
  halo_exchange2d(GT_R, g_ctx.dev[FID_h]);
  halo_exchange3d(GT_R, b.N + 1, g_ctx.dev[FID_z_w]);
  halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_z_r]);
  halo_exchange3d(GT_R, b.N, g_ctx.dev[FID_Hz]);
  return halo_batch_end();
}
