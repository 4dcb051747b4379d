// roms_dev.h -- internal header of libroms_hip.so (gfx950 / CDNA4 only).
//
// One process drives one GPU and one ROMS tile.  All module arrays of the tile
// live in HBM for the whole run (device mirrors of mod_ocean/mod_grid/
// mod_coupling/mod_mixing/mod_forces, same column-major i-fastest layout and
// the same LBi:UBi,LBj:UBj extents as the host arrays), so host and device
// index arithmetic is shared.  Kernels receive a pointer to one device-resident
// constant block (bounds + parameters + field pointers); wave-uniform reads of
// it compile to scalar loads.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include "roms_hip.h"

// The point-source table SOURCES(ng) (mod_sources.F:56-80) on the device, LuvSrc only (roms_hip_set_sources).  The two
// face maps (nij ints: 1 + the index of the source at the u- / v-face of the point, 0 elsewhere; the last source of a
// face wins, as the sequential loops of the reference leave it) let a kernel find "its" source without a search.
#define ROMS_NWS3 10   // 3-D scratch arrays (the _tile routines' automatic arrays; MPDATA holds Ta of three tracers at a time)

struct RomsSrc {
  int n;                      // Nsrc; 0 = no table
  int ltr[ROMS_MAXNT];        // LtracerSrc(itrc)
  const int *I, *J, *D;       // [n] Isrc, Jsrc, INT(Dsrc)
  const double *Qbar;         // [n]
  const double *Qsrc;         // [n * N]       is + n * (k-1)
  const double *Tsrc;         // [n * N * NT]  is + n * ((k-1) + N * (itrc-1))
  const int *umap, *vmap;     // [nij]  (LuvSrc; all zero without)
  const int *wmap;            // [nij]  LwSrc: 1 + the index of the last cell-centred source (Dsrc = 2) of the point
  // the interior cells (Istr:Iend, Jstr:Jend) with a source face or a cell-centred source, as indices into the 2-D
  // arrays: the software-pipelined tracer kernel leaves them to its SRC instantiation, which runs over this list
  const int *cells;
  int ncell;
  double *save;               // [n * N] the mass fluxes of the source faces across k_uv_column (k_step3d_uv.hip)
};

struct RomsDev {
  roms_bounds_t b;
  roms_params_t p;
  roms_fields_t F;      // device pointers
  // device scratch = the _tile routines' automatic work arrays
  double *ws3[ROMS_NWS3];   // 3-D scratch, each nij*(N+1) doubles
  double *ws2[32];      // 2-D scratch, each nij doubles
  const double *rowm;   // row table of the i-uniform metric arrays (k_step2d_mom.hip), or nullptr
  RomsSrc src;          // point sources (LuvSrc), n = 0 without
};

struct RomsCtx {
  bool inited = false;
  int rank = 0, ntileI = 1, ntileJ = 1, device = 0;
  hipStream_t stream = nullptr;
  roms_bounds_t b{};
  roms_params_t p{};
  bool have_bounds = false, have_params = false;
  double *host[FID_COUNT] = {nullptr};
  double *dev[FID_COUNT] = {nullptr};
  double *dev_base[FID_COUNT] = {nullptr};   // start of the allocation (guard band in front of dev[])
  long count[FID_COUNT] = {0};
  double *ws3_base[ROMS_NWS3] = {nullptr};
  double *ws2_base[32] = {nullptr};
  long guard = 0;               // doubles of guard band on either side of every mirror / scratch array
  RomsDev hostc{};            // host copy of the constant block
  RomsDev *devc = nullptr;    // device copy
  bool devc_dirty = true;
  bool timing = false;
  std::string last_error;
  // halo exchange (RCCL) state lives in halo.hip
  void *nccl_comm = nullptr;
  unsigned char nccl_id[128];
  bool have_nccl_id = false;
  // ONE tile with an RCCL id: the periodic wrap of that tile travels through the transport to the tile
  // itself (W and E neighbour = own rank) instead of the local copy kernel, and every kernel takes its
  // multi-tile branch.  Same results; lets a one-GPU box execute the RCCL leg (tests/test_gpu_rccl.py).
  bool loopback = false;
  // row-uniform metrics (k_step2d_mom.hip): 0 = not examined since the last upload of a metric array,
  // 1 = all fifteen arrays are independent of i (table valid), 2 = not
  int rowm_state = 0;
  bool rowh = false;             // ... and h, visc2_r, visc2_p as well (second group of the table)
  double *rowm_dev = nullptr;
  long rowm_nj = 0;
};

extern RomsCtx g_ctx;

int  roms_fail(const char *where, const char *what);
int  roms_flush_consts();                 // upload hostc -> devc when dirty
long roms_field_count(int kind, const roms_bounds_t &b);

#define HIP_TRY(expr)                                                         \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) return roms_fail(#expr, hipGetErrorString(e_));     \
  } while (0)

#define KERNEL_CHECK(name)                                                    \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) return roms_fail(name, hipGetErrorString(e_));      \
  } while (0)

// ------------------------------------------------------------------------
// timing (hipEvents on the library stream) for bench.py's roofline object
// ------------------------------------------------------------------------
struct ScopedTimer {
  const char *name;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  explicit ScopedTimer(const char *n);
  ~ScopedTimer();
};

// ------------------------------------------------------------------------
// device-side index helpers
// ------------------------------------------------------------------------
// XCD-aware renumbering of a 2-D / 3-D grid, inside the kernel (the launch is unchanged): hardware
// deals workgroups to the 8 XCDs round-robin in linear order, so linear id L runs on XCD L % 8.  When
// the grid has a multiple of 8 tile columns each XCD is given a contiguous strip of columns, walked
// row by row with the z-index (component / level) fastest -- x-neighbours then share their boundary
// cache lines in one L2 and z-neighbours share the fields both read (see decode_tile_tracer).  A
// bijection on the same grid; placement only affects speed.
struct Blk { int x, y, z; };
#ifdef __HIPCC__
__device__ __forceinline__ Blk xcd_block()
{
  Blk r{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  const int nbx = gridDim.x, nby = gridDim.y, nz = gridDim.z;
  if ((nbx & 7) == 0) {
    const int L = r.x + nbx * (r.y + nby * r.z);
    const int xcd = L & 7, q = L >> 3, w = nbx >> 3;
    r.z = q % nz;
    const int t = q / nz;
    r.x = xcd * w + t % w;
    r.y = t / w;
  }
  return r;
}
#endif

#define DEV_PROLOGUE(c)                                                        \
  const roms_bounds_t &b = (c)->b;                                             \
  const int LBi = b.LBi, LBj = b.LBj;                                          \
  const int N = b.N;                                                           \
  const long ni = b.UBi - b.LBi + 1, nj = b.UBj - b.LBj + 1, nij = ni * nj;    \
  const long n3r = nij * N, n3w = nij * (N + 1);                               \
  (void)nj; (void)n3r; (void)n3w;

// Field pointers are read from the constant block, so the compiler only knows them as
// generic (flat) addresses.  Casting to the global address space turns flat_load/store
// into global_load/store: no LDS/scratch aperture check, and the loads stop counting
// against lgkmcnt, so LDS traffic and scalar loads no longer wait on them.
typedef const double __attribute__((address_space(1))) *gcd_t;
typedef double __attribute__((address_space(1))) *gd_t;
// field `name` of the constant block as a global-address-space pointer (device code with `c` in scope)
#define GF(name) ((gd_t)c->F.name)

// WET_DRY: the land/sea mask of a point times its wet/dry mask.  Wherever the reference multiplies by the wet/dry mask
// directly after the land/sea mask (x = x*umask; x = x*umask_wet -- every WET_DRY block of the mixing, bulk-flux,
// MPDATA, step3d_uv and ini_fields files) the two small-integer factors combine exactly, signed zeros included:
// x*(m*w) == (x*m)*w bit for bit for m in {0,1}, w in {-1,0,1,2}.  Callers hold `c`; without WET_DRY this is the
// land/sea mask alone.
#define ROMS_MASKW(name)                                                                                  \
  __device__ __forceinline__ double name##w(const RomsDev *__restrict__ c, long q)                       \
  {                                                                                                       \
    double m = ((gcd_t)c->F.name)[q];                                                                     \
    if (c->p.wet_dry) m = m * ((gcd_t)c->F.name##_wet)[q];                                                \
    return m;                                                                                             \
  }
ROMS_MASKW(rmask)
ROMS_MASKW(umask)
ROMS_MASKW(vmask)
ROMS_MASKW(pmask)
#undef ROMS_MASKW
// the factor of the barotropic wet/dry rule (step2d_LF_AM3.h:2124-2126, u2dbc_im.F:1183-1187): 1 on a face between
// two wet cells (mask 2), 0 between two dry ones, and on a one-sided face (+-1) 1 only for flow out of the wet cell
__device__ __forceinline__ double wet_factor(double mask_wet, double vel)
{
  const double cff5 = fabs(fabs(mask_wet) - 1.0);
  const double cff6 = 0.5 + copysign(0.5, vel) * mask_wet;
  return 0.5 * mask_wet * cff5 + cff6 * (1.0 - cff5);
}

// LuvSrc: the horizontal advective tracer fluxes of the four faces of cell c0 at level k (ck = its 3-D index), replaced
// at source faces.  PRE = false: step3d_t.F:734-799 (Huon * Tsrc; without LtracerSrc and under MASKING the upstream value
// of the wet side, T = t(:,:,:,3,itrc)); PRE = true: pre_step3d.F:530-553 (Huon * Tsrc, or zero without LtracerSrc).
// The caller tests c->src.n (uniform) and src_cell_any() first: the body is the rare path.
__device__ __forceinline__ bool src_cell_any(const RomsDev *__restrict__ c, long c0, long ni)
{
  return (c->src.umap[c0] | c->src.umap[c0 + 1] | c->src.vmap[c0] | c->src.vmap[c0 + ni] | c->src.wmap[c0]) != 0;
}
// LwSrc, step3d_t.F:1136-1158 / :1331-1360: the tracer that comes with the volume of the cell-centred sources of point
// c0 at level k, added to tv (Ta before its vertical advection for MPDATA, t(nnew) after it otherwise); cff =
// dt * pm * pn (times oHz outside MPDATA), t3k = t(i,j,k,3,itrc), the value the inflow carries without LtracerSrc.
// Every source of the cell counts, in the order of the table (the sums of the reference's loop).
__device__ __forceinline__ double src_w_tracer(const RomsDev *__restrict__ c, long c0, int k, int itrc, double cff,
                                               double t3k, double tv)
{
  const RomsSrc &S = c->src;
  if (S.wmap[c0] == 0) return tv;
  const long ni = c->b.UBi - c->b.LBi + 1;
  for (int is = 0; is < S.n; is++) {
    if (S.D[is] != 2 || (long)(S.I[is] - c->b.LBi) + (long)(S.J[is] - c->b.LBj) * ni != c0) continue;
    const double cff3 = S.ltr[itrc - 1] ? S.Tsrc[is + (long)S.n * ((k - 1) + (long)c->b.N * (itrc - 1))] : t3k;
    tv = tv + cff * S.Qsrc[is + (long)S.n * (k - 1)] * cff3;
  }
  return tv;
}
template <bool PRE>
__device__ __forceinline__ void src_cell_fluxes(const RomsDev *__restrict__ c, long c0, long ck, long ni, int k, int itrc,
                                                const double *__restrict__ T, double &FXi, double &FXip1, double &FEj,
                                                double &FEjp1)
{
  const RomsSrc &S = c->src;
  const bool ltr = S.ltr[itrc - 1] != 0;
  const int N = c->b.N;
  auto face = [&](int m, long cf, long ckf, long off, const double *__restrict__ H, double &Fv) {
    if (!m) return;
    const int is = m - 1;
    if (ltr) Fv = H[ckf] * S.Tsrc[is + (long)S.n * ((k - 1) + (long)N * (itrc - 1))];
    else if (PRE) Fv = 0.0;
    else if (c->p.masking) {
      const double r1 = c->F.rmask[cf], r0 = c->F.rmask[cf - off];
      if (r1 == 0.0 && r0 == 1.0) Fv = H[ckf] * T[ckf - off];
      else if (r1 == 1.0 && r0 == 0.0) Fv = H[ckf] * T[ckf];
    }
  };
  face(S.umap[c0], c0, ck, 1, c->F.Huon, FXi);
  face(S.umap[c0 + 1], c0 + 1, ck + 1, 1, c->F.Huon, FXip1);
  face(S.vmap[c0], c0, ck, ni, c->F.Hvom, FEj);
  face(S.vmap[c0 + ni], c0 + ni, ck + ni, ni, c->F.Hvom, FEjp1);
}

#define I2(i,j)    ((long)((i) - LBi) + (long)((j) - LBj) * ni)
#define I3(i,j,k)  (I2(i,j) + (long)((k) - 1) * nij)
#define I3W(i,j,k) (I2(i,j) + (long)(k) * nij)

// grid-point type codes for the periodic exchange (exchange_2d.F / _3d.F)
enum { GT_R = 0, GT_U, GT_V, GT_P };

// thread (tx,ty) -> horizontal index; blocks are 64 x 4 so that one wavefront
// covers 64 consecutive i (one 512-byte line per k-level access).
#define BLK_X 64
#define BLK_Y 4
static inline dim3 grid2d(int nx, int ny) {
  return dim3((unsigned)((nx + BLK_X - 1) / BLK_X), (unsigned)((ny + BLK_Y - 1) / BLK_Y), 1);
}
static inline dim3 block2d() { return dim3(BLK_X, BLK_Y, 1); }
// (tile, level) decode of a 1-D grid for kernels with one thread per cell: consecutive workgroups of an XCD take the
// SAME horizontal tile at consecutive levels, so the planes k-1, k, k+1 a vertical stencil reads meet in one L2
// (with the level in gridDim.z they are thousands of workgroups apart and on other XCDs); tiles are dealt to the
// XCDs round-robin, which balances any tile count.  Launch with grid_tile_level(); check .valid.
struct TileLv { int bx, by, k0; bool valid; };
#ifdef __HIPCC__
__device__ __forceinline__ TileLv decode_tile_level(int nx, int ny, int nz)
{
  const int nbx = (nx + BLK_X - 1) / BLK_X, nby = (ny + BLK_Y - 1) / BLK_Y;
  const int B = blockIdx.x, xcd = B & 7, q = B >> 3;
  TileLv r;
  r.k0 = q % nz;
  const int tl = (q / nz) * 8 + xcd;
  r.valid = tl < nbx * nby;
  r.bx = tl % nbx;
  r.by = tl / nbx;
  return r;
}
#endif
static inline dim3 grid_tile_level(int nx, int ny, int nz) {
  const int nbx = (nx + BLK_X - 1) / BLK_X, nby = (ny + BLK_Y - 1) / BLK_Y;
  return dim3((unsigned)(((nbx * nby + 7) / 8) * 8 * nz), 1, 1);
}

// XCD-aware (tile, tracer) decode of a 1-D grid.  Workgroups B and B+8 run on the same
// XCD (round-robin dispatch over the 8 XCDs, MI355X_MICROARCH.md); each XCD has its own
// 4 MB L2.  Two things are arranged through the order in which one XCD meets its tiles:
//  * consecutive workgroups of an XCD take the SAME horizontal tile for consecutive
//    tracers, so the tracer-independent fields (Huon, Hvom, W, Hz, z_r ...) one of them
//    streams are L2 hits for the other (measured on step3d_t: 2.9 -> 2.1 GB fetched);
//  * each XCD owns a contiguous strip of tile columns [x0,x1) and walks it row by row.
//    A 64-double row segment is not 128-B aligned (the reference's row pitch is odd), so
//    it shares its first and last cache line with the x-neighbours; inside a strip those
//    lines are fetched once per XCD instead of once per tile.
// With fewer than 8 tile columns the strips degenerate and tiles are dealt round-robin.
// Placement only affects speed, never results.
struct TileTr { int bx, by, itr; bool valid; };
#ifdef __HIPCC__
__device__ __forceinline__ TileTr decode_tile_tracer(int nx, int ny, int ntr, int ty = BLK_Y)
{
  const int nbx = (nx + BLK_X - 1) / BLK_X, nby = (ny + ty - 1) / ty;
  const int B = blockIdx.x, xcd = B & 7, q = B >> 3;
  TileTr r;
  r.itr = q % ntr;
  const int t = q / ntr;
  if (nbx < 8) {
    const int tl = t * 8 + xcd;
    r.valid = tl < nbx * nby;
    r.bx = tl % nbx;
    r.by = tl / nbx;
  } else {
    const int x0 = (xcd * nbx) >> 3, w = (((xcd + 1) * nbx) >> 3) - x0;
    r.bx = x0 + t % w;
    r.by = t / w;
    r.valid = r.by < nby;
  }
  return r;
}
#endif
static inline dim3 grid_tile_tracer(int nx, int ny, int ntr, int ty = BLK_Y) {
  const int nbx = (nx + BLK_X - 1) / BLK_X, nby = (ny + ty - 1) / ty;
  if (nbx < 8) return dim3((unsigned)(((nbx * nby + 7) / 8) * 8 * ntr), 1, 1);
  return dim3((unsigned)(8 * ((nbx + 7) / 8) * nby * ntr), 1, 1);
}

// ------------------------------------------------------------------------
// internal launchers shared between translation units
// ------------------------------------------------------------------------
int halo_exchange2d(int gtype, double *A, int nfields_stride_unused = 0);
int halo_exchange3d(int gtype, int nk, double *A);
void halo_batch_begin();                  // record the exchanges that follow ...
int halo_batch_end();                     // ... and run them as one message per neighbour and phase
// lateral boundary conditions on the S/N edges (k_base.hip); s = the barotropic time indices (Chapman, Flather),
// nstp = the time level the radiation condition compares with
int bc_zeta(int kout, const roms_step_idx_t *s);
int bc_u2d(int kout, const roms_step_idx_t *s);
int bc_v2d(int kout, const roms_step_idx_t *s);
int bc_u3d(int nout, int nstp);
int bc_v3d(int nout, int nstp);
int bc_t3d(int nout, int itrc, int nstp);
bool lbc2d_all_closed();
int lbc_code(const roms_params_t &p, int sd, int v);   // effective enum roms_lbc of variable v on side sd
int bc_w3d(double *A);
int bc_generic(int gtype_var, int lbv, double *A, int nk);   // bc_2d.F / bc_3d.F rules for derived fields
void snapshot_release();                  // snapshot.hip: waits for and frees an in-flight snapshot
void snapshot_forget(int field_id);       // snapshot.hip: the same for one field (before it is re-registered)
void diag_release();                      // k_diag.hip: frees the buffers of roms_hip_diag
int check_lbc();
int roms_rowm_prepare();               // k_step2d_mom.hip: examine the metric arrays if needed (synchronises once)
void roms_rowm_invalidate();            // a metric array may have changed
bool roms_rowm_is_table_field(int id);  // is field `id` one of the arrays the row table is built from?
void roms_rowm_release();
void step2d_graphs_release();           // k_step2d.hip: drop the captured LOOP_2D graphs (their launch arguments are stale)
