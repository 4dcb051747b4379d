// halo.hip -- ghost-point updates: the reference's exchange_*_tile periodic
// copies (ROMS/Nonlinear/exchange_2d.F:43-788, exchange_3d.F:43-1108) and
// mp_exchange2d/3d/4d (ROMS/Utility/mp_exchange.F:290/1413/2753) re-done for
// one-tile-per-GPU: device pack kernel -> grouped RCCL ncclSend/ncclRecv over
// xGMI -> device unpack kernel, all on the library stream (no host copy, no
// host synchronisation).
//
// Semantics kept from the reference (mp_exchange.F:73-286, tile_neighbors):
//   * two dependent phases, W/E first then S/N over the FULL i-range including
//     the just-received ghost columns, so corners ride along in phase 2;
//   * rank = Jtile*NtileI + Itile; periodic directions wrap to the far tile;
//   * with 2 ghost points and periodicity the west-most tile receives
//     Nghost+1 columns from the east-most tile (which sends Nghost+1).
// RCCL is resolved at run time (dlsym in the already-loaded process image, then
// dlopen of librccl.so) so that a Fortran host and a Python host share one build.
#include "roms_dev.h"
#include <dlfcn.h>
#include <vector>

// ------------------------------------------------------------ RCCL binding --
typedef struct { char internal[128]; } rccl_uid_t;
typedef void *rccl_comm_t;
typedef int (*fn_getuid)(rccl_uid_t *);
typedef int (*fn_initrank)(rccl_comm_t *, int, rccl_uid_t, int);
typedef int (*fn_destroy)(rccl_comm_t);
typedef int (*fn_send)(const void *, size_t, int, int, rccl_comm_t, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, rccl_comm_t, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_errstr)(int);
static struct {
  bool loaded = false;
  fn_getuid getuid = nullptr;
  fn_initrank initrank = nullptr;
  fn_destroy destroy = nullptr;
  fn_send send = nullptr;
  fn_recv recv = nullptr;
  fn_group gstart = nullptr, gend = nullptr;
  fn_errstr errstr = nullptr;
} rccl;
static const int RCCL_FLOAT64 = 8;   // ncclFloat64 / ncclDouble

static int rccl_load()
{
  if (rccl.loaded) return 0;
  void *h = RTLD_DEFAULT;
  if (!dlsym(h, "ncclCommInitRank")) {
    const char *cands[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
    h = nullptr;
    for (const char *c : cands) { h = dlopen(c, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return roms_fail("rccl_load", "librccl.so not found");
  }
  rccl.getuid = (fn_getuid)dlsym(h, "ncclGetUniqueId");
  rccl.initrank = (fn_initrank)dlsym(h, "ncclCommInitRank");
  rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  rccl.send = (fn_send)dlsym(h, "ncclSend");
  rccl.recv = (fn_recv)dlsym(h, "ncclRecv");
  rccl.gstart = (fn_group)dlsym(h, "ncclGroupStart");
  rccl.gend = (fn_group)dlsym(h, "ncclGroupEnd");
  rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  if (!rccl.getuid || !rccl.initrank || !rccl.send || !rccl.recv || !rccl.gstart || !rccl.gend)
    return roms_fail("rccl_load", "RCCL symbols missing");
  rccl.loaded = true;
  return 0;
}

static int rccl_fail(const char *where, int code)
{
  g_ctx.last_error = std::string(where) + ": " + (rccl.errstr ? rccl.errstr(code) : "RCCL error");
  return 2;   // ROMS exit_flag 2, as mp_exchange.F:551
}
#define RCCL_TRY(expr) do { int r_ = (expr); if (r_ != 0) return rccl_fail(#expr, r_); } while (0)

extern "C" int roms_hip_get_unique_id(void *out128)
{
  int rc = rccl_load();
  if (rc) return rc;
  rccl_uid_t id;
  RCCL_TRY(rccl.getuid(&id));
  memcpy(out128, &id, 128);
  return 0;
}

// ---------------------------------------------------------- neighbour table --
struct Neigh {
  int Wtile, Etile, Stile, Ntile;              // -1 = none
  int GsendW, GsendE, GrecvW, GrecvE;
  int GsendS, GsendN, GrecvS, GrecvN;
};

// mp_exchange.F:73-286 (tile_neighbors).  Pure host function, exported so the
// CPU tests can check it against the Python mirror without a GPU.
extern "C" int roms_hip_tile_neighbors(int rank, int ntileI, int ntileJ, int Nghost, int NghostPoints,
                                       int EWperiodic, int NSperiodic, int *out12)
{
  const int I = rank % ntileI, J = rank / ntileI;
  auto table = [&](int i, int j) { return (i < 0 || i >= ntileI || j < 0 || j >= ntileJ) ? -1 : j * ntileI + i; };
  Neigh n;
  n.GsendW = n.GsendE = n.GrecvW = n.GrecvE = Nghost;
  n.GsendS = n.GsendN = n.GrecvS = n.GrecvN = Nghost;
  n.Wtile = table(I - 1, J);
  n.Etile = table(I + 1, J);
  if (EWperiodic && ntileI > 1) {
    if (table(I - 1, J) < 0) { n.Wtile = table(ntileI - 1, J); if (NghostPoints != 3) n.GrecvW = Nghost + 1; }
    else if (table(I + 1, J) < 0) { n.Etile = table(0, J); if (NghostPoints != 3) n.GsendE = Nghost + 1; }
  }
  n.Stile = table(I, J - 1);
  n.Ntile = table(I, J + 1);
  if (NSperiodic && ntileJ > 1) {
    if (table(I, J - 1) < 0) { n.Stile = table(I, ntileJ - 1); if (NghostPoints != 3) n.GrecvS = Nghost + 1; }
    else if (table(I, J + 1) < 0) { n.Ntile = table(I, 0); if (NghostPoints != 3) n.GsendN = Nghost + 1; }
  }
  const int v[12] = {n.Wtile, n.Etile, n.Stile, n.Ntile, n.GsendW, n.GsendE, n.GrecvW, n.GrecvE,
                     n.GsendS, n.GsendN, n.GrecvS, n.GrecvN};
  memcpy(out12, v, sizeof v);
  return 0;
}

static Neigh g_neigh;
static bool g_have_neigh = false;
static double *g_buf[4] = {nullptr, nullptr, nullptr, nullptr};   // sendLo, sendHi, recvLo, recvHi
static double *g_hbuf[4] = {nullptr, nullptr, nullptr, nullptr};  // pinned host mirrors (relay transport)
static size_t g_buf_doubles = 0, g_hbuf_doubles = 0;

// Host-relay transport: the packed ghost lines are handed to a host callback that moves
// them with whatever the host application already has (the reference's own MPI, or gloo
// in the tests) -- the same pack/unpack kernels, neighbour table and phase order as the
// RCCL transport, with a device<->pinned-host copy on either side.
typedef int (*roms_halo_relay_fn)(void *user, int dir, int lo_rank, int hi_rank,
                                  const double *send_lo, long n_send_lo, const double *send_hi, long n_send_hi,
                                  double *recv_lo, long n_recv_lo, double *recv_hi, long n_recv_hi);
static roms_halo_relay_fn g_relay = nullptr;
static void *g_relay_user = nullptr;
extern "C" int roms_hip_set_halo_relay(roms_halo_relay_fn fn, void *user)
{
  g_relay = fn;
  g_relay_user = user;
  return 0;
}

int halo_init()
{
  int rc = rccl_load();
  if (rc) return rc;
  rccl_uid_t id;
  memcpy(&id, g_ctx.nccl_id, 128);
  rccl_comm_t comm = nullptr;
  RCCL_TRY(rccl.initrank(&comm, g_ctx.ntileI * g_ctx.ntileJ, id, g_ctx.rank));
  g_ctx.nccl_comm = comm;
  return 0;
}

int halo_finalize()
{
  for (auto &p : g_buf) { if (p) hipFree(p); p = nullptr; }
  for (auto &p : g_hbuf) { if (p) hipHostFree(p); p = nullptr; }
  g_buf_doubles = g_hbuf_doubles = 0;
  g_have_neigh = false;
  g_relay = nullptr;
  g_relay_user = nullptr;
  if (g_ctx.nccl_comm && rccl.destroy) rccl.destroy((rccl_comm_t)g_ctx.nccl_comm);
  g_ctx.nccl_comm = nullptr;
  return 0;
}

// ------------------------------------------------ single-tile periodic copy --
// exchange_2d.F:229-414 / exchange_3d.F:259-470: A(Lm+1:Lm+Ng)=A(1:Ng),
// A(-2:0)=A(Lm-2:Lm); v- and psi-type start at j=Jstr, rho/u-type at JstrR.
__global__ void k_periodic_ew(const RomsDev *__restrict__ c, double *__restrict__ A, int nk, int jmin, int jmax)
{
  DEV_PROLOGUE(c)
  const int j = jmin + blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (j > jmax || k >= nk) return;
  const int Lm = b.Lm;
  double *P = A + (long)k * nij;
  for (int m = 1; m <= b.NghostPoints; m++) P[I2(Lm + m, j)] = P[I2(m, j)];
  for (int m = 0; m <= 2; m++) P[I2(-m, j)] = P[I2(Lm - m, j)];
}

// the same copy for several fields at once: blockIdx.y runs over all planes of all fields
struct PeriodicArgs {
  double *A[8];
  int koff[8], jmin[8], jmax[8];
  int n, nktot, jlo;
};
__global__ void k_periodic_multi(const RomsDev *__restrict__ c, PeriodicArgs a)
{
  DEV_PROLOGUE(c)
  const int j = a.jlo + blockIdx.x * blockDim.x + threadIdx.x;
  const int kk = blockIdx.y;
  int f = 0;
  while (f + 1 < a.n && kk >= a.koff[f + 1]) f++;
  if (kk >= a.nktot || j < a.jmin[f] || j > a.jmax[f]) return;
  const int Lm = b.Lm;
  double *P = a.A[f] + (long)(kk - a.koff[f]) * nij;
  for (int m = 1; m <= b.NghostPoints; m++) P[I2(Lm + m, j)] = P[I2(m, j)];
  for (int m = 0; m <= 2; m++) P[I2(-m, j)] = P[I2(Lm - m, j)];
}

// --------------------------------------------------- multi-tile pack/unpack --
// dir 0: columns i0..i0+G-1 over the full j-range; dir 1: rows j0..j0+G-1 over
// the full i-range.  Buffer order (k, m, running index) -- ours, not MPI's.
__global__ void k_pack(const RomsDev *__restrict__ c, const double *__restrict__ A, double *__restrict__ buf,
                       int nk, int dir, int start, int G, int unpack)
{
  DEV_PROLOGUE(c)
  const int len = dir == 0 ? (int)nj : (int)ni;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y % G, k = blockIdx.y / G;
  if (r >= len || k >= nk) return;
  const long a = dir == 0 ? I2(start + m, LBj + r) : I2(LBi + r, start + m);
  const long q = ((long)k * G + m) * len + r;
  if (unpack) const_cast<double *>(A)[a + (long)k * nij] = buf[q];
  else buf[q] = A[a + (long)k * nij];
}

// All fields of a batch and both sides of a phase in ONE launch (a step on several tiles issues
// ~70 exchanges; with a launch per field, side and direction the host could not feed the GPU).
#define HALO_MAX_ITEMS 8
struct PackArgs {
  double *A[HALO_MAX_ITEMS];
  int nk[HALO_MAX_ITEMS], koff[HALO_MAX_ITEMS];
  int n, nktot, dir, len, Gmax, unpack;
  int start[2], G[2];          // side 0 = low neighbour, 1 = high neighbour; G = 0: side absent
  double *buf[2];
};
__global__ void k_pack_multi(const RomsDev *__restrict__ c, PackArgs a)
{
  DEV_PROLOGUE(c)
  const int side = blockIdx.z;
  const int G = a.G[side];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y % a.Gmax, kk = blockIdx.y / a.Gmax;
  if (r >= a.len || m >= G || kk >= a.nktot) return;
  int f = 0;
  while (f + 1 < a.n && kk >= a.koff[f + 1]) f++;
  const int k = kk - a.koff[f];
  const int start = a.start[side];
  const long idx = (a.dir == 0 ? I2(start + m, LBj + r) : I2(LBi + r, start + m)) + (long)k * nij;
  const long q = ((long)kk * G + m) * a.len + r;
  if (a.unpack) a.A[f][idx] = a.buf[side][q];
  else a.buf[side][q] = a.A[f][idx];
}

static int ensure_buffers(size_t doubles)
{
  if (doubles <= g_buf_doubles) return 0;
  for (auto &p : g_buf) {
    if (p) hipFree(p);
    HIP_TRY(hipMalloc(&p, sizeof(double) * doubles));
  }
  g_buf_doubles = doubles;
  return 0;
}

// One exchange = a list of fields (each nk planes) whose ghost lines travel in ONE message per
// neighbour and phase: every RCCL group costs a fixed latency, and the barotropic loop alone issues
// hundreds of exchanges per step.  Buffer layout per side: field after field, each (k, m, r).
struct HaloItem { int gtype, nk; double *A; };

static int exchange_phase(const HaloItem *items, int nitems, int dir)
{
  const roms_bounds_t &b = g_ctx.b;
  const Neigh &n = g_neigh;
  const int lo = dir == 0 ? n.Wtile : n.Stile, hi = dir == 0 ? n.Etile : n.Ntile;
  if (lo < 0 && hi < 0) return 0;
  const int GsLo = dir == 0 ? n.GsendW : n.GsendS, GsHi = dir == 0 ? n.GsendE : n.GsendN;
  const int GrLo = dir == 0 ? n.GrecvW : n.GrecvS, GrHi = dir == 0 ? n.GrecvE : n.GrecvN;
  const int len = dir == 0 ? (b.UBj - b.LBj + 1) : (b.UBi - b.LBi + 1);
  const int str = dir == 0 ? b.Istr : b.Jstr, end = dir == 0 ? b.Iend : b.Jend;
  const int Gmax = b.NghostPoints + 1;
  long nktot = 0;
  for (int f = 0; f < nitems; f++) nktot += items[f].nk;
  int rc = ensure_buffers((size_t)nktot * Gmax * len);
  if (rc) return rc;
  const dim3 blk(256);
  PackArgs pa;
  pa.n = nitems; pa.nktot = (int)nktot; pa.dir = dir; pa.len = len; pa.Gmax = Gmax;
  {
    int koff = 0;
    for (int f = 0; f < nitems; f++) { pa.A[f] = items[f].A; pa.nk[f] = items[f].nk; pa.koff[f] = koff; koff += items[f].nk; }
  }
  auto launch_sides = [&](double *blo, int slo, int glo, double *bhi, int shi, int ghi, int unpack) {
    pa.buf[0] = blo; pa.start[0] = slo; pa.G[0] = lo >= 0 ? glo : 0;
    pa.buf[1] = bhi; pa.start[1] = shi; pa.G[1] = hi >= 0 ? ghi : 0;
    pa.unpack = unpack;
    dim3 grid((len + 255) / 256, (unsigned)(nktot * Gmax), 2);
    hipLaunchKernelGGL(k_pack_multi, grid, blk, 0, g_ctx.stream, g_ctx.devc, pa);
  };
  // my first GsLo / last GsHi interior lines
  launch_sides(g_buf[0], str, GsLo, g_buf[1], end - GsHi + 1, GsHi, 0);
  KERNEL_CHECK("k_pack");
  const long nsl = nktot * GsLo * len, nsh = nktot * GsHi * len;
  const long nrl = nktot * GrLo * len, nrh = nktot * GrHi * len;
  if (!g_ctx.nccl_comm) {
    if (!g_relay)
      return roms_fail("halo exchange", "multi-tile run without a transport: pass an RCCL unique id to "
                                        "roms_hip_init or set a host relay (roms_hip_set_halo_relay)");
    const size_t need = (size_t)nktot * Gmax * len;
    if (need > g_hbuf_doubles) {
      for (auto &p : g_hbuf) {
        if (p) hipHostFree(p);
        HIP_TRY(hipHostMalloc(&p, sizeof(double) * need, hipHostMallocDefault));
      }
      g_hbuf_doubles = need;
    }
    if (lo >= 0) HIP_TRY(hipMemcpyAsync(g_hbuf[0], g_buf[0], sizeof(double) * nsl, hipMemcpyDeviceToHost, g_ctx.stream));
    if (hi >= 0) HIP_TRY(hipMemcpyAsync(g_hbuf[1], g_buf[1], sizeof(double) * nsh, hipMemcpyDeviceToHost, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    const int rrc = g_relay(g_relay_user, dir, lo, hi, g_hbuf[0], lo >= 0 ? nsl : 0, g_hbuf[1], hi >= 0 ? nsh : 0,
                            g_hbuf[2], lo >= 0 ? nrl : 0, g_hbuf[3], hi >= 0 ? nrh : 0);
    if (rrc) return roms_fail("halo exchange", "host relay callback failed");
    if (lo >= 0) HIP_TRY(hipMemcpyAsync(g_buf[2], g_hbuf[2], sizeof(double) * nrl, hipMemcpyHostToDevice, g_ctx.stream));
    if (hi >= 0) HIP_TRY(hipMemcpyAsync(g_buf[3], g_hbuf[3], sizeof(double) * nrh, hipMemcpyHostToDevice, g_ctx.stream));
  } else {
    rccl_comm_t comm = (rccl_comm_t)g_ctx.nccl_comm;
    RCCL_TRY(rccl.gstart());
    // order matters when lo == hi (two tiles in a periodic direction): my low-side
    // send pairs with the peer's high-side receive.
    if (lo >= 0) RCCL_TRY(rccl.send(g_buf[0], (size_t)nsl, RCCL_FLOAT64, lo, comm, g_ctx.stream));
    if (hi >= 0) RCCL_TRY(rccl.send(g_buf[1], (size_t)nsh, RCCL_FLOAT64, hi, comm, g_ctx.stream));
    if (hi >= 0) RCCL_TRY(rccl.recv(g_buf[3], (size_t)nrh, RCCL_FLOAT64, hi, comm, g_ctx.stream));
    if (lo >= 0) RCCL_TRY(rccl.recv(g_buf[2], (size_t)nrl, RCCL_FLOAT64, lo, comm, g_ctx.stream));
    RCCL_TRY(rccl.gend());
  }
  launch_sides(g_buf[2], str - GrLo, GrLo, g_buf[3], end + 1, GrHi, 1);
  KERNEL_CHECK("k_unpack");
  return 0;
}

static int halo_run(const HaloItem *items, int nitems)
{
  const roms_bounds_t &b = g_ctx.b;
  if (nitems <= 0) return 0;
  if (b.ntileI * b.ntileJ == 1) {
    if (!b.EWperiodic) return 0;
    // all fields of the batch in one launch (a step issues ~50 of these otherwise)
    for (int f0 = 0; f0 < nitems; f0 += HALO_MAX_ITEMS) {
      PeriodicArgs pa;
      pa.n = nitems - f0 < HALO_MAX_ITEMS ? nitems - f0 : HALO_MAX_ITEMS;
      int nktot = 0, jlo = b.UBj, jhi = b.LBj;
      for (int f = 0; f < pa.n; f++) {
        const HaloItem &it = items[f0 + f];
        int jmin, jmax;
        if (b.NSperiodic) { jmin = b.Jstr; jmax = b.Jend; }
        else { jmin = (it.gtype == GT_R || it.gtype == GT_U) ? b.JstrR : b.Jstr; jmax = b.JendR; }
        pa.A[f] = it.A; pa.koff[f] = nktot; pa.jmin[f] = jmin; pa.jmax[f] = jmax;
        nktot += it.nk;
        jlo = jmin < jlo ? jmin : jlo;
        jhi = jmax > jhi ? jmax : jhi;
      }
      pa.nktot = nktot; pa.jlo = jlo;
      dim3 grid((jhi - jlo + 1 + 63) / 64, nktot);
      hipLaunchKernelGGL(k_periodic_multi, grid, dim3(64), 0, g_ctx.stream, g_ctx.devc, pa);
    }
    KERNEL_CHECK("k_periodic_multi");
    return 0;
  }
  if (!g_have_neigh) {
    int v[12];
    roms_hip_tile_neighbors(g_ctx.rank, b.ntileI, b.ntileJ, b.NghostPoints, b.NghostPoints,
                            b.EWperiodic, b.NSperiodic, v);
    g_neigh = Neigh{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11]};
    g_have_neigh = true;
  }
  // a periodic direction held by ONE tile row/column is a local copy
  if (b.EWperiodic && b.ntileI == 1) {
    for (int f = 0; f < nitems; f++) {
      dim3 grid((b.UBj - b.LBj + 1 + 63) / 64, items[f].nk);
      hipLaunchKernelGGL(k_periodic_ew, grid, dim3(64), 0, g_ctx.stream, g_ctx.devc, items[f].A, items[f].nk, b.LBj, b.UBj);
    }
    KERNEL_CHECK("k_periodic_ew");
  }
  for (int f0 = 0; f0 < nitems; f0 += HALO_MAX_ITEMS) {
    const int n = nitems - f0 < HALO_MAX_ITEMS ? nitems - f0 : HALO_MAX_ITEMS;
    int rc = exchange_phase(items + f0, n, 0);
    if (rc) return rc;
    if ((rc = exchange_phase(items + f0, n, 1))) return rc;
  }
  return 0;
}

// Batching: between halo_batch_begin() and halo_batch_end() the exchange calls only record their
// field; halo_batch_end() moves them all in one message per neighbour and phase.  Callers batch
// exchanges that follow one another with no kernel in between (same result, fewer messages).
static std::vector<HaloItem> g_batch;
static bool g_batching = false;
void halo_batch_begin() { g_batching = true; g_batch.clear(); }
int halo_batch_end()
{
  g_batching = false;
  const int rc = halo_run(g_batch.data(), (int)g_batch.size());
  g_batch.clear();
  return rc;
}

int halo_exchange3d(int gtype, int nk, double *A)
{
  const HaloItem it{gtype, nk, A};
  if (g_batching) { g_batch.push_back(it); return 0; }
  return halo_run(&it, 1);
}

int halo_exchange2d(int gtype, double *A, int) { return halo_exchange3d(gtype, 1, A); }

// Exported for tests: exchange one registered field (all planes, or one
// trailing level when level > 0 and the field has time levels / tracers).
extern "C" int roms_hip_exchange(int field_id, int level)
{
  if (!g_ctx.inited || field_id < 0 || field_id >= FID_COUNT || !g_ctx.dev[field_id])
    return roms_fail("roms_hip_exchange", "field not registered");
  int rc = roms_flush_consts();
  if (rc) return rc;
  const roms_bounds_t &b = g_ctx.b;
  const long nij = (long)(b.UBi - b.LBi + 1) * (long)(b.UBj - b.LBj + 1);
  const int nk = (int)(g_ctx.count[field_id] / nij);
  (void)level;
  return halo_exchange3d(GT_R, nk, g_ctx.dev[field_id]);
}
