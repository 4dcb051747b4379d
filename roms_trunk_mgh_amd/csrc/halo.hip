// halo.hip -- ghost-point updates: the reference's exchange_*_tile periodic
// copies (ROMS/Nonlinear/exchange_2d.F:43-788, exchange_3d.F:43-1108) and
// mp_exchange2d/3d/4d (ROMS/Utility/mp_exchange.F:290/1413/2753) re-done for
// one-tile-per-GPU: device pack kernel -> grouped RCCL ncclSend/ncclRecv over
// xGMI -> device unpack kernel, all on the library stream (no host copy, no
// host synchronisation).
//
// Semantics kept from the reference (mp_exchange.F:73-286, tile_neighbors):
//   * the reference runs two dependent phases, W/E first, then S/N over the FULL i-range including
//     the just-received ghost columns, so that corners ride along in phase 2.  Here ONE phase with up
//     to eight messages (W, E, S, N and the four diagonal tiles) leaves the same values in every ghost
//     point -- a corner block is the diagonal tile's interior either way -- at half the latency, which
//     is what bounds a 4x2 run (about 70 exchanges per step);
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
static double *g_buf[2] = {nullptr, nullptr};     // all outgoing messages, all incoming messages (device)
static double *g_hbuf[2] = {nullptr, nullptr};    // pinned host mirrors (relay transport)
static size_t g_buf_doubles = 0, g_hbuf_doubles = 0;

// Host-relay transport: the packed messages are handed to a host callback that moves them with
// whatever the host application already has (the reference's own MPI, or gloo in the tests) -- the
// same pack/unpack kernels, neighbour table and message list as the RCCL transport, with one
// device<->pinned-host copy on either side.  (typedefs: include/roms_hip.h)
static bool g_have_plan = false;              // message plan of this tile (built at the first exchange)
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
  for (auto &p : g_buf) { if (p) (void)hipFree(p); p = nullptr; }
  for (auto &p : g_hbuf) { if (p) (void)hipHostFree(p); p = nullptr; }
  g_buf_doubles = g_hbuf_doubles = 0;
  g_have_neigh = false;
  g_have_plan = false;
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
// A message = a rectangle [i0,i0+wi) x [j0,j0+wj) of every plane of every field of the batch.
// Buffer order per message: plane kk (field after field), then row, then column -- ours, not MPI's.
#define HALO_MAX_ITEMS 8
#define HALO_MAX_MSGS 8
struct PackArgs {
  double *A[HALO_MAX_ITEMS];
  int koff[HALO_MAX_ITEMS];
  int n, nktot, nmsg, unpack;
  int i0[HALO_MAX_MSGS], j0[HALO_MAX_MSGS], wi[HALO_MAX_MSGS], wj[HALO_MAX_MSGS];
  long off[HALO_MAX_MSGS];      // start of message m in buf (doubles)
  double *buf;
  int LBi, LBj;                 // array origin and pitches: in the kernel arguments, so that the kernel does not
  long ni, nij;                 // wait for a load from the constant block before it can form its first address
};
// All fields of a batch and all messages of an exchange in ONE launch (a step on several tiles issues
// ~70 exchanges; with a launch per field and side the host could not feed the GPU).
__global__ void k_pack_multi(const RomsDev *__restrict__, PackArgs a)
{
  const int LBi = a.LBi, LBj = a.LBj;
  const long ni = a.ni, nij = a.nij;
  const int m = blockIdx.z;
  const int wi = a.wi[m], cnt = wi * a.wj[m];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int kk = blockIdx.y;
  if (r >= cnt) return;
  int f = 0;
  while (f + 1 < a.n && kk >= a.koff[f + 1]) f++;
  const int k = kk - a.koff[f];
  const int li = r % wi, lj = r / wi;
  const long idx = I2(a.i0[m] + li, a.j0[m] + lj) + (long)k * nij;
  const long q = a.off[m] + (long)kk * cnt + r;
  if (a.unpack) a.A[f][idx] = a.buf[q];
  else a.buf[q] = a.A[f][idx];
}

static int ensure_buffers(size_t doubles)
{
  if (doubles <= g_buf_doubles) return 0;
  // a failed allocation must leave no dangling pointer and no stale size behind (halo_finalize frees them)
  g_buf_doubles = 0;
  for (auto &p : g_buf) {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  for (auto &p : g_buf) HIP_TRY(hipMalloc(&p, sizeof(double) * doubles));
  g_buf_doubles = doubles;
  return 0;
}

// Directions: 0 W, 1 E, 2 S, 3 N, 4 SW, 5 SE, 6 NW, 7 NE.  A message carries the direction code of its
// SENDER as tag; the receive that matches a send with tag d is posted by the tile in direction d of the
// sender, i.e. from its own direction opp(d).
struct HaloMsg { int peer, tag, i0, wi, j0, wj; };
struct HaloPlan { int nsend, nrecv; HaloMsg send[HALO_MAX_MSGS], recv[HALO_MAX_MSGS]; };
static HaloPlan g_plan;

static HaloPlan make_plan(const roms_bounds_t &b, const Neigh &n)
{
  auto tile = [&](int i, int j) { return j * b.ntileI + i; };
  // column / row index of the neighbour tiles (with the periodic wrap), -1 = none
  const int iW = n.Wtile >= 0 ? n.Wtile % b.ntileI : -1, iE = n.Etile >= 0 ? n.Etile % b.ntileI : -1;
  const int jS = n.Stile >= 0 ? n.Stile / b.ntileI : -1, jN = n.Ntile >= 0 ? n.Ntile / b.ntileI : -1;
  const bool hW = iW >= 0, hE = iE >= 0, hS = jS >= 0, hN = jN >= 0;
  // strips stop at the tile's own range on a side that has a neighbour (the corner block comes from the
  // diagonal tile) and run to the edge of the array where there is none (wall rows, locally copied columns)
  const int ilo = hW ? b.Istr : b.LBi, ihi = hE ? b.Iend : b.UBi;
  const int jlo = hS ? b.Jstr : b.LBj, jhi = hN ? b.Jend : b.UBj;
  // send / receive extents per side
  const int sW0 = b.Istr, sWn = n.GsendW, sE0 = b.Iend - n.GsendE + 1, sEn = n.GsendE;
  const int rW0 = b.Istr - n.GrecvW, rWn = n.GrecvW, rE0 = b.Iend + 1, rEn = n.GrecvE;
  const int sS0 = b.Jstr, sSn = n.GsendS, sN0 = b.Jend - n.GsendN + 1, sNn = n.GsendN;
  const int rS0 = b.Jstr - n.GrecvS, rSn = n.GrecvS, rN0 = b.Jend + 1, rNn = n.GrecvN;
  HaloPlan p;
  p.nsend = p.nrecv = 0;
  auto add = [&](int d, int peer, int si0, int swi, int sj0, int swj, int ri0, int rwi, int rj0, int rwj) {
    if (peer < 0) return;
    p.send[p.nsend++] = HaloMsg{peer, d, si0, swi, sj0, swj};
    // what I receive FROM direction d was sent by that tile towards its opposite direction
    static const int opp[8] = {1, 0, 3, 2, 7, 6, 5, 4};
    p.recv[p.nrecv++] = HaloMsg{peer, opp[d], ri0, rwi, rj0, rwj};
  };
  add(0, n.Wtile, sW0, sWn, jlo, jhi - jlo + 1, rW0, rWn, jlo, jhi - jlo + 1);
  add(1, n.Etile, sE0, sEn, jlo, jhi - jlo + 1, rE0, rEn, jlo, jhi - jlo + 1);
  add(2, n.Stile, ilo, ihi - ilo + 1, sS0, sSn, ilo, ihi - ilo + 1, rS0, rSn);
  add(3, n.Ntile, ilo, ihi - ilo + 1, sN0, sNn, ilo, ihi - ilo + 1, rN0, rNn);
  add(4, (hW && hS) ? tile(iW, jS) : -1, sW0, sWn, sS0, sSn, rW0, rWn, rS0, rSn);
  add(5, (hE && hS) ? tile(iE, jS) : -1, sE0, sEn, sS0, sSn, rE0, rEn, rS0, rSn);
  add(6, (hW && hN) ? tile(iW, jN) : -1, sW0, sWn, sN0, sNn, rW0, rWn, rN0, rNn);
  add(7, (hE && hN) ? tile(iE, jN) : -1, sE0, sEn, sN0, sNn, rE0, rEn, rN0, rNn);
  // RCCL pairs the k-th send to a peer with the k-th receive from it: sends go out in the order of their
  // tag, receives are posted in the order of the tag they expect -- the same order on both ends of a pair
  // (two tiles in a periodic direction are each other's W and E neighbour, and twice diagonal)
  for (int x = 0; x < p.nrecv; x++)
    for (int y = x + 1; y < p.nrecv; y++)
      if (p.recv[y].tag < p.recv[x].tag) { HaloMsg t = p.recv[x]; p.recv[x] = p.recv[y]; p.recv[y] = t; }
  return p;
}

// Host-only: the message plan of tile `rank` for the bounds *b (no GPU needed), exported so that the CPU
// tests can play a whole 4x2 exchange with numpy.  out = nsend, nrecv, then 6 ints per message (peer, tag,
// i0, wi, j0, wj), sends first.
extern "C" int roms_hip_halo_plan(const roms_bounds_t *b, int rank, int *out)
{
  int v[12];
  roms_hip_tile_neighbors(rank, b->ntileI, b->ntileJ, b->NghostPoints, b->NghostPoints, b->EWperiodic, b->NSperiodic, v);
  const Neigh n{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11]};
  const HaloPlan p = make_plan(*b, n);
  int q = 0;
  out[q++] = p.nsend;
  out[q++] = p.nrecv;
  for (int m = 0; m < p.nsend; m++) {
    const HaloMsg &x = p.send[m];
    out[q++] = x.peer; out[q++] = x.tag; out[q++] = x.i0; out[q++] = x.wi; out[q++] = x.j0; out[q++] = x.wj;
  }
  for (int m = 0; m < p.nrecv; m++) {
    const HaloMsg &x = p.recv[m];
    out[q++] = x.peer; out[q++] = x.tag; out[q++] = x.i0; out[q++] = x.wi; out[q++] = x.j0; out[q++] = x.wj;
  }
  return 0;
}

static void ensure_neigh();

// hipGraph capture of a loop with exchanges inside (k_step2d.hip): possible with the RCCL transport only (the relay
// synchronises the stream), and the message buffers must exist before the capture starts.
bool halo_rccl_capturable() { return g_ctx.nccl_comm != nullptr && g_relay == nullptr; }
int halo_reserve_buffers()
{
  ensure_neigh();
  if (!g_have_plan) { g_plan = make_plan(g_ctx.b, g_neigh); g_have_plan = true; }
  long smax = 0, rmax = 0;
  for (int m = 0; m < g_plan.nsend; m++) smax += (long)g_plan.send[m].wi * g_plan.send[m].wj;
  for (int m = 0; m < g_plan.nrecv; m++) rmax += (long)g_plan.recv[m].wi * g_plan.recv[m].wj;
  const long planes = (long)HALO_MAX_ITEMS * (g_ctx.b.N + 1);       // the largest batch: eight 3-D W-type fields
  return ensure_buffers((size_t)((smax > rmax ? smax : rmax) * planes));
}

// One exchange = a list of fields (each nk planes) whose ghost points travel in ONE message per
// neighbour: every RCCL group costs a fixed latency, and the barotropic loop alone issues
// dozens of exchanges per step.
struct HaloItem { int gtype, nk; double *A; };

static int exchange_all(const HaloItem *items, int nitems)
{
  if (!g_have_plan) { g_plan = make_plan(g_ctx.b, g_neigh); g_have_plan = true; }
  const HaloPlan &pl = g_plan;
  if (pl.nsend == 0 && pl.nrecv == 0) return 0;
  long nktot = 0;
  for (int f = 0; f < nitems; f++) nktot += items[f].nk;
  PackArgs pa;
  pa.n = nitems; pa.nktot = (int)nktot;
  pa.LBi = g_ctx.b.LBi; pa.LBj = g_ctx.b.LBj;
  pa.ni = g_ctx.b.UBi - g_ctx.b.LBi + 1;
  pa.nij = pa.ni * (long)(g_ctx.b.UBj - g_ctx.b.LBj + 1);
  {
    int koff = 0;
    for (int f = 0; f < nitems; f++) { pa.A[f] = items[f].A; pa.koff[f] = koff; koff += items[f].nk; }
  }
  long soff[HALO_MAX_MSGS], roff[HALO_MAX_MSGS], stot = 0, rtot = 0;
  int smax = 1, rmax = 1;
  for (int m = 0; m < pl.nsend; m++) {
    soff[m] = stot;
    const int cnt = pl.send[m].wi * pl.send[m].wj;
    stot += nktot * cnt;
    smax = cnt > smax ? cnt : smax;
  }
  for (int m = 0; m < pl.nrecv; m++) {
    roff[m] = rtot;
    const int cnt = pl.recv[m].wi * pl.recv[m].wj;
    rtot += nktot * cnt;
    rmax = cnt > rmax ? cnt : rmax;
  }
  int rc = ensure_buffers((size_t)(stot > rtot ? stot : rtot));
  if (rc) return rc;
  const dim3 blk(256);
  auto launch = [&](const HaloMsg *msgs, int nmsg, const long *off, int cmax, double *buf, int unpack) {
    pa.nmsg = nmsg; pa.unpack = unpack; pa.buf = buf;
    for (int m = 0; m < nmsg; m++) {
      pa.i0[m] = msgs[m].i0; pa.j0[m] = msgs[m].j0; pa.wi[m] = msgs[m].wi; pa.wj[m] = msgs[m].wj; pa.off[m] = off[m];
    }
    if (nmsg == 0) return;
    dim3 grid((cmax + 255) / 256, (unsigned)nktot, (unsigned)nmsg);
    hipLaunchKernelGGL(k_pack_multi, grid, blk, 0, g_ctx.stream, g_ctx.devc, pa);
  };
  launch(pl.send, pl.nsend, soff, smax, g_buf[0], 0);
  KERNEL_CHECK("k_pack");
  if (!g_ctx.nccl_comm) {
    if (!g_relay)
      return roms_fail("halo exchange", "multi-tile run without a transport: pass an RCCL unique id to "
                                        "roms_hip_init or set a host relay (roms_hip_set_halo_relay)");
    const size_t need = (size_t)(stot > rtot ? stot : rtot);
    if (need > g_hbuf_doubles) {
      g_hbuf_doubles = 0;
      for (auto &p : g_hbuf) {
        if (p) (void)hipHostFree(p);
        p = nullptr;
      }
      for (auto &p : g_hbuf) HIP_TRY(hipHostMalloc(&p, sizeof(double) * need, hipHostMallocDefault));
      g_hbuf_doubles = need;
    }
    if (stot) HIP_TRY(hipMemcpyAsync(g_hbuf[0], g_buf[0], sizeof(double) * stot, hipMemcpyDeviceToHost, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    roms_halo_msg_t sm[HALO_MAX_MSGS], rm[HALO_MAX_MSGS];
    for (int m = 0; m < pl.nsend; m++)
      sm[m] = roms_halo_msg_t{pl.send[m].peer, pl.send[m].tag, nktot * pl.send[m].wi * pl.send[m].wj, g_hbuf[0] + soff[m]};
    for (int m = 0; m < pl.nrecv; m++)
      rm[m] = roms_halo_msg_t{pl.recv[m].peer, pl.recv[m].tag, nktot * pl.recv[m].wi * pl.recv[m].wj, g_hbuf[1] + roff[m]};
    const int rrc = g_relay(g_relay_user, pl.nsend, sm, pl.nrecv, rm);
    if (rrc) return roms_fail("halo exchange", "host relay callback failed");
    if (rtot) HIP_TRY(hipMemcpyAsync(g_buf[1], g_hbuf[1], sizeof(double) * rtot, hipMemcpyHostToDevice, g_ctx.stream));
  } else {
    rccl_comm_t comm = (rccl_comm_t)g_ctx.nccl_comm;
    RCCL_TRY(rccl.gstart());
    for (int m = 0; m < pl.nsend; m++)
      RCCL_TRY(rccl.send(g_buf[0] + soff[m], (size_t)(nktot * pl.send[m].wi * pl.send[m].wj), RCCL_FLOAT64,
                         pl.send[m].peer, comm, g_ctx.stream));
    for (int m = 0; m < pl.nrecv; m++)
      RCCL_TRY(rccl.recv(g_buf[1] + roff[m], (size_t)(nktot * pl.recv[m].wi * pl.recv[m].wj), RCCL_FLOAT64,
                         pl.recv[m].peer, comm, g_ctx.stream));
    RCCL_TRY(rccl.gend());
  }
  launch(pl.recv, pl.nrecv, roff, rmax, g_buf[1], 1);
  KERNEL_CHECK("k_unpack");
  return 0;
}

static void ensure_neigh()
{
  const roms_bounds_t &b = g_ctx.b;
  if (!g_have_neigh) {
    int v[12];
    roms_hip_tile_neighbors(g_ctx.rank, b.ntileI, b.ntileJ, b.NghostPoints, b.NghostPoints,
                            b.EWperiodic, b.NSperiodic, v);
    g_neigh = Neigh{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11]};
    if (g_ctx.loopback && b.EWperiodic) {
      // the tile is its own W and E neighbour; it is west-most and east-most at once, so both halves of
      // the Nghost+1 rule (mp_exchange.F:155-187) apply
      g_neigh.Wtile = g_neigh.Etile = g_ctx.rank;
      if (b.NghostPoints != 3) g_neigh.GrecvW = g_neigh.GsendE = b.NghostPoints + 1;
    }
    g_have_neigh = true;
  }
}

static int halo_run(const HaloItem *items, int nitems)
{
  const roms_bounds_t &b = g_ctx.b;
  if (nitems <= 0) return 0;
  if (b.ntileI * b.ntileJ == 1 && !g_ctx.loopback) {
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
  ensure_neigh();
  // a periodic direction held by ONE tile row/column is a local copy
  if (b.EWperiodic && b.ntileI == 1 && !g_ctx.loopback) {
    for (int f = 0; f < nitems; f++) {
      dim3 grid((b.UBj - b.LBj + 1 + 63) / 64, items[f].nk);
      hipLaunchKernelGGL(k_periodic_ew, grid, dim3(64), 0, g_ctx.stream, g_ctx.devc, items[f].A, items[f].nk, b.LBj, b.UBj);
    }
    KERNEL_CHECK("k_periodic_ew");
  }
  for (int f0 = 0; f0 < nitems; f0 += HALO_MAX_ITEMS) {
    const int n = nitems - f0 < HALO_MAX_ITEMS ? nitems - f0 : HALO_MAX_ITEMS;
    const int rc = exchange_all(items + f0, n);
    if (rc) return rc;
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
