// capi.hip -- life cycle, field registry and host<->device copies behind the
// C ABI of include/roms_hip.h.
#include "roms_dev.h"
#include <algorithm>
#include <map>
#include <vector>

RomsCtx g_ctx;

static const int k_field_kind[FID_COUNT] = {
#define ROMS_FIELD(name, kind, owner) kind,
#include "roms_fields.def"
#undef ROMS_FIELD
};
static const char *k_field_name[FID_COUNT] = {
#define ROMS_FIELD(name, kind, owner) #name,
#include "roms_fields.def"
#undef ROMS_FIELD
};

int roms_fail(const char *where, const char *what)
{
  g_ctx.last_error = std::string(where) + ": " + what;
  return 8;   // ROMS exit_flag 8 = "Fatal algorithm result" (mod_scalars.F:523-532)
}

long roms_field_count(int kind, const roms_bounds_t &b)
{
  const long nij = (long)(b.UBi - b.LBi + 1) * (long)(b.UBj - b.LBj + 1);
  switch (kind) {
  case K_2D:      return nij;
  case K_2D_T2:   return nij * 2;
  case K_2D_T3:   return nij * 3;
  case K_2D_NT:   return nij * b.NT;
  case K_3DR:     return nij * b.N;
  case K_3DW:     return nij * (b.N + 1);
  case K_3DR_T2:  return nij * b.N * 2;
  case K_3DW_T2:  return nij * (b.N + 1) * 2;
  case K_3DW_NAT: return nij * (b.N + 1) * b.NAT;
  case K_4DT:     return nij * b.N * 3 * b.NT;
  case K_3DR_NT:  return nij * b.N * b.NT;
  case K_3DW_T3:  return nij * (b.N + 1) * 3;
  }
  return -1;
}

// ---------------------------------------------------------------- timing --
static std::map<std::string, double> g_last_ms;

ScopedTimer::ScopedTimer(const char *n) : name(n)
{
  if (!g_ctx.timing) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
  (void)hipEventRecord(e0, g_ctx.stream);
}
ScopedTimer::~ScopedTimer()
{
  if (!e0) return;
  float ms = 0.f;
  if (hipEventRecord(e1, g_ctx.stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
      hipEventElapsedTime(&ms, e0, e1) == hipSuccess)
    g_last_ms[name] = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

extern "C" int roms_hip_timing_enable(int on) { g_ctx.timing = on != 0; return 0; }
extern "C" double roms_hip_timing_last_ms(const char *entry)
{
  auto it = g_last_ms.find(entry);
  return it == g_last_ms.end() ? -1.0 : it->second;
}

// counter calibration: a streaming copy with the library's one-double-per-lane pattern
__global__ void __launch_bounds__(256) k_calib_stream(const double *__restrict__ src, double *__restrict__ dst, long n)
{
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) dst[e] = src[e];
}
int roms_entry_check(const char *where);
namespace { void sources_release(); }
extern "C" int roms_hip_calib_stream(long n)
{
  int rc = roms_entry_check("roms_hip_calib_stream");
  if (rc) return rc;
  const roms_bounds_t &b = g_ctx.b;
  const long cap = (long)(b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * (b.N + 1);
  if (n < 1 || n > cap) return roms_fail("roms_hip_calib_stream", "n_doubles outside 1..nij*(N+1)");
  ScopedTimer tm("calib_stream");
  hipLaunchKernelGGL(k_calib_stream, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_ctx.stream,
                     (const double *)g_ctx.hostc.ws3[0], g_ctx.hostc.ws3[1], n);
  KERNEL_CHECK("k_calib_stream");
  return 0;
}

// ---------------------------------------------------------- guard bands --
// Every device mirror and every scratch array is allocated with a guard band of four rows (rounded up to
// 256 B) in front and behind, filled with one quiet-NaN bit pattern.  A stencil that reaches a row or a
// column outside LBi:UBi,LBj:UBj at the first or last plane of an array therefore reads mapped memory (no
// page fault, whatever lies next to the allocation) and gets a NaN that shows up in the parity tests if the
// value is used; a store outside an array destroys the pattern, which roms_hip_check_guards reports.
static const unsigned long long k_guard_bits = 0x7FF8C0DEC0DEC0DEull;

__global__ void k_fill_guard(unsigned long long *p, long n, unsigned long long bits)
{
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) p[e] = bits;
}

static int guarded_alloc(double **user, double **base, long n)
{
  const long g = g_ctx.guard;
  *user = *base = nullptr;
  HIP_TRY(hipMalloc(base, sizeof(double) * (n + 2 * g)));
  *user = *base + g;
  if (g > 0) {
    hipLaunchKernelGGL(k_fill_guard, dim3((unsigned)((g + 255) / 256)), dim3(256), 0, g_ctx.stream,
                       (unsigned long long *)*base, g, k_guard_bits);
    hipLaunchKernelGGL(k_fill_guard, dim3((unsigned)((g + 255) / 256)), dim3(256), 0, g_ctx.stream,
                       (unsigned long long *)(*user + n), g, k_guard_bits);
    KERNEL_CHECK("k_fill_guard");
  }
  HIP_TRY(hipMemsetAsync(*user, 0, sizeof(double) * n, g_ctx.stream));
  return 0;
}

static void guarded_free(double **user, double **base)
{
  if (*base) (void)hipFree(*base);
  *user = *base = nullptr;
}

// 0 = every guard band intact; otherwise an error naming the first damaged array (field name, or ws3[q] /
// ws2[q] for scratch) and the distance of the first damaged word from the array.
extern "C" int roms_hip_check_guards(void)
{
  if (!g_ctx.inited || !g_ctx.have_bounds) return roms_fail("roms_hip_check_guards", "set_bounds first");
  const long g = g_ctx.guard;
  if (g <= 0) return 0;
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  std::vector<unsigned long long> h(2 * g);
  const roms_bounds_t &b = g_ctx.b;
  const long nij = (long)(b.UBi - b.LBi + 1) * (long)(b.UBj - b.LBj + 1);
  auto check = [&](const char *name, int q, const double *base, long n) -> int {
    if (!base) return 0;
    HIP_TRY(hipMemcpy(h.data(), base, sizeof(double) * g, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h.data() + g, base + g + n, sizeof(double) * g, hipMemcpyDeviceToHost));
    for (long e = 0; e < 2 * g; e++)
      if (h[e] != k_guard_bits) {
        char msg[200];
        if (q >= 0) snprintf(msg, sizeof msg, "store outside %s[%d]: %ld doubles %s the array", name, q,
                             e < g ? g - e : e - g + 1, e < g ? "before" : "behind");
        else snprintf(msg, sizeof msg, "store outside field %s: %ld doubles %s the array", name,
                      e < g ? g - e : e - g + 1, e < g ? "before" : "behind");
        return roms_fail("roms_hip_check_guards", msg);
      }
    return 0;
  };
  int rc;
  for (int id = 0; id < FID_COUNT; id++)
    if ((rc = check(k_field_name[id], -1, g_ctx.dev_base[id], g_ctx.count[id]))) return rc;
  for (int q = 0; q < ROMS_NWS3; q++)
    if ((rc = check("ws3", q, g_ctx.ws3_base[q], nij * (b.N + 1)))) return rc;
  for (int q = 0; q < 32; q++)
    if ((rc = check("ws2", q, g_ctx.ws2_base[q], nij))) return rc;
  return 0;
}

// ------------------------------------------------------------- life cycle --
extern "C" int roms_abi_sizeof(int which)
{
  switch (which) {
  case 0: return (int)sizeof(roms_bounds_t);
  case 1: return (int)sizeof(roms_params_t);
  case 2: return (int)sizeof(roms_step_idx_t);
  case 3: return (int)sizeof(roms_fields_t);
  case 4: return (int)FID_COUNT;
  }
  return -1;
}

extern "C" const char *roms_hip_last_error(void) { return g_ctx.last_error.c_str(); }

int halo_init();      // halo.hip
int halo_finalize();
extern "C" int roms_hip_finalize(void);

extern "C" int roms_hip_init(int rank, int ntileI, int ntileJ, int device_id, const void *nccl_unique_id)
{
  if (g_ctx.inited) return roms_fail("roms_hip_init", "already initialised");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return roms_fail("roms_hip_init", "no HIP device: the HIP path has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return roms_fail("roms_hip_init", "bad device id");
  HIP_TRY(hipSetDevice(device_id));
  HIP_TRY(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
  g_ctx.rank = rank;
  g_ctx.ntileI = ntileI;
  g_ctx.ntileJ = ntileJ;
  g_ctx.device = device_id;
  g_ctx.have_nccl_id = false;
  if (nccl_unique_id) {
    memcpy(g_ctx.nccl_id, nccl_unique_id, 128);
    g_ctx.have_nccl_id = true;
  }
  if (hipMalloc(&g_ctx.devc, sizeof(RomsDev)) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamDestroy(g_ctx.stream);
    g_ctx.stream = nullptr;
    g_ctx.devc = nullptr;
    return roms_fail("roms_hip_init", "hipMalloc of the constant block failed");
  }
  memset(&g_ctx.hostc, 0, sizeof(RomsDev));
  g_ctx.devc_dirty = true;
  g_ctx.inited = true;
  g_ctx.loopback = false;
  if (g_ctx.have_nccl_id) {
    // without an id the halos of a multi-tile run must go through a host relay
    // (roms_hip_set_halo_relay); the first exchange fails loudly if neither transport exists
    if (ntileI * ntileJ == 1 && rank != 0) { roms_hip_finalize(); return roms_fail("roms_hip_init", "one tile: rank must be 0"); }
    int rc = halo_init();
    if (rc) { roms_hip_finalize(); return rc; }
    g_ctx.loopback = ntileI * ntileJ == 1;
  }
  return 0;
}

extern "C" int roms_hip_finalize(void)
{
  if (!g_ctx.inited) return 0;
  (void)hipStreamSynchronize(g_ctx.stream);
  step2d_graphs_release();
  roms_rowm_release();
  snapshot_release();
  halo_finalize();
  diag_release();
  for (int i = 0; i < FID_COUNT; i++) {
    guarded_free(&g_ctx.dev[i], &g_ctx.dev_base[i]);
    g_ctx.host[i] = nullptr;
    g_ctx.count[i] = 0;
  }
  for (int q = 0; q < ROMS_NWS3; q++) guarded_free(&g_ctx.hostc.ws3[q], &g_ctx.ws3_base[q]);
  for (int q = 0; q < 32; q++) guarded_free(&g_ctx.hostc.ws2[q], &g_ctx.ws2_base[q]);
  sources_release();
  if (g_ctx.devc) (void)hipFree(g_ctx.devc);
  g_ctx.devc = nullptr;
  if (g_ctx.stream) (void)hipStreamDestroy(g_ctx.stream);
  g_ctx.stream = nullptr;
  g_ctx.inited = false;
  g_ctx.have_bounds = g_ctx.have_params = false;
  return 0;
}

// ---- point sources (mod_sources.F), LuvSrc ----
namespace {
struct SrcStore {
  int *geo = nullptr;        // device: I[n], J[n], D[n], umap[nij], vmap[nij]
  double *val = nullptr;     // device: Qbar[n], Qsrc[n*N], Tsrc[n*N*NT]
  int n = 0, N = 0, NT = 0;
  long nij = 0;
  std::vector<int> I, J, D;
  bool given = false;
} g_src;

void sources_release()
{
  if (g_src.geo) (void)hipFree(g_src.geo);
  if (g_src.val) (void)hipFree(g_src.val);
  g_src = SrcStore{};
  g_ctx.hostc.src = RomsSrc{};
  g_ctx.devc_dirty = true;
}
}  // namespace

bool roms_sources_given() { return g_src.given; }

extern "C" int roms_hip_set_sources(int Nsrc, const int *Isrc, const int *Jsrc, const double *Dsrc, const double *Qbar,
                                    const double *Qsrc, const double *Tsrc, const int *LtracerSrc)
{
  const char *me = "roms_hip_set_sources";
  if (!g_ctx.inited || !g_ctx.have_bounds || !g_ctx.have_params)
    return roms_fail(me, "roms_hip_init, roms_hip_set_bounds and roms_hip_set_params come first");
  if (Nsrc < 0) return roms_fail(me, "Nsrc < 0");
  if (!(g_ctx.p.point_sources & 3))
    return roms_fail(me, "point sources: neither bit of roms_params_t.point_sources (LuvSrc, LwSrc) is set");
  const bool luv = (g_ctx.p.point_sources & 1) != 0, lw = (g_ctx.p.point_sources & 2) != 0;
  if (Nsrc == 0) {
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    sources_release();
    g_src.given = true;
    return 0;
  }
  if (!Isrc || !Jsrc || !Dsrc || !Qbar || !Qsrc || !Tsrc || !LtracerSrc) return roms_fail(me, "null argument");
  const roms_bounds_t &b = g_ctx.b;
  const int N = b.N, NT = b.NT;
  const long ni = b.UBi - b.LBi + 1, nij = ni * (b.UBj - b.LBj + 1);
  std::vector<int> D(Nsrc);
  for (int is = 0; is < Nsrc; is++) {
    D[is] = (int)Dsrc[is];
    if (D[is] < 0 || D[is] > 2) return roms_fail(me, "point sources: Dsrc is 0 (u-face), 1 (v-face) or 2 (cell centre)");
  }
  const bool same = g_src.geo && g_src.n == Nsrc && g_src.N == N && g_src.NT == NT && g_src.nij == nij &&
                    std::equal(g_src.I.begin(), g_src.I.end(), Isrc) && std::equal(g_src.J.begin(), g_src.J.end(), Jsrc) &&
                    g_src.D == D;
  const size_t nval = (size_t)Nsrc * (1 + (size_t)N + (size_t)N * NT);
  if (!same) {
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    step2d_graphs_release();               // captured launches hold the old table's addresses
    sources_release();
    const size_t ngeo = 3 * (size_t)Nsrc + 3 * (size_t)nij + 2 * (size_t)Nsrc;
    std::vector<int> geo(ngeo, 0);
    int *umap = geo.data() + 3 * (size_t)Nsrc, *vmap = umap + nij, *wmap = vmap + nij, *cells = wmap + nij;
    int ncell = 0;
    for (int is = 0; is < Nsrc; is++) {
      geo[is] = Isrc[is]; geo[Nsrc + is] = Jsrc[is]; geo[2 * (size_t)Nsrc + is] = D[is];
      // a source of a kind the application has switched off is not looked at (IF (LuvSrc) / IF (LwSrc) in the reference)
      if (D[is] == 2 ? !lw : !luv) continue;
      // the face / cell maps: the last source of a face wins, as the sequential loops of the reference leave it
      if (Isrc[is] >= b.LBi && Isrc[is] <= b.UBi && Jsrc[is] >= b.LBj && Jsrc[is] <= b.UBj)
        (D[is] == 0 ? umap : D[is] == 1 ? vmap : wmap)[(long)(Isrc[is] - b.LBi) + (long)(Jsrc[is] - b.LBj) * ni] = is + 1;
      // the two cells of the face (the one cell of a cell-centred source), where they are interior cells of this tile
      // (each once)
      for (int side = 0; side < (D[is] == 2 ? 1 : 2); side++) {
        const int ci = Isrc[is] - (D[is] == 0 ? side : 0), cj = Jsrc[is] - (D[is] == 1 ? side : 0);
        if (ci < b.Istr || ci > b.Iend || cj < b.Jstr || cj > b.Jend) continue;
        const int cell = (int)((long)(ci - b.LBi) + (long)(cj - b.LBj) * ni);
        if (std::find(cells, cells + ncell, cell) == cells + ncell) cells[ncell++] = cell;
      }
    }
    HIP_TRY(hipMalloc(&g_src.geo, sizeof(int) * ngeo));
    HIP_TRY(hipMalloc(&g_src.val, sizeof(double) * (nval + (size_t)Nsrc * N)));
    HIP_TRY(hipMemcpy(g_src.geo, geo.data(), sizeof(int) * ngeo, hipMemcpyHostToDevice));
    g_src.n = Nsrc; g_src.N = N; g_src.NT = NT; g_src.nij = nij;
    g_src.I.assign(Isrc, Isrc + Nsrc); g_src.J.assign(Jsrc, Jsrc + Nsrc); g_src.D = D;
    RomsSrc &S = g_ctx.hostc.src;
    S.n = Nsrc;
    S.I = g_src.geo; S.J = g_src.geo + Nsrc; S.D = g_src.geo + 2 * (size_t)Nsrc;
    S.umap = g_src.geo + 3 * (size_t)Nsrc; S.vmap = S.umap + nij;
    S.wmap = S.vmap + nij;
    S.cells = S.wmap + nij; S.ncell = ncell;
    S.Qbar = g_src.val; S.Qsrc = g_src.val + Nsrc; S.Tsrc = g_src.val + Nsrc + (size_t)Nsrc * N;
    S.save = g_src.val + nval;
  }
  // the values change with every set_data: one staged copy (a few kB), in stream order after the kernels that read the old ones
  std::vector<double> val(nval);
  std::copy(Qbar, Qbar + Nsrc, val.begin());
  std::copy(Qsrc, Qsrc + (size_t)Nsrc * N, val.begin() + Nsrc);
  std::copy(Tsrc, Tsrc + (size_t)Nsrc * N * NT, val.begin() + Nsrc + (size_t)Nsrc * N);
  HIP_TRY(hipMemcpyAsync(g_src.val, val.data(), sizeof(double) * nval, hipMemcpyHostToDevice, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  for (int it = 0; it < ROMS_MAXNT; it++) g_ctx.hostc.src.ltr[it] = it < NT ? (LtracerSrc[it] != 0) : 0;
  g_src.given = true;
  g_ctx.devc_dirty = true;
  return 0;
}

extern "C" int roms_hip_set_bounds(const roms_bounds_t *b)
{
  if (!g_ctx.inited) return roms_fail("roms_hip_set_bounds", "library not initialised");
  if (b->N > ROMS_MAXN || b->NT > ROMS_MAXNT) return roms_fail("roms_hip_set_bounds", "N or NT too large");
  if (b->ntileI != g_ctx.ntileI || b->ntileJ != g_ctx.ntileJ)
    return roms_fail("roms_hip_set_bounds", "tiling differs from roms_hip_init");
  step2d_graphs_release();
  roms_rowm_release();
  sources_release();                       // the face maps are in the old bounds' index space
  g_ctx.b = *b;
  g_ctx.hostc.b = *b;
  g_ctx.have_bounds = true;
  g_ctx.devc_dirty = true;
  // new extents invalidate every mirror registered under the old ones
  for (int i = 0; i < FID_COUNT; i++) {
    snapshot_forget(i);
    guarded_free(&g_ctx.dev[i], &g_ctx.dev_base[i]);
    g_ctx.host[i] = nullptr;
    g_ctx.count[i] = 0;
  }
  memset(&g_ctx.hostc.F, 0, sizeof g_ctx.hostc.F);
  // device scratch for the _tile routines' automatic arrays
  const long ni = b->UBi - b->LBi + 1, nij = ni * (long)(b->UBj - b->LBj + 1);
  g_ctx.guard = ((4 * ni + 31) / 32) * 32;
  for (int q = 0; q < ROMS_NWS3; q++) {
    guarded_free(&g_ctx.hostc.ws3[q], &g_ctx.ws3_base[q]);
    int rc = guarded_alloc(&g_ctx.hostc.ws3[q], &g_ctx.ws3_base[q], nij * (b->N + 1));
    if (rc) return rc;
  }
  for (int q = 0; q < 32; q++) {
    guarded_free(&g_ctx.hostc.ws2[q], &g_ctx.ws2_base[q]);
    int rc = guarded_alloc(&g_ctx.hostc.ws2[q], &g_ctx.ws2_base[q], nij);
    if (rc) return rc;
  }
  return 0;
}

static int library_default(int id, double *value);

extern "C" int roms_hip_set_params(const roms_params_t *p)
{
  if (!g_ctx.inited) return roms_fail("roms_hip_set_params", "library not initialised");
  if (2 * p->ndtfast > ROMS_MAXFAST) return roms_fail("roms_hip_set_params", "ndtfast too large");
  step2d_graphs_release();
  if ((p->point_sources & 3) != (g_ctx.p.point_sources & 3) && roms_sources_given()) {   // the maps were built for the old switches
    (void)hipStreamSynchronize(g_ctx.stream);
    sources_release();
  }
  g_ctx.p = *p;
  g_ctx.hostc.p = *p;
  // Library-kept defaults (all-water masks, zero biharmonic coefficients, ZoBot) were created under the previous
  // parameters.  One that the new parameters no longer allow (masking switched on, uv_vis4 ...) is dropped here so
  // that the next entry refuses the call with "field not registered" instead of running on the stale default.
  for (int id = 0; id < FID_COUNT; id++)
    if (g_ctx.dev[id] && !g_ctx.host[id]) {
      double value;
      if (!library_default(id, &value)) {
        (void)hipStreamSynchronize(g_ctx.stream);
        snapshot_forget(id);
        guarded_free(&g_ctx.dev[id], &g_ctx.dev_base[id]);
        g_ctx.count[id] = 0;
        *(reinterpret_cast<double **>(&g_ctx.hostc.F) + id) = nullptr;
        if (roms_rowm_is_table_field(id)) roms_rowm_invalidate();
      }
    }
  g_ctx.have_params = true;
  g_ctx.devc_dirty = true;
  return 0;
}

int roms_flush_consts()
{
  if (!g_ctx.devc_dirty) return 0;
  HIP_TRY(hipMemcpyAsync(g_ctx.devc, &g_ctx.hostc, sizeof(RomsDev), hipMemcpyHostToDevice, g_ctx.stream));
  // the host block may change again before the copy runs: make it synchronous
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  g_ctx.devc_dirty = false;
  return 0;
}

extern "C" int roms_hip_register_field(int id, double *host_ptr, long n_doubles)
{
  if (!g_ctx.inited || !g_ctx.have_bounds) return roms_fail("roms_hip_register_field", "set_bounds first");
  if (id < 0 || id >= FID_COUNT) return roms_fail("roms_hip_register_field", "bad field id");
  const long want = roms_field_count(k_field_kind[id], g_ctx.b);
  if (want != n_doubles) {
    char msg[160];
    snprintf(msg, sizeof msg, "field %s: size %ld does not match bounds (%ld)", k_field_name[id], n_doubles, want);
    return roms_fail("roms_hip_register_field", msg);
  }
  step2d_graphs_release();
  if (roms_rowm_is_table_field(id)) roms_rowm_invalidate();
  snapshot_forget(id);          // staging / page-lock of a previous registration (other size or host array)
  guarded_free(&g_ctx.dev[id], &g_ctx.dev_base[id]);
  {
    int rc = guarded_alloc(&g_ctx.dev[id], &g_ctx.dev_base[id], want);
    if (rc) return rc;
  }
  g_ctx.host[id] = host_ptr;
  g_ctx.count[id] = want;
  double **slot = reinterpret_cast<double **>(&g_ctx.hostc.F) + id;
  *slot = g_ctx.dev[id];
  g_ctx.devc_dirty = true;
  return 0;
}

extern "C" int roms_hip_sync_to_device(int id)
{
  if (id < 0 || id >= FID_COUNT || !g_ctx.dev[id]) return roms_fail("roms_hip_sync_to_device", "field not registered");
  if (!g_ctx.host[id]) return roms_fail("roms_hip_sync_to_device", "field is kept by the library (no host array registered)");
  HIP_TRY(hipMemcpyAsync(g_ctx.dev[id], g_ctx.host[id], sizeof(double) * g_ctx.count[id], hipMemcpyHostToDevice, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  if (roms_rowm_is_table_field(id)) roms_rowm_invalidate();      // a grid-metric array may have changed
  return 0;
}

extern "C" int roms_hip_sync_to_host(int id)
{
  if (id < 0 || id >= FID_COUNT || !g_ctx.dev[id]) return roms_fail("roms_hip_sync_to_host", "field not registered");
  if (!g_ctx.host[id]) return roms_fail("roms_hip_sync_to_host", "field is kept by the library (no host array registered)");
  HIP_TRY(hipMemcpyAsync(g_ctx.host[id], g_ctx.dev[id], sizeof(double) * g_ctx.count[id], hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return 0;
}

extern "C" int roms_hip_sync_all_to_device(void)
{
  for (int id = 0; id < FID_COUNT; id++)
    if (g_ctx.dev[id] && g_ctx.host[id])
      HIP_TRY(hipMemcpyAsync(g_ctx.dev[id], g_ctx.host[id], sizeof(double) * g_ctx.count[id], hipMemcpyHostToDevice, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  roms_rowm_invalidate();
  return 0;
}

extern "C" int roms_hip_sync_all_to_host(void)
{
  for (int id = 0; id < FID_COUNT; id++)
    if (g_ctx.dev[id] && g_ctx.host[id])
      HIP_TRY(hipMemcpyAsync(g_ctx.host[id], g_ctx.dev[id], sizeof(double) * g_ctx.count[id], hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return 0;
}

extern "C" double *roms_hip_device_ptr(int id)
{
  if (id < 0 || id >= FID_COUNT) return nullptr;
  return g_ctx.dev[id];
}

extern "C" int roms_hip_device_synchronize(void)
{
  if (!g_ctx.inited) return roms_fail("roms_hip_device_synchronize", "library not initialised");
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return 0;
}

// Fields an application without the option does not have: the land/sea masks without MASKING, the biharmonic
// coefficients without UV_VIS4 / TS_DIF4.  Left unregistered they are kept by the library -- all water / zero --
// so that every kernel still finds an array; the value is never used by the arithmetic (masking = 0 selects code
// without mask loads, the biharmonic operators are not called).
static int library_default(int id, double *value)
{
  const roms_params_t &p = g_ctx.p;
  switch (id) {
  case FID_rmask: case FID_umask: case FID_vmask: case FID_pmask:
    *value = 1.0; return !p.masking;
  case FID_visc4_p: case FID_visc4_r: *value = 0.0; return !p.uv_vis4;
  case FID_diff4: *value = 0.0; return !p.ts_dif4;
  case FID_ZoBot: *value = 1.0; return p.uv_drag != 3 && p.gls_mixing != 1;      /* MY25_MIXING (2) does not read it */
  case FID_Akp: *value = 0.0; return p.gls_mixing != 1;                               /* GLS_MIXING only, mod_mixing.F:245 */
  case FID_tke: case FID_gls: case FID_Lscale: case FID_Akk: *value = 0.0; return !p.gls_mixing;
  case FID_pmask_wet: case FID_rmask_wet: case FID_umask_wet: case FID_vmask_wet: case FID_rmask_wet_avg:
  case FID_pmask_full: case FID_rmask_full: case FID_umask_full: case FID_vmask_full: *value = 1.0; return !p.wet_dry;
  default: return 0;
  }
}

// Every kernel entry calls this first.
int roms_entry_check(const char *name)
{
  if (!g_ctx.inited) return roms_fail(name, "library not initialised");
  if (!g_ctx.have_bounds || !g_ctx.have_params) return roms_fail(name, "bounds/params not set");
  // WET_DRY exists only with MASKING in the reference (wetdry.F:325-345 reads rmask ... vmask unconditionally)
  if (g_ctx.p.wet_dry && !g_ctx.p.masking) return roms_fail(name, "wet_dry = 1 needs masking = 1");
  // mod_sources.F -- never run a river application without its rivers: LuvSrc / LwSrc need the table of roms_hip_set_sources
  if ((g_ctx.p.point_sources & 3) && !roms_sources_given())
    return roms_fail(name, "point sources: LuvSrc / LwSrc is set but roms_hip_set_sources has not handed over SOURCES(ng)");
  for (int id = 0; id < FID_COUNT; id++)
    if (!g_ctx.dev[id]) {
      double value;
      if (library_default(id, &value)) {
        const long want = roms_field_count(k_field_kind[id], g_ctx.b);
        int rc = guarded_alloc(&g_ctx.dev[id], &g_ctx.dev_base[id], want);
        if (rc) return rc;
        std::vector<double> fill((size_t)want, value);
        // on the library's (non-blocking) stream, behind guarded_alloc's memset of the same array
        HIP_TRY(hipMemcpyAsync(g_ctx.dev[id], fill.data(), sizeof(double) * want, hipMemcpyHostToDevice, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        g_ctx.host[id] = nullptr;
        g_ctx.count[id] = want;
        *(reinterpret_cast<double **>(&g_ctx.hostc.F) + id) = g_ctx.dev[id];
        g_ctx.devc_dirty = true;
        continue;
      }
      std::string m = std::string("field not registered: ") + k_field_name[id];
      return roms_fail(name, m.c_str());
    }
  return roms_flush_consts();
}
