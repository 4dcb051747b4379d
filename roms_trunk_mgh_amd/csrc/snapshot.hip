// snapshot.hip -- asynchronous device-to-host snapshots of registered fields: what the
// unchanged output / wrt_his / wrt_rst of the reference (ROMS/Nonlinear/output.F:123-208) need, without
// stalling the step loop.
//
//   roms_hip_snapshot_begin(ids, n)  on the compute stream: copy the fields into a device staging area
//                                    (HBM to HBM, microseconds per field: with 288 GB the second copy is
//                                    affordable), record an event; on a second stream: wait for the event and
//                                    copy staging -> host arrays.  Returns at once; the step loop goes on and
//                                    may overwrite the fields, the staging copy holds the snapshot.
//   roms_hip_snapshot_end()          waits for the second stream; the host arrays then hold the fields as
//                                    they were when begin() was called.
// The host arrays are page-locked once (hipHostRegister) so that the copy is a true asynchronous DMA; if that
// fails (e.g. not enough lockable memory) a pinned bounce buffer and a host memcpy in end() are used instead.
#include <cstring>
#include <vector>
#include "roms_dev.h"

int roms_entry_check(const char *name);

namespace {

struct SnapItem { int id; double *stage; double *bounce; };
hipStream_t g_copy_stream = nullptr;
hipEvent_t g_snap_event = nullptr;
double *g_stage[FID_COUNT] = {nullptr};
double *g_bounce[FID_COUNT] = {nullptr};
bool g_registered[FID_COUNT] = {false};
bool g_register_failed[FID_COUNT] = {false};
std::vector<SnapItem> g_pending;

}  // namespace

// Drop everything the snapshot machinery holds for field `id` (staging copy, bounce buffer, page-lock of
// the host array).  roms_hip_register_field calls it BEFORE it replaces the host pointer / the size, so a
// re-registered field never meets a staging buffer of the old size or an old locked host array.
void snapshot_forget(int id)
{
  if (id < 0 || id >= FID_COUNT) return;
  if (g_copy_stream) (void)hipStreamSynchronize(g_copy_stream);
  for (size_t q = 0; q < g_pending.size();)
    if (g_pending[q].id == id) g_pending.erase(g_pending.begin() + q);
    else q++;
  if (g_stage[id]) (void)hipFree(g_stage[id]);
  if (g_bounce[id]) (void)hipHostFree(g_bounce[id]);
  if (g_registered[id] && g_ctx.host[id]) (void)hipHostUnregister(g_ctx.host[id]);
  g_stage[id] = g_bounce[id] = nullptr;
  g_registered[id] = g_register_failed[id] = false;
}

void snapshot_release()
{
  for (int id = 0; id < FID_COUNT; id++) snapshot_forget(id);
  g_pending.clear();
  if (g_snap_event) (void)hipEventDestroy(g_snap_event);
  if (g_copy_stream) (void)hipStreamDestroy(g_copy_stream);
  g_snap_event = nullptr;
  g_copy_stream = nullptr;
}

extern "C" int roms_hip_snapshot_begin(const int *field_ids, int n)
{
  int rc = roms_entry_check("roms_hip_snapshot_begin");
  if (rc) return rc;
  if (!g_pending.empty()) return roms_fail("roms_hip_snapshot_begin", "a snapshot is already in flight: call roms_hip_snapshot_end first");
  if (!field_ids || n <= 0) return roms_fail("roms_hip_snapshot_begin", "empty field list");
  if (!g_copy_stream) HIP_TRY(hipStreamCreateWithFlags(&g_copy_stream, hipStreamNonBlocking));
  if (!g_snap_event) HIP_TRY(hipEventCreateWithFlags(&g_snap_event, hipEventDisableTiming));
  for (int q = 0; q < n; q++) {
    const int id = field_ids[q];
    if (id < 0 || id >= FID_COUNT || !g_ctx.dev[id]) return roms_fail("roms_hip_snapshot_begin", "field not registered");
    const size_t bytes = sizeof(double) * g_ctx.count[id];
    if (!g_stage[id]) HIP_TRY(hipMalloc(&g_stage[id], bytes));
    if (!g_registered[id] && !g_register_failed[id]) {
      if (hipHostRegister(g_ctx.host[id], bytes, hipHostRegisterDefault) == hipSuccess) g_registered[id] = true;
      else { (void)hipGetLastError(); g_register_failed[id] = true; }
    }
    if (!g_registered[id] && !g_bounce[id]) HIP_TRY(hipHostMalloc(&g_bounce[id], bytes, hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(g_stage[id], g_ctx.dev[id], bytes, hipMemcpyDeviceToDevice, g_ctx.stream));
  }
  HIP_TRY(hipEventRecord(g_snap_event, g_ctx.stream));
  HIP_TRY(hipStreamWaitEvent(g_copy_stream, g_snap_event, 0));
  for (int q = 0; q < n; q++) {
    const int id = field_ids[q];
    const size_t bytes = sizeof(double) * g_ctx.count[id];
    double *dst = g_registered[id] ? g_ctx.host[id] : g_bounce[id];
    HIP_TRY(hipMemcpyAsync(dst, g_stage[id], bytes, hipMemcpyDeviceToHost, g_copy_stream));
    g_pending.push_back(SnapItem{id, g_stage[id], g_registered[id] ? nullptr : g_bounce[id]});
  }
  return 0;
}

extern "C" int roms_hip_snapshot_end(void)
{
  int rc = roms_entry_check("roms_hip_snapshot_end");
  if (rc) return rc;
  if (g_pending.empty()) return 0;
  HIP_TRY(hipStreamSynchronize(g_copy_stream));
  for (const SnapItem &it : g_pending)
    if (it.bounce) std::memcpy(g_ctx.host[it.id], it.bounce, sizeof(double) * g_ctx.count[it.id]);
  g_pending.clear();
  return 0;
}
