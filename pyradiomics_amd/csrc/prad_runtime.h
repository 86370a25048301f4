// prad_runtime.h -- per-thread device context, workspace cache, error plumbing and kernel timing
// for the MI355X texture-matrix engine.  gfx950 only; no CPU fallback lives here or anywhere in csrc/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/pyradiomics_amd.h"

namespace prad {

// ---------------------------------------------------------------------------------------------
// error state (thread local; no exceptions cross the C ABI)
// ---------------------------------------------------------------------------------------------
struct ErrorState {
  char msg[512];
  ErrorState() { msg[0] = 0; }
};
inline ErrorState &err_state() {
  static thread_local ErrorState e;
  return e;
}
inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_state().msg, sizeof(err_state().msg), fmt, ap);
  va_end(ap);
  return code;
}

#define PRAD_HIP(call)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return ::prad::fail(e_ == hipErrorOutOfMemory ? PRAD_E_NOMEM : PRAD_E_HIP, "%s failed: %s (%s:%d)", \
                          #call, hipGetErrorString(e_), __FILE__, __LINE__);                        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// A named, grow-only device buffer cache.  Texture builds are called repeatedly with the same
// shapes (one per derived image / per voxel batch / per bench step); hipMalloc per call would
// dominate small cases and serialise the device, so buffers are kept and reused.
// ---------------------------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct KernelTime {
  std::string family;
  hipEvent_t a, b;
};

#define PRAD_MAX_LANES 4
#define PRAD_ARENA_BYTES (4u << 20)

struct Context {
  int device = 0;
  bool device_set = false;
  hipStream_t own_stream = nullptr;   // used by the host-pointer entry points
  std::map<std::string, DevBuf> bufs;
  std::map<std::string, DevBuf> pinned;
  // timing of the last call
  std::vector<KernelTime> times;
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
  hipEvent_t call_a = nullptr, call_b = nullptr;
  bool call_timed = false;
  const char *last_path = "none";
  const char *last_variant = "none";   // which sweep kernels served the last GLCM / GLRLM call ("fw", "fw2", "lines")
  // deferred mode (prad_set_deferred): GLCM/GLRLM device calls only enqueue work; the "levels outside [1, Ng]" flag of
  // every such call is latched into a sticky device word that prad_deferred_status() reads after synchronising
  bool deferred = false;
  // accumulated timing (prad_timing_begin): events are not recycled between calls, every bracket is kept
  bool timing_accumulate = false;
  std::string timing_only;     // prad_timing_begin_only: bracket this kernel family alone (no call brackets): two events per step
  std::vector<KernelTime> all_times;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> all_calls;
  // last angle table uploaded (skips the pageable host-to-device copy when a caller repeats the same angles), per
  // workspace buffer: keyed like the device buffer it describes (lane AND device -- a thread that alternates between two
  // GPUs must not take GPU 1's table for GPU 0's)
  std::map<std::string, std::vector<int>> angles_cached;
  // Lanes: deferred whole-volume GLCM/GLRLM calls alternate between `lanes` internal streams, each with a workspace of
  // its own, so that the kernels of consecutive volumes share the GPU: the HBM-bound pack of one volume runs while the
  // issue-bound sweep of the previous one holds other CUs, launch gaps and kernel tails are filled (512^3: 0.63 ->
  // 0.56 ms per volume, 256^3: 0.167 -> 0.107).  Forcing the sweeps of the two lanes to run one after the other (only
  // the pack underneath) measured slower (0.60 ms).  lane = -1: the caller's stream and the plain workspace.
  // Workspace set (prad_set_workspace): a caller that drives TWO streams from one thread -- the case pipeline queues the
  // enqueue-only classes on a side stream while the classes that talk to the host run on the main one -- gives each
  // stream its own set of workspace buffers, as the lanes do for the library's internal streams.
  int workspace = 0;
  int lanes = 0;       // 0 = not configured yet (PRAD_LANES, default 2; 1 = off)
  int lane = -1;
  unsigned long long lane_seq = 0;
  hipStream_t lane_stream[PRAD_MAX_LANES] = {};
  hipEvent_t lane_in[PRAD_MAX_LANES] = {};
  // GLSZM phase-1 -> phase-2 state
  long long glszm_nzones = 0;
  int glszm_nvox = 0;
  int glszm_max_region = 0;
  std::vector<long long> glszm_zone_offsets;  // per kernel, into the device zone list (size nvox+1)

  int ensure_device() {
    if (!device_set) {
      int n = 0;
      hipError_t e = hipGetDeviceCount(&n);
      if (e != hipSuccess || n <= 0)
        return fail(PRAD_E_HIP, "no HIP device visible (hipGetDeviceCount: %s) -- this library has no CPU fallback",
                    hipGetErrorString(e));
      device_set = true;
    }
    PRAD_HIP(hipSetDevice(device));
    if (!own_stream) PRAD_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    return PRAD_OK;
  }

  // workspace key: buffers are per device and per lane, except the ones every lane shares ("deferred_*")
  std::string key(const char *name) const {
    std::string k(name);
    if (k.compare(0, 9, "deferred_") != 0) {
      if (lane >= 0) k += "#" + std::to_string(lane);
      if (workspace) k += "~" + std::to_string(workspace);
    }
    return k + "@" + std::to_string(device);
  }
  bool has(const char *name) const { return bufs.find(key(name)) != bufs.end(); }
  std::vector<int> &angle_cache() { return angles_cached[key("angles")]; }
  int lane_begin(hipStream_t user, hipStream_t *s) {   // the next lane: its stream waits for everything queued on `user`
    if (lanes == 0) {
      const char *e = getenv("PRAD_LANES");
      lanes = e ? atoi(e) : 2;
      lanes = lanes < 1 ? 1 : (lanes > PRAD_MAX_LANES ? PRAD_MAX_LANES : lanes);
    }
    if (lanes == 1) return PRAD_OK;
    const int l = (int)(lane_seq++ % (unsigned)lanes);
    if (!lane_stream[l]) {
      PRAD_HIP(hipStreamCreateWithFlags(&lane_stream[l], hipStreamNonBlocking));
      PRAD_HIP(hipEventCreateWithFlags(&lane_in[l], hipEventDisableTiming));
    }
    PRAD_HIP(hipEventRecord(lane_in[l], user));
    PRAD_HIP(hipStreamWaitEvent(lane_stream[l], lane_in[l], 0));
    lane = l;
    *s = lane_stream[l];
    return PRAD_OK;
  }
  // `s` waits (on the device, no host synchronisation) for everything queued on the lanes so far
  int lanes_join(hipStream_t s) {
    for (int l = 0; l < PRAD_MAX_LANES; l++)
      if (lane_stream[l]) {
        PRAD_HIP(hipEventRecord(lane_in[l], lane_stream[l]));      // (the event of the lane is free between calls)
        PRAD_HIP(hipStreamWaitEvent(s, lane_in[l], 0));
      }
    return PRAD_OK;
  }
  int lanes_sync() {
    for (int l = 0; l < PRAD_MAX_LANES; l++)
      if (lane_stream[l]) PRAD_HIP(hipStreamSynchronize(lane_stream[l]));
    return PRAD_OK;
  }

  // returns device pointer of at least `bytes` (contents undefined)
  int get(const char *name, size_t bytes, void **out) {
    DevBuf &b = bufs[key(name)];
    if (b.cap < bytes) {
      if (b.p) (void)hipFree(b.p);
      b.p = nullptr;
      b.cap = 0;
      size_t want = bytes + bytes / 8 + 256;
      hipError_t e = hipMalloc(&b.p, want);
      if (e != hipSuccess) {
        b.p = nullptr;
        return fail(PRAD_E_NOMEM, "hipMalloc(%zu) for workspace '%s' failed: %s", want, name, hipGetErrorString(e));
      }
      b.cap = want;
    }
    *out = b.p;
    return PRAD_OK;
  }
  template <typename T>
  int get(const char *name, size_t count, T **out) {
    void *p = nullptr;
    int rc = get(name, count * sizeof(T), &p);
    *out = (T *)p;
    return rc;
  }
  int get_pinned(const char *name, size_t bytes, void **out) {
    DevBuf &b = pinned[workspace ? std::string(name) + "~" + std::to_string(workspace) : std::string(name)];
    if (b.cap < bytes) {
      if (b.p) (void)hipHostFree(b.p);
      b.p = nullptr;
      b.cap = 0;
      hipError_t e = hipHostMalloc(&b.p, bytes + 256, hipHostMallocDefault);
      if (e != hipSuccess) {
        b.p = nullptr;
        return fail(PRAD_E_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      }
      b.cap = bytes + 256;
    }
    *out = b.p;
    return PRAD_OK;
  }

  // ---- result arena ------------------------------------------------------------------------
  // Pinned host memory handed out as a ring (prad_result_alloc).  A feature call made in deferred mode whose output
  // pointers lie inside the arena only ENQUEUES its kernels and its device-to-host copies: the values are there once
  // the stream has been synchronised (prad_deferred_status).  The caller reads its results before it has allocated
  // another PRAD_ARENA_BYTES (a 256^3 case of 9 derived images uses ~40 KB).
  char *arena = nullptr;
  size_t arena_pos = 0;
  int arena_alloc(size_t bytes, void **out) {
    bytes = (bytes + 63) & ~(size_t)63;
    if (bytes > PRAD_ARENA_BYTES / 4) return fail(PRAD_E_ARG, "result arena: %zu bytes in one allocation", bytes);
    if (!arena) {
      // (portable + mapped: kernels of whichever device this thread selects later may store into it)
      hipError_t e = hipHostMalloc((void **)&arena, PRAD_ARENA_BYTES, hipHostMallocPortable | hipHostMallocMapped);
      if (e != hipSuccess) {
        arena = nullptr;
        return fail(PRAD_E_NOMEM, "hipHostMalloc(%zu) for the result arena failed: %s", (size_t)PRAD_ARENA_BYTES, hipGetErrorString(e));
      }
    }
    if (arena_pos + bytes > PRAD_ARENA_BYTES) arena_pos = 0;
    *out = arena + arena_pos;
    arena_pos += bytes;
    return PRAD_OK;
  }
  // the arena is pinned, mapped host memory: a kernel may store its few result values straight into it (no staging
  // buffer, no copy operation in the stream); PRAD_ZERO_COPY=0 goes back to device buffer + copy
  static bool zero_copy() {
    static const bool on = !(getenv("PRAD_ZERO_COPY") && atoi(getenv("PRAD_ZERO_COPY")) == 0);
    return on;
  }
  bool in_arena(const void *p, size_t bytes) const {
    const char *q = (const char *)p;
    return arena && q >= arena && q + bytes <= arena + PRAD_ARENA_BYTES;
  }

  // ---- timing ------------------------------------------------------------------------------
  int new_event(hipEvent_t *ev) {
    if (events_used == event_pool.size()) {
      hipEvent_t e;
      PRAD_HIP(hipEventCreateWithFlags(&e, hipEventReleaseToDevice));   // (timing events: no system-scope cache release per record)
      event_pool.push_back(e);
    }
    *ev = event_pool[events_used++];
    return PRAD_OK;
  }
  // event brackets cost two hipEventRecord per launch group: enqueue-only (deferred) calls skip them unless somebody
  // collects (prad_timing_begin) -- a queued derived image of the case pipeline had ~50 of them
  bool timing_on() const { return !deferred || timing_accumulate; }
  int begin_call(hipStream_t s) {
    times.clear();
    if (!timing_accumulate) events_used = 0;
    call_timed = false;
    if (!timing_on() || !timing_only.empty()) return PRAD_OK;
    int rc;
    if ((rc = new_event(&call_a)) != PRAD_OK) return rc;
    if ((rc = new_event(&call_b)) != PRAD_OK) return rc;
    PRAD_HIP(hipEventRecord(call_a, s));
    return PRAD_OK;
  }
  int end_call(hipStream_t s) {
    if (!timing_on()) return PRAD_OK;
    if (!timing_only.empty()) {            // family-only mode: keep the family brackets, there is no call bracket
      if (timing_accumulate) all_times.insert(all_times.end(), times.begin(), times.end());
      return PRAD_OK;
    }
    PRAD_HIP(hipEventRecord(call_b, s));
    call_timed = true;
    if (timing_accumulate) {
      all_calls.push_back(std::make_pair(call_a, call_b));
      all_times.insert(all_times.end(), times.begin(), times.end());
    }
    return PRAD_OK;
  }
  int tic(const char *family, hipStream_t s) {
    if (!timing_only.empty() && timing_only != family) return PRAD_E_UNSUPPORTED;   // (Timed: not bracketed, no error recorded)
    KernelTime t;
    t.family = family;
    int rc;
    if ((rc = new_event(&t.a)) != PRAD_OK) return rc;
    if ((rc = new_event(&t.b)) != PRAD_OK) return rc;
    PRAD_HIP(hipEventRecord(t.a, s));
    times.push_back(t);
    return PRAD_OK;
  }
  int toc(hipStream_t s) {
    PRAD_HIP(hipEventRecord(times.back().b, s));
    return PRAD_OK;
  }
};

inline Context &ctx() {
  static thread_local Context c;
  return c;
}

// RAII helper so every launch group is bracketed by events on the stream it runs on.
struct Timed {
  Context &c;
  hipStream_t s;
  bool ok;
  Timed(Context &c_, const char *family, hipStream_t s_) : c(c_), s(s_) { ok = c.timing_on() && c.tic(family, s) == PRAD_OK; }
  ~Timed() {
    if (ok) (void)c.toc(s);
  }
};

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PRAD_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return PRAD_OK;
}

#define PRAD_TRY(expr)              \
  do {                              \
    int rc_ = (expr);               \
    if (rc_ != PRAD_OK) return rc_; \
  } while (0)

// Several small zero fills as ONE launch: every hipMemsetAsync is a launch of its own (a runtime call on the host, a fill
// kernel and a dependent-launch gap on the stream); a 256^3 case queued 129 of them (profiles/r04_probes.md section 18).
// Ranges are 4-byte aligned and whole words.
struct ZeroRanges {
  unsigned *p[6];
  unsigned long long words[6];
  int count;
};
static __global__ void __launch_bounds__(256) zero_ranges_kernel(ZeroRanges z) {
  unsigned *p = z.p[blockIdx.y];
  const unsigned long long n = z.words[blockIdx.y];
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x)
    p[i] = 0u;
}
struct ZeroBatch {
  ZeroRanges z;
  bool bad;      // a range that cannot be taken (a seventh one, a size that is not a whole number of words): launch() fails
  ZeroBatch() : bad(false) { z.count = 0; }
  // bytes: a multiple of 4 (every caller zeroes int / unsigned / double / 64-bit arrays); 0 bytes: nothing to do
  ZeroBatch &add(void *ptr, size_t bytes) {
    if (bytes == 0) return *this;
    if ((bytes & 3) != 0 || (((size_t)ptr) & 3) != 0 || z.count >= 6) {   // (round 4 dropped such a range silently: ADVICE r4)
      bad = true;
      return *this;
    }
    z.p[z.count] = (unsigned *)ptr;
    z.words[z.count] = bytes / 4;
    z.count++;
    return *this;
  }
  int launch(hipStream_t s) {
    if (bad) return fail(PRAD_E_ARG, "ZeroBatch: more than 6 ranges, or a range that is not a whole number of aligned words");
    if (z.count == 0) return PRAD_OK;
    unsigned long long most = 0;
    for (int i = 0; i < z.count; i++) most = z.words[i] > most ? z.words[i] : most;
    const unsigned long long need = (most + 255) / 256;
    const unsigned gx = (unsigned)(need > 1024 ? 1024 : (need ? need : 1));
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(gx, (unsigned)z.count), dim3(256), 0, s, z);
    return check_launch("zero_ranges_kernel");
  }
};

// ---------------------------------------------------------------------------------------------
// geometry shared by host and device code
// ---------------------------------------------------------------------------------------------
struct Geo {
  int nd;
  int size[PRAD_MAX_ND];
  long long stride[PRAD_MAX_ND];  // C-contiguous element strides
  long long n;                    // total voxels
};

struct VoxMode {
  int nvox;           // number of kernels (1 in segment mode)
  const int *voxels;  // device int32 [nd][nvox] or nullptr (segment mode: box = whole array)
  int radius;
  int f2d;            // collapsed dimension or -1
  long long boxmax;   // upper bound on voxels per box
};

inline int make_geo(const int *size, int Nd, Geo *g) {
  if (!size) return fail(PRAD_E_ARG, "size is NULL");
  if (Nd < 1 || Nd > PRAD_MAX_ND) return fail(PRAD_E_ARG, "Nd=%d outside [1,%d]", Nd, PRAD_MAX_ND);
  g->nd = Nd;
  long long n = 1;
  for (int d = Nd - 1; d >= 0; d--) {
    if (size[d] < 1) return fail(PRAD_E_ARG, "size[%d]=%d < 1", d, size[d]);
    g->size[d] = size[d];
    g->stride[d] = n;
    n *= size[d];
  }
  for (int d = Nd; d < PRAD_MAX_ND; d++) {
    g->size[d] = 1;
    g->stride[d] = 0;
  }
  if (n > 2147483647LL)
    return fail(PRAD_E_UNSUPPORTED, "arrays above 2^31-1 elements are not supported (the reference's int strides overflow too, _cmatrices.c:1082)");
  g->n = n;
  return PRAD_OK;
}

}  // namespace prad
