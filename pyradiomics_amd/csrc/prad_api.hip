// prad_api.hip -- C ABI (include/pyradiomics_amd.h) of the MI355X texture-matrix engine.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC prad_api.hip -o libpyradiomics_amd.so
//
// Dispatch policy (per call):
//   segment mode, Nd <= 3, unit angles, Ng small enough for LDS-private histograms
//       -> pack_levels + sweep kernels (kernels_sweep.h)            path "sweep"
//   anything else (voxel mode, Nd > 3, |offset| > 1, large Ng), or a volume whose masked levels fall outside
//   [1, Ng] (detected on the device by pack_levels)
//       -> exact generic kernels (kernels_generic.h)                 path "generic"
// There is no host-side compute path: without a HIP device every calculate_* call fails with PRAD_E_HIP.
#include "prad_runtime.h"
#include "kernels_generic.h"
#include "kernels_pairs.h"
#include "kernels_sweep.h"
#include "kernels_sweepfw.h"
#include "kernels_sweepfw2.h"
#include "kernels_neigh.h"
#include "kernels_glszm.h"
#include "kernels_voxel.h"
#include "kernels_voxslide.h"
#include "kernels_mcc.h"
#include "kernels_voxtex.h"
#include "kernels_binning.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <limits>
#include <mutex>
#include <thread>

using namespace prad;

namespace {

const char *kVersion = "pyradiomics_amd 0.1.0 (gfx950)";

// ------------------------------------------------------------------------------------------------
// angle enumeration (cmatrices.c:756-892), host side: tiny, integer, runs once per call
// ------------------------------------------------------------------------------------------------
int angle_count(const int *size, const int *distances, int Nd, int Ndist, int bidirectional, int f2d) {
  long long total = 0;
  for (int k = 0; k < Ndist; k++) {
    const int dist = distances[k];
    if (dist < 1) return 0;
    long long shell_outer = 1, shell_inner = 1;
    for (int d = 0; d < Nd; d++) {
      if (d == f2d) continue;
      if (dist < size[d]) {
        shell_outer *= 2LL * dist + 1;
        shell_inner *= 2LL * dist - 1;
      } else {
        const long long reach = 2LL * (size[d] - 1) + 1;
        shell_outer *= reach;
        shell_inner *= reach;
      }
    }
    total += shell_outer - shell_inner;
  }
  if (!bidirectional) total /= 2;
  return (int)total;
}

int angle_build(const int *size, const int *distances, int Nd, int Ndist, int f2d, int Na, int *angles) {
  int maxd = 0;
  for (int k = 0; k < Ndist; k++) {
    if (distances[k] < 1) return 1;
    maxd = std::max(maxd, distances[k]);
  }
  // enumerate offset vectors in the reference's order: every component runs +maxd .. -maxd, last
  // dimension fastest (cmatrices.c:843-860); emit the legal ones whose infinity norm was requested.
  std::vector<int> off(Nd, maxd);
  int got = 0;
  long long guard = 1;
  for (int d = 0; d < Nd; d++) guard *= (2LL * maxd + 1);
  for (long long it = 0; it < guard && got < Na; it++) {
    int norm = 0;
    bool legal = true;
    for (int d = 0; d < Nd && legal; d++) {
      const int o = off[d];
      if ((d == f2d && o != 0) || o >= size[d] || o <= -size[d]) legal = false;
      norm = std::max(norm, std::abs(o));
    }
    if (legal && norm >= 1) {
      for (int k = 0; k < Ndist; k++)
        if (distances[k] == norm) {
          std::copy(off.begin(), off.end(), angles + (size_t)got * Nd);
          got++;
          break;
        }
    }
    for (int d = Nd - 1; d >= 0; d--) {
      if (off[d] > -maxd) { off[d]--; break; }
      off[d] = maxd;
    }
  }
  return got == Na ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// call setup shared by all matrices
// ------------------------------------------------------------------------------------------------
struct Call {
  Context *c;
  hipStream_t s;
  Geo g;
  VoxMode vm;
  const int32_t *image;   // device
  const uint8_t *mask;    // device
  const int *angles_h;    // host [Na][Nd]
  int *angles_d;          // device copy
  int Na;
  int *flags_d;           // device int[4]: [0] pack saw irregular level, [1] generic index error, [2] sweep LDS base != 0, [3] the pack kernel saw a voxel outside the ROI
  int *flags_h;           // pinned
};

int setup_call(Call &k, const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
               int Na, int Nvox, const int *voxels_dev, int kernelRadius, int force2Ddim, hipStream_t s,
               bool zero_flags = true) {
  Context &c = ctx();
  k.c = &c;
  k.s = s;
  if (!image || !mask) return fail(PRAD_E_ARG, "image/mask is NULL");
  if (!angles || Na < 1) return fail(PRAD_E_ARG, "angles is NULL or Na < 1");
  PRAD_TRY(make_geo(size, Nd, &k.g));
  if (Nvox < 1) return fail(PRAD_E_ARG, "Nvox=%d < 1", Nvox);
  k.vm.nvox = Nvox;
  k.vm.voxels = voxels_dev;
  k.vm.radius = kernelRadius;
  k.vm.f2d = force2Ddim;
  if (voxels_dev) {
    if (kernelRadius <= 0) return fail(PRAD_E_ARG, "Expecting kernelRadius > 0");  // _cmatrices.c:1091-1095
    long long b = 1;
    for (int d = 0; d < Nd; d++)
      if (d != force2Ddim) b *= std::min<long long>(2LL * kernelRadius + 1, k.g.size[d]);
    k.vm.boxmax = b;
  } else {
    if (Nvox != 1) return fail(PRAD_E_ARG, "Nvox=%d without a voxel list", Nvox);
    k.vm.boxmax = k.g.n;
  }
  k.image = image;
  k.mask = mask;
  k.angles_h = angles;
  k.Na = Na;
  {
    int *prev = nullptr;
    auto it = c.bufs.find(c.key("angles"));
    if (it != c.bufs.end()) prev = (int *)it->second.p;
    PRAD_TRY(c.get<int>("angles", (size_t)Na * Nd, &k.angles_d));
    std::vector<int> &cached = c.angle_cache();
    const bool same = prev == k.angles_d && cached.size() == (size_t)Na * Nd &&
                      std::equal(cached.begin(), cached.end(), angles);
    if (!same) {   // (a copy from pageable memory blocks the host: skip it when the table on the device is current)
      PRAD_HIP(hipMemcpyAsync(k.angles_d, angles, sizeof(int) * Na * Nd, hipMemcpyHostToDevice, s));
      PRAD_HIP(hipStreamSynchronize(s));
      cached.assign(angles, angles + (size_t)Na * Nd);
    }
  }
  PRAD_TRY(c.get<int>("flags", 4, &k.flags_d));
  if (zero_flags) PRAD_HIP(hipMemsetAsync(k.flags_d, 0, sizeof(int) * 4, s));   // (the sweep path zeroes them with its accumulators)
  void *fh = nullptr;
  PRAD_TRY(c.get_pinned("flags_h", sizeof(int) * 4, &fh));
  k.flags_h = (int *)fh;
  return PRAD_OK;
}

int read_flags(Call &k) {
  PRAD_HIP(hipMemcpyAsync(k.flags_h, k.flags_d, sizeof(int) * 4, hipMemcpyDeviceToHost, k.s));
  PRAD_HIP(hipStreamSynchronize(k.s));
  return PRAD_OK;
}

inline unsigned blocks_for(long long threads, int bs = 256) { return (unsigned)((threads + bs - 1) / bs); }

int check_grid(long long threads, const char *what) {
  if ((threads + 255) / 256 > 2147483647LL)
    return fail(PRAD_E_UNSUPPORTED, "%s: %lld work items exceed the launch grid; split the voxel batch", what, threads);
  return PRAD_OK;
}

// ------------------------------------------------------------------------------------------------
// generic launches
// ------------------------------------------------------------------------------------------------
int generic_glcm(Call &k, int Ng, double *out) {
  const size_t per = (size_t)Ng * Ng * k.Na;
  PRAD_HIP(hipMemsetAsync(out, 0, sizeof(double) * per * k.vm.nvox, k.s));
  const long long threads = (long long)k.vm.nvox * k.vm.boxmax;
  PRAD_TRY(check_grid(threads, "GLCM"));
  Timed t(*k.c, "generic", k.s);
  hipLaunchKernelGGL(gen_glcm_kernel, dim3(blocks_for(threads)), dim3(256), 0, k.s, k.g, k.vm, k.image, k.mask,
                     k.angles_d, k.Na, Ng, out, k.flags_d + 1);
  return check_launch("gen_glcm_kernel");
}

int generic_glrlm(Call &k, int Ng, int Nr, double *out) {
  const size_t per = (size_t)Ng * Nr * k.Na;
  PRAD_HIP(hipMemsetAsync(out, 0, sizeof(double) * per * k.vm.nvox, k.s));
  int *multi = nullptr;
  PRAD_TRY(k.c->get<int>("multi", (size_t)k.vm.nvox * k.Na, &multi));
  PRAD_HIP(hipMemsetAsync(multi, 0, sizeof(int) * k.vm.nvox * k.Na, k.s));
  const long long threads = (long long)k.vm.nvox * k.vm.boxmax * k.Na;
  PRAD_TRY(check_grid(threads, "GLRLM"));
  Timed t(*k.c, "generic", k.s);
  hipLaunchKernelGGL(gen_glrlm_kernel, dim3(blocks_for(threads)), dim3(256), 0, k.s, k.g, k.vm, k.image, k.mask,
                     k.angles_d, k.Na, Ng, Nr, out, multi, k.flags_d + 1);
  PRAD_TRY(check_launch("gen_glrlm_kernel"));
  const long long pthreads = (long long)k.vm.nvox * Ng * k.Na;
  hipLaunchKernelGGL(glrlm_prune_kernel, dim3(blocks_for(pthreads)), dim3(256), 0, k.s, out, multi, k.vm.nvox, Ng,
                     Nr, k.Na);
  return check_launch("glrlm_prune_kernel");
}

int generic_gldm(Call &k, int Ng, int alpha, double *out) {
  const size_t per = (size_t)Ng * (2 * (size_t)k.Na + 1);
  PRAD_HIP(hipMemsetAsync(out, 0, sizeof(double) * per * k.vm.nvox, k.s));
  const long long threads = (long long)k.vm.nvox * k.vm.boxmax;
  PRAD_TRY(check_grid(threads, "GLDM"));
  Timed t(*k.c, "generic", k.s);
  hipLaunchKernelGGL(gen_gldm_kernel, dim3(blocks_for(threads)), dim3(256), 0, k.s, k.g, k.vm, k.image, k.mask,
                     k.angles_d, k.Na, Ng, alpha, out, k.flags_d + 1);
  return check_launch("gen_gldm_kernel");
}

int generic_ngtdm(Call &k, int Ng, double *out) {
  Timed t(*k.c, "generic", k.s);
  if (k.vm.voxels) {
    hipLaunchKernelGGL(gen_ngtdm_voxel_kernel, dim3(blocks_for(k.vm.nvox, 64)), dim3(64), 0, k.s, k.g, k.vm,
                       k.image, k.mask, k.angles_d, k.Na, Ng, out, k.flags_d + 1);
    return check_launch("gen_ngtdm_voxel_kernel");
  }
  unsigned long long *acc = nullptr;
  const size_t nacc = (size_t)Ng * (k.Na + 1);
  PRAD_TRY(k.c->get<unsigned long long>("ngtdm_acc", nacc, &acc));
  PRAD_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * nacc, k.s));
  PRAD_TRY(check_grid(k.g.n, "NGTDM"));
  hipLaunchKernelGGL(gen_ngtdm_segment_kernel, dim3(blocks_for(k.g.n)), dim3(256), 0, k.s, k.g, k.vm, k.image,
                     k.mask, k.angles_d, k.Na, Ng, acc, k.flags_d + 1);
  PRAD_TRY(check_launch("gen_ngtdm_segment_kernel"));
  hipLaunchKernelGGL(ngtdm_finalize_kernel, dim3(blocks_for(Ng, 64)), dim3(64), 0, k.s, acc, Ng, k.Na, out);
  return check_launch("ngtdm_finalize_kernel");
}

// ------------------------------------------------------------------------------------------------
// sweep path (segment mode GLCM / GLRLM)
// ------------------------------------------------------------------------------------------------
struct SweepPlan {
  bool ok = false;
  int Nz = 1, Ny = 1, Nx = 1;  // volume embedded in 3-D
  SweepSet lines;              // angles marching along z or y
  int row_slot = -1;           // slot of the angle along x, or -1
  bool skip1 = false;          // two-table walk: runs of length 1 are not recorded, finalize restores them from the x angle's runs
  bool fused = false;          // one [prev][len][cur] table instead of separate GLCM / GLRLM tables
  int LPL = 1;                 // lines per lane of the lines kernel (1, 2, 4)
  int pitch = 0, padw = 0;     // row pitch / periodic pad of the packed level volume
  bool vec_rows = false;       // Nx multiple of 16: vector pack and vector row staging
  AngleSet aset;               // all angles embedded in 3-D (for multi_check_kernel)
  // lines kernel configuration
  int RS = 0;                  // GLRLM run lengths kept in LDS (>= Nr: every length, no long-run path)
  bool LONG = false;
  int threads = 256;
  size_t lds_bytes = 0;
  // rows kernel configuration (512 threads, 8 staging tiles behind the histograms)
  int RSr = 0;
  bool LONGr = false;
  size_t lds_bytes_rows = 0;
  // fixed-window lines kernel (kernels_sweepfw.h): rows of 65..512 voxels, fused table
  bool fw = false;
  int fwK = 8;                 // window columns per lane (4: rows up to 256, 8: up to 512)
  int fw_blocks = 1;           // workgroups per angle
  int fw_threads = 1024;       // threads per workgroup of the fixed-window launch
  int fw_rows_threads = 512;   // ... and of the x angle's launch (sweep_fw_rows_kernel: one staging tile per wave)
  int RSfw = 0;                // run-length slots of its table (one workgroup per CU: most of the 160 KB)
  bool LONGfw = false;
  size_t lds_fw = 0;
  int RSfw_rows = 0;           // length slots of the rows role's table (it shares the LDS with 16 staging tiles)
  size_t lds_fw_rows = 0;      // the x angle as a launch of its own (sweep_fw_rows_kernel)
  // two-table fixed-window kernel (kernels_sweepfw2.h): 45+ grey levels
  bool fw2 = false;
  int RS2 = 0;                 // run-length slots per level row
  int fw2_copies = 1;          // copies of the run-length slots in a row (lane l uses copy l mod copies)
  bool LONGfw2 = false;
  size_t lds_fw2 = 0;
  int pitch16 = 0;             // bytes per row of the 16-bit level volume
  // the x angle of a two-table volume on the 16-bit levels (sweep_fw2_rows_kernel): table + one staging tile per wave
  bool fw2_rows = false;
  int RS2r = 0, fw2r_copies = 1, fw2r_waves = 16;
  bool LONGfw2r = false;
  size_t lds_fw2_rows = 0;
  FwSet fwset;
};

constexpr size_t kHistBudget = 72 * 1024;      // LDS bytes a lines workgroup may spend on histograms
constexpr size_t kHistBudgetRows = 36 * 1024;  // same for a rows workgroup (which also holds staging tiles)
constexpr int kRowsThreads = 512;
constexpr size_t kHistBudgetFw = 146 * 1024;   // fixed-window kernel: one 16-wave workgroup per CU (+ its dead zone)

int cu_count() {
  static thread_local int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

// largest RS (<= Nr) whose table fits `budget` bytes; -1 if not even RS = 1 fits
int fit_rs(bool glcm, bool glrlm, bool fused, int Ng, int Nr, size_t budget) {
  int lo = 0, hi = Nr;
  if (sizeof(u32) * (size_t)hist_layout(glcm, glrlm, fused, Ng, 1).words > budget) return -1;
  lo = 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) / 2;
    if (sizeof(u32) * (size_t)hist_layout(glcm, glrlm, fused, Ng, mid).words <= budget) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// Can this call run on the sweep kernels?  (see the dispatch policy at the top of this file)
SweepPlan plan_sweep(const Call &k, int Ng, int Nr, bool want_glcm, bool want_glrlm) {
  SweepPlan p;
  if (k.vm.voxels || k.g.nd > 3 || Ng < 1 || Ng > 255 || k.Na > PRAD_MAX_SWEEP) return p;
  int dims[3] = {1, 1, 1};
  for (int d = 0; d < k.g.nd; d++) dims[3 - k.g.nd + d] = k.g.size[d];
  p.Nz = dims[0]; p.Ny = dims[1]; p.Nx = dims[2];
  const int longest = std::max(dims[0], std::max(dims[1], dims[2]));
  if (want_glrlm && Nr < longest) return p;  // a run could overflow Nr
  if (!want_glrlm) Nr = 1;
  // fused table when both matrices are wanted and it holds run lengths up to at least min(Nr, 8)
  if (want_glcm && want_glrlm) {
    const int rs = fit_rs(true, true, true, Ng, Nr, kHistBudget);
    const int rsr = fit_rs(true, true, true, Ng, Nr, kHistBudgetRows);
    if (Ng <= (255 >> PRAD_FUSED_SHIFT) && rs >= std::min(Nr, 8) && rsr >= std::min(Nr, 4)) {
      p.fused = true;
      p.RS = rs;
      p.RSr = rsr;
    }
  }
  // Two-table fixed-window kernel for the line angles when the fused table does not fit (45+ levels): rows that one wave
  // window covers, a level row [Ng + 1 pairs | RS2 lengths] table within the LDS, and the rows kernel (separate tables, here
  // with up to 100 KB for them) for the angle along x
  const bool force_fw2 = getenv("PRAD_FORCE_FW2") != nullptr && Ng >= 8;   // experiment: the two-table kernel at any level count
  if (force_fw2 && want_glcm && want_glrlm && p.Nx > 64 && p.Nx <= 512) p.fused = false;
  if (want_glcm && want_glrlm && !p.fused && (Ng >= 40 || force_fw2) && p.Nx > 64 && p.Nx <= 512 && !getenv("PRAD_NO_FW2")) {
    // run-length slots per level row and copies of them (lane l adds to copy l mod C): as many copies as leave >= 32 slots
    // (or every length), slots capped at 64 -- longer runs take the checked path, and empty slots cost zeroing and flushing
    const long long maxwords = (158 * 1024) / 4;
    const long long room = maxwords / (Ng + 1) - (Ng + 1);
    int copies = 1;
    long long rs2 = std::min<long long>(room, Nr);
    for (int cc = 8; cc >= 2; cc >>= 1)
      if (room / cc >= std::min<long long>(32, Nr)) {
        copies = cc;
        rs2 = std::min<long long>(std::min<long long>(room / cc, 64), Nr);
        break;
      }
    if (const char *e = getenv("PRAD_FW2_COPIES")) {   // tuning override
      copies = std::max(1, std::min(8, atoi(e)));
      rs2 = std::min<long long>(std::min<long long>(room / copies, copies > 1 ? 64 : Nr), Nr);
    }
    if (const char *e = getenv("PRAD_FW2_RS")) rs2 = std::min<long long>(rs2, atoll(e));   // tuning override
    if (rs2 < Nr && ((Ng + 1 + copies * rs2) & 1) == 0) rs2--;      // odd row stride: consecutive levels on different banks
    const int rsr = fit_rs(true, true, false, Ng, Nr, 116 * 1024);   // (+ 40 KB of staging tiles: within the 160 KB)
    if ((rs2 >= 24 || rs2 >= Nr) && rs2 >= 1 && rsr >= std::min(Nr, 4)) {
      p.fw2 = true;
      p.skip1 = getenv("PRAD_FW2_NOSKIP1") == nullptr;   // (needs the x angle in the call: cleared below when it is not)
      p.RS2 = (int)rs2;
      p.LONGfw2 = rs2 < Nr;
      p.fw2_copies = copies;
      p.lds_fw2 = 4 * fw2_table_words(Ng, p.RS2, copies);
      p.RS = 0;
      p.RSr = rsr;
      // the x angle with the same two tables (round 5): as many waves (16, 8, 4 -- the walk is a serial chain per lane) as
      // leave the table next to their 5 KB staging tiles >= 32 length slots (4 waves: >= 24); else the rows kernel of
      // kernels_sweep.h on the 8-bit copy as before
      for (int waves = 16; waves >= 4 && !p.fw2_rows && !getenv("PRAD_NO_FW2_ROWS"); waves >>= 1) {
        if (const char *e = getenv("PRAD_FW2_ROWS_WAVES")) {   // tuning override
          if (atoi(e) != waves) continue;
        }
        const long long mw = ((158 * 1024) - (long long)waves * 64 * PRAD_ROW16_PITCH) / 4;
        const long long rm = mw / (Ng + 1) - (Ng + 1);
        if (rm < std::min<long long>(waves > 4 ? 32 : 24, Nr)) continue;
        int cc = 1;
        long long r2 = std::min<long long>(std::min<long long>(rm, 64), Nr);
        for (int c2 = 4; c2 >= 2; c2 >>= 1)
          if (rm / c2 >= std::min<long long>(32, Nr)) {
            cc = c2;
            r2 = std::min<long long>(std::min<long long>(rm / c2, 64), Nr);
            break;
          }
        if (r2 < Nr && ((Ng + 1 + cc * r2) & 1) == 0) r2--;      // odd row stride, as above
        if (r2 < 1) continue;
        p.fw2_rows = true;
        p.RS2r = (int)r2;
        p.fw2r_copies = cc;
        p.fw2r_waves = waves;
        p.LONGfw2r = r2 < Nr;
        p.lds_fw2_rows = ((4 * fw2_table_words(Ng, p.RS2r, cc) + 15) & ~(size_t)15) + (size_t)waves * 64 * PRAD_ROW16_PITCH;
      }
    }
  }
  if (!p.fused && !p.fw2) {
    p.RS = want_glrlm ? fit_rs(want_glcm, true, false, Ng, Nr, kHistBudget) : 0;
    p.RSr = want_glrlm ? fit_rs(want_glcm, true, false, Ng, Nr, kHistBudgetRows) : 0;
    if (p.RS < 0 || p.RSr < 0) return p;
    if (!want_glrlm && sizeof(u32) * (size_t)hist_layout(true, false, false, Ng, 0).words > kHistBudgetRows) return p;
  }
  p.LONG = want_glrlm && p.RS < Nr;
  p.LONGr = want_glrlm && p.RSr < Nr;
  p.lds_bytes = sizeof(u32) * (size_t)hist_layout(want_glcm, want_glrlm, p.fused, Ng, p.RS).words;
  p.threads = p.lds_bytes <= 20 * 1024 ? 256 : (p.lds_bytes <= 40 * 1024 ? 512 : 1024);
  if (const char *e = getenv("PRAD_THREADS")) {  // tuning/ablation override
    const int v = atoi(e);
    if (v == 256 || v == 512 || v == 1024) p.threads = v;
  }
  const size_t hw = (size_t)hist_layout(want_glcm, want_glrlm, p.fused, Ng, p.RSr).words;
  p.lds_bytes_rows = sizeof(u32) * ((hw + 3) & ~(size_t)3) + (kRowsThreads / 64) * 64 * PRAD_ROW_PITCH;
  // packed layout: each lane of the lines kernel owns LPL adjacent lines; rows get a periodic pad of one wave width
  // Lines per lane: more lines amortise the per-step overhead (a chunk of 256 / 128 / 64 lines costs about
  // 1.0 / 0.575 / 0.33 per step), fewer lines waste less of a partial last chunk and give the dynamic hand-out more,
  // smaller chunks.  Estimate the walk time of one angle as (chunks per wave + half a chunk of tail) x chunk cost.
  {
    const double waves = std::max(1.0, (double)(cu_count() / std::max(1, std::min(k.Na, 12))) * 16.0);
    const double cost[3] = {1.0, 0.575, 0.33};
    const int cand[3] = {4, 2, 1};
    double best = 1e300;
    p.LPL = 1;
    for (int i = 0; i < 3; i++) {
      if (cand[i] > 1 && p.Nx < 48 * cand[i]) continue;        // lanes would mostly idle
      const double chunks = (double)std::max(p.Ny, 1) * ((p.Nx + 64 * cand[i] - 1) / (64 * cand[i]));
      const double t = (chunks <= waves ? 1.0 : chunks / waves + 0.5) * cost[i];
      if (t < best) {
        best = t;
        p.LPL = cand[i];
      }
    }
  }
  if (const char *e = getenv("PRAD_LPL")) {  // tuning/ablation override
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4) p.LPL = v;
  }
  p.vec_rows = (p.Nx % 16) == 0;
  // fixed-window kernel: needs the fused table and rows that one wave window (64 lanes x 4 or 8 columns) covers
  p.fw = p.fused && p.Nx > 64 && p.Nx <= 1024 && !getenv("PRAD_NO_FW");
  if (p.Nx > 512 && getenv("PRAD_NO_FW16")) p.fw = false;
  if (const char *e = getenv("PRAD_FW_MINVOX")) p.fw = p.fw && k.g.n >= atoll(e);
  p.fwK = p.Nx <= 256 ? 4 : (p.Nx <= 512 ? 8 : 16);
  if (p.fw2) p.fwK = p.Nx <= 256 ? 4 : 8;
  p.padw = (p.fw || p.fw2) ? 0 : std::min(64 * p.LPL, p.Nx);
  p.pitch16 = 2 * ((p.Nx + 7) & ~7);
  p.pitch = (p.Nx + p.padw + 15) & ~15;   // 16-byte aligned rows: vector staging in the rows kernel for any Nx
  p.lines.count = 0;
  p.aset.count = k.Na;
  for (int a = 0; a < k.Na; a++) {
    int o[3] = {0, 0, 0};
    for (int d = 0; d < k.g.nd; d++) o[3 - k.g.nd + d] = k.angles_h[a * k.g.nd + d];
    int first = 0;
    for (int d = 0; d < 3; d++) {
      if (o[d] < -1 || o[d] > 1) return p;
      if (!first && o[d]) first = o[d];
      p.aset.off[a][d] = o[d];
    }
    if (first != 1) return p;  // sweeps assume the first moving component is +1 (unidirectional list)
    if (o[0] == 0 && o[1] == 0) {
      if (p.row_slot >= 0) return p;
      p.row_slot = a;
      continue;
    }
    SweepDesc &D = p.lines.d[p.lines.count++];
    D.slot = a;
    D.NX = p.Nx;
    D.dx = o[2];
    if (o[0] == 1) {  // march z, rows = y
      D.NM = p.Nz; D.NU = p.Ny; D.du = o[1];
      D.sM = (long long)p.Ny * p.pitch; D.sU = p.pitch;
    } else {          // march y, rows = z (never moves)
      D.NM = p.Ny; D.NU = p.Nz; D.du = 0;
      D.sM = p.pitch; D.sU = (long long)p.Ny * p.pitch;
    }
    D.LXc = (p.Nx + 64 * p.LPL - 1) / (64 * p.LPL);
    D.chunks = (long long)D.NU * D.LXc;
  }
  if ((p.fw || p.fw2) && p.lines.count > 0) {
    size_t budget = kHistBudgetFw;
    if (const char *e = getenv("PRAD_FW_BUDGET_KB")) budget = (size_t)std::max(8, atoi(e)) * 1024;   // tuning override
    p.RSfw = fit_rs(true, true, true, Ng, Nr, budget);
    // LDS bank of a bin = (row stride * prev + len + cur) mod 32 with row stride = (RS+1)(Ng+1) words.  On smooth images
    // cur ~ prev +- 1, so the banks of one ds_add spread like (stride+1) * prev: give up a few length slots for a stride
    // whose (stride+1) is odd and 5..11 banks away from 0 (512^3 smooth levels: 0.74 ms at stride = 0 mod 32, 0.57 ms at 22)
    if (p.RSfw < Nr) {
      for (int rs = p.RSfw; rs >= std::max(16, p.RSfw - 12); rs--) {
        const int d = (((rs + 1) * (Ng + 1)) % 32 + 1) % 32, dist = std::min(d, 32 - d);
        if ((d & 1) && dist >= 5 && dist <= 11) {
          p.RSfw = rs;
          break;
        }
      }
    }
    if (const char *e = getenv("PRAD_FW_RS")) p.RSfw = std::max(1, std::min(p.RSfw, atoi(e)));   // tuning override
    p.LONGfw = p.RSfw < Nr;
    p.lds_fw = fw_lds_bytes(hist_layout(true, true, true, Ng, p.RSfw));
    if (p.row_slot >= 0 && p.fw) {   // the x angle: a launch of its own, 8 waves + their staging tiles (sweep_fw_rows_kernel)
      if (const char *e = getenv("PRAD_FW_ROWS_THREADS")) p.fw_rows_threads = std::max(64, std::min(1024, atoi(e) & ~63));   // tuning override
      const size_t tiles = (size_t)(p.fw_rows_threads / 64) * 64 * PRAD_ROW_PITCH;
      int rs = fit_rs(true, true, true, Ng, Nr, std::min<size_t>(100 * 1024, (size_t)160 * 1024 - 2048 - tiles));
      for (int r = rs; rs < Nr && r >= std::max(16, rs - 12); r--) {   // same bank-stride rule as above
        const int d = (((r + 1) * (Ng + 1)) % 32 + 1) % 32, dist = std::min(d, 32 - d);
        if ((d & 1) && dist >= 5 && dist <= 11) { rs = r; break; }
      }
      p.RSfw_rows = rs;
      p.lds_fw_rows = ((fw_lds_bytes(hist_layout(true, true, true, Ng, rs)) + 15) & ~(size_t)15) + tiles;
    }
    {
      // Roles of the one launch (kernels_sweepfw.h sweep_fw_kernel): one per line angle, one 16-wave workgroup per CU over
      // all of them (1-D grid; the remainder of the division goes to the first roles)
      const int nroles = p.lines.count;
      int total = cu_count();
      if (const char *e = getenv("PRAD_FW_BLOCKS")) total = std::max(nroles, atoi(e));   // tuning override: workgroups of the launch
      if (const char *e = getenv("PRAD_FW_THREADS")) p.fw_threads = std::max(64, std::min(1024, atoi(e) & ~63));   // ... and their size
      // The remainder of the division goes to the roles that cost most per line (round 5b): the wave life by role of
      // profiles/r05_fw_phases.md x the workgroups each role had gives 19.5 - 19.6 (lines that drift in x AND wrap in the row
      // dimension), 18.8 - 19.2 (one of the two), 18.1 - 18.5 (neither); until then the first `total % nroles` roles of the
      // angle list took it -- two of the light ones among them -- and the launch waited for role 6 (PRAD_FW_EXTRA_FIRST=1: as then)
      int bonus[PRAD_MAX_SWEEP] = {0};
      {
        int left = total % nroles;
        const bool as_listed = getenv("PRAD_FW_EXTRA_FIRST") != nullptr;
        for (int cls = as_listed ? 0 : 2; cls >= 0 && left > 0; cls--)
          for (int r = 0; r < nroles && left > 0; r++) {
            const SweepDesc &S = p.lines.d[r];
            const int c = (S.dx != 0 ? 1 : 0) + (S.du != 0 ? 1 : 0);
            if ((as_listed || c == cls) && !bonus[r]) {
              bonus[r] = 1;
              left--;
            }
          }
      }
      p.fwset.first_block[0] = 0;
      for (int r = 0; r < nroles; r++) p.fwset.first_block[r + 1] = p.fwset.first_block[r] + total / nroles + bonus[r];
      for (int r = nroles + 1; r < PRAD_MAX_SWEEP + 1; r++) p.fwset.first_block[r] = p.fwset.first_block[nroles];
      p.fw_blocks = p.fwset.first_block[nroles];
    }
    int per_wave = 6;
    if (const char *e = getenv("PRAD_FW_PER_WAVE")) per_wave = std::max(1, atoi(e));
    p.fwset.count = p.lines.count;
    p.fwset.NX = p.Nx;
    // XCD-aware chunk hand-out (kernels_sweepfw.h): opt-in -- it takes the fabric reads of the level volume from 10.2 to 3.9
    // per volume at 512^3 but its 8 shorter pieces cost 4 % of time (profiles/r03_probes.md), and time is the metric
    p.fwset.xcd = 0;
    if (const char *e = getenv("PRAD_FW_XCD")) p.fwset.xcd = ((p.fw || p.fw2) && p.Nz >= 64 && atoi(e) != 0) ? 1 : 0;
    for (int i = 0; i < p.lines.count; i++) {
      const SweepDesc &S = p.lines.d[i];
      FwDesc &D = p.fwset.d[i];
      D.slot = S.slot; D.NM = S.NM; D.NU = S.NU; D.du = S.du; D.dx = S.dx; D.sM = S.sM; D.sU = S.sU;
      p.fwset.pitch = p.pitch;
      if (p.fw2) {   // byte strides of the 16-bit level volume
        const bool march_z = S.sM >= S.sU;
        D.sM = march_z ? (long long)p.Ny * p.pitch16 : p.pitch16;
        D.sU = march_z ? p.pitch16 : (long long)p.Ny * p.pitch16;
        p.fwset.pitch = p.pitch16;
      }
      p.fwset.nrows = p.Nz * p.Ny;
      const long long want = (long long)per_wave * (p.fwset.first_block[i + 1] - p.fwset.first_block[i]) * (p.fw_threads / 64);
      int pieces = (int)std::max<long long>(1, (want + D.NU - 1) / D.NU);
      int CL = ((D.NM + pieces - 1) / pieces + 7) & ~7;
      CL = std::max(CL, D.NM >= 128 ? 64 : 16);   // (shorter pieces only multiply the piece start / tail overhead)
      const bool march_z_role = S.sM >= S.sU && p.Nz > 1;
      D.dom_kind = march_z_role ? 0 : 1;
      if (p.fwset.xcd && march_z_role) CL = std::max(32, ((D.NM + PRAD_FW_DOMAINS - 1) / PRAD_FW_DOMAINS + 7) & ~7);   // a piece = an XCD's z-slab
      if (const char *e = getenv("PRAD_FW_CL")) CL = std::max(8, atoi(e) & ~7);
      D.CL = CL;
      D.pieces = (D.NM + CL - 1) / CL;
      const long long chunks = (long long)D.NU * D.pieces;
      if (chunks > 2000000000LL) return p;   // (cannot happen below 2^31 voxels) -> generic path
      D.chunks = (int)chunks;
    }
  } else {
    p.fw = false;
    if (p.fw2) return SweepPlan();   // (only the x angle was asked for: not worth a plan of its own) -> generic path
  }
  if (p.row_slot < 0) p.skip1 = false;   // no x angle in this call: nothing to restore the runs of length 1 from
  p.ok = true;
  return p;
}

template <bool G, bool R, bool LNG, bool F, int LPL>
int launch_lines_lpl(Call &k, const SweepPlan &p, const uint8_t *levels, int Ng, int Nr, u32 *glcm_acc,
                     u32 *glrlm_acc, int *multi) {
  long long maxchunks = 0;
  for (int i = 0; i < p.lines.count; i++) maxchunks = std::max(maxchunks, p.lines.d[i].chunks);
  const int wpb = p.threads / 64;
  // Chunks (one NM-step serial walk of 64*LPL lines) are grabbed dynamically.  ONE workgroup per CU across all angles
  // measured best at 512^3 (20-21 workgroups per angle on 256 CUs: 0.64 ms; 42, i.e. two per CU: 0.66 ms; anything
  // between makes some CUs host two workgroups while others host one: 0.75-0.93 ms): with ~3 chunks per wave the
  // tail is short, and a lone workgroup has the CU's LDS bandwidth to itself.
  const long long blocks_avail = std::max<long long>(1, (long long)cu_count() / p.lines.count);
  unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>(blocks_avail, (maxchunks + wpb - 1) / wpb));
  if (const char *e = getenv("PRAD_LINES_BLOCKS")) {  // tuning override: workgroups per angle
    const int v = atoi(e);
    if (v >= 1) gx = (unsigned)v;
  }
  PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_lines_kernel<G, R, LNG, F, LPL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes));
  hipLaunchKernelGGL((sweep_lines_kernel<G, R, LNG, F, LPL>), dim3(gx, p.lines.count), dim3(p.threads), p.lds_bytes,
                     k.s, p.lines, levels, Ng, Nr, p.RS, glcm_acc, glrlm_acc, multi, multi + PRAD_MAX_SWEEP,
                     k.flags_d);
  return check_launch("sweep_lines_kernel");
}

template <bool G, bool R, bool LNG, bool F>
int launch_lines(Call &k, const SweepPlan &p, const uint8_t *levels, int Ng, int Nr, u32 *glcm_acc, u32 *glrlm_acc,
                 int *multi) {
  if (p.LPL == 4) return launch_lines_lpl<G, R, LNG, F, 4>(k, p, levels, Ng, Nr, glcm_acc, glrlm_acc, multi);
  if (p.LPL == 2) return launch_lines_lpl<G, R, LNG, F, 2>(k, p, levels, Ng, Nr, glcm_acc, glrlm_acc, multi);
  return launch_lines_lpl<G, R, LNG, F, 1>(k, p, levels, Ng, Nr, glcm_acc, glrlm_acc, multi);
}

template <bool LNG, int K, bool HASPAD, bool PACK>
int launch_fw_kpp(Call &k, const SweepPlan &p, const PackJob &pj, const uint8_t *levels, const uint8_t *rowzero, int Ng, int Nr,
                  u32 *glcm_acc, u32 *glrlm_acc, int *multi, int *flags_d) {
  PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_fw_kernel<LNG, K, HASPAD, PACK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_fw));
  hipLaunchKernelGGL((sweep_fw_kernel<LNG, K, HASPAD, PACK>), dim3(p.fw_blocks), dim3(p.fw_threads), p.lds_fw, k.s, p.fwset, pj,
                     levels, rowzero, Ng, Nr, p.RSfw, glcm_acc, glrlm_acc, multi + 2 * PRAD_MAX_SWEEP + PRAD_FW_WORK_STRIDE, flags_d);
  return check_launch("sweep_fw_kernel");
}
template <bool LNG, int K, bool HASPAD>
int launch_fw_kp(Call &k, const SweepPlan &p, const PackJob &pj, const uint8_t *levels, const uint8_t *rowzero, int Ng, int Nr,
                 u32 *glcm_acc, u32 *glrlm_acc, int *multi, int *flags_d) {
  if (pj.n16 > 0) return launch_fw_kpp<LNG, K, HASPAD, true>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
  return launch_fw_kpp<LNG, K, HASPAD, false>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
}
template <bool LNG, int K>
int launch_fw_k(Call &k, const SweepPlan &p, const PackJob &pj, const uint8_t *levels, const uint8_t *rowzero, int Ng, int Nr,
                u32 *glcm_acc, u32 *glrlm_acc, int *multi, int *flags_d) {
  if (p.Nx != 64 * K) return launch_fw_kp<LNG, K, true>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
  return launch_fw_kp<LNG, K, false>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
}
// the fixed-window launch of one volume: every line angle and, optionally, the pack of the NEXT volume as a side job
int launch_fw(Call &k, const SweepPlan &p, const PackJob &pj, const uint8_t *levels, const uint8_t *rowzero, int Ng, int Nr,
              u32 *glcm_acc, u32 *glrlm_acc, int *multi, int *flags_d) {
  if (p.fwK == 16)   // rows of 513 .. 1024 voxels: 16 columns per lane
    return p.LONGfw ? launch_fw_k<true, 16>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d)
                    : launch_fw_k<false, 16>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
  if (p.LONGfw) return p.fwK == 4 ? launch_fw_k<true, 4>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d)
                                : launch_fw_k<true, 8>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
  return p.fwK == 4 ? launch_fw_k<false, 4>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d)
                    : launch_fw_k<false, 8>(k, p, pj, levels, rowzero, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
}

template <bool LNG, int K, bool HASPAD, bool SKIP1, bool PACK>
int launch_fw2_khsp(Call &k, const SweepPlan &p, const PackJob &pj, const uint8_t *levels16, int Ng, int Nr, u32 *glcm_acc, u32 *glrlm_acc,
                    int *work, int *flags_d) {
  PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_fw2_kernel<LNG, K, HASPAD, SKIP1, PACK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_fw2));
  hipLaunchKernelGGL((sweep_fw2_kernel<LNG, K, HASPAD, SKIP1, PACK>), dim3(p.fw_blocks), dim3(1024), p.lds_fw2, k.s, p.fwset, pj, levels16,
                     Ng, Nr, p.RS2, p.fw2_copies, glcm_acc, glrlm_acc, work, flags_d);
  return check_launch("sweep_fw2_kernel");
}
template <bool LNG, int K, bool HASPAD, bool SKIP1>
int launch_fw2_khs(Call &k, const SweepPlan &p, const PackJob &pj, const uint8_t *levels16, int Ng, int Nr, u32 *glcm_acc, u32 *glrlm_acc,
                   int *work, int *flags_d) {
  if (pj.n16 > 0) return launch_fw2_khsp<LNG, K, HASPAD, SKIP1, true>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, work, flags_d);
  return launch_fw2_khsp<LNG, K, HASPAD, SKIP1, false>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, work, flags_d);
}
template <bool LNG, int K>
int launch_fw2_k(Call &k, const SweepPlan &p, const PackJob &pj, const uint8_t *levels16, int Ng, int Nr, u32 *glcm_acc, u32 *glrlm_acc,
                 int *multi, int *flags_d) {
  int *work = multi + 2 * PRAD_MAX_SWEEP + PRAD_FW_WORK_STRIDE;
  if (p.Nx != 64 * K)
    return p.skip1 ? launch_fw2_khs<LNG, K, true, true>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, work, flags_d)
                   : launch_fw2_khs<LNG, K, true, false>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, work, flags_d);
  return p.skip1 ? launch_fw2_khs<LNG, K, false, true>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, work, flags_d)
                 : launch_fw2_khs<LNG, K, false, false>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, work, flags_d);
}
// the two-table launch of one volume: every line angle and, optionally, the pack of the NEXT volume as a side job
int launch_fw2(Call &k, const SweepPlan &p, const PackJob &pj, const uint8_t *levels16, int Ng, int Nr, u32 *glcm_acc, u32 *glrlm_acc,
               int *multi, int *flags_d) {
  if (p.LONGfw2) return p.fwK == 4 ? launch_fw2_k<true, 4>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d)
                                 : launch_fw2_k<true, 8>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
  return p.fwK == 4 ? launch_fw2_k<false, 4>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d)
                    : launch_fw2_k<false, 8>(k, p, pj, levels16, Ng, Nr, glcm_acc, glrlm_acc, multi, flags_d);
}

template <bool LNG>
int launch_fw_rows(Call &k, const SweepPlan &p, const uint8_t *levels, int Ng, int Nr, u32 *glcm_acc, u32 *glrlm_acc, int *flags_d) {
  const long long nrows = (long long)p.Nz * p.Ny, groups = nrows >= 4096 ? ((nrows + 511) / 512) * 8 : (nrows + 63) / 64;
  const int wpb = p.fw_rows_threads / 64;
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((groups + wpb - 1) / wpb, (long long)cu_count()));
  PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_fw_rows_kernel<LNG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_fw_rows));
  hipLaunchKernelGGL((sweep_fw_rows_kernel<LNG>), dim3(gx), dim3(p.fw_rows_threads), p.lds_fw_rows, k.s, levels, nrows, p.Nx,
                     p.pitch, p.row_slot, Ng, Nr, p.RSfw_rows, glcm_acc, glrlm_acc, flags_d);
  return check_launch("sweep_fw_rows_kernel");
}

template <bool LNG>
int launch_fw2_rows(Call &k, const SweepPlan &p, const uint8_t *levels16, int Ng, int Nr, u32 *glcm_acc, u32 *glrlm_acc, int *flags_d) {
  const long long nrows = (long long)p.Nz * p.Ny, groups = nrows >= 4096 ? ((nrows + 511) / 512) * 8 : (nrows + 63) / 64;
  const int wpb = p.fw2r_waves;
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((groups + wpb - 1) / wpb, (long long)cu_count()));
  PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_fw2_rows_kernel<LNG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_fw2_rows));
  hipLaunchKernelGGL((sweep_fw2_rows_kernel<LNG>), dim3(gx), dim3(64 * wpb), p.lds_fw2_rows, k.s, levels16, nrows, p.Nx, p.pitch16,
                     p.row_slot, Ng, Nr, p.RS2r, p.fw2r_copies, glcm_acc, glrlm_acc, flags_d);
  return check_launch("sweep_fw2_rows_kernel");
}

template <bool G, bool R, bool LNG, bool F>
int launch_rows(Call &k, const SweepPlan &p, const uint8_t *levels, int Ng, int Nr, u32 *glcm_acc, u32 *glrlm_acc,
                int *multi) {
  const long long nrows = (long long)p.Nz * p.Ny;
  const long long groups = (nrows + 63) / 64;
  const int wpb = kRowsThreads / 64;
  const int per_cu = std::max(1, std::min(2048 / kRowsThreads, (int)(160 * 1024 / p.lds_bytes_rows)));
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((groups + wpb - 1) / wpb, (long long)cu_count() * per_cu));
  PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_rows_kernel<G, R, LNG, F>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes_rows));
  hipLaunchKernelGGL((sweep_rows_kernel<G, R, LNG, F>), dim3(gx), dim3(kRowsThreads), p.lds_bytes_rows, k.s, levels,
                     nrows, p.Nx, p.pitch, p.row_slot, Ng, Nr, p.RSr, glcm_acc, glrlm_acc, multi, k.flags_d);
  return check_launch("sweep_rows_kernel");
}

// zeroes three word ranges in one launch (a: 16-byte aligned, any length)
__global__ void __launch_bounds__(256) zero3_kernel(u32 *__restrict__ a, size_t na, u32 *__restrict__ b, size_t nb,
                                                    u32 *__restrict__ c3, size_t nc) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  uint4 *a4 = reinterpret_cast<uint4 *>(a);
  for (size_t i = t; i < na / 4; i += nt) a4[i] = make_uint4(0, 0, 0, 0);
  for (size_t i = (na & ~(size_t)3) + t; i < na; i += nt) a[i] = 0;
  for (size_t i = t; i < nb; i += nt) b[i] = 0;
  for (size_t i = t; i < nc; i += nt) c3[i] = 0;
}

// One packed volume between its pack and its finalize: device pointers of ONE workspace set (Context::lane) and the plan.
struct VolState {
  bool valid = false;
  SweepPlan p;
  int Ng = 0, Nr = 0, Na = 0;
  double *glcm = nullptr, *glrlm = nullptr;   // outputs (device), either may be NULL
  uint8_t *levels = nullptr, *rowzero = nullptr;
  uint8_t *levels16 = nullptr;   // 16-bit level*4 elements (two-table fixed-window kernel)
  u32 *acc = nullptr, *glcm_acc = nullptr, *glrlm_acc = nullptr;
  int *multi = nullptr;
  int *flags_d = nullptr;
  bool packed_inline = false;   // the pack rides in the previous volume's sweep launch (PackJob)
  int lane = -1;                // workspace set of the volume (pipeline mode)
};

// PRAD_FINALIZE_STREAM=1 (round-6 experiment): the finalize launches of a pipeline volume on a side stream, see pipeline_retire
bool fin_side() {
  static const bool on = getenv("PRAD_FINALIZE_STREAM") != nullptr && atoi(getenv("PRAD_FINALIZE_STREAM")) != 0;
  return on;
}
int fin_wait_for_lane(hipStream_t s, int lane);      // `s` waits for the side-stream finalize that last read workspace set `lane`

// workspace of a volume + the memsets that must precede its pack and its sweeps
int vol_prepare(Call &k, const SweepPlan &p, int Ng, int Nr, double *glcm, double *glrlm, VolState &v) {
  Context &c = *k.c;
  if (fin_side() && c.lane >= 0) PRAD_TRY(fin_wait_for_lane(k.s, c.lane));
  v.lane = c.lane;
  v.valid = true;
  v.p = p;
  v.Ng = Ng;
  v.Nr = Nr;
  v.Na = k.Na;
  v.glcm = glcm;
  v.glrlm = glrlm;
  v.flags_d = k.flags_d;
  const long long nrows = (long long)p.Nz * p.Ny;
  // (a two-table volume whose x angle walks the 16-bit elements keeps no byte copy: round 5b)
  v.levels = nullptr;
  if (!(p.fw2 && p.fw2_rows && p.lines.count > 0 && p.row_slot >= 0) || getenv("PRAD_FW2_KEEP8")) {
    PRAD_TRY(c.get<uint8_t>("levels", (size_t)nrows * p.pitch + 1024, &v.levels));
    v.levels += 512;   // the fixed-window kernel reads (and masks) up to one window before the first and behind the last row
  }
  v.levels16 = nullptr;
  if (p.fw2) {
    PRAD_TRY(c.get<uint8_t>("levels16", (size_t)nrows * p.pitch16 + 4096, &v.levels16));
    v.levels16 += 2048;
  }
  const size_t nglcm = glcm ? (size_t)k.Na * Ng * Ng : 0, nglrlm = glrlm ? (size_t)k.Na * Ng * Nr : 0;
  // accumulators, then per-angle "multi-element" flags, then per-role work counters of the lines kernels
  const size_t nctl = 2 * PRAD_MAX_SWEEP + (size_t)PRAD_FW_WORK_STRIDE * (PRAD_FW_DOMAINS * PRAD_MAX_SWEEP + 1);
  PRAD_TRY(c.get<u32>("sweep_acc", nglcm + nglrlm + nctl, &v.acc));
  v.glcm_acc = v.acc;
  v.glrlm_acc = v.acc + nglcm;
  v.multi = (int *)(v.acc + nglcm + nglrlm);
  v.rowzero = nullptr;
  if (p.fw) PRAD_TRY(c.get<uint8_t>("rowzero", (size_t)nrows + 64, &v.rowzero));
  // accumulators + control words, row flags and the volume's flag words start from zero: ONE launch
  // (three hipMemsetAsync calls are three launches, each with its dependent-launch gap on the stream)
  const size_t acc_w = nglcm + nglrlm + nctl, rz_w = v.rowzero ? ((size_t)nrows + 3) / 4 : 0;
  const size_t total_w = acc_w + rz_w + 4;
  const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>((total_w / 4 + 255) / 256, 1024));
  hipLaunchKernelGGL(zero3_kernel, dim3(gx), dim3(256), 0, k.s, v.acc, acc_w, reinterpret_cast<u32 *>(v.rowzero), rz_w,
                     reinterpret_cast<u32 *>(v.flags_d), (size_t)4);
  return check_launch("zero3_kernel");
}

// can this volume's pack ride in another volume's sweep launch?  (linear layout, vector loads)
bool pipeline_volume(const SweepPlan &p, bool glcm, bool glrlm) {   // a volume the two-stage deferred pipeline takes
  if (!(p.ok && p.lines.count > 0 && glcm && glrlm)) return false;
  return (p.fw && p.fused) || (p.fw2 && !getenv("PRAD_FW2_LANES"));
}
bool pack_inline_ok(const Call &k, const VolState &v) {
  if (!(v.glcm && v.glrlm && v.p.pitch == v.p.Nx && v.p.padw == 0 && (k.g.n % 16) == 0 &&
        ((((uintptr_t)k.image) | ((uintptr_t)k.mask) | ((uintptr_t)v.levels)) & 15) == 0))
    return false;
  // (PackWave16 keeps its piece's voxel index in 32 bits, PackWave its row-flag index: side-job packs stay below 2^31 voxels --
  //  ADVICE r5; a larger volume packs in a launch of its own)
  if (k.g.n >= (1LL << 31)) return false;
  if (v.p.fw2) return v.levels16 && v.p.pitch16 == 2 * v.p.Nx && (((uintptr_t)v.levels16) & 15) == 0;
  return v.p.fw && v.p.fused;
}

PackJob make_pack_job(const Call &k, const VolState &v, const VolState *host) {
  PackJob j;
  memset(&j, 0, sizeof(j));
  j.image = k.image;
  j.mask = k.mask;
  j.levels = v.levels;
  j.levels16 = v.p.fw2 ? v.levels16 : nullptr;
  j.rowzero = v.rowzero;
  j.flags = v.flags_d;
  j.n16 = k.g.n / 16;
  j.NX = v.p.Nx;
  j.Ng = v.Ng;
  // cadence: spread a wave's units over the plain groups of its walk (both per wave); what is left drains behind the walk
  double groups = 0, waves = 16.0;
  if (host) {
    for (int i = 0; i < host->p.fwset.count; i++) groups += (double)host->p.fwset.d[i].NM * host->p.fwset.d[i].NU / PRAD_FW_U;
    waves = 16.0 * std::max(1, host->p.fw_blocks);
  }
  const double units = (double)j.n16 / 64.0;
  // (two-table walk: a unit every 3rd group instead of every 2nd at 512^3 -- 0.707 -> 0.698 ms per volume, smooth inside its run-to-run spread of 0.75 - 0.78;
  //  the fused-table walk measures best at 2: 0.490 against 0.496 at 3, profiles/r05b_probes.md)
  const double slack = v.p.fw2 ? 1.05 : 0.92;
  j.every = (int)std::max(1.0, std::min(64.0, std::floor(slack * groups / std::max(1.0, units))));
  (void)waves;
  if (const char *e = getenv("PRAD_PACK_EVERY")) j.every = std::max(1, atoi(e));
  return j;
}

int vol_pack_standalone(Call &k, VolState &v) {
  Context &c = *k.c;
  const SweepPlan &p = v.p;
  Timed t(c, "pack", k.s);
  if (p.fw2) {   // 16-bit elements for the line walks + plain 8-bit levels for the rows kernel, one pass
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n / 4 + 255) / 256, 8192));
    hipLaunchKernelGGL(pack_levels16_kernel, dim3(gx), dim3(256), 0, k.s, k.image, k.mask, k.g.n, p.Nx, p.pitch16, p.pitch, v.Ng,
                       v.levels16, v.levels, v.flags_d);
    return check_launch("pack_levels16_kernel");
  }
  const int vec_ok = p.vec_rows && ((((uintptr_t)k.image) | ((uintptr_t)k.mask) | ((uintptr_t)v.levels)) & 15) == 0;
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n / 16 + 255) / 256, 4096));
  // the fused walker reads level*4 bytes (see Walker<true, true, LONG, true>); it only exists for Ng <= 44
  const int shift = (v.glcm && v.glrlm && p.fused) ? PRAD_FUSED_SHIFT : 0;
  hipLaunchKernelGGL(pack_levels_kernel, dim3(gx), dim3(256), 0, k.s, k.image, k.mask, k.g.n, p.Nx, p.pitch, p.padw,
                     v.Ng, v.levels, v.flags_d, vec_ok, shift, v.rowzero);
  return check_launch("pack_levels_kernel");
}

template <bool G, bool R, bool F>
int launch_sweeps(Call &k, const VolState &v, const PackJob &pj) {
  const SweepPlan &p = v.p;
  if (p.lines.count > 0 && p.fw && G && R && F) {
    // PRAD_ROWS_STREAM=1 (round-6 experiment): the x angle's launch on a side stream, concurrent with the walk launch -- it only
    // needs the packed levels, writes its own accumulator slots, and its 8-wave workgroups could take the CUs the walk's
    // workgroups leave in the launch's last 10 % (roles finish up to 8 % apart)
    static const bool rows_side = getenv("PRAD_ROWS_STREAM") != nullptr;
    if (rows_side && p.row_slot >= 0) {
      static thread_local hipStream_t side = nullptr;
      static thread_local hipEvent_t e0 = nullptr, e1 = nullptr;
      if (!side) {
        int lo = 0, hi = 0;
        PRAD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));      // (lo = the numerically largest = least urgent)
        PRAD_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, getenv("PRAD_ROWS_STREAM_PRIO") ? atoi(getenv("PRAD_ROWS_STREAM_PRIO")) : lo));
        PRAD_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
        PRAD_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
      }
      PRAD_HIP(hipEventRecord(e0, k.s));
      PRAD_HIP(hipStreamWaitEvent(side, e0, 0));
      {
        Timed t(*k.c, "sweep", k.s);
        PRAD_TRY(launch_fw(k, p, pj, v.levels, v.rowzero, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi, v.flags_d));
      }
      Call k2 = k;
      k2.s = side;
      if (p.RSfw_rows < v.Nr) PRAD_TRY(launch_fw_rows<true>(k2, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.flags_d));
      else PRAD_TRY(launch_fw_rows<false>(k2, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.flags_d));
      PRAD_HIP(hipEventRecord(e1, side));
      PRAD_HIP(hipStreamWaitEvent(k.s, e1, 0));
      return PRAD_OK;
    }
    {
      Timed t(*k.c, "sweep", k.s);
      PRAD_TRY(launch_fw(k, p, pj, v.levels, v.rowzero, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi, v.flags_d));
    }
    if (p.row_slot >= 0) {
      Timed t(*k.c, "rows", k.s);
      if (p.RSfw_rows < v.Nr) PRAD_TRY(launch_fw_rows<true>(k, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.flags_d));
      else PRAD_TRY(launch_fw_rows<false>(k, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.flags_d));
    }
    return PRAD_OK;
  }
  if (p.fw2 && G && R && !F && p.lines.count > 0) {
    {
      Timed t(*k.c, "sweep", k.s);
      PRAD_TRY(launch_fw2(k, p, pj, v.levels16, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi, v.flags_d));
    }
    if (p.row_slot >= 0) {
      Timed t(*k.c, "rows", k.s);
      if (p.fw2_rows) {
        if (p.LONGfw2r) PRAD_TRY(launch_fw2_rows<true>(k, p, v.levels16, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.flags_d));
        else PRAD_TRY(launch_fw2_rows<false>(k, p, v.levels16, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.flags_d));
      } else if (p.LONGr) {
        PRAD_TRY((launch_rows<G, R, true, F>(k, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi)));
      } else {
        PRAD_TRY((launch_rows<G, R, false, F>(k, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi)));
      }
    }
    return PRAD_OK;
  }
  Timed t(*k.c, "sweep", k.s);
  if (p.lines.count > 0) {
    if (R && p.LONG) PRAD_TRY((launch_lines<G, R, true, F>(k, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi)));
    else PRAD_TRY((launch_lines<G, R, false, F>(k, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi)));
  }
  if (p.row_slot >= 0) {
    if (R && p.LONGr) PRAD_TRY((launch_rows<G, R, true, F>(k, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi)));
    else PRAD_TRY((launch_rows<G, R, false, F>(k, p, v.levels, v.Ng, v.Nr, v.glcm_acc, v.glrlm_acc, v.multi)));
  }
  return PRAD_OK;
}

// the sweeps of volume v; `pj` (n16 > 0) = the pack of another volume as a side job of the fixed-window launch
int vol_sweep(Call &k, const VolState &v, const PackJob &pj) {
  if (pj.n16 > 0 && !pipeline_volume(v.p, v.glcm != nullptr, v.glrlm != nullptr))
    return fail(PRAD_E_ARG, "internal: a pack job needs a fixed-window host launch");
  if (pj.n16 > 0 && (pj.levels16 != nullptr) != v.p.fw2)
    return fail(PRAD_E_ARG, "internal: the pack job's layout is not the host launch's");
  if (v.glcm && v.glrlm && v.p.fused) return launch_sweeps<true, true, true>(k, v, pj);
  if (v.glcm && v.glrlm) return launch_sweeps<true, true, false>(k, v, pj);
  if (v.glcm) return launch_sweeps<true, false, false>(k, v, pj);
  return launch_sweeps<false, true, false>(k, v, pj);
}

__global__ void latch_flags_kernel(const int *__restrict__ flags, int *__restrict__ sticky) {
  if (flags[0] || flags[2]) sticky[0] = 1;
}

// deferred GLDM / NGTDM calls: any flag (levels outside [1, Ng] under the mask, a declined fast path) is a verdict to latch
__global__ void latch_any_flag_kernel(const int *__restrict__ flags, int *__restrict__ sticky) {
  if (flags[0] || flags[1] || flags[2]) sticky[0] = 1;
}
int latch_neigh(Context &c, Call &k) {
  int *sticky = nullptr;
  PRAD_TRY(c.get<int>("deferred_sticky", 16, &sticky));
  hipLaunchKernelGGL(latch_any_flag_kernel, dim3(1), dim3(1), 0, k.s, (const int *)k.flags_d, sticky);
  return check_launch("latch_any_flag_kernel");
}

// u32 accumulators -> float64 matrices in the reference layouts; `sticky` (deferred calls): latch the levels verdict
int vol_finalize(Call &k, const VolState &v, int *sticky) {
  Context &c = *k.c;
  const SweepPlan &p = v.p;
  const int Ng = v.Ng, Nr = v.Nr, Na = v.Na;
  double *glcm = v.glcm, *glrlm = v.glrlm;
  bool latched = false;
  {
    Timed t(c, "finalize", k.s);
    const bool runs_diag = p.fused || p.fw2;     // the walks left the GLCM diagonal to the runs
    // SKIP1 walks (fw2) did not count their runs of length 1: only finalize_glcm_diag_kernel puts them back, and it needs both
    // matrices and the x angle's complete run table (plans with skip1 always carry them: plan_sweep clears the flag otherwise)
    if (p.fw2 && p.skip1 && p.lines.count > 0 && !(glcm && glrlm && p.row_slot >= 0))
      return fail(PRAD_E_UNSUPPORTED, "sweep plan derives runs of length 1 but the call cannot restore them");
    if (glcm && glrlm && runs_diag) {
      const int nb1 = (int)blocks_for((long long)Ng * Ng * Na), nb2 = (Ng * Na + 3) / 4;
      hipLaunchKernelGGL(finalize_glcm_diag_kernel, dim3(nb1 + nb2), dim3(256), 0, k.s, v.glcm_acc, v.glrlm_acc, Ng, Nr, Na, nb1,
                         glcm, v.multi, (p.fw2 && p.skip1 && p.lines.count > 0) ? p.row_slot : -1);
      PRAD_TRY(check_launch("finalize_glcm_diag_kernel"));
    } else if (glcm) {
      hipLaunchKernelGGL(finalize_glcm_kernel, dim3(blocks_for((long long)Ng * Ng * Na)), dim3(256), 0, k.s,
                         v.glcm_acc, v.glrlm_acc, Ng, Nr, Na, p.fused ? 1 : 0, glcm);
      PRAD_TRY(check_launch("finalize_glcm_kernel"));
    }
    if (glrlm && runs_diag) {
      if (!glcm) {
        hipLaunchKernelGGL(glcm_diag_resolve_kernel, dim3(Ng, Na), dim3(64), 0, k.s, v.glcm_acc, v.glrlm_acc, Ng, Nr, Na,
                           glcm, v.multi);
        PRAD_TRY(check_launch("glcm_diag_resolve_kernel"));
      }
      if (v.levels)
        hipLaunchKernelGGL(multi_check_kernel<uint8_t>, dim3(128, Na), dim3(256), 0, k.s, p.aset, (const uint8_t *)v.levels, p.Nz, p.Ny,
                           p.Nx, p.pitch, v.multi);
      else
        hipLaunchKernelGGL(multi_check_kernel<unsigned short>, dim3(128, Na), dim3(256), 0, k.s, p.aset,
                           reinterpret_cast<const unsigned short *>(v.levels16), p.Nz, p.Ny, p.Nx, p.pitch16 / 2, v.multi);
      PRAD_TRY(check_launch("multi_check_kernel"));
    }
    if (glrlm) {
      hipLaunchKernelGGL(finalize_glrlm_kernel, dim3(blocks_for((long long)Ng * Nr * Na)), dim3(256), 0, k.s,
                         v.glrlm_acc, v.multi, Ng, Nr, Na, glrlm, (const int *)v.flags_d, sticky);
      PRAD_TRY(check_launch("finalize_glrlm_kernel"));
      latched = sticky != nullptr;
    }
  }
  if (sticky && !latched) {
    hipLaunchKernelGGL(latch_flags_kernel, dim3(1), dim3(1), 0, k.s, v.flags_d, sticky);
    PRAD_TRY(check_launch("latch_flags_kernel"));
  }
  return PRAD_OK;
}

// returns PRAD_OK with *used=false if the device found irregular levels (caller then runs generic)
int sweep_glcm_glrlm(Call &k, const SweepPlan &p, int Ng, int Nr, double *glcm, double *glrlm, bool *used) {
  Context &c = *k.c;
  VolState v;
  PRAD_TRY(vol_prepare(k, p, Ng, Nr, glcm, glrlm, v));
  PRAD_TRY(vol_pack_standalone(k, v));
  PackJob pj;
  memset(&pj, 0, sizeof(pj));
  PRAD_TRY(vol_sweep(k, v, pj));
  int *sticky = nullptr;       // deferred calls latch their levels verdict
  if (c.deferred) PRAD_TRY(c.get<int>("deferred_sticky", 16, &sticky));
  PRAD_TRY(vol_finalize(k, v, sticky));
  if (c.deferred) {   // enqueue only: the verdict on the levels is latched for prad_deferred_status()
    *used = true;
    return PRAD_OK;
  }
  PRAD_TRY(read_flags(k));
  // flags[2]: the fused walker found its table away from LDS address 0 and did nothing (cannot happen with the current
  // toolchain: the dynamic array is the kernels' only LDS object) -- let the generic kernels redo the call
  *used = (k.flags_h[0] == 0 && k.flags_h[2] == 0);
  return PRAD_OK;
}

// ---- deferred calls as a two-stage pipeline (the default deferred mode; PRAD_DEFERRED_MODE=lanes selects the lanes) -------
// A deferred whole-volume call N launches { walks of volume N-1 + pack of volume N } as ONE fixed-window launch (the pack
// is a side job of the walking waves: kernels_sweepfw.h PackJob), then the x angle and the finalize of volume N-1.  Volume
// N itself stays PENDING (packed, not yet walked) until the next deferred call or until prad_deferred_status /
// prad_deferred_join flush it.  Two workspace sets alternate (Context::lane selects the set; the stream is the caller's).
struct PipeState {
  VolState pending;
  hipStream_t s = nullptr;      // stream the pending volume's pack was enqueued on
  unsigned long long seq = 0;
  hipEvent_t ev = nullptr;
  int mode = -1;                // 1 pipeline, 0 lanes
  // side stream of the finalize launches (fin_side()): fin_done[l] is recorded behind the finalize that read workspace set l
  hipStream_t fin = nullptr;
  hipEvent_t fin_in = nullptr, fin_all = nullptr, fin_done[4] = {};
  bool fin_used[4] = {};
};
PipeState &pipe_state() {
  static thread_local PipeState p;
  return p;
}
bool pipeline_mode() {
  PipeState &ps = pipe_state();
  if (ps.mode < 0) {
    const char *e = getenv("PRAD_DEFERRED_MODE");
    ps.mode = (e && strcmp(e, "lanes") == 0) ? 0 : 1;
  }
  return ps.mode == 1;
}
int pipeline_sticky(Context &c, int **sticky) { return c.get<int>("deferred_sticky", 16, sticky); }

// walks + finalize of the pending volume on stream s (pj: the next volume's pack as a side job, or n16 == 0)
int pipeline_retire(Context &c, hipStream_t s, const PackJob &pj) {
  PipeState &ps = pipe_state();
  if (ps.s != s) {   // the pack ran on another stream: order the two on the device
    if (!ps.ev) PRAD_HIP(hipEventCreateWithFlags(&ps.ev, hipEventDisableTiming));
    PRAD_HIP(hipEventRecord(ps.ev, ps.s));
    PRAD_HIP(hipStreamWaitEvent(s, ps.ev, 0));
  }
  Call k;
  k.c = &c;
  k.s = s;
  k.Na = ps.pending.Na;
  k.flags_d = ps.pending.flags_d;
  int *sticky = nullptr;
  PRAD_TRY(pipeline_sticky(c, &sticky));
  PRAD_TRY(vol_sweep(k, ps.pending, pj));
  if (fin_side() && ps.pending.lane >= 0 && ps.pending.lane < 4) {
    // The three finalize launches (13 us + their launch gaps per 512^3 volume) need a handful of workgroups and no LDS: on a
    // side stream they run under the next volume's walk launch instead of in front of it.  The accumulators they read belong to
    // workspace set `lane`: with THREE alternating sets the next volume that zeroes this set is two steps away.
    if (!ps.fin) {
      PRAD_HIP(hipStreamCreateWithFlags(&ps.fin, hipStreamNonBlocking));
      PRAD_HIP(hipEventCreateWithFlags(&ps.fin_in, hipEventDisableTiming));
      PRAD_HIP(hipEventCreateWithFlags(&ps.fin_all, hipEventDisableTiming));
      for (int l = 0; l < 4; l++) PRAD_HIP(hipEventCreateWithFlags(&ps.fin_done[l], hipEventDisableTiming));
    }
    PRAD_HIP(hipEventRecord(ps.fin_in, s));
    PRAD_HIP(hipStreamWaitEvent(ps.fin, ps.fin_in, 0));
    Call k2 = k;
    k2.s = ps.fin;
    PRAD_TRY(vol_finalize(k2, ps.pending, sticky));
    PRAD_HIP(hipEventRecord(ps.fin_done[ps.pending.lane], ps.fin));
    ps.fin_used[ps.pending.lane] = true;
  } else {
    PRAD_TRY(vol_finalize(k, ps.pending, sticky));
  }
  ps.pending.valid = false;
  return PRAD_OK;
}

int fin_wait_for_lane(hipStream_t s, int lane) {
  PipeState &ps = pipe_state();
  if (lane < 0 || lane >= 4 || !ps.fin || !ps.fin_used[lane]) return PRAD_OK;
  PRAD_HIP(hipStreamWaitEvent(s, ps.fin_done[lane], 0));
  return PRAD_OK;
}
// everything queued on the finalize side stream so far: `s` waits for it (s != nullptr) or the host does
int fin_join(hipStream_t s, bool host) {
  PipeState &ps = pipe_state();
  if (!ps.fin) return PRAD_OK;
  if (host) {
    PRAD_HIP(hipStreamSynchronize(ps.fin));
    return PRAD_OK;
  }
  PRAD_HIP(hipEventRecord(ps.fin_all, ps.fin));
  PRAD_HIP(hipStreamWaitEvent(s, ps.fin_all, 0));
  return PRAD_OK;
}

// nothing pending afterwards (kernels enqueued on the pending volume's own stream; no host synchronisation)
int pipeline_flush(Context &c) {
  PipeState &ps = pipe_state();
  if (!ps.pending.valid) return PRAD_OK;
  PackJob none;
  memset(&none, 0, sizeof(none));
  const size_t before = c.times.size();
  PRAD_TRY(pipeline_retire(c, ps.s, none));
  if (c.timing_accumulate) c.all_times.insert(c.all_times.end(), c.times.begin() + (long)before, c.times.end());
  return PRAD_OK;
}

// one deferred call in pipeline mode; *handled = false: the volume is not a fixed-window one, the caller takes the
// ordinary deferred route (the pipeline has been drained)
int pipeline_step(Call &k, const SweepPlan &p, int Ng, int Nr, double *glcm, double *glrlm, bool *handled) {
  Context &c = *k.c;
  PipeState &ps = pipe_state();
  *handled = false;
  if (!pipeline_volume(p, glcm != nullptr, glrlm != nullptr)) return pipeline_flush(c);
  VolState v;
  PRAD_TRY(vol_prepare(k, p, Ng, Nr, glcm, glrlm, v));
  PackJob pj;
  memset(&pj, 0, sizeof(pj));
  bool inl = false;
  if (ps.pending.valid) {
    // (the side job writes the layout of the launch it rides in: fused-table and two-table volumes do not mix)
    inl = pack_inline_ok(k, v) && ps.pending.p.fw2 == v.p.fw2 && !getenv("PRAD_NO_INLINE_PACK");
    if (inl) pj = make_pack_job(k, v, &ps.pending);
    PRAD_TRY(pipeline_retire(c, k.s, pj));
  }
  if (!inl) PRAD_TRY(vol_pack_standalone(k, v));
  v.packed_inline = inl;
  ps.pending = v;
  ps.s = k.s;
  *handled = true;
  return PRAD_OK;
}

// GLCM and/or GLRLM, device pointers
// The tier between the sweeps and the exact kernels (kernels_pairs.h): segment mode, Nd <= 3, any offset list of up to 128
// angles, up to 38 400 grey levels.  *used = false when the call is not for it or the pack found a level outside 1..Ng
// (the outputs are then untouched zeros and the exact kernels redo the call).
int pairs_glcm_glrlm(Call &k, int Ng, int Nr, double *glcm, double *glrlm, bool *used) {
  *used = false;
  Context &c = *k.c;
  if (k.vm.voxels || k.g.nd > 3 || Ng < 1 || Ng > PRAD_PAIR_LDS_WORDS || k.Na > PRAD_PAIR_MAXA || k.g.n >= 0x7fffffffLL ||
      getenv("PRAD_NO_PAIRS"))
    return PRAD_OK;
  int dims[3] = {1, 1, 1};
  for (int d = 0; d < k.g.nd; d++) dims[3 - k.g.nd + d] = k.g.size[d];
  PairAngles A;
  memset(&A, 0, sizeof(A));
  A.n = k.Na;
  for (int a = 0; a < k.Na; a++)
    for (int d = 0; d < k.g.nd; d++) {
      const int o = k.angles_h[a * k.g.nd + d];
      if (o < -127 || o > 127) return PRAD_OK;
      A.o[a][3 - k.g.nd + d] = (signed char)o;
    }
  if (glrlm && Nr < std::max(dims[0], std::max(dims[1], dims[2]))) return PRAD_OK;   // a run could overflow Nr: the exact kernels say what the reference says
  const size_t gwords = glcm ? (size_t)k.Na * Ng * Ng : 0, rwords = glrlm ? (size_t)k.Na * Ng * Nr : 0;
  if ((gwords + rwords) * sizeof(u32) > ((size_t)1 << 30)) return PRAD_OK;
  hipStream_t s = k.s;
  lev16 *L = nullptr;
  u32 *gacc = nullptr, *racc = nullptr, *counts = nullptr;
  int *multi = nullptr;
  PRAD_TRY(c.get<lev16>("pairs_levels", (size_t)k.g.n + 8, &L));
  if (glcm) PRAD_TRY(c.get<u32>("pairs_glcm_acc", gwords, &gacc));
  if (glrlm) {
    PRAD_TRY(c.get<u32>("pairs_glrlm_acc", rwords, &racc));
    PRAD_TRY(c.get<u32>("pairs_counts", (size_t)Ng, &counts));
    PRAD_TRY(c.get<int>("pairs_multi", (size_t)PRAD_PAIR_MAXA, &multi));
  }
  PRAD_HIP(hipMemsetAsync(k.flags_d, 0, sizeof(int) * 4, s));
  const int cus = cu_count();
  {
    Timed t(c, "pack", s);
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n + 255) / 256, (long long)cus * 16));
    hipLaunchKernelGGL(pairs_pack_kernel, dim3(gx), dim3(256), 0, s, k.image, k.mask, k.g.n, Ng, L, k.flags_d);
    PRAD_TRY(check_launch("pairs_pack_kernel"));
  }
  if (glcm) {
    Timed t(c, "pairs", s);
    PRAD_HIP(hipMemsetAsync(gacc, 0, sizeof(u32) * gwords, s));
    int AG, RT, ntile;
    if ((long long)Ng * Ng <= PRAD_PAIR_LDS_WORDS) {
      AG = std::max(1, std::min(k.Na, PRAD_PAIR_LDS_WORDS / (Ng * Ng)));
      RT = Ng;
      ntile = 1;
    } else {
      AG = 1;
      RT = PRAD_PAIR_LDS_WORDS / Ng;
      ntile = (Ng + RT - 1) / RT;
    }
    const int npass = ((k.Na + AG - 1) / AG) * ntile;
    const size_t lds = sizeof(u32) * (size_t)AG * RT * Ng;
    PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pairs_glcm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // workgroups per pass: the count (2 .. 4 per CU over all passes) that leaves the fewest CUs idle in the last round -- one
    // 150 KB table per CU at a time (186 passes at 300 levels x 3 workgroups were 2.2 rounds, billed as 3)
    long long gbest = 1;
    double ebest = 0;
    for (long long gtry = std::max<long long>(1, 2LL * cus / npass); gtry <= std::max<long long>(1, (4LL * cus + npass - 1) / npass); gtry++) {
      const long long total = gtry * npass, rounds = (total + cus - 1) / cus;
      const double eff = (double)total / (double)(rounds * cus);
      if (eff > ebest + 1e-9) {
        ebest = eff;
        gbest = gtry;
      }
    }
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n / 8 + 1023) / 1024, gbest));
    hipLaunchKernelGGL(pairs_glcm_kernel, dim3(gx, (unsigned)npass), dim3(1024), lds, s, L, dims[0], dims[1], dims[2], A, AG, RT,
                       ntile, Ng, gacc, k.flags_d);
    PRAD_TRY(check_launch("pairs_glcm_kernel"));
    const long long total = (long long)Ng * Ng * k.Na;
    hipLaunchKernelGGL(finalize_glcm_kernel, dim3(blocks_for(total)), dim3(256), 0, s, gacc, (const u32 *)nullptr, Ng, 1, k.Na, 0, glcm);
    PRAD_TRY(check_launch("finalize_glcm_kernel"));
  }
  if (glrlm) {
    Timed t(c, "pairs", s);
    PRAD_HIP(hipMemsetAsync(racc, 0, sizeof(u32) * rwords, s));
    PRAD_HIP(hipMemsetAsync(counts, 0, sizeof(u32) * (size_t)Ng, s));
    PRAD_HIP(hipMemsetAsync(multi, 0, sizeof(int) * PRAD_PAIR_MAXA, s));
    const int use_lds = Ng <= 8192;
    const unsigned gc = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n + 255) / 256, (long long)cus * 8));
    hipLaunchKernelGGL(pairs_level_count_kernel, dim3(gc), dim3(256), use_lds ? sizeof(u32) * (size_t)Ng : 0, s, L, k.g.n, Ng, use_lds,
                       counts, k.flags_d);
    PRAD_TRY(check_launch("pairs_level_count_kernel"));
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n + 255) / 256, std::max(1LL, 16LL * cus / k.Na)));
    hipLaunchKernelGGL(pairs_glrlm_kernel, dim3(gx, (unsigned)k.Na), dim3(256), 0, s, L, dims[0], dims[1], dims[2], A, Ng, Nr, racc, multi,
                       k.flags_d);
    PRAD_TRY(check_launch("pairs_glrlm_kernel"));
    hipLaunchKernelGGL(pairs_multi_check_kernel, dim3(gx, (unsigned)k.Na), dim3(256), 0, s, A, L, dims[0], dims[1], dims[2], multi, k.flags_d);
    PRAD_TRY(check_launch("pairs_multi_check_kernel"));
    hipLaunchKernelGGL(pairs_run1_kernel, dim3((unsigned)((Ng * k.Na + 3) / 4)), dim3(256), 0, s, racc, counts, Ng, Nr, k.Na, k.flags_d);
    PRAD_TRY(check_launch("pairs_run1_kernel"));
    const long long total = (long long)Ng * Nr * k.Na;
    hipLaunchKernelGGL(finalize_glrlm_kernel, dim3(blocks_for(total)), dim3(256), 0, s, racc, multi, Ng, Nr, k.Na, glrlm,
                       (const int *)nullptr, (int *)nullptr);
    PRAD_TRY(check_launch("finalize_glrlm_kernel"));
  }
  PRAD_TRY(read_flags(k));
  *used = k.flags_h[0] == 0;
  if (!*used) {   // a level outside 1..Ng: leave zeros for the exact kernels (the finalize kernels above wrote only zeros: every
                  // accumulating kernel returned at its first line)
    PRAD_HIP(hipMemsetAsync(k.flags_d, 0, sizeof(int) * 4, s));
  }
  return PRAD_OK;
}

int texture_pairs_runs(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                       int Na, int Ng, int Nr, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                       double *glcm, double *glrlm, hipStream_t s) {
  if (!glcm && !glrlm) return fail(PRAD_E_ARG, "both outputs are NULL");
  if (Ng < 1 || (glrlm && Nr < 1)) return fail(PRAD_E_ARG, "Ng/Nr must be >= 1");
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  struct LaneGuard {
    Context &c;
    ~LaneGuard() { c.lane = -1; }
  } lane_guard{c};
  bool pipe = c.deferred && !voxels && pipeline_mode();
  if (c.deferred && !voxels && !pipe) PRAD_TRY(c.lane_begin(s, &s));   // lanes mode: whole-volume deferred calls alternate between lanes
  if (pipe) c.lane = (int)(pipe_state().seq++ % (fin_side() ? 3u : 2u));   // pipeline mode: the workspace sets alternate (three with the finalize side stream), the stream is the caller's
  Call k;
  PRAD_TRY(setup_call(k, image, mask, size, Nd, angles, Na, Nvox, voxels, kernelRadius, force2Ddim, s, false));
  SweepPlan p = plan_sweep(k, Ng, Nr, glcm != nullptr, glrlm != nullptr);
  if (pipe && !pipeline_volume(p, glcm != nullptr, glrlm != nullptr)) {
    // Not a volume for the two-stage pipeline (the wrapped-lines kernels; PRAD_FW2_LANES: the two-table kernel of 45+ grey
    // levels as until round 5): its pack is a launch of its own, so deal the call onto the lanes, where the pack of one
    // volume runs under the walk of the previous one
    const hipStream_t user = s;
    PRAD_TRY(pipeline_flush(c));
    pipe = false;
    c.lane = -1;
    PRAD_TRY(c.lane_begin(user, &s));
    if (c.lane >= 0) c.lane += fin_side() ? 3 : 2;      // (workspace sets of their own: #0 / #1 (/ #2) belong to the pipeline's alternating volumes)
    PRAD_TRY(setup_call(k, image, mask, size, Nd, angles, Na, Nvox, voxels, kernelRadius, force2Ddim, s, false));
    p = plan_sweep(k, Ng, Nr, glcm != nullptr, glrlm != nullptr);
  }
  PRAD_TRY(c.begin_call(s));
  if (!p.ok) PRAD_HIP(hipMemsetAsync(k.flags_d, 0, sizeof(int) * 4, s));   // (vol_prepare does it on the sweep path)
  bool done = false;
  c.last_variant = !p.ok ? "none" : (p.fw2 ? "fw2" : (p.fw && glcm && glrlm ? "fw" : "lines"));
  if (pipe) {
    PRAD_TRY(pipeline_step(k, p, Ng, Nr, glcm, glrlm, &done));
    if (done) c.last_path = "sweep";
  }
  if (!done && p.ok) {
    PRAD_TRY(sweep_glcm_glrlm(k, p, Ng, Nr, glcm, glrlm, &done));
    if (done) c.last_path = "sweep";
  }
  if (!done && !c.deferred) {      // (synchronous calls: the tier reads its levels verdict back)
    PRAD_TRY(pairs_glcm_glrlm(k, Ng, Nr, glcm, glrlm, &done));
    if (done) {
      c.last_path = "pairs";
      c.last_variant = "pairs";
    }
  }
  if (!done) {
    if (glcm) PRAD_TRY(generic_glcm(k, Ng, glcm));
    if (glrlm) PRAD_TRY(generic_glrlm(k, Ng, Nr, glrlm));
    PRAD_TRY(read_flags(k));
    c.last_path = "generic";
  }
  PRAD_TRY(c.end_call(s));
  if (c.deferred && done) return PRAD_OK;   // nothing was read back: see prad_deferred_status()
  PRAD_HIP(hipStreamSynchronize(s));
  return k.flags_h[1] ? PRAD_INDEX_ERROR : PRAD_OK;
}

// GLDM / NGTDM on the 16-bit level volume (kernels_pairs.h): what kernels_neigh.h does not take -- more than 255 levels, bin
// tables beyond its LDS budget.  Synchronous calls only; *used = false: not for this tier, or a level outside 1..Ng.
int pairs_neigh(Call &k, bool ngtdm, int Ng, int alpha, double *out, bool *used, bool box_only = false) {
  *used = false;
  Context &c = *k.c;
  if (k.vm.voxels || k.g.nd > 3 || Ng < 1 || Ng > 65535 || k.g.n >= 0x7fffffffLL || getenv("PRAD_NO_PAIRS")) return PRAD_OK;
  int dims[3] = {1, 1, 1};
  for (int d = 0; d < k.g.nd; d++) dims[3 - k.g.nd + d] = k.g.size[d];
  // NGTDM over a FULL box of neighbours (distances [1], [1, 2], [1, 2, 3] ...; in-plane boxes under force2D): separable box sums
  // (kernels_pairs.h); decided on the host's angle list -- the path needs the radii only, not the tier's 128-entry angle table
  int rad[3] = {0, 0, 0};
  bool box = ngtdm && !getenv("PRAD_NGTDM_NO_BOX");
  long long cube = 1;
  if (box) {
    for (int a = 0; a < k.Na; a++)
      for (int d = 0; d < k.g.nd; d++) rad[3 - k.g.nd + d] = std::max(rad[3 - k.g.nd + d], std::abs(k.angles_h[a * k.g.nd + d]));
    cube = (long long)(2 * rad[0] + 1) * (2 * rad[1] + 1) * (2 * rad[2] + 1);
    box = cube - 1 == k.Na && cube <= 4096 && cube * Ng < (1LL << 40);
    if (box) {        // cube - 1 distinct non-zero offsets inside the cube are the whole cube
      std::vector<char> seen((size_t)cube, 0);
      for (int a = 0; box && a < k.Na; a++) {
        long long idx = 0;
        bool zero = true;
        for (int d = 0; d < 3; d++) {
          const int o = d < 3 - k.g.nd ? 0 : k.angles_h[a * k.g.nd + (d - (3 - k.g.nd))];
          zero = zero && o == 0;
          idx = idx * (2 * rad[d] + 1) + (o + rad[d]);
        }
        if (zero || seen[(size_t)idx]) box = false;
        else seen[(size_t)idx] = 1;
      }
    }
  }
  if (box_only && !box) return PRAD_OK;
  if (!box && k.Na > PRAD_PAIR_MAXA) return PRAD_OK;
  PairAngles A;
  memset(&A, 0, sizeof(A));
  A.n = std::min(k.Na, PRAD_PAIR_MAXA);
  for (int a = 0; a < A.n; a++)
    for (int d = 0; d < k.g.nd; d++) {
      const int o = k.angles_h[a * k.g.nd + d];
      if (o < -127 || o > 127) return PRAD_OK;
      A.o[a][3 - k.g.nd + d] = (signed char)o;
    }
  const size_t nacc = (size_t)Ng * (k.Na + 1);
  if (nacc * sizeof(u64) > ((size_t)1 << 30)) return PRAD_OK;
  hipStream_t s = k.s;
  lev16 *L = nullptr;
  u32 *acc32 = nullptr;
  u64 *acc64 = nullptr;
  PRAD_TRY(c.get<lev16>("pairs_levels", (size_t)k.g.n + 8, &L));
  if (ngtdm) PRAD_TRY(c.get<u64>("pairs_ngtdm_acc", nacc, &acc64));
  else PRAD_TRY(c.get<u32>("pairs_gldm_acc", nacc, &acc32));
  PRAD_HIP(hipMemsetAsync(k.flags_d, 0, sizeof(int) * 4, s));
  const int cus = cu_count();
  {
    Timed t(c, "pack", s);
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n + 255) / 256, (long long)cus * 16));
    hipLaunchKernelGGL(pairs_pack_kernel, dim3(gx), dim3(256), 0, s, k.image, k.mask, k.g.n, Ng, L, k.flags_d);
    PRAD_TRY(check_launch("pairs_pack_kernel"));
  }
  {
    Timed t(c, "pairs", s);
    if (ngtdm) PRAD_HIP(hipMemsetAsync(acc64, 0, sizeof(u64) * nacc, s));
    else PRAD_HIP(hipMemsetAsync(acc32, 0, sizeof(u32) * nacc, s));
    // bins in LDS when they fit 150 KB: GLDM u32; NGTDM u64, or u32 with a grid fine enough that a workgroup's sums
    // (voxels x angles x Ng at most) stay below 2^32
    const size_t budget = (size_t)PRAD_PAIR_LDS_WORDS * 4;
    int mode = 0;
    size_t lds = 0;
    if (!ngtdm) {
      if (nacc * sizeof(u32) <= budget) { mode = 1; lds = nacc * sizeof(u32); }
    } else if (nacc * sizeof(u64) <= budget) {
      mode = 1; lds = nacc * sizeof(u64);
    } else if (nacc * sizeof(u32) <= budget) {
      mode = 2; lds = nacc * sizeof(u32);
    }
    // bins beyond 40 KB: one workgroup per CU, so make it a 16-wave one (4 waves per CU left the neighbour gathers
    // latency-bound: GLDM at 300 levels x 124 angles took 7.8 ms)
    const unsigned bt = (mode && lds > 40 * 1024) ? 1024u : 256u;
    long long want = (long long)cus * (bt == 1024 ? 1 : 16);
    if (mode == 2) {      // voxels per workgroup <= 2^32 / (Na * Ng) / 2
      const long long cap = std::max<long long>(1, (1LL << 31) / ((long long)k.Na * Ng));
      want = std::max(want, (k.g.n + cap - 1) / cap);
    }
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n + bt - 1) / bt, want));
    if (box) {
      const bool narrow = cube <= 127 && cube * Ng < (1LL << 24);      // (ROI voxels << 24 | level sum) fits 32 bits
      void *b1 = nullptr, *b2 = nullptr;
      PRAD_TRY(c.get("pairs_box1", (size_t)k.g.n * (narrow ? 4 : 8), &b1));
      PRAD_TRY(c.get("pairs_box2", (size_t)k.g.n * (narrow ? 4 : 8), &b2));
      const unsigned ga = (unsigned)std::max<long long>(1, std::min<long long>((k.g.n + 255) / 256, (long long)cus * 32));
      const int W = k.Na + 1;
#define PRAD_PB(MODE_, WT_, SH_)                                                                                                \
  do {                                                                                                                          \
    hipLaunchKernelGGL((pairs_box_axis_kernel<WT_, SH_>), dim3(ga), dim3(256), 0, s, (const lev16 *)L, (const WT_ *)nullptr,    \
                       (WT_ *)b1, dims[0], dims[1], dims[2], 2, rad[2], (const int *)k.flags_d);                                \
    hipLaunchKernelGGL((pairs_box_axis_kernel<WT_, SH_>), dim3(ga), dim3(256), 0, s, (const lev16 *)nullptr, (const WT_ *)b1,   \
                       (WT_ *)b2, dims[0], dims[1], dims[2], 1, rad[1], (const int *)k.flags_d);                                \
    if (MODE_) PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pairs_ngtdm_box_kernel<MODE_, WT_, SH_>),           \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                             \
    hipLaunchKernelGGL((pairs_ngtdm_box_kernel<MODE_, WT_, SH_>), dim3(gx), dim3(bt), lds, s, (const lev16 *)L,                 \
                       (const WT_ *)b2, dims[0], dims[1], dims[2], rad[0], Ng, W, acc64, (const int *)k.flags_d);               \
  } while (0)
      if (narrow) {
        if (mode == 2) PRAD_PB(2, u32, 24);
        else if (mode == 1) PRAD_PB(1, u32, 24);
        else PRAD_PB(0, u32, 24);
      } else {
        if (mode == 2) PRAD_PB(2, u64, 40);
        else if (mode == 1) PRAD_PB(1, u64, 40);
        else PRAD_PB(0, u64, 40);
      }
#undef PRAD_PB
      PRAD_TRY(check_launch("pairs_ngtdm_box_kernel"));
    } else {
#define PRAD_PN(NG_, MODE_)                                                                                                     \
  do {                                                                                                                          \
    if (MODE_) PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pairs_neigh_kernel<NG_, MODE_>),                    \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                             \
    hipLaunchKernelGGL((pairs_neigh_kernel<NG_, MODE_>), dim3(gx), dim3(bt), lds, s, A, L, dims[0], dims[1], dims[2], Ng,       \
                       alpha, acc32, acc64, k.flags_d);                                                                         \
  } while (0)
    if (ngtdm && mode == 2) PRAD_PN(true, 2);
    else if (ngtdm && mode == 1) PRAD_PN(true, 1);
    else if (ngtdm) PRAD_PN(true, 0);
    else if (mode == 1) PRAD_PN(false, 1);
    else PRAD_PN(false, 0);
#undef PRAD_PN
    PRAD_TRY(check_launch("pairs_neigh_kernel"));
    }
  }
  if (ngtdm) PRAD_TRY(neigh_finalize_ngtdm(&c, s, acc64, Ng, k.Na, out));
  else PRAD_TRY(neigh_finalize_gldm(&c, s, acc32, Ng, k.Na, out));
  PRAD_TRY(read_flags(k));
  *used = k.flags_h[0] == 0;
  if (!*used) {      // irregular level: the exact kernels redo the call on a zeroed output (the finalize kernels wrote zeros and,
                     // for NGTDM, the level column)
    const size_t outn = ngtdm ? (size_t)Ng * 3 : (size_t)Ng * (2 * k.Na + 1);
    PRAD_HIP(hipMemsetAsync(out, 0, sizeof(double) * outn, s));
    PRAD_HIP(hipMemsetAsync(k.flags_d, 0, sizeof(int) * 4, s));
  }
  return PRAD_OK;
}

int texture_gldm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                 int Ng, int alpha, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, double *out,
                 hipStream_t s) {
  if (!out) return fail(PRAD_E_ARG, "gldm is NULL");
  if (Ng < 1) return fail(PRAD_E_ARG, "Ng must be >= 1");
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Call k;
  PRAD_TRY(setup_call(k, image, mask, size, Nd, angles, Na, Nvox, voxels, kernelRadius, force2Ddim, s));
  PRAD_TRY(c.begin_call(s));
  bool done = false;
  PRAD_TRY(neigh_try_gldm(k.c, k.s, k.g, k.vm, k.image, k.mask, k.angles_h, k.Na, Ng, alpha, out, k.flags_d, &done));
  if (done && c.deferred && !voxels) {   // enqueue only: the flags are latched for prad_deferred_status()
    PRAD_TRY(latch_neigh(c, k));
    c.last_path = "neigh";
    return c.end_call(s);
  }
  bool irregular = false;      // the byte kernels' pack already saw a level outside [1, Ng]: the tier would only find it again
  if (done) {
    PRAD_TRY(read_flags(k));
    done = (k.flags_h[0] == 0);
    irregular = !done;
  }
  if (!done && !c.deferred && !irregular) {
    PRAD_TRY(pairs_neigh(k, false, Ng, alpha, out, &done));
    if (done) c.last_path = "pairs";
  } else if (done) {
    c.last_path = "neigh";
  }
  if (!done) {
    PRAD_TRY(generic_gldm(k, Ng, alpha, out));
    PRAD_TRY(read_flags(k));
    c.last_path = "generic";
  }
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  return k.flags_h[1] ? PRAD_INDEX_ERROR : PRAD_OK;
}

int texture_ngtdm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                  int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, double *out,
                  hipStream_t s) {
  if (!out) return fail(PRAD_E_ARG, "ngtdm is NULL");
  if (Ng < 1) return fail(PRAD_E_ARG, "Ng must be >= 1");
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Call k;
  PRAD_TRY(setup_call(k, image, mask, size, Nd, angles, Na, Nvox, voxels, kernelRadius, force2Ddim, s));
  PRAD_TRY(c.begin_call(s));
  bool done = false;
  if (!c.deferred && !voxels && k.Na > 26) {
    // more neighbours than the 3^3 cube: when they are a whole box (distances [1, 2], [1, 2, 3], ...) the separable box sums of
    // the pairs tier take 0.4 ms per 256^3 volume where the byte kernel visits 124 neighbours per voxel (1.9 ms) and the generic
    // kernels 342 (24 ms)
    PRAD_TRY(pairs_neigh(k, true, Ng, 0, out, &done, true));
    if (done) {
      c.last_path = "pairs";
      PRAD_TRY(c.end_call(s));
      return PRAD_OK;
    }
  }
  PRAD_TRY(neigh_try_ngtdm(k.c, k.s, k.g, k.vm, k.image, k.mask, k.angles_h, k.Na, Ng, out, k.flags_d, &done));
  if (done && c.deferred && !voxels) {   // enqueue only: the flags are latched for prad_deferred_status()
    PRAD_TRY(latch_neigh(c, k));
    c.last_path = "neigh";
    return c.end_call(s);
  }
  bool irregular = false;      // the byte kernels' pack already saw a level outside [1, Ng]: the tier would only find it again
  if (done) {
    PRAD_TRY(read_flags(k));
    done = (k.flags_h[0] == 0);
    irregular = !done;
  }
  if (!done && !c.deferred && !irregular) {
    PRAD_TRY(pairs_neigh(k, true, Ng, 0, out, &done));
    if (done) c.last_path = "pairs";
  } else if (done) {
    c.last_path = "neigh";
  }
  if (!done) {
    PRAD_TRY(generic_ngtdm(k, Ng, out));
    PRAD_TRY(read_flags(k));
    c.last_path = "generic";
  }
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  return k.flags_h[1] ? PRAD_INDEX_ERROR : PRAD_OK;
}

// GLDM and NGTDM of a segment from one pass over the neighbourhoods (segment mode; prad_calculate_gldm_ngtdm_dev)
int texture_gldm_ngtdm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                       int Ng, int alpha, double *gldm, double *ngtdm, hipStream_t s) {
  if (!gldm || !ngtdm) return fail(PRAD_E_ARG, "gldm / ngtdm is NULL");
  if (Ng < 1) return fail(PRAD_E_ARG, "Ng must be >= 1");
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  bool done = false;
  {
    Call k;
    PRAD_TRY(setup_call(k, image, mask, size, Nd, angles, Na, 1, nullptr, 0, -1, s));
    PRAD_TRY(c.begin_call(s));
    PRAD_TRY(neigh_try_both(k.c, k.s, k.g, k.vm, k.image, k.mask, k.angles_h, k.Na, Ng, alpha, gldm, ngtdm, k.flags_d, &done));
    if (done && c.deferred) {   // enqueue only: the flags are latched for prad_deferred_status()
      PRAD_TRY(latch_neigh(c, k));
      c.last_path = "neigh";
      return c.end_call(s);
    }
    if (done) {
      PRAD_TRY(read_flags(k));
      done = (k.flags_h[0] == 0);
      if (done) {
        c.last_path = "neigh";
        PRAD_TRY(c.end_call(s));
        return k.flags_h[1] ? PRAD_INDEX_ERROR : PRAD_OK;
      }
    }
  }
  // not a volume for the packed-byte kernel (or irregular levels: the exact kernels say what the reference says)
  PRAD_TRY(texture_gldm(image, mask, size, Nd, angles, Na, Ng, alpha, 1, nullptr, 0, -1, gldm, s));
  return texture_ngtdm(image, mask, size, Nd, angles, Na, Ng, 1, nullptr, 0, -1, ngtdm, s);
}

// z-slab split of one large segment over several GPUs: integer accumulators of a plane range, summed by the
// caller across ranks (RCCL all-reduce of [Ng][Na+1] int64), then finalized once
int neigh_accumulate_i64(int family, const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                         const int *angles, int Na, int Ng, int alpha, int z_lo, int z_hi, long long *acc,
                         hipStream_t s) {
  if (!acc) return fail(PRAD_E_ARG, "acc is NULL");
  if (family != 0 && family != 1) return fail(PRAD_E_ARG, "family must be 0 (GLDM) or 1 (NGTDM)");
  if (Nd != 3) return fail(PRAD_E_UNSUPPORTED, "plane-range accumulators need a 3-D volume (Nd=%d)", Nd);
  if (Ng < 1) return fail(PRAD_E_ARG, "Ng must be >= 1");
  if (z_lo < 0 || z_hi > size[0] || z_lo > z_hi) return fail(PRAD_E_ARG, "plane range [%d, %d) outside 0..%d", z_lo, z_hi, size[0]);
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Call k;
  PRAD_TRY(setup_call(k, image, mask, size, Nd, angles, Na, 1, nullptr, 0, -1, s));
  PRAD_TRY(c.begin_call(s));
  bool done = false;
  void *raw = nullptr;
  if (family == 0)
    PRAD_TRY(neigh_accumulate<false>(k.c, k.s, k.g, k.vm, k.image, k.mask, k.angles_h, k.Na, Ng, alpha, z_lo, z_hi,
                                     k.flags_d, &raw, &done));
  else
    PRAD_TRY(neigh_accumulate<true>(k.c, k.s, k.g, k.vm, k.image, k.mask, k.angles_h, k.Na, Ng, 0, z_lo, z_hi,
                                    k.flags_d, &raw, &done));
  if (!done) {
    c.end_call(s);
    return fail(PRAD_E_UNSUPPORTED, "plane-range accumulators need Ng <= 255, Na <= %d and [Ng][Na+1] bins within 64 KiB of LDS", PRAD_MAX_NEIGH);
  }
  const long long nacc = (long long)Ng * (Na + 1);
  if (family == 0) {
    hipLaunchKernelGGL(widen_u32_kernel, dim3((unsigned)((nacc + 255) / 256)), dim3(256), 0, s, (const u32 *)raw, nacc, acc);
    PRAD_TRY(check_launch("widen_u32_kernel"));
  } else {
    PRAD_HIP(hipMemcpyAsync(acc, raw, sizeof(long long) * nacc, hipMemcpyDeviceToDevice, s));
  }
  PRAD_TRY(read_flags(k));
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  // a level outside 1..Ng under the mask: the reference fails the whole calculation (cmatrices.c:744,647)
  return k.flags_h[0] ? PRAD_INDEX_ERROR : PRAD_OK;
}

int neigh_finalize_i64(int family, const long long *acc, int Ng, int Na, double *out, hipStream_t s) {
  if (!acc || !out) return fail(PRAD_E_ARG, "acc/out is NULL");
  if (family != 0 && family != 1) return fail(PRAD_E_ARG, "family must be 0 (GLDM) or 1 (NGTDM)");
  if (Ng < 1 || Na < 1) return fail(PRAD_E_ARG, "Ng and Na must be >= 1");
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  PRAD_TRY(c.begin_call(s));
  if (family == 0) {
    const long long nacc = (long long)Ng * (Na + 1);
    u32 *acc32 = nullptr;
    int *flags = nullptr;
    PRAD_TRY(c.get<u32>("gldm_acc", (size_t)nacc, &acc32));
    PRAD_TRY(c.get<int>("flags", 4, &flags));
    PRAD_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, s));
    hipLaunchKernelGGL(narrow_i64_kernel, dim3((unsigned)((nacc + 255) / 256)), dim3(256), 0, s, acc, nacc, acc32, flags);
    PRAD_TRY(check_launch("narrow_i64_kernel"));
    PRAD_TRY(neigh_finalize_gldm(&c, s, acc32, Ng, Na, out));
  } else {
    PRAD_TRY(neigh_finalize_ngtdm(&c, s, reinterpret_cast<const u64 *>(acc), Ng, Na, out));
  }
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  return PRAD_OK;
}

// ------------------------------------------------------------------------------------------------
// host-pointer staging
// ------------------------------------------------------------------------------------------------
struct Staged {
  int32_t *image = nullptr;
  uint8_t *mask = nullptr;
  int *voxels = nullptr;
};

// Large host arrays (the drop-in boundary hands over pageable numpy memory: 671 MB for a 512^3 case) reach the device
// through a ring of pinned buffers: kStageThreads host threads copy 16 MB chunks into the ring while the DMA engine
// drains the chunks already filled.  Measured on the MI355X host (scripts/h2d_bench.hip): 52 GB/s, against 24-28 GB/s for
// hipMemcpyAsync from pageable memory (the runtime's own single-threaded staging) and 17 GB/s for a first-touch
// hipMemcpy (which pins the user's pages on the fly).
constexpr size_t kStageChunk = (size_t)16 << 20;
constexpr int kStageRing = 6;
constexpr int kStageThreads = 4;
constexpr size_t kStageMin = (size_t)32 << 20;   // smaller copies are not worth the thread start-up

struct StageRing {
  char *buf[kStageRing] = {};
  hipEvent_t ev[kStageRing] = {};
  bool busy[kStageRing] = {};
  size_t next = 0;
};
StageRing &stage_ring() {
  static thread_local StageRing r;
  return r;
}

int staged_h2d(Context &c, void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (bytes < kStageMin || getenv("PRAD_NO_STAGING")) {
    PRAD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    return PRAD_OK;
  }
  StageRing &r = stage_ring();
  for (int i = 0; i < kStageRing; i++) {
    if (!r.buf[i]) {
      void *p = nullptr;
      PRAD_TRY(c.get_pinned(("stage_ring" + std::to_string(i)).c_str(), kStageChunk, &p));
      r.buf[i] = (char *)p;
      PRAD_HIP(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
    }
  }
  const size_t nchunks = (bytes + kStageChunk - 1) / kStageChunk;
  const size_t first = r.next;
  const bool dbg = getenv("PRAD_STAGE_DEBUG") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  double t_wait_slot = 0, t_wait_copy = 0;
  int nthreads = kStageThreads;
  if (const char *e = getenv("PRAD_STAGE_THREADS")) nthreads = std::max(1, std::min(16, atoi(e)));   // tuning override
  std::atomic<size_t> go{0}, done{0};
  std::atomic<bool> quit{false};
  std::vector<std::thread> workers;
  for (int w = 0; w < nthreads; w++)
    workers.emplace_back([&, w]() {
      size_t seen = 0;
      for (;;) {
        size_t g;
        while ((g = go.load(std::memory_order_acquire)) == seen)
          if (quit.load(std::memory_order_acquire)) return;
        for (size_t ch = seen; ch < g; ch++) {
          const size_t off = ch * kStageChunk, len = std::min(kStageChunk, bytes - off);
          const size_t sl = ((len + nthreads - 1) / nthreads + 63) & ~(size_t)63;
          const size_t a = std::min(len, (size_t)w * sl), b = std::min(len, a + sl);
          if (b > a) memcpy(r.buf[(first + ch) % kStageRing] + a, (const char *)src + off + a, b - a);
          done.fetch_add(1, std::memory_order_release);
        }
        seen = g;
      }
    });
  hipError_t err = hipSuccess;
  for (size_t ch = 0; ch < nchunks && err == hipSuccess; ch++) {
    const int slot = (int)((first + ch) % kStageRing);
    const auto t0 = std::chrono::steady_clock::now();
    if (r.busy[slot]) err = hipEventSynchronize(r.ev[slot]);   // the DMA that last read this slot has finished
    const auto t1 = std::chrono::steady_clock::now();
    go.store(ch + 1, std::memory_order_release);
    while (done.load(std::memory_order_acquire) < (ch + 1) * (size_t)nthreads) {}
    if (dbg) {
      t_wait_slot += std::chrono::duration<double>(t1 - t0).count();
      t_wait_copy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    }
    const size_t off = ch * kStageChunk, len = std::min(kStageChunk, bytes - off);
    if (err == hipSuccess) err = hipMemcpyAsync((char *)dst + off, r.buf[slot], len, hipMemcpyHostToDevice, s);
    if (err == hipSuccess) err = hipEventRecord(r.ev[slot], s);
    r.busy[slot] = true;
  }
  quit.store(true, std::memory_order_release);
  for (auto &t : workers) t.join();
  r.next = (first + nchunks) % kStageRing;
  if (dbg)
    fprintf(stderr, "staged_h2d %zu MB: %.2f ms enqueue (slot waits %.2f ms, host copies %.2f ms)\n", bytes >> 20,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3, t_wait_slot * 1e3, t_wait_copy * 1e3);
  if (err != hipSuccess) return fail(PRAD_E_HIP, "staged host-to-device copy failed: %s", hipGetErrorString(err));
  return PRAD_OK;
}

int stage_inputs(Context &c, const int32_t *image, const uint8_t *mask, const int *size, int Nd, int Nvox,
                 const int *voxels, Staged *st) {
  PRAD_TRY(c.ensure_device());
  if (!image || !mask) return fail(PRAD_E_ARG, "image/mask is NULL");
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  // 16-byte aligned slots so the vectorised pack path applies
  PRAD_TRY(c.get<int32_t>("h_image", (size_t)g.n + 16, &st->image));
  PRAD_TRY(c.get<uint8_t>("h_mask", (size_t)g.n + 64, &st->mask));
  PRAD_TRY(staged_h2d(c, st->image, image, sizeof(int32_t) * g.n, c.own_stream));
  PRAD_TRY(staged_h2d(c, st->mask, mask, g.n, c.own_stream));
  if (voxels) {
    if (Nvox < 1) return fail(PRAD_E_ARG, "Nvox=%d < 1", Nvox);
    PRAD_TRY(c.get<int>("h_voxels", (size_t)Nd * Nvox, &st->voxels));
    PRAD_HIP(hipMemcpyAsync(st->voxels, voxels, sizeof(int) * Nd * Nvox, hipMemcpyHostToDevice, c.own_stream));
  }
  return PRAD_OK;
}

int fetch_output(Context &c, const char *slot, double *host, size_t count, double **dev) {
  (void)host;
  return c.get<double>(slot, count, dev);
}

int copy_back(Context &c, double *host, const double *dev, size_t count) {
  PRAD_HIP(hipMemcpyAsync(host, dev, sizeof(double) * count, hipMemcpyDeviceToHost, c.own_stream));
  PRAD_HIP(hipStreamSynchronize(c.own_stream));
  return PRAD_OK;
}

#include "prad_voxel.h"      // fused voxel-based feature maps

}  // namespace

// =================================================================================================
// extern "C"
// =================================================================================================
namespace prad {
int zone_features_launch(Context &c, hipStream_t s, const double *P, int Ni, int Njcap, long long stride_i,
                         const double *jvals_d, const int *nj_dev, double *out_d, int *empty_d);   // prad_features.hip
}
#define ZM_FEATURES 16   // ZM_COUNT of kernels_features.h (static_assert there)
namespace {
template <typename T>
int launch_digitize(const T *image, const uint8_t *mask, long long n, const double *e_d, int nedges, int32_t *levels,
                    int *top, unsigned long long *counts_d, hipStream_t s) {
  // 4 workgroups per CU: every workgroup ends with global atomics on the same few words (its maximum, its nb counters), and
  // those serialise in L2 at ~25 ns each -- with 4096 workgroups they were half of the kernel (256^3 float64: 124 -> 57 us;
  // roi_minmax 59 -> 35 with 512 instead of 2048; profiles/r03_probes.md, section 10)
  static const int cap = getenv("PRAD_BIN_BLOCKS") ? atoi(getenv("PRAD_BIN_BLOCKS")) : 1024;
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, cap));
  const size_t edge_b = sizeof(double) * (size_t)nedges, cnt_b = 4 * sizeof(unsigned) * ((size_t)nedges + 1);
  const bool lds_edges = edge_b <= 60 * 1024;
  // counts in LDS while edges + four private tables fit 64 KB (nedges <= 2 700), otherwise straight global atomics
  // (that many levels spread the atomics over as many addresses)
  const size_t cnt3_b = 256 * sizeof(unsigned) * ((size_t)nedges + 1);     // a table per thread
  const int counts = !counts_d ? 0 : (nedges <= 48 ? 3 : ((lds_edges && edge_b + cnt_b <= 64 * 1024) ? 1 : 2));
#define PRAD_DIG(LE, CN)                                                                                              \
  hipLaunchKernelGGL((digitize_kernel<T, LE, CN>), dim3(gx), dim3(256),                                               \
                     (LE ? edge_b : 0) + (CN == 1 ? cnt_b : (CN == 3 ? cnt3_b : 0)), s, image, mask, n, e_d, nedges, levels, top, counts_d)
  if (lds_edges) {
    if (counts == 0) PRAD_DIG(true, 0);
    else if (counts == 1) PRAD_DIG(true, 1);
    else if (counts == 3) PRAD_DIG(true, 3);
    else PRAD_DIG(true, 2);
  } else {
    if (counts == 0) PRAD_DIG(false, 0);
    else PRAD_DIG(false, 2);
  }
#undef PRAD_DIG
  return check_launch("digitize_kernel");
}
}  // namespace

extern "C" {

const char *prad_version(void) { return kVersion; }
const char *prad_last_error(void) { return err_state().msg; }
const char *prad_last_path(void) { return ctx().last_path; }
const char *prad_last_variant(void) { return ctx().last_variant; }
#ifdef PRAD_FW_STAMPS
// instrumentation builds only (scripts/r05_fw_stamps.py): per-wave, per-phase cycle sums of the last sweep_fw_kernel launch
// (slot 1: the launch that carries the pack side job, slot 0: the one without)
int prad_debug_fw_stamps(unsigned long long *out, int slot, int nwaves) {
  if (!out || nwaves < 1 || nwaves > 8192 || slot < 0 || slot > 1) return PRAD_E_ARG;
  PRAD_HIP(hipDeviceSynchronize());
  PRAD_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(prad::prad_fw_stamps), sizeof(unsigned long long) * prad::FP_COUNT * (size_t)nwaves,
                               sizeof(unsigned long long) * prad::FP_COUNT * 8192 * (size_t)slot));
  return PRAD_OK;
}
#endif

int prad_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int prad_set_device(int device) {
  int n = prad_device_count();
  if (device < 0 || device >= n) return fail(PRAD_E_ARG, "device %d not in [0,%d)", device, n);
  Context &c = ctx();
  if (c.device != device) {
    if (pipe_state().pending.valid && c.device_set && hipSetDevice(c.device) == hipSuccess) {
      (void)pipeline_flush(c);   // a volume still pending on the old device: retire it there
      if (pipe_state().s) (void)hipStreamSynchronize(pipe_state().s);
      (void)fin_join(nullptr, true);
    }
    pipe_state().pending.valid = false;
    pipe_state().s = nullptr;
    pipe_state().ev = nullptr;
    pipe_state().fin = nullptr;      // (the side stream and its events belonged to the old device)
    for (int l = 0; l < 4; l++) pipe_state().fin_used[l] = false;
    c.own_stream = nullptr;  // streams belong to a device; new ones are created lazily
    for (int l = 0; l < PRAD_MAX_LANES; l++) {
      if (c.lane_stream[l]) (void)hipStreamSynchronize(c.lane_stream[l]);
      c.lane_stream[l] = nullptr;
      c.lane_in[l] = nullptr;
    }
    c.event_pool.clear();
    c.events_used = 0;
  }
  c.device = device;
  c.device_set = true;
  PRAD_HIP(hipSetDevice(device));
  return PRAD_OK;
}
int prad_get_device(void) { return ctx().device; }

long long prad_workspace_bytes(void) {
  long long total = 0;
  for (const auto &kv : ctx().bufs) total += (long long)kv.second.cap;
  return total;
}
namespace {
void release_image_queues();     // (prad_image_enqueue_dev's streams and events, defined with it below)
}
int prad_release_workspace(void) {
  Context &c = ctx();
  if (pipe_state().pending.valid) {        // the pending volume's buffers are about to go away: retire it first
    (void)pipeline_flush(c);
    if (pipe_state().s) (void)hipStreamSynchronize(pipe_state().s);
    (void)fin_join(nullptr, true);
    pipe_state().pending.valid = false;
  }
  glszm_state().valid = false;            // its zone list lives in the workspace
  for (auto &kv : c.bufs) {
    if (kv.second.p) (void)hipFree(kv.second.p);
    kv.second.p = nullptr;
    kv.second.cap = 0;
  }
  (void)hipDeviceSynchronize();           // nothing may still read the buffers that go away
  for (auto &kv : c.pinned) {
    if (kv.second.p) (void)hipHostFree(kv.second.p);
    kv.second.p = nullptr;
    kv.second.cap = 0;
  }
  c.bufs.clear();
  c.pinned.clear();
  if (c.arena) (void)hipHostFree(c.arena);     // (the result arena: outstanding result views of this thread die with it)
  c.arena = nullptr;
  c.arena_pos = 0;
  release_image_queues();
  c.angles_cached.clear();   // (the cached angle tables and the deferred flag lived in the workspace)
  StageRing &r = stage_ring();
  for (int i = 0; i < kStageRing; i++) {
    r.buf[i] = nullptr;
    r.busy[i] = false;
  }
  return PRAD_OK;
}

int prad_timing_begin(void) {
  Context &c = ctx();
  c.timing_only.clear();
  c.timing_accumulate = true;
  c.all_times.clear();
  c.all_calls.clear();
  c.events_used = 0;
  return PRAD_OK;
}
int prad_timing_begin_only(const char *kernel_family) {
  if (!kernel_family || !*kernel_family) return fail(PRAD_E_ARG, "timing: no kernel family named");
  int rc = prad_timing_begin();
  ctx().timing_only = kernel_family;
  return rc;
}
int prad_timing_count(const char *kernel_family) {
  Context &c = ctx();
  if (!kernel_family) return (int)c.all_calls.size();
  int n = 0;
  for (auto &t : c.all_times) n += t.family == kernel_family;
  return n;
}
int prad_timing_end(void) {
  Context &c = ctx();
  c.timing_only.clear();
  c.timing_accumulate = false;
  c.all_times.clear();
  c.all_calls.clear();
  return PRAD_OK;
}
int prad_timing_calls(void) { return (int)ctx().all_calls.size(); }
double prad_timing_ms(const char *family) {
  Context &c = ctx();
  double total = 0;
  if (!family) {
    for (auto &ab : c.all_calls) {
      float ms = 0;
      if (hipEventSynchronize(ab.second) != hipSuccess || hipEventElapsedTime(&ms, ab.first, ab.second) != hipSuccess) return -1.0;
      total += ms;
    }
    return total;
  }
  for (auto &t : c.all_times) {
    if (t.family != family) continue;
    float ms = 0;
    if (hipEventSynchronize(t.b) != hipSuccess || hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) return -1.0;
    total += ms;
  }
  return total;
}

int prad_set_deferred(int on) {
  Context &c = ctx();
  if (c.ensure_device() != PRAD_OK) return PRAD_E_HIP;
  if (on && !c.has("deferred_sticky")) {
    int *sticky = nullptr;   // first use on this device: a clean sticky flag (afterwards only the status query clears it)
    PRAD_TRY(c.get<int>("deferred_sticky", 16, &sticky));
    PRAD_HIP(hipMemset(sticky, 0, sizeof(int) * 16));
  }
  c.deferred = on != 0;
  return PRAD_OK;
}
int prad_set_lanes(int n) {
  Context &c = ctx();
  if (n < 0 || n > PRAD_MAX_LANES) return fail(PRAD_E_ARG, "lanes=%d outside [0, %d]", n, PRAD_MAX_LANES);
  PRAD_TRY(c.lanes_sync());
  c.lanes = n;          // 0: back to the default (PRAD_LANES or 2) at the next deferred call
  c.lane_seq = 0;
  return PRAD_OK;
}
int prad_set_deferred_mode(int mode) {
  if (mode < -1 || mode > 1) return fail(PRAD_E_ARG, "deferred mode %d (0 lanes, 1 pipeline, -1 environment default)", mode);
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  PipeState &ps = pipe_state();
  PRAD_TRY(pipeline_flush(c));
  if (ps.s) PRAD_HIP(hipStreamSynchronize(ps.s));
  PRAD_TRY(fin_join(nullptr, true));
  PRAD_TRY(c.lanes_sync());
  ps.mode = mode;
  return PRAD_OK;
}
int prad_set_workspace(int id) {
  if (id < 0 || id > 7) return fail(PRAD_E_ARG, "workspace %d outside [0, 7]", id);
  ctx().workspace = id;
  return PRAD_OK;
}
int prad_result_alloc(size_t bytes, void **out) {
  if (!out) return fail(PRAD_E_ARG, "result_alloc: out is NULL");
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  return c.arena_alloc(bytes, out);
}
int prad_deferred_join(void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  PipeState &ps = pipe_state();
  PRAD_TRY(pipeline_flush(c));
  if (ps.s && ps.s != (hipStream_t)stream) {   // the flushed work sits on the stream of its pack
    if (!ps.ev) PRAD_HIP(hipEventCreateWithFlags(&ps.ev, hipEventDisableTiming));
    PRAD_HIP(hipEventRecord(ps.ev, ps.s));
    PRAD_HIP(hipStreamWaitEvent((hipStream_t)stream, ps.ev, 0));
  }
  PRAD_TRY(fin_join((hipStream_t)stream, false));
  return c.lanes_join((hipStream_t)stream);
}
int prad_deferred_mark(int *flag, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!flag || !c.in_arena(flag, sizeof(int))) return fail(PRAD_E_ARG, "deferred_mark: flag must lie in the result arena");
  PRAD_TRY(pipeline_flush(c));
  PRAD_TRY(fin_join((hipStream_t)stream, false));
  int *sticky = nullptr;
  PRAD_TRY(c.get<int>("deferred_sticky", 16, &sticky));
  PRAD_HIP(hipMemcpyAsync(flag, sticky, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return PRAD_OK;
}
int prad_deferred_status(void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  PRAD_TRY(pipeline_flush(c));
  if (pipe_state().s) PRAD_HIP(hipStreamSynchronize(pipe_state().s));
  PRAD_TRY(fin_join(nullptr, true));
  PRAD_HIP(hipStreamSynchronize((hipStream_t)stream));
  PRAD_TRY(c.lanes_sync());
  if (!c.has("deferred_sticky")) return PRAD_OK;  // no deferred call yet
  int *sticky = nullptr;
  PRAD_TRY(c.get<int>("deferred_sticky", 16, &sticky));
  // (on the caller's stream, through pinned memory: a plain hipMemcpy is a null-stream operation that waits for the
  // whole device -- with six case threads each asking once per derived image it serialised the batch, 65 -> 37 cases/s)
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned("sticky_h", 64, &pin));
  PRAD_HIP(hipMemcpyAsync(pin, sticky, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  PRAD_HIP(hipStreamSynchronize((hipStream_t)stream));
  const int h = *(const int *)pin;
  if (h) {
    PRAD_HIP(hipMemsetAsync(sticky, 0, sizeof(int), (hipStream_t)stream));
    PRAD_HIP(hipStreamSynchronize((hipStream_t)stream));
    return fail(PRAD_E_DEFERRED, "a deferred GLCM/GLRLM call saw masked levels outside [1, Ng]; repeat it synchronously");
  }
  return PRAD_OK;
}

double prad_last_device_ms(void) {
  Context &c = ctx();
  if (!c.call_timed) return -1.0;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c.call_a, c.call_b) != hipSuccess) return -1.0;
  return ms;
}

double prad_last_kernel_ms(const char *family) {
  Context &c = ctx();
  double total = 0;
  bool any = false;
  for (auto &t : c.times) {
    if (family && t.family != family) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) return -1.0;
    total += ms;
    any = true;
  }
  return any ? total : -1.0;
}

int prad_get_angle_count(const int *size, const int *distances, int Nd, int Ndist, int bidirectional,
                         int force2Ddim) {
  if (!size || !distances || Nd < 1 || Ndist < 1) return 0;
  return angle_count(size, distances, Nd, Ndist, bidirectional, force2Ddim);
}

int prad_build_angles(const int *size, const int *distances, int Nd, int Ndist, int force2Ddim, int Na,
                      int *angles) {
  if (!size || !distances || !angles || Nd < 1 || Ndist < 1) return 1;
  return angle_build(size, distances, Nd, Ndist, force2Ddim, Na, angles);
}

// ---- GLCM / GLRLM ---------------------------------------------------------------------------
int prad_calculate_glcm_glrlm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                                  const int *angles, int Na, int Ng, int Nr, int Nvox, const int *voxels,
                                  int kernelRadius, int force2Ddim, double *glcm, double *glrlm, void *stream) {
  return texture_pairs_runs(image, mask, size, Nd, angles, Na, Ng, Nr, Nvox, voxels, kernelRadius, force2Ddim, glcm,
                            glrlm, (hipStream_t)stream);
}
int prad_calculate_glcm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                            int Na, int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                            double *glcm, void *stream) {
  if (!glcm) return fail(PRAD_E_ARG, "glcm is NULL");
  return texture_pairs_runs(image, mask, size, Nd, angles, Na, Ng, 1, Nvox, voxels, kernelRadius, force2Ddim, glcm,
                            nullptr, (hipStream_t)stream);
}
int prad_calculate_glrlm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                             int Na, int Ng, int Nr, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                             double *glrlm, void *stream) {
  if (!glrlm) return fail(PRAD_E_ARG, "glrlm is NULL");
  return texture_pairs_runs(image, mask, size, Nd, angles, Na, Ng, Nr, Nvox, voxels, kernelRadius, force2Ddim,
                            nullptr, glrlm, (hipStream_t)stream);
}

int prad_calculate_glcm_glrlm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                              int Na, int Ng, int Nr, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                              double *glcm, double *glrlm) {
  Context &c = ctx();
  Staged st;
  PRAD_TRY(stage_inputs(c, image, mask, size, Nd, Nvox, voxels, &st));
  if (Ng < 1 || Nvox < 1 || Na < 1) return fail(PRAD_E_ARG, "Ng/Nvox/Na must be >= 1");
  const size_t nglcm = glcm ? (size_t)Nvox * Ng * Ng * Na : 0;
  const size_t nglrlm = glrlm ? (size_t)Nvox * Ng * (size_t)std::max(Nr, 1) * Na : 0;
  double *d_glcm = nullptr, *d_glrlm = nullptr;
  if (glcm) PRAD_TRY(fetch_output(c, "o_glcm", glcm, nglcm, &d_glcm));
  if (glrlm) PRAD_TRY(fetch_output(c, "o_glrlm", glrlm, nglrlm, &d_glrlm));
  int rc = texture_pairs_runs(st.image, st.mask, size, Nd, angles, Na, Ng, Nr, Nvox, st.voxels, kernelRadius,
                              force2Ddim, d_glcm, d_glrlm, c.own_stream);
  if (rc != PRAD_OK) return rc;
  if (glcm) PRAD_TRY(copy_back(c, glcm, d_glcm, nglcm));
  if (glrlm) PRAD_TRY(copy_back(c, glrlm, d_glrlm, nglrlm));
  return PRAD_OK;
}
int prad_calculate_glcm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                        int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, double *glcm) {
  if (!glcm) return fail(PRAD_E_ARG, "glcm is NULL");
  return prad_calculate_glcm_glrlm(image, mask, size, Nd, angles, Na, Ng, 1, Nvox, voxels, kernelRadius, force2Ddim,
                                   glcm, nullptr);
}
int prad_calculate_glrlm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                         int Na, int Ng, int Nr, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                         double *glrlm) {
  if (!glrlm) return fail(PRAD_E_ARG, "glrlm is NULL");
  return prad_calculate_glcm_glrlm(image, mask, size, Nd, angles, Na, Ng, Nr, Nvox, voxels, kernelRadius, force2Ddim,
                                   nullptr, glrlm);
}

// ---- GLDM -----------------------------------------------------------------------------------
int prad_calculate_gldm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                            int Na, int Ng, int alpha, int Nvox, const int *voxels, int kernelRadius,
                            int force2Ddim, double *gldm, void *stream) {
  return texture_gldm(image, mask, size, Nd, angles, Na, Ng, alpha, Nvox, voxels, kernelRadius, force2Ddim, gldm,
                      (hipStream_t)stream);
}
int prad_calculate_gldm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                        int Ng, int alpha, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                        double *gldm) {
  if (!gldm) return fail(PRAD_E_ARG, "gldm is NULL");
  Context &c = ctx();
  Staged st;
  PRAD_TRY(stage_inputs(c, image, mask, size, Nd, Nvox, voxels, &st));
  if (Ng < 1 || Nvox < 1 || Na < 1) return fail(PRAD_E_ARG, "Ng/Nvox/Na must be >= 1");
  const size_t n = (size_t)Nvox * Ng * (2 * (size_t)Na + 1);
  double *d = nullptr;
  PRAD_TRY(fetch_output(c, "o_gldm", gldm, n, &d));
  int rc = texture_gldm(st.image, st.mask, size, Nd, angles, Na, Ng, alpha, Nvox, st.voxels, kernelRadius,
                        force2Ddim, d, c.own_stream);
  if (rc != PRAD_OK) return rc;
  return copy_back(c, gldm, d, n);
}

int prad_calculate_gldm_ngtdm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                                  int Na, int Ng, int alpha, double *gldm, double *ngtdm, void *stream) {
  return texture_gldm_ngtdm(image, mask, size, Nd, angles, Na, Ng, alpha, gldm, ngtdm, (hipStream_t)stream);
}

// ---- NGTDM ----------------------------------------------------------------------------------
int prad_calculate_ngtdm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                             int Na, int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                             double *ngtdm, void *stream) {
  return texture_ngtdm(image, mask, size, Nd, angles, Na, Ng, Nvox, voxels, kernelRadius, force2Ddim, ngtdm,
                       (hipStream_t)stream);
}
int prad_calculate_ngtdm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                         int Na, int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                         double *ngtdm) {
  if (!ngtdm) return fail(PRAD_E_ARG, "ngtdm is NULL");
  Context &c = ctx();
  Staged st;
  PRAD_TRY(stage_inputs(c, image, mask, size, Nd, Nvox, voxels, &st));
  if (Ng < 1 || Nvox < 1 || Na < 1) return fail(PRAD_E_ARG, "Ng/Nvox/Na must be >= 1");
  const size_t n = (size_t)Nvox * Ng * 3;
  double *d = nullptr;
  PRAD_TRY(fetch_output(c, "o_ngtdm", ngtdm, n, &d));
  int rc = texture_ngtdm(st.image, st.mask, size, Nd, angles, Na, Ng, Nvox, st.voxels, kernelRadius, force2Ddim, d,
                         c.own_stream);
  if (rc != PRAD_OK) return rc;
  return copy_back(c, ngtdm, d, n);
}

// ---- plane-range accumulators (one segment over several GPUs) ------------------------------
int prad_neigh_accumulate_dev(int family, const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                              const int *angles, int Na, int Ng, int alpha, int z_lo, int z_hi, long long *acc,
                              void *stream) {
  return neigh_accumulate_i64(family, image, mask, size, Nd, angles, Na, Ng, alpha, z_lo, z_hi, acc,
                              (hipStream_t)stream);
}
int prad_neigh_finalize_dev(int family, const long long *acc, int Ng, int Na, double *out, void *stream) {
  return neigh_finalize_i64(family, acc, Ng, Na, out, (hipStream_t)stream);
}

// ---- GLSZM ----------------------------------------------------------------------------------
int prad_calculate_glszm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                             int Na, int Ng, int Ns, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                             long long *nzones, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!image || !mask || !angles || Na < 1 || Ng < 1 || Nvox < 1) return fail(PRAD_E_ARG, "bad GLSZM arguments");
  hipStream_t s = (hipStream_t)stream;
  PRAD_TRY(c.begin_call(s));
  int rc = glszm_zones(c, s, g, image, mask, angles, Na, Ng, Ns, Nvox, voxels, kernelRadius, force2Ddim, nzones);
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  return rc;
}
int prad_glszm_features_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                            int Ng, int Ns, double *out, int *empty, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!image || !mask || !angles || !out || !empty || Na < 1 || Ng < 1 || Ns < 1) return fail(PRAD_E_ARG, "bad GLSZM arguments");
  hipStream_t s = (hipStream_t)stream;
  PRAD_TRY(c.begin_call(s));
  {   // (returns the largest zone, 0 here: nothing came back from the device; negative = error)
    const int zrc = glszm_zones(c, s, g, image, mask, angles, Na, Ng, Ns, 1, nullptr, 0, -1, nullptr, true);
    if (zrc < 0) return zrc;
  }
  GlszmState &st = glszm_state();
  // distinct sizes sum to at most n voxels, so there are fewer than sqrt(2 n) + 1 of them
  const int kcap = (int)std::min<long long>(g.n, (long long)std::sqrt(2.0 * (double)g.n) + 2);
  int *stats = nullptr, *flags_d = nullptr, *large_sorted = nullptr, *meta = nullptr, *err = nullptr;
  double *jv = nullptr, *P = nullptr, *d_out = nullptr;
  PRAD_TRY(c.get<int>("glszm_stats", 8, &stats));
  PRAD_TRY(c.get<int>("flags", 4, &flags_d));
  PRAD_TRY(c.get<int>("glszm_small_rank", PRAD_SMALL_SIZES, &st.small_rank));
  PRAD_TRY(c.get<int>("glszm_large_sorted", PRAD_RANK_LARGE + 1, &large_sorted));
  PRAD_TRY(c.get<int>("glszm_meta", 8, &meta));
  PRAD_TRY(c.get<int>("glszm_err", 4, &err));
  PRAD_TRY(c.get<double>("glszm_jvals", (size_t)kcap, &jv));
  PRAD_TRY(c.get<double>("glszm_compact", (size_t)Ng * kcap, &P));
  PRAD_TRY(c.get<double>("glszm_feat_out", ZM_FEATURES + 4, &d_out));
  int *d_empty = reinterpret_cast<int *>(d_out + ZM_FEATURES + 2);
  PRAD_TRY(ZeroBatch().add(err, sizeof(int) * 4).add(P, sizeof(double) * (size_t)Ng * kcap).launch(s));
  {
    Timed t(c, "glszm", s);
    hipLaunchKernelGGL(glszm_rank_kernel, dim3(1), dim3(1024), 0, s, (const unsigned *)st.small_bits, (const int *)st.large_list,
                       (const int *)st.large_count, st.large_cap, (const int *)flags_d,
                       (const unsigned long long *)(stats + 2), 2LL * Ns, kcap, st.small_rank, large_sorted, jv, meta);
    PRAD_TRY(check_launch("glszm_rank_kernel"));
    const int RL = Ng <= 8192 ? std::max(1, std::min(kcap, 8192 / Ng)) : 0;
    hipLaunchKernelGGL(glszm_fill_compact_kernel, dim3(std::min(glszm_grid(g.n), st.parent ? glszm_fill_blocks() : 2048u)), dim3(256), sizeof(unsigned) * Ng * RL,
                       s, g.n, st.labels, st.sizes, image, Ng, 0, RL, st.small_rank, 0, large_sorted, 0, P, err,
                       (const int *)st.parent, (const int *)st.rootctl, (const int *)meta, kcap, (const unsigned *)st.tinfo);
    PRAD_TRY(check_launch("glszm_fill_compact_kernel"));
  }
  const size_t nb = sizeof(double) * (ZM_FEATURES + 1);
  const bool enq = c.deferred && c.in_arena(out, nb) && c.in_arena(empty, sizeof(int));
  const bool direct = enq && Context::zero_copy();        // values, verdict and flag stored straight into the arena
  PRAD_TRY(zone_features_launch(c, s, P, Ng, kcap, (long long)kcap, jv, meta + 2, direct ? out : d_out, direct ? empty : d_empty));
  hipLaunchKernelGGL(glszm_verdict_kernel, dim3(1), dim3(1), 0, s, (const int *)meta, (const int *)err,
                     (direct ? out : d_out) + ZM_FEATURES);
  PRAD_TRY(check_launch("glszm_verdict_kernel"));
  PRAD_TRY(c.end_call(s));
  c.last_path = "glszm-unionfind";
  if (direct) return PRAD_OK;
  if (enq) {
    PRAD_HIP(hipMemcpyAsync(out, d_out, nb, hipMemcpyDeviceToHost, s));
    PRAD_HIP(hipMemcpyAsync(empty, d_empty, sizeof(int), hipMemcpyDeviceToHost, s));
    return PRAD_OK;
  }
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned("glszm_feat_pin", sizeof(double) * (ZM_FEATURES + 4), &pin));
  PRAD_HIP(hipMemcpyAsync(pin, d_out, sizeof(double) * (ZM_FEATURES + 3), hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  memcpy(out, pin, nb);
  memcpy(empty, (const char *)pin + sizeof(double) * (ZM_FEATURES + 2), sizeof(int));
  const int verdict = (int)out[ZM_FEATURES];
  if (verdict & 2) return fail(PRAD_E_INDEX, "GLSZM: zone list would overflow the reference's Ns-sized scratch (Ns=%d)", Ns);
  if (verdict) return fail(PRAD_E_UNSUPPORTED, "GLSZM features: the device-side ranking declined (verdict %d); use prad_calculate_glszm_dev", verdict);
  return PRAD_OK;
}
int prad_calculate_glszm(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                         int Na, int Ng, int Ns, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                         long long *nzones) {
  Context &c = ctx();
  Staged st;
  PRAD_TRY(stage_inputs(c, image, mask, size, Nd, Nvox, voxels, &st));
  return prad_calculate_glszm_dev(st.image, st.mask, size, Nd, angles, Na, Ng, Ns, Nvox, st.voxels, kernelRadius,
                                  force2Ddim, nzones, c.own_stream);
}
int prad_fill_glszm_dev(double *glszm, int Nvox, int Ng, int maxRegion, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!glszm || Ng < 1 || maxRegion < 1) return fail(PRAD_E_ARG, "bad fill_glszm arguments");
  return glszm_fill(c, (hipStream_t)stream, glszm, Nvox, Ng, maxRegion);
}
int prad_fill_glszm(double *glszm, int Nvox, int Ng, int maxRegion) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!glszm || Ng < 1 || maxRegion < 1 || Nvox < 1) return fail(PRAD_E_ARG, "bad fill_glszm arguments");
  const size_t n = (size_t)Nvox * Ng * maxRegion;
  double *d = nullptr;
  PRAD_TRY(c.get<double>("o_glszm", n, &d));
  int rc = glszm_fill(c, c.own_stream, d, Nvox, Ng, maxRegion);
  if (rc != PRAD_OK) return rc;
  return copy_back(c, glszm, d, n);
}
// ---- fused voxel-based GLCM features -----------------------------------------------------------
int prad_voxel_glcm_features_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                                 const int *angles, int Na, int Ng, int Nvox, const int *voxels, int kernelRadius,
                                 int force2Ddim, int symmetric, const int *feature_ids, int nfeat, double *out,
                                 uint32_t *empty_mask, uint32_t *any_nonempty, void *stream) {
  return voxel_glcm_features_dev(image, mask, size, Nd, angles, Na, Ng, Nvox, voxels, kernelRadius, force2Ddim,
                                 symmetric, feature_ids, nfeat, out, empty_mask, any_nonempty, (hipStream_t)stream);
}
int prad_voxel_glcm_features(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                             int Na, int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                             int symmetric, const int *feature_ids, int nfeat, double *out, uint32_t *empty_mask,
                             uint32_t *any_nonempty) {
  Context &c = ctx();
  Staged st;
  PRAD_TRY(stage_inputs(c, image, mask, size, Nd, Nvox, voxels, &st));
  if (!voxels || !out || nfeat < 1) return fail(PRAD_E_ARG, "voxel_glcm: bad arguments");
  double *d_out = nullptr;
  unsigned *d_masks = nullptr;
  PRAD_TRY(c.get<double>("o_vox", (size_t)nfeat * Nvox, &d_out));
  PRAD_TRY(c.get<unsigned>("o_voxmask", (size_t)Nvox + 1, &d_masks));
  int rc = voxel_glcm_features_dev(st.image, st.mask, size, Nd, angles, Na, Ng, Nvox, st.voxels, kernelRadius,
                                   force2Ddim, symmetric, feature_ids, nfeat, d_out, d_masks + 1, d_masks, c.own_stream);
  if (rc != PRAD_OK) return rc;
  PRAD_TRY(copy_back(c, out, d_out, (size_t)nfeat * Nvox));
  if (empty_mask) PRAD_HIP(hipMemcpy(empty_mask, d_masks + 1, sizeof(unsigned) * Nvox, hipMemcpyDeviceToHost));
  if (any_nonempty) PRAD_HIP(hipMemcpy(any_nonempty, d_masks, sizeof(unsigned), hipMemcpyDeviceToHost));
  return PRAD_OK;
}

__global__ void flag_to_double_kernel(const int *__restrict__ flag, double *__restrict__ out) { out[0] = flag[0] ? 1.0 : 0.0; }

int prad_glcm_mcc_dev(const double *glcm, int Ng, int Na, int symmetric, double *out, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!glcm || !out || Ng < 1 || Na < 1) return fail(PRAD_E_ARG, "glcm_mcc: bad arguments");
  const int nmax = std::min(Ng, PRAD_MCC_NMAX);
  size_t lds = mcc_scratch_bytes(Ng, nmax);
  if (lds > 150 * 1024) return fail(PRAD_E_UNSUPPORTED, "glcm_mcc: Ng=%d exceeds the LDS scratch", Ng);
  const size_t staged_lds = ((lds + 15) & ~(size_t)15) + sizeof(double) * (size_t)Ng * Ng;   // + the angle's counts
  const int staged = staged_lds <= 150 * 1024;
  if (staged) lds = staged_lds;
  hipStream_t s = (hipStream_t)stream;
  double *d_out = nullptr;
  int *d_flag = nullptr;
  PRAD_TRY(c.get<double>("mcc_out", (size_t)Na + 1, &d_out));
  PRAD_TRY(c.get<int>("mcc_flag", 4, &d_flag));
  PRAD_HIP(hipMemsetAsync(d_flag, 0, sizeof(int) * 4, s));
  {
    Timed t(c, "features", s);
    PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&glcm_matrix_mcc_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(glcm_matrix_mcc_kernel, dim3(Na), dim3(PRAD_MCC_BT), lds, s, glcm, Ng, Na, symmetric, nmax, staged, d_out, d_flag);
    PRAD_TRY(check_launch("glcm_matrix_mcc_kernel"));
  }
  if (c.deferred && c.in_arena(out, sizeof(double) * ((size_t)Na + 1))) {
    // enqueue only: out[Na] tells afterwards whether more than PRAD_MCC_NMAX levels occurred (then out[0..Na) is void)
    hipLaunchKernelGGL(flag_to_double_kernel, dim3(1), dim3(1), 0, s, (const int *)d_flag, d_out + Na);
    PRAD_TRY(check_launch("flag_to_double_kernel"));
    PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * ((size_t)Na + 1), hipMemcpyDeviceToHost, s));
    return PRAD_OK;
  }
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned("mcc_pin", sizeof(double) * ((size_t)Na + 2), &pin));
  hipLaunchKernelGGL(flag_to_double_kernel, dim3(1), dim3(1), 0, s, (const int *)d_flag, d_out + Na);
  PRAD_TRY(check_launch("flag_to_double_kernel"));
  PRAD_HIP(hipMemcpyAsync(pin, d_out, sizeof(double) * ((size_t)Na + 1), hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  memcpy(out, pin, sizeof(double) * Na);
  const int flag = ((const double *)pin)[Na] != 0.0;
  if (flag) return fail(PRAD_E_UNSUPPORTED, "glcm_mcc: more than %d grey levels occur; use the host route", PRAD_MCC_NMAX);
  return PRAD_OK;
}

int prad_voxel_glcm_mcc_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                            int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, int symmetric,
                            double *out, void *stream) {
  return voxel_glcm_mcc_dev(image, mask, size, Nd, angles, Na, Ng, Nvox, voxels, kernelRadius, force2Ddim, symmetric, out,
                            (hipStream_t)stream);
}
int prad_voxel_glcm_mcc(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                        int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, int symmetric, double *out) {
  Context &c = ctx();
  Staged st;
  PRAD_TRY(stage_inputs(c, image, mask, size, Nd, Nvox, voxels, &st));
  if (!voxels || !out) return fail(PRAD_E_ARG, "voxel_glcm_mcc: bad arguments");
  double *d_out = nullptr;
  PRAD_TRY(c.get<double>("o_mcc", (size_t)Nvox, &d_out));
  int rc = voxel_glcm_mcc_dev(st.image, st.mask, size, Nd, angles, Na, Ng, Nvox, st.voxels, kernelRadius, force2Ddim,
                              symmetric, d_out, c.own_stream);
  if (rc != PRAD_OK) return rc;
  return copy_back(c, out, d_out, (size_t)Nvox);
}

int prad_voxel_texture_features_dev(int family, const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                                    const int *angles, int Na, int Ng, int alpha, int Nvox, const int *voxels,
                                    int kernelRadius, int force2Ddim, const int *feature_ids, int nfeat, double *out,
                                    void *stream) {
  return voxel_texture_features_dev(family, image, mask, size, Nd, angles, Na, Ng, alpha, Nvox, voxels, kernelRadius,
                                    force2Ddim, feature_ids, nfeat, out, (hipStream_t)stream);
}

// ---- on-device discretisation ----------------------------------------------------------------
int prad_roi_minmax_dev(const void *image, int dtype, const uint8_t *mask, long long n, double *minmax, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!image || !mask || !minmax || n < 1) return fail(PRAD_E_ARG, "roi_minmax: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long *keys = nullptr;
  PRAD_TRY(c.get<unsigned long long>("minmax_keys", 2, &keys));
  // (pinned staging: a copy from / to pageable memory goes through the runtime's bounce buffer and blocks the host)
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned("bin_pin", 4096, &pin));
  unsigned long long *init = (unsigned long long *)pin, *outk = init + 2;
  init[0] = ~0ull;
  init[1] = 0ull;
  PRAD_HIP(hipMemcpyAsync(keys, init, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
  static const int cap = getenv("PRAD_MINMAX_BLOCKS") ? atoi(getenv("PRAD_MINMAX_BLOCKS")) : 512;
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, cap));
  switch (dtype) {
    case 0: hipLaunchKernelGGL(roi_minmax_kernel<float>, dim3(gx), dim3(256), 0, s, (const float *)image, mask, n, keys); break;
    case 1: hipLaunchKernelGGL(roi_minmax_kernel<double>, dim3(gx), dim3(256), 0, s, (const double *)image, mask, n, keys); break;
    case 2: hipLaunchKernelGGL(roi_minmax_kernel<int>, dim3(gx), dim3(256), 0, s, (const int *)image, mask, n, keys); break;
    case 3: hipLaunchKernelGGL(roi_minmax_kernel<short>, dim3(gx), dim3(256), 0, s, (const short *)image, mask, n, keys); break;
    default: return fail(PRAD_E_ARG, "roi_minmax: dtype %d", dtype);
  }
  PRAD_TRY(check_launch("roi_minmax_kernel"));
  PRAD_HIP(hipMemcpyAsync(outk, keys, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  if (outk[1] == 0ull) return fail(PRAD_E_ARG, "roi_minmax: empty ROI");
  minmax[0] = f64_unkey(outk[0]);
  minmax[1] = f64_unkey(outk[1]);
  return PRAD_OK;
}

int prad_digitize_counts_dev(const void *image, int dtype, const uint8_t *mask, long long n, const double *edges,
                             int nedges, int32_t *levels, int *max_level, long long *counts, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!image || !mask || !edges || !levels || n < 1 || nedges < 1) return fail(PRAD_E_ARG, "digitize: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  double *e_d = nullptr;
  int *top = nullptr;
  unsigned long long *cnt_d = nullptr;
  PRAD_TRY(c.get<double>("bin_edges", (size_t)nedges, &e_d));
  PRAD_TRY(c.get<int>("bin_top", 1, &top));
  // host <-> device traffic of this call through one pinned block: [edges (nedges doubles) | top (8 B) | counts]
  const size_t pin_bytes = sizeof(double) * ((size_t)nedges + 1) + sizeof(long long) * ((size_t)nedges + 2);
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned("digitize_pin", pin_bytes, &pin));
  double *e_h = (double *)pin;
  int *top_h = (int *)(e_h + nedges);
  long long *cnt_h = (long long *)(e_h + nedges + 1);
  memcpy(e_h, edges, sizeof(double) * nedges);
  PRAD_HIP(hipMemcpyAsync(e_d, e_h, sizeof(double) * nedges, hipMemcpyHostToDevice, s));
  if (counts) {
    // counts and the level maximum sit in ONE device block [top | counts]: one memset, one copy back
    PRAD_TRY(c.get<unsigned long long>("bin_counts", (size_t)nedges + 2, &cnt_d));
    PRAD_HIP(hipMemsetAsync(cnt_d, 0, sizeof(unsigned long long) * ((size_t)nedges + 2), s));
    top = (int *)cnt_d;
    cnt_d += 1;
  } else {
    PRAD_HIP(hipMemsetAsync(top, 0, sizeof(int), s));
  }
  switch (dtype) {
    case 0: PRAD_TRY(launch_digitize((const float *)image, mask, n, e_d, nedges, levels, top, cnt_d, s)); break;
    case 1: PRAD_TRY(launch_digitize((const double *)image, mask, n, e_d, nedges, levels, top, cnt_d, s)); break;
    case 2: PRAD_TRY(launch_digitize((const int *)image, mask, n, e_d, nedges, levels, top, cnt_d, s)); break;
    case 3: PRAD_TRY(launch_digitize((const short *)image, mask, n, e_d, nedges, levels, top, cnt_d, s)); break;
    default: return fail(PRAD_E_ARG, "digitize: dtype %d", dtype);
  }
  if (counts) {
    PRAD_HIP(hipMemcpyAsync(top_h, cnt_d - 1, sizeof(long long) * ((size_t)nedges + 2), hipMemcpyDeviceToHost, s));   // [top | counts]
  } else {
    PRAD_HIP(hipMemcpyAsync(top_h, top, sizeof(int), hipMemcpyDeviceToHost, s));
  }
  PRAD_HIP(hipStreamSynchronize(s));
  if (max_level) *max_level = *top_h;
  if (counts) memcpy(counts, cnt_h, sizeof(long long) * ((size_t)nedges + 1));
  return PRAD_OK;
}

// the queueing half of prad_bincount_dev: slot in [0, PRAD_BIN_SLOTS) names the device block, the pinned block and the event
namespace {
constexpr int PRAD_BIN_SLOTS = 4;
struct BinSlots {
  int device = -1;
  hipEvent_t done[PRAD_BIN_SLOTS] = {};
  int nedges[PRAD_BIN_SLOTS] = {};
  unsigned long long *host[PRAD_BIN_SLOTS] = {};
  bool used[PRAD_BIN_SLOTS] = {};
  unsigned seq = 0;
};
BinSlots &bin_slots() {
  static thread_local BinSlots b;
  return b;
}
int bincount_queue(Context &c, int slot, const void *image, int dtype, const uint8_t *mask, long long n, int binCount, int32_t *levels,
                   hipStream_t s, unsigned long long **host_out) {
  const int nedges = binCount + 1;
  // one device block: [keys (2 u64) | info (8 B) | top (8 B) | counts (nedges + 1) | edges (nedges doubles)] -- one copy back
  unsigned long long *blk = nullptr;
  const size_t words = 2 + 1 + 1 + ((size_t)nedges + 1) + (size_t)nedges;
  char name_blk[32], name_pin[32];
  snprintf(name_blk, sizeof(name_blk), "bincount_blk%d", slot);
  snprintf(name_pin, sizeof(name_pin), "bincount_pin%d", slot);
  PRAD_TRY(c.get<unsigned long long>(name_blk, words, &blk));
  unsigned long long *keys = blk;
  int *info = (int *)(blk + 2), *top = (int *)(blk + 3);
  unsigned long long *cnt_d = blk + 4;
  double *e_d = (double *)(blk + 4 + (size_t)nedges + 1);
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned(name_pin, sizeof(unsigned long long) * (words + 2), &pin));
  unsigned long long *h = (unsigned long long *)pin;
  PRAD_HIP(hipMemsetAsync(blk, 0, sizeof(unsigned long long) * words, s));
  h[words] = ~0ull;
  h[words + 1] = 0ull;
  PRAD_HIP(hipMemcpyAsync(keys, h + words, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
  static const int cap = getenv("PRAD_MINMAX_BLOCKS") ? atoi(getenv("PRAD_MINMAX_BLOCKS")) : 512;
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, cap));
  switch (dtype) {
    case 0: hipLaunchKernelGGL(roi_minmax_kernel<float>, dim3(gx), dim3(256), 0, s, (const float *)image, mask, n, keys); break;
    case 1: hipLaunchKernelGGL(roi_minmax_kernel<double>, dim3(gx), dim3(256), 0, s, (const double *)image, mask, n, keys); break;
    case 2: hipLaunchKernelGGL(roi_minmax_kernel<int>, dim3(gx), dim3(256), 0, s, (const int *)image, mask, n, keys); break;
    default: hipLaunchKernelGGL(roi_minmax_kernel<short>, dim3(gx), dim3(256), 0, s, (const short *)image, mask, n, keys); break;
  }
  if (dtype == 0)
    hipLaunchKernelGGL(bincount_edges_kernel<true>, dim3(1), dim3(256), 0, s, (const unsigned long long *)keys, binCount, e_d, info);
  else
    hipLaunchKernelGGL(bincount_edges_kernel<false>, dim3(1), dim3(256), 0, s, (const unsigned long long *)keys, binCount, e_d, info);
  PRAD_TRY(check_launch("bincount_edges_kernel"));
  switch (dtype) {
    case 0: PRAD_TRY(launch_digitize((const float *)image, mask, n, e_d, nedges, levels, top, cnt_d, s)); break;
    case 1: PRAD_TRY(launch_digitize((const double *)image, mask, n, e_d, nedges, levels, top, cnt_d, s)); break;
    case 2: PRAD_TRY(launch_digitize((const int *)image, mask, n, e_d, nedges, levels, top, cnt_d, s)); break;
    default: PRAD_TRY(launch_digitize((const short *)image, mask, n, e_d, nedges, levels, top, cnt_d, s)); break;
  }
  PRAD_HIP(hipMemcpyAsync(h, blk, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost, s));
  *host_out = h;
  return PRAD_OK;
}
int bincount_collect(const unsigned long long *h, int nedges, double *minmax, double *edges, int *max_level, long long *counts) {
  if (h[1] == 0ull) return fail(PRAD_E_ARG, "roi_minmax: empty ROI");
  minmax[0] = f64_unkey(h[0]);
  minmax[1] = f64_unkey(h[1]);
  if (*(const int *)(h + 2))
    return fail(PRAD_E_UNSUPPORTED, "bincount: constant or non-finite ROI (np.histogram widens the range: host-built edges)");
  if (max_level) *max_level = *(const int *)(h + 3);
  if (counts) memcpy(counts, h + 4, sizeof(long long) * ((size_t)nedges + 1));
  if (edges) memcpy(edges, h + 4 + (size_t)nedges + 1, sizeof(double) * nedges);
  return PRAD_OK;
}
}  // namespace

int prad_bincount_dev(const void *image, int dtype, const uint8_t *mask, long long n, int binCount, int32_t *levels,
                      double *minmax, double *edges, int *max_level, long long *counts, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!image || !mask || !levels || !minmax || n < 1) return fail(PRAD_E_ARG, "bincount: bad arguments");
  if (dtype < 0 || dtype > 3) return fail(PRAD_E_ARG, "bincount: dtype %d", dtype);
  if (binCount < 1 || binCount > 4096) return fail(PRAD_E_UNSUPPORTED, "bincount: binCount %d outside [1, 4096]", binCount);
  hipStream_t s = (hipStream_t)stream;
  unsigned long long *h = nullptr;
  PRAD_TRY(bincount_queue(c, PRAD_BIN_SLOTS, image, dtype, mask, n, binCount, levels, s, &h));   // (a block of its own: no ticket)
  PRAD_HIP(hipStreamSynchronize(s));
  return bincount_collect(h, binCount + 1, minmax, edges, max_level, counts);
}

// prad_bincount_dev in two halves (round 6: the case pipeline bins image i + 1 while the host still works on image i): the
// queueing half returns a ticket, prad_bincount_wait(ticket, ...) waits for THAT work only (an event behind its copy) and hands
// over what prad_bincount_dev returns.  Up to four tickets per thread; a ticket must be waited for exactly once.
int prad_bincount_enqueue_dev(const void *image, int dtype, const uint8_t *mask, long long n, int binCount, int32_t *levels,
                              int *ticket, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!image || !mask || !levels || !ticket || n < 1) return fail(PRAD_E_ARG, "bincount_enqueue: bad arguments");
  if (dtype < 0 || dtype > 3) return fail(PRAD_E_ARG, "bincount_enqueue: dtype %d", dtype);
  if (binCount < 1 || binCount > 4096) return fail(PRAD_E_UNSUPPORTED, "bincount_enqueue: binCount %d outside [1, 4096]", binCount);
  BinSlots &b = bin_slots();
  if (b.device != c.device) {
    for (int t = 0; t < PRAD_BIN_SLOTS; t++) {
      if (b.done[t]) (void)hipEventDestroy(b.done[t]);
      PRAD_HIP(hipEventCreateWithFlags(&b.done[t], hipEventDisableTiming));
      b.used[t] = false;
    }
    b.device = c.device;
  }
  const int t = (int)(b.seq % PRAD_BIN_SLOTS);
  if (b.used[t]) return fail(PRAD_E_ARG, "bincount_enqueue: %d binnings are in flight on this thread; prad_bincount_wait one first", PRAD_BIN_SLOTS);
  hipStream_t s = (hipStream_t)stream;
  PRAD_TRY(bincount_queue(c, t, image, dtype, mask, n, binCount, levels, s, &b.host[t]));
  PRAD_HIP(hipEventRecord(b.done[t], s));
  b.nedges[t] = binCount + 1;
  b.used[t] = true;
  b.seq++;
  *ticket = t;
  return PRAD_OK;
}

int prad_bincount_wait(int ticket, double *minmax, double *edges, int *max_level, long long *counts) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  BinSlots &b = bin_slots();
  if (ticket < 0 || ticket >= PRAD_BIN_SLOTS || !b.used[ticket] || !minmax) return fail(PRAD_E_ARG, "bincount_wait: ticket %d", ticket);
  b.used[ticket] = false;
  PRAD_HIP(hipEventSynchronize(b.done[ticket]));
  return bincount_collect(b.host[ticket], b.nedges[ticket], minmax, edges, max_level, counts);
}

int prad_digitize_dev(const void *image, int dtype, const uint8_t *mask, long long n, const double *edges, int nedges,
                      int32_t *levels, int *max_level, void *stream) {
  return prad_digitize_counts_dev(image, dtype, mask, n, edges, nedges, levels, max_level, nullptr, stream);
}

int prad_level_counts_dev(const int32_t *levels, const uint8_t *mask, long long n, int Ng, long long *counts,
                          void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!levels || !mask || !counts || n < 1 || Ng < 1) return fail(PRAD_E_ARG, "level_counts: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long *d = nullptr;
  const size_t nb = (size_t)Ng + 1;
  PRAD_TRY(c.get<unsigned long long>("level_counts", nb, &d));
  PRAD_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long) * nb, s));
  // per-block partial sums are 32-bit: a block sees at most n / grid voxels, bounded below 2^32 by the grid size
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 1024));   // (flush atomics: see launch_digitize)
  const int use_lds = 4 * nb * sizeof(unsigned) <= 48 * 1024;
  hipLaunchKernelGGL(level_counts_kernel, dim3(gx), dim3(256), use_lds ? 4 * nb * sizeof(unsigned) : 0, s, levels, mask,
                     n, Ng, use_lds, d);
  PRAD_TRY(check_launch("level_counts_kernel"));
  PRAD_HIP(hipMemcpyAsync(counts, d, sizeof(long long) * nb, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  return PRAD_OK;
}

int prad_glszm_sizes(int *sizes, int capacity) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  return glszm_distinct_sizes(c, sizes, capacity);
}
int prad_fill_glszm_compact_dev(double *glszm, int Ng, int nsizes, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!glszm || Ng < 1 || nsizes < 0) return fail(PRAD_E_ARG, "bad fill_glszm_compact arguments");
  return glszm_fill_compact(c, (hipStream_t)stream, glszm, Ng, nsizes);
}

long long prad_glszm_zones(int v, int *tempData, long long capacity_pairs) {
  Context &c = ctx();
  int rc = c.ensure_device();
  if (rc != PRAD_OK) return rc;
  return glszm_copy_zones(c, v, tempData, capacity_pairs);
}


}  // extern "C"

#include "prad_image.h"   // the case pipeline's native half: image queues, prad_image_enqueue_dev, the launcher thread
