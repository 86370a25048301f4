// kernels_generic.h -- exact, shape-agnostic device kernels.
//
// One lane per (kernel v, box voxel k[, angle a]).  They work for any Nd <= PRAD_MAX_ND, any angle set,
// any level range (incl. the reference's flat-index aliasing / "index out of range" behaviour,
// cmatrices.c:79-84, :460-466, :742-747, :645-650) and any per-kernel bounding box, so they are
// (1) the voxel-based path (thousands of small boxes, outputs written straight into the
//     [Nvox][...] float64 result with fp64 atomics -- exact, the addends are integers), and
// (2) the safety net behind the tiled/sweep kernels for inputs those do not cover.
// Counts are accumulated with global_atomic_add_f64 on pre-zeroed outputs.
#pragma once
#include "prad_runtime.h"

namespace prad {

__device__ __forceinline__ void kernel_box(const Geo &g, const VoxMode &vm, int v, int *lo, int *hi) {
  for (int d = 0; d < g.nd; d++) {
    if (!vm.voxels) {
      lo[d] = 0;
      hi[d] = g.size[d] - 1;
    } else {
      int c = vm.voxels[(long long)d * vm.nvox + v];
      if (d == vm.f2d) {
        lo[d] = hi[d] = c;  // _cmatrices.c:1127-1128
      } else {
        lo[d] = max(c - vm.radius, 0);
        hi[d] = min(c + vm.radius, g.size[d] - 1);
      }
    }
  }
}

// decode box-local raster index k into coordinates; false if k is beyond the box
__device__ __forceinline__ bool box_decode(const Geo &g, const int *lo, const int *hi, long long k, int *c,
                                           long long *offset) {
  long long off = 0;
  for (int d = g.nd - 1; d >= 0; d--) {
    int ext = hi[d] - lo[d] + 1;
    if (ext <= 0) return false;
    int q = (int)(k % ext);
    k /= ext;
    c[d] = lo[d] + q;
    off += (long long)c[d] * g.stride[d];
  }
  *offset = off;
  return k == 0;
}

__device__ __forceinline__ long long box_step(const Geo &g, const int *lo, const int *hi, const int *c,
                                              const int *ang, int sign) {
  long long off = 0;
  for (int d = 0; d < g.nd; d++) {
    int q = c[d] + sign * ang[d];
    if (q < lo[d] || q > hi[d]) return -1;
    off += (long long)q * g.stride[d];
  }
  return off;
}

// ---- GLCM (cmatrices.c:4-92) -------------------------------------------------------------------
__global__ void __launch_bounds__(256) gen_glcm_kernel(Geo g, VoxMode vm, const int *__restrict__ image,
                                                       const uint8_t *__restrict__ mask,
                                                       const int *__restrict__ angles, int Na, int Ng,
                                                       double *__restrict__ out, int *__restrict__ err) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int v = (int)(gid / vm.boxmax);
  long long k = gid % vm.boxmax;
  if (v >= vm.nvox) return;
  int lo[PRAD_MAX_ND], hi[PRAD_MAX_ND], c[PRAD_MAX_ND];
  kernel_box(g, vm, v, lo, hi);
  long long i;
  if (!box_decode(g, lo, hi, k, c, &i)) return;
  if (!mask[i]) return;
  const unsigned long long idx_max = (unsigned long long)Ng * Ng * Na;
  const int li = image[i];
  double *o = out + (size_t)v * idx_max;
  for (int a = 0; a < Na; a++) {
    long long j = box_step(g, lo, hi, c, angles + a * g.nd, +1);
    if (j < 0 || !mask[j]) continue;
    const int lj = image[j];
    unsigned long long idx = (unsigned long long)a + (unsigned long long)((long long)(lj - 1) * Na) +
                             (unsigned long long)((long long)(li - 1) * Na * Ng);
    if (li <= 0 || lj <= 0 || idx >= idx_max) {
      *err = 1;
      return;
    }
    atomicAdd(o + idx, 1.0);
  }
}

// ---- GLDM (cmatrices.c:660-754) ----------------------------------------------------------------
__global__ void __launch_bounds__(256) gen_gldm_kernel(Geo g, VoxMode vm, const int *__restrict__ image,
                                                       const uint8_t *__restrict__ mask,
                                                       const int *__restrict__ angles, int Na, int Ng, int alpha,
                                                       double *__restrict__ out, int *__restrict__ err) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int v = (int)(gid / vm.boxmax);
  long long k = gid % vm.boxmax;
  if (v >= vm.nvox) return;
  int lo[PRAD_MAX_ND], hi[PRAD_MAX_ND], c[PRAD_MAX_ND];
  kernel_box(g, vm, v, lo, hi);
  long long i;
  if (!box_decode(g, lo, hi, k, c, &i)) return;
  if (!mask[i]) return;
  const unsigned long long width = (unsigned long long)Na * 2 + 1, idx_max = (unsigned long long)Ng * width;
  const int li = image[i];
  int dep = 0;
  for (int a = 0; a < Na; a++) {
    long long j = box_step(g, lo, hi, c, angles + a * g.nd, +1);
    if (j < 0 || !mask[j]) continue;
    int diff = li - image[j];
    if (diff < 0) diff = -diff;
    if (diff <= alpha) dep++;
  }
  unsigned long long idx = (unsigned long long)dep + (unsigned long long)((long long)(li - 1) * (long long)width);
  if (li <= 0 || idx >= idx_max) {
    *err = 1;
    return;
  }
  atomicAdd(out + (size_t)v * idx_max + idx, 1.0);
}

// ---- GLRLM (cmatrices.c:299-541) ---------------------------------------------------------------
// one lane per (kernel, angle, box voxel); lanes whose predecessor is inside the box retire at once,
// the rest walk their line.  multi[v*Na+a] is set when a line holds >= 2 masked voxels.
__global__ void __launch_bounds__(256) gen_glrlm_kernel(Geo g, VoxMode vm, const int *__restrict__ image,
                                                        const uint8_t *__restrict__ mask,
                                                        const int *__restrict__ angles, int Na, int Ng, int Nr,
                                                        double *__restrict__ out, int *__restrict__ multi,
                                                        int *__restrict__ err) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long per_v = vm.boxmax * Na;
  int v = (int)(gid / per_v);
  if (v >= vm.nvox) return;
  long long r = gid % per_v;
  int a = (int)(r / vm.boxmax);
  long long k = r % vm.boxmax;
  int lo[PRAD_MAX_ND], hi[PRAD_MAX_ND], c[PRAD_MAX_ND];
  kernel_box(g, vm, v, lo, hi);
  long long j;
  if (!box_decode(g, lo, hi, k, c, &j)) return;
  const int *ang = angles + a * g.nd;
  if (box_step(g, lo, hi, c, ang, -1) >= 0) return;  // not the first voxel of its line
  const unsigned long long idx_max = (unsigned long long)Ng * Nr * Na;
  double *o = out + (size_t)v * idx_max;
  int gl = -1, rl = 0, elements = 0;
  bool bad = false;
  auto emit = [&](int level, int len) {
    unsigned long long idx = (unsigned long long)a + (unsigned long long)len * Na +
                             (unsigned long long)((long long)(level - 1) * Na * Nr);
    if (level <= 0 || idx >= idx_max) bad = true;
    else atomicAdd(o + idx, 1.0);
  };
  while (j >= 0 && !bad) {
    if (mask[j]) {
      elements++;
      int lv = image[j];
      if (gl == -1) gl = lv;
      else if (lv == gl) rl++;
      else {
        emit(gl, rl);
        gl = lv;
        rl = 0;
      }
    } else if (gl > -1) {
      emit(gl, rl);
      gl = -1;
      rl = 0;
    }
    j = box_step(g, lo, hi, c, ang, +1);
    if (j >= 0)
      for (int d = 0; d < g.nd; d++) c[d] += ang[d];
  }
  if (!bad && gl > -1) emit(gl, rl);
  if (bad) *err = 1;
  if (elements > 1) multi[v * Na + a] = 1;
}

// clears the run-length-1 column of angles that never saw a line with >= 2 masked voxels
// (cmatrices.c:524-534)
__global__ void glrlm_prune_kernel(double *__restrict__ out, const int *__restrict__ multi, int nvox, int Ng,
                                   int Nr, int Na) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)nvox * Ng * Na;
  if (gid >= total) return;
  int a = (int)(gid % Na);
  long long r = gid / Na;
  int gl = (int)(r % Ng);
  int v = (int)(r / Ng);
  if (!multi[v * Na + a]) out[((size_t)v * Ng + gl) * Nr * Na + a] = 0.0;
}

// ---- NGTDM (cmatrices.c:543-658) ----------------------------------------------------------------
// voxel mode: one lane per kernel, raster order inside the box => bit-identical float64 sums.
__global__ void __launch_bounds__(64) gen_ngtdm_voxel_kernel(Geo g, VoxMode vm, const int *__restrict__ image,
                                                             const uint8_t *__restrict__ mask,
                                                             const int *__restrict__ angles, int Na, int Ng,
                                                             double *__restrict__ out, int *__restrict__ err) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= vm.nvox) return;
  int lo[PRAD_MAX_ND], hi[PRAD_MAX_ND], c[PRAD_MAX_ND];
  kernel_box(g, vm, v, lo, hi);
  double *o = out + (size_t)v * Ng * 3;
  for (int gl = 0; gl < Ng; gl++) {
    o[gl * 3] = 0.0;
    o[gl * 3 + 1] = 0.0;
    o[gl * 3 + 2] = gl + 1;
  }
  long long total = 1;
  for (int d = 0; d < g.nd; d++) total *= (hi[d] - lo[d] + 1);
  for (long long k = 0; k < total; k++) {
    long long i;
    box_decode(g, lo, hi, k, c, &i);
    if (!mask[i]) continue;
    double count = 0, sum = 0;
    for (int a = 0; a < Na; a++) {
      long long j = box_step(g, lo, hi, c, angles + a * g.nd, +1);
      if (j < 0 || !mask[j]) continue;
      count += 1;
      sum += image[j];
    }
    int li = image[i];
    double diff = (count == 0) ? 0.0 : (double)li - sum / count;
    if (diff < 0) diff = -diff;
    if (li <= 0 || li > Ng) {
      *err = 1;
      return;
    }
    o[(li - 1) * 3] += 1.0;
    o[(li - 1) * 3 + 1] += diff;
  }
}

// segment mode: one lane per voxel, exact integer partial sums S[level][c] = sum |c*level - s|
// (acc layout: [Ng][Na+1] u64, slot 0 = voxel count of the level).
__global__ void __launch_bounds__(256) gen_ngtdm_segment_kernel(Geo g, VoxMode vm, const int *__restrict__ image,
                                                                const uint8_t *__restrict__ mask,
                                                                const int *__restrict__ angles, int Na, int Ng,
                                                                unsigned long long *__restrict__ acc,
                                                                int *__restrict__ err) {
  long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int lo[PRAD_MAX_ND], hi[PRAD_MAX_ND], c[PRAD_MAX_ND];
  kernel_box(g, vm, 0, lo, hi);
  long long i;
  if (!box_decode(g, lo, hi, k, c, &i)) return;
  if (!mask[i]) return;
  long long cnt = 0, sum = 0;
  for (int a = 0; a < Na; a++) {
    long long j = box_step(g, lo, hi, c, angles + a * g.nd, +1);
    if (j < 0 || !mask[j]) continue;
    cnt++;
    sum += image[j];
  }
  int li = image[i];
  if (li <= 0 || li > Ng) {
    *err = 1;
    return;
  }
  unsigned long long *row = acc + (size_t)(li - 1) * (Na + 1);
  atomicAdd(row, 1ull);
  if (cnt > 0) {
    long long d = cnt * (long long)li - sum;
    if (d < 0) d = -d;
    if (d) atomicAdd(row + cnt, (unsigned long long)d);
  }
}

__global__ void ngtdm_finalize_kernel(const unsigned long long *__restrict__ acc, int Ng, int Na,
                                      double *__restrict__ out) {
  int gl = blockIdx.x * blockDim.x + threadIdx.x;
  if (gl >= Ng) return;
  const unsigned long long *row = acc + (size_t)gl * (Na + 1);
  double s = 0.0;
  for (int c = 1; c <= Na; c++) s += (double)row[c] / (double)c;
  out[gl * 3] = (double)row[0];
  out[gl * 3 + 1] = s;
  out[gl * 3 + 2] = gl + 1;
}

}  // namespace prad
