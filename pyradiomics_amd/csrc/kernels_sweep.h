// kernels_sweep.h -- the segment-mode hot path for GLCM + GLRLM on gfx950.
//
// Data layout in HBM
//   image  int32 [Nz][Ny][Nx]  + mask uint8 [Nz][Ny][Nx]     (boundary dtypes, 5 B/voxel, read ONCE)
//   levels uint8 [Nz][Ny][Nx]   level | 0 = outside the ROI     (1 B/voxel, written once by pack_levels,
//                                                              then L2/Infinity-Cache resident for the sweeps)
//   acc    uint32 GLCM [Na][Ng][Ng], GLRLM [Na][Ng][Nr]        (angle-major while accumulating)
//   out    float64 GLCM [Ng][Ng][Na], GLRLM [Ng][Nr][Na]       (reference layout, written by finalize)
//
// Why sweeps.  For a unit angle d the ordered neighbour pairs (p, p+d) of the GLCM are exactly the
// consecutive voxels of the lines the GLRLM walks (cmatrices.c:61-85 vs :448-510), so one walk along every
// line of an angle yields both matrices for that angle: a lane keeps (previous level, current run length)
// in registers, reads ONE byte per step and issues at most one GLCM and one GLRLM increment.
// Lanes of a wavefront are 64 x-adjacent lines, so every step of the walk is a single coalesced 64-byte
// read even for the diagonal angles (the whole wave shifts by one voxel per step).  Lines are indexed by
// their (virtual) position at march coordinate 0, which makes the set of lines of a skewed angle a plain
// rectangle; lanes are simply inactive before their line enters / after it leaves the volume.
// The x-axis angle (0,0,1) marches along the lane dimension itself; there a wave walks one row 64 voxels at
// a time and derives run boundaries from wave ballots (no per-lane state at all).
//
// Histograms are privatised per workgroup in LDS (ds_add_u32, no return), GLRLM only for run lengths
// <= RS (longer runs are rare and go to L2 atomics), and merged with one global atomic per non-zero bin.
// This is integer histogramming: MFMA has no role here; the bound is LDS-atomic issue + HBM.
#pragma once
#include "prad_runtime.h"

namespace prad {

typedef unsigned int u32;
typedef unsigned long long u64;

// flags[0]: set when a masked voxel has a level outside [1, Ng] (=> the exact generic path must run)
__global__ void __launch_bounds__(256) pack_levels_kernel(const int *__restrict__ image,
                                                          const uint8_t *__restrict__ mask, long long n, int Ng,
                                                          uint8_t *__restrict__ levels, int *__restrict__ flags,
                                                          int vec_ok) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  int bad = 0;
  long long done = 0;
  if (vec_ok) {
    const long long n16 = n >> 4;
    const int4 *im4 = reinterpret_cast<const int4 *>(image);
    const uint4 *mk4 = reinterpret_cast<const uint4 *>(mask);
    uint4 *out4 = reinterpret_cast<uint4 *>(levels);
    for (long long t = tid; t < n16; t += nthreads) {
      uint4 m = mk4[t];
      int4 q0 = im4[4 * t], q1 = im4[4 * t + 1], q2 = im4[4 * t + 2], q3 = im4[4 * t + 3];
      const int lv[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                          q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
      const u32 mw[4] = {m.x, m.y, m.z, m.w};
      u32 ow[4];
#pragma unroll
      for (int w = 0; w < 4; w++) {
        u32 o = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const bool in = (mw[w] >> (8 * b)) & 0xffu;
          const int l = lv[w * 4 + b];
          bad |= in && (l < 1 || l > Ng);
          o |= (in ? ((u32)l & 0xffu) : 0u) << (8 * b);
        }
        ow[w] = o;
      }
      out4[t] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
    done = n16 << 4;
  }
  for (long long i = done + tid; i < n; i += nthreads) {
    const bool in = mask[i] != 0;
    const int l = image[i];
    bad |= in && (l < 1 || l > Ng);
    levels[i] = in ? (uint8_t)l : (uint8_t)0;
  }
  if (bad) flags[0] = 1;
}

// One sweepable angle, expressed in (march, row, lane) coordinates.
struct SweepDesc {
  int slot;          // index of the angle in the caller's list (output column)
  int NM, NU, NX;    // extents: march dim, row dim, lane dim (lane dim is always the contiguous x axis)
  int du, dx;        // motion per march step in the row / lane dims (-1, 0, +1)
  long long sM, sU;  // element strides of march and row dims
  int LU, LXc;       // virtual rows, 64-lane chunks per virtual row
  int u0min, x0min;  // first virtual row / lane coordinate
  long long chunks;  // LU * LXc
};
#define PRAD_MAX_SWEEP 16
struct SweepSet {
  int count;
  SweepDesc d[PRAD_MAX_SWEEP];
};

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// LDS layout: [DO_GLCM ? Ng*Ng : 0] GLCM bins, then [DO_GLRLM ? Ng*RS : 0] short-run bins
template <bool DO_GLCM, bool DO_GLRLM>
__device__ __forceinline__ void flush_block_hist(const u32 *lds, int Ng, int Nr, int RS, int slot,
                                                 u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc) {
  __syncthreads();
  if (DO_GLCM) {
    u32 *dst = glcm_acc + (size_t)slot * Ng * Ng;
    for (int i = threadIdx.x; i < Ng * Ng; i += blockDim.x) {
      u32 v = lds[i];
      if (v) atomicAdd(dst + i, v);
    }
  }
  if (DO_GLRLM) {
    const u32 *hr = lds + (DO_GLCM ? Ng * Ng : 0);
    u32 *dst = glrlm_acc + (size_t)slot * Ng * Nr;
    for (int i = threadIdx.x; i < Ng * RS; i += blockDim.x) {
      u32 v = hr[i];
      if (v) atomicAdd(dst + (size_t)(i / RS) * Nr + (i % RS), v);
    }
  }
}

#define PRAD_SWEEP_UNROLL 8

// Angles whose march dimension is NOT the contiguous axis: one lane per line.
template <bool DO_GLCM, bool DO_GLRLM>
__global__ void __launch_bounds__(256) sweep_lines_kernel(SweepSet set, const uint8_t *__restrict__ L, int Ng,
                                                          int Nr, int RS, u32 *__restrict__ glcm_acc,
                                                          u32 *__restrict__ glrlm_acc, int *__restrict__ multi,
                                                          const int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;  // irregular levels: the generic path will redo this call
  const int nbins = (DO_GLCM ? Ng * Ng : 0) + (DO_GLRLM ? Ng * RS : 0);
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  u32 *hc = lds;
  u32 *hr = lds + (DO_GLCM ? Ng * Ng : 0);

  const SweepDesc &D = set.d[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const long long step = D.sM + (long long)D.du * D.sU + D.dx;  // address increment per march step
  u32 *rl_long = glrlm_acc + (size_t)D.slot * Ng * Nr;
  int seen_multi = 0;

  for (long long chunk = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); chunk < D.chunks;
       chunk += nwaves) {
    const int ui = (int)(chunk / D.LXc);
    const int xc = (int)(chunk - (long long)ui * D.LXc);
    const int u0 = D.u0min + ui;
    const int x0 = D.x0min + xc * 64 + lane;
    // march interval during which this lane's line is inside the volume
    int lo = 0, hi = D.NM - 1;
    if (D.du > 0) { lo = max(lo, -u0); hi = min(hi, D.NU - 1 - u0); }
    else if (D.du < 0) { lo = max(lo, u0 - (D.NU - 1)); hi = min(hi, u0); }
    if (D.dx > 0) { lo = max(lo, -x0); hi = min(hi, D.NX - 1 - x0); }
    else if (D.dx < 0) { lo = max(lo, x0 - (D.NX - 1)); hi = min(hi, x0); }
    else if (x0 < 0 || x0 >= D.NX) { lo = 1; hi = 0; }
    const bool live = lo <= hi;
    const int wlo = wave_min_i32(live ? lo : 0x7fffffff);
    const int whi = wave_max_i32(live ? hi : -1);
    if (wlo > whi) continue;
    const uint8_t *p0 = L + (long long)u0 * D.sU + x0;  // address of the line at march coordinate 0
    int prev = 0, rl = 0, nmask = 0;
    for (int t = wlo; t <= whi; t += PRAD_SWEEP_UNROLL) {
      int v[PRAD_SWEEP_UNROLL];
#pragma unroll
      for (int k = 0; k < PRAD_SWEEP_UNROLL; k++) {
        const int tt = t + k;
        v[k] = (tt >= lo && tt <= hi) ? (int)p0[(long long)tt * step] : 0;
      }
#pragma unroll
      for (int k = 0; k < PRAD_SWEEP_UNROLL; k++) {
        const int cur = v[k];
        if (DO_GLCM) {
          if (prev && cur) atomicAdd(&hc[(prev - 1) * Ng + (cur - 1)], 1u);
        }
        if (DO_GLRLM) {
          if (cur != prev) {
            if (prev) {
              if (rl <= RS) atomicAdd(&hr[(prev - 1) * RS + (rl - 1)], 1u);
              else atomicAdd(&rl_long[(size_t)(prev - 1) * Nr + (rl - 1)], 1u);
            }
            rl = 0;
          }
          rl++;
          nmask += (cur != 0);
        }
        prev = cur;
      }
    }
    if (DO_GLRLM) {
      if (prev) {
        if (rl <= RS) atomicAdd(&hr[(prev - 1) * RS + (rl - 1)], 1u);
        else atomicAdd(&rl_long[(size_t)(prev - 1) * Nr + (rl - 1)], 1u);
      }
      seen_multi |= (nmask > 1);
    }
  }
  if (DO_GLRLM && seen_multi) multi[D.slot] = 1;
  flush_block_hist<DO_GLCM, DO_GLRLM>(lds, Ng, Nr, RS, D.slot, glcm_acc, glrlm_acc);
}

// The angle along the contiguous axis: a wave walks one row, 64 voxels per step.
template <bool DO_GLCM, bool DO_GLRLM>
__global__ void __launch_bounds__(256) sweep_rows_kernel(const uint8_t *__restrict__ L, long long nrows, int NX,
                                                         int slot, int Ng, int Nr, int RS,
                                                         u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc,
                                                         int *__restrict__ multi, const int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;
  const int nbins = (DO_GLCM ? Ng * Ng : 0) + (DO_GLRLM ? Ng * RS : 0);
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  u32 *hc = lds;
  u32 *hr = lds + (DO_GLCM ? Ng * Ng : 0);
  const int lane = threadIdx.x & 63;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  u32 *rl_long = glrlm_acc + (size_t)slot * Ng * Nr;
  int seen_multi = 0;

  for (long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < nrows;
       row += nwaves) {
    const uint8_t *p = L + row * NX;
    int nxt_chunk = lane < NX ? (int)p[lane] : 0;
    int carry = 0;      // length of the run that is open at the start of this chunk (0 = none)
    int last_prev = 0;  // level of the voxel just before this chunk
    int nmask = 0;
    for (int xc = 0; xc < NX; xc += 64) {
      const int cur = nxt_chunk;
      const int xn = xc + 64 + lane;
      nxt_chunk = xn < NX ? (int)p[xn] : 0;
      int nxt = __shfl_down(cur, 1);
      const int first_of_next = __shfl(nxt_chunk, 0);
      if (lane == 63) nxt = first_of_next;
      if (DO_GLCM) {
        if (cur && nxt) atomicAdd(&hc[(cur - 1) * Ng + (nxt - 1)], 1u);
      }
      if (DO_GLRLM) {
        int prv = __shfl_up(cur, 1);
        if (lane == 0) prv = last_prev;
        const bool is_start = cur && prv != cur;
        const bool is_end = cur && nxt != cur;
        const u64 S = __ballot(is_start);
        const u64 E = __ballot(is_end);
        if (is_end) {
          const u64 below = S & (~0ull >> (63 - lane));
          const int len = below ? lane - (63 - __clzll((long long)below)) + 1 : lane + 1 + carry;
          if (len <= RS) atomicAdd(&hr[(cur - 1) * RS + (len - 1)], 1u);
          else atomicAdd(&rl_long[(size_t)(cur - 1) * Nr + (len - 1)], 1u);
        }
        const int cur63 = __shfl(cur, 63);
        if (cur63 != 0 && !(E >> 63)) carry = S ? __clzll((long long)S) + 1 : carry + 64;
        else carry = 0;
        last_prev = cur63;
        nmask += __popcll(__ballot(cur != 0));
      }
    }
    if (DO_GLRLM) seen_multi |= (nmask > 1);
  }
  if (DO_GLRLM && seen_multi) multi[slot] = 1;
  flush_block_hist<DO_GLCM, DO_GLRLM>(lds, Ng, Nr, RS, slot, glcm_acc, glrlm_acc);
}

// acc (angle-major u32) -> reference layout float64
__global__ void finalize_glcm_kernel(const u32 *__restrict__ acc, int Ng, int Na, double *__restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Ng * Ng * Na;
  if (idx >= total) return;
  const int a = (int)(idx % Na);
  const long long ij = idx / Na;
  out[idx] = (double)acc[(size_t)a * Ng * Ng + ij];
}

__global__ void finalize_glrlm_kernel(const u32 *__restrict__ acc, const int *__restrict__ multi, int Ng, int Nr,
                                      int Na, double *__restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Ng * Nr * Na;
  if (idx >= total) return;
  const int a = (int)(idx % Na);
  const long long gr = idx / Na;
  const int r = (int)(gr % Nr);
  // cmatrices.c:524-534: an angle without any line of >= 2 masked voxels loses its run-length-1 column
  out[idx] = (r == 0 && !multi[a]) ? 0.0 : (double)acc[(size_t)a * Ng * Nr + gr];
}

}  // namespace prad
