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
// in registers, consumes ONE byte per step and issues one GLCM and one GLRLM LDS increment.
//   * lines kernel (angles marching along z or y): lanes of a wavefront are 64 x-adjacent lines, so every
//     step is one coalesced 64-byte read from a wave-uniform (SGPR) base + lane offset, even for diagonal
//     angles (the whole wave shifts by one voxel per step).  Lines are indexed by their virtual position at
//     march coordinate 0, which turns the line set of a skewed angle into a rectangle; the steps during
//     which every lane of the wave is inside the volume run a predicate-free body.
//   * rows kernel (the angle along x): a wave owns 64 consecutive rows; 64x64-voxel tiles are read coalesced,
//     transposed through LDS, and every lane then walks its own row with the same per-step logic.
//
// Per step and lane the state machine is branch-free: invalid increments are steered to a per-lane dummy
// LDS word instead of being branched around (about a dozen VALU ops per step instead of ~90 with branches).
// Histograms are privatised per workgroup in LDS (ds_add_u32, no return).  The GLRLM table is laid out
// [run length][level] so that the 32 levels of one run length -- the common case, short runs -- fall into 32
// different banks.  When Ng*(Ng+Nr) words fit, the whole GLRLM lives in LDS; otherwise only run lengths
// <= RS do and longer runs (rare) go to L2 atomics.  Workgroups are persistent and merge with one global
// atomic per non-zero bin.  This is integer histogramming: MFMA has no role; the bound is LDS-atomic issue.
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

// ---- LDS histogram layout (u32 words) ----------------------------------------------------------------
//   [0, Ng*Ng)                 GLCM  [prev-1][cur-1]                  (only when DO_GLCM)
//   [.., +RS*Ng)               GLRLM [len-1][level-1], len <= RS      (only when DO_GLRLM)
//   [.., +64)                  per-lane dummy words (targets of masked-out increments)
struct HistLayout {
  int Ng, RS;
  int glrlm0;  // first GLRLM word
  int dummy0;  // first dummy word
  int words;   // total
};
__host__ __device__ inline HistLayout hist_layout(bool glcm, bool glrlm, int Ng, int RS) {
  HistLayout h;
  h.Ng = Ng;
  h.RS = RS;
  h.glrlm0 = glcm ? Ng * Ng : 0;
  h.dummy0 = h.glrlm0 + (glrlm ? RS * Ng : 0);
  h.words = h.dummy0 + 64;
  return h;
}

// Per-lane walk state + the branch-free step.  All LDS positions are kept as BYTE offsets so that a step
// needs no index->address shift:  GLCM byte = prev*4Ng + cur*4 + cG,  GLRLM byte = len*4Ng + prev*4 + cR.
typedef __attribute__((address_space(3))) u32 lds_u32;

template <bool DO_GLCM, bool DO_GLRLM, bool LONG>
struct Walker {
  u32 *rl_long;  // global GLRLM rows of this angle (long runs only)
  int Ng4, Nr;   // Ng4 = 4*Ng
  int cG;        // see above (both include the absolute LDS address of the histogram block, so that a
  int cR;        //            bump is a bare ds_add_u32 on a computed 32-bit LDS address)
  int rl_limit;  // len*4Ng + cR beyond which the run is "long" (LONG only)
  int dummy;     // byte offset of this lane's dummy word
  // state
  int prev;   // level of the previous voxel on the line (0 = none / unmasked)
  int prowG;  // prev*4Ng + cG
  int rlN;    // len*4Ng + cR of the stretch of equal values ending at prev
  int nmask;  // masked voxels seen on this line

  __device__ __forceinline__ void init(u32 *lds_, const HistLayout &h, int Nr_, u32 *rl_long_, int lane) {
    const int base = (int)(unsigned)(size_t)((lds_u32 *)lds_);  // 32-bit LDS address
    rl_long = rl_long_;
    Ng4 = 4 * h.Ng;
    Nr = Nr_;
    cG = base - 4 * (h.Ng + 1);
    cR = base + 4 * (h.glrlm0 - (h.Ng + 1));
    rl_limit = h.RS * Ng4 + cR;
    dummy = base + 4 * (h.dummy0 + lane);
  }
  __device__ __forceinline__ void begin_line() {
    prev = 0;
    prowG = cG;
    rlN = Ng4 + cR;
    nmask = 0;
  }
  __device__ __forceinline__ void bump(int lds_addr) {
    __hip_atomic_fetch_add((lds_u32 *)(size_t)(unsigned)lds_addr, 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void step(int cur) {
    const bool pnz = prev != 0, cnz = cur != 0;
    if (DO_GLCM) {
      bump((pnz && cnz) ? prowG + (cur << 2) : dummy);
      prowG = __mul24(cur, Ng4) + cG;
    }
    if (DO_GLRLM) {
      const bool chg = cur != prev;
      const bool emit = chg && pnz;
      if (LONG) {
        const bool lng = rlN > rl_limit;
        bump((emit && !lng) ? rlN + (prev << 2) : dummy);
        if (emit && lng) atomicAdd(&rl_long[(size_t)(prev - 1) * Nr + ((rlN - cR) / Ng4 - 1)], 1u);
      } else {
        bump(emit ? rlN + (prev << 2) : dummy);
      }
      rlN = chg ? Ng4 + cR : rlN + Ng4;
      nmask += cnz;
    }
    prev = cur;
  }
  // closes the run that is open at the end of a line; returns "line held >= 2 masked voxels"
  __device__ __forceinline__ bool end_line() {
    step(0);
    return nmask > 1;
  }
};

template <bool DO_GLCM, bool DO_GLRLM>
__device__ __forceinline__ void flush_block_hist(const u32 *lds, const HistLayout &h, int Nr, int slot,
                                                 u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc) {
  __syncthreads();
  if (DO_GLCM) {
    u32 *dst = glcm_acc + (size_t)slot * h.Ng * h.Ng;
    for (int i = threadIdx.x; i < h.Ng * h.Ng; i += blockDim.x) {
      const u32 v = lds[i];
      if (v) atomicAdd(dst + i, v);
    }
  }
  if (DO_GLRLM) {
    const u32 *hr = lds + h.glrlm0;
    u32 *dst = glrlm_acc + (size_t)slot * h.Ng * Nr;
    for (int i = threadIdx.x; i < h.RS * h.Ng; i += blockDim.x) {
      const u32 v = hr[i];
      if (v) atomicAdd(dst + (size_t)(i % h.Ng) * Nr + (i / h.Ng), v);
    }
  }
}

#define PRAD_SWEEP_UNROLL 8

// Angles whose march dimension is NOT the contiguous axis: one lane per line.
template <bool DO_GLCM, bool DO_GLRLM, bool LONG>
__global__ void __launch_bounds__(1024) sweep_lines_kernel(SweepSet set, const uint8_t *__restrict__ L, int Ng,
                                                           int Nr, int RS, u32 *__restrict__ glcm_acc,
                                                           u32 *__restrict__ glrlm_acc, int *__restrict__ multi,
                                                           const int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;  // irregular levels: the generic path will redo this call
  const HistLayout h = hist_layout(DO_GLCM, DO_GLRLM, Ng, RS);
  for (int i = threadIdx.x; i < h.words; i += blockDim.x) lds[i] = 0;
  __syncthreads();

  const SweepDesc &D = set.d[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long long nwaves = (long long)gridDim.x * wpb;
  const long long step = D.sM + (long long)D.du * D.sU + D.dx;  // address increment per march step
  Walker<DO_GLCM, DO_GLRLM, LONG> w;
  w.init(lds, h, Nr, glrlm_acc + (size_t)D.slot * Ng * Nr, lane);
  bool seen_multi = false;

  const long long chunk0 = (long long)blockIdx.x * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (long long chunk = chunk0; chunk < D.chunks; chunk += nwaves) {
    const int ui = (int)(chunk / D.LXc);
    const int xc = (int)(chunk - (long long)ui * D.LXc);
    // (wave-uniform; the 64-bit division above is done in VALU, readfirstlane moves the results to SGPRs so
    // that the line base below is a scalar address)
    const int u0 = __builtin_amdgcn_readfirstlane(D.u0min + ui);
    const int xfirst = __builtin_amdgcn_readfirstlane(D.x0min + xc * 64);
    const int x0 = xfirst + lane;
    // march interval during which this lane's line is inside the volume
    int lo = 0, hi = D.NM - 1;
    if (D.du > 0) { lo = max(lo, -u0); hi = min(hi, D.NU - 1 - u0); }
    else if (D.du < 0) { lo = max(lo, u0 - (D.NU - 1)); hi = min(hi, u0); }
    if (D.dx > 0) { lo = max(lo, -x0); hi = min(hi, D.NX - 1 - x0); }
    else if (D.dx < 0) { lo = max(lo, x0 - (D.NX - 1)); hi = min(hi, x0); }
    else if (x0 < 0 || x0 >= D.NX) { lo = 1; hi = 0; }
    const bool live = lo <= hi;
    // readfirstlane: the reductions are wave-uniform by construction; telling the compiler keeps the loop
    // counters, the loop branches and the load base in SGPRs
    const int wlo = __builtin_amdgcn_readfirstlane(wave_min_i32(live ? lo : 0x7fffffff));
    const int whi = __builtin_amdgcn_readfirstlane(wave_max_i32(live ? hi : -1));
    if (wlo > whi) continue;
    // [blo, bhi]: steps at which EVERY lane of the wave is inside (empty if some lane is dead)
    const int blo = __builtin_amdgcn_readfirstlane(wave_max_i32(live ? lo : 0x7fffffff));
    const int bhi = __builtin_amdgcn_readfirstlane(wave_min_i32(live ? hi : -1));
    // wave-uniform base of the 64 lines at march coordinate 0 (may point outside L for dead lanes: never
    // dereferenced there)
    const uint8_t *base = L + ((long long)u0 * D.sU + xfirst);
    const unsigned ulane = (unsigned)lane;  // zero-extended lane offset => global_load ... saddr form
    w.begin_line();
    int t = wlo;
    // head: some lanes have not entered yet
    const int head_end = min(whi, (blo <= bhi ? blo - 1 : whi));
    for (; t <= head_end; t++) {
      const uint8_t *pt = base + (long long)t * step;
      const int cur = (t >= lo && t <= hi) ? (int)pt[ulane] : 0;
      w.step(cur);
    }
    // body: predicate-free, unrolled, loads issued ahead of use
    if (blo <= bhi) {
      for (; t + PRAD_SWEEP_UNROLL - 1 <= bhi; t += PRAD_SWEEP_UNROLL) {
        int v[PRAD_SWEEP_UNROLL];
        const uint8_t *p = base + (long long)t * step;
#pragma unroll
        for (int k = 0; k < PRAD_SWEEP_UNROLL; k++) {
          v[k] = (int)p[ulane];
          p += step;
        }
#pragma unroll
        for (int k = 0; k < PRAD_SWEEP_UNROLL; k++) w.step(v[k]);
      }
      for (; t <= bhi; t++) {
        const uint8_t *pt = base + (long long)t * step;
        w.step((int)pt[ulane]);
      }
    }
    // tail: some lanes have already left
    for (; t <= whi; t++) {
      const uint8_t *pt = base + (long long)t * step;
      const int cur = (t >= lo && t <= hi) ? (int)pt[ulane] : 0;
      w.step(cur);
    }
    seen_multi |= w.end_line();
  }
  if (DO_GLRLM && seen_multi) multi[D.slot] = 1;
  flush_block_hist<DO_GLCM, DO_GLRLM>(lds, h, Nr, D.slot, glcm_acc, glrlm_acc);
}

// The angle along the contiguous axis.  A wave owns 64 consecutive rows (row = flattened (z,y)); it stages
// 64 rows x 64 voxels through LDS (coalesced in, one row per lane out) and walks them with the same Walker.
#define PRAD_ROW_PITCH 80  // bytes per staged row: 64 data + 16 pad => conflict-free ds_read_b128 per lane
template <bool DO_GLCM, bool DO_GLRLM, bool LONG>
__global__ void __launch_bounds__(256) sweep_rows_kernel(const uint8_t *__restrict__ L, long long nrows, int NX,
                                                         int slot, int Ng, int Nr, int RS,
                                                         u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc,
                                                         int *__restrict__ multi, const int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;
  const HistLayout h = hist_layout(DO_GLCM, DO_GLRLM, Ng, RS);
  for (int i = threadIdx.x; i < h.words; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  // per-wave staging tile after the histograms (16-byte aligned)
  uint8_t *tile = reinterpret_cast<uint8_t *>(lds + ((h.words + 3) & ~3)) + (size_t)wave * 64 * PRAD_ROW_PITCH;
  const long long ngroups = (nrows + 63) / 64;
  const long long nwaves = (long long)gridDim.x * wpb;
  Walker<DO_GLCM, DO_GLRLM, LONG> w;
  w.init(lds, h, Nr, glrlm_acc + (size_t)slot * Ng * Nr, lane);
  bool seen_multi = false;
  const bool vec16 = (NX & 15) == 0 && ((uintptr_t)L & 15) == 0;

  for (long long grp = (long long)blockIdx.x * wpb + wave; grp < ngroups; grp += nwaves) {
    const long long r0 = grp * 64;
    w.begin_line();
    for (int xc = 0; xc < NX; xc += 64) {
      // ---- stage rows r0..r0+63, columns xc..xc+63 ----
      if (vec16) {
        // lane -> (row j*16 + lane/4, 16-byte piece lane%4): one instruction covers 16 rows x 64 B
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int rr = j * 16 + (lane >> 2);
          const int cx = xc + (lane & 3) * 16;
          uint4 q = make_uint4(0, 0, 0, 0);
          if (r0 + rr < nrows && cx < NX) q = *reinterpret_cast<const uint4 *>(L + (r0 + rr) * NX + cx);
          *reinterpret_cast<uint4 *>(tile + rr * PRAD_ROW_PITCH + (lane & 3) * 16) = q;
        }
      } else {
        const bool xin = xc + lane < NX;
#pragma unroll 8
        for (int rr = 0; rr < 64; rr++) {
          uint8_t b = 0;
          if (xin && r0 + rr < nrows) b = L[(r0 + rr) * NX + xc + lane];
          tile[rr * PRAD_ROW_PITCH + lane] = b;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // ---- each lane walks its own row (rows >= nrows and columns >= NX were staged as zeros) ----
      const uint4 *row = reinterpret_cast<const uint4 *>(tile + lane * PRAD_ROW_PITCH);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint4 d = row[q];
        const u32 wds[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
          for (int b = 0; b < 4; b++) w.step((int)((wds[k] >> (8 * b)) & 0xffu));
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    seen_multi |= w.end_line();
  }
  if (DO_GLRLM && seen_multi) multi[slot] = 1;
  flush_block_hist<DO_GLCM, DO_GLRLM>(lds, h, Nr, slot, glcm_acc, glrlm_acc);
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
