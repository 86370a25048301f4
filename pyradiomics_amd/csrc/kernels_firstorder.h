// kernels_firstorder.h -- first-order statistics of the ROI intensities (radiomics/firstorder.py:33-474), segment mode.
//   fo_sums_kernel      per-block partial sums of x and (x + c)^2, block minimum / maximum / ROI voxel count
//   fo_compact_kernel   ROI voxels -> dense float64 array (small ROIs / fallback only; order irrelevant: it is sorted)
//   order statistics behind the percentile / median / IQR features: rocPRIM radix sort of the float64 keys for
//   small ROIs, histogram selection (fo_hist_kernel, fo_gather_kernel + a sort of the few selected bins) for large
//   fo_central_kernel   per-block partial sums of |x - mu|, (x - mu)^2..4 and count / sum of the 10-90 percentile band
//   fo_band_kernel      per-block partial sums of |x - mu_band| over the band (robust mean absolute deviation)
// Partials are summed on the host in block order, so results are reproducible run to run.  HBM-bound: every pass
// reads the image at its own dtype + 1 B/voxel of mask.
#pragma once
#include "prad_runtime.h"

namespace prad {

#define PRAD_FO_BLOCKS 1024

// Every block owns a contiguous slab: it counts its ROI voxels, reserves their output range with ONE global atomic
// (a per-wave atomic on the shared cursor serialises ~n/64 same-address operations in L2), then scatters with an LDS
// cursor.  The output order is irrelevant: the values are sorted next.
template <typename T>
__global__ void __launch_bounds__(256) fo_compact_kernel(const T *__restrict__ x, const uint8_t *__restrict__ mask,
                                                         long long n, double *__restrict__ vals,
                                                         unsigned long long *__restrict__ count) {
  __shared__ unsigned cursor, total;
  __shared__ unsigned long long base;
  if (threadIdx.x == 0) { cursor = 0u; total = 0u; }
  __syncthreads();
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = min(n, lo + per);
  const int lane = threadIdx.x & 63;
  unsigned mine = 0;
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) mine += mask[i] != 0;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if (lane == 0 && mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) base = total ? atomicAdd(count, (unsigned long long)total) : 0ull;
  __syncthreads();
  if (!total) return;
  const long long span = hi - lo;
  const long long rounds = (span + blockDim.x - 1) / blockDim.x;
  for (long long r = 0; r < rounds; r++) {
    const long long i = lo + r * blockDim.x + threadIdx.x;
    const bool m = i < hi && mask[i] != 0;
    const unsigned long long B = __ballot(m);
    unsigned wbase = 0;
    if (lane == 0 && B) wbase = atomicAdd(&cursor, (unsigned)__popcll(B));
    wbase = __shfl(wbase, 0);
    if (m) vals[base + wbase + __popcll(B & ((1ull << lane) - 1ull))] = (double)x[i];
  }
}

__device__ __forceinline__ double fo_block_sum(double v, double *sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ROI scan shared by every pass: threads t, t + nthreads, ... take 16-byte pieces (8 int16 / 4 float32 / 2 float64 voxels
// with their 8 / 4 / 2 mask bytes) of [first, last) in a fixed assignment, so a pass issues one wide load per 2..8
// voxels instead of two narrow ones per voxel (the scalar form ran at a fifth of HBM speed: load-issue bound).
// f(x) is called for every ROI voxel of the thread, in index order.  Unaligned views take the scalar form.
template <typename T, typename F>
__device__ __forceinline__ void fo_scan(const T *__restrict__ img, const uint8_t *__restrict__ mask, long long first,
                                        long long last, long long t, long long nthreads, F f) {
  constexpr int E = 16 / (int)sizeof(T);
  if ((((uintptr_t)(img + first)) & 15) == 0 && (((uintptr_t)(mask + first)) & (E - 1)) == 0) {
    const long long nvec = (last - first) / E;
    for (long long v = t; v < nvec; v += nthreads) {
      const long long i = first + v * E;
      const uint4 q = *reinterpret_cast<const uint4 *>(img + i);
      T vals[E];
      memcpy(vals, &q, 16);
      uint8_t mk[E];
      if (E == 8) { const uint2 m = *reinterpret_cast<const uint2 *>(mask + i); memcpy(mk, &m, 8); }
      else if (E == 4) { const unsigned m = *reinterpret_cast<const unsigned *>(mask + i); memcpy(mk, &m, 4); }
      else { const unsigned short m = *reinterpret_cast<const unsigned short *>(mask + i); memcpy(mk, &m, 2); }
#pragma unroll
      for (int e = 0; e < E; e++)
        if (mk[e]) f((double)vals[e]);
    }
    for (long long i = first + nvec * E + t; i < last; i += nthreads)
      if (mask[i]) f((double)img[i]);
  } else {
    for (long long i = first + t; i < last; i += nthreads)
      if (mask[i]) f((double)img[i]);
  }
}

// The reductions read the image and the mask directly, in raster order with a fixed block layout, so their partial
// sums (added on the host in block order) do not depend on the order the compaction happens to produce.
// partial[b][0] = sum x, [1] = sum (x + c)^2, [2] = min x, [3] = max x, [4] = ROI voxels of the block (a block
// without ROI voxels reports +inf / -inf / 0)
template <typename T>
__global__ void __launch_bounds__(256) fo_sums_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                      long long n, double shift, double *__restrict__ partial) {
  __shared__ double sh[4];
  double s1 = 0, s2 = 0, mn = INFINITY, mx = -INFINITY, cnt = 0;
  fo_scan(img, mask, 0, n, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x, [&](double x) {
    const double y = x + shift;
    s1 += x;
    s2 += y * y;
    mn = fmin(mn, x);
    mx = fmax(mx, x);
    cnt += 1.0;
  });
  s1 = fo_block_sum(s1, sh);
  s2 = fo_block_sum(s2, sh);
  cnt = fo_block_sum(cnt, sh);
  for (int o = 32; o > 0; o >>= 1) {
    mn = fmin(mn, __shfl_xor(mn, o));
    mx = fmax(mx, __shfl_xor(mx, o));
  }
  __shared__ double shm[8];
  if ((threadIdx.x & 63) == 0) {
    shm[threadIdx.x >> 6] = mn;
    shm[4 + (threadIdx.x >> 6)] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double *p = partial + blockIdx.x * 5;
    p[0] = s1;
    p[1] = s2;
    p[2] = fmin(fmin(shm[0], shm[1]), fmin(shm[2], shm[3]));
    p[3] = fmax(fmax(shm[4], shm[5]), fmax(shm[6], shm[7]));
    p[4] = cnt;
  }
}

// ---- order statistics by selection (large ROIs) ------------------------------------------------------------------
// The percentile features need a dozen order statistics, not a sorted array.  Values are binned by the monotone map
// bin(x) = min(BINS - 1, int((x - min) * scale)); a histogram pass locates the bins that hold the wanted ranks, a
// gather pass copies just those bins' elements out, and only they are sorted: 2-3 passes over image + mask instead
// of a compaction and the 8 read+write passes of a 64-bit radix sort.
#define PRAD_FO_BINS 16384
#define PRAD_FO_MAXSEL 12
__device__ __forceinline__ int fo_bin(double x, double lo, double scale) {
  const double t = (x - lo) * scale;
  return t >= (double)(PRAD_FO_BINS - 1) ? PRAD_FO_BINS - 1 : (int)t;
}
template <typename T>
__device__ __forceinline__ void fo_hist_body(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                             long long n, double lo, double scale,
                                             unsigned *__restrict__ hist) {
  __shared__ unsigned h[PRAD_FO_BINS];
  for (int k = threadIdx.x; k < PRAD_FO_BINS; k += blockDim.x) h[k] = 0u;
  __syncthreads();
  fo_scan(img, mask, 0, n, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x,
          [&](double x) { atomicAdd(&h[fo_bin(x, lo, scale)], 1u); });
  __syncthreads();
  for (int k = threadIdx.x; k < PRAD_FO_BINS; k += blockDim.x)
    if (h[k]) atomicAdd(hist + k, h[k]);
}
template <typename T>
__global__ void __launch_bounds__(1024) fo_hist_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                       long long n, double lo, double scale,
                                                       unsigned *__restrict__ hist) {
  fo_hist_body(img, mask, n, lo, scale, hist);
}
// Integer images (int16 / int32: CT, MR, level images) whose value range fits the LDS: the EXACT histogram of the ROI,
// count of every value base .. base + R - 1.  Every first-order statistic is a function of it (sums over values
// weighted by counts, order statistics from the cumulative counts), so this pass replaces hist + gather + sort +
// central + band (four more reads of the volume).
#define PRAD_FO_EXACT_MAX 32768
template <typename T>
__global__ void __launch_bounds__(1024) fo_exact_hist_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                             long long n, double base, int R,
                                                             unsigned *__restrict__ hist) {
  extern __shared__ unsigned fo_h[];
  for (int k = threadIdx.x; k < R; k += blockDim.x) fo_h[k] = 0u;
  __syncthreads();
  fo_scan(img, mask, 0, n, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x,
          [&](double x) {
            const int k = (int)(x - base);        // exact: integers below 2^31
            if ((unsigned)k < (unsigned)R) atomicAdd(&fo_h[k], 1u);
          });
  __syncthreads();
  for (int k = threadIdx.x; k < R; k += blockDim.x)
    if (fo_h[k]) atomicAdd(hist + k, fo_h[k]);
}

struct FoSel {
  int nsel;
  int bin[PRAD_FO_MAXSEL];        // ascending
  unsigned off[PRAD_FO_MAXSEL];   // start of the bin's segment in the gathered array
};
// smallest / largest element of every selected bin, as order-preserving u64 keys (fo_key): a bin whose two keys
// agree holds ONE distinct value (discretised / integer images put millions of equal voxels into a bin), and every
// rank inside it is that value -- no gather, no sort.
__device__ __host__ __forceinline__ unsigned long long fo_key(double x) {
  unsigned long long b;
  memcpy(&b, &x, sizeof(b));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __host__ __forceinline__ double fo_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double x;
  memcpy(&x, &b, sizeof(x));
  return x;
}
template <typename T>
__global__ void __launch_bounds__(1024) fo_binrange_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                          long long n, double lo, double scale, FoSel sel,
                                                          unsigned long long *__restrict__ range /*[MAXSEL][2]*/) {
  __shared__ unsigned long long r[PRAD_FO_MAXSEL][2];
  if (threadIdx.x < PRAD_FO_MAXSEL) {
    r[threadIdx.x][0] = ~0ull;
    r[threadIdx.x][1] = 0ull;
  }
  __syncthreads();
  fo_scan(img, mask, 0, n, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x, [&](double x) {
    const int b = fo_bin(x, lo, scale);
    if (b < sel.bin[0] || b > sel.bin[sel.nsel - 1]) return;
    for (int q = 0; q < sel.nsel; q++)
      if (b == sel.bin[q]) {
        const unsigned long long k = fo_key(x);
        if (k < r[q][0]) atomicMin(&r[q][0], k);     // the plain read skips the atomic once the extremes are known
        if (k > r[q][1]) atomicMax(&r[q][1], k);
      }
  });
  __syncthreads();
  if (threadIdx.x < sel.nsel) {   // same-address global atomics serialise in L2: only when they can change the value
    unsigned long long *g = range + 2 * threadIdx.x;
    if (r[threadIdx.x][0] < __builtin_nontemporal_load(g)) atomicMin(g, r[threadIdx.x][0]);
    if (r[threadIdx.x][1] > __builtin_nontemporal_load(g + 1)) atomicMax(g + 1, r[threadIdx.x][1]);
  }
}

// Every block owns a contiguous slab of the volume: it counts its elements of every selected bin in LDS, reserves
// their output ranges with ONE global atomic per bin (per-wave atomics on a dozen cursors serialise in L2), then
// scatters with LDS cursors.  The order inside a bin's segment is irrelevant: the segments are sorted next.
template <typename T>
__device__ __forceinline__ void fo_gather_body(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                               long long n, double lo, double scale, const FoSel &sel,
                                               unsigned *__restrict__ cursors, double *__restrict__ out) {
  __shared__ unsigned cnt[PRAD_FO_MAXSEL], base[PRAD_FO_MAXSEL];
  if (threadIdx.x < PRAD_FO_MAXSEL) cnt[threadIdx.x] = 0u;
  __syncthreads();
  const long long per = (((n + gridDim.x - 1) / gridDim.x) + 15) & ~15LL;   // slabs start on 16-voxel boundaries (wide loads)
  const long long first = min(n, (long long)blockIdx.x * per), last = min(n, first + per);
  auto which = [&](double x) -> int {
    const int b = fo_bin(x, lo, scale);
    int j = -1;
    if (b >= sel.bin[0] && b <= sel.bin[sel.nsel - 1])
      for (int q = 0; q < sel.nsel; q++)
        if (b == sel.bin[q]) j = q;
    return j;
  };
  fo_scan(img, mask, first, last, (long long)threadIdx.x, (long long)blockDim.x, [&](double x) {
    const int j = which(x);
    if (j >= 0) atomicAdd(&cnt[j], 1u);
  });
  __syncthreads();
  if (threadIdx.x < sel.nsel) {
    base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(cursors + threadIdx.x, cnt[threadIdx.x]) : 0u;
    cnt[threadIdx.x] = 0u;
  }
  __syncthreads();
  fo_scan(img, mask, first, last, (long long)threadIdx.x, (long long)blockDim.x, [&](double x) {
    const int j = which(x);
    if (j >= 0) out[sel.off[j] + base[j] + atomicAdd(&cnt[j], 1u)] = x;
  });
}
template <typename T>
__global__ void __launch_bounds__(1024) fo_gather_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                         long long n, double lo, double scale, FoSel sel,
                                                         unsigned *__restrict__ cursors, double *__restrict__ out) {
  fo_gather_body(img, mask, n, lo, scale, sel, cursors, out);
}

// partial[b][0..3] = sum |d|, d^2, d^3, d^4 with d = x - mu; [4] = count, [5] = sum of x with lo <= x <= hi
template <typename T>
__device__ __forceinline__ void fo_central_body(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                long long n, double mu, double lo, double hi,
                                                double *__restrict__ partial) {
#pragma clang fp contract(off)
  __shared__ double sh[4];
  double a1 = 0, a2 = 0, a3 = 0, a4 = 0, bc = 0, bs = 0;
  fo_scan(img, mask, 0, n, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x, [&](double x) {
    const double d = x - mu, d2 = d * d;
    a1 += fabs(d);
    a2 += d2;
    a3 += d2 * d;
    a4 += d2 * d2;
    if (x >= lo && x <= hi) {
      bc += 1.0;
      bs += x;
    }
  });
  a1 = fo_block_sum(a1, sh);
  a2 = fo_block_sum(a2, sh);
  a3 = fo_block_sum(a3, sh);
  a4 = fo_block_sum(a4, sh);
  bc = fo_block_sum(bc, sh);
  bs = fo_block_sum(bs, sh);
  if (threadIdx.x == 0) {
    double *p = partial + blockIdx.x * 6;
    p[0] = a1; p[1] = a2; p[2] = a3; p[3] = a4; p[4] = bc; p[5] = bs;
  }
}
template <typename T>
__global__ void __launch_bounds__(256) fo_central_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                         long long n, double mu, double lo, double hi,
                                                         double *__restrict__ partial) {
  fo_central_body(img, mask, n, mu, lo, hi, partial);
}

template <typename T>
__device__ __forceinline__ void fo_band_body(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                             long long n, double mu, double lo, double hi,
                                             double *__restrict__ partial) {
  __shared__ double sh[4];
  double a = 0;
  fo_scan(img, mask, 0, n, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x, [&](double x) {
    if (x >= lo && x <= hi) a += fabs(x - mu);
  });
  a = fo_block_sum(a, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
template <typename T>
__global__ void __launch_bounds__(256) fo_band_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                      long long n, double mu, double lo, double hi,
                                                      double *__restrict__ partial) {
  fo_band_body(img, mask, n, mu, lo, hi, partial);
}

// ---- the same chain with its scalars in DEVICE memory (prad_firstorder_queue_dev: no host round trip between the passes) --
// What the host does between the passes of prad_firstorder_dev -- add up the block partials in block order, place the
// quantiles, find the histogram bins that hold their ranks, interpolate -- is done by single-workgroup glue kernels on an
// FoDev record; the scan kernels read their parameters from it.  The same operations on the same operands in the same
// order: the same bits.  `problem` != 0 (ROI count different from the caller's, constant ROI, selected bins above the
// gather capacity) makes every later kernel a no-op; the caller repeats the image on the synchronous route.
struct FoDev {
  double s1, s2, vmin, vmax, cnt, mu, scale;
  long long m;
  long long ranks[10];
  double gamma[5];
  FoSel sel;
  long long pick[10];      // index of every order statistic in the sorted gather
  unsigned seg_off[10], seg_cnt[10];   // the gathered segment (one histogram bin) each order statistic lies in ...
  long long within[10];                // ... and its rank inside that segment
  unsigned total;          // gathered elements
  int problem;
  double os[12], pq[5], median;
  double cen[6], mu_band, band;
};

// numpy's linear-interpolation quantile position (prad_firstorder.hip quantile_pos) and _lerp
__device__ __forceinline__ void fo_quantile_pos(long long m, double q, long long *prev, long long *next, double *gamma) {
#pragma clang fp contract(off)
  const double virt = (double)(m - 1) * q;
  long long p = (long long)floor(virt);
  *gamma = virt - (double)p;
  if (p < 0) p = 0;
  if (p > m - 1) p = m - 1;
  *prev = p;
  *next = p + 1 > m - 1 ? m - 1 : p + 1;
}
__device__ __forceinline__ double fo_lerp_np(double a, double b, double t) {
#pragma clang fp contract(off)
  const double d = b - a;
  return t >= 0.5 ? b - d * (1 - t) : a + d * t;
}

// after fo_sums_kernel: block partials -> sums, extremes, count, mean, quantile ranks, histogram scale
__global__ void fo_glue_sums_kernel(const double *__restrict__ partial, int blocks, long long expect_m, FoDev *st) {
#pragma clang fp contract(off)
  __shared__ double col[5];
  __shared__ double stage[PRAD_FO_BLOCKS * 6];     // (a chain of 1024 dependent global loads per column took 167 us)
  const int t = threadIdx.x;
  for (int i = t; i < blocks * 5; i += blockDim.x) stage[i] = partial[i];
  __syncthreads();
  if (t == 0 || t == 1 || t == 4) {          // wave 0: the three sums, in block order
    double acc = 0.0;
    for (int b = 0; b < blocks; b++) acc += stage[b * 5 + t];
    col[t] = acc;
  } else if (t == 64) {                      // waves 1 and 2: the extremes (a divergent loop each would triple the chain)
    double acc = INFINITY;
    for (int b = 0; b < blocks; b++) acc = fmin(acc, stage[b * 5 + 2]);
    col[2] = acc;
  } else if (t == 128) {
    double acc = -INFINITY;
    for (int b = 0; b < blocks; b++) acc = fmax(acc, stage[b * 5 + 3]);
    col[3] = acc;
  }
  __syncthreads();
  if (t == 0) {
    st->s1 = col[0]; st->s2 = col[1]; st->vmin = col[2]; st->vmax = col[3]; st->cnt = col[4];
    const long long m = (long long)col[4];
    st->m = m;
    int problem = 0;
    if (m != expect_m || m < 1) problem |= 1;
    if (!(col[3] > col[2]) || !isfinite(col[2]) || !isfinite(col[3])) problem |= 2;
    st->mu = m > 0 ? col[0] / (double)m : 0.0;
    st->scale = (double)PRAD_FO_BINS / (col[3] - col[2]);
    const double qs[5] = {0.1, 0.25, 0.5, 0.75, 0.9};
    for (int k = 0; k < 5; k++) {
      long long p = 0, nx = 0;
      double g = 0;
      if (m > 0) fo_quantile_pos(m, qs[k], &p, &nx, &g);
      st->ranks[2 * k] = p;
      st->ranks[2 * k + 1] = nx;
      st->gamma[k] = g;
    }
    st->os[10] = col[2];
    st->os[11] = col[3];
    st->problem = problem;
    st->total = 0u;
  }
}

template <typename T>
__global__ void __launch_bounds__(1024) fo_hist_dev_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                           long long n, const FoDev *__restrict__ st,
                                                           unsigned *__restrict__ hist) {
  if (st->problem) return;
  fo_hist_body(img, mask, n, st->vmin, st->scale, hist);
}

// after the histogram: the bin of every rank and the rank's position inside it, the distinct bins with their segment
// offsets (prad_firstorder_dev's host loop: ranks ascend, so do the bins)
__global__ void __launch_bounds__(1024) fo_glue_select_kernel(const unsigned *__restrict__ hist, unsigned capacity, FoDev *st) {
  __shared__ unsigned long long scan[1024];
  __shared__ int rbin[10];
  __shared__ long long rwithin[10];
  const int t = threadIdx.x;
  if (st->problem) return;
  constexpr int PER = PRAD_FO_BINS / 1024;
  unsigned h[PER];
  unsigned long long mine = 0;
  for (int j = 0; j < PER; j++) {
    h[j] = hist[t * PER + j];
    mine += h[j];
  }
  scan[t] = mine;
  if (t < 10) rbin[t] = -1;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned long long v = t >= o ? scan[t - o] : 0ull;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  unsigned long long below = scan[t] - mine;      // elements in the bins before this thread's
  for (int j = 0; j < PER; j++) {
    const unsigned long long upto = below + h[j];
    if (h[j])
      for (int k = 0; k < 10; k++) {
        const unsigned long long r = (unsigned long long)st->ranks[k];
        if (r >= below && r < upto) {
          rbin[k] = t * PER + j;
          rwithin[k] = (long long)(r - below);
        }
      }
    below = upto;
  }
  __syncthreads();
  if (t == 0) {
    FoSel sel;
    sel.nsel = 0;
    int problem = 0;
    unsigned long long total = 0;
    int which[10];
    for (int k = 0; k < 10; k++) {
      if (rbin[k] < 0) { problem |= 4; break; }
      if (sel.nsel == 0 || sel.bin[sel.nsel - 1] != rbin[k]) {
        sel.bin[sel.nsel] = rbin[k];
        sel.off[sel.nsel] = (unsigned)total;
        total += hist[rbin[k]];
        sel.nsel++;
      }
      which[k] = sel.nsel - 1;
    }
    if (total > capacity) problem |= 8;
    if (!problem)
      for (int k = 0; k < 10; k++) {
        st->pick[k] = (long long)sel.off[which[k]] + rwithin[k];
        st->seg_off[k] = sel.off[which[k]];
        st->seg_cnt[k] = hist[rbin[k]];
        st->within[k] = rwithin[k];
      }
    for (int q = sel.nsel; q < PRAD_FO_MAXSEL; q++) { sel.bin[q] = 0; sel.off[q] = 0u; }
    st->sel = sel;
    st->total = (unsigned)(total > capacity ? 0 : total);
    st->problem = problem;
  }
}

template <typename T>
__global__ void __launch_bounds__(1024) fo_gather_dev_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                             long long n, const FoDev *__restrict__ st,
                                                             unsigned *__restrict__ cursors, double *__restrict__ out) {
  if (st->problem) return;
  __shared__ FoSel sel;
  if (threadIdx.x == 0) sel = st->sel;
  __syncthreads();
  fo_gather_body(img, mask, n, st->vmin, st->scale, sel, cursors, out);
}

// The order statistics straight from the gathered (unsorted) segments: workgroup r finds the element of rank within[r] in
// its segment by a radix select over the order-preserving 64-bit keys, most significant byte first -- 8 rounds of
// { count the next byte among the keys that match the prefix so far, pick the byte whose range holds the rank }.  One
// launch instead of the 16 of a device-wide radix sort of the padded gather, and its cost follows the real segment sizes.
// Counts go through per-wave tables, a lane merging the consecutive equal bytes it meets first (a bin of equal values --
// integer-valued images -- would otherwise put every lane on one counter).
__global__ void __launch_bounds__(1024) fo_rank_select_kernel(const double *__restrict__ gath, FoDev *st) {
  __shared__ unsigned cnt[16][256];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned long long s_k;
  if (st->problem) return;
  const int r = blockIdx.x, t = threadIdx.x, w = t >> 6;
  const double *seg = gath + st->seg_off[r];
  const unsigned n = st->seg_cnt[r];
  if (t == 0) {
    s_prefix = 0ull;
    s_k = (unsigned long long)st->within[r];
  }
  __syncthreads();
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int i = t; i < 16 * 256; i += 1024) (&cnt[0][0])[i] = 0u;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const unsigned long long mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
    int last = -1;
    unsigned run = 0;
    for (unsigned i = t; i < n; i += 1024) {
      const unsigned long long key = fo_key(seg[i]);
      if ((key & mask) != (prefix & mask)) continue;
      const int d = (int)((key >> shift) & 255ull);
      if (d != last) {
        if (run) atomicAdd(&cnt[w][last], run);
        last = d;
        run = 0;
      }
      run++;
    }
    if (run) atomicAdd(&cnt[w][last], run);
    __syncthreads();
    if (t < 256) {
      unsigned v = 0;
      for (int q = 0; q < 16; q++) v += cnt[q][t];
      cnt[0][t] = v;
    }
    __syncthreads();
    if (t == 0) {
      unsigned long long k = s_k, cum = 0;
      int d = 0;
      for (; d < 255; d++) {
        if (k < cum + cnt[0][d]) break;
        cum += cnt[0][d];
      }
      s_prefix = prefix | ((unsigned long long)d << shift);
      s_k = k - cum;
    }
    __syncthreads();
  }
  if (t == 0) st->os[r] = fo_unkey(s_prefix);
}

// after fo_rank_select_kernel: interpolated percentiles, median
__global__ void fo_glue_quantiles_kernel(FoDev *st) {
#pragma clang fp contract(off)
  if (threadIdx.x != 0 || st->problem) return;
  for (int k = 0; k < 5; k++) st->pq[k] = fo_lerp_np(st->os[2 * k], st->os[2 * k + 1], st->gamma[k]);
  st->median = (st->m % 2) ? st->os[4] : (st->os[4] + st->os[5]) / 2.0;
}

template <typename T>
__global__ void __launch_bounds__(256) fo_central_dev_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                             long long n, const FoDev *__restrict__ st,
                                                             double *__restrict__ partial) {
  if (st->problem) return;
  fo_central_body(img, mask, n, st->mu, st->pq[0], st->pq[4], partial);
}

__global__ void fo_glue_central_kernel(const double *__restrict__ partial, int blocks, FoDev *st) {
#pragma clang fp contract(off)
  const int t = threadIdx.x;
  if (st->problem) return;
  __shared__ double col[6];
  __shared__ double stage[PRAD_FO_BLOCKS * 6];
  for (int i = t; i < blocks * 6; i += blockDim.x) stage[i] = partial[i];
  __syncthreads();
  if (t < 6) {
    double acc = 0;
    for (int b = 0; b < blocks; b++) acc += stage[b * 6 + t];
    col[t] = acc;
  }
  __syncthreads();
  if (t == 0) {
    for (int k = 0; k < 6; k++) st->cen[k] = col[k];
    st->mu_band = col[4] > 0 ? col[5] / col[4] : 0.0;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) fo_band_dev_kernel(const T *__restrict__ img, const uint8_t *__restrict__ mask,
                                                          long long n, const FoDev *__restrict__ st,
                                                          double *__restrict__ partial) {
  if (st->problem) return;
  if (!(st->cen[4] > 0)) {
    if (threadIdx.x == 0) partial[blockIdx.x] = 0.0;
    return;
  }
  fo_band_body(img, mask, n, st->mu_band, st->pq[0], st->pq[4], partial);
}

// the 15 statistics in the order of PRAD_FO_* (include/pyradiomics_amd.h), then the verdict word
__global__ void fo_glue_final_kernel(const double *__restrict__ partial, int blocks, const FoDev *__restrict__ st,
                                     double *__restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double stage[PRAD_FO_BLOCKS];
  if (!st->problem)
    for (int i = threadIdx.x; i < blocks; i += blockDim.x) stage[i] = partial[i];
  __syncthreads();
  if (threadIdx.x != 0) return;
  out[15] = (double)st->problem;
  if (st->problem) {
    for (int k = 0; k < 15; k++) out[k] = 0.0;
    return;
  }
  double band = 0;
  for (int b = 0; b < blocks; b++) band += stage[b];
  const double dm = (double)st->m;
  out[0] = dm;                         // PRAD_FO_NP
  out[1] = st->s2;                     // ENERGY
  out[2] = st->os[10];                 // MINIMUM
  out[3] = st->pq[0];                  // P10
  out[4] = st->pq[1];                  // P25
  out[5] = st->median;                 // MEDIAN
  out[6] = st->pq[3];                  // P75
  out[7] = st->pq[4];                  // P90
  out[8] = st->os[11];                 // MAXIMUM
  out[9] = st->mu;                     // MEAN
  out[10] = st->cen[0] / dm;           // MAD
  out[11] = st->cen[4] > 0 ? band / st->cen[4] : __builtin_nan("");   // RMAD
  out[12] = st->cen[1] / dm;           // M2
  out[13] = st->cen[2] / dm;           // M3
  out[14] = st->cen[3] / dm;           // M4
}

// ---- voxel mode (firstorder.py:37-118): one wave per centre voxel ------------------------------------------------
// The reference gathers, for every centre, the intensities at centre + kernelOffsets from a NaN-padded copy of the
// image in which non-ROI voxels are NaN, then applies the nan-aware numpy reductions row by row.  Here a 64-lane
// block loads the window into LDS (invalid slots = +inf, level 0), reduces it with wave shuffles, sorts it with a
// bitonic network for the order statistics and writes only the requested feature values: no (Nvox, Nk) intermediate.
struct FoWindow {
  int nd;
  int half[PRAD_MAX_ND];     // per-dimension half width: min(kernelRadius, boundingBoxSize - 1), 0 in the force2D dimension
  int nk;                    // prod(2 half + 1)
  int P;                     // power of two >= nk (LDS slots)
};

enum { PRAD_FOF_ENERGY = 0, PRAD_FOF_TOTALENERGY, PRAD_FOF_ENTROPY, PRAD_FOF_MINIMUM, PRAD_FOF_P10, PRAD_FOF_P90,
       PRAD_FOF_MAXIMUM, PRAD_FOF_MEAN, PRAD_FOF_MEDIAN, PRAD_FOF_IQR, PRAD_FOF_RANGE, PRAD_FOF_MAD, PRAD_FOF_RMAD,
       PRAD_FOF_RMS, PRAD_FOF_STD, PRAD_FOF_SKEWNESS, PRAD_FOF_KURTOSIS, PRAD_FOF_VARIANCE, PRAD_FOF_UNIFORMITY,
       PRAD_FOF_COUNT };

__device__ __forceinline__ double fo_wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double fo_quantile(const double *sorted, int m, double q) {
#pragma clang fp contract(off)
  const double virt = (double)(m - 1) * q;
  int prev = (int)floor(virt);
  const double t = virt - (double)prev;
  prev = min(max(prev, 0), m - 1);
  const int next = min(prev + 1, m - 1);
  const double a = sorted[prev], b = sorted[next], d = b - a;
  return t >= 0.5 ? b - d * (1 - t) : a + d * t;
}

template <typename T>
__global__ void __launch_bounds__(64) voxel_firstorder_kernel(const T *__restrict__ image,
                                                              const uint8_t *__restrict__ mask,
                                                              const int *__restrict__ levels, Geo g, FoWindow w,
                                                              int nvox, const int *__restrict__ voxels, double shift,
                                                              double voxel_volume, const int *__restrict__ feature_ids,
                                                              int nfeat, double *__restrict__ out) {
#pragma clang fp contract(off)
  extern __shared__ double fo_lds[];
  double *val = fo_lds;                 // [P]
  int *lev = (int *)(fo_lds + w.P);     // [P]
  const int lane = threadIdx.x;
  const double INF = __builtin_huge_val();
  for (int v = blockIdx.x; v < nvox; v += gridDim.x) {
    __syncthreads();
    int c[PRAD_MAX_ND];
    for (int d = 0; d < g.nd; d++) c[d] = voxels[(long long)d * nvox + v];
    // load the window
    double s1 = 0, s2 = 0, mn = INF, mx = -INF, cnt = 0;
    for (int k = lane; k < w.P; k += 64) {
      double x = INF;
      int lv = 0;
      if (k < w.nk) {
        int rem = k;
        long long idx = 0;
        bool in = true;
        for (int d = g.nd - 1; d >= 0; d--) {
          const int ext = 2 * w.half[d] + 1;
          const int q = c[d] + (rem % ext) - w.half[d];
          rem /= ext;
          in = in && q >= 0 && q < g.size[d];
          idx += (long long)q * g.stride[d];
        }
        if (in && mask[idx]) {
          x = (double)image[idx];
          lv = levels ? levels[idx] : 0;
          const double y = x + shift;
          s1 += x;
          s2 += y * y;
          mn = fmin(mn, x);
          mx = fmax(mx, x);
          cnt += 1.0;
        }
      }
      val[k] = x;
      lev[k] = lv;
    }
    s1 = fo_wave_sum(s1);
    s2 = fo_wave_sum(s2);
    cnt = fo_wave_sum(cnt);
    for (int o = 32; o > 0; o >>= 1) {
      mn = fmin(mn, __shfl_xor(mn, o));
      mx = fmax(mx, __shfl_xor(mx, o));
    }
    const int m = (int)cnt;
    const double mu = s1 / cnt;
    __syncthreads();
    // central moments, and per-voxel multiplicity of its grey level (entropy / uniformity without a histogram)
    double a1 = 0, a2 = 0, a3 = 0, a4 = 0, ent = 0, uni = 0;
    for (int k = lane; k < w.nk; k += 64) {
      const double x = val[k];
      if (x == INF) continue;
      const double d = x - mu, d2 = d * d;
      a1 += fabs(d);
      a2 += d2;
      a3 += d2 * d;
      a4 += d2 * d2;
      const int lv = lev[k];
      int same = 0;
      for (int j = 0; j < w.nk; j++) same += (lev[j] == lv && val[j] != INF) ? 1 : 0;
      const double p = (double)same / cnt;
      ent += log2(p + 2.220446049250313e-16) / cnt;
      uni += p / cnt;
    }
    a1 = fo_wave_sum(a1);
    a2 = fo_wave_sum(a2);
    a3 = fo_wave_sum(a3);
    a4 = fo_wave_sum(a4);
    ent = -fo_wave_sum(ent);
    uni = fo_wave_sum(uni);
    // bitonic sort of val[0..P) ascending (+inf padding ends up last)
    for (int size = 2; size <= w.P; size <<= 1) {
      for (int str = size >> 1; str > 0; str >>= 1) {
        __syncthreads();
        for (int t = lane; t < (w.P >> 1); t += 64) {
          const int i = ((t / str) * str << 1) + (t % str), j = i + str;
          const bool up = ((i & size) == 0);
          const double a = val[i], b = val[j];
          if ((a > b) == up) {
            val[i] = b;
            val[j] = a;
          }
        }
      }
    }
    __syncthreads();
    const double p10 = fo_quantile(val, m, 0.1), p25 = fo_quantile(val, m, 0.25), p75 = fo_quantile(val, m, 0.75),
                 p90 = fo_quantile(val, m, 0.9);
    const double med = (m & 1) ? val[m >> 1] : (val[(m >> 1) - 1] + val[m >> 1]) / 2.0;
    double bc = 0, bs = 0;
    for (int k = lane; k < m; k += 64) {
      const double x = val[k];
      if (x >= p10 && x <= p90) {
        bc += 1.0;
        bs += x;
      }
    }
    bc = fo_wave_sum(bc);
    bs = fo_wave_sum(bs);
    const double mub = bs / bc;
    double ba = 0;
    for (int k = lane; k < m; k += 64) {
      const double x = val[k];
      if (x >= p10 && x <= p90) ba += fabs(x - mub);
    }
    ba = fo_wave_sum(ba);
    if (lane == 0) {
      const double m2 = a2 / cnt, m3 = a3 / cnt, m4 = a4 / cnt;
      const double m2s = m2 == 0 ? 1.0 : m2;            // firstorder.py:403-405,441-443: flat regions give 0
      for (int f = 0; f < nfeat; f++) {
        double r;
        switch (feature_ids[f]) {
          case PRAD_FOF_ENERGY: r = s2; break;
          case PRAD_FOF_TOTALENERGY: r = s2 * voxel_volume; break;
          case PRAD_FOF_ENTROPY: r = ent; break;
          case PRAD_FOF_MINIMUM: r = mn; break;
          case PRAD_FOF_P10: r = p10; break;
          case PRAD_FOF_P90: r = p90; break;
          case PRAD_FOF_MAXIMUM: r = mx; break;
          case PRAD_FOF_MEAN: r = mu; break;
          case PRAD_FOF_MEDIAN: r = med; break;
          case PRAD_FOF_IQR: r = p75 - p25; break;
          case PRAD_FOF_RANGE: r = mx - mn; break;
          case PRAD_FOF_MAD: r = a1 / cnt; break;
          case PRAD_FOF_RMAD: r = ba / bc; break;
          case PRAD_FOF_RMS: r = sqrt(s2 / cnt); break;
          case PRAD_FOF_STD: r = sqrt(m2); break;
          case PRAD_FOF_SKEWNESS: r = m3 / pow(m2s, 1.5); break;
          case PRAD_FOF_KURTOSIS: r = m4 / (m2s * m2s); break;
          case PRAD_FOF_VARIANCE: { const double sd = sqrt(m2); r = sd * sd; } break;
          case PRAD_FOF_UNIFORMITY: r = uni; break;
          default: r = __builtin_nan("");
        }
        out[(long long)f * nvox + v] = r;
      }
    }
  }
}

}  // namespace prad
