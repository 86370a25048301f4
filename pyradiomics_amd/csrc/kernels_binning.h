// kernels_binning.h -- on-device grey-level discretisation (radiomics/imageoperations.py:67-174, base.py:119-125):
//   roi_minmax_kernel   min / max of the image over the ROI (the only data-dependent input of getBinEdges)
//   digitize_kernel     level = np.digitize(x, edges) = #{edges <= x} inside the ROI, 0 outside; also the largest level
// The edge array itself is built on the host with numpy from (min, max) exactly as getBinEdges does, so the
// comparison sequence -- and therefore every level -- is identical to the reference.  HBM-bound, one pass each.
#pragma once
#include "prad_runtime.h"

namespace prad {

// monotone map double -> uint64 so that min / max can use integer atomics
__device__ __forceinline__ unsigned long long f64_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double f64_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double d;
  memcpy(&d, &b, sizeof(d));
  return d;
}

// 16 bytes of the image + the matching mask bytes per lane and load (one element per lane and load ran these passes at
// 1.7 - 2.7 TB/s: load-issue bound).  f(index of the first element, values, mask bytes) for every vector of E elements,
// g(i) for the unaligned / leftover elements.
template <typename T, typename F, typename G>
__device__ __forceinline__ void bin_scan(const T *__restrict__ x, const uint8_t *__restrict__ mask, long long n,
                                         bool aligned, F f, G g) {
  constexpr int E = 16 / (int)sizeof(T);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long long)gridDim.x * blockDim.x;
  long long done = 0;
  if (aligned) {
    const long long nvec = n / E;
    constexpr int U = 4;                      // independent loads in flight per lane
    long long v = t;
    for (; v + (U - 1) * nthreads < nvec; v += U * nthreads) {
      uint4 q[U];
      uint8_t mk[U][E];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long long i = (v + u * nthreads) * E;
        q[u] = *reinterpret_cast<const uint4 *>(x + i);
        if (E == 8) { const uint2 m = *reinterpret_cast<const uint2 *>(mask + i); memcpy(mk[u], &m, 8); }
        else if (E == 4) { const unsigned m = *reinterpret_cast<const unsigned *>(mask + i); memcpy(mk[u], &m, 4); }
        else { const unsigned short m = *reinterpret_cast<const unsigned short *>(mask + i); memcpy(mk[u], &m, 2); }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        T vals[E];
        memcpy(vals, &q[u], 16);
        f((v + u * nthreads) * E, vals, mk[u]);
      }
    }
    for (; v < nvec; v += nthreads) {
      const long long i = v * E;
      const uint4 q = *reinterpret_cast<const uint4 *>(x + i);
      T vals[E];
      memcpy(vals, &q, 16);
      uint8_t mk[E];
      if (E == 8) { const uint2 m = *reinterpret_cast<const uint2 *>(mask + i); memcpy(mk, &m, 8); }
      else if (E == 4) { const unsigned m = *reinterpret_cast<const unsigned *>(mask + i); memcpy(mk, &m, 4); }
      else { const unsigned short m = *reinterpret_cast<const unsigned short *>(mask + i); memcpy(mk, &m, 2); }
      f(i, vals, mk);
    }
    done = nvec * E;
  }
  for (long long i = done + t; i < n; i += nthreads) g(i);
}

template <typename T>
__global__ void __launch_bounds__(256) roi_minmax_kernel(const T *__restrict__ x, const uint8_t *__restrict__ mask,
                                                         long long n, unsigned long long *__restrict__ keys) {
  constexpr int E = 16 / (int)sizeof(T);
  unsigned long long lo = ~0ull, hi = 0ull;
  auto one = [&](double v) {
    const unsigned long long k = f64_key(v);
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  };
  const bool aligned = (((uintptr_t)x) & 15) == 0 && (((uintptr_t)mask) & (E - 1)) == 0;
  bin_scan(x, mask, n, aligned,
           [&](long long, const T (&vals)[E], const uint8_t (&mk)[E]) {
#pragma unroll
             for (int e = 0; e < E; e++)
               if (mk[e]) one((double)vals[e]);
           },
           [&](long long i) {
             if (mask[i]) one((double)x[i]);
           });
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  // one pair of global atomics per block: thousands of waves hitting the same two addresses serialise in L2
  __shared__ unsigned long long slo[4], shi[4];
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      lo = slo[w] < lo ? slo[w] : lo;
      hi = shi[w] > hi ? shi[w] : hi;
    }
    if (hi != 0ull) {
      atomicMin(keys, lo);
      atomicMax(keys + 1, hi);
    }
  }
}

// LDS_EDGES: the edge list is staged in LDS (<= 7680 edges), otherwise it is searched in global memory (binWidth 1 on a
// 0..30000 image: 30 000 edges, cache-resident).  COUNTS: 0 = none, 1 = per-wave private LDS tables of nedges + 1 words
// (one pass gives levels AND the ROI voxels per level: grayLevels / Ns / the first-order histogram), 2 = global atomics,
// 3 = a private table per THREAD, [level][thread] (<= 48 edges: the usual 16 - 32 grey levels; neighbouring voxels of a smooth
// image share their level, so the lanes of a wave would hit one word of a per-wave table 64 times over).
template <typename T, bool LDS_EDGES, int COUNTS>
__global__ void __launch_bounds__(256) digitize_kernel(const T *__restrict__ x, const uint8_t *__restrict__ mask,
                                                       long long n, const double *__restrict__ edges, int nedges,
                                                       int *__restrict__ levels, int *__restrict__ maxlevel,
                                                       unsigned long long *__restrict__ counts) {
  extern __shared__ double se_raw[];
  constexpr int E = 16 / (int)sizeof(T);
  const double *se = LDS_EDGES ? se_raw : edges;
  const int nb = nedges + 1;
  unsigned int *cnt = reinterpret_cast<unsigned int *>(se_raw + (LDS_EDGES ? nedges : 0));   // [4][nb] (COUNTS == 1) / [nb][256] (3)
  if (LDS_EDGES)
    for (int i = threadIdx.x; i < nedges; i += blockDim.x) se_raw[i] = edges[i];
  if (COUNTS == 1)
    for (int i = threadIdx.x; i < 4 * nb; i += blockDim.x) cnt[i] = 0u;
  if (COUNTS == 3)
    for (int i = threadIdx.x; i < 256 * nb; i += blockDim.x) cnt[i] = 0u;
  if (LDS_EDGES || COUNTS == 1 || COUNTS == 3) __syncthreads();
  unsigned int *mine = COUNTS == 3 ? cnt + threadIdx.x : cnt + (threadIdx.x >> 6) * nb;
  // number of edges <= v (np.digitize): the edges are (nearly) equidistant, so start from the arithmetic guess and let
  // the comparisons against the real edge values decide -- the same answer as a bisection, in ~2 reads instead of 6
  const double e0 = se[0];
  const double elast = se[nedges - 1];
  const double inv = nedges > 1 && elast > e0 ? (double)(nedges - 1) / (elast - e0) : 0.0;
  int top = 0;
  auto level_of = [&](double v) -> int {
    int k = 0;
    if (v == v) {
      const double g = fmin(fmax((v - e0) * inv + 1.0, 0.0), (double)nedges);
      k = (int)g;
      while (k < nedges && se[k] <= v) k++;
      while (k > 0 && se[k - 1] > v) k--;
    }
    top = max(top, k);
    if (COUNTS == 1) atomicAdd(mine + k, 1u);
    if (COUNTS == 2) atomicAdd(counts + k, 1ull);
    if (COUNTS == 3) mine[k * 256] += 1u;          // (this thread's own word: bank = thread mod 32, no conflict)
    return k;
  };
  const bool aligned = (((uintptr_t)x) & 15) == 0 && (((uintptr_t)mask) & (E - 1)) == 0 && (((uintptr_t)levels) & 15) == 0;
  bin_scan(x, mask, n, aligned,
           [&](long long i, const T (&vals)[E], const uint8_t (&mk)[E]) {
             int lv[E];
#pragma unroll
             for (int e = 0; e < E; e++) lv[e] = mk[e] ? level_of((double)vals[e]) : 0;
             if (E == 2) {
               *reinterpret_cast<int2 *>(levels + i) = make_int2(lv[0], lv[1]);
             } else {
#pragma unroll
               for (int e = 0; e < E; e += 4)
                 *reinterpret_cast<int4 *>(levels + i + e) = make_int4(lv[e], lv[(e + 1) % E], lv[(e + 2) % E], lv[(e + 3) % E]);
             }
           },
           [&](long long i) { levels[i] = mask[i] ? level_of((double)x[i]) : 0; });
  for (int o = 32; o > 0; o >>= 1) top = max(top, __shfl_xor(top, o));
  __shared__ int stop[4];
  if ((threadIdx.x & 63) == 0) stop[threadIdx.x >> 6] = top;
  __syncthreads();
  if (threadIdx.x == 0) {
    top = max(max(stop[0], stop[1]), max(stop[2], stop[3]));
    if (top) atomicMax(maxlevel, top);
  }
  if (COUNTS == 3) {
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
      unsigned int v = 0;
      for (int j = 0; j < 256; j++) v += cnt[i * 256 + ((j + i) & 255)];     // (rotated: the threads read different banks)
      if (v) atomicAdd(counts + i, (unsigned long long)v);
    }
  }
  if (COUNTS == 1) {   // (a block sees fewer than 2^32 voxels: 32-bit partial sums)
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
      const unsigned int v = cnt[i] + cnt[nb + i] + cnt[2 * nb + i] + cnt[3 * nb + i];
      if (v) atomicAdd(counts + i, (unsigned long long)v);
    }
  }
}

// binCount edges on the device: np.histogram(x, bins=N)[1] is np.linspace(min, max, N + 1) --
// step = (max - min) / N, edge_i = i * step + min (two roundings, no fused multiply-add), edge_N = max -- and
// getBinEdges moves the last edge up by 1 (imageoperations.py:122-126).  info[0] = 1 when the ROI is constant / not finite /
// the step underflows (np.histogram and np.linspace then take other branches: the host route handles those).
// F32: a float32 image -- numpy then does all of it in float32 (np.histogram's bin_type, np.linspace's dt); integer images take the
// float64 arithmetic on their exact extremes.
template <bool F32>
__global__ void bincount_edges_kernel(const unsigned long long *__restrict__ keys, int N, double *__restrict__ edges,
                                      int *__restrict__ info) {
#pragma clang fp contract(off)
  const double lo = f64_unkey(keys[0]), hi = f64_unkey(keys[1]);
  bool bad = keys[1] == 0ull || !(hi > lo) || !isfinite(lo) || !isfinite(hi);
  if (F32) {
    const float lof = (float)lo, hif = (float)hi;
    const float delta = hif - lof, step = delta / (float)N;
    bad = bad || !isfinite(delta) || step == 0.0f;
    if (threadIdx.x == 0) info[0] = bad ? 1 : 0;
    for (int i = threadIdx.x; i <= N; i += blockDim.x) {
      float e = (float)i * step;
      e = e + lof;
      if (i == N) e = hif + 1.0f;
      edges[i] = bad ? (double)i : (double)e;
    }
  } else {
    const double delta = hi - lo, step = delta / (double)N;
    bad = bad || !isfinite(delta) || step == 0.0;
    if (threadIdx.x == 0) info[0] = bad ? 1 : 0;
    for (int i = threadIdx.x; i <= N; i += blockDim.x) {
      double e = (double)i * step;
      e = e + lo;
      if (i == N) e = hi + 1.0;
      edges[i] = bad ? (double)i : e;       // (ascending placeholders: the digitize pass after it stays well defined)
    }
  }
}

// ROI voxel count per level: per-wave private LDS tables when Ng fits (the common case), global atomics otherwise
__global__ void __launch_bounds__(256) level_counts_kernel(const int *__restrict__ levels,
                                                           const uint8_t *__restrict__ mask, long long n, int Ng,
                                                           int use_lds, unsigned long long *__restrict__ counts) {
  extern __shared__ unsigned int sh[];
  const int nb = Ng + 1;
  if (use_lds) {
    for (int i = threadIdx.x; i < 4 * nb; i += blockDim.x) sh[i] = 0u;
    __syncthreads();
  }
  unsigned int *mine = sh + (threadIdx.x >> 6) * nb;
  auto one = [&](int lv) {
    const int b = (lv >= 1 && lv <= Ng) ? lv : 0;
    if (use_lds) atomicAdd(mine + b, 1u);
    else atomicAdd(counts + b, 1ull);
  };
  const bool aligned = (((uintptr_t)levels) & 15) == 0 && (((uintptr_t)mask) & 3) == 0;
  bin_scan(levels, mask, n, aligned,
           [&](long long, const int (&vals)[4], const uint8_t (&mk)[4]) {
#pragma unroll
             for (int e = 0; e < 4; e++)
               if (mk[e]) one(vals[e]);
           },
           [&](long long i) {
             if (mask[i]) one(levels[i]);
           });
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
      const unsigned int v = sh[i] + sh[nb + i] + sh[2 * nb + i] + sh[3 * nb + i];
      if (v) atomicAdd(counts + i, (unsigned long long)v);
    }
  }
}

}  // namespace prad
