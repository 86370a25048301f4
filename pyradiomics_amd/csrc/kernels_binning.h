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

template <typename T>
__global__ void __launch_bounds__(256) roi_minmax_kernel(const T *__restrict__ x, const uint8_t *__restrict__ mask,
                                                         long long n, unsigned long long *__restrict__ keys) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned long long lo = ~0ull, hi = 0ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (!mask[i]) continue;
    const unsigned long long k = f64_key((double)x[i]);
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
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

template <typename T>
__global__ void __launch_bounds__(256) digitize_kernel(const T *__restrict__ x, const uint8_t *__restrict__ mask,
                                                       long long n, const double *__restrict__ edges, int nedges,
                                                       int *__restrict__ levels, int *__restrict__ maxlevel) {
  extern __shared__ double se[];
  for (int i = threadIdx.x; i < nedges; i += blockDim.x) se[i] = edges[i];
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  int top = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int lv = 0;
    if (mask[i]) {
      const double v = (double)x[i];
      int lo = 0, hi = nedges;             // first edge > v  ==  number of edges <= v
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (se[mid] <= v) lo = mid + 1;
        else hi = mid;
      }
      lv = lo;
      top = max(top, lv);
    }
    levels[i] = lv;
  }
  for (int o = 32; o > 0; o >>= 1) top = max(top, __shfl_xor(top, o));
  __shared__ int stop[4];
  if ((threadIdx.x & 63) == 0) stop[threadIdx.x >> 6] = top;
  __syncthreads();
  if (threadIdx.x == 0) {
    top = max(max(stop[0], stop[1]), max(stop[2], stop[3]));
    if (top) atomicMax(maxlevel, top);
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
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (!mask[i]) continue;
    const int lv = levels[i];
    const int b = (lv >= 1 && lv <= Ng) ? lv : 0;
    if (use_lds) atomicAdd(mine + b, 1u);
    else atomicAdd(counts + b, 1ull);
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
      const unsigned int v = sh[i] + sh[nb + i] + sh[2 * nb + i] + sh[3 * nb + i];
      if (v) atomicAdd(counts + i, (unsigned long long)v);
    }
  }
}

}  // namespace prad
