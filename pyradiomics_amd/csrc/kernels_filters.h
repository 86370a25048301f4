// kernels_filters.h -- the filter stack in front of the texture matrices,
// HBM-bound separable 1-D passes.
//
//   swt_axis_kernel      one level-1 undecimated, periodised analysis step along one axis:
//                        lo[o] = sum_k dec_lo[k] * x[(o + F/2 - k) mod N],  hi likewise   (float64, taps in
//                        ascending k, unfused multiply/add so that rounding follows the CPU restatement of
//                        PyWavelets' downsampling_convolution_periodization; imageoperations.py:921-935)
//   rgauss_line_kernel   ITK RecursiveGaussianImageFilter along one axis: one lane per line, 4th-order causal +
//                        anti-causal recursion with edge-replicating boundary initialisation, float64 inside the
//                        line, float32 images between passes (imageoperations.py:824-830)
//   log_accumulate_kernel  Laplacian accumulation  acc += d2 / spacing^2
// Arithmetic of both filters lives in third-party wheels absent from the reference tree: parity unpinned.
#pragma once
#include "prad_runtime.h"

namespace prad {

#define PRAD_MAX_TAPS 32
struct FilterTaps {
  int F;
  double lo[PRAD_MAX_TAPS];
  double hi[PRAD_MAX_TAPS];
};

// x viewed as [outer][N][inner] (inner = stride of the filtered axis); one lane per element
__global__ void __launch_bounds__(256) swt_axis_kernel(const double *__restrict__ x, long long outer, int N,
                                                       long long inner, FilterTaps T, double *__restrict__ lo,
                                                       double *__restrict__ hi) {
  const long long total = outer * N * inner;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int half = T.F / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long in_ = i % inner;
    const long long r = i / inner;
    const int o = (int)(r % N);
    const long long base = (r / N) * N * inner + in_;
    double sl = 0.0, sh = 0.0;
    for (int k = 0; k < T.F; k++) {
      int p = o + half - k;
      p %= N;
      if (p < 0) p += N;
      const double v = x[base + (long long)p * inner];
      sl = __dadd_rn(sl, __dmul_rn(T.lo[k], v));
      sh = __dadd_rn(sh, __dmul_rn(T.hi[k], v));
    }
    lo[i] = sl;
    hi[i] = sh;
  }
}

struct RGaussCoef {
  double N0, N1, N2, N3, D1, D2, D3, D4, M1, M2, M3, M4, BN1, BN2, BN3, BN4, BM1, BM2, BM3, BM4;
};

// final store of a pass: plain float image, or (last pass of a Laplacian term) acc = first ? v / sp2 : acc + v / sp2 with the
// roundings of the separate accumulation step (the term is rounded to float32 first, ITK's image type)
__device__ __forceinline__ void rg_store(float *__restrict__ o, float *__restrict__ acc, long long idx, double v, double sp2,
                                         int first) {
  const float f = (float)v;
  if (acc) {
    const double a = first ? 0.0 : (double)acc[idx];
    acc[idx] = (float)(a + (double)f / sp2);
  } else {
    o[idx] = f;
  }
}

// data viewed as [outer][ln][inner]; lane = (outer index, inner index); scratch holds the causal pass in float64.
// A lane walks its line serially, so memory parallelism has to come from the lane itself: samples are fetched in
// batches of PRAD_RG_B independent loads ahead of the recursion that consumes them.
#define PRAD_RG_B 8
__global__ void __launch_bounds__(256) rgauss_line_kernel(const float *__restrict__ in, long long outer, int ln,
                                                          long long inner, RGaussCoef c,
                                                          double *__restrict__ scratch, float *__restrict__ out,
                                                          float *__restrict__ acc, double sp2, int first) {
#pragma clang fp contract(off)  // ITK's line arithmetic is plain multiply / add; keep the same roundings
  const long long lines = outer * inner;
  const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= lines) return;
  const long long base = (line / inner) * ln * inner + (line % inner);
  const float *d = in + base;
  double *s = scratch + base;
  const long long st = inner;
  // causal pass (itkRecursiveSeparableImageFilter.hxx FilterDataArray)
  const double v1 = d[0];
  double x1 = d[1 * st], x2 = d[2 * st], x3 = d[3 * st];
  double s0 = v1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
  double s1 = x1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
  double s2 = x2 * c.N0 + x1 * c.N1 + v1 * c.N2 + v1 * c.N3;
  double s3 = x3 * c.N0 + x2 * c.N1 + x1 * c.N2 + v1 * c.N3;
  s0 -= v1 * c.BN1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
  s1 -= s0 * c.D1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
  s2 -= s1 * c.D1 + s0 * c.D2 + v1 * c.BN3 + v1 * c.BN4;
  s3 -= s2 * c.D1 + s1 * c.D2 + s0 * c.D3 + v1 * c.BN4;
  s[0] = s0; s[st] = s1; s[2 * st] = s2; s[3 * st] = s3;
  {
    double dm1 = x3, dm2 = x2, dm3 = x1;       // data[i-1], [i-2], [i-3]
    double p1 = s3, p2 = s2, p3 = s1, p4 = s0; // scratch[i-1..i-4]
    int i = 4;
    for (; i + PRAD_RG_B <= ln; i += PRAD_RG_B) {
      float buf[PRAD_RG_B];
#pragma unroll
      for (int k = 0; k < PRAD_RG_B; k++) buf[k] = d[(long long)(i + k) * st];
#pragma unroll
      for (int k = 0; k < PRAD_RG_B; k++) {
        const double di = buf[k];
        double v = di * c.N0 + dm1 * c.N1 + dm2 * c.N2 + dm3 * c.N3;
        v -= p1 * c.D1 + p2 * c.D2 + p3 * c.D3 + p4 * c.D4;
        s[(long long)(i + k) * st] = v;
        dm3 = dm2; dm2 = dm1; dm1 = di;
        p4 = p3; p3 = p2; p2 = p1; p1 = v;
      }
    }
    for (; i < ln; i++) {
      const double di = d[(long long)i * st];
      double v = di * c.N0 + dm1 * c.N1 + dm2 * c.N2 + dm3 * c.N3;
      v -= p1 * c.D1 + p2 * c.D2 + p3 * c.D3 + p4 * c.D4;
      s[(long long)i * st] = v;
      dm3 = dm2; dm2 = dm1; dm1 = di;
      p4 = p3; p3 = p2; p2 = p1; p1 = v;
    }
  }
  // anti-causal pass
  const double v2 = d[(long long)(ln - 1) * st];
  const double y1 = d[(long long)(ln - 2) * st], y2 = d[(long long)(ln - 3) * st];
  double a1 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-1
  double a2 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-2: data[ln-1] = v2
  double a3 = y1 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-3
  double a4 = y2 * c.M1 + y1 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-4
  a1 -= v2 * c.BM1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
  a2 -= a1 * c.D1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
  a3 -= a2 * c.D1 + a1 * c.D2 + v2 * c.BM3 + v2 * c.BM4;
  a4 -= a3 * c.D1 + a2 * c.D2 + a1 * c.D3 + v2 * c.BM4;
  rg_store(out, acc, base + (long long)(ln - 1) * st, s[(long long)(ln - 1) * st] + a1, sp2, first);
  rg_store(out, acc, base + (long long)(ln - 2) * st, s[(long long)(ln - 2) * st] + a2, sp2, first);
  rg_store(out, acc, base + (long long)(ln - 3) * st, s[(long long)(ln - 3) * st] + a3, sp2, first);
  rg_store(out, acc, base + (long long)(ln - 4) * st, s[(long long)(ln - 4) * st] + a4, sp2, first);
  {
    // scratch[i-1] = data[i]*M1 + data[i+1]*M2 + data[i+2]*M3 + data[i+3]*M4 - (scratch[i]*D1 + ... + scratch[i+3]*D4)
    double dp0 = d[(long long)(ln - 4) * st], dp1 = y2, dp2 = y1, dp3 = v2;  // data[i], [i+1], [i+2], [i+3] at i = ln-4
    double q0 = a4, q1 = a3, q2 = a2, q3 = a1;                              // scratch[i], [i+1], [i+2], [i+3]
    int i = ln - 4;
    for (; i - PRAD_RG_B >= 0; i -= PRAD_RG_B) {      // produces samples i-1 .. i-B
      float buf[PRAD_RG_B];
      double sb[PRAD_RG_B];
#pragma unroll
      for (int k = 0; k < PRAD_RG_B; k++) {
        buf[k] = d[(long long)(i - 1 - k) * st];
        sb[k] = s[(long long)(i - 1 - k) * st];
      }
#pragma unroll
      for (int k = 0; k < PRAD_RG_B; k++) {
        double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
        v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
        rg_store(out, acc, base + (long long)(i - 1 - k) * st, sb[k] + v, sp2, first);
        dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = buf[k];
        q3 = q2; q2 = q1; q1 = q0; q0 = v;
      }
    }
    for (; i > 0; i--) {
      double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
      v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
      rg_store(out, acc, base + (long long)(i - 1) * st, s[(long long)(i - 1) * st] + v, sp2, first);
      dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = d[(long long)(i - 1) * st];
      q3 = q2; q2 = q1; q1 = q0; q0 = v;
    }
  }
}

// The same recursion for the contiguous axis (inner == 1): a lane-per-line walk would read 64 different cache lines
// per step.  One wave owns 64 consecutive lines and moves them through LDS in 64-sample tiles: every global access is
// a 256-byte row segment, the lane then walks its own line inside the tile (pitch 65: conflict-free).  Causal tiles
// run from the start of the line, anti-causal tiles from its end, so each direction's 4 boundary samples sit in
// its first tile.
#define PRAD_RG_T 64     // lines per workgroup (one lane each)
#define PRAD_RG_W 32     // samples per tile: 25 KB of LDS per wave instead of 50, twice the waves per CU
__global__ void __launch_bounds__(64) rgauss_xline_kernel(const float *__restrict__ in, long long lines, int ln,
                                                          RGaussCoef c, double *__restrict__ scratch,
                                                          float *__restrict__ out, float *__restrict__ acc, double sp2,
                                                          int first) {
#pragma clang fp contract(off)
  __shared__ float tin[PRAD_RG_T][PRAD_RG_W + 1];
  __shared__ double tsc[PRAD_RG_T][PRAD_RG_W + 1];
  const int lane = threadIdx.x;
  const long long l0 = (long long)blockIdx.x * PRAD_RG_T;
  const int nl = (int)min((long long)PRAD_RG_T, lines - l0);
  const bool mine = lane < nl;
  // ---- causal ----
  double dm1 = 0, dm2 = 0, dm3 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, v1 = 0;
  for (int c0 = 0; c0 < ln; c0 += PRAD_RG_W) {
    const int w = min(PRAD_RG_W, ln - c0);
    __syncthreads();
    for (int t = 0; t < nl; t += 2) {      // two rows of 32 samples per wave instruction
      const int tt = t + (lane >> 5), cc = lane & 31;
      if (tt < nl && cc < w) tin[tt][cc] = in[(l0 + tt) * ln + c0 + cc];
    }
    __syncthreads();
    if (mine) {
      int j = 0;
      if (c0 == 0) {
        v1 = tin[lane][0];
        const double x1 = tin[lane][1], x2 = tin[lane][2], x3 = tin[lane][3];
        double s0 = v1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
        double s1 = x1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
        double s2 = x2 * c.N0 + x1 * c.N1 + v1 * c.N2 + v1 * c.N3;
        double s3 = x3 * c.N0 + x2 * c.N1 + x1 * c.N2 + v1 * c.N3;
        s0 -= v1 * c.BN1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
        s1 -= s0 * c.D1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
        s2 -= s1 * c.D1 + s0 * c.D2 + v1 * c.BN3 + v1 * c.BN4;
        s3 -= s2 * c.D1 + s1 * c.D2 + s0 * c.D3 + v1 * c.BN4;
        tsc[lane][0] = s0; tsc[lane][1] = s1; tsc[lane][2] = s2; tsc[lane][3] = s3;
        dm1 = x3; dm2 = x2; dm3 = x1;
        p1 = s3; p2 = s2; p3 = s1; p4 = s0;
        j = 4;
      }
      for (; j < w; j++) {
        const double di = tin[lane][j];
        double v = di * c.N0 + dm1 * c.N1 + dm2 * c.N2 + dm3 * c.N3;
        v -= p1 * c.D1 + p2 * c.D2 + p3 * c.D3 + p4 * c.D4;
        tsc[lane][j] = v;
        dm3 = dm2; dm2 = dm1; dm1 = di;
        p4 = p3; p3 = p2; p2 = p1; p1 = v;
      }
    }
    __syncthreads();
    for (int t = 0; t < nl; t += 2) {
      const int tt = t + (lane >> 5), cc = lane & 31;
      if (tt < nl && cc < w) scratch[(l0 + tt) * ln + c0 + cc] = tsc[tt][cc];
    }
  }
  // ---- anti-causal: tiles [e - w, e) walking down from e = ln ----
  double dp0 = 0, dp1 = 0, dp2 = 0, dp3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  for (int e = ln; e > 0; e -= PRAD_RG_W) {
    const int b = max(e - PRAD_RG_W, 0), w = e - b;
    __syncthreads();
    for (int t = 0; t < nl; t += 2) {
      const int tt = t + (lane >> 5), cc = lane & 31;
      if (tt < nl && cc < w) {
        tin[tt][cc] = in[(l0 + tt) * ln + b + cc];
        tsc[tt][cc] = scratch[(l0 + tt) * ln + b + cc];
      }
    }
    __syncthreads();
    if (mine) {
      int j = w - 1;                         // tile-local index of the sample being produced
      if (e == ln) {
        const double v2 = tin[lane][w - 1], y1 = tin[lane][w - 2], y2 = tin[lane][w - 3];
        double a1 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
        double a2 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
        double a3 = y1 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
        double a4 = y2 * c.M1 + y1 * c.M2 + v2 * c.M3 + v2 * c.M4;
        a1 -= v2 * c.BM1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
        a2 -= a1 * c.D1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
        a3 -= a2 * c.D1 + a1 * c.D2 + v2 * c.BM3 + v2 * c.BM4;
        a4 -= a3 * c.D1 + a2 * c.D2 + a1 * c.D3 + v2 * c.BM4;
        tsc[lane][w - 1] += a1; tsc[lane][w - 2] += a2; tsc[lane][w - 3] += a3; tsc[lane][w - 4] += a4;
        dp0 = tin[lane][w - 4]; dp1 = y2; dp2 = y1; dp3 = v2;
        q0 = a4; q1 = a3; q2 = a2; q3 = a1;
        j = w - 5;
      }
      for (; j >= 0; j--) {
        double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
        v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
        dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = tin[lane][j];
        tsc[lane][j] += v;
        q3 = q2; q2 = q1; q1 = q0; q0 = v;
      }
    }
    __syncthreads();
    for (int t = 0; t < nl; t += 2) {
      const int tt = t + (lane >> 5), cc = lane & 31;
      if (tt < nl && cc < w) rg_store(out, acc, (l0 + tt) * ln + b + cc, tsc[tt][cc], sp2, first);
    }
  }
}

}  // namespace prad
