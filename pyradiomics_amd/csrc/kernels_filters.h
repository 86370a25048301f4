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
// Arithmetic of both filters lives in third-party wheels absent from the reference tree; pinned by the brain1 outputs the
// reference recorded in notebooks/helloFeatureClass.ipynb (tests/test_notebook_pin.py).
#pragma once
#include <type_traits>
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

// The same sums with the index arithmetic taken out of the element loop (the kernel above spends two 64-bit divisions and six
// modulo operations per output: 2.4 TB/s of its 24 B per element): the launch geometry carries the coordinates --
// INNER1 (contiguous axis): threadIdx / blockIdx.x run along the axis, blockIdx.y (+ z * 65535) over the outer index;
// otherwise: threadIdx / blockIdx.x run over the inner index, blockIdx.y along the axis, blockIdx.z over the outer index.
// Periodic wrap by one conditional add / subtract (the taps reach at most F < N positions).  Same operations, same order.
template <bool INNER1>
__global__ void __launch_bounds__(256) swt_axis2_kernel(const double *__restrict__ x, long long outer, int N,
                                                        long long inner, FilterTaps T, double *__restrict__ lo,
                                                        double *__restrict__ hi) {
  const int half = T.F / 2;
  int o;
  long long base, idx;
  if (INNER1) {
    o = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const long long r = (long long)blockIdx.y + (long long)blockIdx.z * 65535;
    if (o >= N || r >= outer) return;
    base = r * N;
    idx = base + o;
  } else {
    const long long in_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    o = (int)blockIdx.y;
    if (in_ >= inner) return;
    base = (long long)blockIdx.z * N * inner + in_;
    idx = base + (long long)o * inner;
  }
  double sl = 0.0, sh = 0.0;
  int p = o + half;
  if (p >= N) p -= N;
  for (int k = 0; k < T.F; k++) {
    const double v = x[base + (long long)p * inner];
    sl = __dadd_rn(sl, __dmul_rn(T.lo[k], v));
    sh = __dadd_rn(sh, __dmul_rn(T.hi[k], v));
    p = p == 0 ? N - 1 : p - 1;
  }
  lo[idx] = sl;
  hi[idx] = sh;
}

struct RGaussCoef {
  double N0, N1, N2, N3, D1, D2, D3, D4, M1, M2, M3, M4, BN1, BN2, BN3, BN4, BM1, BM2, BM3, BM4;
};

// final store of a pass: plain float image, or (last pass of a Laplacian term) acc = first ? v / sp2 : acc + v / sp2 with the
// roundings of the separate accumulation step (the term is rounded to float32 first, ITK's image type)
template <typename T>
__device__ __forceinline__ void rg_store(T *__restrict__ o, T *__restrict__ acc, long long idx, double v, double sp2,
                                         int first) {
  const T f = (T)v;
  if (acc) {
    const double a = first ? 0.0 : (double)acc[idx];
    acc[idx] = (T)(a + (double)f / sp2);
  } else {
    o[idx] = f;
  }
}

// data viewed as [outer][ln][inner]; lane = (outer index, inner index); scratch holds the causal pass in float64.
// A lane walks its line serially, so memory parallelism has to come from the lane itself: samples are fetched in
// batches of PRAD_RG_B independent loads ahead of the recursion that consumes them.
#define PRAD_RG_B 8
// ACCLOAD: the pass accumulates into an image that already holds earlier terms (acc != nullptr, first == 0); a template
// parameter because a load behind a run-time condition gets its own `s_waitcnt vmcnt(0)` (every step of the
// derivative passes then waited for its accumulator load alone: 205 us instead of 94)
template <bool ACCLOAD, typename T = float>
__global__ void __launch_bounds__(256) rgauss_line_kernel(const T *__restrict__ in, long long outer, int ln,
                                                          long long inner, RGaussCoef c,
                                                          double *__restrict__ scratch, T *__restrict__ out,
                                                          T *__restrict__ acc, double sp2, int first) {
#pragma clang fp contract(off)  // ITK's line arithmetic is plain multiply / add; keep the same roundings
  const long long lines = outer * inner;
  const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= lines) return;
  const long long base = (line / inner) * ln * inner + (line % inner);
  const T *d = in + base;
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
      T buf[PRAD_RG_B];
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
      T buf[PRAD_RG_B], ab[PRAD_RG_B];
      double sb[PRAD_RG_B];
#pragma unroll
      for (int k = 0; k < PRAD_RG_B; k++) {
        buf[k] = d[(long long)(i - 1 - k) * st];
        sb[k] = s[(long long)(i - 1 - k) * st];
        // (the accumulator too: a load right before its store, one per step, left the derivative passes at 205 us
        // against 94 us for the smoothing ones -- the compiler keeps it behind the previous step's store)
        ab[k] = ACCLOAD ? acc[base + (long long)(i - 1 - k) * st] : (T)0;
      }
#pragma unroll
      for (int k = 0; k < PRAD_RG_B; k++) {
        double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
        v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
        {
          const long long idx = base + (long long)(i - 1 - k) * st;
          const T f = (T)(sb[k] + v);
          if (acc) acc[idx] = (T)((double)ab[k] + (double)f / sp2);     // = rg_store
          else out[idx] = f;
        }
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

// The same pass without the float64 copy of the causal recursion (8 B written + 8 B read per sample: more than the images
// themselves).  out[i] = causal[i] + anticausal[i] needs one of the two directions kept, but not sample by sample: sweep 1
// runs the ANTI-causal recursion from the end of the line and keeps only its state (4 outputs) at every block boundary
// (32 B per RB samples); sweep 2 walks forward block by block -- causal recursion into registers, then the anti-causal
// one of the block RECOMPUTED downward from the saved state (same operations on the same operands: the same bits), sum,
// store.  Traffic per sample: 4 (sweep 1) + 4 (sweep 2) read, 4 written, 2 x 32 / RB of states: 16 B at RB = 16 instead
// of 36; 256^3: 94 -> see profiles (smoothing pass).  AM: 0 plain float output, 1 first Laplacian term (acc = v / sp2),
// 2 later term (acc += v / sp2).
#define PRAD_RG_RB 16
// Several sigmas per launch (blockIdx.y): a 256^3 volume has 65 536 lines = ONE wave per SIMD, and a pass is bound by the
// latency of its serial float64 chain -- the waves of the other sigmas fill the gaps (profiles/r03_probes.md, section 11).
#define PRAD_LOG_MAXSIG 8
struct RGMulti {
  RGaussCoef k[PRAD_LOG_MAXSIG];
  const float *in[PRAD_LOG_MAXSIG];
  double *scratch[PRAD_LOG_MAXSIG];
  float *out[PRAD_LOG_MAXSIG];
  float *acc[PRAD_LOG_MAXSIG];
};

template <int AM>
__device__ __forceinline__ void rgauss_line2_body(const float *__restrict__ in, long long outer, int ln,
                                                  long long inner, const RGaussCoef &c, double *__restrict__ states,
                                                  float *__restrict__ out, float *__restrict__ acc, double sp2) {
#pragma clang fp contract(off)  // ITK's line arithmetic is plain multiply / add; keep the same roundings
  constexpr int RB = PRAD_RG_RB;
  const long long lines = outer * inner;
  const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= lines) return;
  const long long base = (line / inner) * ln * inner + (line % inner);
  const float *d = in + base;
  const long long st = inner;
  // blocks [k RB, (k + 1) RB) for k < nb - 1; the last one takes the remainder, 4 .. RB + 3 samples (ln >= 4)
  const int nb = (ln - 4) / RB + 1;
  // state of boundary k (1 <= k < nb) = anticausal[k RB .. k RB + 3]: double index ((k - 1) * 4 + j) * lines + line
  double *sp = states + line;
  // ---- sweep 1: anti-causal, from the end, states only ----
  {
    const double v2 = d[(long long)(ln - 1) * st];
    const double y1 = d[(long long)(ln - 2) * st], y2 = d[(long long)(ln - 3) * st];
    double a1 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-1
    double a2 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-2
    double a3 = y1 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-3
    double a4 = y2 * c.M1 + y1 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-4
    a1 -= v2 * c.BM1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
    a2 -= a1 * c.D1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
    a3 -= a2 * c.D1 + a1 * c.D2 + v2 * c.BM3 + v2 * c.BM4;
    a4 -= a3 * c.D1 + a2 * c.D2 + a1 * c.D3 + v2 * c.BM4;
    double dp0 = d[(long long)(ln - 4) * st], dp1 = y2, dp2 = y1, dp3 = v2;  // data[i .. i+3] at i = ln-4
    double q0 = a4, q1 = a3, q2 = a2, q3 = a1;                              // anticausal[i .. i+3]
    int i = ln - 4;                // the next sample produced is i - 1
    // (i is a boundary when i % RB == 0 and i >= RB: then q0..q3 = anticausal[i .. i+3] is the state of boundary i / RB)
    while (i > 0) {
      if ((i % RB) == 0) {
        const long long k = i / RB - 1;
        sp[(k * 4 + 0) * lines] = q0;
        sp[(k * 4 + 1) * lines] = q1;
        sp[(k * 4 + 2) * lines] = q2;
        sp[(k * 4 + 3) * lines] = q3;
      }
      const int nstep = min(i, ((i - 1) % RB) + 1);    // down to the next boundary (or to 0)
      if (nstep == RB) {
        float buf[RB];
#pragma unroll
        for (int k = 0; k < RB; k++) buf[k] = d[(long long)(i - 1 - k) * st];
#pragma unroll
        for (int k = 0; k < RB; k++) {
          double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
          v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
          dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = buf[k];
          q3 = q2; q2 = q1; q1 = q0; q0 = v;
        }
      } else {
        for (int k = 0; k < nstep; k++) {
          double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
          v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
          dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = d[(long long)(i - 1 - k) * st];
          q3 = q2; q2 = q1; q1 = q0; q0 = v;
        }
      }
      i -= nstep;
    }
  }
  // ---- sweep 2: forward, block by block ----
  double p1 = 0, p2 = 0, p3 = 0, p4 = 0;        // causal[b-1 .. b-4]
  double dm1 = 0, dm2 = 0, dm3 = 0;             // data[b-1 .. b-3]
  for (int kb = 0; kb < nb; kb++) {
    const int b = kb * RB;
    const bool lastblk = kb == nb - 1;
    const int len = lastblk ? ln - b : RB;      // RB, or 4 .. RB + 3 for the last block
    constexpr int RM = RB + 4;                  // samples held: the block and, for the anti-causal start, the 4 behind it
    float dv[RM + 0];
    double cv[RM];
    // data of the block (+ the first 4 samples of the next block, which the anti-causal recursion starts from)
#pragma unroll
    for (int j = 0; j < RM; j++) {
      const int idx = b + j;
      dv[j] = idx < ln ? d[(long long)idx * st] : 0.f;
    }
    float ab[RM];
    if (AM == 2) {
#pragma unroll
      for (int j = 0; j < RM; j++) ab[j] = (j < len) ? acc[base + (long long)(b + j) * st] : 0.f;
    }
    // causal recursion over the block
    int j0 = 0;
    if (kb == 0) {
      const double v1 = dv[0];
      const double x1 = dv[1], x2 = dv[2], x3 = dv[3];
      double s0 = v1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
      double s1 = x1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
      double s2 = x2 * c.N0 + x1 * c.N1 + v1 * c.N2 + v1 * c.N3;
      double s3 = x3 * c.N0 + x2 * c.N1 + x1 * c.N2 + v1 * c.N3;
      s0 -= v1 * c.BN1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
      s1 -= s0 * c.D1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
      s2 -= s1 * c.D1 + s0 * c.D2 + v1 * c.BN3 + v1 * c.BN4;
      s3 -= s2 * c.D1 + s1 * c.D2 + s0 * c.D3 + v1 * c.BN4;
      cv[0] = s0; cv[1] = s1; cv[2] = s2; cv[3] = s3;
      dm1 = x3; dm2 = x2; dm3 = x1;
      p1 = s3; p2 = s2; p3 = s1; p4 = s0;
      j0 = 4;
    }
#pragma unroll
    for (int j = 0; j < RM; j++) {
      if (j >= j0 && j < len) {
        const double di = dv[j];
        double v = di * c.N0 + dm1 * c.N1 + dm2 * c.N2 + dm3 * c.N3;
        v -= p1 * c.D1 + p2 * c.D2 + p3 * c.D3 + p4 * c.D4;
        cv[j] = v;
        dm3 = dm2; dm2 = dm1; dm1 = di;
        p4 = p3; p3 = p2; p2 = p1; p1 = v;
      }
    }
    // anti-causal recursion of the block, downward, from the saved state (or from the end of the line)
    double q0, q1, q2, q3, e0, e1, e2, e3;      // anticausal[i .. i+3], data[i .. i+3] at i = b + len
    int jtop = len - 1;                         // block-local index of the first sample still to produce
    if (lastblk) {
      const double v2 = dv[len - 1], y1 = dv[len - 2], y2 = dv[len - 3];
      double a1 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
      double a2 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
      double a3 = y1 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
      double a4 = y2 * c.M1 + y1 * c.M2 + v2 * c.M3 + v2 * c.M4;
      a1 -= v2 * c.BM1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
      a2 -= a1 * c.D1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
      a3 -= a2 * c.D1 + a1 * c.D2 + v2 * c.BM3 + v2 * c.BM4;
      a4 -= a3 * c.D1 + a2 * c.D2 + a1 * c.D3 + v2 * c.BM4;
      // the four end samples are final here
#pragma unroll
      for (int j = 0; j < RM; j++) {
        if (j == len - 1) cv[j] += a1;
        if (j == len - 2) cv[j] += a2;
        if (j == len - 3) cv[j] += a3;
        if (j == len - 4) cv[j] += a4;
      }
      q0 = a4; q1 = a3; q2 = a2; q3 = a1;
      e0 = dv[len - 4]; e1 = y2; e2 = y1; e3 = v2;
      jtop = len - 5;
    } else {
      q0 = sp[((long long)kb * 4 + 0) * lines];
      q1 = sp[((long long)kb * 4 + 1) * lines];
      q2 = sp[((long long)kb * 4 + 2) * lines];
      q3 = sp[((long long)kb * 4 + 3) * lines];
      e0 = dv[RB]; e1 = dv[RB + 1]; e2 = dv[RB + 2]; e3 = dv[RB + 3];
    }
#pragma unroll
    for (int jj = RM - 1; jj >= 0; jj--) {
      if (jj <= jtop) {
        double v = e0 * c.M1 + e1 * c.M2 + e2 * c.M3 + e3 * c.M4;
        v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
        cv[jj] += v;
        e3 = e2; e2 = e1; e1 = e0; e0 = dv[jj];
        q3 = q2; q2 = q1; q1 = q0; q0 = v;
      }
    }
    // store
#pragma unroll
    for (int j = 0; j < RM; j++) {
      if (j < len) {
        const long long idx = base + (long long)(b + j) * st;
        const float f = (float)cv[j];
        if (AM == 0) out[idx] = f;
        else acc[idx] = (float)((AM == 2 ? (double)ab[j] : 0.0) + (double)f / sp2);     // = rg_store
      }
    }
  }
}

template <int AM>
__global__ void __launch_bounds__(256) rgauss_line2_kernel(RGMulti M, long long outer, int ln, long long inner, double sp2) {
  const int sg = blockIdx.y;
  rgauss_line2_body<AM>(M.in[sg], outer, ln, inner, M.k[sg], M.scratch[sg], M.out[sg], M.acc[sg], sp2);
}

// ---- whole lines in LDS (round 4) ---------------------------------------------------------------------------------
// Every pass above reads its line twice (the two sweeps) and parks float64 partials or block states in HBM: 16..28 B per
// sample against the 8 B a pass has to move (one read, one write).  Here a wave keeps TL whole lines in LDS: the tile is
// fetched once with full-width coalesced loads, both sweeps of the recursion read it from LDS (one lane per line), the
// result replaces the input sample in place, and the tile is written back with full-width stores -- 8 B per sample and
// pass (12 when the pass adds into the Laplacian).  The arithmetic per line is rgauss_line2_body's, operation for
// operation (anti-causal states every RB samples, forward sweep with the block's anti-causal part recomputed), so the
// outputs are the same bits.  One wave per workgroup; a 256-sample float line group of 32 lines is 33 KB + 7 KB of states:
// four waves per CU, one per SIMD.
//   T        image type between the passes: float (ITK's real type for integer and float32 inputs) or double (float64 inputs)
//   TL       lines per wave (a power of two <= 64; the lanes beyond TL only help to move the tile)
//   CONTIG   the filtered axis is the contiguous one: the tile is one contiguous piece of memory, element (line t, sample j)
//            sits at LDS element t * PT + j (PT odd: the lanes' reads spread over the banks); otherwise lines are TL
//            neighbouring inner positions, element j * TL + t
//   AM       0 plain output, 1 first Laplacian term (acc = v / sp2), 2 later term (acc += v / sp2)
template <typename T>
struct RGMultiT {
  RGaussCoef k[PRAD_LOG_MAXSIG];
  const T *in[PRAD_LOG_MAXSIG];
  T *out[PRAD_LOG_MAXSIG];
  T *acc[PRAD_LOG_MAXSIG];
};
template <typename T> struct RGVec;
template <> struct RGVec<float> { typedef float4 type; };
template <> struct RGVec<double> { typedef double2 type; };
#define PRAD_RGT_RB 16

template <typename T, int TL, bool CONTIG, int AM>
__global__ void __launch_bounds__(64) rgauss_tile_kernel(RGMultiT<T> M, long long outer, int ln, long long inner, double sp2,
                                                         int PT, int vec) {
#pragma clang fp contract(off)  // ITK's line arithmetic is plain multiply / add; keep the same roundings
  extern __shared__ __align__(16) unsigned char rg_smem[];
  constexpr int RB = PRAD_RGT_RB;
  constexpr int V = 16 / (int)sizeof(T);
  typedef typename RGVec<T>::type VT;
  const int lane = threadIdx.x;
  const int sg = blockIdx.y;
  const RGaussCoef &c = M.k[sg];
  const T *__restrict__ in = M.in[sg];
  T *tile = reinterpret_cast<T *>(rg_smem);
  const long long tile_elems = CONTIG ? (long long)TL * PT : (long long)ln * TL;
  double *states = reinterpret_cast<double *>(rg_smem + ((tile_elems * sizeof(T) + 15) & ~(size_t)15));
  long long gbase;
  int nl;
  if (CONTIG) {
    const long long l0 = (long long)blockIdx.x * TL;
    nl = (int)min((long long)TL, outer - l0);
    gbase = l0 * ln;
  } else {
    const long long nchunk = (inner + TL - 1) / TL;
    const long long o = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    nl = (int)min((long long)TL, inner - ch * TL);
    gbase = o * ln * inner + ch * TL;
  }
  // ---- tile in ----
  if (CONTIG) {
    if (vec) {
      const int nq = nl * ln / V;
#pragma unroll 8
      for (int q = lane; q < nq; q += 64) {
        const VT v = *reinterpret_cast<const VT *>(in + gbase + (long long)q * V);
        const int e = q * V, t = e / ln, j = e - t * ln;
        const T *pv = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int i = 0; i < V; i++) tile[t * PT + j + i] = pv[i];
      }
    } else {
      const int ne = nl * ln;
#pragma unroll 8
      for (int e = lane; e < ne; e += 64) {
        const int t = e / ln, j = e - t * ln;
        tile[t * PT + j] = in[gbase + e];
      }
    }
  } else {
    if (vec) {
      constexpr int QR = TL / V > 0 ? TL / V : 1;
      const int nq = ln * QR;
#pragma unroll 8
      for (int q = lane; q < nq; q += 64) {
        const int j = q / QR, t = (q % QR) * V;
        if (t < nl) *reinterpret_cast<VT *>(tile + j * TL + t) = *reinterpret_cast<const VT *>(in + gbase + (long long)j * inner + t);
      }
    } else {
      const int ne = ln * TL;
#pragma unroll 8
      for (int e = lane; e < ne; e += 64) {
        const int j = e / TL, t = e % TL;
        if (t < nl) tile[e] = in[gbase + (long long)j * inner + t];
      }
    }
  }
  __syncthreads();
#define PRAD_RGT_AT(j) tile[CONTIG ? lane * PT + (j) : (j) * TL + lane]
  if (lane < nl) {
    const int nb = (ln - 4) / RB + 1;      // blocks [k RB, (k + 1) RB) for k < nb - 1; the last takes the remainder (4 .. RB + 3)
    double *sp = states + lane;            // boundary k (1 <= k < nb), value i: sp[((k - 1) * 4 + i) * TL]
    // ---- sweep 1: anti-causal, from the end, states only ----
    {
      const double v2 = PRAD_RGT_AT(ln - 1);
      const double y1 = PRAD_RGT_AT(ln - 2), y2 = PRAD_RGT_AT(ln - 3);
      double a1 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-1
      double a2 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-2
      double a3 = y1 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-3
      double a4 = y2 * c.M1 + y1 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-4
      a1 -= v2 * c.BM1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
      a2 -= a1 * c.D1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
      a3 -= a2 * c.D1 + a1 * c.D2 + v2 * c.BM3 + v2 * c.BM4;
      a4 -= a3 * c.D1 + a2 * c.D2 + a1 * c.D3 + v2 * c.BM4;
      double dp0 = PRAD_RGT_AT(ln - 4), dp1 = y2, dp2 = y1, dp3 = v2;    // data[i .. i+3] at i = ln-4
      double q0 = a4, q1 = a3, q2 = a2, q3 = a1;                          // anticausal[i .. i+3]
      int i = ln - 4;                // the next sample produced is i - 1
      while (i > 0) {
        if ((i % RB) == 0) {
          const int k = i / RB - 1;
          sp[(k * 4 + 0) * TL] = q0;
          sp[(k * 4 + 1) * TL] = q1;
          sp[(k * 4 + 2) * TL] = q2;
          sp[(k * 4 + 3) * TL] = q3;
        }
        const int nstep = min(i, ((i - 1) % RB) + 1);    // down to the next boundary (or to 0)
        if (nstep == RB) {
          T buf[RB];
#pragma unroll
          for (int k = 0; k < RB; k++) buf[k] = PRAD_RGT_AT(i - 1 - k);
#pragma unroll
          for (int k = 0; k < RB; k++) {
            double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
            v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
            dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = buf[k];
            q3 = q2; q2 = q1; q1 = q0; q0 = v;
          }
        } else {
          for (int k = 0; k < nstep; k++) {
            double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
            v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
            dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = PRAD_RGT_AT(i - 1 - k);
            q3 = q2; q2 = q1; q1 = q0; q0 = v;
          }
        }
        i -= nstep;
      }
    }
    // ---- sweep 2: forward, block by block; the result replaces the input in the tile ----
    double p1 = 0, p2 = 0, p3 = 0, p4 = 0;        // causal[b-1 .. b-4]
    double dm1 = 0, dm2 = 0, dm3 = 0;             // data[b-1 .. b-3]
    for (int kb = 0; kb < nb; kb++) {
      const int b = kb * RB;
      const bool lastblk = kb == nb - 1;
      const int len = lastblk ? ln - b : RB;      // RB, or 4 .. RB + 3 for the last block
      constexpr int RM = RB + 4;                  // samples held: the block and, for the anti-causal start, the 4 behind it
      T dv[RM];
      double cv[RM];
#pragma unroll
      for (int j = 0; j < RM; j++) {
        const int idx = b + j;
        dv[j] = idx < ln ? PRAD_RGT_AT(idx) : (T)0;
      }
      int j0 = 0;
      if (kb == 0) {
        const double v1 = dv[0];
        const double x1 = dv[1], x2 = dv[2], x3 = dv[3];
        double s0 = v1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
        double s1 = x1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
        double s2 = x2 * c.N0 + x1 * c.N1 + v1 * c.N2 + v1 * c.N3;
        double s3 = x3 * c.N0 + x2 * c.N1 + x1 * c.N2 + v1 * c.N3;
        s0 -= v1 * c.BN1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
        s1 -= s0 * c.D1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
        s2 -= s1 * c.D1 + s0 * c.D2 + v1 * c.BN3 + v1 * c.BN4;
        s3 -= s2 * c.D1 + s1 * c.D2 + s0 * c.D3 + v1 * c.BN4;
        cv[0] = s0; cv[1] = s1; cv[2] = s2; cv[3] = s3;
        dm1 = x3; dm2 = x2; dm3 = x1;
        p1 = s3; p2 = s2; p3 = s1; p4 = s0;
        j0 = 4;
      }
#pragma unroll
      for (int j = 0; j < RM; j++) {
        if (j >= j0 && j < len) {
          const double di = dv[j];
          double v = di * c.N0 + dm1 * c.N1 + dm2 * c.N2 + dm3 * c.N3;
          v -= p1 * c.D1 + p2 * c.D2 + p3 * c.D3 + p4 * c.D4;
          cv[j] = v;
          dm3 = dm2; dm2 = dm1; dm1 = di;
          p4 = p3; p3 = p2; p2 = p1; p1 = v;
        }
      }
      double q0, q1, q2, q3, e0, e1, e2, e3;      // anticausal[i .. i+3], data[i .. i+3] at i = b + len
      int jtop = len - 1;                         // block-local index of the first sample still to produce
      if (lastblk) {
        double v2 = 0, y1 = 0, y2 = 0, y3 = 0;
#pragma unroll
        for (int j = 0; j < RM; j++) {            // (register array: compile-time indices only)
          if (j == len - 1) v2 = dv[j];
          if (j == len - 2) y1 = dv[j];
          if (j == len - 3) y2 = dv[j];
          if (j == len - 4) y3 = dv[j];
        }
        double a1 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
        double a2 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
        double a3 = y1 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;
        double a4 = y2 * c.M1 + y1 * c.M2 + v2 * c.M3 + v2 * c.M4;
        a1 -= v2 * c.BM1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
        a2 -= a1 * c.D1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
        a3 -= a2 * c.D1 + a1 * c.D2 + v2 * c.BM3 + v2 * c.BM4;
        a4 -= a3 * c.D1 + a2 * c.D2 + a1 * c.D3 + v2 * c.BM4;
#pragma unroll
        for (int j = 0; j < RM; j++) {
          if (j == len - 1) cv[j] += a1;
          if (j == len - 2) cv[j] += a2;
          if (j == len - 3) cv[j] += a3;
          if (j == len - 4) cv[j] += a4;
        }
        q0 = a4; q1 = a3; q2 = a2; q3 = a1;
        e0 = y3; e1 = y2; e2 = y1; e3 = v2;
        jtop = len - 5;
      } else {
        q0 = sp[(kb * 4 + 0) * TL];
        q1 = sp[(kb * 4 + 1) * TL];
        q2 = sp[(kb * 4 + 2) * TL];
        q3 = sp[(kb * 4 + 3) * TL];
        e0 = dv[RB]; e1 = dv[RB + 1]; e2 = dv[RB + 2]; e3 = dv[RB + 3];
      }
#pragma unroll
      for (int jj = RM - 1; jj >= 0; jj--) {
        if (jj <= jtop) {
          double v = e0 * c.M1 + e1 * c.M2 + e2 * c.M3 + e3 * c.M4;
          v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
          cv[jj] += v;
          e3 = e2; e2 = e1; e1 = e0; e0 = dv[jj];
          q3 = q2; q2 = q1; q1 = q0; q0 = v;
        }
      }
#pragma unroll
      for (int j = 0; j < RM; j++)
        if (j < len) PRAD_RGT_AT(b + j) = (T)cv[j];
    }
  }
#undef PRAD_RGT_AT
  __syncthreads();
  // ---- tile out ----
  T *__restrict__ dst = AM == 0 ? M.out[sg] : M.acc[sg];
  auto fin = [&](T f, T a) -> T {       // = rg_store
    if (AM == 0) return f;
    return (T)((AM == 2 ? (double)a : 0.0) + (double)f / sp2);
  };
  if (CONTIG) {
    if (vec) {
      const int nq = nl * ln / V;
#pragma unroll 4
      for (int q = lane; q < nq; q += 64) {
        const int e = q * V, t = e / ln, j = e - t * ln;
        VT a;
        if (AM == 2) a = *reinterpret_cast<const VT *>(dst + gbase + (long long)q * V);
        VT r;
        T *pr = reinterpret_cast<T *>(&r);
        const T *pa = reinterpret_cast<const T *>(&a);
#pragma unroll
        for (int i = 0; i < V; i++) pr[i] = fin(tile[t * PT + j + i], AM == 2 ? pa[i] : (T)0);
        *reinterpret_cast<VT *>(dst + gbase + (long long)q * V) = r;
      }
    } else {
      const int ne = nl * ln;
#pragma unroll 4
      for (int e = lane; e < ne; e += 64) {
        const int t = e / ln, j = e - t * ln;
        const T a = AM == 2 ? dst[gbase + e] : (T)0;
        dst[gbase + e] = fin(tile[t * PT + j], a);
      }
    }
  } else {
    if (vec) {
      constexpr int QR = TL / V > 0 ? TL / V : 1;
      const int nq = ln * QR;
#pragma unroll 4
      for (int q = lane; q < nq; q += 64) {
        const int j = q / QR, t = (q % QR) * V;
        if (t < nl) {
          const long long gi = gbase + (long long)j * inner + t;
          VT a;
          if (AM == 2) a = *reinterpret_cast<const VT *>(dst + gi);
          const VT f = *reinterpret_cast<const VT *>(tile + j * TL + t);
          VT r;
          T *pr = reinterpret_cast<T *>(&r);
          const T *pa = reinterpret_cast<const T *>(&a), *pf = reinterpret_cast<const T *>(&f);
#pragma unroll
          for (int i = 0; i < V; i++) pr[i] = fin(pf[i], AM == 2 ? pa[i] : (T)0);
          *reinterpret_cast<VT *>(dst + gi) = r;
        }
      }
    } else {
      const int ne = ln * TL;
#pragma unroll 4
      for (int e = lane; e < ne; e += 64) {
        const int j = e / TL, t = e % TL;
        if (t < nl) {
          const long long gi = gbase + (long long)j * inner + t;
          const T a = AM == 2 ? dst[gi] : (T)0;
          dst[gi] = fin(tile[e], a);
        }
      }
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

// The same pass with the line cut in two and both directions of the recursion in flight: a 256^3 volume has 65 536
// lines = 1 024 waves of 64 lines, one per SIMD of the GPU, each alternating load / recurse / store with nothing to
// hide the latencies behind (290 us per pass, against 125 us for the strided axes).  Here a workgroup is two waves on
// the same 64 lines.  Phase 1: the forward wave runs the causal recursion over [0, m), the backward wave the
// anti-causal one over [m, ln); each leaves its float64 partial in `scratch`.  Phase 2: each continues into the other
// half, adds its value to the partial stored there (causal + anti-causal, the same sum in either order) and stores the
// result.  Twice the waves, half the dependent chain, the same traffic; the next tile is fetched into registers while
// the current one is recursed.  Waves only share LDS with themselves (no workgroup barrier but the one between the
// phases).
// AM: 0 = plain float output, 1 = first Laplacian term (acc = v / sp2), 2 = later term (acc += v / sp2).  The phase
// (partial into scratch / final) and AM are compile-time in every loop: with run-time flags the tile loop was a maze
// of branches, each load group followed by its own wait.
// TL: lines per wave (64: every lane recurses; 32: half the lanes do, but a row piece is 32 samples = a full 128-byte
// line per access at the LDS cost of 16-sample tiles, and a 256^3 volume gets 4 096 waves instead of 2 048)
template <int W, int TL, int AM>
__device__ __forceinline__ void rgauss_xline2_body(const float *__restrict__ in, long long lines, int ln,
                                                   const RGaussCoef &c, double *scratch, float *__restrict__ out,
                                                   float *__restrict__ acc, double sp2) {
#pragma clang fp contract(off)
  constexpr int RPI = 64 / W;            // rows of W samples per wave instruction
  constexpr int NI = TL * W / 64;        // load / store instructions per tile
  __shared__ float tin_[2][TL][W + 1];
  __shared__ double tsc_[2][TL][W + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float(*tin)[W + 1] = tin_[wave];
  double(*tsc)[W + 1] = tsc_[wave];
  const long long l0 = (long long)blockIdx.x * TL;
  const int nl = (int)min((long long)TL, lines - l0);
  const bool mine = lane < nl;
  const int m = (((ln + W - 1) / W) / 2) * W;      // the split (4 <= m <= ln - 4: the caller guarantees ln >= 2 W)
  const int rr = lane / W, cc = lane % W;
  // global index of this lane's k-th element of the tile at column c0 (clamped: loads are never conditional)
  auto gidx = [&](int k, int c0, int w) __attribute__((always_inline)) -> long long {
    return (l0 + min(k * RPI + rr, nl - 1)) * ln + c0 + min(cc, w - 1);
  };
  // (macros, not lambdas taking the arrays by reference: those left all four arrays in scratch memory)
#define PRAD_XL_FETCH(PF, PS, C0, WW)                                                                     \
  {                                                                                                       \
    const int c0_ = (C0), w_ = (WW);                                                                      \
    _Pragma("unroll") for (int k = 0; k < NI; k++) PF[k] = in[gidx(k, c0_, w_)];                          \
    if (FIN) {                                                                                            \
      _Pragma("unroll") for (int k = 0; k < NI; k++)                                                      \
          PS[k] = __hip_atomic_load(scratch + gidx(k, c0_, w_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    }                                                                                                     \
  }
#define PRAD_XL_COMMIT(PF, PS)                                                                            \
  {                                                                                                       \
    _Pragma("unroll") for (int k = 0; k < NI; k++) {                                                      \
      tin[k * RPI + rr][cc] = PF[k];                                                                      \
      if (FIN) tsc[k * RPI + rr][cc] = PS[k];                                                             \
    }                                                                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                \
    __builtin_amdgcn_wave_barrier();                                                                      \
  }
  auto put = [&](auto fin_tag, int c0, int w) __attribute__((always_inline)) {     // tile results (tsc) -> scratch / the output image, coalesced
    constexpr bool FIN = decltype(fin_tag)::value;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float ab[NI];
    if (FIN && AM == 2) {              // all accumulator loads first
#pragma unroll
      for (int k = 0; k < NI; k++) ab[k] = acc[gidx(k, c0, w)];
    }
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const int tt = k * RPI + rr;
      if (tt < nl && cc < w) {
        const long long idx = (l0 + tt) * ln + c0 + cc;
        if (FIN) {
          const float f = (float)tsc[tt][cc];
          if (AM == 0) out[idx] = f;
          else acc[idx] = (float)((AM == 2 ? (double)ab[k] : 0.0) + (double)f / sp2);   // = rg_store
        } else {
          scratch[idx] = tsc[tt][cc];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  double t1 = 0, t2 = 0, t3 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0;   // recursion state: 3 (4) data samples, 4 outputs
  auto forward = [&](auto fin_tag, int r0, int r1) __attribute__((always_inline)) {
    constexpr bool FIN = decltype(fin_tag)::value;
    float pfA[NI], pfB[NI];      // two tiles in flight ahead of the recursion (a 256^3 volume only has 2 048 such waves:
    double psA[NI], psB[NI];     // one tile each left ~4 MB in flight on the whole GPU, 178 us per pass)
    PRAD_XL_FETCH(pfA, psA, r0, min(W, r1 - r0));
    if (r0 + W < r1) PRAD_XL_FETCH(pfB, psB, r0 + W, min(W, r1 - r0 - W));
    int par = 0;
    for (int c0 = r0; c0 < r1; c0 += W, par ^= 1) {
      const int w = min(W, r1 - c0);
      if (par == 0) {
        PRAD_XL_COMMIT(pfA, psA);
        if (c0 + 2 * W < r1) PRAD_XL_FETCH(pfA, psA, c0 + 2 * W, min(W, r1 - c0 - 2 * W));
      } else {
        PRAD_XL_COMMIT(pfB, psB);
        if (c0 + 2 * W < r1) PRAD_XL_FETCH(pfB, psB, c0 + 2 * W, min(W, r1 - c0 - 2 * W));
      }
#ifndef PRAD_DBG_XL_NOCOMPUTE
      if (mine) {
        int j = 0;
        if (c0 == 0) {
          const double v1 = tin[lane][0];
          const double x1 = tin[lane][1], x2 = tin[lane][2], x3 = tin[lane][3];
          double s0 = v1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
          double s1 = x1 * c.N0 + v1 * c.N1 + v1 * c.N2 + v1 * c.N3;
          double s2 = x2 * c.N0 + x1 * c.N1 + v1 * c.N2 + v1 * c.N3;
          double s3 = x3 * c.N0 + x2 * c.N1 + x1 * c.N2 + v1 * c.N3;
          s0 -= v1 * c.BN1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
          s1 -= s0 * c.D1 + v1 * c.BN2 + v1 * c.BN3 + v1 * c.BN4;
          s2 -= s1 * c.D1 + s0 * c.D2 + v1 * c.BN3 + v1 * c.BN4;
          s3 -= s2 * c.D1 + s1 * c.D2 + s0 * c.D3 + v1 * c.BN4;
          tsc[lane][0] = s0; tsc[lane][1] = s1; tsc[lane][2] = s2; tsc[lane][3] = s3;   // (c0 == 0 is never final)
          t1 = x3; t2 = x2; t3 = x1;
          u1 = s3; u2 = s2; u3 = s1; u4 = s0;
          j = 4;
        }
        for (; j < w; j++) {
          const double di = tin[lane][j];
          double v = di * c.N0 + t1 * c.N1 + t2 * c.N2 + t3 * c.N3;
          v -= u1 * c.D1 + u2 * c.D2 + u3 * c.D3 + u4 * c.D4;
          tsc[lane][j] = FIN ? v + tsc[lane][j] : v;
          t3 = t2; t2 = t1; t1 = di;
          u4 = u3; u3 = u2; u2 = u1; u1 = v;
        }
      }
#endif
      put(fin_tag, c0, w);
    }
  };
  double t0 = 0;   // (backward: t0..t3 = data[i], [i+1], [i+2], [i+3]; u1..u4 = outputs [i], [i+1], [i+2], [i+3])
  auto backward = [&](auto fin_tag, int r0, int r1) __attribute__((always_inline)) {
    constexpr bool FIN = decltype(fin_tag)::value;
    float pfA[NI], pfB[NI];      // two tiles in flight ahead of the recursion (a 256^3 volume only has 2 048 such waves:
    double psA[NI], psB[NI];     // one tile each left ~4 MB in flight on the whole GPU, 178 us per pass)
    auto tile_b = [&](int e) { return max(e - W, r0); };       // tile [tile_b(e), e)
    PRAD_XL_FETCH(pfA, psA, tile_b(r1), r1 - tile_b(r1));
    if (tile_b(r1) > r0) PRAD_XL_FETCH(pfB, psB, tile_b(r1 - W), r1 - W - tile_b(r1 - W));
    int par = 0;
    for (int e = r1; e > r0; e -= W, par ^= 1) {
      const int b = tile_b(e), w = e - b;
      const int e2 = e - 2 * W;                                 // end of the tile after next
      if (par == 0) {
        PRAD_XL_COMMIT(pfA, psA);
        if (e2 > r0) PRAD_XL_FETCH(pfA, psA, tile_b(e2), e2 - tile_b(e2));
      } else {
        PRAD_XL_COMMIT(pfB, psB);
        if (e2 > r0) PRAD_XL_FETCH(pfB, psB, tile_b(e2), e2 - tile_b(e2));
      }
#ifndef PRAD_DBG_XL_NOCOMPUTE
      if (mine) {
        int j = w - 1;
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
          tsc[lane][w - 1] = a1; tsc[lane][w - 2] = a2; tsc[lane][w - 3] = a3; tsc[lane][w - 4] = a4;   // (never final)
          t0 = tin[lane][w - 4]; t1 = y2; t2 = y1; t3 = v2;
          u1 = a4; u2 = a3; u3 = a2; u4 = a1;
          j = w - 5;
        }
        for (; j >= 0; j--) {
          double v = t0 * c.M1 + t1 * c.M2 + t2 * c.M3 + t3 * c.M4;
          v -= u1 * c.D1 + u2 * c.D2 + u3 * c.D3 + u4 * c.D4;
          t3 = t2; t2 = t1; t1 = t0; t0 = tin[lane][j];
          tsc[lane][j] = FIN ? tsc[lane][j] + v : v;
          u4 = u3; u3 = u2; u2 = u1; u1 = v;
        }
      }
#endif
      put(fin_tag, b, w);
    }
  };
  using No = std::integral_constant<bool, false>;
  using Yes = std::integral_constant<bool, true>;
  if (wave == 0) forward(No{}, 0, m);
  else backward(No{}, m, ln);
  __threadfence();
  __syncthreads();
  if (wave == 0) forward(Yes{}, m, ln);
  else backward(Yes{}, 0, m);
#undef PRAD_XL_FETCH
#undef PRAD_XL_COMMIT
}

template <int W, int TL, int AM>
__global__ void __launch_bounds__(128) rgauss_xline2_kernel(RGMulti M, long long lines, int ln, double sp2) {
  const int sg = blockIdx.y;
  rgauss_xline2_body<W, TL, AM>(M.in[sg], lines, ln, M.k[sg], M.scratch[sg], M.out[sg], M.acc[sg], sp2);
}

}  // namespace prad
