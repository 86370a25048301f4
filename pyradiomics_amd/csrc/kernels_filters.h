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

// The three axis passes of a 3-D level-1 transform in ONE kernel (round 4).  The separate passes move 168 B per voxel (x: 8 in,
// 16 out; y: 16 in, 32 out; z: 32 in, 64 out); the eight sub-bands themselves are 64 B.  Here a workgroup of 256 lanes owns
// a column of 8 x 32 (y, x) outputs and streams along z: each input plane of the column (with its F - 1 halo rows and columns,
// periodic) is staged in LDS, the x pass leaves lo / hi rows in LDS, every lane runs the y pass for its own (y, x) and keeps
// the four y-bands of the last F planes in REGISTERS, from which the z pass produces the eight outputs of a finished
// plane.  72 B per voxel reach HBM.  Every 1-D sum is the one of swt_axis_kernel -- taps in ascending k from 0.0, unfused
// multiply / add, float64 intermediates -- so the sub-bands are the same bits (tests/test_gpu_filters.py).
// Output order as swt_level1_dev builds it for axes (2, 1, 0): band index = (bx * 2 + by) * 2 + bz.
#define PRAD_SWT3_TY 8
#define PRAD_SWT3_TX 32
// TIN: the image as it is (int16 / int32 / float32 / float64): converted to float64 when staged (exact).  pywt.swtn widens
// integer images to float64 the same way; FLOAT32 images it transforms in float32 and returns float32 sub-bands, where this
// kernel computes and returns float64 (more precision than the reference has: stated deviation).
template <int F, typename TIN>
__global__ void __launch_bounds__(256) swt3_fused_kernel(const TIN *__restrict__ x, int Nz, int Ny, int Nx, FilterTaps T,
                                                         int CZ, double *__restrict__ out) {
  constexpr int TY = PRAD_SWT3_TY, TX = PRAD_SWT3_TX;
  constexpr int HY = TY + F - 1, HX = TX + F - 1, LOH = F - 1 - F / 2;
  __shared__ double S0[HY][HX + 1];
  __shared__ double AD[2][HY][TX + 1];
  const int tid = threadIdx.x, tx = tid & (TX - 1), ty = tid / TX;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, zc0 = blockIdx.z * CZ, zc1 = min(Nz, zc0 + CZ);
  const long long n = (long long)Nz * Ny * Nx;
  const bool mine = (y0 + ty) < Ny && (x0 + tx) < Nx;
  double r[4][F];                      // the y-bands (aa, ad, da, dd) of the last F planes at this lane's (y, x); [0] = newest
#pragma unroll
  for (int b = 0; b < 4; b++)
#pragma unroll
    for (int k = 0; k < F; k++) r[b][k] = 0.0;
  const int steps = (zc1 - zc0) + F - 1;
  for (int s = 0; s < steps; s++) {
    int zp = (zc0 - LOH + s) % Nz;
    if (zp < 0) zp += Nz;
    const TIN *plane = x + (long long)zp * Ny * Nx;
    for (int e = tid; e < HY * HX; e += 256) {
      const int iy = e / HX, ix = e - iy * HX;
      int gy = y0 - LOH + iy, gx = x0 - LOH + ix;
      gy = gy < 0 ? gy + Ny : (gy >= Ny ? gy - Ny : gy);
      gx = gx < 0 ? gx + Nx : (gx >= Nx ? gx - Nx : gx);
      gy = gy >= Ny ? gy % Ny : gy;       // (ragged last tiles reach further than one period; their outputs are not stored)
      gx = gx >= Nx ? gx % Nx : gx;
      S0[iy][ix] = (double)plane[(long long)gy * Nx + gx];
    }
    __syncthreads();
    for (int e = tid; e < HY * TX; e += 256) {
      const int iy = e / TX, ox = e - iy * TX;
      double sl = 0.0, sh = 0.0;
#pragma unroll
      for (int k = 0; k < F; k++) {
        const double v = S0[iy][ox + F - 1 - k];
        sl = __dadd_rn(sl, __dmul_rn(T.lo[k], v));
        sh = __dadd_rn(sh, __dmul_rn(T.hi[k], v));
      }
      AD[0][iy][ox] = sl;
      AD[1][iy][ox] = sh;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int k = F - 1; k >= 1; k--) r[b][k] = r[b][k - 1];
    {
      double aa = 0.0, ad = 0.0, da = 0.0, dd = 0.0;
#pragma unroll
      for (int k = 0; k < F; k++) {
        const double va = AD[0][ty + F - 1 - k][tx], vd = AD[1][ty + F - 1 - k][tx];
        aa = __dadd_rn(aa, __dmul_rn(T.lo[k], va));
        ad = __dadd_rn(ad, __dmul_rn(T.hi[k], va));
        da = __dadd_rn(da, __dmul_rn(T.lo[k], vd));
        dd = __dadd_rn(dd, __dmul_rn(T.hi[k], vd));
      }
      r[0][0] = aa; r[1][0] = ad; r[2][0] = da; r[3][0] = dd;
    }
    if (s >= F - 1 && mine) {
      const int oz = zc0 + s - (F - 1);
      const long long vi = ((long long)oz * Ny + (y0 + ty)) * Nx + (x0 + tx);
#pragma unroll
      for (int b = 0; b < 4; b++) {
        double sl = 0.0, sh = 0.0;
#pragma unroll
        for (int k = 0; k < F; k++) {
          sl = __dadd_rn(sl, __dmul_rn(T.lo[k], r[b][k]));
          sh = __dadd_rn(sh, __dmul_rn(T.hi[k], r[b][k]));
        }
        out[(long long)(2 * b) * n + vi] = sl;
        out[(long long)(2 * b + 1) * n + vi] = sh;
      }
    }
  }
}

struct RGaussCoef {
  double N0, N1, N2, N3, D1, D2, D3, D4, M1, M2, M3, M4, BN1, BN2, BN3, BN4, BM1, BM2, BM3, BM4;
};

// Several sigmas per launch (blockIdx.y): a 256^3 volume has 65 536 lines = ONE wave per SIMD; the waves of the other sigmas
// fill the machine (profiles/r03_probes.md section 11, profiles/r04_probes.md).
#define PRAD_LOG_MAXSIG 8
// T = element type the pass READS, TO = the type it WRITES.  ITK's images between the passes are float whatever the input
// (itkLaplacianRecursiveGaussianImageFilter.h: InternalRealType = float); the first pass of a term -- the derivative filter,
// RecursiveGaussianImageFilter<InputImage, Image<float>> -- reads the input image in its own type: <double, float> for a
// float64 input, <float, float> everywhere else.
template <typename T, typename TO = T>
struct RGMultiT {
  RGaussCoef k[PRAD_LOG_MAXSIG];
  const T *in[PRAD_LOG_MAXSIG];
  double *scratch[PRAD_LOG_MAXSIG];     // block states (rgauss_pass_kernel) / float64 causal pass (rgauss_line_kernel)
  TO *out[PRAD_LOG_MAXSIG];
};

// ---- reference pass: one lane per line, the causal recursion parked in a float64 scratch image ------------------------
// data viewed as [outer][ln][inner]; lane = (outer index, inner index).  The plainest statement of ITK's
// RecursiveSeparableImageFilter::FilterDataArray on the device: kept as the checker of the fast kernel below
// (PRAD_LOG_OLDLINE=1, tests/test_gpu_filters.py) -- 36 B of traffic per sample.
template <typename T, typename TO = T>
__global__ void __launch_bounds__(256) rgauss_line_kernel(const T *__restrict__ in, long long outer, int ln,
                                                          long long inner, RGaussCoef c,
                                                          double *__restrict__ scratch, TO *__restrict__ out) {
#pragma clang fp contract(off)  // ITK's line arithmetic is plain multiply / add; keep the same roundings
  const long long lines = outer * inner;
  const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= lines) return;
  const long long base = (line / inner) * ln * inner + (line % inner);
  const T *d = in + base;
  double *s = scratch + base;
  TO *o = out + base;
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
    for (int i = 4; i < ln; i++) {
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
  o[(long long)(ln - 1) * st] = (TO)(s[(long long)(ln - 1) * st] + a1);
  o[(long long)(ln - 2) * st] = (TO)(s[(long long)(ln - 2) * st] + a2);
  o[(long long)(ln - 3) * st] = (TO)(s[(long long)(ln - 3) * st] + a3);
  o[(long long)(ln - 4) * st] = (TO)(s[(long long)(ln - 4) * st] + a4);
  {
    // scratch[i-1] = data[i]*M1 + data[i+1]*M2 + data[i+2]*M3 + data[i+3]*M4 - (scratch[i]*D1 + ... + scratch[i+3]*D4)
    double dp0 = d[(long long)(ln - 4) * st], dp1 = y2, dp2 = y1, dp3 = v2;  // data[i], [i+1], [i+2], [i+3] at i = ln-4
    double q0 = a4, q1 = a3, q2 = a2, q3 = a1;                              // scratch[i], [i+1], [i+2], [i+3]
    for (int i = ln - 4; i > 0; i--) {
      double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
      v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
      o[(long long)(i - 1) * st] = (TO)(s[(long long)(i - 1) * st] + v);
      dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = d[(long long)(i - 1) * st];
      q3 = q2; q2 = q1; q1 = q0; q0 = v;
    }
  }
}

// ---- the pass the filter runs -----------------------------------------------------------------------------------------
// out[i] = causal[i] + anticausal[i] needs one direction kept while the other runs, but not sample by sample: sweep 1 runs
// the ANTI-causal recursion from the end of the line and keeps only its state (4 outputs) at every block boundary (32 B
// per RB samples); sweep 2 walks forward block by block -- causal recursion into registers, then the anti-causal one of
// the block RECOMPUTED downward from the saved state (same operations on the same operands: the same bits), sum, store.
// Traffic per sample: 4 (sweep 1) + 4 (sweep 2) read, 4 written, 2 x 32 / RB of states = 16 B at RB = 16.
//   CONTIG = false (strided axis): lane = line, neighbouring lanes are neighbouring inner positions: every access of the
//            wave is one row segment.
//   CONTIG = true (the contiguous axis, inner == 1): a wave owns 64 consecutive lines; a lane-per-line walk would touch 64
//            different cache lines per step, so every block goes through a per-wave LDS tile [64 lines][RB + 4 samples]:
//            the wave fetches the block of all its lines with coalesced loads (80-byte row pieces), each lane then takes
//            its own row (pitch 21: conflict-free); results return the same way.  Same sweeps, same arithmetic, same
//            16 B per sample (round 3's kernel for this axis parked float64 partials in HBM: 28 B, 180 us instead of
//            ~65 per sigma at 256^3).
// The Laplacian's  acc += term / spacing^2  is not part of the pass (the accumulating instantiation of round 3 needed 308
// VGPRs = one wave per SIMD and ran at a third of the speed): log_combine_kernel below.
// One step of the 4th-order recursion  y[k] = n[k] - (((y[k-1] D1 + y[k-2] D2) + y[k-3] D3) + y[k-4] D4),
// n[k] = ((w0 A1 + w1 A2) + w2 A3) + w3 A4  over the data window, in exactly that association (ITK's source order).  Only
// five of its sixteen float64 operations hang on the fresh y[k-1]: y[k-1] D1, three adds, the subtraction.  Left to the
// compiler every operation is issued right behind the one it depends on, and a dependent float64 operation waits ~20 cycles
// on this chip: ~300 cycles per step with one wave per SIMD, a third of the float64 issue rate with two (the register
// budget of the kernel allows no more).  Here the step is issued in a fixed order (volatile asm statements keep their order)
// with the numerator of the NEXT step and the products y D2, y D3, y D4 of the already known outputs placed between the
// five dependent operations.  Same operations, same operands, same roundings: bit-identical to the plain loop (the GPU
// tests compare the two kernels).
#define PRAD_DMUL_S(r, a, sc) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(sc))
#define PRAD_DADD(r, a, b) asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))
#define PRAD_DSUB(r, a, b) asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b))
struct RGChain {
  double y1, y2, y3;      // the last three outputs, newest first
  double e2, e3, e4;      // y[k-2] D2, y[k-3] D3, y[k-4] D4 of the coming step
  double n;               // numerator of the coming step
  double w0, w1, w2;      // the three newest samples of the data window
  // A1..A4: numerator coefficients in window order (newest sample first); window = (x0, x1, x2, x3) newest first
  __device__ __forceinline__ void enter(double o1, double o2, double o3, double o4, double x0, double x1, double x2, double x3,
                                        double A1, double A2, double A3, double A4, double D2, double D3, double D4) {
#pragma clang fp contract(off)
    y1 = o1; y2 = o2; y3 = o3;
    e2 = o2 * D2; e3 = o3 * D3; e4 = o4 * D4;
    n = x0 * A1 + x1 * A2 + x2 * A3 + x3 * A4;
    w0 = x0; w1 = x1; w2 = x2;
  }
  // produces the next output; `next` enters the window (NEXT = false: the last step of a stretch, no numerator prepared)
  template <bool NEXT>
  __device__ __forceinline__ double step(double next, double A1, double A2, double A3, double A4, double D1, double D2, double D3,
                                         double D4) {
    double m1, t1, t2, t3, t4, u1, u2, nn, b2, b3, b4, s1, s2, s3, y0;
    PRAD_DMUL_S(m1, y1, D1);
    if (NEXT) PRAD_DMUL_S(t1, next, A1);
    if (NEXT) PRAD_DMUL_S(t2, w0, A2);
    PRAD_DMUL_S(b2, y1, D2);
    PRAD_DADD(s1, m1, e2);
    if (NEXT) PRAD_DMUL_S(t3, w1, A3);
    if (NEXT) PRAD_DADD(u1, t1, t2);
    if (NEXT) PRAD_DMUL_S(t4, w2, A4);
    PRAD_DMUL_S(b3, y2, D3);
    PRAD_DADD(s2, s1, e3);
    PRAD_DMUL_S(b4, y3, D4);
    if (NEXT) PRAD_DADD(u2, u1, t3);
    PRAD_DADD(s3, s2, e4);
    if (NEXT) PRAD_DADD(nn, u2, t4);
    PRAD_DSUB(y0, n, s3);
    y3 = y2; y2 = y1; y1 = y0;
    e2 = b2; e3 = b3; e4 = b4;
    if (NEXT) {
      n = nn;
      w2 = w1; w1 = w0; w0 = next;
    }
    return y0;
  }
};

#define PRAD_RG_RB 16
// PRAD_RG_PREFETCH = 1 starts the loads of the next block before the current one is recursed; measured at 256^3, five sigmas
// per launch: 294 instead of 281 us on the strided axes, 541 instead of 430 on the contiguous one (20 more live registers);
// the causal block values in LDS instead of 40 VGPRs: 405 us.  The pass moves 16 B per sample at 4.8 TB/s (the copy rate of
// this GPU is 5.5): it is bound by those bytes, not by latencies (profiles/r04_probes.md).
#ifndef PRAD_RG_PREFETCH
#define PRAD_RG_PREFETCH 0
#endif
#ifndef PRAD_RG_WAVES
#define PRAD_RG_WAVES 2
#endif
template <typename T, bool CONTIG, bool PLAIN, typename TO = T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PRAD_RG_WAVES, PRAD_RG_WAVES))) rgauss_pass_kernel(RGMultiT<T, TO> M, long long outer, int ln, long long inner) {
#pragma clang fp contract(off)  // ITK's line arithmetic is plain multiply / add; keep the same roundings
  constexpr int RB = PRAD_RG_RB;
  constexpr int RM = RB + 4;                  // samples held: the block and, for the anti-causal start, the 4 behind it
  static_assert(RB == 16, "the staging loops move 16 + 4 columns");
  constexpr int TP = RM + 1;                  // LDS pitch (odd)
  __shared__ T stage_[CONTIG ? 4 : 1][CONTIG ? 64 : 1][TP];
  const int sg = blockIdx.y;
  const RGaussCoef &c = M.k[sg];
  const T *__restrict__ in = M.in[sg];
  TO *__restrict__ out = M.out[sg];
  const long long lines = CONTIG ? outer : outer * inner;
  const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool mine = line < lines;
  const int lane = threadIdx.x & 63;
  T(*tile)[TP] = stage_[CONTIG ? (threadIdx.x >> 6) : 0];
  const long long wl0 = line - lane;                       // first line of this wave (CONTIG)
  if (!CONTIG && !mine) return;
  if (CONTIG && wl0 >= lines) return;
  const long long lc = mine ? line : lines - 1;           // (idle lanes of a ragged last wave shadow the last line)
  const long long base = CONTIG ? lc * ln : (lc / inner) * ln * inner + (lc % inner);
  const T *d = in + base;
  const long long st = CONTIG ? 1 : inner;
  const int nwl = CONTIG ? (int)min((long long)64, lines - wl0) : 0;
  // block [b, b + w) of every line of the wave -> dst[0 .. w)   (w <= RM), in two halves so that the loads of the NEXT block
  // are in flight while the current one is recursed (a wave that loads, waits, computes leaves the memory system idle
  // half of the time at two waves per SIMD): issue() starts the global loads into pf[], commit() hands every lane its row.
  // WF > 0: the width is the compile-time constant WF (16 = a full block in sweep 1, 20 = block + look-ahead in sweep 2):
  // no predicate on the loads.
  auto issue = [&](auto wf_tag, int b, int w, T *pf) __attribute__((always_inline)) {
    constexpr int WF = decltype(wf_tag)::value;
    if (WF > 0) w = WF;
    if (CONTIG) {
      // 16 columns as 4 lines x 64-byte pieces per wave instruction, the (up to 4) columns behind them as 16 lines x 16 bytes
      if (nwl == 64 && WF > 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) pf[k] = in[(wl0 + k * 4 + (lane >> 4)) * ln + b + (lane & 15)];
        if (WF > 16) {
#pragma unroll
          for (int k = 0; k < 4; k++) pf[16 + k] = in[(wl0 + k * 16 + (lane >> 2)) * ln + b + 16 + (lane & 3)];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int t = k * 4 + (lane >> 4), cc = lane & 15;
          pf[k] = (t < nwl && cc < w) ? in[(wl0 + t) * ln + b + cc] : (T)0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int t = k * 16 + (lane >> 2), cc = 16 + (lane & 3);
          pf[16 + k] = (t < nwl && cc < w) ? in[(wl0 + t) * ln + b + cc] : (T)0;
        }
      }
    } else {
      const T *p = d + (long long)b * st;
#pragma unroll
      for (int j = 0; j < RM; j++) pf[j] = (WF > 0 ? j < WF : j < w) ? p[(long long)j * st] : (T)0;
    }
  };
  auto commit = [&](const T *pf, T *dst) __attribute__((always_inline)) {
    if (CONTIG) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 16; k++) tile[k * 4 + (lane >> 4)][lane & 15] = pf[k];
#pragma unroll
      for (int k = 0; k < 4; k++) tile[k * 16 + (lane >> 2)][16 + (lane & 3)] = pf[16 + k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < RM; j++) dst[j] = tile[lane][j];
    } else {
#pragma unroll
      for (int j = 0; j < RM; j++) dst[j] = pf[j];
    }
  };
  using W0 = std::integral_constant<int, 0>;
  using W16 = std::integral_constant<int, 16>;
  using W20 = std::integral_constant<int, 20>;
  const long long nlines_all = lines;
  // blocks [k RB, (k + 1) RB) for k < nb - 1; the last one takes the remainder, 4 .. RB + 3 samples (ln >= 4)
  const int nb = (ln - 4) / RB + 1;
  // state of boundary k (1 <= k < nb) = anticausal[k RB .. k RB + 3]: double index ((k - 1) * 4 + j) * lines + line
  double *sp = M.scratch[sg] + lc;
  // ---- sweep 1: anti-causal, from the end, states only ----
  {
    const int bl = (nb - 1) * RB, ll = ln - bl;          // last block: 4 .. RB + 3 samples
    T dv[RM], pf[RM];
    issue(W0{}, bl, ll, pf);
    commit(pf, dv);
    if (PRAD_RG_PREFETCH && nb >= 3) issue(W16{}, (nb - 2) * RB, RB, pf);          // (block 0 is not walked in sweep 1)
    double v2 = 0, y1 = 0, y2 = 0, y3 = 0;
#pragma unroll
    for (int j = 0; j < RM; j++) {                        // (register array: compile-time indices only)
      if (j == ll - 1) v2 = dv[j];
      if (j == ll - 2) y1 = dv[j];
      if (j == ll - 3) y2 = dv[j];
      if (j == ll - 4) y3 = dv[j];
    }
    double a1 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-1
    double a2 = v2 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-2
    double a3 = y1 * c.M1 + v2 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-3
    double a4 = y2 * c.M1 + y1 * c.M2 + v2 * c.M3 + v2 * c.M4;     // index ln-4
    a1 -= v2 * c.BM1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
    a2 -= a1 * c.D1 + v2 * c.BM2 + v2 * c.BM3 + v2 * c.BM4;
    a3 -= a2 * c.D1 + a1 * c.D2 + v2 * c.BM3 + v2 * c.BM4;
    a4 -= a3 * c.D1 + a2 * c.D2 + a1 * c.D3 + v2 * c.BM4;
    double dp0 = y3, dp1 = y2, dp2 = y1, dp3 = v2;                   // data[i .. i+3] at i = ln-4
    double q0 = a4, q1 = a3, q2 = a2, q3 = a1;                       // anticausal[i .. i+3]
    // the rest of the last block, downward: samples bl + ll - 5 .. bl
#pragma unroll
    for (int jj = RM - 1; jj >= 0; jj--) {
      if (jj <= ll - 5) {
        double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
        v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
        dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = dv[jj];
        q3 = q2; q2 = q1; q1 = q0; q0 = v;
      }
    }
    // full blocks nb - 2 .. 0: q0..q3 = anticausal[(kb + 1) RB .. + 3] is the state of boundary kb + 1
    for (int kb = nb - 2; kb >= 0; kb--) {
      if (mine) {
        sp[((long long)kb * 4 + 0) * nlines_all] = q0;
        sp[((long long)kb * 4 + 1) * nlines_all] = q1;
        sp[((long long)kb * 4 + 2) * nlines_all] = q2;
        sp[((long long)kb * 4 + 3) * nlines_all] = q3;
      }
      if (kb == 0) break;                                  // block 0's anti-causal part is recomputed in sweep 2
      if (!PRAD_RG_PREFETCH) issue(W16{}, kb * RB, RB, pf);
      commit(pf, dv);
      if (PRAD_RG_PREFETCH && kb >= 2) issue(W16{}, (kb - 1) * RB, RB, pf);
      if (PLAIN) {
#pragma unroll
        for (int jj = RB - 1; jj >= 0; jj--) {
          double v = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
          v -= q0 * c.D1 + q1 * c.D2 + q2 * c.D3 + q3 * c.D4;
          dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = dv[jj];
          q3 = q2; q2 = q1; q1 = q0; q0 = v;
        }
      } else {
        RGChain ch;
        ch.enter(q0, q1, q2, q3, dp0, dp1, dp2, dp3, c.M1, c.M2, c.M3, c.M4, c.D2, c.D3, c.D4);
        double o4 = q2, o5 = q3;
#pragma unroll
        for (int jj = RB - 1; jj >= 0; jj--) {
          o5 = o4; o4 = ch.y3;
          ch.step<true>((double)dv[jj], c.M1, c.M2, c.M3, c.M4, c.D1, c.D2, c.D3, c.D4);
        }
        (void)o5;
        q0 = ch.y1; q1 = ch.y2; q2 = ch.y3; q3 = o4;
        dp0 = ch.w0; dp1 = ch.w1; dp2 = ch.w2; dp3 = (double)dv[3];
      }
    }
  }
  // ---- sweep 2: forward, block by block ----
  double p1 = 0, p2 = 0, p3 = 0, p4 = 0;        // causal[b-1 .. b-4]
  double dm1 = 0, dm2 = 0, dm3 = 0;             // data[b-1 .. b-3]
  T pf2[RM];
  auto issue2 = [&](int b) __attribute__((always_inline)) {   // block + the 4 samples behind it (the anti-causal start)
    if (b + RM <= ln) issue(W20{}, b, RM, pf2);
    else issue(W0{}, b, ln - b, pf2);
  };
  if (PRAD_RG_PREFETCH) issue2(0);
  for (int kb = 0; kb < nb; kb++) {
    const int b = kb * RB;
    const bool lastblk = kb == nb - 1;
    const int len = lastblk ? ln - b : RB;      // RB, or 4 .. RB + 3 for the last block
    T dv[RM];
    double cv[RM];
    if (!PRAD_RG_PREFETCH) issue2(b);
    commit(pf2, dv);
    if (PRAD_RG_PREFETCH && !lastblk) issue2(b + RB);
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
    if (PLAIN || len != RB) {
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
    } else {
      // a full block: samples j0 .. RB - 1 on the scheduled chain (the numerator of sample j uses dv[j] itself: it is prepared
      // one step ahead, the first one on entry)
      RGChain ch;
      auto run = [&](auto j0_tag) __attribute__((always_inline)) {
        constexpr int J0 = decltype(j0_tag)::value;
        ch.enter(p1, p2, p3, p4, (double)dv[J0], dm1, dm2, dm3, c.N0, c.N1, c.N2, c.N3, c.D2, c.D3, c.D4);
#pragma unroll
        for (int j = J0; j < RB; j++) {
          if (j < RB - 1) cv[j] = ch.step<true>((double)dv[j + 1], c.N0, c.N1, c.N2, c.N3, c.D1, c.D2, c.D3, c.D4);
          else cv[j] = ch.step<false>(0.0, c.N0, c.N1, c.N2, c.N3, c.D1, c.D2, c.D3, c.D4);
        }
      };
      if (kb == 0) run(std::integral_constant<int, 4>{});
      else run(std::integral_constant<int, 0>{});
      p1 = cv[RB - 1]; p2 = cv[RB - 2]; p3 = cv[RB - 3]; p4 = cv[RB - 4];
      dm1 = dv[RB - 1]; dm2 = dv[RB - 2]; dm3 = dv[RB - 3];
    }
    // anti-causal recursion of the block, downward, from the saved state (or from the end of the line)
    double q0, q1, q2, q3, e0, e1, e2, e3;      // anticausal[i .. i+3], data[i .. i+3] at i = b + len
    int jtop = len - 1;                         // block-local index of the first sample still to produce
    if (lastblk) {
      double v2 = 0, y1 = 0, y2 = 0, y3 = 0;
#pragma unroll
      for (int j = 0; j < RM; j++) {
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
      // the four end samples are final here
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
      q0 = sp[((long long)kb * 4 + 0) * nlines_all];
      q1 = sp[((long long)kb * 4 + 1) * nlines_all];
      q2 = sp[((long long)kb * 4 + 2) * nlines_all];
      q3 = sp[((long long)kb * 4 + 3) * nlines_all];
      e0 = dv[RB]; e1 = dv[RB + 1]; e2 = dv[RB + 2]; e3 = dv[RB + 3];
    }
    if (PLAIN || lastblk) {
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
    } else {
      RGChain ch;
      ch.enter(q0, q1, q2, q3, e0, e1, e2, e3, c.M1, c.M2, c.M3, c.M4, c.D2, c.D3, c.D4);
#pragma unroll
      for (int jj = RB - 1; jj >= 0; jj--) {
        double v;
        if (jj > 0) v = ch.step<true>((double)dv[jj], c.M1, c.M2, c.M3, c.M4, c.D1, c.D2, c.D3, c.D4);
        else v = ch.step<false>(0.0, c.M1, c.M2, c.M3, c.M4, c.D1, c.D2, c.D3, c.D4);
        cv[jj] += v;
      }
    }
    // store
    if (CONTIG) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < RM; j++)
        if (j < len) tile[lane][j] = (T)(TO)cv[j];       // (the staging tile has the input's type: TO values pass through it exactly)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int t = k * 4 + (lane >> 4), cc = lane & 15;
        if (t < nwl && cc < len) out[(wl0 + t) * ln + b + cc] = (TO)tile[t][cc];
      }
      if (len > 16) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int t = k * 16 + (lane >> 2), cc = 16 + (lane & 3);
          if (t < nwl && cc < len) out[(wl0 + t) * ln + b + cc] = (TO)tile[t][cc];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < RM; j++)
        if (j < len) out[base + (long long)(b + j) * st] = (TO)cv[j];
    }
  }
}

// Laplacian = sum over the dimensions of (second-derivative image) / spacing^2, accumulated in ITK's order and with the
// roundings of its separate accumulation step: acc = (T)((double)acc + (double)term / spacing^2), starting from 0
// (itkLaplacianRecursiveGaussianImageFilter.hxx; T = float, the cumulative image's type); the sum is cast to the output
// type TOUT at the end (float64 for float64 inputs).  out may alias term[0] when TOUT = T.
#define PRAD_LOG_MAXTERMS 8
template <typename T>
struct LogTerms {
  const T *term[PRAD_LOG_MAXTERMS];
  double sp2[PRAD_LOG_MAXTERMS];
  int n;
};
template <typename T, typename TOUT = T>
__global__ void __launch_bounds__(256) log_combine_kernel(LogTerms<T> L, long long n, TOUT *out) {
#pragma clang fp contract(off)
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    T acc = (T)0;
#pragma unroll
    for (int k = 0; k < PRAD_LOG_MAXTERMS; k++) {
      if (k < L.n) {
        const T f = L.term[k][i];
        acc = (T)((double)acc + (double)f / L.sp2[k]);
      }
    }
    out[i] = (TOUT)acc;
  }
}

}  // namespace prad
