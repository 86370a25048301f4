// prad_filters.hip -- C ABI of the filter stack in front of the texture matrices (include/pyradiomics_amd.h: prad_swt_level1*,
// prad_log*): SURVEY.md section 8 rows a10 / a11, radiomics/imageoperations.py:756-970.  A translation unit of its own so that
// the filter kernels rebuild without the texture kernels.
#include "prad_runtime.h"
#include "kernels_filters.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>


using namespace prad;

namespace {

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

int copy_back(Context &c, double *host, const double *dev, size_t count) {
  PRAD_HIP(hipMemcpyAsync(host, dev, sizeof(double) * count, hipMemcpyDeviceToHost, c.own_stream));
  PRAD_HIP(hipStreamSynchronize(c.own_stream));
  return PRAD_OK;
}

// ------------------------------------------------------------------------------------------------
// filters
// ------------------------------------------------------------------------------------------------
// dtype: the image's element type (include/pyradiomics_amd.h: 0 float32, 1 float64, 2 int32, 3 int16); anything but float64 is
// only taken by the fused 3-D kernel (PRAD_E_UNSUPPORTED otherwise: the caller converts and calls again)
int swt_level1_dev(const void *in_any, int dtype, const int *size, int Nd, const double *dec_lo, const double *dec_hi, int flen,
                   const int *axes, int naxes, double *out, hipStream_t s) {
  const double *in = static_cast<const double *>(in_any);
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!in || !out || !dec_lo || !dec_hi || !axes) return fail(PRAD_E_ARG, "swt: NULL pointer");
  if (dtype < 0 || dtype > 3) return fail(PRAD_E_ARG, "swt: dtype code %d", dtype);
  if (flen < 2 || flen > PRAD_MAX_TAPS) return fail(PRAD_E_ARG, "swt: filter length %d outside [2,%d]", flen, PRAD_MAX_TAPS);
  if (naxes < 1 || naxes > Nd) return fail(PRAD_E_ARG, "swt: naxes=%d", naxes);
  FilterTaps T;
  T.F = flen;
  for (int k = 0; k < flen; k++) { T.lo[k] = dec_lo[k]; T.hi[k] = dec_hi[k]; }
  for (int a = 0; a < naxes; a++) {
    if (axes[a] < 0 || axes[a] >= Nd) return fail(PRAD_E_ARG, "swt: axis %d out of range", axes[a]);
    if (g.size[axes[a]] % 2) return fail(PRAD_E_ARG, "swt: axis %d has odd length %d (pad first, imageoperations.py:914-919)", axes[a], g.size[axes[a]]);
  }
  PRAD_TRY(c.begin_call(s));
  // 3-D transform over all three axes in x, y, z order (what getWaveletImage asks for): one fused kernel (kernels_filters.h)
  if (Nd == 3 && naxes == 3 && axes[0] == 2 && axes[1] == 1 && axes[2] == 0 && (flen == 2 || flen == 4 || flen == 6) &&
      g.size[0] >= flen && g.size[1] >= flen && g.size[2] >= flen && !getenv("PRAD_SWT_NOFUSE")) {
    {
      Timed t(c, "swt", s);
      const int Nz = g.size[0], Ny = g.size[1], Nx = g.size[2];
      const int tiles = ((Nx + PRAD_SWT3_TX - 1) / PRAD_SWT3_TX) * ((Ny + PRAD_SWT3_TY - 1) / PRAD_SWT3_TY);
      // z chunks: enough workgroups for ~4 per CU, each chunk long against its F - 1 warm-up planes
      int nzc = std::max(1, std::min((4 * cu_count() + tiles - 1) / tiles, Nz / 32));
      const int CZ = (Nz + nzc - 1) / nzc;
      nzc = (Nz + CZ - 1) / CZ;
      const dim3 grid((Nx + PRAD_SWT3_TX - 1) / PRAD_SWT3_TX, (Ny + PRAD_SWT3_TY - 1) / PRAD_SWT3_TY, nzc);
#define PRAD_SWT3(FF)                                                                                                            \
  do {                                                                                                                           \
    if (dtype == 1) hipLaunchKernelGGL((swt3_fused_kernel<FF, double>), grid, dim3(256), 0, s, (const double *)in_any, Nz, Ny, Nx, T, CZ, out); \
    else if (dtype == 0) hipLaunchKernelGGL((swt3_fused_kernel<FF, float>), grid, dim3(256), 0, s, (const float *)in_any, Nz, Ny, Nx, T, CZ, out); \
    else if (dtype == 2) hipLaunchKernelGGL((swt3_fused_kernel<FF, int>), grid, dim3(256), 0, s, (const int *)in_any, Nz, Ny, Nx, T, CZ, out); \
    else hipLaunchKernelGGL((swt3_fused_kernel<FF, short>), grid, dim3(256), 0, s, (const short *)in_any, Nz, Ny, Nx, T, CZ, out); \
  } while (0)
      if (flen == 6) PRAD_SWT3(6);
      else if (flen == 4) PRAD_SWT3(4);
      else PRAD_SWT3(2);
#undef PRAD_SWT3
      PRAD_TRY(check_launch("swt3_fused_kernel"));
    }
    PRAD_TRY(c.end_call(s));
    PRAD_HIP(hipStreamSynchronize(s));
    c.last_path = "swt-fused";
    return PRAD_OK;
  }
  if (dtype != 1) {
    PRAD_TRY(c.end_call(s));
    return fail(PRAD_E_UNSUPPORTED, "swt: only the fused 3-D transform takes images that are not float64");
  }
  // stage k holds 2^k arrays; stages alternate between two workspaces, the last stage writes `out`
  double *ws[2] = {nullptr, nullptr};
  const size_t n = (size_t)g.n;
  if (naxes >= 2) PRAD_TRY(c.get<double>("swt_a", n * ((size_t)1 << (naxes - 1)), &ws[0]));
  if (naxes >= 3) PRAD_TRY(c.get<double>("swt_b", n * ((size_t)1 << (naxes - 2)), &ws[1]));
  const double *src = in;
  {
    Timed t(c, "swt", s);
    for (int a = 0; a < naxes; a++) {
      const int count = 1 << a;
      double *dst = (a == naxes - 1) ? out : ws[(naxes - 2 - a) & 1];
      const int ax = axes[a];
      long long outer = 1;
      for (int d = 0; d < ax; d++) outer *= g.size[d];
      const long long inner = g.stride[ax];
      const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((g.n + 255) / 256, 8192));
      const int N = g.size[ax];
      // geometry-indexed kernel when the taps wrap at most once and the grid limits hold; the general kernel otherwise
      const bool geo_ok = T.F <= N && !getenv("PRAD_SWT_OLD") &&
                          (inner == 1 ? outer <= 65535LL * 65535LL : (N <= 65535 && outer <= 65535));
      for (int j = 0; j < count; j++) {
        const double *in = src + (size_t)j * n;
        double *o_lo = dst + (size_t)(2 * j) * n, *o_hi = dst + (size_t)(2 * j + 1) * n;
        if (geo_ok && inner == 1) {
          const dim3 grid((unsigned)((N + 255) / 256), (unsigned)std::min<long long>(outer, 65535), (unsigned)((outer + 65534) / 65535));
          hipLaunchKernelGGL(swt_axis2_kernel<true>, grid, dim3(256), 0, s, in, outer, N, inner, T, o_lo, o_hi);
        } else if (geo_ok) {
          const dim3 grid((unsigned)((inner + 255) / 256), (unsigned)N, (unsigned)outer);
          hipLaunchKernelGGL(swt_axis2_kernel<false>, grid, dim3(256), 0, s, in, outer, N, inner, T, o_lo, o_hi);
        } else {
          hipLaunchKernelGGL(swt_axis_kernel, dim3(gx), dim3(256), 0, s, in, outer, N, inner, T, o_lo, o_hi);
        }
        PRAD_TRY(check_launch("swt_axis_kernel"));
      }
      src = dst;
    }
  }
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  c.last_path = "swt";
  return PRAD_OK;
}

// ITK itkRecursiveGaussianImageFilter.hxx SetUp / ComputeNCoefficients / ComputeDCoefficients /
// ComputeRemainingCoefficients (symmetric orders 0 and 2)
void rgauss_n(double sigmad, double A1, double B1, double A2, double B2, double N[4], double &SN, double &DN, double &EN) {
  const double W1 = 0.6681, L1 = -1.3932, W2 = 2.0787, L2 = -1.3732;
  const double s1 = sin(W1 / sigmad), s2 = sin(W2 / sigmad), c1 = cos(W1 / sigmad), c2 = cos(W2 / sigmad);
  const double e1 = exp(L1 / sigmad), e2 = exp(L2 / sigmad);
  N[0] = A1 + A2;
  N[1] = e2 * (B2 * s2 - (A2 + 2 * A1) * c2);
  N[1] += e1 * (B1 * s1 - (A1 + 2 * A2) * c1);
  N[2] = (A1 + A2) * c2 * c1;
  N[2] -= B1 * c2 * s1 + B2 * c1 * s2;
  N[2] *= 2 * e1 * e2;
  N[2] += A2 * e1 * e1 + A1 * e2 * e2;
  N[3] = e2 * e1 * e1 * (B2 * s2 - A2 * c2);
  N[3] += e1 * e2 * e2 * (B1 * s1 - A1 * c1);
  SN = N[0] + N[1] + N[2] + N[3];
  DN = N[1] + 2 * N[2] + 3 * N[3];
  EN = N[1] + 4 * N[2] + 9 * N[3];
}

RGaussCoef rgauss_coefficients(double sigma, double spacing, int order, bool normalize) {
  const double W1 = 0.6681, L1 = -1.3932, W2 = 2.0787, L2 = -1.3732;
  const double A1[3] = {1.3530, -0.6724, -1.3563}, B1[3] = {1.8151, -3.4327, 5.2318};
  const double A2[3] = {-0.3531, 0.6724, 0.3446}, B2[3] = {0.0902, 0.6100, -2.2355};
  const double sigmad = sigma / fabs(spacing);
  const double c1 = cos(W1 / sigmad), c2 = cos(W2 / sigmad), e1 = exp(L1 / sigmad), e2 = exp(L2 / sigmad);
  double D[4];
  D[3] = e1 * e1 * e2 * e2;
  D[2] = -2 * c1 * e1 * e2 * e2;
  D[2] += -2 * c2 * e2 * e1 * e1;
  D[1] = 4 * c2 * c1 * e1 * e2;
  D[1] += e1 * e1 + e2 * e2;
  D[0] = -2 * (e2 * c2 + e1 * c1);
  const double SD = 1.0 + D[0] + D[1] + D[2] + D[3];
  const double DD = D[0] + 2 * D[1] + 3 * D[2] + 4 * D[3];
  const double ED = D[0] + 4 * D[1] + 9 * D[2] + 16 * D[3];
  double N[4], SN, DN, EN;
  if (order == 0) {
    rgauss_n(sigmad, A1[0], B1[0], A2[0], B2[0], N, SN, DN, EN);
    const double alpha0 = 2 * SN / SD - N[0];
    for (int i = 0; i < 4; i++) N[i] /= alpha0;
  } else {
    const double scale = normalize ? sigma * sigma : 1.0;
    double N0s[4], N2s[4], SN0, DN0, EN0, SN2, DN2, EN2;
    rgauss_n(sigmad, A1[0], B1[0], A2[0], B2[0], N0s, SN0, DN0, EN0);
    rgauss_n(sigmad, A1[2], B1[2], A2[2], B2[2], N2s, SN2, DN2, EN2);
    const double beta = -(2 * SN2 - SD * N2s[0]) / (2 * SN0 - SD * N0s[0]);
    for (int i = 0; i < 4; i++) N[i] = N2s[i] + beta * N0s[i];
    SN = SN2 + beta * SN0; DN = DN2 + beta * DN0; EN = EN2 + beta * EN0;
    const double alpha2 = (EN * SD * SD - ED * SN * SD - 2 * DN * DD * SD + 2 * DD * DD * SN) / (SD * SD * SD);
    for (int i = 0; i < 4; i++) N[i] = N[i] * scale / alpha2;
  }
  RGaussCoef c;
  c.N0 = N[0]; c.N1 = N[1]; c.N2 = N[2]; c.N3 = N[3];
  c.D1 = D[0]; c.D2 = D[1]; c.D3 = D[2]; c.D4 = D[3];
  c.M1 = N[1] - D[0] * N[0]; c.M2 = N[2] - D[1] * N[0]; c.M3 = N[3] - D[2] * N[0]; c.M4 = -D[3] * N[0];
  const double SNn = c.N0 + c.N1 + c.N2 + c.N3, SM = c.M1 + c.M2 + c.M3 + c.M4, SDd = 1.0 + D[0] + D[1] + D[2] + D[3];
  c.BN1 = D[0] * SNn / SDd; c.BN2 = D[1] * SNn / SDd; c.BN3 = D[2] * SNn / SDd; c.BN4 = D[3] * SNn / SDd;
  c.BM1 = D[0] * SM / SDd; c.BM2 = D[1] * SM / SDd; c.BM3 = D[2] * SM / SDd; c.BM4 = D[3] * SM / SDd;
  return c;
}

// nsig sigmas of one input in the same launches (blockIdx.y = sigma): see RGMultiT in kernels_filters.h.
// TIN = the input's real type (float: integer and float32 images, converted exactly by the caller; double: float64 images).
// ITK's pipeline (itkLaplacianRecursiveGaussianImageFilter.h/.hxx), which the reference reaches through
// sitk.LaplacianRecursiveGaussianImageFilter (imageoperations.py:824-830): per dimension the DERIVATIVE filter reads the
// input image itself (RecursiveGaussianImageFilter<InputImage, Image<float>>), the zero-order smoothing filters of the other
// dimensions follow in increasing ITK direction, every image between the passes and the cumulative image are float
// (InternalRealType = float for any input type), the sum is cast to the output type (= TIN) at the end.  Round 5: this pass
// order reproduces the float32 order statistics the reference recorded for brain1 bit for bit (tests/test_notebook_pin.py);
// rounds 3-4 ran the smoothing passes first (and kept float64 images for float64 inputs), 1-4 ulp off.
template <typename TIN>
int log_multi_dev(const TIN *in, const int *size, int Nd, const double *spacing, const double *sigmas, int nsig,
                  int normalize, TIN *const *outs, hipStream_t s) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!in || !outs || !spacing || !sigmas) return fail(PRAD_E_ARG, "log: NULL pointer");
  if (nsig < 1 || nsig > PRAD_LOG_MAXSIG) return fail(PRAD_E_ARG, "log: %d sigmas per call outside [1, %d]", nsig, PRAD_LOG_MAXSIG);
  if (Nd > PRAD_LOG_MAXTERMS) return fail(PRAD_E_UNSUPPORTED, "log: %d dimensions", Nd);
  for (int q = 0; q < nsig; q++) {
    if (!(sigmas[q] > 0.0)) return fail(PRAD_E_ARG, "log: sigma must be > 0");
    if (!outs[q]) return fail(PRAD_E_ARG, "log: NULL output");
  }
  for (int d = 0; d < Nd; d++)
    if (g.size[d] < 4) return fail(PRAD_E_ARG, "log: axis %d has %d < 4 samples (imageoperations.py:811)", d, g.size[d]);
  PRAD_TRY(c.begin_call(s));
  const size_t n = (size_t)g.n;
  const bool reference_kernel = getenv("PRAD_LOG_OLDLINE") != nullptr;     // one lane per line, float64 scratch image
  const bool plain_steps = getenv("PRAD_LOG_PLAIN") != nullptr;
  float *bufA[PRAD_LOG_MAXSIG], *bufB[PRAD_LOG_MAXSIG];
  float *term[PRAD_LOG_MAXTERMS][PRAD_LOG_MAXSIG];       // second-derivative (and smoothed) image of every dimension
  double *scratch[PRAD_LOG_MAXSIG];
  for (int q = 0; q < nsig; q++) {
    const std::string tag = q ? "#s" + std::to_string(q) : std::string();
    PRAD_TRY(c.get<float>(("log_a" + tag).c_str(), n, &bufA[q]));
    PRAD_TRY(c.get<float>(("log_b" + tag).c_str(), n, &bufB[q]));
    for (int k = 0; k < Nd; k++) PRAD_TRY(c.get<float>(("log_t" + std::to_string(k) + tag).c_str(), n, &term[k][q]));
    // block states of rgauss_pass_kernel: 4 doubles per PRAD_RG_RB samples (n / 4 + a line's worth); the reference kernel
    // parks the whole causal pass
    PRAD_TRY(c.get<double>(("log_scratch" + tag).c_str(), reference_kernel ? n : n / 4 + 4 * (n / g.size[Nd - 1]) + 64, &scratch[q]));
  }
  {
    Timed t(c, "log", s);
    // ITK dimension order x, y, z = array axes Nd-1 .. 0.
    int nterm = 0;
    double term_sp2[PRAD_LOG_MAXTERMS];
    for (int dim = Nd - 1; dim >= 0; dim--) {
      const float *cur[PRAD_LOG_MAXSIG];
      int flip = 0;
      // one pass along `ax`; FIRST: reads the input image (TIN), else the float image `cur`; last: into the term image
      auto pass = [&](auto first_tag, int ax, int order, bool last) -> int {
        constexpr bool FIRST = decltype(first_tag)::value;
        using TI = typename std::conditional<FIRST, TIN, float>::type;
        RGMultiT<TI, float> M;
        memset(&M, 0, sizeof(M));
        float *dst[PRAD_LOG_MAXSIG];
        for (int q = 0; q < nsig; q++) {
          M.k[q] = rgauss_coefficients(sigmas[q], spacing[ax], order, normalize != 0);
          dst[q] = last ? term[nterm][q] : (flip ? bufB[q] : bufA[q]);
          M.in[q] = FIRST ? reinterpret_cast<const TI *>(in) : reinterpret_cast<const TI *>(cur[q]);
          M.scratch[q] = scratch[q];
          M.out[q] = dst[q];
        }
        long long outer = 1;
        for (int d = 0; d < ax; d++) outer *= g.size[d];
        const long long inner = g.stride[ax];
        const long long lines = outer * inner;
        if (lines > 0x7fffffffLL * 256) return fail(PRAD_E_UNSUPPORTED, "log: %lld lines", lines);
        const dim3 grid((unsigned)((lines + 255) / 256), (unsigned)nsig);
        if (reference_kernel) {
          for (int q = 0; q < nsig; q++)
            hipLaunchKernelGGL((rgauss_line_kernel<TI, float>), dim3(grid.x), dim3(256), 0, s, M.in[q], outer, g.size[ax], inner, M.k[q],
                               M.scratch[q], M.out[q]);
          PRAD_TRY(check_launch("rgauss_line_kernel"));
        } else if (plain_steps) {         // (PRAD_LOG_PLAIN=1: the compiler's own order of the recursion step, for A/B runs)
          if (inner == 1) hipLaunchKernelGGL((rgauss_pass_kernel<TI, true, true, float>), grid, dim3(256), 0, s, M, outer, g.size[ax], inner);
          else hipLaunchKernelGGL((rgauss_pass_kernel<TI, false, true, float>), grid, dim3(256), 0, s, M, outer, g.size[ax], inner);
          PRAD_TRY(check_launch("rgauss_pass_kernel"));
        } else {
          if (inner == 1) hipLaunchKernelGGL((rgauss_pass_kernel<TI, true, false, float>), grid, dim3(256), 0, s, M, outer, g.size[ax], inner);
          else hipLaunchKernelGGL((rgauss_pass_kernel<TI, false, false, float>), grid, dim3(256), 0, s, M, outer, g.size[ax], inner);
          PRAD_TRY(check_launch("rgauss_pass_kernel"));
        }
        for (int q = 0; q < nsig; q++) cur[q] = dst[q];
        if (!last) flip ^= 1;
        return PRAD_OK;
      };
      // the derivative filter on the input, then the smoothing filters of the other dimensions in increasing ITK direction
      PRAD_TRY(pass(std::true_type{}, dim, 2, Nd == 1));
      int left = Nd - 1;
      for (int other = Nd - 1; other >= 0; other--) {
        if (other == dim) continue;
        left--;
        PRAD_TRY(pass(std::false_type{}, other, 0, left == 0));
      }
      term_sp2[nterm] = spacing[dim] * spacing[dim];
      nterm++;
    }
    // acc = sum of term / spacing^2 in ITK's order (x, y, z) with the roundings of its separate accumulation step
    for (int q = 0; q < nsig; q++) {
      LogTerms<float> L;
      memset(&L, 0, sizeof(L));
      L.n = nterm;
      for (int k = 0; k < nterm; k++) {
        L.term[k] = term[k][q];
        L.sp2[k] = term_sp2[k];
      }
      const unsigned blocks = (unsigned)std::min<long long>((g.n + 255) / 256, 256LL * 32);
      hipLaunchKernelGGL((log_combine_kernel<float, TIN>), dim3(blocks), dim3(256), 0, s, L, g.n, outs[q]);
    }
    PRAD_TRY(check_launch("log_combine_kernel"));
  }
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  c.last_path = reference_kernel ? "log-reference" : "log";
  return PRAD_OK;
}

template <typename T>
int log_dev_t(const T *in, const int *size, int Nd, const double *spacing, double sigma, int normalize, T *out, hipStream_t s) {
  if (!out) return fail(PRAD_E_ARG, "log: NULL pointer");
  return log_multi_dev<T>(in, size, Nd, spacing, &sigma, 1, normalize, &out, s);
}

int log_dev(const float *in, const int *size, int Nd, const double *spacing, double sigma, int normalize, float *out,
            hipStream_t s) {
  return log_dev_t<float>(in, size, Nd, spacing, sigma, normalize, out, s);
}

template <typename T>
int log_host(const T *in, const int *size, int Nd, const double *spacing, double sigma, int normalize, T *out, const char *tag) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!in || !out) return fail(PRAD_E_ARG, "log: NULL pointer");
  const size_t n = (size_t)g.n;
  T *d_in = nullptr, *d_out = nullptr;
  PRAD_TRY(c.get<T>((std::string("log_in") + tag).c_str(), n, &d_in));
  PRAD_TRY(c.get<T>((std::string("log_out") + tag).c_str(), n, &d_out));
  PRAD_HIP(hipMemcpyAsync(d_in, in, sizeof(T) * n, hipMemcpyHostToDevice, c.own_stream));
  int rc = log_dev_t<T>(d_in, size, Nd, spacing, sigma, normalize, d_out, c.own_stream);
  if (rc != PRAD_OK) return rc;
  PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(T) * n, hipMemcpyDeviceToHost, c.own_stream));
  PRAD_HIP(hipStreamSynchronize(c.own_stream));
  return PRAD_OK;
}

}  // namespace

extern "C" {
// ---- filters ---------------------------------------------------------------------------------
int prad_swt_level1_dev(const double *in, const int *size, int Nd, const double *dec_lo, const double *dec_hi,
                        int flen, const int *axes, int naxes, double *out, void *stream) {
  return swt_level1_dev(in, 1, size, Nd, dec_lo, dec_hi, flen, axes, naxes, out, (hipStream_t)stream);
}
int prad_swt_level1_any_dev(const void *in, int dtype, const int *size, int Nd, const double *dec_lo, const double *dec_hi,
                            int flen, const int *axes, int naxes, double *out, void *stream) {
  return swt_level1_dev(in, dtype, size, Nd, dec_lo, dec_hi, flen, axes, naxes, out, (hipStream_t)stream);
}
int prad_swt_level1(const double *in, const int *size, int Nd, const double *dec_lo, const double *dec_hi, int flen,
                    const int *axes, int naxes, double *out) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!in || !out || naxes < 1 || naxes > Nd) return fail(PRAD_E_ARG, "swt: bad arguments");
  const size_t n = (size_t)g.n, nout = n << naxes;
  double *d_in = nullptr, *d_out = nullptr;
  PRAD_TRY(c.get<double>("swt_in", n, &d_in));
  PRAD_TRY(c.get<double>("swt_out", nout, &d_out));
  PRAD_HIP(hipMemcpyAsync(d_in, in, sizeof(double) * n, hipMemcpyHostToDevice, c.own_stream));
  int rc = swt_level1_dev(d_in, 1, size, Nd, dec_lo, dec_hi, flen, axes, naxes, d_out, c.own_stream);
  if (rc != PRAD_OK) return rc;
  return copy_back(c, out, d_out, nout);
}
int prad_log_multi_dev(const float *in, const int *size, int Nd, const double *spacing, const double *sigmas, int nsig,
                       int normalize, float *const *outs, void *stream) {
  return log_multi_dev<float>(in, size, Nd, spacing, sigmas, nsig, normalize, outs, (hipStream_t)stream);
}
int prad_log_multi_dev_f64(const double *in, const int *size, int Nd, const double *spacing, const double *sigmas, int nsig,
                           int normalize, double *const *outs, void *stream) {
  return log_multi_dev<double>(in, size, Nd, spacing, sigmas, nsig, normalize, outs, (hipStream_t)stream);
}
int prad_log_dev(const float *in, const int *size, int Nd, const double *spacing, double sigma, int normalize,
                 float *out, void *stream) {
  return log_dev(in, size, Nd, spacing, sigma, normalize, out, (hipStream_t)stream);
}
int prad_log_dev_f64(const double *in, const int *size, int Nd, const double *spacing, double sigma, int normalize,
                     double *out, void *stream) {
  return log_dev_t<double>(in, size, Nd, spacing, sigma, normalize, out, (hipStream_t)stream);
}
int prad_log(const float *in, const int *size, int Nd, const double *spacing, double sigma, int normalize,
             float *out) {
  return log_host<float>(in, size, Nd, spacing, sigma, normalize, out, "");
}
int prad_log_f64(const double *in, const int *size, int Nd, const double *spacing, double sigma, int normalize,
                 double *out) {
  return log_host<double>(in, size, Nd, spacing, sigma, normalize, out, "64");
}

}  // extern "C"
