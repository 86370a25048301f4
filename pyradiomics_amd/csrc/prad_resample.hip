// prad_resample.hip -- C ABI of the resampling kernels (include/pyradiomics_amd.h); translation unit of
// libpyradiomics_amd.so.
#include <math.h>
#include "kernels_resample.h"

using namespace prad;

namespace {
template <typename T>
int run_resample(Context &c, hipStream_t s, const T *src, const ResampleGeo &g, int interp, bool is_int, double tmin,
                 double tmax, T *out) {
  const long long nin = (long long)g.in[0] * g.in[1] * g.in[2], nout = (long long)g.out[0] * g.out[1] * g.out[2];
  double *coef = nullptr;
  if (interp == 3) {
    PRAD_TRY(c.get<double>("resample_coef", (size_t)nin, &coef));
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((nin + 255) / 256, 8192));
    hipLaunchKernelGGL(to_f64_kernel<T>, dim3(gx), dim3(256), 0, s, src, nin, coef);
    PRAD_TRY(check_launch("to_f64_kernel"));
    const int horizon = (int)ceil(log(1e-10) / log(fabs(sqrt(3.0) - 2.0)));
    for (int ax = 2; ax >= 0; ax--) {                       // ITK filters dimension 0 (x) first
      const int N = g.in[ax];
      long long outer = 1, inner = 1;
      for (int d = 0; d < ax; d++) outer *= g.in[d];
      for (int d = ax + 1; d < 3; d++) inner *= g.in[d];
      const long long lines = outer * inner;
      hipLaunchKernelGGL(bspline_prefilter_kernel, dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, s, coef, outer, N,
                         inner, horizon);
      PRAD_TRY(check_launch("bspline_prefilter_kernel"));
    }
  }
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((nout + 255) / 256, 16384));
  hipLaunchKernelGGL(resample_kernel<T>, dim3(gx), dim3(256), 0, s, src, coef, g, interp, is_int ? 1 : 0, tmin, tmax, out);
  PRAD_TRY(check_launch("resample_kernel"));
  PRAD_HIP(hipStreamSynchronize(s));
  return PRAD_OK;
}
}  // namespace

extern "C" int prad_resample_dev(const void *image, int dtype, const int *size, int Nd, const double *start,
                                 const double *step, const int *newsize, int interpolator, void *out, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!image || !size || !start || !step || !newsize || !out || Nd < 1 || Nd > 3)
    return fail(PRAD_E_ARG, "resample: bad arguments (Nd must be 1..3)");
  if (interpolator != 0 && interpolator != 1 && interpolator != 3)
    return fail(PRAD_E_UNSUPPORTED, "resample: interpolator %d (0 nearest, 1 linear, 3 cubic B-spline)", interpolator);
  ResampleGeo g;
  g.nd = Nd;
  for (int d = 0; d < 3; d++) {
    g.in[d] = g.out[d] = 1;
    g.start[d] = 0.0;
    g.step[d] = 1.0;
  }
  for (int d = 0; d < Nd; d++) {
    if (size[d] < 1 || newsize[d] < 1) return fail(PRAD_E_ARG, "resample: empty axis");
    g.in[3 - Nd + d] = size[d];
    g.out[3 - Nd + d] = newsize[d];
    g.start[3 - Nd + d] = start[d];
    g.step[3 - Nd + d] = step[d];
  }
  hipStream_t s = (hipStream_t)stream;
  Timed t(c, "resample", s);
  switch (dtype) {
    case 0: return run_resample<float>(c, s, (const float *)image, g, interpolator, false, 0, 0, (float *)out);
    case 1: return run_resample<double>(c, s, (const double *)image, g, interpolator, false, 0, 0, (double *)out);
    case 2: return run_resample<int>(c, s, (const int *)image, g, interpolator, true, -2147483648.0, 2147483647.0, (int *)out);
    case 3: return run_resample<short>(c, s, (const short *)image, g, interpolator, true, -32768.0, 32767.0, (short *)out);
    default: return fail(PRAD_E_ARG, "resample: dtype %d", dtype);
  }
}
