// prad_features.hip -- C ABI of the segment-mode feature kernels (include/pyradiomics_amd.h); third translation unit
// of libpyradiomics_amd.so.
#include "kernels_features.h"

using namespace prad;

extern "C" int prad_glcm_features_dev(const double *glcm, int Ng, int Na, int symmetric, double *out, int *empty,
                                      void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!glcm || !out || !empty || Ng < 1 || Na < 1) return fail(PRAD_E_ARG, "glcm_features: bad arguments");
  const size_t lds = sizeof(double) * ((size_t)5 * Ng + 4);
  if (lds > 60 * 1024) return fail(PRAD_E_UNSUPPORTED, "glcm_features: Ng=%d exceeds the LDS marginals", Ng);
  hipStream_t s = (hipStream_t)stream;
  double *d_out = nullptr;
  int *d_empty = nullptr;
  PRAD_TRY(c.get<double>("gf_out", (size_t)Na * GF_COUNT, &d_out));
  PRAD_TRY(c.get<int>("gf_empty", (size_t)Na, &d_empty));
  {
    Timed t(c, "features", s);
    hipLaunchKernelGGL(glcm_matrix_features_kernel, dim3(Na), dim3(PRAD_FEAT_THREADS), lds, s, glcm, Ng, Na, symmetric,
                       d_out, d_empty);
    PRAD_TRY(check_launch("glcm_matrix_features_kernel"));
  }
  PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * Na * GF_COUNT, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipMemcpyAsync(empty, d_empty, sizeof(int) * Na, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  return PRAD_OK;
}

extern "C" int prad_zone_matrix_features_dev(const double *P, int Ni, int Nj, int Na, long long stride_i,
                                             long long stride_j, long long stride_a, const double *jvals, double *out,
                                             int *empty, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!P || !jvals || !out || !empty || Ni < 1 || Nj < 1 || Na < 1) return fail(PRAD_E_ARG, "zone_features: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  double *d_out = nullptr, *d_j = nullptr, *d_scr = nullptr;
  int *d_empty = nullptr;
  PRAD_TRY(c.get<double>("zf_out", (size_t)Na * ZM_COUNT, &d_out));
  PRAD_TRY(c.get<int>("zf_empty", (size_t)Na, &d_empty));
  PRAD_TRY(c.get<double>("zf_jvals", (size_t)Nj, &d_j));
  PRAD_TRY(c.get<double>("zf_scratch", (size_t)Na * ((size_t)Ni + Nj), &d_scr));
  PRAD_HIP(hipMemcpyAsync(d_j, jvals, sizeof(double) * Nj, hipMemcpyHostToDevice, s));
  {
    Timed t(c, "features", s);
    hipLaunchKernelGGL(zone_matrix_features_kernel, dim3(Na), dim3(PRAD_FEAT_THREADS), 0, s, P, Ni, Nj, Na, stride_i,
                       stride_j, stride_a, d_j, d_scr, d_out, d_empty);
    PRAD_TRY(check_launch("zone_matrix_features_kernel"));
  }
  PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * Na * ZM_COUNT, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipMemcpyAsync(empty, d_empty, sizeof(int) * Na, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  return PRAD_OK;
}

extern "C" int prad_ngtdm_features_dev(const double *P, int Ng, double *out, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!P || !out || Ng < 1) return fail(PRAD_E_ARG, "ngtdm_features: bad arguments");
  const size_t lds = sizeof(double) * ((size_t)3 * Ng + 4);
  if (lds > 60 * 1024) return fail(PRAD_E_UNSUPPORTED, "ngtdm_features: Ng=%d exceeds the LDS level table", Ng);
  hipStream_t s = (hipStream_t)stream;
  double *d_out = nullptr;
  PRAD_TRY(c.get<double>("nf_out", 8, &d_out));
  {
    Timed t(c, "features", s);
    hipLaunchKernelGGL(ngtdm_matrix_features_kernel, dim3(1), dim3(PRAD_FEAT_THREADS), lds, s, P, Ng, d_out);
    PRAD_TRY(check_launch("ngtdm_matrix_features_kernel"));
  }
  PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * 5, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  return PRAD_OK;
}
