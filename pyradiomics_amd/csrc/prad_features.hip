// prad_features.hip -- C ABI of the segment-mode feature kernels (include/pyradiomics_amd.h); third translation unit
// of libpyradiomics_amd.so.
#include "kernels_features.h"

using namespace prad;

extern "C" int prad_glcm_features_dev(const double *glcm, int Ng, int Na, int symmetric, double *out, int *empty,
                                      void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!glcm || !out || !empty || Ng < 1 || Na < 1) return fail(PRAD_E_ARG, "glcm_features: bad arguments");
  const size_t lds = sizeof(double) * ((size_t)5 * Ng + PRAD_FEAT_WAVES);
  if (lds > 60 * 1024) return fail(PRAD_E_UNSUPPORTED, "glcm_features: Ng=%d exceeds the LDS marginals", Ng);
  hipStream_t s = (hipStream_t)stream;
  // values and "empty" flags in ONE device block and one copy into pinned memory (a copy into the caller's pageable array
  // goes through the runtime's bounce buffer and blocks: 45 such calls per case were ~2 ms)
  const size_t nout = (size_t)Na * GF_COUNT;
  double *d_out = nullptr;
  PRAD_TRY(c.get<double>("gf_out", nout + (size_t)(Na + 1) / 2 + 1, &d_out));
  int *d_empty = reinterpret_cast<int *>(d_out + nout);
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned("feat_pin", sizeof(double) * (nout + (size_t)Na + 2), &pin));
  const bool enq = c.deferred && c.in_arena(out, sizeof(double) * nout) && c.in_arena(empty, sizeof(int) * Na);
  const bool direct = enq && Context::zero_copy();      // the kernel stores into the arena itself
  {
    Timed t(c, "features", s);
    hipLaunchKernelGGL(glcm_matrix_features_kernel, dim3(Na), dim3(PRAD_FEAT_THREADS), lds, s, glcm, Ng, Na, symmetric,
                       direct ? out : d_out, direct ? empty : d_empty);
    PRAD_TRY(check_launch("glcm_matrix_features_kernel"));
  }
  if (direct) return PRAD_OK;
  if (enq) {
    // enqueue only (include/pyradiomics_amd.h, "result arena"): the values arrive with the stream
    if ((const void *)empty == (const void *)(out + nout)) {      // flags right behind the values, as on the device: one copy
      PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * nout + sizeof(int) * Na, hipMemcpyDeviceToHost, s));
      return PRAD_OK;
    }
    PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * nout, hipMemcpyDeviceToHost, s));
    PRAD_HIP(hipMemcpyAsync(empty, d_empty, sizeof(int) * Na, hipMemcpyDeviceToHost, s));
    return PRAD_OK;
  }
  PRAD_HIP(hipMemcpyAsync(pin, d_out, sizeof(double) * nout + sizeof(int) * Na, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  memcpy(out, pin, sizeof(double) * nout);
  memcpy(empty, (const char *)pin + sizeof(double) * nout, sizeof(int) * Na);
  return PRAD_OK;
}

// launch only, everything on the device (prad_glszm_features_dev: the column values and their number were found there)
namespace prad {
int zone_features_launch(Context &c, hipStream_t s, const double *P, int Ni, int Njcap, long long stride_i,
                         const double *jvals_d, const int *nj_dev, double *out_d, int *empty_d) {
  double *d_scr = nullptr;
  PRAD_TRY(c.get<double>("zf_scratch", (size_t)Ni + (size_t)Njcap, &d_scr));
  Timed t(c, "features", s);
  hipLaunchKernelGGL(zone_matrix_features_kernel, dim3(1), dim3(PRAD_FEAT_THREADS), 0, s, P, Ni, Njcap, 1, stride_i, 1LL, 0LL,
                     jvals_d, d_scr, out_d, empty_d, nj_dev);
  return check_launch("zone_matrix_features_kernel");
}
}  // namespace prad

extern "C" int prad_zone_matrix_features_dev(const double *P, int Ni, int Nj, int Na, long long stride_i,
                                             long long stride_j, long long stride_a, const double *jvals, double *out,
                                             int *empty, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!P || !out || !empty || Ni < 1 || Nj < 1 || Na < 1) return fail(PRAD_E_ARG, "zone_features: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t nout = (size_t)Na * ZM_COUNT;
  double *d_out = nullptr, *d_j = nullptr, *d_scr = nullptr;
  PRAD_TRY(c.get<double>("zf_out", nout + (size_t)(Na + 1) / 2 + 1, &d_out));
  int *d_empty = reinterpret_cast<int *>(d_out + nout);
  PRAD_TRY(c.get<double>("zf_jvals", (size_t)Nj, &d_j));
  PRAD_TRY(c.get<double>("zf_scratch", (size_t)Na * ((size_t)Ni + Nj), &d_scr));
  const bool enq = c.deferred && c.in_arena(out, sizeof(double) * nout) && c.in_arena(empty, sizeof(int) * Na);
  void *pin = nullptr;        // [jvals in | values + flags out], pinned (see prad_glcm_features_dev)
  if (enq) {
    if (jvals) PRAD_TRY(c.arena_alloc(sizeof(double) * (size_t)Nj, &pin));   // (a staging slot nobody rewrites before the copy ran)
  } else {
    PRAD_TRY(c.get_pinned("zfeat_pin", sizeof(double) * ((size_t)Nj + nout + (size_t)Na + 2), &pin));
  }
  double *pj = (double *)pin, *pout = pin ? pj + Nj : nullptr;
  if (jvals) {      // (NULL: the size values are 1 .. Nj -- run lengths, dependence counts + 1 -- and no table travels)
    memcpy(pj, jvals, sizeof(double) * Nj);
    PRAD_HIP(hipMemcpyAsync(d_j, pj, sizeof(double) * Nj, hipMemcpyHostToDevice, s));
  } else {
    d_j = nullptr;
  }
  {
    Timed t(c, "features", s);
    const bool direct = enq && Context::zero_copy();
    hipLaunchKernelGGL(zone_matrix_features_kernel, dim3(Na), dim3(PRAD_FEAT_THREADS), 0, s, P, Ni, Nj, Na, stride_i,
                       stride_j, stride_a, d_j, d_scr, direct ? out : d_out, direct ? empty : d_empty);
    PRAD_TRY(check_launch("zone_matrix_features_kernel"));
    if (direct) return PRAD_OK;
  }
  if (enq) {
    if ((const void *)empty == (const void *)(out + nout)) {
      PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * nout + sizeof(int) * Na, hipMemcpyDeviceToHost, s));
      return PRAD_OK;
    }
    PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * nout, hipMemcpyDeviceToHost, s));
    PRAD_HIP(hipMemcpyAsync(empty, d_empty, sizeof(int) * Na, hipMemcpyDeviceToHost, s));
    return PRAD_OK;
  }
  PRAD_HIP(hipMemcpyAsync(pout, d_out, sizeof(double) * nout + sizeof(int) * Na, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  memcpy(out, pout, sizeof(double) * nout);
  memcpy(empty, (const char *)pout + sizeof(double) * nout, sizeof(int) * Na);
  return PRAD_OK;
}

extern "C" int prad_ngtdm_features_dev(const double *P, int Ng, double *out, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!P || !out || Ng < 1) return fail(PRAD_E_ARG, "ngtdm_features: bad arguments");
  const size_t lds = sizeof(double) * ((size_t)3 * Ng + PRAD_FEAT_WAVES);
  if (lds > 60 * 1024) return fail(PRAD_E_UNSUPPORTED, "ngtdm_features: Ng=%d exceeds the LDS level table", Ng);
  hipStream_t s = (hipStream_t)stream;
  double *d_out = nullptr;
  PRAD_TRY(c.get<double>("nf_out", 8, &d_out));
  {
    Timed t(c, "features", s);
    const bool direct = c.deferred && c.in_arena(out, sizeof(double) * 5) && Context::zero_copy();
    hipLaunchKernelGGL(ngtdm_matrix_features_kernel, dim3(1), dim3(PRAD_FEAT_THREADS), lds, s, P, Ng, direct ? out : d_out);
    PRAD_TRY(check_launch("ngtdm_matrix_features_kernel"));
    if (direct) return PRAD_OK;
  }
  if (c.deferred && c.in_arena(out, sizeof(double) * 5)) {
    PRAD_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * 5, hipMemcpyDeviceToHost, s));
    return PRAD_OK;
  }
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned("nfeat_pin", sizeof(double) * 8, &pin));
  PRAD_HIP(hipMemcpyAsync(pin, d_out, sizeof(double) * 5, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  memcpy(out, pin, sizeof(double) * 5);
  return PRAD_OK;
}
