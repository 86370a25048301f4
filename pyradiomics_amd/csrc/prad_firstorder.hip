// prad_firstorder.hip -- C ABI of the first-order statistics (include/pyradiomics_amd.h); second translation unit of
// libpyradiomics_amd.so so that the rocPRIM sort templates do not slow down rebuilding the texture kernels.
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdlib.h>
#include "kernels_firstorder.h"

using namespace prad;

namespace {

// numpy's linear-interpolation quantile (numpy/lib/_function_base_impl.py: _compute_virtual_index with
// alpha = beta = 1, _get_gamma, _lerp): the statistic np.nanpercentile / np.nanmedian return for finite data
struct Quantile {
  long long prev, next;
  double gamma;
};
Quantile quantile_pos(long long m, double q) {
  const double virt = (double)(m - 1) * q;
  Quantile r;
  r.prev = (long long)floor(virt);
  r.gamma = virt - (double)r.prev;
  if (r.prev < 0) r.prev = 0;
  if (r.prev > m - 1) r.prev = m - 1;
  r.next = r.prev + 1 > m - 1 ? m - 1 : r.prev + 1;
  return r;
}
double lerp_np(double a, double b, double t) {
  const double d = b - a;
  return t >= 0.5 ? b - d * (1 - t) : a + d * t;
}

// the 10 order statistics in ONE copy through pinned memory (ten 8-byte copies into a pageable array were ten blocking
// round trips per derived image)
struct FoPick {
  long long idx[10];
};
__global__ void fo_pick_kernel(const double *__restrict__ sorted, FoPick p, double *__restrict__ out) {
  if (threadIdx.x < 10) out[threadIdx.x] = p.idx[threadIdx.x] >= 0 ? sorted[p.idx[threadIdx.x]] : 0.0;
}
int pick_sorted(Context &c, hipStream_t s, const double *sorted, const FoPick &p, double *os) {
  double *d = nullptr;
  void *hp = nullptr;
  PRAD_TRY(c.get<double>("fo_pick", 16, &d));
  PRAD_TRY(c.get_pinned("fo_pick_h", sizeof(double) * 16, &hp));
  hipLaunchKernelGGL(fo_pick_kernel, dim3(1), dim3(64), 0, s, sorted, p, d);
  PRAD_TRY(check_launch("fo_pick_kernel"));
  PRAD_HIP(hipMemcpyAsync(hp, d, sizeof(double) * 10, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  for (int k = 0; k < 10; k++)
    if (p.idx[k] >= 0) os[k] = ((const double *)hp)[k];
  return PRAD_OK;
}

int sum_partials(Context &c, hipStream_t s, const double *partial_d, int blocks, int width, double *out) {
  void *hp = nullptr;
  PRAD_TRY(c.get_pinned("fo_partials_h", sizeof(double) * PRAD_FO_BLOCKS * 8, &hp));
  double *h = (double *)hp;
  PRAD_HIP(hipMemcpyAsync(h, partial_d, sizeof(double) * blocks * width, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  for (int k = 0; k < width; k++) {
    double acc = 0;
    for (int b = 0; b < blocks; b++) acc += h[b * width + k];
    out[k] = acc;
  }
  return PRAD_OK;
}

}  // namespace

// launches KERNEL<T> for the image dtype code of include/pyradiomics_amd.h (0 float32, 1 float64, 2 int32, 3 int16)
#define FO_DISPATCH(KERNEL, GRID, BLOCK, ...)                                                                       \
  switch (dtype) {                                                                                                  \
    case 0: hipLaunchKernelGGL(KERNEL<float>, GRID, BLOCK, 0, s, (const float *)image, mask, n, __VA_ARGS__); break;  \
    case 1: hipLaunchKernelGGL(KERNEL<double>, GRID, BLOCK, 0, s, (const double *)image, mask, n, __VA_ARGS__); break; \
    case 2: hipLaunchKernelGGL(KERNEL<int>, GRID, BLOCK, 0, s, (const int *)image, mask, n, __VA_ARGS__); break;      \
    default: hipLaunchKernelGGL(KERNEL<short>, GRID, BLOCK, 0, s, (const short *)image, mask, n, __VA_ARGS__); break; \
  }

extern "C" int prad_firstorder_dev(const void *image, int dtype, const uint8_t *mask, long long n,
                                   double voxelArrayShift, double *out, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!image || !mask || !out || n < 1) return fail(PRAD_E_ARG, "firstorder: bad arguments");
  if (n > 2147483647LL) return fail(PRAD_E_UNSUPPORTED, "firstorder: more than 2^31-1 voxels");
  if (dtype < 0 || dtype > 3) return fail(PRAD_E_ARG, "firstorder: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  double *partial = nullptr;
  PRAD_TRY(c.get<double>("fo_partial", (size_t)PRAD_FO_BLOCKS * 8, &partial));
  void *php = nullptr;
  PRAD_TRY(c.get_pinned("fo_partials_h", sizeof(double) * PRAD_FO_BLOCKS * 8, &php));
  double *ph = (double *)php;
  // fixed block layout over the n voxels of the array: the partial sums do not depend on anything but the input
  const int blocks = (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, PRAD_FO_BLOCKS));
  {
    Timed t(c, "firstorder", s);
    FO_DISPATCH(fo_sums_kernel, dim3(blocks), dim3(256), voxelArrayShift, partial);
    PRAD_TRY(check_launch("fo_sums_kernel"));
  }
  PRAD_HIP(hipMemcpyAsync(ph, partial, sizeof(double) * blocks * 5, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  double sums[2] = {0, 0}, vmin = INFINITY, vmax = -INFINITY, cnt = 0;
  for (int b = 0; b < blocks; b++) {
    sums[0] += ph[b * 5];
    sums[1] += ph[b * 5 + 1];
    vmin = fmin(vmin, ph[b * 5 + 2]);
    vmax = fmax(vmax, ph[b * 5 + 3]);
    cnt += ph[b * 5 + 4];
  }
  const long long m = (long long)cnt;
  if (m < 1) return fail(PRAD_E_ARG, "firstorder: empty ROI");
  const double mu = sums[0] / (double)m;

  // order statistics: the two neighbours of each requested quantile (minimum / maximum come from the reduction)
  const double qs[5] = {0.1, 0.25, 0.5, 0.75, 0.9};
  Quantile qp[5];
  double os[12];
  long long ranks[10];
  for (int k = 0; k < 5; k++) {
    qp[k] = quantile_pos(m, qs[k]);
    ranks[2 * k] = qp[k].prev;
    ranks[2 * k + 1] = qp[k].next;
  }
  os[10] = vmin;
  os[11] = vmax;
  // ---- integer images with a value range that fits the LDS: everything from the exact histogram -------------------
  const bool exact_off = getenv("PRAD_FO_NO_EXACT") != nullptr;   // (tests: the selection route on integer images)
  if (!exact_off && (dtype == 2 || dtype == 3) && vmax - vmin < (double)PRAD_FO_EXACT_MAX && m >= (1LL << 16)) {
    const int R = (int)(vmax - vmin) + 1;
    unsigned *hist = nullptr;
    PRAD_TRY(c.get<unsigned>("fo_exact_hist", PRAD_FO_EXACT_MAX, &hist));
    void *hp = nullptr;
    PRAD_TRY(c.get_pinned("fo_exact_hist_h", sizeof(unsigned) * PRAD_FO_EXACT_MAX, &hp));
    const unsigned *H = (const unsigned *)hp;
    PRAD_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned) * R, s));
    {
      Timed t(c, "firstorder", s);
      const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((n + 8191) / 8192, 512));
      const size_t lds = sizeof(unsigned) * (size_t)R;
      switch (dtype) {
        case 2:
          PRAD_HIP(hipFuncSetAttribute((const void *)fo_exact_hist_kernel<int>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(fo_exact_hist_kernel<int>, dim3(gx), dim3(1024), lds, s, (const int *)image, mask, n, vmin, R, hist);
          break;
        default:
          PRAD_HIP(hipFuncSetAttribute((const void *)fo_exact_hist_kernel<short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(fo_exact_hist_kernel<short>, dim3(gx), dim3(1024), lds, s, (const short *)image, mask, n, vmin, R, hist);
          break;
      }
      PRAD_TRY(check_launch("fo_exact_hist_kernel"));
    }
    PRAD_HIP(hipMemcpyAsync(hp, hist, sizeof(unsigned) * R, hipMemcpyDeviceToHost, s));
    PRAD_HIP(hipStreamSynchronize(s));
    // order statistics from the cumulative counts (ranks ascend)
    {
      long long below = 0;
      int k = 0;
      for (int b = 0; b < R && k < 10; b++) {
        below += H[b];
        while (k < 10 && ranks[k] < below) os[k++] = vmin + (double)b;
      }
      if (k < 10) return fail(PRAD_E_HIP, "firstorder: exact histogram does not cover the ROI (internal error)");
    }
    double pq[5];
    for (int k = 0; k < 5; k++) pq[k] = lerp_np(os[2 * k], os[2 * k + 1], qp[k].gamma);
    const double median = (m % 2) ? os[4] : (os[4] + os[5]) / 2.0;
    // sums over the distinct values, ascending, in extended precision (the reference sums the raw voxels pairwise)
    long double a1 = 0, a2 = 0, a3 = 0, a4 = 0, bc = 0, bs = 0;
    for (int b = 0; b < R; b++) {
      if (!H[b]) continue;
      const long double w = (long double)H[b], x = (long double)(vmin + (double)b), d = x - (long double)mu, d2 = d * d;
      a1 += w * fabsl(d);
      a2 += w * d2;
      a3 += w * d2 * d;
      a4 += w * d2 * d2;
      if ((double)x >= pq[0] && (double)x <= pq[4]) {
        bc += w;
        bs += w * x;
      }
    }
    double rmad = NAN;
    if (bc > 0) {
      const long double mub = bs / bc;
      long double band = 0;
      for (int b = 0; b < R; b++) {
        const double x = vmin + (double)b;
        if (H[b] && x >= pq[0] && x <= pq[4]) band += (long double)H[b] * fabsl((long double)x - mub);
      }
      rmad = (double)(band / bc);
    }
    const double dm = (double)m;
    out[PRAD_FO_NP] = dm;
    out[PRAD_FO_ENERGY] = sums[1];
    out[PRAD_FO_MINIMUM] = os[10];
    out[PRAD_FO_P10] = pq[0];
    out[PRAD_FO_P25] = pq[1];
    out[PRAD_FO_MEDIAN] = median;
    out[PRAD_FO_P75] = pq[3];
    out[PRAD_FO_P90] = pq[4];
    out[PRAD_FO_MAXIMUM] = os[11];
    out[PRAD_FO_MEAN] = mu;
    out[PRAD_FO_MAD] = (double)(a1 / dm);
    out[PRAD_FO_RMAD] = rmad;
    out[PRAD_FO_M2] = (double)(a2 / dm);
    out[PRAD_FO_M3] = (double)(a3 / dm);
    out[PRAD_FO_M4] = (double)(a4 / dm);
    c.last_path = "firstorder-exact";
    return PRAD_OK;
  }
  bool selected = false;
  static const long long select_from = getenv("PRAD_FO_SELECT_MIN") ? atoll(getenv("PRAD_FO_SELECT_MIN")) : (1LL << 20);
  if (m >= select_from && vmax > vmin && isfinite(vmin) && isfinite(vmax)) {
    // selection: histogram over PRAD_FO_BINS monotone bins, gather the bins that hold the ranks, sort only those
    unsigned *hist = nullptr, *cursors = nullptr;
    PRAD_TRY(c.get<unsigned>("fo_hist", PRAD_FO_BINS, &hist));
    PRAD_TRY(c.get<unsigned>("fo_cursors", PRAD_FO_MAXSEL, &cursors));
    PRAD_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned) * PRAD_FO_BINS, s));
    PRAD_HIP(hipMemsetAsync(cursors, 0, sizeof(unsigned) * PRAD_FO_MAXSEL, s));
    const double scale = (double)PRAD_FO_BINS / (vmax - vmin);
    void *hp = nullptr;
    PRAD_TRY(c.get_pinned("fo_hist_h", sizeof(unsigned) * PRAD_FO_BINS, &hp));
    unsigned *hist_h = (unsigned *)hp;
    {
      Timed t(c, "firstorder", s);
      const unsigned hgx = (unsigned)std::max<long long>(1, std::min<long long>((n + 4095) / 4096, 512));
      FO_DISPATCH(fo_hist_kernel, dim3(hgx), dim3(1024), vmin, scale, hist);
      PRAD_TRY(check_launch("fo_hist_kernel"));
    }
    PRAD_HIP(hipMemcpyAsync(hist_h, hist, sizeof(unsigned) * PRAD_FO_BINS, hipMemcpyDeviceToHost, s));
    PRAD_HIP(hipStreamSynchronize(s));
    // bin of every rank, position inside it
    FoSel sel;
    sel.nsel = 0;
    long long within[10];
    int which[10];
    {
      long long below = 0;
      int k = 0;                       // ranks are ascending
      for (int b = 0; b < PRAD_FO_BINS && k < 10; b++) {
        const long long bc = hist_h[b];
        while (k < 10 && ranks[k] < below + bc) {
          if (sel.nsel == 0 || sel.bin[sel.nsel - 1] != b) sel.bin[sel.nsel++] = b;
          which[k] = sel.nsel - 1;
          within[k] = ranks[k] - below;
          k++;
        }
        below += bc;
      }
      if (k < 10) return fail(PRAD_E_HIP, "firstorder: histogram does not cover the ROI (internal error)");
    }
    // bins that hold one distinct value need no gather; the others are gathered and sorted if they are small enough
    const long long cap = 1LL << 22;
    bool single[PRAD_FO_MAXSEL] = {false};
    double single_val[PRAD_FO_MAXSEL] = {0};
    long long all = 0;
    for (int q = 0; q < sel.nsel; q++) all += hist_h[sel.bin[q]];
    const unsigned rgx = (unsigned)std::max<long long>(1, std::min<long long>((n + 4095) / 4096, 512));
    if (all > cap) {
      unsigned long long *range = nullptr;
      PRAD_TRY(c.get<unsigned long long>("fo_range", 2 * PRAD_FO_MAXSEL, &range));
      unsigned long long init[2 * PRAD_FO_MAXSEL], got[2 * PRAD_FO_MAXSEL];
      for (int q = 0; q < PRAD_FO_MAXSEL; q++) { init[2 * q] = ~0ull; init[2 * q + 1] = 0ull; }
      PRAD_HIP(hipMemcpyAsync(range, init, sizeof(init), hipMemcpyHostToDevice, s));
      {
        Timed t(c, "firstorder", s);
        FO_DISPATCH(fo_binrange_kernel, dim3(rgx), dim3(1024), vmin, scale, sel, range);
        PRAD_TRY(check_launch("fo_binrange_kernel"));
      }
      PRAD_HIP(hipMemcpyAsync(got, range, sizeof(got), hipMemcpyDeviceToHost, s));
      PRAD_HIP(hipStreamSynchronize(s));
      for (int q = 0; q < sel.nsel; q++)
        if (got[2 * q] == got[2 * q + 1]) {
          single[q] = true;
          single_val[q] = fo_unkey(got[2 * q]);
        }
    }
    FoSel gsel;                        // the bins to gather
    gsel.nsel = 0;
    int gidx[PRAD_FO_MAXSEL];
    long long total = 0;
    for (int q = 0; q < sel.nsel; q++) {
      gidx[q] = -1;
      if (single[q]) continue;
      gidx[q] = gsel.nsel;
      gsel.bin[gsel.nsel] = sel.bin[q];
      gsel.off[gsel.nsel] = (unsigned)total;
      gsel.nsel++;
      total += hist_h[sel.bin[q]];
    }
    if (total <= cap) {
      if (total > 0) {
        double *gath = nullptr;
        PRAD_TRY(c.get<double>("fo_gather", (size_t)(2 * total), &gath));
        double *gsorted = gath + total;
        size_t tmp_bytes = 0;
        PRAD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, gath, gsorted, (int)total, 0, 64, s));
        uint8_t *tmp = nullptr;
        PRAD_TRY(c.get<uint8_t>("fo_sort_tmp", tmp_bytes + 16, &tmp));
        {
          Timed t(c, "firstorder", s);
          FO_DISPATCH(fo_gather_kernel, dim3(rgx), dim3(1024), vmin, scale, gsel, cursors, gath);
          PRAD_TRY(check_launch("fo_gather_kernel"));
          PRAD_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, gath, gsorted, (int)total, 0, 64, s));
        }
        FoPick pk;
        for (int k = 0; k < 10; k++)
          pk.idx[k] = single[which[k]] ? -1 : (long long)gsel.off[gidx[which[k]]] + within[k];
        PRAD_TRY(pick_sorted(c, s, gsorted, pk, os));
      }
      for (int k = 0; k < 10; k++)
        if (single[which[k]]) os[k] = single_val[which[k]];
      selected = true;
    }
  }
  if (!selected) {
    // small ROIs (and large ones whose selected bins are too full): compact the ROI and sort it
    double *vals = nullptr, *sorted = nullptr;
    unsigned long long *count = nullptr;
    PRAD_TRY(c.get<double>("fo_vals", (size_t)m, &vals));
    PRAD_TRY(c.get<double>("fo_sorted", (size_t)m, &sorted));
    PRAD_TRY(c.get<unsigned long long>("fo_count", 1, &count));
    PRAD_HIP(hipMemsetAsync(count, 0, sizeof(unsigned long long), s));
    size_t tmp_bytes = 0;
    PRAD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, vals, sorted, (int)m, 0, 64, s));
    uint8_t *tmp = nullptr;
    PRAD_TRY(c.get<uint8_t>("fo_sort_tmp", tmp_bytes + 16, &tmp));
    {
      Timed t(c, "firstorder", s);
      const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096));
      FO_DISPATCH(fo_compact_kernel, dim3(gx), dim3(256), vals, count);
      PRAD_TRY(check_launch("fo_compact_kernel"));
      PRAD_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, vals, sorted, (int)m, 0, 64, s));
    }
    FoPick pk;
    for (int k = 0; k < 10; k++) pk.idx[k] = ranks[k];
    PRAD_TRY(pick_sorted(c, s, sorted, pk, os));
  }
  double pq[5];
  for (int k = 0; k < 5; k++) pq[k] = lerp_np(os[2 * k], os[2 * k + 1], qp[k].gamma);
  // np.median averages the two middle elements for even counts (np.mean of the pair), middle element otherwise
  const double median = (m % 2) ? os[4] : (os[4] + os[5]) / 2.0;

  {
    Timed t(c, "firstorder", s);
    FO_DISPATCH(fo_central_kernel, dim3(blocks), dim3(256), mu, pq[0], pq[4], partial);
    PRAD_TRY(check_launch("fo_central_kernel"));
  }
  double cen[6];
  PRAD_TRY(sum_partials(c, s, partial, blocks, 6, cen));
  double rmad = NAN;
  if (cen[4] > 0) {
    const double mu_band = cen[5] / cen[4];
    {
      Timed t(c, "firstorder", s);
      FO_DISPATCH(fo_band_kernel, dim3(blocks), dim3(256), mu_band, pq[0], pq[4], partial);
      PRAD_TRY(check_launch("fo_band_kernel"));
    }
    double band;
    PRAD_TRY(sum_partials(c, s, partial, blocks, 1, &band));
    rmad = band / cen[4];
  }
  const double dm = (double)m;
  out[PRAD_FO_NP] = dm;
  out[PRAD_FO_ENERGY] = sums[1];
  out[PRAD_FO_MINIMUM] = os[10];
  out[PRAD_FO_P10] = pq[0];
  out[PRAD_FO_P25] = pq[1];
  out[PRAD_FO_MEDIAN] = median;
  out[PRAD_FO_P75] = pq[3];
  out[PRAD_FO_P90] = pq[4];
  out[PRAD_FO_MAXIMUM] = os[11];
  out[PRAD_FO_MEAN] = mu;
  out[PRAD_FO_MAD] = cen[0] / dm;
  out[PRAD_FO_RMAD] = rmad;
  out[PRAD_FO_M2] = cen[1] / dm;
  out[PRAD_FO_M3] = cen[2] / dm;
  out[PRAD_FO_M4] = cen[3] / dm;
  c.last_path = selected ? "firstorder-select" : "firstorder-sort";
  return PRAD_OK;
}

#define PRAD_FO_QUEUE_CAP (1u << 18)      // least number of gathered elements the queue route sorts (2 MB)

extern "C" int prad_firstorder_queue_dev(const void *image, int dtype, const uint8_t *mask, long long n, long long roi_count,
                                         double voxelArrayShift, double *out, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!image || !mask || !out || n < 1) return fail(PRAD_E_ARG, "firstorder: bad arguments");
  if (n > 2147483647LL) return fail(PRAD_E_UNSUPPORTED, "firstorder: more than 2^31-1 voxels");
  if (dtype < 0 || dtype > 3) return fail(PRAD_E_ARG, "firstorder: dtype %d", dtype);
  static const long long select_from = getenv("PRAD_FO_SELECT_MIN") ? atoll(getenv("PRAD_FO_SELECT_MIN")) : (1LL << 20);
  if (dtype >= 2 || roi_count < select_from)
    return fail(PRAD_E_UNSUPPORTED, "firstorder queue: float images with %lld+ ROI voxels only (prad_firstorder_dev serves the rest)", select_from);
  hipStream_t s = (hipStream_t)stream;
  double *partial = nullptr, *gath = nullptr, *d_out = nullptr;
  unsigned *hist = nullptr, *cursors = nullptr;
  FoDev *st = nullptr;
  // gather capacity: half of the ROI, at least 2^18 elements (only memory: the selection's cost follows the real sizes of
  // the gathered bins; beyond the capacity -- nearly all of the ROI in the ten selected bins -- the verdict sends the
  // caller to the synchronous route)
  unsigned cap = (unsigned)std::min<long long>(1LL << 30, std::max<long long>(PRAD_FO_QUEUE_CAP, roi_count / 2));
  PRAD_TRY(c.get<double>("fo_partial", (size_t)PRAD_FO_BLOCKS * 8, &partial));
  PRAD_TRY(c.get<unsigned>("fo_hist_q", PRAD_FO_BINS + PRAD_FO_MAXSEL, &hist));     // [histogram | gather cursors]: one memset
  cursors = hist + PRAD_FO_BINS;
  PRAD_TRY(c.get<double>("fo_gather_q", (size_t)cap, &gath));
  PRAD_TRY(c.get<double>("fo_out_q", 16, &d_out));
  {
    void *p = nullptr;
    PRAD_TRY(c.get("fo_state", sizeof(FoDev), &p));
    st = (FoDev *)p;
  }
  PRAD_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned) * (PRAD_FO_BINS + PRAD_FO_MAXSEL), s));
  const int blocks = (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, PRAD_FO_BLOCKS));
  const unsigned hgx = (unsigned)std::max<long long>(1, std::min<long long>((n + 4095) / 4096, 512));
  {
    Timed t(c, "firstorder", s);
    FO_DISPATCH(fo_sums_kernel, dim3(blocks), dim3(256), voxelArrayShift, partial);
    hipLaunchKernelGGL(fo_glue_sums_kernel, dim3(1), dim3(256), 0, s, (const double *)partial, blocks, roi_count, st);
    FO_DISPATCH(fo_hist_dev_kernel, dim3(hgx), dim3(1024), (const FoDev *)st, hist);
    hipLaunchKernelGGL(fo_glue_select_kernel, dim3(1), dim3(1024), 0, s, (const unsigned *)hist, cap, st);
    FO_DISPATCH(fo_gather_dev_kernel, dim3(hgx), dim3(1024), (const FoDev *)st, cursors, gath);
    PRAD_TRY(check_launch("firstorder queue (gather)"));
    hipLaunchKernelGGL(fo_rank_select_kernel, dim3(10), dim3(1024), 0, s, (const double *)gath, st);
    hipLaunchKernelGGL(fo_glue_quantiles_kernel, dim3(1), dim3(64), 0, s, st);
    FO_DISPATCH(fo_central_dev_kernel, dim3(blocks), dim3(256), (const FoDev *)st, partial);
    hipLaunchKernelGGL(fo_glue_central_kernel, dim3(1), dim3(256), 0, s, (const double *)partial, blocks, st);
    FO_DISPATCH(fo_band_dev_kernel, dim3(blocks), dim3(256), (const FoDev *)st, partial);
    const bool direct = c.deferred && c.in_arena(out, sizeof(double) * 16) && Context::zero_copy();
    hipLaunchKernelGGL(fo_glue_final_kernel, dim3(1), dim3(256), 0, s, (const double *)partial, blocks, (const FoDev *)st,
                       direct ? out : d_out);
    PRAD_TRY(check_launch("firstorder queue"));
    if (direct) {
      c.last_path = "firstorder-queue";
      return PRAD_OK;
    }
  }
  c.last_path = "firstorder-queue";
  const size_t nb = sizeof(double) * 16;
  if (c.deferred && c.in_arena(out, nb)) {
    PRAD_HIP(hipMemcpyAsync(out, d_out, nb, hipMemcpyDeviceToHost, s));
    return PRAD_OK;
  }
  void *pin = nullptr;
  PRAD_TRY(c.get_pinned("fo_out_pin", nb, &pin));
  PRAD_HIP(hipMemcpyAsync(pin, d_out, nb, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  memcpy(out, pin, nb);
  if (out[15] != 0.0)
    return fail(PRAD_E_UNSUPPORTED, "firstorder queue: verdict %d (ROI count, constant ROI or gather capacity); use prad_firstorder_dev", (int)out[15]);
  return PRAD_OK;
}

extern "C" int prad_voxel_firstorder_dev(const void *image, int dtype, const uint8_t *mask, const int32_t *levels,
                                         const int *size, int Nd, int Nvox, const int *voxels, int kernelRadius,
                                         int force2Ddim, const int *bbsize, double voxelArrayShift, double voxelVolume,
                                         const int *feature_ids, int nfeat, double *out, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!image || !mask || !voxels || !out || !feature_ids || Nvox < 1 || nfeat < 1 || kernelRadius < 1)
    return fail(PRAD_E_ARG, "voxel_firstorder: bad arguments");
  for (int f = 0; f < nfeat; f++)
    if (feature_ids[f] < 0 || feature_ids[f] >= PRAD_FOF_COUNT) return fail(PRAD_E_ARG, "voxel_firstorder: feature id %d", feature_ids[f]);
  hipStream_t s = (hipStream_t)stream;
  FoWindow w;
  w.nd = Nd;
  long long nk = 1;
  for (int d = 0; d < Nd; d++) {
    int h = kernelRadius;
    if (bbsize) h = std::min(h, std::max(bbsize[d] - 1, 0));
    h = std::min(h, g.size[d] - 1);
    if (d == force2Ddim) h = 0;
    w.half[d] = h;
    nk *= 2 * h + 1;
  }
  if (nk > 4096) return fail(PRAD_E_UNSUPPORTED, "voxel_firstorder: %lld voxels per kernel exceed the 4096-slot window", nk);
  w.nk = (int)nk;
  w.P = 2;
  while (w.P < w.nk) w.P <<= 1;
  int *ids_d = nullptr;
  PRAD_TRY(c.get<int>("fo_ids", (size_t)nfeat, &ids_d));
  PRAD_HIP(hipMemcpyAsync(ids_d, feature_ids, sizeof(int) * nfeat, hipMemcpyHostToDevice, s));
  const size_t lds = (size_t)w.P * (sizeof(double) + sizeof(int));
  const unsigned gx = (unsigned)std::min<long long>(Nvox, 1 << 20);
  Timed t(c, "voxel_firstorder", s);
  switch (dtype) {
    case 0: hipLaunchKernelGGL(voxel_firstorder_kernel<float>, dim3(gx), dim3(64), lds, s, (const float *)image, mask, levels, g, w, Nvox, voxels, voxelArrayShift, voxelVolume, ids_d, nfeat, out); break;
    case 1: hipLaunchKernelGGL(voxel_firstorder_kernel<double>, dim3(gx), dim3(64), lds, s, (const double *)image, mask, levels, g, w, Nvox, voxels, voxelArrayShift, voxelVolume, ids_d, nfeat, out); break;
    case 2: hipLaunchKernelGGL(voxel_firstorder_kernel<int>, dim3(gx), dim3(64), lds, s, (const int *)image, mask, levels, g, w, Nvox, voxels, voxelArrayShift, voxelVolume, ids_d, nfeat, out); break;
    case 3: hipLaunchKernelGGL(voxel_firstorder_kernel<short>, dim3(gx), dim3(64), lds, s, (const short *)image, mask, levels, g, w, Nvox, voxels, voxelArrayShift, voxelVolume, ids_d, nfeat, out); break;
    default: return fail(PRAD_E_ARG, "voxel_firstorder: dtype %d", dtype);
  }
  PRAD_TRY(check_launch("voxel_firstorder_kernel"));
  c.last_path = "voxel-firstorder";
  return PRAD_OK;
}
