// kernels_voxel.h -- fused voxel-based GLCM feature maps.
//
// The reference's voxel mode materialises P[Nvox][Ng][Ng][Na] float64 (32-104 KiB per kernel) and then evaluates
// the feature formulas with numpy over the leading axis (glcm.py:145,208-887; base.py:200-245).  Here ONE WAVE per
// centre voxel builds the window's co-occurrence counts in a private LDS table and reduces them directly to the
// requested features; only 8 B per voxel and feature ever reach HBM.
//
// Per angle (glcm.py conventions: symmetrical matrix, per-angle normalisation, nanmean over non-empty angles):
//   pass A  lanes enumerate the window's voxel pairs (p, p+angle), both masked: integer LDS counts
//           tab[i][j] (+ tab[j][i]), row marginals, |i-j| and i+j histograms; wave sums of #pairs, sum i, sum j
//   pass B  lanes revisit their pairs: every term of a sum over matrix ENTRIES is produced once per pair
//           OCCURRENCE, divided by the entry's count, so no sweep over the (mostly empty) Ng x Ng table is needed;
//           moment-type features (linear in the per-angle sums once T, ux, uy are known) accumulate per lane across
//           angles and are reduced once per voxel; the few non-linear ones are reduced per angle
//   pass C  lanes zero exactly the table entries they touched
// Feature numbering: see VoxelGlcmFeature / pyradiomics_amd/engine.py.  MCC (an eigenvalue problem per kernel and
// angle, glcm.py:665-707) is not evaluated here; callers fall back to the matrix path for it.
#pragma once
#include "prad_runtime.h"
#include "kernels_sweep.h"

namespace prad {

enum VoxelGlcmFeature {
  VF_Autocorrelation = 0, VF_JointAverage, VF_ClusterProminence, VF_ClusterShade, VF_ClusterTendency, VF_Contrast,
  VF_Correlation, VF_DifferenceAverage, VF_DifferenceEntropy, VF_DifferenceVariance, VF_JointEnergy,
  VF_JointEntropy, VF_Imc1, VF_Imc2, VF_Idm, VF_Idmn, VF_Id, VF_Idn, VF_InverseVariance, VF_MaximumProbability,
  VF_SumAverage, VF_SumEntropy, VF_SumSquares, VF_COUNT
};

#define PRAD_VOX_MAX_ANGLES 32
struct VoxAngles {
  int na;
  signed char o[PRAD_VOX_MAX_ANGLES][4];
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

#define PRAD_VOX_WAVES 4  // waves (= centre voxels in flight) per workgroup

// out: [nfeat][nvox] float64; empty_mask[v]: bit a set <=> angle a had no pair in kernel v;
// any_nonempty[0]: OR over all kernels of the non-empty angle bits
// SM: the features this instantiation can produce (a compile-time mask: everything else -- accumulators, histograms, fp64
// terms -- is compiled out).  The all-features kernel needs 209 VGPRs, i.e. 2 waves per SIMD for a kernel that waits on LDS
// and cache round trips; the light one (joint entropy / energy / maximum / average: the usual voxel-map requests) fits
// several times the waves.
#define PRAD_VF_LIGHT ((1u << VF_JointEntropy) | (1u << VF_JointEnergy) | (1u << VF_MaximumProbability) | (1u << VF_JointAverage))
template <unsigned SM>
__global__ void __launch_bounds__(64 * PRAD_VOX_WAVES) voxel_glcm_kernel(
    const uint8_t *__restrict__ L, int Nz, int Ny, int Nx, VoxAngles A, int Ng, int nvox,
    const int *__restrict__ voxels, int vox_nd, int radius, int f2d3, int symmetric, unsigned feat_mask,
    const int *__restrict__ feat_slot, double *__restrict__ out, unsigned *__restrict__ empty_mask,
    unsigned *__restrict__ any_nonempty, const int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int per_wave = Ng * Ng + 2 * Ng + Ng + (2 * Ng + 1);
  u32 *tab = lds + (size_t)wave * per_wave;  // [Ng][Ng]
  u32 *rowx = tab + Ng * Ng;                 // [Ng]
  u32 *rowy = rowx + Ng;                     // [Ng]
  u32 *dif = rowy + Ng;                      // [Ng]      |i-j|
  u32 *sum = dif + Ng;                       // [2Ng+1]   i+j-2
  for (int i = lane; i < per_wave; i += 64) tab[i] = 0;
  const double eps = 2.220446049250313e-16;
  const bool sym = symmetric != 0;
  const int w = sym ? 2 : 1;
  const double Ngd = (double)Ng;
  feat_mask &= SM;
#define PRAD_WANT(f) (((SM >> (f)) & 1u) && ((feat_mask >> (f)) & 1u))
  const bool need_cov = PRAD_WANT(VF_Correlation);
  const bool need_dv = PRAD_WANT(VF_DifferenceVariance);
  const bool need_imc = PRAD_WANT(VF_Imc1) || PRAD_WANT(VF_Imc2);
  const bool need_imc2 = PRAD_WANT(VF_Imc2);
  const bool need_max = PRAD_WANT(VF_MaximumProbability);
  // what a request really needs (wave-uniform branches): a JointEntropy-only map -- the voxel-based benchmark -- skips the
  // marginal / difference / sum histograms and every other feature's fp64 terms (4 divisions and a dozen products per pair)
  const bool need_rows = need_imc;
  const bool need_dif = PRAD_WANT(VF_DifferenceEntropy), need_sum = PRAD_WANT(VF_SumEntropy);
  const bool need_moments = PRAD_WANT(VF_Autocorrelation) || PRAD_WANT(VF_ClusterTendency) || PRAD_WANT(VF_ClusterShade) ||
                            PRAD_WANT(VF_ClusterProminence) || PRAD_WANT(VF_SumSquares) || need_cov;
  const bool need_dterms = PRAD_WANT(VF_Contrast) || PRAD_WANT(VF_Idm) || PRAD_WANT(VF_Idmn) || PRAD_WANT(VF_Id) ||
                           PRAD_WANT(VF_Idn) || PRAD_WANT(VF_InverseVariance) || PRAD_WANT(VF_DifferenceAverage) ||
                           PRAD_WANT(VF_SumAverage) || need_dv;
  const bool need_plogp = PRAD_WANT(VF_JointEntropy) || need_imc;

  unsigned reported = 0;     // angle bits this wave has already OR-ed into *any_nonempty (lane 0)
  for (int v = blockIdx.x * PRAD_VOX_WAVES + wave; v < nvox; v += gridDim.x * PRAD_VOX_WAVES) {
    // centre and window box in the 3-D embedding
    int c[3] = {0, 0, 0};
    for (int d = 0; d < vox_nd; d++) c[3 - vox_nd + d] = voxels[(long long)d * nvox + v];
    const int dims[3] = {Nz, Ny, Nx};
    int lo[3], ext[3];
    for (int d = 0; d < 3; d++) {
      if (d == f2d3 || d < 3 - vox_nd) { lo[d] = c[d]; ext[d] = 1; }
      else {
        lo[d] = max(c[d] - radius, 0);
        ext[d] = min(c[d] + radius, dims[d] - 1) - lo[d] + 1;
      }
    }
    const int W = ext[0] * ext[1] * ext[2];
    // per-lane accumulators of angle-summed (already normalised) feature terms
    double acc[VF_COUNT];
#pragma unroll
    for (int f = 0; f < VF_COUNT; f++) acc[f] = 0.0;
    double u_corr = 0, u_dv = 0, u_imc1 = 0, u_imc2 = 0, u_max = 0, u_ja = 0;  // wave-uniform per-angle features
    int n_angles = 0, n_imc2 = 0;
    unsigned emask = 0;

    // Window voxels of this lane, decomposed and loaded ONCE per centre (windows of up to 128 voxels: every 2-D window,
    // 3^3 and 5^3); the three passes of every angle then only fetch the neighbour level.  The integer divisions of the
    // decomposition and the reloads were most of the kernel's instructions (a 5x5 window fills 25 of 64 lanes).
    constexpr int MAXIT = 2;
    const bool cached = W <= 64 * MAXIT;
    int ckx[MAXIT], cky[MAXIT], ckz[MAXIT], cli[MAXIT];
    if (cached) {
#pragma unroll
      for (int it = 0; it < MAXIT; it++) {
        const int k = lane + 64 * it;
        ckx[it] = k % ext[2];
        const int kr = k / ext[2];
        cky[it] = kr % ext[1];
        ckz[it] = kr / ext[1];
        cli[it] = k < W ? (int)L[((long long)(lo[0] + ckz[it]) * Ny + (lo[1] + cky[it])) * Nx + lo[2] + ckx[it]] : 0;
      }
    }
    for (int a = 0; a < A.na; a++) {
      const int dz = A.o[a][0], dy = A.o[a][1], dx = A.o[a][2];
      int clj[MAXIT];      // neighbour level of the cached voxels for this angle (0: no pair)
      if (cached) {
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
          const int qz = ckz[it] + dz, qy = cky[it] + dy, qx = ckx[it] + dx;
          const bool ok = cli[it] != 0 && (unsigned)qz < (unsigned)ext[0] && (unsigned)qy < (unsigned)ext[1] &&
                          (unsigned)qx < (unsigned)ext[2];
          clj[it] = ok ? (int)L[((long long)(lo[0] + qz) * Ny + (lo[1] + qy)) * Nx + lo[2] + qx] : 0;
        }
      }
      auto for_pairs = [&](auto body) __attribute__((always_inline)) {
        if (cached) {
#pragma unroll
          for (int it = 0; it < MAXIT; it++)
            if (clj[it]) body(cli[it], clj[it]);
        } else {
          for (int k = lane; k < W; k += 64) {
            const int kx = k % ext[2], kr = k / ext[2];
            const int ky = kr % ext[1], kz = kr / ext[1];
            const int qz = kz + dz, qy = ky + dy, qx = kx + dx;
            if ((unsigned)qz >= (unsigned)ext[0] || (unsigned)qy >= (unsigned)ext[1] || (unsigned)qx >= (unsigned)ext[2]) continue;
            const int li = L[((long long)(lo[0] + kz) * Ny + (lo[1] + ky)) * Nx + lo[2] + kx];
            if (!li) continue;
            const int lj = L[((long long)(lo[0] + qz) * Ny + (lo[1] + qy)) * Nx + lo[2] + qx];
            if (!lj) continue;
            body(li, lj);
          }
        }
      };
      // ---- pass A: integer counts ----
      int np = 0, si = 0, sj = 0;
      for_pairs([&](int li, int lj) __attribute__((always_inline)) {
        np++; si += li; sj += lj;
        atomicAdd(&tab[(li - 1) * Ng + (lj - 1)], 1u);
        if (sym) atomicAdd(&tab[(lj - 1) * Ng + (li - 1)], 1u);
        if (need_rows) {
          atomicAdd(&rowx[li - 1], 1u);
          if (sym) atomicAdd(&rowx[lj - 1], 1u);
          else atomicAdd(&rowy[lj - 1], 1u);
        }
        if (need_dif) atomicAdd(&dif[abs(li - lj)], (u32)w);
        if (need_sum) atomicAdd(&sum[li + lj - 2], (u32)w);
            });
      np = wave_sum_i32(np);
      if (np == 0) { emask |= 1u << a; continue; }   // nothing was written
      si = wave_sum_i32(si);
      sj = wave_sum_i32(sj);
      const double T = (double)(w * np), iT = 1.0 / T;
      const double ux = sym ? (double)(si + sj) / T : (double)si / T;
      const double uy = sym ? ux : (double)sj / T;
      const u32 *ry = sym ? rowx : rowy;
      n_angles++;
      u_ja += ux;
      // ---- histogram-based features (linear in the angle sum) ----
      if (need_dif || need_sum) {
        for (int k = lane; k < 2 * Ng + 1; k += 64) {
          if (k < Ng && need_dif) {
            const double pd = (double)dif[k] / T;
            acc[VF_DifferenceEntropy] -= pd * log2(pd + eps);
          }
          if (need_sum) {
            const double ps = (double)sum[k] / T;
            acc[VF_SumEntropy] -= ps * log2(ps + eps);
          }
        }
      }
      // ---- pass B ----
      double cov = 0, vx = 0, vy = 0, da = 0, d2 = 0, hxy = 0, hxy1 = 0, hx = 0, hy = 0, pmax = 0;
      for_pairs([&](int li, int lj) __attribute__((always_inline)) {
        const double cnt = (double)tab[(li - 1) * Ng + (lj - 1)];
        const double p = cnt * iT;
        const double occ = (double)w / cnt;            // this occurrence's share of its entry (both mirrored entries)
        if (PRAD_WANT(VF_JointEnergy)) acc[VF_JointEnergy] += occ * p * p;
        if (need_plogp) {
          const double plogp = p * log2(p + eps);
          acc[VF_JointEntropy] -= occ * plogp;
          hxy -= occ * plogp;
        }
        if (need_max) pmax = fmax(pmax, p);
        const double di = (double)li, dj = (double)lj, d = fabs(di - dj);
        // terms of sums over entries weighted by p: each ordered entry contributes 1/T per occurrence
        const int reps = (sym ? 2 : 1) * (need_moments ? 1 : 0);
        for (int r = 0; r < reps; r++) {
          const double ii = r ? dj : di, jj = r ? di : dj;
          const double s = ii + jj - ux - uy;
          acc[VF_Autocorrelation] += iT * ii * jj;
          acc[VF_ClusterTendency] += iT * s * s;
          acc[VF_ClusterShade] += iT * s * s * s;
          acc[VF_ClusterProminence] += iT * s * s * s * s;
          acc[VF_SumSquares] += iT * (ii - ux) * (ii - ux);
          cov += iT * (ii - ux) * (jj - uy);
          vx += iT * (ii - ux) * (ii - ux);
          vy += iT * (jj - uy) * (jj - uy);
        }
        const double wt = (double)w * iT;
        if (need_dterms) {
          acc[VF_Contrast] += wt * d * d;
          acc[VF_Idm] += wt / (1.0 + d * d);
          acc[VF_Idmn] += wt / (1.0 + (d * d) / (Ngd * Ngd));
          acc[VF_Id] += wt / (1.0 + d);
          acc[VF_Idn] += wt / (1.0 + d / Ngd);
          if (d != 0.0) acc[VF_InverseVariance] += wt / (d * d);
          acc[VF_DifferenceAverage] += wt * d;
          acc[VF_SumAverage] += wt * (di + dj);
          da += wt * d;
          d2 += wt * d * d;
        }
        if (need_imc) {
          const double pxi = (double)rowx[li - 1] * iT, pyj = (double)ry[lj - 1] * iT;
          hxy1 -= wt * log2(pxi * pyj + eps);
          if (sym) {
            const double pxj = (double)rowx[lj - 1] * iT;
            hx -= pxi * log2(pxi + eps) / (double)rowx[li - 1] + pxj * log2(pxj + eps) / (double)rowx[lj - 1];
          } else {
            hx -= pxi * log2(pxi + eps) / (double)rowx[li - 1];
            hy -= pyj * log2(pyj + eps) / (double)rowy[lj - 1];
          }
        }
            });
      // ---- per-angle non-linear features (wave-uniform after reduction) ----
      if (need_cov) {
        cov = wave_sum_f64(cov); vx = wave_sum_f64(vx); vy = wave_sum_f64(vy);
        const double sig = sqrt(vx) * sqrt(vy);
        u_corr += (sig == 0.0) ? 1.0 : cov / (sig + eps);
      }
      if (need_dv) {
        da = wave_sum_f64(da); d2 = wave_sum_f64(d2);
        u_dv += d2 - da * da;
      }
      if (need_max) u_max += wave_max_f64(pmax);
      if (need_imc) {
        hxy = wave_sum_f64(hxy); hxy1 = wave_sum_f64(hxy1); hx = wave_sum_f64(hx);
        hy = sym ? hx : wave_sum_f64(hy);
        const double div = fmax(hx, hy);
        u_imc1 += (div != 0.0) ? (hxy - hxy1) / div : 0.0;
        if (need_imc2) {
          double hxy2 = 0;
          for (int k = lane; k < Ng * Ng; k += 64) {
            const double pp = ((double)rowx[k / Ng] * iT) * ((double)ry[k % Ng] * iT);
            hxy2 -= pp * log2(pp + eps);
          }
          hxy2 = wave_sum_f64(hxy2);
          if (hxy2 == hxy) n_imc2++;                       // contributes 0 (glcm.py:644-645)
          else if (hxy2 > hxy) { u_imc2 += sqrt(1.0 - exp(-2.0 * (hxy2 - hxy))); n_imc2++; }
          // hxy2 < hxy: numpy yields NaN for this angle and nanmean drops it
        }
      }
      // ---- pass C: clear what this angle wrote ----
      for_pairs([&](int li, int lj) __attribute__((always_inline)) {
        tab[(li - 1) * Ng + (lj - 1)] = 0;
        tab[(lj - 1) * Ng + (li - 1)] = 0;
        if (need_rows) { rowx[li - 1] = 0; rowx[lj - 1] = 0; rowy[lj - 1] = 0; }
        if (need_dif) dif[abs(li - lj)] = 0;
        if (need_sum) sum[li + lj - 2] = 0;
            });
    }

    // ---- reduce, average over non-empty angles, store ----
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const double invA = n_angles ? 1.0 / (double)n_angles : nan;
#pragma unroll
    for (int f = 0; f < VF_COUNT; f++) {
      if (!PRAD_WANT(f)) continue;
      double val;
      if (f == VF_Correlation) val = u_corr * invA;
      else if (f == VF_DifferenceVariance) val = u_dv * invA;
      else if (f == VF_MaximumProbability) val = u_max * invA;
      else if (f == VF_Imc1) val = u_imc1 * invA;
      else if (f == VF_Imc2) val = (n_angles && n_imc2) ? u_imc2 / (double)n_imc2 : nan;
      else if (f == VF_JointAverage) val = u_ja * invA;
      else val = wave_sum_f64(acc[f]) * invA;
      if (lane == 0) out[(size_t)feat_slot[f] * nvox + v] = val;
    }
    if (lane == 0) {
      empty_mask[v] = emask;
      const unsigned nonempty = ~emask & (A.na >= 32 ? 0xffffffffu : ((1u << A.na) - 1u));
      // (once per NEW bit and wave: one atomic per centre on this single word serialised the whole map in L2 -- 16.7 M
      // same-address atomics took as long as the kernel's 190 ms)
      if (nonempty & ~reported) {
        atomicOr(any_nonempty, nonempty & ~reported);
        reported |= nonempty;
      }
    }
  }
#undef PRAD_WANT
}


// ---------------------------------------------------------------------------------------------------------------
// The usual voxel-map requests (JointEntropy / JointEnergy / MaximumProbability / JointAverage: PRAD_VF_LIGHT) on windows
// of at most 64 voxels (every 2-D window up to 7x7, 3^3): the general kernel above spent ~1 060 VALU instructions per
// centre on them, most of it fp64 log2 / division sequences and DPP reduction chains.  Here
//   * every sum over matrix ENTRIES is written per pair OCCURRENCE without a division: an entry of count c out of T pairs
//     is met c / w times, so  -sum_entries p log2(p + eps) = -(w / T) sum_occurrences log2(c / T + eps)  and
//     sum_entries p^2 = (w / T^2) sum_occurrences c;
//   * log2(c / T + eps) = LG[c] - LG[T] + eps T / (c ln 2) with LG[n] = log2(n) tabulated for n <= 256 (c and T are
//     small integers; the first-order term in eps is exact to ~1e-30);
//   * the pair count of an angle is a ballot popcount; one DPP reduction per requested feature at the very end;
//   * the window decomposition (integer divisions) is redone only when the window's extents change (the map's border).
// One lane per window voxel; bit-compatible with nothing (float sums in another order): within 1e-12 of the general kernel.
__global__ void __launch_bounds__(64 * PRAD_VOX_WAVES) voxel_glcm_light_kernel(
    const uint8_t *__restrict__ L, int Nz, int Ny, int Nx, VoxAngles A, int Ng, int nvox,
    const int *__restrict__ voxels, int vox_nd, int radius, int f2d3, int symmetric, unsigned feat_mask,
    const int *__restrict__ feat_slot, double *__restrict__ out, unsigned *__restrict__ empty_mask,
    unsigned *__restrict__ any_nonempty, const int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *LG = reinterpret_cast<double *>(lds);          // [257] log2(n), [0] unused
  double *RC = LG + 257;                                 // [257] 1 / n
  u32 *tab = reinterpret_cast<u32 *>(RC + 257) + (size_t)wave * Ng * Ng;
  for (int i = threadIdx.x; i < 257; i += blockDim.x) {
    LG[i] = i ? log2((double)i) : 0.0;
    RC[i] = i ? 1.0 / (double)i : 0.0;
  }
  for (int i = lane; i < Ng * Ng; i += 64) tab[i] = 0;
  __syncthreads();
  const bool sym = symmetric != 0;
  const int w = sym ? 2 : 1;
  const double eps_ln2 = 2.220446049250313e-16 / 0.6931471805599453;
  const bool want_ent = (feat_mask >> VF_JointEntropy) & 1u, want_en = (feat_mask >> VF_JointEnergy) & 1u;
  const bool want_max = (feat_mask >> VF_MaximumProbability) & 1u, want_ja = (feat_mask >> VF_JointAverage) & 1u;
  unsigned reported = 0;
  int pe0 = -1, pe1 = -1, pe2 = -1;      // extents the cached decomposition belongs to
  int kx = 0, ky = 0, kz = 0;
  for (int v = blockIdx.x * PRAD_VOX_WAVES + wave; v < nvox; v += gridDim.x * PRAD_VOX_WAVES) {
    int c[3] = {0, 0, 0};
    for (int d = 0; d < vox_nd; d++) c[3 - vox_nd + d] = voxels[(long long)d * nvox + v];
    const int dims[3] = {Nz, Ny, Nx};
    int lo[3], ext[3];
    for (int d = 0; d < 3; d++) {
      if (d == f2d3 || d < 3 - vox_nd) { lo[d] = c[d]; ext[d] = 1; }
      else {
        lo[d] = max(c[d] - radius, 0);
        ext[d] = min(c[d] + radius, dims[d] - 1) - lo[d] + 1;
      }
    }
    const int W = ext[0] * ext[1] * ext[2];
    if (ext[0] != pe0 || ext[1] != pe1 || ext[2] != pe2) {   // (wave-uniform)
      kx = lane % ext[2];
      const int kr = lane / ext[2];
      ky = kr % ext[1];
      kz = kr / ext[1];
      pe0 = ext[0]; pe1 = ext[1]; pe2 = ext[2];
    }
    const uint8_t *base = L + ((long long)lo[0] * Ny + lo[1]) * Nx + lo[2];
    const int li = lane < W ? (int)base[((long long)kz * Ny + ky) * Nx + kx] : 0;
    double a_ent = 0, a_en = 0, u_max = 0, u_ja = 0;
    int n_angles = 0;
    unsigned emask = 0;
    for (int a = 0; a < A.na; a++) {
      const int qz = kz + A.o[a][0], qy = ky + A.o[a][1], qx = kx + A.o[a][2];
      const bool inb = li != 0 && (unsigned)qz < (unsigned)ext[0] && (unsigned)qy < (unsigned)ext[1] && (unsigned)qx < (unsigned)ext[2];
      const int lj = inb ? (int)base[((long long)qz * Ny + qy) * Nx + qx] : 0;
      const bool pair = lj != 0;
      const int np = __popcll(__ballot(pair));
      if (np == 0) { emask |= 1u << a; continue; }
      const int e1 = (li - 1) * Ng + (lj - 1), e2 = (lj - 1) * Ng + (li - 1);
      if (pair) {
        atomicAdd(&tab[e1], 1u);
        if (sym) atomicAdd(&tab[e2], 1u);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int T = w * np;
      const double iT = RC[T];           // (T <= 2 * 64)
      n_angles++;
      const int cnt = pair ? (int)tab[e1] : 0;
      if (pair) {
        if (want_ent) a_ent -= ((double)w * iT) * (LG[cnt] - LG[T] + eps_ln2 * (double)T * RC[cnt]);
        if (want_en) a_en += ((double)w * iT * iT) * (double)cnt;
      }
      if (want_max) {
        int m = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        u_max += (double)m * iT;
      }
      if (want_ja) {
        int sij = pair ? (sym ? li + lj : li) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sij += __shfl_xor(sij, o);
        u_ja += (double)sij * iT;          // sym: (si + sj) / (2 np); else si / np
      }
      __builtin_amdgcn_wave_barrier();
      if (pair) {
        tab[e1] = 0;
        tab[e2] = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const double invA = n_angles ? 1.0 / (double)n_angles : nan;
    if (want_ent) {
      const double r = wave_sum_f64(a_ent) * invA;
      if (lane == 0) out[(size_t)feat_slot[VF_JointEntropy] * nvox + v] = r;
    }
    if (want_en) {
      const double r = wave_sum_f64(a_en) * invA;
      if (lane == 0) out[(size_t)feat_slot[VF_JointEnergy] * nvox + v] = r;
    }
    if (lane == 0) {
      if (want_max) out[(size_t)feat_slot[VF_MaximumProbability] * nvox + v] = u_max * invA;
      if (want_ja) out[(size_t)feat_slot[VF_JointAverage] * nvox + v] = u_ja * invA;
      empty_mask[v] = emask;
      const unsigned nonempty = ~emask & (A.na >= 32 ? 0xffffffffu : ((1u << A.na) - 1u));
      if (nonempty & ~reported) {
        atomicOr(any_nonempty, nonempty & ~reported);
        reported |= nonempty;
      }
    }
  }
}

}  // namespace prad
