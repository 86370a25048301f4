// kernels_features.h -- segment-mode feature formulas evaluated on the device matrices (SURVEY.md section 8f rank 1:
// "fused on-device feature evaluation ... not only in voxel mode").
//
//   glcm_matrix_features_kernel      one block per angle: the raw co-occurrence counts [Ng][Ng][Na] the sweeps produce
//                                    -> symmetrised, normalised p(i,j) and the 23 features of glcm.py:260-887 that are
//                                    plain sums (everything but MCC).  Marginals are built one row / column /
//                                    diagonal per thread, so the result does not depend on scheduling.
//   zone_matrix_features_kernel      one block per angle: a count matrix P[Ni][Nj] with level values i and size values
//                                    j (run lengths, zone sizes, dependence counts) -> the 16 features GLRLM
//                                    (glrlm.py:196-523), GLSZM (glszm.py:140-434) and GLDM (gldm.py:138-430) share.
// Both kernels read matrices of at most a few hundred KB that are L2-resident; they are latency-, not bandwidth-bound,
// and exist to keep the host out of the per-image loop (9 derived images x 5 classes per case).
#pragma once
#include "prad_runtime.h"

namespace prad {

enum { GF_Autocorrelation = 0, GF_JointAverage, GF_ClusterProminence, GF_ClusterShade, GF_ClusterTendency, GF_Contrast,
       GF_Correlation, GF_DifferenceAverage, GF_DifferenceEntropy, GF_DifferenceVariance, GF_JointEnergy,
       GF_JointEntropy, GF_Imc1, GF_Imc2, GF_Idm, GF_Idmn, GF_Id, GF_Idn, GF_InverseVariance, GF_MaximumProbability,
       GF_SumAverage, GF_SumEntropy, GF_SumSquares, GF_COUNT };

#ifndef PRAD_FEAT_THREADS
#define PRAD_FEAT_THREADS 256      // (1024 lanes per angle were measured: zone 40 -> 68 us, glcm 44 -> 161 us -- the kernels are
#endif                             // chains of block reductions, not loops over the entries)
#define PRAD_FEAT_WAVES (PRAD_FEAT_THREADS / 64)
#define PRAD_FEAT_EPS 2.220446049250313e-16

// deterministic block sum: per-wave shuffle tree, then the wave results in a fixed pairwise order
__device__ __forceinline__ double feat_block_sum(double v, double *shw) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = v;
  __syncthreads();
  double r[PRAD_FEAT_WAVES];
#pragma unroll
  for (int w = 0; w < PRAD_FEAT_WAVES; w++) r[w] = shw[w];
#pragma unroll
  for (int h = PRAD_FEAT_WAVES / 2; h > 0; h >>= 1)
#pragma unroll
    for (int w = 0; w < h; w++) r[w] = r[2 * w] + r[2 * w + 1];
  return r[0];
}
// N independent block sums behind ONE pair of barriers (each of these kernels is a chain of reductions: 16 of them, one
// after the other, were most of the 40 us of the zone kernel); shw: N * PRAD_FEAT_WAVES doubles; same order of additions
// per value as feat_block_sum
template <int N>
__device__ __forceinline__ void feat_block_sum_n(double (&v)[N], double *shw) {
#pragma unroll
  for (int k = 0; k < N; k++)
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N; k++) shw[k * PRAD_FEAT_WAVES + (threadIdx.x >> 6)] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; k++) {
    double r[PRAD_FEAT_WAVES];
#pragma unroll
    for (int w = 0; w < PRAD_FEAT_WAVES; w++) r[w] = shw[k * PRAD_FEAT_WAVES + w];
#pragma unroll
    for (int h = PRAD_FEAT_WAVES / 2; h > 0; h >>= 1)
#pragma unroll
      for (int w = 0; w < h; w++) r[w] = r[2 * w] + r[2 * w + 1];
    v[k] = r[0];
  }
}
__device__ __forceinline__ double feat_block_max(double v, double *shw) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = shw[0];
#pragma unroll
  for (int w = 1; w < PRAD_FEAT_WAVES; w++) r = fmax(r, shw[w]);
  return r;
}

// counts: [Ng][Ng][Na] float64 (reference layout).  out: [Na][GF_COUNT]; empty[a] = 1 when the angle has no pair.
__global__ void __launch_bounds__(PRAD_FEAT_THREADS) glcm_matrix_features_kernel(const double *__restrict__ counts,
                                                                                 int Ng, int Na, int symmetric,
                                                                                 double *__restrict__ out,
                                                                                 int *__restrict__ empty) {
#pragma clang fp contract(off)
  extern __shared__ double fs[];
  double *px = fs, *py = px + Ng, *psum = py + Ng, *pdif = psum + 2 * Ng, *sh4 = pdif + Ng;   // sizes Ng, Ng, 2Ng, Ng, 4
  const int a = blockIdx.x, t = threadIdx.x;
  const double eps = PRAD_FEAT_EPS;
  auto C = [&](int i, int j) -> double {
    const double v = counts[((size_t)i * Ng + j) * Na + a];
    return symmetric ? v + counts[((size_t)j * Ng + i) * Na + a] : v;
  };
  const int n2 = Ng * Ng;
  double tot = 0;
  for (int e = t; e < n2; e += PRAD_FEAT_THREADS) tot += C(e / Ng, e % Ng);
  tot = feat_block_sum(tot, sh4);
  if (tot == 0) {
    if (t == 0) empty[a] = 1;
    for (int f = t; f < GF_COUNT; f += PRAD_FEAT_THREADS) out[(size_t)a * GF_COUNT + f] = __builtin_nan("");
    return;
  }
  if (t == 0) empty[a] = 0;
  // marginals: one row / column / (anti-)diagonal per thread, in index order
  for (int i = t; i < Ng; i += PRAD_FEAT_THREADS) {
    double r = 0, c = 0;
    for (int j = 0; j < Ng; j++) {
      r += C(i, j) / tot;
      c += C(j, i) / tot;
    }
    px[i] = r;
    py[i] = c;
  }
  for (int k = t; k < 2 * Ng - 1; k += PRAD_FEAT_THREADS) {       // i + j = k + 2 (levels are 1-based)
    double s = 0;
    for (int i = max(0, k - (Ng - 1)); i <= min(k, Ng - 1); i++) s += C(i, k - i) / tot;
    psum[k] = s;
  }
  for (int k = t; k < Ng; k += PRAD_FEAT_THREADS) {               // |i - j| = k
    double s = 0;
    for (int i = 0; i + k < Ng; i++) s += (k == 0 ? C(i, i) : C(i, i + k) + C(i + k, i)) / tot;
    pdif[k] = s;
  }
  __syncthreads();
  // pass 1 over the entries: means, entropy, energy, maximum, autocorrelation, contrast
  double ux = 0, uy = 0, hxy = 0, energy = 0, pmax = 0, autoc = 0, contrast = 0;
  for (int e = t; e < n2; e += PRAD_FEAT_THREADS) {
    const int i0 = e / Ng, j0 = e % Ng;
    const double p = C(i0, j0) / tot, i = i0 + 1, j = j0 + 1;
    ux += i * p;
    uy += j * p;
    hxy += p * log2(p + eps);
    energy += p * p;
    pmax = fmax(pmax, p);
    autoc += p * (i * j);
    const double d = fabs(i - j);
    contrast += p * (d * d);
  }
  ux = feat_block_sum(ux, sh4);
  uy = feat_block_sum(uy, sh4);
  hxy = -feat_block_sum(hxy, sh4);
  energy = feat_block_sum(energy, sh4);
  pmax = feat_block_max(pmax, sh4);
  autoc = feat_block_sum(autoc, sh4);
  contrast = feat_block_sum(contrast, sh4);
  // pass 2: moments about the means, information measures
  double cp = 0, cs = 0, ct = 0, vx = 0, vy = 0, cov = 0, hxy1 = 0, hxy2 = 0;
  for (int e = t; e < n2; e += PRAD_FEAT_THREADS) {
    const int i0 = e / Ng, j0 = e % Ng;
    const double p = C(i0, j0) / tot, i = i0 + 1, j = j0 + 1;
    const double s = (i + j) - ux - uy, s2 = s * s;
    ct += p * s2;
    cs += p * (s2 * s);
    cp += p * (s2 * s2);
    const double di = i - ux, dj = j - uy;
    vx += p * (di * di);
    vy += p * (dj * dj);
    cov += p * di * dj;
    const double q = px[i0] * py[j0];
    hxy1 += p * log2(q + eps);
    hxy2 += q * log2(q + eps);
  }
  cp = feat_block_sum(cp, sh4);
  cs = feat_block_sum(cs, sh4);
  ct = feat_block_sum(ct, sh4);
  vx = feat_block_sum(vx, sh4);
  vy = feat_block_sum(vy, sh4);
  cov = feat_block_sum(cov, sh4);
  hxy1 = -feat_block_sum(hxy1, sh4);
  hxy2 = -feat_block_sum(hxy2, sh4);
  // marginal-based sums
  double hx = 0, hy = 0;
  for (int i = t; i < Ng; i += PRAD_FEAT_THREADS) {
    hx += px[i] * log2(px[i] + eps);
    hy += py[i] * log2(py[i] + eps);
  }
  hx = -feat_block_sum(hx, sh4);
  hy = -feat_block_sum(hy, sh4);
  double da = 0, de = 0, idm = 0, idmn = 0, id = 0, idn = 0, iv = 0;
  const double Ngd = (double)Ng;
  for (int k = t; k < Ng; k += PRAD_FEAT_THREADS) {
    const double p = pdif[k], kd = (double)k;
    da += kd * p;
    de += p * log2(p + eps);
    idm += p / (1 + kd * kd);
    idmn += p / (1 + (kd * kd) / (Ngd * Ngd));
    id += p / (1 + kd);
    idn += p / (1 + kd / Ngd);
    if (k >= 1) iv += p / (kd * kd);
  }
  da = feat_block_sum(da, sh4);
  de = -feat_block_sum(de, sh4);
  idm = feat_block_sum(idm, sh4);
  idmn = feat_block_sum(idmn, sh4);
  id = feat_block_sum(id, sh4);
  idn = feat_block_sum(idn, sh4);
  iv = feat_block_sum(iv, sh4);
  double dv = 0;
  for (int k = t; k < Ng; k += PRAD_FEAT_THREADS) {
    const double d = (double)k - da;
    dv += pdif[k] * (d * d);
  }
  dv = feat_block_sum(dv, sh4);
  double sa = 0, se = 0;
  for (int k = t; k < 2 * Ng - 1; k += PRAD_FEAT_THREADS) {
    const double p = psum[k];
    sa += (double)(k + 2) * p;
    se += p * log2(p + eps);
  }
  sa = feat_block_sum(sa, sh4);
  se = -feat_block_sum(se, sh4);
  if (t == 0) {
    double *o = out + (size_t)a * GF_COUNT;
    const double sigx = sqrt(vx), sigy = sqrt(vy);
    o[GF_Autocorrelation] = autoc;
    o[GF_JointAverage] = ux;
    o[GF_ClusterProminence] = cp;
    o[GF_ClusterShade] = cs;
    o[GF_ClusterTendency] = ct;
    o[GF_Contrast] = contrast;
    o[GF_Correlation] = (sigx * sigy == 0) ? 1.0 : cov / (sigx * sigy + eps);       // glcm.py:409-410
    o[GF_DifferenceAverage] = da;
    o[GF_DifferenceEntropy] = de;
    o[GF_DifferenceVariance] = dv;
    o[GF_JointEnergy] = energy;
    o[GF_JointEntropy] = hxy;
    const double div = fmax(hx, hy);
    o[GF_Imc1] = div != 0 ? (hxy - hxy1) / div : 0.0;                                // :605-610
    o[GF_Imc2] = (hxy2 == hxy) ? 0.0 : sqrt(1 - exp(-2 * (hxy2 - hxy)));            // :641-647 (NaN when negative)
    o[GF_Idm] = idm;
    o[GF_Idmn] = idmn;
    o[GF_Id] = id;
    o[GF_Idn] = idn;
    o[GF_InverseVariance] = iv;
    o[GF_MaximumProbability] = pmax;
    o[GF_SumAverage] = sa;
    o[GF_SumEntropy] = se;
    o[GF_SumSquares] = vx;
  }
}

// ---- GLRLM / GLSZM / GLDM ------------------------------------------------------------------------------------
enum { ZM_SmallEmphasis = 0, ZM_LargeEmphasis, ZM_GrayLevelNonUniformity, ZM_GrayLevelNonUniformityNormalized,
       ZM_SizeNonUniformity, ZM_SizeNonUniformityNormalized, ZM_Percentage, ZM_GrayLevelVariance, ZM_SizeVariance,
       ZM_Entropy, ZM_LowGrayLevelEmphasis, ZM_HighGrayLevelEmphasis, ZM_SmallLowGrayLevelEmphasis,
       ZM_SmallHighGrayLevelEmphasis, ZM_LargeLowGrayLevelEmphasis, ZM_LargeHighGrayLevelEmphasis, ZM_COUNT };

static_assert(ZM_COUNT == 16, "prad_api.hip (ZM_FEATURES) and include/pyradiomics_amd.h say 16 zone-matrix features");
// P(i, j, a) = counts[i * si + j * sj + a * sa]; level value = i + 1; size value = jvals[j] (jvals == NULL: j + 1).
// scratch: [Na][Ni + Nj] float64 (marginals).  out: [Na][ZM_COUNT]; empty[a] = 1 when the matrix of angle a is all zero.
__global__ void __launch_bounds__(PRAD_FEAT_THREADS) zone_matrix_features_kernel(
    const double *__restrict__ counts, int Ni, int Nj, int Na, long long si, long long sj, long long sa,
    const double *__restrict__ jvals, double *__restrict__ scratch, double *__restrict__ out, int *__restrict__ empty,
    const int *__restrict__ nj_dev = nullptr) {
#pragma clang fp contract(off)
  __shared__ double sh4[PRAD_FEAT_WAVES];
  __shared__ double shn[13 * PRAD_FEAT_WAVES];
  const int a = blockIdx.x, t = threadIdx.x;
  if (nj_dev) Nj = min(Nj, nj_dev[0]);      // (the column count was found on the device: glszm_rank_kernel; Nj = capacity)
  const double eps = PRAD_FEAT_EPS;
  double *pg = scratch + (size_t)a * (Ni + Nj), *pj = pg + Ni;
  auto P = [&](int i, int j) -> double { return counts[i * si + j * sj + a * sa]; };
  // row sums: one wave per level row, lanes stride over the sizes, shuffle tree (a GLSZM row has thousands of columns:
  // one thread per row was 100+ us of dependent loads); the order of the additions is fixed, so runs reproduce
  for (int i = t >> 6; i < Ni; i += PRAD_FEAT_THREADS >> 6) {
    double s = 0;
    for (int j = t & 63; j < Nj; j += 64) s += P(i, j);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((t & 63) == 0) pg[i] = s;
  }
  for (int j = t; j < Nj; j += PRAD_FEAT_THREADS) {
    double s = 0;
    int i = 0;
    for (; i + 8 <= Ni; i += 8) {            // eight loads in flight, added in row order
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = P(i + u, j);
#pragma unroll
      for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; i < Ni; i++) s += P(i, j);
    pj[j] = s;
  }
  __syncthreads();
  double n = 0;
  for (int i = t; i < Ni; i += PRAD_FEAT_THREADS) n += pg[i];
  n = feat_block_sum(n, sh4);
  if (n == 0) {
    if (t == 0) empty[a] = 1;
    for (int f = t; f < ZM_COUNT; f += PRAD_FEAT_THREADS) out[(size_t)a * ZM_COUNT + f] = __builtin_nan("");
    return;
  }
  if (t == 0) empty[a] = 0;
  double i1 = 0, i2 = 0, inv_i2 = 0, mg = 0;
  for (int i = t; i < Ni; i += PRAD_FEAT_THREADS) {
    const double g = pg[i], iv = i + 1;
    i1 += g * iv;
    i2 += g * (iv * iv);
    inv_i2 += g / (iv * iv);
    mg += g * g;
  }
  double j1 = 0, j2 = 0, inv_j2 = 0, mj = 0;
  for (int j = t; j < Nj; j += PRAD_FEAT_THREADS) {
    const double s = pj[j], jv = jvals ? jvals[j] : (double)(j + 1);
    j1 += s * jv;
    j2 += s * (jv * jv);
    inv_j2 += s / (jv * jv);
    mj += s * s;
  }
  // the entries, spread over all lanes (a lane that walks a whole column makes the kernel as long as its log2 / division
  // chain: 32 rows x ~300 instructions); 1 / n once, 1 / i^2 from a table
  double ent = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
  const double rn = 1.0 / n;
  const long long nij = (long long)Ni * Nj;
  for (long long e = t; e < nij; e += PRAD_FEAT_THREADS) {
    const int i = (int)(e / Nj), j = (int)(e - (long long)i * Nj);
    const double v = P(i, j);
    if (v == 0) continue;
    const double iv = i + 1, jv = jvals ? jvals[j] : (double)(j + 1), i2v = iv * iv, j2v = jv * jv, ri2 = 1.0 / i2v, rj2 = 1.0 / j2v, p = v * rn;
    ent += p * log2(p + eps);
    c1 += v * (ri2 * rj2);
    c2 += v * (i2v * rj2);
    c3 += v * (j2v * ri2);
    c4 += v * (i2v * j2v);
  }
  {
    double r[13] = {i1, i2, inv_i2, mg, j1, j2, inv_j2, mj, ent, c1, c2, c3, c4};
    feat_block_sum_n(r, shn);
    i1 = r[0]; i2 = r[1]; inv_i2 = r[2]; mg = r[3]; j1 = r[4]; j2 = r[5]; inv_j2 = r[6]; mj = r[7];
    ent = -r[8]; c1 = r[9]; c2 = r[10]; c3 = r[11]; c4 = r[12];
  }
  const double ui = i1 / n, uj = j1 / n;
  double vi = 0, vj = 0;
  for (int i = t; i < Ni; i += PRAD_FEAT_THREADS) {
    const double d = (double)(i + 1) - ui;
    vi += (pg[i] / n) * (d * d);
  }
  for (int j = t; j < Nj; j += PRAD_FEAT_THREADS) {
    const double d = (jvals ? jvals[j] : (double)(j + 1)) - uj;
    vj += (pj[j] / n) * (d * d);
  }
  {
    double r[2] = {vi, vj};
    feat_block_sum_n(r, shn);
    vi = r[0]; vj = r[1];
  }
  if (t == 0) {
    double *o = out + (size_t)a * ZM_COUNT;
    o[ZM_SmallEmphasis] = inv_j2 / n;
    o[ZM_LargeEmphasis] = j2 / n;
    o[ZM_GrayLevelNonUniformity] = mg / n;
    o[ZM_GrayLevelNonUniformityNormalized] = mg / (n * n);
    o[ZM_SizeNonUniformity] = mj / n;
    o[ZM_SizeNonUniformityNormalized] = mj / (n * n);
    o[ZM_Percentage] = n / j1;
    o[ZM_GrayLevelVariance] = vi;
    o[ZM_SizeVariance] = vj;
    o[ZM_Entropy] = ent;
    o[ZM_LowGrayLevelEmphasis] = inv_i2 / n;
    o[ZM_HighGrayLevelEmphasis] = i2 / n;
    o[ZM_SmallLowGrayLevelEmphasis] = c1 / n;
    o[ZM_SmallHighGrayLevelEmphasis] = c2 / n;
    o[ZM_LargeLowGrayLevelEmphasis] = c3 / n;
    o[ZM_LargeHighGrayLevelEmphasis] = c4 / n;
  }
}

// ---- NGTDM (ngtdm.py:133-287): P[Ng][3] = (n_i, s_i, level) -> Coarseness, Contrast, Busyness, Complexity, Strength ----
__global__ void __launch_bounds__(PRAD_FEAT_THREADS) ngtdm_matrix_features_kernel(const double *__restrict__ P, int Ng,
                                                                                  double *__restrict__ out) {
#pragma clang fp contract(off)
  extern __shared__ double ns[];
  double *pi = ns, *si = pi + Ng, *lv = si + Ng, *sh4 = lv + Ng;     // present levels, compacted in level order
  __shared__ int ngp_s;
  const int t = threadIdx.x;
  if (t == 0) {                       // compaction in level order (Ng <= a few hundred: serial is fine)
    int k = 0;
    for (int g = 0; g < Ng; g++)
      if (P[g * 3] > 0) {
        pi[k] = P[g * 3];
        si[k] = P[g * 3 + 1];
        lv[k] = P[g * 3 + 2];
        k++;
      }
    ngp_s = k;
  }
  __syncthreads();
  const int ngp = ngp_s;
  double nvp = 0, stot = 0;
  for (int k = t; k < ngp; k += PRAD_FEAT_THREADS) {
    nvp += pi[k];
    stot += si[k];
  }
  nvp = feat_block_sum(nvp, sh4);
  stot = feat_block_sum(stot, sh4);
  __syncthreads();
  for (int k = t; k < ngp; k += PRAD_FEAT_THREADS) pi[k] = pi[k] / nvp;
  __syncthreads();
  double coarse = 0, contrast = 0, absdiff = 0, complexity = 0, strength = 0;
  for (int k = t; k < ngp; k += PRAD_FEAT_THREADS) coarse += pi[k] * si[k];
  coarse = feat_block_sum(coarse, sh4);
  for (int e = t; e < ngp * ngp; e += PRAD_FEAT_THREADS) {
    const int a = e / ngp, b = e % ngp;
    const double pa = pi[a], pb = pi[b], d = lv[a] - lv[b];
    contrast += pa * pb * (d * d);
    absdiff += fabs(lv[a] * pa - lv[b] * pb);
    complexity += fabs(d) * (pa * si[a] + pb * si[b]) / (pa + pb);
    strength += (pa + pb) * (d * d);
  }
  contrast = feat_block_sum(contrast, sh4);
  absdiff = feat_block_sum(absdiff, sh4);
  complexity = feat_block_sum(complexity, sh4);
  strength = feat_block_sum(strength, sh4);
  if (t == 0) {
    const double div = (double)ngp * (double)(ngp - 1);
    out[0] = coarse != 0 ? 1.0 / coarse : 1e6;                                    // ngtdm.py:148-150
    out[1] = div != 0 ? contrast * stot / nvp / div : 0.0;                        // :187-188
    out[2] = absdiff != 0 ? coarse / absdiff : 0.0;                               // :219-220
    out[3] = nvp != 0 ? complexity / nvp : __builtin_nan("");
    out[4] = stot != 0 ? strength / stot : 0.0;                                   // :284-285
  }
}

}  // namespace prad
