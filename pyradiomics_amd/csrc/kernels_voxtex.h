// kernels_voxtex.h -- fused voxel-based feature maps for GLRLM, GLSZM, GLDM and NGTDM.
//
// The reference's voxel mode materialises one matrix per kernel window -- P[Nvox][Ng][Nr][Na], [Nvox][Ng][maxRegion],
// [Nvox][Ng][2Na+1], [Nvox][Ng][3] float64 -- and evaluates the feature formulas with numpy over the leading axis
// (glrlm.py:99-523, glszm.py:84-434, gldm.py:86-430, ngtdm.py:97-287).  Here ONE WAVE per centre voxel gathers the
// window's packed levels into LDS, derives the window's item list and reduces it to the requested feature values:
//     GLDM   items = ROI voxels of the window          (i = level, j = dependence count + 1)
//     GLRLM  items = runs along one angle              (i = level, j = run length), per angle, then the mean over
//                                                      the non-empty angles (np.nanmean, e.g. glrlm.py:225)
//     GLSZM  items = zones (connected components)      (i = level, j = zone size)
// Every feature of these three classes is a function of the same sums over items (zl_accumulate): sums over matrix
// ENTRIES such as sum_i pg(i)^2 or -sum p log2 p are produced per item through the item's multiplicity (number of
// items sharing its level / its j / both), so the mostly empty matrix is never formed.
//     NGTDM  per-level n_i and s_i (s_i summed in raster order like cmatrices.c:637-652, hence bit-identical),
//            then the pairwise level formulas.
// Only 8 B per voxel and feature leave the chip.
#pragma once
#include "prad_runtime.h"
#include "kernels_voxel.h"

namespace prad {

#define PRAD_VT_WAVES 4
#define PRAD_VT_MAXW 512          // window voxels per kernel (e.g. 7^3 = 343)

enum { PRAD_VT_GLDM = 1, PRAD_VT_NGTDM = 2, PRAD_VT_GLRLM = 3, PRAD_VT_GLSZM = 4 };

struct VtWindow {
  int lo[3], ext[3], W;
};

// window of centre v (set_bb, _cmatrices.c:1120-1147: centre +- radius clamped to the array, collapsed along the
// force2D dimension) and its packed levels (0 = outside the ROI) -> wl[0..W)
__device__ __forceinline__ VtWindow vt_load_window(const uint8_t *__restrict__ L, int Nz, int Ny, int Nx, int nvox,
                                                   const int *__restrict__ voxels, int vox_nd, int radius, int f2d3,
                                                   int v, int lane, int *wl) {
  VtWindow w;
  int c[3] = {0, 0, 0};
  for (int d = 0; d < vox_nd; d++) c[3 - vox_nd + d] = voxels[(long long)d * nvox + v];
  const int dims[3] = {Nz, Ny, Nx};
  for (int d = 0; d < 3; d++) {
    if (d == f2d3 || d < 3 - vox_nd) { w.lo[d] = c[d]; w.ext[d] = 1; }
    else {
      w.lo[d] = max(c[d] - radius, 0);
      w.ext[d] = min(c[d] + radius, dims[d] - 1) - w.lo[d] + 1;
    }
  }
  w.W = w.ext[0] * w.ext[1] * w.ext[2];
  for (int k = lane; k < w.W; k += 64) {
    const int kx = k % w.ext[2], kr = k / w.ext[2];
    const int ky = kr % w.ext[1], kz = kr / w.ext[1];
    wl[k] = L[((long long)(w.lo[0] + kz) * Ny + (w.lo[1] + ky)) * Nx + w.lo[2] + kx];
  }
  return w;
}
// window-local index of voxel k displaced by offset o, or -1 when it leaves the window
__device__ __forceinline__ int vt_shift(const VtWindow &w, int k, const signed char *o) {
  const int kx = k % w.ext[2], kr = k / w.ext[2];
  const int ky = kr % w.ext[1], kz = kr / w.ext[1];
  const int qz = kz + o[0], qy = ky + o[1], qx = kx + o[2];
  if ((unsigned)qz >= (unsigned)w.ext[0] || (unsigned)qy >= (unsigned)w.ext[1] || (unsigned)qx >= (unsigned)w.ext[2]) return -1;
  return (qz * w.ext[1] + qy) * w.ext[2] + qx;
}
// The same without the two integer divisions per call (a GLDM voxel asks for 26 neighbours, a GLRLM run walks its line
// step by step): the voxel's coordinates once, then adds and bounds tests.
struct VtPos {
  int x, y, z;
};
__device__ __forceinline__ VtPos vt_pos(const VtWindow &w, int k) {
  VtPos p;
  p.x = k % w.ext[2];
  const int kr = k / w.ext[2];
  p.y = kr % w.ext[1];
  p.z = kr / w.ext[1];
  return p;
}
// index of p displaced by m * o, or -1 outside the window
__device__ __forceinline__ int vt_at(const VtWindow &w, const VtPos &p, const signed char *o, int m = 1) {
  const int qz = p.z + m * o[0], qy = p.y + m * o[1], qx = p.x + m * o[2];
  if ((unsigned)qz >= (unsigned)w.ext[0] || (unsigned)qy >= (unsigned)w.ext[1] || (unsigned)qx >= (unsigned)w.ext[2]) return -1;
  return (qz * w.ext[1] + qy) * w.ext[2] + qx;
}

// ---- features of an item list (i, j) -----------------------------------------------------------------------
// feature numbering shared by GLRLM / GLSZM / GLDM (names in pyradiomics_amd/cmatrices.py)
enum { ZF_SmallEmphasis = 0, ZF_LargeEmphasis, ZF_GrayLevelNonUniformity, ZF_GrayLevelNonUniformityNormalized,
       ZF_SizeNonUniformity, ZF_SizeNonUniformityNormalized, ZF_Percentage, ZF_GrayLevelVariance, ZF_SizeVariance,
       ZF_Entropy, ZF_LowGrayLevelEmphasis, ZF_HighGrayLevelEmphasis, ZF_SmallLowGrayLevelEmphasis,
       ZF_SmallHighGrayLevelEmphasis, ZF_LargeLowGrayLevelEmphasis, ZF_LargeHighGrayLevelEmphasis, ZF_COUNT };

// tables in global memory (12 KB, cache-resident; in LDS they cost the kernel two of its five workgroups per CU):
// n = 0 .. PRAD_VT_MAXW  ->  log2(n), 1 / n, 1 / n^2
#define PRAD_VT_TAB (PRAD_VT_MAXW + 1)
struct ZlTables {
  const double *lg, *rc, *rc2;
};

// Adds the 16 features of one item list (one angle of GLRLM; the whole window for GLDM / GLSZM) to acc[] -- as PER-LANE
// partial sums: every feature but Percentage is (a sum over items) / n or / n^2, so the lane's terms are scaled by the
// list's 1 / n right away and ONE wave reduction per feature at the end of the centre serves all lists (13 angles used to
// mean 13 x 14 fp64 DPP chains and 13 x 16 divisions by every lane).  Only the two means need a reduction per list, and
// they are integer sums.  1 / i^2, 1 / j^2 come from a table; log2(m / n + eps) = lg[m] - lg[n] + eps n / (m ln 2) (m, n
// small integers; first order in eps is exact to 1e-30).  Percentage (n / sum j) is added on lane 0.
__device__ __forceinline__ void zl_accumulate(const int *it_i, const int *it_j, int n, int lane, const ZlTables &T,
                                              double (&acc)[ZF_COUNT]) {
  const double nd = (double)n, inv_n = T.rc[n], inv_n2 = inv_n * inv_n;
  const double eps_n_ln2 = (2.220446049250313e-16 / 0.6931471805599453) * nd;
  int si = 0, sj = 0;
  for (int k = lane; k < n; k += 64) {
    si += it_i[k];
    sj += it_j[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    si += __shfl_xor(si, o);
    sj += __shfl_xor(sj, o);
  }
  const double ui = (double)si / nd, uj = (double)sj / nd;
  for (int k = lane; k < n; k += 64) {
    const int ii = it_i[k], jj = it_j[k];
    const double i = (double)ii, j = (double)jj, i2 = i * i, j2 = j * j, ri = T.rc2[ii], rj = T.rc2[jj];
    int same_i = 0, same_j = 0, same_ij = 0;
    for (int q = 0; q < n; q++) {
      const bool ei = it_i[q] == ii, ej = it_j[q] == jj;
      same_i += ei;
      same_j += ej;
      same_ij += ei && ej;
    }
    const double di = i - ui, dj = j - uj;
    acc[ZF_SmallEmphasis] += rj * inv_n;
    acc[ZF_LargeEmphasis] += j2 * inv_n;
    acc[ZF_GrayLevelNonUniformity] += (double)same_i * inv_n;
    acc[ZF_GrayLevelNonUniformityNormalized] += (double)same_i * inv_n2;
    acc[ZF_SizeNonUniformity] += (double)same_j * inv_n;
    acc[ZF_SizeNonUniformityNormalized] += (double)same_j * inv_n2;
    acc[ZF_GrayLevelVariance] += di * di * inv_n;
    acc[ZF_SizeVariance] += dj * dj * inv_n;
    acc[ZF_Entropy] -= (T.lg[same_ij] - T.lg[n] + eps_n_ln2 * T.rc[same_ij]) * inv_n;
    acc[ZF_LowGrayLevelEmphasis] += ri * inv_n;
    acc[ZF_HighGrayLevelEmphasis] += i2 * inv_n;
    acc[ZF_SmallLowGrayLevelEmphasis] += ri * rj * inv_n;
    acc[ZF_SmallHighGrayLevelEmphasis] += i2 * rj * inv_n;
    acc[ZF_LargeLowGrayLevelEmphasis] += j2 * ri * inv_n;
    acc[ZF_LargeHighGrayLevelEmphasis] += i2 * j2 * inv_n;
  }
  if (lane == 0) acc[ZF_Percentage] += nd / (double)sj;
}

// append `flag` lanes' (i, j) to the item list; returns the new length (wave-uniform)
__device__ __forceinline__ int vt_append(int *it_i, int *it_j, int n, bool flag, int i, int j, int lane) {
  const unsigned long long B = __ballot(flag);
  if (flag) {
    const int pos = n + __popcll(B & ((1ull << lane) - 1ull));
    it_i[pos] = i;
    it_j[pos] = j;
  }
  return n + __popcll(B);
}

// LDS per wave: wl[MAXW] | it_i[MAXW] | it_j[MAXW] | aux[MAXW]
#define PRAD_VT_LDS_PER_WAVE (4 * PRAD_VT_MAXW)

// family: GLDM / GLRLM / GLSZM.  out[f][v]
__global__ void __launch_bounds__(64 * PRAD_VT_WAVES) voxel_zonelike_kernel(
    int family, const uint8_t *__restrict__ L, int Nz, int Ny, int Nx, VoxAngles A, int alpha, int nvox,
    const int *__restrict__ voxels, int vox_nd, int radius, int f2d3, const int *__restrict__ feature_ids, int nfeat,
    double *__restrict__ out, const int *__restrict__ flags, const double *__restrict__ tabs) {
  __shared__ int lds[PRAD_VT_WAVES * PRAD_VT_LDS_PER_WAVE];
  if (flags[0]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const ZlTables T{tabs, tabs + PRAD_VT_TAB, tabs + 2 * PRAD_VT_TAB};
  int *wl = lds + wave * PRAD_VT_LDS_PER_WAVE, *it_i = wl + PRAD_VT_MAXW, *it_j = it_i + PRAD_VT_MAXW,
      *aux = it_j + PRAD_VT_MAXW;
  for (int v = blockIdx.x * PRAD_VT_WAVES + wave; v < nvox; v += gridDim.x * PRAD_VT_WAVES) {
    const VtWindow w = vt_load_window(L, Nz, Ny, Nx, nvox, voxels, vox_nd, radius, f2d3, v, lane, wl);
    __builtin_amdgcn_wave_barrier();
    double acc[ZF_COUNT];
#pragma unroll
    for (int f = 0; f < ZF_COUNT; f++) acc[f] = 0.0;
    int groups = 0;                                   // item lists that contributed (angles for GLRLM, else 0/1)
    if (family == PRAD_VT_GLDM) {
      int n = 0;
      for (int k0 = 0; k0 < w.W; k0 += 64) {
        const int k = k0 + lane;
        const int lv = k < w.W ? wl[k] : 0;
        int dep = 0;
        if (lv) {
          const VtPos pk = vt_pos(w, k);
          for (int a = 0; a < A.na; a++) {
            const int q = vt_at(w, pk, A.o[a]);
            if (q >= 0) {
              const int lq = wl[q];
              dep += (lq && abs(lq - lv) <= alpha) ? 1 : 0;
            }
          }
        }
        n = vt_append(it_i, it_j, n, lv != 0, lv, dep + 1, lane);
      }
      __builtin_amdgcn_wave_barrier();
      if (n) {
        zl_accumulate(it_i, it_j, n, lane, T, acc);
        groups = 1;
      }
    } else if (family == PRAD_VT_GLRLM) {
      for (int a = 0; a < A.na; a++) {
        signed char back[4] = {(signed char)-A.o[a][0], (signed char)-A.o[a][1], (signed char)-A.o[a][2], 0};
        int n = 0;
        bool multi = false;
        for (int k0 = 0; k0 < w.W; k0 += 64) {
          const int k = k0 + lane;
          const int lv = k < w.W ? wl[k] : 0;
          bool start = false;
          int len = 0;
          if (lv) {
            const VtPos pk = vt_pos(w, k);
            const int p = vt_at(w, pk, back);
            start = !(p >= 0 && wl[p] == lv);
            for (int m = 1; !multi; m++) {           // a second ROI voxel further along this line
              const int t = vt_at(w, pk, A.o[a], m);
              if (t < 0) break;
              multi = wl[t] != 0;
            }
            if (start) {
              len = 1;
              for (int m = 1;; m++) {
                const int q = vt_at(w, pk, A.o[a], m);
                if (q < 0 || wl[q] != lv) break;
                len++;
              }
            }
          }
          n = vt_append(it_i, it_j, n, start, lv, len, lane);
        }
        __builtin_amdgcn_wave_barrier();
        // cmatrices.c:524-534: no line of this angle holds more than one ROI voxel -> its run-length-1 column (all it
        // can contain) is zeroed: the angle is empty
        if (__ballot(multi) == 0ull || n == 0) continue;
        zl_accumulate(it_i, it_j, n, lane, T, acc);
        groups++;
        __builtin_amdgcn_wave_barrier();
      }
    } else {  // GLSZM: connected components of equal level by label propagation in LDS
      for (int k = lane; k < w.W; k += 64) aux[k] = wl[k] ? k : -1;
      __builtin_amdgcn_wave_barrier();
      bool changed = true;
      while (__ballot(changed)) {
        changed = false;
        for (int k = lane; k < w.W; k += 64) {
          const int lv = wl[k];
          if (!lv) continue;
          int best = aux[k];
          const VtPos pk = vt_pos(w, k);
          for (int a = 0; a < A.na; a++) {
            const int q = vt_at(w, pk, A.o[a]);
            if (q >= 0 && wl[q] == lv) best = min(best, aux[q]);
          }
          best = min(best, aux[best]);                 // one pointer jump
          if (best < aux[k]) {
            aux[k] = best;
            changed = true;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      // sizes: it_j reused as a counter array first
      for (int k = lane; k < w.W; k += 64) it_j[k] = 0;
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < w.W; k += 64)
        if (aux[k] >= 0) atomicAdd(it_j + aux[k], 1);
      __builtin_amdgcn_wave_barrier();
      int n = 0;
      for (int k0 = 0; k0 < w.W; k0 += 64) {
        const int k = k0 + lane;
        const bool root = k < w.W && aux[k] == k;
        const int lv = root ? wl[k] : 0, sz = root ? it_j[k] : 0;
        __builtin_amdgcn_wave_barrier();
        // roots are appended at positions <= their own index, so the counter array can be compacted in place
        n = vt_append(it_i, it_j, n, root, lv, sz, lane);
        __builtin_amdgcn_wave_barrier();
      }
      if (n) {
        zl_accumulate(it_i, it_j, n, lane, T, acc);
        groups = 1;
      }
    }
    // one reduction per requested feature (acc[] holds per-lane partial sums); static indices keep acc[] in registers
    unsigned want = 0;
    for (int f = 0; f < nfeat; f++) want |= 1u << feature_ids[f];
    double red[ZF_COUNT];
#pragma unroll
    for (int id = 0; id < ZF_COUNT; id++) red[id] = ((want >> id) & 1u) ? wave_sum_f64(acc[id]) : 0.0;
    if (lane == 0) {
      for (int f = 0; f < nfeat; f++) {
        const int id = feature_ids[f];
        double val = 0.0;
#pragma unroll
        for (int q = 0; q < ZF_COUNT; q++) val = q == id ? red[q] : val;
        double r;
        if (groups == 0) r = family == PRAD_VT_GLRLM ? __builtin_nan("") : 0.0;
        else r = val / (double)groups;
        out[(long long)f * nvox + v] = r;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- NGTDM ------------------------------------------------------------------------------------------------
enum { NF_Coarseness = 0, NF_Contrast, NF_Busyness, NF_Complexity, NF_Strength, NF_COUNT };

// LDS per wave: wl[MAXW] (int) | vd[MAXW] (double) | lev[256] (int) | pn[256] (double) | ps[256] (double)
#define PRAD_VN_LDS_BYTES (PRAD_VT_MAXW * 4 + PRAD_VT_MAXW * 8 + 256 * 4 + 256 * 8 + 256 * 8)

__global__ void __launch_bounds__(64 * PRAD_VT_WAVES) voxel_ngtdm_kernel(
    const uint8_t *__restrict__ L, int Nz, int Ny, int Nx, VoxAngles A, int Ng, int nvox,
    const int *__restrict__ voxels, int vox_nd, int radius, int f2d3, const int *__restrict__ feature_ids, int nfeat,
    double *__restrict__ out, const int *__restrict__ flags) {
#pragma clang fp contract(off)
  __shared__ double lds64[PRAD_VT_WAVES * PRAD_VN_LDS_BYTES / 8];
  if (flags[0]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *base = lds64 + (size_t)wave * (PRAD_VN_LDS_BYTES / 8);
  double *vd = base;                           // per window voxel: |i - mean of its neighbours|
  double *pn = vd + PRAD_VT_MAXW;              // per present level: n_i, then p_i
  double *ps = pn + 256;                       // per present level: s_i
  int *wl = (int *)(ps + 256);
  int *lev = wl + PRAD_VT_MAXW;                // present level values, ascending
  for (int v = blockIdx.x * PRAD_VT_WAVES + wave; v < nvox; v += gridDim.x * PRAD_VT_WAVES) {
    const VtWindow w = vt_load_window(L, Nz, Ny, Nx, nvox, voxels, vox_nd, radius, f2d3, v, lane, wl);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < w.W; k += 64) {
      const int lv = wl[k];
      double diff = 0.0;
      if (lv) {
        int cnt = 0, sum = 0;
        const VtPos pk = vt_pos(w, k);
        for (int a = 0; a < A.na; a++) {
          const int q = vt_at(w, pk, A.o[a]);
          if (q >= 0 && wl[q]) {
            cnt++;
            sum += wl[q];
          }
        }
        if (cnt) diff = fabs((double)lv - (double)sum / (double)cnt);     // cmatrices.c:637-643
      }
      vd[k] = diff;
    }
    __builtin_amdgcn_wave_barrier();
    // per level: n_i and s_i, the latter summed in raster order exactly like the reference's voxel loop
    int ngp = 0;
    for (int g0 = 1; g0 <= Ng; g0 += 64) {
      const int gl = g0 + lane;
      int n = 0;
      double s = 0.0;
      if (gl <= Ng)
        for (int k = 0; k < w.W; k++)
          if (wl[k] == gl) {
            n++;
            s += vd[k];
          }
      const unsigned long long B = __ballot(n > 0);
      if (n > 0) {
        const int pos = ngp + __popcll(B & ((1ull << lane) - 1ull));
        lev[pos] = gl;
        pn[pos] = (double)n;
        ps[pos] = s;
      }
      ngp += __popcll(B);
    }
    __builtin_amdgcn_wave_barrier();
    double nvp = 0, stot = 0;
    for (int k = lane; k < ngp; k += 64) {
      nvp += pn[k];
      stot += ps[k];
    }
    nvp = wave_sum_f64(nvp);
    stot = wave_sum_f64(stot);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < ngp; k += 64) pn[k] = pn[k] / nvp;      // p_i (ngtdm.py:120)
    __builtin_amdgcn_wave_barrier();
    double coarse = 0, contrast = 0, absdiff = 0, complexity = 0, strength = 0;
    for (int k = lane; k < ngp; k += 64) coarse += pn[k] * ps[k];
    coarse = wave_sum_f64(coarse);
    for (int t = lane; t < ngp * ngp; t += 64) {
      const int a = t / ngp, b = t % ngp;
      const double pa = pn[a], pb = pn[b], ia = (double)lev[a], ib = (double)lev[b], d = ia - ib;
      contrast += pa * pb * d * d;
      absdiff += fabs(ia * pa - ib * pb);
      complexity += fabs(d) * (pa * ps[a] + pb * ps[b]) / (pa + pb);
      strength += (pa + pb) * d * d;
    }
    contrast = wave_sum_f64(contrast);
    absdiff = wave_sum_f64(absdiff);
    complexity = wave_sum_f64(complexity);
    strength = wave_sum_f64(strength);
    if (lane == 0) {
      const double div = (double)ngp * (double)(ngp - 1);
      for (int f = 0; f < nfeat; f++) {
        double r;
        switch (feature_ids[f]) {
          case NF_Coarseness: r = coarse != 0 ? 1.0 / coarse : 1e6; break;                       // ngtdm.py:148-150
          case NF_Contrast: r = div != 0 ? contrast * stot / nvp / div : 0.0; break;             // :187-188
          case NF_Busyness: r = absdiff != 0 ? coarse / absdiff : 0.0; break;                    // :219-220
          case NF_Complexity: r = complexity / nvp; break;
          case NF_Strength: r = stot != 0 ? strength / stot : 0.0; break;                        // :284-285
          default: r = __builtin_nan("");
        }
        out[(long long)f * nvox + v] = r;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace prad
