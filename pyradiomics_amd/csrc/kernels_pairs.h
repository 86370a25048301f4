// kernels_pairs.h -- the tier between the sweeps and the exact generic kernels (round 5).
//
// The sweep kernels (kernels_sweep*.h) need unit angles and <= 160 grey levels; everything else -- `distances: [1, 2]`
// (62 angles, cmatrices.c:807-892), 161+ levels -- used to drop to kernels_generic.h: one fp64 L2 atomic per pair, 43 ms for
// a 256^3 GLCM and 82 ms for GLCM + GLRLM at 300 levels (270 - 500 x the sweep).  Here:
//
//   pairs_pack_kernel    int32 level + uint8 mask -> uint16 level, 0 outside the ROI; flags[0] on a level outside 1..Ng
//                        (the caller then re-runs the exact kernels: they reproduce the reference's aliasing / IndexError)
//   pairs_glcm_kernel    GLCM (cmatrices.c:4-92) for ANY offset list: one lane per voxel, its neighbours read from the
//                        packed volume (L1 / L2 hits), u32 counts in LDS tables [angles of the pass][level rows of the
//                        pass][Ng]: as many angles per pass as fit 150 KB (32 levels: 36 angles), or one angle and a tile of
//                        level rows when Ng x Ng does not fit (300 levels: 128 rows).  blockIdx.y = pass.
//   pairs_glrlm_kernel   GLRLM (cmatrices.c:299-541) for any level count: a lane whose voxel STARTS a run walks it; runs of
//                        length 1 are derived, not counted (every ROI voxel of level g lies on exactly one line of an
//                        angle: GLRLM[g][1] = N_g - sum_{len >= 2} len GLRLM[g][len]), so the u32 L2 atomics are as rare
//                        as the longer runs.
//   pairs_level_count_kernel, pairs_run1_kernel, pairs_multi_check_kernel   N_g, the derivation above, the reference's
//                        "no line of this angle holds two ROI voxels" rule (cmatrices.c:524-534).
// Counts are integers below 2^31 (the reference's own limit) accumulated in u32 and converted once: bit-exact.
#pragma once
#include "prad_runtime.h"
#include "kernels_sweep.h"

namespace prad {

#define PRAD_PAIR_MAXA 128
#define PRAD_PAIR_LDS_WORDS (150 * 256)    // 150 KB of u32
struct PairAngles {
  int n;
  signed char o[PRAD_PAIR_MAXA][4];   // dz, dy, dx of the volume embedded in 3-D
};

typedef unsigned short lev16;

__global__ void __launch_bounds__(256) pairs_pack_kernel(const int *__restrict__ image, const uint8_t *__restrict__ mask,
                                                         long long n, int Ng, lev16 *__restrict__ L, int *__restrict__ flags) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  bool bad = false;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    lev16 v = 0;
    if (mask[i]) {
      const int l = image[i];
      if (l >= 1 && l <= Ng) v = (lev16)l;
      else bad = true;
    }
    L[i] = v;
  }
  if (bad) flags[0] = 1;
}

// pass p: angles [ag * AG, min(Na, (ag + 1) * AG)), level rows [tile * RT, min(Ng, (tile + 1) * RT)) with ag = p / ntile
__global__ void __launch_bounds__(1024) pairs_glcm_kernel(const lev16 *__restrict__ L, int Nz, int Ny, int Nx, PairAngles A,
                                                          int AG, int RT, int ntile, int Ng, u32 *__restrict__ acc,
                                                          const int *__restrict__ flags) {
  extern __shared__ u32 tab[];
  if (flags[0]) return;
  const int pass = blockIdx.y, ag = pass / ntile, tile = pass - ag * ntile;
  const int a0 = ag * AG, na = min(A.n - a0, AG);
  const int r0 = tile * RT, nr = min(Ng - r0, RT);
  const int words = na * nr * Ng;
  for (int i = threadIdx.x; i < words; i += blockDim.x) tab[i] = 0;
  __syncthreads();
  // (the tier takes volumes below 2^31 voxels: 32-bit index arithmetic -- two 64-bit divisions per voxel visit were most of
  // this kernel's instructions)
  const unsigned plane = (unsigned)Ny * (unsigned)Nx, n = (unsigned)Nz * plane;
  const unsigned stride = gridDim.x * blockDim.x;
  // Eight voxels per lane and visit (round 6): one 16-byte load, the neighbour loads of the voxels that fall into this pass's
  // level rows issued TOGETHER, then their atomics.  One voxel per visit was a chain of dependent L2 round trips -- the level,
  // then its neighbour, ~1 us per voxel and lane with one workgroup per CU: 10.7 ms for 62 angles x 3 row tiles at 300 levels.
  const unsigned n8 = n >> 3;
  for (unsigned q8 = blockIdx.x * blockDim.x + threadIdx.x; q8 < n8; q8 += stride) {
    const uint4 w = reinterpret_cast<const uint4 *>(L)[q8];
    const unsigned ww[4] = {w.x, w.y, w.z, w.w};
    int row[8];
    unsigned okm = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int c = (int)((ww[e >> 1] >> (16 * (e & 1))) & 0xffffu);
      row[e] = c - 1 - r0;
      if (c != 0 && (unsigned)row[e] < (unsigned)nr) okm |= 1u << e;
    }
    if (!okm) continue;
    const unsigned i0 = q8 << 3;
    int ze[8], ye[8], xe[8];
    {
      const unsigned zq = i0 / plane, r = i0 - zq * plane, yq = r / (unsigned)Nx;
      int z = (int)zq, y = (int)yq, x = (int)(r - yq * (unsigned)Nx);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        ze[e] = z; ye[e] = y; xe[e] = x;
        if (++x == Nx) {
          x = 0;
          if (++y == Ny) { y = 0; z++; }
        }
      }
    }
    for (int k = 0; k < na; k++) {
      const int dz = A.o[a0 + k][0], dy = A.o[a0 + k][1], dx = A.o[a0 + k][2];
      const int off = dz * (int)plane + dy * Nx + dx;
      int v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int zz = ze[e] + dz, yy = ye[e] + dy, xx = xe[e] + dx;
        const bool in = ((okm >> e) & 1u) && (unsigned)zz < (unsigned)Nz && (unsigned)yy < (unsigned)Ny && (unsigned)xx < (unsigned)Nx;
        v[e] = in ? (int)L[(int)i0 + e + off] : 0;
      }
      u32 *tk = tab + (size_t)k * nr * Ng;
#pragma unroll
      for (int e = 0; e < 8; e++)
        if (v[e]) atomicAdd(tk + row[e] * Ng + (v[e] - 1), 1u);
    }
  }
  for (unsigned i = (n8 << 3) + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {     // (the last n % 8 voxels)
    const int c = L[i];
    const int row = c - 1 - r0;
    if (c == 0 || (unsigned)row >= (unsigned)nr) continue;
    const unsigned zq = i / plane, r = i - zq * plane, yq = r / (unsigned)Nx;
    const int z = (int)zq, y = (int)yq, x = (int)(r - yq * (unsigned)Nx);
    u32 *t = tab + row * Ng;
    for (int k = 0; k < na; k++) {
      const int dz = A.o[a0 + k][0], dy = A.o[a0 + k][1], dx = A.o[a0 + k][2];
      const int zz = z + dz, yy = y + dy, xx = x + dx;
      if ((unsigned)zz < (unsigned)Nz && (unsigned)yy < (unsigned)Ny && (unsigned)xx < (unsigned)Nx) {
        const int v = L[(int)i + dz * (int)plane + dy * Nx + dx];
        if (v) atomicAdd(t + (size_t)k * nr * Ng + (v - 1), 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < words; i += blockDim.x) {
    const u32 v = tab[i];
    if (v) {
      const int k = i / (nr * Ng), rc = i - k * nr * Ng;
      atomicAdd(acc + ((size_t)(a0 + k) * Ng + r0) * Ng + rc, v);
    }
  }
}

// voxels per level (the N_g of the run-length derivation), LDS-private when Ng words fit
__global__ void __launch_bounds__(256) pairs_level_count_kernel(const lev16 *__restrict__ L, long long n, int Ng, int use_lds,
                                                                u32 *__restrict__ counts, const int *__restrict__ flags) {
  extern __shared__ u32 tab[];
  if (flags[0]) return;
  if (use_lds) {
    for (int i = threadIdx.x; i < Ng; i += blockDim.x) tab[i] = 0;
    __syncthreads();
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = L[i];
    if (c) atomicAdd((use_lds ? tab : counts) + (c - 1), 1u);
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < Ng; i += blockDim.x) {
      const u32 v = tab[i];
      if (v) atomicAdd(counts + i, v);
    }
  }
}

// blockIdx.y = angle.  acc [Na][Ng][Nr] u32 takes the runs of length >= 2; multi[a] is raised by the first voxel that has an
// ROI voxel next to it along the angle (the cheap half of the reference's multiElement rule; pairs_multi_check_kernel decides
// the rest)
__global__ void __launch_bounds__(256) pairs_glrlm_kernel(const lev16 *__restrict__ L, int Nz, int Ny, int Nx, PairAngles A,
                                                          int Ng, int Nr, u32 *__restrict__ acc, int *__restrict__ multi,
                                                          const int *__restrict__ flags) {
  if (flags[0]) return;
  const int a = blockIdx.y;
  const int dz = A.o[a][0], dy = A.o[a][1], dx = A.o[a][2];
  const unsigned plane = (unsigned)Ny * (unsigned)Nx, n = (unsigned)Nz * plane;      // (below 2^31 voxels: 32-bit index arithmetic)
  const long long step = (long long)dz * plane + (long long)dy * Nx + dx;
  const unsigned stride = gridDim.x * blockDim.x;
  bool seen_pair = false;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = L[i];
    if (!c) continue;
    const unsigned zq = i / plane, rq = i - zq * plane, yq = rq / (unsigned)Nx;
    const int z = (int)zq, y = (int)yq, x = (int)(rq - yq * (unsigned)Nx);
    const int pz = z - dz, py = y - dy, px = x - dx;
    int zz = z + dz, yy = y + dy, xx = x + dx;
    const bool next_in = (unsigned)zz < (unsigned)Nz && (unsigned)yy < (unsigned)Ny && (unsigned)xx < (unsigned)Nx;
    const int nv = next_in ? L[(long long)i + step] : 0;
    seen_pair = seen_pair || nv != 0;
    if (nv != c) continue;                       // a run of length 1, or the last voxel of a longer one
    if ((unsigned)pz < (unsigned)Nz && (unsigned)py < (unsigned)Ny && (unsigned)px < (unsigned)Nx && L[(long long)i - step] == c) continue;   // not the start
    int len = 2;
    long long j = (long long)i + step;
    for (;;) {
      zz += dz; yy += dy; xx += dx;
      if ((unsigned)zz >= (unsigned)Nz || (unsigned)yy >= (unsigned)Ny || (unsigned)xx >= (unsigned)Nx) break;
      j += step;
      if (L[j] != c) break;
      len++;
    }
    if (len <= Nr) atomicAdd(acc + ((size_t)a * Ng + (c - 1)) * Nr + (len - 1), 1u);
  }
  if (__ballot(seen_pair) != 0 && (threadIdx.x & 63) == 0 && !multi[a]) multi[a] = 1;
}

// exact multiElement test for the angles the walk left open (every ROI voxel isolated along the angle): a line start
// counts the ROI voxels of its line, early exit
__global__ void __launch_bounds__(256) pairs_multi_check_kernel(PairAngles A, const lev16 *__restrict__ L, int Nz, int Ny, int Nx,
                                                                int *__restrict__ multi, const int *__restrict__ flags) {
  if (flags[0]) return;
  const int a = blockIdx.y;
  if (multi[a]) return;
  const int dz = A.o[a][0], dy = A.o[a][1], dx = A.o[a][2];
  const long long plane = (long long)Ny * Nx, n = (long long)Nz * plane;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int z = (int)(i / plane);
    const int r = (int)(i - (long long)z * plane);
    int y = r / Nx, x = r - y * Nx;
    const int pz = z - dz, py = y - dy, px = x - dx;
    if ((unsigned)pz < (unsigned)Nz && (unsigned)py < (unsigned)Ny && (unsigned)px < (unsigned)Nx) continue;
    int cnt = 0;
    while ((unsigned)z < (unsigned)Nz && (unsigned)y < (unsigned)Ny && (unsigned)x < (unsigned)Nx) {
      cnt += L[(long long)z * plane + (long long)y * Nx + x] != 0;
      if (cnt > 1) {
        multi[a] = 1;
        return;
      }
      z += dz; y += dy; x += dx;
    }
  }
}

// GLRLM[a][g][1] = N_g - sum_{len >= 2} len * GLRLM[a][g][len]: one wave per (angle, level)
__global__ void __launch_bounds__(256) pairs_run1_kernel(u32 *__restrict__ acc, const u32 *__restrict__ counts, int Ng, int Nr,
                                                         int Na, const int *__restrict__ flags) {
  if (flags[0]) return;
  const int pair = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pair >= Ng * Na) return;
  const int g = pair % Ng, a = pair / Ng;
  u32 *row = acc + ((size_t)a * Ng + g) * Nr;
  unsigned long long s = 0;
  for (int r = 1 + lane; r < Nr; r += 64) s += (unsigned long long)(r + 1) * row[r];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) row[0] = (u32)(counts[g] - s);    // (every run lies inside the level's voxels: s <= N_g)
}

// GLDM (cmatrices.c:660-754) / NGTDM (cmatrices.c:543-658) for level counts and angle sets beyond kernels_neigh.h (more than
// 255 levels, bin tables beyond 64 KB): the same per-voxel arithmetic on the 16-bit level volume, bins in LDS when they fit
// 150 KB, else straight in the global accumulators.  Bin layout as kernels_neigh.h ([Ng][Na+1]; NGTDM: u64, slot 0 = voxels
// of the level, slot c = sum of |c * level - sum of the c valid neighbours| -- exact integers, one division per slot in
// ngtdm_finalize_kernel), so the finalize kernels are shared.
// LDSMODE: 0 = bins in the global accumulators, 1 = bins in LDS in their own width (GLDM u32, NGTDM u64), 2 = NGTDM with u32
// bins in LDS, widened in the flush: the caller sizes the grid so that a workgroup's sums stay below 2^32 (voxels per
// workgroup x angles x Ng < 2^32), which halves the table (300 levels x 124 angles: 150 KB instead of 300)
template <bool NGTDM, int LDSMODE>
__global__ void __launch_bounds__(1024) pairs_neigh_kernel(PairAngles A, const lev16 *__restrict__ L, int Nz, int Ny, int Nx, int Ng,
                                                           int alpha, u32 *__restrict__ gldm_acc, u64 *__restrict__ ngtdm_acc,
                                                           const int *__restrict__ flags) {
  extern __shared__ u64 pn_lds64[];
  constexpr bool USE_LDS = LDSMODE != 0;
  constexpr bool BINS32 = !NGTDM || LDSMODE == 2;       // LDS bin width
  if (flags[0]) return;
  const int W = A.n + 1;
  u32 *h32 = reinterpret_cast<u32 *>(pn_lds64);
  const int nbins = Ng * W;
  if (USE_LDS) {
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
      if (BINS32) h32[i] = 0;
      else pn_lds64[i] = 0;
    }
    __syncthreads();
  }
  const unsigned plane = (unsigned)Ny * (unsigned)Nx, n = (unsigned)Nz * plane;      // (below 2^31 voxels: 32-bit index arithmetic)
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = L[i];
    if (!c) continue;
    const unsigned zq = i / plane, r = i - zq * plane, yq = r / (unsigned)Nx;
    const int z = (int)zq, y = (int)yq, x = (int)(r - yq * (unsigned)Nx);
    int cnt = 0, dep = 0;
    long long sum = 0;
    // (four neighbours per round, their loads issued together: one at a time every load waited for the one before it -- the
    //  `continue`s keep the compiler from overlapping them -- and 124 angles were 124 L1 / L2 round trips per voxel)
    for (int a4 = 0; a4 < A.n; a4 += 4) {
      int v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int a = min(a4 + u, A.n - 1);
        const int dz = A.o[a][0], dy = A.o[a][1], dx = A.o[a][2];
        const int zz = z + dz, yy = y + dy, xx = x + dx;
        const bool in = a4 + u < A.n && (unsigned)zz < (unsigned)Nz && (unsigned)yy < (unsigned)Ny && (unsigned)xx < (unsigned)Nx;
        v[u] = in ? (int)L[(int)i + dz * (int)plane + dy * Nx + dx] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (!v[u]) continue;
        if (NGTDM) {
          cnt++;
          sum += v[u];
        } else {
          int d = c - v[u];
          d = d < 0 ? -d : d;
          dep += (d <= alpha);
        }
      }
    }
    if (NGTDM) {
      long long d = (long long)cnt * c - sum;
      d = d < 0 ? -d : d;
      if (USE_LDS && BINS32) {
        u32 *row = h32 + (size_t)(c - 1) * W;
        atomicAdd(row, 1u);
        if (cnt && d) atomicAdd(row + cnt, (u32)d);
      } else {
        u64 *row = (USE_LDS ? pn_lds64 : ngtdm_acc) + (size_t)(c - 1) * W;
        atomicAdd(row, 1ull);
        if (cnt && d) atomicAdd(row + cnt, (u64)d);
      }
    } else {
      atomicAdd((USE_LDS ? h32 : gldm_acc) + (size_t)(c - 1) * W + dep, 1u);
    }
  }
  if (USE_LDS) {
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
      if (NGTDM) {
        const u64 v = BINS32 ? (u64)h32[i] : pn_lds64[i];
        if (v) atomicAdd(ngtdm_acc + i, v);
      } else {
        const u32 v = h32[i];
        if (v) atomicAdd(gldm_acc + i, v);
      }
    }
  }
}

// ---- NGTDM over a FULL box of neighbours as separable box sums (round 6) ---------------------------------------------------
// `distances: [1]`, `[1, 2]`, ... ask for every offset of a (2r+1)^3 cube but the centre (26, 124 angles; in-plane boxes under
// force2D).  The two numbers the NGTDM needs per voxel -- how many neighbours lie in the ROI and the sum of their levels
// (cmatrices.c:543-658) -- are then box sums of the ROI indicator and of the level volume (0 outside the ROI): three 1-D passes of
// 2r + 1 loads instead of (2r+1)^3 - 1 neighbour visits per voxel (3.0 ms for 124 neighbours at 256^3), exact integers either way.
// A word carries (ROI voxels << SH | level sum): u32 words with SH = 24 for boxes of at most 127 voxels with 127 x Ng < 2^24 (distances
// up to 2), u64 words with SH = 40 beyond (distances [1, 2, 3]: 342 neighbours, more than the tier's angle table holds -- the box
// path needs the radii only).
template <typename WT, int SH>
__global__ void __launch_bounds__(256) pairs_box_axis_kernel(const lev16 *__restrict__ L, const WT *__restrict__ in, WT *__restrict__ out,
                                                             int Nz, int Ny, int Nx, int axis, int r, const int *__restrict__ flags) {
  if (flags[0]) return;
  const unsigned plane = (unsigned)Ny * (unsigned)Nx, n = (unsigned)Nz * plane;
  const unsigned stride = gridDim.x * blockDim.x;
  const int ext = axis == 0 ? Nz : (axis == 1 ? Ny : Nx);
  const int step = axis == 0 ? (int)plane : (axis == 1 ? Nx : 1);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned zq = i / plane, rr = i - zq * plane, yq = rr / (unsigned)Nx;
    const int pos = axis == 0 ? (int)zq : (axis == 1 ? (int)yq : (int)(rr - yq * (unsigned)Nx));
    WT acc = 0;
    for (int o = -r; o <= r; o++) {
      if ((unsigned)(pos + o) >= (unsigned)ext) continue;
      const int j = (int)i + o * step;
      if (L) {
        const WT c = L[j];
        acc += c ? (((WT)1 << SH) | c) : (WT)0;
      } else {
        acc += in[j];
      }
    }
    out[i] = acc;
  }
}

// the bins of pairs_neigh_kernel<true, LDSMODE> from the box sums: B = the x- and y-summed words, the z sum is taken here
template <int LDSMODE, typename WT, int SH>
__global__ void __launch_bounds__(1024) pairs_ngtdm_box_kernel(const lev16 *__restrict__ L, const WT *__restrict__ B, int Nz, int Ny, int Nx,
                                                               int rz, int Ng, int W, u64 *__restrict__ ngtdm_acc,
                                                               const int *__restrict__ flags) {
  extern __shared__ u64 pn_lds64[];
  constexpr bool USE_LDS = LDSMODE != 0;
  constexpr bool BINS32 = LDSMODE == 2;
  if (flags[0]) return;
  u32 *h32 = reinterpret_cast<u32 *>(pn_lds64);
  const int nbins = Ng * W;
  if (USE_LDS) {
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
      if (BINS32) h32[i] = 0;
      else pn_lds64[i] = 0;
    }
    __syncthreads();
  }
  const unsigned plane = (unsigned)Ny * (unsigned)Nx, n = (unsigned)Nz * plane;
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = L[i];
    if (!c) continue;
    const int z = (int)(i / plane);
    WT box = 0;
    for (int o = -rz; o <= rz; o++)
      if ((unsigned)(z + o) < (unsigned)Nz) box += B[(int)i + o * (int)plane];
    const int cnt = (int)(box >> SH) - 1;                     // (the box holds the voxel itself)
    const long long sum = (long long)(box & (((WT)1 << SH) - 1)) - c;
    long long d = (long long)cnt * c - sum;
    d = d < 0 ? -d : d;
    if (USE_LDS && BINS32) {
      u32 *row = h32 + (size_t)(c - 1) * W;
      atomicAdd(row, 1u);
      if (cnt && d) atomicAdd(row + cnt, (u32)d);
    } else {
      u64 *row = (USE_LDS ? pn_lds64 : ngtdm_acc) + (size_t)(c - 1) * W;
      atomicAdd(row, 1ull);
      if (cnt && d) atomicAdd(row + cnt, (u64)d);
    }
  }
  if (USE_LDS) {
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
      const u64 v = BINS32 ? (u64)h32[i] : pn_lds64[i];
      if (v) atomicAdd(ngtdm_acc + i, v);
    }
  }
}

}  // namespace prad
