// kernels_glszm.h -- grey-level size-zone matrix on gfx950 (cmatrices.c:94-297).
//
// Segment mode (one box = the whole array): zones are the connected components of equal-level masked voxels under the
// angle set.  Two models:
//   * packed-byte path (<= 3-D, the full 26- or in-plane 8-neighbourhood, levels 1..255) -- the DENSE model:
//       glszm_tile8_kernel        zones labelled inside 8 x 8 x 64 tiles in LDS (union-find over in-tile indices); every tile
//                                 component gets a dense id; vid[voxel] = id of its tile component; the same-level
//                                 neighbours in OTHER tiles that no in-tile neighbour already ties to the voxel go to a
//                                 work list of pairs
//       glszm_dense_init_kernel   parent[id] = id, zsize[id] = voxels of the tile component
//       glszm_pairs_kernel        union-find over the dense ids along the work list (atomicMin on the larger root)
//       glszm_rootsum_dense_kernel  counts folded into the zone roots, parent[] flattened
//       glszm_stats / fill kernels  walk the ids (a zone = an id with parent[id] == id; level in tinfo[id] >> 16)
//     The ordered zone list (tempData parity, cmatrices.c:255-258) needs the first voxel of every zone: a scan of vid[]
//     (glszm_zmin_kernel), on demand.
//   * any other call -- the label-volume model: label[i] = i (masked) | -1, unions by atomicMin on the larger root (the
//     returned old value keeps the structure consistent when views are stale), flatten, count; the root of a zone is its
//     smallest linear index = the voxel where the reference's raster scan discovers the zone, so listing roots in index
//     order reproduces the reference's tempData order exactly.
//
// Voxel mode (many small boxes): one lane per kernel runs the reference's raster-order flood fill inside its
// box with a private visited map / stack (interleaved [slot][kernel] so neighbouring lanes touch neighbouring
// addresses); the mask itself is never modified, which replaces the reference's processedStack restore
// (cmatrices.c:264-272).
//
// Phase 2 (fill) histograms (level, size) into float64 [Nvox][Ng][maxRegion] once the caller knows maxRegion.
#pragma once
#include <algorithm>
#include "prad_runtime.h"
#include "kernels_generic.h"
#include "kernels_neigh.h"

namespace prad {

struct GlszmState {
  bool valid = false;
  bool voxel_mode = false;
  int device = -1;
  Geo g;
  int nvox = 0;
  long long boxmax = 0;
  long long nzones = 0;
  int max_region = 0;
  int nsizes = -1;                 // segment: distinct zone sizes found by glszm_distinct_sizes (-1 = not run)
  int nsmall = 0, nlarge = 0;      //   ... of which below / at or above PRAD_SMALL_SIZES
  unsigned *small_bits = nullptr;  // segment: bitmap of the zone sizes < PRAD_SMALL_SIZES that occur
  int *large_list = nullptr;       // segment: every zone size >= PRAD_SMALL_SIZES (unsorted, with repeats)
  int *large_count = nullptr;
  int large_cap = 0;
  int *small_rank = nullptr;       // segment, after glszm_distinct_sizes: size -> compact column
  const int32_t *image = nullptr;  // segment mode: levels are read from the image at fill time
  int *labels = nullptr;           // segment: [n] root label or -1
  unsigned *sizes = nullptr;       // segment: [n] zone size at root index
  // segment, packed-byte path = the dense model (glszm_tile8_kernel): labels[] is vid[] (voxel -> dense id of its tile
  // root), sizes[] is tsize[] (indexed by dense id); parent == nullptr: the label volume model of the int32 kernels
  int *parent = nullptr;           // [n] dense union-find, flat after glszm_rootsum_dense_kernel
  unsigned *tinfo = nullptr;       // [n] dense id -> voxels of the tile component | grey level << 16
  size_t idcap = 0;                // entries the dense arrays hold (ids come in chunks: holes)
  bool published = false;          // the ordered zone list turned labels[] (vid[]) into its label-volume view
  int *rootctl = nullptr;          // [0] tile roots, [2] pairs in the work list, [3] work list overflow
  int *zones = nullptr;            // voxel: [boxmax][2][nvox] (level,size) interleaved by kernel
  int *zone_count = nullptr;       // voxel: [nvox]
};
inline GlszmState &glszm_state() {
  static thread_local GlszmState s;
  return s;
}

// ---- segment mode --------------------------------------------------------------------------------
__global__ void glszm_init_kernel(const uint8_t *__restrict__ mask, long long n, int *__restrict__ labels,
                                  unsigned *__restrict__ sizes) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    labels[i] = mask[i] ? (int)i : -1;
    sizes[i] = 0;
  }
}

// find with path halving.  Labels only ever decrease towards the root, so a racing / stale view at worst repeats
// a hop; writing the grandparent is always a valid shortcut (it is an ancestor).
__device__ __forceinline__ int uf_find(int *labels, int i) {
  int p = __builtin_nontemporal_load(labels + i);
  while (p != i) {
    const int g = __builtin_nontemporal_load(labels + p);
    if (g != p) labels[i] = g;
    i = p;
    p = g;
  }
  return i;
}

__device__ __forceinline__ void uf_union(int *labels, int a, int b) {
  while (true) {
    a = uf_find(labels, a);
    b = uf_find(labels, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }
    const int old = atomicMin(labels + a, b);
    if (old == a) return;  // a was a root and now points to b
    a = old;               // a had already been linked elsewhere: join that set with b's
  }
}

// the same union with the two finds walking their chains side by side (two independent loads in flight per hop instead
// of one): for the wide phase of glszm_border8s_kernel, where a lane has exactly one union to make
__device__ __forceinline__ void uf_union_wide(int *labels, int a, int b) {
  while (true) {
    int pa = __builtin_nontemporal_load(labels + a), pb = __builtin_nontemporal_load(labels + b);
    while (pa != a || pb != b) {
      const int ga = __builtin_nontemporal_load(labels + pa), gb = __builtin_nontemporal_load(labels + pb);
      if (pa != a) {
        if (ga != pa) labels[a] = ga;
        a = pa;
        pa = ga;
      }
      if (pb != b) {
        if (gb != pb) labels[b] = gb;
        b = pb;
        pb = gb;
      }
    }
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }
    const int old = atomicMin(labels + a, b);
    if (old == a) return;
    a = old;
  }
}

__global__ void __launch_bounds__(256) glszm_merge_kernel(Geo g, const int *__restrict__ angles, int Na,
                                                          const int *__restrict__ image,
                                                          const uint8_t *__restrict__ mask,
                                                          int *__restrict__ labels) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += stride) {
    if (!mask[i]) continue;
    int c[PRAD_MAX_ND];
    long long rem = i;
    for (int d = 0; d < g.nd; d++) {
      c[d] = (int)(rem / g.stride[d]);
      rem -= (long long)c[d] * g.stride[d];
    }
    const int gl = image[i];
    for (int a = 0; a < Na; a++) {
      long long j = 0;
      bool in = true;
      for (int d = 0; d < g.nd; d++) {
        const int q = c[d] + angles[a * g.nd + d];
        if (q < 0 || q >= g.size[d]) { in = false; break; }
        j += (long long)q * g.stride[d];
      }
      if (!in || j >= i) continue;  // only backward neighbours: each adjacent pair is united once
      if (mask[j] && image[j] == gl) uf_union(labels, (int)i, (int)j);
    }
  }
}

// ---- tiled variant (Nd <= 3): zones are first labelled inside 4 x 8 x 64 tiles entirely in LDS, then only voxel
// pairs that straddle a tile boundary are united in HBM.  Most unions (and all of their retries on large zones)
// never leave the CU; the global forest starts from one root per (zone, tile) instead of one per voxel.
#ifndef PRAD_TZ
#define PRAD_TZ 8
#endif
#ifndef PRAD_TY
#define PRAD_TY 8
#endif
#define PRAD_TX 64
#define PRAD_TVOX (PRAD_TZ * PRAD_TY * PRAD_TX)
#define PRAD_GZ_TILE_ROOTS 1024    // roots per tile the tile-root list takes (glszm_tile8_kernel)
struct Offsets3 {
  int na;
  signed char o[32][4];   // backward neighbours only (linear offset < 0), embedded in 3-D
};

// `lab` is a __shared__ array.  The loads are typed as LDS loads: through a generic `volatile int *` the compiler emits
// flat_load ... sc0 sc1 (system scope, the flat path's latency) for every hop of the pointer chase -- 64 M of them per
// 512^3 volume in glszm_tile8_kernel, 2/3 of its time (profiles/r04_probes.md section 11).
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ int lds_find(int *lab, int i) {
  volatile lds_int_t *l = (volatile lds_int_t *)lab;
  int p;
#ifdef PRAD_T8_HALVE
  p = l[i];
  while (p != i) {             // (probe build: path halving -- a label only ever moves to a smaller ancestor)
    const int g = l[p];
    if (g != p) l[i] = g;
    i = p;
    p = g;
  }
#else
  while ((p = l[i]) != i) i = p;
#endif
  return i;
}
__device__ __forceinline__ void lds_union(int *lab, int a, int b) {
  while (true) {
    a = lds_find(lab, a);
    b = lds_find(lab, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }
    const int old = atomicMin(lab + a, b);
    if (old == a) return;
    a = old;
  }
}

// mode: 0 = any offset list, 1 = the full 26-neighbourhood (13 backward offsets), 2 = the full in-plane
// 8-neighbourhood (4 backward offsets, dz = 0)
__global__ void __launch_bounds__(256) glszm_tile_kernel(Offsets3 A, int mode, const int *__restrict__ image,
                                                         const uint8_t *__restrict__ mask, int Nz, int Ny, int Nx,
                                                         int *__restrict__ labels, unsigned *__restrict__ sizes) {
  __shared__ int img[PRAD_TVOX];
  __shared__ int lab[PRAD_TVOX];
  __shared__ uint8_t msk[PRAD_TVOX];
  const int tx = (Nx + PRAD_TX - 1) / PRAD_TX, ty = (Ny + PRAD_TY - 1) / PRAD_TY;
  const int bz = blockIdx.x / (ty * tx), br = blockIdx.x % (ty * tx);
  const int z0 = bz * PRAD_TZ, y0 = (br / tx) * PRAD_TY, x0 = (br % tx) * PRAD_TX;
  for (int k = threadIdx.x; k < PRAD_TVOX; k += blockDim.x) {
    const int lx = k % PRAD_TX, ly = (k / PRAD_TX) % PRAD_TY, lz = k / (PRAD_TX * PRAD_TY);
    const int z = z0 + lz, y = y0 + ly, x = x0 + lx;
    const bool in = z < Nz && y < Ny && x < Nx;
    const long long gi = ((long long)z * Ny + y) * Nx + x;
    const bool m = in && mask[gi] != 0;
    msk[k] = m;
    img[k] = m ? image[gi] : 0;
    lab[k] = k;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < PRAD_TVOX; k += blockDim.x) {
    if (!msk[k]) continue;
    const int lx = k % PRAD_TX, ly = (k / PRAD_TX) % PRAD_TY, lz = k / (PRAD_TX * PRAD_TY);
    const int gl = img[k];
    // in-tile index of the neighbour at (dz, dy, dx) if it lies in the tile and continues the zone, else -1
    auto same = [&](int dz, int dy, int dx) -> int {
      const int qz = lz + dz, qy = ly + dy, qx = lx + dx;
      if ((unsigned)qz >= PRAD_TZ || (unsigned)qy >= PRAD_TY || (unsigned)qx >= PRAD_TX) return -1;
      const int j = (qz * PRAD_TY + qy) * PRAD_TX + qx;
      return (msk[j] && img[j] == gl) ? j : -1;
    };
    if (mode == 0) {
      for (int a = 0; a < A.na; a++) {
        const int j = same(A.o[a][0], A.o[a][1], A.o[a][2]);
        if (j >= 0) lds_union(lab, k, j);
      }
      continue;
    }
    // Full 8- / 26-neighbourhoods: a neighbour that is itself adjacent (within its plane) to one already united
    // with k belongs to the same zone through that plane's own unions, so its union is redundant.  On smooth data
    // this leaves ~2 unions per voxel instead of ~10.
    const int b = same(0, -1, 0);                    // in-plane: b is adjacent to a, c and d
    if (b >= 0) lds_union(lab, k, b);
    else {
      const int a = same(0, -1, -1), c = same(0, -1, 1), d = same(0, 0, -1);
      if (c >= 0) lds_union(lab, k, c);
      if (a >= 0) lds_union(lab, k, a);              // a and d are adjacent to each other
      else if (d >= 0) lds_union(lab, k, d);
    }
    if (mode == 1) {                                 // plane above: its centre is adjacent to the other eight
      const int m = same(-1, 0, 0);
      if (m >= 0) lds_union(lab, k, m);
      else {
        const int e1 = same(-1, -1, 0), e2 = same(-1, 1, 0), e3 = same(-1, 0, -1), e4 = same(-1, 0, 1);
        if (e1 >= 0) lds_union(lab, k, e1);
        if (e2 >= 0) lds_union(lab, k, e2);
        if (e3 >= 0) lds_union(lab, k, e3);
        if (e4 >= 0) lds_union(lab, k, e4);
        int q;                                       // a corner is adjacent to the two edges next to it
        if (e1 < 0 && e3 < 0 && (q = same(-1, -1, -1)) >= 0) lds_union(lab, k, q);
        if (e1 < 0 && e4 < 0 && (q = same(-1, -1, 1)) >= 0) lds_union(lab, k, q);
        if (e2 < 0 && e3 < 0 && (q = same(-1, 1, -1)) >= 0) lds_union(lab, k, q);
        if (e2 < 0 && e4 < 0 && (q = same(-1, 1, 1)) >= 0) lds_union(lab, k, q);
      }
    }
  }
  __syncthreads();
  // flatten inside the tile and count each local component: img[] is reused for the counts (its levels are no
  // longer needed), runs of equal roots among the lanes of a wave issue one LDS atomic
  int root[PRAD_TVOX / 256];
#pragma unroll
  for (int q = 0; q < PRAD_TVOX / 256; q++) {
    const int k = threadIdx.x + q * 256;
    root[q] = msk[k] ? lds_find(lab, k) : -1;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < PRAD_TVOX; k += blockDim.x) img[k] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < PRAD_TVOX / 256; q++) {
    const int r = root[q];
    int left = __shfl_up(r, 1);
    if (lane == 0) left = -2;
    const unsigned long long B = __ballot(r != left || lane == 0);
    if (r >= 0 && r != left) {
      const unsigned long long above = lane == 63 ? 0ull : (B >> (lane + 1));
      atomicAdd((unsigned *)img + r, (unsigned)(above ? __ffsll((long long)above) : 64 - lane));
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PRAD_TVOX / 256; q++) {
    const int k = threadIdx.x + q * 256;
    const int lx = k % PRAD_TX, ly = (k / PRAD_TX) % PRAD_TY, lz = k / (PRAD_TX * PRAD_TY);
    const int z = z0 + lz, y = y0 + ly, x = x0 + lx;
    if (z >= Nz || y >= Ny || x >= Nx) continue;
    const long long gi = ((long long)z * Ny + y) * Nx + x;
    const int r = root[q];
    if (r < 0) { labels[gi] = -1; sizes[gi] = 0; continue; }
    const int rx = r % PRAD_TX, ry = (r / PRAD_TX) % PRAD_TY, rz = r / (PRAD_TX * PRAD_TY);
    labels[gi] = (int)(((long long)(z0 + rz) * Ny + (y0 + ry)) * Nx + (x0 + rx));   // local raster order == global order
    sizes[gi] = r == k ? (unsigned)img[k] : 0u;     // voxels of the local component, kept at its (tile) root
  }
}

// unions across tile boundaries only.  The union runs between the two tile roots (labels[] of a non-root voxel is
// one hop from its tile root and is never rewritten), and a lane whose (root, root) pair repeats its left
// neighbour's -- the usual case along a face shared by two large components -- skips the union altogether.
__global__ void __launch_bounds__(256) glszm_border_kernel(Offsets3 A, const int *__restrict__ image,
                                                           const uint8_t *__restrict__ mask, int Nz, int Ny, int Nx,
                                                           int *__restrict__ labels) {
  // One wave per row (z, y).  Rows on a z- or y-face of their tile are scanned completely; in all other rows only
  // the two x-faces of every tile (x mod 64 in {0, 63}) can have a backward neighbour in another tile, so the wave
  // visits just those.  No per-voxel index decoding: z, y are wave-uniform, x comes from the lane.
  const int lane = threadIdx.x & 63;
  const long long nrows = (long long)Nz * Ny;
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const int xtiles = (Nx + PRAD_TX - 1) / PRAD_TX;
  for (long long row = wave0; row < nrows; row += nwaves) {
    const int z = (int)(row / Ny), y = (int)(row - (long long)z * Ny);
    const bool row_edge = (z % PRAD_TZ == 0) || (y % PRAD_TY == 0) || (y % PRAD_TY == PRAD_TY - 1);
    const int count = row_edge ? Nx : 2 * xtiles;            // candidates in this row
    const long long rbase = row * Nx;
    for (int c0 = 0; c0 < count; c0 += 64) {
      const int c = c0 + lane;
      int x = -1;
      if (c < count) x = row_edge ? c : (c >> 1) * PRAD_TX + ((c & 1) ? PRAD_TX - 1 : 0);
      int gl = 0, li = -1;
      if (x >= 0 && x < Nx && mask[rbase + x]) {
        gl = image[rbase + x];
        li = labels[rbase + x];
      }
      if (__ballot(li >= 0) == 0ull) continue;
      for (int a = 0; a < A.na; a++) {
        int lj = -1;
        if (li >= 0) {
          const int qz = z + A.o[a][0], qy = y + A.o[a][1], qx = x + A.o[a][2];
          const bool in = (unsigned)qz < (unsigned)Nz && (unsigned)qy < (unsigned)Ny && (unsigned)qx < (unsigned)Nx;
          const bool same_tile = qz / PRAD_TZ == z / PRAD_TZ && qy / PRAD_TY == y / PRAD_TY && qx / PRAD_TX == x / PRAD_TX;
          if (in && !same_tile) {
            const long long j = ((long long)qz * Ny + qy) * Nx + qx;
            if (mask[j] && image[j] == gl) lj = labels[j];
          }
        }
        const int pi = __shfl_up(li, 1), pj = __shfl_up(lj, 1);
        if (lj >= 0 && !(lane > 0 && pi == li && pj == lj)) uf_union(labels, li, lj);
      }
    }
  }
}

// The same scan for the full 26- (MODE 1) / in-plane 8-neighbourhood (MODE 2) with the redundancy rules of the tile
// kernel applied to GLOBAL sameness: a backward neighbour whose own (earlier) unions already tie it to a neighbour we
// unite with is skipped -- and so are its loads.  By induction over raster order every adjacent same-level pair still
// ends up connected: each voxel is united with one representative of every cluster of mutually adjacent backward
// neighbours, and the members of a cluster are adjacent voxels that precede it.  Pairs inside one tile are left to
// the tile kernel (which applies the rules to in-tile sameness and therefore performs a superset of them).
template <int MODE>
__global__ void __launch_bounds__(256) glszm_border_full_kernel(const int *__restrict__ image,
                                                                const uint8_t *__restrict__ mask, int Nz, int Ny,
                                                                int Nx, int *__restrict__ labels) {
  const int lane = threadIdx.x & 63;
  const long long nrows = (long long)Nz * Ny;
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const int xtiles = (Nx + PRAD_TX - 1) / PRAD_TX;
  for (long long row = wave0; row < nrows; row += nwaves) {
    const int z = (int)(row / Ny), y = (int)(row - (long long)z * Ny);
    const bool row_edge = (MODE == 1 && z % PRAD_TZ == 0) || (y % PRAD_TY == 0) || (MODE == 1 && y % PRAD_TY == PRAD_TY - 1);
    const int count = row_edge ? Nx : 2 * xtiles;
    const long long rbase = row * Nx;
    for (int c0 = 0; c0 < count; c0 += 64) {
      const int c = c0 + lane;
      int x = -1;
      if (c < count) x = row_edge ? c : (c >> 1) * PRAD_TX + ((c & 1) ? PRAD_TX - 1 : 0);
      int gl = 0, li = -1;
      if (x >= 0 && x < Nx && mask[rbase + x]) {
        gl = image[rbase + x];
        li = labels[rbase + x];
      }
      if (__ballot(li >= 0) == 0ull) continue;
      // linear index of the neighbour if it continues the zone, else -1 (only evaluated for live lanes)
      auto same = [&](int dz, int dy, int dx) -> long long {
        const int qz = z + dz, qy = y + dy, qx = x + dx;
        if ((unsigned)qz >= (unsigned)Nz || (unsigned)qy >= (unsigned)Ny || (unsigned)qx >= (unsigned)Nx) return -1;
        const long long j = ((long long)qz * Ny + qy) * Nx + qx;
        return (mask[j] && image[j] == gl) ? j : -1;
      };
      // unite across the tile boundary; every lane of the wave takes part in the shuffle
      auto pair = [&](long long j, int dz, int dy, int dx) {
        int lj = -1;
        if (j >= 0) {
          const int qz = z + dz, qy = y + dy, qx = x + dx;
          const bool same_tile = qz / PRAD_TZ == z / PRAD_TZ && qy / PRAD_TY == y / PRAD_TY && qx / PRAD_TX == x / PRAD_TX;
          if (!same_tile) lj = labels[j];
        }
        const int pi = __shfl_up(li, 1), pj = __shfl_up(lj, 1);
        if (lj >= 0 && !(lane > 0 && pi == li && pj == lj)) uf_union(labels, li, lj);
      };
      const bool live = li >= 0;
      const long long jb = live ? same(0, -1, 0) : -1;
      pair(jb, 0, -1, 0);
      const bool nb = live && jb < 0;
      const long long jc = nb ? same(0, -1, 1) : -1;
      pair(jc, 0, -1, 1);
      const long long ja = nb ? same(0, -1, -1) : -1;
      pair(ja, 0, -1, -1);
      const long long jd = (nb && ja < 0) ? same(0, 0, -1) : -1;
      pair(jd, 0, 0, -1);
      if (MODE == 1) {
        const long long jm = live ? same(-1, 0, 0) : -1;
        pair(jm, -1, 0, 0);
        const bool nm = live && jm < 0;
        const long long e1 = nm ? same(-1, -1, 0) : -1, e2 = nm ? same(-1, 1, 0) : -1;
        const long long e3 = nm ? same(-1, 0, -1) : -1, e4 = nm ? same(-1, 0, 1) : -1;
        pair(e1, -1, -1, 0);
        pair(e2, -1, 1, 0);
        pair(e3, -1, 0, -1);
        pair(e4, -1, 0, 1);
        pair((nm && e1 < 0 && e3 < 0) ? same(-1, -1, -1) : -1, -1, -1, -1);
        pair((nm && e1 < 0 && e4 < 0) ? same(-1, -1, 1) : -1, -1, -1, 1);
        pair((nm && e2 < 0 && e3 < 0) ? same(-1, 1, -1) : -1, -1, 1, -1);
        pair((nm && e2 < 0 && e4 < 0) ? same(-1, 1, 1) : -1, -1, 1, 1);
      }
    }
  }
}

// ---- packed-byte variants (levels 1..255 as uint8, 0 = outside the ROI; the volume pack_levels writes for the
// GLDM / NGTDM kernels) for the full 26- (MODE 1) / in-plane 8-neighbourhood (MODE 2).
// Tile kernel: a lane owns 4 x-adjacent voxels (one LDS dword).  The tile sits in LDS with a zero halo (one plane
// above, one row either side in y, one dword either side in x), so a neighbour outside the tile simply never
// compares equal and no bounds are tested.  For each of the 4 (dz, dy) rows of backward neighbours the lane reads
// 3 dwords and derives the "same level" flags of its 4 voxels with packed-byte arithmetic (xor with the replicated
// centre level, zero-byte test) -- ~27 VALU and ~4 LDS reads per voxel instead of 13 neighbours x (bounds test,
// address, two LDS reads, compare).  Unions, flattening and the per-component counts are those of glszm_tile_kernel.
#define PRAD_T8_ROWDW (PRAD_TX / 4 + 2)                       // dwords per padded LDS row
#define PRAD_T8_DW ((PRAD_TZ + 1) * (PRAD_TY + 2) * PRAD_T8_ROWDW)
__device__ __forceinline__ unsigned t8_nonzero3(unsigned w) {  // bit 7 of every non-zero byte of the low 3 (exact)
  return (((w & 0x7f7f7fu) + 0x7f7f7fu) | w) & 0x808080u;
}
// The 13 backward neighbours of a voxel as bit positions n:  n = 3r + (dx + 1) for the backward rows
//   r = 0: (dz 0, dy -1)   1: (dz -1, dy -1)   2: (dz -1, dy 0)   3: (dz -1, dy +1),   and n = 12: (0, 0, -1).
// t8_flags3 turns the per-byte equality bits (7 / 15 / 23) of one row into 3 contiguous bits.
__device__ __forceinline__ unsigned t8_flags3(unsigned e) {
  const unsigned t = (e >> 7) & 0x010101u;
  return (t | (t >> 7) | (t >> 14)) & 7u;
}
__device__ __forceinline__ void t8_offset(int n, int &dz, int &dy, int &dx) {
  const int r = (n * 11) >> 5;                       // n / 3 for n < 16
  dz = (n >= 3 && n < 12) ? -1 : 0;
  dy = n >= 12 ? 0 : (r == 0 ? -1 : r - 2);
  dx = n >= 12 ? -1 : n - 3 * r - 1;
}
// Which of the "same level" neighbours S (13 bits) a voxel still has to be united with (redundancy rules of
// glszm_tile_kernel): a neighbour adjacent -- within its plane -- to one already selected is tied to it by that
// plane's own unions.  Branch-free, so the selection of a whole lane is plain ALU work and the (divergent, latency
// bound) unions run afterwards in one compact loop over the set bits.
template <int MODE>
__device__ __forceinline__ unsigned t8_select(unsigned S) {
  const bool b = S & 2u, a = S & 1u;
  unsigned sel = b ? 2u : ((S & 4u) | (a ? 1u : (S & 0x1000u)));
  if (MODE == 1) {
    const bool m = S & 0x80u, e1 = S & 0x10u, e3 = S & 0x40u, e4 = S & 0x100u, e2 = S & 0x400u;
    unsigned up = S & 0x550u;                                            // the four edge neighbours
    up |= (!e1 && !e3) ? (S & 0x8u) : 0u;
    up |= (!e1 && !e4) ? (S & 0x20u) : 0u;
    up |= (!e2 && !e3) ? (S & 0x200u) : 0u;
    up |= (!e2 && !e4) ? (S & 0x800u) : 0u;
    sel |= m ? 0x80u : up;
  }
  return sel;
}

// One neighbour per 26-adjacency CLUSTER of the same-level backward neighbours (scripts/gen_glszm_sel13.py has the
// argument): on structured volumes nearly all 13 neighbours of a voxel have its level and form one cluster -- one union
// instead of the two (own plane, plane above) t8_select asks for, and none at all for a voxel whose left neighbour in
// its own quad has its level (bit 12 represents its cluster; the tile kernel ties the two through the initial label).
__device__ const unsigned short t8_sel13[1 << 13] = {
#include "glszm_sel13.inc"
};

// The neighbours (13 bits, numbering above) that are members of T or 26-adjacent to a member of T: rows r0..r2 see each
// other and r2 sees r3 at |dx| <= 1, the voxel to the left (bit 12) sees dx = -1 and 0 of every row.
__device__ __forceinline__ unsigned t8_closed_nbhd(unsigned T) {
  const unsigned W = T & 0xfffu;
  const unsigned U = W | ((W << 1) & 0xdb6u) | ((W >> 1) & 0x6dbu);            // |dx| <= 1 inside every row
  const unsigned g3 = (U >> 9) & 7u, g2 = (U >> 6) & 7u, a = (U | (U >> 3) | (U >> 6)) & 7u;
  unsigned N = a | (a << 3) | ((a | g3) << 6) | ((g2 | g3) << 9);
  if (T & 0x1000u) N |= 0x16dbu;
  if (W & 0x6dbu) N |= 0x1000u;
  return N;
}

// t8_closed_nbhd of two sets at once (13-bit fields at bits 0 and 16 of T)
__device__ __forceinline__ unsigned t8_closed_nbhd2(unsigned T) {
  const unsigned W = T & 0x0fff0fffu;
  const unsigned U = W | ((W << 1) & 0x0db60db6u) | ((W >> 1) & 0x06db06dbu);
  const unsigned a = (U | (U >> 3) | (U >> 6)) & 0x00070007u, g3 = (U >> 9) & 0x00070007u, g2 = (U >> 6) & 0x00070007u;
  unsigned N = a | (a << 3) | ((a | g3) << 6) | ((g2 | g3) << 9);
  N |= ((T >> 12) & 0x00010001u) * 0x16dbu;                                   // the voxel to the left sees dx = -1, 0 of every row
  N |= (((W & 0x06db06dbu) + 0x7fff7fffu) & 0x80008000u) >> 3;                // ... and they see it
  return N;
}

#ifndef PRAD_T8_JUMPS
#define PRAD_T8_JUMPS 3
#endif
#ifndef PRAD_T8_FLOOD
#define PRAD_T8_FLOOD 2
#endif
#ifdef PRAD_T8_PROF
__device__ unsigned long long prad_t8_prof[16];
__global__ void glszm_t8_prof_kernel() {
  unsigned long long tot = 0;
  for (int i = 0; i < 12; i++) tot += prad_t8_prof[i];
  for (int i = 0; i < 12; i++) printf("phase %d: %5.1f %%\n", i, 100.0 * prad_t8_prof[i] / (double)tot);
  for (int i = 0; i < 16; i++) prad_t8_prof[i] = 0;
}
#define PRAD_T8_TICK(i)                                                       \
  do {                                                                        \
    const unsigned long long now_ = __builtin_readcyclecounter();            \
    if (threadIdx.x == 0) atomicAdd(&prad_t8_prof[i], now_ - tick_);          \
    tick_ = now_;                                                             \
  } while (0)
#else
#define PRAD_T8_TICK(i)
#endif
// ---- the dense model of the packed-byte path ------------------------------------------------------------------------
// The tile is loaded WITH its halo of real levels, so a voxel sees the same-level backward neighbours S it has in the
// whole volume.  Those inside the tile (IN) are united here, in LDS.  A neighbour in another tile only matters when its
// 26-adjacency cluster of S has no member inside the tile: the members of a cluster are tied to each other by their own
// (earlier) unions, and the voxel reaches a cluster with a member in the tile through that member.  Clusters of S made of
// other-tile voxels only are components of Y = (S outside the tile) minus everything adjacent to IN; one representative
// of every component of Y goes to the work list (a component of Y that belongs to a cluster reached otherwise costs a
// redundant union, never a wrong one).
//
// Every tile-local component (a "tile root") gets a DENSE id -- its position in the list of tile roots:
//     vid[voxel]   = dense id of the voxel's tile root, -1 outside the ROI      (int32 [n], the only per-voxel output)
//     tinfo[id]    = voxels of the tile component (<= 4096) | grey level << 16   (ONE store per root)
//     worklist     = pairs (dense id of a voxel's tile root, linear index of a same-level neighbour in another tile)
// and glszm_dense_init_kernel, a stream over the ids, sets parent[id] = id and zsize[id] = the component's voxels
// so that what follows -- uniting tile roots across tile faces (glszm_pairs_kernel), folding the counts into the zone
// roots, zone statistics, the fill -- chases pointers in arrays of ~7 M entries (512^3 smooth: 30 MB, cache resident)
// instead of the 537 MB label volume: 18 M pairs x ~7 dependent 4-byte reads were 1.4 - 1.7 ms of random sectors there.
// rootctl[0] = tile roots, rootctl[2] = pairs written, rootctl[3] != 0: the work list was too small and
// glszm_border8d_kernel scans the tile faces instead.
template <int MODE>
__global__ void __launch_bounds__(256) glszm_tile8_kernel(const uint8_t *__restrict__ L, int Nz, int Ny, int Nx,
                                                          int *__restrict__ vid, const int *__restrict__ flags,
                                                          int *__restrict__ rootctl, unsigned *__restrict__ tinfo,
                                                          int2 *__restrict__ worklist, int workcap) {
  __shared__ unsigned lev[PRAD_T8_DW];
  __shared__ int lab[PRAD_TVOX];
  __shared__ int lcount, lbase, wcount, wbase;
  __shared__ int offt[16];                   // in-tile index offset of neighbour n (t8_offset)
  if (flags[0]) return;   // a masked level outside 1..Ng: the int32 kernels redo this call
  if (threadIdx.x < 16) {
    int dz, dy, dx;
    t8_offset(threadIdx.x, dz, dy, dx);
    offt[threadIdx.x] = threadIdx.x < 13 ? dz * (PRAD_TX * PRAD_TY) + dy * PRAD_TX + dx : 0;
  }
#ifdef PRAD_T8_PROF
  unsigned long long tick_ = __builtin_readcyclecounter();
#endif
  const int tx = (Nx + PRAD_TX - 1) / PRAD_TX, ty = (Ny + PRAD_TY - 1) / PRAD_TY;
  const int bz = blockIdx.x / (ty * tx), br = blockIdx.x % (ty * tx);
  const int z0 = bz * PRAD_TZ, y0 = (br / tx) * PRAD_TY, x0 = (br % tx) * PRAD_TX;
  const bool vec = (Nx & 3) == 0;   // every quad is a 4-byte aligned dword of the volume (and of the vid rows)
  if (threadIdx.x == 0) lcount = wcount = 0;
  for (int i = threadIdx.x; i < PRAD_T8_DW; i += blockDim.x) {
    const int c = i % PRAD_T8_ROWDW, rw = (i / PRAD_T8_ROWDW) % (PRAD_TY + 2), pl = i / (PRAD_T8_ROWDW * (PRAD_TY + 2));
    const int z = z0 + pl - 1, y = y0 + rw - 1, x = x0 + 4 * (c - 1);
    unsigned w = 0u;
    if ((MODE == 1 || pl > 0) && (unsigned)z < (unsigned)Nz && (unsigned)y < (unsigned)Ny && x >= 0 && x < Nx) {
      const long long gi = ((long long)z * Ny + y) * Nx + x;
      if (vec) w = *reinterpret_cast<const unsigned *>(L + gi);
      else
        for (int b = 0; b < 4; b++)
          if (x + b < Nx) w |= (unsigned)L[gi + b] << (8 * b);
    }
    lev[i] = w;
  }
  __syncthreads();
  PRAD_T8_TICK(0);
  constexpr int QPT = PRAD_TVOX / 4 / 256;   // quads per lane
  unsigned cw[QPT];
  unsigned long long btodo[QPT];             // bit 16k + n of [q]: voxel k of quad q goes to the work list with its neighbour n
#pragma unroll
  for (int q = 0; q < QPT; q++) {
    const int quad = threadIdx.x + q * 256;
    const int lx4 = quad & 15, ly = (quad >> 4) % PRAD_TY, lz = quad / (16 * PRAD_TY);
    const unsigned w = lev[((lz + 1) * (PRAD_TY + 2) + (ly + 1)) * PRAD_T8_ROWDW + 1 + lx4];
    cw[q] = w;
  }
  // ---- selection.  Every voxel takes ONE of the neighbours it has to be united with (one per 26-adjacency cluster of its
  // same-level backward neighbours inside the tile: t8_sel13) as its initial label -- a plain store to its own cell, links
  // only go to smaller indices -- and keeps the others (`todo`: 0.13 per voxel on smooth volumes where uniting with every
  // representative took 0.78 finds + atomics) for the union phase.
  unsigned long long todo[QPT];              // bit 16k + n of [q]: unite voxel k of quad q with its neighbour n
#pragma unroll
  for (int q = 0; q < QPT; q++) {
    const unsigned centre = cw[q];
    btodo[q] = 0ull;
    todo[q] = 0ull;
    const int quad0 = threadIdx.x + q * 256;
    if (!centre) {
      *reinterpret_cast<int4 *>(lab + quad0 * 4) = make_int4(quad0 * 4, quad0 * 4 + 1, quad0 * 4 + 2, quad0 * 4 + 3);
      continue;
    }
    const int quad = threadIdx.x + q * 256;
    const int lx4 = quad & 15, ly = (quad >> 4) % PRAD_TY, lz = quad / (16 * PRAD_TY);
    const unsigned *row0 = lev + ((lz + 1) * (PRAD_TY + 2) + (ly + 1)) * PRAD_T8_ROWDW + lx4;   // [0] left, [1] mid, [2] right
    // "Same level" flags of the 4 voxels against one neighbour direction at a time, packed-byte arithmetic on the whole
    // dword (bit 7 of byte k: voxel k has the level of that neighbour), gathered into Wlo / Whi: byte k of Wlo = bits 0..7,
    // byte k of Whi = bits 8..12 of voxel k's set S (numbering of t8_offset).  ~30 VALU per voxel where the per-voxel
    // 3-byte windows took ~60; the kernel is bound by instruction issue (4 000 instructions per wave and tile).
    auto eq4 = [](unsigned a, unsigned b) -> unsigned {
      const unsigned x = a ^ b;
      return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
    };
    unsigned Wlo = 0u, Whi = 0u;
#pragma unroll
    for (int r = 0; r < (MODE == 1 ? 4 : 1); r++) {
      const int dz = r == 0 ? 0 : -1, dy = r == 0 ? -1 : r - 2;
      const unsigned *rp = row0 + (dz * (PRAD_TY + 2) + dy) * PRAD_T8_ROWDW;
      const unsigned left = rp[0], mid = rp[1], right = rp[2];
      const unsigned em = eq4(centre, __builtin_amdgcn_alignbyte(mid, left, 3));    // dx = -1
      const unsigned e0 = eq4(centre, mid);
      const unsigned ep = eq4(centre, __builtin_amdgcn_alignbyte(right, mid, 1));   // dx = +1
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int n = 3 * r + d;
        const unsigned e = d == 0 ? em : (d == 1 ? e0 : ep);
        if (n < 8) Wlo |= e >> (7 - n);
        else Whi |= e >> (7 - (n - 8));
      }
    }
    Whi |= eq4(centre, __builtin_amdgcn_alignbyte(centre, row0[0], 3)) >> 3;        // n = 12: the voxel to the left
    {
      const unsigned nz = (((centre & 0x7f7f7f7fu) + 0x7f7f7f7fu) | centre) & 0x80808080u;
      const unsigned inroi = (nz >> 7) * 0xffu;                  // 0xff in the bytes of voxels inside the ROI
      Wlo &= inroi;
      Whi &= inroi;
    }
    // the neighbours of this row that lie in another tile
    const unsigned rowcross = (lz == 0 ? 0xff8u : 0u) | (ly == 0 ? 0x3fu : 0u) | (ly == PRAD_TY - 1 ? 0xe00u : 0u);
    unsigned IN[4], X[4];
    int ilab[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned S = __builtin_amdgcn_ubfe(Wlo, 8 * k, 8) | (__builtin_amdgcn_ubfe(Whi, 8 * k, 5) << 8);
      const unsigned cross = rowcross | ((k == 0 && lx4 == 0) ? 0x1249u : 0u) | ((k == 3 && lx4 == 15) ? 0x924u : 0u);
      X[k] = S & cross;
      IN[k] = S & ~cross;
      // (a set of at most one neighbour is its own selection: only the lanes with two or more go to the table -- a gather
      // from global memory costs the vector L1 one tag look-up per active lane)
      unsigned sel = IN[k];
      if (sel & (sel - 1u)) sel = (unsigned)t8_sel13[sel];
      // the initial label: the voxel to the left when it is one of them (runs along x become chains the jumping rounds
      // flatten), else the first in bit order
      const unsigned pick = (sel & 0x1000u) ? 0x1000u : (sel & (0u - sel));
      ilab[k] = quad * 4 + k + (pick ? offt[__ffs((int)pick) - 1] : 0);
      todo[q] |= (unsigned long long)(sel ^ pick) << (16 * k);
    }
    *reinterpret_cast<int4 *>(lab + quad * 4) = make_int4(ilab[0], ilab[1], ilab[2], ilab[3]);
#if !(defined(PRAD_DBG_T8) && (PRAD_DBG_T8 & 1))     // (ablation builds: bit 0 no work list, 1 no unions, 2 no finds, 3 no stores)
    // the work list: Y = X minus everything adjacent to IN, two voxels per word (fields at bits 0 and 16)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      // (PRAD_T8_FLOOD rounds of "grow through S": 1 = everything adjacent to IN, 4 = the exact clusters)
      const unsigned S2 = (IN[2 * h] | X[2 * h]) | ((IN[2 * h + 1] | X[2 * h + 1]) << 16);
      unsigned N2 = IN[2 * h] | (IN[2 * h + 1] << 16);
#pragma unroll
      for (int r = 0; r < PRAD_T8_FLOOD; r++) N2 = S2 & t8_closed_nbhd2(N2);
      const unsigned Ya = X[2 * h] & ~N2, Yb = X[2 * h + 1] & ~(N2 >> 16);
      unsigned sa = Ya, sb = Yb;
      if (sa & (sa - 1u)) sa = (unsigned)t8_sel13[sa];
      if (sb & (sb - 1u)) sb = (unsigned)t8_sel13[sb];
      btodo[q] |= ((unsigned long long)sa << (32 * h)) | ((unsigned long long)sb << (32 * h + 16));
    }
#endif
  }
  __syncthreads();
  PRAD_T8_TICK(1);
  // ---- pointer jumping: label <- label of the label, every voxel, PRAD_T8_JUMPS rounds (any interleaving of these
  // stores is valid: a label only moves to a smaller member of the same component)
#pragma unroll 1
  for (int round = 0; round < PRAD_T8_JUMPS; round++) {
#pragma unroll
    for (int q = 0; q < QPT; q++) {
      if (!cw[q]) continue;
      int4 *cell = reinterpret_cast<int4 *>(lab + (threadIdx.x + q * 256) * 4);
      const int4 p = *cell;
      volatile lds_int_t *l = (volatile lds_int_t *)lab;
      *cell = make_int4(l[p.x], l[p.y], l[p.z], l[p.w]);
    }
  }
  __syncthreads();
  PRAD_T8_TICK(10);
  // ---- the unions that are left
#pragma unroll
  for (int q = 0; q < QPT; q++) {
    const int quad = threadIdx.x + q * 256;
    unsigned long long t = todo[q];
    while (t) {
      const int bit = __ffsll((long long)t) - 1;
      t &= t - 1;
      const int idx = quad * 4 + (bit >> 4);
      lds_union(lab, idx, idx + offt[bit & 15]);
    }
  }
  __syncthreads();
  PRAD_T8_TICK(2);
  // roots of the 16 voxels of the lane, all chases in flight together
  int root[QPT][4];
  {
    volatile lds_int_t *l = (volatile lds_int_t *)lab;
#pragma unroll
    for (int q = 0; q < QPT; q++) {
      const int4 p = *reinterpret_cast<const int4 *>(lab + (threadIdx.x + q * 256) * 4);
      root[q][0] = p.x; root[q][1] = p.y; root[q][2] = p.z; root[q][3] = p.w;
    }
    bool moving = true;
    while (__any(moving)) {
      moving = false;
#pragma unroll
      for (int q = 0; q < QPT; q++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int nx = l[root[q][k]];
          moving = moving || nx != root[q][k];
          root[q][k] = nx;
        }
    }
#pragma unroll
    for (int q = 0; q < QPT; q++)
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (!((cw[q] >> (8 * k)) & 0xffu)) root[q][k] = -1;
  }
  __syncthreads();
  PRAD_T8_TICK(3);
  unsigned *cnt = reinterpret_cast<unsigned *>(lab);          // the roots are in registers: lab becomes the counts
  for (int k = threadIdx.x; k < PRAD_TVOX; k += blockDim.x) cnt[k] = 0u;
  __syncthreads();
  PRAD_T8_TICK(4);
#pragma unroll
  for (int q = 0; q < QPT; q++) {
    int prev = -1;
    unsigned run = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int r = root[q][k];
      if (r != prev) {
        if (prev >= 0) atomicAdd(cnt + prev, run);
        prev = r;
        run = 0;
      }
      run++;
    }
    if (prev >= 0) atomicAdd(cnt + prev, run);
  }
  __syncthreads();
  PRAD_T8_TICK(5);
  // dense ids: the roots this lane owns, a block of the tile-root list for the tile, a block of the work list
  int nroots = 0, npairs = 0;
#pragma unroll
  for (int q = 0; q < QPT; q++) {
#pragma unroll
    for (int k = 0; k < 4; k++) nroots += root[q][k] == (threadIdx.x + q * 256) * 4 + k ? 1 : 0;
    npairs += __popcll(btodo[q]);
  }
  int mypos = nroots ? atomicAdd(&lcount, nroots) : 0;
  int wpos = npairs ? atomicAdd(&wcount, npairs) : 0;
  __syncthreads();
  PRAD_T8_TICK(6);
  // (one atomic per tile and list on one address each: 65 k at 512^3.  They are only harmless behind __syncthreads(), which
  // also waits for the wave's global stores -- with a barrier that orders LDS traffic alone, or with workgroups that walk
  // many tiles, the same atomics took 3.6 ms; profiles/r04_probes.md section 11)
  if (threadIdx.x == 0) {
    lbase = lcount > 0 ? atomicAdd(rootctl, lcount) : 0;       // (the arrays hold n entries: no overflow)
    wbase = -1;
    if (wcount > 0 && !__builtin_nontemporal_load(rootctl + 3)) {       // (workcap <= 2^30: the counter cannot wrap)
      const int at = atomicAdd(rootctl + 2, wcount);
      if (at < 0 || at > workcap - wcount) atomicOr(rootctl + 3, 1);     // (the list is full: glszm_border8d_kernel takes over)
      else wbase = at;
    }
  }
  __syncthreads();
  PRAD_T8_TICK(7);
  if (nroots) {
    int id = lbase + mypos;
#pragma unroll
    for (int q = 0; q < QPT; q++) {
      const int quad = threadIdx.x + q * 256;
      const int lx4 = quad & 15, ly = (quad >> 4) % PRAD_TY, lz = quad / (16 * PRAD_TY);
      const long long gi = ((long long)(z0 + lz) * Ny + (y0 + ly)) * Nx + (x0 + 4 * lx4);
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (root[q][k] == quad * 4 + k) {
          const unsigned sz = cnt[quad * 4 + k];             // <= 4096
          tinfo[id] = sz | (((cw[q] >> (8 * k)) & 0xffu) << 16);     // (one store per root: four cost 0.75 ms on 512^3 noise)
          lab[quad * 4 + k] = id;                              // (its count is in the dense array now)
          id++;
        }
    }
  }
  __syncthreads();
  PRAD_T8_TICK(8);
  const int plane = Ny * Nx;
#pragma unroll
  for (int q = 0; q < QPT; q++) {
    const int quad = threadIdx.x + q * 256;
    const int lx4 = quad & 15, ly = (quad >> 4) % PRAD_TY, lz = quad / (16 * PRAD_TY);
    const int z = z0 + lz, y = y0 + ly, x = x0 + 4 * lx4;
    if (z >= Nz || y >= Ny || x >= Nx) continue;
    const long long gi = ((long long)z * Ny + y) * Nx + x;
    int lb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) lb[k] = root[q][k] >= 0 ? lab[root[q][k]] : -1;
#if defined(PRAD_DBG_T8) && (PRAD_DBG_T8 & 8)
    if (lb[0] == 0x12345678) vid[gi] = 0;
    else if (false)
#endif
    if (vec) {
      *reinterpret_cast<int4 *>(vid + gi) = make_int4(lb[0], lb[1], lb[2], lb[3]);
    } else {
      for (int k = 0; k < 4; k++)
        if (x + k < Nx) vid[gi + k] = lb[k];
    }
    unsigned long long b = btodo[q];
    if (wbase < 0) b = 0ull;
    while (b) {
      const int bit = __ffsll((long long)b) - 1;
      b &= b - 1;
      int dz, dy, dx;
      t8_offset(bit & 15, dz, dy, dx);
      worklist[wbase + wpos++] = make_int2(lb[bit >> 4], (int)gi + (bit >> 4) + dz * plane + dy * Nx + dx);
    }
  }
  PRAD_T8_TICK(9);
}

// find / union in the dense parent array (smaller id = closer to the root, as in the label volume)
#ifndef PRAD_DN_LOAD
#define PRAD_DN_LOAD 2
#endif
__device__ __forceinline__ int dn_load(const int *p) {
#if PRAD_DN_LOAD == 1      // (probe builds) device-scope load: past this XCD's L2
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif PRAD_DN_LOAD == 2    // plain load
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
  return __builtin_nontemporal_load(p);
#endif
}
// (dn_union does not halve the paths it walks: with thousands of lanes in the same trees the halving stores cost more
// than the shorter chains gain -- 676 -> 612 us at 512^3 smooth; glszm_rootsum_dense_kernel flattens afterwards)
#ifdef PRAD_DN_HALVING
#define PRAD_DN_HALVE(stmt) stmt
#else
#define PRAD_DN_HALVE(stmt)
#endif
__device__ __forceinline__ int dn_find(int *parent, int i) {
  int p = dn_load(parent + i);
  while (p != i) {
    const int g = dn_load(parent + p);
    if (g != p) parent[i] = g;
    i = p;
    p = g;
  }
  return i;
}
__device__ __forceinline__ void dn_union(int *parent, int a, int b) {     // both finds in flight together
  while (true) {
    int pa = dn_load(parent + a), pb = dn_load(parent + b);
    while (pa != a || pb != b) {
      const int ga = dn_load(parent + pa), gb = dn_load(parent + pb);
      if (pa != a) {
        PRAD_DN_HALVE(if (ga != pa) parent[a] = ga;)
        a = pa;
        pa = ga;
      }
      if (pb != b) {
        PRAD_DN_HALVE(if (gb != pb) parent[b] = gb;)
        b = pb;
        pb = gb;
      }
    }
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }
    const int old = atomicMin(parent + a, b);
    if (old == a) return;
    a = old;
  }
}

#ifdef PRAD_PAIRS_STATS
__device__ unsigned long long prad_pairs_dbg[16];
__global__ void glszm_pairs_dbg_kernel() {
  printf("pairs %llu unions %llu same-root %llu hops %llu atomics %llu links %llu maxhops %llu maxtries %llu cycles/union avg %llu max %llu\n",
         prad_pairs_dbg[0], prad_pairs_dbg[1], prad_pairs_dbg[2], prad_pairs_dbg[3], prad_pairs_dbg[4], prad_pairs_dbg[5],
         prad_pairs_dbg[6], prad_pairs_dbg[7], prad_pairs_dbg[1] ? prad_pairs_dbg[8] / prad_pairs_dbg[1] : 0ull, prad_pairs_dbg[9]);
  for (int i = 0; i < 16; i++) prad_pairs_dbg[i] = 0;
}
__device__ __forceinline__ void dn_union_dbg(int *parent, int a, int b) {
  unsigned hops = 0, atom = 0, links = 0, same = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (true) {
    int pa = dn_load(parent + a), pb = dn_load(parent + b);
    while (pa != a || pb != b) {
      const int ga = dn_load(parent + pa), gb = dn_load(parent + pb);
      hops++;
      if (pa != a) { if (ga != pa) parent[a] = ga; a = pa; pa = ga; }
      if (pb != b) { if (gb != pb) parent[b] = gb; b = pb; pb = gb; }
    }
    if (a == b) { same = atom == 0; break; }
    if (a < b) { int t = a; a = b; b = t; }
    const int old = atomicMin(parent + a, b);
    atom++;
    if (old == a) { links++; break; }
    a = old;
  }
  const unsigned long long dt = __builtin_readcyclecounter() - t0;
  atomicAdd(&prad_pairs_dbg[1], 1ull);
  atomicAdd(&prad_pairs_dbg[2], same);
  atomicAdd(&prad_pairs_dbg[3], hops);
  atomicAdd(&prad_pairs_dbg[4], atom);
  atomicAdd(&prad_pairs_dbg[5], links);
  atomicMax(&prad_pairs_dbg[6], (unsigned long long)hops);
  atomicMax(&prad_pairs_dbg[7], (unsigned long long)atom);
  atomicAdd(&prad_pairs_dbg[8], dt);
  atomicMax(&prad_pairs_dbg[9], dt);
}
#define PRAD_DN_UNION dn_union_dbg
#else
#define PRAD_DN_UNION dn_union
#endif
// The work list: one lane per pair.  A pair of tile roots that repeats the previous lane's (neighbouring voxels of the
// same two tile components) is skipped.
__global__ void __launch_bounds__(256) glszm_pairs_kernel(const int2 *__restrict__ worklist,
                                                          const int *__restrict__ rootctl, const int *__restrict__ vid,
                                                          int *__restrict__ parent, const int *__restrict__ flags) {
  if (flags[0] || rootctl[3]) return;
  const int total = rootctl[2];
  const int stride = gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  // the pair of the NEXT round is fetched (list entry, then the neighbour's id: two dependent loads) while this round's
  // union chases its pointers: the kernel is a chain of memory latencies, 97 % of its wave cycles are waits
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int a = -1, b = -1;
  if (j < total) {
    const int2 p = worklist[j];
    a = p.x;
    b = vid[p.y];
  }
  for (int j0 = blockIdx.x * blockDim.x; j0 < total; j0 += stride) {      // (whole waves stay in the loop: shuffles)
    const int jn = j + stride;
    int an = -1, bn = -1;
    if (jn < total) {
      const int2 p = worklist[jn];
      an = p.x;
      bn = vid[p.y];
    }
    const int pa = __shfl_up(a, 1), pb = __shfl_up(b, 1);
    if (a >= 0 && a != b && !(lane > 0 && pa == a && pb == b)) PRAD_DN_UNION(parent, a, b);
    j = jn;
    a = an;
    b = bn;
  }
}

// The route of a full work list: the faces of the tiles scanned on the level volume, one voxel per lane, the redundancy
// rules of glszm_border_full_kernel on GLOBAL sameness (a superset of what the work list would have held).
template <int MODE>
__global__ void __launch_bounds__(256) glszm_border8d_kernel(const uint8_t *__restrict__ L, int Nz, int Ny, int Nx,
                                                             const int *__restrict__ vid, int *__restrict__ parent,
                                                             const int *__restrict__ flags,
                                                             const int *__restrict__ rootctl) {
  if (flags[0] || !rootctl[3]) return;
  const int lane = threadIdx.x & 63;
  const long long nrows = (long long)Nz * Ny;
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const int xtiles = (Nx + PRAD_TX - 1) / PRAD_TX;
  for (long long row = wave0; row < nrows; row += nwaves) {
    const int z = (int)(row / Ny), y = (int)(row - (long long)z * Ny);
    const bool row_edge = (MODE == 1 && z % PRAD_TZ == 0) || (y % PRAD_TY == 0) || (MODE == 1 && y % PRAD_TY == PRAD_TY - 1);
    const int count = row_edge ? Nx : 2 * xtiles;
    const long long rbase = row * Nx;
    for (int c0 = 0; c0 < count; c0 += 64) {
      const int c = c0 + lane;
      int x = -1;
      if (c < count) x = row_edge ? c : (c >> 1) * PRAD_TX + ((c & 1) ? PRAD_TX - 1 : 0);
      int gl = 0, li = -1;
      if (x >= 0 && x < Nx) {
        gl = L[rbase + x];
        if (gl) li = vid[rbase + x];
      }
      if (__ballot(li >= 0) == 0ull) continue;
      auto same = [&](int dz, int dy, int dx) -> long long {
        const int qz = z + dz, qy = y + dy, qx = x + dx;
        if ((unsigned)qz >= (unsigned)Nz || (unsigned)qy >= (unsigned)Ny || (unsigned)qx >= (unsigned)Nx) return -1;
        const long long j = ((long long)qz * Ny + qy) * Nx + qx;
        return L[j] == gl ? j : -1;
      };
      auto pair = [&](long long j, int dz, int dy, int dx) {
        int lj = -1;
        if (j >= 0) {
          const int qz = z + dz, qy = y + dy, qx = x + dx;
          const bool same_tile = qz / PRAD_TZ == z / PRAD_TZ && qy / PRAD_TY == y / PRAD_TY && qx / PRAD_TX == x / PRAD_TX;
          if (!same_tile) lj = vid[j];
        }
        const int pi = __shfl_up(li, 1), pj = __shfl_up(lj, 1);
        if (lj >= 0 && !(lane > 0 && pi == li && pj == lj)) dn_union(parent, li, lj);
      };
      const bool live = li >= 0;
      const long long jb = live ? same(0, -1, 0) : -1;
      pair(jb, 0, -1, 0);
      const bool nb = live && jb < 0;
      const long long jc = nb ? same(0, -1, 1) : -1;
      pair(jc, 0, -1, 1);
      const long long ja = nb ? same(0, -1, -1) : -1;
      pair(ja, 0, -1, -1);
      const long long jd = (nb && ja < 0) ? same(0, 0, -1) : -1;
      pair(jd, 0, 0, -1);
      if (MODE == 1) {
        const long long jm = live ? same(-1, 0, 0) : -1;
        pair(jm, -1, 0, 0);
        const bool nm = live && jm < 0;
        const long long e1 = nm ? same(-1, -1, 0) : -1, e2 = nm ? same(-1, 1, 0) : -1;
        const long long e3 = nm ? same(-1, 0, -1) : -1, e4 = nm ? same(-1, 0, 1) : -1;
        pair(e1, -1, -1, 0);
        pair(e2, -1, 1, 0);
        pair(e3, -1, 0, -1);
        pair(e4, -1, 0, 1);
        pair((nm && e1 < 0 && e3 < 0) ? same(-1, -1, -1) : -1, -1, -1, -1);
        pair((nm && e1 < 0 && e4 < 0) ? same(-1, -1, 1) : -1, -1, -1, 1);
        pair((nm && e2 < 0 && e3 < 0) ? same(-1, 1, -1) : -1, -1, 1, -1);
        pair((nm && e2 < 0 && e4 < 0) ? same(-1, 1, 1) : -1, -1, 1, 1);
      }
    }
  }
}

__global__ void __launch_bounds__(256) glszm_dense_init_kernel(const int *__restrict__ rootctl, const unsigned *__restrict__ tinfo,
                                                               int *__restrict__ parent, unsigned *__restrict__ zsize,
                                                               const int *__restrict__ flags) {
  if (flags[0]) return;
  const int m = rootctl[0], m4 = m >> 2;           // 16 B per lane and access (the streams over the ids: noise has 85 M of them)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int q = gid; q < m4; q += stride) {
    const uint4 ti = reinterpret_cast<const uint4 *>(tinfo)[q];
    reinterpret_cast<int4 *>(parent)[q] = make_int4(4 * q, 4 * q + 1, 4 * q + 2, 4 * q + 3);
    reinterpret_cast<uint4 *>(zsize)[q] = make_uint4(ti.x & 0xffffu, ti.y & 0xffffu, ti.z & 0xffffu, ti.w & 0xffffu);
  }
  for (int j = 4 * m4 + gid; j < m; j += stride) {
    parent[j] = j;
    zsize[j] = tinfo[j] & 0xffffu;
  }
}

// Counts of the tile components folded into their zone roots: one find per tile root in the dense array; contributions
// to the same zone root are combined in an LDS hash table first (a zone that spans thousands of tiles would otherwise
// receive thousands of atomics on one address).  parent[] is flat afterwards (every entry points at its zone root).
#define PRAD_RS_SLOTS 1024
#ifndef PRAD_RS_PROBES
#define PRAD_RS_PROBES 2      // (dense kernel; on noise the table fills up and every further probe is a wasted LDS atomic)
#endif
__global__ void __launch_bounds__(256) glszm_rootsum_dense_kernel(const int *__restrict__ rootctl, int *__restrict__ parent,
                                                                  unsigned *__restrict__ tsize,
                                                                  const int *__restrict__ flags) {
  __shared__ int hkey[PRAD_RS_SLOTS];
  __shared__ unsigned hval[PRAD_RS_SLOTS];
  if (flags[0]) return;
  const int total = rootctl[0];
  const int per = (max(1024, (total + (int)gridDim.x - 1) / (int)gridDim.x) + 3) & ~3;
  const long long lo = (long long)blockIdx.x * per, hi = min((long long)total, lo + per);
  if (lo >= hi) return;
  for (int k = threadIdx.x; k < PRAD_RS_SLOTS; k += blockDim.x) {
    hkey[k] = -1;
    hval[k] = 0u;
  }
  __syncthreads();
  auto fold = [&](int j, int p) {
    if (p == j) return;
    const int r = dn_find(parent, p);
    parent[j] = r;
    const unsigned sz = tsize[j];
    unsigned slot = ((unsigned)r * 2654435761u) >> 22;          // 10 bits
    bool done = false;
    for (int probe = 0; probe < PRAD_RS_PROBES && !done; probe++, slot = (slot + 1) & (PRAD_RS_SLOTS - 1)) {
      const int old = atomicCAS(hkey + slot, -1, r);
      if (old == -1 || old == r) {
        atomicAdd(hval + slot, sz);
        done = true;
      }
    }
    if (!done) atomicAdd(tsize + r, sz);
  };
  // (one id per lane and round: four per lane -- a 16-byte load -- made the kernel slower, 509 -> 628 us on 512^3 noise: the
  // finds and hash updates of a lane's four ids run one after the other)
  for (int j = (int)lo + threadIdx.x; j < (int)hi; j += blockDim.x) fold(j, parent[j]);
  __syncthreads();
  for (int k = threadIdx.x; k < PRAD_RS_SLOTS; k += blockDim.x)
    if (hkey[k] >= 0 && hval[k]) atomicAdd(tsize + hkey[k], hval[k]);
}

// int32 tile path: sizes[] holds the voxel count of every tile-local component at its tile root; fold the counts of
// tile roots that were linked elsewhere into their global root.  One find per (zone, tile) instead of per voxel.
// A zone that spans thousands of tiles would otherwise receive thousands of atomics on one address, so every block
// owns a CONTIGUOUS slab of voxels and first combines contributions to the same root in an LDS hash table.
#ifndef PRAD_RS_CHUNK
#define PRAD_RS_CHUNK (64 * 256)       // voxels per block (256^3: 1024 blocks; with 65536 the kernel ran 256 blocks, 3x slower)
#endif
__global__ void __launch_bounds__(256) glszm_rootsum_kernel(long long n, int *__restrict__ labels,
                                                            unsigned *__restrict__ sizes) {
  __shared__ int hkey[PRAD_RS_SLOTS];
  __shared__ unsigned hval[PRAD_RS_SLOTS];
  const long long lo = (long long)blockIdx.x * PRAD_RS_CHUNK, hi = min(n, lo + PRAD_RS_CHUNK);
  if (lo >= hi) return;
  for (int k = threadIdx.x; k < PRAD_RS_SLOTS; k += blockDim.x) {
    hkey[k] = -1;
    hval[k] = 0u;
  }
  __syncthreads();
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const unsigned sz = sizes[i];
    if (sz == 0u) continue;
    const int l = labels[i];
    if (l == (int)i) continue;
    const int r = uf_find(labels, l);
    labels[i] = r;
    unsigned slot = ((unsigned)r * 2654435761u) >> 22;          // 10 bits
    bool done = false;
    for (int probe = 0; probe < 8 && !done; probe++, slot = (slot + 1) & (PRAD_RS_SLOTS - 1)) {
      const int old = atomicCAS(hkey + slot, -1, r);
      if (old == -1 || old == r) {
        atomicAdd(hval + slot, sz);
        done = true;
      }
    }
    if (!done) atomicAdd(sizes + r, sz);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < PRAD_RS_SLOTS; k += blockDim.x)
    if (hkey[k] >= 0 && hval[k]) atomicAdd(sizes + hkey[k], hval[k]);
}

__global__ void __launch_bounds__(256) glszm_flatten_count_kernel(long long n, int *__restrict__ labels,
                                                                   unsigned *__restrict__ sizes) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const long long nround = ((n + stride - 1) / stride) * stride;  // keep whole waves in the loop for the shuffles
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += stride) {
    int r = -1;
    if (i < n) {
      const int l = labels[i];
      if (l >= 0) {
        r = l;
        int p;
        while ((p = labels[r]) != r) r = p;
        if (r != l) labels[i] = r;
      }
    }
    int left = __shfl_up(r, 1);
    if (lane == 0) left = -2;
    const bool head = (r >= 0) && (r != left);
    const bool brk = (r != left) || lane == 0;          // a run (of any value) starts here
    const unsigned long long B = __ballot(brk);
    if (head) {
      const unsigned long long above = lane == 63 ? 0ull : (B >> (lane + 1));
      const int len = above ? __ffsll((long long)above) : 64 - lane;  // distance to the next run start
      atomicAdd(sizes + r, (unsigned)len);
    }
  }
}

// stats[0] = max zone size, stats64[0] = zone count.  Also records which zone sizes occur, for the compact fill:
// sizes below PRAD_SMALL_SIZES in a bitmap (LDS per block, OR-ed out once), the few larger ones (at most
// n / PRAD_SMALL_SIZES zones) appended to a list.
#define PRAD_SMALL_SIZES 8192
__global__ void __launch_bounds__(256) glszm_stats_kernel(long long n, const int *__restrict__ labels,
                                                          const unsigned *__restrict__ sizes, int *__restrict__ stats,
                                                          unsigned long long *__restrict__ stats64,
                                                          unsigned *__restrict__ small_bits, int *__restrict__ large_list,
                                                          int large_cap, int *__restrict__ large_count,
                                                          const int *__restrict__ flags,
                                                          const int *__restrict__ parent = nullptr,
                                                          const int *__restrict__ rootctl = nullptr) {
  // parent != nullptr: the dense model (glszm_tile8_kernel) -- entry j < rootctl[0] is a zone root when parent[j] == j and
  // `sizes` is tsize[]; otherwise the zone roots are the voxels with labels[i] == i
  __shared__ unsigned bits[PRAD_SMALL_SIZES / 32];
  __shared__ unsigned smx;
  __shared__ unsigned long long scnt;
  if (flags && flags[0]) return;
  for (int k = threadIdx.x; k < PRAD_SMALL_SIZES / 32; k += blockDim.x) bits[k] = 0u;
  if (threadIdx.x == 0) { smx = 0u; scnt = 0ull; }
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned mx = 0;
  unsigned long long cnt = 0;
  auto rootsz = [&](unsigned sz) {
    cnt++;
    mx = max(mx, sz);
    if (sz < PRAD_SMALL_SIZES) {
      const unsigned bit = 1u << (sz & 31);
      if (!(bits[sz >> 5] & bit)) atomicOr(bits + (sz >> 5), bit);
    } else {
      const int pos = atomicAdd(large_count, 1);
      if (pos < large_cap) large_list[pos] = (int)sz;
    }
  };
  auto root = [&](long long i) { rootsz(sizes[i]); };
  if (parent) {
    const long long m = rootctl[0], m4 = m >> 2;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < m4; q += stride) {
      const int4 p = reinterpret_cast<const int4 *>(parent)[q];
      const int j = (int)(q << 2);
      if (p.x == j || p.y == j + 1 || p.z == j + 2 || p.w == j + 3) {
        const uint4 z = reinterpret_cast<const uint4 *>(sizes)[q];
        if (p.x == j) rootsz(z.x);
        if (p.y == j + 1) rootsz(z.y);
        if (p.z == j + 2) rootsz(z.z);
        if (p.w == j + 3) rootsz(z.w);
      }
    }
    for (long long j = (m4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride)
      if (parent[j] == (int)j) root(j);
  } else {
    // the scan is a stream over the labels: 16 B per lane and load
    const long long n4 = n >> 2;
    const int4 *lab4 = reinterpret_cast<const int4 *>(labels);
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
      const int4 l = lab4[q];
      const long long i = q << 2;
      if (l.x == (int)i) root(i);
      if (l.y == (int)i + 1) root(i + 1);
      if (l.z == (int)i + 2) root(i + 2);
      if (l.w == (int)i + 3) root(i + 3);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      if (labels[i] == (int)i) root(i);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    cnt += __shfl_xor(cnt, o);
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicMax(&smx, mx);
    atomicAdd(&scnt, cnt);
  }
  __syncthreads();
  if (threadIdx.x == 0 && scnt) {
    atomicMax(stats, (int)smx);
    atomicAdd(stats64, scnt);
  }
  for (int k = threadIdx.x; k < PRAD_SMALL_SIZES / 32; k += blockDim.x)
    if (bits[k]) atomicOr(small_bits + k, bits[k]);
}

// column of a zone size in the compact matrix: table lookup below PRAD_SMALL_SIZES, binary search above
__device__ __forceinline__ int glszm_rank(unsigned sz, const int *__restrict__ small_rank, int nsmall,
                                          const int *__restrict__ large_sorted, int nlarge) {
  if (sz < PRAD_SMALL_SIZES) return small_rank[sz];
  int lo = 0, hi = nlarge;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((unsigned)large_sorted[mid] < sz) lo = mid + 1;
    else hi = mid;
  }
  return (lo < nlarge && (unsigned)large_sorted[lo] == sz) ? nsmall + lo : -1;
}

// Zone counts concentrate on the smallest sizes (noise-like volumes have millions of 1..10-voxel zones), so the
// first RL size columns of every level are accumulated in LDS per block and flushed once; the rest go to HBM.
__global__ void __launch_bounds__(256) glszm_fill_segment_kernel(long long n, const int *__restrict__ labels,
                                                                 const unsigned *__restrict__ sizes,
                                                                 const int *__restrict__ image, int Ng, int maxRegion,
                                                                 int RL, double *__restrict__ out,
                                                                 int *__restrict__ err,
                                                                 const int *__restrict__ parent = nullptr,
                                                                 const int *__restrict__ rootctl = nullptr,
                                                                 const unsigned *__restrict__ tinfo = nullptr) {
  extern __shared__ unsigned fh[];
  for (int k = threadIdx.x; k < Ng * RL; k += blockDim.x) fh[k] = 0u;
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  const unsigned long long idx_max = (unsigned long long)Ng * maxRegion;
  if (parent) n = rootctl[0];            // the dense model: entries of the tile-root list, levels from tinfo[]
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if ((parent ? parent[i] : labels[i]) != (int)i) continue;
    const int gl = parent ? (int)(tinfo[i] >> 16) : image[i];
    const unsigned sz = sizes[i];
    const unsigned long long idx = (unsigned long long)((long long)(gl - 1) * maxRegion + (long long)sz - 1);
    if (gl <= 0 || idx >= idx_max) {  // cmatrices.c:290-291
      *err = 1;
      continue;
    }
    if (gl <= Ng && sz >= 1u && sz <= (unsigned)RL) atomicAdd(fh + (gl - 1) * RL + (sz - 1), 1u);
    else atomicAdd(out + idx, 1.0);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < Ng * RL; k += blockDim.x)
    if (fh[k]) atomicAdd(out + (size_t)(k / RL) * maxRegion + (k % RL), (double)fh[k]);
}

// compact fill (segment mode): the reference's [Ng][maxRegion] layout is almost entirely zero columns when zones are
// large (a smooth 256^3 volume has maxRegion in the millions but only a few thousand distinct sizes -- at most
// sqrt(2 n) since distinct sizes sum to <= n).  The distinct sizes come from glszm_stats_kernel's bookkeeping.
__global__ void __launch_bounds__(256) glszm_fill_compact_kernel(long long n, const int *__restrict__ labels,
                                                                 const unsigned *__restrict__ sizes,
                                                                 const int *__restrict__ image, int Ng, int k, int RL,
                                                                 const int *__restrict__ small_rank, int nsmall,
                                                                 const int *__restrict__ large_sorted, int nlarge,
                                                                 double *__restrict__ out, int *__restrict__ err,
                                                                 const int *__restrict__ parent = nullptr,
                                                                 const int *__restrict__ rootctl = nullptr,
                                                                 const int *__restrict__ meta = nullptr, int kstride = 0,
                                                                 const unsigned *__restrict__ tinfo = nullptr) {
  extern __shared__ unsigned fh[];
  // meta (glszm_rank_kernel: the distinct sizes were ranked on the device, the host does not know k): nsmall, nlarge, k
  // come from there, the rows of `out` are kstride (the capacity) apart; meta[3] != 0: nothing to fill
  if (meta) {
    if (meta[3]) return;
    nsmall = meta[0];
    nlarge = meta[1];
    k = meta[2];
  } else {
    kstride = k;
  }
  // (dense ids: every workgroup takes a contiguous block of at least 4096 of them, the ones left without any return before
  // they touch their 32 KB histogram -- a structured volume has a few hundred thousand ids, noise has millions)
  long long dlo = 0, dhi = 0;
  if (parent) {
    const long long m = rootctl[0];
    const long long per = (max(4096LL, (m + gridDim.x - 1) / gridDim.x) + 3) & ~3LL;
    dlo = (long long)blockIdx.x * per;
    dhi = min(m, dlo + per);
    if (dlo >= dhi) return;
  }
  for (int q = threadIdx.x; q < Ng * RL; q += blockDim.x) fh[q] = 0u;
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  auto rootls = [&](int gl, unsigned sz) {
    const int r = glszm_rank(sz, small_rank, nsmall, large_sorted, nlarge);
    if (gl <= 0 || gl > Ng || r < 0 || r >= k) {
      *err = 1;
      return;
    }
    if (r < RL) atomicAdd(fh + (gl - 1) * RL + r, 1u);
    else atomicAdd(out + (size_t)(gl - 1) * kstride + r, 1.0);
  };
  auto root = [&](long long i) { rootls(parent ? (int)(tinfo[i] >> 16) : image[i], sizes[i]); };
  if (parent) {      // the dense model (glszm_tile8_kernel): `sizes` is tsize[]; four ids per lane and load
    const long long q1 = dhi >> 2;
    for (long long q = (dlo >> 2) + threadIdx.x; q < q1; q += blockDim.x) {
      const int4 p = reinterpret_cast<const int4 *>(parent)[q];
      const int j = (int)(q << 2);
      if (p.x == j || p.y == j + 1 || p.z == j + 2 || p.w == j + 3) {
        const uint4 z = reinterpret_cast<const uint4 *>(sizes)[q], ti = reinterpret_cast<const uint4 *>(tinfo)[q];
        if (p.x == j) rootls((int)(ti.x >> 16), z.x);
        if (p.y == j + 1) rootls((int)(ti.y >> 16), z.y);
        if (p.z == j + 2) rootls((int)(ti.z >> 16), z.z);
        if (p.w == j + 3) rootls((int)(ti.w >> 16), z.w);
      }
    }
    for (long long j = (q1 << 2) + threadIdx.x; j < dhi; j += blockDim.x)
      if (parent[j] == (int)j) root(j);
  } else {
    const long long n4 = n >> 2;
    const int4 *lab4 = reinterpret_cast<const int4 *>(labels);
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
      const int4 l = lab4[q];
      const long long i = q << 2;
      if (l.x == (int)i) root(i);
      if (l.y == (int)i + 1) root(i + 1);
      if (l.z == (int)i + 2) root(i + 2);
      if (l.w == (int)i + 3) root(i + 3);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      if (labels[i] == (int)i) root(i);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < Ng * RL; q += blockDim.x)
    if (fh[q]) atomicAdd(out + (size_t)(q / RL) * kstride + (q % RL), (double)fh[q]);
}

// ---- the distinct zone sizes ranked ON THE DEVICE (the case pipeline: no host round trip between zones and features) ----
// One workgroup: small sizes from the bitmap (popcount scan), large sizes sorted (bitonic, LDS) and de-duplicated.
// Writes small_rank[PRAD_SMALL_SIZES], large_sorted[nlarge], jvals[k] (the sizes as doubles, ascending: the columns of the
// compact matrix) and meta = {nsmall, nlarge, k, problem}; problem != 0: more large zones than PRAD_RANK_LARGE, more
// distinct sizes than kcap, a zone list beyond the reference's scratch rule (nzones >= 2 Ns: bit 1), irregular levels
// (bit 0) -- the caller then repeats the call on the synchronous route, which reports what the reference reports.
#define PRAD_RANK_LARGE 4096
__global__ void __launch_bounds__(1024) glszm_rank_kernel(const unsigned *__restrict__ small_bits,
                                                          const int *__restrict__ large_list,
                                                          const int *__restrict__ large_count, int large_cap,
                                                          const int *__restrict__ flags,
                                                          const unsigned long long *__restrict__ nzones, long long Ns2,
                                                          int kcap, int *__restrict__ small_rank,
                                                          int *__restrict__ large_sorted, double *__restrict__ jvals,
                                                          int *__restrict__ meta) {
  __shared__ int key[PRAD_RANK_LARGE];
  __shared__ int scan[1024];
  __shared__ int s_nsmall;
  const int t = threadIdx.x;
  int problem = 0;
  if (flags && flags[0]) problem |= 1;
  if ((long long)nzones[0] >= Ns2) problem |= 2;
  // small sizes: word t of the bitmap (PRAD_SMALL_SIZES / 32 = 256 words)
  const int nwords = PRAD_SMALL_SIZES / 32;
  const unsigned w = t < nwords ? small_bits[t] : 0u;
  scan[t] = __popc(w);
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {          // inclusive Hillis-Steele scan
    const int v = t >= o ? scan[t - o] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  if (t == 1023) s_nsmall = scan[t];
  int base = scan[t] - __popc(w);
  __syncthreads();
  const int nsmall = s_nsmall;
  if (t < nwords) {
    for (int b = 0; b < 32; b++) {
      const int sz = t * 32 + b;
      if (w & (1u << b)) {
        small_rank[sz] = base;
        if (base < kcap) jvals[base] = (double)sz;
        base++;
      } else {
        small_rank[sz] = -1;
      }
    }
  }
  // large sizes
  int nl = large_count[0];
  if (nl > large_cap || nl > PRAD_RANK_LARGE) {
    problem |= 4;
    nl = 0;
  }
  int P = 1;
  while (P < nl) P <<= 1;
  for (int i = t; i < P; i += 1024) key[i] = i < nl ? large_list[i] : 0x7fffffff;
  __syncthreads();
  for (int kk = 2; kk <= P; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = t; i < P; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const int a = key[i], b = key[l];
          const bool up = (i & kk) == 0;
          if ((a > b) == up) { key[i] = b; key[l] = a; }
        }
      }
      __syncthreads();
    }
  // unique: element i starts a new value; ranks by a scan over chunks of 1024
  int running = 0;
  for (int c0 = 0; c0 < nl; c0 += 1024) {
    const int i = c0 + t;
    const int first = (i < nl && (i == 0 || key[i] != key[i - 1])) ? 1 : 0;
    scan[t] = first;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int v = t >= o ? scan[t - o] : 0;
      __syncthreads();
      scan[t] += v;
      __syncthreads();
    }
    if (first) {
      const int r = running + scan[t] - 1;
      large_sorted[r] = key[i];
      if (nsmall + r < kcap) jvals[nsmall + r] = (double)key[i];
    }
    running += scan[1023];
    __syncthreads();
  }
  if (t == 0) {
    if (nsmall + running > kcap) problem |= 4;
    meta[0] = nsmall;
    meta[1] = running;
    meta[2] = nsmall + running;
    meta[3] = problem;
  }
}

// out[16] of the feature block <- verdict (0 = fine): meta[3] of the ranking, the fill's error word
__global__ void glszm_verdict_kernel(const int *__restrict__ meta, const int *__restrict__ err, double *__restrict__ out) {
  out[0] = (double)(meta[3] | (err[0] ? 8 : 0));
}

// dense model -> the label-volume view the ordered zone list needs: the first voxel of every zone (a scan of vid[])
__global__ void __launch_bounds__(256) glszm_zmin_kernel(long long n, const int *__restrict__ vid, const int *__restrict__ parent,
                                                         int *__restrict__ zmin) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) {
    const int id = vid[v];
    if (id < 0) continue;
    const int r = parent[id];
    if (zmin[r] > (int)v) atomicMin(zmin + r, (int)v);
  }
}
__global__ void __launch_bounds__(256) glszm_publish_kernel(const int *__restrict__ rootctl, const int *__restrict__ parent,
                                                            const unsigned *__restrict__ tsize, const int *__restrict__ zmin,
                                                            int *__restrict__ labels, unsigned *__restrict__ zsz) {
  const int m = rootctl[0];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x)
    if (parent[j] == j) {
      const int v = zmin[j];
      labels[v] = v;
      zsz[v] = tsize[j];
    }
}

// ordered zone list (tempData parity): block counts -> scan -> scatter
#define PRAD_ZL_BLOCK 1024
__global__ void __launch_bounds__(256) glszm_root_count_kernel(long long n, const int *__restrict__ labels,
                                                               unsigned *__restrict__ block_counts) {
  __shared__ unsigned s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * PRAD_ZL_BLOCK;
  unsigned c = 0;
  for (int k = threadIdx.x; k < PRAD_ZL_BLOCK; k += 256) {
    const long long i = base + k;
    if (i < n && labels[i] == (int)i) c++;
  }
  if (c) atomicAdd(&s, c);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s;
}

// single-workgroup exclusive scan (nblocks is at most n/1024 ~ 2M for a 2^31 array)
__global__ void __launch_bounds__(1024) glszm_scan_kernel(long long nblocks, const unsigned *__restrict__ counts,
                                                          unsigned long long *__restrict__ offsets) {
  __shared__ unsigned long long part[1024];
  const long long per = (nblocks + 1023) / 1024;
  const long long lo = (long long)threadIdx.x * per, hi = min(nblocks, lo + per);
  unsigned long long s = 0;
  for (long long i = lo; i < hi; i++) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int t = 0; t < 1024; t++) {
      unsigned long long v = part[t];
      part[t] = run;
      run += v;
    }
  }
  __syncthreads();
  unsigned long long run = part[threadIdx.x];
  for (long long i = lo; i < hi; i++) {
    offsets[i] = run;
    run += counts[i];
  }
}

__global__ void __launch_bounds__(64) glszm_root_scatter_kernel(long long n, const int *__restrict__ labels,
                                                                const unsigned *__restrict__ sizes,
                                                                const int *__restrict__ image,
                                                                const unsigned long long *__restrict__ offsets,
                                                                int *__restrict__ pairs) {
  // one wave per 1024-voxel block, walking it 64 voxels at a time so the output order is the index order
  const long long base = (long long)blockIdx.x * PRAD_ZL_BLOCK;
  unsigned long long out = offsets[blockIdx.x];
  const int lane = threadIdx.x;
  for (int k = 0; k < PRAD_ZL_BLOCK; k += 64) {
    const long long i = base + k + lane;
    const bool root = i < n && labels[i] == (int)i;
    const unsigned long long m = __ballot(root);
    if (root) {
      const unsigned long long rank = __popcll(m & ((1ull << lane) - 1ull));
      pairs[(out + rank) * 2] = image[i];
      pairs[(out + rank) * 2 + 1] = (int)sizes[i];
    }
    out += __popcll(m);
  }
}

// ---- voxel mode ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) glszm_voxel_kernel(Geo g, VoxMode vm, const int *__restrict__ image,
                                                         const uint8_t *__restrict__ mask,
                                                         const int *__restrict__ angles, int Na,
                                                         uint8_t *__restrict__ visited, int *__restrict__ stack,
                                                         int *__restrict__ zones, int *__restrict__ zone_count,
                                                         int *__restrict__ stats,
                                                         unsigned long long *__restrict__ stats64, int Ns_eff) {
  // (one wave per workgroup: the wave's maximum / zone count are combined in LDS and lane 0 -- active whenever any lane of
  // the wave is -- issues the two global atomics; one pair per CENTRE on these two words serialises in L2)
  __shared__ int smx;
  __shared__ unsigned long long snz;
  if (threadIdx.x == 0) { smx = 0; snz = 0ull; }
  __syncthreads();
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= vm.nvox) return;
  const long long NV = vm.nvox;
  int lo[PRAD_MAX_ND], hi[PRAD_MAX_ND], ext[PRAD_MAX_ND], c[PRAD_MAX_ND];
  kernel_box(g, vm, v, lo, hi);
  long long total = 1;
  for (int d = 0; d < g.nd; d++) {
    ext[d] = hi[d] - lo[d] + 1;
    total *= ext[d];
  }
  for (long long k = 0; k < total; k++) visited[k * NV + v] = 0;
  int nz = 0, mx = 0, nproc = 0;
  for (long long k = 0; k < total; k++) {
    if (visited[k * NV + v]) continue;
    long long i;
    box_decode(g, lo, hi, k, c, &i);
    if (!mask[i]) continue;
    if (nz >= 2 * Ns_eff) { stats[1] = 1; break; }  // cmatrices.c:245 (tempData full)
    const int gl = image[i];
    int region = 0;
    long long top = 0;
    stack[(top++) * NV + v] = (int)k;
    visited[k * NV + v] = 1;
    while (top > 0) {
      const long long kk = stack[(--top) * NV + v];
      region++;
      nproc++;
      long long ii;
      box_decode(g, lo, hi, kk, c, &ii);
      for (int a = 0; a < Na; a++) {
        const int *ang = angles + a * g.nd;
        long long j = 0, kj = 0;
        bool in = true;
        for (int d = 0; d < g.nd; d++) {
          const int q = c[d] + ang[d];
          if (q < lo[d] || q > hi[d]) { in = false; break; }
          j += (long long)q * g.stride[d];
          kj = kj * ext[d] + (q - lo[d]);
        }
        if (!in || visited[kj * NV + v] || !mask[j] || image[j] != gl) continue;
        visited[kj * NV + v] = 1;
        stack[(top++) * NV + v] = (int)kj;
      }
    }
    zones[((long long)nz * 2) * NV + v] = gl;
    zones[((long long)nz * 2 + 1) * NV + v] = region;
    nz++;
    mx = max(mx, region);
  }
  if (nproc > Ns_eff || nz >= 2 * Ns_eff) stats[1] = 1;  // cmatrices.c:174,226 (processedStack) and :274
  zone_count[v] = nz;
  if (nz) {
    atomicMax(&smx, mx);
    atomicAdd(&snz, (unsigned long long)nz);
  }
  __syncthreads();
  if (threadIdx.x == 0 && snz) {
    atomicMax(stats, smx);
    atomicAdd(stats64, snz);
  }
}

__global__ void glszm_fill_voxel_kernel(int nvox, long long boxmax, const int *__restrict__ zones,
                                        const int *__restrict__ zone_count, int Ng, int maxRegion,
                                        double *__restrict__ out, int *__restrict__ err) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long z = gid / nvox;
  const int v = (int)(gid % nvox);
  if (z >= boxmax || z >= zone_count[v]) return;
  const int gl = zones[(z * 2) * nvox + v], sz = zones[(z * 2 + 1) * nvox + v];
  const unsigned long long idx_max = (unsigned long long)Ng * maxRegion;
  const unsigned long long idx = (unsigned long long)((long long)(gl - 1) * maxRegion + (long long)sz - 1);
  if (gl <= 0 || idx >= idx_max) {
    *err = 1;
    return;
  }
  atomicAdd(out + (size_t)v * idx_max + idx, 1.0);
}

__global__ void glszm_gather_zones_kernel(int nvox, int v, int count, const int *__restrict__ zones,
                                          int *__restrict__ pairs) {
  const int z = blockIdx.x * blockDim.x + threadIdx.x;
  if (z >= count) return;
  pairs[z * 2] = zones[((long long)z * 2) * nvox + v];
  pairs[z * 2 + 1] = zones[((long long)z * 2 + 1) * nvox + v];
}

// ---- host drivers --------------------------------------------------------------------------------
// workgroups of the compact fill over the dense ids: every one of them zeroes and flushes a 32 KB LDS histogram, the ids
// are a fraction of the voxels (PRAD_GLSZM_FILLBLOCKS: probe)
inline unsigned glszm_fill_blocks() {
  static const int b = getenv("PRAD_GLSZM_FILLBLOCKS") ? atoi(getenv("PRAD_GLSZM_FILLBLOCKS")) : 2048;
  return (unsigned)std::max(1, b);
}
inline unsigned glszm_grid(long long n) {
  static const int cap = getenv("PRAD_GLSZM_BLOCKS") ? atoi(getenv("PRAD_GLSZM_BLOCKS")) : 8192;
  return (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, cap));
}

inline int glszm_zones(Context &c, hipStream_t s, const Geo &g, const int32_t *image, const uint8_t *mask,
                       const int *angles_h, int Na, int Ng, int Ns, int Nvox, const int *voxels_dev, int kernelRadius,
                       int force2Ddim, long long *nzones_out, bool enqueue_only = false) {
  // enqueue_only (prad_glszm_features_dev): the packed-byte tile path or nothing (PRAD_E_UNSUPPORTED before any launch);
  // no copy back, no synchronisation -- zone count, largest zone and the flags stay on the device ("glszm_stats", "flags")
  GlszmState &st = glszm_state();
  st.valid = false;
  st.published = false;
  int *stats = nullptr;
  PRAD_TRY(c.get<int>("glszm_stats", 8, &stats));
  unsigned long long *stats64 = (unsigned long long *)(stats + 2);
  PRAD_HIP(hipMemsetAsync(stats, 0, sizeof(int) * 8, s));
  void *hp = nullptr;
  PRAD_TRY(c.get_pinned("glszm_stats_h", sizeof(int) * 8, &hp));
  int *stats_h = (int *)hp;
  bool stats_copied = false;

  if (!voxels_dev) {
    if (Nvox != 1) return fail(PRAD_E_ARG, "Nvox=%d without a voxel list", Nvox);
    int *angles_d = nullptr;
    PRAD_TRY(c.get<int>("angles", (size_t)Na * g.nd, &angles_d));
    PRAD_HIP(hipMemcpyAsync(angles_d, angles_h, sizeof(int) * Na * g.nd, hipMemcpyHostToDevice, s));
    PRAD_TRY(c.get<int>("glszm_labels", (size_t)g.n, &st.labels));
    PRAD_TRY(c.get<unsigned>("glszm_sizes", (size_t)g.n, &st.sizes));
    st.idcap = (size_t)g.n;
    st.large_cap = (int)(g.n / PRAD_SMALL_SIZES + 2);
    PRAD_TRY(c.get<unsigned>("glszm_small_bits", PRAD_SMALL_SIZES / 32, &st.small_bits));
    PRAD_TRY(c.get<int>("glszm_large_list", (size_t)st.large_cap, &st.large_list));
    PRAD_TRY(c.get<int>("glszm_large_count", 1, &st.large_count));
    PRAD_TRY(ZeroBatch().add(st.small_bits, PRAD_SMALL_SIZES / 8).add(st.large_count, sizeof(int)).launch(s));
    Timed t(c, "glszm", s);
    // tiled path for <= 3-D volumes whose neighbour offsets are unit steps; generic union-find otherwise
    Offsets3 A3;
    A3.na = 0;
    bool tiled = g.nd <= 3 && Na <= 64;
    int dims3[3] = {1, 1, 1};
    for (int d = 0; d < g.nd && tiled; d++) dims3[3 - g.nd + d] = g.size[d];
    for (int a = 0; a < Na && tiled; a++) {
      int o[3] = {0, 0, 0};
      for (int d = 0; d < g.nd; d++) o[3 - g.nd + d] = angles_h[a * g.nd + d];
      if (o[0] < -1 || o[0] > 1 || o[1] < -1 || o[1] > 1 || o[2] < -1 || o[2] > 1) { tiled = false; break; }
      const long long lin = ((long long)o[0] * dims3[1] + o[1]) * dims3[2] + o[2];
      if (lin >= 0) continue;                 // forward neighbour: its pair is handled from the other end
      if (A3.na >= 32) { tiled = false; break; }
      for (int d = 0; d < 3; d++) A3.o[A3.na][d] = (signed char)o[d];
      A3.o[A3.na][3] = 0;
      A3.na++;
    }
    // all 13 (all 4 in-plane) backward unit offsets present => the redundancy rules of the tile kernels apply
    int mode = 0;
    if (tiled) {
      if (A3.na == 13) mode = 1;
      else if (A3.na == 4) {
        mode = 2;
        for (int a = 0; a < 4; a++)
          if (A3.o[a][0] != 0) mode = 0;
      }
    }
    // Full neighbourhoods with levels that fit a byte run on the packed uint8 volume; a masked level outside 1..Ng
    // (flags[0], found by the pack) sends the call through the int32 kernels instead.
    int *flags_d = nullptr, *flags_h = nullptr;
    bool bytes = tiled && mode != 0 && Ng >= 1 && Ng <= 255;
    if (bytes) {
      PRAD_TRY(c.get<int>("flags", 4, &flags_d));
      void *fp = nullptr;
      PRAD_TRY(c.get_pinned("glszm_flags_h", sizeof(int) * 4, &fp));
      flags_h = (int *)fp;
    }
    if (enqueue_only && !bytes) return fail(PRAD_E_UNSUPPORTED, "GLSZM: the enqueue-only route needs the packed-byte tile kernels");
    for (int attempt = 0; attempt < 2; attempt++) {
      if (tiled) {
        const long long tiles = (long long)((dims3[0] + PRAD_TZ - 1) / PRAD_TZ) * ((dims3[1] + PRAD_TY - 1) / PRAD_TY) *
                                ((dims3[2] + PRAD_TX - 1) / PRAD_TX);
        const dim3 bgrid((unsigned)std::min<long long>(((long long)dims3[0] * dims3[1] + 3) / 4, 16384));
        if (bytes) {
          // the dense model: glszm_tile8_kernel lists the cross-tile pairs (it holds the levels with their halo in LDS),
          // glszm_pairs_kernel unites the tile roots; a full work list (decided on the device: rootctl[3]) sends the
          // call through glszm_border8d_kernel instead.  PRAD_GLSZM_WORKCAP: a test hook -- a tiny list forces that route.
          uint8_t *levels = nullptr;
          const char *wcap = getenv("PRAD_GLSZM_WORKCAP");
          const int workcap = wcap ? std::max(1, atoi(wcap)) : (int)std::min<long long>(1LL << 30, std::max<long long>(g.n, 65536));
          int2 *work = nullptr;
          // lanes of glszm_pairs_kernel: MORE of them are slower on structured volumes (they collide in the trees of the
          // large zones: 256^3 smooth 273 us with 2048 workgroups, 121 with 384; 512^3 790 / 612 with 768)
          static const int pgrid_env = getenv("PRAD_GLSZM_PGRID") ? atoi(getenv("PRAD_GLSZM_PGRID")) : 0;
          const int pgrid = pgrid_env > 0 ? pgrid_env : (int)std::min(2048.0, std::max(64.0, 1.5 * std::cbrt((double)g.n)));
          PRAD_TRY(c.get<int>("glszm_parent", (size_t)g.n, &st.parent));
          PRAD_TRY(c.get<unsigned>("glszm_tinfo", (size_t)g.n, &st.tinfo));
          PRAD_TRY(c.get<int>("glszm_rootctl", 4, &st.rootctl));
          PRAD_TRY(c.get<int2>("glszm_worklist", (size_t)workcap, &work));
          PRAD_TRY(ZeroBatch().add(flags_d, sizeof(int) * 4).add(st.rootctl, sizeof(int) * 4).launch(s));
          PRAD_TRY(neigh_pack(&c, s, g, image, mask, Ng, flags_d, &levels));
#define PRAD_GLSZM_LAUNCH(M)                                                                                                 \
          do {                                                                                                               \
            hipLaunchKernelGGL(glszm_tile8_kernel<M>, dim3((unsigned)tiles), dim3(256), 0, s, levels, dims3[0], dims3[1],    \
                               dims3[2], st.labels, (const int *)flags_d, st.rootctl, st.tinfo, work, workcap);              \
            hipLaunchKernelGGL(glszm_dense_init_kernel, dim3(2048), dim3(256), 0, s, (const int *)st.rootctl,                \
                               (const unsigned *)st.tinfo, st.parent, st.sizes, (const int *)flags_d);                       \
            hipLaunchKernelGGL(glszm_pairs_kernel, dim3(pgrid), dim3(256), 0, s, (const int2 *)work, (const int *)st.rootctl, \
                               (const int *)st.labels, st.parent, (const int *)flags_d);                                     \
            hipLaunchKernelGGL(glszm_border8d_kernel<M>, bgrid, dim3(256), 0, s, levels, dims3[0], dims3[1], dims3[2],       \
                               (const int *)st.labels, st.parent, (const int *)flags_d, (const int *)st.rootctl);            \
          } while (0)
          if (mode == 1) PRAD_GLSZM_LAUNCH(1);
          else PRAD_GLSZM_LAUNCH(2);
#undef PRAD_GLSZM_LAUNCH
#ifdef PRAD_T8_PROF
          hipLaunchKernelGGL(glszm_t8_prof_kernel, dim3(1), dim3(1), 0, s);
#endif
#ifdef PRAD_PAIRS_STATS
          hipLaunchKernelGGL(glszm_pairs_dbg_kernel, dim3(1), dim3(1), 0, s);
#endif
          hipLaunchKernelGGL(glszm_rootsum_dense_kernel, dim3(2048), dim3(256), 0, s, (const int *)st.rootctl, st.parent,
                             st.sizes, (const int *)flags_d);
          PRAD_TRY(check_launch("glszm_tile8/pairs/rootsum_kernel"));
        } else {
          st.parent = st.rootctl = nullptr;
          st.tinfo = nullptr;
          hipLaunchKernelGGL(glszm_tile_kernel, dim3((unsigned)tiles), dim3(256), 0, s, A3, mode, image, mask, dims3[0],
                             dims3[1], dims3[2], st.labels, st.sizes);
          PRAD_TRY(check_launch("glszm_tile_kernel"));
          if (mode == 1)
            hipLaunchKernelGGL(glszm_border_full_kernel<1>, bgrid, dim3(256), 0, s, image, mask, dims3[0], dims3[1], dims3[2], st.labels);
          else if (mode == 2)
            hipLaunchKernelGGL(glszm_border_full_kernel<2>, bgrid, dim3(256), 0, s, image, mask, dims3[0], dims3[1], dims3[2], st.labels);
          else
            hipLaunchKernelGGL(glszm_border_kernel, bgrid, dim3(256), 0, s, A3, image, mask, dims3[0], dims3[1], dims3[2], st.labels);
          PRAD_TRY(check_launch("glszm_border_kernel"));
        }
        if (!bytes) {
          hipLaunchKernelGGL(glszm_rootsum_kernel, dim3((unsigned)((g.n + PRAD_RS_CHUNK - 1) / PRAD_RS_CHUNK)), dim3(256), 0, s,
                             g.n, st.labels, st.sizes);
          PRAD_TRY(check_launch("glszm_rootsum_kernel"));
        }
      } else {
        st.parent = st.rootctl = nullptr;
        st.tinfo = nullptr;
        hipLaunchKernelGGL(glszm_init_kernel, dim3(glszm_grid(g.n)), dim3(256), 0, s, mask, g.n, st.labels, st.sizes);
        PRAD_TRY(check_launch("glszm_init_kernel"));
        hipLaunchKernelGGL(glszm_merge_kernel, dim3(glszm_grid(g.n)), dim3(256), 0, s, g, angles_d, Na, image, mask, st.labels);
        PRAD_TRY(check_launch("glszm_merge_kernel"));
        hipLaunchKernelGGL(glszm_flatten_count_kernel, dim3(glszm_grid(g.n)), dim3(256), 0, s, g.n, st.labels, st.sizes);
        PRAD_TRY(check_launch("glszm_flatten_count_kernel"));
      }
      hipLaunchKernelGGL(glszm_stats_kernel, dim3(std::min(glszm_grid(g.n), 1024u)), dim3(256), 0, s, g.n, st.labels,
                         st.sizes, stats, stats64, st.small_bits, st.large_list, st.large_cap, st.large_count,
                         (const int *)(bytes ? flags_d : nullptr), (const int *)(bytes ? st.parent : nullptr),
                         (const int *)st.rootctl);
      PRAD_TRY(check_launch("glszm_stats_kernel"));
      if (!bytes || enqueue_only) break;
      PRAD_HIP(hipMemcpyAsync(flags_h, flags_d, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
      PRAD_HIP(hipMemcpyAsync(stats_h, stats, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
      PRAD_HIP(hipStreamSynchronize(s));
      stats_copied = true;
      if (!flags_h[0]) break;
      bytes = false;   // irregular levels: nothing was computed, run the int32 kernels
      stats_copied = false;
    }
    st.voxel_mode = false;
    st.image = image;
    st.boxmax = g.n;
    if (enqueue_only) {
      st.g = g;
      st.nvox = 1;
      st.device = c.device;
      return 0;                  // (st.valid stays false: the host-side follow-ups need the synchronous call)
    }
  } else {
    if (kernelRadius <= 0) return fail(PRAD_E_ARG, "Expecting kernelRadius > 0");
    VoxMode vm;
    vm.nvox = Nvox;
    vm.voxels = voxels_dev;
    vm.radius = kernelRadius;
    vm.f2d = force2Ddim;
    long long b = 1;
    for (int d = 0; d < g.nd; d++)
      if (d != force2Ddim) b *= std::min<long long>(2LL * kernelRadius + 1, g.size[d]);
    vm.boxmax = b;
    int *angles_d = nullptr;
    PRAD_TRY(c.get<int>("angles", (size_t)Na * g.nd, &angles_d));
    PRAD_HIP(hipMemcpyAsync(angles_d, angles_h, sizeof(int) * Na * g.nd, hipMemcpyHostToDevice, s));
    // _cmatrices.c:298-314: Ns is clipped to the nominal kernel volume (2r+1)^(Nd or Nd-1)
    long long nominal = 1;
    for (int d = 0; d < (force2Ddim >= 0 ? g.nd - 1 : g.nd); d++) nominal *= (2LL * kernelRadius + 1);
    const int Ns_eff = (int)std::min<long long>(Ns, nominal);
    uint8_t *visited = nullptr;
    int *stack = nullptr;
    PRAD_TRY(c.get<uint8_t>("glszm_visited", (size_t)b * Nvox, &visited));
    PRAD_TRY(c.get<int>("glszm_stack", (size_t)b * Nvox, &stack));
    PRAD_TRY(c.get<int>("glszm_zones", (size_t)b * 2 * Nvox, &st.zones));
    PRAD_TRY(c.get<int>("glszm_zone_count", (size_t)Nvox, &st.zone_count));
    Timed t(c, "glszm", s);
    hipLaunchKernelGGL(glszm_voxel_kernel, dim3((unsigned)((Nvox + 63) / 64)), dim3(64), 0, s, g, vm, image, mask,
                       angles_d, Na, visited, stack, st.zones, st.zone_count, stats, stats64, Ns_eff);
    PRAD_TRY(check_launch("glszm_voxel_kernel"));
    st.voxel_mode = true;
    st.boxmax = b;
  }
  if (!stats_copied) {
    PRAD_HIP(hipMemcpyAsync(stats_h, stats, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
    PRAD_HIP(hipStreamSynchronize(s));
  }
  st.g = g;
  st.nvox = Nvox;
  st.device = c.device;
  st.max_region = stats_h[0];
  st.nsizes = -1;
  st.nzones = (long long)*(unsigned long long *)(stats_h + 2);
  st.valid = true;
  if (nzones_out) *nzones_out = st.nzones;
  // scratch-exhaustion rule of the reference (see include/pyradiomics_amd.h)
  if (!st.voxel_mode ? st.nzones >= 2LL * Ns : stats_h[1] != 0)
    return fail(PRAD_E_INDEX, "GLSZM: zone list would overflow the reference's Ns-sized scratch (Ns=%d)", Ns);
  c.last_path = st.voxel_mode ? "glszm-voxel" : "glszm-unionfind";
  return st.max_region;
}

inline int glszm_fill(Context &c, hipStream_t s, double *out_dev, int Nvox, int Ng, int maxRegion) {
  GlszmState &st = glszm_state();
  if (!st.valid || st.device != c.device) return fail(PRAD_E_ARG, "prad_fill_glszm without a preceding prad_calculate_glszm");
  if (Nvox != st.nvox) return fail(PRAD_E_ARG, "fill_glszm: Nvox=%d but the zone list holds %d kernels", Nvox, st.nvox);
  int *err = nullptr;
  PRAD_TRY(c.get<int>("glszm_err", 4, &err));
  PRAD_HIP(hipMemsetAsync(err, 0, sizeof(int) * 4, s));
  PRAD_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * (size_t)Nvox * Ng * maxRegion, s));
  if (!st.voxel_mode) {
    const int RL = Ng <= 8192 ? std::max(1, std::min(maxRegion, 8192 / Ng)) : 0;
    hipLaunchKernelGGL(glszm_fill_segment_kernel, dim3(std::min(glszm_grid(st.g.n), 2048u)), dim3(256),
                       sizeof(unsigned) * Ng * RL, s, st.g.n, st.labels, st.sizes, st.image, Ng, maxRegion, RL, out_dev,
                       err, (const int *)st.parent, (const int *)st.rootctl, (const unsigned *)st.tinfo);
    PRAD_TRY(check_launch("glszm_fill_segment_kernel"));
  } else {
    const long long threads = (long long)st.nvox * st.boxmax;
    hipLaunchKernelGGL(glszm_fill_voxel_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, st.nvox,
                       st.boxmax, st.zones, st.zone_count, Ng, maxRegion, out_dev, err);
    PRAD_TRY(check_launch("glszm_fill_voxel_kernel"));
  }
  void *hp = nullptr;
  PRAD_TRY(c.get_pinned("glszm_err_h", sizeof(int) * 4, &hp));
  PRAD_HIP(hipMemcpyAsync(hp, err, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  return ((int *)hp)[0] ? PRAD_INDEX_ERROR : PRAD_OK;
}

// distinct zone sizes of the preceding segment-mode phase-1 call, ascending, to the host; the size -> column map
// stays on the device for glszm_fill_compact
inline int glszm_distinct_sizes(Context &c, int *sizes_host, int cap) {
  GlszmState &st = glszm_state();
  if (!st.valid || st.device != c.device) return fail(PRAD_E_ARG, "prad_glszm_sizes without a preceding prad_calculate_glszm");
  if (st.voxel_mode) return fail(PRAD_E_UNSUPPORTED, "prad_glszm_sizes: segment mode only");
  if (!sizes_host || cap < 1) return fail(PRAD_E_ARG, "prad_glszm_sizes: bad buffer");
  hipStream_t s = c.own_stream;
  std::vector<unsigned> bits(PRAD_SMALL_SIZES / 32);
  int nl = 0;
  PRAD_HIP(hipMemcpyAsync(bits.data(), st.small_bits, PRAD_SMALL_SIZES / 8, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipMemcpyAsync(&nl, st.large_count, sizeof(int), hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  if (nl > st.large_cap) return fail(PRAD_E_ARG, "prad_glszm_sizes: internal list overflow (%d > %d)", nl, st.large_cap);
  std::vector<int> large((size_t)nl);
  if (nl) {
    PRAD_HIP(hipMemcpyAsync(large.data(), st.large_list, sizeof(int) * nl, hipMemcpyDeviceToHost, s));
    PRAD_HIP(hipStreamSynchronize(s));
    std::sort(large.begin(), large.end());
    large.erase(std::unique(large.begin(), large.end()), large.end());
  }
  std::vector<int> rank(PRAD_SMALL_SIZES, -1);
  int k = 0;
  for (int sz = 0; sz < PRAD_SMALL_SIZES; sz++)
    if (bits[sz >> 5] & (1u << (sz & 31))) {
      if (k < cap) sizes_host[k] = sz;
      rank[sz] = k++;
    }
  const int nsmall = k;
  for (int v : large) {
    if (k < cap) sizes_host[k] = v;
    k++;
  }
  if (k > cap) return fail(PRAD_E_ARG, "prad_glszm_sizes: %d distinct sizes exceed capacity %d", k, cap);
  int *large_d = nullptr;
  PRAD_TRY(c.get<int>("glszm_small_rank", PRAD_SMALL_SIZES, &st.small_rank));
  PRAD_TRY(c.get<int>("glszm_large_sorted", large.size() + 1, &large_d));
  PRAD_HIP(hipMemcpyAsync(st.small_rank, rank.data(), sizeof(int) * PRAD_SMALL_SIZES, hipMemcpyHostToDevice, s));
  if (!large.empty())
    PRAD_HIP(hipMemcpyAsync(large_d, large.data(), sizeof(int) * large.size(), hipMemcpyHostToDevice, s));
  PRAD_HIP(hipStreamSynchronize(s));
  st.nsizes = k;
  st.nsmall = nsmall;
  st.nlarge = (int)large.size();
  return k;
}

inline int glszm_fill_compact(Context &c, hipStream_t s, double *out_dev, int Ng, int k) {
  GlszmState &st = glszm_state();
  if (!st.valid || st.device != c.device || st.voxel_mode || st.nsizes < 0)
    return fail(PRAD_E_ARG, "prad_fill_glszm_compact without a preceding prad_glszm_sizes");
  if (k != st.nsizes) return fail(PRAD_E_ARG, "fill_glszm_compact: k=%d but %d distinct sizes were found", k, st.nsizes);
  int *err = nullptr, *large_d = nullptr;
  PRAD_TRY(c.get<int>("glszm_err", 4, &err));
  PRAD_TRY(c.get<int>("glszm_large_sorted", (size_t)st.nlarge + 1, &large_d));
  PRAD_TRY(ZeroBatch().add(err, sizeof(int) * 4).add(out_dev, sizeof(double) * (size_t)Ng * std::max(k, 1)).launch(s));
  if (k) {
    const int RL = Ng <= 8192 ? std::max(1, std::min(k, 8192 / Ng)) : 0;
    hipLaunchKernelGGL(glszm_fill_compact_kernel, dim3(std::min(glszm_grid(st.g.n), st.parent ? glszm_fill_blocks() : 2048u)), dim3(256),
                       sizeof(unsigned) * Ng * RL, s, st.g.n, st.labels, st.sizes, st.image, Ng, k, RL, st.small_rank,
                       st.nsmall, large_d, st.nlarge, out_dev, err, (const int *)st.parent, (const int *)st.rootctl,
                       (const int *)nullptr, 0, (const unsigned *)st.tinfo);
    PRAD_TRY(check_launch("glszm_fill_compact_kernel"));
  }
  void *hp = nullptr;
  PRAD_TRY(c.get_pinned("glszm_err_h", sizeof(int) * 4, &hp));
  PRAD_HIP(hipMemcpyAsync(hp, err, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  return ((int *)hp)[0] ? PRAD_INDEX_ERROR : PRAD_OK;
}

inline long long glszm_copy_zones(Context &c, int v, int *tempData, long long capacity_pairs) {
  GlszmState &st = glszm_state();
  if (!st.valid || st.device != c.device) return fail(PRAD_E_ARG, "prad_glszm_zones without a preceding prad_calculate_glszm");
  if (!tempData || v < 0 || v >= st.nvox) return fail(PRAD_E_ARG, "prad_glszm_zones: bad kernel index or NULL buffer");
  hipStream_t s = c.own_stream;
  long long count = 0;
  int *pairs = nullptr;
  if (st.voxel_mode) {
    int cnt = 0;
    PRAD_HIP(hipMemcpy(&cnt, st.zone_count + v, sizeof(int), hipMemcpyDeviceToHost));
    count = cnt;
    if (count > capacity_pairs) return fail(PRAD_E_ARG, "prad_glszm_zones: %lld zones exceed capacity %lld", count, capacity_pairs);
    PRAD_TRY(c.get<int>("glszm_pairs", (size_t)count * 2 + 2, &pairs));
    if (count) {
      hipLaunchKernelGGL(glszm_gather_zones_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, st.nvox, v,
                         (int)count, st.zones, pairs);
      PRAD_TRY(check_launch("glszm_gather_zones_kernel"));
    }
  } else {
    count = st.nzones;
    if (count > capacity_pairs) return fail(PRAD_E_ARG, "prad_glszm_zones: %lld zones exceed capacity %lld", count, capacity_pairs);
    const long long n = st.g.n, nblocks = (n + PRAD_ZL_BLOCK - 1) / PRAD_ZL_BLOCK;
    unsigned *counts = nullptr;
    unsigned long long *offsets = nullptr;
    PRAD_TRY(c.get<unsigned>("glszm_bcounts", (size_t)nblocks, &counts));
    PRAD_TRY(c.get<unsigned long long>("glszm_boffsets", (size_t)nblocks, &offsets));
    PRAD_TRY(c.get<int>("glszm_pairs", (size_t)count * 2 + 2, &pairs));
    const unsigned *vsizes = st.sizes;
    if (st.parent) {
      // the dense model: vid[] has done its work (the unions are over) -- the label volume becomes the view this scan needs:
      // labels[v] == v at the first voxel v of every zone, -1 elsewhere, the zone sizes in a voxel-indexed scratch array
      int *zmin = nullptr;
      unsigned *zsz = nullptr;
      PRAD_TRY(c.get<int>("glszm_zmin", st.idcap, &zmin));
      PRAD_TRY(c.get<unsigned>("glszm_zsz", (size_t)n, &zsz));
      if (!st.published) {                       // (st.labels is vid[] only until the first call)
        PRAD_HIP(hipMemsetAsync(zmin, 0x7f, sizeof(int) * st.idcap, s));
        hipLaunchKernelGGL(glszm_zmin_kernel, dim3(2048), dim3(256), 0, s, n, (const int *)st.labels, (const int *)st.parent, zmin);
        st.published = true;
      }
      PRAD_HIP(hipMemsetAsync(st.labels, 0xff, sizeof(int) * (size_t)n, s));
      hipLaunchKernelGGL(glszm_publish_kernel, dim3(1024), dim3(256), 0, s, (const int *)st.rootctl, (const int *)st.parent,
                         (const unsigned *)st.sizes, (const int *)zmin, st.labels, zsz);
      PRAD_TRY(check_launch("glszm_publish_kernel"));
      vsizes = zsz;
    }
    hipLaunchKernelGGL(glszm_root_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, n, st.labels, counts);
    PRAD_TRY(check_launch("glszm_root_count_kernel"));
    hipLaunchKernelGGL(glszm_scan_kernel, dim3(1), dim3(1024), 0, s, nblocks, counts, offsets);
    PRAD_TRY(check_launch("glszm_scan_kernel"));
    hipLaunchKernelGGL(glszm_root_scatter_kernel, dim3((unsigned)nblocks), dim3(64), 0, s, n, st.labels, vsizes,
                       st.image, offsets, pairs);
    PRAD_TRY(check_launch("glszm_root_scatter_kernel"));
  }
  if (count) PRAD_HIP(hipMemcpyAsync(tempData, pairs, sizeof(int) * 2 * count, hipMemcpyDeviceToHost, s));
  PRAD_HIP(hipStreamSynchronize(s));
  tempData[count * 2] = -1;  // cmatrices.c:276
  return count;
}

}  // namespace prad
