// kernels_neigh.h -- segment-mode GLDM and NGTDM on gfx950: one lane per ROI voxel, the 26 (or 8, or any
// small set of) neighbours gathered from the packed uint8 level volume, workgroup-private histograms in LDS.
//
//   GLDM  (cmatrices.c:660-754)  bins [Ng][Na+1] u32  (dependence counts 0..Na; the reference's row stride
//                                2*Na+1 only matters for the output layout, columns > Na stay zero)
//   NGTDM (cmatrices.c:543-658)  bins [Ng][Na+1] u64: slot 0 = voxels of that level,
//                                slot c = sum over voxels with c valid neighbours of |c*level - sum(neigh)|
//                                => s_i = sum_c bins[i][c] / c  (exact integers inside, one division per c)
// Inputs outside [1,Ng] under the mask are detected by pack_levels (flags[0]) and rerouted to the
// generic kernels by the caller.
#pragma once
#include <algorithm>
#include "prad_runtime.h"
#include "kernels_sweep.h"
#include "kernels_generic.h"

namespace prad {

#define PRAD_MAX_NEIGH 128
struct NeighSet {
  int na;
  signed char o[PRAD_MAX_NEIGH][4];  // (dz, dy, dx, unused), volume embedded in 3-D
};

template <bool NGTDM>
__global__ void __launch_bounds__(256) neigh_kernel(NeighSet A, const uint8_t *__restrict__ L, int Nz, int Ny,
                                                    int Nx, int zlo, int zhi, int Ng, int alpha,
                                                    u32 *__restrict__ gldm_acc, u64 *__restrict__ ngtdm_acc,
                                                    const int *__restrict__ flags) {
  extern __shared__ u64 lds64[];
  if (flags[0]) return;
  const int W = A.na + 1;
  u32 *h32 = reinterpret_cast<u32 *>(lds64);
  const int nbins = Ng * W;
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
    if (NGTDM) lds64[i] = 0;
    else h32[i] = 0;
  }
  __syncthreads();
  const long long plane = (long long)Ny * Nx;
  const long long n = (long long)zhi * plane;   // centres: planes zlo .. zhi-1 (neighbours: the whole volume)
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)zlo * plane + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = L[i];
    if (!c) continue;
    const int z = (int)(i / plane);
    const int r = (int)(i - (long long)z * plane);
    const int y = r / Nx;
    const int x = r - y * Nx;
    int cnt = 0, sum = 0, dep = 0;
    for (int a = 0; a < A.na; a++) {
      const int zz = z + A.o[a][0], yy = y + A.o[a][1], xx = x + A.o[a][2];
      if ((unsigned)zz >= (unsigned)Nz || (unsigned)yy >= (unsigned)Ny || (unsigned)xx >= (unsigned)Nx) continue;
      const int v = L[(long long)zz * plane + (long long)yy * Nx + xx];
      if (!v) continue;
      if (NGTDM) {
        cnt++;
        sum += v;
      } else {
        int d = c - v;
        d = d < 0 ? -d : d;
        dep += (d <= alpha);
      }
    }
    if (NGTDM) {
      u64 *row = lds64 + (c - 1) * W;
      atomicAdd(row, 1ull);
      if (cnt) {
        int d = cnt * c - sum;
        d = d < 0 ? -d : d;
        if (d) atomicAdd(row + cnt, (u64)d);
      }
    } else {
      atomicAdd(&h32[(c - 1) * W + dep], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
    if (NGTDM) {
      u64 v = lds64[i];
      if (v) atomicAdd(ngtdm_acc + i, v);
    } else {
      u32 v = h32[i];
      if (v) atomicAdd(gldm_acc + i, v);
    }
  }
}

// ---- packed-byte variant: 4 x-adjacent voxels per lane ---------------------------------------------------------
// For the standard distance-1 neighbourhoods (26 in 3-D, 8 in-plane with force2D, ...) the neighbours of voxel x in
// one of the 9 (dz, dy) rows are the 3 bytes x-1, x, x+1 of that row.  A lane loads 6 bytes per row (one aligned
// dword + 2 edge bytes) for its 4 voxels and works on 3-byte windows with packed-byte arithmetic:
//   NGTDM  sum   += v_sad_u8(window, 0)               (sum of the 3 neighbour levels in one op)
//          count += popcount(nonzero-byte mask)        (SWAR: ((w & 0x7f7f7f) + 0x7f7f7f | w) & 0x808080)
//   GLDM   (alpha = 0)  dependence += 3 - #nonzero bytes of (window ^ centre level replicated)
// = ~8 VALU per voxel-row instead of ~10 per NEIGHBOUR in neigh_kernel.
struct RowMasks {
  unsigned m[9];   // per (dz+1)*3 + (dy+1): 24-bit byte mask of the dx = -1, 0, +1 neighbours that belong to the angle set
};

__device__ __forceinline__ unsigned nonzero_bytes3(unsigned w) {  // bit 7 of every non-zero byte (exact, no carries)
  return (((w & 0x7f7f7fu) + 0x7f7f7fu) | w) & 0x808080u;
}
__device__ __forceinline__ unsigned nonzero_bytes4(unsigned w) {  // bit 0 of every non-zero byte of the whole word
  return ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7) & 0x01010101u;
}

__device__ __forceinline__ unsigned neigh_from_lower_lane(unsigned v) {  // lane i <- lane i-1, lane 0 <- 0 (one DPP move)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned neigh_from_upper_lane(unsigned v) {  // lane i <- lane i+1, lane 63 <- 0
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

// MODE: 0 = GLDM (alpha = 0), 1 = NGTDM, 2 = both from the same neighbourhood reads (the case pipeline asks for both)
// LDS tables are u32 ([Ng][Na+1] NGTDM sums, then [Ng][Na+1] GLDM counts): a workgroup sees 2^18 voxels or fewer
// (2^19 at most: neigh_blocks) and a voxel adds at most 26 * 255, so the per-workgroup sums stay below 2^32; the flush widens to the
// u64 global accumulators.  With both tables (MODE 2) the voxel count of a level -- NGTDM slot 0 -- is the sum of its GLDM
// row and is only formed in the flush: one LDS atomic per voxel less (the one with the fewest distinct addresses).
// FULL26: the angle set is the whole 3 x 3 x 3 neighbourhood (the default of the classes in 3-D): the row masks are known
// at compile time -- no branch between the rows of an iteration (the scalar branches of the general form keep the
// compiler from overlapping a row's loads with the arithmetic of the row before).
template <int MODE, bool FULL26>
__global__ void __launch_bounds__(1024) neigh4_kernel(RowMasks R, const uint8_t *__restrict__ L, int Nz, int Ny,
                                                     int Nx, int zlo, int zhi, int Ng, int Na,
                                                     u32 *__restrict__ gldm_acc, u64 *__restrict__ ngtdm_acc,
                                                     const int *__restrict__ flags) {
  extern __shared__ u32 lds32[];
  if (flags[0]) return;
  constexpr bool NGTDM = MODE != 0, GLDM = MODE != 1;
  const int W = Na + 1;
  const int nbins = Ng * W;
  u32 *hn = lds32;                                   // NGTDM sums (slot 0: voxels of the level, MODE 1 only)
  u32 *h32 = MODE == 2 ? lds32 + nbins : lds32;      // GLDM counts
  for (int i = threadIdx.x; i < (MODE == 2 ? 2 : 1) * nbins; i += blockDim.x) lds32[i] = 0;
  __syncthreads();
  const int qpr = Nx >> 2;                      // quads per row
  const long long nquads = (long long)zhi * Ny * qpr;   // centres: planes zlo .. zhi-1
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  // byte offset of the (dz, dy) row relative to the centre row (rows are Nx bytes apart, planes Ny rows: wave-uniform)
  int rowoff[9];
#pragma unroll
  for (int r = 0; r < 9; r++) rowoff[r] = ((r / 3 - 1) * Ny + (r % 3 - 1)) * Nx;
  // Consecutive lanes hold consecutive quads: the edge bytes x0-1 / x0+4 of a row ARE the last / first byte of the
  // neighbour lanes' dword of that row (a quad at x0 = 0 / x0 + 4 = Nx has no such neighbour: 0), so only lane 0 / lane 63
  // load theirs -- 9 dword loads per quad instead of 9 dwords + 18 bytes (the texture-address unit was the bound:
  // see profiles/r03_probes.md).  Every lane of a wave runs the same trips (the lane moves need them all).
  const long long q0 = (long long)zlo * Ny * qpr + (long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63);
  for (long long qw = q0; qw < nquads; qw += stride) {
    const long long q = qw + lane;
    const bool live = q < nquads;
    // (32-bit arithmetic: a volume holds fewer than 2^31 voxels, so quad numbers, row numbers and byte offsets fit; the
    // 64-bit forms -- four quarter-rate multiplies per neighbour row -- were a quarter of an iteration's issue time)
    const unsigned q32 = live ? (unsigned)q : 0u;
    const unsigned row32 = q32 / (unsigned)qpr;
    const int x0 = (int)(q32 - row32 * (unsigned)qpr) << 2;
    const int z = (int)(row32 / (unsigned)Ny), y = (int)(row32 - (unsigned)z * (unsigned)Ny);
    const unsigned base = q32 << 2;             // byte offset of the quad (rows are Nx = 4 qpr bytes)
    const unsigned centre = live ? *reinterpret_cast<const unsigned *>(L + base) : 0u;
    if (__ballot(centre != 0) == 0) continue;   // no ROI voxel in these 256 columns (wave-uniform)
    int sum[4] = {0, 0, 0, 0};
    unsigned cnt4 = 0, neq4 = 0;   // per voxel of the quad, one byte each: valid neighbours / neighbours of another level
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const unsigned wm = FULL26 ? (r == 4 ? 0xff00ffu : 0xffffffu) : R.m[r];
      if (!FULL26 && !wm) continue;
      const int zz = z + r / 3 - 1, yy = y + r % 3 - 1;
      const bool in = live && (unsigned)zz < (unsigned)Nz && (unsigned)yy < (unsigned)Ny;   // row outside the volume: zeros
      const unsigned off = in ? base + (unsigned)rowoff[r] : 0u;
      const uint8_t *rp = L + off;
      const unsigned mid = in ? *reinterpret_cast<const unsigned *>(rp) : 0u;
      unsigned left = neigh_from_lower_lane(mid >> 24), right = neigh_from_upper_lane(mid & 0xffu);
      if (lane == 0) left = (in && x0 > 0) ? rp[-1] : 0u;
      if (lane == 63) right = (in && x0 + 4 < Nx) ? rp[4] : 0u;
      if (x0 == 0) left = 0u;
      if (x0 + 4 >= Nx) right = 0u;
      const unsigned lo = left | (mid << 8);          // bytes x0-1 .. x0+2
      const unsigned hi = (mid >> 24) | (right << 8); // bytes x0+3, x0+4
      // Counts for the four voxels of the quad at once: the row's neighbours dx = -1, 0, +1 of voxel x0+k are byte k of the
      // shifted words wL, wC (= mid), wR; "byte != 0" as bit 0 of every byte (nonzero_bytes4), summed in packed bytes
      // (at most 26 per byte).  The three-byte window per voxel is only kept for the level sums (v_sad_u8).
      const bool mL = FULL26 || (wm & 0xffu) != 0, mC = FULL26 ? r != 4 : (wm & 0xff00u) != 0,
                 mR = FULL26 || (wm & 0xff0000u) != 0;   // (wave-uniform; constants with FULL26)
      if (NGTDM) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          unsigned w = __builtin_amdgcn_alignbyte(hi, lo, k);   // neighbours x-1, x, x+1 of voxel x0+k (+ one byte beyond)
          if (FULL26 && r != 4) w &= 0xffffffu;
          else w &= wm;
          sum[k] = (int)__builtin_amdgcn_sad_u8(w, 0u, (unsigned)sum[k]);
        }
        const unsigned nzm = nonzero_bytes4(mid);
        if (mL) cnt4 += min(left, 1u) | (nzm << 8);
        if (mC) cnt4 += nzm;
        if (mR) cnt4 += (nzm >> 8) | (min(right, 1u) << 24);
      }
      if (GLDM) {
        // (a neighbour outside the volume or the ROI is 0, hence never equal to a non-zero centre)
        if (mL) neq4 += nonzero_bytes4(lo ^ centre);
        if (mC) neq4 += nonzero_bytes4(mid ^ centre);
        if (mR) neq4 += nonzero_bytes4(__builtin_amdgcn_alignbyte(hi, lo, 2) ^ centre);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = (int)((centre >> (8 * k)) & 0xffu);
      if (!c) continue;
      if (NGTDM) {
        u32 *rowp = hn + __mul24(c - 1, W);
        if (MODE == 1) atomicAdd(rowp, 1u);
        const int cn = (int)((cnt4 >> (8 * k)) & 0xffu);
        if (cn) {
          int d = __mul24(cn, c) - sum[k];
          d = d < 0 ? -d : d;
          if (d) atomicAdd(rowp + cn, (u32)d);
        }
      }
      if (GLDM) atomicAdd(&h32[__mul24(c - 1, W) + Na - (int)((neq4 >> (8 * k)) & 0xffu)], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
    if (NGTDM) {
      u64 v = hn[i];
      if (MODE == 2 && i % W == 0) {      // slot 0 of level i / W: its voxels = the sum of its GLDM row
        v = 0;
        for (int d = 0; d < W; d++) v += h32[i + d];
      }
      if (v) atomicAdd(ngtdm_acc + i, v);
    }
    if (GLDM) {
      const u32 v = h32[i];
      if (v) atomicAdd(gldm_acc + i, v);
    }
  }
}

// Launch geometry of neigh4_kernel: workgroups of 512 threads, ~8 quads (32 voxels) per thread.  Every workgroup flushes its
// [Ng][Na+1] table with global atomics on the same addresses (they serialise in L2), so a grid sized "one workgroup per
// 1024 voxels" spent half of a 232^3 launch there (152 -> 66 us with 768 workgroups); at 512^3 the same 768 would leave
// the main loop 30 % slower than 8192 do (686 -> 512 us): the sweet spot follows the volume (profiles/r03_probes.md, 10).
inline unsigned neigh_threads() {
  static const int bt = getenv("PRAD_NEIGH_THREADS") ? atoi(getenv("PRAD_NEIGH_THREADS")) : 512;
  return (unsigned)bt;
}
inline unsigned neigh_blocks(long long nquads, unsigned bt) {
  static const int fixed = getenv("PRAD_NEIGH_BLOCKS") ? atoi(getenv("PRAD_NEIGH_BLOCKS")) : 0;
  const long long need = (nquads + bt - 1) / bt;
  // the kernel's u32 LDS sums hold 26 * 255 per voxel: never more than 2^19 voxels (2^17 quads) per workgroup
  const long long floor32 = (nquads + (1LL << 17) - 1) >> 17;
  if (fixed > 0) return (unsigned)std::max<long long>(1, std::min<long long>(need, std::max<long long>(fixed, floor32)));
  return (unsigned)std::max<long long>(1, std::min<long long>(need, std::max<long long>(std::max<long long>(256, floor32), std::min<long long>(8192, nquads / (8LL * bt)))));
}

// the angle set as 9 row masks; false if an offset is outside {-1,0,1}^3 \ {0} or repeated
inline bool row_masks_from(const NeighSet &A, RowMasks *R) {
  for (int r = 0; r < 9; r++) R->m[r] = 0;
  for (int a = 0; a < A.na; a++) {
    const int dz = A.o[a][0], dy = A.o[a][1], dx = A.o[a][2];
    if (dz < -1 || dz > 1 || dy < -1 || dy > 1 || dx < -1 || dx > 1 || (!dz && !dy && !dx)) return false;
    const unsigned bit = 0xffu << (8 * (dx + 1));
    unsigned &m = R->m[(dz + 1) * 3 + (dy + 1)];
    if (m & bit) return false;
    m |= bit;
  }
  return true;
}

inline bool row_masks_full26(const RowMasks &R) {
  for (int r = 0; r < 9; r++)
    if (R.m[r] != (r == 4 ? 0xff00ffu : 0xffffffu)) return false;
  return true;
}

__global__ void finalize_gldm_kernel(const u32 *__restrict__ acc, int Ng, int Na, double *__restrict__ out) {
  const int width = 2 * Na + 1;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Ng * width) return;
  const int g = (int)(idx / width), k = (int)(idx % width);
  out[idx] = k <= Na ? (double)acc[(size_t)g * (Na + 1) + k] : 0.0;
}

struct NeighPlan {
  bool ok = false;
  int Nz = 1, Ny = 1, Nx = 1;
  NeighSet set;
};

inline NeighPlan plan_neigh(const Geo &g, const VoxMode &vm, const int *angles_h, int Na, int Ng, size_t bin_bytes) {
  NeighPlan p;
  if (vm.voxels || g.nd > 3 || Ng < 1 || Ng > 255 || Na > PRAD_MAX_NEIGH) return p;
  if ((size_t)Ng * (Na + 1) * bin_bytes > 64 * 1024) return p;
  int dims[3] = {1, 1, 1};
  for (int d = 0; d < g.nd; d++) dims[3 - g.nd + d] = g.size[d];
  p.Nz = dims[0]; p.Ny = dims[1]; p.Nx = dims[2];
  p.set.na = Na;
  for (int a = 0; a < Na; a++) {
    int o[3] = {0, 0, 0};
    for (int d = 0; d < g.nd; d++) o[3 - g.nd + d] = angles_h[a * g.nd + d];
    for (int d = 0; d < 3; d++) {
      if (o[d] < -127 || o[d] > 127) return p;
      p.set.o[a][d] = (signed char)o[d];
    }
    p.set.o[a][3] = 0;
  }
  p.ok = true;
  return p;
}

// A packed copy of the WHOLE volume that somebody made already (prad_image_enqueue_dev, round 6: GLDM / NGTDM and the GLSZM of
// one derived image each packed the same int32 levels + mask into the same bytes, 35 us each at 256^3): a neigh_pack() call
// for the same (image, mask, voxel count, Ng) hands it out instead of packing again and ORs the pack's verdict word into the
// caller's.  Thread local, set and cleared by the one caller that knows the consumers run behind the pack (same stream, or a
// stream that waits for it).
struct SharedPack {
  const int32_t *image = nullptr;
  const uint8_t *mask = nullptr;
  long long n = 0;
  int Ng = 0;
  uint8_t *levels = nullptr;
  const int *flags = nullptr;      // [0] != 0: a masked level outside [1, Ng]
};
inline SharedPack &shared_pack() {
  static thread_local SharedPack p;
  return p;
}
__global__ void or_flag_kernel(int *__restrict__ dst, const int *__restrict__ src) {
  if (src[0]) dst[0] = 1;
}
inline bool take_shared_pack(hipStream_t s, const Geo &g, const int32_t *image, const uint8_t *mask, int Ng, int *flags_d,
                             uint8_t **levels) {
  const SharedPack &sp = shared_pack();
  if (!sp.levels || sp.image != image || sp.mask != mask || sp.n != g.n || sp.Ng != Ng) return false;
  *levels = sp.levels;
  hipLaunchKernelGGL(or_flag_kernel, dim3(1), dim3(1), 0, s, flags_d, sp.flags);
  return true;
}

// packs planes [plo, phi) of the 3-D-embedded volume (the rest of `levels` is not touched / not read)
inline int neigh_pack(Context *c, hipStream_t s, const Geo &g, const NeighPlan &p, int plo, int phi,
                      const int32_t *image, const uint8_t *mask, int Ng, int *flags_d, uint8_t **levels) {
  if (take_shared_pack(s, g, image, mask, Ng, flags_d, levels)) return check_launch("or_flag_kernel");
  PRAD_TRY(c->get<uint8_t>("levels", (size_t)g.n + 64, levels));
  Timed t(*c, "pack", s);
  const long long plane = (long long)p.Ny * p.Nx, off = plo * plane, n = (phi - plo) * plane;
  const int vec_ok = ((((uintptr_t)(image + off)) | ((uintptr_t)(mask + off)) | ((uintptr_t)(*levels + off))) & 15) == 0;
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((n / 16 + 255) / 256, 4096));
  hipLaunchKernelGGL(pack_levels_kernel, dim3(gx), dim3(256), 0, s, image + off, mask + off, n, p.Nx, p.Nx, 0, Ng,
                     *levels + off, flags_d, vec_ok, 0);
  return check_launch("pack_levels_kernel");
}

// whole volume (voxel-map callers)
inline int neigh_pack(Context *c, hipStream_t s, const Geo &g, const int32_t *image, const uint8_t *mask, int Ng,
                      int *flags_d, uint8_t **levels) {
  NeighPlan p;
  p.Nz = 1; p.Ny = 1; p.Nx = g.size[g.nd - 1];
  const long long rows = g.n / p.Nx;
  if (rows > 0x7fffffffll) return fail(PRAD_E_ARG, "volume too large");
  p.Ny = (int)rows;
  return neigh_pack(c, s, g, p, 0, 1, image, mask, Ng, flags_d, levels);
}

// Integer accumulators [Ng][Na+1] (u32 for GLDM, u64 for NGTDM, layout in the header comment) of the centre
// voxels in planes [zlo, zhi) of the 3-D-embedded volume, neighbours taken from the whole volume.  The
// accumulators of disjoint plane ranges add up to those of the whole volume -- the z-slab split of one
// large segment over several GPUs sums them before the finalize step.
template <bool NGTDM>
inline int neigh_accumulate(Context *c, hipStream_t s, const Geo &g, const VoxMode &vm, const int32_t *image,
                            const uint8_t *mask, const int *angles_h, int Na, int Ng, int alpha, int zlo, int zhi,
                            int *flags_d, void **acc_out, bool *done) {
  *done = false;
  NeighPlan p = plan_neigh(g, vm, angles_h, Na, Ng, NGTDM ? sizeof(u64) : sizeof(u32));
  if (!p.ok) return PRAD_OK;
  if (zlo < 0) zlo = 0;
  if (zhi > p.Nz) zhi = p.Nz;
  int reach = 0;
  for (int a = 0; a < Na; a++) reach = std::max(reach, std::abs((int)p.set.o[a][0]));
  uint8_t *levels = nullptr;
  const int plo = std::max(0, zlo - reach), phi = std::min(p.Nz, zhi + reach);
  if (zhi > zlo) PRAD_TRY(neigh_pack(c, s, g, p, plo, phi, image, mask, Ng, flags_d, &levels));
  const size_t nacc = (size_t)Ng * (Na + 1);
  u32 *acc32 = nullptr;
  u64 *acc64 = nullptr;
  if (NGTDM) {
    PRAD_TRY(c->get<u64>("ngtdm_acc", nacc, &acc64));
    PRAD_HIP(hipMemsetAsync(acc64, 0, sizeof(u64) * nacc, s));
    *acc_out = acc64;
  } else {
    PRAD_TRY(c->get<u32>("gldm_acc", nacc, &acc32));
    PRAD_HIP(hipMemsetAsync(acc32, 0, sizeof(u32) * nacc, s));
    *acc_out = acc32;
  }
  if (zhi > zlo) {
    Timed t(*c, "neigh", s);
    const size_t lds = (NGTDM ? sizeof(u64) : sizeof(u32)) * nacc;
    const long long ncent = (long long)(zhi - zlo) * p.Ny * p.Nx;
    RowMasks R;
    // the packed-byte path reads planes z-1 .. z+1 only where the row masks say so, all inside [plo, phi)
    if ((NGTDM || alpha == 0) && (p.Nx & 3) == 0 && row_masks_from(p.set, &R)) {
      // (grid: neigh_blocks above)
      const unsigned bt = neigh_threads();
      const unsigned gx = neigh_blocks(ncent >> 2, bt);
      if (row_masks_full26(R))
        hipLaunchKernelGGL((neigh4_kernel<NGTDM, true>), dim3(gx), dim3(bt), lds, s, R, levels, p.Nz, p.Ny, p.Nx, zlo, zhi,
                           Ng, Na, acc32, acc64, flags_d);
      else
        hipLaunchKernelGGL((neigh4_kernel<NGTDM, false>), dim3(gx), dim3(bt), lds, s, R, levels, p.Nz, p.Ny, p.Nx, zlo, zhi,
                           Ng, Na, acc32, acc64, flags_d);
      PRAD_TRY(check_launch("neigh4_kernel"));
    } else {
      const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((ncent + 255) / 256, 4096));
      hipLaunchKernelGGL((neigh_kernel<NGTDM>), dim3(gx), dim3(256), lds, s, p.set, levels, p.Nz, p.Ny, p.Nx, zlo,
                         zhi, Ng, alpha, acc32, acc64, flags_d);
      PRAD_TRY(check_launch("neigh_kernel"));
    }
  }
  *done = true;
  return PRAD_OK;
}

inline int neigh_finalize_gldm(Context *c, hipStream_t s, const u32 *acc, int Ng, int Na, double *out) {
  Timed t(*c, "finalize", s);
  const long long total = (long long)Ng * (2 * Na + 1);
  hipLaunchKernelGGL(finalize_gldm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, acc, Ng, Na, out);
  return check_launch("finalize_gldm_kernel");
}

inline int neigh_finalize_ngtdm(Context *c, hipStream_t s, const u64 *acc, int Ng, int Na, double *out) {
  Timed t(*c, "finalize", s);
  hipLaunchKernelGGL(ngtdm_finalize_kernel, dim3((unsigned)((Ng + 63) / 64)), dim3(64), 0, s, acc, Ng, Na, out);
  return check_launch("ngtdm_finalize_kernel");
}

// GLDM (alpha = 0) and NGTDM of the whole volume from ONE pass over the packed levels; *done = false: not a case for the
// packed-byte kernel, the caller takes the two separate calls
inline int neigh_try_both(Context *c, hipStream_t s, const Geo &g, const VoxMode &vm, const int32_t *image,
                          const uint8_t *mask, const int *angles_h, int Na, int Ng, int alpha, double *gldm_out,
                          double *ngtdm_out, int *flags_d, bool *done) {
  *done = false;
  NeighPlan p = plan_neigh(g, vm, angles_h, Na, Ng, sizeof(u64) + sizeof(u32));
  RowMasks R;
  if (!p.ok || alpha != 0 || (p.Nx & 3) != 0 || !row_masks_from(p.set, &R)) return PRAD_OK;
  uint8_t *levels = nullptr;
  PRAD_TRY(neigh_pack(c, s, g, p, 0, p.Nz, image, mask, Ng, flags_d, &levels));
  const size_t nacc = (size_t)Ng * (Na + 1);
  u32 *acc32 = nullptr;
  u64 *acc64 = nullptr;
  PRAD_TRY(c->get<u64>("ngtdm_acc", nacc, &acc64));
  PRAD_TRY(c->get<u32>("gldm_acc", nacc, &acc32));
  PRAD_TRY(ZeroBatch().add(acc64, sizeof(u64) * nacc).add(acc32, sizeof(u32) * nacc).launch(s));
  {
    Timed t(*c, "neigh", s);
    const long long ncent = (long long)p.Nz * p.Ny * p.Nx;
    const unsigned bt = neigh_threads();
    const unsigned gx = neigh_blocks(ncent >> 2, bt);
    if (row_masks_full26(R))
      hipLaunchKernelGGL((neigh4_kernel<2, true>), dim3(gx), dim3(bt), (sizeof(u64) + sizeof(u32)) * nacc, s, R, levels, p.Nz, p.Ny,
                         p.Nx, 0, p.Nz, Ng, Na, acc32, acc64, flags_d);
    else
      hipLaunchKernelGGL((neigh4_kernel<2, false>), dim3(gx), dim3(bt), (sizeof(u64) + sizeof(u32)) * nacc, s, R, levels, p.Nz, p.Ny,
                         p.Nx, 0, p.Nz, Ng, Na, acc32, acc64, flags_d);
    PRAD_TRY(check_launch("neigh4_kernel"));
  }
  PRAD_TRY(neigh_finalize_gldm(c, s, acc32, Ng, Na, gldm_out));
  PRAD_TRY(neigh_finalize_ngtdm(c, s, acc64, Ng, Na, ngtdm_out));
  *done = true;
  return PRAD_OK;
}

inline int neigh_try_gldm(Context *c, hipStream_t s, const Geo &g, const VoxMode &vm, const int32_t *image,
                          const uint8_t *mask, const int *angles_h, int Na, int Ng, int alpha, double *out,
                          int *flags_d, bool *done) {
  void *acc = nullptr;
  PRAD_TRY(neigh_accumulate<false>(c, s, g, vm, image, mask, angles_h, Na, Ng, alpha, 0, 1 << 30, flags_d, &acc, done));
  if (!*done) return PRAD_OK;
  return neigh_finalize_gldm(c, s, (const u32 *)acc, Ng, Na, out);
}

inline int neigh_try_ngtdm(Context *c, hipStream_t s, const Geo &g, const VoxMode &vm, const int32_t *image,
                           const uint8_t *mask, const int *angles_h, int Na, int Ng, double *out, int *flags_d,
                           bool *done) {
  void *acc = nullptr;
  PRAD_TRY(neigh_accumulate<true>(c, s, g, vm, image, mask, angles_h, Na, Ng, 0, 0, 1 << 30, flags_d, &acc, done));
  if (!*done) return PRAD_OK;
  return neigh_finalize_ngtdm(c, s, (const u64 *)acc, Ng, Na, out);
}

// widening copy of the GLDM accumulators for the exchange step (RCCL sums int64)
__global__ void widen_u32_kernel(const u32 *__restrict__ in, long long n, long long *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (long long)in[i];
}
__global__ void narrow_i64_kernel(const long long *__restrict__ in, long long n, u32 *__restrict__ out, int *flags) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (in[i] < 0 || in[i] > 0xffffffffll) flags[1] = 1;
  out[i] = (u32)in[i];
}

}  // namespace prad
