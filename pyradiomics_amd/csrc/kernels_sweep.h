// kernels_sweep.h -- the segment-mode hot path for GLCM + GLRLM on gfx950.
//
// Data layout in HBM
//   image  int32 [Nz][Ny][Nx]  + mask uint8 [Nz][Ny][Nx]     (boundary dtypes, 5 B/voxel, read ONCE)
//   levels uint8 [Nz*Ny][pitch]  level | 0 = outside the ROI; pitch = Nx + padw and the first padw bytes of
//                                every row are repeated behind it (periodic pad), written once by pack_levels,
//                                then L2 / Infinity-Cache resident for the 13 sweeps
//   acc    uint32 GLCM [Na][Ng][Ng], GLRLM [Na][Ng][Nr]        (angle-major while accumulating)
//   out    float64 GLCM [Ng][Ng][Na], GLRLM [Ng][Nr][Na]       (reference layout, written by finalize)
//
// Why sweeps.  For a unit angle d the ordered neighbour pairs (p, p+d) of the GLCM are exactly the
// consecutive voxels of the lines the GLRLM walks (cmatrices.c:61-85 vs :448-510), so one walk along every
// line of an angle yields both matrices for that angle: a lane keeps (previous level, current run length)
// in registers and consumes one byte per step.
//
//   * lines kernel (angles marching along z or y).  Lines are walked WRAPPED: at march step t the lines of a
//     wave sit at row (u0 + t*du) mod NU and columns (x0 + t*dx) mod NX.  Every lane is therefore busy for all
//     NM steps (no skew triangles, no predicates), a step is one contiguous 64*LPL-byte read per wave from a
//     wave-uniform (SGPR) base -- the periodic row pad makes the read contiguous across the wrap -- and a wrap
//     is simply a line break (close the run, start a new line).  Row wraps are wave-uniform; at most one of a
//     wave's lines wraps in x per step, and groups of 8 steps without any wrap run a check-free fast path.
//     Each lane owns LPL (1/2/4) adjacent lines = one (possibly unaligned) byte/short/dword load per step.
//   * rows kernel (the angle along x): a wave owns 64 consecutive rows; 64x64-voxel tiles are read coalesced
//     (16 B per lane), transposed through LDS, and every lane walks its own row with the same walker.
//
// The per-step state machine is branch-free: non-events are steered to a per-lane dummy LDS word.
// Histograms are privatised per workgroup in LDS (ds_add_u32, no return); workgroups are persistent and
// merge with one global atomic per non-zero bin.  Integer histogramming: MFMA has no role here.
#pragma once
#include "prad_runtime.h"

namespace prad {

typedef unsigned int u32;
typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------------
// pack: (int32 level, uint8 mask) -> padded uint8 rows.  flags[0] is set when a masked voxel has a level
// outside [1, Ng] (=> the exact generic path must run instead).  padw == 0: plain linear layout.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_levels_kernel(const int *__restrict__ image,
                                                          const uint8_t *__restrict__ mask, long long n, int NX,
                                                          int pitch, int padw, int Ng,
                                                          uint8_t *__restrict__ levels, int *__restrict__ flags,
                                                          int vec_ok, int shift, uint8_t *__restrict__ rowzero = nullptr) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  const bool linear = (padw == 0 && pitch == NX);
  int bad = 0;
  long long done = 0;
  if (vec_ok) {  // 16 voxels per lane; with a pad, NX / pitch / padw are multiples of 16
    const long long n16 = n >> 4;
    const int upr = NX >> 4;  // 16-voxel units per row (pad layout only)
    const int4 *im4 = reinterpret_cast<const int4 *>(image);
    const uint4 *mk4 = reinterpret_cast<const uint4 *>(mask);
    for (long long t = tid; t < n16; t += nthreads) {
      const uint4 m = mk4[t];
      const int4 q0 = im4[4 * t], q1 = im4[4 * t + 1], q2 = im4[4 * t + 2], q3 = im4[4 * t + 3];
      const int lv[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                          q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
      const u32 mw[4] = {m.x, m.y, m.z, m.w};
      u32 ow[4];
#pragma unroll
      for (int w = 0; w < 4; w++) {
        u32 o = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const bool in = (mw[w] >> (8 * b)) & 0xffu;
          const int l = lv[w * 4 + b];
          bad |= in && (l < 1 || l > Ng);
          o |= (in ? (((u32)l << shift) & 0xffu) : 0u) << (8 * b);
        }
        ow[w] = o;
      }
      const uint4 o4 = make_uint4(ow[0], ow[1], ow[2], ow[3]);
      if (rowzero) {   // rows that hold a voxel outside the ROI (kernels_sweepfw.h); zeroed by the caller beforehand
        u32 zb = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) zb |= (ow[w] - 0x01010101u) & ~ow[w] & 0x80808080u;
        if (zb) {
          flags[3] = 1;
          const long long e0 = t << 4;
          rowzero[e0 / NX] = 1;
          rowzero[(e0 + 15) / NX] = 1;      // (a 16-voxel piece spans at most two rows when NX >= 16; marking both is safe)
        }
      }
      if (linear) {
        reinterpret_cast<uint4 *>(levels)[t] = o4;
      } else {
        const long long row = t / upr;
        const int x = (int)(t - row * upr) << 4;
        uint8_t *dst = levels + row * pitch + x;
        *reinterpret_cast<uint4 *>(dst) = o4;
        if (x < padw) *reinterpret_cast<uint4 *>(dst + NX) = o4;
      }
    }
    done = n16 << 4;  // (n is a multiple of 16 whenever a pad layout takes this path)
  }
  for (long long i = done + tid; i < n; i += nthreads) {
    const bool in = mask[i] != 0;
    const int l = image[i];
    bad |= in && (l < 1 || l > Ng);
    const uint8_t v = in ? (uint8_t)(l << shift) : (uint8_t)0;
    if (rowzero && !v) {
      rowzero[i / NX] = 1;
      flags[3] = 1;
    }
    if (linear) {
      levels[i] = v;
    } else {
      const long long row = i / NX;
      const int x = (int)(i - row * NX);
      levels[row * pitch + x] = v;
      if (x < padw) levels[row * pitch + NX + x] = v;
    }
  }
  if (bad) flags[0] = 1;
}

// One sweepable angle in (march, row, lane) coordinates; strides are in bytes of the padded level volume.
struct SweepDesc {
  int slot;          // index of the angle in the caller's list (output column)
  int NM, NU, NX;    // extents: march dim, row dim, lane dim (lane dim is always the contiguous x axis)
  int du, dx;        // motion per march step in the row / lane dims (-1, 0, +1)
  long long sM, sU;  // strides of march and row dims
  int LXc;           // chunks (of 64*LPL lines) per row
  long long chunks;  // NU * LXc
};
#define PRAD_MAX_SWEEP 16
struct SweepSet {
  int count;
  SweepDesc d[PRAD_MAX_SWEEP];
};

// ---- LDS histogram layouts (u32 words) -----------------------------------------------------------------
// separate tables (GLCM only / GLRLM only / both when the fused table does not fit):
//   [0, Ng*Ng)                 GLCM  [prev-1][cur-1]                  (only when DO_GLCM)
//   [.., +RS*Ng)               GLRLM [len-1][level-1], len <= RS      (only when DO_GLRLM)
//   [.., +64)                  per-lane dummy words (targets of masked-out increments)
// fused table (GLCM and GLRLM together, FUSED):
//   [0, (Ng+1)*(RS+1)*(Ng+1))  H [prev][min(len,RS+1)-1][cur]         cur = 0: the run ended at an unmasked
//   [.., +Ng*RL)               G [prev-1][len-1-RS], RS < len <= RS+RL voxel / line end;  row prev = 0 collects
//   [.., +64)                  dummies                                the "events" that follow unmasked voxels
//                                                                     and is ignored (saves a compare per step)
//   One event per RUN END carries everything both matrices need:
//     GLRLM[prev][len]      = sum_cur H[prev][len][cur]          (len <= RS; a longer run puts its pair into the
//                                                                 len slot RS and its length into G, or, beyond
//                                                                 RS+RL, into a wave-aggregated L2 atomic)
//     GLCM[prev][cur!=prev] = sum_len H[prev][len][cur]          (a pair of different levels IS a run boundary)
//     GLCM[g][g]            = sum_len (len-1) * GLRLM[g][len]    (pairs inside runs; evaluated in finalize)
//   so the walk issues ONE ds_add per step instead of two, and none of them hits the hot diagonal bins.
#define PRAD_LONG_BINS 64
struct HistLayout {
  int Ng, RS;
  bool fused;
  int glrlm0;  // first GLRLM word (separate tables)
  int g0, RL;  // fused: first word / bins per level of the long-run table G
  int dummy0;  // first dummy word
  int words;   // total
};
__host__ __device__ inline HistLayout hist_layout(bool glcm, bool glrlm, bool fused, int Ng, int RS) {
  HistLayout h;
  h.Ng = Ng;
  h.RS = RS;
  h.fused = fused;
  h.g0 = 0;
  h.RL = 0;
  if (fused) {
    h.glrlm0 = 0;
    h.g0 = (Ng + 1) * (RS + 1) * (Ng + 1);
    h.RL = PRAD_LONG_BINS;
    h.dummy0 = h.g0 + Ng * h.RL;
  } else {
    h.glrlm0 = glcm ? Ng * Ng : 0;
    h.dummy0 = h.glrlm0 + (glrlm ? RS * Ng : 0);
  }
  h.words = h.dummy0 + 64;
  return h;
}

// cond ? a : b as ONE v_cndmask: both operands are pinned in VGPRs first, otherwise hipcc sinks their computation
// into a divergent if/else (s_and_saveexec / s_xor / s_andn2_saveexec / s_or: 4 scalar ops + a second compare per
// voxel-step, which made the scalar unit the bottleneck of the sweeps)
__device__ __forceinline__ int select_i32(bool cond, int a, int b) {
  asm volatile("" : "+v"(a), "+v"(b));
  return cond ? a : b;
}

typedef __attribute__((address_space(3))) u32 lds_u32;
__device__ __forceinline__ void lds_bump(int lds_addr) {  // bare ds_add_u32 on a 32-bit LDS byte address
#ifdef PRAD_DBG_NOBUMP  // ablation build: keep the address computation alive, drop the LDS atomic
  asm volatile("" ::"v"(lds_addr));
  return;
#endif
  __hip_atomic_fetch_add((lds_u32 *)(size_t)(unsigned)lds_addr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Per-line walk state + the branch-free step, separate-table flavour.  All LDS positions are absolute LDS
// BYTE addresses:  GLCM byte = prev*4Ng + cur*4 + cG,  GLRLM byte = len*4Ng + prev*4 + cR.
template <bool DO_GLCM, bool DO_GLRLM, bool LONG, bool FUSED>
struct Walker {
  u32 *rl_long;  // global GLRLM rows of this angle (long runs only)
  int Ng4, Nr;   // Ng4 = 4*Ng
  int cG, cR;    // see above
  int rl_limit;  // len*4Ng + cR beyond which the run is "long" (LONG only)
  int dummy;     // LDS address of this lane's dummy word
  // state
  int prev;   // level of the previous voxel on the line (0 = none / unmasked)
  int prowG;  // prev*4Ng + cG
  int rlN;    // len*4Ng + cR of the stretch of equal values ending at prev
  int nmask;  // masked voxels seen on this line

  __device__ __forceinline__ void init(u32 *lds_, const HistLayout &h, int Nr_, u32 *rl_long_, int lane) {
    const int base = (int)(unsigned)(size_t)((lds_u32 *)lds_);
    rl_long = rl_long_;
    Ng4 = 4 * h.Ng;
    Nr = Nr_;
    cG = base - 4 * (h.Ng + 1);
    cR = base + 4 * (h.glrlm0 - (h.Ng + 1));
    rl_limit = h.RS * Ng4 + cR;
    dummy = base + 4 * (h.dummy0 + lane);
  }
  __device__ __forceinline__ void begin_line() {
    prev = 0;
    prowG = cG;
    rlN = Ng4 + cR;
    nmask = 0;
  }
  // true if a run of this line could be emitted as "long" within the next `steps` steps
  __device__ __forceinline__ bool risky(int steps) const { return LONG && DO_GLRLM && rlN + steps * Ng4 > rl_limit; }
  template <bool CHECK = true>
  __device__ __forceinline__ void step(int cur) {
    const bool pnz = prev != 0, cnz = cur != 0;
    if (DO_GLCM) {
      lds_bump((pnz && cnz) ? prowG + (cur << 2) : dummy);
      prowG = __mul24(cur, Ng4) + cG;
    }
    if (DO_GLRLM) {
      const bool chg = cur != prev;
      const bool emit = chg && pnz;
      if (LONG && CHECK) {
        const bool lng = rlN > rl_limit;
        lds_bump((emit && !lng) ? rlN + (prev << 2) : dummy);
        if (emit && lng) atomicAdd(&rl_long[(size_t)(prev - 1) * Nr + ((rlN - cR) / Ng4 - 1)], 1u);
      } else {
        lds_bump(emit ? rlN + (prev << 2) : dummy);
      }
      rlN = select_i32(chg, Ng4 + cR, rlN + Ng4);
      nmask += cnz;
    }
    prev = cur;
  }
  // closes the run that is open at the end of a line; returns "the line held >= 2 masked voxels"
  __device__ __forceinline__ bool end_line() {
    step<true>(0);
    return nmask > 1;
  }
  // a step that may first break the line (wrapped sweeps); returns end_line()'s verdict for the closed line
  template <bool CHECK = true>
  __device__ __forceinline__ bool step_brk(int cur, bool brk) {
    bool m = false;
    if (brk) {
      m = end_line();
      begin_line();
    }
    step<CHECK>(cur);
    return m;
  }
};

// Fused flavour: H byte = prev*P + min(len-1, RS)*Q + cur*4 with Q = 4(Ng+1), P = (RS+1)*Q.
// State is ONE running LDS address pl = prev*P + (len-1)*Q (the bin row of the open run, relative to the table) plus prev.
// The fused table only fits for Ng <= 44, so the packed volume holds level*4 (PRAD_FUSED_SHIFT) and `prev` / `cur`
// below are those pre-scaled bytes: the bin offset cur*4 is the byte itself and prev*P = byte * (P/4), which lets
// the compiler read the byte lanes of the packed word directly in v_add / v_mul_u32_u24 / v_cmp (SDWA) instead of
// extracting them first.
// Whether a line held >= 2 masked voxels is NOT tracked here (it would cost mask logic on every step); it is
// recovered after the sweeps by resolve_multi_kernel / multi_check_kernel.
#define PRAD_FUSED_SHIFT 2
template <bool LONG>
struct Walker<true, true, LONG, true> {
  u32 *rl_long;
  int Nr, P4, Q;
  // H sits at LDS address 0 (the kernels' only LDS object is the dynamic array; they check it and flag an internal
  // error otherwise), so table offsets ARE LDS addresses and no base has to be carried through the address math
  int lenmax;  // RS*Q: byte offset of the "long" slot
  int gB;      // offset of G minus the offset of its first bin (len-1 = RS, prev = 1)
  int RL4;     // bytes per level row of G
  unsigned Qinv;  // ceil(2^32 / Q): (len-1) = umulhi((len-1)*Q, Qinv)
  int RS, RL;
  int dummy;
  // state
  int prev;  // level*4 of the previous voxel (0 = none / outside the ROI)
  int pl;    // level*P + (len-1)*Q, unclamped

  __device__ __forceinline__ void init(u32 *lds_, const HistLayout &h, int Nr_, u32 *rl_long_, int lane) {
    rl_long = rl_long_;
    Nr = Nr_;
    Q = 4 * (h.Ng + 1);
    P4 = (h.RS + 1) * (h.Ng + 1);               // P / 4
    lenmax = h.RS * Q;
    RS = h.RS;
    RL = h.RL;
    RL4 = 4 * h.RL;
    gB = 4 * h.g0 - RL4 - 4 * h.RS;       // + level*RL4 + (len-1)*4 addresses G[level-1][len-1-RS]
    Qinv = (unsigned)((0x100000000ull + (unsigned)Q - 1) / (unsigned)Q);
    dummy = 4 * (h.dummy0 + lane);
  }
  __device__ __forceinline__ void begin_line() {
    prev = 0;
    pl = 0;
  }
  // the run of level `lv` (!= 0) that just ended was longer than RS: record its length.  Rare, divergent.
  __device__ __forceinline__ void long_event(int lv, int lb) {
    const int idx = (int)__umulhi((unsigned)lb, Qinv);            // len - 1
    if (idx < RS + RL) {
      lds_bump(gB + __mul24(lv, RL4) + (idx << 2));
      return;
    }
    // very long runs (flat regions): lanes of the wave that close the same (level, length) share one L2 atomic
    const unsigned key = ((unsigned)lv << 20) | (unsigned)idx;
    bool pending = true;
    while (pending) {
      const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)key);
      const bool same = key == first;
      const unsigned long long m = __ballot(same);       // evaluated by every still-pending lane
      if (same) {
        if ((int)(__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63))
          atomicAdd(&rl_long[(size_t)(lv - 1) * Nr + idx], (u32)__popcll(m));
        pending = false;
      }
    }
  }
  __device__ __forceinline__ int lenb() const { return pl - __mul24(prev, P4); }
  // stretches of unmasked voxels count too: their events land in row 0, but an unclamped slot must stay in range
  __device__ __forceinline__ bool risky(int steps) const { return LONG && lenb() + steps * Q > lenmax; }
  template <bool CHECK = true>
  __device__ __forceinline__ void step(int cur) {
    const bool chg = cur != prev;
    int bin = pl;
    if (LONG && CHECK) {
      const int lb = lenb();
      bin = pl - lb + min(lb, lenmax);
      if (chg && prev != 0 && lb >= lenmax) long_event(prev >> PRAD_FUSED_SHIFT, lb);
    }
    lds_bump(chg ? bin + cur : dummy);
    pl = select_i32(chg, __mul24(cur, P4), pl + Q);
    prev = cur;
  }
  __device__ __forceinline__ bool end_line() {
    step<true>(0);
    return false;
  }
  // branch-free line break: the event of the closed run records cur = 0 (no pair across the break)
  template <bool CHECK = true>
  __device__ __forceinline__ bool step_brk(int cur, bool brk) {
    const bool chg = (cur != prev) || brk;
    const int evt = brk ? 0 : cur;
    int bin = pl;
    if (LONG && CHECK) {
      const int lb = lenb();
      bin = pl - lb + min(lb, lenmax);
      if (chg && prev != 0 && lb >= lenmax) long_event(prev >> PRAD_FUSED_SHIFT, lb);
    }
    lds_bump(chg ? bin + evt : dummy);
    pl = select_i32(chg, __mul24(cur, P4), pl + Q);
    prev = cur;
    return false;
  }
};

// One march step of LPL independent lines with the work of the lines interleaved (all compares first, then all
// address selects / bumps, then all state updates): the hardware needs a wait state between a VALU compare and the
// v_cndmask that consumes it, and hipcc otherwise walks the lines one after the other and fills it with s_nop.
template <int LPL, typename W>
__device__ __forceinline__ void step_lines_plain(W (&w)[LPL], unsigned v) {
#pragma unroll
  for (int j = 0; j < LPL; j++) w[j].template step<false>((int)((v >> (8 * j)) & 0xffu));
}
// Fused flavour: `pw` is the packed word of the previous step (the prev bytes of the LPL lines).  Every operand that
// involves a level is a byte lane of v / pw, so compare, bin address and fresh row each cost one SDWA instruction:
// 6 VALU + 1 ds_add per voxel-step.  The walkers' own `prev` members are NOT maintained here; the caller packs them
// into pw before a group of plain steps and unpacks the last word afterwards (pack_prev / unpack_prev).
template <int LPL, bool LONG>
__device__ __forceinline__ void step_lines_plain(Walker<true, true, LONG, true> (&w)[LPL], unsigned v, unsigned pw) {
  int addr[LPL], fresh[LPL], grown[LPL];
  bool chg[LPL];
  // explicit bit-field extracts: the SDWA peephole folds a byte-aligned v_bfe into each VOP2 / VOPC user, whereas the
  // shift-and-mask form gets rewritten into (v ^ pw) & mask != 0, which costs two instructions
#pragma unroll
  for (int j = 0; j < LPL; j++) chg[j] = __builtin_amdgcn_ubfe(v, 8 * j, 8) != __builtin_amdgcn_ubfe(pw, 8 * j, 8);
#pragma unroll
  for (int j = 0; j < LPL; j++) {
    addr[j] = w[j].pl + (int)__builtin_amdgcn_ubfe(v, 8 * j, 8);
    fresh[j] = (int)__umul24(__builtin_amdgcn_ubfe(v, 8 * j, 8), (unsigned)w[j].P4);
    grown[j] = w[j].pl + w[j].Q;
  }
#pragma unroll
  for (int j = 0; j < LPL; j++) lds_bump(chg[j] ? addr[j] : w[j].dummy);
#pragma unroll
  for (int j = 0; j < LPL; j++) w[j].pl = select_i32(chg[j], fresh[j], grown[j]);
}
template <int LPL, typename W>
__device__ __forceinline__ unsigned pack_prev(const W (&w)[LPL]) {
  unsigned pw = 0;
#pragma unroll
  for (int j = 0; j < LPL; j++) pw |= (unsigned)w[j].prev << (8 * j);
  return pw;
}
template <int LPL, typename W>
__device__ __forceinline__ void unpack_prev(W (&w)[LPL], unsigned pw) {
#pragma unroll
  for (int j = 0; j < LPL; j++) w[j].prev = (int)((pw >> (8 * j)) & 0xffu);
}
template <int LPL, int U, typename W>
__device__ __forceinline__ void steps_plain_group(W (&w)[LPL], const unsigned (&v)[U]) {
#pragma unroll
  for (int k = 0; k < U; k++) step_lines_plain<LPL>(w, v[k]);
}
template <int LPL, int U, bool LONG>
__device__ __forceinline__ void steps_plain_group(Walker<true, true, LONG, true> (&w)[LPL], const unsigned (&v)[U]) {
  unsigned pw = pack_prev<LPL>(w);
#pragma unroll
  for (int k = 0; k < U; k++) {
    step_lines_plain<LPL, LONG>(w, v[k], pw);
    pw = v[k];
  }
  unpack_prev<LPL>(w, pw);
}

// A group of U steps in which lines break (wrapped sweeps).  sb[k] (wave-uniform) is the one line of the wave that
// wrapped in x when entering step k, or -1; uw[k] says the row wrapped (every line breaks).  Every line tests for its
// break on every step, branch-free.  (Letting only the lane that owns the breaking line close its run with an extra
// step(0) under a divergent branch, then taking the plain step wave-wide, measured 20 % SLOWER at 256^3 / 512^3.)
template <int LPL, int U, bool CHECK, typename W>
__device__ __forceinline__ bool steps_break_group(W (&w)[LPL], const unsigned (&v)[U], const int (&sb)[U],
                                                  const bool (&uw)[U], int lane4) {
  bool m = false;
#pragma unroll
  for (int k = 0; k < U; k++) {
#pragma unroll
    for (int j = 0; j < LPL; j++) {
      const bool brk = uw[k] || (sb[k] == lane4 + j);
      m |= w[j].template step_brk<CHECK>((int)((v[k] >> (8 * j)) & 0xffu), brk);
    }
  }
  return m;
}
template <bool DO_GLCM, bool DO_GLRLM, bool FUSED>
__device__ __forceinline__ void flush_block_hist(const u32 *lds, const HistLayout &h, int Nr, int slot,
                                                 u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc) {
  __syncthreads();
  const int Ng = h.Ng;
  if (FUSED) {
    const int Q = Ng + 1, P = (h.RS + 1) * Q;  // word strides
    u32 *gd = glcm_acc + (size_t)slot * Ng * Ng;
    for (int i = threadIdx.x; i < Ng * Ng; i += blockDim.x) {
      const int p = i / Ng, c = i - p * Ng;
      if (p == c) continue;  // diagonal comes from the GLRLM in finalize
      u32 v = 0;
      for (int l = 0; l <= h.RS; l++) v += lds[(p + 1) * P + l * Q + c + 1];
      if (v) atomicAdd(gd + i, v);
    }
    u32 *rd = glrlm_acc + (size_t)slot * Ng * Nr;
    for (int i = threadIdx.x; i < Ng * h.RS; i += blockDim.x) {  // the "long" slot RS is not a GLRLM bin
      const int p = i / h.RS, l = i - p * h.RS;
      u32 v = 0;
      for (int c = 0; c <= Ng; c++) v += lds[(p + 1) * P + l * Q + c];
      if (v) atomicAdd(rd + (size_t)p * Nr + l, v);
    }
    for (int i = threadIdx.x; i < Ng * h.RL; i += blockDim.x) {  // lengths RS+1 .. RS+RL
      const int p = i / h.RL, l = h.RS + (i - p * h.RL);
      const u32 v = lds[h.g0 + i];
      if (v && l < Nr) atomicAdd(rd + (size_t)p * Nr + l, v);
    }
    return;
  }
  if (DO_GLCM) {
    u32 *dst = glcm_acc + (size_t)slot * Ng * Ng;
    for (int i = threadIdx.x; i < Ng * Ng; i += blockDim.x) {
      const u32 v = lds[i];
      if (v) atomicAdd(dst + i, v);
    }
  }
  if (DO_GLRLM) {
    const u32 *hr = lds + h.glrlm0;
    u32 *dst = glrlm_acc + (size_t)slot * Ng * Nr;
    for (int i = threadIdx.x; i < h.RS * Ng; i += blockDim.x) {
      const u32 v = hr[i];
      if (v) atomicAdd(dst + (size_t)(i % Ng) * Nr + (i / Ng), v);
    }
  }
}

#ifndef PRAD_SWEEP_UNROLL
#define PRAD_SWEEP_UNROLL 8
#endif

struct __attribute__((packed)) u32_unaligned { u32 v; };
struct __attribute__((packed)) u16_unaligned { unsigned short v; };
// LPL adjacent level bytes as one (possibly unaligned) global load
template <int LPL>
__device__ __forceinline__ u32 load_lines(const uint8_t *p) {
#ifdef PRAD_DBG_NOLOAD  // ablation build: synthetic levels, no memory traffic
  const u32 x = (u32)(size_t)p * 2654435761u;
  return ((x >> 7) & 0x1f1f1f1fu) + 0x01010101u;
#else
  if (LPL == 4) return reinterpret_cast<const u32_unaligned *>(p)->v;
  if (LPL == 2) return reinterpret_cast<const u16_unaligned *>(p)->v;
  return *p;
#endif
}

// Wave-uniform position of a wave of wrapped lines, kept entirely in SGPRs and advanced incrementally:
//   off     byte offset of (march t, row u, column b) in the padded level volume
//   u, b    current row / first column of the wave
//   ent_uw  the row wrapped when entering the current step (every line of the wave breaks)
//   ent_sb  x-offset (0..CW-1) of the one line that wrapped in x when entering the current step, or -1
struct WrapPos {
  long long off;
  int u, b, ent_sb;
  bool ent_uw;
};
struct WrapGeo {
  int du, dx, NU, NX;
  long long delta;    // sM + du*sU + dx: offset change of a step without wraps
  long long uwrapfix; // -du*NU*sU: correction when the row wraps
};
template <int CW>
__device__ __forceinline__ void wrap_advance(WrapPos &p, const WrapGeo &g) {
  p.off += g.delta;
  p.u += g.du;
  p.ent_uw = false;
  if (p.u < 0 || p.u >= g.NU) {
    p.u -= g.du * g.NU;
    p.off += g.uwrapfix;
    p.ent_uw = true;
  }
  p.ent_sb = -1;
  if (g.dx != 0) {
    p.b += g.dx;
    if (p.b < 0 || p.b >= g.NX) {
      p.b -= g.dx * g.NX;
      p.off -= (long long)g.dx * g.NX;
    }
    // dx > 0: the line at offset NX - b now sits at column 0 (b = 0: offset 0);  dx < 0: offset NX-1-b sits at NX-1
    int sb = g.dx > 0 ? (p.b == 0 ? 0 : g.NX - p.b) : g.NX - 1 - p.b;
    p.ent_sb = sb < CW ? sb : -1;
  }
}
// true if the next `steps` advances cause no row wrap, no base wrap and no line of the wave to wrap in x
template <int CW>
__device__ __forceinline__ bool wrap_quiet(const WrapPos &p, const WrapGeo &g, int steps) {
  bool q = true;
  if (g.du > 0) q = q && (p.u + steps < g.NU);
  else if (g.du < 0) q = q && (p.u - steps >= 0);
  if (g.dx > 0) q = q && (p.b + steps <= g.NX - CW);       // all offsets NX - b_k stay >= CW
  else if (g.dx < 0) q = q && (p.b - steps >= 0) && (p.b <= g.NX - 1 - CW);
  return q;
}

// Angles whose march dimension is NOT the contiguous axis (see the header comment).
template <bool DO_GLCM, bool DO_GLRLM, bool LONG, bool FUSED, int LPL>
__global__ void __launch_bounds__(1024, 8) sweep_lines_kernel(SweepSet set, const uint8_t *__restrict__ L, int Ng,
                                                              int Nr, int RS, u32 *__restrict__ glcm_acc,
                                                              u32 *__restrict__ glrlm_acc, int *__restrict__ multi,
                                                              int *__restrict__ work, int *__restrict__ flags) {
  constexpr int CW = 64 * LPL;  // lines per wave
  constexpr int U = PRAD_SWEEP_UNROLL;
  extern __shared__ u32 lds[];
  if (flags[0]) return;  // irregular levels: the generic path will redo this call
  const HistLayout h = hist_layout(DO_GLCM, DO_GLRLM, FUSED, Ng, RS);
  if (FUSED && (unsigned)(size_t)((lds_u32 *)lds) != 0u) {  // the fused walker addresses its table from LDS address 0
    if (threadIdx.x == 0) atomicExch(flags + 2, 1);
    return;
  }
  for (int i = threadIdx.x; i < h.words; i += blockDim.x) lds[i] = 0;
  __syncthreads();

  const SweepDesc &D = set.d[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long long nwaves = (long long)gridDim.x * wpb;
  Walker<DO_GLCM, DO_GLRLM, LONG, FUSED> w[LPL];
#pragma unroll
  for (int j = 0; j < LPL; j++) w[j].init(lds, h, Nr, glrlm_acc + (size_t)D.slot * Ng * Nr, lane);
  bool seen_multi = false;
  const int lane4 = lane * LPL;
  const int NM = D.NM, NU = D.NU, NX = D.NX, du = D.du, dx = D.dx;
  const long long sU = D.sU;
  WrapGeo geo;
  geo.du = du; geo.dx = dx; geo.NU = NU; geo.NX = NX;
  geo.delta = D.sM + (long long)du * D.sU + dx;
  geo.uwrapfix = -(long long)du * NU * D.sU;

  // Chunks (one NM-step walk of 64*LPL lines) are handed out dynamically, one atomic per chunk: a chunk is a long
  // serial walk, so a static split leaves whole walks of imbalance between CUs that host 1 vs 2 workgroups.
  (void)nwaves;
  for (;;) {
    int grabbed = 0;
    if (lane == 0) grabbed = atomicAdd(&work[blockIdx.y], 1);
    const int chunk = __builtin_amdgcn_readfirstlane(grabbed);
    if (chunk >= D.chunks) break;
    // wave-uniform chunk coordinates
    const int u0 = chunk / D.LXc;
    const int xfirst = (chunk - u0 * D.LXc) * CW;
    bool dead[LPL];
    bool anydead_l = false;
#pragma unroll
    for (int j = 0; j < LPL; j++) {
      dead[j] = xfirst + lane4 + j >= NX;  // lines beyond the row end (partial last chunk)
      anydead_l = anydead_l || dead[j];
    }
    const bool anydead = __ballot(anydead_l) != 0;
    // lines beyond the row end read zeros (= voxels outside the ROI: their events land in the ignored row 0), so a
    // partial last chunk runs on the same fast paths as a full one
    u32 livemask = 0;
#pragma unroll
    for (int j = 0; j < LPL; j++) livemask |= dead[j] ? 0u : (0xffu << (8 * j));
#pragma unroll
    for (int j = 0; j < LPL; j++) w[j].begin_line();

    WrapPos pos;
    pos.off = (long long)u0 * sU + xfirst;
    pos.u = u0;
    pos.b = xfirst;
    pos.ent_sb = -1;
    pos.ent_uw = false;
    int t0 = 0;
    for (; t0 + U <= NM; t0 += U) {
      u32 v[U];
      bool risky_l = false;
#pragma unroll
      for (int j = 0; j < LPL; j++) risky_l = risky_l || w[j].risky(U);
      const bool quiet = !pos.ent_uw && pos.ent_sb < 0 && wrap_quiet<CW>(pos, geo, U);
      if (quiet) {
        // no line break and no dead line inside this group: constant stride
        const uint8_t *pl = L + pos.off + lane4;
#pragma unroll
        for (int k = 0; k < U; k++) {
          v[k] = load_lines<LPL>(pl);
          pl += geo.delta;
        }
        if (anydead) {
#pragma unroll
          for (int k = 0; k < U; k++) v[k] &= livemask;
        }
        pos.off += (long long)U * geo.delta;
        pos.u += U * du;
        pos.b += U * dx;
        if (LONG && __ballot(risky_l) != 0) {  // some run may exceed RS: clamp + long-run test per step
#pragma unroll
          for (int k = 0; k < U; k++) {
#pragma unroll
            for (int j = 0; j < LPL; j++) w[j].template step<true>((int)((v[k] >> (8 * j)) & 0xffu));
          }
        } else {
          steps_plain_group<LPL, U>(w, v);
        }
      } else {
        int sb[U];
        bool uw[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
          v[k] = load_lines<LPL>(L + pos.off + lane4) & livemask;
          sb[k] = pos.ent_sb;
          uw[k] = pos.ent_uw;
          wrap_advance<CW>(pos, geo);
        }
        if (LONG && __ballot(risky_l) != 0) seen_multi |= steps_break_group<LPL, U, true>(w, v, sb, uw, lane4);
        else seen_multi |= steps_break_group<LPL, U, false>(w, v, sb, uw, lane4);   // no run can exceed RS in this group
      }
    }
    for (int t = t0; t < NM; t++) {  // remainder steps
      const u32 v = load_lines<LPL>(L + pos.off + lane4) & livemask;
#pragma unroll
      for (int j = 0; j < LPL; j++) {
        const bool brk = pos.ent_uw || (pos.ent_sb == lane4 + j);
        const int cur = (int)((v >> (8 * j)) & 0xffu);
        seen_multi |= w[j].step_brk(cur, brk);
      }
      wrap_advance<CW>(pos, geo);
    }
#pragma unroll
    for (int j = 0; j < LPL; j++) seen_multi |= w[j].end_line();
  }
  if (DO_GLRLM && !FUSED && seen_multi) multi[D.slot] = 1;
  flush_block_hist<DO_GLCM, DO_GLRLM, FUSED>(lds, h, Nr, D.slot, glcm_acc, glrlm_acc);
}

// The angle along the contiguous axis.  A wave owns 64 consecutive rows (row = flattened (z,y)); it stages
// 64 rows x 64 voxels through LDS (coalesced in, one row per lane out) and walks them with the same Walker.
#define PRAD_ROW_PITCH 80  // bytes per staged row: 64 data + 16 pad => conflict-free ds_read_b128 per lane
template <bool DO_GLCM, bool DO_GLRLM, bool LONG, bool FUSED>
__global__ void __launch_bounds__(512) sweep_rows_kernel(const uint8_t *__restrict__ L, long long nrows, int NX,
                                                         int pitch, int slot, int Ng, int Nr, int RS,
                                                         u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc,
                                                         int *__restrict__ multi, int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;
  const HistLayout h = hist_layout(DO_GLCM, DO_GLRLM, FUSED, Ng, RS);
  if (FUSED && (unsigned)(size_t)((lds_u32 *)lds) != 0u) {
    if (threadIdx.x == 0) atomicExch(flags + 2, 1);
    return;
  }
  for (int i = threadIdx.x; i < h.words; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  // per-wave staging tile after the histograms (16-byte aligned)
  uint8_t *tile = reinterpret_cast<uint8_t *>(lds + ((h.words + 3) & ~3)) + (size_t)wave * 64 * PRAD_ROW_PITCH;
  const long long ngroups = (nrows + 63) / 64;
  const long long nwaves = (long long)gridDim.x * wpb;
  Walker<DO_GLCM, DO_GLRLM, LONG, FUSED> w;
  w.init(lds, h, Nr, glrlm_acc + (size_t)slot * Ng * Nr, lane);
  bool seen_multi = false;
  const bool vec16 = (pitch & 15) == 0 && ((uintptr_t)L & 15) == 0;   // any NX: the piece that straddles NX is masked

  for (long long grp = (long long)blockIdx.x * wpb + wave; grp < ngroups; grp += nwaves) {
    const long long r0 = grp * 64;
    w.begin_line();
    for (int xc = 0; xc < NX; xc += 64) {
      // ---- stage rows r0..r0+63, columns xc..xc+63 ----
      if (vec16) {
        // lane -> (row j*16 + lane/4, 16-byte piece lane%4): one instruction covers 16 rows x 64 B
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int rr = j * 16 + (lane >> 2);
          const int cx = xc + (lane & 3) * 16;
          uint4 q = make_uint4(0, 0, 0, 0);
          if (r0 + rr < nrows && cx < NX) {
            q = *reinterpret_cast<const uint4 *>(L + (r0 + rr) * pitch + cx);
            const int valid = NX - cx;             // bytes of this piece that belong to the row (the rest is the pad)
            if (valid < 16) {
              u32 *qw = reinterpret_cast<u32 *>(&q);
#pragma unroll
              for (int wd = 0; wd < 4; wd++) {
                const int keep = valid - 4 * wd;
                qw[wd] = keep >= 4 ? qw[wd] : (keep <= 0 ? 0u : (qw[wd] & ((1u << (8 * keep)) - 1u)));
              }
            }
          }
          *reinterpret_cast<uint4 *>(tile + rr * PRAD_ROW_PITCH + (lane & 3) * 16) = q;
        }
      } else {
        const bool xin = xc + lane < NX;
#pragma unroll 8
        for (int rr = 0; rr < 64; rr++) {
          uint8_t b = 0;
          if (xin && r0 + rr < nrows) b = L[(r0 + rr) * pitch + xc + lane];
          tile[rr * PRAD_ROW_PITCH + lane] = b;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // ---- each lane walks its own row (rows >= nrows and columns >= NX were staged as zeros) ----
      const uint4 *row = reinterpret_cast<const uint4 *>(tile + lane * PRAD_ROW_PITCH);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint4 d = row[q];
        const u32 wds[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (LONG && __ballot(w.risky(4)) != 0) {
#pragma unroll
            for (int b = 0; b < 4; b++) w.template step<true>((int)((wds[k] >> (8 * b)) & 0xffu));
          } else {
#pragma unroll
            for (int b = 0; b < 4; b++) w.template step<false>((int)((wds[k] >> (8 * b)) & 0xffu));
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    seen_multi |= w.end_line();
  }
  if (DO_GLRLM && !FUSED && seen_multi) multi[slot] = 1;
  flush_block_hist<DO_GLCM, DO_GLRLM, FUSED>(lds, h, Nr, slot, glcm_acc, glrlm_acc);
}

// ---- fused mode post-pass: one wave per (level i, angle a) --------------------------------------------------
//   GLCM diagonal  out[i][i][a] = sum_len (len-1) * GLRLM[a][i][len]     (pairs inside runs)
//   multi[a]      |= some run of level i is longer than 1, or row i of the (off-diagonal) GLCM is not empty:
//                    cheap sufficient conditions for "some line of angle a holds >= 2 masked voxels"
//                    (cmatrices.c:524-534); multi_check_kernel settles the angles they leave open.
__global__ void __launch_bounds__(64) glcm_diag_resolve_kernel(const u32 *__restrict__ glcm_acc,
                                                               const u32 *__restrict__ glrlm_acc, int Ng, int Nr,
                                                               int Na, double *__restrict__ glcm_out,
                                                               int *__restrict__ multi) {
  const int i = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
  const u32 *row = glrlm_acc + ((size_t)a * Ng + i) * Nr;
  unsigned long long pairs = 0;
  int found = 0;
  for (int r = lane; r < Nr; r += 64) {
    const u32 v = row[r];
    pairs += (unsigned long long)r * v;
    found |= (r > 0 && v != 0);
  }
  const u32 *grow = glcm_acc + ((size_t)a * Ng + i) * Ng;
  for (int j = lane; j < Ng; j += 64) found |= (grow[j] != 0);
  for (int o = 32; o > 0; o >>= 1) pairs += __shfl_xor(pairs, o);
  if (lane == 0) glcm_out[((size_t)i * Ng + i) * Na + a] = (double)pairs;
  if (__ballot(found) != 0 && lane == 0) multi[a] = 1;
}

// Exact test for the angles the conditions above leave open (every masked voxel isolated along the angle):
// one lane per line start, early exit once the flag is known.  Rarely does any work.
struct AngleSet {
  int count;
  int off[PRAD_MAX_SWEEP][3];
};
// T: element type of the packed volume (uint8_t, or the 16-bit elements of kernels_sweepfw2.h); pitch in ELEMENTS
template <typename T>
__global__ void __launch_bounds__(256) multi_check_kernel(AngleSet A, const T *__restrict__ L, int Nz, int Ny,
                                                          int Nx, int pitch, int *__restrict__ multi) {
  const int a = blockIdx.y;
  if (multi[a]) return;
  const int dz = A.off[a][0], dy = A.off[a][1], dx = A.off[a][2];
  const long long n = (long long)Nz * Ny * Nx, plane = (long long)Ny * Nx;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int z = (int)(i / plane);
    const int r = (int)(i - (long long)z * plane);
    int y = r / Nx, x = r - y * Nx;
    const int pz = z - dz, py = y - dy, px = x - dx;
    if ((unsigned)pz < (unsigned)Nz && (unsigned)py < (unsigned)Ny && (unsigned)px < (unsigned)Nx) continue;
    int cnt = 0;
    while ((unsigned)z < (unsigned)Nz && (unsigned)y < (unsigned)Ny && (unsigned)x < (unsigned)Nx) {
      cnt += L[((long long)z * Ny + y) * pitch + x] != 0;
      if (cnt > 1) {
        multi[a] = 1;
        return;
      }
      z += dz; y += dy; x += dx;
    }
  }
}

// acc (angle-major u32) -> reference layout float64 (fused mode: off-diagonal entries only)
__global__ void finalize_glcm_kernel(const u32 *__restrict__ acc, const u32 *__restrict__ glrlm_acc, int Ng, int Nr,
                                     int Na, int diag_from_runs, double *__restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Ng * Ng * Na;
  if (idx >= total) return;
  const int a = (int)(idx % Na);
  const long long ij = idx / Na;
  const int i = (int)(ij / Ng), j = (int)(ij - (long long)i * Ng);
  if (diag_from_runs && i == j) return;  // written by glcm_diag_resolve_kernel
  out[idx] = (double)acc[(size_t)a * Ng * Ng + ij];
}

// The two GLCM post-passes as one launch (five tiny kernels per volume were 17 % of a 256^3 call): the first nb1
// workgroups convert the off-diagonal counts, the others resolve the diagonal from the runs, one wave per (level, angle).
__global__ void __launch_bounds__(256) finalize_glcm_diag_kernel(const u32 *__restrict__ glcm_acc, u32 *glrlm_acc, int Ng, int Nr,
                                                                 int Na, int nb1, double *__restrict__ glcm_out,
                                                                 int *__restrict__ multi, int restore_from = -1) {
  if ((int)blockIdx.x < nb1) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)Ng * Ng * Na;
    if (idx >= total) return;
    const int a = (int)(idx % Na);
    const long long ij = idx / Na;
    const int i = (int)(ij / Ng), j = (int)(ij - (long long)i * Ng);
    if (i == j) return;  // the diagonal comes from the runs (below)
    glcm_out[idx] = (double)glcm_acc[(size_t)a * Ng * Ng + ij];
    return;
  }
  const int pair = ((int)blockIdx.x - nb1) * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pair >= Ng * Na) return;
  const int i = pair % Ng, a = pair / Ng;
  u32 *row = glrlm_acc + ((size_t)a * Ng + i) * Nr;
  unsigned long long pairs = 0, longer = 0;   // sum (len-1) * runs; runs of length >= 2
  int found = 0;
  for (int r = lane; r < Nr; r += 64) {
    const u32 v = row[r];
    pairs += (unsigned long long)r * v;
    if (r > 0) longer += v;
    found |= (r > 0 && v != 0);
  }
  const u32 *grow = glcm_acc + ((size_t)a * Ng + i) * Ng;
  for (int j = lane; j < Ng; j += 64) found |= (grow[j] != 0);
  for (int o = 32; o > 0; o >>= 1) pairs += __shfl_xor(pairs, o);
  if (lane == 0) glcm_out[((size_t)i * Ng + i) * Na + a] = (double)pairs;
  if (__ballot(found) != 0 && lane == 0) multi[a] = 1;
  // restore_from >= 0 (two-table walk with SKIP1, kernels_sweepfw2.h): the line angles did not record their runs of length
  // 1.  Every voxel of level i lies on exactly one line of every angle, so sum_len len * GLRLM_a[i][len] is the same number
  // N_i for all angles; the angle along x (slot restore_from) recorded all of its runs:
  //   GLRLM_a[i][1] = N_i - sum_{len >= 2} len * GLRLM_a[i][len] = N_i - (pairs + longer)
  if (restore_from >= 0 && a != restore_from) {
    const u32 *xrow = glrlm_acc + ((size_t)restore_from * Ng + i) * Nr;
    unsigned long long nvox = 0;
    for (int r = lane; r < Nr; r += 64) nvox += (unsigned long long)(r + 1) * xrow[r];
    for (int o = 32; o > 0; o >>= 1) {
      nvox += __shfl_xor(nvox, o);
      longer += __shfl_xor(longer, o);
    }
    if (lane == 0) row[0] = (u32)(nvox - (pairs + longer));
  }
}

// flags / sticky (deferred calls, both may be null): the levels verdict of the call is latched by this launch too
__global__ void finalize_glrlm_kernel(const u32 *__restrict__ acc, const int *__restrict__ multi, int Ng, int Nr,
                                      int Na, double *__restrict__ out, const int *__restrict__ flags = nullptr,
                                      int *__restrict__ sticky = nullptr) {
  if (sticky && blockIdx.x == 0 && threadIdx.x == 0 && (flags[0] || flags[2])) sticky[0] = 1;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Ng * Nr * Na;
  if (idx >= total) return;
  const int a = (int)(idx % Na);
  const long long gr = idx / Na;
  const int r = (int)(gr % Nr);
  // cmatrices.c:524-534: an angle without any line of >= 2 masked voxels loses its run-length-1 column
  out[idx] = (r == 0 && !multi[a]) ? 0.0 : (double)acc[(size_t)a * Ng * Nr + gr];
}

}  // namespace prad
