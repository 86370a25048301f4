// kernels_sweepfw.h -- fixed-window line sweeps: the 12 GLCM + GLRLM angles that march along z or y, gfx950.
//
// Same arithmetic as kernels_sweep.h (one event per RUN END into the fused LDS table H[prev][len][cur], GLCM
// diagonal recovered from the GLRLM), different geometry.  A wave owns a WINDOW of 64*K columns (K = 4 or 8
// adjacent voxels per lane = one dword / dwordx2 load per march step) that covers whole rows of the packed level
// volume, and marches it along z (or y).  Lines of an angle with dx != 0 drift through the window one column per
// step: their run state `pl` lives in lane registers that are RENAMED from step to step (an 8-step group is fully
// unrolled, so column j of step k is register (j - k*dx) mod K) and exactly one register per step crosses to the
// neighbouring lane through a DPP wave shift.  The previous levels of the drifting lines are the previous row
// shifted by one byte (v_alignbyte + one DPP move per step).  There are NO line breaks inside a walk: no periodic
// pad, no unaligned reads, no per-lane wrap logic -- every step of every group runs the 6-VALU + 1 ds_add plain
// path of kernels_sweep.h unless a run is about to outgrow the table (checked path, same as there).
//   * window placement: left-aligned for dx >= 0, right-aligned for dx < 0, so columns beyond the row only ever lie
//     on the side the lines LEAVE through: a line that walks off the row closes itself on the first zero column,
//     lines never enter from the padding.  When the row fills the window exactly (Nx == 64*K, e.g. 512 with K = 8)
//     the line that leaves is closed by one extra predicated ds_add on lane 63 (dx > 0) or lane 0 (dx < 0).
//   * work units: a walk of NM march steps is cut into pieces of CL steps so that every wave slot of the GPU gets
//     several.  A piece that does not begin at a line start begins DEAD: it knows the previous row but not how long
//     the open runs already are, so every line ignores its first event; the piece in which a run STARTED owns the
//     run and keeps walking past its end (the TAIL) until all of its open runs have ended.  Events are thereby
//     recorded exactly once and no stitching pass is needed.
//   * rows of an angle with dy != 0 are walked wrapped ((u0 + t*dy) mod NU) as in kernels_sweep.h: the wrap is
//     wave-uniform (all lines of the window end, new ones begin).
#pragma once
#include "kernels_sweep.h"

namespace prad {

// -DPRAD_FW_STAMPS: per-phase cycle accounting of the walk ("where does the issue time go").  Every wave
// keeps one s_memtime stamp and adds the cycles since the last stamp to the phase that just ended; lane 0 writes the sums to
// prad_fw_stamps[wave][phase] at the end (read back by prad_debug_fw_stamps, scripts/r05_fw_stamps.py).  A stamp is an SMEM
// instruction + s_waitcnt lgkmcnt(0): it drains the ds_adds in flight, ~5 stamps per plain group of ~7000 wave cycles.
enum FwPhase { FP_INIT, FP_GRAB, FP_CTRL, FP_ISSUE, FP_WAIT_ROWS, FP_WAIT_PACK, FP_GROUP, FP_PACK, FP_SLOW, FP_FLUSH, FP_DRAIN,
               FP_TOTAL, FP_REALTIME, FP_NGROUP, FP_NGROUP_PACK, FP_NSLOW, FP_COUNT };
#ifdef PRAD_FW_STAMPS
__device__ unsigned long long prad_fw_stamps[2 * 8192 * FP_COUNT];   // [PACK][wave][phase]
struct FwClock {
  unsigned long long last, first, rt0;
  unsigned acc[FP_COUNT];
  __device__ __forceinline__ void start() {
#pragma unroll
    for (int i = 0; i < FP_COUNT; i++) acc[i] = 0;
    rt0 = __builtin_amdgcn_s_memrealtime();
    first = last = __builtin_readcyclecounter();
  }
  template <int PH>
  __device__ __forceinline__ void lap() {
    const unsigned long long now = __builtin_readcyclecounter();
    acc[PH] += (unsigned)(now - last);
    last = now;
  }
  template <int PH>
  __device__ __forceinline__ void count() { acc[PH]++; }
  __device__ __forceinline__ void store(int slot) {
    acc[FP_TOTAL] = (unsigned)(__builtin_readcyclecounter() - first);
    acc[FP_REALTIME] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);
    const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && w < 8192) {
#pragma unroll
      for (int i = 0; i < FP_COUNT; i++) prad_fw_stamps[((size_t)slot * 8192 + w) * FP_COUNT + i] = acc[i];
    }
  }
};
#else
struct FwClock {
  __device__ __forceinline__ void start() {}
  template <int PH>
  __device__ __forceinline__ void lap() {}
  template <int PH>
  __device__ __forceinline__ void count() {}
  __device__ __forceinline__ void store(int) {}
};
#endif

struct FwDesc {
  int slot;          // index of the angle in the caller's list (output column)
  int NM, NU;        // extents of the march and row dimensions
  int du, dx;        // motion per march step in the row dim / along the contiguous axis
  long long sM, sU;  // byte strides of the march and row dimensions in the packed volume
  int CL, pieces;    // march steps per piece (multiple of 8), pieces per walk
  int chunks;        // NU * pieces
  int dom_kind;      // XCD-aware hand-out (FwSet::xcd): 0 = an XCD owns the pieces p = its id mod 8 (z-march roles: a piece is a
                     // z-slab), 1 = it owns an eighth of the rows (y-march roles: a row is a z plane)
};
struct FwSet {
  int count;         // line angles (roles 0 .. count-1)
  int NX;
  int pitch;         // bytes per row of the packed volume (row r of rowzero[] starts at r * pitch)
  int nrows;         // rows of the packed volume
  int xcd;           // 1: chunks are handed out per XCD (each XCD's L2 then serves ONE z-slab of the level volume to all roles)
  int first_block[PRAD_MAX_SWEEP + 1];   // role r owns workgroups [first_block[r], first_block[r + 1]) of the 1-D grid
  FwDesc d[PRAD_MAX_SWEEP];
};

// The NEXT volume's pack (int32 level + uint8 mask -> level*4 bytes, row flags, "irregular level" flag) as a side job of
// this launch: every wave of the line roles owns the 1024-voxel units pw, pw + W, pw + 2W, ... (pw = its index among the W
// packing waves), issues a unit's loads in front of a plain group of its walk and converts / stores behind it.  The pack is
// HBM-bound and the walk is issue-bound, so the pack's memory time disappears behind the walk's VALU / LDS work and only
// its VALU instructions remain (a pure pack kernel of its own takes 0.15 ms per 512^3 volume and cannot share a CU's
// issue slots with the sweep without slowing it down by as much; packing whole units BETWEEN the chunks of a walk costs
// twice what the interleaved form does -- a wave that waits for HBM is missed by its SIMD: profiles/r03_probes.md).
// Needs the linear layout (pitch == NX, hence NX % 16 == 0) and 16-byte aligned arrays; n16 == 0: nothing to pack.
struct PackJob {
  const int *image;
  const uint8_t *mask;
  uint8_t *levels;
  uint8_t *levels16;  // two-table walk (kernels_sweepfw2.h PackWave16): 16-bit level*4 elements; `levels` then holds plain level bytes
  uint8_t *rowzero;   // zeroed by the caller (fused-table walk only)
  int *flags;         // the packed volume's own flag words ([0] irregular level under the mask, [3] some voxel outside the ROI)
  long long n16;      // 16-voxel pieces
  int NX, Ng;
  int every;          // a wave loads one of its units in front of every `every`-th plain group of its walk
};

#define PRAD_FW_U 8
#ifndef PRAD_FW_MAXG
#define PRAD_FW_MAXG 4           // plain groups one margin check may vouch for (RS >= 4 * 8 steps at 32 levels)
#endif
#define PRAD_FW_WORK_STRIDE 64   // ints between two chunk counters (256 B: a line of their own)
#define PRAD_FW_DOMAINS 8        // chunk counters per role (one per XCD)
// The whole layout of kernels_sweep.h (fused table H, long-run table G, dummies) sits Q = 4(Ng+1) bytes into the LDS,
// because the run state kept here is s = level*P + len*Q with len counted from 1 (the exec-masked step adds Q to every
// line after the event lanes took their fresh value, see fw_plain_word): the bin of an event is s + cur, as there.
// Level 0 (a stretch of voxels outside the ROI, or a line that has not seen a voxel yet) owns the block [0, P + Q) in
// front of the real rows: whatever is added there is never read.  A DEAD line (one that must ignore its next event:
// a piece that does not begin at a line start knows the previous row but not how long the open runs are) is a line
// whose state says level 0 although its previous level is not 0: s = age*Q < P + Q = alive0.  The plain path walks it
// like any other line for up to RS steps -- its ignored event is a ds_add into the level-0 block -- and the checked path
// keeps it from growing out of that block.  (Until round 3 dead states sat BEHIND the table, s >= deadbase, which an open
// run of a high level and several hundred voxels also reached: its end was then taken for a dead line's and dropped.)
__host__ __device__ inline size_t fw_lds_bytes(const HistLayout &h) { return 4 * ((size_t)h.words + (size_t)(h.Ng + 1)); }


struct FwTab {   // wave-uniform constants of the fused table (layout: hist_layout(true, true, true, Ng, RS))
  u32 *rl_long;
  int Nr, P4, Q, lenlim, gB, RL4, RS, RL, dummy0b, alive0, deadmax;
  unsigned Qinv;
  __device__ __forceinline__ void init(const HistLayout &h, int Nr_, u32 *rl_long_) {
    rl_long = rl_long_;
    Nr = Nr_;
    Q = 4 * (h.Ng + 1);
    P4 = (h.RS + 1) * (h.Ng + 1);
    lenlim = (h.RS + 1) * Q;   // len*Q of the first run length without a slot of its own
    RS = h.RS;
    RL = h.RL;
    RL4 = 4 * h.RL;
    gB = Q + 4 * h.g0 - RL4 - 4 * h.RS;
    Qinv = (unsigned)((0x100000000ull + (unsigned)Q - 1) / (unsigned)Q);
    dummy0b = Q + 4 * h.dummy0;
    alive0 = (h.RS + 2) * Q;     // smallest state of a line inside a run of a real level (level 1, len 1)
    deadmax = (h.RS + 1) * Q;    // a dead line's state stops growing here (checked path)
  }
};

// a run of level lv (!= 0) longer than RS just ended: record its length (LDS table G, or a wave-aggregated L2 atomic)
//   lb = (len - 1) * Q
__device__ __noinline__ void fw_long_event(const FwTab &T, int lv, int lb) {
  const int idx = (int)__umulhi((unsigned)lb, T.Qinv);  // len - 1
  if (idx < T.RS + T.RL) {
    lds_bump(T.gB + __mul24(lv, T.RL4) + (idx << 2));
    return;
  }
  const unsigned key = ((unsigned)lv << 20) | (unsigned)idx;
  bool pending = true;
  while (pending) {
    const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)key);
    const bool same = key == first;
    const unsigned long long m = __ballot(same);
    if (same) {
      if ((int)(__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63))
        atomicAdd(&T.rl_long[(size_t)(lv - 1) * T.Nr + idx], (u32)__popcll(m));
      pending = false;
    }
  }
}

// One voxel-step of one line, every case handled: dead lines, runs beyond the table (clamped bin + length record).
//   s  level*P + len*Q of the open run (unclamped; 0 = a line that has not seen a voxel yet), or < alive0 = dead / level 0
//   x, c = level*4 of the previous / current voxel
template <bool LONG>
__device__ __forceinline__ void fw_checked(const FwTab &T, int dummy, int &s, int x, int c, bool tail) {
  const bool chg = c != x;
  const bool alive = s >= T.alive0;
  const bool ev = chg && alive;
  int bin = s;
  if (LONG) {
    const int lb = s - __mul24(x, T.P4);   // len*Q
    bin = s - lb + min(lb, T.lenlim);
    if (ev && x != 0 && lb >= T.lenlim) fw_long_event(T, x >> PRAD_FUSED_SHIFT, lb - T.Q);
  }
  lds_bump(ev ? bin + c : dummy);
  s = select_i32(chg, tail ? 0 : __mul24(c, T.P4) + T.Q, alive ? s + T.Q : min(s + T.Q, T.deadmax));
}

// The plain step of four lines whose levels are the byte lanes of c (current) and x (previous).  Only valid when no
// run can outgrow its slots (margin()) -- dead lines are fine, their events land in the dead zone.
// Exec-masked: v_cmpx leaves only the lanes whose level changed active; they compute the bin (state + level byte),
// bump it and take the fresh state level*P; then every lane adds Q (one more voxel on the open run / the first voxel
// of the fresh one).  4 VALU + 1 SALU + 1 ds_add per voxel-step and the LDS only sees the event lanes
// (the branch-free select form below costs 7 VALU and sends the other lanes to dummy words).
// Lines inside a stretch of voxels outside the ROI (level 0) run the same step: their "run" ends in the scratch words in
// front of the table; FwWave::calm_zero() keeps their length from growing (it would walk the address into the table).
__device__ __forceinline__ void fw_plain_word(const FwTab &T, int dummy, u32 one, int &p0, int &p1, int &p2, int &p3, u32 c, u32 x) {
#ifndef PRAD_FW_NOASM
  int t;
#ifdef PRAD_DBG_NOBUMP   // ablation build: everything but the LDS atomic
#define PRAD_FW_DSADD ""
#else
#define PRAD_FW_DSADD "ds_add_u32 %[t], %[one]\n\t"
#endif
#ifdef PRAD_FW_DSLATE   // A/B (round 6): the atomic behind the fresh-state instruction (one VALU between the address and its use)
#define PRAD_FW_COL(J, PJ)                                                                                              \
  "v_cmpx_ne_u32_sdwa vcc, %[c], %[x] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                    \
  "v_add_u32_sdwa %[t], %[" PJ "], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"      \
  "v_mul_u32_u24_sdwa %[" PJ "], %[P4], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
  PRAD_FW_DSADD                                                                                                          \
  "s_mov_b64 exec, -1\n\t"                                                                                              \
  "v_add_u32 %[" PJ "], %[Q], %[" PJ "]\n\t"
#else
#define PRAD_FW_COL(J, PJ)                                                                                              \
  "v_cmpx_ne_u32_sdwa vcc, %[c], %[x] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                    \
  "v_add_u32_sdwa %[t], %[" PJ "], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"      \
  PRAD_FW_DSADD                                                                                                          \
  "v_mul_u32_u24_sdwa %[" PJ "], %[P4], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
  "s_mov_b64 exec, -1\n\t"                                                                                              \
  "v_add_u32 %[" PJ "], %[Q], %[" PJ "]\n\t"
#endif
  asm volatile(PRAD_FW_COL(0, "p0") PRAD_FW_COL(1, "p1") PRAD_FW_COL(2, "p2") PRAD_FW_COL(3, "p3")
               : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [t] "=&v"(t)
               : [c] "v"(c), [x] "v"(x), [one] "v"(one), [P4] "s"(T.P4), [Q] "s"(T.Q)
               : "vcc", "memory");
#undef PRAD_FW_COL
#else
  int *p[4] = {&p0, &p1, &p2, &p3};
  int addr[4], fresh[4], grown[4];
  bool chg[4];
#pragma unroll
  for (int j = 0; j < 4; j++) chg[j] = __builtin_amdgcn_ubfe(c, 8 * j, 8) != __builtin_amdgcn_ubfe(x, 8 * j, 8);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    addr[j] = *p[j] + (int)__builtin_amdgcn_ubfe(c, 8 * j, 8);
    fresh[j] = (int)__umul24(__builtin_amdgcn_ubfe(c, 8 * j, 8), (unsigned)T.P4) + T.Q;
    grown[j] = *p[j] + T.Q;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) lds_bump(chg[j] ? addr[j] : dummy);
#pragma unroll
  for (int j = 0; j < 4; j++) *p[j] = select_i32(chg[j], fresh[j], grown[j]);
#endif
}

__device__ __forceinline__ u32 fw_shr1(u32 v) {  // lane i <- lane i-1, lane 0 <- 0
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ u32 fw_shl1(u32 v) {  // lane i <- lane i+1, lane 63 <- 0
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

// previous levels of the lines that ARRIVE at this lane's columns: the previous row shifted by dx columns
template <int K, int DX>
__device__ __forceinline__ void fw_make_x(const u32 (&P)[K / 4], u32 (&X)[K / 4]) {
  constexpr int KW = K / 4;
  if (DX == 0) {
#pragma unroll
    for (int w = 0; w < KW; w++) X[w] = P[w];
  } else if (DX > 0) {  // column j gets the byte of column j-1
    const u32 in = fw_shr1(P[KW - 1]);
#pragma unroll
    for (int w = KW - 1; w >= 1; w--) X[w] = __builtin_amdgcn_alignbyte(P[w], P[w - 1], 3);
    X[0] = __builtin_amdgcn_alignbyte(P[0], in, 3);
  } else {              // column j gets the byte of column j+1
    const u32 in = fw_shl1(P[0]);
#pragma unroll
    for (int w = 0; w < KW - 1; w++) X[w] = __builtin_amdgcn_alignbyte(P[w + 1], P[w], 1);
    X[KW - 1] = __builtin_amdgcn_alignbyte(in, P[KW - 1], 1);
  }
}

struct __attribute__((packed)) u64_unaligned { unsigned long long v; };
struct __attribute__((packed)) u128_unaligned_fw { u32 a, b, c, d; };
template <int KW>
__device__ __forceinline__ void fw_load(const uint8_t *p, u32 (&v)[KW]) {
#ifdef PRAD_DBG_NOLOAD  // ablation build: synthetic iid levels 1..32, no memory traffic
#pragma unroll
  for (int w = 0; w < KW; w++) {
    u32 x = ((u32)(size_t)p + 977u * w) * 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    v[w] = (((x >> 3) & 0x1f1f1f1fu) + 0x01010101u) << PRAD_FUSED_SHIFT;
  }
#else
#ifdef PRAD_DBG_L2ONLY  // ablation build: every read lands in the first MB of the volume (wrong results, L2-resident traffic)
  p = (const uint8_t *)((size_t)p & ~(size_t)0xFFFFFFF) + ((size_t)p & 0xFFFF8);
#endif
  if (KW == 4) {
    const u128_unaligned_fw q = *reinterpret_cast<const u128_unaligned_fw *>(p);
    v[0] = q.a;
    v[KW > 1 ? 1 : 0] = q.b;
    v[KW > 2 ? 2 : 0] = q.c;
    v[KW - 1] = q.d;
  } else if (KW == 2) {
    const unsigned long long q = reinterpret_cast<const u64_unaligned *>(p)->v;
    v[0] = (u32)q;
    v[KW - 1] = (u32)(q >> 32);
  } else {
    v[0] = reinterpret_cast<const u32_unaligned *>(p)->v;
  }
#endif
}

// One wave's share of a PackJob (see there).  begin() issues the loads of the wave's next unit in front of a plain group of
// the walk when one is due, finish() converts and stores it behind the group: the loads fly under the group's VALU / LDS
// work, no wave ever waits for the pack.  Everything the hot loop carries for the pack sits in VECTOR registers (pinned:
// the loop is at the limit of its scalar registers, and wave-uniform pack state left to the compiler went into SGPRs and
// pushed loop scalars out -- v_readlane inside the plain groups, the walk alone 7 % slower); the kernel has 50 to spare.
struct PackWave {
  unsigned long long img, msk, lev;   // this lane's next piece: image + 64 t, mask + 16 t, levels + 16 t (byte addresses)
  unsigned long long step16;          // pieces between two units of this wave
  int units_left;                     // units this wave still owns
  int last_lanes;                     // lanes that hold a piece in the wave's LAST unit
  int tick, every, loaded, bad, lane;
  int4 q0, q1, q2, q3;
  uint4 m;
  template <typename T>
  static __device__ __forceinline__ void pin(T &x) { asm volatile("" : "+v"(x)); }
  __device__ __forceinline__ PackWave(const PackJob &J, long long pw, long long W) {
    lane = threadIdx.x & 63;
    const long long n16 = J.n16, first = pw * 64;
    long long units = 0;
    if (pw >= 0 && first < n16) units = (n16 - first + W * 64 - 1) / (W * 64);
    units_left = (int)units;
    const long long last_first = first + (units - 1) * W * 64;
    last_lanes = units > 0 ? (int)((n16 - last_first) < 64 ? (n16 - last_first) : 64) : 0;
    const unsigned long long t = (unsigned long long)(first + lane);
    img = (unsigned long long)(size_t)J.image + 64ull * t;
    msk = (unsigned long long)(size_t)J.mask + 16ull * t;
    lev = (unsigned long long)(size_t)J.levels + 16ull * t;
    step16 = (unsigned long long)(W * 64);
    tick = 0;
    every = J.every;
    loaded = 0;
    bad = 0;
    pin(img); pin(msk); pin(lev); pin(step16); pin(units_left); pin(last_lanes); pin(tick); pin(every); pin(loaded); pin(bad);
  }
  __device__ __forceinline__ void load() {
    q0 = q1 = q2 = q3 = make_int4(0, 0, 0, 0);
    m = make_uint4(0, 0, 0, 0);
    if (units_left > 1 || lane < last_lanes) {
      const int4 *im4 = reinterpret_cast<const int4 *>((size_t)img);
      m = *reinterpret_cast<const uint4 *>((size_t)msk);
      q0 = im4[0];
      q1 = im4[1];
      q2 = im4[2];
      q3 = im4[3];
    }
    loaded = 1;
    pin(loaded);
  }
  __device__ __forceinline__ void begin() {
    if (units_left <= 0) return;
    tick++;
    pin(tick);
    if (tick < every) return;
    tick = 0;
    pin(tick);
    load();
  }
  __device__ __forceinline__ void finish(const PackJob &J) {
    if (!loaded) return;
    loaded = 0;
    pin(loaded);
    const bool mine = units_left > 1 || lane < last_lanes;
    if (mine) {
      const int lv[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
      const u32 mw[4] = {m.x, m.y, m.z, m.w};
      u32 ow[4];
      // Four voxels at a time with packed-byte arithmetic (the per-voxel form cost ~12 VALU per voxel on a kernel that is
      // bound by its instruction issue, profiles/r03_probes.md section 3): low bytes of the four levels gathered by three
      // v_perm.  FAST PATH, decided once for the 16 voxels of the piece: every voxel lies inside the mask and every level
      // is regular (1..Ng, no bits beyond the low byte) -- then the packed bytes are the levels and nothing is outside the
      // ROI.  With q = pk - 0x01010101 and r = q + (0x80 - Ng) * 0x01010101, bit 7 of some byte of pk | q | r is set iff
      // some byte of pk is 0 or > Ng (the lowest irregular byte sees no borrow / carry from below; Ng <= 44 on this path).
      const u32 K80 = 0x80808080u, K7F = 0x7f7f7f7fu, K01 = 0x01010101u;
      const u32 radd = (0x80u - (u32)J.Ng) * K01;
      u32 pk[4], nzw[4], badbits = 0, wideall = 0, nzall = K80;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const u32 l0 = (u32)lv[w * 4], l1 = (u32)lv[w * 4 + 1], l2 = (u32)lv[w * 4 + 2], l3 = (u32)lv[w * 4 + 3];
        wideall |= l0 | l1 | l2 | l3;
        const u32 p01 = __builtin_amdgcn_perm(l1, l0, 0x0c0c0400u), p23 = __builtin_amdgcn_perm(l3, l2, 0x0c0c0400u);
        pk[w] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
        nzw[w] = ((mw[w] & K7F) + K7F) | mw[w];                 // bit 7 of every non-zero mask byte
        nzall &= nzw[w];
        const u32 q = pk[w] - K01;
        badbits |= pk[w] | q | (q + radd);
      }
      if ((((badbits | ~nzall) & K80) | (wideall & 0xffffff00u)) == 0) {
#pragma unroll
        for (int w = 0; w < 4; w++) ow[w] = pk[w] << PRAD_FUSED_SHIFT;
      } else {
        // the general form, word by word: mask bytes widened to 0xff, "level > Ng" and "level 0 under the mask" as any-byte
        // tests; a word that is not plain -- a level with bits beyond its low byte, an irregular level under the mask --
        // takes the exact per-voxel form (a rare, divergent branch): the results are the same in every case
        const u32 ngadd = (0x7fu - (u32)J.Ng) * K01;
#pragma unroll
        for (int w = 0; w < 4; w++) {
          const u32 l0 = (u32)lv[w * 4], l1 = (u32)lv[w * 4 + 1], l2 = (u32)lv[w * 4 + 2], l3 = (u32)lv[w * 4 + 3];
          const u32 wide = (l0 | l1 | l2 | l3) & 0xffffff00u;
          const u32 nz = nzw[w] & K80;
          const u32 m8 = (nz << 1) - (nz >> 7);                                            // 0xff for every non-zero mask byte
          const u32 lvm = pk[w] & m8;                                                      // levels under the mask, 0 elsewhere
          const u32 gt = (((lvm & K7F) + ngadd) | lvm) & K80;                              // a byte > Ng
          const u32 t = pk[w] | ~m8;
          const u32 zero_in = (t - K01) & ~t & K80;                                        // != 0 iff a masked voxel has level 0
          u32 o = lvm << PRAD_FUSED_SHIFT;
          if (wide | gt | zero_in) {
            o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const bool in = (mw[w] >> (8 * b)) & 0xffu;
              const int l = lv[w * 4 + b];
              const bool regular = in && l >= 1 && l <= J.Ng;
              bad |= in && !regular;
              o |= (regular ? ((u32)l << PRAD_FUSED_SHIFT) : 0u) << (8 * b);   // (an irregular level packs as 0: never a table index)
            }
          }
          ow[w] = o;
        }
        u32 zb = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) zb |= (ow[w] - K01) & ~ow[w] & K80;
        if (zb) {   // the piece holds a voxel outside the ROI: flag its row(s) (a 16-voxel piece spans at most two rows)
          J.flags[3] = 1;
          const unsigned e0 = (unsigned)(lev - (unsigned long long)(size_t)J.levels);       // (volumes stay below 2^31 voxels)
          J.rowzero[e0 / (unsigned)J.NX] = 1;
          J.rowzero[(e0 + 15u) / (unsigned)J.NX] = 1;
        }
      }
      *reinterpret_cast<uint4 *>((size_t)lev) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
    img += 64ull * step16;
    msk += 16ull * step16;
    lev += 16ull * step16;
    units_left--;
    pin(img); pin(msk); pin(lev); pin(units_left); pin(bad);
  }
  // whatever the walk left over (a launch whose walk is much shorter than the pack, or no walk at all)
  __device__ __forceinline__ void drain(const PackJob &J) {
#pragma unroll 1
    while (units_left > 0) {
      load();
      finish(J);
    }
    if (bad) J.flags[0] = 1;
  }
};

#define FW_BYTE(W, j) ((int)__builtin_amdgcn_ubfe((W)[(j) >> 2], 8 * ((j) & 3), 8))

template <bool LONG, int K, int DX, bool HASPAD, bool PACK>
struct FwWave {
  static constexpr int KW = K / 4;
  static constexpr int U = PRAD_FW_U;
  const FwTab &T;
  int dummy, lane, edge_lane;
  u32 one;         // the ds_add operand, pinned in a VGPR
  static constexpr bool haspad = HASPAD;   // the row is shorter than the window (NX != 64*K)
  u32 cmask[KW];   // byte lanes of this lane's window columns that lie inside the row
  u32 calm[KW];    // byte lanes of window columns no line can be open on (beyond the row, not next to its exit side)
  int pl[K];       // run state of the line that arrives at column j at the next step
  u32 P[KW];       // levels of the previous row (this lane's columns)

  __device__ __forceinline__ FwWave(const FwTab &T_, int NX) : T(T_) {
    lane = threadIdx.x & 63;
    dummy = T.dummy0b + 4 * lane;
    edge_lane = haspad ? -1 : (DX > 0 ? 63 : (DX < 0 ? 0 : -1));
    one = 1;
    asm volatile("" : "+v"(one));
    const int col0 = first_col(NX);
#pragma unroll
    for (int w = 0; w < KW; w++) {
      u32 m = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int col = col0 + 4 * w + b;
        if (col >= 0 && col < NX) m |= 0xffu << (8 * b);
      }
      cmask[w] = m;
      u32 q = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int col = col0 + 4 * w + b;
        const bool outside = col < 0 || col >= NX, feeder_outside = col - DX < 0 || col - DX >= NX;
        if (outside && feeder_outside) q |= 0xffu << (8 * b);
      }
      calm[w] = q;
    }
  }
  // volume column of this lane's first window column (window right-aligned for dx < 0)
  __device__ __forceinline__ int first_col(int NX) const { return (threadIdx.x & 63) * K - (DX < 0 ? 64 * K - NX : 0); }

  __device__ __forceinline__ void reset_lines(int state) {
#pragma unroll
    for (int j = 0; j < K; j++) pl[j] = state;
  }
  __device__ __forceinline__ void load_row(const uint8_t *p, u32 (&v)[KW]) const {
    fw_load<KW>(p, v);
    if (haspad) {
#pragma unroll
      for (int w = 0; w < KW; w++) v[w] &= cmask[w];
    }
  }
  // cross-lane move of the one line that changes lane, after the line that leaves the row was closed
  template <bool PLAIN>
  __device__ __forceinline__ void rotate_reg(int &r, int xlevel) {
    if (!haspad) {
      if (PLAIN) {
#if defined(PRAD_DBG_NOEDGE)   // ablation build (wrong results): the line that leaves the row is not closed
#elif !defined(PRAD_FW_NOASM)
        const unsigned long long em = DX > 0 ? 0x8000000000000000ull : 1ull;   // lane 63 / lane 0
        asm volatile("s_mov_b64 exec, %[m]\n\tds_add_u32 %[r], %[one]\n\ts_mov_b64 exec, -1" ::[m] "s"(em), [r] "v"(r), [one] "v"(one) : "memory");
#else
        lds_bump(lane == edge_lane ? r : dummy);
#endif
      } else if (lane == edge_lane) {
        fw_checked<LONG>(T, dummy, r, xlevel, 0, false);
      }
    }
    r = (int)(DX > 0 ? fw_shr1((u32)r) : fw_shl1((u32)r));
  }
  // one march step on the canonical register assignment (register j = column j), all cases handled
  __device__ __forceinline__ void single_step(const u32 (&C)[KW], bool tail) {
    u32 X[KW];
    fw_make_x<K, DX>(P, X);
#pragma unroll
    for (int j = 0; j < K; j++) fw_checked<LONG>(T, dummy, pl[j], FW_BYTE(X, j), FW_BYTE(C, j), tail);
    if (DX > 0) {
      rotate_reg<false>(pl[K - 1], FW_BYTE(C, K - 1));
      const int in = pl[K - 1];
#pragma unroll
      for (int j = K - 1; j >= 1; j--) pl[j] = pl[j - 1];
      pl[0] = in;
    } else if (DX < 0) {
      rotate_reg<false>(pl[0], FW_BYTE(C, 0));
      const int in = pl[0];
#pragma unroll
      for (int j = 0; j < K - 1; j++) pl[j] = pl[j + 1];
      pl[K - 1] = in;
    }
#pragma unroll
    for (int w = 0; w < KW; w++) P[w] = C[w];
  }
  // U plain steps with renamed registers (no line dead, no run near the end of the table)
  // pk_late (PRAD_FW_PACK_LATE): the pack unit's loads go out behind the FIRST step of the group instead of in front of the
  // group, so that the wait for the group's level rows (vmcnt(0): the compiler cannot count across the conditional) does not
  // also wait for the pack's HBM loads
  __device__ __forceinline__ void plain_group(const u32 (&v)[U][KW], PackWave *pk_late = nullptr) {
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (k == 1 && pk_late) pk_late->begin();
      u32 X[KW];
      fw_make_x<K, DX>(P, X);
#pragma unroll
      for (int w = 0; w < KW; w++) {
        constexpr int dummy_ce = 0;
        (void)dummy_ce;
        const int r0 = (((4 * w + 0 - k * DX) % K) + K) % K, r1 = (((4 * w + 1 - k * DX) % K) + K) % K;
        const int r2 = (((4 * w + 2 - k * DX) % K) + K) % K, r3 = (((4 * w + 3 - k * DX) % K) + K) % K;
        fw_plain_word(T, dummy, one, pl[r0], pl[r1], pl[r2], pl[r3], v[k][w], X[w]);
      }
      if (DX > 0) rotate_reg<true>(pl[(((K - 1 - k) % K) + K) % K], FW_BYTE(v[k], K - 1));
      if (DX < 0) rotate_reg<true>(pl[k % K], FW_BYTE(v[k], 0));
#pragma unroll
      for (int w = 0; w < KW; w++) P[w] = v[k][w];
    }
    if (DX != 0 && (U % K) != 0) {   // K = 16: U steps leave the assignment rotated by U registers -- back to register j = column j
      int tmp[K];
#pragma unroll
      for (int j = 0; j < K; j++) tmp[j] = pl[(((j - U * DX) % K) + K) % K];
#pragma unroll
      for (int j = 0; j < K; j++) pl[j] = tmp[j];
    }
  }
  // How far the lines of this lane are from needing the checked path, as the largest "bytes into the row" of any open
  // run: a plain walk of n steps is safe while margin + n*Q <= lenlim.  YOUNG: dead lines count with their age in the
  // dead zone (which has the same RS+1 slots); otherwise a dead line reads as "unsafe".
  template <bool YOUNG>
  __device__ __forceinline__ unsigned margin() {
    u32 X[KW];
    fw_make_x<K, DX>(P, X);
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
      const int xj = FW_BYTE(X, j);
      unsigned a = (unsigned)(pl[j] - __mul24(xj, T.P4));   // len*Q
      // A dead line (state below alive0 although its previous level is a real one): its age in the dead zone while YOUNG,
      // unsafe otherwise.  Said explicitly since round 5: until then a dead line read as "huge" only because the subtraction
      // went negative -- not so for previous level 1 once the checked path has clamped the age at deadmax = 1 * P: a = 0, the
      // line walked on on the plain path, grew into alive0 = (level 1, length 1) and its next change was recorded as a short
      // run of level 1 (found by scripts/r05_stress.py: 138 x 58 x 300, runs of level 1 longer than RS + 1 across a piece
      // boundary; fixture tests/golden/regress/fw_long_runs_138x58x300.npz)
#ifdef PRAD_DBG_R5BUG1   // test build: the margin as it was until round 5 (tests/test_gpu_stress.py must find it)
      if (YOUNG) a = min(a, (unsigned)pl[j]);
#else
      if (pl[j] < T.alive0 && xj != 0) a = YOUNG ? (unsigned)pl[j] : 0x7fffffffu;   // (not ~0u: the callers add the groups' growth to it)
#endif
      m = max(m, a);
    }
    return m;
  }
  // lines that never see a voxel (window columns beyond the row) must not look like runs about to outgrow the table;
  // the column next to the row's exit side is spared: the line on it is still open (it closes on the next step)
  __device__ __forceinline__ void calm_padding() {
    if (!haspad) return;
#pragma unroll
    for (int j = 0; j < K; j++)
      if ((calm[j >> 2] >> (8 * (j & 3))) & 0xffu) pl[j] = 0;
  }
  // The same for lines inside a stretch of voxels outside the ROI (previous level 0), called before every group that
  // touches a row holding such voxels: the stretch is no run, its "length" restarts (a dead line too: whatever run it
  // was waiting on has ended, and nobody records the end of a stretch), so it never grows more than U steps and its
  // end lands in the scratch words below the table.  2 VALU per column per group, on masked rows only; making the
  // step itself skip such lines costs a VALU per voxel-step AND a second copy of the group (measured +5 % on full
  // masks from the register copies between the two).
  __device__ __forceinline__ void calm_zero() {
    u32 X[KW];
    fw_make_x<K, DX>(P, X);
#pragma unroll
    for (int j = 0; j < K; j++)
      if (FW_BYTE(X, j) == 0) pl[j] = 0;
  }
  __device__ __forceinline__ bool any_alive() {
    u32 X[KW];
    fw_make_x<K, DX>(P, X);
    bool a = false;
#pragma unroll
    for (int j = 0; j < K; j++) a = a || (pl[j] >= T.alive0 && FW_BYTE(X, j) != 0);
    return __ballot(a) != 0;
  }
  __device__ __forceinline__ void run(const FwDesc &D, int NX, int pitch, long long nrows, const uint8_t *__restrict__ L,
                                      const uint8_t *__restrict__ rowzero, bool anyzero, int *work, int bx, int nblocks,
                                      PackWave &pk, const PackJob &pj, bool xcd, FwClock &clk) {
    const int NM = D.NM, NU = D.NU, du = D.du;
    const long long delta = D.sM + (long long)du * D.sU;
    // row numbers of the packed volume (rowzero[r] != 0: row r holds a voxel outside the ROI), wave-uniform like `off`
    const long long sMr = D.sM / pitch, sUr = D.sU / pitch, dri = sMr + (long long)du * sUr;
    // Row flags, 64 marching steps at a time: bit j of `zm` = flag of row ri0 + (j - 1) * dri (bit 0 is the row the
    // lines come from), one vector load + ballot per 56 steps; between reloads a group costs three SALU instructions
    // (per-group loads cost 4 % of a SIMD-issue-bound kernel; scalar loads share lgkmcnt with the ds_adds in flight and
    // their wait drains the LDS queue, +8 %).  A row wrap invalidates the window.
    const int rlast = (int)nrows - 1;                                   // (volumes stay below 2^31 voxels)
    const int zlane = (lane - 1) * (int)dri;
    auto zwindow = [&](long long r) -> unsigned long long {
      return __ballot(rowzero[min(max((int)r + zlane, 0), rlast)] != 0);
    };
    const uint8_t *lp = L + first_col(NX);
    // chunk hand-out: the first chunk of a wave is its index, the rest come from a counter that has a cache line to
    // itself (a dequeue word saturates near 90 grabs per microsecond: 12 angles sharing one line, or one angle cut
    // into 16 384 tiny chunks, serialised the kernel on it; dealing out MORE rounds statically measured 10 % slower at
    // 512^3 -- walks of wrapping rows and of the window edges do not cost the same)
    const int nwaves = (int)(nblocks * (blockDim.x >> 6));      // waves of this role (bx = workgroup index within it)
    const int wid = __builtin_amdgcn_readfirstlane((int)(bx * (blockDim.x >> 6) + (threadIdx.x >> 6)));  // wave-uniform
    // XCD-aware hand-out (xcd): the chunks of a role are split into 8 DOMAINS -- z-march roles by piece (= z-slab), y-march
    // roles by row (= z plane) -- and a wave serves the domain of its own XCD first, the others once that one is empty.
    // Every role then walks slab x of the level volume on XCD x at about the same time (chunks are handed out in the
    // order of the row a line STARTS the piece on), and that XCD's L2 serves the slab to all 12 roles instead of every
    // role streaming the whole volume through every L2.
    int dom = 0, tried = 0;
    const int ndom = xcd ? PRAD_FW_DOMAINS : 1;
    if (xcd) {
      int id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
      dom = id & (PRAD_FW_DOMAINS - 1);
    }
    for (int it = 0;; it++) {
      int piece, u0;
      if (!xcd) {
        int chunk;
        if (it == 0) {
          chunk = wid;
        } else {
#ifdef PRAD_DBG_NOGRAB      // ablation build: static round-robin hand-out, no dequeue atomics
          chunk = wid + it * nwaves;
#else
          int grabbed = 0;
          if (lane == 0) grabbed = atomicAdd(work, 1);
          chunk = nwaves + __builtin_amdgcn_readfirstlane(grabbed);
#endif
        }
        if (chunk >= D.chunks) break;
        piece = chunk / NU;
        u0 = chunk - piece * NU;  // piece-major: concurrent waves share planes
      } else {
        // size of domain `dom`
        int wd = NU, ulo = 0, np = 0;
        if (D.dom_kind == 0) {
          np = dom < D.pieces ? (D.pieces - dom + PRAD_FW_DOMAINS - 1) / PRAD_FW_DOMAINS : 0;
        } else {
          ulo = (int)((long long)NU * dom / PRAD_FW_DOMAINS);
          wd = (int)((long long)NU * (dom + 1) / PRAD_FW_DOMAINS) - ulo;
          np = D.pieces;
        }
        int grabbed = 0;
        if (lane == 0) grabbed = atomicAdd(work + dom * PRAD_FW_WORK_STRIDE, 1);
        const int i = __builtin_amdgcn_readfirstlane(grabbed);
        if (i >= wd * np) {      // this domain is done: help the next one
          if (++tried >= ndom) break;
          dom = (dom + 1) & (PRAD_FW_DOMAINS - 1);
          continue;
        }
        const int pi = i / wd, rs = ulo + (i - pi * wd);
        if (D.dom_kind == 0) {
          piece = dom + PRAD_FW_DOMAINS * pi;
          // rs = the row the line is on HALFWAY through the piece (roles that drift up and roles that drift down then
          // cover the same band of rows at the same time); its family index u0 = row at march step 0
          long long u = ((long long)rs - ((long long)piece * D.CL + D.CL / 2) * du) % NU;
          if (u < 0) u += NU;
          u0 = (int)u;
        } else {
          piece = pi;
          u0 = rs;
        }
      }
      const int t0 = piece * D.CL, t1 = min(NM, t0 + D.CL);
      int row = (int)((u0 + (long long)t0 * du) % NU);
      if (row < 0) row += NU;
      long long off = (long long)t0 * D.sM + (long long)row * D.sU;
      long long ri = (long long)t0 * sMr + (long long)row * sUr;   // row number of `off`
      unsigned long long zm = 0;   // zwindow() of the stretch being marched; zpos = bit of the current row
      int zpos = 65;
#ifdef PRAD_DBG_NODEAD     // ablation build (wrong results): every piece begins like a line start
      const bool starts = true;
#else
      const bool starts = t0 == 0 || (du > 0 && row == 0) || (du < 0 && row == NU - 1);
#endif
      int young = 0;  // plain groups that may still carry dead lines (their ignored events go to the dead zone)
      int safe = 0;   // plain groups the last margin check vouched for
      bool maybe_dead = !starts;
      if (starts) {
#pragma unroll
        for (int w = 0; w < KW; w++) P[w] = 0;
        reset_lines(0);
      } else {
        load_row(lp + (off - delta), P);
        reset_lines(0);            // dead: level 0 as far as the state goes, although P holds real levels
        young = 2;
      }
      int t = t0;
      clk.template lap<FP_GRAB>();
      bool wrap = false;  // the next step would leave the row range: all lines end first
      bool tail = false;  // past the end of the piece: only runs that began inside it are still recorded
      // One loop, two kinds of iteration: a plain group of U steps, or ONE slow step (every special case funnels into
      // the single inlined copy of single_step: instruction-cache footprint matters more than the rare paths' speed).
      for (;;) {
        bool closing = false, finish = false;   // closing: every line ends here = a step onto a row of zeros
        if (!tail && t >= t1) {
          if (t1 == NM || wrap) closing = finish = true;   // the lines end with the piece
          else tail = true;
#ifdef PRAD_DBG_NOTAIL   // ablation build (wrong results): pieces stop at their end
          if (tail) break;
#endif
        }
        if (tail && !closing) {
          if (t == NM || wrap) closing = finish = true;
          else if (!any_alive()) break;
        }
        if (!tail && !closing && wrap) closing = true;     // row wrap inside the piece: new lines begin behind it
        if (!tail && !closing) {
          const int room = du > 0 ? NU - row : (du < 0 ? row + 1 : (1 << 30));  // steps before the row range ends
          const bool grp = t + U <= t1 && room >= U;
          bool za = false;   // some row of this group (or the one before it) has voxels outside the ROI
          if (grp) {
#ifdef PRAD_DBG_NOZ      // ablation build (wrong results on partial masks): no row flags, never zero-aware
            za = false;
#else
            if (!rowzero) {
              za = true;
            } else if (anyzero) {
              if (zpos + U > 64) {
                zm = zwindow(ri);
                zpos = 1;
              }
              za = ((zm >> (zpos - 1)) & ((1ull << (U + 1)) - 1)) != 0;
            }
#endif
          }
          if (LONG && za) calm_zero();   // (without LONG every length a line can reach has its slot)
          if (grp && safe == 0) {
            calm_padding();
#ifdef PRAD_DBG_NOMARGIN   // ablation build (wrong on long runs): no margin checks
            if (true) {
#else
            if (!LONG && !maybe_dead) {
#endif
              safe = PRAD_FW_MAXG;   // every run length has its slot and no line is dead
            } else {
              // how many plain groups the longest open run (or the oldest dead line) leaves room for: up to PRAD_FW_MAXG
              // (round 5: one check per RUN of groups -- on iid levels no run is longer than a handful of voxels, four groups
              // fit the table's length slots -- instead of one per two groups: the check and the loop's scalar bookkeeping
              // were 15 % of a wave's cycles, profiles/r05_fw_phases.md)
              const unsigned m = young > 0 ? margin<true>() : margin<false>();
              // An event inside a run of n plain steps records at most len + n - 1 <= RS (the change is seen one step after
              // the run's last voxel), but the line that LEAVES a window-filling row is closed by rotate_reg in the step of
              // its last voxel: len + n.  Until round 5 the bound below was lenlim for both, so a run of exactly RS + 1
              // voxels that ended at the x edge on the plain path landed in the slot of "longer than RS" -- which only the GLCM
              // reads: lost for the GLRLM and the GLCM diagonal (found by scripts/r05_stress.py / r05_fw_bug_probe2.py on
              // the 256- and 512-wide crops of tests/golden/regress/fw_long_runs_138x58x300.npz: two runs of level 21, length 26 = RS + 1)
#ifdef PRAD_DBG_R5BUG2   // test build: the plain-path bound of edge roles as it was until round 5
              const unsigned lim = (unsigned)T.lenlim;
#else
              const unsigned lim = (unsigned)T.lenlim - ((!haspad && DX != 0) ? (unsigned)T.Q : 0u);
#endif
#pragma unroll
              for (int q = PRAD_FW_MAXG; q >= 1; q--) {
                if (safe == 0 && __ballot(m + q * U * T.Q > lim) == 0) safe = q;
              }
              if (safe > 0 && young == 0) maybe_dead = false;   // a dead line reads as unsafe in margin<false>
            }
          }
          if (grp && safe > 0) {
            // ng plain groups back to back with nothing between them but the pointer bumps
            int ng = min(safe, (t1 - t) / U);
            if (du != 0) ng = min(ng, room / U);
            if (young > 0) ng = min(ng, young);
            if (anyzero || !rowzero) ng = 1;       // (the row flags were read for one group)
#ifdef PRAD_FW_ONEGROUP
            ng = 1;
#endif
            clk.template lap<FP_CTRL>();
            u32 va[U][KW];
            auto load_group = [&](u32 (&v)[U][KW], long long o) __attribute__((always_inline)) {
              const uint8_t *p = lp + o;
#pragma unroll
              for (int k = 0; k < U; k++) {
                load_row(p, v[k]);
                p += delta;
              }
            };
            auto one_group = [&](const u32 (&v)[U][KW], bool first) __attribute__((always_inline)) {
#ifndef PRAD_FW_PACK_LATE
              if (PACK) pk.begin();            // (the next volume's pack rides along: loads out, ...
#endif
              if (!first) calm_padding();      // (the groups before let the padding lines grow)
#ifdef PRAD_FW_STAMPS
              clk.template lap<FP_ISSUE>();
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              if (PACK && pk.loaded) {
                clk.template lap<FP_WAIT_PACK>();
                clk.template count<FP_NGROUP_PACK>();
              } else {
                clk.template lap<FP_WAIT_ROWS>();
              }
              clk.template count<FP_NGROUP>();
#endif
#ifdef PRAD_FW_PACK_LATE
              plain_group(v, PACK ? &pk : nullptr);
#else
              plain_group(v);
#endif
              clk.template lap<FP_GROUP>();
              if (PACK) pk.finish(pj);         //  ... bytes stored behind the group's VALU / LDS work)
              clk.template lap<FP_PACK>();
              safe--;
              if (young > 0) young--;
              t += U;
              row += U * du;
              off += (long long)U * delta;
              ri += (long long)U * dri;
              zpos += U;
            };
            // (Rows of group g + 1 loaded while group g is walked -- a second register set -- was built and measured in round 5:
            // 124 VGPRs, the pack side job's pinned state spilled to scratch, 0.68 instead of 0.43 ms.  One set it is.)
            for (int gi = 0; gi < ng; gi++) {
              load_group(va, off);
              one_group(va, gi == 0);
            }
            if (du != 0 && (row < 0 || row >= NU)) wrap = true;
            continue;
          }
        }
        u32 c[KW];
        if (closing) {
#pragma unroll
          for (int w = 0; w < KW; w++) c[w] = 0;
        } else {
          load_row(lp + off, c);
        }
        clk.template lap<FP_CTRL>();
        single_step(c, tail);
        clk.template lap<FP_SLOW>();
        clk.template count<FP_NSLOW>();
        safe = 0;
        young = 0;
        if (closing) {
          if (finish) break;
          reset_lines(0);    // (P is the row of zeros already)
          row -= du * NU;
          off -= (long long)du * NU * D.sU;
          ri -= (long long)du * NU * sUr;
          zpos = 65;
          wrap = false;
          continue;
        }
        t++;
        zpos++;
        row += du;
        off += delta;
        ri += dri;
        if (du != 0 && (row < 0 || row >= NU)) wrap = true;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// The angle along the contiguous axis with the same exec-masked step: a wave owns 64 consecutive rows (row = flattened
// (z, y)), stages 64 x 64-voxel tiles through LDS (16 B per lane coalesced in, one row per lane out) and every lane walks
// its own row, four voxels (one staged word) per asm block.  Table layout and run state as above.
// ---------------------------------------------------------------------------------------------------------------
// ZA (zero aware): a stretch of voxels outside the ROI never touches the table (second compare: event lanes whose previous
// voxel is 0 drop out before the ds_add).  Without ZA -- the pack saw no voxel outside the ROI, flags[3] == 0 -- the only
// previous level 0 is the one in front of a row's first voxel, whose event (state 0) lands in the scratch words in front of
// the table: 4 VALU per voxel instead of 5.
template <bool ZA>
__device__ __forceinline__ void fw_row_word(const FwTab &T, u32 one, int &s, u32 c, u32 x) {
  int t;
#define PRAD_FW_RCOL(J)                                                                                          \
  "v_cmpx_ne_u32_sdwa vcc, %[c], %[x] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                             \
  "v_add_u32_sdwa %[t], %[s], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"   \
  "v_mul_u32_u24_sdwa %[s], %[P4], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
  "v_cmpx_ne_u32_sdwa vcc, %[x], %[z] src0_sel:BYTE_" #J " src1_sel:DWORD\n\t"                                  \
  "ds_add_u32 %[t], %[one]\n\t"                                                                                  \
  "s_mov_b64 exec, -1\n\t"                                                                                       \
  "v_add_u32 %[s], %[Q], %[s]\n\t"
#define PRAD_FW_RCOL_NZ(J)                                                                                       \
  "v_cmpx_ne_u32_sdwa vcc, %[c], %[x] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                             \
  "v_add_u32_sdwa %[t], %[s], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"   \
  "ds_add_u32 %[t], %[one]\n\t"                                                                                  \
  "v_mul_u32_u24_sdwa %[s], %[P4], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
  "s_mov_b64 exec, -1\n\t"                                                                                       \
  "v_add_u32 %[s], %[Q], %[s]\n\t"
  if (ZA) {
    u32 z = 0;      // zero aware: a stretch of unmasked voxels never touches the table
    asm volatile("" : "+v"(z));
    asm volatile(PRAD_FW_RCOL(0) PRAD_FW_RCOL(1) PRAD_FW_RCOL(2) PRAD_FW_RCOL(3)
                 : [s] "+v"(s), [t] "=&v"(t)
                 : [c] "v"(c), [x] "v"(x), [one] "v"(one), [P4] "s"(T.P4), [Q] "s"(T.Q), [z] "v"(z)
                 : "vcc", "memory");
  } else {
    asm volatile(PRAD_FW_RCOL_NZ(0) PRAD_FW_RCOL_NZ(1) PRAD_FW_RCOL_NZ(2) PRAD_FW_RCOL_NZ(3)
                 : [s] "+v"(s), [t] "=&v"(t)
                 : [c] "v"(c), [x] "v"(x), [one] "v"(one), [P4] "s"(T.P4), [Q] "s"(T.Q)
                 : "vcc", "memory");
  }
#undef PRAD_FW_RCOL
#undef PRAD_FW_RCOL_NZ
}

// the walk along x for workgroup bx of nblocks; the LDS table (layout h, zeroed by the caller) sits at address 0
template <bool LONG, bool ZA>
__device__ __forceinline__ void fw_rows_role(const HistLayout &h, const FwTab &T, u32 *lds, const uint8_t *__restrict__ L,
                                             long long nrows, int NX, int pitch, int bx, int nblocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int dummy = T.dummy0b + 4 * lane;
  u32 one = 1;
  asm volatile("" : "+v"(one));
  // per-wave staging tile behind the table and its dead zone (16-byte aligned)
  uint8_t *tile = reinterpret_cast<uint8_t *>(lds) + ((fw_lds_bytes(h) + 15) & ~(size_t)15) + (size_t)wave * 64 * PRAD_ROW_PITCH;
  // A wave's 64 rows lie 8 rows apart (group g: rows 512 (g / 8) + g % 8 + 8 i): neighbouring rows of a smooth image hold
  // nearly the same levels, and 64 ADJACENT rows sent the lanes of one ds_add to the same few bins (same-address atomics
  // serialise: the x angle of the smooth 512^3 volume took 0.12 ms against 0.06 on iid levels).  Every row is its own
  // set of cache lines either way.  (Small volumes keep adjacent rows: their last group of 512 would be mostly empty.)
  const int RSTEP = nrows >= 4096 ? 8 : 1;
  const long long ngroups = RSTEP == 8 ? ((nrows + 511) / 512) * 8 : (nrows + 63) / 64;
  const long long nwaves = (long long)nblocks * wpb;
  const bool vec16 = (pitch & 15) == 0 && ((uintptr_t)L & 15) == 0;
  for (long long grp = (long long)bx * wpb + wave; grp < ngroups; grp += nwaves) {
    const long long r0 = RSTEP == 8 ? (grp >> 3) * 512 + (grp & 7) : grp * 64;     // row of tile slot i: r0 + RSTEP * i
    int s = 0;     // run state of this lane's row
    u32 pw = 0;    // previous staged word (its last byte is the previous voxel)
    // vec16: lane -> (row j*16 + lane/4, 16-byte piece lane%4) of a tile; the pieces of the NEXT tile are loaded into registers
    // before the current one is walked (round 5: a wave waited ~2 us for each of its tiles with one other wave on its SIMD to
    // fill the gap -- kernels_sweepfw2.h sweep_fw2_rows_kernel measured the same walk at 0.121 ms without and 0.086 with)
    uint4 nx4[4];
    auto fetch = [&](int xc) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int rr = j * 16 + (lane >> 2);
        const int cx = xc + (lane & 3) * 16;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (r0 + RSTEP * rr < nrows && cx < NX) {
          q = *reinterpret_cast<const uint4 *>(L + (r0 + RSTEP * rr) * pitch + cx);
          const int valid = NX - cx;
          if (valid < 16) {
            u32 *qw = reinterpret_cast<u32 *>(&q);
#pragma unroll
            for (int wd = 0; wd < 4; wd++) {
              const int keep = valid - 4 * wd;
              qw[wd] = keep >= 4 ? qw[wd] : (keep <= 0 ? 0u : (qw[wd] & ((1u << (8 * keep)) - 1u)));
            }
          }
        }
        nx4[j] = q;
      }
    };
    if (vec16) fetch(0);
    for (int xc = 0; xc < NX; xc += 64) {
      if (vec16) {
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<uint4 *>(tile + (j * 16 + (lane >> 2)) * PRAD_ROW_PITCH + (lane & 3) * 16) = nx4[j];
        __builtin_amdgcn_wave_barrier();
        if (xc + 64 < NX) fetch(xc + 64);
      } else {
        const bool xin = xc + lane < NX;
#pragma unroll 8
        for (int rr = 0; rr < 64; rr++) {
          uint8_t b = 0;
          if (xin && r0 + RSTEP * rr < nrows) b = L[(r0 + RSTEP * rr) * pitch + xc + lane];
          tile[rr * PRAD_ROW_PITCH + lane] = b;
        }
      }
      __builtin_amdgcn_wave_barrier();
      const uint4 *row = reinterpret_cast<const uint4 *>(tile + lane * PRAD_ROW_PITCH);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint4 d = row[q];
        const u32 wds[4] = {d.x, d.y, d.z, d.w};
        // 16 steps: safe on the plain path while len*Q + 16 Q stays within the table's length slots
        const unsigned m = (pw >> 24) ? (unsigned)(s - __mul24((int)(pw >> 24), T.P4)) : 0u;
        if (LONG && __ballot(m + 16 * T.Q > (unsigned)T.lenlim) != 0) {
          // some run is within 16 steps of the table's last length slot: decide word by word (4 steps) -- on smooth images
          // nearly every 16-step stretch holds such a run in one of the 64 rows, and walking all of it on the checked
          // path doubled the x angle's time
#pragma unroll 1
          for (int k = 0; k < 4; k++) {
            const u32 c = wds[k], x = __builtin_amdgcn_alignbyte(c, pw, 3);
            const unsigned m4 = (pw >> 24) ? (unsigned)(s - __mul24((int)(pw >> 24), T.P4)) : 0u;
            if (__ballot(m4 + 4 * T.Q > (unsigned)T.lenlim) != 0) {
#pragma unroll
              for (int b = 0; b < 4; b++) fw_checked<LONG>(T, dummy, s, (int)((x >> (8 * b)) & 0xffu), (int)((c >> (8 * b)) & 0xffu), false);
            } else {
              fw_row_word<ZA>(T, one, s, c, x);
            }
            pw = c;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            fw_row_word<ZA>(T, one, s, wds[k], __builtin_amdgcn_alignbyte(wds[k], pw, 3));
            pw = wds[k];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    fw_checked<LONG>(T, dummy, s, (int)(pw >> 24), 0, false);   // the row ends: close its open run
  }
}

// the angle along x as a launch of its own (8-wave workgroups: table + 8 staging tiles).  As a ROLE of sweep_fw_kernel
// (16-wave workgroups next to the line roles') the same walk measured slower per CU -- 22.9 CU-ms instead of 14 at 512^3,
// the whole launch 0.477 ms against 0.38 + 0.055 for the two launches -- and its code made the line roles' kernel half as
// large again (profiles/r03_probes.md), so it stays a kernel of its own.
template <bool LONG>
__global__ void __launch_bounds__(1024) sweep_fw_rows_kernel(const uint8_t *__restrict__ L, long long nrows, int NX, int pitch,
                                                            int slot, int Ng, int Nr, int RS, u32 *__restrict__ glcm_acc,
                                                            u32 *__restrict__ glrlm_acc, int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;
  const HistLayout h = hist_layout(true, true, true, Ng, RS);
  if ((unsigned)(size_t)((lds_u32 *)lds) != 0u) {
    if (threadIdx.x == 0) atomicExch(flags + 2, 1);
    return;
  }
  for (int i = threadIdx.x; i < h.words + Ng + 1; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  FwTab T;
  T.init(h, Nr, glrlm_acc + (size_t)slot * Ng * Nr);
  if (flags[3] != 0) fw_rows_role<LONG, true>(h, T, lds, L, nrows, NX, pitch, (int)blockIdx.x, (int)gridDim.x);
  else fw_rows_role<LONG, false>(h, T, lds, L, nrows, NX, pitch, (int)blockIdx.x, (int)gridDim.x);   // (no voxel outside the ROI)
  flush_block_hist<true, true, true>(lds + Ng + 1, h, Nr, slot, glcm_acc, glrlm_acc);
}

// One launch per volume: a 1-D grid of one 16-wave workgroup per CU, cut into ROLES -- one per line angle -- plus, with PACK,
// the pack of the NEXT volume as a side job of the walking waves (PackJob; the instantiation without it spares the walk
// the side job's registers: 0.39 instead of 0.405 ms per 512^3 volume).  A volume whose pack found irregular levels
// (flags[0]) is skipped -- the generic kernels redo that call -- but the side job still runs.
template <bool LONG, int K, bool HASPAD, bool PACK>
__global__ void __launch_bounds__(1024) sweep_fw_kernel(FwSet set, PackJob pj, const uint8_t *__restrict__ L,
                                                        const uint8_t *__restrict__ rowzero, int Ng, int Nr, int RS,
                                                        u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc,
                                                        int *__restrict__ work, int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  int role = 0;
  while (role + 1 < set.count && (int)blockIdx.x >= set.first_block[role + 1]) role++;
  const int bx = (int)blockIdx.x - set.first_block[role], nblocks = set.first_block[role + 1] - set.first_block[role];
  const int wpb = (int)(blockDim.x >> 6);
  const long long pw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
  PackWave pk(pj, pw, (long long)gridDim.x * wpb);
  FwClock clk;
  clk.start();
#ifdef PRAD_FW_SETPRIO   // experiment: issue priority over co-resident waves of another launch
  __builtin_amdgcn_s_setprio(PRAD_FW_SETPRIO);
#endif
  if (flags[0] == 0) {     // (else: irregular levels, nothing to walk)
    const bool anyzero = flags[3] != 0;   // the pack saw a voxel outside the ROI (else the row flags are not read)
    const HistLayout h = hist_layout(true, true, true, Ng, RS);
    if ((unsigned)(size_t)((lds_u32 *)lds) != 0u) {  // table offsets are used as LDS addresses
      if (threadIdx.x == 0) atomicExch(flags + 2, 1);
    } else {
      for (int i = threadIdx.x; i < h.words + Ng + 1; i += blockDim.x) lds[i] = 0;
      __syncthreads();
      clk.lap<FP_INIT>();
      const FwDesc &D = set.d[role];
      FwTab T;
      T.init(h, Nr, glrlm_acc + (size_t)D.slot * Ng * Nr);
      int *wk = work + PRAD_FW_WORK_STRIDE * PRAD_FW_DOMAINS * role;
      if (D.dx == 0) {
        FwWave<LONG, K, 0, HASPAD, PACK> w(T, set.NX);
        w.run(D, set.NX, set.pitch, set.nrows, L, rowzero, anyzero, wk, bx, nblocks, pk, pj, set.xcd != 0, clk);
      } else if (D.dx > 0) {
        FwWave<LONG, K, 1, HASPAD, PACK> w(T, set.NX);
        w.run(D, set.NX, set.pitch, set.nrows, L, rowzero, anyzero, wk, bx, nblocks, pk, pj, set.xcd != 0, clk);
      } else {
        FwWave<LONG, K, -1, HASPAD, PACK> w(T, set.NX);
        w.run(D, set.NX, set.pitch, set.nrows, L, rowzero, anyzero, wk, bx, nblocks, pk, pj, set.xcd != 0, clk);
      }
      clk.lap<FP_CTRL>();
      flush_block_hist<true, true, true>(lds + Ng + 1, h, Nr, D.slot, glcm_acc, glrlm_acc);
      clk.lap<FP_FLUSH>();
    }
  }
  if (PACK) pk.drain(pj);
  clk.lap<FP_DRAIN>();
  clk.store(PACK ? 1 : 0);
}

}  // namespace prad
