// kernels_sweepfw2.h -- the fixed-window line sweeps for 44 .. ~160 grey levels (binWidth 25 on CT gives 60 - 160), gfx950.
//
// The fused table H[prev][len][cur] of kernels_sweepfw.h grows with (Ng+1)^2 x length slots: beyond 44 levels the 160 KB
// of LDS no longer hold the 16+ slots the plain path needs, and level*4 no longer fits the byte lanes the step reads.
// Same geometry here (a wave owns a window of 64*K columns that covers whole rows, lines with dx != 0 drift through
// renamed registers, pieces / dead lines / tails exactly as there), different accumulation:
//   * levels are 16-bit elements (level*4; 0 outside the ROI), two columns per dword, SDWA word selects;
//   * TWO tables, interleaved row by row at LDS address 0:  row(level) = [ A: cur 0..Ng | B: C copies of len 1..RS2 ],
//     S = 4 (Ng + 1 + C RS2) bytes; lane l uses copy l mod C of B (on iid levels nearly every run has length 1: the 64
//     lanes of a ds_add would otherwise share Ng addresses, and same-address atomics serialise at 2 cycles a lane).  A run end (prev, len, cur) adds to A[prev][cur] (the GLCM pair; the diagonal comes from
//     the runs as before) and to B[prev][len] (the GLRLM bin): 5 VALU + 2 ds_add + 1 SALU per voxel-step against
//     4 + 1 + 1 of the fused step -- about 1.3x its time -- but the table is (Ng+1)(Ng+1+RS2) words: 64 levels with 128
//     run lengths take 50 KB, 128 levels with 128 lengths 132 KB, and with that many length slots the checked path
//     (runs beyond the slots) is rare on any image;
//   * a line's state is two registers: p = level * S (its A row) and q = p + 4 Ng + 4 len (its B cursor); level 0 (a
//     stretch outside the ROI, a line that has not seen a voxel, a DEAD line of a piece that did not begin at a line
//     start) is row 0, which is never read.
// The angle along x is walked by the separate-table rows kernel of kernels_sweep.h on an 8-bit copy of the levels.
#pragma once
#include "kernels_sweepfw.h"

namespace prad {

struct Fw2Tab {
  u32 *rl_long;
  int Nr, Ng, RS2, C;
  int S, S4, K1, lenlim;
  __device__ __forceinline__ void init(int Ng_, int RS2_, int C_, int Nr_, u32 *rl_long_) {
    rl_long = rl_long_;
    Nr = Nr_;
    Ng = Ng_;
    RS2 = RS2_;
    C = C_;
    S4 = Ng_ + 1 + C_ * RS2_;
    S = 4 * S4;
    K1 = 4 * Ng_;              // q = p + K1 + 4 RS2 copy + 4 len: len = 1 of copy 0 sits right behind the A row
    lenlim = 4 * RS2_;         // 4 len of the last length slot
  }
};
__host__ __device__ inline size_t fw2_table_words(int Ng, int RS2, int C) { return (size_t)(Ng + 1) * (size_t)(Ng + 1 + C * RS2); }

// a run of level lv (!= 0) longer than RS2 just ended: one L2 atomic per distinct (level, length) of the wave
__device__ __noinline__ void fw2_long_event(const Fw2Tab &T, int lv, int idx) {
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

// One voxel-step of one line, every case handled.  x, c = level*4 of the previous / current voxel.
//   SKIP1: runs of length 1 are NOT recorded (the caller restores GLRLM[.][1] from the level counts: Fw2Wave::SKIP1)
template <bool LONG, bool SKIP1>
__device__ __forceinline__ void fw2_checked(const Fw2Tab &T, int k1, int dummy, int &p, int &q, int x, int c, bool tail) {
  const bool chg = c != x;
  const bool alive = p >= T.S;
  const bool ev = chg && alive;
  const int lb = q - p - k1;                   // 4 len (k1: this lane's K1 + offset of its B copy)
  const bool inlds = !LONG || lb <= T.lenlim;
  lds_bump(ev ? p + c : dummy);                // the pair (cur = 0: the run ended at a line end / outside the ROI: ignored)
  lds_bump((ev && inlds && !(SKIP1 && lb == 4)) ? q : dummy);   // the run
  if (LONG && ev && !inlds) fw2_long_event(T, x >> PRAD_FUSED_SHIFT, (lb >> 2) - 1);
  const int fp = tail ? 0 : __mul24(c, T.S4);
  const int grown = alive ? q + 4 : min(q + 4, k1 + T.lenlim);   // (a level-0 line's cursor stops at the last slot)
  q = select_i32(chg, fp + k1 + 4, grown);
  p = select_i32(chg, fp, p);
}

// The plain step of the two lines whose levels are the 16-bit halves of c (current) and x (previous); exec-masked like
// fw_plain_word.  Only valid while no run can outgrow its length slots (margin()).
template <bool SKIP1>
__device__ __forceinline__ void fw2_plain_word(const Fw2Tab &T, int k1, int k14, u32 one, int &p0, int &q0, int &p1, int &q1, u32 c, u32 x,
                                               unsigned long long &ev0, unsigned long long &ev1) {
  int t;
#ifdef PRAD_DBG_FW2_NOPAIR    // ablation builds (wrong results): the step without its pair / run atomic
#define PRAD_FW2_DSA ""
#else
#define PRAD_FW2_DSA "ds_add_u32 %[t], %[one]\n\t"
#endif
#ifdef PRAD_DBG_FW2_NORUN
#define PRAD_FW2_DSB(QJ) ""
#else
#define PRAD_FW2_DSB(QJ) "ds_add_u32 %[" QJ "], %[one]\n\t"
#endif
#define PRAD_FW2_COL(J, PJ, QJ)                                                                                          \
  "v_cmpx_ne_u32_sdwa vcc, %[c], %[x] src0_sel:WORD_" #J " src1_sel:WORD_" #J "\n\t"                                     \
  "v_add_u32_sdwa %[t], %[" PJ "], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #J "\n\t"       \
  PRAD_FW2_DSA PRAD_FW2_DSB(QJ)                                                                                          \
  "v_mul_u32_u24_sdwa %[" PJ "], %[S4], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #J "\n\t" \
  "v_add_u32 %[" QJ "], %[K1], %[" PJ "]\n\t"                                                                            \
  "s_mov_b64 exec, -1\n\t"                                                                                               \
  "v_add_u32 %[" QJ "], 4, %[" QJ "]\n\t"
// SKIP1: among the event lanes only those whose run is longer than one voxel (cursor - row != K1 + 4) add to B; vcc keeps
// the event lanes for the state update behind it.  On iid levels 31 of 32 runs have length 1: the second ds_add of the
// step, which made the two-table walk LDS-bound (0.73 -> 0.44 ms per 512^3 volume without it), goes out almost empty.
#define PRAD_FW2_COL_S(J, PJ, QJ)                                                                                        \
  "v_cmpx_ne_u32_sdwa vcc, %[c], %[x] src0_sel:WORD_" #J " src1_sel:WORD_" #J "\n\t"                                     \
  "v_add_u32_sdwa %[t], %[" PJ "], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #J "\n\t"       \
  PRAD_FW2_DSA                                                                                                           \
  "v_sub_u32 %[t], %[" QJ "], %[" PJ "]\n\t"                                                                             \
  "v_mul_u32_u24_sdwa %[" PJ "], %[S4], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #J "\n\t" \
  "v_cmpx_ne_u32_e64 %[sm], %[t], %[K14]\n\t"                                                                            \
  PRAD_FW2_DSB(QJ)                                                                                                       \
  "s_mov_b64 exec, vcc\n\t"                                                                                              \
  "v_add_u32 %[" QJ "], %[K1], %[" PJ "]\n\t"                                                                            \
  "s_mov_b64 exec, -1\n\t"                                                                                               \
  "v_add_u32 %[" QJ "], 4, %[" QJ "]\n\t"
#define PRAD_FW2_COL_M(J, PJ, QJ, EV)                                                                                    \
  "v_cmpx_ne_u32_sdwa vcc, %[c], %[x] src0_sel:WORD_" #J " src1_sel:WORD_" #J "\n\t"                                     \
  "v_add_u32_sdwa %[t], %[" PJ "], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #J "\n\t"       \
  PRAD_FW2_DSA                                                                                                           \
  "s_andn2_b64 exec, vcc, %[" EV "]\n\t"                                                                                 \
  PRAD_FW2_DSB(QJ)                                                                                                       \
  "s_mov_b64 exec, vcc\n\t"                                                                                              \
  "v_mul_u32_u24_sdwa %[" PJ "], %[S4], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #J "\n\t" \
  "v_add_u32 %[" QJ "], %[K1], %[" PJ "]\n\t"                                                                            \
  "s_mov_b64 %[" EV "], vcc\n\t"                                                                                         \
  "s_mov_b64 exec, -1\n\t"                                                                                               \
  "v_add_u32 %[" QJ "], 4, %[" QJ "]\n\t"
  if (SKIP1) {
#ifdef PRAD_FW2_SKIP_BY_STATE   // A/B build: "length 1" read from the line's state (v_sub + v_cmpx) instead of the event masks
    unsigned long long sm;
    asm volatile(PRAD_FW2_COL_S(0, "p0", "q0") PRAD_FW2_COL_S(1, "p1", "q1")
                 : [p0] "+v"(p0), [q0] "+v"(q0), [p1] "+v"(p1), [q1] "+v"(q1), [t] "=&v"(t), [sm] "=&s"(sm)
                 : [c] "v"(c), [x] "v"(x), [one] "v"(one), [S4] "s"(T.S4), [K1] "v"(k1), [K14] "v"(k14)
                 : "vcc", "memory");
#else
    // A line's run has length 1 exactly when the line had an event at its previous step: the event lanes of the previous
    // step (ev, a scalar mask that travels with the line's registers) leave before the run atomic -- no vector instruction
    asm volatile(PRAD_FW2_COL_M(0, "p0", "q0", "ev0") PRAD_FW2_COL_M(1, "p1", "q1", "ev1")
                 : [p0] "+v"(p0), [q0] "+v"(q0), [p1] "+v"(p1), [q1] "+v"(q1), [t] "=&v"(t), [ev0] "+s"(ev0), [ev1] "+s"(ev1)
                 : [c] "v"(c), [x] "v"(x), [one] "v"(one), [S4] "s"(T.S4), [K1] "v"(k1)
                 : "vcc", "scc", "memory");     // (s_andn2 writes SCC)
#endif
  } else {
    asm volatile(PRAD_FW2_COL(0, "p0", "q0") PRAD_FW2_COL(1, "p1", "q1")
                 : [p0] "+v"(p0), [q0] "+v"(q0), [p1] "+v"(p1), [q1] "+v"(q1), [t] "=&v"(t)
                 : [c] "v"(c), [x] "v"(x), [one] "v"(one), [S4] "s"(T.S4), [K1] "v"(k1)
                 : "vcc", "memory");
  }
#undef PRAD_FW2_COL
#undef PRAD_FW2_COL_S
#undef PRAD_FW2_COL_M
}

struct __attribute__((packed)) u128_unaligned { u32 a, b, c, d; };
template <int KW>
__device__ __forceinline__ void fw2_load(const uint8_t *p, u32 (&v)[KW]) {
  if (KW == 4) {
    const u128_unaligned q = *reinterpret_cast<const u128_unaligned *>(p);
    v[0] = q.a;
    v[1] = q.b;
    v[KW - 2] = q.c;
    v[KW - 1] = q.d;
  } else {
    const unsigned long long q = reinterpret_cast<const u64_unaligned *>(p)->v;
    v[0] = (u32)q;
    v[KW - 1] = (u32)(q >> 32);
  }
}

// previous levels of the lines that ARRIVE at this lane's columns: the previous row shifted by dx columns (16-bit elements)
template <int K, int DX>
__device__ __forceinline__ void fw2_make_x(const u32 (&P)[K / 2], u32 (&X)[K / 2]) {
  constexpr int KW = K / 2;
  if (DX == 0) {
#pragma unroll
    for (int w = 0; w < KW; w++) X[w] = P[w];
  } else if (DX > 0) {  // column j gets the element of column j-1
    const u32 in = fw_shr1(P[KW - 1]);
#pragma unroll
    for (int w = KW - 1; w >= 1; w--) X[w] = __builtin_amdgcn_alignbyte(P[w], P[w - 1], 2);
    X[0] = __builtin_amdgcn_alignbyte(P[0], in, 2);
  } else {              // column j gets the element of column j+1
    const u32 in = fw_shl1(P[0]);
#pragma unroll
    for (int w = 0; w < KW - 1; w++) X[w] = __builtin_amdgcn_alignbyte(P[w + 1], P[w], 2);
    X[KW - 1] = __builtin_amdgcn_alignbyte(in, P[KW - 1], 2);
  }
}

// The NEXT volume's pack for this path as a side job of the walking waves (round 5; PackJob / PackWave in kernels_sweepfw.h
// have the reasons: the pack is HBM-bound, the walk is bound by its LDS atomics, so the pack's memory time disappears behind
// the walk and only its VALU instructions remain -- as a launch of its own the pack costs 0.20 ms of a 0.85 ms step at
// 512^3 x 64 levels, profiles/r05b_probe64.json).  A unit is 64 pieces of 16 voxels; a lane turns the 16 int32 levels + 16
// mask bytes of its piece into 16 16-bit elements (level*4: two 16-byte stores into J.levels16) and, when J.levels != NULL, 16
// plain level bytes (one 16-byte store: only plans whose x angle still takes the rows kernel of kernels_sweep.h read them).  Needs the linear layouts (pitch16 == 2 NX, pitch == NX: NX % 16 == 0)
// and 16-byte aligned arrays.  Same results as pack_levels16_kernel for every input (tests/test_gpu_fw2_pack.py).
struct PackWave16 {
  unsigned long long img, msk;   // this lane's next piece: image + 64 t, mask + 16 t (byte addresses)
  unsigned long long step16;     // pieces between two units of this wave
  unsigned e;                    // first voxel of that piece (16 t; volumes stay below 2^31 voxels)
  int units_left, last_lanes, tick, every, loaded, bad, lane;
  int4 q0, q1, q2, q3;
  uint4 m;
  template <typename T>
  static __device__ __forceinline__ void pin(T &x) { asm volatile("" : "+v"(x)); }
  __device__ __forceinline__ PackWave16(const PackJob &J, long long pw, long long W) {
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
    e = (unsigned)(16ull * t);
    step16 = (unsigned long long)(W * 64);
    tick = 0;
    every = J.every;
    loaded = 0;
    bad = 0;
    pin(img); pin(msk); pin(e); pin(step16); pin(units_left); pin(last_lanes); pin(tick); pin(every); pin(loaded); pin(bad);
  }
  __device__ __forceinline__ void load() {
    q0 = q1 = q2 = q3 = make_int4(0, 0, 0, 0);
    m = make_uint4(0, 0, 0, 0);
    if (units_left > 1 || lane < last_lanes) {
      const int4 *im4 = reinterpret_cast<const int4 *>((size_t)img);
#ifdef PRAD_PACK16_NT     // A/B build: the image and the mask stream past the caches the walk's re-reads live in
      typedef int v4i __attribute__((ext_vector_type(4)));
      const v4i *iv = reinterpret_cast<const v4i *>((size_t)img);
      const v4i mv = __builtin_nontemporal_load(reinterpret_cast<const v4i *>((size_t)msk));
      const v4i a0 = __builtin_nontemporal_load(iv), a1 = __builtin_nontemporal_load(iv + 1),
                a2 = __builtin_nontemporal_load(iv + 2), a3 = __builtin_nontemporal_load(iv + 3);
      m = make_uint4((u32)mv.x, (u32)mv.y, (u32)mv.z, (u32)mv.w);
      q0 = make_int4(a0.x, a0.y, a0.z, a0.w);
      q1 = make_int4(a1.x, a1.y, a1.z, a1.w);
      q2 = make_int4(a2.x, a2.y, a2.z, a2.w);
      q3 = make_int4(a3.x, a3.y, a3.z, a3.w);
      (void)im4;
#else
      m = *reinterpret_cast<const uint4 *>((size_t)msk);
      q0 = im4[0];
      q1 = im4[1];
      q2 = im4[2];
      q3 = im4[3];
#endif
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
      // Two voxels per dword (16-bit halves).  FAST PATH, decided once for the 16 voxels of the piece: every voxel lies inside
      // the mask and every level is regular (1..Ng, no bits beyond the low half).  With q = h - 0x00010001 and
      // r = q + (0x8000 - Ng) * 0x00010001, bit 15 of some half of h | q | r is set iff some half of h is 0 or > Ng (the lowest
      // irregular half sees no borrow / carry from below; Ng <= 255 on this path).  SECOND FAST PATH: no voxel inside the mask.
      const u32 K80 = 0x80808080u, K7F = 0x7f7f7f7fu, H1 = 0x00010001u, H80 = 0x80008000u;
      const u32 radd = (0x8000u - (u32)J.Ng) * H1;
      u32 h[8], badbits = 0, wideall = 0, nzall = K80, nzany = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const u32 l0 = (u32)lv[2 * i], l1 = (u32)lv[2 * i + 1];
        wideall |= l0 | l1;
        h[i] = __builtin_amdgcn_perm(l1, l0, 0x05040100u);     // low halves of the two levels
        const u32 q = h[i] - H1;
        badbits |= h[i] | q | (q + radd);
      }
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const u32 nz = ((mw[w] & K7F) + K7F) | mw[w];           // bit 7 of every non-zero mask byte
        nzall &= nz;
        nzany |= nz;
      }
      u32 o16[8], o8[4];
      const bool want8 = J.levels != nullptr;      // (wave-uniform: the plain byte copy, only for plans that still read one)
      if ((((~nzall) & K80) | (badbits & H80) | (wideall & 0xffff0000u)) == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) o16[i] = h[i] << PRAD_FUSED_SHIFT;
        if (want8) {
#pragma unroll
          for (int w = 0; w < 4; w++) o8[w] = __builtin_amdgcn_perm(h[2 * w + 1], h[2 * w], 0x06040200u);
        }
      } else if ((nzany & K80) == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) o16[i] = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) o8[w] = 0;
      } else {
        // the exact per-voxel form (pieces that straddle the ROI's surface, irregular levels): same results in every case
#pragma unroll
        for (int i = 0; i < 8; i++) o16[i] = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) o8[w] = 0;
#pragma unroll
        for (int b = 0; b < 16; b++) {
          const bool in = (mw[b >> 2] >> (8 * (b & 3))) & 0xffu;
          const int l = lv[b];
          const bool regular = in && l >= 1 && l <= J.Ng;
          bad |= in && !regular;
          const u32 el = regular ? (u32)l : 0u;                 // (an irregular level packs as 0: never a table index)
          o16[b >> 1] |= (el << PRAD_FUSED_SHIFT) << (16 * (b & 1));
          o8[b >> 2] |= el << (8 * (b & 3));
        }
      }
      uint4 *d16 = reinterpret_cast<uint4 *>(J.levels16 + 2ull * e);
      d16[0] = make_uint4(o16[0], o16[1], o16[2], o16[3]);
      d16[1] = make_uint4(o16[4], o16[5], o16[6], o16[7]);
      if (want8) *reinterpret_cast<uint4 *>(J.levels + e) = make_uint4(o8[0], o8[1], o8[2], o8[3]);
    }
    img += 64ull * step16;
    msk += 16ull * step16;
    e += (unsigned)(16ull * step16);
    units_left--;
    pin(img); pin(msk); pin(e); pin(units_left); pin(bad);
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

#define FW2_EL(W, j) ((int)__builtin_amdgcn_ubfe((W)[(j) >> 1], 16 * ((j) & 1), 16))

// SKIP1: runs of length 1 are not recorded by the walk.  Every ROI voxel lies on exactly one line of an angle, so
// sum_len len * GLRLM_a[g][len] = N_g (the voxels of level g) for EVERY angle a; the x angle's kernel records all of its runs,
// and the finalize step sets GLRLM_a[g][1] = N_g - sum_{len >= 2} len * GLRLM_a[g][len] (exact integers).
template <bool LONG, int K, int DX, bool HASPAD, bool SKIP1, bool PACK>
struct Fw2Wave {
  static constexpr int KW = K / 2;
  static constexpr int U = PRAD_FW_U;
  const Fw2Tab &T;
  int dummy, lane, edge_lane;
  int k1v;         // K1 + byte offset of this lane's B copy
  int k14;         // k1v + 4: cursor - row of a run of length 1
  int kdelta;      // k1v minus the k1v of the lane a drifting line comes from (lane 0 / 63: the line comes from outside, k1v)
  u32 one;
  static constexpr bool haspad = HASPAD;
  u32 cmask[KW];   // halves of this lane's window columns that lie inside the row
  u32 calm[KW];    // halves of window columns no line can be open on
  int lp[K], lq[K];   // A row / B cursor of the line that arrives at column j at the next step
  unsigned long long ev[K];   // SKIP1, inside a plain group: lanes whose line (register j) had an event at its previous step
  u32 P[KW];       // levels of the previous row (this lane's columns)

  __device__ __forceinline__ Fw2Wave(const Fw2Tab &T_, int NX) : T(T_) {
    lane = threadIdx.x & 63;
    dummy = 4 * lane;            // row 0 is scratch and at least 64 words long (Ng >= 40, RS2 >= 24 or Nr)
    edge_lane = haspad ? -1 : (DX > 0 ? 63 : (DX < 0 ? 0 : -1));
    one = 1;
    asm volatile("" : "+v"(one));
    k1v = T.K1 + (lane % T.C) * T.lenlim;
    {
      const int src = lane - (DX > 0 ? 1 : -1);
      kdelta = (DX == 0 || src < 0 || src > 63) ? k1v : ((lane % T.C) - (src % T.C)) * T.lenlim;
    }
    k14 = k1v + 4;
    asm volatile("" : "+v"(k1v), "+v"(kdelta), "+v"(k14));
    const int col0 = first_col(NX);
#pragma unroll
    for (int w = 0; w < KW; w++) {
      u32 m = 0, q = 0;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int col = col0 + 2 * w + b;
        const bool outside = col < 0 || col >= NX, feeder_outside = col - DX < 0 || col - DX >= NX;
        if (!outside) m |= 0xffffu << (16 * b);
        if (outside && feeder_outside) q |= 0xffffu << (16 * b);
      }
      cmask[w] = m;
      calm[w] = q;
    }
  }
  __device__ __forceinline__ int first_col(int NX) const { return (threadIdx.x & 63) * K - (DX < 0 ? 64 * K - NX : 0); }

  __device__ __forceinline__ void reset_lines() {
#pragma unroll
    for (int j = 0; j < K; j++) {
      lp[j] = 0;
      lq[j] = k1v;
    }
  }
  __device__ __forceinline__ void load_row(const uint8_t *p, u32 (&v)[KW]) const {
    fw2_load<KW>(p, v);
    if (haspad) {
#pragma unroll
      for (int w = 0; w < KW; w++) v[w] &= cmask[w];
    }
  }
  // cross-lane move of the one line that changes lane, after the line that leaves the row was closed (its run is recorded:
  // B[prev][len]; there is no pair across the row's edge)
  template <bool PLAIN>
  __device__ __forceinline__ void rotate_reg(int &rp, int &rq, int xlevel, unsigned long long &rev) {
    if (PLAIN && SKIP1 && haspad) (void)rev;
    if (!haspad) {
      if (PLAIN) {
        const unsigned long long em = DX > 0 ? 0x8000000000000000ull : 1ull;   // lane 63 / lane 0
        if (SKIP1) {
#ifdef PRAD_FW2_SKIP_BY_STATE
          int t;
          asm volatile("v_sub_u32 %[t], %[r], %[p]\n\tv_cmp_ne_u32 vcc, %[t], %[K14]\n\ts_and_b64 exec, vcc, %[m]\n\t"
                       "ds_add_u32 %[r], %[one]\n\ts_mov_b64 exec, -1"
                       : [t] "=&v"(t) : [m] "s"(em), [r] "v"(rq), [p] "v"(rp), [one] "v"(one), [K14] "v"(k14) : "vcc", "scc", "memory");
#else
          asm volatile("s_andn2_b64 exec, %[m], %[e]\n\tds_add_u32 %[r], %[one]\n\ts_mov_b64 exec, -1"
                       :: [m] "s"(em), [e] "s"(rev), [r] "v"(rq), [one] "v"(one) : "scc", "memory");
#endif
        } else {
          asm volatile("s_mov_b64 exec, %[m]\n\tds_add_u32 %[r], %[one]\n\ts_mov_b64 exec, -1" ::[m] "s"(em), [r] "v"(rq), [one] "v"(one) : "memory");
        }
      } else if (lane == edge_lane) {
        fw2_checked<LONG, SKIP1>(T, k1v, dummy, rp, rq, xlevel, 0, false);
      }
    }
    rp = (int)(DX > 0 ? fw_shr1((u32)rp) : fw_shl1((u32)rp));
    rq = (int)(DX > 0 ? fw_shr1((u32)rq) : fw_shl1((u32)rq)) + kdelta;   // (the cursor moves to this lane's copy of the slots;
                                                                           //  the line that enters from outside: level 0, length 0)
    if (PLAIN && SKIP1) rev = DX > 0 ? rev << 1 : rev >> 1;                // (its event bit travels with it; the entering line: none)
  }
  __device__ __forceinline__ void single_step(const u32 (&C)[KW], bool tail) {
    u32 X[KW];
    fw2_make_x<K, DX>(P, X);
#pragma unroll
    for (int j = 0; j < K; j++) fw2_checked<LONG, SKIP1>(T, k1v, dummy, lp[j], lq[j], FW2_EL(X, j), FW2_EL(C, j), tail);
    if (DX > 0) {
      rotate_reg<false>(lp[K - 1], lq[K - 1], FW2_EL(C, K - 1), ev[0]);
      const int ip = lp[K - 1], iq = lq[K - 1];
#pragma unroll
      for (int j = K - 1; j >= 1; j--) {
        lp[j] = lp[j - 1];
        lq[j] = lq[j - 1];
      }
      lp[0] = ip;
      lq[0] = iq;
    } else if (DX < 0) {
      rotate_reg<false>(lp[0], lq[0], FW2_EL(C, 0), ev[0]);
      const int ip = lp[0], iq = lq[0];
#pragma unroll
      for (int j = 0; j < K - 1; j++) {
        lp[j] = lp[j + 1];
        lq[j] = lq[j + 1];
      }
      lp[K - 1] = ip;
      lq[K - 1] = iq;
    }
#pragma unroll
    for (int w = 0; w < KW; w++) P[w] = C[w];
  }
  // U plain steps with renamed registers
  __device__ __forceinline__ void plain_group(const u32 (&v)[U][KW]) {
    if (SKIP1) {   // the event masks from the lines' states: a run has length 1 exactly behind an event
#pragma unroll
      for (int j = 0; j < K; j++) ev[j] = __ballot(lq[j] - lp[j] == k14);
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      u32 X[KW];
      fw2_make_x<K, DX>(P, X);
#pragma unroll
      for (int w = 0; w < KW; w++) {
        const int r0 = (((2 * w + 0 - k * DX) % K) + K) % K, r1 = (((2 * w + 1 - k * DX) % K) + K) % K;
        fw2_plain_word<SKIP1>(T, k1v, k14, one, lp[r0], lq[r0], lp[r1], lq[r1], v[k][w], X[w], ev[r0], ev[r1]);
      }
      if (DX > 0) {
        const int r = (((K - 1 - k) % K) + K) % K;
        rotate_reg<true>(lp[r], lq[r], FW2_EL(v[k], K - 1), ev[r]);
      }
      if (DX < 0) {
        const int r = k % K;
        rotate_reg<true>(lp[r], lq[r], FW2_EL(v[k], 0), ev[r]);
      }
#pragma unroll
      for (int w = 0; w < KW; w++) P[w] = v[k][w];
    }
  }
  // 4 len of the longest open run (level-0 lines count with their age): a plain walk of n steps is safe while
  // margin + 4 n <= lenlim
  __device__ __forceinline__ unsigned margin() {
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < K; j++) m = max(m, (unsigned)(lq[j] - lp[j] - k1v));
    return m;
  }
  __device__ __forceinline__ void calm_padding() {
    if (!haspad) return;
#pragma unroll
    for (int j = 0; j < K; j++)
      if ((calm[j >> 1] >> (16 * (j & 1))) & 0xffffu) {
        lp[j] = 0;
        lq[j] = k1v;
      }
  }
  // lines inside a stretch of voxels outside the ROI (and dead lines): the stretch is no run, its "length" restarts
  __device__ __forceinline__ void calm_level0() {
#pragma unroll
    for (int j = 0; j < K; j++)
      if (lp[j] < T.S) lq[j] = k1v;
  }
  __device__ __forceinline__ bool any_alive() {
    u32 X[KW];
    fw2_make_x<K, DX>(P, X);
    bool a = false;
#pragma unroll
    for (int j = 0; j < K; j++) a = a || (lp[j] >= T.S && FW2_EL(X, j) != 0);
    return __ballot(a) != 0;
  }
  __device__ __forceinline__ void run(const FwDesc &D, int NX, int pitch, long long nrows, const uint8_t *__restrict__ L,
                                      int *work, int bx, int nblocks, bool xcd, PackWave16 &pk, const PackJob &pj) {
    const int NM = D.NM, NU = D.NU, du = D.du;
    const long long delta = D.sM + (long long)du * D.sU;
    const uint8_t *lpb = L + 2 * (long long)first_col(NX);
    const int nwaves = (int)(nblocks * (blockDim.x >> 6));
    const int wid = __builtin_amdgcn_readfirstlane((int)(bx * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    // XCD-aware hand-out (FwWave::run in kernels_sweepfw.h has the description): at 512^3 the 16-bit level volume (268 MB,
    // read by 12 roles) does not fit the Infinity Cache -- here the re-reads cost TIME, not only fabric bytes
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
          int grabbed = 0;
          if (lane == 0) grabbed = atomicAdd(work, 1);
          chunk = nwaves + __builtin_amdgcn_readfirstlane(grabbed);
        }
        if (chunk >= D.chunks) break;
        piece = chunk / NU;
        u0 = chunk - piece * NU;
      } else {
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
      const bool starts = t0 == 0 || (du > 0 && row == 0) || (du < 0 && row == NU - 1);
      int safe = 0;
      reset_lines();               // (a piece that does not begin at a line start: every line DEAD = level 0)
      if (starts) {
#pragma unroll
        for (int w = 0; w < KW; w++) P[w] = 0;
      } else {
        load_row(lpb + (off - delta), P);
      }
      int t = t0;
      bool wrap = false, tail = false;
      for (;;) {
        bool closing = false, finish = false;
        if (!tail && t >= t1) {
          if (t1 == NM || wrap) closing = finish = true;
          else tail = true;
        }
        if (tail && !closing) {
          if (t == NM || wrap) closing = finish = true;
          else if (!any_alive()) break;
        }
        if (!tail && !closing && wrap) closing = true;
        if (!tail && !closing) {
          const int room = du > 0 ? NU - row : (du < 0 ? row + 1 : (1 << 30));
          const bool grp = t + U <= t1 && room >= U;
          if (grp && safe == 0) {
            calm_padding();
            if (!LONG) {
              safe = 2;            // every run length (and every age) has its slot
            } else {
              calm_level0();       // (ages restart: only real runs decide)
              const unsigned m = margin();
              if (__ballot(m + 4 * 2 * U > (unsigned)T.lenlim) == 0) safe = 2;
              else if (__ballot(m + 4 * U > (unsigned)T.lenlim) == 0) safe = 1;
            }
          }
          if (grp && safe > 0) {
            u32 v[U][KW];
            const uint8_t *p = lpb + off;
#pragma unroll
            for (int k = 0; k < U; k++) {
              load_row(p, v[k]);
              p += delta;
            }
            if (PACK) pk.begin();            // (the next volume's pack rides along: loads out, ...
            if (safe == 1) calm_padding();
            plain_group(v);
            if (PACK) pk.finish(pj);         //  ... elements stored behind the group's VALU / LDS work)
            safe--;
            t += U;
            row += U * du;
            off += (long long)U * delta;
            if (du != 0 && (row < 0 || row >= NU)) wrap = true;
            continue;
          }
        }
        u32 c[KW];
        if (closing) {
#pragma unroll
          for (int w = 0; w < KW; w++) c[w] = 0;
        } else {
          load_row(lpb + off, c);
        }
        single_step(c, tail);
        safe = 0;
        if (closing) {
          if (finish) break;
          reset_lines();
          row -= du * NU;
          off -= (long long)du * NU * D.sU;
          wrap = false;
          continue;
        }
        t++;
        row += du;
        off += delta;
        if (du != 0 && (row < 0 || row >= NU)) wrap = true;
      }
    }
  }
};

// table -> global accumulators (angle-major u32, the layout of kernels_sweep.h)
__device__ __forceinline__ void fw2_flush(const u32 *lds, const Fw2Tab &T, int slot, u32 *__restrict__ glcm_acc,
                                          u32 *__restrict__ glrlm_acc) {
  __syncthreads();
  const int Ng = T.Ng;
  u32 *gd = glcm_acc + (size_t)slot * Ng * Ng;
  for (int i = threadIdx.x; i < Ng * Ng; i += blockDim.x) {
    const int p = i / Ng, c = i - p * Ng;
    if (p == c) continue;  // the diagonal comes from the GLRLM in finalize
    const u32 v = lds[(size_t)(p + 1) * T.S4 + c + 1];
    if (v) atomicAdd(gd + i, v);
  }
  u32 *rd = glrlm_acc + (size_t)slot * Ng * T.Nr;
  const int nl = min(T.RS2, T.Nr);
  for (int i = threadIdx.x; i < Ng * nl; i += blockDim.x) {
    const int p = i / nl, l = i - p * nl;
    u32 v = 0;
    for (int cp = 0; cp < T.C; cp++) v += lds[(size_t)(p + 1) * T.S4 + Ng + 1 + cp * T.RS2 + l];
    if (v) atomicAdd(rd + (size_t)p * T.Nr + l, v);
  }
}

// PACK: the pack of the NEXT volume rides in this launch (PackWave16).  A volume whose pack found irregular levels (flags[0])
// is skipped -- the generic kernels redo that call -- but the side job still runs.
template <bool LONG, int K, bool HASPAD, bool SKIP1, bool PACK>
__global__ void __launch_bounds__(1024) sweep_fw2_kernel(FwSet set, PackJob pj, const uint8_t *__restrict__ L, int Ng, int Nr, int RS2,
                                                         int C, u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc,
                                                         int *__restrict__ work, int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  const int wpb = (int)(blockDim.x >> 6);
  const long long pw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
  PackWave16 pk(pj, PACK ? pw : -1, (long long)gridDim.x * wpb);
  bool walk = flags[0] == 0;   // (else: irregular levels, the generic path will redo this call)
  if (walk && (unsigned)(size_t)((lds_u32 *)lds) != 0u) {  // table offsets are used as LDS addresses
    if (threadIdx.x == 0) atomicExch(flags + 2, 1);
    walk = false;
  }
  if (walk) {
    int role = 0;
    while (role + 1 < set.count && (int)blockIdx.x >= set.first_block[role + 1]) role++;
    const int bx = (int)blockIdx.x - set.first_block[role], nblocks = set.first_block[role + 1] - set.first_block[role];
    const int words = (int)fw2_table_words(Ng, RS2, C);
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const FwDesc &D = set.d[role];
    Fw2Tab T;
    T.init(Ng, RS2, C, Nr, glrlm_acc + (size_t)D.slot * Ng * Nr);
    int *wk = work + PRAD_FW_WORK_STRIDE * PRAD_FW_DOMAINS * role;
    if (D.dx == 0) {
      Fw2Wave<LONG, K, 0, HASPAD, SKIP1, PACK> w(T, set.NX);
      w.run(D, set.NX, set.pitch, set.nrows, L, wk, bx, nblocks, set.xcd != 0, pk, pj);
    } else if (D.dx > 0) {
      Fw2Wave<LONG, K, 1, HASPAD, SKIP1, PACK> w(T, set.NX);
      w.run(D, set.NX, set.pitch, set.nrows, L, wk, bx, nblocks, set.xcd != 0, pk, pj);
    } else {
      Fw2Wave<LONG, K, -1, HASPAD, SKIP1, PACK> w(T, set.NX);
      w.run(D, set.NX, set.pitch, set.nrows, L, wk, bx, nblocks, set.xcd != 0, pk, pj);
    }
    fw2_flush(lds, T, D.slot, glcm_acc, glrlm_acc);
  }
  if (PACK) pk.drain(pj);
}

// ---------------------------------------------------------------------------------------------------------------
// The angle along x on the 16-bit level volume with the same exec-masked two-table step (round 5; until then the rows kernel
// of kernels_sweep.h walked an 8-bit copy with the branch-free Walker: ~12 VALU and two unconditional ds_adds per voxel,
// 0.11 / 0.17 ms per 512^3 volume of iid / smooth levels against 0.06 / 0.08 for the fused-table rows kernel at 32 levels).
// A wave owns 64 rows (8 rows apart on large volumes, see fw_rows_role), stages 64 x 32-element tiles through LDS (16 B per
// lane coalesced in, one row per lane out) and every lane walks its own row, two voxels (one staged dword) per asm block.
// Table layout, state (p = A row, q = B cursor) and flush as in the line walk; this angle records ALL its runs (the SKIP1
// restore of the line angles reads N_g off it).
// ---------------------------------------------------------------------------------------------------------------
#define PRAD_ROW16_EL 32      // elements of a row per staged tile (a tile = 64 rows x 32 elements: 5 KB per wave, so that 16 waves
                              // and their tiles fit next to a 64-level table: the walk is a serial chain per lane, it needs the waves)
#define PRAD_ROW16_PITCH 80    // bytes per staged row: 64 data + 16 pad => conflict-free ds_read_b128 per lane
__device__ __forceinline__ void fw2_row_word(const Fw2Tab &T, int k1, u32 one, int &p, int &q, u32 c, u32 x) {
  int t;
#define PRAD_FW2_RCOL(J)                                                                                                 \
  "v_cmpx_ne_u32_sdwa vcc, %[c], %[x] src0_sel:WORD_" #J " src1_sel:WORD_" #J "\n\t"                                     \
  "v_add_u32_sdwa %[t], %[p], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #J "\n\t"            \
  "ds_add_u32 %[t], %[one]\n\t"                                                                                          \
  "ds_add_u32 %[q], %[one]\n\t"                                                                                          \
  "v_mul_u32_u24_sdwa %[p], %[S4], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #J "\n\t"      \
  "v_add_u32 %[q], %[K1], %[p]\n\t"                                                                                      \
  "s_mov_b64 exec, -1\n\t"                                                                                               \
  "v_add_u32 %[q], 4, %[q]\n\t"
  asm volatile(PRAD_FW2_RCOL(0) PRAD_FW2_RCOL(1)
               : [p] "+v"(p), [q] "+v"(q), [t] "=&v"(t)
               : [c] "v"(c), [x] "v"(x), [one] "v"(one), [S4] "s"(T.S4), [K1] "v"(k1)
               : "vcc", "memory");
#undef PRAD_FW2_RCOL
}

template <bool LONG>
__global__ void __launch_bounds__(1024) sweep_fw2_rows_kernel(const uint8_t *__restrict__ L16, long long nrows, int NX, int pitch16,
                                                             int slot, int Ng, int Nr, int RS2, int C, u32 *__restrict__ glcm_acc,
                                                             u32 *__restrict__ glrlm_acc, int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;
  if ((unsigned)(size_t)((lds_u32 *)lds) != 0u) {   // table offsets are used as LDS addresses
    if (threadIdx.x == 0) atomicExch(flags + 2, 1);
    return;
  }
  const int words = (int)fw2_table_words(Ng, RS2, C);
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  Fw2Tab T;
  T.init(Ng, RS2, C, Nr, glrlm_acc + (size_t)slot * Ng * Nr);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int dummy = 4 * lane;      // row 0 is scratch and at least 64 words long
  u32 one = 1;
  int k1v = T.K1 + (lane % T.C) * T.lenlim;
  asm volatile("" : "+v"(one), "+v"(k1v));
  uint8_t *tile = reinterpret_cast<uint8_t *>(lds) + ((4 * (size_t)words + 15) & ~(size_t)15) + (size_t)wave * 64 * PRAD_ROW16_PITCH;
  const int RSTEP = nrows >= 4096 ? 8 : 1;
  const long long ngroups = RSTEP == 8 ? ((nrows + 511) / 512) * 8 : (nrows + 63) / 64;
  const long long nwaves = (long long)gridDim.x * wpb;
  for (long long grp = (long long)blockIdx.x * wpb + wave; grp < ngroups; grp += nwaves) {
    const long long r0 = RSTEP == 8 ? (grp >> 3) * 512 + (grp & 7) : grp * 64;     // row of tile slot i: r0 + RSTEP * i
    int p = 0, q = k1v;   // level 0, no voxel seen
    u32 pw = 0;           // previous staged dword (its high half is the previous voxel)
    // lane -> (row j*16 + lane/4, 16-byte piece lane%4) of a tile; the pieces of the NEXT tile are loaded into registers before
    // the current one is walked (the global-load latency of a wave's tiles was a third of this kernel's time)
    uint4 nx4[4];
    auto fetch = [&](int xc) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int rr = j * 16 + (lane >> 2);
        const int cx = xc + (lane & 3) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + (long long)RSTEP * rr < nrows && cx < NX) {
          v = *reinterpret_cast<const uint4 *>(L16 + (r0 + (long long)RSTEP * rr) * pitch16 + 2 * cx);
          const int valid = NX - cx;             // elements of this piece that belong to the row (the rest is the pad)
          if (valid < 8) {
            u32 *vw = reinterpret_cast<u32 *>(&v);
#pragma unroll
            for (int wd = 0; wd < 4; wd++) {
              const int keep = valid - 2 * wd;
              vw[wd] = keep >= 2 ? vw[wd] : (keep <= 0 ? 0u : (vw[wd] & 0xffffu));
            }
          }
        }
        nx4[j] = v;
      }
    };
    fetch(0);
    for (int xc = 0; xc < NX; xc += PRAD_ROW16_EL) {
      // ---- stage rows r0 + RSTEP i (i = 0..63), elements xc..xc+31 ----
#pragma unroll
      for (int j = 0; j < 4; j++) *reinterpret_cast<uint4 *>(tile + (j * 16 + (lane >> 2)) * PRAD_ROW16_PITCH + (lane & 3) * 16) = nx4[j];
      __builtin_amdgcn_wave_barrier();
      if (xc + PRAD_ROW16_EL < NX) fetch(xc + PRAD_ROW16_EL);
      // ---- each lane walks its own row (rows >= nrows and elements >= NX were staged as zeros) ----
      const uint4 *row = reinterpret_cast<const uint4 *>(tile + lane * PRAD_ROW16_PITCH);
#pragma unroll 2
      for (int qd = 0; qd < PRAD_ROW16_EL / 8; qd++) {
        const uint4 d = row[qd];
        const u32 wds[4] = {d.x, d.y, d.z, d.w};
        // a stretch outside the ROI is no run: its "length" restarts (the cursor must stay inside row 0's slots)
        q = select_i32(p < T.S, k1v, q);
        // 8 plain steps are safe while 4 len + 4 * 8 stays within the length slots (an event records at most len + 7)
        if (LONG && __ballot((unsigned)(q - p - k1v) + 32u > (unsigned)T.lenlim) != 0) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const u32 c = wds[k], x = __builtin_amdgcn_alignbyte(c, pw, 2);
#pragma unroll
            for (int b = 0; b < 2; b++)
              fw2_checked<LONG, false>(T, k1v, dummy, p, q, (int)((x >> (16 * b)) & 0xffffu), (int)((c >> (16 * b)) & 0xffffu), false);
            pw = c;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            fw2_row_word(T, k1v, one, p, q, wds[k], __builtin_amdgcn_alignbyte(wds[k], pw, 2));
            pw = wds[k];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    fw2_checked<LONG, false>(T, k1v, dummy, p, q, (int)(pw >> 16), 0, false);   // the row ends: close its open run
  }
  fw2_flush(lds, T, slot, glcm_acc, glrlm_acc);
}

// pack for this path: 16-bit level*4 elements (rows of pitch16 BYTES) and, when L8 != NULL (plans whose x angle still takes the
// rows kernel of kernels_sweep.h), plain 8-bit levels
__global__ void __launch_bounds__(256) pack_levels16_kernel(const int *__restrict__ image, const uint8_t *__restrict__ mask,
                                                            long long n, int NX, int pitch16, int pitch8, int Ng,
                                                            uint8_t *__restrict__ L16, uint8_t *__restrict__ L8,
                                                            int *__restrict__ flags) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  int bad = 0;
  // 4 voxels per lane and step when rows are whole 4-voxel pieces; else one by one
  if ((NX & 3) == 0) {
    const long long n4 = n >> 2;
    const int upr = NX >> 2;
    for (long long t = tid; t < n4; t += nthreads) {
      const int4 q = reinterpret_cast<const int4 *>(image)[t];
      const u32 m = reinterpret_cast<const u32 *>(mask)[t];
      const int lv[4] = {q.x, q.y, q.z, q.w};
      u32 e[4];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const bool in = (m >> (8 * b)) & 0xffu;
        const bool regular = in && lv[b] >= 1 && lv[b] <= Ng;
        bad |= in && !regular;
        e[b] = regular ? (u32)lv[b] : 0u;
      }
      const long long row = t / upr;
      const int x = (int)(t - row * upr) << 2;
      *reinterpret_cast<uint2 *>(L16 + row * pitch16 + 2 * x) =
          make_uint2((e[0] << PRAD_FUSED_SHIFT) | (e[1] << (16 + PRAD_FUSED_SHIFT)), (e[2] << PRAD_FUSED_SHIFT) | (e[3] << (16 + PRAD_FUSED_SHIFT)));
      if (L8) *reinterpret_cast<u32 *>(L8 + row * pitch8 + x) = e[0] | (e[1] << 8) | (e[2] << 16) | (e[3] << 24);
    }
  } else {
    for (long long i = tid; i < n; i += nthreads) {
      const bool in = mask[i] != 0;
      const int l = image[i];
      const bool regular = in && l >= 1 && l <= Ng;
      bad |= in && !regular;
      const long long row = i / NX;
      const int x = (int)(i - row * NX);
      *reinterpret_cast<unsigned short *>(L16 + row * pitch16 + 2 * x) = (unsigned short)(regular ? (l << PRAD_FUSED_SHIFT) : 0);
      if (L8) L8[row * pitch8 + x] = (uint8_t)(regular ? l : 0);
    }
  }
  if (bad) flags[0] = 1;
}

}  // namespace prad
