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

struct FwDesc {
  int slot;          // index of the angle in the caller's list (output column)
  int NM, NU;        // extents of the march and row dimensions
  int du, dx;        // motion per march step in the row dim / along the contiguous axis
  long long sM, sU;  // byte strides of the march and row dimensions in the packed volume
  int CL, pieces;    // march steps per piece (multiple of 8), pieces per walk
  int chunks;        // NU * pieces
};
struct FwSet {
  int count;
  int NX;
  FwDesc d[PRAD_MAX_SWEEP];
};

#define PRAD_FW_DEAD (-(1 << 30))   // run state of a line that must ignore its next event (stays negative for any walk)
#define PRAD_FW_U 8

struct FwTab {   // wave-uniform constants of the fused table (layout: hist_layout(true, true, true, Ng, RS))
  u32 *rl_long;
  int Nr, P4, Q, lenmax, gB, RL4, RS, RL, dummy0b;
  unsigned Qinv;
  __device__ __forceinline__ void init(const HistLayout &h, int Nr_, u32 *rl_long_) {
    rl_long = rl_long_;
    Nr = Nr_;
    Q = 4 * (h.Ng + 1);
    P4 = (h.RS + 1) * (h.Ng + 1);
    lenmax = h.RS * Q;
    RS = h.RS;
    RL = h.RL;
    RL4 = 4 * h.RL;
    gB = 4 * h.g0 - RL4 - 4 * h.RS;
    Qinv = (unsigned)((0x100000000ull + (unsigned)Q - 1) / (unsigned)Q);
    dummy0b = 4 * h.dummy0;
  }
};

// a run of level lv (!= 0) longer than RS just ended: record its length (LDS table G, or a wave-aggregated L2 atomic)
__device__ __forceinline__ void fw_long_event(const FwTab &T, int lv, int lb) {
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
//   pl  level*P + (len-1)*Q of the open run (unclamped), or negative = dead;  x, c = level*4 of the previous / current voxel
template <bool LONG, bool TAIL>
__device__ __forceinline__ void fw_checked(const FwTab &T, int dummy, int &pl, int x, int c) {
  const bool chg = c != x;
  const bool ev = chg && pl >= 0;
  int bin = pl;
  if (LONG) {
    const int lb = pl - __mul24(x, T.P4);
    bin = pl - lb + min(lb, T.lenmax);
    if (ev && x != 0 && lb >= T.lenmax) fw_long_event(T, x >> PRAD_FUSED_SHIFT, lb);
  }
  lds_bump(ev ? bin + c : dummy);
  pl = select_i32(chg, TAIL ? PRAD_FW_DEAD : __mul24(c, T.P4), pl + T.Q);
}

// the branch-free plain step of four lines whose levels are the byte lanes of c (current) and x (previous):
// 6 VALU (SDWA byte operands) + 1 ds_add per line.  Only valid when no line is dead and no run can reach RS.
__device__ __forceinline__ void fw_plain_word(const FwTab &T, int dummy, int &p0, int &p1, int &p2, int &p3, u32 c, u32 x) {
  int *p[4] = {&p0, &p1, &p2, &p3};
  int addr[4], fresh[4], grown[4];
  bool chg[4];
#pragma unroll
  for (int j = 0; j < 4; j++) chg[j] = __builtin_amdgcn_ubfe(c, 8 * j, 8) != __builtin_amdgcn_ubfe(x, 8 * j, 8);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    addr[j] = *p[j] + (int)__builtin_amdgcn_ubfe(c, 8 * j, 8);
    fresh[j] = (int)__umul24(__builtin_amdgcn_ubfe(c, 8 * j, 8), (unsigned)T.P4);
    grown[j] = *p[j] + T.Q;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) lds_bump(chg[j] ? addr[j] : dummy);
#pragma unroll
  for (int j = 0; j < 4; j++) *p[j] = select_i32(chg[j], fresh[j], grown[j]);
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
template <int KW>
__device__ __forceinline__ void fw_load(const uint8_t *p, u32 (&v)[KW]) {
#ifdef PRAD_DBG_NOLOAD  // ablation build: synthetic levels, no memory traffic
#pragma unroll
  for (int w = 0; w < KW; w++) {
    const u32 x = ((u32)(size_t)p + 977u * w) * 2654435761u;
    v[w] = (((x >> 7) & 0x1f1f1f1fu) + 0x01010101u) << PRAD_FUSED_SHIFT;
  }
#else
  if (KW == 2) {
    const unsigned long long q = reinterpret_cast<const u64_unaligned *>(p)->v;
    v[0] = (u32)q;
    v[KW - 1] = (u32)(q >> 32);
  } else {
    v[0] = reinterpret_cast<const u32_unaligned *>(p)->v;
  }
#endif
}

#define FW_BYTE(W, j) ((int)__builtin_amdgcn_ubfe((W)[(j) >> 2], 8 * ((j) & 3), 8))

template <bool LONG, int K, int DX>
struct FwWave {
  static constexpr int KW = K / 4;
  static constexpr int U = PRAD_FW_U;
  const FwTab &T;
  int dummy, lane, edge_lane;
  bool haspad;
  u32 cmask[KW];   // byte lanes of this lane's window columns that lie inside the row
  u32 calm[KW];    // byte lanes of window columns no line can be open on (beyond the row, not next to its exit side)
  int pl[K];       // run state of the line that arrives at column j at the next step
  u32 P[KW];       // levels of the previous row (this lane's columns)

  __device__ __forceinline__ FwWave(const FwTab &T_, int NX) : T(T_) {
    lane = threadIdx.x & 63;
    dummy = T.dummy0b + 4 * lane;
    haspad = NX != 64 * K;
    edge_lane = haspad ? -1 : (DX > 0 ? 63 : (DX < 0 ? 0 : -1));
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
        lds_bump(lane == edge_lane ? r : dummy);
      } else if (lane == edge_lane) {
        fw_checked<LONG, false>(T, dummy, r, xlevel, 0);
      }
    }
    r = (int)(DX > 0 ? fw_shr1((u32)r) : fw_shl1((u32)r));
  }
  // one march step on the canonical register assignment (register j = column j), all cases handled
  template <bool TAIL>
  __device__ __forceinline__ void single_step(const u32 (&C)[KW]) {
    u32 X[KW];
    fw_make_x<K, DX>(P, X);
#pragma unroll
    for (int j = 0; j < K; j++) fw_checked<LONG, TAIL>(T, dummy, pl[j], FW_BYTE(X, j), FW_BYTE(C, j));
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
  __device__ __forceinline__ void plain_group(const u32 (&v)[U][KW]) {
#pragma unroll
    for (int k = 0; k < U; k++) {
      u32 X[KW];
      fw_make_x<K, DX>(P, X);
#pragma unroll
      for (int w = 0; w < KW; w++) {
        constexpr int dummy_ce = 0;
        (void)dummy_ce;
        const int r0 = (((4 * w + 0 - k * DX) % K) + K) % K, r1 = (((4 * w + 1 - k * DX) % K) + K) % K;
        const int r2 = (((4 * w + 2 - k * DX) % K) + K) % K, r3 = (((4 * w + 3 - k * DX) % K) + K) % K;
        fw_plain_word(T, dummy, pl[r0], pl[r1], pl[r2], pl[r3], v[k][w], X[w]);
      }
      if (DX > 0) rotate_reg<true>(pl[(((K - 1 - k) % K) + K) % K], 0);
      if (DX < 0) rotate_reg<true>(pl[k % K], 0);
#pragma unroll
      for (int w = 0; w < KW; w++) P[w] = v[k][w];
    }
  }
  // true if a plain group of U steps is not safe for some line of this lane
  __device__ __forceinline__ bool risky() {
    u32 X[KW];
    fw_make_x<K, DX>(P, X);
    bool r = false;
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (LONG) r = r || (unsigned)(pl[j] - __mul24(FW_BYTE(X, j), T.P4) + U * T.Q) > (unsigned)T.lenmax;
      else r = r || pl[j] < 0;
    }
    return r;
  }
  // lines that never see a voxel (window columns beyond the row) must not look like runs about to outgrow the table;
  // the column next to the row's exit side is spared: the line on it is still open (it closes on the next step)
  __device__ __forceinline__ void calm_padding() {
    if (!haspad) return;
#pragma unroll
    for (int j = 0; j < K; j++)
      if ((calm[j >> 2] >> (8 * (j & 3))) & 0xffu) pl[j] = 0;
  }
  __device__ __forceinline__ bool any_alive() {
    u32 X[KW];
    fw_make_x<K, DX>(P, X);
    bool a = false;
#pragma unroll
    for (int j = 0; j < K; j++) a = a || (pl[j] >= 0 && FW_BYTE(X, j) != 0);
    return __ballot(a) != 0;
  }
  // every line of the window ends here (end of the walk / row wrap): close the open runs
  __device__ __forceinline__ void close_all() {
    u32 X[KW];
    fw_make_x<K, DX>(P, X);
#pragma unroll
    for (int j = 0; j < K; j++) fw_checked<LONG, false>(T, dummy, pl[j], FW_BYTE(X, j), 0);
  }

  __device__ __forceinline__ void run(const FwDesc &D, int NX, const uint8_t *__restrict__ L, int *work) {
    const int NM = D.NM, NU = D.NU, du = D.du;
    const long long delta = D.sM + (long long)du * D.sU;
    const uint8_t *lp = L + first_col(NX);
    for (;;) {
      int grabbed = 0;
      if (lane == 0) grabbed = atomicAdd(work, 1);
      const int chunk = __builtin_amdgcn_readfirstlane(grabbed);
      if (chunk >= D.chunks) break;
      const int piece = chunk / NU, u0 = chunk - piece * NU;  // piece-major: concurrent waves share planes
      const int t0 = piece * D.CL, t1 = min(NM, t0 + D.CL);
      int row = (int)((u0 + (long long)t0 * du) % NU);
      if (row < 0) row += NU;
      long long off = (long long)t0 * D.sM + (long long)row * D.sU;
      const bool starts = t0 == 0 || (du > 0 && row == 0) || (du < 0 && row == NU - 1);
      if (starts) {
#pragma unroll
        for (int w = 0; w < KW; w++) P[w] = 0;
        reset_lines(0);
      } else {
        load_row(lp + (off - delta), P);
        reset_lines(PRAD_FW_DEAD);
      }
      int t = t0;
      bool wrap = false;  // the next step would leave the row range: all lines end first
      while (t < t1) {
        if (wrap) {
          close_all();
#pragma unroll
          for (int w = 0; w < KW; w++) P[w] = 0;
          reset_lines(0);
          row -= du * NU;
          off -= (long long)du * NU * D.sU;
          wrap = false;
        }
        const int room = du > 0 ? NU - row : (du < 0 ? row + 1 : (1 << 30));  // steps before the row range ends
        if (t + U <= t1 && room >= U) {
          u32 v[U][KW];
          const uint8_t *p = lp + off;
#pragma unroll
          for (int k = 0; k < U; k++) {
            load_row(p, v[k]);
            p += delta;
          }
          calm_padding();
          if (__ballot(risky()) != 0) {
#pragma unroll
            for (int k = 0; k < U; k++) single_step<false>(v[k]);
          } else {
            plain_group(v);
          }
          t += U;
          row += U * du;
          off += (long long)U * delta;
        } else {
          u32 c[KW];
          load_row(lp + off, c);
          single_step<false>(c);
          t++;
          row += du;
          off += delta;
        }
        if (du != 0 && (row < 0 || row >= NU)) wrap = true;
      }
      if (t1 == NM || wrap) {  // the lines end with the piece
        close_all();
        continue;
      }
      // tail: runs that began in this piece are walked to their end; nothing that begins later is recorded
      while (any_alive()) {
        u32 c[KW];
        load_row(lp + off, c);
        single_step<true>(c);
        t++;
        row += du;
        off += delta;
        if (t == NM || (du != 0 && (row < 0 || row >= NU))) {
          close_all();
          break;
        }
      }
    }
  }
};

template <bool LONG, int K>
__global__ void __launch_bounds__(1024) sweep_fw_kernel(FwSet set, const uint8_t *__restrict__ L, int Ng, int Nr, int RS,
                                                        u32 *__restrict__ glcm_acc, u32 *__restrict__ glrlm_acc,
                                                        int *__restrict__ work, int *__restrict__ flags) {
  extern __shared__ u32 lds[];
  if (flags[0]) return;  // irregular levels: the generic path will redo this call
  const HistLayout h = hist_layout(true, true, true, Ng, RS);
  if ((unsigned)(size_t)((lds_u32 *)lds) != 0u) {  // table offsets are used as LDS addresses
    if (threadIdx.x == 0) atomicExch(flags + 2, 1);
    return;
  }
  for (int i = threadIdx.x; i < h.words; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const FwDesc &D = set.d[blockIdx.y];
  FwTab T;
  T.init(h, Nr, glrlm_acc + (size_t)D.slot * Ng * Nr);
  if (D.dx == 0) {
    FwWave<LONG, K, 0> w(T, set.NX);
    w.run(D, set.NX, L, work + blockIdx.y);
  } else if (D.dx > 0) {
    FwWave<LONG, K, 1> w(T, set.NX);
    w.run(D, set.NX, L, work + blockIdx.y);
  } else {
    FwWave<LONG, K, -1> w(T, set.NX);
    w.run(D, set.NX, L, work + blockIdx.y);
  }
  flush_block_hist<true, true, true>(lds, h, Nr, D.slot, glcm_acc, glrlm_acc);
}

}  // namespace prad
