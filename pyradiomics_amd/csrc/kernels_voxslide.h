// kernels_voxslide.h -- voxel-based GLCM feature MAPS with a sliding window (round 4).
//
// kernels_voxel.h builds every kernel window from scratch: one wave per centre, three passes over the window's pairs per
// angle.  Neighbouring centres of a row share all but one plane of their windows: moving the centre by one voxel along x
// retires the pairs that touch the plane leaving the window and adds the pairs that touch the plane entering it -- 482
// updates instead of 3 x 1036 pair visits for a 5^3 window and 13 angles (34 instead of 3 x 72 for the 5 x 5 window of
// exampleVoxel.yaml).  The reference has no analogue: it materialises P[Nvox][Ng][Ng][Na] and evaluates the formulas with
// numpy (glcm.py:145-205, base.py:200-245); the values produced here are the same features of the same matrices.
//
// Work decomposition: the co-occurrence counts of one ANGLE of one row of centres live in an LDS table for the whole run of
// centres: symmetric counts m(lo, hi) of the unordered level pair, one byte each (a window of radius <= 2 holds at most 100 pairs
// per angle), Ng (Ng + 1) / 2 <= 528 bytes.  2-D windows: a lane per (row, angle) walks the plane's positions and owns the
// table.  3-D windows: the sixteen lanes of a group share the work of its thirteen angles (VoxSlideBal below).  Per angle:
//     S   = sum over matrix entries of n log2 n   (n = 2 m on the diagonal, m elsewhere and there twice), as a 2^-40
//           fixed-point integer built from table DIFFERENCES g(c) = f(c + 1) - f(c): the sum telescopes, so S is exactly
//           the sum of the rounded f over the current counts whatever the history of the window -- no drift along a run,
//           results independent of where a run starts, of which lane applied a pair and of the order of the atomics
//     nnz = non-zero entries (carried in the bits of S above its 51), E2 = sum of n^2, P = pairs (one word: P << 20 | E2),
//     IJ = sum of (i + j) over the pairs
// and JointEntropy = log2 T - S / T - nnz eps / ln 2 (the first-order expansion of -sum p log2(p + eps), p = n / T,
// T = 2 P; the neglected term is < 1e-28), JointEnergy = E2 / T^2, JointAverage = IJ / T follow per angle; the mean over
// the non-empty angles of a centre (np.nanmean, glcm.py:260-887) is a sum over the 16 (3-D, 13 angles) or 4 (2-D window,
// 4 angles) lanes of a group.  A wave holds 4 (16) groups = 4 (16) neighbouring rows.
// The levels a group needs -- its row's planes of (2R+1)^2 (2R+1) voxels per x -- are staged once per run into LDS, a
// plane per 32 (8) bytes behind one plane of zeros; a run starts 2R+1 planes early with an empty table (planes only enter).
//
// Covered: 3 image dimensions, symmetrical GLCM, distance 1, kernelRadius 1 or 2, Ng <= 64 (round 5: the table size TB and
// the waves per workgroup are template parameters -- 32 levels: 532 B tables, 4 waves; 40: 828 B, 2 / 3 waves; 48: 1180 B,
// 1 / 2 waves; 64: 2084 B, one wave -- so that the reference's own example, exampleVoxel.yaml on brain1 with its 33 levels,
// takes this kernel), full 3-D windows or force2D along the first (slice) axis, the features above.  Everything else stays
// on kernels_voxel.h.  Checked against the reference route (tests/test_gpu_configs.py, tests/test_gpu_stress.py) and the
// window kernel (tests/test_gpu_features.py).  Round 6 (profiles/r06_probes.md section 12): 5^3 window at 512^3 207 -> 91 ms,
// 5 x 5 window 23.0 -> 12.2 ms, bit-identical maps.
#pragma once
#include "prad_runtime.h"
#include "kernels_voxel.h"

namespace prad {

#define PRAD_VS_TB 532            // bytes of a count table at Ng <= 32: 528 of counts and a spare word; larger Ng: template TB.
                                  // An ODD number of words: neighbouring voxels have neighbouring levels, the lanes of a wave
                                  // update the same few entries of their tables at once, and with an even stride (544 B until
                                  // round 6) every fourth table put that entry on the same LDS bank
#define PRAD_VS_FIX 40            // fixed-point fraction bits of S
#define PRAD_VS_LUT 104           // counts of a table entry <= 100 (pairs of one angle in a 5^3 window); the last entry: the absent pair
#define PRAD_VS_NNZ_SHIFT 52      // S carries nnz above its 11 + 40 bits

// WIDE (round 5): fourteen more features whose sums update pair by pair -- Autocorrelation, ClusterProminence / Shade / Tendency,
// Contrast, DifferenceAverage / Variance, Id, Idm, Idn, Idmn, InverseVariance, SumAverage, SumSquares (glcm.py:260-887).  A lane
// additionally carries the integer sums A = sum ij, Q2 = sum (i^2 + j^2), D1 = sum |i - j|, S3 = sum (i + j)^3, S4 = sum (i + j)^4
// over its angle's pairs and five 2^-40 fixed-point sums of g(|i - j|) (1 / (1 + k), 1 / (1 + k^2), 1 / (1 + k / Ng),
// 1 / (1 + k^2 / Ng^2), 1 / k^2); the central moments come out of EXACT int64 expressions (P^3 S4 - 4 P^2 S1 S3 + ... fits: a
// window of radius <= 2 holds at most 100 pairs per angle, levels <= 64), so there is no cancellation to lose digits in.
// Not carried (they stay on kernels_voxel.h): Correlation (the reference's sigma == 0 test is a float accident), the entropies of
// the sum / difference distributions, Imc1 / Imc2, MCC, MaximumProbability.
#define PRAD_VS_KMAX 64
// Tuning knobs (the defaults are what profiles/r06_probes.md section 12 measured best; scripts/build_variant.sh builds others):
#ifndef PRAD_VS_CH
#define PRAD_VS_CH 9              // pair positions in flight per stage where a lane per angle walks a 3-D plane (the WIDE sums)
#endif
#ifndef PRAD_VS_CHB
#define PRAD_VS_CHB 16            // ... on the lane-balanced schedule (16 entering, then 16 leaving)
#endif
#ifndef PRAD_VS_UNROLL
#define PRAD_VS_UNROLL 1          // sliding steps per loop body of the balanced schedule (3: no gain)
#endif
#ifndef PRAD_VS_BAL_R1
#define PRAD_VS_BAL_R1 1          // 3^3 windows on the balanced schedule too (0: a lane per angle, 51 instead of 43 ms at 512^3)
#endif
struct VoxSlideLutK {            // g_f(k) * 2^40, f = Id, Idm, Idn, Idmn, InverseVariance (built per call: Idn / Idmn depend on Ng)
  long long g[5][PRAD_VS_KMAX];
};
struct VoxSlideSlots {           // output slot of every VoxelGlcmFeature (kernels_voxel.h), -1 = not requested
  int s[VF_COUNT];
};

// What ONE pair adds to (takes from) a lane's sums when the count of its table entry goes c -> c + 1 (c + 1 -> c), one 16-byte
// LDS read per pair (round 6; until then three sums were computed with selects per pair).  Entry PRAD_VS_LUT - 1 is all zero:
// the read of an absent pair (a position outside the angle's plane overlap, a voxel outside the mask) lands there.
struct VoxSlideLutE {
  long long g;                   // off-diagonal: 2 (f(c+1) - f(c)), f(c) = round(c log2 c * 2^40) -- the pair fills two entries;
                                 // diagonal: f(2c+2) - f(2c) -- it adds 2 to one; plus (c == 0 ? entries that become non-zero : 0) << 52
  int ep;                        // 1 << 20 (the pair itself) | change of sum n^2: 2 (2c + 1) / 4 (2c + 1)   (sum n^2 <= 200^2 < 2^20)
  int pad;
};
struct VoxSlideLut {             // built once on the host (prad_api.hip), lives in global memory, copied to LDS per workgroup
  VoxSlideLutE off[PRAD_VS_LUT];
  VoxSlideLutE dia[PRAD_VS_LUT];
  struct PT { double lg2T, inv; } pt[PRAD_VS_LUT];      // log2(2 P) and 1 / (2 P) for P pairs (the quotient the division gives)
  double inv_na[16];             // 1 / n for n non-empty angles (n = 0: NaN)
  // LIGHT kernels (JointEntropy alone: no sum n^2) read 8 bytes per pair -- the g column of the entries above -- and count the
  // pairs themselves; in LDS the two columns take the place of off / dia
  long long g_off[PRAD_VS_LUT], g_dia[PRAD_VS_LUT];
};
#define PRAD_VS_LUT_LDS (sizeof(VoxSlideLutE) * 2 * PRAD_VS_LUT + 16 * PRAD_VS_LUT + 128)      // bytes of LDS the LUTs take

// sum over the lanes of a group of GS (4 or 16) neighbouring lanes, result in every lane of the group
template <int GS>
__device__ __forceinline__ int group_sum_i32(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
  if (GS == 16) {
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true);   // row_mirror
  }
  return v;
}
template <int GS>
__device__ __forceinline__ double group_sum_f64(double v) {
  auto sh = [](double x, const int ctrl) __attribute__((always_inline)) {
    const long long b = __double_as_longlong(x);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = ctrl == 0 ? __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true)
       : ctrl == 1 ? __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xf, 0xf, true)
       : ctrl == 2 ? __builtin_amdgcn_mov_dpp(lo, 0x141, 0xf, 0xf, true)
                   : __builtin_amdgcn_mov_dpp(lo, 0x140, 0xf, 0xf, true);
    hi = ctrl == 0 ? __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true)
       : ctrl == 1 ? __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xf, 0xf, true)
       : ctrl == 2 ? __builtin_amdgcn_mov_dpp(hi, 0x141, 0xf, 0xf, true)
                   : __builtin_amdgcn_mov_dpp(hi, 0x140, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  };
  v += sh(v, 0);
  v += sh(v, 1);
  if (GS == 16) {
    v += sh(v, 2);
    v += sh(v, 3);
  }
  return v;
}

// Lane-balanced schedule of a 3-D window (round 6).  An angle (dz, dy, dx) pairs up (D - |dz|) (D - |dy|) of the D^2 positions
// of a plane: 25, 20 or 16 of 25 at radius 2, and three of a group's sixteen lanes have no angle at all -- with a lane per angle
// walking all 25 positions, 40 % of the slots of a step were spent on absent pairs.  Instead every lane visits NSLOT = 16 pair
// positions per entering (leaving) plane: an angle's lane takes the first 16 of its own, the three helper lanes take the
// overflow (9 positions of the angle (0, 0, 1), 4 of each of the six angles with 20) in segments of SEG slots, one angle per segment, and
// update THAT angle's table (LDS atomics: whose lane adds is immaterial).  Their sums stay apart per segment; once per centre a
// helper leaves them in LDS records and the angle's lane adds the (at most MAXQ) records that belong to it.  The schedule is a
// function of the 13 angles and the radius only; prad_api.hip builds it (voxslide_schedule).
template <int R>
struct VoxSlideBal {
  static constexpr int NSLOT = R == 2 ? 16 : 5;    // pair positions a lane visits per entering (leaving) plane
  static constexpr int SEG = R == 2 ? 4 : 1;       // slots of a helper lane that serve one angle
  static constexpr int NSEG = NSLOT / SEG;
  static constexpr int MAXQ = R == 2 ? 3 : 4;      // helper segments the overflow of one angle takes at most
  static constexpr int NREC = 3 * NSEG + 1;        // records of a group: 3 helpers x NSEG segments, and one of zeros
};
struct VoxSlideSched {
  unsigned slot[16][16];      // [lane of a group][k]: p | q << 8 | angle whose table it counts in << 16 | (dx + 1) << 20 | valid << 24
  unsigned char rec[16][4];   // [angle lane][j]: the helper records that hold parts of its sums (255: none)
};
struct VoxSlideRec {          // a helper segment's sums (16 bytes)
  long long S;
  int EP, IJ;
};

// R: kernel radius (1, 2); TWO_D: the window has no extent along z (force2D on the slice axis, or a single slice)
// RUN: centres per run.  Grid: x = runs along x, y = groups of rows, z = slices; a workgroup = 4 waves (fewer above 32 levels) =
// consecutive runs -- what 160 KB of LDS hold: 13 x 4 (64) tables of 532 B and 4 (16) rows of staged planes per wave.
// maps: [nmaps][Nz][Ny][Nx] float64 (slot < 0: feature not requested); empty: [Nz][Ny][Nx] angle bits without a pair.
template <int R, bool TWO_D, int RUN, int TB, int WAVES, bool WIDE = false>
constexpr size_t voxel_glcm_slide_lds() {
  return PRAD_VS_LUT_LDS + (WIDE ? sizeof(VoxSlideLutK) : 0) +
         (size_t)WAVES * ((TWO_D ? 64 : 13 * (64 / 16)) * TB + (64 / (TWO_D ? 4 : 16)) * (RUN + 2 * R + (TWO_D ? 1 : 2)) * (TWO_D ? 8 : 32) +
                          (!TWO_D && !WIDE ? 4 * VoxSlideBal<R>::NREC * 16 : 0));
}
template <int R, bool TWO_D, int RUN, int TB = PRAD_VS_TB, int WAVES = (TWO_D ? 3 : 4), bool WIDE = false, bool JA = true, bool LIGHT = false>
__global__ void __launch_bounds__(64 * WAVES) voxel_glcm_slide_kernel(const uint8_t *__restrict__ L, int Nz, int Ny, int Nx, VoxAngles A,
                                                              int Ng, const VoxSlideLut *__restrict__ lut_g,
                                                              const VoxSlideLutK *__restrict__ lutk_g,
                                                              const VoxSlideSched *__restrict__ sched, VoxSlideSlots sl,
                                                              double *__restrict__ maps,
                                                              unsigned *__restrict__ empty, const int *__restrict__ flags, int z_begin) {
  const int slot_ent = sl.s[VF_JointEntropy], slot_en = sl.s[VF_JointEnergy], slot_ja = sl.s[VF_JointAverage];
  constexpr int D = 2 * R + 1;
  constexpr int PZ = TWO_D ? 1 : D;            // plane extent along z
  constexpr int NP = PZ * D;                   // voxels of a plane
  constexpr int PB = TWO_D ? 8 : 32;           // bytes of a staged plane
  constexpr int GS = TWO_D ? 4 : 16;           // lanes per group (>= angles)
  constexpr int NGR = 64 / GS;                 // groups (rows) per wave
  constexpr int XL = RUN + 2 * R;              // planes a run needs
  constexpr int GSTR = (XL + (TWO_D ? 1 : 2)) * PB;   // bytes of a group's planes: one of zeros, then the XL of the run; 3-D: and
                                               // one to make the stride 16 banks (mod 32) -- the two groups of a half wave read
                                               // two neighbouring planes = 16 banks each, side by side
  constexpr bool BAL = !TWO_D && !WIDE && (R == 2 || PRAD_VS_BAL_R1);        // the lane-balanced schedule (above)
  using BL = VoxSlideBal<R>;
  constexpr int NT = TWO_D ? 64 : 13 * NGR;    // count tables per wave (3-D: the 13 angle lanes of each group)
  static_assert(TB % 8 == 4 && (NT * TB) % 16 == 0, "tables: an odd number of words each, cleared 16 bytes at a time");
  static_assert(NP < PB, "a staged plane keeps a zero byte behind its voxels");
  extern __shared__ __align__(16) unsigned char vs_smem[];
  if (flags[0]) return;                        // a level outside [1, Ng]: the caller reruns on the matrix path
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroup LDS: the LUTs, then per wave: tables [64][TB], planes [NGR][1 + XL][PB], the helper records [NGR][NREC]
  static_assert(!LIGHT || (!WIDE && !JA), "LIGHT: JointEntropy alone");
  VoxSlideLut *lut = reinterpret_cast<VoxSlideLut *>(vs_smem);      // (off, dia, pt, inv_na; LIGHT: g_off, g_dia at the place of off)
  long long *lg_off = reinterpret_cast<long long *>(vs_smem), *lg_dia = lg_off + PRAD_VS_LUT;
  constexpr int WAVE_BYTES = NT * TB + NGR * GSTR + (BAL ? NGR * BL::NREC * 16 : 0);
  static_assert(PRAD_VS_LUT_LDS % 16 == 0 && sizeof(VoxSlideLutK) % 16 == 0 && WAVE_BYTES % 16 == 0 && (NGR * GSTR) % 16 == 0,
                "16-byte clears and reads");
  long long *gk = reinterpret_cast<long long *>(vs_smem + PRAD_VS_LUT_LDS);      // WIDE: [5][PRAD_VS_KMAX]
  unsigned char *wbase = vs_smem + PRAD_VS_LUT_LDS + (WIDE ? sizeof(VoxSlideLutK) : 0) + (size_t)wave * WAVE_BYTES;
  unsigned char *planes = wbase + NT * TB;
  VoxSlideRec *recs = reinterpret_cast<VoxSlideRec *>(planes + NGR * GSTR);
  for (int i = threadIdx.x; i < (int)(PRAD_VS_LUT_LDS / 16); i += blockDim.x)
    reinterpret_cast<uint4 *>(vs_smem)[i] = reinterpret_cast<const uint4 *>(lut_g)[i];
  if (LIGHT) {
    __syncthreads();                                             // (the copy above put off / dia there)
    for (int i = threadIdx.x; i < 2 * PRAD_VS_LUT; i += blockDim.x) lg_off[i] = lut_g->g_off[i];      // (g_dia follows g_off)
  }
  if (WIDE) {
    for (int i = threadIdx.x; i < 5 * PRAD_VS_KMAX; i += blockDim.x) gk[i] = lutk_g->g[i / PRAD_VS_KMAX][i % PRAD_VS_KMAX];
  }
  // this wave's run
  const int nruns = (Nx + RUN - 1) / RUN;
  const int run = blockIdx.x * WAVES + wave;
  const int y0 = blockIdx.y * NGR, z = blockIdx.z + z_begin;
  const bool live = run < nruns;               // (all waves reach the barrier below)
  const int x0 = run * RUN;
  const int grp = lane / GS, a = lane % GS;
  const int y = y0 + grp;
  const bool has_angle = a < A.na && a < (TWO_D ? 4 : 13);
  unsigned char *tbl = wbase + (TWO_D ? lane : grp * 13 + min(a, 12)) * TB;
  const int dz = has_angle ? A.o[a][0] : 0, dy = has_angle ? A.o[a][1] : 0, dx = has_angle ? A.o[a][2] : 0;
  if (live) {
    // clear the tables (wave-private: 64 x TB bytes), the planes (their spare bytes and the planes of zeros stay 0), the records
    for (int i = lane; i < WAVE_BYTES / 16; i += 64) reinterpret_cast<uint4 *>(wbase)[i] = make_uint4(0, 0, 0, 0);
    // stage the planes: slab x index k <-> global x0 - R + k; plane byte p = pz * D + py <-> (z - R + pz (TWO_D: z), y - R + py)
    for (int e = lane; e < NGR * XL * NP; e += 64) {
      const int p = e % NP, r = e / NP, k = r % XL, g = r / XL;
      const int pz = p / D, py = p % D;
      const int gz = TWO_D ? z : z - R + pz, gy = y0 + g - R + py, gx = x0 - R + k;
      unsigned char v = 0;
      if ((unsigned)gz < (unsigned)Nz && (unsigned)gy < (unsigned)Ny && (unsigned)gx < (unsigned)Nx)
        v = L[((long long)gz * Ny + gy) * Nx + gx];
      planes[g * GSTR + (k + 1) * PB + p] = v;
    }
  }
  __syncthreads();
  if (!live) return;
  const unsigned char *gp = planes + grp * GSTR + PB;   // this group's planes of the run (gp - PB: the plane of zeros)
  constexpr int TRASH = TB - 4;                          // (the byte index of) a word no level pair counts in
  const VoxSlideLutE *lut_off = lut->off, *lut_dia = lut->dia;
  int wA = 0, wQ2 = 0, wD1 = 0, wS3 = 0;            // WIDE: sum ij, sum (i^2 + j^2), sum |i - j|, sum (i + j)^3
  long long wS4 = 0, wF0 = 0, wF1 = 0, wF2 = 0, wF3 = 0, wF4 = 0;
  const bool wantF = WIDE && (sl.s[VF_Id] >= 0 || sl.s[VF_Idm] >= 0 || sl.s[VF_Idn] >= 0 || sl.s[VF_Idmn] >= 0 || sl.s[VF_InverseVariance] >= 0);
  const double eps_ln2 = 2.220446049250313e-16 / 0.6931471805599453;
  const double fix = 1.0 / (double)(1LL << PRAD_VS_FIX);
  // the centre at slab x index s - R, from this lane's sums: S (nnz in its bits from PRAD_VS_NNZ_SHIFT up), EP = P << 20 | E2, IJ
  auto emit = [&](int s, long long S, int EP, int IJ) __attribute__((always_inline)) {
    const int gx = x0 + s - 2 * R;
#ifdef PRAD_DBG_NOEMIT      // (timing probe, wrong maps: what the pair code alone costs)
    if (a == 0 && y < Ny && gx < Nx) maps[((long long)z * Ny + y) * Nx + gx] = __longlong_as_double(S + EP + IJ);
    return;
#endif
    const int P = EP >> 20, E2 = EP & 0xfffff;
    const bool nonempty = has_angle && P > 0;
    const int pc = nonempty ? P : 1;
    const VoxSlideLut::PT pt = lut->pt[pc];
    const double iT = pt.inv;                              // 1 / T, T = 2 P
    double h = 0, en = 0, ja = 0;
    if (nonempty) {
      if (slot_ent >= 0) {
        const long long Sf = S & ((1LL << PRAD_VS_NNZ_SHIFT) - 1);
        const int nnz = (int)(S >> PRAD_VS_NNZ_SHIFT);
        h = pt.lg2T - ((double)Sf * fix) * iT - (double)nnz * eps_ln2;
      }
      if (slot_en >= 0) en = (double)E2 * iT * iT;
      if (slot_ja >= 0) ja = (double)IJ * iT;
    }
    const int na = group_sum_i32<GS>(nonempty ? 1 : 0);
    const int em = group_sum_i32<GS>(has_angle && !nonempty ? (1 << a) : 0);
    const double inv = lut->inv_na[na];
    if (slot_ent >= 0) h = group_sum_f64<GS>(h) * inv;
    if (slot_en >= 0) en = group_sum_f64<GS>(en) * inv;
    if (slot_ja >= 0) ja = group_sum_f64<GS>(ja) * inv;
    const bool writer = a == 0 && y < Ny && gx < Nx;
    const long long vi = ((long long)z * Ny + y) * Nx + gx, nmap = (long long)Nz * Ny * Nx;
    if (writer) {
      if (slot_ent >= 0) maps[(long long)slot_ent * nmap + vi] = h;
      if (slot_en >= 0) maps[(long long)slot_en * nmap + vi] = en;
      if (slot_ja >= 0) maps[(long long)slot_ja * nmap + vi] = ja;
      empty[vi] = (unsigned)em;
    }
    if (WIDE) {
      // per angle (P pairs, T = 2 P entries, symmetric matrix): every expression below is the reference's sum over the
      // normalised matrix written in the pair sums; integer numerators are exact
      const double dP = (double)pc, iP = 1.0 / dP, iP2 = iP * iP;
      const long long lP = pc, S1 = IJ, S2 = (long long)wQ2 + 2LL * wA, D2 = (long long)wQ2 - 2LL * wA;
      auto put = [&](int f, double v) __attribute__((always_inline)) {
        const int slot = sl.s[f];
        if (slot < 0) return;                                   // (wave-uniform)
        const double m = group_sum_f64<GS>(nonempty ? v : 0.0) * inv;
        if (writer) maps[(long long)slot * nmap + vi] = m;
      };
      put(VF_Autocorrelation, (double)wA * iP);
      put(VF_SumAverage, (double)IJ * iP);
      put(VF_Contrast, (double)D2 * iP);
      put(VF_DifferenceAverage, (double)wD1 * iP);
      put(VF_DifferenceVariance, (double)(lP * D2 - (long long)wD1 * wD1) * iP2);
      put(VF_SumSquares, (double)(2LL * lP * wQ2 - S1 * S1) * 0.25 * iP2);                 // (T Q2 - IJ^2) / T^2
      put(VF_ClusterTendency, (double)(lP * S2 - S1 * S1) * iP2);
      put(VF_ClusterShade, (double)(lP * lP * wS3 - 3LL * lP * S1 * S2 + 2LL * S1 * S1 * S1) * iP2 * iP);
      put(VF_ClusterProminence,
          (double)(lP * lP * lP * wS4 - 4LL * lP * lP * S1 * wS3 + 6LL * lP * S1 * S1 * S2 - 3LL * S1 * S1 * S1 * S1) * iP2 * iP2);
      put(VF_Id, (double)wF0 * fix * iP);
      put(VF_Idm, (double)wF1 * fix * iP);
      put(VF_Idn, (double)wF2 * fix * iP);
      put(VF_Idmn, (double)wF3 * fix * iP);
      put(VF_InverseVariance, (double)wF4 * fix * iP);
    }
  };
  // A pair enters (SIGN = +1) or leaves (-1) a count table.  Straight-line code in four stages per chunk of pair positions --
  // level reads, table atomics, LUT reads, accumulation -- without branches: a pair at a time under its own branch left every LDS
  // round trip exposed (three per pair, ~14 000 cycles per step of a 5^3 window; profiles/r04_probes.md).  Round 6: an absent
  // pair costs no selects either -- its q read lands on a zero byte (the spare byte of a staged plane; the plane of zeros in
  // front of the run), so that its smaller level is 0; then its table update goes to the spare word behind the counts and its
  // LUT read to the all-zero entry, and the sums take what the read returns (profiles/r06_probes.md section 12).
  //   l1, l2: the levels; tb: the table; returns through e / ok what stage 4 adds
#define PRAD_VS_STAGE2(K, TBP)                                                                                       \
  {                                                                                                                  \
    const int lo = min(l1[K], l2[K]), hi = max(l1[K], l2[K]);                                                        \
    ok[K] = lo != 0;                                                                                                 \
    const int real = (hi * (hi - 1) >> 1) + lo - 1;     /* (outside the select: one v_cndmask, not an exec-masked region) */ \
    const int idx = ok[K] ? real : TRASH;                                                                            \
    dg[K] = lo == hi;                                                                                                \
    shf[K] = idx * 8;                                   /* (shifts and bit-field extracts take its low five bits) */ \
    const unsigned inc = SIGN > 0 ? (1u << (shf[K] & 31)) : (0u - (1u << (shf[K] & 31)));                            \
    old[K] = __hip_atomic_fetch_add((TBP) + (idx >> 2), inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        \
  }
  // (c = the smaller of the entry's counts before / after)
#define PRAD_VS_STAGE3(K)                                                                                            \
  {                                                                                                                  \
    const int c = ok[K] ? (int)__builtin_amdgcn_ubfe(old[K], (unsigned)shf[K], 8u) - (SIGN > 0 ? 0 : 1) : PRAD_VS_LUT - 1; \
    if (LIGHT) e[K].g = (dg[K] ? lg_dia : lg_off)[c];                                                                \
    else e[K] = (dg[K] ? lut_dia : lut_off)[c];                                                                      \
  }
  if constexpr (BAL) {
    constexpr int NSLOT = BL::NSLOT, SEG = BL::SEG, NSEG = BL::NSEG, MAXQ = BL::MAXQ, NREC = BL::NREC;
    unsigned *tb[NSLOT];
    int a1P[NSLOT], a2P[NSLOT], a1M[NSLOT], a2M[NSLOT];    // bytes of the p and q voxels relative to the entering (leaving) plane
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
      const unsigned d = sched->slot[a][k];
      const int p = d & 31, q = (d >> 8) & 31, t = (d >> 16) & 15, dxk = (int)((d >> 20) & 3) - 1;
      const bool valid = (d >> 24) != 0;
      tb[k] = reinterpret_cast<unsigned *>(wbase + (grp * 13 + t) * TB);
      // entering plane s: pairs inside it (dx = 0) or with plane s - 1 (p side for dx > 0, q side for dx < 0);
      // leaving plane o = s - D: pairs inside it or with plane o + 1
      a1P[k] = p - (dxk > 0 ? PB : 0);
      a2P[k] = valid ? q - (dxk < 0 ? PB : 0) : PB - 1;
      a1M[k] = p + (dxk < 0 ? PB : 0);
      a2M[k] = valid ? q + (dxk > 0 ? PB : 0) : PB - 1;
    }
    VoxSlideRec *grec = recs + grp * NREC;
    const VoxSlideRec *mine[MAXQ];
#pragma unroll
    for (int j = 0; j < MAXQ; j++) {
      const int r = sched->rec[a][j];
      mine[j] = grec + (r == 255 ? NREC - 1 : r);
    }
    VoxSlideRec *hrec = grec + max(a - 13, 0) * NSEG;      // a helper's NSEG records
    long long Sq[NSEG];
    int EPq[NSEG], IJq[NSEG];
#pragma unroll
    for (int r = 0; r < NSEG; r++) { Sq[r] = 0; EPq[r] = 0; IJq[r] = 0; }
    // the pairs of entering plane s (base_p) and, with BOTH, of leaving plane s - D (base_m) in one straight line: the level
    // reads of the leaving plane are under way while the entering plane's pairs are counted
    auto slots = [&](const unsigned char *base_p, const unsigned char *base_m, auto both_tag) __attribute__((always_inline)) {
      constexpr int T = decltype(both_tag)::value ? 2 * NSLOT : NSLOT;
      constexpr int CH = T > PRAD_VS_CHB ? PRAD_VS_CHB : T;      // slots in flight
#pragma unroll
      for (int c0 = 0; c0 < T; c0 += CH) {
        int l1[CH], l2[CH], shf[CH];
        unsigned old[CH];
        bool ok[CH], dg[CH];
        VoxSlideLutE e[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int j = c0 + k, jj = j % NSLOT;
          if (j < T) {
            l1[k] = j < NSLOT ? base_p[a1P[jj]] : base_m[a1M[jj]];
            l2[k] = j < NSLOT ? base_p[a2P[jj]] : base_m[a2M[jj]];
          }
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int j = c0 + k, SIGN = j < NSLOT ? 1 : -1;
          if (j < T) PRAD_VS_STAGE2(k, tb[j % NSLOT])
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int j = c0 + k, SIGN = j < NSLOT ? 1 : -1;
          if (j < T) PRAD_VS_STAGE3(k)
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int j = c0 + k, r = (j % NSLOT) / SEG;
          if (j < T) {
            const int ij = JA && ok[k] ? l1[k] + l2[k] : 0;
            const int ep = LIGHT ? (ok[k] ? 1 : 0) : e[k].ep;      // (LIGHT: EPq counts the pairs)
            if (j < NSLOT) { Sq[r] += e[k].g; EPq[r] += ep; IJq[r] += ij; }
            else { Sq[r] -= e[k].g; EPq[r] -= ep; IJq[r] -= ij; }
          }
        }
      }
    };
    auto centre = [&](int s) __attribute__((always_inline)) {
      if (a >= 13) {
#pragma unroll
        for (int r = 0; r < NSEG; r++) hrec[r] = VoxSlideRec{Sq[r], EPq[r], IJq[r]};
      }
      long long S = Sq[0];
      int EP = EPq[0], IJ = IJq[0];
#pragma unroll
      for (int r = 1; r < NSEG; r++) { S += Sq[r]; EP += EPq[r]; IJ += IJq[r]; }
#pragma unroll
      for (int j = 0; j < MAXQ; j++) {
        const VoxSlideRec rr = *mine[j];
        S += rr.S;
        EP += rr.EP;
        IJ += rr.IJ;
      }
      emit(s, S, LIGHT ? EP << 20 : EP, IJ);
    };
    // the window fills (planes enter only), then slides: XL - D = RUN - 1 steps (PRAD_VS_UNROLL steps to a loop body: the 64
    // level addresses of a step could then be bumped once per body -- measured no gain at 3, profiles/r06_probes.md section 12)
    for (int s = 0; s < D; s++) {
      slots(gp + s * PB, gp, std::false_type{});
      if (s >= 2 * R) centre(s);
    }
    static_assert((XL - D) % PRAD_VS_UNROLL == 0, "the sliding steps come in whole loop bodies");
    for (int s = D; s < XL; s += PRAD_VS_UNROLL) {
#pragma unroll
      for (int u = 0; u < PRAD_VS_UNROLL; u++) {
        slots(gp + (s + u) * PB, gp + (s + u - D) * PB, std::true_type{});
        centre(s + u);
      }
    }
  } else {
    // a lane per angle walks the plane: position p = (pz, py) pairs up with q = (pz + dz, py + dy) where that lies inside
    const int pz_lo = max(0, -dz), pz_hi = min(PZ, PZ - dz), py_lo = max(0, -dy), py_hi = min(D, D - dy);
    const int qoff = dz * D + dy;                          // byte offset of q relative to p inside a plane
    int qa[NP];                                            // byte of the q plane position p pairs up with (PB - 1: none)
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int pz = p / D, py = p % D;
      qa[p] = has_angle && pz >= pz_lo && pz < pz_hi && py >= py_lo && py < py_hi ? p + qoff : PB - 1;
    }
    unsigned *tbl32 = reinterpret_cast<unsigned *>(tbl);
    long long S = 0;
    int EP = 0, IJ = 0;
    // plane s enters -- pairs inside it (dx = 0) or with plane s - 1 (p side for dx > 0, q side for dx < 0; the run's first plane
    // has the plane of zeros there: no pairs) -- and, with BOTH, plane o = s - D leaves -- pairs inside it or with plane o + 1 --
    // in one straight line for the 2-D windows (the 5 x 5 window has 5 positions per plane: one sign at a time left four LDS
    // round trips per five pairs exposed)
    // (MODE 0: the entering plane, 1: both, 2: the leaving plane)
    auto plane_pairs = [&](int s, auto mode_tag) __attribute__((always_inline)) {
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr int J0 = MODE == 2 ? NP : 0, T = MODE == 0 ? NP : 2 * NP;
      constexpr int CH = T - J0 <= 10 ? T - J0 : PRAD_VS_CH;         // positions in flight
      const int o = s - D;
      const unsigned char *pp_p = gp + (dx > 0 ? s - 1 : s) * PB, *pq_p = gp + (dx < 0 ? s - 1 : s) * PB;
      const unsigned char *pp_m = gp + (dx < 0 ? o + 1 : o) * PB, *pq_m = gp + (dx > 0 ? o + 1 : o) * PB;
#pragma unroll
      for (int c0 = J0; c0 < T; c0 += CH) {
        int l1[CH], l2[CH], shf[CH];
        unsigned old[CH];
        bool ok[CH], dg[CH];
        VoxSlideLutE e[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int j = c0 + k, p = j % NP;
          if (j < T) {
            l1[k] = j < NP ? pp_p[p] : pp_m[p];
            l2[k] = j < NP ? pq_p[qa[p]] : pq_m[qa[p]];
          }
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int j = c0 + k, SIGN = j < NP ? 1 : -1;
          if (j < T) PRAD_VS_STAGE2(k, tbl32)
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int j = c0 + k, SIGN = j < NP ? 1 : -1;
          if (j < T) PRAD_VS_STAGE3(k)
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int j = c0 + k, SIGN = j < NP ? 1 : -1;
          if (j < T) {
            const int ij = JA && ok[k] ? l1[k] + l2[k] : 0;
            const int ep = LIGHT ? (ok[k] ? 1 : 0) : e[k].ep;      // (LIGHT: EP counts the pairs)
            if (SIGN > 0) { S += e[k].g; EP += ep; IJ += ij; }
            else { S -= e[k].g; EP -= ep; IJ -= ij; }
            if (WIDE) {
              const int i1 = ok[k] ? l1[k] : 0, j1 = ok[k] ? l2[k] : 0;
              const int kd = i1 > j1 ? i1 - j1 : j1 - i1, sm = i1 + j1, sm2 = sm * sm;
              const int a = i1 * j1, q2 = i1 * i1 + j1 * j1, s3 = sm2 * sm;
              const long long s4 = (long long)sm2 * sm2;
              if (SIGN > 0) { wA += a; wQ2 += q2; wD1 += kd; wS3 += s3; wS4 += s4; }
              else { wA -= a; wQ2 -= q2; wD1 -= kd; wS3 -= s3; wS4 -= s4; }
              if (wantF) {                                     // (wave-uniform)
                const int kk = min(kd, PRAD_VS_KMAX - 1);
                const long long f0 = ok[k] ? gk[kk] : 0, f1 = ok[k] ? gk[PRAD_VS_KMAX + kk] : 0, f2 = ok[k] ? gk[2 * PRAD_VS_KMAX + kk] : 0,
                                f3 = ok[k] ? gk[3 * PRAD_VS_KMAX + kk] : 0, f4 = ok[k] ? gk[4 * PRAD_VS_KMAX + kk] : 0;
                if (SIGN > 0) { wF0 += f0; wF1 += f1; wF2 += f2; wF3 += f3; wF4 += f4; }
                else { wF0 -= f0; wF1 -= f1; wF2 -= f2; wF3 -= f3; wF4 -= f4; }
              }
            }
          }
        }
      }
    };
    if constexpr (TWO_D) {
      for (int s = 0; s < D; s++) {                          // the window fills
        plane_pairs(s, std::integral_constant<int, 0>{});
        if (s >= 2 * R) emit(s, S, LIGHT ? EP << 20 : EP, IJ);
      }
      for (int s = D; s < XL; s++) {                         // and slides
        plane_pairs(s, std::integral_constant<int, 1>{});
        emit(s, S, LIGHT ? EP << 20 : EP, IJ);
      }
    } else {
      // (3-D windows come here with the WIDE sums only: 25 positions per plane at ~80 instructions each -- one sign at a time and
      // one copy of the code measured 4 % faster than the merged / peeled form)
      for (int s = 0; s < XL; s++) {
        plane_pairs(s, std::integral_constant<int, 0>{});
        if (s >= D) plane_pairs(s, std::integral_constant<int, 2>{});      // (wave-uniform)
        if (s >= 2 * R) emit(s, S, LIGHT ? EP << 20 : EP, IJ);
      }
    }
  }
#undef PRAD_VS_STAGE2
#undef PRAD_VS_STAGE3
}

// smallest and largest slice index of the requested centres (the map is only built for those slices): zr[0] = min, zr[1] = max
__global__ void __launch_bounds__(256) voxel_zrange_kernel(const int *__restrict__ voxels, int nvox, int *__restrict__ zr) {
  int lo = 0x7fffffff, hi = -1;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += gridDim.x * blockDim.x) {
    const int z = voxels[v];
    lo = min(lo, z);
    hi = max(hi, z);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0 && hi >= 0) {
    atomicMin(&zr[0], lo);
    atomicMax(&zr[1], hi);
  }
}

// out[f][v] = maps[f][voxel v]; empty_mask[v]; any_nonempty |= the angle bits that hold a pair somewhere
__global__ void __launch_bounds__(256) voxel_map_gather_kernel(const double *__restrict__ maps, const unsigned *__restrict__ empty,
                                                              int Nz, int Ny, int Nx, int nfeat, int nvox,
                                                              const int *__restrict__ voxels, unsigned allbits,
                                                              double *__restrict__ out, unsigned *__restrict__ empty_mask,
                                                              unsigned *__restrict__ any_nonempty) {
  const long long n = (long long)Nz * Ny * Nx;
  unsigned seen = 0;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += gridDim.x * blockDim.x) {
    const long long vi = ((long long)voxels[v] * Ny + voxels[(long long)nvox + v]) * Nx + voxels[2LL * nvox + v];
    for (int f = 0; f < nfeat; f++) out[(long long)f * nvox + v] = maps[(long long)f * n + vi];
    const unsigned em = empty[vi];
    empty_mask[v] = em;
    seen |= ~em & allbits;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) seen |= __shfl_xor(seen, o);
  if ((threadIdx.x & 63) == 0 && seen) atomicOr(any_nonempty, seen);
}

}  // namespace prad
