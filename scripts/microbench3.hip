// microbench3.hip -- round 6: what the walk step of sweep_fw_kernel gains from (a) more waves per SIMD (a table small enough for
// two workgroups per CU), (b) a conflict-free LDS address pattern, (c) two 16-bit run states per VGPR, and how ds_add_u32 scales
// with the number of active lanes and with bank conflicts.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/microbench3 scripts/microbench3.hip && scripts/microbench3
// Register-resident synthetic level words (level*4 bytes, 32 iid levels), 8 steps x 4 columns per iteration.  Reported: cycles per
// SIMD and wave-level voxel-step at the clock measured with s_memtime / s_memrealtime under the same load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32;

#define NGP 33
#define SLOTS 13
#define QA (NGP * 4)
#define PA4 (SLOTS * NGP)
#define TABLE (SLOTS * NGP * NGP * 4 + QA + 1024)   // 57.7 KB: two workgroups per CU fit

enum { M_VALU = 0, M_CUR, M_CUR_LATE, M_CUR_CF, M_CUR16, M_CUR16_NOLDS, M_LDS_RANDOM, M_LDS_CF, M_LDS_SAMEBANK, M_LDS_16LANES, M_LDS_32LANES,
       M_LDS_EVERY4, M_LDS_EVERY2, M_LDS_2X32, M_LDS_1LANE, M_CUR_2DS, M_PRIV, M_PRIV_NORUNS, M_TRIPLE, M_TRIPLE_NORUNS, M_COUNT };

template <int MODE>
__global__ void __launch_bounds__(1024) bench_kernel(const u32 *__restrict__ data, u32 *__restrict__ out, int iters, unsigned long long *clk) {
  extern __shared__ u32 lds[];
  for (int i = threadIdx.x; i < TABLE / 4; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32 d[8];
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = data[(size_t)((blockIdx.x & 255) * 1024 + threadIdx.x) * 8 + k];
  u32 one = 1;
  asm volatile("" : "+v"(one));
  u32 acc = 0;
  u32 pw = d[7];
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == M_VALU || MODE == M_CUR || MODE == M_CUR_LATE || MODE == M_CUR_CF || MODE == M_CUR_2DS) {
    u32 pl[4] = {QA + PA4 * 4, QA + PA4 * 8, QA + PA4 * 12, QA + PA4 * 16};
    const u32 pp4 = PA4, lq = QA;
    u32 lane4 = (u32)(lane & 31) * 4u;
    asm volatile("" : "+v"(lane4));
    const u32 keep = ~0x7cu;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u32 v = d[k];
        u32 t0;
#define CSTEP(J, T, LDSOP)                                                                                                          \
  asm volatile("v_cmpx_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                   \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" LDSOP \
               "v_mul_u32_u24_sdwa %[pl], %[pp], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
               "s_mov_b64 exec, -1\n\tv_add_u32 %[pl], %[lq], %[pl]\n\t"                                                           \
               : [pl] "+v"(pl[J]), [t] "=&v"(T) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [pp] "s"(pp4), [lq] "s"(lq) : "vcc", "memory")
#define CSTEP_LATE(J, T)                                                                                                            \
  asm volatile("v_cmpx_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                   \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"       \
               "v_mul_u32_u24_sdwa %[pl], %[pp], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
               "ds_add_u32 %[t], %[one]\n\t"                                                                                       \
               "s_mov_b64 exec, -1\n\tv_add_u32 %[pl], %[lq], %[pl]\n\t"                                                           \
               : [pl] "+v"(pl[J]), [t] "=&v"(T) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [pp] "s"(pp4), [lq] "s"(lq) : "vcc", "memory")
// conflict-free pattern (wrong bins, right cost): the address keeps its 128-byte row and takes the word of the lane
#define CSTEP_CF(J, T)                                                                                                              \
  asm volatile("v_cmpx_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                   \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"       \
               "v_and_or_b32 %[t], %[t], %[keep], %[l4]\n\t"                                                                       \
               "ds_add_u32 %[t], %[one]\n\t"                                                                                       \
               "v_mul_u32_u24_sdwa %[pl], %[pp], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
               "s_mov_b64 exec, -1\n\tv_add_u32 %[pl], %[lq], %[pl]\n\t"                                                           \
               : [pl] "+v"(pl[J]), [t] "=&v"(T) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [pp] "s"(pp4), [lq] "s"(lq), [keep] "s"(keep), [l4] "v"(lane4) : "vcc", "memory")
        if (MODE == M_CUR) {
          CSTEP(0, t0, "ds_add_u32 %[t], %[one]\n\t"); CSTEP(1, t0, "ds_add_u32 %[t], %[one]\n\t");
          CSTEP(2, t0, "ds_add_u32 %[t], %[one]\n\t"); CSTEP(3, t0, "ds_add_u32 %[t], %[one]\n\t");
        } else if (MODE == M_CUR_2DS) {   // the two-table step's LDS load: two atomics per step
          CSTEP(0, t0, "ds_add_u32 %[t], %[one]\n\tds_add_u32 %[t], %[one] offset:4\n\t"); CSTEP(1, t0, "ds_add_u32 %[t], %[one]\n\tds_add_u32 %[t], %[one] offset:4\n\t");
          CSTEP(2, t0, "ds_add_u32 %[t], %[one]\n\tds_add_u32 %[t], %[one] offset:4\n\t"); CSTEP(3, t0, "ds_add_u32 %[t], %[one]\n\tds_add_u32 %[t], %[one] offset:4\n\t");
        } else if (MODE == M_CUR_LATE) {
          CSTEP_LATE(0, t0); CSTEP_LATE(1, t0); CSTEP_LATE(2, t0); CSTEP_LATE(3, t0);
        } else if (MODE == M_CUR_CF) {
          CSTEP_CF(0, t0); CSTEP_CF(1, t0); CSTEP_CF(2, t0); CSTEP_CF(3, t0);
        } else {
          CSTEP(0, t0, ""); CSTEP(1, t0, ""); CSTEP(2, t0, ""); CSTEP(3, t0, "");
        }
        pw = v;
      }
    }
    acc = pl[0] + pl[1] + pl[2] + pl[3];
  } else if (MODE == M_CUR16 || MODE == M_CUR16_NOLDS) {
    // two run states per VGPR (16-bit LDS addresses: the table stays below 64 KB); the +Q of all lanes is one v_pk_add_u16 per two columns
    u32 pl[2] = {(QA + PA4 * 4) | ((QA + PA4 * 8) << 16), (QA + PA4 * 12) | ((QA + PA4 * 16) << 16)};
    const u32 pp4 = PA4;
    u32 lq2 = QA | (QA << 16);
    asm volatile("" : "+v"(lq2));
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u32 v = d[k];
        u32 t0;
#define HSTEP(J, R, H, LDSOP)                                                                                                       \
  asm volatile("v_cmpx_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                   \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_" #H " src1_sel:BYTE_" #J "\n\t" LDSOP \
               "v_mul_u32_u24_sdwa %[pl], %[pp], %[c] dst_sel:WORD_" #H " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
               "s_mov_b64 exec, -1\n\t"                                                                                            \
               : [pl] "+v"(pl[R]), [t] "=&v"(t0) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [pp] "s"(pp4) : "vcc", "memory")
        if (MODE == M_CUR16) {
          HSTEP(0, 0, 0, "ds_add_u32 %[t], %[one]\n\t"); HSTEP(1, 0, 1, "ds_add_u32 %[t], %[one]\n\t");
          asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(pl[0]) : "v"(lq2));
          HSTEP(2, 1, 0, "ds_add_u32 %[t], %[one]\n\t"); HSTEP(3, 1, 1, "ds_add_u32 %[t], %[one]\n\t");
          asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(pl[1]) : "v"(lq2));
        } else {
          HSTEP(0, 0, 0, ""); HSTEP(1, 0, 1, "");
          asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(pl[0]) : "v"(lq2));
          HSTEP(2, 1, 0, ""); HSTEP(3, 1, 1, "");
          asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(pl[1]) : "v"(lq2));
        }
        pw = v;
      }
    }
    acc = pl[0] + pl[1];
  } else if (MODE == M_PRIV || MODE == M_PRIV_NORUNS || MODE == M_TRIPLE || MODE == M_TRIPLE_NORUNS) {
    // VERDICT r5 item 1, the two formulations with fewer / conflict-free LDS atomics, as instruction mixes (bins are not checked):
    //  PRIV   (a) pairs into T[prev][cur][lane & 15] (16 words per bin: a 32-lane group is at most 2-way), run lengths as packed byte
    //         counters (units of 4), runs of length >= 2 into R[prev][len] behind a second exec mask (length 1 derived: SKIP1)
    //  TRIPLE (b) T[a][b][c] of three consecutive voxels, one atomic per TWO steps, plus the same run mechanism
    // NORUNS: the pair part alone (what a GLCM-only call would cost)
    constexpr bool RUNS = MODE == M_PRIV || MODE == M_TRIPLE;
    u32 xs[4], len4 = 0x04040404u;
#pragma unroll
    for (int j = 0; j < 4; j++) xs[j] = (u32)(j + 1) * 33u * 64u + (u32)(lane & 15) * 4u;
    u32 lanebase = (u32)(lane & 15) * 4u;
    const u32 rbase = 33u * 33u * 64u;   // run table behind the pair table (PRIV) -- TRIPLE keeps it inside its 144 KB
    asm volatile("" : "+v"(lanebase));
    const u32 k16 = 16, k33 = 33, rs4 = 13, k4444 = 0x04040404u, k1089 = 1089, k33b = 33;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u32 v = d[k];
        u32 t0, u0;
#define RUNPART(J, CLR)                                                                                                            \
  asm volatile("v_cmpx_lt_u32_sdwa vcc, %[four], %[len] src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"                                    \
               "v_mul_u32_u24_sdwa %[t], %[rs], %[p] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"   \
               "v_add_u32_sdwa %[t], %[t], %[len] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"       \
               "ds_add_u32 %[t], %[one] offset:4096\n\t"                                                                            \
               "s_mov_b64 exec, %[ev]\n\t"                                                                                         \
               "v_and_b32 %[len], %[clr], %[len]\n\t"                                                                              \
               : [len] "+v"(len4), [t] "=&v"(t0) : [p] "v"(pw), [one] "v"(one), [rs] "s"(rs4), [four] "s"(4u), [clr] "s"(CLR), [ev] "s"(evm) : "vcc", "memory")
#define PSTEP(J, CLR)                                                                                                               \
  {                                                                                                                                \
    unsigned long long evm;                                                                                                        \
    asm volatile("v_cmpx_ne_u32_sdwa %[ev], %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                               \
                 "v_mul_u32_u24_sdwa %[u], %[k16], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
                 "v_add_u32 %[t], %[xs], %[u]\n\t"                                                                                 \
                 "ds_add_u32 %[t], %[one]\n\t"                                                                                     \
                 "v_mad_u32_u24 %[xs], %[u], %[k33], %[lb]\n\t"                                                                    \
                 : [xs] "+v"(xs[J]), [t] "=&v"(t0), [u] "=&v"(u0), [ev] "=&s"(evm) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [k16] "s"(k16), [k33] "s"(k33), [lb] "v"(lanebase) : "memory"); \
    if (RUNS) RUNPART(J, CLR);                                                                                                     \
    asm volatile("s_mov_b64 exec, -1" ::: "memory");                                                                               \
  }
// triple: state = a*1089*4 (+ b*33*4 after the middle step); the middle step adds b*33*4, the closing step adds c*4, bumps, restarts from c
#define TMID(J)                                                                                                                    \
  asm volatile("v_mul_u32_u24_sdwa %[u], %[k33], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"  \
               "v_add_u32 %[xs], %[xs], %[u]\n\t" : [xs] "+v"(xs[J]), [u] "=&v"(u0) : [c] "v"(v), [k33] "s"(k33b) : "memory")
#define TEND(J)                                                                                                                    \
  asm volatile("v_add_u32_sdwa %[t], %[xs], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"       \
               "ds_add_u32 %[t], %[one]\n\t"                                                                                       \
               "v_mul_u32_u24_sdwa %[xs], %[k1089], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
               : [xs] "+v"(xs[J]), [t] "=&v"(t0) : [c] "v"(v), [one] "v"(one), [k1089] "s"(k1089) : "memory")
#define TRUN(J, CLR)                                                                                                               \
  {                                                                                                                                \
    unsigned long long evm;                                                                                                        \
    asm volatile("v_cmpx_ne_u32_sdwa %[ev], %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J : [ev] "=&s"(evm) : [c] "v"(v), [p] "v"(pw) : "memory"); \
    RUNPART(J, CLR);                                                                                                               \
    asm volatile("s_mov_b64 exec, -1" ::: "memory");                                                                               \
  }
        if (MODE == M_PRIV || MODE == M_PRIV_NORUNS) {
          PSTEP(0, 0xffffff00u); PSTEP(1, 0xffff00ffu); PSTEP(2, 0xff00ffffu); PSTEP(3, 0x00ffffffu);
        } else {
          if (k & 1) { TEND(0); TEND(1); TEND(2); TEND(3); } else { TMID(0); TMID(1); TMID(2); TMID(3); }
          if (RUNS) { TRUN(0, 0xffffff00u); TRUN(1, 0xffff00ffu); TRUN(2, 0xff00ffffu); TRUN(3, 0x00ffffffu); }
        }
        if (RUNS) asm volatile("v_add_u32 %0, %1, %0" : "+v"(len4) : "s"(k4444));
        pw = v;
      }
      (void)rbase;
    }
    acc = xs[0] + xs[1] + xs[2] + xs[3] + len4;
  } else {
    // ds_add_u32 alone, fixed addresses per lane: random bins of the fused table / one word per lane (conflict-free) / one bank
    u32 a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 bin = (((d[k] >> 2) & 31) + 1) * PA4 * 4 + (1 + ((d[k] >> 18) % 12)) * QA + (((d[k] >> 10) & 31) + 1) * 4;
      if (MODE == M_LDS_CF) a[k] = (bin & ~0x7cu) | ((u32)(lane & 31) * 4u);
      else if (MODE == M_LDS_SAMEBANK) a[k] = (bin & ~0x7cu);
      else a[k] = bin;
    }
    unsigned long long em = ~0ull;
    if (MODE == M_LDS_16LANES) em = 0xffffull;
    if (MODE == M_LDS_32LANES) em = 0xffffffffull;
    if (MODE == M_LDS_EVERY4) em = 0x1111111111111111ull;
    if (MODE == M_LDS_EVERY2) em = 0x5555555555555555ull;
    if (MODE == M_LDS_1LANE) em = 0x8000000000000000ull;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (MODE == M_LDS_2X32)   // the same 64 lane-atomics as two half-wave instructions
            asm volatile("s_mov_b64 exec, 0xffffffff\n\tds_add_u32 %0, %1\n\ts_not_b64 exec, exec\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(a[k]), "v"(one) : "memory", "scc");
          else
            asm volatile("s_mov_b64 exec, %2\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(a[k]), "v"(one), "s"(em) : "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = r1 - r0;
  }
  __syncthreads();
  u32 s = acc;
  for (int i = threadIdx.x; i < TABLE / 4; i += blockDim.x) s += lds[i];
  out[(blockIdx.x & 255) * 1024 + threadIdx.x] = s;
}

static const char *NAMES[M_COUNT] = {
    "step, no ds_add (VALU + SALU only)", "step as built (4 VALU + s_mov + ds_add)", "step, ds_add behind the fresh state",
    "step + v_and_or: conflict-free addresses", "two 16-bit states per VGPR (3.5 VALU)", "two 16-bit states per VGPR, no ds_add",
    "ds_add alone: random bins of the fused table", "ds_add alone: conflict-free (word = lane)", "ds_add alone: all lanes on one bank",
    "ds_add alone: lanes 0-15 active", "ds_add alone: lanes 0-31 active", "ds_add alone: every 4th lane active",
    "ds_add alone: every 2nd lane active", "ds_add alone: two half-wave instructions", "ds_add alone: one lane active",
    "step with TWO ds_add", "(a) pairs T[x][c][lane&15] + packed lengths + runs >= 2", "(a) pair part alone",
    "(b) triple table, one atomic per two steps + runs >= 2", "(b) triple part alone"};

template <int MODE>
void run(const u32 *ddata, u32 *dout, unsigned long long *dclk, int blocks, int threads, int iters) {
  const size_t shm = (MODE == M_PRIV || MODE == M_PRIV_NORUNS || MODE == M_TRIPLE || MODE == M_TRIPLE_NORUNS) ? 150 * 1024 : TABLE;
  CK(hipFuncSetAttribute((const void *)bench_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  bench_kernel<MODE><<<blocks, threads, shm>>>(ddata, dout, iters / 4, dclk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  bench_kernel<MODE><<<blocks, threads, shm>>>(ddata, dout, iters, dclk);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  unsigned long long h[2];
  CK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;   // s_memrealtime counts at 100 MHz
  const double waves_per_simd = (double)blocks * threads / 64.0 / 1024.0;
  const double cyc = ms * 1e-3 * ghz * 1e9 / (32.0 * iters * waves_per_simd);
  printf("%-48s %2.0f waves/SIMD  %8.3f ms  %5.2f GHz  %7.2f cyc per SIMD and step (%6.2f per CU)\n", NAMES[MODE], waves_per_simd, ms, ghz, cyc, cyc / 4);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  const size_t n = (size_t)256 * 1024 * 8;
  std::vector<u32> h(n);
  u32 st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (size_t i = 0; i < n; i++) {
    u32 w = 0;
    for (int b = 0; b < 4; b++) w |= (((rnd() % 32) + 1) * 4) << (8 * b);
    h[i] = w;
  }
  u32 *du, *dout;
  unsigned long long *dclk;
  CK(hipMalloc(&du, n * 4)); CK(hipMalloc(&dout, 256 * 1024 * 4)); CK(hipMalloc(&dclk, 16));
  CK(hipMemcpy(du, h.data(), n * 4, hipMemcpyHostToDevice));
  struct Cfg { int blocks, threads; };
  const Cfg cfgs[] = {{256, 512}, {256, 1024}, {512, 640}, {512, 768}, {512, 1024}};
  for (const Cfg &c : cfgs) {
    printf("--- %d workgroups x %d threads ---\n", c.blocks, c.threads);
    run<M_VALU>(du, dout, dclk, c.blocks, c.threads, iters);
    run<M_CUR>(du, dout, dclk, c.blocks, c.threads, iters);
    run<M_CUR_LATE>(du, dout, dclk, c.blocks, c.threads, iters);
    run<M_CUR_CF>(du, dout, dclk, c.blocks, c.threads, iters);
    run<M_CUR16_NOLDS>(du, dout, dclk, c.blocks, c.threads, iters);
    run<M_CUR16>(du, dout, dclk, c.blocks, c.threads, iters);
    run<M_CUR_2DS>(du, dout, dclk, c.blocks, c.threads, iters);
    if (c.blocks == 256) {   // (their tables take the whole LDS: one workgroup per CU)
      run<M_PRIV>(du, dout, dclk, c.blocks, c.threads, iters);
      run<M_PRIV_NORUNS>(du, dout, dclk, c.blocks, c.threads, iters);
      run<M_TRIPLE>(du, dout, dclk, c.blocks, c.threads, iters);
      run<M_TRIPLE_NORUNS>(du, dout, dclk, c.blocks, c.threads, iters);
    }
    run<M_LDS_RANDOM>(du, dout, dclk, c.blocks, c.threads, iters);
    run<M_LDS_CF>(du, dout, dclk, c.blocks, c.threads, iters);
    if (c.threads == 1024) {
      run<M_LDS_SAMEBANK>(du, dout, dclk, c.blocks, c.threads, iters / 8);
      run<M_LDS_16LANES>(du, dout, dclk, c.blocks, c.threads, iters);
      run<M_LDS_32LANES>(du, dout, dclk, c.blocks, c.threads, iters);
      run<M_LDS_EVERY4>(du, dout, dclk, c.blocks, c.threads, iters);
      run<M_LDS_EVERY2>(du, dout, dclk, c.blocks, c.threads, iters);
      run<M_LDS_2X32>(du, dout, dclk, c.blocks, c.threads, iters);
      run<M_LDS_1LANE>(du, dout, dclk, c.blocks, c.threads, iters);
    }
  }
  return 0;
}
