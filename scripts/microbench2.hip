// microbench2.hip -- round 4: what keeps the walk step's LDS atomic from hiding under its VALU work, and a
// three-VALU step (fresh state by v_dot4_u32_u8, run length implicit in the ds_add offset field).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/microbench2 scripts/microbench2.hip && scripts/microbench2
// One 16-wave workgroup per CU, register-resident synthetic level words (level*4 bytes), 8 steps x 4 columns per
// iteration.  Reported: cycles per SIMD and wave-level voxel-step at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32;

#define NGP 33
#define SLOTS 13
// layout A (kernel of rounds 2-4): [prev][len][cur]: Q = 132 B per length slot, P = SLOTS * Q per level
#define QA (NGP * 4)
#define PA4 (SLOTS * NGP)
// layout B: [plane][prev][cur], plane = SLOTS - len: PP = 132 B per level, QQ = 33 * 132 B per plane
#define PPB (NGP * 4)
#define QQB (NGP * PPB)
#define TABLE_B (SLOTS * NGP * NGP * 4 + 1024)

enum { M_CUR = 0, M_CUR_NOLDS, M_DOT, M_DOT_NOLDS, M_DOT_LATE, M_DOT_SEL, M_MIX_INDEP, M_MIX_ADDR, M_MIX_VALU_ONLY, M_MIX_LDS_ONLY, M_DOT4_RATE,
       M_DOT_T4, M_CUR_T4, M_WRAP, M_NEW, M_NEW_NOLDS, M_SDWA_RATE, M_MAD_RATE, M_ADD3_RATE, M_CMPX_RATE, M_CMPX_SDWA_RATE };

template <int MODE>
__global__ void __launch_bounds__(1024) bench_kernel(const u32 *__restrict__ data, u32 *__restrict__ out, int iters) {
  extern __shared__ u32 lds[];
  for (int i = threadIdx.x; i < TABLE_B / 4; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32 d[8];
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = data[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 8 + k];
  u32 one = 1;
  asm volatile("" : "+v"(one));
  u32 acc = 0;
  u32 pw = d[7];
  if (MODE == M_WRAP) {
    // does VGPR address + offset wrap at 32 bits?  address = -4096 + lane*4 (wrapped), offset 4096 -> lds[lane]
    u32 a = (u32)(-4096) + 4u * (u32)lane;
    asm volatile("ds_add_u32 %0, %1 offset:4096\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(one) : "memory");
    __syncthreads();
    acc = lds[lane];
  } else if (MODE == M_MIX_INDEP || MODE == M_MIX_VALU_ONLY || MODE == M_MIX_LDS_ONLY || MODE == M_MIX_ADDR) {
    // 4 independent VALU + 1 ds_add per unit, nothing depends on anything (INDEP), or the ds_add's address comes out of one
    // of the four VALU (ADDR)
    u32 a[8], x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      a[k] = (((d[k] >> 2) & 31) * 33 + ((d[k] >> 10) & 31) + 33 * 33 * (k % 3)) * 4;
      x[k] = d[k];
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (MODE == M_MIX_INDEP)
            asm volatile("v_add_u32 %0, %2, %0\n\tv_add_u32 %1, %2, %1\n\tds_add_u32 %3, %4\n\tv_add_u32 %0, %2, %0\n\tv_add_u32 %1, %2, %1"
                         : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]), "v"(a[k]), "v"(one) : "memory");
          else if (MODE == M_MIX_VALU_ONLY)
            asm volatile("v_add_u32 %0, %2, %0\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %0, %2, %0\n\tv_add_u32 %1, %2, %1"
                         : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]), "v"(a[k]), "v"(one) : "memory");
          else if (MODE == M_MIX_LDS_ONLY)
            asm volatile("ds_add_u32 %3, %4" : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]), "v"(a[k]), "v"(one) : "memory");
          else {
            u32 t;
            asm volatile("v_add_u32 %0, %2, %0\n\tv_xor_b32 %5, %3, %6\n\tds_add_u32 %5, %4\n\tv_add_u32 %0, %2, %0\n\tv_add_u32 %1, %2, %1"
                         : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]), "v"(a[k]), "v"(one), "v"(t), "s"(r * 4) : "memory");
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += x[k];
  } else if (MODE == M_DOT4_RATE) {
    u32 x[8];
    u32 w = 33;
    asm volatile("" : "+v"(w));
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = d[k];
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
          asm volatile("v_dot4_u32_u8 %0, %1, %2, %0\n\tv_dot4_u32_u8 %3, %1, %2, %3\n\tv_dot4_u32_u8 %0, %1, %2, %0\n\tv_dot4_u32_u8 %3, %1, %2, %3"
                       : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]), "v"(w));
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += x[k];
  } else if (MODE == M_SDWA_RATE || MODE == M_MAD_RATE || MODE == M_ADD3_RATE || MODE == M_CMPX_RATE || MODE == M_CMPX_SDWA_RATE) {
    u32 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = d[k];
    const u32 pp = 429;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (MODE == M_SDWA_RATE)
            asm volatile("v_add_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
                         "v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                         "v_add_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
                         : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]));
          else if (MODE == M_MAD_RATE)
            asm volatile("v_mad_u32_u24 %0, %2, %3, %0\n\tv_mad_u32_u24 %1, %2, %3, %1\n\tv_mad_u32_u24 %0, %2, %3, %0\n\tv_mad_u32_u24 %1, %2, %3, %1"
                         : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]), "s"(pp));
          else if (MODE == M_ADD3_RATE)
            asm volatile("v_add3_u32 %0, %2, %3, %0\n\tv_add3_u32 %1, %2, %3, %1\n\tv_add3_u32 %0, %2, %3, %0\n\tv_add3_u32 %1, %2, %3, %1"
                         : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]), "s"(pp));
          else if (MODE == M_CMPX_RATE)
            asm volatile("v_cmpx_ne_u32 vcc, %2, %0\n\ts_mov_b64 exec, -1\n\tv_cmpx_ne_u32 vcc, %2, %1\n\ts_mov_b64 exec, -1\n\t"
                         "v_cmpx_ne_u32 vcc, %2, %0\n\ts_mov_b64 exec, -1\n\tv_cmpx_ne_u32 vcc, %2, %1\n\ts_mov_b64 exec, -1"
                         : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]) : "vcc");
          else
            asm volatile("v_cmpx_ne_u32_sdwa vcc, %2, %0 src0_sel:BYTE_0 src1_sel:BYTE_0\n\ts_mov_b64 exec, -1\n\tv_cmpx_ne_u32_sdwa vcc, %2, %1 src0_sel:BYTE_1 src1_sel:BYTE_1\n\ts_mov_b64 exec, -1\n\t"
                         "v_cmpx_ne_u32_sdwa vcc, %2, %0 src0_sel:BYTE_2 src1_sel:BYTE_2\n\ts_mov_b64 exec, -1\n\tv_cmpx_ne_u32_sdwa vcc, %2, %1 src0_sel:BYTE_3 src1_sel:BYTE_3\n\ts_mov_b64 exec, -1"
                         : "+v"(x[k]), "+v"(x[(k + 3) & 7]) : "v"(d[(k + 1) & 7]) : "vcc");
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += x[k];
  } else if (MODE == M_NEW || MODE == M_NEW_NOLDS) {
    // unpacked levels, run length implicit inside the group: state = level*P - k0*Q, address = state + cur + k*Q (v_add3),
    // fresh state = v_mad_u32_u24(cur, P4, -k*Q); one v_add per column and group brings the states back to level*P + len*Q
    u32 pl[4] = {QA + PA4 * 4, QA + PA4 * 8, QA + PA4 * 12, QA + PA4 * 16};
    const u32 pp4 = PA4;
    u32 xj[4];
#pragma unroll
    for (int j = 0; j < 4; j++) xj[j] = (pw >> (8 * j)) & 0xffu;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u32 v = d[k];
        u32 cj[4];
        asm volatile("v_and_b32 %0, 0xff, %4\n\tv_bfe_u32 %1, %4, 8, 8\n\tv_bfe_u32 %2, %4, 16, 8\n\tv_lshrrev_b32 %3, 24, %4"
                     : "=&v"(cj[0]), "=&v"(cj[1]), "=&v"(cj[2]), "=&v"(cj[3]) : "v"(v));
        const u32 kq = (u32)k * QA, nkq = 0u - (u32)k * QA;
        u32 t;
#define NSTEP(J, LDSOP)                                                                                             \
  asm volatile("v_cmpx_ne_u32 vcc, %[c], %[x]\n\t"                                                                  \
               "v_add3_u32 %[t], %[pl], %[c], %[kq]\n\t" LDSOP                                                      \
               "v_mad_u32_u24 %[pl], %[c], %[pp], %[nkq]\n\t"                                                       \
               "s_mov_b64 exec, -1\n\t"                                                                             \
               : [pl] "+v"(pl[J]), [t] "=&v"(t) : [c] "v"(cj[J]), [x] "v"(xj[J]), [one] "v"(one), [pp] "v"(ppv), [kq] "s"(kq), [nkq] "s"(nkq) : "vcc", "memory")
        u32 ppv = pp4;
        asm volatile("" : "+v"(ppv));
        if (MODE == M_NEW) { NSTEP(0, "ds_add_u32 %[t], %[one]\n\t"); NSTEP(1, "ds_add_u32 %[t], %[one]\n\t"); NSTEP(2, "ds_add_u32 %[t], %[one]\n\t"); NSTEP(3, "ds_add_u32 %[t], %[one]\n\t"); }
        else { NSTEP(0, ""); NSTEP(1, ""); NSTEP(2, ""); NSTEP(3, ""); }
#pragma unroll
        for (int j = 0; j < 4; j++) xj[j] = cj[j];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(pl[j]) : "s"(8u * QA));
    }
    acc = pl[0] + pl[1] + pl[2] + pl[3];
  } else if (MODE == M_CUR || MODE == M_CUR_NOLDS || MODE == M_CUR_T4) {
    u32 pl[4] = {QA + PA4 * 4, QA + PA4 * 8, QA + PA4 * 12, QA + PA4 * 16};
    const u32 pp4 = PA4, lq = QA;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u32 v = d[k];
        u32 t0, t1, t2, t3;
#define CSTEP(J, T, LDSOP)                                                                                                          \
  asm volatile("v_cmpx_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                   \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" LDSOP \
               "v_mul_u32_u24_sdwa %[pl], %[pp], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
               "s_mov_b64 exec, -1\n\tv_add_u32 %[pl], %[lq], %[pl]\n\t"                                                           \
               : [pl] "+v"(pl[J]), [t] "=&v"(T) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [pp] "s"(pp4), [lq] "s"(lq) : "vcc", "memory")
        if (MODE == M_CUR) {
          CSTEP(0, t0, "ds_add_u32 %[t], %[one]\n\t"); CSTEP(1, t0, "ds_add_u32 %[t], %[one]\n\t");
          CSTEP(2, t0, "ds_add_u32 %[t], %[one]\n\t"); CSTEP(3, t0, "ds_add_u32 %[t], %[one]\n\t");
        } else if (MODE == M_CUR_T4) {
          CSTEP(0, t0, "ds_add_u32 %[t], %[one]\n\t"); CSTEP(1, t1, "ds_add_u32 %[t], %[one]\n\t");
          CSTEP(2, t2, "ds_add_u32 %[t], %[one]\n\t"); CSTEP(3, t3, "ds_add_u32 %[t], %[one]\n\t");
          asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3));
        } else {
          CSTEP(0, t0, ""); CSTEP(1, t0, ""); CSTEP(2, t0, ""); CSTEP(3, t0, "");
        }
        pw = v;
      }
    }
    acc = pl[0] + pl[1] + pl[2] + pl[3];
  } else {
    // layout B.  canonical state at a group start: prev*PP + (SLOTS - len - 7) * QQ; a run of length 1: plane SLOTS - 8
    u32 pl[4];
#pragma unroll
    for (int j = 0; j < 4; j++) pl[j] = 4 * (j + 1) * 33 + (SLOTS - 8) * QQB;
    u32 W[4] = {33u, 33u << 8, 33u << 16, 33u << 24};
    asm volatile("" : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]));
    u32 dummy = TABLE_B - 512 + 4 * lane;
    asm volatile("" : "+v"(dummy));
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u32 v = d[k];
        const u32 Kk = (u32)(k + SLOTS - 7) * QQB;   // fresh state of a run that begins at step k
        u32 t0, t1, t2, t3;
#define DSTEP(J, T, OFF, LDSOP)                                                                                                     \
  asm volatile("v_cmpx_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                   \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" LDSOP \
               "v_dot4_u32_u8 %[pl], %[c], %[w], %[kk]\n\t"                                                                        \
               "s_mov_b64 exec, -1\n\t"                                                                                            \
               : [pl] "+v"(pl[J]), [t] "=&v"(T) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [w] "v"(W[J]), [kk] "s"(Kk), [off] "n"(OFF) : "vcc", "memory")
#define DSTEP_LATE(J, T, OFF)                                                                                                       \
  asm volatile("v_cmpx_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                   \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"       \
               "v_dot4_u32_u8 %[pl], %[c], %[w], %[kk]\n\t"                                                                        \
               "ds_add_u32 %[t], %[one] offset:%[off]\n\t"                                                                        \
               "s_mov_b64 exec, -1\n\t"                                                                                            \
               : [pl] "+v"(pl[J]), [t] "=&v"(T) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [w] "v"(W[J]), [kk] "s"(Kk), [off] "n"(OFF) : "vcc", "memory")
#define DSTEP_SEL(J, T, OFF)                                                                                                        \
  asm volatile("v_cmp_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t"                                    \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t"       \
               "v_cndmask_b32 %[t], %[dm], %[t], vcc\n\t"                                                                          \
               "ds_add_u32 %[t], %[one] offset:%[off]\n\t"                                                                        \
               "v_dot4_u32_u8 %[f], %[c], %[w], %[kk]\n\t"                                                                         \
               "v_cndmask_b32 %[pl], %[pl], %[f], vcc\n\t"                                                                         \
               : [pl] "+v"(pl[J]), [t] "=&v"(T), [f] "=&v"(fr) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [w] "v"(W[J]), [kk] "s"(Kk), [off] "n"(OFF), [dm] "v"(dummy) : "vcc", "memory")
        const int OFFK = 0;
        (void)OFFK;
        u32 fr;
        (void)fr;
#define ALL4(M, TA, TB, TC, TD, ...)                                                                                                \
  switch (k) {                                                                                                                      \
    case 0: M(0, TA, 7 * QQB, ##__VA_ARGS__); M(1, TB, 7 * QQB, ##__VA_ARGS__); M(2, TC, 7 * QQB, ##__VA_ARGS__); M(3, TD, 7 * QQB, ##__VA_ARGS__); break; \
    case 1: M(0, TA, 6 * QQB, ##__VA_ARGS__); M(1, TB, 6 * QQB, ##__VA_ARGS__); M(2, TC, 6 * QQB, ##__VA_ARGS__); M(3, TD, 6 * QQB, ##__VA_ARGS__); break; \
    case 2: M(0, TA, 5 * QQB, ##__VA_ARGS__); M(1, TB, 5 * QQB, ##__VA_ARGS__); M(2, TC, 5 * QQB, ##__VA_ARGS__); M(3, TD, 5 * QQB, ##__VA_ARGS__); break; \
    case 3: M(0, TA, 4 * QQB, ##__VA_ARGS__); M(1, TB, 4 * QQB, ##__VA_ARGS__); M(2, TC, 4 * QQB, ##__VA_ARGS__); M(3, TD, 4 * QQB, ##__VA_ARGS__); break; \
    case 4: M(0, TA, 3 * QQB, ##__VA_ARGS__); M(1, TB, 3 * QQB, ##__VA_ARGS__); M(2, TC, 3 * QQB, ##__VA_ARGS__); M(3, TD, 3 * QQB, ##__VA_ARGS__); break; \
    case 5: M(0, TA, 2 * QQB, ##__VA_ARGS__); M(1, TB, 2 * QQB, ##__VA_ARGS__); M(2, TC, 2 * QQB, ##__VA_ARGS__); M(3, TD, 2 * QQB, ##__VA_ARGS__); break; \
    case 6: M(0, TA, 1 * QQB, ##__VA_ARGS__); M(1, TB, 1 * QQB, ##__VA_ARGS__); M(2, TC, 1 * QQB, ##__VA_ARGS__); M(3, TD, 1 * QQB, ##__VA_ARGS__); break; \
    default: M(0, TA, 0, ##__VA_ARGS__); M(1, TB, 0, ##__VA_ARGS__); M(2, TC, 0, ##__VA_ARGS__); M(3, TD, 0, ##__VA_ARGS__); break; \
  }
        if (MODE == M_DOT) { ALL4(DSTEP, t0, t0, t0, t0, "ds_add_u32 %[t], %[one] offset:%[off]\n\t") }
        else if (MODE == M_DOT_T4) { ALL4(DSTEP, t0, t1, t2, t3, "ds_add_u32 %[t], %[one] offset:%[off]\n\t") asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3)); }
        else if (MODE == M_DOT_NOLDS) { ALL4(DSTEP, t0, t0, t0, t0, "") }
        else if (MODE == M_DOT_LATE) { ALL4(DSTEP_LATE, t0, t1, t2, t3) asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3)); }
        else { ALL4(DSTEP_SEL, t0, t1, t2, t3) asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3)); }
        pw = v;
      }
      // group end: every line's plane moves 8 up (one VALU per column and group)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        pl[j] -= 8 * QQB;
        // (synthetic data: keep the state inside the table -- a real walk leaves long runs to the checked path)
        pl[j] = (int)pl[j] < 0 ? pl[j] + 8 * QQB : pl[j];
      }
    }
    acc = pl[0] + pl[1] + pl[2] + pl[3];
  }
  __syncthreads();
  u32 s = acc;
  for (int i = threadIdx.x; i < TABLE_B / 4; i += blockDim.x) s += lds[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(const char *name, const u32 *ddata, u32 *dout, int threads, int iters, double units_per_iter) {
  const int blocks = 256;
  const size_t shm = TABLE_B;
  CK(hipFuncSetAttribute((const void *)bench_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  bench_kernel<MODE><<<blocks, threads, shm>>>(ddata, dout, iters / 8);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  bench_kernel<MODE><<<blocks, threads, shm>>>(ddata, dout, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double waves_per_cu = threads / 64.0;
  const double cyc = ms * 1e-3 * 2.4e9 / (units_per_iter * iters * waves_per_cu) * 4;
  printf("%-44s thr=%4d  %8.3f ms  %7.3f cyc per SIMD and unit\n", name, threads, ms, cyc);
  fflush(stdout);
  return ms;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
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
  CK(hipMalloc(&du, n * 4)); CK(hipMalloc(&dout, 256 * 1024 * 4));
  CK(hipMemcpy(du, h.data(), n * 4, hipMemcpyHostToDevice));
  {
    CK(hipFuncSetAttribute((const void *)bench_kernel<M_WRAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TABLE_B));
    bench_kernel<M_WRAP><<<1, 64, TABLE_B>>>(du, dout, 1);
    u32 r[64];
    CK(hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost));
    printf("ds address wrap (VGPR -4096 + lane*4, offset 4096): lds[lane] = %u %u %u ... -> %s\n", r[0], r[1], r[63], r[0] >= 1 ? "wraps at 32 bits" : "does NOT wrap");
  }
  for (int threads : {1024, 512}) {
    printf("--- %d threads per CU (%d waves/SIMD) ---\n", threads, threads / 256);
    run<M_MIX_VALU_ONLY>("4 indep v_add", du, dout, threads, iters, 32);
    run<M_MIX_LDS_ONLY>("1 ds_add (fixed random addresses)", du, dout, threads, iters, 32);
    run<M_MIX_INDEP>("4 indep v_add + 1 ds_add, no dependence", du, dout, threads, iters, 32);
    run<M_MIX_ADDR>("3 v_add + v_xor -> ds_add address", du, dout, threads, iters, 32);
    run<M_DOT4_RATE>("4 v_dot4_u32_u8", du, dout, threads, iters, 32);
    run<M_CUR_NOLDS>("current step, no ds_add", du, dout, threads, iters, 32);
    run<M_CUR>("current step (4 VALU + s_mov + ds_add)", du, dout, threads, iters, 32);
    run<M_CUR_T4>("current step, 4 address temporaries", du, dout, threads, iters, 32);
    run<M_DOT_NOLDS>("dot4 step, no ds_add", du, dout, threads, iters, 32);
    run<M_DOT>("dot4 step (3 VALU + s_mov + ds_add)", du, dout, threads, iters, 32);
    run<M_DOT_T4>("dot4 step, 4 address temporaries", du, dout, threads, iters, 32);
    run<M_DOT_LATE>("dot4 step, ds_add behind the dot4", du, dout, threads, iters, 32);
    run<M_SDWA_RATE>("4 v_add_u32_sdwa", du, dout, threads, iters, 32);
    run<M_MAD_RATE>("4 v_mad_u32_u24", du, dout, threads, iters, 32);
    run<M_ADD3_RATE>("4 v_add3_u32", du, dout, threads, iters, 32);
    run<M_CMPX_RATE>("4 (v_cmpx_ne_u32 + s_mov exec)", du, dout, threads, iters, 32);
    run<M_CMPX_SDWA_RATE>("4 (v_cmpx_ne_u32_sdwa + s_mov exec)", du, dout, threads, iters, 32);
    run<M_NEW_NOLDS>("new step (unpack + 3 VALU), no ds_add", du, dout, threads, iters, 32);
    run<M_NEW>("new step (unpack + 3 VALU + s_mov + ds_add)", du, dout, threads, iters, 32);
  }
  return 0;
}
