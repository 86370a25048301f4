// microbench.hip -- instruction-level floors on gfx950 for the sweep redesign (round 2).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/microbench scripts/microbench.hip && scripts/microbench
// Every kernel runs ITERS iterations of an 8-step x 4-line body on register-resident synthetic level words
// (level*4 bytes, as in the packed volume), one workgroup per CU.  Reported: ns per wave-level voxel-step
// instruction group and the equivalent cycles per CU at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u32;
typedef __attribute__((address_space(3))) u32 lds_u32;

// table geometry of the planned kernel: len-major, 8 length slots, 33 prev rows, 3 angles, 33 cur columns
#define NGP 33
#define Q_B (NGP * 4)            // bytes per (prev, angle) row
#define PP_B (3 * Q_B)           // bytes per prev
#define LQ_B (NGP * PP_B)        // bytes per length slot
#define SLOTS 8
#define TABLE_B (SLOTS * LQ_B)   // 104544 B

__device__ __forceinline__ void lds_add(u32 addr, u32 v) {
  __hip_atomic_fetch_add((lds_u32 *)(size_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int sel(bool c, int a, int b) {
  asm volatile("" : "+v"(a), "+v"(b));
  return c ? a : b;
}

enum { M_STEP6 = 0, M_STEP6_NOLDS, M_STEP4X, M_STEP4X_NOLDS, M_STEP5X_CHECK, M_LDS_ONLY, M_VALU_ADD, M_VALU_CNDMASK, M_VALU_SDWA, M_SALU_MIX, M_STEP4X_NOSALU };

template <int MODE>
__global__ void __launch_bounds__(1024) bench_kernel(const u32 *__restrict__ data, u32 *__restrict__ out, int iters, int addr_mode) {
  extern __shared__ u32 lds[];
  for (int i = threadIdx.x; i < TABLE_B / 4 + 64; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32 d[8];
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = data[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 8 + k];
  const u32 dummy = TABLE_B + 4 * lane;
  u32 pl[4] = {LQ_B, LQ_B, LQ_B, LQ_B};
  u32 one = 1;
  u32 acc = 0;
  const u32 pp4 = PP_B / 4, lq = LQ_B;
  u32 pw = d[7];
  if (MODE == M_LDS_ONLY) {
    // precomputed addresses: addr_mode 0 random bins, 1 conflict-free (lane*4), 2 all lanes same address, 3 random with half the lanes masked
    u32 a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 prev = (d[k] >> 2) & 31, cur = (d[k] >> 10) & 31;
      a[k] = addr_mode == 1 ? (u32)(lane * 4 + k * 256) : addr_mode == 2 ? (u32)(k * 4) : LQ_B + prev * PP_B + cur * 4 + (k % 3) * Q_B;
    }
    const bool act = addr_mode != 3 || ((d[0] >> 3) & 1);
    for (int it = 0; it < iters; it++) {
      if (act) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
          for (int k = 0; k < 8; k++) lds_add(a[k], one);
        }
      }
    }
  } else if (MODE == M_VALU_ADD) {
    u32 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = d[k];
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 24; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[k]) : "v"(d[(k + 1) & 7]));
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += x[k];
  } else if (MODE == M_VALU_CNDMASK) {
    u32 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = d[k];
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 12; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
          asm volatile("v_cmp_ne_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:BYTE_2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(d[(k + 1) & 7]), "v"(d[(k + 2) & 7]) : "vcc");
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += x[k];
  } else if (MODE == M_VALU_SDWA) {
    u32 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = d[k];
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 12; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
          asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
                       "v_mul_u32_u24_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                       : "+v"(x[k]) : "v"(d[(k + 1) & 7]), "s"(pp4));
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += x[k];
  } else if (MODE == M_SALU_MIX) {
    // 4 VALU + 1 SALU (s_mov exec) per unit, no LDS, no cmpx: does the scalar op share the wave's issue slot?
    u32 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = d[k];
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
          asm volatile("v_add_u32 %0, %1, %0\n\tv_add_u32 %0, %1, %0\n\tv_add_u32 %0, %1, %0\n\ts_mov_b64 exec, -1\n\tv_add_u32 %0, %1, %0" : "+v"(x[k]) : "v"(d[(k + 1) & 7]));
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += x[k];
  } else {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u32 v = d[k];
        if (MODE == M_STEP6 || MODE == M_STEP6_NOLDS) {
          bool chg[4];
          int addr[4], fresh[4], grown[4];
#pragma unroll
          for (int j = 0; j < 4; j++) chg[j] = __builtin_amdgcn_ubfe(v, 8 * j, 8) != __builtin_amdgcn_ubfe(pw, 8 * j, 8);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            addr[j] = pl[j] + (int)__builtin_amdgcn_ubfe(v, 8 * j, 8);
            fresh[j] = (int)__umul24(__builtin_amdgcn_ubfe(v, 8 * j, 8), pp4) + lq;
            grown[j] = pl[j] + lq;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int a = sel(chg[j], addr[j], (int)dummy);
            if (MODE == M_STEP6) lds_add((u32)a, one); else asm volatile("" ::"v"(a));
          }
#pragma unroll
          for (int j = 0; j < 4; j++) pl[j] = (u32)sel(chg[j], fresh[j], grown[j]) & 0xffffu;   // (& keeps the synthetic address in range: +1 VALU the real kernel does not have)
        } else {
#define XSTEP(J, LDSOP, EXTRA, RESTORE)                                                                                             \
  asm volatile("v_cmpx_ne_u32_sdwa vcc, %[c], %[p] src0_sel:BYTE_" #J " src1_sel:BYTE_" #J "\n\t" EXTRA                          \
               "v_add_u32_sdwa %[t], %[pl], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" LDSOP \
               "v_mul_u32_u24_sdwa %[pl], %[pp], %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #J "\n\t" \
               RESTORE "v_add_u32 %[pl], %[lq], %[pl]\n\t"                                                                         \
               : [pl] "+v"(pl[J]), [t] "=&v"(tmp) : [c] "v"(v), [p] "v"(pw), [one] "v"(one), [pp] "s"(pp4), [lq] "s"(lq), [lim] "s"((u32)(7 * LQ_B)) : "vcc", "memory")
          u32 tmp;
          if (MODE == M_STEP4X) {
            XSTEP(0, "ds_add_u32 %[t], %[one]\n\t", "", "s_mov_b64 exec, -1\n\t"); XSTEP(1, "ds_add_u32 %[t], %[one] offset:132\n\t", "", "s_mov_b64 exec, -1\n\t");
            XSTEP(2, "ds_add_u32 %[t], %[one] offset:264\n\t", "", "s_mov_b64 exec, -1\n\t"); XSTEP(3, "ds_add_u32 %[t], %[one]\n\t", "", "s_mov_b64 exec, -1\n\t");
          } else if (MODE == M_STEP4X_NOLDS) {
            XSTEP(0, "", "", "s_mov_b64 exec, -1\n\t"); XSTEP(1, "", "", "s_mov_b64 exec, -1\n\t"); XSTEP(2, "", "", "s_mov_b64 exec, -1\n\t"); XSTEP(3, "", "", "s_mov_b64 exec, -1\n\t");
          } else if (MODE == M_STEP4X_NOSALU) {   // (wrong results: exec stays masked; issue-cost probe only)
            XSTEP(0, "ds_add_u32 %[t], %[one]\n\t", "", ""); XSTEP(1, "ds_add_u32 %[t], %[one] offset:132\n\t", "", "");
            XSTEP(2, "ds_add_u32 %[t], %[one] offset:264\n\t", "", ""); XSTEP(3, "ds_add_u32 %[t], %[one]\n\t", "", "s_mov_b64 exec, -1\n\t");
          } else {  // M_STEP5X_CHECK: + long-run test and (never taken) branch
#define CHK "v_cmp_ge_u32 vcc, %[pl], %[lim]\n\ts_cbranch_vccnz 1f\n\t1:\n\t"
            XSTEP(0, "ds_add_u32 %[t], %[one]\n\t", CHK, "s_mov_b64 exec, -1\n\t"); XSTEP(1, "ds_add_u32 %[t], %[one] offset:132\n\t", CHK, "s_mov_b64 exec, -1\n\t");
            XSTEP(2, "ds_add_u32 %[t], %[one] offset:264\n\t", CHK, "s_mov_b64 exec, -1\n\t"); XSTEP(3, "ds_add_u32 %[t], %[one]\n\t", CHK, "s_mov_b64 exec, -1\n\t");
          }
#pragma unroll
          for (int j = 0; j < 4; j++) pl[j] &= 0xffffu;  // same extra op as the STEP6 flavour
        }
        pw = v;
      }
    }
    acc = pl[0] + pl[1] + pl[2] + pl[3];
  }
  __syncthreads();
  u32 s = acc;
  for (int i = threadIdx.x; i < TABLE_B / 4; i += blockDim.x) s += lds[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

struct Res { double ms; };

template <int MODE>
double run(const char *name, const u32 *ddata, u32 *dout, int threads, int iters, int addr_mode, double units_per_iter, const char *unitname) {
  const int blocks = 256;
  const size_t shm = TABLE_B + 256;
  CK(hipFuncSetAttribute((const void *)bench_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  bench_kernel<MODE><<<blocks, threads, shm>>>(ddata, dout, iters / 8, addr_mode);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  bench_kernel<MODE><<<blocks, threads, shm>>>(ddata, dout, iters, addr_mode);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double waves_per_cu = threads / 64.0;
  const double units = units_per_iter * iters;                 // wave-level units per wave
  const double cyc_per_unit_cu = ms * 1e-3 * 2.4e9 / (units * waves_per_cu);   // CU cycles per wave-unit (all waves of the CU share the time)
  printf("%-34s thr=%4d  %8.3f ms  %7.3f cyc/CU per %s  (%.3f cyc per SIMD-unit)\n", name, threads, ms, cyc_per_unit_cu, unitname, cyc_per_unit_cu * 4);
  fflush(stdout);
  return ms;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  const size_t n = (size_t)256 * 1024 * 8;
  std::vector<u32> h(n), hs(n);
  u32 st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (size_t i = 0; i < n; i++) {
    u32 w = 0;
    for (int b = 0; b < 4; b++) w |= (((rnd() % 32) + 1) * 4) << (8 * b);
    h[i] = w;
  }
  // "smooth": runs of ~6 along k for every byte lane
  for (size_t t = 0; t < n / 8; t++) {
    u32 lv[4];
    for (int b = 0; b < 4; b++) lv[b] = (rnd() % 32) + 1;
    for (int k = 0; k < 8; k++) {
      u32 w = 0;
      for (int b = 0; b < 4; b++) {
        if (rnd() % 6 == 0) lv[b] = (rnd() % 32) + 1;
        w |= (lv[b] * 4) << (8 * b);
      }
      hs[t * 8 + k] = w;
    }
  }
  u32 *du, *ds, *dout;
  CK(hipMalloc(&du, n * 4)); CK(hipMalloc(&ds, n * 4)); CK(hipMalloc(&dout, 256 * 1024 * 4));
  CK(hipMemcpy(du, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ds, hs.data(), n * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  for (int threads : {1024, 512, 256}) {
    printf("--- %d threads per CU (%d waves/SIMD) ---\n", threads, threads / 256);
    run<M_VALU_ADD>("valu v_add_u32 (indep x8)", du, dout, threads, iters, 0, 24 * 8, "VALU");
    run<M_VALU_CNDMASK>("valu cmp_sdwa+cndmask", du, dout, threads, iters, 0, 12 * 8 * 2, "VALU");
    run<M_VALU_SDWA>("valu add_sdwa+mul_u24_sdwa (dep)", du, dout, threads, iters, 0, 12 * 8 * 2, "VALU");
    run<M_SALU_MIX>("4 valu + s_mov exec", du, dout, threads, iters, 0, 4 * 8, "5-instr unit");
    run<M_LDS_ONLY>("ds_add random bins", du, dout, threads, iters, 0, 32, "ds_add");
    run<M_LDS_ONLY>("ds_add conflict-free", du, dout, threads, iters, 1, 32, "ds_add");
    run<M_LDS_ONLY>("ds_add same address", du, dout, threads, iters / 4, 2, 32, "ds_add");
    run<M_LDS_ONLY>("ds_add random, half waves idle", du, dout, threads, iters, 3, 32, "ds_add");
    run<M_STEP6>("step6 cndmask+ds_add  uniform", du, dout, threads, iters, 0, 32, "voxel-step");
    run<M_STEP6>("step6 cndmask+ds_add  smooth", ds, dout, threads, iters, 0, 32, "voxel-step");
    run<M_STEP6_NOLDS>("step6 no lds", du, dout, threads, iters, 0, 32, "voxel-step");
    run<M_STEP4X>("step4x exec-masked    uniform", du, dout, threads, iters, 0, 32, "voxel-step");
    run<M_STEP4X>("step4x exec-masked    smooth", ds, dout, threads, iters, 0, 32, "voxel-step");
    run<M_STEP4X_NOLDS>("step4x no lds", du, dout, threads, iters, 0, 32, "voxel-step");
    run<M_STEP4X_NOSALU>("step4x exec restored 1 in 4", du, dout, threads, iters, 0, 32, "voxel-step");
    run<M_STEP5X_CHECK>("step5x +long check   uniform", du, dout, threads, iters, 0, 32, "voxel-step");
    run<M_STEP5X_CHECK>("step5x +long check   smooth", ds, dout, threads, iters, 0, 32, "voxel-step");
  }
  return 0;
}
