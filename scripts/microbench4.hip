// round 6: issue rate of ONE wave per SIMD on gfx950 -- a chain of dependent VALU instructions against independent chains,
// and the cost of an SALU instruction that consumes a VALU compare.  The sliding-window map kernels (kernels_voxslide.h) run at
// one wave per SIMD because their tables fill the LDS.   hipcc --offload-arch=gfx950 -O3 scripts/microbench4.hip -o mb4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, unsigned seed) {
  extern __shared__ unsigned lds[];
  unsigned a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u;
  unsigned long long e = a, f = a + 1;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) {   // one dependent chain: 4 adds
        asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      } else if (MODE == 1) {   // four independent chains
        asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if (MODE == 2) {   // two independent chains
        asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(seed));
      } else if (MODE == 3) {   // dependent chain through SDWA / mul / shift like the table index
        asm volatile("v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_mul_i32_i24 %0, %0, %1\n\tv_ashrrev_i32 %0, 1, %0\n\tv_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      } else if (MODE == 4) {   // dependent 64-bit adds
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %0, %0, 0, %1" : "+v"(e) : "v"((unsigned long long)b));
      } else if (MODE == 5) {   // compare -> cndmask (vector only), 4 instructions
        asm volatile("v_cmp_ne_u32 vcc, 0, %0\n\tv_cndmask_b32 %0, %1, %0, vcc\n\tv_cmp_ne_u32 vcc, 0, %0\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(a) : "v"(b) : "vcc");
      } else if (MODE == 6) {   // compare -> s_and_saveexec -> add -> s_or exec (the form the compiler chose), 4 instructions + 1 add
        asm volatile("v_cmp_ne_u32 vcc, 0, %0\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_add_u32 %0, %0, %1\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
      } else if (MODE == 7) {   // independent 64-bit adds (2 chains)
        asm volatile("v_lshl_add_u64 %0, %0, 0, %2\n\tv_lshl_add_u64 %1, %1, 0, %2\n\tv_lshl_add_u64 %0, %0, 0, %2\n\tv_lshl_add_u64 %1, %1, 0, %2" : "+v"(e), "+v"(f) : "v"((unsigned long long)b));
      } else if (MODE == 8) {   // cndmask with an SGPR-pair mask written by the preceding compare (e64 forms)
        asm volatile("v_cmp_ne_u32 s[20:21], 0, %0\n\tv_cndmask_b32 %0, %1, %0, s[20:21]\n\tv_cmp_ne_u32 s[22:23], 0, %0\n\tv_cndmask_b32 %0, %1, %0, s[22:23]" : "+v"(a) : "v"(b) : "s20", "s21", "s22", "s23");
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (unsigned)e + (unsigned)f;
}
template <int MODE>
void run(const char *name, unsigned *out) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<256, 256, 150 * 1024>>>(out, 100, 1);
  hipEventRecord(e0);
  k<MODE><<<256, 256, 150 * 1024>>>(out, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 16 * 4;
  printf("%-70s %8.3f ms  %6.2f ns per group of 4 = %5.2f cycles per instruction at 2.4 GHz\n", name, ms, ms * 1e6 / (iters * 16.0), ms * 1e-3 * 2.4e9 / instr);
  fflush(stdout);
}
int main() {
  unsigned *out;
  hipMalloc(&out, 256 * 256 * 4);
  run<0>("one chain of dependent v_add_u32", out);
  run<2>("two independent chains", out);
  run<1>("four independent chains", out);
  run<3>("dependent sdwa max / mul24 / ashr / add", out);
  run<4>("dependent v_lshl_add_u64", out);
  run<7>("two chains of v_lshl_add_u64", out);
  run<5>("v_cmp (vcc) -> v_cndmask, dependent", out);
  run<8>("v_cmp (sgpr pair) -> v_cndmask e64, dependent", out);
  // (mode 6, v_cmp -> s_and_saveexec -> v_add -> s_or exec as hand-written asm, did not return on the box: not run)
  return 0;
}
