// microbenchmark (round 4): float64 VALU issue rate and dependent-operation latency on gfx950, one wave per SIMD and several.
// hipcc --offload-arch=gfx950 -O3 scripts/fp64_latency.hip -o scripts/fp64_latency && scripts/fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void dep_add(double *out, double a, int iters) {
  double x = threadIdx.x, y = a;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 64; k++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + (double)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[N] = (double)(t1 - t0) / (64.0 * iters);
}
__global__ void dep_mul(double *out, double a, int iters) {
  double x = 1.0 + threadIdx.x * 1e-9, y = a;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 64; k++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[N] = (double)(t1 - t0) / (64.0 * iters);
}
template <int CH>
__global__ void indep_add(double *out, double a, int iters) {
  double x[CH];
  for (int c = 0; c < CH; c++) x[c] = threadIdx.x + c;
  double y = a;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 64 / CH; k++) {
#pragma unroll
      for (int c = 0; c < CH; c++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y));
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CH; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[N] = (double)(t1 - t0) / (64.0 / CH * CH * iters);
}
__global__ void dep_add_f32(double *out, float a, int iters) {
  float x = threadIdx.x, y = a;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 64; k++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[N] = (double)(t1 - t0) / (64.0 * iters);
}
int main() {
  double *d;
  hipMalloc(&d, sizeof(double) * (N + 1) * 64);
  double h;
  auto rep = [&](const char *name) {
    hipDeviceSynchronize();
    hipMemcpy(&h, d + N, sizeof(double), hipMemcpyDeviceToHost);
    printf("%-44s %.2f clock64 ticks per instruction (wave 0)\n", name, h);
  };
  // clock64 = s_memtime: 100 MHz constant clock on this family?  calibrate against wall time
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {1, 2, 4, 8}) {
    const int threads = 64 * waves;   // waves of one workgroup share a CU: 1 wave -> one SIMD, 4 -> one per SIMD, 8 -> two per SIMD
    printf("--- %d wave(s) in one workgroup\n", waves);
    hipEventRecord(e0);
    hipLaunchKernelGGL(dep_add, dim3(1), dim3(threads), 0, 0, d, 1.0, 2000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    rep("dependent v_add_f64");
    printf("   (kernel %.3f ms for %d instructions per wave -> %.2f ns per instruction)\n", ms, 64 * 2000, ms * 1e6 / (64.0 * 2000));
    hipLaunchKernelGGL(dep_mul, dim3(1), dim3(threads), 0, 0, d, 1.0000001, 2000); rep("dependent v_mul_f64");
    hipLaunchKernelGGL(indep_add<2>, dim3(1), dim3(threads), 0, 0, d, 1.0, 2000); rep("2 independent chains v_add_f64");
    hipLaunchKernelGGL(indep_add<4>, dim3(1), dim3(threads), 0, 0, d, 1.0, 2000); rep("4 independent chains v_add_f64");
    hipLaunchKernelGGL(indep_add<8>, dim3(1), dim3(threads), 0, 0, d, 1.0, 2000); rep("8 independent chains v_add_f64");
    hipLaunchKernelGGL(dep_add_f32, dim3(1), dim3(threads), 0, 0, d, 1.0f, 2000); rep("dependent v_add_f32");
  }
  return 0;
}
