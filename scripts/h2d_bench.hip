// h2d_bench.hip -- how fast can 671 MB of pageable host memory (a 512^3 int32 volume + uint8 mask) reach HBM?
//   hipcc --offload-arch=gfx950 -O3 -pthread -o scripts/h2d_bench scripts/h2d_bench.hip && scripts/h2d_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t n = (size_t)512 * 512 * 512, bytes = n * 5;
  char *src = (char *)malloc(bytes);
  memset(src, 1, bytes);
  char *dev;
  CK(hipMalloc(&dev, bytes));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  printf("host threads available: %u\n", std::thread::hardware_concurrency());
  for (int rep = 0; rep < 2; rep++) {
    double t = now();
    CK(hipMemcpy(dev, src, bytes, hipMemcpyHostToDevice));
    printf("pageable hipMemcpy            : %.2f ms  %.1f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
  }
  for (int rep = 0; rep < 2; rep++) {
    double t = now();
    CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
    double t1 = now();
    CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    double t2 = now();
    CK(hipHostUnregister(src));
    double t3 = now();
    printf("register %.2f + copy %.2f (%.1f GB/s) + unregister %.2f = %.2f ms\n", (t1 - t) * 1e3, (t2 - t1) * 1e3,
           bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t) * 1e3);
  }
  // staged: T worker threads memcpy chunks into a ring of pinned buffers, one DMA per chunk
  for (int T : {1, 2, 4, 8, 12}) {
    for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20}) {
      const int R = 8;
      char *ring[R];
      hipEvent_t ev[R];
      for (int i = 0; i < R; i++) { CK(hipHostMalloc(&ring[i], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
      const size_t nchunks = (bytes + chunk - 1) / chunk;
      double best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        double t = now();
        std::atomic<size_t> filled[R];
        for (int i = 0; i < R; i++) filled[i] = 0;
        // per chunk: workers split the memcpy into T slices; the main thread waits for all slices then issues the DMA
        std::vector<std::thread> th;
        std::atomic<size_t> go{0}, done{0};
        std::atomic<bool> quit{false};
        for (int w = 0; w < T; w++)
          th.emplace_back([&, w]() {
            size_t seen = 0;
            for (;;) {
              size_t g;
              while ((g = go.load(std::memory_order_acquire)) == seen) { if (quit.load()) return; }
              for (size_t c = seen; c < g; c++) {
                const size_t off = c * chunk, len = std::min(chunk, bytes - off);
                const size_t sl = (len + T - 1) / T, a = std::min(len, w * sl), b = std::min(len, a + sl);
                memcpy(ring[c % R] + a, src + off + a, b - a);
                done.fetch_add(1, std::memory_order_release);
              }
              seen = g;
            }
          });
        for (size_t c = 0; c < nchunks; c++) {
          if (c >= R) CK(hipEventSynchronize(ev[c % R]));   // slot free again
          go.store(c + 1, std::memory_order_release);
          while (done.load(std::memory_order_acquire) < (c + 1) * T) {}
          const size_t off = c * chunk, len = std::min(chunk, bytes - off);
          CK(hipMemcpyAsync(dev + off, ring[c % R], len, hipMemcpyHostToDevice, s));
          CK(hipEventRecord(ev[c % R], s));
        }
        CK(hipStreamSynchronize(s));
        quit = true;
        for (auto &x : th) x.join();
        best = std::min(best, now() - t);
      }
      printf("staged T=%2d chunk=%2zu MB       : %.2f ms  %.1f GB/s\n", T, chunk >> 20, best * 1e3, bytes / best / 1e9);
      for (int i = 0; i < R; i++) { CK(hipHostFree(ring[i])); CK(hipEventDestroy(ev[i])); }
    }
  }
  return 0;
}
