// Host-to-device copy rates on this box, for the host-buffer entry points (bftkv_gpu_collective_verify): what a caller's
// pageable memory costs against pinned memory, whole and in pieces, whether hipMemcpyAsync from pageable memory returns before
// the copy is done, what registering the caller's buffer costs, and what a multi-threaded memcpy into a pinned ring reaches.
//   hipcc --offload-arch=gfx950 -O2 -o h2d_rates h2d_rates.hip -lpthread && ./h2d_rates
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const size_t N = 226u << 20, PIECE = 16u << 20;
  uint8_t* pageable = (uint8_t*)malloc(N);
  memset(pageable, 1, N);
  uint8_t *pinned = nullptr, *dev = nullptr;
  CK(hipHostMalloc((void**)&pinned, N, hipHostMallocDefault));
  memset(pinned, 2, N);
  CK(hipMalloc((void**)&dev, N));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    CK(hipMemcpyAsync(dev, pageable, N, hipMemcpyHostToDevice, s));
    double t1 = now();
    CK(hipStreamSynchronize(s));
    double t2 = now();
    printf("pageable whole   %zu MB: call returns after %.3f ms, done after %.3f ms = %.1f GB/s\n", N >> 20, (t1 - t0) * 1e3, (t2 - t0) * 1e3, N / (t2 - t0) / 1e9);
  }
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    for (size_t o = 0; o < N; o += PIECE) CK(hipMemcpyAsync(dev + o, pageable + o, std::min(PIECE, N - o), hipMemcpyHostToDevice, s));
    double t1 = now();
    CK(hipStreamSynchronize(s));
    double t2 = now();
    printf("pageable 16 MB pieces: calls return after %.3f ms, done after %.3f ms = %.1f GB/s\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3, N / (t2 - t0) / 1e9);
  }
  for (int rep = 0; rep < 2; ++rep) {      // two threads, two streams, halves
    double t0 = now();
    std::thread a([&] { (void)hipMemcpyAsync(dev, pageable, N / 2, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); });
    std::thread b([&] { (void)hipMemcpyAsync(dev + N / 2, pageable + N / 2, N - N / 2, hipMemcpyHostToDevice, s2); (void)hipStreamSynchronize(s2); });
    a.join(); b.join();
    double t2 = now();
    printf("pageable halves on two threads / streams: done after %.3f ms = %.1f GB/s\n", (t2 - t0) * 1e3, N / (t2 - t0) / 1e9);
  }
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    CK(hipMemcpyAsync(dev, pinned, N, hipMemcpyHostToDevice, s));
    double t1 = now();
    CK(hipStreamSynchronize(s));
    double t2 = now();
    printf("pinned whole: call returns after %.3f ms, done after %.3f ms = %.1f GB/s\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3, N / (t2 - t0) / 1e9);
  }
  {
    double t0 = now();
    for (size_t o = 0; o < N; o += PIECE) CK(hipMemcpyAsync(dev + o, pinned + o, std::min(PIECE, N - o), hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    double t2 = now();
    printf("pinned 16 MB pieces: done after %.3f ms = %.1f GB/s\n", (t2 - t0) * 1e3, N / (t2 - t0) / 1e9);
  }
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    hipError_t e = hipHostRegister(pageable, N, hipHostRegisterDefault);
    double t1 = now();
    if (e != hipSuccess) { printf("hipHostRegister: %s\n", hipGetErrorString(e)); break; }
    CK(hipMemcpyAsync(dev, pageable, N, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    double t2 = now();
    CK(hipHostUnregister(pageable));
    double t3 = now();
    printf("register %.3f ms, copy %.3f ms = %.1f GB/s, unregister %.3f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, N / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
  }
  for (int nt : {1, 2, 4, 8, 16}) {
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t lo = N * t / nt, hi = N * (t + 1) / nt; memcpy(pinned + lo, pageable + lo, hi - lo); });
      for (auto& x : th) x.join();
      best = std::min(best, now() - t0);
    }
    printf("memcpy pageable -> pinned, %2d threads: %.3f ms = %.1f GB/s\n", nt, best * 1e3, N / best / 1e9);
  }
  // D2H of a small result (10 KB) and an event round trip: the fixed costs at the end of a call
  {
    double t0 = now();
    for (int i = 0; i < 100; ++i) { CK(hipMemcpyAsync(pinned, dev, 10240, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
    printf("10 KB D2H + sync: %.1f us each\n", (now() - t0) * 1e4);
  }
  return 0;
}
