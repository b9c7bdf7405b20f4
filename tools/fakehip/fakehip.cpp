// A stand-in for libamdhip64 on a machine WITHOUT a GPU, for ONE purpose: running the host side of libbftkv_gpu.so -- the
// micro-batcher's door / leader / lanes, the host-buffer pipeline's worker threads, the context locks -- under ThreadSanitizer
// (tools/tsan_host.sh).  Device memory is host memory, copies are memcpy, every stream is synchronous, kernels are NOT run
// (hipLaunchKernel counts the launch and returns): results are whatever the zeroed buffers say, only the host-side concurrency
// is under test.  Never linked into the product; the product library links the real runtime and has no fallback.
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace {
struct Cfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local Cfg t_cfg[8];
thread_local int t_depth = 0;
std::atomic<unsigned long long> g_launches{0}, g_bytes{0};
// the one kernel whose effect the HOST waits for without a stream synchronisation: k_finish_staged publishes a staged small call's
// results and its sequence number in mapped host memory (csrc/kernels.hip; batcher_capi.inc spins on the number).  Registered at
// load time (single-threaded), read afterwards.
const void* g_finish_staged = nullptr;
void* zalloc(size_t n) {
  void* p = nullptr;
  if (posix_memalign(&p, 256, n ? n : 1)) return nullptr;
  memset(p, 0, n);
  return p;
}
}  // namespace

extern "C" {
unsigned long long fakehip_launches() { return g_launches.load(); }

void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void* host_fn, char*, const char* device_name, unsigned, void*, void*, void*, void*, int*) {
  if (device_name && strstr(device_name, "k_finish_staged")) g_finish_staged = host_fn;
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
  if (t_depth < 8) t_cfg[t_depth] = Cfg{grid, block, shmem, stream};
  ++t_depth;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
  --t_depth;
  const Cfg& c = t_cfg[t_depth < 8 ? t_depth : 7];
  *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void* fn, dim3, dim3, void** args, size_t, hipStream_t) {
  g_launches.fetch_add(1, std::memory_order_relaxed);
  if (fn == g_finish_staged && fn) {     // (verdict_or_err, from_verdict, item_flags, item_hash_mask, n, rehash_bits, host_out, host_flag, seq, total)
    const uint8_t* v = *(const uint8_t**)args[0];
    const uint32_t from_verdict = *(uint32_t*)args[1], n = *(uint32_t*)args[4], seq = *(uint32_t*)args[8];
    uint8_t* out = *(uint8_t**)args[6];
    uint32_t* flag = *(uint32_t**)args[7];
    for (uint32_t i = 0; i < n; ++i) { out[i] = from_verdict ? ((v[i] & 4) ? 0 : 2) : v[i]; out[n + i] = 0; }
    memset(out + 2 * (size_t)n, 0, 8);
    __atomic_store_n(flag, seq, __ATOMIC_RELEASE);
  }
  return hipSuccess;
}

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : a == hipDeviceAttributeClockRate ? 2400000 : 0;
  return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = 2ull << 30; *total_b = 4ull << 30; return hipSuccess; }   // small: no wide DSA tables
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fakehip"; }

hipError_t hipMalloc(void** p, size_t n) { *p = zalloc(n); g_bytes += n; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = zalloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)zalloc(64); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)zalloc(64); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)zalloc(64); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)zalloc(64); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free((void*)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
}
