// FETCH_SIZE / WRITE_SIZE calibration for the access patterns of the verifier's kernels (MI355X_MICROARCH.md, HBM section:
// "FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read ... other access widths and WRITE_SIZE are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel touches a known set of bytes of a
// buffer far larger than the 256 MB Infinity Cache, once; run under `rocprofv3 --pmc FETCH_SIZE` (and `--pmc WRITE_SIZE`) and
// compare the counter with the 64-byte and 128-byte lines the pattern touches (printed by this program).
//   stream16     every lane one 16-byte load, consecutive lanes consecutive addresses (the guide's calibrated case)
//   window64     lane r: four 16-byte loads at r*STRIDE + 29           (k_parse_body's LDS window: 64 bytes per packet)
//   bytes256     four lanes per record, 64 byte loads each at r*STRIDE + 31 .. +287   (k_rsa_modexp reading the MPI)
//   hdr12        lane r: three aligned dword loads around r*STRIDE     (k_walk's header read)
//   rec48        lane r: a 48-byte struct at r*48 (three 16-byte loads) (SigRec reads of every kernel)
//   write96      lane r: 24 dwords at r*96, written by 4 lanes x 6      (k_rsa_modexp's result)
//   write48      lane r: 48-byte struct stores                          (k_parse_body's SigRec)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/build/fetch_calib tools/microbench/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr uint64_t STRIDE = 287;            // a Go-shaped RSA-2048 signature packet

__global__ void stream16(const uint4* __restrict__ p, uint64_t n, uint32_t* sink) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 v = p[i];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = 1;
}
__global__ void window64(const uint8_t* __restrict__ p, uint64_t n, uint32_t* sink) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint8_t* q = p + r * STRIDE + 29;
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { uint32_t v[4]; __builtin_memcpy(v, q + 16 * j, 16); acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (acc == 0x12345678u) *sink = 1;
}
__global__ void bytes256(const uint8_t* __restrict__ p, uint64_t n, uint32_t* sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r = t >> 2;
  if (r >= n) return;
  const uint8_t* q = p + r * STRIDE + 31 + (t & 3) * 64;
  uint32_t acc = 0;
  for (int j = 0; j < 64; ++j) acc = acc * 31 + q[j];
  if (acc == 0x12345678u) *sink = 1;
}
__global__ void hdr12(const uint8_t* __restrict__ p, uint64_t n, uint32_t* sink) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint32_t* w = (const uint32_t*)(p + ((r * STRIDE) & ~3ull));
  if ((w[0] ^ w[1] ^ w[2]) == 0x12345678u) *sink = 1;
}
struct Rec48 { uint32_t w[12]; };
__global__ void rec48(const Rec48* __restrict__ p, uint64_t n, uint32_t* sink) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const Rec48 v = p[r];
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < 12; ++j) acc ^= v.w[j];
  if (acc == 0x12345678u) *sink = 1;
}
__global__ void write96(uint32_t* __restrict__ p, uint64_t n) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r = t >> 2;
  if (r >= n) return;
  uint32_t* o = p + r * 24 + (t & 3) * 6;
#pragma unroll
  for (int j = 0; j < 6; ++j) o[j] = (uint32_t)t + j;
}
__global__ void write48(Rec48* __restrict__ p, uint64_t n) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  Rec48 v;
#pragma unroll
  for (int j = 0; j < 12; ++j) v.w[j] = (uint32_t)r + j;
  p[r] = v;
}

// lines of `line` bytes touched by ranges [off + r*stride + a, + len) for r < n
static uint64_t lines(uint64_t n, uint64_t stride, uint64_t a, uint64_t len, uint64_t line, bool align4 = false) {
  uint64_t cnt = 0, last = ~0ull;
  for (uint64_t r = 0; r < n; ++r) {
    uint64_t lo = r * stride + a;
    if (align4) lo &= ~3ull;
    const uint64_t hi = lo + len - 1;
    for (uint64_t l = lo / line; l <= hi / line; ++l) if (l != last) { ++cnt; last = l; }
  }
  return cnt;
}

int main() {
  const uint64_t BYTES = 1536ull << 20;                 // 1.5 GiB: six times the Infinity Cache
  const uint64_t n_rec = (BYTES - 4096) / STRIDE;       // ~5.6 M records
  uint8_t* buf; uint32_t* sink;
  CHECK(hipMalloc(&buf, BYTES));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 0x5A, BYTES));
  CHECK(hipMemset(sink, 0, 4));
  CHECK(hipDeviceSynchronize());
  auto flush = [&]() { CHECK(hipMemset(buf, 0x5A, BYTES)); CHECK(hipDeviceSynchronize()); };   // every pattern starts with cold caches
  const uint64_t n16 = BYTES / 16, n48 = BYTES / 48, n96 = BYTES / 96;
  printf("pattern      units       bytes_asked   lines64*64    lines128*128\n");
  printf("stream16  %10llu %14llu %14llu %14llu\n", (unsigned long long)n16, (unsigned long long)(n16 * 16), (unsigned long long)BYTES, (unsigned long long)BYTES);
  printf("window64  %10llu %14llu %14llu %14llu\n", (unsigned long long)n_rec, (unsigned long long)(n_rec * 64), (unsigned long long)(lines(n_rec, STRIDE, 29, 64, 64) * 64), (unsigned long long)(lines(n_rec, STRIDE, 29, 64, 128) * 128));
  printf("bytes256  %10llu %14llu %14llu %14llu\n", (unsigned long long)n_rec, (unsigned long long)(n_rec * 256), (unsigned long long)(lines(n_rec, STRIDE, 31, 256, 64) * 64), (unsigned long long)(lines(n_rec, STRIDE, 31, 256, 128) * 128));
  printf("hdr12     %10llu %14llu %14llu %14llu\n", (unsigned long long)n_rec, (unsigned long long)(n_rec * 12), (unsigned long long)(lines(n_rec, STRIDE, 0, 12, 64, true) * 64), (unsigned long long)(lines(n_rec, STRIDE, 0, 12, 128, true) * 128));
  printf("rec48     %10llu %14llu %14llu %14llu\n", (unsigned long long)n48, (unsigned long long)(n48 * 48), (unsigned long long)(n48 * 48), (unsigned long long)(n48 * 48));
  printf("write96   %10llu %14llu %14llu %14llu\n", (unsigned long long)n96, (unsigned long long)(n96 * 96), (unsigned long long)(n96 * 96), (unsigned long long)(n96 * 96));
  printf("write48   %10llu %14llu %14llu %14llu\n", (unsigned long long)n48, (unsigned long long)(n48 * 48), (unsigned long long)(n48 * 48), (unsigned long long)(n48 * 48));
  flush(); hipLaunchKernelGGL(stream16, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const uint4*)buf, n16, sink);
  flush(); hipLaunchKernelGGL(window64, dim3((n_rec + 255) / 256), dim3(256), 0, 0, buf, n_rec, sink);
  flush(); hipLaunchKernelGGL(bytes256, dim3((n_rec * 4 + 255) / 256), dim3(256), 0, 0, buf, n_rec, sink);
  flush(); hipLaunchKernelGGL(hdr12, dim3((n_rec + 255) / 256), dim3(256), 0, 0, buf, n_rec, sink);
  flush(); hipLaunchKernelGGL(rec48, dim3((n48 + 255) / 256), dim3(256), 0, 0, (const Rec48*)buf, n48, sink);
  flush(); hipLaunchKernelGGL(write96, dim3((n96 * 4 + 255) / 256), dim3(256), 0, 0, (uint32_t*)buf, n96);
  flush(); hipLaunchKernelGGL(write48, dim3((n48 + 255) / 256), dim3(256), 0, 0, (Rec48*)buf, n48);
  CHECK(hipDeviceSynchronize());
  CHECK(hipGetLastError());
  return 0;
}
