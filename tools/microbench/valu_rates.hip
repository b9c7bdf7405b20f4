// Micro-benchmark: issue cost of the integer/fp64 VALU instructions a big-integer
// Montgomery kernel is built from, on gfx950.  Standalone (hipcc), not part of the product.
// Output per instruction and occupancy (1/2/4/8 waves per SIMD): wall time, chip-wide wave-instructions per second, the
// SHADER CLOCK during the run -- measured inside the kernel as (s_memtime ticks) / (s_memrealtime ticks) x 100 MHz over the
// whole life of wave 0 -- and from these the cycles one wave-instruction occupies a SIMD's issue port:
//     cycles/inst/SIMD = wall_time x sclk / (instructions per wave x waves per SIMD).
// (Round 1 printed wave 0's own cycle count divided by wall time as "eff clk"; wave 0 does not live for the whole launch at
// 4 and 8 waves per SIMD, so that column fell with the occupancy and said nothing about the clock.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int CHAINS = 8;

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed, unsigned long long* cyc) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
  uint64_t acc[CHAINS];
  double d[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { acc[i] = seed + i; d[i] = 1.0 + i + seed; }
  uint32_t lo[CHAINS];
  for (int i = 0; i < CHAINS; ++i) lo[i] = seed + i;
  double da = 1.0000001 + seed, db = 0.5;
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if constexpr (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if constexpr (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if constexpr (OP == 3) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo[i]) : "v"(a), "v"(b));
      if constexpr (OP == 4) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if constexpr (OP == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
      if constexpr (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo[i]) : "v"(a) : "vcc");
      if constexpr (OP == 7) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[i]));
      if constexpr (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if constexpr (OP == 9) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(lo[i]) : "v"(a));
      if constexpr (OP == 10) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(lo[i]));
      if constexpr (OP == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, s4, %0" : "+v"(acc[i]) : "v"(a) : "vcc", "s4");
      if constexpr (OP == 12) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[i]) : "v"(a) : "vcc");
      if constexpr (OP == 13) asm volatile("v_mov_b64 %0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % CHAINS]));
      if constexpr (OP == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if constexpr (OP == 15) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % CHAINS]));
      if constexpr (OP == 16) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if constexpr (OP == 17) asm volatile("v_readlane_b32 s4, %0, 3\n\tv_add_u32 %0, s4, %0" : "+v"(lo[i]) : : "s4");
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  uint64_t r = 0;
  for (int i = 0; i < CHAINS; ++i) r += acc[i] + lo[i] + (uint64_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

// single dependent chain, one wave per SIMD: latency
template <int OP>
__global__ void __launch_bounds__(256) lat(uint64_t* out, uint32_t seed, unsigned long long* cyc) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
  uint64_t acc = seed; uint32_t lo = seed; double d = seed, da = 1.0000001, db = 0.5;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
      if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
      if constexpr (OP == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d) : "v"(da), "v"(db));
      if constexpr (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lo + (uint64_t)d;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

typedef void (*kern_t)(uint64_t*, uint32_t, unsigned long long*);

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int ndev = 0; CHECK(hipGetDeviceCount(&ndev));
  printf("device %s, CUs %d, clock %d kHz, ndev %d\n", prop.name, prop.multiProcessorCount, prop.clockRate, ndev);
  int cus = prop.multiProcessorCount;
  uint64_t* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, sizeof(uint64_t) * cus * 8 * 256));
  CHECK(hipMalloc(&cyc, 16));
  struct { const char* name; kern_t fn; } ops[] = {
    {"v_mad_u64_u32", k<0>}, {"v_mul_lo_u32", k<1>}, {"v_mul_hi_u32", k<2>}, {"v_mad_u32_u24", k<3>},
    {"v_mul_hi_u32_u24", k<4>}, {"v_fma_f64", k<5>}, {"v_add_co_u32", k<6>}, {"v_lshrrev_b64", k<7>},
    {"v_add_u32", k<8>}, {"v_alignbit_b32", k<9>}, {"v_mov_b32_dpp", k<10>}, {"v_mad_u64_u32(sgpr)", k<11>},
    {"v_addc_co_u32", k<12>}, {"v_mov_b64", k<13>}, {"v_and_b32", k<14>}, {"v_lshl_add_u64", k<15>},
    {"v_mul_u32_u24", k<16>}, {"v_readlane+v_add", k<17>},
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (auto& op : ops) {
    for (int wps : {1, 2, 4, 8}) {
      // blocks of 256 threads = 4 waves = 1 wave per SIMD; wps blocks per CU
      int blocks = cus * wps;
      op.fn<<<blocks, 256>>>(out, 1, cyc);  // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      op.fn<<<blocks, 256>>>(out, 1, cyc);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c[2]; CHECK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
      double n_inst_per_wave = (double)ITERS * CHAINS;
      double sclk = c[1] ? (double)c[0] / (double)c[1] * 100e6 : 0.0;   // s_memrealtime ticks at a constant 100 MHz
      double cyc_per_inst_simd = (ms * 1e-3) * sclk / (n_inst_per_wave * wps);  // SIMD issue-port cycles per wave-instruction
      double ginst = n_inst_per_wave * blocks * 4 / (ms * 1e-3) / 1e9;  // wave-instr/s
      printf("%-22s waves/SIMD %d  %.3f ms  %.1f G wave-inst/s (%.2f T lane-op/s)  sclk %.2f GHz  => %.2f cyc/wave-inst/SIMD  (wave 0 alone: %.2f cyc/inst)\n",
             op.name, wps, ms, ginst, ginst * 64 / 1e3, sclk / 1e9, cyc_per_inst_simd, (double)c[0] / n_inst_per_wave);
    }
  }
  struct { const char* name; kern_t fn; } lats[] = {
    {"lat v_mad_u64_u32", lat<0>}, {"lat v_mul_lo_u32", lat<1>}, {"lat v_fma_f64", lat<5>}, {"lat v_add_u32", lat<8>}};
  for (auto& op : lats) {
    op.fn<<<cus, 256>>>(out, 1, cyc);
    CHECK(hipDeviceSynchronize());
    op.fn<<<cus, 256>>>(out, 1, cyc);
    CHECK(hipDeviceSynchronize());
    unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-22s dependent chain: %.2f cyc/inst\n", op.name, (double)c / (ITERS * 8.0));
  }
  return 0;
}
