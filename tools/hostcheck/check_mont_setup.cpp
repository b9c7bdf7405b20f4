// CPU: hostbn::mont_setup (Montgomery exponentiation on 64-bit words -- what the library calls for every key and modulus) against
// hostbn::mont_setup_by_doubling (the definition: 1 doubled 2*28*nlimbs times) word for word: random and edge-case moduli of every
// width the callers use, for every limb count they use (76, 80, 112, 152).   g++ -O2 -std=c++17 check_mont_setup.cpp && ./a.out [n]
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include "../../bftkv_amd/csrc/host_bignum.h"
using namespace bftkv::hostbn;

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3000;
  const int limbs[4] = {76, 80, 112, 152};
  const int max_bytes[4] = {300, 300, 420, 560};   // (past what the words hold: the top bytes are dropped by both forms)
  long cases = 0, bad = 0;
  double t_new = 0, t_old = 0;
  for (int li = 0; li < 4; ++li) {
    const int nl = limbs[li];
    for (int r = 0; r < rounds; ++r) {
      uint8_t n[600] = {0};
      uint32_t len = 1 + rnd() % (uint32_t)max_bytes[li];
      if (r % 7 == 0) len = (uint32_t)max_bytes[li];
      for (uint32_t i = 0; i < len; ++i) n[i] = (uint8_t)rnd();
      const int style = r % 16;
      if (style == 1) { for (uint32_t i = 0; i < len; ++i) n[i] = 0xFF; }                 // 2^(8 len) - 1
      else if (style == 2) { for (uint32_t i = 0; i < len; ++i) n[i] = 0; n[len - 1] = 1; }   // one
      else if (style == 3) { for (uint32_t i = 0; i < len; ++i) n[i] = 0; n[len - 1] = 3; }   // three
      else if (style == 4) { for (uint32_t i = 1; i < len; ++i) n[i] = 0; n[0] = 0x80; }      // 2^k + 1 below
      else if (style == 5) { for (uint32_t i = 0; i + 1 < len; ++i) n[i] = 0; }             // one byte
      else if (style == 6) { n[0] = 0; if (len > 1) n[1] = 0; }                             // leading zero bytes
      if (style != 7) n[len - 1] |= 1; else n[len - 1] &= 0xFE;                             // style 7: even (refused)
      if (style == 8) for (uint32_t i = 0; i < len; ++i) n[i] = 0;                           // zero (refused)
      if (style == 9 && len > 8) { for (uint32_t i = 0; i < len; ++i) n[i] = 0xFF; n[len / 2] = 0; }
      std::vector<uint32_t> a_n(nl), a_r(nl), b_n(nl), b_r(nl);
      uint32_t a0 = 0x55, b0 = 0xAA;
      auto t0 = std::chrono::steady_clock::now();
      const bool oka = mont_setup(n, len, nl, a_n.data(), a_r.data(), &a0);
      auto t1 = std::chrono::steady_clock::now();
      const bool okb = mont_setup_by_doubling(n, len, nl, b_n.data(), b_r.data(), &b0);
      auto t2 = std::chrono::steady_clock::now();
      t_new += std::chrono::duration<double>(t1 - t0).count(); t_old += std::chrono::duration<double>(t2 - t1).count();
      ++cases;
      if (oka != okb || (oka && (a0 != b0 || a_n != b_n || a_r != b_r))) {
        if (++bad < 5) printf("MISMATCH limbs %d len %u style %d ok %d/%d\n", nl, len, style, (int)oka, (int)okb);
      }
    }
  }
  printf("%ld cases, %ld mismatches; mont_setup %.1f us, by doubling %.1f us per modulus\n", cases, bad, 1e6 * t_new / cases, 1e6 * t_old / cases);
  return bad != 0;
}
