/* The division-step inverse of bftkv_amd/csrc/safegcd.inc compiled for the CPU (the same text k_modinv compiles for the GPU), so that
 * tests/test_safegcd.py can check its arithmetic against Python's pow(x, -1, m) in the CPU suite.  Test infrastructure only. */
#include <stdint.h>
#define SG_FN static
#include "../../bftkv_amd/csrc/safegcd.inc"

/* x, m, out: 76 limbs of 28 bits, little-endian (the multiplier's layout).  Returns 1 and the inverse, or 0. */
int sg_host_modinv(const uint32_t* x28, const uint32_t* m28, uint32_t* out28) {
  sg_num x, m, o;
  sg_from28(x28, 76, &x);
  sg_from28(m28, 76, &m);
  if (!sg_modinv(&x, &m, &o)) { for (int j = 0; j < 76; ++j) out28[j] = 0; return 0; }
  sg_to28(&o, out28, 76);
  return 1;
}

/* the rounds a pair takes (the bound in the header is on the worst case) */
int sg_host_rounds(const uint32_t* x28, const uint32_t* m28) {
  sg_num f, g;
  sg_from28(m28, 76, &f);
  sg_from28(x28, 76, &g);
  int32_t eta = -1, t[4];
  for (int round = 0; round < 1000; ++round) {
    int32_t nz = 0;
    for (int i = 0; i < SG_NL; ++i) nz |= g.v[i];
    if (!nz) return round;
    eta = sg_divsteps30(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    sg_update_fg(&f, &g, t);
  }
  return -1;
}
