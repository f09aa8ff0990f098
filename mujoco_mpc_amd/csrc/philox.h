// philox.h — Philox4x32-10 counter RNG + Box-Muller standard normals, shared by the noise kernel (candidate policies) and
// the rollout kernel (Ornstein-Uhlenbeck external-force noise of the robust planner).  Counter = (c0, c1, stream lo, stream hi),
// key = seed: every (candidate, element) pair has its own sample, no state to carry.
#pragma once
#include "spmd.h"

DEV void philox4x32_10(unsigned long long seed, unsigned long long stream, unsigned c0, unsigned c1, unsigned out[4]) {
  unsigned c[4] = {c0, c1, (unsigned)stream, (unsigned)(stream >> 32)};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  for (int r = 0; r < 10; r++) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
DEV double philox_normal(unsigned long long seed, unsigned long long stream, unsigned i, unsigned e) {
  unsigned o[4];
  philox4x32_10(seed, stream, i, e, o);
  unsigned long long x1 = ((unsigned long long)o[0] << 32) | o[1], x2 = ((unsigned long long)o[2] << 32) | o[3];
  double u1 = (double)((x1 >> 11) + 1) * (1.0 / 9007199254740992.0);
  double u2 = (double)(x2 >> 11) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cos(2.0 * 3.14159265358979323846 * u2);
}
#define XFRC_STREAM 0x5846524300000000ull    // "XFRC": separates the force noise from the knot noise of the same plan
