/* orc_rng.h -- TEST INFRASTRUCTURE (oracle). Not part of the product path.
 *
 * Two random sources for the CPU restatement:
 *
 *  (A) mt19937 + libstdc++-11 distribution semantics -- what the reference
 *      draws from (one shared std::mt19937, SingleSiteSampler.cpp:487;
 *      generate_canonical<double,53> = two 32-bit words,
 *      /usr/include/c++/11/bits/random.tcc:3348-3380;
 *      exponential_distribution = -log(1-u)/lambda, bits/random.h).
 *
 *  (B) Philox4x32-10 (Salmon et al., SC'11 -- the generator rocRAND ships as
 *      rocrand_philox4x32_10) used as a *random-access* function of
 *      (seed, site, sweep, branch, segment, trial, block).  This is the
 *      parallel-schedule contract the gfx950 kernels reproduce bit-for-bit.
 */
#ifndef ORC_RNG_H
#define ORC_RNG_H

#include <stdint.h>

/* ---------------- mt19937 (32-bit Mersenne twister) -------------------- */
typedef struct {
  uint32_t mt[624];
  int idx;
} orc_mt19937;

static inline void orc_mt_seed(orc_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

static inline uint32_t orc_mt_next(orc_mt19937 *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      g->mt[i] = v;
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* libstdc++ generate_canonical<double,53>(mt19937): sum = g1 + g2*2^32
 * (rounded to double), / 2^64, clamped below 1. */
static inline double orc_mt_canonical(orc_mt19937 *g) {
  const double g1 = (double)orc_mt_next(g);
  const double g2 = (double)orc_mt_next(g);
  double sum = 0.0;
  sum += g1 * 1.0;
  sum += g2 * 4294967296.0;
  double ret = sum / 18446744073709551616.0;
  if (ret >= 1.0) ret = 0.99999999999999988897769753748; /* nextafter(1,0) */
  return ret;
}

/* ---------------- Philox4x32-10 ----------------------------------------- */
#define ORC_PHILOX_M0 0xD2511F53u
#define ORC_PHILOX_M1 0xCD9E8D57u
#define ORC_PHILOX_W0 0x9E3779B9u
#define ORC_PHILOX_W1 0xBB67AE85u

static inline void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                                     uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)ORC_PHILOX_M0 * c0;
    const uint64_t p1 = (uint64_t)ORC_PHILOX_M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += ORC_PHILOX_W0;
    k1 += ORC_PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Counter layout of the parallel-schedule contract:
 *   key = (seed lo, seed hi)
 *   c0 = site, c1 = sweep, c2 = trial,
 *   c3 = branch<<20 | segment<<8 | block          (12 / 12 / 8 bits)
 * A block yields two doubles in [0,1): d0 from words (1:0), d1 from (3:2),
 * each the top 53 bits of the 64-bit word pair times 2^-53.
 *   accept uniform              : (b=0,k=0,t=0,blk=0).d0
 *   segment end-state uniform   : (b,k,t=0,blk=0).d0
 *   trial 1, first draw         : (b,k,t=0,blk=0).d1        (same block as the line above)
 *   trial t>=2, first draw      : (b,k,t>>1,blk=255).d[t&1]  (trials 2m, 2m+1 share a block)
 *   trial t>=1, draw d>=1       : (b,k,t,blk=(d-1)>>1).d[(d-1)&1]
 * Most trials end at their first draw (no jump inside the segment), so the common
 * case costs one Philox block per segment.
 */
#define ORC_FIRST_DRAW_BLOCK 255u
#define ORC_MAX_BRANCH 4095u
#define ORC_MAX_SEG 4095u
#define ORC_MAX_BLOCK 255u

static inline void orc_keyed_block(uint64_t seed, uint32_t site, uint32_t sweep,
                                   uint32_t b, uint32_t k, uint32_t t, uint32_t blk,
                                   double d[2]) {
  const uint32_t ctr[4] = {site, sweep, t, (b << 20) | (k << 8) | blk};
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t w[4];
  orc_philox4x32_10(ctr, key, w);
  const uint64_t a = ((uint64_t)w[1] << 32) | w[0];
  const uint64_t c = ((uint64_t)w[3] << 32) | w[2];
  d[0] = (double)(a >> 11) * 1.1102230246251565404e-16; /* 2^-53 */
  d[1] = (double)(c >> 11) * 1.1102230246251565404e-16;
}

#endif
