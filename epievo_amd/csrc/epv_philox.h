// epv_philox.h -- random-access Philox4x32-10 for the gfx950 kernels.
//
// Philox4x32-10 (Salmon et al., SC'11) is the generator rocRAND ships as
// rocrand_philox4x32_10; here it is used as a pure function of its counter instead
// of through rocRAND's per-thread state objects, because the sampler needs random
// ACCESS: trial t of segment k of branch b of site s in sweep w must give the same
// numbers whichever lane evaluates it (the wave-cooperative rejection search in
// epv_kernels.hip evaluates 64 trials of one segment at once), and keeping no
// generator state in HBM makes the RNG cost zero bytes of traffic.
//
// Counter layout (identical in the CPU oracle, oracle/orc_rng.h):
//   key = (seed lo, seed hi)
//   c0 = global site index, c1 = sweep, c2 = trial,
//   c3 = branch<<20 | segment<<8 | block            (12 / 12 / 8 bits)
// A block yields two doubles in [0,1): d0 from words (1:0), d1 from (3:2), each the
// top 53 bits of the 64-bit pair times 2^-53.
//   accept uniform            : (b=0,k=0,t=0,blk=0).d0      (SingleSiteSampler.cpp:520-521)
//   segment end-state uniform : (b,k,t=0,blk=0).d0          (SingleSiteSampler.cpp:206)
//   trial 1, first draw       : (b,k,t=0,blk=0).d1          (same block as the line above)
//   trial t>=2, first draw    : (b,k,t>>1,blk=255).d[t&1]   (trials 2m, 2m+1 share a block)
//   trial t>=1, draw d>=1     : (b,k,t,blk=(d-1)>>1).d[(d-1)&1]   (EndCondSampling.cpp:470-474)
// Most trials end at their first draw (no jump inside the segment), so the common case
// costs ONE Philox block per segment.
#ifndef EPV_PHILOX_H
#define EPV_PHILOX_H

#include <stdint.h>

#define EPV_PHILOX_M0 0xD2511F53u
#define EPV_PHILOX_M1 0xCD9E8D57u
#define EPV_PHILOX_W0 0x9E3779B9u
#define EPV_PHILOX_W1 0xBB67AE85u
#define EPV_FIRST_DRAW_BLOCK 255u

struct epv_block2 {
  double d0, d1;
};

__device__ __forceinline__ epv_block2 epv_keyed_block(uint32_t seed_lo, uint32_t seed_hi,
                                                      uint32_t site, uint32_t sweep,
                                                      uint32_t b, uint32_t k, uint32_t t,
                                                      uint32_t blk) {
  uint32_t c0 = site, c1 = sweep, c2 = t, c3 = (b << 20) | (k << 8) | blk;
  uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#if !defined(EPV_PHILOX_MUL_HI_LO)
    // one v_mad_u64_u32 per 32 x 32 -> 64 product instead of the v_mul_hi_u32 + v_mul_lo_u32 the
    // compiler picks (all three are quarter-rate on CDNA: 20 instead of 40 slow multiplies per
    // block; +2..4 % end to end, tools/ab_bench.py; the same bits)
    unsigned long long p0, p1;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "s"(EPV_PHILOX_M0), "v"(c0) : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "s"(EPV_PHILOX_M1), "v"(c2) : "vcc");
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
    const uint32_t hi0 = __umulhi(EPV_PHILOX_M0, c0);
    const uint32_t lo0 = EPV_PHILOX_M0 * c0;
    const uint32_t hi1 = __umulhi(EPV_PHILOX_M1, c2);
    const uint32_t lo1 = EPV_PHILOX_M1 * c2;
#endif
#if !defined(EPV_PHILOX_XOR2)
    // a ^ b ^ key in ONE instruction: gfx950's three-input boolean op with the XOR3 truth table (0x96);
    // the compiler emits two v_xor_b32 (40 instead of 20 per block).  The key is wave-uniform (kernel
    // arguments plus round constants), hence the scalar operand.
    uint32_t n0, n2;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"(hi1), "v"(c1), "s"(k0));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"(hi0), "v"(c3), "s"(k1));
#else
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
#endif
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += EPV_PHILOX_W0;
    k1 += EPV_PHILOX_W1;
#ifndef EPV_PHILOX_HOISTKEYS
    // keep the key schedule a chain of scalar adds next to its use: hoisted out of the kernels' loops the
    // twenty round keys live in SGPRs from the first block to the last and push other values into spills
    // (fused phase: 108 -> 91 spilled SGPRs, +1..2 % on tree.nwk; the large-tree kernels: +-0)
    asm volatile("" : "+s"(k0), "+s"(k1));
#endif
  }
  const uint64_t a = ((uint64_t)c1 << 32) | c0;
  const uint64_t c = ((uint64_t)c3 << 32) | c2;
  epv_block2 out;
  out.d0 = (double)(a >> 11) * 1.1102230246251565404e-16;  // 2^-53
  out.d1 = (double)(c >> 11) * 1.1102230246251565404e-16;
  return out;
}

#endif
