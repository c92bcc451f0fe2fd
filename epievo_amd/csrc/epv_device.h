// epv_device.h -- device-side data layout shared by the kernels and the ABI glue.
//
// HBM layout (SoA, "jump times packed as SoA for coalesced access"):
//   meta [2][B][n]     u16  bit15 = Path::init_state, bits0-14 = number of jumps.  Branch-major:
//                           a wave's loads of one branch are stride-3 words (a few cache
//                           lines); the site-major alternative [2][n][B] costs a line per two
//                           lanes at 30 branches (measured: 2.41 vs 1.63 ms per phase)
//   jumps[2][B][C][n]  f64  jump k of (buffer, branch, site) at ((buf*B+b)*C+k)*n+site
//   sel  [n]           u8   which of the two buffers holds the CURRENT path of a site
//   tri  [n]           f64  cached complete-data log-likelihood of the triple centred
//                           at each site (SingleSiteSampler's private tri_llh)
//   prop_llr [n] f64, prop_flag [n] u8   hand-over from the propose to the accept kernel
// Lanes of a wavefront map to sites, so meta/sel/tri/jump-plane loads are unit- or
// stride-3-coalesced.  The proposal of a site is written straight into the site's
// OTHER buffer; accepting it is a one-byte flip of sel[site] (no copy), rejecting it
// costs nothing.
#ifndef EPV_DEVICE_H
#define EPV_DEVICE_H

#include <stdint.h>

// meta word of a path: bit 15 = Path::init_state, bits 0-14 = number of jumps.  The capacity
// (jump slots per path) is bounded by the 12-bit segment field of the Philox address: a branch
// has at most 2 C + 1 <= 4095 segments.
typedef uint16_t epv_meta_t;
#define EPV_INIT_SHIFT 15
#define EPV_NJ_MASK 0x7fffu
#define EPV_MAX_CAP 2047u
#define EPV_WAVE 64

struct EpvModelConst {  // staged into LDS by every block
  double rates[8];
  double log_rates[8];
  double T[4];
};

struct EpvSegTask {      // one dirty segment of a proposal: sample its jump times
  unsigned long long w0; // site | node << 40 | segment index << 52
  double len;            // segment length (< 0: blank entry)
  double start;          // time of the segment's start on the branch
  unsigned long long w3; // start state | end state << 1 | trip0 << 2
};
struct EpvSegOut {       // what the search found
  uint32_t cnt;          // jumps of the winning trial
  uint32_t tstar;        // the winning trial
  uint32_t maxm;         // most jumps any trial up to the winner made (capacity check)
  uint32_t pad;
  double j0, j1;         // its first two jump times, absolute on the branch
};

struct EpvDev {
  uint64_t n;        // local sites
  uint64_t g0;       // global index of local site 0
  uint64_t n_global; // genome length (for the two global boundary special cases)
  uint32_t B;        // branches = n_nodes - 1
  uint32_t C;        // jump capacity per (site, branch)
  uint32_t N;        // nodes
  epv_meta_t *meta;
  double *jumps;
  uint8_t *sel;
  double *tri;
  double *prop_llr;   // [phase_cap] q(old)-q(new) of the pending proposal (propose -> accept kernel)
  uint8_t *prop_flag; // [phase_cap] 1 = the pending proposal overflowed the capacity
  uint64_t *prop_states;  // [B][phase_cap][W] sampled segment end states of the pending proposal, 1 bit each
  unsigned long long *tasks;  // dirty (branch<<40 | site) pairs of the current colour phase,
                              // EPV_SHARDS x 2 regions of task_cap entries each (four buckets by
                              // segment count: each region is filled from both ends)
  uint64_t task_cap;
  uint64_t phase_cap;  // max sites of one colour phase; the hand-over arrays are indexed by
                       // the phase-local thread id (site = s0 + 3*tid) so they are written densely
  uint32_t *alist;     // [EPV_SHARDS][alist_cap] phase-local ids of the sites whose proposal differs
                       // from their current path (the accept kernel's work list)
  uint64_t alist_cap;
  // segment-parallel jump sampling (epv_jumps2.h): per counter shard a list of dirty segments,
  // the results of their searches, and a list of the branches they belong to
  struct EpvSegTask *segs;     // [EPV_SHARDS][seg_cap]
  struct EpvSegOut *segout;    // [EPV_SHARDS][seg_cap]
  unsigned long long *btasks;  // [EPV_SHARDS][btask_cap] site | branch << 40 | dirty segments << 52 | end state << 59
  uint32_t *bfirst;            // [EPV_SHARDS][btask_cap] index of the branch's first dirty segment
  uint64_t seg_cap, btask_cap;
  uint32_t W;        // 64-bit words per (site, branch) in prop_states = ceil((2C+1)/64)
  uint32_t flags;    // EPV_OPT_* (epv_set_options)
  const EpvModelConst *model;  // device copy
  const uint32_t *parent;      // [N]
  const uint32_t *subtree;     // [N]
  const double *blen;          // [N]
};

// Site-independent model (IndepSite.cpp): per-node constants evaluated once on the host
// with epv_exp -- every site shares the two rates, so the per-site kernels need no
// transcendental at all.  2x2 matrices row-major {00,01,10,11}.
struct EpvIndepConst {
  double P[4];                     // continuous_time_trans_prob_mat(r0, r1, branch length)
  double J0[4], J1[4], D0[4], D1[4];  // expectation_J / expectation_D
};

// task words of the jump kernels: branch << 40 | site; a one-segment branch may carry what the kernel
// would otherwise load in the upper bits (epv_flush_tasks)
#define EPV_TASK_COMPACT (1ull << 63)
#define EPV_TASK_SELP_SHIFT 62     /* buffer the proposal is written to */
#define EPV_TASK_START_SHIFT 61    /* its start state */
#define EPV_TASK_SAMPLED_SHIFT 60  /* the sampled end state of its only segment */
#define EPV_TASK_LINIT_SHIFT 59    /* start states of the left / right neighbour's path on the branch */
#define EPV_TASK_RINIT_SHIFT 58

// run-time options (epv_set_options; the same values as the public header)
#define EPV_FLAG_REFERENCE_PROPOSAL_RATIO 1u  /* evaluate q(old)/q(new) with the reference's sums */
#define EPV_FLAG_FORWARD_REJECTION 2u         /* state-changing segments by forward rejection too */
#define EPV_FLAG_SAMPLE_ROOT 4u               /* SingleSiteSampler::SAMPLE_ROOT: propose the root state too (reference arithmetic kernels) */

// counters[] slots
enum { EPV_CNT_ACCEPT = 0, EPV_CNT_OVERFLOW = 1, EPV_CNT_COOP = 2, EPV_CNT_TASKS = 3, EPV_CNT_TASKS2 = 4,
       EPV_CNT_ALIST0 = 5, EPV_CNT_ALIST1 = 6,   // accept-list lengths, double-buffered by phase parity
       EPV_CNT_SEG = 7,                          // dirty segments (low word) and their branches (high word)
       EPV_CNT_N = 8 };
// Every counter is sharded 64 ways with a 128-byte stride (one device-scope atomic word
// saturates near 90 ops/us; 5000 waves hitting ONE word would serialise for ~60 us).
// A block uses shard (blockIdx.x & 63); the host sums the shards.
#define EPV_SHARDS 64u
#define EPV_SHARD_STRIDE 16u /* u64 words = 128 B */
#define EPV_CNT_WORDS (EPV_CNT_N * EPV_SHARDS * EPV_SHARD_STRIDE)
#define EPV_CNT_IDX(kind, shard) (((kind) * EPV_SHARDS + (shard)) * EPV_SHARD_STRIDE)

#endif
