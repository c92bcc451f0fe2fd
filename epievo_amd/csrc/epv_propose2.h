#ifndef EPV_PROPOSE2_H
#define EPV_PROPOSE2_H
// epv_propose2.h -- the proposal kernel of a colour phase, second generation (included by
// epv_kernels.h).  Same contract as epv_mh_propose_kernel<., false> -- pruning
// (SingleSiteSampler.cpp:116-157) and downward sampling of the segment end states (:180-255) of
// one site per lane, bit for bit the numbers of the oracle's parallel rung -- reorganised
// around what the lanes of a wave have in common:
//
//  * ~90 % of the (site, branch) pairs on a short tree have ONE segment (no neighbour jump on
//    that branch).  Its transition matrix depends only on (branch, left state, right state):
//    epv_segtab_kernel evaluates the 4 B matrices once per reset (the very expressions of the
//    per-segment code, so the bits are the same) and the lanes read them from LDS -- no exp,
//    no division.
//  * the other ("heavy") branches drove the cost of the first kernel: a wave iterated
//    max_lanes(K) times over exp + divisions although the average lane has 1.1 segments
//    (lane utilisation 44 %, profiles/r02b_pmc_valu_tree.csv).  Here the lanes only LIST their
//    heavy segments (length, context, Philox address) in LDS; the wave then evaluates the list
//    densely, one segment per lane -- matrices, no-jump bounds and the segment's Philox block,
//    none of which depends on the recursion state -- and the two recursions (Felsenstein
//    partials upwards, end states downwards) read finished 64-byte records: their divergent
//    loops are a handful of FMAs (and one division) per iteration.
//  * a proposal that EQUALS the current path on every branch (no jumps on either, same states:
//    ~70 % of the sites of tree.nwk) is accepted with probability one and changes nothing --
//    the likelihoods it would recompute are the cached ones.  Such a site is counted here and
//    never reaches the accept kernel; all other sites are appended to a compact list that
//    epv_mh_accept_kernel walks with dense lanes.
//
// LDS per wave: constants, the matrix table, the node table regA[N][64] (record offset of the
// branch above the node | proposal end state << 31) and a pool of doubles holding, for the
// lanes that run in this round, their Felsenstein records {p0, p1} (K per branch, +1 for the
// q of an internal node) followed by the wave's heavy-segment records.  GPOOL = true keeps the
// pool in a per-wave slab of global memory (large trees), records interleaved by lane.

struct EpvSegRec {      // one segment, everything the recursions need
  double P00, P11;      // continuous_time_trans_prob_mat (pruning form: h = 1 / exp(len (r0 + r1)))
  double PT00, PT10;    // get_trans_prob(len, prev, 0) for prev = 0, 1 (h = exp(-len (r0 + r1)))
  double nb0, nb1;      // no-jump bounds of trial 1 for a segment that waits in state 0 / 1
  double u_end, u_first;  // the segment's Philox block: end-state uniform, trial 1's first draw
  double len;           // segment length (kept for the jump tasks)
  uint64_t info;        // trip0 | owner lane << 3 | node << 9 | k << 21
};
#define EPV_SEGTAB_DBL 6u   /* table entry = the first six fields */
#define EPV_HREC 10u        /* doubles per heavy-segment record: the eight fields, length, address word */
#define EPV_HREC_SHORT 8u   /* ... when nothing reads length and address word after the dense evaluation
                               (sequential jump kernel): they are parked in the slots of the two uniforms */

__device__ __forceinline__ void epv_seg_matrices(double len, double r0, double r1, double out[6]) {
  const double denom = r0 + r1;
  const double h = 1.0 / epv_exp(len * (r0 + r1));     // ContinuousTimeMarkovModel.cpp:143-161
  out[0] = (r0 * h + r1) / denom;
  out[1] = (r0 + r1 * h) / denom;
  const double h2 = epv_exp(-len * (r0 + r1));         // :116-125, shared by both start states
  out[2] = gtp(r0, r1, h2, denom, 0u, 0u);
  out[3] = gtp(r0, r1, h2, denom, 1u, 0u);
  out[4] = nojump_bound(len * r0);
  out[5] = nojump_bound(len * r1);
}

// tab[(b * 4 + 2 * left_state + right_state) * 6 ..]: the single segment of branch b (length =
// the branch length) in each of the four neighbour contexts
__global__ void epv_segtab_kernel(EpvDev S, double *tab) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= S.B * 4u) return;
  const uint32_t b = t >> 2, li = (t >> 1) & 1u, ri = t & 1u;
  const double *rates = reinterpret_cast<const double *>(S.model);
  const uint32_t trip0 = 4u * li + ri;
  const double len = S.blen[b + 1u] - 0.0;
  double m[6];
  epv_seg_matrices(len, rates[trip0], rates[trip0 | 2u], m);
#pragma unroll
  for (int i = 0; i < 6; ++i) tab[(size_t)t * EPV_SEGTAB_DBL + i] = m[i];
}

// Append the dirty (site, branch) pairs of the last <= 64 branches to the task lists of this
// block's shard -- a WAVE-WIDE operation (lanes without dirty bits pass zeros).  Four buckets by
// segment count so that the waves of the jumps kernel, which take consecutive tasks, hold lanes
// with similar loop counts: a shard has two regions, each filled from BOTH ends (counts packed
// lo/hi in one word) -- region 0: K = 1 from the front, K = 2 from the back; region 1: K = 3
// from the front, K >= 4 from the back.  One atomic per wave and region reserves the slots.
// A task word is branch << 40 | site.  A ONE-SEGMENT branch (no neighbour jump) may carry everything the
// jump kernels would otherwise load -- buffer of the proposal, start state, sampled end state, the
// neighbours' start states -- in its upper bits (EPV_TASK_COMPACT, epv_device.h): epv_mh_jumps1_kernel
// then starts its trials one round trip after reading the word.  `compact` marks those branches; selP,
// xstart, xsamp, xL, xR are the values, one bit per branch like `dirty`.
__device__ __forceinline__ void epv_flush_tasks(const EpvDev &S, unsigned long long *counters,
                                                unsigned long long dirty, unsigned long long multi,
                                                unsigned long long deep, uint32_t b, uint64_t site, int lane,
                                                uint32_t shard, unsigned long long compact = 0ull, uint32_t selP = 0u,
                                                unsigned long long xstart = 0ull, unsigned long long xsamp = 0ull,
                                                unsigned long long xL = 0ull, unsigned long long xR = 0ull) {
#pragma unroll
  for (uint32_t reg = 0; reg < 2u; ++reg) {
    const unsigned long long mine_reg = reg ? (dirty & deep) : (dirty & ~deep);
    const uint32_t mineB = (uint32_t)__popcll(mine_reg & multi);
    const uint32_t mineA = (uint32_t)__popcll(mine_reg) - mineB;
    const uint32_t inclA = wave_incl_scan_u32(mineA), inclB = wave_incl_scan_u32(mineB);
    const uint32_t totalA = epv_bcast(inclA, 63), totalB = epv_bcast(inclB, 63);
    if (totalA | totalB) {
      unsigned long long base = 0ull;
      if (lane == 0)
        base = atomicAdd(&counters[EPV_CNT_IDX(reg ? EPV_CNT_TASKS2 : EPV_CNT_TASKS, shard)],
                         (unsigned long long)totalA | ((unsigned long long)totalB << 32));
      const uint32_t baseA = epv_bcast((uint32_t)base, 0), baseB = epv_bcast((uint32_t)(base >> 32), 0);
      unsigned long long *region = S.tasks + ((unsigned long long)shard * 2u + reg) * S.task_cap;
      unsigned long long slotA = (unsigned long long)baseA + (inclA - mineA);
      unsigned long long slotB = S.task_cap - 1ull - ((unsigned long long)baseB + (inclB - mineB));
      unsigned long long d = mine_reg;
      while (d) {
        const uint32_t bit = (uint32_t)(__ffsll((long long)d) - 1);
        d &= d - 1ull;
        unsigned long long t = ((unsigned long long)((b & ~63u) + bit) << 40) | site;
        if ((compact >> bit) & 1ull)
          t |= EPV_TASK_COMPACT | ((unsigned long long)selP << EPV_TASK_SELP_SHIFT) |
               (((xstart >> bit) & 1ull) << EPV_TASK_START_SHIFT) | (((xsamp >> bit) & 1ull) << EPV_TASK_SAMPLED_SHIFT) |
               (((xL >> bit) & 1ull) << EPV_TASK_LINIT_SHIFT) | (((xR >> bit) & 1ull) << EPV_TASK_RINIT_SHIFT);
        if ((multi >> bit) & 1ull) region[slotB--] = t;
        else region[slotA++] = t;
      }
    }
  }
}

#ifndef EPV_P2_DENSE_LIST
#define EPV_P2_DENSE_LIST 1
#endif
#ifndef EPV_PROPOSE2_WAVES
#define EPV_PROPOSE2_WAVES 3
#endif
// the fused phase emits its segment / branch tasks in one dense pass behind the node loop (1) or, as the
// separate kernels do, node by node inside it (0; A/B runs)
#ifndef EPV_FUSED_DENSE_EMIT
#define EPV_FUSED_DENSE_EMIT 1
#endif

// -DEPV_P2_PROFILE: wave-time per section of the kernel, read back by tools/p2_profile.py
#ifdef EPV_P2_PROFILE
// one row per wave and plain stores: atomics on shared counters would serialise the waves and
// show up as waiting time wherever the compiler put the next s_waitcnt
#define EPV_P2_PROF_ROWS 8192
__device__ unsigned long long epv_p2_prof[16 * EPV_P2_PROF_ROWS];
#define P2_MARK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); t_acc_[i] += t_ - t_prev_; t_prev_ = t_; } while (0)
#else
#define P2_MARK(i) do {} while (0)
#endif
// FUSED (small launches, LDS pool only): the wave keeps the segment and branch lists of ITS 64 sites
// in a private region of global memory and runs the whole colour phase itself -- proposal, then
// the segment search, the assembly of its dirty branches and the acceptance of its sites with the
// device functions of epv_seg_search_kernel / epv_seg_assemble_kernel / epv_mh_accept_kernel.
// Every site of a colour is independent of the others, so nothing in a phase needs more than a
// wave; what the separate kernels buy is dense lanes, and what they cost is one ~20 us wave chain
// per kernel, which is all there is to pay when a launch has fewer waves than the chip has SIMDs.
struct EpvFused {
  EpvSegTask *segs;            // [waves][seg_cap]
  EpvSegOut *outs;             // [waves][seg_cap]
  unsigned long long *bt;      // [waves][bt_cap]
  uint32_t *bfirst;            // [waves][bt_cap]
  uint32_t seg_cap, bt_cap;    // per wave: 64 B (2C+1) and 64 B -- the worst case, so nothing overflows
  uint32_t meta_cache;         // accept stage: meta words of the five columns in LDS (B <= 8)
  uint32_t lanes;              // sites per wave
  uint32_t grouped_rounds;     // rounds of the grouped search of short segment lists (0 = off)
};

template <bool GPOOL, bool SEG, bool FUSED>
__global__ __launch_bounds__(256, EPV_PROPOSE2_WAVES) void epv_mh_propose2_kernel(
    EpvDev S, uint32_t colour, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep, uint64_t first,
    uint64_t last, uint64_t own_first, uint64_t own_last, uint32_t pool_dbl, uint32_t list_cap,
    uint32_t parity, unsigned long long *counters, double *gpool, const double *segtab, EpvFused F) {
  static_assert(!FUSED || (SEG && !GPOOL), "the fused phase emits segments and keeps its pool in LDS");
  // SEG: true = dirty SEGMENTS go to the segment-parallel jump kernels (epv_jumps2.h); 0 = dirty
  // branches go to epv_mh_jumps_kernel's bucketed lists
  // pool_dbl: LDS pool -- doubles per wave; GPOOL -- record ROWS per lane (list_cap heavy records
  // behind them; unused for the LDS pool, where records and list share pool_dbl)
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  constexpr uint32_t HREC = SEG ? EPV_HREC : EPV_HREC_SHORT, LEN_AT = HREC - 2u, INFO_AT = HREC - 1u;
#ifdef EPV_P2_PROFILE
  unsigned long long t_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif
  const uint32_t const_dbl = (20u + S.N + 1u) & ~1u;
  const uint32_t tab_dbl = S.B * 4u * EPV_SEGTAB_DBL;
  const uint32_t regA_dbl = ((S.N * 64u + 1u) / 2u + 1u) & ~1u;
  // meta words of the three columns of every lane, + the column two sites left of lane 0 and two
  // sites right of the last lane (the acceptance stage's outer triples; the other lanes' outer
  // columns ARE their neighbour lanes' inner ones: lane l + 1 sits three sites further)
  const uint32_t mc_dbl = ((3u * 64u + 2u) * S.B * (uint32_t)sizeof(epv_meta_t) + 15u) / 16u * 2u;
  // a block holds blockDim.x / 64 waves: constants and the matrix table once, then per wave the
  // node table, the meta cache and the pool
  const uint32_t wave_id = threadIdx.x >> 6;
  // the counter shard and list regions are per WAVE (as if every wave were its own block)
  const uint32_t my_shard = (blockIdx.x * (blockDim.x >> 6) + wave_id) & (EPV_SHARDS - 1u);
  const uint32_t wave_dbl = regA_dbl + mc_dbl + (GPOOL ? 0u : ((pool_dbl + 1u) & ~1u));
  double *s_const = s_mem;
  double *s_tab = s_mem + const_dbl;
  double *s_wave = s_mem + const_dbl + tab_dbl + (size_t)wave_id * wave_dbl;
  uint32_t *regA = reinterpret_cast<uint32_t *>(s_wave);
  // s_meta[(which * B + b) * 64 + lane], which = 0 left, 1 right, 2 this site: fetched ONCE, in one
  // batch of independent loads; the four passes below then read LDS instead of paying a global
  // round trip per branch (the kernel is bound by memory latency, not by issue)
  epv_meta_t *s_meta = reinterpret_cast<epv_meta_t *>(s_wave + regA_dbl);
  epv_meta_t *s_edge = s_meta + 3u * S.B * 64u;     // [0 .. B): left of lane 0, [B .. 2B): right of the last valid lane
  const int lane = epv_lane();
  // GPOOL: rows of 64 interleaved records (row r of lane l at (r * 64 + l) * 2 doubles), then
  // the flat heavy list; LDS: records packed by a wave prefix sum, the heavy list behind them
  double *pool = GPOOL ? gpool + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave_id) *
                                     ((size_t)pool_dbl * 128u + (size_t)list_cap * EPV_HREC)
                       : s_wave + regA_dbl + mc_dbl;
  const uint64_t gfirst = S.g0 + first;
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)(gfirst % 3u)) % 3u);
  // FUSED: F.lanes (64, 32 or 16) sites per wave -- fewer sites, hence fewer search rounds, per wave
  // when that still leaves SIMDs idle; the other lanes only help in the cooperative stages
  const uint32_t lpw = FUSED ? F.lanes : 64u;
  const uint64_t tid = FUSED ? ((uint64_t)blockIdx.x * (blockDim.x >> 6) + wave_id) * lpw + (uint32_t)(threadIdx.x & 63u)
                             : (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site = s0 + 3u * tid;
  const bool valid = site <= last && (threadIdx.x & 63u) < lpw;
  const uint64_t n = S.n;
  const uint32_t B = S.B;
  const uint32_t gsite = (uint32_t)(S.g0 + site);
  const uint32_t gsite_lane0 = gsite - 3u * (uint32_t)lane;

  // the head of the kernel's dependent chain (sel -> meta -> jumps) starts before the constants are
  // staged, so that the two round trips overlap
  uint32_t selL = 0, selM = 0, selR = 0;
  if (valid) { selL = S.sel[site - 1]; selM = S.sel[site]; selR = S.sel[site + 1]; }
  // FUSED: the buffers of the columns two sites away, for the acceptance stage's outer triples (they
  // belong to other colours and do not change in this phase), in the same batch of loads
  uint32_t selLL = 0, selRR = 0;
  if (FUSED && valid) {
    if (S.g0 + site > 1u) selLL = S.sel[site - 2];
    if (S.g0 + site < S.n_global - 2u) selRR = S.sel[site + 2];
  }
  for (uint32_t i = threadIdx.x; i < tab_dbl; i += blockDim.x) s_tab[i] = segtab[i];
  stage_constants(S, s_const);
  const double *s_rates = s_const;
  const double *s_blen = s_const + 20;

  // plane offsets without per-lane 64-bit multiplies (quarter-rate on the vector unit): the buffer
  // select is a conditional add, the branch stride a wave-uniform product
  const uint64_t Bn = (uint64_t)B * n, Cn = (uint64_t)S.C * n;
  const uint64_t mbaseL = (selL ? Bn : 0ull) + (site - 1), mbaseR = (selR ? Bn : 0ull) + (site + 1);
  const uint64_t mbaseM = (selM ? Bn : 0ull) + site;
  const uint64_t jbaseL = (selL ? Bn * S.C : 0ull) + (site - 1), jbaseR = (selR ? Bn * S.C : 0ull) + (site + 1);
  uint32_t need_rec = 0, heavy = 0;
  if (valid) {
#pragma unroll 4
    for (uint32_t b = 0; b < B; ++b) {
      const uint32_t mL = S.meta[mbaseL + (uint64_t)b * n];
      const uint32_t mR = S.meta[mbaseR + (uint64_t)b * n];
      const uint32_t mM = S.meta[mbaseM + (uint64_t)b * n];
      s_meta[(0u * B + b) * 64u + lane] = (epv_meta_t)mL;
      s_meta[(1u * B + b) * 64u + lane] = (epv_meta_t)mR;
      s_meta[(2u * B + b) * 64u + lane] = (epv_meta_t)mM;
      const uint32_t K = (mL & EPV_NJ_MASK) + (mR & EPV_NJ_MASK) + 1u;
      need_rec += K + (S.subtree[b + 1u] != 1u ? 1u : 0u);
      if (K >= 2u) heavy += K;
    }
  }

  // FUSED: the two outer columns no neighbour lane holds
  const unsigned long long vmask = __ballot(valid);
  const int last_lane = vmask ? 63 - __clzll((long long)vmask) : 0;
  if (FUSED && valid && (lane == 0 || lane == last_lane)) {
    const bool hasLL = S.g0 + site > 1u, hasRR = S.g0 + site < S.n_global - 2u;
    for (uint32_t b = 0; b < B; ++b) {
      if (lane == 0) s_edge[b] = hasLL ? S.meta[(selLL ? Bn : 0ull) + (uint64_t)b * n + (site - 2)] : (epv_meta_t)0;
      if (lane == last_lane) s_edge[B + b] = hasRR ? S.meta[(selRR ? Bn : 0ull) + (uint64_t)b * n + (site + 2)] : (epv_meta_t)0;
    }
  }
  P2_MARK(0);
  // FUSED: lengths of the wave's private lists (wave-uniform), and which lanes' sites await acceptance
  uint32_t f_nseg = 0u, f_nbt = 0u;
  bool f_listed = false;
  const uint64_t f_wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave_id;
  bool pending = valid;
  while (__any(pending)) {
    const uint32_t wantR = pending ? need_rec : 0u, wantH = pending ? heavy : 0u;
    const uint32_t inclR = wave_incl_scan_u32(wantR), inclH = wave_incl_scan_u32(wantH);
    // non-decreasing in the lane index, so the lanes that run are a prefix of the pending ones
    const bool run = pending && (GPOOL ? (need_rec <= pool_dbl && inclH <= list_cap)
                                       : (2u * inclR + HREC * inclH <= pool_dbl));
    const unsigned long long rmask = __ballot(run);
    const int hi_lane = rmask ? 63 - __clzll((long long)rmask) : 0;
    const uint32_t totR = rmask ? epv_bcast(inclR, hi_lane) : 0u, totH = rmask ? epv_bcast(inclH, hi_lane) : 0u;
    constexpr size_t RS = GPOOL ? 128u : 2u;   // doubles between consecutive records of a lane
    double *my = GPOOL ? pool + (size_t)lane * 2u : pool + (size_t)(inclR - wantR) * 2u;
    double *list = GPOOL ? pool + (size_t)pool_dbl * 128u : pool + (size_t)totR * 2u;
    const uint32_t hbase = inclH - wantH;

    P2_MARK(1);
    // ---- 1. list the heavy segments: forward merge of the neighbours' jumps (Segment.cpp:35-79).
    //      The merges of different branches are independent, and a lane has a heavy branch at
    //      about every eighth (lane, node) pair of a short tree: walking the nodes with whichever
    //      lanes are heavy there kept an eighth of the wave busy.  LDS pool: the heavy (lane, node)
    //      pairs are listed first (in the node table, which pruning fills only afterwards) and
    //      then merged one pair per LANE.
    auto merge_branch = [&](uint32_t owner, uint32_t node, uint32_t hcur, uint64_t jl, uint64_t jr) __attribute__((always_inline)) {
      const uint32_t b = node - 1u;
      const uint32_t cL = s_meta[(0u * B + b) * 64u + owner], cR = s_meta[(1u * B + b) * 64u + owner];
      const uint32_t K = (cL & EPV_NJ_MASK) + (cR & EPV_NJ_MASK) + 1u;
      PathRef L, R;
      L.j = S.jumps + jl + (uint64_t)b * Cn; L.nj = cL & EPV_NJ_MASK; L.init = cL >> EPV_INIT_SHIFT;
      R.j = S.jumps + jr + (uint64_t)b * Cn; R.nj = cR & EPV_NJ_MASK; R.init = cR >> EPV_INIT_SHIFT;
      uint32_t trip0 = 4u * L.init + R.init, i = 0, j = 0;
      double seg_start = 0.0;
      double tl = L.nj ? L.j[0] : EPV_INF, tr = R.nj ? R.j[0] : EPV_INF;
      for (uint32_t k = 0; k < K; ++k) {
        const bool last_seg = (k + 1u == K);
        const bool take_left = tl < tr;
        const double seg_end = last_seg ? s_blen[node] : (take_left ? tl : tr);
        double *rec = list + (size_t)(hcur + k) * HREC;
        rec[LEN_AT] = seg_end - seg_start;
        rec[INFO_AT] = epv_u2d((uint64_t)trip0 | ((uint64_t)owner << 3) | ((uint64_t)node << 9) | ((uint64_t)k << 21));
        if (!last_seg) {
          if (take_left) { trip0 ^= 4u; ++i; tl = i < L.nj ? L.j[(uint64_t)i * n] : EPV_INF; }
          else { trip0 ^= 1u; ++j; tr = j < R.nj ? R.j[(uint64_t)j * n] : EPV_INF; }
          seg_start = seg_end;
        }
      }
    };
    if (!GPOOL && EPV_P2_DENSE_LIST && B > 1u) {      // (a single branch: every lane is its own pair)
      // 1a. pair words lane | node << 6 | first record << 18 (an LDS pool holds fewer than 2^14 records)
      uint32_t n_pairs = 0u;
      if (run && heavy)
        for (uint32_t b = 0; b < B; ++b)
          n_pairs += ((s_meta[(0u * B + b) * 64u + lane] | s_meta[(1u * B + b) * 64u + lane]) & EPV_NJ_MASK) ? 1u : 0u;
      const uint32_t inclP = wave_incl_scan_u32(n_pairs);
      const uint32_t totP = epv_bcast(inclP, 63);
      if (n_pairs) {
        uint32_t hcur = hbase, at = inclP - n_pairs;
        for (uint32_t node = 1u; node < S.N; ++node) {
          const uint32_t b = node - 1u;
          const uint32_t K = (s_meta[(0u * B + b) * 64u + lane] & EPV_NJ_MASK) + (s_meta[(1u * B + b) * 64u + lane] & EPV_NJ_MASK) + 1u;
          if (K < 2u) continue;
          regA[at++] = (uint32_t)lane | (node << 6) | (hcur << 18);
          hcur += K;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // 1b. one pair per lane
      for (uint32_t p0 = 0; p0 < totP; p0 += 64u) {
        const uint32_t pidx = p0 + (uint32_t)lane;
        const uint32_t pr = pidx < totP ? regA[pidx] : 0u;
        const uint32_t owner = pr & 63u;
        const uint64_t jl = (uint64_t)__shfl((uint32_t)jbaseL, (int)owner) | ((uint64_t)__shfl((uint32_t)(jbaseL >> 32), (int)owner) << 32);
        const uint64_t jr = (uint64_t)__shfl((uint32_t)jbaseR, (int)owner) | ((uint64_t)__shfl((uint32_t)(jbaseR >> 32), (int)owner) << 32);
        if (pidx < totP) merge_branch(owner, (pr >> 6) & 4095u, pr >> 18, jl, jr);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();     // regA goes back to pruning
    } else if (run && heavy) {
      uint32_t hcur = hbase;
      for (uint32_t node = 1u; node < S.N; ++node) {
        const uint32_t b = node - 1u;
        const uint32_t K = (s_meta[(0u * B + b) * 64u + lane] & EPV_NJ_MASK) + (s_meta[(1u * B + b) * 64u + lane] & EPV_NJ_MASK) + 1u;
        if (K < 2u) continue;
        merge_branch((uint32_t)lane, node, hcur, jbaseL, jbaseR);
        hcur += K;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    P2_MARK(2);
    // ---- 2. evaluate them densely, one segment per lane
    for (uint32_t i = (uint32_t)lane; i < totH; i += 64u) {
      double *rec = list + (size_t)i * HREC;
      const double len = rec[LEN_AT];
      const uint64_t info = epv_d2u(rec[INFO_AT]);
      const uint32_t trip0 = (uint32_t)info & 7u, owner = (uint32_t)(info >> 3) & 63u;
      const uint32_t node = (uint32_t)(info >> 9) & 4095u, k = (uint32_t)(info >> 21);
      double m[6];
      epv_seg_matrices(len, s_rates[trip0], s_rates[trip0 | 2u], m);
      const epv_block2 blk = epv_keyed_block(seed_lo, seed_hi, gsite_lane0 + 3u * owner, sweep, node, k, 0u, 0u);
#pragma unroll
      for (int q = 0; q < 6; ++q) rec[q] = m[q];
      rec[6] = blk.d0;
      rec[7] = blk.d1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    P2_MARK(3);
    bool ident = true;
    if (run) {
      // ---- 3. pruning, reverse pre-order (SingleSiteSampler.cpp:116-157)
      uint32_t off = need_rec, hcur = hbase + heavy;
      for (uint32_t node = S.N - 1u; node >= 1u; --node) {
        const uint32_t b = node - 1u;
        const uint32_t mL = s_meta[(0u * B + b) * 64u + lane], mR = s_meta[(1u * B + b) * 64u + lane];
        const uint32_t K = (mL & EPV_NJ_MASK) + (mR & EPV_NJ_MASK) + 1u;
        const uint32_t sub = S.subtree[node];
        off -= K + (sub != 1u ? 1u : 0u);
        double n0 = 1.0, n1 = 1.0;
        if (sub == 1u) {
          const uint32_t mM = s_meta[(2u * B + b) * 64u + lane];
          const uint32_t leaf_state = (mM >> EPV_INIT_SHIFT) ^ (mM & 1u);
          n0 = leaf_state ? 0.0 : 1.0;
          n1 = leaf_state ? 1.0 : 0.0;
        } else {
          for (uint32_t ch = 1u; ch < sub; ch += S.subtree[node + ch]) {
            const double *a = my + (size_t)(regA[(node + ch) * 64u + lane] & 0x3fffffffu) * RS;
            n0 *= a[0];   // p.front() of the child's branch
            n1 *= a[1];
          }
          double *recq = my + (size_t)(off + K) * RS;
          recq[0] = n0; recq[1] = n1;
        }
        regA[node * 64u + lane] = off;
        if (K == 1u) {
          const double *t = s_tab + (b * 4u + 2u * (mL >> EPV_INIT_SHIFT) + (mR >> EPV_INIT_SHIFT)) * EPV_SEGTAB_DBL;   // (32-bit index: an LDS address)
          const double P00 = t[0], P11 = t[1];
          const double P01 = 1.0 - P00, P10 = 1.0 - P11;
          double *rec = my + (size_t)off * RS;
          rec[0] = P00 * n0 + P01 * n1;
          rec[1] = P10 * n0 + P11 * n1;
        } else {
          hcur -= K;
          for (uint32_t kk = K; kk-- > 0u;) {
            const double *hr = list + (size_t)(hcur + kk) * HREC;
            const double P00 = hr[0], P11 = hr[1];
            const double P01 = 1.0 - P00, P10 = 1.0 - P11;
            const double a = P00 * n0 + P01 * n1;
            const double c = P10 * n0 + P11 * n1;
            double *rec = my + (size_t)(off + kk) * RS;
            rec[0] = a; rec[1] = c;
            n0 = a; n1 = c;
          }
        }
      }
    }

    P2_MARK(4);
    // ---- 4. downward sampling of the segment END STATES (:180-255); the jump times are drawn
    //         by epv_mh_jumps_kernel for the dirty branches only (see epv_kernels.h).  The node
    //         loop is wave-uniform: the task flush inside it is a wave-wide operation.
    unsigned long long dirty = 0ull, multi = 0ull, deep = 0ull;
    const uint32_t root_state = run ? (uint32_t)(s_meta[(2u * B) * 64u + lane] >> EPV_INIT_SHIFT) : 0u;
    uint32_t hcur = hbase;
    for (uint32_t node = 1u; node < S.N; ++node) {
      const uint32_t b = node - 1u;
      // what the wave-wide hand-over below needs from this lane's branch
      uint32_t nds = 0, Kb = 0, st_b = 0, end_b = 0, hrec0 = 0, trip_b = 0;
      unsigned long long w64 = 0ull;
      bool dirty_b = false;
      if (run) {
        const uint32_t mL = s_meta[(0u * B + b) * 64u + lane], mR = s_meta[(1u * B + b) * 64u + lane];
        const uint32_t K = (mL & EPV_NJ_MASK) + (mR & EPV_NJ_MASK) + 1u;
        const uint32_t off = regA[node * 64u + lane];
        const uint32_t par = S.parent[node];
        const uint32_t start_state = (par == 0u) ? root_state : (regA[par * 64u + lane] >> 31);
        const bool leaf = S.subtree[node] == 1u;
        const uint32_t mM = s_meta[(2u * B + b) * 64u + lane];
        uint32_t prev = start_state;
        bool clean = true;
        unsigned long long word = 0ull;
        uint64_t *states = S.prop_states + ((uint64_t)b * S.phase_cap + tid) * S.W;
        double pk0 = my[(size_t)off * RS], pk1 = my[(size_t)off * RS + 1u];
        for (uint32_t k = 0; k < K; ++k) {
          const bool last_seg = (k + 1u == K);
          double nxt0, nxt1;
          if (last_seg && leaf) {          // q of a leaf: the observed state
            const uint32_t leaf_state = (mM >> EPV_INIT_SHIFT) ^ (mM & 1u);
            nxt0 = leaf_state ? 0.0 : 1.0;
            nxt1 = leaf_state ? 1.0 : 0.0;
          } else {
            nxt0 = my[(size_t)(off + k + 1u) * RS];      // p[k+1], or the node's q
            nxt1 = my[(size_t)(off + k + 1u) * RS + 1u];
          }
          double PT0, nb, u_end, u_first;
          if (K == 1u) {
            const double *t = s_tab + (b * 4u + 2u * (mL >> EPV_INIT_SHIFT) + (mR >> EPV_INIT_SHIFT)) * EPV_SEGTAB_DBL;   // (32-bit index: an LDS address)
            PT0 = prev ? t[3] : t[2];
            nb = prev ? t[5] : t[4];
            const epv_block2 blk = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, 0u, 0u, 0u);
            u_end = blk.d0; u_first = blk.d1;
          } else {
            const double *hr = list + (size_t)(hcur + k) * HREC;
            PT0 = prev ? hr[3] : hr[2];
            nb = prev ? hr[5] : hr[4];
            u_end = hr[6]; u_first = hr[7];
          }
          const double p0 = PT0 * nxt0 / (prev ? pk1 : pk0);
          const uint32_t sampled = (u_end > p0) ? 1u : 0u;
          // trial 1's first draw is the other half of the same Philox block: a segment that keeps
          // its state and provably has no jump in trial 1 is clean
          const bool seg_clean = (sampled == prev) && (1.0 - u_first < nb);
          clean = clean && seg_clean;
          nds += seg_clean ? 0u : 1u;
          word |= (unsigned long long)sampled << (k & 63u);
          if (k < 64u) w64 = word;
          if (!FUSED && (k & 63u) == 63u) { states[k >> 6] = word; word = 0ull; }   // (the fused phase hands the states over in its segment tasks)
          prev = sampled;
          pk0 = nxt0; pk1 = nxt1;
        }
        hrec0 = hcur;
        if (K >= 2u) hcur += K;
        if (!FUSED && (K & 63u) && !clean) states[(K - 1u) >> 6] = word;   // only a dirty branch is read back
#if EPV_FUSED_DENSE_EMIT
        // FUSED: proposal end state for the children (bit 31); "dirty" (bit 30) and the branch's first
        // heavy record (bits 14..27) for the emission pass behind this loop, which gives every dirty
        // (site, branch) pair its own lane; the sampled end states of a heavy branch wait in the
        // slot of its first record's end-state uniform, which has been used
        if (FUSED) {
          regA[node * 64u + lane] = off | (prev << 31) | (clean ? 0u : 1u << 30) | (K >= 2u ? hrec0 << 14 : 0u);
          if (!clean && K >= 2u) list[(size_t)hrec0 * HREC + 6u] = epv_u2d(w64);
        } else
#endif
        regA[node * 64u + lane] = off | (prev << 31);  // proposal end state for the children
        // the proposal's meta word (start state, no jumps yet; the fused phase's assembly adds the
        // count) takes the place of the current path's, which this lane has just read for the last
        // time: the acceptance stage finds it there.  (NOT in the node table: a later round of this
        // wave lists its heavy pairs over the finished lanes' rows.)
        if (FUSED) s_meta[(2u * B + b) * 64u + lane] = (epv_meta_t)(start_state << EPV_INIT_SHIFT);
        // same as the current path?  (no jumps on either, same start state)
        ident = ident && clean && mM == (start_state << EPV_INIT_SHIFT);
        dirty_b = !clean;
        Kb = K; st_b = start_state; end_b = prev;
        trip_b = 4u * (mL >> EPV_INIT_SHIFT) + (mR >> EPV_INIT_SHIFT);
      }
      bool old_list = dirty_b;
      if (SEG && !(FUSED && EPV_FUSED_DENSE_EMIT)) {
        // ---- dirty SEGMENTS onto the segment list (one lane each in epv_seg_search_kernel) and the
        //      branch onto the assemble list: one atomic per wave and node reserves both ranges.
        //      A branch with more than 64 segments, or one that finds the lists full, takes the
        //      bucketed lists of the sequential kernel instead.
        const bool cand = dirty_b && Kb <= 64u;
        const uint32_t wS = cand ? nds : 0u, wB = cand ? 1u : 0u;
        const uint32_t inclS = wave_incl_scan_u32(wS), inclB = wave_incl_scan_u32(wB);
        const uint32_t totS = epv_bcast(inclS, 63), totB = epv_bcast(inclB, 63);
        if (totB) {
          const uint32_t shard = my_shard;
          uint32_t bS, bB;
          if (FUSED) {
            bS = f_nseg; bB = f_nbt;
            f_nseg += totS; f_nbt += totB;
          } else {
            unsigned long long base = 0ull;
            if (lane == 0)
              base = atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_SEG, shard)],
                               (unsigned long long)totS | ((unsigned long long)totB << 32));
            bS = epv_bcast((uint32_t)base, 0); bB = epv_bcast((uint32_t)(base >> 32), 0);
          }
          if (cand) {
            const uint64_t i0 = (uint64_t)bS + (inclS - wS), j0 = (uint64_t)bB + (inclB - 1u);
            EpvSegTask *segs = FUSED ? F.segs + f_wave * F.seg_cap : S.segs + (uint64_t)shard * S.seg_cap;
            unsigned long long *btl = FUSED ? F.bt + f_wave * F.bt_cap : S.btasks + (uint64_t)shard * S.btask_cap;
            uint32_t *bfl = FUSED ? F.bfirst + f_wave * F.bt_cap : S.bfirst + (uint64_t)shard * S.btask_cap;
            const uint64_t seg_room = FUSED ? F.seg_cap : S.seg_cap, bt_room = FUSED ? F.bt_cap : S.btask_cap;
            if (i0 + nds <= seg_room && j0 < bt_room) {
              btl[j0] = site | ((unsigned long long)b << 40) | ((unsigned long long)nds << 52) | ((unsigned long long)end_b << 59) |
                        ((unsigned long long)st_b << 60) | ((unsigned long long)(selM ^ 1u) << 61);
              bfl[j0] = (uint32_t)i0;
              uint32_t prev = st_b;
              double tp = 0.0;     // running sum of the segment lengths (SingleSiteSampler.cpp:218)
              uint64_t at = i0;
              for (uint32_t k = 0; k < Kb; ++k) {
                const uint32_t sampled = (uint32_t)(w64 >> k) & 1u;
                double len;
                uint32_t trip0;
                bool seg_clean;
                if (Kb == 1u) {
                  len = s_blen[node];
                  trip0 = trip_b;
                  seg_clean = false;       // the only segment of a dirty branch
                } else {
                  const double *hr = list + (size_t)(hrec0 + k) * HREC;
                  len = hr[LEN_AT];
                  trip0 = (uint32_t)epv_d2u(hr[INFO_AT]) & 7u;
                  seg_clean = (sampled == prev) && (1.0 - hr[7] < (prev ? hr[5] : hr[4]));
                }
                if (!seg_clean) {
                  EpvSegTask t;
                  t.w0 = site | ((unsigned long long)node << 40) | ((unsigned long long)k << 52);
                  t.len = len;
                  t.start = tp;
                  t.w3 = prev | (sampled << 1) | (trip0 << 2);
                  segs[at++] = t;
                }
                tp += len;
                prev = sampled;
              }
              old_list = false;
            } else {
              // no room: blank what this lane reserved inside the lists, fall back
              for (uint64_t i = i0; i < i0 + nds && i < seg_room; ++i) segs[i].len = -1.0;
              if (j0 < bt_room) btl[j0] = ~0ull;
            }
          }
        }
      }
      if (!FUSED && old_list) {      // (the fused phase's lists hold the worst case: never taken there)
        dirty |= 1ull << (b & 63u);
        if (Kb == 2u || Kb >= 4u) multi |= 1ull << (b & 63u);   // four buckets by segment count
        if (Kb >= 3u) deep |= 1ull << (b & 63u);
      }
      if (!FUSED && ((b & 63u) == 63u || node + 1u == S.N)) {
        epv_flush_tasks(S, counters, dirty, multi, deep, b, site, lane, my_shard);
        dirty = multi = deep = 0ull;
      }
    }

    P2_MARK(5);
    // ---- 5. hand-over.  A proposal equal to the current path is accepted with probability one
    //         and changes neither the paths nor the cached likelihoods: count it and be done.
    //         Everything else: start states of the proposal's branches into the other buffer,
    //         and the site onto the accept list of this block's shard.
    const bool to_list = run && !ident;
    if (to_list) {
      for (uint32_t node = 1u; node < S.N; ++node) {
        const uint32_t par = S.parent[node];
        const uint32_t st = (par == 0u) ? root_state : (regA[par * 64u + lane] >> 31);
        S.meta[(selM ? 0ull : Bn) + (uint64_t)(node - 1u) * n + site] = (epv_meta_t)(st << EPV_INIT_SHIFT);
      }
      S.prop_flag[tid] = 0u;
    }
    {
      const uint32_t shard = my_shard;
      f_listed = f_listed || to_list;
      const unsigned long long lm = FUSED ? 0ull : __ballot(to_list);
      if (lm) {
        unsigned long long base = 0ull;
        if (lane == 0)
          base = atomicAdd(&counters[EPV_CNT_IDX(parity ? EPV_CNT_ALIST1 : EPV_CNT_ALIST0, shard)],
                           (unsigned long long)__popcll(lm));
        const uint32_t b0 = epv_bcast((uint32_t)base, 0);
        if (to_list)
          S.alist[(uint64_t)shard * S.alist_cap + b0 + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))] = (uint32_t)tid;
      }
      const unsigned long long am = __ballot(run && ident && site >= own_first && site <= own_last);
      if (am && lane == 0)
        atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_ACCEPT, shard)], (unsigned long long)__popcll(am));
    }
#if EPV_FUSED_DENSE_EMIT
    if (FUSED) {
      // ---- 6. the dirty (site, branch) pairs of this round, ONE LANE EACH: their dirty segments
      //      onto the wave's segment list, the branch onto its assemble list.  (Doing this inside the
      //      node loop cost two wave scans and ~130 instructions per node for the one lane in
      //      fourteen that had a dirty branch there.)  The pair list lives in the Felsenstein records
      //      of this round, which are dead now; everything a pair needs is in the node table, the
      //      staged meta words and the heavy records.
      uint32_t *plist = reinterpret_cast<uint32_t *>(pool);
      uint32_t npair = 0u;
      if (run)
        for (uint32_t node = 1u; node < S.N; ++node) npair += (regA[node * 64u + lane] >> 30) & 1u;
      const uint32_t inclP = wave_incl_scan_u32(npair);
      const uint32_t totP = epv_bcast(inclP, 63);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (npair) {
        uint32_t at = inclP - npair;
        for (uint32_t node = 1u; node < S.N; ++node)
          if ((regA[node * 64u + lane] >> 30) & 1u) plist[at++] = (uint32_t)lane | (node << 6);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint64_t site_lane0 = site - 3u * (uint64_t)lane;
      EpvSegTask *segs_w = F.segs + f_wave * F.seg_cap;
      unsigned long long *btl = F.bt + f_wave * F.bt_cap;
      uint32_t *bfl = F.bfirst + f_wave * F.bt_cap;
      for (uint32_t p0 = 0; p0 < totP; p0 += 64u) {
        const uint32_t pidx = p0 + (uint32_t)lane;
        const bool act = pidx < totP;
        const uint32_t pr = act ? plist[pidx] : 0u;
        const uint32_t owner = pr & 63u, node = act ? pr >> 6 : 1u, b = node - 1u;
        const uint32_t selM_o = (uint32_t)__shfl((int)selM, (int)owner);
        const uint32_t ra = regA[node * 64u + owner];
        const uint32_t end_b = ra >> 31, hrec0 = (ra >> 14) & 0x3fffu;
        const uint32_t st_b = (uint32_t)(s_meta[(2u * B + b) * 64u + owner] >> EPV_INIT_SHIFT);
        const uint32_t mL = s_meta[(0u * B + b) * 64u + owner], mR = s_meta[(1u * B + b) * 64u + owner];
        const uint32_t Kb = (mL & EPV_NJ_MASK) + (mR & EPV_NJ_MASK) + 1u;
        // which segments are dirty (trial 1 does not settle them)
        unsigned long long w64 = end_b, dmask = 1ull;
        uint32_t nds = act ? 1u : 0u;
        if (act && Kb >= 2u) {
          w64 = epv_d2u(list[(size_t)hrec0 * HREC + 6u]);
          dmask = 0ull;
          nds = 0u;
          uint32_t prev = st_b;
          for (uint32_t k = 0; k < Kb; ++k) {
            const double *hr = list + (size_t)(hrec0 + k) * HREC;
            const uint32_t sampled = (uint32_t)(w64 >> k) & 1u;
            const bool seg_clean = (sampled == prev) && (1.0 - hr[7] < (prev ? hr[5] : hr[4]));
            if (!seg_clean) { dmask |= 1ull << k; ++nds; }
            prev = sampled;
          }
        }
        const uint32_t inclS = wave_incl_scan_u32(nds);
        const uint32_t totS = epv_bcast(inclS, 63);
        if (act) {
          const uint64_t osite = site_lane0 + 3u * (uint64_t)owner;
          const uint64_t i0 = (uint64_t)f_nseg + (inclS - nds), j0 = (uint64_t)f_nbt + (uint32_t)lane;
          btl[j0] = osite | ((unsigned long long)b << 40) | ((unsigned long long)nds << 52) | ((unsigned long long)end_b << 59) |
                    ((unsigned long long)st_b << 60) | ((unsigned long long)(selM_o ^ 1u) << 61);
          bfl[j0] = (uint32_t)i0;
          uint32_t prev = st_b;
          double tp = 0.0;     // running sum of the segment lengths (SingleSiteSampler.cpp:218)
          uint64_t at = i0;
          for (uint32_t k = 0; k < Kb; ++k) {
            const uint32_t sampled = (uint32_t)(w64 >> k) & 1u;
            double len;
            uint32_t trip0;
            if (Kb == 1u) {
              len = s_blen[node];
              trip0 = 4u * (mL >> EPV_INIT_SHIFT) + (mR >> EPV_INIT_SHIFT);
            } else {
              const double *hr = list + (size_t)(hrec0 + k) * HREC;
              len = hr[LEN_AT];
              trip0 = (uint32_t)epv_d2u(hr[INFO_AT]) & 7u;
            }
            if ((dmask >> k) & 1ull) {
              EpvSegTask t;
              t.w0 = osite | ((unsigned long long)node << 40) | ((unsigned long long)k << 52);
              t.len = len;
              t.start = tp;
              t.w3 = prev | (sampled << 1) | (trip0 << 2);
              segs_w[at++] = t;
            }
            tp += len;
            prev = sampled;
          }
        }
        f_nseg += totS;
        f_nbt += (totP - p0 < 64u) ? totP - p0 : 64u;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();     // the pool goes to the next round's records
    }
#endif
    pending = pending && !run;
    P2_MARK(6);
  }
  if (FUSED) {
    // the pool is free now: the search's cooperative area, then the accept stage's accumulators
    // and meta cache, are carved from it (plan_p2 sizes it for both)
    const bool nielsen = !(S.flags & EPV_FLAG_FORWARD_REJECTION);
    const EpvSegTask *segs = F.segs + f_wave * F.seg_cap;
    EpvSegOut *outs = F.outs + f_wave * F.seg_cap;
    // lists, start states and flags pass between the lanes of THIS wave through global memory: a
    // workgroup-scope fence (wait for the stores; one CU, one vector L1) is enough -- an agent-scope
    // one writes back and invalidates the XCD's L2, ~30 us each here
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // -DEPV_DBG_SKIP=1 / 2 / 3 leaves out the acceptance / + assembly / + search (instruction counts
    // per stage: tools/inst_count.sh; the chain then never moves, nothing reads what is missing)
#ifndef EPV_DBG_SKIP
#define EPV_DBG_SKIP 0
#endif
    if (f_nseg && EPV_DBG_SKIP < 3) {
      EpvCoop W;
      W.len = pool; W.r0 = pool + 64; W.r1 = pool + 128; W.trunc = pool + 192; W.tj = pool + 256;
      uint32_t *u = reinterpret_cast<uint32_t *>(pool + 384);
      W.misc = u; W.gsite = u + 64; W.tbase = u + 128; W.nk = u + 192; W.res = u + 256; W.tw = u + 320; W.mm = u + 384;
      // a short list: several lanes per segment first (epv_seg_search_grouped); the wave-wide search
      // takes whatever is longer, and whatever a few rounds of that leave open
      bool done = false;
      if (f_nseg <= 32u && F.grouped_rounds) {
        const uint32_t G = 64u / f_nseg > 8u ? 8u : 64u / f_nseg;
        done = epv_seg_search_grouped(S, s_rates, segs, outs, f_nseg, G, F.grouped_rounds, seed_lo, seed_hi, sweep, nielsen);
      }
      if (!done) epv_seg_search_wave(S, s_rates, W, segs, outs, f_nseg, 0u, 64u, seed_lo, seed_hi, sweep, nielsen);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
      P2_MARK(7);
      for (uint32_t i = (uint32_t)lane; i < (EPV_DBG_SKIP < 2 ? f_nbt : 0u); i += 64u)
        epv_seg_assemble_one(S, s_rates, segs, outs, F.bt[f_wave * F.bt_cap + i], F.bfirst[f_wave * F.bt_cap + i], s0,
                             seed_lo, seed_hi, sweep, nielsen, s_meta + 2u * B * 64u, site - 3u * (uint64_t)lane);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
      P2_MARK(8);
    }
    // ---- acceptance.  A site's ratio needs the three triples around it (log_accept_rate,
    //      SingleSiteSampler.cpp:409-429); they are independent, so every (listed site, triple)
    //      pair gets its own lane -- a third of the chain, and on a short tree the lanes whose
    //      proposal equalled their path would idle anyway.  pool: own[64] | res[192] | D[8][64] |
    //      J[8][64] | meta words [3][B][64]
    bool accepted = false, overflowed = false;
    {
      uint32_t *s_own = reinterpret_cast<uint32_t *>(pool);
      double *s_res = pool + 32;
      AccLds A;
      A.d = pool + 224 + lane; A.stride = 64u;
      A.j = reinterpret_cast<uint32_t *>(pool + 736) + lane;
      epv_meta_t *mc = reinterpret_cast<epv_meta_t *>(pool + 992) + lane;
      const unsigned long long lmask = EPV_DBG_SKIP >= 1 ? 0ull : __ballot(f_listed);
      const uint32_t n_listed = (uint32_t)__popcll(lmask);
      const uint32_t my_rank = (uint32_t)__popcll(lmask & ((1ull << lane) - 1ull));
      // the owner publishes what its three task lanes need -- the five columns' buffers (known
      // since the kernel's head) and the capacity flag the assembly may have raised -- and starts
      // its own loads and its accept uniform before the tasks, whose waits then hide them
      bool ovf = false;
      double llh_l = 0.0, llh_m = 0.0, llh_r = 0.0, u_acc = 0.0;
      if (f_listed) {
        ovf = S.prop_flag[tid] != 0;
        llh_l = S.tri[site - 1]; llh_m = S.tri[site]; llh_r = S.tri[site + 1];
        s_own[my_rank] = (uint32_t)lane | (selLL << 8) | (selL << 9) | ((selM ^ 1u) << 10) | (selR << 11) | (selRR << 12) |
                         ((ovf ? 1u : 0u) << 13);
        u_acc = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, 0u, 0u, 0u, 0u).d0;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint64_t site_lane0 = site - 3u * (uint64_t)lane;
      for (uint32_t t = (uint32_t)lane; t < 3u * n_listed; t += 64u) {
        const uint32_t r = t / 3u, w = t - 3u * r, ow = s_own[r], o = ow & 63u;
        const uint64_t osite = site_lane0 + 3u * (uint64_t)o;
        const uint64_t g = S.g0 + osite;
        const bool skip = ((ow >> 13) & 1u) || (w == 0u && !(g > 1u)) || (w == 2u && !(g < S.n_global - 2u));
        double v = 0.0;
        if (!skip) {
          const uint64_t c = osite - 1u + (uint64_t)w;            // centre column of this triple
          // buffers of the three columns (bits 8.. of the owner's word: LL, L, proposal, R, RR)
          const uint32_t bl = (ow >> (8u + w)) & 1u, bm = (ow >> (9u + w)) & 1u, br = (ow >> (10u + w)) & 1u;
#ifdef EPV_ACC_GLOBAL_META
          if (F.meta_cache) {
            const uint32_t B2 = S.B;
#pragma unroll 4
            for (uint32_t b = 0; b < B2; ++b) {
              const epv_meta_t m0 = S.meta[meta_idx(S, bl, b, c - 1u)];
              const epv_meta_t m1 = S.meta[meta_idx(S, bm, b, c)];
              const epv_meta_t m2 = S.meta[meta_idx(S, br, b, c + 1u)];
              mc[(0u * B2 + b) * 64u] = m0;
              mc[(1u * B2 + b) * 64u] = m1;
              mc[(2u * B2 + b) * 64u] = m2;
            }
#else
          if (F.meta_cache) {
            // the meta words of the triple's columns without a global round trip: the neighbours'
            // current paths were staged at the head of the kernel (they belong to other colours and
            // do not change in this phase), the proposal's sit where the site's own column was
            const uint32_t B2 = S.B;
#pragma unroll 4
            for (uint32_t b = 0; b < B2; ++b) {
              const epv_meta_t mP = s_meta[(2u * B2 + b) * 64u + o];
              const epv_meta_t mL = s_meta[(0u * B2 + b) * 64u + o], mR = s_meta[(1u * B2 + b) * 64u + o];
              epv_meta_t m0, m1, m2;
              if (w == 0u) { m0 = o > 0u ? s_meta[(1u * B2 + b) * 64u + (o - 1u)] : s_edge[b]; m1 = mL; m2 = mP; }
              else if (w == 1u) { m0 = mL; m1 = mP; m2 = mR; }
              else { m0 = mP; m1 = mR; m2 = o < (uint32_t)last_lane ? s_meta[(0u * B2 + b) * 64u + (o + 1u)] : s_edge[B2 + b]; }
              mc[(0u * B2 + b) * 64u] = m0;
              mc[(1u * B2 + b) * 64u] = m1;
              mc[(2u * B2 + b) * 64u] = m2;
            }
#endif
            v = triple_llh_cached(S, s_const, s_blen, mc, 64u, 0u, bl, c - 1u, 1u, bm, c, 2u, br, c + 1u, A);
          } else {
            v = triple_llh(S, s_const, s_blen, bl, c - 1u, bm, c, br, c + 1u, A);
          }
        }
        s_res[t] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (f_listed && EPV_DBG_SKIP < 1) {
        // the owner: Metropolis_Hastings_site :510-533 with the three values (the arithmetic of
        // epv_accept_site, term by term)
        double llr = (S.flags & EPV_FLAG_REFERENCE_PROPOSAL_RATIO) ? S.prop_llr[tid] : 0.0;
        const double llh_l_orig = llh_l, llh_r_orig = llh_r;
        if (!ovf) {
          const uint64_t g = S.g0 + site;
          if (g > 1u) llh_l = s_res[3u * my_rank];
          llh_m = s_res[3u * my_rank + 1u];
          if (g < S.n_global - 2u) llh_r = s_res[3u * my_rank + 2u];
        }
        llr += (llh_l + llh_r - llh_l_orig - llh_r_orig);
        bool acc = (llr >= 0.0) || (u_acc < epv_exp(llr));
        if (ovf) { acc = false; overflowed = true; }
        if (acc) {
          S.sel[site] = (uint8_t)(selM ^ 1u);
          S.tri[site - 1] = llh_l;
          S.tri[site] = llh_m;
          S.tri[site + 1] = llh_r;
          accepted = site >= own_first && site <= own_last;
        }
      }
    }
    P2_MARK(9);
    const unsigned long long am = __ballot(accepted), om = __ballot(overflowed);
    if (lane == 0) {
      if (am) atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_ACCEPT, my_shard)], (unsigned long long)__popcll(am));
      if (om) atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_OVERFLOW, my_shard)], (unsigned long long)__popcll(om));
      if (f_nbt) atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_COOP, my_shard)], (unsigned long long)f_nbt);
    }
  }
#ifdef EPV_P2_PROFILE
  if (epv_lane() == 0) {
    unsigned long long *row = epv_p2_prof + 16u * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % EPV_P2_PROF_ROWS);
    for (int q = 0; q < 15; ++q) row[q] += t_acc_[q];
    row[15] += 1ull;
  }
#endif
}

#endif
