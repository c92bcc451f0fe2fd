#ifndef EPV_JUMPS2_H
#define EPV_JUMPS2_H
// epv_jumps2.h -- jump times of a proposal, SEGMENT-parallel (included by epv_kernels.h).
//
// epv_mh_jumps_kernel gives every dirty (site, branch) pair a lane that walks the branch's
// segments one after the other; a wave runs as long as its slowest lane, and on a long branch
// (T = 1: three segments per branch, every one of them dirty, several trials each) that chain
// is the whole kernel: 109 us for 33 000 branches (profiles/r02_bench_config2_pair_n1e5.json).
// But once epv_mh_propose2_kernel has drawn the end states, the segments of a branch are
// INDEPENDENT: the end-conditioned sampler of segment k is keyed by (site, node, k) and needs
// only (length, start state, end state, context).  What ties them together is bookkeeping --
// where in the path's jump slots a segment's times go, and whether the path overflows its
// capacity on the way -- and that is cheap:
//
//   epv_seg_search_kernel    one lane per DIRTY segment: the first non-failing trial t*
//                            (EndCondSampling.cpp:466-509, :576-617 -- the very scan of the
//                            sequential kernel, wave-cooperative beyond EPV_INLINE_TRIALS), its
//                            jump count, its first two jump times, and M = the most jumps any
//                            trial t <= t* made;
//   epv_seg_assemble_kernel  one lane per dirty branch: its segments' results in order.  The
//                            sequential sampler rejects the proposal (capacity overflow) exactly
//                            when some trial up to the winner needs more slots than are left,
//                            i.e. when M exceeds the room at that segment -- so the decision, the
//                            counts and the times are those of the sequential walk, bit for bit.
//                            Times come from the search (<= 2 jumps) or from replaying trial t*
//                            (random-access RNG) straight into the path.
//
// Branches with more than 64 segments, or that find the lists full, stay with
// epv_mh_jumps_kernel (launched behind these two with a small grid; it finds empty lists
// otherwise).

#ifndef EPV_SEARCH_WAVES
#define EPV_SEARCH_WAVES 2
#endif

// cooperative-search area of ONE wave (LDS): task slots by rank and per-lane trial results
struct EpvCoop {
  double *len, *r0, *r1, *trunc, *tj;     // [64] each, tj [128]
  uint32_t *misc, *gsite, *tbase, *nk, *res, *tw, *mm;   // [64] each
};
#define EPV_COOP_BYTES (64u * (6u * 8u + 7u * 4u))

// the wave's share of a segment list: entries base_first + lane, + stride, ... below n_seg
__device__ __forceinline__ void epv_seg_search_wave(const EpvDev &S, const double *s_rates, const EpvCoop &W,
                                                    const EpvSegTask *segs, EpvSegOut *outs, uint64_t n_seg,
                                                    uint64_t base_first, uint64_t stride, uint32_t seed_lo,
                                                    uint32_t seed_hi, uint32_t sweep, bool nielsen) {
  const int lane = epv_lane();
  double *c_len = W.len, *c_r0 = W.r0, *c_r1 = W.r1, *c_trunc = W.trunc, *c_tj = W.tj;
  uint32_t *c_misc = W.misc, *c_gsite = W.gsite, *c_tbase = W.tbase, *c_nk = W.nk, *c_res = W.res, *c_tw = W.tw,
           *c_mm = W.mm;
  for (uint64_t base = base_first; base < n_seg; base += stride) {
    const uint64_t i = base + (unsigned)lane;
    EpvSegTask t;
    t.w0 = 0ull; t.len = -1.0; t.start = 0.0; t.w3 = 0ull;
    if (i < n_seg) t = segs[i];
    const bool active = t.len >= 0.0;
    const uint32_t gsite = (uint32_t)(S.g0 + (t.w0 & 0xffffffffffull));
    const uint32_t node = (uint32_t)(t.w0 >> 40) & 4095u, k = (uint32_t)(t.w0 >> 52);
    const uint32_t prev = (uint32_t)t.w3 & 1u, sampled = (uint32_t)(t.w3 >> 1) & 1u, trip0 = (uint32_t)(t.w3 >> 2) & 7u;
    const double len = t.len, r0 = s_rates[trip0], r1 = s_rates[trip0 | 2u];
    // sample_trunc_exp's 1 - exp(-rate_a T) (EndCondSampling.cpp:577-580), state changes only
    double trunc = 0.0;
    if (active && sampled != prev) trunc = 1.0 - epv_exp(-(prev ? r1 : r0) * len);
    uint32_t cnt = 0, tstar = 0, maxm = 0, tbase = EPV_INLINE_TRIALS + 1u;
    double jt[2] = {0.0, 0.0};
    bool pend = false;
    if (active) {
      // unlimited room: whether the path overflows is decided when the branch is assembled
      const int oc = scan_trials(seed_lo, seed_hi, gsite, sweep, node, k, 1u, EPV_INLINE_TRIALS, prev, sampled, len,
                                 r0, r1, trunc, 0xffffffffu, jt, 1u, 2u, t.start, tstar, cnt, nielsen, &maxm);
      pend = oc != TRIAL_OK;
    }
    // wave-cooperative search for t > EPV_INLINE_TRIALS, as in epv_mh_jumps_kernel: the P pending
    // segments share the 64 lanes, each helper lane scans EPV_COOP_WINDOW consecutive trials;
    // the lowest non-failing trial wins.  M also takes the failed trials of the windows BEFORE
    // the winner's (they are all below t*) and of the winner's own window up to t*.
    unsigned long long todo = __ballot(pend);
    while (todo) {
      const uint32_t P = (uint32_t)__popcll(todo);
      const uint32_t lg = 31u - (uint32_t)__clz((int)(64u / P));
      const uint32_t G = 1u << lg;
      const uint32_t rank = (uint32_t)__popcll(todo & ((1ull << lane) - 1ull));
      if (pend) {
        c_len[rank] = len; c_r0[rank] = r0; c_r1[rank] = r1; c_trunc[rank] = trunc;
        c_misc[rank] = prev | (sampled << 1);
        c_gsite[rank] = gsite;
        c_tbase[rank] = tbase;
        c_nk[rank] = (node << 12) | k;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t tj = (uint32_t)lane >> lg, tr_ = (uint32_t)lane & (G - 1u);
      int oc = TRIAL_FAIL;
      if (tj < P) {
        const uint32_t misc = c_misc[tj], nk = c_nk[tj];
        const uint32_t t0 = c_tbase[tj] + tr_ * EPV_COOP_WINDOW;
        uint32_t njt = 0, tw = 0u, mm = 0u;
        oc = scan_trials(seed_lo, seed_hi, c_gsite[tj], sweep, nk >> 12, nk & 4095u, t0, EPV_COOP_WINDOW, misc & 1u,
                         (misc >> 1) & 1u, c_len[tj], c_r0[tj], c_r1[tj], c_trunc[tj], 0xffffffffu,
                         c_tj + (size_t)lane * 2u, 1u, 2u, 0.0, tw, njt, nielsen, &mm);
        c_tw[lane] = tw;
        c_res[lane] = (uint32_t)oc | (njt << 8);
        c_mm[lane] = mm;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const unsigned long long hit = __ballot(oc != TRIAL_FAIL);
      if (pend) {
        const unsigned long long mine = (hit >> (rank * G)) & (G == 64u ? ~0ull : ((1ull << G) - 1ull));
        const uint32_t upto = mine ? (uint32_t)(__ffsll((long long)mine) - 1) : G - 1u;   // helper lanes whose trials count
        for (uint32_t q = 0; q <= upto; ++q) { const uint32_t v = c_mm[rank * G + q]; maxm = v > maxm ? v : maxm; }
        if (mine) {
          const uint32_t w = rank * G + upto;
          cnt = c_res[w] >> 8;
          tstar = c_tw[w];
          if (cnt >= 1u) jt[0] = c_tj[(size_t)w * 2u] + t.start;
          if (cnt >= 2u) jt[1] = c_tj[(size_t)w * 2u + 1u] + t.start;
          pend = false;
        } else {
          tbase += G * EPV_COOP_WINDOW;
        }
      }
      todo = __ballot(pend);
      __builtin_amdgcn_wave_barrier();
    }
    if (i < n_seg) {
      EpvSegOut o;
      o.cnt = cnt; o.tstar = tstar; o.maxm = maxm; o.pad = 0u;
      o.j0 = jt[0]; o.j1 = jt[1];
      outs[i] = o;
    }
  }
}

__global__ __launch_bounds__(256, EPV_SEARCH_WAVES) void epv_seg_search_kernel(EpvDev S, uint32_t seed_lo,
                                                                               uint32_t seed_hi, uint32_t sweep,
                                                                               unsigned long long *counters) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  // per-wave cooperative-search area: task slots by rank and per-lane trial results
  __shared__ double c_len_[4][64], c_r0_[4][64], c_r1_[4][64], c_trunc_[4][64], c_tj_[4][64 * 2];
  __shared__ uint32_t c_misc_[4][64], c_gsite_[4][64], c_tbase_[4][64], c_nk_[4][64], c_res_[4][64],
      c_tw_[4][64], c_mm_[4][64];
  stage_constants(S, s_mem);
  const double *s_rates = s_mem;
  const uint32_t wave = threadIdx.x >> 6;
  EpvCoop W;
  W.len = c_len_[wave]; W.r0 = c_r0_[wave]; W.r1 = c_r1_[wave]; W.trunc = c_trunc_[wave]; W.tj = c_tj_[wave];
  W.misc = c_misc_[wave]; W.gsite = c_gsite_[wave]; W.tbase = c_tbase_[wave]; W.nk = c_nk_[wave];
  W.res = c_res_[wave]; W.tw = c_tw_[wave]; W.mm = c_mm_[wave];
  const bool nielsen = !(S.flags & EPV_FLAG_FORWARD_REJECTION);
  const uint32_t shard = blockIdx.y;
  const unsigned long long packed = counters[EPV_CNT_IDX(EPV_CNT_SEG, shard)];
  const uint64_t n_seg = (packed & 0xffffffffull) < S.seg_cap ? (uint64_t)(packed & 0xffffffffull) : S.seg_cap;
  const EpvSegTask *segs = S.segs + (uint64_t)shard * S.seg_cap;
  EpvSegOut *outs = S.segout + (uint64_t)shard * S.seg_cap;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  epv_seg_search_wave(S, s_rates, W, segs, outs, n_seg, (uint64_t)blockIdx.x * blockDim.x + wave * 64u, stride, seed_lo,
                      seed_hi, sweep, nielsen);
}

// A SHORT list (n_seg <= 32, one wave): G = 64 / n_seg lanes per segment, lane `sub` of a group
// evaluating trials 1 + sub, 1 + sub + G, ... -- one trial per lane and round instead of a lane
// walking its segment's trials alone while most of the wave idles (a 64-site wave of tree.nwk
// lists ~20 dirty segments).  The lowest non-failing trial wins and M is the most jumps of any
// trial up to it, whoever evaluated them, so the results are those of epv_seg_search_wave.
// Returns false when some segment is still open after `rounds` rounds (long tails belong to the
// wave-wide cooperative search): the caller then runs epv_seg_search_wave over the whole list.
__device__ __forceinline__ bool epv_seg_search_grouped(const EpvDev &S, const double *s_rates, const EpvSegTask *segs,
                                                       EpvSegOut *outs, uint32_t n_seg, uint32_t G, uint32_t rounds,
                                                       uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep, bool nielsen) {
  const int lane = epv_lane();
  const uint32_t seg = (uint32_t)lane / G, sub = (uint32_t)lane - seg * G, gbase = seg * G;
  EpvSegTask t;
  t.w0 = 0ull; t.len = -1.0; t.start = 0.0; t.w3 = 0ull;
  if (seg < n_seg) t = segs[seg];
  const uint32_t gsite = (uint32_t)(S.g0 + (t.w0 & 0xffffffffffull));
  const uint32_t node = (uint32_t)(t.w0 >> 40) & 4095u, k = (uint32_t)(t.w0 >> 52);
  const uint32_t prev = (uint32_t)t.w3 & 1u, sampled = (uint32_t)(t.w3 >> 1) & 1u, trip0 = (uint32_t)(t.w3 >> 2) & 7u;
  const double len = t.len, r0 = s_rates[trip0], r1 = s_rates[trip0 | 2u];
  bool pend = seg < n_seg && t.len >= 0.0;
  double trunc = 0.0;
  if (pend && sampled != prev) trunc = 1.0 - epv_exp(-(prev ? r1 : r0) * len);
  uint32_t tcur = 1u + sub, hist = 0u;      // hist: most jumps of this lane's failed trials so far
  for (uint32_t r = 0; r < rounds && __any(pend); ++r) {
    int oc = TRIAL_FAIL;
    uint32_t cnt = 0u, tw = 0u, mm = 0u;
    double jt[2] = {0.0, 0.0};
    if (pend)
      oc = scan_trials(seed_lo, seed_hi, gsite, sweep, node, k, tcur, 1u, prev, sampled, len, r0, r1, trunc, 0xffffffffu,
                       jt, 1u, 2u, t.start, tw, cnt, nielsen, &mm);
    const unsigned long long hit = __ballot(pend && oc != TRIAL_FAIL);
    const uint32_t gh = (uint32_t)(hit >> gbase) & ((1u << G) - 1u);     // this group's hits (G <= 8)
    const uint32_t wsub = gh ? (uint32_t)__ffs((int)gh) - 1u : G;         // the group's lowest hit
    // jumps that count towards M: everybody's earlier (failed) trials, this round's trials up to the winner's
    uint32_t m = (sub <= wsub && mm > hist) ? mm : hist;
    uint32_t m_all = 0u;
    for (uint32_t q = 0; q < G; ++q) {
      const uint32_t v = __shfl(m, (int)((gbase + q) & 63u));
      m_all = v > m_all ? v : m_all;
    }
    if (pend && gh) {
      if (sub == wsub) {
        EpvSegOut o;
        o.cnt = cnt; o.tstar = tcur; o.maxm = m_all; o.pad = 0u;
        o.j0 = jt[0]; o.j1 = jt[1];
        outs[seg] = o;
      }
      pend = false;
    } else {
      hist = m;
      tcur += G;
    }
  }
  return !__any(pend);
}

// one dirty branch (task word bt, its segments at segs/outs[first ..]): results in order into the
// proposal; a capacity overflow flags the site (phase index tid).  The task word carries the
// proposal's buffer and the branch's start state (bits 61, 60) so that the lane's chain does not
// begin with two dependent loads (sel, then the meta word the proposal kernel wrote)
__device__ __forceinline__ void epv_seg_assemble_one(const EpvDev &S, const double *s_rates, const EpvSegTask *segs,
                                                     const EpvSegOut *outs, unsigned long long bt, uint64_t first,
                                                     uint64_t s0, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep,
                                                     bool nielsen, epv_meta_t *prop_meta = nullptr, uint64_t site_lane0 = 0) {
  // prop_meta (fused phase): the wave's LDS column [branch][lane] of proposal meta words; the
  // assembled branch's word goes to its owner's entry, where the acceptance stage picks it up
  const uint64_t n = S.n;
  const uint32_t B = S.B, C = S.C;
  const uint64_t site = bt & 0xffffffffffull;
  const uint32_t b = (uint32_t)(bt >> 40) & 4095u, nds = (uint32_t)(bt >> 52) & 127u, end_state = (uint32_t)(bt >> 59) & 1u;
  const uint32_t start_state = (uint32_t)(bt >> 60) & 1u, selP = (uint32_t)(bt >> 61) & 1u;
  epv_meta_t *meta = S.meta + meta_idx(S, selP, b, site);
  double *dst = S.jumps + jump_idx(S, selP, b, site);
  uint32_t cnt = 0;
  bool ovf = false;
  for (uint32_t q = 0; q < nds && !ovf; ++q) {
    const EpvSegOut o = outs[first + q];
    const uint32_t room = C - cnt;
    if (o.maxm > room) { ovf = true; break; }     // some trial up to the winner needed more slots
    if (o.cnt <= 2u) {
      if (o.cnt >= 1u) dst[(uint64_t)cnt * n] = o.j0;
      if (o.cnt >= 2u) dst[(uint64_t)(cnt + 1u) * n] = o.j1;
    } else {                                      // replay the winning trial into the path
      const EpvSegTask t = segs[first + q];
      const uint32_t gsite = (uint32_t)(S.g0 + site);
      const uint32_t node = (uint32_t)(t.w0 >> 40) & 4095u, k = (uint32_t)(t.w0 >> 52);
      const uint32_t prev = (uint32_t)t.w3 & 1u, sampled = (uint32_t)(t.w3 >> 1) & 1u, trip0 = (uint32_t)(t.w3 >> 2) & 7u;
      const double r0 = s_rates[trip0], r1 = s_rates[trip0 | 2u];
      double trunc = 0.0;
      if (sampled != prev) trunc = 1.0 - epv_exp(-(prev ? r1 : r0) * t.len);
      const double u0 = first_draw(seed_lo, seed_hi, gsite, sweep, node, k, o.tstar);
      uint32_t nj2 = 0;
      run_trial(seed_lo, seed_hi, gsite, sweep, node, k, o.tstar, u0, prev, sampled, t.len, r0, r1, 0.0, 0.0, trunc,
                room, dst + (uint64_t)cnt * n, n, 0xffffffffu, t.start, nj2, nielsen);
    }
    cnt += o.cnt;
  }
  if (ovf) {
    cnt = (start_state ^ end_state) & 1u;         // keep the end-state parity; the proposal is rejected
    S.prop_flag[(site - s0) / 3u] = 1u;
  }
  *meta = (epv_meta_t)((start_state << EPV_INIT_SHIFT) | cnt);
  if (prop_meta) prop_meta[b * 64u + (uint32_t)((site - site_lane0) / 3u)] = (epv_meta_t)((start_state << EPV_INIT_SHIFT) | cnt);
}

// one lane per dirty branch: the results of its dirty segments, in order, into the proposal
__global__ __launch_bounds__(256) void epv_seg_assemble_kernel(EpvDev S, uint32_t seed_lo, uint32_t seed_hi,
                                                               uint32_t sweep, uint64_t s0,
                                                               unsigned long long *counters) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  stage_constants(S, s_mem);
  const double *s_rates = s_mem;
  const bool nielsen = !(S.flags & EPV_FLAG_FORWARD_REJECTION);
  const uint32_t shard = blockIdx.y;
  const unsigned long long packed = counters[EPV_CNT_IDX(EPV_CNT_SEG, shard)];
  const uint64_t n_b = (packed >> 32) < S.btask_cap ? (uint64_t)(packed >> 32) : S.btask_cap;
  const EpvSegTask *segs = S.segs + (uint64_t)shard * S.seg_cap;
  const EpvSegOut *outs = S.segout + (uint64_t)shard * S.seg_cap;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_b; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long bt = S.btasks[(uint64_t)shard * S.btask_cap + i];
    if (bt == ~0ull) continue;                      // blanked: the branch went to the sequential kernel
    epv_seg_assemble_one(S, s_rates, segs, outs, bt, S.bfirst[(uint64_t)shard * S.btask_cap + i], s0, seed_lo, seed_hi,
                         sweep, nielsen);
  }
}

#endif
