#ifndef EPV_PROPOSE3_H
#define EPV_PROPOSE3_H
// epv_propose3.h -- the proposal kernel of a colour phase for LARGE trees (included by
// epv_kernels.h behind epv_propose2.h).  Same contract as epv_mh_propose2_kernel<true, false, false>
// -- pruning (SingleSiteSampler.cpp:116-157) and downward sampling of the segment end states
// (:180-255) of one site per lane, bit for bit the numbers of the oracle's parallel rung; dirty
// (site, branch) pairs onto the bucketed lists of epv_mh_jumps_kernel, sites whose proposal differs
// from their path onto the accept list -- built around what the counters said about the 16-leaf
// tree (profiles/r03f_pmc_*_bal16.csv): the first kernel there is bound by memory latency and
// fabric bandwidth, not by issue.  Its waves walk 30 branches with two or three dependent global
// round trips each (meta words, jump planes of whichever lane is heavy at that node, the records of
// the children), and the Felsenstein records of 64 sites x 30 branches (2 KB per site) do not fit
// LDS, so they stream through a slab of global memory: 2 GB written and read back per phase.
//
// Here a wave keeps in LDS only a 16-bit word per (node, lane) -- neighbour start states, leaf
// state, "has a neighbour jump", min(K - 1, 255), later the proposal's end state -- and a stack of
// partial products, one level per tree depth.  What goes to global memory per wave:
//   * q of the INTERNAL nodes (14 of 30 on the 16-leaf tree), one coalesced 1 KB row each, written
//     once by pruning and read once by the downward pass.  p.front of a branch is never stored: its
//     parent multiplies it into the stack level of its depth the moment it exists (a node with two
//     children: q = p_first * p_second either way round, the product of two doubles commutes), and
//     the downward pass recomputes it from q and the matrix table (one segment) or finds it in the
//     heavy record (several segments);
//   * the records of the heavy segments (branches with a neighbour jump: ~10 % of the pairs), 64 B
//     each, listed by one lane per (site, branch) pair and evaluated one SEGMENT per lane as in
//     epv_mh_propose2_kernel -- all merges of a wave in one chain of round trips instead of one
//     chain per node.  A record's matrix entries give way to the partials p[k] once pruning has used them.
// No node table, no record pool: ~8 KB of LDS per wave instead of 19.5 (second kernel) and about a
// fifth of the first kernel's record traffic.
//
// Preconditions (plan_p3, epv_abi.hip): N <= 64 (node masks are one word), every node but the
// root has at most two children, the stack fits its LDS budget.

#define EPV_P3_PCAP 512u   /* heavy (lane, node) pairs a wave lists per round (at least 64: one lane's worst case) */

// s_node[node]: parent | (depth - 1) << 6 | q row << 12 | leaf << 18 | last child of its parent << 19
#define EPV_P3_PARENT(w) ((w) & 63u)
#define EPV_P3_LEVEL(w) (((w) >> 6) & 63u)
#define EPV_P3_QROW(w) (((w) >> 12) & 63u)
#define EPV_P3_LEAF(w) (((w) >> 18) & 1u)
#define EPV_P3_LASTCHILD(w) (((w) >> 19) & 1u)

// s_ent[node * 64 + lane]: bit 0 right neighbour's start state, 1 left neighbour's, 2 this path's,
// 3 parity of this path's jump count, 4 this path has jumps, 5 heavy (a neighbour jumps on the
// branch), 6 the proposal's end state (downward pass), 7 more than 64 segments (the sequential
// loop of the downward pass), bits 8..11 min(K - 1, 15), bits 12..15 of a heavy branch: end state
// and "clean" for start state 0, the same for start state 1 (the pair pass behind pruning)
#define EPV_P3_HEAVY 32u
#define EPV_P3_SLOW 128u

template <bool DUMMY>
__global__ __launch_bounds__(256, 2) void epv_mh_propose3_kernel(
    EpvDev S, uint32_t colour, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep, uint64_t first,
    uint64_t last, uint64_t own_first, uint64_t own_last, uint32_t list_cap, uint32_t n_qrows, uint32_t levels,
    uint32_t parity, unsigned long long *counters, double *gpool, const double *segtab, const uint32_t *nodetab) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  constexpr uint32_t HREC = EPV_HREC_SHORT, LEN_AT = HREC - 2u, INFO_AT = HREC - 1u;
#ifdef EPV_P2_PROFILE
  unsigned long long t_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif
  const uint32_t const_dbl = (20u + S.N + 1u) & ~1u;
  const uint32_t tab_dbl = S.B * 4u * EPV_SEGTAB_DBL;
  const uint32_t node_dbl = (S.N + 1u) / 2u;
  const uint32_t ent_dbl = (S.N * 64u * (uint32_t)sizeof(uint16_t) + 15u) / 16u * 2u;
  const uint32_t stk_dbl = levels * 128u;     // the stack of partial products (pruning)
  const uint32_t wave_id = threadIdx.x >> 6;
  const uint32_t wave_dbl = ent_dbl + stk_dbl;
  double *s_const = s_mem;
  double *s_tab = s_mem + const_dbl;
  uint32_t *s_node = reinterpret_cast<uint32_t *>(s_mem + const_dbl + tab_dbl);
  double *s_wave = s_mem + const_dbl + tab_dbl + node_dbl + (size_t)wave_id * wave_dbl;
  uint16_t *s_ent = reinterpret_cast<uint16_t *>(s_wave);
  double *s_stk = s_wave + ent_dbl;
  const int lane = epv_lane();
  const uint32_t my_shard = (blockIdx.x * (blockDim.x >> 6) + wave_id) & (EPV_SHARDS - 1u);
  // per-wave slab: the pair list, q rows of 64 interleaved records, the flat heavy list
  double *slab = gpool + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave_id) *
                             ((size_t)EPV_P3_PCAP + (size_t)n_qrows * 128u + (size_t)list_cap * HREC);
  unsigned long long *plist = reinterpret_cast<unsigned long long *>(slab);
  double *qrows = slab + EPV_P3_PCAP;
  double *list = qrows + (size_t)n_qrows * 128u;
  const uint64_t gfirst = S.g0 + first;
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)(gfirst % 3u)) % 3u);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site = s0 + 3u * tid;
  const bool valid = site <= last;
  const uint64_t n = S.n;
  const uint32_t B = S.B;
  const uint32_t gsite = (uint32_t)(S.g0 + site);
  const uint32_t gsite_lane0 = gsite - 3u * (uint32_t)lane;

  uint32_t selL = 0, selM = 0, selR = 0;
  if (valid) { selL = S.sel[site - 1]; selM = S.sel[site]; selR = S.sel[site + 1]; }
  for (uint32_t i = threadIdx.x; i < tab_dbl; i += blockDim.x) s_tab[i] = segtab[i];
  for (uint32_t i = threadIdx.x; i < S.N; i += blockDim.x) s_node[i] = nodetab[i];
  stage_constants(S, s_const);
  const double *s_rates = s_const;
  const double *s_blen = s_const + 20;

  const uint64_t Bn = (uint64_t)B * n, Cn = (uint64_t)S.C * n;
  const uint64_t mbaseL = (selL ? Bn : 0ull) + (site - 1), mbaseR = (selR ? Bn : 0ull) + (site + 1);
  const uint64_t mbaseM = (selM ? Bn : 0ull) + site;
  const uint64_t jbaseL = (selL ? Bn * S.C : 0ull) + (site - 1), jbaseR = (selR ? Bn * S.C : 0ull) + (site + 1);
  // ---- 0. the meta words of the three columns, in batches of independent loads, condensed to one
  //         16-bit word per (node, lane)
  uint32_t heavy = 0, n_pairs = 0;
  if (valid) {
#pragma unroll 6
    for (uint32_t b = 0; b < B; ++b) {
      const uint32_t mL = S.meta[mbaseL + (uint64_t)b * n];
      const uint32_t mR = S.meta[mbaseR + (uint64_t)b * n];
      const uint32_t mM = S.meta[mbaseM + (uint64_t)b * n];
      const uint32_t K = (mL & EPV_NJ_MASK) + (mR & EPV_NJ_MASK) + 1u;
      const uint32_t e = (mR >> EPV_INIT_SHIFT) | ((mL >> EPV_INIT_SHIFT) << 1) | ((mM >> EPV_INIT_SHIFT) << 2) |
                         ((mM & 1u) << 3) | ((mM & EPV_NJ_MASK) ? 16u : 0u) | (K >= 2u ? EPV_P3_HEAVY : 0u) |
                         (K > 64u ? EPV_P3_SLOW : 0u) | ((K - 1u < 15u ? K - 1u : 15u) << 8);
      s_ent[(b + 1u) * 64u + lane] = (uint16_t)e;
      if (K >= 2u) { heavy += K; ++n_pairs; }
    }
  }
  // segments of the branch above `node` (the word holds it up to 15; beyond, from the meta words)
  auto segments_of = [&](uint32_t e, uint32_t node) __attribute__((always_inline)) -> uint32_t {
    const uint32_t k4 = (e >> 8) & 15u;
    if (k4 < 15u) return k4 + 1u;
    return (uint32_t)(S.meta[mbaseL + (uint64_t)(node - 1u) * n] & EPV_NJ_MASK) +
           (uint32_t)(S.meta[mbaseR + (uint64_t)(node - 1u) * n] & EPV_NJ_MASK) + 1u;
  };

  P2_MARK(0);
  bool pending = valid;
  while (__any(pending)) {
    const uint32_t wantH = pending ? heavy : 0u, wantP = pending ? n_pairs : 0u;
    const uint32_t inclH = wave_incl_scan_u32(wantH), inclP = wave_incl_scan_u32(wantP);
    // both sums are non-decreasing in the lane index: the lanes that run are a prefix of the pending ones
    const bool run = pending && inclH <= list_cap && inclP <= EPV_P3_PCAP;
    const unsigned long long rmask = __ballot(run);
    const int hi_lane = rmask ? 63 - __clzll((long long)rmask) : 0;
    const uint32_t totH = rmask ? epv_bcast(inclH, hi_lane) : 0u, totP = rmask ? epv_bcast(inclP, hi_lane) : 0u;
    const uint32_t hbase = inclH - wantH;

    P2_MARK(1);
    // ---- 1. the heavy (lane, node) pairs: lane | node << 6 | segments << 12 | first record << 32
    if (run && n_pairs) {
      uint32_t hcur = hbase, at = inclP - n_pairs;
      for (uint32_t node = 1u; node < S.N; ++node) {
        const uint32_t e = s_ent[node * 64u + lane];
        if (!(e & EPV_P3_HEAVY)) continue;
        const uint32_t K = segments_of(e, node);
        plist[at++] = (unsigned long long)lane | ((unsigned long long)node << 6) | ((unsigned long long)K << 12) |
                      ((unsigned long long)hcur << 32);
        hcur += K;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 2. one pair per lane: forward merge of the neighbours' jumps (Segment.cpp:35-79) into the
    //         records' length and address fields.  Three passes of pairs at a time, stage by stage, so
    //         that their round trips (pair word -> meta words -> first jumps) overlap
    for (uint32_t p0 = 0; p0 < totP; p0 += 3u * 64u) {
      constexpr int NP = 3;
      unsigned long long pr[NP];
      bool act[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const uint32_t pidx = p0 + 64u * (uint32_t)q + (uint32_t)lane;
        act[q] = pidx < totP;
        pr[q] = act[q] ? plist[pidx] : 0ull;
      }
      const double *Lj[NP], *Rj[NP];
      uint32_t cL[NP], cR[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const uint32_t owner = (uint32_t)pr[q] & 63u, b = ((((uint32_t)pr[q] >> 6) & 63u) - 1u) & 63u;
        const uint64_t jl = (uint64_t)__shfl((uint32_t)jbaseL, (int)owner) | ((uint64_t)__shfl((uint32_t)(jbaseL >> 32), (int)owner) << 32);
        const uint64_t jr = (uint64_t)__shfl((uint32_t)jbaseR, (int)owner) | ((uint64_t)__shfl((uint32_t)(jbaseR >> 32), (int)owner) << 32);
        const uint64_t ml = (uint64_t)__shfl((uint32_t)mbaseL, (int)owner) | ((uint64_t)__shfl((uint32_t)(mbaseL >> 32), (int)owner) << 32);
        const uint64_t mr = (uint64_t)__shfl((uint32_t)mbaseR, (int)owner) | ((uint64_t)__shfl((uint32_t)(mbaseR >> 32), (int)owner) << 32);
        cL[q] = act[q] ? (uint32_t)S.meta[ml + (uint64_t)b * n] : 0u;
        cR[q] = act[q] ? (uint32_t)S.meta[mr + (uint64_t)b * n] : 0u;
        Lj[q] = S.jumps + jl + (uint64_t)b * Cn;
        Rj[q] = S.jumps + jr + (uint64_t)b * Cn;
      }
      double tl0[NP], tr0[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        tl0[q] = (cL[q] & EPV_NJ_MASK) ? Lj[q][0] : EPV_INF;
        tr0[q] = (cR[q] & EPV_NJ_MASK) ? Rj[q][0] : EPV_INF;
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        if (!act[q]) continue;
        const uint32_t owner = (uint32_t)pr[q] & 63u, node = ((uint32_t)pr[q] >> 6) & 63u, hcur = (uint32_t)(pr[q] >> 32);
        const uint32_t nL = cL[q] & EPV_NJ_MASK, nR = cR[q] & EPV_NJ_MASK, K = nL + nR + 1u;
        uint32_t trip0 = 4u * (cL[q] >> EPV_INIT_SHIFT) + (cR[q] >> EPV_INIT_SHIFT), i = 0, j = 0;
        double seg_start = 0.0;
        double tl = tl0[q], tr = tr0[q];
        for (uint32_t k = 0; k < K; ++k) {
          const bool last_seg = (k + 1u == K);
          const bool take_left = tl < tr;
          const double seg_end = last_seg ? s_blen[node] : (take_left ? tl : tr);
          double *rec = list + (size_t)(hcur + k) * HREC;
          rec[LEN_AT] = seg_end - seg_start;
          rec[INFO_AT] = epv_u2d((uint64_t)trip0 | ((uint64_t)owner << 3) | ((uint64_t)node << 9) | ((uint64_t)k << 21));
          if (!last_seg) {
            if (take_left) { trip0 ^= 4u; ++i; tl = i < nL ? Lj[q][(uint64_t)i * n] : EPV_INF; }
            else { trip0 ^= 1u; ++j; tr = j < nR ? Rj[q][(uint64_t)j * n] : EPV_INF; }
            seg_start = seg_end;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    P2_MARK(2);
    // ---- 3. evaluate them densely, one segment per lane: matrices, no-jump bounds, the segment's
    //         Philox block (none of which depends on the recursion state)
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
    // ---- 4. pruning, reverse pre-order (SingleSiteSampler.cpp:116-157).  The matrix entries of a heavy
    //         branch are fetched four segments at a time, at the top of the node's iteration: one
    //         round trip per branch (K <= 4) instead of one per segment
    if (run) {
      uint32_t hcur = hbase + heavy;      // one past the next record (records are consumed last to first)
      for (uint32_t node = S.N - 1u; node >= 1u; --node) {
        const uint32_t nw = s_node[node];
        const uint32_t e = s_ent[node * 64u + lane];
        uint32_t K = 0u, m = 0u;
        double m00 = 0.0, m01 = 0.0, m10 = 0.0, m11 = 0.0, m20 = 0.0, m21 = 0.0, m30 = 0.0, m31 = 0.0;
        if (e & EPV_P3_HEAVY) {
          K = segments_of(e, node);
          m = K < 4u ? K : 4u;
          const double *r = list + (size_t)(hcur - 1u) * HREC;
          m00 = r[0]; m01 = r[1];
          m10 = r[-(int)HREC]; m11 = r[1 - (int)HREC];      // (K >= 2: the record exists)
          if (m > 2u) { m20 = r[-2 * (int)HREC]; m21 = r[1 - 2 * (int)HREC]; }
          if (m > 3u) { m30 = r[-3 * (int)HREC]; m31 = r[1 - 3 * (int)HREC]; }
        }
        double n0, n1;
        if (EPV_P3_LEAF(nw)) {
          const uint32_t leaf_state = ((e >> 2) ^ (e >> 3)) & 1u;
          n0 = leaf_state ? 0.0 : 1.0;
          n1 = leaf_state ? 1.0 : 0.0;
        } else {
          const double *a = s_stk + (size_t)EPV_P3_LEVEL(nw) * 128u + (size_t)lane * 2u;
          n0 = a[0]; n1 = a[1];
          double *qr = qrows + (size_t)EPV_P3_QROW(nw) * 128u + (size_t)lane * 2u;
          qr[0] = n0; qr[1] = n1;
        }
        if (!(e & EPV_P3_HEAVY)) {
          const double *t = s_tab + (size_t)((node - 1u) * 4u + (e & 3u)) * EPV_SEGTAB_DBL;
          const double P00 = t[0], P11 = t[1];
          const double P01 = 1.0 - P00, P10 = 1.0 - P11;
          const double a = P00 * n0 + P01 * n1;
          const double c = P10 * n0 + P11 * n1;
          n0 = a; n1 = c;
        } else {
          uint32_t left = K;
          for (;;) {
            double *hr = list + (size_t)(hcur - 1u) * HREC;
            // p[kk] = M[kk] p[kk + 1]; it takes the place of the matrix in the record
#define EPV_P3_STEP(P00_, P11_, BACK)                                        \
            {                                                                 \
              const double P01 = 1.0 - (P00_), P10 = 1.0 - (P11_);            \
              const double a = (P00_) * n0 + P01 * n1;                         \
              const double c = P10 * n0 + (P11_) * n1;                         \
              hr[-(int)((BACK) * HREC)] = a; hr[1 - (int)((BACK) * HREC)] = c; \
              n0 = a; n1 = c;                                                  \
            }
            EPV_P3_STEP(m00, m01, 0u)
            if (m > 1u) EPV_P3_STEP(m10, m11, 1u)
            if (m > 2u) EPV_P3_STEP(m20, m21, 2u)
            if (m > 3u) EPV_P3_STEP(m30, m31, 3u)
#undef EPV_P3_STEP
            hcur -= m;
            left -= m;
            if (left == 0u) break;
            m = left < 4u ? left : 4u;
            const double *r = list + (size_t)(hcur - 1u) * HREC;
            m00 = r[0]; m01 = r[1];
            if (m > 1u) { m10 = r[-(int)HREC]; m11 = r[1 - (int)HREC]; }
            if (m > 2u) { m20 = r[-2 * (int)HREC]; m21 = r[1 - 2 * (int)HREC]; }
            if (m > 3u) { m30 = r[-3 * (int)HREC]; m31 = r[1 - 3 * (int)HREC]; }
          }
        }
        // p.front of this branch into its parent's product (a child of the root has no use for it:
        // the root state is kept)
        const uint32_t par = EPV_P3_PARENT(nw);
        if (par != 0u) {
          double *acc = s_stk + (size_t)EPV_P3_LEVEL(s_node[par]) * 128u + (size_t)lane * 2u;
          if (EPV_P3_LASTCHILD(nw)) { acc[0] = n0; acc[1] = n1; }
          else { acc[0] = acc[0] * n0; acc[1] = acc[1] * n1; }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    P2_MARK(4);
    // ---- 5. the heavy branches' end states for BOTH start states, one pair per lane.  The uniforms
    //         of a segment are fixed, so the chain of end states along a branch is a function of the
    //         start state alone: evaluating it for 0 and for 1 here, densely, leaves the sequential
    //         walk over the tree below with a table lookup where the first kernel iterated
    //         max_lanes(K) times over a division with a tenth of the lanes busy.  The words of
    //         sampled states for both start states wait in the first record's PT slots, which this
    //         pass has used by then.  (More than 64 segments: the walk's own loop, EPV_P3_SLOW.)
    for (uint32_t p0 = 0; p0 < totP; p0 += 64u) {
      const uint32_t pidx = p0 + (uint32_t)lane;
      if (pidx < totP) {
        const unsigned long long pr = plist[pidx];
        const uint32_t owner = (uint32_t)pr & 63u, node = ((uint32_t)pr >> 6) & 63u, K = ((uint32_t)pr >> 12) & 0xfffffu;
        const uint32_t hrec0 = (uint32_t)(pr >> 32);
        if (K <= 64u) {
          const uint32_t nw = s_node[node];
          const uint32_t e = s_ent[node * 64u + owner];
          double q0, q1;
          if (EPV_P3_LEAF(nw)) {
            const uint32_t leaf_state = ((e >> 2) ^ (e >> 3)) & 1u;
            q0 = leaf_state ? 0.0 : 1.0;
            q1 = leaf_state ? 1.0 : 0.0;
          } else {
            const double *qr = qrows + (size_t)EPV_P3_QROW(nw) * 128u + (size_t)owner * 2u;
            q0 = qr[0]; q1 = qr[1];
          }
          double *hr = list + (size_t)hrec0 * HREC;
          uint32_t prevA = 0u, prevB = 1u;            // chains from start state 0 and from start state 1
          bool cleanA = true, cleanB = true;
          unsigned long long wA = 0ull, wB = 0ull;
          double pk0 = hr[0], pk1 = hr[1];
          for (uint32_t k = 0; k < K; ++k) {
            const bool last_seg = (k + 1u == K);
            const double nxt0 = last_seg ? q0 : hr[HREC], nxt1 = last_seg ? q1 : hr[HREC + 1u];
            const double PT00 = hr[2], PT10 = hr[3], nb0 = hr[4], nb1 = hr[5], u_end = hr[6], u_first = hr[7];
            const double p0A = (prevA ? PT10 : PT00) * nxt0 / (prevA ? pk1 : pk0);
            const double p0B = (prevB ? PT10 : PT00) * nxt0 / (prevB ? pk1 : pk0);
            const uint32_t sA = (u_end > p0A) ? 1u : 0u, sB = (u_end > p0B) ? 1u : 0u;
            cleanA = cleanA && (sA == prevA) && (1.0 - u_first < (prevA ? nb1 : nb0));
            cleanB = cleanB && (sB == prevB) && (1.0 - u_first < (prevB ? nb1 : nb0));
            wA |= (unsigned long long)sA << k;
            wB |= (unsigned long long)sB << k;
            prevA = sA; prevB = sB;
            pk0 = nxt0; pk1 = nxt1;
            hr += HREC;
          }
          double *h0 = list + (size_t)hrec0 * HREC;
          h0[2] = epv_u2d(wA);
          h0[3] = epv_u2d(wB);
          s_ent[node * 64u + owner] = (uint16_t)((e & 0x0fffu) | (prevA << 12) | ((cleanA ? 1u : 0u) << 13) |
                                                 (prevB << 14) | ((cleanB ? 1u : 0u) << 15));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    P2_MARK(5);
    // ---- 6. downward sampling of the segment END STATES (:180-255); the jump times are drawn by
    //         epv_mh_jumps_kernel for the dirty branches only.  q of the next internal node is
    //         requested one node ahead.
    unsigned long long dirty = 0ull, multi = 0ull, deep = 0ull;
    bool ident = true;
    const uint32_t root_state = run ? ((uint32_t)s_ent[64u + lane] >> 2) & 1u : 0u;
    if (run) {
      uint32_t hcur = hbase;
      double qn0 = 0.0, qn1 = 0.0;
      {
        const uint32_t nw1 = s_node[1];
        if (!EPV_P3_LEAF(nw1)) { const double *qr = qrows + (size_t)EPV_P3_QROW(nw1) * 128u + (size_t)lane * 2u; qn0 = qr[0]; qn1 = qr[1]; }
      }
      for (uint32_t node = 1u; node < S.N; ++node) {
        const uint32_t b = node - 1u;
        const uint32_t nw = s_node[node];
        const uint32_t e = s_ent[node * 64u + lane];
        const uint32_t par = EPV_P3_PARENT(nw);
        const uint32_t start_state = (par == 0u) ? root_state : ((uint32_t)s_ent[par * 64u + lane] >> 6) & 1u;
        double q0 = qn0, q1 = qn1;
        if (EPV_P3_LEAF(nw)) {
          const uint32_t leaf_state = ((e >> 2) ^ (e >> 3)) & 1u;
          q0 = leaf_state ? 0.0 : 1.0;
          q1 = leaf_state ? 1.0 : 0.0;
        }
        if (node + 1u < S.N) {
          const uint32_t nwn = s_node[node + 1u];
          if (!EPV_P3_LEAF(nwn)) { const double *qr = qrows + (size_t)EPV_P3_QROW(nwn) * 128u + (size_t)lane * 2u; qn0 = qr[0]; qn1 = qr[1]; }
        }
        uint32_t prev = start_state, K = 1u;
        bool clean;
        if (!(e & EPV_P3_HEAVY)) {
          const double *t = s_tab + (size_t)(b * 4u + (e & 3u)) * EPV_SEGTAB_DBL;
          const double P00 = t[0], P11 = t[1];
          const double P01 = 1.0 - P00, P10 = 1.0 - P11;
          const double pk0 = P00 * q0 + P01 * q1;     // p.front, as pruning computed it
          const double pk1 = P10 * q0 + P11 * q1;
          const double PT0 = prev ? t[3] : t[2];
          const double nb = prev ? t[5] : t[4];
          const epv_block2 blk = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, 0u, 0u, 0u);
          const double p0 = PT0 * q0 / (prev ? pk1 : pk0);
          const uint32_t sampled = (blk.d0 > p0) ? 1u : 0u;
          clean = (sampled == prev) && (1.0 - blk.d1 < nb);
          if (!clean) S.prop_states[((uint64_t)b * S.phase_cap + tid) * S.W] = (unsigned long long)sampled;
          prev = sampled;
        } else if (!(e & EPV_P3_SLOW)) {
          K = segments_of(e, node);
          hcur += K;
          clean = (e >> (start_state ? 15u : 13u)) & 1u;
          prev = (e >> (start_state ? 14u : 12u)) & 1u;
        } else {
          K = segments_of(e, node);
          clean = true;
          unsigned long long word = 0ull;
          uint64_t *states = S.prop_states + ((uint64_t)b * S.phase_cap + tid) * S.W;
          const double *hr = list + (size_t)hcur * HREC;
          double pk0 = hr[0], pk1 = hr[1];
          for (uint32_t k = 0; k < K; ++k) {
            const bool last_seg = (k + 1u == K);
            const double nxt0 = last_seg ? q0 : hr[HREC], nxt1 = last_seg ? q1 : hr[HREC + 1u];
            const double PT0 = prev ? hr[3] : hr[2];
            const double nb = prev ? hr[5] : hr[4];
            const double p0 = PT0 * nxt0 / (prev ? pk1 : pk0);
            const uint32_t sampled = (hr[6] > p0) ? 1u : 0u;
            clean = clean && (sampled == prev) && (1.0 - hr[7] < nb);
            word |= (unsigned long long)sampled << (k & 63u);
            if ((k & 63u) == 63u) { states[k >> 6] = word; word = 0ull; }
            prev = sampled;
            pk0 = nxt0; pk1 = nxt1;
            hr += HREC;
          }
          hcur += K;
          if ((K & 63u) && !clean) states[(K - 1u) >> 6] = word;   // only a dirty branch is read back
        }
        // proposal end state for the children (bit 6), "dirty" for the pair pass below (bit 7 of a heavy branch)
        s_ent[node * 64u + lane] = (uint16_t)((e & ~EPV_P3_SLOW) | (prev << 6) |
                                              ((e & EPV_P3_HEAVY) && !(e & EPV_P3_SLOW) && !clean ? EPV_P3_SLOW : 0u));
        // same as the current path?  (no jumps on either, same start state)
        ident = ident && clean && !(e & 16u) && ((e >> 2) & 1u) == start_state;
        if (!clean) {
          dirty |= 1ull << b;
          if (K == 2u || K >= 4u) multi |= 1ull << b;   // four buckets by segment count
          if (K >= 3u) deep |= 1ull << b;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 7. the words of sampled states of the dirty heavy branches, one pair per lane again
    for (uint32_t p0 = 0; p0 < totP; p0 += 64u) {
      const uint32_t pidx = p0 + (uint32_t)lane;
      if (pidx < totP) {
        const unsigned long long pr = plist[pidx];
        const uint32_t owner = (uint32_t)pr & 63u, node = ((uint32_t)pr >> 6) & 63u;
        const uint32_t e = s_ent[node * 64u + owner];
        if (e & EPV_P3_SLOW) {           // (bit 7 now: heavy, at most 64 segments, dirty)
          const uint32_t par = EPV_P3_PARENT(s_node[node]);
          const uint32_t st = (par == 0u) ? ((uint32_t)s_ent[64u + owner] >> 2) & 1u : ((uint32_t)s_ent[par * 64u + owner] >> 6) & 1u;
          const double *h0 = list + (size_t)(uint32_t)(pr >> 32) * HREC;
          S.prop_states[((uint64_t)(node - 1u) * S.phase_cap + (tid - (uint32_t)lane + owner)) * S.W] = epv_d2u(h0[2u + st]);
        }
      }
    }
    P2_MARK(6);
    epv_flush_tasks(S, counters, dirty, multi, deep, B - 1u, site, lane, my_shard);

    // ---- 8. hand-over.  A proposal equal to the current path is accepted with probability one and
    //         changes neither the paths nor the cached likelihoods: count it and be done.  Everything
    //         else: start states of the proposal's branches into the other buffer, and the site onto
    //         the accept list of this wave's shard.
    const bool to_list = run && !ident;
    if (to_list) {
      for (uint32_t node = 1u; node < S.N; ++node) {
        const uint32_t par = EPV_P3_PARENT(s_node[node]);
        const uint32_t st = (par == 0u) ? root_state : ((uint32_t)s_ent[par * 64u + lane] >> 6) & 1u;
        S.meta[(selM ? 0ull : Bn) + (uint64_t)(node - 1u) * n + site] = (epv_meta_t)(st << EPV_INIT_SHIFT);
      }
      S.prop_flag[tid] = 0u;
    }
    {
      const unsigned long long lm = __ballot(to_list);
      if (lm) {
        unsigned long long base = 0ull;
        if (lane == 0)
          base = atomicAdd(&counters[EPV_CNT_IDX(parity ? EPV_CNT_ALIST1 : EPV_CNT_ALIST0, my_shard)],
                           (unsigned long long)__popcll(lm));
        const uint32_t b0 = epv_bcast((uint32_t)base, 0);
        if (to_list)
          S.alist[(uint64_t)my_shard * S.alist_cap + b0 + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))] = (uint32_t)tid;
      }
      const unsigned long long am = __ballot(run && ident && site >= own_first && site <= own_last);
      if (am && lane == 0)
        atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_ACCEPT, my_shard)], (unsigned long long)__popcll(am));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();     // (a later round reuses the pair list and the records)
    pending = pending && !run;
    P2_MARK(7);
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
