#ifndef EPV_PROPOSE3_H
#define EPV_PROPOSE3_H
// epv_propose3.h -- the proposal kernel of a colour phase for LARGE trees (included by
// epv_kernels.h behind epv_propose2.h).  Same contract as epv_mh_propose2_kernel<true, false, false>
// -- pruning (SingleSiteSampler.cpp:116-157) and downward sampling of the segment end states
// (:180-255) of one site per lane, bit for bit the numbers of the oracle's parallel rung; dirty
// (site, branch) pairs onto the bucketed lists of epv_mh_jumps_kernel, sites whose proposal differs
// from their path onto the accept list -- built around what the counters said about the 16-leaf
// tree (profiles/r03f_pmc_*_bal16.csv): the first kernel there is bound by memory latency and
// fabric bandwidth, not by issue.  Its waves walk 30 branches, each step with two or three
// dependent global round trips (meta words, jump planes of whichever lane is heavy at that node,
// the records of the children), and the Felsenstein records of 64 sites x 30 branches (2 KB per
// site) do not fit LDS, so they stream through a slab of global memory: 2 GB written and read back
// per phase.
//
// Here nothing about a (node, lane) pair lives in memory at all unless the branch is heavy:
//   * what the recursions need to know about a branch -- the neighbours' start states, this path's
//     start state / jump parity / "has jumps", "a neighbour jumps on it" (heavy) -- are six 64-bit
//     masks per lane, one bit per node (one or two words: N <= 128), filled from one batch of meta-word loads;
//   * the tree is walked LEVEL BY LEVEL, not node by node: the nodes of a level are independent, so
//     their global loads are issued together and a wave waits once per level (4 on the 16-leaf tree)
//     instead of once or twice per node (30);
//   * q of an internal node = the product of its children's p.front, written once as a coalesced
//     1 KB row per wave (14 rows on the 16-leaf tree) and read back by the level above and by the
//     downward pass; p.front itself is recomputed from q and the matrix table where needed;
//   * heavy branches (~10 % of the pairs) get a lane each: the pairs are listed, their segments
//     merged (Segment.cpp:35-79) and evaluated densely as in epv_mh_propose2_kernel; then, level by
//     level behind the nodes' pass, a pair lane runs the branch's pruning chain from registers and --
//     the uniforms of a segment being fixed -- the chain of end states for BOTH start states, leaving
//     p.front (16 B) and six result bits in LDS for the lanes that own the sites.  The sequential
//     parts never iterate over segments.
// LDS per wave: 6.9 KB (pair list and results) instead of 19.5 (second kernel); global traffic per
// wave: the 14 q rows and the heavy records.
//
// Preconditions (plan_p3, epv_abi.hip): N <= 128 (node masks of one or two 64-bit words: the kernel is a
// template on the word count), every node but the root has at most two children.

#define EPV_P3_PCAP 256u   /* heavy (lane, node) pairs a wave lists per round (at least 64: one lane's worst case) */

// s_tree[node]: parent | first child << 7 | second child << 14 | q row << 21 | leaf << 27; s_dep[node]: depth
#define EPV_P3_PARENT(w) ((w) & 127u)
#define EPV_P3_CHILD1(w) (((w) >> 7) & 127u)
#define EPV_P3_CHILD2(w) (((w) >> 14) & 127u)
#define EPV_P3_QROW(w) (((w) >> 21) & 63u)
#define EPV_P3_LEAF(w) (((w) >> 27) & 1u)
// pair word: lane | node << 6 | segments << 13 | first record << 32 | leaf state << 62
#define EPV_P3_PAIR_NODE(pr) (((uint32_t)(pr) >> 6) & 127u)
#define EPV_P3_PAIR_K(pr) (((uint32_t)(pr) >> 13) & 0x7ffffu)
// s_pb[pair]: end state for start state 0 | clean << 1 | end state for start 1 << 2 | clean << 3 |
//             (K == 2 or K >= 4) << 4 | (K >= 3) << 5   (the task buckets of epv_flush_tasks)

// One bit per node in W 64-bit words (W = 1: up to 64 nodes, W = 2: up to 128).  Node indices are
// wave-uniform wherever these are called with one, so the word select is scalar work.
template <int W> struct P3Mask { unsigned long long w[W]; };
template <int W> __device__ __forceinline__ void p3_zero(P3Mask<W> &m) {
#pragma unroll
  for (int q = 0; q < W; ++q) m.w[q] = 0ull;
}
template <int W> __device__ __forceinline__ uint32_t p3_get(const P3Mask<W> &m, uint32_t i) {
  if constexpr (W == 1) return (uint32_t)(m.w[0] >> i) & 1u;
  else {
    unsigned long long x = m.w[0];
#pragma unroll
    for (int q = 1; q < W; ++q) x = (i >> 6) == (uint32_t)q ? m.w[q] : x;
    return (uint32_t)(x >> (i & 63u)) & 1u;
  }
}
template <int W> __device__ __forceinline__ void p3_or(P3Mask<W> &m, uint32_t i, uint32_t bit) {
  if constexpr (W == 1) m.w[0] |= (unsigned long long)bit << i;
  else {
#pragma unroll
    for (int q = 0; q < W; ++q) m.w[q] |= (i >> 6) == (uint32_t)q ? (unsigned long long)bit << (i & 63u) : 0ull;
  }
}
// popcount of a & g, and of a & g restricted to the bits below i
template <int W> __device__ __forceinline__ uint32_t p3_count(const P3Mask<W> &a, const P3Mask<W> &g) {
  uint32_t c = 0;
#pragma unroll
  for (int q = 0; q < W; ++q) c += (uint32_t)__popcll(a.w[q] & g.w[q]);
  return c;
}
template <int W> __device__ __forceinline__ uint32_t p3_count_below(const P3Mask<W> &a, const P3Mask<W> &g, uint32_t i) {
  if constexpr (W == 1) return (uint32_t)__popcll(a.w[0] & g.w[0] & ((1ull << i) - 1ull));
  else {
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < W; ++q) {
      const unsigned long long below = (i >> 6) > (uint32_t)q ? ~0ull : (i >> 6) == (uint32_t)q ? (1ull << (i & 63u)) - 1ull : 0ull;
      c += (uint32_t)__popcll(a.w[q] & g.w[q] & below);
    }
    return c;
  }
}
// the mask shifted down by one bit (node masks -> branch masks: branch b hangs above node b + 1)
template <int W> __device__ __forceinline__ P3Mask<W> p3_shr1(const P3Mask<W> &m) {
  P3Mask<W> r;
#pragma unroll
  for (int q = 0; q < W; ++q) r.w[q] = (m.w[q] >> 1) | (q + 1 < W ? m.w[q + 1 < W ? q + 1 : q] << 63 : 0ull);
  return r;
}

#ifndef EPV_P3_KREG
#define EPV_P3_KREG 3u   /* segments of a heavy branch the pair pass keeps in registers (8 doubles each); longer ones go through memory */
#endif
#ifndef EPV_P3_NP
#define EPV_P3_NP 2      /* chunks of 64 pairs the merge pass interleaves */
#endif
#ifndef EPV_P3_MINBLOCKS
#define EPV_P3_MINBLOCKS 3   /* blocks of four waves per CU the register allocation aims for (<= 168 VGPRs) */
#endif
template <int W>
__global__ __launch_bounds__(256, EPV_P3_MINBLOCKS) void epv_mh_propose3_kernel(
    EpvDev S, uint32_t colour, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep, uint64_t first,
    uint64_t last, uint64_t own_first, uint64_t own_last, uint32_t list_cap, uint32_t n_qrows, uint32_t n_up,
    uint32_t depth, uint32_t parity, unsigned long long *counters, double *gpool, const double *segtab,
    const uint32_t *nodetab, uint32_t *slab_flags, uint32_t slab_slots) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  constexpr uint32_t HREC = EPV_HREC_SHORT, LEN_AT = HREC - 2u, INFO_AT = HREC - 1u;
#ifdef EPV_P2_PROFILE
  unsigned long long t_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif
  const uint32_t N = S.N, B = S.B;
  const uint32_t const_dbl = (20u + N + 1u) & ~1u;
  const uint32_t tab_dbl = B * 4u * EPV_SEGTAB_DBL;
  // tree tables: node words [N], internal nodes deepest level first [n_up] with level starts
  // [depth + 2] (level d from s_upstart[d + 1] to s_upstart[d]), all nodes but the root by depth
  // [N - 1] with level starts [depth + 2] (level d from s_dnstart[d] to s_dnstart[d + 1])
  // the node masks of the pair groups [4 * depth] (group 0 = leaves, d = internal nodes of depth d) and
  // the nodes' depths [N]
  const uint32_t tree_words = N + n_up + (depth + 2u) + (N - 1u) + (depth + 2u) + 4u * depth + N;
  const uint32_t tree_dbl = (tree_words + 1u) / 2u;
  // per wave: p.front and result bits of the pairs, each lane's first slot per group, the groups' starts
  const uint32_t goff_dbl = (depth * 64u * 2u + 7u) / 8u, gstart_dbl = (depth + 2u + 1u) / 2u;
  const uint32_t wave_dbl = EPV_P3_PCAP * 3u + EPV_P3_PCAP / 8u + goff_dbl + gstart_dbl;
  const uint32_t wave_id = threadIdx.x >> 6;
  double *s_const = s_mem;
  double *s_tab = s_mem + const_dbl;
  uint32_t *s_tree = reinterpret_cast<uint32_t *>(s_mem + const_dbl + tab_dbl);
  const uint32_t *s_up = s_tree + N, *s_upstart = s_up + n_up;
  const uint32_t *s_dn = s_upstart + (depth + 2u), *s_dnstart = s_dn + (N - 1u);
  const uint32_t *s_gmask = s_dnstart + (depth + 2u), *s_dep = s_gmask + 4u * depth;
  double *s_pf = s_mem + const_dbl + tab_dbl + tree_dbl + (size_t)wave_id * wave_dbl;
  uint8_t *s_pb = reinterpret_cast<uint8_t *>(s_pf + EPV_P3_PCAP * 2u);
  uint16_t *s_goff = reinterpret_cast<uint16_t *>(s_pf + EPV_P3_PCAP * 2u + EPV_P3_PCAP / 8u);
  uint32_t *s_gstart = reinterpret_cast<uint32_t *>(s_pf + EPV_P3_PCAP * 2u + EPV_P3_PCAP / 8u + goff_dbl);
  // the pair list (read by five passes: in LDS, a round trip less in each)
  unsigned long long *plist = reinterpret_cast<unsigned long long *>(s_pf + EPV_P3_PCAP * 2u + EPV_P3_PCAP / 8u + goff_dbl + gstart_dbl);
  auto group_mask = [&](uint32_t g) __attribute__((always_inline)) -> P3Mask<W> {
    P3Mask<W> m;
#pragma unroll
    for (int q = 0; q < W; ++q)
      m.w[q] = (unsigned long long)s_gmask[4u * g + 2u * q] | ((unsigned long long)s_gmask[4u * g + 2u * q + 1u] << 32);
    return m;
  };
  const int lane = epv_lane();
  const uint32_t my_shard = (blockIdx.x * (blockDim.x >> 6) + wave_id) & (EPV_SHARDS - 1u);
  // per-wave slab: q rows of 64 interleaved records, the flat heavy list.  slab_slots != 0: the block
  // takes a slab from a pool of `slab_slots` per XCD (a flag each, claimed with a compare-and-swap)
  // instead of owning one by its index -- a slab is scratch for the lifetime of a wave, and with one
  // per RESIDENT block the slabs' lines are reused while they are still in the XCD's L2 / the
  // memory-side cache instead of streaming 80 KB per wave through HBM (and the allocation shrinks
  // from phase_cap / 64 slabs to 8 x slab_slots x 4).  The pool is per XCD because the L2s of
  // different XCDs are not coherent with each other: a slab must never change XCD.
  __shared__ uint32_t s_slab;
  if (slab_slots) {
    if (threadIdx.x == 0) {
      const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;     // HW_REG_XCC_ID[3:0]
      uint32_t *pool = slab_flags + xcc * slab_slots;
      uint32_t i = (blockIdx.x >> 3) % slab_slots, tries = 0;
      while (atomicCAS(&pool[i], 0u, 1u) != 0u) {
        i = i + 1u == slab_slots ? 0u : i + 1u;
        // (the pool holds more slabs than an XCD can have resident blocks of this kernel: a free one always exists)
        if (++tries > (1u << 24)) __builtin_trap();
      }
      s_slab = xcc * slab_slots + i;
    }
    __syncthreads();
  }
  const size_t slab_index = slab_slots ? (size_t)s_slab * 4u + wave_id : (size_t)blockIdx.x * (blockDim.x >> 6) + wave_id;
  double *qrows = gpool + slab_index * ((size_t)n_qrows * 128u + (size_t)list_cap * HREC);
  double *list = qrows + (size_t)n_qrows * 128u;
  const uint64_t gfirst = S.g0 + first;
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)(gfirst % 3u)) % 3u);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site = s0 + 3u * tid;
  const bool valid = site <= last;
  const uint64_t n = S.n;
  const uint32_t gsite = (uint32_t)(S.g0 + site);
  const uint32_t gsite_lane0 = gsite - 3u * (uint32_t)lane;

  uint32_t selL = 0, selM = 0, selR = 0;
  if (valid) { selL = S.sel[site - 1]; selM = S.sel[site]; selR = S.sel[site + 1]; }
  for (uint32_t i = threadIdx.x; i < tab_dbl; i += blockDim.x) s_tab[i] = segtab[i];
  for (uint32_t i = threadIdx.x; i < tree_words; i += blockDim.x) s_tree[i] = nodetab[i];
  stage_constants(S, s_const);
  const double *s_rates = s_const;
  const double *s_blen = s_const + 20;

  const uint64_t Bn = (uint64_t)B * n, Cn = (uint64_t)S.C * n;
  // (the plane offsets of the three columns are recomputed from the site and the buffer bits where they
  // are needed: ten registers that would otherwise live through the whole kernel)
  const uint32_t selbits = selL | (selM << 1) | (selR << 2);
  const uint32_t tid32 = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site_lane0 = s0 + 3u * (uint64_t)(tid32 - (uint32_t)lane);
  // ---- 0. the meta words of the three columns, in batches of independent loads, condensed to six
  //         masks (bit `node` = the branch above the node)
  P3Mask<W> mR, mL, mM, mMp, mMj, mH;
  p3_zero(mR); p3_zero(mL); p3_zero(mM); p3_zero(mMp); p3_zero(mMj); p3_zero(mH);
  uint32_t heavy = 0, n_pairs = 0;
  if (valid) {
    const uint64_t mbaseL = (selL ? Bn : 0ull) + (site - 1), mbaseR = (selR ? Bn : 0ull) + (site + 1);
    const uint64_t mbaseM = (selM ? Bn : 0ull) + site;
#pragma unroll 6
    for (uint32_t b = 0; b < B; ++b) {
      const uint32_t wL = S.meta[mbaseL + (uint64_t)b * n];
      const uint32_t wR = S.meta[mbaseR + (uint64_t)b * n];
      const uint32_t wM = S.meta[mbaseM + (uint64_t)b * n];
      const uint32_t K = (wL & EPV_NJ_MASK) + (wR & EPV_NJ_MASK) + 1u;
      const uint32_t node = b + 1u;
      p3_or(mR, node, wR >> EPV_INIT_SHIFT);
      p3_or(mL, node, wL >> EPV_INIT_SHIFT);
      p3_or(mM, node, wM >> EPV_INIT_SHIFT);
      p3_or(mMp, node, wM & 1u);
      p3_or(mMj, node, (wM & EPV_NJ_MASK) ? 1u : 0u);
      if (K >= 2u) { p3_or(mH, node, 1u); heavy += K; ++n_pairs; }
    }
  }
  const uint32_t root_state = (uint32_t)(mM.w[0] >> 1) & 1u;     // init of branch 0's path (PATH(1, site)->init)

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
    // The pairs are grouped by level -- leaf branches, then the internal nodes of depth 1, 2, ... --
    // so that a level's pair pass finds its pairs side by side: within a group by lane, then by node
    {
      uint32_t gb = 0u;
      for (uint32_t g = 0; g < depth; ++g) {
        const uint32_t cnt = run ? p3_count(mH, group_mask(g)) : 0u;
        const uint32_t incl = wave_incl_scan_u32(cnt);
        s_goff[g * 64u + lane] = (uint16_t)(gb + incl - cnt);
        if (lane == 0) s_gstart[g] = gb;
        gb += epv_bcast(incl, 63);
      }
      if (lane == 0) s_gstart[depth] = gb;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // slot of this lane's heavy pair at `node` in the pair list (and in s_pf / s_pb)
    auto slot_of = [&](uint32_t node) __attribute__((always_inline)) -> uint32_t {
      const uint32_t g = EPV_P3_LEAF(s_tree[node]) ? 0u : s_dep[node];
      return (uint32_t)s_goff[g * 64u + lane] + p3_count_below(mH, group_mask(g), node);
    };

    P2_MARK(1);
    // ---- 1. the heavy (lane, node) pairs, up to four per step so that the re-reads of their meta
    //         words (the segment counts) share a round trip
    if (run && n_pairs) {
      const uint64_t mbaseL = ((selbits & 1u) ? Bn : 0ull) + (site - 1), mbaseR = ((selbits & 4u) ? Bn : 0ull) + (site + 1);
      uint32_t hcur = hbase;
#pragma unroll
      for (int hw = 0; hw < W; ++hw) {
      unsigned long long h = mH.w[hw];
      while (h) {
        uint32_t nd[4], K[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          nd[q] = h ? 64u * (uint32_t)hw + (uint32_t)(__ffsll((long long)h) - 1) : 0u;
          if (h) h &= h - 1ull;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          K[q] = nd[q] ? (uint32_t)(S.meta[mbaseL + (uint64_t)(nd[q] - 1u) * n] & EPV_NJ_MASK) +
                             (uint32_t)(S.meta[mbaseR + (uint64_t)(nd[q] - 1u) * n] & EPV_NJ_MASK) + 1u
                       : 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!nd[q]) continue;
          const uint32_t leaf_state = p3_get(mM, nd[q]) ^ p3_get(mMp, nd[q]);
          plist[slot_of(nd[q])] = (unsigned long long)lane | ((unsigned long long)nd[q] << 6) | ((unsigned long long)K[q] << 13) |
                        ((unsigned long long)hcur << 32) | ((unsigned long long)leaf_state << 62);
          hcur += K[q];
        }
      }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 2. one pair per lane: forward merge of the neighbours' jumps (Segment.cpp:35-79) into the
    //         records' length and address fields.  A few chunks of pairs at a time, stage by stage, so
    //         that their round trips (pair word -> meta words -> first jumps) overlap
    for (uint32_t p0 = 0; p0 < totP; p0 += (uint32_t)EPV_P3_NP * 64u) {
      constexpr int NP = EPV_P3_NP;
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
        const uint32_t owner = (uint32_t)pr[q] & 63u, b = (EPV_P3_PAIR_NODE(pr[q]) - 1u) & 127u;
        const uint32_t osel = (uint32_t)__shfl((int)selbits, (int)owner);
        const uint64_t osite = site_lane0 + 3u * (uint64_t)owner;
        const uint64_t ml = ((osel & 1u) ? Bn : 0ull) + (osite - 1), mr = ((osel & 4u) ? Bn : 0ull) + (osite + 1);
        const uint64_t jl = ((osel & 1u) ? Bn * S.C : 0ull) + (osite - 1), jr = ((osel & 4u) ? Bn * S.C : 0ull) + (osite + 1);
        cL[q] = act[q] ? (uint32_t)S.meta[ml + (uint64_t)b * n] : 0u;
        cR[q] = act[q] ? (uint32_t)S.meta[mr + (uint64_t)b * n] : 0u;
        Lj[q] = S.jumps + jl + (uint64_t)b * Cn;
        Rj[q] = S.jumps + jr + (uint64_t)b * Cn;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): every meta word in before the first conditional load (epv_accept3.h)
      double tl0[NP], tr0[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        tl0[q] = (cL[q] & EPV_NJ_MASK) ? Lj[q][0] : EPV_INF;
        tr0[q] = (cR[q] & EPV_NJ_MASK) ? Rj[q][0] : EPV_INF;
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        if (!act[q]) continue;
        const uint32_t owner = (uint32_t)pr[q] & 63u, node = EPV_P3_PAIR_NODE(pr[q]), hcur = (uint32_t)(pr[q] >> 32) & 0xfffffu;
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
    //         (two chunks of segments per step: their loads share a round trip)
    auto eval_segment = [&](double *rec, double len, uint64_t info) __attribute__((always_inline)) {
      const uint32_t trip0 = (uint32_t)info & 7u, owner = (uint32_t)(info >> 3) & 63u;
      const uint32_t node = (uint32_t)(info >> 9) & 4095u, k = (uint32_t)(info >> 21);
      double m[6];
      epv_seg_matrices(len, s_rates[trip0], s_rates[trip0 | 2u], m);
      const epv_block2 blk = epv_keyed_block(seed_lo, seed_hi, gsite_lane0 + 3u * owner, sweep, node, k, 0u, 0u);
#pragma unroll
      for (int q = 0; q < 6; ++q) rec[q] = m[q];
      rec[6] = blk.d0;
      rec[7] = blk.d1;
    };
    for (uint32_t i = (uint32_t)lane; i < totH; i += 128u) {
      double *recA = list + (size_t)i * HREC, *recB = recA + 64u * HREC;
      const bool hasB = i + 64u < totH;
      const double lenA = recA[LEN_AT];
      const uint64_t infoA = epv_d2u(recA[INFO_AT]);
      const double lenB = hasB ? recB[LEN_AT] : 0.0;
      const uint64_t infoB = hasB ? epv_d2u(recB[INFO_AT]) : 0ull;
      eval_segment(recA, lenA, infoA);
      if (hasB) eval_segment(recB, lenB, infoB);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    P2_MARK(3);

    // The heavy pairs of one level of the tree, one pair per lane: the branch's pruning chain
    // p[k] = M[k] p[k + 1] from q of its node, then the chain of end states for both start states.
    // level 0 = the leaf branches (q is the observed state), d >= 1 = the internal nodes of depth d
    // (q from the row the nodes' pass has just written).
    auto pair_pass = [&](uint32_t level) __attribute__((always_inline)) {
      const uint32_t pend = s_gstart[level + 1u];
      for (uint32_t p0 = s_gstart[level]; p0 < pend; p0 += 64u) {
        const uint32_t pidx = p0 + (uint32_t)lane;
        if (pidx >= pend) continue;
        const unsigned long long pr = plist[pidx];
        const uint32_t owner = (uint32_t)pr & 63u, node = EPV_P3_PAIR_NODE(pr), K = EPV_P3_PAIR_K(pr);
        const uint32_t hrec0 = (uint32_t)(pr >> 32) & 0xfffffu;
        const uint32_t nw = s_tree[node];
        double q0, q1;
        if (EPV_P3_LEAF(nw)) {
          const uint32_t leaf_state = (uint32_t)(pr >> 62) & 1u;
          q0 = leaf_state ? 0.0 : 1.0;
          q1 = leaf_state ? 1.0 : 0.0;
        } else {
          const double *qr = qrows + (size_t)EPV_P3_QROW(nw) * 128u + (size_t)owner * 2u;
          q0 = qr[0]; q1 = qr[1];
        }
        double *h0 = list + (size_t)hrec0 * HREC;
        uint32_t prevA = 0u, prevB = 1u;            // chains from start state 0 and from start state 1
        bool cleanA = true, cleanB = true;
        unsigned long long wA = 0ull, wB = 0ull;
        double pf0, pf1;
        if (K <= EPV_P3_KREG) {
          // everything from registers: the records in one batch of loads
          double R[EPV_P3_KREG][8];
#pragma unroll
          for (int k = 0; k < (int)EPV_P3_KREG; ++k) {
#pragma unroll
            for (int f = 0; f < 8; ++f) R[k][f] = 0.0;
            if ((uint32_t)k < K) {
#pragma unroll
              for (int f = 0; f < 8; ++f) R[k][f] = h0[(size_t)k * HREC + f];
            }
          }
          double p[EPV_P3_KREG + 1][2];
          p[EPV_P3_KREG][0] = q0; p[EPV_P3_KREG][1] = q1;
#pragma unroll
          for (int k = (int)EPV_P3_KREG - 1; k >= 0; --k) {
            if ((uint32_t)k < K) {
              const double n0 = ((uint32_t)k + 1u == K) ? q0 : p[k + 1][0], n1 = ((uint32_t)k + 1u == K) ? q1 : p[k + 1][1];
              const double P00 = R[k][0], P11 = R[k][1];
              const double P01 = 1.0 - P00, P10 = 1.0 - P11;
              p[k][0] = P00 * n0 + P01 * n1;
              p[k][1] = P10 * n0 + P11 * n1;
            } else { p[k][0] = 0.0; p[k][1] = 0.0; }
          }
          pf0 = p[0][0]; pf1 = p[0][1];
#pragma unroll
          for (int k = 0; k < (int)EPV_P3_KREG; ++k) {
            if ((uint32_t)k < K) {
              const bool last_seg = ((uint32_t)k + 1u == K);
              const double nxt0 = last_seg ? q0 : p[k + 1][0];
              const double pk0 = p[k][0], pk1 = p[k][1];
              const double PT00 = R[k][2], PT10 = R[k][3], nb0 = R[k][4], nb1 = R[k][5], u_end = R[k][6], u_first = R[k][7];
              const double p0A = (prevA ? PT10 : PT00) * nxt0 / (prevA ? pk1 : pk0);
              const double p0B = (prevB ? PT10 : PT00) * nxt0 / (prevB ? pk1 : pk0);
              const uint32_t sA = (u_end > p0A) ? 1u : 0u, sB = (u_end > p0B) ? 1u : 0u;
              cleanA = cleanA && (sA == prevA) && (1.0 - u_first < (prevA ? nb1 : nb0));
              cleanB = cleanB && (sB == prevB) && (1.0 - u_first < (prevB ? nb1 : nb0));
              wA |= (unsigned long long)sA << k;
              wB |= (unsigned long long)sB << k;
              prevA = sA; prevB = sB;
            }
          }
        } else {
          // a long branch: the partials through the records (they take the place of the matrices)
          double n0 = q0, n1 = q1;
          for (uint32_t kk = K; kk-- > 0u;) {
            double *hr = h0 + (size_t)kk * HREC;
            const double P00 = hr[0], P11 = hr[1];
            const double P01 = 1.0 - P00, P10 = 1.0 - P11;
            const double a = P00 * n0 + P01 * n1;
            const double c = P10 * n0 + P11 * n1;
            hr[0] = a; hr[1] = c;
            n0 = a; n1 = c;
          }
          pf0 = n0; pf1 = n1;
          const double *hr = h0;
          double pk0 = pf0, pk1 = pf1;
          for (uint32_t k = 0; k < K; ++k) {
            const bool last_seg = (k + 1u == K);
            const double nxt0 = last_seg ? q0 : hr[HREC], nxt1 = last_seg ? q1 : hr[HREC + 1u];
            const double PT00 = hr[2], PT10 = hr[3], nb0 = hr[4], nb1 = hr[5], u_end = hr[6], u_first = hr[7];
            const double p0A = (prevA ? PT10 : PT00) * nxt0 / (prevA ? pk1 : pk0);
            const double p0B = (prevB ? PT10 : PT00) * nxt0 / (prevB ? pk1 : pk0);
            const uint32_t sA = (u_end > p0A) ? 1u : 0u, sB = (u_end > p0B) ? 1u : 0u;
            cleanA = cleanA && (sA == prevA) && (1.0 - u_first < (prevA ? nb1 : nb0));
            cleanB = cleanB && (sB == prevB) && (1.0 - u_first < (prevB ? nb1 : nb0));
            if (k < 64u) { wA |= (unsigned long long)sA << k; wB |= (unsigned long long)sB << k; }
            prevA = sA; prevB = sB;
            pk0 = nxt0; pk1 = nxt1;
            hr += HREC;
          }
        }
        // the words of sampled states wait in the first record's PT slots (used by now) for the pass
        // behind the downward walk, which knows the start state; a branch of more than 64 segments
        // keeps its record and is walked again there
        if (K <= 64u) { h0[2] = epv_u2d(wA); h0[3] = epv_u2d(wB); }
        s_pf[2u * pidx] = pf0; s_pf[2u * pidx + 1u] = pf1;
        s_pb[pidx] = (uint8_t)(prevA | ((cleanA ? 1u : 0u) << 1) | (prevB << 2) | ((cleanB ? 1u : 0u) << 3) |
                               ((K == 2u || K >= 4u) ? 16u : 0u) | (K >= 3u ? 32u : 0u));
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    // q of a leaf: the observed state (the end state of the site's current path)
    auto leaf_q = [&](uint32_t node, double &q0, double &q1) __attribute__((always_inline)) {
      const uint32_t leaf_state = p3_get(mM, node) ^ p3_get(mMp, node);
      q0 = leaf_state ? 0.0 : 1.0;
      q1 = leaf_state ? 1.0 : 0.0;
    };

    // ---- 4. pruning (SingleSiteSampler.cpp:116-157), deepest level first.  A level: q of its internal
    //         nodes from their children's p.front -- a heavy child's from its pair lane, any other
    //         recomputed from the child's q and the matrix table -- then the level's heavy pairs.
    //         (one loop, so that the pair pass -- the largest piece of code here -- is instantiated once)
    for (uint32_t d = depth; d >= 1u; --d) {
      if (run && d < depth) {
        const uint32_t i0 = s_upstart[d + 1u], i1 = s_upstart[d];     // (deepest first: level d + 1 precedes level d)
        for (uint32_t ib = i0; ib < i1; ib += 4u) {
          // the q rows of up to eight internal children in one batch
          double cq[4][2][2];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int w = 0; w < 2; ++w) { cq[j][w][0] = 0.0; cq[j][w][1] = 0.0; }
            if (ib + (uint32_t)j < i1) {
              const uint32_t nw = s_tree[s_up[ib + j]];
#pragma unroll
              for (int w = 0; w < 2; ++w) {
                const uint32_t c = w ? EPV_P3_CHILD2(nw) : EPV_P3_CHILD1(nw);
                if (c == 0u) continue;
                const uint32_t cw = s_tree[c];
                if (!EPV_P3_LEAF(cw) && !p3_get(mH, c)) {
                  const double *qr = qrows + (size_t)EPV_P3_QROW(cw) * 128u + (size_t)lane * 2u;
                  cq[j][w][0] = qr[0]; cq[j][w][1] = qr[1];
                }
              }
            }
          }
#pragma unroll 1
          for (uint32_t j = 0; j < 4u; ++j) {
            if (ib + j >= i1) break;
            const uint32_t P = __builtin_amdgcn_readfirstlane(s_up[ib + j]);
            const uint32_t nw = __builtin_amdgcn_readfirstlane(s_tree[P]);
            double a0 = 1.0, a1 = 1.0;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              const uint32_t c = w ? EPV_P3_CHILD2(nw) : EPV_P3_CHILD1(nw);
              if (c == 0u) continue;
              double f0, f1;
              if (p3_get(mH, c)) {
                const uint32_t sl = slot_of(c);
                f0 = s_pf[2u * sl]; f1 = s_pf[2u * sl + 1u];
              } else {
                double q0 = j == 0u ? cq[0][w][0] : j == 1u ? cq[1][w][0] : j == 2u ? cq[2][w][0] : cq[3][w][0];
                double q1 = j == 0u ? cq[0][w][1] : j == 1u ? cq[1][w][1] : j == 2u ? cq[2][w][1] : cq[3][w][1];
                if (EPV_P3_LEAF(s_tree[c])) leaf_q(c, q0, q1);
                const uint32_t ctx = (p3_get(mL, c) << 1) | p3_get(mR, c);
                const double *t = s_tab + ((c - 1u) * 4u + ctx) * EPV_SEGTAB_DBL;     // (32-bit index: an LDS address)
                const double P00 = t[0], P11 = t[1];
                const double P01 = 1.0 - P00, P10 = 1.0 - P11;
                f0 = P00 * q0 + P01 * q1;
                f1 = P10 * q0 + P11 * q1;
              }
              a0 *= f0;     // q *= p.front of the child, children in pre-order (:121-127)
              a1 *= f1;
            }
            double *qr = qrows + (size_t)EPV_P3_QROW(nw) * 128u + (size_t)lane * 2u;
            qr[0] = a0; qr[1] = a1;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      pair_pass(d == depth ? 0u : d);
    }
    P2_MARK(4);

    // ---- 5. downward sampling of the segment END STATES (:180-255), level by level from the root;
    //         the jump times are drawn by epv_mh_jumps_kernel for the dirty branches only
    P3Mask<W> mEnd, mStart, dirty, multi, deep;
    p3_zero(mEnd); p3_zero(mStart); p3_zero(dirty); p3_zero(multi); p3_zero(deep);
    const uint64_t pcW = (uint64_t)S.phase_cap * S.W, tidW = tid * S.W;
    bool ident = true;
    if (run) {
      for (uint32_t d = 1u; d <= depth; ++d) {
        const uint32_t i0 = s_dnstart[d], i1 = s_dnstart[d + 1u];
        for (uint32_t ib = i0; ib < i1; ib += 4u) {
          double cq[4][2];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            cq[j][0] = 0.0; cq[j][1] = 0.0;
            if (ib + (uint32_t)j < i1) {
              const uint32_t c = s_dn[ib + j];
              const uint32_t cw = s_tree[c];
              if (!EPV_P3_LEAF(cw) && !p3_get(mH, c)) {
                const double *qr = qrows + (size_t)EPV_P3_QROW(cw) * 128u + (size_t)lane * 2u;
                cq[j][0] = qr[0]; cq[j][1] = qr[1];
              }
            }
          }
          // (one node at a time: four interleaved Philox blocks cost more registers than they hide latency)
#pragma unroll 1
          for (uint32_t j = 0; j < 4u; ++j) {
            if (ib + j >= i1) break;
            // (wave-uniform by construction: say so, and what follows from them is scalar work)
            const uint32_t node = __builtin_amdgcn_readfirstlane(s_dn[ib + j]), b = node - 1u;
            const uint32_t nw = __builtin_amdgcn_readfirstlane(s_tree[node]);
            const uint32_t par = EPV_P3_PARENT(nw);
            const uint32_t start_state = (par == 0u) ? root_state : p3_get(mEnd, par);
            uint32_t prev;
            bool clean, b_multi = false, b_deep = false;
            if (!p3_get(mH, node)) {
              double q0 = j == 0u ? cq[0][0] : j == 1u ? cq[1][0] : j == 2u ? cq[2][0] : cq[3][0];
              double q1 = j == 0u ? cq[0][1] : j == 1u ? cq[1][1] : j == 2u ? cq[2][1] : cq[3][1];
              if (EPV_P3_LEAF(nw)) leaf_q(node, q0, q1);
              const uint32_t ctx = (p3_get(mL, node) << 1) | p3_get(mR, node);
              const double *t = s_tab + (b * 4u + ctx) * EPV_SEGTAB_DBL;
              const double P00 = t[0], P11 = t[1];
              const double P01 = 1.0 - P00, P10 = 1.0 - P11;
              const double pk0 = P00 * q0 + P01 * q1;     // p.front, as pruning computed it
              const double pk1 = P10 * q0 + P11 * q1;
              const double PT0 = start_state ? t[3] : t[2];
              const double nb = start_state ? t[5] : t[4];
              const epv_block2 blk = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, 0u, 0u, 0u);
              const double p0 = PT0 * q0 / (start_state ? pk1 : pk0);
              prev = (blk.d0 > p0) ? 1u : 0u;
              clean = (prev == start_state) && (1.0 - blk.d1 < nb);
              if (!clean) S.prop_states[(uint64_t)b * pcW + tidW] = (unsigned long long)prev;
            } else {
              const uint32_t r = s_pb[slot_of(node)];
              prev = (r >> (start_state ? 2u : 0u)) & 1u;
              clean = (r >> (start_state ? 3u : 1u)) & 1u;
              b_multi = (r >> 4) & 1u; b_deep = (r >> 5) & 1u;
            }
            p3_or(mEnd, node, prev);     // proposal end state for the children
            p3_or(mStart, node, start_state);
            // same as the current path?  (no jumps on either, same start state)
            ident = ident && clean && !p3_get(mMj, node) && p3_get(mM, node) == start_state;
            if (!clean) {
              p3_or(dirty, b, 1u);
              if (b_multi) p3_or(multi, b, 1u);   // four buckets by segment count
              if (b_deep) p3_or(deep, b, 1u);
            }
          }
        }
      }
    }
    P2_MARK(5);
    // ---- 6. the words of sampled states of the dirty heavy branches, one pair per lane again
    for (uint32_t p0 = 0; p0 < totP; p0 += 64u) {
      const uint32_t pidx = p0 + (uint32_t)lane;
      const unsigned long long pr = pidx < totP ? plist[pidx] : 0ull;
      const uint32_t owner = (uint32_t)pr & 63u, node = EPV_P3_PAIR_NODE(pr);
      P3Mask<W> oEnd;
#pragma unroll
      for (int q = 0; q < W; ++q)
        oEnd.w[q] = (unsigned long long)__shfl((uint32_t)mEnd.w[q], (int)owner) | ((unsigned long long)__shfl((uint32_t)(mEnd.w[q] >> 32), (int)owner) << 32);
      const uint32_t oRoot = (uint32_t)__shfl((int)root_state, (int)owner);
      if (pidx < totP) {
        const uint32_t K = EPV_P3_PAIR_K(pr), hrec0 = (uint32_t)(pr >> 32) & 0xfffffu;
        const uint32_t nw = s_tree[node];
        const uint32_t par = EPV_P3_PARENT(nw);
        const uint32_t st = (par == 0u) ? oRoot : p3_get(oEnd, par);
        const uint32_t r = s_pb[pidx];
        const bool clean = (r >> (st ? 3u : 1u)) & 1u;
        if (!clean) {
          uint64_t *states = S.prop_states + ((uint64_t)(node - 1u) * S.phase_cap + (tid - (uint32_t)lane + owner)) * S.W;
          const double *h0 = list + (size_t)hrec0 * HREC;
          if (K <= 64u) {
            states[0] = epv_d2u(h0[2u + st]);
          } else {
            // more than 64 segments: walk the branch again from its start state, word by word
            double q0, q1;
            if (EPV_P3_LEAF(nw)) {
              const uint32_t leaf_state = (uint32_t)(pr >> 62) & 1u;
              q0 = leaf_state ? 0.0 : 1.0;
              q1 = leaf_state ? 1.0 : 0.0;
            } else {
              const double *qr = qrows + (size_t)EPV_P3_QROW(nw) * 128u + (size_t)owner * 2u;
              q0 = qr[0]; q1 = qr[1];
            }
            uint32_t prev = st;
            unsigned long long word = 0ull;
            const double *hr = h0;
            double pk0 = hr[0], pk1 = hr[1];
            for (uint32_t k = 0; k < K; ++k) {
              const bool last_seg = (k + 1u == K);
              const double nxt0 = last_seg ? q0 : hr[HREC], nxt1 = last_seg ? q1 : hr[HREC + 1u];
              const double p0 = (prev ? hr[3] : hr[2]) * nxt0 / (prev ? pk1 : pk0);
              const uint32_t sampled = (hr[6] > p0) ? 1u : 0u;
              word |= (unsigned long long)sampled << (k & 63u);
              if ((k & 63u) == 63u) { states[k >> 6] = word; word = 0ull; }
              prev = sampled;
              pk0 = nxt0; pk1 = nxt1;
              hr += HREC;
            }
            if (K & 63u) states[(K - 1u) >> 6] = word;
          }
        }
      }
    }
    // (a dirty branch of one segment hands the jump kernel all it needs in the task word)
    {
      const P3Mask<W> bH = p3_shr1(mH), bStart = p3_shr1(mStart), bEnd = p3_shr1(mEnd), bL = p3_shr1(mL), bR = p3_shr1(mR);
#pragma unroll
      for (int q = 0; q < W; ++q) {     // (a wave-wide operation per word of 64 branches)
        if (64u * (uint32_t)q >= B) break;
        const uint32_t b_last = 64u * (uint32_t)q + 63u < B ? 64u * (uint32_t)q + 63u : B - 1u;
        epv_flush_tasks(S, counters, dirty.w[q], multi.w[q], deep.w[q], b_last, site, lane, my_shard, dirty.w[q] & ~bH.w[q],
                        ((selbits >> 1) & 1u) ^ 1u, bStart.w[q], bEnd.w[q], bL.w[q], bR.w[q]);
      }
    }
    P2_MARK(6);

    // ---- 7. hand-over.  A proposal equal to the current path is accepted with probability one and
    //         changes neither the paths nor the cached likelihoods: count it and be done.  Everything
    //         else: start states of the proposal's branches into the other buffer, and the site onto
    //         the accept list of this wave's shard.
    const bool to_list = run && !ident;
    if (to_list) {
      for (uint32_t node = 1u; node < N; ++node) {
        const uint32_t par = EPV_P3_PARENT(s_tree[node]);
        const uint32_t st = (par == 0u) ? root_state : p3_get(mEnd, par);
        S.meta[((selbits & 2u) ? 0ull : Bn) + (uint64_t)(node - 1u) * n + site] = (epv_meta_t)(st << EPV_INIT_SHIFT);
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
    __builtin_amdgcn_wave_barrier();     // (a later round reuses the pair list, the rows and the records)
    pending = pending && !run;
    P2_MARK(7);
  }
  if (slab_slots) {
    // every store of this block into the slab has reached the L2 before the flag lets the next block in
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) atomicExch(&slab_flags[s_slab], 0u);
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
