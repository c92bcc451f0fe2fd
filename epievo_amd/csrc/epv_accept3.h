#ifndef EPV_ACCEPT3_H
#define EPV_ACCEPT3_H
// epv_accept3.h -- acceptance of a colour phase's proposals on LARGE trees (included by
// epv_kernels.h).  Same decisions as epv_mh_accept_kernel, list mode or not (log_accept_rate
// SingleSiteSampler.cpp:396-433, Metropolis_Hastings_site :510-533), organised around what bounds
// that kernel on the 16-leaf tree: with 30 branches the meta words of a site's five columns do not
// fit its LDS cache, so a lane walks three triples x 30 branches with a dependent global round trip
// for the meta words and another for the jump planes at nearly every step -- 47 M instructions in
// 445 us, a sixth of its issue bound (profiles/r03f_pmc_valu_bal16.csv).  Here
//   * every (site, triple) pair has its own lane (the three likelihoods of a site are independent):
//     21 sites per wave, a third of the chain per lane; the middle lane of a site collects the three
//     values by lane shuffles and decides;
//   * the branch loop runs in groups: the meta words of a group's 3 x G paths in one batch of loads,
//     then the first jump of every path that has one in a second batch, then the merges from
//     registers (a later jump of a path -- rare -- is fetched inside the merge as before).
// Two round trips per G branches instead of two per branch.

template <class ACC>
__device__ __forceinline__ void merge3_pre(const PathRef &L, const PathRef &M, const PathRef &R, double tl, double tm,
                                           double tr, uint64_t n, double tot_time, ACC &A) {
  // merge3 (Path.cpp:206-301) with the first jump of each path already loaded
  int ctx = (int)(4u * L.init + 2u * M.init + R.init);
  double prev = 0.0;
  uint32_t i = 0, j = 0, k = 0;
  while (i < L.nj || j < M.nj || k < R.nj) {
    if (tl < (tm < tr ? tm : tr)) {
      acc_add(A, ctx, tl - prev, false);
      prev = tl; ctx ^= 4; ++i;
      tl = i < L.nj ? L.j[(uint64_t)i * n] : EPV_INF;
    } else if (tm < tr) {
      acc_add(A, ctx, tm - prev, true);
      prev = tm; ctx ^= 2; ++j;
      tm = j < M.nj ? M.j[(uint64_t)j * n] : EPV_INF;
    } else {
      acc_add(A, ctx, tr - prev, false);
      prev = tr; ctx ^= 1; ++k;
      tr = k < R.nj ? R.j[(uint64_t)k * n] : EPV_INF;
    }
  }
  acc_add(A, ctx, tot_time - prev, false);
}

#ifndef EPV_ACC3_GROUP
#define EPV_ACC3_GROUP 4   /* branches per batch of loads (5 or 6: more registers than 4 waves per SIMD have, no faster) */
#endif

// path_log_likelihood (SingleSiteSampler.cpp:374-391) of one triple, branches in groups
template <class ACC>
__device__ __forceinline__ double triple_llh_grouped(const EpvDev &S, const double *s_model, const double *s_blen,
                                                     uint32_t bl, uint64_t sl, uint32_t bm, uint64_t sm, uint32_t br,
                                                     uint64_t sr, ACC &A) {
  constexpr uint32_t G = EPV_ACC3_GROUP;
  acc_clear(A);
  const uint32_t B = S.B;
  const uint64_t n = S.n, Cn = (uint64_t)S.C * n;
  const uint64_t ml0 = (bl ? (uint64_t)B * n : 0ull) + sl, mm0 = (bm ? (uint64_t)B * n : 0ull) + sm,
                 mr0 = (br ? (uint64_t)B * n : 0ull) + sr;
  const uint64_t jl0 = (bl ? (uint64_t)B * Cn : 0ull) + sl, jm0 = (bm ? (uint64_t)B * Cn : 0ull) + sm,
                 jr0 = (br ? (uint64_t)B * Cn : 0ull) + sr;
  uint32_t rl = 0, rm = 0, rr = 0;
  for (uint32_t b0 = 0; b0 < B; b0 += G) {
    uint32_t wl[G], wm[G], wr[G];
#pragma unroll
    for (uint32_t q = 0; q < G; ++q) {
      const bool in = b0 + q < B;
      const uint64_t off = (uint64_t)(b0 + q) * n;
      wl[q] = in ? (uint32_t)S.meta[ml0 + off] : 0u;
      wm[q] = in ? (uint32_t)S.meta[mm0 + off] : 0u;
      wr[q] = in ? (uint32_t)S.meta[mr0 + off] : 0u;
    }
    // all meta words in before the first conditional load goes out: behind a load under a lane mask the
    // compiler can no longer count what is in flight and waits for EVERYTHING at the next use of a meta
    // word -- the first jumps would leave one round trip apart instead of together
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    double tl[G], tm[G], tr[G];
#pragma unroll
    for (uint32_t q = 0; q < G; ++q) {
      const uint64_t off = (uint64_t)(b0 + q) * Cn;
      tl[q] = (wl[q] & EPV_NJ_MASK) ? S.jumps[jl0 + off] : EPV_INF;
      tm[q] = (wm[q] & EPV_NJ_MASK) ? S.jumps[jm0 + off] : EPV_INF;
      tr[q] = (wr[q] & EPV_NJ_MASK) ? S.jumps[jr0 + off] : EPV_INF;
    }
    if (b0 == 0u) { rl = wl[0] >> EPV_INIT_SHIFT; rm = wm[0] >> EPV_INIT_SHIFT; rr = wr[0] >> EPV_INIT_SHIFT; }
#pragma unroll
    for (uint32_t q = 0; q < G; ++q) {
      if (b0 + q >= B) break;
      const uint64_t off = (uint64_t)(b0 + q) * Cn;
      PathRef L, M, R;
      L.j = S.jumps + jl0 + off; L.nj = wl[q] & EPV_NJ_MASK; L.init = wl[q] >> EPV_INIT_SHIFT;
      M.j = S.jumps + jm0 + off; M.nj = wm[q] & EPV_NJ_MASK; M.init = wm[q] >> EPV_INIT_SHIFT;
      R.j = S.jumps + jr0 + off; R.nj = wr[q] & EPV_NJ_MASK; R.init = wr[q] >> EPV_INIT_SHIFT;
      merge3_pre(L, M, R, tl[q], tm[q], tr[q], n, s_blen[b0 + q + 1u], A);
    }
  }
  const double *rates = s_model, *lrates = s_model + 8, *T = s_model + 16;
  double llh = T[2 * rl + rm] * T[2 * rm + rr];
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += (double)acc_j(A, c) * lrates[c] - acc_d(A, c) * rates[c];
  llh += s;
  return llh;
}

#define EPV_ACC3_SITES 21u   /* sites per wave: three lanes each, lane 63 idles */
#ifndef EPV_ACC3_MINBLOCKS
#define EPV_ACC3_MINBLOCKS 4
#endif

__global__ __launch_bounds__(256, EPV_ACC3_MINBLOCKS) void epv_mh_accept3_kernel(
    EpvDev S, uint32_t colour, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep, uint64_t first, uint64_t last,
    uint64_t own_first, uint64_t own_last, unsigned long long *counters, uint32_t list_mode, uint64_t n_all) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  __shared__ double s_accd[8 * 256];
  __shared__ uint32_t s_accj[8 * 256];
  stage_constants(S, s_mem);
  const double *s_const = s_mem, *s_blen = s_mem + 20;
  const int lane = epv_lane();
  const uint32_t wave = threadIdx.x >> 6;
  AccLds A;
  A.d = s_accd + threadIdx.x; A.j = s_accj + threadIdx.x; A.stride = 256u;
  const uint64_t gfirst = S.g0 + first;
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)(gfirst % 3u)) % 3u);
  const uint32_t shard_row = blockIdx.y;
  // list_mode 1 / 2: the sites the proposal kernel listed (parity list_mode - 1), one shard of the list per
  // grid row; 0: every site of the colour (the first proposal kernel lists nothing), one grid row
  const unsigned long long n_list = list_mode ? counters[EPV_CNT_IDX(list_mode == 2u ? EPV_CNT_ALIST1 : EPV_CNT_ALIST0, shard_row)]
                                              : (unsigned long long)n_all;
  const uint32_t sidx = (uint32_t)lane / 3u, w = (uint32_t)lane - 3u * sidx;
  const uint64_t per_block = 4ull * EPV_ACC3_SITES;
  for (uint64_t base = (uint64_t)blockIdx.x * per_block; base < n_list; base += (uint64_t)gridDim.x * per_block) {
    const uint64_t li = base + (uint64_t)wave * EPV_ACC3_SITES + sidx;
    bool have = lane < 63 && li < n_list;
    const uint64_t tid = !have ? 0u : list_mode ? (uint64_t)S.alist[(uint64_t)shard_row * S.alist_cap + li] : li;
    const uint64_t site = s0 + 3u * tid;
    have = have && site <= last;
    double v = 0.0, llh_l = 0.0, llh_m = 0.0, llh_r = 0.0, llr = 0.0;
    bool ovf = false, hasLL = false, hasRR = false;
    uint32_t selM = 0;
    if (have) {
      const uint64_t g = S.g0 + site;
      hasLL = g > 1u; hasRR = g < S.n_global - 2u;
      const uint32_t selL = S.sel[site - 1], selR = S.sel[site + 1];
      selM = S.sel[site];
      const uint32_t selLL = hasLL ? S.sel[site - 2] : 0u, selRR = hasRR ? S.sel[site + 2] : 0u;
      ovf = S.prop_flag[tid] != 0;
      if (w == 1u) {      // the deciding lane's own loads, in flight while the triples are merged
        llh_l = S.tri[site - 1]; llh_m = S.tri[site]; llh_r = S.tri[site + 1];
        llr = (S.flags & (EPV_FLAG_REFERENCE_PROPOSAL_RATIO | EPV_FLAG_SAMPLE_ROOT)) ? S.prop_llr[tid] : 0.0;
      }
      const uint32_t selP = selM ^ 1u;
      const uint64_t c = site - 1u + (uint64_t)w;
      const uint32_t bl = (w == 0u) ? selLL : (w == 1u) ? selL : selP;
      const uint32_t bm = (w == 0u) ? selL : (w == 1u) ? selP : selR;
      const uint32_t br = (w == 0u) ? selP : (w == 1u) ? selR : selRR;
      const bool skip = ovf || (w == 0u && !hasLL) || (w == 2u && !hasRR);
      if (!skip) v = triple_llh_grouped(S, s_const, s_blen, bl, c - 1u, bm, c, br, c + 1u, A);
    }
    // the triples left and right of the site, from the neighbouring lanes
    const double vl = shfl_f64(v, lane > 0 ? lane - 1 : 0), vr = shfl_f64(v, lane < 63 ? lane + 1 : 63);
    bool accepted = false, overflowed = false;
    if (have && w == 1u) {
      const double llh_l_orig = llh_l, llh_r_orig = llh_r;
      if (!ovf) {
        if (hasLL) llh_l = vl;
        llh_m = v;
        if (hasRR) llh_r = vr;
      }
      llr += (llh_l + llh_r - llh_l_orig - llh_r_orig);
      const double u = epv_keyed_block(seed_lo, seed_hi, (uint32_t)(S.g0 + site), sweep, 0u, 0u, 0u, 0u).d0;
      bool acc = (llr >= 0.0) || (u < epv_exp(llr));
      if (ovf) { acc = false; overflowed = true; }
      if (acc) {
        S.sel[site] = (uint8_t)(selM ^ 1u);
        S.tri[site - 1] = llh_l;
        S.tri[site] = llh_m;
        S.tri[site + 1] = llh_r;
        accepted = site >= own_first && site <= own_last;   // redundant updates of halo columns are not counted
      }
    }
    const unsigned long long am = __ballot(accepted), om = __ballot(overflowed);
    if (lane == 0) {
      const uint32_t shard = (blockIdx.x + blockIdx.y) & (EPV_SHARDS - 1u);
      if (am) atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_ACCEPT, shard)], (unsigned long long)__popcll(am));
      if (om) atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_OVERFLOW, shard)], (unsigned long long)__popcll(om));
    }
  }
  // the task lists of this phase have been consumed (stream order): fold their lengths into the running
  // total and clear them for the next propose kernel, as epv_mh_accept_kernel does
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < EPV_SHARDS) {
    const unsigned long long packed = counters[EPV_CNT_IDX(EPV_CNT_TASKS, threadIdx.x)];
    const unsigned long long packed2 = counters[EPV_CNT_IDX(EPV_CNT_TASKS2, threadIdx.x)];
    counters[EPV_CNT_IDX(EPV_CNT_COOP, threadIdx.x)] +=
        (packed & 0xffffffffull) + (packed >> 32) + (packed2 & 0xffffffffull) + (packed2 >> 32);
    counters[EPV_CNT_IDX(EPV_CNT_TASKS, threadIdx.x)] = 0ull;
    counters[EPV_CNT_IDX(EPV_CNT_TASKS2, threadIdx.x)] = 0ull;
    if (list_mode) counters[EPV_CNT_IDX(list_mode == 2u ? EPV_CNT_ALIST0 : EPV_CNT_ALIST1, threadIdx.x)] = 0ull;
    counters[EPV_CNT_IDX(EPV_CNT_COOP, threadIdx.x)] += counters[EPV_CNT_IDX(EPV_CNT_SEG, threadIdx.x)] >> 32;
    counters[EPV_CNT_IDX(EPV_CNT_SEG, threadIdx.x)] = 0ull;
  }
}

#endif
