#ifndef EPV_FORWARD_H
#define EPV_FORWARD_H
// epv_forward.h -- forward simulation of epigenome evolution along the tree, site-parallel
// (included by epv_kernels.h).  What epievo_sim computes (/root/reference/src/prog/epievo_sim.cpp:
// 102-152, 329-352 over TripletSampler, src/libepievo/TripletSampler.cpp:165-184) is ONE sequential
// chain of events per branch: total rate sum_c count_c rate_c, a context drawn in proportion to
// count_c rate_c, a uniform position of that context.  The same law, written so that sites can run
// in parallel:
//   * THINNING.  Every interior site carries its own Poisson stream of CANDIDATE events at the rate
//     lam_max = max_c rate_c (gaps -log(1 - u)/lam_max); a candidate at time t flips the site with
//     probability rate[context of the site at t-]/lam_max.  The superposition of the n streams,
//     thinned that way, is the reference's process exactly.
//   * KEYED RANDOMNESS.  Candidate k of (node, site) reads one Philox block addressed by
//     (seed, site, node, k): d0 -> its gap, d1 -> its acceptance uniform.  The outcome is a
//     function of the seed, whatever order the candidates are resolved in.
//   * LOCAL MINIMA.  A candidate depends only on the earlier candidates of its site and its two
//     neighbours.  A site whose next candidate is earlier than both neighbours' next candidates
//     sees the true context (everything before t is resolved around it) and can decide now; two
//     neighbours are never both local minima, so a round resolves about a third of the fronts at
//     once.  A block runs many rounds on a tile in LDS; the tile's halo sites are re-simulated
//     redundantly from the launch's snapshot (double-buffered front state), a site next to the
//     tile edge simply waits, so whatever IS resolved is right and launch shape never shows.
// Accepted flips are appended straight to the device path storage (buffer 0: meta count, jump
// planes), the layout the sampler's kernels read: a simulated genome is resident for MCMC, or
// downloaded through epv_download_paths.  The CPU oracle (oracle/epv_oracle.c,
// orc_forward_thinning) processes the same candidates in global time order; tests demand equal bits.
// Sites 0 and n-1 never change (TripletSampler.cpp:37-70 buckets interior positions only).

#define EPV_FWD_SWEEP 0xfffffffeu        /* Philox "sweep" word of the candidate streams */
#define EPV_FWD_ROOT_SWEEP 0xfffffffdu   /* ... of the root sequence's uniforms */

struct EpvFwd {
  uint8_t *st[2];     // [n] state of every site at its front (double-buffered between launches)
  uint32_t *k[2];     // [n] index of the site's next candidate
  double *t[2];       // [n] its time (+inf: none left on this branch)
  uint8_t *end;       // [N][n] end state of every node (node 0: the root sequence)
  double pacc[8];     // rate_c / lam_max
  double lam_max;
};

// ---- root sequence (EpiEvoModel::sample_state_sequence, EpiEvoModel.cpp:281-298): a first-order
// chain along the sites, state_i = f_i(state_{i-1}) with f_i(prev) = (u_i <= T[prev][prev]) ? prev : !prev.
// Maps of a two-point set compose associatively: two bits per map (f(0) | f(1) << 1), a block
// composes 4096 sites, one block scans the block aggregates, the blocks replay from their prefix.
#define EPV_ROOT_PER_THREAD 16u
__device__ __forceinline__ uint32_t epv_map_after(uint32_t g, uint32_t f) {   // g after f
  return ((g >> (f & 1u)) & 1u) | (((g >> ((f >> 1) & 1u)) & 1u) << 1);
}
__device__ __forceinline__ uint32_t epv_root_map(uint32_t seed_lo, uint32_t seed_hi, uint64_t gsite, double T00,
                                                 double T11, double pi1) {
  const double u = epv_keyed_block(seed_lo, seed_hi, (uint32_t)gsite, EPV_FWD_ROOT_SWEEP, 0u, 0u, 0u, 0u).d0;
  if (gsite == 0) { const uint32_t s = u < pi1 ? 1u : 0u; return s | (s << 1); }   // site 0: a constant
  const uint32_t f0 = (u <= T00) ? 0u : 1u, f1 = (u <= T11) ? 1u : 0u;
  return f0 | (f1 << 1);
}
// pass 0: agg[block] = composition of the block's sites.  pass 1: states, from prefix[block] (the
// state left of the block; blocks' prefixes come from epv_fwd_root_scan_kernel)
__global__ __launch_bounds__(256) void epv_fwd_root_kernel(uint64_t n, uint64_t g0, uint32_t seed_lo, uint32_t seed_hi,
                                                           double T00, double T11, double pi1, uint32_t pass,
                                                           uint8_t *agg, const uint8_t *prefix, uint8_t *out) {
  __shared__ uint8_t s_map[256];
  const uint64_t base = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * EPV_ROOT_PER_THREAD;
  uint32_t maps = 0u, comp = 2u;   // identity = f(0)=0, f(1)=1 -> bits 0b10
#pragma unroll
  for (uint32_t q = 0; q < EPV_ROOT_PER_THREAD; ++q) {
    const uint64_t s = base + q;
    const uint32_t f = s < n ? epv_root_map(seed_lo, seed_hi, g0 + s, T00, T11, pi1) : 2u;
    maps |= f << (2u * q);
    comp = epv_map_after(f, comp);
  }
  s_map[threadIdx.x] = (uint8_t)comp;
  __syncthreads();
  if (pass == 0u) {
    if (threadIdx.x == 0) {
      uint32_t a = 2u;
      for (uint32_t i = 0; i < 256u; ++i) a = epv_map_after(s_map[i], a);
      agg[blockIdx.x] = (uint8_t)a;
    }
    return;
  }
  // state entering this thread's run: the block's prefix through the earlier threads' maps
  uint32_t st = prefix[blockIdx.x];
  for (uint32_t i = 0; i < threadIdx.x; ++i) st = (s_map[i] >> st) & 1u;
#pragma unroll
  for (uint32_t q = 0; q < EPV_ROOT_PER_THREAD; ++q) {
    const uint64_t s = base + q;
    st = ((maps >> (2u * q)) >> st) & 1u;
    if (s < n) out[s] = (uint8_t)st;
  }
}
__global__ void epv_fwd_root_scan_kernel(const uint8_t *agg, uint64_t n_blocks, uint8_t *prefix) {
  if (threadIdx.x || blockIdx.x) return;
  uint32_t st = 0u;    // site 0's map is a constant: the value entering block 0 does not matter
  for (uint64_t b = 0; b < n_blocks; ++b) { prefix[b] = (uint8_t)st; st = (agg[b] >> st) & 1u; }
}

// ---- start of a branch: every site takes the state its parent node ended in; first candidate
__global__ __launch_bounds__(256) void epv_fwd_begin_kernel(EpvDev S, EpvFwd F, uint32_t node, uint32_t parent,
                                                            uint32_t seed_lo, uint32_t seed_hi, uint32_t p) {
  const uint64_t site = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= S.n) return;
  const uint32_t st = F.end[(uint64_t)parent * S.n + site];
  S.meta[meta_idx(S, 0u, node - 1u, site)] = (epv_meta_t)(st << EPV_INIT_SHIFT);
  if (node == 1u) { S.sel[site] = 0; S.tri[site] = 0.0; }
  F.st[p][site] = (uint8_t)st;
  F.k[p][site] = 0u;
  const uint64_t g = S.g0 + site;
  double t = EPV_INF;
  if (g >= 1 && g + 1 < S.n_global)
    t = -epv_log(1.0 - epv_keyed_block(seed_lo, seed_hi, (uint32_t)g, EPV_FWD_SWEEP, node, 0u, 0u, 0u).d0) / F.lam_max;
  F.t[p][site] = t;
}

// ---- up to `rounds` rounds of local-minimum resolution on tiles of blockDim sites, of which the
// inner blockDim - 2 halo are owned (written back); reads front buffer p, writes 1 - p.
// info[0] += owned sites with a candidate left, info[1] += capacity overflows, info[2] += flips of owned sites
#define EPV_FWD_THREADS 512
__global__ __launch_bounds__(EPV_FWD_THREADS) void epv_fwd_rounds_kernel(EpvDev S, EpvFwd F, uint32_t node, double T,
                                                                          uint32_t seed_lo, uint32_t seed_hi, uint32_t p,
                                                                          uint32_t halo, uint32_t rounds,
                                                                          unsigned long long *info) {
  __shared__ double s_t[EPV_FWD_THREADS];
  __shared__ uint8_t s_st[EPV_FWD_THREADS];
  const uint32_t i = threadIdx.x, W = blockDim.x, own_w = W - 2u * halo;
  const int64_t site_s = (int64_t)blockIdx.x * own_w - (int64_t)halo + (int64_t)i;
  const bool inside = site_s >= 0 && (uint64_t)site_s < S.n;
  const uint64_t site = inside ? (uint64_t)site_s : 0u;
  const bool owned = inside && i >= halo && i < W - halo;
  const uint32_t gsite = (uint32_t)(S.g0 + site);
  double t = EPV_INF;
  uint32_t st = 0u, k = 0u, cnt = 0u;
  if (inside) { t = F.t[p][site]; st = F.st[p][site]; k = F.k[p][site]; }
  const uint32_t b = node - 1u;
  if (owned) cnt = S.meta[meta_idx(S, 0u, b, site)] & EPV_NJ_MASK;
  double *jp = S.jumps + jump_idx(S, 0u, b, site);
  s_t[i] = t;
  s_st[i] = (uint8_t)st;
  bool ovf = false;
  uint32_t flips = 0u;
  const bool inner = i > 0u && i + 1u < W;      // the tile's two edge sites only wait
  for (uint32_t r = 0; r < rounds; ++r) {
    __syncthreads();
    bool can = false;
    uint32_t ctx = 0u;
    if (inner && t < T) {
      const double tl = s_t[i - 1u], tr = s_t[i + 1u];
      // (time, site) in lexicographic order: equal times (probability ~0) resolve left to right
      can = (t < tl || (t == tl && false)) && (t < tr || t == tr);
      ctx = 4u * s_st[i - 1u] + 2u * st + s_st[i + 1u];
    }
    if (!__syncthreads_or(can ? 1 : 0)) break;
    if (can) {
      const epv_block2 me = epv_keyed_block(seed_lo, seed_hi, gsite, EPV_FWD_SWEEP, node, 0u, k, 0u);
      if (me.d1 < F.pacc[ctx]) {
        st ^= 1u;
        if (owned) {
          if (cnt < S.C) jp[(uint64_t)cnt * S.n] = t; else ovf = true;
          ++cnt;
          ++flips;
        }
      }
      ++k;
      t += -epv_log(1.0 - epv_keyed_block(seed_lo, seed_hi, gsite, EPV_FWD_SWEEP, node, 0u, k, 0u).d0) / F.lam_max;
      s_t[i] = t;
      s_st[i] = (uint8_t)st;
    }
  }
  if (owned) {
    F.t[p ^ 1u][site] = t;
    F.st[p ^ 1u][site] = (uint8_t)st;
    F.k[p ^ 1u][site] = k;
    F.end[(uint64_t)node * S.n + site] = (uint8_t)st;
    const uint32_t init = S.meta[meta_idx(S, 0u, b, site)] >> EPV_INIT_SHIFT;
    S.meta[meta_idx(S, 0u, b, site)] = (epv_meta_t)((init << EPV_INIT_SHIFT) | (cnt > S.C ? S.C : cnt));
  }
  const int pend = __syncthreads_count(owned && t < T ? 1 : 0);
  const int novf = __syncthreads_count(ovf ? 1 : 0);
  if (threadIdx.x == 0) {
    if (pend) atomicAdd(&info[0], (unsigned long long)pend);
    if (novf) atomicAdd(&info[1], (unsigned long long)novf);
  }
  // one atomic per wave for the flips (the total of the histories' jumps: no pass over them afterwards)
  const uint32_t wsum = epv_bcast(wave_incl_scan_u32(flips), 63);
  if (epv_lane() == 0 && wsum) atomicAdd(&info[2], (unsigned long long)wsum);
}

#endif
