#ifndef EPV_KERNELS_H
#define EPV_KERNELS_H
// epv_kernels.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for epievo's
// MCEM inner loop.  One lane = one genomic site; one launch = one colour phase of the
// 3-colour sweep (sites congruent mod 3 are conditionally independent and
// write-disjoint).  No MFMA: this is an fp64 latency/divergence-bound
// sampler, not a contraction.
//
// What each kernel replaces in the reference (/root/reference/src/libepievo):
//   epv_mh_propose_kernel + epv_mh_accept_kernel
//                          SingleSiteSampler::Metropolis_Hastings_site (SingleSiteSampler.cpp:482-536)
//                          = collect_segment_info (Segment.cpp:35-79) + pruning (:145-157)
//                          + downward_sampling (:227-255) + end_cond_sample_forward_rejection
//                          (EndCondSampling.cpp:479-509) + log_accept_rate (:396-433)
//   epv_reset_kernel       SingleSiteSampler::reset (:449-475)
//   epv_suffstat_kernel    get_sufficient_statistics (ParamEstimation.cpp:92-114)
//   epv_tree_reduce_kernel (the upper levels of the canonical binary-tree reduction)
//   epv_scale_kernel       scale_jump_times (ParamEstimation.cpp:369-380)
//
// Arithmetic contract: every fp64 operation follows the order of the reference
// expression it restates; the file is compiled -ffp-contract=off; exp/log are the
// deterministic epv_exp/epv_log.  The CPU oracle (oracle/epv_oracle.c, rung B) makes
// the same choices independently, and tests/test_gpu_parity.py demands bit equality.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "epv_device.h"
#include "epv_math.h"
#include "epv_philox.h"

#define EPV_INF __builtin_inf()

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int epv_lane() { return (int)(threadIdx.x & 63); }

// the value of lane `src` in every lane, src wave-uniform: one v_readlane_b32 (the result lives in a
// scalar register) instead of a ds_bpermute with a per-lane index
__device__ __forceinline__ uint32_t epv_bcast(uint32_t v, int src) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
}
// inclusive prefix sum over the 64 lanes (all active).  DPP: shifts by 1, 2, 4, 8 inside the
// 16-lane rows (lanes shifted in from outside a row read 0), then lane 15 of rows 0 and 2 into rows
// 1 and 3 (row_bcast:15, row mask 0b1010) and lane 31 into rows 2 and 3 (row_bcast:31, 0b1100):
// six v_add_u32_dpp instead of six ds_bpermute with index arithmetic and predicated adds
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
#ifndef EPV_SCAN_BPERMUTE
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
  return v;
#else
  const int lane = epv_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
#endif
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_xor(v, d);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
  const uint64_t u = epv_d2u(v);
  const uint32_t lo = __shfl((uint32_t)u, src), hi = __shfl((uint32_t)(u >> 32), src);
  return epv_u2d(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  const uint64_t u = epv_d2u(v);
  const uint32_t lo = __shfl_xor((uint32_t)u, m), hi = __shfl_xor((uint32_t)(u >> 32), m);
  return epv_u2d(((uint64_t)hi << 32) | lo);
}

// Sum 16 per-lane values over the wave in the canonical balanced-tree order with a
// transpose-reduce: at stage s (xor 1,2,4,8) a lane keeps half of its values and
// trades the other half with its partner, so the 16 butterflies of 6 stages each (96
// shuffles + 96 adds) become 8+4+2+1 exchanges plus two full-width stages.  Every
// individual sum still pairs lanes (i, i^1), then (i, i^2), ... exactly as the plain
// xor butterfly does, so the result is bit-identical to it.  On return lane l holds
// the wave total of value index idx(l) = 8*bit0 + 4*bit1 + 2*bit2 + bit3 of l.
__device__ __forceinline__ double wave_tree_sum16(const double v[16], int lane, int &idx) {
  double w[8], x[4], y[2], z;
  const bool h0 = lane & 1, h1 = lane & 2, h2 = lane & 4, h3 = lane & 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = (h0 ? v[i + 8] : v[i]) + shfl_xor_f64(h0 ? v[i] : v[i + 8], 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = (h1 ? w[i + 4] : w[i]) + shfl_xor_f64(h1 ? w[i] : w[i + 4], 2);
#pragma unroll
  for (int i = 0; i < 2; ++i) y[i] = (h2 ? x[i + 2] : x[i]) + shfl_xor_f64(h2 ? x[i] : x[i + 2], 4);
  z = (h3 ? y[1] : y[0]) + shfl_xor_f64(h3 ? y[0] : y[1], 8);
  z = z + shfl_xor_f64(z, 16);
  z = z + shfl_xor_f64(z, 32);
  idx = (h0 ? 8 : 0) + (h1 ? 4 : 0) + (h2 ? 2 : 0) + (h3 ? 1 : 0);
  return z;
}

// ------------------------------------------------------------------ path access
// meta layout, see epv_device.h (branch-major by default)
// (buf, b, site) -> element; buf is 0 or 1 and differs from lane to lane, so the buffer offset is a
// select rather than a multiply (64-bit and 32-bit integer multiplies are quarter-rate on the
// vector unit; b is wave-uniform at nearly every call site, where b * n stays on the scalar unit)
__device__ __forceinline__ uint64_t meta_idx(const EpvDev &S, uint32_t buf, uint32_t b, uint64_t site) {
  return (buf ? (uint64_t)S.B * S.n : 0ull) + (uint64_t)b * S.n + site;
}
// first jump slot of (buf, b, site) in S.jumps
__device__ __forceinline__ uint64_t jump_idx(const EpvDev &S, uint32_t buf, uint32_t b, uint64_t site) {
  const uint64_t Cn = (uint64_t)S.C * S.n;
  return (buf ? (uint64_t)S.B * Cn : 0ull) + (uint64_t)b * Cn + site;
}
struct PathRef {
  const double *j;  // jump k lives at j[k * n]
  uint32_t nj;
  uint32_t init;
};
__device__ __forceinline__ PathRef path_ref(const EpvDev &S, uint32_t buf, uint32_t b,
                                            uint64_t site) {
  const epv_meta_t m = S.meta[meta_idx(S, buf, b, site)];
  PathRef p;
  p.j = S.jumps + jump_idx(S, buf, b, site);
  p.nj = m & EPV_NJ_MASK;
  p.init = m >> EPV_INIT_SHIFT;
  return p;
}

// ------------------------------------------------ sufficient statistics of a triple
struct Acc8 {
  double d[8];
  uint32_t j[8];
};
__device__ __forceinline__ void acc_clear(Acc8 &A) {
#pragma unroll
  for (int c = 0; c < 8; ++c) { A.d[c] = 0.0; A.j[c] = 0u; }
}
// D[ctx] += dt (and J[ctx] += 1 for a middle-site jump) without a runtime-indexed
// register array: adding +0.0 to the seven other sums leaves them bit-unchanged.
__device__ __forceinline__ void acc_add(Acc8 &A, int ctx, double dt, bool mid) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bool h = (ctx == c);
    A.d[c] += h ? dt : 0.0;
    A.j[c] += (h && mid) ? 1u : 0u;
  }
}

// Path.cpp:206-301 as one 3-way merge with +inf sentinels; tie rules: left only if
// strictly below min(mid,right), else mid only if strictly below right, else right.
// Same accumulator kept in LDS (one column per thread, element c at [c * stride]): a
// runtime-indexed read-modify-write costs ~6 instructions instead of the ~40 of the
// register select chain; used where a kernel is bound by instruction issue.
struct AccLds {
  double *d;
  uint32_t *j;
  uint32_t stride;
};
__device__ __forceinline__ void acc_clear(AccLds &A) {
#pragma unroll
  for (int c = 0; c < 8; ++c) { A.d[c * A.stride] = 0.0; A.j[c * A.stride] = 0u; }
}
__device__ __forceinline__ void acc_add(AccLds &A, int ctx, double dt, bool mid) {
  A.d[ctx * A.stride] += dt;
  if (mid) A.j[ctx * A.stride] += 1u;
}
__device__ __forceinline__ double acc_d(const Acc8 &A, int c) { return A.d[c]; }
__device__ __forceinline__ uint32_t acc_j(const Acc8 &A, int c) { return A.j[c]; }
__device__ __forceinline__ double acc_d(const AccLds &A, int c) { return A.d[c * A.stride]; }
__device__ __forceinline__ uint32_t acc_j(const AccLds &A, int c) { return A.j[c * A.stride]; }

template <class ACC>
__device__ __forceinline__ void merge3(const PathRef &L, const PathRef &M, const PathRef &R,
                                       uint64_t n, double tot_time, ACC &A) {
  int ctx = (int)(4u * L.init + 2u * M.init + R.init);
  double prev = 0.0;
  uint32_t i = 0, j = 0, k = 0;
  double tl = L.nj ? L.j[0] : EPV_INF;
  double tm = M.nj ? M.j[0] : EPV_INF;
  double tr = R.nj ? R.j[0] : EPV_INF;
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

// path_log_likelihood (SingleSiteSampler.cpp:374-391): un-logged root prior (reference
// quirk) + sum_c J_c log(rate_c) - D_c rate_c over all branches of one triple.
// (bl,sl),(bm,sm),(br,sr) = (buffer, site) of the left / middle / right column.
template <class ACC>
__device__ __forceinline__ double triple_llh(const EpvDev &S, const double *s_model,
                                             const double *s_blen, uint32_t bl, uint64_t sl,
                                             uint32_t bm, uint64_t sm, uint32_t br,
                                             uint64_t sr, ACC &A) {
  acc_clear(A);
  uint32_t rl = 0, rm = 0, rr = 0;
  for (uint32_t b = 0; b < S.B; ++b) {
    const PathRef L = path_ref(S, bl, b, sl), M = path_ref(S, bm, b, sm),
                  R = path_ref(S, br, b, sr);
    if (b == 0) { rl = L.init; rm = M.init; rr = R.init; }
    merge3(L, M, R, S.n, s_blen[b + 1], A);
  }
  const double *rates = s_model, *lrates = s_model + 8, *T = s_model + 16;
  double llh = T[2 * rl + rm] * T[2 * rm + rr];
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += (double)acc_j(A, c) * lrates[c] - acc_d(A, c) * rates[c];
  llh += s;
  return llh;
}

// the same with the meta words of the three columns already in LDS (mc[(col * B + b) * stride],
// col = the column's index in the caller's cache): the accept kernel fetches the 5 B words a
// site needs in ONE batch of independent loads instead of paying a dependent global round trip
// per branch and triple (it is bound by memory latency: SQ_WAIT_ANY was 70 % of its wave cycles)
template <class ACC>
__device__ __forceinline__ double triple_llh_cached(const EpvDev &S, const double *s_model,
                                                    const double *s_blen, const epv_meta_t *mc, uint32_t stride,
                                                    uint32_t cl, uint32_t bl, uint64_t sl, uint32_t cm, uint32_t bm,
                                                    uint64_t sm, uint32_t cr, uint32_t br, uint64_t sr, ACC &A) {
  acc_clear(A);
  uint32_t rl = 0, rm = 0, rr = 0;
  const uint32_t B = S.B;
  for (uint32_t b = 0; b < B; ++b) {
    const uint32_t ml = mc[(cl * B + b) * stride], mm = mc[(cm * B + b) * stride], mr = mc[(cr * B + b) * stride];
    PathRef L, M, R;
    L.j = S.jumps + jump_idx(S, bl, b, sl); L.nj = ml & EPV_NJ_MASK; L.init = ml >> EPV_INIT_SHIFT;
    M.j = S.jumps + jump_idx(S, bm, b, sm); M.nj = mm & EPV_NJ_MASK; M.init = mm >> EPV_INIT_SHIFT;
    R.j = S.jumps + jump_idx(S, br, b, sr); R.nj = mr & EPV_NJ_MASK; R.init = mr >> EPV_INIT_SHIFT;
    if (b == 0) { rl = L.init; rm = M.init; rr = R.init; }
    merge3(L, M, R, S.n, s_blen[b + 1], A);
  }
  const double *rates = s_model, *lrates = s_model + 8, *T = s_model + 16;
  double llh = T[2 * rl + rm] * T[2 * rm + rr];
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += (double)acc_j(A, c) * lrates[c] - acc_d(A, c) * rates[c];
  llh += s;
  return llh;
}

// ------------------------------------------------ forward rejection trial
enum { TRIAL_FAIL = 0, TRIAL_OK = 1, TRIAL_OVERFLOW = 2 };

// One trial t of the end-conditioned sampler of segment (node,k), start state a0, target `end`:
//   a0 == end  forward_sampling (EndCondSampling.cpp:466-476): hold times ~ Exp(rate of the
//              current state) = -log(1-u)/rate until T is passed;
//   a0 != end  end_cond_sampling_Nielsen (EndCondSampling.cpp:583-617): the first jump from the
//              truncated exponential -log(1 - u0 (1 - exp(-rate_a T)))/rate_a (`trunc` = the
//              bracket, the same for every trial of the segment), then forward sampling from
//              the other state.  Same conditional law as the reference's hot-path forward
//              rejection, but the acceptance probability does not vanish with T: a flip on a
//              short branch costs ~1 trial instead of ~1/P(a->b) = 1e2..1e5.
// `u0` is the trial's first draw (its Philox block is shared, see epv_philox.h); later
// draws come from the trial's own blocks.  `room` = jump slots left in this path.
// Jump times (+start_time) are written to dst[0], dst[stride], ... for the first
// `max_store` jumps only (0 = store nothing); the count is always complete.
//
// Shortcut (execution only, results unchanged): most trials on short branches end at the
// first hold time because no (further) jump falls inside the segment, i.e.
// 1-u <= exp(-rate*T).  `nojump0/1` = that bound for state 0/1 from a float exp, shrunk by
// 1e-4 -- far more than the float error (~1e-5 for rate*T < 40) and astronomically more than
// the fp64 rounding of the exact test -- so "1-u < nojump" PROVES the exact computation
// -log(1-u)/rate >= T (>= the time left) without evaluating log or the division.
// Everything else takes the exact path below, so the outcome is always the oracle's.
__device__ __forceinline__ int run_trial(uint32_t seed_lo, uint32_t seed_hi, uint32_t gsite,
                                         uint32_t sweep, uint32_t node, uint32_t k, uint32_t t,
                                         double u0, uint32_t a0, uint32_t end, double T, double r0,
                                         double r1, double nojump0, double nojump1, double trunc,
                                         uint32_t room, double *dst, uint64_t stride, uint32_t max_store,
                                         double start_time, uint32_t &nj_out, bool nielsen = true) {
  nj_out = 0;
  uint32_t nj = 0, a = a0, d = 0;
  double tau = 0.0;
  epv_block2 blk;
  blk.d0 = 0.0; blk.d1 = 0.0;
  double u = u0;
  if (nielsen && a0 != end) {
    tau = -epv_log(1.0 - u0 * trunc) / (a0 ? r1 : r0);
    if (!(tau < T)) return TRIAL_FAIL;   // a draw within rounding of 1: redraw (oracle: same guard)
    if (room == 0u) return TRIAL_OVERFLOW;
    a ^= 1u;
    if (max_store > 0u) dst[0] = tau + start_time;
    nj = 1u;
    blk = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, k, t, 0u);
    u = blk.d0;
    d = 1u;
    if (1.0 - u < (a ? nojump1 : nojump0)) { nj_out = 1u; return TRIAL_OK; }   // a == end now
  } else if (1.0 - u0 < (a0 ? nojump1 : nojump0)) {
    return a0 == end ? TRIAL_OK : TRIAL_FAIL;   // no jump: the state is kept (a flip only in forward mode)
  }
  int outcome;
  for (;;) {
    tau += -epv_log(1.0 - u) / (a ? r1 : r0);
    if (!(tau < T)) { outcome = (a == end) ? TRIAL_OK : TRIAL_FAIL; break; }
    if (nj >= room) { outcome = TRIAL_OVERFLOW; break; }
    a ^= 1u;
    if (nj < max_store) dst[(uint64_t)nj * stride] = tau + start_time;
    ++nj;
    if ((d & 1u) == 0u) blk = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, k, t, d >> 1);
    u = (d & 1u) ? blk.d1 : blk.d0;
    ++d;
  }
  nj_out = nj;
  return outcome;
}

// exp(-x) bound for the shortcut above: float exp, shrunk; 0 (never taken) when x is
// large enough that float accuracy is not guaranteed
__device__ __forceinline__ double nojump_bound(double x) {
  return x < 40.0 ? (double)__expf(-(float)x) * 0.9999 : 0.0;
}

// First draw of trial t of segment (node,k) -- see the address table in epv_philox.h
__device__ __forceinline__ double first_draw(uint32_t seed_lo, uint32_t seed_hi, uint32_t gsite,
                                             uint32_t sweep, uint32_t node, uint32_t k, uint32_t t) {
  if (t == 1u) return epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, k, 0u, 0u).d1;
  const epv_block2 fb = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, k, t >> 1,
                                        EPV_FIRST_DRAW_BLOCK);
  return (t & 1u) ? fb.d1 : fb.d0;
}

// Evaluate trials t0 .. t0+W-1 of one segment IN ORDER and return the first one that does
// not fail (its index in t_out), or TRIAL_FAIL when all W fail.  For a segment that keeps
// its state the scan is two-level to keep the lanes of a wave together: a cheap pass
// classifies each trial by its first draw alone (no jump inside the segment: success --
// exact thanks to run_trial's guard band), and only a trial that does jump is evaluated
// exactly.  On a short branch ~95 % of trials never leave the scan.  A segment that changes
// state (Nielsen) evaluates each trial exactly; nearly every one succeeds.
__device__ __forceinline__ int scan_trials(uint32_t seed_lo, uint32_t seed_hi, uint32_t gsite,
                                           uint32_t sweep, uint32_t node, uint32_t k, uint32_t t0,
                                           uint32_t W, uint32_t a0, uint32_t end, double T, double r0,
                                           double r1, double trunc, uint32_t room, double *dst,
                                           uint64_t stride, uint32_t max_store, double start_time,
                                           uint32_t &t_out, uint32_t &nj_out, bool nielsen = true,
                                           uint32_t *maxm = nullptr) {
  // maxm (optional): running maximum of the jumps made by the trials evaluated here, failed ones
  // included (a trial classified "no jump" makes none) -- what decides a capacity overflow when
  // the room left in the path is only known later (epv_jumps2.h)
  // flip: the segment changes state AND is sampled by Nielsen's method (forward-rejection mode,
  // EPV_FLAG_FORWARD_REJECTION, treats such a segment like any other: a trial must END in `end`)
  const bool flip = nielsen && a0 != end;
  const bool keep = a0 == end;
  // no-(further-)jump bound of the state the chain waits in: a0, or `end` after Nielsen's first jump
  const double bound = nojump_bound(T * ((flip ? end : a0) ? r1 : r0));
  uint32_t t = t0;
  const uint32_t t_end = t0 + W;
  nj_out = 0;
  epv_block2 fb;             // first-draw block of the trial pair (2m, 2m+1), reused for 2m+1
  fb.d0 = fb.d1 = 0.0;
  uint32_t fb_pair = 0xffffffffu;
  for (;;) {
    double u = 0.0;
    bool cand = false;
    for (; t < t_end; ++t) {
      if (t == 1u) {
        u = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, k, 0u, 0u).d1;
      } else {
        if ((t >> 1) != fb_pair) {
          fb_pair = t >> 1;
          fb = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, k, fb_pair, EPV_FIRST_DRAW_BLOCK);
        }
        u = (t & 1u) ? fb.d1 : fb.d0;
      }
      if (!flip && 1.0 - u < bound) {     // provably no jump in this trial: the state is kept
        if (keep) { t_out = t; nj_out = 0; return TRIAL_OK; }
        continue;                         // forward mode, state must change: this trial fails
      }
      cand = true;
      break;
    }
    if (!cand) return TRIAL_FAIL;
    const double nb = flip ? bound : 0.0;
    const int oc = run_trial(seed_lo, seed_hi, gsite, sweep, node, k, t, u, a0, end, T, r0, r1, nb, nb,
                             trunc, room, dst, stride, max_store, start_time, nj_out, nielsen);
    if (maxm && nj_out > *maxm) *maxm = nj_out;
    if (oc != TRIAL_FAIL) { t_out = t; return oc; }
    ++t;
  }
}

// ------------------------------------------------ LDS staging of constants
// layout (doubles): [0..19] model (rates, log_rates, T), [20 .. 20+N) branch lengths
__device__ __forceinline__ void stage_constants(const EpvDev &S, double *s_const) {
  const double *m = reinterpret_cast<const double *>(S.model);
  for (uint32_t i = threadIdx.x; i < 20u + S.N; i += blockDim.x)
    s_const[i] = (i < 20u) ? m[i] : S.blen[i - 20u];
  __syncthreads();
}

// 2-state CTMC transition probability from the shared h = exp(-t (r0+r1))
// (TwoStateCTMarkovModel::get_trans_prob, ContinuousTimeMarkovModel.cpp:116-125)
__device__ __forceinline__ double gtp(double r0, double r1, double h, double denom, uint32_t a,
                                      uint32_t b) {
  const double prob = (a ? r0 + r1 * h : r0 * h + r1) / denom;
  return (a == b) ? prob : 1.0 - prob;
}

// =========================================================================
//  Metropolis-Hastings colour phase, part 1: the proposal's segment end states
// =========================================================================
// One lane per site.  Per-wave LDS: node table regA[N][64] u32 (record offset of the
// branch above the node | proposal end state << 31) followed by a pool of
// `pool_entries` 16-byte records {p0, p1} (Felsenstein partials at the top of each
// segment) handed out to the lanes by a wave prefix sum; a branch with K segments uses
// K+1 records, the extra one holding the node's q.  Segment lengths and contexts are
// NOT stored: both passes re-derive them by merging the neighbours' jump planes on the
// fly (backward in the pruning pass, forward in the sampling pass), which keeps the
// pool small enough for ~12 waves per CU.  When the lanes of a wave together need more
// records than the pool has, the wave runs the update in several rounds over a prefix of
// its lanes.
#define EPV_MH_THREADS 256

// GPOOL = false: the record pool lives in LDS (short trees: ~13 KB per wave, 12 waves/CU).
// GPOOL = true : the pool is a per-block slab in global memory with the very same indexing.
//   On a large tree the pool alone (64 lanes x (N-1) branches x ~2 records x 16 B = 84 KB for
//   the 16-leaf tree) would leave ONE wave per CU; with the slab in HBM/L2 only the node table
//   stays in LDS and the kernel keeps its 3 waves/SIMD.
#ifndef EPV_PROPOSE_WAVES
#define EPV_PROPOSE_WAVES 3   /* waves per SIMD the register allocation aims for (<= 168 VGPRs) */
#endif
// REFQ = true : q(old)/q(new) evaluated as the reference does (downward_sampling_branch and
//   proposal_prob_branch, SingleSiteSampler.cpp:180-255, :272-339: per segment
//   log P(end | start, data) - log PT(start -> end), for the proposal and for the current path).
// REFQ = false: that ratio is 1 EXACTLY when the root state is not resampled (SAMPLE_ROOT is
//   hard-wired false, :441): per segment the term is log(p[k+1][end] / p[k][start]), which
//   telescopes along a branch and over the tree to minus the log of the proposal's normalising
//   constant prod_{c child of root} p_c[0][root state] -- a function of the neighbours and the
//   leaf data, not of the path.  The reference's own sums differ from 0 by rounding only
//   (<= 2e-12 over millions of updates, tests/test_proposal_ratio.py), and both modes produce
//   the same paths.  The default; the kernel then needs no log, no current-path walk and no
//   prop_llr hand-over.
template <bool GPOOL, bool REFQ>
__global__ __launch_bounds__(64, EPV_PROPOSE_WAVES) void epv_mh_propose_kernel(
    EpvDev S, uint32_t colour, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep,
    uint64_t first, uint64_t last, uint32_t pool_entries, unsigned long long *counters,
    double *gpool) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  double *s_const = s_mem;                     // 20 + N doubles (padded to even)
  const uint32_t const_dbl = (20u + S.N + 1u) & ~1u;
  const uint32_t wave = threadIdx.x >> 6;
  const int lane = epv_lane();
  const uint32_t regA_dbl = (S.N * 64u + 1u) / 2u;   // N * 64 u32
  const uint32_t wave_dbl = ((regA_dbl + 1u) & ~1u) + (GPOOL ? 0u : pool_entries * 2u);
  uint32_t *regA = reinterpret_cast<uint32_t *>(s_mem + const_dbl + (size_t)wave * wave_dbl);
  double *pool = GPOOL ? gpool + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * pool_entries * 128u
                       : s_mem + const_dbl + (size_t)wave * wave_dbl + ((regA_dbl + 1u) & ~1u);
  stage_constants(S, s_const);
  const double *s_rates = s_const;
  const double *s_blen = s_const + 20;

  // site of this lane: the t-th local site >= first whose GLOBAL index has `colour`
  const uint64_t gfirst = S.g0 + first;
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)(gfirst % 3u)) % 3u);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site = s0 + 3u * tid;
  const bool valid = site <= last;
  const uint64_t n = S.n;
  const uint32_t B = S.B;
  const uint32_t gsite = (uint32_t)(S.g0 + site);

  uint32_t selL = 0, selM = 0, selR = 0;
  uint32_t need = 0;
  if (valid) {
    selL = S.sel[site - 1]; selM = S.sel[site]; selR = S.sel[site + 1];
    for (uint32_t b = 0; b < B; ++b) {
      const uint32_t mL = S.meta[meta_idx(S, selL, b, site - 1)];
      const uint32_t mR = S.meta[meta_idx(S, selR, b, site + 1)];
      need += (mL & EPV_NJ_MASK) + (mR & EPV_NJ_MASK) + 2u;  // K segments + 1 record for q
    }
  }

  bool pending = valid;
  while (__any(pending)) {
    const uint32_t want = pending ? need : 0u;
    const uint32_t incl = wave_incl_scan_u32(want);
    // LDS pool: the lanes' record ranges are packed by the prefix sum.  Global slab: records
    // are INTERLEAVED -- record r of lane l at (r * 64 + l) -- so that a wave-wide access is
    // one contiguous KB instead of 64 cache lines; a lane then needs `need` rows of its own
    constexpr size_t RS = GPOOL ? 128u : 2u;   // doubles between consecutive records of a lane
    const bool run = pending && (GPOOL ? need <= pool_entries : incl <= pool_entries);
    double *my = GPOOL ? pool + (size_t)lane * 2u : pool + (size_t)(incl - want) * 2u;

    // ---- 1. pruning, reverse pre-order (SingleSiteSampler.cpp:116-157) over segments
    //         re-derived backwards from the neighbours' jumps (Segment.cpp:35-79: on a
    //         tie the forward merge takes the right jump first, so backwards the left
    //         one comes first)
    if (run) {
      uint32_t off = need;
      for (uint32_t node = S.N - 1u; node >= 1u; --node) {
        const uint32_t b = node - 1u;
        const PathRef L = path_ref(S, selL, b, site - 1), R = path_ref(S, selR, b, site + 1);
        const uint32_t K = L.nj + R.nj + 1u;
        off -= K + 1u;
        double q0 = 1.0, q1 = 1.0;
        const uint32_t sub = S.subtree[node];
        if (sub == 1u) {
          const uint32_t mM = S.meta[meta_idx(S, selM, b, site)];
          const uint32_t leaf_state = (mM >> EPV_INIT_SHIFT) ^ (mM & 1u);
          q0 = leaf_state ? 0.0 : 1.0;
          q1 = leaf_state ? 1.0 : 0.0;
        } else {
          for (uint32_t ch = 1u; ch < sub; ch += S.subtree[node + ch]) {
            const double *a = my + (size_t)(regA[(node + ch) * 64u + lane] & 0x7fffffffu) * RS;
            q0 *= a[0];   // p.front() of the child's branch
            q1 *= a[1];
          }
        }
        double *recq = my + (size_t)(off + K) * RS;
        recq[0] = q0; recq[1] = q1;
        regA[node * 64u + lane] = off;
        double n0 = q0, n1 = q1;
        uint32_t i = L.nj, j = R.nj;
        uint32_t trip0 = (4u * L.init + R.init) ^ ((i & 1u) << 2) ^ (j & 1u);  // context of the LAST segment
        double seg_end = s_blen[node];
        double tl = i ? L.j[(uint64_t)(i - 1u) * n] : -EPV_INF;
        double tr = j ? R.j[(uint64_t)(j - 1u) * n] : -EPV_INF;
        for (uint32_t kk = K; kk-- > 0u;) {
          const bool take_left = (kk > 0u) && (tl >= tr);
          const double seg_start = (kk == 0u) ? 0.0 : (take_left ? tl : tr);
          const double len = seg_end - seg_start;
          const double r0 = s_rates[trip0], r1 = s_rates[trip0 | 2u];
          // continuous_time_trans_prob_mat (ContinuousTimeMarkovModel.cpp:143-161)
          const double h = 1.0 / epv_exp(len * (r0 + r1));
          const double denom = r0 + r1;
          const double P00 = (r0 * h + r1) / denom;
          const double P01 = 1.0 - P00;
          const double P11 = (r0 + r1 * h) / denom;
          const double P10 = 1.0 - P11;
          const double a = P00 * n0 + P01 * n1;
          const double c = P10 * n0 + P11 * n1;
          double *rec = my + (size_t)(off + kk) * RS;
          rec[0] = a; rec[1] = c;
          n0 = a; n1 = c;
          if (kk > 0u) {
            if (take_left) { trip0 ^= 4u; --i; tl = i ? L.j[(uint64_t)(i - 1u) * n] : -EPV_INF; }
            else { trip0 ^= 1u; --j; tr = j ? R.j[(uint64_t)(j - 1u) * n] : -EPV_INF; }
            seg_end = seg_start;
          }
        }
      }
    }

    // ---- 2. downward sampling of the segment END STATES (:180-255) fused with
    //         proposal_prob of the current path (:272-339).  The jump times inside the
    //         segments do not influence any state or log-probability, so they are NOT
    //         drawn here: a branch whose every segment keeps its state and provably
    //         (run_trial's shortcut) has no jump in trial 1 is "clean" -- its proposal is
    //         the empty jump list -- and every other (site, branch) pair is appended to a
    //         compact task list for epv_mh_jumps_kernel, which runs the exact forward
    //         rejection with dense lanes instead of making 64 lanes wait for 3.
    double log_prob = 0.0, orig_proposal = 0.0;
    unsigned long long dirty = 0ull;   // bit (node-1) & 63, flushed every 64 branches
    unsigned long long multi = 0ull;   // ... of those, the branches with an even bucket bit (K = 2 or K >= 4)
    unsigned long long deep = 0ull;    // ... and those with K >= 3 (second region)
    {
      uint32_t root_state = run ? (uint32_t)(S.meta[meta_idx(S, selM, 0u, site)] >> EPV_INIT_SHIFT) : 0u;
      if (REFQ && run && (S.flags & EPV_FLAG_SAMPLE_ROOT)) {
        // SAMPLE_ROOT (SingleSiteSampler.cpp:167-176, :246-249, :325-329): the root state from its
        // posterior given the neighbours' root states and the data below (q of node 0 = the product
        // of the root's children's p.front); both log-probabilities carry the term
        double q0 = 1.0, q1 = 1.0;
        for (uint32_t ch = 1u; ch < S.subtree[0]; ch += S.subtree[ch]) {
          const double *a = my + (size_t)(regA[ch * 64u + lane] & 0x7fffffffu) * RS;
          q0 *= a[0];
          q1 *= a[1];
        }
        const uint32_t rl = (uint32_t)(S.meta[meta_idx(S, selL, 0u, site - 1)] >> EPV_INIT_SHIFT);
        const uint32_t rr = (uint32_t)(S.meta[meta_idx(S, selR, 0u, site + 1)] >> EPV_INIT_SHIFT);
        const double *T = s_const + 16;
        const double p0 = (T[2u * rl + 0u] * T[0u + rr]) * q0;
        const double p1 = (T[2u * rl + 1u] * T[2u + rr]) * q1;
        const double root_p0 = p0 / (p0 + p1);
        const double u_root = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, 0u, 0u, 0u, 0u).d1;
        const uint32_t cur_root = root_state;
        root_state = (u_root > root_p0) ? 1u : 0u;
        log_prob = root_state ? epv_log(1.0 - root_p0) : epv_log(root_p0);
        orig_proposal = cur_root ? epv_log(1.0 - root_p0) : epv_log(root_p0);
      }
      for (uint32_t node = 1u; node < S.N; ++node) {
        const uint32_t b = node - 1u;
        if (run) {
          const PathRef L = path_ref(S, selL, b, site - 1), R = path_ref(S, selR, b, site + 1);
          const uint32_t K = L.nj + R.nj + 1u;
          const uint32_t off = regA[node * 64u + lane];
          const uint32_t par = S.parent[node];
          const uint32_t start_state = (par == 0u) ? root_state : (regA[par * 64u + lane] >> 31);
          PathRef cur;
          cur.j = nullptr; cur.nj = 0; cur.init = 0;
          if (REFQ) cur = path_ref(S, selM, b, site);
          uint32_t prev = start_state;
          bool clean = true;
          unsigned long long word = 0ull;
          uint64_t *states = S.prop_states + ((uint64_t)b * S.phase_cap + tid) * S.W;
          // forward merge of the neighbours' jumps (Segment.cpp:35-79)
          uint32_t trip0 = 4u * L.init + R.init, i = 0, j = 0;
          double seg_start = 0.0;
          double tl = L.nj ? L.j[0] : EPV_INF, tr = R.nj ? R.j[0] : EPV_INF;
          // walk of the current path
          uint32_t cs_start = cur.init, cs_end = cur.init, sj = 0, ej = 0;
          double end_time = 0.0, lp = 0.0;
          double cur_next = cur.nj ? cur.j[0] : EPV_INF;
          double pk0 = my[(size_t)off * RS], pk1 = my[(size_t)off * RS + 1u];
          for (uint32_t k = 0; k < K; ++k) {
            const bool last = (k + 1u == K);
            const bool take_left = tl < tr;
            const double seg_end = last ? s_blen[node] : (take_left ? tl : tr);
            const double len = seg_end - seg_start;
            const double nxt0 = my[(size_t)(off + k + 1u) * RS];      // p[k+1][0], or q[0]
            const double nxt1 = my[(size_t)(off + k + 1u) * RS + 1u];
            const double r0 = s_rates[trip0], r1 = s_rates[trip0 | 2u];
            const double h = epv_exp(-len * (r0 + r1));
            const double denom = r0 + r1;
            // proposal: end state of the segment
            const double PT0 = gtp(r0, r1, h, denom, prev, 0u);
            const double p0 = PT0 * nxt0 / (prev ? pk1 : pk0);
            const epv_block2 sblk = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, k, 0u, 0u);
            const uint32_t sampled = (sblk.d0 > p0) ? 1u : 0u;
            if (REFQ) {
              log_prob += (sampled == 0u) ? epv_log(p0) : epv_log(1.0 - p0);
              log_prob -= epv_log(gtp(r0, r1, h, denom, prev, sampled));
            }
            // trial 1's first draw is the other half of the same Philox block
            clean = clean && (sampled == prev) &&
                    (1.0 - sblk.d1 < nojump_bound(len * (prev ? r1 : r0)));
            word |= (unsigned long long)sampled << (k & 63u);
            if ((k & 63u) == 63u) { states[k >> 6] = word; word = 0ull; }
            if (REFQ) {
            // current path: where does it stand at the end of this segment
            end_time += len;
            while (ej < cur.nj && cur_next < end_time) {
              ++ej;
              cur_next = ej < cur.nj ? cur.j[(uint64_t)ej * n] : EPV_INF;
            }
            if ((ej - sj) & 1u) cs_end ^= 1u;
            const double PT0c = gtp(r0, r1, h, denom, cs_start, 0u);
            lp -= epv_log(gtp(r0, r1, h, denom, cs_start, cs_end));
            const double p0c = PT0c / (cs_start ? pk1 : pk0) * nxt0;
            lp += (cs_end == 0u) ? epv_log(p0c) : epv_log(1.0 - p0c);
            sj = ej;
            cs_start = cs_end;
            }
            prev = sampled;
            pk0 = nxt0; pk1 = nxt1;
            if (!last) {
              if (take_left) { trip0 ^= 4u; ++i; tl = i < L.nj ? L.j[(uint64_t)i * n] : EPV_INF; }
              else { trip0 ^= 1u; ++j; tr = j < R.nj ? R.j[(uint64_t)j * n] : EPV_INF; }
              seg_start = seg_end;
            }
          }
          // only a dirty branch is ever read back (by epv_mh_jumps_kernel)
          if ((K & 63u) && !clean) states[(K - 1u) >> 6] = word;
          // proposal so far: no jumps; epv_mh_jumps_kernel fills dirty branches in
          S.meta[meta_idx(S, selM ^ 1u, b, site)] = (epv_meta_t)(start_state << EPV_INIT_SHIFT);
          regA[node * 64u + lane] = off | (prev << 31);  // proposal end state for the children
          orig_proposal += lp;
          if (!clean) {
            dirty |= 1ull << (b & 63u);
            // four buckets by segment count: K = 1 | 2 | 3 | >= 4
            if (K == 2u || K >= 4u) multi |= 1ull << (b & 63u);
            if (K >= 3u) deep |= 1ull << (b & 63u);
          }
        }
        // flush the dirty (site, branch) pairs of the last <= 64 branches.  They are bucketed by
        // segment count so that the waves of the jumps kernel, which take consecutive tasks,
        // hold lanes with similar loop counts: a shard has two regions, each filled from BOTH
        // ends (counts packed lo/hi in one word) -- region 0: K = 1 from the front, K = 2 from
        // the back; region 1: K = 3 from the front, K >= 4 from the back.  One atomic per wave
        // and region reserves the slots, the lanes fill them in.
        if ((b & 63u) == 63u || node + 1u == S.N) {
          const uint32_t shard = blockIdx.x & (EPV_SHARDS - 1u);
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
                const unsigned long long t = ((unsigned long long)((b & ~63u) + bit) << 40) | site;
                if ((multi >> bit) & 1ull) region[slotB--] = t;
                else region[slotA++] = t;
              }
            }
          }
          dirty = 0ull;
          multi = 0ull;
          deep = 0ull;
        }
      }
    }

    // ---- hand-over to epv_mh_accept_kernel: q(old) - q(new) and the overflow flag
    if (run) {
      if (REFQ) S.prop_llr[tid] = orig_proposal - log_prob;
      S.prop_flag[tid] = 0u;
      pending = false;
    }
  }
}


// =========================================================================
//  exact forward rejection (EndCondSampling.cpp:466-509) for the dirty (site, branch)
//  pairs listed by epv_mh_propose_kernel: ONE LANE PER PAIR, so lanes are dense.  The
//  lane re-derives the branch's segments from the neighbours' jump planes (2-way merge,
//  nothing was stored), reads the sampled end states from the bit words, and for every
//  segment runs trials t = 1, 2, ... until one does not fail, appending the accepted
//  jump times to the site's proposal buffer.  Random-access Philox makes this the same
//  numbers the oracle draws inside its single per-site function.
// =========================================================================
#define EPV_TJ 4u          /* jumps a search lane hands over through LDS */
#ifndef EPV_INLINE_TRIALS
#define EPV_INLINE_TRIALS 8u /* trials a lane scans by itself before asking the wave for help */
#endif
#ifndef EPV_COOP_WINDOW
#define EPV_COOP_WINDOW 4u   /* consecutive trials one helper lane scans per round */
#endif

#ifndef EPV_JUMPS_WAVES
#define EPV_JUMPS_WAVES 3   /* waves per SIMD the register allocation aims for: 3 = 168 VGPRs, no scratch since the Philox XOR3 (round 2: 26 registers spilled, 2 was as fast); 16-leaf tree -5..-8 % per phase, tree.nwk n = 3e6 -5 % */
#endif
// (a device function with the block's place in its grid as arguments: epv_mh_jumps_all_kernel runs it in
// the first blocks of a launch whose other blocks take the one-segment tasks)
__device__ __forceinline__ void epv_jumps_body(const EpvDev &S, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep,
                                               uint32_t tasks_per_wave, uint64_t s0, double indep_r0, double indep_r1,
                                               unsigned long long *counters, uint32_t skip_single, uint32_t block_x,
                                               uint32_t grid_x, uint32_t shard) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  // per-wave cooperative-search area: task slots by rank and per-lane trial results
  __shared__ double c_len_[4][64], c_r0_[4][64], c_r1_[4][64], c_trunc_[4][64], c_tj_[4][64 * EPV_TJ];
  __shared__ uint32_t c_misc_[4][64], c_gsite_[4][64], c_tbase_[4][64], c_nk_[4][64], c_res_[4][64],
      c_tw_[4][64];
  stage_constants(S, s_mem);
  const double *s_rates = s_mem, *s_blen = s_mem + 20;
  const uint32_t wave = threadIdx.x >> 6;
  const int lane = epv_lane();
  double *c_len = c_len_[wave], *c_r0 = c_r0_[wave], *c_r1 = c_r1_[wave], *c_trunc = c_trunc_[wave],
         *c_tj = c_tj_[wave];
  uint32_t *c_misc = c_misc_[wave], *c_gsite = c_gsite_[wave], *c_tbase = c_tbase_[wave],
           *c_nk = c_nk_[wave], *c_res = c_res_[wave], *c_tw = c_tw_[wave];
  const bool indep = indep_r0 > 0.0;   // epv_indep_update_paths: rates are not context dependent
  const bool nielsen = !(S.flags & EPV_FLAG_FORWARD_REJECTION);
  // (`shard`: one task-list region per counter shard)
  // two regions per shard, each filled from both ends: buckets K = 1, 2, 3, >= 4 in this order
  const unsigned long long packed = counters[EPV_CNT_IDX(EPV_CNT_TASKS, shard)];
  const unsigned long long packed2 = counters[EPV_CNT_IDX(EPV_CNT_TASKS2, shard)];
  const unsigned long long n0 = packed & 0xffffffffull, n1 = n0 + (packed >> 32);
  const unsigned long long n2 = n1 + (packed2 & 0xffffffffull), n_tasks = n2 + (packed2 >> 32);
  const unsigned long long *tasks = S.tasks + (unsigned long long)shard * 2u * S.task_cap;
  const uint64_t n = S.n;
  const uint32_t B = S.B, C = S.C;
  // Only the first `tasks_per_wave` lanes of a wave own a task; the others are pure
  // helpers for the cooperative search.  Fewer tasks per wave = more waves in flight and a
  // shorter critical path for the rare very hard task (1/P(a->b) in the hundreds).
  // Work is handed out in chunks, one per wave and pass.  The DEEP tasks (K >= 3 segments: the
  // long loops) come first and in small chunks -- a quarter of tasks_per_wave -- so that the
  // longest chains start early and stall fewer neighbours; the shallow ones (K <= 2) follow.
#ifdef EPV_JUMPS_UNIFORM_CHUNKS
  const uint32_t tpw_deep = tasks_per_wave;
#else
  const uint32_t tpw_deep = tasks_per_wave >= 32u ? tasks_per_wave / 4u : tasks_per_wave;
#endif
  const unsigned long long n_deep = n_tasks - n1;
  // skip_single: the one-segment tasks (the first bucket) have been done by epv_mh_jumps1_kernel
  const unsigned long long first_task = skip_single ? n0 : 0ull;
  const unsigned long long W2 = (n_deep + tpw_deep - 1u) / tpw_deep, W1 = (n1 - first_task + tasks_per_wave - 1u) / tasks_per_wave;
  for (unsigned long long cidx = (unsigned long long)block_x * 4u + wave; cidx < W1 + W2;
       cidx += (unsigned long long)grid_x * 4u) {
    const bool deep_chunk = cidx < W2;
    const uint32_t my_tpw = deep_chunk ? tpw_deep : tasks_per_wave;
    const unsigned long long base = deep_chunk ? n1 + cidx * tpw_deep : first_task + (cidx - W2) * tasks_per_wave;
    const unsigned long long limit = deep_chunk ? n_tasks : n1;
    const unsigned long long ti = base + (unsigned)lane;
    bool active = (uint32_t)lane < my_tpw && ti < limit;
    uint64_t site = 0, ptid = 0;
    uint32_t b = 0, node = 1, gsite = 0, start_state = 0, prev = 0, cnt = 0, k = 0;
    uint32_t trip0 = 0, i = 0, j = 0;
    PathRef L, R;
    L.j = R.j = nullptr; L.nj = R.nj = 0; L.init = R.init = 0;
    const uint64_t *states = nullptr;
    epv_meta_t *meta = nullptr;
    double *dst = nullptr;
    double seg_start = 0.0;    // time of the previous neighbour jump (Segment.cpp's prev_time)
    double time_passed = 0.0;  // running SUM of segment lengths (SingleSiteSampler.cpp:218)
    double tl = EPV_INF, tr = EPV_INF;
    unsigned long long word = 0ull;
    bool ovf = false;
    if (active) {
      const unsigned long long task = ti < n0   ? tasks[ti]
                                      : ti < n1 ? tasks[S.task_cap - 1ull - (ti - n0)]
                                      : ti < n2 ? tasks[S.task_cap + (ti - n1)]
                                                : tasks[2ull * S.task_cap - 1ull - (ti - n2)];
      site = task & 0xffffffffffull;
      b = (uint32_t)(task >> 40) & 0xfffu;     // (the bits above may describe a one-segment branch: epv_mh_jumps1_kernel)
      node = b + 1u;
      gsite = (uint32_t)(S.g0 + site);
      const uint32_t selP = S.sel[site] ^ 1u;
      if (!indep) {   // the site-independent model has one segment per branch, no context
        L = path_ref(S, S.sel[site - 1], b, site - 1);
        R = path_ref(S, S.sel[site + 1], b, site + 1);
      }
      ptid = (site - s0) / 3u;
      states = S.prop_states + ((uint64_t)b * S.phase_cap + ptid) * S.W;
      meta = S.meta + meta_idx(S, selP, b, site);
      dst = S.jumps + ((uint64_t)selP * B + b) * C * n + site;
      start_state = (uint32_t)(*meta >> EPV_INIT_SHIFT);
      prev = start_state;
      trip0 = 4u * L.init + R.init;
      tl = L.nj ? L.j[0] : EPV_INF;
      tr = R.nj ? R.j[0] : EPV_INF;
      word = states[0];
    }
    // one segment per active lane and iteration, in Segment.cpp:35-79's merge order
    while (__any(active)) {
      bool pend = false, last = false, take_left = false;
      uint32_t sampled = 0, tbase = EPV_INLINE_TRIALS + 1u;
      double len = 0.0, r0 = 1.0, r1 = 1.0, seg_end = 0.0, trunc = 0.0;
      if (active) {
        last = !(i < L.nj || j < R.nj);
        take_left = tl < tr;
        seg_end = last ? s_blen[node] : (take_left ? tl : tr);
        len = seg_end - seg_start;
        sampled = (uint32_t)(word >> (k & 63u)) & 1u;
        r0 = indep ? indep_r0 : s_rates[trip0];
        r1 = indep ? indep_r1 : s_rates[trip0 | 2u];
        // sample_trunc_exp's 1 - exp(-rate_a T) (EndCondSampling.cpp:577-580), state changes only
        if (sampled != prev) trunc = 1.0 - epv_exp(-(prev ? r1 : r0) * len);
        if (!ovf) {
          uint32_t njt, tw;
          const int oc = scan_trials(seed_lo, seed_hi, gsite, sweep, node, k, 1u, EPV_INLINE_TRIALS, prev,
                                     sampled, len, r0, r1, trunc, C - cnt, dst + (uint64_t)cnt * n, n,
                                     0xffffffffu, time_passed, tw, njt, nielsen);
          if (oc == TRIAL_OK) cnt += njt;
          else if (oc == TRIAL_OVERFLOW) ovf = true;
          else pend = true;
        }
      }
      // Wave-cooperative search for the first non-failing trial t > EPV_INLINE_TRIALS of
      // every still-pending lane.  The P pending tasks share the 64 lanes: the task of rank
      // r gets the G = 2^floor(log2(64/P)) lanes [rG, rG+G), each scanning a window of
      // EPV_COOP_WINDOW consecutive trials of that ONE segment (random-access RNG); the lowest non-failing trial wins
      // and hands its jump times over through LDS; tasks that found none advance by
      // G * EPV_COOP_WINDOW trials.  A leaf that forces a flip on a short branch needs ~1/P(a->b) trials
      // (hundreds for the slowest context); this turns that serial tail into a few
      // full-width rounds.  Identical to the sequential "first non-failing t" whatever P, G.
      unsigned long long todo = __ballot(pend);
      while (todo) {
        const uint32_t P = (uint32_t)__popcll(todo);
        const uint32_t lg = 31u - (uint32_t)__clz((int)(64u / P));
        const uint32_t G = 1u << lg;
        const uint32_t rank = (uint32_t)__popcll(todo & ((1ull << lane) - 1ull));
        if (pend) {
          c_len[rank] = len; c_r0[rank] = r0; c_r1[rank] = r1; c_trunc[rank] = trunc;
          c_misc[rank] = prev | (sampled << 1) | ((C - cnt) << 8);
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
          const double t_len = c_len[tj], t_r0 = c_r0[tj], t_r1 = c_r1[tj], t_trunc = c_trunc[tj];
          const uint32_t t0 = c_tbase[tj] + tr_ * EPV_COOP_WINDOW, t_site = c_gsite[tj];
          uint32_t njt, tw = 0u;
          oc = scan_trials(seed_lo, seed_hi, t_site, sweep, nk >> 12, nk & 4095u, t0, EPV_COOP_WINDOW,
                           misc & 1u, (misc >> 1) & 1u, t_len, t_r0, t_r1, t_trunc, misc >> 8,
                           c_tj + (size_t)lane * EPV_TJ, 1u, EPV_TJ, 0.0, tw, njt, nielsen);
          c_tw[lane] = tw;
          c_res[lane] = (uint32_t)oc | (njt << 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long hit = __ballot(oc != TRIAL_FAIL);
        if (pend) {
          const unsigned long long mine =
              (hit >> (rank * G)) & (G == 64u ? ~0ull : ((1ull << G) - 1ull));
          if (mine) {
            const uint32_t wi = (uint32_t)(__ffsll((long long)mine) - 1);
            const uint32_t w = rank * G + wi;
            const uint32_t res = c_res[w], njt = res >> 8;
            if ((res & 0xffu) == (uint32_t)TRIAL_OK) {
              double *d2 = dst + (uint64_t)cnt * n;
              if (njt <= EPV_TJ) {
                for (uint32_t q = 0; q < njt; ++q)
                  d2[(uint64_t)q * n] = c_tj[(size_t)w * EPV_TJ + q] + time_passed;
              } else {  // rare: more jumps than the LDS hand-over holds -> replay the winner
                const uint32_t tw = c_tw[w];
                const epv_block2 fb = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, k, tw >> 1,
                                                      EPV_FIRST_DRAW_BLOCK);
                uint32_t nj2;
                run_trial(seed_lo, seed_hi, gsite, sweep, node, k, tw, (tw & 1u) ? fb.d1 : fb.d0, prev,
                          sampled, len, r0, r1, 0.0, 0.0, trunc, C - cnt, d2, n, 0xffffffffu, time_passed, nj2, nielsen);
              }
              cnt += njt;
            } else {
              ovf = true;
            }
            pend = false;
          } else {
            tbase += G * EPV_COOP_WINDOW;
          }
        }
        todo = __ballot(pend);
        __builtin_amdgcn_wave_barrier();
      }
      if (active) {
        prev = sampled;
        time_passed += len;
        if (last) {
          if (ovf) {
            cnt = (start_state ^ prev) & 1u;  // keep the end-state parity; the proposal is rejected
            S.prop_flag[ptid] = 1u;
          }
          *meta = (epv_meta_t)((start_state << EPV_INIT_SHIFT) | cnt);
          active = false;
        } else {
          if (take_left) { trip0 ^= 4u; ++i; tl = i < L.nj ? L.j[(uint64_t)i * n] : EPV_INF; }
          else { trip0 ^= 1u; ++j; tr = j < R.nj ? R.j[(uint64_t)j * n] : EPV_INF; }
          seg_start = seg_end;
          ++k;
          if ((k & 63u) == 0u) word = states[k >> 6];
        }
      }
    }
  }
}


__global__ __launch_bounds__(256, EPV_JUMPS_WAVES) void epv_mh_jumps_kernel(EpvDev S, uint32_t seed_lo,
                                                           uint32_t seed_hi, uint32_t sweep,
                                                           uint32_t tasks_per_wave, uint64_t s0,
                                                           double indep_r0, double indep_r1,
                                                           unsigned long long *counters, uint32_t skip_single = 0u) {
  epv_jumps_body(S, seed_lo, seed_hi, sweep, tasks_per_wave, s0, indep_r0, indep_r1, counters, skip_single, blockIdx.x,
                 gridDim.x, blockIdx.y);
}

// =========================================================================
//  The one-segment tasks (no neighbour jump on the branch: the first bucket of the lists, ~95 % of the
//  tasks on short branches) in a kernel of their own: no merge state, no cooperative search, a third
//  of the general kernel's work per task.  A task word written by epv_flush_tasks with EPV_TASK_COMPACT
//  carries everything; any other is completed from the meta words.  The trials are the general kernel's
//  (scan_trials): the first non-failing one wins -- searched by this lane alone, window after window
//  (with Nielsen's method a trial succeeds with probability ~1/2 or better; forward-rejection mode,
//  where a flip can need 1e5 trials, keeps the general kernel and its wave-wide search).
// =========================================================================
__device__ __forceinline__ void epv_jumps1_body(const EpvDev &S, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep,
                                                uint64_t s0, unsigned long long *counters, uint32_t block_x, uint32_t grid_x,
                                                uint32_t shard) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  stage_constants(S, s_mem);
  const double *s_rates = s_mem, *s_blen = s_mem + 20;
  const unsigned long long n0 = counters[EPV_CNT_IDX(EPV_CNT_TASKS, shard)] & 0xffffffffull;
  const unsigned long long *tasks = S.tasks + (unsigned long long)shard * 2u * S.task_cap;
  const uint64_t n = S.n;
  const uint32_t B = S.B, C = S.C;
  for (unsigned long long ti = (unsigned long long)block_x * blockDim.x + threadIdx.x; ti < n0;
       ti += (unsigned long long)grid_x * blockDim.x) {
    const unsigned long long task = tasks[ti];
    const uint64_t site = task & 0xffffffffffull;
    const uint32_t b = (uint32_t)(task >> 40) & 0xfffu, node = b + 1u;
    const uint32_t gsite = (uint32_t)(S.g0 + site);
    const uint64_t ptid = (site - s0) / 3u;
    uint32_t selP, start_state, sampled, li, ri;
    if (task & EPV_TASK_COMPACT) {
      selP = (uint32_t)(task >> EPV_TASK_SELP_SHIFT) & 1u;
      start_state = (uint32_t)(task >> EPV_TASK_START_SHIFT) & 1u;
      sampled = (uint32_t)(task >> EPV_TASK_SAMPLED_SHIFT) & 1u;
      li = (uint32_t)(task >> EPV_TASK_LINIT_SHIFT) & 1u;
      ri = (uint32_t)(task >> EPV_TASK_RINIT_SHIFT) & 1u;
    } else {
      selP = S.sel[site] ^ 1u;
      li = (uint32_t)(S.meta[meta_idx(S, S.sel[site - 1], b, site - 1)] >> EPV_INIT_SHIFT);
      ri = (uint32_t)(S.meta[meta_idx(S, S.sel[site + 1], b, site + 1)] >> EPV_INIT_SHIFT);
      start_state = (uint32_t)(S.meta[meta_idx(S, selP, b, site)] >> EPV_INIT_SHIFT);
      sampled = (uint32_t)(S.prop_states[((uint64_t)b * S.phase_cap + ptid) * S.W] & 1ull);
    }
    const uint32_t trip0 = 4u * li + ri;
    const double len = s_blen[node] - 0.0;      // (the only segment: seg_end - seg_start with seg_start = 0)
    const double r0 = s_rates[trip0], r1 = s_rates[trip0 | 2u];
    // sample_trunc_exp's 1 - exp(-rate_a T) (EndCondSampling.cpp:577-580), state changes only
    const double trunc = (sampled != start_state) ? 1.0 - epv_exp(-(start_state ? r1 : r0) * len) : 0.0;
    double *dst = S.jumps + ((uint64_t)selP * B + b) * C * n + site;
    uint32_t cnt = 0u;
    bool ovf = false;
    for (uint32_t t0 = 1u;; t0 += EPV_INLINE_TRIALS) {
      uint32_t njt, tw;
      const int oc = scan_trials(seed_lo, seed_hi, gsite, sweep, node, 0u, t0, EPV_INLINE_TRIALS, start_state, sampled, len,
                                 r0, r1, trunc, C, dst, n, 0xffffffffu, 0.0, tw, njt, true);
      if (oc == TRIAL_OK) { cnt = njt; break; }
      if (oc == TRIAL_OVERFLOW) { ovf = true; break; }
    }
    if (ovf) {
      cnt = (start_state ^ sampled) & 1u;   // keep the end-state parity; the proposal is rejected
      S.prop_flag[ptid] = 1u;
    }
    S.meta[meta_idx(S, selP, b, site)] = (epv_meta_t)((start_state << EPV_INIT_SHIFT) | cnt);
  }
}

// Both in ONE launch: the first `general_blocks` blocks of a grid row run the general kernel on the tasks
// of two and more segments -- few, but each a long dependent chain: alone they were a 58 us launch for 5 %
// of the tasks of the 16-leaf tree -- and the others the one-segment tasks next to them.
__global__ __launch_bounds__(256, EPV_JUMPS_WAVES) void epv_mh_jumps_all_kernel(EpvDev S, uint32_t seed_lo, uint32_t seed_hi,
                                                                                 uint32_t sweep, uint32_t tasks_per_wave,
                                                                                 uint64_t s0, unsigned long long *counters,
                                                                                 uint32_t general_blocks) {
  // grid (shards, blocks per shard): blocks are dispatched x-fastest, so every shard's general blocks -- the
  // long chains -- are on the chip before the first one-segment block
  if (blockIdx.y < general_blocks)
    epv_jumps_body(S, seed_lo, seed_hi, sweep, tasks_per_wave, s0, 0.0, 0.0, counters, 1u, blockIdx.y, general_blocks, blockIdx.x);
  else
    epv_jumps1_body(S, seed_lo, seed_hi, sweep, s0, counters, blockIdx.y - general_blocks, gridDim.y - general_blocks, blockIdx.x);
}

#include "epv_jumps2.h"

// acceptance of ONE site's proposal (the lane's): bit 0 = accepted and counted (own site), bit 1 =
// rejected for capacity.  s_mc = this lane's column of the meta cache (element (col * B + b) at
// [(col * B + b) * mc_stride]) when meta_cache != 0.
__device__ __forceinline__ uint32_t epv_accept_site(const EpvDev &S, const double *s_const, const double *s_blen,
                                                    epv_meta_t *s_mc, uint32_t mc_stride, uint32_t meta_cache,
                                                    AccLds &A, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep,
                                                    uint64_t tid, uint64_t site, uint64_t own_first, uint64_t own_last) {
  bool accepted = false, overflowed = false;
  const uint32_t selL = S.sel[site - 1], selM = S.sel[site], selR = S.sel[site + 1];
  const uint32_t gsite = (uint32_t)(S.g0 + site);
  const bool ovf = S.prop_flag[tid] != 0;
  double llh_l = S.tri[site - 1];
  double llh_m = S.tri[site];
  double llh_r = S.tri[site + 1];
  double llr = (S.flags & (EPV_FLAG_REFERENCE_PROPOSAL_RATIO | EPV_FLAG_SAMPLE_ROOT)) ? S.prop_llr[tid] : 0.0;
  const double llh_l_orig = llh_l, llh_r_orig = llh_r;
  if (!ovf) {
    // the three triples centred at site-1, site, site+1 with the proposal standing in
    // for this site's column (one loop, so merge3 is instantiated once)
    const uint32_t selP = selM ^ 1u;
    const uint64_t g = S.g0 + site;
    const bool hasLL = g > 1u, hasRR = g < S.n_global - 2u;
    const uint32_t selLL = hasLL ? S.sel[site - 2] : 0u;
    const uint32_t selRR = hasRR ? S.sel[site + 2] : 0u;
    if (meta_cache) {
      // all the meta words of the five columns in one batch of independent loads
      const uint32_t B = S.B;
#pragma unroll 4
      for (uint32_t b = 0; b < B; ++b) {
        const epv_meta_t m0 = hasLL ? S.meta[meta_idx(S, selLL, b, site - 2)] : (epv_meta_t)0;
        const epv_meta_t m1 = S.meta[meta_idx(S, selL, b, site - 1)];
        const epv_meta_t m2 = S.meta[meta_idx(S, selP, b, site)];
        const epv_meta_t m3 = S.meta[meta_idx(S, selR, b, site + 1)];
        const epv_meta_t m4 = hasRR ? S.meta[meta_idx(S, selRR, b, site + 2)] : (epv_meta_t)0;
        s_mc[(0u * B + b) * mc_stride] = m0;
        s_mc[(1u * B + b) * mc_stride] = m1;
        s_mc[(2u * B + b) * mc_stride] = m2;
        s_mc[(3u * B + b) * mc_stride] = m3;
        s_mc[(4u * B + b) * mc_stride] = m4;
      }
    }
    for (int w = 0; w < 3; ++w) {
      if ((w == 0 && !hasLL) || (w == 2 && !hasRR)) continue;
      const uint64_t c = site - 1u + (uint64_t)w;
      const uint32_t bl = (w == 0) ? selLL : (w == 1) ? selL : selP;
      const uint32_t bm = (w == 0) ? selL : (w == 1) ? selP : selR;
      const uint32_t br = (w == 0) ? selP : (w == 1) ? selR : selRR;
      const double v = meta_cache
                           ? triple_llh_cached(S, s_const, s_blen, s_mc, mc_stride, (uint32_t)w, bl, c - 1u, (uint32_t)w + 1u,
                                               bm, c, (uint32_t)w + 2u, br, c + 1u, A)
                           : triple_llh(S, s_const, s_blen, bl, c - 1u, bm, c, br, c + 1u, A);
      if (w == 0) llh_l = v; else if (w == 1) llh_m = v; else llh_r = v;
    }
  }
  llr += (llh_l + llh_r - llh_l_orig - llh_r_orig);
  const double u = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, 0u, 0u, 0u, 0u).d0;
  bool acc = (llr >= 0.0) || (u < epv_exp(llr));
  if (ovf) { acc = false; overflowed = true; }
  if (acc) {
    S.sel[site] = (uint8_t)(selM ^ 1u);
    S.tri[site - 1] = llh_l;
    S.tri[site] = llh_m;
    S.tri[site + 1] = llh_r;
    // redundant updates of halo columns (site-sharded runs) are not counted
    accepted = site >= own_first && site <= own_last;
  }
  return (accepted ? 1u : 0u) | (overflowed ? 2u : 0u);
}

// =========================================================================
//  acceptance (log_accept_rate SingleSiteSampler.cpp:396-433, Metropolis_Hastings_site
//  :510-533): one lane per site of the colour, after epv_mh_propose_kernel has written
//  the proposal into the site's other buffer.  No record pool here, so this part runs
//  at full occupancy.
// =========================================================================
// 6 waves/SIMD (<= 80 VGPRs) = 6 blocks per CU, matching the 24 KB of LDS per block: the
// 1302 blocks of a 1e6-site phase then fit the 1536 slots in ONE round (at the natural 92
// VGPRs there are 1280 slots and 22 blocks run alone in a second round: +15 us)
#ifndef EPV_ACCEPT_WAVES
#define EPV_ACCEPT_WAVES 4
#endif
__global__ __launch_bounds__(256, EPV_ACCEPT_WAVES) void epv_mh_accept_kernel(
    EpvDev S, uint32_t colour, uint32_t seed_lo, uint32_t seed_hi, uint32_t sweep,
    uint64_t first, uint64_t last, uint64_t own_first, uint64_t own_last,
    unsigned long long *counters, uint32_t list_mode, uint32_t meta_cache) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  __shared__ double s_accd[8 * 256];
  __shared__ uint32_t s_accj[8 * 256];
  stage_constants(S, s_mem);
  const double *s_const = s_mem, *s_blen = s_mem + 20;
  // meta cache [5 columns][B][256 lanes] behind the constants when the launch provided room for
  // it (meta_cache != 0: small trees); column 2 = the proposal
  epv_meta_t *s_mc = reinterpret_cast<epv_meta_t *>(s_mem + ((20u + S.N + 1u) & ~1u)) + threadIdx.x;
  const int lane = epv_lane();
  AccLds A;
  A.d = s_accd + threadIdx.x; A.j = s_accj + threadIdx.x; A.stride = 256u;
  const uint64_t gfirst = S.g0 + first;
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)(gfirst % 3u)) % 3u);
  // list_mode 0: one lane per site of the colour (grid x covers them).  list_mode 1 / 2: the
  // sites epv_mh_propose2_kernel listed (proposal differs from the current path), parity
  // list_mode - 1, one shard of the list per grid row, grid-stride over its entries.
  const uint32_t shard_row = blockIdx.y;
  unsigned long long n_list = 0ull;
  if (list_mode) n_list = counters[EPV_CNT_IDX(list_mode == 2u ? EPV_CNT_ALIST1 : EPV_CNT_ALIST0, shard_row)];
  const uint64_t step = list_mode ? (uint64_t)gridDim.x * blockDim.x : ~0ull;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; list_mode ? base < n_list : base == (uint64_t)blockIdx.x * blockDim.x;
       base += step) {
  uint64_t tid = base + threadIdx.x;
  bool have = true;
  if (list_mode) {
    have = tid < n_list;
    tid = have ? S.alist[(uint64_t)shard_row * S.alist_cap + tid] : 0u;
  }
  const uint64_t site = s0 + 3u * tid;
  bool accepted = false, overflowed = false;
  if (have && site <= last) {
    const uint32_t r = epv_accept_site(S, s_const, s_blen, s_mc, 256u, meta_cache, A, seed_lo, seed_hi, sweep, tid, site,
                                       own_first, own_last);
    accepted = r & 1u; overflowed = (r & 2u) != 0u;
  }
  const unsigned long long am = __ballot(accepted), om = __ballot(overflowed);
  if (lane == 0) {
    const uint32_t shard = (blockIdx.x + blockIdx.y) & (EPV_SHARDS - 1u);
    if (am) atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_ACCEPT, shard)], (unsigned long long)__popcll(am));
    if (om) atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_OVERFLOW, shard)], (unsigned long long)__popcll(om));
  }
  }
  // the task lists of this phase have been consumed (stream order): fold their lengths
  // into the running total and clear them for the next propose kernel; the accept list of the
  // OTHER parity was consumed by the previous phase's accept kernel and is filled next
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

// the second proposal kernel; its fused variant runs the search, assembly and acceptance above
#include "epv_propose2.h"
#include "epv_propose3.h"
#include "epv_accept3.h"

// =========================================================================
//  initialize_paths_indep (src/prog/epievo_sim_pairwise.cpp:62-110) on the device: every
//  interior site of a single branch gets an independent end-conditioned path
//  root[i] -> leaf[i] by forward rejection with the context rates read off the ROOT
//  sequence.  The device paths start as (init = root, no jumps); per colour (so that the
//  phase-sized hand-over arrays suffice) epv_init_tasks_kernel lists every site as a
//  task with end state leaf[i], epv_mh_jumps_kernel -- the very kernel of the MCMC phase,
//  run with the reserved sweep index EPV_INIT_SWEEP -- draws the jumps into the other
//  buffer, epv_init_commit_kernel collects overflow flags, and once all three colours are
//  done epv_init_flip_kernel flips every interior site over to its new path.
// =========================================================================
#define EPV_INIT_SWEEP 0xffffffffu

__global__ __launch_bounds__(256) void epv_init_tasks_kernel(EpvDev S, uint32_t colour, uint64_t first,
                                                             uint64_t last, const uint8_t *leaf,
                                                             unsigned long long *counters) {
  const int lane = epv_lane();
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)((S.g0 + first) % 3u)) % 3u);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site = s0 + 3u * tid;
  const bool valid = site <= last;
  if (valid) {
    const uint32_t sel = S.sel[site];
    const uint32_t root = (uint32_t)(S.meta[meta_idx(S, sel, 0u, site)] >> EPV_INIT_SHIFT);
    S.meta[meta_idx(S, sel ^ 1u, 0u, site)] = (epv_meta_t)(root << EPV_INIT_SHIFT);
    S.prop_states[tid * S.W] = leaf[site] ? 1ull : 0ull;
    S.prop_flag[tid] = 0u;
  }
  const uint32_t mine = valid ? 1u : 0u;
  const uint32_t incl = wave_incl_scan_u32(mine);
  const uint32_t total = epv_bcast(incl, 63);
  if (total) {
    const uint32_t shard = blockIdx.x & (EPV_SHARDS - 1u);
    unsigned long long base = 0ull;
    if (lane == 0)
      base = atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_TASKS, shard)], (unsigned long long)total);
    base = ((unsigned long long)epv_bcast((uint32_t)(base >> 32), 0) << 32) |
           (unsigned long long)epv_bcast((uint32_t)base, 0);
    if (valid) S.tasks[(unsigned long long)shard * 2u * S.task_cap + base + (incl - mine)] = site;
  }
}

__global__ __launch_bounds__(256) void epv_init_commit_kernel(EpvDev S, uint32_t colour, uint64_t first,
                                                              uint64_t last,
                                                              unsigned long long *counters) {
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)((S.g0 + first) % 3u)) % 3u);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site = s0 + 3u * tid;
  bool overflowed = false;
  // the flip to the new paths is deferred until every colour has been drawn
  // (epv_init_flip_kernel): the contexts must come from the ROOT sequence alone, not from
  // neighbours that already carry their new jumps
  if (site <= last && S.prop_flag[tid]) overflowed = true;
  const unsigned long long om = __ballot(overflowed);
  if (epv_lane() == 0 && om)
    atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_OVERFLOW, blockIdx.x & (EPV_SHARDS - 1u))],
              (unsigned long long)__popcll(om));
  if (blockIdx.x == 0 && threadIdx.x < EPV_SHARDS) counters[EPV_CNT_IDX(EPV_CNT_TASKS, threadIdx.x)] = 0ull;
}

__global__ __launch_bounds__(256) void epv_init_flip_kernel(EpvDev S, uint64_t first, uint64_t last) {
  const uint64_t site = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site <= last) S.sel[site] ^= 1u;
}

// the two end sites: at most one jump, placed uniformly (epievo_sim_pairwise.cpp:77-88)
__global__ void epv_init_ends_kernel(EpvDev S, const uint8_t *leaf, uint32_t seed_lo, uint32_t seed_hi,
                                     double T) {
  if (threadIdx.x > 1) return;
  const uint64_t site = threadIdx.x ? S.n - 1u : 0u;
  const uint32_t sel = S.sel[site];
  const uint32_t root = (uint32_t)(S.meta[meta_idx(S, sel, 0u, site)] >> EPV_INIT_SHIFT);
  if (root != (uint32_t)leaf[site]) {
    const double u = epv_keyed_block(seed_lo, seed_hi, (uint32_t)(S.g0 + site), EPV_INIT_SWEEP, 1u, 0u, 0u, 0u).d0;
    S.jumps[((uint64_t)sel * S.B) * S.C * S.n + site] = u * (T - 0.0) + 0.0;
    S.meta[meta_idx(S, sel, 0u, site)] = (epv_meta_t)((root << EPV_INIT_SHIFT) | 1u);
  }
}

// =========================================================================
//  Site-independent model (IndepSite.cpp), used by epievo_initialization.  One lane per
//  site; per-node Felsenstein values {q0,q1,p0,p1} in LDS [node][lane]; the per-branch
//  matrices come from EpvIndepConst, so there is no exp/log on this path.
// =========================================================================
__device__ __forceinline__ void indep_upward(const EpvDev &S, const EpvIndepConst *ic, uint32_t sel,
                                             uint64_t site, double *fh /* [N][blockDim][4] */) {
  // upward_process (IndepSite.cpp:53-96)
  const uint32_t T = blockDim.x, t = threadIdx.x;
  for (uint32_t node = S.N; node-- > 0u;) {
    double a = 1.0, b = 1.0;
    const uint32_t sub = S.subtree[node];
    if (sub == 1u) {
      const uint32_t m = S.meta[meta_idx(S, sel, node - 1u, site)];
      const uint32_t leaf_state = (m >> EPV_INIT_SHIFT) ^ (m & 1u);
      a = leaf_state ? 0.0 : 1.0;
      b = leaf_state ? 1.0 : 0.0;
    } else {
      for (uint32_t ch = 1u; ch < sub; ch += S.subtree[node + ch]) {
        const double *c = fh + ((size_t)(node + ch) * T + t) * 4u;
        a *= c[2];
        b *= c[3];
      }
    }
    double *me = fh + ((size_t)node * T + t) * 4u;
    me[0] = a; me[1] = b;
    if (node == 0u) continue;
    const double *P = ic[node].P;
    me[2] = P[0] * a + P[1] * b;
    me[3] = P[2] * a + P[3] * b;
  }
}

// expectation_sufficient_statistics (IndepSite.cpp:98-175, :222-238): per branch the
// conditional means {J0, J1, D0, D1} of every site, reduced in the canonical tree order.
// partial layout [block][V16], column 4*(node-1) + {0,1,2,3}, zero padded to V16.
// what = 0: conditional expectations; what = 1: counts of the current paths
// (compute_sufficient_statistics, :266-297).
__global__ __launch_bounds__(256) void epv_indep_stats_kernel(EpvDev S, const EpvIndepConst *ic,
                                                              double pi_0, uint32_t what, uint32_t V16,
                                                              double *partial) {
  extern __shared__ __attribute__((aligned(16))) double fh[];
  __shared__ double s_part[2][4][16];
  const int lane = epv_lane();
  const uint32_t wave = threadIdx.x >> 6;
  const uint64_t site = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = site < S.n;
  const uint32_t sel = on ? S.sel[site] : 0u;
  double *pm = fh + (size_t)S.N * blockDim.x * 4u;   // [N][blockDim] marginal P(state 0)
  if (on && what == 0u) {
    indep_upward(S, ic, sel, site, fh);
    const double *root = fh + (size_t)threadIdx.x * 4u;
    const double a = pi_0 * root[0], b = (1 - pi_0) * root[1];
    pm[threadIdx.x] = a / (a + b);   // root_post_prob0 (:98-104)
  }
  for (uint32_t g = 0; g < V16 / 16u; ++g) {
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t node = g * 4u + (uint32_t)q + 1u;
      if (!on || node >= S.N) continue;
      double j0 = 0.0, j1 = 0.0, d0 = 0.0, d1 = 0.0;
      if (what == 0u) {
        // joint_post + weighted_J_D_branch (:106-152)
        const double *me = fh + ((size_t)node * blockDim.x + threadIdx.x) * 4u;
        const EpvIndepConst &k = ic[node];
        const double p0u = pm[(size_t)S.parent[node] * blockDim.x + threadIdx.x];
        double pj[4];
        pj[0] = k.P[0] * me[0] * p0u / me[2];
        pj[1] = k.P[1] * me[1] * p0u / me[2];
        pj[2] = k.P[2] * me[0] * (1 - p0u) / me[3];
        pj[3] = k.P[3] * me[1] * (1 - p0u) / me[3];
        const double Z = pj[0] + pj[1] + pj[2] + pj[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) pj[i] /= Z;
        pm[(size_t)node * blockDim.x + threadIdx.x] = pj[0] + pj[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          j0 += pj[i] * k.J0[i];
          j1 += pj[i] * k.J1[i];
          d0 += pj[i] * k.D0[i];
          d1 += pj[i] * k.D1[i];
        }
      } else {
        const PathRef p = path_ref(S, sel, node - 1u, site);
        uint32_t prev = p.init;
        double time = 0.0;
        for (uint32_t jx = 0; jx < p.nj; ++jx) {
          const double tj = p.j[(uint64_t)jx * S.n];
          if (prev) { j1 += 1; d1 += (tj - time); } else { j0 += 1; d0 += (tj - time); }
          prev ^= 1u;
          time = tj;
        }
        if (prev) d1 += (S.blen[node] - time); else d0 += (S.blen[node] - time);
      }
      v[4 * q + 0] = j0; v[4 * q + 1] = j1; v[4 * q + 2] = d0; v[4 * q + 3] = d1;
    }
    int idx;
    const double tot = wave_tree_sum16(v, lane, idx);
    if (lane < 16) s_part[g & 1u][wave][idx] = tot;
    __syncthreads();
    if (threadIdx.x < 16) {
      // blocks are 256 lanes (4 waves) or, for large trees, 64 lanes (1 wave): any
      // power-of-two grouping is a level of the same balanced binary tree
      const double(*p)[16] = s_part[g & 1u];
      partial[(uint64_t)blockIdx.x * V16 + g * 16u + threadIdx.x] =
          (blockDim.x == 64u) ? p[0][threadIdx.x]
                              : (p[0][threadIdx.x] + p[1][threadIdx.x]) + (p[2][threadIdx.x] + p[3][threadIdx.x]);
    }
  }
}

// update_paths_indep (IndepSite.cpp:177-215, :241-259), part 1: end state of every
// branch; clean branches (state kept, provably no jump in trial 1) are final, the rest
// goes to epv_mh_jumps_kernel (run with the two context-free rates).  Processed in three
// thirds of the sites so that the phase-sized hand-over arrays suffice.
__global__ __launch_bounds__(64) void epv_indep_propose_kernel(EpvDev S, const EpvIndepConst *ic,
                                                              double r0, double r1, uint32_t colour,
                                                              uint64_t first, uint64_t last,
                                                              uint32_t seed_lo, uint32_t seed_hi,
                                                              uint32_t sweep,
                                                              unsigned long long *counters) {
  extern __shared__ __attribute__((aligned(16))) double fh[];
  const int lane = epv_lane();
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)((S.g0 + first) % 3u)) % 3u);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site = s0 + 3u * tid;
  const bool valid = site <= last;
  unsigned long long dirty = 0ull;
  uint32_t sel = 0, root_state = 0;
  if (valid) {
    sel = S.sel[site];
    indep_upward(S, ic, sel, site, fh);
    root_state = (uint32_t)(S.meta[meta_idx(S, sel, 0u, site)] >> EPV_INIT_SHIFT);
    S.prop_flag[tid] = 0u;
  }
  const uint32_t gsite = (uint32_t)(S.g0 + site);
  for (uint32_t node = 1u; node < S.N; ++node) {
    const uint32_t b = node - 1u;
    if (valid) {
      double *me = fh + ((size_t)node * blockDim.x + threadIdx.x) * 4u;
      const uint32_t par = S.parent[node];
      const uint32_t start =
          (par == 0u) ? root_state : (uint32_t)epv_d2u(fh[((size_t)par * blockDim.x + threadIdx.x) * 4u + 2u]);
      const double *P = ic[node].P;
      const double pr0 = P[2u * start] * me[0] / (start ? me[3] : me[2]);
      const epv_block2 sblk = epv_keyed_block(seed_lo, seed_hi, gsite, sweep, node, 0u, 0u, 0u);
      const uint32_t sampled = (sblk.d0 > pr0) ? 1u : 0u;
      const bool clean = (sampled == start) &&
                         (1.0 - sblk.d1 < nojump_bound(S.blen[node] * (start ? r1 : r0)));
      S.prop_states[((uint64_t)b * S.phase_cap + tid) * S.W] = sampled;
      S.meta[meta_idx(S, sel ^ 1u, b, site)] = (epv_meta_t)(start << EPV_INIT_SHIFT);
      me[2] = epv_u2d((uint64_t)sampled);   // proposal end state for the children
      if (!clean) dirty |= 1ull << (b & 63u);
    }
    if ((b & 63u) == 63u || node + 1u == S.N) {
      const uint32_t mine = (uint32_t)__popcll(dirty);
      const uint32_t incl_t = wave_incl_scan_u32(mine);
      const uint32_t total = epv_bcast(incl_t, 63);
      if (total) {
        unsigned long long base = 0ull;
        const uint32_t shard = blockIdx.x & (EPV_SHARDS - 1u);
        if (lane == 0)
          base = atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_TASKS, shard)], (unsigned long long)total);
        base = ((unsigned long long)epv_bcast((uint32_t)(base >> 32), 0) << 32) |
               (unsigned long long)epv_bcast((uint32_t)base, 0);
        unsigned long long slot = (unsigned long long)shard * 2u * S.task_cap + base + (incl_t - mine);
        unsigned long long d = dirty;
        while (d) {
          const uint32_t bit = (uint32_t)(__ffsll((long long)d) - 1);
          d &= d - 1ull;
          S.tasks[slot++] = ((unsigned long long)((b & ~63u) + bit) << 40) | site;
        }
      }
      dirty = 0ull;
    }
  }
}

// part 3: the new paths always replace the old ones (this is direct sampling, not MH)
__global__ __launch_bounds__(256) void epv_indep_commit_kernel(EpvDev S, uint32_t colour, uint64_t first,
                                                               uint64_t last,
                                                               unsigned long long *counters) {
  const uint64_t s0 = first + ((colour + 3u - (uint32_t)((S.g0 + first) % 3u)) % 3u);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t site = s0 + 3u * tid;
  bool overflowed = false;
  if (site <= last) {
    if (S.prop_flag[tid]) overflowed = true;
    else S.sel[site] ^= 1u;
  }
  const unsigned long long om = __ballot(overflowed);
  if (epv_lane() == 0 && om)
    atomicAdd(&counters[EPV_CNT_IDX(EPV_CNT_OVERFLOW, blockIdx.x & (EPV_SHARDS - 1u))],
              (unsigned long long)__popcll(om));
  if (blockIdx.x == 0 && threadIdx.x < EPV_SHARDS) counters[EPV_CNT_IDX(EPV_CNT_TASKS, threadIdx.x)] = 0ull;
}

// =========================================================================
//  reset: tri[s] = path_log_likelihood(s-1, s, s+1) for every local interior site
// =========================================================================
__global__ __launch_bounds__(256) void epv_reset_kernel(EpvDev S) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  stage_constants(S, s_mem);
  const uint64_t site = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= S.n) return;
  double v = 0.0;
  if (site >= 1 && site + 1 < S.n) {
    Acc8 A;
    v = triple_llh(S, s_mem, s_mem + 20, S.sel[site - 1], site - 1, S.sel[site], site,
                   S.sel[site + 1], site + 1, A);
  }
  S.tri[site] = v;
}

// =========================================================================
//  sufficient statistics (get_sufficient_statistics, ParamEstimation.cpp:92-114;
//  add_sufficient_statistics, Path.cpp:206-301): per-branch J[8], D[8] over the triples centred
//  at the owned local sites, as EXACT integers -- J as counts, every dwell time as the
//  fixed-point integer rint(dt * 2^k_b) (statscale[b + 1] = 2^k_b from the host, see
//  stat_scale_exp in epv_abi.hip and oracle/epv_oracle.c).  int64 sums are associative, so the
//  totals do not depend on the launch shape, on how a genome is cut into contexts and GPUs, or on
//  the order atomics land in; the host turns them back into doubles.
//
//  One block = 256 consecutive sites, all branches (blockIdx.y: chunks of EPV_STAT_BCH).
//  Per branch a lane loads ITS site's meta word, the neighbours' come through LDS.  86 % of the
//  (site, branch) pairs of a short tree have no jump in the triple: D[ctx] += T, i.e. a
//  histogram of the 3-bit context -- three ballots and eight popcounts per wave, multiplied by
//  fix(T) once per block.  The other pairs are queued in an LDS ring and merged with DENSE lanes,
//  256 at a time (walking the branches with whichever lanes have a jump there kept one lane in
//  seven busy), their events added to the block's LDS accumulators with ds_add_u64.
//  partial layout: [block][b][16] int64 (J then D).
// =========================================================================
#define EPV_STAT_BCH 32u
struct AccExact {
  unsigned long long *acc;   // this branch's 16 LDS words: J[8], D[8]
  double scale;              // 2^k_b
};
// rint(x) for |x| < 2^51 as an integer: adding 1.5 * 2^52 leaves it in the low mantissa bits
// (round to nearest even, what llrint does on the host)
__device__ __forceinline__ unsigned long long epv_stat_fix(double dt, double scale) {
  const double y = dt * scale + 6755399441055744.0;
  return epv_d2u(y) - 0x4338000000000000ull;
}
__device__ __forceinline__ void acc_add(AccExact &A, int ctx, double dt, bool mid) {
  atomicAdd(&A.acc[8 + ctx], epv_stat_fix(dt, A.scale));
  if (mid) atomicAdd(&A.acc[ctx], 1ull);
}

__global__ __launch_bounds__(256) void epv_suffstat_kernel(EpvDev S, uint64_t first, uint64_t last, uint64_t block0,
                                                           const double *statscale, unsigned long long *partial) {
  __shared__ unsigned long long s_acc[EPV_STAT_BCH * 16u];
  __shared__ uint32_t s_cnt[EPV_STAT_BCH * 8u];     // triples without a jump, per (branch, context)
  __shared__ epv_meta_t s_meta[2][258];
  __shared__ uint8_t s_sel[258];
  // queue of the pairs that need a merge: thread | local branch << 8, and the triple's three meta words
  __shared__ uint32_t s_ring[512], s_ring_lm[512];
  __shared__ epv_meta_t s_ring_r[512];
  __shared__ uint32_t s_tail;
  const uint32_t t = threadIdx.x;
  const int lane = epv_lane();
  const uint32_t B = S.B;
  const uint32_t b_lo = blockIdx.y * EPV_STAT_BCH, b_hi = (b_lo + EPV_STAT_BCH < B) ? b_lo + EPV_STAT_BCH : B;
  const uint64_t n = S.n, Bn = (uint64_t)B * n, Cn = (uint64_t)S.C * n;
  // block0: first 256-site block to process (partial[] is relative to it)
  const uint64_t site0 = (block0 + blockIdx.x) * 256u, site = site0 + t;
  const bool on = site >= first && site <= last && site >= 1 && site + 1 < n;
  // the kernel is a chain of memory round trips (sel -> meta -> jumps), not of instructions: every
  // load that does not depend on another is issued with its siblings
  const uint32_t my_sel = site < n ? S.sel[site] : 0u;
  // the two columns next to the block: threads 0 and 1 fetch them
  const bool edge = (t == 0 && site0 >= 1) || (t == 1 && site0 + 256u < n);
  const uint64_t esite = t == 0 ? site0 - 1 : site0 + 256u;
  const uint32_t e_sel = edge ? S.sel[esite] : 0u;
  for (uint32_t i = t; i < EPV_STAT_BCH * 16u; i += 256u) s_acc[i] = 0ull;
  s_cnt[t] = 0u;      // EPV_STAT_BCH * 8 = 256 entries
  if (t == 0) s_tail = 0u;
  s_sel[t + 1u] = (uint8_t)my_sel;
  if (t < 2u) s_sel[t == 0 ? 0 : 257] = (uint8_t)e_sel;
  const uint64_t mbase = (my_sel ? Bn : 0ull) + site;
  const uint64_t ebase = (e_sel ? Bn : 0ull) + esite;

  auto merge_item = [&](uint32_t slot) __attribute__((always_inline)) {
    const uint32_t item = s_ring[slot], lm = s_ring_lm[slot], mr = s_ring_r[slot];
    const uint32_t ti = item & 255u, bl = item >> 8, b = b_lo + bl;
    const uint64_t si = site0 + ti;
    const uint32_t ml = lm & 0xffffu, mm = lm >> 16;
    PathRef L, M, R;
    L.j = S.jumps + (s_sel[ti] ? Bn * S.C : 0ull) + (uint64_t)b * Cn + (si - 1); L.nj = ml & EPV_NJ_MASK; L.init = ml >> EPV_INIT_SHIFT;
    M.j = S.jumps + (s_sel[ti + 1u] ? Bn * S.C : 0ull) + (uint64_t)b * Cn + si; M.nj = mm & EPV_NJ_MASK; M.init = mm >> EPV_INIT_SHIFT;
    R.j = S.jumps + (s_sel[ti + 2u] ? Bn * S.C : 0ull) + (uint64_t)b * Cn + (si + 1); R.nj = mr & EPV_NJ_MASK; R.init = mr >> EPV_INIT_SHIFT;
    AccExact A;
    A.acc = s_acc + bl * 16u;
    A.scale = statscale[b + 1u];
    merge3(L, M, R, n, S.blen[b + 1u], A);
  };

  uint32_t head = 0u, buf = 0u;
  constexpr uint32_t GB = 4u;      // branches whose meta words are fetched together
  for (uint32_t b0 = b_lo; b0 < b_hi; b0 += GB) {
    epv_meta_t mq[GB], eq[GB];
#pragma unroll
    for (uint32_t q = 0; q < GB; ++q) {
      const bool have = b0 + q < b_hi;
      mq[q] = (have && site < n) ? S.meta[mbase + (uint64_t)(b0 + q) * n] : (epv_meta_t)0;
      eq[q] = (have && edge) ? S.meta[ebase + (uint64_t)(b0 + q) * n] : (epv_meta_t)0;
    }
#pragma unroll
    for (uint32_t q = 0; q < GB; ++q) {
      const uint32_t b = b0 + q;
      if (b >= b_hi) break;
      const uint32_t m = mq[q];
      s_meta[buf][t + 1u] = (epv_meta_t)m;
      if (t < 2u) s_meta[buf][t == 0 ? 0 : 257] = eq[q];
      __syncthreads();
      const uint32_t ml = s_meta[buf][t], mr = s_meta[buf][t + 2u];
      const uint32_t or3 = (ml | m | mr) & EPV_NJ_MASK;
      const bool fast = on && or3 == 0u, slow = on && or3 != 0u;
      const unsigned long long mf = __ballot(fast), m2 = __ballot((ml >> EPV_INIT_SHIFT) != 0u),
                               m1 = __ballot((m >> EPV_INIT_SHIFT) != 0u), m0 = __ballot((mr >> EPV_INIT_SHIFT) != 0u);
      if (lane < 8) {
        const unsigned long long x = mf & ((lane & 4) ? m2 : ~m2) & ((lane & 2) ? m1 : ~m1) & ((lane & 1) ? m0 : ~m0);
        const uint32_t cnt = (uint32_t)__popcll(x);
        if (cnt) atomicAdd(&s_cnt[(b - b_lo) * 8u + (uint32_t)lane], cnt);
      }
      const unsigned long long ms = __ballot(slow);
      if (ms) {
        uint32_t base = 0u;
        if (lane == 0) base = atomicAdd(&s_tail, (uint32_t)__popcll(ms));
        base = epv_bcast(base, 0);
        if (slow) {
          const uint32_t slot = (base + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull))) & 511u;
          s_ring[slot] = t | ((b - b_lo) << 8);
          s_ring_lm[slot] = ml | (m << 16);
          s_ring_r[slot] = (epv_meta_t)mr;
        }
      }
      __syncthreads();
      // s_tail is only written again behind the next iteration's barrier
      if (s_tail - head >= 256u) {
        merge_item((head + t) & 511u);
        head += 256u;
      }
      buf ^= 1u;
    }
  }
  __syncthreads();
  if (t < s_tail - head) merge_item((head + t) & 511u);
  __syncthreads();
  for (uint32_t i = t; i < (b_hi - b_lo) * 16u; i += 256u) {
    const uint32_t bl = i >> 4, c = i & 15u, b = b_lo + bl;
    unsigned long long v = s_acc[i];
    if (c >= 8u) v += (unsigned long long)s_cnt[bl * 8u + (c - 8u)] * epv_stat_fix(S.blen[b + 1u] - 0.0, statscale[b + 1u]);
    partial[((uint64_t)blockIdx.x * B + b) * 16u + c] = v;
  }
}

// The same statistics with ONE WAVE per block and ~2 KB of LDS, for launches that run next to other
// contexts' colour phases: those fill the CUs' LDS to within 11 KB (nine 16.5 KB blocks of 160 KB), so
// a 256-lane block with 11.5 KB waits for a colour-phase block to leave and then keeps the next one
// out, while a one-wave block slips into the gap.  64 sites and up to EPV_STATW_BCH branches per
// block; neighbours' meta words through LDS, no block barrier; the four waves of a 256-site block add
// their sums into the block's row of `partial` with 64-bit atomics (the row must be zero: integer sums
// commute, so the result is the one of epv_suffstat_kernel).
#define EPV_STATW_BCH 8u
__global__ __launch_bounds__(64) void epv_suffstat_wave_kernel(EpvDev S, uint64_t first, uint64_t last, uint64_t block0,
                                                               const double *statscale, unsigned long long *partial) {
  __shared__ unsigned long long s_acc[EPV_STATW_BCH * 16u];
  __shared__ uint32_t s_cnt[EPV_STATW_BCH * 8u];
  __shared__ epv_meta_t s_meta[66];
  __shared__ uint8_t s_sel[66];
  __shared__ uint32_t s_ring[128], s_ring_lm[128];
  __shared__ epv_meta_t s_ring_r[128];
  const uint32_t t = threadIdx.x;
  const int lane = (int)t;
  const uint32_t B = S.B;
  const uint32_t b_lo = blockIdx.y * EPV_STATW_BCH, b_hi = (b_lo + EPV_STATW_BCH < B) ? b_lo + EPV_STATW_BCH : B;
  const uint64_t n = S.n, Bn = (uint64_t)B * n, Cn = (uint64_t)S.C * n;
  const uint64_t site0 = block0 * 256u + (uint64_t)blockIdx.x * 64u, site = site0 + t;
  const bool on = site >= first && site <= last && site >= 1 && site + 1 < n;
  const uint32_t my_sel = site < n ? S.sel[site] : 0u;
  const bool edge = (t == 0 && site0 >= 1) || (t == 1 && site0 + 64u < n);
  const uint64_t esite = t == 0 ? site0 - 1 : site0 + 64u;
  const uint32_t e_sel = edge ? S.sel[esite] : 0u;
  for (uint32_t i = t; i < EPV_STATW_BCH * 16u; i += 64u) s_acc[i] = 0ull;
  s_cnt[t] = 0u;     // EPV_STATW_BCH * 8 = 64 entries
  s_sel[t + 1u] = (uint8_t)my_sel;
  if (t < 2u) s_sel[t == 0 ? 0 : 65] = (uint8_t)e_sel;
  const uint64_t mbase = (my_sel ? Bn : 0ull) + site;
  const uint64_t ebase = (e_sel ? Bn : 0ull) + esite;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();

  auto merge_item = [&](uint32_t slot) __attribute__((always_inline)) {
    const uint32_t item = s_ring[slot], lm = s_ring_lm[slot], mr = s_ring_r[slot];
    const uint32_t ti = item & 63u, bl = item >> 8, b = b_lo + bl;
    const uint64_t si = site0 + ti;
    const uint32_t ml = lm & 0xffffu, mm = lm >> 16;
    PathRef L, M, R;
    L.j = S.jumps + (s_sel[ti] ? Bn * S.C : 0ull) + (uint64_t)b * Cn + (si - 1); L.nj = ml & EPV_NJ_MASK; L.init = ml >> EPV_INIT_SHIFT;
    M.j = S.jumps + (s_sel[ti + 1u] ? Bn * S.C : 0ull) + (uint64_t)b * Cn + si; M.nj = mm & EPV_NJ_MASK; M.init = mm >> EPV_INIT_SHIFT;
    R.j = S.jumps + (s_sel[ti + 2u] ? Bn * S.C : 0ull) + (uint64_t)b * Cn + (si + 1); R.nj = mr & EPV_NJ_MASK; R.init = mr >> EPV_INIT_SHIFT;
    AccExact A;
    A.acc = s_acc + bl * 16u;
    A.scale = statscale[b + 1u];
    merge3(L, M, R, n, S.blen[b + 1u], A);
  };

  uint32_t head = 0u, tail = 0u;     // wave-uniform
  constexpr uint32_t GB = 4u;
  for (uint32_t b0 = b_lo; b0 < b_hi; b0 += GB) {
    epv_meta_t mq[GB], eq[GB];
#pragma unroll
    for (uint32_t q = 0; q < GB; ++q) {
      const bool have = b0 + q < b_hi;
      mq[q] = (have && site < n) ? S.meta[mbase + (uint64_t)(b0 + q) * n] : (epv_meta_t)0;
      eq[q] = (have && edge) ? S.meta[ebase + (uint64_t)(b0 + q) * n] : (epv_meta_t)0;
    }
#pragma unroll
    for (uint32_t q = 0; q < GB; ++q) {
      const uint32_t b = b0 + q;
      if (b >= b_hi) break;
      const uint32_t m = mq[q];
      s_meta[t + 1u] = (epv_meta_t)m;
      if (t < 2u) s_meta[t == 0 ? 0 : 65] = eq[q];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t ml = s_meta[t], mr = s_meta[t + 2u];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();     // everybody has read: the next branch may overwrite
      const uint32_t or3 = (ml | m | mr) & EPV_NJ_MASK;
      const bool fast = on && or3 == 0u, slow = on && or3 != 0u;
      const unsigned long long mf = __ballot(fast), m2 = __ballot((ml >> EPV_INIT_SHIFT) != 0u),
                               m1 = __ballot((m >> EPV_INIT_SHIFT) != 0u), m0 = __ballot((mr >> EPV_INIT_SHIFT) != 0u);
      if (lane < 8) {
        const unsigned long long x = mf & ((lane & 4) ? m2 : ~m2) & ((lane & 2) ? m1 : ~m1) & ((lane & 1) ? m0 : ~m0);
        s_cnt[(b - b_lo) * 8u + (uint32_t)lane] += (uint32_t)__popcll(x);     // this lane's own counter
      }
      const unsigned long long ms = __ballot(slow);
      if (slow) {
        const uint32_t slot = (tail + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull))) & 127u;
        s_ring[slot] = t | ((b - b_lo) << 8);
        s_ring_lm[slot] = ml | (m << 16);
        s_ring_r[slot] = (epv_meta_t)mr;
      }
      tail += (uint32_t)__popcll(ms);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (tail - head >= 64u) {
        merge_item((head + t) & 127u);
        head += 64u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (t < tail - head) merge_item((head + t) & 127u);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint64_t row = (site0 - block0 * 256u) / 256u;      // the 256-site block this wave belongs to
  for (uint32_t i = t; i < (b_hi - b_lo) * 16u; i += 64u) {
    const uint32_t bl = i >> 4, c = i & 15u, b = b_lo + bl;
    unsigned long long v = s_acc[i];
    if (c >= 8u) v += (unsigned long long)s_cnt[bl * 8u + (c - 8u)] * epv_stat_fix(S.blen[b + 1u] - 0.0, statscale[b + 1u]);
    if (v) atomicAdd(&partial[(row * B + b) * 16u + c], v);
  }
}

// Sums of rows of 64-bit integers: out(r, z, c) = the sum of the G consecutive input rows
// r*G .. r*G+G-1 (rows >= m do not exist; G = 0: all m rows) of column c in slice z.  A block takes 16
// columns and walks the rows 16 at a time (a wave reads four 128-byte pieces per load).  Integer sums
// need no fixed order: any chain of such stages (256-site blocks -> rows of 2^g blocks -> all-gather
// over the GPUs -> total) gives the same bits.
__global__ __launch_bounds__(256) void epv_isum_kernel(const unsigned long long *in, uint64_t m, uint32_t V, uint64_t G,
                                                       uint64_t in_row_stride, uint64_t in_z_stride,
                                                       unsigned long long *out, uint64_t out_row_stride,
                                                       uint64_t out_z_stride) {
  __shared__ unsigned long long s_part[16][17];
  const uint32_t rl = threadIdx.x >> 4, cl = threadIdx.x & 15u;
  const uint32_t c = blockIdx.x * 16u + cl;
  const uint64_t r = blockIdx.y, z = blockIdx.z;
  const uint64_t lo = G ? r * G : 0u;
  uint64_t hi = G ? lo + G : m;
  if (hi > m) hi = m;
  unsigned long long acc = 0ull;
  if (c < V) {
    const unsigned long long *p = in + z * in_z_stride + c;
#pragma unroll 4
    for (uint64_t i = lo + rl; i < hi; i += 16u) acc += p[i * in_row_stride];
  }
  s_part[rl][cl] = acc;
  __syncthreads();
  if (threadIdx.x < 16u && c < V) {
    unsigned long long tot = 0ull;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += s_part[q][threadIdx.x];
    out[r * out_row_stride + z * out_z_stride + c] = tot;
  }
}

// one level of the tree: in[m][V] -> out[ceil(m/256)][V] (V a multiple of 16), each block
// sums 256 consecutive (aligned) entries of one group of 16 value columns (blockIdx.y) in
// balanced order
__global__ __launch_bounds__(256) void epv_tree_reduce_kernel(const double *in, uint64_t m,
                                                              uint32_t V, double *out,
                                                              uint64_t in_stride = 0, uint64_t out_stride = 0) {
  __shared__ double s_part[4][16];
  in += (uint64_t)blockIdx.z * in_stride;     // blockIdx.z: one of several independent reductions
  out += (uint64_t)blockIdx.z * out_stride;   // (the batch sweeps of epv_reduce_blocks)
  const int lane = epv_lane();
  const uint32_t wave = threadIdx.x >> 6;
  const uint64_t idx_in = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint32_t g = blockIdx.y;
  double v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = idx_in < m ? in[idx_in * V + g * 16u + c] : 0.0;
  int idx;
  const double tot = wave_tree_sum16(v, lane, idx);
  if (lane < 16) s_part[wave][idx] = tot;
  __syncthreads();
  if (threadIdx.x < 16)
    out[(uint64_t)blockIdx.x * V + g * 16u + threadIdx.x] =
        (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

// =========================================================================
//  scale_jump_times (ParamEstimation.cpp:369-380): jumps *= scale[b]
// =========================================================================
__global__ __launch_bounds__(256) void epv_scale_kernel(EpvDev S, const double *scale) {
  const uint64_t site = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= S.n) return;
  const uint32_t buf = S.sel[site];
  for (uint32_t b = 0; b < S.B; ++b) {
    const uint64_t plane = (uint64_t)buf * S.B + b;
    const uint32_t nj = S.meta[meta_idx(S, buf, b, site)] & EPV_NJ_MASK;
    double *j = S.jumps + plane * S.C * S.n + site;
    const double sc = scale[b + 1];
    for (uint32_t k = 0; k < nj; ++k) j[(uint64_t)k * S.n] *= sc;
  }
}

// =========================================================================
//  layout conversion between the ABI's node-major CSR form and the device SoA form
// =========================================================================
__global__ __launch_bounds__(256) void epv_scatter_kernel(EpvDev S, const uint8_t *init,
                                                          const uint64_t *offsets,
                                                          const double *jumps_csr) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (uint64_t)S.B * S.n) return;
  const uint64_t b = e / S.n, site = e % S.n;
  const uint64_t o = offsets[e];
  const uint32_t cnt = (uint32_t)(offsets[e + 1] - o);
  S.meta[meta_idx(S, 0u, (uint32_t)b, site)] = (epv_meta_t)((init[e] ? (1u << EPV_INIT_SHIFT) : 0u) | cnt);  // buffer 0
  double *j = S.jumps + b * S.C * S.n + site;
  for (uint32_t k = 0; k < cnt; ++k) j[(uint64_t)k * S.n] = jumps_csr[o + k];
  if (b == 0) { S.sel[site] = 0; S.tri[site] = 0.0; }
}

// counts[e] = number of jumps of the CURRENT path of entry e; init_out[e] likewise
__global__ __launch_bounds__(256) void epv_count_kernel(EpvDev S, uint8_t *init_out,
                                                        uint64_t *counts) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (uint64_t)S.B * S.n) return;
  const uint64_t b = e / S.n, site = e % S.n;
  const epv_meta_t m = S.meta[meta_idx(S, S.sel[site], (uint32_t)b, site)];
  init_out[e] = m >> EPV_INIT_SHIFT;
  counts[e] = m & EPV_NJ_MASK;
}

__global__ __launch_bounds__(256) void epv_gather_kernel(EpvDev S, const uint64_t *offsets,
                                                         double *jumps_csr) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (uint64_t)S.B * S.n) return;
  const uint64_t b = e / S.n, site = e % S.n;
  const uint64_t plane = (uint64_t)S.sel[site] * S.B + b;
  const uint32_t cnt = S.meta[meta_idx(S, S.sel[site], (uint32_t)b, site)] & EPV_NJ_MASK;
  const double *j = S.jumps + plane * S.C * S.n + site;
  const uint64_t o = offsets[e];
  for (uint32_t k = 0; k < cnt; ++k) jumps_csr[o + k] = j[(uint64_t)k * S.n];
}

// ------------------------------------------------ halo columns (site-sharded runs)
// packed column = [B meta words (current path)] padded to 8 bytes, [B*C doubles jumps],
// [3 doubles tri(s-1), tri(s), tri(s+1)]
__device__ __forceinline__ uint64_t epv_col_meta_bytes(uint32_t B) {
  return ((uint64_t)B * sizeof(epv_meta_t) + 7u) & ~7ull;
}
__device__ __forceinline__ uint64_t epv_col_bytes(uint32_t B, uint32_t C) {
  return epv_col_meta_bytes(B) + ((uint64_t)B * C + 3u) * 8u;
}
__global__ void epv_pack_columns_kernel(EpvDev S, uint64_t first, uint64_t count,
                                        uint8_t *packed) {
  const uint64_t c = blockIdx.x;
  if (c >= count) return;
  const uint64_t site = first + c;
  const uint64_t cb = epv_col_bytes(S.B, S.C);
  epv_meta_t *col = reinterpret_cast<epv_meta_t *>(packed + c * cb);
  double *dj = reinterpret_cast<double *>(packed + c * cb + epv_col_meta_bytes(S.B));
  const uint32_t buf = S.sel[site];
  for (uint32_t i = threadIdx.x; i < S.B * S.C; i += blockDim.x) {
    const uint32_t b = i / S.C, k = i % S.C;
    const uint64_t plane = (uint64_t)buf * S.B + b;
    const uint32_t nj = S.meta[meta_idx(S, buf, b, site)] & EPV_NJ_MASK;
    dj[i] = k < nj ? S.jumps[(plane * S.C + k) * S.n + site] : 0.0;
  }
  for (uint32_t b = threadIdx.x; b < S.B; b += blockDim.x)
    col[b] = S.meta[meta_idx(S, buf, b, site)];
  if (threadIdx.x < 3) {
    const int64_t s = (int64_t)site + (int64_t)threadIdx.x - 1;
    dj[(uint64_t)S.B * S.C + threadIdx.x] = (s >= 0 && (uint64_t)s < S.n) ? S.tri[s] : 0.0;
  }
}
__global__ void epv_unpack_columns_kernel(EpvDev S, uint64_t first, uint64_t count,
                                          const uint8_t *packed) {
  const uint64_t c = blockIdx.x;
  if (c >= count) return;
  const uint64_t site = first + c;
  const uint64_t cb = epv_col_bytes(S.B, S.C);
  const epv_meta_t *col = reinterpret_cast<const epv_meta_t *>(packed + c * cb);
  const double *dj = reinterpret_cast<const double *>(packed + c * cb + epv_col_meta_bytes(S.B));
  const uint32_t buf = S.sel[site];  // overwrite the current buffer in place
  for (uint32_t i = threadIdx.x; i < S.B * S.C; i += blockDim.x) {
    const uint32_t b = i / S.C, k = i % S.C;
    const uint32_t nj = col[b] & EPV_NJ_MASK;
    if (k < nj) S.jumps[(((uint64_t)buf * S.B + b) * S.C + k) * S.n + site] = dj[i];
  }
  for (uint32_t b = threadIdx.x; b < S.B; b += blockDim.x)
    S.meta[meta_idx(S, buf, b, site)] = col[b];
  if (threadIdx.x < 3) {
    const int64_t s = (int64_t)site + (int64_t)threadIdx.x - 1;
    if (s >= 0 && (uint64_t)s < S.n) S.tri[s] = dj[(uint64_t)S.B * S.C + threadIdx.x];
  }
}

// site-parallel forward simulation (epievo_sim's process by thinning)
#include "epv_forward.h"

#endif
