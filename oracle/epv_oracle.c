/* epv_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of epievo's MCEM inner
 * loop (the per-site Metropolis-Hastings end-conditioned path sampler).
 *
 * This file is the parity oracle for the MI355X build.  It is NOT part of the
 * product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  It restates, in plain C over flat arrays,
 * what these reference functions compute (file:line under /root/reference):
 *
 *   collect_segment_info              src/libepievo/Segment.cpp:35-79
 *   continuous_time_trans_prob_mat    src/libepievo/ContinuousTimeMarkovModel.cpp:143-161
 *   TwoStateCTMarkovModel::get_trans_prob             ...:116-125
 *   pruning / pruning_branch / process_branch_above   src/libepievo/SingleSiteSampler.cpp:80-157
 *   downward_sampling(_branch)                        ...:180-255
 *   forward_sampling / end_cond_sample_forward_rejection  src/libepievo/EndCondSampling.cpp:466-509
 *   sample_trunc_exp / end_cond_sampling_Nielsen          ...:576-617 (parallel rung, state changes)
 *   proposal_prob(_branch)                            SingleSiteSampler.cpp:272-339
 *   path_log_likelihood / root_prior_lh / log_likelihood  ...:263-269,342-391
 *   add_sufficient_statistics                         src/libepievo/Path.cpp:206-301
 *   log_accept_rate                                   SingleSiteSampler.cpp:396-433
 *   Metropolis_Hastings_site / single_iteration / reset / run_mcmc  ...:449-598
 *   get_sufficient_statistics (per branch)            src/libepievo/ParamEstimation.cpp:92-114
 *   scale_jump_times                                  ...:369-380
 *
 * Two modes (the "oracle ladder" of SURVEY.md section 8c):
 *   rung A  reference-schedule: sequential sweep, mt19937 + libstdc++
 *           distribution semantics, glibc exp/log, sequential J/D sums.
 *           Pinned bit-for-bit against the linked reference (oracle/_ref) and
 *           the golden fixtures in tests/golden/.
 *   rung B  parallel-schedule: 3-colour sweep, random-access Philox4x32-10,
 *           orc_exp/orc_log, exact fixed-point (int64) J/D sums, bounded
 *           jump capacity, and the reference's Nielsen sampler instead of
 *           forward rejection for segments that change state (see
 *           ORC_SAMPLER_* below; pinned against the linked function too).
 *           This is the contract the gfx950 kernels match bit-for-bit.
 * Both rungs run the SAME per-site function below; only the random source,
 * the exp/log pair, the visiting order, the reduction order and that sampler
 * choice differ.
 *
 * Path storage: node-major flat arrays, index (node*n_sites + site), node 0
 * (the root) unused -- as in the reference, where paths[site][0] is a dummy.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>

#include "orc_math.h"
#include "orc_rng.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

enum { ORC_RNG_MT = 0, ORC_RNG_PHILOX = 1 };
enum { ORC_MATH_LIBM = 0, ORC_MATH_EPV = 1 };
enum { ORC_SCHED_SEQ = 0, ORC_SCHED_3COLOUR = 1 };
/* J/D reduction: SEQ = the reference's sequential fp64 sums (ParamEstimation.cpp:92-114);
 * EXACT = the parallel rung's order-free integers: J as counts, every dwell time as the
 * fixed-point integer rint(dt * 2^k_b) (k_b per branch, stat_scale_exp below), summed in
 * int64 -- exactly associative, so the result does not depend on the launch shape, the
 * number of GPUs or the order atomics land in. */
enum { ORC_REDUCE_SEQ = 0, ORC_REDUCE_EXACT = 1 };
/* end-conditioned sampler of a segment whose end state differs from its start state:
 *   FORWARD  plain forward rejection, what the reference's hot path calls
 *            (EndCondSampling.cpp:479-509) -- needs ~1/P(a->b) trials, hundreds to 1e5 on a
 *            short branch;
 *   NIELSEN  the reference's own end_cond_sampling_Nielsen (EndCondSampling.cpp:576-617):
 *            first jump from the truncated exponential, then forward sampling; the same
 *            conditional law, but the acceptance probability no longer vanishes with the
 *            segment length.  The parallel rung and the GPU use this one.
 * Segments that keep their state use forward rejection in both (as Nielsen's routine does). */
enum { ORC_SAMPLER_FORWARD = 0, ORC_SAMPLER_NIELSEN = 1 };
enum { ORC_PROPOSAL_REFERENCE = 0, ORC_PROPOSAL_TELESCOPED = 1 };

typedef struct {
  uint8_t init;
  uint32_t n, cap;
  double *t;
} orc_path;

typedef struct {
  int K, cap;
  double *len;      /* segment length */
  uint8_t *trip0;   /* context index with middle bit 0 (trip1 = trip0|2) */
  double *p0, *p1;  /* Felsenstein partials at the top of each segment */
} orc_segs;

typedef struct {
  orc_segs *segs;   /* [n_nodes] */
  double *q0, *q1;  /* [n_nodes] */
  orc_path *prop;   /* [n_nodes] proposed path */
  double *trial;    /* jumps of the current rejection trial */
  uint32_t trial_cap;
} orc_scratch;

typedef struct orc_state {
  size_t n_sites;
  size_t g0, n_global; /* site-sharded runs: global index of local site 0, genome length */
  int n_nodes;
  uint32_t *parent, *subtree;
  double *blen;
  double rates[8], log_rates[8], T[4];
  orc_path *paths;
  double *tri_llh;
  int rng_mode, math_mode, schedule, reduce_mode, sampler_mode;
  uint32_t cap; /* max jumps per (site,branch) path; 0 = unbounded */
  orc_mt19937 mt;
  uint64_t seed;
  double (*fexp)(double);
  double (*flog)(double);
  orc_scratch scr;
  orc_scratch *thr_scr; /* per-thread scratch for the OpenMP colour phases */
  int n_thr_scr;
  /* counters */
  uint64_t n_overflow, n_trials, n_draws, n_segments;
  int proposal_mode;       /* ORC_PROPOSAL_* */
  int sample_root;         /* SingleSiteSampler::SAMPLE_ROOT (hard-wired false in the reference, :441) */
  double max_qdiff;        /* max |proposal_prob(old) - proposal_prob(new)| seen (reference arithmetic) */
} orc_state;

static double libm_exp(double x) { return exp(x); }
static double libm_log(double x) { return log(x); }
static double epv_exp_(double x) { return orc_exp(x); }
static double epv_log_(double x) { return orc_log(x); }

/* ------------------------------------------------------------------ paths */
static void path_reserve(orc_path *p, uint32_t need) {
  if (need <= p->cap) return;
  uint32_t c = p->cap ? p->cap * 2 : 4;
  while (c < need) c *= 2;
  p->t = (double *)realloc(p->t, (size_t)c * sizeof(double));
  p->cap = c;
}
static void path_push(orc_path *p, double t) {
  path_reserve(p, p->n + 1);
  p->t[p->n++] = t;
}
static inline int path_end_state(const orc_path *p) { return p->init ^ (p->n & 1u); }
static inline orc_path *PATH(const orc_state *st, int node, size_t site) {
  return &st->paths[(size_t)node * st->n_sites + site];
}

static inline int is_leaf(const orc_state *st, int node) { return st->subtree[node] == 1; }

/* ------------------------------------------------------------- segments */
static void segs_reserve(orc_segs *s, int need) {
  if (need <= s->cap) return;
  int c = s->cap ? s->cap * 2 : 8;
  while (c < need) c *= 2;
  s->len = (double *)realloc(s->len, (size_t)c * sizeof(double));
  s->trip0 = (uint8_t *)realloc(s->trip0, (size_t)c);
  s->p0 = (double *)realloc(s->p0, (size_t)c * sizeof(double));
  s->p1 = (double *)realloc(s->p1, (size_t)c * sizeof(double));
  s->cap = c;
}

/* Segment.cpp:35-79: 2-way merge of the neighbours' jump lists; a left jump
 * is taken only when strictly earlier than the right one. */
static void collect_segments(const orc_path *l, const orc_path *r, double tot_time,
                             orc_segs *s) {
  segs_reserve(s, (int)(l->n + r->n + 1));
  int K = 0;
  uint8_t trip0 = (uint8_t)(4 * l->init + r->init);
  double prev = 0.0;
  uint32_t i = 0, j = 0;
  while (i < l->n && j < r->n) {
    if (l->t[i] < r->t[j]) {
      s->len[K] = l->t[i] - prev; s->trip0[K] = trip0; ++K;
      trip0 ^= 4; prev = l->t[i++];
    } else {
      s->len[K] = r->t[j] - prev; s->trip0[K] = trip0; ++K;
      trip0 ^= 1; prev = r->t[j++];
    }
  }
  for (; i < l->n; ++i) {
    s->len[K] = l->t[i] - prev; s->trip0[K] = trip0; ++K;
    trip0 ^= 4; prev = l->t[i];
  }
  for (; j < r->n; ++j) {
    s->len[K] = r->t[j] - prev; s->trip0[K] = trip0; ++K;
    trip0 ^= 1; prev = r->t[j];
  }
  s->len[K] = tot_time - prev; s->trip0[K] = trip0; ++K;
  s->K = K;
}

/* --------------------------------------------------- 2-state CTMC pieces */
/* ContinuousTimeMarkovModel.cpp:143-161 (h = 1.0/exp(.)) */
static inline void trans_prob_mat(const orc_state *st, double r0, double r1, double t,
                                  double P[4]) {
  const double h = 1.0 / st->fexp(t * (r0 + r1));
  const double denom = r0 + r1;
  P[0] = (r0 * h + r1) / denom;
  P[1] = 1.0 - P[0];
  P[3] = (r0 + r1 * h) / denom;
  P[2] = 1.0 - P[3];
}
/* ContinuousTimeMarkovModel.cpp:116-125 (h = exp(-.)) */
static inline double get_trans_prob(const orc_state *st, double r0, double r1, double t,
                                    int a, int b) {
  const double h = st->fexp(-t * (r0 + r1));
  const double denom = r0 + r1;
  const double prob = (a ? r0 + r1 * h : r0 * h + r1) / denom;
  return (a == b) ? prob : 1.0 - prob;
}

/* ------------------------------------------------------------- pruning */
/* SingleSiteSampler.cpp:80-157 */
static void pruning(const orc_state *st, size_t site, orc_scratch *sc) {
  for (int node = st->n_nodes - 1; node >= 0; --node) {
    double q0 = 1.0, q1 = 1.0;
    if (is_leaf(st, node)) {
      const int leaf_state = path_end_state(PATH(st, node, site));
      q0 = leaf_state ? 0.0 : 1.0;
      q1 = leaf_state ? 1.0 : 0.0;
    } else {
      /* children of `node` in pre-order: node+1, then skip subtrees */
      for (uint32_t ch = 1; ch < st->subtree[node]; ch += st->subtree[node + ch]) {
        const orc_segs *cs = &sc->segs[node + ch];
        q0 *= cs->p0[0];
        q1 *= cs->p1[0];
      }
    }
    sc->q0[node] = q0;
    sc->q1[node] = q1;
    if (node == 0) continue;
    orc_segs *s = &sc->segs[node];
    double n0 = q0, n1 = q1;
    for (int k = s->K - 1; k >= 0; --k) {
      double P[4];
      trans_prob_mat(st, st->rates[s->trip0[k]], st->rates[s->trip0[k] | 2], s->len[k], P);
      const double a = P[0] * n0 + P[1] * n1;
      const double b = P[2] * n0 + P[3] * n1;
      s->p0[k] = a; s->p1[k] = b;
      n0 = a; n1 = b;
    }
  }
}

/* ------------------------------------------------------ random sources */
typedef struct {
  orc_state *st;
  uint32_t site, sweep;
  uint32_t b, k, t, d;  /* current trial address and draw index */
  double blk[2];
} orc_rng;

static inline double rng_segment_uniform(orc_rng *g, uint32_t b, uint32_t k) {
  if (g->st->rng_mode == ORC_RNG_MT) return orc_mt_canonical(&g->st->mt);
  double d[2];
  orc_keyed_block(g->st->seed, g->site, g->sweep, b, k, 0, 0, d);
  return d[0];
}
/* the root state's uniform (SingleSiteSampler.cpp:247, SAMPLE_ROOT only): the FIRST draw of a site
 * update in the reference's stream; the unused half of the accept uniform's block in the keyed one */
static inline double rng_root_uniform(orc_rng *g) {
  if (g->st->rng_mode == ORC_RNG_MT) return orc_mt_canonical(&g->st->mt);
  double d[2];
  orc_keyed_block(g->st->seed, g->site, g->sweep, 0, 0, 0, 0, d);
  return d[1];
}
static inline double rng_accept_uniform(orc_rng *g) {
  if (g->st->rng_mode == ORC_RNG_MT) return orc_mt_canonical(&g->st->mt);
  double d[2];
  orc_keyed_block(g->st->seed, g->site, g->sweep, 0, 0, 0, 0, d);
  return d[0];
}
static inline void rng_trial_begin(orc_rng *g, uint32_t b, uint32_t k, uint32_t t) {
  g->b = b; g->k = k; g->t = t; g->d = 0;
}
static inline double rng_trial_canonical(orc_rng *g) {
  if (g->st->rng_mode == ORC_RNG_MT) return orc_mt_canonical(&g->st->mt);
  double blk[2];
  const uint32_t d = g->d++;
  if (d == 0) {
    /* first draw: trial 1 shares the segment's block (its d1); trials 2m, 2m+1 share
     * block (t = m, blk = 255) */
    if (g->t == 1) {
      orc_keyed_block(g->st->seed, g->site, g->sweep, g->b, g->k, 0, 0, blk);
      return blk[1];
    }
    orc_keyed_block(g->st->seed, g->site, g->sweep, g->b, g->k, g->t >> 1, ORC_FIRST_DRAW_BLOCK, blk);
    return blk[g->t & 1u];
  }
  if (((d - 1u) & 1u) == 0)
    orc_keyed_block(g->st->seed, g->site, g->sweep, g->b, g->k, g->t, (d - 1u) >> 1, g->blk);
  return g->blk[(d - 1u) & 1u];
}

/* ------------------------------------------------ end-conditioned sampling */
/* EndCondSampling.cpp:466-509.  Returns 0 ok, 1 capacity overflow.
 * Appends accepted jump times (offset by start_time) to `out`. */
static int forward_rejection(orc_state *st, orc_scratch *sc, orc_rng *g,
                             uint32_t b, uint32_t k, double rate0, double rate1,
                             int start, int end, double T, double start_time,
                             orc_path *out) {
  const uint32_t cap = st->cap;
  const uint32_t room = cap ? cap - out->n : 0xffffffffu; /* jumps still storable */
  const int nielsen = st->sampler_mode == ORC_SAMPLER_NIELSEN && start != end;
  /* sample_trunc_exp (EndCondSampling.cpp:577-580): 1 - exp(-lambda T), the same every trial */
  const double rate_a = start ? rate1 : rate0;
  const double trunc = nielsen ? 1.0 - st->fexp(-rate_a * T) : 0.0;
  uint32_t t = 1;
  for (;;) {
    rng_trial_begin(g, b, k, t);
    uint32_t nj = 0;
    int a = start;
    double tau = 0.0;
    int overflow = 0, bad = 0;
#ifndef _OPENMP
    ++st->n_trials;
#endif
    if (nielsen) {
      /* EndCondSampling.cpp:606-610: the first jump, then the other state */
      const double u = rng_trial_canonical(g);
#ifndef _OPENMP
      ++st->n_draws;
#endif
      tau = -st->flog(1.0 - u * trunc) / rate_a;
      /* mathematically tau < T; a draw within rounding of 1 could land on T: redraw (the
       * reference has no such guard; the event has probability ~1e-13 per draw) */
      if (!(tau < T)) bad = 1;
      else if (cap && nj >= room) overflow = 1;
      else {
        a ^= 1;
        if (nj >= sc->trial_cap) {
          sc->trial_cap = sc->trial_cap ? sc->trial_cap * 2 : 16;
          sc->trial = (double *)realloc(sc->trial, sc->trial_cap * sizeof(double));
        }
        sc->trial[nj++] = tau;
      }
    }
    while (!bad && !overflow) {
      const double u = rng_trial_canonical(g);
#ifndef _OPENMP
      ++st->n_draws;
#endif
      tau += -st->flog(1.0 - u) / (a ? rate1 : rate0);
      if (!(tau < T)) break;
      if (cap && nj >= room) { overflow = 1; break; }
      a ^= 1;
      if (nj >= sc->trial_cap) {
        sc->trial_cap = sc->trial_cap ? sc->trial_cap * 2 : 16;
        sc->trial = (double *)realloc(sc->trial, sc->trial_cap * sizeof(double));
      }
      sc->trial[nj++] = tau;
    }
    if (overflow) return 1;
    if (!bad && a == end) {
      for (uint32_t i = 0; i < nj; ++i) path_push(out, sc->trial[i] + start_time);
      return 0;
    }
    ++t; /* the reference caps at 1e10 trials (EndCondSampling.cpp:52); unreachable */
  }
}

/* root_post_prob0, SingleSiteSampler.cpp:167-176: posterior probability of state 0 at the root given
 * the neighbours' root states and the data below (q of node 0) */
static double root_post_prob0(const orc_state *st, size_t site, const orc_scratch *sc) {
  const int l = PATH(st, 1, site - 1)->init, r = PATH(st, 1, site + 1)->init;
  const double p0 = (st->T[2 * l + 0] * st->T[2 * 0 + r]) * sc->q0[0];
  const double p1 = (st->T[2 * l + 1] * st->T[2 * 1 + r]) * sc->q1[0];
  return p0 / (p0 + p1);
}

/* SingleSiteSampler.cpp:180-255.  Returns 1 if the proposal overflowed. */
static int downward_sampling(orc_state *st, size_t site, orc_scratch *sc, orc_rng *g,
                             double *log_prob_out) {
  double log_prob = 0.0;
  int overflow = 0;
  int root_state = PATH(st, 1, site)->init; /* SAMPLE_ROOT == false (:246) */
  if (st->sample_root) {                    /* :246-249 */
    const double root_p0 = root_post_prob0(st, site, sc);
    root_state = rng_root_uniform(g) > root_p0;
    log_prob = root_state ? st->flog(1.0 - root_p0) : st->flog(root_p0);
  }
  sc->prop[0].init = (uint8_t)root_state;
  sc->prop[0].n = 0;
  for (int node = 1; node < st->n_nodes; ++node) {
    orc_path *pp = &sc->prop[node];
    const int start_state = path_end_state(&sc->prop[st->parent[node]]);
    pp->init = (uint8_t)start_state;
    pp->n = 0;
    const orc_segs *s = &sc->segs[node];
    int prev = start_state;
    double time_passed = 0.0;
    for (int i = 0; i < s->K; ++i) {
      const double r0 = st->rates[s->trip0[i]], r1 = st->rates[s->trip0[i] | 2];
      const double PT0 = get_trans_prob(st, r0, r1, s->len[i], prev, 0);
      const double nxt0 = (i == s->K - 1) ? sc->q0[node] : s->p0[i + 1];
      const double p0 = PT0 * nxt0 / (prev ? s->p1[i] : s->p0[i]);
      const double u = rng_segment_uniform(g, (uint32_t)node, (uint32_t)i);
      const int sampled = (u > p0);
      const int ref_q = st->proposal_mode == ORC_PROPOSAL_REFERENCE;
      if (ref_q) log_prob += (sampled == 0) ? st->flog(p0) : st->flog(1.0 - p0);
#ifndef _OPENMP
      ++st->n_segments;
#endif
      if (!overflow)
        overflow = forward_rejection(st, sc, g, (uint32_t)node, (uint32_t)i, r0, r1, prev,
                                     sampled, s->len[i], time_passed, pp);
      if (ref_q) log_prob -= st->flog(get_trans_prob(st, r0, r1, s->len[i], prev, sampled));
      time_passed += s->len[i];
      prev = sampled;
    }
    if (overflow) {
      /* keep the end state consistent for the children: parity of the jump
       * count must equal start^prev.  The proposal is rejected anyway. */
      pp->n = (uint32_t)((start_state ^ prev) & 1);
    }
  }
  *log_prob_out = log_prob;
  return overflow;
}

/* SingleSiteSampler.cpp:272-339 */
static double proposal_prob(const orc_state *st, size_t site, const orc_scratch *sc) {
  double log_prob = 0.0;
  if (st->sample_root) {   /* :325-329 */
    const double root_p0 = root_post_prob0(st, site, sc);
    log_prob += PATH(st, 1, site)->init ? st->flog(1.0 - root_p0) : st->flog(root_p0);
  }
  for (int node = 1; node < st->n_nodes; ++node) {
    const orc_segs *s = &sc->segs[node];
    const orc_path *path = PATH(st, node, site);
    int start_state = path->init, end_state = path->init;
    double end_time = 0.0;
    uint32_t start_jump = 0, end_jump = 0;
    double lp = 0.0;
    for (int i = 0; i < s->K; ++i) {
      end_time += s->len[i];
      while (end_jump < path->n && path->t[end_jump] < end_time) ++end_jump;
      if ((end_jump - start_jump) % 2 == 1) end_state ^= 1;
      const double r0 = st->rates[s->trip0[i]], r1 = st->rates[s->trip0[i] | 2];
      const double PT0 = get_trans_prob(st, r0, r1, s->len[i], start_state, 0);
      lp -= st->flog(get_trans_prob(st, r0, r1, s->len[i], start_state, end_state));
      const double nxt0 = (i == s->K - 1) ? sc->q0[node] : s->p0[i + 1];
      const double p0 = PT0 / (start_state ? s->p1[i] : s->p0[i]) * nxt0;
      lp += (end_state == 0) ? st->flog(p0) : st->flog(1.0 - p0);
      start_jump = end_jump;
      start_state = end_state;
    }
    log_prob += lp;
  }
  return log_prob;
}

/* ------------------------------------------------ sufficient statistics */
/* Path.cpp:206-301 as one 3-way merge with +inf sentinels.  Tie rules:
 * left only if strictly below min(mid,right); else mid only if strictly
 * below right; else right. */
static void add_suff_stats(const orc_path *l, const orc_path *m, const orc_path *r,
                           double tot_time, double J[8], double D[8]) {
  int trip = 4 * l->init + 2 * m->init + r->init;
  double prev = 0.0;
  uint32_t i = 0, j = 0, k = 0;
  for (;;) {
    const double tl = i < l->n ? l->t[i] : INFINITY;
    const double tm = j < m->n ? m->t[j] : INFINITY;
    const double tr = k < r->n ? r->t[k] : INFINITY;
    if (i >= l->n && j >= m->n && k >= r->n) break;
    if (tl < (tm < tr ? tm : tr)) {
      D[trip] += tl - prev; prev = tl; trip ^= 4; ++i;
    } else if (tm < tr) {
      D[trip] += tm - prev; J[trip] += 1.0; prev = tm; trip ^= 2; ++j;
    } else {
      D[trip] += tr - prev; prev = tr; trip ^= 1; ++k;
    }
  }
  D[trip] += tot_time - prev;
}

/* SingleSiteSampler.cpp:263-269,342-391.  The root prior is added un-logged
 * (reference quirk). */
static double path_llh(const orc_state *st, const orc_path *const *l,
                       const orc_path *const *m, const orc_path *const *r) {
  double J[8] = {0, 0, 0, 0, 0, 0, 0, 0}, D[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double llh = st->T[2 * l[1]->init + m[1]->init] * st->T[2 * m[1]->init + r[1]->init];
  for (int b = 1; b < st->n_nodes; ++b) add_suff_stats(l[b], m[b], r[b], st->blen[b], J, D);
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += J[i] * st->log_rates[i] - D[i] * st->rates[i];
  llh += s;
  return llh;
}

#define ORC_MAX_NODES 4096
static double site_triple_llh(const orc_state *st, size_t c, const orc_path *prop, int which) {
  /* llh of the triple centred at c; `prop` (if non-NULL) replaces site
   * (c + which) where which in {-1,0,+1}. */
  const orc_path *l[ORC_MAX_NODES], *m[ORC_MAX_NODES], *r[ORC_MAX_NODES];
  for (int b = 1; b < st->n_nodes; ++b) {
    l[b] = (prop && which == -1) ? &prop[b] : PATH(st, b, c - 1);
    m[b] = (prop && which == 0) ? &prop[b] : PATH(st, b, c);
    r[b] = (prop && which == 1) ? &prop[b] : PATH(st, b, c + 1);
  }
  return path_llh(st, l, m, r);
}

/* ------------------------------------------------------------ MH update */
/* SingleSiteSampler.cpp:482-536 (+ log_accept_rate :396-433) */
static int mh_site(orc_state *st, size_t site, uint32_t sweep, orc_scratch *sc) {
  for (int node = 1; node < st->n_nodes; ++node)
    collect_segments(PATH(st, node, site - 1), PATH(st, node, site + 1), st->blen[node],
                     &sc->segs[node]);
  pruning(st, site, sc);

  orc_rng g;
  g.st = st; g.site = (uint32_t)(st->g0 + site); g.sweep = sweep;
  double proposal_log_prob = 0.0;
  const int overflow = downward_sampling(st, site, sc, &g, &proposal_log_prob);

  double llh_l = st->tri_llh[site - 1];
  double llh_m = st->tri_llh[site];
  double llh_r = st->tri_llh[site + 1];

  /* q(old)/q(new), SingleSiteSampler.cpp:503-507.  With SAMPLE_ROOT false (hard-wired, :441) the
   * two log-probabilities are the SAME number in exact arithmetic: per segment the reference
   * accumulates log P(end | start, data) - log PT(start -> end) = log(p[k+1][end] / p[k][start]),
   * which telescopes along a branch to log(q[end] / p[0][start]) and over the tree (q of a node is
   * the product of its children's p[0]; a leaf's q is 1 at the observed state) to
   * -log prod_{children c of the root} p_c[0][root state]: the normalising constant of the
   * proposal, which depends on the neighbours and the leaf data but not on the path.  What the
   * reference evaluates is therefore rounding noise (max_qdiff records it, ~1e-15).
   * ORC_PROPOSAL_REFERENCE keeps that arithmetic; ORC_PROPOSAL_TELESCOPED uses the exact 0. */
  double llr = 0.0;
  if (st->proposal_mode == ORC_PROPOSAL_REFERENCE) {
    const double orig_proposal = proposal_prob(st, site, sc);
    llr = orig_proposal - proposal_log_prob;
    const double a = fabs(llr);
#ifdef _OPENMP
#pragma omp critical(orc_qdiff)
#endif
    if (a > st->max_qdiff) st->max_qdiff = a;
  }
  const double llh_l_orig = llh_l, llh_r_orig = llh_r;
  if (!overflow) {
    if (st->g0 + site > 1) llh_l = site_triple_llh(st, site - 1, sc->prop, 1);
    llh_m = site_triple_llh(st, site, sc->prop, 0);
    if (st->g0 + site < st->n_global - 2) llh_r = site_triple_llh(st, site + 1, sc->prop, -1);
  }
  llr += (llh_l + llh_r - llh_l_orig - llh_r_orig);

  const double u = rng_accept_uniform(&g);
  int accepted = 0;
  if (llr >= 0 || u < st->fexp(llr)) accepted = 1;
  if (overflow) {
    accepted = 0;
#ifdef _OPENMP
#pragma omp atomic
#endif
    ++st->n_overflow;
  }

  if (accepted) {
    for (int b = 1; b < st->n_nodes; ++b) {
      orc_path *dst = PATH(st, b, site), *src = &sc->prop[b];
      orc_path tmp = *dst; *dst = *src; *src = tmp; /* std::swap, :529 */
    }
    st->tri_llh[site - 1] = llh_l;
    st->tri_llh[site] = llh_m;
    st->tri_llh[site + 1] = llh_r;
  }
  return accepted;
}

/* ------------------------------------------------------------- plumbing */
static void scratch_init(orc_scratch *sc, int n_nodes) {
  memset(sc, 0, sizeof(*sc));
  sc->segs = (orc_segs *)calloc((size_t)n_nodes, sizeof(orc_segs));
  sc->q0 = (double *)calloc((size_t)n_nodes, sizeof(double));
  sc->q1 = (double *)calloc((size_t)n_nodes, sizeof(double));
  sc->prop = (orc_path *)calloc((size_t)n_nodes, sizeof(orc_path));
}
static void scratch_free(orc_scratch *sc, int n_nodes) {
  for (int i = 0; i < n_nodes; ++i) {
    free(sc->segs[i].len); free(sc->segs[i].trip0); free(sc->segs[i].p0); free(sc->segs[i].p1);
    free(sc->prop[i].t);
  }
  free(sc->segs); free(sc->q0); free(sc->q1); free(sc->prop); free(sc->trial);
}

static void set_math(orc_state *st) {
  st->fexp = st->math_mode == ORC_MATH_EPV ? epv_exp_ : libm_exp;
  st->flog = st->math_mode == ORC_MATH_EPV ? epv_log_ : libm_log;
}

/* init[(b-1)*n + site], offsets[(b-1)*n + site] (size B*n+1), b = 1..n_nodes-1 */
ORC_API orc_state *orc_create(uint64_t n_sites, int n_nodes, const uint32_t *parent,
                              const uint32_t *subtree, const double *branches,
                              const double *rates, const double *T, const uint8_t *init,
                              const uint64_t *offsets, const double *jumps) {
  if (n_nodes < 2 || n_nodes > ORC_MAX_NODES || n_sites < 3) return NULL;
  orc_state *st = (orc_state *)calloc(1, sizeof(orc_state));
  st->n_sites = n_sites;
  st->g0 = 0;
  st->n_global = n_sites;
  st->n_nodes = n_nodes;
  st->parent = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n_nodes);
  st->subtree = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n_nodes);
  st->blen = (double *)malloc(sizeof(double) * (size_t)n_nodes);
  memcpy(st->parent, parent, sizeof(uint32_t) * (size_t)n_nodes);
  memcpy(st->subtree, subtree, sizeof(uint32_t) * (size_t)n_nodes);
  memcpy(st->blen, branches, sizeof(double) * (size_t)n_nodes);
  memcpy(st->rates, rates, sizeof(double) * 8);
  memcpy(st->T, T, sizeof(double) * 4);
  st->paths = (orc_path *)calloc((size_t)n_nodes * n_sites, sizeof(orc_path));
  st->tri_llh = (double *)calloc(n_sites, sizeof(double));
  for (int b = 1; b < n_nodes; ++b)
    for (size_t s = 0; s < n_sites; ++s) {
      const size_t idx = (size_t)(b - 1) * n_sites + s;
      orc_path *p = PATH(st, b, s);
      p->init = init[idx];
      const uint32_t cnt = (uint32_t)(offsets[idx + 1] - offsets[idx]);
      if (cnt) {
        path_reserve(p, cnt);
        memcpy(p->t, jumps + offsets[idx], cnt * sizeof(double));
        p->n = cnt;
      }
    }
  st->rng_mode = ORC_RNG_MT;
  st->math_mode = ORC_MATH_LIBM;
  st->schedule = ORC_SCHED_SEQ;
  st->reduce_mode = ORC_REDUCE_SEQ;
  set_math(st);
  orc_mt_seed(&st->mt, 5489u);
  scratch_init(&st->scr, n_nodes);
  return st;
}

ORC_API void orc_destroy(orc_state *st) {
  if (!st) return;
  for (size_t i = 0; i < (size_t)st->n_nodes * st->n_sites; ++i) free(st->paths[i].t);
  scratch_free(&st->scr, st->n_nodes);
  for (int t = 0; t < st->n_thr_scr; ++t) scratch_free(&st->thr_scr[t], st->n_nodes);
  free(st->thr_scr);
  free(st->paths); free(st->tri_llh); free(st->parent); free(st->subtree); free(st->blen);
  free(st);
}

ORC_API void orc_set_modes(orc_state *st, int rng_mode, int math_mode, int schedule,
                           int reduce_mode, uint32_t cap) {
  st->rng_mode = rng_mode; st->math_mode = math_mode; st->schedule = schedule;
  st->reduce_mode = reduce_mode; st->cap = cap;
  /* default pairing: the reference-schedule rung keeps the reference's forward rejection,
   * the parallel (Philox) rung uses the Nielsen sampler the GPU implements */
  st->sampler_mode = rng_mode == ORC_RNG_PHILOX ? ORC_SAMPLER_NIELSEN : ORC_SAMPLER_FORWARD;
  /* ... and the exact (telescoped) proposal ratio; the reference rung keeps the reference's sums */
  st->proposal_mode = rng_mode == ORC_RNG_PHILOX ? ORC_PROPOSAL_TELESCOPED : ORC_PROPOSAL_REFERENCE;
  set_math(st);
}
ORC_API void orc_set_sampler(orc_state *st, int sampler_mode) { st->sampler_mode = sampler_mode; }
ORC_API void orc_set_proposal_mode(orc_state *st, int mode) { st->proposal_mode = mode; }
/* SAMPLE_ROOT: the proposal then includes the root state, whose posterior does not cancel against
 * the (un-logged) root prior of the likelihood: the ratio must be evaluated, never elided */
ORC_API void orc_set_sample_root(orc_state *st, int on) {
  st->sample_root = on;
  if (on) st->proposal_mode = ORC_PROPOSAL_REFERENCE;
}
ORC_API double orc_get_max_qdiff(const orc_state *st) { return st->max_qdiff; }
ORC_API void orc_seed_mt(orc_state *st, uint64_t seed) { orc_mt_seed(&st->mt, (uint32_t)seed); }
ORC_API void orc_seed_philox(orc_state *st, uint64_t seed) { st->seed = seed; }
ORC_API void orc_set_model(orc_state *st, const double *rates, const double *T) {
  memcpy(st->rates, rates, sizeof(double) * 8);
  memcpy(st->T, T, sizeof(double) * 4);
}

/* SingleSiteSampler.cpp:449-475.  log(rates) always uses glibc: in the build
 * it is computed on the host and shipped to the device as constants. */
ORC_API void orc_reset(orc_state *st) {
  for (int i = 0; i < 8; ++i) st->log_rates[i] = log(st->rates[i]);
  for (size_t s = 1; s + 1 < st->n_sites; ++s) st->tri_llh[s] = site_triple_llh(st, s, NULL, 0);
}
ORC_API void orc_get_tri_llh(const orc_state *st, double *out) {
  memcpy(out, st->tri_llh, st->n_sites * sizeof(double));
}

/* SingleSiteSampler.cpp:538-548 (sequential) or the 3-colour schedule */
ORC_API uint64_t orc_sweep(orc_state *st, uint32_t sweep) {
  uint64_t n_acc = 0;
  const size_t n = st->n_sites;
  if (st->schedule == ORC_SCHED_SEQ) {
    for (size_t s = 1; s + 1 < n; ++s) n_acc += (uint64_t)mh_site(st, s, sweep, &st->scr);
  } else {
    /* sites of one colour are independent and write-disjoint, so a colour phase may run
     * on all host cores (the "fair" CPU baseline of BASELINE.md section 3 item 2);
     * counter-based draws make the result identical for any thread count */
    int n_thr = 1;
#ifdef _OPENMP
    n_thr = omp_get_max_threads();
#endif
    if (st->rng_mode != ORC_RNG_PHILOX) n_thr = 1;
    if (n_thr > 1 && st->n_thr_scr < n_thr) {
      st->thr_scr = (orc_scratch *)realloc(st->thr_scr, sizeof(orc_scratch) * (size_t)n_thr);
      for (int t = st->n_thr_scr; t < n_thr; ++t) scratch_init(&st->thr_scr[t], st->n_nodes);
      st->n_thr_scr = n_thr;
    }
    for (size_t c = 0; c < 3; ++c) {
      const size_t s_first = 1 + ((c + 3 - (st->g0 + 1) % 3) % 3);
      if (n_thr == 1) {
        for (size_t s = s_first; s + 1 < n; s += 3) n_acc += (uint64_t)mh_site(st, s, sweep, &st->scr);
      } else {
        long long acc = 0;
        const long long cnt = (s_first + 1 < n) ? (long long)((n - 2 - s_first) / 3 + 1) : 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 4096) reduction(+ : acc)
#endif
        for (long long i = 0; i < cnt; ++i) {
          int t = 0;
#ifdef _OPENMP
          t = omp_get_thread_num();
#endif
          acc += mh_site(st, s_first + 3 * (size_t)i, sweep, &st->thr_scr[t]);
        }
        n_acc += (uint64_t)acc;
      }
    }
  }
  return n_acc;
}

/* ---- site-sharded runs (the multi-GPU driver's logic is tested against these) */
ORC_API void orc_set_shard(orc_state *st, uint64_t g0, uint64_t n_global) {
  st->g0 = g0; st->n_global = n_global;
}
/* one colour phase over local sites [first,last]; counts accepts inside [own_first,own_last] */
ORC_API uint64_t orc_sweep_phase(orc_state *st, uint32_t colour, uint32_t sweep, uint64_t first,
                                 uint64_t last, uint64_t own_first, uint64_t own_last) {
  uint64_t n_acc = 0;
  for (size_t s = first; s <= last; ++s)
    if ((st->g0 + s) % 3 == colour) {
      const int a = mh_site(st, s, sweep, &st->scr);
      if (a && s >= own_first && s <= own_last) ++n_acc;
    }
  return n_acc;
}
/* overwrite the current path column of one site (halo refresh) */
ORC_API void orc_set_site(orc_state *st, uint64_t site, const uint8_t *init, const uint32_t *cnt,
                          const double *jumps) {
  for (int b = 1; b < st->n_nodes; ++b) {
    orc_path *p = PATH(st, b, site);
    p->init = init[b - 1];
    p->n = 0;
    for (uint32_t j = 0; j < cnt[b - 1]; ++j) path_push(p, *jumps++);
  }
}
ORC_API void orc_set_tri(orc_state *st, uint64_t site, double v) { st->tri_llh[site] = v; }

/* one MH update of a single site (epievo_sim_pairwise.cpp:267-273 calls
 * Metropolis_Hastings_site per site itself) */
ORC_API int orc_mh_site(orc_state *st, uint64_t site, uint32_t sweep) {
  return mh_site(st, site, sweep, &st->scr);
}

/* ParamEstimation.cpp:92-114.  J/D laid out [(b-1)*8 + ctx]. */
static void site_contrib(const orc_state *st, int b, size_t s, double J[8], double D[8]) {
  add_suff_stats(PATH(st, b, s - 1), PATH(st, b, s), PATH(st, b, s + 1), st->blen[b], J, D);
}
static size_t g_first = 0, g_last = (size_t)-1; /* owned range for the statistics */

/* ---- exact statistics of the parallel rung.  The dwell times of branch b are summed as
 * integers q = rint(dt * 2^k_b) with
 *     k_b = min(61 - e(n_global * T_b), 50 - e(T_b)),   e(x) = the frexp exponent (x < 2^e):
 * the sum over a whole genome stays below 2^61 and a single term below 2^50 (so that the
 * device can round with one add of 1.5 * 2^52).  The quantum 2^-k_b is <= 2^-41 of the branch
 * length at n = 1e6: the total is closer to the exact sum than a sequential fp64 sum.  The
 * device gets 2^k_b from the host (epv_abi.hip: stat_scale_exp is restated there). */
static int stat_scale_exp(uint64_t n_global, double T) {
  if (!(T > 0.0) || !isfinite(T)) return 0;
  int e_t = 0, e_nt = 0;
  (void)frexp(T, &e_t);
  (void)frexp((double)n_global * T, &e_nt);
  int k = 61 - e_nt;
  if (50 - e_t < k) k = 50 - e_t;
  if (k > 1000) k = 1000;
  if (k < -1000) k = -1000;
  return k;
}
static int64_t stat_fix(double dt, double scale) { return (int64_t)llrint(dt * scale); }

/* Path.cpp:206-301 once more, into integers (same merge and tie rules as add_suff_stats) */
static void add_suff_stats_exact(const orc_path *l, const orc_path *m, const orc_path *r,
                                 double tot_time, double scale, int64_t J[8], int64_t D[8]) {
  int trip = 4 * l->init + 2 * m->init + r->init;
  double prev = 0.0;
  uint32_t i = 0, j = 0, k = 0;
  for (;;) {
    const double tl = i < l->n ? l->t[i] : INFINITY;
    const double tm = j < m->n ? m->t[j] : INFINITY;
    const double tr = k < r->n ? r->t[k] : INFINITY;
    if (i >= l->n && j >= m->n && k >= r->n) break;
    if (tl < (tm < tr ? tm : tr)) {
      D[trip] += stat_fix(tl - prev, scale); prev = tl; trip ^= 4; ++i;
    } else if (tm < tr) {
      D[trip] += stat_fix(tm - prev, scale); J[trip] += 1; prev = tm; trip ^= 2; ++j;
    } else {
      D[trip] += stat_fix(tr - prev, scale); prev = tr; trip ^= 1; ++k;
    }
  }
  D[trip] += stat_fix(tot_time - prev, scale);
}
/* integer sums over the owned interior sites of [lo, lo + count): out[16] = J[8], D[8] */
static void exact_sum(const orc_state *st, int b, size_t lo, size_t count, int64_t out[16]) {
  for (int i = 0; i < 16; ++i) out[i] = 0;
  const double scale = ldexp(1.0, stat_scale_exp(st->n_global, st->blen[b]));
  for (size_t s = lo; s < lo + count && s + 1 < st->n_sites; ++s)
    if (s >= 1 && s >= g_first && s <= g_last)
      add_suff_stats_exact(PATH(st, b, s - 1), PATH(st, b, s), PATH(st, b, s + 1), st->blen[b], scale, out, out + 8);
}
/* 2^k_b of every node (index 0 unused): what turns the integer D back into time */
ORC_API void orc_stat_scales(const orc_state *st, double *scale) {
  scale[0] = 1.0;
  for (int b = 1; b < st->n_nodes; ++b) scale[b] = ldexp(1.0, stat_scale_exp(st->n_global, st->blen[b]));
}
ORC_API void orc_suffstats(const orc_state *st, double *J, double *D) {
  const int B = st->n_nodes - 1;
  for (int i = 0; i < B * 8; ++i) { J[i] = 0.0; D[i] = 0.0; }
  if (st->reduce_mode == ORC_REDUCE_SEQ) {
    for (size_t s = 1; s + 1 < st->n_sites; ++s)
      if (s >= g_first && s <= g_last)
        for (int b = 1; b <= B; ++b) site_contrib(st, b, s, J + (b - 1) * 8, D + (b - 1) * 8);
  } else {
    for (int b = 1; b <= B; ++b) {
      int64_t out[16];
      exact_sum(st, b, 0, st->n_sites, out);
      const double inv = ldexp(1.0, -stat_scale_exp(st->n_global, st->blen[b]));
      for (int c = 0; c < 8; ++c) {
        J[(b - 1) * 8 + c] = (double)out[c];
        D[(b - 1) * 8 + c] = (double)out[8 + c] * inv;
      }
    }
  }
}

ORC_API void orc_suffstats_range(const orc_state *st, uint64_t first, uint64_t last, double *J,
                                 double *D) {
  g_first = first; g_last = last;
  orc_suffstats(st, J, D);
  g_first = 0; g_last = (size_t)-1;
}

/* Rows of integer sums, for the tests of the multi-GPU statistics stage: row r covers the
 * row_sites local sites starting at first_site + r * row_sites, restricted to the owned range
 * [own_first, own_last]; out[r][b-1][16] int64 (J counts, then fixed-point D). */
ORC_API void orc_suffstats_rows(const orc_state *st, uint64_t first_site, uint64_t row_sites,
                                uint64_t n_rows, uint64_t own_first, uint64_t own_last, int64_t *out) {
  const int B = st->n_nodes - 1;
  g_first = own_first; g_last = own_last;
  for (uint64_t r = 0; r < n_rows; ++r)
    for (int b = 1; b <= B; ++b)
      exact_sum(st, b, first_site + r * row_sites, row_sites, out + (r * (uint64_t)B + (uint64_t)(b - 1)) * 16);
  g_first = 0; g_last = (size_t)-1;
}

/* SingleSiteSampler.cpp:550-598.  J/D are batch averages on return. */
ORC_API void orc_run_mcmc(orc_state *st, uint64_t burn_in, uint64_t batch,
                          uint32_t sweep_base, double *J, double *D,
                          uint64_t *n_accepted, double *acc_rate) {
  const int B = st->n_nodes - 1;
  uint32_t sweep = sweep_base;
  for (uint64_t i = 0; i < burn_in; ++i) orc_sweep(st, sweep++);
  for (int i = 0; i < B * 8; ++i) { J[i] = 0.0; D[i] = 0.0; }
  double *J1 = (double *)malloc(sizeof(double) * (size_t)B * 8);
  double *D1 = (double *)malloc(sizeof(double) * (size_t)B * 8);
  uint64_t n_acc = 0;
  for (uint64_t i = 0; i < batch; ++i) {
    n_acc += orc_sweep(st, sweep++);
    orc_suffstats(st, J1, D1);
    for (int k = 0; k < B * 8; ++k) { J[k] += J1[k]; D[k] += D1[k]; }
  }
  for (int k = 0; k < B * 8; ++k) { J[k] /= (double)batch; D[k] /= (double)batch; }
  free(J1); free(D1);
  if (n_accepted) *n_accepted = n_acc;
  if (acc_rate) *acc_rate = (double)n_acc / (double)(batch * (st->n_sites - 2));
}

/* ParamEstimation.cpp:369-380 */
ORC_API void orc_scale_jump_times(orc_state *st, const double *new_branches) {
  for (size_t s = 0; s < st->n_sites; ++s)
    for (int b = 1; b < st->n_nodes; ++b) {
      orc_path *p = PATH(st, b, s);
      const double scale = new_branches[b] / st->blen[b];
      for (uint32_t j = 0; j < p->n; ++j) p->t[j] *= scale;
    }
  for (int b = 1; b < st->n_nodes; ++b) st->blen[b] = new_branches[b];
}

ORC_API uint64_t orc_total_jumps(const orc_state *st) {
  uint64_t tot = 0;
  for (int b = 1; b < st->n_nodes; ++b)
    for (size_t s = 0; s < st->n_sites; ++s) tot += PATH(st, b, s)->n;
  return tot;
}
ORC_API void orc_get_paths(const orc_state *st, uint8_t *init, uint64_t *offsets, double *jumps) {
  uint64_t off = 0;
  for (int b = 1; b < st->n_nodes; ++b)
    for (size_t s = 0; s < st->n_sites; ++s) {
      const size_t idx = (size_t)(b - 1) * st->n_sites + s;
      const orc_path *p = PATH(st, b, s);
      init[idx] = p->init;
      offsets[idx] = off;
      if (p->n) memcpy(jumps + off, p->t, p->n * sizeof(double));
      off += p->n;
    }
  offsets[(size_t)(st->n_nodes - 1) * st->n_sites] = off;
}
ORC_API void orc_get_counters(const orc_state *st, uint64_t *out) {
  out[0] = st->n_overflow; out[1] = st->n_trials; out[2] = st->n_draws; out[3] = st->n_segments;
}

/* ------------------------------------------------ site-independent model (IndepSite.cpp)
 * Used by epievo_initialization to fit a context-free 2-rate model and to draw the
 * initial histories of the MCEM.  rates = {r0, r1}.  J/D layouts: [(b-1)*2 + state]. */

/* upward_process (IndepSite.cpp:53-96): q per node, p per non-root node; one segment per
 * branch, no neighbour context */
static void indep_upward(const orc_state *st, const double *rates, size_t site, double *q0,
                         double *q1, double *p0, double *p1) {
  for (int node = st->n_nodes - 1; node >= 0; --node) {
    double a = 1.0, b = 1.0;
    if (is_leaf(st, node)) {
      const int leaf_state = path_end_state(PATH(st, node, site));
      a = leaf_state ? 0.0 : 1.0;
      b = leaf_state ? 1.0 : 0.0;
    } else {
      for (uint32_t ch = 1; ch < st->subtree[node]; ch += st->subtree[node + ch]) {
        a *= p0[node + ch];
        b *= p1[node + ch];
      }
    }
    q0[node] = a; q1[node] = b;
    if (node == 0) continue;
    double P[4];
    trans_prob_mat(st, rates[0], rates[1], st->blen[node], P);
    p0[node] = P[0] * a + P[1] * b;
    p1[node] = P[2] * a + P[3] * b;
  }
}

/* expectation_J / expectation_D (ContinuousTimeMarkovModel.cpp:168-226) */
static void expectation_JD(const orc_state *st, double r0, double r1, double T, double J0[4],
                           double J1[4], double D0[4], double D1[4]) {
  const double s = r0 + r1, p = r0 * r1, d = r1 - r0;
  const double e = st->fexp(-s * T);
  const double C1 = d * (1 - e) / s;
  J0[0] = p * (T * (r1 - r0 * e) - C1) / (s * (r1 + r0 * e));
  J1[0] = J0[0];
  J0[3] = p * (T * (r0 - r1 * e) + C1) / (s * (r0 + r1 * e));
  J1[3] = J0[3];
  const double C2 = p * T * (1 + e) / (s * (1 - e));
  const double C3 = (r0 * r0 + r1 * r1) / (s * s);
  const double C4 = (2 * p) / (s * s);
  J0[1] = C2 + C3; J1[1] = C2 - C4; J0[2] = J1[1]; J1[2] = J0[1];
  const double r00 = r0 * r0, r11 = r1 * r1;
  const double E1 = 2 * p * (1 - e) / s;
  D0[0] = ((r11 + r00 * e) * T + E1) / (s * (r1 + r0 * e));
  D1[0] = T - D0[0];
  D1[3] = ((r00 + r11 * e) * T + E1) / (s * (r0 + r1 * e));
  D0[3] = T - D1[3];
  const double E2 = (p - r00) * (1 - e) / s;
  D1[1] = ((r00 - p * e) * T + E2) / (s * (r0 - r0 * e));
  D0[1] = T - D1[1];
  const double E3 = (p - r11) * (1 - e) / s;
  D0[2] = ((r11 - p * e) * T + E3) / (s * (r1 - r1 * e));
  D1[2] = T - D0[2];
}

/* one site's contribution to the conditional means: downward_process + weighted_J_D_branch
 * + joint_post (IndepSite.cpp:98-175); adds into J/D [(b-1)*2 + state] */
static void indep_site_expectation(const orc_state *st, const double *rates, size_t site,
                                   double *J, double *D, double *w /* 5*n_nodes scratch */) {
  const int N = st->n_nodes;
  double *q0 = w, *q1 = w + N, *p0 = w + 2 * N, *p1 = w + 3 * N, *pm = w + 4 * N;
  indep_upward(st, rates, site, q0, q1, p0, p1);
  const double pi_0 = rates[1] / (rates[0] + rates[1]);
  const double a = pi_0 * q0[0], b = (1 - pi_0) * q1[0];
  pm[0] = a / (a + b);
  for (int node = 1; node < N; ++node) {
    const double T = st->blen[node];
    double P[4], pj[4], J0[4], J1[4], D0[4], D1[4];
    trans_prob_mat(st, rates[0], rates[1], T, P);
    const double p0u = pm[st->parent[node]];
    pj[0] = P[0] * q0[node] * p0u / p0[node];
    pj[1] = P[1] * q1[node] * p0u / p0[node];
    pj[2] = P[2] * q0[node] * (1 - p0u) / p1[node];
    pj[3] = P[3] * q1[node] * (1 - p0u) / p1[node];
    const double Z = pj[0] + pj[1] + pj[2] + pj[3];
    for (int i = 0; i < 4; ++i) pj[i] /= Z;
    pm[node] = pj[0] + pj[2];
    expectation_JD(st, rates[0], rates[1], T, J0, J1, D0, D1);
    for (int i = 0; i < 4; ++i) {
      J[(node - 1) * 2 + 0] += pj[i] * J0[i];
      J[(node - 1) * 2 + 1] += pj[i] * J1[i];
      D[(node - 1) * 2 + 0] += pj[i] * D0[i];
      D[(node - 1) * 2 + 1] += pj[i] * D1[i];
    }
  }
}

static void indep_tree(const orc_state *st, const double *rates, size_t lo, size_t size, double *out,
                       double *w, int what);

/* expectation_sufficient_statistics (IndepSite.cpp:222-238): sums over ALL sites,
 * sequentially (reduce_mode SEQ, as the reference) or in the canonical tree order */
ORC_API void orc_indep_expectation(const orc_state *st, const double *rates, double *J, double *D) {
  const int B = st->n_nodes - 1;
  double *w = (double *)malloc(sizeof(double) * 5 * (size_t)st->n_nodes);
  for (int i = 0; i < 2 * B; ++i) { J[i] = 0.0; D[i] = 0.0; }
  if (st->reduce_mode == ORC_REDUCE_SEQ) {
    for (size_t s = 0; s < st->n_sites; ++s) indep_site_expectation(st, rates, s, J, D, w);
  } else {
    size_t pad = 1;
    while (pad < st->n_sites) pad *= 2;
    double *out = (double *)malloc(sizeof(double) * 4 * (size_t)B);
    indep_tree(st, rates, 0, pad, out, w, 0);
    memcpy(J, out, sizeof(double) * 2 * (size_t)B);
    memcpy(D, out + 2 * B, sizeof(double) * 2 * (size_t)B);
    free(out);
  }
  free(w);
}

/* per-site J/D of the CURRENT path on every branch (compute_sufficient_statistics,
 * IndepSite.cpp:266-297); adds into J/D [(b-1)*2 + state] */
static void indep_site_counts(const orc_state *st, size_t site, double *J, double *D) {
  for (int b = 1; b < st->n_nodes; ++b) {
    const orc_path *p = PATH(st, b, site);
    int prev = p->init;
    double time = 0.0;
    for (uint32_t j = 0; j < p->n; ++j) {
      J[(b - 1) * 2 + prev] += 1;
      D[(b - 1) * 2 + prev] += (p->t[j] - time);
      prev = 1 - prev;
      time = p->t[j];
    }
    D[(b - 1) * 2 + prev] += (st->blen[b] - time);
  }
}

/* what: 0 = conditional expectations, 1 = counts of the current paths; out = [J | D] */
static void indep_tree(const orc_state *st, const double *rates, size_t lo, size_t size, double *out,
                       double *w, int what) {
  const int B = st->n_nodes - 1;
  for (int i = 0; i < 4 * B; ++i) out[i] = 0.0;
  if (lo >= st->n_sites) return;
  if (size == 1) {
    if (what == 0) indep_site_expectation(st, rates, lo, out, out + 2 * B, w);
    else indep_site_counts(st, lo, out, out + 2 * B);
    return;
  }
  double *L = (double *)malloc(sizeof(double) * 8 * (size_t)B), *R = L + 4 * B;
  indep_tree(st, rates, lo, size / 2, L, w, what);
  indep_tree(st, rates, lo + size / 2, size / 2, R, w, what);
  for (int i = 0; i < 4 * B; ++i) out[i] = L[i] + R[i];
  free(L);
}

/* compute_sufficient_statistics: per-branch AVERAGES over the sites */
ORC_API void orc_indep_suffstats(const orc_state *st, double *J, double *D) {
  const int B = st->n_nodes - 1;
  for (int i = 0; i < 2 * B; ++i) { J[i] = 0.0; D[i] = 0.0; }
  if (st->reduce_mode == ORC_REDUCE_SEQ) {
    /* the reference sums one branch at a time over all sites */
    for (size_t s = 0; s < st->n_sites; ++s) indep_site_counts(st, s, J, D);
  } else {
    size_t pad = 1;
    while (pad < st->n_sites) pad *= 2;
    double *out = (double *)malloc(sizeof(double) * 4 * (size_t)B);
    indep_tree(st, NULL, 0, pad, out, NULL, 1);
    memcpy(J, out, sizeof(double) * 2 * (size_t)B);
    memcpy(D, out + 2 * B, sizeof(double) * 2 * (size_t)B);
    free(out);
  }
  for (int i = 0; i < 2 * B; ++i) { J[i] /= (double)st->n_sites; D[i] /= (double)st->n_sites; }
}

/* update_paths_indep (IndepSite.cpp:241-259) = upward_process + sampling_downward
 * (:177-215) for every site 0..n-1: the root state is kept, every branch gets a fresh
 * end state and an end-conditioned path by forward rejection. */
ORC_API void orc_indep_update_paths(orc_state *st, const double *rates, uint32_t sweep) {
  const int N = st->n_nodes;
  double *w = (double *)malloc(sizeof(double) * 4 * (size_t)N);
  double *q0 = w, *q1 = w + N, *p0 = w + 2 * N, *p1 = w + 3 * N;
  orc_scratch *sc = &st->scr;
  for (size_t site = 0; site < st->n_sites; ++site) {
    indep_upward(st, rates, site, q0, q1, p0, p1);
    orc_rng g;
    g.st = st; g.site = (uint32_t)(st->g0 + site); g.sweep = sweep;
    sc->prop[0].init = PATH(st, 1, site)->init;
    sc->prop[0].n = 0;
    for (int node = 1; node < N; ++node) {
      orc_path *pp = &sc->prop[node];
      const int start = path_end_state(&sc->prop[st->parent[node]]);
      pp->init = (uint8_t)start;
      pp->n = 0;
      double P[4];
      trans_prob_mat(st, rates[0], rates[1], st->blen[node], P);
      const double pr0 = P[2 * start] * q0[node] / (start ? p1[node] : p0[node]);
      const double u = rng_segment_uniform(&g, (uint32_t)node, 0);
      const int sampled = (u > pr0);
      forward_rejection(st, sc, &g, (uint32_t)node, 0, rates[0], rates[1], start, sampled,
                        st->blen[node], 0.0, pp);
    }
    for (int node = 1; node < N; ++node) {
      orc_path *dst = PATH(st, node, site), *src = &sc->prop[node];
      orc_path tmp = *dst; *dst = *src; *src = tmp;
    }
  }
  free(w);
}

/* ------------------------------------------------ initial paths for one branch
 * initialize_paths_indep (src/prog/epievo_sim_pairwise.cpp:62-110): every site gets an
 * end-conditioned path root[i] -> leaf[i] on [0,T] drawn independently by forward
 * rejection (EndCondSampling.cpp:512-542) with the context rates read off the ROOT
 * sequence; the two end sites get at most one jump, placed uniformly.  Draw order of the
 * reference (rung A): site 0, site n-1, then sites 1..n-2.  Rung B keys the draws by
 * (site, sweep = ORC_INIT_SWEEP, branch 1, segment 0).  Output: node-major flat paths of
 * the two-node tree (init, offsets[n+1], jumps); returns the number of jumps, or the
 * number needed when `jumps_cap` is too small. */
#define ORC_INIT_SWEEP 0xffffffffu
ORC_API uint64_t orc_init_paths_indep(int rng_mode, int math_mode, uint64_t seed, const double *rates,
                                      uint64_t n, const uint8_t *root, const uint8_t *leaf, double T,
                                      uint8_t *init, uint64_t *offsets, double *jumps,
                                      uint64_t jumps_cap) {
  orc_state st;
  memset(&st, 0, sizeof(st));
  st.rng_mode = rng_mode; st.math_mode = math_mode; st.seed = seed; st.cap = 0;
  st.sampler_mode = rng_mode == ORC_RNG_PHILOX ? ORC_SAMPLER_NIELSEN : ORC_SAMPLER_FORWARD;
  set_math(&st);
  orc_mt_seed(&st.mt, (uint32_t)seed);
  orc_scratch sc;
  memset(&sc, 0, sizeof(sc));
  orc_path *paths = (orc_path *)calloc(n, sizeof(orc_path));
  orc_rng g;
  g.st = &st; g.sweep = ORC_INIT_SWEEP;
  const uint64_t ends[2] = {0, n - 1};
  for (int e = 0; e < 2; ++e) {
    const uint64_t s = ends[e];
    paths[s].init = root[s];
    g.site = (uint32_t)s;
    if (root[s] != leaf[s]) path_push(&paths[s], rng_segment_uniform(&g, 1, 0) * (T - 0.0) + 0.0);
  }
  for (uint64_t s = 1; s + 1 < n; ++s) {
    const int c0 = 4 * root[s - 1] + root[s + 1];
    paths[s].init = root[s];
    g.site = (uint32_t)s;
    forward_rejection(&st, &sc, &g, 1, 0, rates[c0], rates[c0 | 2], root[s], leaf[s], T, 0.0, &paths[s]);
  }
  uint64_t tot = 0;
  for (uint64_t s = 0; s < n; ++s) {
    init[s] = paths[s].init;
    offsets[s] = tot;
    for (uint32_t k = 0; k < paths[s].n; ++k, ++tot)
      if (tot < jumps_cap) jumps[tot] = paths[s].t[k];
    free(paths[s].t);
  }
  offsets[n] = tot;
  free(paths); free(sc.trial);
  return tot;
}

/* ------------------------------------------------ exact posterior by whole-sequence rejection
 * The reference's own end-to-end check of the sampler (src/harnesses/MCMC_test.cpp:191-216,
 * 367-379): forward-simulate the whole sequence from the root (Gillespie over the interior
 * sites, ends fixed) and keep only histories that hit the observed leaf sequence; the kept
 * histories are exact draws from the posterior the MCMC targets.  Feasible for a handful of
 * sites only.  Returns the number of kept histories; Jm/Dm = means of the 8-context
 * sufficient statistics over the interior triples, J2/D2 = means of their squares. */
ORC_API uint64_t orc_exact_posterior(const double *rates, uint64_t n, const uint8_t *root,
                                     const uint8_t *leaf, double T, uint64_t seed, uint64_t want,
                                     uint64_t max_trials, double *Jm, double *Dm, double *J2, double *D2) {
  orc_mt19937 g;
  orc_mt_seed(&g, (uint32_t)seed);
  orc_path *paths = (orc_path *)calloc(n, sizeof(orc_path));
  uint8_t *seq = (uint8_t *)malloc(n);
  for (int c = 0; c < 8; ++c) { Jm[c] = Dm[c] = J2[c] = D2[c] = 0.0; }
  uint64_t kept = 0;
  for (uint64_t trial = 0; trial < max_trials && kept < want; ++trial) {
    memcpy(seq, root, n);
    for (uint64_t s = 0; s < n; ++s) { paths[s].init = root[s]; paths[s].n = 0; }
    double t = 0.0;
    for (;;) {
      double total = 0.0;
      for (uint64_t s = 1; s + 1 < n; ++s) total += rates[4 * seq[s - 1] + 2 * seq[s] + seq[s + 1]];
      t += -log(1.0 - orc_mt_canonical(&g)) / total;
      if (!(t < T)) break;
      double x = orc_mt_canonical(&g) * total;
      uint64_t pick = n - 2;
      for (uint64_t s = 1; s + 1 < n; ++s) {
        const double r = rates[4 * seq[s - 1] + 2 * seq[s] + seq[s + 1]];
        if (x < r) { pick = s; break; }
        x -= r;
      }
      seq[pick] ^= 1;
      path_push(&paths[pick], t);
    }
    if (memcmp(seq, leaf, n) != 0) continue;
    double J[8] = {0, 0, 0, 0, 0, 0, 0, 0}, D[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t s = 1; s + 1 < n; ++s) add_suff_stats(&paths[s - 1], &paths[s], &paths[s + 1], T, J, D);
    for (int c = 0; c < 8; ++c) { Jm[c] += J[c]; Dm[c] += D[c]; J2[c] += J[c] * J[c]; D2[c] += D[c] * D[c]; }
    ++kept;
  }
  for (int c = 0; c < 8 && kept; ++c) { Jm[c] /= kept; Dm[c] /= kept; J2[c] /= kept; D2[c] /= kept; }
  for (uint64_t s = 0; s < n; ++s) free(paths[s].t);
  free(paths); free(seq);
  return kept;
}

/* The same on a TREE (MCMC_test.cpp runs a single branch; the sampler's real workload is a
 * tree): the whole sequence evolves down every branch in pre-order from the fixed root sequence
 * (SAMPLE_ROOT is false), the two end sites never jump (the sampler does not update them), and
 * a history is kept only if every LEAF node ends in its observed sequence (leaf[node][site];
 * rows of internal nodes are ignored).  Jm/Dm/J2/D2: [(b-1)*8 + ctx] per branch. */
ORC_API uint64_t orc_exact_posterior_tree(const double *rates, uint64_t n, int n_nodes,
                                          const uint32_t *parent, const uint32_t *subtree,
                                          const double *blen, const uint8_t *root, const uint8_t *leaf,
                                          uint64_t seed, uint64_t want, uint64_t max_trials, double *Jm,
                                          double *Dm, double *J2, double *D2) {
  orc_mt19937 g;
  orc_mt_seed(&g, (uint32_t)seed);
  const int B = n_nodes - 1;
  orc_path *paths = (orc_path *)calloc((size_t)n_nodes * n, sizeof(orc_path));
  uint8_t *seq = (uint8_t *)malloc((size_t)n_nodes * n);   /* end sequence of every node */
  for (int i = 0; i < B * 8; ++i) { Jm[i] = Dm[i] = J2[i] = D2[i] = 0.0; }
  double *J = (double *)malloc(sizeof(double) * (size_t)B * 8), *D = (double *)malloc(sizeof(double) * (size_t)B * 8);
  uint64_t kept = 0;
  for (uint64_t trial = 0; trial < max_trials && kept < want; ++trial) {
    memcpy(seq, root, n);
    int ok = 1;
    for (int node = 1; node < n_nodes && ok; ++node) {
      uint8_t *cur = seq + (size_t)node * n;
      memcpy(cur, seq + (size_t)parent[node] * n, n);
      orc_path *pp = paths + (size_t)node * n;
      for (uint64_t s = 0; s < n; ++s) { pp[s].init = cur[s]; pp[s].n = 0; }
      double t = 0.0;
      for (;;) {
        double total = 0.0;
        for (uint64_t s = 1; s + 1 < n; ++s) total += rates[4 * cur[s - 1] + 2 * cur[s] + cur[s + 1]];
        t += -log(1.0 - orc_mt_canonical(&g)) / total;
        if (!(t < blen[node])) break;
        double x = orc_mt_canonical(&g) * total;
        uint64_t pick = n - 2;
        for (uint64_t s = 1; s + 1 < n; ++s) {
          const double r = rates[4 * cur[s - 1] + 2 * cur[s] + cur[s + 1]];
          if (x < r) { pick = s; break; }
          x -= r;
        }
        cur[pick] ^= 1;
        path_push(&pp[pick], t);
      }
      if (subtree[node] == 1 && memcmp(cur, leaf + (size_t)node * n, n) != 0) ok = 0;
    }
    if (!ok) continue;
    for (int i = 0; i < B * 8; ++i) { J[i] = 0.0; D[i] = 0.0; }
    for (int node = 1; node < n_nodes; ++node) {
      const orc_path *pp = paths + (size_t)node * n;
      for (uint64_t s = 1; s + 1 < n; ++s)
        add_suff_stats(&pp[s - 1], &pp[s], &pp[s + 1], blen[node], J + (node - 1) * 8, D + (node - 1) * 8);
    }
    for (int i = 0; i < B * 8; ++i) { Jm[i] += J[i]; Dm[i] += D[i]; J2[i] += J[i] * J[i]; D2[i] += D[i] * D[i]; }
    ++kept;
  }
  for (int i = 0; i < B * 8 && kept; ++i) { Jm[i] /= kept; Dm[i] /= kept; J2[i] /= kept; D2[i] /= kept; }
  for (size_t i = 0; i < (size_t)n_nodes * n; ++i) free(paths[i].t);
  free(paths); free(seq); free(J); free(D);
  return kept;
}

/* ------------------------------------------------- per-function KAT hooks */
ORC_API void orc_kat_trans_prob_mat(int math_mode, double r0, double r1, double t, double *P) {
  orc_state st; st.math_mode = math_mode; set_math(&st);
  trans_prob_mat(&st, r0, r1, t, P);
}
ORC_API double orc_kat_get_trans_prob(int math_mode, double r0, double r1, double t, int a, int b) {
  orc_state st; st.math_mode = math_mode; set_math(&st);
  return get_trans_prob(&st, r0, r1, t, a, b);
}
ORC_API int orc_kat_segments(const double *rates, int l_init, uint32_t nl, const double *lj,
                             int r_init, uint32_t nr, const double *rj, double tot_time,
                             double *rate0, double *rate1, uint64_t *trip0, uint64_t *trip1,
                             double *len) {
  orc_path l = {(uint8_t)l_init, nl, nl, (double *)lj}, r = {(uint8_t)r_init, nr, nr, (double *)rj};
  orc_segs s; memset(&s, 0, sizeof(s));
  collect_segments(&l, &r, tot_time, &s);
  for (int k = 0; k < s.K; ++k) {
    rate0[k] = rates[s.trip0[k]]; rate1[k] = rates[s.trip0[k] | 2];
    trip0[k] = s.trip0[k]; trip1[k] = s.trip0[k] | 2; len[k] = s.len[k];
  }
  const int K = s.K;
  free(s.len); free(s.trip0); free(s.p0); free(s.p1);
  return K;
}
ORC_API void orc_kat_suffstats(int l_init, uint32_t nl, const double *lj, int m_init, uint32_t nm,
                               const double *mj, int r_init, uint32_t nr, const double *rj,
                               double tot_time, double *J, double *D) {
  orc_path l = {(uint8_t)l_init, nl, nl, (double *)lj}, m = {(uint8_t)m_init, nm, nm, (double *)mj},
           r = {(uint8_t)r_init, nr, nr, (double *)rj};
  add_suff_stats(&l, &m, &r, tot_time, J, D);
}
/* end-conditioned sampler alone: n samples of a 2-state path a->b on [0,T] by forward
 * rejection (sample i uses Philox site index i); returns the means of
 * (#jumps 0->1, #jumps 1->0, time spent in state 0) -- to be compared with the
 * reference's closed forms expectation_J / expectation_D. */
ORC_API void orc_kat_end_cond_means(int rng_mode, int math_mode, uint64_t seed, double r0,
                                    double r1, int a, int b, double T, uint64_t n, double *out) {
  orc_state st;
  memset(&st, 0, sizeof(st));
  st.rng_mode = rng_mode & 0xff; st.math_mode = math_mode; st.seed = seed; st.cap = 0;
  /* bit 8 of rng_mode selects the sampler explicitly: 0x100 forward, 0x200 Nielsen */
  st.sampler_mode = (rng_mode & 0x200) ? ORC_SAMPLER_NIELSEN : (rng_mode & 0x100) ? ORC_SAMPLER_FORWARD
                    : (st.rng_mode == ORC_RNG_PHILOX ? ORC_SAMPLER_NIELSEN : ORC_SAMPLER_FORWARD);
  set_math(&st);
  orc_mt_seed(&st.mt, (uint32_t)seed);
  orc_scratch sc;
  memset(&sc, 0, sizeof(sc));
  orc_path p;
  memset(&p, 0, sizeof(p));
  double j01 = 0, j10 = 0, d0 = 0;
  for (uint64_t i = 0; i < n; ++i) {
    orc_rng g;
    g.st = &st; g.site = (uint32_t)i; g.sweep = 0;
    p.n = 0;
    forward_rejection(&st, &sc, &g, 1, 0, r0, r1, a, b, T, 0.0, &p);
    int s = a;
    double prev = 0.0;
    for (uint32_t k = 0; k < p.n; ++k) {
      if (s == 0) { j01 += 1.0; d0 += p.t[k] - prev; } else j10 += 1.0;
      prev = p.t[k]; s ^= 1;
    }
    if (s == 0) d0 += T - prev;
  }
  out[0] = j01 / (double)n; out[1] = j10 / (double)n; out[2] = d0 / (double)n;
  free(p.t); free(sc.trial);
}

/* the raw paths of the same sampler: n consecutive samples (one mt19937 stream in MT mode,
 * Philox site index i otherwise); counts[i] jumps each, times concatenated (up to cap).
 * rng_mode carries the sampler bits described above.  Returns the total number of jumps. */
ORC_API uint64_t orc_kat_end_cond_paths(int rng_mode, int math_mode, uint64_t seed, double r0, double r1,
                                        int a, int b, double T, uint64_t n, uint32_t *counts,
                                        double *times, uint64_t cap) {
  orc_state st;
  memset(&st, 0, sizeof(st));
  st.rng_mode = rng_mode & 0xff; st.math_mode = math_mode; st.seed = seed; st.cap = 0;
  st.sampler_mode = (rng_mode & 0x200) ? ORC_SAMPLER_NIELSEN : (rng_mode & 0x100) ? ORC_SAMPLER_FORWARD
                    : (st.rng_mode == ORC_RNG_PHILOX ? ORC_SAMPLER_NIELSEN : ORC_SAMPLER_FORWARD);
  set_math(&st);
  orc_mt_seed(&st.mt, (uint32_t)seed);
  orc_scratch sc;
  memset(&sc, 0, sizeof(sc));
  orc_path p;
  memset(&p, 0, sizeof(p));
  uint64_t tot = 0;
  for (uint64_t i = 0; i < n; ++i) {
    orc_rng g;
    g.st = &st; g.site = (uint32_t)i; g.sweep = 0;
    p.n = 0;
    forward_rejection(&st, &sc, &g, 1, 0, r0, r1, a, b, T, 0.0, &p);
    counts[i] = p.n;
    for (uint32_t k = 0; k < p.n; ++k, ++tot)
      if (tot < cap) times[tot] = p.t[k];
  }
  free(p.t); free(sc.trial);
  return tot;
}

ORC_API double orc_kat_exp(double x) { return orc_exp(x); }
ORC_API double orc_kat_log(double x) { return orc_log(x); }
ORC_API void orc_kat_exp_log_array(const double *x, uint64_t n, double *e, double *l) {
  for (uint64_t i = 0; i < n; ++i) { e[i] = orc_exp(x[i]); l[i] = orc_log(x[i]); }
}
ORC_API void orc_kat_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
  orc_philox4x32_10(ctr, key, out);
}
ORC_API void orc_kat_keyed_block(uint64_t seed, uint32_t site, uint32_t sweep, uint32_t b,
                                 uint32_t k, uint32_t t, uint32_t blk, double *d) {
  orc_keyed_block(seed, site, sweep, b, k, t, blk, d);
}
ORC_API void orc_kat_mt_canonical(uint64_t seed, uint64_t n, double *out) {
  orc_mt19937 g; orc_mt_seed(&g, (uint32_t)seed);
  for (uint64_t i = 0; i < n; ++i) out[i] = orc_mt_canonical(&g);
}

/* ======================================================================================
 * Forward simulation, parallel rung: what epievo_sim computes
 * (src/prog/epievo_sim.cpp:102-152,329-352 over TripletSampler.cpp:165-184) restated as
 * THINNING, the form the GPU generator (epievo_amd/csrc/epv_forward.h) runs site-parallel:
 * every interior site of a branch carries a Poisson stream of candidate events at the rate
 * lam_max = max_c rate_c (gaps -log(1 - u)/lam_max), and a candidate at time t flips its site with
 * probability rate[context at t-]/lam_max.  Superposition + thinning give exactly the law of the
 * reference's Gillespie loop (total rate sum_c count_c rate_c, a context drawn in proportion to
 * count_c rate_c, a uniform site of that context).  The randomness is keyed by (seed, node, site,
 * candidate index), so the outcome is a function of the seed alone: here the candidates of a branch
 * are simply processed in global time order; the GPU resolves them in any order consistent with
 * the nearest-neighbour dependencies and must produce the same bits.
 * Root sequence: EpiEvoModel::sample_state_sequence (EpiEvoModel.cpp:281-298) with keyed uniforms.
 * Sites 0 and n-1 never change (TripletSampler.cpp:37-70 buckets interior positions only).
 * Returns the total number of jumps (or UINT64_MAX when jumps_cap is too small). */
#define ORC_FWD_SWEEP 0xfffffffeu
#define ORC_FWD_ROOT_SWEEP 0xfffffffdu
typedef struct { double t; uint32_t site; double u; } orc_cand;
static int cand_cmp(const void *a, const void *b) {
  const orc_cand *x = (const orc_cand *)a, *y = (const orc_cand *)b;
  if (x->t < y->t) return -1;
  if (x->t > y->t) return 1;
  return x->site < y->site ? -1 : (x->site > y->site ? 1 : 0);
}
ORC_API uint64_t orc_forward_thinning(const double *rates, const double *T4, int n_nodes, const uint32_t *parent,
                                      const double *blen, uint64_t n_sites, uint64_t seed, const uint8_t *root_in,
                                      uint8_t *init_out, uint64_t *offsets_out, double *jumps_out, uint64_t jumps_cap,
                                      uint8_t *states_out /* [n_nodes][n_sites] end states, may be NULL */) {
  const size_t n = (size_t)n_sites;
  uint8_t *end = (uint8_t *)malloc((size_t)n_nodes * n);
  double d[2];
  if (root_in) {
    memcpy(end, root_in, n);
  } else {
    const double pi1 = (1.0 - T4[0]) / (2.0 - T4[3] - T4[0]);
    orc_keyed_block(seed, 0u, ORC_FWD_ROOT_SWEEP, 0, 0, 0, 0, d);
    end[0] = d[0] < pi1;
    for (size_t i = 1; i < n; ++i) {
      orc_keyed_block(seed, (uint32_t)i, ORC_FWD_ROOT_SWEEP, 0, 0, 0, 0, d);
      const double p = end[i - 1] ? T4[3] : T4[0];
      end[i] = (d[0] <= p) ? end[i - 1] : (uint8_t)!end[i - 1];
    }
  }
  double lam_max = rates[0];
  for (int c = 1; c < 8; ++c) if (rates[c] > lam_max) lam_max = rates[c];
  double pacc[8];
  for (int c = 0; c < 8; ++c) pacc[c] = rates[c] / lam_max;
  uint64_t total = 0;
  size_t cap = 1024, cnt = 0;
  orc_cand *cand = (orc_cand *)malloc(cap * sizeof(orc_cand));
  uint32_t *per_site = (uint32_t *)malloc(n * sizeof(uint32_t));
  double **sj = (double **)calloc(n, sizeof(double *));
  int ok = 1;
  for (int node = 1; node < n_nodes && ok; ++node) {
    uint8_t *st = end + (size_t)node * n;
    memcpy(st, end + (size_t)parent[node] * n, n);
    memcpy(init_out + (size_t)(node - 1) * n, st, n);
    cnt = 0;
    for (size_t s = 1; s + 1 < n; ++s) {
      double t = 0.0;
      for (uint32_t k = 0;; ++k) {
        orc_keyed_block(seed, (uint32_t)s, ORC_FWD_SWEEP, (uint32_t)node, 0, k, 0, d);
        t += -orc_log(1.0 - d[0]) / lam_max;
        if (!(t < blen[node])) break;
        if (cnt == cap) { cap *= 2; cand = (orc_cand *)realloc(cand, cap * sizeof(orc_cand)); }
        cand[cnt].t = t; cand[cnt].site = (uint32_t)s; cand[cnt].u = d[1];
        ++cnt;
      }
    }
    qsort(cand, cnt, sizeof(orc_cand), cand_cmp);
    memset(per_site, 0, n * sizeof(uint32_t));
    /* two passes: count the accepted flips of every site, then lay them out site by site */
    uint8_t *work = (uint8_t *)malloc(n);
    memcpy(work, st, n);
    uint8_t *acc = (uint8_t *)malloc(cnt ? cnt : 1);
    for (size_t i = 0; i < cnt; ++i) {
      const size_t s = cand[i].site;
      const int ctx = 4 * work[s - 1] + 2 * work[s] + work[s + 1];
      acc[i] = cand[i].u < pacc[ctx];
      if (acc[i]) { work[s] ^= 1; ++per_site[s]; }
    }
    uint64_t *off = offsets_out + (size_t)(node - 1) * n;
    for (size_t s = 0; s < n; ++s) { off[s] = total; total += per_site[s]; }
    if (total > jumps_cap) { ok = 0; free(work); free(acc); break; }
    memset(per_site, 0, n * sizeof(uint32_t));
    for (size_t i = 0; i < cnt; ++i)
      if (acc[i]) { const size_t s = cand[i].site; jumps_out[off[s] + per_site[s]++] = cand[i].t; }
    memcpy(st, work, n);
    free(work); free(acc);
  }
  offsets_out[(size_t)(n_nodes - 1) * n] = total;
  if (states_out && ok) memcpy(states_out, end, (size_t)n_nodes * n);
  free(cand); free(per_site); free(sj); free(end);
  return ok ? total : UINT64_MAX;
}
