/* ref_shim.cpp -- TEST INFRASTRUCTURE.  A flat extern "C" face over the
 * UNMODIFIED reference library (src/libepievo of /root/reference), compiled
 * where its sources lie by oracle/Makefile into oracle/_ref/.  Nothing of the
 * reference is copied into this repository: this file only *calls* it, the way
 * a ~60-line main() of src/prog would (epievo_est_params_histories.cpp:169-264).
 *
 * Used (a) to pin the CPU restatement oracle/epv_oracle.c bit-for-bit
 * (tests/test_oracle_vs_ref.py) and (b) to write the golden fixtures in
 * tests/golden/ (tests/golden/make_golden.py).  It is only buildable where
 * /root/reference exists; the built .so travels to the GPU box.
 */
#include <limits>
#include <random>
#include <functional>
#include <vector>
#include <array>
#include <string>
#include <sstream>
#include <cstring>
#include <cstdint>
#include <numeric>
#include <algorithm>

#include "Path.hpp"
#include "Segment.hpp"
#include "EpiEvoModel.hpp"
#include "TreeHelper.hpp"
#include "ContinuousTimeMarkovModel.hpp"
#include "SingleSiteSampler.hpp"
#include "ParamEstimation.hpp"
#include "EndCondSampling.hpp"
#include "IndepSite.hpp"
#include "TripletSampler.hpp"
#include "GlobalJump.hpp"
#include "PhyloTreePreorder.hpp"
#include "epievo_utils.hpp"
#include <fstream>

using std::vector;
using std::array;

/* defined (non-static) in SingleSiteSampler.cpp:356, not declared in its header */
double path_log_likelihood(const EpiEvoModel &mod, const vector<Path> &l,
                           const vector<Path> &m, const vector<Path> &r,
                           const array<double, 8> &log_rates);

namespace {
struct RefState {
  EpiEvoModel model;
  TreeHelper th;
  vector<vector<Path> > paths;  // [site][node]
  std::mt19937 gen;
  SingleSiteSampler *mcmc;
  RefState() : mcmc(nullptr) {}
  ~RefState() { delete mcmc; }
};

void set_model(EpiEvoModel &m, const double *rates, const double *T) {
  for (size_t i = 0; i < 8; ++i) m.triplet_rates[i] = rates[i];
  m.T = two_by_two(T[0], T[1], T[2], T[3]);
}
}  // namespace

extern "C" {

void *ref_create(uint64_t n_sites, int n_nodes, const uint32_t *parent,
                 const uint32_t *subtree, const double *branches, const double *rates,
                 const double *T, const uint8_t *init, const uint64_t *offsets,
                 const double *jumps) {
  RefState *st = new RefState();
  set_model(st->model, rates, T);
  st->th.n_nodes = n_nodes;
  for (int i = 0; i < n_nodes; ++i) {
    st->th.parent_ids.push_back(parent[i]);
    st->th.subtree_sizes.push_back(subtree[i]);
    st->th.branches.push_back(branches[i]);
    st->th.node_names.push_back("node_" + std::to_string(i));
  }
  st->paths.assign(n_sites, vector<Path>(n_nodes));
  for (int b = 1; b < n_nodes; ++b)
    for (uint64_t s = 0; s < n_sites; ++s) {
      const uint64_t idx = (uint64_t)(b - 1) * n_sites + s;
      Path &p = st->paths[s][b];
      p.init_state = init[idx];
      p.tot_time = branches[b];
      p.jumps.assign(jumps + offsets[idx], jumps + offsets[idx + 1]);
    }
  return st;
}

void ref_destroy(void *h) { delete static_cast<RefState *>(h); }

void ref_set_model(void *h, const double *rates, const double *T) {
  set_model(static_cast<RefState *>(h)->model, rates, T);
}

void ref_seed(void *h, uint64_t seed) { static_cast<RefState *>(h)->gen.seed(seed); }

/* SingleSiteSampler(burn_in, batch) + reset(), as est_params_histories.cpp:236,241 */
static bool g_sample_root = false;
void ref_reset(void *h, uint64_t burn_in, uint64_t batch) {
  RefState *st = static_cast<RefState *>(h);
  delete st->mcmc;
  st->mcmc = new SingleSiteSampler(burn_in, batch);
  st->mcmc->SAMPLE_ROOT = g_sample_root;
  st->mcmc->reset(st->model, st->paths);
}
/* the public field SingleSiteSampler::SAMPLE_ROOT (SingleSiteSampler.hpp:78) for the samplers made afterwards */
void ref_set_sample_root(int on) { g_sample_root = on != 0; }

/* what reset() caches privately: path_log_likelihood of every interior triple */
void ref_tri_llh(void *h, double *out) {
  RefState *st = static_cast<RefState *>(h);
  array<double, 8> log_rates;
  for (size_t i = 0; i < 8; ++i) log_rates[i] = std::log(st->model.triplet_rates[i]);
  const size_t n = st->paths.size();
  for (size_t s = 0; s < n; ++s) out[s] = 0.0;
  for (size_t s = 1; s + 1 < n; ++s)
    out[s] = path_log_likelihood(st->model, st->paths[s - 1], st->paths[s], st->paths[s + 1],
                                 log_rates);
}

/* sweeps through the public per-site entry, as epievo_sim_pairwise.cpp:267-273 */
uint64_t ref_sweeps(void *h, uint64_t n_sweeps) {
  RefState *st = static_cast<RefState *>(h);
  uint64_t n_acc = 0;
  for (uint64_t k = 0; k < n_sweeps; ++k)
    for (size_t s = 1; s + 1 < st->paths.size(); ++s)
      n_acc += st->mcmc->Metropolis_Hastings_site(st->model, st->th, s, st->paths, st->gen);
  return n_acc;
}

int ref_mh_site(void *h, uint64_t site) {
  RefState *st = static_cast<RefState *>(h);
  return st->mcmc->Metropolis_Hastings_site(st->model, st->th, site, st->paths, st->gen);
}

void ref_run_mcmc(void *h, double *J, double *D, double *acc_rate) {
  RefState *st = static_cast<RefState *>(h);
  vector<vector<double> > Jv, Dv;
  double acc = 0.0;
  st->mcmc->run_mcmc(st->model, st->th, st->paths, st->gen, Jv, Dv, acc);
  for (size_t b = 1; b < st->th.n_nodes; ++b)
    for (size_t i = 0; i < 8; ++i) {
      J[(b - 1) * 8 + i] = Jv[b][i];
      D[(b - 1) * 8 + i] = Dv[b][i];
    }
  *acc_rate = acc;
}

void ref_suffstats(void *h, double *J, double *D) {
  RefState *st = static_cast<RefState *>(h);
  vector<vector<double> > Jv, Dv;
  get_sufficient_statistics(st->paths, Jv, Dv);
  for (size_t b = 1; b < st->th.n_nodes; ++b)
    for (size_t i = 0; i < 8; ++i) {
      J[(b - 1) * 8 + i] = Jv[b][i];
      D[(b - 1) * 8 + i] = Dv[b][i];
    }
}

void ref_scale_jump_times(void *h, const double *new_branches) {
  RefState *st = static_cast<RefState *>(h);
  for (size_t b = 0; b < st->th.n_nodes; ++b) st->th.branches[b] = new_branches[b];
  scale_jump_times(st->paths, st->th);
}

uint64_t ref_total_jumps(void *h) {
  RefState *st = static_cast<RefState *>(h);
  uint64_t tot = 0;
  for (size_t s = 0; s < st->paths.size(); ++s)
    for (size_t b = 1; b < st->th.n_nodes; ++b) tot += st->paths[s][b].jumps.size();
  return tot;
}

void ref_get_paths(void *h, uint8_t *init, uint64_t *offsets, double *jumps) {
  RefState *st = static_cast<RefState *>(h);
  const size_t n = st->paths.size();
  uint64_t off = 0;
  for (size_t b = 1; b < st->th.n_nodes; ++b)
    for (size_t s = 0; s < n; ++s) {
      const Path &p = st->paths[s][b];
      const size_t idx = (b - 1) * n + s;
      init[idx] = p.init_state;
      offsets[idx] = off;
      for (size_t j = 0; j < p.jumps.size(); ++j) jumps[off + j] = p.jumps[j];
      off += p.jumps.size();
    }
  offsets[(st->th.n_nodes - 1) * n] = off;
}

/* ---- host-side model / M-step (EpiEvoModel.cpp:319-377, ParamEstimation.cpp:337-422) */
int ref_read_model(const char *param_file, int scale, double *rates, double *T,
                   double *baseline) {
  try {
    EpiEvoModel m;
    read_model(param_file, m);
    if (scale) m.scale_triplet_rates();
    for (size_t i = 0; i < 8; ++i) rates[i] = m.triplet_rates[i];
    T[0] = m.T(0, 0); T[1] = m.T(0, 1); T[2] = m.T(1, 0); T[3] = m.T(1, 1);
    baseline[0] = m.stationary_baseline(0, 0); baseline[1] = m.stationary_baseline(0, 1);
    baseline[2] = m.stationary_baseline(1, 0); baseline[3] = m.stationary_baseline(1, 1);
    return 0;
  } catch (...) { return 1; }
}

/* M-step as est_params_histories.cpp:253-263.  J/D: [(b-1)*8+i].  branches in/out. */
double ref_m_step(int optimize_branches, int n_nodes, const double *J, const double *D,
                  double *rates, double *T, double *baseline, double *branches,
                  char *param_text, int param_text_len) {
  EpiEvoModel m;
  array<double, 8> r;
  for (size_t i = 0; i < 8; ++i) r[i] = rates[i];
  m.rebuild_from_triplet_rates(r);
  vector<vector<double> > Jv(n_nodes), Dv(n_nodes);
  for (int b = 1; b < n_nodes; ++b) {
    Jv[b].assign(J + (b - 1) * 8, J + b * 8);
    Dv[b].assign(D + (b - 1) * 8, D + b * 8);
  }
  TreeHelper th;
  th.n_nodes = n_nodes;
  th.branches.assign(branches, branches + n_nodes);
  double llh = 0.0;
  if (!optimize_branches) {
    llh = estimate_rates(false, 1e-10, Jv, Dv, m);
    set_one_change_per_site_per_unit_time(m.triplet_rates, th.branches);
  } else {
    llh = estimate_rates_and_branches(false, 1e-10, Jv, Dv, th, m);
  }
  for (size_t i = 0; i < 8; ++i) rates[i] = m.triplet_rates[i];
  T[0] = m.T(0, 0); T[1] = m.T(0, 1); T[2] = m.T(1, 0); T[3] = m.T(1, 1);
  baseline[0] = m.stationary_baseline(0, 0); baseline[1] = m.stationary_baseline(0, 1);
  baseline[2] = m.stationary_baseline(1, 0); baseline[3] = m.stationary_baseline(1, 1);
  for (int b = 0; b < n_nodes; ++b) branches[b] = th.branches[b];
  if (param_text && param_text_len > 0) {
    const std::string s = m.format_for_param_file();
    std::strncpy(param_text, s.c_str(), param_text_len - 1);
    param_text[param_text_len - 1] = '\0';
  }
  return llh;
}

/* the glue of initialize_paths_indep (src/prog/epievo_sim_pairwise.cpp:62-110) around
 * the LINKED end_cond_sample_forward_rejection(ctmm, ...) (EndCondSampling.cpp:512-542):
 * same call order, same std::mt19937 */
uint64_t ref_init_paths_indep(uint64_t seed, const double *rates, uint64_t n, const uint8_t *root,
                              const uint8_t *leaf, double T, uint8_t *init, uint64_t *offsets,
                              double *jumps, uint64_t jumps_cap) {
  std::mt19937 gen(seed);
  std::uniform_real_distribution<double> dist(0.0, T);
  vector<vector<double> > pj(n);
  if (root[0] != leaf[0]) pj[0].push_back(dist(gen));
  if (root[n - 1] != leaf[n - 1]) pj[n - 1].push_back(dist(gen));
  for (uint64_t s = 1; s + 1 < n; ++s) {
    const size_t p0 = triple2idx(root[s - 1], false, root[s + 1]);
    const size_t p1 = triple2idx(root[s - 1], true, root[s + 1]);
    const TwoStateCTMarkovModel ctmm(rates[p0], rates[p1]);
    end_cond_sample_forward_rejection(ctmm, root[s], leaf[s], T, gen, pj[s], 0.0);
  }
  uint64_t tot = 0;
  for (uint64_t s = 0; s < n; ++s) {
    init[s] = root[s];
    offsets[s] = tot;
    for (size_t k = 0; k < pj[s].size(); ++k, ++tot)
      if (tot < jumps_cap) jumps[tot] = pj[s][k];
  }
  offsets[n] = tot;
  return tot;
}

/* the LINKED end_cond_sampling_Nielsen (EndCondSampling.cpp:583-617) and, for sampler = 0,
 * end_cond_sample_forward_rejection(ctmm, ...) (:512-542): n consecutive samples a -> b on
 * [0, T] from one std::mt19937(seed) */
uint64_t ref_kat_end_cond_paths(int sampler, uint64_t seed, double r0, double r1, int a, int b, double T,
                                uint64_t n, uint32_t *counts, double *times, uint64_t cap) {
  std::mt19937 gen(seed);
  const TwoStateCTMarkovModel ctmm(r0, r1);
  uint64_t tot = 0;
  for (uint64_t i = 0; i < n; ++i) {
    vector<double> jt;
    if (sampler == 1) end_cond_sampling_Nielsen(ctmm, a, b, T, gen, jt, 0.0);
    else end_cond_sample_forward_rejection(ctmm, a, b, T, gen, jt, 0.0);
    counts[i] = (uint32_t)jt.size();
    for (size_t k = 0; k < jt.size(); ++k, ++tot)
      if (tot < cap) times[tot] = jt[k];
  }
  return tot;
}

/* ---- site-independent model: the linked IndepSite.cpp */
void ref_indep_expectation(void *h, const double *rates, double *J, double *D) {
  RefState *st = static_cast<RefState *>(h);
  vector<double> r(rates, rates + 2);
  vector<vector<double> > Jv, Dv;
  expectation_sufficient_statistics(r, st->th, st->paths, Jv, Dv);
  for (size_t b = 1; b < st->th.n_nodes; ++b)
    for (size_t i = 0; i < 2; ++i) { J[(b - 1) * 2 + i] = Jv[b][i]; D[(b - 1) * 2 + i] = Dv[b][i]; }
}
void ref_indep_suffstats(void *h, double *J, double *D) {
  RefState *st = static_cast<RefState *>(h);
  vector<vector<double> > Jv, Dv;
  compute_sufficient_statistics(st->paths, Jv, Dv);
  for (size_t b = 1; b < st->th.n_nodes; ++b)
    for (size_t i = 0; i < 2; ++i) { J[(b - 1) * 2 + i] = Jv[b][i]; D[(b - 1) * 2 + i] = Dv[b][i]; }
}
void ref_indep_update_paths(void *h, const double *rates) {
  RefState *st = static_cast<RefState *>(h);
  vector<double> r(rates, rates + 2);
  update_paths_indep(r, st->th, st->paths, st->gen);
}
/* M-steps of the site-independent model (IndepSite.cpp:299-360); J/D [(b-1)*2+i];
 * rates, branches in/out; with optimize != 0 the resident paths are rescaled too */
void ref_indep_m_step(void *h, int optimize, const double *J, const double *D, double *rates,
                      double *branches) {
  RefState *st = static_cast<RefState *>(h);
  const size_t N = st->th.n_nodes;
  vector<vector<double> > Jv(N, vector<double>(2, 0.0)), Dv(N, vector<double>(2, 0.0));
  for (size_t b = 1; b < N; ++b)
    for (size_t i = 0; i < 2; ++i) { Jv[b][i] = J[(b - 1) * 2 + i]; Dv[b][i] = D[(b - 1) * 2 + i]; }
  vector<double> r(rates, rates + 2);
  if (!optimize) estimate_rates_indep(Jv, Dv, r, st->th);
  else estimate_rates_and_branches_indep(Jv, Dv, r, st->th, st->paths);
  rates[0] = r[0]; rates[1] = r[1];
  for (size_t b = 0; b < N; ++b) branches[b] = st->th.branches[b];
}

/* ---- forward simulation: the glue of epievo_sim's main (src/prog/epievo_sim.cpp:102-152,
 * 288-352) around the LINKED TripletSampler and EpiEvoModel::sample_state_sequence.
 * Outputs: sequences [n_nodes][n_sites]; global jumps of all nodes concatenated with
 * jump_offsets[n_nodes+1]; returns the number of jumps (or what is needed if > cap). */
uint64_t ref_forward_sim(uint64_t seed, const double *rates, const double *T, int n_nodes,
                         const uint32_t *parent, const double *branches, uint64_t n_sites,
                         uint8_t *sequences, uint64_t *jump_offsets, double *jump_times,
                         uint64_t *jump_positions, uint64_t cap) {
  EpiEvoModel m;
  set_model(m, rates, T);
  std::mt19937 gen(seed);
  state_seq root;
  m.sample_state_sequence(n_sites, gen, root);
  vector<state_seq> seqs(n_nodes, root);
  uint64_t tot = 0;
  jump_offsets[0] = 0;
  jump_offsets[1] = 0;
  for (int node = 1; node < n_nodes; ++node) {
    TripletSampler ts(seqs[parent[node]]);
    double time_value = 0;
    while (time_value < branches[node]) {
      vector<size_t> counts;
      ts.get_triplet_counts(counts);
      const double holding_rate =
          std::inner_product(counts.begin(), counts.end(), m.triplet_rates.begin(), 0.0);
      std::exponential_distribution<double> exp_distr(holding_rate);
      const double holding_time = std::max(exp_distr(gen), std::numeric_limits<double>::min());
      time_value += holding_time;
      if (time_value < branches[node]) {
        vector<double> prob(8, 0.0);
        for (size_t i = 0; i < 8; ++i) prob[i] = counts[i] * m.triplet_rates[i] / holding_rate;
        std::discrete_distribution<size_t> multinom(prob.begin(), prob.end());
        const size_t context = multinom(gen);
        const size_t pos = ts.random_mutate(context, gen);
        if (tot < cap) { jump_times[tot] = time_value; jump_positions[tot] = pos; }
        ++tot;
      }
    }
    ts.get_sequence(seqs[node]);
    jump_offsets[node + 1] = tot;
  }
  for (int node = 0; node < n_nodes; ++node)
    for (uint64_t s = 0; s < n_sites; ++s) sequences[(uint64_t)node * n_sites + s] = seqs[node][s];
  return tot;
}

/* ---- per-function known answers */
int ref_kat_segments(const double *rates, int l_init, uint32_t nl, const double *lj,
                     int r_init, uint32_t nr, const double *rj, double tot_time,
                     double *rate0, double *rate1, uint64_t *trip0, uint64_t *trip1,
                     double *len) {
  array<double, 8> r;
  for (size_t i = 0; i < 8; ++i) r[i] = rates[i];
  Path l(l_init, tot_time, vector<double>(lj, lj + nl));
  Path rr(r_init, tot_time, vector<double>(rj, rj + nr));
  vector<SegmentInfo> seg;
  collect_segment_info(r, l, rr, seg);
  for (size_t k = 0; k < seg.size(); ++k) {
    rate0[k] = seg[k].rate0; rate1[k] = seg[k].rate1;
    trip0[k] = seg[k].trip0; trip1[k] = seg[k].trip1; len[k] = seg[k].len;
  }
  return (int)seg.size();
}

void ref_kat_trans_prob_mat(double r0, double r1, double t, double *P) {
  two_by_two M;
  continuous_time_trans_prob_mat(r0, r1, t, M);
  P[0] = M(0, 0); P[1] = M(0, 1); P[2] = M(1, 0); P[3] = M(1, 1);
}

double ref_kat_get_trans_prob(double r0, double r1, double t, int a, int b) {
  const TwoStateCTMarkovModel ctmm(r0, r1);
  return ctmm.get_trans_prob(t, a, b);
}

void ref_kat_suffstats(int l_init, uint32_t nl, const double *lj, int m_init, uint32_t nm,
                       const double *mj, int r_init, uint32_t nr, const double *rj,
                       double tot_time, double *J, double *D) {
  Path l(l_init, tot_time, vector<double>(lj, lj + nl));
  Path m(m_init, tot_time, vector<double>(mj, mj + nm));
  Path r(r_init, tot_time, vector<double>(rj, rj + nr));
  vector<double> Jv(J, J + 8), Dv(D, D + 8);
  add_sufficient_statistics(l, m, r, Jv, Dv);
  for (size_t i = 0; i < 8; ++i) { J[i] = Jv[i]; D[i] = Dv[i]; }
}

/* closed-form E[J], E[D] of an end-conditioned 2-state path
 * (ContinuousTimeMarkovModel.cpp:168-226): a known-answer source for the means
 * of ANY exact end-conditioned sampler. out: J0,J1,D0,D1 each 4 doubles. */
void ref_kat_expectations(double r0, double r1, double T, double *out) {
  two_by_two J0, J1, D0, D1;
  expectation_J(r0, r1, T, J0, J1);
  expectation_D(r0, r1, T, D0, D1);
  const two_by_two *m[4] = {&J0, &J1, &D0, &D1};
  for (int k = 0; k < 4; ++k) {
    out[4 * k + 0] = (*m[k])(0, 0); out[4 * k + 1] = (*m[k])(0, 1);
    out[4 * k + 2] = (*m[k])(1, 0); out[4 * k + 3] = (*m[k])(1, 1);
  }
}

/* libstdc++ draws, to pin orc_mt_canonical */
void ref_kat_mt_canonical(uint64_t seed, uint64_t n, double *out) {
  std::mt19937 g(seed);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  for (uint64_t i = 0; i < n; ++i) out[i] = u(g);
}


/* ---------------------------------------------------------------- file formats
 * The reference's own writers and readers, so that tests can pin the product's text IO
 * (epv_io.cpp, epv_forward.cpp) byte for byte and value for value.  local_paths: rows are the
 * LINKED operator<<(ostream&, const Path&) (Path.cpp:62-71); the three lines around it
 * ("NODE:<name>", "<site>\t<path>\n") are spelled as the static writers of the un-buildable
 * mains spell them (epievo_est_params_histories.cpp:56-75).  Readers: read_paths (Path.cpp:123-148),
 * read_pathfile_global / write_root_to_pathfile_global / append_to_pathfile_global
 * (GlobalJump.cpp:71-140), read_states_file (epievo_utils.cpp:90-125), PhyloTree's operator>> and
 * Newick_format (PhyloTree.cpp:110-122,189-203).  `names` are '\n'-joined. */
namespace {
vector<std::string> split_names(const char *joined) {
  vector<std::string> out;
  std::stringstream ss(joined);
  std::string t;
  while (std::getline(ss, t, '\n')) out.push_back(t);
  return out;
}
struct LocalCache { vector<std::string> names; vector<vector<Path> > paths; } g_lp;
struct GlobalCache { state_seq root; vector<std::string> names; vector<vector<GlobalJump> > paths; } g_gp;
struct StatesCache { vector<std::string> names; vector<state_seq> states; } g_st;
int put_names(const vector<std::string> &names, char *buf, uint64_t cap) {
  std::string j;
  for (size_t i = 0; i < names.size(); ++i) j += (i ? "\n" : "") + names[i];
  if (j.size() + 1 > cap) return 1;
  std::memcpy(buf, j.c_str(), j.size() + 1);
  return 0;
}
}  // namespace

int ref_write_local_paths(const char *file, int n_nodes, uint64_t n_sites, const char *names,
                          const double *tot_times, const uint8_t *init, const uint64_t *offsets,
                          const double *jumps) {
  const vector<std::string> nm = split_names(names);
  if ((int)nm.size() != n_nodes) return 1;
  {
    std::ofstream out(file);
    if (!out) return 2;
    out << "NODE:" << nm[0] << std::endl;
  }
  for (int b = 1; b < n_nodes; ++b) {
    std::ofstream out(file, std::ofstream::app);
    if (!out) return 2;
    out << "NODE:" << nm[b] << std::endl;
    for (uint64_t s = 0; s < n_sites; ++s) {
      const uint64_t e = (uint64_t)(b - 1) * n_sites + s;
      const Path p(init[e] != 0, tot_times[b], vector<double>(jumps + offsets[e], jumps + offsets[e + 1]));
      out << s << '\t' << p << '\n';
    }
  }
  return 0;
}

int ref_read_local_paths(const char *file, int *n_nodes, uint64_t *n_sites, uint64_t *total_jumps) {
  g_lp = LocalCache();
  try { read_paths(file, g_lp.names, g_lp.paths); } catch (const std::exception &) { return 1; }
  *n_nodes = (int)g_lp.paths.size();
  *n_sites = g_lp.paths.size() > 1 ? g_lp.paths[1].size() : 0;
  uint64_t tot = 0;
  for (size_t b = 1; b < g_lp.paths.size(); ++b)
    for (const Path &p : g_lp.paths[b]) tot += p.jumps.size();
  *total_jumps = tot;
  return 0;
}
int ref_local_paths_copy(uint8_t *init, double *tot_times, uint64_t *offsets, double *jumps, char *names,
                         uint64_t names_cap) {
  uint64_t e = 0, at = 0;
  for (size_t b = 1; b < g_lp.paths.size(); ++b) {
    tot_times[b] = g_lp.paths[b].empty() ? 0.0 : g_lp.paths[b][0].tot_time;
    for (const Path &p : g_lp.paths[b]) {
      init[e] = p.init_state;
      offsets[e++] = at;
      for (double t : p.jumps) jumps[at++] = t;
    }
  }
  offsets[e] = at;
  return put_names(g_lp.names, names, names_cap);
}

int ref_write_global_jumps(const char *file, int n_nodes, const char *names, uint64_t n_sites,
                           const uint8_t *root, const uint64_t *node_offsets, const double *times,
                           const uint64_t *positions) {
  const vector<std::string> nm = split_names(names);
  if ((int)nm.size() != n_nodes) return 1;
  try {
    state_seq r(root, root + n_sites);
    write_root_to_pathfile_global(file, nm[0], r);
    for (int b = 1; b < n_nodes; ++b) {
      vector<GlobalJump> v;
      for (uint64_t i = node_offsets[b]; i < node_offsets[b + 1]; ++i) v.push_back(GlobalJump(times[i], positions[i]));
      append_to_pathfile_global(file, nm[b], v);
    }
  } catch (const std::exception &) { return 2; }
  return 0;
}
int ref_read_global_jumps(const char *file, int *n_nodes, uint64_t *n_sites, uint64_t *total) {
  g_gp = GlobalCache();
  try { read_pathfile_global(file, g_gp.root, g_gp.names, g_gp.paths); } catch (const std::exception &) { return 1; }
  *n_nodes = (int)g_gp.paths.size();
  *n_sites = g_gp.root.size();
  uint64_t tot = 0;
  for (const vector<GlobalJump> &v : g_gp.paths) tot += v.size();
  *total = tot;
  return 0;
}
int ref_global_jumps_copy(uint8_t *root, uint64_t *node_offsets, double *times, uint64_t *positions,
                          char *names, uint64_t names_cap) {
  for (size_t i = 0; i < g_gp.root.size(); ++i) root[i] = g_gp.root[i];
  uint64_t at = 0;
  for (size_t b = 0; b < g_gp.paths.size(); ++b) {
    node_offsets[b] = at;
    for (const GlobalJump &j : g_gp.paths[b]) { times[at] = j.timepoint; positions[at] = j.position; ++at; }
  }
  node_offsets[g_gp.paths.size()] = at;
  return put_names(g_gp.names, names, names_cap);
}

int ref_read_states(const char *file, int *n_seqs, uint64_t *n_sites) {
  g_st = StatesCache();
  try { read_states_file(file, g_st.names, g_st.states); } catch (const std::exception &) { return 1; }
  *n_seqs = (int)g_st.states.size();
  *n_sites = g_st.states.empty() ? 0 : g_st.states[0].size();
  return 0;
}
int ref_states_copy(uint8_t *states, char *names, uint64_t names_cap) {
  uint64_t at = 0;
  for (const state_seq &sq : g_st.states)
    for (size_t i = 0; i < sq.size(); ++i) states[at++] = sq[i];
  return put_names(g_st.names, names, names_cap);
}

/* parse Newick text with the reference and print it back (PhyloTree::Newick_format); also the
 * pre-order arrays TreeHelper derives (TreeHelper.cpp:43-51) */
int ref_newick_roundtrip(const char *text, char *out, uint64_t cap, int *n_nodes, uint32_t *parent,
                         uint32_t *subtree, double *branches, char *names, uint64_t names_cap) {
  PhyloTreePreorder t;
  std::istringstream in(text);
  if (!(in >> t)) return 1;
  const std::string nw = t.Newick_format();
  if (nw.size() + 1 > cap) return 2;
  std::memcpy(out, nw.c_str(), nw.size() + 1);
  TreeHelper th(t);
  *n_nodes = (int)th.n_nodes;
  for (size_t i = 0; i < th.n_nodes; ++i) {
    parent[i] = (uint32_t)th.parent_ids[i];
    subtree[i] = (uint32_t)th.subtree_sizes[i];
    branches[i] = th.branches[i];
  }
  return put_names(th.node_names, names, names_cap);
}

}  // extern "C"
