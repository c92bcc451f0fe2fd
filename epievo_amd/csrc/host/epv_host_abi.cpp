// epv_host_abi.cpp -- flat C face of the host-side library (libepv_host.so) so that
// Python tests / bench.py can drive the same C++ host code the CLI programs use.
// No GPU code here; no exceptions cross the boundary (non-zero return = failure,
// message via epvh_last_error).
#include <cstring>
#include <string>

#include "epv_model.hpp"
#include "epv_sim.hpp"
#include "epv_io.hpp"
#include "epv_indep.hpp"
#include "epv_forward.hpp"

namespace {
thread_local std::string g_err;
void put_model(const epv::Model &m, double *rates, double *T, double *baseline) {
  for (int i = 0; i < 8; ++i) rates[i] = m.rates[i];
  for (int i = 0; i < 4; ++i) { T[i] = m.T[i]; baseline[i] = m.baseline[i]; }
}
void put_text(const std::string &s, char *buf, int len) {
  if (!buf || len <= 0) return;
  std::strncpy(buf, s.c_str(), (size_t)len - 1);
  buf[len - 1] = '\0';
}
}  // namespace

#define EPVH_API extern "C" __attribute__((visibility("default")))

EPVH_API const char *epvh_last_error() { return g_err.c_str(); }

// read_model (+ optional scale_triplet_rates), est_params_histories.cpp:169-171
EPVH_API int epvh_model_read(const char *param_file, int scale, double *rates, double *T,
                             double *baseline) {
  try {
    epv::Model m = epv::Model::read(param_file);
    if (scale) m.scale_triplet_rates();
    put_model(m, rates, T, baseline);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

EPVH_API double epvh_rate_scaling_factor(const double *rates) {
  std::array<double, 8> r;
  for (int i = 0; i < 8; ++i) r[i] = rates[i];
  return epv::rate_scaling_factor(r);
}

// the M-step exactly as the EM driver sequences it (est_params_histories.cpp:253-263).
// rates/branches are in-out; T, baseline, param_text are outputs; returns 0 / non-zero.
EPVH_API int epvh_m_step(int optimize_branches, int n_nodes, const double *J, const double *D,
                         double *rates, double *T, double *baseline, double *branches,
                         double *llh, char *param_text, int param_text_len) {
  try {
    epv::Model m;
    std::array<double, 8> r;
    for (int i = 0; i < 8; ++i) r[i] = rates[i];
    m.rebuild_from_triplet_rates(r);
    std::vector<double> br(branches, branches + n_nodes);
    if (!optimize_branches) {
      *llh = epv::estimate_rates(1e-10, n_nodes, J, D, m);
      epv::set_one_change_per_site_per_unit_time(m.rates, br);
    } else {
      *llh = epv::estimate_rates_and_branches(1e-10, n_nodes, J, D, br, m);
    }
    put_model(m, rates, T, baseline);
    for (int b = 0; b < n_nodes; ++b) branches[b] = br[b];
    put_text(m.format_for_param_file(), param_text, param_text_len);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// ---- forward simulation of synthetic histories
EPVH_API void *epvh_simulate(const double *rates, const double *T, int n_nodes,
                             const uint32_t *parent, const double *branches,
                             uint64_t n_sites, uint64_t seed) {
  try {
    epv::Model m;
    for (int i = 0; i < 8; ++i) m.rates[i] = rates[i];
    for (int i = 0; i < 4; ++i) m.T[i] = T[i];
    epv::FlatPaths *fp = new epv::FlatPaths(
        epv::simulate_histories(m, n_nodes, parent, branches, n_sites, seed));
    return fp;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
EPVH_API uint64_t epvh_paths_total_jumps(void *h) {
  return static_cast<epv::FlatPaths *>(h)->jumps.size();
}
EPVH_API uint64_t epvh_paths_n_sites(void *h) { return static_cast<epv::FlatPaths *>(h)->n_sites; }
EPVH_API int epvh_paths_n_nodes(void *h) { return static_cast<epv::FlatPaths *>(h)->n_nodes; }
EPVH_API void epvh_paths_copy(void *h, uint8_t *init, uint64_t *offsets, double *jumps) {
  const epv::FlatPaths *fp = static_cast<epv::FlatPaths *>(h);
  std::memcpy(init, fp->init.data(), fp->init.size());
  std::memcpy(offsets, fp->offsets.data(), fp->offsets.size() * sizeof(uint64_t));
  if (!fp->jumps.empty())
    std::memcpy(jumps, fp->jumps.data(), fp->jumps.size() * sizeof(double));
}
EPVH_API void epvh_paths_free(void *h) { delete static_cast<epv::FlatPaths *>(h); }

// ---- file formats (local_paths, Newick tree)
EPVH_API void *epvh_read_paths(const char *path_file, char *names_buf, int names_len,
                               double *tot_times, int max_nodes) {
  try {
    std::vector<std::string> names;
    std::vector<double> tt;
    epv::FlatPaths *fp = new epv::FlatPaths(epv::read_local_paths(path_file, names, tt));
    std::string joined;
    for (size_t i = 0; i < names.size(); ++i) joined += (i ? "\n" : "") + names[i];
    put_text(joined, names_buf, names_len);
    for (int b = 0; b < fp->n_nodes && b < max_nodes; ++b) tot_times[b] = tt[b];
    return fp;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

EPVH_API int epvh_write_paths(const char *path_file, const char *names_joined, int n_nodes,
                              uint64_t n_sites, const double *tot_times, const uint8_t *init,
                              const uint64_t *offsets, const double *jumps) {
  try {
    std::vector<std::string> names;
    std::string cur;
    for (const char *p = names_joined;; ++p) {
      if (*p == '\n' || *p == '\0') { names.push_back(cur); cur.clear(); if (!*p) break; }
      else cur.push_back(*p);
    }
    epv::write_local_paths(path_file, names, n_nodes, n_sites, tot_times, init, offsets, jumps);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// parse a Newick file into pre-order arrays; returns n_nodes (<= max_nodes) or -1
EPVH_API int epvh_read_tree(const char *tree_file, int max_nodes, uint32_t *subtree,
                            uint32_t *parent, double *branches, char *names_buf,
                            int names_len) {
  try {
    epv::Tree t = epv::Tree::read(tree_file);
    const int n = (int)t.subtree_sizes.size();
    if (n > max_nodes) { g_err = "tree too large"; return -1; }
    std::string joined;
    for (int i = 0; i < n; ++i) {
      subtree[i] = t.subtree_sizes[i];
      parent[i] = t.parent_ids[i];
      branches[i] = t.branches[i];
      joined += (i ? "\n" : "") + t.node_names[i];
    }
    put_text(joined, names_buf, names_len);
    return n;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// the tree of a Newick file printed back (Tree::newick) -- the -t output of the drop-in CLIs
EPVH_API int epvh_tree_newick(const char *tree_file, char *out, int out_len) {
  try {
    put_text(epv::Tree::read(tree_file).newick(), out, out_len);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// ---- global_jumps and states files (epievo_sim's outputs)
namespace {
std::vector<std::string> split_lines(const char *joined) {
  std::vector<std::string> names;
  std::string cur;
  for (const char *p = joined;; ++p) {
    if (*p == '\n' || *p == '\0') { names.push_back(cur); cur.clear(); if (!*p) break; }
    else cur.push_back(*p);
  }
  return names;
}
struct GlobalFile { std::vector<uint8_t> root; std::vector<std::string> names; std::vector<std::vector<epv::GlobalJump>> paths; };
struct StatesFile { std::vector<std::string> names; std::vector<std::vector<uint8_t>> states; };
}  // namespace

EPVH_API int epvh_write_global_jumps(const char *file, int n_nodes, const char *names_joined, uint64_t n_sites,
                                     const uint8_t *root, const uint64_t *node_offsets, const double *times,
                                     const uint64_t *positions) {
  try {
    std::vector<std::vector<epv::GlobalJump>> paths(n_nodes);
    for (int b = 1; b < n_nodes; ++b)
      for (uint64_t i = node_offsets[b]; i < node_offsets[b + 1]; ++i) paths[b].push_back({times[i], (size_t)positions[i]});
    epv::write_global_jumps(file, split_lines(names_joined), std::vector<uint8_t>(root, root + n_sites), paths);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
EPVH_API void *epvh_read_global_jumps(const char *file, int *n_nodes, uint64_t *n_sites, uint64_t *total) {
  try {
    GlobalFile *g = new GlobalFile();
    epv::read_global_jumps(file, g->root, g->names, g->paths);
    *n_nodes = (int)g->paths.size();
    *n_sites = g->root.size();
    uint64_t tot = 0;
    for (const auto &v : g->paths) tot += v.size();
    *total = tot;
    return g;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
EPVH_API void epvh_global_jumps_copy(void *h, uint8_t *root, uint64_t *node_offsets, double *times,
                                     uint64_t *positions, char *names, int names_len) {
  GlobalFile *g = static_cast<GlobalFile *>(h);
  std::memcpy(root, g->root.data(), g->root.size());
  uint64_t at = 0;
  for (size_t b = 0; b < g->paths.size(); ++b) {
    node_offsets[b] = at;
    for (const epv::GlobalJump &j : g->paths[b]) { times[at] = j.timepoint; positions[at] = j.position; ++at; }
  }
  node_offsets[g->paths.size()] = at;
  std::string joined;
  for (size_t i = 0; i < g->names.size(); ++i) joined += (i ? "\n" : "") + g->names[i];
  put_text(joined, names, names_len);
  delete g;
}

// states[seq][site] row-major; `names_joined` = the column names of the header line
EPVH_API int epvh_write_states(const char *file, int only_leaves, int n_nodes, const uint32_t *subtree,
                               const char *names_joined, uint64_t n_sites, const uint8_t *states) {
  try {
    epv::Tree t;
    t.subtree_sizes.assign(subtree, subtree + n_nodes);
    t.parent_ids.assign(n_nodes, 0);
    t.branches.assign(n_nodes, 0.0);
    t.node_names = split_lines(names_joined);
    std::vector<std::vector<uint8_t>> seqs(n_nodes);
    for (int b = 0; b < n_nodes; ++b) seqs[b].assign(states + (uint64_t)b * n_sites, states + (uint64_t)(b + 1) * n_sites);
    epv::write_states(file, only_leaves != 0, t, seqs);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
EPVH_API void *epvh_read_states(const char *file, int *n_seqs, uint64_t *n_sites) {
  try {
    StatesFile *f = new StatesFile();
    epv::read_states_file(file, f->names, f->states);
    *n_seqs = (int)f->states.size();
    *n_sites = f->states.empty() ? 0 : f->states[0].size();
    return f;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
EPVH_API void epvh_states_copy(void *h, uint8_t *states, char *names, int names_len) {
  StatesFile *f = static_cast<StatesFile *>(h);
  uint64_t at = 0;
  for (const auto &sq : f->states) { std::memcpy(states + at, sq.data(), sq.size()); at += sq.size(); }
  std::string joined;
  for (size_t i = 0; i < f->names.size(); ++i) joined += (i ? "\n" : "") + f->names[i];
  put_text(joined, names, names_len);
  delete f;
}

// ---- site-independent stage of epievo_initialization (host parts)
EPVH_API int epvh_indep_m_step(int optimize_branches, int n_nodes, const double *J, const double *D,
                               double *rates, double *branches) {
  try {
    if (!optimize_branches) {
      epv::estimate_rates_indep(n_nodes, J, D, rates);
    } else {
      std::vector<double> br(branches, branches + n_nodes);
      epv::estimate_rates_and_branches_indep(n_nodes, J, D, rates, br);
      for (int b = 0; b < n_nodes; ++b) branches[b] = br[b];
    }
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

EPVH_API int epvh_model_from_indep_rates(const double *rates2, double *rates, double *T, double *baseline) {
  try {
    put_model(epv::model_from_indep_rates(rates2), rates, T, baseline);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// states: [n_nodes][n_sites] row-major, updated in place with the drawn internal states
EPVH_API void *epvh_initialize_paths_heuristic(uint64_t seed, int n_nodes, const uint32_t *subtree,
                                               const uint32_t *parent, const double *branches,
                                               uint64_t n_sites, uint8_t *states) {
  try {
    epv::Tree t;
    t.subtree_sizes.assign(subtree, subtree + n_nodes);
    t.parent_ids.assign(parent, parent + n_nodes);
    t.branches.assign(branches, branches + n_nodes);
    t.node_names.assign(n_nodes, "");
    std::vector<std::vector<uint8_t>> st(n_nodes);
    for (int b = 0; b < n_nodes; ++b) st[b].assign(states + (uint64_t)b * n_sites, states + (uint64_t)(b + 1) * n_sites);
    epv::FlatPaths *fp = new epv::FlatPaths(epv::initialize_paths_heuristic(seed, t, st));
    for (int b = 0; b < n_nodes; ++b) std::memcpy(states + (uint64_t)b * n_sites, st[b].data(), n_sites);
    return fp;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

// ---- forward simulation (epievo_sim): same outputs as the test shim's ref_forward_sim
EPVH_API uint64_t epvh_forward_sim(uint64_t seed, const double *rates, const double *T, int n_nodes,
                                   const uint32_t *parent, const double *branches, uint64_t n_sites,
                                   uint8_t *sequences, uint64_t *jump_offsets, double *jump_times,
                                   uint64_t *jump_positions, uint64_t cap) {
  epv::Model m;
  for (int i = 0; i < 8; ++i) m.rates[i] = rates[i];
  for (int i = 0; i < 4; ++i) m.T[i] = T[i];
  epv::Tree th;
  th.parent_ids.assign(parent, parent + n_nodes);
  th.branches.assign(branches, branches + n_nodes);
  th.subtree_sizes.assign(n_nodes, 1);   // only n_nodes() is needed here
  th.node_names.assign(n_nodes, "");
  std::mt19937 gen(seed);
  std::vector<uint8_t> root;
  epv::sample_root(m, n_sites, gen, root);
  std::vector<std::vector<uint8_t>> seqs;
  std::vector<std::vector<epv::GlobalJump>> paths;
  std::vector<size_t> events;
  epv::simulate_tree(m, th, root, gen, seqs, paths, events);
  uint64_t tot = 0;
  jump_offsets[0] = 0;
  for (int node = 0; node < n_nodes; ++node) {
    for (const epv::GlobalJump &j : paths[node]) {
      if (tot < cap) { jump_times[tot] = j.timepoint; jump_positions[tot] = j.position; }
      ++tot;
    }
    jump_offsets[node + 1] = tot;
    std::memcpy(sequences + (uint64_t)node * n_sites, seqs[node].data(), n_sites);
  }
  return tot;
}
