// epv_indep.cpp -- see epv_indep.hpp
#include "epv_indep.hpp"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <functional>
#include <random>
#include <sstream>
#include <stdexcept>

namespace epv {

void estimate_rates_indep(int n_nodes, const double *J, const double *D, double rates[2]) {
  double J_sum[2] = {0.0, 0.0}, D_sum[2] = {0.0, 0.0};
  for (int b = 1; b < n_nodes; ++b)
    for (int i = 0; i < 2; ++i) {
      J_sum[i] += J[(b - 1) * 2 + i];
      D_sum[i] += D[(b - 1) * 2 + i];
    }
  if (D_sum[0] > 0) rates[0] = std::max(J_sum[0] / D_sum[0], 10e-6);
  if (D_sum[1] > 0) rates[1] = std::max(J_sum[1] / D_sum[1], 10e-6);
}

void estimate_rates_and_branches_indep(int n_nodes, const double *J, const double *D, double rates[2],
                                       std::vector<double> &branches) {
  estimate_rates_indep(n_nodes, J, D, rates);
  for (int b = 1; b < n_nodes; ++b)
    branches[b] *= (J[(b - 1) * 2] + J[(b - 1) * 2 + 1]) /
                   (D[(b - 1) * 2] * rates[0] + D[(b - 1) * 2 + 1] * rates[1]);
  // indep_rate_scaling_factor (IndepSite.cpp:320-326): one change per site per unit time
  const double pi[2] = {rates[1] / (rates[0] + rates[1]), rates[0] / (rates[0] + rates[1])};
  const double scale_factor = pi[0] * rates[0] + pi[1] * rates[1];
  for (double &b : branches) b = b * scale_factor;
  rates[0] = rates[0] / scale_factor;
  rates[1] = rates[1] / scale_factor;
}

Model model_from_indep_rates(const double rates[2]) {
  std::array<double, 8> r;
  for (int i = 0; i < 8; ++i) r[i] = rates[(i / 2) % 2];
  Model m;
  m.rebuild_from_triplet_rates(r);
  return m;
}

std::vector<std::vector<uint8_t>> read_states_for_tree(const std::string &states_file, const Tree &th) {
  std::ifstream in(states_file);
  if (!in) throw std::runtime_error("bad states file: " + states_file);
  std::string buffer;
  if (!std::getline(in, buffer)) throw std::runtime_error("cannot read nodes line in: " + states_file);
  std::istringstream nodes_iss(buffer);
  std::vector<std::string> names;
  std::string nm;
  while (nodes_iss >> nm) names.push_back(nm);
  if (names.size() < 2) throw std::runtime_error("fewer than 2 nodes names in: " + states_file);
  if (names.front()[0] == '#') {
    if (names.front().length() == 1) names.erase(names.begin());
    else names.front() = names.front().substr(1);
  }
  const int N = th.n_nodes();
  std::vector<int> idx_in_tree(names.size(), N);
  for (int node = 0; node < N; ++node) {
    auto it = std::find(names.begin(), names.end(), th.node_names[node]);
    if (it != names.end()) idx_in_tree[it - names.begin()] = node;
    else if (th.is_leaf(node)) throw std::runtime_error("no data in leaf node: " + th.node_names[node]);
  }
  std::vector<std::vector<uint8_t>> states(N);
  size_t site_count = 0;
  while (std::getline(in, buffer)) {
    std::istringstream iss(buffer);
    size_t site_index = 0;
    iss >> site_index;
    size_t k = 0;
    int v = 0;
    while (k < names.size() && iss >> v) {
      if (idx_in_tree[k] < N) states[idx_in_tree[k]].push_back(v != 0);
      ++k;
    }
    if (k < names.size())
      throw std::runtime_error("inconsistent number of states: " + std::to_string(k) + "/" +
                               std::to_string(names.size()));
    ++site_count;
  }
  if (site_count == 0) throw std::runtime_error("no sites read from states file: " + states_file);
  for (auto &s : states)
    if (s.size() < site_count) s.resize(site_count, 0);
  return states;
}

FlatPaths initialize_paths_heuristic(uint64_t seed, const Tree &th,
                                     std::vector<std::vector<uint8_t>> &states) {
  const int N = th.n_nodes();
  const uint64_t n = states.front().size();
  std::mt19937 gen(seed);
  auto unif = std::bind(std::uniform_real_distribution<double>(0.0, 1.0), std::ref(gen));
  // per (node, site): at most one jump
  std::vector<std::vector<double>> jump(N, std::vector<double>(n, -1.0));
  std::vector<std::vector<uint8_t>> init(N, std::vector<uint8_t>(n, 0));
  uint8_t child_states[64];
  for (int node = N - 1; node >= 0; --node) {
    if (th.is_leaf(node)) continue;
    for (uint64_t s = 0; s < n; ++s) {
      size_t n_ch = 0;
      for (uint32_t c = 1; c < th.subtree_sizes[node]; c += th.subtree_sizes[node + c])
        if (n_ch < 64) child_states[n_ch++] = states[node + c][s];
      if (node == 0) states[node][s] = states[0][s];
      else states[node][s] = child_states[(size_t)std::floor(unif() * n_ch)];
      for (uint32_t c = 1; c < th.subtree_sizes[node]; c += th.subtree_sizes[node + c]) {
        const int ch = node + (int)c;
        init[ch][s] = states[node][s];
        if (states[ch][s] != init[ch][s]) jump[ch][s] = unif() * th.branches[ch];
      }
    }
  }
  FlatPaths fp;
  fp.n_sites = n;
  fp.n_nodes = N;
  for (int b = 1; b < N; ++b)
    for (uint64_t s = 0; s < n; ++s) {
      fp.init.push_back(init[b][s]);
      fp.offsets.push_back(fp.jumps.size());
      if (jump[b][s] >= 0.0) fp.jumps.push_back(jump[b][s]);
    }
  fp.offsets.push_back(fp.jumps.size());
  return fp;
}

}  // namespace epv
