// epv_forward.cpp -- see epv_forward.hpp
#include "epv_forward.hpp"

#include <atomic>
#include <thread>

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace epv {

namespace {
inline size_t ctx3(const std::vector<uint8_t> &s, size_t i) { return 4u * s[i - 1] + 2u * s[i] + s[i + 1]; }
}

// counting sort of the interior positions by context, filled from the back of each block
// exactly as TripletSampler.cpp:37-70 does (the initial order decides which position a
// given uniform draw picks)
ContextIndex::ContextIndex(const std::vector<uint8_t> &seq)
    : order_(seq.size() - 2), slot_(seq.size()), start_(8, 0), first_(seq.front()), last_(seq.back()) {
  const size_t n = seq.size();
  for (size_t i = 1; i + 1 < n; ++i) ++start_[ctx3(seq, i)];
  for (size_t c = 1; c < 8; ++c) start_[c] += start_[c - 1];   // inclusive prefix sums
  for (size_t i = 1; i + 1 < n; ++i) {
    const size_t x = --start_[ctx3(seq, i)];
    order_[x] = i;
    slot_[i] = x;
  }
  start_.push_back(n - 2);
}

size_t ContextIndex::context_of(size_t pos) const {
  const size_t x = slot_[pos];
  size_t c = 0;
  while (start_[c + 1] <= x) ++c;   // TripletSampler.cpp:129-142
  return c;
}

// move `pos` from block `from` to block `to` by walking the block boundaries one context
// at a time, swapping it to the edge of each block it leaves (TripletSampler.cpp:85-116)
void ContextIndex::move(size_t pos, size_t from, size_t to) {
  size_t prev = slot_[pos];
  auto swap_slots = [&](size_t a, size_t b) {
    std::swap(slot_[order_[a]], slot_[order_[b]]);
    std::swap(order_[a], order_[b]);
  };
  if (from > to) {
    while (from > to) {
      const size_t cur = start_[from];   // first element of the block being left
      swap_slots(prev, cur);
      prev = cur;
      ++start_[from--];
    }
  } else {
    while (from < to) {
      --start_[++from];
      const size_t cur = start_[from];
      swap_slots(prev, cur);
      prev = cur;
    }
  }
}

void ContextIndex::mutate(size_t pos, size_t context) {
  move(pos, context, context ^ 2u);
  const size_t n = slot_.size();
  if (pos - 1 > 0) {
    const size_t c = context_of(pos - 1);
    move(pos - 1, c, c ^ 1u);
  }
  if (pos + 1 < n - 1) {
    const size_t c = context_of(pos + 1);
    move(pos + 1, c, c ^ 4u);
  }
}

size_t ContextIndex::random_mutate(size_t context, std::mt19937 &gen) {
  std::uniform_int_distribution<size_t> pick(start_[context], start_[context + 1] - 1);
  const size_t pos = order_[pick(gen)];
  mutate(pos, context);
  return pos;
}

void ContextIndex::sequence(std::vector<uint8_t> &seq) const {
  seq.assign(slot_.size(), 1);
  size_t x = 0;
  for (size_t c = 0; c < 8; ++c)
    for (; x < start_[c + 1]; ++x) seq[order_[x]] = (c >> 1) & 1u;
  seq.front() = first_;
  seq.back() = last_;
}

void sample_root(const Model &m, size_t n_sites, std::mt19937 &gen, std::vector<uint8_t> &seq) {
  seq.assign(n_sites, 1);
  const double T00 = m.T[0], T11 = m.T[3];
  const double pi1 = (1.0 - T00) / (2.0 - T11 - T00);
  std::uniform_real_distribution<double> unif(0.0, 1.0);
  seq[0] = (unif(gen) < pi1);
  for (size_t i = 1; i < n_sites; ++i) {
    const double r = unif(gen);
    const double p = seq[i - 1] ? T11 : T00;
    seq[i] = (r <= p) ? seq[i - 1] : (uint8_t)!seq[i - 1];
  }
}

void simulate_branch(const Model &m, double branch_len, std::mt19937 &gen, ContextIndex &index,
                     std::vector<GlobalJump> &path, std::vector<size_t> &events) {
  double time_value = 0;
  while (time_value < branch_len) {
    // one call of sample_jump (epievo_sim.cpp:102-152)
    double holding_rate = 0.0;
    for (size_t c = 0; c < 8; ++c) holding_rate = holding_rate + (double)index.count(c) * m.rates[c];
    std::exponential_distribution<double> exp_distr(holding_rate);
    const double holding_time = std::max(exp_distr(gen), std::numeric_limits<double>::min());
    time_value += holding_time;
    if (time_value < branch_len) {
      std::vector<double> prob(8, 0.0);
      for (size_t c = 0; c < 8; ++c) prob[c] = index.count(c) * m.rates[c] / holding_rate;
      std::discrete_distribution<size_t> multinom(prob.begin(), prob.end());
      const size_t context = multinom(gen);
      ++events[context];
      const size_t pos = index.random_mutate(context, gen);
      path.push_back(GlobalJump{time_value, pos});
    }
  }
}

void simulate_tree(const Model &m, const Tree &th, const std::vector<uint8_t> &root_seq,
                   std::mt19937 &gen, std::vector<std::vector<uint8_t>> &sequences,
                   std::vector<std::vector<GlobalJump>> &paths, std::vector<size_t> &events) {
  const int N = th.n_nodes();
  sequences.assign(N, root_seq);
  paths.assign(N, {});
  events.assign(8, 0);
  for (int node = 1; node < N; ++node) {
    ContextIndex index(sequences[th.parent_ids[node]]);
    simulate_branch(m, th.branches[node], gen, index, paths[node], events);
    index.sequence(sequences[node]);
  }
}

// Sibling subtrees in parallel (new; the reference threads ONE std::mt19937 through the branches
// in pre-order, epievo_sim.cpp:329-352, so its stream cannot be split).  Once a node's end
// sequence is known its children are independent: every branch gets its own generator seeded
// from (seed, node index), a node's children run on their own threads while the budget lasts,
// and the result depends on the seed and the tree only -- not on the number of threads.  The law
// of the process is the reference's (same sample_jump per event); the numbers differ from the
// sequential mode's, which stays the default and the one pinned to the linked TripletSampler.
namespace {
struct ParSim {
  const Model &m;
  const Tree &th;
  uint64_t seed;
  std::vector<std::vector<uint8_t>> &sequences;
  std::vector<std::vector<GlobalJump>> &paths;
  std::vector<std::vector<size_t>> events;   // per node
  std::atomic<int> budget;
  ParSim(const Model &m_, const Tree &th_, uint64_t seed_, std::vector<std::vector<uint8_t>> &sq,
         std::vector<std::vector<GlobalJump>> &pp, int threads)
      : m(m_), th(th_), seed(seed_), sequences(sq), paths(pp), events(th_.n_nodes(), std::vector<size_t>(8, 0)),
        budget(threads - 1) {}
  void subtree(int node) {
    if (node > 0) {
      std::seed_seq sq{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)node, 0x65707673u};
      std::mt19937 gen(sq);
      ContextIndex index(sequences[th.parent_ids[node]]);
      simulate_branch(m, th.branches[node], gen, index, paths[node], events[node]);
      index.sequence(sequences[node]);
    }
    std::vector<std::thread> spawned;
    std::vector<int> kids;
    for (uint32_t c = 1; c < th.subtree_sizes[node]; c += th.subtree_sizes[node + c]) kids.push_back(node + (int)c);
    for (size_t i = 0; i < kids.size(); ++i) {
      const int k = kids[i];
      if (i + 1 < kids.size() && budget.fetch_sub(1) > 0) {
        spawned.emplace_back([this, k] { subtree(k); budget.fetch_add(1); });
      } else {
        if (i + 1 < kids.size()) budget.fetch_add(1);   // the failed reservation
        subtree(k);
      }
    }
    for (std::thread &t : spawned) t.join();
  }
};
}  // namespace

void simulate_tree_parallel(const Model &m, const Tree &th, const std::vector<uint8_t> &root_seq, uint64_t seed,
                            int threads, std::vector<std::vector<uint8_t>> &sequences,
                            std::vector<std::vector<GlobalJump>> &paths, std::vector<size_t> &events) {
  const int N = th.n_nodes();
  sequences.assign(N, {});
  sequences[0] = root_seq;
  paths.assign(N, {});
  ParSim ps(m, th, seed, sequences, paths, std::max(1, threads));
  ps.subtree(0);
  events.assign(8, 0);
  for (int node = 1; node < N; ++node)
    for (size_t c = 0; c < 8; ++c) events[c] += ps.events[node][c];
}

void write_global_jumps(const std::string &file, const std::vector<std::string> &node_names,
                        const std::vector<uint8_t> &root, const std::vector<std::vector<GlobalJump>> &paths) {
  std::FILE *f = file.empty() ? stdout : std::fopen(file.c_str(), "w");
  if (!f) throw std::runtime_error("bad output file: " + file);
  std::fprintf(f, "ROOT:%s\n", node_names[0].c_str());
  std::string bits(root.size(), '0');
  for (size_t i = 0; i < root.size(); ++i) bits[i] = root[i] ? '1' : '0';
  std::fprintf(f, "%s\n", bits.c_str());
  for (size_t node = 1; node < paths.size(); ++node) {
    std::fprintf(f, "NODE:%s\n", node_names[node].c_str());
    for (const GlobalJump &j : paths[node]) std::fprintf(f, "%.17g\t%zu\n", j.timepoint, j.position);
  }
  if (f != stdout) std::fclose(f);
}

void read_global_jumps(const std::string &file, std::vector<uint8_t> &root, std::vector<std::string> &node_names,
                       std::vector<std::vector<GlobalJump>> &paths) {
  std::ifstream in(file);
  if (!in) throw std::runtime_error("cannot read: " + file);
  std::string line;
  std::getline(in, line);
  if (line.size() <= 4 || line.compare(0, 4, "ROOT") != 0) throw std::runtime_error("cannot read root seq: " + file);
  node_names.push_back(line.substr(line.find(':') + 1));
  std::getline(in, line);
  root.clear();
  for (char c : line) root.push_back(c == '1');
  paths.assign(1, {});
  while (std::getline(in, line)) {
    if (line.size() > 4 && line.compare(0, 4, "NODE") == 0) {
      node_names.push_back(line.substr(line.find(':') + 1));
      paths.emplace_back();
    } else {
      GlobalJump j;
      std::istringstream iss(line);
      if (!(iss >> j.timepoint >> j.position)) throw std::runtime_error("bad line: " + line);
      paths.back().push_back(j);
    }
  }
}

void write_states(const std::string &file, bool only_leaves, const Tree &th,
                  const std::vector<std::vector<uint8_t>> &sequences) {
  std::FILE *f = std::fopen(file.c_str(), "w");
  if (!f) throw std::runtime_error("bad output file: " + file);
  std::fputc('#', f);
  bool first = true;
  for (int i = 0; i < th.n_nodes(); ++i)
    if (!only_leaves || th.is_leaf(i)) {
      if (!first) std::fputc('\t', f);
      first = false;
      std::fputs(th.node_names[i].c_str(), f);
    }
  std::fputc('\n', f);
  const size_t n = sequences.front().size();
  for (size_t s = 0; s < n; ++s) {
    std::fprintf(f, "%zu", s);
    for (int j = 0; j < th.n_nodes(); ++j)
      if (!only_leaves || th.is_leaf(j)) std::fprintf(f, "\t%d", (int)sequences[j][s]);
    std::fputc('\n', f);
  }
  std::fclose(f);
}

FlatPaths global_to_local(const Tree &th, const std::vector<std::vector<uint8_t>> &states,
                          const std::vector<std::vector<GlobalJump>> &paths) {
  const int N = th.n_nodes();
  const uint64_t n = states.front().size();
  FlatPaths fp;
  fp.n_sites = n;
  fp.n_nodes = N;
  std::vector<std::vector<double>> by_site(n);
  for (int node = 1; node < N; ++node) {
    for (auto &v : by_site) v.clear();
    for (const GlobalJump &j : paths[node]) by_site[j.position].push_back(j.timepoint);
    const std::vector<uint8_t> &start = states[th.parent_ids[node]];
    for (uint64_t s = 0; s < n; ++s) {
      fp.init.push_back(start[s]);
      fp.offsets.push_back(fp.jumps.size());
      fp.jumps.insert(fp.jumps.end(), by_site[s].begin(), by_site[s].end());
    }
  }
  fp.offsets.push_back(fp.jumps.size());
  return fp;
}

}  // namespace epv
