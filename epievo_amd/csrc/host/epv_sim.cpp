// epv_sim.cpp -- host-side forward (Gillespie) simulation of complete histories along
// a tree.  It produces the synthetic inputs of the hot path (BASELINE.md section 3,
// SURVEY.md section 8d "concrete synthetic inputs"): a root sequence from the stationary
// first-order Markov chain (what EpiEvoModel::sample_state_sequence does,
// src/libepievo/EpiEvoModel.cpp:281-298) and, on every branch, context-dependent
// flips of interior sites at the triplet rates (what epievo_sim's sample_jump loop
// does, src/prog/epievo_sim.cpp:102-152,329-352), emitted directly as per-site local
// paths (what global_jumps_to_paths would convert to).  Sites 0 and n-1 never change,
// as in the reference's TripletSampler (src/libepievo/TripletSampler.cpp:37-70).
//
// This is an input generator, not a bit-level restatement of TripletSampler's
// libstdc++-specific draw sequence (that is "next" row f3 of SURVEY.md section 8);
// it uses its own mt19937_64 stream.
#include "epv_sim.hpp"

#include <cmath>
#include <random>
#include <stdexcept>

namespace epv {

namespace {

struct Rng {
  std::mt19937_64 g;
  explicit Rng(uint64_t seed) : g(seed) {}
  double unif() { return (double)(g() >> 11) * (1.0 / 9007199254740992.0); }
  double expo(double rate) { return -std::log(1.0 - unif()) / rate; }
};

// positions 1..n-2 bucketed by their current 3-bit context; O(1) pick and move
struct ContextBuckets {
  std::vector<std::vector<uint32_t>> bucket;
  std::vector<uint32_t> where;
  std::vector<uint8_t> &seq;
  explicit ContextBuckets(std::vector<uint8_t> &s) : bucket(8), where(s.size(), 0), seq(s) {
    for (size_t i = 1; i + 1 < seq.size(); ++i) insert((uint32_t)i, ctx(i));
  }
  int ctx(size_t i) const { return 4 * seq[i - 1] + 2 * seq[i] + seq[i + 1]; }
  void insert(uint32_t pos, int c) {
    where[pos] = (uint32_t)bucket[c].size();
    bucket[c].push_back(pos);
  }
  void erase(uint32_t pos, int c) {
    std::vector<uint32_t> &b = bucket[c];
    const uint32_t k = where[pos];
    b[k] = b.back();
    where[b[k]] = k;
    b.pop_back();
  }
  void flip(uint32_t pos) {
    const size_t n = seq.size();
    for (uint32_t q = pos - 1; q <= pos + 1; ++q)
      if (q >= 1 && q + 1 < n) erase(q, ctx(q));
    seq[pos] ^= 1;
    for (uint32_t q = pos - 1; q <= pos + 1; ++q)
      if (q >= 1 && q + 1 < n) insert(q, ctx(q));
  }
};

}  // namespace

FlatPaths simulate_histories(const Model &model, int n_nodes, const uint32_t *parent,
                             const double *branches, uint64_t n_sites, uint64_t seed) {
  if (n_sites < 3 || n_sites > 0xfffffffeull) throw std::runtime_error("bad n_sites");
  Rng rng(seed);
  const std::array<double, 4> &T = model.T;
  std::vector<std::vector<uint8_t>> end_seq(n_nodes);

  // root sequence
  std::vector<uint8_t> &root = end_seq[0];
  root.resize(n_sites);
  const double pi1 = (1.0 - T[0]) / (2.0 - T[3] - T[0]);
  root[0] = rng.unif() < pi1;
  for (uint64_t i = 1; i < n_sites; ++i) {
    const double stay = root[i - 1] ? T[3] : T[0];
    root[i] = (rng.unif() <= stay) ? root[i - 1] : (uint8_t)(root[i - 1] ^ 1);
  }

  FlatPaths out;
  out.n_sites = n_sites;
  out.n_nodes = n_nodes;
  const uint64_t B = (uint64_t)(n_nodes - 1);
  out.init.resize(B * n_sites);
  out.offsets.assign(B * n_sites + 1, 0);

  std::vector<std::vector<double>> site_jumps(n_sites);
  for (int node = 1; node < n_nodes; ++node) {
    const std::vector<uint8_t> &start = end_seq[parent[node]];
    std::vector<uint8_t> seq(start);
    for (uint64_t s = 0; s < n_sites; ++s) out.init[(uint64_t)(node - 1) * n_sites + s] = start[s];
    for (auto &v : site_jumps) v.clear();

    ContextBuckets cb(seq);
    double t = 0.0;
    for (;;) {
      double total = 0.0;
      for (int c = 0; c < 8; ++c) total += (double)cb.bucket[c].size() * model.rates[c];
      if (!(total > 0.0)) break;
      t += rng.expo(total);
      if (!(t < branches[node])) break;
      double x = rng.unif() * total;
      int c = 0;
      for (; c < 7; ++c) {
        const double w = (double)cb.bucket[c].size() * model.rates[c];
        if (x < w) break;
        x -= w;
      }
      while (cb.bucket[c].empty()) c = (c + 7) % 8;  // guard against rounding at the edge
      const uint64_t k = (uint64_t)(rng.unif() * (double)cb.bucket[c].size());
      const uint32_t pos = cb.bucket[c][k < cb.bucket[c].size() ? k : cb.bucket[c].size() - 1];
      cb.flip(pos);
      site_jumps[pos].push_back(t);
    }
    for (uint64_t s = 0; s < n_sites; ++s) {
      const uint64_t idx = (uint64_t)(node - 1) * n_sites + s;
      out.offsets[idx] = out.jumps.size();
      out.jumps.insert(out.jumps.end(), site_jumps[s].begin(), site_jumps[s].end());
    }
    end_seq[node].swap(seq);
  }
  out.offsets[B * n_sites] = out.jumps.size();
  return out;
}

}  // namespace epv
