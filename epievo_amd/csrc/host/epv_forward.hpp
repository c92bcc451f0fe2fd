// epv_forward.hpp -- forward (Gillespie) simulation of epigenome evolution along a tree,
// the host-side restatement of what epievo_sim does:
//   ContextIndex     = TripletSampler (src/libepievo/TripletSampler.{hpp,cpp}): interior
//                      positions bucketed by their 3-bit context, O(1) random pick and O(1)
//                      re-bucketing of the three positions a flip touches
//   sample_root      = EpiEvoModel::sample_state_sequence (EpiEvoModel.cpp:281-298)
//   simulate_branch  = the sample_jump loop of src/prog/epievo_sim.cpp:102-152,329-352
// It draws from std::mt19937 through the same libstdc++ distributions in the same order
// as the reference, so for a given seed the global jumps are bit-identical
// (tests/test_forward_sim.py pins this against the linked TripletSampler).
// The event chain of one branch is strictly sequential, so this stays on the host.
#ifndef EPV_FORWARD_HPP
#define EPV_FORWARD_HPP

#include <cstdint>
#include <random>
#include <string>
#include <vector>

#include "epv_io.hpp"
#include "epv_model.hpp"

namespace epv {

struct GlobalJump {   // src/libepievo/GlobalJump.hpp:33-44
  double timepoint;
  size_t position;
};

class ContextIndex {
public:
  explicit ContextIndex(const std::vector<uint8_t> &seq);
  size_t count(size_t context) const { return start_[context + 1] - start_[context]; }
  // flip a uniformly chosen position of the given context; returns the position
  size_t random_mutate(size_t context, std::mt19937 &gen);
  void mutate(size_t pos, size_t context);
  void sequence(std::vector<uint8_t> &seq) const;

private:
  size_t context_of(size_t pos) const;
  void move(size_t pos, size_t from, size_t to);
  std::vector<size_t> order_;   // positions grouped by context (TripletSampler::pos_by_pat)
  std::vector<size_t> slot_;    // where each position sits in order_ (idx_in_pat)
  std::vector<size_t> start_;   // 9 block boundaries (cum_pat_count)
  uint8_t first_, last_;
};

void sample_root(const Model &m, size_t n_sites, std::mt19937 &gen, std::vector<uint8_t> &seq);

// one branch: mutates `index` in place, appends the jumps (time order) to `path`
void simulate_branch(const Model &m, double branch_len, std::mt19937 &gen, ContextIndex &index,
                     std::vector<GlobalJump> &path, std::vector<size_t> &events);

// whole tree, as main() of epievo_sim.cpp:288-352: sequences[node], paths[node]
void simulate_tree(const Model &m, const Tree &th, const std::vector<uint8_t> &root_seq,
                   std::mt19937 &gen, std::vector<std::vector<uint8_t>> &sequences,
                   std::vector<std::vector<GlobalJump>> &paths, std::vector<size_t> &events);

// the same process with sibling subtrees on their own threads and one generator per branch,
// seeded from (seed, node): independent of the number of threads, NOT the sequential stream
void simulate_tree_parallel(const Model &m, const Tree &th, const std::vector<uint8_t> &root_seq, uint64_t seed,
                            int threads, std::vector<std::vector<uint8_t>> &sequences,
                            std::vector<std::vector<GlobalJump>> &paths, std::vector<size_t> &events);

// global_jumps file (GlobalJump.cpp:71-140) and the states file writer (epievo_sim.cpp:66-96)
void write_global_jumps(const std::string &file, const std::vector<std::string> &node_names,
                        const std::vector<uint8_t> &root, const std::vector<std::vector<GlobalJump>> &paths);
void read_global_jumps(const std::string &file, std::vector<uint8_t> &root, std::vector<std::string> &node_names,
                       std::vector<std::vector<GlobalJump>> &paths);
void write_states(const std::string &file, bool only_leaves, const Tree &th,
                  const std::vector<std::vector<uint8_t>> &sequences);

// global_jumps_to_paths (src/prog/global_jumps_to_paths.cpp:46-53,154-168)
FlatPaths global_to_local(const Tree &th, const std::vector<std::vector<uint8_t>> &states,
                          const std::vector<std::vector<GlobalJump>> &paths);

}  // namespace epv

#endif
