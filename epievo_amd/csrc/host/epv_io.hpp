// epv_io.hpp -- the text formats that form the drop-in surface of the hot path
// (SURVEY.md section 8b "file formats to keep"): Newick tree, local_paths, states.
#ifndef EPV_IO_HPP
#define EPV_IO_HPP

#include <cstdint>
#include <string>
#include <vector>

#include "epv_sim.hpp"  // FlatPaths

namespace epv {

// Tree in the pre-order array form the sampler consumes -- the fields of the
// reference's TreeHelper (src/libepievo/TreeHelper.hpp:47-51).
struct Tree {
  std::vector<uint32_t> subtree_sizes;
  std::vector<uint32_t> parent_ids;
  std::vector<double> branches;
  std::vector<std::string> node_names;
  std::vector<bool> name_generated;   // node_<k> made up for an unnamed node: not printed back
  int n_nodes() const { return (int)subtree_sizes.size(); }
  bool is_leaf(int node) const { return subtree_sizes[node] == 1; }

  // operator>>(istream&, PhyloTree&) + TreeHelper(PhyloTreePreorder)
  // (src/libepievo/PhyloTree.cpp:110-122,144-203,286-301; TreeHelper.cpp:43-51)
  static Tree parse(const std::string &newick);
  static Tree read(const std::string &tree_file);
  // the two-node tree of TreeHelper(const double &evo_time), TreeHelper.cpp:53-60
  static Tree single_branch(double evo_time);
  // PhyloTree::Newick_format (PhyloTree.cpp:110-122) with the current branch lengths
  std::string newick() const;
};

// read_paths(path_file, node_names, paths) (src/libepievo/Path.cpp:123-148), returned
// node-major.  tot_times[b] is the tot_time column of node b (taken from its first
// site; every site of a node must agree, else std::runtime_error).
FlatPaths read_local_paths(const std::string &path_file, std::vector<std::string> &node_names,
                           std::vector<double> &tot_times);

// the writers of src/prog/epievo_est_params_histories.cpp:56-75 (root line, then per
// node "NODE:<name>" and "site\tinit\ttot_time\tjump\t..." at max_digits10)
void write_local_paths(const std::string &path_file, const std::vector<std::string> &node_names,
                       int n_nodes, uint64_t n_sites, const double *tot_times,
                       const uint8_t *init, const uint64_t *offsets, const double *jumps);

// read_states_file (src/libepievo/epievo_utils.cpp:90-125): states[seq][site]
void read_states_file(const std::string &states_file, std::vector<std::string> &names,
                      std::vector<std::vector<uint8_t>> &states);

}  // namespace epv

#endif
