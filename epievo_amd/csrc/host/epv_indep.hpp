// epv_indep.hpp -- host side of the site-independent 2-rate model that
// epievo_initialization fits before the context-dependent MCEM
// (/root/reference/src/libepievo/IndepSite.hpp:40-72, src/prog/epievo_initialization.cpp).
// The O(n) parts run on the GPU (epv_indep_* of include/epievo_mi355x.h); what stays here
// is O(nodes): the M-steps, the heuristic initial paths and the states-file reader.
#ifndef EPV_INDEP_HPP
#define EPV_INDEP_HPP

#include <cstdint>
#include <string>
#include <vector>

#include "epv_io.hpp"
#include "epv_model.hpp"

namespace epv {

// J/D: per-branch statistics of the 2-state model, [(b-1)*2 + state]

// estimate_rates_indep (IndepSite.cpp:299-318)
void estimate_rates_indep(int n_nodes, const double *J, const double *D, double rates[2]);

// estimate_rates_and_branches_indep (IndepSite.cpp:328-360) without its last loop: the
// caller rescales the resident paths with scale_jump_times(branches) afterwards
void estimate_rates_and_branches_indep(int n_nodes, const double *J, const double *D, double rates[2],
                                       std::vector<double> &branches);

// initialize_model_from_indep_rates (epievo_initialization.cpp:235-247)
Model model_from_indep_rates(const double rates[2]);

// read_states_file of epievo_initialization.cpp:56-134: columns are matched to the tree's
// node names; leaves must be present, missing internal nodes are filled with 0.
// Returns states[node][site].
std::vector<std::vector<uint8_t>> read_states_for_tree(const std::string &states_file, const Tree &th);

// initialize_paths (epievo_initialization.cpp:141-185): internal states are drawn from the
// children's states (the root keeps the states file's column / zeros), and a branch whose
// ends differ gets one uniformly placed jump.  Uses std::mt19937 exactly as the reference.
FlatPaths initialize_paths_heuristic(uint64_t seed, const Tree &th,
                                     std::vector<std::vector<uint8_t>> &states);

}  // namespace epv

#endif
