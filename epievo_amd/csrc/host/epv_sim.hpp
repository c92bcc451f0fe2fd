// epv_sim.hpp -- forward simulation of complete histories (synthetic inputs); see epv_sim.cpp
#ifndef EPV_SIM_HPP
#define EPV_SIM_HPP

#include <cstdint>
#include <vector>

#include "epv_model.hpp"

namespace epv {

// Node-major flat local paths, the layout that crosses the C ABI
// (include/epievo_mi355x.h, epv_upload_paths): entry (b-1)*n_sites + site for
// node b = 1..n_nodes-1; jumps of that entry are jumps[offsets[e] .. offsets[e+1]).
struct FlatPaths {
  uint64_t n_sites = 0;
  int n_nodes = 0;
  std::vector<uint8_t> init;
  std::vector<uint64_t> offsets;
  std::vector<double> jumps;
};

FlatPaths simulate_histories(const Model &model, int n_nodes, const uint32_t *parent,
                             const double *branches, uint64_t n_sites, uint64_t seed);

}  // namespace epv

#endif
