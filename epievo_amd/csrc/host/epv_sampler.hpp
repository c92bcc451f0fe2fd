// epv_sampler.hpp -- host-side C++ face of the GPU sampler, shaped like the reference's
// SingleSiteSampler (/root/reference/src/libepievo/SingleSiteSampler.hpp:35-81) so that
// the EM driver reads like the reference's (epievo_est_params_histories.cpp:236-264).
// It is a thin wrapper over the C ABI of include/epievo_mi355x.h; errors become
// std::runtime_error (the reference's mains catch std::exception and return
// EXIT_FAILURE, epievo_est_params_histories.cpp:296-299).
//
// Differences forced by the device boundary:
//  * paths live on the GPU between calls: reset(model, tree, paths) uploads them once,
//    reset(model) re-derives the cached log-likelihoods after a model change, and
//    download(paths) brings them back when the driver wants to write them;
//  * the reference threads one std::mt19937 through every call; the parallel schedule
//    uses a counter-based stream, so run_mcmc takes (seed, em_iteration) instead;
//  * the EM driver's path (reset / run_mcmc / scale_jump_times / download) runs TWO contexts
//    per GPU when the genome is long enough (EPV_CONTEXTS_PER_GPU, default 2): each owns a
//    range of whole 256-site blocks plus 512 redundant halo columns, their dependent kernels
//    overlap on two streams, and the statistics come from shared block partials, so every
//    result is bit-identical to one context.
#ifndef EPV_SAMPLER_HPP
#define EPV_SAMPLER_HPP

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "epv_io.hpp"
#include "epv_model.hpp"

struct epv_ctx;

namespace epv {

class SingleSiteSampler {
public:
  SingleSiteSampler(size_t n_burn_in, size_t n_batch, int device = 0, uint32_t capacity = 0);
  ~SingleSiteSampler();
  SingleSiteSampler(const SingleSiteSampler &) = delete;
  SingleSiteSampler &operator=(const SingleSiteSampler &) = delete;

  // SingleSiteSampler::reset (SingleSiteSampler.cpp:449-475)
  void reset(const Model &the_model, const Tree &th, const FlatPaths &paths);
  void reset(const Model &the_model);

  // initialize_paths_indep (src/prog/epievo_sim_pairwise.cpp:62-110) on the device for the
  // two-node tree `th`; afterwards the paths are resident as after reset(model, th, paths)
  void init_paths_indep(const Model &the_model, const Tree &th, const std::vector<uint8_t> &root_seq,
                        const std::vector<uint8_t> &leaf_seq, uint64_t seed);

  // SingleSiteSampler::run_mcmc (:550-598).  J/D are resized to n_nodes rows of 8 (row 0
  // empty) and hold batch averages, exactly as the reference returns them.
  void run_mcmc(uint64_t seed, uint64_t em_iteration, std::vector<std::vector<double>> &J_all_sites,
                std::vector<std::vector<double>> &D_all_sites, double &acceptance_rate);

  // `n` x single_iteration (:538-548); epievo_sim_pairwise.cpp:267-273 spells this loop
  // out with Metropolis_Hastings_site.  Returns the number of accepted proposals.
  size_t sweeps(size_t n, uint64_t seed, uint32_t sweep_base);

  // scale_jump_times (ParamEstimation.cpp:369-380)
  void scale_jump_times(const std::vector<double> &new_branches);

  void download(FlatPaths &paths);

  // upload paths without touching the model (the site-independent stage has no EpiEvoModel yet)
  void upload(const Tree &th, const FlatPaths &paths);
  // get_sufficient_statistics, per-branch overload (ParamEstimation.cpp:92-114), of the
  // resident paths: rows 1..n_nodes-1 of 8 contexts
  void get_sufficient_statistics(std::vector<std::vector<double>> &J, std::vector<std::vector<double>> &D);
  // the site-independent model of IndepSite.hpp:40-72; J/D flat [(b-1)*2 + state]
  void indep_expectation(const double rates[2], std::vector<double> &J, std::vector<double> &D);
  void indep_sufficient_statistics(std::vector<double> &J, std::vector<double> &D);
  void indep_update_paths(const double rates[2], uint64_t seed, uint32_t sweep);

  // MCMC parameter constants (public fields of the reference class)
  bool SAMPLE_ROOT;  // hard-wired false in the reference (SingleSiteSampler.cpp:441)
  size_t burn_in;
  size_t batch;

  // capacity overflows that were absorbed by widening the device jump slots (verbose output)
  std::vector<std::string> capacity_events;

private:
  void check(int rc, const char *what);
  void check_mcmc(int rc, const char *what);
  void check_on(epv_ctx *c, int rc, const char *what);
  bool grouped() const { return group_.size() > 1; }
  void drop_group();          // back to the single context ctx_
  void refresh_group();       // equal capacities, internal halo columns, fresh halo marks
  epv_ctx *ctx_;
  int device_;
  int contexts_wanted_ = 2;
  std::vector<epv_ctx *> group_;            // group_[0] == ctx_ when several contexts share the GPU
  std::vector<uint64_t> lo_, a_, b_, hi_;   // local range [lo, hi) and owned range [a, b) of each
  void *d_blocks_ = nullptr;                // [batch][blocks][16 (N-1)] level-0 partials of the group
  uint64_t blocks_batch_ = 0;
  uint32_t capacity_;
  int n_nodes_ = 0;
  uint64_t n_sites_ = 0;
};

}  // namespace epv

#endif
