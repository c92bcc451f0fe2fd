// epv_sampler.cpp -- see epv_sampler.hpp
#include "epv_sampler.hpp"

#include <stdexcept>
#include <string>

#include "epievo_mi355x.h"

namespace epv {

SingleSiteSampler::SingleSiteSampler(size_t n_burn_in, size_t n_batch, int device, uint32_t capacity)
    : SAMPLE_ROOT(false), burn_in(n_burn_in), batch(n_batch), ctx_(epv_create(device)),
      capacity_(capacity) {
  if (!ctx_)
    throw std::runtime_error("cannot open HIP device " + std::to_string(device) +
                             " (this build has no CPU fallback)");
}

SingleSiteSampler::~SingleSiteSampler() { epv_destroy(ctx_); }

void SingleSiteSampler::check(int rc, const char *what) {
  if (rc != EPV_OK) throw std::runtime_error(std::string(what) + ": " + epv_last_error(ctx_));
}

// after an MCMC call: an overflow leaves a valid chain and complete outputs (the over-long
// proposals were rejected), so widen the jump slots for the following calls -- what the
// reference's std::vector paths do on their own -- and carry on
void SingleSiteSampler::check_mcmc(int rc, const char *what) {
  if (rc == EPV_ERR_CAPACITY) {
    uint32_t cap = 0;
    if (epv_get_capacity(ctx_, &cap) == EPV_OK && cap < 127u) {
      const std::string msg = epv_last_error(ctx_);
      check(epv_set_capacity(ctx_, cap * 2u > 127u ? 127u : cap * 2u), "epv_set_capacity");
      capacity_events.push_back(std::string(what) + ": " + msg);
      return;
    }
  }
  check(rc, what);
}

void SingleSiteSampler::reset(const Model &m, const Tree &th, const FlatPaths &paths) {
  n_nodes_ = th.n_nodes();
  n_sites_ = paths.n_sites;
  check(epv_set_tree(ctx_, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(),
                     th.branches.data()), "epv_set_tree");
  check(epv_upload_paths(ctx_, paths.n_sites, paths.init.data(), paths.offsets.data(),
                         paths.jumps.data(), capacity_, 0), "epv_upload_paths");
  reset(m);
}

void SingleSiteSampler::init_paths_indep(const Model &m, const Tree &th, const std::vector<uint8_t> &root_seq,
                                         const std::vector<uint8_t> &leaf_seq, uint64_t seed) {
  if (root_seq.size() != leaf_seq.size()) throw std::runtime_error("sequences differ in length");
  n_nodes_ = th.n_nodes();
  n_sites_ = root_seq.size();
  check(epv_set_tree(ctx_, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(),
                     th.branches.data()), "epv_set_tree");
  check(epv_set_model(ctx_, m.rates.data(), m.T.data()), "epv_set_model");
  check(epv_init_paths_indep(ctx_, root_seq.size(), root_seq.data(), leaf_seq.data(), seed, capacity_),
        "epv_init_paths_indep");
  check(epv_reset(ctx_), "epv_reset");
}

void SingleSiteSampler::reset(const Model &m) {
  check(epv_set_model(ctx_, m.rates.data(), m.T.data()), "epv_set_model");
  check(epv_reset(ctx_), "epv_reset");
}

void SingleSiteSampler::run_mcmc(uint64_t seed, uint64_t em_iteration,
                                 std::vector<std::vector<double>> &J,
                                 std::vector<std::vector<double>> &D, double &acceptance_rate) {
  const size_t B = (size_t)n_nodes_ - 1;
  std::vector<double> Jf(B * 8), Df(B * 8);
  uint64_t n_acc = 0;
  const uint32_t base = (uint32_t)(em_iteration * (burn_in + batch));
  check_mcmc(epv_run_mcmc(ctx_, burn_in, batch, seed, base, Jf.data(), Df.data(), &n_acc), "epv_run_mcmc");
  J.assign(n_nodes_, {});
  D.assign(n_nodes_, {});
  for (size_t b = 1; b <= B; ++b) {
    J[b].assign(Jf.begin() + (b - 1) * 8, Jf.begin() + b * 8);
    D[b].assign(Df.begin() + (b - 1) * 8, Df.begin() + b * 8);
  }
  acceptance_rate = static_cast<double>(n_acc) / (batch * (n_sites_ - 2));
}

size_t SingleSiteSampler::sweeps(size_t n, uint64_t seed, uint32_t sweep_base) {
  uint64_t n_acc = 0;
  check_mcmc(epv_sweep(ctx_, n, seed, sweep_base, &n_acc), "epv_sweep");
  return n_acc;
}

void SingleSiteSampler::scale_jump_times(const std::vector<double> &new_branches) {
  check(epv_scale_jump_times(ctx_, new_branches.data()), "epv_scale_jump_times");
}

void SingleSiteSampler::upload(const Tree &th, const FlatPaths &paths) {
  n_nodes_ = th.n_nodes();
  n_sites_ = paths.n_sites;
  check(epv_set_tree(ctx_, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(),
                     th.branches.data()), "epv_set_tree");
  // the kernels of the site-independent stage never read the 8 rates; park neutral ones
  const double ones[8] = {1, 1, 1, 1, 1, 1, 1, 1}, T[4] = {0.5, 0.5, 0.5, 0.5};
  check(epv_set_model(ctx_, ones, T), "epv_set_model");
  const double dummy = 0.0;
  check(epv_upload_paths(ctx_, paths.n_sites, paths.init.data(), paths.offsets.data(),
                         paths.jumps.empty() ? &dummy : paths.jumps.data(), capacity_, 0),
        "epv_upload_paths");
}

void SingleSiteSampler::get_sufficient_statistics(std::vector<std::vector<double>> &J,
                                                  std::vector<std::vector<double>> &D) {
  const size_t B = (size_t)n_nodes_ - 1;
  std::vector<double> Jf(B * 8), Df(B * 8);
  check(epv_get_sufficient_statistics(ctx_, Jf.data(), Df.data()), "epv_get_sufficient_statistics");
  J.assign(n_nodes_, {});
  D.assign(n_nodes_, {});
  for (size_t b = 1; b <= B; ++b) {
    J[b].assign(Jf.begin() + (b - 1) * 8, Jf.begin() + b * 8);
    D[b].assign(Df.begin() + (b - 1) * 8, Df.begin() + b * 8);
  }
}

void SingleSiteSampler::indep_expectation(const double rates[2], std::vector<double> &J,
                                          std::vector<double> &D) {
  J.assign(((size_t)n_nodes_ - 1) * 2, 0.0);
  D.assign(((size_t)n_nodes_ - 1) * 2, 0.0);
  check(epv_indep_expectation(ctx_, rates, J.data(), D.data()), "epv_indep_expectation");
}

void SingleSiteSampler::indep_sufficient_statistics(std::vector<double> &J, std::vector<double> &D) {
  J.assign(((size_t)n_nodes_ - 1) * 2, 0.0);
  D.assign(((size_t)n_nodes_ - 1) * 2, 0.0);
  check(epv_indep_sufficient_statistics(ctx_, J.data(), D.data()), "epv_indep_sufficient_statistics");
}

void SingleSiteSampler::indep_update_paths(const double rates[2], uint64_t seed, uint32_t sweep) {
  check_mcmc(epv_indep_update_paths(ctx_, rates, seed, sweep), "epv_indep_update_paths");
}

void SingleSiteSampler::download(FlatPaths &paths) {
  uint64_t total = 0;
  check(epv_paths_total_jumps(ctx_, &total), "epv_paths_total_jumps");
  paths.n_sites = n_sites_;
  paths.n_nodes = n_nodes_;
  const uint64_t E = (uint64_t)(n_nodes_ - 1) * n_sites_;
  paths.init.assign(E, 0);
  paths.offsets.assign(E + 1, 0);
  paths.jumps.assign(total ? total : 1, 0.0);
  check(epv_download_paths(ctx_, paths.init.data(), paths.offsets.data(), paths.jumps.data()),
        "epv_download_paths");
  paths.jumps.resize(total);
}

}  // namespace epv
