// epv_sampler.cpp -- see epv_sampler.hpp
#include "epv_sampler.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <thread>

#include "epievo_mi355x.h"

namespace epv {

namespace {
const uint64_t kHalo = 512, kBlock = 256;   // internal halo columns; reduction block (sites)

// sites [lo, hi) of node-major flat paths
FlatPaths slice_sites(const FlatPaths &p, uint64_t lo, uint64_t hi) {
  FlatPaths q;
  q.n_sites = hi - lo;
  q.n_nodes = p.n_nodes;
  const uint64_t B = (uint64_t)p.n_nodes - 1, n = p.n_sites, m = hi - lo;
  q.init.resize(B * m);
  q.offsets.assign(B * m + 1, 0);
  for (uint64_t b = 0; b < B; ++b) {
    std::copy(p.init.begin() + b * n + lo, p.init.begin() + b * n + hi, q.init.begin() + b * m);
    const uint64_t j0 = p.offsets[b * n + lo], j1 = p.offsets[b * n + hi];
    for (uint64_t s = 0; s < m; ++s)
      q.offsets[b * m + s] = q.jumps.size() + (p.offsets[b * n + lo + s] - j0);
    q.jumps.insert(q.jumps.end(), p.jumps.begin() + j0, p.jumps.begin() + j1);
  }
  q.offsets[B * m] = q.jumps.size();
  return q;
}

FlatPaths concat_sites(const std::vector<FlatPaths> &parts) {
  FlatPaths q;
  q.n_nodes = parts[0].n_nodes;
  for (const FlatPaths &p : parts) q.n_sites += p.n_sites;
  const uint64_t B = (uint64_t)q.n_nodes - 1, n = q.n_sites;
  q.init.resize(B * n);
  q.offsets.assign(B * n + 1, 0);
  for (uint64_t b = 0; b < B; ++b) {
    uint64_t at = 0;
    for (const FlatPaths &p : parts) {
      const uint64_t m = p.n_sites;
      std::copy(p.init.begin() + b * m, p.init.begin() + (b + 1) * m, q.init.begin() + b * n + at);
      const uint64_t j0 = p.offsets[b * m], j1 = p.offsets[(b + 1) * m];
      for (uint64_t s = 0; s < m; ++s) q.offsets[b * n + at + s] = q.jumps.size() + (p.offsets[b * m + s] - j0);
      q.jumps.insert(q.jumps.end(), p.jumps.begin() + j0, p.jumps.begin() + j1);
      at += m;
    }
  }
  q.offsets[B * n] = q.jumps.size();
  return q;
}
}  // namespace

SingleSiteSampler::SingleSiteSampler(size_t n_burn_in, size_t n_batch, int device, uint32_t capacity)
    : SAMPLE_ROOT(false), burn_in(n_burn_in), batch(n_batch), ctx_(epv_create(device)), device_(device),
      capacity_(capacity) {
  if (!ctx_)
    throw std::runtime_error("cannot open HIP device " + std::to_string(device) +
                             " (this build has no CPU fallback)");
  if (const char *e = std::getenv("EPV_CONTEXTS_PER_GPU")) contexts_wanted_ = std::max(1, std::atoi(e));
  group_.push_back(ctx_);
}

SingleSiteSampler::~SingleSiteSampler() {
  drop_group();
  epv_destroy(ctx_);
}

void SingleSiteSampler::drop_group() {
  if (d_blocks_) { epv_dev_free(ctx_, d_blocks_); d_blocks_ = nullptr; blocks_batch_ = 0; }
  for (size_t j = 1; j < group_.size(); ++j) epv_destroy(group_[j]);
  group_.assign(1, ctx_);
  lo_.clear(); a_.clear(); b_.clear(); hi_.clear();
}

void SingleSiteSampler::check_on(epv_ctx *c, int rc, const char *what) {
  if (rc != EPV_OK) throw std::runtime_error(std::string(what) + ": " + epv_last_error(c));
}

// before every reset of a group: equal jump-slot widths (an overflow may have widened one
// context), each context's edge columns into its neighbour's halo, halos marked fresh
void SingleSiteSampler::refresh_group() {
  const size_t k = group_.size();
  uint32_t cap = 0;
  for (epv_ctx *c : group_) { uint32_t v = 0; check_on(c, epv_get_capacity(c, &v), "epv_get_capacity"); cap = std::max(cap, v); }
  for (epv_ctx *c : group_) check_on(c, epv_set_capacity(c, cap), "epv_set_capacity");
  for (size_t j = 0; j + 1 < k; ++j) {
    epv_ctx *L = group_[j], *R = group_[j + 1];
    check_on(R, epv_copy_columns(L, b_[j] - kHalo - lo_[j], kHalo, R, 0), "epv_copy_columns");
    check_on(L, epv_copy_columns(R, kHalo, kHalo, L, b_[j] - lo_[j]), "epv_copy_columns");
  }
  for (size_t j = 0; j < k; ++j)
    check_on(group_[j], epv_set_halo(group_[j], j == 0 ? 0 : kHalo, j + 1 == k ? 0 : kHalo), "epv_set_halo");
}

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
  drop_group();
  const uint64_t n = paths.n_sites;
  size_t k = (size_t)contexts_wanted_;
  while (k > 1 && n < k * (2 * kHalo + 2 * kBlock)) --k;   // every context must own more than its halos
  if (k == 1) {
    check(epv_set_tree(ctx_, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(),
                       th.branches.data()), "epv_set_tree");
    check(epv_upload_paths(ctx_, paths.n_sites, paths.init.data(), paths.offsets.data(),
                           paths.jumps.data(), capacity_, 0), "epv_upload_paths");
    reset(m);
    return;
  }
  uint32_t cap = capacity_;
  if (cap == 0) {   // the library's default rule, evaluated once for the whole genome
    uint64_t maxj = 0;
    for (size_t e = 0; e + 1 < paths.offsets.size(); ++e) maxj = std::max(maxj, paths.offsets[e + 1] - paths.offsets[e]);
    cap = (uint32_t)std::min<uint64_t>(127, std::max<uint64_t>(16, 2 * maxj + 8));
  }
  for (size_t j = 1; j < k; ++j) {
    epv_ctx *c = epv_create(device_);
    if (!c) throw std::runtime_error("cannot open a second context on HIP device " + std::to_string(device_));
    group_.push_back(c);
  }
  for (size_t j = 0; j <= k; ++j) {   // cut points on whole reduction blocks
    const uint64_t cut = j == 0 ? 0 : j == k ? n : (uint64_t)((double)j * (double)n / (double)k / kBlock + 0.5) * kBlock;
    if (j < k) a_.push_back(cut);
    if (j > 0) b_.push_back(cut);
  }
  for (size_t j = 0; j < k; ++j) {
    lo_.push_back(a_[j] - (j > 0 ? kHalo : 0));
    hi_.push_back(b_[j] + (j + 1 < k ? kHalo : 0));
    epv_ctx *c = group_[j];
    check_on(c, epv_set_tree(c, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(), th.branches.data()),
             "epv_set_tree");
    const FlatPaths part = slice_sites(paths, lo_[j], hi_[j]);
    const double dummy = 0.0;
    check_on(c, epv_upload_paths(c, part.n_sites, part.init.data(), part.offsets.data(),
                                 part.jumps.empty() ? &dummy : part.jumps.data(), cap, lo_[j]), "epv_upload_paths");
    check_on(c, epv_set_global_length(c, n), "epv_set_global_length");
  }
  reset(m);
}

void SingleSiteSampler::init_paths_indep(const Model &m, const Tree &th, const std::vector<uint8_t> &root_seq,
                                         const std::vector<uint8_t> &leaf_seq, uint64_t seed) {
  if (root_seq.size() != leaf_seq.size()) throw std::runtime_error("sequences differ in length");
  drop_group();   // epievo_sim_pairwise's path runs on one context
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
  for (epv_ctx *c : group_) check_on(c, epv_set_model(c, m.rates.data(), m.T.data()), "epv_set_model");
  if (grouped()) refresh_group();
  for (epv_ctx *c : group_) check_on(c, epv_reset(c), "epv_reset");
}

void SingleSiteSampler::run_mcmc(uint64_t seed, uint64_t em_iteration,
                                 std::vector<std::vector<double>> &J,
                                 std::vector<std::vector<double>> &D, double &acceptance_rate) {
  const size_t B = (size_t)n_nodes_ - 1;
  std::vector<double> Jf(B * 8), Df(B * 8);
  uint64_t n_acc = 0;
  const uint32_t base = (uint32_t)(em_iteration * (burn_in + batch));
  if (!grouped()) {
    check_mcmc(epv_run_mcmc(ctx_, burn_in, batch, seed, base, Jf.data(), Df.data(), &n_acc), "epv_run_mcmc");
  } else {
    const uint64_t nb_total = (n_sites_ + kBlock - 1) / kBlock, V = B * 16;
    if (!d_blocks_ || blocks_batch_ < batch) {
      if (d_blocks_) check(epv_dev_free(ctx_, d_blocks_), "epv_dev_free");
      d_blocks_ = nullptr;
      check(epv_dev_alloc(ctx_, batch * nb_total * V * sizeof(double), &d_blocks_), "epv_dev_alloc");
      blocks_batch_ = batch;
    }
    // one host thread per context: their colour phases run concurrently on their own streams
    const size_t k = group_.size();
    std::vector<int> rcs(k, EPV_OK);
    std::vector<uint64_t> acc(k, 0);
    std::vector<std::thread> th;
    for (size_t j = 0; j < k; ++j)
      th.emplace_back([&, j] {
        rcs[j] = epv_run_mcmc_blocks(group_[j], burn_in, batch, seed, base, static_cast<double *>(d_blocks_),
                                     nb_total, lo_[j] / kBlock, &acc[j]);
      });
    for (std::thread &t : th) t.join();
    for (size_t j = 0; j < k; ++j) {
      if (rcs[j] == EPV_ERR_CAPACITY) {   // absorbed as in check_mcmc; refresh_group() evens the widths out
        uint32_t cap = 0;
        if (epv_get_capacity(group_[j], &cap) == EPV_OK && cap < 127u) {
          capacity_events.push_back(std::string("epv_run_mcmc_blocks: ") + epv_last_error(group_[j]));
          check_on(group_[j], epv_set_capacity(group_[j], cap * 2u > 127u ? 127u : cap * 2u), "epv_set_capacity");
          rcs[j] = EPV_OK;
        }
      }
      check_on(group_[j], rcs[j], "epv_run_mcmc_blocks");
      n_acc += acc[j];
    }
    check(epv_reduce_blocks(ctx_, static_cast<const double *>(d_blocks_), nb_total, batch, 1, Jf.data(), Df.data()),
          "epv_reduce_blocks");
  }
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
  if (!grouped()) {
    check_mcmc(epv_sweep(ctx_, n, seed, sweep_base, &n_acc), "epv_sweep");
    return n_acc;
  }
  // a group can run as many sweeps as its internal halos last, then they are refreshed
  size_t done = 0;
  while (done < n) {
    uint64_t left = ~0ull;
    for (epv_ctx *c : group_) { uint64_t v = 0; check_on(c, epv_halo_phases_left(c, &v), "epv_halo_phases_left"); left = std::min(left, v); }
    const size_t kk = std::min<size_t>(n - done, (size_t)(left / 3));
    if (kk == 0) {
      refresh_group();
      for (epv_ctx *c : group_) check_on(c, epv_reset(c), "epv_reset");
      continue;
    }
    const size_t k = group_.size();
    std::vector<int> rcs(k, EPV_OK);
    std::vector<uint64_t> acc(k, 0);
    std::vector<std::thread> th;
    for (size_t j = 0; j < k; ++j)
      th.emplace_back([&, j] { rcs[j] = epv_sweep(group_[j], kk, seed, sweep_base + (uint32_t)done, &acc[j]); });
    for (std::thread &t : th) t.join();
    for (size_t j = 0; j < k; ++j) { check_on(group_[j], rcs[j], "epv_sweep"); n_acc += acc[j]; }
    done += kk;
  }
  return n_acc;
}

void SingleSiteSampler::scale_jump_times(const std::vector<double> &new_branches) {
  for (epv_ctx *c : group_) check_on(c, epv_scale_jump_times(c, new_branches.data()), "epv_scale_jump_times");
}

void SingleSiteSampler::upload(const Tree &th, const FlatPaths &paths) {
  drop_group();   // the site-independent stage runs on one context
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
  if (grouped()) throw std::runtime_error("get_sufficient_statistics: load the paths with upload() (one context)");
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
  if (grouped()) {
    std::vector<FlatPaths> owned;
    for (size_t j = 0; j < group_.size(); ++j) {
      epv_ctx *c = group_[j];
      uint64_t total = 0;
      check_on(c, epv_paths_total_jumps(c, &total), "epv_paths_total_jumps");
      FlatPaths p;
      p.n_sites = hi_[j] - lo_[j];
      p.n_nodes = n_nodes_;
      const uint64_t E = (uint64_t)(n_nodes_ - 1) * p.n_sites;
      p.init.assign(E, 0);
      p.offsets.assign(E + 1, 0);
      p.jumps.assign(total ? total : 1, 0.0);
      check_on(c, epv_download_paths(c, p.init.data(), p.offsets.data(), p.jumps.data()), "epv_download_paths");
      p.jumps.resize(total);
      owned.push_back(slice_sites(p, a_[j] - lo_[j], b_[j] - lo_[j]));
    }
    paths = concat_sites(owned);
    return;
  }
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
