// epv_driver_abi.cpp -- include/epievo_mi355x_driver.h: a flat C face over epv::SingleSiteSampler so
// that bench.py and the tests drive the product's C++ EM driver (epv_sampler.cpp + libepv_rccl.so).
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <memory>
#include <string>

#include "epievo_mi355x_comm.h"
#include "epievo_mi355x_driver.h"
#include "epv_sampler.hpp"

#define EPVD_API extern "C" __attribute__((visibility("default")))

struct epvd_sampler {
  std::unique_ptr<epv::SingleSiteSampler> s;
  bool rank_mode = false;
  std::string err;
  epv::FlatPaths staged;   // between epvd_download_sizes and epvd_download
};

namespace {
thread_local std::string g_err;

epv::Model make_model(const double *rates, const double *T) {
  epv::Model m;
  for (int i = 0; i < 8; ++i) m.rates[i] = rates[i];
  for (int i = 0; i < 4; ++i) m.T[i] = T[i];
  return m;
}

template <class F>
int guarded(epvd_sampler *h, F &&f) {
  if (!h) return 1;
  try { f(); return 0; }
  catch (const std::exception &e) { h->err = e.what(); return 1; }
}
}  // namespace

EPVD_API epvd_sampler *epvd_create(uint64_t burn_in, uint64_t batch, int n_devices, const int *devices, uint32_t capacity) {
  try {
    std::unique_ptr<epvd_sampler> h(new epvd_sampler());
    std::vector<int> devs(devices, devices + (n_devices > 0 ? n_devices : 0));
    h->s.reset(new epv::SingleSiteSampler(burn_in, batch, devs, capacity));
    return h.release();
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

EPVD_API int epvd_unique_id(void *id128) { return epv_comm_get_unique_id(id128); }

EPVD_API epvd_sampler *epvd_create_rank(uint64_t burn_in, uint64_t batch, int device, int world, int rank,
                                        const void *id128, uint32_t capacity) {
  try {
    if (!id128) throw std::runtime_error("null communicator id");
    std::unique_ptr<epvd_sampler> h(new epvd_sampler());
    epv::RankSpec r;
    r.device = device; r.world = world; r.rank = rank;
    std::memcpy(r.id, id128, sizeof r.id);
    h->s.reset(new epv::SingleSiteSampler(burn_in, batch, r, capacity));
    h->rank_mode = true;
    return h.release();
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

EPVD_API void epvd_destroy(epvd_sampler *h) { delete h; }
EPVD_API const char *epvd_last_error(const epvd_sampler *h) { return h ? h->err.c_str() : g_err.c_str(); }

EPVD_API int epvd_shard_cuts(uint64_t n_sites, int world, uint64_t burn_in, uint64_t batch, uint64_t *cuts) {
  if (!cuts || world < 1) return 0;
  const std::vector<uint64_t> c = epv::SingleSiteSampler::shard_cuts(n_sites, (size_t)world, burn_in, batch);
  for (size_t i = 0; i < c.size(); ++i) cuts[i] = c[i];
  return (int)c.size() - 1;
}

EPVD_API int epvd_reset(epvd_sampler *h, const double *rates, const double *T, int n_nodes, const uint32_t *parent_ids,
                        const uint32_t *subtree_sizes, const double *branches, uint64_t n_sites, const uint8_t *init_state,
                        const uint64_t *offsets, const double *jumps, uint64_t n_global) {
  return guarded(h, [&] {
    epv::Tree th;
    th.subtree_sizes.assign(subtree_sizes, subtree_sizes + n_nodes);
    th.parent_ids.assign(parent_ids, parent_ids + n_nodes);
    th.branches.assign(branches, branches + n_nodes);
    th.node_names.resize(n_nodes);
    th.name_generated.assign(n_nodes, true);
    epv::FlatPaths fp;
    fp.n_sites = n_sites;
    fp.n_nodes = n_nodes;
    const uint64_t E = (uint64_t)(n_nodes - 1) * n_sites;
    fp.init.assign(init_state, init_state + E);
    fp.offsets.assign(offsets, offsets + E + 1);
    fp.jumps.assign(jumps, jumps + offsets[E]);
    if (h->rank_mode) h->s->reset(make_model(rates, T), th, fp, n_global ? n_global : n_sites);
    else h->s->reset(make_model(rates, T), th, fp);
  });
}

EPVD_API int epvd_reset_model(epvd_sampler *h, const double *rates, const double *T) {
  return guarded(h, [&] { h->s->reset(make_model(rates, T)); });
}

EPVD_API int epvd_run_mcmc(epvd_sampler *h, uint64_t seed, uint64_t em_iteration, double *J, double *D, double *acc_rate) {
  return guarded(h, [&] {
    std::vector<std::vector<double>> Jv, Dv;
    double acc = 0.0;
    h->s->run_mcmc(seed, em_iteration, Jv, Dv, acc);
    for (size_t b = 1; b < Jv.size(); ++b)
      for (int k = 0; k < 8; ++k) { J[(b - 1) * 8 + k] = Jv[b][k]; D[(b - 1) * 8 + k] = Dv[b][k]; }
    if (acc_rate) *acc_rate = acc;
  });
}

EPVD_API int epvd_scale_jump_times(epvd_sampler *h, const double *new_branches, int n_nodes) {
  return guarded(h, [&] { h->s->scale_jump_times(std::vector<double>(new_branches, new_branches + n_nodes)); });
}

EPVD_API int epvd_download_sizes(epvd_sampler *h, uint64_t *n_sites, uint64_t *total_jumps) {
  return guarded(h, [&] {
    h->s->download(h->staged);
    *n_sites = h->staged.n_sites;
    *total_jumps = h->staged.jumps.size();
  });
}

EPVD_API int epvd_download(epvd_sampler *h, uint8_t *init_state, uint64_t *offsets, double *jumps) {
  return guarded(h, [&] {
    const epv::FlatPaths &p = h->staged;
    if (p.offsets.empty()) throw std::runtime_error("epvd_download_sizes first");
    std::copy(p.init.begin(), p.init.end(), init_state);
    std::copy(p.offsets.begin(), p.offsets.end(), offsets);
    std::copy(p.jumps.begin(), p.jumps.end(), jumps);
    h->staged = epv::FlatPaths();
  });
}

EPVD_API int epvd_layout(epvd_sampler *h, char *buf, int len, int *n_slots_here, int *n_parts_here, int *uses_rccl,
                         uint64_t *halo_columns) {
  return guarded(h, [&] {
    if (buf && len > 0) { std::strncpy(buf, h->s->layout().c_str(), (size_t)len - 1); buf[len - 1] = '\0'; }
    if (n_slots_here) *n_slots_here = (int)h->s->n_slots();
    if (n_parts_here) *n_parts_here = (int)h->s->n_parts();
    if (uses_rccl) *uses_rccl = h->s->uses_rccl() ? 1 : 0;
    if (halo_columns) *halo_columns = h->s->halo_columns();
  });
}

EPVD_API int epvd_set_options(epvd_sampler *h, uint32_t flags) { return guarded(h, [&] { h->s->set_options(flags); }); }
EPVD_API int epvd_set_timing(epvd_sampler *h, int every) { return guarded(h, [&] { h->s->set_timing(every); }); }
EPVD_API int epvd_kernel_time_ms(epvd_sampler *h, double *avg_ms, uint64_t *n_launches) {
  return guarded(h, [&] { h->s->kernel_time_ms(*avg_ms, *n_launches); });
}
EPVD_API int epvd_phase_mode(epvd_sampler *h, uint32_t *mode) { return guarded(h, [&] { *mode = h->s->phase_mode(); }); }
