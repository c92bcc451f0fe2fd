/* epievo_mi355x_driver.h -- flat C face of epv::SingleSiteSampler (epievo_amd/csrc/host/epv_sampler.hpp),
 * the C++ mirror of the reference's class (/root/reference/src/libepievo/SingleSiteSampler.hpp:35-81)
 * that the drop-in CLIs drive: the EM loop's E-step over every GPU of a node, RCCL linked directly
 * (libepv_rccl.so).  It exists so that programs without a C++ compiler -- bench.py, the tests --
 * run THE product driver instead of a parallel implementation of it.  -> epievo_amd/libepv_driver.so
 *
 *   epvd_create       every GPU slot in this process (device list; repeats rehearse an N-GPU run
 *                     on fewer GPUs through the loopback transport)            ncclCommInitAll
 *   epvd_create_rank  one slot per process (torchrun-style launchers); rank 0 makes the id with
 *                     epvd_unique_id and the launcher passes it around         ncclCommInitRank
 * Calls return 0 or non-zero; the text is in epvd_last_error (per thread for failed creates).
 * Paths are node-major flat as in epievo_mi355x.h; J/D are [(b-1)*8 + ctx] batch averages.
 */
#ifndef EPIEVO_MI355X_DRIVER_H
#define EPIEVO_MI355X_DRIVER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct epvd_sampler epvd_sampler;

epvd_sampler *epvd_create(uint64_t burn_in, uint64_t batch, int n_devices, const int *devices, uint32_t capacity);
int epvd_unique_id(void *id128);
epvd_sampler *epvd_create_rank(uint64_t burn_in, uint64_t batch, int device, int world, int rank,
                               const void *id128, uint32_t capacity);
void epvd_destroy(epvd_sampler *s);
const char *epvd_last_error(const epvd_sampler *s);

/* cut points of `world` slots of an n-site genome (world + 1 entries written; returns how many
 * slots the genome can feed: fewer than `world` when it is too short) */
int epvd_shard_cuts(uint64_t n_sites, int world, uint64_t burn_in, uint64_t batch, uint64_t *cuts);

/* SingleSiteSampler::reset(model, paths) with the tree of TreeHelper; n_global = 0: `paths` is the
 * whole genome (epvd_create); otherwise the owned columns of this process's slot (epvd_create_rank) */
int epvd_reset(epvd_sampler *s, const double *rates, const double *T, int n_nodes, const uint32_t *parent_ids,
               const uint32_t *subtree_sizes, const double *branches, uint64_t n_sites, const uint8_t *init_state,
               const uint64_t *offsets, const double *jumps, uint64_t n_global);
/* reset after a model change (the EM loop's second and later iterations) */
int epvd_reset_model(epvd_sampler *s, const double *rates, const double *T);
/* SingleSiteSampler::run_mcmc (SingleSiteSampler.cpp:550-598) */
int epvd_run_mcmc(epvd_sampler *s, uint64_t seed, uint64_t em_iteration, double *J, double *D, double *acc_rate);
int epvd_scale_jump_times(epvd_sampler *s, const double *new_branches, int n_nodes);
/* the resident paths (one slot per process: its owned columns): sizes first, then the copy */
int epvd_download_sizes(epvd_sampler *s, uint64_t *n_sites, uint64_t *total_jumps);
int epvd_download(epvd_sampler *s, uint8_t *init_state, uint64_t *offsets, double *jumps);

/* how the genome is laid out (tests, bench lines) */
int epvd_layout(epvd_sampler *s, char *buf, int len, int *n_slots_here, int *n_parts_here, int *uses_rccl,
                uint64_t *halo_columns);
int epvd_set_options(epvd_sampler *s, uint32_t flags);           /* EPV_OPT_* on every context */
int epvd_set_timing(epvd_sampler *s, int every);                 /* epv_set_timing on every context */
int epvd_kernel_time_ms(epvd_sampler *s, double *avg_ms, uint64_t *n_launches);
int epvd_phase_mode(epvd_sampler *s, uint32_t *mode);

#ifdef __cplusplus
}
#endif

#endif
