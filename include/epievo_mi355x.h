/* epievo_mi355x.h -- C ABI of the MI355X (gfx950) implementation of epievo's MCEM
 * inner loop: the per-site Metropolis-Hastings end-conditioned path sampler.
 *
 * The reference has no FFI; its boundary for this path is the C++ class
 * SingleSiteSampler (/root/reference/src/libepievo/SingleSiteSampler.hpp:35-81) plus
 * two free functions of ParamEstimation.hpp.  Each entry point below names the
 * reference interface it replaces.  The C++ wrapper epv::SingleSiteSampler
 * (epievo_amd/csrc/host/epv_sampler.hpp) keeps the reference's names and argument
 * meaning on top of this ABI; INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *  - plain pointers and sizes only; all buffers are caller-owned HOST memory unless a
 *    name ends in _dev (then it is a device pointer on the context's GPU);
 *  - every call returns 0 on success or a non-zero EPV_ERR_* code; the message is
 *    available from epv_last_error(ctx).  No exception crosses the boundary;
 *  - one context per GPU, calls on one context are serialised by the caller (the
 *    reference class is single-threaded and not re-entrant either);
 *  - paths are passed "node-major flat": for node b = 1..n_nodes-1 and site s the
 *    entry index is e = (b-1)*n_sites + s; init_state[e] is Path::init_state, the
 *    jumps are jumps[offsets[e] .. offsets[e+1]) (Path::jumps, ascending, absolute
 *    times in (0, branch length)).  Node 0 (the root) has no path, as in the
 *    reference where paths[site][0] is a dummy (epievo_est_params_histories.cpp:186-192);
 *  - J and D are per-branch sufficient statistics laid out [(b-1)*8 + ctx] with
 *    ctx = 4*left + 2*mid + right (epievo_utils.hpp:85-88).
 */
#ifndef EPIEVO_MI355X_H
#define EPIEVO_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct epv_ctx epv_ctx;

enum {
  EPV_OK = 0,
  EPV_ERR_ARG = 1,      /* bad argument / call order */
  EPV_ERR_HIP = 2,      /* a HIP runtime call failed (no device, OOM, ...) */
  EPV_ERR_CAPACITY = 3, /* a proposed path needed more than `capacity` jumps */
  EPV_ERR_STATE = 4     /* paths/model/tree not set */
};

/* counters returned by epv_get_counters */
typedef struct {
  uint64_t n_overflow;   /* proposals rejected because a branch exceeded `capacity` jumps */
  uint64_t n_coop_tasks; /* rejection tasks resolved by the wave-cooperative search */
  uint64_t n_sweeps;     /* colour-complete sweeps executed since create */
  uint64_t reserved;
} epv_counters;

/* Run-time options of a context (epv_set_options; default 0).
 * EPV_OPT_REFERENCE_PROPOSAL_RATIO  evaluate the proposal ratio q(old)/q(new) of
 *     Metropolis_Hastings_site with the reference's sums (downward_sampling_branch,
 *     SingleSiteSampler.cpp:180-225, and proposal_prob_branch, :272-314).  With SAMPLE_ROOT false
 *     (hard-wired, :441) that ratio is EXACTLY 1: per segment the reference accumulates
 *     log(p[k+1][end] / p[k][start]), which telescopes along every branch and over the tree to the
 *     log of the proposal's normalising constant -- independent of the path.  The reference's
 *     value differs from 0 by rounding only; by default the kernels use the exact 0 and skip
 *     the sums (same paths, same statistics; tests/test_proposal_ratio.py).
 * EPV_OPT_FORWARD_REJECTION  sample state-changing segments by forward rejection
 *     (end_cond_sample_forward_rejection, EndCondSampling.cpp:479-509, the sampler the
 *     reference's hot path calls) instead of the reference library's
 *     end_cond_sampling_Nielsen (:576-617), the default here because its acceptance probability
 *     does not vanish on short branches.  Same conditional law; for parity tests.
 * EPV_OPT_SAMPLE_ROOT  the reference class's public field SAMPLE_ROOT (SingleSiteSampler.hpp:78): the
 *     proposal also draws the site's root state from its posterior given the neighbours' root
 *     states and the data below (root_post_prob0 / downward_sampling, SingleSiteSampler.cpp:167-176,
 *     :246-249) and both proposal log-probabilities carry the term (:325-329).  The reference
 *     hard-wires it to false (:441) and none of its programs sets it.  With it the proposal ratio
 *     is no longer identically 1, so the kernels evaluate it as under
 *     EPV_OPT_REFERENCE_PROPOSAL_RATIO (the first-generation proposal kernel; about half the
 *     default throughput).  Pinned: oracle rung A with the field set == the linked reference bit
 *     for bit; the GPU == rung B bit for bit (tests/test_sample_root.py). */
enum { EPV_OPT_REFERENCE_PROPOSAL_RATIO = 1, EPV_OPT_FORWARD_REJECTION = 2, EPV_OPT_SAMPLE_ROOT = 4 };

/* Create a context on HIP device `device_id`.  Returns NULL when the device cannot be
 * initialised (the product has no CPU fallback).  Replaces
 * SingleSiteSampler::SingleSiteSampler (SingleSiteSampler.cpp:439-447). */
epv_ctx *epv_create(int device_id);
void epv_destroy(epv_ctx *ctx);
const char *epv_last_error(const epv_ctx *ctx);

/* Tree in pre-order array form = the fields of TreeHelper (TreeHelper.hpp:47-51):
 * subtree_sizes, parent_ids, branches (branches[0] = 0, the root). */
int epv_set_tree(epv_ctx *ctx, int n_nodes, const uint32_t *parent_ids,
                 const uint32_t *subtree_sizes, const double *branches);

/* Model = the two members of EpiEvoModel the sampler reads (EpiEvoModel.hpp:37-41):
 * triplet_rates[8] and the horizontal transition matrix T (row-major 2x2).
 * log(rate) is taken here on the host with libm, as reset() does
 * (SingleSiteSampler.cpp:464-468). */
int epv_set_model(epv_ctx *ctx, const double *triplet_rates, const double *T);

/* Upload all paths (replaces handing `vector<vector<Path>>&` to reset()).
 * `capacity` is the fixed number of jump slots kept per (site, branch) on the device
 * (1..2047: a branch has at most 2 * capacity + 1 segments and the segment field of the random
 * stream's address has 12 bits); 0 picks max(16, 2*max_jumps_in_input + 8).  A proposal that would need
 * more is rejected and counted (epv_counters.n_overflow) and the MCMC call that saw it
 * returns EPV_ERR_CAPACITY after completing -- re-upload with a larger capacity.
 * `global_site_offset`: index of local site 0 in the whole genome (0 on one GPU);
 * it keys the RNG and the 3-colouring so that sharded runs reproduce unsharded ones. */
int epv_upload_paths(epv_ctx *ctx, uint64_t n_sites, const uint8_t *init_state,
                     const uint64_t *offsets, const double *jumps, uint32_t capacity,
                     uint64_t global_site_offset);

/* Change the number of jump slots per (site, branch) of the RESIDENT paths, on the device
 * (no host round trip): the jump planes are re-strided, states and cached log-likelihoods
 * stay valid.  The reference's std::vector paths grow on demand; this is the equivalent a
 * wrapper calls after EPV_ERR_CAPACITY (the run that reported it is a valid chain on the
 * histories with at most `capacity` jumps per branch).  Fails with EPV_ERR_CAPACITY when a
 * resident path has more jumps than `capacity`; capacity is clamped to 1..2047. */
int epv_set_capacity(epv_ctx *ctx, uint32_t capacity);
int epv_get_capacity(epv_ctx *ctx, uint32_t *capacity);
int epv_set_options(epv_ctx *ctx, uint32_t flags);
int epv_get_options(epv_ctx *ctx, uint32_t *flags);
/* Which kernels a colour phase of the resident paths launches (chosen by tree size, mean jumps per
 * branch and launch size; results are bit-identical in every mode) -- for profiles and bench lines:
 * 0 = epv_mh_propose_kernel + epv_mh_jumps_kernel + epv_mh_accept_kernel (reference
 *     proposal arithmetic), 1 = epv_mh_propose2_kernel + jumps + accept, 2 = epv_mh_propose2_kernel +
 *     epv_seg_search_kernel + epv_seg_assemble_kernel + accept (long branches), 3 = the fused phase:
 *     one epv_mh_propose2_kernel launch that also samples the jump times and accepts (launches of few
 *     waves), 4 = epv_mh_propose3_kernel + epv_mh_jumps_all_kernel + epv_mh_accept3_kernel (large trees:
 *     the tree walked level by level, heavy branches one lane each, a lane per (site, triple) in the
 *     acceptance).  No reference counterpart. */
enum { EPV_PHASE_V1 = 0, EPV_PHASE_V2 = 1, EPV_PHASE_V2_SEGMENTS = 2, EPV_PHASE_FUSED = 3, EPV_PHASE_V3 = 4 };
int epv_phase_mode(epv_ctx *ctx, uint32_t *mode);

/* initialize_paths_indep (src/prog/epievo_sim_pairwise.cpp:62-110) on the device, for the
 * two-node tree of one branch (epv_set_tree with n_nodes = 2 and epv_set_model first):
 * every interior site gets an independent end-conditioned path root[i] -> leaf[i] by
 * forward rejection (EndCondSampling.cpp:512-542) with the context rates read off the
 * ROOT sequence; the two end sites get at most one uniformly placed jump.  Replaces the
 * upload: afterwards the paths are resident as if epv_upload_paths had been called.
 * capacity 0 = 32 jump slots. */
int epv_init_paths_indep(epv_ctx *ctx, uint64_t n_sites, const uint8_t *root_states,
                         const uint8_t *leaf_states, uint64_t seed, uint32_t capacity);

/* epievo_sim's forward simulation (src/prog/epievo_sim.cpp:102-152, 329-352 over TripletSampler,
 * src/libepievo/TripletSampler.cpp:165-184) on the device, for the tree and model set before:
 * the root sequence (root_states, one byte per site; NULL = EpiEvoModel::sample_state_sequence,
 * EpiEvoModel.cpp:281-298, with keyed uniforms), then every branch in pre-order.  The reference
 * runs ONE sequential event chain per branch; here every interior site carries its own candidate
 * stream at the rate max_c rate_c and a candidate flips the site with probability
 * rate[context]/max rate (thinning: the same law), candidates being resolved site-parallel in
 * any order their nearest-neighbour dependencies allow (csrc/epv_forward.h).  The outcome is a
 * function of `seed` alone and equals oracle/epv_oracle.c's orc_forward_thinning bit for bit; it is
 * NOT the std::mt19937 stream of the reference (that one is restated on the host:
 * csrc/host/epv_forward.cpp, bit-identical to the linked TripletSampler).  Afterwards the
 * histories are resident as after epv_upload_paths (epv_download_paths brings them to the host;
 * node states = init ^ parity of the jump count).  capacity 0 = 16 jump slots; EPV_ERR_CAPACITY
 * when a path needs more (call again with a larger capacity: same histories). */
int epv_forward_simulate(epv_ctx *ctx, uint64_t n_sites, const uint8_t *root_states, uint64_t seed,
                         uint32_t capacity, uint64_t *total_jumps);
/* wall clock of the last epv_forward_simulate: device memory management (freeing the previous paths,
 * allocating the new ones: tens of GB at n = 1e7) and the simulation itself (root sequence, every
 * branch, the count of the jumps) */
int epv_forward_last_ms(epv_ctx *ctx, double *alloc_ms, double *simulate_ms);

/* ---- the site-independent 2-rate model of epievo_initialization (IndepSite.hpp:40-72);
 * rates = {r0, r1}; J/D laid out [(b-1)*2 + state].
 * epv_indep_expectation            expectation_sufficient_statistics (IndepSite.cpp:222-238):
 *                                  conditional means summed over ALL sites
 * epv_indep_sufficient_statistics  compute_sufficient_statistics (:266-297): per-branch
 *                                  averages of the resident paths
 * epv_indep_update_paths           update_paths_indep (:241-259): fresh end-conditioned paths
 *                                  for every site; `sweep` keys the random stream */
int epv_indep_expectation(epv_ctx *ctx, const double *rates, double *J, double *D);
int epv_indep_sufficient_statistics(epv_ctx *ctx, double *J, double *D);
int epv_indep_update_paths(epv_ctx *ctx, const double *rates, uint64_t seed, uint32_t sweep);

/* Site-sharded runs only: total genome length (default: global_site_offset + n_sites),
 * so that the two special cases at the genome ends (SingleSiteSampler.cpp:422,427) are
 * decided on global indices. */
int epv_set_global_length(epv_ctx *ctx, uint64_t n_global);

/* Restrict MH updates to local sites [first, last] (inclusive); default [1, n_sites-2].
 * Sites outside are read-only halo/boundary columns. */
int epv_set_update_range(epv_ctx *ctx, uint64_t first, uint64_t last);

/* Site-sharded runs with wide halos ("temporal blocking"): declare the first `left` and
 * the last `right` local columns to be copies of the neighbouring shards' edge columns
 * (0 = this side is the genome end).  Because the RNG and the colouring are keyed by
 * the GLOBAL site index, a shard can update its halo columns redundantly and obtain
 * exactly what the owner computes; every colour phase makes two more columns at each
 * internal edge stale, so a halo of H columns lasts H/2 phases (H/6 sweeps) before
 * the columns must be refreshed (epv_put_columns, then epv_set_halo again).  While
 * this mode is on, sweeps shrink their update range automatically, and J/D and the
 * accept count cover the owned columns only.  Calling it also marks the halos fresh. */
int epv_set_halo(epv_ctx *ctx, uint64_t left, uint64_t right);
/* how many more colour phases the current halos allow (UINT64_MAX when unbounded) */
int epv_halo_phases_left(epv_ctx *ctx, uint64_t *phases);

/* SingleSiteSampler::reset (SingleSiteSampler.cpp:449-475): cache the complete-data
 * log-likelihood of every interior triple. */
int epv_reset(epv_ctx *ctx);
/* the same without waiting for the device: whatever is called next on the context runs behind it */
int epv_reset_async(epv_ctx *ctx);

/* n_sweeps x single_iteration (SingleSiteSampler.cpp:538-548) under the 3-colour
 * schedule; this is also the loop epievo_sim_pairwise.cpp:267-273 spells out by hand.
 * Sweep w uses RNG sweep index sweep_base + w.  n_accepted may be NULL. */
int epv_sweep(epv_ctx *ctx, uint64_t n_sweeps, uint64_t seed, uint32_t sweep_base,
              uint64_t *n_accepted);

/* One colour phase (colour = global_site % 3) of sweep `sweep`; used by multi-GPU
 * drivers that exchange halo columns between phases. */
int epv_sweep_phase(epv_ctx *ctx, int colour, uint64_t seed, uint32_t sweep,
                    uint64_t *n_accepted);

/* SingleSiteSampler::run_mcmc (SingleSiteSampler.cpp:550-598): burn_in sweeps, then
 * batch x {sweep; get_sufficient_statistics; accumulate}.  J/D ((n_nodes-1)*8 doubles
 * each) return the batch AVERAGES as the reference does; n_accepted counts the batch
 * sweeps only (acc_rate = n_accepted / (batch*(n_sites-2))). */
int epv_run_mcmc(epv_ctx *ctx, uint64_t burn_in, uint64_t batch, uint64_t seed,
                 uint32_t sweep_base, double *J, double *D, uint64_t *n_accepted);

/* Same, but with average = 0 J/D return the SUMS over the batch sweeps: a site-sharded
 * driver adds the shards' sums (exact for J) and divides once, which reproduces the
 * unsharded averages bit-for-bit. */
int epv_run_mcmc_sums(epv_ctx *ctx, uint64_t burn_in, uint64_t batch, uint64_t seed,
                      uint32_t sweep_base, int average, double *J, double *D,
                      uint64_t *n_accepted);

/* get_sufficient_statistics, per-branch overload (ParamEstimation.cpp:92-114), over
 * the update range's triples.  The sums are EXACT integers on the device -- J as counts, every
 * dwell time of branch b as rint(dt * 2^k_b) with k_b = min(61 - e(n_global * T_b), 50 - e(T_b)),
 * e(x) the frexp exponent -- and become doubles on the host (D = integer * 2^-k_b): the result
 * does not depend on the launch shape, the number of contexts or GPUs, or any summation order,
 * and lies within 2^-41 T_b per term of the exact sum at n = 1e6 (closer than a sequential
 * fp64 sum). */
int epv_get_sufficient_statistics(epv_ctx *ctx, double *J, double *D);

/* scale_jump_times (ParamEstimation.cpp:369-380): jumps *= new/old per branch. */
int epv_scale_jump_times(epv_ctx *ctx, const double *new_branches);

/* Download the current paths (what the EM driver writes out each iteration,
 * epievo_est_params_histories.cpp:280-283).  Call epv_paths_total_jumps first to size
 * `jumps`; offsets has (n_nodes-1)*n_sites + 1 entries. */
int epv_paths_total_jumps(epv_ctx *ctx, uint64_t *total);
int epv_download_paths(epv_ctx *ctx, uint8_t *init_state, uint64_t *offsets, double *jumps);

/* the cached triple log-likelihoods (private member tri_llh of the reference class);
 * exposed for parity tests.  out has n_sites entries. */
int epv_get_tri_llh(epv_ctx *ctx, double *out);

/* Halo exchange for site-sharded runs: copy `count` whole site columns (all branches)
 * starting at local site `first` to / from a packed host buffer of
 * epv_column_bytes(ctx) bytes per column. */
uint64_t epv_column_bytes(const epv_ctx *ctx);
int epv_get_columns(epv_ctx *ctx, uint64_t first, uint64_t count, void *packed);
int epv_put_columns(epv_ctx *ctx, uint64_t first, uint64_t count, const void *packed);
/* the same with the packed columns in DEVICE memory of the context's GPU (the buffer a driver
 * hands to RCCL: include/epievo_mi355x_comm.h, or a torch tensor's storage); synchronous */
int epv_pack_columns_dev(epv_ctx *ctx, uint64_t first, uint64_t count, void *d_packed);
int epv_unpack_columns_dev(epv_ctx *ctx, uint64_t first, uint64_t count, const void *d_packed);
int epv_device_of(const epv_ctx *ctx);
/* the same between two contexts of ONE GPU (equal tree and capacity), without leaving the device */
int epv_copy_columns(epv_ctx *src, uint64_t src_first, uint64_t count, epv_ctx *dst, uint64_t dst_first);
/* ... without waiting on the host: packed on src's stream into half `slot` (0 / 1) of its staging
 * buffer, unpacked on dst's stream behind an event; what the caller launches on dst's stream next
 * (epv_reset) sees the columns.  One use of a (src, slot) pair per refresh. */
int epv_copy_columns_async(epv_ctx *src, uint64_t src_first, uint64_t count, epv_ctx *dst, uint64_t dst_first,
                           int slot);

/* ---- several shards on one GPU (new; the reference is single-process).  Two or three
 * contexts on one device, each owning a contiguous range of 256-aligned site blocks plus
 * redundant halos, run their colour phases on their own streams so that the ramps and tails
 * of the dependent kernels overlap (+17 % on one MI355X).  Every shard writes the integer
 * partial sums (see epv_get_sufficient_statistics) of its OWNED 256-site blocks, per batch sweep,
 * into one buffer shared by the group (epv_run_mcmc_blocks; d_blocks[w][block][16 (N-1)] 64-bit
 * words -- int64 behind the double pointer, J counts then fixed-point D; block_offset = index of
 * the shard's local block 0 in the group -- negative when a halo precedes the first owned block of
 * the buffer; the shard's global_site_offset must be a multiple of 256), and one reduction
 * (epv_reduce_blocks, on any context of the group) adds them up.  Integer sums are exact, so J AND
 * D equal the one-context run bit for bit however the genome is cut.
 * epv_dev_alloc returns zero-filled device memory (blocks nobody owns must read as 0). */
int epv_dev_alloc(epv_ctx *ctx, uint64_t bytes, void **device_ptr);
int epv_dev_free(epv_ctx *ctx, void *device_ptr);
int epv_dev_write(epv_ctx *ctx, void *d_dst, const void *src, uint64_t bytes);
int epv_dev_read(epv_ctx *ctx, void *dst, const void *d_src, uint64_t bytes);
int epv_run_mcmc_blocks(epv_ctx *ctx, uint64_t burn_in, uint64_t batch, uint64_t seed, uint32_t sweep_base,
                        double *d_blocks, uint64_t n_blocks_total, int64_t block_offset,
                        uint64_t *n_accepted);
int epv_reduce_blocks(epv_ctx *ctx, const double *d_blocks, uint64_t n_blocks_total, uint64_t batch,
                      int average, double *J, double *D);

/* ---- statistics of a genome sharded over several GPUs (new).  Shards are cut on multiples of
 * 256 * row_blocks sites (row_blocks a power of two).  Every GPU reduces the level-0 partials of
 * ITS blocks (d_blocks as written by epv_run_mcmc_blocks, nb_total blocks) to rows of row_blocks
 * blocks, d_rows[row][w][16 (N-1)]; the rows of all GPUs, concatenated in genome order (one RCCL
 * all-gather per EM iteration), go through epv_reduce_rows.  Every stage adds 64-bit integers, so
 * J AND D equal the one-context results bit for bit, whatever the number of GPUs and rows. */
int epv_blocks_to_rows(epv_ctx *ctx, const double *d_blocks, uint64_t n_blocks_total, uint64_t batch,
                       uint32_t row_blocks, double *d_rows);
int epv_reduce_rows(epv_ctx *ctx, const double *d_rows, uint64_t n_rows, uint64_t batch, int average,
                    double *J, double *D);
/* the same straight on the output of an all-gather of equally sized pieces:
 * rank r's piece starts at d_gathered + r * piece_doubles (0 = max_rows * batch * 16 (N-1); larger
 * when the pieces carry a tail, e.g. the shard's accept count) and is laid out
 * [max_rows][batch][16 (N-1)], of which the first rows_per_rank[r] rows count */
int epv_reduce_gathered_rows(epv_ctx *ctx, const double *d_gathered, uint32_t world, uint64_t max_rows,
                             uint64_t piece_doubles, const uint64_t *rows_per_rank, uint64_t batch, int average,
                             double *J, double *D);

int epv_get_counters(epv_ctx *ctx, epv_counters *out);

/* Timing hook for bench.py: average duration (ms) of the colour-phase kernel launches
 * issued since the last call, measured with HIP events on the context's stream, and
 * how many launches that covers.  epv_set_timing(ctx, N): 0 = off, N >= 1 = events around every
 * N-th colour-phase launch (every launch costs ~4 % of a step at three contexts per GPU). */
int epv_kernel_time_ms(epv_ctx *ctx, double *avg_ms, uint64_t *n_launches);
int epv_set_timing(epv_ctx *ctx, int enabled);

#ifdef __cplusplus
}
#endif

#endif
