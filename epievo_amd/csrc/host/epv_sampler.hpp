// epv_sampler.hpp -- host-side C++ face of the GPU sampler, shaped like the reference's
// SingleSiteSampler (/root/reference/src/libepievo/SingleSiteSampler.hpp:35-81) so that
// the EM driver reads like the reference's (epievo_est_params_histories.cpp:236-264).
// It is a wrapper over the C ABIs of include/epievo_mi355x.h (kernels) and
// include/epievo_mi355x_comm.h (RCCL); errors become std::runtime_error (the reference's mains
// catch std::exception and return EXIT_FAILURE, epievo_est_params_histories.cpp:296-299).
//
// Differences forced by the device boundary:
//  * paths live on the GPU between calls: reset(model, tree, paths) uploads them once,
//    reset(model) re-derives the cached log-likelihoods after a model change, and
//    download(paths) brings them back when the driver wants to write them;
//  * the reference threads one std::mt19937 through every call; the parallel schedule
//    uses a counter-based stream, so run_mcmc takes (seed, em_iteration) instead.
//
// Sharding (new; the reference is single-threaded).  The genome is cut into PARTS in genome
// order: G device slots (one per GPU of the node: EPV_DEVICES / the constructor's device list)
// times k contexts per GPU (EPV_CONTEXTS_PER_GPU, default 2).  Every part owns whole 256-site
// blocks plus redundant halo columns wide enough for a whole run_mcmc; the RNG and the colouring
// are keyed by the global site index, so the parts need no communication inside the E-step.
// Per EM iteration the slots exchange, device to device through RCCL,
//   (1) the edge columns of neighbouring parts before reset()   (epv_comm_exchange), and
//   (2) their rows of the J/D reduction tree after run_mcmc()   (epv_comm_all_gather);
// slots are cut on whole rows, every stage sums aligned subtrees of ONE balanced binary tree
// over the site index, and so paths, J, D and the acceptance rate are bit-identical to the
// one-context run for any G and k.  A device list with repeats
// (EPV_DEVICES=0,0,0,0) rehearses an N-GPU run on a smaller box through the loopback transport
// of the exchange layer.
//
// Two ways to place the GPU slots (include/epievo_mi355x_comm.h):
//   * every slot in THIS process (the CLIs, `bench.py --gpus N` called plainly): the constructor's
//     device list, one RCCL communicator rank per slot through ncclCommInitAll;
//   * one slot per process (RankSpec; torchrun-style launchers): the process owns slot `rank` of
//     `world`, reset() takes the OWNED columns of that slot (shard_cuts tells which), the halo
//     columns arrive from the neighbouring ranks before the first reset, and the communicator
//     comes from ncclCommInitRank with an id the launcher passes around.
// Both run the same code below; the loops simply cover the slots that live here.
#ifndef EPV_SAMPLER_HPP
#define EPV_SAMPLER_HPP

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "epv_io.hpp"
#include "epv_model.hpp"

struct epv_ctx;
struct epv_comm;

namespace epv {

// "all" | "0,1,2,3" | "" (-> {0}); repeats allowed (rehearsal).  Throws on a malformed list.
std::vector<int> parse_device_list(const std::string &spec);
// the device list of the environment (EPV_DEVICES), {0} when unset
std::vector<int> devices_from_env();

// one slot of a run whose other slots live in other processes
struct RankSpec {
  int device = 0;           // the GPU of this process
  int world = 1, rank = 0;  // slots of the run, and this one's place in genome order
  unsigned char id[128] = {0};   // epv_comm_get_unique_id of rank 0, passed around by the launcher
};

class SingleSiteSampler {
public:
  SingleSiteSampler(size_t n_burn_in, size_t n_batch, int device = 0, uint32_t capacity = 0);
  SingleSiteSampler(size_t n_burn_in, size_t n_batch, const std::vector<int> &devices, uint32_t capacity = 0);
  SingleSiteSampler(size_t n_burn_in, size_t n_batch, const RankSpec &rank, uint32_t capacity = 0);
  // cut points of `world` contiguous slots of an n-site genome (whole statistics rows of
  // 256 * row_blocks sites; every slot must be able to hold its halos): world + 1 entries, or fewer
  // when the genome cannot feed that many slots
  static std::vector<uint64_t> shard_cuts(uint64_t n_sites, size_t world, size_t n_burn_in, size_t n_batch,
                                          uint32_t row_blocks = 64);
  ~SingleSiteSampler();
  SingleSiteSampler(const SingleSiteSampler &) = delete;
  SingleSiteSampler &operator=(const SingleSiteSampler &) = delete;

  // SingleSiteSampler::reset (SingleSiteSampler.cpp:449-475)
  void reset(const Model &the_model, const Tree &th, const FlatPaths &paths);
  void reset(const Model &the_model);
  // one slot per process: `owned` = the columns [cuts[rank], cuts[rank + 1]) of an n_global-site genome
  void reset(const Model &the_model, const Tree &th, const FlatPaths &owned, uint64_t n_global);

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

  // the resident paths (one slot per process: the owned columns of this slot)
  void download(FlatPaths &paths);
  // EPV_OPT_* of include/epievo_mi355x.h on every context; HIP-event timing of the colour phases
  void set_options(uint32_t flags);
  void set_timing(int every);
  void kernel_time_ms(double &avg_ms, uint64_t &n_launches);
  uint32_t phase_mode();
  uint64_t halo_columns() const { return halo_; }

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
  bool SAMPLE_ROOT;  // hard-wired false in the reference (SingleSiteSampler.cpp:441); true = EPV_OPT_SAMPLE_ROOT on
                     // every context (root states are proposed too; reference-arithmetic kernels)
  size_t burn_in;
  size_t batch;

  // capacity overflows that were absorbed by widening the device jump slots (verbose output)
  std::vector<std::string> capacity_events;

  // how the genome is laid out right now (verbose output, tests)
  size_t n_parts() const { return parts_.size(); }
  size_t n_slots() const { return slots_.size(); }
  bool uses_rccl() const;
  std::string layout() const;
  int contexts_per_gpu() const { return contexts_wanted_ > 0 ? contexts_wanted_ : (n_nodes_ - 1 <= 8 ? 3 : 2); }

private:
  struct Part {            // one context: local columns [lo, hi), owned columns [a, b) of the genome
    epv_ctx *ctx = nullptr;
    size_t slot = 0;
    uint64_t lo = 0, a = 0, b = 0, hi = 0;
  };
  struct Slot {            // one GPU of the run (or one rehearsal slot on a shared GPU) that lives here
    int device = 0;
    size_t gidx = 0;                     // its place among the run's slots
    epv_comm *comm = nullptr;
    uint64_t first = 0, last = 0;        // owned columns [first, last) of the genome
    size_t part0 = 0, part1 = 0;         // its parts [part0, part1)
    uint64_t n_blocks = 0, n_rows = 0;
    void *d_blocks = nullptr;            // [batch][n_blocks][V] level-0 partials of the slot
    void *d_rows = nullptr;              // [max_rows][batch][V] rows of the slot (zero padded)
    void *d_gather = nullptr;            // [slots][max_rows][batch][V]
    void *d_halo[4] = {nullptr, nullptr, nullptr, nullptr};  // send prev, recv prev, send next, recv next
    uint64_t halo_bytes = 0;
  };
  void check(int rc, const char *what);
  void check_mcmc(int rc, const char *what);
  void check_on(epv_ctx *c, int rc, const char *what);
  void check_comm(epv_comm *c, int rc, const char *what);
  bool sharded() const { return !parts_.empty(); }
  void drop_parts();          // back to the single context ctx_
  void refresh_parts();       // equal capacities, halo columns of every inner edge, fresh halo marks
  void apply_sample_root();
  void build(const Tree &th, const FlatPaths &paths, uint64_t n_global, bool rank_mode);
  std::vector<epv_ctx *> contexts() const;
  void equalize_capacity();
  void ensure_stat_buffers();
  void free_stat_buffers();
  epv_ctx *ctx_;              // the context of the unsharded paths (== parts_[0].ctx when sharded)
  std::vector<int> devices_;
  int contexts_wanted_ = 0;      // contexts per GPU; 0 = by tree size (3 up to 8 branches: fused colour phase; else 2)
  uint32_t row_blocks_ = 64;  // 256-site blocks per row of the cross-GPU statistics stage
  bool force_comm_ = false;   // EPV_FORCE_COMM=1: the exchange layer even for one slot (tests)
  std::vector<Part> parts_;
  std::vector<Slot> slots_;
  uint64_t halo_ = 0;         // halo columns at every inner edge (multiple of 256)
  uint64_t max_rows_ = 0, stat_batch_ = 0;
  std::vector<uint64_t> rows_of_slot_;   // statistics rows of every slot of the run
  uint32_t capacity_;
  int n_nodes_ = 0;
  uint64_t n_sites_ = 0;      // genome length (all slots)
  size_t world_ = 1;          // slots of the run (== slots_.size() unless one slot per process)
  bool rank_mode_ = false;
  RankSpec rank_;
  epv_comm *rank_comm_ = nullptr;   // one slot per process: the communicator outlives the parts
};

}  // namespace epv

#endif
