// epievo_est_params_histories -- MI355X drop-in for the reference's MCEM driver
// (/root/reference/src/prog/epievo_est_params_histories.cpp:93-300): same flags, same
// positionals, same param / Newick / local_paths formats, same per-iteration outputs and
// -v TSV line.  The E-step (reset + run_mcmc) and scale_jump_times run on the GPU; the
// O(8) M-step runs on the host.
// One extension: -g/--gpus <list> (or the environment variable EPV_DEVICES) names the GPUs of
// the node to shard the sites over -- "all", or HIP device ids "0,1,2,3" -- with RCCL between
// them (epv_sampler.hpp); results do not depend on the list.  A list with repeats ("0,0,0,0")
// rehearses a multi-GPU run on fewer GPUs.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>
#include <stdexcept>
#include <chrono>
#include <cmath>
#include <thread>

#include "epv_io.hpp"
#include "epv_model.hpp"
#include "epv_options.hpp"
#include "epv_sampler.hpp"

using std::cerr;
using std::endl;
using std::string;
using std::vector;

static string strip_path(const string &full) {
  const size_t p = full.find_last_of('/');
  return p == string::npos ? full : full.substr(p + 1);
}

int main(int argc, const char **argv) {
  try {
    bool VERBOSE = false, single_branch = false, optimize_branches = false;
    string outfile, param_file_updated, tree_file, treefile_updated, gpu_list;
    size_t iteration = 10, batch = 10, burnin = 10, paths_every = 1;
    size_t rng_seed = std::numeric_limits<size_t>::max();
    static const double param_tol = 1e-10;

    epv::OptionParser opt_parse(strip_path(argv[0]), "estimate parameters and evolutionary histories",
                                "<param> (<treefile>) <path_file>");
    opt_parse.add_opt("iteration", 'i', "number of MCMC-EM iteration", false, iteration);
    opt_parse.add_opt("batch", 'B', "number of MCMC iteration", false, batch);
    opt_parse.add_opt("burnin", 'L', "MCMC burn-in length", false, burnin);
    opt_parse.add_opt("seed", 's', "rng seed", false, rng_seed);
    opt_parse.add_opt("outfile", 'o', "output file of local paths", true, outfile);
    opt_parse.add_opt("outparam", 'p', "output file of parameters", false, param_file_updated);
    opt_parse.add_opt("outtree", 't', "output file of tree", false, treefile_updated);
    opt_parse.add_opt("single_branch", 'T', "pairwise process (assumes no tree)", false, single_branch);
    opt_parse.add_opt("branch", 'b', "optimize branch lengths", false, optimize_branches);
    opt_parse.add_opt("verbose", 'v', "print more run info", false, VERBOSE);
    opt_parse.add_opt("gpus", 'g', "GPUs to shard the sites over: all | 0,1,.. (default: EPV_DEVICES or 0)", false,
                      gpu_list);
    // extension: the reference rewrites the whole paths file after EVERY iteration (:280-283; the file
    // doubles as a checkpoint).  -e k keeps that for every k-th iteration and the last one: the final
    // file is the same bytes, the ones in between are not written
    opt_parse.add_opt("paths-every", 'e', "write the paths file every k-th iteration and after the last (default 1: "
                      "every iteration, as the reference does)", false, paths_every);
    vector<string> leftover_args;
    opt_parse.parse(argc, argv, leftover_args);
    if (paths_every == 0) paths_every = 1;
    if (argc == 1 || opt_parse.help_requested()) {
      cerr << opt_parse.help_message() << endl << opt_parse.about_message() << endl;
      return EXIT_SUCCESS;
    }
    if (opt_parse.option_missing()) {
      cerr << opt_parse.option_missing_message() << endl;
      return EXIT_SUCCESS;
    }
    if (leftover_args.size() == 2) {
      if (!single_branch) { cerr << opt_parse.help_message() << endl; return EXIT_SUCCESS; }
    } else if (leftover_args.size() != 3) {
      cerr << opt_parse.help_message() << endl;
      return EXIT_SUCCESS;
    } else {
      tree_file = leftover_args[1];
    }
    const string param_file(leftover_args.front()), input_file(leftover_args.back());

    if (VERBOSE) cerr << "[READING PARAMETERS: " << param_file << "]" << endl;
    epv::Model the_model = epv::Model::read(param_file);
    the_model.scale_triplet_rates();

    if (VERBOSE) cerr << "[READING PATHS FILE: " << input_file << "]" << endl;
    vector<string> node_names;
    vector<double> tot_times;
    epv::FlatPaths paths = epv::read_local_paths(input_file, node_names, tot_times);

    epv::Tree th;
    if (single_branch) {
      if (VERBOSE) cerr << "[INITIALIZING TWO NODE TREE WITH TIME: " << tot_times.back() << "]" << endl;
      th = epv::Tree::single_branch(tot_times.back());
    } else {
      if (VERBOSE) cerr << "[READING TREE: " << tree_file << "]" << endl;
      th = epv::Tree::read(tree_file);
    }
    if (th.n_nodes() != paths.n_nodes)
      throw std::runtime_error("tree and paths file have different numbers of nodes");
    // The reference compares nothing here: branch lengths come from the tree, each Path keeps the
    // tot_time of the file, and scale_jump_times (ParamEstimation.cpp:369-380) brings the two
    // together at the end of the first iteration.  epievo_initialization without -b writes
    // rate-scaled tot_times next to an unscaled tree, so a mismatch is an ordinary input.  The
    // device keeps one length per branch, hence the same rescaling is applied at load time.
    for (int b = 1; b < th.n_nodes(); ++b)
      if (tot_times[b] != th.branches[b]) {
        if (!(tot_times[b] > 0.0) || !std::isfinite(tot_times[b]))
          throw std::runtime_error("paths of node " + th.node_names[b] + ": tot_time must be positive and finite");
        const double scale = th.branches[b] / tot_times[b];
        const uint64_t n = paths.n_sites;
        for (uint64_t k = paths.offsets[(uint64_t)(b - 1) * n]; k < paths.offsets[(uint64_t)b * n]; ++k)
          paths.jumps[k] *= scale;
        // always reported: first-iteration statistics of such an input differ from the reference's,
        // which keeps the file's tot_time until scale_jump_times ("tot_time")
        cerr << "[RESCALING PATHS OF NODE " << th.node_names[b] << ": tot_time " << tot_times[b]
             << " -> branch length " << th.branches[b] << "]" << endl;
      }

    if (rng_seed == std::numeric_limits<size_t>::max()) {
      std::random_device rd;
      rng_seed = rd();
    }
    if (VERBOSE) {
      cerr << "rng seed: " << rng_seed << endl;
      cerr << "itr\tstationary\tbaseline\tinit\tacc_rate\tllh\t\n";
      cerr << "0" << "\t" << the_model.T[0] << "\t" << the_model.T[3] << "\t" << the_model.baseline[0]
           << "\t" << the_model.baseline[3] << endl;
    }

    // The reference rewrites the paths file after every iteration (:280-283).  Formatting and
    // writing 111 MB (n = 1e6, tree.nwk) takes longer than the E-step of the next iteration, and
    // that E-step needs only the device-resident paths: the file of iteration i is written by a
    // background thread while the GPU runs iteration i + 1 (same bytes, same order of files).
    std::string writer_error;
    epv::FlatPaths out_paths;
    vector<double> out_branches;
    epv::SingleSiteSampler mcmc(burnin, batch,
                                gpu_list.empty() ? epv::devices_from_env() : epv::parse_device_list(gpu_list));
    // declared AFTER everything the thread references (and after the sampler): on an exception the
    // join runs first, while out_paths, out_branches and writer_error are still alive, and the file
    // of the last completed iteration is written out in full, as the synchronous reference leaves it
    std::thread writer;
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{writer};
    // EPV_CLI_TIMING=1: where the wall clock of the iterations went (stderr, at the end)
    const bool timing = std::getenv("EPV_CLI_TIMING") != nullptr;
    double t_reset = 0, t_mcmc = 0, t_mstep = 0, t_scale = 0, t_wait = 0, t_download = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (size_t itr = 0; itr < iteration; itr++) {
      double t0 = now();
      if (itr == 0) {
        mcmc.reset(the_model, th, paths);
        if (VERBOSE) cerr << "[GPU LAYOUT: " << mcmc.layout() << "]" << endl;
      } else {
        mcmc.reset(the_model);
      }

      t_reset += now() - t0; t0 = now();
      double acceptance_rate;
      vector<vector<double>> J_accum, D_accum;
      mcmc.run_mcmc(rng_seed, itr, J_accum, D_accum, acceptance_rate);
      t_mcmc += now() - t0; t0 = now();

      /* PARAMETER ESTIMATION (host) */
      const int B = th.n_nodes() - 1;
      vector<double> J(B * 8), D(B * 8);
      for (int b = 1; b <= B; ++b)
        for (int i = 0; i < 8; ++i) { J[(b - 1) * 8 + i] = J_accum[b][i]; D[(b - 1) * 8 + i] = D_accum[b][i]; }
      double llh = 0.0;
      if (!optimize_branches) {
        llh = epv::estimate_rates(param_tol, th.n_nodes(), J.data(), D.data(), the_model);
        epv::set_one_change_per_site_per_unit_time(the_model.rates, th.branches);
      } else {
        llh = epv::estimate_rates_and_branches(param_tol, th.n_nodes(), J.data(), D.data(), th.branches,
                                               the_model);
      }
      t_mstep += now() - t0; t0 = now();
      mcmc.scale_jump_times(th.branches);
      t_scale += now() - t0;

      if (VERBOSE)
        cerr << itr + 1 << "\t" << the_model.T[0] << "\t" << the_model.T[3] << "\t"
             << the_model.baseline[0] << "\t" << the_model.baseline[3] << "\t" << acceptance_rate
             << "\t" << llh << endl;

      if (!param_file_updated.empty()) {
        std::ofstream out_param(param_file_updated);
        // EPV_TEST_FAIL_PARAM_AT=k (tests): pretend the k-th rewrite failed, to exercise the error path
        // while the previous iteration's paths file is still being written
        const char *fail_at = std::getenv("EPV_TEST_FAIL_PARAM_AT");
        if (fail_at && (size_t)std::atol(fail_at) == itr + 1) out_param.setstate(std::ios::failbit);
        if (!out_param) throw std::runtime_error("bad output param file: " + param_file_updated);
        out_param << the_model.format_for_param_file() << endl;
      }
      if ((itr + 1) % paths_every != 0 && itr + 1 != iteration) {
        if (optimize_branches && !treefile_updated.empty()) {
          std::ofstream out_tree(treefile_updated);
          if (!out_tree) throw std::runtime_error("bad output param file: " + treefile_updated);
          out_tree << th.newick() << endl;
        }
        continue;
      }
      t0 = now();
      if (writer.joinable()) writer.join();
      if (!writer_error.empty()) throw std::runtime_error(writer_error);
      t_wait += now() - t0; t0 = now();
      mcmc.download(out_paths);
      t_download += now() - t0;
      out_branches = th.branches;
      writer = std::thread([&outfile, &th, &out_paths, &out_branches, &writer_error] {
        try {
          epv::write_local_paths(outfile, th.node_names, th.n_nodes(), out_paths.n_sites, out_branches.data(),
                                 out_paths.init.data(), out_paths.offsets.data(), out_paths.jumps.data());
        } catch (const std::exception &e) { writer_error = e.what(); }
      });
      if (optimize_branches && !treefile_updated.empty()) {
        std::ofstream out_tree(treefile_updated);
        if (!out_tree) throw std::runtime_error("bad output param file: " + treefile_updated);
        out_tree << th.newick() << endl;
      }
    }
    const double t0 = now();
    if (writer.joinable()) writer.join();
    if (!writer_error.empty()) throw std::runtime_error(writer_error);
    t_wait += now() - t0;
    if (timing)
      cerr << "[TIMING over " << iteration << " iterations, seconds: reset " << t_reset << ", run_mcmc " << t_mcmc
           << ", M-step " << t_mstep << ", scale_jump_times " << t_scale << ", waiting for the previous file "
           << t_wait << ", download " << t_download << "]" << endl;
  } catch (const std::exception &e) {
    cerr << e.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
