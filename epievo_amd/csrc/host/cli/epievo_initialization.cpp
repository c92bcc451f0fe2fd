// epievo_initialization -- MI355X drop-in for /root/reference/src/prog/epievo_initialization.cpp:
// from the states observed at the leaves, fit a site-independent 2-rate model by EM, draw
// initial histories from it, and derive initial parameters of the context-dependent model --
// i.e. produce the inputs of epievo_est_params_histories.  Same flags and positionals
// (-v -s -i -B -p -t -T -o -b, (<tree-file>) <states-file>), same file formats.
// The O(n) parts (conditional expectations, path resampling, both kinds of sufficient
// statistics, rescaling) run on the GPU; the O(nodes) M-steps and the one-pass heuristic
// start run on the host.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>
#include <stdexcept>

#include "epv_indep.hpp"
#include "epv_io.hpp"
#include "epv_model.hpp"
#include "epv_options.hpp"
#include "epv_sampler.hpp"

using std::cerr;
using std::endl;
using std::string;
using std::vector;

static const uint32_t EPV_INDEP_SWEEP_BASE = 0xF0000000u;  // random-stream ids of the resampling passes

int main(int argc, const char **argv) {
  try {
    static const double param_tol = 1e-10;
    bool VERBOSE = false, optimize_branches = false;
    double evolutionary_time = 0.0;
    size_t rng_seed = std::numeric_limits<size_t>::max();
    size_t iterations = 10, batch = 10;
    string paramfile, pathfile, tree_file, treefile_updated;

    const string prog = string(argv[0]).substr(string(argv[0]).find_last_of('/') + 1);
    epv::OptionParser opt_parse(prog, "generate initial paths and parameters given states at leaves",
                                "(<tree-file>) <states-file>");
    opt_parse.add_opt("verbose", 'v', "print more run info", false, VERBOSE);
    opt_parse.add_opt("seed", 's', "rng seed", false, rng_seed);
    opt_parse.add_opt("iterations", 'i', "number of iterations", false, iterations);
    opt_parse.add_opt("batch", 'B', "number of MCMC iteration", false, batch);
    opt_parse.add_opt("param", 'p', "output file of parameters", false, paramfile);
    opt_parse.add_opt("outtree", 't', "output file of tree", false, treefile_updated);
    opt_parse.add_opt("evo-time", 'T', "evolutionary time (assumes no tree)", false, evolutionary_time);
    opt_parse.add_opt("path", 'o', "output file of local paths (default: stdout)", false, pathfile);
    opt_parse.add_opt("branch", 'b', "optimize branch lengths as well", false, optimize_branches);
    vector<string> leftover_args;
    opt_parse.parse(argc, argv, leftover_args);
    if (argc == 1 || opt_parse.help_requested()) {
      cerr << opt_parse.help_message() << endl << opt_parse.about_message() << endl;
      return EXIT_SUCCESS;
    }
    if (leftover_args.size() == 1) {
      if (evolutionary_time == 0.0) { cerr << opt_parse.help_message() << endl; return EXIT_SUCCESS; }
    } else if (leftover_args.size() != 2) {
      cerr << opt_parse.help_message() << endl;
      return EXIT_SUCCESS;
    } else {
      tree_file = leftover_args.front();
    }
    const string statesfile(leftover_args.back());

    epv::Tree th;
    if (evolutionary_time > 0.0) {
      if (VERBOSE) cerr << "[INITIALIZING TWO NODE TREE WITH TIME: " << evolutionary_time << "]" << endl;
      th = epv::Tree::single_branch(evolutionary_time);
    } else {
      if (VERBOSE) cerr << "[READING TREE: " << tree_file << "]" << endl;
      th = epv::Tree::read(tree_file);
    }
    if (VERBOSE) cerr << "[READING STATES FILE: " << statesfile << "]" << endl;
    vector<vector<uint8_t>> state_sequences = epv::read_states_for_tree(statesfile, th);

    if (rng_seed == std::numeric_limits<size_t>::max()) { std::random_device rd; rng_seed = rd(); }
    if (VERBOSE) cerr << "rng seed: " << rng_seed << endl;

    /* generate initial paths by heuristics (host) */
    epv::FlatPaths paths = epv::initialize_paths_heuristic(rng_seed, th, state_sequences);

    epv::SingleSiteSampler gpu(0, batch, 0, 32);
    gpu.upload(th, paths);

    /* Run EM to learn a site-independent model */
    double rates[2] = {0.0, 0.0};
    vector<double> J, D;
    gpu.indep_sufficient_statistics(J, D);
    if (VERBOSE) {
      cerr << "itr\trate0\trate1\t\n";
      cerr << "0" << "\t" << rates[0] << "\t" << rates[1] << endl;
    }
    for (size_t itr = 0; itr < iterations; itr++) {
      if (!optimize_branches) {
        epv::estimate_rates_indep(th.n_nodes(), J.data(), D.data(), rates);
      } else {
        epv::estimate_rates_and_branches_indep(th.n_nodes(), J.data(), D.data(), rates, th.branches);
        gpu.scale_jump_times(th.branches);
      }
      gpu.indep_expectation(rates, J, D);
      if (VERBOSE) cerr << itr + 1 << "\t" << rates[0] << "\t" << rates[1] << endl;
    }

    /* Re-sample a better initial path (sample_summary_stats, epievo_initialization.cpp:188-232) */
    const int B = th.n_nodes() - 1;
    vector<double> J_trip(B * 8, 0.0), D_trip(B * 8, 0.0);
    for (size_t i = 0; i < batch; i++) {
      gpu.indep_update_paths(rates, rng_seed, EPV_INDEP_SWEEP_BASE + (uint32_t)i);
      vector<vector<double>> J1, D1;
      gpu.get_sufficient_statistics(J1, D1);
      for (int b = 1; b <= B; ++b)
        for (int k = 0; k < 8; ++k) { J_trip[(b - 1) * 8 + k] += J1[b][k]; D_trip[(b - 1) * 8 + k] += D1[b][k]; }
    }
    for (double &v : J_trip) v /= batch;
    for (double &v : D_trip) v /= batch;

    /* Generate initial parameters of context-dependent model */
    if (VERBOSE) cerr << "[CONSTRUCTING EPIEVO MODEL]" << endl;
    epv::Model the_model = epv::model_from_indep_rates(rates);
    if (!optimize_branches) {
      epv::estimate_rates(param_tol, th.n_nodes(), J_trip.data(), D_trip.data(), the_model);
      epv::set_one_change_per_site_per_unit_time(the_model.rates, th.branches);
    } else {
      epv::estimate_rates_and_branches(param_tol, th.n_nodes(), J_trip.data(), D_trip.data(), th.branches,
                                       the_model);
    }
    gpu.scale_jump_times(th.branches);

    if (VERBOSE) cerr << "[WRITING PATHS]" << endl;
    gpu.download(paths);
    epv::write_local_paths(pathfile.empty() ? "/dev/stdout" : pathfile, th.node_names, th.n_nodes(),
                           paths.n_sites, th.branches.data(), paths.init.data(), paths.offsets.data(),
                           paths.jumps.data());
    if (!paramfile.empty()) {
      std::ofstream of_param(paramfile);
      if (!of_param) throw std::runtime_error("bad output param file: " + paramfile);
      of_param << the_model.format_for_param_file() << endl;
    } else {
      std::cout << the_model.format_for_param_file() << endl;
    }
    if (optimize_branches) {
      if (!treefile_updated.empty()) {
        std::ofstream of_tree(treefile_updated);
        if (!of_tree) throw std::runtime_error("bad output param file: " + treefile_updated);
        of_tree << th.newick() << endl;
      } else {
        std::cout << th.newick() << endl;
      }
    }
  } catch (const std::exception &e) {
    cerr << e.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
