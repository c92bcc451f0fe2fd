// epievo_sim_pairwise -- MI355X drop-in for /root/reference/src/prog/epievo_sim_pairwise.cpp:
// sample a history on one branch between two observed sequences by `-L` MCMC sweeps.
// Flags: -L burn-in (10), -T evolutionary time (double, 1.0), -s seed, -o output
// local_paths (required), -p input paths, -v; positionals <param> <states>.
// The initial paths (initialize_paths_indep, :62-110) are drawn on the GPU as well
// (epv_init_paths_indep), so the whole program runs on the device end to end.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>
#include <stdexcept>

#include "epv_io.hpp"
#include "epv_model.hpp"
#include "epv_options.hpp"
#include "epv_sampler.hpp"

using std::cerr;
using std::endl;
using std::string;
using std::vector;

int main(int argc, const char **argv) {
  try {
    bool VERBOSE = false;
    string outfile, pathfile;
    size_t burnin = 10;
    double evolutionary_time = 1.0;
    size_t rng_seed = std::numeric_limits<size_t>::max();

    const string prog = string(argv[0]).substr(string(argv[0]).find_last_of('/') + 1);
    epv::OptionParser opt_parse(prog, "simulate a path between two observed sequences", "<param> <states>");
    opt_parse.add_opt("burnin", 'L', "MCMC burn-in length", false, burnin);
    opt_parse.add_opt("time", 'T', "evolutionary time", false, evolutionary_time);
    opt_parse.add_opt("seed", 's', "rng seed", false, rng_seed);
    opt_parse.add_opt("outfile", 'o', "output file of local paths", true, outfile);
    opt_parse.add_opt("paths", 'p', "input file of initial paths", false, pathfile);
    opt_parse.add_opt("verbose", 'v', "print more run info", false, VERBOSE);
    vector<string> leftover_args;
    opt_parse.parse(argc, argv, leftover_args);
    if (argc == 1 || opt_parse.help_requested()) {
      cerr << opt_parse.help_message() << endl << opt_parse.about_message() << endl;
      return EXIT_SUCCESS;
    }
    if (opt_parse.option_missing()) { cerr << opt_parse.option_missing_message() << endl; return EXIT_SUCCESS; }
    if (leftover_args.size() != 2) { cerr << opt_parse.help_message() << endl; return EXIT_SUCCESS; }
    const string param_file(leftover_args.front()), states_file(leftover_args.back());

    if (VERBOSE) cerr << "[READING PARAMETERS: " << param_file << "]" << endl;
    epv::Model the_model = epv::Model::read(param_file);
    the_model.scale_triplet_rates();

    if (rng_seed == std::numeric_limits<size_t>::max()) { std::random_device rd; rng_seed = rd(); }
    if (VERBOSE) cerr << "rng seed: " << rng_seed << endl;

    epv::Tree th = epv::Tree::single_branch(evolutionary_time);
    vector<string> names;
    vector<vector<uint8_t>> states;
    epv::read_states_file(states_file, names, states);
    if (states.size() != 2) throw std::runtime_error("states file must have exactly two sequences");
    th.node_names = names;

    epv::FlatPaths paths;
    epv::SingleSiteSampler mcmc(burnin, 1);
    if (!pathfile.empty()) {
      vector<string> nn;
      vector<double> tt;
      paths = epv::read_local_paths(pathfile, nn, tt);
      if (paths.n_nodes != 2 || tt[1] != evolutionary_time)
        throw std::runtime_error("input paths do not match a single branch of the given time");
      mcmc.reset(the_model, th, paths);
    } else {
      mcmc.init_paths_indep(the_model, th, states[0], states[1], rng_seed);
      paths.n_sites = states[0].size();
    }
    const size_t n_acc = mcmc.sweeps(burnin, rng_seed, 0);
    if (VERBOSE)
      cerr << "acceptance rate: " << (double)n_acc / ((double)burnin * (paths.n_sites - 2)) << endl;
    mcmc.download(paths);
    epv::write_local_paths(outfile, th.node_names, 2, paths.n_sites, th.branches.data(),
                           paths.init.data(), paths.offsets.data(), paths.jumps.data());
  } catch (const std::exception &e) {
    cerr << e.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
