// epievo_sim -- drop-in for /root/reference/src/prog/epievo_sim.cpp: forward simulation of
// epigenome evolution (root sequence from the stationary chain, Gillespie events along
// every branch) -> states file + global_jumps file.  Same flags (-n -p -s -r -t -T -l
// -unscaled-param -scale-time -R -v, <params-file> <outfile>), same formats, and -- for a
// given seed -- the same random draws in the same order, hence identical outputs
// (tests/test_forward_sim.py).  The event chain is strictly sequential: host code.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>
#include <stdexcept>

#include "epv_forward.hpp"
#include "epv_io.hpp"
#include "epv_model.hpp"
#include "epv_options.hpp"

using std::cerr;
using std::endl;
using std::string;
using std::vector;

static bool file_is_readable(const string &f) { std::ifstream in(f); return in.good(); }

int main(int argc, const char **argv) {
  try {
    string pathfile, tree_file, root_states_file;
    bool VERBOSE = false, unscaled_model_params = false, scale_time = false, TRPARAM = false,
         write_only_leaves = false;
    size_t n_sites = 100, n_threads = 1;
    double evolutionary_time = std::numeric_limits<double>::lowest();
    size_t rng_seed = std::numeric_limits<size_t>::max();

    const string prog = string(argv[0]).substr(string(argv[0]).find_last_of('/') + 1);
    epv::OptionParser opt_parse(prog, "simulate epigenome evolution", "<params-file> <outfile>");
    opt_parse.add_opt("n-sites", 'n', "length of sequence to simulate", false, n_sites);
    opt_parse.add_opt("paths", 'p', "name of output file for evolution paths as sorted jump times", false, pathfile);
    opt_parse.add_opt("seed", 's', "rng seed", false, rng_seed);
    opt_parse.add_opt("root", 'r', "root states file", false, root_states_file);
    opt_parse.add_opt("tree", 't', "Newick format tree file", false, tree_file);
    opt_parse.add_opt("evo-time", 'T', "evolutionary time", false, evolutionary_time);
    opt_parse.add_opt("leaf", 'l', "write only leaf states (default: all nodes)", false, write_only_leaves);
    opt_parse.add_opt("unscaled-param", '\0', "do not scale model parameters", false, unscaled_model_params);
    opt_parse.add_opt("scale-time", '\0', "scale time", false, scale_time);
    opt_parse.add_opt("rates", 'R', "use triplet transition rates", false, TRPARAM);
    opt_parse.add_opt("verbose", 'v', "print more run info", false, VERBOSE);
    // extension: sibling subtrees on their own threads, one generator per branch seeded from
    // (seed, node).  1 (default) = the reference's single sequential stream
    opt_parse.add_opt("threads", 'j', "simulate sibling subtrees in parallel on this many threads "
                      "(other random stream than the default)", false, n_threads);
    vector<string> leftover_args;
    opt_parse.parse(argc, argv, leftover_args);
    if (argc == 1 || opt_parse.help_requested()) {
      cerr << opt_parse.help_message() << endl << opt_parse.about_message() << endl;
      return EXIT_SUCCESS;
    }
    if (leftover_args.size() != 2) { cerr << opt_parse.help_message() << endl; return EXIT_SUCCESS; }
    const string param_file(leftover_args.front()), outfile(leftover_args.back());
    if (!file_is_readable(param_file)) { cerr << "cannot read file: " << param_file << endl; return EXIT_SUCCESS; }
    if (!tree_file.empty()) {
      if (evolutionary_time != std::numeric_limits<double>::lowest()) {
        cerr << "specify exactly one of: tree or time" << endl;
        return EXIT_SUCCESS;
      }
      if (!file_is_readable(tree_file)) { cerr << "cannot read file: " << tree_file << endl; return EXIT_SUCCESS; }
    } else if (evolutionary_time == std::numeric_limits<double>::lowest()) {
      cerr << "specify exactly one of: tree or time" << endl;
      return EXIT_SUCCESS;
    }

    if (VERBOSE) cerr << "reading parameter file: " << param_file << endl;
    epv::Model the_model = epv::Model::read(param_file);
    if (scale_time) evolutionary_time /= epv::rate_scaling_factor(the_model.rates);
    if (!unscaled_model_params) the_model.scale_triplet_rates();

    epv::Tree th;
    if (!tree_file.empty()) {
      if (VERBOSE) cerr << "reading tree file: " << tree_file << endl;
      th = epv::Tree::read(tree_file);
    } else {
      if (VERBOSE) cerr << "[initializing two node tree with time: " << evolutionary_time << "]" << endl;
      th = epv::Tree::single_branch(evolutionary_time);
    }
    if (rng_seed == std::numeric_limits<size_t>::max()) { std::random_device rd; rng_seed = rd(); }
    if (VERBOSE) cerr << "[rng seed: " << rng_seed << "]" << endl;
    std::mt19937 gen(rng_seed);

    vector<uint8_t> root_seq;
    if (root_states_file.empty()) {
      if (VERBOSE) cerr << "[SIMULATING: " << th.node_names[0] << " (ROOT)]" << endl;
      epv::sample_root(the_model, n_sites, gen, root_seq);
    } else {
      if (VERBOSE) cerr << "[READING ROOT FILE: " << root_states_file << "]" << endl;
      vector<string> names;
      vector<vector<uint8_t>> seqs;
      epv::read_states_file(root_states_file, names, seqs);
      root_seq = seqs.front();
      n_sites = root_seq.size();
    }
    if (VERBOSE) cerr << "[ROOT LENGTH: " << n_sites << "]" << endl;

    vector<vector<uint8_t>> sequences;
    vector<vector<epv::GlobalJump>> paths;
    vector<size_t> events;
    if (n_threads > 1) {
      if (VERBOSE) cerr << "[PARALLEL MODE: one generator per branch, up to " << n_threads << " threads]" << endl;
      epv::simulate_tree_parallel(the_model, th, root_seq, rng_seed, (int)n_threads, sequences, paths, events);
    } else {
      epv::simulate_tree(the_model, th, root_seq, gen, sequences, paths, events);
    }
    epv::write_global_jumps(pathfile, th.node_names, root_seq, paths);
    if (VERBOSE) {
      cerr << "[FREQUENCIES OF SAMPLED EVENTS]" << endl;
      size_t total = 0;
      for (size_t i = 0; i < 8; ++i) { cerr << ((i >> 2) & 1) << ((i >> 1) & 1) << (i & 1) << '\t' << events[i] << endl; total += events[i]; }
      cerr << "[TOTAL SAMPLED EVENTS: " << total << "]" << endl;
      cerr << "[WRITING EPIGENOMIC STATES]" << endl;
    }
    epv::write_states(outfile, write_only_leaves, th, sequences);
  } catch (const std::exception &e) {
    cerr << e.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
