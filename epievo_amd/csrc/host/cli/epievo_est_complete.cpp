// epievo_est_complete -- drop-in for /root/reference/src/prog/epievo_est_complete.cpp:
// the M-step alone, from complete histories.  Flags -v -b -T(switch) -o(required) -t;
// positionals <param> (<treefile>) <path_file>.  The sufficient statistics come from the
// GPU (epv_get_sufficient_statistics); the fit is the host M-step.
// Known difference: with -b the reference normalises path lengths to 1 for every site
// EXCEPT site 0 (its loop starts at index 1 of the site-major array,
// ParamEstimation.cpp:430-434); this program normalises all sites.
#include <cstdlib>
#include <fstream>
#include <iostream>
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
    static const double param_tol = 1e-10;
    bool VERBOSE = false, optimize_branches = false, single_branch = false;
    string outfile, tree_file, treefile_updated;
    const string prog = string(argv[0]).substr(string(argv[0]).find_last_of('/') + 1);
    epv::OptionParser opt_parse(prog, "estimate parameters from complete data (site-specific paths)",
                                "<param> (<treefile>) <path_file>");
    opt_parse.add_opt("verbose", 'v', "print more run info", false, VERBOSE);
    opt_parse.add_opt("branch", 'b', "optimize branch lengths as well", false, optimize_branches);
    opt_parse.add_opt("single_branch", 'T', "pairwise process (assumes no tree)", false, single_branch);
    opt_parse.add_opt("output", 'o', "output parameter file", true, outfile);
    opt_parse.add_opt("outtree", 't', "output file of tree", false, treefile_updated);
    vector<string> leftover_args;
    opt_parse.parse(argc, argv, leftover_args);
    if (argc == 1 || opt_parse.help_requested()) {
      cerr << opt_parse.help_message() << endl << opt_parse.about_message() << endl;
      return EXIT_SUCCESS;
    }
    if (opt_parse.option_missing()) { cerr << opt_parse.option_missing_message() << endl; return EXIT_SUCCESS; }
    if (leftover_args.size() == 2) {
      if (!single_branch) { cerr << opt_parse.help_message() << endl; return EXIT_SUCCESS; }
    } else if (leftover_args.size() != 3) {
      cerr << opt_parse.help_message() << endl;
      return EXIT_SUCCESS;
    } else {
      tree_file = leftover_args[1];
    }
    const string param_file(leftover_args.front()), path_file(leftover_args.back());

    if (VERBOSE) cerr << "[READING PATHS: " << path_file << "]" << endl;
    vector<string> node_names;
    vector<double> tot_times;
    epv::FlatPaths paths = epv::read_local_paths(path_file, node_names, tot_times);
    epv::Tree th = single_branch ? epv::Tree::single_branch(tot_times.back()) : epv::Tree::read(tree_file);
    if (th.n_nodes() != paths.n_nodes) throw std::runtime_error("tree and paths file have different numbers of nodes");
    if (VERBOSE) cerr << "[READING PARAMETER FILE: " << param_file << "]" << endl;
    epv::Model the_model = epv::Model::read(param_file);
    the_model.scale_triplet_rates();

    epv::SingleSiteSampler gpu(0, 1);
    epv::Tree th_paths = th;
    for (int b = 1; b < th.n_nodes(); ++b) th_paths.branches[b] = tot_times[b];
    gpu.upload(th_paths, paths);
    const int B = th.n_nodes() - 1;
    vector<double> J(B * 8), D(B * 8);
    auto stats = [&]() {
      vector<vector<double>> Jv, Dv;
      gpu.get_sufficient_statistics(Jv, Dv);
      for (int b = 1; b <= B; ++b)
        for (int i = 0; i < 8; ++i) { J[(b - 1) * 8 + i] = Jv[b][i]; D[(b - 1) * 8 + i] = Dv[b][i]; }
    };
    if (!optimize_branches) {
      stats();
      epv::estimate_rates(param_tol, th.n_nodes(), J.data(), D.data(), the_model);
    } else {
      // estimate_rates_and_branches(paths overload), ParamEstimation.cpp:425-446: unit-length paths
      gpu.scale_jump_times(vector<double>(th.n_nodes(), 1.0));
      stats();
      epv::estimate_rates_and_branches(param_tol, th.n_nodes(), J.data(), D.data(), th.branches, the_model);
    }
    if (VERBOSE) cerr << "[WRITING PARAMETERS]" << endl;
    std::ofstream out(outfile);
    if (!out) throw std::runtime_error("bad output file: " + outfile);
    out << the_model.format_for_param_file() << endl;
    if (optimize_branches && !treefile_updated.empty()) {
      std::ofstream out_tree(treefile_updated);
      if (!out_tree) throw std::runtime_error("bad output param file: " + treefile_updated);
      out_tree << th.newick() << endl;
    }
  } catch (const std::exception &e) {
    cerr << e.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
