// global_jumps_to_paths -- drop-in for /root/reference/src/prog/global_jumps_to_paths.cpp:
// (states file, global_jumps file) -> local_paths file.  Flags -t tree, -T evo-time, -v;
// positionals <statefile> <jumpfile> <outfile>.
#include <cstdlib>
#include <iostream>
#include <limits>
#include <stdexcept>

#include "epv_forward.hpp"
#include "epv_io.hpp"
#include "epv_options.hpp"

using std::cerr;
using std::endl;
using std::string;
using std::vector;

int main(int argc, const char **argv) {
  try {
    bool VERBOSE = false;
    string treefile;
    double evolutionary_time = std::numeric_limits<double>::max();
    const string prog = string(argv[0]).substr(string(argv[0]).find_last_of('/') + 1);
    epv::OptionParser opt_parse(prog, "convert path file format", "<statefile> <jumpfile> <outfile>");
    opt_parse.add_opt("tree", 't', "Newick format tree file", false, treefile);
    opt_parse.add_opt("evo-time", 'T', "evolutionary time", false, evolutionary_time);
    opt_parse.add_opt("verbose", 'v', "print more run info", false, VERBOSE);
    vector<string> leftover_args;
    opt_parse.parse(argc, argv, leftover_args);
    if (argc == 1 || opt_parse.help_requested()) {
      cerr << opt_parse.help_message() << endl << opt_parse.about_message() << endl;
      return EXIT_SUCCESS;
    }
    if (leftover_args.size() != 3) { cerr << opt_parse.help_message() << endl; return EXIT_SUCCESS; }
    const string statesfile(leftover_args[0]), pathsfile(leftover_args[1]), outfile(leftover_args[2]);

    epv::Tree th = treefile.empty() ? epv::Tree::single_branch(evolutionary_time) : epv::Tree::read(treefile);
    if (VERBOSE) cerr << "[READING JUMPS: " << pathsfile << "]" << endl;
    vector<uint8_t> root;
    vector<string> names_paths;
    vector<vector<epv::GlobalJump>> the_paths;
    epv::read_global_jumps(pathsfile, root, names_paths, the_paths);
    if (VERBOSE) cerr << "[READING STATES FILE: " << statesfile << "]" << endl;
    vector<vector<uint8_t>> the_states;
    vector<string> names_states;
    epv::read_states_file(statesfile, names_states, the_states);
    // the reference asserts this (global_jumps_to_paths.cpp:150-151)
    if (th.node_names != names_states || th.node_names != names_paths)
      throw std::runtime_error("node names of tree, states file and jumps file differ");
    if ((int)the_paths.size() != th.n_nodes()) throw std::runtime_error("jumps file and tree differ in size");
    const epv::FlatPaths fp = epv::global_to_local(th, the_states, the_paths);
    if (VERBOSE) cerr << "[WRITING PATHS: " << outfile << "]" << endl;
    epv::write_local_paths(outfile, th.node_names, th.n_nodes(), fp.n_sites, th.branches.data(),
                           fp.init.data(), fp.offsets.data(), fp.jumps.data());
  } catch (const std::exception &e) {
    cerr << e.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
