// epievo_sim -- drop-in for /root/reference/src/prog/epievo_sim.cpp: forward simulation of
// epigenome evolution (root sequence from the stationary chain, Gillespie events along
// every branch) -> states file + global_jumps file.  Same flags (-n -p -s -r -t -T -l
// -unscaled-param -scale-time -R -v, <params-file> <outfile>), same formats, and -- for a
// given seed -- the same random draws in the same order, hence identical outputs
// (tests/test_forward_sim.py).  That event chain is strictly sequential: host code.
//
// -g <device> (extension): the same process simulated on the GPU, site-parallel, by thinning with
// keyed randomness (epv_forward_simulate, csrc/epv_forward.h) -- another random stream than the
// reference's std::mt19937 (a seed gives the histories of oracle/epv_oracle.c's
// orc_forward_thinning), the same law, the same files.  -P <file> additionally writes the
// histories as local paths, what global_jumps_to_paths would make of the two other files.
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>
#include <stdexcept>

#include "epievo_mi355x.h"
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
    size_t gpu_opt = std::numeric_limits<size_t>::max();   // -g <device>; unset = the host simulators
    string local_paths_file;
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
    opt_parse.add_opt("gpu", 'g', "simulate on this GPU, site-parallel (other random stream than the default)",
                      false, gpu_opt);
    opt_parse.add_opt("local-paths", 'P', "also write the histories as local paths (what global_jumps_to_paths "
                      "makes of the states and jumps files)", false, local_paths_file);
    vector<string> leftover_args;
    opt_parse.parse(argc, argv, leftover_args);
    const int gpu_device = gpu_opt == std::numeric_limits<size_t>::max() ? -1 : (int)gpu_opt;
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
      if (gpu_device < 0) epv::sample_root(the_model, n_sites, gen, root_seq);   // (on the device otherwise)
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
    epv::FlatPaths local;
    if (gpu_device >= 0) {
      if (VERBOSE) cerr << "[GPU MODE: site-parallel thinning on device " << gpu_device << "]" << endl;
      struct Ctx { epv_ctx *c; ~Ctx() { if (c) epv_destroy(c); } } ctx{epv_create(gpu_device)};
      if (!ctx.c) throw std::runtime_error("cannot open HIP device " + std::to_string(gpu_device));
      auto check = [&](int rc, const char *what) {
        if (rc != EPV_OK) throw std::runtime_error(string(what) + ": " + epv_last_error(ctx.c));
      };
      const int N = th.n_nodes();
      check(epv_set_tree(ctx.c, N, th.parent_ids.data(), th.subtree_sizes.data(), th.branches.data()), "epv_set_tree");
      check(epv_set_model(ctx.c, the_model.rates.data(), the_model.T.data()), "epv_set_model");
      uint64_t total = 0;
      uint32_t cap = 16;
      for (;;) {     // histories longer than the slots: the same seed with wider slots gives the same histories
        const int rc = epv_forward_simulate(ctx.c, n_sites, root_seq.empty() ? nullptr : root_seq.data(), rng_seed, cap, &total);
        if (rc == EPV_ERR_CAPACITY && cap < 2047) { cap = std::min<uint32_t>(2047, cap * 2); continue; }
        check(rc, "epv_forward_simulate");
        break;
      }
      const uint64_t n = n_sites, B = (uint64_t)N - 1;
      local.n_sites = n;
      local.n_nodes = N;
      local.init.assign(B * n, 0);
      local.offsets.assign(B * n + 1, 0);
      local.jumps.assign(total ? total : 1, 0.0);
      check(epv_download_paths(ctx.c, local.init.data(), local.offsets.data(), local.jumps.data()), "epv_download_paths");
      local.jumps.resize(total);
      // node states: a child starts in its parent's end state, ends in init ^ parity(jumps)
      sequences.assign(N, vector<uint8_t>(n, 0));
      for (int node = 1; node < N; ++node) {
        if (th.parent_ids[node] == 0)
          for (uint64_t s2 = 0; s2 < n; ++s2) sequences[0][s2] = local.init[(uint64_t)(node - 1) * n + s2];
        for (uint64_t s2 = 0; s2 < n; ++s2) {
          const uint64_t e = (uint64_t)(node - 1) * n + s2;
          sequences[node][s2] = local.init[e] ^ (uint8_t)((local.offsets[e + 1] - local.offsets[e]) & 1u);
        }
      }
      root_seq = sequences[0];
      // events by context = the J statistics of the histories, summed over the branches
      vector<double> J(B * 8), D(B * 8);
      check(epv_get_sufficient_statistics(ctx.c, J.data(), D.data()), "epv_get_sufficient_statistics");
      events.assign(8, 0);
      for (uint64_t b = 0; b < B; ++b) for (int k = 0; k < 8; ++k) events[k] += (size_t)J[b * 8 + k];
      // (the end sites never change, so every event is the middle site of an interior triple)
      if (!pathfile.empty()) {
        paths.assign(N, {});
        for (int node = 1; node < N; ++node) {
          vector<epv::GlobalJump> &gp = paths[node];
          const uint64_t e0 = (uint64_t)(node - 1) * n;
          gp.reserve(local.offsets[e0 + n] - local.offsets[e0]);
          for (uint64_t s2 = 0; s2 < n; ++s2)
            for (uint64_t k = local.offsets[e0 + s2]; k < local.offsets[e0 + s2 + 1]; ++k)
              gp.push_back(epv::GlobalJump{local.jumps[k], (size_t)s2});
          std::sort(gp.begin(), gp.end(), [](const epv::GlobalJump &a, const epv::GlobalJump &b2) {
            return a.timepoint < b2.timepoint || (a.timepoint == b2.timepoint && a.position < b2.position);
          });
        }
      }
    } else if (n_threads > 1) {
      if (VERBOSE) cerr << "[PARALLEL MODE: one generator per branch, up to " << n_threads << " threads]" << endl;
      epv::simulate_tree_parallel(the_model, th, root_seq, rng_seed, (int)n_threads, sequences, paths, events);
    } else {
      epv::simulate_tree(the_model, th, root_seq, gen, sequences, paths, events);
    }
    if (gpu_device < 0 || !pathfile.empty()) epv::write_global_jumps(pathfile, th.node_names, root_seq, paths);
    if (!local_paths_file.empty()) {
      if (gpu_device < 0) {
        vector<vector<uint8_t>> st(sequences);
        local = epv::global_to_local(th, st, paths);
      }
      epv::write_local_paths(local_paths_file, th.node_names, th.n_nodes(), local.n_sites, th.branches.data(),
                             local.init.data(), local.offsets.data(), local.jumps.data());
    }
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
