#include <chrono>
#include <cstdio>
#include <random>
#include "epv_io.hpp"
using namespace std::chrono;
int main(int argc, char **argv) {
  const uint64_t n = 1000000; const int N = 5;
  epv::FlatPaths fp; fp.n_sites = n; fp.n_nodes = N;
  fp.init.assign(4 * n, 0); fp.offsets.assign(4 * n + 1, 0);
  std::mt19937_64 g(1);
  for (uint64_t e = 0; e < 4 * n; ++e) { fp.offsets[e] = fp.jumps.size(); fp.init[e] = g() & 1; if (g() % 20 == 0) fp.jumps.push_back((g() >> 11) * 1e-17); }
  fp.offsets[4 * n] = fp.jumps.size();
  std::vector<std::string> names = {"G", "E", "C", "D", "F"};
  std::vector<double> tt = {0, 0.02, 0.03, 0.06, 0.1};
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = steady_clock::now();
    epv::write_local_paths(argv[1], names, N, n, tt.data(), fp.init.data(), fp.offsets.data(), fp.jumps.data());
    auto t1 = steady_clock::now();
    std::vector<std::string> nm; std::vector<double> t2v;
    epv::FlatPaths q = epv::read_local_paths(argv[1], nm, t2v);
    auto t2 = steady_clock::now();
    std::printf("write %.3f s, read %.3f s (%zu jumps)\n", duration<double>(t1 - t0).count(), duration<double>(t2 - t1).count(), q.jumps.size());
  }
}
