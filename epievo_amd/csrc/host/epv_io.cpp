// epv_io.cpp -- see epv_io.hpp
#include "epv_io.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace epv {

namespace {

struct Node {
  std::string name;
  double len = 0.0;
  bool generated = false;   // the name was made up by name_missing
  std::vector<Node> child;
};

// One Newick sub-expression "(a,b)name:len" -> Node.  Follows the reference's
// conventions (PhyloTree.cpp:124-203): a representation without a comma is a leaf;
// the name runs from after the last ')' to the first ':' after it; a missing length
// is 0.0; the length text is handed to atof.
Node parse_node(const std::string &rep) {
  Node nd;
  const size_t last_paren = rep.find_last_of(')');
  const size_t tail = (last_paren == std::string::npos) ? 0 : last_paren + 1;
  const size_t colon = rep.find(':', tail);
  nd.name = rep.substr(tail, (colon == std::string::npos ? rep.size() : colon) - tail);
  nd.len = (colon == std::string::npos) ? 0.0 : std::atof(rep.c_str() + colon + 1);
  if (rep.find(',') == std::string::npos) return nd;  // leaf
  // split the top-level comma list between the outer parentheses
  const size_t first = (rep[0] == '(') ? 1 : 0;
  const std::string inner = rep.substr(first, last_paren - first);
  int depth = 0;
  size_t start = 0;
  for (size_t i = 0; i <= inner.size(); ++i) {
    if (i == inner.size() || (depth == 0 && inner[i] == ',')) {
      nd.child.push_back(parse_node(inner.substr(start, i - start)));
      start = i + 1;
    } else if (inner[i] == '(') {
      ++depth;
    } else if (inner[i] == ')') {
      --depth;
    }
  }
  return nd;
}

// PhyloTreePreorder.cpp:79-86: the counter is passed BY VALUE into the children, so
// unnamed siblings can share a generated name -- kept for drop-in fidelity.
void name_missing(Node &nd, size_t count) {
  if (nd.name.empty()) { nd.name = "node_" + std::to_string(count++); nd.generated = true; }
  for (Node &c : nd.child) name_missing(c, count);
}

uint32_t flatten(const Node &nd, Tree &t) {
  const size_t me = t.subtree_sizes.size();
  t.subtree_sizes.push_back(1);
  t.branches.push_back(nd.len);
  t.node_names.push_back(nd.name);
  t.name_generated.push_back(nd.generated);
  for (const Node &c : nd.child) t.subtree_sizes[me] += flatten(c, t);
  return t.subtree_sizes[me];
}

void newick_of(const Tree &t, int node, std::ostringstream &oss) {
  if (t.subtree_sizes[node] > 1) {
    oss << '(';
    bool first = true;
    for (uint32_t c = 1; c < t.subtree_sizes[node]; c += t.subtree_sizes[node + c]) {
      if (!first) oss << ',';
      first = false;
      newick_of(t, node + (int)c, oss);
    }
    oss << ')';
  }
  // each reference node formats through its own fresh ostringstream (default precision)
  std::ostringstream num;
  num << t.branches[node];
  // the mains print the tree they READ (names are generated on TreeHelper's private copy,
  // TreeHelper.cpp:43-46), so a node without a name in the input stays without one
  const bool gen = (size_t)node < t.name_generated.size() && t.name_generated[node];
  oss << (gen ? std::string() : t.node_names[node]) << ':' << num.str();
}

}  // namespace

Tree Tree::parse(const std::string &newick_in) {
  std::string rep;
  int balance = 0;
  for (char c : newick_in) {
    if (std::isspace((unsigned char)c)) continue;
    if (c == '(') ++balance;
    if (c == ')') --balance;
    rep.push_back(c);
  }
  if (balance != 0) throw std::runtime_error("Unbalanced parentheses in Newick format: " + newick_in);
  if (rep.empty()) throw std::runtime_error("bad tree format");
  if (rep.back() == ';') rep.pop_back();
  Node root = parse_node(rep);
  name_missing(root, 0);
  Tree t;
  flatten(root, t);
  t.parent_ids.assign(t.subtree_sizes.size(), 0);
  for (size_t i = 0; i < t.subtree_sizes.size(); ++i)
    for (uint32_t c = 1; c < t.subtree_sizes[i]; c += t.subtree_sizes[i + c])
      t.parent_ids[i + c] = (uint32_t)i;
  return t;
}

Tree Tree::read(const std::string &tree_file) {
  std::ifstream in(tree_file);
  if (!in) throw std::runtime_error("bad tree file: " + tree_file);
  std::string rep;
  char c;
  bool found_end = false;
  while (!found_end && in >> c) {
    rep += c;
    if (c == ';') found_end = true;
  }
  if (!found_end) throw std::runtime_error("bad tree file: " + tree_file);
  return parse(rep);
}

Tree Tree::single_branch(double evo_time) {
  Tree t;
  t.subtree_sizes = {2, 1};
  t.node_names = {"root", "leaf"};
  t.parent_ids = {0, 0};
  t.branches = {0.0, evo_time};
  return t;
}

std::string Tree::newick() const {
  std::ostringstream oss;
  newick_of(*this, 0, oss);
  oss << ';';
  return oss.str();
}

// ---- local_paths files are the EM driver's bulk IO: 4e6 rows (111 MB) at n = 1e6 on tree.nwk, read
// once and REWRITTEN EVERY ITERATION (epievo_est_params_histories.cpp:280-283).  Both directions
// run on all host cores: the file is cut at row boundaries, the pieces are parsed / formatted
// independently and land at their own offsets (pread / pwrite), byte for byte the rows the
// sequential code produced.  EPV_IO_THREADS overrides the thread count (1 = sequential).
namespace {

unsigned io_threads(uint64_t rows) {
  unsigned t = std::thread::hardware_concurrency();
  if (t == 0) t = 1;
  if (t > 16) t = 16;
  if (const char *e = std::getenv("EPV_IO_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 256) t = (unsigned)v; }
  if (rows < 65536) t = 1;
  return t;
}

template <class F>
void parallel_jobs(unsigned n_jobs, unsigned n_threads, F &&job) {
  if (n_threads <= 1 || n_jobs <= 1) { for (unsigned j = 0; j < n_jobs; ++j) job(j); return; }
  std::vector<std::string> errors(n_jobs);
  std::atomic<unsigned> next(0);
  auto worker = [&] {
    for (;;) {
      const unsigned j = next.fetch_add(1);
      if (j >= n_jobs) return;
      try { job(j); } catch (const std::exception &e) { errors[j] = e.what(); }
    }
  };
  std::vector<std::thread> th;
  struct Join { std::vector<std::thread> &t; ~Join() { for (auto &x : t) if (x.joinable()) x.join(); } } join{th};
  for (unsigned i = 0; i + 1 < std::min(n_threads, n_jobs); ++i) th.emplace_back(worker);
  worker();
  for (auto &x : th) x.join();
  for (const std::string &e : errors) if (!e.empty()) throw std::runtime_error(e);
}

struct RowChunk {           // what one piece of a node's rows parses to
  std::vector<uint8_t> init;
  std::vector<uint64_t> cnt;
  std::vector<double> jumps;
};

// rows [p, end) of one node ('\n' separated, the buffer is NUL terminated behind its last byte);
// tt_text / tot_time: the node's tot_time token and value, known from its first row
void parse_rows(const char *p, const char *end, const std::string &tt_text, double tot_time, const std::string &path_file,
                RowChunk &out) {
  while (p < end) {
    const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
    if (!eol) eol = end;
    char *stop = nullptr;
    // fast path for the common row "<site>\t<0|1>\t<tot_time>\t[jumps...]": digits, one state
    // character, and a tot_time token that repeats the node's first one byte for byte (then its
    // value is known without another strtod)
    const char *q = p;
    while (*q >= '0' && *q <= '9') ++q;
    bool have_row = true;
    if (q != p && *q == '\t' && (q[1] == '0' || q[1] == '1') && q[2] == '\t' &&
        (size_t)(eol - (q + 3)) >= tt_text.size() && std::memcmp(q + 3, tt_text.data(), tt_text.size()) == 0 &&
        (q[3 + tt_text.size()] == '\t' || q + 3 + tt_text.size() == eol)) {
      out.init.push_back(q[1] == '1');
      p = q + 3 + tt_text.size();
    } else {
      std::strtoull(p, &stop, 10);  // site index (ignored, rows are in order)
      if (stop == p || stop > eol) { have_row = false; }      // blank line
      else {
        p = stop;
        const long is = std::strtol(p, &stop, 10);
        p = stop;
        while (*p == ' ' || *p == '\t') ++p;
        const double tt = std::strtod(p, &stop);
        if (tt != tot_time) throw std::runtime_error("paths of one node disagree on tot_time: " + path_file);
        p = stop;
        out.init.push_back(is != 0);
      }
    }
    if (have_row) {
      uint64_t c = 0;
      while (p < eol && (*p == '\t' || *p == ' ')) ++p;
      while (p < eol) {             // most rows end here: no jumps
        const double v = std::strtod(p, &stop);
        if (stop == p) break;
        out.jumps.push_back(v);
        ++c;
        p = stop;
        while (p < eol && (*p == '\t' || *p == ' ')) ++p;
      }
      out.cnt.push_back(c);
    }
    p = eol + 1;
  }
}

}  // namespace

FlatPaths read_local_paths(const std::string &path_file, std::vector<std::string> &node_names,
                           std::vector<double> &tot_times) {
  const int fd = ::open(path_file.c_str(), O_RDONLY);
  if (fd < 0) throw std::runtime_error("cannot read: " + path_file);
  struct Fd { int fd; ~Fd() { ::close(fd); } } guard{fd};
  struct stat st;
  if (::fstat(fd, &st) != 0) throw std::runtime_error("cannot read: " + path_file);
  const uint64_t size = (uint64_t)st.st_size;
  std::vector<char> text(size + 1);
  const unsigned T = io_threads(size / 32);
  {
    const unsigned pieces = std::max(1u, T);
    parallel_jobs(pieces, T, [&](unsigned j) {
      uint64_t lo = size * j / pieces, hi = size * (j + 1) / pieces;
      while (lo < hi) {
        const ssize_t k = ::pread(fd, text.data() + lo, (size_t)std::min<uint64_t>(hi - lo, (uint64_t)1 << 30), (off_t)lo);
        if (k <= 0) throw std::runtime_error("error reading: " + path_file);
        lo += (uint64_t)k;
      }
    });
  }
  text[size] = '\0';
  const char *base = text.data(), *fin = base + size;
  // the NODE lines: at the start of the file, and behind a newline
  std::vector<const char *> heads;
  {
    const unsigned pieces = std::max(1u, T);
    std::vector<std::vector<const char *>> found(pieces);
    parallel_jobs(pieces, T, [&](unsigned j) {
      const char *lo = base + size * j / pieces, *hi = base + size * (j + 1) / pieces;
      for (const char *p = lo; p < hi;) {
        const char *q = (const char *)std::memchr(p, 'N', (size_t)(hi - p));
        if (!q) break;
        if ((q == base || q[-1] == '\n') && fin - q > 4 && std::memcmp(q, "NODE", 4) == 0) found[j].push_back(q);
        p = q + 1;
      }
    });
    for (auto &v : found) heads.insert(heads.end(), v.begin(), v.end());
  }
  if (heads.empty()) throw std::runtime_error("bad paths file (no NODE line): " + path_file);
  {
    // anything before the first NODE line must be blank
    for (const char *p = base; p < heads[0]; ++p)
      if (*p != '\n' && *p != ' ' && *p != '\t' && *p != '\r')
        throw std::runtime_error("bad paths file (no NODE line): " + path_file);
  }
  const size_t N = heads.size();
  if (N < 2) throw std::runtime_error("bad paths file: " + path_file);
  struct NodeBlock { const char *rows, *end; std::string tt_text; double tot_time = 0.0; };
  std::vector<NodeBlock> nb(N);
  for (size_t b = 0; b < N; ++b) {
    const char *eol = (const char *)std::memchr(heads[b], '\n', (size_t)(fin - heads[b]));
    if (!eol) eol = fin;
    const char *colon = (const char *)std::memchr(heads[b], ':', (size_t)(eol - heads[b]));
    node_names.push_back(colon ? std::string(colon + 1, eol) : std::string(heads[b], eol));
    nb[b].rows = eol < fin ? eol + 1 : fin;
    nb[b].end = b + 1 < N ? heads[b + 1] : fin;
    // the node's tot_time token: third field of its first row
    const char *p = nb[b].rows;
    while (p < nb[b].end && (*p == '\n' || *p == '\r')) ++p;
    if (p < nb[b].end) {
      char *stop = nullptr;
      std::strtoull(p, &stop, 10);
      if (stop != p) {
        p = stop;
        std::strtol(p, &stop, 10);
        p = stop;
        while (*p == ' ' || *p == '\t') ++p;
        nb[b].tot_time = std::strtod(p, &stop);
        nb[b].tt_text.assign(p, (size_t)(stop - p));
      }
    }
  }
  // every node's rows in `pieces` chunks cut at row boundaries
  const unsigned pieces = std::max(1u, T);
  std::vector<RowChunk> chunks(N * pieces);
  parallel_jobs((unsigned)(N * pieces), T, [&](unsigned job) {
    const size_t b = job / pieces, j = job % pieces;
    const char *lo = nb[b].rows, *hi = nb[b].end;
    const uint64_t len = (uint64_t)(hi - lo);
    const char *a = lo + len * j / pieces, *z = lo + len * (j + 1) / pieces;
    auto align = [&](const char *p) {     // first row start at or behind p
      if (p <= lo) return lo;
      if (p >= hi) return hi;
      const char *q = (const char *)std::memchr(p - 1, '\n', (size_t)(hi - (p - 1)));
      return q ? q + 1 : hi;
    };
    a = align(a);
    z = align(z);
    if (a < z) parse_rows(a, z, nb[b].tt_text, nb[b].tot_time, path_file, chunks[job]);
  });
  FlatPaths fp;
  fp.n_nodes = (int)N;
  tot_times.assign(N, 0.0);
  const uint64_t B = N - 1;
  std::vector<uint64_t> rows_of(N, 0), jumps_of(N, 0);
  for (size_t b = 0; b < N; ++b)
    for (unsigned j = 0; j < pieces; ++j) { rows_of[b] += chunks[b * pieces + j].init.size(); jumps_of[b] += chunks[b * pieces + j].jumps.size(); }
  fp.n_sites = rows_of[1];
  uint64_t total_jumps = 0;
  for (size_t b = 1; b < N; ++b) {
    if (rows_of[b] != fp.n_sites) throw std::runtime_error("nodes have different numbers of sites: " + path_file);
    tot_times[b] = nb[b].tot_time;
    total_jumps += jumps_of[b];
  }
  fp.init.resize(B * fp.n_sites);
  fp.offsets.resize(B * fp.n_sites + 1);
  fp.jumps.resize(total_jumps);
  // where every chunk lands
  std::vector<uint64_t> row0(N * pieces, 0), jmp0(N * pieces, 0);
  {
    uint64_t r = 0, jj = 0;
    for (size_t b = 1; b < N; ++b)
      for (unsigned j = 0; j < pieces; ++j) {
        row0[b * pieces + j] = r; jmp0[b * pieces + j] = jj;
        r += chunks[b * pieces + j].init.size(); jj += chunks[b * pieces + j].jumps.size();
      }
  }
  parallel_jobs((unsigned)(N * pieces), T, [&](unsigned job) {
    if (job < pieces) return;     // the root's block has no rows
    const RowChunk &c = chunks[job];
    std::copy(c.init.begin(), c.init.end(), fp.init.begin() + row0[job]);
    std::copy(c.jumps.begin(), c.jumps.end(), fp.jumps.begin() + jmp0[job]);
    uint64_t off = jmp0[job];
    for (size_t i = 0; i < c.cnt.size(); ++i) { fp.offsets[row0[job] + i] = off; off += c.cnt[i]; }
  });
  fp.offsets[B * fp.n_sites] = total_jumps;
  return fp;
}

// The EM driver rewrites the whole local_paths file every iteration
// (epievo_est_params_histories.cpp:280-283): 4e6 lines at n = 1e6.  Lines are assembled
// by hand: the site index by a backwards itoa, the per-branch constant "\t<init>\t<tot_time>\t"
// from two prebuilt strings, and only actual jump times (5 % of the lines on tree.nwk) go through
// printf's %.17g -- which is what ostream precision(max_digits10) in the default float format
// prints (Path.cpp:62-71).  A node's rows are formatted in pieces on all cores and written at their
// offsets with pwrite.
void write_local_paths(const std::string &path_file, const std::vector<std::string> &node_names,
                       int n_nodes, uint64_t n_sites, const double *tot_times,
                       const uint8_t *init, const uint64_t *offsets, const double *jumps) {
  const int fd = ::open(path_file.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) throw std::runtime_error("bad output file: " + path_file);
  struct Fd { int fd; ~Fd() { if (fd >= 0) ::close(fd); } } guard{fd};
  auto put = [&](const char *p, uint64_t len, uint64_t at) {
    while (len) {
      const ssize_t k = ::pwrite(fd, p, (size_t)std::min<uint64_t>(len, (uint64_t)1 << 30), (off_t)at);
      if (k <= 0) throw std::runtime_error("error writing: " + path_file);
      p += k; len -= (uint64_t)k; at += (uint64_t)k;
    }
  };
  uint64_t at = 0;
  {
    const std::string head = "NODE:" + node_names[0] + "\n";
    put(head.data(), head.size(), at);
    at += head.size();
  }
  const unsigned T = io_threads(n_sites);
  const unsigned pieces = T;
  std::vector<std::vector<char>> buf(pieces);
  for (int b = 1; b < n_nodes; ++b) {
    const std::string head = "NODE:" + node_names[b] + "\n";
    put(head.data(), head.size(), at);
    at += head.size();
    char num0[64];
    std::string tail[2];
    for (int is = 0; is < 2; ++is) {
      std::snprintf(num0, sizeof num0, "\t%d\t%.17g\t", is, tot_times[b]);
      tail[is] = num0;
    }
    parallel_jobs(pieces, T, [&](unsigned j) {
      std::vector<char> &out = buf[j];
      out.clear();
      const uint64_t s_lo = n_sites * j / pieces, s_hi = n_sites * (j + 1) / pieces;
      const uint64_t e0 = (uint64_t)(b - 1) * n_sites;
      const uint64_t nj = offsets[e0 + s_hi] - offsets[e0 + s_lo];
      out.resize((s_hi - s_lo) * (22 + tail[0].size()) + nj * 26 + 16);   // upper bound: 20 digits, tail, '\n'; 25 chars per jump
      char *w = out.data();
      char num[64];
      for (uint64_t s = s_lo; s < s_hi; ++s) {
        const uint64_t e = e0 + s;
        char *p = num + sizeof num;
        uint64_t v = s;
        do { *--p = (char)('0' + v % 10); v /= 10; } while (v);
        const size_t nd = (size_t)(num + sizeof num - p);
        std::memcpy(w, p, nd);
        w += nd;
        const std::string &t = tail[init[e] ? 1 : 0];
        std::memcpy(w, t.data(), t.size());
        w += t.size();
        for (uint64_t q = offsets[e]; q < offsets[e + 1]; ++q) w += std::snprintf(w, 32, "%.17g\t", jumps[q]);
        *w++ = '\n';
      }
      out.resize((size_t)(w - out.data()));
    });
    std::vector<uint64_t> pos(pieces + 1, at);
    for (unsigned j = 0; j < pieces; ++j) pos[j + 1] = pos[j] + buf[j].size();
    parallel_jobs(pieces, T, [&](unsigned j) { put(buf[j].data(), buf[j].size(), pos[j]); });
    at = pos[pieces];
  }
  const int rc = ::close(fd);
  guard.fd = -1;
  if (rc != 0) throw std::runtime_error("error writing: " + path_file);
}

void read_states_file(const std::string &states_file, std::vector<std::string> &names,
                      std::vector<std::vector<uint8_t>> &states) {
  std::ifstream in(states_file);
  if (!in) throw std::runtime_error("cannot read states file: " + states_file);
  std::string line;
  std::getline(in, line);
  if (!line.empty() && line[0] == '#') line = line.substr(1);
  std::istringstream hs(line);
  std::string nm;
  while (hs >> nm) names.push_back(nm);
  states.assign(names.size(), {});
  while (std::getline(in, line)) {
    std::istringstream ls(line);
    size_t site = 0;
    ls >> site;
    size_t k = 0;
    char v = 0;
    while (k < names.size() && ls >> v) states[k++].push_back(v == '1');
    if (k != names.size()) throw std::runtime_error("bad line in states file");
  }
}

}  // namespace epv
