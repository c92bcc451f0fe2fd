// epv_io.cpp -- see epv_io.hpp
#include "epv_io.hpp"

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

FlatPaths read_local_paths(const std::string &path_file, std::vector<std::string> &node_names,
                           std::vector<double> &tot_times) {
  std::ifstream in(path_file);
  if (!in) throw std::runtime_error("cannot read: " + path_file);
  // node-major staging: the file lists, per node, one row per site
  struct NodeRows {
    std::vector<uint8_t> init;
    std::vector<uint64_t> cnt;
    std::vector<double> jumps;
    double tot_time = 0.0;
    bool have_tt = false;
    std::string tt_text;   // the first row's tot_time token
  };
  std::vector<NodeRows> rows;
  std::string line;
  while (std::getline(in, line)) {
    if (line.size() > 4 && line.compare(0, 4, "NODE") == 0) {
      node_names.push_back(line.substr(line.find(':') + 1));
      rows.emplace_back();
      continue;
    }
    if (rows.empty()) throw std::runtime_error("bad paths file (no NODE line): " + path_file);
    // "site\tinit\ttot_time\tjump\tjump..." ; tokens are whitespace separated
    const char *p = line.c_str();
    char *end = nullptr;
    NodeRows &nr = rows.back();
    // fast path for the common row "<site>\t<0|1>\t<tot_time>\t[jumps...]": digits, one state
    // character, and a tot_time token that repeats the node's first one byte for byte (then its
    // value is known without another strtod)
    const char *q = p;
    while (*q >= '0' && *q <= '9') ++q;
    if (q != p && *q == '\t' && (q[1] == '0' || q[1] == '1') && q[2] == '\t' && nr.have_tt &&
        line.compare((size_t)(q + 3 - p), nr.tt_text.size(), nr.tt_text) == 0 &&
        (q[3 + nr.tt_text.size()] == '\t' || q[3 + nr.tt_text.size()] == '\0')) {
      nr.init.push_back(q[1] == '1');
      p = q + 3 + nr.tt_text.size();
    } else {
      std::strtoull(p, &end, 10);  // site index (ignored, rows are in order)
      if (end == p) continue;      // blank line
      p = end;
      const long is = std::strtol(p, &end, 10);
      p = end;
      while (*p == ' ' || *p == '\t') ++p;
      const double tt = std::strtod(p, &end);
      if (!nr.have_tt) { nr.tot_time = tt; nr.have_tt = true; nr.tt_text.assign(p, (size_t)(end - p)); }
      else if (tt != nr.tot_time)
        throw std::runtime_error("paths of one node disagree on tot_time: " + path_file);
      p = end;
      nr.init.push_back(is != 0);
    }
    uint64_t c = 0;
    while (*p == '\t' || *p == ' ') ++p;
    while (*p) {             // most rows end here: no jumps
      const double v = std::strtod(p, &end);
      if (end == p) break;
      nr.jumps.push_back(v);
      ++c;
      p = end;
      while (*p == '\t' || *p == ' ') ++p;
    }
    nr.cnt.push_back(c);
  }
  if (rows.size() < 2) throw std::runtime_error("bad paths file: " + path_file);
  FlatPaths fp;
  fp.n_nodes = (int)rows.size();
  fp.n_sites = rows[1].init.size();
  tot_times.assign(rows.size(), 0.0);
  const uint64_t B = rows.size() - 1;
  fp.init.reserve(B * fp.n_sites);
  fp.offsets.reserve(B * fp.n_sites + 1);
  for (size_t b = 1; b < rows.size(); ++b) {
    if (rows[b].init.size() != fp.n_sites)
      throw std::runtime_error("nodes have different numbers of sites: " + path_file);
    tot_times[b] = rows[b].tot_time;
    fp.init.insert(fp.init.end(), rows[b].init.begin(), rows[b].init.end());
    uint64_t off = fp.jumps.size();
    for (uint64_t c : rows[b].cnt) { fp.offsets.push_back(off); off += c; }
    fp.jumps.insert(fp.jumps.end(), rows[b].jumps.begin(), rows[b].jumps.end());
  }
  fp.offsets.push_back(fp.jumps.size());
  return fp;
}

// The EM driver rewrites the whole local_paths file every iteration
// (epievo_est_params_histories.cpp:280-283): 4e6 lines at n = 1e6.  Lines are assembled
// by hand into one large buffer per node: the site index by a backwards itoa, the
// per-branch constant "\t<init>\t<tot_time>\t" from two prebuilt strings, and only actual
// jump times (5 % of the lines on tree.nwk) go through printf's %.17g -- which is what
// ostream precision(max_digits10) in the default float format prints (Path.cpp:62-71).
void write_local_paths(const std::string &path_file, const std::vector<std::string> &node_names,
                       int n_nodes, uint64_t n_sites, const double *tot_times,
                       const uint8_t *init, const uint64_t *offsets, const double *jumps) {
  std::FILE *f = std::fopen(path_file.c_str(), "w");
  if (!f) throw std::runtime_error("bad output file: " + path_file);
  std::fprintf(f, "NODE:%s\n", node_names[0].c_str());
  std::vector<char> buf;
  buf.reserve((size_t)64 << 20);
  char num[64];
  for (int b = 1; b < n_nodes; ++b) {
    std::fprintf(f, "NODE:%s\n", node_names[b].c_str());
    std::string tail[2];
    for (int is = 0; is < 2; ++is) {
      std::snprintf(num, sizeof num, "\t%d\t%.17g\t", is, tot_times[b]);
      tail[is] = num;
    }
    buf.clear();
    for (uint64_t s = 0; s < n_sites; ++s) {
      const uint64_t e = (uint64_t)(b - 1) * n_sites + s;
      char *p = num + sizeof num;
      uint64_t v = s;
      do { *--p = (char)('0' + v % 10); v /= 10; } while (v);
      buf.insert(buf.end(), p, num + sizeof num);
      const std::string &t = tail[init[e] ? 1 : 0];
      buf.insert(buf.end(), t.begin(), t.end());
      for (uint64_t j = offsets[e]; j < offsets[e + 1]; ++j) {
        const int k = std::snprintf(num, sizeof num, "%.17g\t", jumps[j]);
        buf.insert(buf.end(), num, num + k);
      }
      buf.push_back('\n');
      if (buf.size() > ((size_t)60 << 20)) {
        std::fwrite(buf.data(), 1, buf.size(), f);
        buf.clear();
      }
    }
    std::fwrite(buf.data(), 1, buf.size(), f);
  }
  if (std::fclose(f) != 0) throw std::runtime_error("error writing: " + path_file);
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
