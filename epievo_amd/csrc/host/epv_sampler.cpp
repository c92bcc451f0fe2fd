// epv_sampler.cpp -- see epv_sampler.hpp
#include "epv_sampler.hpp"

#include <algorithm>
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>

#include "epievo_mi355x.h"
#include "epievo_mi355x_comm.h"

namespace epv {

namespace {
const uint64_t kBlock = 256;   // sites per level-0 block of the statistics tree
const uint32_t kMaxCap = 2047;   // EPV_MAX_CAP: 2 C + 1 segments must fit the 12-bit segment field of the Philox address

// worker threads that are joined on every exit path: if starting thread i + 1 throws
// (std::system_error on thread exhaustion), the i threads already running are joined before the
// exception leaves the scope instead of taking the process down through std::terminate
struct ThreadGroup {
  std::vector<std::thread> th;
  template <class F> void spawn(F &&f) { th.emplace_back(std::forward<F>(f)); }
  void join() { for (std::thread &t : th) if (t.joinable()) t.join(); }
  ~ThreadGroup() { join(); }
};

uint64_t round_to(double x, uint64_t unit) { return (uint64_t)(x / (double)unit + 0.5) * unit; }

// sites [lo, hi) of node-major flat paths
FlatPaths slice_sites(const FlatPaths &p, uint64_t lo, uint64_t hi) {
  FlatPaths q;
  q.n_sites = hi - lo;
  q.n_nodes = p.n_nodes;
  const uint64_t B = (uint64_t)p.n_nodes - 1, n = p.n_sites, m = hi - lo;
  q.init.resize(B * m);
  q.offsets.assign(B * m + 1, 0);
  for (uint64_t b = 0; b < B; ++b) {
    std::copy(p.init.begin() + b * n + lo, p.init.begin() + b * n + hi, q.init.begin() + b * m);
    const uint64_t j0 = p.offsets[b * n + lo], j1 = p.offsets[b * n + hi];
    for (uint64_t s = 0; s < m; ++s)
      q.offsets[b * m + s] = q.jumps.size() + (p.offsets[b * n + lo + s] - j0);
    q.jumps.insert(q.jumps.end(), p.jumps.begin() + j0, p.jumps.begin() + j1);
  }
  q.offsets[B * m] = q.jumps.size();
  return q;
}

// columns [lo, hi) of a genome of which `p` holds [first, first + p.n_sites): columns outside
// what is held come out blank (state 0, no jump) -- the halo columns of a slot whose neighbours
// live in other processes, filled by the first halo exchange
FlatPaths slice_sites_padded(const FlatPaths &p, uint64_t first, uint64_t lo, uint64_t hi) {
  if (lo >= first && hi <= first + p.n_sites) return slice_sites(p, lo - first, hi - first);
  const uint64_t a = std::max(lo, first), b = std::min(hi, first + p.n_sites);
  const FlatPaths mid = slice_sites(p, a - first, b - first);
  FlatPaths q;
  q.n_sites = hi - lo;
  q.n_nodes = p.n_nodes;
  const uint64_t B = (uint64_t)p.n_nodes - 1, m = hi - lo, mm = mid.n_sites, pad = a - lo;
  q.init.assign(B * m, 0);
  q.offsets.assign(B * m + 1, 0);
  q.jumps = mid.jumps;
  for (uint64_t br = 0; br < B; ++br) {
    std::copy(mid.init.begin() + br * mm, mid.init.begin() + (br + 1) * mm, q.init.begin() + br * m + pad);
    for (uint64_t s2 = 0; s2 < m; ++s2) {
      const uint64_t k = s2 < pad ? 0 : (s2 - pad < mm ? s2 - pad : mm);
      q.offsets[br * m + s2] = mid.offsets[br * mm + k];
    }
  }
  q.offsets[B * m] = mid.offsets[B * mm];
  return q;
}

FlatPaths concat_sites(const std::vector<FlatPaths> &parts) {
  FlatPaths q;
  q.n_nodes = parts[0].n_nodes;
  for (const FlatPaths &p : parts) q.n_sites += p.n_sites;
  const uint64_t B = (uint64_t)q.n_nodes - 1, n = q.n_sites;
  q.init.resize(B * n);
  q.offsets.assign(B * n + 1, 0);
  for (uint64_t b = 0; b < B; ++b) {
    uint64_t at = 0;
    for (const FlatPaths &p : parts) {
      const uint64_t m = p.n_sites;
      std::copy(p.init.begin() + b * m, p.init.begin() + (b + 1) * m, q.init.begin() + b * n + at);
      const uint64_t j0 = p.offsets[b * m], j1 = p.offsets[(b + 1) * m];
      for (uint64_t s = 0; s < m; ++s) q.offsets[b * n + at + s] = q.jumps.size() + (p.offsets[b * m + s] - j0);
      q.jumps.insert(q.jumps.end(), p.jumps.begin() + j0, p.jumps.begin() + j1);
      at += m;
    }
  }
  q.offsets[B * n] = q.jumps.size();
  return q;
}
}  // namespace

std::vector<int> parse_device_list(const std::string &spec) {
  std::vector<int> out;
  if (spec.empty()) return {0};
  if (spec == "all") {
    // every GPU HIP shows; counted by probing contexts so that this file needs no HIP header
    for (int d = 0; d < 64; ++d) {
      epv_ctx *c = epv_create(d);
      if (!c) break;
      epv_destroy(c);
      out.push_back(d);
    }
    if (out.empty()) throw std::runtime_error("no HIP device found (this build has no CPU fallback)");
    return out;
  }
  std::stringstream ss(spec);
  std::string tok;
  while (std::getline(ss, tok, ',')) {
    char *end = nullptr;
    const long v = std::strtol(tok.c_str(), &end, 10);
    if (tok.empty() || *end != '\0' || v < 0) throw std::runtime_error("bad device list: " + spec);
    out.push_back((int)v);
  }
  if (out.empty()) throw std::runtime_error("bad device list: " + spec);
  return out;
}

std::vector<int> devices_from_env() {
  const char *e = std::getenv("EPV_DEVICES");
  return parse_device_list(e ? e : "");
}

SingleSiteSampler::SingleSiteSampler(size_t n_burn_in, size_t n_batch, int device, uint32_t capacity)
    : SingleSiteSampler(n_burn_in, n_batch, std::vector<int>(1, device), capacity) {}

SingleSiteSampler::SingleSiteSampler(size_t n_burn_in, size_t n_batch, const std::vector<int> &devices,
                                     uint32_t capacity)
    : SAMPLE_ROOT(false), burn_in(n_burn_in), batch(n_batch), ctx_(nullptr), devices_(devices),
      capacity_(capacity) {
  if (devices_.empty()) devices_.push_back(0);
  ctx_ = epv_create(devices_[0]);
  if (!ctx_)
    throw std::runtime_error("cannot open HIP device " + std::to_string(devices_[0]) +
                             " (this build has no CPU fallback)");
  if (const char *e = std::getenv("EPV_CONTEXTS_PER_GPU")) contexts_wanted_ = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("EPV_ROW_BLOCKS")) {
    const int v = std::atoi(e);
    if (v < 1 || v > 4096 || (v & (v - 1))) throw std::runtime_error("EPV_ROW_BLOCKS must be a power of two");
    row_blocks_ = (uint32_t)v;
  }
  if (const char *e = std::getenv("EPV_FORCE_COMM")) force_comm_ = std::atoi(e) != 0;
}

SingleSiteSampler::SingleSiteSampler(size_t n_burn_in, size_t n_batch, const RankSpec &rank, uint32_t capacity)
    : SingleSiteSampler(n_burn_in, n_batch, std::vector<int>(1, rank.device), capacity) {
  if (rank.world < 1 || rank.rank < 0 || rank.rank >= rank.world) throw std::runtime_error("bad rank / world");
  rank_mode_ = true;
  rank_ = rank;
  world_ = (size_t)rank.world;
}

SingleSiteSampler::~SingleSiteSampler() {
  drop_parts();
  if (rank_comm_) epv_comm_destroy(rank_comm_);
  epv_destroy(ctx_);
}

std::vector<epv_ctx *> SingleSiteSampler::contexts() const {
  std::vector<epv_ctx *> v;
  if (parts_.empty()) v.push_back(ctx_);
  for (const Part &p : parts_) v.push_back(p.ctx);
  return v;
}
void SingleSiteSampler::set_options(uint32_t flags) {
  SAMPLE_ROOT = (flags & (uint32_t)EPV_OPT_SAMPLE_ROOT) != 0;
  for (epv_ctx *c : contexts()) check_on(c, epv_set_options(c, flags), "epv_set_options");
}
void SingleSiteSampler::set_timing(int every) {
  for (epv_ctx *c : contexts()) check_on(c, epv_set_timing(c, every), "epv_set_timing");
}
void SingleSiteSampler::kernel_time_ms(double &avg_ms, uint64_t &n_launches) {
  double tot = 0.0;
  n_launches = 0;
  for (epv_ctx *c : contexts()) {
    double a = 0.0;
    uint64_t k = 0;
    check_on(c, epv_kernel_time_ms(c, &a, &k), "epv_kernel_time_ms");
    tot += a * (double)k;
    n_launches += k;
  }
  avg_ms = n_launches ? tot / (double)n_launches : 0.0;
}
uint32_t SingleSiteSampler::phase_mode() {
  uint32_t m = 0;
  epv_ctx *c = contexts()[0];
  check_on(c, epv_phase_mode(c, &m), "epv_phase_mode");
  return m;
}

void SingleSiteSampler::free_stat_buffers() {
  for (Slot &s : slots_) {
    epv_ctx *c = parts_[s.part0].ctx;
    if (s.d_blocks) epv_dev_free(c, s.d_blocks);
    if (s.d_rows) epv_dev_free(c, s.d_rows);
    if (s.d_gather) epv_dev_free(c, s.d_gather);
    s.d_blocks = s.d_rows = s.d_gather = nullptr;
  }
  stat_batch_ = 0;
}

void SingleSiteSampler::drop_parts() {
  free_stat_buffers();
  for (Slot &s : slots_) {
    epv_ctx *c = parts_[s.part0].ctx;
    for (void *&p : s.d_halo) { if (p) epv_dev_free(c, p); p = nullptr; }
    if (s.comm && s.comm != rank_comm_) epv_comm_destroy(s.comm);
    s.comm = nullptr;
  }
  for (Part &p : parts_) if (p.ctx != ctx_) epv_destroy(p.ctx);
  parts_.clear();
  slots_.clear();
}

bool SingleSiteSampler::uses_rccl() const {
  return !slots_.empty() && slots_[0].comm && epv_comm_is_rccl(slots_[0].comm);
}

std::string SingleSiteSampler::layout() const {
  std::ostringstream o;
  if (!sharded()) { o << "1 context on device " << devices_[0]; return o.str(); }
  o << world_ << " GPU slot(s)";
  if (rank_mode_) o << " (this process: slot " << rank_.rank << ")";
  o << " x up to " << contexts_per_gpu() << " context(s) = " << parts_.size() << " parts"
    << (rank_mode_ ? " here" : "") << ", halo " << halo_ << " columns";
  if (slots_[0].comm)
    o << ", statistics rows of " << kBlock * row_blocks_ << " sites, exchange over "
      << (uses_rccl() ? "RCCL" : "the loopback transport");
  o << "; devices";
  for (const Slot &s : slots_) o << " " << s.device;
  return o.str();
}

void SingleSiteSampler::check_on(epv_ctx *c, int rc, const char *what) {
  if (rc != EPV_OK) throw std::runtime_error(std::string(what) + ": " + epv_last_error(c));
}
void SingleSiteSampler::check_comm(epv_comm *c, int rc, const char *what) {
  if (rc != EPV_OK) throw std::runtime_error(std::string(what) + ": " + epv_comm_last_error(c));
}
void SingleSiteSampler::check(int rc, const char *what) { check_on(ctx_, rc, what); }

// after an MCMC call: an overflow leaves a valid chain and complete outputs (the over-long
// proposals were rejected), so widen the jump slots for the following calls -- what the
// reference's std::vector paths do on their own -- and carry on
void SingleSiteSampler::check_mcmc(int rc, const char *what) {
  if (rc == EPV_ERR_CAPACITY) {
    uint32_t cap = 0;
    if (epv_get_capacity(ctx_, &cap) == EPV_OK && cap < kMaxCap) {
      const std::string msg = epv_last_error(ctx_);
      check(epv_set_capacity(ctx_, std::min(kMaxCap, cap * 2u)), "epv_set_capacity");
      capacity_events.push_back(std::string(what) + ": " + msg);
      return;
    }
  }
  check(rc, what);
}

void SingleSiteSampler::equalize_capacity() {
  uint32_t cap = 0;
  for (Part &p : parts_) { uint32_t v = 0; check_on(p.ctx, epv_get_capacity(p.ctx, &v), "epv_get_capacity"); cap = std::max(cap, v); }
  for (Part &p : parts_) check_on(p.ctx, epv_set_capacity(p.ctx, cap), "epv_set_capacity");
}

// Before every reset of a sharded genome: equal jump-slot widths (an overflow may have widened
// one part), each part's edge columns into its neighbour's halo -- a device-to-device copy
// inside a GPU, one RCCL send/receive pair per GPU boundary -- and the halos marked fresh.
void SingleSiteSampler::refresh_parts() {
  const size_t P = parts_.size();
  const uint64_t H = halo_;
  equalize_capacity();
  const uint64_t bytes = H * epv_column_bytes(parts_[0].ctx);
  // between two parts of one GPU: device-to-device copies
  for (size_t p = 0; p + 1 < P; ++p) {
    Part &L = parts_[p], &R = parts_[p + 1];
    if (L.slot != R.slot) continue;
    // (no host wait: the unpack sits on the receiving context's stream, in front of its reset;
    // a part's right-going edge uses half 0 of its staging buffer, its left-going edge half 1)
    check_on(R.ctx, epv_copy_columns_async(L.ctx, L.b - H - L.lo, H, R.ctx, 0, 0), "epv_copy_columns_async");
    check_on(L.ctx, epv_copy_columns_async(R.ctx, H, H, L.ctx, L.b - L.lo, 1), "epv_copy_columns_async");
  }
  // between two GPU slots: the first part of a slot sends its left edge to the slot before it, the
  // last part its right edge to the slot after it -- one exchange call per slot that lives here
  // (all of them, or this process's one), the neighbours' calls pair up inside RCCL
  if (world_ > 1) {
    for (Slot &s : slots_) {
      epv_ctx *c = parts_[s.part0].ctx;
      if (s.halo_bytes != bytes) {
        for (void *&q : s.d_halo) {
          if (q) check_on(c, epv_dev_free(c, q), "epv_dev_free");
          q = nullptr;
          check_on(c, epv_dev_alloc(c, bytes, &q), "epv_dev_alloc");
        }
        s.halo_bytes = bytes;
      }
      Part &F = parts_[s.part0], &Lp = parts_[s.part1 - 1];
      if (s.gidx > 0) check_on(F.ctx, epv_pack_columns_dev(F.ctx, F.a - F.lo, H, s.d_halo[0]), "epv_pack_columns_dev");
      if (s.gidx + 1 < world_)
        check_on(Lp.ctx, epv_pack_columns_dev(Lp.ctx, Lp.b - H - Lp.lo, H, s.d_halo[2]), "epv_pack_columns_dev");
    }
    if (epv_comm_group_start() != EPV_OK) throw std::runtime_error("epv_comm_group_start failed");
    for (Slot &s : slots_)
      check_comm(s.comm, epv_comm_exchange(s.comm, s.d_halo[0], s.d_halo[1], s.gidx > 0 ? bytes : 0, s.d_halo[2],
                                           s.d_halo[3], s.gidx + 1 < world_ ? bytes : 0), "epv_comm_exchange");
    if (epv_comm_group_end() != EPV_OK) throw std::runtime_error("epv_comm_group_end failed (halo exchange)");
    for (Slot &s : slots_) check_comm(s.comm, epv_comm_sync(s.comm), "epv_comm_sync");
    for (Slot &s : slots_) {
      Part &F = parts_[s.part0], &Lp = parts_[s.part1 - 1];
      if (s.gidx > 0) check_on(F.ctx, epv_unpack_columns_dev(F.ctx, 0, H, s.d_halo[1]), "epv_unpack_columns_dev");
      if (s.gidx + 1 < world_)
        check_on(Lp.ctx, epv_unpack_columns_dev(Lp.ctx, Lp.b - Lp.lo, H, s.d_halo[3]), "epv_unpack_columns_dev");
    }
  }
  for (size_t p = 0; p < P; ++p) {
    const bool first = parts_[p].lo == parts_[p].a, last = parts_[p].hi == parts_[p].b;   // the genome's ends
    check_on(parts_[p].ctx, epv_set_halo(parts_[p].ctx, first ? 0 : H, last ? 0 : H), "epv_set_halo");
  }
}

// SAMPLE_ROOT (a public field of the reference class, hard-wired false at SingleSiteSampler.cpp:441 and
// set by none of its programs): the proposal also draws the root state (:167-176, :246-249).  The
// field is forwarded to every context as EPV_OPT_SAMPLE_ROOT before each call that runs updates.
void SingleSiteSampler::apply_sample_root() {
  for (epv_ctx *c : contexts()) {
    uint32_t flags = 0;
    check_on(c, epv_get_options(c, &flags), "epv_get_options");
    const uint32_t want = SAMPLE_ROOT ? (flags | (uint32_t)EPV_OPT_SAMPLE_ROOT) : (flags & ~(uint32_t)EPV_OPT_SAMPLE_ROOT);
    if (want != flags) check_on(c, epv_set_options(c, want), "epv_set_options");
  }
}

// a halo that lasts one whole run_mcmc (two columns per colour phase), in whole blocks
static uint64_t halo_for(size_t n_burn_in, size_t n_batch) {
  return std::max<uint64_t>(kBlock, (6 * (uint64_t)(n_burn_in + n_batch) + 2 + kBlock - 1) / kBlock * kBlock);
}

std::vector<uint64_t> SingleSiteSampler::shard_cuts(uint64_t n, size_t world, size_t n_burn_in, size_t n_batch,
                                                    uint32_t row_blocks) {
  const uint64_t H = halo_for(n_burn_in, n_batch);
  const uint64_t min_part = 2 * H + 2 * kBlock;     // every part must own more than its halos
  const uint64_t RS = kBlock * row_blocks;
  size_t G = std::max<size_t>(1, world);
  std::vector<uint64_t> cut;
  for (;; --G) {
    cut.assign(G + 1, 0);
    cut[G] = n;
    bool ok = true;
    for (size_t g = 1; g < G; ++g) cut[g] = round_to((double)g * (double)n / (double)G, RS);
    for (size_t g = 0; g < G && ok; ++g) ok = cut[g + 1] > cut[g] && cut[g + 1] - cut[g] >= min_part;
    if (ok || G == 1) break;
  }
  return cut;
}

void SingleSiteSampler::reset(const Model &m, const Tree &th, const FlatPaths &paths) {
  if (rank_mode_) throw std::runtime_error("one slot per process: reset(model, tree, owned columns, n_global)");
  build(th, paths, paths.n_sites, false);
  reset(m);
}

void SingleSiteSampler::reset(const Model &m, const Tree &th, const FlatPaths &owned, uint64_t n_global) {
  if (!rank_mode_) throw std::runtime_error("reset(..., n_global) belongs to the one-slot-per-process constructor");
  build(th, owned, n_global, true);
  reset(m);
}

void SingleSiteSampler::build(const Tree &th, const FlatPaths &paths, uint64_t n, bool rank_mode) {
  n_nodes_ = th.n_nodes();
  n_sites_ = n;
  drop_parts();
  const uint64_t H = halo_for(burn_in, batch);
  const uint64_t min_part = 2 * H + 2 * kBlock;
  // slots: as many of the requested GPUs as the genome can feed, cut on whole statistics rows
  const std::vector<uint64_t> cut = shard_cuts(n, rank_mode ? world_ : devices_.size(), burn_in, batch, row_blocks_);
  const size_t G = cut.size() - 1;
  if (rank_mode && G != world_)
    throw std::runtime_error("a genome of " + std::to_string(n) + " sites cannot feed " + std::to_string(world_) + " GPU slots");
  world_ = G;
  // the slots that live in this process, and the columns of the genome `paths` holds
  const size_t g_lo = rank_mode ? (size_t)rank_.rank : 0, g_hi = rank_mode ? (size_t)rank_.rank + 1 : G;
  const uint64_t held_first = rank_mode ? cut[g_lo] : 0;
  if (rank_mode && paths.n_sites != cut[g_lo + 1] - cut[g_lo])
    throw std::runtime_error("slot " + std::to_string(g_lo) + " owns " + std::to_string(cut[g_lo + 1] - cut[g_lo]) +
                             " columns (shard_cuts), got " + std::to_string(paths.n_sites));
  struct Piece { size_t slot; uint64_t a, b; };
  std::vector<Piece> pieces;
  for (size_t g = g_lo; g < g_hi; ++g) {
    const uint64_t len = cut[g + 1] - cut[g];
    size_t k = (size_t)contexts_per_gpu();
    while (k > 1 && len < k * min_part) --k;
    for (size_t j = 0; j < k; ++j) {
      const uint64_t a = j == 0 ? cut[g] : cut[g] + round_to((double)j * (double)len / (double)k, kBlock);
      const uint64_t b = j + 1 == k ? cut[g + 1] : cut[g] + round_to((double)(j + 1) * (double)len / (double)k, kBlock);
      pieces.push_back({g - g_lo, a, b});
    }
  }
  if (pieces.size() == 1 && G == 1 && !force_comm_) {
    check(epv_set_tree(ctx_, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(),
                       th.branches.data()), "epv_set_tree");
    check(epv_upload_paths(ctx_, paths.n_sites, paths.init.data(), paths.offsets.data(),
                           paths.jumps.data(), capacity_, 0), "epv_upload_paths");
    return;
  }
  uint32_t cap = capacity_;
  if (cap == 0) {   // the library's default rule, evaluated once for the whole genome
    if (rank_mode) {
      cap = 16;     // the slots cannot see each other's inputs: start narrow, widen on demand like every overflow
    } else {
      uint64_t maxj = 0;
      for (size_t e = 0; e + 1 < paths.offsets.size(); ++e) maxj = std::max(maxj, paths.offsets[e + 1] - paths.offsets[e]);
      cap = (uint32_t)std::min<uint64_t>(kMaxCap, std::max<uint64_t>(16, 2 * maxj + 8));
    }
  }
  if (rank_mode) {   // every slot must propose under one capacity: at least what its own input needs
    uint64_t maxj = 0;
    for (size_t e = 0; e + 1 < paths.offsets.size(); ++e) maxj = std::max(maxj, paths.offsets[e + 1] - paths.offsets[e]);
    if (maxj > cap) throw std::runtime_error("one slot per process: pass a capacity of at least " + std::to_string(maxj));
  }
  halo_ = H;
  slots_.resize(g_hi - g_lo);
  for (size_t g = g_lo; g < g_hi; ++g) {
    Slot &sl = slots_[g - g_lo];
    sl.device = devices_[g - g_lo];
    sl.gidx = g;
    sl.first = cut[g];
    sl.last = cut[g + 1];
    sl.n_blocks = (cut[g + 1] - cut[g] + kBlock - 1) / kBlock;
    sl.n_rows = (sl.n_blocks + row_blocks_ - 1) / row_blocks_;
  }
  max_rows_ = 0;   // the same on every process: all slots of the run
  for (size_t g = 0; g < G; ++g)
    max_rows_ = std::max<uint64_t>(max_rows_, ((cut[g + 1] - cut[g] + kBlock - 1) / kBlock + row_blocks_ - 1) / row_blocks_);
  rows_of_slot_.assign(G, 0);
  for (size_t g = 0; g < G; ++g) rows_of_slot_[g] = ((cut[g + 1] - cut[g] + kBlock - 1) / kBlock + row_blocks_ - 1) / row_blocks_;
  const size_t P = pieces.size();
  for (size_t p = 0; p < P; ++p) {
    Part q;
    q.slot = pieces[p].slot;
    q.a = pieces[p].a;
    q.b = pieces[p].b;
    q.lo = q.a - (q.a > 0 ? H : 0);
    q.hi = q.b + (q.b < n ? H : 0);
    if (p == 0) {
      q.ctx = ctx_;
    } else {
      q.ctx = epv_create(slots_[q.slot].device);
      if (!q.ctx)
        throw std::runtime_error("cannot open a context on HIP device " + std::to_string(slots_[q.slot].device));
    }
    if (parts_.empty() || parts_.back().slot != q.slot) slots_[q.slot].part0 = parts_.size();
    slots_[q.slot].part1 = parts_.size() + 1;
    parts_.push_back(q);
    epv_ctx *c = q.ctx;
    check_on(c, epv_set_tree(c, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(), th.branches.data()),
             "epv_set_tree");
    const FlatPaths part = slice_sites_padded(paths, held_first, q.lo, q.hi);
    const double dummy = 0.0;
    check_on(c, epv_upload_paths(c, part.n_sites, part.init.data(), part.offsets.data(),
                                 part.jumps.empty() ? &dummy : part.jumps.data(), cap, q.lo), "epv_upload_paths");
    check_on(c, epv_set_global_length(c, n), "epv_set_global_length");
  }
  if (rank_mode) {
    // one communicator rank in this process, made once (an RCCL id serves one ncclCommInitRank)
    if (!rank_comm_ && epv_comm_init_rank(rank_.device, rank_.world, rank_.rank, rank_.id, &rank_comm_) != EPV_OK)
      throw std::runtime_error("cannot join the RCCL communicator as rank " + std::to_string(rank_.rank) + " of " +
                               std::to_string(rank_.world));
    slots_[0].comm = rank_comm_;
  } else if (G > 1 || force_comm_) {
    // one communicator rank per slot, all driven from this process: RCCL over the node's xGMI
    // links when the devices are distinct, the loopback transport when they repeat
    std::vector<int> devs(G);
    std::vector<epv_comm *> comms(G, nullptr);
    for (size_t g = 0; g < G; ++g) devs[g] = slots_[g].device;
    if (epv_comm_init_all((int)G, devs.data(), comms.data()) != EPV_OK)
      throw std::runtime_error("cannot set up the RCCL communicator over " + std::to_string(G) + " GPU slot(s)");
    for (size_t g = 0; g < G; ++g) slots_[g].comm = comms[g];
  }
}

void SingleSiteSampler::init_paths_indep(const Model &m, const Tree &th, const std::vector<uint8_t> &root_seq,
                                         const std::vector<uint8_t> &leaf_seq, uint64_t seed) {
  if (root_seq.size() != leaf_seq.size()) throw std::runtime_error("sequences differ in length");
  drop_parts();   // epievo_sim_pairwise's path runs on one context
  n_nodes_ = th.n_nodes();
  n_sites_ = root_seq.size();
  check(epv_set_tree(ctx_, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(),
                     th.branches.data()), "epv_set_tree");
  check(epv_set_model(ctx_, m.rates.data(), m.T.data()), "epv_set_model");
  check(epv_init_paths_indep(ctx_, root_seq.size(), root_seq.data(), leaf_seq.data(), seed, capacity_),
        "epv_init_paths_indep");
  check(epv_reset(ctx_), "epv_reset");
}

void SingleSiteSampler::reset(const Model &m) {
  apply_sample_root();
  if (!sharded()) {
    check(epv_set_model(ctx_, m.rates.data(), m.T.data()), "epv_set_model");
    check(epv_reset(ctx_), "epv_reset");
    return;
  }
  for (Part &p : parts_) check_on(p.ctx, epv_set_model(p.ctx, m.rates.data(), m.T.data()), "epv_set_model");
  refresh_parts();
  // the parts' cached likelihoods: launched on every part's stream, not waited for -- the MCMC calls
  // that follow queue up behind them (reset -> run_mcmc is how the EM loop goes,
  // epievo_est_params_histories.cpp:241-248), download() and the other readers synchronise anyway
  for (Part &p : parts_) check_on(p.ctx, epv_reset_async(p.ctx), "epv_reset_async");
}

// words appended to a slot's rows in the all-gather: its accept count
static const uint64_t kTail = 8;

void SingleSiteSampler::ensure_stat_buffers() {
  if (stat_batch_ >= batch && slots_[0].d_blocks) return;
  free_stat_buffers();
  const uint64_t V = ((uint64_t)n_nodes_ - 1) * 16;
  for (Slot &s : slots_) {
    epv_ctx *c = parts_[s.part0].ctx;
    check_on(c, epv_dev_alloc(c, batch * s.n_blocks * V * sizeof(double), &s.d_blocks), "epv_dev_alloc");
    if (s.comm) {
      const uint64_t piece = max_rows_ * batch * V + kTail;
      check_on(c, epv_dev_alloc(c, piece * sizeof(double), &s.d_rows), "epv_dev_alloc");
      check_on(c, epv_dev_alloc(c, world_ * piece * sizeof(double), &s.d_gather), "epv_dev_alloc");
    }
  }
  stat_batch_ = batch;
}

void SingleSiteSampler::run_mcmc(uint64_t seed, uint64_t em_iteration,
                                 std::vector<std::vector<double>> &J,
                                 std::vector<std::vector<double>> &D, double &acceptance_rate) {
  apply_sample_root();
  const size_t B = (size_t)n_nodes_ - 1;
  std::vector<double> Jf(B * 8), Df(B * 8);
  uint64_t n_acc = 0;
  const uint32_t base = (uint32_t)(em_iteration * (burn_in + batch));
  if (!sharded()) {
    check_mcmc(epv_run_mcmc(ctx_, burn_in, batch, seed, base, Jf.data(), Df.data(), &n_acc), "epv_run_mcmc");
  } else {
    ensure_stat_buffers();
    // one host thread per part: their colour phases run concurrently, each GPU on its own,
    // the contexts of one GPU on their own streams
    const size_t P = parts_.size();
    std::vector<int> rcs(P, EPV_OK);
    std::vector<uint64_t> acc(P, 0);
    ThreadGroup th;
    for (size_t p = 0; p < P; ++p)
      th.spawn([&, p] {
        const Part &q = parts_[p];
        const Slot &s = slots_[q.slot];
        rcs[p] = epv_run_mcmc_blocks(q.ctx, burn_in, batch, seed, base, static_cast<double *>(s.d_blocks),
                                     s.n_blocks, ((int64_t)q.lo - (int64_t)s.first) / (int64_t)kBlock, &acc[p]);
      });
    th.join();
    for (size_t p = 0; p < P; ++p) {
      epv_ctx *c = parts_[p].ctx;
      if (rcs[p] == EPV_ERR_CAPACITY) {   // absorbed as in check_mcmc; refresh_parts() evens the widths out
        uint32_t cap = 0;
        if (epv_get_capacity(c, &cap) == EPV_OK && cap < kMaxCap) {
          capacity_events.push_back(std::string("epv_run_mcmc_blocks: ") + epv_last_error(c));
          check_on(c, epv_set_capacity(c, std::min(kMaxCap, cap * 2u)), "epv_set_capacity");
          rcs[p] = EPV_OK;
        }
      }
      check_on(c, rcs[p], "epv_run_mcmc_blocks");
      n_acc += acc[p];
    }
    if (!slots_[0].comm) {
      const Slot &s = slots_[0];
      check(epv_reduce_blocks(ctx_, static_cast<const double *>(s.d_blocks), s.n_blocks, batch, 1, Jf.data(), Df.data()),
            "epv_reduce_blocks");
    } else {
      // the one collective of an EM iteration: every GPU's rows of integer statistics, its accept
      // count riding in the tail of its piece
      const uint64_t row_words = batch * B * 16, piece = max_rows_ * row_words + kTail;
      for (Slot &s : slots_) {
        epv_ctx *c = parts_[s.part0].ctx;
        check_on(c, epv_blocks_to_rows(c, static_cast<const double *>(s.d_blocks), s.n_blocks, batch, row_blocks_,
                                       static_cast<double *>(s.d_rows)), "epv_blocks_to_rows");
        // the tail of the piece: the slot's accept count, and the jump capacity of its contexts
        // (a slot that absorbed an overflow widened its slots: every slot of the run must follow
        // before the next halo exchange, whose column size depends on it)
        uint64_t tail[2] = {0, 0};
        for (size_t p = s.part0; p < s.part1; ++p) {
          tail[0] += acc[p];
          uint32_t cap = 0;
          check_on(parts_[p].ctx, epv_get_capacity(parts_[p].ctx, &cap), "epv_get_capacity");
          tail[1] = std::max<uint64_t>(tail[1], cap);
        }
        check_on(c, epv_dev_write(c, static_cast<double *>(s.d_rows) + max_rows_ * row_words, tail, sizeof tail),
                 "epv_dev_write");
      }
      if (epv_comm_group_start() != EPV_OK) throw std::runtime_error("epv_comm_group_start failed");
      for (Slot &s : slots_)
        check_comm(s.comm, epv_comm_all_gather(s.comm, s.d_rows, s.d_gather, piece * sizeof(double)), "epv_comm_all_gather");
      if (epv_comm_group_end() != EPV_OK) throw std::runtime_error("epv_comm_group_end failed (statistics all-gather)");
      for (Slot &s : slots_) check_comm(s.comm, epv_comm_sync(s.comm), "epv_comm_sync");
      // every GPU now holds the same rows; the host M-step needs one copy of the totals
      check(epv_reduce_gathered_rows(ctx_, static_cast<const double *>(slots_[0].d_gather), (uint32_t)world_, max_rows_, piece,
                                     rows_of_slot_.data(), batch, 1, Jf.data(), Df.data()), "epv_reduce_gathered_rows");
      n_acc = 0;
      uint64_t cap_all = 0;
      for (size_t g = 0; g < world_; ++g) {
        uint64_t v[2] = {0, 0};
        check(epv_dev_read(ctx_, v, static_cast<const double *>(slots_[0].d_gather) + g * piece + max_rows_ * row_words, sizeof v),
              "epv_dev_read");
        n_acc += v[0];
        cap_all = std::max(cap_all, v[1]);
      }
      for (Part &q : parts_) {     // (slots in other processes may have grown)
        uint32_t cap = 0;
        check_on(q.ctx, epv_get_capacity(q.ctx, &cap), "epv_get_capacity");
        if (cap < cap_all) check_on(q.ctx, epv_set_capacity(q.ctx, (uint32_t)cap_all), "epv_set_capacity");
      }
    }
  }
  J.assign(n_nodes_, {});
  D.assign(n_nodes_, {});
  for (size_t b = 1; b <= B; ++b) {
    J[b].assign(Jf.begin() + (b - 1) * 8, Jf.begin() + b * 8);
    D[b].assign(Df.begin() + (b - 1) * 8, Df.begin() + b * 8);
  }
  acceptance_rate = static_cast<double>(n_acc) / (batch * (n_sites_ - 2));
}

size_t SingleSiteSampler::sweeps(size_t n, uint64_t seed, uint32_t sweep_base) {
  apply_sample_root();
  uint64_t n_acc = 0;
  if (!sharded()) {
    check_mcmc(epv_sweep(ctx_, n, seed, sweep_base, &n_acc), "epv_sweep");
    return n_acc;
  }
  // a sharded genome can run as many sweeps as its halos last, then they are refreshed
  size_t done = 0;
  while (done < n) {
    uint64_t left = ~0ull;
    for (Part &p : parts_) { uint64_t v = 0; check_on(p.ctx, epv_halo_phases_left(p.ctx, &v), "epv_halo_phases_left"); left = std::min(left, v); }
    const size_t kk = std::min<size_t>(n - done, (size_t)(left / 3));
    if (kk == 0) {
      refresh_parts();
      for (Part &p : parts_) check_on(p.ctx, epv_reset(p.ctx), "epv_reset");
      continue;
    }
    const size_t P = parts_.size();
    std::vector<int> rcs(P, EPV_OK);
    std::vector<uint64_t> acc(P, 0);
    ThreadGroup th;
    for (size_t p = 0; p < P; ++p)
      th.spawn([&, p] { rcs[p] = epv_sweep(parts_[p].ctx, kk, seed, sweep_base + (uint32_t)done, &acc[p]); });
    th.join();
    bool grew = false;
    for (size_t p = 0; p < P; ++p) {
      epv_ctx *c = parts_[p].ctx;
      if (rcs[p] == EPV_ERR_CAPACITY) {
        uint32_t cap = 0;
        if (epv_get_capacity(c, &cap) == EPV_OK && cap < kMaxCap) {
          capacity_events.push_back(std::string("epv_sweep: ") + epv_last_error(c));
          check_on(c, epv_set_capacity(c, std::min(kMaxCap, cap * 2u)), "epv_set_capacity");
          rcs[p] = EPV_OK;
          grew = true;
        }
      }
      check_on(c, rcs[p], "epv_sweep");
      n_acc += acc[p];
    }
    // the parts update their shared halo columns redundantly: they must keep proposing under
    // the same capacity, or one accepts what the other rejects
    if (grew) equalize_capacity();
    done += kk;
  }
  return n_acc;
}

void SingleSiteSampler::scale_jump_times(const std::vector<double> &new_branches) {
  if (!sharded()) { check(epv_scale_jump_times(ctx_, new_branches.data()), "epv_scale_jump_times"); return; }
  for (Part &p : parts_) check_on(p.ctx, epv_scale_jump_times(p.ctx, new_branches.data()), "epv_scale_jump_times");
}

void SingleSiteSampler::upload(const Tree &th, const FlatPaths &paths) {
  drop_parts();   // the site-independent stage runs on one context
  n_nodes_ = th.n_nodes();
  n_sites_ = paths.n_sites;
  check(epv_set_tree(ctx_, th.n_nodes(), th.parent_ids.data(), th.subtree_sizes.data(),
                     th.branches.data()), "epv_set_tree");
  // the kernels of the site-independent stage never read the 8 rates; park neutral ones
  const double ones[8] = {1, 1, 1, 1, 1, 1, 1, 1}, T[4] = {0.5, 0.5, 0.5, 0.5};
  check(epv_set_model(ctx_, ones, T), "epv_set_model");
  const double dummy = 0.0;
  check(epv_upload_paths(ctx_, paths.n_sites, paths.init.data(), paths.offsets.data(),
                         paths.jumps.empty() ? &dummy : paths.jumps.data(), capacity_, 0),
        "epv_upload_paths");
}

void SingleSiteSampler::get_sufficient_statistics(std::vector<std::vector<double>> &J,
                                                  std::vector<std::vector<double>> &D) {
  if (sharded()) throw std::runtime_error("get_sufficient_statistics: load the paths with upload() (one context)");
  const size_t B = (size_t)n_nodes_ - 1;
  std::vector<double> Jf(B * 8), Df(B * 8);
  check(epv_get_sufficient_statistics(ctx_, Jf.data(), Df.data()), "epv_get_sufficient_statistics");
  J.assign(n_nodes_, {});
  D.assign(n_nodes_, {});
  for (size_t b = 1; b <= B; ++b) {
    J[b].assign(Jf.begin() + (b - 1) * 8, Jf.begin() + b * 8);
    D[b].assign(Df.begin() + (b - 1) * 8, Df.begin() + b * 8);
  }
}

void SingleSiteSampler::indep_expectation(const double rates[2], std::vector<double> &J,
                                          std::vector<double> &D) {
  J.assign(((size_t)n_nodes_ - 1) * 2, 0.0);
  D.assign(((size_t)n_nodes_ - 1) * 2, 0.0);
  check(epv_indep_expectation(ctx_, rates, J.data(), D.data()), "epv_indep_expectation");
}

void SingleSiteSampler::indep_sufficient_statistics(std::vector<double> &J, std::vector<double> &D) {
  J.assign(((size_t)n_nodes_ - 1) * 2, 0.0);
  D.assign(((size_t)n_nodes_ - 1) * 2, 0.0);
  check(epv_indep_sufficient_statistics(ctx_, J.data(), D.data()), "epv_indep_sufficient_statistics");
}

void SingleSiteSampler::indep_update_paths(const double rates[2], uint64_t seed, uint32_t sweep) {
  check_mcmc(epv_indep_update_paths(ctx_, rates, seed, sweep), "epv_indep_update_paths");
}

void SingleSiteSampler::download(FlatPaths &paths) {
  if (sharded()) {
    // every part on its own host thread (its context has its own stream): gather, copy, trim halos
    std::vector<FlatPaths> owned(parts_.size());
    std::vector<std::string> errors(parts_.size());
    ThreadGroup workers;
    for (size_t i = 0; i < parts_.size(); ++i)
      workers.spawn([this, i, &owned, &errors] {
        try {
          Part &q = parts_[i];
          epv_ctx *c = q.ctx;
          uint64_t total = 0;
          check_on(c, epv_paths_total_jumps(c, &total), "epv_paths_total_jumps");
          FlatPaths p;
          p.n_sites = q.hi - q.lo;
          p.n_nodes = n_nodes_;
          const uint64_t E = (uint64_t)(n_nodes_ - 1) * p.n_sites;
          p.init.assign(E, 0);
          p.offsets.assign(E + 1, 0);
          p.jumps.assign(total ? total : 1, 0.0);
          check_on(c, epv_download_paths(c, p.init.data(), p.offsets.data(), p.jumps.data()), "epv_download_paths");
          p.jumps.resize(total);
          owned[i] = slice_sites(p, q.a - q.lo, q.b - q.lo);
        } catch (const std::exception &e) { errors[i] = e.what(); }
      });
    workers.join();
    for (const std::string &e : errors) if (!e.empty()) throw std::runtime_error(e);
    paths = concat_sites(owned);
    return;
  }
  uint64_t total = 0;
  check(epv_paths_total_jumps(ctx_, &total), "epv_paths_total_jumps");
  paths.n_sites = n_sites_;
  paths.n_nodes = n_nodes_;
  const uint64_t E = (uint64_t)(n_nodes_ - 1) * n_sites_;
  paths.init.assign(E, 0);
  paths.offsets.assign(E + 1, 0);
  paths.jumps.assign(total ? total : 1, 0.0);
  check(epv_download_paths(ctx_, paths.init.data(), paths.offsets.data(), paths.jumps.data()),
        "epv_download_paths");
  paths.jumps.resize(total);
}

}  // namespace epv
