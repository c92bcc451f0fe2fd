// epv_abi.hip -- the extern "C" boundary of include/epievo_mi355x.h: device memory,
// stream, launches.  No torch types, no exceptions across the boundary, no CPU
// fallback: when HIP is unavailable every entry point fails with EPV_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "epievo_mi355x.h"
#include "epv_device.h"

#define EPV_API extern "C" __attribute__((visibility("default")))

#include "epv_kernels.h"  // all __global__ kernels (single translation unit, no -fgpu-rdc)
#include "epv_math.h"

struct epv_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  EpvDev S{};
  bool have_tree = false, have_model = false, have_paths = false, have_reset = false;
  // host copies
  std::vector<uint32_t> parent, subtree;
  std::vector<double> blen;
  EpvModelConst model{};
  uint64_t first = 0, last = 0;          // fixed update range (no halo mode)
  uint64_t halo_left = 0, halo_right = 0;  // halo mode: widths of the halo column blocks
  bool halo_mode = false;
  uint64_t phases_used = 0;              // colour phases run since the halos were fresh
  // device allocations
  EpvModelConst *d_model = nullptr;
  uint32_t *d_parent = nullptr, *d_subtree = nullptr;
  double *d_blen = nullptr;
  unsigned long long *d_counters = nullptr;
  unsigned long long *h_counters = nullptr;  // pinned staging for the sharded counters
  unsigned long long *d_cnt_snap = nullptr;  // the counters as they stood when the batch sweeps began (stream-ordered copy:
  unsigned long long *h_cnt_snap = nullptr;  // the host does not wait for the burn-in to read the accept base)
  double *d_partial[2] = {nullptr, nullptr};  // tree-reduction ping-pong
  uint64_t partial_cap[2] = {0, 0};   // doubles allocated in d_partial[0], [1]
  unsigned long long *d_sweep_tot = nullptr;  // [sweep][B*16] integer statistics of the batch sweeps
  uint64_t sweep_tot_cap = 0;                 // sweeps allocated
  double *d_statscale = nullptr;              // [N] 2^k_b of the fixed-point dwell times (epv_suffstat_kernel)
  std::vector<double> statscale;              // host copy; refreshed when the tree or the genome length changes
  double *d_scale = nullptr;
  double *d_lvl = nullptr;      // level outputs of epv_reduce_blocks (all batch sweeps at once)
  uint64_t lvl_cap = 0;
  double *d_rows = nullptr;     // compacted statistic rows of the whole genome (epv_reduce_gathered_rows)
  uint64_t rows_cap = 0;
  uint8_t *d_stage = nullptr;   // packed-column staging for the halo exchange (grown on demand)
  uint64_t stage_cap = 0;
  hipEvent_t ev_copy[2] = {nullptr, nullptr};   // epv_copy_columns_async: "slot s of my staging buffer is packed"
  EpvIndepConst *d_indep = nullptr;  // [N] constants of the site-independent model
  // launch shape of the MH kernel
  uint32_t mh_threads = 64, pool_entries = 0;
  bool mh_gpool = false;        // record pool of the propose kernel in global memory (large trees)
  double *d_gpool = nullptr;
  uint64_t gpool_cap = 0;       // doubles allocated
  uint64_t gpool_need = 0, gpool2_need = 0, gpool3_need = 0;   // what the plans ask for; allocated when a launch first uses the slab
  // second-generation proposal kernel (epv_propose2.h)
  bool use_p2 = true;            // EPV_PROPOSE_V1=1 falls back to the first kernel (A/B runs)
  uint32_t p2_pool = 0, p2_list_cap = 0;   // LDS: doubles per wave; global slab: rows per lane + heavy records
  uint32_t p2_waves = 1;                   // waves per block
  bool p2_gpool = false;
  size_t p2_lds = 0;
  double *d_gpool2 = nullptr;
  uint64_t gpool2_cap = 0;
  double *d_gpool3 = nullptr;
  uint64_t gpool3_cap = 0;
  double *d_segtab = nullptr;    // [B][4][6] single-segment matrices, refreshed by epv_reset
  // third proposal kernel (epv_propose3.h): large trees, where the record pool does not fit LDS
  int use_p3 = -1;               // -1 = whenever the plan allows it, EPV_PROPOSE_V3=0/1 forces
  bool p3 = false;               // decided by plan_p3 for the uploaded tree and paths
  uint32_t p3_list_cap = 0, p3_qrows = 0, p3_nup = 0, p3_depth = 0;
  uint32_t p3_slots = 0;         // slabs per XCD handed out to resident blocks (0 = a slab per block of the launch)
  uint32_t *d_slabflags = nullptr;
  size_t p3_lds = 0;
  uint32_t *d_nodetab = nullptr; // node words and level lists of epv_mh_propose3_kernel (EPV_P3_*)
  uint32_t phase_parity = 0;     // accept lists are double-buffered by phase parity
  int use_seg = -1;              // segment-parallel jump kernels (epv_jumps2.h): -1 = by workload
                                 // (long branches, kbar >= 0.25), EPV_SEG_JUMPS=0/1 forces
  // fused colour phase (epv_propose2.h, FUSED): one kernel per phase for launches of few waves
  int use_fused = -1;            // -1 = by launch size (EPV_FUSED_MAX_WAVES), EPV_FUSED_PHASE=0/1 forces
  uint32_t fused_max_waves = 3072;   // measured on tree.nwk: +46 % at 520 waves, +20 % at 1700, +5 % at 2600, -4..-14 % at 5200 (tools/fused_scan.sh)
  bool fused = false;            // decided by plan_p2 for the uploaded paths
  uint32_t fused_lanes = 64;     // sites per wave of the fused phase (64 / 32 / 16: EPV_FUSED_LANES, else by launch size)
  EpvFused F{};                  // per-wave lists, allocated on first use
  uint64_t fused_waves = 0;      // waves the lists are allocated for
  uint32_t tasks_per_wave = 0;   // epv_mh_jumps_kernel: lanes of a wave that own a task (0 = by workload)
  double kbar = 0.0;  // mean jumps per (site, branch) of the uploaded paths
  double fwd_alloc_ms = 0.0, fwd_sim_ms = 0.0;   // epv_forward_simulate: device memory management / the simulation itself
  size_t mh_lds = 0;
  // counters / timing
  uint64_t n_sweeps = 0, tot_overflow = 0, tot_coop = 0;
  bool timing = false;           // events around THIS launch (set per launch from timing_every)
  uint32_t timing_every = 0;     // 0 = off, N = HIP events around every N-th colour-phase launch
  uint64_t timing_seen = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double timed_ms = 0.0;
  uint64_t timed_launches = 0;
};

namespace {

int fail(epv_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}
#define HIP_TRY(c, call)                                                              \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess)                                                             \
      return fail((c), EPV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class T>
void dfree(T *&p) {
  if (p) { (void)hipFree(p); p = nullptr; }
}

// device scratch that is released on every exit path of an entry point
template <class T>
struct DevTmp {
  T *p = nullptr;
  DevTmp() = default;
  DevTmp(const DevTmp &) = delete;
  DevTmp &operator=(const DevTmp &) = delete;
  ~DevTmp() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t count) { return hipMalloc(&p, count * sizeof(T)); }
  T *release() { T *q = p; p = nullptr; return q; }
};

void free_paths(epv_ctx *c) {
  dfree(c->S.meta); dfree(c->S.jumps); dfree(c->S.sel); dfree(c->S.tri);
  dfree(c->S.prop_llr); dfree(c->S.prop_flag); dfree(c->S.prop_states); dfree(c->S.tasks); dfree(c->S.alist);
  dfree(c->S.segs); dfree(c->S.segout); dfree(c->S.btasks); dfree(c->S.bfirst);
  dfree(c->F.segs); dfree(c->F.outs); dfree(c->F.bt); dfree(c->F.bfirst);
  c->fused_waves = 0;
  dfree(c->d_partial[0]); dfree(c->d_partial[1]);
  c->partial_cap[0] = c->partial_cap[1] = 0;
  c->have_paths = c->have_reset = false;
}

int ensure_stage(epv_ctx *c, uint64_t bytes) {
  if (bytes <= c->stage_cap) return EPV_OK;
  dfree(c->d_stage);
  c->stage_cap = 0;
  HIP_TRY(c, hipMalloc(&c->d_stage, bytes));
  c->stage_cap = bytes;
  return EPV_OK;
}

size_t const_lds_bytes(uint32_t N) { return (size_t)((20u + N + 1u) & ~1u) * 8u; }

// choose the MH launch shape: one wave per block; the record pool is as large as fits
// six blocks per CU, but never smaller than one lane's worst case
int plan_mh(epv_ctx *c) {
  const uint32_t B = c->S.B, C = c->S.C, N = c->S.N;
  const uint64_t worst = (uint64_t)B * (2u * C + 2u);
  // per wave: node table N*64 u32 (rounded to 16 B) + pool of 16-byte records
  const size_t fixed = const_lds_bytes(N) + (((size_t)N * 64u * 4u + 15u) & ~(size_t)15u);
  if (fixed > 150u * 1024u) return fail(c, EPV_ERR_ARG, "tree too large for the 160 KiB LDS node table");
  // typical demand of 64 lanes: B * (2 kbar + 2) records each, with a 30 % margin
  const uint64_t typical = (uint64_t)(64.0 * B * (2.0 * c->kbar + 2.0) * 1.3) + 32u;
  const uint64_t max_fit = fixed + 16u < 160u * 1024u ? (160u * 1024u - fixed) / 16u : 0u;
  const uint64_t want = std::max<uint64_t>(worst, typical);
  c->mh_threads = 64;
  // LDS pool while it leaves >= 8 waves per CU (2 per SIMD); otherwise a global-memory slab
  // per block (the working set of the resident waves stays in L2 / Infinity Cache)
  const bool lds_ok = want <= max_fit && (fixed + std::min(want, max_fit) * 16u) * 8u <= 160u * 1024u;
  // tuning knobs (tools/ab_env.py): EPV_FORCE_LDS_POOL / EPV_FORCE_GLOBAL_POOL
  const bool use_lds = std::getenv("EPV_FORCE_GLOBAL_POOL") ? false
                       : std::getenv("EPV_FORCE_LDS_POOL") ? want <= max_fit : lds_ok;
  if (use_lds) {
    const uint64_t pool = std::min(want, max_fit);
    c->mh_gpool = false;
    c->pool_entries = (uint32_t)pool;
    c->mh_lds = fixed + (size_t)pool * 16u;
    return EPV_OK;
  }
  // global slab: `worst` ROWS of 64 interleaved records per wave (a lane can always run)
  const uint64_t pool = worst;
  const uint64_t blocks = (c->S.phase_cap + 63u) / 64u;
  c->gpool_need = blocks * pool * 128u;     // (allocated by ensure_slab when a launch takes this kernel)
  c->mh_gpool = true;
  c->pool_entries = (uint32_t)pool;
  c->mh_lds = fixed;
  return EPV_OK;
}

// the two ping-pong buffers of the canonical tree reduction, sized in DOUBLES for the launch
// that uses them: level 0 holds one row per block, level 1 one row per 256 blocks
// launch shape of epv_mh_propose2_kernel: per wave the constants, the matrix table, the node
// table and a pool of doubles shared by the Felsenstein records (2 doubles) and the heavy-segment
// records (EPV_HREC doubles).  The pool covers the typical demand of 64 lanes with a margin (a wave that
// needs more runs in rounds) and always one lane's worst case.
bool seg_jumps_on(const epv_ctx *c) { return c->use_seg < 0 ? c->kbar >= 0.25 : c->use_seg != 0; }
double p2_margin() {
  if (const char *e = std::getenv("EPV_P2_MARGIN")) { const double v = std::atof(e); if (v >= 1.0 && v <= 4.0) return v; }
  return 1.15;     // 1.25 -> 1.15 with the 8-double records: one more wave per CU, ~4 % of the waves take a second round
}
int plan_p2(epv_ctx *c) {
  const uint32_t B = c->S.B, C = c->S.C, N = c->S.N;
  const size_t shared = const_lds_bytes(N) + (size_t)B * 4u * 6u * 8u;     // constants, matrix table: once per block
  const size_t per_wave_fixed = ((((size_t)N * 64u + 1u) / 2u + 1u) & ~(size_t)1u) * 8u +
                                ((3u * 64u + 2u) * (size_t)B * sizeof(epv_meta_t) + 15u) / 16u * 16u;   // node table, meta cache (+ 2 edge columns)
  const size_t fixed = shared + per_wave_fixed;
  // one lane's worst case: every branch with 2C+1 segments (records K+1, heavy K)
  const uint64_t worst_rec = (uint64_t)B * (2u * C + 2u), worst_heavy = (uint64_t)B * (2u * C + 1u);
  // the fused phase: few waves (a launch that leaves SIMDs idle pays one wave chain instead of
  // three to five), at most 64 segments per branch (the hand-over's bit word), lists for the worst
  // case of every wave within 4 GB
  // sites per wave: halve while every SIMD could still get two waves
  uint32_t f_lanes = 64u;
  if (const char *e = std::getenv("EPV_FUSED_LANES")) { const int v = std::atoi(e); if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) f_lanes = (uint32_t)v; }
  else while (f_lanes > 16u && (c->S.phase_cap + f_lanes / 2u - 1u) / (f_lanes / 2u) <= 2048u) f_lanes /= 2u;
  const uint64_t phase_waves = (c->S.phase_cap + 63u) / 64u;
  const uint64_t f_seg_cap = 64ull * B * (2u * C + 1u), f_bt_cap = 64ull * B;
  const uint64_t f_bytes = (c->S.phase_cap + f_lanes - 1u) / f_lanes * (f_seg_cap * (sizeof(EpvSegTask) + sizeof(EpvSegOut)) + f_bt_cap * 12u);
  bool fused = c->use_p2 && 2u * C + 1u <= 64u && f_bytes <= (4ull << 30) &&
               (c->use_fused < 0 ? phase_waves <= c->fused_max_waves : c->use_fused != 0);
  // heavy-segment records: 8 doubles, 10 when the segment-parallel jump kernels read them back
  const uint64_t hrec = (fused || seg_jumps_on(c)) ? EPV_HREC : EPV_HREC_SHORT;
  const uint64_t worst_dbl = 2u * worst_rec + hrec * worst_heavy;
  // typical: K = 1 + Poisson(2 kbar) segments per branch; heavy segments E[K; K >= 2]
  const double lam = 2.0 * c->kbar;
  const double heavy_per_branch = (1.0 + lam) - std::exp(-lam);
  uint32_t n_internal = 0;
  for (uint32_t node = 1; node < N; ++node) n_internal += c->subtree[node] != 1u;
  const double rec_per_lane = B * (1.0 + lam) + n_internal;     // K per branch, +1 for an internal node's q
  const uint64_t typical_dbl = (uint64_t)(64.0 * (2.0 * rec_per_lane + (double)hrec * B * heavy_per_branch) * p2_margin()) + 64u;
  const uint64_t max_fit = fixed + 64u < 160u * 1024u ? (160u * 1024u - fixed) / 8u : 0u;
  uint64_t want = std::max(worst_dbl, typical_dbl);
  // the fused phase reuses the pool for the search's cooperative area and then for the accept
  // stage's task table, results, accumulators (992 doubles) and meta words (3 B columns of 64)
  if (fused) want = std::max<uint64_t>(want, std::max<uint64_t>(EPV_COOP_BYTES / 8u, 992u + 48u * (uint64_t)B));
  const bool lds_ok = want <= max_fit && (fixed + want * 8u) * 5u <= 160u * 1024u;   // >= 5 waves per CU
  const bool use_lds = std::getenv("EPV_FORCE_GLOBAL_POOL") ? false
                       : std::getenv("EPV_FORCE_LDS_POOL") ? want <= max_fit : lds_ok;
  if (fixed > 150u * 1024u) return fail(c, EPV_ERR_ARG, "tree too large for the 160 KiB LDS node table");
  c->p2_waves = 1;
  if (const char *e = std::getenv("EPV_P2_WAVES_PER_BLOCK")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) c->p2_waves = (uint32_t)v; }
  c->fused = fused && use_lds;
  c->fused_lanes = f_lanes;
  if (use_lds) {
    c->p2_gpool = false;
    c->p2_pool = (uint32_t)((want + 1u) & ~(uint64_t)1u);
    c->p2_list_cap = 0;
    c->p2_lds = shared + c->p2_waves * (per_wave_fixed + (size_t)c->p2_pool * 8u);
    return EPV_OK;
  }
  // global slab per wave: worst_rec rows of 64 interleaved records + a heavy list for the wave
  const uint64_t rows = worst_rec;
  const uint64_t list_cap = std::max<uint64_t>(worst_heavy, (uint64_t)(64.0 * B * heavy_per_branch * 2.0) + 64u);
  const uint64_t blocks = (c->S.phase_cap + 63u) / 64u;
  c->gpool2_need = blocks * (rows * 128u + list_cap * EPV_HREC);
  c->p2_gpool = true;
  c->p2_pool = (uint32_t)rows;
  c->p2_list_cap = (uint32_t)list_cap;
  c->p2_lds = shared + c->p2_waves * per_wave_fixed;
  return EPV_OK;
}

// launch shape of epv_mh_propose3_kernel, for trees whose record pool does not fit LDS: per wave a
// 16-bit word per (node, lane) and a stack of partial products in LDS, q rows of the internal nodes
// and the heavy-segment records in a slab of global memory
int plan_p3(epv_ctx *c) {
  c->p3 = false;
  const uint32_t B = c->S.B, C = c->S.C, N = c->S.N;
  if (c->use_p3 == 0 || !c->use_p2 || N > 128u || N < 2u) return EPV_OK;     // (node masks of one or two 64-bit words)
  if (c->use_p3 < 0 && !c->p2_gpool) return EPV_OK;     // the LDS pool is the better place while it fits
  // per node: parent, children, depth; q rows for the internal nodes below the root
  std::vector<uint32_t> depth(N, 0u), c1(N, 0u), c2(N, 0u), kids(N, 0u), qrow(N, 0u);
  uint32_t qrows = 0, max_depth = 0;
  for (uint32_t node = 1; node < N; ++node) {
    const uint32_t par = c->parent[node];
    if (par >= node) return EPV_OK;                      // (pre-order is what epv_set_tree checks; be safe)
    depth[node] = depth[par] + 1u;
    max_depth = std::max(max_depth, depth[node]);
    if (kids[par] == 0u) c1[par] = node; else if (kids[par] == 1u) c2[par] = node;
    ++kids[par];
  }
  for (uint32_t node = 1; node < N; ++node) {
    if (kids[node] > 2u) return EPV_OK;                  // two child fields per node word
    if (c->subtree[node] != 1u) qrow[node] = qrows++;
  }
  if (qrows > 63u) return EPV_OK;                        // (six bits in the node word)
  // tables: node words [N] | internal nodes deepest first [n_up] | their level starts [D + 2] |
  //         all nodes but the root by depth [N - 1] | their level starts [D + 2]
  std::vector<uint32_t> tab;
  for (uint32_t node = 0; node < N; ++node)
    tab.push_back(c->parent[node] | (c1[node] << 7) | (c2[node] << 14) | (qrow[node] << 21) |
                  ((c->subtree[node] == 1u ? 1u : 0u) << 27));
  std::vector<uint32_t> up, upstart(max_depth + 2u, 0u), dn, dnstart(max_depth + 2u, 0u);
  for (uint32_t d = max_depth + 1u; d-- > 0u;) {        // upstart[d + 1] .. upstart[d] = internal nodes of depth d
    if (d <= max_depth && d >= 1u)
      for (uint32_t node = 1; node < N; ++node)
        if (depth[node] == d && c->subtree[node] != 1u) up.push_back(node);
    upstart[d] = (uint32_t)up.size();
  }
  upstart[max_depth + 1u] = 0u;
  for (uint32_t d = 0; d <= max_depth; ++d) {           // dnstart[d] .. dnstart[d + 1] = nodes of depth d
    dnstart[d] = (uint32_t)dn.size();
    if (d >= 1u)
      for (uint32_t node = 1; node < N; ++node)
        if (depth[node] == d) dn.push_back(node);
  }
  dnstart[max_depth + 1u] = (uint32_t)dn.size();
  const uint32_t n_up = (uint32_t)up.size();
  tab.insert(tab.end(), up.begin(), up.end());
  tab.insert(tab.end(), upstart.begin(), upstart.end());
  tab.insert(tab.end(), dn.begin(), dn.end());
  tab.insert(tab.end(), dnstart.begin(), dnstart.end());
  for (uint32_t g = 0; g < max_depth; ++g) {            // pair groups: leaves, then internal nodes by depth
    uint64_t m = 0, m2 = 0;      // nodes 0..63, 64..127
    for (uint32_t node = 1; node < N; ++node) {
      const bool leaf = c->subtree[node] == 1u;
      if (g == 0u ? leaf : (!leaf && depth[node] == g)) (node < 64u ? m : m2) |= 1ull << (node & 63u);
    }
    tab.push_back((uint32_t)m);
    tab.push_back((uint32_t)(m >> 32));
    tab.push_back((uint32_t)m2);
    tab.push_back((uint32_t)(m2 >> 32));
  }
  for (uint32_t node = 0; node < N; ++node) tab.push_back(depth[node]);
  const double lam = 2.0 * c->kbar;
  const double heavy_per_branch = (1.0 + lam) - std::exp(-lam);
  const uint64_t worst_heavy = (uint64_t)B * (2u * C + 1u);
  // EPV_P3_MIN_LIST=1 (tests): one lane's worst case only, so that busy waves run in several rounds
  static const bool min_list = std::getenv("EPV_P3_MIN_LIST") != nullptr;
  const uint64_t list_cap = min_list ? worst_heavy : std::max<uint64_t>(worst_heavy, (uint64_t)(64.0 * B * heavy_per_branch * 1.5) + 64u);
  if (list_cap >= (1ull << 20)) return EPV_OK;             // the pair word's record field
  // a pool of slabs per XCD, claimed by resident blocks (EPV_P3_SLAB_POOL=0: one per block of the launch):
  // 4 blocks of this kernel fit a CU (LDS) and an XCD of the MI355X has 32 CUs: 128 resident blocks at most,
  // 160 slabs per XCD.  Used when the launch has more blocks than the pool has slabs (EPV_P3_SLAB_POOL=2: always)
  static const int pool_env = std::getenv("EPV_P3_SLAB_POOL") ? std::atoi(std::getenv("EPV_P3_SLAB_POOL")) : 1;
  const uint64_t launch_waves = ((c->S.phase_cap + 255u) / 256u) * 4u;
  c->p3_slots = (pool_env == 2 || (pool_env && launch_waves > 8u * 160u * 4u)) ? 160u : 0u;
  const uint64_t waves = c->p3_slots ? 8ull * c->p3_slots * 4u : launch_waves;
  const uint64_t need = waves * ((uint64_t)qrows * 128u + list_cap * EPV_HREC_SHORT);
  if (need * sizeof(double) > (24ull << 30)) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  c->gpool3_need = need;
  if (tab.size() > 2048u) return EPV_OK;
  if (!c->d_nodetab) HIP_TRY(c, hipMalloc(&c->d_nodetab, 2048u * sizeof(uint32_t)));
  if (!c->d_slabflags) {
    HIP_TRY(c, hipMalloc(&c->d_slabflags, 8u * 256u * sizeof(uint32_t)));
    HIP_TRY(c, hipMemset(c->d_slabflags, 0, 8u * 256u * sizeof(uint32_t)));
  }
  HIP_TRY(c, hipMemcpy(c->d_nodetab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  const size_t shared = const_lds_bytes(N) + (size_t)B * 4u * EPV_SEGTAB_DBL * 8u + (tab.size() + 1u) / 2u * 8u;
  const size_t per_wave = ((size_t)EPV_P3_PCAP * 3u + EPV_P3_PCAP / 8u + (max_depth * 64u * 2u + 7u) / 8u + (max_depth + 3u) / 2u) * 8u;   // pair list, pair results, group offsets
  c->p3_lds = shared + 4u * per_wave;
  if (c->p3_lds > 120u * 1024u) return EPV_OK;      // (a very deep tree's group offsets: keep the first kernels)
  c->p3_list_cap = (uint32_t)list_cap;
  c->p3_qrows = qrows;
  c->p3_nup = n_up;
  c->p3_depth = max_depth;
  c->p3 = true;
  return EPV_OK;
}

// per-wave lists of the fused phase, on first use
int ensure_fused_buffers(epv_ctx *c) {
  const uint64_t waves = (c->S.phase_cap + c->fused_lanes - 1u) / c->fused_lanes + 4u;
  const uint64_t seg_cap = 64ull * c->S.B * (2u * c->S.C + 1u), bt_cap = 64ull * c->S.B;
  if (c->F.segs && c->fused_waves >= waves && c->F.seg_cap == seg_cap && c->F.bt_cap == bt_cap) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  dfree(c->F.segs); dfree(c->F.outs); dfree(c->F.bt); dfree(c->F.bfirst);
  c->fused_waves = 0;
  HIP_TRY(c, hipMalloc(&c->F.segs, waves * seg_cap * sizeof(EpvSegTask)));
  HIP_TRY(c, hipMalloc(&c->F.outs, waves * seg_cap * sizeof(EpvSegOut)));
  HIP_TRY(c, hipMalloc(&c->F.bt, waves * bt_cap * sizeof(unsigned long long)));
  HIP_TRY(c, hipMalloc(&c->F.bfirst, waves * bt_cap * sizeof(uint32_t)));
  c->F.seg_cap = (uint32_t)seg_cap;
  c->F.bt_cap = (uint32_t)bt_cap;
  c->fused_waves = waves;
  return EPV_OK;
}

// lists of the segment-parallel jump kernels, on first use
int ensure_seg_buffers(epv_ctx *c) {
  if (c->S.segs) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t btask_cap = c->S.task_cap;
  const uint64_t seg_cap = (uint64_t)((double)c->S.task_cap * std::min(8.0, std::max(0.5, 0.5 + 3.0 * c->kbar))) + 256u;
  HIP_TRY(c, hipMalloc(&c->S.segs, seg_cap * EPV_SHARDS * sizeof(EpvSegTask)));
  HIP_TRY(c, hipMalloc(&c->S.segout, seg_cap * EPV_SHARDS * sizeof(EpvSegOut)));
  HIP_TRY(c, hipMalloc(&c->S.btasks, btask_cap * EPV_SHARDS * sizeof(unsigned long long)));
  HIP_TRY(c, hipMalloc(&c->S.bfirst, btask_cap * EPV_SHARDS * sizeof(uint32_t)));
  c->S.btask_cap = btask_cap;
  c->S.seg_cap = seg_cap;
  return EPV_OK;
}

int ensure_partial_doubles(epv_ctx *c, uint64_t need0, uint64_t need1) {
  if (c->partial_cap[0] >= need0 && c->partial_cap[1] >= need1) return EPV_OK;
  need0 = std::max(need0, c->partial_cap[0]);
  need1 = std::max(need1, c->partial_cap[1]);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  dfree(c->d_partial[0]); dfree(c->d_partial[1]);
  c->partial_cap[0] = c->partial_cap[1] = 0;
  HIP_TRY(c, hipMalloc(&c->d_partial[0], need0 * sizeof(double)));
  HIP_TRY(c, hipMalloc(&c->d_partial[1], need1 * sizeof(double)));
  c->partial_cap[0] = need0;
  c->partial_cap[1] = need1;
  return EPV_OK;
}
int ensure_partials(epv_ctx *c, uint64_t nb_min = 0) {
  const uint64_t nb = std::max<uint64_t>((c->S.n + 255u) / 256u, nb_min);
  const uint64_t V = (uint64_t)c->S.B * 16u;
  return ensure_partial_doubles(c, nb * V, ((nb + 255u) / 256u) * V);
}

// Range of local sites a colour phase may update, and the owned range that statistics
// and the accept counter cover.  In halo mode every phase since the last refresh makes
// two more columns at each shard-internal edge stale (their own neighbours were not
// available), so the updatable range shrinks by 2 per phase; the halo must be wide
// enough that it never reaches the owned columns.
void owned_range(const epv_ctx *c, uint64_t *lo, uint64_t *hi) {
  if (!c->halo_mode) { *lo = c->first; *hi = c->last; return; }
  *lo = c->halo_left ? c->halo_left : 1u;
  *hi = c->halo_right ? c->S.n - c->halo_right - 1u : c->S.n - 2u;
}
int phase_range(epv_ctx *c, uint64_t *lo, uint64_t *hi) {
  if (!c->halo_mode) { *lo = c->first; *hi = c->last; return EPV_OK; }
  const uint64_t shrink = 2u * (c->phases_used + 1u);
  *lo = c->halo_left ? shrink : 1u;
  *hi = c->halo_right ? c->S.n - 1u - shrink : c->S.n - 2u;
  if ((c->halo_left && *lo > c->halo_left) || (c->halo_right && *hi + c->halo_right < c->S.n - 1u))
    return fail(c, EPV_ERR_STATE, "halo exhausted: refresh the halo columns (epv_put_columns + "
                                  "epv_set_halo) before running more sweeps");
  return EPV_OK;
}

// ---- exact statistics.  The dwell times of branch b are summed as integers rint(dt * 2^k_b),
//   k_b = min(61 - e(n_global * T_b), 50 - e(T_b)),   e(x) = the frexp exponent (x < 2^e):
// a whole genome's sum stays below 2^61, a single term below 2^50 (epv_stat_fix rounds with one
// add).  oracle/epv_oracle.c (stat_scale_exp) makes the same choice independently.
int stat_scale_exp(uint64_t n_global, double T) {
  if (!(T > 0.0) || !std::isfinite(T)) return 0;
  int e_t = 0, e_nt = 0;
  (void)std::frexp(T, &e_t);
  (void)std::frexp((double)n_global * T, &e_nt);
  int k = std::min(61 - e_nt, 50 - e_t);
  return std::max(-1000, std::min(1000, k));
}
// 2^k_b of every branch on the device, refreshed when the branch lengths or the genome length changed
int ensure_stat_scale(epv_ctx *c) {
  std::vector<double> sc(c->S.N, 1.0);
  for (uint32_t b = 1; b < c->S.N; ++b) sc[b] = std::ldexp(1.0, stat_scale_exp(c->S.n_global, c->blen[b]));
  if (sc == c->statscale) return EPV_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(c->d_statscale, sc.data(), sizeof(double) * c->S.N, hipMemcpyHostToDevice));
  c->statscale = sc;
  return EPV_OK;
}
int ensure_sweep_tot(epv_ctx *c, uint64_t sweeps) {
  if (sweeps <= c->sweep_tot_cap) return EPV_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  dfree(c->d_sweep_tot);
  c->sweep_tot_cap = 0;
  HIP_TRY(c, hipMalloc(&c->d_sweep_tot, sweeps * c->S.B * 16u * sizeof(unsigned long long)));
  c->sweep_tot_cap = sweeps;
  return EPV_OK;
}
void launch_isum(epv_ctx *c, const void *in, uint64_t m, uint64_t G, uint64_t in_row_stride, uint64_t in_z_stride,
                 void *out, uint64_t out_row_stride, uint64_t out_z_stride, uint64_t n_out_rows, uint64_t Z) {
  const uint32_t V = c->S.B * 16u;
  hipLaunchKernelGGL(epv_isum_kernel, dim3(V / 16u, (unsigned)n_out_rows, (unsigned)Z), dim3(256), 0, c->stream,
                     (const unsigned long long *)in, m, V, G, in_row_stride, in_z_stride, (unsigned long long *)out,
                     out_row_stride, out_z_stride);
}
// integer J/D of nb_total blocks x Z slices ([z][block][V] at d_blocks) -> d_sweep_tot[slot0 + z][V]
int reduce_blocks_to_tot(epv_ctx *c, const void *d_blocks, uint64_t nb_total, uint64_t Z, uint64_t slot0) {
  const uint32_t V = c->S.B * 16u;
  int rc = ensure_sweep_tot(c, slot0 + Z);
  if (rc) return rc;
  if (Z > 65535u) return fail(c, EPV_ERR_ARG, "too many batch sweeps for one launch");
  unsigned long long *out = c->d_sweep_tot + slot0 * V;
  if (nb_total <= 1024u) {
    launch_isum(c, d_blocks, nb_total, 0u, V, nb_total * V, out, 0u, V, 1u, Z);
  } else {
    // two stages: groups of 64 blocks (wide grid), then their sums
    const uint64_t m1 = (nb_total + 63u) / 64u;
    if (m1 > 65535u) return fail(c, EPV_ERR_ARG, "more than 2^22 blocks per context");
    if (Z * m1 * V > c->lvl_cap) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      dfree(c->d_lvl);
      c->lvl_cap = 0;
      HIP_TRY(c, hipMalloc(&c->d_lvl, Z * m1 * V * sizeof(double)));
      c->lvl_cap = Z * m1 * V;
    }
    launch_isum(c, d_blocks, nb_total, 64u, V, nb_total * V, c->d_lvl, V, m1 * V, m1, Z);
    launch_isum(c, c->d_lvl, m1, 0u, V, m1 * V, out, 0u, V, 1u, Z);
  }
  HIP_TRY(c, hipGetLastError());
  return EPV_OK;
}
// the integer totals of `batch` sweeps (d_sweep_tot[0 .. batch)) -> J, D as run_mcmc returns them
// (SingleSiteSampler.cpp:576-594): every sweep's statistics become doubles -- J = the count,
// D = the integer * 2^-k_b, exact but for the one rounding of int64 -> double -- and are added up
// sweep by sweep in fp64 like J_all_sites += J_one_site, then divided by the batch size
int finish_stats(epv_ctx *c, uint64_t batch, int average, double *J, double *D) {
  const uint32_t V = c->S.B * 16u;
  std::vector<long long> tot(batch * V);
  HIP_TRY(c, hipMemcpyAsync(tot.data(), c->d_sweep_tot, batch * V * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const double nb = average ? (double)batch : 1.0;
  for (uint32_t b = 0; b < c->S.B; ++b) {
    const double inv = 1.0 / c->statscale[b + 1u];     // a power of two: exact
    for (int k = 0; k < 8; ++k) {
      double aj = 0.0, ad = 0.0;
      for (uint64_t w = 0; w < batch; ++w) {
        aj += (double)tot[w * V + b * 16u + k];
        ad += (double)tot[w * V + b * 16u + 8u + k] * inv;
      }
      J[b * 8 + k] = aj / nb;
      D[b * 8 + k] = ad / nb;
    }
  }
  return EPV_OK;
}

// integer J/D of the current paths over the owned range into d_sweep_tot[slot]
int launch_suffstats(epv_ctx *c, uint64_t slot) {
  int rc = ensure_partials(c);
  if (rc) return rc;
  if ((rc = ensure_stat_scale(c))) return rc;
  const uint64_t nb = (c->S.n + 255u) / 256u;
  uint64_t own_lo = 0, own_hi = 0;
  owned_range(c, &own_lo, &own_hi);
  hipLaunchKernelGGL(epv_suffstat_kernel, dim3((unsigned)nb, (c->S.B + EPV_STAT_BCH - 1u) / EPV_STAT_BCH), dim3(256), 0,
                     c->stream, c->S, own_lo, own_hi, (uint64_t)0, c->d_statscale, (unsigned long long *)c->d_partial[0]);
  return reduce_blocks_to_tot(c, c->d_partial[0], nb, 1u, slot);
}

// the global-memory slab of a proposal kernel, allocated when a launch first takes that kernel (a
// context on a large tree plans three kernels but runs one: 5 - 50 GB each at full size)
int ensure_slab(epv_ctx *c, double **slab, uint64_t *cap, uint64_t need) {
  if (need <= *cap) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  dfree(*slab);
  *cap = 0;
  HIP_TRY(c, hipMalloc(slab, need * sizeof(double)));
  *cap = need;
  return EPV_OK;
}

int launch_phase(epv_ctx *c, int colour, uint64_t seed, uint32_t sweep) {
  uint64_t first = 0, last = 0, own_lo = 0, own_hi = 0;
  int prc = phase_range(c, &first, &last);
  if (prc) return prc;
  owned_range(c, &own_lo, &own_hi);
  const uint64_t span = last - first + 1u;
  const uint64_t threads = (span + 2u) / 3u;
  // first local site of this colour (the kernels derive the same value)
  const uint64_t s0 = first + (((uint32_t)colour + 3u - (uint32_t)((c->S.g0 + first) % 3u)) % 3u);
  const uint64_t blocks = (threads + c->mh_threads - 1u) / c->mh_threads;
  if (blocks == 0) return EPV_OK;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  c->timing = c->timing_every && (c->timing_seen++ % c->timing_every) == 0u;
  if (c->timing) {
    if (c->ev_used == c->ev_pool.size()) {
      hipEvent_t a, b;
      HIP_TRY(c, hipEventCreate(&a));
      HIP_TRY(c, hipEventCreate(&b));
      c->ev_pool.emplace_back(a, b);
    }
    e0 = c->ev_pool[c->ev_used].first;
    e1 = c->ev_pool[c->ev_used].second;
    ++c->ev_used;
    HIP_TRY(c, hipEventRecord(e0, c->stream));
  }
  // (root resampling changes the proposal's normalising constant with the path: the ratio must be evaluated)
  const bool refq = c->S.flags & (EPV_FLAG_REFERENCE_PROPOSAL_RATIO | EPV_FLAG_SAMPLE_ROOT);
  // the reference-arithmetic mode keeps the first kernel, and so do trees whose record pool does
  // not fit LDS: with the pool in global memory the second kernel's extra passes over it cost
  // more than its dense evaluation saves (16-leaf tree: 830 vs 676 us);
  // EPV_PROPOSE_V2_GLOBAL=1 forces it for A/B runs
  static const bool p2_global = std::getenv("EPV_PROPOSE_V2_GLOBAL") != nullptr;
  const bool p3 = c->p3 && !refq;
  const bool p2 = !p3 && c->use_p2 && !refq && (!c->p2_gpool || p2_global);
  uint32_t list_mode = 0;
  // segment-parallel jumps pay on long branches (single branch T = 1: +17 %, every segment is
  // dirty and needs several trials) and cost on short ones (tree.nwk: -12 %, one dirty segment in
  // fourteen branches does not repay the extra hand-over): profiles/r02_ab_seg_jumps.txt
  static const bool no_cache = std::getenv("EPV_ACCEPT_NO_CACHE") != nullptr;
  const uint32_t meta_cache = (c->S.B <= 8u && !no_cache) ? 1u : 0u;
  if (p2 && c->fused) {
    // the whole phase in one kernel, one wave per 64 sites (see epv_propose2.h)
    const int frc = ensure_fused_buffers(c);
    if (frc) return frc;
    const unsigned pt = 64u * c->p2_waves, per_block = c->fused_lanes * c->p2_waves;
    const unsigned pb = (unsigned)((threads + per_block - 1u) / per_block);
    if ((uint64_t)pb * c->p2_waves > c->fused_waves) return fail(c, EPV_ERR_STATE, "fused phase: launch larger than its lists");
    EpvFused F = c->F;
    F.meta_cache = meta_cache;
    F.lanes = c->fused_lanes;
    static const int grouped = std::getenv("EPV_FUSED_GROUPED_ROUNDS") ? std::atoi(std::getenv("EPV_FUSED_GROUPED_ROUNDS")) : 4;
    F.grouped_rounds = (uint32_t)std::max(0, grouped);
    hipLaunchKernelGGL((epv_mh_propose2_kernel<false, true, true>), dim3(pb), dim3(pt), c->p2_lds, c->stream, c->S,
                       (uint32_t)colour, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, first, last, own_lo, own_hi,
                       c->p2_pool, c->p2_list_cap, 0u, c->d_counters, (double *)nullptr, c->d_segtab, F);
    if (c->timing) HIP_TRY(c, hipEventRecord(e1, c->stream));
    HIP_TRY(c, hipGetLastError());
    if (c->halo_mode) ++c->phases_used;
    return EPV_OK;
  }
  const uint32_t seg_mode = (p2 && seg_jumps_on(c)) ? 1u : 0u;
  if (seg_mode) { const int src = ensure_seg_buffers(c); if (src) return src; }
  if (p3) {
    // large tree: a 16-bit word per (node, lane) in LDS, q rows and heavy records in a slab (epv_propose3.h)
    list_mode = 1u + (c->phase_parity & 1u);
    const unsigned pb = (unsigned)((threads + 255u) / 256u);
    { const int rc3 = ensure_slab(c, &c->d_gpool3, &c->gpool3_cap, c->gpool3_need); if (rc3) return rc3; }
    auto k3 = c->S.N > 64u ? epv_mh_propose3_kernel<2> : epv_mh_propose3_kernel<1>;
    hipLaunchKernelGGL(k3, dim3(pb), dim3(256), c->p3_lds, c->stream, c->S, (uint32_t)colour,
                       (uint32_t)seed, (uint32_t)(seed >> 32), sweep, first, last, own_lo, own_hi, c->p3_list_cap,
                       c->p3_qrows, c->p3_nup, c->p3_depth, c->phase_parity & 1u, c->d_counters, c->d_gpool3, c->d_segtab,
                       c->d_nodetab, c->d_slabflags, c->p3_slots);
    ++c->phase_parity;
  } else if (p2) {
    list_mode = 1u + (c->phase_parity & 1u);
    const unsigned pt = 64u * c->p2_waves, pb = (unsigned)((threads + pt - 1u) / pt);
    if (c->p2_gpool) { const int rc2 = ensure_slab(c, &c->d_gpool2, &c->gpool2_cap, c->gpool2_need); if (rc2) return rc2; }
    auto kern = c->p2_gpool ? (seg_mode ? epv_mh_propose2_kernel<true, true, false> : epv_mh_propose2_kernel<true, false, false>)
                            : (seg_mode ? epv_mh_propose2_kernel<false, true, false> : epv_mh_propose2_kernel<false, false, false>);
    hipLaunchKernelGGL(kern, dim3(pb), dim3(pt), c->p2_lds, c->stream, c->S, (uint32_t)colour, (uint32_t)seed,
                       (uint32_t)(seed >> 32), sweep, first, last, own_lo, own_hi, c->p2_pool, c->p2_list_cap,
                       c->phase_parity & 1u, c->d_counters, c->p2_gpool ? c->d_gpool2 : (double *)nullptr, c->d_segtab,
                       EpvFused{});
    ++c->phase_parity;
  } else {
    if (c->mh_gpool) { const int rc1 = ensure_slab(c, &c->d_gpool, &c->gpool_cap, c->gpool_need); if (rc1) return rc1; }
    auto kern = c->mh_gpool ? (refq ? epv_mh_propose_kernel<true, true> : epv_mh_propose_kernel<true, false>)
                            : (refq ? epv_mh_propose_kernel<false, true> : epv_mh_propose_kernel<false, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(c->mh_threads), c->mh_lds, c->stream, c->S,
                       (uint32_t)colour, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, first, last,
                       c->pool_entries, c->d_counters, c->mh_gpool ? c->d_gpool : (double *)nullptr);
  }
  if (seg_mode) {
    // dirty segments one lane each, then their branches one lane each; both lists are sized on
    // the device, the grids cover a quarter / an eighth of the capacity and stride over the rest
    const unsigned sx = (unsigned)std::min<uint64_t>(32u, std::max<uint64_t>(1u, (c->S.seg_cap / 4u + 255u) / 256u));
    const unsigned bx = (unsigned)std::min<uint64_t>(16u, std::max<uint64_t>(1u, (c->S.btask_cap / 8u + 255u) / 256u));
    hipLaunchKernelGGL(epv_seg_search_kernel, dim3(sx, EPV_SHARDS), dim3(256), const_lds_bytes(c->S.N), c->stream,
                       c->S, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, c->d_counters);
    hipLaunchKernelGGL(epv_seg_assemble_kernel, dim3(bx, EPV_SHARDS), dim3(256), const_lds_bytes(c->S.N), c->stream,
                       c->S, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, s0, c->d_counters);
    // the sequential kernel behind them with a minimal grid: branches of more than 64 segments
    // and whatever did not fit the lists (normally nothing: it reads empty lists and leaves)
    hipLaunchKernelGGL(epv_mh_jumps_kernel, dim3(1, EPV_SHARDS), dim3(256), const_lds_bytes(c->S.N), c->stream, c->S,
                       (uint32_t)seed, (uint32_t)(seed >> 32), sweep, 32u, s0, 0.0, 0.0, c->d_counters);
  } else {
    // one lane per dirty (site, branch) pair; the count is only known on the device, so
    // launch a grid that covers the typical case and grid-stride over the rest
    const uint64_t max_tasks = blocks / EPV_SHARDS * 64u * c->S.B + 64u * c->S.B;  // per shard
    // lanes of a wave that own a task (the others only help in the cooperative search): full
    // waves when there is work for every SIMD, fewer tasks per wave -- a shorter critical path --
    // on a small genome (measured: tools/ab_envbench.py EPV_TASKS_PER_WAVE)
    uint32_t tpw = c->tasks_per_wave;
    if (tpw == 0) {
      const double est = (double)threads * c->S.B * std::min(1.0, 0.1 + c->kbar) / 2048.0;
      tpw = est >= 64.0 ? 64u : est >= 32.0 ? 32u : est >= 16.0 ? 16u : 8u;
    }
    // the one-segment tasks (the first bucket: ~95 % on short branches) in their own lean kernel; the general
    // one then takes the rest with a grid sized for it.  (Forward-rejection mode keeps the general kernel for
    // everything: a flip there can need 1e5 trials, which only the wave-wide search takes in reasonable time.)
    static const int j1_env = std::getenv("EPV_JUMPS1") ? std::atoi(std::getenv("EPV_JUMPS1")) : 1;
    const bool j1 = j1_env != 0 && !(c->S.flags & EPV_FLAG_FORWARD_REJECTION);
    // a block (4 waves) takes 4*tpw tasks per pass; size the grid for ~1/4 of the worst case
    if (j1) {
      const uint64_t j1b = std::min<uint64_t>((max_tasks / 4u + 255u) / 256u + 1u, 256u);
      const uint64_t jgb = std::min<uint64_t>((max_tasks / 16u + 4u * tpw - 1u) / (4u * tpw) + 1u, 64u);
      hipLaunchKernelGGL(epv_mh_jumps_all_kernel, dim3(EPV_SHARDS, (unsigned)(jgb + j1b)), dim3(256), const_lds_bytes(c->S.N),
                         c->stream, c->S, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, tpw, s0, c->d_counters, (uint32_t)jgb);
    } else {
      const uint64_t jb = std::min<uint64_t>((max_tasks / 4u + 4u * tpw - 1u) / (4u * tpw) + 1u, 256u);
      hipLaunchKernelGGL(epv_mh_jumps_kernel, dim3((unsigned)jb, EPV_SHARDS), dim3(256), const_lds_bytes(c->S.N),
                         c->stream, c->S, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, tpw, s0, 0.0, 0.0, c->d_counters, 0u);
    }
  }
  // meta cache of the accept kernel: 5 columns x B words per lane in LDS while that stays small
  // (B <= 8: 20 KB per block next to the 24 KB of accumulators)
  const size_t acc_lds = const_lds_bytes(c->S.N) + (meta_cache ? (size_t)5u * c->S.B * 256u * sizeof(epv_meta_t) : 0u);
  if (list_mode) {
    // the listed sites (proposal differs from the current path) per shard: typically ~30 % of the
    // colour; the grid covers half of the worst case and strides over the rest
    // (on a large tree nearly every site is listed -- one clean proposal in thirty branches is rare --
    // and a block that strides twice runs two of the kernel's long dependent chains back to back)
    const uint64_t per_shard = (threads + EPV_SHARDS - 1u) / EPV_SHARDS;
    static const int acc_full = std::getenv("EPV_ACCEPT_FULL_GRID") ? std::atoi(std::getenv("EPV_ACCEPT_FULL_GRID")) : -1;
    const bool full = acc_full >= 0 ? acc_full != 0 : c->S.B > 8u;
    const unsigned ax = (unsigned)std::max<uint64_t>(1u, ((full ? per_shard : per_shard / 2u) + 255u) / 256u);
    // large trees (no room for the meta cache): a lane per (site, triple), branches in groups (epv_accept3.h)
    static const int acc_v3 = std::getenv("EPV_ACCEPT_V3") ? std::atoi(std::getenv("EPV_ACCEPT_V3")) : -1;
    if (acc_v3 >= 0 ? acc_v3 != 0 : !meta_cache) {
      const unsigned ax3 = (unsigned)std::max<uint64_t>(1u, (per_shard + 4u * EPV_ACC3_SITES - 1u) / (4u * EPV_ACC3_SITES));
      hipLaunchKernelGGL(epv_mh_accept3_kernel, dim3(ax3, EPV_SHARDS), dim3(256), const_lds_bytes(c->S.N), c->stream,
                         c->S, (uint32_t)colour, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, first, last, own_lo,
                         own_hi, c->d_counters, list_mode, (uint64_t)0);
    } else
    hipLaunchKernelGGL(epv_mh_accept_kernel, dim3(ax, EPV_SHARDS), dim3(256), acc_lds, c->stream,
                       c->S, (uint32_t)colour, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, first, last, own_lo,
                       own_hi, c->d_counters, list_mode, meta_cache);
  } else if (static const int acc_v3b = std::getenv("EPV_ACCEPT_V3") ? std::atoi(std::getenv("EPV_ACCEPT_V3")) : -1;
             acc_v3b >= 0 ? acc_v3b != 0 : !meta_cache) {
    // every site of the colour (reference proposal arithmetic on a large tree), a lane per (site, triple)
    const unsigned ax3 = (unsigned)std::max<uint64_t>(1u, (threads + 4u * EPV_ACC3_SITES - 1u) / (4u * EPV_ACC3_SITES));
    hipLaunchKernelGGL(epv_mh_accept3_kernel, dim3(ax3, 1), dim3(256), const_lds_bytes(c->S.N), c->stream,
                       c->S, (uint32_t)colour, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, first, last, own_lo,
                       own_hi, c->d_counters, 0u, (uint64_t)threads);
  } else {
    hipLaunchKernelGGL(epv_mh_accept_kernel, dim3((unsigned)((threads + 255u) / 256u)), dim3(256),
                       acc_lds, c->stream, c->S, (uint32_t)colour, (uint32_t)seed,
                       (uint32_t)(seed >> 32), sweep, first, last, own_lo, own_hi, c->d_counters, 0u, meta_cache);
  }
  if (c->timing) HIP_TRY(c, hipEventRecord(e1, c->stream));
  HIP_TRY(c, hipGetLastError());
  if (c->halo_mode) ++c->phases_used;
  return EPV_OK;
}

// fold finished timing events into the running totals (stream must be idle)
int drain_timing(epv_ctx *c) {
  for (size_t i = 0; i < c->ev_used; ++i) {
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second));
    c->timed_ms += ms;
    ++c->timed_launches;
  }
  c->ev_used = 0;
  return EPV_OK;
}

int read_counters(epv_ctx *c, unsigned long long out[EPV_CNT_N]) {
  unsigned long long *raw = c->h_counters;
  HIP_TRY(c, hipMemcpyAsync(raw, c->d_counters, sizeof(unsigned long long) * EPV_CNT_WORDS,
                            hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (uint32_t k = 0; k < EPV_CNT_N; ++k) {
    out[k] = 0;
    for (uint32_t s = 0; s < EPV_SHARDS; ++s) out[k] += raw[EPV_CNT_IDX(k, s)];
  }
  return drain_timing(c);
}

int check_ready(epv_ctx *c, bool need_reset) {
  if (!c) return EPV_ERR_ARG;
  if (!c->have_tree || !c->have_model || !c->have_paths)
    return fail(c, EPV_ERR_STATE, "tree, model and paths must be set first");
  if (need_reset && !c->have_reset) return fail(c, EPV_ERR_STATE, "epv_reset has not been called");
  return EPV_OK;
}

}  // namespace

EPV_API epv_ctx *epv_create(int device_id) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count) return nullptr;
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  epv_ctx *c = new epv_ctx();
  c->device = device_id;
  if (const char *e = std::getenv("EPV_PROPOSE_V1")) c->use_p2 = std::atoi(e) == 0;
  if (const char *e = std::getenv("EPV_PROPOSE_V3")) c->use_p3 = std::atoi(e) != 0 ? 1 : 0;
  if (const char *e = std::getenv("EPV_SEG_JUMPS")) c->use_seg = std::atoi(e) != 0 ? 1 : 0;
  if (const char *e = std::getenv("EPV_FUSED_PHASE")) c->use_fused = std::atoi(e) != 0 ? 1 : 0;
  if (const char *e = std::getenv("EPV_FUSED_MAX_WAVES")) { const long v = std::atol(e); if (v >= 0) c->fused_max_waves = (uint32_t)v; }
  if (const char *e = std::getenv("EPV_TASKS_PER_WAVE")) {  // tuning knob
    const int v = std::atoi(e);
    if (v >= 1 && v <= 64) c->tasks_per_wave = (uint32_t)v;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&c->d_model, sizeof(EpvModelConst)) != hipSuccess ||
      hipMalloc(&c->d_counters, sizeof(unsigned long long) * EPV_CNT_WORDS) != hipSuccess ||
      hipHostMalloc(&c->h_counters, sizeof(unsigned long long) * EPV_CNT_WORDS) != hipSuccess ||
      hipMalloc(&c->d_cnt_snap, sizeof(unsigned long long) * EPV_CNT_WORDS) != hipSuccess ||
      hipHostMalloc(&c->h_cnt_snap, sizeof(unsigned long long) * EPV_CNT_WORDS) != hipSuccess ||
      hipMemset(c->d_counters, 0, sizeof(unsigned long long) * EPV_CNT_WORDS) != hipSuccess) {
    delete c;
    return nullptr;
  }
  // the MH kernel asks for more dynamic LDS than the 64 KiB default
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose3_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);   // (it has static LDS too)
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose3_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose2_kernel<false, false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose2_kernel<false, true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose2_kernel<false, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose2_kernel<true, false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose2_kernel<true, true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose_kernel<false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose_kernel<true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose_kernel<false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(epv_mh_propose_kernel<true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return c;
}

EPV_API void epv_destroy(epv_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_paths(c);
  dfree(c->d_model); dfree(c->d_parent); dfree(c->d_subtree); dfree(c->d_blen);
  dfree(c->d_counters); dfree(c->d_sweep_tot); dfree(c->d_statscale); dfree(c->d_scale); dfree(c->d_indep); dfree(c->d_gpool); dfree(c->d_stage); dfree(c->d_lvl); dfree(c->d_rows); dfree(c->d_gpool2); dfree(c->d_gpool3); dfree(c->d_segtab); dfree(c->d_nodetab); dfree(c->d_slabflags);
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  if (c->h_cnt_snap) (void)hipHostFree(c->h_cnt_snap);
  for (hipEvent_t &e : c->ev_copy) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  dfree(c->d_cnt_snap);
  for (auto &p : c->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  (void)hipStreamDestroy(c->stream);
  delete c;
}

EPV_API const char *epv_last_error(const epv_ctx *c) { return c ? c->err.c_str() : "null context"; }

EPV_API int epv_set_tree(epv_ctx *c, int n_nodes, const uint32_t *parent_ids,
                         const uint32_t *subtree_sizes, const double *branches) {
  if (!c) return EPV_ERR_ARG;
  if (n_nodes < 2 || n_nodes > 4095 || !parent_ids || !subtree_sizes || !branches)
    return fail(c, EPV_ERR_ARG, "bad tree");
  if (c->have_paths && (uint32_t)n_nodes != c->S.N)
    return fail(c, EPV_ERR_ARG, "tree size differs from the uploaded paths");
  for (int i = 1; i < n_nodes; ++i)
    if (parent_ids[i] >= (uint32_t)i || subtree_sizes[i] < 1 || i + subtree_sizes[i] > (uint32_t)n_nodes ||
        !(branches[i] > 0.0))
      return fail(c, EPV_ERR_ARG, "tree arrays are not a valid pre-order tree with positive branches");
  HIP_TRY(c, hipSetDevice(c->device));
  c->parent.assign(parent_ids, parent_ids + n_nodes);
  c->subtree.assign(subtree_sizes, subtree_sizes + n_nodes);
  c->blen.assign(branches, branches + n_nodes);
  dfree(c->d_parent); dfree(c->d_subtree); dfree(c->d_blen); dfree(c->d_statscale); dfree(c->d_scale);
  c->statscale.clear();
  HIP_TRY(c, hipMalloc(&c->d_parent, sizeof(uint32_t) * n_nodes));
  HIP_TRY(c, hipMalloc(&c->d_subtree, sizeof(uint32_t) * n_nodes));
  HIP_TRY(c, hipMalloc(&c->d_blen, sizeof(double) * n_nodes));
  HIP_TRY(c, hipMalloc(&c->d_statscale, sizeof(double) * n_nodes));
  HIP_TRY(c, hipMalloc(&c->d_scale, sizeof(double) * n_nodes));
  HIP_TRY(c, hipMemcpy(c->d_parent, parent_ids, sizeof(uint32_t) * n_nodes, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_subtree, subtree_sizes, sizeof(uint32_t) * n_nodes, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_blen, branches, sizeof(double) * n_nodes, hipMemcpyHostToDevice));
  c->S.N = (uint32_t)n_nodes;
  c->S.B = (uint32_t)n_nodes - 1u;
  c->S.parent = c->d_parent;
  c->S.subtree = c->d_subtree;
  c->S.blen = c->d_blen;
  c->have_tree = true;
  c->have_reset = false;
  if (c->have_paths) { int rc = plan_mh(c); if (!rc) rc = plan_p2(c); return rc ? rc : plan_p3(c); }
  return EPV_OK;
}

EPV_API int epv_set_model(epv_ctx *c, const double *triplet_rates, const double *T) {
  if (!c) return EPV_ERR_ARG;
  if (!triplet_rates || !T) return fail(c, EPV_ERR_ARG, "null model");
  for (int i = 0; i < 8; ++i) {
    if (!(triplet_rates[i] > 0.0)) return fail(c, EPV_ERR_ARG, "triplet rates must be positive");
    c->model.rates[i] = triplet_rates[i];
    c->model.log_rates[i] = std::log(triplet_rates[i]);  // SingleSiteSampler.cpp:464-468
  }
  for (int i = 0; i < 4; ++i) c->model.T[i] = T[i];
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // (a kernel of epv_reset_async may still read the old constants)
  HIP_TRY(c, hipMemcpy(c->d_model, &c->model, sizeof(EpvModelConst), hipMemcpyHostToDevice));
  c->S.model = c->d_model;
  c->have_model = true;
  c->have_reset = false;
  return EPV_OK;
}

// device storage of n_sites x B paths with `capacity` jump slots each (both buffers), the phase
// hand-over arrays and the work lists; the paths themselves are filled in by the caller
static int alloc_paths(epv_ctx *c, uint64_t n_sites, uint32_t capacity, uint64_t global_site_offset) {
  const uint64_t B = c->S.B, E = B * n_sites;
  HIP_TRY(c, hipSetDevice(c->device));
  free_paths(c);
  c->S.n = n_sites;
  c->S.g0 = global_site_offset;
  c->S.n_global = global_site_offset + n_sites;
  c->S.C = capacity;
  HIP_TRY(c, hipMalloc(&c->S.meta, 2u * E * sizeof(epv_meta_t)));
  HIP_TRY(c, hipMalloc(&c->S.jumps, 2u * E * capacity * sizeof(double)));
  HIP_TRY(c, hipMalloc(&c->S.sel, n_sites));
  HIP_TRY(c, hipMalloc(&c->S.tri, n_sites * sizeof(double)));
  c->S.phase_cap = (n_sites + 2u) / 3u + 1u;
  HIP_TRY(c, hipMalloc(&c->S.prop_llr, c->S.phase_cap * sizeof(double)));
  HIP_TRY(c, hipMalloc(&c->S.prop_flag, c->S.phase_cap));
  c->S.W = (2u * capacity + 1u + 63u) / 64u;
  HIP_TRY(c, hipMalloc(&c->S.prop_states, B * c->S.phase_cap * c->S.W * sizeof(uint64_t)));
  // one task region per counter shard, sized for the worst case of the blocks that use it
  c->S.task_cap = ((((n_sites + 2u) / 3u + 63u) / 64u + EPV_SHARDS - 1u) / EPV_SHARDS + 1u) * 64u * B;
  HIP_TRY(c, hipMalloc(&c->S.tasks, c->S.task_cap * EPV_SHARDS * 2u * sizeof(unsigned long long)));
  // segment-parallel jump sampling: branch list as large as the task regions, segment list for
  // the expected number of dirty segments with a wide margin (what does not fit falls back to the
  // sequential kernel's lists)
  //   (allocated by ensure_seg_buffers when that path is first used)
  c->S.btask_cap = c->S.seg_cap = 0;
  // accept list: one region per counter shard, room for every site of the blocks that use it
  c->S.alist_cap = ((((n_sites + 2u) / 3u + 63u) / 64u + EPV_SHARDS - 1u) / EPV_SHARDS + 1u) * 64u;
  HIP_TRY(c, hipMalloc(&c->S.alist, c->S.alist_cap * EPV_SHARDS * sizeof(uint32_t)));
  HIP_TRY(c, hipMemsetAsync(c->S.meta, 0, 2u * E * sizeof(epv_meta_t), c->stream));
  c->first = 1;
  c->last = n_sites - 2;
  c->halo_mode = false;
  c->halo_left = c->halo_right = 0;
  c->phases_used = 0;
  return EPV_OK;
}

EPV_API int epv_upload_paths(epv_ctx *c, uint64_t n_sites, const uint8_t *init_state,
                             const uint64_t *offsets, const double *jumps, uint32_t capacity,
                             uint64_t global_site_offset) {
  if (!c) return EPV_ERR_ARG;
  if (!c->have_tree) return fail(c, EPV_ERR_STATE, "epv_set_tree must come before epv_upload_paths");
  if (n_sites < 3 || !init_state || !offsets) return fail(c, EPV_ERR_ARG, "bad paths");
  if (global_site_offset + n_sites > 0xffffffffull)
    return fail(c, EPV_ERR_ARG, "site indices must fit 32 bits (Philox counter word)");
  const uint64_t B = c->S.B, E = B * n_sites;
  uint64_t maxj = 0;
  for (uint64_t e = 0; e < E; ++e) {
    if (offsets[e + 1] < offsets[e]) return fail(c, EPV_ERR_ARG, "offsets must be non-decreasing");
    maxj = std::max<uint64_t>(maxj, offsets[e + 1] - offsets[e]);
  }
  if (capacity == 0) capacity = (uint32_t)std::max<uint64_t>(16u, 2u * maxj + 8u);
  if (capacity > EPV_MAX_CAP) capacity = EPV_MAX_CAP;
  if (maxj > capacity) return fail(c, EPV_ERR_CAPACITY, "an input path has more jumps than the capacity");
  int arc = alloc_paths(c, n_sites, capacity, global_site_offset);
  if (arc) return arc;
  c->kbar = E ? (double)offsets[E] / (double)E : 0.0;
  // staging of the CSR form
  DevTmp<uint8_t> d_init;
  DevTmp<uint64_t> d_off;
  DevTmp<double> d_j;
  const uint64_t tot = offsets[E];
  HIP_TRY(c, d_init.alloc(E));
  HIP_TRY(c, d_off.alloc(E + 1));
  HIP_TRY(c, d_j.alloc(std::max<uint64_t>(tot, 1)));
  HIP_TRY(c, hipMemcpyAsync(d_init.p, init_state, E, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_off.p, offsets, (E + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  if (tot) HIP_TRY(c, hipMemcpyAsync(d_j.p, jumps, tot * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(epv_scatter_kernel, dim3((unsigned)((E + 255u) / 256u)), dim3(256), 0, c->stream,
                     c->S, d_init.p, d_off.p, d_j.p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_paths = true;
  c->have_reset = false;
  { int rc = plan_mh(c); if (!rc) rc = plan_p2(c); return rc ? rc : plan_p3(c); }
}

// epievo_sim's forward simulation on the device (epv_forward.h): root sequence (given, or
// EpiEvoModel::sample_state_sequence with keyed uniforms), then every branch in pre-order by
// site-parallel thinning.  The histories end up resident like uploaded paths.
EPV_API int epv_forward_simulate(epv_ctx *c, uint64_t n_sites, const uint8_t *root_states, uint64_t seed,
                                 uint32_t capacity, uint64_t *total_jumps) {
  if (!c) return EPV_ERR_ARG;
  if (!c->have_tree || !c->have_model)
    return fail(c, EPV_ERR_STATE, "epv_set_tree and epv_set_model must come before epv_forward_simulate");
  if (n_sites < 3 || n_sites > 0xffffffffull) return fail(c, EPV_ERR_ARG, "bad number of sites");
  if (capacity == 0) capacity = 16u;
  if (capacity > EPV_MAX_CAP) capacity = EPV_MAX_CAP;
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = alloc_paths(c, n_sites, capacity, 0);
  if (rc) return rc;
  const uint64_t n = n_sites, N = c->S.N;
  const uint32_t seed_lo = (uint32_t)seed, seed_hi = (uint32_t)(seed >> 32);
  EpvFwd F{};
  DevTmp<uint8_t> st0, st1, endv, agg, prefix;
  DevTmp<uint32_t> k0, k1;
  DevTmp<double> t0, t1;
  DevTmp<unsigned long long> info;
  HIP_TRY(c, st0.alloc(n)); HIP_TRY(c, st1.alloc(n));
  HIP_TRY(c, k0.alloc(n)); HIP_TRY(c, k1.alloc(n));
  HIP_TRY(c, t0.alloc(n)); HIP_TRY(c, t1.alloc(n));
  HIP_TRY(c, endv.alloc(N * n));
  HIP_TRY(c, info.alloc(3));
  F.st[0] = st0.p; F.st[1] = st1.p; F.k[0] = k0.p; F.k[1] = k1.p; F.t[0] = t0.p; F.t[1] = t1.p; F.end = endv.p;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const auto t_alloc = std::chrono::steady_clock::now();
  F.lam_max = c->model.rates[0];
  for (int i = 1; i < 8; ++i) F.lam_max = std::max(F.lam_max, c->model.rates[i]);
  for (int i = 0; i < 8; ++i) F.pacc[i] = c->model.rates[i] / F.lam_max;
  if (root_states) {
    HIP_TRY(c, hipMemcpyAsync(F.end, root_states, n, hipMemcpyHostToDevice, c->stream));
  } else {
    const uint64_t per_block = 256u * EPV_ROOT_PER_THREAD, nb = (n + per_block - 1u) / per_block;
    HIP_TRY(c, agg.alloc(nb)); HIP_TRY(c, prefix.alloc(nb));
    const double T00 = c->model.T[0], T11 = c->model.T[3], pi1 = (1.0 - T00) / (2.0 - T11 - T00);   // EpiEvoModel.cpp:289
    hipLaunchKernelGGL(epv_fwd_root_kernel, dim3((unsigned)nb), dim3(256), 0, c->stream, n, (uint64_t)0, seed_lo, seed_hi,
                       T00, T11, pi1, 0u, agg.p, (const uint8_t *)nullptr, (uint8_t *)nullptr);
    hipLaunchKernelGGL(epv_fwd_root_scan_kernel, dim3(1), dim3(64), 0, c->stream, agg.p, nb, prefix.p);
    hipLaunchKernelGGL(epv_fwd_root_kernel, dim3((unsigned)nb), dim3(256), 0, c->stream, n, (uint64_t)0, seed_lo, seed_hi,
                       T00, T11, pi1, 1u, agg.p, prefix.p, F.end);
  }
  HIP_TRY(c, hipGetLastError());
  // tiles of EPV_FWD_THREADS sites with `halo` redundant ones on each side, `rounds` rounds a launch
  const uint32_t halo = 16u, rounds = 64u, own_w = EPV_FWD_THREADS - 2u * halo;
  const unsigned tiles = (unsigned)((n + own_w - 1u) / own_w);
  unsigned long long h_info[3] = {0ull, 0ull, 0ull};
  uint64_t tot = 0;
  for (uint32_t node = 1; node < N; ++node) {
    uint32_t p = 0u;
    hipLaunchKernelGGL(epv_fwd_begin_kernel, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, c->stream, c->S, F, node,
                       c->parent[node], seed_lo, seed_hi, p);
    for (uint32_t launch = 0;; ++launch) {
      HIP_TRY(c, hipMemsetAsync(info.p, 0, 3 * sizeof(unsigned long long), c->stream));
      hipLaunchKernelGGL(epv_fwd_rounds_kernel, dim3(tiles), dim3(EPV_FWD_THREADS), 0, c->stream, c->S, F, node,
                         c->blen[node], seed_lo, seed_hi, p, halo, rounds, info.p);
      HIP_TRY(c, hipGetLastError());
      HIP_TRY(c, hipMemcpyAsync(h_info, info.p, sizeof h_info, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      p ^= 1u;
      tot += h_info[2];
      if (h_info[1]) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "forward simulation: %llu paths of node %u need more than %u jump slots; "
                      "call again with a larger capacity", h_info[1], node, capacity);
        free_paths(c);
        return fail(c, EPV_ERR_CAPACITY, buf);
      }
      if (h_info[0] == 0ull) break;
      if (launch > 100000u) { free_paths(c); return fail(c, EPV_ERR_STATE, "forward simulation does not terminate"); }
    }
  }
  c->have_paths = true;
  c->have_reset = false;
  if (total_jumps) *total_jumps = tot;
  c->kbar = (double)tot / (double)(c->S.B * n);
  const auto t_end = std::chrono::steady_clock::now();
  c->fwd_alloc_ms = std::chrono::duration<double, std::milli>(t_alloc - t_begin).count();
  c->fwd_sim_ms = std::chrono::duration<double, std::milli>(t_end - t_alloc).count();
  { int prc = plan_mh(c); if (!prc) prc = plan_p2(c); return prc ? prc : plan_p3(c); }
}

EPV_API int epv_forward_last_ms(epv_ctx *c, double *alloc_ms, double *simulate_ms) {
  if (!c || !alloc_ms || !simulate_ms) return EPV_ERR_ARG;
  *alloc_ms = c->fwd_alloc_ms;
  *simulate_ms = c->fwd_sim_ms;
  return EPV_OK;
}

static int finish_mcmc(epv_ctx *c, uint64_t *n_accepted, uint64_t acc_base);

EPV_API int epv_set_options(epv_ctx *c, uint32_t flags) {
  if (!c) return EPV_ERR_ARG;
  if (flags & ~(uint32_t)(EPV_OPT_REFERENCE_PROPOSAL_RATIO | EPV_OPT_FORWARD_REJECTION | EPV_OPT_SAMPLE_ROOT))
    return fail(c, EPV_ERR_ARG, "unknown option bits");
  static_assert(EPV_OPT_REFERENCE_PROPOSAL_RATIO == EPV_FLAG_REFERENCE_PROPOSAL_RATIO &&
                EPV_OPT_FORWARD_REJECTION == EPV_FLAG_FORWARD_REJECTION && EPV_OPT_SAMPLE_ROOT == EPV_FLAG_SAMPLE_ROOT,
                "option bits");
  c->S.flags = flags;
  return EPV_OK;
}
EPV_API int epv_get_options(epv_ctx *c, uint32_t *flags) {
  if (!c || !flags) return EPV_ERR_ARG;
  *flags = c->S.flags;
  return EPV_OK;
}

EPV_API int epv_phase_mode(epv_ctx *c, uint32_t *mode) {
  if (!c || !mode || !c->have_paths) return EPV_ERR_ARG;
  const bool refq = c->S.flags & (EPV_FLAG_REFERENCE_PROPOSAL_RATIO | EPV_FLAG_SAMPLE_ROOT);
  static const bool p2_global = std::getenv("EPV_PROPOSE_V2_GLOBAL") != nullptr;
  const bool p2 = c->use_p2 && !refq && (!c->p2_gpool || p2_global);
  if (c->p3 && !refq) { *mode = EPV_PHASE_V3; return EPV_OK; }
  *mode = !p2 ? EPV_PHASE_V1 : c->fused ? EPV_PHASE_FUSED : seg_jumps_on(c) ? EPV_PHASE_V2_SEGMENTS : EPV_PHASE_V2;
  return EPV_OK;
}

EPV_API int epv_get_capacity(epv_ctx *c, uint32_t *capacity) {
  if (!c || !capacity || !c->have_paths) return EPV_ERR_ARG;
  *capacity = c->S.C;
  return EPV_OK;
}

EPV_API int epv_set_capacity(epv_ctx *c, uint32_t capacity) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (capacity < 1u) capacity = 1u;
  if (capacity > EPV_MAX_CAP) capacity = EPV_MAX_CAP;
  if (capacity == c->S.C) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t n = c->S.n, B = c->S.B, E = B * n;
  if (capacity < c->S.C) {
    // shrinking: every resident path (either buffer: a stale proposal is overwritten before
    // it is read, but keep the test simple) must fit
    std::vector<epv_meta_t> meta(2u * E);
    HIP_TRY(c, hipMemcpyAsync(meta.data(), c->S.meta, 2u * E * sizeof(epv_meta_t), hipMemcpyDeviceToHost, c->stream));
    std::vector<uint8_t> sel(n);
    HIP_TRY(c, hipMemcpyAsync(sel.data(), c->S.sel, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint64_t s = 0; s < n; ++s)
      for (uint64_t b = 0; b < B; ++b)
        if ((meta[((uint64_t)sel[s] * B + b) * n + s] & EPV_NJ_MASK) > capacity)
          return fail(c, EPV_ERR_CAPACITY, "a resident path has more jumps than the requested capacity");
  }
  DevTmp<double> nj;
  HIP_TRY(c, nj.alloc(2u * E * capacity));
  const uint32_t keep = std::min(capacity, c->S.C);
  // plane (buf, b) holds C rows of n doubles: rows 0..keep-1 move to the new stride
  for (uint64_t plane = 0; plane < 2u * B; ++plane)
    HIP_TRY(c, hipMemcpyAsync(nj.p + plane * capacity * n, c->S.jumps + plane * c->S.C * n,
                              (size_t)keep * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  const uint32_t W = (2u * capacity + 1u + 63u) / 64u;
  DevTmp<uint64_t> ns;
  HIP_TRY(c, ns.alloc(B * c->S.phase_cap * W));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(c->S.jumps);
  (void)hipFree(c->S.prop_states);
  c->S.jumps = nj.release();
  c->S.prop_states = ns.release();
  c->S.C = capacity;
  c->S.W = W;
  { int rc2 = plan_mh(c); if (!rc2) rc2 = plan_p2(c); return rc2 ? rc2 : plan_p3(c); }
}

EPV_API int epv_init_paths_indep(epv_ctx *c, uint64_t n_sites, const uint8_t *root_states,
                                 const uint8_t *leaf_states, uint64_t seed, uint32_t capacity) {
  if (!c) return EPV_ERR_ARG;
  if (!c->have_tree || !c->have_model)
    return fail(c, EPV_ERR_STATE, "epv_set_tree and epv_set_model must come before epv_init_paths_indep");
  if (c->S.N != 2) return fail(c, EPV_ERR_ARG, "epv_init_paths_indep needs the two-node (single branch) tree");
  if (n_sites < 3 || !root_states || !leaf_states) return fail(c, EPV_ERR_ARG, "bad states");
  // device paths start as (init = root, no jumps)
  std::vector<uint64_t> off(n_sites + 1, 0);
  double dummy = 0.0;
  int rc = epv_upload_paths(c, n_sites, root_states, off.data(), &dummy, capacity ? capacity : 32u, 0);
  if (rc) return rc;
  DevTmp<uint8_t> leaf_tmp;
  HIP_TRY(c, leaf_tmp.alloc(n_sites));
  uint8_t *d_leaf = leaf_tmp.p;
  HIP_TRY(c, hipMemcpyAsync(d_leaf, leaf_states, n_sites, hipMemcpyHostToDevice, c->stream));
  const uint64_t first = 1, last = n_sites - 2;
  const uint64_t threads = (last - first + 1u + 2u) / 3u;
  const unsigned blocks = (unsigned)((threads + 255u) / 256u);
  for (uint32_t colour = 0; colour < 3; ++colour) {
    const uint64_t s0 = first + ((colour + 3u - (uint32_t)((c->S.g0 + first) % 3u)) % 3u);
    // 64-lane blocks like epv_mh_propose_kernel: the per-shard task regions are sized for them
    hipLaunchKernelGGL(epv_init_tasks_kernel, dim3((unsigned)((threads + 63u) / 64u)), dim3(64), 0, c->stream,
                       c->S, colour, first, last, d_leaf, c->d_counters);
    const uint64_t per_shard = threads / EPV_SHARDS + 256u;
    const uint32_t tpw = c->tasks_per_wave ? c->tasks_per_wave : 32u;
    const uint64_t jb = std::min<uint64_t>((per_shard + 4u * tpw - 1u) / (4u * tpw), 256u);
    hipLaunchKernelGGL(epv_mh_jumps_kernel, dim3((unsigned)jb, EPV_SHARDS), dim3(256), const_lds_bytes(c->S.N),
                       c->stream, c->S, (uint32_t)seed, (uint32_t)(seed >> 32), EPV_INIT_SWEEP,
                       tpw, s0, 0.0, 0.0, c->d_counters);
    hipLaunchKernelGGL(epv_init_commit_kernel, dim3(blocks), dim3(256), 0, c->stream, c->S, colour, first,
                       last, c->d_counters);
  }
  hipLaunchKernelGGL(epv_init_flip_kernel, dim3((unsigned)((last - first + 256u) / 256u)), dim3(256), 0,
                     c->stream, c->S, first, last);
  hipLaunchKernelGGL(epv_init_ends_kernel, dim3(1), dim3(64), 0, c->stream, c->S, d_leaf, (uint32_t)seed,
                     (uint32_t)(seed >> 32), c->blen[1]);
  HIP_TRY(c, hipGetLastError());
  return finish_mcmc(c, nullptr, 0);  // synchronises (d_leaf is released afterwards); reports overflow
}

// ---------------------------------------------------------------------------------------
//  site-independent model (IndepSite.cpp), used by epievo_initialization
// ---------------------------------------------------------------------------------------
namespace {
// continuous_time_trans_prob_mat + expectation_J/D (ContinuousTimeMarkovModel.cpp:143-226)
// for every branch, with epv_exp so that the values equal the oracle's parallel rung
int upload_indep_consts(epv_ctx *c, const double *rates) {
  if (!(rates[0] > 0.0) || !(rates[1] > 0.0)) return fail(c, EPV_ERR_ARG, "indep rates must be positive");
  std::vector<EpvIndepConst> k(c->S.N);
  const double r0 = rates[0], r1 = rates[1];
  for (uint32_t node = 1; node < c->S.N; ++node) {
    const double T = c->blen[node];
    EpvIndepConst &q = k[node];
    {
      const double h = 1.0 / epv_exp(T * (r0 + r1));
      const double denom = r0 + r1;
      q.P[0] = (r0 * h + r1) / denom;
      q.P[1] = 1.0 - q.P[0];
      q.P[3] = (r0 + r1 * h) / denom;
      q.P[2] = 1.0 - q.P[3];
    }
    const double s = r0 + r1, p = r0 * r1, d = r1 - r0;
    const double e = epv_exp(-s * T);
    const double C1 = d * (1 - e) / s;
    q.J0[0] = p * (T * (r1 - r0 * e) - C1) / (s * (r1 + r0 * e));
    q.J1[0] = q.J0[0];
    q.J0[3] = p * (T * (r0 - r1 * e) + C1) / (s * (r0 + r1 * e));
    q.J1[3] = q.J0[3];
    const double C2 = p * T * (1 + e) / (s * (1 - e));
    const double C3 = (r0 * r0 + r1 * r1) / (s * s);
    const double C4 = (2 * p) / (s * s);
    q.J0[1] = C2 + C3; q.J1[1] = C2 - C4; q.J0[2] = q.J1[1]; q.J1[2] = q.J0[1];
    const double r00 = r0 * r0, r11 = r1 * r1;
    const double E1 = 2 * p * (1 - e) / s;
    q.D0[0] = ((r11 + r00 * e) * T + E1) / (s * (r1 + r0 * e));
    q.D1[0] = T - q.D0[0];
    q.D1[3] = ((r00 + r11 * e) * T + E1) / (s * (r0 + r1 * e));
    q.D0[3] = T - q.D1[3];
    const double E2 = (p - r00) * (1 - e) / s;
    q.D1[1] = ((r00 - p * e) * T + E2) / (s * (r0 - r0 * e));
    q.D0[1] = T - q.D1[1];
    const double E3 = (p - r11) * (1 - e) / s;
    q.D0[2] = ((r11 - p * e) * T + E3) / (s * (r1 - r1 * e));
    q.D1[2] = T - q.D0[2];
  }
  if (!c->d_indep) HIP_TRY(c, hipMalloc(&c->d_indep, sizeof(EpvIndepConst) * c->S.N));
  HIP_TRY(c, hipMemcpy(c->d_indep, k.data(), sizeof(EpvIndepConst) * c->S.N, hipMemcpyHostToDevice));
  return EPV_OK;
}

int indep_stats(epv_ctx *c, const double *rates, uint32_t what, double *J, double *D) {
  const uint32_t B = c->S.B, V16 = ((B * 4u + 15u) / 16u) * 16u;
  // 256-lane blocks when the per-node LDS table fits, 64-lane blocks for large trees
  uint32_t threads = 256u;
  size_t lds = (size_t)c->S.N * threads * 5u * sizeof(double);
  if (lds > 60u * 1024u) { threads = 64u; lds = (size_t)c->S.N * threads * 5u * sizeof(double); }
  const uint64_t nb = (c->S.n + threads - 1u) / threads;
  // the tree-reduction buffers are shared with the 8-context statistics but sized for THIS
  // launch shape: nb rows of V16 doubles at level 0, ceil(nb/256) rows at level 1
  int rc = ensure_partial_doubles(c, nb * V16, ((nb + 255u) / 256u) * V16);
  if (rc) return rc;
  if (lds > 150u * 1024u) return fail(c, EPV_ERR_ARG, "tree too large for the site-independent kernels");
  if (lds > 60u * 1024u &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(epv_indep_stats_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(c, EPV_ERR_HIP, "cannot raise the dynamic LDS limit");
  double pi_0 = 0.0;
  if (what == 0u) pi_0 = rates[1] / (rates[0] + rates[1]);
  hipLaunchKernelGGL(epv_indep_stats_kernel, dim3((unsigned)nb), dim3(threads), lds, c->stream, c->S, c->d_indep,
                     pi_0, what, V16, c->d_partial[0]);
  uint64_t m = nb;
  int cur = 0;
  while (m > 1) {
    const uint64_t mb = (m + 255u) / 256u;
    hipLaunchKernelGGL(epv_tree_reduce_kernel, dim3((unsigned)mb, V16 / 16u), dim3(256), 0, c->stream,
                       c->d_partial[cur], m, V16, c->d_partial[cur ^ 1]);
    m = mb;
    cur ^= 1;
  }
  HIP_TRY(c, hipGetLastError());
  std::vector<double> v(V16);
  HIP_TRY(c, hipMemcpyAsync(v.data(), c->d_partial[cur], V16 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (uint32_t b = 0; b < B; ++b) {
    J[b * 2 + 0] = v[b * 4 + 0]; J[b * 2 + 1] = v[b * 4 + 1];
    D[b * 2 + 0] = v[b * 4 + 2]; D[b * 2 + 1] = v[b * 4 + 3];
  }
  return EPV_OK;
}
}  // namespace

EPV_API int epv_indep_expectation(epv_ctx *c, const double *rates, double *J, double *D) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (!rates || !J || !D) return fail(c, EPV_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->device));
  if ((rc = upload_indep_consts(c, rates))) return rc;
  return indep_stats(c, rates, 0u, J, D);
}

EPV_API int epv_indep_sufficient_statistics(epv_ctx *c, double *J, double *D) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (!J || !D) return fail(c, EPV_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->device));
  const double one[2] = {1.0, 1.0};
  if ((rc = upload_indep_consts(c, one))) return rc;   // constants unused by the counting mode
  if ((rc = indep_stats(c, one, 1u, J, D))) return rc;
  for (uint32_t i = 0; i < 2u * c->S.B; ++i) {   // averages over the sites (IndepSite.cpp:291-295)
    J[i] /= (double)c->S.n;
    D[i] /= (double)c->S.n;
  }
  return EPV_OK;
}

EPV_API int epv_indep_update_paths(epv_ctx *c, const double *rates, uint64_t seed, uint32_t sweep) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (!rates) return fail(c, EPV_ERR_ARG, "null argument");
  if (c->S.g0 != 0) return fail(c, EPV_ERR_ARG, "epv_indep_update_paths works on an unsharded genome");
  HIP_TRY(c, hipSetDevice(c->device));
  if ((rc = upload_indep_consts(c, rates))) return rc;
  const uint64_t first = 0, last = c->S.n - 1;
  const uint64_t threads = (last - first + 1u + 2u) / 3u;
  const size_t lds = (size_t)c->S.N * 64u * 4u * sizeof(double);
  if (lds > 150u * 1024u) return fail(c, EPV_ERR_ARG, "tree too large for the site-independent kernels");
  if (lds > 60u * 1024u &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(epv_indep_propose_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(c, EPV_ERR_HIP, "cannot raise the dynamic LDS limit");
  for (uint32_t colour = 0; colour < 3; ++colour) {
    const uint64_t s0 = first + ((colour + 3u - (uint32_t)((c->S.g0 + first) % 3u)) % 3u);
    const uint64_t blocks = (threads + 63u) / 64u;
    hipLaunchKernelGGL(epv_indep_propose_kernel, dim3((unsigned)blocks), dim3(64), lds, c->stream, c->S,
                       c->d_indep, rates[0], rates[1], colour, first, last, (uint32_t)seed,
                       (uint32_t)(seed >> 32), sweep, c->d_counters);
    const uint64_t max_tasks = blocks / EPV_SHARDS * 64u * c->S.B + 64u * c->S.B;
    const uint32_t tpw = c->tasks_per_wave ? c->tasks_per_wave : 32u;
    const uint64_t jb = std::min<uint64_t>((max_tasks / 4u + 4u * tpw - 1u) / (4u * tpw) + 1u, 256u);
    hipLaunchKernelGGL(epv_mh_jumps_kernel, dim3((unsigned)jb, EPV_SHARDS), dim3(256), const_lds_bytes(c->S.N),
                       c->stream, c->S, (uint32_t)seed, (uint32_t)(seed >> 32), sweep, tpw, s0, rates[0],
                       rates[1], c->d_counters);
    hipLaunchKernelGGL(epv_indep_commit_kernel, dim3((unsigned)((threads + 255u) / 256u)), dim3(256), 0,
                       c->stream, c->S, colour, first, last, c->d_counters);
  }
  HIP_TRY(c, hipGetLastError());
  c->have_reset = false;
  return finish_mcmc(c, nullptr, 0);
}

EPV_API int epv_set_global_length(epv_ctx *c, uint64_t n_global) {
  if (!c || !c->have_paths) return EPV_ERR_ARG;
  if (n_global < c->S.g0 + c->S.n) return fail(c, EPV_ERR_ARG, "n_global smaller than the shard");
  c->S.n_global = n_global;
  return EPV_OK;
}

EPV_API int epv_set_update_range(epv_ctx *c, uint64_t first, uint64_t last) {
  if (!c || !c->have_paths) return EPV_ERR_ARG;
  if (first < 1 || last > c->S.n - 2 || first > last) return fail(c, EPV_ERR_ARG, "bad update range");
  if (c->S.g0 + first > 1 && first < 2) return fail(c, EPV_ERR_ARG, "shard needs a 2-site left halo");
  if (c->S.g0 + last < c->S.n_global - 2 && last > c->S.n - 3)
    return fail(c, EPV_ERR_ARG, "shard needs a 2-site right halo");
  c->first = first;
  c->last = last;
  c->halo_mode = false;
  return EPV_OK;
}

EPV_API int epv_set_halo(epv_ctx *c, uint64_t left, uint64_t right) {
  if (!c || !c->have_paths) return EPV_ERR_ARG;
  if ((left && left < 2) || (right && right < 2) || left + right + 1 > c->S.n)
    return fail(c, EPV_ERR_ARG, "halo blocks must be 0 or >= 2 columns and leave owned columns");
  if (!left && c->S.g0 != 0) return fail(c, EPV_ERR_ARG, "a shard that does not start the genome needs a left halo");
  if (!right && c->S.g0 + c->S.n != c->S.n_global)
    return fail(c, EPV_ERR_ARG, "a shard that does not end the genome needs a right halo");
  c->halo_left = left;
  c->halo_right = right;
  c->halo_mode = true;
  c->phases_used = 0;
  return EPV_OK;
}

EPV_API int epv_halo_phases_left(epv_ctx *c, uint64_t *phases) {
  if (!c || !phases || !c->have_paths) return EPV_ERR_ARG;
  if (!c->halo_mode || (!c->halo_left && !c->halo_right)) { *phases = ~0ull; return EPV_OK; }
  uint64_t h = ~0ull;
  if (c->halo_left) h = std::min(h, c->halo_left);
  if (c->halo_right) h = std::min(h, c->halo_right);
  const uint64_t total = h / 2u;  // phase p needs 2(p+1) <= halo
  *phases = total > c->phases_used ? total - c->phases_used : 0;
  return EPV_OK;
}

// the reset's launches without waiting for them: the MCMC calls that follow sit behind them on the
// context's stream (an EM driver goes from reset straight into run_mcmc: the device need not idle
// while the host finds that out)
EPV_API int epv_reset_async(epv_ctx *c) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->d_segtab) HIP_TRY(c, hipMalloc(&c->d_segtab, (size_t)4095u * 4u * 6u * sizeof(double)));
  hipLaunchKernelGGL(epv_reset_kernel, dim3((unsigned)((c->S.n + 255u) / 256u)), dim3(256),
                     const_lds_bytes(c->S.N), c->stream, c->S);
  // the single-segment matrices of every (branch, neighbour context): model and branch lengths
  // are fixed until the next reset
  hipLaunchKernelGGL(epv_segtab_kernel, dim3((c->S.B * 4u + 63u) / 64u), dim3(64), 0, c->stream, c->S, c->d_segtab);
  HIP_TRY(c, hipGetLastError());
  c->have_reset = true;
  return EPV_OK;
}

EPV_API int epv_reset(epv_ctx *c) {
  int rc = epv_reset_async(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return EPV_OK;
}

static int finish_mcmc(epv_ctx *c, uint64_t *n_accepted, uint64_t acc_base) {
  unsigned long long cnt[EPV_CNT_N];
  int rc = read_counters(c, cnt);
  if (rc) return rc;
  if (n_accepted) *n_accepted = cnt[EPV_CNT_ACCEPT] - acc_base;
  const uint64_t new_ovf = cnt[EPV_CNT_OVERFLOW] - c->tot_overflow;
  c->tot_overflow = cnt[EPV_CNT_OVERFLOW];
  c->tot_coop = cnt[EPV_CNT_COOP];
  if (new_ovf) {
    char buf[160];
    std::snprintf(buf, sizeof buf,
                  "%llu proposals needed more than %u jumps on a branch and were rejected; "
                  "re-upload with a larger capacity",
                  (unsigned long long)new_ovf, c->S.C);
    return fail(c, EPV_ERR_CAPACITY, buf);
  }
  return EPV_OK;
}

// the accept counters of this moment of the stream, kept on the device; finish_mcmc_snapshot reads them
// back together with the final ones -- no host synchronisation in the middle of a run
static int snapshot_counters(epv_ctx *c) {
  HIP_TRY(c, hipMemcpyAsync(c->d_cnt_snap, c->d_counters, sizeof(unsigned long long) * EPV_CNT_WORDS,
                            hipMemcpyDeviceToDevice, c->stream));
  return EPV_OK;
}
static int finish_mcmc_snapshot(epv_ctx *c, uint64_t *n_accepted) {
  HIP_TRY(c, hipMemcpyAsync(c->h_cnt_snap, c->d_cnt_snap, sizeof(unsigned long long) * EPV_CNT_WORDS,
                            hipMemcpyDeviceToHost, c->stream));
  uint64_t total = 0;
  int rc = finish_mcmc(c, &total, 0);      // synchronises the stream
  uint64_t base = 0;
  for (uint32_t sh = 0; sh < EPV_SHARDS; ++sh) base += c->h_cnt_snap[EPV_CNT_IDX(EPV_CNT_ACCEPT, sh)];
  if (n_accepted) *n_accepted = total - base;
  return rc;
}

static int current_accepts(epv_ctx *c, uint64_t *out) {
  unsigned long long cnt[EPV_CNT_N];
  int rc = read_counters(c, cnt);
  if (rc) return rc;
  *out = cnt[EPV_CNT_ACCEPT];
  return EPV_OK;
}

EPV_API int epv_sweep_phase(epv_ctx *c, int colour, uint64_t seed, uint32_t sweep,
                            uint64_t *n_accepted) {
  int rc = check_ready(c, true);
  if (rc) return rc;
  if (colour < 0 || colour > 2) return fail(c, EPV_ERR_ARG, "colour must be 0, 1 or 2");
  HIP_TRY(c, hipSetDevice(c->device));
  uint64_t base = 0;
  if ((rc = current_accepts(c, &base))) return rc;
  if ((rc = launch_phase(c, colour, seed, sweep))) return rc;
  return finish_mcmc(c, n_accepted, base);
}

EPV_API int epv_sweep(epv_ctx *c, uint64_t n_sweeps, uint64_t seed, uint32_t sweep_base,
                      uint64_t *n_accepted) {
  int rc = check_ready(c, true);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  uint64_t base = 0;
  if ((rc = current_accepts(c, &base))) return rc;
  for (uint64_t w = 0; w < n_sweeps; ++w) {
    for (int colour = 0; colour < 3; ++colour)
      if ((rc = launch_phase(c, colour, seed, sweep_base + (uint32_t)w))) return rc;
    ++c->n_sweeps;
  }
  return finish_mcmc(c, n_accepted, base);
}

EPV_API int epv_run_mcmc_sums(epv_ctx *c, uint64_t burn_in, uint64_t batch, uint64_t seed,
                              uint32_t sweep_base, int average, double *J, double *D,
                              uint64_t *n_accepted) {
  int rc = check_ready(c, true);
  if (rc) return rc;
  if (!J || !D || batch == 0) return fail(c, EPV_ERR_ARG, "bad run_mcmc arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t sweep = sweep_base;
  for (uint64_t w = 0; w < burn_in; ++w, ++sweep) {
    for (int colour = 0; colour < 3; ++colour)
      if ((rc = launch_phase(c, colour, seed, sweep))) return rc;
    ++c->n_sweeps;
  }
  if ((rc = ensure_sweep_tot(c, batch))) return rc;
  if ((rc = ensure_partials(c))) return rc;      // (allocations synchronise: before the snapshot)
  if ((rc = snapshot_counters(c))) return rc;
  for (uint64_t w = 0; w < batch; ++w, ++sweep) {
    for (int colour = 0; colour < 3; ++colour)
      if ((rc = launch_phase(c, colour, seed, sweep))) return rc;
    ++c->n_sweeps;
    if ((rc = launch_suffstats(c, w))) return rc;
  }
  HIP_TRY(c, hipGetLastError());
  rc = finish_mcmc_snapshot(c, n_accepted);  // synchronises the stream
  const int src = finish_stats(c, batch, average, J, D);
  return rc ? rc : src;
}

// ---- several shards on ONE GPU (epievo_amd.parallel.LocalGroup): each shard writes the
// level-0 block partials of its owned 256-site blocks, per batch sweep, into a buffer shared
// by the group; one reduction afterwards gives exactly the sums of the unsharded run
EPV_API int epv_dev_alloc(epv_ctx *c, uint64_t bytes, void **p) {
  if (!c || !p || !bytes) return EPV_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMalloc(p, bytes));
  HIP_TRY(c, hipMemset(*p, 0, bytes));
  HIP_TRY(c, hipDeviceSynchronize());   // the contexts' streams do not wait for the null stream
  return EPV_OK;
}
// small host <-> device copies into / out of such buffers (e.g. the accept count a shard appends
// to its statistic rows so that ONE all-gather carries everything)
EPV_API int epv_dev_write(epv_ctx *c, void *d_dst, const void *src, uint64_t bytes) {
  if (!c || !d_dst || !src) return EPV_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
  return EPV_OK;
}
EPV_API int epv_dev_read(epv_ctx *c, void *dst, const void *d_src, uint64_t bytes) {
  if (!c || !dst || !d_src) return EPV_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return EPV_OK;
}
EPV_API int epv_dev_free(epv_ctx *c, void *p) {
  if (!c) return EPV_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  if (p) HIP_TRY(c, hipFree(p));
  return EPV_OK;
}

EPV_API int epv_run_mcmc_blocks(epv_ctx *c, uint64_t burn_in, uint64_t batch, uint64_t seed,
                                uint32_t sweep_base, double *d_blocks, uint64_t nb_total,
                                int64_t block_offset, uint64_t *n_accepted) {
  int rc = check_ready(c, true);
  if (rc) return rc;
  if (!d_blocks || batch == 0) return fail(c, EPV_ERR_ARG, "bad run_mcmc_blocks arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t V = c->S.B * 16u;
  uint64_t own_lo = 0, own_hi = 0;
  owned_range(c, &own_lo, &own_hi);
  const uint64_t blk_lo = own_lo / 256u, blk_hi = own_hi / 256u;
  // local block 0 may lie left of the buffer (a halo in front of the owned columns); the OWNED blocks must be inside
  if (block_offset + (int64_t)blk_lo < 0 || block_offset + (int64_t)blk_hi >= (int64_t)nb_total)
    return fail(c, EPV_ERR_ARG, "owned blocks exceed the group's block buffer");
  if ((c->S.g0 & 255u) != 0u) return fail(c, EPV_ERR_ARG, "the shard must start on a 256-site block of the genome");
  if ((rc = ensure_stat_scale(c))) return rc;
  uint32_t sweep = sweep_base;
  for (uint64_t w = 0; w < burn_in; ++w, ++sweep) {
    for (int colour = 0; colour < 3; ++colour)
      if ((rc = launch_phase(c, colour, seed, sweep))) return rc;
    ++c->n_sweeps;
  }
  if ((rc = snapshot_counters(c))) return rc;
  // the statistics of a context that shares its GPU with others run as one-wave blocks (they fit
  // into the LDS the colour phases leave free); the waves of a 256-site block add into its row, so
  // this context's rows start from zero.  EPV_STAT_BLOCKS=1: the 256-lane kernel (A/B runs)
  static const bool stat_blocks = std::getenv("EPV_STAT_BLOCKS") != nullptr;
  unsigned long long *rows0 = (unsigned long long *)d_blocks + (uint64_t)(block_offset + (int64_t)blk_lo) * V;
  const uint64_t n_own = blk_hi - blk_lo + 1u;
  if (!stat_blocks)
    HIP_TRY(c, hipMemset2DAsync(rows0, nb_total * V * sizeof(unsigned long long), 0, n_own * V * sizeof(unsigned long long),
                                batch, c->stream));
  for (uint64_t w = 0; w < batch; ++w, ++sweep) {
    for (int colour = 0; colour < 3; ++colour)
      if ((rc = launch_phase(c, colour, seed, sweep))) return rc;
    ++c->n_sweeps;
    if (stat_blocks)
      hipLaunchKernelGGL(epv_suffstat_kernel, dim3((unsigned)n_own, (c->S.B + EPV_STAT_BCH - 1u) / EPV_STAT_BCH),
                         dim3(256), 0, c->stream, c->S, own_lo, own_hi, blk_lo, c->d_statscale, rows0 + w * nb_total * V);
    else
      hipLaunchKernelGGL(epv_suffstat_wave_kernel, dim3((unsigned)(n_own * 4u), (c->S.B + EPV_STATW_BCH - 1u) / EPV_STATW_BCH),
                         dim3(64), 0, c->stream, c->S, own_lo, own_hi, blk_lo, c->d_statscale, rows0 + w * nb_total * V);
  }
  HIP_TRY(c, hipGetLastError());
  return finish_mcmc_snapshot(c, n_accepted);  // synchronises the stream
}

EPV_API int epv_reduce_blocks(epv_ctx *c, const double *d_blocks, uint64_t nb_total, uint64_t batch,
                              int average, double *J, double *D) {
  if (!c || !d_blocks || !J || !D || !nb_total || !batch) return EPV_ERR_ARG;
  if (!c->have_tree) return fail(c, EPV_ERR_STATE, "epv_set_tree must come first");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_stat_scale(c);
  if (rc) return rc;
  if ((rc = reduce_blocks_to_tot(c, d_blocks, nb_total, batch, 0u))) return rc;
  return finish_stats(c, batch, average, J, D);
}

// ---- statistics of a genome sharded over several GPUs: every GPU turns the level-0 partials
// of its blocks into ROWS of row_blocks (a power of two) blocks, the rows of all GPUs are
// all-gathered (RCCL; the caller's business), and the last stage sums the rows of the whole
// genome -- the same balanced tree as the one-context reduction, so J and D keep their bits
EPV_API int epv_blocks_to_rows(epv_ctx *c, const double *d_blocks, uint64_t nb_total, uint64_t batch,
                               uint32_t row_blocks, double *d_rows) {
  if (!c || !d_blocks || !d_rows || !nb_total || !batch) return EPV_ERR_ARG;
  if (!c->have_tree) return fail(c, EPV_ERR_STATE, "epv_set_tree must come first");
  if (row_blocks == 0 || (row_blocks & (row_blocks - 1u))) return fail(c, EPV_ERR_ARG, "row_blocks must be a power of two");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t V = c->S.B * 16u;
  const uint64_t n_rows = (nb_total + row_blocks - 1u) / row_blocks;
  if (n_rows > 65535u || batch > 65535u) return fail(c, EPV_ERR_ARG, "too many rows or batch sweeps for one launch");
  // in: [w][block][V]   out: [row][w][V] (a GPU's rows are one contiguous piece of the gathered buffer)
  launch_isum(c, d_blocks, nb_total, (uint64_t)row_blocks, (uint64_t)V, nb_total * V, d_rows, batch * V, (uint64_t)V,
              n_rows, batch);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return EPV_OK;
}

EPV_API int epv_reduce_rows(epv_ctx *c, const double *d_rows, uint64_t n_rows, uint64_t batch, int average,
                            double *J, double *D) {
  if (!c || !d_rows || !J || !D || !n_rows || !batch) return EPV_ERR_ARG;
  if (!c->have_tree) return fail(c, EPV_ERR_STATE, "epv_set_tree must come first");
  if (batch > 65535u) return fail(c, EPV_ERR_ARG, "too many batch sweeps for one launch");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t V = c->S.B * 16u;
  int rc = ensure_stat_scale(c);
  if (rc) return rc;
  if ((rc = ensure_sweep_tot(c, batch))) return rc;
  // per batch sweep the integer total over all rows ([row][sweep][V]), then the host's accumulation
  launch_isum(c, d_rows, n_rows, 0u, batch * V, (uint64_t)V, c->d_sweep_tot, 0u, (uint64_t)V, 1u, batch);
  HIP_TRY(c, hipGetLastError());
  return finish_stats(c, batch, average, J, D);
}

// the same on the buffer an all-gather of equally sized pieces leaves behind:
// d_gathered[rank][max_rows][batch][V], of which the first rows_per_rank[rank] rows count
EPV_API int epv_reduce_gathered_rows(epv_ctx *c, const double *d_gathered, uint32_t world, uint64_t max_rows,
                                     uint64_t piece_doubles, const uint64_t *rows_per_rank, uint64_t batch,
                                     int average, double *J, double *D) {
  if (!c || !d_gathered || !world || !max_rows || !rows_per_rank || !batch) return EPV_ERR_ARG;
  if (!c->have_tree) return fail(c, EPV_ERR_STATE, "epv_set_tree must come first");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t V = (uint64_t)c->S.B * 16u, row = batch * V;
  if (piece_doubles == 0) piece_doubles = max_rows * row;
  if (piece_doubles < max_rows * row) return fail(c, EPV_ERR_ARG, "piece_doubles smaller than max_rows rows");
  uint64_t total = 0;
  for (uint32_t r = 0; r < world; ++r) {
    if (rows_per_rank[r] > max_rows) return fail(c, EPV_ERR_ARG, "rows_per_rank exceeds max_rows");
    total += rows_per_rank[r];
  }
  if (!total) return fail(c, EPV_ERR_ARG, "no rows");
  if (total * row > c->rows_cap) {
    dfree(c->d_rows);
    c->rows_cap = 0;
    HIP_TRY(c, hipMalloc(&c->d_rows, total * row * sizeof(double)));
    c->rows_cap = total * row;
  }
  uint64_t at = 0;
  for (uint32_t r = 0; r < world; ++r) {
    if (rows_per_rank[r])
      HIP_TRY(c, hipMemcpyAsync(c->d_rows + at * row, d_gathered + (uint64_t)r * piece_doubles,
                                rows_per_rank[r] * row * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    at += rows_per_rank[r];
  }
  return epv_reduce_rows(c, c->d_rows, total, batch, average, J, D);
}

EPV_API int epv_run_mcmc(epv_ctx *c, uint64_t burn_in, uint64_t batch, uint64_t seed,
                         uint32_t sweep_base, double *J, double *D, uint64_t *n_accepted) {
  return epv_run_mcmc_sums(c, burn_in, batch, seed, sweep_base, 1, J, D, n_accepted);
}

EPV_API int epv_get_sufficient_statistics(epv_ctx *c, double *J, double *D) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  if ((rc = launch_suffstats(c, 0u))) return rc;
  return finish_stats(c, 1u, 0, J, D);
}

EPV_API int epv_scale_jump_times(epv_ctx *c, const double *new_branches) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<double> scale(c->S.N, 1.0);
  for (uint32_t b = 1; b < c->S.N; ++b) {
    if (!(new_branches[b] > 0.0)) return fail(c, EPV_ERR_ARG, "branch lengths must be positive");
    scale[b] = new_branches[b] / c->blen[b];  // ParamEstimation.cpp:372
  }
  HIP_TRY(c, hipMemcpyAsync(c->d_scale, scale.data(), sizeof(double) * c->S.N, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(epv_scale_kernel, dim3((unsigned)((c->S.n + 255u) / 256u)), dim3(256), 0,
                     c->stream, c->S, c->d_scale);
  HIP_TRY(c, hipGetLastError());
  for (uint32_t b = 1; b < c->S.N; ++b) c->blen[b] = new_branches[b];
  HIP_TRY(c, hipMemcpyAsync(c->d_blen, c->blen.data(), sizeof(double) * c->S.N, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_reset = false;  // cached log-likelihoods are stale, as in the reference
  return EPV_OK;
}

static int export_paths(epv_ctx *c, uint8_t *init_state, uint64_t *offsets, double *jumps,
                        uint64_t *total) {
  const uint64_t E = (uint64_t)c->S.B * c->S.n;
  DevTmp<uint8_t> init_tmp;
  DevTmp<uint64_t> cnt_tmp;
  HIP_TRY(c, init_tmp.alloc(E));
  HIP_TRY(c, cnt_tmp.alloc(E + 1));
  uint8_t *d_init = init_tmp.p;
  uint64_t *d_cnt = cnt_tmp.p;
  hipLaunchKernelGGL(epv_count_kernel, dim3((unsigned)((E + 255u) / 256u)), dim3(256), 0, c->stream,
                     c->S, d_init, d_cnt);
  std::vector<uint64_t> off(E + 1);
  HIP_TRY(c, hipMemcpyAsync(off.data(), d_cnt, E * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint64_t run = 0;
  for (uint64_t e = 0; e < E; ++e) { const uint64_t k = off[e]; off[e] = run; run += k; }
  off[E] = run;
  if (total) *total = run;
  if (offsets) {
    std::memcpy(offsets, off.data(), (E + 1) * sizeof(uint64_t));
    HIP_TRY(c, hipMemcpy(init_state, d_init, E, hipMemcpyDeviceToHost));
    if (run) {
      DevTmp<double> j_tmp;
      HIP_TRY(c, j_tmp.alloc(run));
      double *d_j = j_tmp.p;
      HIP_TRY(c, hipMemcpyAsync(d_cnt, off.data(), (E + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(epv_gather_kernel, dim3((unsigned)((E + 255u) / 256u)), dim3(256), 0,
                         c->stream, c->S, d_cnt, d_j);
      HIP_TRY(c, hipMemcpyAsync(jumps, d_j, run * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
  }
  return EPV_OK;
}

EPV_API int epv_paths_total_jumps(epv_ctx *c, uint64_t *total) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  return export_paths(c, nullptr, nullptr, nullptr, total);
}

EPV_API int epv_download_paths(epv_ctx *c, uint8_t *init_state, uint64_t *offsets, double *jumps) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (!init_state || !offsets) return fail(c, EPV_ERR_ARG, "null output");
  HIP_TRY(c, hipSetDevice(c->device));
  return export_paths(c, init_state, offsets, jumps, nullptr);
}

EPV_API int epv_get_tri_llh(epv_ctx *c, double *out) {
  int rc = check_ready(c, true);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy(out, c->S.tri, c->S.n * sizeof(double), hipMemcpyDeviceToHost));
  return EPV_OK;
}

EPV_API uint64_t epv_column_bytes(const epv_ctx *c) {
  if (!c || !c->have_paths) return 0;
  return (((uint64_t)c->S.B * sizeof(epv_meta_t) + 7u) & ~7ull) + ((uint64_t)c->S.B * c->S.C + 3u) * 8u;
}

EPV_API int epv_get_columns(epv_ctx *c, uint64_t first, uint64_t count, void *packed) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (first + count > c->S.n || !packed) return fail(c, EPV_ERR_ARG, "bad column range");
  if (count == 0) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t bytes = count * epv_column_bytes(c);
  if ((rc = ensure_stage(c, bytes))) return rc;
  uint8_t *d = c->d_stage;
  hipLaunchKernelGGL(epv_pack_columns_kernel, dim3((unsigned)count), dim3(64), 0, c->stream, c->S,
                     first, count, d);
  HIP_TRY(c, hipMemcpyAsync(packed, d, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return EPV_OK;
}

// columns [src_first, src_first+count) of `src` -> columns [dst_first, ...) of `dst`, both on
// the same GPU, without leaving it (halo refresh between the shards of a LocalGroup)
EPV_API int epv_copy_columns(epv_ctx *src, uint64_t src_first, uint64_t count, epv_ctx *dst,
                             uint64_t dst_first) {
  int rc = check_ready(src, false);
  if (rc) return rc;
  if ((rc = check_ready(dst, false))) return rc;
  if (src->device != dst->device || src->S.B != dst->S.B || src->S.C != dst->S.C)
    return fail(dst, EPV_ERR_ARG, "epv_copy_columns needs two contexts of one GPU with equal tree and capacity");
  if (src_first + count > src->S.n || dst_first + count > dst->S.n) return fail(dst, EPV_ERR_ARG, "bad column range");
  if (count == 0) return EPV_OK;
  HIP_TRY(src, hipSetDevice(src->device));
  const uint64_t bytes = count * epv_column_bytes(src);
  if ((rc = ensure_stage(src, bytes))) return rc;
  hipLaunchKernelGGL(epv_pack_columns_kernel, dim3((unsigned)count), dim3(64), 0, src->stream, src->S,
                     src_first, count, src->d_stage);
  HIP_TRY(src, hipGetLastError());
  HIP_TRY(src, hipStreamSynchronize(src->stream));
  hipLaunchKernelGGL(epv_unpack_columns_kernel, dim3((unsigned)count), dim3(64), 0, dst->stream, dst->S,
                     dst_first, count, src->d_stage);
  HIP_TRY(dst, hipGetLastError());
  HIP_TRY(dst, hipStreamSynchronize(dst->stream));
  return EPV_OK;
}

// the same without a host synchronisation: the columns are packed on src's stream into half `slot`
// (0 or 1) of its staging buffer and unpacked on dst's stream behind an event -- whatever the caller
// launches on dst's stream afterwards (epv_reset) sees them; src must not be asked for the same slot
// again before dst's stream has passed the unpack
EPV_API int epv_copy_columns_async(epv_ctx *src, uint64_t src_first, uint64_t count, epv_ctx *dst,
                                   uint64_t dst_first, int slot) {
  int rc = check_ready(src, false);
  if (rc) return rc;
  if ((rc = check_ready(dst, false))) return rc;
  if (slot < 0 || slot > 1) return fail(dst, EPV_ERR_ARG, "slot must be 0 or 1");
  if (src->device != dst->device || src->S.B != dst->S.B || src->S.C != dst->S.C)
    return fail(dst, EPV_ERR_ARG, "epv_copy_columns_async needs two contexts of one GPU with equal tree and capacity");
  if (src_first + count > src->S.n || dst_first + count > dst->S.n) return fail(dst, EPV_ERR_ARG, "bad column range");
  if (count == 0) return EPV_OK;
  HIP_TRY(src, hipSetDevice(src->device));
  const uint64_t bytes = count * epv_column_bytes(src);
  if (2u * bytes > src->stage_cap) {
    // (growing the buffer waits for both streams: a half of it may still be read)
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    if ((rc = ensure_stage(src, 2u * bytes))) return rc;
  }
  if (!src->ev_copy[slot]) HIP_TRY(src, hipEventCreateWithFlags(&src->ev_copy[slot], hipEventDisableTiming));
  uint8_t *stage = src->d_stage + (slot ? src->stage_cap / 2u : 0u);
  hipLaunchKernelGGL(epv_pack_columns_kernel, dim3((unsigned)count), dim3(64), 0, src->stream, src->S,
                     src_first, count, stage);
  HIP_TRY(src, hipGetLastError());
  HIP_TRY(src, hipEventRecord(src->ev_copy[slot], src->stream));
  HIP_TRY(dst, hipStreamWaitEvent(dst->stream, src->ev_copy[slot], 0));
  hipLaunchKernelGGL(epv_unpack_columns_kernel, dim3((unsigned)count), dim3(64), 0, dst->stream, dst->S,
                     dst_first, count, stage);
  HIP_TRY(dst, hipGetLastError());
  return EPV_OK;
}

// the same with the packed columns staying in DEVICE memory of the context's GPU (a buffer the
// caller hands to RCCL): nothing passes through the host
EPV_API int epv_pack_columns_dev(epv_ctx *c, uint64_t first, uint64_t count, void *d_packed) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (first + count > c->S.n || !d_packed) return fail(c, EPV_ERR_ARG, "bad column range");
  if (count == 0) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(epv_pack_columns_kernel, dim3((unsigned)count), dim3(64), 0, c->stream, c->S,
                     first, count, static_cast<uint8_t *>(d_packed));
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return EPV_OK;
}
EPV_API int epv_unpack_columns_dev(epv_ctx *c, uint64_t first, uint64_t count, const void *d_packed) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (first + count > c->S.n || !d_packed) return fail(c, EPV_ERR_ARG, "bad column range");
  if (count == 0) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(epv_unpack_columns_kernel, dim3((unsigned)count), dim3(64), 0, c->stream, c->S,
                     first, count, static_cast<const uint8_t *>(d_packed));
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return EPV_OK;
}
EPV_API int epv_device_of(const epv_ctx *c) { return c ? c->device : -1; }

EPV_API int epv_put_columns(epv_ctx *c, uint64_t first, uint64_t count, const void *packed) {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (first + count > c->S.n || !packed) return fail(c, EPV_ERR_ARG, "bad column range");
  if (count == 0) return EPV_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t bytes = count * epv_column_bytes(c);
  if ((rc = ensure_stage(c, bytes))) return rc;
  uint8_t *d = c->d_stage;
  HIP_TRY(c, hipMemcpyAsync(d, packed, bytes, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(epv_unpack_columns_kernel, dim3((unsigned)count), dim3(64), 0, c->stream, c->S,
                     first, count, d);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return EPV_OK;
}

EPV_API int epv_get_counters(epv_ctx *c, epv_counters *out) {
  if (!c || !out) return EPV_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  unsigned long long cnt[EPV_CNT_N];
  int rc = read_counters(c, cnt);
  if (rc) return rc;
  out->n_overflow = cnt[EPV_CNT_OVERFLOW];
  out->n_coop_tasks = cnt[EPV_CNT_COOP];
  out->n_sweeps = c->n_sweeps;
  out->reserved = 0;
  return EPV_OK;
}

#ifdef EPV_P2_PROFILE
// sums over the per-wave rows: out[0..14] section cycles, out[15] waves
EPV_API int epv_debug_p2_profile(unsigned long long *out) {
  std::vector<unsigned long long> rows(16u * EPV_P2_PROF_ROWS);
  if (hipMemcpyFromSymbol(rows.data(), HIP_SYMBOL(epv_p2_prof), rows.size() * sizeof(unsigned long long)) != hipSuccess) return 1;
  for (int q = 0; q < 16; ++q) out[q] = 0;
  for (size_t r = 0; r < EPV_P2_PROF_ROWS; ++r)
    for (int q = 0; q < 16; ++q) out[q] += rows[16u * r + q];
  return 0;
}
#endif

EPV_API int epv_set_timing(epv_ctx *c, int enabled) {
  if (!c) return EPV_ERR_ARG;
  c->timing_every = enabled > 0 ? (uint32_t)enabled : 0u;
  c->timing_seen = 0;
  c->timing = false;
  return EPV_OK;
}

EPV_API int epv_kernel_time_ms(epv_ctx *c, double *avg_ms, uint64_t *n_launches) {
  if (!c || !avg_ms || !n_launches) return EPV_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int rc = drain_timing(c);
  if (rc) return rc;
  *n_launches = c->timed_launches;
  *avg_ms = c->timed_launches ? c->timed_ms / (double)c->timed_launches : 0.0;
  c->timed_ms = 0.0;
  c->timed_launches = 0;
  return EPV_OK;
}
