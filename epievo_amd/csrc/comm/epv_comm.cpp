// epv_comm.cpp -- include/epievo_mi355x_comm.h: the halo exchange and the statistics
// all-gather of a site-sharded genome, on RCCL linked directly (xGMI inside a node), with a
// loopback transport for rehearsing several ranks on ONE physical GPU (RCCL refuses that).
// Host code only (HIP runtime + RCCL API); built into epievo_amd/libepv_rccl.so.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "epievo_mi355x.h"
#include "epievo_mi355x_comm.h"

#define EPV_API extern "C" __attribute__((visibility("default")))

namespace {

// loopback: the ranks of one process post their buffers; epv_comm_group_end moves the bytes
struct LoopOp {
  int kind = 0;   // 1 = exchange, 2 = all-gather
  const void *send_prev = nullptr, *send_next = nullptr, *send = nullptr;
  void *recv_prev = nullptr, *recv_next = nullptr, *recv = nullptr;
  uint64_t bytes_prev = 0, bytes_next = 0, bytes = 0;
};
struct LoopGroup {
  std::mutex mu;
  std::vector<int> devices;
  std::vector<std::vector<LoopOp>> posted;   // per rank, in call order
  int alive = 0;
};

thread_local int g_group_depth = 0;
thread_local std::vector<LoopGroup *> g_touched;   // loopback groups with posts in the open bracket

}  // namespace

struct epv_comm {
  int device = 0, rank = 0, world = 1;
  bool rccl = false;
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;
  std::shared_ptr<LoopGroup> loop;
  std::string err;
};

namespace {

int fail(epv_comm *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}
#define HIP_TRY(c, call)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail((c), EPV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
  } while (0)
#define NCCL_TRY(c, call)                                                                 \
  do {                                                                                    \
    ncclResult_t r_ = (call);                                                             \
    if (r_ != ncclSuccess && r_ != ncclInProgress)                                        \
      return fail((c), EPV_ERR_HIP, std::string(#call) + ": " + ncclGetErrorString(r_));    \
  } while (0)

int copy_between(int dev_src, const void *src, int dev_dst, void *dst, uint64_t bytes) {
  if (!bytes) return EPV_OK;
  hipError_t e = (dev_src == dev_dst) ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice)
                                      : hipMemcpyPeer(dst, dev_dst, src, dev_src, bytes);
  return e == hipSuccess ? EPV_OK : EPV_ERR_HIP;
}

// run the posted operations of a loopback group: call k of every rank belongs together
int run_loop_group(LoopGroup *g) {
  std::lock_guard<std::mutex> lk(g->mu);
  const size_t n = g->devices.size();
  size_t calls = g->posted[0].size();
  for (size_t r = 1; r < n; ++r)
    if (g->posted[r].size() != calls) { for (auto &p : g->posted) p.clear(); return EPV_ERR_ARG; }
  int rc = EPV_OK;
  for (int d : std::set<int>(g->devices.begin(), g->devices.end())) {   // producers are done
    if (hipSetDevice(d) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = EPV_ERR_HIP;
  }
  for (size_t k = 0; k < calls && rc == EPV_OK; ++k) {
    for (size_t r = 0; r < n && rc == EPV_OK; ++r) {
      const LoopOp &me = g->posted[r][k];
      if (me.kind == 1) {
        if (r + 1 < n) {
          const LoopOp &nx = g->posted[r + 1][k];
          if (nx.kind != 1 || nx.bytes_prev != me.bytes_next) { rc = EPV_ERR_ARG; break; }
          if ((rc = copy_between(g->devices[r], me.send_next, g->devices[r + 1], nx.recv_prev, me.bytes_next))) break;
          if ((rc = copy_between(g->devices[r + 1], nx.send_prev, g->devices[r], me.recv_next, me.bytes_next))) break;
        }
      } else if (me.kind == 2) {
        for (size_t q = 0; q < n; ++q) {
          const LoopOp &dst = g->posted[q][k];
          if (dst.kind != 2 || dst.bytes != me.bytes) { rc = EPV_ERR_ARG; break; }
          if ((rc = copy_between(g->devices[r], me.send, g->devices[q],
                                 static_cast<char *>(dst.recv) + r * me.bytes, me.bytes))) break;
        }
      }
    }
  }
  for (int d : std::set<int>(g->devices.begin(), g->devices.end()))
    if (hipSetDevice(d) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = rc ? rc : EPV_ERR_HIP;
  for (auto &p : g->posted) p.clear();
  return rc;
}

int post(epv_comm *c, const LoopOp &op) {
  LoopGroup *g = c->loop.get();
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->posted[c->rank].push_back(op);
  }
  if (g_group_depth == 0) {
    if (c->world != 1) return fail(c, EPV_ERR_ARG, "loopback ranks must be driven inside epv_comm_group_start/end");
    return run_loop_group(g) ? fail(c, EPV_ERR_HIP, "loopback copy failed") : EPV_OK;
  }
  bool seen = false;
  for (LoopGroup *t : g_touched) seen = seen || t == g;
  if (!seen) g_touched.push_back(g);
  return EPV_OK;
}

}  // namespace

EPV_API int epv_comm_init_all(int n_ranks, const int *devices, epv_comm **comms) {
  if (n_ranks < 1 || !devices || !comms) return EPV_ERR_ARG;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return EPV_ERR_HIP;
  std::set<int> distinct;
  for (int i = 0; i < n_ranks; ++i) {
    if (devices[i] < 0 || devices[i] >= count) return EPV_ERR_ARG;
    distinct.insert(devices[i]);
  }
  const bool rccl = (int)distinct.size() == n_ranks;
  std::vector<ncclComm_t> nc(n_ranks, nullptr);
  if (rccl && ncclCommInitAll(nc.data(), n_ranks, devices) != ncclSuccess) return EPV_ERR_HIP;
  std::shared_ptr<LoopGroup> loop;
  if (!rccl) {
    loop = std::make_shared<LoopGroup>();
    loop->devices.assign(devices, devices + n_ranks);
    loop->posted.resize(n_ranks);
    for (int a : distinct)           // copies between two rehearsal GPUs, if there are two
      for (int b : distinct)
        if (a != b) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can && hipSetDevice(a) == hipSuccess)
            (void)hipDeviceEnablePeerAccess(b, 0);
        }
  }
  for (int i = 0; i < n_ranks; ++i) {
    epv_comm *c = new epv_comm();
    c->device = devices[i];
    c->rank = i;
    c->world = n_ranks;
    c->rccl = rccl;
    c->nccl = nc[i];
    c->loop = loop;
    if (hipSetDevice(c->device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      for (int j = 0; j < i; ++j) epv_comm_destroy(comms[j]);
      return EPV_ERR_HIP;
    }
    comms[i] = c;
  }
  return EPV_OK;
}

EPV_API int epv_comm_get_unique_id(void *id) {
  if (!id) return EPV_ERR_ARG;
  ncclUniqueId u;
  static_assert(sizeof(u) == EPV_COMM_ID_BYTES, "RCCL unique id size");
  if (ncclGetUniqueId(&u) != ncclSuccess) return EPV_ERR_HIP;
  std::memcpy(id, &u, sizeof u);
  return EPV_OK;
}

EPV_API int epv_comm_init_rank(int device, int world, int rank, const void *id, epv_comm **comm) {
  if (!id || !comm || world < 1 || rank < 0 || rank >= world) return EPV_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return EPV_ERR_HIP;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  epv_comm *c = new epv_comm();
  c->device = device;
  c->rank = rank;
  c->world = world;
  c->rccl = true;
  if (ncclCommInitRank(&c->nccl, world, u, rank) != ncclSuccess ||
      hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return EPV_ERR_HIP;
  }
  *comm = c;
  return EPV_OK;
}

EPV_API void epv_comm_destroy(epv_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); }
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

EPV_API const char *epv_comm_last_error(const epv_comm *c) { return c ? c->err.c_str() : "null communicator"; }
EPV_API int epv_comm_is_rccl(const epv_comm *c) { return c && c->rccl ? 1 : 0; }
EPV_API int epv_comm_rank(const epv_comm *c) { return c ? c->rank : -1; }
EPV_API int epv_comm_world(const epv_comm *c) { return c ? c->world : 0; }

EPV_API int epv_comm_group_start(void) {
  if (g_group_depth++ == 0) g_touched.clear();
  return ncclGroupStart() == ncclSuccess ? EPV_OK : EPV_ERR_HIP;
}

EPV_API int epv_comm_group_end(void) {
  if (g_group_depth <= 0) return EPV_ERR_ARG;
  const ncclResult_t r = ncclGroupEnd();
  int rc = (r == ncclSuccess || r == ncclInProgress) ? EPV_OK : EPV_ERR_HIP;
  if (--g_group_depth == 0) {
    for (LoopGroup *g : g_touched) {
      const int lrc = run_loop_group(g);
      if (!rc) rc = lrc;
    }
    g_touched.clear();
  }
  return rc;
}

EPV_API int epv_comm_exchange(epv_comm *c, const void *d_send_prev, void *d_recv_prev, uint64_t bytes_prev,
                              const void *d_send_next, void *d_recv_next, uint64_t bytes_next) {
  if (!c) return EPV_ERR_ARG;
  if (c->rank == 0) bytes_prev = 0;
  if (c->rank == c->world - 1) bytes_next = 0;
  if ((bytes_prev && (!d_send_prev || !d_recv_prev)) || (bytes_next && (!d_send_next || !d_recv_next)))
    return fail(c, EPV_ERR_ARG, "null halo buffer");
  if (!c->rccl) {
    LoopOp op;
    op.kind = 1;
    op.send_prev = d_send_prev; op.recv_prev = d_recv_prev; op.bytes_prev = bytes_prev;
    op.send_next = d_send_next; op.recv_next = d_recv_next; op.bytes_next = bytes_next;
    return post(c, op);
  }
  HIP_TRY(c, hipSetDevice(c->device));
  // all four transfers of a rank in one RCCL group: no ordering between the neighbours' calls
  NCCL_TRY(c, ncclGroupStart());
  if (bytes_prev) {
    NCCL_TRY(c, ncclSend(d_send_prev, bytes_prev, ncclUint8, c->rank - 1, c->nccl, c->stream));
    NCCL_TRY(c, ncclRecv(d_recv_prev, bytes_prev, ncclUint8, c->rank - 1, c->nccl, c->stream));
  }
  if (bytes_next) {
    NCCL_TRY(c, ncclSend(d_send_next, bytes_next, ncclUint8, c->rank + 1, c->nccl, c->stream));
    NCCL_TRY(c, ncclRecv(d_recv_next, bytes_next, ncclUint8, c->rank + 1, c->nccl, c->stream));
  }
  NCCL_TRY(c, ncclGroupEnd());
  return EPV_OK;
}

EPV_API int epv_comm_all_gather(epv_comm *c, const void *d_send, void *d_recv, uint64_t bytes) {
  if (!c || !d_send || !d_recv || !bytes) return fail(c, EPV_ERR_ARG, "bad all-gather arguments");
  if (!c->rccl) {
    LoopOp op;
    op.kind = 2;
    op.send = d_send; op.recv = d_recv; op.bytes = bytes;
    return post(c, op);
  }
  HIP_TRY(c, hipSetDevice(c->device));
  NCCL_TRY(c, ncclAllGather(d_send, d_recv, bytes, ncclUint8, c->nccl, c->stream));
  return EPV_OK;
}

EPV_API int epv_comm_sync(epv_comm *c) {
  if (!c) return EPV_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->rccl) {
    ncclResult_t async = ncclSuccess;
    NCCL_TRY(c, ncclCommGetAsyncError(c->nccl, &async));
    if (async != ncclSuccess) return fail(c, EPV_ERR_HIP, std::string("RCCL: ") + ncclGetErrorString(async));
  }
  return EPV_OK;
}
