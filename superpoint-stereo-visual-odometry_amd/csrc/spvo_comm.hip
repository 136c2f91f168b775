// spvo_comm.hip -- the path's only collective: all-gather of per-frame relative poses.
//
// One process per GPU, one FeatureFrontEnd per process; stereo streams never exchange images or features, so the whole
// multi-GPU layer is this: every rank contributes n poses (7 doubles each: quaternion x, y, z, w + translation -- what
// solveStereoOdometry returns, feature_detection_base.cpp:377-385) and receives everybody's.  RCCL over xGMI
// (ncclAllGather on a stream of the communicator's own); 56 bytes per rank and frame is pure latency, so callers batch
// frames.  librccl is opened at run time (spvo_comm_create), never at load time: a single-GPU ROS node does not need it.
//
// spvo_comm_create_host is a host-memory transport (files in a directory) for CPU tests of the N > 1 code path only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spvo.h"

extern "C" void spvo_internal_set_error(const char *msg);   // spvo_core.hip: what spvo_last_error(NULL) returns

namespace {

int comm_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  spvo_internal_set_error(buf);
  return code;
}

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// The system RCCL, the one built against the HIP runtime this library links (a Python host may have loaded another copy
// bundled with its framework: an explicit path keeps the two apart).
Rccl *rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  std::vector<std::string> paths;
  if (const char *p = std::getenv("SPVO_RCCL_LIB")) paths.push_back(p);
  if (const char *p = std::getenv("ROCM_PATH")) paths.push_back(std::string(p) + "/lib/librccl.so.1");
  paths.push_back("/opt/rocm/lib/librccl.so.1");
  paths.push_back("librccl.so.1");
  for (const auto &p : paths)
    if ((r.handle = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
  if (!r.handle) return nullptr;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
  r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
    dlclose(r.handle);
    r.handle = nullptr;
    return nullptr;
  }
  return &r;
}

}  // namespace

struct spvo_comm {
  int rank = 0, world = 1, device = 0;
  // RCCL transport
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;
  double *d_send = nullptr, *d_recv = nullptr, *h_send = nullptr, *h_recv = nullptr;
  int cap = 0;   // poses per rank the buffers hold
  // host transport
  bool host = false;
  std::string dir;
  long seq = 0;
};

#define COMM_HIP(expr)                                                                                         \
  do {                                                                                                         \
    hipError_t _e = (expr);                                                                                    \
    if (_e != hipSuccess) return comm_fail(SPVO_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e));    \
  } while (0)
#define COMM_NCCL(expr)                                                                                        \
  do {                                                                                                         \
    ncclResult_t _e = (expr);                                                                                  \
    if (_e != ncclSuccess)                                                                                     \
      return comm_fail(SPVO_ERR_DEVICE, "%s failed: %s", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_e) : "rccl error"); \
  } while (0)

static int comm_reserve(spvo_comm *c, int n) {
  if (n <= c->cap) return SPVO_OK;
  const int cap = n < 64 ? 64 : n;
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->h_send) (void)hipHostFree(c->h_send);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  c->d_send = c->d_recv = c->h_send = c->h_recv = nullptr;
  c->cap = 0;
  COMM_HIP(hipMalloc((void **)&c->d_send, (size_t)cap * 7 * sizeof(double)));
  COMM_HIP(hipMalloc((void **)&c->d_recv, (size_t)c->world * cap * 7 * sizeof(double)));
  COMM_HIP(hipHostMalloc((void **)&c->h_send, (size_t)cap * 7 * sizeof(double)));
  COMM_HIP(hipHostMalloc((void **)&c->h_recv, (size_t)c->world * cap * 7 * sizeof(double)));
  c->cap = cap;
  return SPVO_OK;
}

extern "C" {

int spvo_comm_unique_id(unsigned char id[SPVO_COMM_ID_BYTES]) {
  static_assert(SPVO_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return comm_fail(SPVO_ERR_INVALID, "null id");
  Rccl *r = rccl();
  if (!r) {
    const char *why = dlerror();   // (a second call would return NULL: the message is cleared by the first)
    return comm_fail(SPVO_ERR_DEVICE, "librccl.so.1 not found (set SPVO_RCCL_LIB or ROCM_PATH): %s", why ? why : "");
  }
  ncclUniqueId u;
  COMM_NCCL(r->GetUniqueId(&u));
  std::memcpy(id, u.internal, SPVO_COMM_ID_BYTES);
  return SPVO_OK;
}

int spvo_comm_create(int device, int rank, int world, const unsigned char id[SPVO_COMM_ID_BYTES], spvo_comm **out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return comm_fail(SPVO_ERR_INVALID, "bad rank / world / id");
  *out = nullptr;
  Rccl *r = rccl();
  if (!r) return comm_fail(SPVO_ERR_DEVICE, "librccl.so.1 not found (set SPVO_RCCL_LIB or ROCM_PATH)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return comm_fail(SPVO_ERR_DEVICE, "no HIP device %d visible: the pose gather runs on RCCL, there is no CPU path (tests: spvo_comm_create_host)", device);
  COMM_HIP(hipSetDevice(device));
  spvo_comm *c = new spvo_comm;
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId u;
  std::memcpy(u.internal, id, SPVO_COMM_ID_BYTES);
  ncclResult_t e = r->CommInitRank(&c->nccl, world, u, rank);
  if (e != ncclSuccess) {
    delete c;
    return comm_fail(SPVO_ERR_DEVICE, "ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(e) : "rccl error");
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    r->CommDestroy(c->nccl);
    delete c;
    return comm_fail(SPVO_ERR_DEVICE, "hipStreamCreateWithFlags failed");
  }
  *out = c;
  return SPVO_OK;
}

static std::string host_file(const spvo_comm *c, long seq, int rank);

int spvo_comm_available(void) {
  return rccl() ? SPVO_OK : comm_fail(SPVO_ERR_DEVICE, "librccl.so.1 cannot be opened (SPVO_RCCL_LIB, $ROCM_PATH/lib, /opt/rocm/lib, the loader path)");
}

int spvo_comm_create_host(const char *dir, int rank, int world, spvo_comm **out) {
  if (!out || !dir || !*dir || world < 1 || rank < 0 || rank >= world) return comm_fail(SPVO_ERR_INVALID, "bad rank / world / directory");
  struct stat st;
  if (stat(dir, &st) != 0 || !S_ISDIR(st.st_mode)) return comm_fail(SPVO_ERR_IO, "no such directory: %s", dir);
  spvo_comm *c = new spvo_comm;
  c->rank = rank; c->world = world; c->host = true; c->dir = dir;
  // a directory that served an earlier communicator: this rank's done_ marker and pose files of that one must not be taken for this
  // one's (the destroy handshake would skip its wait and a peer could read a stale gather)
  std::remove((c->dir + "/done_" + std::to_string(rank)).c_str());
  for (long s = 0; s < 4; ++s) std::remove(host_file(c, s, rank).c_str());
  *out = c;
  return SPVO_OK;
}

int spvo_comm_rank(const spvo_comm *c) { return c ? c->rank : -1; }
int spvo_comm_world(const spvo_comm *c) { return c ? c->world : -1; }

static std::string host_file(const spvo_comm *c, long seq, int rank) {
  return c->dir + "/pose_" + std::to_string(seq) + "_" + std::to_string(rank) + ".bin";
}

int spvo_pose_allgather_n(spvo_comm *c, const double *poses, int n, double *all) {
  if (!c || !poses || !all || n < 1) return comm_fail(SPVO_ERR_INVALID, "spvo_pose_allgather: null argument or n < 1");
  const size_t mine = (size_t)n * 7 * sizeof(double);
  if (c->host) {
    // every rank publishes <dir>/pose_<seq>_<rank>.bin (write + rename: readers never see a partial file), then reads the
    // others'.  A rank that sees a peer's file for seq has proof that the peer finished reading seq - 1.
    const long seq = c->seq++;
    const std::string path = host_file(c, seq, c->rank), tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(poses, 1, mine, f) != mine) { if (f) std::fclose(f); return comm_fail(SPVO_ERR_IO, "cannot write %s", tmp.c_str()); }
    std::fclose(f);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) return comm_fail(SPVO_ERR_IO, "cannot publish %s", path.c_str());
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->world; ++r) {
      double *dst = all + (size_t)r * n * 7;
      if (r == c->rank) { std::memcpy(dst, poses, mine); continue; }
      const std::string peer = host_file(c, seq, r);
      for (;;) {
        struct stat st;
        if (stat(peer.c_str(), &st) == 0) {
          if ((size_t)st.st_size != mine) return comm_fail(SPVO_ERR_INVALID, "rank %d sent %ld bytes for gather %ld, expected %zu (all ranks must pass the same n)", r, (long)st.st_size, seq, mine);
          FILE *g = std::fopen(peer.c_str(), "rb");
          if (g && std::fread(dst, 1, mine, g) == mine) { std::fclose(g); break; }
          if (g) std::fclose(g);
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return comm_fail(SPVO_ERR_DEVICE, "timeout waiting for rank %d in gather %ld", r, seq);
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
    }
    if (seq >= 1) std::remove(host_file(c, seq - 1, c->rank).c_str());
    return SPVO_OK;
  }
  Rccl *r = rccl();
  COMM_HIP(hipSetDevice(c->device));
  int rc = comm_reserve(c, n);
  if (rc) return rc;
  std::memcpy(c->h_send, poses, mine);
  COMM_HIP(hipMemcpyAsync(c->d_send, c->h_send, mine, hipMemcpyHostToDevice, c->stream));
  COMM_NCCL(r->AllGather(c->d_send, c->d_recv, (size_t)n * 7, ncclDouble, c->nccl, c->stream));
  COMM_HIP(hipMemcpyAsync(c->h_recv, c->d_recv, mine * c->world, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(hipStreamSynchronize(c->stream));
  std::memcpy(all, c->h_recv, mine * c->world);
  return SPVO_OK;
}

int spvo_pose_allgather(spvo_comm *c, const double pose[7], double *all) { return spvo_pose_allgather_n(c, pose, 1, all); }

void spvo_comm_destroy(spvo_comm *c) {
  if (!c) return;
  if (c->host) {
    // A peer may still be reading this rank's last file (seeing a peer's file for gather s only proves that the peer has finished
    // s - 1).  So: publish done_<rank>, wait until every peer has published its own -- a rank does that after its last gather
    // returned, i.e. after its last read -- and only then remove this rank's pose files.  The markers stay for whoever removes
    // the directory; a directory that has vanished means the others are gone.  Bounded: a peer that died must not hang this one.
    const std::string mark = c->dir + "/done_" + std::to_string(c->rank);
    if (FILE *f = std::fopen(mark.c_str(), "wb")) std::fclose(f);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->world; ++r) {
      if (r == c->rank) continue;
      const std::string peer = c->dir + "/done_" + std::to_string(r);
      struct stat st;
      while (stat(peer.c_str(), &st) != 0 && stat(c->dir.c_str(), &st) == 0 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(30))
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    for (long s = c->seq > 2 ? c->seq - 2 : 0; s < c->seq; ++s) std::remove(host_file(c, s, c->rank).c_str());
  } else {
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->nccl && rccl()) rccl()->CommDestroy(c->nccl);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
  }
  delete c;
}

}  // extern "C"
