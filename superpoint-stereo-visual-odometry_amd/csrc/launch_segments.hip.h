// launch_segments.hip.h -- runs of kernel launches recorded and replayed from HIP graphs (included by spvo_internal.hip.h; see below).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include <type_traits>
// ---------------------------------------------------------------------------------------------------------------- launch segments
// The frame loop of the small engines (FP16 / INT8: BASELINE configs 3 and 5) is bound by the HOST: ~20 kernel launches per frame at
// 2.6-4 us each against ~0.1 ms of network (NOTES.md round 6; tools/graph_bench.hip: ONE hipGraphLaunch costs the host 4.7 us whatever the
// number of kernel nodes).  So a run of kernel launches on one stream between two event operations -- a SEGMENT: a group's trunk, its heads,
// a pair's heat map + NMS + sampling, its two matches -- is recorded instead of launched (the launch macro below lands here), and when the
// segment closes it is either replayed from the HIP graph built the last time the same segment (same key: buffer set, slots, batch, plan
// and tuning generation ...) came by, or launched kernel by kernel and turned into a graph for the next time.  The enqueueing code is
// unchanged: every hipLaunchKernelGGL of the library goes through launch_kernel(), which launches at once unless a segment is open on
// that stream.  Anything that is not a kernel launch (event record / wait, copies, memsets -- also the profiler's events) inside an
// open segment flushes it and falls back to plain launches for the rest of that segment (rec_poison), so ordering is never at risk.
#include <initializer_list>
#include <tuple>
#include <utility>
namespace spvo_int {
struct LaunchNode { const void *func; dim3 grid, block; unsigned lds; int n_args; unsigned arg_off[32]; };
struct GraphEntry {
  bool valid = false, never = false;
  unsigned long long key = 0, seen_key = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<const void *> funcs;           // the replayed segment must be the same kernels with the same launch dimensions
  std::vector<unsigned long long> dims;
};
struct LaunchRecorder {
  bool active = false, poisoned = false;
  hipStream_t stream = nullptr;
  GraphEntry *entry = nullptr;
  unsigned long long key = 0;
  std::vector<LaunchNode> nodes;
  std::vector<char> arena;                   // argument copies of the open segment
  size_t used = 0;
  long graph_launches = 0, direct_segments = 0, poisoned_segments = 0;
};
extern __thread LaunchRecorder *t_rec;   // (__thread, not thread_local: no dynamic-initialisation wrapper between the translation units of the library)
void rec_flush_direct(LaunchRecorder *r);    // launches the recorded nodes one by one, in order, and forgets them
inline void rec_poison() {
  LaunchRecorder *r = t_rec;
  if (!r || !r->active) return;
  rec_flush_direct(r);
  r->poisoned = true;
  r->active = false;
  ++r->poisoned_segments;   // (not held against the entry: the profiler's events come and go)
}
template <typename T> inline void rec_put_arg(LaunchRecorder *r, LaunchNode &n, const T &v, bool &ok) {
  static_assert(std::is_trivially_copyable<T>::value, "kernel arguments are plain data");
  const size_t at = (r->used + 15) & ~(size_t)15;
  if (n.n_args >= 32 || at + sizeof(T) > r->arena.size()) { ok = false; return; }
  std::memcpy(r->arena.data() + at, &v, sizeof(T));
  n.arg_off[n.n_args++] = (unsigned)at;
  r->used = at + sizeof(T);
}
template <typename... KArgs, typename... Args>
inline void launch_kernel(void (*k)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args &&...a) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  std::tuple<std::decay_t<KArgs>...> vals{static_cast<std::decay_t<KArgs>>(std::forward<Args>(a))...};
  LaunchRecorder *r = t_rec;
  if (r && r->active) {
    if (stream == r->stream) {
      LaunchNode n{(const void *)k, grid, block, (unsigned)lds, 0, {}};
      bool ok = true;
      std::apply([&](const auto &...v) { (rec_put_arg(r, n, v, ok), ...); }, vals);
      if (ok) { r->nodes.push_back(n); return; }
    }
    rec_poison();   // another stream inside the segment, or the arena is full: plain launches from here on
  }
  void *params[sizeof...(KArgs) + 1];
  int i = 0;
  std::apply([&](auto &...v) { ((params[i++] = (void *)&v), ...); }, vals);
  (void)hipLaunchKernel((const void *)k, grid, block, params, lds, stream);
}
// stream operations that are not kernel launches close an open segment first (defined before the macros below rename the calls)
template <typename... A> inline hipError_t guarded_memset_async(A... a) { rec_poison(); return hipMemsetAsync(a...); }
template <typename... A> inline hipError_t guarded_memcpy_async(A... a) { rec_poison(); return hipMemcpyAsync(a...); }
template <typename... A> inline hipError_t guarded_event_record(A... a) { rec_poison(); return hipEventRecord(a...); }
template <typename... A> inline hipError_t guarded_stream_wait_event(A... a) { rec_poison(); return hipStreamWaitEvent(a...); }
template <typename... A> inline hipError_t guarded_stream_synchronize(A... a) { rec_poison(); return hipStreamSynchronize(a...); }
}  // namespace spvo_int
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ::spvo_int::launch_kernel(kernel, grid, block, lds, stream, ##__VA_ARGS__)
#define hipMemsetAsync ::spvo_int::guarded_memset_async
#define hipMemcpyAsync ::spvo_int::guarded_memcpy_async
#define hipEventRecord ::spvo_int::guarded_event_record
#define hipStreamWaitEvent ::spvo_int::guarded_stream_wait_event
#define hipStreamSynchronize ::spvo_int::guarded_stream_synchronize

