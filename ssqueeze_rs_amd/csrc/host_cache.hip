// host_cache.hip -- see host_cache.h.
#include "host_cache.h"

#include <algorithm>
#include <map>
#include <vector>

#include "../../include/ssq_hip.h"

namespace ssq {
namespace hostpath {

namespace {
std::mutex g_mu;
struct Scratch {
  void* p = nullptr;
  long long cap = 0;
};
struct DevState {
  Scratch slots[SLOT_COUNT];
  hipStream_t st[2] = {nullptr, nullptr};
};
std::map<int, DevState> g_dev;

// pinned pool
std::mutex g_pin_mu;
std::map<void*, long long> g_live;                    // handed out: ptr -> capacity
std::multimap<long long, void*> g_free;               // capacity -> ptr
long long g_free_bytes = 0;
long long g_free_limit = 8LL << 30;                   // keep at most this much idle pinned memory

long long round_cap(long long b) {
  const long long q = 1LL << 21;                      // 2 MiB granules: page-table friendly, few distinct sizes
  return ((b > 0 ? b : 1) + q - 1) / q * q;
}
}  // namespace

std::mutex& mutex() { return g_mu; }

int scratch(Slot s, long long bytes, void** p) {
  int dev = 0;
  SSQ_HIP(hipGetDevice(&dev));
  Scratch& sc = g_dev[dev].slots[s];
  if (bytes < 16) bytes = 16;
  if (sc.cap < bytes) {
    if (sc.p) SSQ_HIP(hipFree(sc.p));
    sc.p = nullptr;
    sc.cap = 0;
    const long long want = bytes + bytes / 8;          // a little headroom: batch sizes wobble
    if (hipMalloc(&sc.p, (size_t)want) != hipSuccess) {
      (void)hipGetLastError();
      SSQ_HIP(hipMalloc(&sc.p, (size_t)bytes));
      sc.cap = bytes;
    } else {
      sc.cap = want;
    }
  }
  *p = sc.p;
  return 0;
}

hipStream_t stream(int i) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  DevState& d = g_dev[dev];
  if (!d.st[i & 1]) (void)hipStreamCreateWithFlags(&d.st[i & 1], hipStreamNonBlocking);
  return d.st[i & 1];
}

void drop_device_state() {
  for (auto& kv : g_dev) {
    (void)hipSetDevice(kv.first);
    for (auto& s : kv.second.slots) {
      if (s.p) (void)hipFree(s.p);
      s = Scratch{};
    }
    for (auto& st : kv.second.st) {
      if (st) (void)hipStreamDestroy(st);
      st = nullptr;
    }
  }
  g_dev.clear();
}

}  // namespace hostpath
}  // namespace ssq

using namespace ssq;
using namespace ssq::hostpath;

extern "C" {

int ssq_pinned_alloc(void** p, int64_t bytes) {
  if (!p) SSQ_FAIL("p is NULL");
  *p = nullptr;
  if (bytes < 0) SSQ_FAIL("negative size");
  const long long cap = round_cap(bytes);
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_free.lower_bound(cap);
  if (it != g_free.end() && it->first <= cap + cap / 4 + (4LL << 20)) {      // close fit: reuse
    *p = it->second;
    g_live[*p] = it->first;
    g_free_bytes -= it->first;
    g_free.erase(it);
    return 0;
  }
  void* q = nullptr;
  hipError_t e = hipHostMalloc(&q, (size_t)cap, hipHostMallocDefault);
  if (e != hipSuccess) {                                                    // make room and retry once
    (void)hipGetLastError();
    for (auto& kv : g_free) (void)hipHostFree(kv.second);
    g_free.clear();
    g_free_bytes = 0;
    SSQ_HIP(hipHostMalloc(&q, (size_t)cap, hipHostMallocDefault));
  }
  g_live[q] = cap;
  *p = q;
  return 0;
}

int ssq_pinned_free(void* p) {
  if (!p) return 0;
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_live.find(p);
  if (it == g_live.end()) SSQ_FAIL("ssq_pinned_free: not a live block of the pool");
  const long long cap = it->second;
  g_live.erase(it);
  g_free.emplace(cap, p);
  g_free_bytes += cap;
  while (g_free_bytes > g_free_limit && !g_free.empty()) {                  // trim the largest idle blocks first
    auto last = std::prev(g_free.end());
    (void)hipHostFree(last->second);
    g_free_bytes -= last->first;
    g_free.erase(last);
  }
  return 0;
}

int ssq_host_cache_clear(void) {
  {
    std::lock_guard<std::mutex> lk(mutex());
    clear_stft_plans();
    clear_cwt_plans();
    drop_device_state();
  }
  std::lock_guard<std::mutex> lk(g_pin_mu);
  for (auto& kv : g_free) (void)hipHostFree(kv.second);
  g_free.clear();
  g_free_bytes = 0;
  return 0;
}

int ssq_host_cache_limit(int64_t idle_pinned_bytes) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_free_limit = idle_pinned_bytes < 0 ? 0 : idle_pinned_bytes;
  return 0;
}

int ssq_host_cache_stats(int64_t* live_pinned_bytes, int64_t* idle_pinned_bytes) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  long long live = 0;
  for (auto& kv : g_live) live += kv.second;
  if (live_pinned_bytes) *live_pinned_bytes = live;
  if (idle_pinned_bytes) *idle_pinned_bytes = g_free_bytes;
  return 0;
}

}  // extern "C"
