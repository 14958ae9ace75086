// api_common.hip -- library/device plumbing and the host-math entry points of the C-ABI
// (include/ssq_hip.h).
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ssq_hip.h"
#include "host_math.h"
#include "ssq_common.h"

namespace ssq {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* tune_env(const char* name) {
#ifdef SSQ_TUNING
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
}  // namespace ssq

using namespace ssq;

extern "C" {

const char* ssq_last_error(void) { return ssq::g_err.c_str(); }

const char* ssq_hello_from_bin(void) { return "Hello from ssqueeze!"; }   // lib.rs:16-19

int ssq_device_count(int* count) {
  if (!count) SSQ_FAIL("count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    ssq::set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return 2;
  }
  *count = n;
  return 0;
}

int ssq_set_device(int device) {
  SSQ_HIP(hipSetDevice(device));
  return 0;
}

int ssq_device_info(int* cu_count, int64_t* hbm_bytes, char* name, int name_len) {
  int dev = 0;
  SSQ_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SSQ_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  if (name && name_len > 0) {
    std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    std::strncpy(name, s.c_str(), (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  return 0;
}

// stft.rs:32-34 / ssq_stft.rs:182-184
int ssq_stft_shape(int64_t n_signal, int64_t n_fft, int64_t hop, int64_t* n_freqs, int64_t* n_frames) {
  if (n_signal <= 0) SSQ_FAIL("empty input signal (the reference panics on n_samples - n_fft underflow)");
  if (n_fft <= 0) SSQ_FAIL("n_fft must be positive");
  if (hop <= 0) SSQ_FAIL("attempt to divide by zero (hop length 0)");
  const int64_t padded = n_signal + n_fft - 1;
  if (n_frames) *n_frames = (padded - n_fft) / hop + 1;
  if (n_freqs) *n_freqs = n_fft / 2 + 1;
  return 0;
}

int ssq_cwt_pad_len(int64_t n_signal, int64_t* pad_len, int64_t* n1) {
  if (n_signal <= 0) SSQ_FAIL("empty input signal");
  const int64_t P = host::next_power_of_2(n_signal + n_signal / 2);   // cwt.rs:87
  if (pad_len) *pad_len = P;
  if (n1) *n1 = (P - n_signal) / 2;                                    // cwt.rs:98
  return 0;
}

int ssq_log_scales(int64_t n_signal, int64_t nv, int simd_variant, int64_t* na, double* scales) {
  std::vector<double> s = host::log_scales(n_signal, nv, simd_variant != 0);
  if (na) *na = (int64_t)s.size();
  if (scales) std::memcpy(scales, s.data(), s.size() * sizeof(double));
  return 0;
}

int ssq_size_window(const double* window, int64_t win_n, int64_t n_fft, double* out) {
  if (!window || !out || win_n < 0 || n_fft <= 0) SSQ_FAIL("bad window arguments");
  std::vector<double> w = host::size_window(window, win_n, n_fft);
  std::memcpy(out, w.data(), w.size() * sizeof(double));
  return 0;
}

int ssq_diff_window(const double* window, int64_t n_fft, double* out) {
  if (!window || !out || n_fft <= 0) SSQ_FAIL("bad window arguments");
  std::vector<double> d = host::diff_window(window, n_fft);
  std::memcpy(out, d.data(), d.size() * sizeof(double));
  return 0;
}

int ssq_cwt_ssq_freqs(const double* scales, int64_t na, int64_t n_signal, double dt, int maprange,
                      int freq_dist, double* ssq_freqs) {
  if (!scales || !ssq_freqs || na <= 0) SSQ_FAIL("index out of bounds: scales is empty (ssq_cwt.rs:459)");
  double fmin, fmax;
  if (maprange == SSQ_MAPRANGE_MAXIMAL) {          // ssq_cwt.rs:451-455
    const double dT = (double)n_signal * dt;
    fmin = 1.0 / dT;
    fmax = 0.5 / dt;
  } else {                                         // ssq_cwt.rs:456-460
    fmin = 1.0 / scales[na - 1];
    fmax = 1.0 / scales[0];
  }
  std::vector<double> f = host::cwt_ssq_freqs(na, fmin, fmax, freq_dist == SSQ_FREQS_LINEAR);
  std::memcpy(ssq_freqs, f.data(), f.size() * sizeof(double));
  return 0;
}

// ---- device memory / stream / event plumbing --------------------------------
int ssq_dev_malloc(void** ptr, int64_t bytes) {
  if (!ptr) SSQ_FAIL("ptr is NULL");
  *ptr = nullptr;
  if (bytes <= 0) return 0;
  SSQ_HIP(hipMalloc(ptr, (size_t)bytes));
  return 0;
}
int ssq_dev_free(void* ptr) {
  if (ptr) SSQ_HIP(hipFree(ptr));
  return 0;
}
int ssq_dev_memset(void* ptr, int value, int64_t bytes, void* stream) {
  if (bytes > 0) SSQ_HIP(hipMemsetAsync(ptr, value, (size_t)bytes, (hipStream_t)stream));
  return 0;
}
int ssq_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream) {
  if (bytes > 0) SSQ_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}
int ssq_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream) {
  if (bytes > 0) SSQ_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}
int ssq_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream) {
  if (bytes > 0) SSQ_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
int ssq_stream_create(void** stream) {
  if (!stream) SSQ_FAIL("stream is NULL");
  hipStream_t s;
  SSQ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return 0;
}
int ssq_stream_destroy(void* stream) {
  if (stream) SSQ_HIP(hipStreamDestroy((hipStream_t)stream));
  return 0;
}
int ssq_stream_sync(void* stream) {
  SSQ_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
int ssq_device_sync(void) {
  SSQ_HIP(hipDeviceSynchronize());
  return 0;
}
int ssq_event_create(void** event) {
  if (!event) SSQ_FAIL("event is NULL");
  hipEvent_t e;
  SSQ_HIP(hipEventCreate(&e));
  *event = (void*)e;
  return 0;
}
int ssq_event_destroy(void* event) {
  if (event) SSQ_HIP(hipEventDestroy((hipEvent_t)event));
  return 0;
}
int ssq_event_record(void* event, void* stream) {
  SSQ_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return 0;
}
int ssq_event_sync(void* event) {
  SSQ_HIP(hipEventSynchronize((hipEvent_t)event));
  return 0;
}
int ssq_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) SSQ_FAIL("ms is NULL");
  SSQ_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}


// ---- HIP graphs: capture a sequence of plan execs on a stream once, replay it with one launch ----------------------
// The plan execs take device pointers and a stream and neither allocate nor synchronise (the ssq_cwt exec forks its side
// stream by events, which joins the capture), so a whole per-signal or per-chunk pipeline becomes ONE graph launch: what
// launch-bound callers want (one 2^20-sample ssq_stft is 17 us of which the kernel is ~7; a chunked multi-channel run is
// a long chain of small launches).  Run the sequence once before capturing it (lazy one-time setup inside the plans).
int ssq_graph_capture_begin(void* stream) {
  if (!stream) SSQ_FAIL("capture needs an explicit stream (not the default stream)");
  SSQ_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}
int ssq_graph_capture_end(void* stream, void** graph_exec) {
  if (!stream || !graph_exec) SSQ_FAIL("stream or graph_exec is NULL");
  *graph_exec = nullptr;
  hipGraph_t g = nullptr;
  SSQ_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ge = nullptr;
  const hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  SSQ_HIP(e);
  *graph_exec = ge;
  return 0;
}
int ssq_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) SSQ_FAIL("graph_exec is NULL");
  SSQ_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return 0;
}
int ssq_graph_destroy(void* graph_exec) {
  if (!graph_exec) return 0;
  SSQ_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}

}  // extern "C"

extern "C" int ssq_build_has_tuning(void) {
#ifdef SSQ_TUNING
  return 1;
#else
  return 0;
#endif
}
