// AddressSanitizer + UBSan harness over the library's host-side fp64 helpers (csrc/host_math.h: window sizing, spectral
// diff-window, any-length host FFT incl. Bluestein, scale / frequency vectors).  Built and run on the CPU by
// tests/test_sanitizers.py (GPU sanitizers are not available on this pool; the device code is exercised by the parity
// tests).  Exit code 0 = no sanitizer report and the self-checks hold.
#include <cstdio>
#include <cstdlib>

#include "../../ssqueeze_rs_amd/csrc/host_math.h"

using namespace ssq::host;

static int fails = 0;
#define CHECK(c)                                          \
  do {                                                    \
    if (!(c)) {                                           \
      std::fprintf(stderr, "check failed: %s\n", #c);     \
      ++fails;                                            \
    }                                                     \
  } while (0)

static double dft_err(int64_t n) {
  std::vector<cd> a((size_t)n), ref((size_t)n);
  for (int64_t i = 0; i < n; ++i) a[i] = cd(std::sin(0.37 * (double)i) + 0.1, std::cos(1.3 * (double)i));
  for (int64_t k = 0; k < n; ++k) {
    cd s(0, 0);
    for (int64_t j = 0; j < n; ++j) s += a[j] * std::polar(1.0, -2.0 * M_PI * (double)((j * k) % n) / (double)n);
    ref[k] = s;
  }
  std::vector<cd> b = a;
  fft_any(b, -1);
  double e = 0, m = 0;
  for (int64_t k = 0; k < n; ++k) {
    e = std::fmax(e, std::abs(b[k] - ref[k]));
    m = std::fmax(m, std::abs(ref[k]));
  }
  fft_any(b, +1);                                       // round trip
  for (int64_t k = 0; k < n; ++k) e = std::fmax(e, std::abs(b[k] / (double)n - a[k]) * m);
  return e / m;
}

int main() {
  for (int64_t n : {1, 2, 3, 8, 17, 100, 256, 1000, 1024, 8193, 10007}) CHECK(dft_err(n) < 1e-10);
  for (int64_t L : {1, 5, 200, 256, 300}) {
    std::vector<double> w((size_t)L, 1.0);
    std::vector<double> s = size_window(w.data(), L, 256);
    CHECK((int64_t)s.size() == 256);
    double sum = 0;
    for (double v : s) sum += v;
    CHECK(sum == (double)(L < 256 ? L : 256));
  }
  for (int64_t n : {2, 7, 64, 1000}) {
    std::vector<double> w((size_t)n);
    for (int64_t i = 0; i < n; ++i) w[i] = 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)i / (double)n);
    std::vector<double> d = diff_window(w.data(), n);
    CHECK((int64_t)d.size() == n);
    for (double v : d) CHECK(std::isfinite(v));
  }
  CHECK(next_power_of_2(0) == 1 && next_power_of_2(1) == 1 && next_power_of_2(1000 + 500) == 2048);
  CHECK(next_power_of_2((1 << 20) + (1 << 19)) == (1 << 21));
  for (int64_t N : {1, 2, 3, 4, 1000, 1 << 20})
    for (int64_t nv : {0, 1, 32}) {
      std::vector<double> s0 = log_scales(N, nv, false), s1 = log_scales(N, nv, true);
      CHECK(s0.size() == s1.size());
      for (size_t i = 0; i < s0.size(); ++i) CHECK(std::fabs(s0[i] - s1[i]) <= 1e-12 * s0[i]);
    }
  std::vector<double> f = cwt_ssq_freqs(256, 1.0 / 524288.0, 0.5, false), g = cwt_ssq_freqs(1, 0.1, 0.4, true);
  CHECK(f.size() == 256 && f[0] > 0 && f[255] <= 0.5000001 && g.size() == 1 && g[0] == 0.1);
  CHECK(cwt_ssq_freqs(0, 0.1, 0.4, false).empty());
  std::printf(fails ? "FAILED\n" : "ok\n");
  return fails ? 1 : 0;
}
