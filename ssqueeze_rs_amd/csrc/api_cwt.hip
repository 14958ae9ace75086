// api_cwt.hip -- CWT-family plans and host entry points of the C-ABI (include/ssq_hip.h).
// Replaces the PyO3 functions `cwt` / `cwt_simd` (rust/src/spectral/cwt.rs:46-144,
// cwt_simd.rs:52-150) and `ssq_cwt` (rust/src/spectral/ssq_cwt.rs:244-493).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ssq_hip.h"
#include <mutex>

#include "cwt_kernels.h"
#include "host_cache.h"
#include "host_math.h"

using namespace ssq;

struct ssq_cwt_plan {
  int dtype = SSQ_F32;
  long long N = 0, P = 0, n1 = 0;
  int logP = 0, log_p1 = 0, log_p2 = 0;
  bool two_step = false, naive = false;
  bool big = false;            // P > 2^24: batched generic device FFT instead of the tile transforms
  int wavelet = 0, padtype = 0, na = 0;
  // upstream-parity mode (SSQ_VARIANT_UPSTREAM; old/ssqueezepy): p2up padding, normalised wavelets (wavelet table codes
  // 2 = GMW(gamma = wp0, beta = wp1), 3 = Morlet(mu = wp0)) with a halved Nyquist bin, the generic transforms only.
  // ups_freqs / ups_nv: the ascending frequencies and the voices per octave the next ssq exec bins with.
  int variant = 0, table_code = 0;
  double wp0 = 0.0, wp1 = 0.0;
  std::vector<double> ups_freqs;
  int ups_nv = 0;
  double dt = 1.0;
  std::vector<double> scales;
  void* d_psih = nullptr;      // wavelet table: scale s at psi_off[s], band[s] entries of T (zero beyond: not stored)
  long long* d_psi_off = nullptr;
  void* d_tw1 = nullptr;       // W_{P1}^i
  void* d_tw2 = nullptr;       // W_{P2}^i
  void* d_twz = nullptr;       // W_Q^i for Q = 16 .. 2048 (mode Z) at twz_off[log2 Q] (fp64: + the compact per-pass tables)
  long long twz_off[16] = {0};
  void* d_f2a = nullptr;       // step A: [P1][C] W_P^(c k1)
  void* d_f2z[13] = {};        // mode Z, per log2 Q: [C][Q] W_P^(c k)
  std::vector<int> zoom_logq;  // per scale: log2 Q of the band-limited single-pass path, 0 = two-step path
  std::vector<int> band;       // per scale: psih_s[k] == 0 for k >= band
  int* d_band = nullptr;
  void* d_twhi = nullptr;      // W_P^(i<<12)
  void* d_twlo = nullptr;      // W_P^i, i < 4096
  void* d_scale_l1 = nullptr;  // [na] 1/P
  void* d_scale_l2 = nullptr;  // [na] sqrt(a)/P
  double* d_scales = nullptr;
  int chunk = 1;               // scales per inverse-FFT chunk
  // register-core path of the two-step scales (cwt_reg.hip): fp32, P = 2^20 or 2^21
  bool reg = false;
  int reg_D = 0;               // P / 2^20
  void* d_psiT = nullptr;      // transposed wavelet table of the two-step scales
  long long* d_psiT_off = nullptr;
  int* d_psiT_A = nullptr;
  void* d_tw1024 = nullptr;    // W_1024^j
  void* d_tw20 = nullptr;      // W_{2^20}^i, i < 1024
  int n_cus = 256;
  // time-tiled (overlap-save) ssq path of the short-wavelet scales [os_s0, os_s1) (cwt_os.hip); empty = off
  // [os_s0, os_mid): 4096-point tiles (halo 1024, two blocks per CU); [os_mid, os_s1): 8192-point tiles (halo 2048)
  // [os_s1, os_d1): 16-fold decimated 8192-point tiles (long wavelets that are band-limited far below Nyquist)
  int os_s0 = 0, os_mid = 0, os_s1 = 0, os_d1 = 0;
  void* d_osHd = nullptr;      // [os_d1 - os_s1][4096] psih on the 131072-point grid
  // [os_z0, os_z1): band-limited scales (mode Z's) of plans with P = 2 N by full-circle phase blocks (cwt_os.hip)
  int os_z0 = 0, os_z1 = 0;
  void* d_osHz = nullptr;      // [os_z1 - os_z0][2048] psih on the P-point grid
  // [os_a0, os_s0): the finest scales on tiles of the ANALYTIC signal (wavelet spectrum continued beyond Nyquist)
  int os_a0 = 0;
  void* d_osHa = nullptr;      // [os_s0 - os_a0][4096] psih on the 4096-point grid, all bins
  void* d_xa_psiT = nullptr;   // the pseudo-scale "1 for k <= P/2" in the register-core layout (xa = ifft of X times it)
  long long* d_xa_off = nullptr;   // its one-entry tables: psiT_off, psi_off | A, band | 1/P
  int* d_xa_int = nullptr;
  float* d_xa_scale = nullptr;
  std::vector<char> os_mask;   // per scale: 1 = computed by the time-tile family inside ssq_cwt
  bool all_tiled = false;      // every scale is: the ssq path needs no Wx / dWx workspaces at all
  void* d_osH4 = nullptr;      // [os_mid - os_s0][2048] psih on the 4096-point grid
  void* d_osH = nullptr;       // [os_s1 - os_mid][4096] psih on the 8192-point grid
  // ssq path of two-step plans: Tx is cleared on a side stream while the transforms run
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool can_fuse_ssq() const { return two_step && !naive && na <= 32767; }
  // SSQ_CWT_FUSED=1 selects the fused step B / mode Z (phase transform + bin in the store phase, Wx + 16-bit row index
  // out, dWx never in memory: 37 % less traffic between the transforms and the reassignment).  Measured on C4
  // (profiles/r02_ab_cwt_fused.txt): 8.64 ms against 8.16 ms for the unfused narrow step B with two blocks per CU --
  // the tile kernel's phases add, so the extra store-phase arithmetic costs more than the bytes it saves.  Default off.
  bool fused_ssq() const {
    if (!can_fuse_ssq()) return false;
    const char* e = tune_env("SSQ_CWT_FUSED");
    return e && std::atoi(e) != 0;
  }
};

namespace {

const long double kPI = 3.14159265358979323846264338327950288L;
constexpr long long kZoomTabQ = 4096;   // twiddle table extent of the single-pass (mode Z) transforms (tile kernel: LOGM <= 12)
// longest single-pass (mode Z) transform actually used (SSQ_CWT_ZMAXQ, a power of two <= 4096; read at plan creation)
long long zoom_max_q() {
  long long q = 2048;
  if (const char* e = tune_env("SSQ_CWT_ZMAXQ")) {
    const long long v = std::atoll(e);
    if (v >= 16 && v <= kZoomTabQ && (v & (v - 1)) == 0) q = v;
  }
  return q;
}

// SSQ_CWT_GROUP (read per call): n > 0 reassigns every n scales right behind their transforms (while their Wx / dWx are
// still in the Infinity Cache) instead of once per call; default 0: measured 5 % slower on C4, profiles/r02_ab_cwt_group.txt
int ssq_group_env(int dflt) {
  const char* e = tune_env("SSQ_CWT_GROUP");
  if (!e) return dflt;
  const int v = std::atoi(e);
  return v < 0 ? dflt : v;
}

// Smallest w beyond which the wavelet table entry is exactly zero in T (the table is evaluated in fp64 and
// rounded once to T: csrc/cwt_kernels.hip::wavelet_table_kernel), with margin.
//   morlet (cwt.rs:497-520): exp(-(w-6)^2/2) underflows T    -> fp32 (1.4e-45): |w-6| > 14.4 ; fp64: |w-6| > 38.6
//   gmw    (cwt.rs:522-542): exp(60 ln w - w^3) underflows T -> fp32: w > 6.0 ; fp64: w > 9.6
double wavelet_support(int wavelet, int dtype) {
  if (wavelet == SSQ_WAVELET_MORLET) return dtype == SSQ_F32 ? 21.0 : 46.0;
  return dtype == SSQ_F32 ? 6.5 : 10.0;
}

// the same bound for the upstream wavelets (codes 2, 3 of wavelet_table_kernel): scan upward from the peak until the
// value rounds to zero in T
double upstream_support(int wavelet, double p0, double p1, int dtype) {
  if (wavelet == SSQ_WAVELET_MORLET) return p0 + (dtype == SSQ_F32 ? 15.0 : 39.5);   // exp(-(w - mu)^2 / 2) underflows
  const double wc = std::exp((1.0 / p0) * (std::log(p1) - std::log(p0)));
  const double c0 = -p1 * std::log(wc) + std::pow(wc, p0);
  const double lim = dtype == SSQ_F32 ? -104.0 : -746.0;                             // ln of the smallest denormal
  double w = wc;
  for (int it = 0; it < 100000; ++it) {
    w += 0.01 * wc;
    if (c0 + p1 * std::log(w) - std::pow(w, p0) < lim) break;
  }
  return w;
}

template <typename T>
int upload_tw(void** dst, long long n, long long P_total, long long stride) {
  // dst[i] = exp(-2*pi*i * (i*stride) / P_total), i in [0, n)
  std::vector<cpx<T>> h((size_t)(n > 0 ? n : 1));
  for (long long i = 0; i < n; ++i) {
    const long double ang = 2.0L * kPI * (long double)(i * stride) / (long double)P_total;
    h[i] = {(T)cosl(ang), (T)(-sinl(ang))};
  }
  SSQ_HIP(hipMalloc(dst, sizeof(cpx<T>) * h.size()));
  SSQ_HIP(hipMemcpy(*dst, h.data(), sizeof(cpx<T>) * h.size(), hipMemcpyHostToDevice));
  return 0;
}

template <typename T>
int build_tables(ssq_cwt_plan* pl) {
  const long long P1 = 1LL << pl->log_p1, P2 = 1LL << pl->log_p2;
  // W_M table, and behind it (fp64 plans) the per-pass compact tables [m][k] = exp(-2 pi i k m / (NS R)) of the tile transform
  auto upload_tw_tile = [&](void** dst, int logm) -> int {
    const long long M = 1LL << logm;
    const long long extra = (sizeof(T) == 8 && logm >= 4 && logm <= 12) ? cwt_tw_compact_elems(logm) : 0;
    std::vector<cpx<T>> h((size_t)(M + extra));
    for (long long i = 0; i < M; ++i) {
      const long double ang = 2.0L * kPI * (long double)i / (long double)M;
      h[(size_t)i] = {(T)cosl(ang), (T)(-sinl(ang))};
    }
    if (extra > 0) {
      long long off = M;
      for (int P = 1; P < num_passes(logm); ++P) {
        const int R = pass_radix(logm, P), NS = pass_ns(logm, P);
        for (int m = 0; m < R; ++m)
          for (int k = 0; k < NS; ++k) {
            const long double ang = 2.0L * kPI * (long double)((long long)k * m) / (long double)((long long)NS * R);
            h[(size_t)(off + (long long)m * NS + k)] = {(T)cosl(ang), (T)(-sinl(ang))};
          }
        off += (long long)R * NS;
      }
    }
    SSQ_HIP(hipMalloc(dst, sizeof(cpx<T>) * h.size()));
    SSQ_HIP(hipMemcpy(*dst, h.data(), sizeof(cpx<T>) * h.size(), hipMemcpyHostToDevice));
    return 0;
  };
  if (int rc = upload_tw_tile(&pl->d_tw1, pl->log_p1)) return rc;
  if (int rc = upload_tw_tile(&pl->d_tw2, pl->log_p2)) return rc;
  if (pl->two_step) {
    // in-tile factors of the W_P twiddle (the other factor, W_P^(t0 k), is formed per tile in LDS)
    auto upload_f2 = [&](void** dst, long long rows, long long cols, bool row_is_k) -> int {
      std::vector<cpx<T>> h((size_t)(rows * cols));
      for (long long i = 0; i < rows; ++i)
        for (long long j = 0; j < cols; ++j) {
          const long double ang = 2.0L * kPI * (long double)(i * j) / (long double)pl->P;   // c*k either way
          h[(size_t)(i * cols + j)] = {(T)cosl(ang), (T)(-sinl(ang))};
        }
      (void)row_is_k;
      SSQ_HIP(hipMalloc(dst, sizeof(cpx<T>) * h.size()));
      SSQ_HIP(hipMemcpy(*dst, h.data(), sizeof(cpx<T>) * h.size(), hipMemcpyHostToDevice));
      return 0;
    };
    if (int rc = upload_f2(&pl->d_f2a, P1, cwt_tile_rows<T>(pl->log_p1), true)) return rc;
    for (int lq = 4; lq <= 12; ++lq) {
      bool used = false;
      for (int v : pl->zoom_logq) used = used || v == lq;
      if (!used) continue;
      if (int rc = upload_f2(&pl->d_f2z[lq], cwt_tile_rows<T>(lq), 1LL << lq, false)) return rc;
    }
    // W_Q for Q = 16 .. kZoomTabQ, each followed (fp64 plans) by its per-pass compact tables; pl->twz_off[log2 Q] says where
    std::vector<cpx<T>> hz;
    for (int lq = 4; (1LL << lq) <= kZoomTabQ; ++lq) {
      const long long Q = 1LL << lq;
      pl->twz_off[lq] = (long long)hz.size();
      for (long long i = 0; i < Q; ++i) {
        const long double ang = 2.0L * kPI * (long double)i / (long double)Q;
        hz.push_back({(T)cosl(ang), (T)(-sinl(ang))});
      }
      if (sizeof(T) == 8) {
        for (int P = 1; P < num_passes(lq); ++P) {
          const int R = pass_radix(lq, P), NS = pass_ns(lq, P);
          for (int m = 0; m < R; ++m)
            for (int k = 0; k < NS; ++k) {
              const long double ang = 2.0L * kPI * (long double)((long long)k * m) / (long double)((long long)NS * R);
              hz.push_back({(T)cosl(ang), (T)(-sinl(ang))});
            }
        }
      }
    }
    SSQ_HIP(hipMalloc(&pl->d_twz, sizeof(cpx<T>) * hz.size()));
    SSQ_HIP(hipMemcpy(pl->d_twz, hz.data(), sizeof(cpx<T>) * hz.size(), hipMemcpyHostToDevice));
  }
  const long long nlo = pl->P < 4096 ? pl->P : 4096;
  const long long nhi = pl->P < 4096 ? 1 : pl->P / 4096;
  if (int rc = upload_tw<T>(&pl->d_twlo, nlo, pl->P, 1)) return rc;
  if (int rc = upload_tw<T>(&pl->d_twhi, nhi, pl->P, 4096)) return rc;
  std::vector<T> s1((size_t)pl->na), s2((size_t)pl->na);
  const double norm = 1.0 / (double)pl->P;                              // cwt.rs:251
  for (int i = 0; i < pl->na; ++i) {
    s1[i] = (T)norm;
    s2[i] = (T)(norm * std::sqrt(pl->scales[i]));                       // cwt.rs:253
  }
  SSQ_HIP(hipMalloc(&pl->d_scale_l1, sizeof(T) * (pl->na > 0 ? pl->na : 1)));
  SSQ_HIP(hipMalloc(&pl->d_scale_l2, sizeof(T) * (pl->na > 0 ? pl->na : 1)));
  SSQ_HIP(hipMalloc((void**)&pl->d_scales, sizeof(double) * (pl->na > 0 ? pl->na : 1)));
  SSQ_HIP(hipMalloc((void**)&pl->d_band, sizeof(int) * (pl->na > 0 ? pl->na : 1)));
  if (pl->na > 0) SSQ_HIP(hipMemcpy(pl->d_band, pl->band.data(), sizeof(int) * pl->na, hipMemcpyHostToDevice));
  if (pl->na > 0) {
    SSQ_HIP(hipMemcpy(pl->d_scale_l1, s1.data(), sizeof(T) * pl->na, hipMemcpyHostToDevice));
    SSQ_HIP(hipMemcpy(pl->d_scale_l2, s2.data(), sizeof(T) * pl->na, hipMemcpyHostToDevice));
    SSQ_HIP(hipMemcpy(pl->d_scales, pl->scales.data(), sizeof(double) * pl->na, hipMemcpyHostToDevice));
  }
  std::vector<long long> off((size_t)(pl->na > 0 ? pl->na : 1), 0);
  long long total = 0;
  int max_band = 1;
  for (int i = 0; i < pl->na; ++i) {
    off[(size_t)i] = total;
    total += pl->band[(size_t)i];
    if (pl->band[(size_t)i] > max_band) max_band = pl->band[(size_t)i];
  }
  SSQ_HIP(hipMalloc(&pl->d_psih, sizeof(T) * (size_t)(total > 0 ? total : 1)));
  SSQ_HIP(hipMalloc((void**)&pl->d_psi_off, sizeof(long long) * off.size()));
  SSQ_HIP(hipMemcpy(pl->d_psi_off, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice));
  if (pl->na > 0) {
    SSQ_HIP(launch_wavelet_table<T>((T*)pl->d_psih, pl->d_psi_off, pl->d_band, max_band, pl->d_scales, pl->na, pl->P,
                                    pl->variant ? pl->table_code : pl->wavelet, nullptr, pl->wp0, pl->wp1));
    SSQ_HIP(hipDeviceSynchronize());
  }
  if constexpr (sizeof(T) == 4) {
    if (pl->os_s0 > pl->os_a0) {                               // analytic-input tiles (register-core plans only)
      const int nA = pl->os_s0 - pl->os_a0;
      SSQ_HIP(hipMalloc(&pl->d_osHa, sizeof(float) * 4096 * (size_t)nA));
      SSQ_HIP(launch_cwt_os_table((float*)pl->d_osHa, pl->d_scales, pl->os_a0, nA, pl->wavelet, 4, 0, true, nullptr));
      const long long half = pl->P / 2;
      const long long bnd = half + 1 < (1LL << 20) ? half + 1 : (1LL << 20);
      const int A = (int)((bnd + 1023) / 1024);
      std::vector<float> ones((size_t)A * 1024);
      for (int b = 0; b < 1024; ++b)
        for (int aa = 0; aa < A; ++aa) ones[(size_t)b * A + aa] = (1024LL * aa + b <= half) ? 1.0f : 0.0f;
      SSQ_HIP(hipMalloc(&pl->d_xa_psiT, sizeof(float) * ones.size()));
      SSQ_HIP(hipMemcpy(pl->d_xa_psiT, ones.data(), sizeof(float) * ones.size(), hipMemcpyHostToDevice));
      const long long offs[2] = {0, -half};                    // psiT_off | psi_off: psih[psi_off + P/2] = entry 0 = 1
      const int ints[2] = {A, (int)(half + 1)};                // A | band
      const float sc = (float)(1.0 / (double)pl->P);
      SSQ_HIP(hipMalloc((void**)&pl->d_xa_off, sizeof(offs)));
      SSQ_HIP(hipMalloc((void**)&pl->d_xa_int, sizeof(ints)));
      SSQ_HIP(hipMalloc((void**)&pl->d_xa_scale, sizeof(float)));
      SSQ_HIP(hipMemcpy(pl->d_xa_off, offs, sizeof(offs), hipMemcpyHostToDevice));
      SSQ_HIP(hipMemcpy(pl->d_xa_int, ints, sizeof(ints), hipMemcpyHostToDevice));
      SSQ_HIP(hipMemcpy(pl->d_xa_scale, &sc, sizeof(float), hipMemcpyHostToDevice));
      SSQ_HIP(hipDeviceSynchronize());
    }
    if (pl->os_z1 > pl->os_z0) {
      const int nz = pl->os_z1 - pl->os_z0;
      SSQ_HIP(hipMalloc(&pl->d_osHz, sizeof(float) * 2048 * (size_t)nz));
      SSQ_HIP(launch_cwt_os_table((float*)pl->d_osHz, pl->d_scales, pl->os_z0, nz, pl->wavelet, 4, pl->logP - 12, false, nullptr));
      SSQ_HIP(hipDeviceSynchronize());
      if (!pl->reg && !(pl->os_s1 > pl->os_s0))
        if (int rc = upload_tw<float>(&pl->d_tw1024, 1024, 1024, 1)) return rc;
    }
    if (pl->os_s1 > pl->os_s0) {                               // time-tiled ssq path (any two-step fp32 plan)
      const int n4 = pl->os_mid - pl->os_s0, n8 = pl->os_s1 - pl->os_mid;
      if (!pl->reg)
        if (int rc = upload_tw<float>(&pl->d_tw1024, 1024, 1024, 1)) return rc;
      SSQ_HIP(hipMalloc(&pl->d_osH4, sizeof(float) * 2048 * (size_t)(n4 > 0 ? n4 : 1)));
      SSQ_HIP(hipMalloc(&pl->d_osH, sizeof(float) * 4096 * (size_t)(n8 > 0 ? n8 : 1)));
      const int nd = pl->os_d1 - pl->os_s1;
      SSQ_HIP(hipMalloc(&pl->d_osHd, sizeof(float) * 4096 * (size_t)(nd > 0 ? nd : 1)));
      SSQ_HIP(launch_cwt_os_table((float*)pl->d_osH4, pl->d_scales, pl->os_s0, n4, pl->wavelet, 4, 0, false, nullptr));
      SSQ_HIP(launch_cwt_os_table((float*)pl->d_osH, pl->d_scales, pl->os_mid, n8, pl->wavelet, 8, 0, false, nullptr));
      SSQ_HIP(launch_cwt_os_table((float*)pl->d_osHd, pl->d_scales, pl->os_s1, nd, pl->wavelet, 8, kOsLogDec, false, nullptr));
      SSQ_HIP(hipDeviceSynchronize());
    }
    if (pl->reg && pl->na > 0) {
      std::vector<int> A((size_t)pl->na, 0);
      std::vector<long long> offT((size_t)pl->na, 0);
      long long totT = 0;
      int max_A = 0;
      for (int i = 0; i < pl->na; ++i) {
        if (pl->zoom_logq[(size_t)i] != 0) continue;             // band-limited scales keep the single-pass path
        long long b = pl->band[(size_t)i];
        if (b > (1LL << 20)) b = 1LL << 20;                      // D = 2: k = P/2 is added separately
        A[(size_t)i] = (int)((b + 1023) / 1024);
        offT[(size_t)i] = totT;
        totT += (long long)A[(size_t)i] * 1024;
        if (A[(size_t)i] > max_A) max_A = A[(size_t)i];
      }
      SSQ_HIP(hipMalloc(&pl->d_psiT, sizeof(float) * (size_t)(totT > 0 ? totT : 1)));
      SSQ_HIP(hipMalloc((void**)&pl->d_psiT_off, sizeof(long long) * A.size()));
      SSQ_HIP(hipMalloc((void**)&pl->d_psiT_A, sizeof(int) * A.size()));
      SSQ_HIP(hipMemcpy(pl->d_psiT_off, offT.data(), sizeof(long long) * A.size(), hipMemcpyHostToDevice));
      SSQ_HIP(hipMemcpy(pl->d_psiT_A, A.data(), sizeof(int) * A.size(), hipMemcpyHostToDevice));
      SSQ_HIP(launch_cwt_reg_table((float*)pl->d_psiT, pl->d_psiT_off, pl->d_psiT_A, max_A, (const float*)pl->d_psih,
                                   pl->d_psi_off, pl->d_band, pl->na, nullptr));
      SSQ_HIP(hipDeviceSynchronize());
      if (int rc = upload_tw<float>(&pl->d_tw1024, 1024, 1024, 1)) return rc;
      if (int rc = upload_tw<float>(&pl->d_tw20, 1024, 1LL << 20, 1)) return rc;
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        pl->n_cus = prop.multiProcessorCount;
    }
  }
  return 0;
}

struct WsLayout {
  long long xh = 0, ybuf = 0, w = 0, dw = 0, xc = 0, os_xs = 0, xa = 0, total = 0;
};
WsLayout ws_layout(const ssq_cwt_plan* pl) {
  const long long csz = pl->dtype == SSQ_F32 ? 8 : 16;
  auto align = [](long long v) { return (v + 255) / 256 * 256; };
  WsLayout L;
  long long off = 0;
  L.xh = off;
  off += align(pl->P * csz);
  L.ybuf = off;
  off += align(pl->big ? 2LL * pl->chunk * 2 * pl->P * csz              // spectra of a chunk + the FFT's ping-pong half
                       : (pl->two_step ? (long long)pl->chunk * 2 * pl->P * csz : 0));
  L.w = off;                   // (all scales time-tiled: Wx / dWx never exist in memory, nothing is reserved)
  if (!pl->all_tiled) off += align((long long)pl->na * pl->N * csz);
  L.dw = off;                  // fused ssq path: the 16-bit row indices K live here instead of dWx
  if (!pl->all_tiled) off += align((long long)pl->na * pl->N * csz);   // dWx (unfused path) or row indices (fused path)
  L.xc = off;                  // register-core path: the transposed, residue-twiddled spectrum
  if (pl->reg) off += align(((long long)pl->reg_D << 20) * csz);
  L.os_xs = off;               // time-tiled path: the tiles' spectra
  if (pl->os_s1 > pl->os_s0) {                                 // (any geometry: N rounded up to whole tiles; the
    const long long span = (long long)kOsL << kOsLogDec;       //  analytic-input tiles keep all F bins: twice)
    off += align(((pl->N + span - 1) / span) * span * 8 * (pl->os_s0 > pl->os_a0 ? 2 : 1));
  }
  L.xa = off;                  // the analytic signal on the padded grid
  if (pl->os_s0 > pl->os_a0) off += align(pl->P * 8);
  L.total = off;
  return L;
}

template <typename T>
CwtDev<T> base_dev(const ssq_cwt_plan* pl, char* ws) {
  const WsLayout L = ws_layout(pl);
  CwtDev<T> p;
  std::memset(&p, 0, sizeof(p));
  p.xh = (cpx<T>*)(ws + L.xh);
  p.ybuf = (cpx<T>*)(ws + L.ybuf);
  p.psih = (const T*)pl->d_psih;
  p.psi_off = pl->d_psi_off;
  p.tw_compact = sizeof(T) == 8 ? 1 : 0;      // d_tw1 / d_tw2 of fp64 plans carry the per-pass compact tables (build_tables)
  p.tw_hi = (const cpx<T>*)pl->d_twhi;
  p.tw_lo = (const cpx<T>*)pl->d_twlo;
  p.band = pl->d_band;
  p.tw_f2 = (const cpx<T>*)pl->d_f2a;
  p.n_signal = pl->N;
  p.P = pl->P;
  p.n1 = pl->n1;
  p.cols = pl->N;
  p.log_p1 = pl->log_p1;
  p.log_p2 = pl->log_p2;
  p.padtype = pl->padtype;
  p.n_kinds = 1;
  p.n_transforms = 1;
  // xi_k / dt = k * ((2*pi/P) / dt): base.rs:20-24 with cwt.rs:207
  p.xi_step = (T)((1.0 * (2.0 * M_PI) / (double)pl->P) / pl->dt);
  return p;
}

// parameter block of the register-core kernels from the tile kernels' one (fp32 plans with pl->reg only)
template <typename T>
CwtRegDev reg_dev(const ssq_cwt_plan* pl, const CwtDev<T>& p) {
  CwtRegDev r;
  std::memset(&r, 0, sizeof(r));
  if constexpr (sizeof(T) == 4) {
    r.xc = (cpx<float>*)((char*)p.xh + (ws_layout(pl).xc - ws_layout(pl).xh));
    r.psiT = (const float*)pl->d_psiT;
    r.psiT_off = pl->d_psiT_off;
    r.psiT_A = pl->d_psiT_A;
    r.tw1024 = (const cpx<float>*)pl->d_tw1024;
    r.tw20 = (const cpx<float>*)pl->d_tw20;
    r.ybuf = p.ybuf;
    r.xh = p.xh;
    r.psih = p.psih;
    r.psi_off = p.psi_off;
    r.band = p.band;
    r.tw_hi = p.tw_hi;
    r.tw_lo = p.tw_lo;
    r.out_scale = p.out_scale;
    r.Wx = p.Wx;
    r.dWx = p.dWx;
    r.n_signal = p.n_signal;
    r.P = p.P;
    r.n1 = p.n1;
    r.cols = p.cols;
    r.rpadded = p.rpadded;
    r.D = pl->reg_D;
    r.scale0 = p.scale0;
    r.n_kinds = p.n_kinds;
    r.n_transforms = p.n_transforms;
    r.xi_step = p.xi_step;
  }
  return r;
}

// forward FFT of the padded signal  (cwt.rs:87-95)
template <typename T>
int run_forward(const ssq_cwt_plan* pl, CwtDev<T> p, const T* d_x, hipStream_t st) {
  p.x = d_x;
  p.n_transforms = 1;
  if (pl->big) {
    SSQ_HIP(launch_cwt_big_fwd<T>(p, p.ybuf, st));
  } else if (pl->naive) {
    SSQ_HIP(launch_cwt_naive_fwd<T>(p, st));
  } else if (pl->two_step) {
    p.tw_m = (const cpx<T>*)pl->d_tw1;
    SSQ_HIP(launch_cwt_tile<T>(CWT_FWD_A, p, st));
    p.tw_m = (const cpx<T>*)pl->d_tw2;
    SSQ_HIP(launch_cwt_tile<T>(CWT_FWD_B, p, st));
    if (pl->reg) SSQ_HIP(launch_cwt_reg_prep(reg_dev<T>(pl, p), st));
  } else {
    p.tw_m = (const cpx<T>*)pl->d_tw1;
    SSQ_HIP(launch_cwt_tile<T>(CWT_FWD_S, p, st));
  }
  return 0;
}

// per-scale wavelet multiply + inverse FFT + normalise + unpad  (cwt.rs:228-310, :108-129)
// `after(s0, s1)` runs behind every group of scales whose outputs are complete (the ssq path reassigns there); with
// group > 0 the band-limited runs are cut into launches of at most `group` scales so that a group fits the cache.
struct NoAfter {
  int operator()(int, int) const { return 0; }
};
// Scales with skip[s] != 0 are left out (the ssq path computes them by time tiles, cwt_os.hip).
template <typename T, typename After = NoAfter>
int run_inverse(const ssq_cwt_plan* pl, CwtDev<T> p, cpx<T>* Wx, cpx<T>* dWx, bool l1_norm, bool rpadded,
                hipStream_t st, After after = After(), int group = 0, const char* skip = nullptr) {
  p.Wx = Wx;
  p.dWx = dWx;
  p.n_kinds = dWx ? 2 : 1;
  p.rpadded = rpadded ? 1 : 0;
  p.cols = rpadded ? pl->P : pl->N;
  p.out_scale = (const T*)(l1_norm ? pl->d_scale_l1 : pl->d_scale_l2);
  if (pl->big) {
    for (int c0 = 0; c0 < pl->na; c0 += pl->chunk) {
      const int ns = (pl->na - c0 < pl->chunk) ? pl->na - c0 : pl->chunk;
      p.scale0 = c0;
      p.n_transforms = ns * p.n_kinds;
      SSQ_HIP(launch_cwt_big_inv<T>(p, p.ybuf + (long long)pl->chunk * 2 * pl->P, st));
      if (int rc = after(c0, c0 + ns)) return rc;
    }
    return 0;
  }
  if (pl->naive) {
    p.scale0 = 0;
    p.n_transforms = pl->na * p.n_kinds;
    SSQ_HIP(launch_cwt_naive_inv<T>(p, p.n_transforms, st));
    return after(0, pl->na);
  }
  if (!pl->two_step) {
    p.scale0 = 0;
    p.n_transforms = pl->na * p.n_kinds;
    p.tw_m = (const cpx<T>*)pl->d_tw1;
    SSQ_HIP(launch_cwt_tile<T>(CWT_INV_S, p, st));
    return after(0, pl->na);
  }
  // runs of consecutive scales on the same path: band-limited ones (mode Z, grouped by Q) in one launch per run,
  // the others through the two-step transform in chunks whose ybuf stays inside the Infinity Cache
  int s0 = 0;
  while (s0 < pl->na) {
    if (skip && skip[s0]) {                                // (only two-step plans reach this loop)
      ++s0;
      continue;
    }
    const int lq = pl->zoom_logq[(size_t)s0];
    int s1 = s0 + 1;
    while (s1 < pl->na && pl->zoom_logq[(size_t)s1] == lq && !(skip && skip[s1])) ++s1;
    if (lq > 0) {
      CwtDev<T> z = p;
      z.log_p2 = lq;
      z.log_p1 = pl->logP - lq;
      z.tw_m = (const cpx<T>*)pl->d_twz + pl->twz_off[lq];
      z.tw_compact = sizeof(T) == 8 ? 1 : 0;
      z.tw_f2 = (const cpx<T>*)pl->d_f2z[lq];
      const int step = group > 0 ? group : s1 - s0;
      for (int c0 = s0; c0 < s1; c0 += step) {
        const int ns = (s1 - c0 < step) ? s1 - c0 : step;
        z.scale0 = c0;
        z.n_transforms = ns * p.n_kinds;
        SSQ_HIP(launch_cwt_tile<T>(CWT_INV_Z, z, st));
        if (int rc = after(c0, c0 + ns)) return rc;
      }
    } else {
      for (int c0 = s0; c0 < s1; c0 += pl->chunk) {
        const int ns = (s1 - c0 < pl->chunk) ? s1 - c0 : pl->chunk;
        p.scale0 = c0;
        p.n_transforms = ns * p.n_kinds;
        if (pl->reg) {
          SSQ_HIP(launch_cwt_reg_inv(reg_dev<T>(pl, p), pl->n_cus, st));
        } else if (ns == 1 && p.n_kinds == 2 && 2 * pl->P * (long long)sizeof(cpx<T>) > (192LL << 20)) {
          // one scale's TWO step buffers do not fit the 256 MB Infinity Cache (fp64 at P = 2^23: 268 MB): run the kinds one
          // after the other through ONE buffer (134 MB), so that step B reads what step A just wrote from the cache
          for (int kind = 0; kind < 2; ++kind) {
            p.tr0 = kind;
            p.n_transforms = 1;
            p.tw_m = (const cpx<T>*)pl->d_tw1;
            SSQ_HIP(launch_cwt_tile<T>(CWT_INV_A, p, st));
            p.tw_m = (const cpx<T>*)pl->d_tw2;
            SSQ_HIP(launch_cwt_tile<T>(CWT_INV_B, p, st));
          }
          p.tr0 = 0;
        } else {
          p.tw_m = (const cpx<T>*)pl->d_tw1;
          SSQ_HIP(launch_cwt_tile<T>(CWT_INV_A, p, st));
          p.tw_m = (const cpx<T>*)pl->d_tw2;
          SSQ_HIP(launch_cwt_tile<T>(CWT_INV_B, p, st));
        }
        if (int rc = after(c0, c0 + ns)) return rc;
      }
    }
    s0 = s1;
  }
  return 0;
}

// ssq path of two-step plans: step A as above, then the FUSED step B / mode Z (Wx + row index out, dWx stays on chip)
template <typename T>
int run_inverse_ssq(const ssq_cwt_plan* pl, CwtDev<T> p, const CwtSsqDev<T>& q, cpx<T>* Wx, short* K, cpx<T>* dWx_dbg,
                    hipStream_t st) {
  p.Wx = Wx;
  p.K = K;
  p.dWx = dWx_dbg;
  p.n_kinds = 2;
  p.rpadded = 0;
  p.cols = pl->N;
  p.out_scale = (const T*)pl->d_scale_l1;                 // ssq_cwt is always L1 (ssq_cwt.rs:405)
  int s0 = 0;
  while (s0 < pl->na) {
    const int lq = pl->zoom_logq[(size_t)s0];
    int s1 = s0 + 1;
    while (s1 < pl->na && pl->zoom_logq[(size_t)s1] == lq) ++s1;
    if (lq > 0) {
      CwtDev<T> z = p;
      z.scale0 = s0;
      z.n_transforms = (s1 - s0) * 2;
      z.log_p2 = lq;
      z.log_p1 = pl->logP - lq;
      z.tw_m = (const cpx<T>*)pl->d_twz + pl->twz_off[lq];
      z.tw_compact = sizeof(T) == 8 ? 1 : 0;
      z.tw_f2 = (const cpx<T>*)pl->d_f2z[lq];
      SSQ_HIP(launch_cwt_tile_ssq<T>(CWT_INV_Z, z, q, st));
    } else {
      for (int c0 = s0; c0 < s1; c0 += pl->chunk) {
        const int ns = (s1 - c0 < pl->chunk) ? s1 - c0 : pl->chunk;
        p.scale0 = c0;
        p.n_transforms = ns * 2;
        p.tw_m = (const cpx<T>*)pl->d_tw1;
        SSQ_HIP(launch_cwt_tile<T>(CWT_INV_A, p, st));
        p.tw_m = (const cpx<T>*)pl->d_tw2;
        SSQ_HIP(launch_cwt_tile_ssq<T>(CWT_INV_B, p, q, st));
      }
    }
    s0 = s1;
  }
  return 0;
}

// the analytic signal xa = ifft_P(X 1[k <= P/2]) of the padded signal into the workspace: the register-core transform of
// the pseudo-scale "one" (register-core plans; p.xh must hold the forward transform)
template <typename T>
int analytic_signal(const ssq_cwt_plan* pl, const CwtDev<T>& p, char* ws, hipStream_t st) {
  if constexpr (sizeof(T) == 4) {
    const WsLayout L = ws_layout(pl);
    CwtDev<T> pa = p;
    pa.scale0 = 0;
    pa.n_kinds = 1;
    pa.n_transforms = 1;
    pa.rpadded = 1;
    pa.cols = pl->P;
    CwtRegDev r = reg_dev<T>(pl, pa);
    r.psiT = (const float*)pl->d_xa_psiT;
    r.psiT_off = pl->d_xa_off;
    r.psiT_A = pl->d_xa_int;
    r.band = pl->d_xa_int + 1;
    r.psih = (const float*)pl->d_xa_psiT;
    r.psi_off = pl->d_xa_off + 1;
    r.out_scale = pl->d_xa_scale;
    r.Wx = (cpx<float>*)(ws + L.xa);
    r.dWx = nullptr;
    SSQ_HIP(launch_cwt_reg_inv(r, pl->n_cus, st));
  }
  return 0;
}

template <typename T>
int exec_cwt_typed(ssq_cwt_plan* pl, const void* d_x, long long batch, bool l1, bool rpadded, void* d_Wx,
                   void* d_dWx, char* ws, hipStream_t st) {
  const long long cols = rpadded ? pl->P : pl->N;
  // The short-wavelet scales by time tiles that store Wx / dWx (contiguous columns: the analytic-input and the plain
  // groups; the decimated and full-circle groups would write 8-byte pieces); SSQ_CWT_OS_STORE=0 keeps the transforms
  const char* es = std::getenv("SSQ_CWT_OS_STORE");
  const bool tiles = sizeof(T) == 4 && !rpadded && d_dWx && pl->os_s1 > pl->os_s0 && !(es && std::atoi(es) == 0);
  std::vector<char> mask;
  if (tiles) {
    mask.assign((size_t)pl->na, 0);
    for (int i = pl->os_a0; i < pl->os_s1; ++i) mask[(size_t)i] = 1;
  }
  const WsLayout L = ws_layout(pl);
  for (long long b = 0; b < batch; ++b) {
    CwtDev<T> p = base_dev<T>(pl, ws);
    if (int rc = run_forward<T>(pl, p, (const T*)d_x + b * pl->N, st)) return rc;
    cpx<T>* W = (cpx<T>*)d_Wx + b * pl->na * cols;
    cpx<T>* dW = d_dWx ? (cpx<T>*)d_dWx + b * pl->na * cols : nullptr;
    if (int rc = run_inverse<T>(pl, p, W, dW, l1, rpadded, st, NoAfter(), 0, tiles ? mask.data() : nullptr)) return rc;
    if constexpr (sizeof(T) == 4) {
      if (tiles) {
        CwtOsDev o;
        std::memset(&o, 0, sizeof(o));
        o.x = (const float*)d_x + b * pl->N;
        o.xs = (cpx<float>*)(ws + L.os_xs);
        o.tw1024 = (const cpx<float>*)pl->d_tw1024;
        o.q.N = pl->N;
        o.q.na = pl->na;
        o.dbg_Wx = (cpx<float>*)W;
        o.dbg_dWx = (cpx<float>*)dW;
        o.store_only = 1;
        o.out_mul = l1 ? nullptr : (const float*)pl->d_scale_l2;     // sqrt(a) / P: inv_F below carries the P
        o.n_signal = pl->N;
        o.padtype = pl->padtype;
        const double mulP = l1 ? 1.0 : (double)pl->P;
        if (pl->os_s0 > pl->os_a0) {
          if (int rc = analytic_signal<T>(pl, p, ws, st)) return rc;
          o.xa = (const cpx<float>*)(ws + L.xa);
          o.xa_off = pl->n1;
          o.H = (const float*)pl->d_osHa;
          o.s_begin = pl->os_a0;
          o.s_end = pl->os_s0;
          o.xi_step = (float)((2.0 * M_PI / 4096.0) / pl->dt);
          o.inv_F = (float)(mulP / 4096.0);
          SSQ_HIP(launch_cwt_os_analytic(o, st));
        }
        for (int rows = 4; rows <= 8; rows += 4) {
          const double F = 1024.0 * rows;
          o.H = (const float*)(rows == 4 ? pl->d_osH4 : pl->d_osH);
          o.s_begin = rows == 4 ? pl->os_s0 : pl->os_mid;
          o.s_end = rows == 4 ? pl->os_mid : pl->os_s1;
          o.xi_step = (float)((2.0 * M_PI / F) / pl->dt);
          o.inv_F = (float)(mulP / F);
          SSQ_HIP(launch_cwt_os(o, rows, 0, st));
        }
      }
    }
  }
  return 0;
}

template <typename T>
int exec_ssq_typed(ssq_cwt_plan* pl, const void* d_x, long long batch, int freq_dist, int maprange,
                   int squeezing, int flipud, double gamma, void* d_Tx, void* d_dbg_Wx, void* d_dbg_dWx,
                   void* d_dbg_wk, char* ws, hipStream_t st) {
  const WsLayout L = ws_layout(pl);
  std::vector<double> f((size_t)pl->na);
  const bool ups = pl->variant != 0;
  if (ups) {
    if ((int)pl->ups_freqs.size() != pl->na || pl->ups_nv < 1) SSQ_FAIL("upstream ssq exec: frequencies / nv not set");
    f = pl->ups_freqs;
  } else if (int rc = ssq_cwt_ssq_freqs(pl->scales.data(), pl->na, pl->N, pl->dt, maprange, freq_dist, f.data())) {
    return rc;
  }
  const int n = pl->na;
  CwtSsqDev<T> q;
  std::memset(&q, 0, sizeof(q));
  q.N = pl->N;
  q.na = n;
  q.is_log = (n > 1 && (f[1] / f[0] > 1.1)) ? 1 : 0;                      // ssq_cwt.rs:135-139
  if (ups) {
    // old/ssqueezepy/algos.py:356-363 (log: vlmin = log2 v[0], dvl = log2 v[1] - log2 v[0]) and :87-90 (linear)
    if (n < 2) SSQ_FAIL("upstream ssq_cwt needs at least 2 scales");
    q.is_log = freq_dist == SSQ_FREQS_LOG ? 1 : 0;
    const double vmin = q.is_log ? std::log2(f[0]) : f[0];
    const double dv = q.is_log ? std::log2(f[1]) - std::log2(f[0]) : f[1] - f[0];
    q.bin_min = (T)vmin;
    q.bin_step = (T)dv;
    q.inv_bin_step = (T)(1.0 / dv);
    q.variant = 1;
    q.tx_const = (T)(std::log(2.0) / (double)pl->ups_nv);                // ssqueezing.py:122-124
  } else if (q.is_log) {                                                  // :142-149
    const double lmin = std::log2(f[0]);
    const double lstep = n > 1 ? (std::log2(f[n - 1]) - lmin) / (double)(n - 1) : 1.0;
    q.bin_min = (T)lmin;
    q.bin_step = (T)lstep;
    q.inv_bin_step = (T)(1.0 / lstep);
  } else {                                                                // :150-157
    const double lin_min = f[0];
    const double lstep = n > 1 ? (f[n - 1] - lin_min) / (double)(n - 1) : 1.0;
    q.bin_min = (T)lin_min;
    q.bin_step = (T)lstep;
    q.inv_bin_step = (T)(1.0 / lstep);
  }
  q.squeezing = squeezing;
  q.flipud = flipud;
  q.gamma = (T)(gamma < 0 ? 10.0 * 2.2204460492503131e-16 : gamma);       // ssq_cwt.rs:438-441
  if (ups && gamma < 0) q.gamma = (T)(10.0 * (sizeof(T) == 8 ? 2.2204460492503131e-16 : 1.1920928955078125e-07));
  q.leb_val = (T)(1.0 / (double)n);
  const long long plane = (long long)n * pl->N;
  if (pl->fused_ssq() && !pl->all_tiled && !ups) {
    if (!pl->side) {
      SSQ_HIP(hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking));
      SSQ_HIP(hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming));
      SSQ_HIP(hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming));
    }
    for (long long b = 0; b < batch; ++b) {
      CwtDev<T> p = base_dev<T>(pl, ws);
      cpx<T>* W = (cpx<T>*)(ws + L.w);
      short* K = (short*)(ws + L.dw);
      q.Wx = W;
      q.dWx = nullptr;
      q.Tx = (cpx<T>*)d_Tx + b * plane;
      q.wk = d_dbg_wk ? (cpx<T>*)d_dbg_wk + b * plane : nullptr;
      // fork: clear this signal's Tx (2.15 GB at C4) beside the transforms instead of in front of the reassignment
      SSQ_HIP(hipEventRecord(pl->ev_fork, st));
      SSQ_HIP(hipStreamWaitEvent(pl->side, pl->ev_fork, 0));
      SSQ_HIP(hipMemsetAsync(q.Tx, 0, (size_t)plane * sizeof(cpx<T>), pl->side));
      SSQ_HIP(hipEventRecord(pl->ev_join, pl->side));
      if (int rc = run_forward<T>(pl, p, (const T*)d_x + b * pl->N, st)) return rc;
      if (int rc = run_inverse_ssq<T>(pl, p, q, W, K, d_dbg_dWx ? (cpx<T>*)d_dbg_dWx + b * plane : nullptr, st)) return rc;
      SSQ_HIP(hipStreamWaitEvent(st, pl->ev_join, 0));            // join
      SSQ_HIP(launch_cwt_reassign_k<T>(q, K, st));
      if (d_dbg_Wx)
        SSQ_HIP(hipMemcpyAsync((cpx<T>*)d_dbg_Wx + b * plane, W, (size_t)plane * sizeof(cpx<T>),
                               hipMemcpyDeviceToDevice, st));
    }
    return 0;
  }
  const int group = (pl->can_fuse_ssq() && !pl->all_tiled && !ups) ? ssq_group_env(0) : 0;
  // reassignment with a written-rows bitmap (first run of a row: plain store).  SSQ_CWT_SWEEP: 0 = read-modify-write
  // of a cleared Tx; 1 (default) = bitmap, Tx cleared beside the transforms; 2 = bitmap and the kernel writes the
  // untouched rows as zeros itself, no clear (measured slower on C4: the clear overlaps the transforms, the zero rows
  // would not -- profiles/r02_ab_cwt_sweep.txt)
  const char* sweep_env = tune_env("SSQ_CWT_SWEEP");
  const int sweep_mode = sweep_env ? std::atoi(sweep_env) : 1;
  const bool os = sizeof(T) == 4 && group == 0 && (pl->os_s1 > pl->os_s0 || pl->os_z1 > pl->os_z0);   // time-tile family
  const bool sweep = !ups && !os && group == 0 && cwt_reassign_can_sweep<T>(n) && sweep_mode != 0;
  const bool self_zero = sweep && (sweep_mode == 2 || !pl->can_fuse_ssq());
  const bool side_clear = pl->can_fuse_ssq() && !self_zero;   // clear Tx beside the transforms
  if (side_clear && !pl->side) {
    SSQ_HIP(hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking));
    SSQ_HIP(hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming));
    SSQ_HIP(hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming));
  }
  for (long long b = 0; b < batch; ++b) {
    CwtDev<T> p = base_dev<T>(pl, ws);
    q.Tx = (cpx<T>*)d_Tx + b * plane;
    if (side_clear) {
      SSQ_HIP(hipEventRecord(pl->ev_fork, st));
      SSQ_HIP(hipStreamWaitEvent(pl->side, pl->ev_fork, 0));
      SSQ_HIP(hipMemsetAsync(q.Tx, 0, (size_t)plane * sizeof(cpx<T>), pl->side));
      SSQ_HIP(hipEventRecord(pl->ev_join, pl->side));
    }
    if (int rc = run_forward<T>(pl, p, (const T*)d_x + b * pl->N, st)) return rc;
    cpx<T>* W = (cpx<T>*)(ws + L.w);
    cpx<T>* dW = (cpx<T>*)(ws + L.dw);
    q.Wx = W;
    q.dWx = dW;
    q.wk = d_dbg_wk ? (cpx<T>*)d_dbg_wk + b * plane : nullptr;
    if (group > 0) {
      // reassign each group of scales right behind its transforms, while its Wx / dWx are still in the Infinity Cache
      bool joined = false;
      auto after = [&](int s0, int s1) -> int {
        if (!joined) SSQ_HIP(hipStreamWaitEvent(st, pl->ev_join, 0));
        joined = true;
        q.s_begin = s0;
        q.s_end = s1;
        SSQ_HIP(launch_cwt_reassign<T>(q, st, false));
        return 0;
      };
      if (int rc = run_inverse<T>(pl, p, W, dW, true, false, st, after, group)) return rc;   // always L1 (:405)
    } else if (os) {
      if constexpr (sizeof(T) == 4) {
        // every other scale through the transforms into the workspaces ...
        if (!pl->all_tiled) {
          if (int rc = run_inverse<T>(pl, p, W, dW, true, false, st, NoAfter(), 0, pl->os_mask.data())) return rc;
          if (d_dbg_Wx)
            SSQ_HIP(hipMemcpyAsync((cpx<T>*)d_dbg_Wx + b * plane, W, (size_t)plane * sizeof(cpx<T>),
                                   hipMemcpyDeviceToDevice, st));
          if (d_dbg_dWx)
            SSQ_HIP(hipMemcpyAsync((cpx<T>*)d_dbg_dWx + b * plane, dW, (size_t)plane * sizeof(cpx<T>),
                                   hipMemcpyDeviceToDevice, st));
        }
        SSQ_HIP(hipStreamWaitEvent(st, pl->ev_join, 0));           // Tx is clear from here on
        // the scales in FRONT of the tiled ones first, while Tx is still empty: with the written-rows bitmap their runs
        // are plain stores (behind the tile kernels each would be a dependent read-modify-write chain per column)
        int lead = 0;
        while (lead < n && !pl->os_mask[(size_t)lead]) ++lead;
        const bool lead_sweep = lead > 0 && lead < n && cwt_reassign_can_sweep<T>(n) && sweep_mode != 0;
        if (lead_sweep) {
          q.s_begin = 0;
          q.s_end = lead;
          SSQ_HIP(launch_cwt_reassign_sweep<T>(q, st, false));
        }
        // ... the short-wavelet scales tile by tile straight into Tx (their Wx / dWx exist only on chip) ...
        CwtOsDev o;
        std::memset(&o, 0, sizeof(o));
        o.x = (const float*)d_x + b * pl->N;
        o.xs = (cpx<float>*)(ws + L.os_xs);
        o.tw1024 = (const cpx<float>*)pl->d_tw1024;
        o.q = q;
        o.dbg_Wx = d_dbg_Wx ? (cpx<float>*)d_dbg_Wx + b * plane : nullptr;
        o.dbg_dWx = d_dbg_dWx ? (cpx<float>*)d_dbg_dWx + b * plane : nullptr;
        o.n_signal = pl->N;
        o.padtype = pl->padtype;
        if (pl->os_s0 > pl->os_a0) {
          if (int rc = analytic_signal<T>(pl, p, ws, st)) return rc;
          o.xa = (const cpx<float>*)(ws + L.xa);
          o.xa_off = pl->n1;
          o.H = (const float*)pl->d_osHa;
          o.s_begin = pl->os_a0;
          o.s_end = pl->os_s0;
          o.xi_step = (float)((2.0 * M_PI / 4096.0) / pl->dt);
          o.inv_F = (float)(1.0 / 4096.0);
          SSQ_HIP(launch_cwt_os_analytic(o, st));
        }
        for (int rows = 4; rows <= 8; rows += 4) {               // ascending scales: the short tiles first
          const double F = 1024.0 * rows;
          o.H = (const float*)(rows == 4 ? pl->d_osH4 : pl->d_osH);
          o.s_begin = rows == 4 ? pl->os_s0 : pl->os_mid;
          o.s_end = rows == 4 ? pl->os_mid : pl->os_s1;
          o.xi_step = (float)((2.0 * M_PI / F) / pl->dt);
          o.inv_F = (float)(1.0 / F);
          SSQ_HIP(launch_cwt_os(o, rows, 0, st));
        }
        if (pl->os_d1 > pl->os_s1) {
          const double S = (double)kOsF * (double)(1 << kOsLogDec);
          o.H = (const float*)pl->d_osHd;
          o.s_begin = pl->os_s1;
          o.s_end = pl->os_d1;
          o.xi_step = (float)((2.0 * M_PI / S) / pl->dt);
          o.inv_F = (float)(1.0 / S);
          SSQ_HIP(launch_cwt_os(o, 8, kOsLogDec, st));
        }
        if (pl->os_z1 > pl->os_z0) {                             // band-limited scales: full-circle phase blocks
          o.H = (const float*)pl->d_osHz;
          o.xh = p.xh;
          o.log_dec = pl->logP - 12;
          o.full_n0 = pl->P / 4 - pl->n1;                        // <= 0: the emitted window covers [n1, n1 + N)
          o.s_begin = pl->os_z0;
          o.s_end = pl->os_z1;
          o.xi_step = (float)((2.0 * M_PI / (double)pl->P) / pl->dt);
          o.inv_F = (float)(1.0 / (double)pl->P);
          SSQ_HIP(launch_cwt_os_full(o, st));
        }
        // ... and the rest added by the column-ordered reassignment (read-modify-write), run by run
        for (int s0 = lead_sweep ? lead : 0; s0 < n;) {
          if (pl->os_mask[(size_t)s0]) {
            ++s0;
            continue;
          }
          int s1 = s0;
          while (s1 < n && !pl->os_mask[(size_t)s1]) ++s1;
          q.s_begin = s0;
          q.s_end = s1;
          SSQ_HIP(launch_cwt_reassign<T>(q, st, false));
          s0 = s1;
        }
      }
      continue;
    } else {
      if (int rc = run_inverse<T>(pl, p, W, dW, true, false, st)) return rc;
      if (side_clear) SSQ_HIP(hipStreamWaitEvent(st, pl->ev_join, 0));
      q.s_begin = 0;
      q.s_end = n;
      if (sweep) SSQ_HIP(launch_cwt_reassign_sweep<T>(q, st, self_zero));
      else SSQ_HIP(launch_cwt_reassign<T>(q, st, !side_clear));
    }
    if (d_dbg_Wx)
      SSQ_HIP(hipMemcpyAsync((cpx<T>*)d_dbg_Wx + b * plane, W, (size_t)plane * sizeof(cpx<T>),
                             hipMemcpyDeviceToDevice, st));
    if (d_dbg_dWx)
      SSQ_HIP(hipMemcpyAsync((cpx<T>*)d_dbg_dWx + b * plane, dW, (size_t)plane * sizeof(cpx<T>),
                             hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

}  // namespace

extern "C" {

int ssq_cwt_plan_create(ssq_cwt_plan** plan, int dtype, int64_t n_signal, int wavelet, const double* scales,
                        int64_t na, double dt, int padtype) {
  return ssq_cwt_plan_create_v(plan, dtype, n_signal, wavelet, 0.0, 0.0, scales, na, dt, padtype, SSQ_VARIANT_RUST);
}

int ssq_cwt_plan_create_v(ssq_cwt_plan** plan, int dtype, int64_t n_signal, int wavelet, double p0, double p1,
                          const double* scales, int64_t na, double dt, int padtype, int variant) {
  if (!plan) SSQ_FAIL("plan is NULL");
  *plan = nullptr;
  if (dtype != SSQ_F32 && dtype != SSQ_F64) SSQ_FAIL("dtype must be SSQ_F32 or SSQ_F64");
  if (n_signal <= 0) SSQ_FAIL("empty input signal");
  if (na < 0 || (na > 0 && !scales)) SSQ_FAIL("bad scales");
  if (na > 32767) SSQ_FAIL("too many scales (max 32767)");
  ssq_cwt_plan* pl = new ssq_cwt_plan();
  pl->dtype = dtype;
  pl->N = n_signal;
  pl->P = host::next_power_of_2(n_signal + n_signal / 2);        // cwt.rs:87
  pl->n1 = (pl->P - pl->N) / 2;                                  // cwt.rs:98
  const bool ups = (variant & SSQ_VARIANT_UPSTREAM) != 0;
  if (ups) {
    int64_t up = 0, n1 = 0, n2 = 0;
    host::p2up(n_signal, &up, &n1, &n2);                         // old/ssqueezepy/utils/common.py:32-51
    pl->P = up;
    pl->n1 = n1;
    pl->variant = variant;
    pl->table_code = wavelet == SSQ_WAVELET_MORLET ? 3 : 2;
    pl->wp0 = p0;
    pl->wp1 = p1;
    if (wavelet == SSQ_WAVELET_MORLET && !(p0 > 0.0)) {
      delete pl;
      SSQ_FAIL("upstream morlet needs mu > 0");
    }
    if (wavelet != SSQ_WAVELET_MORLET && !(p0 > 0.0 && p1 > 0.0)) {
      delete pl;
      SSQ_FAIL("upstream gmw needs gamma > 0 and beta > 0");
    }
  }
  pl->wavelet = wavelet;
  pl->padtype = padtype;
  pl->na = (int)na;
  pl->dt = dt;
  pl->scales.assign(scales, scales + na);
  int lp = 0;
  while ((1LL << lp) < pl->P) ++lp;
  pl->logP = lp;
  const char* force_big = std::getenv("SSQ_CWT_FORCE_BIG");     // tests: the P > 2^24 path at a small size
  if (lp > 24 || (force_big && force_big[0] == '1' && lp >= 4)) {
    if (lp > 30) {
      delete pl;
      SSQ_FAIL("signal too long: padded length above 2^30");
    }
    pl->big = true;                                              // cwt.rs:87 takes any N
    pl->log_p1 = 4;                                              // (tile-kernel fields: unused on this path)
    pl->log_p2 = 0;
  } else if (lp < 4) {
    pl->naive = true;
    pl->log_p1 = lp;
    pl->log_p2 = 0;
  } else if (lp <= 12) {
    pl->log_p1 = lp;
    pl->log_p2 = 0;
  } else {
    pl->two_step = true;
    pl->log_p2 = lp / 2;                                          // step B gets the shorter transforms (two blocks per CU)
    // fp64: a tile of step A holds C = 160 KB / (17 M bytes) columns, i.e. 16 C-byte global segments -- the SHORTER first
    // step doubles them (C5: 4096 -> 2048 points, 32 -> 64 B; C4: 2048 -> 1024, 64 -> 128 B): one C5 signal 106 -> 91 ms,
    // C4 fp64 19.0 -> 17.2 ms (profiles/r03_ab_cwt_f64_split.txt)
    if (dtype == SSQ_F64) pl->log_p2 = (lp + 1) / 2;
    if (const char* e = tune_env("SSQ_CWT_P2UP")) pl->log_p2 = (lp + std::atoi(e)) / 2;   // tuning switch
    pl->log_p1 = lp - pl->log_p2;
  }
  // SSQ_CWT_REG=0 keeps the tile kernels for the two-step scales (A/B and tests)
  {
    const char* e = std::getenv("SSQ_CWT_REG");
    pl->reg = !ups && dtype == SSQ_F32 && pl->two_step && (lp == 20 || lp == 21) && !(e && std::atoi(e) == 0);
    pl->reg_D = pl->reg ? (int)(pl->P >> 20) : 0;
  }
  const long long csz = dtype == SSQ_F32 ? 8 : 16;
  long long chunk_mb = 128;                                      // ybuf of a chunk stays inside the 256 MB Infinity Cache
  if (const char* e = tune_env("SSQ_CWT_CHUNK_MB")) chunk_mb = std::atoll(e) > 0 ? std::atoll(e) : chunk_mb;   // tuning switch
  long long ch = (chunk_mb << 20) / (2 * pl->P * csz);
  if (ch < 1) ch = 1;
  if (ch > (na > 0 ? na : 1)) ch = (na > 0 ? na : 1);
  pl->chunk = (int)ch;
  pl->zoom_logq.assign((size_t)na, 0);
  pl->band.assign((size_t)na, (int)(pl->P / 2 + 1));
  // SSQ_CWT_NOPRUNE=1 (tests): every scale through the plain two-step transform, no band-limit shortcuts
  const char* noprune = std::getenv("SSQ_CWT_NOPRUNE");
  const char* os_env = std::getenv("SSQ_CWT_OS");               // tests: 0 switches the time-tile family off
  if ((pl->two_step || pl->big) && !(noprune && noprune[0] == '1')) {
    const double h = 2.0 * M_PI / (double)pl->P;                     // base.rs:20
    const double wmax = ups ? upstream_support(wavelet, p0, p1, dtype) : wavelet_support(wavelet, dtype);
    for (int64_t i = 0; i < na; ++i) {
      const double a = scales[i];
      if (!(a > 0.0) || !std::isfinite(a)) continue;
      const double kb = wmax / (a * h) + 2.0;                        // psih_i[k] == 0 for k >= kb
      if (kb < (double)(pl->P / 2 + 1)) pl->band[(size_t)i] = (int)kb + 1;
      if (kb > (double)zoom_max_q() || pl->big) continue;
      int lq = 4;
      while ((double)(1LL << lq) < kb) ++lq;
      pl->zoom_logq[(size_t)i] = lq;
    }
  }
  // time-tiled ssq path: the longest run of ascending scales whose wavelet fits the tile halo in time (6 sigma_t <= halo,
  // sigma_t = a for the Morlet wavelet, 4.943 a for the GMW: 1 / the spectral width at the peak) and is negligible at the
  // Nyquist frequency (so that the spectrum's cut leaves no slow tail); SSQ_CWT_OS=0 switches it off
  {
    const char* e = os_env;
    if (!ups && dtype == SSQ_F32 && pl->two_step && !(e && std::atoi(e) == 0)) {
      const double sig = wavelet == SSQ_WAVELET_MORLET ? 1.0 : 4.943;
      const double a_hi = (double)kOsHalo / (6.0 * sig);
      const double a_lo = wavelet == SSQ_WAVELET_MORLET ? 4.0 : 1.3;
      bool ascending = true;
      for (int64_t i = 1; i < na; ++i) ascending = ascending && scales[i] >= scales[i - 1];
      int best0 = 0, best1 = 0, run0 = -1;
      for (int64_t i = 0; ascending && i <= na; ++i) {
        const bool ok = i < na && scales[i] >= a_lo && scales[i] <= a_hi && pl->zoom_logq[(size_t)i] == 0;
        if (ok && run0 < 0) run0 = (int)i;
        if (!ok && run0 >= 0) {
          if ((int)i - run0 > best1 - best0) {
            best0 = run0;
            best1 = (int)i;
          }
          run0 = -1;
        }
      }
      // (a block walks the scales of ITS tile one after the other: worth it only with enough tiles to fill the chip)
      if (best1 - best0 >= 8 && n_signal >= 64LL * kOsL) {
        pl->os_s0 = best0;
        pl->os_s1 = best1;
        // the scales whose wavelet fits HALF the halo take the 4096-point tiles (SSQ_CWT_OS_ROWS=8: all on 8192 points)
        const char* er = tune_env("SSQ_CWT_OS_ROWS");
        int mid = best0;
        if (!(er && std::atoi(er) == 8))
          while (mid < best1 && scales[mid] <= 0.5 * a_hi) ++mid;
        if (mid - best0 < 8) mid = best0;
        pl->os_mid = mid;
        pl->os_d1 = best1;
        // behind them the long wavelets that are band-limited below 1 / 64 cycles per sample: 16-fold decimated tiles
        // (halo 16 * 2048 samples; psih_s[k] == 0 from band[s] <= P / 32 on); SSQ_CWT_OS_DEC=0 keeps the transforms
        const char* ed = tune_env("SSQ_CWT_OS_DEC");
        if (!(ed && std::atoi(ed) == 0)) {
          int d1 = best1;
          while (d1 < (int)na && scales[d1] <= a_hi * (double)(1 << kOsLogDec) && pl->zoom_logq[(size_t)d1] == 0 &&
                 (long long)pl->band[(size_t)d1] <= pl->P / 32)
            ++d1;
          if (d1 - best1 >= 4) pl->os_d1 = d1;
        }
        // in front of them the finest scales, whose psih is NOT negligible at Nyquist, on tiles of the analytic signal
        // (register-core plans compute it with one extra transform); the continued spectrum must vanish by 2 pi:
        // Morlet a >= 1.97, GMW a >= 0.64.  SSQ_CWT_OS_ANALYTIC=0 keeps the register-core transforms for them
        const char* ea = tune_env("SSQ_CWT_OS_ANALYTIC");
        pl->os_a0 = best0;
        if (pl->reg && !(ea && std::atoi(ea) == 0)) {
          const double a_ext = wavelet == SSQ_WAVELET_MORLET ? 1.97 : 0.64;
          int a0 = best0;
          while (a0 > 0 && scales[a0 - 1] >= a_ext && pl->zoom_logq[(size_t)(a0 - 1)] == 0) --a0;
          pl->os_a0 = a0;
        }
      }
    }
  }
  // full-circle phase blocks for the band-limited scales: 2 N <= P (the kept samples lie inside the middle half of the
  // padded length, which is what the tile kernel emits), spectrum below 2048 bins; SSQ_CWT_OS_FULL=0 keeps mode Z + the column reassignment for them
  {
    const char* e = os_env;
    const char* ef = tune_env("SSQ_CWT_OS_FULL");
    if (!ups && dtype == SSQ_F32 && pl->two_step && !(e && std::atoi(e) == 0) && !(ef && std::atoi(ef) == 0) &&
        2 * pl->N <= pl->P && n_signal >= 64LL * kOsL) {
      bool ascending = true;
      for (int64_t i = 1; i < na; ++i) ascending = ascending && scales[i] >= scales[i - 1];
      int z0 = pl->os_d1;
      while (ascending && z0 < (int)na && !(pl->zoom_logq[(size_t)z0] > 0 && pl->band[(size_t)z0] <= 2048)) ++z0;
      int z1 = z0;
      while (ascending && z1 < (int)na && pl->zoom_logq[(size_t)z1] > 0 && pl->band[(size_t)z1] <= 2048) ++z1;
      if (z1 - z0 >= 8) {
        pl->os_z0 = z0;
        pl->os_z1 = z1;
      }
    }
    pl->os_mask.assign((size_t)(na > 0 ? na : 1), 0);
    for (int i = (pl->os_s1 > pl->os_s0 ? pl->os_a0 : 0); i < pl->os_d1; ++i) pl->os_mask[(size_t)i] = 1;
    for (int i = pl->os_z0; i < pl->os_z1; ++i) pl->os_mask[(size_t)i] = 1;
    pl->all_tiled = na > 0;
    for (int64_t i = 0; i < na; ++i) pl->all_tiled = pl->all_tiled && pl->os_mask[(size_t)i];
  }
  int rc = dtype == SSQ_F32 ? build_tables<float>(pl) : build_tables<double>(pl);
  if (rc) {
    ssq_cwt_plan_destroy(pl);
    return rc;
  }
  *plan = pl;
  return 0;
}

int ssq_cwt_plan_destroy(ssq_cwt_plan* pl) {
  if (!pl) return 0;
  hipFree(pl->d_psih);
  hipFree(pl->d_psi_off);
  hipFree(pl->d_tw1);
  hipFree(pl->d_tw2);
  hipFree(pl->d_twz);
  hipFree(pl->d_f2a);
  for (void* q : pl->d_f2z) hipFree(q);
  hipFree(pl->d_band);
  hipFree(pl->d_twhi);
  hipFree(pl->d_twlo);
  hipFree(pl->d_scale_l1);
  hipFree(pl->d_scale_l2);
  hipFree(pl->d_scales);
  hipFree(pl->d_psiT);
  hipFree(pl->d_psiT_off);
  hipFree(pl->d_psiT_A);
  hipFree(pl->d_tw1024);
  hipFree(pl->d_tw20);
  hipFree(pl->d_osH);
  hipFree(pl->d_osH4);
  hipFree(pl->d_osHd);
  hipFree(pl->d_osHz);
  hipFree(pl->d_osHa);
  hipFree(pl->d_xa_psiT);
  hipFree(pl->d_xa_off);
  hipFree(pl->d_xa_int);
  hipFree(pl->d_xa_scale);
  if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
  if (pl->ev_join) (void)hipEventDestroy(pl->ev_join);
  if (pl->side) (void)hipStreamDestroy(pl->side);
  delete pl;
  return 0;
}

int64_t ssq_cwt_plan_workspace_bytes(const ssq_cwt_plan* pl, int64_t batch) {
  (void)batch;   // signals of a batch are processed back to back through one workspace
  if (!pl) return 0;
  return ws_layout(pl).total;
}

int ssq_cwt_plan_exec_cwt(ssq_cwt_plan* pl, const void* d_x, int64_t batch, int l1_norm, int rpadded,
                          void* d_Wx, void* d_dWx, void* d_workspace, int64_t workspace_bytes, void* stream) {
  if (!pl) SSQ_FAIL("plan is NULL");
  if (batch <= 0 || pl->na == 0) return 0;
  if (!d_x || !d_Wx) SSQ_FAIL("device pointer is NULL");
  if (!d_workspace || workspace_bytes < ws_layout(pl).total) SSQ_FAIL("workspace too small");
  if (pl->dtype == SSQ_F32)
    return exec_cwt_typed<float>(pl, d_x, batch, l1_norm != 0, rpadded != 0, d_Wx, d_dWx, (char*)d_workspace,
                                 (hipStream_t)stream);
  return exec_cwt_typed<double>(pl, d_x, batch, l1_norm != 0, rpadded != 0, d_Wx, d_dWx, (char*)d_workspace,
                                (hipStream_t)stream);
}

int ssq_cwt_plan_exec_ssq(ssq_cwt_plan* pl, const void* d_x, int64_t batch, int freq_dist, int maprange,
                          int squeezing, int flipud, double gamma, void* d_Tx, void* d_dbg_Wx, void* d_dbg_dWx,
                          void* d_dbg_wk, void* d_workspace, int64_t workspace_bytes, void* stream) {
  if (!pl) SSQ_FAIL("plan is NULL");
  if (pl->na == 0) SSQ_FAIL("index out of bounds: scales is empty (ssq_cwt.rs:459)");
  if (batch <= 0) return 0;
  if (!d_x || !d_Tx) SSQ_FAIL("device pointer is NULL");
  if (!d_workspace || workspace_bytes < ws_layout(pl).total) SSQ_FAIL("workspace too small");
  if (pl->dtype == SSQ_F32)
    return exec_ssq_typed<float>(pl, d_x, batch, freq_dist, maprange, squeezing, flipud, gamma, d_Tx, d_dbg_Wx,
                                 d_dbg_dWx, d_dbg_wk, (char*)d_workspace, (hipStream_t)stream);
  return exec_ssq_typed<double>(pl, d_x, batch, freq_dist, maprange, squeezing, flipud, gamma, d_Tx, d_dbg_Wx,
                                d_dbg_dWx, d_dbg_wk, (char*)d_workspace, (hipStream_t)stream);
}

}  // extern "C"

// ---- host-pointer entry points: cached plan (the wavelet table alone is ~1 GB at C4), cached device buffers,
// ---- D2H of signal b on a second stream while signal b+1 computes -----------------------------------------------
namespace {

// the four TEST hooks plan creation reads from the environment are part of the key (tests flip them between calls);
// tuning switches exist only in -DSSQ_TUNING builds (ssq_common.h::tune_env) and are not
std::string plan_env() {
  std::string k;
  for (const char* v : {"SSQ_CWT_NOPRUNE", "SSQ_CWT_FORCE_BIG", "SSQ_CWT_REG", "SSQ_CWT_OS"}) {
    const char* e = std::getenv(v);
    k += e ? e : "";
    k += '|';
  }
  return k;
}
struct CwtKey {
  int dtype, wavelet, padtype;
  int64_t n_signal;
  double dt;
  std::vector<double> scales;
  int variant = 0;
  double p0 = 0.0, p1 = 0.0;
  std::string env = plan_env();
  bool operator==(const CwtKey& o) const {
    return dtype == o.dtype && wavelet == o.wavelet && padtype == o.padtype && n_signal == o.n_signal && dt == o.dt &&
           scales == o.scales && variant == o.variant && p0 == o.p0 && p1 == o.p1 && env == o.env;
  }
};
struct CachedCwt {
  CwtKey key;
  ssq_cwt_plan* pl;
  int dev;
};
std::vector<CachedCwt> g_cwt_plans;            // guarded by hostpath::mutex()
constexpr size_t kMaxCwtPlans = 2;

int cached_cwt_plan(const CwtKey& key, ssq_cwt_plan** out) {
  int dev = 0;
  SSQ_HIP(hipGetDevice(&dev));
  for (size_t i = 0; i < g_cwt_plans.size(); ++i) {
    if (g_cwt_plans[i].dev == dev && g_cwt_plans[i].key == key) {
      CachedCwt c = g_cwt_plans[i];
      g_cwt_plans.erase(g_cwt_plans.begin() + (long)i);
      g_cwt_plans.insert(g_cwt_plans.begin(), c);
      *out = c.pl;
      return 0;
    }
  }
  while (g_cwt_plans.size() >= kMaxCwtPlans) {     // make room first: the tables are large
    ssq_cwt_plan_destroy(g_cwt_plans.back().pl);
    g_cwt_plans.pop_back();
  }
  ssq_cwt_plan* pl = nullptr;
  if (int rc = ssq_cwt_plan_create_v(&pl, key.dtype, key.n_signal, key.wavelet, key.p0, key.p1, key.scales.data(),
                                     (int64_t)key.scales.size(), key.dt, key.padtype, key.variant))
    return rc;
  g_cwt_plans.insert(g_cwt_plans.begin(), CachedCwt{key, pl, dev});
  *out = pl;
  return 0;
}

struct EventPair {
  hipEvent_t ev[2] = {nullptr, nullptr};
  ~EventPair() {
    for (auto e : ev)
      if (e) (void)hipEventDestroy(e);
  }
  int init() {
    for (auto& e : ev) SSQ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return 0;
  }
};

}  // namespace

namespace ssq {
namespace hostpath {
void clear_cwt_plans() {
  for (auto& c : g_cwt_plans) ssq_cwt_plan_destroy(c.pl);
  g_cwt_plans.clear();
}
}  // namespace hostpath
}  // namespace ssq


static int cwt_host_impl(int dtype, const void* x, int64_t batch, int64_t n_signal, int wavelet, const double* scales,
                         int64_t na, double dt, int l1_norm, int padtype, int rpadded, void* Wx, void* dWx, int variant,
                         double p0, double p1) {
  if (!x || !Wx) SSQ_FAIL("x or Wx is NULL");
  if (batch <= 0) SSQ_FAIL("batch must be positive");
  if (na == 0) return 0;
  if (!scales) SSQ_FAIL("scales is NULL");
  std::lock_guard<std::mutex> lk(hostpath::mutex());
  ssq_cwt_plan* pl = nullptr;
  if (int rc = cached_cwt_plan(CwtKey{dtype, wavelet, padtype, n_signal, dt, std::vector<double>(scales, scales + na), variant,
                                      p0, p1}, &pl))
    return rc;
  const long long esz = dtype == SSQ_F32 ? 4 : 8;
  const long long cols = rpadded ? pl->P : n_signal;
  const long long out1 = na * cols * 2 * esz, in1 = n_signal * esz;
  void *dx = nullptr, *dW = nullptr, *ddW = nullptr, *ws = nullptr;
  if (int rc = hostpath::scratch(hostpath::SLOT_X, batch * in1, &dx)) return rc;
  if (int rc = hostpath::scratch(hostpath::SLOT_OUT, 2 * out1, &dW)) return rc;
  if (dWx)
    if (int rc = hostpath::scratch(hostpath::SLOT_A, 2 * out1, &ddW)) return rc;
  const long long wsb = ssq_cwt_plan_workspace_bytes(pl, 1);
  if (int rc = hostpath::scratch(hostpath::SLOT_WS0, wsb, &ws)) return rc;
  hipStream_t s0 = hostpath::stream(0), s1 = hostpath::stream(1);
  EventPair done, freed;
  if (int rc = done.init()) return rc;
  if (int rc = freed.init()) return rc;
  SSQ_HIP(hipMemcpyAsync(dx, x, (size_t)(batch * in1), hipMemcpyHostToDevice, s0));
  int rc = 0;
  // one signal: any failure is recorded and the loop left -- both streams are ALWAYS synchronised before returning, the
  // asynchronous copies write into the caller's (pinned) arrays
  auto one = [&](int64_t b) -> int {
    const int k = (int)(b & 1);
    char* oW = (char*)dW + (long long)k * out1;
    char* odW = dWx ? (char*)ddW + (long long)k * out1 : nullptr;
    if (b >= 2) SSQ_HIP(hipStreamWaitEvent(s0, freed.ev[k], 0));          // slot k was downloaded
    if (int r = ssq_cwt_plan_exec_cwt(pl, (char*)dx + b * in1, 1, l1_norm, rpadded, oW, odW, ws, wsb, s0)) return r;
    SSQ_HIP(hipEventRecord(done.ev[k], s0));
    SSQ_HIP(hipStreamWaitEvent(s1, done.ev[k], 0));
    SSQ_HIP(hipMemcpyAsync((char*)Wx + b * out1, oW, (size_t)out1, hipMemcpyDeviceToHost, s1));
    if (dWx) SSQ_HIP(hipMemcpyAsync((char*)dWx + b * out1, odW, (size_t)out1, hipMemcpyDeviceToHost, s1));
    SSQ_HIP(hipEventRecord(freed.ev[k], s1));
    return 0;
  };
  for (int64_t b = 0; b < batch && rc == 0; ++b) rc = one(b);
  const hipError_t e0 = hipStreamSynchronize(s0), e1 = hipStreamSynchronize(s1);
  if (rc) return rc;
  SSQ_HIP(e0);
  SSQ_HIP(e1);
  return 0;
}

extern "C" {

int ssq_cwt_host(int dtype, const void* x, int64_t batch, int64_t n_signal, int wavelet, const double* scales,
                 int64_t na, double dt, int l1_norm, int padtype, int rpadded, void* Wx, void* dWx) {
  return cwt_host_impl(dtype, x, batch, n_signal, wavelet, scales, na, dt, l1_norm, padtype, rpadded, Wx, dWx,
                       SSQ_VARIANT_RUST, 0.0, 0.0);
}

int ssq_cwt_host_v(int dtype, const void* x, int64_t batch, int64_t n_signal, int wavelet, double p0, double p1,
                   const double* scales, int64_t na, double dt, int l1_norm, int padtype, int rpadded, int variant,
                   void* Wx, void* dWx) {
  return cwt_host_impl(dtype, x, batch, n_signal, wavelet, scales, na, dt, l1_norm, padtype, rpadded, Wx, dWx, variant, p0,
                       p1);
}

}  // extern "C"

static int ssq_cwt_host_impl(int dtype, const void* x, int64_t batch, int64_t n_signal, int wavelet,
                             const double* scales, int64_t na, double dt, int freq_dist, int maprange, int padtype,
                             int squeezing, int flipud, double gamma, void* Tx, double* ssq_freqs, void* dbg_Wx,
                             void* dbg_dWx, void* dbg_wk, int variant, double p0, double p1, const double* ups_freqs,
                             int ups_nv) {
  if (!x || !Tx) SSQ_FAIL("x or Tx is NULL");
  if (batch <= 0) SSQ_FAIL("batch must be positive");
  if (na <= 0) SSQ_FAIL("index out of bounds: scales is empty (ssq_cwt.rs:459)");
  if (!scales) SSQ_FAIL("scales is NULL");
  std::lock_guard<std::mutex> lk(hostpath::mutex());
  ssq_cwt_plan* pl = nullptr;
  if (int rc = cached_cwt_plan(CwtKey{dtype, wavelet, padtype, n_signal, dt, std::vector<double>(scales, scales + na), variant,
                                      p0, p1}, &pl))
    return rc;
  if (variant & SSQ_VARIANT_UPSTREAM) {
    if (!ups_freqs) SSQ_FAIL("ssq_freqs_asc is NULL");
    pl->ups_freqs.assign(ups_freqs, ups_freqs + na);
    pl->ups_nv = ups_nv;
  }
  const long long esz = dtype == SSQ_F32 ? 4 : 8;
  const long long out1 = na * n_signal * 2 * esz, in1 = n_signal * esz;
  void *dx = nullptr, *dT = nullptr, *d1 = nullptr, *d2 = nullptr, *d3 = nullptr, *ws = nullptr;
  if (int rc = hostpath::scratch(hostpath::SLOT_X, batch * in1, &dx)) return rc;
  if (int rc = hostpath::scratch(hostpath::SLOT_OUT, 2 * out1, &dT)) return rc;
  if (dbg_Wx)
    if (int rc = hostpath::scratch(hostpath::SLOT_A, 2 * out1, &d1)) return rc;
  if (dbg_dWx)
    if (int rc = hostpath::scratch(hostpath::SLOT_B, 2 * out1, &d2)) return rc;
  if (dbg_wk)
    if (int rc = hostpath::scratch(hostpath::SLOT_C, 2 * out1, &d3)) return rc;
  const long long wsb = ssq_cwt_plan_workspace_bytes(pl, 1);
  if (int rc = hostpath::scratch(hostpath::SLOT_WS0, wsb, &ws)) return rc;
  hipStream_t s0 = hostpath::stream(0), s1 = hostpath::stream(1);
  EventPair done, freed;
  if (int rc = done.init()) return rc;
  if (int rc = freed.init()) return rc;
  SSQ_HIP(hipMemcpyAsync(dx, x, (size_t)(batch * in1), hipMemcpyHostToDevice, s0));
  int rc = 0;
  auto one = [&](int64_t b) -> int {             // (errors leave the loop; the streams are synchronised below either way)
    const int k = (int)(b & 1);
    const long long so = (long long)k * out1;
    if (b >= 2) SSQ_HIP(hipStreamWaitEvent(s0, freed.ev[k], 0));
    if (int r = ssq_cwt_plan_exec_ssq(pl, (char*)dx + b * in1, 1, freq_dist, maprange, squeezing, flipud, gamma,
                                      (char*)dT + so, dbg_Wx ? (char*)d1 + so : nullptr,
                                      dbg_dWx ? (char*)d2 + so : nullptr, dbg_wk ? (char*)d3 + so : nullptr, ws, wsb, s0))
      return r;
    SSQ_HIP(hipEventRecord(done.ev[k], s0));
    SSQ_HIP(hipStreamWaitEvent(s1, done.ev[k], 0));
    SSQ_HIP(hipMemcpyAsync((char*)Tx + b * out1, (char*)dT + so, (size_t)out1, hipMemcpyDeviceToHost, s1));
    if (dbg_Wx) SSQ_HIP(hipMemcpyAsync((char*)dbg_Wx + b * out1, (char*)d1 + so, (size_t)out1, hipMemcpyDeviceToHost, s1));
    if (dbg_dWx) SSQ_HIP(hipMemcpyAsync((char*)dbg_dWx + b * out1, (char*)d2 + so, (size_t)out1, hipMemcpyDeviceToHost, s1));
    if (dbg_wk) SSQ_HIP(hipMemcpyAsync((char*)dbg_wk + b * out1, (char*)d3 + so, (size_t)out1, hipMemcpyDeviceToHost, s1));
    SSQ_HIP(hipEventRecord(freed.ev[k], s1));
    return 0;
  };
  for (int64_t b = 0; b < batch && rc == 0; ++b) rc = one(b);
  const hipError_t e0 = hipStreamSynchronize(s0), e1 = hipStreamSynchronize(s1);
  if (rc) return rc;
  SSQ_HIP(e0);
  SSQ_HIP(e1);
  if (ssq_freqs && !(variant & SSQ_VARIANT_UPSTREAM))
    if (int rc2 = ssq_cwt_ssq_freqs(scales, na, n_signal, dt, maprange, freq_dist, ssq_freqs)) return rc2;
  return 0;
}

extern "C" {

int ssq_ssq_cwt_host(int dtype, const void* x, int64_t batch, int64_t n_signal, int wavelet,
                     const double* scales, int64_t na, double dt, int freq_dist, int maprange, int padtype,
                     int squeezing, int flipud, double gamma, void* Tx, double* ssq_freqs, void* dbg_Wx,
                     void* dbg_dWx, void* dbg_wk) {
  return ssq_cwt_host_impl(dtype, x, batch, n_signal, wavelet, scales, na, dt, freq_dist, maprange, padtype, squeezing,
                           flipud, gamma, Tx, ssq_freqs, dbg_Wx, dbg_dWx, dbg_wk, SSQ_VARIANT_RUST, 0.0, 0.0, nullptr, 0);
}

int ssq_ssq_cwt_host_v(int dtype, const void* x, int64_t batch, int64_t n_signal, int wavelet, double p0, double p1,
                       const double* scales, int64_t na, double dt, int nv, const double* ssq_freqs_asc, int freq_dist,
                       int padtype, int squeezing, double gamma, int variant, void* Tx, void* Wx, void* dWx, void* wk) {
  return ssq_cwt_host_impl(dtype, x, batch, n_signal, wavelet, scales, na, dt, freq_dist, SSQ_MAPRANGE_PEAK, padtype,
                           squeezing, (variant & SSQ_VARIANT_FLIPUD) ? 1 : 0, gamma, Tx, nullptr, Wx, dWx, wk, variant, p0,
                           p1, ssq_freqs_asc, nv);
}

}  // extern "C"
