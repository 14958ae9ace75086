// cwt_kernels.h -- device parameter blocks + launch entry points of the CWT family.
#pragma once
#include "ssq_common.h"

namespace ssq {

// What a tile-FFT launch loads and stores (see cwt_kernels.hip).
enum CwtMode {
  CWT_FWD_A = 0,   // padded real signal -> column FFTs (+ W_P twiddle) -> ybuf
  CWT_FWD_B = 1,   // ybuf rows -> row FFTs -> xh (natural order)
  CWT_FWD_S = 2,   // P <= 4096: padded real signal -> xh in one step
  CWT_INV_A = 3,   // xh * psih (* i*xi/dt) -> column iFFTs (+ conj twiddle) -> ybuf
  CWT_INV_B = 4,   // ybuf rows -> row iFFTs -> Wx / dWx (scaled, unpadded)
  CWT_INV_S = 5,   // P <= 4096: xh * psih -> Wx / dWx in one step
  CWT_INV_Z = 6    // band-limited scale (psih == 0 for k >= Q = 2^log_p2): P = D*Q, one length-Q iFFT per residue
                   // d = n mod D of the spectrum times W_P^(-k d); no ybuf pass at all
};

template <typename T>
struct CwtDev {
  const T* x;            // [n_signal] one real signal
  cpx<T>* xh;            // [P] FFT of the padded signal             (cwt.rs:147-162)
  cpx<T>* ybuf;          // [transforms in chunk][P] step-A output
  const T* psih;         // wavelet table, scale s at psi_off[s], band[s] entries (zero beyond)   (cwt.rs:492-547)
  const long long* psi_off;
  const cpx<T>* tw_m;    // W_M^i for the transform length of this launch
  int tw_compact;        // 1: behind the M entries of tw_m lie the passes' own tables [m][k] (cwt_tw_compact_elems): lanes read
                         //    consecutive entries instead of gathering k*m*stride (fp64 tiles: no twiddles in registers)
  const cpx<T>* tw_hi;   // W_P^(i << 12)
  const cpx<T>* tw_lo;   // W_P^i, i < 4096
  const cpx<T>* tw_f2;   // in-tile factor of the W_P twiddle: step A [M][C]: W_P^(c k1); mode Z [C][M]: W_P^(c k)
  const T* out_scale;    // [na] 1/P (* sqrt(a) if !l1_norm)          (cwt.rs:251-262)
  const int* band;       // [na] psih_s[k] == 0 for k >= band[s] (<= P/2 + 1)
  cpx<T>* Wx;            // [na][cols]
  cpx<T>* dWx;           // [na][cols] or NULL
  short* K;              // fused ssq kernels: [na][N] Tx row of every (scale, time) element, -1 = skipped
  long long n_signal;
  long long P;           // padded length (power of two)
  long long n1;          // (P - N)/2                                 (cwt.rs:98)
  long long cols;        // N, or P when rpadded
  int log_p1, log_p2;    // P = P1 * P2 (two-step) ; single step: log_p1 = log P, log_p2 = 0
  int padtype;
  int rpadded;
  int scale0;            // first scale of this chunk
  int n_kinds;           // 1 (Wx) or 2 (Wx and dWx)
  int n_transforms;      // transforms in this launch (chunk scales * kinds; 1 for forward)
  int tr0;               // tile kernels: logical index of the launch's first transform (kind = (tr0 + y) % n_kinds); the
                         // step buffer is indexed by the launch-local y (one kind at a time when two would not fit the cache)
  T xi_step;             // (2*pi/P)/dt : xi_k/dt = k * xi_step      (wavelets/base.rs:18-33, cwt.rs:207)
};

// elements of the per-pass compact twiddle tables of a length-2^logm tile transform (passes 1 .. last: R * NS each)
inline long long cwt_tw_compact_elems(int logm) {
  long long n = 0;
  for (int P = 1; P < num_passes(logm); ++P) n += (long long)pass_radix(logm, P) * pass_ns(logm, P);
  return n;
}
template <typename T>
hipError_t launch_cwt_tile(int mode, const CwtDev<T>& p, hipStream_t stream);
// transforms per tile (C) of the length-2^logm tile kernel: the host sizes the tw_f2 tables with it
template <typename T>
int cwt_tile_rows(int logm);

// wavelet table psih[off[s] + k] = psi_hat(scale_s * 2*pi*k/P), k in [0, band[s])   (cwt.rs:492-547): only the span
// where the value is not exactly zero in T is stored (1.07 GB -> 0.2 GB at C4)
template <typename T>
hipError_t launch_wavelet_table(T* psih, const long long* d_off, const int* d_band, int max_band, const double* d_scales,
                                int na, long long P, int wavelet, hipStream_t stream, double p0 = 0.0, double p1 = 0.0);

// P > 2^24 (beyond the two-step tile transforms): the same pipeline through the batched any-length device FFT
// (fft_generic.h, Stockham passes through global memory) -- functional for any length the memory holds, not tuned.
template <typename T>
hipError_t launch_cwt_big_fwd(const CwtDev<T>& p, cpx<T>* work, hipStream_t stream);                // p.x -> p.xh
template <typename T>
hipError_t launch_cwt_big_inv(const CwtDev<T>& p, cpx<T>* work, hipStream_t stream);                // p.n_transforms via p.ybuf

// any-P fallback for tiny signals (P < 64): direct sums
template <typename T>
hipError_t launch_cwt_naive_fwd(const CwtDev<T>& p, hipStream_t stream);
template <typename T>
hipError_t launch_cwt_naive_inv(const CwtDev<T>& p, int na, hipStream_t stream);

template <typename T>
struct CwtSsqDev {
  const cpx<T>* Wx;      // [na][N]
  const cpx<T>* dWx;     // [na][N]
  cpx<T>* Tx;            // [na][N]
  cpx<T>* wk;            // [na][N] (w, k or -1) or NULL
  long long N;
  int na;
  int s_begin, s_end;    // scales this launch reassigns (launch_cwt_reassign; the whole call: 0, na)
  int is_log;            // ssq_cwt.rs:135-139
  int squeezing;
  int flipud;
  T bin_min;             // log2(f0) or f0                           (ssq_cwt.rs:142-158)
  T bin_step;
  T inv_bin_step;        // 1 / bin_step (evaluated in fp64 on the host)
  T gamma;
  T leb_val;             // 1/na                                      (ssq_cwt.rs:201-204)
  // upstream-parity mode (old/ssqueezepy/algos.py:899-940, ssqueezing.py:122-128): keep |Wx| > gamma,
  // k = min(rint(max(v, 0)), na-1) (clamped, half to even), every contribution times tx_const (= ln 2 / nv)
  int variant;
  T tx_const;
};
template <typename T>
hipError_t launch_cwt_reassign(const CwtSsqDev<T>& p, hipStream_t stream, bool clear);   // clear: zero Tx first
// the same with a per-lane bitmap of written rows: the first run in a row stores without reading.  zero_fill: the rows
// never touched are stored as zeros at the end, so Tx needs no clear at all; otherwise Tx must be zero on entry
template <typename T>
bool cwt_reassign_can_sweep(int na);
template <typename T>
hipError_t launch_cwt_reassign_sweep(const CwtSsqDev<T>& p, hipStream_t stream, bool zero_fill);

// Fused synchrosqueezing variants of inverse step B / mode Z (two-step plans): one block runs BOTH transforms (Wx and
// dWx, ssq_cwt.rs:387-402) of one scale for its rows, applies the phase transform (ssq_cwt.rs:15-47) and the bin
// formula (:160-196) in the store phase and writes Wx plus a 16-bit row index -- dWx never reaches memory.
// p.n_transforms = 2 * scales of the launch; q carries the binning parameters and the optional (w, k) hook buffer.
template <typename T>
hipError_t launch_cwt_tile_ssq(int mode, const CwtDev<T>& p, const CwtSsqDev<T>& q, hipStream_t stream);
// Tx from (Wx, K): one thread per time column, scales ascending, runs of equal rows summed in registers.
// Tx must be zero on entry (the caller clears it on a side stream while the transforms run).
template <typename T>
hipError_t launch_cwt_reassign_k(const CwtSsqDev<T>& p, const short* K, hipStream_t stream);

// ---- cwt_reg.hip: fp32 inverse transforms of P = 2^20 / 2^21 on the per-wave register FFT core ----
struct CwtRegDev {
  cpx<float>* xc;              // [D][1024 b][1024 a]  conj(X[k] e^{+2 pi i k d/P}), k = 1024 a + b
  const float* psiT;           // transposed wavelet table: scale s at psiT_off[s], [1024 b][A_s]
  const long long* psiT_off;
  const int* psiT_A;           // a-extent of scale s = ceil(min(band_s, 2^20) / 1024)
  const cpx<float>* tw1024;    // W_1024^j
  const cpx<float>* tw20;      // W_{2^20}^i, i < 1024
  cpx<float>* ybuf;            // [transform][d][b][n_a]
  const cpx<float>* xh;        // natural-order spectrum (prep input; the k = P/2 term)
  const float* psih;           // natural-order table (the k = P/2 term)
  const long long* psi_off;
  const int* band;
  const cpx<float>* tw_hi;     // W_P^(i << 12)
  const cpx<float>* tw_lo;     // W_P^i, i < 4096
  const float* out_scale;
  cpx<float>* Wx;
  cpx<float>* dWx;
  long long n_signal, P, n1, cols;
  int rpadded, D, scale0, n_kinds, n_transforms;
  float xi_step;
};
hipError_t launch_cwt_reg_prep(const CwtRegDev& p, hipStream_t stream);                       // xh -> xc
hipError_t launch_cwt_reg_table(float* psiT, const long long* d_offT, const int* d_A, int max_A, const float* psih,
                                const long long* d_off, const int* d_band, int na, hipStream_t stream);
hipError_t launch_cwt_reg_inv(const CwtRegDev& p, int n_cus, hipStream_t stream);             // R1 + R2 of p.n_transforms

// ---- cwt_os.hip: ssq_cwt of the short-wavelet scales by time tiles (overlap-save), fp32 ----
// geometry of `rows` = 8 | 4: transform length 1024 rows, output samples 512 rows per tile, halo 256 rows on either side
constexpr int kOsF = 8192;        // rows = 8
constexpr int kOsL = 4096;
constexpr int kOsHalo = 2048;     // input samples on either side: the wavelet's time support must fit
constexpr int kOsLogDec = 4;      // decimated tiles (long, band-limited wavelets): 16-fold, halo 16 * 2048 samples
struct CwtOsDev {
  const float* x;              // [n_signal] one real signal
  cpx<float>* xs;              // scratch [tiles][F / 2]: the tiles' spectra (k < F / 2)
  const cpx<float>* xh;        // full-circle mode: the padded signal's spectrum in natural order
  int log_dec;                 // full-circle mode: log2(P / 4096)
  long long full_n0;           // full-circle mode: P / 4 - n1 (unpadded time of the emitted window's first sample)
  const cpx<float>* xa;        // analytic-input tiles: ifft_P(X 1[k <= P/2]) on the padded grid
  long long xa_off;            //   index of unpadded time 0 in xa (= n1)
  const float* H;              // [s_end - s_begin][F / 2] psih(scale * 2 pi k / F)
  const cpx<float>* tw1024;    // W_1024^j
  CwtSsqDev<float> q;          // binning parameters, Tx (zero or partial sums on entry), optional (w, k) hook
  cpx<float>* dbg_Wx;          // optional [na][N] copies of Wx / dWx (the `_debug` hooks; the RESULT when store_only)
  cpx<float>* dbg_dWx;
  int store_only;              // `cwt`: write Wx / dWx and skip the bins / Tx
  const float* out_mul;        // optional per-scale factor on top of inv_F (cwt with the L2 norm)
  long long n_signal;
  int padtype;
  int s_begin, s_end;          // scales of this launch (ascending)
  float xi_step;               // (2 pi / F) / dt
  float inv_F;                 // 1 / F
};
hipError_t launch_cwt_os_table(float* H, const double* d_scales, int s_begin, int n_scales, int wavelet, int rows,
                               int log_dec, bool all_bins, hipStream_t stream);
// the finest scales (psih not negligible at Nyquist) on 4096-point tiles of the analytic signal p.xa
hipError_t launch_cwt_os_analytic(const CwtOsDev& p, hipStream_t stream);
hipError_t launch_cwt_os(const CwtOsDev& p, int rows, int log_dec, hipStream_t stream);
// band-limited scales (spectrum below 2048 bins) of plans with 2 N <= P: one output phase per block over the whole padded
// signal (4096-point transforms on the P / 4096-fold decimated grid), bins and run merge on chip
hipError_t launch_cwt_os_full(const CwtOsDev& p, hipStream_t stream);

}  // namespace ssq
