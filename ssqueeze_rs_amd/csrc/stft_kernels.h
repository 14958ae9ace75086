// stft_kernels.h -- device parameter block + launch entry points of the STFT family.
#pragma once
#include "ssq_common.h"

namespace ssq {

// Device-side parameters of one STFT-family launch (passed by value).
template <typename T>
struct StftDev {
  const T* x;              // [batch][n_signal]
  cpx<T>* out;             // [batch][n_freqs][n_frames]
  const cpx<T>* tw;        // W_N^i = exp(-2*pi*i*i/N), i in [0, N)
  const cpx<T>* win2;      // fused kernel: (g[n]/2, g'[n]*fs*alpha/2), alpha = power of two balancing the channels
  const T* ssq_freqs;      // [n_freqs] reference expression (ssq_stft.rs:50), exact fix-up table
  long long n_signal;      // length of one (virtual) signal
  // where signal `sig` of the batch starts: x + (sig / group) * group_stride + (sig % group) * sig_stride.
  // Plain batches: group = 1, group_stride = n_signal.  The chunked front end (ssq_stft_plan_exec_strided) makes the
  // overlapping extended chunks of one channel a group: sig_stride = chunk, group_stride = channel pitch.
  long long sig_stride, group_stride;
  int group;
  long long total_tiles;
  int n_frames;
  int n_freqs;
  int hop;
  int pad_left;            // (n_fft-1)/2   stft_utils.rs:22
  int padtype;
  int tiles_per_signal;    // tiles of one signal covered by THIS launch: [ta0, ta0+ta_n) then [tb0, ...)
  int ta0, ta_n, tb0;      // (interior tiles and edge tiles go to separate launches)
  int out_kind;            // SSQ_OUT_*
  int squeezing;
  T sfs_step;              // Sfs[i] = i*sfs_step          (ssq_stft.rs:255)
  T dw;                    // ssq_freqs[1]-ssq_freqs[0]    (ssq_stft.rs:273)
  T inv_dw;
  T gamma2;                // gamma^2 (|Sx| < gamma  <=>  |Sx|^2 < gamma^2)
  T leb_val;               // (1/n_freqs)*dw               (ssq_stft.rs:294,298)
  T f_last;                // ssq_freqs[n_freqs-1]
  T inv_alpha;             // dSx = unpacked imaginary channel * inv_alpha (exact: power of two)
  T two_pi_eff;            // 2*pi*alpha when phase_bin is fed alpha*dSx (fused), 2*pi otherwise
  unsigned long long* stamps;  // diagnostic builds only (-DSSQ_STAMPS): per-wave cycle totals per phase
  float keep_big, keep_bias;   // fp32 TX kernel: keep-mask of |Sx|^2 >= gamma^2 as clamp(den*keep_big + keep_bias)
  int ablate;              // timing experiments only (env SSQ_ABLATE): bit mask of stages to skip; 0 in production
  T leb_unit;              // 1/n_freqs weight of "lebesgue" (ssq_stft.rs:294), without the dw factor
  // Bluestein mode of the fused kernel (n_fft not a power of two: rustfft plans any length, stft.rs:43-44): the frame
  // of n_eff samples is transformed through two FFTs of the kernel's own length m = 2^LOGN >= 2*n_eff - 1.
  // win2 then holds (window * chirp) zero-padded to m, blue_b the spectrum of the chirp filter (times 1/m) and
  // blue_post the output chirp exp(-i*pi*k^2/n_eff), k < n_eff.
  // Upstream-parity mode (/root/reference/old/ssqueezepy; unfused kernels only): variant & 1 switches
  // the bin rule to algos.py:957-968 (keep |Sx| > gamma; k = min(rint(max((w - f0)/dw, 0)), n-1), half-to-even, with
  // Sfs read from the table = np.linspace), variant & 4 flips the rows (k = n-1-k); `rot` = n_fft/2 rotates the frame
  // (modulated STFT: stft_utils.py:70-83) -- sample j of a frame is transform input (j - rot) mod n_fft.
  int variant;
  int rot;
  T gamma;                 // the threshold itself (the upstream rule compares |Sx| > gamma)
  int n_eff;               // n_fft actually transformed (== 2^LOGN outside Bluestein mode)
  const cpx<T>* blue_b;
  const cpx<T>* blue_post;
  // Mixed-radix mode of the fused kernel (n_fft = 2^a 3^b 5^c 7^d 11^e 13^f, not a power of two): mr_np > 0 passes
  // inside the frame's exchange row (fft_mixed.h), radix of pass i = 1 + ((mr_radix >> 4i) & 15); tw is then the
  // W_{n_eff} table, win2 the plain window tables zero-padded to 2^LOGN.
  int mr_np;
  unsigned mr_radix;
};

template <typename T>
__device__ __forceinline__ const T* sig_base(const StftDev<T>& p, long long sig) {
  if (p.group <= 1) return p.x + sig * p.group_stride;
  const unsigned g = (unsigned)sig / (unsigned)p.group;
  return p.x + (long long)g * p.group_stride + (long long)((unsigned)sig - g * (unsigned)p.group) * p.sig_stride;
}

// fused LDS-tile kernel (stft_fused.hip): 64 <= n_fft <= 4096, power of two
template <typename T>
bool fused_supported(int n_fft);
template <typename T>
int fused_tile_frames(int n_fft);          // frames per output tile (F)
// launches the interior-tile kernel (direct loads) and the edge-tile kernel (mirrored/zero padding)
template <typename T>
hipError_t launch_stft_fused(const StftDev<T>& p, int n_fft, int cu_count, long long batch, hipStream_t stream);
// the any-length modes of the same kernel (stft_anylen.hip): p.n_eff != fft_len
template <typename T>
hipError_t launch_stft_anylen(const StftDev<T>& p, int fft_len, int cu_count, long long batch, hipStream_t stream);

// generic any-n_fft kernels (stft_generic.hip); tables are always double
struct GenericTabs {
  const double* g;       // [n_fft]
  const double* gd;      // [n_fft]  g' * fs
  const double* tw_re;   // [n_fft]  cos(2*pi*i/n)
  const double* tw_im;   // [n_fft] -sin(2*pi*i/n)
};
template <typename T>
hipError_t launch_dft_frames(const StftDev<T>& p, long long batch, int n_fft, const GenericTabs& tabs,
                             cpx<T>* Sx, cpx<T>* dSx /*nullable*/, hipStream_t stream);
// any-n_fft path WITHOUT the O(n_fft) sums per bin (n_fft beyond the fused kernels: > 2048 not a power of two, > 4096):
// frames packed as z = x*g + i*x*g'*fs*alpha into [n_frames][n_fft] rows, a batched device FFT of any length
// (fft_generic.h: Stockham passes, Bluestein for non-powers of two), unpacked into Sx / dSx [n_freqs][n_frames].
// One signal per call; `Z` holds n_frames*n_fft elements, `work` fft_work_elems(n_fft, n_frames).
template <typename T>
hipError_t launch_fft_frames(const StftDev<T>& p, long long sig, int n_fft, const GenericTabs& tabs, double alpha,
                             cpx<T>* Z, cpx<T>* work, cpx<T>* Sx, cpx<T>* dSx /*nullable*/, hipStream_t stream);
// Sx,dSx -> out (Tx or WK), thread per time column, rows ascending (reference order, no atomics)
template <typename T>
hipError_t launch_reassign_cols(const StftDev<T>& p, const cpx<T>* Sx, const cpx<T>* dSx,
                                long long batch, hipStream_t stream);

// ---------------------------------------------------------------------------
// Phase transform + bin index shared by the fused and the generic kernels.
//   w  = |Sfs[i] - Im(dSx/Sx)/(2*pi)|           ssq_stft.rs:23-33
//   kk = first argmin_idx |w - ssq_freqs[idx]|   ssq_stft.rs:280-289
// Returns false when the bin is skipped (|Sx| < gamma or w infinite, :23,:278).
// `dS` is dSx times the factor folded into p.two_pi_eff (alpha in the fused kernel, 1 otherwise).
// ---------------------------------------------------------------------------
// The upstream variant's rule (unfused kernels only; kept OUT of phase_bin: the fused kernels inline that nine times per
// frame and the extra branch cost the fp64 kernel 30 % through register pressure).
template <typename T>
__device__ __forceinline__ bool phase_bin_upstream(const StftDev<T>& p, int i, cpx<T> S, cpx<T> dS, T& w_out, int& kk_out) {
  {
    // upstream (old/ssqueezepy/algos.py:957-968), evaluated in T like numba does for the array's dtype
    const T A = dS.x, B = dS.y, C = S.x, D = S.y;
    const T w = fabs(p.ssq_freqs[i] - (B * C - A * D) / ((C * C + D * D) * (T)6.283185307179586));
    const bool keep = hypot(C, D) > p.gamma;
    w_out = keep ? w : (T)INFINITY;
    const int last = p.n_freqs - 1;
    const T v = fmax((w - p.ssq_freqs[0]) / p.dw, (T)0);
    int kk = (v >= (T)last) ? last : (int)rint(v);      // NaN: comparisons false, rint(NaN) -> 0 by the cast below
    if (!(v == v)) kk = 0;
    if (p.variant & 4) kk = last - kk;
    kk_out = kk;
    return keep;
  }
}

#ifndef SSQ_F64_FASTDIV
#define SSQ_F64_FASTDIV 1     // fp64: the two quotients of the phase / bin by reciprocal + Newton / residual correction (<= 1 ulp)
#endif                        // instead of IEEE division sequences: 1 605 -> 1 512 vector instructions per frame, batch 64 1.568 -> 1.510 ms
template <typename T>
__device__ __forceinline__ bool phase_bin(const StftDev<T>& p, int i, cpx<T> S, cpx<T> dS, T& w_out, int& kk_out) {
  const T den = S.x * S.x + S.y * S.y;
  const T num = dS.y * S.x - dS.x * S.y;
  const T two_pi = p.two_pi_eff;                    // 6.283185307179586 (ssq_stft.rs:32) [* alpha]
  T pd;
  if constexpr (sizeof(T) == 4) {
    pd = num * __builtin_amdgcn_rcpf(den * two_pi);
  } else {
#if SSQ_F64_FASTDIV
    {
      // reciprocal by two Newton steps on v_rcp_f64 and one residual correction: <= 1 ulp, 7 instructions instead of the
      // 11 of the IEEE division sequence (the quotient feeds a bin decision with its own tie window, not an output)
      const T d = den * two_pi;
      T r = __builtin_amdgcn_rcp(d);
      r = __builtin_fma(__builtin_fma(-d, r, (T)1), r, r);
      r = __builtin_fma(__builtin_fma(-d, r, (T)1), r, r);
      const T q0 = num * r;
      pd = __builtin_fma(__builtin_fma(-d, q0, num), r, q0);
    }
#else
    pd = num / (den * two_pi);
#endif
  }
  const T sfs = (T)i * p.sfs_step;
  T w = fabs(sfs - pd);
  const bool skip = (den < p.gamma2) || isinf(w);
  if (den < p.gamma2) w = (T)INFINITY;
  w_out = w;
  const int last = p.n_freqs - 1;
  int kk;
  if constexpr (sizeof(T) == 4) {
    // fp32 mode semantics (documented in DESIGN.md): u = fma(w, 1/dw, -0.5), kk = ceil(u)
    // clamped to [0, n_freqs-1]; exact half-bin ties go to the lower bin like the scan's strict `<`.
    const T u = __builtin_fmaf(w, p.inv_dw, (T)-0.5);
    kk = (u >= (T)last) ? last : (int)__builtin_ceilf(u);
    if (w != w) kk = 0;          // NaN never wins the scan: k stays 0
  } else {
#if SSQ_F64_FASTDIV
    const T tq0 = w * p.inv_dw;
    const T tq = __builtin_fma(__builtin_fma(-p.dw, tq0, w), p.inv_dw, tq0);
#else
    const T tq = w / p.dw;          // (w * inv_dw would do -- the tie window below re-decides exactly -- but measured 19 %
                                    //  SLOWER in the fp64 fused kernel: profiles/r03_ab_f64_fixed.txt)
#endif
    const T u = tq - (T)0.5;
    kk = (u >= (T)last) ? last : (int)ceil(u);
    if (w != w) kk = 0;
    // ONE rarely taken branch around both exact re-decisions (the fused kernels inline this nine times per frame)
    const T fr = fabs(tq - floor(tq) - (T)0.5);
    if (!skip && w == w && (w > p.f_last || fr < (T)1e-7)) {
      if (w > p.f_last) {
        // fl(w - f_k) may tie for several k: the scan keeps the FIRST minimum.
        const T target = fabs(w - p.ssq_freqs[last]);
        int lo = 0, hi = last;            // smallest k with |w - f_k| == target (dist non-increasing in k)
        if (last > 0 && fabs(w - p.ssq_freqs[last - 1]) != target) {
          lo = last;
        } else {
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (fabs(w - p.ssq_freqs[mid]) == target) hi = mid; else lo = mid + 1;
          }
        }
        kk = lo;
      } else {
        // near a half-bin tie: decide with the reference's own distance expression
        int k0 = kk - 2 < 0 ? 0 : kk - 2;
        int k1 = kk + 2 > last ? last : kk + 2;
        T best = (T)INFINITY;
        int bk = 0;
        for (int c = k0; c <= k1; ++c) {
          const T d = fabs(w - p.ssq_freqs[c]);
          if (d < best) { best = d; bk = c; }
        }
        kk = bk;
      }
    }
  }
  kk_out = kk;
  return !skip;
}

// padded sample fetch: stft_utils.rs:19-65 by index mirroring (no padded copy)
template <typename T>
__device__ __forceinline__ T load_padded(const T* __restrict__ xs, long long m, long long n, int padtype) {
  if (m >= 0 && m < n) return xs[m];
  if (padtype != 0) return (T)0;
  long long mm = (m < 0) ? -m : (2 * n - 2 - m);
  if (mm >= 0 && mm < n) return xs[mm];
  return (T)0;
}

// The same fetch without branches (index selects, one unconditional load): a loader that takes it for many samples in
// a row keeps all its loads in flight (the branchy form waits for each load at the join).  `live` = false returns 0.
template <typename T>
__device__ __forceinline__ T load_padded_flat(const T* __restrict__ xs, long long m, long long n, int padtype, bool live) {
  const bool in = m >= 0 && m < n;
  const long long mm = (m < 0) ? -m : (2 * n - 2 - m);
  const bool ok = live && (in || (padtype == 0 && mm >= 0 && mm < n));
  const long long idx = in ? m : mm;
  const T v = xs[ok ? idx : 0];
  return ok ? v : (T)0;
}

}  // namespace ssq
