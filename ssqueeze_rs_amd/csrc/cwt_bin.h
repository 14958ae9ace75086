// cwt_bin.h -- phase transform + bin of one (scale, time) element of ssq_cwt, shared by the reassignment kernels
// (cwt_kernels.hip) and the time-tiled kernel (cwt_os.hip).
#pragma once
#include <cmath>
#include "cwt_kernels.h"

namespace ssq {

#ifndef SSQ_CWT_FAST_BIN
#define SSQ_CWT_FAST_BIN 1      // fp32: hardware rcp / sqrt / log2 (1 ulp) in the phase transform and the bin formula
#endif
// ------------------------------------------------ phase transform + reassignment ----
// ssq_cwt.rs:15-47 (phase_cwt) and :116-222 (ssqueeze).
// Phase transform and bin of one (scale, time) element: returns the Tx row (after flipud) or -1.
// The upstream variant's rule (old/ssqueezepy/algos.py:899-940), plain arithmetic in T: the normalised wavelets cannot
// overflow fp32.  Called by the column-ordered reassignment only (cwt_reassign_kernel), never by the tile kernels.
template <typename T>
__device__ __forceinline__ int reassign_bin_upstream(const CwtSsqDev<T>& p, cpx<T> Wv, cpx<T> dW, T& w) {
    const T A = dW.x, B = dW.y, C = Wv.x, D = Wv.y;
    if (!(hypot(C, D) > p.gamma)) {
      w = (T)INFINITY;
      return -1;
    }
    w = fabs((B * C - A * D) / ((C * C + D * D) * (T)6.283185307179586));
    const T v = fmax(((p.is_log ? log2(w) : w) - p.bin_min) / p.bin_step, (T)0);
    int bin = (v >= (T)(p.na - 1)) ? p.na - 1 : (int)rint(v);
    if (!(v == v)) bin = 0;
    return p.flipud ? (p.na - 1 - bin) : bin;
  }

template <typename T>
__device__ __forceinline__ int reassign_bin(const CwtSsqDev<T>& p, cpx<T> Wv, cpx<T> dW, T& w) {
  const T two_pi = (T)(2.0 * 3.14159265358979323846);
  bool small;
  if constexpr (sizeof(T) == 4) {
    // pre-scaled ratio: the reference's GMW is un-normalised (peak ~4e17), so |Wx|^2 and
    // b*c - a*d overflow fp32 long before the ratio does.  The scale factor and the quotient use the hardware
    // reciprocal (1 ulp): IEEE divisions here made the kernel's arithmetic as long as its memory time
    const T mx = fmaxf(fabsf(Wv.x), fabsf(Wv.y));
#if SSQ_CWT_FAST_BIN
    const T sc = __builtin_amdgcn_rcpf(mx);
#else
    const T sc = (T)1 / mx;
#endif
    const T cs = Wv.x * sc, ds = Wv.y * sc;
    const T den = Wv.x * cs + Wv.y * ds;               // |Wx|^2 / mx
#if SSQ_CWT_FAST_BIN
    small = !(mx * __builtin_amdgcn_sqrtf(cs * cs + ds * ds) >= p.gamma);
    w = fabsf((dW.y * cs - dW.x * ds) * __builtin_amdgcn_rcpf(den * two_pi));
#else
    small = !(mx * sqrtf(cs * cs + ds * ds) >= p.gamma);
    w = fabsf((dW.y * cs - dW.x * ds) / (den * two_pi));
#endif
  } else {
    const T den = Wv.x * Wv.x + Wv.y * Wv.y;
    small = hypot(Wv.x, Wv.y) < p.gamma;               // Complex::norm()  ssq_cwt.rs:29
    w = fabs((dW.y * Wv.x - dW.x * Wv.y) / (den * two_pi));
  }
  if (small) w = (T)INFINITY;
  int kk = -1;
  if (!(isinf(w) || w != w)) {                         // ssq_cwt.rs:167
    T v;
#if SSQ_CWT_FAST_BIN
    if constexpr (sizeof(T) == 4) {
      const T lw = p.is_log ? __builtin_amdgcn_logf(w) : w;   // v_log_f32 = log2, 1 ulp (w is finite, positive or zero here)
      v = (lw - p.bin_min) * p.inv_bin_step;
    } else
#endif
    {
      if (p.is_log) v = (log2(w) - p.bin_min) / p.bin_step;    // ssq_cwt.rs:175-176
      else v = (w - p.bin_min) / p.bin_step;                   // ssq_cwt.rs:187
    }
    const T r = round(v);                              // half away from zero
    int bin;
    if (r != r) bin = 0;                               // NaN as isize == 0
    else if (r < (T)0 || r >= (T)p.na) bin = -1;       // out of range: dropped (:177,:188)
    else bin = (int)r;
    if (bin >= 0) kk = p.flipud ? (p.na - 1 - bin) : bin;
  }
  return kk;
}

}  // namespace ssq
