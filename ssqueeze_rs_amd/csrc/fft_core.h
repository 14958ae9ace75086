// fft_core.h -- the per-lane FFT core shared by the STFT and CWT kernels (gfx950).
//
// A length-N = 2^LOGN transform is held by L = N/16 lanes, 16 complex values per lane
// (lane t owns elements t + L*q).  Stockham autosort: every pass does in-register radix-16
// (last pass: the remainder radix) butterflies, then one exchange through an LDS row of
// N + N/16 elements.  After the last pass lane t holds X[t + L*q] in natural order.
// For L <= 64 a transform lives inside one wavefront and the exchange needs no block barrier.
#pragma once
#include <utility>
#include "ssq_common.h"

namespace ssq {

__device__ __forceinline__ int exch_phys(int idx) { return idx + (idx >> 4); }

template <bool MULTIWAVE>
__device__ __forceinline__ void frame_sync() {
  if (MULTIWAVE) {
    __syncthreads();
  } else {
    // a frame lives in one wavefront: the LDS unit executes a wave's DS ops in order, so a
    // compiler-level ordering point is all that is needed between the writes and the reads
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <bool INV, int R, int NB, typename T, int... Bs>
__device__ __forceinline__ void butterflies(cpx<T> (&v)[16], std::integer_sequence<int, Bs...>) {
  (dft_strided<INV, R, Bs, NB>(v), ...);
}

// Twiddle multiply (P > 0) + in-register butterflies of Stockham pass P (see ssq_common.h pass tables).
// TW_COMPACT: `tw_tab` is this pass's own table laid out [m][k] (k = position inside the sub-transform,
// NS entries per m), so that the lanes of one read touch consecutive elements (no LDS bank conflicts);
// otherwise it is the full W_N table indexed k*m*(N/(NS*R)).
template <typename T, int LOGN, int P, bool INV, bool TW_REGS, bool TW_COMPACT = false>
__device__ __forceinline__ void fft_compute(cpx<T> (&v)[16], const cpx<T> (&twr)[3][16],
                                            const cpx<T>* __restrict__ tw_tab, int t) {
  constexpr int N = 1 << LOGN, L = N / 16;
  constexpr int R = pass_radix(LOGN, P), NS = pass_ns(LOGN, P), NB = 16 / R;
  if constexpr (P > 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int m = 1; m < R; ++m) {
        cpx<T> w;
        if constexpr (TW_REGS) {
          w = twr[P - 1][b + m * NB];
        } else {
          const int k = (t + L * b) & (NS - 1);
          w = TW_COMPACT ? tw_tab[m * NS + k] : tw_tab[k * m * (N / (NS * R))];
        }
        if (INV) w.y = -w.y;
        v[b + m * NB] = cmul(v[b + m * NB], w);
      }
    }
  }
  butterflies<INV, R, NB>(v, std::make_integer_sequence<int, NB>{});
}

// LDS exchange after pass P: lane t writes its butterfly outputs at their Stockham positions and
// reads back elements t + L*q.  The DS unit executes a wave's operations in order, so a second
// frame may reuse the same row right behind this one without waiting.
template <typename T, int LOGN, int P, bool MULTIWAVE>
__device__ __forceinline__ void fft_exchange(cpx<T> (&v)[16], cpx<T>* exch, int t) {
  constexpr int N = 1 << LOGN, L = N / 16;
  constexpr int R = pass_radix(LOGN, P), NS = pass_ns(LOGN, P), NB = 16 / R;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = t + L * b;
    const int k = j & (NS - 1);
    const int base = (j - k) * R + k;
#pragma unroll
    for (int u = 0; u < R; ++u) exch[exch_phys(base + u * NS)] = v[b + u * NB];
  }
  frame_sync<MULTIWAVE>();
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = exch[exch_phys(t + L * q)];
  frame_sync<MULTIWAVE>();
}

// The whole transform of one frame: passes P .. NP-1.
template <typename T, int LOGN, int P, bool INV, bool TW_REGS, bool MULTIWAVE>
__device__ __forceinline__ void fft_pass(cpx<T> (&v)[16], cpx<T>* exch, const cpx<T> (&twr)[3][16],
                                         const cpx<T>* __restrict__ tw_tab, int t) {
  fft_compute<T, LOGN, P, INV, TW_REGS>(v, twr, tw_tab, t);
  if constexpr (P < num_passes(LOGN) - 1) {
    fft_exchange<T, LOGN, P, MULTIWAVE>(v, exch, t);
    fft_pass<T, LOGN, P + 1, INV, TW_REGS, MULTIWAVE>(v, exch, twr, tw_tab, t);
  }
}

// The transform with every pass's twiddles read from that pass's own table [m][k] (R * NS entries, k fastest): the lanes
// of a load touch consecutive entries instead of gathering k * m * stride from the W_N table.  `tw_c` = the tables of
// passes 1, 2, ... back to back.
template <typename T, int LOGN, int P, bool MULTIWAVE>
__device__ __forceinline__ void fft_pass_compact(cpx<T> (&v)[16], cpx<T>* exch, const cpx<T>* __restrict__ tw_c, int t) {
  const cpx<T> unused[3][16] = {};
  fft_compute<T, LOGN, P, false, false, true>(v, unused, tw_c, t);
  if constexpr (P < num_passes(LOGN) - 1) {
    fft_exchange<T, LOGN, P, MULTIWAVE>(v, exch, t);
    constexpr int ADV = (P == 0) ? 0 : pass_radix(LOGN, P) * pass_ns(LOGN, P);
    fft_pass_compact<T, LOGN, P + 1, MULTIWAVE>(v, exch, tw_c + ADV, t);
  }
}

// The same transform for a frame inside ONE wave with an exchange row of T instead of cpx<T>: the real parts go through
// the row, then the imaginary parts (fp64: twice the DS instructions of half the width -- the same bytes -- for half the
// LDS footprint, which is what lets eight fp64 frames share a CU).
template <typename T, int LOGN, int P>
__device__ __forceinline__ void fft_exchange_split(cpx<T> (&v)[16], T* exch, int t) {
  constexpr int N = 1 << LOGN, L = N / 16;
  static_assert(L <= 64, "single-wave frames");
  constexpr int R = pass_radix(LOGN, P), NS = pass_ns(LOGN, P), NB = 16 / R;
  T nx[16];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = t + L * b;
    const int k = j & (NS - 1);
    const int base = (j - k) * R + k;
#pragma unroll
    for (int u = 0; u < R; ++u) exch[exch_phys(base + u * NS)] = v[b + u * NB].x;
  }
  frame_sync<false>();
#pragma unroll
  for (int q = 0; q < 16; ++q) nx[q] = exch[exch_phys(t + L * q)];
  frame_sync<false>();
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = t + L * b;
    const int k = j & (NS - 1);
    const int base = (j - k) * R + k;
#pragma unroll
    for (int u = 0; u < R; ++u) exch[exch_phys(base + u * NS)] = v[b + u * NB].y;
  }
  frame_sync<false>();
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = {nx[q], exch[exch_phys(t + L * q)]};
  frame_sync<false>();
}

// `tw_c`: the passes' compact tables [m][k] back to back (see fft_pass_compact): coalesced reads instead of gathers
template <typename T, int LOGN, int P>
__device__ __forceinline__ void fft_pass_split(cpx<T> (&v)[16], T* exch, const cpx<T> (&twr)[3][16],
                                               const cpx<T>* __restrict__ tw_c, int t) {
  fft_compute<T, LOGN, P, false, false, true>(v, twr, tw_c, t);
  if constexpr (P < num_passes(LOGN) - 1) {
    fft_exchange_split<T, LOGN, P>(v, exch, t);
    constexpr int ADV = (P == 0) ? 0 : pass_radix(LOGN, P) * pass_ns(LOGN, P);
    fft_pass_split<T, LOGN, P + 1>(v, exch, twr, tw_c + ADV, t);
  }
}

// Two frames of one wave, staggered: while one frame's exchange is in flight the other computes.
template <typename T, int LOGN, int P, bool INV, bool TW_REGS>
__device__ __forceinline__ void fft_pass_pair(cpx<T> (&a)[16], cpx<T> (&b)[16], cpx<T>* exch,
                                              const cpx<T> (&twr)[3][16], const cpx<T>* __restrict__ tw_tab, int t) {
  fft_compute<T, LOGN, P, INV, TW_REGS>(a, twr, tw_tab, t);
  if constexpr (P < num_passes(LOGN) - 1) fft_exchange<T, LOGN, P, false>(a, exch, t);
  fft_compute<T, LOGN, P, INV, TW_REGS>(b, twr, tw_tab, t);
  if constexpr (P < num_passes(LOGN) - 1) {
    fft_exchange<T, LOGN, P, false>(b, exch, t);
    fft_pass_pair<T, LOGN, P + 1, INV, TW_REGS>(a, b, exch, twr, tw_tab, t);
  }
}

}  // namespace ssq
