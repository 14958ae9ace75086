// fft_mixed.h -- mixed-radix Stockham transform of one frame inside its LDS exchange row (gfx950).
//
// rustfft plans any length (stft.rs:43-44); lengths n = 2^a 3^b 5^c 7^d 11^e 13^f run here, inside the fused STFT kernel
// (radices 2..16: the primes, 4 / 8 / 16, and 6 / 10 / 12 / 15 as twiddle-free Good-Thomas pairs),
// instead of through Bluestein's two power-of-two transforms of >= 2n - 1 points.  The frame's L lanes walk the passes of
// the host's radix list: a pass of radix R has n/R butterflies, lane t takes butterflies t, t + L, ...; all inputs of a
// lane are read before any output is written (the transform is in place in the row), outputs land at their autosort
// positions, and after the last pass the row holds X[0..n) in natural order.  Twiddles come from the W_n table.
#pragma once
#include "fft_core.h"

namespace ssq {

constexpr int kMixMaxPasses = 8;

// cos / sin of 2*pi*m/R, m = 0 .. (R-1)/2
template <int R>
struct OddTab;
template <>
struct OddTab<3> {
  static constexpr double c[2] = {1.0, -0.5};
  static constexpr double s[2] = {0.0, 0.8660254037844386467637};
};
template <>
struct OddTab<5> {
  static constexpr double c[3] = {1.0, 0.3090169943749474241023, -0.8090169943749474241023};
  static constexpr double s[3] = {0.0, 0.9510565162951535721164, 0.5877852522924731291687};
};
template <>
struct OddTab<7> {
  static constexpr double c[4] = {1.0, 0.623489801858733530525, -0.2225209339563144042889, -0.9009688679024191262361};
  static constexpr double s[4] = {0.0, 0.7818314824680298087084, 0.9749279121818236070181, 0.4338837391175581204758};
};
template <>
struct OddTab<11> {
  static constexpr double c[6] = {1.0, 0.8412535328311811688618, 0.4154150130018864255293, -0.1423148382732851404438,
                                  -0.6548607339452850640569, -0.9594929736144973898904};
  static constexpr double s[6] = {0.0, 0.5406408174555975821076, 0.9096319953545183714117, 0.9898214418809327323761,
                                  0.755749574354258283774, 0.2817325568414296977114};
};
template <>
struct OddTab<13> {
  static constexpr double c[7] = {1.0, 0.8854560256532098959004, 0.5680647467311558025118, 0.1205366802553230533491,
                                  -0.3546048870425356259696, -0.7485107481711010986346, -0.970941817426052027157};
  static constexpr double s[7] = {0.0, 0.464723172043768545656, 0.8229838658936563945796, 0.9927088740980539928008,
                                  0.9350162426854148234398, 0.6631226582407952023768, 0.2393156642875577671488};
};

// Forward DFT of odd prime length R, in place, natural order:
//   X[k], X[R-k] = a0 + sum_j cos(2 pi jk/R) (a_j + a_{R-j})  -/+  i sum_j sin(2 pi jk/R) (a_j - a_{R-j}),  j = 1 .. (R-1)/2
template <typename T, int R>
__device__ __forceinline__ void dft_odd(cpx<T> (&a)[R]) {
  constexpr int H = (R - 1) / 2;
  cpx<T> tp[H + 1], tm[H + 1];
#pragma unroll
  for (int j = 1; j <= H; ++j) {
    tp[j] = a[j] + a[R - j];
    tm[j] = a[j] - a[R - j];
  }
  const cpx<T> a0 = a[0];
  cpx<T> x0 = a0;
#pragma unroll
  for (int j = 1; j <= H; ++j) x0 = x0 + tp[j];
  a[0] = x0;
#pragma unroll
  for (int k = 1; k <= H; ++k) {
    cpx<T> m = a0;
    cpx<T> n = {(T)0, (T)0};
#pragma unroll
    for (int j = 1; j <= H; ++j) {
      const int jk = (j * k) % R;
      const int i = jk <= H ? jk : R - jk;
      const T c = (T)OddTab<R>::c[i];
      const T s = (T)(jk <= H ? OddTab<R>::s[i] : -OddTab<R>::s[i]);
      m.x += c * tp[j].x;
      m.y += c * tp[j].y;
      n.x += s * tm[j].x;
      n.y += s * tm[j].y;
    }
    // -i*n = (n.y, -n.x)
    a[k] = {m.x + n.y, m.y - n.x};
    a[R - k] = {m.x - n.y, m.y + n.x};
  }
}

template <typename T, int R>
__device__ __forceinline__ void dft_any(cpx<T> (&a)[R]);

constexpr int mod_inverse(int a, int m) {            // a^-1 mod m, gcd(a, m) = 1
  for (int x = 1; x < m; ++x)
    if ((a * x) % m == 1) return x;
  return 1;
}

// Good-Thomas: R = N1 * N2 with gcd(N1, N2) = 1 needs no twiddles between its two stages.
//   n = (N2 n1 + N1 n2) mod R,   k = (k1 N2 (N2^-1 mod N1) + k2 N1 (N1^-1 mod N2)) mod R
template <typename T, int N1, int N2>
__device__ __forceinline__ void dft_pfa(cpx<T> (&a)[N1 * N2]) {
  constexpr int R = N1 * N2;
  constexpr int E1 = N2 * mod_inverse(N2 % N1, N1), E2 = N1 * mod_inverse(N1 % N2, N2);
  cpx<T> b[N2][N1];
#pragma unroll
  for (int n2 = 0; n2 < N2; ++n2) {
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) b[n2][n1] = a[(N2 * n1 + N1 * n2) % R];
    dft_any<T, N1>(b[n2]);
  }
#pragma unroll
  for (int k1 = 0; k1 < N1; ++k1) {
    cpx<T> c[N2];
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) c[n2] = b[n2][k1];
    dft_any<T, N2>(c);
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) a[(k1 * E1 + k2 * E2) % R] = c[k2];
  }
}

template <typename T, int R>
__device__ __forceinline__ void dft_any(cpx<T> (&a)[R]) {
  if constexpr (R == 2) dft2<false>(a[0], a[1]);
  else if constexpr (R == 4) dft4<false>(a[0], a[1], a[2], a[3]);
  else if constexpr (R == 8) dft8<false>(a);
  else if constexpr (R == 16) dft16<false>(a);
  else if constexpr (R == 6) dft_pfa<T, 2, 3>(a);
  else if constexpr (R == 10) dft_pfa<T, 2, 5>(a);
  else if constexpr (R == 12) dft_pfa<T, 4, 3>(a);
  else if constexpr (R == 15) dft_pfa<T, 3, 5>(a);
  else dft_odd<T, R>(a);
}

// Ordering point between a pass's reads and its writes (and the next pass's reads).  A frame inside one wave needs no
// wait: the LDS unit runs a wave's DS operations in issue order, so only the compiler must keep the order (aliasing
// accesses to the same row: it does; the barrier pins the schedule).  A multi-wave frame needs the block barrier.
template <bool MULTIWAVE>
__device__ __forceinline__ void row_order() {
  if constexpr (MULTIWAVE) __syncthreads();
  else __builtin_amdgcn_wave_barrier();
}

// One pass: radix R, Ns = product of the radices of the earlier passes, rem = n / (Ns * R).
template <typename T, int L, int R, bool MULTIWAVE>
__device__ __forceinline__ void mixed_pass(cpx<T>* row, int n, int Ns, int rem, const cpx<T>* __restrict__ tw, int t) {
  constexpr int ROUNDS = (16 + R - 1) / R;           // the row's 16*L slots hold at most 16*L/R butterflies
  const int nb = Ns * rem;                           // n / R
  const float inv_ns = 1.0f / (float)Ns;
  cpx<T> a[ROUNDS][R];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int j = t + L * r;
    if (j < nb) {
#pragma unroll
      for (int u = 0; u < R; ++u) a[r][u] = row[exch_phys(j + u * nb)];
    }
  }
  row_order<MULTIWAVE>();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int j = t + L * r;
    if (j < nb) {
      // q = j / Ns by float (exact here: q*Ns <= 4096), k = j mod Ns
      const int q = (Ns == 1) ? j : (int)(((float)j + 0.5f) * inv_ns);
      const int k = j - q * Ns;
      if (Ns > 1) {
        const int kr = k * rem;
#pragma unroll
        for (int u = 1; u < R; ++u) a[r][u] = cmul(a[r][u], tw[u * kr]);
      }
      dft_any<T, R>(a[r]);
      const int base = q * Ns * R + k;
#pragma unroll
      for (int u = 0; u < R; ++u) row[exch_phys(base + u * Ns)] = a[r][u];
    }
  }
  row_order<MULTIWAVE>();
}

// All passes of the host's plan (np radices, 4 bits each: R - 1; every lane of the block sees the same list, so the barriers inside a
// multi-wave frame's passes are uniform).
template <typename T, int L, bool MULTIWAVE>
__device__ __forceinline__ void fft_mixed_row(cpx<T>* row, int n, int np, unsigned radix_packed,
                                              const cpx<T>* __restrict__ tw, int t) {
  int Ns = 1;
#pragma unroll 1
  for (int ps = 0; ps < np; ++ps) {
    const int R = 1 + (int)((radix_packed >> (4 * ps)) & 15u);
    const int rem = n / (Ns * R);
    switch (R) {
      case 2: mixed_pass<T, L, 2, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 3: mixed_pass<T, L, 3, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 4: mixed_pass<T, L, 4, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 5: mixed_pass<T, L, 5, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 7: mixed_pass<T, L, 7, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 8: mixed_pass<T, L, 8, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 6: mixed_pass<T, L, 6, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 10: mixed_pass<T, L, 10, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 12: mixed_pass<T, L, 12, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 15: mixed_pass<T, L, 15, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 11: mixed_pass<T, L, 11, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      case 13: mixed_pass<T, L, 13, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
      default: mixed_pass<T, L, 16, MULTIWAVE>(row, n, Ns, rem, tw, t); break;
    }
    Ns *= R;
  }
}

}  // namespace ssq
