// ssq_common.h -- shared device/host helpers for libssq_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace ssq {

// ---------------------------------------------------------------- errors ----
void set_error(const std::string& msg);
// Tuning / diagnostic switches (measured-slower alternatives, ablation masks, stamp buffers) exist only in variant
// builds (-DSSQ_TUNING: `python -m ssqueeze_rs_amd.build --variant tune -DSSQ_TUNING`, and the --abl / --stamps builds):
// there this is getenv, in the product library it returns NULL -- the shipped dispatch reads no tuning environment.
const char* tune_env(const char* name);
#define SSQ_FAIL(msg)                 \
  do {                                \
    ::ssq::set_error(msg);            \
    return 1;                         \
  } while (0)
#define SSQ_HIP(call)                                                          \
  do {                                                                         \
    hipError_t e__ = (call);                                                   \
    if (e__ != hipSuccess) {                                                   \
      ::ssq::set_error(std::string(#call) + ": " + hipGetErrorString(e__));    \
      return 2;                                                                \
    }                                                                          \
  } while (0)

constexpr int kWave = 64;   // CDNA4 wavefront

// --------------------------------------------------------------- complex ----
template <typename T>
struct cpx {
  T x, y;
};
template <typename T>
__host__ __device__ __forceinline__ cpx<T> operator+(cpx<T> a, cpx<T> b) {
  return {a.x + b.x, a.y + b.y};
}
template <typename T>
__host__ __device__ __forceinline__ cpx<T> operator-(cpx<T> a, cpx<T> b) {
  return {a.x - b.x, a.y - b.y};
}
// (a.x + i a.y) * (b.x + i b.y)
template <typename T>
__host__ __device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T>
__host__ __device__ __forceinline__ cpx<T> cscale(cpx<T> a, T s) {
  return {a.x * s, a.y * s};
}
// multiply by -i (forward transform) or +i (inverse transform)
template <bool INV, typename T>
__host__ __device__ __forceinline__ cpx<T> mul_mi(cpx<T> a) {
  if (INV) return {-a.y, a.x};
  return {a.y, -a.x};
}

// ------------------------------------------------ in-register small DFTs ----
// All are in place with natural-order output.  INV selects exp(+2*pi*i*nk/R).
template <bool INV, typename T>
__host__ __device__ __forceinline__ void dft2(cpx<T>& a, cpx<T>& b) {
  cpx<T> t = a - b;
  a = a + b;
  b = t;
}

template <bool INV, typename T>
__host__ __device__ __forceinline__ void dft4(cpx<T>& a0, cpx<T>& a1, cpx<T>& a2, cpx<T>& a3) {
  cpx<T> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi<INV>(a1 - a3);
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = t1 + t3;
  a3 = t1 - t3;
}

// multiply by W8^k = exp(-/+ 2*pi*i*k/8), k compile-time
template <bool INV, int K, typename T>
__host__ __device__ __forceinline__ cpx<T> mul_w8(cpx<T> a) {
  constexpr int k = ((K % 8) + 8) % 8;
  const T c = (T)0.70710678118654752440;
  if (k == 0) return a;
  if (k == 4) return {-a.x, -a.y};
  if (k == 2) return mul_mi<INV>(a);
  if (k == 6) return mul_mi<!INV>(a);
  // forward: W8 = (1 - i)/sqrt2 ; W8^3 = (-1 - i)/sqrt2 ; W8^5 = (-1 + i)/sqrt2 ; W8^7 = (1 + i)/sqrt2
  // inverse: conjugates.
  if (k == 1) return INV ? cpx<T>{(a.x - a.y) * c, (a.x + a.y) * c} : cpx<T>{(a.x + a.y) * c, (a.y - a.x) * c};
  if (k == 7) return INV ? cpx<T>{(a.x + a.y) * c, (a.y - a.x) * c} : cpx<T>{(a.x - a.y) * c, (a.x + a.y) * c};
  if (k == 3) return INV ? cpx<T>{(-a.x - a.y) * c, (a.x - a.y) * c} : cpx<T>{(a.y - a.x) * c, (-a.x - a.y) * c};
  /* k == 5 */ return INV ? cpx<T>{(a.y - a.x) * c, (-a.x - a.y) * c} : cpx<T>{(-a.x - a.y) * c, (a.x - a.y) * c};
}

// multiply by W16^k = exp(-/+ 2*pi*i*k/16), k compile-time in [0,16)
template <bool INV, int K, typename T>
__host__ __device__ __forceinline__ cpx<T> mul_w16(cpx<T> a) {
  constexpr int k = ((K % 16) + 16) % 16;
  if (k % 2 == 0) return mul_w8<INV, k / 2>(a);
  // cos/sin of k*pi/8
  constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173;
  constexpr double cs[8][2] = {{1, 0}, {C1, S1}, {0, 0}, {S1, C1}, {0, 0}, {-S1, C1}, {0, 0}, {-C1, S1}};
  constexpr double cr = (k < 8) ? cs[k % 8][0] : -cs[k % 8][0];
  constexpr double sr = (k < 8) ? cs[k % 8][1] : -cs[k % 8][1];
  // W = cos - i sin (forward), cos + i sin (inverse)
  const T c = (T)cr;
  const T s = (T)(INV ? sr : -sr);
  return {a.x * c - a.y * s, a.x * s + a.y * c};
}

template <bool INV, typename T>
__host__ __device__ __forceinline__ void dft8(cpx<T> (&v)[8]) {
  dft4<INV>(v[0], v[2], v[4], v[6]);   // E[k] at v[2k]
  dft4<INV>(v[1], v[3], v[5], v[7]);   // O[k] at v[2k+1]
  cpx<T> o1 = mul_w8<INV, 1>(v[3]);
  cpx<T> o2 = mul_w8<INV, 2>(v[5]);
  cpx<T> o3 = mul_w8<INV, 3>(v[7]);
  cpx<T> o0 = v[1];
  cpx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  v[0] = e0 + o0;
  v[4] = e0 - o0;
  v[1] = e1 + o1;
  v[5] = e1 - o1;
  v[2] = e2 + o2;
  v[6] = e2 - o2;
  v[3] = e3 + o3;
  v[7] = e3 - o3;
}

template <bool INV, typename T>
__host__ __device__ __forceinline__ void dft16(cpx<T> (&v)[16]) {
  // stage 1: A_r[k] = DFT4 over s of v[r + 4s]  -> stored at v[r + 4k]
  dft4<INV>(v[0], v[4], v[8], v[12]);
  dft4<INV>(v[1], v[5], v[9], v[13]);
  dft4<INV>(v[2], v[6], v[10], v[14]);
  dft4<INV>(v[3], v[7], v[11], v[15]);
  // twiddle W16^{r k} on v[r + 4k]
  v[5] = mul_w16<INV, 1>(v[5]);
  v[6] = mul_w16<INV, 2>(v[6]);
  v[7] = mul_w16<INV, 3>(v[7]);
  v[9] = mul_w16<INV, 2>(v[9]);
  v[10] = mul_w16<INV, 4>(v[10]);
  v[11] = mul_w16<INV, 6>(v[11]);
  v[13] = mul_w16<INV, 3>(v[13]);
  v[14] = mul_w16<INV, 6>(v[14]);
  v[15] = mul_w16<INV, 9>(v[15]);
  // stage 2: for each k, DFT4 over r of v[4k + r] -> out[k + 4m] at v[4k + m]
  dft4<INV>(v[0], v[1], v[2], v[3]);
  dft4<INV>(v[4], v[5], v[6], v[7]);
  dft4<INV>(v[8], v[9], v[10], v[11]);
  dft4<INV>(v[12], v[13], v[14], v[15]);
  // transpose 4x4 to natural order: y[k + 4m] = v[4k + m]
#define SSQ_SWAP(i, j) \
  {                    \
    cpx<T> t_ = v[i];  \
    v[i] = v[j];       \
    v[j] = t_;         \
  }
  SSQ_SWAP(1, 4) SSQ_SWAP(2, 8) SSQ_SWAP(3, 12) SSQ_SWAP(6, 9) SSQ_SWAP(7, 13) SSQ_SWAP(11, 14)
#undef SSQ_SWAP
}

// R-point DFT on a compile-time strided subset of a 16-register array:
// inputs v[B + m*S], m = 0..R-1; outputs (natural order u) land at the same slots.
template <bool INV, int R, int B, int S, typename T>
__host__ __device__ __forceinline__ void dft_strided(cpx<T> (&v)[16]) {
  if constexpr (R == 2) {
    dft2<INV>(v[B], v[B + S]);
  } else if constexpr (R == 4) {
    dft4<INV>(v[B], v[B + S], v[B + 2 * S], v[B + 3 * S]);
  } else if constexpr (R == 8) {
    cpx<T> t[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) t[m] = v[B + m * S];
    dft8<INV>(t);
#pragma unroll
    for (int m = 0; m < 8; ++m) v[B + m * S] = t[m];
  } else {
    static_assert(R == 16 && B == 0 && S == 1, "unsupported radix");
    dft16<INV>(v);
  }
}

// ----------------------------------------------------------- pass tables ----
// N = 2^LOGN points handled by L = N/16 threads holding E = 16 elements each
// (thread t owns elements t + L*q).  Radix list: 16,16,..., then the remainder.
__host__ __device__ constexpr int num_passes(int logn) { return (logn + 3) / 4; }
__host__ __device__ constexpr int pass_radix(int logn, int p) {
  return (p < logn / 4) ? 16 : (1 << (logn % 4));
}
__host__ __device__ constexpr int pass_ns(int logn, int p) {   // product of earlier radices
  return 1 << (4 * p);
}

}  // namespace ssq
