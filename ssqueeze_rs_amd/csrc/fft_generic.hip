// fft_generic.hip -- see fft_generic.h.  Not a hot path: correctness and generality first, every pass is a plain
// streaming kernel (coalesced loads, strided stores in the early passes), twiddles from sincospi in fp64.
#include "fft_generic.h"

namespace ssq {

namespace {

constexpr int kThreads = 256;

inline long long next_pow2_ll(long long v) {
  long long m = 1;
  while (m < v) m <<= 1;
  return m;
}
inline bool is_pow2_ll(long long v) { return v > 0 && (v & (v - 1)) == 0; }

template <typename T>
__device__ __forceinline__ cpx<T> unit(double turns) {     // exp(2*pi*i*turns)
  double s, c;
  sincospi(2.0 * turns, &s, &c);
  return {(T)c, (T)s};
}

// One Stockham autosort pass of radix R over rows of n elements; ns = product of the radices already done.
template <typename T, int R>
__global__ void stockham_pass_kernel(const cpx<T>* __restrict__ in, cpx<T>* __restrict__ out, long long n, long long ns,
                                     int sign, long long batch) {
  const long long per = n / R;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per * batch) return;
  const long long b = idx / per, j = idx - b * per;
  const long long k = j % ns;
  const cpx<T>* src = in + b * n + j;
  cpx<T> v[R];
#pragma unroll
  for (int t = 0; t < R; ++t) v[t] = src[t * per];
  const double base = (double)sign * (double)k / (double)(ns * R);
#pragma unroll
  for (int t = 1; t < R; ++t) v[t] = cmul(v[t], unit<T>(base * (double)t));
  if constexpr (R == 2) {
    dft2<false>(v[0], v[1]);
  } else {
    if (sign < 0) dft4<false>(v[0], v[1], v[2], v[3]);
    else dft4<true>(v[0], v[1], v[2], v[3]);
  }
  cpx<T>* dst = out + b * n + (j - k) * R + k;
#pragma unroll
  for (int t = 0; t < R; ++t) dst[t * ns] = v[t];
}

// power-of-two rows, in place in `buf` with `tmp` (same size) as the other half of the ping-pong
template <typename T>
hipError_t fft_pow2(cpx<T>* buf, cpx<T>* tmp, long long n, long long batch, int sign, hipStream_t st) {
  if (n <= 1 || batch <= 0) return hipSuccess;
  cpx<T>* a = buf;
  cpx<T>* b = tmp;
  long long ns = 1;
  while (ns < n) {
    const int R = (n / ns >= 4) ? 4 : 2;
    const long long work = (n / R) * batch;
    const dim3 grid((unsigned)((work + kThreads - 1) / kThreads));
    if (R == 4) hipLaunchKernelGGL((stockham_pass_kernel<T, 4>), grid, dim3(kThreads), 0, st, a, b, n, ns, sign, batch);
    else hipLaunchKernelGGL((stockham_pass_kernel<T, 2>), grid, dim3(kThreads), 0, st, a, b, n, ns, sign, batch);
    ns *= R;
    cpx<T>* t = a;
    a = b;
    b = t;
  }
  if (a != buf) {
    const hipError_t e = hipMemcpyAsync(buf, a, sizeof(cpx<T>) * (size_t)(n * batch), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

// chirp[j] = exp(sign * i*pi*j^2/n), j^2 reduced mod 2n in integers
template <typename T>
__device__ __forceinline__ cpx<T> chirp(long long j, long long n, int sign) {
  const unsigned long long j2 = ((unsigned long long)j * (unsigned long long)j) % (unsigned long long)(2 * n);
  return unit<T>(0.5 * (double)sign * (double)j2 / (double)n);
}

template <typename T>
__global__ void bluestein_pre_kernel(const cpx<T>* __restrict__ data, cpx<T>* __restrict__ A, long long n, long long m,
                                     long long batch, int sign) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m * batch) return;
  const long long b = idx / m, j = idx - b * m;
  cpx<T> v = {(T)0, (T)0};
  if (j < n) v = cmul(data[b * n + j], chirp<T>(j, n, sign));
  A[idx] = v;
}

template <typename T>
__global__ void bluestein_filter_kernel(cpx<T>* __restrict__ B, long long n, long long m, int sign) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  cpx<T> v = {(T)0, (T)0};
  if (j < n) v = chirp<T>(j, n, -sign);
  else if (m - j < n) v = chirp<T>(m - j, n, -sign);
  B[j] = v;
}

template <typename T>
__global__ void bluestein_mul_kernel(cpx<T>* __restrict__ A, const cpx<T>* __restrict__ Bh, long long m, long long batch) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m * batch) return;
  A[idx] = cmul(A[idx], Bh[idx % m]);
}

template <typename T>
__global__ void bluestein_post_kernel(const cpx<T>* __restrict__ A, cpx<T>* __restrict__ data, long long n, long long m,
                                      long long batch, int sign) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * batch) return;
  const long long b = idx / n, k = idx - b * n;
  const cpx<T> v = cmul(A[b * m + k], chirp<T>(k, n, sign));
  const T s = (T)(1.0 / (double)m);
  data[idx] = {v.x * s, v.y * s};
}

}  // namespace

long long fft_work_elems(long long n, long long batch) {
  if (n <= 1) return 1;
  if (is_pow2_ll(n)) return n * batch;
  const long long m = next_pow2_ll(2 * n - 1);
  return 2 * m * batch + 2 * m;
}

template <typename T>
hipError_t fft_any_batched(cpx<T>* data, cpx<T>* work, long long n, long long batch, int sign, hipStream_t st) {
  if (n <= 1 || batch <= 0) return hipSuccess;
  if (is_pow2_ll(n)) return fft_pow2<T>(data, work, n, batch, sign, st);
  const long long m = next_pow2_ll(2 * n - 1);
  cpx<T>* A = work;
  cpx<T>* At = work + m * batch;
  cpx<T>* B = At + m * batch;
  cpx<T>* Bt = B + m;
  auto blocks = [](long long w) { return dim3((unsigned)((w + kThreads - 1) / kThreads)); };
  hipLaunchKernelGGL(bluestein_pre_kernel<T>, blocks(m * batch), dim3(kThreads), 0, st, data, A, n, m, batch, sign);
  hipLaunchKernelGGL(bluestein_filter_kernel<T>, blocks(m), dim3(kThreads), 0, st, B, n, m, sign);
  hipError_t e = fft_pow2<T>(B, Bt, m, 1, -1, st);
  if (e != hipSuccess) return e;
  e = fft_pow2<T>(A, At, m, batch, -1, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bluestein_mul_kernel<T>, blocks(m * batch), dim3(kThreads), 0, st, A, B, m, batch);
  e = fft_pow2<T>(A, At, m, batch, +1, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bluestein_post_kernel<T>, blocks(n * batch), dim3(kThreads), 0, st, A, data, n, m, batch, sign);
  return hipGetLastError();
}

template hipError_t fft_any_batched<float>(cpx<float>*, cpx<float>*, long long, long long, int, hipStream_t);
template hipError_t fft_any_batched<double>(cpx<double>*, cpx<double>*, long long, long long, int, hipStream_t);

}  // namespace ssq
