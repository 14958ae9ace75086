// fft_generic.h -- batched complex DFT of ANY length on the device, for the paths that are not hot:
// the two-integral icwt (cwt.rs:629-714: rustfft plans any x_len) and the any-n_fft STFT family beyond
// the fused kernels' range (stft.rs:43-44).  Power-of-two lengths run Stockham radix-4/2 passes through global
// memory (log4 n launches, every pass streams the batch once); other lengths run Bluestein's chirp-z on top of them.
#pragma once
#include "ssq_common.h"

namespace ssq {

// Unnormalised DFT of `batch` rows of `n` complex elements (row pitch `n`), sign = -1 forward / +1 inverse.
// `data` is transformed in place; `work` needs fft_work_elems(n, batch) elements.
long long fft_work_elems(long long n, long long batch);
template <typename T>
hipError_t fft_any_batched(cpx<T>* data, cpx<T>* work, long long n, long long batch, int sign, hipStream_t stream);

}  // namespace ssq
