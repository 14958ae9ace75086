// stft_anylen.hip -- launches of the fused STFT kernel's any-length modes (rustfft plans any length, stft.rs:43-44):
// n_fft = p.n_eff that is not a power of two runs inside the kernel of the power of two `fft_len` the plan chose,
//   MODE 2: mixed-radix passes in the frame's exchange row (fft_mixed.h), fft_len >= n_fft, prime factors <= 13;
//   MODE 1: Bluestein's chirp-z through two transforms of fft_len >= 2 n_fft - 1.
// One launch of the edge-capable loader over all tiles (the transform, not the loader, bounds these modes).
// A translation unit of its own: the two modes triple the kernel instantiations, and hipcc runs the units in parallel.
#include "stft_fused_kernel.h"

namespace ssq {

template <typename T, int LOGN>
static hipError_t launch_anylen_one(const StftDev<T>& p0, int cu_count, long long batch, hipStream_t stream) {
  using C = FusedCfg<T, LOGN, true>;
  int per_cu = (160 * 1024) / C::LDS_BYTES;
  if (per_cu < 1) per_cu = 1;
  if (per_cu * C::W > 32) per_cu = 32 / C::W;
  const int tps_all = (p0.n_frames + C::F - 1) / C::F;
  StftDev<T> p = p0;
  p.ta0 = 0;
  p.ta_n = tps_all;
  p.tb0 = 0;
  p.tiles_per_signal = tps_all;
  p.total_tiles = (long long)tps_all * batch;
  if (p.total_tiles <= 0) return hipSuccess;
  if (p.n_eff > C::N || (p.mr_np == 0 && 2 * p.n_eff - 1 > C::N)) return hipErrorInvalidValue;
  long long blocks = (long long)cu_count * per_cu;
  if (blocks > p.total_tiles) blocks = p.total_tiles;
  const dim3 g((unsigned)blocks), b(C::W * 64);
#define SSQ_LAUNCH_ANY(MODE)                                                                                 \
  do {                                                                                                       \
    if (p.out_kind == 0 && p.squeezing == 1)                                                                 \
      hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, true, false, MODE>), g, b, 0, stream, p);   \
    else if (p.out_kind == 0)                                                                                \
      hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, false, false, MODE>), g, b, 0, stream, p);  \
    else if (p.out_kind == 3)                                                                                \
      hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, false, true, MODE>), g, b, 0, stream, p);   \
    else                                                                                                     \
      hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, false, true, false, false, MODE>), g, b, 0, stream, p); \
  } while (0)
  if (p.mr_np > 0) SSQ_LAUNCH_ANY(2);
  else SSQ_LAUNCH_ANY(1);
#undef SSQ_LAUNCH_ANY
  return hipGetLastError();
}

template <typename T>
hipError_t launch_stft_anylen(const StftDev<T>& p, int fft_len, int cu_count, long long batch, hipStream_t stream) {
  switch (fft_len) {
    case 64: return launch_anylen_one<T, 6>(p, cu_count, batch, stream);
    case 128: return launch_anylen_one<T, 7>(p, cu_count, batch, stream);
    case 256: return launch_anylen_one<T, 8>(p, cu_count, batch, stream);
    case 512: return launch_anylen_one<T, 9>(p, cu_count, batch, stream);
    case 1024: return launch_anylen_one<T, 10>(p, cu_count, batch, stream);
    case 2048: return launch_anylen_one<T, 11>(p, cu_count, batch, stream);
    case 4096: return launch_anylen_one<T, 12>(p, cu_count, batch, stream);
  }
  return hipErrorInvalidValue;
}

template hipError_t launch_stft_anylen<float>(const StftDev<float>&, int, int, long long, hipStream_t);
template hipError_t launch_stft_anylen<double>(const StftDev<double>&, int, int, long long, hipStream_t);

}  // namespace ssq
