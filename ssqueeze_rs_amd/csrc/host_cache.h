// host_cache.h -- what makes the drop-in call (`_rs.*` on NumPy arrays -> ssq_*_host) cheap to repeat: a pool of
// pinned host blocks (results land in them by DMA and are handed to NumPy without a copy), device scratch that
// survives between calls, and the two streams the host-pointer entry points pipeline H2D / kernels / D2H on.
// The reference allocates fresh arrays per call too (ssq_stft.rs:307-312 into_pyarray); its cost there is a malloc.
#pragma once
#include <mutex>
#include "ssq_common.h"

namespace ssq {
namespace hostpath {

std::mutex& mutex();                       // the host entry points are serialised (one pipeline, one scratch set)

// device scratch slots, grown on demand and kept (per device); freed by ssq_host_cache_clear
enum Slot { SLOT_X = 0, SLOT_OUT, SLOT_WS0, SLOT_WS1, SLOT_A, SLOT_B, SLOT_C, SLOT_COUNT };
int scratch(Slot s, long long bytes, void** p);
hipStream_t stream(int i);                 // i in {0, 1}; created on first use
void drop_device_state();
void clear_stft_plans();                   // api_stft.hip
void clear_cwt_plans();                    // api_cwt.hip

}  // namespace hostpath
}  // namespace ssq
