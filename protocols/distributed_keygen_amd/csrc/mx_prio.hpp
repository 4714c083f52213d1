// Issue priority of the short kernels of a step (Jacobi filter, selection, R mod N, verdict, recombination, sieve).
//
// Several steps are in flight at once (bench.py, Engine._pipelined, BiprimeRound): a step's short kernels then share
// their SIMDs with 2-4 wavefronts of OTHER steps' exponentiation kernels, and the step's own exponentiation waits for
// them.  A wavefront that raises its priority (s_setprio, user levels 0..3) is picked first by the SIMD's instruction
// arbiter, so the short kernels run at close to their lone latency while the exponentiations, which carry > 95 % of
// the instructions, lose almost nothing.  -DMX_DEV_AUX_WAVE_PRIO=0 builds the library without it (tools/build_variant.py).
#pragma once
#include "mx_dev.hpp"          // MX_DEV_AUX_WAVE_PRIO, default 3
namespace mx {
__device__ __forceinline__ void aux_wave_priority() {
  if constexpr (MX_DEV_AUX_WAVE_PRIO != 0) __builtin_amdgcn_s_setprio(MX_DEV_AUX_WAVE_PRIO);
}
}  // namespace mx
