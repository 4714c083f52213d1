// Issue priority of the short kernels of a step (Jacobi filter, selection, R mod N, verdict, recombination, sieve).
//
// Several steps are in flight at once (bench.py, Engine._pipelined, BiprimeRound): a step's short kernels then share
// their SIMDs with 2-4 wavefronts of OTHER steps' exponentiation kernels, and the step's own exponentiation waits for
// them.  A wavefront that raises its priority (s_setprio, user levels 0..3) is picked first by the SIMD's instruction
// arbiter, so the short kernels run at close to their lone latency while the exponentiations, which carry > 95 % of
// the instructions, lose almost nothing.  -DMX_AUX_WAVE_PRIO=0 builds the library without it (tools/build_variant.py).
#pragma once
#ifndef MX_AUX_WAVE_PRIO
#define MX_AUX_WAVE_PRIO 3
#endif
namespace mx {
__device__ __forceinline__ void aux_wave_priority() {
  if constexpr (MX_AUX_WAVE_PRIO != 0) __builtin_amdgcn_s_setprio(MX_AUX_WAVE_PRIO);
}
}  // namespace mx
