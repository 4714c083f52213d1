// Per-group stream compaction: keep the first `keep` rows of every group whose flag equals 1, in
// order — the selection `if jacobi_symbol(g, N) != 1: continue ... stop at correct_param_biprime`
// of the reference's v-calculation loop (distributed_keygen.py:1084-1099), so that the generators
// go from the Jacobi kernel to the modexp kernel without leaving the device.
// One wavefront per group; rows that are not filled (fewer than `keep` flagged rows) are zeroed and
// the number of kept rows is reported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mx_prio.hpp"

namespace mx {

struct SelectArgs {
  const uint32_t* rows;      // [groups][group_size][limbs] device
  const signed char* flags;  // [groups][group_size] device
  uint32_t* out;             // [groups][keep][limbs] device
  int* counts;               // [groups] device: rows kept (<= keep)
  long long groups;
  int group_size, keep, limbs;
};

__global__ void __launch_bounds__(64) select_first_kernel(SelectArgs A) {
  aux_wave_priority();
  const long long g = blockIdx.x;
  const int lane = threadIdx.x;
  const signed char* fl = A.flags + g * A.group_size;
  const uint32_t* src = A.rows + g * (long long)A.group_size * A.limbs;
  uint32_t* dst = A.out + g * (long long)A.keep * A.limbs;
  int kept = 0;
  for (int base = 0; base < A.group_size && kept < A.keep; base += 64) {
    const int k = base + lane;
    const bool on = (k < A.group_size) && (fl[k] == 1);
    const unsigned long long m = __ballot(on);
    const int slot = kept + __popcll(m & ((1ull << lane) - 1ull));
    // the selected row is copied by its own lane, one word at a time (rows are short: <= 257 words)
    if (on && slot < A.keep) {
      for (int w = 0; w < A.limbs; ++w) dst[(long long)slot * A.limbs + w] = src[(long long)k * A.limbs + w];
    }
    kept += __popcll(m);
  }
  if (kept > A.keep) kept = A.keep;
  for (int idx = kept * A.limbs + lane; idx < A.keep * A.limbs; idx += 64) dst[idx] = 0u;
  if (lane == 0) A.counts[g] = kept;
}

}  // namespace mx
