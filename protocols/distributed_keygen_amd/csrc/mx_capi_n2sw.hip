// Two-wavefront instantiations of the N^2-modulus pair kernel (mx_powmod_n2_split.hpp) for the wide geometry
// (18 limbs per lane; translation unit of its own, built in parallel with the others).
#include "mx_upload.hpp"
#include "mx_powmod_n2_split.hpp"

namespace mxs {
template <int K, bool TS>
static int launch_form(const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  size_t lds = mx::powmod_n2_split_lds_bytes<K, LIMBS_PER_LANE_WIDE>();
  if (lds > 64 * 1024) {     // above the default limit of dynamic LDS per workgroup: opt in once per instance and device
    static bool allowed[MX_MAX_DEVICES] = {};
    MX_HIP(mx_allow_dynamic_lds(reinterpret_cast<const void*>(&mx::powmod_n2_split_kernel<K, LIMBS_PER_LANE_WIDE, LIMB_BITS, TS>), (int)lds, allowed));
  }
  hipLaunchKernelGGL((mx::powmod_n2_split_kernel<K, LIMBS_PER_LANE_WIDE, LIMB_BITS, TS>), dim3((unsigned)nblocks), dim3(64 * 2 * mx::N2_SPLIT_PAIRS), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

// Time-sliced instances at 18 limbs per lane: groups of 4 and 8 lanes (key_length 2048 and 4096), one resident
// workgroup per CU.  (Rounds 3 and 4 did not build them: first the unit loop pushed them over 256 registers, then — at
// 239 registers, no scratch — they lost to the 9-limb form, 45.0-47.6 against 42.9 ms for 10 000 ciphertexts.  That was
// the FIFO unit queue, not the instances: mx_powmod_n2_split.hpp, tools/ts_schedule_model.py.)
template <int K>
static int launch(bool ts, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  if constexpr (K == 4 || K == 8) {
    if (ts) return launch_form<K, true>(a, nblocks, s);
  } else {
    if (ts) return MX_ERR_SIZE;
  }
  return launch_form<K, false>(a, nblocks, s);
}

int launch_n2_split_wide(int K, bool ts, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  switch (K) {
    case 1: return launch<1>(ts, a, nblocks, s);
    case 2: return launch<2>(ts, a, nblocks, s);
    case 4: return launch<4>(ts, a, nblocks, s);
    case 8: return launch<8>(ts, a, nblocks, s);
    case 16: return launch<16>(ts, a, nblocks, s);
  }
  return MX_ERR_SIZE;
}
}  // namespace mxs
