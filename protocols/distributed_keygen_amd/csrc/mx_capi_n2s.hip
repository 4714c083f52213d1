// Two-wavefront instantiations of the N^2-modulus pair kernel (mx_powmod_n2_split.hpp) for 3 and 9 limbs per
// lane (translation unit of its own, built in parallel with the others).
#include "mx_upload.hpp"
#include "mx_powmod_n2_split.hpp"

namespace mxs {
template <int K, int L, bool TS, bool FR = (L == 3)>
static int launch_form(const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  size_t lds = mx::powmod_n2_split_lds_bytes<K, L>(FR);
  if (lds > 64 * 1024) {     // above the default limit of dynamic LDS per workgroup: opt in once per instance and device
    static bool allowed[MX_MAX_DEVICES] = {};
    MX_HIP(mx_allow_dynamic_lds(reinterpret_cast<const void*>(&mx::powmod_n2_split_kernel<K, L, LIMB_BITS, TS, FR>), (int)lds, allowed));
  }
  hipLaunchKernelGGL((mx::powmod_n2_split_kernel<K, L, LIMB_BITS, TS, FR>), dim3((unsigned)nblocks), dim3(64 * 2 * mx::N2_SPLIT_PAIRS), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}
// time-sliced instances exist for 9 limbs per lane and groups of at most 16 lanes (key_length up to 4096): a launch
// of wider groups that outnumbers the resident wavefronts is better served by more limbs per lane
template <int K, int L>
static int launch(bool ts, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  // friendly-modulus instances of the 9-limb kernel: groups of 8 and 16 lanes (key_length 2048 and 4096), where the
  // host found LIMB_BITS + 6 bits of room in R (a.friendly)
  if constexpr (L == LIMBS_PER_LANE && (K == 8 || K == 16)) {
    if (a.friendly) return ts ? launch_form<K, L, true, true>(a, nblocks, s) : launch_form<K, L, false, true>(a, nblocks, s);
  }
  if constexpr (L == LIMBS_PER_LANE && K <= 16) {
    if (ts) return launch_form<K, L, true>(a, nblocks, s);
  }
  if (ts) return MX_ERR_SIZE;
  return launch_form<K, L, false>(a, nblocks, s);
}

template <int L>
static int launch_l(int K, bool ts, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  switch (K) {
    case 1: return launch<1, L>(ts, a, nblocks, s);
    case 2: return launch<2, L>(ts, a, nblocks, s);
    case 4: return launch<4, L>(ts, a, nblocks, s);
    case 8: return launch<8, L>(ts, a, nblocks, s);
    case 16: return launch<16, L>(ts, a, nblocks, s);
    case 32: return launch<32, L>(ts, a, nblocks, s);
  }
  return MX_ERR_SIZE;
}

int launch_n2_split(int K, int L, bool ts, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  if (L == 3) return K == 64 ? launch<64, 3>(ts, a, nblocks, s) : launch_l<3>(K, ts, a, nblocks, s);
  if (L == LIMBS_PER_LANE) return launch_l<LIMBS_PER_LANE>(K, ts, a, nblocks, s);
  return MX_ERR_SIZE;
}
}  // namespace mxs
