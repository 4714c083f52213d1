// Wide-geometry (L = 18) instantiations of the N^2-modulus pair kernel (third translation unit).
#include "mx_upload.hpp"
#include "mx_powmod_n2.hpp"

namespace mxw {
template <int K, bool FR>
static int launch_form(const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  size_t lds = mx::powmod_n2_lds_bytes<K, LIMBS_PER_LANE_WIDE>(FR);
  hipLaunchKernelGGL((mx::powmod_n2_kernel<K, LIMBS_PER_LANE_WIDE, LIMB_BITS, FR>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}
// friendly-modulus instances (mx_powmod_n2.hpp) for groups of 4 and 8 lanes — key_length 2048 and 4096, the launches
// that fill the machine — where the host found LIMB_BITS + 6 bits of room in R (a.friendly)
template <int K>
static int launch(const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  if constexpr (K == 4 || K == 8) {
    if (a.friendly) return launch_form<K, true>(a, nblocks, s);
  }
  return launch_form<K, false>(a, nblocks, s);
}

int launch_n2_wide(int K, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  switch (K) {
    case 1: return launch<1>(a, nblocks, s);
    case 2: return launch<2>(a, nblocks, s);
    case 4: return launch<4>(a, nblocks, s);
    case 8: return launch<8>(a, nblocks, s);
    case 16: return launch<16>(a, nblocks, s);
  }
  return MX_ERR_SIZE;
}
}  // namespace mxw

#ifdef MX_DEV_PRIVATE_PAD_WORDS
// developer build: faults the pad check of powmod_n2_kernel counted since the last call (and the first 32 of them,
// 7 words each: tag, workgroup, lane, index, expected, found, first*2+last); resets the counter
extern "C" int mx_debug_pad_faults(uint32_t* log_words, int max_entries) {
  uint32_t n = 0, zero = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(mx::g_pad_faults), 4) != hipSuccess) return -5;
  const int m = (int)(n < 32u ? n : 32u) < max_entries ? (int)(n < 32u ? n : 32u) : max_entries;
  if (m > 0 && log_words && hipMemcpyFromSymbol(log_words, HIP_SYMBOL(mx::g_pad_fault_log), (size_t)m * sizeof(mx::PadFault)) != hipSuccess) return -5;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mx::g_pad_faults), &zero, 4) != hipSuccess) return -5;
  return (int)(n > 0x7FFFFFFFu ? 0x7FFFFFFFu : n);
}
#endif
