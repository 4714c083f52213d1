// Wide-geometry (L = 18) instantiations of the N^2-modulus pair kernel (third translation unit).
#include "mx_upload.hpp"
#include "mx_powmod_n2.hpp"

namespace mxw {
template <int K>
static int launch(const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  size_t lds = mx::powmod_n2_lds_bytes<K, LIMBS_PER_LANE_WIDE>();
  hipLaunchKernelGGL((mx::powmod_n2_kernel<K, LIMBS_PER_LANE_WIDE, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
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
