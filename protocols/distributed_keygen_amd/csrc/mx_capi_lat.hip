// Latency instances of the generic-modulus modexp (mx_powmod.hpp) and of its per-group setup kernel (mx_setup.hpp):
// 3 limbs per lane, the products of the exponentiation modulo the friendly multiple of N.  For launches that leave
// SIMDs idle — the biprimality-test modexps of a key-generation round at the reference's batch sizes leave 2-25
// survivors, i.e. a few hundred to a thousand modexps (distributed_keygen.py:1084-1099 looped at :1313-1329): the
// time of such a launch is the dependent chain of ONE wavefront, and fewer limbs per lane shorten it.
// Translation unit of its own, built in parallel with the others.
#include "mx_upload.hpp"
#include "mx_powmod.hpp"
#include "mx_setup.hpp"
#include "mx_bimont.hpp"

namespace mxl {
template <int K>
static int launch_powmod(const mx::PowmodArgs& a, int64_t nblocks, hipStream_t s) {
  using M_t = mx::Mont<K, LIMBS_PER_LANE_LAT, LIMB_BITS, true>;
  const size_t lds = (size_t)(64 / K) * M_t::LDS_WORDS * 4;
  MxKernelTimer timer(s);
  if (a.nops > 0)
    hipLaunchKernelGGL((mx::powmod_kernel<K, LIMBS_PER_LANE_LAT, LIMB_BITS, true, true>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  else
    hipLaunchKernelGGL((mx::powmod_kernel<K, LIMBS_PER_LANE_LAT, LIMB_BITS, false, true>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

template <int K>
static int launch_rmodn(const mx::RmodnArgs& a, hipStream_t s) {
  using M_t = mx::Mont<K, LIMBS_PER_LANE_LAT, LIMB_BITS, true>;
  const int gpw = 64 / K;
  const int64_t nblocks = (a.groups + gpw - 1) / gpw;
  hipLaunchKernelGGL((mx::rmodn_kernel<K, LIMBS_PER_LANE_LAT, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), (size_t)gpw * M_t::LDS_WORDS * 4, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

// bipartite form (mx_bimont.hpp): two wavefronts per workgroup, groups of 4 .. 64 lanes
template <int K>
static int launch_bi(const mx::PowmodBiArgs& a, int64_t nblocks, hipStream_t s) {
  using M_t = mx::Mont<K, LIMBS_PER_LANE_LAT, LIMB_BITS, true, false>;
  const size_t lds = (size_t)mx::BI_PAIRS * (64 / K) * (3 * (LIMBS_PER_LANE_LAT * K + 4) + M_t::LDS_WORDS) * 4;
  MxKernelTimer timer(s);
  hipLaunchKernelGGL((mx::powmod_bi_kernel<K, LIMB_BITS>), dim3((unsigned)nblocks), dim3(128 * mx::BI_PAIRS), lds, s, a);
  MX_HIP(hipGetLastError());
#ifdef MX_DEV_BI_TRACE          // developer builds: cycles per phase of the products of pair 0 (tools/bi_phase_probe.py reads stderr)
  {
    unsigned long long h[16] = {};
    MX_HIP(hipStreamSynchronize(s));
    MX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(mx::mx_bi_trace), sizeof(h)));
    fprintf(stderr, "bi_trace K=%d pivot=%d of %d: L half %llu wait %llu post-idle %llu barrier2+read %llu | H half %llu pre %llu wait %llu post %llu barrier2 %llu\n",
            K, a.h_lo, 3 * a.nblk, h[0], h[2], h[3], h[4], h[8], h[9], h[10], h[11], h[12]);
  }
#endif
  return MX_OK;
}
template <int K>
static int launch_bis(const mx::BiSetupArgs& a, hipStream_t s) {
  using M_t = mx::Mont<K, LIMBS_PER_LANE_LAT, LIMB_BITS, true>;
  const int gpw = 64 / K;
  const int64_t nblocks = (a.groups + gpw - 1) / gpw;
  hipLaunchKernelGGL((mx::bisetup_kernel<K, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), (size_t)gpw * M_t::LDS_WORDS * 4, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}
#define MX_BI_CASES(FN, ...) \
  switch (K) {               \
    case 4: return FN<4>(__VA_ARGS__); case 8: return FN<8>(__VA_ARGS__); case 16: return FN<16>(__VA_ARGS__); \
    case 32: return FN<32>(__VA_ARGS__); case 64: return FN<64>(__VA_ARGS__);                                  \
  }                                                                                                            \
  return MX_ERR_SIZE;

#define MX_LAT_CASES(FN, ...) \
  switch (K) {                \
    case 1: return FN<1>(__VA_ARGS__); case 2: return FN<2>(__VA_ARGS__); case 4: return FN<4>(__VA_ARGS__);     \
    case 8: return FN<8>(__VA_ARGS__); case 16: return FN<16>(__VA_ARGS__); case 32: return FN<32>(__VA_ARGS__); \
    case 64: return FN<64>(__VA_ARGS__);                                                                         \
  }                                                                                                              \
  return MX_ERR_SIZE;

int launch_powmod_lat(int K, const mx::PowmodArgs& a, int64_t nblocks, hipStream_t s) { MX_LAT_CASES(launch_powmod, a, nblocks, s) }
int launch_rmodn_lat(int K, const mx::RmodnArgs& a, hipStream_t s) { MX_LAT_CASES(launch_rmodn, a, s) }
int launch_powmod_bi(int K, const mx::PowmodBiArgs& a, int64_t nblocks, hipStream_t s) { MX_BI_CASES(launch_bi, a, nblocks, s) }
int launch_bisetup(int K, const mx::BiSetupArgs& a, hipStream_t s) { MX_BI_CASES(launch_bis, a, s) }
}  // namespace mxl
