// Five-wavefront latency form of the N^2 pair kernel (mx_bipair.hpp): one decrypt(), or a batch small enough to leave most
// of the chip idle (paillier_shared_key.py:92 at distributed_keygen.py:345-349).  Translation unit of its own, built in
// parallel with the others; the launcher is called from mx_capi_n2.hip.
#include "mx_upload.hpp"
#include "mx_bipair.hpp"

namespace mxb {
template <int K>
static int launch(const mx::PowmodBiPairArgs& a, int64_t nblocks, hipStream_t s) {
  const size_t lds = mx::powmod_n2_bipair_lds_bytes<K, LIMB_BITS>();
  hipLaunchKernelGGL((mx::powmod_n2_bipair_kernel<K, LIMB_BITS>), dim3((unsigned)nblocks), dim3(mx::BP_THREADS), lds, s, a);
  MX_HIP(hipGetLastError());
#ifdef MX_DEV_BP_TRACE          // developer builds: cycles per phase and role of workgroup 0 (tools/bp_phase_probe.py reads stderr)
  {
    unsigned long long h[25] = {};
    MX_HIP(hipStreamSynchronize(s));
    MX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(mx::mx_bp_trace), sizeof(h)));
    const char* names[5] = {"AL", "AH", "BL", "BH", "Q"};
    for (int r = 0; r < 5; ++r)
      fprintf(stderr, "bp_trace K=%d %s: phase1 %llu wait1 %llu phase2 %llu wait2 %llu tape loop %llu\n", K, names[r], h[r * 4], h[r * 4 + 1], h[r * 4 + 2], h[r * 4 + 3], h[20 + r]);
  }
#endif
  return MX_OK;
}
// groups of 16 / 32 / 64 lanes: moduli of ~800 .. 2560 and ~2800 .. 5500 bits (key_length 1024, 2048, 4096)
bool n2_bipair_instance(int K) { return K == 16 || K == 32 || K == 64; }
int launch_n2_bipair(int K, const mx::PowmodBiPairArgs& a, int64_t nblocks, hipStream_t s) {
  switch (K) {
    case 16: return launch<16>(a, nblocks, s);
    case 32: return launch<32>(a, nblocks, s);
    case 64: return launch<64>(a, nblocks, s);
  }
  return MX_ERR_SIZE;
}
}  // namespace mxb
