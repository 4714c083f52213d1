// Per-modulus setup for the kernels that take one modulus PER GROUP (biprimality-test modexps and
// verdict, distributed_keygen.py:1084-1099 and :1147-1158): the Montgomery form of 1, R mod N, for
// every candidate modulus, computed on the device from the moduli themselves.  The candidate moduli
// of a keygen round (distributed_keygen.py:1284) can therefore stay device-resident from their
// reconstruction to the verdict; no host big-integer work and no host operand besides the moduli.
#pragma once
#include "mx_mont.hpp"
#include "mx_prio.hpp"

namespace mx {

struct RmodnArgs {
  const u32* mods;   // [groups][limbs] device
  u32* rmodn;        // [groups][limbs] device (out)
  long long groups;
  int limbs, nblk;
};

template <int K, int L, int W>
__global__ void __launch_bounds__(64) rmodn_kernel(RmodnArgs A) {
  aux_wave_priority();
  using M_t = Mont<K, L, W, true>;
  extern __shared__ u32 smem[];
  constexpr int GPW = 64 / K;
  const int gw = threadIdx.x / K;
  const long long g_raw = (long long)blockIdx.x * GPW + gw;
  const bool valid = g_raw < A.groups;
  const long long g = valid ? g_raw : A.groups - 1;
  M_t M;
  M.init(smem + gw * M_t::LDS_WORDS, A.nblk);
  M.load(M.n, A.mods + g * A.limbs, A.limbs);
  u32 x[L];
  M.rmodn_by_doubling(x);
  M.store(A.rmodn + g * A.limbs, A.limbs, x, valid);
}

}  // namespace mx
