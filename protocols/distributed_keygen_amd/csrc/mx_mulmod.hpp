// Batched modular multiplication  out[e] = a[e] * b[e] mod N  (one modulus per launch).
// The glue operation around the modexps: (1 + mN) * r^N of Paillier encryption, homomorphic
// addition of ciphertexts, and the product tree of a batched modular inversion (Montgomery's
// trick) used for the negative Lagrange exponents of paillier_shared_key.py:89-91.
#pragma once
#include "mx_mont.hpp"

namespace mx {

struct MulmodArgs {
  const u32* a;      // [batch][limbs] device
  const u32* b;      // [batch][limbs] device
  u32* out;          // [batch][limbs] device
  const u32* mod;    // [limbs] device
  const u32* rmodn;  // [limbs] device
  long long batch;
  int limbs, nblk;
};

template <int K, int L, int W>
__global__ void __launch_bounds__(64) mulmod_kernel(MulmodArgs A) {
  using M_t = Mont<K, L, W, true>;
  extern __shared__ u32 smem[];
  constexpr int GPW = 64 / K;
  const int gw = threadIdx.x / K;
  const long long elem_raw = (long long)blockIdx.x * GPW + gw;
  const bool valid = elem_raw < A.batch;
  const long long elem = valid ? elem_raw : A.batch - 1;
  M_t M;
  M.init(smem + gw * M_t::LDS_WORDS, A.nblk);
  M.load(M.n, A.mod, A.limbs);
  M.setup_modulus();
  u32 one_m[L], r2[L], x[L], y[L];
  M.load(one_m, A.rmodn, A.limbs);
  M.compute_r2(r2, one_m);
  M.load(x, A.a + elem * A.limbs, A.limbs);
  M.load(y, A.b + elem * A.limbs, A.limbs);
  M.mul(x, x, r2);          // a R
  M.mul(x, x, y);           // a b   (lazy, < 2N)
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = x[j];
    M.normalize_full(x, t);
  }
  M.cond_sub(x);
  M.store(A.out + elem * A.limbs, A.limbs, x, valid);
}

}  // namespace mx
