// Arithmetic in the Shamir field Z_P of the key generation (P = nextprime(2^(2(L + log2 n))),
// distributed_keygen.py:647-651), batched over the candidates of a round:
//
//   fma_kernel      out[e] = (a[e] * b[e] + c[e]) mod P
//       this party's share of the candidate modulus, `prime_candidate_p * prime_candidate_q` followed
//       by `candidate_n += zero` (distributed_keygen.py:1274-1277; ShamirVariable.__mul__/__add__,
//       utils.py:205-250: share-wise product and sum modulo P)
//   lincomb_kernel  out[e] = sum_t coeff[t] * x[t][e] mod P
//       `candidate_n.reconstruct()` (distributed_keygen.py:1284; utils.py:263-270, 465-471): Lagrange
//       interpolation at 0 of the parties' shares, coeff[t] = prod_{j != t} x_j / (x_j - x_t) mod P —
//       the candidate moduli N of the round, which then go to the sieve without leaving the device.
//
// Same lane-distributed Montgomery engine as the modexps (mx_mont.hpp); one modulus per launch.
#pragma once
#include "mx_mont.hpp"

namespace mx {

struct FieldArgs {
  const u32* a;       // fma: [batch][limbs];  lincomb: x, [terms][batch][limbs]
  const u32* b;       // fma: [batch][limbs];  lincomb: coefficients, [terms][limbs]
  const u32* c;       // fma: [batch][limbs];  lincomb: unused
  u32* out;           // [batch][limbs]
  const u32* mod;     // [limbs]
  const u32* rmodn;   // [limbs]: R mod P
  long long batch;
  int limbs, nblk, terms;
};

template <int K, int L, int W>
__global__ void __launch_bounds__(64) fma_kernel(FieldArgs A) {
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
  u32 one_m[L], r2[L], x[L], y[L], z[L];
  M.load(one_m, A.rmodn, A.limbs);
  M.compute_r2(r2, one_m);
  M.load(x, A.a + elem * A.limbs, A.limbs);
  M.load(y, A.b + elem * A.limbs, A.limbs);
  M.load(z, A.c + elem * A.limbs, A.limbs);
  M.mul(x, x, r2);          // a R
  M.mul(x, x, y);           // a b        (lazy, < 2P)
  M.add(x, x, z);           // a b + c    (< 2P + c)
  M.mul(x, x, one_m);       // (a b + c) R / R = a b + c mod P, lazy < 2P for any input < R
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = x[j];
    M.normalize_full(x, t);
  }
  M.cond_sub(x);
  M.store(A.out + elem * A.limbs, A.limbs, x, valid);
}

template <int K, int L, int W>
__global__ void __launch_bounds__(64) lincomb_kernel(FieldArgs A) {
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
  u32 one_m[L], r2[L], acc[L];
  M.load(one_m, A.rmodn, A.limbs);
  M.compute_r2(r2, one_m);
#pragma unroll
  for (int j = 0; j < L; ++j) acc[j] = 0;
  for (int t = 0; t < A.terms; ++t) {
    u32 cf[L], x[L];
    M.load(cf, A.b + (long long)t * A.limbs, A.limbs);
    M.mul(cf, cf, r2);                                   // coeff_t R
    M.load(x, A.a + ((long long)t * A.batch + elem) * A.limbs, A.limbs);
    M.mul(x, x, cf);                                     // coeff_t x_t   (lazy, < 2P)
    M.add(acc, acc, x);
    if ((t & 3) == 3) M.mul(acc, acc, one_m);            // keep the lazy sum below 10 P: acc <- acc mod P (lazy)
  }
  M.mul(acc, acc, one_m);
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = acc[j];
    M.normalize_full(acc, t);
  }
  M.cond_sub(acc);
  M.store(A.out + elem * A.limbs, A.limbs, acc, valid);
}

}  // namespace mx
