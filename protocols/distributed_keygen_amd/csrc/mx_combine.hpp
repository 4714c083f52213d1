// Share recombination of threshold Paillier decryption, batched over ciphertexts.
// Replaces PaillierSharedKey.decrypt (paillier_shared_key.py:95-127) looped at
// distributed_keygen.py:510-515:
//     x = prod_i partial_i mod N^2                       (PSK:115-117)
//     if (x - 1) % N != 0: ValueError                    (PSK:119-123)  -> status 1
//     m = ((x - 1) // N * theta_inv) % N                 (PSK:125)
// The exact division by N is a Montgomery reduction of y = x - 1 modulo N whose quotient digits
// are recorded: REDC gives r = (y + Q N) / R with Q = -y N^-1 mod R; N | y  <=>  r in {0, N},
// and then y / N = (R - Q) mod R.  So the whole function is Montgomery products in the lane
// geometry of N^2 — no long division anywhere.
#pragma once
#include "mx_mont.hpp"
#include "mx_prio.hpp"

namespace mx {

struct CombineArgs {
  const u32* partials;   // [np][batch][limbs2] device
  u32* out;              // [batch][out_stride] device: limbs words of plaintext, then (if the row is
                         // wider) one status word and zeros — one row = one unit of an all-gather
  unsigned char* status; // [batch] device, or null
  const u32* n;          // [limbs2] device, N zero padded
  const u32* n2;         // [limbs2] device, N^2
  const u32* rmodn1;     // [limbs2] device, R1 mod N
  const u32* rmodn2;     // [limbs2] device, R2 mod N^2
  const u32* theta_inv;  // [limbs2] device
  long long batch;
  int limbs, limbs2, np, out_stride;
  int nblk1, nblk2;
};

template <int K, int L, int W>
__global__ void __launch_bounds__(64) combine_kernel(CombineArgs A) {
  aux_wave_priority();
  using M_t = Mont<K, L, W, true>;
  extern __shared__ u32 smem[];
  constexpr int GPW = 64 / K;
  const int gw = threadIdx.x / K;
  const long long elem_raw = (long long)blockIdx.x * GPW + gw;
  const bool valid = elem_raw < A.batch;
  const long long elem = valid ? elem_raw : A.batch - 1;
  u32* lds = smem + gw * M_t::LDS_WORDS;

  // ---- product of the partial decryptions modulo N^2
  M_t M2;
  M2.init(lds, A.nblk2);
  M2.load(M2.n, A.n2, A.limbs2);
  M2.setup_modulus();
  u32 one2[L], r2sq[L];
  M2.load(one2, A.rmodn2, A.limbs2);
  M2.compute_r2(r2sq, one2);
  u32 x[L];
  M2.load(x, A.partials + elem * A.limbs2, A.limbs2);
  M2.mul(x, x, r2sq);
  for (int i = 1; i < A.np; ++i) {
    u32 y[L];
    M2.load(y, A.partials + ((long long)i * A.batch + elem) * A.limbs2, A.limbs2);
    M2.mul(y, y, r2sq);
    M2.mul(x, x, y);
  }
  u32 xc[L];
  M2.from_mont_canonical(xc, x);                 // x in [0, N^2)
  const bool x_zero = M2.is_zero(xc);

  // ---- y = x - 1 (for x >= 1): add 2^(W*S) - 1, drop the carry
  u32 y[L];
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = (u64)xc[j] + M_t::MASK;
    M2.normalize_full(y, t);
  }

  // ---- exact division by N through a quotient-recording Montgomery reduction modulo N
  M_t M1;
  M1.init(lds, A.nblk1);
  M1.load(M1.n, A.n, A.limbs2);
  M1.setup_modulus();
  u32 one[L], q[L], r[L];
  M1.set_small(one, 1);
  M1.template mul<true>(r, y, one, q);           // r = (y + Q N) / R1, lazy
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = r[j];
    M1.normalize_full(r, t);
  }
  const bool r_zero = M1.is_zero(r);
  const bool r_is_n = M1.equal(r, M1.n);
  const bool divisible = (r_zero || r_is_n) && !x_zero;
  // u = (R1 - Q) mod R1 (limbs below nblk1*L), or 0 when r == 0
  u32 u[L];
  {
    u64 t[L];
    const bool in_range = M1.p < A.nblk1;
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = (in_range && r_is_n) ? (u64)(M_t::MASK - q[j]) : 0;
    if (M1.p == 0 && r_is_n) t[0] += 1;
    M1.normalize_full(u, t);
    if (!in_range) {
#pragma unroll
      for (int j = 0; j < L; ++j) u[j] = 0;
    }
  }

  // ---- m = u * theta_inv mod N
  u32 one1[L], r1sq[L], th[L], m[L];
  M1.load(one1, A.rmodn1, A.limbs2);
  M1.compute_r2(r1sq, one1);
  M1.load(th, A.theta_inv, A.limbs2);
  M1.mul(th, th, r1sq);                          // theta_inv * R1
  M1.mul(m, u, th);                              // u * theta_inv (lazy)
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = m[j];
    M1.normalize_full(m, t);
  }
  M1.cond_sub(m);
  if (!divisible) {
#pragma unroll
    for (int j = 0; j < L; ++j) m[j] = 0;
  }
  M1.store(A.out + elem * A.out_stride, A.limbs, m, valid);
  if (valid && M1.p == 0) {
    if (A.status) A.status[elem] = divisible ? 0 : 1;
    for (int k = A.limbs; k < A.out_stride; ++k) A.out[elem * A.out_stride + k] = (k == A.limbs && !divisible) ? 1u : 0u;
  }
}

}  // namespace mx
