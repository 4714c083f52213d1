// Batched modular exponentiation  out[e] = bases[e] ^ exp[g(e)] mod N[g(e)]  on gfx950.
//
// Replaces the reference's scalar pow_mod calls in their two batch shapes:
//   * biprimality test v-values  g^((N-p_i-q_i+1)/4) mod N     (distributed_keygen.py:1084-1099):
//     `group_size` (=40) consecutive bases share one (modulus, exponent) pair;
//   * partial decryption         c^exp_i mod N^2               (paillier_shared_key.py:92, looped at
//     distributed_keygen.py:463-466): one group spans the whole batch.
//
// One wavefront handles 64/K elements; an element is a group of K lanes (mx_mont.hpp).
// Fixed-window exponentiation, window table in HBM laid out [entry][limb][lane] so every table
// access is a fully coalesced 256-byte row per wave.
#pragma once
#include "mx_mont.hpp"

namespace mx {

struct PowmodArgs {
  const u32* bases;   // [batch][limbs]      device, radix 2^32 little-endian words
  u32* out;           // [batch][limbs]      device
  const u32* mods;    // [groups][limbs]     device (workspace copy)
  const u32* rmodn;   // [groups][limbs]     device: R mod N per group (rmodn_kernel, mx_setup.hpp)
  const u32* exps;    // [groups][elimbs]    device
  u32* table;         // [2^win][L][nlanes]  device
  i64 batch;
  i64 group_size;
  int limbs;
  int elimbs;
  int ndigits;        // ceil(max exponent bits / win), >= 1
  int win;            // window width in bits
  int nblk;
  // Sliding-window schedule (shared exponent only, i.e. one group): nops > 0 selects it.
  // ops[k] = (squarings << 16) | (table index + 1); table index + 1 == 0: no multiplication.
  // The table then holds the odd powers x^(2k+1), k < 2^(win-1).
  const u32* ops;
  int nops;
};

// SLIDING = the shared-exponent schedule (A.ops), otherwise the fixed window with per-group digits;
// two kernels rather than one branch so that neither carries the other's register pressure.
// FR (the 3-limb latency instances): the products of the exponentiation reduce modulo the friendly multiple of N
// (mx_mont.hpp: F_FRIENDLY — no multiplication on the quotient digit's dependent chain); N~ + 1 is derived from each
// group's modulus in the kernel, and the conversion out of the Montgomery domain runs the plain passes, which bring
// the result below N + 1 whatever multiple of N it carried (x < 2 N~ < R).
template <int K, int L, int W, bool SLIDING, bool FR = false>
__global__ void __launch_bounds__(64, (L > 9 ? 2 : 1)) powmod_kernel(PowmodArgs A) {
  using M_t = Mont<K, L, W, true>;
  extern __shared__ u32 smem[];
  constexpr int GPW = 64 / K;
  const int lane = threadIdx.x;
  const int gw = lane / K;
  const i64 elem_raw = (i64)blockIdx.x * GPW + gw;
  const bool valid = elem_raw < A.batch;
  const i64 elem = valid ? elem_raw : A.batch - 1;   // surplus groups redo the last element (no store)
  const i64 grp = elem / A.group_size;
  const i64 nlanes = (i64)gridDim.x * 64;
  const i64 gl = (i64)blockIdx.x * 64 + lane;

  M_t M;
  M.init(smem + gw * M_t::LDS_WORDS, A.nblk);
  M.load(M.n, A.mods + grp * A.limbs, A.limbs);
  M.setup_modulus();

  u32 one_m[L], r2[L], x[L];
  M.load(one_m, A.rmodn + grp * A.limbs, A.limbs);   // Montgomery form of 1
  M.compute_r2(r2, one_m);
  M.load(x, A.bases + elem * A.limbs, A.limbs);
  M.mul(x, x, r2);                                   // x = base * R mod N (lazy)
  if constexpr (FR) M.compute_friendly();
  auto MUL = [&](u32 (&r)[L], const u32 (&a)[L], const u32 (&b)[L]) {
    if constexpr (FR) M.template mul_friendly<false>(r, a, b); else M.mul(r, a, b);
  };
  auto SQR = [&](u32 (&r)[L], const u32 (&a)[L]) {
    if constexpr (FR) M.template mul_friendly<true>(r, a, a); else M.sqr(r, a);
  };

  u32* tbl = A.table + gl;
  if constexpr (SLIDING) {
    // ---- sliding window over a shared exponent: odd powers only, multiplications only where
    // the exponent has a window (the schedule is the same for every lane: uniform control flow)
    u32 x2[L], y[L];
    SQR(x2, x);
#pragma unroll
    for (int j = 0; j < L; ++j) { y[j] = x[j]; tbl[(i64)j * nlanes] = x[j]; }
    const int nodd = 1 << (A.win - 1);
    for (int k = 1; k < nodd; ++k) {
      MUL(y, y, x2);
#pragma unroll
      for (int j = 0; j < L; ++j) tbl[((i64)k * L + j) * nlanes] = y[j];
    }
    u32 acc[L];
    {
      const u32 first = A.ops[0] & 0xFFFFu;
#pragma unroll
      for (int j = 0; j < L; ++j) acc[j] = tbl[((i64)(first - 1) * L + j) * nlanes];
    }
    for (int k = 1; k < A.nops; ++k) {
      const u32 op = A.ops[k];
      const int nsq = (int)(op >> 16);
      const u32 idx1 = op & 0xFFFFu;
      // narrow geometry: the table row is requested before the squarings and used after them
      // (latency fully hidden); wide geometry: registers are the scarcer resource, so the row is
      // fetched after the squarings (one exposed L2/HBM round trip per ~8 Montgomery products)
      u32 f[L];
      if (L <= 9 && idx1) {
#pragma unroll
        for (int j = 0; j < L; ++j) f[j] = tbl[((i64)(idx1 - 1) * L + j) * nlanes];
      }
      for (int s = 0; s < nsq; ++s) SQR(acc, acc);
      if (L > 9 && idx1) {
#pragma unroll
        for (int j = 0; j < L; ++j) f[j] = tbl[((i64)(idx1 - 1) * L + j) * nlanes];
      }
      if (idx1) MUL(acc, acc, f);
    }
    u32 res[L];
    M.from_mont_canonical(res, acc);
    M.store(A.out + elem * A.limbs, A.limbs, res, valid);
  } else {
  // ---- fixed window (per-group exponents): tbl[0] = 1, tbl[1] = x, tbl[k] = tbl[k-1] * x
  const int nent = 1 << A.win;
#pragma unroll
  for (int j = 0; j < L; ++j) tbl[(i64)j * nlanes] = one_m[j];
#pragma unroll
  for (int j = 0; j < L; ++j) tbl[((i64)L + j) * nlanes] = x[j];
  {
    u32 y[L];
#pragma unroll
    for (int j = 0; j < L; ++j) y[j] = x[j];
    for (int k = 2; k < nent; ++k) {
      MUL(y, y, x);
#pragma unroll
      for (int j = 0; j < L; ++j) tbl[((i64)k * L + j) * nlanes] = y[j];
    }
  }

  const u32* ex = A.exps + grp * A.elimbs;
  const u32 wmask = (1u << A.win) - 1u;
  auto digit = [&](int d) -> u32 {
    int bit = d * A.win;
    int w = bit >> 5, off = bit & 31;
    u64 v = (u64)ex[w] | ((u64)(w + 1 < A.elimbs ? ex[w + 1] : 0u) << 32);
    return (u32)(v >> off) & wmask;
  };

  u32 acc[L];
  {
    u32 dg = digit(A.ndigits - 1);
#pragma unroll
    for (int j = 0; j < L; ++j) acc[j] = tbl[((i64)dg * L + j) * nlanes];
  }
  for (int d = A.ndigits - 2; d >= 0; --d) {
    u32 dg = digit(d);
    u32 y[L];
#pragma unroll
    for (int j = 0; j < L; ++j) y[j] = tbl[((i64)dg * L + j) * nlanes];   // issued early, used after the squarings
    for (int s = 0; s < A.win; ++s) SQR(acc, acc);
    MUL(acc, acc, y);
  }

  u32 res[L];
  M.from_mont_canonical(res, acc);
  M.store(A.out + elem * A.limbs, A.limbs, res, valid);
  }
}

}  // namespace mx
