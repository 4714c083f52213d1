// Small-prime sieve: out[e] = 1 iff some prime of the list divides candidate e.
// Replaces __small_prime_divisors_test (distributed_keygen.py:1197-1209) applied to every
// candidate modulus of a batch (distributed_keygen.py:1288-1292).
//
// N mod l is evaluated as  sum_j N_j * (2^(32j) mod l)  — one v_mad_u64_u32 per (limb, prime),
// no division in the loop — and divisibility of that 64-bit sum by the odd prime l is decided
// by the exact-division test  (x * l^-1 mod 2^64) <= floor((2^64-1)/l).
// The 64-bit column holds `chunk` products (each < 2^32 l) on top of a folded value: after every `chunk` limbs
// the column is folded with the table's own 2^32 mod l:  x <- lo32(x) + hi32(x) * (2^32 mod l)  < 2^32 (l + 1),
// which leaves the residue unchanged.  The host sets chunk = floor((2^32 - 2) / max prime) - 1, so that
// 2^32 ((chunk + 1) l + 1) < 2^64: no fold at all for lists below 2^21 (the reference's default threshold is 2000),
// one every 63 limbs at 2^26, one per limb only above 2^30.  Primes < 2^31.
// One wavefront handles SIEVE_C candidates: lanes run over primes (coalesced table reads), the
// candidates' limbs are staged transposed in LDS so that one ds_read_b128 feeds four MACs.
#pragma once
#include "mx_lanes.hpp"

namespace mx {

constexpr int SIEVE_C = 8;   // candidates per wavefront

struct SieveArgs {
  const u32* cands;    // [batch][limbs] device
  unsigned char* out;  // [batch] device
  const u32* primes;   // [np] device
  u32* pw;             // [limbs][np_pad] device: 2^(32 j) mod prime_k
  u64* inv;            // [np_pad] device: prime^-1 mod 2^64
  u64* lim;            // [np_pad] device: floor((2^64-1)/prime)
  long long batch;
  int limbs;
  int np;
  int np_pad;          // multiple of 64
  int chunk;           // limbs between two folds of the 64-bit columns (>= limbs: never)
};

// one thread per prime: powers of 2^32, inverse and limit
__global__ void sieve_setup_kernel(SieveArgs A) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= A.np_pad) return;
  if (k >= A.np) {   // padding lanes (masked out in the main kernel)
    for (int j = 0; j < A.limbs; ++j) A.pw[(long long)j * A.np_pad + k] = 0;
    A.inv[k] = 1; A.lim[k] = 0;
    return;
  }
  u64 l = A.primes[k];
  u64 r = 1 % l;
  for (int j = 0; j < A.limbs; ++j) {
    A.pw[(long long)j * A.np_pad + k] = (u32)r;
    r = (r << 32) % l;
  }
  u64 x = l;                       // Newton inverse mod 2^64 (l odd): 3 -> 6 -> ... -> 96 bits
  for (int i = 0; i < 5; ++i) x *= 2 - l * x;
  A.inv[k] = x;
  A.lim[k] = ~0ull / l;
}

__global__ void __launch_bounds__(64) sieve_kernel(SieveArgs A) {
  extern __shared__ u32 smem[];   // [limbs][SIEVE_C]
  const int lane = threadIdx.x;
  const long long e0 = (long long)blockIdx.x * SIEVE_C;
  // stage candidates transposed; rows beyond the batch are zero (zero is reported divisible,
  // but such rows are never stored)
  for (int idx = lane; idx < A.limbs * SIEVE_C; idx += 64) {
    int c = idx / A.limbs, j = idx - c * A.limbs;   // consecutive lanes read consecutive words
    long long e = e0 + c;
    smem[j * SIEVE_C + c] = (e < A.batch) ? A.cands[e * A.limbs + j] : 0u;
  }
  __syncthreads();
  unsigned hit[SIEVE_C];
#pragma unroll
  for (int c = 0; c < SIEVE_C; ++c) hit[c] = 0;
  for (int k0 = 0; k0 < A.np_pad; k0 += 64) {
    const int k = k0 + lane;
    u64 acc[SIEVE_C];
#pragma unroll
    for (int c = 0; c < SIEVE_C; ++c) acc[c] = 0;
    for (int j0 = 0; j0 < A.limbs; j0 += A.chunk) {
      const int j1 = j0 + A.chunk < A.limbs ? j0 + A.chunk : A.limbs;
      for (int j = j0; j < j1; ++j) {
        u32 w = A.pw[(long long)j * A.np_pad + k];
        const uint4 lo = *reinterpret_cast<const uint4*>(&smem[j * SIEVE_C]);
        const uint4 hi = *reinterpret_cast<const uint4*>(&smem[j * SIEVE_C + 4]);
        acc[0] += (u64)lo.x * w; acc[1] += (u64)lo.y * w; acc[2] += (u64)lo.z * w; acc[3] += (u64)lo.w * w;
        acc[4] += (u64)hi.x * w; acc[5] += (u64)hi.y * w; acc[6] += (u64)hi.z * w; acc[7] += (u64)hi.w * w;
      }
      if (j1 < A.limbs) {          // more limbs to come: fold the columns (limbs >= 2 here, so row 1 of the table exists)
        const u32 w1 = A.pw[(long long)A.np_pad + k];          // 2^32 mod l
#pragma unroll
        for (int c = 0; c < SIEVE_C; ++c) acc[c] = (acc[c] & 0xFFFFFFFFull) + (acc[c] >> 32) * w1;
      }
    }
    const u64 inv = A.inv[k], lim = A.lim[k];
#pragma unroll
    for (int c = 0; c < SIEVE_C; ++c) hit[c] |= (k < A.np && acc[c] * inv <= lim) ? 1u : 0u;
  }
#pragma unroll
  for (int c = 0; c < SIEVE_C; ++c) {
    bool any = __any(hit[c] != 0);
    if (lane == 0 && e0 + c < A.batch) A.out[e0 + c] = any ? 1 : 0;
  }
}

}  // namespace mx
