// Modular inverse  out[e] = values[e]^-1 mod M  for a handful of elements: the root of the product
// tree that replaces `mod_inv(ciphertext_value, n_square)` per ciphertext for a negative Lagrange
// exponent (paillier_shared_key.py:89-91), and `mod_inv(theta, n)` of a key (paillier_shared_key.py:50).
//
// No multiplications and one long dependent chain per element, so the layout differs from the
// Montgomery kernels: ONE WAVEFRONT PER ELEMENT, the big integers spread over all 64 lanes (LPL
// radix-2^32 limbs per lane, lane 0 least significant).  Every step of the algorithm costs a few
// instructions per lane regardless of the operand size:
//   shift by one bit     v_alignbit per limb, the bit crossing a lane boundary by DPP wave_shl/shr
//   add / subtract       lane-local carry chain, then the carries BETWEEN lanes from two wave ballots
//                        (generate G, propagate P):  carry-in mask = (P + (G << 1)) ^ P   — a 64-bit
//                        scalar addition does the ripple across the 64 lanes at once
//   compare              per-lane "greater"/"less" ballots compared as 64-bit integers
//   parity / zero test   readfirstlane, ballot
// All branch conditions are wave-uniform (scalar), so the wavefront never diverges.
//
// Algorithm: Kaliski's almost-inverse (u = M, v = a, r = 0, s = 1; k halving steps keeping
// M = u*s + v*r), giving a^-1 * 2^k mod M with only shifts, additions and subtractions, followed by
// k halvings modulo M.  gcd(a, M) != 1 (incl. a = 0) is reported in the status byte, like the
// ValueError of `pow(a, -1, M)`.
#pragma once
#include "mx_lanes.hpp"

namespace mx {

struct ModinvArgs {
  const u32* vals;        // [batch][limbs] device
  const u32* mod;         // [limbs] device
  u32* out;               // [batch][limbs] device
  unsigned char* status;  // [batch] device: 0 = ok, 1 = not invertible
  long long batch;
  int limbs;
};

template <int LPL>
struct WaveInt {
  u32 x[LPL];

  static __device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }

  __device__ __forceinline__ void load(const u32* __restrict__ src, int limbs) {
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const int idx = lane() * LPL + j;
      x[j] = idx < limbs ? src[idx] : 0u;
    }
  }
  __device__ __forceinline__ void store(u32* __restrict__ dst, int limbs) const {
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const int idx = lane() * LPL + j;
      if (idx < limbs) dst[idx] = x[j];
    }
  }
  __device__ __forceinline__ void set_small(u32 v) {
#pragma unroll
    for (int j = 0; j < LPL; ++j) x[j] = 0;
    if (lane() == 0) x[0] = v;
  }
  __device__ __forceinline__ bool is_zero() const {
    u32 o = 0;
#pragma unroll
    for (int j = 0; j < LPL; ++j) o |= x[j];
    return __ballot(o != 0) == 0;
  }
  __device__ __forceinline__ bool is_one() const {
    u32 o = 0;
#pragma unroll
    for (int j = 1; j < LPL; ++j) o |= x[j];
    const bool bad = lane() == 0 ? (o != 0 || x[0] != 1u) : ((o | x[0]) != 0);
    return __ballot(bad) == 0;
  }
  __device__ __forceinline__ u32 low() const { return (u32)__builtin_amdgcn_readfirstlane((int)x[0]); }

  __device__ __forceinline__ void shr1() {
    const u32 next = dpp_mov<DPP_WAVE_SHL1, 0xF, 0xF, true>(0, x[0]);       // lane l <- lane l+1, top lane <- 0
#pragma unroll
    for (int j = 0; j < LPL - 1; ++j) x[j] = __builtin_amdgcn_alignbit(x[j + 1], x[j], 1);
    x[LPL - 1] = __builtin_amdgcn_alignbit(next, x[LPL - 1], 1);
  }
  __device__ __forceinline__ void shl1() {
    const u32 prev = dpp_mov<DPP_WAVE_SHR1, 0xF, 0xF, true>(0, x[LPL - 1]);  // lane l <- lane l-1, lane 0 <- 0
#pragma unroll
    for (int j = LPL - 1; j > 0; --j) x[j] = __builtin_amdgcn_alignbit(x[j], x[j - 1], 31);
    x[0] = __builtin_amdgcn_alignbit(x[0], prev, 31);
  }
  // carries between lanes from the generate / propagate ballots
  static __device__ __forceinline__ bool carry_in(bool g, bool p) {
    const unsigned long long G = __ballot(g), P = __ballot(p);
    const unsigned long long cin = (P + (G << 1)) ^ P;
    return (cin >> lane()) & 1ull;
  }
  __device__ __forceinline__ void add(const WaveInt& y) {
    u32 c = 0, all = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const u64 t = (u64)x[j] + y.x[j] + c;
      x[j] = (u32)t;
      c = (u32)(t >> 32);
      all &= x[j];
    }
    if (carry_in(c != 0, all == 0xFFFFFFFFu)) {
      u32 k = 1;
#pragma unroll
      for (int j = 0; j < LPL; ++j) { x[j] += k; k = (x[j] == 0u) ? k : 0u; }
    }
  }
  __device__ __forceinline__ void sub(const WaveInt& y) {
    u32 b = 0, any = 0;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const u64 t = (u64)x[j] - y.x[j] - b;
      x[j] = (u32)t;
      b = (u32)(t >> 63);
      any |= x[j];
    }
    if (carry_in(b != 0, any == 0u)) {
      u32 k = 1;
#pragma unroll
      for (int j = 0; j < LPL; ++j) { const u32 old = x[j]; x[j] = old - k; k = (old == 0u) ? k : 0u; }
    }
  }
  // this > y
  __device__ __forceinline__ bool gt(const WaveInt& y) const {
    int s = 0;
#pragma unroll
    for (int j = LPL - 1; j >= 0; --j) s = (s != 0) ? s : (x[j] > y.x[j] ? 1 : (x[j] < y.x[j] ? -1 : 0));
    return __ballot(s > 0) > __ballot(s < 0);
  }
};

template <int LPL>
__global__ void __launch_bounds__(64) modinv_kernel(ModinvArgs A) {
  using WI = WaveInt<LPL>;
  const long long e = blockIdx.x;
  WI m, u, v, r, s;
  m.load(A.mod, A.limbs);
  u = m;
  v.load(A.vals + e * A.limbs, A.limbs);
  r.set_small(0);
  s.set_small(1);
  int k = 0;
  const int bound = 64 * A.limbs + 8;            // k <= 2 * bits(M)
  while (k < bound && !v.is_zero()) {
    if (!(u.low() & 1u)) { u.shr1(); s.shl1(); }
    else if (!(v.low() & 1u)) { v.shr1(); r.shl1(); }
    else if (u.gt(v)) { u.sub(v); u.shr1(); r.add(s); s.shl1(); }
    else { v.sub(u); v.shr1(); s.add(r); r.shl1(); }
    ++k;
  }
  const bool ok = v.is_zero() && u.is_one();
  if (!m.gt(r)) r.sub(m);                        // r < 2M  ->  r mod M
  WI x = m;
  x.sub(r);                                      // a^-1 * 2^k mod M
  for (int i = 0; i < k; ++i) {
    if (x.low() & 1u) x.add(m);
    x.shr1();
  }
  if (!ok) x.set_small(0);
  x.store(A.out + e * A.limbs, A.limbs);
  if (WI::lane() == 0) A.status[e] = ok ? 0 : 1;
}

}  // namespace mx
