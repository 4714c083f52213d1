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
// M = u*s + v*r), giving a^-1 * 2^k mod M with only shifts, additions and subtractions — the trailing
// zeros a subtraction leaves taken in the same trip of the loop —, followed by k halvings modulo M,
// 32 at a time (one multiply-accumulate of M per word: a Montgomery reduction step).  gcd(a, M) != 1 (incl. a = 0) is reported in the status byte, like the
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
  // the same by 1 .. 31 bits
  __device__ __forceinline__ void shr(int n) {
    const u32 next = dpp_mov<DPP_WAVE_SHL1, 0xF, 0xF, true>(0, x[0]);
#pragma unroll
    for (int j = 0; j < LPL - 1; ++j) x[j] = __builtin_amdgcn_alignbit(x[j + 1], x[j], (u32)n);
    x[LPL - 1] = __builtin_amdgcn_alignbit(next, x[LPL - 1], (u32)n);
  }
  __device__ __forceinline__ void shl(int n) {
    const u32 prev = dpp_mov<DPP_WAVE_SHR1, 0xF, 0xF, true>(0, x[LPL - 1]);
    const u32 inv = 32u - (u32)n;
#pragma unroll
    for (int j = LPL - 1; j > 0; --j) x[j] = __builtin_amdgcn_alignbit(x[j], x[j - 1], inv);
    x[0] = __builtin_amdgcn_alignbit(x[0], prev, inv);
  }
  // by one whole word
  __device__ __forceinline__ void shr_word() {
    const u32 next = dpp_mov<DPP_WAVE_SHL1, 0xF, 0xF, true>(0, x[0]);
#pragma unroll
    for (int j = 0; j < LPL - 1; ++j) x[j] = x[j + 1];
    x[LPL - 1] = next;
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
  // ---- the pieces of add / sub apart, for a trip of the main loop that runs several of them side by side: the lane-local
  // chain (result words, carry / borrow out of the lane, "all ones" / "all zero" of the lane's words), the ripple between the
  // lanes from the two ballots, and the lane's own +1 / -1
  static __device__ __forceinline__ void add_local(u32 (&w)[LPL], const WaveInt& a, const WaveInt& b, u32& carry, u32& all) {
    u32 c = 0, al = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const u64 t = (u64)a.x[j] + b.x[j] + c;
      w[j] = (u32)t;
      c = (u32)(t >> 32);
      al &= w[j];
    }
    carry = c; all = al;
  }
  static __device__ __forceinline__ void sub_local(u32 (&w)[LPL], const WaveInt& a, const WaveInt& b, u32& borrow, u32& any) {
    u32 bo = 0, an = 0;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const u64 t = (u64)a.x[j] - b.x[j] - bo;
      w[j] = (u32)t;
      bo = (u32)(t >> 63);
      an |= w[j];
    }
    borrow = bo; any = an;
  }
  static __device__ __forceinline__ unsigned long long ripple(unsigned long long G, unsigned long long P) { return (P + (G << 1)) ^ P; }
  __device__ __forceinline__ void take_plus(const u32 (&w)[LPL], unsigned long long cin) {
    u32 k = (u32)((cin >> lane()) & 1ull);
#pragma unroll
    for (int j = 0; j < LPL; ++j) { x[j] = w[j] + k; k = (x[j] == 0u) ? k : 0u; }
  }
  __device__ __forceinline__ void take_minus(const u32 (&w)[LPL], unsigned long long cin) {
    u32 k = (u32)((cin >> lane()) & 1ull);
#pragma unroll
    for (int j = 0; j < LPL; ++j) { const u32 old = w[j]; x[j] = old - k; k = (old == 0u) ? k : 0u; }
  }
  // this += q * y for a one-word q: lane-local multiply-accumulate chains, the carry WORD of every lane added to the next
  // lane's first limb (the ripple of that addition through the ballots, as in add).  The sum must fit the 64 * LPL words.
  __device__ __forceinline__ void add_mul(const WaveInt& y, u32 q) {
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const u64 t = (u64)x[j] + (u64)q * y.x[j] + c;
      x[j] = (u32)t;
      c = (u32)(t >> 32);
    }
    WaveInt w;
#pragma unroll
    for (int j = 1; j < LPL; ++j) w.x[j] = 0;
    w.x[0] = dpp_mov<DPP_WAVE_SHR1, 0xF, 0xF, true>(0, c);                   // lane l <- lane l-1, lane 0 <- 0
    add(w);
  }
  // this > y
  __device__ __forceinline__ bool gt(const WaveInt& y) const {
    // per lane through one borrow chain (this lane's limbs as one number: borrow out = less, all-zero difference = equal; a
    // limb-by-limb select chain was thirty instructions with its wait states), the lanes' verdicts compared as 64-bit integers
    u32 b = 0, any = 0;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const u64 t = (u64)x[j] - y.x[j] - b;
      b = (u32)(t >> 63);
      any |= (u32)t;
    }
    return __ballot(b == 0 && any != 0) > __ballot(b != 0);
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
  // (A subtraction leaves an even number: all its trailing zero bits — two on average — go in the same trip as the
  // subtraction, one variable shift each for the halved and the doubled number, instead of a trip of the loop per bit; the
  // sequence of Kaliski's steps, and k, are the same.  A wavefront alone on its SIMD issues an instruction every five cycles
  // whatever it is: the trips saved were a third of this kernel's 2.1 ms at 4102 bits.)
  auto zeros = [](u32 low) -> int { return low ? __builtin_ctz(low) : 31; };     // (a zero low word: 31 now, the rest next trip)
  bool vzero = v.is_zero();                      // (v reaches zero only through v - u with v = u: looked at there)
  while (k < bound && !vzero) {
    const u32 ul = u.low(), vl = v.low();
    int n;
    if (!(ul & 1u)) { n = zeros(ul); u.shr(n); s.shl(n); }
    else if (!(vl & 1u)) { n = zeros(vl); v.shr(n); r.shl(n); }
    else {
      // both odd: u - v, v - u and r + s lane by lane SIDE BY SIDE, their six ballots, the three ripples — then the difference
      // that is not negative and the sum are taken.  (One after the other — compare, subtract, add — every one of them was a
      // round trip vector -> scalar -> vector of its own, and a wavefront alone on its SIMD has nothing to put into those.)
      u32 duv[LPL], dvu[LPL], sm[LPL], b1, z1, b2, z2, c3, a3;
      WI::sub_local(duv, u, v, b1, z1);
      WI::sub_local(dvu, v, u, b2, z2);
      WI::add_local(sm, r, s, c3, a3);
      const unsigned long long G1 = __ballot(b1 != 0), Z1 = __ballot(z1 == 0u), G2 = __ballot(b2 != 0), Z2 = __ballot(z2 == 0u);
      const unsigned long long G3 = __ballot(c3 != 0), P3 = __ballot(a3 == 0xFFFFFFFFu);
      const unsigned long long cin1 = WI::ripple(G1, Z1), cin2 = WI::ripple(G2, Z2), cin3 = WI::ripple(G3, P3);
      const bool u_less = (((G1 >> 63) | ((Z1 >> 63) & (cin1 >> 63))) & 1ull) != 0;        // the borrow out of the top lane
      const bool equal = !u_less && Z1 == ~0ull;
      if (!u_less && !equal) {                                                 // u > v
        u.take_minus(duv, cin1);
        r.take_plus(sm, cin3);
        n = zeros(u.low());                                                    // (u - v > 0: never all zero)
        u.shr(n); s.shl(n);
      } else {
        v.take_minus(dvu, cin2);
        s.take_plus(sm, cin3);
        n = zeros(v.low());
        if (equal) { vzero = true; n = 1; }                                    // v = u: the one halving step Kaliski's loop ends on
        v.shr(n); r.shl(n);
      }
    }
    k += n;
  }
  const bool ok = vzero && u.is_one();
  if (!m.gt(r)) r.sub(m);                        // r < 2M  ->  r mod M
  WI x = m;
  x.sub(r);                                      // a^-1 * 2^k mod M
  // k halvings modulo M, a word at a time: x <- (x + q M) / 2^32 with q = -x M^-1 mod 2^32 (the low word becomes zero; the
  // value stays below M), then the k mod 32 bits that are left the same way.  (Bit by bit this was a fifth of the kernel.)
  u32 minv = m.low();                            // Newton: M^-1 mod 2^32 (M odd)
#pragma unroll
  for (int i = 0; i < 4; ++i) minv *= 2u - m.low() * minv;
  const u32 nminv = 0u - minv;
  int left = k;
  for (; left >= 32; left -= 32) {
    x.add_mul(m, x.low() * nminv);
    x.shr_word();
  }
  if (left > 0) {
    x.add_mul(m, (x.low() * nminv) & ((1u << left) - 1u));
    x.shr(left);
  }
  if (!ok) x.set_small(0);
  x.store(A.out + e * A.limbs, A.limbs);
  if (WI::lane() == 0) A.status[e] = ok ? 0 : 1;
}

}  // namespace mx
