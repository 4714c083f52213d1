// Host-side helpers of the C ABI: tiny fixed-purpose big-integer routines (only what is needed
// to prepare per-modulus constants), geometry selection, error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include <string>
#include <cstdlib>
#include "../../../include/mxpaillier.h"

namespace mxh {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int LIMB_BITS = 29;   // W
constexpr int LIMBS_PER_LANE = 9;   // L of the narrow geometry (more lanes per element)
constexpr int LIMBS_PER_LANE_WIDE = 18;   // L of the wide geometry (fewer, busier lanes)

struct Geometry {
  int K = 0;      // lanes per element
  int L = LIMBS_PER_LANE;
  int W = LIMB_BITS;
  int nblk = 0;   // Montgomery R = 2^(W*L*nblk)
};

// R = 2^(W*L*nblk) must be >= 16 N (lazy reduction bound, mx_mont.hpp), nblk <= K.
inline bool choose_geometry(int mod_bits, Geometry& g, int limbs_per_lane = LIMBS_PER_LANE) {
  if (mod_bits < 2) return false;
  g.L = limbs_per_lane;
  int need = mod_bits + 4;
  int per_blk = g.W * g.L;
  g.nblk = (need + per_blk - 1) / per_blk;
  int k = 1;
  while (k < g.nblk) k <<= 1;
  if (k > 64) return false;
  g.K = k;
  return true;
}

inline int bit_length(const u32* x, int limbs) {
  for (int i = limbs - 1; i >= 0; --i)
    if (x[i]) return 32 * i + (32 - __builtin_clz(x[i]));
  return 0;
}

inline bool geq(const std::vector<u32>& a, const u32* b, int limbs) {   // a has limbs+1 words
  if (a[limbs]) return true;
  for (int i = limbs - 1; i >= 0; --i) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}

// out[0..limbs) = 2^m mod n  (n odd, >= 3, m >= bit_length(n) - 1)
inline void two_pow_mod(u32* out, const u32* n, int limbs, int m) {
  int bits = bit_length(n, limbs);
  std::vector<u32> x(limbs + 1, 0u);
  x[(bits - 1) / 32] = 1u << ((bits - 1) % 32);   // 2^(bits-1) < n
  for (int e = bits - 1; e < m; ++e) {
    u32 carry = 0;
    for (int i = 0; i <= limbs; ++i) {
      u32 v = x[i];
      x[i] = (v << 1) | carry;
      carry = v >> 31;
    }
    if (geq(x, n, limbs)) {
      u64 borrow = 0;
      for (int i = 0; i < limbs; ++i) {
        u64 d = (u64)x[i] - n[i] - borrow;
        x[i] = (u32)d;
        borrow = (d >> 63) & 1;
      }
      x[limbs] -= (u32)borrow;
    }
  }
  for (int i = 0; i < limbs; ++i) out[i] = x[i];
}

// out[0..limbs) = 2^m mod n for any m >= 0 (n odd, >= 3)
inline void pow2_mod(u32* out, const u32* n, int limbs, int m) {
  int bits = bit_length(n, limbs);
  if (m < bits - 1) {
    for (int i = 0; i < limbs; ++i) out[i] = 0;
    out[m / 32] = 1u << (m % 32);
    return;
  }
  two_pow_mod(out, n, limbs, m);
}

// q[0..la) , r[0..lb) = divmod(a[0..la), b[0..lb)), b != 0; binary long division (host constants only)
inline void divmod_words(u32* q, u32* r, const u32* a, int la, const u32* b, int lb) {
  std::vector<u32> rem(lb + 1, 0u);
  for (int i = 0; i < la; ++i) q[i] = 0;
  for (int bit = bit_length(a, la) - 1; bit >= 0; --bit) {
    u32 carry = (a[bit >> 5] >> (bit & 31)) & 1u;
    for (int i = 0; i <= lb; ++i) {
      u32 v = rem[i];
      rem[i] = (v << 1) | carry;
      carry = v >> 31;
    }
    if (geq(rem, b, lb)) {
      u64 borrow = 0;
      for (int i = 0; i < lb; ++i) {
        u64 d = (u64)rem[i] - b[i] - borrow;
        rem[i] = (u32)d;
        borrow = (d >> 63) & 1;
      }
      rem[lb] -= (u32)borrow;
      q[bit >> 5] |= 1u << (bit & 31);
    }
  }
  for (int i = 0; i < lb; ++i) r[i] = rem[i];
}

// x*y for small host-side products (schoolbook); out has la+lb words
inline void mul_words(u32* out, const u32* a, int la, const u32* b, int lb) {
  for (int i = 0; i < la + lb; ++i) out[i] = 0;
  for (int i = 0; i < la; ++i) {
    u64 carry = 0;
    for (int j = 0; j < lb; ++j) {
      u64 t = (u64)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (u32)t;
      carry = t >> 32;
    }
    out[i + lb] = (u32)carry;
  }
}

inline int fixed_window(int exp_bits) {
  // minimise ceil(bits/w) multiplications + 2^w - 2 table products
  int best = 1;
  long bestc = -1;
  for (int w = 1; w <= 7; ++w) {
    long c = (exp_bits + w - 1) / w + (1L << w) - 2;
    if (bestc < 0 || c < bestc) { bestc = c; best = w; }
  }
  return best;
}

inline int sliding_window(int exp_bits) {
  // minimise 2^(w-1) table products + bits/(w+1) expected multiplications
  int best = 1;
  double bestc = -1;
  for (int w = 1; w <= 8; ++w) {
    double c = (double)(1L << (w - 1)) + (double)exp_bits / (w + 1);
    if (bestc < 0 || c < bestc) { bestc = c; best = w; }
  }
  return best;
}

// Left-to-right sliding-window schedule of a non-zero exponent: ops[k] = (squarings << 16) |
// (index of the odd power + 1), index + 1 == 0 for the trailing squarings.  ops[0] only loads.
inline std::vector<u32> sliding_schedule(const u32* e, int limbs, int w) {
  std::vector<u32> ops;
  int bits = bit_length(e, limbs);
  auto bit = [&](int i) { return (e[i >> 5] >> (i & 31)) & 1u; };
  int i = bits - 1, pending = 0;
  while (i >= 0) {
    if (!bit(i)) { ++pending; --i; continue; }
    int l = (i + 1 < w) ? i + 1 : w;
    while (!bit(i - l + 1)) --l;
    u32 val = 0;
    for (int k = 0; k < l; ++k) val = (val << 1) | bit(i - k);
    u32 idx1 = (val - 1) / 2 + 1;
    ops.push_back(ops.empty() ? idx1 : (((u32)(pending + l) << 16) | idx1));
    pending = 0;
    i -= l;
  }
  if (pending) ops.push_back((u32)pending << 16);
  return ops;
}

// Which limbs-per-lane to run a modexp batch with.  The wide geometry spends a larger share of its
// instructions on multiply-accumulates (the per-limb bookkeeping is amortised over 2L MACs) but
// puts half as many lanes on the machine; it pays once the batch still fills every SIMD with
// at least two wavefronts (1024 SIMDs x 64 lanes) — or when the caller keeps several batches in
// flight on different streams, which the library cannot see: mx_set_limbs_per_lane(9|18) or the
// environment variable MX_LIMBS_PER_LANE override the automatic choice.
extern int g_limbs_per_lane;   // 0 = automatic; set by mx_set_limbs_per_lane
inline int pick_limbs_per_lane(int mod_bits, int64_t batch) {
  if (g_limbs_per_lane == LIMBS_PER_LANE || g_limbs_per_lane == LIMBS_PER_LANE_WIDE) return g_limbs_per_lane;
  if (const char* e = getenv("MX_LIMBS_PER_LANE")) {
    int v = atoi(e);
    if (v == LIMBS_PER_LANE || v == LIMBS_PER_LANE_WIDE) return v;
  }
  Geometry wide;
  if (!choose_geometry(mod_bits, wide, LIMBS_PER_LANE_WIDE)) return LIMBS_PER_LANE;
  int64_t waves = (batch * wide.K + 63) / 64;
  return waves >= 2 * 1024 ? LIMBS_PER_LANE_WIDE : LIMBS_PER_LANE;
}

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern thread_local hipError_t g_last_hip;
inline bool hip_ok(hipError_t e) {
  if (e != hipSuccess) { g_last_hip = e; return false; }
  return true;
}
#define MX_HIP(call) do { if (!mxh::hip_ok(call)) return MX_ERR_HIP; } while (0)

}  // namespace mxh
