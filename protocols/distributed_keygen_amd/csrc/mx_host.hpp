// Host-side helpers of the C ABI: tiny fixed-purpose big-integer routines (only what is needed
// to prepare per-modulus constants), geometry selection, error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include <string>
#include <cstdlib>
#include <atomic>
#include "../../../include/mxpaillier.h"

namespace mxh {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int LIMB_BITS = 29;   // W
constexpr int LIMBS_PER_LANE = 9;   // L of the narrow geometry (more lanes per element)
constexpr int LIMBS_PER_LANE_WIDE = 18;   // L of the wide geometry (fewer, busier lanes)
constexpr int LIMBS_PER_LANE_LAT = 3;     // L of the latency geometry (many lanes per element: launches that leave SIMDs idle)
// limbs_per_lane value of the BIPARTITE latency form of the generic modexp (mx_bimont.hpp): 3 limbs per lane, every product
// split over two wavefronts (half the multiplier's limbs each) — "3 x 2"
constexpr int LIMBS_PER_LANE_BI = 6;

// Limbs per lane (3 latency / 9 narrow / 18 wide) and the other launch-shape choices are per-call ARGUMENTS of the
// entry points; the entry points without such an argument leave the choice to the library.  The library reads NO
// environment variable and keeps no tuning state that a production caller can set: what remains process-wide are the
// developer knobs below (explicit mx_debug_knob calls, for A/B runs and for tests that must reach a fallback path), held
// in atomics so that a knob flipped by one thread while another launches is a defined read of the old or the new value.
struct Knob {
  std::atomic<int> v{0};
  operator int() const { return v.load(std::memory_order_relaxed); }
  Knob& operator=(int x) { v.store(x, std::memory_order_relaxed); return *this; }
};
extern Knob g_knob_n2_segments;        // MX_KNOB_N2_SEGMENTS: 0 = automatic
extern Knob g_knob_n2_timeslice;       // MX_KNOB_N2_TIMESLICE: 0 = automatic, 1 = never, 2 = always (two-wavefront launches)
extern Knob g_knob_n2_friendly_1w;     // MX_KNOB_N2_FRIENDLY_1W: 0 = friendly-modulus instances of the one-wavefront wide kernel where they exist, 1 = never
extern Knob g_knob_n2_split;           // MX_KNOB_N2_SPLIT: 0 = mx_nsquare_launch_split reports a split where it pays, 1 = never, 2 = whenever a split exists
extern Knob g_knob_generic_latency;    // MX_KNOB_GENERIC_LATENCY: 0 = the automatic choice may take the 3-limb instances of the generic kernel, 1 = never
extern Knob g_knob_jacobi_max_batches; // MX_KNOB_JACOBI_MAX_BATCHES: 0 = the kernel's own bound, v = at most v - 1 batches
extern Knob g_knob_lat_lanes;          // MX_KNOB_LAT_LANES: 0 = the smallest group that holds the number, v = at least v lanes per element for the 3-limb latency forms of the generic kernel
extern Knob g_knob_n2_bipair;          // MX_KNOB_N2_BIPAIR: 0 = small launches of the pair kernel take the five-wavefront latency form where it exists, 1 = never
extern Knob g_knob_bi_pivot;           // MX_KNOB_BI_PIVOT: 0 = the library's pivot of the bipartite form, v = v multiplier limbs on wavefront L (rounded down to a multiple of 3)

struct Geometry {
  int K = 0;      // lanes per element
  int L = LIMBS_PER_LANE;
  int W = LIMB_BITS;
  int nblk = 0;   // Montgomery R = 2^(W*L*nblk)
  int bi = 0;     // 1: the bipartite form (two wavefronts per group of elements); Pd = L * nblk data positions
  int h_lo = 0;   // bipartite: the pivot (multiplier limbs [0, h_lo) on wavefront L)
};

// R = 2^(W*L*nblk) must be >= 16 N (lazy reduction bound, mx_mont.hpp), nblk <= K.
// Largest modulus the engine takes (any geometry): the narrow geometry's R = 2^(W*9*64) >= 16 N.
constexpr int MAX_MOD_BITS = LIMB_BITS * LIMBS_PER_LANE * 64 - 4;
inline bool choose_geometry(int mod_bits, Geometry& g, int limbs_per_lane = LIMBS_PER_LANE) {
  if (mod_bits < 2 || mod_bits > MAX_MOD_BITS) return false;
  g.bi = 0; g.h_lo = 0;
  if (limbs_per_lane == LIMBS_PER_LANE_BI) {
    // tools/bimont_model.py Geometry: Pd = 3 * nblk data positions with W * Pd >= bits + 35, wavefront H needs Pd / 3 + 2 lanes
    g.L = LIMBS_PER_LANE_LAT;
    const int per = g.W * g.L;
    g.nblk = (mod_bits + 35 + per - 1) / per;
    int k = 4;
    while (k < g.nblk + 2 || k < (int)g_knob_lat_lanes) k <<= 1;
    if (k > 64) return false;
    g.K = k;
    g.bi = 1;
    const int steps = g.L * g.nblk + g.L;                  // multiplier limbs Pd + 2 .. 0
    // the pivot: a step of wavefront H costs ~1.15 x a step of wavefront L (32 against 28 ns at key_length 2048) and H carries
    // the end of the product, so L takes a little more than half of the steps (tools/bi_pivot_sweep.py,
    // profiles/r05_bi_pivot_sweep.txt: best at 24 of 42 steps at key_length 1024, 39 of 75 at 2048; groups of 64 lanes — one
    // element per wavefront pair, moduli beyond 2600 bits — at 54 of 147 at key_length 4096)
    g.h_lo = g.K == 64 ? g.L * ((37 * steps + 150) / (100 * g.L)) : g.L * ((46 * steps + 600) / (100 * g.L));
    if (g_knob_bi_pivot > 0) g.h_lo = g.L * ((int)g_knob_bi_pivot / g.L);
    if (g.h_lo > steps - g.L) g.h_lo = steps - g.L;        // (tiny moduli: at least one block for wavefront H)
    if (g.h_lo < g.L) g.h_lo = g.L;
    // the factor that leaves the domain, 2^(W * (Pd - hL)), is a Montgomery operand of the epilogue: it must be below N
    const int lo_min = g.L * ((g.L * g.nblk - (mod_bits - 2) / g.W + g.L - 1) / g.L);
    if (g.h_lo < lo_min) g.h_lo = lo_min;
    return true;
  }
  g.L = limbs_per_lane;
  // 3 limbs per lane (the pair kernel's latency instances) reduce modulo a multiple of N that is LIMB_BITS bits
  // longer (mx_mont.hpp: F_FRIENDLY) and keep two more bits of head room for the lazy bound
  int need = mod_bits + 4 + (limbs_per_lane == 3 ? LIMB_BITS + 2 : 0);
  int per_blk = g.W * g.L;
  g.nblk = (need + per_blk - 1) / per_blk;
  int k = 1;
  while (k < g.nblk || (limbs_per_lane == 3 && k < (int)g_knob_lat_lanes)) k <<= 1;
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

// Left-to-right sliding-window schedule of a non-zero exponent: step k squares `squarings` times and
// then multiplies by the odd power with index `index1 - 1` (index1 == 0: trailing squarings, no
// multiplication).  Step 0 only loads its odd power.
struct SlidingOp {
  int squarings;
  u32 index1;
};
inline std::vector<SlidingOp> sliding_schedule(const u32* e, int limbs, int w) {
  std::vector<SlidingOp> ops;
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
    ops.push_back(SlidingOp{ops.empty() ? 0 : pending + l, idx1});
    pending = 0;
    i -= l;
  }
  if (pending) ops.push_back(SlidingOp{pending, 0u});
  return ops;
}

// The schedule as the words powmod_kernel<..., SLIDING> reads: (squarings << 16) | index1.  A run of
// more than 65535 squarings (an exponent with that many consecutive zero bits) is split into
// squaring-only words so that the 16-bit field never overflows.
inline std::vector<u32> pack_sliding_ops(const std::vector<SlidingOp>& ops) {
  std::vector<u32> out;
  for (size_t k = 0; k < ops.size(); ++k) {
    int nsq = ops[k].squarings;
    while (nsq > 0xFFFF) { out.push_back(0xFFFFu << 16); nsq -= 0xFFFF; }
    out.push_back(((u32)nsq << 16) | ops[k].index1);
  }
  return out;
}

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern thread_local hipError_t g_last_hip;
inline bool hip_ok(hipError_t e) {
  if (e != hipSuccess) { g_last_hip = e; return false; }
  return true;
}
#define MX_HIP(call) do { if (!mxh::hip_ok(call)) return MX_ERR_HIP; } while (0)

}  // namespace mxh
