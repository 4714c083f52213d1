// Batched Jacobi symbol (g / N), N odd: the filter `sympy.jacobi_symbol(g, modulus) != 1` of the
// reference's biprimality test (distributed_keygen.py:1089), ~4x40 symbols per candidate modulus.
//
// Unlike the Montgomery kernels this is a gcd-type algorithm, so the layout is different: ONE THREAD
// PER SYMBOL, both operands in registers as NL radix-2^32 limbs (fully unrolled limb loops,
// compile-time indices only; the top limbs of the 257-word instance in LDS, see jacobi_reg_limbs).  The algorithm is the division-step ("divstep") form of the binary
// Euclidean algorithm in its all-positive variant, in batches of 30 steps:
//   * 30 divsteps are decided from the LOW 64 bits of (f, g) alone (f = N, g = the value): strip the
//     trailing zeros of g; when the step counter eta turns negative swap f and g; add the multiple w
//     of f that cancels the next 4-6 low bits of g.  The symbol's sign follows the two classical
//     rules on those low bits: dividing g by 2 an odd number of times flips it when f = 3, 5 (mod 8);
//     swapping flips it when f = g = 3 (mod 4).  Adding multiples of f to g changes nothing.
//     The 30 steps accumulate into a 2x2 matrix (u v; q r) with non-negative entries <= 2^30.
//   * the matrix is applied to the full operands: f' = (u f + v g) / 2^30, g' = (q f + r g) / 2^30
//     (exact) — four v_mad_u64_u32 and two v_alignbit per limb for 30 steps, instead of a
//     compare / subtract / shift pass over all limbs per step as in the plain binary algorithm:
//     ~5x fewer instructions at 2053 bits.  f and g never exceed N, so NL = limbs suffices.
//   * f = 1: the symbol is the accumulated sign; f = g > 1: gcd > 1, the symbol is 0.
// About 3 divsteps per operand bit are needed (203 batches at 2053 bits); the batch count is bounded,
// so no input can hang the GPU.  (The formulation follows the posdivsteps Jacobi routine of
// libsecp256k1's modinv64 module, re-derived here for 30-step batches and 32-bit limbs; the Python
// model it was validated with against sympy is in tests/test_host_logic.py.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mx_prio.hpp"

namespace mx {

struct JacobiArgs {
  const uint32_t* a;      // [count][limbs] device: numerators (< modulus of their group)
  const uint32_t* mods;   // [groups][limbs] device
  signed char* out;       // [count] device: -1, 0, +1
  long long count;        // symbols evaluated by this launch
  long long group_size;   // rows per modulus in a / out
  int limbs;
  // Range form: only the rows [first, first + per_group) of every group are evaluated (count =
  // groups * per_group); with `skip` given the launch has ceil(per_group / 64) workgroups per group and
  // groups whose skip[g] >= skip_threshold are left out entirely — the tail of the generator list is
  // only needed where the head did not already yield enough symbols equal to 1 (DK:1086).
  int first, per_group;
  const int* skip;
  int skip_threshold;
  int max_batches;        // divstep batches before the binary-algorithm safety net takes over
};

// Limbs are processed in chunks of JC; chunks above the highest limb that is non-zero in ANY lane of
// the wavefront (for f or g) are skipped with a wave-uniform branch.  Both operands shrink steadily,
// so on average about half of the chunks are live; `live` is refreshed every JREFRESH batches.
// Where the limbs of the two operands live: in registers — except for the 257-word instance (key_length 8192), whose
// 2 x 257 words per lane exceed the 512 registers of a lane.  Its limbs from JREG upwards live in LDS, one column per
// lane ([limb][lane]: conflict-free), 33 KB per wavefront; the operands shrink from the top, so after the first quarter
// of the batches those limbs are no longer touched (`live`).  Round 3 shipped this instance on 204 B / 1848 B of scratch.
// (The safety-net kernel keeps fewer limbs in registers: its subtract-with-borrow chains need more temporaries.)
template <int NL, bool FALLBACK = false> constexpr int jacobi_reg_limbs() { return NL <= 129 ? NL : FALLBACK ? 128 : 192; }
template <int NL, bool FALLBACK = false> constexpr size_t jacobi_lds_bytes() { return (size_t)2 * (NL - jacobi_reg_limbs<NL, FALLBACK>()) * 64 * 4; }

constexpr int JC = 8;
constexpr int JREFRESH = 4;
constexpr int JSTEPS = 30;      // divsteps per batch
constexpr signed char JACOBI_UNFINISHED = 2;   // marker in the output for jacobi_fallback_kernel

template <int NL>
__global__ void __launch_bounds__(64) jacobi_kernel(JacobiArgs A) {
  aux_wave_priority();
  constexpr int NCH = (NL + JC - 1) / JC;
  long long grp, row;
  bool valid;
  if (A.skip) {
    const int bpg = (A.per_group + 63) / 64;
    grp = blockIdx.x / bpg;
    if (A.skip[grp] >= A.skip_threshold) return;            // uniform for the workgroup
    const int k = (int)(blockIdx.x % bpg) * 64 + threadIdx.x;
    valid = k < A.per_group;
    row = grp * A.group_size + A.first + (valid ? k : A.per_group - 1);
  } else {
    const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
    valid = idx < A.count;
    const long long e = valid ? idx : A.count - 1;
    grp = e / A.per_group;
    row = grp * A.group_size + A.first + (e - grp * A.per_group);
  }
  const uint32_t* pa = A.a + row * A.limbs;
  const uint32_t* pn = A.mods + grp * A.limbs;
  constexpr int JREG = jacobi_reg_limbs<NL>();
  extern __shared__ uint32_t jlds[];
  uint32_t* const lf = jlds + threadIdx.x;                 // this lane's column of the limbs >= JREG of f, then of g
  uint32_t* const lg = lf + (NL - JREG) * 64;
  uint32_t f[JREG], g[JREG];
  // (the index is a compile-time constant wherever these are used: the limb loops are fully unrolled)
  auto F = [&](int j) -> uint32_t { return j < JREG ? f[j < JREG ? j : 0] : lf[(j - JREG) * 64]; };
  auto G = [&](int j) -> uint32_t { return j < JREG ? g[j < JREG ? j : 0] : lg[(j - JREG) * 64]; };
  auto setF = [&](int j, uint32_t v) { if (j < JREG) f[j < JREG ? j : 0] = v; else lf[(j - JREG) * 64] = v; };
  auto setG = [&](int j, uint32_t v) { if (j < JREG) g[j < JREG ? j : 0] = v; else lg[(j - JREG) * 64] = v; };
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    setG(j, j < A.limbs ? pa[j] : 0u);
    setF(j, j < A.limbs ? pn[j] : 0u);
  }
  int result = 0;
  bool done;
  {
    uint32_t gz = 0, fhi = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) gz |= G(j);
#pragma unroll
    for (int j = 1; j < NL; ++j) fhi |= F(j);
    const bool f_is_one = fhi == 0 && f[0] == 1u;
    // (x / 1) = 1; (0 / N) = 0 for N > 1; an even "modulus" is not a Jacobi symbol: reported as 0
    done = f_is_one || gz == 0 || !(f[0] & 1u);
    result = f_is_one ? 1 : 0;
  }
  int jac = 0;
  int eta = -1;
  int live = NCH;
  // ~3.1 divsteps per bit of the modulus are needed (measured over operand sizes and shapes); there is no
  // proven bound for the all-positive variant, so the batch loop stops at 4.5 per bit and whatever is
  // not finished by then (nothing, on all inputs tried) is completed by jacobi_fallback_kernel
  // (A.max_batches = (32 * limbs * 9 / 2) / 30 + 8, set by the launcher)
  const int max_batches = A.max_batches;
  for (int batch = 0; batch < max_batches; ++batch) {
    if (!__any(!done)) break;
    if ((batch % JREFRESH) == 0) {
      int top = 0;           // 1 + index of this lane's highest non-zero limb of f | g
#pragma unroll
      for (int j = 0; j < NL; ++j) top = (F(j) | G(j)) ? j + 1 : top;
      if (done) top = 0;
      for (int off = 32; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off));
      live = __builtin_amdgcn_readfirstlane((top + JC - 1) / JC);
    }
    // ---- 30 divsteps on the low 64 bits -> matrix (u v; q r), entries <= 2^30
    uint32_t u = 1, v = 0, q = 0, r = 1;
    {
      unsigned long long fl = (unsigned long long)f[0] | ((unsigned long long)(NL > 1 ? f[1] : 0u) << 32);
      unsigned long long gl = (unsigned long long)g[0] | ((unsigned long long)(NL > 1 ? g[1] : 0u) << 32);
      int i = done ? 0 : JSTEPS;
      while (i > 0) {
        const int zeros = __builtin_ctzll(gl | (~0ull << i));
        gl >>= zeros;
        u <<= zeros;
        v <<= zeros;
        eta -= zeros;
        i -= zeros;
        jac ^= (int)(zeros & (unsigned)((fl >> 1) ^ (fl >> 2)));
        if (i == 0) break;
        unsigned long long m;
        uint32_t w;
        if (eta < 0) {
          eta = -eta;
          { const unsigned long long t = fl; fl = gl; gl = t; }
          { const uint32_t t = u; u = q; q = t; }
          { const uint32_t t = v; v = r; r = t; }
          jac ^= (int)((fl & gl) >> 1);
          const int limit = (eta + 1) > i ? i : (eta + 1);
          m = (~0ull >> (64 - limit)) & 63ull;
          const uint32_t f32 = (uint32_t)fl;
          w = (uint32_t)((f32 * (uint32_t)gl * (f32 * f32 - 2u)) & (uint32_t)m);
        } else {
          const int limit = (eta + 1) > i ? i : (eta + 1);
          m = (~0ull >> (64 - limit)) & 15ull;
          const uint32_t f32 = (uint32_t)fl;
          const uint32_t t = f32 + (((f32 + 1u) & 4u) << 1);
          w = ((0u - t) * (uint32_t)gl) & (uint32_t)m;
        }
        gl += fl * w;
        q += u * w;
        r += v * w;
      }
    }
    // ---- f' = (u f + v g) >> 30, g' = (q f + r g) >> 30   (exact; identity for finished lanes: u = r = 1 << 0 ...)
    if (!done) {
      unsigned long long cf = 0, cg = 0;
      uint32_t pf = 0, pg = 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c < live) {
#pragma unroll
          for (int j = c * JC; j < (c + 1) * JC && j < NL; ++j) {
            const uint32_t fj = F(j), gj = G(j);
            const unsigned long long af = (unsigned long long)u * fj + (unsigned long long)v * gj + cf;
            const unsigned long long ag = (unsigned long long)q * fj + (unsigned long long)r * gj + cg;
            const uint32_t lowf = (uint32_t)af, lowg = (uint32_t)ag;
            cf = af >> 32;
            cg = ag >> 32;
            if (j > 0) {
              setF(j - 1, __builtin_amdgcn_alignbit(lowf, pf, JSTEPS));
              setG(j - 1, __builtin_amdgcn_alignbit(lowg, pg, JSTEPS));
            }
            pf = lowf;
            pg = lowg;
          }
          // the top limb of the last live chunk receives the final carries (everything above is zero in
          // every lane of the wavefront, and f', g' <= max(f, g) fit below it)
          if (c == live - 1) {
            const int jend = ((c + 1) * JC < NL ? (c + 1) * JC : NL) - 1;      // compile-time per chunk
            setF(jend, __builtin_amdgcn_alignbit((uint32_t)cf, pf, JSTEPS));
            setG(jend, __builtin_amdgcn_alignbit((uint32_t)cg, pg, JSTEPS));
          }
        }
      }
      // ---- finished?  f == 1: the symbol is the sign; f == g (> 1): common factor, symbol 0
      uint32_t fhi = 0, ghi = 0, diff = 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c < live) {
#pragma unroll
          for (int j = c * JC; j < (c + 1) * JC && j < NL; ++j) {
            const uint32_t fj = F(j), gj = G(j);
            if (j > 0) { fhi |= fj; ghi |= gj; }
            diff |= fj ^ gj;
          }
        }
      }
      // (g / 1) = 1 and (1 / f) = 1: either operand reaching 1 ends the computation (the divsteps keep
      // passing the 1 back and forth between f and g, so f alone would be seen only at some batch ends)
      if ((fhi == 0 && f[0] == 1u) || (ghi == 0 && g[0] == 1u)) { done = true; result = 1 - 2 * (jac & 1); }
      else if (diff == 0) { done = true; result = 0; }
    }
  }
  // not converged within the batch bound: marked for jacobi_fallback_kernel (below)
  if (valid) A.out[row] = (signed char)(done ? result : JACOBI_UNFINISHED);
}

// Safety net of the divstep kernel: the plain binary algorithm, from the original operands, for the
// symbols the divstep kernel marked JACOBI_UNFINISHED (none, on all inputs tried: there is no proven
// step bound for the all-positive divsteps, so the bound of jacobi_kernel is backed by an algorithm
// that has one).  Launched right after it with the same arguments; a wavefront without a marked
// symbol returns at once.
//     t <- 1;  while a != 0:
//         strip the z trailing zero bits of a;  if z odd and n mod 8 in {3,5}: t <- -t
//         if a < n: swap (a, n); if a = n = 3 (mod 4): t <- -t
//         a <- a - n
//     result t if n == 1 else 0          (every pass removes a bit: at most 64 * NL + 2 passes)
template <int NL>
__global__ void __launch_bounds__(64) jacobi_fallback_kernel(JacobiArgs A) {
  aux_wave_priority();
  long long grp, row;
  bool valid;
  if (A.skip) {
    const int bpg = (A.per_group + 63) / 64;
    grp = blockIdx.x / bpg;
    if (A.skip[grp] >= A.skip_threshold) return;
    const int k = (int)(blockIdx.x % bpg) * 64 + threadIdx.x;
    valid = k < A.per_group;
    row = grp * A.group_size + A.first + (valid ? k : A.per_group - 1);
  } else {
    const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
    valid = idx < A.count;
    const long long e = valid ? idx : A.count - 1;
    grp = e / A.per_group;
    row = grp * A.group_size + A.first + (e - grp * A.per_group);
  }
  const bool todo = valid && A.out[row] == JACOBI_UNFINISHED;
  if (!__any(todo)) return;
  const uint32_t* pa = A.a + row * A.limbs;
  const uint32_t* pn = A.mods + grp * A.limbs;
  constexpr int JREG = jacobi_reg_limbs<NL, true>();
  extern __shared__ uint32_t jlds[];
  uint32_t* const la = jlds + threadIdx.x;                 // limbs >= JREG of a, then of n (see jacobi_reg_limbs)
  uint32_t* const ln = la + (NL - JREG) * 64;
  uint32_t a[JREG], n[JREG];
  auto AA = [&](int j) -> uint32_t { return j < JREG ? a[j < JREG ? j : 0] : la[(j - JREG) * 64]; };
  auto NN = [&](int j) -> uint32_t { return j < JREG ? n[j < JREG ? j : 0] : ln[(j - JREG) * 64]; };
  auto setA = [&](int j, uint32_t v) { if (j < JREG) a[j < JREG ? j : 0] = v; else la[(j - JREG) * 64] = v; };
  auto setN = [&](int j, uint32_t v) { if (j < JREG) n[j < JREG ? j : 0] = v; else ln[(j - JREG) * 64] = v; };
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    setA(j, j < A.limbs ? pa[j] : 0u);
    setN(j, j < A.limbs ? pn[j] : 0u);
  }
  int t = 1;
  for (int pass = 0; pass < 64 * NL + 2; ++pass) {
    uint32_t nz = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) nz |= AA(j);
    const bool active = todo && nz != 0;
    if (!__any(active)) break;
    if (!active) continue;
    while (a[0] == 0) {            // a != 0, so this terminates; 32 zero bits: even count, no sign change
#pragma unroll
      for (int j = 0; j < NL - 1; ++j) setA(j, AA(j + 1));
      setA(NL - 1, 0);
    }
    const int z = __builtin_ctz(a[0]);
    if (z) {
#pragma unroll
      for (int j = 0; j < NL; ++j) setA(j, (j + 1 < NL) ? __builtin_amdgcn_alignbit(AA(j + 1), AA(j), z) : (AA(j) >> z));
      const uint32_t n8 = n[0] & 7u;
      if ((z & 1) && (n8 == 3u || n8 == 5u)) t = -t;
    }
    unsigned int borrow = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) (void)__builtin_subc(AA(j), NN(j), borrow, &borrow);
    unsigned int b = 0;
    if (borrow) {                  // (a, n) <- (n - a, a), quadratic reciprocity for the swap
      if ((a[0] & 3u) == 3u && (n[0] & 3u) == 3u) t = -t;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const uint32_t x = AA(j);
        setA(j, __builtin_subc(NN(j), x, b, &b));
        setN(j, x);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NL; ++j) setA(j, __builtin_subc(AA(j), NN(j), b, &b));
    }
  }
  uint32_t hi = 0, az = 0;
#pragma unroll
  for (int j = 1; j < NL; ++j) hi |= NN(j);
#pragma unroll
  for (int j = 0; j < NL; ++j) az |= AA(j);
  if (todo) A.out[row] = (signed char)((az == 0 && hi == 0 && n[0] == 1u) ? t : 0);
}

}  // namespace mx
