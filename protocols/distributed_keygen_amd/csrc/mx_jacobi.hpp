// Batched Jacobi symbol (g / N), N odd: the filter `sympy.jacobi_symbol(g, modulus) != 1` of the
// reference's biprimality test (distributed_keygen.py:1089), ~4x40 symbols per candidate modulus.
//
// Unlike the Montgomery kernels this is a subtract-and-shift algorithm with no multiplications,
// so the layout is different: ONE THREAD PER SYMBOL, both operands in registers as NL radix-2^32
// limbs (fully unrolled limb loops, compile-time indices only).  Binary algorithm:
//     a <- g mod N handled by the caller (g < N as the reference guarantees, UT:361);  t <- 1
//     while a != 0:
//         strip the z trailing zero bits of a;  if z odd and N mod 8 in {3,5}: t <- -t
//         if a < N: swap (a, N); if a = N = 3 (mod 4): t <- -t
//         a <- a - N
//     result t if N == 1 else 0
// Compare (one borrow chain, nothing stored) then subtract in the right direction, in place;
// lanes of a wave run different trip counts (the loop is ~1.4 iterations per bit), so the kernel is
// launched with consecutive threads on consecutive symbols of the SAME modulus size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
};

// Limbs are processed in chunks of JC; chunks above the highest limb that is non-zero in ANY lane of
// the wavefront (for a or n) are skipped with a wave-uniform branch.  Both operands shrink steadily,
// so on average about half of the chunks are live; `live` is refreshed every JREFRESH passes.
constexpr int JC = 8;
constexpr int JREFRESH = 16;

template <int NL>
__global__ void __launch_bounds__(64) jacobi_kernel(JacobiArgs A) {
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
  uint32_t a[NL], n[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    a[j] = j < A.limbs ? pa[j] : 0u;
    n[j] = j < A.limbs ? pn[j] : 0u;
  }
  int t = 1;
  int live = NCH;            // wave-uniform number of live chunks
  // Every pass removes at least one bit from a or n, so 64*NL passes always suffice; the bound
  // makes the kernel terminate on ANY input (an even "modulus" would otherwise never finish).
  bool done = false;
  for (int pass = 0; pass < 64 * NL + 2; ++pass) {
    if ((pass % JREFRESH) == 0) {
      int top = 0;           // 1 + index of this lane's highest non-zero limb of a | n
#pragma unroll
      for (int j = 0; j < NL; ++j) top = (a[j] | n[j]) ? j + 1 : top;
      // wave maximum, made uniform
      for (int off = 32; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off));
      live = __builtin_amdgcn_readfirstlane((top + JC - 1) / JC);
    }
    uint32_t nz = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < live) {
#pragma unroll
        for (int j = c * JC; j < (c + 1) * JC && j < NL; ++j) nz |= a[j];
      }
    }
    // Lanes whose a has reached 0 idle until the whole wavefront is finished: the exit must be
    // wave-uniform because the refresh above reduces over all 64 lanes.
    const bool active = nz != 0;
    if (!__any(active)) { done = true; break; }
    if (!active) continue;
    // ---- strip trailing zeros (whole limbs first, then bits)
    while (a[0] == 0) {            // a != 0, so this terminates; 32 zero bits: even count, no sign change
#pragma unroll
      for (int j = 0; j < NL - 1; ++j) a[j] = a[j + 1];
      a[NL - 1] = 0;
    }
    const int z = __builtin_ctz(a[0]);
    if (z) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c < live) {
#pragma unroll
          for (int j = c * JC; j < (c + 1) * JC && j < NL; ++j)
            a[j] = (j + 1 < NL) ? __builtin_amdgcn_alignbit(a[j + 1], a[j], z) : (a[j] >> z);
        }
      }
      const uint32_t n8 = n[0] & 7u;
      if ((z & 1) && (n8 == 3u || n8 == 5u)) t = -t;
    }
    // ---- a < n ?  (borrow chain of a - n: v_sub_co / v_subb_co, nothing stored)
    unsigned int borrow = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < live) {
#pragma unroll
        for (int j = c * JC; j < (c + 1) * JC && j < NL; ++j) (void)__builtin_subc(a[j], n[j], borrow, &borrow);
      }
    }
    if (borrow) {
      // (a, n) <- (n - a, a), quadratic reciprocity for the swap
      if ((a[0] & 3u) == 3u && (n[0] & 3u) == 3u) t = -t;
      unsigned int b = 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c < live) {
#pragma unroll
          for (int j = c * JC; j < (c + 1) * JC && j < NL; ++j) {
            const uint32_t x = a[j];
            a[j] = __builtin_subc(n[j], x, b, &b);
            n[j] = x;
          }
        }
      }
    } else {
      unsigned int b = 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c < live) {
#pragma unroll
          for (int j = c * JC; j < (c + 1) * JC && j < NL; ++j) a[j] = __builtin_subc(a[j], n[j], b, &b);
        }
      }
    }
  }
  uint32_t hi = 0, az = 0;
#pragma unroll
  for (int j = 1; j < NL; ++j) hi |= n[j];
#pragma unroll
  for (int j = 0; j < NL; ++j) az |= a[j];
  done = done && (az == 0);
  const bool n_is_one = done && (hi == 0) && (n[0] == 1u);
  if (valid) A.out[row] = (signed char)(n_is_one ? t : 0);
}

}  // namespace mx
