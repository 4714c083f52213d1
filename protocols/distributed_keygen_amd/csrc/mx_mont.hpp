// Lane-distributed, reduced-radix ("lazy carry") Montgomery arithmetic for gfx950.
//
// Number format
//   A big integer is held by a GROUP of K consecutive lanes of one wavefront; lane p owns the
//   L limbs [p*L, (p+1)*L) of radix 2^W (W = 29): capacity S = K*L limbs.  Operands are "almost
//   normalised" (every limb < 2^W + 2^7); accumulators are 64-bit columns that are NOT carried
//   while a multiplication runs.  This is what makes the inner loop one instruction per
//   multiply-accumulate: on gfx950 v_mad_u64_u32 (32x32 + 64 -> 64) issues at the rate of a plain
//   add, but it has no carry-in, so a radix-2^32 carry chain would cost a second instruction per
//   MAC (profiles/r01_ubench_valu_rates.txt).  With W = 29 a column can absorb 2*L products
//   (< 2^58 each) for L <= 18 before it is handed to the neighbouring lane, which is exactly how
//   long a column lives in one lane — no carry handling in the loop at all.
//
// Montgomery multiplication (word-serial CIOS, R = 2^(W*L*nblk) >= 16 N so no conditional
// subtraction is ever needed between operations: inputs < 2N give outputs < 2N):
//   for every limb b_i of b (fetched from an LDS copy of b, same address for the whole group):
//     t += a * b_i                     L  x v_mad_u64_u32 per lane
//     q  = (t_0 * n0inv) mod 2^W       computed by the group's lane 0, broadcast by DPP
//     t += N * q                       L  x v_mad_u64_u32 per lane
//     t >>= W                          lane-local register renaming; one W-bit word crosses to
//                                      the lower neighbour by DPP, the rest of column 0 is a
//                                      local 64-bit add into column 1
//
// Everything in this header is device code shared by the modexp, share-combine and biprimality
// verdict kernels.  It replaces the reference's calls into gmpy2/CPython big-int pow
// (reference: distributed_keygen.py:1094,1097; paillier_shared_key.py:92,115-125).
#pragma once
#include "mx_dev.hpp"
#include "mx_lanes.hpp"
#include <utility>

namespace mx {

typedef long long i64;

// WG_SYNC: how the group's LDS scratch is ordered between its writers and readers.  true (default): a
// workgroup barrier — the kernels whose workgroup is ONE wavefront (for them the barrier costs nothing
// and the scratch of every group belongs to that wavefront).  false: a wavefront-level fence only — for
// kernels whose workgroup holds several wavefronts that run DIFFERENT code and own disjoint scratch
// (mx_powmod_n2_split.hpp): the LDS operations of one wavefront execute in order, so a fence that stops
// the compiler from reordering them is all a wavefront needs to read what its own lanes wrote.
template <int K, int L, int W, bool USE_DPP = true, bool WG_SYNC = true>
struct Mont {
  static_assert(L >= 2, "L >= 2");
  static_assert(W >= 16 && W <= 30, "radix");
  using LN = Lanes<K, USE_DPP>;
  static constexpr u32 MASK = (1u << W) - 1u;
  static constexpr int LIMBS = L;
  static constexpr int S = K * L;              // capacity in limbs
  // per-group scratch (32-bit words): multiplier b, second multiplier d.  MX_DEV_LDS_PAD_WORDS (a build flag for A/B runs,
  // default 0) lengthens the stride between the groups of a wavefront: with 152 words (18-limb groups of 4 lanes) the 16
  // groups' broadcast reads of a multiplier limb fall on 4 of the 32 banks; an odd stride spreads them over 16
  // (profiles/r05_lds_stride_ab.txt: what that is worth)
  static constexpr int LDS_WORDS = 2 * S + 8 + MX_DEV_LDS_PAD_WORDS;
  static constexpr int LDS_D = S + 4;          // offset of the second multiplier

  u32 n[L];      // modulus slice (exact W-bit limbs)
  u32 nf[L];     // F_FRIENDLY: slice of N~ + 1, N~ = uf * N = -1 mod 2^W (set_friendly; unused and dropped otherwise)
  u32 uf;        // F_FRIENDLY: uf = -N^-1 mod 2^W (= n0inv), kept apart for the quotient bookkeeping of the pair kernel
  u32 n0inv;     // -N^-1 mod 2^W
  u32 keep_next, keep_prev;
  u32 maskv;     // MASK held in a VGPR (a DPP-modified VOP2 cannot take a literal operand)
  u32 onev;      // 1 held in a VGPR the compiler cannot see through (see limb_step)
  u32 next_mask; // keep_next & MASK
  int p;         // lane position in the group
  int nblk;      // R = 2^(W*L*nblk)
  u32* lds;      // this group's LDS scratch, LDS_WORDS words

  static __device__ __forceinline__ void sync() {
    if constexpr (WG_SYNC) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }

  // ------------------------------------------------------------------ setup / conversion
  __device__ __forceinline__ void init(u32* lds_group, int nblk_) {
    p = LN::pos();
    keep_next = LN::keep_next_mask();
    keep_prev = LN::keep_prev_mask();
    maskv = MASK;
    asm volatile("" : "+v"(maskv));   // opaque: keeps the constant in a VGPR
    onev = 1u;
    asm volatile("" : "+v"(onev));
    next_mask = keep_next & maskv;
    lds = lds_group;
    nblk = nblk_;
  }

  // Cooperative copy of `nwords` 32-bit words (global memory, contiguous per element: the
  // group reads one contiguous span, lane-consecutive) into the group's LDS scratch, zero padded.
  __device__ __forceinline__ void stage_words(const u32* __restrict__ src, int nwords) {
    sync();
    for (int k = p; k < LDS_WORDS; k += K) lds[k] = (k < nwords) ? src[k] : 0u;
    sync();
  }

  // radix-2^32 words in LDS -> this lane's L radix-2^W limbs
  __device__ __forceinline__ void limbs_from_lds(u32 (&dst)[L]) const {
#pragma unroll
    for (int j = 0; j < L; ++j) {
      int bit = W * (p * L + j);
      int w = bit >> 5, off = bit & 31;
      u64 v = (u64)lds[w] | ((u64)lds[w + 1] << 32);
      dst[j] = (u32)(v >> off) & MASK;
    }
  }

  __device__ __forceinline__ void load(u32 (&dst)[L], const u32* __restrict__ src, int nwords) {
    stage_words(src, nwords);
    limbs_from_lds(dst);
  }

  // exact W-bit limbs (this lane's slice) -> radix-2^32 words in global memory
  __device__ __forceinline__ void store(u32* __restrict__ dst, int nwords, const u32 (&x)[L], bool valid) {
    sync();
#pragma unroll
    for (int j = 0; j < L; ++j) lds[p * L + j] = x[j];
    if (p == 0) { lds[S] = 0; lds[S + 1] = 0; lds[S + 2] = 0; lds[S + 3] = 0; }
    sync();
    for (int k = p; k < nwords; k += K) {
      int bit = 32 * k;
      int g = bit / W, off = bit - g * W;
      u32 out = 0;
      if (g < S) {
        u64 v = (u64)lds[g] >> off;
        v |= (u64)lds[g + 1] << (W - off);
        if (2 * W - off < 32) v |= (u64)lds[g + 2] << (2 * W - off);
        out = (u32)v;
      }
      if (valid) dst[k] = out;
    }
    sync();
  }

  // n[] must be loaded; computes n0inv = -N^-1 mod 2^W from the group's limb 0
  __device__ __forceinline__ void setup_modulus() {
    u32 n0 = LN::bcast0(n[0]);
    u32 x = n0;                     // Newton: x <- x (2 - n0 x) doubles the correct low bits (3 -> 48)
#pragma unroll
    for (int i = 0; i < 4; ++i) x *= 2u - n0 * x;
    n0inv = (0u - x) & MASK;
  }

  // F_FRIENDLY: nf[] must hold this lane's limbs of N~ + 1 (loaded by the caller from the host's constants)
  __device__ __forceinline__ void setup_friendly() { uf = n0inv; }
  // ... or computed here from the modulus alone (per-candidate moduli have no host constants): N~ + 1 = u N + 1 with
  // u = -N^-1 mod 2^W — one multiply-add per limb and a carry sweep; the geometry leaves the W bits of room.
  __device__ __forceinline__ void compute_friendly() {
    uf = n0inv;
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = (u64)n[j] * uf;
    if (p == 0) t[0] += 1;
    normalize_full(nf, t);
  }

  // ------------------------------------------------------------------ carry handling
  // 64-bit columns -> almost-normalised limbs (limb 1 may exceed 2^W by < 2^7).  One local carry
  // sweep, one neighbour exchange, one two-limb fix-up; value preserved exactly.
  __device__ __forceinline__ void normalize_weak(u32 (&r)[L], u64 (&t)[L]) const {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      u64 v = t[j] + c;
      r[j] = (u32)v & MASK;
      c = v >> W;
    }
    if constexpr (K > 1) {
      u32 clo = LN::from_prev((u32)c, keep_prev);
      u32 chi = LN::from_prev((u32)(c >> 32), keep_prev);
      u64 v = (u64)r[0] + ((u64)clo | ((u64)chi << 32));
      r[0] = (u32)v & MASK;
      r[1] += (u32)(v >> W);
    }
  }

  // 64-bit columns -> exact W-bit limbs; returns (in the group's top lane) the carry that left
  // the S-limb number.  Wave-uniform loop: runs until no lane of the wave has a pending carry.
  __device__ __forceinline__ u64 normalize_full(u32 (&r)[L], u64 (&t)[L]) const {
    u64 c = 0, top = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      u64 v = t[j] + c;
      r[j] = (u32)v & MASK;
      c = v >> W;
    }
    while (true) {
      if (p == K - 1) top += c;
      u64 cin = 0;
      if constexpr (K > 1) {
        u32 clo = LN::from_prev((u32)c, keep_prev);
        u32 chi = LN::from_prev((u32)(c >> 32), keep_prev);
        cin = (u64)clo | ((u64)chi << 32);
      }
      if (!__any(cin != 0)) break;
      c = cin;
#pragma unroll
      for (int j = 0; j < L; ++j) {
        u64 v = (u64)r[j] + c;
        r[j] = (u32)v & MASK;
        c = v >> W;
      }
    }
    return top;
  }

  // x (exact limbs, x <= 2N-ish but < 2^(W*S)) -> x mod N for x < 2N: one conditional subtraction,
  // done as x + (2^(W*S) - N) and a test of the carry out of the top limb.
  // Returns 1 (in every lane of the group) when N was subtracted.
  __device__ __forceinline__ u32 cond_sub(u32 (&x)[L]) const {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = (u64)x[j] + (u64)(MASK - n[j]);
    if (p == 0) t[0] += 1;
    u32 u[L];
    u64 top = normalize_full(u, t);
    u32 ge = LN::bcast_from((u32)(top != 0), K - 1);
#pragma unroll
    for (int j = 0; j < L; ++j) x[j] = ge ? u[j] : x[j];
    return ge;
  }

  // true (group-uniform) iff x == y limb for limb; both exact
  __device__ __forceinline__ bool equal(const u32 (&x)[L], const u32 (&y)[L]) const {
    u32 d = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) d |= x[j] ^ y[j];
    return !LN::group_any(d != 0);
  }
  __device__ __forceinline__ bool is_zero(const u32 (&x)[L]) const {
    u32 d = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) d |= x[j];
    return !LN::group_any(d != 0);
  }

  // r = x + y (lazy: result almost-normalised, value exact)
  __device__ __forceinline__ void add(u32 (&r)[L], const u32 (&x)[L], const u32 (&y)[L]) const {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = (u64)x[j] + (u64)y[j];
    normalize_weak(r, t);
  }

  // ------------------------------------------------------------------ one block of L limbs, unrolled by templates
  // weight of slot J at step I of a block: 0 = skip, 1 = a_J * b_I, 2 = a_J * 2 b_I (see SQUARE below)
  template <bool SQUARE, int I, int J>
  static constexpr int slot_weight() {
    if (!SQUARE) return 1;
    constexpr int d = (J - I + L) % L;
    if (d == 0 || (L % 2 == 0 && d == L / 2)) return 1;
    return (2 * d < L) ? 2 : 0;
  }

  template <bool SQUARE, int I, int J>
  __device__ __forceinline__ void product_mac(u64 (&t)[L], const u32 (&a)[L], u32 bi, u32 bi2) const {
    constexpr int w = slot_weight<SQUARE, I, J>();
    if constexpr (w == 1) t[J] += (u64)a[J] * bi;
    if constexpr (w == 2) t[J] += (u64)a[J] * bi2;
  }

  // Compile-time options of the word-serial product (bit mask F):
  //   F_RECORD_Q  keep the quotient digits q_i (limb i in the lane/slot that owns limb i)
  //   F_SQUARE    b is a: symmetric product (see below)
  //   F_TWO       two product rows per limb: t += a*b_i + c*d_i   (d staged next to b in LDS)
  //   F_INIT      the accumulator starts from `init` (lazy limbs) instead of 0
  //   F_PLAIN     no reduction: plain product; the limb leaving the group's lane 0 at every step
  //               is the next low limb of the result and is written to `emit` (LDS), the value left
  //               in the accumulator is the high part
  //   F_BDOUBLE   the multiplier staged in LDS is 2*b (used for the 2*X0*X1 row of a pair squaring)
  //   F_STAGED    the multiplier(s) are already in LDS (stage_multipliers): b and d are not read
  //   F_FRIENDLY  reduce modulo N~ = uf * N, the multiple of N that is -1 modulo 2^W, instead of N: the quotient
  //               digit of a step is then the low limb itself, q = t_0 mod 2^W — no multiplication by n0inv on
  //               the dependent chain — and t + q * N~ = t - q + q * (N~ + 1): column 0 leaves as before (its low
  //               W bits are q and are dropped), the other columns take q times the limbs of N~ + 1 (nf[], whose
  //               limb 0 is 0).  The result is = a*b/R modulo N as always, but only < 2 N~: the caller keeps
  //               W + 2 more bits of head room in R and reduces modulo N itself where it needs < 2N.  With
  //               F_RECORD_Q the digits recorded are those of Q~ with a*b + Q~ * N~ = r * R.
  //   F_INITQ     (with F_INIT) the accumulator starts from init + uf * initq, limb-wise
  static constexpr int F_RECORD_Q = 1, F_SQUARE = 2, F_TWO = 4, F_INIT = 8, F_PLAIN = 16, F_BDOUBLE = 32,
                       F_STAGED = 64, F_FRIENDLY = 128, F_INITQ = 256;

  // b (and d) where every lane of the group can read any limb; they stay valid until the next staging
  __device__ __forceinline__ void stage_multipliers(const u32 (&b)[L], const u32 (&d)[L]) {
    sync();
#pragma unroll
    for (int j = 0; j < L; ++j) { lds[p * L + j] = b[j]; lds[LDS_D + p * L + j] = d[j]; }
    sync();
  }

  template <int F, int I, int J>
  __device__ __forceinline__ void slot_macs(u64 (&t)[L], const u32 (&a)[L], const u32 (&c)[L], u32 bi, u32 bi2,
                                            u32 di, u32 q) const {
    if constexpr (J != 0) {
      product_mac<(F & F_SQUARE) != 0, I, J>(t, a, bi, bi2);
      if constexpr (F & F_TWO) t[J] += (u64)c[J] * di;
      if constexpr (!(F & F_PLAIN)) t[J] += (u64)((F & F_FRIENDLY) ? nf[J] : n[J]) * q;
    }
  }
  // the same in two rounds (few-limb instances): all products of the step in front of the quotient digit's broadcast,
  // so that the DPP move that reads column 0 has independent work between it and the multiply-add that wrote it
  // (2 wait states otherwise filled with s_nop), the reduction products behind it
  template <int F, int I, int J>
  __device__ __forceinline__ void slot_products(u64 (&t)[L], const u32 (&a)[L], const u32 (&c)[L], u32 bi, u32 bi2, u32 di) const {
    if constexpr (J != 0) {
      product_mac<(F & F_SQUARE) != 0, I, J>(t, a, bi, bi2);
      if constexpr (F & F_TWO) t[J] += (u64)c[J] * di;
    }
  }
  template <int F, int I, int J>
  __device__ __forceinline__ void slot_reduction(u64 (&t)[L], u32 q) const {
    if constexpr (J != 0 && !(F & F_PLAIN)) t[J] += (u64)((F & F_FRIENDLY) ? nf[J] : n[J]) * q;
  }

  template <int F, int I, int... Js>
  __device__ __forceinline__ void row_macs(u64 (&t)[L], const u32 (&a)[L], const u32 (&c)[L], u32 bi, u32 bi2, u32 di,
                                           u32 q, std::integer_sequence<int, Js...>) const {
    (slot_macs<F, I, Js>(t, a, c, bi, bi2, di, q), ...);
  }
  template <int F, int I, int... Js>
  __device__ __forceinline__ void row_products(u64 (&t)[L], const u32 (&a)[L], const u32 (&c)[L], u32 bi, u32 bi2, u32 di,
                                               std::integer_sequence<int, Js...>) const {
    (slot_products<F, I, Js>(t, a, c, bi, bi2, di), ...);
  }
  template <int F, int I, int... Js>
  __device__ __forceinline__ void row_reductions(u64 (&t)[L], u32 q, std::integer_sequence<int, Js...>) const {
    (slot_reduction<F, I, Js>(t, q), ...);
  }

  template <int F, int I>
  __device__ __forceinline__ void limb_step(u64 (&t)[L], const u32 (&a)[L], const u32 (&c)[L], const u32 (&bb)[L],
                                            const u32 (&dd)[L], u32 (&qr)[L], int blk, u32* emit) const {
    const u32 bi = bb[I];
    // doubled multiplier limb for the weight-2 products of a squaring (a shift is cheaper than a
    // second LDS copy of b: measured)
    const u32 bi2 = (F & F_SQUARE) ? (bi << 1) : 0u;
    const u32 di = dd[I];
    product_mac<(F & F_SQUARE) != 0, I, 0>(t, a, bi, bi2);
    if constexpr (F & F_TWO) t[0] += (u64)c[0] * di;
    constexpr bool TWO_ROUNDS = L <= 4;
    // pinned in place for the one-row full products (measured in instructions per loop trip: 111 -> 105 and 102 -> 97;
    // the symmetric and the two-row flavours came out longer with the pins and are left to the scheduler)
    constexpr bool PINNED = TWO_ROUNDS && !(F & F_SQUARE) && !(F & F_TWO);
    if constexpr (TWO_ROUNDS) {
      if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
      row_products<F, I>(t, a, c, bi, bi2, di, std::make_integer_sequence<int, L>{});
      if constexpr (PINNED) __builtin_amdgcn_sched_barrier(0);
    }
    u32 q = 0;
    if constexpr (F & F_PLAIN) {
      if (p == 0) emit[blk * L + I] = (u32)t[0] & MASK;
    } else {
      // the mask is applied after the broadcast so that it folds into the DPP move (v_and_b32_dpp)
      if constexpr (F & F_FRIENDLY) {
        q = LN::bcast0_and((u32)t[0], maskv);
      } else {
        q = LN::bcast0_and((u32)t[0] * n0inv, maskv);
      }
      if constexpr (F & F_RECORD_Q) qr[I] = (blk == p) ? q : qr[I];
      t[0] += (u64)((F & F_FRIENDLY) ? nf[0] : n[0]) * q;
    }
    if constexpr (TWO_ROUNDS) {
      row_reductions<F, I>(t, q, std::make_integer_sequence<int, L>{});
    } else {
      row_macs<F, I>(t, a, c, bi, bi2, di, q, std::make_integer_sequence<int, L>{});
    }
    // divide by 2^W: column 0 leaves; its low W bits belong to the lower neighbour's top column
    // (zero for the group's lane 0 by construction of q), the rest carries into column 1
    const u64 carry = t[0] >> W;
    const u32 recv = LN::from_next_raw((u32)t[0]) & next_mask;
    t[0] = t[1] + carry;
#pragma unroll
    for (int j = 1; j < L - 1; ++j) t[j] = t[j + 1];
    // the incoming word opens the new top column as recv * 1: one multiply-add wherever the
    // compiler places it in the column's chain, instead of a zero-extension plus a 64-bit add
    t[L - 1] = (u64)recv * onev;
  }

  template <int F, int... Is>
  __device__ __forceinline__ void block_steps(u64 (&t)[L], const u32 (&a)[L], const u32 (&c)[L], const u32 (&bb)[L],
                                              const u32 (&dd)[L], u32 (&qr)[L], int blk,
                                              u32* emit, std::integer_sequence<int, Is...>) const {
    (limb_step<F, Is>(t, a, c, bb, dd, qr, blk, emit), ...);
  }

  template <int F, int I0, int... Is>
  __device__ __forceinline__ void block_steps_from(u64 (&t)[L], const u32 (&a)[L], const u32 (&c)[L], const u32 (&bb)[L],
                                                   const u32 (&dd)[L], u32 (&qr)[L], int blk,
                                                   u32* emit, std::integer_sequence<int, Is...>) const {
    (limb_step<F, I0 + Is>(t, a, c, bb, dd, qr, blk, emit), ...);
  }

  // ------------------------------------------------------------------ Montgomery product
  // r = (a*b [+ c*d] [+ init]) / R mod N   (lazy: r < 2N for operands < 4N, R >= 16 N); r may alias
  // any operand.  With F_PLAIN: r = high part of a*b [+ c*d] [+ init], low limbs in `emit`.
  //
  // F_SQUARE (b is a): the product part uses the symmetry a_u a_v = a_v a_u without moving any data.
  // At the unrolled step i of a block (multiplier limb u with u mod L == i) a lane only multiplies
  // its slots j whose cyclic distance d = (j - i) mod L is <= L/2: with weight 2 for 0 < d < L/2 and
  // weight 1 for d == 0 (and d == L/2 when L is even).  For u != v exactly one of the two orders
  // has distance < L/2 (weight 2), or both have distance 0 or L/2 (weight 1 + 1); u == v is met once
  // with weight 1 — so every term of a^2 gets its coefficient, the selection is the same in every
  // lane (compile-time register indices), and floor(L/2)+1 instead of L product MACs are issued.
  template <int F>
  // a and c are not const: their registers are passed through an empty asm once per block (see
  // below); values are unchanged.
  __device__ __forceinline__ void mulx(u32 (&r)[L], u32 (&a)[L], const u32 (&b)[L], u32 (&c)[L],
                                       const u32 (&d)[L], const u32 (&init)[L], u32* qrec, u32* emit, int nsteps_blk,
                                       const u32* initq = nullptr) {
    // stage the multiplier(s) where every lane of the group can read any limb
    if constexpr (!(F & F_STAGED)) {
      sync();
#pragma unroll
      for (int j = 0; j < L; ++j) lds[p * L + j] = (F & F_BDOUBLE) ? (b[j] << 1) : b[j];
      if constexpr (F & F_TWO) {
#pragma unroll
        for (int j = 0; j < L; ++j) lds[LDS_D + p * L + j] = d[j];
      }
      sync();
    }
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
      if constexpr ((F & F_INIT) && (F & F_INITQ)) {
        t[j] = (u64)init[j] + (u64)initq[j] * uf;
      } else {
        t[j] = (F & F_INIT) ? (u64)init[j] : 0;
      }
    }
    u32 qr[L];
#pragma unroll
    for (int j = 0; j < L; ++j) qr[j] = 0;
    // The multiplier limbs of a block come from LDS.  With few limbs per lane a block is only a few dozen
    // instructions, and a wavefront that has its SIMD to itself (the latency geometry's reason to exist) would
    // sit out the LDS round trip at the head of every block: there the limbs of block blk + 1 are fetched
    // before block blk is worked on.  The large-L instances hide the latency behind their own work and keep
    // the registers.
    constexpr bool PREFETCH = L <= 4;
    constexpr bool HALVES = L >= 16 && (F & F_TWO) != 0;
    u32 nb[L], nd[L];
    if constexpr (PREFETCH) {
#pragma unroll
      for (int j = 0; j < L; ++j) { nb[j] = lds[j]; nd[j] = (F & F_TWO) ? lds[LDS_D + j] : 0u; }
    }
    auto do_block = [&](int blk) {
      // The multiplicand limbs are loop invariant, and the compiler would hoist their zero
      // extension to 64 bits out of this loop — every limb then occupies a register PAIR for the
      // whole product (v_mad_u64_u32 only reads the low half).  Opaque per block: the extension
      // folds into the multiply-add and the limbs stay in single registers.
      // (not for the few-limb instances: registers are no concern there, and the barriers cost them moves)
#pragma unroll
      for (int j = 0; j < (PREFETCH ? 0 : L); ++j) {
        asm volatile("" : "+v"(a[j]));
        if constexpr (F & F_TWO) asm volatile("" : "+v"(c[j]));
        if constexpr (!(F & F_PLAIN)) {
          if constexpr (F & F_FRIENDLY) {
            asm volatile("" : "+v"(nf[j]));
          } else {
            asm volatile("" : "+v"(n[j]));
          }
        }
      }
      u32 bb[L], dd[L];
      if constexpr (PREFETCH) {
        const int nx = blk + 1 < nsteps_blk ? blk + 1 : blk;
#pragma unroll
        for (int j = 0; j < L; ++j) { bb[j] = nb[j]; dd[j] = nd[j]; }
#pragma unroll
        for (int j = 0; j < L; ++j) { nb[j] = lds[nx * L + j]; nd[j] = (F & F_TWO) ? lds[LDS_D + nx * L + j] : 0u; }
      } else if constexpr (HALVES) {
        // two product rows at 18 limbs per lane: the 2 L multiplier limbs of a block are fetched in two halves, each in
        // front of the steps that use it — with all 36 in registers at once the pass did not fit its 256 registers
        // (round 3: 28 spilled registers, 116 B of scratch per lane)
        constexpr int H = L / 2;
#pragma unroll
        for (int j = 0; j < H; ++j) { bb[j] = lds[blk * L + j]; dd[j] = lds[LDS_D + blk * L + j]; }
        block_steps_from<F, 0>(t, a, c, bb, dd, qr, blk, emit, std::make_integer_sequence<int, H>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = H; j < L; ++j) { bb[j] = lds[blk * L + j]; dd[j] = lds[LDS_D + blk * L + j]; }
        block_steps_from<F, H>(t, a, c, bb, dd, qr, blk, emit, std::make_integer_sequence<int, L - H>{});
        return;
      } else {
#pragma unroll
        for (int j = 0; j < L; ++j) bb[j] = lds[blk * L + j];
#pragma unroll
        for (int j = 0; j < L; ++j) dd[j] = (F & F_TWO) ? lds[LDS_D + blk * L + j] : 0u;
      }
      block_steps<F>(t, a, c, bb, dd, qr, blk, emit, std::make_integer_sequence<int, L>{});
    };
    if constexpr (PREFETCH) {
      // three blocks per trip: with 3 limbs per lane the loop control of a trip is a tenth of the instructions of two
      int blk = 0;
      for (; blk + 2 < nsteps_blk; blk += 3) { do_block(blk); do_block(blk + 1); do_block(blk + 2); }
      for (; blk < nsteps_blk; ++blk) do_block(blk);
    } else {
      for (int blk = 0; blk < nsteps_blk; ++blk) do_block(blk);
    }
    normalize_weak(r, t);
    if constexpr (F & F_RECORD_Q) {
#pragma unroll
      for (int j = 0; j < L; ++j) qrec[j] = qr[j];
    }
  }

  template <bool RECORD_Q = false, bool SQUARE = false>
  __device__ __forceinline__ void mul(u32 (&r)[L], const u32 (&a)[L], const u32 (&b)[L], u32* qrec = nullptr) {
    u32 av[L];
#pragma unroll
    for (int j = 0; j < L; ++j) av[j] = a[j];
    mulx<(RECORD_Q ? F_RECORD_Q : 0) | (SQUARE ? F_SQUARE : 0)>(r, av, b, av, b, b, qrec, nullptr, nblk);
  }

  // r = a^2 / R mod N (lazy), with the symmetric product
  __device__ __forceinline__ void sqr(u32 (&r)[L], const u32 (&a)[L]) { mul<false, true>(r, a, a); }

  // the same two modulo the friendly multiple of N (F_FRIENDLY: results below 2 N~; nf / uf must be set)
  template <bool SQUARE = false>
  __device__ __forceinline__ void mul_friendly(u32 (&r)[L], const u32 (&a)[L], const u32 (&b)[L]) {
    u32 av[L];
#pragma unroll
    for (int j = 0; j < L; ++j) av[j] = a[j];
    mulx<F_FRIENDLY | (SQUARE ? F_SQUARE : 0)>(r, av, b, av, b, b, nullptr, nullptr, nblk);
  }

  // ------------------------------------------------------------------ constants
  // one: the integer 1 (limb 0 of the group's lane 0)
  __device__ __forceinline__ void set_small(u32 (&x)[L], u32 v) const {
#pragma unroll
    for (int j = 0; j < L; ++j) x[j] = 0;
    if (p == 0) x[0] = v;
  }

  // Given rmodn = R mod N (Montgomery form of 1), returns R^2 mod N (Montgomery form of R),
  // by raising 2 to the power W*L*nblk in the Montgomery domain (square-and-double).
  __device__ __forceinline__ void compute_r2(u32 (&r2)[L], const u32 (&rmodn)[L]) {
    const int m = W * L * nblk;
    u32 x[L];
#pragma unroll
    for (int j = 0; j < L; ++j) x[j] = rmodn[j];
    int top = 31 - __builtin_clz((unsigned)m);
    for (int bit = top; bit >= 0; --bit) {
      if (bit != top) sqr(x, x);
      if ((m >> bit) & 1) add(x, x, x);
    }
#pragma unroll
    for (int j = 0; j < L; ++j) r2[j] = x[j];
  }

  // R mod N (the Montgomery form of 1) from the modulus alone, exact limbs: start from the largest
  // power of two below N and double with a conditional subtraction until 2^(W*L*nblk) is reached.
  // R >= 16 N, so this is at most W*L + 4 doublings for a modulus that fills its geometry; groups of
  // one wavefront may hold moduli of different lengths, so the loop runs over the wave-wide range
  // and every group joins in at its own starting bit (uniform control flow, no divergence).
  __device__ __forceinline__ void rmodn_by_doubling(u32 (&x)[L]) {
    int top = -1;
#pragma unroll
    for (int j = 0; j < L; ++j) top = n[j] ? W * (p * L + j) + (31 - __builtin_clz(n[j])) : top;
#pragma unroll
    for (int off = K / 2; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off));   // stays inside the group
    int lo;
    lo = top;
    for (int off = 32; off > 0; off >>= 1) lo = min(lo, __shfl_xor(lo, off));
    lo = __builtin_amdgcn_readfirstlane(lo);
#pragma unroll
    for (int j = 0; j < L; ++j) x[j] = (top >= 0 && top / W == p * L + j) ? (1u << (top % W)) : 0u;
    const int m = W * L * nblk;
    for (int i = (lo < 0 ? 0 : lo); i < m; ++i) {
      u64 t[L];
#pragma unroll
      for (int j = 0; j < L; ++j) t[j] = (u64)x[j] << 1;
      u32 y[L];
      normalize_full(y, t);
      cond_sub(y);
      const bool take = i >= top;
#pragma unroll
      for (int j = 0; j < L; ++j) x[j] = take ? y[j] : x[j];
    }
  }

  // lazy Montgomery-domain value -> canonical residue in [0, N), exact limbs
  __device__ __forceinline__ void from_mont_canonical(u32 (&out)[L], const u32 (&x)[L]) {
    u32 one[L];
    set_small(one, 1);
    u32 y[L];
    mul(y, x, one);               // y = x / R mod N, y <= N
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = y[j];
    normalize_full(out, t);
    cond_sub(out);
  }
};

}  // namespace mx
