// Bipartite latency form of the generic-modulus modexp:  out[e] = bases[e] ^ exp[g(e)] mod N[g(e)]  with every modular
// product split over TWO wavefronts (tools/bimont_model.py is the column-exact model of this file; tests/test_bimont_model.py
// runs it against pow()).
//
// Why.  The biprimality-test modexps of a key-generation round at the reference's batch sizes are a few dozen to a thousand
// per launch (distributed_keygen.py:1084-1099 looped at :1313-1329 over the 2-25 survivors of a round).  Such a launch lasts
// as long as ONE wavefront's dependent chain: a Montgomery product is word-serial over the limbs of the multiplier, and a lone
// wavefront issues an instruction every ~5 cycles whatever it is.  Fewer limbs per lane do not shorten
// the chain (the number of limb steps is the multiplier's length); only fewer STEPS do.  A word-serial product can shed at
// most half of them: with a pivot hL,
//
//     a * b * theta (mod N),  theta = 2^(-W*hL),  b = b_lo + 2^(W*hL) * b_hi
//       = [ a * b_lo * 2^(-W*hL) ]   hL least-significant-first steps: the friendly-modulus Montgomery passes of mx_mont.hpp
//       + [ a * b_hi ]               the remaining steps most-significant-first: shift the accumulator UP one limb, add a * b_i,
//                                    fold what left the top back in through 2^(W*(Ptop+1)) mod N
//
// (Kaihara & Takagi's bipartite modular multiplication) — the two halves are independent until their sum.  A workgroup is two
// wavefronts: wavefront L runs the first half with the unchanged Mont<K,3,W>::mulx, wavefront H the second half on the SAME
// lane-distributed reduced-radix format held in REVERSED order (lane 0 = most significant), so that its data moves the way
// mx_mont.hpp's does (towards lane 0, by the same DPP moves) and its fold digit is born where the quotient digit is (lane 0:
// the same broadcast).  H then adds L's half (handed over through LDS), folds the six top positions, sweeps the carries and
// publishes the product in LDS, position-indexed — which is at once the next multiplier (every lane of either wavefront reads
// any limb of it) and the hand-over of the accumulator to L.  Two workgroup barriers per product (a workgroup holds two such
// pairs, which execute the same sequence of products).
//
// Width discipline (asserted by the model): columns are lazy 64-bit sums as in mx_mont.hpp; the words that cross lanes are a
// W-bit limb and a carry word < 2^32; the fold digit is < 2^32 because the TWO most significant lanes of H take no products and
// no folds (the operands and 2^k mod N are below them): a carry word that enters them is split once more on its way to the top,
// which makes the feedback from a fold to the next fold digit 3 * 2^-29 instead of 3.
#pragma once
#include "mx_mont.hpp"

namespace mx {

constexpr int BI_ROWS = 9;   // per group in PowmodBiArgs::consts: rows 0..5 = 2^(W*(Pd+k)) mod N, 6 = 2^(W*(Pd+6)) mod N, 7 = 2^(W*(hL+Pd)) mod N, 8 = 2^(W*(Pd-hL)) mod N

struct PowmodBiArgs {
  const u32* bases;   // [batch][limbs]      device, radix 2^32 words
  u32* out;           // [batch][limbs]
  const u32* mods;    // [groups][limbs]
  const u32* exps;    // [groups][elimbs]
  const u32* consts;  // [groups][BI_ROWS][3*K] W-bit limbs, position-indexed (bisetup_kernel)
  u32* table;         // [2^win][3][nlanes]   window table in wavefront L's register layout
  i64 batch;
  i64 group_size;
  int limbs, elimbs;
  int ndigits, win;
  int nblk;           // Montgomery radix of the conversions: R = 2^(W*3*nblk) = 2^(W*Pd)
  int pd;             // data positions: operands < 2^(W*Pd) * (1 + tiny), limb Pd is 0 or 1
  int h_lo;           // pivot: multiplier limbs [0, h_lo) go to wavefront L (multiple of 3)
};

struct BiSetupArgs {
  const u32* mods;    // [groups][limbs]
  u32* consts;        // [groups][BI_ROWS][3*K]
  i64 groups;
  int limbs, nblk, pd, h_lo;
};

// ---- the constants of a modulus: 2^m mod N for the seven fold positions, and the conversion factor ---------------------------
template <int K, int W>
__global__ void __launch_bounds__(64) bisetup_kernel(BiSetupArgs A) {
  constexpr int L = 3, PW = L * K;
  using M_t = Mont<K, L, W, true>;
  extern __shared__ u32 smem[];
  constexpr int GPW = 64 / K;
  const int gw = threadIdx.x / K;
  const i64 g_raw = (i64)blockIdx.x * GPW + gw;
  const bool valid = g_raw < A.groups;
  const i64 g = valid ? g_raw : A.groups - 1;
  M_t M;
  M.init(smem + gw * M_t::LDS_WORDS, A.nblk);
  M.load(M.n, A.mods + g * A.limbs, A.limbs);
  M.setup_modulus();
  u32* rows = A.consts + g * BI_ROWS * PW;
  const int p = M.p;
  u32 x[L], rmodn[L];
  M.rmodn_by_doubling(x);                      // 2^(W*3*nblk) = 2^(W*Pd) mod N, exact limbs
#pragma unroll
  for (int j = 0; j < L; ++j) rmodn[j] = x[j];
  for (int k = 0; k < 7; ++k) {
    if (valid) {
#pragma unroll
      for (int j = 0; j < L; ++j) rows[k * PW + p * L + j] = x[j];
    }
    for (int d = 0; d < W; ++d) {              // times 2^W: W doublings with a conditional subtraction each
      u64 t[L];
#pragma unroll
      for (int j = 0; j < L; ++j) t[j] = (u64)x[j] << 1;
      M.normalize_full(x, t);
      M.cond_sub(x);
    }
  }
  // conversion into the domain: g -> g * theta^-1 = MontMul(g, kin), kin = 2^(W*hL) * R mod N = MontMul(2^(W*hL), R^2)
  u32 r2[L], pw[L], kin[L];
  M.compute_r2(r2, rmodn);
#pragma unroll
  for (int j = 0; j < L; ++j) pw[j] = (p * L + j == A.h_lo) ? 1u : 0u;
  M.mul(kin, pw, r2);
  if (valid) {
#pragma unroll
    for (int j = 0; j < L; ++j) rows[7 * PW + p * L + j] = kin[j];
  }
  // the factor that leaves the domain, theta * R = 2^(W*(Pd - hL)), REDUCED modulo this group's N.  (Round 5 first used the
  // power of two as it is: the geometry keeps it below a modulus of the launch's bit length, but a launch has one geometry
  // and every group its own modulus — for a modulus a few bits shorter than the longest the epilogue's two conditional
  // subtractions then left N + 1 where 1 was due; found by tools/soak_round5.py, seed 19.)
  u32 one[L], up[L], ko[L];
#pragma unroll
  for (int j = 0; j < L; ++j) {
    pw[j] = (p * L + j == A.pd - A.h_lo) ? 1u : 0u;
    one[j] = (p * L + j == 0) ? 1u : 0u;
  }
  M.mul(up, pw, r2);                           // 2^(W*(Pd-hL)) * R mod N
  M.mul(ko, up, one);                          // ... / R: the power itself, lazy
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = ko[j];
    M.normalize_full(ko, t);
    M.cond_sub(ko);
    M.cond_sub(ko);
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < L; ++j) rows[8 * PW + p * L + j] = ko[j];
  }
}

// ---- wavefront H: the most-significant-first half, the sum and the final reduction -------------------------------------------
template <int K, int W>
struct BiHi {
  static constexpr int L = 3;
  using LN = Lanes<K, true>;
  static constexpr u32 MASK = (1u << W) - 1u;
  u32 rf[L];          // 2^(W*(Ptop+1)) mod N, this lane's positions
  u32 fin[6][L];      // 2^(W*(Pd+k)) mod N
  u32 limb_mask;      // MASK if the lane below belongs to the number, else 0 (folds into the DPP move)
  u32 word_mask;      // ~0 / 0 likewise, for carry words
  u32 top_mask;       // lane 0 keeps the excess of its top column (nothing lies above it), the others W bits
  u32 low_keep;       // 0 in the two top lanes (cleared by the final fold), ~0 below
  u32 onev;           // 1 in a VGPR the compiler cannot see through (mx_mont.hpp: limb_step)
  int p, ptop;
  int addr[L];        // LDS index of this lane's positions (clamped to 0 where the lane lies below the number)
  int waddr[L];       // ... for writes: a spare word behind the row instead
  u32 pos_ok[L];      // ~0 where the position exists
  u32 data_ok[L];     // ~0 where it is a data position (0 <= position < Pd)

  __device__ __forceinline__ void init(int pd) {
    p = LN::pos();
    ptop = pd + 5;
    const int lanes = pd / L + 2;
    const bool below = p + 1 < lanes;
    limb_mask = below ? MASK : 0u;
    word_mask = below ? ~0u : 0u;
    top_mask = p == 0 ? ~0u : MASK;
    low_keep = p >= 2 ? ~0u : 0u;
    onev = 1u;
    asm volatile("" : "+v"(onev));
    asm volatile("" : "+v"(limb_mask));
    asm volatile("" : "+v"(word_mask));
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const int pos = ptop - (L * p + j);
      addr[j] = pos >= 0 ? pos : 0;
      waddr[j] = pos >= 0 ? pos : L * K;
      pos_ok[j] = pos >= 0 ? ~0u : 0u;
      data_ok[j] = (pos >= 0 && pos < pd) ? ~0u : 0u;
    }
  }

  // this lane's limbs of a position-indexed row (global or LDS)
  __device__ __forceinline__ void gather(u32 (&dst)[L], const u32* row) const {
#pragma unroll
    for (int j = 0; j < L; ++j) dst[j] = row[addr[j]] & pos_ok[j];
  }

  // weight of column j at a step whose multiplier limb index is I3 modulo 3 (positions are 2 - j modulo 3: Ptop = 2 mod 3)
  template <bool SQ, int I3, int J>
  static constexpr int weight() {
    if (!SQ) return 1;
    constexpr int d = ((2 - J - I3) % 3 + 3) % 3;
    return d == 0 ? 1 : (d == 1 ? 2 : 0);
  }

  template <bool SQ, int I3>
  __device__ __forceinline__ void step(u64 (&t)[L], const u32 (&ar)[L], const u32 (&rf)[L], u32 onev, u32 onew, u32 bi) const {
    const u64 out = t[0];                                            // leaves this lane: everything of it moves up
    const u32 v = LN::bcast0((u32)out);                              // ... and what leaves lane 0 is folded
    const u32 rl = LN::from_next_raw((u32)out) & limb_mask;          // the limb from the lane below -> new bottom column
    const u32 rc = LN::from_next_raw((u32)(out >> W)) & word_mask;   // its carry word -> the column above that
    t[0] = t[1];
    t[1] = t[2] + (u64)rc * onew;        // (two different opaque ones: with one the compiler factors it out of the sum and
    t[2] = (u64)rl * onev;               // multiplies a 64-bit value by it — two multiply-adds and a move more per step)
    const u32 bi2 = SQ ? (bi << 1) : 0u;
    if constexpr (weight<SQ, I3, 0>() == 1) t[0] += (u64)ar[0] * bi;
    if constexpr (weight<SQ, I3, 0>() == 2) t[0] += (u64)ar[0] * bi2;
    if constexpr (weight<SQ, I3, 1>() == 1) t[1] += (u64)ar[1] * bi;
    if constexpr (weight<SQ, I3, 1>() == 2) t[1] += (u64)ar[1] * bi2;
    if constexpr (weight<SQ, I3, 2>() == 1) t[2] += (u64)ar[2] * bi;
    if constexpr (weight<SQ, I3, 2>() == 2) t[2] += (u64)ar[2] * bi2;
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] += (u64)rf[j] * v;
  }

  // t = ar * (B[pd+2] .. B[h_lo]) most significant limb first, folded to Ptop + 1 positions; B position-indexed in LDS
  template <bool SQ>
  __device__ __forceinline__ void half(u64 (&t)[L], const u32 (&ar)[L], const u32* B, int pd, int h_lo) const {
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = 0;
    u32 al[L], rl[L], one = onev, one2 = onev;
#pragma unroll
    for (int j = 0; j < L; ++j) { al[j] = ar[j]; rl[j] = rf[j]; }
    // loop-invariant 32-bit multiplicands: opaque once per trip, or the compiler hoists their zero-extension out of the loop
    // and multiplies register PAIRS (two multiply-adds per product; mx_mont.hpp mulx has the same guard)
    auto opaque = [&]() {
#pragma unroll
      for (int j = 0; j < L; ++j) { asm volatile("" : "+v"(al[j])); asm volatile("" : "+v"(rl[j])); }
      asm volatile("" : "+v"(one));
      asm volatile("" : "+v"(one2));
    };
    // The chain opens at limb Pd (round 6): the multiplier limbs at Pd + 1 and Pd + 2 are zero in every row this kernel
    // multiplies by (a product leaves at most 2 at Pd and nothing above it, the converted base is below 2 N;
    // tools/bimont_model.py asserts it), so the two steps that would run on an empty accumulator are left out.
    int i = pd - 1;
    // three blocks (nine limb steps) per trip — the loop control and address arithmetic of a one-block trip are a fifth of
    // its instructions —, the next trip's multiplier limbs fetched behind this trip's work
    const bool trips = i - 8 >= h_lo;
    u32 nb[9];
    const u32 btop = B[pd];
    if (trips) {
#pragma unroll
      for (int k = 0; k < 9; ++k) nb[k] = B[i - k];
    }
    opaque();
    step<SQ, 0>(t, al, rl, one, one2, btop);
    if (trips) {
      for (; i - 8 >= h_lo; i -= 9) {
        opaque();
        u32 b[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) b[k] = nb[k];
        const int nx = i - 17 >= h_lo ? i - 9 : i;
#pragma unroll
        for (int k = 0; k < 9; ++k) nb[k] = B[nx - k];
        step<SQ, 2>(t, al, rl, one, one2, b[0]); step<SQ, 1>(t, al, rl, one, one2, b[1]); step<SQ, 0>(t, al, rl, one, one2, b[2]);
        step<SQ, 2>(t, al, rl, one, one2, b[3]); step<SQ, 1>(t, al, rl, one, one2, b[4]); step<SQ, 0>(t, al, rl, one, one2, b[5]);
        step<SQ, 2>(t, al, rl, one, one2, b[6]); step<SQ, 1>(t, al, rl, one, one2, b[7]); step<SQ, 0>(t, al, rl, one, one2, b[8]);
      }
    }
    for (; i >= h_lo; i -= 3) {
      opaque();
      const u32 b2 = B[i], b1 = B[i - 1], b0 = B[i - 2];
      step<SQ, 2>(t, al, rl, one, one2, b2);
      step<SQ, 1>(t, al, rl, one, one2, b1);
      step<SQ, 0>(t, al, rl, one, one2, b0);
    }
  }

  // carry sweep towards the more significant end (the mirror image of Mont::normalize_weak): column 2 -> 1 -> 0 inside the
  // lane, one hop to the lane above, a two-limb fix-up there.  Value preserved; every limb < 2^W + 2^8 afterwards, except that
  // lane 0's top column keeps whatever excess it has (TOP).
  template <bool TOP>
  __device__ __forceinline__ void sweep(u32 (&r)[L], const u64 (&t)[L]) const {
    u64 c = 0;
    u64 v = t[2];
    r[2] = (u32)v & MASK; c = v >> W;
    v = t[1] + c;
    r[1] = (u32)v & MASK; c = v >> W;
    v = t[0] + c;
    r[0] = (u32)v & (TOP ? top_mask : MASK); c = v >> W;
    const u32 clo = LN::from_next_raw((u32)c) & word_mask;
    const u32 chi = LN::from_next_raw((u32)(c >> 32)) & word_mask;
    const u64 w = (u64)r[2] + ((u64)clo | ((u64)chi << 32));
    r[2] = (u32)w & MASK;
    r[1] += (u32)(w >> W);
  }

  // The end of a product, in two parts around the hand-over.
  // BEFORE it (wavefront L may still be working): sweep this half — the lazy columns hold up to 2^61, afterwards what stands
  // at positions >= Pd is its true top —, clear those six positions and fold five of them (Pd+1 .. Pd+5: wavefront L's half
  // has nothing there) back in; the sixth digit, position Pd, waits for L's limb there.  Returns that digit's own part.
  __device__ __forceinline__ u32 pre(u64 (&t)[L]) const {
    u32 r[L];
    sweep<true>(r, t);
    // lane 0 holds positions Pd+5, Pd+4, Pd+3, lane 1 Pd+2, Pd+1, Pd
    u32 dg[6];
    dg[5] = LN::bcast0(r[0]); dg[4] = LN::bcast0(r[1]); dg[3] = LN::bcast0(r[2]);
    dg[2] = LN::bcast0(LN::from_next_raw(r[0])); dg[1] = LN::bcast0(LN::from_next_raw(r[1])); dg[0] = LN::bcast0(LN::from_next_raw(r[2]));
#pragma unroll
    for (int j = 0; j < L; ++j) {
      u64 s = (u64)(r[j] & low_keep);
#pragma unroll
      for (int k = 1; k < 6; ++k) {
        u32 f = fin[k][j];
        asm volatile("" : "+v"(f));            // a 32-bit multiplicand, not a hoisted register pair (see half())
        s += (u64)f * dg[k];
      }
      t[j] = s;
    }
    return dg[0];
  }

  // AFTER it: + wavefront L's half (TL, position-indexed; its limb at Pd joins the last digit), fold that digit, sweep,
  // publish the product's limbs in C and keep them in ar
  __device__ __forceinline__ void post(u64 (&t)[L], u32 dg0, const u32* TL, u32* C, u32 (&ar)[L], int pd) const {
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] += (u64)(TL[addr[j]] & data_ok[j]);
    const u32 d = dg0 + TL[pd];                // every lane reads the same word
#pragma unroll
    for (int j = 0; j < L; ++j) {
      u32 f = fin[0][j];
      asm volatile("" : "+v"(f));
      t[j] += (u64)f * d;
    }
    u32 r[L];
    sweep<false>(r, t);
#pragma unroll
    for (int j = 0; j < L; ++j) {
      ar[j] = r[j] & pos_ok[j];
      C[waddr[j]] = r[j];                      // lanes below the number write their zeros to a spare word behind the row
    }
  }
};

// ---- the kernel --------------------------------------------------------------------------------------------------------------
// A workgroup is TWO such pairs — four wavefronts, one per SIMD of a compute unit (with one pair per workgroup the dispatcher
// puts the wavefronts of a second workgroup on SIMDs the first already uses: 500 pairs took 6.0 ms where 250 take 4.3).
constexpr int BI_PAIRS = 2;

#ifdef MX_DEV_BI_TRACE
__device__ u64 mx_bi_trace[16];
#endif
template <int K, int W>
__global__ void __launch_bounds__(128 * BI_PAIRS) powmod_bi_kernel(PowmodBiArgs A) {
  constexpr int L = 3, PW = L * K, GPW = 64 / K;
  using M_t = Mont<K, L, W, true, false>;        // wavefront-level fences only: the two wavefronts run different code
  using H_t = BiHi<K, W>;
  constexpr int ROW = PW + 4;                    // a position-indexed row + a spare word for wavefront H's out-of-range lanes
  constexpr int GROUP_WORDS = 3 * ROW + M_t::LDS_WORDS;
  extern __shared__ u32 smem[];
  const int role = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) & 1));     // 0 = wavefront L, 1 = wavefront H
  const int pair = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));           // which pair of the workgroup
  const int lane = threadIdx.x & 63;
  const int gw = lane / K;
  u32* C = smem + (pair * GPW + gw) * GROUP_WORDS;   // the accumulator, position-indexed: multiplier of a squaring, hand-over H -> L
  u32* F = C + ROW;                              // multiplier of a multiplication (a table row)
  u32* TL = F + ROW;                             // wavefront L's half of a product, hand-over L -> H
  u32* ST = TL + ROW;                            // staging of Mont::load / store / mul (wavefront L only)
  const i64 pair_id = (i64)blockIdx.x * BI_PAIRS + pair;
  const i64 elem_raw = pair_id * GPW + gw;
  const bool valid = elem_raw < A.batch;
  const i64 elem = valid ? elem_raw : A.batch - 1;       // surplus groups (and a surplus pair) redo the last element, no store
  const i64 grp = elem / A.group_size;
  const i64 nlanes = (i64)gridDim.x * BI_PAIRS * 64;
  const u32* crow = A.consts + grp * BI_ROWS * PW;
  const int nblk_lo = A.h_lo / L;

  M_t M;
  H_t H;
  u32 a[L];                                      // the accumulator: L in its layout, H in the reversed one
  u32* tbl = A.table + pair_id * 64 + lane;
  const int p = lane & (K - 1);

  // ---- prologue: L converts the base and stages it, H collects its constants
  M.init(ST, A.nblk);                            // (both wavefronts: members defined on one path only would reach the loops
  H.init(A.pd);                                  //  through phi nodes and lose what the compiler knows about them)
  if (role == 0) {
    M.load(M.n, A.mods + grp * A.limbs, A.limbs);
    M.setup_modulus();
    M.compute_friendly();
    u32 g[L], kin[L];
    M.load(g, A.bases + elem * A.limbs, A.limbs);
#pragma unroll
    for (int j = 0; j < L; ++j) kin[j] = crow[7 * PW + p * L + j];
    M.mul(a, g, kin);                            // g * theta^-1 mod N, lazy
#pragma unroll
    for (int j = 0; j < L; ++j) {
      C[p * L + j] = a[j];
      F[p * L + j] = a[j];
      tbl[(i64)j * nlanes] = (p * L + j == A.h_lo) ? 1u : 0u;      // the domain's one: theta^-1 = 2^(W*hL) < N
      tbl[((i64)L + j) * nlanes] = a[j];
    }
  } else {
    H.gather(H.rf, crow + 6 * PW);
#pragma unroll
    for (int k = 0; k < 6; ++k) H.gather(H.fin[k], crow + k * PW);
  }
  __syncthreads();
  if (role == 1) H.gather(a, C);

  // one product: a <- a * B * theta (B = C: squaring), both wavefronts
#ifdef MX_DEV_BI_TRACE          // developer builds (tools/bi_phase_probe.py): shader-clock cycles per phase of a product, pair 0 of workgroup 0
  u64 trc[5] = {0, 0, 0, 0, 0};
#define MX_BI_MARK(k) { const u64 now_ = __builtin_readcyclecounter(); trc[k] += now_ - mark_; mark_ = now_; }
#else
#define MX_BI_MARK(k)
#endif
  auto product = [&](auto sq_tag, const u32* B) {
    constexpr bool SQ = decltype(sq_tag)::value;
    u64 t[L];
    u32 dg0 = 0;
#ifdef MX_DEV_BI_TRACE
    u64 mark_ = __builtin_readcyclecounter();
#endif
    if (role == 0) {
      u32 r[L];
      M.lds = const_cast<u32*>(B);
      M.template mulx<M_t::F_FRIENDLY | M_t::F_STAGED | (SQ ? M_t::F_SQUARE : 0)>(r, a, a, a, a, a, nullptr, nullptr, nblk_lo);
#pragma unroll
      for (int j = 0; j < L; ++j) TL[p * L + j] = r[j];
      MX_BI_MARK(0)                                  // L: its half
    } else {
      H.template half<SQ>(t, a, B, A.pd, A.h_lo);
      MX_BI_MARK(0)                                  // H: its half
      dg0 = H.pre(t);
      MX_BI_MARK(1)                                  // H: sweep + fold of five top positions
    }
    __syncthreads();
    MX_BI_MARK(2)                                    // waiting for the other half
    if (role == 1) H.post(t, dg0, TL, C, a, A.pd);
    MX_BI_MARK(3)                                    // H: + L's half, last fold, sweep, publish (L: nothing)
    __syncthreads();
    if (role == 0) {
#pragma unroll
      for (int j = 0; j < L; ++j) a[j] = C[p * L + j];
    }
    MX_BI_MARK(4)                                    // second barrier (+ L reading the product)
  };
  using sq_t = std::integral_constant<bool, true>;
  using mul_t = std::integral_constant<bool, false>;

  // ---- window table: tbl[k] = tbl[k-1] * x   (F holds x)
  const int nent = 1 << A.win;
  for (int k = 2; k < nent; ++k) {
    product(mul_t{}, F);
    if (role == 0) {
#pragma unroll
      for (int j = 0; j < L; ++j) tbl[((i64)k * L + j) * nlanes] = a[j];
    }
  }

  // ---- fixed-window exponentiation, per-group digits (uniform control flow: the digit only selects a table row)
  const u32* ex = A.exps + grp * A.elimbs;
  const u32 wmask = (1u << A.win) - 1u;
  auto digit = [&](int d) -> u32 {
    const int bit = d * A.win;
    const int w = bit >> 5, off = bit & 31;
    const u64 v = (u64)ex[w] | ((u64)(w + 1 < A.elimbs ? ex[w + 1] : 0u) << 32);
    return (u32)(v >> off) & wmask;
  };
  __syncthreads();                               // nobody reads F / C of the table build any more
  if (role == 0) {
    const u32 dg = digit(A.ndigits - 1);
#pragma unroll
    for (int j = 0; j < L; ++j) { a[j] = tbl[((i64)dg * L + j) * nlanes]; C[p * L + j] = a[j]; }
  }
  __syncthreads();
  if (role == 1) H.gather(a, C);
  for (int d = A.ndigits - 2; d >= 0; --d) {
    u32 y[L];
    if (role == 0) {                             // the row of this window's multiplication: requested before the squarings,
      const u32 dg = digit(d);                   // staged behind them (the load's latency is hidden, F is free by then)
#pragma unroll
      for (int j = 0; j < L; ++j) y[j] = tbl[((i64)dg * L + j) * nlanes];
    }
    for (int s = 0; s < A.win; ++s) product(sq_t{}, C);
    if (role == 0) {
#pragma unroll
      for (int j = 0; j < L; ++j) F[p * L + j] = y[j];
    }
    __syncthreads();
    product(mul_t{}, F);
  }

  // ---- epilogue (wavefront L): out of the domain by one plain Montgomery product, canonical residue
  if (role == 0) {
    M.lds = ST;
    u32 kout[L], y[L], res[L];
#pragma unroll
    for (int j = 0; j < L; ++j) kout[j] = crow[8 * PW + p * L + j];                    // theta * R = 2^(W*(Pd - hL)) mod N (bisetup_kernel)
    M.mul(y, a, kout);
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = y[j];
    M.normalize_full(res, t);
    M.cond_sub(res);
    M.cond_sub(res);
    M.store(A.out + elem * A.limbs, A.limbs, res, valid);
  }
#ifdef MX_DEV_BI_TRACE
  if (blockIdx.x == 0 && pair == 0 && lane == 0) {
    for (int k = 0; k < 5; ++k) mx_bi_trace[role * 8 + k] = trc[k];
  }
#endif
}

}  // namespace mx
