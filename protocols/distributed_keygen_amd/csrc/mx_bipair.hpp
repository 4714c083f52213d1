// Five-wavefront latency form of the N^2 pair kernel (wavefronts_per_group 4: four wavefronts carry the two passes, a fifth the
// quotient correction):  out[e] = bases[e]^exp mod N^2  with BOTH passes of every pair product
// split over two wavefronts each (tools/bipair_model.py is the column-exact model of the arithmetic in this file;
// tests/test_bipair_model.py runs it against big-integer arithmetic with every width asserted).
//
// Why.  One decrypt() (paillier_shared_key.py:92 called once at distributed_keygen.py:345-349) is a chain of ~4800 pair
// products; in the two-wavefront form (mx_powmod_n2_split.hpp) a product costs the 72 limb steps of its heavier pass on a
// wavefront that issues one instruction per ~5.3 cycles whatever it is: 12.9 ms at key_length 2048 = one gmpy2 core.  Only
// fewer STEPS per wavefront shorten that chain.  mx_bimont.hpp halves the steps of a product modulo N (Kaihara & Takagi's
// bipartite multiplication: hL least-significant-first friendly Montgomery steps on wavefront L, the other limbs
// most-significant-first with folds on wavefront H); here the same split is applied to both digit chains of the pair
// arithmetic x = theta (X0 + X1 N), theta = 2^(-W hL):
//
//   pass 1 (wavefronts AL, AH)   Z0 = BP(X0, Y0), and the EXACT quotient of that reduction in two parts:
//        AL   TL  = (X0 Y0lo + Qm N~) / 2^(W hL)            the digits of Qm are recorded (Mont::F_RECORD_Q)
//        AH   tH  =  X0 Y0hi - Qc N                         Qc = c Vq + sum dg_k cf_k:  every fold replaces v 2^(W k) by
//                                                           v (2^(W k) mod N) = v 2^(W k) - v floor(2^(W k) / N) N; the fold
//                                                           digits v_i (and the six final ones dg_k) are recorded
//        so   X0 Y0 = Z0 2^(W hL) - (u Qm - Qc 2^(W hL)) N
//   pass 2 (wavefronts BL, BH)   Z1 = theta (X0 Y1 + X1 Y0) - theta u Qm + Qc   (mod N):
//        BL   TL2 = (X0 Y1lo + X1 Y0lo + [C2' + u (2^(W hL) - 1 - Qm)] + q' N~) / 2^(W hL)  +  Qc
//        BH   tH2 =  X0 Y1hi + X1 Y0hi   folded;   Z1 = TL2 + tH2
//
// Qc (6-11 limbs of c times the ~36 fold digits + six small terms) is formed by a fifth wavefront, Q, one product behind the A
// pair — from the digits AH recorded in double-buffered LDS rows, while AH is already on the next product — and reaches BH's
// post as a row of almost-normalised limbs.  (Formed by the L wavefronts between the barriers, where they have nothing else to
// do, it was 840 cycles of every slot: a wavefront alone on its SIMD issues one instruction per five cycles, and the H
// wavefronts' posts take 450.)  BH's columns take twelve 2^58 terms in their three
// steps in a lane (two product rows, or one row with a doubled multiplier limb, plus the fold) instead of AH's six: a column
// that moves to the lane above therefore crosses as its low 30 bits and a carry word with weight 2 (29 bits and weight 1 in
// pass 1), which keeps that word below 2^32 (the model asserts both).
//
// A workgroup is the five wavefronts of ONE group set (64 / K elements).  The B pair works one product BEHIND the A pair
// (pass 2 of product s needs Qm, Qc and X0 of pass 1 of s), so in a run of products all of them are busy in every time slot; two workgroup barriers per slot (between the halves and the end of a product, as in
// mx_bimont.hpp).  Operations of the tape that are not products (LOAD / STORE / ADD: conversion and table build) drain that
// pipeline first.  The kernel runs the tape up to, not including, its last product (N2_MULC) and leaves the accumulator in
// the carry slot; the last product and the epilogue — plain passes that bring both digits below 2 N — run as a last segment
// on the two-wavefront kernel of the same geometry, whose slot layout this kernel shares (E is stored as (2^(W (Pd - hL)), 0):
// the plain product by it leaves theta S, the residue, where (1, 0) leaves S / R).
#pragma once
#include "mx_bimont.hpp"
#include "mx_powmod_n2.hpp"

namespace mx {

constexpr int BP_QROWS = 7;      // quotient rows in PowmodBiPairArgs::quot: cf_0 .. cf_5, c

struct PowmodBiPairArgs {
  const u32* bases;    // [batch][limbs2] device
  const u32* consts;   // the rows of n2_constants for R' = 2^(W h_lo): N | ONE | K1 | K2 | C' | N~ + 1 | C2'   (32-bit words)
  const u32* fold;     // [BI_ROWS][3 K] W-bit limbs, position-indexed (bisetup_kernel): 2^(W (Pd + k)) mod N, k = 0 .. 6
  const u32* quot;     // [BP_QROWS][3 K] W-bit limbs, position-indexed: floor(2^(W (Pd + k)) / N), k = 0 .. 6 (row 6 = c)
  const u32* tape;
  u32* slots;          // [nslots][2][3][nlanes]: the pair slots of the two-wavefront kernel's launch (same addressing)
  i64 batch, nlanes;
  int limbsn, limbs2, ntape;
  int nblk, pd, h_lo, ksplit;
  int nc;              // limbs of c (at most 11; informational)
  int pos_end;         // tape positions [0, pos_end) belong to this kernel (everything in front of N2_MULC)
  int e_pos;           // E = (2^(W e_pos), 0)
};

// ---- wavefront H of pass 2, and the recording wavefront H of pass 1 -----------------------------------------------------------
template <int K, int W>
struct BiHiPair : BiHi<K, W> {
  using Base = BiHi<K, W>;
  using LN = typename Base::LN;
  static constexpr int L = 3;
  static constexpr u32 MASK = Base::MASK;
  u32 limb_mask30;     // 2^30 - 1 if the lane below belongs to the number, else 0
  u32 twow;            // 2 in a VGPR the compiler cannot see through

  __device__ __forceinline__ void init2() {
    limb_mask30 = this->limb_mask ? ((1u << (W + 1)) - 1u) : 0u;
    asm volatile("" : "+v"(limb_mask30));
    twow = 2u;
    asm volatile("" : "+v"(twow));
  }

  // ---- pass 1: BiHi::half with the fold digit of every step written to V[i - h_lo] (all lanes of the group hold it; the
  // compiler pairs the stores of neighbouring steps.  Recording costs wavefront AH 270 of its 4100 cycles per product;
  // stores deferred to the end of a trip and made by one lane of the group came out 210 cycles slower)
  template <bool SQ, int I3>
  __device__ __forceinline__ void step_rec(u64 (&t)[L], const u32 (&ar)[L], const u32 (&rf)[L], u32 onev, u32 onew, u32 bi, u32* vslot) const {
    const u64 out = t[0];
    const u32 v = LN::bcast0((u32)out);
    *vslot = v;
    const u32 rl = LN::from_next_raw((u32)out) & this->limb_mask;
    const u32 rc = LN::from_next_raw((u32)(out >> W)) & this->word_mask;
    t[0] = t[1];
    t[1] = t[2] + (u64)rc * onew;
    t[2] = (u64)rl * onev;
    const u32 bi2 = SQ ? (bi << 1) : 0u;
    if constexpr (Base::template weight<SQ, I3, 0>() == 1) t[0] += (u64)ar[0] * bi;
    if constexpr (Base::template weight<SQ, I3, 0>() == 2) t[0] += (u64)ar[0] * bi2;
    if constexpr (Base::template weight<SQ, I3, 1>() == 1) t[1] += (u64)ar[1] * bi;
    if constexpr (Base::template weight<SQ, I3, 1>() == 2) t[1] += (u64)ar[1] * bi2;
    if constexpr (Base::template weight<SQ, I3, 2>() == 1) t[2] += (u64)ar[2] * bi;
    if constexpr (Base::template weight<SQ, I3, 2>() == 2) t[2] += (u64)ar[2] * bi2;
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] += (u64)rf[j] * v;
  }

  template <bool SQ>
  __device__ __forceinline__ void half_rec(u64 (&t)[L], const u32 (&ar)[L], const u32* B, int pd, int h_lo, u32* V) const {
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = 0;
    u32 al[L], rl[L], one = this->onev, one2 = this->onev;
#pragma unroll
    for (int j = 0; j < L; ++j) { al[j] = ar[j]; rl[j] = this->rf[j]; }
    auto opaque = [&]() {
#pragma unroll
      for (int j = 0; j < L; ++j) { asm volatile("" : "+v"(al[j])); asm volatile("" : "+v"(rl[j])); }
      asm volatile("" : "+v"(one));
      asm volatile("" : "+v"(one2));
    };
    // The chain opens at limb Pd: the multiplier limbs at Pd + 1 and Pd + 2 are zero in every row this kernel multiplies by
    // (a product leaves at most 4 at Pd and nothing above it, constants and the halves of x end below Pd;
    // tools/bipair_model.py asserts it), so the two steps BiHi::half spends on an empty accumulator are not run — 2 of
    // 39 at key_length 2048 — and their fold digits stay the zeros the V buffers were cleared to.
    int i = pd - 1;
    // nine limb steps per trip, the next trip's multiplier limbs fetched behind this trip's work (as BiHi::half)
    const bool trips = i - 8 >= h_lo;
    u32 nb[9];
    const u32 btop = B[pd];
    if (trips) {
#pragma unroll
      for (int k = 0; k < 9; ++k) nb[k] = B[i - k];
    }
    opaque();
    step_rec<SQ, 0>(t, al, rl, one, one2, btop, V + (pd - h_lo));
    if (trips) {
      for (; i - 8 >= h_lo; i -= 9) {
        opaque();
        u32 b[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) b[k] = nb[k];
        const int nx = i - 17 >= h_lo ? i - 9 : i;
#pragma unroll
        for (int k = 0; k < 9; ++k) nb[k] = B[nx - k];
        u32* vs = V + (i - h_lo);
        step_rec<SQ, 2>(t, al, rl, one, one2, b[0], vs);     step_rec<SQ, 1>(t, al, rl, one, one2, b[1], vs - 1); step_rec<SQ, 0>(t, al, rl, one, one2, b[2], vs - 2);
        step_rec<SQ, 2>(t, al, rl, one, one2, b[3], vs - 3); step_rec<SQ, 1>(t, al, rl, one, one2, b[4], vs - 4); step_rec<SQ, 0>(t, al, rl, one, one2, b[5], vs - 5);
        step_rec<SQ, 2>(t, al, rl, one, one2, b[6], vs - 6); step_rec<SQ, 1>(t, al, rl, one, one2, b[7], vs - 7); step_rec<SQ, 0>(t, al, rl, one, one2, b[8], vs - 8);
      }
    }
    for (; i >= h_lo; i -= 3) {
      opaque();
      const u32 b2 = B[i], b1 = B[i - 1], b0 = B[i - 2];
      u32* vs = V + (i - h_lo);
      step_rec<SQ, 2>(t, al, rl, one, one2, b2, vs);
      step_rec<SQ, 1>(t, al, rl, one, one2, b1, vs - 1);
      step_rec<SQ, 0>(t, al, rl, one, one2, b0, vs - 2);
    }
  }

  // BiHi::pre with the five digits it folds (positions Pd+1 .. Pd+5) and its own part of the sixth written to DG[1..5], DG[0]
  __device__ __forceinline__ u32 pre_rec(u64 (&t)[L], u32* DG) const {
    u32 r[L];
    this->template sweep<true>(r, t);
    u32 dg[6];
    dg[5] = LN::bcast0(r[0]); dg[4] = LN::bcast0(r[1]); dg[3] = LN::bcast0(r[2]);
    dg[2] = LN::bcast0(LN::from_next_raw(r[0])); dg[1] = LN::bcast0(LN::from_next_raw(r[1])); dg[0] = LN::bcast0(LN::from_next_raw(r[2]));
    if (DG) {
#pragma unroll
      for (int k = 0; k < 6; ++k) DG[k] = dg[k];
    }
#pragma unroll
    for (int j = 0; j < L; ++j) {
      u64 s = (u64)(r[j] & this->low_keep);
#pragma unroll
      for (int k = 1; k < 6; ++k) {
        u32 f = this->fin[k][j];
        asm volatile("" : "+v"(f));
        s += (u64)f * dg[k];
      }
      t[j] = s;
    }
    return dg[0];
  }

  // BiHi::post with a second row beside the L half's (pass 2: the quotient correction Qc, wavefront Q's row): both are read
  // at the same words and under the same masks, the limb at Pd of their sum is the digit the last fold takes
  __device__ __forceinline__ void post_sum(u64 (&t)[L], u32 dg0, const u32* TL, const u32* QR, u32* C, u32 (&ar)[L], int pd) const {
    u32 tl[L], qr[L];
#pragma unroll
    for (int j = 0; j < L; ++j) { tl[j] = TL[this->addr[j]]; qr[j] = QR[this->addr[j]]; }
    const u32 d = dg0 + TL[pd] + QR[pd];        // every lane reads the same words
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] += (u64)((tl[j] + qr[j]) & this->data_ok[j]);
#pragma unroll
    for (int j = 0; j < L; ++j) {
      u32 f = this->fin[0][j];
      asm volatile("" : "+v"(f));
      t[j] += (u64)f * d;
    }
    u32 r[L];
    this->template sweep<false>(r, t);
#pragma unroll
    for (int j = 0; j < L; ++j) {
      ar[j] = r[j] & this->pos_ok[j];
      C[this->waddr[j]] = r[j];
    }
  }

  // ---- pass 2: one or two product rows, a column crosses to the lane above as 30 bits + a carry word of weight 2
  template <bool TWO, bool DBL>
  __device__ __forceinline__ void step2(u64 (&t)[L], const u32 (&ar)[L], const u32 (&cr)[L], const u32 (&rf)[L], u32 onev, u32 two, u32 bi, u32 di) const {
    const u64 out = t[0];
    const u32 v = LN::bcast0((u32)out);
    const u32 rl = LN::from_next_raw((u32)out) & limb_mask30;
    const u32 rc = LN::from_next_raw((u32)(out >> (W + 1))) & this->word_mask;
    t[0] = t[1];
    t[1] = t[2] + (u64)rc * two;
    t[2] = (u64)rl * onev;
    const u32 b = DBL ? (bi << 1) : bi;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      t[j] += (u64)ar[j] * b;
      if constexpr (TWO) t[j] += (u64)cr[j] * di;
      t[j] += (u64)rf[j] * v;
    }
  }

  template <bool TWO, bool DBL>
  __device__ __forceinline__ void half2(u64 (&t)[L], const u32 (&ar)[L], const u32 (&cr)[L], const u32* B, const u32* D, int pd, int h_lo) const {
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = 0;
    u32 al[L], cl[L], rl[L], one = this->onev, two = twow;
#pragma unroll
    for (int j = 0; j < L; ++j) { al[j] = ar[j]; cl[j] = cr[j]; rl[j] = this->rf[j]; }
    auto opaque = [&]() {
#pragma unroll
      for (int j = 0; j < L; ++j) { asm volatile("" : "+v"(al[j])); asm volatile("" : "+v"(rl[j])); if constexpr (TWO) asm volatile("" : "+v"(cl[j])); }
      asm volatile("" : "+v"(one));
      asm volatile("" : "+v"(two));
    };
    int i = pd - 1;                 // (the chain opens at limb Pd, as in half_rec)
    const bool trips = i - 8 >= h_lo;
    u32 nb[9], nd[9];
    const u32 btop = B[pd], dtop = TWO ? D[pd] : 0u;
    if (trips) {
#pragma unroll
      for (int k = 0; k < 9; ++k) { nb[k] = B[i - k]; nd[k] = TWO ? D[i - k] : 0u; }
    }
    opaque();
    step2<TWO, DBL>(t, al, cl, rl, one, two, btop, dtop);
    if (trips) {
      for (; i - 8 >= h_lo; i -= 9) {
        opaque();
        u32 b[9], d[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { b[k] = nb[k]; d[k] = nd[k]; }
        const int nx = i - 17 >= h_lo ? i - 9 : i;
#pragma unroll
        for (int k = 0; k < 9; ++k) { nb[k] = B[nx - k]; nd[k] = TWO ? D[nx - k] : 0u; }
#pragma unroll
        for (int k = 0; k < 9; ++k) step2<TWO, DBL>(t, al, cl, rl, one, two, b[k], d[k]);
      }
    }
    for (; i >= h_lo; i -= 3) {
      opaque();
      const u32 b2 = B[i], b1 = B[i - 1], b0 = B[i - 2];
      const u32 d2 = TWO ? D[i] : 0u, d1 = TWO ? D[i - 1] : 0u, d0 = TWO ? D[i - 2] : 0u;
      step2<TWO, DBL>(t, al, cl, rl, one, two, b2, d2);
      step2<TWO, DBL>(t, al, cl, rl, one, two, b1, d1);
      step2<TWO, DBL>(t, al, cl, rl, one, two, b0, d0);
    }
  }
};

// ---- the kernel ------------------------------------------------------------------------------------------------------------------
#ifdef MX_DEV_BP_TRACE          // developer builds (tools/bp_phase_probe.py): shader-clock cycles per phase and role, workgroup 0
__device__ u64 mx_bp_trace[25];
#define MX_BP_MARK(k) { const u64 now_ = __builtin_readcyclecounter(); trc[k] += now_ - mark_; mark_ = now_; }
#else
#define MX_BP_MARK(k)
#endif
constexpr int BP_THREADS = 320;   // five wavefronts: AL AH BL BH and Q (the quotient correction, one product behind the A pair)
template <int K, int W>
__global__ void __launch_bounds__(BP_THREADS) powmod_n2_bipair_kernel(PowmodBiPairArgs A) {
  constexpr int L = 3, PW = L * K, GPW = 64 / K;
  using M_t = Mont<K, L, W, true, false>;
  using H_t = BiHiPair<K, W>;
  constexpr int ROW = PW + 4;
  static_assert(ROW == M_t::LDS_D, "two rows side by side are the two multipliers of Mont::mulx<F_TWO>");
  // per group of lanes: CA[2] CB F[2][2] TLA TLB QM[2] V[2] DG[2] QC[2] ST_A ST_B.  A V buffer is VROW words: the fold digits from
  // word VOFF on, zeros in front of and behind them (wavefront AL reads V[pos - i] for every position of its lanes without
  // a condition: a conditional read per limb of c was a wait per read, 2400 cycles per product)
  constexpr int VROW = 2 * ROW, VOFF = 16;
  constexpr int O_CA = 0, O_CB = 2 * ROW, O_F = 3 * ROW, O_TLA = 7 * ROW, O_TLB = 8 * ROW, O_QM = 9 * ROW, O_V = 11 * ROW,
                O_DG = O_V + 2 * VROW, O_QC = O_DG + 16, O_STA = O_QC + 2 * ROW, O_STB = O_STA + M_t::LDS_WORDS, O_STQ = O_STB + M_t::LDS_WORDS,
                GROUP_WORDS = O_STQ + M_t::LDS_WORDS;
  constexpr int NC = 11;         // limbs of c, at most: c <= 2^(W (Pd + 6) - bits + 1) < 2^(6 W + 123)
  static_assert(VOFF >= NC + 2 && VOFF + 3 * K + 2 <= VROW, "every V[pos - i] lies inside the buffer");
  extern __shared__ u32 smem[];
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // 0 AL, 1 AH, 2 BL, 3 BH, 4 Q
  const bool is_l = role == 0 || role == 2;                                     // Mont-layout wavefronts of the pairs
  const bool is_q = role == 4;
  const bool is_a = role < 2;
  const int lane = threadIdx.x & 63;
  const int gw = lane / K;
  const int p = lane & (K - 1);
  u32* G = smem + gw * GROUP_WORDS;
  u32* CA = G + O_CA; u32* CB = G + O_CB; u32* F = G + O_F; u32* TLA = G + O_TLA; u32* TLB = G + O_TLB;
  u32* QM = G + O_QM; u32* V = G + O_V; u32* DG = G + O_DG; u32* QC = G + O_QC;
  u32* ST = G + (role == 0 ? O_STA : (role == 4 ? O_STQ : O_STB));     // (Mont::load stages through it: one per Mont wavefront)

  const i64 slot_id = (i64)blockIdx.x;
  const i64 elem_raw = slot_id * GPW + gw;
  const i64 elem = elem_raw < A.batch ? elem_raw : A.batch - 1;
  u32* slots = A.slots + (slot_id * 64 + lane);
  const int dig = role >> 1;                                                    // the digit an L wavefront owns in the slots
  auto slot_at = [&](int sl, int j) __attribute__((always_inline)) -> u32& { return slots[(((i64)sl * 2 + dig) * L + j) * A.nlanes]; };
  const int nblk_lo = A.h_lo / L;

  M_t M;
  H_t H;
  M.init(ST, A.nblk);
  H.init(A.pd);
  H.init2();
  // zero every row once (positions beyond a number's top are read as zero limbs)
  for (int k = threadIdx.x; k < GPW * GROUP_WORDS; k += BP_THREADS) smem[k] = 0;
  __syncthreads();
  u32 c2p[L];                    // BL: C2' (this lane's limbs)
  u32 cfr[6][L];                 // Q: the final folds' quotients (this lane's limbs)
  u32 climb[11];                 // Q: the limbs of c (the same in every lane; fetched once — a scalar load per limb and product
                                 // cost 2700 cycles per slot, profiles/r06_bp_phase_probe.txt)
  // (The L wavefronts AND Q load all of these, each staging through an LDS area of its own: with the modulus left undefined in
  // Q the compiler treated the L wavefronts' limbs of N~ as 64-bit values — a third more multiply-adds in every block of
  // their products, 3700 -> 5070 cycles per product — and with constants in its place the slots came out 5 % slower.)
  if (is_l || is_q) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int j = 0; j < L; ++j) cfr[k][j] = A.quot[k * PW + p * L + j];
    }
#pragma unroll
    for (int i = 0; i < 11; ++i) climb[i] = A.quot[6 * PW + i];
    M.load(M.n, A.consts, A.limbsn);
    M.setup_modulus();
    M.load(M.nf, A.consts + 8 * A.limbsn, A.limbsn + 1);
    M.setup_friendly();
    M.load(c2p, A.consts + 8 * A.limbsn + (A.limbsn + 1), A.limbsn + 1);
  } else {
    H.gather(H.rf, A.fold + 6 * PW);
#pragma unroll
    for (int k = 0; k < 6; ++k) H.gather(H.fin[k], A.fold + k * PW);
  }

  // ---- prologue: constant pairs and the two halves of x into their slots (as mx_powmod_n2_split.hpp, L wavefronts only)
  if (is_l) {
    u32 v[L];
    const int rows[3][3] = {{N2_SLOT_K1, 3, 4}, {N2_SLOT_K2, 5, 6}, {N2_SLOT_ONE, 1, 2}};
    for (int r = 0; r < 3; ++r) {
      M.load(v, A.consts + (i64)rows[r][1 + dig] * A.limbsn, A.limbsn);
#pragma unroll
      for (int j = 0; j < L; ++j) slot_at(rows[r][0], j) = v[j];
    }
#pragma unroll
    for (int j = 0; j < L; ++j) slot_at(N2_SLOT_E, j) = (dig == 0 && p * L + j == A.e_pos) ? 1u : 0u;
    if (dig == 0) {
      M_t::sync();
      const u32* src = A.bases + elem * A.limbs2;
      for (int k = p; k < M_t::LDS_WORDS; k += K) ST[k] = (k < A.limbs2) ? src[k] : 0u;
      M_t::sync();
#pragma unroll
      for (int j = 0; j < L; ++j) {
        const int bit = W * (p * L + j);
        const int room = A.ksplit - bit;
        const u32 lo = room <= 0 ? 0u : extract_field(ST, bit, room < W ? room : W);
        const int hbit = A.ksplit + bit;
        const u32 hi = (hbit + W + 32 <= 32 * M_t::LDS_WORDS) ? extract_field(ST, hbit, W) : 0u;
        slot_at(N2_SLOT_LO, j) = lo;
        slot_at(N2_SLOT_HI, j) = hi;
      }
    } else {
#pragma unroll
      for (int j = 0; j < L; ++j) { slot_at(N2_SLOT_LO, j) = 0; slot_at(N2_SLOT_HI, j) = 0; }
    }
  }
  __syncthreads();

  // ---- one time slot: the A pair runs pass 1 of product `pa`, the B pair pass 2 of product `pb` (kind 0: none, 1: squaring,
  // 2: multiplication by the pair staged in F[f]); `nx` / `nxf`: a multiplication that follows `pa` directly — its table pair
  // is fetched during this slot and staged in F[nxf] before the slot's last barrier.
#ifdef MX_DEV_BP_TRACE
  u64 trc[4] = {0, 0, 0, 0};
  const u64 trc_begin = __builtin_readcyclecounter();     // (behind the prologue)
#endif
  int ca = 0;          // CA[ca]: the accumulator's first digit (the A pair's operand), CA[ca ^ 1] receives its product
  int qa = 0;          // QM[qa] receives the quotient digits of the A pair's product
  // ---- what each role does in a time slot (phase 1: its half of a product; phase 2, H wavefronts: the post).  `ca`, `qa` as
  // above; ca_b, qb: the same two of the product the B pair works on (one slot older).
  // (always_inline: called from two places each, they were left as functions otherwise — their captures, LDS pointers among
  // them, read back from a closure in scratch memory through flat loads: 29.8 ms instead of 9.9)
  auto ah_half = [&](int pa, int fa, u64 (&t)[L], u32 (&a)[L], bool fetch) __attribute__((always_inline)) -> u32 {
    if (fetch) H.gather(a, CA + ca * ROW);          // (in a window AH keeps X0 = its own last post's Z0 in registers)
    if (pa == 1) {
      H.template half_rec<true>(t, a, CA + ca * ROW, A.pd, A.h_lo, V + qa * VROW + VOFF);
    } else {
      H.template half_rec<false>(t, a, F + (fa * 2 + 1) * ROW, A.pd, A.h_lo, V + qa * VROW + VOFF);
    }
    return H.pre_rec(t, DG + qa * 8);
  };
  auto bh_half = [&](int pb, int fb, int ca_b, u64 (&t)[L]) __attribute__((always_inline)) -> u32 {
    u32 x0[L], x1[L] = {0u, 0u, 0u};
    H.gather(x0, CA + ca_b * ROW);
    if (pb == 1) {
      H.template half2<false, true>(t, x0, x0, CB, CB, A.pd, A.h_lo);          // 2 X0 X1: multiplier limb doubled
    } else {
      H.gather(x1, CB);
      H.template half2<true, false>(t, x0, x1, F + fb * 2 * ROW, F + (fb * 2 + 1) * ROW, A.pd, A.h_lo);
    }
    return H.pre_rec(t, nullptr);
  };
  auto al_half = [&](int pa, int fa) __attribute__((always_inline)) {
    u32 a[L], r[L], q[L];
#pragma unroll
    for (int j = 0; j < L; ++j) a[j] = CA[ca * ROW + p * L + j];
    M.lds = pa == 1 ? CA + ca * ROW : F + (fa * 2 + 1) * ROW;                // squaring: X0 itself; multiplication: Y0
    if (pa == 1) {
      M.template mulx<M_t::F_FRIENDLY | M_t::F_STAGED | M_t::F_RECORD_Q | M_t::F_SQUARE>(r, a, a, a, a, a, q, nullptr, nblk_lo);
    } else {
      M.template mulx<M_t::F_FRIENDLY | M_t::F_STAGED | M_t::F_RECORD_Q>(r, a, a, a, a, a, q, nullptr, nblk_lo);
    }
#pragma unroll
    for (int j = 0; j < L; ++j) {
      TLA[p * L + j] = r[j];
      QM[qa * ROW + p * L + j] = q[j];
      if (p * L + j == A.pd) DG[qa * 8 + 6] = r[j];          // (TLA is this wavefront's again before Q reads it)
    }
  };
  auto bl_half = [&](int pb, int fb, int ca_b, int qb) __attribute__((always_inline)) {
    u32 x0[L], x1[L], r[L], qq[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
      x0[j] = CA[ca_b * ROW + p * L + j];
      x1[j] = CB[p * L + j];
      const u32 qd = QM[qb * ROW + p * L + j];
      qq[j] = (p < nblk_lo) ? (M_t::MASK - qd) : 0u;                          // u (2^(W hL) - 1 - Qm), limb-wise
    }
    if (pb == 1) {
      M.lds = ST;
      M.template mulx<M_t::F_INIT | M_t::F_INITQ | M_t::F_BDOUBLE | M_t::F_FRIENDLY>(r, x0, x1, x0, x1, c2p, nullptr, nullptr, nblk_lo, qq);
    } else {
      M.lds = F + fb * 2 * ROW;                                               // b = Y1 (with X0), d = Y0 (with X1)
      M.template mulx<M_t::F_TWO | M_t::F_INIT | M_t::F_INITQ | M_t::F_STAGED | M_t::F_FRIENDLY>(r, x0, x0, x1, x1, c2p, nullptr, nullptr, nblk_lo, qq);
    }
#pragma unroll
    for (int j = 0; j < L; ++j) TLB[p * L + j] = r[j];
  };
  auto q_row = [&](int qb) __attribute__((always_inline)) {
    // Qc = c * Vq + sum dg_k cf_k of the product whose pass 2 the B pair runs in this slot, for this lane's positions: from
    // the digits its pass 1 recorded one slot ago (V, DG: double buffers) into the row QC that wavefront BH adds in its post.
    // (Rounds of this work by the L wavefronts between the barriers cost 840 cycles of every slot, a lone wavefront issuing
    // an instruction every five cycles; a wavefront of its own has a whole phase for them.)
    const u32* Vb = V + qb * VROW + VOFF + p * L;            // V[pos] of this lane's first position
    // (loop-invariant 32-bit multiplicands: opaque IN PLACE once per product, or the compiler keeps their zero-extensions
    // in register pairs; the reads are issued as one batch — a scheduling barrier keeps them in front of their uses)
#pragma unroll
    for (int i = 0; i < NC; ++i) asm volatile("" : "+v"(climb[i]));
    u32 vw[NC + 2], dgv[6];                                  // V[base + 2 - k], k = 0 .. NC + 1
#pragma unroll
    for (int k = 0; k < NC + 2; ++k) vw[k] = Vb[2 - k];
#pragma unroll
    for (int k = 0; k < 6; ++k) dgv[k] = DG[qb * 8 + k];
    const u32 tl_pd = DG[qb * 8 + 6];
    __builtin_amdgcn_sched_barrier(0);
    dgv[0] += tl_pd;                                         // dg_0: wavefront AH's own part + wavefront AL's limb at Pd
    // (six independent chains — two per column: a lone wavefront waits for every dependent multiply-accumulate)
    u64 ev[L] = {0, 0, 0}, od[L] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int j = 0; j < L; ++j) {
        if (k & 1) od[j] += (u64)cfr[k][j] * dgv[k];
        else ev[j] += (u64)cfr[k][j] * dgv[k];
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {                                               // (limbs of c beyond its length are zero)
#pragma unroll
      for (int j = 0; j < L; ++j) {
        if (i & 1) od[j] += (u64)climb[i] * vw[2 - j + i];                       // V[base + j - i]
        else ev[j] += (u64)climb[i] * vw[2 - j + i];
      }
    }
    u64 qc[L];
#pragma unroll
    for (int j = 0; j < L; ++j) qc[j] = ev[j] + od[j];
    u32 r[L];
    M.normalize_weak(r, qc);
#pragma unroll
    for (int j = 0; j < L; ++j) QC[p * L + j] = r[j];
  };

  auto run_slot = [&](int pa, int fa, int pb, int fb, int ca_b, int qb, int nx_slot, int nxf) __attribute__((always_inline)) {
    u64 t[L];
    u32 dg0 = 0;
    u32 ynext[L];
#ifdef MX_DEV_BP_TRACE
    u64 mark_ = __builtin_readcyclecounter();
#endif
    if (is_l && nx_slot >= 0) {
#pragma unroll
      for (int j = 0; j < L; ++j) ynext[j] = slot_at(nx_slot, j);
    }
    // ---- phase 1
    if (role == 1 && pa) {                   // (the H wavefronts first: theirs is the longest path of a slot)
      u32 a[L];
      dg0 = ah_half(pa, fa, t, a, true);
    } else if (role == 3 && pb) {
      dg0 = bh_half(pb, fb, ca_b, t);
    } else if (role == 0 && pa) {
      al_half(pa, fa);
    } else if (role == 2 && pb) {
      bl_half(pb, fb, ca_b, qb);
    } else if (is_q && pb) {
      q_row(qb);
    }
    MX_BP_MARK(0)                                    // phase 1: this role's half (+ pre)
    __syncthreads();
    MX_BP_MARK(1)                                    // waiting for the other halves
    // ---- phase 2
    if (role == 1 && pa) {
      u32 a[L];
      H.post(t, dg0, TLA, CA + (ca ^ 1) * ROW, a, A.pd);
    } else if (role == 3 && pb) {
      u32 a[L];
      H.post_sum(t, dg0, TLB, QC, CB, a, A.pd);
    }
    if (is_l && nx_slot >= 0) {
      // the table pair of the multiplication that follows: digit 0 -> F[nxf][1] (Y0), digit 1 -> F[nxf][0] (Y1)
#pragma unroll
      for (int j = 0; j < L; ++j) F[(nxf * 2 + (dig == 0 ? 1 : 0)) * ROW + p * L + j] = ynext[j];
    }
    MX_BP_MARK(2)                                    // phase 2: post (H wavefronts), staging (L wavefronts)
    __syncthreads();
    MX_BP_MARK(3)                                    // second barrier
  };

  // ---- a whole window of the exponent — w squarings and the multiplication by table pair T behind them — with every wavefront
  // in a loop of its own: the same halves, posts and two barriers per slot as run_slot, but no tape decode, no role dispatch and
  // no look-ahead bookkeeping between two products (every instruction of that control path was five cycles on the wavefront
  // the others wait for: 10.5 -> 9.9 ms for the inside of the runs alone).  Slot s < w: the A pair squares; slot w: it multiplies
  // by F[f], which the L wavefronts fill from the slots during slot w - 1.  The B pair is one product behind: in slot 0 it
  // finishes the product in front of the window (kind pb0 — 0: none —, F buffer fb0), then the squarings.
  auto run_window = [&](int w, int pb0, int fb0, int T, int f) __attribute__((always_inline)) {
#ifdef MX_DEV_BP_TRACE
    u64 mark_ = __builtin_readcyclecounter();       // (the probe's four intervals per slot, as in run_slot)
#endif
    if (role == 1) {
      u32 a[L];
      H.gather(a, CA + ca * ROW);
      for (int s = 0; s <= w; ++s) {
        u64 t[L];
        const u32 dg0 = ah_half(s < w ? 1 : 2, f, t, a, false);
        MX_BP_MARK(0)
        __syncthreads();
        MX_BP_MARK(1)
        H.post(t, dg0, TLA, CA + (ca ^ 1) * ROW, a, A.pd);
        MX_BP_MARK(2)
        __syncthreads();
        MX_BP_MARK(3)
        ca ^= 1; qa ^= 1;
      }
    } else if (role == 3) {
      for (int s = 0; s <= w; ++s) {
        const int pb = s == 0 ? pb0 : 1;
        u64 t[L];
        u32 dg0 = 0;
        if (pb) dg0 = bh_half(pb, fb0, ca ^ 1, t);
        MX_BP_MARK(0)
        __syncthreads();
        MX_BP_MARK(1)
        if (pb) {
          u32 a[L];
          H.post_sum(t, dg0, TLB, QC, CB, a, A.pd);
        }
        MX_BP_MARK(2)
        __syncthreads();
        MX_BP_MARK(3)
        ca ^= 1; qa ^= 1;
      }
    } else if (role == 0 || role == 2) {
      for (int s = 0; s <= w; ++s) {
        u32 ynext[L];
        if (s == w - 1) {
#pragma unroll
          for (int j = 0; j < L; ++j) ynext[j] = slot_at(T, j);
        }
        if (role == 0) {
          al_half(s < w ? 1 : 2, f);
        } else {
          const int pb = s == 0 ? pb0 : 1;
          if (pb) bl_half(pb, fb0, ca ^ 1, qa ^ 1);
        }
        MX_BP_MARK(0)
        __syncthreads();
        MX_BP_MARK(1)
        if (s == w - 1) {                  // digit 0 -> F[f][1] (Y0), digit 1 -> F[f][0] (Y1)
#pragma unroll
          for (int j = 0; j < L; ++j) F[(f * 2 + (dig == 0 ? 1 : 0)) * ROW + p * L + j] = ynext[j];
        }
        MX_BP_MARK(2)
        __syncthreads();
        MX_BP_MARK(3)
        ca ^= 1; qa ^= 1;
      }
    } else {
      for (int s = 0; s <= w; ++s) {
        if (s > 0 || pb0) q_row(qa ^ 1);
        MX_BP_MARK(0)
        __syncthreads();
        MX_BP_MARK(1)
        MX_BP_MARK(2)
        __syncthreads();
        MX_BP_MARK(3)
        ca ^= 1; qa ^= 1;
      }
    }
  };

  // ---- the tape: ONE call site of run_slot (the body holds five roles' code paths for both kinds of product), driven by a
  // small state machine over the tape words; whole windows go through run_window
  const tape_ptr_t tape = (tape_ptr_t)A.tape;
  int pend = 0, pend_f = 0, pend_ca = 0, pend_q = 0;      // the product whose pass 2 is outstanding
  int fcur = 0;                                           // F buffer of the next multiplication
  int staged = -1;                                        // table slot already staged in F[fcur] by a look-ahead
  int k = 0, pos = 0;
  // the tape word at k and the one behind it, fetched when k moved last — a slot or more before they are looked at (a scalar
  // load at the point of use was 200 cycles in which none of the five wavefronts had anything to issue, 1200 times per launch)
  constexpr u32 TAPE_END = N2_MULC << 28;
  u32 w0 = A.ntape > 0 ? tape[0] : TAPE_END, w1 = A.ntape > 1 ? tape[1] : TAPE_END;
  auto next_word = [&]() __attribute__((always_inline)) {
    ++k;
    w0 = w1;
    w1 = k + 1 < A.ntape ? tape[k + 1] : TAPE_END;
  };
  int rem = 0, run_nx = -1;                               // squarings left in the current run; the multiplication behind it
  bool done = false;
  while (!done) {
    int kind = 0, f = 0, nx = -1;
    if (rem > 0) {
      kind = 1;
      --rem;
      if (rem == 0) { nx = run_nx; if (nx >= 0) staged = nx; }
    } else {
      u32 op = N2_MULC;
      int arg = 0;
      if (k < A.ntape && pos < A.pos_end) {
        const u32 word = w0;
        op = word >> 28;
        arg = (int)(word & 0x0FFFFFFFu);
      }
      if (op == N2_MULC) {                                 // the last product is another kernel's segment: finish
        if (!pend) break;
        done = true;                                       // (kind 0: the outstanding pass 2)
      } else if (op == N2_SQR) {
        const int hi = pos + arg < A.pos_end ? pos + arg : A.pos_end;
        rem = hi - pos;
        run_nx = -1;
        if (pos + arg <= A.pos_end && k + 1 < A.ntape) {
          if ((w1 >> 28) == N2_MUL) run_nx = (int)(w1 & 0x0FFFFFFFu);
        }
        pos += arg;
        next_word();
        if (run_nx >= 0 && rem >= 2 && pos < A.pos_end) {   // a whole window: the run and the multiplication behind it (now at k)
          const int fw = fcur;
          fcur ^= 1;
          run_window(rem, pend, pend_f, run_nx, fw);
          next_word();
          rem = 0;
          staged = -1;
          pend = 2; pend_f = fw; pend_ca = ca ^ 1; pend_q = qa ^ 1;
        }
        continue;
      } else if (op == N2_MUL) {
        if (staged != arg) {
          // not staged by a look-ahead (a multiplication behind a LOAD / STORE): stage it now, with a barrier of its own
          if (is_l) {
#pragma unroll
            for (int j = 0; j < L; ++j) F[(fcur * 2 + (dig == 0 ? 1 : 0)) * ROW + p * L + j] = slot_at(arg, j);
          }
          __syncthreads();
        }
        staged = -1;
        kind = 2;
        f = fcur;
        fcur ^= 1;
        next_word();
      } else if (pend) {
        // LOAD / STORE / ADD act on the complete pair: the outstanding pass 2 first (kind 0), the operation next time round
      } else {
        if (is_l) {
          u32* row = dig == 0 ? CA + ca * ROW : CB;
          if (op == N2_STORE) {
#pragma unroll
            for (int j = 0; j < L; ++j) slot_at(arg, j) = row[p * L + j];
          } else if (op == N2_LOAD) {
#pragma unroll
            for (int j = 0; j < L; ++j) row[p * L + j] = slot_at(arg, j);
          } else {   // N2_ADD
            u32 x[L], fv[L];
#pragma unroll
            for (int j = 0; j < L; ++j) { x[j] = row[p * L + j]; fv[j] = slot_at(arg, j); }
            M.add(x, x, fv);
#pragma unroll
            for (int j = 0; j < L; ++j) row[p * L + j] = x[j];
          }
        }
        __syncthreads();
        next_word();
        continue;
      }
    }
    run_slot(kind, f, pend, pend_f, pend_ca, pend_q, nx, fcur);
    pend = kind; pend_f = f; pend_ca = ca; pend_q = qa;
    if (kind) { ca ^= 1; qa ^= 1; }
  }
  if (is_l) {
    const u32* row = dig == 0 ? CA + ca * ROW : CB;
#pragma unroll
    for (int j = 0; j < L; ++j) slot_at(N2_SLOT_CARRY, j) = row[p * L + j];
  }
#ifdef MX_DEV_BP_TRACE
  if (blockIdx.x == 0 && lane == 0) {
    for (int k = 0; k < 4; ++k) mx_bp_trace[role * 4 + k] = trc[k];
    mx_bp_trace[20 + role] = __builtin_readcyclecounter() - trc_begin;
  }
#endif
}

template <int K, int W>
constexpr size_t powmod_n2_bipair_lds_bytes() {
  using M_t = Mont<K, 3, W, true, false>;
  constexpr int ROW = 3 * K + 4;
  return (size_t)(64 / K) * (11 * ROW + 4 * ROW + 16 + 2 * ROW + 3 * M_t::LDS_WORDS) * 4;
}

}  // namespace mx
