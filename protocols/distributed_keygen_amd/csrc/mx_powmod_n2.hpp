// Batched modular exponentiation modulo a SQUARE:  out[e] = bases[e]^exp mod N^2,  one (N, exp) per
// launch — the partial decryption of threshold Paillier, pow_mod(c, exp, n_square)
// (paillier_shared_key.py:92 looped at distributed_keygen.py:463-466).
//
// Arithmetic modulo N^2 with operations of the size of N only.  Let R = 2^(W*L*nblk) >= 16 N be the
// Montgomery radix for N and rho = R^-1 mod N^2.  A residue x is held as a pair (X0, X1) of lazy
// residues modulo N with
//                         x = rho * (X0 + X1 * N)   (mod N^2).
// For two such pairs, x*y = rho^2 * (X0*Y0 + (X0*Y1 + X1*Y0) * N) because N^2 = 0.  A Montgomery
// pass of X0*Y0 modulo N gives t0 and the quotient Q with  X0*Y0 = t0*R - Q*N  EXACTLY (integers), so
//       x*y = rho * ( t0 + rho * (X0*Y1 + X1*Y0 - Q) * N )
// and, since only  rho * (...) mod N  matters in front of N, a second Montgomery pass gives
//       Z0 = t0,     Z1 = REDC_N( X0*Y1 + X1*Y0 + (C - Q) ),        C = N * ceil(R / N)  (C = 0 mod N, C >= Q)
// i.e. two half-size passes (3 half-size products + 2 half-size reductions; 2 + 2 for a squaring, one
// of them symmetric) instead of one full-size product and reduction: ~1.8x fewer multiply-accumulates
// than Montgomery modulo N^2 directly, with HALF the lanes per element.  C - Q is formed limb-wise as
// C' + (R - 1 - Q), C' = C - R + 1, i.e. (MASK - q_i) + c'_i: no borrow anywhere.
//
// Conversions (once per exponentiation): x = x_lo + 2^k x_hi is brought into pair form by two pair
// products with the constants that represent R and 2^k R; the result pair is multiplied by (1, 0)
// (which represents rho) so that its plain N-adic value Y0 + Y1*N IS the residue, both digits are
// reduced to [0, N), and Y0 + Y1*N is formed by a plain (reduction-free) product whose low limbs
// leave through the group's lane 0.
#pragma once
#include "mx_mont.hpp"

namespace mx {

// The kernel is an interpreter of a short "tape" built by the host (the same for every lane, so
// control flow is uniform): conversion into pair form, the table of odd powers and the
// sliding-window exponentiation are all sequences of five operations on one pair register `acc`
// and pair slots in device memory.  This keeps ONE squaring and ONE multiplication call site in
// the kernel (each is two inlined Montgomery passes), which is what bounds code size and registers.
// N2_MULC: a multiplication whose digits must come out below 2N (the last one, by (1, 0)); kernels without the friendly
// passes treat it as N2_MUL
enum : u32 { N2_SQR = 0, N2_MUL = 1, N2_ADD = 2, N2_LOAD = 3, N2_STORE = 4, N2_MULC = 5 };
// tape word = (op << 28) | argument   (argument: repeat count for SQR, slot index otherwise)
// slots: 0 K1 (represents R), 1 K2 (represents 2^k R), 2 E = (1, 0), 3 ONE, 4 (x_lo, 0), 5 (x_hi, 0),
//        6 scratch, 7 x^2, 8.. odd powers x^(2k+1)
constexpr int N2_SLOT_K1 = 0, N2_SLOT_K2 = 1, N2_SLOT_E = 2, N2_SLOT_ONE = 3, N2_SLOT_LO = 4, N2_SLOT_HI = 5,
              N2_SLOT_TMP = 6, N2_SLOT_SQ = 7, N2_SLOT_TABLE = 8;
// the accumulator between two segments of one exponentiation: x_hi's slot, which only the conversion
// at tape position 0 reads
constexpr int N2_SLOT_CARRY = N2_SLOT_HI;

// The tape is written once per plan and is the same for every wavefront: the kernels read it through the constant
// address space (a scalar load through the scalar cache per word instead of a vector load + readfirstlane) and leave
// the tape loop at the end of their segment.  Measured against the vector form: headline +0.5 %, small time-sliced
// launches 2-4 % shorter, everything else unchanged.  (Loading one word AHEAD was worse: the outstanding scalar load
// turns the first LDS wait of every operation into a full drain — a lone decryption 12.9 -> 13.3 ms.)
typedef const __attribute__((address_space(4))) u32* tape_ptr_t;

// LDS of one workgroup (one wavefront): per group the Montgomery scratch (which the input row and
// the output limbs reuse), plus ONE copy of C' for all groups.  Small enough that the register
// file, not LDS, bounds occupancy: 10 KB per wavefront for <4,18>, 5 KB for <8,9>.
template <int K, int L>
constexpr size_t powmod_n2_lds_bytes(bool friendly = false) { return ((size_t)(64 / K) * (2 * K * L + 8) + (size_t)(friendly ? 2 : 1) * K * L) * 4; }

struct PowmodN2Args {
  const u32* bases;   // [batch][limbs2] device
  u32* out;           // [batch][limbs2] device
  const u32* consts;  // [8][limbsn] device: N, ONE0, ONE1, K1_0, K1_1, K2_0, K2_1, C'; then [2][limbsn + 1]: N~ + 1, C2' (friendly passes)
  const u32* tape;    // [ntape] device
  u32* slots;         // [nslots][2][L][nlanes] device
  i64 batch;
  int limbsn, limbs2;
  int ntape, nblk;
  int ksplit;         // x = x_lo + 2^ksplit * x_hi, ksplit = bits(N) - 1
  // One exponentiation may be run as several consecutive launches ("segments"), each executing the
  // part of the tape whose position — the number of squarings executed so far — lies in
  // [pos_begin, pos_end); the accumulator travels between them through the scratch slot.  A wavefront
  // then lives 1/segments as long, which is the granularity at which a burst of launches drains.
  int pos_begin, pos_end;
  int first, last;    // first segment: input conversion prologue; last segment: output epilogue
  int friendly;       // host: the modulus leaves room for the friendly-modulus passes (the 9-limb two-wavefront kernels pick their instance by it)
  // persistent (time-sliced) form of the two-wavefront kernel, mx_powmod_n2_split.hpp: scheduling words in device
  // memory (ticket, finished segments per group), groups of elements, segments per group, squarings of the tape
  u32* sched;
  int sched_groups, sched_segments, sched_n_sqr;
};

// FR ("friendly"): the passes reduce modulo N~ = u * N = -1 mod 2^W instead of N (Mont::F_FRIENDLY: no multiplication
// on the quotient digit's chain).  Pass 1 then gives X0*Y0 = t0*R - Q~*N~ = t0*R - (u Q~)*N, so the correction in
// front of N is u*Q~, and the second pass starts from
//       C2 - u*Q~  =  C2' + u * (R - 1 - Q~),     C2 = N * ceil(u (R - 1) / N),   C2' = C2 - u (R - 1)  in [0, N)
// limb-wise: c2'_i + u * (MASK - q~_i), again without a borrow (64-bit lazy columns).  Digits stay below 2 N~; the
// last multiplication of an exponentiation (by (1, 0)) is done with the plain passes, which bring both digits of
// the result below 2N for the epilogue.
template <class M_t>
struct PairArithT {
  static constexpr int L = M_t::LIMBS;
  static constexpr u32 MASK = M_t::MASK;
  static constexpr int FRF = M_t::F_FRIENDLY;
  M_t& M;
  const u32* cp;      // LDS: limbs of C' = C - R + 1 (one copy per workgroup, slice of lane p at cp[p*L ..])
  const u32* cp2;     // LDS: limbs of C2' (friendly passes), same layout; nullptr where they are not used

  __device__ __forceinline__ PairArithT(M_t& m, const u32* cprime_lds, const u32* c2prime_lds = nullptr)
      : M(m), cp(cprime_lds), cp2(c2prime_lds) {}

  // accumulator start of the second pass: C' + (R - 1 - Q), limb-wise (in place in q)
  __device__ __forceinline__ void second_pass_init(u32 (&q)[L]) const {
    const bool has_q = M.p < M.nblk;
#pragma unroll
    for (int j = 0; j < L; ++j) q[j] = cp[M.p * L + j] + (has_q ? (MASK - q[j]) : 0u);
  }
  // friendly form: init = C2' (returned in c2), q becomes MASK - q~ (scaled by u inside the product)
  __device__ __forceinline__ void second_pass_init_friendly(u32 (&c2)[L], u32 (&q)[L]) const {
    const bool has_q = M.p < M.nblk;
#pragma unroll
    for (int j = 0; j < L; ++j) { c2[j] = cp2[M.p * L + j]; q[j] = has_q ? (MASK - q[j]) : 0u; }
  }

  // The two passes of a pair product, separately: the first (Z0 and the quotient Q) involves only the
  // first digits of the operands, the second only reads Q.  One wavefront runs both in turn (mul / sqr
  // below); the split kernel gives each pass its own wavefront (mx_powmod_n2_split.hpp).
  //   multiplication, multipliers staged in LDS by M.stage_multipliers(y0, y1) (pass 1 reads y0 only)
  template <bool FR = false>
  __device__ __forceinline__ void mul_pass1(u32 (&t0)[L], u32 (&q)[L], u32 (&x0)[L]) {
    M.template mulx<M_t::F_RECORD_Q | M_t::F_STAGED | (FR ? FRF : 0)>(t0, x0, x0, x0, x0, x0, q, nullptr, M.nblk);
  }
  //   pass 1 on its own (stages y0 itself; the split kernel's first wavefront has no use for y1)
  template <bool FR = false>
  __device__ __forceinline__ void mul_pass1_unstaged(u32 (&t0)[L], u32 (&q)[L], u32 (&x0)[L], const u32 (&y0)[L]) {
    M.template mulx<M_t::F_RECORD_Q | (FR ? FRF : 0)>(t0, x0, y0, x0, y0, y0, q, nullptr, M.nblk);
  }
  template <bool FR = false>
  __device__ __forceinline__ void mul_pass2(u32 (&z1)[L], u32 (&x0)[L], u32 (&x1)[L], u32 (&q)[L]) {
    if constexpr (FR) {
      u32 c2[L];
      second_pass_init_friendly(c2, q);
      M.template mulx<M_t::F_TWO | M_t::F_INIT | M_t::F_INITQ | M_t::F_STAGED | FRF>(z1, x1, x1, x0, x1, c2, nullptr, nullptr, M.nblk, q);
    } else {
      second_pass_init(q);
      M.template mulx<M_t::F_TWO | M_t::F_INIT | M_t::F_STAGED>(z1, x1, x1, x0, x1, q, nullptr, nullptr, M.nblk);
    }
  }
  //   squaring
  template <bool FR = false>
  __device__ __forceinline__ void sqr_pass1(u32 (&t0)[L], u32 (&q)[L], u32 (&x0)[L]) {
    M.template mulx<M_t::F_RECORD_Q | M_t::F_SQUARE | (FR ? FRF : 0)>(t0, x0, x0, x0, x0, x0, q, nullptr, M.nblk);
  }
  template <bool FR = false>
  __device__ __forceinline__ void sqr_pass2(u32 (&z1)[L], u32 (&x0)[L], u32 (&x1)[L], u32 (&q)[L]) {
    if constexpr (FR) {
      u32 c2[L];
      second_pass_init_friendly(c2, q);
      M.template mulx<M_t::F_INIT | M_t::F_INITQ | M_t::F_BDOUBLE | FRF>(z1, x0, x1, x0, x1, c2, nullptr, nullptr, M.nblk, q);   // 2 * X0 * X1
    } else {
      second_pass_init(q);
      M.template mulx<M_t::F_INIT | M_t::F_BDOUBLE>(z1, x0, x1, x0, x1, q, nullptr, nullptr, M.nblk);   // 2 * X0 * X1
    }
  }

  // (z0, z1) = (x0, x1) * (y0, y1); outputs may alias inputs
  template <bool FR = false>
  __device__ __forceinline__ void mul(u32 (&z0)[L], u32 (&z1)[L], u32 (&x0)[L], u32 (&x1)[L],
                                      const u32 (&y0)[L], const u32 (&y1)[L]) {
    // both multipliers go to LDS once: pass 1 reads y0, pass 2 reads y0 (row X1*Y0) and y1 (row X0*Y1)
    M.stage_multipliers(y0, y1);
    u32 t0[L], q[L];
    mul_pass1<FR>(t0, q, x0);
    // Z0 is needed only after pass 2, and the compiler would sink pass 1's carry sweep behind pass 2's loop: the
    // un-carried 64-bit columns of t0 (2 L registers) then stay live through the two-row loop — the 24 spilled
    // registers of the 18-limb instances (round 3: 116 B of scratch).  Pinned here, Z0 crosses the loop as L words.
#pragma unroll
    for (int j = 0; j < L; ++j) asm volatile("" : "+v"(t0[j]));
    mul_pass2<FR>(z1, x0, x1, q);
#pragma unroll
    for (int j = 0; j < L; ++j) z0[j] = t0[j];
  }

  // (z0, z1) = (x0, x1)^2
  template <bool FR = false>
  __device__ __forceinline__ void sqr(u32 (&z0)[L], u32 (&z1)[L], u32 (&x0)[L], u32 (&x1)[L]) {
    u32 t0[L], q[L];
    sqr_pass1<FR>(t0, q, x0);
    sqr_pass2<FR>(z1, x0, x1, q);
#pragma unroll
    for (int j = 0; j < L; ++j) z0[j] = t0[j];
  }
};
template <int K, int L, int W>
using PairArith = PairArithT<Mont<K, L, W, true>>;

// W-bit field of a little-endian word array starting at bit `bitpos` (words zero padded by the caller)
__device__ __forceinline__ u32 extract_field(const u32* words, int bitpos, int nbits) {
  const int w = bitpos >> 5, off = bitpos & 31;
  const u64 v = (u64)words[w] | ((u64)words[w + 1] << 32);
  return (u32)(v >> off) & (nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u));
}

#ifdef MX_DEV_PRIVATE_PAD_WORDS
// Developer build only (tools/build_variant.py -DMX_DEV_PRIVATE_PAD_WORDS=n, tools/concurrency_census.py): every lane of the
// one-wavefront pair kernel keeps a private array of n words in scratch memory, writes a pattern that names its
// writer (launch tag, workgroup, lane, index) before the tape and checks it after the tape.  A word that changed while
// the wavefront ran was overwritten by somebody else: the fault is counted and the first few are recorded, so that the
// census can say whether wrong rows under concurrent launches are the runtime's scratch or the kernel's arithmetic.
struct PadFault { u32 tag, block, lane, index, want, got, first_last; };
static __device__ u32 g_pad_faults;
static __device__ PadFault g_pad_fault_log[32];
__device__ __forceinline__ u32 pad_word(u32 tag, u32 block, u32 lane, u32 i) {
  return (tag << 24) ^ (block << 12) ^ (lane << 6) ^ (i & 63u) ^ ((i >> 6) * 0x9E3779B1u);
}
#endif

// FR: the passes of the tape reduce modulo the friendly multiple of N (see PairArithT above) — one multiply-class
// instruction less per limb step.  Needs W + 6 bits of room in R beyond what N needs (the host checks: A.friendly) and
// two more constant rows (N~ + 1 in registers INSTEAD of N, C2' in LDS).  The last product of an exponentiation
// (N2_MULC: by (1, 0), plain passes) and the epilogue need N itself and digits below 2N: a friendly instance never runs
// them — the host enqueues them as one more segment of the exponentiation on the plain instance of the same geometry
// (the accumulator travels through the scratch slot as between any two segments; N2_MULC has a tape position of its
// own for that), so that this kernel carries neither N nor the code of the plain passes (the 18-limb instances have no
// register to spare: with both in one kernel they spilled 49-55 registers).
template <int K, int L, int W, bool FR = false>
__global__ void __launch_bounds__(64, (L > 9 ? 2 : 3)) powmod_n2_kernel(PowmodN2Args A) {
  using M_t = Mont<K, L, W, true>;
  constexpr int S = M_t::S;
  constexpr int WIDE = M_t::LDS_WORDS;            // words: input row staging / output limbs (reuses the scratch)
  constexpr int GROUP_WORDS = M_t::LDS_WORDS;
  static_assert(WIDE >= 2 * S + 8, "row staging");
  extern __shared__ u32 smem[];
  constexpr int GPW = 64 / K;
  const int lane = threadIdx.x;
  const int gw = lane / K;
  u32* wide = smem + gw * GROUP_WORDS;            // same words as the Montgomery scratch M.lds
  // the element of this group of lanes (surplus groups redo the last one and store nothing).  Prologue and epilogue
  // each derive it from the lane index on their own — the epilogue from an opaque copy — so that nothing of it (the
  // row pointer, the flag) has to stay in registers, or in scratch, while the tape runs.
  auto element_of = [&](int ln, bool& valid_out) -> i64 {
    const i64 raw = (i64)blockIdx.x * GPW + ln / K;
    valid_out = raw < A.batch;
    return valid_out ? raw : A.batch - 1;
  };

  M_t M;
  M.init(smem + gw * GROUP_WORDS, A.nblk);
  M.load(M.n, A.consts, A.limbsn);
  M.setup_modulus();
  const int p = M.p;
  u32* cp_lds = smem + GPW * GROUP_WORDS;
  {
    u32 v[L];
    M.load(v, A.consts + 7 * A.limbsn, A.limbsn);
    if (gw == 0) {
#pragma unroll
      for (int j = 0; j < L; ++j) cp_lds[p * L + j] = v[j];
    }
    __syncthreads();
  }
  u32* cp2_lds = cp_lds + K * L;
  if constexpr (FR) {
    M.load(M.nf, A.consts + 8 * A.limbsn, A.limbsn + 1);          // N~ + 1
    M.setup_friendly();
    u32 v[L];
    M.load(v, A.consts + 8 * A.limbsn + (A.limbsn + 1), A.limbsn + 1);      // C2'
    if (gw == 0) {
#pragma unroll
      for (int j = 0; j < L; ++j) cp2_lds[p * L + j] = v[j];
    }
    __syncthreads();
  }
  PairArith<K, L, W> P(M, cp_lds, FR ? cp2_lds : nullptr);
  const i64 nlanes = (i64)gridDim.x * 64;
  u32* slots = A.slots + ((i64)blockIdx.x * 64 + lane);
  auto slot_at = [&](int slot, int half, int j) -> u32& { return slots[(((i64)slot * 2 + half) * L + j) * nlanes]; };

  // ---- prologue: constant pairs and the two halves of x into their slots (no arithmetic)
  if (A.first) {
    u32 v[L];
    const int rows[4][3] = {{N2_SLOT_K1, 3, 4}, {N2_SLOT_K2, 5, 6}, {N2_SLOT_ONE, 1, 2}, {N2_SLOT_E, -1, -1}};
    for (int r = 0; r < 4; ++r) {
      for (int half = 0; half < 2; ++half) {
        const int row = rows[r][1 + half];
        if (row >= 0) {
          M.load(v, A.consts + (i64)row * A.limbsn, A.limbsn);
        } else {
          M.set_small(v, half == 0 ? 1u : 0u);
        }
#pragma unroll
        for (int j = 0; j < L; ++j) slot_at(rows[r][0], half, j) = v[j];
      }
    }
    __syncthreads();
    bool valid_in;
    const u32* src = A.bases + element_of(lane, valid_in) * A.limbs2;
    for (int k = p; k < WIDE; k += K) wide[k] = (k < A.limbs2) ? src[k] : 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const int bit = W * (p * L + j);
      const int room = A.ksplit - bit;                       // bits of this limb that belong to x_lo
      const u32 lo = room <= 0 ? 0u : extract_field(wide, bit, room < W ? room : W);
      const int hbit = A.ksplit + bit;
      const u32 hi = (hbit + W + 32 <= 32 * WIDE) ? extract_field(wide, hbit, W) : 0u;
      slot_at(N2_SLOT_LO, 0, j) = lo;
      slot_at(N2_SLOT_LO, 1, j) = 0;
      slot_at(N2_SLOT_HI, 0, j) = hi;
      slot_at(N2_SLOT_HI, 1, j) = 0;
    }
  }

#ifdef MX_DEV_PRIVATE_PAD_WORDS
  volatile u32 pad[MX_DEV_PRIVATE_PAD_WORDS];
  const u32 pad_tag = (u32)((unsigned long long)A.out >> 12) & 0xFFu;       // distinguishes the launches in flight
#pragma unroll 1
  for (int i = 0; i < MX_DEV_PRIVATE_PAD_WORDS; ++i)
    pad[(i + lane) % MX_DEV_PRIVATE_PAD_WORDS] = pad_word(pad_tag, blockIdx.x, lane, (u32)((i + lane) % MX_DEV_PRIVATE_PAD_WORDS));
#endif
  // ---- the tape (this segment's part of it)
  u32 acc0[L], acc1[L];
  if (A.first) {
#pragma unroll
    for (int j = 0; j < L; ++j) { acc0[j] = 0; acc1[j] = 0; }
  } else {
#pragma unroll
    for (int j = 0; j < L; ++j) { acc0[j] = slot_at(N2_SLOT_CARRY, 0, j); acc1[j] = slot_at(N2_SLOT_CARRY, 1, j); }
  }
  int pos = 0;                                      // squarings executed by the tape so far
  const tape_ptr_t tape = (tape_ptr_t)A.tape;
  for (int k = 0; k < A.ntape; ++k) {
    if (pos >= A.pos_end) break;                       // the rest belongs to later segments
    const u32 word = tape[k];
    const u32 op = word >> 28;
    const int arg = (int)(word & 0x0FFFFFFFu);
    if (op == N2_MULC) pos += 1;                       // the last product: a position (and possibly a segment) of its own
    if (op == N2_SQR) {
      const int lo = pos > A.pos_begin ? pos : A.pos_begin;
      const int hi = pos + arg < A.pos_end ? pos + arg : A.pos_end;
      for (int s = lo; s < hi; ++s) P.template sqr<FR>(acc0, acc1, acc0, acc1);
      pos += arg;
      continue;
    }
    if (pos < A.pos_begin || pos >= A.pos_end) continue;   // another segment's operation
    if (op == N2_STORE) {
#pragma unroll
      for (int j = 0; j < L; ++j) { slot_at(arg, 0, j) = acc0[j]; slot_at(arg, 1, j) = acc1[j]; }
    } else {
      u32 f0[L], f1[L];
#pragma unroll
      for (int j = 0; j < L; ++j) { f0[j] = slot_at(arg, 0, j); f1[j] = slot_at(arg, 1, j); }
      if (op == N2_MUL || op == N2_MULC) {
        P.template mul<FR>(acc0, acc1, acc0, acc1, f0, f1);
      } else if (op == N2_ADD) {
        M.add(acc0, acc0, f0);
        M.add(acc1, acc1, f1);
      } else {   // N2_LOAD
#pragma unroll
        for (int j = 0; j < L; ++j) { acc0[j] = f0[j]; acc1[j] = f1[j]; }
      }
    }
  }
#ifdef MX_DEV_PRIVATE_PAD_WORDS
#pragma unroll 1
  for (int i = 0; i < MX_DEV_PRIVATE_PAD_WORDS; ++i) {
    const u32 idx = (u32)((i + lane) % MX_DEV_PRIVATE_PAD_WORDS);
    const u32 want = pad_word(pad_tag, blockIdx.x, lane, idx), got = pad[idx];
    if (got != want) {
      const u32 k = atomicAdd(&g_pad_faults, 1u);
      if (k < 32) g_pad_fault_log[k] = PadFault{pad_tag, blockIdx.x, (u32)lane, idx, want, got, (u32)(A.first * 2 + A.last)};
    }
  }
#endif
  if (FR || !A.last) {
#pragma unroll
    for (int j = 0; j < L; ++j) { slot_at(N2_SLOT_CARRY, 0, j) = acc0[j]; slot_at(N2_SLOT_CARRY, 1, j) = acc1[j]; }
    return;
  }

  // ---- epilogue: acc already is the pair whose N-adic value Y0 + Y1*N is the residue (the tape ends
  // with a multiplication by E = (1, 0)).  Digits into [0, N) — a lazy Y0 >= N carries one unit into
  // Y1 (Y0 <= N and Y1 <= N + 1 after that last product) — then z = Y0 + Y1 * N by a plain product.
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = acc0[j];
    M.normalize_full(acc0, t);
    const u32 carry = M.cond_sub(acc0);
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = acc1[j];
    if (p == 0) t[0] += carry;
    M.normalize_full(acc1, t);
    M.cond_sub(acc1);
  }
  // the low limbs are written over the staged multiplier: block blk emits into the L words it has
  // just read (LDS operations of a wavefront execute in order)
  u32 hi[L];
  __syncthreads();
  M.template mulx<M_t::F_INIT | M_t::F_PLAIN>(hi, acc1, M.n, acc1, M.n, acc0, nullptr, wide, A.nblk);
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = hi[j];
    M.normalize_full(hi, t);
  }
  const int it = A.nblk * L;
#pragma unroll
  for (int j = 0; j < L; ++j) wide[it + p * L + j] = hi[j];
  if (p == 0) { wide[it + S] = 0; wide[it + S + 1] = 0; wide[it + S + 2] = 0; wide[it + S + 3] = 0; }
  __syncthreads();
  // (the lane index again from the wavefront itself: a workgroup is one wavefront, and threadIdx.x would have to be
  // kept from the first instruction of the kernel)
  int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane_e));
  bool valid;
  u32* dst = A.out + element_of(lane_e, valid) * A.limbs2;
  const int nl = it + S;
  for (int k = lane_e % K; k < A.limbs2; k += K) {
    const int bit = 32 * k;
    const int g = bit / W, off = bit - g * W;
    u32 o = 0;
    if (g < nl) {
      u64 v = (u64)wide[g] >> off;
      v |= (u64)wide[g + 1] << (W - off);
      if (2 * W - off < 32) v |= (u64)wide[g + 2] << (2 * W - off);
      o = (u32)v;
    }
    if (valid) dst[k] = o;
  }
}

}  // namespace mx
