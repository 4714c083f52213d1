// The N^2-modulus modexp of mx_powmod_n2.hpp with the two passes of every pair product on TWO
// wavefronts — the geometry for launches that do not fill the machine (a lone decrypt(), a keygen-sized
// batch, one 10 000-ciphertext sequence: paillier_shared_key.py:92 called once at
// distributed_keygen.py:345-349 or looped once at :463-466).
//
// Why it splits.  With x = rho (X0 + X1 N) the first digit of a product depends on the first digits only:
//       Z0 = REDC(X0 Y0)   (and the quotient Q of that reduction),
//       Z1 = REDC(X0 Y1 + X1 Y0 + C - Q).
// So the chain of first digits X0 is an ordinary Montgomery exponentiation modulo N that never looks at
// X1, and the chain of second digits only consumes what the first chain produced one operation earlier:
// (X0 before the operation, Q).  A pair of wavefronts on two SIMDs of one CU shares the work:
//       wavefront A   for every operation s:  pass 1 of s,  mailbox[s & 1] <- (X0 before s, Q_s),  produced = s + 1
//       wavefront B   for every operation s:  wait for produced > s,  (X0, Q) <- mailbox[s & 1],  consumed = s + 1,  pass 2 of s
// (A waits for consumed >= s - 1 before it overwrites an entry).  B runs behind A; the two counters (LDS,
// release / acquire at workgroup scope, polled with s_sleep) are the only synchronisation and the mailbox (LDS,
// two entries) the only traffic (2 L words per lane per operation).  No workgroup barrier inside the tape: the
// pairs of a workgroup do not wait for each other.  Pass 1 is the lighter one (a squaring's
// pass 1 is symmetric; a multiplication's pass 2 has two product rows), so an operation costs what its
// pass 2 costs: 0.54-0.58 of the one-wavefront kernel's time per operation, with twice the wavefronts in
// the launch.  Slots in device memory are split the same way (A owns the first digits, B the second), so
// the table, the conversion and the segment hand-over need no further exchange; only the epilogue, which
// needs both digits in one place, receives A's final X0 through the mailbox.
//
// A workgroup holds TWO such pairs (four wavefronts): the dispatcher deals the wavefronts of a workgroup round
// robin over the four SIMDs of its CU, but starts every workgroup at the same SIMD — two-wavefront workgroups
// ended up stacked on SIMDs 0 and 1 while 2 and 3 idled (tools/sweep_shapes.py: a launch with two of them per CU
// took 1.5x as long as one with a single one).  With four wavefronts a workgroup covers the CU evenly.  The two
// pairs share nothing but the copy of C' (with a workgroup barrier per operation instead of the counters they
// waited for each other: 35.2 instead of 32.9 ms per wide launch at key_length 2048, 23.3 instead of 21.5 at L = 9;
// A/B on one box).  (Swapping the roles of the two wavefronts in every second workgroup a CU receives, so that every
// SIMD carries the same mix of light and heavy wavefronts, changed nothing measurable: not kept.)
//
// Small-L instances (L = 3: 32 lanes per element at key_length 2048, 64 at 4096) exist only in this form:
// they are the latency geometry — a limb step costs 2 L multiply-accumulates plus ~10 instructions of
// quotient / carry handling, so fewer limbs per lane shortens the dependent chain of one element at the
// price of more instructions per element, which only pays while wavefront slots are idle anyway.
#pragma once
#include "mx_powmod_n2.hpp"

namespace mx {

constexpr int N2_SPLIT_PAIRS = 2;      // wavefront pairs per workgroup (4 wavefronts: one per SIMD of a CU)

template <int K, int L>
constexpr size_t powmod_n2_split_lds_bytes() {
  // per pair: two wavefronts' Montgomery scratch, the two mailbox entries of 2 L words per lane and the two
  // hand-over counters; one C'
  return ((size_t)N2_SPLIT_PAIRS * ((size_t)2 * (64 / K) * (2 * K * L + 8) + (size_t)2 * 2 * L * 64 + 4) + (size_t)K * L) * 4;
}

template <int K, int L, int W>
__global__ void __launch_bounds__(64 * 2 * N2_SPLIT_PAIRS, (L > 9 ? 2 : 3)) powmod_n2_split_kernel(PowmodN2Args A) {
  using M_t = Mont<K, L, W, true, false>;          // wavefront-level ordering of the group scratch
  constexpr int S = M_t::S;
  constexpr int GROUP_WORDS = M_t::LDS_WORDS;
  constexpr int WIDE = GROUP_WORDS;
  extern __shared__ u32 smem[];
  constexpr int GPW = 64 / K;
  const int lane = threadIdx.x & 63;
  // 0: wavefront A (first digits), 1: wavefront B (second digits); in an SGPR, so that the two roles are
  // uniform branches
  const int half = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) & 1));
  const int pair = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));          // which pair of the workgroup
  const int gw = lane / K;
  const i64 wave_slot = (i64)blockIdx.x * N2_SPLIT_PAIRS + pair;                     // the pair's index in the launch
  const i64 elem_raw = wave_slot * GPW + gw;
  const bool valid = elem_raw < A.batch;
  const i64 elem = valid ? elem_raw : A.batch - 1;
  constexpr int PAIR_WORDS = 2 * GPW * GROUP_WORDS + 2 * 2 * L * 64 + 4;
  u32* pair_lds = smem + pair * PAIR_WORDS;
  u32* wide = pair_lds + (half * GPW + gw) * GROUP_WORDS;  // this wavefront's scratch of this group
  u32* mbox = pair_lds + 2 * GPW * GROUP_WORDS;             // [2 entries][2 L words][64 lanes]
  u32* produced = mbox + 2 * 2 * L * 64;                    // entries A has handed over / B has taken (this pair)
  u32* consumed = produced + 1;
  if (half == 0 && lane == 0) { *produced = 0; *consumed = 0; }
  u32* cp_lds = smem + N2_SPLIT_PAIRS * PAIR_WORDS;
  auto mb = [&](int entry, int j) -> u32& { return mbox[(entry * 2 * L + j) * 64 + lane]; };

  M_t M;
  M.init(wide, A.nblk);
  M.load(M.n, A.consts, A.limbsn);
  M.setup_modulus();
  const int p = M.p;
  if (half == 1) {
    u32 v[L];
    M.load(v, A.consts + 7 * A.limbsn, A.limbsn);
    if (gw == 0 && pair == 0) {
#pragma unroll
      for (int j = 0; j < L; ++j) cp_lds[p * L + j] = v[j];
    }
  }
  PairArithT<M_t> P(M, cp_lds);
  const i64 nlanes = (i64)gridDim.x * N2_SPLIT_PAIRS * 64;
  u32* slots = A.slots + (wave_slot * 64 + lane);
  // this wavefront's digit of a pair slot
  auto slot_at = [&](int slot, int j) -> u32& { return slots[(((i64)slot * 2 + half) * L + j) * nlanes]; };
  auto slot_other = [&](int slot, int j) -> u32& { return slots[(((i64)slot * 2 + (1 - half)) * L + j) * nlanes]; };

  // ---- prologue: constant pairs and the two halves of x into their slots, every wavefront its own digit
  if (A.first) {
    u32 v[L];
    const int rows[4][3] = {{N2_SLOT_K1, 3, 4}, {N2_SLOT_K2, 5, 6}, {N2_SLOT_ONE, 1, 2}, {N2_SLOT_E, -1, -1}};
    for (int r = 0; r < 4; ++r) {
      const int row = rows[r][1 + half];
      if (row >= 0) {
        M.load(v, A.consts + (i64)row * A.limbsn, A.limbsn);
      } else {
        M.set_small(v, half == 0 ? 1u : 0u);
      }
#pragma unroll
      for (int j = 0; j < L; ++j) slot_at(rows[r][0], j) = v[j];
    }
    if (half == 0) {
      M_t::sync();
      const u32* src = A.bases + elem * A.limbs2;
      for (int k = p; k < WIDE; k += K) wide[k] = (k < A.limbs2) ? src[k] : 0u;
      M_t::sync();
#pragma unroll
      for (int j = 0; j < L; ++j) {
        const int bit = W * (p * L + j);
        const int room = A.ksplit - bit;                       // bits of this limb that belong to x_lo
        const u32 lo = room <= 0 ? 0u : extract_field(wide, bit, room < W ? room : W);
        const int hbit = A.ksplit + bit;
        const u32 hi = (hbit + W + 32 <= 32 * WIDE) ? extract_field(wide, hbit, W) : 0u;
        slot_at(N2_SLOT_LO, j) = lo;
        slot_at(N2_SLOT_HI, j) = hi;
      }
    } else {
#pragma unroll
      for (int j = 0; j < L; ++j) { slot_at(N2_SLOT_LO, j) = 0; slot_at(N2_SLOT_HI, j) = 0; }
    }
  }
  __syncthreads();            // C', the counters and the prologue's slots are in place (the kernel's only workgroup barrier)

  // ---- the tape (this segment's part of it): acc is THIS wavefront's digit of the accumulator pair
  u32 acc[L];
  if (A.first) {
#pragma unroll
    for (int j = 0; j < L; ++j) acc[j] = 0;
  } else {
#pragma unroll
    for (int j = 0; j < L; ++j) acc[j] = slot_at(N2_SLOT_CARRY, j);
  }
  int pos = 0;                                      // squarings executed by the tape so far
  u32 seq = 0;                                      // operations handed over so far (the same count in both wavefronts)
  // A: after pass 1 of an operation, hand (X0 before it, Q) to B.  B: take them before its pass 2.
  auto send = [&](const u32 (&x0)[L], const u32 (&q)[L]) {
    while (__hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) + 2u <= seq) __builtin_amdgcn_s_sleep(2);
    const int entry = (int)(seq & 1u);
#pragma unroll
    for (int j = 0; j < L; ++j) { mb(entry, j) = x0[j]; mb(entry, L + j) = q[j]; }
    ++seq;
    __hip_atomic_store(produced, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto receive = [&](u32 (&x0)[L], u32 (&q)[L]) {
    while (__hip_atomic_load(produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= seq) __builtin_amdgcn_s_sleep(2);
    const int entry = (int)(seq & 1u);
#pragma unroll
    for (int j = 0; j < L; ++j) { x0[j] = mb(entry, j); q[j] = mb(entry, L + j); }
    ++seq;
    __hip_atomic_store(consumed, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  for (int k = 0; k < A.ntape; ++k) {
    const u32 word = A.tape[k];
    const u32 op = word >> 28;
    const int arg = (int)(word & 0x0FFFFFFFu);
    if (op == N2_SQR) {
      const int lo = pos > A.pos_begin ? pos : A.pos_begin;
      const int hi = pos + arg < A.pos_end ? pos + arg : A.pos_end;
      for (int s = lo; s < hi; ++s) {
        u32 q[L];
        if (half == 0) {
          u32 t0[L];
          P.sqr_pass1(t0, q, acc);
          send(acc, q);
#pragma unroll
          for (int j = 0; j < L; ++j) acc[j] = t0[j];
        } else {
          u32 x0[L];
          receive(x0, q);
          P.sqr_pass2(acc, x0, acc, q);
        }
      }
      pos += arg;
      continue;
    }
    if (pos < A.pos_begin || pos >= A.pos_end) continue;   // another segment's operation
    if (op == N2_STORE) {
#pragma unroll
      for (int j = 0; j < L; ++j) slot_at(arg, j) = acc[j];
    } else if (op == N2_MUL) {
      u32 q[L], f0[L];
      if (half == 0) {
        u32 t0[L];
#pragma unroll
        for (int j = 0; j < L; ++j) f0[j] = slot_at(arg, j);
        P.mul_pass1_unstaged(t0, q, acc, f0);
        send(acc, q);
#pragma unroll
        for (int j = 0; j < L; ++j) acc[j] = t0[j];
      } else {
        u32 x0[L], f1[L];
#pragma unroll
        for (int j = 0; j < L; ++j) f1[j] = slot_at(arg, j);
        // the first digit of the table entry was written by wavefront A: read it behind the hand-over of this
        // operation (A stored it before it released `produced`; the acquire orders this wavefront's loads after it)
        receive(x0, q);
#pragma unroll
        for (int j = 0; j < L; ++j) f0[j] = slot_other(arg, j);
        M.stage_multipliers(f0, f1);
        P.mul_pass2(acc, x0, acc, q);
      }
    } else {
      u32 f[L];
#pragma unroll
      for (int j = 0; j < L; ++j) f[j] = slot_at(arg, j);
      if (op == N2_ADD) {
        M.add(acc, acc, f);
      } else {   // N2_LOAD
#pragma unroll
        for (int j = 0; j < L; ++j) acc[j] = f[j];
      }
    }
  }
  if (!A.last) {
#pragma unroll
    for (int j = 0; j < L; ++j) slot_at(N2_SLOT_CARRY, j) = acc[j];
    return;
  }

  // ---- epilogue on wavefront B, which receives A's final first digit through the mailbox
  u32 acc0[L], acc1[L];
  if (half == 0) {
    send(acc, acc);
    return;
  }
  {
    u32 unused[L];
    receive(acc0, unused);
  }
#pragma unroll
  for (int j = 0; j < L; ++j) acc1[j] = acc[j];
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
  u32 hi[L];
  M_t::sync();
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
  M_t::sync();
  u32* dst = A.out + elem * A.limbs2;
  const int nl = it + S;
  for (int k = p; k < A.limbs2; k += K) {
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
