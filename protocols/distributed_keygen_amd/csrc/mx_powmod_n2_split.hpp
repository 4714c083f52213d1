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
// PERSISTENT form (time-slicing).  A launch runs at its "one workgroup per CU" time until it needs a second workgroup
// per CU, then at its "two per CU" time, and so on: 10 000 ciphertexts are 625 workgroups of the
// L = 9 shape, 2.44 per CU, and cost what 768 would.  In the persistent form the launch has exactly r x CUs
// workgroups (r = 1, 2 ...), and every pair of wavefronts repeatedly takes the next UNIT of work from a queue in
// device memory: a unit is one segment (a stretch of the tape, as between the launches of a segmented
// exponentiation: the accumulator travels through the scratch slot) of one group of elements.  The queue starts
// with the first segment of every group; the pair that finishes a segment appends the group's next one (release /
// acquire at agent scope around the slots in device memory), and it is taken — by whichever pair is free, on
// whichever CU — when its turn comes.  The groups thus share the resident wavefronts in time and the launch costs
// (groups / resident pairs) x the time of a full CU instead of the next whole multiple.
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
constexpr size_t powmod_n2_split_lds_bytes(bool friendly = L == 3) {
  // per pair: two wavefronts' Montgomery scratch, the two mailbox entries of 2 L words per lane and the two
  // hand-over counters (+ the unit word of the persistent form); one C'
  return ((size_t)N2_SPLIT_PAIRS * ((size_t)2 * (64 / K) * (2 * K * L + 8) + (size_t)2 * 2 * L * 64 + 4) + (size_t)(friendly ? 2 : 1) * K * L) * 4;
}

// Scheduling state of a time-sliced launch in device memory (zeroed by the host before the launch): the work queues
// described at the unit loop below, N2_TS_HEADER + groups x (segments - 1) words.
constexpr int N2_TS_LEVELS = 16;       // units per group of a time-sliced launch, at most
constexpr int N2_TS_HEADER = 2 * N2_TS_LEVELS;
// Register budget: three workgroups per CU (168 registers) for the 9- and 3-limb instances, plain and time-sliced, two
// (256) for the 18-limb ones.  No instance has a private segment (tools/scratch_report.py, tests/test_instances.py).
// The time-sliced instances must NOT be given the whole register file: with 256 registers a launch of 2 x CUs resident
// workgroups leaves no slot for anybody else, and four such launches started a few milliseconds apart took 4-38 SECONDS
// for their first round (three of the four stalled until the scheduler's time slice came round; measured in round 4,
// profiles/r04_timesliced_first_round.txt) — and were no faster than the plain launch (53 vs 47 ms for 10 000).
// Agent-scope release / acquire of a wavefront's global stores for the hand-over of a group between pairs that may sit
// on different XCDs (each XCD has its own L2): spelled out, not left to the atomics' memory orders.  The compiler's
// sequence for `__hip_atomic_store(..., __ATOMIC_RELEASE, agent)` behind an atomic whose result had just been waited for
// was  buffer_wbl2 sc1 ; s_waitcnt lgkmcnt(0) ; global_store sc1  — no wait for the WRITE-BACK before the flag (its
// waitcnt pass does not count buffer_wbl2 as outstanding), and a pair that took the group the moment the flag appeared
// read the previous segment's accumulator half-written: 1 group in ~1000 hand-overs wrong when takers were idle and
// waiting (round 5, found when the scheduler below began to hand groups over hot; the FIFO of rounds 3-4 mostly
// handed a group back to the pair that had pushed it).
__device__ __forceinline__ void ts_release_agent() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void ts_acquire_agent() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
}

// (MX_DEV_TS_MIN_WAVES and the other MX_DEV_ switches below: developer builds, csrc/mx_dev.hpp)
template <int K, int L, int W, bool PERSISTENT, bool FRIENDLY = (L == 3)>
__global__ void __launch_bounds__(64 * 2 * N2_SPLIT_PAIRS, (L > 9 ? 2 : PERSISTENT ? MX_DEV_TS_MIN_WAVES : 3)) powmod_n2_split_kernel(PowmodN2Args A) {
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
  constexpr int PAIR_WORDS = 2 * GPW * GROUP_WORDS + 2 * 2 * L * 64 + 4;
  u32* pair_lds = smem + pair * PAIR_WORDS;
  u32* wide = pair_lds + (half * GPW + gw) * GROUP_WORDS;  // this wavefront's scratch of this group
  u32* mbox = pair_lds + 2 * GPW * GROUP_WORDS;             // [2 entries][2 L words][64 lanes]
  u32* produced = mbox + 2 * 2 * L * 64;                    // entries A has handed over / B has taken (this pair)
  u32* consumed = produced + 1;
  u32* unit_word = produced + 2;                            // persistent form: the unit A has taken for the pair
  if (half == 0 && lane == 0) { *produced = 0; *consumed = 0; }
  u32* cp_lds = smem + N2_SPLIT_PAIRS * PAIR_WORDS;
  // The 3-limb (latency) instances run their passes modulo the friendly multiple of N (mx_powmod_n2.hpp): the
  // multiplication by -N^-1 leaves every limb step.  They have 3 * K * 29 bits of room where N needs K * 3 * 29 - 35
  // at most (mx_host.hpp: choose_geometry).  The 9-limb instances of key_length 2048 and 4096 (groups of 8 and 16
  // lanes) exist in both forms; the host launches the friendly one where the modulus leaves that room.
  constexpr bool FR = FRIENDLY;
  u32* cp2_lds = cp_lds + K * L;
  auto mb = [&](int entry, int j) -> u32& { return mbox[(entry * 2 * L + j) * 64 + lane]; };

  M_t M;
  M.init(wide, A.nblk);
  M.load(M.n, A.consts, A.limbsn);
  M.setup_modulus();
  const int p0 = M.p;
  if constexpr (FR) {
    M.load(M.nf, A.consts + 8 * A.limbsn, A.limbsn + 1);          // N~ + 1
    M.setup_friendly();
  }
  if (half == 1) {
    u32 v[L];
    M.load(v, A.consts + 7 * A.limbsn, A.limbsn);
    if (gw == 0 && pair == 0) {
#pragma unroll
      for (int j = 0; j < L; ++j) cp_lds[p0 * L + j] = v[j];
    }
    if constexpr (FR) {
      M.load(v, A.consts + 8 * A.limbsn + (A.limbsn + 1), A.limbsn + 1);      // C2'
      if (gw == 0 && pair == 0) {
#pragma unroll
        for (int j = 0; j < L; ++j) cp2_lds[p0 * L + j] = v[j];
      }
    }
  }
  PairArithT<M_t> P(M, cp_lds, FR ? cp2_lds : nullptr);
  __syncthreads();            // C' and the counters are in place (the kernel's only workgroup barrier)

  u32 seq = 0;                                      // entries handed over so far (the same count in both wavefronts)
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
  // entries without payload: "everything I stored before this is yours to read" (A -> B)
  auto send_token = [&]() {
    while (__hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) + 2u <= seq) __builtin_amdgcn_s_sleep(2);
    ++seq;
    __hip_atomic_store(produced, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto receive_token = [&]() {
    while (__hip_atomic_load(produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= seq) __builtin_amdgcn_s_sleep(2);
    ++seq;
    __hip_atomic_store(consumed, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  // ---- the units of this pair: one (the launch's segment of its own group) in the plain form.  Time-sliced form:
  // work queues in device memory.  Unit = (segment, group), numbered segment-major.  There is one queue per LEVEL —
  // level k holds the groups whose first k segments are done — and a free pair takes from the LOWEST level that has an
  // entry: the group with the most work left goes first.  (Round 5.  The first form of this scheduler was one FIFO
  // over all units: every group then advances at the same pace, the groups that start a round late — 113 of 625 at
  // key_length 2048 on 512 resident pairs — also end a round late, and the launch ends with a few chains of several
  // segments each and most pairs idle: 51 ms for 10 000 ciphertexts in 4 units per group where 625 x 4 / 512 units take
  // 41.  A discrete-event model of both disciplines is tools/ts_schedule_model.py.)  Level 0 is implicit (every group,
  // in order); an entry of level k >= 1 is written by the pair that finished segment k - 1 of that group.  A pair that
  // finds every level empty sleeps and looks again until all groups x segments units have been claimed.
  const u32 groups = (u32)A.sched_groups, nseg = (u32)A.sched_segments;
  u32* const q_head = A.sched;                               // [N2_TS_LEVELS] entries granted per level
  u32* const q_tail = A.sched + N2_TS_LEVELS;                // [N2_TS_LEVELS] entries reserved by their writers per level (level 0 unused)
  u32* const q_ring = A.sched + N2_TS_HEADER;                // level k >= 1, entry i at [(k - 1) * groups + i]: group + 1, 0 = not yet written
  for (;;) {
    // Time-sliced form: everything the prologue and the epilogue of a unit derive from the lane position (bit offsets,
    // masks and LDS addresses of the limb conversions: some fifty values) is invariant across units, and the compiler
    // would hoist it out of this loop and keep it — in scratch memory, the register budget being what it is (round 3
    // shipped these instances with 59-75 spilled registers).  An opaque lane position per unit keeps those values
    // inside the unit, where they are computed, used and dropped.
    if constexpr (PERSISTENT) asm volatile("" : "+v"(M.p));
    const int p = M.p;
    i64 slot, nlanes;
    int first, last, pos_begin, pos_end;
    u32 g = 0, sg = 0;
#ifdef MX_DEV_TS_TRACE
    u64 trace_t0 = 0;
#endif
    if constexpr (!PERSISTENT) {
      slot = (i64)blockIdx.x * N2_SPLIT_PAIRS + pair;
      nlanes = (i64)gridDim.x * N2_SPLIT_PAIRS * 64;
      first = A.first; last = A.last; pos_begin = A.pos_begin; pos_end = A.pos_end;
    } else {
      u32 u;
      if (half == 0) {
        // B has taken everything of the previous unit (its last entry is the end token): the pair is free
        while (__hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq) __builtin_amdgcn_s_sleep(2);
        u = groups * nseg;                                   // "no unit left"
        if (lane == 0) {
          // Level 0 is taken with a fetch-and-add (1024 pairs start at once: with a compare-and-swap there the launch
          // began with 12 ms of retries; the head may overshoot, an index beyond the groups is no claim).  The other
          // levels need the EXACT test "granted < written" — a pair must not pass over a level that has an entry — and
          // take with a compare-and-swap: pairs finish their units microseconds apart, a handful contend at a time.
          // (Counting semaphores per level — decrement, give back on failure — are cheaper and not exact: while one
          // pair's failed decrement is outstanding the level looks empty to the others, a few groups were passed over
          // for a whole round and the launch ended with two chains running alone: 50 instead of 42 ms, traced with
          // tools/ts_trace.py.)
          for (;;) {
            u32 claimed = groups;
            bool got = false;
            {
              u32 h = __hip_atomic_load(q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (h < groups) h = __hip_atomic_fetch_add(q_head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (h < groups) { u = h; got = true; }           // the groups in order
            }
            for (u32 k = 1; k < nseg && !got; ++k) {
              u32 h = __hip_atomic_load(q_head + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const u32 t = __hip_atomic_load(q_tail + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              while (h < t && !got)
                got = __hip_atomic_compare_exchange_strong(q_head + k, &h, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (got) {
                // the writer reserved the entry before it wrote it: a moment at most
                const u32* e = q_ring + (size_t)(k - 1u) * groups + h;
                u32 g1;
                while ((g1 = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(4);
                u = k * groups + (g1 - 1u);
                break;
              }
              claimed += h;
            }
            if (got || claimed >= groups * nseg) break;       // (the heads only grow: their sum reaches the total once)
            __builtin_amdgcn_s_sleep(127);                     // ~4 us: idle pairs must not crowd the queue words
            __builtin_amdgcn_s_sleep(127);
          }
        }
        u = (u32)__builtin_amdgcn_readfirstlane((int)u);
        if (lane == 0) *unit_word = u;
        send_token();
      } else {
        receive_token();
        u = (u32)__builtin_amdgcn_readfirstlane((int)*unit_word);
      }
      if (u >= groups * nseg) break;
      g = u % groups; sg = u / groups;
#ifdef MX_DEV_TS_TRACE          // developer builds (tools/ts_trace.py): when and where every unit ran, behind the queues
      trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
      // what the pair that pushed this unit stored (slots, accumulator) before its release of the entry
      if (sg > 0) ts_acquire_agent();
      slot = (i64)g;
      nlanes = (i64)groups * 64;
      first = sg == 0;
      last = sg == nseg - 1;
      pos_begin = (int)((i64)A.sched_n_sqr * sg / nseg);
      pos_end = last ? 0x7FFFFFFF : (int)((i64)A.sched_n_sqr * (sg + 1) / nseg);
    }
    // ---- one unit of work: the part of the tape with positions [pos_begin, pos_end) for the group `slot`
    const i64 elem_raw = slot * GPW + gw;
    const bool valid = elem_raw < A.batch;
    const i64 elem = valid ? elem_raw : A.batch - 1;
    u32* slots = A.slots + (slot * 64 + lane);
    // this wavefront's digit of a pair slot
    auto slot_at = [&](int sl, int j) -> u32& { return slots[(((i64)sl * 2 + half) * L + j) * nlanes]; };
    auto slot_other = [&](int sl, int j) -> u32& { return slots[(((i64)sl * 2 + (1 - half)) * L + j) * nlanes]; };

    // ---- prologue: constant pairs and the two halves of x into their slots, every wavefront its own digit
    if (first) {
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
        send_token();           // A's constant digits are in their slots before B multiplies by them
      } else {
#pragma unroll
        for (int j = 0; j < L; ++j) { slot_at(N2_SLOT_LO, j) = 0; slot_at(N2_SLOT_HI, j) = 0; }
        receive_token();
      }
    }

    // ---- the tape (this segment's part of it): acc is THIS wavefront's digit of the accumulator pair
    u32 acc[L];
    if (first) {
#pragma unroll
      for (int j = 0; j < L; ++j) acc[j] = 0;
    } else {
#pragma unroll
      for (int j = 0; j < L; ++j) acc[j] = slot_at(N2_SLOT_CARRY, j);
    }
    int pos = 0;                                      // squarings executed by the tape so far
    const tape_ptr_t tape = (tape_ptr_t)A.tape;        // scalar loads (mx_powmod_n2.hpp)
    for (int k = 0; k < A.ntape; ++k) {
      if (pos >= pos_end) break;                       // the rest belongs to later segments
      const u32 word = tape[k];
      const u32 op = word >> 28;
      const int arg = (int)(word & 0x0FFFFFFFu);
      if (op == N2_MULC) pos += 1;                       // the last product has a tape position of its own (mx_powmod_n2.hpp)
      if (op == N2_SQR) {
        const int lo = pos > pos_begin ? pos : pos_begin;
        const int hi = pos + arg < pos_end ? pos + arg : pos_end;
        for (int s = lo; s < hi; ++s) {
          u32 q[L];
          if (half == 0) {
            u32 t0[L];
            P.template sqr_pass1<FR>(t0, q, acc);
            send(acc, q);
#pragma unroll
            for (int j = 0; j < L; ++j) acc[j] = t0[j];
          } else {
            u32 x0[L];
            receive(x0, q);
            P.template sqr_pass2<FR>(acc, x0, acc, q);
          }
        }
        pos += arg;
        continue;
      }
      if (pos < pos_begin || pos >= pos_end) continue;   // another segment's operation
      if (op == N2_STORE) {
#pragma unroll
        for (int j = 0; j < L; ++j) slot_at(arg, j) = acc[j];
      } else if (op == N2_MUL || op == N2_MULC) {
        // N2_MULC (the last product of an exponentiation): plain passes, digits below 2N for the epilogue
        const bool friendly = FR && op == N2_MUL;
        u32 q[L], f0[L];
        if (half == 0) {
          u32 t0[L];
#pragma unroll
          for (int j = 0; j < L; ++j) f0[j] = slot_at(arg, j);
          if (friendly) {
            P.template mul_pass1_unstaged<FR>(t0, q, acc, f0);
          } else {
            P.template mul_pass1_unstaged<false>(t0, q, acc, f0);
          }
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
          if (friendly) {
            P.template mul_pass2<FR>(acc, x0, acc, q);
          } else {
            P.template mul_pass2<false>(acc, x0, acc, q);
          }
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
    if (!last) {
#pragma unroll
      for (int j = 0; j < L; ++j) slot_at(N2_SLOT_CARRY, j) = acc[j];
    } else if (half == 0) {
      send(acc, acc);
    } else {
    // ---- epilogue on wavefront B, which receives A's final first digit through the mailbox
    u32 acc0[L], acc1[L];
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
    M_t::sync();          // the scratch is reused by the next unit of a time-sliced launch
    }

    if constexpr (!PERSISTENT) {
      break;
    } else {
      // end of the unit: B pushes the group's next segment for both wavefronts, and the pair that pops it may sit on
      // another CU or XCD.  B's agent-scope release covers B's own stores only, and A's hand-over to B is a
      // workgroup-scope LDS release that neither waits for A's global stores (carry slot, table slots) nor writes
      // them back from this XCD's L2 — so A releases them at agent scope itself, before the token that lets B push.
      if (half == 0) {
#ifndef MX_DEV_TS_NO_A_FENCE
        if (!last) ts_release_agent();
#endif
        send_token();
      } else {
        receive_token();
#ifdef MX_DEV_TS_TRACE
        if (lane == 0) {
          u32* tr = q_ring + (size_t)groups * (N2_TS_LEVELS - 1) + ((size_t)sg * groups + g) * 4;
          const u64 t1 = __builtin_amdgcn_s_memrealtime();
          u32 hw;
          asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
          tr[0] = (u32)trace_t0; tr[1] = (u32)t1; tr[2] = (u32)(blockIdx.x * N2_SPLIT_PAIRS + pair); tr[3] = hw;
        }
#endif
        if (!last) {
          // (sg + 1 <= nseg - 1 < N2_TS_LEVELS)
#ifndef MX_DEV_TS_COMPILER_RELEASE          // developer builds (tools/ts_handover_check.py): the sequence that lost groups
          ts_release_agent();                                  // B's own stores; A released its own before the token
#endif
          if (lane == 0) {
            const u32 t = __hip_atomic_fetch_add(q_tail + (sg + 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef MX_DEV_TS_COMPILER_RELEASE
            __hip_atomic_store(q_ring + (size_t)sg * groups + t, g + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
            __hip_atomic_store(q_ring + (size_t)sg * groups + t, g + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
          }
        }
      }
    }
  }
}

}  // namespace mx
