/*
 * mxpaillier — C ABI of the MI355X (gfx950) big-integer engine for the compute hot path of
 * TNO-MPC/protocols.distributed_keygen (distributed Paillier keygen + threshold decryption).
 *
 * The reference is pure Python; its arithmetic leaf is `pow_mod` / `mod_inv` of the un-vendored
 * package tno.mpc.encryption_schemes.utils (gmpy2 -> libgmp when installed), bound by name at
 *   src/tno/mpc/protocols/distributed_keygen/distributed_keygen.py:35   (DK)
 *   src/tno/mpc/protocols/distributed_keygen/paillier_shared_key.py:20  (PSK)
 * Each entry point below replaces one Python loop over those scalar calls; INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - Big integers are little-endian arrays of `limbs` uint32 words ("rows"), element-major:
 *     element e occupies words [e*limbs, (e+1)*limbs).  This is `int.to_bytes(4*limbs,"little")`.
 *   - d_* pointers are DEVICE pointers (e.g. torch.Tensor.data_ptr()), h_* are HOST pointers.
 *   - The caller owns every buffer, including the workspace; the library allocates nothing that
 *     outlives a call (the one exception is explicit: mx_stream_create_cu_slice hands out a stream that
 *     the caller destroys), keeps no state between calls and reads no environment variable.  All work is enqueued on `stream` (a hipStream_t, NULL = default stream);
 *     calls return after enqueueing and NEVER synchronise the stream or the device.  Host arrays
 *     (h_*) travel by value in kernel-argument blocks and may be freed/reused on return; operand
 *     sets of thousands of moduli should use the *_dev entry points (device-resident operands).
 *   - Return value: MX_OK (0) or a negative MX_ERR_* code; nothing throws across the ABI.
 *   - Moduli must be odd and >= 3.  Supported modulus size: up to 16 700 bits.
 *   - Bases / partials must be < their modulus (the reference guarantees this: UT:361, PSK:92);
 *     values up to 16 * modulus (and < 2^(32*limbs)) are still reduced correctly.
 */
#ifndef MXPAILLIER_H
#define MXPAILLIER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MX_OK 0
#define MX_ERR_ARG (-1)          /* null pointer / non-positive size */
#define MX_ERR_SIZE (-2)         /* modulus too large for the engine */
#define MX_ERR_MODULUS (-3)      /* even or < 3 modulus */
#define MX_ERR_WORKSPACE (-4)    /* workspace too small */
#define MX_ERR_HIP (-5)          /* a HIP runtime call failed (see mx_last_hip_error) */

/* ABI version (major*100 + minor). */
int mx_version(void);
const char* mx_error_string(int code);
/* hipGetErrorString of the last failing HIP call made by this library in this thread. */
const char* mx_last_hip_error(void);

/* ---- modular exponentiation ------------------------------------------------------------
 * Bytes of device workspace needed by mx_powmod_shared / mx_powmod_multi for these sizes. */
int64_t mx_powmod_workspace_bytes(int limbs, int exp_limbs, int64_t batch, int64_t groups);

/* d_out[e] = d_bases[e] ^ exp mod mod   for e in [0, batch): one modulus and one exponent for the
 * whole batch.  Replaces the loop `[secret_key.partial_decrypt(c) for c in ciphertext_sequence]`
 * (DK:463-466, single: DK:345-349) whose body is `pow_mod(c, exp, n_square)` (PSK:92), and any
 * other fixed-(exponent, modulus) batch (e.g. r^N mod N^2 of Paillier encryption).
 *   h_mod: limbs words, h_exp: exp_limbs words (exponent >= 0; a negative Lagrange exponent,
 *   PSK:89-91, is handled by the caller inverting the base, as the reference does). */
int mx_powmod_shared(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_mod,
                     const uint32_t* h_exp, int limbs, int exp_limbs, int64_t batch,
                     void* d_workspace, int64_t workspace_bytes, void* stream);

/* d_out[g*group_size + k] = d_bases[g*group_size + k] ^ h_exps[g] mod h_mods[g].
 * Replaces the candidate loop DK:1313-1329 over __biprime_test_v_calculation, whose body is
 * `pow_mod(g, (N - p_i - q_i + 1) // 4, N)` (DK:1094) or `pow_mod(g, (p_i + q_i) // 4, N)`
 * (DK:1097): group g = candidate modulus, group_size = number of Jacobi-1 bases kept (40). */
int mx_powmod_multi(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_mods,
                    const uint32_t* h_exps, int limbs, int exp_limbs, int64_t groups,
                    int64_t group_size, void* d_workspace, int64_t workspace_bytes, void* stream);

/* mx_powmod_shared with the lane geometry as an argument: limbs_per_lane 9 (narrow: more lanes per
 * element), 18 (wide: fewer, busier lanes), 3 (latency: many lanes per element, the exponentiation's products modulo
 * the friendly multiple of N — for launches that leave SIMDs idle, moduli up to 5533 bits), 6 (the bipartite latency form:
 * 3 limbs per lane, every product on two wavefronts, mx_powmod_launch_form) or 0 = automatic from an estimate of the
 * launch's duration: 6 while the launch leaves SIMDs idle, 3 up to about two such wavefronts per SIMD, else 9 or 18.
 * The same argument of mx_powmod_multi_dev and mx_powmod_geometry_for. */
int mx_powmod_shared_lpl(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_mod, const uint32_t* h_exp,
                         int limbs, int exp_limbs, int64_t batch, int limbs_per_lane, void* d_workspace,
                         int64_t workspace_bytes, void* stream);

/* mx_powmod_multi with DEVICE-resident moduli and exponents (d_mods [groups][limbs], d_exps
 * [groups][exp_limbs]): the form for thousands of candidate moduli per keygen round (DK:1313-1329
 * with batch_size in the thousands) — nothing but launches is enqueued, and the moduli may be the
 * output of the device-side reconstruction of DK:1284.  mod_bits / exp_bits: upper bounds on the bit
 * lengths (they select the lane geometry and the number of exponent digits).  The moduli must be
 * odd and >= 3; this cannot be checked for device operands — an even modulus yields an unspecified
 * residue for its group, never a hang.  Workspace: mx_powmod_workspace_bytes. */
int mx_powmod_multi_dev(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* d_mods, const uint32_t* d_exps,
                        int limbs, int exp_limbs, int mod_bits, int exp_bits, int64_t groups, int64_t group_size,
                        int limbs_per_lane, void* d_workspace, int64_t workspace_bytes, void* stream);

/* d_out[e] = d_bases[e] ^ exp mod N^2 with the modulus given by its ROOT h_n — the partial decryption
 * `pow_mod(ciphertext_value, exp, self.n_square)` of PSK:92 (n_square = n*n, PSK:46).  Same result
 * as mx_powmod_shared with the modulus N^2, computed with operations of the size of N only
 * (pairs x = rho*(X0 + X1*N), two half-size Montgomery passes per product; see mx_powmod_n2.hpp):
 * ~1.8x fewer multiply-accumulates and half the lanes per ciphertext.
 *   d_bases/d_out: [batch][limbs2] rows, limbs2 >= words of N^2;  h_n: limbs_n words;  h_exp: exp_limbs
 *   words (exponent >= 0); bases must be < N^2. */
int64_t mx_powmod_nsquare_workspace_bytes(int limbs_n, int exp_limbs, int64_t batch);
int mx_powmod_nsquare(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_n, const uint32_t* h_exp,
                      int limbs_n, int limbs2, int exp_limbs, int64_t batch, void* d_workspace,
                      int64_t workspace_bytes, void* stream);

/* ---- per-key plans ------------------------------------------------------------------------
 * Everything mx_powmod_nsquare derives from (N, exp) — the N-adic constant pairs, C', and the tape
 * (conversion, table of odd powers, sliding-window schedule) — depends on the KEY only: N is the
 * public modulus (PSK:46) and exp the party's Lagrange-folded share (PSK:79-85), both fixed for the
 * life of a PaillierSharedKey.  prepare derives them once (about 1 ms of host work) and writes them
 * into a caller-owned device block; run launches the modexp kernel and nothing else: no host
 * arithmetic, no operand upload.  The plan descriptor is a plain struct the caller keeps on the
 * host; the device block must stay valid, and the stream passed to prepare must be complete or
 * ordered before the streams passed to run (prepare's uploads are enqueued on it).
 * limbs_per_lane of run: 9 (narrow geometry), 18 (wide), 3 (latency geometry, two wavefronts per group only)
 * or 0 = the library's choice.
 * wavefronts_per_group of run: 1 = one wavefront executes both Montgomery passes of every pair product; 2 = the
 * passes run on two wavefronts of one workgroup, the second one operation behind the first (the chain of first
 * N-adic digits never reads the second digits) — 0.54-0.58 of the time per operation with twice the wavefronts,
 * for launches that leave SIMDs idle: a single ciphertext (PSK:92 called from DK:345-349), a keygen-sized batch,
 * one 10 000-ciphertext sequence; 4 (ABI 4.3; limbs_per_lane 3 or 0; moduli whose groups have 16, 32 or 64 lanes: key_length
 * 1024, 2048 and 4096 — ~800 .. 2560 and ~2800 .. 5500 bits, MX_ERR_SIZE elsewhere) = BOTH passes bipartite on two wavefronts each — four wavefronts per group of elements and a fifth that
 * forms the quotient correction one product behind —, the shortest dependent chain there is, for launches of at most one
 * workgroup per compute unit (a lone decrypt: 9.4 instead of 12.95 ms at key_length 2048, 31.8 instead of 47 at 4096); 0 = the library's choice (which takes 4 for such launches).  With both at 0 the library estimates the duration
 * of ONE launch of this batch on an idle GPU for every shape and takes the shortest; callers that keep several
 * launches in flight fill the machine between them and should pass 18 / 1.  mx_nsquare_launch_shape reports the
 * choice.  Same result bit for bit in every shape.
 * segments of run: the exponentiation is enqueued as this many consecutive launches, each executing a
 * stretch of the tape (the accumulator travels through the workspace); a wavefront then lives
 * 1/segments as long, which is the grain at which a burst of launches on several streams drains.
 * 1..64, 0 = automatic (4 for long exponents on large batches, else 1).  Same result bit for bit.
 * A two-wavefront launch with somewhat more groups of elements than the GPU holds at once runs in the time-sliced
 * form (mx_nsquare_launch_timesliced): ONE launch of resident workgroups whose wavefront pairs take the segments of
 * all groups from queues kept in the workspace (the group with the most work left first); `segments` (at most 16 then)
 * is the number of units per group. */
typedef struct mx_nsquare_plan {
  const void* d_plan;     /* device block written by prepare */
  int64_t plan_bytes;
  int32_t limbs_n;        /* words of N */
  int32_t n_bits;
  int32_t exp_bits;
  int32_t window;         /* sliding-window width of the tape (table of 2^(window-1) odd powers) */
  int32_t ntape;          /* tape words */
  int32_t n_sqr;          /* pair squarings one exponentiation executes */
  int32_t n_mul;          /* pair multiplications one exponentiation executes */
  int32_t geometries;     /* bit mask of the limbs_per_lane values with a kernel instance for this modulus: 1 = 9, 2 = 18, 4 = 3 */
  int32_t n_slot_reads;   /* pair slots (2 * limbs_per_lane words per lane) one exponentiation reads from ... */
  int32_t n_slot_writes;  /* ... and writes to the workspace: the kernel's HBM traffic model */
} mx_nsquare_plan;
int64_t mx_nsquare_plan_bytes(int limbs_n, int exp_limbs);
int mx_powmod_nsquare_prepare(mx_nsquare_plan* plan, const uint32_t* h_n, const uint32_t* h_exp, int limbs_n,
                              int exp_limbs, void* d_plan, int64_t plan_bytes, void* stream);
/* The same with options (flags = 0: exactly mx_powmod_nsquare_prepare).
 * MX_PLAN_FIXED_WINDOW: a fixed-window tape — w squarings and ONE multiplication per w-bit window, a window of zero
 * bits multiplying by the domain's one — instead of the sliding-window tape, whose runs of squarings and number of
 * multiplications are a function of the exponent's bits.  The exponent is the party's Lagrange-folded secret share
 * (PSK:79-85): with this flag the sequence and number of operations a launch executes, hence its duration and the
 * kernel-trace of a profiler, depend on the exponent's LENGTH only.  The table row a window reads is still selected by
 * the secret digit (addresses, not timing of the instruction stream).  Cost: 728 instead of 592 pair multiplications
 * for a 4197-bit exponent (w = 7: a 127-row table instead of 64 odd powers, twice the workspace) = +3.6 % instructions.  plan->window then reports w + 1 (the
 * table region is sized as 2^(window - 1) rows either way).  Same results bit for bit. */
#define MX_PLAN_FIXED_WINDOW 1
int mx_powmod_nsquare_prepare_ex(mx_nsquare_plan* plan, const uint32_t* h_n, const uint32_t* h_exp, int limbs_n,
                                 int exp_limbs, int flags, void* d_plan, int64_t plan_bytes, void* stream);
/* workspace of one run (the table of odd powers of every base; one per launch in flight) */
int64_t mx_powmod_nsquare_run_workspace_bytes(const mx_nsquare_plan* plan, int64_t batch);
int mx_powmod_nsquare_run(const mx_nsquare_plan* plan, const uint32_t* d_bases, uint32_t* d_out, int limbs2,
                          int64_t batch, int limbs_per_lane, int wavefronts_per_group, int segments,
                          void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- small-prime sieve -----------------------------------------------------------------
 * d_out[e] = 1 if some h_primes[k] divides candidate e else 0.  Replaces
 * `__small_prime_divisors_test(prime_list, n)` (DK:1197-1209) looped over the batch of
 * candidates at DK:1288-1292.  Primes must be odd, >= 3 and < 2^31; limbs <= 1024
 * (lists whose largest prime is below 2^21 — the reference's default threshold is 2000 — run without any
 * intermediate reduction; larger ones fold the 64-bit columns every floor(2^32 / max prime) - 1 limbs). */
int64_t mx_sieve_workspace_bytes(int limbs, int n_primes);
int mx_sieve(const uint32_t* d_candidates, uint8_t* d_out, const uint32_t* h_primes, int n_primes,
             int limbs, int64_t batch, void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- share recombination ---------------------------------------------------------------
 * For every ciphertext e:  x = prod_{i<n_partials} d_partials[i][e] mod N^2;
 *   d_status[e] = 1 and d_out[e] = 0           if (x - 1) % N != 0   (PSK:119-123 -> ValueError)
 *   d_status[e] = 0 and d_out[e] = ((x - 1) / N * theta_inv) % N     (PSK:125)
 * Replaces `PaillierSharedKey.decrypt` (PSK:95-127) looped at DK:510-515 (single: DK:378-380).
 *   d_partials: [n_partials][batch][limbs2] words, residues mod N^2 (players 1..degree+1 in order)
 *   h_n: limbs words (N; N^2 is derived), h_theta_inv: limbs words; limbs2 >= words of N^2
 *   d_out: [batch][limbs] words. */
int64_t mx_combine_workspace_bytes(int limbs, int limbs2, int n_partials, int64_t batch);
int mx_combine(const uint32_t* d_partials, uint32_t* d_out, uint8_t* d_status, const uint32_t* h_n,
               const uint32_t* h_theta_inv, int limbs, int limbs2, int n_partials, int64_t batch,
               void* d_workspace, int64_t workspace_bytes, void* stream);

/* Per-key plan of the recombination (N, N^2, the Montgomery constants and theta_inv depend on the
 * key only, PSK:46-50): prepare once, then run enqueues the kernel and nothing else.  Rows of d_out
 * are out_stride >= limbs words wide; if out_stride > limbs, word [limbs] of every row receives the
 * status (0 / 1) and the remaining words are zero, so that plaintext and status travel as ONE row
 * (one all-gather when ciphertexts are sharded over GPUs); d_status may then be NULL. */
typedef struct mx_combine_plan {
  const void* d_plan;
  int64_t plan_bytes;
  int32_t limbs, limbs2, n_bits, n2_bits;
} mx_combine_plan;
int64_t mx_combine_plan_bytes(int limbs, int limbs2);
int mx_combine_prepare(mx_combine_plan* plan, const uint32_t* h_n, const uint32_t* h_theta_inv, int limbs,
                       int limbs2, void* d_plan, int64_t plan_bytes, void* stream);
int mx_combine_run(const mx_combine_plan* plan, const uint32_t* d_partials, uint32_t* d_out, int out_stride,
                   uint8_t* d_status, int n_partials, int64_t batch, void* stream);

/* ---- biprimality verdict ---------------------------------------------------------------
 * d_pass[g*n_slots + k] = 1 iff  v_1 == +-prod_{i>=2} v_i (mod N_g) for test slot k, where
 * d_v is [n_parties][groups][n_slots][limbs] (party index 1 first).  Replaces the per-slot test
 * of __biprime_test_with_v_i (DK:1147-1158); the caller ANDs the slots (DK:1160-1172). */
int64_t mx_verdict_workspace_bytes(int limbs, int n_parties, int64_t groups, int64_t n_slots);
int mx_biprime_verdict(const uint32_t* d_v, uint8_t* d_pass, const uint32_t* h_mods, int limbs,
                       int n_parties, int64_t groups, int64_t n_slots, void* d_workspace,
                       int64_t workspace_bytes, void* stream);

/* The same with device-resident moduli (d_mods [groups][limbs], all odd and >= 3, at most mod_bits
 * bits); workspace: mx_verdict_workspace_bytes. */
int mx_biprime_verdict_dev(const uint32_t* d_v, uint8_t* d_pass, const uint32_t* d_mods, int limbs, int mod_bits,
                           int n_parties, int64_t groups, int64_t n_slots, void* d_workspace,
                           int64_t workspace_bytes, void* stream);

/* ---- modular multiplication ------------------------------------------------------------
 * d_out[e] = d_a[e] * d_b[e] mod h_mod.  The glue around the modexps: (1 + mN) * r^N of Paillier
 * encryption (the un-vendored tno.mpc.encryption_schemes.paillier used by the reference's tests,
 * test_distributed_keygen.py:125), homomorphic addition, and the product tree of the batched
 * modular inversion that replaces `mod_inv(ciphertext_value, n_square)` per ciphertext (PSK:89-91).
 * d_out may alias d_a or d_b. */
int64_t mx_mulmod_workspace_bytes(int limbs);
int mx_mulmod_shared(const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, const uint32_t* h_mod,
                     int limbs, int64_t batch, void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- Shamir-field arithmetic of the candidate moduli ---------------------------------------
 * The key generation holds p, q and a sharing of zero as Shamir shares modulo a prime P of
 * 2*(prime_length + log2(parties)) bits (DK:647-651).  Per candidate of a round:
 *   mx_fma_mod      d_out[e] = (d_a[e] * d_b[e] + d_c[e]) mod P — this party's share of N,
 *                   `prime_candidate_p * prime_candidate_q` then `candidate_n += zero` (DK:1274-1277;
 *                   ShamirVariable.__mul__ / __add__, UT:205-250)
 *   mx_lincomb_mod  d_out[e] = sum_t h_coeffs[t] * d_x[t][e] mod P — `candidate_n.reconstruct()`
 *                   (DK:1284; UT:263-270, 465-471) with the Lagrange coefficients at 0 of the
 *                   parties' evaluation points; d_x is [terms][batch][limbs], h_coeffs [terms][limbs].
 * P odd; operands < P (values up to 16 P are still reduced correctly).  The reconstructed moduli
 * are exactly the rows mx_sieve / mx_jacobi_dev / mx_powmod_multi_dev take. */
int64_t mx_field_workspace_bytes(int limbs, int terms);
int mx_fma_mod(const uint32_t* d_a, const uint32_t* d_b, const uint32_t* d_c, uint32_t* d_out, const uint32_t* h_mod,
               int limbs, int64_t batch, void* d_workspace, int64_t workspace_bytes, void* stream);
int mx_lincomb_mod(const uint32_t* d_x, const uint32_t* h_coeffs, uint32_t* d_out, const uint32_t* h_mod, int limbs,
                   int terms, int64_t batch, void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- modular inverse ------------------------------------------------------------------------
 * d_out[e] = d_values[e]^-1 mod h_mod, d_status[e] = 0; or d_out[e] = 0, d_status[e] = 1 when
 * gcd(value, modulus) != 1 (where `pow(v, -1, m)` raises ValueError).  Replaces `mod_inv(theta, n)`
 * of a key (PSK:50) and — as the root of a product tree of mx_mulmod_shared launches (Montgomery's
 * trick: 3 multiplications per element) — `mod_inv(ciphertext_value, n_square)` per ciphertext for
 * a negative Lagrange exponent (PSK:89-91).  One wavefront per element (the big integers are spread
 * over its 64 lanes): meant for a handful of elements, ~1-4 ms each at 2048-8200 bits. */
int64_t mx_modinv_workspace_bytes(int limbs);
int mx_modinv(const uint32_t* d_values, uint32_t* d_out, uint8_t* d_status, const uint32_t* h_mod, int limbs,
              int64_t batch, void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- Jacobi symbol ---------------------------------------------------------------------
 * d_out[g*group_size + k] = Jacobi symbol (d_values[g*group_size + k] / h_mods[g]) in {-1, 0, +1}.
 * Replaces the filter `sympy.jacobi_symbol(g, modulus) != 1` of the biprimality test (DK:1089),
 * evaluated for the up to 4*40 jointly random generators of every candidate (DK:1028, 1084-1099).
 * Values must be < their modulus (UT:361); moduli odd; limbs <= 257 (8224 bits: key_length 8192,
 * the widest key whose N^2 the modexp kernels take). */
int64_t mx_jacobi_workspace_bytes(int limbs, int64_t groups);
int mx_jacobi(const uint32_t* d_values, int8_t* d_out, const uint32_t* h_mods, int limbs, int64_t groups,
              int64_t group_size, void* d_workspace, int64_t workspace_bytes, void* stream);

/* The same with device-resident moduli (no workspace). */
int mx_jacobi_dev(const uint32_t* d_values, int8_t* d_out, const uint32_t* d_mods, int limbs, int64_t groups,
                  int64_t group_size, void* stream);
/* Only the rows [first, first + count) of every group (d_out entries outside the range are left
 * untouched); with d_skip_counts given, groups whose d_skip_counts[g] >= skip_threshold are not
 * evaluated at all.  The v-calculation stops at correct_param_biprime generators with symbol 1
 * (DK:1086): the head of the generator list almost always yields them, so the tail is evaluated
 * only for the candidates where it did not — the same selection as the reference, ~35 % fewer symbols. */
int mx_jacobi_dev_range(const uint32_t* d_values, int8_t* d_out, const uint32_t* d_mods, int limbs, int64_t groups,
                        int64_t group_size, int first, int count, const int32_t* d_skip_counts, int skip_threshold,
                        void* stream);

/* ---- selection of the generators -------------------------------------------------------
 * For every group, copies the first `keep` rows whose flag is 1 (in order) to d_out[g][0..keep) and
 * stores how many were found in d_counts[g]; unfilled rows are zero.  With d_flags = the output of
 * mx_jacobi this is `if jacobi_symbol(g, N) != 1: continue` + `stop at correct_param_biprime` of
 * the v-calculation loop (DK:1084-1099), so the generators never leave the device between the
 * Jacobi filter and mx_powmod_multi. */
int mx_select_first(const uint32_t* d_rows, const int8_t* d_flags, uint32_t* d_out, int32_t* d_counts,
                    int limbs, int64_t groups, int group_size, int keep, void* stream);

/* ---- diagnostics -----------------------------------------------------------------------
 * Runs the DPP cross-lane primitives against their ds_bpermute reference forms for every group
 * width on the current device; returns the number of mismatching lanes (0 = pass) or MX_ERR_*. */
int mx_selftest_lanes(void* stream);
/* Developer overrides, explicit calls only (the library reads no environment variables).  value 0 restores
 * the default.  MX_KNOB_N2_SEGMENTS: launches per mx_powmod_nsquare_run exponentiation when the caller passes
 * segments = 0 (1..64).  MX_KNOB_JACOBI_MAX_BATCHES: value v > 0 limits the Jacobi kernel to v - 1 divstep batches
 * so that its fallback kernel has to finish the symbols (test knob for the safety net).  MX_KNOB_N2_TIMESLICE: the
 * time-sliced form of two-wavefront launches (resident workgroups that share the groups of elements segment by
 * segment): 0 = where the estimate favours it, 1 = never, 2 = always, 16 + r = always, with r
 * workgroups per CU (r = 1..3).  MX_KNOB_N2_FRIENDLY_1W: 1 = the one-wavefront wide kernel never takes its
 * friendly-modulus instances (A/B runs against the plain ones).  MX_KNOB_GENERIC_LATENCY: 1 = the automatic geometry of
 * the generic-modulus modexp never takes the 3-limb latency instances (one or two wavefronts), 2 = never the bipartite form.  MX_KNOB_N2_SPLIT: mx_nsquare_launch_split
 * 1 = never reports a split, 2 = whenever one exists.  These are the ONLY process-wide settings the library has (ABI 4.0
 * dropped mx_set_limbs_per_lane: launch shapes are call arguments, the entry points without one leave the choice to the
 * library); each is an atomic integer, so flipping one while another thread launches is well defined (that launch sees
 * the old or the new value).  Production callers never need them.  Returns MX_OK / MX_ERR_ARG. */
#define MX_KNOB_N2_SEGMENTS 1
#define MX_KNOB_JACOBI_MAX_BATCHES 2
#define MX_KNOB_N2_TIMESLICE 3
#define MX_KNOB_N2_FRIENDLY_1W 4
#define MX_KNOB_GENERIC_LATENCY 5
#define MX_KNOB_N2_SPLIT 6
#define MX_KNOB_LAT_LANES 8         /* 3-limb latency forms of the generic kernel: at least this many lanes per element (0 = the smallest group that holds the number) */
#define MX_KNOB_N2_BIPAIR 9         /* 1 = mx_powmod_nsquare_run never chooses the five-wavefront latency form by itself (A/B runs) */
#define MX_KNOB_BI_PIVOT 7          /* bipartite form of the generic kernel: multiplier limbs on the Montgomery wavefront (0 = the library's pivot) */
int mx_debug_knob(int knob, int value);
/* Enqueues a kernel of ONE wavefront that idles for `microseconds` (0..1 000 000) on `stream` and touches no
 * memory.  A concurrency probe: two streams that the HIP runtime has mapped to the same hardware queue run
 * their spins one after the other, two streams on different queues run them side by side — which is what
 * decides whether chunks of a batch launched on those streams fill the machine together or serialise
 * (GPU_MAX_HW_QUEUES is read once when the runtime initialises and cannot be queried).  Never synchronises. */
int mx_spin(int64_t microseconds, void* stream);
/* Streams confined to a slice of the compute units, for callers that keep several SMALL launches in flight: the
 * dispatcher places concurrent launches of a few dozen workgroups each on the same CUs of every XCD (four 64-workgroup
 * launches on four ordinary streams ran 1.7x longer than one of them alone), while launches on streams whose CU
 * masks are disjoint each get their own part of the chip.  Slice k of n is the same CU range [k*CUs/n, (k+1)*CUs/n) of
 * the mask in EVERY XCD (mask bit i = CU i/8 of XCD i%8 on MI355X; a mask that empties an XCD is ignored by the
 * runtime), n <= 8.  The stream belongs to the caller (mx_stream_destroy).  Not for launches that fill the machine
 * on their own: a static partition cannot balance load. */
int mx_stream_create_cu_slice(int slice, int n_slices, int reserved, void** stream_out);
int mx_stream_destroy(void* stream);
/* Enqueues a kernel of ONE wavefront that idles for `microseconds` and writes two counters to d_ticks[0..1]: the
 * advance of the shader clock counter and of the 100 MHz real-time counter over that interval; their ratio x 100 MHz
 * is the clock the SIMDs run at under the load that is in flight at that moment (MI355X drops from its nominal
 * 2.4 GHz to ~2.1 GHz when every SIMD streams multiply-accumulates: profiles/r03_ubench_valu_peak.txt).  Launch it
 * on a stream of its own while the work of interest runs.  Never synchronises. */
int mx_clock_probe(int64_t microseconds, uint64_t* d_ticks, void* stream);
/* Engine geometry chosen for a modulus of `mod_bits` bits: lanes per element (K), limbs per lane
 * (L), limb width (W) and Montgomery blocks; returns MX_OK or MX_ERR_SIZE. */
int mx_geometry(int mod_bits, int* lanes_per_element, int* limbs_per_lane, int* limb_bits, int* blocks);
/* Kernel timing for benchmarks.  mx_profile(1): every following mx_powmod_shared / mx_powmod_multi /
 * mx_powmod_nsquare call records two events on its stream around its modexp kernel (after the operand
 * uploads; mx_powmod_nsquare_run and mx_powmod_multi_dev are timed too).  mx_profile_collect waits for all recorded launches and returns the sum of their durations
 * and their number, then forgets them.  mx_profile(0) stops recording.  Returns MX_OK / MX_ERR_HIP. */
int mx_profile(int enable);
int mx_profile_collect(double* total_ms, int* launches);
/* Geometry mx_powmod_nsquare launches for a modulus N of `n_bits` bits and `batch` bases (it depends
 * on the batch: the wide geometry is chosen when one launch brings enough wavefronts); returns MX_OK or
 * MX_ERR_SIZE. */
int mx_nsquare_geometry(int n_bits, int64_t batch, int* lanes_per_element, int* limbs_per_lane, int* limb_bits,
                        int* blocks);

/* Geometry for an explicit limbs_per_lane (9 | 18 | 0 = the library's automatic choice for this
 * batch): of a mx_powmod_nsquare_run launch, and of a
 * mx_powmod_shared_lpl (groups = 1) / mx_powmod_multi_dev (groups > 1) launch. */
int mx_nsquare_geometry_for(int n_bits, int64_t batch, int limbs_per_lane, int* lanes_per_element,
                            int* limbs_per_lane_out, int* limb_bits, int* blocks);
/* The full launch shape mx_powmod_nsquare_run uses for (limbs_per_lane, wavefronts_per_group), either or both 0 =
 * the library's choice for this batch: the geometry as above plus the wavefronts per group of elements (1 | 2). */
int mx_nsquare_launch_shape(int n_bits, int64_t batch, int limbs_per_lane, int wavefronts_per_group,
                            int* lanes_per_element, int* limbs_per_lane_out, int* limb_bits, int* blocks,
                            int* wavefronts_per_group_out);
/* For callers that keep SEVERAL launches in flight which together process `total` elements (several streams, several
 * keys): the shape to pass to each of them — the plain form with the lowest estimate for one launch of the total, never a
 * time-sliced one (only a lone launch can be that).  Arguments as mx_nsquare_launch_shape; ABI 4.2. */
int mx_nsquare_pieces_shape(int n_bits, int64_t total, int limbs_per_lane, int wavefronts_per_group,
                            int* limbs_per_lane_out, int* wavefronts_per_group_out);
/* Whether mx_powmod_nsquare_run runs this launch in the time-sliced form of the two-wavefront kernel: a fixed number
 * of resident workgroups per CU that take (segment, group of elements) units from a queue in the workspace, chosen
 * when a lone launch has somewhat more groups than the GPU holds at once (e.g. 10 000 ciphertexts at key_length
 * 2048: 41 ms instead of 52-60; 18 limbs per lane, one workgroup per CU, 8 units per group).  *resident_per_cu = 0:
 * the plain launch; otherwise the workgroups per CU and
 * *units_per_group the segments each group is cut into when run is called with segments = 0. */
int mx_nsquare_launch_timesliced(int n_bits, int64_t batch, int limbs_per_lane, int wavefronts_per_group,
                                 int* resident_per_cu, int* units_per_group);
/* The kernel INSTANCE behind that launch, for tests that must see every instance: geometry and wavefronts per group as
 * mx_nsquare_launch_shape reports them, *friendly = 1 if the tape runs modulo the friendly multiple of N (a different
 * template instance: every 3-limb one, the 9-limb two-wavefront ones for groups of 8 / 16 lanes and the 18-limb
 * one-wavefront ones for groups of 4 / 8 lanes where the modulus leaves the room), *timesliced = 1 for the time-sliced
 * form.  A friendly one-wavefront launch also runs the plain instance of the same geometry (last product, epilogue). */
/* The five-wavefront latency form (wavefronts_per_group 4) for moduli of n_bits bits (ABI 4.4): MX_ERR_SIZE where it has no
 * instance; otherwise *lanes per element (16, 32 or 64), the data *positions Pd of its rows, its *pivot (multiplier limbs on the
 * least-significant-first wavefronts: 0.52 of Pd + 3 to the nearest multiple of 3 — tools/bipair_model.py: pair_geometry) and the
 * largest batch (*max_batch: one workgroup per compute unit of the current device) for which mx_powmod_nsquare_run takes
 * the form by itself.  Any of the pointers may be NULL. */
int mx_nsquare_latency_form(int n_bits, int* lanes, int* positions, int* pivot, int64_t* max_batch);
int mx_nsquare_launch_instance(int n_bits, int64_t batch, int limbs_per_lane, int wavefronts_per_group,
                               int* lanes_per_element, int* limbs_per_lane_out, int* wavefronts_per_group_out,
                               int* friendly, int* timesliced);
/* For callers that own a second stream: whether ONE batch is better run as two launches side by side — the first
 * *first_rows elements in the shape (*first_lpl, *first_wpg), the rest in (*rest_lpl, *rest_wpg) on another stream at the
 * same time (each with its own workspace; mx_powmod_nsquare_run with explicit shapes).  *first_rows = 0: no split.
 * Since ABI 4.1 the library reports a split only when MX_KNOB_N2_SPLIT = 2 asks for it (then for every batch between
 * one and two capacities of the wide two-wavefront shape — 8192 ciphertexts at key_length 2048): the time-sliced wide
 * launch covers 8192 .. 12 288 ciphertexts in 36-48 ms, as fast as or faster than the split (48.5).  The library itself
 * never uses a stream the caller did not pass; protocols/distributed_keygen_amd/engine.py follows this hint. */
int mx_nsquare_launch_split(int n_bits, int64_t batch, int64_t* first_rows, int* first_lpl, int* first_wpg,
                            int* rest_lpl, int* rest_wpg);
/* Launch form of a generic-modulus modexp (mx_powmod_shared_lpl / mx_powmod_multi_dev) for this limbs_per_lane (0 = the
 * library's choice): *wavefronts_per_group = 1, or 2 for the BIPARTITE latency form (limbs_per_lane 6 = "3 limbs per lane on
 * two wavefronts", csrc/mx_bimont.hpp): every modular product is split at a pivot — *pivot multiplier limbs on a wavefront that
 * runs least-significant-first Montgomery steps, the rest on a second wavefront that runs most-significant-first steps with a
 * fold of the overflow — so that a product costs half the dependent limb steps plus a hand-over through LDS.  For launches that
 * leave SIMDs idle (a key-generation round at the reference's batch sizes: a few dozen to a thousand modexps,
 * distributed_keygen.py:1313-1329): such a launch lasts as long as one wavefront's dependent chain whatever its size.  Fixed
 * windows; moduli up to 5359 bits; mx_powmod_geometry_for reports its lanes per element with limbs_per_lane_out = 3. */
int mx_powmod_launch_form(int mod_bits, int64_t batch, int64_t groups, int limbs_per_lane, int* wavefronts_per_group, int* pivot);
int mx_powmod_geometry_for(int mod_bits, int64_t batch, int64_t groups, int limbs_per_lane, int* lanes_per_element,
                           int* limbs_per_lane_out, int* limb_bits, int* blocks);

#ifdef __cplusplus
}
#endif
#endif /* MXPAILLIER_H */
