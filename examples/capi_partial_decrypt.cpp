// The C ABI used from plain C++/HIP, no Python and no PyTorch: a batch of partial decryptions
// c^exp mod N^2 through the per-key plan (mx_powmod_nsquare_prepare once, mx_powmod_nsquare_run per
// batch: pairs modulo N), through the one-shot form mx_powmod_nsquare and, as a cross-check, through
// mx_powmod_shared on the modulus N^2 — independent kernels that must agree bit for bit — and c^1 = c.
// (N is just an odd 2048-bit number here; parity with the reference is what tests/ check.)
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude examples/capi_partial_decrypt.cpp \
//         -Lprotocols/distributed_keygen_amd -lmxpaillier -Wl,-rpath,$PWD/protocols/distributed_keygen_amd \
//         -o /tmp/capi_partial_decrypt && /tmp/capi_partial_decrypt
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "mxpaillier.h"

#define CHECK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)
#define MX(x) do { int rc_ = (x); if (rc_ != MX_OK) { std::fprintf(stderr, "%s -> %s (%s)\n", #x, mx_error_string(rc_), mx_last_hip_error()); return 3; } } while (0)

// little-endian schoolbook square: out[2n] = a[n]^2
static void square(const std::vector<uint32_t>& a, std::vector<uint32_t>& out) {
  size_t n = a.size();
  out.assign(2 * n, 0);
  for (size_t i = 0; i < n; ++i) {
    uint64_t carry = 0;
    for (size_t j = 0; j < n; ++j) {
      uint64_t t = (uint64_t)a[i] * a[j] + out[i + j] + carry;
      out[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    out[i + n] = (uint32_t)carry;
  }
}

int main() {
  const int limbs_n = 65, limbs2 = 130, exp_limbs = 132;       // key_length 2048: N < 2^2051, exponent < 2^4224
  const int64_t batch = 2048;
  std::mt19937_64 rng(2048);
  std::vector<uint32_t> n(limbs_n), exp(exp_limbs), n2;
  for (auto& w : n) w = (uint32_t)rng();
  n[0] |= 1u; n[limbs_n - 1] = 0x5u;                           // odd, 2051 bits
  for (auto& w : exp) w = (uint32_t)rng();
  exp[exp_limbs - 1] = 0; exp[exp_limbs - 2] &= 0x1Fu;          // 4197 bits
  square(n, n2);
  std::vector<uint32_t> bases((size_t)batch * limbs2);
  for (int64_t e = 0; e < batch; ++e) {
    for (int j = 0; j < limbs2; ++j) bases[e * limbs2 + j] = (uint32_t)rng();
    bases[e * limbs2 + limbs2 - 1] = 0; bases[e * limbs2 + limbs2 - 2] &= 0xFu;   // < 2^4100 < N^2
  }
  uint32_t *d_in, *d_a, *d_b;
  size_t bytes = bases.size() * 4;
  CHECK(hipMalloc(&d_in, bytes)); CHECK(hipMalloc(&d_a, bytes)); CHECK(hipMalloc(&d_b, bytes));
  CHECK(hipMemcpy(d_in, bases.data(), bytes, hipMemcpyHostToDevice));
  int64_t ws1 = mx_powmod_nsquare_workspace_bytes(limbs_n, exp_limbs, batch);
  int64_t ws2 = mx_powmod_workspace_bytes(limbs2, exp_limbs, batch, 1);
  if (ws1 < 0 || ws2 < 0) { std::fprintf(stderr, "workspace query failed\n"); return 3; }
  void *w1, *w2;
  CHECK(hipMalloc(&w1, ws1)); CHECK(hipMalloc(&w2, ws2));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t t0, t1;
  CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
  CHECK(hipEventRecord(t0, s));
  MX(mx_powmod_nsquare(d_in, d_a, n.data(), exp.data(), limbs_n, limbs2, exp_limbs, batch, w1, ws1, s));
  CHECK(hipEventRecord(t1, s));
  MX(mx_powmod_shared(d_in, d_b, n2.data(), exp.data(), limbs2, exp_limbs, batch, w2, ws2, s));
  CHECK(hipStreamSynchronize(s));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, t0, t1));
  std::vector<uint32_t> a(bases.size()), b(bases.size());
  CHECK(hipMemcpy(a.data(), d_a, bytes, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(b.data(), d_b, bytes, hipMemcpyDeviceToHost));
  if (a != b) { std::fprintf(stderr, "mx_powmod_nsquare and mx_powmod_shared disagree\n"); return 1; }
  // the same through the per-key plan: prepare once (what a PaillierSharedKey would do in __init__),
  // then every batch is launches only — here two batches on the same plan, explicit geometry and segments
  mx_nsquare_plan plan;
  int64_t pb = mx_nsquare_plan_bytes(limbs_n, exp_limbs);
  void* d_plan;
  CHECK(hipMalloc(&d_plan, pb));
  MX(mx_powmod_nsquare_prepare(&plan, n.data(), exp.data(), limbs_n, exp_limbs, d_plan, pb, s));
  int64_t wr = mx_powmod_nsquare_run_workspace_bytes(&plan, batch);
  if (wr < 0 || wr > ws1) { std::fprintf(stderr, "run workspace query failed\n"); return 3; }
  MX(mx_powmod_nsquare_run(&plan, d_in, d_b, limbs2, batch, 18, 1, 4, w1, ws1, s));     // wide lanes, one wavefront per group, 4 segments
  CHECK(hipStreamSynchronize(s));
  CHECK(hipMemcpy(b.data(), d_b, bytes, hipMemcpyDeviceToHost));
  if (a != b) { std::fprintf(stderr, "plan run (wide, 4 segments) differs from the one-shot form\n"); return 1; }
  MX(mx_powmod_nsquare_run(&plan, d_in, d_b, limbs2, batch, 0, 0, 1, w1, ws1, s));      // the library's choice of shape, one launch
  CHECK(hipStreamSynchronize(s));
  CHECK(hipMemcpy(b.data(), d_b, bytes, hipMemcpyDeviceToHost));
  if (a != b) { std::fprintf(stderr, "plan run (narrow) differs from the one-shot form\n"); return 1; }
  std::vector<uint32_t> one(exp_limbs, 0);
  one[0] = 1;
  MX(mx_powmod_nsquare(d_in, d_a, n.data(), one.data(), limbs_n, limbs2, exp_limbs, batch, w1, ws1, s));
  CHECK(hipStreamSynchronize(s));
  CHECK(hipMemcpy(a.data(), d_a, bytes, hipMemcpyDeviceToHost));
  if (a != bases) { std::fprintf(stderr, "c^1 != c\n"); return 1; }
  std::printf("ABI %d: %lld partial decryptions (key_length 2048) in %.1f ms, identical through the one-shot form, the "
              "per-key plan (%d squarings, %d multiplications on its tape) and the generic kernel; c^1 == c\n",
              mx_version(), (long long)batch, ms, plan.n_sqr, plan.n_mul);
  return 0;
}
