// Biprimality-test verdict per (candidate, test slot):  v_1 == +-prod_{i>=2} v_i (mod N).
// Replaces the per-slot body of __biprime_test_with_v_i (distributed_keygen.py:1147-1158);
// the caller ANDs the slots of a candidate (distributed_keygen.py:1160-1172).
#pragma once
#include "mx_mont.hpp"
#include "mx_prio.hpp"

namespace mx {

struct VerdictArgs {
  const u32* v;          // [n_parties][groups][n_slots][limbs] device (party index 1 first)
  unsigned char* pass;   // [groups][n_slots] device
  const u32* mods;       // [groups][limbs] device
  const u32* rmodn;      // [groups][limbs] device
  long long groups, n_slots;
  int limbs, n_parties, nblk;
};

template <int K, int L, int W>
__global__ void __launch_bounds__(64) verdict_kernel(VerdictArgs A) {
  aux_wave_priority();
  using M_t = Mont<K, L, W, true>;
  extern __shared__ u32 smem[];
  constexpr int GPW = 64 / K;
  const int gw = threadIdx.x / K;
  const long long total = A.groups * A.n_slots;
  const long long elem_raw = (long long)blockIdx.x * GPW + gw;
  const bool valid = elem_raw < total;
  const long long elem = valid ? elem_raw : total - 1;
  const long long grp = elem / A.n_slots;

  M_t M;
  M.init(smem + gw * M_t::LDS_WORDS, A.nblk);
  M.load(M.n, A.mods + grp * A.limbs, A.limbs);
  M.setup_modulus();
  u32 one_m[L], r2[L];
  M.load(one_m, A.rmodn + grp * A.limbs, A.limbs);
  M.compute_r2(r2, one_m);

  u32 prod[L];
#pragma unroll
  for (int j = 0; j < L; ++j) prod[j] = one_m[j];
  for (int i = 1; i < A.n_parties; ++i) {
    u32 y[L];
    M.load(y, A.v + ((long long)i * total + elem) * A.limbs, A.limbs);
    M.mul(y, y, r2);
    M.mul(prod, prod, y);
  }
  u32 pc[L], v1[L];
  M.from_mont_canonical(pc, prod);               // product % N
  M.load(v1, A.v + elem * A.limbs, A.limbs);
  M.mul(v1, v1, r2);
  M.from_mont_canonical(v1, v1);                 // value1 % N
  const bool same = M.equal(v1, pc);
  u32 s[L];
  {
    u64 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = (u64)v1[j] + pc[j];
    M.normalize_full(s, t);
  }
  const bool neg = M.equal(s, M.n);              // value1 == (-product) % N  (product != 0)
  if (valid && M.p == 0) A.pass[elem] = (same || neg) ? 1 : 0;
}

}  // namespace mx
