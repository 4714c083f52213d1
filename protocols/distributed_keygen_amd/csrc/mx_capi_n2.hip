// extern "C" entry points of the N^2-modulus modexp (second translation unit of libmxpaillier.so:
// compiled in parallel with mx_capi.hip, the pair kernels are the most expensive to build).
#include "mx_upload.hpp"
#include "mx_powmod_n2.hpp"

// ---- modexp modulo N^2 through pairs modulo N --------------------------------------------------
namespace {
// Shape of one launch: geometry, grid, table of pair slots in the caller's workspace.
struct N2Shape {
  Geometry geo;
  int64_t nblocks = 0, nlanes = 0;
  int nslots = 0;
  int64_t table_bytes = 0;
};

bool shape_n2(int n_bits, int window, int64_t batch, int limbs_per_lane, N2Shape& p) {
  if (!choose_geometry(n_bits, p.geo, limbs_per_lane)) return false;
  if (p.geo.K > (limbs_per_lane == LIMBS_PER_LANE_WIDE ? 16 : 32)) return false;   // instances that exist
  int gpw = 64 / p.geo.K;
  p.nblocks = (batch + gpw - 1) / gpw;
  p.nlanes = p.nblocks * 64;
  p.nslots = mx::N2_SLOT_TABLE + (1 << (window - 1));
  p.table_bytes = align256((int64_t)p.nslots * 2 * p.geo.L * p.nlanes * 4);
  return true;
}

template <int K, int L>
int launch_n2_kl(const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  size_t lds = mx::powmod_n2_lds_bytes<K, L>();
  hipLaunchKernelGGL((mx::powmod_n2_kernel<K, L, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

}  // namespace
// wide-geometry instantiations live in mx_capi_n2w.hip (third translation unit, built in parallel)
namespace mxw { int launch_n2_wide(int K, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s); }
namespace {
template <int K>
int launch_n2_k(const mx::PowmodN2Args& a, int64_t nblocks, int lpl, hipStream_t s) {
  if (lpl == LIMBS_PER_LANE_WIDE) return mxw::launch_n2_wide(K, a, nblocks, s);
  return launch_n2_kl<K, LIMBS_PER_LANE>(a, nblocks, s);
}

// Geometry of the pair kernel when the caller leaves the choice to the library.  Measured on MI355X
// (tools/ab_geometry.sh, tools/ab_streams.sh, tools/sweep_keys.sh, profiles/): the wide geometry issues
// 18 % fewer instructions per element, runs 2 wavefronts per SIMD and puts twice the elements into a
// wavefront.  At key_length 2048 it is faster for a lone 10 000-element launch (164 k vs 144 k
// modexps/s), with four or more launches in flight (262-272 k vs 241-255 k) and saturated (300 k vs
// 262 k); for small launches the narrow geometry, which makes twice the wavefronts, fills the
// machine better (2 000 elements: 80 k vs 55 k).  So: wide from ~480 wavefronts per launch.
int n2_auto_limbs_per_lane(int n_bits, int64_t batch) {
  Geometry narrow, wide;
  if (!choose_geometry(n_bits, narrow, LIMBS_PER_LANE) || !choose_geometry(n_bits, wide, LIMBS_PER_LANE_WIDE))
    return LIMBS_PER_LANE;
  if (narrow.K < 8 || wide.K > 16) return LIMBS_PER_LANE;
  const int64_t waves = (batch * wide.K + 63) / 64;
  return waves >= 480 ? LIMBS_PER_LANE_WIDE : LIMBS_PER_LANE;
}

// Segments when the caller leaves the choice to the library: a launch whose wavefronts would live for
// tens of milliseconds is cut so that a burst of such launches drains at a finer grain (measured with
// bench.py --steps 20: the last round of 4 launches in flight costs ~3 % of the run unsegmented).
int n2_auto_segments(int n_sqr, int64_t nblocks) {
  if (g_knob_n2_segments >= 1) return g_knob_n2_segments;
  return (n_sqr >= 2048 && nblocks >= 256) ? 4 : 1;
}

inline int64_t n2_consts_bytes(int limbs_n) { return align256((int64_t)8 * limbs_n * 4); }

// The eight constant rows of one geometry (R = 2^m), each limbs_n words:
//   N | ONE0 ONE1 | K1_0 K1_1 | K2_0 K2_1 | C'
void n2_constants(u32* c, const u32* h_n, int limbs_n, int m, int k) {
  const int l2 = 2 * limbs_n;
  std::vector<u32> n2(l2), tmp(l2), qq(l2), rr(limbs_n);
  mul_words(n2.data(), h_n, limbs_n, h_n, limbs_n);
  std::memset(c, 0, (size_t)8 * limbs_n * 4);
  std::memcpy(&c[0], h_n, (size_t)limbs_n * 4);
  auto pair_of = [&](int pow2, int row) {                      // N-adic digits of 2^pow2 mod N^2
    pow2_mod(tmp.data(), n2.data(), l2, pow2);
    divmod_words(qq.data(), rr.data(), tmp.data(), l2, h_n, limbs_n);
    std::memcpy(&c[(size_t)row * limbs_n], rr.data(), (size_t)limbs_n * 4);          // digit 0
    std::memcpy(&c[(size_t)(row + 1) * limbs_n], qq.data(), (size_t)limbs_n * 4);    // digit 1 (< N)
  };
  pair_of(m, 1);              // represents 1      (rho * V = 1  ->  V = R)
  pair_of(2 * m, 3);          // represents R      (V = R^2)
  pair_of(2 * m + k, 5);      // represents 2^k R  (V = 2^k R^2)
  // C' = N*ceil(R/N) - R + 1 = N - (R mod N) + 1   (R mod N != 0 as N is odd > 1)
  two_pow_mod(rr.data(), h_n, limbs_n, m);
  u64 borrow = 0, carry = 1;
  u32* cp = &c[(size_t)7 * limbs_n];
  for (int i = 0; i < limbs_n; ++i) {
    u64 d = (u64)h_n[i] - rr[i] - borrow;
    borrow = (d >> 63) & 1;
    u64 e = (u64)(u32)d + carry;
    cp[i] = (u32)e;
    carry = e >> 32;
  }
}
}  // namespace

extern "C" int mx_nsquare_geometry_for(int n_bits, int64_t batch, int limbs_per_lane, int* k, int* l, int* w,
                                       int* blocks) {
  if (!k || !l || !w || !blocks || batch <= 0) return MX_ERR_ARG;
  if (limbs_per_lane != 0 && limbs_per_lane != LIMBS_PER_LANE && limbs_per_lane != LIMBS_PER_LANE_WIDE) return MX_ERR_ARG;
  N2Shape p;
  const int lpl = limbs_per_lane ? limbs_per_lane : n2_auto_limbs_per_lane(n_bits, batch);
  if (!shape_n2(n_bits, 1, batch, lpl, p)) return MX_ERR_SIZE;
  *k = p.geo.K; *l = p.geo.L; *w = p.geo.W; *blocks = p.geo.nblk;
  return MX_OK;
}

extern "C" int mx_nsquare_geometry(int n_bits, int64_t batch, int* k, int* l, int* w, int* blocks) {
  return mx_nsquare_geometry_for(n_bits, batch, override_limbs_per_lane(), k, l, w, blocks);
}

extern "C" int64_t mx_nsquare_plan_bytes(int limbs_n, int exp_limbs) {
  if (limbs_n <= 0 || exp_limbs <= 0) return MX_ERR_ARG;
  return 2 * n2_consts_bytes(limbs_n) + align256((int64_t)MAX_SLIDING_OPS * 4);
}

extern "C" int mx_powmod_nsquare_prepare(mx_nsquare_plan* plan, const uint32_t* h_n, const uint32_t* h_exp,
                                         int limbs_n, int exp_limbs, void* d_plan, int64_t plan_bytes, void* stream) {
  if (!plan || !h_n || !h_exp || !d_plan) return MX_ERR_ARG;
  if (limbs_n <= 0 || exp_limbs <= 0) return MX_ERR_ARG;
  if (!(h_n[0] & 1u)) return MX_ERR_MODULUS;
  const int bits = bit_length(h_n, limbs_n);
  if (bits < 2) return MX_ERR_MODULUS;
  Geometry narrow, wide;
  if (!choose_geometry(bits, narrow, LIMBS_PER_LANE) || narrow.K > 32) return MX_ERR_SIZE;
  const bool has_wide = choose_geometry(bits, wide, LIMBS_PER_LANE_WIDE) && wide.K <= 16;
  if (mx_nsquare_plan_bytes(limbs_n, exp_limbs) > plan_bytes) return MX_ERR_WORKSPACE;
  const int ebits = bit_length(h_exp, exp_limbs);
  const int k = bits - 1;                                      // x = x_lo + 2^k x_hi
  // ---- the tape (mx_powmod_n2.hpp): conversion, table of odd powers, sliding window, times E
  std::vector<u32> tape;
  int n_sqr = 0, n_mul = 0, w = 1;
  int n_reads = 0, n_writes = 6;                               // the prologue writes 4 constant pairs and x_lo, x_hi
  auto emit = [&](u32 op, int arg) {
    tape.push_back((op << 28) | (u32)arg);
    if (op == mx::N2_SQR) n_sqr += arg;
    if (op == mx::N2_MUL) n_mul += 1;
    if (op == mx::N2_STORE) n_writes += 1;
    if (op == mx::N2_MUL || op == mx::N2_ADD || op == mx::N2_LOAD) n_reads += 1;
  };
  if (ebits == 0) {
    emit(mx::N2_LOAD, mx::N2_SLOT_ONE);
  } else {
    w = sliding_window(ebits);
    std::vector<SlidingOp> ops = sliding_schedule(h_exp, exp_limbs, w);
    // x = (x_lo, 0) * K1 + (x_hi, 0) * K2
    emit(mx::N2_LOAD, mx::N2_SLOT_LO); emit(mx::N2_MUL, mx::N2_SLOT_K1); emit(mx::N2_STORE, mx::N2_SLOT_TMP);
    emit(mx::N2_LOAD, mx::N2_SLOT_HI); emit(mx::N2_MUL, mx::N2_SLOT_K2); emit(mx::N2_ADD, mx::N2_SLOT_TMP);
    emit(mx::N2_STORE, mx::N2_SLOT_TABLE);
    const int nodd = 1 << (w - 1);
    if (nodd > 1) {
      emit(mx::N2_SQR, 1); emit(mx::N2_STORE, mx::N2_SLOT_SQ); emit(mx::N2_LOAD, mx::N2_SLOT_TABLE);
      for (int t = 1; t < nodd; ++t) { emit(mx::N2_MUL, mx::N2_SLOT_SQ); emit(mx::N2_STORE, mx::N2_SLOT_TABLE + t); }
    }
    emit(mx::N2_LOAD, mx::N2_SLOT_TABLE + (int)ops[0].index1 - 1);
    for (size_t t = 1; t < ops.size(); ++t) {
      if (ops[t].squarings) emit(mx::N2_SQR, ops[t].squarings);     // 28-bit repeat count on the tape
      if (ops[t].index1) emit(mx::N2_MUL, mx::N2_SLOT_TABLE + (int)ops[t].index1 - 1);
    }
  }
  emit(mx::N2_MUL, mx::N2_SLOT_E);
  if ((int)tape.size() > MAX_SLIDING_OPS) return MX_ERR_SIZE;

  const int64_t cb = n2_consts_bytes(limbs_n);
  std::vector<u32> c((size_t)8 * limbs_n);
  hipStream_t s = (hipStream_t)stream;
  char* dp = (char*)d_plan;
  n2_constants(c.data(), h_n, limbs_n, narrow.W * narrow.L * narrow.nblk, k);
  MX_TRY(upload_words(dp, c.data(), c.size(), s));
  if (has_wide) {
    n2_constants(c.data(), h_n, limbs_n, wide.W * wide.L * wide.nblk, k);
    MX_TRY(upload_words(dp + cb, c.data(), c.size(), s));
  }
  MX_TRY(upload_words(dp + 2 * cb, tape.data(), tape.size(), s));
  plan->d_plan = d_plan;
  plan->plan_bytes = plan_bytes;
  plan->limbs_n = limbs_n;
  plan->n_bits = bits;
  plan->exp_bits = ebits;
  plan->window = w;
  plan->ntape = (int)tape.size();
  plan->n_sqr = n_sqr;
  plan->n_mul = n_mul;
  plan->has_wide = has_wide ? 1 : 0;
  plan->n_slot_reads = n_reads;
  plan->n_slot_writes = n_writes;
  return MX_OK;
}

extern "C" int64_t mx_powmod_nsquare_run_workspace_bytes(const mx_nsquare_plan* plan, int64_t batch) {
  if (!plan || batch <= 0 || plan->window < 1) return MX_ERR_ARG;
  N2Shape p, q;
  if (!shape_n2(plan->n_bits, plan->window, batch, LIMBS_PER_LANE, p)) return MX_ERR_SIZE;
  if (shape_n2(plan->n_bits, plan->window, batch, LIMBS_PER_LANE_WIDE, q) && q.table_bytes > p.table_bytes)
    return q.table_bytes;
  return p.table_bytes;
}

extern "C" int mx_powmod_nsquare_run(const mx_nsquare_plan* plan, const uint32_t* d_bases, uint32_t* d_out,
                                     int limbs2, int64_t batch, int limbs_per_lane, int segments, void* d_ws,
                                     int64_t ws_bytes, void* stream) {
  if (!plan || !plan->d_plan || !d_bases || !d_out || !d_ws) return MX_ERR_ARG;
  if (limbs2 <= 0 || batch <= 0 || plan->limbs_n <= 0 || plan->ntape <= 0) return MX_ERR_ARG;
  if (limbs_per_lane != 0 && limbs_per_lane != LIMBS_PER_LANE && limbs_per_lane != LIMBS_PER_LANE_WIDE) return MX_ERR_ARG;
  if (segments < 0 || segments > 64) return MX_ERR_ARG;
  const int bits = plan->n_bits;
  if (2 * bits - 1 > 32 * limbs2) return MX_ERR_ARG;          // rows too narrow for N^2
  const int lpl = limbs_per_lane ? limbs_per_lane : n2_auto_limbs_per_lane(bits, batch);
  N2Shape p;
  if (!shape_n2(bits, plan->window, batch, lpl, p)) return MX_ERR_SIZE;
  if (lpl == LIMBS_PER_LANE_WIDE && !plan->has_wide) return MX_ERR_SIZE;
  if (p.table_bytes > ws_bytes) return MX_ERR_WORKSPACE;
  if (2 * p.geo.K * p.geo.L + 8 < limbs2 + 2) return MX_ERR_ARG;   // row wider than the staging area
  const int64_t cb = n2_consts_bytes(plan->limbs_n);
  const char* dp = (const char*)plan->d_plan;
  mx::PowmodN2Args a;
  a.bases = d_bases; a.out = d_out;
  a.consts = (const u32*)(dp + (lpl == LIMBS_PER_LANE_WIDE ? cb : 0));
  a.tape = (const u32*)(dp + 2 * cb);
  a.ntape = plan->ntape;
  a.slots = (u32*)d_ws;
  a.batch = batch; a.limbsn = plan->limbs_n; a.limbs2 = limbs2; a.nblk = p.geo.nblk; a.ksplit = bits - 1;
  hipStream_t s = (hipStream_t)stream;
  // segments: consecutive launches that each execute a stretch of the tape (mx_powmod_n2.hpp); positions
  // are counted in squarings, the accumulator travels through a scratch slot of the workspace
  int nseg = segments > 0 ? segments : n2_auto_segments(plan->n_sqr, p.nblocks);
  if (nseg > plan->n_sqr / 16) nseg = plan->n_sqr / 16;
  if (nseg < 1) nseg = 1;
  MxKernelTimer timer(s);                        // one timed interval per exponentiation (all its segments)
  for (int sg = 0; sg < nseg; ++sg) {
    a.first = sg == 0;
    a.last = sg == nseg - 1;
    a.pos_begin = (int)((int64_t)plan->n_sqr * sg / nseg);
    a.pos_end = a.last ? 0x7FFFFFFF : (int)((int64_t)plan->n_sqr * (sg + 1) / nseg);
    int rc = MX_ERR_SIZE;
    switch (p.geo.K) {
      case 1: rc = launch_n2_k<1>(a, p.nblocks, p.geo.L, s); break;
      case 2: rc = launch_n2_k<2>(a, p.nblocks, p.geo.L, s); break;
      case 4: rc = launch_n2_k<4>(a, p.nblocks, p.geo.L, s); break;
      case 8: rc = launch_n2_k<8>(a, p.nblocks, p.geo.L, s); break;
      case 16: rc = launch_n2_k<16>(a, p.nblocks, p.geo.L, s); break;
      case 32: rc = launch_n2_k<32>(a, p.nblocks, p.geo.L, s); break;
    }
    if (rc != MX_OK) return rc;
  }
  return MX_OK;
}

// ---- one-shot form: prepare into the head of the workspace, run with the rest
extern "C" int64_t mx_powmod_nsquare_workspace_bytes(int limbs_n, int exp_limbs, int64_t batch) {
  if (limbs_n <= 0 || exp_limbs <= 0 || batch <= 0) return MX_ERR_ARG;
  mx_nsquare_plan sizing{};
  sizing.n_bits = sizing_bits(limbs_n);
  sizing.window = sliding_window(32 * exp_limbs);
  int64_t run = mx_powmod_nsquare_run_workspace_bytes(&sizing, batch);
  if (run < 0) return run;
  return mx_nsquare_plan_bytes(limbs_n, exp_limbs) + run;
}

extern "C" int mx_powmod_nsquare(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_n, const uint32_t* h_exp,
                                 int limbs_n, int limbs2, int exp_limbs, int64_t batch, void* d_ws, int64_t ws_bytes,
                                 void* stream) {
  if (!d_bases || !d_out || !h_n || !h_exp || !d_ws) return MX_ERR_ARG;
  if (limbs_n <= 0 || limbs2 <= 0 || exp_limbs <= 0 || batch <= 0) return MX_ERR_ARG;
  const int64_t pb = mx_nsquare_plan_bytes(limbs_n, exp_limbs);
  if (pb > ws_bytes) {
    if (!(h_n[0] & 1u) || bit_length(h_n, limbs_n) < 2) return MX_ERR_MODULUS;
    return MX_ERR_WORKSPACE;
  }
  mx_nsquare_plan plan;
  MX_TRY(mx_powmod_nsquare_prepare(&plan, h_n, h_exp, limbs_n, exp_limbs, d_ws, pb, stream));
  return mx_powmod_nsquare_run(&plan, d_bases, d_out, limbs2, batch, override_limbs_per_lane(), 0,
                               (char*)d_ws + pb, ws_bytes - pb, stream);
}
