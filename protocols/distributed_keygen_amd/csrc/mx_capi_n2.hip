// extern "C" entry points of the N^2-modulus modexp (second translation unit of libmxpaillier.so:
// compiled in parallel with mx_capi.hip, the pair kernels are the most expensive to build).
#include "mx_upload.hpp"
#include "mx_powmod_n2.hpp"

// ---- modexp modulo N^2 through pairs modulo N --------------------------------------------------
namespace {
struct N2Plan {
  Geometry geo;
  int win = 1;
  int64_t nblocks = 0, nlanes = 0;
  int64_t off_consts = 0, off_ops = 0, off_table = 0, total = 0;
  int nslots = 0;
};

bool plan_n2(int n_bits, int limbs_n, int exp_bits, int64_t batch, N2Plan& p, int limbs_per_lane) {
  if (!choose_geometry(n_bits, p.geo, limbs_per_lane)) return false;
  p.win = sliding_window(exp_bits > 0 ? exp_bits : 1);
  int gpw = 64 / p.geo.K;
  p.nblocks = (batch + gpw - 1) / gpw;
  p.nlanes = p.nblocks * 64;
  int64_t o = 0;
  p.off_consts = o; o += align256((int64_t)8 * limbs_n * 4);
  p.off_ops = o;    o += align256((int64_t)MAX_SLIDING_OPS * 4);
  p.nslots = mx::N2_SLOT_TABLE + (1 << (p.win - 1));
  p.off_table = o;  o += align256((int64_t)p.nslots * 2 * p.geo.L * p.nlanes * 4);
  p.total = o;
  return true;
}

template <int K, int L>
int launch_n2_kl(const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s) {
  size_t lds = mx::powmod_n2_lds_bytes<K, L>();
  MxKernelTimer timer(s);
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

// Geometry of the pair kernel.  Measured on MI355X (tools/ab_geometry.sh, tools/ab_streams.sh,
// tools/sweep_keys.sh, profiles/): the wide geometry issues 18 % fewer instructions per element, runs 2
// wavefronts per SIMD and puts twice the elements into a wavefront.  At key_length 2048 it is faster
// for a lone 10 000-element launch (164 k vs 144 k modexps/s), with four or more launches in flight
// (262-272 k vs 241-255 k) and saturated (300 k vs 262 k); with exactly three 10 000-element launches
// in flight (1875 wavefronts for 2048 slots) its rate depends on how they interleave (228 k or 263 k;
// narrow 240-256 k), and for small launches the narrow geometry, which makes twice the wavefronts,
// fills the machine better (2 000 elements: 80 k vs 55 k).  So: wide from ~480 wavefronts per launch.
int n2_limbs_per_lane(int n_bits, int64_t batch) {
  if (g_limbs_per_lane == LIMBS_PER_LANE || g_limbs_per_lane == LIMBS_PER_LANE_WIDE) return g_limbs_per_lane;
  if (const char* e = getenv("MX_LIMBS_PER_LANE")) {
    int v = atoi(e);
    if (v == LIMBS_PER_LANE || v == LIMBS_PER_LANE_WIDE) return v;
  }
  Geometry narrow, wide;
  if (!choose_geometry(n_bits, narrow, LIMBS_PER_LANE) || !choose_geometry(n_bits, wide, LIMBS_PER_LANE_WIDE))
    return LIMBS_PER_LANE;
  if (narrow.K < 8 || wide.K > 16) return LIMBS_PER_LANE;
  const int64_t waves = (batch * wide.K + 63) / 64;
  return waves >= 480 ? LIMBS_PER_LANE_WIDE : LIMBS_PER_LANE;
}
}  // namespace

extern "C" int mx_nsquare_geometry(int n_bits, int64_t batch, int* k, int* l, int* w, int* blocks) {
  if (!k || !l || !w || !blocks || batch <= 0) return MX_ERR_ARG;
  Geometry g;
  if (!choose_geometry(n_bits, g, n2_limbs_per_lane(n_bits, batch))) return MX_ERR_SIZE;
  *k = g.K; *l = g.L; *w = g.W; *blocks = g.nblk;
  return MX_OK;
}

extern "C" int64_t mx_powmod_nsquare_workspace_bytes(int limbs_n, int exp_limbs, int64_t batch) {
  if (limbs_n <= 0 || exp_limbs <= 0 || batch <= 0) return MX_ERR_ARG;
  N2Plan p, q;
  if (!plan_n2(sizing_bits(limbs_n), limbs_n, 32 * exp_limbs, batch, p, LIMBS_PER_LANE)) return MX_ERR_SIZE;
  if (plan_n2(sizing_bits(limbs_n), limbs_n, 32 * exp_limbs, batch, q, LIMBS_PER_LANE_WIDE) && q.total > p.total)
    return q.total;
  return p.total;
}

extern "C" int mx_powmod_nsquare(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_n, const uint32_t* h_exp,
                                 int limbs_n, int limbs2, int exp_limbs, int64_t batch, void* d_ws, int64_t ws_bytes,
                                 void* stream) {
  if (!d_bases || !d_out || !h_n || !h_exp || !d_ws) return MX_ERR_ARG;
  if (limbs_n <= 0 || limbs2 <= 0 || exp_limbs <= 0 || batch <= 0) return MX_ERR_ARG;
  if (!(h_n[0] & 1u)) return MX_ERR_MODULUS;
  const int bits = bit_length(h_n, limbs_n);
  if (bits < 2) return MX_ERR_MODULUS;
  if (2 * bits - 1 > 32 * limbs2) return MX_ERR_ARG;          // rows too narrow for N^2
  const int ebits = bit_length(h_exp, exp_limbs);
  N2Plan p;
  if (!plan_n2(bits, limbs_n, 32 * exp_limbs, batch, p, n2_limbs_per_lane(bits, batch))) return MX_ERR_SIZE;
  if (p.total > ws_bytes) return MX_ERR_WORKSPACE;
  const int m = p.geo.W * p.geo.L * p.geo.nblk;                // R = 2^m
  const int k = bits - 1;                                      // x = x_lo + 2^k x_hi
  if (2 * p.geo.K * p.geo.L + 8 < limbs2 + 2) return MX_ERR_ARG;   // row wider than the staging area

  // ---- constants, each limbs_n words: N | ONE0 ONE1 | K1_0 K1_1 | K2_0 K2_1 | C'
  const int l2 = 2 * limbs_n;
  std::vector<u32> n2(l2), tmp(l2), qq(l2), rr(limbs_n);
  mul_words(n2.data(), h_n, limbs_n, h_n, limbs_n);
  std::vector<u32> c((size_t)8 * limbs_n, 0u);
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
  {                           // C' = N*ceil(R/N) - R + 1 = N - (R mod N) + 1   (R mod N != 0 as N is odd > 1)
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
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)d_ws;
  MX_TRY(upload_words(ws + p.off_consts, c.data(), c.size(), s));
  mx::PowmodN2Args a;
  a.bases = d_bases; a.out = d_out;
  a.consts = (const u32*)(ws + p.off_consts);
  a.slots = (u32*)(ws + p.off_table);
  a.batch = batch; a.limbsn = limbs_n; a.limbs2 = limbs2; a.nblk = p.geo.nblk; a.ksplit = k;
  // ---- the tape (mx_powmod_n2.hpp): conversion, table of odd powers, sliding window, times E
  std::vector<u32> tape;
  auto emit = [&](u32 op, int arg) { tape.push_back((op << 28) | (u32)arg); };
  if (ebits == 0) {
    emit(mx::N2_LOAD, mx::N2_SLOT_ONE);
  } else {
    int w = sliding_window(ebits);
    if (w > p.win) w = p.win;
    std::vector<u32> ops = sliding_schedule(h_exp, exp_limbs, w);
    // x = (x_lo, 0) * K1 + (x_hi, 0) * K2
    emit(mx::N2_LOAD, mx::N2_SLOT_LO); emit(mx::N2_MUL, mx::N2_SLOT_K1); emit(mx::N2_STORE, mx::N2_SLOT_TMP);
    emit(mx::N2_LOAD, mx::N2_SLOT_HI); emit(mx::N2_MUL, mx::N2_SLOT_K2); emit(mx::N2_ADD, mx::N2_SLOT_TMP);
    emit(mx::N2_STORE, mx::N2_SLOT_TABLE);
    const int nodd = 1 << (w - 1);
    if (nodd > 1) {
      emit(mx::N2_SQR, 1); emit(mx::N2_STORE, mx::N2_SLOT_SQ); emit(mx::N2_LOAD, mx::N2_SLOT_TABLE);
      for (int t = 1; t < nodd; ++t) { emit(mx::N2_MUL, mx::N2_SLOT_SQ); emit(mx::N2_STORE, mx::N2_SLOT_TABLE + t); }
    }
    emit(mx::N2_LOAD, mx::N2_SLOT_TABLE + (int)(ops[0] & 0xFFFFu) - 1);
    for (size_t t = 1; t < ops.size(); ++t) {
      int nsq = (int)(ops[t] >> 16), idx1 = (int)(ops[t] & 0xFFFFu);
      if (nsq) emit(mx::N2_SQR, nsq);
      if (idx1) emit(mx::N2_MUL, mx::N2_SLOT_TABLE + idx1 - 1);
    }
  }
  emit(mx::N2_MUL, mx::N2_SLOT_E);
  if ((int)tape.size() > MAX_SLIDING_OPS) return MX_ERR_SIZE;
  MX_TRY(upload_words(ws + p.off_ops, tape.data(), tape.size(), s));
  a.tape = (const u32*)(ws + p.off_ops);
  a.ntape = (int)tape.size();
  switch (p.geo.K) {
    case 1: return launch_n2_k<1>(a, p.nblocks, p.geo.L, s);
    case 2: return launch_n2_k<2>(a, p.nblocks, p.geo.L, s);
    case 4: return launch_n2_k<4>(a, p.nblocks, p.geo.L, s);
    case 8: return launch_n2_k<8>(a, p.nblocks, p.geo.L, s);
    case 16: return launch_n2_k<16>(a, p.nblocks, p.geo.L, s);
    case 32: return launch_n2_k<32>(a, p.nblocks, p.geo.L, s);
  }
  return MX_ERR_SIZE;
}
