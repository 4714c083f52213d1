// extern "C" entry points of libmxpaillier.so (declared in include/mxpaillier.h).
#include "mx_upload.hpp"
#include "mx_powmod.hpp"
#include "mx_sieve.hpp"
#include "mx_combine.hpp"
#include "mx_verdict.hpp"
#include "mx_jacobi.hpp"
#include "mx_mulmod.hpp"
#include "mx_select.hpp"
#include "mx_setup.hpp"
#include "mx_field.hpp"
#include "mx_modinv.hpp"
#include "mx_bimont.hpp"
#include <cstring>
#include <algorithm>

namespace mxh {
thread_local hipError_t g_last_hip = hipSuccess;
Knob g_knob_n2_segments;
Knob g_knob_n2_timeslice;
Knob g_knob_n2_friendly_1w;
Knob g_knob_generic_latency;
Knob g_knob_n2_split;
Knob g_knob_n2_bipair;
Knob g_knob_jacobi_max_batches;
Knob g_knob_bi_pivot;
Knob g_knob_lat_lanes;
}
MxProfile g_mx_profile;

namespace {


// ---- modexp ------------------------------------------------------------------------------
struct PowmodPlan {
  Geometry geo;
  int win = 1;
  int64_t nblocks = 0, nlanes = 0;
  int64_t off_mods = 0, off_rmodn = 0, off_exps = 0, off_ops = 0, off_table = 0, off_bi = 0, total = 0;
};

}  // namespace
// latency instances (3 limbs per lane) live in mx_capi_lat.hip
namespace mxl {
int launch_powmod_lat(int K, const mx::PowmodArgs& a, int64_t nblocks, hipStream_t s);
int launch_rmodn_lat(int K, const mx::RmodnArgs& a, hipStream_t s);
int launch_powmod_bi(int K, const mx::PowmodBiArgs& a, int64_t nblocks, hipStream_t s);
int launch_bisetup(int K, const mx::BiSetupArgs& a, hipStream_t s);
}
namespace {
inline bool generic_lpl_ok(int lpl) {
  return lpl == 0 || lpl == LIMBS_PER_LANE || lpl == LIMBS_PER_LANE_WIDE || lpl == LIMBS_PER_LANE_LAT || lpl == LIMBS_PER_LANE_BI;
}

// Geometry when the caller leaves the choice to the library: the shape with the shortest estimated duration of ONE launch
// on an idle GPU.  A wavefront of the 3- / 9- / 18-limb instances carries 64 / K elements through the whole exponentiation
// in 1 : 1.43 : 2.12 of the time (instructions on its dependent chain: 16 / 28 / 47 per limb step, measured ratios), alone
// on its SIMD; every further wavefront per SIMD adds 0.6 / 0.8 / 0.96 of that (a second wavefront fills the issue slots
// the first one leaves: more of them in the short-limbed instances) — fitted to tools/sweep_generic.py at key_length 1024
// and 2048 (profiles/r04_sweep_generic.txt; the choice matches the fastest measured shape at every point of the sweep
// within 3 %).  Consequences: a keygen round's few hundred to a thousand modexps (2-25 survivors) run the latency
// instances (6.0 instead of 8.7 ms at key_length 2048), 256-384 candidates the wide ones (one wavefront per SIMD: 12.8
// instead of 15.3 ms), the saturating launches whichever packs the SIMDs more evenly.
double generic_estimate(int mod_bits, int64_t batch, int lpl) {
  Geometry g;
  if (!choose_geometry(mod_bits, g, lpl)) return -1.0;
  if (lpl == LIMBS_PER_LANE_WIDE && g.K > 32) return -1.0;          // no <64, 18> instance
  const int64_t simds = (int64_t)4 * mx_device_cus();
  if (g.bi) {
    // One workgroup = two wavefronts per 64 / K elements.  Alone on its compute unit a product costs the longer half plus
    // the hand-over (tools/bi_pivot_sweep.py, tools/sweep_generic.py; profiles/r05_*): relative to the one-wavefront latency
    // instance 0.42 + 22 / steps — 0.94 at key_length 1024 (42 steps), 0.71 at 2048 (75 steps) — and 0.84 for groups of 64
    // lanes (key_length 3072 / 4096: the hand-over costs them more).  A second workgroup on a compute unit adds most of
    // that again, so the form only pays while the launch leaves compute units idle.
    const int steps = g.L * g.nblk + g.L;
    const int64_t wgs = ((batch * g.K + 63) / 64 + mx::BI_PAIRS - 1) / mx::BI_PAIRS;      // four wavefronts each: one per SIMD of a CU
    const int64_t cus = mx_device_cus();
    const int64_t per_cu = (wgs + cus - 1) / cus;
    const double t1 = g.K == 64 ? 0.84 : 0.42 + 22.0 / (double)steps;
    return t1 * (1.0 + 0.9 * (double)(per_cu - 1));
  }
  const int64_t waves = (batch * g.K + 63) / 64;
  const int64_t per_simd = (waves + simds - 1) / simds;
  const double t1 = lpl == LIMBS_PER_LANE_LAT ? 1.0 : lpl == LIMBS_PER_LANE ? 1.43 : 2.12;
  const double more = lpl == LIMBS_PER_LANE_LAT ? 0.6 : lpl == LIMBS_PER_LANE ? 0.8 : 0.96;
  return t1 * (1.0 + more * (double)(per_simd - 1));
}
int auto_limbs_per_lane(int mod_bits, int64_t batch, int64_t groups) {
  (void)groups;
  int best = LIMBS_PER_LANE;
  double best_t = -1.0;
  for (int lpl : {LIMBS_PER_LANE, LIMBS_PER_LANE_WIDE, LIMBS_PER_LANE_LAT, LIMBS_PER_LANE_BI}) {
    if (lpl == LIMBS_PER_LANE_LAT && g_knob_generic_latency == 1) continue;
    if (lpl == LIMBS_PER_LANE_BI && g_knob_generic_latency != 0) continue;      // 1: no latency instances at all, 2: no bipartite form
    const double t = generic_estimate(mod_bits, batch, lpl);
    if (t > 0 && (best_t < 0 || t < best_t * 0.999)) { best_t = t; best = lpl; }
  }
  return best;
}

bool plan_powmod(int mod_bits, int limbs, int exp_limbs, int64_t batch, int64_t groups, PowmodPlan& p,
                 int limbs_per_lane) {
  if (limbs_per_lane == 0) limbs_per_lane = auto_limbs_per_lane(mod_bits, batch, groups);
  if (!choose_geometry(mod_bits, p.geo, limbs_per_lane)) return false;
  p.win = fixed_window(32 * exp_limbs);
  int gpw = 64 / p.geo.K;
  p.nblocks = (batch + gpw - 1) / gpw;
  p.nlanes = p.nblocks * 64;
  if (p.geo.bi) {                               // workgroups of BI_PAIRS wavefront pairs; the table has a column per lane of every pair
    p.nblocks = (p.nblocks + mx::BI_PAIRS - 1) / mx::BI_PAIRS;
    p.nlanes = p.nblocks * mx::BI_PAIRS * 64;
  }
  int64_t o = 0;
  p.off_mods = o;  o += align256((int64_t)groups * limbs * 4);
  p.off_rmodn = o; o += align256((int64_t)groups * limbs * 4);
  p.off_exps = o;  o += align256((int64_t)groups * exp_limbs * 4);
  p.off_ops = o;   o += align256((int64_t)MAX_SLIDING_OPS * 4);
  p.off_table = o; o += align256(((int64_t)1 << p.win) * p.geo.L * p.nlanes * 4);
  p.off_bi = o;
  if (p.geo.bi) o += align256((int64_t)groups * mx::BI_ROWS * p.geo.L * p.geo.K * 4);
  p.total = o;
  return true;
}

template <int K, int L>
int launch_powmod_kl(const mx::PowmodArgs& a, int64_t nblocks, hipStream_t s) {
  using M_t = mx::Mont<K, L, LIMB_BITS, true>;
  size_t lds = (size_t)(64 / K) * M_t::LDS_WORDS * 4;
  MxKernelTimer timer(s);
  if (a.nops > 0)
    hipLaunchKernelGGL((mx::powmod_kernel<K, L, LIMB_BITS, true>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  else
    hipLaunchKernelGGL((mx::powmod_kernel<K, L, LIMB_BITS, false>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

template <int K>
int launch_powmod_k(const mx::PowmodArgs& a, int64_t nblocks, int limbs_per_lane, hipStream_t s) {
  if (limbs_per_lane == LIMBS_PER_LANE_LAT) return mxl::launch_powmod_lat(K, a, nblocks, s);
  if (limbs_per_lane == LIMBS_PER_LANE_WIDE) {
    // the largest supported modulus (MAX_MOD_BITS) needs 32 wide lanes: no <64, 18> instance
    if constexpr (K <= 32) return launch_powmod_kl<K, LIMBS_PER_LANE_WIDE>(a, nblocks, s);
    return MX_ERR_SIZE;
  }
  return launch_powmod_kl<K, LIMBS_PER_LANE>(a, nblocks, s);
}

// R mod N of every group, on the device (mx_setup.hpp); always the narrow-L instance of the
// geometry's K would give a different R, so the instance follows the geometry of the consumer
template <int K, int L>
int launch_rmodn_kl(const mx::RmodnArgs& a, hipStream_t s) {
  using M_t = mx::Mont<K, L, LIMB_BITS, true>;
  int gpw = 64 / K;
  int64_t nblocks = (a.groups + gpw - 1) / gpw;
  hipLaunchKernelGGL((mx::rmodn_kernel<K, L, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64),
                     (size_t)gpw * M_t::LDS_WORDS * 4, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

int launch_rmodn(const Geometry& g, const u32* d_mods, u32* d_rmodn, int limbs, int64_t groups, hipStream_t s) {
  mx::RmodnArgs a;
  a.mods = d_mods; a.rmodn = d_rmodn; a.groups = groups; a.limbs = limbs; a.nblk = g.nblk;
  if (g.L == LIMBS_PER_LANE_LAT) return mxl::launch_rmodn_lat(g.K, a, s);
  const bool wide = g.L == LIMBS_PER_LANE_WIDE;
  switch (g.K) {
#define MX_CASE(KK) case KK: return wide ? launch_rmodn_kl<KK, LIMBS_PER_LANE_WIDE>(a, s) : launch_rmodn_kl<KK, LIMBS_PER_LANE>(a, s);
    MX_CASE(1) MX_CASE(2) MX_CASE(4) MX_CASE(8) MX_CASE(16) MX_CASE(32)
#undef MX_CASE
    case 64: return wide ? MX_ERR_SIZE : launch_rmodn_kl<64, LIMBS_PER_LANE>(a, s);
  }
  return MX_ERR_SIZE;
}

// Common tail of the modexp entry points: moduli and exponents are device-resident
// (d_mods [groups][limbs], d_exps [groups][exp_limbs]); h_exp0 is the host copy of the exponent when
// the launch shares one (sliding-window schedule), else null.
int powmod_launch(const uint32_t* d_bases, uint32_t* d_out, const u32* d_mods, const u32* d_exps, const u32* h_exp0,
                  int limbs, int exp_limbs, int max_ebits, int64_t groups, int64_t group_size, const PowmodPlan& p,
                  char* ws, hipStream_t s) {
  int ndigits = (max_ebits + p.win - 1) / p.win;
  if (ndigits < 1) ndigits = 1;
  if (p.geo.bi) {
    // the bipartite latency form (mx_bimont.hpp): its own setup kernel (fold constants and the conversion factor of every
    // modulus), fixed windows only, two wavefronts per workgroup
    mx::BiSetupArgs sa;
    sa.mods = d_mods; sa.consts = (u32*)(ws + p.off_bi); sa.groups = groups; sa.limbs = limbs;
    sa.nblk = p.geo.nblk; sa.pd = p.geo.L * p.geo.nblk; sa.h_lo = p.geo.h_lo;
    MX_TRY(mxl::launch_bisetup(p.geo.K, sa, s));
    mx::PowmodBiArgs b;
    b.bases = d_bases; b.out = d_out; b.mods = d_mods; b.exps = d_exps; b.consts = (const u32*)(ws + p.off_bi);
    b.table = (u32*)(ws + p.off_table);
    b.batch = groups * group_size; b.group_size = group_size;
    b.limbs = limbs; b.elimbs = exp_limbs; b.ndigits = ndigits; b.win = p.win;
    b.nblk = p.geo.nblk; b.pd = sa.pd; b.h_lo = p.geo.h_lo;
    return mxl::launch_powmod_bi(p.geo.K, b, p.nblocks, s);
  }
  MX_TRY(launch_rmodn(p.geo, d_mods, (u32*)(ws + p.off_rmodn), limbs, groups, s));
  mx::PowmodArgs a;
  a.bases = d_bases; a.out = d_out;
  a.mods = d_mods;
  a.rmodn = (const u32*)(ws + p.off_rmodn);
  a.exps = d_exps;
  a.table = (u32*)(ws + p.off_table);
  a.batch = groups * group_size; a.group_size = group_size;
  a.limbs = limbs; a.elimbs = exp_limbs; a.ndigits = ndigits; a.win = p.win; a.nblk = p.geo.nblk;
  a.ops = nullptr; a.nops = 0;
  if (groups == 1 && max_ebits > 0 && h_exp0) {
    // one exponent for the whole launch: sliding window (odd powers only)
    int w = sliding_window(max_ebits);
    if (w > p.win) w = p.win;                       // the table region was sized for 2^win entries
    std::vector<u32> ops = pack_sliding_ops(sliding_schedule(h_exp0, exp_limbs, w));
    if ((int)ops.size() <= MAX_SLIDING_OPS) {
      MX_TRY(upload_words(ws + p.off_ops, ops.data(), ops.size(), s));
      a.ops = (const u32*)(ws + p.off_ops);
      a.nops = (int)ops.size();
      a.win = w;
    }
  }
  switch (p.geo.K) {
    case 1: return launch_powmod_k<1>(a, p.nblocks, p.geo.L, s);
    case 2: return launch_powmod_k<2>(a, p.nblocks, p.geo.L, s);
    case 4: return launch_powmod_k<4>(a, p.nblocks, p.geo.L, s);
    case 8: return launch_powmod_k<8>(a, p.nblocks, p.geo.L, s);
    case 16: return launch_powmod_k<16>(a, p.nblocks, p.geo.L, s);
    case 32: return launch_powmod_k<32>(a, p.nblocks, p.geo.L, s);
    case 64: return launch_powmod_k<64>(a, p.nblocks, p.geo.L, s);
  }
  return MX_ERR_SIZE;
}

// host operands: validate, upload, launch
int powmod_impl(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_mods, const uint32_t* h_exps,
                int limbs, int exp_limbs, int64_t groups, int64_t group_size, int limbs_per_lane, void* d_ws,
                int64_t ws_bytes, void* stream) {
  if (!d_bases || !d_out || !h_mods || !h_exps || !d_ws) return MX_ERR_ARG;
  if (limbs <= 0 || exp_limbs <= 0 || groups <= 0 || group_size <= 0) return MX_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int64_t batch = groups * group_size;
  int max_bits = 0;
  for (int64_t g = 0; g < groups; ++g) {
    const u32* n = h_mods + g * limbs;
    if (!(n[0] & 1u)) return MX_ERR_MODULUS;
    int b = bit_length(n, limbs);
    if (b < 2) return MX_ERR_MODULUS;
    if (b > max_bits) max_bits = b;
  }
  PowmodPlan p;
  if (!plan_powmod(max_bits, limbs, exp_limbs, batch, groups, p, limbs_per_lane)) return MX_ERR_SIZE;
  // the sizing call only knows `limbs`; it assumes the largest modulus that fits them
  if (p.total > ws_bytes) return MX_ERR_WORKSPACE;
  int max_ebits = 0;
  for (int64_t g = 0; g < groups; ++g) {
    int b = bit_length(h_exps + g * exp_limbs, exp_limbs);
    if (b > max_ebits) max_ebits = b;
  }
  char* ws = (char*)d_ws;
  MX_TRY(upload_words(ws + p.off_mods, h_mods, (size_t)groups * limbs, s));
  MX_TRY(upload_words(ws + p.off_exps, h_exps, (size_t)groups * exp_limbs, s));
  return powmod_launch(d_bases, d_out, (const u32*)(ws + p.off_mods), (const u32*)(ws + p.off_exps),
                       groups == 1 ? h_exps : nullptr, limbs, exp_limbs, max_ebits, groups, group_size, p, ws, s);
}

// ---- lane self-test ------------------------------------------------------------------------
template <int K>
__device__ int lanes_check(unsigned v) {
  using D = mx::Lanes<K, true>;
  using R = mx::Lanes<K, false>;
  unsigned kn = D::keep_next_mask(), kp = D::keep_prev_mask();
  int bad = 0;
  bad += D::bcast0(v) != R::bcast0(v);
  bad += D::from_next(v, kn) != R::from_next(v, kn);
  bad += D::from_prev(v, kp) != R::from_prev(v, kp);
  // semantic anchors independent of either implementation
  unsigned lane = threadIdx.x & 63, pos = lane & (K - 1), base = lane - pos;
  unsigned f = lane * 2654435761u + 12345u;   // f(lane)
  auto F = [](unsigned l) { return l * 2654435761u + 12345u; };
  bad += D::bcast0(f) != F(base);
  bad += D::from_next(f, kn) != (pos == K - 1 ? 0u : F(lane + 1));
  bad += D::from_prev(f, kp) != (pos == 0 ? 0u : F(lane - 1));
  return bad;
}

// Occupies one wavefront slot for `ticks` periods of the 100 MHz real-time counter and does nothing else:
// the probe with which a caller finds out whether two of its streams run concurrently (mx_spin).
__global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// One wavefront idles for `ticks` periods of the 100 MHz real-time counter and reports how far the shader clock
// counter (s_memtime) advanced meanwhile: the clock the SIMDs are actually running at while whatever else is in
// flight on the device keeps them busy (mx_clock_probe).
__global__ void clock_probe_kernel(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

__global__ void lanes_selftest_kernel(int* out) {
  unsigned v = threadIdx.x * 40503u + 977u;
  int bad = 0;
  bad += lanes_check<1>(v);
  bad += lanes_check<2>(v);
  bad += lanes_check<4>(v);
  bad += lanes_check<8>(v);
  bad += lanes_check<16>(v);
  bad += lanes_check<32>(v);
  bad += lanes_check<64>(v);
  atomicAdd(out, bad);
}

}  // namespace

extern "C" {

int mx_version(void) { return 404; }

const char* mx_error_string(int code) {
  switch (code) {
    case MX_OK: return "ok";
    case MX_ERR_ARG: return "invalid argument";
    case MX_ERR_SIZE: return "modulus too large for the engine";
    case MX_ERR_MODULUS: return "modulus must be odd and >= 3";
    case MX_ERR_WORKSPACE: return "workspace too small";
    case MX_ERR_HIP: return "HIP runtime error";
  }
  return "unknown error";
}

const char* mx_last_hip_error(void) { return hipGetErrorString(g_last_hip); }

int mx_debug_knob(int knob, int value) {
  if (value < 0) return MX_ERR_ARG;
  switch (knob) {
    case MX_KNOB_N2_SEGMENTS: if (value > 64) return MX_ERR_ARG; g_knob_n2_segments = value; return MX_OK;
    case MX_KNOB_N2_TIMESLICE: if (value > 2 && (value < 17 || value > 19)) return MX_ERR_ARG; g_knob_n2_timeslice = value; return MX_OK;
    case MX_KNOB_JACOBI_MAX_BATCHES: g_knob_jacobi_max_batches = value; return MX_OK;
    case MX_KNOB_N2_FRIENDLY_1W: if (value > 1) return MX_ERR_ARG; g_knob_n2_friendly_1w = value; return MX_OK;
    case MX_KNOB_GENERIC_LATENCY: if (value > 2) return MX_ERR_ARG; g_knob_generic_latency = value; return MX_OK;
    case MX_KNOB_N2_SPLIT: if (value > 2) return MX_ERR_ARG; g_knob_n2_split = value; return MX_OK;
    case MX_KNOB_N2_BIPAIR: if (value > 1) return MX_ERR_ARG; g_knob_n2_bipair = value; return MX_OK;
    case MX_KNOB_BI_PIVOT: if (value > 192) return MX_ERR_ARG; g_knob_bi_pivot = value; return MX_OK;
    case MX_KNOB_LAT_LANES: if (value > 64) return MX_ERR_ARG; g_knob_lat_lanes = value; return MX_OK;
  }
  return MX_ERR_ARG;
}

int mx_profile(int enable) {
  std::lock_guard<std::mutex> lock(g_mx_profile.mu);
  g_mx_profile.on = enable != 0;
  return MX_OK;
}

int mx_profile_collect(double* total_ms, int* launches) {
  if (!total_ms || !launches) return MX_ERR_ARG;
  std::lock_guard<std::mutex> lock(g_mx_profile.mu);
  double sum = 0;
  int n = 0, rc = MX_OK;
  for (auto& ev : g_mx_profile.events) {
    float ms = 0;
    if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      sum += ms;
      ++n;
    } else {
      rc = MX_ERR_HIP;
    }
    hipEventDestroy(ev.first);
    hipEventDestroy(ev.second);
  }
  g_mx_profile.events.clear();
  *total_ms = sum;
  *launches = n;
  return rc;
}

int mx_geometry(int mod_bits, int* k, int* l, int* w, int* blocks) {
  Geometry g;
  if (!choose_geometry(mod_bits, g, LIMBS_PER_LANE)) return MX_ERR_SIZE;
  if (k) *k = g.K;
  if (l) *l = g.L;
  if (w) *w = g.W;
  if (blocks) *blocks = g.nblk;
  return MX_OK;
}

int64_t mx_powmod_workspace_bytes(int limbs, int exp_limbs, int64_t batch, int64_t groups) {
  if (limbs <= 0 || exp_limbs <= 0 || batch <= 0 || groups <= 0) return MX_ERR_ARG;
  PowmodPlan p, q;
  if (!plan_powmod(sizing_bits(limbs), limbs, exp_limbs, batch, groups, p, LIMBS_PER_LANE)) return MX_ERR_SIZE;
  int64_t most = p.total;
  if (plan_powmod(sizing_bits(limbs), limbs, exp_limbs, batch, groups, q, LIMBS_PER_LANE_WIDE) && q.total > most) most = q.total;
  // (the latency geometry only exists up to 64 x 3 limbs: rows that could hold a wider modulus are sized with the
  // widest one it takes)
  const int lat_bits = std::min(sizing_bits(limbs), LIMB_BITS * LIMBS_PER_LANE_LAT * 64 - 4 - LIMB_BITS - 2);
  if (plan_powmod(lat_bits, limbs, exp_limbs, batch, groups, q, LIMBS_PER_LANE_LAT) && q.total > most) most = q.total;
  const int bi_bits = std::min(sizing_bits(limbs), LIMB_BITS * LIMBS_PER_LANE_LAT * 62 - 35);
  if (plan_powmod(bi_bits, limbs, exp_limbs, batch, groups, q, LIMBS_PER_LANE_BI) && q.total > most) most = q.total;
  return most;
}

int mx_powmod_shared(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_mod, const uint32_t* h_exp,
                     int limbs, int exp_limbs, int64_t batch, void* d_workspace, int64_t workspace_bytes,
                     void* stream) {
  return powmod_impl(d_bases, d_out, h_mod, h_exp, limbs, exp_limbs, 1, batch, 0,
                     d_workspace, workspace_bytes, stream);
}

int mx_powmod_shared_lpl(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_mod, const uint32_t* h_exp,
                         int limbs, int exp_limbs, int64_t batch, int limbs_per_lane, void* d_workspace,
                         int64_t workspace_bytes, void* stream) {
  if (!generic_lpl_ok(limbs_per_lane)) return MX_ERR_ARG;
  return powmod_impl(d_bases, d_out, h_mod, h_exp, limbs, exp_limbs, 1, batch, limbs_per_lane, d_workspace,
                     workspace_bytes, stream);
}

int mx_powmod_multi(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* h_mods, const uint32_t* h_exps,
                    int limbs, int exp_limbs, int64_t groups, int64_t group_size, void* d_workspace,
                    int64_t workspace_bytes, void* stream) {
  return powmod_impl(d_bases, d_out, h_mods, h_exps, limbs, exp_limbs, groups, group_size,
                     0, d_workspace, workspace_bytes, stream);
}

int mx_powmod_multi_dev(const uint32_t* d_bases, uint32_t* d_out, const uint32_t* d_mods, const uint32_t* d_exps,
                        int limbs, int exp_limbs, int mod_bits, int exp_bits, int64_t groups, int64_t group_size,
                        int limbs_per_lane, void* d_workspace, int64_t workspace_bytes, void* stream) {
  if (!d_bases || !d_out || !d_mods || !d_exps || !d_workspace) return MX_ERR_ARG;
  if (limbs <= 0 || exp_limbs <= 0 || groups <= 0 || group_size <= 0) return MX_ERR_ARG;
  if (mod_bits < 2 || mod_bits > 32 * limbs || exp_bits < 0 || exp_bits > 32 * exp_limbs) return MX_ERR_ARG;
  if (!generic_lpl_ok(limbs_per_lane)) return MX_ERR_ARG;
  PowmodPlan p;
  if (!plan_powmod(mod_bits, limbs, exp_limbs, groups * group_size, groups, p, limbs_per_lane)) return MX_ERR_SIZE;
  if (p.total > workspace_bytes) return MX_ERR_WORKSPACE;
  return powmod_launch(d_bases, d_out, d_mods, d_exps, nullptr, limbs, exp_limbs, exp_bits, groups, group_size, p,
                       (char*)d_workspace, (hipStream_t)stream);
}

int mx_powmod_geometry_for(int mod_bits, int64_t batch, int64_t groups, int limbs_per_lane, int* k, int* l, int* w,
                           int* blocks) {
  if (!k || !l || !w || !blocks || batch <= 0 || groups <= 0) return MX_ERR_ARG;
  if (!generic_lpl_ok(limbs_per_lane)) return MX_ERR_ARG;
  Geometry g;
  if (!choose_geometry(mod_bits, g, limbs_per_lane ? limbs_per_lane : auto_limbs_per_lane(mod_bits, batch, groups)))
    return MX_ERR_SIZE;
  *k = g.K; *l = g.L; *w = g.W; *blocks = g.nblk;
  return MX_OK;
}

int mx_powmod_launch_form(int mod_bits, int64_t batch, int64_t groups, int limbs_per_lane, int* wavefronts_per_group,
                          int* pivot) {
  if (!wavefronts_per_group || batch <= 0 || groups <= 0) return MX_ERR_ARG;
  if (!generic_lpl_ok(limbs_per_lane)) return MX_ERR_ARG;
  Geometry g;
  if (!choose_geometry(mod_bits, g, limbs_per_lane ? limbs_per_lane : auto_limbs_per_lane(mod_bits, batch, groups)))
    return MX_ERR_SIZE;
  *wavefronts_per_group = g.bi ? 2 : 1;
  if (pivot) *pivot = g.h_lo;
  return MX_OK;
}

int mx_spin(int64_t microseconds, void* stream) {
  if (microseconds < 0 || microseconds > 1000000) return MX_ERR_ARG;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

int mx_stream_create_cu_slice(int slice, int n_slices, int reserved, void** stream_out) {
  // CU mask bit i is CU (i / n_xcd) of XCD (i % n_xcd) (tools/ubench/cu_mask_probe.hip: 32 consecutive bits = 4 CUs
  // in each of the 8 XCDs; a mask that leaves an XCD without any CU is ignored by the runtime).  A slice is a
  // contiguous range of bits: the same CUs of every XCD.
  if (!stream_out || n_slices < 1 || slice < 0 || slice >= n_slices) return MX_ERR_ARG;
  int dev = 0, cus = 0;
  MX_HIP(hipGetDevice(&dev));
  MX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (cus < 32 * n_slices || n_slices > 8) return MX_ERR_ARG;      // at least 4 CUs per XCD and slice
  const int per = cus / n_slices, lo = slice * per, hi = slice == n_slices - 1 ? cus : lo + per;
  std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
  for (int i = lo; i < hi; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
  hipStream_t s = nullptr;
  (void)reserved;
  MX_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  *stream_out = (void*)s;
  return MX_OK;
}

int mx_stream_destroy(void* stream) {
  if (!stream) return MX_ERR_ARG;
  MX_HIP(hipStreamDestroy((hipStream_t)stream));
  return MX_OK;
}

int mx_clock_probe(int64_t microseconds, uint64_t* d_ticks, void* stream) {
  if (microseconds <= 0 || microseconds > 1000000 || !d_ticks) return MX_ERR_ARG;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull,
                     (unsigned long long*)d_ticks);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

int mx_selftest_lanes(void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int* d = nullptr;
  MX_HIP(hipMalloc(&d, sizeof(int)));
  int h = 0;
  hipError_t e = hipMemsetAsync(d, 0, sizeof(int), s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(lanes_selftest_kernel, dim3(1), dim3(64), 0, s, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(d);
  MX_HIP(e);
  return h;
}

}  // extern "C"

// ---- sieve ---------------------------------------------------------------------------------
namespace {
struct SievePlan { int np_pad; int64_t off_primes, off_pw, off_inv, off_lim, total; };
SievePlan plan_sieve(int limbs, int np) {
  SievePlan p;
  p.np_pad = (np + 63) / 64 * 64;
  int64_t o = 0;
  p.off_primes = o; o += align256((int64_t)np * 4);
  p.off_pw = o;     o += align256((int64_t)limbs * p.np_pad * 4);
  p.off_inv = o;    o += align256((int64_t)p.np_pad * 8);
  p.off_lim = o;    o += align256((int64_t)p.np_pad * 8);
  p.total = o;
  return p;
}
}  // namespace

extern "C" int64_t mx_sieve_workspace_bytes(int limbs, int n_primes) {
  if (limbs <= 0 || n_primes <= 0) return MX_ERR_ARG;
  return plan_sieve(limbs, n_primes).total;
}

extern "C" int mx_sieve(const uint32_t* d_cands, uint8_t* d_out, const uint32_t* h_primes, int n_primes, int limbs,
                        int64_t batch, void* d_ws, int64_t ws_bytes, void* stream) {
  if (!d_cands || !d_out || !h_primes || !d_ws || limbs <= 0 || n_primes <= 0 || batch <= 0) return MX_ERR_ARG;
  if (limbs > 1024) return MX_ERR_SIZE;
  uint32_t top = 3;
  for (int k = 0; k < n_primes; ++k) {
    if (!(h_primes[k] & 1u) || h_primes[k] < 3 || h_primes[k] >= (1u << 31)) return MX_ERR_ARG;
    if (h_primes[k] > top) top = h_primes[k];
  }
  hipStream_t s = (hipStream_t)stream;
  SievePlan p = plan_sieve(limbs, n_primes);
  if (p.total > ws_bytes) return MX_ERR_WORKSPACE;
  char* ws = (char*)d_ws;
  MX_TRY(upload_words(ws + p.off_primes, h_primes, (size_t)n_primes, s));
  mx::SieveArgs a;
  a.cands = d_cands; a.out = d_out;
  a.primes = (const u32*)(ws + p.off_primes);
  a.pw = (u32*)(ws + p.off_pw);
  a.inv = (u64*)(ws + p.off_inv);
  a.lim = (u64*)(ws + p.off_lim);
  a.batch = batch; a.limbs = limbs; a.np = n_primes; a.np_pad = p.np_pad;
  // limbs a 64-bit column takes between two folds (mx_sieve.hpp): 2^32 ((chunk + 1) top + 1) < 2^64
  const int64_t chunk = (int64_t)(0xFFFFFFFEull / top) - 1;          // >= 1 for top < 2^31
  a.chunk = (int)std::min<int64_t>(chunk, limbs);
  hipLaunchKernelGGL(mx::sieve_setup_kernel, dim3((unsigned)(p.np_pad / 64)), dim3(64), 0, s, a);
  MX_HIP(hipGetLastError());
  int64_t nblocks = (batch + mx::SIEVE_C - 1) / mx::SIEVE_C;
  size_t lds = (size_t)limbs * mx::SIEVE_C * 4;
  hipLaunchKernelGGL(mx::sieve_kernel, dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

// ---- share recombination -------------------------------------------------------------------
namespace {
template <int K>
int launch_combine_k(const mx::CombineArgs& a, hipStream_t s) {
  using M_t = mx::Mont<K, LIMBS_PER_LANE, LIMB_BITS, true>;
  int gpw = 64 / K;
  int64_t nblocks = (a.batch + gpw - 1) / gpw;
  size_t lds = (size_t)gpw * M_t::LDS_WORDS * 4;
  hipLaunchKernelGGL((mx::combine_kernel<K, LIMBS_PER_LANE, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}
}  // namespace

inline int64_t combine_plan_words(int limbs2) { return (int64_t)5 * limbs2; }

extern "C" int64_t mx_combine_plan_bytes(int limbs, int limbs2) {
  if (limbs <= 0 || limbs2 < limbs) return MX_ERR_ARG;
  return align256(combine_plan_words(limbs2) * 4);
}

// constants of a key, each limbs2 words: N | N^2 | R1 mod N | R2 mod N^2 | theta_inv
extern "C" int mx_combine_prepare(mx_combine_plan* plan, const uint32_t* h_n, const uint32_t* h_theta_inv, int limbs,
                                  int limbs2, void* d_plan, int64_t plan_bytes, void* stream) {
  if (!plan || !h_n || !h_theta_inv || !d_plan) return MX_ERR_ARG;
  if (limbs <= 0 || limbs2 < limbs) return MX_ERR_ARG;
  if (!(h_n[0] & 1u)) return MX_ERR_MODULUS;
  int bits1 = bit_length(h_n, limbs);
  if (bits1 < 2) return MX_ERR_MODULUS;
  std::vector<u32> n2((size_t)2 * limbs);
  mul_words(n2.data(), h_n, limbs, h_n, limbs);
  int bits2 = bit_length(n2.data(), 2 * limbs);
  if ((bits2 + 31) / 32 > limbs2) return MX_ERR_ARG;   // rows too narrow for N^2
  Geometry g2;
  if (!choose_geometry(bits2, g2)) return MX_ERR_SIZE;
  Geometry g1 = g2;
  g1.nblk = (bits1 + 4 + g1.W * g1.L - 1) / (g1.W * g1.L);
  if (mx_combine_plan_bytes(limbs, limbs2) > plan_bytes) return MX_ERR_WORKSPACE;
  std::vector<u32> c((size_t)5 * limbs2, 0u);
  std::memcpy(&c[0], h_n, (size_t)limbs * 4);
  std::memcpy(&c[limbs2], n2.data(), (size_t)std::min(limbs2, 2 * limbs) * 4);
  two_pow_mod(&c[(size_t)2 * limbs2], h_n, limbs, g1.W * g1.L * g1.nblk);
  {
    std::vector<u32> n2p(limbs2, 0u);
    std::memcpy(n2p.data(), n2.data(), (size_t)std::min(limbs2, 2 * limbs) * 4);
    two_pow_mod(&c[(size_t)3 * limbs2], n2p.data(), limbs2, g2.W * g2.L * g2.nblk);
  }
  std::memcpy(&c[(size_t)4 * limbs2], h_theta_inv, (size_t)limbs * 4);
  MX_TRY(upload_words(d_plan, c.data(), c.size(), (hipStream_t)stream));
  plan->d_plan = d_plan;
  plan->plan_bytes = plan_bytes;
  plan->limbs = limbs;
  plan->limbs2 = limbs2;
  plan->n_bits = bits1;
  plan->n2_bits = bits2;
  return MX_OK;
}

extern "C" int mx_combine_run(const mx_combine_plan* plan, const uint32_t* d_partials, uint32_t* d_out, int out_stride,
                              uint8_t* d_status, int n_partials, int64_t batch, void* stream) {
  if (!plan || !plan->d_plan || !d_partials || !d_out) return MX_ERR_ARG;
  if (n_partials <= 0 || batch <= 0 || out_stride < plan->limbs) return MX_ERR_ARG;
  if (!d_status && out_stride == plan->limbs) return MX_ERR_ARG;      // the status must go somewhere
  Geometry g2;
  if (!choose_geometry(plan->n2_bits, g2)) return MX_ERR_SIZE;
  Geometry g1 = g2;
  g1.nblk = (plan->n_bits + 4 + g1.W * g1.L - 1) / (g1.W * g1.L);
  const int limbs2 = plan->limbs2;
  const u32* w = (const u32*)plan->d_plan;
  mx::CombineArgs a;
  a.partials = d_partials; a.out = d_out; a.status = d_status;
  a.n = w; a.n2 = w + limbs2; a.rmodn1 = w + 2 * (size_t)limbs2; a.rmodn2 = w + 3 * (size_t)limbs2;
  a.theta_inv = w + 4 * (size_t)limbs2;
  a.batch = batch; a.limbs = plan->limbs; a.limbs2 = limbs2; a.np = n_partials; a.out_stride = out_stride;
  a.nblk1 = g1.nblk; a.nblk2 = g2.nblk;
  hipStream_t s = (hipStream_t)stream;
  switch (g2.K) {
    case 1: return launch_combine_k<1>(a, s);
    case 2: return launch_combine_k<2>(a, s);
    case 4: return launch_combine_k<4>(a, s);
    case 8: return launch_combine_k<8>(a, s);
    case 16: return launch_combine_k<16>(a, s);
    case 32: return launch_combine_k<32>(a, s);
    case 64: return launch_combine_k<64>(a, s);
  }
  return MX_ERR_SIZE;
}

// ---- one-shot form: prepare into the workspace, run
extern "C" int64_t mx_combine_workspace_bytes(int limbs, int limbs2, int n_partials, int64_t batch) {
  if (limbs <= 0 || limbs2 < limbs || n_partials <= 0 || batch <= 0) return MX_ERR_ARG;
  return mx_combine_plan_bytes(limbs, limbs2);
}

extern "C" int mx_combine(const uint32_t* d_partials, uint32_t* d_out, uint8_t* d_status, const uint32_t* h_n,
                          const uint32_t* h_theta_inv, int limbs, int limbs2, int n_partials, int64_t batch,
                          void* d_ws, int64_t ws_bytes, void* stream) {
  if (!d_partials || !d_out || !d_status || !h_n || !h_theta_inv || !d_ws) return MX_ERR_ARG;
  if (limbs <= 0 || limbs2 < limbs || n_partials <= 0 || batch <= 0) return MX_ERR_ARG;
  mx_combine_plan plan;
  MX_TRY(mx_combine_prepare(&plan, h_n, h_theta_inv, limbs, limbs2, d_ws, ws_bytes, stream));
  return mx_combine_run(&plan, d_partials, d_out, limbs, d_status, n_partials, batch, stream);
}

// ---- biprimality verdict -------------------------------------------------------------------
namespace {
template <int K>
int launch_verdict_k(const mx::VerdictArgs& a, hipStream_t s) {
  using M_t = mx::Mont<K, LIMBS_PER_LANE, LIMB_BITS, true>;
  int gpw = 64 / K;
  int64_t total = a.groups * a.n_slots;
  int64_t nblocks = (total + gpw - 1) / gpw;
  size_t lds = (size_t)gpw * M_t::LDS_WORDS * 4;
  hipLaunchKernelGGL((mx::verdict_kernel<K, LIMBS_PER_LANE, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}
}  // namespace

extern "C" int64_t mx_verdict_workspace_bytes(int limbs, int n_parties, int64_t groups, int64_t n_slots) {
  if (limbs <= 0 || n_parties <= 0 || groups <= 0 || n_slots <= 0) return MX_ERR_ARG;
  return 2 * align256((int64_t)groups * limbs * 4);
}

extern "C" int mx_biprime_verdict_dev(const uint32_t* d_v, uint8_t* d_pass, const uint32_t* d_mods, int limbs,
                                      int mod_bits, int n_parties, int64_t groups, int64_t n_slots, void* d_ws,
                                      int64_t ws_bytes, void* stream) {
  if (!d_v || !d_pass || !d_mods || !d_ws || limbs <= 0 || n_parties <= 0 || groups <= 0 || n_slots <= 0)
    return MX_ERR_ARG;
  if (mod_bits < 2 || mod_bits > 32 * limbs) return MX_ERR_ARG;
  Geometry geo;
  if (!choose_geometry(mod_bits, geo)) return MX_ERR_SIZE;
  if (align256((int64_t)groups * limbs * 4) > ws_bytes) return MX_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  MX_TRY(launch_rmodn(geo, d_mods, (u32*)d_ws, limbs, groups, s));
  mx::VerdictArgs a;
  a.v = d_v; a.pass = d_pass; a.mods = d_mods; a.rmodn = (const u32*)d_ws;
  a.groups = groups; a.n_slots = n_slots; a.limbs = limbs; a.n_parties = n_parties; a.nblk = geo.nblk;
  switch (geo.K) {
    case 1: return launch_verdict_k<1>(a, s);
    case 2: return launch_verdict_k<2>(a, s);
    case 4: return launch_verdict_k<4>(a, s);
    case 8: return launch_verdict_k<8>(a, s);
    case 16: return launch_verdict_k<16>(a, s);
    case 32: return launch_verdict_k<32>(a, s);
    case 64: return launch_verdict_k<64>(a, s);
  }
  return MX_ERR_SIZE;
}

extern "C" int mx_biprime_verdict(const uint32_t* d_v, uint8_t* d_pass, const uint32_t* h_mods, int limbs,
                                  int n_parties, int64_t groups, int64_t n_slots, void* d_ws, int64_t ws_bytes,
                                  void* stream) {
  if (!d_v || !d_pass || !h_mods || !d_ws || limbs <= 0 || n_parties <= 0 || groups <= 0 || n_slots <= 0)
    return MX_ERR_ARG;
  int max_bits = 0;
  for (int64_t g = 0; g < groups; ++g) {
    const u32* n = h_mods + g * limbs;
    if (!(n[0] & 1u)) return MX_ERR_MODULUS;
    int b = bit_length(n, limbs);
    if (b < 2) return MX_ERR_MODULUS;
    if (b > max_bits) max_bits = b;
  }
  int64_t part = align256((int64_t)groups * limbs * 4);
  if (2 * part > ws_bytes) return MX_ERR_WORKSPACE;
  char* ws = (char*)d_ws;
  MX_TRY(upload_words(ws, h_mods, (size_t)groups * limbs, (hipStream_t)stream));
  return mx_biprime_verdict_dev(d_v, d_pass, (const u32*)ws, limbs, max_bits, n_parties, groups, n_slots, ws + part,
                                ws_bytes - part, stream);
}

// ---- Jacobi symbol -------------------------------------------------------------------------
namespace {
template <int NL>
int launch_jacobi(const mx::JacobiArgs& a, hipStream_t s) {
  int64_t nblocks = a.skip ? (a.count / a.per_group) * ((a.per_group + 63) / 64) : (a.count + 63) / 64;
  // dynamic LDS: 0 except for the 257-word instance, whose top limbs live there (mx_jacobi.hpp)
  hipLaunchKernelGGL((mx::jacobi_kernel<NL>), dim3((unsigned)nblocks), dim3(64), mx::jacobi_lds_bytes<NL>(), s, a);
  MX_HIP(hipGetLastError());
  // safety net for symbols the divstep kernel did not finish within its batch bound (mx_jacobi.hpp)
  constexpr size_t lds_fb = mx::jacobi_lds_bytes<NL, true>();
  if constexpr (lds_fb > 64 * 1024) {
    static bool allowed[MX_MAX_DEVICES] = {};
    MX_HIP(mx_allow_dynamic_lds(reinterpret_cast<const void*>(&mx::jacobi_fallback_kernel<NL>), (int)lds_fb, allowed));
  }
  hipLaunchKernelGGL((mx::jacobi_fallback_kernel<NL>), dim3((unsigned)nblocks), dim3(64), lds_fb, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}
}  // namespace

extern "C" int64_t mx_jacobi_workspace_bytes(int limbs, int64_t groups) {
  if (limbs <= 0 || groups <= 0) return MX_ERR_ARG;
  return align256((int64_t)groups * limbs * 4);
}

extern "C" int mx_jacobi_dev_range(const uint32_t* d_values, int8_t* d_out, const uint32_t* d_mods, int limbs,
                                   int64_t groups, int64_t group_size, int first, int count, const int32_t* d_skip_counts,
                                   int skip_threshold, void* stream) {
  if (!d_values || !d_out || !d_mods || limbs <= 0 || groups <= 0 || group_size <= 0) return MX_ERR_ARG;
  if (first < 0 || count <= 0 || (int64_t)first + count > group_size) return MX_ERR_ARG;
  if (limbs > 257) return MX_ERR_SIZE;
  hipStream_t s = (hipStream_t)stream;
  mx::JacobiArgs a;
  a.a = d_values; a.mods = d_mods; a.out = (signed char*)d_out;
  a.count = groups * count; a.group_size = group_size; a.limbs = limbs;
  a.first = first; a.per_group = count; a.skip = d_skip_counts; a.skip_threshold = skip_threshold;
  a.max_batches = (32 * limbs * 9 / 2) / mx::JSTEPS + 8;
  if (g_knob_jacobi_max_batches > 0) a.max_batches = g_knob_jacobi_max_batches - 1;   // developer knob: exercises the safety net
  if (limbs <= 3) return launch_jacobi<3>(a, s);
  if (limbs <= 5) return launch_jacobi<5>(a, s);
  if (limbs <= 9) return launch_jacobi<9>(a, s);
  if (limbs <= 17) return launch_jacobi<17>(a, s);
  if (limbs <= 33) return launch_jacobi<33>(a, s);
  if (limbs <= 65) return launch_jacobi<65>(a, s);
  if (limbs <= 129) return launch_jacobi<129>(a, s);
  // key_length 8192 (the widest modulus the modexp kernels take for N^2): operands no longer fit the register file —
  // the limbs from 192 upwards live in LDS (mx_jacobi.hpp: jacobi_reg_limbs; one wavefront per SIMD, no scratch)
  return launch_jacobi<257>(a, s);
}

extern "C" int mx_jacobi_dev(const uint32_t* d_values, int8_t* d_out, const uint32_t* d_mods, int limbs,
                             int64_t groups, int64_t group_size, void* stream) {
  if (group_size > 0x7FFFFFFF) return MX_ERR_ARG;
  return mx_jacobi_dev_range(d_values, d_out, d_mods, limbs, groups, group_size, 0, (int)group_size, nullptr, 0, stream);
}

extern "C" int mx_jacobi(const uint32_t* d_values, int8_t* d_out, const uint32_t* h_mods, int limbs, int64_t groups,
                         int64_t group_size, void* d_ws, int64_t ws_bytes, void* stream) {
  if (!d_values || !d_out || !h_mods || !d_ws || limbs <= 0 || groups <= 0 || group_size <= 0) return MX_ERR_ARG;
  if (limbs > 257) return MX_ERR_SIZE;
  for (int64_t g = 0; g < groups; ++g)
    if (!(h_mods[g * limbs] & 1u)) return MX_ERR_MODULUS;
  if (align256((int64_t)groups * limbs * 4) > ws_bytes) return MX_ERR_WORKSPACE;
  MX_TRY(upload_words(d_ws, h_mods, (size_t)groups * limbs, (hipStream_t)stream));
  return mx_jacobi_dev(d_values, d_out, (const u32*)d_ws, limbs, groups, group_size, stream);
}

// ---- modular multiplication ------------------------------------------------------------------
namespace {
template <int K>
int launch_mulmod_k(const mx::MulmodArgs& a, hipStream_t s) {
  using M_t = mx::Mont<K, LIMBS_PER_LANE, LIMB_BITS, true>;
  int gpw = 64 / K;
  int64_t nblocks = (a.batch + gpw - 1) / gpw;
  size_t lds = (size_t)gpw * M_t::LDS_WORDS * 4;
  hipLaunchKernelGGL((mx::mulmod_kernel<K, LIMBS_PER_LANE, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}
}  // namespace

extern "C" int64_t mx_mulmod_workspace_bytes(int limbs) {
  if (limbs <= 0) return MX_ERR_ARG;
  return align256((int64_t)2 * limbs * 4);
}

extern "C" int mx_mulmod_shared(const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, const uint32_t* h_mod,
                                int limbs, int64_t batch, void* d_ws, int64_t ws_bytes, void* stream) {
  if (!d_a || !d_b || !d_out || !h_mod || !d_ws || limbs <= 0 || batch <= 0) return MX_ERR_ARG;
  if (!(h_mod[0] & 1u)) return MX_ERR_MODULUS;
  int bits = bit_length(h_mod, limbs);
  if (bits < 2) return MX_ERR_MODULUS;
  Geometry geo;
  if (!choose_geometry(bits, geo)) return MX_ERR_SIZE;
  if (align256((int64_t)2 * limbs * 4) > ws_bytes) return MX_ERR_WORKSPACE;
  std::vector<u32> c((size_t)2 * limbs);
  std::memcpy(c.data(), h_mod, (size_t)limbs * 4);
  two_pow_mod(c.data() + limbs, h_mod, limbs, geo.W * geo.L * geo.nblk);
  hipStream_t s = (hipStream_t)stream;
  MX_TRY(upload_words(d_ws, c.data(), c.size(), s));
  mx::MulmodArgs a;
  a.a = d_a; a.b = d_b; a.out = d_out; a.mod = (const u32*)d_ws; a.rmodn = (const u32*)d_ws + limbs;
  a.batch = batch; a.limbs = limbs; a.nblk = geo.nblk;
  switch (geo.K) {
    case 1: return launch_mulmod_k<1>(a, s);
    case 2: return launch_mulmod_k<2>(a, s);
    case 4: return launch_mulmod_k<4>(a, s);
    case 8: return launch_mulmod_k<8>(a, s);
    case 16: return launch_mulmod_k<16>(a, s);
    case 32: return launch_mulmod_k<32>(a, s);
    case 64: return launch_mulmod_k<64>(a, s);
  }
  return MX_ERR_SIZE;
}

// ---- Shamir-field arithmetic of the candidate moduli ----------------------------------------------
namespace {
template <int K>
int launch_field_k(const mx::FieldArgs& a, bool lincomb, hipStream_t s) {
  using M_t = mx::Mont<K, LIMBS_PER_LANE, LIMB_BITS, true>;
  int gpw = 64 / K;
  int64_t nblocks = (a.batch + gpw - 1) / gpw;
  size_t lds = (size_t)gpw * M_t::LDS_WORDS * 4;
  if (lincomb)
    hipLaunchKernelGGL((mx::lincomb_kernel<K, LIMBS_PER_LANE, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  else
    hipLaunchKernelGGL((mx::fma_kernel<K, LIMBS_PER_LANE, LIMB_BITS>), dim3((unsigned)nblocks), dim3(64), lds, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

int field_launch(mx::FieldArgs& a, bool lincomb, const uint32_t* h_mod, const uint32_t* h_extra, int extra_words,
                 int limbs, void* d_ws, int64_t ws_bytes, hipStream_t s) {
  if (!(h_mod[0] & 1u)) return MX_ERR_MODULUS;
  int bits = bit_length(h_mod, limbs);
  if (bits < 2) return MX_ERR_MODULUS;
  Geometry geo;
  if (!choose_geometry(bits, geo)) return MX_ERR_SIZE;
  if (align256((int64_t)(2 * limbs + extra_words) * 4) > ws_bytes) return MX_ERR_WORKSPACE;
  std::vector<u32> c((size_t)2 * limbs + extra_words);
  std::memcpy(c.data(), h_mod, (size_t)limbs * 4);
  two_pow_mod(c.data() + limbs, h_mod, limbs, geo.W * geo.L * geo.nblk);
  if (extra_words) std::memcpy(c.data() + 2 * limbs, h_extra, (size_t)extra_words * 4);
  MX_TRY(upload_words(d_ws, c.data(), c.size(), s));
  a.mod = (const u32*)d_ws; a.rmodn = (const u32*)d_ws + limbs;
  if (lincomb) a.b = (const u32*)d_ws + 2 * limbs;
  a.limbs = limbs; a.nblk = geo.nblk;
  switch (geo.K) {
    case 1: return launch_field_k<1>(a, lincomb, s);
    case 2: return launch_field_k<2>(a, lincomb, s);
    case 4: return launch_field_k<4>(a, lincomb, s);
    case 8: return launch_field_k<8>(a, lincomb, s);
    case 16: return launch_field_k<16>(a, lincomb, s);
    case 32: return launch_field_k<32>(a, lincomb, s);
    case 64: return launch_field_k<64>(a, lincomb, s);
  }
  return MX_ERR_SIZE;
}
}  // namespace

extern "C" int64_t mx_field_workspace_bytes(int limbs, int terms) {
  if (limbs <= 0 || terms < 0) return MX_ERR_ARG;
  return align256((int64_t)(2 + terms) * limbs * 4);
}

extern "C" int mx_fma_mod(const uint32_t* d_a, const uint32_t* d_b, const uint32_t* d_c, uint32_t* d_out,
                          const uint32_t* h_mod, int limbs, int64_t batch, void* d_ws, int64_t ws_bytes, void* stream) {
  if (!d_a || !d_b || !d_c || !d_out || !h_mod || !d_ws || limbs <= 0 || batch <= 0) return MX_ERR_ARG;
  mx::FieldArgs a;
  a.a = d_a; a.b = d_b; a.c = d_c; a.out = d_out; a.batch = batch; a.terms = 0;
  return field_launch(a, false, h_mod, nullptr, 0, limbs, d_ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mx_lincomb_mod(const uint32_t* d_x, const uint32_t* h_coeffs, uint32_t* d_out, const uint32_t* h_mod,
                              int limbs, int terms, int64_t batch, void* d_ws, int64_t ws_bytes, void* stream) {
  if (!d_x || !h_coeffs || !d_out || !h_mod || !d_ws || limbs <= 0 || terms <= 0 || batch <= 0) return MX_ERR_ARG;
  mx::FieldArgs a;
  a.a = d_x; a.b = nullptr; a.c = nullptr; a.out = d_out; a.batch = batch; a.terms = terms;
  return field_launch(a, true, h_mod, h_coeffs, terms * limbs, limbs, d_ws, ws_bytes, (hipStream_t)stream);
}

// ---- modular inverse (one wavefront per element) ---------------------------------------------------
namespace {
template <int LPL>
int launch_modinv(const mx::ModinvArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((mx::modinv_kernel<LPL>), dim3((unsigned)a.batch), dim3(64), 0, s, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}
}  // namespace

extern "C" int64_t mx_modinv_workspace_bytes(int limbs) {
  if (limbs <= 0) return MX_ERR_ARG;
  return align256((int64_t)limbs * 4);
}

extern "C" int mx_modinv(const uint32_t* d_values, uint32_t* d_out, uint8_t* d_status, const uint32_t* h_mod, int limbs,
                         int64_t batch, void* d_ws, int64_t ws_bytes, void* stream) {
  if (!d_values || !d_out || !d_status || !h_mod || !d_ws || limbs <= 0 || batch <= 0) return MX_ERR_ARG;
  if (!(h_mod[0] & 1u)) return MX_ERR_MODULUS;
  const int bits = bit_length(h_mod, limbs);
  if (bits < 2) return MX_ERR_MODULUS;
  if (bits > MAX_MOD_BITS) return MX_ERR_SIZE;
  if (align256((int64_t)limbs * 4) > ws_bytes) return MX_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  MX_TRY(upload_words(d_ws, h_mod, (size_t)limbs, s));
  mx::ModinvArgs a;
  a.vals = d_values; a.mod = (const u32*)d_ws; a.out = d_out; a.status = d_status; a.batch = batch; a.limbs = limbs;
  // the almost-inverse keeps values below 2 M, the word-wise halvings form x + q M with a one-word q: capacity
  // 64 * LPL words >= bits + 34 bits, and >= limbs words
  const int need = std::max(limbs, (bits + 34 + 31) / 32);
  if (need <= 64) return launch_modinv<1>(a, s);
  if (need <= 128) return launch_modinv<2>(a, s);
  if (need <= 192) return launch_modinv<3>(a, s);
  if (need <= 320) return launch_modinv<5>(a, s);
  if (need <= 576) return launch_modinv<9>(a, s);
  return MX_ERR_SIZE;
}

// ---- selection of the Jacobi-1 generators ---------------------------------------------------
extern "C" int mx_select_first(const uint32_t* d_rows, const int8_t* d_flags, uint32_t* d_out, int32_t* d_counts,
                               int limbs, int64_t groups, int group_size, int keep, void* stream) {
  if (!d_rows || !d_flags || !d_out || !d_counts || limbs <= 0 || groups <= 0 || group_size <= 0 || keep <= 0)
    return MX_ERR_ARG;
  mx::SelectArgs a;
  a.rows = d_rows; a.flags = (const signed char*)d_flags; a.out = d_out; a.counts = d_counts;
  a.groups = groups; a.group_size = group_size; a.keep = keep; a.limbs = limbs;
  hipLaunchKernelGGL(mx::select_first_kernel, dim3((unsigned)groups), dim3(64), 0, (hipStream_t)stream, a);
  MX_HIP(hipGetLastError());
  return MX_OK;
}

