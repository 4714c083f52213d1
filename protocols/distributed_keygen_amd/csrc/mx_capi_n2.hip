// extern "C" entry points of the N^2-modulus modexp (second translation unit of libmxpaillier.so:
// compiled in parallel with mx_capi.hip, the pair kernels are the most expensive to build).
#include "mx_upload.hpp"
#include "mx_powmod_n2_split.hpp"
#include "mx_bipair.hpp"

// ---- modexp modulo N^2 through pairs modulo N --------------------------------------------------
namespace {
// Shape of one launch: geometry, grid, table of pair slots in the caller's workspace.
struct N2Shape {
  Geometry geo;
  int64_t nblocks = 0, nlanes = 0;
  int nslots = 0;
  int64_t table_bytes = 0;
  int64_t groups = 0;          // groups of elements (one per wavefront or pair of wavefronts)
  int64_t sched_bytes = 0;     // scheduling words of the time-sliced form, behind the table in the workspace
};

// limbs_per_lane values of the pair kernel: 9 and 18 in both forms (one or two wavefronts per group of
// elements), 3 — the latency geometry — only in the two-wavefront form (mx_powmod_n2_split.hpp).  (2 limbs per
// lane were tried too: fewer instructions per limb step, but the quotient digit then travels through an SGPR
// (v_readfirstlane for 64-lane groups) and the dependent chain got longer, 18.5 vs 16.8 ms for one ciphertext.)
constexpr int N2_GEOS = 3;                      // constant sets of a plan, in geo_index order
constexpr int N2_LPLS[N2_GEOS] = {LIMBS_PER_LANE, LIMBS_PER_LANE_WIDE, LIMBS_PER_LANE_LAT};
inline int geo_index(int lpl) {
  for (int g = 0; g < N2_GEOS; ++g) if (N2_LPLS[g] == lpl) return g;
  return -1;
}
inline int max_lanes(int lpl, int wpg) {
  if (lpl == LIMBS_PER_LANE_WIDE) return 16;
  if (lpl == LIMBS_PER_LANE) return 32;
  return wpg == 2 ? 64 : 0;                 // L = 3 exists as a split kernel only
}

constexpr int N2_TIMESLICE_MAX_SEGMENTS = mx::N2_TS_LEVELS;      // units per group of a time-sliced launch, at most

bool shape_n2(int n_bits, int window, int64_t batch, int limbs_per_lane, int wpg, N2Shape& p) {
  if (wpg == 4) {                                                // the five-wavefront latency form (mx_bipair.hpp) shares the
    if (limbs_per_lane != LIMBS_PER_LANE_LAT) return false;      // slots of the two-wavefront 3-limb form
    wpg = 2;
  }
  if (geo_index(limbs_per_lane) < 0 || (wpg != 1 && wpg != 2)) return false;
  if (!choose_geometry(n_bits, p.geo, limbs_per_lane)) return false;
  if (p.geo.K > max_lanes(limbs_per_lane, wpg)) return false;   // instances that exist
  int gpw = (64 / p.geo.K) * (wpg == 2 ? mx::N2_SPLIT_PAIRS : 1);      // elements per workgroup
  p.nblocks = (batch + gpw - 1) / gpw;
  p.nlanes = p.nblocks * 64 * (wpg == 2 ? mx::N2_SPLIT_PAIRS : 1);
  p.nslots = mx::N2_SLOT_TABLE + (1 << (window - 1));
  p.table_bytes = align256((int64_t)p.nslots * 2 * p.geo.L * p.nlanes * 4);
  p.groups = (batch + 64 / p.geo.K - 1) / (64 / p.geo.K);
  p.sched_bytes = wpg == 2 ? align256((mx::N2_TS_HEADER + p.groups * (N2_TIMESLICE_MAX_SEGMENTS - 1)) * 4) : 0;
#ifdef MX_DEV_TS_TRACE          // four words per unit behind the queues (mx_powmod_n2_split.hpp)
  if (wpg == 2) p.sched_bytes += align256(p.groups * N2_TIMESLICE_MAX_SEGMENTS * 16);
#endif
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
// wide-geometry instantiations live in mx_capi_n2w.hip, the two-wavefront kernels in mx_capi_n2s.hip (L = 3, 9) and
// mx_capi_n2sw.hip (L = 18): translation units of their own, built in parallel
namespace mxw { int launch_n2_wide(int K, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s); }
namespace mxs {
int launch_n2_split(int K, int L, bool timesliced, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s);
int launch_n2_split_wide(int K, bool timesliced, const mx::PowmodN2Args& a, int64_t nblocks, hipStream_t s);
}
namespace mxb {
bool n2_bipair_instance(int K);
int launch_n2_bipair(int K, const mx::PowmodBiPairArgs& a, int64_t nblocks, hipStream_t s);
}
namespace mxl { int launch_bisetup(int K, const mx::BiSetupArgs& a, hipStream_t s); }
namespace {
int launch_n2(const mx::PowmodN2Args& a, const N2Shape& p, int wpg, hipStream_t s, int64_t timesliced_blocks = 0) {
  const int K = p.geo.K, L = p.geo.L;
  if (wpg == 2) {
    const bool ts = timesliced_blocks > 0;
    const int64_t nb = ts ? timesliced_blocks : p.nblocks;
    return L == LIMBS_PER_LANE_WIDE ? mxs::launch_n2_split_wide(K, ts, a, nb, s) : mxs::launch_n2_split(K, L, ts, a, nb, s);
  }
  if (L == LIMBS_PER_LANE_WIDE) return mxw::launch_n2_wide(K, a, p.nblocks, s);
  switch (K) {
    case 1: return launch_n2_kl<1, LIMBS_PER_LANE>(a, p.nblocks, s);
    case 2: return launch_n2_kl<2, LIMBS_PER_LANE>(a, p.nblocks, s);
    case 4: return launch_n2_kl<4, LIMBS_PER_LANE>(a, p.nblocks, s);
    case 8: return launch_n2_kl<8, LIMBS_PER_LANE>(a, p.nblocks, s);
    case 16: return launch_n2_kl<16, LIMBS_PER_LANE>(a, p.nblocks, s);
    case 32: return launch_n2_kl<32, LIMBS_PER_LANE>(a, p.nblocks, s);
  }
  return MX_ERR_SIZE;
}

// Which instances run their tape modulo the friendly multiple of N (mx_powmod_n2.hpp): every 3-limb instance (its
// geometry reserves the room), the 9-limb two-wavefront instances for groups of 8 and 16 lanes and the 18-limb
// one-wavefront instances for groups of 4 and 8 lanes (key_length 2048 and 4096) where the modulus leaves LIMB_BITS + 6
// bits of room in R.  One place decides; the launchers of the other translation units obey PowmodN2Args::friendly.
inline bool n2_friendly_instance(const Geometry& g, int wpg, int n_bits) {
  if (g.L == LIMBS_PER_LANE_LAT) return true;
  if (n_bits + 4 + LIMB_BITS + 2 > g.W * g.L * g.nblk) return false;
  if (wpg == 2) return g.L == LIMBS_PER_LANE && (g.K == 8 || g.K == 16);
  return g.L == LIMBS_PER_LANE_WIDE && (g.K == 4 || g.K == 8) && g_knob_n2_friendly_1w != 1;
}

// Launch shape of the pair kernel when the caller leaves the choice (limbs per lane and/or wavefronts per
// group) to the library: the candidate with the lowest estimated duration for ONE launch of this batch on an
// otherwise idle GPU.  The estimate is a cycle count per pair operation for the wavefront that bounds the launch,
// fitted to tools/sweep_shapes.py (profiles/r03_sweep_shapes.txt; within ~8 % of every measured point at
// key_length 2048 and 4096):
//   a limb step is m multiply-accumulates plus o other instructions (quotient digit, carry hand-over, shifts);
//   a wavefront that has its SIMD to itself issues one instruction every ~5.3 cycles whatever it is, and waits
//   ~80 / L cycles per step for the multiplier limbs it fetches from LDS once per L steps;
//   every further wavefront resident on the same SIMD adds what its instructions cost when the SIMD is shared:
//   4.2 cycles per multiply-accumulate, 2.3 per other instruction (tools/ubench/valu_peak.hip);
//   wavefronts beyond what the register file holds per SIMD (2 at L = 18, 3 at L = 9, 8 at L = 3) are back-filled
//   as the first ones finish.
// Two-wavefront groups: the second pass (m = 2L, o = 11) bounds an operation, its wavefronts sit on SIMDs 1 and 3 of
// a CU, one per workgroup.  One-wavefront groups: both passes (m = L/2 + 1 + 3L, o = 20), a wavefront per workgroup.
// What comes out (key_length 2048): up to ~1000 ciphertexts the latency geometry (3 limbs per lane, two
// wavefronts), to ~4000 L = 9 split, to ~8000 L = 18 split, and the one-wavefront wide kernel once a launch
// fills the machine on its own (from ~24 000).  Callers that keep several launches in flight fill the machine
// between them and should say so by passing limbs_per_lane = 18, wavefronts_per_group = 1 (bench.py's
// steady-state leg does).
struct N2Choice { int lpl, wpg, resident, units; };      // resident: workgroups per CU of the time-sliced form (0 = plain launch), units per group unless the caller says
inline int device_cus() { return mx_device_cus(); }      // of the CURRENT device (mx_upload.hpp)
double n2_estimate(int n_bits, int64_t batch, int lpl, int wpg, int* resident = nullptr, int* units = nullptr) {
  if (resident) *resident = 0;
  if (units) *units = 0;
  N2Shape p;
  if (!shape_n2(n_bits, 1, batch, lpl, wpg, p)) return -1.0;
  const double L = p.geo.L, steps = (double)p.geo.nblk * p.geo.L;
  // (3 limbs per lane: the friendly-modulus passes have 8 other instructions per step, and a second wavefront on
  // the SIMD costs them 1.22x what the issue costs alone would say — 13.0 / 19.9 / 27.4 / 35.3 ms for 1 .. 4 per SIMD)
  const bool lat = lpl == LIMBS_PER_LANE_LAT;
  const double m = wpg == 2 ? 2 * L : (p.geo.L / 2 + 1) + 3 * L, o = lat ? 8.0 : wpg == 2 ? 11.0 : 20.0;
  // (round 5 refit, after the build's alignment pass: a lone 18-limb wavefront runs 3 % / 8 % faster than these counts
  // say — 32.1 ms for 8192 ciphertexts on two wavefronts, 54.4 for 16 384 on one — and a second 9-limb pair on a SIMD
  // costs 8 % more (35.6 ms for 8192), a second 18-limb pair 18 % more (59.5 ms for 16 384); profiles/r05_ts_probe_2048.txt,
  // r05_sweep_shapes.txt)
  const bool wide_geo = lpl == LIMBS_PER_LANE_WIDE;
  const double alone = steps * (5.3 * (m + o) + 80.0 / L) * (wide_geo ? (wpg == 2 ? 0.967 : 0.924) : 1.0);
  const double shared = steps * (4.2 * m + 2.3 * o) * (lat ? 1.22 : wpg == 2 ? (wide_geo ? 1.18 : 1.08) : 1.17);     // one wavefront doing both passes overlaps less
  const int cus = device_cus();
  const int64_t per_simd = wpg == 2 ? (p.nblocks + cus - 1) / cus : (p.nblocks + 4 * cus - 1) / (4 * cus);
  const int64_t fit = lpl == LIMBS_PER_LANE_WIDE ? 2 : lpl == LIMBS_PER_LANE ? 3 : 8;
  // every further wavefront of a two-wavefront shape costs ~12 % more than the one before (L9: 21.3 / 35.7 / 51.1 ms
  // for 1 / 2 / 3 per SIMD; L3: increments of 7.4, 8.1, 8.9 ms)
  auto round = [&](int64_t r) {
    double t = r <= 0 ? 0.0 : alone, add = shared;
    for (int64_t k = 1; k < r; ++k) { t += add; if (wpg == 2) add *= 1.12; }
    return t;
  };
  double plain;
  if (per_simd <= fit) {
    plain = round(per_simd);          // everything resident at once: the fullest SIMD bounds the launch
  } else {
    // more wavefronts than fit: every further one per SIMD (whole ones: the fullest SIMD bounds the launch) adds
    // 0.93 of its share of a full round — measured at 9 limbs per lane, two-wavefront groups: 53.4 ms for 3 per
    // SIMD, then 70 / 86.5 / 102.5 for 4 / 5 / 6; at 18: 59.9 for 2, 89 for 3, 115 for 4
    plain = round(fit) * (1.0 + 0.93 * (double)(per_simd - fit) / (double)fit);
  }
  // Time-sliced form (mx_powmod_n2_split.hpp; instances: 9 limbs per lane, groups of at most 16 lanes): r workgroups
  // per CU stay resident and take the groups' segments from a queue.  Measured over r = 1..3 and 2..8 segments at
  // key_length 2048 and 4096 (tools/ts_probe.py, profiles/r03_ts_probe_*.txt), two settings pay:
  //   r = 2 with 2 segments: the launch costs groups / resident pairs x a full launch of 2 per SIMD (when a wavefront
  //     runs out of work its neighbour speeds up, so the coarse grain costs nothing) — 44-47 ms instead of 54 for
  //     9 200..10 500 ciphertexts at key_length 2048;
  //   r = 1 with 8 segments, for launches just above one workgroup per CU (up to 1.25x): the fine grain costs ~13 %
  //     per operation (hand-overs on a SIMD that has nothing else to issue), still 12-23 % below the plain launch.
  const bool wide = lpl == LIMBS_PER_LANE_WIDE;
  const bool sliceable = wpg == 2 && ((lpl == LIMBS_PER_LANE && p.geo.K <= 16) || (wide && (p.geo.K == 4 || p.geo.K == 8)));
  if (!sliceable || !resident || g_knob_n2_timeslice == 1) return plain;
  const bool forced = g_knob_n2_timeslice >= 2;
  double best = forced ? -1.0 : plain * 0.97;          // a time-sliced launch has to win by 3 %
  int best_units = 0;
  if (wide) {
    // 18 limbs per lane (round 5): ONE workgroup per CU — a second one doubles what every SIMD carries — and the groups'
    // units handed out most-work-left-first (mx_powmod_n2_split.hpp): the launch takes ceil(groups x units / pairs)
    // rounds of 1 / units of a full launch, plus 0.4 % per unit for the hand-overs (key_length 2048: 8704 ciphertexts
    // in 12 units 36.1 ms, 10 000 in 8 units 40.8, 11 264 in 8 units 45.2, 12 288 in 2 units 47.9 where the plain
    // launches take 51-60; key_length 4096: 4352 in 12 units 141 ms, 5000 in 8 units 151, 5632 in 8 units 166 where they
    // take 182; profiles/r05_ts_probe_*.txt).  Four units are not offered: 10 000 ciphertexts are 4.88 rounds of them, and
    // whether the launch then takes five or six (42 or 49 ms) changed from build to build.
    const int64_t pairs = (int64_t)cus * mx::N2_SPLIT_PAIRS;
    if (p.groups > pairs || forced) {
      for (int u : {2, 8, 12}) {
        const int64_t rounds = (p.groups * u + pairs - 1) / pairs;
        const double t = round(1) * (double)rounds / (double)u * (1.0 + 0.004 * u);
        if (best < 0 || t < best) { best = t; best_units = u; *resident = 1; }
      }
    }
    if (units) *units = best_units;
    return *resident ? best : plain;
  }
  for (int64_t r = 1; r <= 2; ++r) {
    int64_t rr = r;
    if (g_knob_n2_timeslice > 16) {               // developer: this many per CU
      if (r != 1) break;
      rr = std::min<int64_t>(fit, g_knob_n2_timeslice - 16);
    }
    const int64_t pairs = rr * cus * mx::N2_SPLIT_PAIRS;
    if (p.groups <= pairs && !forced) continue;
    const double load = std::max(1.0, (double)p.groups / (double)pairs);
    if (rr == 1 && load > 1.5 && !forced) continue;
    // the time-sliced instances run 5 % (shared SIMDs) to 11 % (alone) slower per operation than the plain ones since
    // the build aligns 64-bit instructions (asm_align.py: the plain instances gained, these did not), and one
    // workgroup per CU pays ~13 % for its hand-overs
    // (two per CU, two units: 1.19-1.26 x a full launch for loads of 1.03-1.25 — 42-45 ms for 8448 .. 10 240 ciphertexts)
    const double t = (rr == 2 ? std::max(load, 1.17) : load) * round(rr) * (rr == 1 ? 1.28 : 1.05);
    if (best < 0 || t < best) { best = t; *resident = (int)rr; best_units = rr == 1 ? 8 : 2; }
  }
  if (units) *units = best_units;
  return *resident ? best : plain;
}
bool bipair_geometry(int n_bits, Geometry& gb);
N2Choice n2_auto_shape(int n_bits, int64_t batch, int limbs_per_lane, int wpg) {
  if (wpg == 4) return N2Choice{LIMBS_PER_LANE_LAT, 4, 0, 0};      // explicit: the five-wavefront latency form (mx_bipair.hpp)
  N2Choice best{LIMBS_PER_LANE, 1, 0, 0};
  double best_t = -1.0;
  for (int l : N2_LPLS) {
    if (limbs_per_lane && l != limbs_per_lane) continue;
    for (int w : {1, 2}) {
      if (wpg && w != wpg) continue;
      int resident = 0, units = 0;
      const double t = n2_estimate(n_bits, batch, l, w, &resident, &units);
      if (t > 0 && (best_t < 0 || t < best_t)) { best_t = t; best = N2Choice{l, w, resident, units}; }
    }
  }
  if (best_t < 0) best = N2Choice{limbs_per_lane ? limbs_per_lane : LIMBS_PER_LANE, wpg ? wpg : 1, 0, 0};   // reported as MX_ERR_SIZE by the caller
  // Launches that the two-wavefront latency form would run with at most ONE workgroup of the five-wavefront form per compute
  // unit take that form where it exists (key_length 1024 / 2048 / 4096): both passes of every product on two wavefronts each,
  // 9.4 instead of 12.95 ms for 1 .. 512 ciphertexts at key_length 2048, 31.8 instead of 47 for 1 .. 256 at 4096
  // (tools/lone_decrypt_time.py, profiles/r06_lone_decrypt.txt); a second workgroup per unit costs more than it saves.
  if (wpg == 0 && best.lpl == LIMBS_PER_LANE_LAT && best.wpg == 2 && g_knob_n2_bipair != 1) {
    Geometry gb;
    if (bipair_geometry(n_bits, gb) && batch <= (int64_t)device_cus() * (64 / gb.K)) best.wpg = 4;
  }
  return best;
}

// Segments when the caller leaves the choice to the library: a launch whose wavefronts would live for
// tens of milliseconds is cut so that a burst of such launches drains at a finer grain (measured with
// bench.py --steps 20: the last round of 4 launches in flight costs ~3 % of the run unsegmented).
int n2_auto_segments(int n_sqr, int64_t nblocks) {
  if (g_knob_n2_segments >= 1) return g_knob_n2_segments;
  return (n_sqr >= 2048 && nblocks >= 256) ? 4 : 1;
}

inline int64_t n2_consts_words(int limbs_n) { return (int64_t)8 * limbs_n + 2 * (limbs_n + 1); }
inline int64_t n2_consts_bytes(int limbs_n) { return align256(n2_consts_words(limbs_n) * 4); }

// The five-wavefront latency form (mx_bipair.hpp): exists where the bipartite geometry of the modulus (mx_host.hpp: the
// pivot, Pd data positions) and the pair kernel's 3-limb geometry agree on lanes and blocks, and the kernel is instantiated.
bool bipair_geometry(int n_bits, Geometry& gb) {
  Geometry g3;
  if (!choose_geometry(n_bits, gb, LIMBS_PER_LANE_BI) || !choose_geometry(n_bits, g3, LIMBS_PER_LANE_LAT)) return false;
  if (gb.K != g3.K || gb.nblk != g3.nblk || !mxb::n2_bipair_instance(gb.K)) return false;
  const int pd = gb.L * gb.nblk;
  // The pivot of the PAIR form (tools/bipair_model.py: pair_geometry): 0.52 of the steps on the L wavefronts, to the nearest
  // block — 21 of 42 at key_length 1024, 39 of 75 at 2048, 75 of 147 at 4096.  The generic form's pivot (mx_host.hpp: 24, 39,
  // 54) is fitted to ITS halves; here the H wavefronts record fold digits or carry two product rows, and at K = 64 its 54
  // left the L wavefronts waiting for 45 % of every slot (tools/bp_phase_probe.py; one decrypt 38.9 -> 31.5 ms, 3.55 -> 3.27 ms
  // at key_length 1024; profiles/r06_bipair_pivot_sweep.txt).
  if (g_knob_bi_pivot <= 0) {
    const int steps = pd + gb.L;
    int h = gb.L * ((52 * steps + 150) / (100 * gb.L));
    if (h > steps - gb.L) h = steps - gb.L;
    if (h < gb.L) h = gb.L;
    const int lo_min = gb.L * ((pd - (n_bits - 2) / gb.W + gb.L - 1) / gb.L);
    if (h < lo_min) h = lo_min;
    gb.h_lo = h;
  }
  return gb.h_lo < pd && LIMB_BITS * (pd - gb.h_lo) < n_bits - 1;      // E = 2^(W (Pd - hL)) is a digit below N
}
// its section of a plan's device block, behind the tape: constants for R' = 2^(W hL) | fold rows | quotient rows
inline int64_t bipair_fold_bytes() { return align256((int64_t)mx::BI_ROWS * 3 * 64 * 4); }
inline int64_t bipair_quot_bytes() { return align256((int64_t)mx::BP_QROWS * 3 * 64 * 4); }
inline int64_t bipair_section_bytes(int limbs_n) { return n2_consts_bytes(limbs_n) + bipair_fold_bytes() + bipair_quot_bytes(); }
inline int64_t bipair_section_offset(int limbs_n) { return N2_GEOS * n2_consts_bytes(limbs_n) + align256((int64_t)MAX_SLIDING_OPS * 4); }
constexpr int N2_GEO_BIPAIR = 8;      // mx_nsquare_plan::geometries: the plan holds the constants of the five-wavefront form


// The eight constant rows of one geometry (R = 2^m), each limbs_n words:
//   N | ONE0 ONE1 | K1_0 K1_1 | K2_0 K2_1 | C'
// followed by two rows of limbs_n + 1 words for the passes modulo the friendly multiple N~ = u N, u = -N^-1 mod 2^W
// (mx_powmod_n2.hpp; read by the 3-limb instances):
//   N~ + 1 | C2' = C2 - u (R - 1),  C2 = N * ceil(u (R - 1) / N)
void n2_constants(u32* c, const u32* h_n, int limbs_n, int m, int k) {
  const int l2 = 2 * limbs_n;
  std::vector<u32> n2(l2), tmp(l2), qq(l2), rr(limbs_n);
  mul_words(n2.data(), h_n, limbs_n, h_n, limbs_n);
  std::memset(c, 0, (size_t)n2_consts_words(limbs_n) * 4);
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
  pow2_mod(rr.data(), h_n, limbs_n, m);          // (R below N for the five-wavefront form: R = 2^(W hL))
  u64 borrow = 0, carry = 1;
  u32* cp = &c[(size_t)7 * limbs_n];
  for (int i = 0; i < limbs_n; ++i) {
    u64 d = (u64)h_n[i] - rr[i] - borrow;
    borrow = (d >> 63) & 1;
    u64 e = (u64)(u32)d + carry;
    cp[i] = (u32)e;
    carry = e >> 32;
  }
  // ---- friendly rows
  u32 inv = h_n[0];                                            // Newton: N^-1 mod 2^32
  for (int i = 0; i < 5; ++i) inv *= 2u - h_n[0] * inv;
  const u32 u = (0u - inv) & ((1u << LIMB_BITS) - 1u);
  u32* nt = &c[(size_t)8 * limbs_n];
  mul_words(nt, h_n, limbs_n, &u, 1);                          // N~ = u N  (limbs_n + 1 words), = -1 mod 2^W
  for (int i = 0; i <= limbs_n; ++i) { if (++nt[i] != 0u) break; }      // + 1
  // C2' = (u (1 - R)) mod N = (u * ((N + 1 - R mod N) mod N)) mod N      (rr = R mod N from above, != 0)
  std::vector<u32> d(limbs_n), prod(limbs_n + 1), quo(limbs_n + 1), rem(limbs_n);
  u64 br = 0, ca = 1;
  for (int i = 0; i < limbs_n; ++i) {
    u64 x = (u64)h_n[i] - rr[i] - br;
    br = (x >> 63) & 1;
    u64 e = (u64)(u32)x + ca;
    d[i] = (u32)e;
    ca = e >> 32;
  }
  mul_words(prod.data(), d.data(), limbs_n, &u, 1);
  divmod_words(quo.data(), rem.data(), prod.data(), limbs_n + 1, h_n, limbs_n);
  std::memcpy(&c[(size_t)8 * limbs_n + (limbs_n + 1)], rem.data(), (size_t)limbs_n * 4);
}
}  // namespace

extern "C" int mx_nsquare_launch_shape(int n_bits, int64_t batch, int limbs_per_lane, int wavefronts_per_group, int* k,
                                       int* l, int* w, int* blocks, int* wavefronts) {
  if (!k || !l || !w || !blocks || !wavefronts || batch <= 0) return MX_ERR_ARG;
  if (limbs_per_lane != 0 && geo_index(limbs_per_lane) < 0) return MX_ERR_ARG;
  if (wavefronts_per_group == 4) {                         // the five-wavefront latency form: explicit only (mx_bipair.hpp)
    Geometry gb;
    if ((limbs_per_lane != 0 && limbs_per_lane != LIMBS_PER_LANE_LAT) || !bipair_geometry(n_bits, gb)) return MX_ERR_SIZE;
    *k = gb.K; *l = gb.L; *w = gb.W; *blocks = gb.nblk; *wavefronts = 4;
    return MX_OK;
  }
  if (wavefronts_per_group < 0 || wavefronts_per_group > 2) return MX_ERR_ARG;
  const N2Choice ch = n2_auto_shape(n_bits, batch, limbs_per_lane, wavefronts_per_group);
  N2Shape p;
  if (!shape_n2(n_bits, 1, batch, ch.lpl, ch.wpg, p)) return MX_ERR_SIZE;
  *k = p.geo.K; *l = p.geo.L; *w = p.geo.W; *blocks = p.geo.nblk; *wavefronts = ch.wpg;
  return MX_OK;
}

// Shape for launches that are IN FLIGHT TOGETHER as pieces of `total` elements (several streams, several keys): the plain
// form with the lowest estimate for one launch of the total — the pieces then resemble it — never a time-sliced one, which
// only a lone launch can be (4 x 2500 ciphertexts at key_length 2048: 214 k/s at 9 limbs per lane on two wavefronts, 179 k/s
// in the shape of the time-sliced choice for a lone 10 000).
extern "C" int mx_nsquare_pieces_shape(int n_bits, int64_t total, int limbs_per_lane, int wavefronts_per_group,
                                       int* limbs_per_lane_out, int* wavefronts_per_group_out) {
  if (!limbs_per_lane_out || !wavefronts_per_group_out || total <= 0) return MX_ERR_ARG;
  if (limbs_per_lane != 0 && geo_index(limbs_per_lane) < 0) return MX_ERR_ARG;
  if (wavefronts_per_group < 0 || wavefronts_per_group > 2) return MX_ERR_ARG;
  double best_t = -1.0;
  for (int l : N2_LPLS) {
    if (limbs_per_lane && l != limbs_per_lane) continue;
    for (int w : {1, 2}) {
      if (wavefronts_per_group && w != wavefronts_per_group) continue;
      double t = n2_estimate(n_bits, total, l, w);                // (no `resident` out-parameter: the plain launch)
      // pieces of the wide two-wavefront form overlap better than one launch of their sum (4 x 5000 ciphertexts: 279 k/s
      // against 257 k/s at 9 limbs per lane, where the estimates for one launch of 20 000 say 87 against 83 ms)
      if (l == LIMBS_PER_LANE_WIDE && w == 2) t *= 0.93;
      if (t > 0 && (best_t < 0 || t < best_t)) { best_t = t; *limbs_per_lane_out = l; *wavefronts_per_group_out = w; }
    }
  }
  return best_t < 0 ? MX_ERR_SIZE : MX_OK;
}

extern "C" int mx_nsquare_launch_timesliced(int n_bits, int64_t batch, int limbs_per_lane, int wavefronts_per_group,
                                            int* resident_per_cu, int* units_per_group) {
  if (!resident_per_cu || !units_per_group || batch <= 0) return MX_ERR_ARG;
  if (limbs_per_lane != 0 && geo_index(limbs_per_lane) < 0) return MX_ERR_ARG;
  if (wavefronts_per_group == 4) { *resident_per_cu = 0; *units_per_group = 0; return MX_OK; }      // never time-sliced
  if (wavefronts_per_group < 0 || wavefronts_per_group > 2) return MX_ERR_ARG;
  const N2Choice ch = n2_auto_shape(n_bits, batch, limbs_per_lane, wavefronts_per_group);
  N2Shape p;
  if (!shape_n2(n_bits, 1, batch, ch.lpl, ch.wpg, p)) return MX_ERR_SIZE;
  *resident_per_cu = ch.resident;
  *units_per_group = ch.resident ? ch.units : 0;
  return MX_OK;
}

extern "C" int mx_nsquare_latency_form(int n_bits, int* lanes, int* positions, int* pivot, int64_t* max_batch) {
  if (n_bits < 2) return MX_ERR_ARG;
  Geometry gb;
  if (!bipair_geometry(n_bits, gb)) return MX_ERR_SIZE;
  if (lanes) *lanes = gb.K;
  if (positions) *positions = gb.L * gb.nblk;
  if (pivot) *pivot = gb.h_lo;
  if (max_batch) *max_batch = (int64_t)device_cus() * (64 / gb.K);
  return MX_OK;
}

extern "C" int mx_nsquare_launch_instance(int n_bits, int64_t batch, int limbs_per_lane, int wavefronts_per_group,
                                          int* k, int* l, int* wavefronts, int* friendly, int* timesliced) {
  if (!k || !l || !wavefronts || !friendly || !timesliced || batch <= 0) return MX_ERR_ARG;
  if (limbs_per_lane != 0 && geo_index(limbs_per_lane) < 0) return MX_ERR_ARG;
  if (wavefronts_per_group == 4) {
    Geometry gb;
    if ((limbs_per_lane != 0 && limbs_per_lane != LIMBS_PER_LANE_LAT) || !bipair_geometry(n_bits, gb)) return MX_ERR_SIZE;
    *k = gb.K; *l = gb.L; *wavefronts = 4; *friendly = 1; *timesliced = 0;
    return MX_OK;
  }
  if (wavefronts_per_group < 0 || wavefronts_per_group > 2) return MX_ERR_ARG;
  const N2Choice ch = n2_auto_shape(n_bits, batch, limbs_per_lane, wavefronts_per_group);
  N2Shape p;
  if (!shape_n2(n_bits, 1, batch, ch.lpl, ch.wpg, p)) return MX_ERR_SIZE;
  *k = p.geo.K; *l = p.geo.L; *wavefronts = ch.wpg;
  *friendly = n2_friendly_instance(p.geo, ch.wpg, n_bits) ? 1 : 0;
  *timesliced = ch.resident > 0 ? 1 : 0;
  return MX_OK;
}

// A lone launch above a capacity step of the wide two-wavefront shape (one workgroup of 2 x 64/K elements per CU: 8192
// ciphertexts at key_length 2048) pays for a second workgroup per CU: 55-59 ms for 10 000-12 288 instead of 33 for 8192.
// A caller that owns a second stream has one more option: the first `cap` elements in that shape, the rest at 9 limbs
// per lane on two wavefronts AT THE SAME TIME — a 9-limb workgroup (168 registers per wavefront) fits beside the 18-limb
// one (256) on a CU but not beside another 9-limb one, so the dispatcher spreads the remainder one workgroup per CU.
// Measured (tools/sweep_split.py, profiles/r04_split_launch.txt): a CU that hosts both takes 48-49 ms whatever the
// share of such CUs — the two wavefronts of a SIMD add up almost fully (a lone wavefront already issues 83 % of what
// its SIMD can).  Rounds 3-4 took the split where the single launches were slower still (10 752 .. 12 288 ciphertexts at
// key_length 2048: 48.5 instead of 51-56 ms).  Since round 5 the time-sliced 18-limb launch covers that range in 44-48 ms
// (n2_estimate above): the library no longer proposes a split by itself; the developer knob still forces one (tests,
// tools/sweep_split.py).
extern "C" int mx_nsquare_launch_split(int n_bits, int64_t batch, int64_t* first_rows, int* first_lpl, int* first_wpg,
                                       int* rest_lpl, int* rest_wpg) {
  if (!first_rows || !first_lpl || !first_wpg || !rest_lpl || !rest_wpg || batch <= 0) return MX_ERR_ARG;
  *first_rows = 0; *first_lpl = *first_wpg = *rest_lpl = *rest_wpg = 0;
  if (g_knob_n2_split == 1) return MX_OK;
  N2Shape wide, narrow;
  if (!shape_n2(n_bits, 1, batch, LIMBS_PER_LANE_WIDE, 2, wide) || !shape_n2(n_bits, 1, batch, LIMBS_PER_LANE, 2, narrow)) return MX_OK;
  const int cus = device_cus();
  const int64_t per_wg_wide = (int64_t)mx::N2_SPLIT_PAIRS * (64 / wide.geo.K), per_wg_narrow = (int64_t)mx::N2_SPLIT_PAIRS * (64 / narrow.geo.K);
  const int64_t cap = (int64_t)cus * per_wg_wide;
  if (batch <= cap || batch >= 2 * cap) return MX_OK;
  const int64_t rest = batch - cap;
  const int64_t rest_wgs = (rest + per_wg_narrow - 1) / per_wg_narrow;
  if (g_knob_n2_split != 2) return MX_OK;
  if (rest_wgs > cus) return MX_OK;
  *first_rows = cap; *first_lpl = LIMBS_PER_LANE_WIDE; *first_wpg = 2; *rest_lpl = LIMBS_PER_LANE; *rest_wpg = 2;
  return MX_OK;
}

extern "C" int mx_nsquare_geometry_for(int n_bits, int64_t batch, int limbs_per_lane, int* k, int* l, int* w,
                                       int* blocks) {
  int waves = 0;
  return mx_nsquare_launch_shape(n_bits, batch, limbs_per_lane, 0, k, l, w, blocks, &waves);
}

extern "C" int mx_nsquare_geometry(int n_bits, int64_t batch, int* k, int* l, int* w, int* blocks) {
  return mx_nsquare_geometry_for(n_bits, batch, 0, k, l, w, blocks);
}

extern "C" int64_t mx_nsquare_plan_bytes(int limbs_n, int exp_limbs) {
  if (limbs_n <= 0 || exp_limbs <= 0) return MX_ERR_ARG;
  return bipair_section_offset(limbs_n) + bipair_section_bytes(limbs_n);
}

extern "C" int mx_powmod_nsquare_prepare(mx_nsquare_plan* plan, const uint32_t* h_n, const uint32_t* h_exp,
                                         int limbs_n, int exp_limbs, void* d_plan, int64_t plan_bytes, void* stream) {
  return mx_powmod_nsquare_prepare_ex(plan, h_n, h_exp, limbs_n, exp_limbs, 0, d_plan, plan_bytes, stream);
}

// Width of the fixed-window schedule (MX_PLAN_FIXED_WINDOW): every window costs a multiplication, so narrower windows
// than the sliding schedule's pay; the table holds x^1 .. x^(2^w - 1) (digit 0 multiplies by the domain's one).
static int fixed_window_n2(int exp_bits) {
  int best = 1;
  long bestc = -1;
  for (int w = 1; w <= 7; ++w) {
    const long c = (exp_bits + w - 1) / w + (1L << w) - 2;
    if (bestc < 0 || c < bestc) { bestc = c; best = w; }
  }
  return best;
}

extern "C" int mx_powmod_nsquare_prepare_ex(mx_nsquare_plan* plan, const uint32_t* h_n, const uint32_t* h_exp,
                                            int limbs_n, int exp_limbs, int flags, void* d_plan, int64_t plan_bytes,
                                            void* stream) {
  if (!plan || !h_n || !h_exp || !d_plan) return MX_ERR_ARG;
  if (flags & ~MX_PLAN_FIXED_WINDOW) return MX_ERR_ARG;
  if (limbs_n <= 0 || exp_limbs <= 0) return MX_ERR_ARG;
  if (!(h_n[0] & 1u)) return MX_ERR_MODULUS;
  const int bits = bit_length(h_n, limbs_n);
  if (bits < 2) return MX_ERR_MODULUS;
  // constants of every geometry that has an instance for this modulus (they differ in R = 2^(W*L*blocks))
  const int* lpls = N2_LPLS;
  Geometry geos[N2_GEOS];
  int geometries = 0;
  for (int g = 0; g < N2_GEOS; ++g)
    if (choose_geometry(bits, geos[g], lpls[g]) && geos[g].K <= max_lanes(lpls[g], 2)) geometries |= 1 << g;
  if (!(geometries & 1)) return MX_ERR_SIZE;
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
    if (op == mx::N2_MUL || op == mx::N2_MULC) n_mul += 1;
    if (op == mx::N2_STORE) n_writes += 1;
    if (op == mx::N2_MUL || op == mx::N2_MULC || op == mx::N2_ADD || op == mx::N2_LOAD) n_reads += 1;
  };
  if (ebits == 0) {
    emit(mx::N2_LOAD, mx::N2_SLOT_ONE);
  } else if (flags & MX_PLAN_FIXED_WINDOW) {
    // Fixed windows of fw bits from the least significant end: the SEQUENCE of operations — fw squarings and one
    // multiplication per window, whatever the digits — depends on the exponent's bit length only, not on its bits
    // (the sliding schedule's run lengths and multiplication count do).  Which table row a window reads still does.
    const int fw = fixed_window_n2(ebits);
    auto digit = [&](int d) {
      u32 v = 0;
      for (int b = fw - 1; b >= 0; --b) {
        const int i = d * fw + b;
        v = (v << 1) | ((i < 32 * exp_limbs) ? ((h_exp[i >> 5] >> (i & 31)) & 1u) : 0u);
      }
      return (int)v;
    };
    auto slot_of = [&](int dg) { return dg ? mx::N2_SLOT_TABLE + dg - 1 : mx::N2_SLOT_ONE; };
    emit(mx::N2_LOAD, mx::N2_SLOT_LO); emit(mx::N2_MUL, mx::N2_SLOT_K1); emit(mx::N2_STORE, mx::N2_SLOT_TMP);
    emit(mx::N2_LOAD, mx::N2_SLOT_HI); emit(mx::N2_MUL, mx::N2_SLOT_K2); emit(mx::N2_ADD, mx::N2_SLOT_TMP);
    emit(mx::N2_STORE, mx::N2_SLOT_TABLE);
    for (int t = 2; t < (1 << fw); ++t) { emit(mx::N2_MUL, mx::N2_SLOT_TABLE); emit(mx::N2_STORE, mx::N2_SLOT_TABLE + t - 1); }
    const int nwin = (ebits + fw - 1) / fw;
    emit(mx::N2_LOAD, slot_of(digit(nwin - 1)));
    for (int d = nwin - 2; d >= 0; --d) { emit(mx::N2_SQR, fw); emit(mx::N2_MUL, slot_of(digit(d))); }
    w = fw + 1;                              // the table region: 2^(w - 1) = 2^fw pair slots (one more than used)
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
  emit(mx::N2_MULC, mx::N2_SLOT_E);          // the last product: digits below 2N for the epilogue (mx_powmod_n2.hpp)
  if ((int)tape.size() > MAX_SLIDING_OPS) return MX_ERR_SIZE;

  const int64_t cb = n2_consts_bytes(limbs_n);
  hipStream_t s = (hipStream_t)stream;
  char* dp = (char*)d_plan;
  int done_m[N2_GEOS] = {};
  std::vector<u32> rows[N2_GEOS];
  for (int g = 0; g < N2_GEOS; ++g) {
    if (!(geometries & (1 << g))) continue;
    const int m = geos[g].W * geos[g].L * geos[g].nblk;
    done_m[g] = m;
    int same = -1;
    for (int h = 0; h < g; ++h) if (done_m[h] == m) same = h;       // geometries often share R (2048: all three)
    if (same >= 0) {
      rows[g] = rows[same];
    } else {
      rows[g].resize((size_t)n2_consts_words(limbs_n));
      n2_constants(rows[g].data(), h_n, limbs_n, m, k);
    }
    MX_TRY(upload_words(dp + g * cb, rows[g].data(), rows[g].size(), s));
  }
  MX_TRY(upload_words(dp + N2_GEOS * cb, tape.data(), tape.size(), s));
  // ---- the five-wavefront latency form (mx_bipair.hpp): constants for R' = 2^(W hL), the fold rows (computed on the
  // device from N by the bipartite form's setup kernel) and the quotients of the folds, floor(2^(W (Pd + k)) / N)
  Geometry gb;
  if (bipair_geometry(bits, gb)) {
    char* bp = dp + bipair_section_offset(limbs_n);
    std::vector<u32> rows((size_t)n2_consts_words(limbs_n));
    n2_constants(rows.data(), h_n, limbs_n, gb.W * gb.h_lo, k);
    MX_TRY(upload_words(bp, rows.data(), rows.size(), s));
    const int pd = gb.L * gb.nblk, pw = gb.L * gb.K;
    std::vector<u32> quot((size_t)mx::BP_QROWS * pw, 0u);
    for (int r = 0; r < mx::BP_QROWS; ++r) {
      const int m = gb.W * (pd + r);                            // 2^m / N
      const int la = m / 32 + 1;
      std::vector<u32> num(la, 0u), q(la, 0u), rem(limbs_n, 0u);
      num[m / 32] = 1u << (m % 32);
      divmod_words(q.data(), rem.data(), num.data(), la, h_n, limbs_n);
      for (int i = 0; i < pw; ++i) {                            // W-bit limb i of the quotient
        const int bit = gb.W * i, w0 = bit >> 5, off = bit & 31;
        u64 v = w0 < la ? q[w0] : 0u;
        if (w0 + 1 < la) v |= (u64)q[w0 + 1] << 32;
        quot[(size_t)r * pw + i] = (u32)(v >> off) & ((1u << gb.W) - 1u);
      }
      if (bit_length(q.data(), la) > gb.W * pw) return MX_ERR_SIZE;
    }
    char* qp = bp + n2_consts_bytes(limbs_n) + bipair_fold_bytes();
    MX_TRY(upload_words(qp, quot.data(), quot.size(), s));
    mx::BiSetupArgs sa;
    sa.mods = (const u32*)bp;                                   // row 0 of the constants: N
    sa.consts = (u32*)(bp + n2_consts_bytes(limbs_n));
    sa.groups = 1; sa.limbs = limbs_n; sa.nblk = gb.nblk; sa.pd = pd; sa.h_lo = gb.h_lo;
    MX_TRY(mxl::launch_bisetup(gb.K, sa, s));
    geometries |= N2_GEO_BIPAIR;
  }
  plan->d_plan = d_plan;
  plan->plan_bytes = plan_bytes;
  plan->limbs_n = limbs_n;
  plan->n_bits = bits;
  plan->exp_bits = ebits;
  plan->window = w;
  plan->ntape = (int)tape.size();
  plan->n_sqr = n_sqr;
  plan->n_mul = n_mul;
  plan->geometries = geometries;
  plan->n_slot_reads = n_reads;
  plan->n_slot_writes = n_writes;
  return MX_OK;
}

extern "C" int64_t mx_powmod_nsquare_run_workspace_bytes(const mx_nsquare_plan* plan, int64_t batch) {
  if (!plan || batch <= 0 || plan->window < 1) return MX_ERR_ARG;
  // the largest table any launch shape of this batch needs (the caller may leave the choice to the library)
  int64_t most = -1;
  for (int l : N2_LPLS) {
    N2Shape p;
    if (shape_n2(plan->n_bits, plan->window, batch, l, 2, p) && p.table_bytes + p.sched_bytes > most) most = p.table_bytes + p.sched_bytes;
  }
  return most < 0 ? MX_ERR_SIZE : most;
}

extern "C" int mx_powmod_nsquare_run(const mx_nsquare_plan* plan, const uint32_t* d_bases, uint32_t* d_out,
                                     int limbs2, int64_t batch, int limbs_per_lane, int wavefronts_per_group,
                                     int segments, void* d_ws, int64_t ws_bytes, void* stream) {
  if (!plan || !plan->d_plan || !d_bases || !d_out || !d_ws) return MX_ERR_ARG;
  if (limbs2 <= 0 || batch <= 0 || plan->limbs_n <= 0 || plan->ntape <= 0) return MX_ERR_ARG;
  if (limbs_per_lane != 0 && geo_index(limbs_per_lane) < 0) return MX_ERR_ARG;
  if (wavefronts_per_group < 0 || (wavefronts_per_group > 2 && wavefronts_per_group != 4)) return MX_ERR_ARG;
  if (segments < 0 || segments > 64) return MX_ERR_ARG;
  const int bits = plan->n_bits;
  if (2 * bits - 1 > 32 * limbs2) return MX_ERR_ARG;          // rows too narrow for N^2
  if (wavefronts_per_group == 4 && limbs_per_lane != 0 && limbs_per_lane != LIMBS_PER_LANE_LAT) return MX_ERR_ARG;
  N2Choice ch = n2_auto_shape(bits, batch, limbs_per_lane, wavefronts_per_group);
  if (ch.wpg == 4 && wavefronts_per_group != 4 && !(plan->geometries & N2_GEO_BIPAIR)) ch.wpg = 2;      // a plan of an older layout
  if (ch.wpg == 4) {
    // the five-wavefront latency form: its own kernel for everything in front of the last product, then the last product and
    // the epilogue as a last segment of the two-wavefront 3-limb kernel, whose slots it shares
    Geometry gb;
    if (!(plan->geometries & N2_GEO_BIPAIR) || !bipair_geometry(bits, gb)) return MX_ERR_SIZE;
    N2Shape p;
    if (!shape_n2(bits, plan->window, batch, LIMBS_PER_LANE_LAT, 2, p)) return MX_ERR_SIZE;
    if (p.table_bytes > ws_bytes) return MX_ERR_WORKSPACE;
    if (2 * p.geo.K * p.geo.L + 8 < limbs2 + 2) return MX_ERR_ARG;
    const int64_t cb = n2_consts_bytes(plan->limbs_n);
    const char* dp = (const char*)plan->d_plan;
    const char* bp = dp + bipair_section_offset(plan->limbs_n);
    hipStream_t s = (hipStream_t)stream;
    const int pd = gb.L * gb.nblk;
    mx::PowmodBiPairArgs b;
    b.bases = d_bases;
    b.consts = (const u32*)bp;
    b.fold = (const u32*)(bp + cb);
    b.quot = (const u32*)(bp + cb + bipair_fold_bytes());
    b.tape = (const u32*)(dp + N2_GEOS * cb);
    b.slots = (u32*)d_ws;
    b.batch = batch; b.nlanes = p.nlanes;
    b.limbsn = plan->limbs_n; b.limbs2 = limbs2; b.ntape = plan->ntape;
    b.nblk = gb.nblk; b.pd = pd; b.h_lo = gb.h_lo; b.ksplit = bits - 1;
    b.nc = (gb.W * (pd + 6) - bits + 1) / gb.W + 1;           // c = floor(2^(W (Pd + 6)) / N) <= 2^(W (Pd + 6) - bits + 1)
    if (b.nc > 11) return MX_ERR_SIZE;
    b.pos_end = 0x7FFFFFFF;
    b.e_pos = pd - gb.h_lo;
    MxKernelTimer timer(s);
    MX_TRY(mxb::launch_n2_bipair(gb.K, b, p.nblocks * mx::N2_SPLIT_PAIRS, s));
    mx::PowmodN2Args a;
    a.bases = d_bases; a.out = d_out;
    a.consts = (const u32*)(dp + geo_index(LIMBS_PER_LANE_LAT) * cb);
    a.tape = b.tape; a.ntape = plan->ntape; a.slots = (u32*)d_ws;
    a.batch = batch; a.limbsn = plan->limbs_n; a.limbs2 = limbs2; a.nblk = p.geo.nblk; a.ksplit = bits - 1;
    a.friendly = 1;
    a.sched = nullptr; a.sched_groups = a.sched_segments = a.sched_n_sqr = 0;
    // (the last product is the last word of the tape: the segment is handed that word alone, at position 1 — walking the
    // whole tape to it, a dependent scalar load per word, was 100 of this launch's 136 us)
    a.tape = b.tape + (plan->ntape - 1); a.ntape = 1;
    a.first = 0; a.last = 1; a.pos_begin = 1; a.pos_end = 0x7FFFFFFF;
    return launch_n2(a, p, 2, s);
  }
  N2Shape p;
  if (!shape_n2(bits, plan->window, batch, ch.lpl, ch.wpg, p)) return MX_ERR_SIZE;
  const int gi = geo_index(ch.lpl);
  if (!(plan->geometries & (1 << gi))) return MX_ERR_SIZE;
  if (p.table_bytes + (ch.resident ? p.sched_bytes : 0) > ws_bytes) return MX_ERR_WORKSPACE;
  if (2 * p.geo.K * p.geo.L + 8 < limbs2 + 2) return MX_ERR_ARG;   // row wider than the staging area
  const int64_t cb = n2_consts_bytes(plan->limbs_n);
  const char* dp = (const char*)plan->d_plan;
  mx::PowmodN2Args a;
  a.bases = d_bases; a.out = d_out;
  a.consts = (const u32*)(dp + gi * cb);
  a.tape = (const u32*)(dp + N2_GEOS * cb);
  a.ntape = plan->ntape;
  a.slots = (u32*)d_ws;
  a.batch = batch; a.limbsn = plan->limbs_n; a.limbs2 = limbs2; a.nblk = p.geo.nblk; a.ksplit = bits - 1;
  a.friendly = n2_friendly_instance(p.geo, ch.wpg, bits);
  hipStream_t s = (hipStream_t)stream;
  // segments: consecutive launches that each execute a stretch of the tape (mx_powmod_n2.hpp); positions
  // are counted in squarings, the accumulator travels through a scratch slot of the workspace
  int nseg = segments > 0 ? segments : n2_auto_segments(plan->n_sqr, p.nlanes / 64 * ch.wpg);
  if (nseg > plan->n_sqr / 16) nseg = plan->n_sqr / 16;
  if (nseg < 1) nseg = 1;
  MxKernelTimer timer(s);                        // one timed interval per exponentiation (all its segments)
  if (ch.resident > 0) {
    // time-sliced: one launch of resident workgroups, the segments are units of its own scheduler
    nseg = segments > 0 ? segments : ch.units;
    if (nseg > N2_TIMESLICE_MAX_SEGMENTS) nseg = N2_TIMESLICE_MAX_SEGMENTS;
    if (nseg > plan->n_sqr / 16) nseg = plan->n_sqr / 16;
    if (nseg < 1) nseg = 1;
    a.sched = (u32*)((char*)d_ws + p.table_bytes);
    a.sched_groups = (int)p.groups; a.sched_segments = nseg; a.sched_n_sqr = plan->n_sqr;
    a.first = a.last = 1; a.pos_begin = 0; a.pos_end = 0x7FFFFFFF;
    MX_HIP(hipMemsetAsync(a.sched, 0, (size_t)(mx::N2_TS_HEADER + p.groups * (nseg - 1)) * 4, s));
    // no more workgroups than there are groups for their pairs
    int64_t wgs = (int64_t)ch.resident * device_cus();
    if (wgs > p.nblocks) wgs = p.nblocks;
    return launch_n2(a, p, ch.wpg, s, wgs);
  }
  a.sched = nullptr; a.sched_groups = a.sched_segments = a.sched_n_sqr = 0;
  // The friendly-modulus instances of the one-wavefront wide kernel (mx_capi_n2w.hip: groups of 4 and 8 lanes) run the
  // tape up to, not including, its last product (N2_MULC, tape position n_sqr + 1); that product and the epilogue run as
  // one more segment on the plain instance (mx_powmod_n2.hpp).
  const bool fr_tape = ch.wpg == 1 && a.friendly;
  for (int sg = 0; sg < nseg; ++sg) {
    const bool final_sg = sg == nseg - 1;
    a.first = sg == 0;
    a.last = final_sg && !fr_tape;
    a.pos_begin = (int)((int64_t)plan->n_sqr * sg / nseg);
    a.pos_end = final_sg ? (fr_tape ? plan->n_sqr + 1 : 0x7FFFFFFF) : (int)((int64_t)plan->n_sqr * (sg + 1) / nseg);
    MX_TRY(launch_n2(a, p, ch.wpg, s));
  }
  if (fr_tape) {
    a.first = 0; a.last = 1; a.friendly = 0;
    a.tape += plan->ntape - 1; a.ntape = 1;          // (the last word of the tape alone, as in the latency form above)
    a.pos_begin = 1; a.pos_end = 0x7FFFFFFF;
    MX_TRY(launch_n2(a, p, ch.wpg, s));
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
  return mx_powmod_nsquare_run(&plan, d_bases, d_out, limbs2, batch, 0, 0, 0,
                               (char*)d_ws + pb, ws_bytes - pb, stream);
}
