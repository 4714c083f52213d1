// Cross-lane primitives for K-lane groups inside a 64-wide wavefront (gfx950).
//
// A big integer is spread over K consecutive lanes (K = 1,2,4,...,64).  Three movements are
// needed by the Montgomery engine, all of them on the VALU via DPP (data-parallel primitives:
// a lane permutation folded into the operand fetch of a v_mov) because on gfx950 a
// ds_bpermute/ds_swizzle costs 2-5x a VALU issue slot (profiles/r01_ubench_valu_rates.txt):
//   bcast0     every lane of a group receives the value held by the group's lane 0
//   from_next  lane p receives lane p+1's value, the group's top lane receives 0
//   from_prev  lane p receives lane p-1's value, the group's lane 0 receives 0
// A DPP "row" is 16 lanes, a "bank" 4 lanes.  Groups smaller than a row need the boundary lane
// masked explicitly (keep-masks are computed once per kernel).  The SHFL variants (ds_bpermute)
// are kept as the reference the self-test compares the DPP forms against.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mx {

typedef uint32_t u32;
typedef uint64_t u64;

template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND_ZERO>
__device__ __forceinline__ u32 dpp_mov(u32 old, u32 src) {
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, BOUND_ZERO);
}

// DPP control encodings (GFX9 family)
enum : int {
  DPP_QUAD_0000 = 0x00,   // quad_perm:[0,0,0,0]
  DPP_QUAD_0022 = 0xA0,   // quad_perm:[0,0,2,2]
  DPP_QUAD_1133 = 0xF5,   // quad_perm:[1,1,3,3]
  DPP_ROW_SHL1 = 0x101,   // lane i <- lane i+1 within a row
  DPP_ROW_SHR1 = 0x111,   // lane i <- lane i-1 within a row
  DPP_ROW_SHR4 = 0x114,
  DPP_WAVE_SHL1 = 0x130,  // lane i <- lane i+1 across the wave
  DPP_WAVE_SHR1 = 0x138,  // lane i <- lane i-1 across the wave
  DPP_ROW_BCAST15 = 0x142,  // lane 15 of each row -> every lane of the next row
  DPP_ROW_NEWBCAST0 = 0x150,  // lane 0 of each row -> every lane of the row (gfx90a+)
};

template <int K, bool USE_DPP = true>
struct Lanes {
  static_assert(K == 1 || K == 2 || K == 4 || K == 8 || K == 16 || K == 32 || K == 64, "K must be a power of two <= 64");

  // lane index inside the group
  static __device__ __forceinline__ int pos() { return (int)(threadIdx.x & (K - 1)); }

  static __device__ __forceinline__ u32 bcast0(u32 x) {
    if constexpr (K == 1) {
      return x;
    } else if constexpr (!USE_DPP) {
      return (u32)__shfl((int)x, 0, K);
    } else if constexpr (K == 2) {
      return dpp_mov<DPP_QUAD_0022, 0xF, 0xF, true>(0, x);
    } else if constexpr (K == 4) {
      return dpp_mov<DPP_QUAD_0000, 0xF, 0xF, true>(0, x);
    } else if constexpr (K == 8) {
      // lanes 4 and 12 first fetch lanes 0 and 8 (banks 1,3 <- banks 0,2), then every quad takes
      // its lane 0; the second move has full masks so it folds into the consumer
      u32 t = dpp_mov<DPP_ROW_SHR4, 0xF, 0xA, false>(x, x);
      return dpp_mov<DPP_QUAD_0000, 0xF, 0xF, true>(0, t);
    } else if constexpr (K == 16) {
      return dpp_mov<DPP_ROW_NEWBCAST0, 0xF, 0xF, true>(0, x);   // full masks + bound_ctrl: foldable into the consumer
    } else if constexpr (K == 32) {
      // the first move writes every lane (full masks, bound_ctrl): no "old" value, so x itself survives without a copy
      u32 t = dpp_mov<DPP_ROW_NEWBCAST0, 0xF, 0xF, true>(0, x);
      return dpp_mov<DPP_ROW_BCAST15, 0xA, 0xF, false>(t, t);  // rows 1,3 <- lane 15 of rows 0,2
    } else {
      return (u32)__builtin_amdgcn_readfirstlane((int)x);
    }
  }

  // bcast0(x) & m.  The mask commutes with the lane permutation; for 32-lane groups it is applied between the two
  // moves, where it folds into the first one (v_and_b32_dpp) — behind the second, partial, move it is an instruction.
  static __device__ __forceinline__ u32 bcast0_and(u32 x, u32 m) {
    if constexpr (USE_DPP && K == 32) {
      u32 t = dpp_mov<DPP_ROW_NEWBCAST0, 0xF, 0xF, true>(0, x) & m;
      return dpp_mov<DPP_ROW_BCAST15, 0xA, 0xF, false>(t, t);
    } else {
      return bcast0(x) & m;
    }
  }

  // keep-mask helpers: 0 for the boundary lane, ~0 otherwise
  static __device__ __forceinline__ u32 keep_next_mask() { return pos() == K - 1 ? 0u : ~0u; }
  static __device__ __forceinline__ u32 keep_prev_mask() { return pos() == 0 ? 0u : ~0u; }

  static __device__ __forceinline__ u32 from_next(u32 x, u32 keep_next) {
    if constexpr (K == 1) {
      return 0;
    } else if constexpr (!USE_DPP) {
      u32 r = (u32)__shfl_down((int)x, 1, K);
      return r & keep_next;
    } else if constexpr (K == 2) {
      return dpp_mov<DPP_QUAD_1133, 0xF, 0xF, false>(x, x) & keep_next;
    } else if constexpr (K == 16) {
      return dpp_mov<DPP_ROW_SHL1, 0xF, 0xF, true>(0, x);   // row end reads out of range -> 0
    } else if constexpr (K < 16) {
      return dpp_mov<DPP_ROW_SHL1, 0xF, 0xF, true>(0, x) & keep_next;
    } else if constexpr (K == 32) {
      return dpp_mov<DPP_WAVE_SHL1, 0xF, 0xF, true>(0, x) & keep_next;
    } else {
      return dpp_mov<DPP_WAVE_SHL1, 0xF, 0xF, true>(0, x);
    }
  }

  // from_next without the boundary mask: the caller ANDs with a per-lane VGPR mask that already
  // contains keep_next (so that the AND and the DPP move become one v_and_b32_dpp)
  static __device__ __forceinline__ u32 from_next_raw(u32 x) {
    if constexpr (K == 1) {
      return 0;
    } else if constexpr (!USE_DPP) {
      return (u32)__shfl_down((int)x, 1, K);
    } else if constexpr (K == 2) {
      return dpp_mov<DPP_QUAD_1133, 0xF, 0xF, true>(0, x);
    } else if constexpr (K <= 16) {
      return dpp_mov<DPP_ROW_SHL1, 0xF, 0xF, true>(0, x);
    } else {
      return dpp_mov<DPP_WAVE_SHL1, 0xF, 0xF, true>(0, x);
    }
  }

  static __device__ __forceinline__ u32 from_prev(u32 x, u32 keep_prev) {
    if constexpr (K == 1) {
      return 0;
    } else if constexpr (!USE_DPP) {
      u32 r = (u32)__shfl_up((int)x, 1, K);
      return r & keep_prev;
    } else if constexpr (K == 16) {
      return dpp_mov<DPP_ROW_SHR1, 0xF, 0xF, true>(0, x);
    } else if constexpr (K < 16) {
      return dpp_mov<DPP_ROW_SHR1, 0xF, 0xF, true>(0, x) & keep_prev;
    } else if constexpr (K == 32) {
      return dpp_mov<DPP_WAVE_SHR1, 0xF, 0xF, true>(0, x) & keep_prev;
    } else {
      return dpp_mov<DPP_WAVE_SHR1, 0xF, 0xF, true>(0, x);
    }
  }

  // rarely used (once per exponentiation): any lane of the group -> all lanes; goes through ds_bpermute
  static __device__ __forceinline__ u32 bcast_from(u32 x, int src_pos) {
    if constexpr (K == 1) return x;
    return (u32)__shfl((int)x, src_pos, K);
  }
  // true in every lane of the group iff pred holds in some lane of the group
  static __device__ __forceinline__ bool group_any(bool pred) {
    if constexpr (K == 1) return pred;
    unsigned long long m = __ballot(pred);
    if constexpr (K == 64) return m != 0;
    unsigned long long gm = ((1ull << K) - 1ull) << ((threadIdx.x & 63) & ~(K - 1));
    return (m & gm) != 0;
  }
};

}  // namespace mx
