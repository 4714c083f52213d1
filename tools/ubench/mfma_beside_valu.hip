// Can the i8 matrix pipe (with its B tiles streaming from LDS) run beside a v_mad_u64_u32 stream
// without slowing it down?  (the shared-modulus half of the Montgomery products as
// Toeplitz matrix products.)  Per loop iteration: 32 multiply-adds (VALU), and/or NM MFMAs
// v_mfma_i32_16x16x64_i8 whose B operand (1 KB per MFMA) is re-read from LDS every time.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_beside_valu.hip -o mfma_beside_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int VALU, int NM, bool BLDS>
__global__ void __launch_bounds__(64) bench(unsigned* out, int iters, unsigned seed) {
  __shared__ v4i tiles[64 * 8];                    // 8 B tiles of 1 KB
  unsigned x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  for (int k = 0; k < 8; ++k) tiles[k * 64 + threadIdx.x] = v4i{(int)(x + k), (int)(y + k), (int)(x ^ k), (int)(y ^ k)};
  __syncthreads();
  unsigned long long a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;
  unsigned r0 = x, r1 = y, r2 = x + 1, r3 = y + 1, r4 = x + 2, r5 = y + 2, r6 = x + 3, r7 = y + 3;
  v4i acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  v4i af = {(int)x, (int)y, (int)(x + 7), (int)(y + 9)};
  for (int it = 0; it < iters; ++it) {
    if constexpr (VALU == 2) {      // 32 plain 32-bit adds instead of the multiply-adds
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
                     "v_add_u32 %4, %9, %4\n v_add_u32 %5, %9, %5\n v_add_u32 %6, %9, %6\n v_add_u32 %7, %9, %7\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x), "v"(y));
    }
    if constexpr (VALU == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      v4i b;
      if constexpr (BLDS) {
        b = tiles[((it + m) & 7) * 64 + threadIdx.x];         // ds_read_b128, a different tile each time
      } else {
        b = af;
        b[0] += it;                                           // register-resident B (one VALU op to vary it)
      }
      if (m & 1) acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, b, acc1, 0, 0, 0);
      else       acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, b, acc0, 0, 0, 0);
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) ^ (unsigned)(acc0[0] + acc0[3] + acc1[1] + acc1[2]);
}

template <int VALU, int NM, bool BLDS>
double run(int waves_per_simd, int iters) {
  int blocks = 256 * 4 * waves_per_simd;
  unsigned* out;
  hipMalloc(&out, (size_t)blocks * 64 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  bench<VALU, NM, BLDS><<<blocks, 64>>>(out, 1000, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  bench<VALU, NM, BLDS><<<blocks, 64>>>(out, iters, 2);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipFree(out);
  return ms * 1e-3 / iters / waves_per_simd * 2.4e9;     // cycles per iteration per wave-slot of a SIMD
}

int main() {
  const int iters = 100000;
  for (int w : {2, 3, 4}) {
    printf("waves/SIMD=%d, cycles per iteration per SIMD\n", w);
    printf("  32 v_mad_u64_u32 alone %.1f | 32 v_add_u32 alone %.1f\n", run<1, 0, true>(w, iters), run<2, 0, true>(w, iters));
    printf("  MFMA alone, B from LDS:  x1 %.1f  x2 %.1f  x4 %.1f | B in registers: x1 %.1f  x2 %.1f  x4 %.1f\n",
           run<0, 1, true>(w, iters), run<0, 2, true>(w, iters), run<0, 4, true>(w, iters),
           run<0, 1, false>(w, iters), run<0, 2, false>(w, iters), run<0, 4, false>(w, iters));
    printf("  32 MACs + MFMA (LDS B):  x1 %.1f  x2 %.1f  x4 %.1f | (register B): x1 %.1f  x2 %.1f  x4 %.1f\n",
           run<1, 1, true>(w, iters), run<1, 2, true>(w, iters), run<1, 4, true>(w, iters),
           run<1, 1, false>(w, iters), run<1, 2, false>(w, iters), run<1, 4, false>(w, iters));
    printf("  32 adds + MFMA (LDS B):  x1 %.1f  x2 %.1f  x4 %.1f | (register B): x1 %.1f  x2 %.1f  x4 %.1f\n",
           run<2, 1, true>(w, iters), run<2, 2, true>(w, iters), run<2, 4, true>(w, iters),
           run<2, 1, false>(w, iters), run<2, 2, false>(w, iters), run<2, 4, false>(w, iters));
  }
  return 0;
}
