// Does the VGPR bank (register index mod 4) of the operands of v_mad_u64_u32 matter on gfx950?
// Eight independent accumulators at v[20:21], v[24:25] ... (banks 0,1); the two 32-bit multiplicands
// are placed in chosen banks.  Build: hipcc -O3 --offload-arch=gfx950 mad_banks.hip -o mad_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define MAD(acc, a, b) "v_mad_u64_u32 v[" acc "], vcc, " a ", " b ", v[" acc "]\n"
#define EIGHT(a, b) MAD("20:21", a, b) MAD("24:25", a, b) MAD("28:29", a, b) MAD("32:33", a, b) \
                    MAD("36:37", a, b) MAD("40:41", a, b) MAD("44:45", a, b) MAD("48:49", a, b)
#define CLOBBERS "vcc", "v8", "v9", "v10", "v11", "v14", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", \
                 "v36", "v37", "v40", "v41", "v44", "v45", "v48", "v49"

template <int PATTERN>
__global__ void __launch_bounds__(64) bench(unsigned* out, int iters, unsigned seed) {
  unsigned x = threadIdx.x * 2654435761u + seed;
  asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v14, %0\n"
               "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n"
               "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n"
               "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n" : : "v"(x) : CLOBBERS);
  for (int it = 0; it < iters; ++it) {
    if constexpr (PATTERN == 0) asm volatile(EIGHT("v10", "v11") EIGHT("v10", "v11") : : : CLOBBERS);   // banks 2,3 | 0,1
    if constexpr (PATTERN == 1) asm volatile(EIGHT("v8", "v11") EIGHT("v8", "v11") : : : CLOBBERS);     // banks 0,3 | 0,1
    if constexpr (PATTERN == 2) asm volatile(EIGHT("v8", "v9") EIGHT("v8", "v9") : : : CLOBBERS);       // banks 0,1 | 0,1
    if constexpr (PATTERN == 3) asm volatile(EIGHT("v10", "v14") EIGHT("v10", "v14") : : : CLOBBERS);   // banks 2,2 | 0,1
    if constexpr (PATTERN == 4) asm volatile(EIGHT("v9", "v9") EIGHT("v9", "v9") : : : CLOBBERS);       // same register twice
  }
  unsigned r;
  asm volatile("v_xor_b32 %0, v20, v25\n v_xor_b32 %0, %0, v49" : "=v"(r) : : CLOBBERS);
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int P>
void run(const char* name, int waves_per_simd) {
  int blocks = 256 * 4 * waves_per_simd, iters = 400000;
  unsigned* out;
  hipMalloc(&out, (size_t)blocks * 64 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  bench<P><<<blocks, 64>>>(out, 100, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  bench<P><<<blocks, 64>>>(out, iters, 2);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double per = ms * 1e-3 / ((double)iters * 16 * waves_per_simd);   // seconds per wave-instruction per SIMD
  printf("%-34s waves/SIMD=%d  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4 GHz)\n", name, waves_per_simd, per * 1e9, per * 2.4e9);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 3, 4}) {
    run<0>("mul banks 2,3 | acc banks 0,1", w);
    run<1>("mul banks 0,3 | acc banks 0,1", w);
    run<2>("mul banks 0,1 | acc banks 0,1", w);
    run<3>("mul banks 2,2 | acc banks 0,1", w);
    run<4>("mul same register | acc 0,1", w);
  }
  return 0;
}
