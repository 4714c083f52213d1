// Microbenchmark: issue cost / throughput of the integer-VALU and cross-lane
// instructions the big-integer engine is built from, on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
// Output: one line per (instruction, waves/SIMD): cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int UNROLL = 64;   // instructions per loop body
constexpr int ITERS  = 2000;

enum Kind { MAD64_IND, MAD64_DEP, MAD64_ADDC, MUL_LO, MUL_HI, MAD_U24, MULHI_U24, ADD_ADDC, LSHR64, ALIGNBIT,
            FMA64, FMA32, DPP_SHL, DPP_BCAST, SWIZZLE, BPERM, READLANE, MADLO_I32, ADD3, MAD64_SGPR, NKINDS };
static const char* names[] = {"v_mad_u64_u32(indep x8)", "v_mad_u64_u32(dep chain)", "v_mad_u64_u32+v_addc pair", "v_mul_lo_u32", "v_mul_hi_u32",
  "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_add_co+v_addc_co pair", "v_lshrrev_b64", "v_alignbit_b32", "v_fma_f64", "v_fma_f32",
  "v_mov_dpp row_shl:1", "v_mov_dpp row_newbcast:0", "ds_swizzle_b32", "ds_bpermute_b32", "v_readlane_b32", "v_mad_i32_i24", "v_add3_u32", "v_mad_u64_u32(sgpr src)"};

template <int KIND>
__global__ void __launch_bounds__(1024) bench(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
  uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  uint64_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;
  uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
  double d0 = x, d1 = y, d2 = 1.0, d3 = 2.0;
  float f0 = x, f1 = y, f2 = 1.0f, f3 = 2.0f;
  uint32_t sy = __builtin_amdgcn_readfirstlane(y);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
      if constexpr (KIND == MAD64_IND) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
      } else if constexpr (KIND == MAD64_SGPR) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "s"(sy) : "vcc");
      } else if constexpr (KIND == MAD64_DEP) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                     : "+v"(a0) : "v"(x), "v"(y) : "vcc");
      } else if constexpr (KIND == MAD64_ADDC) {   // 4 pairs = 8 instructions
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n"
                     "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(x), "v"(y) : "vcc");
      } else if constexpr (KIND == MUL_LO || KIND == MUL_HI || KIND == MULHI_U24) {
        uint32_t* p = reinterpret_cast<uint32_t*>(&a0);
        uint32_t r0 = (uint32_t)a0, r1 = (uint32_t)a1, r2 = (uint32_t)a2, r3 = (uint32_t)a3, r4 = (uint32_t)a4, r5 = (uint32_t)a5, r6 = (uint32_t)a6, r7 = (uint32_t)a7;
        (void)p;
#define OP8(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n" \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(y))
        if constexpr (KIND == MUL_LO) OP8("v_mul_lo_u32");
        else if constexpr (KIND == MUL_HI) OP8("v_mul_hi_u32");
        else OP8("v_mul_hi_u32_u24");
        a0 = r0; a1 = r1; a2 = r2; a3 = r3; a4 = r4; a5 = r5; a6 = r6; a7 = r7;
      } else if constexpr (KIND == MAD_U24 || KIND == MADLO_I32 || KIND == ADD3 || KIND == ALIGNBIT) {
        uint32_t r0 = (uint32_t)a0, r1 = (uint32_t)a1, r2 = (uint32_t)a2, r3 = (uint32_t)a3, r4 = (uint32_t)a4, r5 = (uint32_t)a5, r6 = (uint32_t)a6, r7 = (uint32_t)a7;
#define OP8T(op) asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n" \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(y), "v"(x))
        if constexpr (KIND == MAD_U24) OP8T("v_mad_u32_u24");
        else if constexpr (KIND == MADLO_I32) OP8T("v_mad_i32_i24");
        else if constexpr (KIND == ADD3) OP8T("v_add3_u32");
        else OP8T("v_alignbit_b32");
        a0 = r0; a1 = r1; a2 = r2; a3 = r3; a4 = r4; a5 = r5; a6 = r6; a7 = r7;
      } else if constexpr (KIND == ADD_ADDC) {     // 4 pairs
        uint32_t l0 = (uint32_t)a0, l1 = (uint32_t)a1, l2 = (uint32_t)a2, l3 = (uint32_t)a3;
        asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %4, vcc, %4, %9, vcc\n v_add_co_u32 %1, vcc, %1, %8\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n"
                     "v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %6, vcc, %6, %9, vcc\n v_add_co_u32 %3, vcc, %3, %8\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n"
                     : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(x), "v"(y) : "vcc");
        a0 = l0; a1 = l1; a2 = l2; a3 = l3;
      } else if constexpr (KIND == LSHR64) {
        asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1\n v_lshrrev_b64 %2, 3, %2\n v_lshrrev_b64 %3, 3, %3\n"
                     "v_lshrrev_b64 %4, 3, %4\n v_lshrrev_b64 %5, 3, %5\n v_lshrrev_b64 %6, 3, %6\n v_lshrrev_b64 %7, 3, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if constexpr (KIND == FMA64) {
        asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                     "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001), "v"(0.5));
      } else if constexpr (KIND == FMA32) {
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001f), "v"(0.5f));
      } else if constexpr (KIND == DPP_SHL || KIND == DPP_BCAST) {
        uint32_t r0 = (uint32_t)a0, r1 = (uint32_t)a1, r2 = (uint32_t)a2, r3 = (uint32_t)a3;
        if constexpr (KIND == DPP_SHL)
          asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n"
                       "s_nop 1\n v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n"
                       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        else
          asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                       "s_nop 1\n v_mov_b32_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        a0 = r0; a1 = r1; a2 = r2; a3 = r3;
      } else if constexpr (KIND == SWIZZLE) {
        uint32_t r0 = (uint32_t)a0, r1 = (uint32_t)a1, r2 = (uint32_t)a2, r3 = (uint32_t)a3, r4 = (uint32_t)a4, r5 = (uint32_t)a5, r6 = (uint32_t)a6, r7 = (uint32_t)a7;
        r0 = __builtin_amdgcn_ds_swizzle(r0, 0x0010); r1 = __builtin_amdgcn_ds_swizzle(r1, 0x0010); r2 = __builtin_amdgcn_ds_swizzle(r2, 0x0010); r3 = __builtin_amdgcn_ds_swizzle(r3, 0x0010);
        r4 = __builtin_amdgcn_ds_swizzle(r4, 0x0010); r5 = __builtin_amdgcn_ds_swizzle(r5, 0x0010); r6 = __builtin_amdgcn_ds_swizzle(r6, 0x0010); r7 = __builtin_amdgcn_ds_swizzle(r7, 0x0010);
        a0 = r0; a1 = r1; a2 = r2; a3 = r3; a4 = r4; a5 = r5; a6 = r6; a7 = r7;
      } else if constexpr (KIND == BPERM) {
        uint32_t r0 = (uint32_t)a0, r1 = (uint32_t)a1, r2 = (uint32_t)a2, r3 = (uint32_t)a3, r4 = (uint32_t)a4, r5 = (uint32_t)a5, r6 = (uint32_t)a6, r7 = (uint32_t)a7;
        int addr = ((threadIdx.x + 1) & 63) * 4;
        r0 = __builtin_amdgcn_ds_bpermute(addr, r0); r1 = __builtin_amdgcn_ds_bpermute(addr, r1); r2 = __builtin_amdgcn_ds_bpermute(addr, r2); r3 = __builtin_amdgcn_ds_bpermute(addr, r3);
        r4 = __builtin_amdgcn_ds_bpermute(addr, r4); r5 = __builtin_amdgcn_ds_bpermute(addr, r5); r6 = __builtin_amdgcn_ds_bpermute(addr, r6); r7 = __builtin_amdgcn_ds_bpermute(addr, r7);
        a0 = r0; a1 = r1; a2 = r2; a3 = r3; a4 = r4; a5 = r5; a6 = r6; a7 = r7;
      } else if constexpr (KIND == READLANE) {
        uint32_t r0 = (uint32_t)a0, r1 = (uint32_t)a1, r2 = (uint32_t)a2, r3 = (uint32_t)a3;
        uint32_t s0, s1, s2, s3;
        asm volatile("s_nop 0\n v_readlane_b32 %4, %0, 0\n v_readlane_b32 %5, %1, 16\n v_readlane_b32 %6, %2, 32\n v_readlane_b32 %7, %3, 48\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3));
        a0 = r0; a1 = r1; a2 = r2; a3 = r3;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  r ^= (uint64_t)(h0 + h1 + h2 + h3);
  r ^= (uint64_t)(d0 + d1 + d2 + d3) ^ (uint64_t)(f0 + f1 + f2 + f3);
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(int wavesPerSimd, uint32_t* d_out, unsigned long long* d_cyc, int nCU) {
  int threads = 64 * 4 * wavesPerSimd;           // one block per CU, waves spread over 4 SIMDs
  int blocks = nCU;
  if (threads > 1024) { blocks = nCU * (threads / 1024); threads = 1024; }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  bench<KIND><<<blocks, threads>>>(d_out, d_cyc, 1);           // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  bench<KIND><<<blocks, threads>>>(d_out, d_cyc, 2);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  int nw = blocks * threads / 64;
  std::vector<unsigned long long> h(nw);
  CHECK(hipMemcpy(h.data(), d_cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double avg = 0; for (auto v : h) avg += (double)v; avg /= nw;
  double ninstr = (double)ITERS * UNROLL;
  // s_memtime ticks at 100 MHz constant on gfx9? report both tick-based and wall-based numbers
  double wavesSimd = (double)nw / (nCU * 4.0);
  double ns_per_instr_per_simd = (ms * 1e6) / (ninstr * wavesSimd);
  printf("%-28s waves/SIMD=%d  wall=%.3f ms  memtime_ticks/instr(wave)=%.3f  ns per wave-instr per SIMD=%.3f  (=%.2f cyc @2.4GHz)\n",
         names[KIND], wavesPerSimd, ms, avg / ninstr, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
}

template <int K> void sweep(uint32_t* o, unsigned long long* c, int nCU) { for (int w : {1, 2, 4, 8}) run<K>(w, o, c, nCU); }

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int nCU = p.multiProcessorCount;
  uint32_t* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, (size_t)nCU * 8 * 1024 * 4)); CHECK(hipMalloc(&d_cyc, (size_t)nCU * 8 * 16 * 8));
  sweep<FMA32>(d_out, d_cyc, nCU);
  sweep<MAD64_IND>(d_out, d_cyc, nCU);
  sweep<MAD64_SGPR>(d_out, d_cyc, nCU);
  sweep<MAD64_DEP>(d_out, d_cyc, nCU);
  sweep<MAD64_ADDC>(d_out, d_cyc, nCU);
  sweep<MUL_LO>(d_out, d_cyc, nCU);
  sweep<MUL_HI>(d_out, d_cyc, nCU);
  sweep<MAD_U24>(d_out, d_cyc, nCU);
  sweep<MULHI_U24>(d_out, d_cyc, nCU);
  sweep<MADLO_I32>(d_out, d_cyc, nCU);
  sweep<ADD3>(d_out, d_cyc, nCU);
  sweep<ADD_ADDC>(d_out, d_cyc, nCU);
  sweep<LSHR64>(d_out, d_cyc, nCU);
  sweep<ALIGNBIT>(d_out, d_cyc, nCU);
  sweep<FMA64>(d_out, d_cyc, nCU);
  sweep<DPP_SHL>(d_out, d_cyc, nCU);
  sweep<DPP_BCAST>(d_out, d_cyc, nCU);
  sweep<SWIZZLE>(d_out, d_cyc, nCU);
  sweep<BPERM>(d_out, d_cyc, nCU);
  sweep<READLANE>(d_out, d_cyc, nCU);
  return 0;
}
