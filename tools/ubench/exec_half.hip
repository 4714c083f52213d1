// Does a wave64 VALU instruction whose upper (or lower) 32 lanes are disabled in EXEC issue faster on gfx950?
//
// MI355X_MICROARCH.md describes the vector ALU as SIMD-32: a wave64 instruction takes two passes of 32 lanes
// (tools/ubench/valu_peak.hip measured 2.3 cycles for plain VALU and 4.2 for integer multiplies).  If a pass whose
// 32 lanes are all inactive were skipped, a kernel could run "half wavefronts" — twice the wavefronts with half the
// elements each at the same instruction count — for launches that leave SIMDs idle.  This
// microbenchmark times streams of independent v_mad_u64_u32 / v_add_u32 with EXEC = all 64 lanes, the low 32, the
// high 32, every other lane (32 active, spread over both halves), and 1 lane, at 1, 2 and 4 wavefronts per SIMD.
//
// Build: hipcc -O3 --offload-arch=gfx950 exec_half.hip -o exec_half
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4000;     // x 4 blocks of 16 instructions
#define R16(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7) op(8) op(9) op(10) op(11) op(12) op(13) op(14) op(15)

template <int KIND>     // 0: v_mad_u64_u32, 1: v_add_u32
__global__ void __launch_bounds__(1024) stream_kernel(uint32_t* out, unsigned long long* ticks, unsigned long long mask, uint32_t seed) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const uint32_t x = threadIdx.x * 2654435761u + seed, y = (x ^ 0x9e3779b9u) | 1u;
  uint32_t a[16];
  uint64_t w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = x + i; w[i] = ((uint64_t)y << 20) + i; }
  unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  if ((mask >> lane) & 1ull) {          // the compiler narrows EXEC to the active lanes for the whole timed loop
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if constexpr (KIND == 0) {
#define OPW(i) "v_mad_u64_u32 %" #i ", vcc, %16, %17, %" #i "\n"
          asm volatile(R16(OPW) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
                                  "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15])
                                : "v"(x), "v"(y) : "vcc");
        } else {
#define OPA(i) "v_add_u32 %" #i ", %" #i ", %16\n"
          asm volatile(R16(OPA) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                                  "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                                : "v"(y));
        }
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  const int first = __builtin_ctzll(mask);
  if (lane == first) { ticks[2 * wave] = t1 - t0; ticks[2 * wave + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, const char* mname, unsigned long long mask, int wavesPerSimd, uint32_t* d_out, unsigned long long* d_ticks, int nCU, bool report) {
  int threads = 64 * 4 * wavesPerSimd, blocks = nCU;
  if (threads > 1024) { blocks = nCU * (threads / 1024); threads = 1024; }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  stream_kernel<KIND><<<blocks, threads>>>(d_out, d_ticks, mask, 7u);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  if (!report) return;
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = blocks * threads / 64;
  std::vector<unsigned long long> h(2 * (size_t)nw);
  CHECK(hipMemcpy(h.data(), d_ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double st = 0, sr = 0;
  for (int i = 0; i < nw; ++i) { st += (double)h[2 * i]; sr += (double)h[2 * i + 1]; }
  const double clock_mhz = st / sr * 100.0;
  const double ninstr = (double)ITERS * 64;
  // wall-clock view (the one to read): every SIMD issues ninstr x wavesPerSimd instructions during the dispatch.  The
  // in-kernel figure (mean ticks of a wavefront's own timed loop) understates the cost with several wavefronts per SIMD,
  // whose loops do not overlap for their whole length.
  const double cyc_wall = ms * 1e-3 * clock_mhz * 1e6 / (ninstr * wavesPerSimd);
  const double cyc_loop = (st / nw) / (ninstr * wavesPerSimd);
  printf("%-14s EXEC=%-12s waves/SIMD=%d  wall=%.3f ms  clock=%.0f MHz  cycles per wave-instruction per SIMD: %.2f (wall clock; "
         "wavefront's own loop: %.2f)\n", name, mname, wavesPerSimd, ms, clock_mhz, cyc_wall, cyc_loop);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int nCU = p.multiProcessorCount;
  printf("device %s  CUs=%d  %d instructions per wavefront per dispatch\n", p.name, nCU, ITERS * 64);
  uint32_t* d_out; unsigned long long* d_ticks;
  CHECK(hipMalloc(&d_out, (size_t)nCU * 2 * 1024 * 4));
  CHECK(hipMalloc(&d_ticks, (size_t)nCU * 32 * 2 * 8));
  struct { const char* n; unsigned long long m; } masks[] = {
      {"all 64", ~0ull}, {"low 32", 0xFFFFFFFFull}, {"high 32", 0xFFFFFFFF00000000ull}, {"even lanes", 0x5555555555555555ull},
      {"low 16", 0xFFFFull}, {"lane 0", 1ull}};
  run<0>("warm-up", "all 64", ~0ull, 4, d_out, d_ticks, nCU, false);
  for (auto& mk : masks)
    for (int w : {1, 2, 4}) run<0>("v_mad_u64_u32", mk.n, mk.m, w, d_out, d_ticks, nCU, true);
  for (auto& mk : masks)
    for (int w : {1, 2, 4}) run<1>("v_add_u32", mk.n, mk.m, w, d_out, d_ticks, nCU, true);
  return 0;
}
