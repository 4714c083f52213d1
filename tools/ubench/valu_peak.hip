// Microbenchmark that settles the VALU issue peak used by bench.py's roofline ("weak" 2).
//
// For each instruction form, a stream of 16 fully INDEPENDENT instructions (16 different destination
// registers, sources that no instruction of the stream writes) is issued 16 x ITERS times by every
// wavefront, at 1/2/4/8 wavefronts per SIMD and with 1..4 SIMDs of every CU populated (wavefronts that
// land on a SIMD with id >= simds leave at once; HW_REG_HW_ID bits 5:4).  Reported per configuration:
//   wall time (HIP events), cycles per wave-instruction per POPULATED SIMD at the MEASURED shader clock,
//   the shader clock itself = s_memtime ticks / s_memrealtime ticks x 100 MHz, measured inside the kernel.
// Every configuration is exactly ONE dispatch (after one warm-up dispatch per instruction form), in the
// order printed, so that a rocprofv3 --pmc pass (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES,
// GRBM_GUI_ACTIVE) of the same binary can be joined by dispatch order (tools/ubench/join_valu_peak.py).
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4000;     // x 4 blocks of 16 = 256 000 instructions per wavefront
enum Kind { FMA32, PKFMA32, ADD32, MULLO, MAD64, MAD64_CARRYCHAIN, NK };
static const char* kname[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_mul_lo_u32", "v_mad_u64_u32",
                              "v_mad_u64_u32 (acc chain x16)"};

#define R16(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7) op(8) op(9) op(10) op(11) op(12) op(13) op(14) op(15)

template <int KIND>
__global__ void __launch_bounds__(1024) stream_kernel(uint32_t* out, unsigned long long* ticks, int simds, uint32_t seed) {
  uint32_t hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const int simd = (hwid >> 4) & 3;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (simd >= simds) {
    if ((threadIdx.x & 63) == 0) { ticks[3 * wave] = 0; ticks[3 * wave + 1] = 0; ticks[3 * wave + 2] = simd; }
    return;
  }
  const uint32_t x = threadIdx.x * 2654435761u + seed, y = (x ^ 0x9e3779b9u) | 1u;
  uint32_t a[16];
  uint64_t w[16];
  float f[16];
  double d[16];   // 64-bit containers of float2 for v_pk_fma_f32
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = x + i; w[i] = ((uint64_t)y << 20) + i; f[i] = (float)(x & 0xff) + i; d[i] = 0.0; }
  const float fa = 1.0000001f, fb = 0.5f;
  double pa, pb;
  { float2 t = {fa, fa}; pa = *reinterpret_cast<double*>(&t); float2 u = {fb, fb}; pb = *reinterpret_cast<double*>(&u); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (KIND == FMA32) {
#define OPF(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
        asm volatile(R16(OPF) : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                                "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15])
                              : "v"(fa), "v"(fb));
      } else if constexpr (KIND == PKFMA32) {
#define OPP(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %17\n"
        asm volatile(R16(OPP) : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                                "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15])
                              : "v"(pa), "v"(pb));
      } else if constexpr (KIND == ADD32) {
#define OPA(i) "v_add_u32 %" #i ", %" #i ", %16\n"
        asm volatile(R16(OPA) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                                "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                              : "v"(y));
      } else if constexpr (KIND == MULLO) {
#define OPM(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
        asm volatile(R16(OPM) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                                "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                              : "v"(y));
      } else if constexpr (KIND == MAD64) {
        // 16 independent 64-bit accumulators, sources x and y are never written
#define OPW(i) "v_mad_u64_u32 %" #i ", vcc, %16, %17, %" #i "\n"
        asm volatile(R16(OPW) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
                                "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15])
                              : "v"(x), "v"(y) : "vcc");
      } else {
        // the form of the Montgomery inner loop: 16 different multiplicand registers, one multiplier,
        // 16 different accumulators (t[j] += a[j] * b_i)
#define OPC(i) "v_mad_u64_u32 %" #i ", vcc, %" #i "+16, %32, %" #i "\n"
        asm volatile(
            "v_mad_u64_u32 %0, vcc, %16, %32, %0\n v_mad_u64_u32 %1, vcc, %17, %32, %1\n v_mad_u64_u32 %2, vcc, %18, %32, %2\n"
            "v_mad_u64_u32 %3, vcc, %19, %32, %3\n v_mad_u64_u32 %4, vcc, %20, %32, %4\n v_mad_u64_u32 %5, vcc, %21, %32, %5\n"
            "v_mad_u64_u32 %6, vcc, %22, %32, %6\n v_mad_u64_u32 %7, vcc, %23, %32, %7\n v_mad_u64_u32 %8, vcc, %24, %32, %8\n"
            "v_mad_u64_u32 %9, vcc, %25, %32, %9\n v_mad_u64_u32 %10, vcc, %26, %32, %10\n v_mad_u64_u32 %11, vcc, %27, %32, %11\n"
            "v_mad_u64_u32 %12, vcc, %28, %32, %12\n v_mad_u64_u32 %13, vcc, %29, %32, %13\n v_mad_u64_u32 %14, vcc, %30, %32, %14\n"
            "v_mad_u64_u32 %15, vcc, %31, %32, %15\n"
            : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
              "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
              "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(y)
            : "vcc");
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32) ^ __float_as_uint(f[i]) ^ (uint32_t)__double_as_longlong(d[i]);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) { ticks[3 * wave] = t1 - t0; ticks[3 * wave + 1] = r1 - r0; ticks[3 * wave + 2] = simd; }
}

static int g_dispatch = 0;

template <int KIND>
void run(int wavesPerSimd, int simds, uint32_t* d_out, unsigned long long* d_ticks, int nCU, bool report) {
  // one workgroup = 4 x w wavefronts (w per SIMD, the dispatcher deals a workgroup's wavefronts round
  // robin over the SIMDs of its CU); workgroups of more than 16 wavefronts are split into two per CU
  int threads = 64 * 4 * wavesPerSimd, blocks = nCU;
  if (threads > 1024) { blocks = nCU * (threads / 1024); threads = 1024; }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  stream_kernel<KIND><<<blocks, threads>>>(d_out, d_ticks, simds, 7u + g_dispatch);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  const int disp = g_dispatch++;
  if (!report) return;
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = blocks * threads / 64;
  std::vector<unsigned long long> h(3 * (size_t)nw);
  CHECK(hipMemcpy(h.data(), d_ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double st = 0, sr = 0, tmax = 0; long active = 0; long persimd[4] = {0, 0, 0, 0};
  for (int i = 0; i < nw; ++i) {
    persimd[h[3 * i + 2] & 3]++;
    if (h[3 * i + 1] == 0) continue;
    st += (double)h[3 * i]; sr += (double)h[3 * i + 1]; active++;
    if ((double)h[3 * i] > tmax) tmax = (double)h[3 * i];
  }
  const double ninstr = (double)ITERS * 64;
  const double clock_mhz = st / sr * 100.0;                        // s_memrealtime ticks at 100 MHz
  const double waves_per_simd = (double)active / (nCU * (double)simds);
  const double ns = ms * 1e6 / (ninstr * waves_per_simd);          // wall ns per wave-instruction per populated SIMD
  printf("dispatch %3d  %-30s waves/SIMD=%d simds/CU=%d  active_waves=%ld (per-SIMD placement %ld/%ld/%ld/%ld)  wall=%.3f ms  "
         "clock=%.0f MHz  cycles/instr/SIMD: %.3f @measured clock, %.3f @2400 MHz  (longest wave %.3f ms)\n",
         disp, kname[KIND], wavesPerSimd, simds, active, persimd[0], persimd[1], persimd[2], persimd[3], ms, clock_mhz,
         ns * clock_mhz * 1e-3, ns * 2.4, tmax / (clock_mhz * 1e3));
}

template <int KIND>
void sweep(uint32_t* o, unsigned long long* t, int nCU) {
  run<KIND>(8, 4, o, t, nCU, false);     // warm-up dispatch (clock ramp, code load)
  for (int w : {1, 2, 4, 8}) run<KIND>(w, 4, o, t, nCU, true);
  for (int s : {1, 2, 3}) run<KIND>(8, s, o, t, nCU, true);
  for (int s : {1, 2, 3}) run<KIND>(2, s, o, t, nCU, true);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs=%d  nominal clock=%d kHz  instructions per wavefront per dispatch=%d\n", p.name, p.multiProcessorCount,
         p.clockRate, ITERS * 64);
  const int nCU = p.multiProcessorCount;
  uint32_t* d_out; unsigned long long* d_ticks;
  CHECK(hipMalloc(&d_out, (size_t)nCU * 2 * 1024 * 4));
  CHECK(hipMalloc(&d_ticks, (size_t)nCU * 32 * 3 * 8));
  sweep<FMA32>(d_out, d_ticks, nCU);
  sweep<PKFMA32>(d_out, d_ticks, nCU);
  sweep<ADD32>(d_out, d_ticks, nCU);
  sweep<MULLO>(d_out, d_ticks, nCU);
  sweep<MAD64>(d_out, d_ticks, nCU);
  sweep<MAD64_CARRYCHAIN>(d_out, d_ticks, nCU);
  return 0;
}
