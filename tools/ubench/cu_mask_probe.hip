// Where do the workgroups of a stream created with hipExtStreamCreateWithCUMask land on MI355X (8 XCDs x 32 CUs)?
// For a few masks: launches 4096 one-wavefront workgroups that record (XCC id, SE id, CU id) and idle for a while
// (so that all CUs of the mask are needed), and prints how many distinct CUs per XCD received work.
// Build: hipcc -O3 --offload-arch=gfx950 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void where(uint32_t* out, unsigned long long ticks) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  const int n = 4096;
  uint32_t* d; CHECK(hipMalloc(&d, n * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  where<<<n, 64, 0, s>>>(d, 2000);      // 20 us per workgroup
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint32_t> h(n); CHECK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  std::set<uint32_t> cus[16];
  for (uint32_t v : h) { uint32_t xcc = v >> 16, hw = v & 0xFFFF; cus[xcc].insert((hw >> 8) & 0xFF); }   // CU_ID 11:8, SH 12, SE 15:13
  int bits = 0; for (uint32_t w : mask) bits += __builtin_popcount(w);
  printf("%-34s bits=%3d  %.2f ms  distinct (SE,SH,CU) per XCC:", name, bits, ms);
  for (int x = 0; x < 8; ++x) printf(" %zu", cus[x].size());
  printf("\n");
  CHECK(hipFree(d)); CHECK(hipStreamDestroy(s));
}

int main() {
  std::vector<uint32_t> all(8, 0xFFFFFFFFu);
  run("all 256 bits", all);
  { std::vector<uint32_t> m(8, 0); m[0] = 0xFFFFFFFFu; run("bits 0..31", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = m[1] = 0xFFFFFFFFu; run("bits 0..63", m); }
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); run("bits = 0 mod 8", m); }
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if ((i % 8) < 2) m[i / 32] |= 1u << (i % 32); run("bits = 0,1 mod 8", m); }
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if ((i % 8) >= 6) m[i / 32] |= 1u << (i % 32); run("bits = 6,7 mod 8", m); }
  { std::vector<uint32_t> m(8, 0); m[7] = 0xFFFFFFFFu; run("bits 224..255", m); }
  return 0;
}
