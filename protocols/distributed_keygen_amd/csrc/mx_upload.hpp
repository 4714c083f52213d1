// Shared by the translation units of libmxpaillier.so: stream-ordered upload of small host operands
// and a few launch-planning constants.
#pragma once
#include "mx_host.hpp"
#include <cstring>
#include <algorithm>

using namespace mxh;

// ---- stream-ordered upload of small host operands --------------------------------------------
// Host operands (moduli, exponents, per-modulus constants) are a few KB.  Copying them with
// hipMemcpyAsync from pageable memory would either block the host behind all earlier work of the
// stream or leave the caller's buffer in use after return.  Instead they travel BY VALUE in the
// kernel-argument block of a one-block copy kernel: the runtime captures the arguments at launch,
// so the host buffer is free on return, nothing synchronises, and the copy is ordered in the stream.
namespace {
constexpr int UPLOAD_WORDS = 896;   // 3.5 KiB of the 4 KiB kernel-argument block
struct UploadChunk { uint32_t w[UPLOAD_WORDS]; };

__global__ void __launch_bounds__(256) upload_kernel(uint32_t* dst, UploadChunk c, int n) {
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = c.w[i];
}

int upload_words(void* d_dst, const uint32_t* h_src, size_t n, hipStream_t s) {
  // any size: one launch per 3.5 KiB chunk, never a synchronisation (operand sets of thousands of
  // moduli are better passed device-resident through the *_dev entry points)
  UploadChunk c;
  for (size_t off = 0; off < n; off += UPLOAD_WORDS) {
    int m = (int)std::min<size_t>(UPLOAD_WORDS, n - off);
    std::memcpy(c.w, h_src + off, (size_t)m * 4);
    hipLaunchKernelGGL(upload_kernel, dim3(1), dim3(256), 0, s, (uint32_t*)d_dst + off, c, m);
    MX_HIP(hipGetLastError());
  }
  return MX_OK;
}
#define MX_TRY(call) do { int rc__ = (call); if (rc__ != MX_OK) return rc__; } while (0)
}  // namespace


// ---- optional timing of the modexp kernels -------------------------------------------------------
// mx_profile(1) makes every mx_powmod_* call bracket ITS KERNEL (not the operand uploads before it)
// with two events on the caller's stream; mx_profile_collect waits for them and reports the sum.
// This is what bench.py's roofline.kernel_ms is, and what rocprofv3 --kernel-trace reports per launch.
#include <mutex>
#include <utility>
#include <vector>
struct MxProfile {
  std::mutex mu;
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};
extern MxProfile g_mx_profile;      // defined in mx_capi.hip
struct MxKernelTimer {
  hipStream_t s;
  hipEvent_t stop = nullptr;
  explicit MxKernelTimer(hipStream_t stream) : s(stream) {
    std::lock_guard<std::mutex> lock(g_mx_profile.mu);
    if (!g_mx_profile.on) return;
    hipEvent_t start;
    if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) { stop = nullptr; return; }
    hipEventRecord(start, s);
    g_mx_profile.events.emplace_back(start, stop);
  }
  ~MxKernelTimer() { if (stop) hipEventRecord(stop, s); }
};

namespace {
// ---- per-device state -----------------------------------------------------------------------------
// A process may drive several GPUs (Engine(device_index=1)): what the runtime keeps per device is cached per
// device here, keyed by hipGetDevice() at the time of the call.
constexpr int MX_MAX_DEVICES = 64;
inline int mx_current_device() {
  int dev = 0;
  return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MX_MAX_DEVICES) ? dev : -1;
}
// Opt a kernel into more than 64 KB of dynamic LDS per workgroup, once per (instance, device): the attribute belongs
// to the device's copy of the function.  `done` is a zero-initialised flag array of MX_MAX_DEVICES entries that the
// calling template instance owns.
inline hipError_t mx_allow_dynamic_lds(const void* kernel, int bytes, bool* done) {
  const int dev = mx_current_device();
  if (dev >= 0 && done[dev]) return hipSuccess;
  const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (err == hipSuccess && dev >= 0) done[dev] = true;
  return err;
}
// Compute units of the current device
inline int mx_device_cus() {
  static int cus[MX_MAX_DEVICES] = {};
  const int dev = mx_current_device();
  if (dev >= 0 && cus[dev] > 0) return cus[dev];
  int n = 0;
  if (dev < 0 || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0) cus[dev] = n;
  return n;
}

// Sizing queries only know the row width; assume the widest modulus that fits it.
inline int sizing_bits(int limbs) { return 32 * limbs < MAX_MOD_BITS ? 32 * limbs : MAX_MOD_BITS; }
constexpr int MAX_SLIDING_OPS = 16384;   // covers exponents up to 16384 bits
}  // namespace
