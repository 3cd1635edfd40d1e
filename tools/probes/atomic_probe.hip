// Micro-benchmark: throughput of random global atomics by scope / type on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/atomic_probe.hip -o gpurun_out/atomic_probe
// Each thread adds to `per` pseudo-random words of a [nsamp][plane] buffer; blocks are tied to a sample either
// arbitrarily (sample = block / blocks_per_sample) or by XCD (sample = XCC_ID, so every adder of a line sits
// behind the same L2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

template <int SCOPE, bool INT>
__global__ void k_atomics(float* buf, int plane, int per, int by_xcd, int bps, uint32_t* xcc_seen) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int samp = by_xcd ? (int)xcc_id() : blockIdx.x / bps;
  if (threadIdx.x == 0 && xcc_seen) xcc_seen[blockIdx.x] = xcc_id();
  uint32_t h = tid * 2654435761u + 12345u;
  float* base = buf + (long)samp * plane;
  for (int k = 0; k < per; ++k) {
    h = h * 1664525u + 1013904223u;
    const int idx = (h >> 8) % plane;
    if (INT)
      __hip_atomic_fetch_add((unsigned*)base + idx, 1u, __ATOMIC_RELAXED, SCOPE);
    else
      __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, SCOPE);
  }
}

template <int SCOPE, bool INT>
float run(float* buf, int plane, int nsamp, int blocks, int per, int by_xcd, uint32_t* seen) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 5; ++it) {
    hipMemset(buf, 0, sizeof(float) * (size_t)plane * nsamp);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_atomics<SCOPE, INT>), dim3(blocks), dim3(256), 0, 0, buf, plane, per, by_xcd, blocks / nsamp, seen);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int plane = 2 * 128 * 128, nsamp = 8;
  float* buf;
  hipMalloc(&buf, sizeof(float) * (size_t)plane * nsamp);
  uint32_t* seen;
  hipMalloc(&seen, 4 * 4096);
  for (int blocks : {256, 512, 2048}) {
    for (int per : {2, 16, 256}) {
      const double n = (double)blocks * 256 * per;
      struct R { const char* name; float ms; } rs[] = {
          {"agent f32", run<__HIP_MEMORY_SCOPE_AGENT, false>(buf, plane, nsamp, blocks, per, 0, seen)},
          {"agent u32", run<__HIP_MEMORY_SCOPE_AGENT, true>(buf, plane, nsamp, blocks, per, 0, seen)},
          {"wg f32", run<__HIP_MEMORY_SCOPE_WORKGROUP, false>(buf, plane, nsamp, blocks, per, 0, seen)},
          {"wg u32", run<__HIP_MEMORY_SCOPE_WORKGROUP, true>(buf, plane, nsamp, blocks, per, 0, seen)},
          {"wg f32 byxcd", run<__HIP_MEMORY_SCOPE_WORKGROUP, false>(buf, plane, nsamp, blocks, per, 1, seen)},
          {"wg u32 byxcd", run<__HIP_MEMORY_SCOPE_WORKGROUP, true>(buf, plane, nsamp, blocks, per, 1, seen)},
          {"agent f32 byxcd", run<__HIP_MEMORY_SCOPE_AGENT, false>(buf, plane, nsamp, blocks, per, 1, seen)},
      };
      for (auto& r : rs) printf("blocks %5d per %4d  %-16s %8.2f us  %7.1f G atomics/s\n", blocks, per, r.name, r.ms * 1e3, n / (r.ms * 1e-3) / 1e9);
    }
  }
  // correctness of the XCD-tied workgroup-scope form: every add must be counted
  {
    const int blocks = 2048, per = 64;
    hipMemset(buf, 0, sizeof(float) * (size_t)plane * nsamp);
    hipLaunchKernelGGL((k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP, true>), dim3(blocks), dim3(256), 0, 0, buf, plane, per, 1, blocks / nsamp, seen);
    hipDeviceSynchronize();
    std::vector<uint32_t> h((size_t)plane * nsamp), s(blocks);
    hipMemcpy(h.data(), buf, 4 * h.size(), hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), seen, 4 * blocks, hipMemcpyDeviceToHost);
    unsigned long long tot = 0;
    for (auto v : h) tot += v;
    int hist[16] = {0};
    int modmatch = 0;
    for (int b = 0; b < blocks; ++b) hist[s[b] & 15]++, modmatch += ((s[b] & 15) == (uint32_t)(b % 8));
    printf("byxcd wg u32: counted %llu of %llu adds; blocks per XCC:", tot, (unsigned long long)blocks * 256 * per);
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
    printf("; block %% 8 == xcc for %d of %d blocks\n", modmatch, blocks);
    // NOT tied to the XCD: workgroup-scope adds from different L2s to the same words lose updates?
    hipMemset(buf, 0, sizeof(float) * (size_t)plane * nsamp);
    hipLaunchKernelGGL((k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP, true>), dim3(blocks), dim3(256), 0, 0, buf, plane, per, 0, blocks / nsamp, seen);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), buf, 4 * h.size(), hipMemcpyDeviceToHost);
    tot = 0;
    for (auto v : h) tot += v;
    printf("untied wg u32: counted %llu of %llu adds\n", tot, (unsigned long long)blocks * 256 * per);
  }
  return 0;
}
