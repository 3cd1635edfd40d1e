// Micro-benchmark: throughput of random LDS atomics by type on MI355X (what bounds k_cm_splat_lds / k_iwe_*_lds).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/lds_atomic_probe.hip -o gpurun_out/lds_atomic_probe
// 256 blocks x 1024 threads; every thread adds to `per` pseudo-random slots of a 32 K-word (f32 / u32) or 16 K-slot (u64) LDS image.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND>  // 0 f32, 1 u32, 2 u64, 3 f32 through a CAS-free "returning" add
__global__ __launch_bounds__(1024) void k_lds(int per, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* f = (float*)smem;
  unsigned* u = (unsigned*)smem;
  unsigned long long* q = (unsigned long long*)smem;
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) u[i] = 0u;
  __syncthreads();
  uint32_t h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  for (int k = 0; k < per; ++k) {
    h = h * 1664525u + 1013904223u;
    const int idx = (h >> 8) & 32767;
    if (KIND == 0) atomicAdd(f + idx, 1.0f);
    if (KIND == 1) atomicAdd(u + idx, 1u);
    if (KIND == 2) atomicAdd(q + (idx & 16383), 1ull);
    if (KIND == 3) acc += atomicAdd(f + idx, 1.0f);
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = f[0] + acc;
}

template <int KIND>
static void run(const char* name, int per, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k_lds<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_lds<KIND>, dim3(256), dim3(1024), 131072, 0, per, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double n = 256.0 * 1024.0 * per;
  printf("%-12s %8.1f us  %7.1f G atomics/s  %6.3f per CU and ns\n", name, best * 1e3, n / (best * 1e-3) / 1e9, n / 256.0 / (best * 1e6));
}

int main() {
  float* sink;
  hipMalloc(&sink, 4096);
  const int per = 256;
  run<0>("f32", per, sink);
  run<1>("u32", per, sink);
  run<2>("u64", per, sink);
  run<3>("f32 rtn", per, sink);
  return 0;
}
