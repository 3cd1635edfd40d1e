// Pure WRITE streams on MI355X: what a kernel that only writes (the window forward's tape) can reach.  One float4 per lane and
// store, rows of 128 B per 8 lanes like team E's potential lines; plain / nt / sc1 / sc0 sc1 stores; persistent grids of 256 ..
// 2048 blocks over 1 GiB (rotating over two buffers: nothing stays in the Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/write_probe.hip -o /tmp/write_probe && /tmp/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k_write(float4* dst, size_t n4) {
  const v4f v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    v4f* p = (v4f*)(dst + i);
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_copy(const float4* src, float4* dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const v4f v = *(const v4f*)(src + i);
    if (MODE == 0) *(v4f*)(dst + i) = v;
    else __builtin_nontemporal_store(v, (v4f*)(dst + i));
  }
}
int main() {
  const size_t bytes = 1ull << 30, n4 = bytes / 16;
  float4 *a, *b, *c;
  hipMalloc(&a, bytes), hipMalloc(&b, bytes), hipMalloc(&c, bytes);
  hipMemset(c, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const char* names[4] = {"plain", "nt", "sc1", "sc0 sc1"};
  for (int grid : {256, 512, 1024, 2048, 8192})
    for (int mode = 0; mode < 4; ++mode) {
      auto go = [&](float4* d) {
        if (mode == 0) hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, 0, d, n4);
        if (mode == 1) hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, d, n4);
        if (mode == 2) hipLaunchKernelGGL(k_write<2>, dim3(grid), dim3(256), 0, 0, d, n4);
        if (mode == 3) hipLaunchKernelGGL(k_write<3>, dim3(grid), dim3(256), 0, 0, d, n4);
      };
      go(a), go(b);
      hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) go(r & 1 ? a : b);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("write %-8s grid %5d: %7.1f us / GiB  %.2f TB/s\n", names[mode], grid, ms * 100.f, bytes * 10 / (ms * 1e-3) / 1e12);
    }
  for (int grid : {256, 1024, 8192})
    for (int mode = 0; mode < 2; ++mode) {
      auto go = [&](float4* d) {
        if (mode == 0) hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, c, d, n4);
        else hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, c, d, n4);
      };
      go(a), go(b);
      hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) go(r & 1 ? a : b);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("copy  %-8s grid %5d: %7.1f us / GiB  %.2f TB/s (r + w)\n", names[mode], grid, ms * 100.f, 2.0 * bytes * 10 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
