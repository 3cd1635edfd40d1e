// What can the memory system sustain for the fused backward's traffic pattern (4 fp32 tensors in, 2 out, 16.8 MB each,
// streamed once) on MI355X -- by launch shape and prefetch depth?  Standalone:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o tools/probes/bin/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// A: occupancy-driven, one float4 of each tensor per thread
__global__ __launch_bounds__(256) void k_simple(const float4* a, const float4* b, const float4* c, const float4* d, float4* o1,
                                                float4* o2, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 x = a[i], y = b[i], z = c[i], w = d[i];
  o1[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  o2[i] = make_float4(z.x * w.x, z.y * w.y, z.z * w.z, z.w * w.w);
}

// A2: the same with write-through (sc1) or non-temporal stores -- does the end-of-kernel L2 write-back shrink?
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f tov(float4 v) { v4f r = {v.x, v.y, v.z, v.w}; return r; }
__device__ __forceinline__ void st_sc1(float4* p, float4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(tov(v)) : "memory"); }
__device__ __forceinline__ void st_sc01(float4* p, float4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(tov(v)) : "memory"); }
__device__ __forceinline__ void st_nt(float4* p, float4 v) { asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(tov(v)) : "memory"); }
template <int MODE>
__global__ __launch_bounds__(256) void k_simple_st(const float4* a, const float4* b, const float4* c, const float4* d, float4* o1,
                                                   float4* o2, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 x = a[i], y = b[i], z = c[i], w = d[i];
  const float4 r1 = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w), r2 = make_float4(z.x * w.x, z.y * w.y, z.z * w.z, z.w * w.w);
  if (MODE == 1) st_sc1(o1 + i, r1), st_sc1(o2 + i, r2);
  if (MODE == 2) st_nt(o1 + i, r1), st_nt(o2 + i, r2);
  if (MODE == 3) st_sc01(o1 + i, r1), st_sc01(o2 + i, r2);
}
// A3: compiler builtins: non-temporal stores (MODE 0), + non-temporal loads (MODE 1)
template <int MODE>
__global__ __launch_bounds__(256) void k_simple_nt(const v4f* a, const v4f* b, const v4f* c, const v4f* d, v4f* o1, v4f* o2, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  v4f x, y, z, w;
  if (MODE == 1) {
    x = __builtin_nontemporal_load(a + i), y = __builtin_nontemporal_load(b + i), z = __builtin_nontemporal_load(c + i), w = __builtin_nontemporal_load(d + i);
  } else {
    x = a[i], y = b[i], z = c[i], w = d[i];
  }
  __builtin_nontemporal_store(x + y, o1 + i);
  __builtin_nontemporal_store(z * w, o2 + i);
}

// B: persistent blocks of 512 threads, units of 512 float4 dealt round-robin, DEPTH units of loads in flight (registers)
template <int DEPTH>
__global__ __launch_bounds__(512) void k_pipe(const float4* a, const float4* b, const float4* c, const float4* d, float4* o1,
                                              float4* o2, int nunits) {
  float4 x[DEPTH], y[DEPTH], z[DEPTH], w[DEPTH];
  const int nblk = gridDim.x, tid = threadIdx.x;
  const int nu = (nunits - (int)blockIdx.x + nblk - 1) / nblk;
  auto idx = [&](int k) { return ((long)(blockIdx.x + (long)min(k, nu - 1) * nblk)) * 512 + tid; };
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) {
    const long i = idx(s);
    x[s] = a[i], y[s] = b[i], z[s] = c[i], w[s] = d[i];
  }
  for (int k0 = 0; k0 < nu; k0 += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int k = k0 + s;
      const float4 xx = x[s], yy = y[s], zz = z[s], ww = w[s];
      const long j = idx(k + DEPTH);
      x[s] = a[j], y[s] = b[j], z[s] = c[j], w[s] = d[j];
      if (k < nu) {
        const long i = idx(k);
        o1[i] = make_float4(xx.x + yy.x, xx.y + yy.y, xx.z + yy.z, xx.w + yy.w);
        o2[i] = make_float4(zz.x * ww.x, zz.y * ww.y, zz.z * ww.z, zz.w * ww.w);
      }
    }
  }
}

// C: the same with an LDS ring filled by LDS-DMA (no VGPRs held by the loads)
template <int DEPTH>
__global__ __launch_bounds__(512) void k_dma(const float4* a, const float4* b, const float4* c, const float4* d, float4* o1,
                                             float4* o2, int nunits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* ring = (float4*)smem;  // [DEPTH][4][512]
  const int nblk = gridDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nu = (nunits - (int)blockIdx.x + nblk - 1) / nblk;
  auto issue = [&](int k) {
    const long i = ((long)(blockIdx.x + (long)min(k, nu - 1) * nblk)) * 512 + wv * 64 + lane;
    float4* slot = ring + (k % DEPTH) * 4 * 512 + wv * 64;
    __builtin_amdgcn_global_load_lds((glb_void*)(a + i), (lds_void*)(slot), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(b + i), (lds_void*)(slot + 512), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(c + i), (lds_void*)(slot + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(d + i), (lds_void*)(slot + 1536), 16, 0, 0);
  };
  for (int s = 0; s < DEPTH - 1; ++s) issue(s);
  for (int k = 0; k < nu; ++k) {
    issue(k + DEPTH - 1);
    // the 4 pieces of unit k are the oldest outstanding VMEM ops of this wave: everything issued after them
    // (4 * (DEPTH - 1) DMA pieces + the stores of the previous DEPTH-1 iterations) may stay in flight
    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    const float4* slot = ring + (k % DEPTH) * 4 * 512 + tid;
    const float4 xx = slot[0], yy = slot[512], zz = slot[1024], ww = slot[1536];
    const long i = ((long)(blockIdx.x + (long)k * nblk)) * 512 + tid;
    o1[i] = make_float4(xx.x + yy.x, xx.y + yy.y, xx.z + yy.z, xx.w + yy.w);
    o2[i] = make_float4(zz.x * ww.x, zz.y * ww.y, zz.z * ww.z, zz.w * ww.w);
  }
}

int main() {
  const long n = 8L * 128 * 128 * 8;  // float4 per tensor (B*H*W*32 floats)
  float4* t[6];
  // cycle through several buffer sets (> the 256 MiB Infinity Cache) so every launch streams from HBM
  const int NSET = 4;
  std::vector<float4*> sets;
  for (int s = 0; s < NSET * 6; ++s) {
    float4* p;
    hipMalloc(&p, n * 16);
    hipMemset(p, 0, n * 16);
    sets.push_back(p);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int nunits = (int)(n / 512);
  auto timeit = [&](const char* name, auto launch) {
    float best = 1e9, sum = 0;
    const int reps = 20;
    for (int it = 0; it < reps + 3; ++it) {
      for (int q = 0; q < 6; ++q) t[q] = sets[(it % NSET) * 6 + q];
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (it >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = 6.0 * n * 16;
    printf("%-34s mean %7.2f us  best %7.2f us   %6.2f TB/s (mean)\n", name, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12);
  };
  timeit("simple 256thr x n/256 blocks", [&] { hipLaunchKernelGGL(k_simple, dim3((n + 255) / 256), dim3(256), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], n); });
  timeit("simple, sc1 stores", [&] { hipLaunchKernelGGL(k_simple_st<1>, dim3((n + 255) / 256), dim3(256), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], n); });
  timeit("simple, nt stores", [&] { hipLaunchKernelGGL(k_simple_st<2>, dim3((n + 255) / 256), dim3(256), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], n); });
  timeit("simple, sc0 sc1 stores", [&] { hipLaunchKernelGGL(k_simple_st<3>, dim3((n + 255) / 256), dim3(256), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], n); });
  timeit("simple, builtin nt stores", [&] { hipLaunchKernelGGL(k_simple_nt<0>, dim3((n + 255) / 256), dim3(256), 0, 0, (v4f*)t[0], (v4f*)t[1], (v4f*)t[2], (v4f*)t[3], (v4f*)t[4], (v4f*)t[5], n); });
  timeit("simple, builtin nt loads+stores", [&] { hipLaunchKernelGGL(k_simple_nt<1>, dim3((n + 255) / 256), dim3(256), 0, 0, (v4f*)t[0], (v4f*)t[1], (v4f*)t[2], (v4f*)t[3], (v4f*)t[4], (v4f*)t[5], n); });
  timeit("2 x nt loads+stores (b2b)", [&] { hipLaunchKernelGGL(k_simple_nt<1>, dim3((n + 255) / 256), dim3(256), 0, 0, (v4f*)t[0], (v4f*)t[1], (v4f*)t[2], (v4f*)t[3], (v4f*)t[4], (v4f*)t[5], n);
                                            hipLaunchKernelGGL(k_simple_nt<1>, dim3((n + 255) / 256), dim3(256), 0, 0, (v4f*)t[4], (v4f*)t[5], (v4f*)t[2], (v4f*)t[3], (v4f*)t[0], (v4f*)t[1], n); });
  // pairs of dependent launches (the second reads what the first wrote): time per pair
  timeit("2 x simple plain (b2b)", [&] { hipLaunchKernelGGL(k_simple, dim3((n + 255) / 256), dim3(256), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], n);
                                         hipLaunchKernelGGL(k_simple, dim3((n + 255) / 256), dim3(256), 0, 0, t[4], t[5], t[2], t[3], t[0], t[1], n); });
  timeit("2 x simple sc1 (b2b)", [&] { hipLaunchKernelGGL(k_simple_st<1>, dim3((n + 255) / 256), dim3(256), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], n);
                                       hipLaunchKernelGGL(k_simple_st<1>, dim3((n + 255) / 256), dim3(256), 0, 0, t[4], t[5], t[2], t[3], t[0], t[1], n); });
  timeit("2 x simple nt (b2b)", [&] { hipLaunchKernelGGL(k_simple_st<2>, dim3((n + 255) / 256), dim3(256), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], n);
                                      hipLaunchKernelGGL(k_simple_st<2>, dim3((n + 255) / 256), dim3(256), 0, 0, t[4], t[5], t[2], t[3], t[0], t[1], n); });
  for (int blocks : {256}) {
    char nm[64];
    snprintf(nm, 64, "pipe depth1 %d blocks", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL(k_pipe<1>, dim3(blocks), dim3(512), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], nunits); });
    snprintf(nm, 64, "pipe depth2 %d blocks", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL(k_pipe<2>, dim3(blocks), dim3(512), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], nunits); });
    snprintf(nm, 64, "pipe depth3 %d blocks", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL(k_pipe<3>, dim3(blocks), dim3(512), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], nunits); });
    snprintf(nm, 64, "pipe depth4 %d blocks", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL(k_pipe<4>, dim3(blocks), dim3(512), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], nunits); });
  }
  hipFuncSetAttribute((const void*)k_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_dma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int blocks : {256}) {
    char nm[64];
    snprintf(nm, 64, "dma ring depth2 %d blocks", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL(k_dma<2>, dim3(blocks), dim3(512), 2 * 32768, 0, t[0], t[1], t[2], t[3], t[4], t[5], nunits); });
    snprintf(nm, 64, "dma ring depth3 %d blocks", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL(k_dma<3>, dim3(blocks), dim3(512), 3 * 32768, 0, t[0], t[1], t[2], t[3], t[4], t[5], nunits); });
    snprintf(nm, 64, "dma ring depth4 %d blocks", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL(k_dma<4>, dim3(blocks), dim3(512), 4 * 32768, 0, t[0], t[1], t[2], t[3], t[4], t[5], nunits); });
  }
  return 0;
}
