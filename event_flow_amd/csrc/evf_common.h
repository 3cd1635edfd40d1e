// Shared helpers for the gfx950 kernels of libevflow_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/evflow.h"

#define EVF_STREAM(s) ((hipStream_t)(s))

static inline int evf_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? EVF_OK : -(1000 + (int)e);
}
static inline int evf_hip(hipError_t e) { return e == hipSuccess ? EVF_OK : -(1000 + (int)e); }

static inline int evf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hardware fp32 atomic add (global_atomic_add_f32 / ds_add_f32); plain
// atomicAdd would lower to a CAS loop without -munsafe-fp-atomics.
__device__ __forceinline__ void evf_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }

// wave64 sum, result valid in every lane
__device__ __forceinline__ float evf_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of up to 1024 threads; result valid in thread 0
__device__ __forceinline__ float evf_block_sum(float v, float* smem /* >= 16 floats */) {
  v = evf_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < 64) {
    const int nw = (blockDim.x + 63) >> 6;
    r = (lane < nw) ? smem[lane] : 0.f;
    r = evf_wave_sum(r);
  }
  __syncthreads();
  return r;
}

// evf_dgrad_ws.hip: wave-specialised input-gradient kernel behind evf_conv_dgrad_b3_f32[_pair]
int evf_dgrad_ws_launch(const float* g_cur, const void* wT_b3, float* g_x, int accumulate, int B, int H, int W, const float* g_P,
                        const uint32_t* x_bits, const void* wT2_b3, float* g_x2, int max_blocks, void* stream);
