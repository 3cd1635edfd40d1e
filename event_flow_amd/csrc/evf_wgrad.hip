// Weight-gradient conv on bit-packed inputs (gfx950, fp32 MFMA):
//   dW[tau][ci][co] = sum_pix x[pix + tau][ci] * g[pix][co]
// GEMM view: M = ci (32), N = co (32), K = pixels (2 per v_mfma_f32_32x32x2_f32).
//
// "Tap per wave": a block owns WG_UNITS row segments (<= 128 pixels each) and
// 8 waves.  Wave w accumulates tap w over ALL the block's pixels, so each of
// the first 8 taps has exactly one accumulator per block and needs no
// cross-wave reduction; the ninth tap is shared (wave w takes every 8th pixel
// pair) and its 8 partial tiles are summed through LDS with plain stores.
// The B operand g[pix][co] is staged once per row segment in LDS (register
// staged, double buffered: the loads for the next segment are in flight while
// the current one feeds the MFMAs) and read by all 8 waves; the A operand is
// one bit of the spike word of pixel (pix + tau), fetched as an LDS broadcast.
// Per block the result is one slab [9][32][32] (written or accumulated).
#include "evf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define C32 32
#define WG_UNITS 4      // row segments per block
#define WG_CW 128       // pixels per row segment
#define WG_THREADS 512  // 8 waves, two per SIMD
#define WG_BW (WG_CW + 2)

__device__ __forceinline__ int wg_mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(WG_THREADS) void k_conv_wgrad_bits(const uint32_t* __restrict__ x,
                                                                const float* __restrict__ g, int B, int H, int W,
                                                                int nchunk, long nunits, int accumulate,
                                                                float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float s_g[2][WG_CW * C32];  // 2 x 16 KiB
  __shared__ uint32_t s_bits[2][3][WG_BW];
  // partial tiles of the shared ninth tap: 8 x 4 KiB, aliased onto s_g once the main loop is done
  float (*s_t8)[C32 * C32] = reinterpret_cast<float (*)[C32 * C32]>(&s_g[0][0]);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int dy = wv / 3, dx = wv % 3;  // tap of this wave (taps 0..7); tap 8 = (2,2) is shared

  f32x16 acc = {0}, acc8 = {0};
  float4 stage[2];  // register staging of the next segment's g rows: 128*32 floats / 512 threads = 2 float4

  auto unit_geom = [&](long u, int& b, int& y, int& x0, int& cw) {
    const long row = u / nchunk;
    const int c = (int)(u % nchunk);
    b = (int)(row / H);
    y = (int)(row % H);
    x0 = c * WG_CW;
    cw = min(WG_CW, W - x0);
  };
  auto issue_loads = [&](long u) {
    int b, y, x0, cw;
    unit_geom(u, b, y, x0, cw);
    const float4* src = (const float4*)(g + (((long)b * H + y) * W + x0) * C32);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + k * WG_THREADS;  // float4 index within the segment
      stage[k] = (e < cw * 8) ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](long u, int buf) {
    int b, y, x0, cw;
    unit_geom(u, b, y, x0, cw);
#pragma unroll
    for (int k = 0; k < 2; ++k) ((float4*)s_g[buf])[tid + k * WG_THREADS] = stage[k];
    for (int e = tid; e < 3 * WG_BW; e += WG_THREADS) {
      const int yy = y + e / WG_BW - 1, xx = x0 + e % WG_BW - 1;
      s_bits[buf][e / WG_BW][e % WG_BW] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[((long)b * H + yy) * W + xx] : 0u;
    }
  };

  const long u0 = (long)blockIdx.x * WG_UNITS;
  const int nu = (int)min((long)WG_UNITS, nunits - u0);
  if (nu > 0) {
    issue_loads(u0);
    commit(u0, 0);
  }
  __syncthreads();
  for (int k = 0; k < nu; ++k) {
    const int buf = k & 1;
    if (k + 1 < nu) issue_loads(u0 + k + 1);  // in flight during the MFMAs below
    const float* gb = s_g[buf] + lane;        // lane (j, h): g[pixel 2s + h][co = j] at (2s + h)*32 + j = s*64 + lane
    const uint32_t* wa = &s_bits[buf][dy][dx + h];  // word of pixel (2s + h) + tap
    const uint32_t* w8 = &s_bits[buf][2][2 + h];
#pragma unroll 1
    for (int s0 = 0; s0 < WG_CW / 2; s0 += 8) {
      float bv[8];
      uint32_t wd[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bv[q] = gb[(s0 + q) * 64];
        wd[q] = wa[2 * (s0 + q)];
      }
      const uint32_t wd8 = w8[2 * (s0 + wv)];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((float)((wd[q] >> i) & 1u), bv[q], acc, 0, 0, 0);
        if (q == wv) acc8 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)((wd8 >> i) & 1u), bv[q], acc8, 0, 0, 0);
      }
    }
    if (k + 1 < nu) commit(u0 + k + 1, buf ^ 1);
    __syncthreads();
  }

  // taps 0..7: one wave each, straight to the slab
  float* slab = partial + (long)blockIdx.x * (9 * C32 * C32);
  {
    float* d = slab + wv * (C32 * C32) + i;
    float prev[16];  // read together, selected afterwards (a load under `accumulate ? :` is its own round trip)
#pragma unroll
    for (int q = 0; q < 16; ++q) prev[q] = d[wg_mfma_row(q, lane) * C32];
#pragma unroll
    for (int q = 0; q < 16; ++q) d[wg_mfma_row(q, lane) * C32] = (accumulate ? prev[q] : 0.f) + acc[q];
  }
  // tap 8: sum the 8 partial tiles through LDS
#pragma unroll
  for (int q = 0; q < 16; ++q) s_t8[wv][wg_mfma_row(q, lane) * C32 + i] = acc8[q];
  __syncthreads();
  for (int e = tid; e < C32 * C32; e += WG_THREADS) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += s_t8[w][e];
    float* p = slab + 8 * (C32 * C32) + e;
    *p = accumulate ? *p + v : v;
  }
}

static long wg_units(int B, int H, int W) { return (long)B * H * ((W + WG_CW - 1) / WG_CW); }

extern "C" int evf_conv_wgrad_slabs(int B, int H, int W) { return evf_cdiv(wg_units(B, H, W), WG_UNITS); }

extern "C" int evf_conv_wgrad_bits(const uint32_t* x, const float* g_cur, int B, int H, int W, float* wg_partial,
                                   int accumulate, void* stream) {
  if (!x || !g_cur || !wg_partial || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  const long nunits = wg_units(B, H, W);
  hipLaunchKernelGGL(k_conv_wgrad_bits, dim3(evf_cdiv(nunits, WG_UNITS)), dim3(WG_THREADS), 0, EVF_STREAM(stream), x,
                     g_cur, B, H, W, (W + WG_CW - 1) / WG_CW, nunits, accumulate ? 1 : 0, wg_partial);
  return evf_status();
}
