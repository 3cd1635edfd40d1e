// 3x3 stride-1 convolution (forward and input gradient) of LOW-resolution, many-channel layers: images of at most 16 x 16 pixels
// (the 512-channel layers of the spiking EV-FlowNet at 256 x 256 input: encoder 4's recurrent block and the two residual
// blocks, reference models/unet.py:418-465, models/spiking_submodules.py:878-975).
//
// Neither of the other two kernels fits them.  k_conv3_b3t (evf_conv_b3tile.hip) tiles 16 rows x 32 columns: a 16 x 16 image
// fills half a tile, its plan refuses them.  k_conv2d_b3 (evf_conv_b3gen.hip), which took them, is a GATHER kernel: every wave
// re-reads and re-splits its pixels per tap and reads one weight fragment per MFMA from LDS -- 88 us for the real-valued
// (six-term) input gradient of a 512 -> 512 layer at B = 8, 0.26 of the dense bf16 peak issued (tools/debug/c4_entry_times.py).
//
// Here ONE block owns ONE image x 64 output channels (x a share of the contraction channels, split-K):
//   block     I_WAVES = 8 waves, two per SIMD; wave w owns image rows 2w, 2w + 1 = ONE M tile of 2 rows x 16 pixels, and both
//             32-channel N tiles: every activation fragment feeds 2 N tiles (9 LDS fragment reads per 12 MFMAs in the six-term
//             form; I_WAVES = 4: two M tiles per wave, 12 reads per 24 MFMAs, but one wave per SIMD with nobody to hide its
//             fragment reads and its share of the split -- measured slower, see evf_conv3_b3i_plan);
//   LDS       3 planes x (18 x 18 halo pixels) x 48 B (16 channels x bf16 + 16 B pad: conflict-free b128 reads)   46 KiB
//             2 x (2 N tiles x 9 taps x 3 planes x 1 KiB) weight fragments, DOUBLE buffered (LDS-DMA, no VGPRs)   108 KiB
//   pipeline  group g + 1 (16 contraction channels): its weight fragments arrive by global_load_lds into the other buffer and
//             its halo floats into registers UNDER the matrix phase of group g; behind the phase: barrier, exact 3-way bf16
//             split of the halo into the planes, barrier.  216 MFMAs per wave and group.
// Same arithmetic as the two other kernels: weights w = hi + mid + lo (the packed operand of evf_pack_conv2d_weight_b3),
// activations split exactly on the fly, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, a block-uniform vote per group picks
// 3 products (all residuals of the staged planes zero) or the 6 terms above 2^-24 of the leading one, smallest terms first.
// Split-K partial sums go to slabs the caller reduces in index order (k_b3_reduce): deterministic.
#include "evf_common.h"
#include "evf_split.h"
#include <stdlib.h>

typedef float i_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 i_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t i_u32x4 __attribute__((ext_vector_type(4)));
typedef float i_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void i_lds_void;
typedef __attribute__((address_space(1))) const void i_glb_void;

#define I_DIM 16                      // image rows / columns a block covers (smaller images: masked)
#define I_HD (I_DIM + 2)              // halo edge
#define I_PIX (I_HD * I_HD)           // 324 halo pixels
#define I_PSTRIDE 48                  // bytes per halo pixel and plane
#define I_PLANE (I_PIX * I_PSTRIDE)   // 15552
#define I_ATASKS (I_PIX * 4)          // float4 loads per group
#ifndef I_WAVES
#define I_WAVES 8                     // 8: a wave owns ONE M tile (2 image rows), two waves per SIMD hide each other's fragment reads and
#endif                                //    the split; 4: two M tiles per wave (every weight fragment read once for both), one wave per SIMD
#define I_THREADS (64 * I_WAVES)
#define I_MT (8 / I_WAVES)            // M tiles (pairs of image rows) per wave
#define I_AITER ((I_ATASKS + I_THREADS - 1) / I_THREADS)
#define I_NT 2                        // 32-channel N tiles per block
#define I_WFRAG (I_NT * 27)           // 1 KiB weight fragments per group
#define I_WBUF (I_WFRAG * 1024)       // bytes per weight buffer
#define I_LDS (3 * I_PLANE + 2 * I_WBUF)  // (the register form uses one weight buffer; the allocation stays: one block per CU either way)
#ifndef I_UNROLL_TAPS
#define I_UNROLL_TAPS 1  // the nine taps unrolled: the next tap's fragment reads issue under this tap's MFMAs (one wave per SIMD: nobody else hides them)
#endif
#if I_UNROLL_TAPS
#define I_TAP_UNROLL _Pragma("unroll")
#else
#define I_TAP_UNROLL _Pragma("unroll 1")
#endif
#define I_STAGE (4 * 3 * 64)          // uint4 per (N tile, tap, 64-channel group) of the packed weights: [chunk 4][term 3][lane 64]

struct ImgGeo {
  int B, H, W, K, N;  // image (input = output size, H, W <= 16), contraction channels, output channels
  int lds, ldo;       // pixel strides (floats)
  int flip;           // 0 forward (tap (dy,dx) reads pixel (+dy-1,+dx-1)), 1 input gradient (reads (+1-dy,+1-dx))
};

__global__ __launch_bounds__(I_THREADS) void k_conv3_b3i(const float* __restrict__ src, const uint4* __restrict__ wp,
                                                   const float* __restrict__ bias, float* __restrict__ out, ImgGeo g, int accumulate,
                                                   int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_a = smem;                 // [3 planes][324 px][48 B]
  char* s_w = smem + 3 * I_PLANE;   // [2 buffers][N tile][9 taps][3 terms][64 lanes] uint4
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kg = lane >> 5;
  const int mr = (lane & 31) >> 4, mc = lane & 15;  // this lane's pixel inside an M tile: row mr of the pair, column mc
  const int b = blockIdx.x, nt_base = (int)blockIdx.y * I_NT;
  const int G64 = (g.K + 63) >> 6, KC = (g.K + 15) >> 4, ntiles = (g.N + 31) >> 5;
  const long wtile = (long)(9 * G64) * I_STAGE;
  const float* img = src + (long)b * g.H * g.W * g.lds;

  // split-K: blockIdx.z owns the channel groups [kc_lo, kc_hi) and writes its partial sums to its own slab
  int kc_lo = 0, kc_hi = KC;
  if (ksplit > 1) {
    const int per = (KC + ksplit - 1) / ksplit;
    kc_lo = min((int)blockIdx.z * per, KC - 1), kc_hi = min(kc_lo + per, KC);
    if ((int)blockIdx.z * per >= KC) kc_hi = kc_lo;  // (an empty split still writes its zeros)
    out += (long)blockIdx.z * g.B * g.H * g.W * g.ldo;
  }

  // ---- staging.  Halo floats: global -> registers (before the matrix phase) -> split -> LDS planes (behind it).
  i_f32x4 pa[I_AITER];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int i = 0; i < I_AITER; ++i) {
      const int task = min(tid + I_THREADS * i, I_ATASKS - 1), px = task >> 2, q = task & 3;
      const int hy = px / I_HD, hx = px - hy * I_HD;
      const int sy = min(max(hy - 1, 0), g.H - 1), sx = min(max(hx - 1, 0), g.W - 1);
      const int c = kc * 16 + 4 * q;
      pa[i] = *(const i_f32x4*)(img + ((long)sy * g.W + sx) * g.lds + (c + 4 <= g.K ? c : 0));
    }
  };
  // weight fragments of group kc: 54 pieces of 1 KiB by LDS-DMA (lane-contiguous in the packed operand), 13-14 per wave
  auto dma_w = [&](int kc, int buf) {
    const int gg = kc >> 2, ch = kc & 3;
    for (int f = wv; f < I_WFRAG; f += I_WAVES) {
      const int term = f % 3, f2 = f / 3, tap = f2 % 9, t = f2 / 9;
      const uint4* srcw = wp + min(nt_base + t, ntiles - 1) * wtile + (((long)tap * G64 + gg) * 4 + ch) * 192 + term * 64 + lane;
      __builtin_amdgcn_global_load_lds((i_glb_void*)srcw, (i_lds_void*)(s_w + buf * I_WBUF + f * 1024), 16, 0, 0);
    }
  };
#ifndef I_WDMA
#define I_WDMA 1  // 1: the weight fragments by LDS-DMA into the other of two buffers (the matrix waves issue the pieces themselves); 0: through
#endif            //    registers like the halo -- requested before the matrix phase, written to LDS behind it into ONE buffer: measured
                  //    110-114 against 64-65 us per 512 -> 512 input gradient (the write sits exposed between two barriers, 198 registers)
  constexpr int WITER = (I_WFRAG * 64 + I_THREADS - 1) / I_THREADS;  // uint4 per thread and group (register form)
  uint4 pw[I_WDMA ? 1 : WITER];
  auto fetch_w = [&](int kc) {
    const int gg = kc >> 2, ch = kc & 3;
#pragma unroll
    for (int i = 0; i < (I_WDMA ? 0 : WITER); ++i) {
      const int idx = min(tid + I_THREADS * i, I_WFRAG * 64 - 1), ln = idx & 63, f = idx >> 6;
      const int term = f % 3, f2 = f / 3, tap = f2 % 9, t = f2 / 9;
      pw[i] = wp[min(nt_base + t, ntiles - 1) * wtile + (((long)tap * G64 + gg) * 4 + ch) * 192 + term * 64 + ln];
    }
  };
  auto commit_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < (I_WDMA ? 0 : WITER); ++i) {
      const int idx = tid + I_THREADS * i;
      if (idx < I_WFRAG * 64) ((uint4*)(s_w + buf * I_WBUF))[idx] = pw[i];
    }
  };
  auto commit = [&](int kc) -> int {  // returns "some residual is not zero" for this thread's elements
    uint32_t nz = 0u;
#pragma unroll
    for (int i = 0; i < I_AITER; ++i) {
      const int task = tid + I_THREADS * i, px = task >> 2, q = task & 3;
      const int hy = px / I_HD, hx = px - hy * I_HD;
      const int sy = hy - 1, sx = hx - 1;
      const bool ok = sy >= 0 && sy < g.H && sx >= 0 && sx < g.W && kc * 16 + 4 * q + 4 <= g.K;
      const i_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const i_f32x4 v = ok ? pa[i] : zero4;
      uint32_t h0, m0, l0, h1, m1, l1;
      evf_split3_pair(v.x, v.y, h0, m0, l0);
      evf_split3_pair(v.z, v.w, h1, m1, l1);
      nz |= m0 | m1;  // (mid = bf16(residual): zero iff the residual is zero)
      if (task < I_ATASKS) {
        char* p = s_a + px * I_PSTRIDE + q * 8;
        *(uint2*)(p) = make_uint2(h0, h1);
        *(uint2*)(p + I_PLANE) = make_uint2(m0, m1);
        *(uint2*)(p + 2 * I_PLANE) = make_uint2(l0, l1);
      }
    }
    return (nz & 0x7FFF7FFFu) != 0u;
  };

  i_f32x16 acc[I_MT][I_NT];
#pragma unroll
  for (int m = 0; m < I_MT; ++m)
#pragma unroll
    for (int t = 0; t < I_NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  int inexact = 0;
  if (kc_hi > kc_lo) {
    if (I_WDMA) dma_w(kc_lo, 0);
    else fetch_w(kc_lo);
    fetch(kc_lo);
    const int nzv = commit(kc_lo);
    if (!I_WDMA) commit_w(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the weight DMA of this wave has landed)
    inexact = __syncthreads_or(nzv);
  }
#pragma unroll 1
  for (int kc = kc_lo; kc < kc_hi; ++kc) {
    const int buf = I_WDMA ? (kc - kc_lo) & 1 : 0;
    const bool more = kc + 1 < kc_hi;
    if (I_WDMA) {
      if (more) dma_w(kc + 1, buf ^ 1);  // (the other buffer was last read in group kc - 1: every wave is past that barrier)
    } else {
      fetch_w(min(kc + 1, kc_hi - 1));
    }
    fetch(min(kc + 1, kc_hi - 1));
    // ---- matrix phase: 9 taps x (2 M tiles x 2 N tiles) x 3 | 6 products
    // M tile m of this wave = image rows 4 wv + 2 m, + 1; the lane's pixel is (row 4 wv + 2 m + mr, column mc)
    const char* arow = s_a + ((2 * I_MT * wv + mr) * I_HD + mc) * I_PSTRIDE + kg * 16;
    const uint4* wbuf = (const uint4*)(s_w + buf * I_WBUF) + lane;
    if (!inexact) {
      I_TAP_UNROLL
      for (int oy = 0; oy < 3; ++oy) {
        I_TAP_UNROLL
        for (int ox = 0; ox < 3; ++ox) {
          const int wtap = g.flip ? (2 - oy) * 3 + (2 - ox) : oy * 3 + ox;
          const char* ap = arow + (oy * I_HD + ox) * I_PSTRIDE;
          const uint4* wq = wbuf + wtap * 192;
          i_bf16x8 xs[I_MT];
#pragma unroll
          for (int m = 0; m < I_MT; ++m) {
            const uint4 xq = *(const uint4*)(ap + m * 2 * I_HD * I_PSTRIDE);
            xs[m] = *(const i_bf16x8*)&xq;
          }
#pragma unroll
          for (int t = 0; t < I_NT; ++t) {
            const uint4 q0 = wq[t * 27 * 64], q1 = wq[t * 27 * 64 + 64], q2 = wq[t * 27 * 64 + 128];
            const i_bf16x8 wh = *(const i_bf16x8*)&q0, wm = *(const i_bf16x8*)&q1, wl = *(const i_bf16x8*)&q2;
#pragma unroll
            for (int m = 0; m < I_MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xs[m], acc[m][t], 0, 0, 0);  // smallest terms first
#pragma unroll
            for (int m = 0; m < I_MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xs[m], acc[m][t], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < I_MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xs[m], acc[m][t], 0, 0, 0);
          }
        }
      }
    } else {
      I_TAP_UNROLL
      for (int oy = 0; oy < 3; ++oy) {
        I_TAP_UNROLL
        for (int ox = 0; ox < 3; ++ox) {
          const int wtap = g.flip ? (2 - oy) * 3 + (2 - ox) : oy * 3 + ox;
          const char* ap = arow + (oy * I_HD + ox) * I_PSTRIDE;
          const uint4* wq = wbuf + wtap * 192;
          i_bf16x8 xh[I_MT], xm[I_MT], xl[I_MT];
#pragma unroll
          for (int m = 0; m < I_MT; ++m) {
            const char* am = ap + m * 2 * I_HD * I_PSTRIDE;
            const uint4 a0 = *(const uint4*)am, a1 = *(const uint4*)(am + I_PLANE), a2 = *(const uint4*)(am + 2 * I_PLANE);
            xh[m] = *(const i_bf16x8*)&a0, xm[m] = *(const i_bf16x8*)&a1, xl[m] = *(const i_bf16x8*)&a2;
          }
#pragma unroll
          for (int t = 0; t < I_NT; ++t) {
            const uint4 q0 = wq[t * 27 * 64], q1 = wq[t * 27 * 64 + 64], q2 = wq[t * 27 * 64 + 128];
            const i_bf16x8 wh = *(const i_bf16x8*)&q0, wm = *(const i_bf16x8*)&q1, wl = *(const i_bf16x8*)&q2;
#pragma unroll
            for (int m = 0; m < I_MT; ++m) {  // smallest terms first
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm[m], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[m], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[m], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh[m], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm[m], acc[m][t], 0, 0, 0);
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[m], acc[m][t], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();  // every wave is done with this group's planes (and with weight buffer `buf`)
    if (more) {
      if (!I_WDMA) commit_w(0);
      const int nzv = commit(kc + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's pieces of the next weight buffer have landed)
      inexact = __syncthreads_or(nzv);
    }
  }

  // ---- epilogue: weights are the A operand, the tile comes out transposed: lane = pixel (row 4 wv + 2 m + mr, column mc),
  // registers 4 q + e = channel n0 + 8 q + 4 kg + e of each N tile -> float4 stores (bias, accumulate)
  const bool vec = (g.ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0;  // uniform
#pragma unroll
  for (int m = 0; m < I_MT; ++m) {
    const int oy = 2 * I_MT * wv + 2 * m + mr, oxx = mc;
    const bool mok = oy < g.H && oxx < g.W;
    float* orow = out + (((long)b * g.H + min(oy, g.H - 1)) * g.W + min(oxx, g.W - 1)) * g.ldo;
#pragma unroll
    for (int t = 0; t < I_NT; ++t) {
      const int n0 = (nt_base + t) * 32 + 4 * kg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 8 * q;
        if (!mok || n >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[m][t][4 * q + e];
          if (bias && n + e < g.N) x += bias[n + e];
          if (accumulate && n + e < g.N) x += orow[n + e];
          v[e] = x;
        }
        if (vec && n + 4 <= g.N) {
          *(float4*)(orow + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < g.N) orow[n + e] = v[e];
        }
      }
    }
  }
}

// Is this kernel the choice for the product, and with how many K splits?  0 = no, 1 = yes unsplit, n > 1 = yes with n slabs
// (max_split = slabs the caller's scratch holds).  Images of at most 16 x 16 pixels with enough channels that the staging pays.
int evf_conv3_b3i_plan(const float* src, int B, int H, int W, int K, int N, int lds, bool force, int max_split, int force_split) {
  static const bool on = !(getenv("EVF_CONV_IMG") && getenv("EVF_CONV_IMG")[0] == '0');
  if (!on || H > I_DIM || W > I_DIM || H < 1 || W < 1) return 0;
  if (K % 4 != 0 || lds % 4 != 0 || (((uintptr_t)src) & 15) != 0) return 0;  // float4 halo loads
  if (!force && (K < 64 || N < 32 || H < 4 || W < 4)) return 0;  // (few channels / pixels: the gather kernel's launch is as good)
  const int KC = evf_cdiv(K, 16);
  const long blocks = (long)B * evf_cdiv(N, 32 * I_NT);
  const int smax = max(1, min(max_split, KC / 4));  // at least 4 channel groups per split
  int ks = blocks >= 256 ? 1 : (int)min((long)smax, evf_cdiv(256L, blocks));
  if (force_split > 0) ks = max(1, min(min(force_split, max(max_split, 1)), KC));
  if (force) return ks;
  return (blocks * ks >= 96) ? ks : 0;  // (too few blocks even when split: the gather kernel spreads over pixels as well)
}

int evf_conv3_b3i_launch(const float* src, int lds, const void* wp, const float* bias, float* out, int ldo, int B, int H, int W,
                         int K, int N, int flip, int accumulate, int ksplit, hipStream_t st) {
  ImgGeo g;
  g.B = B, g.H = H, g.W = W, g.K = K, g.N = N, g.lds = lds, g.ldo = ldo, g.flip = flip;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)k_conv3_b3i, hipFuncAttributeMaxDynamicSharedMemorySize, (int)I_LDS);
    once = true;
  }
  hipLaunchKernelGGL(k_conv3_b3i, dim3(B, evf_cdiv(N, 32 * I_NT), ksplit), dim3(I_THREADS), I_LDS, st, src, (const uint4*)wp, bias, out, g,
                     accumulate, ksplit);
  return evf_status();
}
