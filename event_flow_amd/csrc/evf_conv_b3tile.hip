// 3x3 stride-1 convolution (forward and input gradient) of wide, high-resolution layers: spatially tiled, the halo tile of
// the input staged ONCE per 16-channel group in LDS as bf16 split planes -- read from HBM once, converted once, reused by
// the nine taps and by every output channel of the block -- against k_conv2d_b3 (evf_conv_b3gen.hip), whose waves re-read
// and re-convert their pixels per tap and whose LDS pipe carries one weight fragment per MFMA.  Same arithmetic: weights
// w = hi + mid + lo (three bf16 planes, the packed operand of evf_pack_conv2d_weight_b3), activations as bf16 head +
// residual planes, v_mfma_f32_32x32x16_bf16 with fp32 accumulation; a block-uniform vote per channel group (are all
// residuals of the staged tile zero?) picks 3 products (spikes, counts, bilinear blends of spikes) or the 6 terms above
// 2^-24 of the leading one.  The vote never changes a result.
//
//   block      512 threads = 8 waves, output tile 16 rows x 32 columns, 32*NT output channels; wave w owns rows 2w, 2w+1
//              (two 32-pixel M tiles sharing every weight fragment: 2 + 3 NT LDS fragment reads per 6 NT MFMAs)
//   LDS        3 planes x (18 x 34 halo pixels) x 48 B (16 channels x bf16 + 16 B pad: conflict-free b128 reads)  86 KiB
//              9 taps x NT x 3 planes x 1 KiB weight fragments                                                    27 NT KiB
//   pipeline   the next group's halo floats and weight fragments are fetched into registers before the matrix phase and
//              written to LDS after it (two barriers per group)
//   epilogue   weights are the A operand: the tile comes out transposed, a lane owns one pixel and 4 x 4 consecutive
//              channels per N tile -> float4 stores (bias, accumulate)
#include "evf_common.h"
#include "evf_split.h"

typedef float t_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 t_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t t_u32x4 __attribute__((ext_vector_type(4)));
typedef float t_f32x4 __attribute__((ext_vector_type(4)));

#define T_ROWS 16
#define T_COLS 32
#define T_HR (T_ROWS + 2)
#define T_HC (T_COLS + 2)
#define T_PIX (T_HR * T_HC)          // 612 halo pixels
#define T_PSTRIDE 48                 // bytes per halo pixel and plane
#define T_PLANE (T_PIX * T_PSTRIDE)  // 29376
#define T_ATASKS (T_PIX * 4)         // float4 loads per group
#define T_AITER ((T_ATASKS + 511) / 512)
#define B3_STAGE (4 * 3 * 64)  // uint4 per (N tile, tap, 64-channel group) of the packed weights: [chunk 4][term 3][lane 64]

struct TileGeo {
  int B, H, W, K, N;  // image (input = output size), contraction channels, output channels
  int lds, ldo;       // pixel strides (floats)
  int flip;           // 0 forward (tap (dy,dx) reads pixel (+dy-1,+dx-1)), 1 input gradient (reads (+1-dy,+1-dx))
  int tiles_y, tiles_x;
  int nt_off;         // first 32-channel N tile of this launch (a wide N with a short remainder: 64-wide blocks + one 32-wide launch)
};

// 512 threads on a CU that holds ONE block of this kernel (LDS) = two waves per SIMD: 256 registers each.  Without the attribute
// the allocator settled on 128 and spilled 64 bytes (NT = 2); with it the LIF-EV-FlowNet input gradients run 4 % faster.
#ifndef B3T_WAVES_ATTR
#define B3T_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#ifdef B3T_UNROLL_OX  // (A/B: the three taps of a kernel row unrolled, so that the next tap's fragment reads may issue under this tap's MFMAs)
#define B3T_OX_LOOP _Pragma("unroll")
#else
#define B3T_OX_LOOP _Pragma("unroll 1")
#endif
template <int NT>
__global__ __launch_bounds__(512) B3T_WAVES_ATTR void k_conv3_b3t(const float* __restrict__ src, const uint4* __restrict__ wp,
                                                   const float* __restrict__ bias, float* __restrict__ out, TileGeo g,
                                                   int accumulate, int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_a = smem;                                // [3 planes][612 px][48 B]
  uint4* s_w = (uint4*)(smem + 3 * T_PLANE);       // [NT][9 taps][3 terms][64 lanes]
  constexpr int WFRAG = NT * 27 * 64;              // uint4 per group
  constexpr int WITER = (WFRAG + 511) / 512;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, kg = lane >> 5;
  // XCD-aware tile order: the blocks one XCD receives (blockIdx.x % 8) are spatial neighbours -> halo rows hit its L2
  const int ntile = g.B * g.tiles_y * g.tiles_x, per = (ntile + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tile >= ntile) return;
  const int txi = tile % g.tiles_x, t1 = tile / g.tiles_x, tyi = t1 % g.tiles_y, b = t1 / g.tiles_y;
  const int y0 = tyi * T_ROWS, x0 = txi * T_COLS;
  const int G64 = (g.K + 63) >> 6, KC = (g.K + 15) >> 4, ntiles = (g.N + 31) >> 5;
  // this block's N tiles (a tile past the end re-reads the last one; never stored)
  const long wtile = (long)(9 * G64) * B3_STAGE;
  const int nt_base = g.nt_off + (int)blockIdx.y * NT;

  // ---- staging: global -> registers (before the matrix phase) -> LDS (after it)
  t_f32x4 pa[T_AITER];
  t_u32x4 pw[WITER];
  const float* img = src + (long)b * g.H * g.W * g.lds;
  auto fetch = [&](int kc) {
#pragma unroll
    for (int i = 0; i < T_AITER; ++i) {
      const int task = min(tid + 512 * i, T_ATASKS - 1), px = task >> 2, q = task & 3;
      const int hy = px / T_HC, hx = px - hy * T_HC;
      const int sy = min(max(y0 + hy - 1, 0), g.H - 1), sx = min(max(x0 + hx - 1, 0), g.W - 1);
      const int c = kc * 16 + 4 * q;
      pa[i] = *(const t_f32x4*)(img + ((long)sy * g.W + sx) * g.lds + (c + 4 <= g.K ? c : 0));
    }
    const int gg = kc >> 2, ch = kc & 3;
#pragma unroll
    for (int i = 0; i < WITER; ++i) {
      const int idx = min(tid + 512 * i, WFRAG - 1), ln = idx & 63, f = idx >> 6;
      const int term = f % 3, f2 = f / 3, tap = f2 % 9, t = f2 / 9;
      pw[i] = ((const t_u32x4*)wp)[min(nt_base + t, ntiles - 1) * wtile + (((long)tap * G64 + gg) * 4 + ch) * 192 + term * 64 + ln];
    }
  };
  // returns "some residual is not zero" for this thread's elements
  auto commit = [&](int kc) -> int {
    uint32_t nz = 0u;
#pragma unroll
    for (int i = 0; i < T_AITER; ++i) {
      const int task = tid + 512 * i, px = task >> 2, q = task & 3;
      const int hy = px / T_HC, hx = px - hy * T_HC;
      const int sy = y0 + hy - 1, sx = x0 + hx - 1;
      const bool ok = sy >= 0 && sy < g.H && sx >= 0 && sx < g.W && kc * 16 + 4 * q + 4 <= g.K;
      const t_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const t_f32x4 v = ok ? pa[i] : zero4;
      uint32_t h0, m0, l0, h1, m1, l1;
      evf_split3_pair(v.x, v.y, h0, m0, l0);
      evf_split3_pair(v.z, v.w, h1, m1, l1);
      nz |= m0 | m1;  // (mid = bf16(residual): zero iff the residual is zero; -0 cannot arise from x - head(x))
      if (task < T_ATASKS) {
        char* p = s_a + px * T_PSTRIDE + q * 8;
        *(uint2*)(p) = make_uint2(h0, h1);
        *(uint2*)(p + T_PLANE) = make_uint2(m0, m1);
        *(uint2*)(p + 2 * T_PLANE) = make_uint2(l0, l1);
      }
    }
#pragma unroll
    for (int i = 0; i < WITER; ++i) {
      const int idx = tid + 512 * i;
      if (idx < WFRAG) ((t_u32x4*)s_w)[idx] = pw[i];
    }
    return (nz & 0x7FFF7FFFu) != 0u;
  };

  t_f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  // split-K: blockIdx.z owns the channel groups [kc_lo, kc_hi) and writes its partial sums to its own slab
  int kc_lo = 0, kc_hi = KC;
  if (ksplit > 1) {
    const int per = (KC + ksplit - 1) / ksplit;
    kc_lo = min((int)blockIdx.z * per, KC - 1), kc_hi = min(kc_lo + per, KC);
    if ((int)blockIdx.z * per >= KC) kc_hi = kc_lo;  // (an empty split still writes its zeros)
    out += (long)blockIdx.z * g.B * g.H * g.W * g.ldo;
  }
  fetch(kc_lo);
  int inexact = __syncthreads_or(commit(kc_lo));

#pragma unroll 1
  for (int kc = kc_lo; kc < kc_hi; ++kc) {
    fetch(min(kc + 1, kc_hi - 1));
    // ---- matrix phase: 9 taps x (2 M tiles x NT N tiles) x 3 | 6 products
    const char* arow = s_a + ((2 * wv) * T_HC + col) * T_PSTRIDE + kg * 16;
    if (!inexact) {
#pragma unroll 1
      for (int oy = 0; oy < 3; ++oy) {
        B3T_OX_LOOP
        for (int ox = 0; ox < 3; ++ox) {
          const int wtap = g.flip ? (2 - oy) * 3 + (2 - ox) : oy * 3 + ox;
          const char* ap = arow + (oy * T_HC + ox) * T_PSTRIDE;
          const uint4* wq = s_w + wtap * 192 + lane;
          const uint4 x0q = *(const uint4*)ap, x1q = *(const uint4*)(ap + T_HC * T_PSTRIDE);
          const t_bf16x8 xa = *(const t_bf16x8*)&x0q, xb = *(const t_bf16x8*)&x1q;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const uint4 q0 = wq[t * 27 * 64], q1 = wq[t * 27 * 64 + 64], q2 = wq[t * 27 * 64 + 128];
            const t_bf16x8 wh = *(const t_bf16x8*)&q0, wm = *(const t_bf16x8*)&q1, wl = *(const t_bf16x8*)&q2;
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xa, acc[0][t], 0, 0, 0);  // smallest terms first
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xb, acc[1][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xa, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xb, acc[1][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xa, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xb, acc[1][t], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll 1
      for (int oy = 0; oy < 3; ++oy) {
        B3T_OX_LOOP
        for (int ox = 0; ox < 3; ++ox) {
          const int wtap = g.flip ? (2 - oy) * 3 + (2 - ox) : oy * 3 + ox;
          const char* ap = arow + (oy * T_HC + ox) * T_PSTRIDE;
          const uint4* wq = s_w + wtap * 192 + lane;
          t_bf16x8 xh[2], xm[2], xl[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const uint4 a0 = *(const uint4*)(ap + m * T_HC * T_PSTRIDE), a1 = *(const uint4*)(ap + m * T_HC * T_PSTRIDE + T_PLANE),
                        a2 = *(const uint4*)(ap + m * T_HC * T_PSTRIDE + 2 * T_PLANE);
            xh[m] = *(const t_bf16x8*)&a0, xm[m] = *(const t_bf16x8*)&a1, xl[m] = *(const t_bf16x8*)&a2;
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const uint4 q0 = wq[t * 27 * 64], q1 = wq[t * 27 * 64 + 64], q2 = wq[t * 27 * 64 + 128];
            const t_bf16x8 wh = *(const t_bf16x8*)&q0, wm = *(const t_bf16x8*)&q1, wl = *(const t_bf16x8*)&q2;
#pragma unroll
            for (int m = 0; m < 2; ++m) {  // smallest terms first
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
    __syncthreads();  // every wave is done with this group's planes
    if (kc + 1 < kc_hi) inexact = __syncthreads_or(commit(kc + 1));
  }

  // ---- epilogue: lane = pixel (row 2wv + m, column col); channels n0 + 8q + 4kg + e
  const bool vec = (g.ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0;  // uniform
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int oy = y0 + 2 * wv + m, oxx = x0 + col;
    const bool mok = oy < g.H && oxx < g.W;
    float* orow = out + (((long)b * g.H + min(oy, g.H - 1)) * g.W + min(oxx, g.W - 1)) * g.ldo;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n0 = (nt_base + t) * 32 + 4 * kg;
      float4 oldv[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 8 * q, nq = min(n, max(g.N - 4, 0));
        const bool full = n + 4 <= g.N;
        if (vec) {
          const float4 o = *(const float4*)(orow + nq);
          oldv[q] = (accumulate && full) ? o : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = orow[min(n + e, g.N - 1)];
          oldv[q] = accumulate ? make_float4(o[0], o[1], o[2], o[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float bb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float bl = (bias ? bias : out)[bias ? min(n + e, g.N - 1) : 0];
          bb[e] = (bias && n + e < g.N) ? bl : 0.f;
        }
        bv[q] = make_float4(bb[0], bb[1], bb[2], bb[3]);
      }
      if (mok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + 8 * q;
          float ov[4] = {oldv[q].x, oldv[q].y, oldv[q].z, oldv[q].w};
          if (accumulate && vec && n + 4 > g.N) {  // ragged last quad: the clamped float4 above is not this quad
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = n + e < g.N ? orow[n + e] : 0.f;
          }
          const float v[4] = {(acc[m][t][4 * q + 0] + bv[q].x) + ov[0], (acc[m][t][4 * q + 1] + bv[q].y) + ov[1],
                              (acc[m][t][4 * q + 2] + bv[q].z) + ov[2], (acc[m][t][4 * q + 3] + bv[q].w) + ov[3]};
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
}

// Is the tiled kernel the better choice for this 3x3 stride-1 product, and with how many K splits?  0 = no (the caller
// falls back to k_conv2d_b3), 1 = yes, unsplit, n > 1 = yes with n slabs (max_split = slabs the caller's scratch holds).
int evf_conv3_b3t_plan(const float* src, int B, int H, int W, int K, int N, int lds, bool force, int max_split, int force_split) {
  if (K % 4 != 0 || lds % 4 != 0 || (((uintptr_t)src) & 15) != 0) return 0;  // float4 halo loads
  const long tiles = (long)B * evf_cdiv(H, T_ROWS) * evf_cdiv(W, T_COLS);
  const long blocks = tiles * evf_cdiv(N, N > 32 ? 64 : 32);
  const int KC = evf_cdiv(K, 16);
  const int smax = max(1, min(max_split, KC / 4));  // at least 4 channel groups per split
  int ks = blocks >= 256 ? 1 : (int)min((long)smax, evf_cdiv(512L, blocks));  // (less than one block per CU: split)
  if (force_split > 0) ks = max(1, min(min(force_split, max(max_split, 1)), KC));
  if (force) return ks;
  // enough blocks for the 256 CUs, and tiles that are mostly inside the image
  const double fill = (double)H * W / ((double)evf_cdiv(H, T_ROWS) * T_ROWS * evf_cdiv(W, T_COLS) * T_COLS);
  return (blocks * ks >= 192 && fill >= 0.7) ? ks : 0;
}

int evf_conv3_b3t_launch(const float* src, int lds, const void* wp, const float* bias, float* out, int ldo, int B, int H, int W,
                         int K, int N, int flip, int accumulate, int ksplit, hipStream_t st) {
  TileGeo g;
  g.B = B, g.H = H, g.W = W, g.K = K, g.N = N, g.lds = lds, g.ldo = ldo, g.flip = flip;
  g.tiles_y = evf_cdiv(H, T_ROWS), g.tiles_x = evf_cdiv(W, T_COLS);
  const int ntile = B * g.tiles_y * g.tiles_x, gx = 8 * evf_cdiv(ntile, 8);
  g.nt_off = 0;
  // N = 64 q + r with 0 < r <= 32 (a decoder's input gradient: 130 / 258 / 514 channels + alignment): q blocks of 64 channels, and
  // the remainder as ONE launch of 32-channel blocks instead of a 64-channel block that is 94 % padding (LIF-EV-FlowNet, 132
  // channels at 256 x 256: 3 x 64 -> 2 x 64 + 32)
  const int rem = N % 64;
  const bool tail32 = N > 64 && rem > 0 && rem <= 32;
  static bool once2 = false, once1 = false;
  const size_t smem2 = 3 * T_PLANE + 2 * 27 * 1024, smem1 = 3 * T_PLANE + 27 * 1024;
  if (!once2) {
    (void)hipFuncSetAttribute((const void*)k_conv3_b3t<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
    once2 = true;
  }
  if (!once1) {
    (void)hipFuncSetAttribute((const void*)k_conv3_b3t<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
    once1 = true;
  }
  if (tail32) {
    hipLaunchKernelGGL((k_conv3_b3t<2>), dim3(gx, N / 64, ksplit), dim3(512), smem2, st, src, (const uint4*)wp, bias, out, g, accumulate,
                       ksplit);
    g.nt_off = 2 * (N / 64);
    hipLaunchKernelGGL((k_conv3_b3t<1>), dim3(gx, 1, ksplit), dim3(512), smem1, st, src, (const uint4*)wp, bias, out, g, accumulate,
                       ksplit);
    return evf_status();
  }
  if (N > 32) {
    const size_t smem = smem2;
    hipLaunchKernelGGL((k_conv3_b3t<2>), dim3(gx, evf_cdiv(N, 64), ksplit), dim3(512), smem, st, src, (const uint4*)wp, bias, out,
                       g, accumulate, ksplit);
  } else {
    hipLaunchKernelGGL((k_conv3_b3t<1>), dim3(gx, 1, ksplit), dim3(512), smem1, st, src, (const uint4*)wp, bias, out, g, accumulate,
                       ksplit);
  }
  return evf_status();
}
