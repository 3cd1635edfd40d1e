// Weight (and bias) gradient of the general convolution on the fp32 matrix cores:
//   gw[co][ci][tap] = sum_px x[px (+) tap][ci] * g[px][co]      (autograd of nn.Conv2d w.r.t. weight)
// a GEMM whose contraction runs over pixels.  For v_mfma_f32_32x32x2_f32 a lane
// holds ONE k value, so [pixel][channel] tiles in LDS are already in fragment
// order: A = x^T (32 ci x 2 px), B = g (2 px x 32 co), no transposes.
//
// 3x3 (k_wgrad9): one block (8 waves) owns a (32 CT ci) x (32 NT co) weight tile
// for ALL nine taps and walks over 32-pixel tiles (R rows x TW cols) of its
// pixel range.  Per tile it stages the g tile and ONE x halo tile in LDS and
// every wave applies its own tap (wave w = tap w, the ninth tap is shared):
// g is read once instead of nine times and x once (plus halo) instead of nine
// times -- the first version (one tap per block) was bound by those re-reads.
// Partial sums of the pixel splits go to slabs [split][tap][ci][co] (coalesced
// stores, no atomics) and are summed by k_wgrad_reduce into the torch layout.
//
// 1x1, 5x5, 7x7 (k_conv2d_wgrad_f32): one tap per block (blockIdx.z), 4 waves split the pixel pairs, split-K + atomics.
#include <stdlib.h>
#include <string.h>

#include "evf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int cg_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
static inline int cg_out_dim(int n, int ksz, int stride) { return (n + 2 * (ksz >> 1) - ksz) / stride + 1; }

// ---------------------------------------------------------------------------
// 1x1: one tap, the 4 waves split the pixel pairs, split-K + atomics
// ---------------------------------------------------------------------------
struct WgGeo {
  int B, H, W, Cin, OH, OW, Cout, ksz, stride, ldx, ldg, cin_total, cin_off;
  int stages;  // 32-pixel stages per K split
};

template <int CT, int NT, int VEC, int VG>
__global__ __launch_bounds__(256) void k_conv2d_wgrad_f32(const float* __restrict__ x, const float* __restrict__ gy,
                                                          float* __restrict__ gw, float* __restrict__ gbias, WgGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s = (float*)smem_raw;
  constexpr int XW = 32 * CT, GW = 32 * NT, STG = 32 * (XW + GW);  // floats per stage
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane & 31, kg = lane >> 5;
  const int n_ct = (g.Cin + XW - 1) / XW;
  const int cit = blockIdx.y % n_ct, cot = blockIdx.y / n_ct;
  const int ci0 = cit * XW, co0 = cot * GW;
  const int tap = blockIdx.z, dy = tap / g.ksz, dx = tap - dy * g.ksz, pad = g.ksz >> 1;
  const long M = (long)g.B * g.OH * g.OW;
  const long m_begin = (long)blockIdx.x * g.stages * 32;
  const bool do_bias = gbias && cit == 0 && tap == 0;

  f32x16 acc[CT][NT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
  float bsum[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bsum[t] = 0.f;

  // tile loads: x tile 32 px x XW ch = 8*CT float4 per pixel -> CT float4 per thread; g likewise.
  // The pixel a thread loads advances by 32 per stage: its (b, oy, ox) is tracked incrementally
  // (one 32-bit division at kernel start instead of three 64-bit ones per load per stage).
  float4 xr[CT], gr[NT];
  int x_ox[CT], x_oy[CT], x_b[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int px = (tid + 256 * i) / (8 * CT);
    const long m = m_begin + px;
    const int mi = (int)(m < M ? m : M - 1);
    x_ox[i] = mi % g.OW;
    const int t1 = mi / g.OW;
    x_oy[i] = t1 % g.OH, x_b[i] = t1 / g.OH;
    if (m >= M) x_b[i] = g.B;  // past the end: stays invalid
  }
  auto load_tiles = [&](int st) {
    const long m0 = m_begin + (long)st * 32;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int idx = tid + 256 * i, px = idx / (8 * CT), q = idx - px * (8 * CT);
      const bool mok = x_b[i] < g.B;
      const int oy = x_oy[i], ox = x_ox[i], b = mok ? x_b[i] : g.B - 1;
      int sy = oy * g.stride + dy - pad, sx = ox * g.stride + dx - pad;
      const bool ok = mok && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W;
      sy = min(max(sy, 0), g.H - 1), sx = min(max(sx, 0), g.W - 1);
      const float* p = x + (((long)b * g.H + sy) * g.W + sx) * g.ldx;
      const int c = ci0 + 4 * q;
      float4 v;
      if (VEC == 4) {
        const bool cv = c + 4 <= g.Cin;
        v = *(const float4*)(p + (cv ? c : 0));
        if (!(ok && cv)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = p[c + j < g.Cin ? c + j : 0];
          e[j] = (ok && c + j < g.Cin) ? t : 0.f;
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
      }
      xr[i] = v;
      // advance this load's pixel by one stage
      x_ox[i] += 32;
      while (x_ox[i] >= g.OW) x_ox[i] -= g.OW, ++x_oy[i];
      while (x_oy[i] >= g.OH) x_oy[i] -= g.OH, ++x_b[i];
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int idx = tid + 256 * i, px = idx / (8 * NT), q = idx - px * (8 * NT);
      const long m = m0 + px;
      const bool mok = m < M;
      const float* p = gy + (mok ? m : M - 1) * g.ldg;
      const int c = co0 + 4 * q;
      float4 v;
      if (VG == 4) {
        const bool cv = c + 4 <= g.Cout;
        v = *(const float4*)(p + (cv ? c : 0));
        if (!(mok && cv)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = p[c + j < g.Cout ? c + j : 0];
          e[j] = (mok && c + j < g.Cout) ? t : 0.f;
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
      }
      gr[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    float* sx = s + buf * STG;
    float* sg = sx + 32 * XW;
#pragma unroll
    for (int i = 0; i < CT; ++i) *(float4*)(sx + (tid + 256 * i) * 4) = xr[i];
#pragma unroll
    for (int i = 0; i < NT; ++i) *(float4*)(sg + (tid + 256 * i) * 4) = gr[i];
  };

  load_tiles(0);
  store_tiles(0);
  __syncthreads();
#pragma unroll 1
  for (int st = 0; st < g.stages; ++st) {
    load_tiles(st + 1);  // past the split's last stage this prefetch is unused (and may read a neighbour split's pixels)
    const float* sx = s + (st & 1) * STG;
    const float* sg = sx + 32 * XW;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int p = 2 * (wv + 4 * jj) + kg;
      float av[CT], bv[NT];
#pragma unroll
      for (int c = 0; c < CT; ++c) av[c] = sx[p * XW + c * 32 + row];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bv[t] = sg[p * GW + t * 32 + row];
        bsum[t] += bv[t];
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[t], acc[c][t], 0, 0, 0);
    }
    store_tiles((st + 1) & 1);
    __syncthreads();
  }

  // cross-wave reduction in LDS, then one atomic per output element
  float* red = s;  // [4 waves][CT*NT][32 rows][32 cols]
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wv * CT * NT + c * NT + t) * 32 + cg_row(r, lane)) * 32 + row] = acc[c][t][r];
  __syncthreads();
  const int T = g.ksz * g.ksz;
  for (int e = tid; e < CT * NT * 1024; e += 256) {
    const int sub = e >> 10, ci_l = (e >> 5) & 31, co_l = e & 31;
    const int c = sub / NT, t = sub - c * NT;
    float v = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) v += red[(w4 * CT * NT + sub) * 1024 + (e & 1023)];
    const int ci = ci0 + c * 32 + ci_l, co = co0 + t * 32 + co_l;
    if (ci < g.Cin && co < g.Cout && g.cin_off + ci < g.cin_total)
      evf_atomic_add(gw + ((long)co * g.cin_total + g.cin_off + ci) * T + tap, v);
  }
  if (do_bias) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float v = bsum[t];
      v += __shfl_xor(v, 32, 64);
      const int co = co0 + t * 32 + row;
      if (kg == 0 && co < g.Cout) evf_atomic_add(gbias + co, v);
    }
  }
}

template <int CT, int NT>
static void wg_launch(const float* x, const float* gy, float* gw, float* gbias, const WgGeo& g, int ksplit, bool vec4,
                      hipStream_t st) {
  const int n_ct = evf_cdiv(g.Cin, 32 * CT), n_nt = evf_cdiv(g.Cout, 32 * NT);
  dim3 grid(ksplit, n_ct * n_nt, g.ksz * g.ksz), block(256);
  const size_t stage = 2 * 32 * (32 * CT + 32 * NT) * sizeof(float), red = 4 * CT * NT * 1024 * sizeof(float);
  const size_t smem = stage > red ? stage : red;
  const bool vx = g.Cin % 4 == 0 && g.ldx % 4 == 0 && ((uintptr_t)x & 15) == 0;
  const bool vg = g.Cout % 4 == 0 && g.ldg % 4 == 0 && ((uintptr_t)gy & 15) == 0;
  (void)vec4;
  if (vx && vg)
    hipLaunchKernelGGL((k_conv2d_wgrad_f32<CT, NT, 4, 4>), grid, block, smem, st, x, gy, gw, gbias, g);
  else if (vx)  // e.g. the 32 -> 2 prediction heads
    hipLaunchKernelGGL((k_conv2d_wgrad_f32<CT, NT, 4, 1>), grid, block, smem, st, x, gy, gw, gbias, g);
  else if (vg)
    hipLaunchKernelGGL((k_conv2d_wgrad_f32<CT, NT, 1, 4>), grid, block, smem, st, x, gy, gw, gbias, g);
  else
    hipLaunchKernelGGL((k_conv2d_wgrad_f32<CT, NT, 1, 1>), grid, block, smem, st, x, gy, gw, gbias, g);
}


// ---------------------------------------------------------------------------
// 1x1 with a handful of output channels (the 32 -> 2 flow prediction heads):
// a streaming reduction, not a matrix product.  Each thread owns one 4-channel
// group of x for every PPB-th pixel of its block's pixel range and keeps
// COUT x 4 sums in registers; blocks write partials to the workspace and
// k_wgrad1_reduce sums them (no same-address atomics: 1024 blocks hammering 64
// addresses was what bound the matrix-core version of this layer).
// ---------------------------------------------------------------------------
constexpr int WG1_BLOCKS = 512;

template <int COUT>
__global__ __launch_bounds__(256) void k_wgrad1_small(const float* __restrict__ x, const float* __restrict__ gy,
                                                      float* __restrict__ part, long M, int Cin, int ldx, int ldg,
                                                      int px_per_block) {
  __shared__ float red[256 * (COUT * 4 + 1)];
  const int tid = threadIdx.x, Q = Cin >> 2, q = tid % Q, pp = tid / Q, PPB = 256 / Q;
  const long m0 = (long)blockIdx.x * px_per_block;
  const long m1 = m0 + px_per_block < M ? m0 + px_per_block : M;
  float acc[COUT][4];
  float bs[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) {
    bs[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
  }
  constexpr int U = 4;
  for (long mb = m0 + pp; mb < m1; mb += (long)U * PPB) {
    float4 xv[U];
    float gv[U][COUT];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // clamped addresses + select: no load sits under a branch
      const long m = mb + (long)u * PPB;
      const long mc = m < m1 ? m : M - 1;
      xv[u] = *(const float4*)(x + mc * ldx + 4 * q);
#pragma unroll
      for (int c = 0; c < COUT; ++c) gv[u][c] = gy[mc * ldg + c];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = mb + (long)u * PPB < m1;
#pragma unroll
      for (int c = 0; c < COUT; ++c) {
        const float gg = ok ? gv[u][c] : 0.f;
        bs[c] += gg;
        acc[c][0] += gg * xv[u].x, acc[c][1] += gg * xv[u].y, acc[c][2] += gg * xv[u].z, acc[c][3] += gg * xv[u].w;
      }
    }
  }
  constexpr int RW = COUT * 4 + 1;  // odd row stride: conflict-free column reads
#pragma unroll
  for (int c = 0; c < COUT; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[tid * RW + c * 4 + j] = acc[c][j];
  __syncthreads();
  float* out = part + (long)blockIdx.x * (COUT * Cin + COUT);
  for (int e = tid; e < COUT * Cin; e += 256) {
    const int co = e / Cin, ci = e - co * Cin, qq = ci >> 2, j = ci & 3;
    float v = 0.f;
    for (int k = 0; k < PPB; ++k) v += red[(k * Q + qq) * RW + co * 4 + j];
    out[e] = v;
  }
  __syncthreads();
  if (q == 0)
#pragma unroll
    for (int c = 0; c < COUT; ++c) red[pp * COUT + c] = bs[c];
  __syncthreads();
  if (tid < COUT) {
    float v = 0.f;
    for (int k = 0; k < PPB; ++k) v += red[k * COUT + tid];
    out[COUT * Cin + tid] = v;
  }
}

// gw[co][cin_off + ci] (+)= sum_blocks part[blk][co*Cin + ci];  gbias[co] += sum_blocks part[blk][Cout*Cin + co]
__global__ void k_wgrad1_reduce(const float* __restrict__ part, int nblk, int Cin, int Cout, int cin_total, int cin_off,
                                int accumulate, float* __restrict__ gw, float* __restrict__ gbias) {
  const int per = Cout * Cin + Cout, e = blockIdx.x, lane = threadIdx.x;  // one wave per output element
  float s = 0.f;
  for (int k = lane; k < nblk; k += 64) s += part[(long)k * per + e];
#pragma unroll
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane) return;
  if (e >= Cout * Cin) {
    if (gbias) gbias[e - Cout * Cin] += s;  // zeroed by the caller when not accumulating
    return;
  }
  const int co = e / Cin, ci = e - co * Cin;
  if (cin_off + ci >= cin_total) return;
  float* d = gw + (long)co * cin_total + cin_off + ci;
  *d = accumulate ? *d + s : s;
}

// ---------------------------------------------------------------------------
// The whole 1x1 layer with a handful of outputs in one pass each way: y = act(W x + b) (models/submodules.py:52-61, the flow
// predictions of every scale, models/unet.py:355-369 of the reference) and its backward -- g_pre = g_y * act'(y), g_x = W^T g_pre,
// partial sums of g_W / g_b -- instead of general conv + activation kernels forward and nchw_to_nhwc + activation backward +
// streaming weight gradient + general input-gradient conv backward (LIF-EV-FlowNet: 4 scales, 0.11 + 0.17 ms per step for a
// layer that moves 67 MB at the finest scale).  A thread owns one 4-channel group q of every PPB-th pixel (as k_wgrad1_small).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float h1_act(int kind, float v) {
  if (kind == 1) return tanhf(v);
  if (kind == 2) return 1.0f / (1.0f + expf(-v));
  if (kind == 3) return fmaxf(v, 0.f);
  return v;
}
template <int COUT>
__global__ __launch_bounds__(256) void k_head1_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                   int act, long M, int Cin, int ldx, float* __restrict__ y, int ldy) {
  const int tid = threadIdx.x, Q = Cin >> 2, q = tid % Q, pp = tid / Q, PPB = 256 / Q;
  float4 wr[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) wr[c] = *(const float4*)(w + (long)c * Cin + 4 * q);
  constexpr int U = 4;
  for (long mb = (long)blockIdx.x * PPB * U + pp; mb - pp < M; mb += (long)gridDim.x * PPB * U) {  // (uniform trip count)
    float4 xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = mb + (long)u * PPB;
      xv[u] = *(const float4*)(x + (m < M ? m : M - 1) * ldx + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = mb + (long)u * PPB;
      float s[COUT];
#pragma unroll
      for (int c = 0; c < COUT; ++c) s[c] = (wr[c].x * xv[u].x + wr[c].y * xv[u].y) + (wr[c].z * xv[u].z + wr[c].w * xv[u].w);
      // the Q lanes of a pixel are consecutive lanes (Q a power of two <= 64) or whole waves (Q > 64: LDS)
      if (Q <= 64) {
        for (int o = 1; o < Q; o <<= 1)
#pragma unroll
          for (int c = 0; c < COUT; ++c) s[c] += __shfl_xor(s[c], o, 64);
        if (q == 0 && m < M)
#pragma unroll
          for (int c = 0; c < COUT; ++c) y[m * ldy + c] = h1_act(act, s[c] + (bias ? bias[c] : 0.f));
      }
    }
  }
}

template <int COUT>
__global__ __launch_bounds__(256) void k_head1_bwd(const float* __restrict__ x, const float* __restrict__ yo, const float* __restrict__ gy,
                                                   long gy_plane, const float* __restrict__ w, int act, long M, long HW, int Cin, int ldx,
                                                   int ldy, float* __restrict__ gx, int ldgx, float* __restrict__ part,
                                                   int px_per_block) {
  __shared__ float red[256 * (COUT * 4 + 1)];
  const int tid = threadIdx.x, Q = Cin >> 2, q = tid % Q, pp = tid / Q, PPB = 256 / Q;
  const long m0 = (long)blockIdx.x * px_per_block;
  const long m1 = m0 + px_per_block < M ? m0 + px_per_block : M;
  float4 wr[COUT];
  float acc[COUT][4], bs[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) {
    wr[c] = *(const float4*)(w + (long)c * Cin + 4 * q);
    bs[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
  }
  constexpr int U = 4;
  for (long mb = m0 + pp; mb < m1; mb += (long)U * PPB) {
    float4 xv[U];
    float gv[U][COUT], yv[U][COUT];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // clamped addresses + select: no load sits under a branch
      const long m = mb + (long)u * PPB;
      const long mc = m < m1 ? m : M - 1;
      xv[u] = *(const float4*)(x + mc * ldx + 4 * q);
      // g_y: NHWC rows (gy_plane = 0, pixel stride COUT) or NCHW planes (gy_plane = H*W: [b][c][pix], the flow maps' layout)
      const long gb = gy_plane ? (mc / HW) * COUT * HW + (mc % HW) : mc * COUT;
#pragma unroll
      for (int c = 0; c < COUT; ++c) {
        gv[u][c] = gy[gb + (gy_plane ? c * gy_plane : c)];
        yv[u][c] = yo[mc * ldy + c];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = mb + (long)u * PPB;
      const bool ok = m < m1;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < COUT; ++c) {
        const float o = yv[u][c];
        float d = 1.f;
        if (act == 1) d = 1.0f - o * o;
        else if (act == 2) d = o * (1.0f - o);
        else if (act == 3) d = o > 0.f ? 1.f : 0.f;
        const float gg = ok ? gv[u][c] * d : 0.f;
        bs[c] += gg;
        acc[c][0] += gg * xv[u].x, acc[c][1] += gg * xv[u].y, acc[c][2] += gg * xv[u].z, acc[c][3] += gg * xv[u].w;
        r.x += wr[c].x * gg, r.y += wr[c].y * gg, r.z += wr[c].z * gg, r.w += wr[c].w * gg;
      }
      if (ok && gx) *(float4*)(gx + m * ldgx + 4 * q) = r;
    }
  }
  constexpr int RW = COUT * 4 + 1;  // odd row stride: conflict-free column reads
#pragma unroll
  for (int c = 0; c < COUT; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[tid * RW + c * 4 + j] = acc[c][j];
  __syncthreads();
  float* out = part + (long)blockIdx.x * (COUT * Cin + COUT);
  for (int e = tid; e < COUT * Cin; e += 256) {
    const int co = e / Cin, ci = e - co * Cin, qq = ci >> 2, j = ci & 3;
    float v = 0.f;
    for (int k = 0; k < PPB; ++k) v += red[(k * Q + qq) * RW + co * 4 + j];
    out[e] = v;
  }
  __syncthreads();
  if (q == 0)
#pragma unroll
    for (int c = 0; c < COUT; ++c) red[pp * COUT + c] = bs[c];
  __syncthreads();
  if (tid < COUT) {
    float v = 0.f;
    for (int k = 0; k < PPB; ++k) v += red[k * COUT + tid];
    out[COUT * Cin + tid] = v;
  }
}

static inline bool wg1_small_ok(int Cin, int Cout, int ksz, int stride, int ldx, const void* x) {
  const int Q = Cin >> 2;
  return ksz == 1 && stride == 1 && Cout <= 4 && Cin % 4 == 0 && ldx % 4 == 0 && Q >= 1 && Q <= 256 && 256 % Q == 0 &&
         ((uintptr_t)x & 15) == 0;
}

// ---------------------------------------------------------------------------
// 3x3: tap-per-wave
// ---------------------------------------------------------------------------
struct Wg9Geo {
  int B, H, W, Cin, OH, OW, Cout, ldx, ldg;
  int TW, R, lgTW;        // pixel tile: R rows x TW cols (TW * R = 32)
  int HR, HC;             // halo tile rows / cols
  int tiles_x, tiles_y;   // tiles per image row / column
  int tiles_per_split;    // pixel tiles per block
  long ntiles;
  int n_ct;
};

// (no uniform branches around the tile loads: a load under a branch is followed by s_waitcnt vmcnt(0))
template <int CT, int NT, int S, int VX, int VG>
__global__ __launch_bounds__(512) void k_wgrad9(const float* __restrict__ x, const float* __restrict__ gy,
                                                float* __restrict__ slab, float* __restrict__ gbias, Wg9Geo g,
                                                const int* __restrict__ redo) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // second pass behind k_wgrad9_b3 (evf_wgrad_b3gen.hip): only the input-channel tiles it flagged as not exactly
  // representable in bf16 are recomputed here
  if (redo && redo[(blockIdx.y % g.n_ct)] == 0) return;
  constexpr int XW = 32 * CT, GW = 32 * NT, XQ = XW / 4, GQ = GW / 4;
  constexpr int MAXHP = S == 1 ? 102 : 195;               // halo pixels: max over the (TW, R) choices
  constexpr int NLX = (MAXHP * XQ + 511) / 512;           // x float4 loads per thread
  constexpr int NLG = (32 * GQ + 511) / 512;              // g float4 loads per thread (1)
  float* s_x = (float*)smem_raw;                          // [HR*HC][XW]
  float* s_g = s_x + MAXHP * XW;                          // [32][GW]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane & 31, kg = lane >> 5;
  const int cit = blockIdx.y % g.n_ct, cot = blockIdx.y / g.n_ct;
  const int ci0 = cit * XW, co0 = cot * GW;
  const int nhp = g.HR * g.HC;
  const bool do_bias = gbias && cit == 0 && wv == 0;

  // per-thread halo slots of the x loads, fixed for all tiles: (hr, hc) and the element offset
  // from the tile's halo origin -- per tile only bounds compares and one add remain
  int h_rc[NLX], h_off[NLX];
#pragma unroll
  for (int i = 0; i < NLX; ++i) {
    const int idx = tid + 512 * i, hp = idx / XQ, q = idx - hp * XQ;
    const int hr = hp / g.HC, hc = hp - hr * g.HC;
    h_rc[i] = hp < nhp ? ((hr << 16) | hc) : -1;
    h_off[i] = (hr * g.W + hc) * g.ldx + ci0 + 4 * q;
  }
  // g tile slots
  int g_rc[NLG], g_off[NLG];
#pragma unroll
  for (int i = 0; i < NLG; ++i) {
    const int idx = tid + 512 * i, px = idx / GQ, q = idx - px * GQ;
    const int r = px >> g.lgTW, cc = px & (g.TW - 1);
    g_rc[i] = px < 32 ? ((r << 16) | cc) : -1;
    g_off[i] = (r * g.OW + cc) * g.ldg + co0 + 4 * q;
  }

  f32x16 acc[CT][NT], acc8[CT][NT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f, acc8[c][t][r] = 0.f;
  float bsum[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bsum[t] = 0.f;

  float4 xr[NLX], gr[NLG];
  // tile coordinates of the NEXT prefetch, advanced incrementally (no divisions in the loop)
  int n_tx, n_ty, n_b;
  {
    const long t0 = (long)blockIdx.x * g.tiles_per_split;
    n_tx = (int)(t0 % g.tiles_x);
    const long t1 = t0 / g.tiles_x;
    n_ty = (int)(t1 % g.tiles_y), n_b = (int)(t1 / g.tiles_y);
  }
  auto prefetch = [&](bool want) {
    const bool tok = want && n_b < g.B;
    const int b = min(n_b, g.B - 1);
    const int oy0 = n_ty * g.R, ox0 = n_tx * g.TW;
    const int y_org = oy0 * S - 1, x_org = ox0 * S - 1;
    const float* xo = x + (((long)b * g.H + y_org) * g.W + x_org) * g.ldx;  // may point before the image: only
                                                                             // dereferenced for in-range slots
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int hr = h_rc[i] >> 16, hc = h_rc[i] & 0xFFFF;
      const int sy = y_org + hr, sx = x_org + hc;
      const bool ok = tok && h_rc[i] >= 0 && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W;
      const int c = ci0 + 4 * ((tid + 512 * i) % XQ);
      float e[4];
      if (VX == 4) {
        const bool cv = ok && c + 4 <= g.Cin;
        const float4 v = *(const float4*)(cv ? xo + h_off[i] : x);
        e[0] = cv ? v.x : 0.f, e[1] = cv ? v.y : 0.f, e[2] = cv ? v.z : 0.f, e[3] = cv ? v.w : 0.f;
      } else if (VX == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool cv = ok && c + 2 * j + 2 <= g.Cin;
          const float2 v = *(const float2*)(cv ? xo + h_off[i] + 2 * j : x);
          e[2 * j] = cv ? v.x : 0.f, e[2 * j + 1] = cv ? v.y : 0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool cv = ok && c + j < g.Cin;
          const float t = *(cv ? xo + h_off[i] + j : x);
          e[j] = cv ? t : 0.f;
        }
      }
      xr[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
    const float* go = gy + (((long)b * g.OH + oy0) * g.OW + ox0) * g.ldg;
#pragma unroll
    for (int i = 0; i < NLG; ++i) {
      const int r = g_rc[i] >> 16, cc = g_rc[i] & 0xFFFF;
      const bool ok = tok && g_rc[i] >= 0 && oy0 + r < g.OH && ox0 + cc < g.OW;
      const int c = co0 + 4 * ((tid + 512 * i) % GQ);
      float e[4];
      if (VG == 4) {
        const bool cv = ok && c + 4 <= g.Cout;
        const float4 v = *(const float4*)(cv ? go + g_off[i] : gy);
        e[0] = cv ? v.x : 0.f, e[1] = cv ? v.y : 0.f, e[2] = cv ? v.z : 0.f, e[3] = cv ? v.w : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool cv = ok && c + j < g.Cout;
          const float t = *(cv ? go + g_off[i] + j : gy);
          e[j] = cv ? t : 0.f;
        }
      }
      gr[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
    // advance to the next tile
    if (++n_tx == g.tiles_x) {
      n_tx = 0;
      if (++n_ty == g.tiles_y) n_ty = 0, ++n_b;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NLX; ++i)
      if (h_rc[i] >= 0) *(float4*)(s_x + (tid + 512 * i) * 4) = xr[i];
#pragma unroll
    for (int i = 0; i < NLG; ++i)
      if (tid + 512 * i < 32 * GQ) *(float4*)(s_g + (tid + 512 * i) * 4) = gr[i];
  };

  const int dy = wv / 3, dx = wv - 3 * dy;  // this wave's tap (waves 0..7 = taps 0..7)
  prefetch(true);
#pragma unroll 1
  for (int it = 0; it < g.tiles_per_split; ++it) {
    commit();
    __syncthreads();
    prefetch(it + 1 < g.tiles_per_split);
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int p = 2 * j + kg, r = p >> g.lgTW, cc = p & (g.TW - 1);
      const int hidx = (r * S + dy) * g.HC + cc * S + dx;
      float av[CT], bv[NT];
#pragma unroll
      for (int c = 0; c < CT; ++c) av[c] = s_x[hidx * XW + c * 32 + row];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bv[t] = s_g[p * GW + t * 32 + row];
        bsum[t] += bv[t];
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[t], acc[c][t], 0, 0, 0);
    }
    // ninth tap (dy = dx = 2): the 16 pixel pairs are split over the 8 waves
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int p = 2 * (wv + 8 * jj) + kg, r = p >> g.lgTW, cc = p & (g.TW - 1);
      const int hidx = (r * S + 2) * g.HC + cc * S + 2;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc8[c][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_x[hidx * XW + c * 32 + row], s_g[p * GW + t * 32 + row],
                                                            acc8[c][t], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: slab[split][tap][ci][co] (co fastest)
  float* sl = slab + (long)blockIdx.x * 9 * g.Cin * g.Cout;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = co0 + t * 32 + row;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + c * 32 + cg_row(r, lane);
        if (ci < g.Cin && co < g.Cout) sl[((long)wv * g.Cin + ci) * g.Cout + co] = acc[c][t][r];
      }
    }
  // ninth tap: the 8 waves' partial tiles are summed through LDS (the staging buffers are free
  // now), one ci half at a time: red[8 waves][NT][32][32]
  float* red = (float*)smem_raw;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wv * NT + t) * 32 + cg_row(r, lane)) * 32 + row] = acc8[c][t][r];
    __syncthreads();
    for (int e = tid; e < NT * 1024; e += 512) {
      float v = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) v += red[w8 * NT * 1024 + e];
      const int t = e >> 10, ci_l = (e >> 5) & 31, co_l = e & 31;
      const int ci = ci0 + c * 32 + ci_l, co = co0 + t * 32 + co_l;
      if (ci < g.Cin && co < g.Cout) sl[((long)8 * g.Cin + ci) * g.Cout + co] = v;
    }
  }
  if (do_bias) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float v = bsum[t];
      v += __shfl_xor(v, 32, 64);
      const int co = co0 + t * 32 + row;
      if (kg == 0 && co < g.Cout) evf_atomic_add(gbias + co, v);
    }
  }
}

// gw[(co*cin_total + cin_off + ci)*9 + tap] (+)= sum_split slab[split][tap][ci][co]
// block = 64 outputs x WGR_G split groups (a single thread walking up to 256 splits was one exposed load
// latency per split); the groups meet in LDS, one plain store per output.
#define WGR_G 16
__global__ __launch_bounds__(64 * WGR_G) void k_wgrad_reduce(const float* __restrict__ slab, int nsplit, int Cin, int Cout,
                                                             int cin_total, int cin_off, int accumulate,
                                                             float* __restrict__ gw, int* __restrict__ clear_flags) {
  __shared__ float red[WGR_G][64];
  // the redo flags of the bf16 pass (read by the fp32 pass, a kernel earlier in the stream) are handed back zeroed
  if (clear_flags && blockIdx.x == 0 && threadIdx.x < 64) clear_flags[threadIdx.x] = 0;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, G = blockDim.x >> 6;  // G = min(16, nsplit) groups
  const long per = (long)9 * Cin * Cout;
  const long e = (long)blockIdx.x * 64 + tx;
  const long ec = e < per ? e : per - 1;
  const float* __restrict__ p = slab + ec;
  float s0 = 0.f, s1 = 0.f;
  float s2 = 0.f, s3 = 0.f;
  int k = ty;
  for (; k + 3 * G < nsplit; k += 4 * G) {  // four independent loads in flight (hundreds of slabs of a few outputs: latency)
    s0 += p[(long)k * per];
    s1 += p[(long)(k + G) * per];
    s2 += p[(long)(k + 2 * G) * per];
    s3 += p[(long)(k + 3 * G) * per];
  }
  for (; k < nsplit; k += G) s0 += p[(long)k * per];
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty != 0 || e >= per) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += red[g][tx];
  const int co = (int)(e % Cout), ci = (int)((e / Cout) % Cin), tap = (int)(e / ((long)Cout * Cin));
  if (cin_off + ci >= cin_total) return;  // alignment-padding channel of the activation: no weight behind it
  float* d = gw + ((long)co * cin_total + cin_off + ci) * 9 + tap;
  *d = accumulate ? *d + s : s;
}

// ---------------------------------------------------------------------------
// 3x3 stride 1 with FOUR input channels: the real-valued head of a decoder's input (its two flow-prediction channels + the
// alignment pair, models/unet.py:371-388 of the reference via hip_ops.conv_wgrad(analog_head=...)).  36 x Cout sums over all
// pixels are a streaming reduction, not a matrix product: k_wgrad9 padded the four channels to a 32-wide MFMA tile and staged
// every 32 pixels through LDS behind two barriers -- 149 / 57 / 31 us per LIF-EV-FlowNet decoder for 1.2 GFLOP.  Here a thread
// owns ONE output channel and a run of FW_SEG pixels of one image row: it slides a 3 x 3 window of float4 (the four input
// channels of a pixel) along the row -- three new float4 per pixel, the same for all Cout threads of the row (L1 broadcast) --
// and keeps its 9 x 4 sums in registers; g is read once, coalesced along co.  The 256 / Cout rows of a block meet in LDS, blocks
// write slabs [block][tap][4][co] for k_wgrad_reduce.
// ---------------------------------------------------------------------------
#define FW_SEG 64
__global__ __launch_bounds__(256) void k_wgrad9_fewin(const float* __restrict__ x, const float* __restrict__ gy,
                                                      float* __restrict__ slab, int B, int H, int W, int Cout, int ldx, int ldg,
                                                      int nseg, long nitems) {
  // xs: the three input rows of every row segment of the block, columns c0 - 1 .. c0 + FW_SEG (zero outside the image), staged
  // once by the whole block -- the Cout threads of a segment then read them as LDS broadcasts instead of issuing the same three
  // global loads each (4 -> 1 global loads per pixel and thread); red: the block's 256 / Cout partial sums meet here afterwards
  __shared__ float4 xs[8 * 3 * (FW_SEG + 2)];
  __shared__ float red[256 * 36];
  const int tid = threadIdx.x, co = tid % Cout, grp = tid / Cout, NG = 256 / Cout;
  float acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  constexpr int RW = FW_SEG + 2;
  for (long it0 = (long)blockIdx.x * NG; it0 < nitems; it0 += (long)gridDim.x * NG) {  // (uniform trip count)
    __syncthreads();
    for (int idx = tid; idx < NG * 3 * RW; idx += 256) {
      const int gi = idx / (3 * RW), r3 = idx - gi * 3 * RW, r = r3 / RW, cc = r3 - r * RW;
      const long it = it0 + gi;
      const long itc = it < nitems ? it : nitems - 1;
      const int seg = (int)(itc % nseg);
      const long row = itc / nseg;
      const int y = (int)(row % H) + r - 1, c = seg * FW_SEG + cc - 1;
      const long b = row / H;
      const bool ok = it < nitems && y >= 0 && y < H && c >= 0 && c < W;
      const float4 v = *(const float4*)(x + ((b * H + min(max(y, 0), H - 1)) * (long)W + min(max(c, 0), W - 1)) * ldx);
      xs[idx] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const long it = it0 + grp;
    if (it < nitems) {
      const int seg = (int)(it % nseg);
      const long row = it / nseg;
      const int c0 = seg * FW_SEG, n = min(FW_SEG, W - c0);
      const float* gr = gy + (row * W + c0) * (long)ldg + co;
      const float4* xr = xs + grp * 3 * RW;
#pragma unroll 8
      for (int c = 0; c < n; ++c) {
        const float g = gr[(long)c * ldg];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float4 w0 = xr[r * RW + c], w1 = xr[r * RW + c + 1], w2 = xr[r * RW + c + 2];
          acc[3 * r + 0][0] = __fmaf_rn(w0.x, g, acc[3 * r + 0][0]), acc[3 * r + 0][1] = __fmaf_rn(w0.y, g, acc[3 * r + 0][1]);
          acc[3 * r + 0][2] = __fmaf_rn(w0.z, g, acc[3 * r + 0][2]), acc[3 * r + 0][3] = __fmaf_rn(w0.w, g, acc[3 * r + 0][3]);
          acc[3 * r + 1][0] = __fmaf_rn(w1.x, g, acc[3 * r + 1][0]), acc[3 * r + 1][1] = __fmaf_rn(w1.y, g, acc[3 * r + 1][1]);
          acc[3 * r + 1][2] = __fmaf_rn(w1.z, g, acc[3 * r + 1][2]), acc[3 * r + 1][3] = __fmaf_rn(w1.w, g, acc[3 * r + 1][3]);
          acc[3 * r + 2][0] = __fmaf_rn(w2.x, g, acc[3 * r + 2][0]), acc[3 * r + 2][1] = __fmaf_rn(w2.y, g, acc[3 * r + 2][1]);
          acc[3 * r + 2][2] = __fmaf_rn(w2.z, g, acc[3 * r + 2][2]), acc[3 * r + 2][3] = __fmaf_rn(w2.w, g, acc[3 * r + 2][3]);
        }
      }
    }
  }
  // the NG segments of the block meet in LDS: red[grp][tap * 4 + ci][co]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(grp * 36 + t * 4 + j) * Cout + co] = acc[t][j];
  __syncthreads();
  float* out = slab + (long)blockIdx.x * 36 * Cout;  // [tap][ci 4][co]
  for (int e = tid; e < 36 * Cout; e += 256) {
    float v = 0.f;
    for (int k = 0; k < NG; ++k) v += red[k * 36 * Cout + e];
    out[e] = v;
  }
}

static bool wg_fewin_ok(const float* x, const float* gy, int Cin, int Cout, int ksz, int stride, int ldx, const float* gbias) {
  return ksz == 3 && stride == 1 && Cin == 4 && !gbias && (Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256) && (ldx & 3) == 0 &&
         (((uintptr_t)x) & 15) == 0 && gy;
}
static int wg_fewin_blocks(int B, int H, int W, int Cout) {
  const long nitems = (long)B * H * evf_cdiv(W, FW_SEG);
  const long nb = evf_cdiv(nitems, 256 / Cout);
  return (int)(nb < 1024 ? nb : 1024);
}

// The same reduction with COALESCED gradient accesses, for the big weights: k_wgrad_reduce walks the slab layout [tap][ci][co], so
// the 64 lanes of a wave write (and, accumulating, read) 64 different lines of gw [co][ci][tap] -- 18 KB apart for 512 input
// channels -- and every line is touched by 32 different waves: 27 us for the 56 MB of a 512 x 512 layer (2.1 TB/s), 0.47 ms per
// LIF-EV-FlowNet step.  Here a block owns 32 output x 16 input channels x 9 taps: the slabs are read in 128-byte lines along co
// and summed over the splits (index order, four loads in flight), the tile is transposed in LDS, and each output channel's
// 16 x 9 = 144 consecutive floats of gw go out (and come in) as whole lines.
#define WRT_CO 32  // (32 output channels = one whole 128-byte line of a slab row per (tap, input channel))
#define WRT_CI 16
template <bool V4>
__global__ __launch_bounds__(256) void k_wgrad_reduce_t(const float* __restrict__ slab, int nsplit, int Cin, int Cout, int cin_total,
                                                        int cin_off, int accumulate, float* __restrict__ gw, int* __restrict__ clear_flags) {
  __shared__ float tile[WRT_CO * (WRT_CI * 9 + 1)];
  if (clear_flags && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) clear_flags[threadIdx.x] = 0;
  const int tid = threadIdx.x, co0 = blockIdx.x * WRT_CO, ci0 = blockIdx.y * WRT_CI;
  const long per = (long)9 * Cin * Cout;
  constexpr int NE = WRT_CO * WRT_CI * 9, NJ = NE / 256, TP = WRT_CI * 9 + 1;
  if (V4) {
    // 16-byte loads along co (round 6; Cout % 4 == 0, slabs 16-byte aligned): a thread takes four output channels of one (tap, input
    // channel) -- 5 trips of 256 threads for the 1152 quads instead of 18 trips of scalar loads; per element the same four partial
    // sums in the same order: the same bits
    constexpr int NQ = NE / 4, NJ4 = (NQ + 255) / 256;
#pragma unroll
    for (int j = 0; j < NJ4; ++j) {
      const int idx = tid + 256 * j, co4 = idx % (WRT_CO / 4), r = idx / (WRT_CO / 4), ci_l = r % WRT_CI, tap = r / WRT_CI;
      const int co = co0 + 4 * co4, ci = ci0 + ci_l;
      const bool ok = idx < NQ && co < Cout && ci < Cin;  // (Cout % 4 == 0: a quad is inside or outside as a whole)
      const float4* __restrict__ p = (const float4*)(slab + ((long)min(tap, 8) * Cin + (ok ? ci : 0)) * Cout + (ok ? co : 0));
      const long per4 = per >> 2;
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
      auto acc4 = [](float4& a, const float4 v) { a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w; };
      int k = 0;
      for (; k + 3 < nsplit; k += 4) {
        acc4(a0, p[(long)k * per4]);
        acc4(a1, p[(long)(k + 1) * per4]);
        acc4(a2, p[(long)(k + 2) * per4]);
        acc4(a3, p[(long)(k + 3) * per4]);
      }
      for (; k < nsplit; ++k) acc4(a0, p[(long)k * per4]);
      if (idx < NQ) {
        float* t = tile + (4 * co4) * TP + ci_l * 9 + tap;
        t[0] = ok ? (a0.x + a1.x) + (a2.x + a3.x) : 0.f;
        t[TP] = ok ? (a0.y + a1.y) + (a2.y + a3.y) : 0.f;
        t[2 * TP] = ok ? (a0.z + a1.z) + (a2.z + a3.z) : 0.f;
        t[3 * TP] = ok ? (a0.w + a1.w) + (a2.w + a3.w) : 0.f;
      }
    }
  } else {
#pragma unroll 3
  for (int j = 0; j < NJ; ++j) {  // (18 trips: three of them, i.e. twelve loads, in flight)
    const int idx = tid + 256 * j, co_l = idx % WRT_CO, r = idx / WRT_CO, ci_l = r % WRT_CI, tap = r / WRT_CI;
    const int co = co0 + co_l, ci = ci0 + ci_l;
    const bool ok = co < Cout && ci < Cin;
    const float* __restrict__ p = slab + ((long)tap * Cin + (ok ? ci : 0)) * Cout + (ok ? co : 0);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = 0;
    for (; k + 3 < nsplit; k += 4) {
      a0 += p[(long)k * per];
      a1 += p[(long)(k + 1) * per];
      a2 += p[(long)(k + 2) * per];
      a3 += p[(long)(k + 3) * per];
    }
    for (; k < nsplit; ++k) a0 += p[(long)k * per];
    tile[co_l * TP + ci_l * 9 + tap] = ok ? (a0 + a1) + (a2 + a3) : 0.f;
  }
  }
  __syncthreads();
#pragma unroll 2
  for (int j = 0; j < NJ; ++j) {
    const int idx = tid + 256 * j, r = idx % (WRT_CI * 9), co_l = idx / (WRT_CI * 9), ci_l = r / 9;
    const int co = co0 + co_l, ci = ci0 + ci_l;
    if (co < Cout && ci < Cin && cin_off + ci < cin_total) {
      float* d = gw + ((long)co * cin_total + cin_off + ci0) * 9 + r;
      const float v = tile[co_l * TP + r];
      *d = accumulate ? *d + v : v;
    }
  }
}

struct Wg9Plan {
  Wg9Geo g;
  int CT, NT, nsplit, n_nt;
};

static Wg9Plan wg9_plan(int B, int H, int W, int Cin, int Cout, int stride, int ldx, int ldg) {
  Wg9Plan p;
  Wg9Geo& g = p.g;
  g.B = B, g.H = H, g.W = W, g.Cin = Cin, g.Cout = Cout, g.ldx = ldx, g.ldg = ldg;
  g.OH = cg_out_dim(H, 3, stride), g.OW = cg_out_dim(W, 3, stride);
  g.TW = g.OW > 16 ? 32 : (g.OW > 8 ? 16 : 8);
  g.R = 32 / g.TW;
  g.lgTW = g.TW == 32 ? 5 : (g.TW == 16 ? 4 : 3);
  g.HR = (g.R - 1) * stride + 3, g.HC = (g.TW - 1) * stride + 3;
  g.tiles_x = evf_cdiv(g.OW, g.TW), g.tiles_y = evf_cdiv(g.OH, g.R);
  g.ntiles = (long)B * g.tiles_x * g.tiles_y;
  p.CT = Cin > 32 ? 2 : 1, p.NT = Cout > 32 ? 2 : 1;
  g.n_ct = evf_cdiv(Cin, 32 * p.CT);
  p.n_nt = evf_cdiv(Cout, 32 * p.NT);
  const long wt = (long)g.n_ct * p.n_nt;
  // one resident block per CU (192 VGPRs x 8 waves): at most 256 blocks = ONE round (a 257th block would run alone
  // in a second round), at least 4 pixel tiles per block (the slab traffic is 9*Cin*Cout per split)
  long ns = 256 / wt;  // floor: weight tiles x splits <= 256
  if (ns > g.ntiles / 4) ns = g.ntiles / 4;
  if (ns < 1) ns = 1;
  g.tiles_per_split = evf_cdiv(g.ntiles, ns);
  p.nsplit = evf_cdiv(g.ntiles, g.tiles_per_split);
  return p;
}

// workspace (floats) evf_conv2d_wgrad needs for this geometry (1x1: only the few-output-channel streaming kernel uses one)
extern "C" int64_t evf_conv2d_wgrad_ws(int B, int H, int W, int Cin, int Cout, int ksz, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return 0;
  if (ksz == 1) return wg1_small_ok(Cin, Cout, ksz, stride, Cin, nullptr) ? (int64_t)WG1_BLOCKS * (Cout * Cin + Cout) : 0;
  if (ksz != 3) return 0;
  const Wg9Plan p = wg9_plan(B, H, W, Cin, Cout, stride, Cin, Cout);
  int64_t n = (int64_t)p.nsplit * 9 * Cin * Cout;
  if (Cin == 4 && stride == 1 && (Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256)) {  // (k_wgrad9_fewin's slabs)
    const int64_t m = (int64_t)wg_fewin_blocks(B, H, W, Cout) * 36 * Cout;
    if (m > n) n = m;
  }
  return n;
}

template <int CT, int NT, int S>
static void wg9_launch(const float* x, const float* gy, float* slab, float* gbias, const Wg9Plan& p, hipStream_t st,
                       const int* redo = nullptr) {
  const Wg9Geo& g = p.g;
  dim3 grid(p.nsplit, g.n_ct * p.n_nt), block(512);
  constexpr int MAXHP = S == 1 ? 102 : 195;
  size_t smem = (size_t)(MAXHP * 32 * CT + 32 * 32 * NT) * sizeof(float);
  const size_t red = (size_t)8 * NT * 1024 * sizeof(float);
  if (smem < red) smem = red;
  const bool a16 = ((uintptr_t)x & 15) == 0, a8 = ((uintptr_t)x & 7) == 0;
  const bool gv = (g.ldg & 3) == 0 && (g.Cout & 3) == 0 && ((uintptr_t)gy & 15) == 0;
  const int vx = (g.Cin % 4 == 0 && g.ldx % 4 == 0 && a16) ? 4 : ((g.Cin % 2 == 0 && g.ldx % 2 == 0 && a8) ? 2 : 1);
#define WG9_GO(VX_, VG_) hipLaunchKernelGGL((k_wgrad9<CT, NT, S, VX_, VG_>), grid, block, smem, st, x, gy, slab, gbias, g, redo)
  if (gv) {
    if (vx == 4) WG9_GO(4, 4);
    else if (vx == 2) WG9_GO(2, 4);
    else WG9_GO(1, 4);
  } else {
    if (vx == 4) WG9_GO(4, 1);
    else if (vx == 2) WG9_GO(2, 1);
    else WG9_GO(1, 1);
  }
#undef WG9_GO
}

__device__ int g_wg_redo[64];
#define WG_TICKETS 1024
__device__ int g_wg_ticket[WG_TICKETS];  // arrival tickets of the fused slab reduction (k_wgrad9_b3): zero at load, handed back zero
static int* wg_tickets() {
  static int* ptr[64] = {nullptr};  // per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!ptr[dev] && hipGetSymbolAddress((void**)&ptr[dev], HIP_SYMBOL(g_wg_ticket)) != hipSuccess) return nullptr;
  return ptr[dev];
}
static int* wg_redo_flags() {
  static int* ptr[64] = {nullptr};  // per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!ptr[dev] && hipGetSymbolAddress((void**)&ptr[dev], HIP_SYMBOL(g_wg_redo)) != hipSuccess) return nullptr;
  return ptr[dev];
}

// g_w [Cout][cin_total][k][k] (torch layout; this call fills input channels cin_off .. cin_off+Cin) and optional
// g_bias [Cout]; accumulate = 0 overwrites them (only allowed when the call covers the whole weight).
// ws: evf_conv2d_wgrad_ws() floats of scratch (3x3 only).
extern "C" int evf_conv2d_wgrad(const float* x, int ldx, const float* g_y, int ldg, float* g_w, float* g_bias, int B, int H,
                                int W, int Cin, int Cout, int ksz, int stride, int cin_total, int cin_off, int accumulate,
                                float* ws, void* stream) {
  if (!x || !g_y || !g_w || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !EVF_KSZ_OK(ksz) ||
      (stride != 1 && stride != 2) || ldx < Cin || ldg < Cout || cin_off < 0 || cin_off >= cin_total ||
      (!(accumulate & 1) && (cin_off != 0 || Cin < cin_total)) || (ksz == 3 && !ws) || ((uintptr_t)g_y & 15))
    return EVF_EINVAL;
  const bool analog = (accumulate & 2) != 0;  // the caller knows x is not spike-valued: no bf16 attempt
  // 4: the caller GUARANTEES an x that is exactly representable in bf16 (spikes, {0,1,2,..} residual sums, bilinear x2 blends of
  // those -- known from how the tensor was made, models/hip_ops.py): ONE launch, slab reduction fused into the kernel's tail
  const bool exact = (accumulate & 4) != 0;
  accumulate &= 1;
  hipStream_t st = EVF_STREAM(stream);
  if (!accumulate && g_bias) {
    const int rc = evf_hip(evf_memset_async(g_bias, 0, sizeof(float) * (size_t)Cout, st));
    if (rc) return rc;
  }
  static const bool fewin_on = !(getenv("EVF_WGRAD_FEWIN") && !strcmp(getenv("EVF_WGRAD_FEWIN"), "0"));
  if (fewin_on && wg_fewin_ok(x, g_y, Cin, Cout, ksz, stride, ldx, g_bias)) {
    const int nseg = evf_cdiv(W, FW_SEG), nblk = wg_fewin_blocks(B, H, W, Cout);
    hipLaunchKernelGGL(k_wgrad9_fewin, dim3(nblk), dim3(256), 0, st, x, g_y, ws, B, H, W, Cout, ldx, ldg, nseg, (long)B * H * nseg);
    int rc = evf_status();
    if (rc) return rc;
    const long per = (long)36 * Cout;
    const int rg = nblk >= 16 ? 16 : (nblk >= 8 ? 8 : (nblk >= 4 ? 4 : (nblk >= 2 ? 2 : 1)));
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(evf_cdiv(per, 64)), dim3(64 * rg), 0, st, ws, nblk, 4, Cout, cin_total, cin_off, accumulate,
                       g_w, (int*)nullptr);
    return evf_status();
  }
  if (ksz == 3) {
    const Wg9Plan p = wg9_plan(B, H, W, Cin, Cout, stride, ldx, ldg);
    // stride 1: the bf16 matrix-core kernel first (exact for spike-valued inputs); the fp32 kernel then recomputes only the
    // input-channel tiles it flagged.  EVF_WGRAD=f32 keeps the fp32 kernel alone.
    static const bool b3 = !(getenv("EVF_WGRAD") && !strcmp(getenv("EVF_WGRAD"), "f32"));
    const int* redo = nullptr;
    float* bias_f32 = g_bias;
    static const bool b3_s2 = !(getenv("EVF_WGRAD_S2") && !strcmp(getenv("EVF_WGRAD_S2"), "f32"));  // (stride 2 on the bf16 kernel: A/B switch)
    if (b3 && !analog && (stride == 1 || b3_s2) && p.g.n_ct <= 64 && evf_wgrad9_b3_ok(x, g_y, Cin, Cout, ldx, ldg)) {
      // 64 persistent flags per device (zero at load, cleared again by k_wgrad_reduce: no memset per call; calls are
      // stream-ordered on one stream per device)
      int* flags = wg_redo_flags();
      if (!flags) return EVF_EINVAL;
      // The fused reduction is OFF by default (EVF_WGRAD_FUSE=1 switches it on), measured on the LIF-EV-FlowNet step: the blocks of
      // a launch finish together, so every tile's last block runs its 9 x (read 4 slabs, transpose, scattered store) chain at the END
      // of the kernel with nothing to hide it -- +47 us per call over k_wgrad_reduce's 25 us launch (8.75 against 8.42 ms per step)
      // for <= 8 splits; with 16..256 splits (few weight tiles) the last block reads 1..9 MB through ONE CU: 13.4 ms per step.
      static const bool fuse_ok = getenv("EVF_WGRAD_FUSE") && !strcmp(getenv("EVF_WGRAD_FUSE"), "1");
      if (exact && fuse_ok && stride == 1 && p.nsplit <= 8 && (long)p.g.n_ct * p.n_nt <= WG_TICKETS) {
        int* tickets = wg_tickets();
        if (!tickets) return EVF_EINVAL;
        return evf_wgrad9_b3_launch(x, ldx, g_y, ldg, ws, g_bias, flags, B, H, W, Cin, Cout, p.nsplit, p.CT, p.NT, st, g_w, tickets,
                                    cin_total, cin_off, accumulate, 1, 1);
      }
      int rc = evf_wgrad9_b3_launch(x, ldx, g_y, ldg, ws, g_bias, flags, B, H, W, Cin, Cout, p.nsplit, p.CT, p.NT, st, nullptr, nullptr,
                                    0, 0, 0, exact ? 1 : 0, stride);
      if (rc) return rc;
      redo = flags;
      bias_f32 = nullptr;  // (summed by the bf16 kernel from the exact fp32 gradients)
    }
#define WG9(CT_, NT_)                                                 \
  if (stride == 1)                                                    \
    wg9_launch<CT_, NT_, 1>(x, g_y, ws, bias_f32, p, st, redo);       \
  else                                                                \
    wg9_launch<CT_, NT_, 2>(x, g_y, ws, bias_f32, p, st, redo)
    if (redo && exact) {
      // x exact by construction: the bf16 kernel raised no flag, the fp32 pass would exit at once in every block -- not launched
    } else if (p.CT == 2 && p.NT == 2) {
      WG9(2, 2);
    } else if (p.CT == 2) {
      WG9(2, 1);
    } else if (p.NT == 2) {
      WG9(1, 2);
    } else {
      WG9(1, 1);
    }
#undef WG9
    int rc = evf_status();
    if (rc) return rc;
    const long per = (long)9 * Cin * Cout;
    static const bool rt_on = !(getenv("EVF_WGRAD_REDUCE_T") && !strcmp(getenv("EVF_WGRAD_REDUCE_T"), "0"));
    // big weights with few splits: the transposing reduction (coalesced gradient lines); many splits of a small weight keep the
    // 16 split groups per output of k_wgrad_reduce
    if (rt_on && p.nsplit <= 32 && (long)evf_cdiv(Cout, WRT_CO) * evf_cdiv(Cin, WRT_CI) >= 128) {
      static const bool v4_on = !(getenv("EVF_WGRAD_REDUCE_V4") && !strcmp(getenv("EVF_WGRAD_REDUCE_V4"), "0"));
      if (v4_on && Cout % 4 == 0 && (((uintptr_t)ws) & 15) == 0 && (per & 3) == 0)
        hipLaunchKernelGGL(k_wgrad_reduce_t<true>, dim3(evf_cdiv(Cout, WRT_CO), evf_cdiv(Cin, WRT_CI)), dim3(256), 0, st, ws, p.nsplit, Cin,
                           Cout, cin_total, cin_off, accumulate, g_w, (int*)redo);
      else
        hipLaunchKernelGGL(k_wgrad_reduce_t<false>, dim3(evf_cdiv(Cout, WRT_CO), evf_cdiv(Cin, WRT_CI)), dim3(256), 0, st, ws, p.nsplit, Cin,
                           Cout, cin_total, cin_off, accumulate, g_w, (int*)redo);
      return evf_status();
    }
    const int rg = p.nsplit >= 16 ? 16 : (p.nsplit >= 8 ? 8 : (p.nsplit >= 4 ? 4 : (p.nsplit >= 2 ? 2 : 1)));
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(evf_cdiv(per, 64)), dim3(64 * rg), 0, st, ws, p.nsplit, Cin, Cout, cin_total,
                       cin_off, accumulate, g_w, (int*)redo);
    return evf_status();
  }
  if (ws && wg1_small_ok(Cin, Cout, ksz, stride, ldx, x)) {
    const long M = (long)B * H * W;
    const int ppb = (int)evf_cdiv(M, WG1_BLOCKS);
    const int nblk = (int)evf_cdiv(M, ppb);
#define WG1(C_)                                                                                                     \
  case C_:                                                                                                          \
    hipLaunchKernelGGL(k_wgrad1_small<C_>, dim3(nblk), dim3(256), 0, st, x, g_y, ws, M, Cin, ldx, ldg, ppb);        \
    break
    switch (Cout) {
      WG1(1);
      WG1(2);
      WG1(3);
      WG1(4);
    }
#undef WG1
    int rc = evf_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_wgrad1_reduce, dim3(Cout * Cin + Cout), dim3(64), 0, st, ws, nblk, Cin, Cout, cin_total, cin_off,
                       accumulate, g_w, g_bias);
    return evf_status();
  }
  if (!accumulate) {
    const int rc = evf_hip(evf_memset_async(g_w, 0, sizeof(float) * (size_t)Cout * cin_total * ksz * ksz, st));
    if (rc) return rc;
  }
  WgGeo g;
  g.B = B, g.H = H, g.W = W, g.Cin = Cin, g.Cout = Cout, g.ksz = ksz, g.stride = stride, g.ldx = ldx, g.ldg = ldg;
  g.cin_total = cin_total, g.cin_off = cin_off;
  g.OH = cg_out_dim(H, ksz, stride), g.OW = cg_out_dim(W, ksz, stride);
  const long M = (long)B * g.OH * g.OW;
  const int CT = Cin > 32 ? 2 : 1, NT = Cout > 32 ? 2 : 1;
  const long tiles = (long)evf_cdiv(Cin, 32 * CT) * evf_cdiv(Cout, 32 * NT) * ksz * ksz;
  const long st_total = evf_cdiv(M, 32);
  long ksplit = evf_cdiv(1024, tiles);
  if (ksplit > st_total / 8) ksplit = st_total / 8;
  if (ksplit < 1) ksplit = 1;
  g.stages = evf_cdiv(st_total, ksplit);
  ksplit = evf_cdiv(st_total, g.stages);
  const bool vec4 = Cin % 4 == 0 && Cout % 4 == 0 && ldx % 4 == 0 && ldg % 4 == 0 && ((uintptr_t)x & 15) == 0 &&
                    ((uintptr_t)g_y & 15) == 0;
  if (CT == 2 && NT == 2)
    wg_launch<2, 2>(x, g_y, g_w, g_bias, g, (int)ksplit, vec4, st);
  else if (CT == 2)
    wg_launch<2, 1>(x, g_y, g_w, g_bias, g, (int)ksplit, vec4, st);
  else if (NT == 2)
    wg_launch<1, 2>(x, g_y, g_w, g_bias, g, (int)ksplit, vec4, st);
  else
    wg_launch<1, 1>(x, g_y, g_w, g_bias, g, (int)ksplit, vec4, st);
  return evf_status();
}

// floats of scratch evf_head1x1_bwd needs
extern "C" int64_t evf_head1x1_ws(int Cin, int Cout) { return (int64_t)WG1_BLOCKS * ((int64_t)Cout * Cin + Cout); }

static inline bool head1_ok(int Cin, int Cout, int ldx, const void* x) {
  const int Q = Cin >> 2;
  return Cout >= 1 && Cout <= 4 && Cin % 4 == 0 && ldx % 4 == 0 && Q >= 1 && Q <= 64 && (Q & (Q - 1)) == 0 && ((uintptr_t)x & 15) == 0;
}

// y [M][ldy] (first Cout columns) = act(x [M][ldx] W^T + bias); W [Cout][Cin] (torch layout of a 1x1 conv), Cout <= 4, Cin / 4 a power
// of two <= 64; act: 0 none, 1 tanh, 2 sigmoid, 3 relu.
extern "C" int evf_head1x1_fwd(const float* x, int ldx, const float* w, const float* bias, int act, int64_t M, int Cin, int Cout,
                               float* y, int ldy, void* stream) {
  if (!x || !w || !y || M <= 0 || act < 0 || act > 3 || ldy < Cout || !head1_ok(Cin, Cout, ldx, x)) return EVF_EINVAL;
  const int PPB = 256 / (Cin >> 2);
  long nb = evf_cdiv(M, (long)PPB * 4);
  if (nb > 4096) nb = 4096;
#define H1F(C_)                                                                                                                 \
  case C_:                                                                                                                      \
    hipLaunchKernelGGL(k_head1_fwd<C_>, dim3((int)nb), dim3(256), 0, EVF_STREAM(stream), x, w, bias, act, (long)M, Cin, ldx, y, ldy); \
    break
  switch (Cout) {
    H1F(1);
    H1F(2);
    H1F(3);
    H1F(4);
  }
#undef H1F
  return evf_status();
}

// Backward of the layer above: g_x [M][ldgx] (null: not wanted) = W^T g_pre, g_w [Cout][Cin] and g_bias [Cout] (null: no bias)
// += (accumulate) or = their sums over the M pixels, with g_pre = g_y * act'(y).  g_y: [M][Cout] rows (gy_nchw_hw = 0) or NCHW planes
// [b][Cout][H*W] (gy_nchw_hw = H*W: the layout the flow maps' gradient arrives in).  ws: evf_head1x1_ws() floats.
extern "C" int evf_head1x1_bwd(const float* x, int ldx, const float* y, int ldy, const float* g_y, int64_t gy_nchw_hw, const float* w, int act,
                               int64_t M, int Cin, int Cout, float* g_x, int ldgx, float* g_w, float* g_bias, int accumulate, float* ws,
                               void* stream) {
  if (!x || !y || !g_y || !w || !g_w || !ws || M <= 0 || act < 0 || act > 3 || ldy < Cout || (g_x && (ldgx < Cin || (ldgx & 3))) ||
      gy_nchw_hw < 0 || (gy_nchw_hw && M % gy_nchw_hw) || !head1_ok(Cin, Cout, ldx, x))
    return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  if (!accumulate && g_bias) {
    const int rc = evf_hip(evf_memset_async(g_bias, 0, sizeof(float) * (size_t)Cout, st));
    if (rc) return rc;
  }
  const int ppb = (int)evf_cdiv(M, WG1_BLOCKS);
  const int nblk = (int)evf_cdiv(M, ppb);
#define H1B(C_)                                                                                                                    \
  case C_:                                                                                                                         \
    hipLaunchKernelGGL(k_head1_bwd<C_>, dim3(nblk), dim3(256), 0, st, x, y, g_y, (long)gy_nchw_hw, w, act, (long)M,                 \
                       (long)(gy_nchw_hw ? gy_nchw_hw : 1), Cin, ldx, ldy, g_x, ldgx, ws, ppb);                                     \
    break
  switch (Cout) {
    H1B(1);
    H1B(2);
    H1B(3);
    H1B(4);
  }
#undef H1B
  int rc = evf_status();
  if (rc) return rc;
  hipLaunchKernelGGL(k_wgrad1_reduce, dim3(Cout * Cin + Cout), dim3(64), 0, st, ws, nblk, Cin, Cout, Cin, 0, accumulate, g_w, g_bias);
  return evf_status();
}
