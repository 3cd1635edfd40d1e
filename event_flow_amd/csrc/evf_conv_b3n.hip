// 3x3 stride-1 convolution (forward and input gradient) of layers with MANY OUTPUT channels per staged input tile: the input
// gradients of the spiking EV-FlowNet's decoders (reference models/unet.py:371-388, :455-465: a decoder reads cat(prediction, x,
// skip), so its input gradient maps 32 / 64 / 128 / 256 channels of dL/d(current) back onto 132 / 260 / 516 / 1028).
//
// k_conv3_b3t (evf_conv_b3tile.hip) gives every 64 output channels a block of their own (grid.y): each of them fetches and splits
// the SAME halo tile again, holds the CU alone (LDS) and lives for K / 16 channel groups -- two, for the 256 x 256 decoder --, so
// its prologue (a global round trip, the split, two barriers) and its epilogue are never hidden, and a remainder of 4 channels
// costs a launch of its own: 0.27 / 0.36 / 0.37 / 0.48 of the dense bf16 peak issued for the four decoders
// (tools/debug/c4_entry_times.py), 16 % of the LIF-EV-FlowNet step.  Here the tile is staged ONCE per channel group and all
// output channels of a chunk of up to 6 N tiles (192 channels) stream past it:
//   block     512 threads = 8 waves, output tile 8 rows x 32 columns, wave w owns row w (one 32-pixel M tile) and keeps the
//             accumulators of ALL N tiles of the chunk (6 x 16 registers);
//   LDS       3 planes x (10 x 34 halo pixels) x 48 B of the current 16-channel group, single buffered               48 KiB
//             2 x (2 N tiles x 9 taps x 3 planes x 1 KiB) weight fragments: a STAGE = (group, pair of N tiles), double
//             buffered by LDS-DMA                                                                                   108 KiB
//   registers the wave's nine activation fragments per plane of the current group (27 x 4), read once per group
//   pipeline  stage s + 1's weights arrive under the MFMAs of stage s (108 per wave in the six-term form); one barrier per stage.
//             Per group: the next group's halo floats are requested before its last stage and split into the planes behind it
//             (two barriers, amortised over all the group's stages instead of one pair of N tiles).
// Same arithmetic as the other kernels of the family: weights hi + mid + lo (evf_pack_conv2d_weight_b3), activations split exactly
// on the fly, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, a block-uniform vote per group picks 3 products or the 6 terms
// above 2^-24 of the leading one, smallest terms first; per output element the channel groups are added in index order like
// k_conv3_b3t: bit-identical results.  Split-K partial sums go to slabs the caller reduces (k_b3_reduce).
//
// MEASURED (round 6, tools/debug/c4_entry_times.py, us per input gradient of the four decoders, K = 32 / 64 / 128 / 256): 290-306 /
// 242-251 / 267-271 / 256-260 against the tile kernel's 347-356 / 260-267 / 248-253 / 185-191 -- it wins where the tile kernel's
// blocks live for two or four channel groups and loses where they live long: the plan takes it up to K = 64.  Probe builds
// (tools/ab_variant.py ... evf_conv_b3n.hip:-DN_PROBE_NOMFMA / -DN_PROBE_NODMA) say why it stops at 0.32-0.38 of the bf16 peak issued:
// without the MFMAs 185 / 143 / 151 / 135, without the weight DMA 237 / 181 / 188 / 189, with neither 133 / 88 / 80 / 70 -- the three
// add up instead of overlapping.  An 8 x 32 tile re-reads ALL weights of its chunk (276 KiB for the 256 x 256 decoder against 43 KiB of
// halo) and the matrix waves issue the DMA pieces themselves (130-160 cycles each, evf_dgrad_diag.hip), 7 per wave and stage; loader
// waves of their own do not fit beside eight waves of 256 registers.  Keeping all three activation planes in registers spills (98);
// hi + mid in registers, lo from LDS (this form) runs like the all-LDS form: the fragment reads were not the bound.
#include "evf_common.h"
#include "evf_split.h"
#include <stdlib.h>

typedef float n_f32x16 __attribute__((ext_vector_type(16)));
#ifdef N_PROBE_NOMFMA  // (probe build: the operand reads stay, the matrix pipe is idle)
#define N_MFMA(a, b, c) ([&]() { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define N_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef __bf16 n_bf16x8 __attribute__((ext_vector_type(8)));
typedef float n_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void n_lds_void;
typedef __attribute__((address_space(1))) const void n_glb_void;

#define N_ROWS 8
#define N_COLS 32
#define N_HR (N_ROWS + 2)
#define N_HC (N_COLS + 2)
#define N_PIX (N_HR * N_HC)           // 340 halo pixels
#define N_PSTRIDE 48                  // bytes per halo pixel and plane
#define N_PLANE (N_PIX * N_PSTRIDE)   // 16320
#define N_ATASKS (N_PIX * 4)          // float4 loads per group
#define N_AITER ((N_ATASKS + 511) / 512)
#define N_NC 6                        // N tiles (of 32 channels) per block at most: 3 pairs
#define N_WFRAG 54                    // 1 KiB weight fragments per stage: 2 N tiles x 9 taps x 3 terms
#define N_WBUF (N_WFRAG * 1024)
#define N_LDS (3 * N_PLANE + 2 * N_WBUF)
#define N_STAGE (4 * 3 * 64)          // uint4 per (N tile, tap, 64-channel group) of the packed weights: [chunk 4][term 3][lane 64]

struct NsGeo {
  int B, H, W, K, N;  // image (input = output size), contraction channels, output channels
  int lds, ldo;       // pixel strides (floats)
  int flip;           // 0 forward (tap (dy,dx) reads pixel (+dy-1,+dx-1)), 1 input gradient (reads (+1-dy,+1-dx))
  int tiles_y, tiles_x;
  int nc;             // N tiles per chunk (<= N_NC): chunk c of a block covers tiles [c nc, min((c + 1) nc, ntiles))
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv3_b3n(
    const float* __restrict__ src, const uint4* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ out, NsGeo g,
    int accumulate, int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_a = smem;                // [3 planes][340 px][48 B]
  char* s_w = smem + 3 * N_PLANE;  // [2 buffers][2 N tiles][9 taps][3 terms][64 lanes] uint4
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, kg = lane >> 5;
  // XCD-aware tile order: the blocks one XCD receives (blockIdx.x % 8) are spatial neighbours -> halo rows hit its L2
  const int ntile = g.B * g.tiles_y * g.tiles_x, per = (ntile + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tile >= ntile) return;
  const int txi = tile % g.tiles_x, t1 = tile / g.tiles_x, tyi = t1 % g.tiles_y, b = t1 / g.tiles_y;
  const int y0 = tyi * N_ROWS, x0 = txi * N_COLS;
  const int G64 = (g.K + 63) >> 6, KC = (g.K + 15) >> 4, ntiles = (g.N + 31) >> 5;
  const long wtile = (long)(9 * G64) * N_STAGE;
  const int nt_base = (int)blockIdx.y * g.nc;
  const int nt_here = min(g.nc, ntiles - nt_base);  // N tiles of this block (>= 1)
  const int npair = (nt_here + 1) >> 1;             // stages per group
  const float* img = src + (long)b * g.H * g.W * g.lds;

  // split-K: blockIdx.z owns the channel groups [kc_lo, kc_hi) and writes its partial sums to its own slab
  int kc_lo = 0, kc_hi = KC;
  if (ksplit > 1) {
    const int pr = (KC + ksplit - 1) / ksplit;
    kc_lo = min((int)blockIdx.z * pr, KC - 1), kc_hi = min(kc_lo + pr, KC);
    if ((int)blockIdx.z * pr >= KC) kc_hi = kc_lo;  // (an empty split still writes its zeros)
    out += (long)blockIdx.z * g.B * g.H * g.W * g.ldo;
  }

  // ---- staging of the halo: global -> registers -> exact split -> LDS planes
  n_f32x4 pa[N_AITER];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int i = 0; i < N_AITER; ++i) {
      const int task = min(tid + 512 * i, N_ATASKS - 1), px = task >> 2, q = task & 3;
      const int hy = px / N_HC, hx = px - hy * N_HC;
      const int sy = min(max(y0 + hy - 1, 0), g.H - 1), sx = min(max(x0 + hx - 1, 0), g.W - 1);
      const int c = kc * 16 + 4 * q;
      pa[i] = *(const n_f32x4*)(img + ((long)sy * g.W + sx) * g.lds + (c + 4 <= g.K ? c : 0));
    }
  };
  auto commit = [&](int kc) -> int {  // returns "some residual is not zero" for this thread's elements
    uint32_t nz = 0u;
#pragma unroll
    for (int i = 0; i < N_AITER; ++i) {
      const int task = tid + 512 * i, px = task >> 2, q = task & 3;
      const int hy = px / N_HC, hx = px - hy * N_HC;
      const int sy = y0 + hy - 1, sx = x0 + hx - 1;
      const bool ok = sy >= 0 && sy < g.H && sx >= 0 && sx < g.W && kc * 16 + 4 * q + 4 <= g.K;
      const n_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const n_f32x4 v = ok ? pa[i] : zero4;
      uint32_t h0, m0, l0, h1, m1, l1;
      evf_split3_pair(v.x, v.y, h0, m0, l0);
      evf_split3_pair(v.z, v.w, h1, m1, l1);
      nz |= m0 | m1;  // (mid = bf16(residual): zero iff the residual is zero)
      if (task < N_ATASKS) {
        char* p = s_a + px * N_PSTRIDE + q * 8;
        *(uint2*)(p) = make_uint2(h0, h1);
        *(uint2*)(p + N_PLANE) = make_uint2(m0, m1);
        *(uint2*)(p + 2 * N_PLANE) = make_uint2(l0, l1);
      }
    }
    return (nz & 0x7FFF7FFFu) != 0u;
  };
  // weight fragments of stage (kc, pair pr): 54 pieces of 1 KiB by LDS-DMA, 6-7 per wave (an N tile past the end: the last one again)
  auto dma_w = [&](int kc, int pr, int buf) {
#ifdef N_PROBE_NODMA
    return;
#endif
    const int gg = kc >> 2, ch = kc & 3;
    for (int f = wv; f < N_WFRAG; f += 8) {
      const int term = f % 3, f2 = f / 3, tap = f2 % 9, t = f2 / 9;
      const int nt = min(nt_base + 2 * pr + t, ntiles - 1);
      const uint4* srcw = wp + nt * wtile + (((long)tap * G64 + gg) * 4 + ch) * 192 + term * 64 + lane;
      __builtin_amdgcn_global_load_lds((n_glb_void*)srcw, (n_lds_void*)(s_w + buf * N_WBUF + f * 1024), 16, 0, 0);
    }
  };

  n_f32x16 acc[N_NC];
#pragma unroll
  for (int t = 0; t < N_NC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // The wave's activation fragments of a GROUP live in registers: tap (oy, ox) reads the pixel row wv + oy at column col + ox -- nine
  // fragments per plane, loaded once behind the group's commit and used by every stage of the group (all N tiles of the chunk).
  // Per stage the LDS then carries the weight fragments only: 6 reads per 12 MFMAs, the tile kernel's ratio (with the fragments
  // re-read per stage: 9 per 12, and the kernel ran at 0.33-0.39 of the bf16 peak issued whatever the layer).
  const char* arow = s_a + (wv * N_HC + col) * N_PSTRIDE + kg * 16;
  uint4 af[9][2];  // [tap][hi | mid]  (the lo plane, one of a tap's six products per N tile, is read per stage: 27 x 4 registers
                   //  for all three planes spill beside 96 accumulator registers; a group without residuals uses hi only)
  auto load_a = [&](const int inexact) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const char* ap = arow + ((tp / 3) * N_HC + (tp % 3)) * N_PSTRIDE;
      af[tp][0] = *(const uint4*)ap;
      if (inexact) af[tp][1] = *(const uint4*)(ap + N_PLANE);
    }
  };
  // the MFMAs of one stage: nine taps, this wave's row against two N tiles (acc a0 / a1; `two`: the second tile exists)
  auto stage = [&](const int buf, const int inexact, n_f32x16& a0, n_f32x16& a1, const bool two) {
    const uint4* wbuf = (const uint4*)(s_w + buf * N_WBUF) + lane;
    if (!inexact) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int wtap = g.flip ? 8 - tp : tp;
        const n_bf16x8 xa = *(const n_bf16x8*)&af[tp][0];
        const uint4* wq = wbuf + wtap * 192;
        {
          const uint4 q0 = wq[0], q1 = wq[64], q2 = wq[128];
          const n_bf16x8 wh = *(const n_bf16x8*)&q0, wm = *(const n_bf16x8*)&q1, wl = *(const n_bf16x8*)&q2;
          a0 = N_MFMA(wl, xa, a0);  // smallest terms first
          a0 = N_MFMA(wm, xa, a0);
          a0 = N_MFMA(wh, xa, a0);
        }
        if (two) {
          const uint4 q0 = wq[27 * 64], q1 = wq[27 * 64 + 64], q2 = wq[27 * 64 + 128];
          const n_bf16x8 wh = *(const n_bf16x8*)&q0, wm = *(const n_bf16x8*)&q1, wl = *(const n_bf16x8*)&q2;
          a1 = N_MFMA(wl, xa, a1);
          a1 = N_MFMA(wm, xa, a1);
          a1 = N_MFMA(wh, xa, a1);
        }
      }
    } else {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int wtap = g.flip ? 8 - tp : tp;
        const uint4 xlq = *(const uint4*)(arow + ((tp / 3) * N_HC + (tp % 3)) * N_PSTRIDE + 2 * N_PLANE);
        const n_bf16x8 xh = *(const n_bf16x8*)&af[tp][0], xm = *(const n_bf16x8*)&af[tp][1], xl = *(const n_bf16x8*)&xlq;
        const uint4* wq = wbuf + wtap * 192;
        {
          const uint4 q0 = wq[0], q1 = wq[64], q2 = wq[128];
          const n_bf16x8 wh = *(const n_bf16x8*)&q0, wm = *(const n_bf16x8*)&q1, wl = *(const n_bf16x8*)&q2;
          a0 = N_MFMA(wm, xm, a0);  // smallest terms first
          a0 = N_MFMA(wl, xh, a0);
          a0 = N_MFMA(wh, xl, a0);
          a0 = N_MFMA(wm, xh, a0);
          a0 = N_MFMA(wh, xm, a0);
          a0 = N_MFMA(wh, xh, a0);
        }
        if (two) {
          const uint4 q0 = wq[27 * 64], q1 = wq[27 * 64 + 64], q2 = wq[27 * 64 + 128];
          const n_bf16x8 wh = *(const n_bf16x8*)&q0, wm = *(const n_bf16x8*)&q1, wl = *(const n_bf16x8*)&q2;
          a1 = N_MFMA(wm, xm, a1);
          a1 = N_MFMA(wl, xh, a1);
          a1 = N_MFMA(wh, xl, a1);
          a1 = N_MFMA(wm, xh, a1);
          a1 = N_MFMA(wh, xm, a1);
          a1 = N_MFMA(wh, xh, a1);
        }
      }
    }
  };

  int inexact = 0;
  if (kc_hi > kc_lo) {
    dma_w(kc_lo, 0, 0);
    fetch(kc_lo);
    const int nzv = commit(kc_lo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's pieces of the first weight stage have landed)
    inexact = __syncthreads_or(nzv);
    load_a(inexact);
  }
  int sidx = 0;  // stages so far: the weight buffer of a stage is sidx & 1
#pragma unroll 1
  for (int kc = kc_lo; kc < kc_hi; ++kc) {
    const bool more = kc + 1 < kc_hi;
#pragma unroll
    for (int pr = 0; pr < N_NC / 2; ++pr) {
      if (pr < npair) {  // (block-uniform)
        const int buf = sidx & 1;
        const bool lastp = pr + 1 == npair;
        // the next stage's weights into the other buffer (last read in stage sidx - 1: every wave is past that barrier), and
        // before a group's last stage the next group's halo floats
        if (!lastp) dma_w(kc, pr + 1, buf ^ 1);
        else if (more) dma_w(kc + 1, 0, buf ^ 1);
        if (lastp && more) fetch(kc + 1);
        stage(buf, inexact, acc[2 * pr], acc[2 * pr + 1], 2 * pr + 1 < nt_here);
        ++sidx;
        if (!lastp) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the next stage's pieces of this wave have landed)
          __syncthreads();  // ... everybody's, and every wave is done with buffer `buf`
        }
      }
    }
    __syncthreads();  // every wave is done with this group's planes (and its last weight buffer)
    if (more) {
      const int nzv = commit(kc + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      inexact = __syncthreads_or(nzv);
      load_a(inexact);
    }
  }

  // ---- epilogue: weights are the A operand, the tile comes out transposed: lane = pixel (row wv, column col), registers
  // 4 q + e = channel n0 + 8 q + 4 kg + e of each N tile -> float4 stores (bias, accumulate)
  const bool vec = (g.ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0;  // uniform
  const int oy = y0 + wv, oxx = x0 + col;
  const bool mok = oy < g.H && oxx < g.W;
  float* orow = out + (((long)b * g.H + min(oy, g.H - 1)) * g.W + min(oxx, g.W - 1)) * g.ldo;
#pragma unroll
  for (int t = 0; t < N_NC; ++t) {
    if (t >= nt_here || !mok) continue;
    const int n0 = (nt_base + t) * 32 + 4 * kg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 8 * q;
      if (n >= g.N) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = acc[t][4 * q + e];
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

// Is this kernel the choice for the product, and with how many K splits?  0 = no, 1 = yes unsplit, n > 1 = yes with n slabs.
// It pays where the tile kernel would stage the same halo for several 64-channel blocks: at least three N tiles.
int evf_conv3_b3n_plan(const float* src, int B, int H, int W, int K, int N, int lds, bool force, int max_split, int force_split) {
  static const bool on = !(getenv("EVF_CONV_NSTREAM") && getenv("EVF_CONV_NSTREAM")[0] == '0');
  if (!on) return 0;
  if (K % 4 != 0 || lds % 4 != 0 || (((uintptr_t)src) & 15) != 0) return 0;  // float4 halo loads
  const int ntiles = evf_cdiv(N, 32);
  // (three or more N tiles, at most four channel groups: above that the tile kernel's longer-lived blocks win, see the header)
  if (!force && (ntiles < 3 || K < 16 || K > 64)) return 0;
  const long tiles = (long)B * evf_cdiv(H, N_ROWS) * evf_cdiv(W, N_COLS);
  const int nchunk = evf_cdiv(ntiles, N_NC);
  const long blocks = tiles * nchunk;
  const int KC = evf_cdiv(K, 16);
  const int smax = max(1, min(max_split, KC / 4));  // at least 4 channel groups per split
  int ks = blocks >= 192 ? 1 : (int)min((long)smax, evf_cdiv(256L, blocks));
  if (force_split > 0) ks = max(1, min(min(force_split, max(max_split, 1)), KC));
  if (force) return ks;
  // enough blocks for the 256 CUs, and tiles that are mostly inside the image
  const double fill = (double)H * W / ((double)evf_cdiv(H, N_ROWS) * N_ROWS * evf_cdiv(W, N_COLS) * N_COLS);
  return (blocks * ks >= 160 && fill >= 0.7) ? ks : 0;
}

int evf_conv3_b3n_launch(const float* src, int lds, const void* wp, const float* bias, float* out, int ldo, int B, int H, int W,
                         int K, int N, int flip, int accumulate, int ksplit, hipStream_t st) {
  NsGeo g;
  g.B = B, g.H = H, g.W = W, g.K = K, g.N = N, g.lds = lds, g.ldo = ldo, g.flip = flip;
  g.tiles_y = evf_cdiv(H, N_ROWS), g.tiles_x = evf_cdiv(W, N_COLS);
  const int ntiles = evf_cdiv(N, 32), nchunk = evf_cdiv(ntiles, N_NC);
  g.nc = evf_cdiv(ntiles, nchunk);  // balanced chunks (9 tiles: 5 + 4, not 6 + 3)
  const int ntile = B * g.tiles_y * g.tiles_x, gx = 8 * evf_cdiv(ntile, 8);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)k_conv3_b3n, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N_LDS);
    once = true;
  }
  hipLaunchKernelGGL(k_conv3_b3n, dim3(gx, evf_cdiv(ntiles, g.nc), ksplit), dim3(512), N_LDS, st, src, (const uint4*)wp, bias, out, g,
                     accumulate, ksplit);
  return evf_status();
}
