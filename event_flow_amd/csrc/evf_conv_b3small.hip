// 3x3 stride-1 forward convolution of layers whose input is spike-valued BY CONSTRUCTION (hip_ops.spike_tag: spikes, small sums of
// spikes, their bilinear x2 blends -- exactly representable in bf16): the recurrent / residual cells and the decoders of the
// spiking EV-FlowNet (reference models/unet.py:418-465, models/spiking_submodules.py:878-1013).
//
// Why a kernel of its own.  The voting kernels (evf_conv_b3gen.hip, evf_conv_b3tile.hip) must assume real-valued input: the tiled
// one stages 16 channels at a time as THREE bf16 planes (split on the vector unit) behind two barriers per 108 MFMAs of a wave,
// the gather kernel re-reads and re-converts every pixel per tap.  With the promise the input is ONE bf16 plane:
//   block     512 threads = 8 waves, 512 output pixels = 16 M tiles (a wave owns two, sharing every weight fragment: 8 LDS fragment
//             reads per 12 MFMAs), 32 NT output channels, a range of 64-channel input groups (blockIdx.z = the K split);
//             GEOM 0: the pixels are TWO whole 16 x 16 images (the 512-channel layers: 2048 x 512 x 4608 at batch 8, where
//             spatial tiles are half empty), GEOM 1: one 16 x 32 tile of a larger image
//   LDS       the halo pixels of a 64-channel group as one plane (144 B per pixel: conflict-free ds_read_b128), 88-91 KiB, staged
//             once per group and reused by the nine taps: a group is 9 x 48 MFMAs per wave between A-tile barriers (the tiled
//             kernel: 108); the weights of one tap (4 chunks x NT x 3 planes x 1 KiB) double-buffered through registers
//   head      a decoder's input starts with its two real-valued flow channels (exact_from = 4): channels 0..15 run FIRST as the
//             exact 3-way split (three planes, the six products above 2^-24), through the same LDS, by the K split 0
//   output    raw partial sums: [split][pixel][N] slabs (added in index order by k_b3_reduce or by the neuron kernel,
//             evf_lif_fwd_parts), or the output tensor itself when the contraction is not split; weights are the A operand, so a
//             lane owns a pixel and stores float4s
//   guard     an input value behind exact_from that is NOT exactly representable in bf16 (the caller's promise broken) turns the
//             block's output into NaN -- never a silently rounded product.
// Measured (LIF-EV-FlowNet step, B = 8): the five 512 -> 512 layers at 16 x 16 75 -> 35 us each (GEOM 0).
#include "evf_common.h"
#include "evf_split.h"

typedef float s_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t s_u32x4 __attribute__((ext_vector_type(4)));
typedef float s_f32x4 __attribute__((ext_vector_type(4)));

#define SX_KB 64                        // input channels per group
#define SX_PITCH (SX_KB * 2 + 16)       // 144 bytes per halo pixel of the exact plane
#define SX_APITCH 48                    // bytes per halo pixel and plane of the 16-channel head (3 planes)
#define SX_CHUNK (3 * 64)               // uint4 per (N tile, tap, 16-channel chunk) of the packed weights: [term 3][lane 64]
#define SX_WBYTES (9 * 2 * SX_CHUNK * 16)  // weight region: the head's nine taps x 2 N tiles x one chunk = 54 KiB (>= 2 tap stages)

struct SxGeo {
  int B, H, W, K, N, lds, ldo;  // ldo: pixel stride of dst (N for slabs)
  int tiles_y, tiles_x;         // GEOM 1
  int chunk0, nchunk;           // exact chunks [chunk0, chunk0 + nchunk) of 16 channels; chunk0 = 1: the head chunk 0 runs first
  int gper;                     // 64-channel groups per K split
  long slab;                    // floats between the K splits' outputs (0: unsplit)
};

template <int GEOM, int NT>
__global__ __launch_bounds__(512) void k_conv3_b3x(const float* __restrict__ src, const uint4* __restrict__ wp, float* __restrict__ dst,
                                                   SxGeo g) {
  constexpr int HPW = GEOM == 0 ? 18 : 34;                    // halo row length
  constexpr int NPX = GEOM == 0 ? 2 * 18 * 18 : 18 * 34;      // halo pixels of the block: 648 / 612
  constexpr int ATASKS = NPX * (SX_KB / 4), AITER = (ATASKS + 511) / 512;  // float4 loads per group: 21 / 20
  constexpr int WSTAGE = NT * 4 * SX_CHUNK;                   // uint4 per tap stage
  constexpr int WJ = (WSTAGE + 511) / 512;                    // 3 / 2 (NT = 1: 768 uint4, the second trip half used)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_a = smem;
  uint4* s_w = (uint4*)(smem + NPX * SX_PITCH);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, p = lane & 31, kg = lane >> 5;
  const int nb = blockIdx.y, ks = blockIdx.z;
  int b0, y0 = 0, x0 = 0;
  if (GEOM == 0) {
    b0 = 2 * (int)blockIdx.x;
  } else {
    const int t = blockIdx.x, txi = t % g.tiles_x, t1 = t / g.tiles_x;
    b0 = t1 / g.tiles_y, y0 = (t1 % g.tiles_y) * 16, x0 = txi * 32;
  }
  const int KC = (g.K + 15) >> 4, G = (KC + 3) >> 2, ntiles = (g.N + 31) >> 5;
  const long wtile = (long)(9 * G) * 4 * SX_CHUNK;  // uint4 per N tile of the packed weights
  // halo pixel hp of the block -> (valid, source pixel offset in floats)
  auto hsrc = [&](int hp, bool& in) -> long {
    if (GEOM == 0) {
      const int img = hp >= 324 ? 1 : 0, q = hp - img * 324, hy = q / 18, hx = q - hy * 18;
      in = hy >= 1 && hy <= 16 && hx >= 1 && hx <= 16;
      const int sy = min(max(hy - 1, 0), 15), sx = min(max(hx - 1, 0), 15);
      return ((long)(b0 + img) * 256 + sy * 16 + sx) * g.lds;
    }
    const int hy = hp / 34, hx = hp - hy * 34, sy = y0 + hy - 1, sx = x0 + hx - 1;
    in = sy >= 0 && sy < g.H && sx >= 0 && sx < g.W;
    return (((long)b0 * g.H + min(max(sy, 0), g.H - 1)) * g.W + min(max(sx, 0), g.W - 1)) * g.lds;
  };

  s_f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
  // this lane's pixel of the wave's two M tiles (halo pixel index of its (0, 0) tap)
  int hpix[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int mt = 2 * wv + m;
    if (GEOM == 0) hpix[m] = (mt >> 3) * 324 + (2 * (mt & 7) + (p >> 4)) * 18 + (p & 15);
    else hpix[m] = mt * 34 + p;
  }
  int inexact = 0;

  // ---- the real-valued head: channels 0..15 as the exact 3-way split, six products per MFMA position (K split 0 only)
  if (g.chunk0 == 1 && ks == 0) {
    constexpr int HT = NPX * 4, HI = (HT + 511) / 512;  // float4 tasks of the 16-channel chunk
    constexpr int HPL = NPX * SX_APITCH;                // bytes per plane
#pragma unroll
    for (int j = 0; j < HI; ++j) {
      const int task = min(tid + 512 * j, HT - 1), hp = task >> 2, q = task & 3;
      bool in;
      const long off = hsrc(hp, in);
      const int c = 4 * q;
      const s_f32x4 v = *(const s_f32x4*)(src + off + (c + 4 <= g.K ? c : 0));
      const s_f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const s_f32x4 a = (in && c + 4 <= g.K) ? v : z4;
      uint32_t h0, m0, l0, h1, m1, l1;
      evf_split3_pair(a.x, a.y, h0, m0, l0);
      evf_split3_pair(a.z, a.w, h1, m1, l1);
      if (tid + 512 * j < HT) {
        char* d = s_a + hp * SX_APITCH + q * 8;
        *(uint2*)(d) = make_uint2(h0, h1);
        *(uint2*)(d + HPL) = make_uint2(m0, m1);
        *(uint2*)(d + 2 * HPL) = make_uint2(l0, l1);
      }
    }
    // all nine taps' weights of chunk 0: [tap][N tile][term][lane]
    constexpr int HW = 9 * NT * SX_CHUNK;
    for (int idx = tid; idx < HW; idx += 512) {
      const int tap = idx / (NT * SX_CHUNK), r = idx - tap * (NT * SX_CHUNK), t = r / SX_CHUNK, rem = r - t * SX_CHUNK;
      s_w[idx] = wp[(long)min(nb * NT + t, ntiles - 1) * wtile + ((long)tap * G * 4) * SX_CHUNK + rem];
    }
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - 3 * dy;
      s_bf16x8 xh[2], xm[2], xl[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const char* ap = s_a + (hpix[m] + dy * HPW + dx) * SX_APITCH + kg * 16;
        const uint4 a0 = *(const uint4*)ap, a1 = *(const uint4*)(ap + HPL), a2 = *(const uint4*)(ap + 2 * HPL);
        xh[m] = *(const s_bf16x8*)&a0, xm[m] = *(const s_bf16x8*)&a1, xl[m] = *(const s_bf16x8*)&a2;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const uint4* wq = s_w + (tap * NT + t) * SX_CHUNK + lane;
        const uint4 q0 = wq[0], q1 = wq[64], q2 = wq[128];
        const s_bf16x8 wh = *(const s_bf16x8*)&q0, wm = *(const s_bf16x8*)&q1, wl = *(const s_bf16x8*)&q2;
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
    __syncthreads();  // (the plane region and the weight region are rewritten by the groups below)
  }

  // ---- the exact groups of this K split
  const int ngroups = (g.nchunk + 3) >> 2;
  const int g_lo = min(ks * g.gper, ngroups), g_hi = min(g_lo + g.gper, ngroups);
  s_u32x4 pw[WJ];
  auto wfetch = [&](int grp, int tap) {  // the tap's weights of the group's chunks: [N tile][chunk 4][term][lane]
    const int c16 = g.chunk0 + 4 * grp;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int idx = min(tid + 512 * j, WSTAGE - 1), t = idx / (4 * SX_CHUNK), r = idx - t * (4 * SX_CHUNK), ch = r / SX_CHUNK,
                rem = r - ch * SX_CHUNK;
      const int cc = min(c16 + ch, KC - 1);  // (a chunk past the end re-reads the last one: its plane is zero)
      pw[j] = ((const s_u32x4*)wp)[(long)min(nb * NT + t, ntiles - 1) * wtile + ((long)tap * G * 4 + cc) * SX_CHUNK + rem];
    }
  };
  auto wcommit = [&](int stage) {
#pragma unroll
    for (int j = 0; j < WJ; ++j)
      if (tid + 512 * j < WSTAGE) ((s_u32x4*)s_w)[stage * WSTAGE + tid + 512 * j] = pw[j];
  };
#pragma unroll 1
  for (int grp = g_lo; grp < g_hi; ++grp) {
    const int k0 = (g.chunk0 + 4 * grp) * 16;
    wfetch(grp, 0);
    // the group's A tile: fp32 -> bf16, exactness checked
    constexpr int AU = GEOM == 0 ? 7 : 5;  // AITER = 21 = 3 x 7 / 20 = 4 x 5
#pragma unroll 1
    for (int j0 = 0; j0 < AITER; j0 += AU) {
      s_f32x4 v[AU];
      bool in[AU];
#pragma unroll
      for (int jj = 0; jj < AU; ++jj) {
        const int task = min(tid + 512 * (j0 + jj), ATASKS - 1), hp = task >> 4, q = task & 15;
        const long off = hsrc(hp, in[jj]);
        const int c = k0 + 4 * q;
        in[jj] = in[jj] && c + 4 <= g.K;
        v[jj] = *(const s_f32x4*)(src + off + (c + 4 <= g.K ? c : 0));
      }
#pragma unroll
      for (int jj = 0; jj < AU; ++jj) {
        const int task = tid + 512 * (j0 + jj), hp = task >> 4, q = task & 15;
        const s_f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const s_f32x4 a = in[jj] ? v[jj] : z4;
        const uint32_t h01 = evf_pk_bf16(a.x, a.y), h23 = evf_pk_bf16(a.z, a.w);
        inexact |= (int)(a.x != __uint_as_float(h01 << 16)) | (int)(a.y != __uint_as_float(h01 & 0xFFFF0000u)) |
                   (int)(a.z != __uint_as_float(h23 << 16)) | (int)(a.w != __uint_as_float(h23 & 0xFFFF0000u));
        if (task < ATASKS) *(uint2*)(s_a + hp * SX_PITCH + q * 8) = make_uint2(h01, h23);
      }
    }
    wcommit(0);
    __syncthreads();
    const int nch = min(4, g.nchunk - 4 * grp);  // chunks of this group that hold real channels (uniform)
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) wfetch(grp, tap + 1);
      const int dy = tap / 3, dx = tap - 3 * dy;
      const uint4* sw = s_w + (tap & 1) * WSTAGE + lane;
      const char* a0p = s_a + (hpix[0] + dy * HPW + dx) * SX_PITCH + kg * 16;
      const char* a1p = s_a + (hpix[1] + dy * HPW + dx) * SX_PITCH + kg * 16;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        if (ch < nch) {
          const uint4 x0q = *(const uint4*)(a0p + ch * 32), x1q = *(const uint4*)(a1p + ch * 32);
          const s_bf16x8 xa = *(const s_bf16x8*)&x0q, xb = *(const s_bf16x8*)&x1q;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const uint4 q0 = sw[((t * 4 + ch) * 3 + 0) * 64], q1 = sw[((t * 4 + ch) * 3 + 1) * 64], q2 = sw[((t * 4 + ch) * 3 + 2) * 64];
            const s_bf16x8 wh = *(const s_bf16x8*)&q0, wm = *(const s_bf16x8*)&q1, wl = *(const s_bf16x8*)&q2;
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xa, acc[0][t], 0, 0, 0);  // smallest terms first
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xb, acc[1][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xa, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xb, acc[1][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xa, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xb, acc[1][t], 0, 0, 0);
          }
        }
      }
      if (tap + 1 < 9) wcommit((tap + 1) & 1);  // (the other stage: last read one tap ago, before the previous trip's barrier)
      __syncthreads();                           // (after tap 8: every wave is done with the plane before the next group rewrites it)
    }
  }

  // ---- epilogue: raw sums; lane = pixel, channels n0 + 32 t + 8 q + 4 kg + e
  const int bad = __syncthreads_or(inexact);
  const float poison = bad ? __builtin_nanf("") : 0.f;
  float* out = dst + (long)ks * g.slab;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int mt = 2 * wv + m;
    long pix;
    bool ok;
    if (GEOM == 0) {
      pix = (long)(b0 + (mt >> 3)) * 256 + (2 * (mt & 7) + (p >> 4)) * 16 + (p & 15);
      ok = true;
    } else {
      const int oy = y0 + mt, ox = x0 + p;
      ok = oy < g.H && ox < g.W;
      pix = ((long)b0 * g.H + min(oy, g.H - 1)) * g.W + min(ox, g.W - 1);
    }
    float* o = out + pix * g.ldo + nb * (32 * NT) + 4 * kg;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb * (32 * NT) + 32 * t + 8 * q + 4 * kg;
        if (ok && n + 4 <= g.N)
          *(float4*)(o + 32 * t + 8 * q) = make_float4(acc[m][t][4 * q + 0] + poison, acc[m][t][4 * q + 1] + poison,
                                                       acc[m][t][4 * q + 2] + poison, acc[m][t][4 * q + 3] + poison);
      }
  }
}

// Plan for a 3x3 stride-1 forward product whose input channels from `exact_from` on are exactly representable in bf16.
// -> number of K splits (>= 1: the kernel takes the product; the caller provides that many slabs, or none for 1), 0 = no.
// slab_cap: slabs of B*H*W*N floats the caller's scratch holds.
struct SxPlan {
  int geom, NT, ksplit;
  SxGeo g;
};
static bool sx_plan(const float* src, int B, int H, int W, int K, int N, int lds, int exact_from, long slab_cap, SxPlan& p) {
  if (exact_from < 0 || exact_from > 16 || (K & 3) || (lds & 3) || (N & 3) || (((uintptr_t)src) & 15) || K < 32) return false;
  SxGeo& g = p.g;
  g.B = B, g.H = H, g.W = W, g.K = K, g.N = N, g.lds = lds;
  g.tiles_y = evf_cdiv(H, 16), g.tiles_x = evf_cdiv(W, 32);
  long tiles;
  if (H == 16 && W == 16 && !(B & 1)) {
    p.geom = 0, tiles = B / 2;
  } else {
    const double fill = (double)H * W / ((double)g.tiles_y * 16 * g.tiles_x * 32);
    if (fill < 0.7) return false;  // (mostly empty tiles: the gather kernel does better)
    p.geom = 1, tiles = (long)B * g.tiles_y * g.tiles_x;
  }
  p.NT = N > 32 ? 2 : 1;
  const int KC = evf_cdiv(K, 16);
  g.chunk0 = exact_from > 0 ? 1 : 0;
  g.nchunk = KC - g.chunk0;
  if (g.nchunk < 1) return false;
  const int ngroups = evf_cdiv(g.nchunk, 4);
  const long blocks = tiles * evf_cdiv(N, 32 * p.NT);
  long ks = blocks >= 200 ? 1 : evf_cdiv(256L, blocks);
  if (ks > ngroups) ks = ngroups;
  if (ks > 1 && ks > slab_cap) ks = slab_cap;
  if (ks < 1) ks = 1;
  if (blocks * ks < 96) return false;  // (cannot fill the chip: the general kernels' finer splits do better)
  g.gper = evf_cdiv(ngroups, (int)ks);
  p.ksplit = evf_cdiv(ngroups, g.gper);
  return true;
}

// Does the exact-input kernel take this product?  -> K splits (1: writes `out`-shaped raw sums itself), 0 = no
int evf_conv3_b3x_plan(const float* src, int B, int H, int W, int K, int N, int lds, int exact_from, long slab_cap) {
  SxPlan p;
  return sx_plan(src, B, H, W, K, N, lds, exact_from, slab_cap, p) ? p.ksplit : 0;
}

// dst: the output itself (pixel stride ldo) when the plan does not split, else ksplit slabs [B*H*W][N] (ldo = N)
int evf_conv3_b3x_launch(const float* src, int lds, const void* wp, float* dst, int ldo, int B, int H, int W, int K, int N,
                         int exact_from, long slab_cap, hipStream_t st) {
  SxPlan p;
  if (!sx_plan(src, B, H, W, K, N, lds, exact_from, slab_cap, p)) return EVF_EINVAL;
  SxGeo& g = p.g;
  g.ldo = p.ksplit > 1 ? N : ldo;
  g.slab = p.ksplit > 1 ? (long)B * H * W * N : 0;
  const int npx = p.geom == 0 ? 648 : 612;
  const size_t smem = (size_t)npx * SX_PITCH + SX_WBYTES;
  const dim3 grid(p.geom == 0 ? B / 2 : B * g.tiles_y * g.tiles_x, evf_cdiv(N, 32 * p.NT), p.ksplit);
#define SX_GO(G_, T_)                                                                                                       \
  do {                                                                                                                      \
    static bool once = false;                                                                                               \
    if (!once) {                                                                                                            \
      (void)hipFuncSetAttribute((const void*)k_conv3_b3x<G_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);   \
      once = true;                                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL((k_conv3_b3x<G_, T_>), grid, dim3(512), smem, st, src, (const uint4*)wp, dst, g);                    \
  } while (0)
  if (p.geom == 0 && p.NT == 2) SX_GO(0, 2);
  else if (p.geom == 0) SX_GO(0, 1);
  else if (p.NT == 2) SX_GO(1, 2);
  else SX_GO(1, 1);
#undef SX_GO
  return evf_status();
}
