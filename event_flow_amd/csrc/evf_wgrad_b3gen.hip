// Weight gradient of the general 3x3 stride-1 convolution on the bf16 matrix cores:
//   gw[co][ci][tap] = sum_px x[px (+) tap][ci] * g[px][co]
// with the contraction over PIXELS (v_mfma_f32_32x32x16_bf16: 16 pixels per instruction against 2 for the fp32 MFMA of
// k_wgrad9, evf_wgrad_gen.hip).  A spiking network's x is exactly representable in bf16 (binary spikes, event counts,
// {0,1,2} residual sums, bilinear blends of those) and g = hi + mid + lo exactly (three bf16 planes), so
//   sum x g = sum x g_hi + sum x g_mid + sum x g_lo      (three MFMAs of 32 cycles per 16 pixels, fp32 accumulation)
// is the fp32 result up to summation order: 5.3x fewer matrix cycles.
//
// A lane of a bf16 MFMA holds EIGHT consecutive k (pixels) of one channel, so both operands are transposed on their way
// into LDS ([channel][pixel] images, written by 2-byte stores while the tile is converted); the eight pixels of a lane
// are one row of the 8 x 8 pixel tile.  Taps with dx = 1 read their row 16-byte aligned; dx = 0 / 2 are one pixel off:
// one extra ds_read_b32 and four v_alignbit funnel shifts rebuild the fragment -- one x image serves all nine taps.
//
//   block   576 threads = 9 waves, one per tap: every tap has exactly one accumulator per block (no cross-wave reduction);
//           a block owns 32 CT input x 32 NT output channels and walks a range of pixel tiles; partial sums go to the slabs
//           [split][tap][ci][co] k_wgrad_reduce sums (same layout as k_wgrad9)
//   x not exactly representable (a decoder's two flow channels, analog networks): the block raises redo[ci tile] and the
//           caller re-runs the fp32 kernel for exactly those channel tiles (k_wgrad9 exits at once for the others).
#include <stdlib.h>

#include "evf_common.h"
#include "evf_split.h"

typedef float w_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 w_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wb_s16x4 __attribute__((ext_vector_type(4)));
typedef short wb_s16x8 __attribute__((ext_vector_type(8)));
#define WB_LDS __attribute__((address_space(3)))

#define WB_T 8                    // tile rows = tile columns
#define WB_HP ((WB_T + 2) * (WB_T + 2))  // 100 halo pixels
#define WB_ROWB 48                // bytes per halo row and channel: element 7 + hc, 16-byte aligned at hc = 1
#define WB_XP ((WB_T + 2) * WB_ROWB + 16)  // 496: channel pitch of the x image (conflict-free b128 fragment reads)
#define WB_GP (WB_T * WB_T * 2 + 16)       // 144: channel pitch of a g plane
// WB_TR (default): the LDS images stay in the NATURAL [pixel][channel] order -- a thread drops the four bf16 of its float4 with ONE
// 8-byte store (the [channel][pixel] images above take four 2-byte stores into four channel rows: 36 per thread and tile, with
// 8-way bank conflicts; a probe build without them: LIF-EV-FlowNet weight gradients 2.12 -> 1.72 ms per step) -- and the
// transposition happens in the READ: ds_read_b64_tr_b16 (gfx950) hands lane c of a 16-lane group column c of a [4 pixels][16
// channels] block, i.e. four consecutive pixels of one channel = half an MFMA fragment.  A pixel pitch of 64 (mod 256) bytes puts
// the 2 x 4 row pieces a half-wave reads on distinct bank slots; the tap's column shift is just another base address (no
// funnel shifts).
#ifndef WB_TR
#define WB_TR 1
#endif
#define WB_PPITCH(CH) ((CH) == 64 ? 192 : 64)  // bytes per pixel of a [pixel][CH channels] bf16 image

struct WgB3Geo {
  int B, H, W, Cin, Cout, ldx, ldg;  // H, W: the INPUT image
  int OH, OW;                        // the output image (= H, W at stride 1; (H - 1) / 2 + 1 at stride 2, padding 1)
  int tiles_x, tiles_y, tiles_per_split, n_ct;
  long ntiles;
};

__device__ __forceinline__ int wb_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Fused slab reduction (WgFuse.ticket != null; the caller guarantees an x that is exactly representable in bf16, so no fp32
// pass follows): the block that finishes LAST among the pixel splits of its weight tile -- one arrival ticket per tile, after a
// device-scope release of its slab -- sums the tile's partial sums over the splits in INDEX order (a fixed order: the same
// bits whatever the arrival order) and writes the torch layout [co][ci][tap] itself, tap by tap through an LDS transpose (slab
// reads coalesced along co, gradient writes running along ci).  k_wgrad_reduce's launch, its ~5 us boundary and the second HBM
// round trip of the slabs are gone.  MEASURED AND NOT THE DEFAULT (EVF_WGRAD_FUSE=1): the blocks of a launch finish together, so
// the last blocks' chains run exposed at the end of the kernel -- +47 us per call over the 25 us reduce launch they replace
// (evf_wgrad_gen.hip has the numbers).  An x that does break the promise is not silently rounded: NaN.
struct WgFuse {
  float* gw;     // [Cout][cin_total][3][3]
  int* ticket;   // one per weight tile (zero on entry, handed back zero)
  int cin_total, cin_off, accumulate;
  int promised;  // the caller promised an exact x (with or without the fused reduction): a violation poisons the block's slab
};

// STRIDE 2 (round 6; the encoders of the spiking EV-FlowNet, reference models/unet.py:335-353): the 8 x 8 tile is a tile of the OUTPUT
// (g) image, the x patch behind it is 17 x 17 input pixels (input pixel (2 oy + dy - 1, 2 ox + dx - 1) meets output pixel (oy, ox)), and
// a fragment's pixel run steps over every other pixel of a patch row: the transposing read takes an address PER LANE, so the stride is
// two pixel pitches instead of one -- the same kernel, nine waves = nine taps.  (Before: these layers ran the fp32-MFMA kernel
// k_wgrad9 at 0.07-0.09 of the bf16 peak.)
template <int CT, int NT, int STRIDE = 1>
__global__ __launch_bounds__(576) void k_wgrad9_b3(const float* __restrict__ x, const float* __restrict__ gy,
                                                   float* __restrict__ slab, float* __restrict__ gbias, int* __restrict__ redo,
                                                   WgB3Geo g, WgFuse fz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XQ = 8 * CT, GQ = 8 * NT;                 // float4 per pixel
  constexpr int HS = (WB_T - 1) * STRIDE + 3;             // edge of the x patch of a tile: 10 (stride 1) / 17 (stride 2)
  constexpr int XHP = HS * HS;                            // its pixels
  static_assert(STRIDE == 1 || WB_TR, "stride 2: transposing-read layout only");
  constexpr int NLX = (XHP * XQ + 575) / 576;             // x float4 loads per thread
  constexpr int NLG = (WB_T * WB_T * GQ + 575) / 576;     // g float4 loads per thread
  constexpr int GPL = 32 * NT * WB_GP;                    // bytes per g plane
#if WB_TR
  constexpr int XPP = WB_PPITCH(32 * CT), GPP = WB_PPITCH(32 * NT);  // pixel pitches
  constexpr int GPL_TR = WB_T * WB_T * GPP;                             // bytes per g plane
  char* s_x = smem;                                       // [XHP patch pixels][XPP]
  char* s_g = smem + XHP * XPP;                           // [3 planes][64 pixels][GPP]
  float* s_b = (float*)(s_g + 3 * GPL_TR);                // [32 NT] bias partial sums
#else
  char* s_x = smem;                                       // [32 CT channels][WB_XP]
  char* s_g = smem + 32 * CT * WB_XP;                     // [3 planes][32 NT channels][WB_GP]
  float* s_b = (float*)(s_g + 3 * GPL);                   // [32 NT] bias partial sums
#endif
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane & 31, kg = lane >> 5;
  const int cit = blockIdx.y % g.n_ct, cot = blockIdx.y / g.n_ct;
  const int ci0 = cit * 32 * CT, co0 = cot * 32 * NT;
  const bool do_bias = gbias && cit == 0;
  if (tid < 32 * NT) s_b[tid] = 0.f;

  w_f32x16 acc[CT][NT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

  float4 xr[NLX], gr[NLG];
  float bs[4] = {0.f, 0.f, 0.f, 0.f};  // this thread's output-channel quad (fixed: 576 % GQ == 0)
  int inexact = 0;
  int n_tx, n_ty, n_b;
  {
    const long t0 = (long)blockIdx.x * g.tiles_per_split;
    n_tx = (int)(t0 % g.tiles_x);
    const long t1 = t0 / g.tiles_x;
    n_ty = (int)(t1 % g.tiles_y), n_b = (int)(t1 / g.tiles_y);
  }
  // (no uniform branches around the tile loads: a load under a branch is followed by s_waitcnt vmcnt(0))
  auto prefetch = [&](bool want) {
    const bool tok = want && n_b < g.B;
    const int b = min(n_b, g.B - 1);
    const int y0 = n_ty * WB_T, x0 = n_tx * WB_T;
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int idx = tid + 576 * i, hp = idx / XQ, q = idx - hp * XQ;
      const int hr = hp / HS, hc = hp - hr * HS;
      const int sy = STRIDE * y0 + hr - 1, sx = STRIDE * x0 + hc - 1, c = ci0 + 4 * q;
      const bool ok = tok && hp < XHP && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W && c + 4 <= g.Cin;
      const float4 v = *(const float4*)(ok ? x + (((long)b * g.H + sy) * g.W + sx) * g.ldx + c : x);
      xr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NLG; ++i) {
      const int idx = tid + 576 * i, px = idx / GQ, q = idx - px * GQ;
      const int oy = y0 + (px >> 3), ox = x0 + (px & 7), c = co0 + 4 * q;
      const bool ok = tok && px < WB_T * WB_T && oy < g.OH && ox < g.OW && c + 4 <= g.Cout;
      const float4 v = *(const float4*)(ok ? gy + (((long)b * g.OH + oy) * g.OW + ox) * g.ldg + c : gy);
      gr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (++n_tx == g.tiles_x) {
      n_tx = 0;
      if (++n_ty == g.tiles_y) n_ty = 0, ++n_b;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int idx = tid + 576 * i, hp = idx / XQ, q = idx - hp * XQ;
      if (hp >= XHP) continue;
      const int hr = hp / HS, hc = hp - hr * HS;
      const uint32_t h01 = evf_pk_bf16(xr[i].x, xr[i].y), h23 = evf_pk_bf16(xr[i].z, xr[i].w);
      // exactly representable?  (else this channel tile is redone in fp32)
      inexact |= (int)(xr[i].x != __uint_as_float(h01 << 16)) | (int)(xr[i].y != __uint_as_float(h01 & 0xFFFF0000u)) |
                 (int)(xr[i].z != __uint_as_float(h23 << 16)) | (int)(xr[i].w != __uint_as_float(h23 & 0xFFFF0000u));
#if WB_TR
      (void)hr, (void)hc;
      *(uint2*)(s_x + hp * XPP + q * 8) = make_uint2(h01, h23);
#else
      char* p = s_x + (4 * q) * WB_XP + hr * WB_ROWB + (7 + hc) * 2;
      *(uint16_t*)(p) = (uint16_t)h01;
      *(uint16_t*)(p + WB_XP) = (uint16_t)(h01 >> 16);
      *(uint16_t*)(p + 2 * WB_XP) = (uint16_t)h23;
      *(uint16_t*)(p + 3 * WB_XP) = (uint16_t)(h23 >> 16);
#endif
    }
#pragma unroll
    for (int i = 0; i < NLG; ++i) {
      const int idx = tid + 576 * i, px = idx / GQ, q = idx - px * GQ;
      if (px >= WB_T * WB_T) continue;
      uint32_t h0, m0, l0, h1, m1, l1;
      evf_split3_pair(gr[i].x, gr[i].y, h0, m0, l0);
      evf_split3_pair(gr[i].z, gr[i].w, h1, m1, l1);
      bs[0] += gr[i].x, bs[1] += gr[i].y, bs[2] += gr[i].z, bs[3] += gr[i].w;
#if WB_TR
      {
        char* pt = s_g + px * GPP + q * 8;
        *(uint2*)(pt) = make_uint2(h0, h1);
        *(uint2*)(pt + GPL_TR) = make_uint2(m0, m1);
        *(uint2*)(pt + 2 * GPL_TR) = make_uint2(l0, l1);
        continue;
      }
#endif
      char* p = s_g + (4 * q) * WB_GP + px * 2;
      *(uint16_t*)(p) = (uint16_t)h0, *(uint16_t*)(p + WB_GP) = (uint16_t)(h0 >> 16);
      *(uint16_t*)(p + 2 * WB_GP) = (uint16_t)h1, *(uint16_t*)(p + 3 * WB_GP) = (uint16_t)(h1 >> 16);
      *(uint16_t*)(p + GPL) = (uint16_t)m0, *(uint16_t*)(p + GPL + WB_GP) = (uint16_t)(m0 >> 16);
      *(uint16_t*)(p + GPL + 2 * WB_GP) = (uint16_t)m1, *(uint16_t*)(p + GPL + 3 * WB_GP) = (uint16_t)(m1 >> 16);
      *(uint16_t*)(p + 2 * GPL) = (uint16_t)l0, *(uint16_t*)(p + 2 * GPL + WB_GP) = (uint16_t)(l0 >> 16);
      *(uint16_t*)(p + 2 * GPL + 2 * WB_GP) = (uint16_t)l1, *(uint16_t*)(p + 2 * GPL + 3 * WB_GP) = (uint16_t)(l1 >> 16);
    }
  };

  const int dy = wv / 3, dx = wv - 3 * dy;  // this wave's tap
  prefetch(true);
#pragma unroll 1
  for (int it = 0; it < g.tiles_per_split; ++it) {
#ifdef WB_PROBE_NOCOMMIT  // (A/B probe, never shipped: what the transposing LDS stores cost -- the first tile's image is reused)
    if (it == 0)
#endif
    commit();
    __syncthreads();
    prefetch(it + 1 < g.tiles_per_split);
#if WB_TR
    // transpose reads: lane i16 = lane & 15 of a 16-lane group addresses pixel (i16 >> 2) of a 4-pixel run and the 4-channel piece
    // (i16 & 3) of the group's 16 channels ((lane >> 4) & 1: lower / upper half of the 32-channel tile) and RECEIVES the four
    // pixels of channel (lane & 31); two reads (pixels 0..3, 4..7 of the tile row) make the fragment
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int lane_x = (i16 >> 2) * STRIDE * XPP + (g16 * 16 + (i16 & 3) * 4) * 2;
    const int lane_g = (i16 >> 2) * GPP + (g16 * 16 + (i16 & 3) * 4) * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int r = 2 * ks + kg;
      w_bf16x8 xa[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const char* pp = s_x + ((STRIDE * r + dy) * HS + dx) * XPP + c * 64 + lane_x;
        const wb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WB_LDS wb_s16x4*)pp);
        const wb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WB_LDS wb_s16x4*)(pp + 4 * STRIDE * XPP));
        const wb_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        xa[c] = *(const w_bf16x8*)&v;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        w_bf16x8 gq[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const char* pp = s_g + pl * GPL_TR + (r * WB_T) * GPP + t * 64 + lane_g;
          const wb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WB_LDS wb_s16x4*)pp);
          const wb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WB_LDS wb_s16x4*)(pp + 4 * GPP));
          const wb_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          gq[pl] = *(const w_bf16x8*)&v;
        }
#ifdef WB_PROBE_NOMFMA  // (A/B probe, never shipped: the operand reads stay, the matrix pipe is idle)
#pragma unroll
        for (int c = 0; c < CT; ++c) asm volatile("" ::"v"(xa[c]), "v"(gq[0]), "v"(gq[1]), "v"(gq[2]));
#else
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[c], gq[2], acc[c][t], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[c], gq[1], acc[c][t], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[c], gq[0], acc[c][t], 0, 0, 0);
#endif
      }
    }
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // 16 pixels = tile rows 2 ks, 2 ks + 1 (lane half kg)
      const int r = 2 * ks + kg;
      w_bf16x8 xa[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const char* p = s_x + (c * 32 + row) * WB_XP + (r + dy) * WB_ROWB;  // element 7 + hc at byte 14 + 2 hc
        const uint4 q = *(const uint4*)(p + 16);                             // elements 8 .. 15 (hc 1 .. 8)
        uint4 o;
        if (dx == 1) {
          o = q;
        } else if (dx == 0) {  // elements 7 .. 14
          const uint32_t d = *(const uint32_t*)(p + 12);
          o.x = __builtin_amdgcn_alignbit(q.x, d, 16), o.y = __builtin_amdgcn_alignbit(q.y, q.x, 16);
          o.z = __builtin_amdgcn_alignbit(q.z, q.y, 16), o.w = __builtin_amdgcn_alignbit(q.w, q.z, 16);
        } else {  // elements 9 .. 16
          const uint32_t d = *(const uint32_t*)(p + 32);
          o.x = __builtin_amdgcn_alignbit(q.y, q.x, 16), o.y = __builtin_amdgcn_alignbit(q.z, q.y, 16);
          o.z = __builtin_amdgcn_alignbit(q.w, q.z, 16), o.w = __builtin_amdgcn_alignbit(d, q.w, 16);
        }
        xa[c] = *(const w_bf16x8*)&o;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const char* p = s_g + (t * 32 + row) * WB_GP + (16 * ks + 8 * kg) * 2;
        const uint4 q0 = *(const uint4*)p, q1 = *(const uint4*)(p + GPL), q2 = *(const uint4*)(p + 2 * GPL);
        const w_bf16x8 gh = *(const w_bf16x8*)&q0, gm = *(const w_bf16x8*)&q1, gl = *(const w_bf16x8*)&q2;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[c], gl, acc[c][t], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[c], gm, acc[c][t], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[c], gh, acc[c][t], 0, 0, 0);
      }
    }
#endif
    __syncthreads();
  }

  // an x the caller promised to be exact and that is not: this block's partial sums become NaN (whoever reduces them propagates it)
  const int blk_inexact = __syncthreads_or(inexact);
  const float poison = (fz.promised && blk_inexact) ? __builtin_nanf("") : 0.f;
  // epilogue: slab[split][tap][ci][co] (co fastest); wave = tap
  float* sl = slab + (long)blockIdx.x * 9 * g.Cin * g.Cout;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = co0 + t * 32 + row;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + c * 32 + wb_row(r, lane);
        if (ci < g.Cin && co < g.Cout) sl[((long)wv * g.Cin + ci) * g.Cout + co] = acc[c][t][r] + poison;
      }
    }
  if (blk_inexact && tid == 0 && !fz.promised) atomicOr(redo + cit, 1);
  if (do_bias) {
    const int q = tid % GQ;
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(s_b + 4 * q + j, bs[j]);
    __syncthreads();
    if (tid < 32 * NT && co0 + tid < g.Cout) evf_atomic_add(gbias + co0 + tid, s_b[tid]);
  }
  if (!fz.ticket) return;
  // ---- fused reduction: last block of the tile
  // (no static __shared__ in this kernel: statics precede the dynamic region unpadded, and a 4-byte one would take the 16-byte
  //  alignment the b128 / transpose reads need away from it -- the flag lives behind the bias sums)
  int& s_last = *(int*)(s_b + 32 * NT);
  // release: every thread waits for ITS slab stores (workgroup scope: a wait, no cache maintenance), the barrier collects them, and
  // ONE thread makes the block's stores visible device-wide (the agent-scope fence writes the XCD's L2 back: as 576 per-thread
  // fences it cost ~170 us per launch) before it draws the ticket
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const int t = atomicAdd(fz.ticket + blockIdx.y, 1 + (blk_inexact ? 0x10000 : 0));
    const bool last = (t & 0xFFFF) + 1 == (int)gridDim.x;
    s_last = last ? (1 | (((t >> 16) != 0 || blk_inexact) ? 2 : 0)) : 0;
    if (last) {
      fz.ticket[blockIdx.y] = 0;  // (calls are stream-ordered: the next one finds its tickets at zero)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // acquire: the other splits' slabs (invalidates this CU's L1 / stale L2 lines)
    }
  }
  __syncthreads();
  if (!s_last) return;
  const bool tile_bad = (s_last & 2) != 0;
  float* tile = (float*)smem;  // [32 CT][32 NT + 1]
  constexpr int NE = 32 * CT * 32 * NT, NJ = (NE + 575) / 576, TP = 32 * NT + 1;
  const long per = (long)9 * g.Cin * g.Cout;
  const int nsplit = (int)gridDim.x;
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = tid + 576 * j, co_l = idx % (32 * NT), ci_l = idx / (32 * NT);
      const int ci = ci0 + ci_l, co = co0 + co_l;
      const bool ok = idx < NE && ci < g.Cin && co < g.Cout;
      const float* __restrict__ p = slab + ((long)tap * g.Cin + (ok ? ci : 0)) * g.Cout + (ok ? co : 0);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int k = 0;
      for (; k + 3 < nsplit; k += 4) {  // four independent loads in flight; the association is fixed
        a0 += p[(long)k * per];
        a1 += p[(long)(k + 1) * per];
        a2 += p[(long)(k + 2) * per];
        a3 += p[(long)(k + 3) * per];
      }
      for (; k < nsplit; ++k) a0 += p[(long)k * per];
      if (idx < NE) tile[ci_l * TP + co_l] = ok ? (a0 + a1) + (a2 + a3) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = tid + 576 * j, ci_l = idx % (32 * CT), co_l = idx / (32 * CT);
      const int ci = ci0 + ci_l, co = co0 + co_l;
      if (idx < NE && ci < g.Cin && co < g.Cout && fz.cin_off + ci < fz.cin_total) {
        float* d = fz.gw + ((long)co * fz.cin_total + fz.cin_off + ci) * 9 + tap;
        const float v = tile_bad ? __builtin_nanf("") : tile[ci_l * TP + co_l];
        *d = fz.accumulate ? *d + v : v;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_wgrad9_b3v (round 6): the same product, slabs and accumulation order as k_wgrad9_b3 at stride 1, as TWO TEAMS.
//
// What k_wgrad9_b3 loses (probe builds): nine tap waves on four SIMDs (the SIMD with three of them sets the pace: 3 x 48
// MFMAs per tile against 108 per SIMD when balanced), every tap wave re-reads the tile's g fragments (64 transposing reads per 48
// MFMAs: the LDS pipe is busier than the matrix pipe), and the conversion + LDS stores of the next tile sit BETWEEN two barriers
// with the matrix pipe idle.  Here:
//   waves 0..3  MATRIX team, one wave per SIMD: wave w owns ONE 32 ci x 32 co weight tile of the block's CTB x NTB = 4 and ALL nine
//               taps of it (nine accumulators, 144 registers): 108 MFMAs per 8 x 8 pixel tile and wave -- balanced --; a g fragment
//               is read once for the nine taps, an x fragment (patch row, column shift dx) once for the up to two (pixel row, dy)
//               pairs that meet it: 78 transposing reads per 108 MFMAs;
//   waves 4..7  LOADER team: float4 loads of the tile after next, bf16 conversion / exact 3-way split and the LDS stores of the next
//               tile into the OTHER half of a double buffer, behind the matrix team's MFMAs: ONE barrier per tile.
// Per accumulator the products arrive in k_wgrad9_b3's order (pixel tiles ascending, pixel rows 2 ks + kg ascending, lo / mid / hi),
// so the slabs are bit-identical.  Blocks and slabs as before (one block per CU, <= 256 blocks).
// ---------------------------------------------------------------------------------------------------------------------
#define WV_PITCH(CH) ((CH) * 2 <= 64 ? 64 : (CH) * 2 + 64)  // bytes per pixel of a [pixel][CH] bf16 image: 64 (mod 256)
#define WV_HS (WB_T + 2)
#define WV_XHP (WV_HS * WV_HS)

__device__ __forceinline__ w_bf16x8 wv_frag(const char* p, int half) {  // two transposing reads: pixels 0..3, 4..7 of the row
  const wb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WB_LDS wb_s16x4*)p);
  const wb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WB_LDS wb_s16x4*)(p + half));
  const wb_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return *(const w_bf16x8*)&v;
}

template <int CTB, int NTB>
__global__ __launch_bounds__(512) void k_wgrad9_b3v(const float* __restrict__ x, const float* __restrict__ gy,
                                                    float* __restrict__ slab, float* __restrict__ gbias, int* __restrict__ redo,
                                                    WgB3Geo g, int promised) {
  static_assert(CTB * NTB == 4, "four matrix waves = four 32 x 32 weight tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XQ = 8 * CTB, GQ = 8 * NTB;                       // float4 per pixel
  constexpr int XPP = WV_PITCH(32 * CTB), GPP = WV_PITCH(32 * NTB);
  constexpr int XB = WV_XHP * XPP, GPL = WB_T * WB_T * GPP, BUF = XB + 3 * GPL;  // bytes: x image, one g plane, one buffer
  constexpr int NLX = (WV_XHP * XQ + 255) / 256, NLG = (WB_T * WB_T * GQ) / 256;
  static_assert((WB_T * WB_T * GQ) % 256 == 0 && 256 % GQ == 0, "loader thread <-> output-channel quad");
  float* s_b = (float*)(smem + 2 * BUF);                          // [32 NTB] bias partial sums
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cit = blockIdx.y % g.n_ct, cot = blockIdx.y / g.n_ct;
  const int ci0 = cit * 32 * CTB, co0 = cot * 32 * NTB;
  const bool do_bias = gbias && cit == 0;
  if (tid < 32 * NTB) s_b[tid] = 0.f;
  const int ntile = g.tiles_per_split;
  int inexact = 0;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  w_f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (wv >= 4) {
    // ---------------- loader team
    const int lt = tid - 256;
    float4 xA[NLX], gA[NLG], xB[NLX], gB[NLG];  // two tiles of loads in flight (a tile's loads get two trips to land)
    int n_tx, n_ty, n_b;
    {
      const long t0 = (long)blockIdx.x * g.tiles_per_split;
      n_tx = (int)(t0 % g.tiles_x);
      const long t1 = t0 / g.tiles_x;
      n_ty = (int)(t1 % g.tiles_y), n_b = (int)(t1 / g.tiles_y);
    }
    // (no uniform branches around the loads: a load under a branch is followed by s_waitcnt vmcnt(0))
    auto prefetch = [&](float4 (&xr)[NLX], float4 (&gr)[NLG], bool want) {
      const bool tok = want && n_b < g.B;
      const int b = min(n_b, g.B - 1);
      const int y0 = n_ty * WB_T, x0 = n_tx * WB_T;
#pragma unroll
      for (int i = 0; i < NLX; ++i) {
        const int idx = lt + 256 * i, hp = idx / XQ, q = idx - hp * XQ;
        const int hr = hp / WV_HS, hc = hp - hr * WV_HS;
        const int sy = y0 + hr - 1, sx = x0 + hc - 1, c = ci0 + 4 * q;
        const bool ok = tok && hp < WV_XHP && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W && c + 4 <= g.Cin;
        const float4 v = *(const float4*)(ok ? x + (((long)b * g.H + sy) * g.W + sx) * g.ldx + c : x);
        xr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < NLG; ++i) {
        const int idx = lt + 256 * i, px = idx / GQ, q = idx - px * GQ;
        const int oy = y0 + (px >> 3), ox = x0 + (px & 7), c = co0 + 4 * q;
        const bool ok = tok && oy < g.OH && ox < g.OW && c + 4 <= g.Cout;
        const float4 v = *(const float4*)(ok ? gy + (((long)b * g.OH + oy) * g.OW + ox) * g.ldg + c : gy);
        gr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (++n_tx == g.tiles_x) {
        n_tx = 0;
        if (++n_ty == g.tiles_y) n_ty = 0, ++n_b;
      }
    };
    auto commit = [&](const float4 (&xr)[NLX], const float4 (&gr)[NLG], int buf) {
      char* s_x = smem + buf * BUF;
      char* s_g = s_x + XB;
#pragma unroll
      for (int i = 0; i < NLX; ++i) {
        const int idx = lt + 256 * i, hp = idx / XQ, q = idx - hp * XQ;
        if (hp >= WV_XHP) continue;
        const uint32_t h01 = evf_pk_bf16(xr[i].x, xr[i].y), h23 = evf_pk_bf16(xr[i].z, xr[i].w);
        inexact |= (int)(xr[i].x != __uint_as_float(h01 << 16)) | (int)(xr[i].y != __uint_as_float(h01 & 0xFFFF0000u)) |
                   (int)(xr[i].z != __uint_as_float(h23 << 16)) | (int)(xr[i].w != __uint_as_float(h23 & 0xFFFF0000u));
        *(uint2*)(s_x + hp * XPP + q * 8) = make_uint2(h01, h23);
      }
#pragma unroll
      for (int i = 0; i < NLG; ++i) {
        const int idx = lt + 256 * i, px = idx / GQ, q = idx - px * GQ;
        uint32_t h0, m0, l0, h1, m1, l1;
        evf_split3_pair(gr[i].x, gr[i].y, h0, m0, l0);
        evf_split3_pair(gr[i].z, gr[i].w, h1, m1, l1);
        bs[0] += gr[i].x, bs[1] += gr[i].y, bs[2] += gr[i].z, bs[3] += gr[i].w;
        char* pt = s_g + px * GPP + q * 8;
        *(uint2*)(pt) = make_uint2(h0, h1);
        *(uint2*)(pt + GPL) = make_uint2(m0, m1);
        *(uint2*)(pt + 2 * GPL) = make_uint2(l0, l1);
      }
    };
    // tile k travels in register set A (k even) / B (k odd) and lands in buffer k & 1
    prefetch(xA, gA, true);
    prefetch(xB, gB, ntile > 1);
    commit(xA, gA, 0);
    prefetch(xA, gA, ntile > 2);
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < ntile; it += 2) {
      // trip `it` (even): the matrix team reads buffer 0; tile it + 1 -> buffer 1 (last read in trip it - 1, before that trip's barrier)
#ifndef WV_PROBE_NOCOMMIT  // (probe builds, never shipped: what do the loader team's LDS stores / loads cost the matrix team?)
      if (it + 1 < ntile) commit(xB, gB, 1);
#endif
#ifndef WV_PROBE_NOLOAD
      prefetch(xB, gB, it + 3 < ntile);
#endif
      __syncthreads();
      if (it + 1 >= ntile) break;
      // trip it + 1 (odd): tile it + 2 -> buffer 0
#ifndef WV_PROBE_NOCOMMIT
      if (it + 2 < ntile) commit(xA, gA, 0);
#endif
#ifndef WV_PROBE_NOLOAD
      prefetch(xA, gA, it + 4 < ntile);
#endif
      __syncthreads();
    }
  } else {
    // ---------------- matrix team: wave = weight tile (c, t), nine taps
    const int c = wv / NTB, t = wv - c * NTB, kg = lane >> 5;
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    // transposing reads (see k_wgrad9_b3): lane i16 of a 16-lane group addresses pixel (i16 >> 2) of a 4-pixel run and the 4-channel
    // piece (i16 & 3) of the group's 16 channels and RECEIVES the four pixels of channel (lane & 31)
    const int off_x = (kg * WV_HS) * XPP + c * 64 + (i16 >> 2) * XPP + (g16 * 16 + (i16 & 3) * 4) * 2;
    const int off_g = (kg * WB_T) * GPP + t * 64 + (i16 >> 2) * GPP + (g16 * 16 + (i16 & 3) * 4) * 2;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < ntile; ++it) {
      const char* bx = smem + (it & 1) * BUF + off_x;
      const char* bg = smem + (it & 1) * BUF + XB + off_g;
      w_bf16x8 gq[2][3];
#pragma unroll
      for (int j = 0; j < 9; ++j) {  // patch row j + kg meets pixel row 2 ks + kg at dy = j - 2 ks
        w_bf16x8 xf[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xf[dx] = wv_frag(bx + (j * WV_HS + dx) * XPP, 4 * XPP);
        if (!(j & 1) && j < 8) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) gq[(j >> 1) & 1][pl] = wv_frag(bg + pl * GPL + (j * WB_T) * GPP, 4 * GPP);
        }
#pragma unroll
        for (int dy = 2; dy >= 0; --dy) {  // (the older pixel row first: its g fragments are the ones about to be replaced)
          const int k2 = j - dy;
          if (k2 < 0 || (k2 & 1) || k2 > 6) continue;
          const int sl = (k2 >> 1) & 1;
#ifdef WV_PROBE_NOMFMA  // (probe build: the operand reads stay, the matrix pipe is idle)
          asm volatile("" ::"v"(xf[0]), "v"(xf[1]), "v"(xf[2]), "v"(gq[sl][0]), "v"(gq[sl][1]), "v"(gq[sl][2]));
#else
#pragma unroll
          for (int pl = 2; pl >= 0; --pl)  // smallest terms first
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
              acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[dx], gq[sl][pl], acc[dy * 3 + dx], 0, 0, 0);
#endif
        }
      }
      __syncthreads();
    }
  }

  const int blk_inexact = __syncthreads_or(inexact);
  const float poison = (promised && blk_inexact) ? __builtin_nanf("") : 0.f;
  if (wv < 4) {  // slab[split][tap][ci][co] (co fastest)
    const int c = wv / NTB, t = wv - c * NTB, row = lane & 31;
    float* sl = slab + (long)blockIdx.x * 9 * g.Cin * g.Cout;
    const int co = co0 + t * 32 + row;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + c * 32 + wb_row(r, lane);
        if (ci < g.Cin && co < g.Cout) sl[((long)tap * g.Cin + ci) * g.Cout + co] = acc[tap][r] + poison;
      }
  }
  if (blk_inexact && tid == 0 && !promised) atomicOr(redo + cit, 1);
  if (do_bias) {
    if (wv >= 4) {
      const int q = (tid - 256) % GQ;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(s_b + 4 * q + j, bs[j]);
    }
    __syncthreads();
    if (tid < 32 * NTB && co0 + tid < g.Cout) evf_atomic_add(gbias + co0 + tid, s_b[tid]);
  }
}

template <int CTB, int NTB>
static void wv_go(const float* x, const float* gy, float* slab, float* gbias, int* redo, const WgB3Geo& g, int nsplit, int n_nt,
                  hipStream_t st, int promised) {
  constexpr int XPP = WV_PITCH(32 * CTB), GPP = WV_PITCH(32 * NTB);
  const size_t smem = (size_t)2 * (WV_XHP * XPP + 3 * WB_T * WB_T * GPP) + 32 * NTB * sizeof(float) + 16;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)k_wgrad9_b3v<CTB, NTB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    once = true;
  }
  hipLaunchKernelGGL((k_wgrad9_b3v<CTB, NTB>), dim3(nsplit, g.n_ct * n_nt), dim3(512), smem, st, x, gy, slab, gbias, redo, g, promised);
}

// MEASURED, NOT THE DEFAULT (EVF_WGRAD_TEAMS / evf_wgrad_teams_select: 0 off (default), 1 for blocks that walk >= 16 pixel tiles, 2
// wherever the block shape fits (tests)).  LIF-EV-FlowNet shapes, B = 8, us per call incl. the slab reduction, the call repeated on
// warm data, tap-per-wave / two teams: 64^2 x 512 -> 128: 130 / 121, 32^2 x 1024 -> 256: 127 / 119, 128^2 x 256 -> 64: 129 / 120;
// 16^2 x 512 -> 512: 50.6 / 53.0, 32^2 x 256 -> 256: 51.1 / 53.4 (8 tiles per block: the fixed costs -- slab stores, reduction launch,
// prologue -- decide).  Inside the replayed step (cold operands) mode 1 changes nothing: 6.540 / 6.535 against 6.533 / 6.562 ms.
// Probe builds of THIS kernel at 128^2 x 256 -> 64 (us): complete 124; matrix team without its MFMAs 83; loader team idle (barriers
// only) 98; loads without LDS stores 99; stores without loads 101; neither team working (operand reads, barriers, slab stores,
// reduction) 43.  So: the fixed part is 43 of 124; the matrix team alone needs 55 on top of it for a bare-MFMA time of 46 (32 trips x
// 108 MFMAs x 32 cycles at 2.1 GHz); the loader team alone 40 -- its 344 MB per call (fp32 spikes, 1.56x halo of the 8 x 8 tile, g
// once per input-channel tile) are 5.9 TB/s there --; and together they take 81, not max(55, 40): matrix work under full memory
// traffic runs at the lower clock the input-gradient kernels also see.  Two tiles of loads in flight instead of one
// changed nothing (123 -> 120), and neither does halving x's bytes (a probe build of k_wgrad9_b3 that loads 8 instead of 16 bytes
// per four channels: 132 -> 128, 132 -> 131, 123 -> 125 us): what the loader team costs is not a matter of its bytes.
static int g_wb_teams = -1;
static int wb_two_team() {
  if (g_wb_teams < 0) {
    const char* e = getenv("EVF_WGRAD_TEAMS");
    g_wb_teams = e ? atoi(e) : 0;
    if (g_wb_teams < 0 || g_wb_teams > 2) g_wb_teams = 0;
  }
  return g_wb_teams;
}
extern "C" int evf_wgrad_teams_select(int mode) {
  if (mode < 0 || mode > 2) return EVF_EINVAL;
  g_wb_teams = mode;
  return EVF_OK;
}

template <int CT, int NT, int STRIDE = 1>
static void wb_go(const float* x, const float* gy, float* slab, float* gbias, int* redo, const WgB3Geo& g, int nsplit, int n_nt,
                  hipStream_t st, const WgFuse& fz) {
#if WB_TR
  constexpr int HS = (WB_T - 1) * STRIDE + 3;
  const size_t smem = (size_t)HS * HS * WB_PPITCH(32 * CT) + 3 * WB_T * WB_T * WB_PPITCH(32 * NT) + 32 * NT * sizeof(float) + 16;
#else
  const size_t smem = (size_t)32 * CT * WB_XP + 3 * 32 * NT * WB_GP + 32 * NT * sizeof(float) + 16;
#endif
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)k_wgrad9_b3<CT, NT, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    once = true;
  }
  hipLaunchKernelGGL((k_wgrad9_b3<CT, NT, STRIDE>), dim3(nsplit, g.n_ct * n_nt), dim3(576), smem, st, x, gy, slab, gbias, redo, g, fz);
}

// Can the bf16 kernel take this 3x3 stride-1 weight gradient?  (float4 tile loads of both operands)
bool evf_wgrad9_b3_ok(const float* x, const float* gy, int Cin, int Cout, int ldx, int ldg) {
  return Cin % 4 == 0 && ldx % 4 == 0 && Cout % 4 == 0 && ldg % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)gy) & 15) == 0;
}

// nsplit slabs [tap][ci][co] (every split written, also the empty ones); redo[n_ct] must be zero on entry
// fuse_gw != null: the fused reduction (x promised exact; `tickets` >= n_ct * n_nt zeroed ints, handed back zeroed; nsplit < 65536)
int evf_wgrad9_b3_launch(const float* x, int ldx, const float* gy, int ldg, float* slab, float* gbias, int* redo, int B, int H,
                         int W, int Cin, int Cout, int nsplit, int CT, int NT, hipStream_t st, float* fuse_gw, int* tickets,
                         int cin_total, int cin_off, int accumulate, int promised, int stride) {
  WgFuse fz;
  fz.gw = fuse_gw, fz.ticket = fuse_gw ? tickets : nullptr, fz.cin_total = cin_total, fz.cin_off = cin_off, fz.accumulate = accumulate;
  fz.promised = (promised || fuse_gw) ? 1 : 0;
  WgB3Geo g;
  g.B = B, g.H = H, g.W = W, g.Cin = Cin, g.Cout = Cout, g.ldx = ldx, g.ldg = ldg;
  g.OH = stride == 2 ? (H - 1) / 2 + 1 : H, g.OW = stride == 2 ? (W - 1) / 2 + 1 : W;  // (3 x 3, padding 1)
  g.tiles_x = evf_cdiv(g.OW, WB_T), g.tiles_y = evf_cdiv(g.OH, WB_T);
  g.ntiles = (long)B * g.tiles_x * g.tiles_y;
  g.tiles_per_split = (int)evf_cdiv(g.ntiles, (long)nsplit);
  g.n_ct = evf_cdiv(Cin, 32 * CT);
  const int n_nt = evf_cdiv(Cout, 32 * NT);
  if (stride == 2) {  // (no fused reduction at stride 2: fz.ticket is null there)
    if (CT == 2 && NT == 2) wb_go<2, 2, 2>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz);
    else if (CT == 2) wb_go<2, 1, 2>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz);
    else if (NT == 2) wb_go<1, 2, 2>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz);
    else wb_go<1, 1, 2>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz);
    return evf_status();
  }
  if (CT == 2 && NT == 2 && !fz.ticket && (wb_two_team() == 2 || (wb_two_team() == 1 && g.tiles_per_split >= 16))) {
    wv_go<2, 2>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz.promised);
    return evf_status();
  }
  if (CT == 2 && NT == 2)
    wb_go<2, 2>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz);
  else if (CT == 2)
    wb_go<2, 1>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz);
  else if (NT == 2)
    wb_go<1, 2>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz);
  else
    wb_go<1, 1>(x, gy, slab, gbias, redo, g, nsplit, n_nt, st, fz);
  return evf_status();
}
