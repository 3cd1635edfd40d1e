// Input-gradient conv in exact-split bf16 ("bf16x3 x bf16x3", 6 products):
//   g_x[pix][ci] (+)= sum_{tap,co} g[pix + tap][co] * Wt[tap][co][ci]
// Both operands are real valued.  g arrives already split into three bf16
// planes g = gh + gm + gl (written by evf_lif_bwd_wgrad), the transposed /
// flipped weights are split at pack time; the product keeps the six terms
// gh*wh, gh*wm, gm*wh, gh*wl, gl*wh, gm*wm (the dropped three are below 2^-24
// of the leading one), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// 108 MFMAs of 32 cycles per 32-pixel tile against 144 of 64 cycles in fp32.
//
// One wave = one M tile (32 pixels of a row) x 32 input channels; a block = an
// 8-row x 32-pixel tile whose gradient halo (3 planes) and split weights both
// live in LDS, filled by LDS-DMA (see k_conv_dgrad_b3_lds below).
#include "evf_common.h"
#include "evf_dgrad_mma.h"
#include "evf_split.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define C32 32
#define DG_ROWS 8  // tile rows = waves per block
#define NFRAG 54

__device__ __forceinline__ int dg_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ uint32_t dg_bf16(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// dst[((tau*2+m)*3+s)*64 + lane]: element e = term s of Wt[tau][co=16m+8kg+e][ci=lane&31] = w[co][ci][8-tau]
__global__ void k_pack_conv_weight_b3t(const float* __restrict__ w, uint4* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 18 * 64) return;
  const int lane = idx & 63, tm = idx >> 6, m = tm & 1, tau = tm >> 1;
  const int j = lane & 31, kg = lane >> 5;
  uint32_t t[3][8];
  for (int e = 0; e < 8; ++e) {
    const float v = w[((16 * m + 8 * kg + e) * C32 + j) * 9 + (8 - tau)];
    const uint32_t hi = dg_bf16(v);
    const float r1 = v - __uint_as_float(hi << 16);
    const uint32_t mid = dg_bf16(r1);
    const float r2 = r1 - __uint_as_float(mid << 16);
    const uint32_t lo = dg_bf16(r2);
    t[0][e] = hi, t[1][e] = mid, t[2][e] = lo;
  }
  for (int s = 0; s < 3; ++s)
    dst[(tm * 3 + s) * 64 + lane] = make_uint4(t[s][0] | (t[s][1] << 16), t[s][2] | (t[s][3] << 16),
                                               t[s][4] | (t[s][5] << 16), t[s][6] | (t[s][7] << 16));
}

extern "C" int evf_pack_conv_weight_b3t(const float* w, int Cout, int Cin, void* dst, void* stream) {
  if (!w || !dst || Cout != C32 || Cin != C32) return EVF_EINVAL;
  hipLaunchKernelGGL(k_pack_conv_weight_b3t, dim3(evf_cdiv(18 * 64, 256)), dim3(256), 0, EVF_STREAM(stream), w,
                     (uint4*)dst);
  return evf_status();
}

// All 32->32 conv weights of a network in ONE launch (the weights change every optimizer step, and a
// launch is ~4 us of graph time whatever it does): blockIdx.y = tensor, blockIdx.z = 0 forward layout
// (evf_pack_conv_weight_b3), 1 transposed/flipped layout for the input gradient (evf_pack_conv_weight_b3t).
struct PackMulti {
  const float* w[16];
  uint4* fwd[16];
  uint4* bwd[16];
};
__global__ void k_pack_conv_weight_b3_multi(PackMulti a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 18 * 64) return;
  const float* __restrict__ w = a.w[blockIdx.y];
  const bool tr = blockIdx.z != 0;
  uint4* __restrict__ dst = tr ? a.bwd[blockIdx.y] : a.fwd[blockIdx.y];
  if (!dst) return;
  const int lane = idx & 63, tm = idx >> 6, m = tm & 1, tau = tm >> 1;
  const int j = lane & 31, kg = lane >> 5;
  uint32_t t[3][8];
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * m + 8 * kg + e;
    const float v = tr ? w[(k * C32 + j) * 9 + (8 - tau)] : w[(j * C32 + k) * 9 + tau];
    const uint32_t hi = dg_bf16(v);
    const float r1 = v - __uint_as_float(hi << 16);
    const uint32_t mid = dg_bf16(r1);
    const float r2 = r1 - __uint_as_float(mid << 16);
    const uint32_t lo = dg_bf16(r2);
    t[0][e] = hi, t[1][e] = mid, t[2][e] = lo;
  }
  for (int s = 0; s < 3; ++s)
    dst[(tm * 3 + s) * 64 + lane] = make_uint4(t[s][0] | (t[s][1] << 16), t[s][2] | (t[s][3] << 16),
                                               t[s][4] | (t[s][5] << 16), t[s][6] | (t[s][7] << 16));
}

extern "C" int evf_pack_conv_weights_b3_multi(const void* const* w, void* const* dst_b3, void* const* dst_b3t, int count,
                                              void* stream) {
  if (!w || count <= 0 || count > 16 || (!dst_b3 && !dst_b3t)) return EVF_EINVAL;
  PackMulti a;
  for (int i = 0; i < 16; ++i) {
    a.w[i] = i < count ? (const float*)w[i] : nullptr;
    a.fwd[i] = (i < count && dst_b3) ? (uint4*)dst_b3[i] : nullptr;
    a.bwd[i] = (i < count && dst_b3t) ? (uint4*)dst_b3t[i] : nullptr;
    if (i < count && !a.w[i]) return EVF_EINVAL;
  }
  hipLaunchKernelGGL(k_pack_conv_weight_b3_multi, dim3(evf_cdiv(18 * 64, 256), count, 2), dim3(256), 0, EVF_STREAM(stream),
                     a);
  return evf_status();
}

// ---------------------------------------------------------------------------
// The split gradient halo of an 8-row x 32-pixel tile (10 x 34 pixels x 3 planes x 64 B =
// 65 KiB) and the 54 KiB of split weights are brought in by LDS-DMA (global_load_lds_dwordx4: no VGPRs, full
// 64-byte pixel lines, everything in flight at once), ONE wait, then the 108 MFMAs of every wave read both operands
// from LDS.  An earlier register-fragment version paid a chain of dependent latencies (weights -> row 0 -> row 1 -> row 2 ->
// read-modify-write) and fetched each A fragment through the L1 nine times in 16-byte pieces (26.4 vs 22.5 us).
// LDS image of a plane: [halo pixel][4 x 16-byte chunks], chunk c of pixel p stored in slot c ^ ((p >> 2) & 3): the
// DMA writes lane-linear (64 lanes = 16 pixels x 4 slots), the swizzle is applied on the SOURCE address, and a
// fragment read (16 consecutive pixels, one chunk) touches all 64 banks once.
// ---------------------------------------------------------------------------
#define DL_HW 34
#define DL_HR (DG_ROWS + 2)
#define DL_HP (DL_HR * DL_HW)  // 340 halo pixels
#define DL_UPP 22              // 16-pixel DMA units per plane (352 pixel slots, the last 12 are padding)
#define DL_HPP (DL_UPP * 16)
typedef __attribute__((address_space(3))) void dl_lds_void;
typedef __attribute__((address_space(1))) const void dl_glb_void;

// F32IN: the gradient arrives as ONE fp32 tensor [B,H,W,32] (g_cur of evf_lif_bwd_wgrad); the exact 3-way bf16
// split happens here while the halo is staged through registers (same split, so the result is bit-identical to the
// pre-split form) -- the producer writes 128 instead of 192 B/pixel and this kernel reads 128 instead of 192.
// ACC / PLIF: compile-time, so that a launch that neither accumulates nor carries the PLIF term issues none of the 48
// per-lane operand loads (as dummy loads they still cost the texture addresser 16 cycles each: 6 k cycles per tile).
#ifdef EVF_SPAN  // start / end of every block in the chip-wide 100 MHz counter (probe build through EVF_LIB)
__device__ unsigned long long dg_span[2 * 4096];
extern "C" int evf_debug_dg_span(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(dg_span), sizeof(dg_span))); }
#define DG_SPAN_MARK(w)                                                                                                     \
  do {                                                                                                                      \
    const int bid_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                                        \
    if (threadIdx.x == 0 && bid_ < 4096) dg_span[2 * bid_ + (w)] = __builtin_amdgcn_s_memrealtime();                        \
  } while (0)
#else
#define DG_SPAN_MARK(w) do {} while (0)
#endif

#define DG_SP 36  // floats per pixel of the epilogue staging tile (32 + 4: conflict-free 16-byte writes)

// PAIR: two weight sets (wt2 / gx2) -- compile time, so that the profiler sees the one- and the two-product launches as
// different kernels and the tile's product loop has a fixed trip count
template <bool F32IN, bool ACC, bool PLIF, bool PAIR>
__device__ __forceinline__ void dg_body(const int zz, const int zb,  // first sample of this block and the sample stride
                                        const uint4* __restrict__ gs, long plane_stride,
                                                                    const uint4* __restrict__ wt, float* __restrict__ gx,
                                                                    int accumulate, int B, int H, int W,
                                                                    const float* __restrict__ gPb,
                                                                    const uint32_t* __restrict__ xbits,
                                                                    const uint4* __restrict__ wt2,
                                                                    float* __restrict__ gx2) {
  // wt2 / gx2 (optional): a SECOND weight set applied to the same gradient tile, written (not accumulated) to gx2 --
  // the recurrent conv's input gradient next to the feed-forward one: one launch and one halo load instead of two.
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_w = (uint4*)smem_raw;  // NFRAG*64
  uint4* s_a = s_w + NFRAG * 64;  // [3][DL_HPP][4]
  float* s_stage = (float*)(s_a + 3 * DL_HPP * 4);  // [DG_ROWS waves][32 pixels][DG_SP]: epilogue staging
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  DG_SPAN_MARK(0);
  const int y0 = blockIdx.y * DG_ROWS, y = y0 + wv, x0 = blockIdx.x * 32;
  const int i = lane & 31, kg = lane >> 5;
  // the split weights are staged once per block and serve all its tiles (samples blockIdx.z, blockIdx.z + gridDim.z, ...)
#ifndef PROBE_NO_WEIGHT_DMA  // (probe build: what does staging the 54 KiB of weights per block cost?)
  for (int u = wv; u < NFRAG; u += DG_ROWS)
    __builtin_amdgcn_global_load_lds((dl_glb_void*)(wt + u * 64 + lane), (dl_lds_void*)(s_w + u * 64), 16, 0, 0);
#endif
  int tile_it = 0;
  for (int b = zz; b < B; b += zb, ++tile_it) {
    if (b != zz) __syncthreads();  // every wave is done reading the previous tile's halo
    // ---- everything this tile reads, requested at once
    if (F32IN) {
      const float4* gf = (const float4*)gs;
      constexpr int NIT = (DL_HP * 4 + DG_ROWS * 64 - 1) / (DG_ROWS * 64);  // (halo pixel, 8-channel chunk) items per thread
      float4 lo4[NIT], hi4[NIT];
#pragma unroll
      for (int n = 0; n < NIT; ++n) {
        const int it = min(tid + n * DG_ROWS * 64, DL_HP * 4 - 1), p = it >> 2, c = it & 3;
        const int hr = p / DL_HW, hc = p - hr * DL_HW;
        const int yy = min(max(y0 - 1 + hr, 0), H - 1), xx = min(max(x0 - 1 + hc, 0), W - 1);  // out-of-image: masked at use
        const float4* src = gf + (((long)b * H + yy) * W + xx) * 8 + 2 * c;
        lo4[n] = src[0], hi4[n] = src[1];
      }
#pragma unroll
      for (int n = 0; n < NIT; ++n) {
        const int it = tid + n * DG_ROWS * 64;
        if (it < DL_HP * 4) {
          const int p = it >> 2, c = it & 3;
          const float v[8] = {lo4[n].x, lo4[n].y, lo4[n].z, lo4[n].w, hi4[n].x, hi4[n].y, hi4[n].z, hi4[n].w};
          uint32_t t3[3][4];
#pragma unroll
          for (int e = 0; e < 4; ++e)  // g = hi + mid + lo, the split of evf_lif_bwd_wgrad, two channels per step (evf_split.h)
            evf_split3_pair(v[2 * e], v[2 * e + 1], t3[0][e], t3[1][e], t3[2][e]);
          const int slot = p * 4 + (c ^ ((p >> 2) & 3));
#pragma unroll
          for (int sp = 0; sp < 3; ++sp)
            s_a[sp * DL_HPP * 4 + slot] = make_uint4(t3[sp][0], t3[sp][1], t3[sp][2], t3[sp][3]);
        }
      }
    }
    for (int q = wv; !F32IN && q < 3 * DL_UPP; q += DG_ROWS) {
      const int sp = q / DL_UPP, u = q - sp * DL_UPP;
      const int p = 16 * u + (lane >> 2), pc = min(p, DL_HP - 1);
      const int hr = pc / DL_HW, hc = pc - hr * DL_HW;
      const int yy = min(max(y0 - 1 + hr, 0), H - 1), xx = min(max(x0 - 1 + hc, 0), W - 1);  // out-of-image: masked at use
      const int c = (lane & 3) ^ ((p >> 2) & 3);
      const uint4* g = gs + sp * plane_stride + (((long)b * H + yy) * W + xx) * 4 + c;
      __builtin_amdgcn_global_load_lds((dl_glb_void*)g, (dl_lds_void*)(s_a + (sp * DL_HPP + 16 * u) * 4), 16, 0, 0);
    }
    // epilogue operands of this lane's pixel (column x0 + i): channels 8q + 4kg .. +3, q = 0..3
    const long pixq = ((long)b * H + min(y, H - 1)) * W + min(x0 + i, W - 1);
    float4 oldv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) oldv[q] = ACC ? *(const float4*)(gx + pixq * C32 + 8 * q + 4 * kg) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float pv = PLIF ? evf_plif_gp(gPb, (accumulate & 2) ? 1 : 0, b, y, x0 + i, H, W) : 0.f;
    const uint32_t xb = PLIF ? xbits[pixq] : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t msk[9];
#pragma unroll
    for (int tau = 0; tau < 9; ++tau) {
      const int yy = y + tau / 3 - 1, xx = x0 + i + tau % 3 - 1;
      msk[tau] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 0xFFFFFFFFu : 0u;
    }
    auto matrix_phase = [&]() -> f32x16 { return dg_matrix_phase<true>(s_w, s_a, DL_HPP * 4, wv * DL_HW + i, lane, msk); };
    // with two weight sets the order alternates from tile to tile: the set left in LDS by the previous tile goes first
    constexpr int nset = PAIR ? 2 : 1;
#pragma unroll
    for (int k = 0; k < nset; ++k) {
      const int set = PAIR ? ((tile_it + k) & 1) : 0;
      if (k) {  // swap the weight set
        __syncthreads();  // every wave is done with the current one
        const uint4* wsrc = set ? wt2 : wt;
        for (int u = wv; u < NFRAG; u += DG_ROWS)
          __builtin_amdgcn_global_load_lds((dl_glb_void*)(wsrc + u * 64 + lane), (dl_lds_void*)(s_w + u * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      const f32x16 acc = matrix_phase();
      // Epilogue through a wave-private LDS tile [32 pixels][32 channels] (144-byte pixel pitch: conflict-free b128
      // writes): the MFMA layout gives a lane 4 x 16 bytes of its pixel's 128-byte line, i.e. four partial-line store
      // instructions; read back as 8 pixels x 128 bytes per instruction the wave stores FULL lines, and those can be
      // non-temporal -- the output does not stay dirty in the L2s, so the kernel no longer ends in their write-back
      // (every block of this kernel was done 6-7 us before the kernel retired: tools/probes/span_step.py).
      float* st = s_stage + wv * (32 * DG_SP);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
        if (set == 0) {
          // PLIF: the pooled pre-synaptic trace also reads the input spikes: d mean_c|x| / dx_c = 1/32 where the
          // spike is set, AvgPool3x3^T = box filter / 9 -- gPb is that filtered, scaled map (evf_plif_trace_bwd)
          const uint32_t xq = xb >> (8 * q + 4 * kg);
          const float o[4] = {oldv[q].x, oldv[q].y, oldv[q].z, oldv[q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e] + o[e] + (((xq >> e) & 1u) ? pv : 0.f);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e];
        }
        *(float4*)(st + i * DG_SP + 8 * q + 4 * kg) = make_float4(v[0], v[1], v[2], v[3]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float* dstp = set == 0 ? gx : gx2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
        const float4 v = *(const float4*)(st + p * DG_SP + c4);
        if (y < H && x0 + p < W) evf_store_nt(dstp + (((long)b * H + y) * W + x0 + p) * C32 + c4, v);
      }
      __builtin_amdgcn_wave_barrier();  // (the tile is rewritten by this wave's next product)
    }
  }
  DG_SPAN_MARK(1);
}

template <bool F32IN, bool ACC, bool PLIF, bool PAIR>
__global__ __launch_bounds__(DG_ROWS * 64) void k_conv_dgrad_b3_lds(const uint4* __restrict__ gs, long plane_stride,
                                                                    const uint4* __restrict__ wt, float* __restrict__ gx,
                                                                    int accumulate, int B, int H, int W,
                                                                    const float* __restrict__ gPb,
                                                                    const uint32_t* __restrict__ xbits,
                                                                    const uint4* __restrict__ wt2,
                                                                    float* __restrict__ gx2) {
  dg_body<F32IN, ACC, PLIF, PAIR>(blockIdx.z, gridDim.z, gs, plane_stride, wt, gx, accumulate, B, H, W, gPb, xbits, wt2, gx2);
}

// Several independent input-gradient cells of a window in one launch (evf_bwd_defer_*, see evf_bwd_fused.hip): the fp32
// gradient form without accumulation, one or two weight sets per cell.  blockIdx.z = cell * zb + first sample.
#define DG_MAX_JOBS 8
struct DgJob {
  const uint4* gs;   // fp32 g_cur [B,H,W,32], or (split) the three bf16 planes [term][B,H,W,32]
  const uint4* wt;
  float* gx;
  const uint4* wt2;  // NULL: one weight set
  float* gx2;
  int split;         // the gradient arrives pre-split (evf_conv_dgrad_b3[_pair]): launched by k_dgrad_diag_dma
};
struct DgJobs {
  DgJob j[DG_MAX_JOBS];
};
__global__ __launch_bounds__(DG_ROWS * 64) void k_dgrad_diag(DgJobs jobs, int B, int H, int W, int zb) {
  const int jb = blockIdx.z / zb, zz = blockIdx.z - jb * zb;
  const DgJob& J = jobs.j[jb];
  if (J.wt2)
    dg_body<true, false, false, true>(zz, zb, J.gs, 0, J.wt, J.gx, 0, B, H, W, nullptr, nullptr, J.wt2, J.gx2);
  else
    dg_body<true, false, false, false>(zz, zb, J.gs, 0, J.wt, J.gx, 0, B, H, W, nullptr, nullptr, nullptr, nullptr);
}

struct DgDefer {
  int B, H, W, split;
  int n[EVF_BWD_DIAGS];
  DgJob job[EVF_BWD_DIAGS][DG_MAX_JOBS];
};
static DgDefer dg_tab[EVF_CTX_MAX];

static int dg_zb(int B, int H, int W) {  // samples per block column, as in dg_launch
  const long tiles = (long)evf_cdiv(W, 32) * evf_cdiv(H, DG_ROWS);
  int zb = B;
  for (int z = 1; z < B; ++z)
    if (B % z == 0 && tiles * z <= 256 && tiles * z >= 192) zb = z;
  return zb;
}
static size_t dg_lds_bytes() {
  return (size_t)(NFRAG * 64 + 3 * DL_HPP * 4) * sizeof(uint4) + (size_t)DG_ROWS * 32 * DG_SP * 4;
}

static int dg_diag_select = -1;  // -1 environment / default, 0 k_dgrad_diag, 1 k_dgrad_diag_ws / _dma, 2 ... with the ring-halo kernel
int evf_dgrad_ring_select = -1;  // (evf_dgrad_diag.hip) -1 environment, 0 k_dgrad_diag_dma, 1 k_dgrad_diag_ring
extern "C" int evf_dgrad_diag_select(int which) {
  if (which < -1 || which > 2) return EVF_EINVAL;
  dg_diag_select = which == 2 ? 1 : which;
  evf_dgrad_ring_select = which < 0 ? -1 : (which == 2 ? 1 : 0);
  return EVF_OK;
}

int evf_dg_defer_count(int ctx) {
  const DgDefer& dg_defer = dg_tab[ctx];
  int n = 0;
  for (int d = 0; d < EVF_BWD_DIAGS; ++d) n += dg_defer.n[d];
  return n;
}
int evf_dg_defer_pending(int ctx, int d) { return dg_tab[ctx].n[d]; }
int evf_dg_defer_launch(int ctx, int d, void* stream) {
  DgDefer& dg_defer = dg_tab[ctx];
  const int n = dg_defer.n[d];
  if (!n) return EVF_OK;
  // default: the persistent wave-specialised launch over the flat product list (evf_dgrad_diag.hip); EVF_DGRAD_DIAG=lds keeps
  // the one-phase-after-the-other body below (A/B measurements, the bit-identity test)
  static const bool env_ws = []() {
    const char* e = getenv("EVF_DGRAD_DIAG");
    return !(e && e[0] == 'l');
  }();
  if (dg_defer.split || (dg_diag_select < 0 ? env_ws : dg_diag_select == 1)) {
    EvfDgProds P;
    int np = 0;
    for (int k = 0; k < n; ++k) {
      const DgJob& J = dg_defer.job[d][k];
      P.p[np++] = EvfDgProd{J.gs, J.wt, J.gx};
      if (J.wt2) P.p[np++] = EvfDgProd{J.gs, J.wt2, J.gx2};
    }
    for (int k = np; k < EVF_DG_MAX_PROD; ++k) P.p[k] = P.p[0];
    evf_prof_mark(2, 0, stream);
    const int rc = dg_defer.split ? evf_dgrad_diag_dma_launch(P, np, dg_defer.B, dg_defer.H, dg_defer.W, stream)
                                  : evf_dgrad_diag_ws_launch(P, np, dg_defer.B, dg_defer.H, dg_defer.W, stream);
    evf_prof_mark(2, 1, stream);
    dg_defer.n[d] = 0;
    return rc;
  }
  const size_t lds = dg_lds_bytes();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_dgrad_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  DgJobs jobs;
  for (int k = 0; k < DG_MAX_JOBS; ++k) jobs.j[k] = dg_defer.job[d][k < n ? k : 0];
  const int zb = dg_zb(dg_defer.B, dg_defer.H, dg_defer.W);
  dim3 grid(evf_cdiv(dg_defer.W, 32), evf_cdiv(dg_defer.H, DG_ROWS), zb * n), block(DG_ROWS * 64);
  evf_prof_mark(2, 0, stream);
  hipLaunchKernelGGL(k_dgrad_diag, grid, block, lds, EVF_STREAM(stream), jobs, dg_defer.B, dg_defer.H, dg_defer.W, zb);
  evf_prof_mark(2, 1, stream);
  dg_defer.n[d] = 0;
  return evf_status();
}

static int dg_select = -1;  // -1 by shape, 0 k_conv_dgrad_b3_lds, 1 k_conv_dgrad_ws
extern "C" int evf_conv_dgrad_select(int which) {
  if (which < -1 || which > 1) return EVF_EINVAL;
  dg_select = which;
  return EVF_OK;
}

static int dg_launch(const void* g, int f32in, const void* wT_b3, float* g_x, int accumulate, int B, int H, int W,
                     const float* g_P, const uint32_t* x_bits, const void* wT2_b3, float* g_x2, void* stream) {
  if (!g || !wT_b3 || !g_x || B <= 0 || H <= 0 || W <= 0 || ((g_P != nullptr) != (x_bits != nullptr)) ||
      ((wT2_b3 != nullptr) != (g_x2 != nullptr)))
    return EVF_EINVAL;
  // fp32 gradient in: two kernels with bit-identical results (shared matrix phase, evf_dgrad_mma.h).  The wave-specialised
  // one (evf_dgrad_ws.hip: producers split the next tile's halo while the consumers' MFMAs run) wins once a CU gets several
  // tiles (260 x 346 x B4: 46 vs 60 us); with <= 2 tiles per CU the cold first fetch dominates either way and the
  // one-phase-after-the-other kernel below is ~1 % ahead in the train step (128 x 128 x B8: 20.1 vs 21.3 us in the step).
  // evf_conv_dgrad_select() / EVF_DGRAD=lds|ws override the choice (A/B measurements, the equivalence test).
  const int bctx = evf_ctx_find(stream);
  const EvfBwdDefer evf_bwd_defer = bctx >= 0 ? evf_bwd_defer_tab[bctx] : EvfBwdDefer{false, 0, false};
  DgDefer& dg_defer = dg_tab[bctx < 0 ? 0 : bctx];
  if (evf_bwd_defer.active) {  // a recording is open on this stream: record the cell (any size: the persistent launch of evf_dgrad_diag.hip) ...
    const bool any = evf_dg_defer_count(bctx) != 0;
    const int split = f32in ? 0 : 1;  // (one kind per recording: the two are launched by different kernels)
    const bool same = !any || (dg_defer.B == B && dg_defer.H == H && dg_defer.W == W && dg_defer.split == split);
    if (!accumulate && !g_P && same && dg_defer.n[evf_bwd_defer.slot] < DG_MAX_JOBS && evf_dgrad_diag_fits(split, B, H, W)) {
      dg_defer.B = B, dg_defer.H = H, dg_defer.W = W, dg_defer.split = split;
      dg_defer.job[evf_bwd_defer.slot][dg_defer.n[evf_bwd_defer.slot]++] =
          DgJob{(const uint4*)g, (const uint4*)wT_b3, g_x, (const uint4*)wT2_b3, g_x2, split};
      return EVF_OK;
    }
    const int rc = evf_bwd_defer_flush_now(bctx, stream);  // ... or, not recordable: everything recorded runs first
    if (rc) return rc;
  }
  if (f32in) {
    int mode = dg_select;
    if (mode < 0) {
      static const int env_mode = []() {
        const char* e = getenv("EVF_DGRAD");
        return !e ? -1 : (e[0] == 'l' ? 0 : (e[0] == 'w' ? 1 : -1));
      }();
      mode = env_mode;
    }
    if (mode < 0) mode = ((long)B * evf_cdiv(H, 4) * evf_cdiv(W, 32) >= 6L * 256) ? 1 : 0;
    if (mode == 1)
      return evf_dgrad_ws_launch((const float*)g, wT_b3, g_x, accumulate, B, H, W, g_P, x_bits, wT2_b3, g_x2, 0, stream);
  }
  // Samples per block (the 54 KiB of split weights are staged once per block): several only when the whole grid
  // then is ONE round of the 256 CUs (B = 8 at 128 x 128: 256 blocks x 2 tiles, 1 % faster than 512 x 1); with more
  // rounds than that, fat blocks only coarsen the tail (260 x 346: 726 x 2 tiles was 11 % slower than 1452 x 1).
  const long tiles = (long)evf_cdiv(W, 32) * evf_cdiv(H, DG_ROWS);
  int zb = B;
  for (int z = 1; z < B; ++z)
    if (B % z == 0 && tiles * z <= 256 && tiles * z >= 192) zb = z;
  dim3 grid(evf_cdiv(W, 32), evf_cdiv(H, DG_ROWS), zb), block(DG_ROWS * 64);
  const long plane_stride = (long)B * H * W * 4;  // uint4 per term plane: npix * 32 bf16 / 8
  const size_t lds = (size_t)(NFRAG * 64 + 3 * DL_HPP * 4) * sizeof(uint4) + (size_t)DG_ROWS * 32 * DG_SP * 4;  // 138 KiB
  const bool acc = (accumulate & 1) != 0, plif = g_P != nullptr;  // (accumulate & 2: g_P is the raw map, see evf_plif_gp)
#define DG_GO2(F_, A_, P_, R_)                                                                                                  \
  do {                                                                                                                     \
    static bool attr = false;                                                                                              \
    if (!attr) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)k_conv_dgrad_b3_lds<F_, A_, P_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)lds);                                                                                 \
      attr = true;                                                                                                         \
    }                                                                                                                      \
    hipLaunchKernelGGL((k_conv_dgrad_b3_lds<F_, A_, P_, R_>), grid, block, lds, EVF_STREAM(stream), (const uint4*)g,           \
                       plane_stride, (const uint4*)wT_b3, g_x, accumulate, B, H, W, g_P, x_bits, (const uint4*)wT2_b3,     \
                       g_x2);                                                                                              \
  } while (0)
#define DG_GO(F_, A_, P_)          \
  do {                             \
    if (wT2_b3)                    \
      DG_GO2(F_, A_, P_, true);    \
    else                           \
      DG_GO2(F_, A_, P_, false);   \
  } while (0)
#define DG_AP(F_)                  \
  do {                             \
    if (acc && plif)               \
      DG_GO(F_, true, true);       \
    else if (acc)                  \
      DG_GO(F_, true, false);      \
    else if (plif)                 \
      DG_GO(F_, false, true);      \
    else                           \
      DG_GO(F_, false, false);     \
  } while (0)
  if (f32in)
    DG_AP(true);
  else
    DG_AP(false);
#undef DG_AP
#undef DG_GO
#undef DG_GO2
  return evf_status();
}

extern "C" int evf_conv_dgrad_b3(const void* g_split, const void* wT_b3, float* g_x, int accumulate, int B, int H, int W,
                                 const float* g_P, const uint32_t* x_bits, void* stream) {
  return dg_launch(g_split, 0, wT_b3, g_x, accumulate, B, H, W, g_P, x_bits, nullptr, nullptr, stream);
}

// ... and both input gradients of a recurrent cell from the pre-split planes (see evf_conv_dgrad_b3_f32_pair)
extern "C" int evf_conv_dgrad_b3_pair(const void* g_split, const void* wT_b3, float* g_x, int accumulate, const void* wT2_b3,
                                      float* g_x2, int B, int H, int W, const float* g_P, const uint32_t* x_bits, void* stream) {
  if (!wT2_b3 || !g_x2) return EVF_EINVAL;
  return dg_launch(g_split, 0, wT_b3, g_x, accumulate, B, H, W, g_P, x_bits, wT2_b3, g_x2, stream);
}

// the same from the fp32 gradient g_cur [B,H,W,32]: split on the fly, bit-identical result
extern "C" int evf_conv_dgrad_b3_f32(const float* g_cur, const void* wT_b3, float* g_x, int accumulate, int B, int H, int W,
                                     const float* g_P, const uint32_t* x_bits, void* stream) {
  return dg_launch(g_cur, 1, wT_b3, g_x, accumulate, B, H, W, g_P, x_bits, nullptr, nullptr, stream);
}

// Both input gradients of a recurrent cell from one pass over g_cur: g_x (+)= conv^T(g, W_ff) as above and
// g_x2 = conv^T(g, W_rec) (written: it is dL/d(previous output spikes), spiking_submodules.py:530).
extern "C" int evf_conv_dgrad_b3_f32_pair(const float* g_cur, const void* wT_b3, float* g_x, int accumulate, const void* wT2_b3,
                                          float* g_x2, int B, int H, int W, const float* g_P, const uint32_t* x_bits,
                                          void* stream) {
  if (!wT2_b3 || !g_x2) return EVF_EINVAL;
  return dg_launch(g_cur, 1, wT_b3, g_x, accumulate, B, H, W, g_P, x_bits, wT2_b3, g_x2, stream);
}
