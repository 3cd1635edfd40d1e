// Spiking conv cells for gfx950: 3x3 convolutions as implicit GEMMs on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains), fused
// with the LIF neuron update, plus the BPTT backward (neuron backward, input-
// gradient conv, weight-gradient conv).
//
// Data layout:
//   membrane potential / gradients : [B][H][W][32] float32 (channels last)
//   spikes                         : one uint32 per pixel, bit c = channel c
//   conv weights                   : packed per (tap, k-step) as the MFMA B
//                                    operand, see k_pack_conv_weight
//
// GEMM view of one 3x3 conv with 32 in / 32 out channels:
//   M = pixels (tile of 32 consecutive x), N = 32 output channels,
//   K = 9 taps x 32 input channels.  The K order is permuted so that lane
//   half h (= lane>>5) owns input channels 16h..16h+15: k-step t of tap tau
//   multiplies channel 16h+t.  A operands of binary layers are built from the
//   spike bit mask in registers (v_bfe + v_cvt), so spikes cost 4 B/pixel of
//   HBM/LDS traffic instead of 128 B.
#include "evf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define C32 32
#define TH 8   // tile rows  (4 waves x 2 rows)
#define TW 32  // tile cols  (= one MFMA M tile)
#define HALO_W (TW + 2)
#define HALO_H (TH + 2)
#define WPACK (9 * 16 * 64)  // floats per packed 32x32x3x3 weight

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// C/D layout of the 32x32 MFMA: column = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float evf_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// --------------------------------------------------------------------------
// weight packing
// --------------------------------------------------------------------------
// fwd:  dst[(tau*16+t)*64 + h*32 + j] = w[co=j][ci=16h+t][tau]
// bwd:  dst[(tau*16+t)*64 + h*32 + j] = w[co=16h+t][ci=j][8-tau]   (flipped, transposed)
__global__ void k_pack_conv_weight(const float* __restrict__ w, int transposed, float* __restrict__ dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= WPACK) return;
  const int l = e & 63, t = (e >> 6) & 15, tau = e >> 10;
  const int h = l >> 5, j = l & 31;
  float v;
  if (!transposed)
    v = w[(j * C32 + (16 * h + t)) * 9 + tau];
  else
    v = w[((16 * h + t) * C32 + j) * 9 + (8 - tau)];
  dst[e] = v;
}

extern "C" int evf_pack_conv_weight(const float* w, int Cout, int Cin, int transposed, float* dst, void* stream) {
  if (!w || !dst || Cout != C32 || Cin != C32) return EVF_EINVAL;
  hipLaunchKernelGGL(k_pack_conv_weight, dim3(evf_cdiv(WPACK, 256)), dim3(256), 0, EVF_STREAM(stream), w, transposed,
                     dst);
  return evf_status();
}

// slabs [nslab][9][ci][co] -> torch layout dst[co][ci][tau] (+)=
__global__ void k_reduce_wgrad(const float* __restrict__ partial, int nslab, int accumulate, float* __restrict__ dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // e = (tau*32 + ci)*32 + co
  if (e >= 9 * C32 * C32) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += partial[(long)k * (9 * C32 * C32) + e];
  const int co = e & 31, ci = (e >> 5) & 31, tau = e >> 10;
  float* d = dst + (co * C32 + ci) * 9 + tau;
  *d = accumulate ? *d + s : s;
}

extern "C" int evf_unpack_conv_wgrad(const float* packed, int Cout, int Cin, int accumulate, float* dst, void* stream) {
  if (!packed || !dst || Cout != C32 || Cin != C32) return EVF_EINVAL;
  hipLaunchKernelGGL(k_reduce_wgrad, dim3(evf_cdiv(9 * C32 * C32, 256)), dim3(256), 0, EVF_STREAM(stream), packed, 1,
                     accumulate, dst);
  return evf_status();
}

// slab groups in parallel: block (x, y) sums slabs y, y+G, ... for 256 outputs, then adds
// its partial into dst (torch layout) with one atomic per output
#define RS_GROUPS 8
__global__ void k_reduce_wgrad_par(const float* __restrict__ partial, int nslab, float* __restrict__ dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // e = (tau*32 + ci)*32 + co
  if (e >= 9 * C32 * C32) return;
  float s = 0.f;
  for (int k = blockIdx.y; k < nslab; k += RS_GROUPS) s += partial[(long)k * (9 * C32 * C32) + e];
  const int co = e & 31, ci = (e >> 5) & 31, tau = e >> 10;
  evf_atomic_add(dst + (co * C32 + ci) * 9 + tau, s);
}

extern "C" int evf_reduce_slabs(const float* partial, int nslab, int n, int accumulate, float* dst, void* stream) {
  if (!partial || !dst || nslab <= 0 || n != 9 * C32 * C32) return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  if (!accumulate) {
    int rc = evf_hip(evf_memset_async(dst, 0, sizeof(float) * n, st));
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_reduce_wgrad_par, dim3(evf_cdiv(n, 256), RS_GROUPS), dim3(256), 0, st, partial, nslab, dst);
  return evf_status();
}

// The slabs of all weight tensors of a network in one launch (blockIdx.z = tensor): 8 launches of 16 us
// become one of about the same length.  Always accumulates into dst[t] (torch layout).
struct RsMulti {
  const float* src[16];
  float* dst[16];
};
#define RSM_GROUPS 16
// block = 64 outputs x 16 slab groups; the groups meet in LDS and ONE thread per output does a plain
// read-modify-write (the first version had every group add atomically: 1.2 M atomics on 74 k addresses
// took 50 of its 68 us)
__global__ __launch_bounds__(64 * RSM_GROUPS) void k_reduce_wgrad_multi(RsMulti a, int nslab) {
  __shared__ float red[RSM_GROUPS][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + tx;  // e = (tau*32 + ci)*32 + co;  9216 = 144 * 64
  const float* __restrict__ p = a.src[blockIdx.z] + e;
  constexpr long N = 9 * C32 * C32;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = ty;
  for (; k + 3 * RSM_GROUPS < nslab; k += 4 * RSM_GROUPS) {  // four independent loads in flight
    s0 += p[(long)k * N];
    s1 += p[(long)(k + RSM_GROUPS) * N];
    s2 += p[(long)(k + 2 * RSM_GROUPS) * N];
    s3 += p[(long)(k + 3 * RSM_GROUPS) * N];
  }
  for (; k < nslab; k += RSM_GROUPS) s0 += p[(long)k * N];
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < RSM_GROUPS; ++g) s += red[g][tx];
    const int co = e & 31, ci = (e >> 5) & 31, tau = e >> 10;
    a.dst[blockIdx.z][(co * C32 + ci) * 9 + tau] += s;
  }
}

extern "C" int evf_reduce_slabs_multi(const void* const* partial, void* const* dst, int count, int nslab, int n,
                                      void* stream) {
  if (!partial || !dst || count <= 0 || count > 16 || nslab <= 0 || n != 9 * C32 * C32) return EVF_EINVAL;
  RsMulti a;
  for (int i = 0; i < 16; ++i) {
    a.src[i] = i < count ? (const float*)partial[i] : nullptr;
    a.dst[i] = i < count ? (float*)dst[i] : nullptr;
    if (i < count && (!a.src[i] || !a.dst[i])) return EVF_EINVAL;
  }
  hipLaunchKernelGGL(k_reduce_wgrad_multi, dim3(n / 64, 1, count), dim3(64 * RSM_GROUPS), 0, EVF_STREAM(stream), a, nslab);
  return evf_status();
}

// --------------------------------------------------------------------------
// forward: conv(s) + LIF update
// --------------------------------------------------------------------------
// LIF epilogue shared by the binary and the dense-input kernels.
// ConvLIF.forward spiking_submodules.py:103-126 / ConvLIFRecurrent :523-551:
//   v' = (v*leak)*(1-z) + (1-leak)*cur          (hard reset)
//   v' = v*leak + (1-leak)*cur - z*thresh       (soft reset)
//   z' = (v' - thresh) > 0
// Previous state of the 16 pixels of this lane (MFMA accumulator layout): unconditional loads from clamped addresses,
// all in flight together (a load under `if (ok)` is followed by its own s_waitcnt vmcnt(0): 16 serial round trips).
// Split from the update so that a kernel can issue them BEFORE its staging / matrix phase.
template <typename ZPrev>
__device__ __forceinline__ void lif_load_prev(int b, int row, int x0, int H, int W, int lane,
                                              const float* __restrict__ v_prev, const float* __restrict__ v_out,
                                              ZPrev zprev_word, float (&vpv)[16], uint32_t (&zw)[16]) {
  const int j = lane & 31;
  const int rq = min(row, H - 1);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int cq = min(x0 + mfma_row(r, lane), W - 1);
    const float* src = v_prev ? v_prev : v_out;
    const float val = src[(((long)b * H + rq) * W + cq) * C32 + j];
    vpv[r] = v_prev ? val : 0.f;
    zw[r] = zprev_word(rq, cq);
  }
}

// thx: NULL, or the 16 per-element increments of the threshold (XLIF head: t1 * pt', spiking_submodules.py:419; hard reset only)
__device__ __forceinline__ void lif_update(const f32x16& acc, const float (&vpv)[16], const uint32_t (&zw)[16], int b,
                                           int row, int x0, int H, int W, int lane, float lam, float th, int hard_reset,
                                           float* __restrict__ v_out, uint32_t* __restrict__ z_out,
                                           uint32_t* __restrict__ zT_out, const float* thx = nullptr) {
  const int j = lane & 31;
  const bool row_ok = row < H;
  uint32_t plane = 0u;  // this channel's spikes over the tile's 32 pixels (bit = column)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int col = x0 + mfma_row(r, lane);
    const bool ok = row_ok && col < W;
    const long pix = ((long)b * H + row) * W + col;
    bool spike = false;
    if (ok) {
      const float v = vpv[r];
      const float z = (float)((zw[r] >> j) & 1u);
      const float cur = acc[r];
      const float vo_hard = (v * lam) * (1.0f - z) + (1.0f - lam) * cur;
      const float vo_soft = v * lam + (1.0f - lam) * cur - z * th;
      const float vo = hard_reset ? vo_hard : vo_soft;  // (a select, not a branch, inside the unrolled pixel loop)
      v_out[pix * C32 + j] = vo;
      spike = (vo - (thx ? th + thx[r] : th)) > 0.f;
    }
    const unsigned long long m = __ballot(spike);
    if (ok && j == 0) z_out[pix] = (lane >> 5) ? (uint32_t)(m >> 32) : (uint32_t)m;
    plane |= (spike ? 1u : 0u) << mfma_row(r, lane);
  }
  if (zT_out) {  // channel-major bit planes [B][H][32][ceil(W/32)] for the weight-gradient kernel
    plane |= __shfl_xor(plane, 32, 64);
    if (row_ok && lane < 32) zT_out[(((long)b * H + row) * C32 + j) * ((W + 31) / 32) + x0 / 32] = plane;
  }
}

template <typename ZPrev>
__device__ __forceinline__ void lif_epilogue(const f32x16& acc, int b, int row, int x0, int H, int W, int lane,
                                             float lam, float th, int hard_reset, const float* __restrict__ v_prev,
                                             ZPrev zprev_word, float* __restrict__ v_out,
                                             uint32_t* __restrict__ z_out, uint32_t* __restrict__ zT_out) {
  float vpv[16];
  uint32_t zw[16];
  lif_load_prev(b, row, x0, H, W, lane, v_prev, v_out, zprev_word, vpv, zw);
  lif_update(acc, vpv, zw, b, row, x0, H, W, lane, lam, th, hard_reset, v_out, z_out, zT_out);
}

template <bool REC>
__global__ __launch_bounds__(256) void k_conv_lif_fwd(const uint32_t* __restrict__ x, const float* __restrict__ wff,
                                                      const float* __restrict__ wrec, const float* __restrict__ leak,
                                                      const float* __restrict__ thresh,
                                                      const float* __restrict__ v_prev,
                                                      const uint32_t* __restrict__ z_prev, int B, int H, int W,
                                                      int hard_reset, float* __restrict__ v_out,
                                                      uint32_t* __restrict__ z_out, uint32_t* __restrict__ zT_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_wff = (float*)smem_raw;                                 // WPACK
  float* s_wrec = s_wff + WPACK;                                   // WPACK (REC only)
  uint32_t* s_x = (uint32_t*)(s_wff + (REC ? 2 : 1) * WPACK);      // HALO_H*HALO_W
  uint32_t* s_z = s_x + HALO_H * HALO_W;                           // HALO_H*HALO_W (REC only)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;

  // stage weights (coalesced float4) and the spike-bit halo tiles
  for (int i = tid; i < WPACK / 4; i += 256) ((float4*)s_wff)[i] = ((const float4*)wff)[i];
  if (REC)
    for (int i = tid; i < WPACK / 4; i += 256) ((float4*)s_wrec)[i] = ((const float4*)wrec)[i];
  for (int i = tid; i < HALO_H * HALO_W; i += 256) {
    const int yy = y0 + i / HALO_W - 1, xx = x0 + i % HALO_W - 1;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const long p = ((long)b * H + yy) * W + xx;
    s_x[i] = in ? x[p] : 0u;
    if (REC) s_z[i] = (in && z_prev) ? z_prev[p] : 0u;
  }
  __syncthreads();

  f32x16 acc0 = {0}, acc1 = {0};
  const int i = lane & 31, sh = (lane >> 5) * 16;
  const int r0 = 2 * wv;  // this wave's two tile rows
#pragma unroll 1
  for (int tau = 0; tau < 9; ++tau) {
    const int dy = tau / 3, dx = tau % 3;
    const uint32_t w0 = s_x[(r0 + dy) * HALO_W + i + dx] >> sh;
    const uint32_t w1 = s_x[(r0 + 1 + dy) * HALO_W + i + dx] >> sh;
    const float* wp = s_wff + tau * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float bw = wp[t * 64];
      acc0 = mfma32((float)((w0 >> t) & 1u), bw, acc0);
      acc1 = mfma32((float)((w1 >> t) & 1u), bw, acc1);
    }
    if (REC) {
      const uint32_t q0 = s_z[(r0 + dy) * HALO_W + i + dx] >> sh;
      const uint32_t q1 = s_z[(r0 + 1 + dy) * HALO_W + i + dx] >> sh;
      const float* wq = s_wrec + tau * 1024 + lane;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float bw = wq[t * 64];
        acc0 = mfma32((float)((q0 >> t) & 1u), bw, acc0);
        acc1 = mfma32((float)((q1 >> t) & 1u), bw, acc1);
      }
    }
  }

  const int j = lane & 31;
  const float lam = evf_sigmoid(leak[j]);        // torch.sigmoid(self.leak)   :111/:536
  const float th = fmaxf(thresh[j], 0.01f);      // self.thresh.clamp_min(0.01) :108/:533
  auto zword = [&](int row, int col) -> uint32_t {
    if (REC) return s_z[(row - y0 + 1) * HALO_W + (col - x0 + 1)];
    return z_prev ? z_prev[((long)b * H + row) * W + col] : 0u;
  };
  lif_epilogue(acc0, b, y0 + r0, x0, H, W, lane, lam, th, hard_reset, v_prev, zword, v_out, z_out, zT_out);
  lif_epilogue(acc1, b, y0 + r0 + 1, x0, H, W, lane, lam, th, hard_reset, v_prev, zword, v_out, z_out, zT_out);
}

extern "C" int evf_conv_lif_fwd(const uint32_t* x, const float* w_ff, const float* w_rec, const float* leak,
                                const float* thresh, const float* v_prev, const uint32_t* z_prev, int B, int H, int W,
                                int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, void* stream) {
  if (!x || !w_ff || !leak || !thresh || !v_out || !z_out || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  dim3 grid(evf_cdiv(W, TW), evf_cdiv(H, TH), B), block(256);
  hipStream_t st = EVF_STREAM(stream);
  if (w_rec) {
    const size_t lds = 2 * WPACK * 4 + 2 * HALO_H * HALO_W * 4;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)k_conv_lif_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    hipLaunchKernelGGL(k_conv_lif_fwd<true>, grid, block, lds, st, x, w_ff, w_rec, leak, thresh, v_prev, z_prev, B, H, W,
                       hard_reset, v_out, z_out, zT_out);
  } else {
    const size_t lds = WPACK * 4 + HALO_H * HALO_W * 4;
    hipLaunchKernelGGL(k_conv_lif_fwd<false>, grid, block, lds, st, x, w_ff, (const float*)nullptr, leak, thresh, v_prev,
                       z_prev, B, H, W, hard_reset, v_out, z_out, zT_out);
  }
  return evf_status();
}

// Head: real-valued NCHW input with few channels (event counts / voxels).
// K = 9 taps x Cin; one MFMA k-step covers channels (2s, 2s+1).
#define HEAD_MAX_CIN 8
template <int S2>  // S2 = ceil(Cin / 2): compile-time trip counts, so that every staging load of a thread is in flight at once
__global__ __launch_bounds__(256) void k_head_lif_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ leak, const float* __restrict__ thresh,
                                                      const float* __restrict__ v_prev,
                                                      const uint32_t* __restrict__ z_prev, int B, int Cin, int H, int W,
                                                      int hard_reset, float* __restrict__ v_out,
                                                      uint32_t* __restrict__ z_out, uint32_t* __restrict__ zT_out,
                                                      const float* __restrict__ leak_pt,
                                                      const float* __restrict__ add_pt,
                                                      const float* __restrict__ pt_prev, float* __restrict__ pt_out,
                                                      float* __restrict__ P_out) {
  __shared__ float s_x[HEAD_MAX_CIN][HALO_H * HALO_W];
  __shared__ float s_w[9 * (HEAD_MAX_CIN / 2) * 64];
  __shared__ float s_P[TH * TW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  // previous membrane potentials and spike words of this wave's two rows: issued first, consumed after the matrix phase
  auto zword = [&](int row, int col) -> uint32_t {
    const uint32_t wd = *(z_prev ? z_prev + ((long)b * H + row) * W + col : (const uint32_t*)v_out);  // no branch around the load
    return z_prev ? wd : 0u;
  };
  float vp0[16], vp1[16];
  uint32_t zw0[16], zw1[16];
  lif_load_prev(b, y0 + 2 * wv, x0, H, W, lane, v_prev, v_out, zword, vp0, zw0);
  lif_load_prev(b, y0 + 2 * wv + 1, x0, H, W, lane, v_prev, v_out, zword, vp1, zw1);
  {
    constexpr int NW = (9 * S2 * 64 + 255) / 256, NX = (2 * S2 * HALO_H * HALO_W + 255) / 256;
    float wreg[NW], xreg[NX];
#pragma unroll
    for (int k = 0; k < NW; ++k) {  // clamped addresses, selected afterwards: no load under a branch
      const int e = min(tid + 256 * k, 9 * S2 * 64 - 1);
      const int l = e & 63, s = (e >> 6) % S2, tau = (e >> 6) / S2;
      const int ci = 2 * s + (l >> 5), j = l & 31;
      const float wv0 = w[(j * Cin + min(ci, Cin - 1)) * 9 + tau];
      wreg[k] = ci < Cin ? wv0 : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int e = min(tid + 256 * k, 2 * S2 * HALO_H * HALO_W - 1);
      const int ci = e / (HALO_H * HALO_W), p = e % (HALO_H * HALO_W);
      const int yy = y0 + p / HALO_W - 1, xx = x0 + p % HALO_W - 1;
      const bool in = ci < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float xv = x[(((long)b * Cin + min(ci, Cin - 1)) * H + min(max(yy, 0), H - 1)) * W + min(max(xx, 0), W - 1)];
      xreg[k] = in ? xv : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NW; ++k)
      if (tid + 256 * k < 9 * S2 * 64) s_w[tid + 256 * k] = wreg[k];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int e = tid + 256 * k;
      if (e < 2 * S2 * HALO_H * HALO_W) s_x[e / (HALO_H * HALO_W)][e % (HALO_H * HALO_W)] = xreg[k];
    }
  }
  __syncthreads();
  f32x16 acc0 = {0}, acc1 = {0};
  const int i = lane & 31, h = lane >> 5, r0 = 2 * wv;
  for (int tau = 0; tau < 9; ++tau) {
    const int dy = tau / 3, dx = tau % 3;
    for (int s = 0; s < S2; ++s) {
      const float bw = s_w[(tau * S2 + s) * 64 + lane];
      const float* xp = s_x[2 * s + h];
      acc0 = mfma32(xp[(r0 + dy) * HALO_W + i + dx], bw, acc0);
      acc1 = mfma32(xp[(r0 + 1 + dy) * HALO_W + i + dx], bw, acc1);
    }
  }
  const int j = lane & 31;
  const float lam = evf_sigmoid(leak[j]);
  const float th = fmaxf(thresh[j], 0.01f);
  const bool xl = pt_out && (hard_reset & 6);  // XLIF / ALIF head (bits 1-2 of the flag, evf_head_plif_fwd): add_pt = t1, thresh = t0
  const bool al = pt_out && ((hard_reset >> 1) & 3) == 2;  // ALIF: the trace is driven by the cell's own previous spikes (:311)
  hard_reset &= 1;
  float thx0[16], thx1[16];  // (XLIF: t1 * pt' per element of the wave's two rows)
  if (pt_out) {  // PLIF head (spiking_submodules.py:191-227): cur = ff - sigma(add_pt) * pt'
    const int py = tid >> 5, px = tid & 31;
    float sum9 = 0.f;
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx) {
        float a = 0.f;
        for (int ci = 0; ci < Cin; ++ci) a += fabsf(s_x[ci][(py + dy) * HALO_W + px + dx]);
        sum9 += a / (float)Cin;  // input_.abs().mean(1)
      }
    const float P = sum9 / 9.0f;  // AvgPool2d(3, 1, 1), count_include_pad
    s_P[tid] = P;
    if (y0 + py < H && x0 + px < W) P_out[((long)b * H + y0 + py) * W + x0 + px] = P;
    __syncthreads();
    const float lpt = evf_plif_sigmoid(leak_pt[j]), apt = xl ? fmaxf(add_pt[j], 0.f) : evf_plif_sigmoid(add_pt[j]);  // (XLIF: t1.clamp_min(0), :365)
    const float* ptsrc = pt_prev ? pt_prev : pt_out;  // dummy source when there is no previous trace
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x16& acc = m ? acc1 : acc0;
      float (&thx)[16] = m ? thx1 : thx0;
      const int row = y0 + r0 + m, rq = min(row, H - 1);
      // the 16 previous-trace values of this lane: unconditional loads from clamped addresses, all in flight
      // together (under `if (ok)` each load is its own round trip: this loop was 32 serial latencies)
      float ptv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ptv[r] = ptsrc[(((long)b * H + rq) * W + min(x0 + mfma_row(r, lane), W - 1)) * C32 + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = mfma_row(r, lane), col = x0 + cl;
        const float zprev = (float)((((m ? zw1 : zw0)[r]) >> j) & 1u);
        const float pto = evf_plif_trace(pt_prev ? ptv[r] : 0.f, lpt, al ? zprev : s_P[(r0 + m) * TW + cl]);
        thx[r] = apt * pto;
        if (!xl) acc[r] = acc[r] - thx[r];  // (XLIF: the current stays ff, the threshold becomes t0 + t1 * pt', :419)
        if (row < H && col < W) pt_out[(((long)b * H + row) * W + col) * C32 + j] = pto;
      }
    }
  }
  if (xl) {  // (block-uniform)
    lif_update(acc0, vp0, zw0, b, y0 + r0, x0, H, W, lane, lam, th, hard_reset, v_out, z_out, zT_out, thx0);
    lif_update(acc1, vp1, zw1, b, y0 + r0 + 1, x0, H, W, lane, lam, th, hard_reset, v_out, z_out, zT_out, thx1);
  } else {
    lif_update(acc0, vp0, zw0, b, y0 + r0, x0, H, W, lane, lam, th, hard_reset, v_out, z_out, zT_out);
    lif_update(acc1, vp1, zw1, b, y0 + r0 + 1, x0, H, W, lane, lam, th, hard_reset, v_out, z_out, zT_out);
  }
}

static void launch_head_fwd(dim3 grid, dim3 block, hipStream_t st, const float* x, const float* w, const float* leak,
                            const float* thresh, const float* v_prev, const uint32_t* z_prev, int B, int Cin, int H, int W,
                            int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, const float* leak_pt,
                            const float* add_pt, const float* pt_prev, float* pt_out, float* P_out) {
#define HEAD_FWD(S2)                                                                                                       \
  hipLaunchKernelGGL(k_head_lif_fwd<S2>, grid, block, 0, st, x, w, leak, thresh, v_prev, z_prev, B, Cin, H, W, hard_reset, \
                     v_out, z_out, zT_out, leak_pt, add_pt, pt_prev, pt_out, P_out)
  switch ((Cin + 1) / 2) {
    case 1: HEAD_FWD(1); break;
    case 2: HEAD_FWD(2); break;
    case 3: HEAD_FWD(3); break;
    default: HEAD_FWD(4); break;
  }
#undef HEAD_FWD
}

// ---- the head layer of a whole WINDOW in one launch -----------------------------------------------------------------
// The head (models/model.py: ConvLIF on the event input) reads only the network input and its OWN state, and that state is
// per pixel: pass t of a tile needs nothing of any other tile's pass t-1.  So when the passes of a window are known up front
// (a forward recording, evf_fwd_defer_*) one block runs ALL passes of its tile: v and the spike words stay in registers
// between the passes (no v_prev / z_prev reads), the input halo of pass t+1 is fetched while pass t computes, and the weights
// are staged once.  Every pass does exactly the arithmetic of k_head_lif_fwd (same MFMA order, same update), so the states
// and spikes are bit-identical to P separate launches.
#define HEADWIN_MAX_P 16
#ifndef HEADWIN_LB
#define HEADWIN_LB 256
#endif
struct HeadWinPass {
  const float* x;
  float* v_out;
  uint32_t* z_out;
  uint32_t* zT_out;
  float *pt_out, *P_out;  // PLIF: the trace after the pass [B,H,W,32], the pooled input activity [B,H,W]
};
struct HeadWin {
  HeadWinPass p[HEADWIN_MAX_P];
  const float *w, *leak, *thresh, *v_prev;
  const uint32_t* z_prev;
  int np, B, Cin, H, W, hard_reset;
  const float *leak_pt, *add_pt, *pt_prev;  // PLIF (pt_prev NULL: zero trace)
};

// PLIF: the presynaptic trace stays in registers between the passes as well (k_head_lif_fwd read it back every pass); every pass
// does the arithmetic of k_head_lif_fwd's PLIF branch: the same bits.
// NWV = waves per block: 4 (a wave takes two rows of the 8 x 32 tile) or 8 (one row each).  With two rows per wave the kernel
// holds potential (+ trace) and accumulators of 32 pixels per lane: 212 registers (LIF) / 288 wanted (PLIF, i.e. one wave per
// SIMD or spills) -- the block then runs at the latency of its own store -> barrier -> matrix chain.  One row per wave halves the
// per-lane state; twice the waves per CU hide each other's chains.  Same MFMA order per row: the same bits.
template <int S2, bool PLIF = false, int NWV = 4>
// (second launch bound = waves per SIMD: LIF x 8 waves 130 -> 128 registers = two blocks per CU; PLIF x 4 waves: two blocks, spills)
__global__ __launch_bounds__(64 * NWV, (NWV == 8 && !PLIF) ? 4 : ((NWV == 4 && PLIF) ? 2 : 1)) void k_head_lif_fwd_win(HeadWin a) {
  constexpr int NTHR = 64 * NWV, RPW = TH / NWV;  // threads, rows per wave
  static_assert(TH == 8 && TW == 32 && (NWV == 4 || NWV == 8), "8 x 32 tile");
  __shared__ float s_x[2][2 * S2][HALO_H * HALO_W];
  __shared__ float s_w[9 * S2 * 64];
  __shared__ float s_P[PLIF ? TH * TW : 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int B = a.B, Cin = a.Cin, H = a.H, W = a.W;
  (void)B;
  const uint32_t* z_prev = a.z_prev;
  auto zword = [&](int row, int col) -> uint32_t {
    const uint32_t wd = *(z_prev ? z_prev + ((long)b * H + row) * W + col : (const uint32_t*)a.p[0].v_out);
    return z_prev ? wd : 0u;
  };
  const int r0 = RPW * wv;
  float vp[RPW][16];
  uint32_t zb[RPW];  // bit r: the previous spike of this lane's channel at its pixel r (all a lane needs of the words)
#pragma unroll
  for (int m = 0; m < RPW; ++m) {
    uint32_t zw[16];
    lif_load_prev(b, y0 + r0 + m, x0, H, W, lane, a.v_prev, a.p[0].v_out, zword, vp[m], zw);
    zb[m] = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) zb[m] |= ((zw[r] >> (lane & 31)) & 1u) << r;
  }
  constexpr int NW = (9 * S2 * 64 + NTHR - 1) / NTHR, NX = (2 * S2 * HALO_H * HALO_W + NTHR - 1) / NTHR;
  float xreg[NX];
  // (the in-image select happens in put_x, from a mask that does not depend on the loaded value: a select right behind the load
  // makes the wave wait for it at once -- and, vmcnt being in order, for every tape store of the pass before: one exposed store
  // drain per pass, which was this kernel's time.  Now the loads of pass t + 1 are only waited for behind pass t's stores, with a
  // count that leaves those stores in flight.)
  unsigned xin = 0u;
  auto fetch_x = [&](const float* __restrict__ x) {  // clamped addresses, selected afterwards: no load under a branch
    xin = 0u;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int e = min(tid + NTHR * k, 2 * S2 * HALO_H * HALO_W - 1);
      const int ci = e / (HALO_H * HALO_W), p = e % (HALO_H * HALO_W);
      const int yy = y0 + p / HALO_W - 1, xx = x0 + p % HALO_W - 1;
      const bool in = ci < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W;
      xreg[k] = x[(((long)b * Cin + min(ci, Cin - 1)) * H + min(max(yy, 0), H - 1)) * W + min(max(xx, 0), W - 1)];
      xin |= (in ? 1u : 0u) << k;
    }
  };
  static_assert(NX <= 32, "one mask bit per staged element");
  auto put_x = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int e = tid + NTHR * k;
      if (e < 2 * S2 * HALO_H * HALO_W) s_x[buf][e / (HALO_H * HALO_W)][e % (HALO_H * HALO_W)] = ((xin >> k) & 1u) ? xreg[k] : 0.f;
    }
  };
  {
    float wreg[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int e = min(tid + NTHR * k, 9 * S2 * 64 - 1);
      const int l = e & 63, s = (e >> 6) % S2, tau = (e >> 6) / S2;
      const int ci = 2 * s + (l >> 5), j = l & 31;
      const float wv0 = a.w[(j * Cin + min(ci, Cin - 1)) * 9 + tau];
      wreg[k] = ci < Cin ? wv0 : 0.f;
    }
    fetch_x(a.p[0].x);
#pragma unroll
    for (int k = 0; k < NW; ++k)
      if (tid + NTHR * k < 9 * S2 * 64) s_w[tid + NTHR * k] = wreg[k];
    put_x(0);
  }
  const int i = lane & 31, h = lane >> 5, j = lane & 31;
  const float lam = evf_sigmoid(a.leak[j]);
  const float th = fmaxf(a.thresh[j], 0.01f);
  const int hard_reset = a.hard_reset & 1;
  const bool xl = PLIF && (a.hard_reset & 6);  // XLIF / ALIF head (bits 1-2 of the flag, evf_head_plif_fwd): add_pt = t1, thresh = t0
  const bool al = PLIF && ((a.hard_reset >> 1) & 3) == 2;  // ALIF: the trace is driven by the cell's own previous spikes
  const int nW = (W + 31) / 32;
  float pt[PLIF ? RPW : 1][16];
  float lpt = 0.f, apt = 0.f;
  if (PLIF) {
    lpt = evf_plif_sigmoid(a.leak_pt[j]), apt = xl ? fmaxf(a.add_pt[j], 0.f) : evf_plif_sigmoid(a.add_pt[j]);  // (XLIF: t1.clamp_min(0), :365)
    const float* ptsrc = a.pt_prev ? a.pt_prev : a.p[0].v_out;  // (dummy source: selected away)
#pragma unroll
    for (int m = 0; m < RPW; ++m) {
      const int rq = min(y0 + r0 + m, H - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = ptsrc[(((long)b * H + rq) * W + min(x0 + mfma_row(r, lane), W - 1)) * C32 + j];
        pt[PLIF ? m : 0][r] = a.pt_prev ? v : 0.f;
      }
    }
  }
  // LIF update of one row (lif_update) that also leaves the new state in vpv / zb for the next pass.  FULL: the tile lies
  // inside the image (block-uniform) -- no per-pixel branch.  The 16 spike words of the row go out in ONE store (lane r of
  // each half wave keeps word r) instead of 16 stores by lanes 0 and 32.
  // ptr: the row's new traces (PLIF / XLIF; XLIF: threshold t0 + t1 * pt' per element, spiking_submodules.py:419, hard reset only)
  auto update = [&](const f32x16& acc, float (&vpv)[16], uint32_t& zbr, int row, const HeadWinPass& o, const bool FULL,
                    const float (&ptr)[16]) {
    const bool row_ok = FULL || row < H;
    uint32_t plane = 0u, znew = 0u, zsel = 0u;
    // one base address per row; pixel r of the lane is a compile-time offset from it ((r & 3) + 8 (r >> 2) pixels)
    float* const vrow = o.v_out + (((long)b * H + row) * W + x0 + 4 * (lane >> 5)) * C32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = x0 + mfma_row(r, lane);
      const bool ok = FULL || (row_ok && col < W);
      bool spike = false;
      if (ok) {
        const float v = vpv[r];
        const float z = (float)((zbr >> r) & 1u);
        const float cur = acc[r];
        const float vo_hard = (v * lam) * (1.0f - z) + (1.0f - lam) * cur;
        const float vo_soft = v * lam + (1.0f - lam) * cur - z * th;
        const float vo = hard_reset ? vo_hard : vo_soft;
        vrow[((r & 3) + 8 * (r >> 2)) * C32] = vo;
        spike = (vo - ((PLIF && xl) ? th + apt * ptr[r] : th)) > 0.f;
        vpv[r] = vo;
      }
      const unsigned long long m = __ballot(spike);
      const uint32_t mw = (lane >> 5) ? (uint32_t)(m >> 32) : (uint32_t)m;
      zsel = j == r ? mw : zsel;
      znew |= (spike ? 1u : 0u) << r;  // (pixels outside the image: never stored, never spiking)
      plane |= (spike ? 1u : 0u) << mfma_row(r, lane);
    }
    zbr = znew;
    if (j < 16) {
      const int col = x0 + mfma_row(j, lane);
      if (row_ok && (FULL || col < W)) o.z_out[((long)b * H + row) * W + col] = zsel;
    }
    if (o.zT_out) {
      plane |= __shfl_xor(plane, 32, 64);
      if (row_ok && lane < 32) o.zT_out[(((long)b * H + row) * C32 + j) * nW + x0 / 32] = plane;
    }
  };
  const bool full = y0 + TH <= H && x0 + TW <= W;
  for (int t = 0; t < a.np; ++t) {
    const int buf = t & 1;
    if (t + 1 < a.np) fetch_x(a.p[t + 1].x);  // (in flight under this pass's matrix phase and stores)
    __syncthreads();  // s_x[buf] (and, at t = 0, s_w) written; every wave is done reading s_x[buf ^ 1] (pass t - 1)
    f32x16 acc[RPW];
#pragma unroll
    for (int m = 0; m < RPW; ++m) acc[m] = f32x16{0};
#pragma unroll 3
    for (int tau = 0; tau < 9; ++tau) {
      const int dy = tau / 3, dx = tau % 3;
#pragma unroll
      for (int s = 0; s < S2; ++s) {
        const float bw = s_w[(tau * S2 + s) * 64 + lane];
        const float* xp = s_x[buf][2 * s + h];
#pragma unroll
        for (int m = 0; m < RPW; ++m) acc[m] = mfma32(xp[(r0 + m + dy) * HALO_W + i + dx], bw, acc[m]);
      }
    }
    // the next pass's input halo into the other buffer BEFORE this pass's tape stores are issued: the wait for its loads (in
    // order: it also covers the stores of pass t - 1, which have had a whole matrix phase to drain) then leaves pass t's stores
    // in flight across the barrier and the next matrix phase.  (s_x[buf ^ 1] was last read in pass t - 1, before this pass's barrier.)
    if (t + 1 < a.np) put_x(buf ^ 1);
    if (PLIF) {  // cur = ff - sigma(add_pt) * pt' (k_head_lif_fwd, spiking_submodules.py:191-227)
      if (tid < TH * TW) {
        const int py = tid >> 5, px = tid & 31;
        float sum9 = 0.f;
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx) {
            float av = 0.f;
            for (int ci = 0; ci < Cin; ++ci) av += fabsf(s_x[buf][ci][(py + dy) * HALO_W + px + dx]);
            sum9 += av / (float)Cin;
          }
        const float Pv = sum9 / 9.0f;
        s_P[tid] = Pv;
        if (y0 + py < H && x0 + px < W) a.p[t].P_out[((long)b * H + y0 + py) * W + x0 + px] = Pv;
      }
      __syncthreads();  // (the next pass writes s_P behind the loop's barrier)
#pragma unroll
      for (int m = 0; m < RPW; ++m) {
        const int row = y0 + r0 + m;
        float* const prow = a.p[t].pt_out + (((long)b * H + min(row, H - 1)) * W + x0) * C32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = mfma_row(r, lane), col = x0 + cl;
          const float pto = evf_plif_trace(pt[PLIF ? m : 0][r], lpt, al ? (float)((zb[m] >> r) & 1u) : s_P[(r0 + m) * TW + cl]);
          if (!xl) acc[m][r] = acc[m][r] - apt * pto;  // (XLIF / ALIF: the current stays ff)
          pt[PLIF ? m : 0][r] = pto;
          if (row < H && col < W) prow[cl * C32] = pto;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < RPW; ++m) {
      if (full) update(acc[m], vp[m], zb[m], y0 + r0 + m, a.p[t], true, pt[PLIF ? m : 0]);
      else update(acc[m], vp[m], zb[m], y0 + r0 + m, a.p[t], false, pt[PLIF ? m : 0]);
    }
  }
}

// head cells recorded by a forward recording (evf_fwd_defer_*): launched by evf_hf_defer_launch before the diagonals
struct HfJob {
  const float *x, *w, *leak, *thresh, *v_prev;
  const uint32_t* z_prev;
  int B, Cin, H, W, hard_reset;
  float* v_out;
  uint32_t *z_out, *zT_out;
  const float *leak_pt, *add_pt, *pt_prev;  // PLIF head (pt_out != NULL)
  float *pt_out, *P_out;
};
#define HF_MAX_JOBS 96
struct HfDefer {
  int n = 0;
  HfJob job[HF_MAX_JOBS];
};
static HfDefer hf_tab[EVF_CTX_MAX];
int evf_hf_defer_count(int ctx) { return hf_tab[ctx].n; }
void evf_hf_defer_reset(int ctx) { hf_tab[ctx].n = 0; }

static void head_fwd_one(const HfJob& q, hipStream_t st) {
  dim3 grid(evf_cdiv(q.W, TW), evf_cdiv(q.H, TH), q.B), block(256);
  launch_head_fwd(grid, block, st, q.x, q.w, q.leak, q.thresh, q.v_prev, q.z_prev, q.B, q.Cin, q.H, q.W, q.hard_reset, q.v_out,
                  q.z_out, q.zT_out, q.leak_pt, q.add_pt, q.pt_prev, q.pt_out, q.P_out);
}
int evf_hf_defer_launch(int ctx, void* stream) {
  HfDefer& hf = hf_tab[ctx];
  static const bool window = []() {  // EVF_HEAD_WIN=0: the recorded head cells one launch each (A/B measurements)
    const char* e = getenv("EVF_HEAD_WIN");
    return !(e && e[0] == '0');
  }();
  hipStream_t st = EVF_STREAM(stream);
  int k = 0;
  while (k < hf.n) {
    // longest run of cells that continue each other: pass t + 1 starts from the state pass t wrote, same layer and geometry
    int m = 1;
    while (window && k + m < hf.n && m < HEADWIN_MAX_P) {
      const HfJob &p = hf.job[k + m - 1], &q = hf.job[k + m];
      if (!(q.v_prev == p.v_out && q.z_prev == p.z_out && q.w == p.w && q.leak == p.leak && q.thresh == p.thresh && q.B == p.B &&
            q.Cin == p.Cin && q.H == p.H && q.W == p.W && q.hard_reset == p.hard_reset))
        break;
      if ((q.pt_out != nullptr) != (p.pt_out != nullptr)) break;
      if (q.pt_out && !(q.pt_prev == p.pt_out && q.leak_pt == p.leak_pt && q.add_pt == p.add_pt)) break;
      ++m;
    }
    const HfJob& f = hf.job[k];
    if (m == 1) {
      head_fwd_one(f, st);
    } else {
      evf_prof_mark(4, 0, stream);
      HeadWin a;
      for (int t = 0; t < HEADWIN_MAX_P; ++t) {
        const HfJob& q = hf.job[k + (t < m ? t : 0)];
        a.p[t] = HeadWinPass{q.x, q.v_out, q.z_out, q.zT_out, q.pt_out, q.P_out};
      }
      a.w = f.w, a.leak = f.leak, a.thresh = f.thresh, a.v_prev = f.v_prev, a.z_prev = f.z_prev;
      a.np = m, a.B = f.B, a.Cin = f.Cin, a.H = f.H, a.W = f.W, a.hard_reset = f.hard_reset;
      a.leak_pt = f.leak_pt, a.add_pt = f.add_pt, a.pt_prev = f.pt_prev;
      static const int waves = []() {  // EVF_HEAD_FWD_WAVES=4: two rows of the tile per wave (A/B; default 8: one row each)
        const char* e = getenv("EVF_HEAD_FWD_WAVES");
        return (e && e[0] == '4') ? 4 : 8;
      }();
      dim3 grid(evf_cdiv(f.W, TW), evf_cdiv(f.H, TH), f.B), block(64 * waves);
#define HEAD_FWD_WIN(S2_, PLIF_)                                                                          \
  do {                                                                                                    \
    if (waves == 8) hipLaunchKernelGGL((k_head_lif_fwd_win<S2_, PLIF_, 8>), grid, block, 0, st, a);        \
    else hipLaunchKernelGGL((k_head_lif_fwd_win<S2_, PLIF_, 4>), grid, block, 0, st, a);                   \
  } while (0)
      if (f.pt_out) {
        switch ((f.Cin + 1) / 2) {
          case 1: HEAD_FWD_WIN(1, true); break;
          case 2: HEAD_FWD_WIN(2, true); break;
          case 3: HEAD_FWD_WIN(3, true); break;
          default: HEAD_FWD_WIN(4, true); break;
        }
      } else {
        switch ((f.Cin + 1) / 2) {
          case 1: HEAD_FWD_WIN(1, false); break;
          case 2: HEAD_FWD_WIN(2, false); break;
          case 3: HEAD_FWD_WIN(3, false); break;
          default: HEAD_FWD_WIN(4, false); break;
        }
      }
#undef HEAD_FWD_WIN
      evf_prof_mark(4, 1, stream);
    }
    k += m;
  }
  hf.n = 0;
  return evf_status();
}

extern "C" int evf_head_lif_fwd(const float* x, const float* w, const float* leak, const float* thresh,
                                const float* v_prev, const uint32_t* z_prev, int B, int Cin, int H, int W,
                                int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, void* stream) {
  if (!x || !w || !leak || !thresh || !v_out || !z_out || B <= 0 || Cin <= 0 || Cin > HEAD_MAX_CIN || H <= 0 || W <= 0)
    return EVF_EINVAL;
  const int fctx = evf_ctx_find(stream);
  if (fctx >= 0 && evf_fwd_defer_active(fctx)) {  // recorded: launched (all passes of the window in one launch) by the flush
    HfDefer& hf = hf_tab[fctx];
    if (hf.n == HF_MAX_JOBS) {
      const int rc = evf_hf_defer_launch(fctx, stream);
      if (rc) return rc;
    }
    hf.job[hf.n++] = HfJob{x, w, leak, thresh, v_prev, z_prev, B, Cin, H, W, hard_reset, v_out, z_out, zT_out,
                           nullptr, nullptr, nullptr, nullptr, nullptr};
    if (evf_defer_poisoned()) {
      const size_t npix = (size_t)B * H * W;
      int rc = evf_hip(evf_memset_async(v_out, 0xFF, npix * C32 * sizeof(float), EVF_STREAM(stream)));
      if (!rc) rc = evf_hip(evf_memset_async(z_out, 0xFF, npix * sizeof(uint32_t), EVF_STREAM(stream)));
      if (rc) return rc;
    }
    return EVF_OK;
  }
  dim3 grid(evf_cdiv(W, TW), evf_cdiv(H, TH), B), block(256);
  launch_head_fwd(grid, block, EVF_STREAM(stream), x, w, leak, thresh, v_prev, z_prev, B, Cin, H, W, hard_reset, v_out, z_out,
                  zT_out, nullptr, nullptr, nullptr, nullptr, nullptr);
  return evf_status();
}

extern "C" int evf_head_plif_fwd(const float* x, const float* w, const float* leak_v, const float* leak_pt,
                                 const float* add_pt, const float* thresh, const float* v_prev, const uint32_t* z_prev,
                                 const float* pt_prev, int B, int Cin, int H, int W, int hard_reset, float* v_out,
                                 uint32_t* z_out, uint32_t* zT_out, float* pt_out, float* P_out, void* stream) {
  if (!x || !w || !leak_v || !leak_pt || !add_pt || !thresh || !v_out || !z_out || !pt_out || !P_out || B <= 0 ||
      Cin <= 0 || Cin > HEAD_MAX_CIN || H <= 0 || W <= 0)
    return EVF_EINVAL;
  if ((hard_reset & 6) && !(hard_reset & 1)) return EVF_ENOTSUP;  // (an XLIF / ALIF head with the soft reset: its threshold of the pass before is not kept here)
  const int fctx = evf_ctx_find(stream);
  if (fctx >= 0 && evf_fwd_defer_active(fctx)) {  // recorded like evf_head_lif_fwd: the window's passes in one launch at the flush
    HfDefer& hf = hf_tab[fctx];
    if (hf.n == HF_MAX_JOBS) {
      const int rc = evf_hf_defer_launch(fctx, stream);
      if (rc) return rc;
    }
    hf.job[hf.n++] = HfJob{x, w, leak_v, thresh, v_prev, z_prev, B, Cin, H, W, hard_reset, v_out, z_out, zT_out,
                           leak_pt, add_pt, pt_prev, pt_out, P_out};
    if (evf_defer_poisoned()) {
      const size_t npix = (size_t)B * H * W;
      int rc = evf_hip(evf_memset_async(v_out, 0xFF, npix * C32 * sizeof(float), EVF_STREAM(stream)));
      if (!rc) rc = evf_hip(evf_memset_async(z_out, 0xFF, npix * sizeof(uint32_t), EVF_STREAM(stream)));
      if (!rc) rc = evf_hip(evf_memset_async(pt_out, 0xFF, npix * C32 * sizeof(float), EVF_STREAM(stream)));
      if (!rc) rc = evf_hip(evf_memset_async(P_out, 0xFF, npix * sizeof(float), EVF_STREAM(stream)));
      if (rc) return rc;
    }
    return EVF_OK;
  }
  dim3 grid(evf_cdiv(W, TW), evf_cdiv(H, TH), B), block(256);
  launch_head_fwd(grid, block, EVF_STREAM(stream), x, w, leak_v, thresh, v_prev, z_prev, B, Cin, H, W, hard_reset, v_out,
                  z_out, zT_out, leak_pt, add_pt, pt_prev, pt_out, P_out);
  return evf_status();
}

// --------------------------------------------------------------------------
// PLIF trace backward (elementwise).  With g_cur = dL/d(current) from the neuron backward:
//   g_pt  = g_pt_carry - sigma(add_pt) * g_cur           (pt' enters the current with -add_pt)
//   g_pt_prev = g_pt * sigma(leak_pt)                    -> carry to the previous pass
//   g_P[pix]  = sum_c g_pt[c] * (1 - sigma(leak_pt[c]))  -> gradient on the pooled activity
//   d add_pt  = ap(1-ap) * sum(-g_cur * pt'),  d leak_pt = lp(1-lp) * sum(g_pt * (pt - P))
// autograd of spiking_submodules.py:204-222 / :634-652.
// --------------------------------------------------------------------------
__global__ void k_plif_trace_bwd(const float4* __restrict__ g_cur, const float4* __restrict__ g_pt_carry,
                                 const float4* __restrict__ pt_prev, const float4* __restrict__ pt_out,
                                 const float* __restrict__ P, const float* __restrict__ leak_pt,
                                 const float* __restrict__ add_pt, long npix, float4* __restrict__ g_pt_prev,
                                 float* __restrict__ g_P, float* __restrict__ g_leak_pt,
                                 float* __restrict__ g_add_pt, int row_ld) {
  __shared__ float s_red[2][4][C32];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, cg = tid & 7;
  float lp[4], ap[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lp[k] = evf_plif_sigmoid(leak_pt[4 * cg + k]);
    ap[k] = evf_plif_sigmoid(add_pt[4 * cg + k]);
  }
  float sl[4] = {0, 0, 0, 0}, sa[4] = {0, 0, 0, 0};
  // TWO grid strides per trip: all eight loads of a thread's two elements are requested before the first is used (one element per
  // trip kept ~3 KB per wave in flight: 41.6 us per launch at 260 x 346 x B4 = 0.56 of the HBM peak, 70 launches per PLIF step)
  const long total = npix * 8, gs = (long)gridDim.x * blockDim.x;
  for (long e0 = (long)blockIdx.x * blockDim.x; e0 < total; e0 += 2 * gs) {
    long ev[2], ecv[2];
    bool okv[2];
    float4 gc4v[2], gklv[2], pplv[2];
    float Pvv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ev[u] = e0 + u * gs + tid;
      okv[u] = ev[u] < total;
      ecv[u] = okv[u] ? ev[u] : total - 1;
      gc4v[u] = g_cur[ecv[u]];
      // optional tensors: load from a valid dummy, select afterwards (no load under a branch)
      gklv[u] = (g_pt_carry ? g_pt_carry : g_cur)[ecv[u]];
      pplv[u] = (pt_prev ? pt_prev : g_cur)[ecv[u]];
      Pvv[u] = P[ecv[u] >> 3];
    }
    (void)pt_out;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long e = ev[u], pix = ecv[u] >> 3;
      const bool ok = okv[u];
      const float4 gc4 = gc4v[u];
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 gk4 = g_pt_carry ? gklv[u] : z4, pp4 = pt_prev ? pplv[u] : z4;
      const float Pv = Pvv[u];
      const float gc[4] = {gc4.x, gc4.y, gc4.z, gc4.w};
      const float gk[4] = {gk4.x, gk4.y, gk4.z, gk4.w}, pp[4] = {pp4.x, pp4.y, pp4.z, pp4.w};
      // pt' of the forward pass is RECOMPUTED from its two operands (pt_prev is read anyway, P is one word per pixel) with the
      // forward's own expression (evf_fwd_b3.hip: pto = p * lpt + (1 - lpt) * P): 128 of the kernel's 640 B/px less
      float po[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) po[k] = evf_plif_trace(pp[k], lp[k], Pv);
      float gp[4], gPp = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = gk[k] - ap[k] * gc[k];
        gp[k] = g * lp[k];
        gPp += g * (1.0f - lp[k]);
        if (ok) {
          sl[k] += g * (pp[k] - Pv);
          sa[k] -= gc[k] * po[k];
        }
      }
      if (ok) g_pt_prev[e] = make_float4(gp[0], gp[1], gp[2], gp[3]);
      // the 8 threads of a pixel hold its 32 channels
      gPp += __shfl_xor(gPp, 1, 64);
      gPp += __shfl_xor(gPp, 2, 64);
      gPp += __shfl_xor(gPp, 4, 64);
      if (ok && cg == 0) g_P[pix] = gPp;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      sl[k] += __shfl_xor(sl[k], o, 64);
      sa[k] += __shfl_xor(sa[k], o, 64);
    }
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_red[0][wv][4 * lane + k] = sl[k];
      s_red[1][wv][4 * lane + k] = sa[k];
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, c = tid & 31;
    float v = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_red[which][w][c];
    const float sgm = evf_sigmoid(which == 0 ? leak_pt[c] : add_pt[c]);
    float* dst = (which == 0 ? g_leak_pt : g_add_pt) + c;
    if (row_ld) dst[(size_t)blockIdx.x * row_ld] += v * sgm * (1.0f - sgm);  // (per-block rows, see k_lif_bwd_wgrad)
    else evf_atomic_add(dst, v * sgm * (1.0f - sgm));
  }
}

// AvgPool3x3^T of g_P (zero padded) / 32: the factor the input-gradient kernel adds where an input spike is set
__global__ void k_plif_box(const float* __restrict__ g, int B, int H, int W, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int xx = (int)(idx % W), yy = (int)((idx / W) % H);
  const long base = idx - (long)yy * W - xx;
  float s = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int y2 = yy + dy, x2 = xx + dx;
      if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) s += g[base + (long)y2 * W + x2];
    }
  out[idx] = (s / 9.0f) / 32.0f;
}

// g_P_raw [B,H,W]: scratch (d loss / d pooled activity); g_P_in [B,H,W]: its adjoint through the pooling and the
// channel mean, i.e. d loss / d(input spike) contributed by the trace wherever the spike is set
// (operand of evf_conv_dgrad_b3).
extern "C" int evf_plif_trace_bwd(const float* g_cur, const float* g_pt_carry, const float* pt_prev, const float* pt_out,
                                  const float* P, const float* leak_pt, const float* add_pt, int B, int H, int W,
                                  float* g_pt_prev, float* g_P_raw, float* g_P_in, float* g_leak_pt, float* g_add_pt,
                                  int row_ld, void* stream) {
  (void)pt_out;  // (not read since round 5: pt' is recomputed from pt_prev and P with evf_plif_trace, the forward's expression; may be NULL)
  if (!g_cur || !P || !leak_pt || !add_pt || !g_pt_prev || !g_P_raw || !g_leak_pt || !g_add_pt ||
      B <= 0 || H <= 0 || W <= 0)
    return EVF_EINVAL;
  const long npix = (long)B * H * W;
  const int nblk = (int)((npix * 8 + 255) / 256 < 512 ? (npix * 8 + 255) / 256 : 512);
  hipLaunchKernelGGL(k_plif_trace_bwd, dim3(nblk), dim3(256), 0, EVF_STREAM(stream), (const float4*)g_cur,
                     (const float4*)g_pt_carry, (const float4*)pt_prev, (const float4*)pt_out, P, leak_pt, add_pt, npix,
                     (float4*)g_pt_prev, g_P_raw, g_leak_pt, g_add_pt, row_ld);
  // (g_P_in == NULL: the input-gradient kernels apply the pooling's adjoint to the raw map themselves, `accumulate | 2`)
  if (g_P_in) hipLaunchKernelGGL(k_plif_box, dim3(evf_cdiv(npix, 256)), dim3(256), 0, EVF_STREAM(stream), g_P_raw, B, H, W, g_P_in);
  return evf_status();
}

// --------------------------------------------------------------------------
// neuron backward (elementwise, 4 channels per thread)
// --------------------------------------------------------------------------
__device__ __forceinline__ float evf_surrogate(int kind, float x, float width) {
  // models/spiking_util.py:38-43 (superspike), :55-65 (multi-gauss), :74-79 (triangle), :88-93 (arctan)
  switch (kind) {
    case EVF_SUPERSPIKE: {
      const float d = 1.0f + width * fabsf(x);
      return __builtin_amdgcn_rcpf(d * d);
    }
    case EVF_TRIANGLE:
      return fmaxf(0.f, 1.0f - width * fabsf(x));
    case EVF_MULTIGAUSS: {
      const float s1 = width, s2 = 6.f * width;
      const float k = 0.3989422804014327f;  // 1/sqrt(2*pi)
      auto gs = [&](float v, float mu, float sg) { return expf(-((v - mu) * (v - mu)) / (2.f * sg * sg)) / sg * k; };
      return 1.15f * gs(x, 0.f, s1) - 0.15f * gs(x, width, s2) - 0.15f * gs(x, -width, s2);
    }
    default:
      return __builtin_amdgcn_rcpf(1.0f + width * x * x);  // v_rcp_f32 (1 ulp): the kernel is VALU bound
  }
}

__global__ __launch_bounds__(256) void k_lif_bwd(const float4* __restrict__ g_z_out, const float4* __restrict__ g_v_out,
                          const float4* __restrict__ v_out, const float4* __restrict__ v_prev,
                          const uint32_t* __restrict__ z_prev, const float* __restrict__ leak,
                          const float* __restrict__ thresh, long npix, int hard_reset, int surrogate, float width,
                          float4* __restrict__ g_cur, float4* __restrict__ g_v_prev, float* __restrict__ g_leak,
                          float* __restrict__ g_thresh) {
  __shared__ float s_red[2][4][C32];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cg = tid & 7;  // channel group: channels 4cg..4cg+3
  float lam[4], th[4], oml[4], inv_oml[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lam[k] = evf_sigmoid(leak[4 * cg + k]);
    th[k] = fmaxf(thresh[4 * cg + k], 0.01f);
    oml[k] = 1.0f - lam[k];
    inv_oml[k] = 1.0f / oml[k];  // per-channel constant: no division in the element loop
  }
  float sl[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0};
  // optional tensors are redirected to v_out and zeroed by a select: no branch around a load (each would end
  // its basic block with s_waitcnt vmcnt(0))
  const float4* pgz = g_z_out ? g_z_out : v_out;
  const float4* pgv = g_v_out ? g_v_out : v_out;
  const float4* pvp = v_prev ? v_prev : v_out;
  const uint32_t* pzw = z_prev ? z_prev : (const uint32_t*)v_out;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long e = (long)blockIdx.x * blockDim.x + tid; e < npix * 8; e += (long)gridDim.x * blockDim.x) {
    const long pix = e >> 3;
    const float4 vo4 = v_out[e];
    const float4 gzl = pgz[e], gvl = pgv[e], vpl = pvp[e];
    const uint32_t zwl = pzw[pix];
    const float4 gz4 = g_z_out ? gzl : zero4, gv4 = g_v_out ? gvl : zero4, vp4 = v_prev ? vpl : zero4;
    const uint32_t zw = z_prev ? (zwl >> (4 * cg)) : 0u;
    const float vo[4] = {vo4.x, vo4.y, vo4.z, vo4.w}, gz[4] = {gz4.x, gz4.y, gz4.z, gz4.w};
    const float gvo[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, vp[4] = {vp4.x, vp4.y, vp4.z, vp4.w};
    float gc[4], gp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float z = (float)((zw >> k) & 1u);
      const float sg = evf_surrogate(surrogate, vo[k] - th[k], width);
      const float gsp = gz[k] * sg;          // through the spike: d z'/d(v'-th)
      const float gv = gvo[k] + gsp;         // total gradient on v'
      gc[k] = gv * oml[k];                   // -> input current (ff + rec)
      float cur, dlam;
      if (hard_reset) {
        gp[k] = gv * lam[k] * (1.0f - z);    // z detached (:539-540)
        cur = (vo[k] - (vp[k] * lam[k]) * (1.0f - z)) * inv_oml[k];
        dlam = vp[k] * (1.0f - z) - cur;
      } else {
        gp[k] = gv * lam[k];
        cur = (vo[k] - vp[k] * lam[k] + z * th[k]) * inv_oml[k];
        dlam = vp[k] - cur;
        st[k] -= gv * z;                      // - z * thresh term
      }
      sl[k] += gv * dlam;
      st[k] -= gsp;                           // spike fn sees (v' - thresh)
    }
    if (g_cur) g_cur[e] = make_float4(gc[0], gc[1], gc[2], gc[3]);
    g_v_prev[e] = make_float4(gp[0], gp[1], gp[2], gp[3]);
  }
  // reduce over the 8 lanes-per-pixel pattern: lanes with equal (lane & 7) share channels
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      sl[k] += __shfl_xor(sl[k], o, 64);
      st[k] += __shfl_xor(st[k], o, 64);
    }
  }
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_red[0][wv][4 * lane + k] = sl[k];
      s_red[1][wv][4 * lane + k] = st[k];
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, c = tid & 31;
    float v = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_red[which][w][c];
    if (which == 0) {
      const float l = evf_sigmoid(leak[c]);
      evf_atomic_add(g_leak + c, v * l * (1.0f - l));  // d sigmoid
    } else if (thresh[c] > 0.01f) {                     // clamp_min passes gradient only above the floor
      evf_atomic_add(g_thresh + c, v);
    }
  }
}

extern "C" int evf_lif_bwd(const float* g_z_out, const float* g_v_out, const float* v_out, const float* v_prev,
                           const uint32_t* z_prev, const float* leak, const float* thresh, int B, int H, int W,
                           int hard_reset, int surrogate, float act_width, float* g_cur, float* g_v_prev,
                           float* g_leak, float* g_thresh, void* stream) {
  if (!v_out || !leak || !thresh || !g_cur || !g_v_prev || !g_leak || !g_thresh || B <= 0 || H <= 0 || W <= 0)
    return EVF_EINVAL;
  const long npix = (long)B * H * W;
  // few, fat blocks: every block ends with 64 global atomics on the same 64 addresses
  const int nblk = (int)((npix * 8 + 255) / 256 < 768 ? (npix * 8 + 255) / 256 : 768);
  hipLaunchKernelGGL(k_lif_bwd, dim3(nblk), dim3(256), 0, EVF_STREAM(stream), (const float4*)g_z_out,
                     (const float4*)g_v_out, (const float4*)v_out, (const float4*)v_prev, z_prev, leak, thresh, npix,
                     hard_reset, surrogate, act_width, (float4*)g_cur, (float4*)g_v_prev, g_leak, g_thresh);
  return evf_status();
}

// Head layer, matrix-core form: neuron backward as in k_lif_bwd, and the head's weight gradient
//   dW[co][(ci, tap)] += sum_pix g_cur[pix][co] * x[b][ci][pix + tap]        (<= 32 (ci, tap) columns: Cin <= 3)
// as a 32 x 32 x (pixels) product on v_mfma_f32_32x32x2_f32: per wave and trip its 8 pixels of g_cur go through
// LDS into A-operand order (lane (co, k) <- pixel 2m + k), the B operand x[pixel 2m + k][(ci, tap) = lane & 31] is
// read straight from the input (one load per MFMA and lane).  16 accumulators instead of the 36 Cin sums per thread
// of the earlier VALU form: ~3x the occupancy, which is what hides the HBM latency of this kernel.
// FAST: arctan surrogate + hard reset (the reference's default neuron) fixed at compile time -- no `switch` / `if` per
// channel in the element-wise part (see k_lif_bwd_wgrad).
// (No __restrict__ on the pointers: k_head_bwd_win runs this body once per pass of a window, and a pass reads what the SAME
//  thread wrote in the pass before -- g_v_prev -> g_v_out, the slab and the per-channel rows; program order keeps that right.)
struct HeadBwdPass {  // one pass of k_head_bwd_win
  const float4 *g_z_out, *g_v_out, *v_out, *v_prev;
  const uint32_t* z_prev;
  const float* x_in;
  float4* g_v_prev;
  int slab_acc;
  // PLIF (HeadPlifPass below): carry from pass t + 1, trace of pass t - 1, pooled activity (NULL: a LIF cell), carry to pass t - 1
  const float4 *g_pt_out, *pt_prev;
  const float* P;
  float4* g_pt_prev;
};
struct HeadTripIn {  // what one trip loads
  float4 vo4, gvl, gzl, vpl;
  uint32_t zwl;
  float xb[4];
  float4 gkl, ppl;  // PLIF: dL/d(pt') carried from pass t + 1 (first pass / NT = 0 only), pt of pass t - 1
  float Pl;         // PLIF: pooled input activity of the pixel
  float4 gxl;       // ALIF: the part of dL/d(spikes) that the pass after sent through the threshold trace
};
// PLIF head (spiking_submodules.py:191-227 / :634-652): the presynaptic trace's backward in the same pass (evf_plif_trace_bwd read
// g_cur back from HBM in a launch of its own).  The head's input is the event tensor: dL/d(pooled activity) is not needed.
struct HeadPlifPass {
  const float4 *g_pt_out, *pt_prev;  // carry from pass t + 1 (NULL: none), trace of pass t - 1 (NULL: zero state)
  const float* P;                    // [B,H,W] pooled activity of pass t (NULL: not a PLIF cell).  ALIF head (XL = 2): instead the
                                     // buffer g_zx [B,H,W,32] -- (1 - sigma(leak_t)) * dL/d(t') of pass t + 1 is READ from it as a second
                                     // part of dL/d(spikes) (when g_pt_out != NULL: there is a pass after), this pass's WRITTEN to it
  float4* g_pt_prev;                 // carry to pass t - 1
};
struct HeadPlifPrm {
  const float *leak_pt, *add_pt;
  float *g_leak_pt, *g_add_pt;
};
// NT > 0 (k_head_bwd_win, a block makes at most NT trips): the carried gradient and the membrane potential of the pass before
// stay in REGISTERS between the passes of the launch -- gvc[trip] = dL/dv written by the previous pass (read instead of
// g_v_out), voc[trip] = the v_prev it loaded, which IS this pass's v_out.  FIRST: the launch's first pass loads both from
// memory; store_gv: the launch's last pass writes the carried gradient out.  Same values either way: bit-identical.
// What a block keeps across the passes of k_head_bwd_win (NT > 0): besides the carried tensors the per-channel constants and
// the SUMS -- weight-gradient accumulator and per-channel partial sums run over all passes of the launch and are reduced
// across the block's waves once, after the last pass (equal to the pass-by-pass sums to fp32 round-off, not bit for bit).
struct HeadBwdKeep {
  f32x16 acc;
  float sl[4], st[4], lam[4], th[4], oml[4], inv_oml[4];
};
struct HeadBwdKeepPlif {
  float slp[4], sap[4], lpt[4], apt[4];
};
// XL (PLIF bodies only): an XLIF head -- add_pt = t1, thresh = t0; the trace raised the THRESHOLD (t0 + t1 * pt',
// spiking_submodules.py:419), so it takes -t1 * dL/d(thresh) instead of -sigma(add_pt) * dL/d(current).  A template parameter: with
// a run-time flag the PLIF window kernel kept eight more values live across the element loop and spilled (401 -> 628 us per window)
// XL = 2: an ALIF head (spiking_submodules.py:230-334) -- the XLIF arithmetic with the trace driven by the cell's OWN previous spikes
// (un-detached, :311): their gradient (1 - sigma(leak_t)) * dL/d(t') goes to the pass before through the buffer HeadPlifPass::P.
template <bool FAST, int NT = 0, bool FIRST = true, bool PLIF = false, int XL = 0>
__device__ __forceinline__ void head_bwd_pass(
    const float4* g_z_out, const float4* g_v_out, const float4* v_out, const float4* v_prev, const uint32_t* z_prev,
    const float* __restrict__ leak, const float* __restrict__ thresh, long npix, int hard_reset_rt, int surrogate_rt, float width,
    float4* g_cur, float4* g_v_prev, float* g_leak, float* g_thresh, const float* __restrict__ x_in, int Cin, int H, int W,
    float* slab, int slab_acc, int row_ld, float4 (&gvc)[NT ? NT : 1], float4 (&voc)[NT ? NT : 1], int (&xo)[NT ? NT : 1][4],
    HeadBwdKeep& K, bool store_gv,  // store_gv: the launch's last pass (NT = 0: every pass is first and last)
    const HeadPlifPass pq = HeadPlifPass{}, const HeadPlifPrm pm = HeadPlifPrm{}, float4* gpc_ = nullptr, HeadBwdKeepPlif* KP_ = nullptr) {
  const int hard_reset = FAST ? 1 : (hard_reset_rt & 1), surrogate = FAST ? EVF_ARCTAN : surrogate_rt;
  constexpr bool xl = PLIF && XL != 0, al = PLIF && XL == 2;
  float4* const gzxb = al ? (float4*)pq.P : nullptr;
  const bool has_gx = al && (FIRST ? pq.g_pt_out != nullptr : true);  // (a pass after this one exists: its g_zx is in the buffer)
  float4 gpc_none[1];
  HeadBwdKeepPlif kp_none;
  float4* gpc = PLIF ? gpc_ : gpc_none;          // [NT] the carried dL/d(pt') of the block's trips
  HeadBwdKeepPlif& KP = PLIF ? *KP_ : kp_none;
  __shared__ float s_red[2][4][C32];
  __shared__ __attribute__((aligned(16))) float s_g[2][4][8 * C32];  // [buffer][wave][pixel][channel]
  __shared__ float s_d[4][C32 * C32];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cg = tid & 7;  // channel group of the element-wise part: channels 4cg..4cg+3
  const int i = lane & 31, kg = lane >> 5;
  float(&lam)[4] = K.lam, (&th)[4] = K.th, (&oml)[4] = K.oml, (&inv_oml)[4] = K.inv_oml, (&sl)[4] = K.sl, (&st)[4] = K.st;
  f32x16& acc = K.acc;
  const bool last = NT == 0 || store_gv;
  if (NT == 0 || FIRST) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lam[k] = evf_sigmoid(leak[4 * cg + k]);
      th[k] = fmaxf(thresh[4 * cg + k], 0.01f);
      oml[k] = 1.0f - lam[k];
      inv_oml[k] = 1.0f / oml[k];
      sl[k] = st[k] = 0.f;
      if (PLIF) {
        KP.lpt[k] = evf_plif_sigmoid(pm.leak_pt[4 * cg + k]);
        KP.apt[k] = xl ? fmaxf(pm.add_pt[4 * cg + k], 0.f) : evf_plif_sigmoid(pm.add_pt[4 * cg + k]);
        KP.slp[k] = KP.sap[k] = 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  }
  const float4* pgk = (PLIF && pq.g_pt_out) ? pq.g_pt_out : v_out;
  const float4* ppp = (PLIF && pq.pt_prev) ? pq.pt_prev : v_out;
  const float4* pgz = g_z_out ? g_z_out : v_out;
  const float4* pgv = g_v_out ? g_v_out : v_out;
  const float4* pvp = v_prev ? v_prev : v_out;
  const uint32_t* pzw = z_prev ? z_prev : (const uint32_t*)v_out;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // B operand column of this lane: (ci, dy, dx); columns >= 9 Cin are padding
  const int ncol = 9 * Cin;
  const bool colok = i < ncol;
  const int ci = colok ? i / 9 : 0, tap = colok ? i - 9 * ci : 0, dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
  const long HW = (long)H * W;
  const long total = npix * 8, stride = (long)gridDim.x * 256;
  // operands of the epilogue, requested now: this block's previous slab partial sums (ncol <= 32: <= 4 per thread) and its
  // row of per-channel sums -- after the loop they were two dependent HBM round trips at the end of the block's life
  float* sl_out = slab + (long)blockIdx.x * (C32 * ncol);
  float prev[4] = {0.f, 0.f, 0.f, 0.f}, row_prev = 0.f;
  const size_t row_off = (size_t)blockIdx.x * row_ld;
  if (last) {  // (block-uniform; a window launch: in its last pass only)
#pragma unroll
    for (int h = 0; h < 4; ++h) prev[h] = sl_out[min(tid + 256 * h, C32 * ncol - 1)];
    row_prev = (((tid >> 5) & 1) ? g_thresh : g_leak)[row_off + (tid & 31)];  // (used by threads < 64)
    if (PLIF && tid >= 64 && tid < 128) row_prev = (((tid >> 5) & 1) ? pm.g_add_pt : pm.g_leak_pt)[row_off + (tid & 31)];
  }
  // a trip in two halves: fetch() issues its loads (unconditional, clamped addresses), work() consumes them.  NT > 0: the
  // loads of ALL trips of the pass are issued before the first is consumed -- with one trip's loads in flight at a time (the
  // barrier inside a trip keeps the next trip's loads behind it) the kernel ran at the memory latency, not the bandwidth.
  typedef HeadTripIn TripIn;
  // (`first`: also the carried operands and the geometry.  Requesting the NEXT pass's operands early was measured twice and
  //  dropped: all of them after the last trip, ahead of the pass's reductions: 166.6 against 161.3 us per window; trip by trip
  //  as the registers come free, with the sums kept across the passes: 148.9 against 131.1.)
  auto fetch = [&](const long base, const int ic, TripIn& in, const bool first, const float4* pgz, const float4* pvp,
                   const uint32_t* pzw, const float* x_in, const float4* v_out, const float4* pgv) {  // ic: the trip's carry register
    const long e = base + tid;
    const bool ok = e < total;
    const long ec = ok ? e : total - 1;
    const long pix = ec >> 3;
    if (NT == 0 || first) {
      in.vo4 = v_out[ec];
      in.gvl = pgv[ec];
      if (PLIF) in.gkl = pgk[ec];
    }
    in.gzl = pgz[ec], in.vpl = pvp[ec];
    in.zwl = pzw[pix];
    if (PLIF) {
      in.ppl = ppp[ec];
      if constexpr (al) in.gxl = (has_gx ? (const float4*)gzxb : v_out)[ec];
      else in.Pl = pq.P[pix];
    }
    // B operand: the input value of the wave's pixels 2m + kg at this lane's (ci, tap)
    const long wp0 = (base >> 3) + wv * 8;  // first pixel of this wave
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (NT == 0) {
        const long q = wp0 + 2 * m + kg;
        const long qc = q < npix ? q : npix - 1;
        const int b = (int)(qc / HW), rem = (int)(qc - (long)b * HW), y = rem / W, x = rem - y * W;
        const int y2 = y + dy, x2 = x + dx;
        const bool inb = colok && q < npix && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
        const float xv = x_in[((long)(b * Cin + ci) * H + min(max(y2, 0), H - 1)) * W + min(max(x2, 0), W - 1)];
        in.xb[m] = inb ? xv : 0.f;
      } else {
        // the pixel geometry is the same in every pass of the launch: the element's offset into the input (-1: outside the
        // image / padding column) is worked out by the first pass only -- two 64-bit and two 32-bit divisions by run-time
        // values per element were most of this kernel's instructions
        int off;
        if (first) {
          const long q = wp0 + 2 * m + kg;
          const long qc = q < npix ? q : npix - 1;
          const int b = (int)(qc / HW), rem = (int)(qc - (long)b * HW), y = rem / W, x = rem - y * W;
          const int y2 = y + dy, x2 = x + dx;
          const bool inb = colok && q < npix && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
          off = inb ? (int)(((long)(b * Cin + ci) * H + y2) * W + x2) : -1;
          xo[ic][m] = off;
        } else {
          off = xo[ic][m];
        }
        const float xv = x_in[max(off, 0)];
        in.xb[m] = off >= 0 ? xv : 0.f;
      }
    }
  };
  auto work = [&](const long base, const int it, const int ic, const TripIn& in) {  // it: trip number (LDS buffer parity)
    const long e = base + tid;
    const bool ok = e < total;
    float4 vo4, gvl;
    if (NT == 0 || FIRST) {
      vo4 = in.vo4, gvl = in.gvl;
    } else {
      vo4 = voc[ic], gvl = gvc[ic];
    }
    const float4 gzl = in.gzl, vpl = in.vpl;
    const uint32_t zwl = in.zwl;
    float4 gz4 = g_z_out ? gzl : zero4;
    if constexpr (al) {
      const float4 gx4 = has_gx ? in.gxl : zero4;
      gz4 = make_float4(gz4.x + gx4.x, gz4.y + gx4.y, gz4.z + gx4.z, gz4.w + gx4.w);
    }
    const float4 gv4 = (g_v_out || (NT > 0 && !FIRST)) ? gvl : zero4, vp4 = v_prev ? vpl : zero4;
    const uint32_t zw = z_prev ? (zwl >> (4 * cg)) : 0u;
    const float vo[4] = {vo4.x, vo4.y, vo4.z, vo4.w}, gz[4] = {gz4.x, gz4.y, gz4.z, gz4.w};
    const float gvo[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, vp[4] = {vp4.x, vp4.y, vp4.z, vp4.w};
    float gc[4], gp[4], gsv[xl ? 4 : 1], pov[xl ? 4 : 1];
    if constexpr (xl) {  // pt' of the forward pass, recomputed: it is part of the threshold
      const float4 pp4 = pq.pt_prev ? in.ppl : zero4;
      const float pp[4] = {pp4.x, pp4.y, pp4.z, pp4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) pov[k] = evf_plif_trace(pp[k], KP.lpt[k], al ? (float)((zw >> k) & 1u) : in.Pl);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float z = (float)((zw >> k) & 1u);
      float the = th[k];
      if constexpr (xl) the = th[k] + KP.apt[k] * pov[k];
      const float sg = evf_surrogate(surrogate, vo[k] - the, width);
      const float gsp = gz[k] * sg;
      if constexpr (xl) gsv[k] = gsp;
      const float gv = gvo[k] + gsp;
      gc[k] = gv * oml[k];
      float cur, dlam, dth = 0.f;
      if (hard_reset) {
        gp[k] = gv * lam[k] * (1.0f - z);
        cur = (vo[k] - (vp[k] * lam[k]) * (1.0f - z)) * inv_oml[k];
        dlam = vp[k] * (1.0f - z) - cur;
      } else {
        gp[k] = gv * lam[k];
        cur = (vo[k] - vp[k] * lam[k] + z * th[k]) * inv_oml[k];
        dlam = vp[k] - cur;
        dth = gv * z;
      }
      if (ok) {
        sl[k] += gv * dlam;
        st[k] -= dth + gsp;
      }
    }
    if (NT > 0) {
      gvc[ic] = make_float4(gp[0], gp[1], gp[2], gp[3]);
      voc[ic] = vpl;  // (v_prev == NULL: never read again -- the pass that starts from the zero state is the window's first)
    }
    if (PLIF) {  // trace backward: the expressions of k_plif_trace_bwd, the same bits per element
      const float4 gk4 = (NT == 0 || FIRST) ? (pq.g_pt_out ? in.gkl : zero4) : gpc[ic];
      const float4 pp4 = pq.pt_prev ? in.ppl : zero4;
      const float gk[4] = {gk4.x, gk4.y, gk4.z, gk4.w}, pp[4] = {pp4.x, pp4.y, pp4.z, pp4.w};
      const float Pv = al ? 0.f : in.Pl;
      float gq[4], gzx[al ? 4 : 1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float po, gx;  // pt'; what the trace scaled in the forward pass: the threshold's / the current's gradient (negated)
        if constexpr (xl) po = pov[k], gx = gsv[k];
        else po = evf_plif_trace(pp[k], KP.lpt[k], Pv), gx = gc[k];
        const float g = gk[k] - KP.apt[k] * gx;
        gq[k] = g * KP.lpt[k];
        if constexpr (al) gzx[k] = g * (1.0f - KP.lpt[k]);
        if (ok) {
          KP.slp[k] += g * (pp[k] - (al ? (float)((zw >> k) & 1u) : Pv));
          KP.sap[k] -= gx * po;
        }
      }
      if constexpr (al) {
        if (ok) gzxb[e] = make_float4(gzx[0], gzx[1], gzx[2], gzx[3]);
      }
      if (NT > 0) gpc[ic] = make_float4(gq[0], gq[1], gq[2], gq[3]);
      if (ok && (NT == 0 || store_gv)) pq.g_pt_prev[e] = make_float4(gq[0], gq[1], gq[2], gq[3]);
    }
    if (ok) {
      if (g_cur) g_cur[e] = make_float4(gc[0], gc[1], gc[2], gc[3]);
      if (NT == 0 || store_gv) g_v_prev[e] = make_float4(gp[0], gp[1], gp[2], gp[3]);
    }
    float* sg_w = s_g[it & 1][wv];
    *(float4*)(sg_w + (lane >> 3) * C32 + 4 * cg) = ok ? make_float4(gc[0], gc[1], gc[2], gc[3]) : zero4;
    __syncthreads();  // (double-buffered: one barrier per trip)
#pragma unroll
    for (int m = 0; m < 4; ++m)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sg_w[(2 * m + kg) * C32 + i], in.xb[m], acc, 0, 0, 0);
  };
  if (NT == 0) {
    int it = 0;
    for (long base = (long)blockIdx.x * 256; base < total; base += stride, ++it) {  // block-uniform trip count
      TripIn in;
      fetch(base, 0, in, true, pgz, pvp, pzw, x_in, v_out, pgv);
      work(base, it, 0, in);
    }
  } else {
    TripIn tin[NT ? NT : 1];
#pragma unroll
    for (int it = 0; it < (NT ? NT : 1); ++it)
      fetch((long)blockIdx.x * 256 + it * stride, it, tin[it], FIRST, pgz, pvp, pzw, x_in, v_out, pgv);
#pragma unroll
    for (int it = 0; it < (NT ? NT : 1); ++it) {
      const long base = (long)blockIdx.x * 256 + it * stride;
      if (base < total) work(base, it, it, tin[it]);
    }
  }
  if (!last) return;  // (the sums go on into the next pass)
  // D[co][col] of the 4 waves -> slab[block][co][col] (torch layout [32][Cin][3][3])
#pragma unroll
  for (int r = 0; r < 16; ++r) s_d[wv][((r & 3) + 8 * (r >> 2) + 4 * kg) * C32 + i] = acc[r];
  __syncthreads();
  {  // (previous partial sums: prologue)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int e2 = tid + 256 * h;
      if (e2 < C32 * ncol) {
        const int co = e2 / ncol, col = e2 - co * ncol;
        const int q = co * C32 + col;
        const float v = (s_d[0][q] + s_d[1][q]) + (s_d[2][q] + s_d[3][q]);
        sl_out[e2] = (slab_acc ? prev[h] : 0.f) + v;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      sl[k] += __shfl_xor(sl[k], o, 64);
      st[k] += __shfl_xor(st[k], o, 64);
    }
  }
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_red[0][wv][4 * lane + k] = sl[k];
      s_red[1][wv][4 * lane + k] = st[k];
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, c = tid & 31;
    float v = 0.f;
    for (int w = 0; w < 4; ++w) v += s_red[which][w][c];
    const size_t ro = row_off;  // (row_ld > 0: per-block rows instead of same-address atomics; previous value: prologue)
    if (which == 0) {
      const float l = evf_sigmoid(leak[c]), t = v * l * (1.0f - l);
      if (row_ld) g_leak[ro + c] = row_prev + t;
      else evf_atomic_add(g_leak + c, t);
    } else if (thresh[c] > 0.01f) {
      if (row_ld) g_thresh[ro + c] = row_prev + v;
      else evf_atomic_add(g_thresh + c, v);
    }
  }
  if (PLIF) {  // the trace parameters' sums through the same arrays
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        KP.slp[k] += __shfl_xor(KP.slp[k], o, 64);
        KP.sap[k] += __shfl_xor(KP.sap[k], o, 64);
      }
    }
    if (lane < 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s_red[0][wv][4 * lane + k] = KP.slp[k];
        s_red[1][wv][4 * lane + k] = KP.sap[k];
      }
    }
    __syncthreads();
    if (tid >= 64 && tid < 128) {
      const int which = (tid >> 5) & 1, c = tid & 31;
      float v = 0.f;
      for (int w = 0; w < 4; ++w) v += s_red[which][w][c];
      const float sgm = evf_sigmoid(which == 0 ? pm.leak_pt[c] : pm.add_pt[c]);
      float* dst = which == 0 ? pm.g_leak_pt : pm.g_add_pt;
      // chain factor of the raw parameter: sigmoid' -- or, XLIF's t1.clamp_min(0), one where the clamp is inactive
      const float dv = (xl && which == 1) ? (pm.add_pt[c] > 0.f ? v : 0.f) : v * sgm * (1.0f - sgm);
      if (row_ld) dst[row_off + c] = row_prev + dv;
      else evf_atomic_add(dst + c, dv);
    }
  }
}

template <bool FAST>
__global__ __launch_bounds__(256) void k_head_bwd_mfma(
    const float4* __restrict__ g_z_out, const float4* __restrict__ g_v_out, const float4* __restrict__ v_out,
    const float4* __restrict__ v_prev, const uint32_t* __restrict__ z_prev, const float* __restrict__ leak,
    const float* __restrict__ thresh, long npix, int hard_reset_rt, int surrogate_rt, float width, float4* __restrict__ g_cur,
    float4* __restrict__ g_v_prev, float* __restrict__ g_leak, float* __restrict__ g_thresh,
    const float* __restrict__ x_in, int Cin, int H, int W, float* __restrict__ slab, int slab_acc, int row_ld) {
  float4 none[1];
  int nox[1][4];
  HeadBwdKeep keep;
  head_bwd_pass<FAST>(g_z_out, g_v_out, v_out, v_prev, z_prev, leak, thresh, npix, hard_reset_rt, surrogate_rt, width, g_cur,
                      g_v_prev, g_leak, g_thresh, x_in, Cin, H, W, slab, slab_acc, row_ld, none, none, nox, keep, true);
}

// PLIF head, one pass: the same with the trace backward inside (default neuron)
template <int XL>
__global__ __launch_bounds__(256) void k_head_plif_bwd_mfma(
    const float4* __restrict__ g_z_out, const float4* __restrict__ g_v_out, const float4* __restrict__ v_out,
    const float4* __restrict__ v_prev, const uint32_t* __restrict__ z_prev, const float* __restrict__ leak,
    const float* __restrict__ thresh, long npix, float width, float4* __restrict__ g_v_prev, float* __restrict__ g_leak,
    float* __restrict__ g_thresh, const float* __restrict__ x_in, int Cin, int H, int W, float* __restrict__ slab, int slab_acc,
    int row_ld, HeadPlifPass pq, HeadPlifPrm pm) {
  float4 none[1], gpc[1];
  int nox[1][4];
  HeadBwdKeep keep;
  HeadBwdKeepPlif kp;
  head_bwd_pass<true, 0, true, true, XL>(g_z_out, g_v_out, v_out, v_prev, z_prev, leak, thresh, npix, 1, EVF_ARCTAN, width, nullptr, g_v_prev,
                                     g_leak, g_thresh, x_in, Cin, H, W, slab, slab_acc, row_ld, none, none, nox, keep, true, pq, pm, gpc, &kp);
}

// The head layer's backward of a whole WINDOW in one launch.  The head's backward chain is per pixel (dL/dv carried from pass
// t to pass t-1 by the thread that computed it; the weight-gradient and per-channel partial sums are per block), so once the
// input gradients of the layer above have written dL/d(spikes) of EVERY pass -- each into its own buffer -- one launch runs
// the passes back to back in every block: no device-wide launch boundary between them, the carried gradient and the block's
// partial sums are re-read by the thread that just wrote them (cache hits).  Each pass is the body of k_head_bwd_mfma with the
// same grid: bit-identical to one launch per pass.
#define HEADBWD_MAX_P 16
#ifndef HEADBWD_LB
#define HEADBWD_LB 256
#endif
struct HeadBwdWin {
  HeadBwdPass p[HEADBWD_MAX_P];
  HeadPlifPrm pm;  // (PLIF windows)
  const float *leak, *thresh;
  float *g_leak, *g_thresh, *slab;
  long npix;
  int np, hard_reset, surrogate, Cin, H, W, row_ld;
  float width;
};
#define HEADBWD_NT 4  // trips of a block whose carried values fit registers (8 x 128 x 128 on 1024 blocks: 4)
template <bool FAST, int NT, bool PLIF = false, int XL = 0>
__global__ __launch_bounds__(HEADBWD_LB) void k_head_bwd_win(HeadBwdWin a) {
  float4 gvc[NT ? NT : 1], voc[NT ? NT : 1], gpc[NT ? NT : 1];
  int xo[NT ? NT : 1][4];
  HeadBwdKeep keep;
  HeadBwdKeepPlif kp;
  const int acc0 = a.p[0].slab_acc;  // (NT > 0: the sums of all passes are added to the slab once, by the first pass's rule)
  {
    const HeadBwdPass& q = a.p[0];
    head_bwd_pass<FAST, NT, true, PLIF, XL>(q.g_z_out, q.g_v_out, q.v_out, q.v_prev, q.z_prev, a.leak, a.thresh, a.npix, a.hard_reset,
                                        a.surrogate, a.width, nullptr, q.g_v_prev, a.g_leak, a.g_thresh, q.x_in, a.Cin, a.H, a.W, a.slab,
                                        NT > 0 ? acc0 : q.slab_acc, a.row_ld, gvc, voc, xo, keep, a.np == 1,
                                        HeadPlifPass{q.g_pt_out, q.pt_prev, q.P, q.g_pt_prev}, a.pm, gpc, &kp);
  }
  for (int t = 1; t < a.np; ++t) {
    __syncthreads();  // (the pass's last reads of the reduction arrays before the next pass rewrites them)
    const HeadBwdPass& q = a.p[t];
    head_bwd_pass<FAST, NT, false, PLIF, XL>(q.g_z_out, q.g_v_out, q.v_out, q.v_prev, q.z_prev, a.leak, a.thresh, a.npix, a.hard_reset,
                                         a.surrogate, a.width, nullptr, q.g_v_prev, a.g_leak, a.g_thresh, q.x_in, a.Cin, a.H, a.W, a.slab,
                                         NT > 0 ? acc0 : q.slab_acc, a.row_ld, gvc, voc, xo, keep, t == a.np - 1,
                                         HeadPlifPass{q.g_pt_out, q.pt_prev, q.P, q.g_pt_prev}, a.pm, gpc, &kp);
  }
}

#define HEAD_BWD_BLOCKS 1024  // 4 blocks per CU (512: 26.9 us, 768-1024: 25.5 us, 2048: 29.4 us at 8 x 128 x 128)
static int head_bwd_blocks() {  // EVF_HEAD_BWD_BLOCKS: A/B of the grid (blocks resident per CU = bytes in flight)
  static int n = 0;
  if (!n) {
    const char* e = getenv("EVF_HEAD_BWD_BLOCKS");
    n = e ? atoi(e) : HEAD_BWD_BLOCKS;
    if (n < 1 || n > 4096) n = HEAD_BWD_BLOCKS;
  }
  return n;
}
extern "C" int evf_head_lif_bwd_wgrad_slabs(int B, int H, int W) {
  const long npix = (long)B * H * W;
  const long nb = (npix * 8 + 255) / 256;
  const int cap = head_bwd_blocks();
  // beyond 4 trips per block at the default grid (sensor-size inputs: 260 x 346 x B4 = 11 trips of 1024 blocks) the window launch
  // could not keep its carried values in registers (HEADBWD_NT): more blocks instead, three trips each
  if (cap == HEAD_BWD_BLOCKS && nb > 4 * (long)HEAD_BWD_BLOCKS) {
    const long n3 = (nb + 2) / 3;
    return (int)(n3 < 4096 ? n3 : 4096);
  }
  return (int)(nb < cap ? nb : cap);
}

// Head layer: neuron backward + weight gradient in one pass over g (models/spiking_submodules.py:96-126
// autograd).  x_in [B,Cin,H,W] is the network input (Cin = 2); slab [evf_head_lif_bwd_wgrad_slabs][32*Cin*9]
// receives (accumulate = 1: is added) the per-block weight-gradient partials in torch layout.
struct HdArgs {
  const float *g_z_out, *g_v_out, *v_out, *v_prev;
  const uint32_t* z_prev;
  const float *x_in, *leak, *thresh;
  int B, Cin, H, W, hard_reset, surrogate;
  float act_width;
  float *g_cur, *g_v_prev, *g_leak, *g_thresh, *slab;
  int accumulate;
  // PLIF head (evf_head_plif_bwd_wgrad): P != NULL
  const float *g_pt_out, *pt_prev, *P, *leak_pt, *add_pt;
  float *g_pt_prev, *g_leak_pt, *g_add_pt;
};
static int head_bwd_go(const HdArgs& a, void* stream) {
  const long npix = (long)a.B * a.H * a.W;
  const int nblk = evf_head_lif_bwd_wgrad_slabs(a.B, a.H, a.W);
  const int row_ld = a.accumulate >> 8;  // pitch of the per-block parameter-gradient rows (0: dense outputs, atomics)
  const int accumulate = a.accumulate & 1;
  if (a.P) {
#define HEAD_PLIF_BWD(XL_)                                                                                                       \
  hipLaunchKernelGGL(k_head_plif_bwd_mfma<XL_>, dim3(nblk), dim3(256), 0, EVF_STREAM(stream), (const float4*)a.g_z_out,           \
                     (const float4*)a.g_v_out, (const float4*)a.v_out, (const float4*)a.v_prev, a.z_prev, a.leak, a.thresh, npix, \
                     a.act_width, (float4*)a.g_v_prev, a.g_leak, a.g_thresh, a.x_in, a.Cin, a.H, a.W, a.slab, accumulate, row_ld,  \
                     HeadPlifPass{(const float4*)a.g_pt_out, (const float4*)a.pt_prev, a.P, (float4*)a.g_pt_prev},               \
                     HeadPlifPrm{a.leak_pt, a.add_pt, a.g_leak_pt, a.g_add_pt})
    if (((a.hard_reset >> 1) & 3) == 2) HEAD_PLIF_BWD(2);  // (bits 1-2: 1 an XLIF head, 2 an ALIF head)
    else if (a.hard_reset & 2) HEAD_PLIF_BWD(1);
    else HEAD_PLIF_BWD(0);
#undef HEAD_PLIF_BWD
    return evf_status();
  }
#define HEAD_BWD(FAST_)                                                                                                    \
  hipLaunchKernelGGL(k_head_bwd_mfma<FAST_>, dim3(nblk), dim3(256), 0, EVF_STREAM(stream), (const float4*)a.g_z_out,       \
                     (const float4*)a.g_v_out, (const float4*)a.v_out, (const float4*)a.v_prev, a.z_prev, a.leak, a.thresh,  \
                     npix, a.hard_reset, a.surrogate, a.act_width, (float4*)a.g_cur, (float4*)a.g_v_prev, a.g_leak,         \
                     a.g_thresh, a.x_in, a.Cin, a.H, a.W, a.slab, accumulate, row_ld)
  if (a.hard_reset != 0 && a.surrogate == EVF_ARCTAN)
    HEAD_BWD(true);
  else
    HEAD_BWD(false);
#undef HEAD_BWD
  return evf_status();
}

// head cells recorded by evf_bwd_defer_* (evf_common.h): launched one by one when their index comes up
#define HD_MAX_JOBS 4
struct HdDefer {
  int n[EVF_BWD_DIAGS];
  HdArgs job[EVF_BWD_DIAGS][HD_MAX_JOBS];
};
static HdDefer hd_tab[EVF_CTX_MAX];
int evf_hd_defer_count(int ctx) {
  const HdDefer& hd_defer = hd_tab[ctx];
  int n = 0;
  for (int d = 0; d < EVF_BWD_DIAGS; ++d) n += hd_defer.n[d];
  return n;
}
int evf_hd_defer_pending(int ctx, int d) { return hd_tab[ctx].n[d]; }
// All recorded head cells at the END of a flush (after the last index), consecutive passes in one launch.  Allowed when no two
// of them read the same dL/d(spikes) buffer (with ONE buffer the next pass's input gradient overwrites it: the cells then have
// to run where they were recorded) -- nothing else they read is written by a recorded cell, nothing they write is read by one.
int evf_hd_defer_window_ok(int ctx) {
  static const bool window = []() {  // EVF_HEAD_WIN=0: head cells where they were recorded, one launch each
    const char* e = getenv("EVF_HEAD_WIN");
    return !(e && e[0] == '0');
  }();
  if (!window) return 0;
  const HdDefer& hd = hd_tab[ctx];
  const float* seen[EVF_BWD_DIAGS * HD_MAX_JOBS];
  int ns = 0, n = 0;
  for (int d = 0; d < EVF_BWD_DIAGS; ++d)
    for (int k = 0; k < hd.n[d]; ++k, ++n) {
      const float* g = hd.job[d][k].g_z_out;
      if (!g) continue;
      for (int i = 0; i < ns; ++i)
        if (seen[i] == g) return 0;
      seen[ns++] = g;
    }
  return n >= 2 ? 1 : 0;
}
int evf_hd_defer_launch_window(int ctx, void* stream) {
  HdDefer& hd = hd_tab[ctx];
  const HdArgs* jobs[EVF_BWD_DIAGS * HD_MAX_JOBS];
  int n = 0;
  for (int d = 0; d < EVF_BWD_DIAGS; ++d) {
    for (int k = 0; k < hd.n[d]; ++k) jobs[n++] = &hd.job[d][k];
  }
  int rc = EVF_OK;
  int k = 0;
  while (k < n && !rc) {
    const HdArgs& f = *jobs[k];
    int m = 1;
    while (k + m < n && m < HEADBWD_MAX_P && !f.g_cur) {  // pass k+m continues pass k+m-1: carried gradient, same accumulators
      const HdArgs &p = *jobs[k + m - 1], &q = *jobs[k + m];
      if (!(q.g_v_out == p.g_v_prev && q.v_out == p.v_prev && p.v_prev && !q.g_cur && q.leak == p.leak && q.thresh == p.thresh && q.B == p.B && q.Cin == p.Cin &&
            q.H == p.H && q.W == p.W && q.hard_reset == p.hard_reset && q.surrogate == p.surrogate &&
            q.act_width == p.act_width && q.g_leak == p.g_leak && q.g_thresh == p.g_thresh && q.slab == p.slab &&
            (q.accumulate >> 8) == (p.accumulate >> 8)))
        break;
      if ((q.P != nullptr) != (p.P != nullptr)) break;
      if (q.P && !(q.g_pt_out == p.g_pt_prev && q.leak_pt == p.leak_pt && q.add_pt == p.add_pt && q.g_leak_pt == p.g_leak_pt &&
                   q.g_add_pt == p.g_add_pt))
        break;
      ++m;
    }
    evf_prof_mark(m == 1 ? 3 : 5, 0, stream);
    if (m == 1) {
      rc = head_bwd_go(f, stream);
    } else {
      HeadBwdWin a;
      for (int t = 0; t < HEADBWD_MAX_P; ++t) {
        const HdArgs& q = *jobs[k + (t < m ? t : 0)];
        a.p[t] = HeadBwdPass{(const float4*)q.g_z_out, (const float4*)q.g_v_out, (const float4*)q.v_out, (const float4*)q.v_prev,
                             q.z_prev, q.x_in, (float4*)q.g_v_prev, q.accumulate & 1,
                             (const float4*)q.g_pt_out, (const float4*)q.pt_prev, q.P, (float4*)q.g_pt_prev};
      }
      a.pm = HeadPlifPrm{f.leak_pt, f.add_pt, f.g_leak_pt, f.g_add_pt};
      a.leak = f.leak, a.thresh = f.thresh, a.g_leak = f.g_leak, a.g_thresh = f.g_thresh, a.slab = f.slab;
      a.npix = (long)f.B * f.H * f.W;
      a.np = m, a.hard_reset = f.hard_reset, a.surrogate = f.surrogate, a.Cin = f.Cin, a.H = f.H, a.W = f.W;
      a.row_ld = f.accumulate >> 8, a.width = f.act_width;
      const int nblk = evf_head_lif_bwd_wgrad_slabs(f.B, f.H, f.W);
      const bool fast = f.hard_reset != 0 && f.surrogate == EVF_ARCTAN;
      const long trips = (a.npix * 8 + (long)nblk * 256 - 1) / ((long)nblk * 256);
      static const bool carry = []() {  // EVF_HEAD_WIN=mem: carried values through memory (A/B measurements)
        const char* e = getenv("EVF_HEAD_WIN");
        return !(e && e[0] == 'm');
      }();
#define HEAD_BWD_WIN(FAST_, NT_) hipLaunchKernelGGL((k_head_bwd_win<FAST_, NT_>), dim3(nblk), dim3(256), 0, EVF_STREAM(stream), a)
      if (f.P) {  // PLIF (default neuron only: evf_head_plif_bwd_wgrad): three trips' carried values fit the registers
        const bool win3 = carry && trips <= 3 && (long)f.B * f.Cin * f.H * f.W < (1L << 31);
        if (((f.hard_reset >> 1) & 3) == 2) {  // (bits 1-2: 2 an ALIF head, 1 an XLIF head)
          if (win3) hipLaunchKernelGGL((k_head_bwd_win<true, 3, true, 2>), dim3(nblk), dim3(256), 0, EVF_STREAM(stream), a);
          else hipLaunchKernelGGL((k_head_bwd_win<true, 0, true, 2>), dim3(nblk), dim3(256), 0, EVF_STREAM(stream), a);
        } else if (f.hard_reset & 2) {
          if (win3) hipLaunchKernelGGL((k_head_bwd_win<true, 3, true, 1>), dim3(nblk), dim3(256), 0, EVF_STREAM(stream), a);
          else hipLaunchKernelGGL((k_head_bwd_win<true, 0, true, 1>), dim3(nblk), dim3(256), 0, EVF_STREAM(stream), a);
        } else {
          if (win3) hipLaunchKernelGGL((k_head_bwd_win<true, 3, true>), dim3(nblk), dim3(256), 0, EVF_STREAM(stream), a);
          else hipLaunchKernelGGL((k_head_bwd_win<true, 0, true>), dim3(nblk), dim3(256), 0, EVF_STREAM(stream), a);
        }
      } else if (carry && fast && trips <= HEADBWD_NT && (long)f.B * f.Cin * f.H * f.W < (1L << 31)) {  // (other surrogates: 253 VGPRs)
        HEAD_BWD_WIN(true, HEADBWD_NT);
      } else {
        if (fast) HEAD_BWD_WIN(true, 0); else HEAD_BWD_WIN(false, 0);
      }
#undef HEAD_BWD_WIN
      rc = evf_status();
    }
    evf_prof_mark(m == 1 ? 3 : 5, 1, stream);
    k += m;
  }
  for (int d = 0; d < EVF_BWD_DIAGS; ++d) hd.n[d] = 0;
  return rc;
}
int evf_hd_defer_launch(int ctx, int d, void* stream) {
  HdDefer& hd_defer = hd_tab[ctx];
  for (int k = 0; k < hd_defer.n[d]; ++k) {
    evf_prof_mark(3, 0, stream);
    const int rc = head_bwd_go(hd_defer.job[d][k], stream);
    evf_prof_mark(3, 1, stream);
    if (rc) return rc;
  }
  hd_defer.n[d] = 0;
  return EVF_OK;
}

static int head_bwd_record_or_go(const HdArgs& a, void* stream) {
  const int bctx = evf_ctx_find(stream);
  const EvfBwdDefer evf_bwd_defer = bctx >= 0 ? evf_bwd_defer_tab[bctx] : EvfBwdDefer{false, 0, false};
  HdDefer& hd_defer = hd_tab[bctx < 0 ? 0 : bctx];
  if (evf_bwd_defer.active) {
    if (hd_defer.n[evf_bwd_defer.slot] < HD_MAX_JOBS) {
      hd_defer.job[evf_bwd_defer.slot][hd_defer.n[evf_bwd_defer.slot]++] = a;
      return EVF_OK;
    }
    const int rc = evf_bwd_defer_flush_now(bctx, stream);
    if (rc) return rc;
  }
  return head_bwd_go(a, stream);
}

extern "C" int evf_head_lif_bwd_wgrad(const float* g_z_out, const float* g_v_out, const float* v_out, const float* v_prev,
                                      const uint32_t* z_prev, const float* x_in, const float* leak, const float* thresh,
                                      int B, int Cin, int H, int W, int hard_reset, int surrogate, float act_width,
                                      float* g_cur, float* g_v_prev, float* g_leak, float* g_thresh, float* slab,
                                      int accumulate, void* stream) {
  if (!v_out || !x_in || !leak || !thresh || !g_v_prev || !g_leak || !g_thresh || !slab || B <= 0 || H <= 0 || W <= 0 ||
      Cin != 2)
    return EVF_EINVAL;
  const HdArgs a{g_z_out, g_v_out, v_out, v_prev, z_prev, x_in, leak, thresh, B, Cin, H, W, hard_reset, surrogate, act_width,
                 g_cur, g_v_prev, g_leak, g_thresh, slab, accumulate, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  return head_bwd_record_or_go(a, stream);
}

// PLIF head: evf_head_lif_bwd_wgrad with the presynaptic trace's backward inside (what evf_plif_trace_bwd computes from g_cur in a
// launch of its own; g_cur is not written).  Recordable (evf_bwd_defer_*): the recorded cells of a window run in one launch.
extern "C" int evf_head_plif_bwd_wgrad(const float* g_z_out, const float* g_v_out, const float* v_out, const float* v_prev,
                                       const uint32_t* z_prev, const float* x_in, const float* leak, const float* thresh, int B,
                                       int Cin, int H, int W, int hard_reset, int surrogate, float act_width, float* g_v_prev,
                                       float* g_leak, float* g_thresh, float* slab, int accumulate, const float* g_pt_carry,
                                       const float* pt_prev, const float* P, const float* leak_pt, const float* add_pt,
                                       float* g_pt_prev, float* g_leak_pt, float* g_add_pt, void* stream) {
  if (!v_out || !x_in || !leak || !thresh || !g_v_prev || !g_leak || !g_thresh || !slab || B <= 0 || H <= 0 || W <= 0 ||
      Cin != 2 || !P || !leak_pt || !add_pt || !g_pt_prev || !g_leak_pt || !g_add_pt)
    return EVF_EINVAL;
  if (!((hard_reset & 1) && surrogate == EVF_ARCTAN)) return EVF_ENOTSUP;  // (the two-call path serves the other neurons; bit 1: an XLIF head)
  const HdArgs a{g_z_out, g_v_out, v_out, v_prev, z_prev, x_in, leak, thresh, B, Cin, H, W, hard_reset, surrogate, act_width,
                 nullptr, g_v_prev, g_leak, g_thresh, slab, accumulate, g_pt_carry, pt_prev, P, leak_pt, add_pt, g_pt_prev,
                 g_leak_pt, g_add_pt};
  return head_bwd_record_or_go(a, stream);
}

// dst_k[i] += src[off_k + i], i < n_k, for up to 32 segments in one launch (block y = segment)
struct AddSegs {
  float* dst[32];
  int off[32];
  int n[32];
};
__global__ void k_add_segments(float* __restrict__ src, AddSegs sg, int clear) {
  const int k = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < sg.n[k]; i += gridDim.x * blockDim.x) {
    sg.dst[k][i] += src[sg.off[k] + i];
    if (clear) src[sg.off[k] + i] = 0.f;  // a persistent accumulator is handed back zeroed
  }
}
extern "C" int evf_add_segments(float* src, void* const* dst, const int* off, const int* n, int nseg, int clear, void* stream) {
  if (!src || !dst || !off || !n || nseg <= 0 || nseg > 32) return EVF_EINVAL;
  AddSegs sg;
  int nmax = 1;
  for (int k = 0; k < 32; ++k) {
    sg.dst[k] = k < nseg ? (float*)dst[k] : nullptr;
    sg.off[k] = k < nseg ? off[k] : 0;
    sg.n[k] = k < nseg ? n[k] : 0;
    if (sg.n[k] > nmax) nmax = sg.n[k];
  }
  hipLaunchKernelGGL(k_add_segments, dim3(evf_cdiv(nmax, 256) < 8 ? evf_cdiv(nmax, 256) : 8, nseg), dim3(256), 0,
                     EVF_STREAM(stream), src, sg, clear);
  return evf_status();
}

// dst[e] (+)= sum_k rows[k][e]: 16 columns x 16 row groups per block, LDS tree over the groups
__global__ __launch_bounds__(256) void k_sum_rows(float* __restrict__ rows, int nrows, int n, int accumulate,
                                                  float* __restrict__ dst, int clear) {
  __shared__ float s[16][17];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + c;
  float v = 0.f;
  if (e < n)
    for (int k = grp; k < nrows; k += 16) {
      v += rows[(long)k * n + e];
      if (clear) rows[(long)k * n + e] = 0.f;  // (a persistent buffer of per-block partials is handed back zeroed)
    }
  s[grp][c] = v;
  __syncthreads();
  if (grp == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += s[k][c];
    dst[e] = accumulate ? dst[e] + t : t;
  }
}
// accumulating form for many rows: block (x, y) sums rows [64 y, 64 y + 64) of 64 columns (256-byte row segments) and
// adds its partial to dst (nrows / 64 adds per word): 8x the blocks of k_sum_rows and coalesced reads
__global__ __launch_bounds__(256) void k_sum_rows_acc(float* __restrict__ rows, int nrows, int n, float* __restrict__ dst,
                                                      int clear) {
  __shared__ float s[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + c, r0 = blockIdx.y * 64;
  float v = 0.f;
  if (e < n)
    for (int k = r0 + grp; k < min(r0 + 64, nrows); k += 4) {
      v += rows[(long)k * n + e];
      if (clear) rows[(long)k * n + e] = 0.f;
    }
  s[grp][c] = v;
  __syncthreads();
  if (grp == 0 && e < n) {
    const float t = (s[0][c] + s[1][c]) + (s[2][c] + s[3][c]);
    if (t != 0.f) evf_atomic_add(dst + e, t);
  }
}

extern "C" int evf_sum_rows(float* rows, int nrows, int n, int accumulate, float* dst, void* stream) {
  if (!rows || !dst || nrows <= 0 || n <= 0) return EVF_EINVAL;
  if ((accumulate & 1) && nrows >= 128) {
    hipLaunchKernelGGL(k_sum_rows_acc, dim3(evf_cdiv(n, 64), evf_cdiv(nrows, 64)), dim3(256), 0, EVF_STREAM(stream), rows, nrows, n,
                       dst, (accumulate >> 1) & 1);
    return evf_status();
  }
  // accumulate bit 0: dst += (else =); bit 1: zero the rows after reading them
  hipLaunchKernelGGL(k_sum_rows, dim3(evf_cdiv(n, 16)), dim3(256), 0, EVF_STREAM(stream), rows, nrows, n, accumulate & 1, dst,
                     (accumulate >> 1) & 1);
  return evf_status();
}

// --------------------------------------------------------------------------
// input-gradient conv (fp32 activations through an LDS halo tile)
// --------------------------------------------------------------------------
#define PIX_STRIDE 36  // floats per halo pixel: 32 channels + 4 pad -> conflict-free ds_read_b128

template <bool TWO>
__global__ __launch_bounds__(256) void k_conv_dgrad(const float* __restrict__ g, const float* __restrict__ wa,
                                                    float* __restrict__ ga, int acc_a, const float* __restrict__ wb,
                                                    float* __restrict__ gb, int acc_b, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_wa = (float*)smem_raw;
  float* s_wb = s_wa + WPACK;
  float* s_g = s_wa + (TWO ? 2 : 1) * WPACK;  // HALO_H*HALO_W*PIX_STRIDE
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  for (int i = tid; i < WPACK / 4; i += 256) ((float4*)s_wa)[i] = ((const float4*)wa)[i];
  if (TWO)
    for (int i = tid; i < WPACK / 4; i += 256) ((float4*)s_wb)[i] = ((const float4*)wb)[i];
  for (int e = tid; e < HALO_H * HALO_W * 8; e += 256) {
    const int p = e >> 3, c4 = e & 7;
    const int yy = y0 + p / HALO_W - 1, xx = x0 + p % HALO_W - 1;
    float4 v = make_float4(0, 0, 0, 0);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = ((const float4*)g)[(((long)b * H + yy) * W + xx) * 8 + c4];
    *(float4*)(s_g + p * PIX_STRIDE + c4 * 4) = v;
  }
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, b0 = {0}, b1 = {0};
  const int i = lane & 31, h = lane >> 5, r0 = 2 * wv;
#pragma unroll 1
  for (int tau = 0; tau < 9; ++tau) {
    const int dy = tau / 3, dx = tau % 3;
    const float* p0 = s_g + ((r0 + dy) * HALO_W + i + dx) * PIX_STRIDE + 16 * h;
    const float* p1 = p0 + HALO_W * PIX_STRIDE;
    float x0v[16], x1v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 u = *(const float4*)(p0 + 4 * q), w = *(const float4*)(p1 + 4 * q);
      x0v[4 * q] = u.x, x0v[4 * q + 1] = u.y, x0v[4 * q + 2] = u.z, x0v[4 * q + 3] = u.w;
      x1v[4 * q] = w.x, x1v[4 * q + 1] = w.y, x1v[4 * q + 2] = w.z, x1v[4 * q + 3] = w.w;
    }
    const float* wpa = s_wa + tau * 1024 + lane;
    const float* wpb = s_wb + tau * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float bw = wpa[t * 64];
      a0 = mfma32(x0v[t], bw, a0);
      a1 = mfma32(x1v[t], bw, a1);
      if (TWO) {
        const float bw2 = wpb[t * 64];
        b0 = mfma32(x0v[t], bw2, b0);
        b1 = mfma32(x1v[t], bw2, b1);
      }
    }
  }
  const int j = lane & 31;
  auto store = [&](const f32x16& acc, int row, float* __restrict__ out, int accf) {
    if (row >= H) return;
    float prev[16];  // read together from clamped addresses, selected afterwards (no load under a branch)
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[r] = out[(((long)b * H + row) * W + min(x0 + mfma_row(r, lane), W - 1)) * C32 + j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = x0 + mfma_row(r, lane);
      if (col < W) out[(((long)b * H + row) * W + col) * C32 + j] = (accf ? prev[r] : 0.f) + acc[r];
    }
  };
  store(a0, y0 + r0, ga, acc_a);
  store(a1, y0 + r0 + 1, ga, acc_a);
  if (TWO) {
    store(b0, y0 + r0, gb, acc_b);
    store(b1, y0 + r0 + 1, gb, acc_b);
  }
}

extern "C" int evf_conv_dgrad(const float* g_cur, const float* wT_a, float* g_a, int acc_a, const float* wT_b,
                              float* g_b, int acc_b, int B, int H, int W, void* stream) {
  if (!g_cur || !wT_a || !g_a || B <= 0 || H <= 0 || W <= 0 || ((wT_b != nullptr) != (g_b != nullptr)))
    return EVF_EINVAL;
  dim3 grid(evf_cdiv(W, TW), evf_cdiv(H, TH), B), block(256);
  hipStream_t st = EVF_STREAM(stream);
  const size_t halo = (size_t)HALO_H * HALO_W * PIX_STRIDE * 4;
  static bool attr1 = false, attr2 = false;
  if (wT_b) {
    const size_t lds = 2 * WPACK * 4 + halo;
    if (!attr2) {
      (void)hipFuncSetAttribute((const void*)k_conv_dgrad<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr2 = true;
    }
    hipLaunchKernelGGL(k_conv_dgrad<true>, grid, block, lds, st, g_cur, wT_a, g_a, acc_a, wT_b, g_b, acc_b, B, H, W);
  } else {
    const size_t lds = WPACK * 4 + halo;
    if (!attr1) {
      (void)hipFuncSetAttribute((const void*)k_conv_dgrad<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr1 = true;
    }
    hipLaunchKernelGGL(k_conv_dgrad<false>, grid, block, lds, st, g_cur, wT_a, g_a, acc_a, (const float*)nullptr,
                       (float*)nullptr, 0, B, H, W);
  }
  return evf_status();
}

// (weight-gradient conv on bit-packed inputs: see evf_wgrad.hip)

// Head weight gradient: GEMM rows = (ci, tau) pairs (<= 72 -> up to 3 M tiles), cols = co,
// K = pixels.  Each wave walks whole image rows (grid-strided); the per-lane tap decode is
// hoisted out of the pixel loop; the 4 waves of a block are summed through LDS stores and
// the block adds its tile to dw (torch layout) with global atomics.
#define HW_BLOCKS 256
__global__ __launch_bounds__(256) void k_head_wgrad(const float* __restrict__ x, const float* __restrict__ g, int B,
                                                    int Cin, int H, int W, float* __restrict__ dw) {
  __shared__ float s_p[4][3 * C32 * C32];  // 48 KiB
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int nrow = 9 * Cin, NT = (nrow + 31) / 32;
  int ci_[3], dy_[3], dx_[3];
  bool on_[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int rr = 32 * t + i;
    on_[t] = t < NT && rr < nrow;
    ci_[t] = on_[t] ? rr / 9 : 0;
    const int tau = rr % 9;
    dy_[t] = tau / 3 - 1;
    dx_[t] = tau % 3 - 1;
  }
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = (f32x16){0};
  const long nrows = (long)B * H;
  for (long row = (long)blockIdx.x * 4 + wv; row < nrows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / H), y = (int)(row % H);
    const float* grow = g + row * W * C32 + i;
    const float* xr[3];
    bool yok[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int yy = y + dy_[t];
      yok[t] = on_[t] && yy >= 0 && yy < H;
      xr[t] = x + (((long)b * Cin + ci_[t]) * H + (yok[t] ? yy : 0)) * W;
    }
    for (int xs = 0; xs < W; xs += 8) {
      float bv[4], av[3][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int xc = xs + 2 * u + h;
        bv[u] = xc < W ? grow[(long)xc * C32] : 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int xx = xc + dx_[t];
          av[t][u] = (yok[t] && xx >= 0 && xx < W) ? xr[t][xx] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (t < NT) acc[t] = mfma32(av[t][u], bv[u], acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_p[wv][(t * C32 + mfma_row(r, lane)) * C32 + i] = acc[t][r];
  __syncthreads();
  for (int e = tid; e < NT * C32 * C32; e += 256) {
    const int co = e & 31, rr = e >> 5;
    if (rr < nrow) {
      const float v = (s_p[0][e] + s_p[1][e]) + (s_p[2][e] + s_p[3][e]);
      evf_atomic_add(dw + (co * Cin + rr / 9) * 9 + rr % 9, v);
    }
  }
}

extern "C" int evf_head_wgrad(const float* x, const float* g_cur, int B, int Cin, int H, int W, float* dw,
                              void* stream) {
  if (!x || !g_cur || !dw || B <= 0 || Cin <= 0 || Cin > HEAD_MAX_CIN || H <= 0 || W <= 0) return EVF_EINVAL;
  const long nb = ((long)B * H + 3) / 4;
  hipLaunchKernelGGL(k_head_wgrad, dim3((int)(nb < HW_BLOCKS ? nb : HW_BLOCKS)), dim3(256), 0, EVF_STREAM(stream), x,
                     g_cur, B, Cin, H, W, dw);
  return evf_status();
}

// --------------------------------------------------------------------------
// prediction layer: 1x1 conv 32 -> 2, bias, tanh
// --------------------------------------------------------------------------
__global__ void k_pred_fwd(const uint32_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                           int B, int HW, float* __restrict__ flow) {
  __shared__ float s_w[2 * C32 + 2];
  if (threadIdx.x < 2 * C32) s_w[threadIdx.x] = w[threadIdx.x];
  if (threadIdx.x < 2) s_w[2 * C32 + threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)B * HW) return;
  const uint32_t m = x[p];
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < C32; ++c) {
    const float z = (float)((m >> c) & 1u);
    s0 += z * s_w[c];
    s1 += z * s_w[C32 + c];
  }
  const long b = p / HW, q = p % HW;
  flow[(b * 2) * HW + q] = tanhf(s0 + s_w[2 * C32]);
  flow[(b * 2 + 1) * HW + q] = tanhf(s1 + s_w[2 * C32 + 1]);
}

extern "C" int evf_pred_fwd(const uint32_t* x, const float* w, const float* bias, int B, int H, int W, float* flow,
                            void* stream) {
  if (!x || !w || !bias || !flow || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_pred_fwd, dim3(evf_cdiv((long)B * H * W, 256)), dim3(256), 0, EVF_STREAM(stream), x, w, bias, B,
                     H * W, flow);
  return evf_status();
}

// g_x[pix][c] = sum_o gpre[o] * w[o][c];  dw[o][c] += sum_pix gpre[o]*z[pix][c];  dbias[o] += sum gpre[o]
// Grid-strided with few blocks: each block ends with 66 global atomics on the same 66 words.
#define PB_BLOCKS 256
__global__ __launch_bounds__(256) void k_pred_bwd(const uint32_t* __restrict__ x, const float* __restrict__ flow,
                                                  const float* __restrict__ g_flow, const float* __restrict__ w,
                                                  int B, int HW, float* __restrict__ g_x, float* __restrict__ dw,
                                                  float* __restrict__ dbias) {
  __shared__ float s_w[2 * C32];
  __shared__ float s_p[4][2 * C32 + 2];
  if (threadIdx.x < 2 * C32) s_w[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float accw0 = 0.f, accb0 = 0.f, accb1 = 0.f;  // lane owns dw[lane >> 5][lane & 31]
  const long npix = (long)B * HW;
  for (long p0 = (long)blockIdx.x * 256 + wv * 64; p0 < npix; p0 += (long)gridDim.x * 256) {
    const long p = p0 + lane;
    float gp0 = 0.f, gp1 = 0.f;
    uint32_t m = 0u;
    if (p < npix) {
      const long b = p / HW, q = p % HW;
      const float f0 = flow[(b * 2) * HW + q], f1 = flow[(b * 2 + 1) * HW + q];
      gp0 = g_flow[(b * 2) * HW + q] * (1.0f - f0 * f0);  // tanh'
      gp1 = g_flow[(b * 2 + 1) * HW + q] * (1.0f - f1 * f1);
      m = x[p];
    }
    // g_x rows of the wave's 64 pixels, written coalesced: per store instruction lane l covers float4 (l & 7) of pixel
    // 8 i + (l >> 3), i.e. 8 whole 128-B rows; the pixel's (gp0, gp1) come from its owner lane by shuffle
    {
      const int c4 = lane & 7;
      const float wa0 = s_w[4 * c4], wa1 = s_w[4 * c4 + 1], wa2 = s_w[4 * c4 + 2], wa3 = s_w[4 * c4 + 3];
      const float wb0 = s_w[C32 + 4 * c4], wb1 = s_w[C32 + 4 * c4 + 1], wb2 = s_w[C32 + 4 * c4 + 2],
                  wb3 = s_w[C32 + 4 * c4 + 3];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int src = 8 * i + (lane >> 3);
        const float a = __shfl(gp0, src, 64), bq = __shfl(gp1, src, 64);
        const long pp = p0 + src;
        if (pp < npix)
          *(float4*)(g_x + pp * C32 + 4 * c4) = make_float4(a * wa0 + bq * wb0, a * wa1 + bq * wb1, a * wa2 + bq * wb2,
                                                            a * wa3 + bq * wb3);
      }
    }
    // dw[o][c] += sum over the wave's 64 pixels of bit_c(pixel) * gp_o(pixel): lane (c = lane & 31, o = lane >> 5)
    // walks the pixels with scalar broadcasts (v_readlane), no cross-lane reductions
    {
      const int c = lane & 31;
      const bool second = lane >= 32;
#pragma unroll
      for (int jp = 0; jp < 64; ++jp) {
        const uint32_t mj = __builtin_amdgcn_readlane(m, jp);
        const float g0j = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gp0), jp));
        const float g1j = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gp1), jp));
        accw0 += ((mj >> c) & 1u) ? (second ? g1j : g0j) : 0.f;
      }
    }
    accb0 += gp0;
    accb1 += gp1;
  }
  accb0 = evf_wave_sum(accb0);
  accb1 = evf_wave_sum(accb1);
  s_p[wv][lane] = accw0;  // [0..31] = dw[0][c], [32..63] = dw[1][c]
  if (lane == 0) {
    s_p[wv][2 * C32] = accb0;
    s_p[wv][2 * C32 + 1] = accb1;
  }
  __syncthreads();
  if (threadIdx.x < 2 * C32 + 2) {
    const float v = (s_p[0][threadIdx.x] + s_p[1][threadIdx.x]) + (s_p[2][threadIdx.x] + s_p[3][threadIdx.x]);
    evf_atomic_add(threadIdx.x < 2 * C32 ? dw + threadIdx.x : dbias + (threadIdx.x - 2 * C32), v);
  }
}

extern "C" int evf_pred_bwd(const uint32_t* x, const float* flow, const float* g_flow, const float* w, int B, int H,
                            int W, float* g_x, float* dw, float* dbias, void* stream) {
  if (!x || !flow || !g_flow || !w || !g_x || !dw || !dbias || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  const long nb = ((long)B * H * W + 255) / 256;
  hipLaunchKernelGGL(k_pred_bwd, dim3((int)(nb < PB_BLOCKS ? nb : PB_BLOCKS)), dim3(256), 0, EVF_STREAM(stream), x, flow,
                     g_flow, w, B, H * W, g_x, dw, dbias);
  return evf_status();
}

// --------------------------------------------------------------------------
// layout conversions for the state API (models/model.py:203-209)
// --------------------------------------------------------------------------
__global__ void k_bits_to_nchw(const uint32_t* __restrict__ bits, int B, int HW, float* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*32*HW
  if (e >= (long)B * C32 * HW) return;
  const long q = e % HW, c = (e / HW) % C32, b = e / ((long)HW * C32);
  out[e] = (float)((bits[b * HW + q] >> c) & 1u);
}
__global__ void k_nchw_to_bits(const float* __restrict__ in, int B, int HW, uint32_t* __restrict__ bits) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)B * HW) return;
  const long b = p / HW, q = p % HW;
  uint32_t m = 0u;
  for (int c = 0; c < C32; ++c) m |= (in[(b * C32 + c) * HW + q] != 0.f ? 1u : 0u) << c;
  bits[p] = m;
}
__global__ void k_nhwc_to_nchw(const float* __restrict__ in, int B, int C, int HW, float* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * C * HW) return;
  const long q = e % HW, c = (e / HW) % C, b = e / ((long)HW * C);
  out[e] = in[(b * HW + q) * C + c];
}
__global__ void k_nchw_to_nhwc(const float* __restrict__ in, int B, int C, int HW, float* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * C * HW) return;
  const long c = e % C, q = (e / C) % HW, b = e / ((long)HW * C);
  out[e] = in[(b * C + c) * HW + q];
}

// pixel-major spike words [B][H][W] -> channel-major bit planes [B][H][32][ceil(W/32)]
__global__ void k_bits_transpose(const uint32_t* __restrict__ bits, int B, int H, int W, uint32_t* __restrict__ planes) {
  const int nW = (W + 31) / 32;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*H*32*nW
  if (e >= (long)B * H * C32 * nW) return;
  const int xw = (int)(e % nW), c = (int)((e / nW) % C32);
  const long row = e / ((long)nW * C32);
  uint32_t m = 0u;
  for (int k = 0; k < 32; ++k) {
    const int x = xw * 32 + k;
    if (x < W) m |= ((bits[row * W + x] >> c) & 1u) << k;
  }
  planes[e] = m;
}
extern "C" int evf_bits_transpose(const uint32_t* bits, int B, int H, int W, uint32_t* planes, void* stream) {
  if (!bits || !planes || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  const long n = (long)B * H * C32 * ((W + 31) / 32);
  hipLaunchKernelGGL(k_bits_transpose, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), bits, B, H, W, planes);
  return evf_status();
}
extern "C" int evf_bits_to_nchw(const uint32_t* bits, int B, int H, int W, float* out, void* stream) {
  if (!bits || !out || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_bits_to_nchw, dim3(evf_cdiv((long)B * C32 * H * W, 256)), dim3(256), 0, EVF_STREAM(stream), bits,
                     B, H * W, out);
  return evf_status();
}
extern "C" int evf_nchw_to_bits(const float* in, int B, int H, int W, uint32_t* bits, void* stream) {
  if (!in || !bits || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_nchw_to_bits, dim3(evf_cdiv((long)B * H * W, 256)), dim3(256), 0, EVF_STREAM(stream), in, B, H * W,
                     bits);
  return evf_status();
}
extern "C" int evf_nhwc_to_nchw(const float* in, int B, int C, int H, int W, float* out, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(evf_cdiv((long)B * C * H * W, 256)), dim3(256), 0, EVF_STREAM(stream), in, B, C,
                     H * W, out);
  return evf_status();
}
extern "C" int evf_nchw_to_nhwc(const float* in, int B, int C, int H, int W, float* out, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(evf_cdiv((long)B * C * H * W, 256)), dim3(256), 0, EVF_STREAM(stream), in, B, C,
                     H * W, out);
  return evf_status();
}

// --------------------------------------------------------------------------
// optimiser: global-norm clip + Adam on one flat buffer (train_flow.py:157-163)
// --------------------------------------------------------------------------
// ws[0] = sum of squares (zeroed per call), ws[1] = optimizer step counter kept on the
// device (so that a captured hipGraph replays with an advancing step).
__global__ void k_sumsq(const float* __restrict__ g, long n, float* __restrict__ ws, int device_step) {
  __shared__ float red[16];
  // float4 trips with four accumulators (the scalar single-chain loop read 81 MB of LIF-EV-FlowNet gradient in 46 us = 1.8 TB/s)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, gs = (long)gridDim.x * blockDim.x;
  if ((((uintptr_t)g) & 15) == 0) {
    const long n4 = n >> 2;
    const float4* g4 = (const float4*)g;
    long i = gt;
    for (; i + gs < n4; i += 2 * gs) {
      const float4 a = g4[i], b = g4[i + gs];
      s0 += a.x * a.x + b.x * b.x, s1 += a.y * a.y + b.y * b.y, s2 += a.z * a.z + b.z * b.z, s3 += a.w * a.w + b.w * b.w;
    }
    if (i < n4) {
      const float4 a = g4[i];
      s0 += a.x * a.x, s1 += a.y * a.y, s2 += a.z * a.z, s3 += a.w * a.w;
    }
    for (long j = (n4 << 2) + gt; j < n; j += gs) s0 += g[j] * g[j];
  } else {
    for (long i = gt; i < n; i += gs) s0 += g[i] * g[i];
  }
  float s = (s0 + s1) + (s2 + s3);
  s = evf_block_sum(s, red);
  if (threadIdx.x == 0) {
    evf_atomic_add(ws, s);
    if (device_step && blockIdx.x == 0) ws[1] += 1.0f;  // single writer
  }
}

__global__ void k_clip_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float max_norm, float lr, float b1, float b2,
                            float host_step_size, float host_bc2_sqrt, float eps, const float* __restrict__ ws,
                            int device_step, int zero_grad) {
  // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(1.f, max_norm / (sqrtf(ws[0]) + 1e-6f));
  float step_size = host_step_size, bc2_sqrt = host_bc2_sqrt;
  if (device_step) {  // bias corrections from the device-side counter
    // in double, like the host path: an fp32 1 - b2^t is off by ~3e-5 at small t, and the spiking network
    // amplifies such a difference into visibly different trajectories after a few steps
    const double t = (double)ws[1];
    step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    if (zero_grad) g[i] = 0.f;  // optimizer.zero_grad() of the next step, without its fill kernel
  }
}

int evf_clip_adam_step_impl(float* param, float* grad, float* m, float* v, int64_t n, float max_norm, float lr, float beta1,
                            float beta2, float eps, int step, float* norm_ws, int zero_grad, void* stream);
extern "C" int evf_clip_adam_step(float* param, float* grad, float* m, float* v, int64_t n, float max_norm,
                                  float lr, float beta1, float beta2, float eps, int step, float* norm_ws,
                                  int zero_grad, void* stream) {
  return evf_clip_adam_step_impl(param, grad, m, v, n, max_norm, lr, beta1, beta2, eps, step, norm_ws, zero_grad, stream);
}
int evf_clip_adam_step_impl(float* param, float* grad, float* m, float* v, int64_t n, float max_norm, float lr, float beta1,
                            float beta2, float eps, int step, float* norm_ws, int zero_grad, void* stream) {
  if (!param || !grad || !m || !v || !norm_ws || n <= 0) return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  const int device_step = step <= 0;  // step <= 0: use (and advance) the counter in norm_ws[1]
  int rc = evf_hip(evf_memset_async(norm_ws, 0, sizeof(float), st));
  if (rc) return rc;
  const int nblk = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_sumsq, dim3(nblk), dim3(256), 0, st, grad, (long)n, norm_ws, device_step);
  double bc1 = 1.0, bc2 = 1.0;
  if (!device_step) {
    bc1 = 1.0 - pow((double)beta1, (double)step);
    bc2 = 1.0 - pow((double)beta2, (double)step);
  }
  hipLaunchKernelGGL(k_clip_adam, dim3(nblk), dim3(256), 0, st, param, grad, m, v, (long)n, max_norm, lr, beta1, beta2,
                     (float)((double)lr / bc1), (float)sqrt(bc2), eps, norm_ws, device_step, zero_grad);
  return evf_status();
}
