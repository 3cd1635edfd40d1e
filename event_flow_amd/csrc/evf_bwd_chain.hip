// One backward kernel per (layer pair, pass) of the fused 32->32 LIF stack (gfx950):
//
//     input gradient of layer l   g_z(l-1) = conv^T(g_cur(l), W_ff(l))            [evf_conv_dgrad_b3_f32]
//   + neuron backward of layer l-1  g_cur(l-1), g_v_prev(l-1), dleak, dthresh     [evf_lif_bwd_wgrad, element-wise part]
//   + weight gradients of layer l-1 dW_ff(l-1) (+ dW_rec(l-1))                    [evf_lif_bwd_wgrad, matrix part]
//   (+ for a recurrent layer l: g_z(l) of the previous pass = conv^T(g_cur(l), W_rec(l)))
//
// The input-gradient tile leaves the matrix cores in registers (lane = channel, 16 pixels of one row) and goes
// straight through the neuron backward: dL/d(spikes) of layer l-1 is never written to HBM and never read back, and a
// (layer, pass) costs one launch instead of two.  Same arithmetic as the separate kernels: exact 3-way bf16 splits of
// the fp32 operands, fp32 accumulation, the six-term product for the real x real input gradient.
//
// STATUS: correct (bit-identical g_cur / g_v_prev / g_x2, weight gradients to 3e-7 of the separate kernels) but not
// yet faster: 53 / 68 / 75 us (plain / with the recurrent input gradient / recurrent lower layer) against
// 51 / 64 / 58 us for evf_conv_dgrad_b3_f32[_pair] + evf_lif_bwd_wgrad at B=8, 128x128 -- its phases (stage halo,
// MFMAs, load operands, neuron backward, re-layout, MFMAs) run back to back, where the separate fused backward streams
// its units through a double-buffered pipeline, and the recurrent form spills (4 x 16 persistent accumulators).  The
// engine therefore uses it only with EVF_CHAIN=1; tests/test_gpu_network.py keeps it honest.
//
// Block = 8 waves = an 8-row x 32-pixel tile (wave = row), persistent over samples like the input-gradient kernel.
// LDS: split weights 54 KiB | halo image of g_cur(l) 3 x 352 x 64 B = 66 KiB, overlaid after the input-gradient MFMAs
// by the split g_cur(l-1) tile in weight-gradient operand order (48 KiB) | spike bit planes of the tile + halo
// (2 x 3.75 KiB) | byte -> 8 x bf16 table (4 KiB).
#include "evf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define C32 32
#define BC_ROWS 8
#define BC_THREADS (BC_ROWS * 64)
#define NFRAG 54
#define BC_HW 34
#define BC_HP ((BC_ROWS + 2) * BC_HW)  // 340 halo pixels
#define BC_HPP 352                      // padded
#define BC_NW 3                         // plane words per (row, channel): tile word + one halo word each side
#define BC_PLANE ((BC_ROWS + 2) * C32 * BC_NW)
typedef __attribute__((address_space(3))) void bc_lds_void;
typedef __attribute__((address_space(1))) const void bc_glb_void;

__device__ __forceinline__ int bc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ float bc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ uint32_t bc_bf16(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bc_surrogate(int kind, float x, float width) {
  switch (kind) {  // models/spiking_util.py:38-43, 55-65, 74-79, 88-93 (as in evf_bwd_fused.hip)
    case EVF_SUPERSPIKE: {
      const float d = 1.0f + width * fabsf(x);
      return __builtin_amdgcn_rcpf(d * d);
    }
    case EVF_TRIANGLE:
      return fmaxf(0.f, 1.0f - width * fabsf(x));
    case EVF_MULTIGAUSS: {
      const float s2 = 6.f * width, k = 0.3989422804014327f;
      auto gs = [&](float v, float mu, float sg) { return expf(-((v - mu) * (v - mu)) / (2.f * sg * sg)) / sg * k; };
      return 1.15f * gs(x, 0.f, width) - 0.15f * gs(x, width, s2) - 0.15f * gs(x, -width, s2);
    }
    default:
      return __builtin_amdgcn_rcpf(1.0f + width * x * x);
  }
}

struct BcArgs {
  // layer l (upper): its current gradient and transposed split weights
  const float4* g_hi;   // g_cur(l) [B,H,W,32] fp32
  const uint4* wt;      // evf_pack_conv_weight_b3t(W_ff(l))
  const uint4* wt2;     // evf_pack_conv_weight_b3t(W_rec(l)) or NULL
  float* gx2;           // [B,H,W,32] written: dL/d(previous output spikes of layer l); with wt2
  // layer l-1 (lower)
  const float* gz_add;  // [B,H,W,32] or NULL: the recurrent part of dL/d(spikes of layer l-1), added to the tile
  const float* g_v_out; // [B,H,W,32] or NULL
  const float* v_out;   // [B,H,W,32]
  const float* v_prev;  // [B,H,W,32] or NULL
  const uint32_t* z_prev;  // [B,H,W] or NULL
  const uint32_t* xT;      // [B,H,32,nW] input spike planes of layer l-1
  const uint32_t* zT_prev; // [B,H,32,nW] or NULL (REC)
  const float* leak;
  const float* thresh;
  float* g_cur;      // [B,H,W,32] written
  float* g_v_prev;   // [B,H,W,32] written
  float* g_leak;     // [32] accumulated
  float* g_thresh;   // [32] accumulated
  float* slab_ff;    // [blocks][9][32][32]
  float* slab_rec;   // or NULL
  int B, H, W, hard_reset, surrogate, accumulate;
  float width;
};

template <bool REC>
__global__ __launch_bounds__(BC_THREADS) void k_bwd_chain(BcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_w = (uint4*)smem_raw;                       // NFRAG*64 uint4 = 54 KiB
  uint4* s_a = s_w + NFRAG * 64;                       // [3][BC_HPP][4] uint4 = 66 KiB (halo image)
  unsigned short* s_b = (unsigned short*)s_a;          // overlay: [3][256 px * 32 ch] bf16 = 48 KiB
  uint32_t* s_px = (uint32_t*)(s_a + 3 * BC_HPP * 4);  // [10][32][3]
  uint32_t* s_pz = s_px + BC_PLANE;                    // same (REC)
  uint4* s_lut = (uint4*)(s_pz + BC_PLANE);            // [256]
  float* s_red = (float*)(s_lut + 256);                // [2][8][32]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int H = a.H, W = a.W, B = a.B;
  const int y0 = blockIdx.y * BC_ROWS, y = y0 + wv, x0 = blockIdx.x * 32;
  const int i = lane & 31, kg = lane >> 5;
  const int nW = (W + 31) / 32;
  const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;

  if (tid < 256) {  // byte -> 8 x bf16 {0, 1.0}
    const uint32_t t = tid;
    auto pr = [&](int e) { return ((t >> e) & 1u) * 0x3F80u | (((t >> (e + 1)) & 1u) * 0x3F80u) << 16; };
    s_lut[tid] = make_uint4(pr(0), pr(2), pr(4), pr(6));
  }
  for (int u = wv; u < NFRAG; u += BC_ROWS)
    __builtin_amdgcn_global_load_lds((bc_glb_void*)(a.wt + u * 64 + lane), (bc_lds_void*)(s_w + u * 64), 16, 0, 0);
  int wset = 0;  // weight set currently in LDS

  const float lam = bc_sigmoid(a.leak[i]), th = fmaxf(a.thresh[i], 0.01f), oml = 1.0f - lam, inv_oml = 1.0f / oml;
  float sl = 0.f, st = 0.f;  // this lane's channel: sums for dleak / dthresh
  f32x16 wacc = {0}, wacc8 = {0}, zacc = {0}, zacc8 = {0};
  const int tdy = wv / 3, tdx = wv % 3;  // taps 0..7 by wave; tap 8 = (2, 2) is shared

  const float* p_add = a.gz_add ? a.gz_add : a.v_out;
  const float* p_gv = a.g_v_out ? a.g_v_out : a.v_out;
  const float* p_vp = a.v_prev ? a.v_prev : a.v_out;
  const uint32_t* p_zw = a.z_prev ? a.z_prev : a.xT;
  const uint32_t* p_zt = REC ? a.zT_prev : a.xT;

  int tile_it = 0;
  for (int b = blockIdx.z; b < B; b += gridDim.z, ++tile_it) {
    __syncthreads();  // previous tile: every wave is done with the LDS operands
    // ---- stage the fp32 halo of g_cur(l) as three bf16 planes (exact split), swizzled like evf_conv_dgrad_b3
    {
      constexpr int NIT = (BC_HP * 4 + BC_THREADS - 1) / BC_THREADS;
      float4 lo4[NIT], hi4[NIT];
#pragma unroll
      for (int n = 0; n < NIT; ++n) {
        const int it = min(tid + n * BC_THREADS, BC_HP * 4 - 1), p = it >> 2, c = it & 3;
        const int hr = p / BC_HW, hc = p - hr * BC_HW;
        const int yy = min(max(y0 - 1 + hr, 0), H - 1), xx = min(max(x0 - 1 + hc, 0), W - 1);  // out-of-image: masked at use
        const float4* src = a.g_hi + (((long)b * H + yy) * W + xx) * 8 + 2 * c;
        lo4[n] = src[0], hi4[n] = src[1];
      }
      // spike planes of layer l-1's inputs for the tile + halo (960 words: two per thread)
      uint32_t pxw[2], pzw[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int t2 = min(tid + h2 * BC_THREADS, BC_PLANE - 1), wq = t2 % BC_NW, c = (t2 / BC_NW) % C32, r = t2 / (BC_NW * C32);
        const int yy = y0 - 1 + r, xw = x0 / 32 - 1 + wq;
        const bool in = yy >= 0 && yy < H && xw >= 0 && xw < nW;
        const long src = in ? (((long)b * H + yy) * C32 + c) * nW + xw : 0;
        const uint32_t vx = a.xT[src], vz = p_zt[src];
        pxw[h2] = in ? vx : 0u, pzw[h2] = in ? vz : 0u;
      }
#pragma unroll
      for (int n = 0; n < NIT; ++n) {
        const int it = tid + n * BC_THREADS;
        if (it < BC_HP * 4) {
          const int p = it >> 2, c = it & 3;
          const float v[8] = {lo4[n].x, lo4[n].y, lo4[n].z, lo4[n].w, hi4[n].x, hi4[n].y, hi4[n].z, hi4[n].w};
          uint32_t t3[3][8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t hi = bc_bf16(v[e]);
            const float r1 = v[e] - __uint_as_float(hi << 16);
            const uint32_t mid = bc_bf16(r1);
            const float r2 = r1 - __uint_as_float(mid << 16);
            t3[0][e] = hi, t3[1][e] = mid, t3[2][e] = bc_bf16(r2);
          }
          const int slot = p * 4 + (c ^ ((p >> 2) & 3));
#pragma unroll
          for (int sp = 0; sp < 3; ++sp)
            s_a[sp * BC_HPP * 4 + slot] = make_uint4(t3[sp][0] | (t3[sp][1] << 16), t3[sp][2] | (t3[sp][3] << 16),
                                                     t3[sp][4] | (t3[sp][5] << 16), t3[sp][6] | (t3[sp][7] << 16));
        }
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
        if (tid + h2 * BC_THREADS < BC_PLANE) {
          s_px[tid + h2 * BC_THREADS] = pxw[h2];
          if (REC) s_pz[tid + h2 * BC_THREADS] = pzw[h2];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- input gradient: 108 MFMAs per wave and weight set
    auto matrix_phase = [&]() -> f32x16 {
      f32x16 acc = {0};
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        const bool yin = yy >= 0 && yy < H;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int tau = dy * 3 + dx;
          const int xx = x0 + i + dx - 1;
          const uint32_t msk = (yin && xx >= 0 && xx < W) ? 0xFFFFFFFFu : 0u;
          const int hp = (wv + dy) * BC_HW + i + dx, sw = (hp >> 2) & 3;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const uint4* wf = s_w + ((tau * 2 + m) * 3) * 64 + lane;
            const uint4 w0 = wf[0], w1 = wf[64], w2 = wf[128];
            const bf16x8 wh = *(const bf16x8*)&w0, wm = *(const bf16x8*)&w1, wl = *(const bf16x8*)&w2;
            const int slot = hp * 4 + ((2 * m + kg) ^ sw);
            uint4 u0 = s_a[slot], u1 = s_a[BC_HPP * 4 + slot], u2 = s_a[2 * BC_HPP * 4 + slot];
            u0.x &= msk, u0.y &= msk, u0.z &= msk, u0.w &= msk;
            u1.x &= msk, u1.y &= msk, u1.z &= msk, u1.w &= msk;
            u2.x &= msk, u2.y &= msk, u2.z &= msk, u2.w &= msk;
            const bf16x8 ah = *(const bf16x8*)&u0, am = *(const bf16x8*)&u1, al = *(const bf16x8*)&u2;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
          }
        }
      }
      return acc;
    };
    f32x16 gz = {0};
    const int nset = a.wt2 ? 2 : 1;
    for (int k = 0; k < nset; ++k) {
      const int set = a.wt2 ? ((tile_it + k) & 1) : 0;
      if (set != wset) {  // swap the weight set
        __syncthreads();
        const uint4* wsrc = set ? a.wt2 : a.wt;
        for (int u = wv; u < NFRAG; u += BC_ROWS)
          __builtin_amdgcn_global_load_lds((bc_glb_void*)(wsrc + u * 64 + lane), (bc_lds_void*)(s_w + u * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        wset = set;
      }
      const f32x16 acc = matrix_phase();
      if (set == 0) {
        gz = acc;
      } else if (y < H) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = x0 + bc_row(r, lane);
          if (col < W) a.gx2[((b * H + y) * W + col) * C32 + i] = acc[r];
        }
      }
    }

    // ---- layer l-1 operands of this lane's 16 pixels (channel i): requested after the input-gradient MFMAs
    // (held across them they cost 80 registers and the kernel spilled)
    const int yq = min(y, H - 1);
    float addv[16], gvo[16], vo[16], vp[16];
    uint32_t zw[16];
    // (one 32-bit element offset per pixel serves all five tensors: uniform base + offset addressing instead of
    //  five 64-bit pointers per pixel; the host wrapper guarantees the tensors are below 2 GiB)
    const int rowpix = (b * H + yq) * W;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pix = rowpix + min(x0 + bc_row(r, lane), W - 1), e = pix * C32 + i;
      const float t0 = p_add[e], t1 = p_gv[e], t2 = p_vp[e];
      vo[r] = a.v_out[e];
      const uint32_t t3 = p_zw[a.z_prev ? pix : 0];
      addv[r] = a.gz_add ? t0 : 0.f;
      gvo[r] = a.g_v_out ? t1 : 0.f;
      vp[r] = a.v_prev ? t2 : 0.f;
      zw[r] = a.z_prev ? t3 : 0u;
    }
    // ---- neuron backward of layer l-1 (autograd of spiking_submodules.py:103-126 / :523-551), lane = channel i
    float gc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = x0 + bc_row(r, lane);
      const bool ok = y < H && col < W;
      const float z = (float)((zw[r] >> i) & 1u);
      const float sg = bc_surrogate(a.surrogate, vo[r] - th, a.width);
      const float gsp = (gz[r] + addv[r]) * sg;
      const float gv = gvo[r] + gsp;
      gc[r] = ok ? gv * oml : 0.f;
      float gp, cur, dlam, dth = 0.f;
      if (a.hard_reset) {
        gp = gv * lam * (1.0f - z);
        cur = (vo[r] - (vp[r] * lam) * (1.0f - z)) * inv_oml;
        dlam = vp[r] * (1.0f - z) - cur;
      } else {
        gp = gv * lam;
        cur = (vo[r] - vp[r] * lam + z * th) * inv_oml;
        dlam = vp[r] - cur;
        dth = gv * z;
      }
      if (ok) {
        sl += gv * dlam;
        st -= dth + gsp;
        const int e = (rowpix + col) * C32 + i;
        a.g_cur[e] = gc[r];
        a.g_v_prev[e] = gp;
      }
    }
    __syncthreads();  // every wave is done reading the halo image: overlay it with the weight-gradient operand
    // exact split of g_cur(l-1), B-operand order: k-step ks = 2*row + (col >> 4), element ((ks*2 + kgp)*32 + ch)*8 + ee
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t t3[3][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g0 = gc[4 * q + e];
        const uint32_t hi = bc_bf16(g0);
        const float r1 = g0 - __uint_as_float(hi << 16);
        const uint32_t mid = bc_bf16(r1);
        const float r2 = r1 - __uint_as_float(mid << 16);
        t3[0][e] = hi, t3[1][e] = mid, t3[2][e] = bc_bf16(r2);
      }
      const int o = ((((wv * 2 + (q >> 1)) * 2 + (q & 1)) * C32 + i) * 8 + 4 * kg);  // ushort index
#pragma unroll
      for (int sp = 0; sp < 3; ++sp)
        *(uint2*)(s_b + sp * (256 * C32) + o) = make_uint2(t3[sp][0] | (t3[sp][1] << 16), t3[sp][2] | (t3[sp][3] << 16));
    }
    __syncthreads();
    // ---- weight gradients of layer l-1: A = spike planes (bits -> bf16 through the table), B = split g_cur(l-1)
    {
      const uint4* sbh = (const uint4*)s_b;
      auto afrag = [&](const uint32_t* planes, int prow, int ddx, int h) -> bf16x8 {
        const int q = 32 + 16 * h + 8 * kg + ddx - 1;  // bit offset of the first of the 8 pixels
        const uint32_t* wr = planes + (prow * C32 + i) * BC_NW + (q >> 5);
        const uint32_t byte = __funnelshift_r(wr[0], wr[1], q & 31) & 0xFFu;
        const uint4 t = s_lut[byte];
        return *(const bf16x8*)&t;
      };
#pragma unroll 2
      for (int ks = 0; ks < 16; ++ks) {
        const int rr = ks >> 1, h = ks & 1;
        const int fo = (ks * 2 + kg) * C32 + i;  // uint4 index: this lane's 8 pixels of channel i (= co)
        const uint4 uh = sbh[fo], um = sbh[256 * C32 / 8 + fo], ul = sbh[2 * 256 * C32 / 8 + fo];
        const bf16x8 bh = *(const bf16x8*)&uh, bm = *(const bf16x8*)&um, bl = *(const bf16x8*)&ul;
        const bf16x8 ax = afrag(s_px, rr + tdy, tdx, h);
        wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bh, wacc, 0, 0, 0);
        wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bm, wacc, 0, 0, 0);
        wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bl, wacc, 0, 0, 0);
        if (REC) {
          const bf16x8 az = afrag(s_pz, rr + tdy, tdx, h);
          zacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bh, zacc, 0, 0, 0);
          zacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bm, zacc, 0, 0, 0);
          zacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bl, zacc, 0, 0, 0);
        }
        if ((ks & 7) == wv) {  // this wave's share of the ninth tap
          const bf16x8 a8 = afrag(s_px, rr + 2, 2, h);
          wacc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bh, wacc8, 0, 0, 0);
          wacc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bm, wacc8, 0, 0, 0);
          wacc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bl, wacc8, 0, 0, 0);
          if (REC) {
            const bf16x8 z8 = afrag(s_pz, rr + 2, 2, h);
            zacc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z8, bh, zacc8, 0, 0, 0);
            zacc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z8, bm, zacc8, 0, 0, 0);
            zacc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z8, bl, zacc8, 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- weight-gradient slabs (one per block): taps 0..7 from the owning wave, the ninth through LDS
  __syncthreads();
  {
    const long off = (long)blk * (9 * C32 * C32) + wv * (C32 * C32) + i;
    float* d = a.slab_ff + off;
    float* dz = REC ? a.slab_rec + off : d;
    float old[16], oldz[REC ? 16 : 1], prev8[2], prev8z[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      old[q] = d[bc_row(q, lane) * C32];
      if (REC) oldz[q] = dz[bc_row(q, lane) * C32];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long o8 = (long)blk * (9 * C32 * C32) + 8 * (C32 * C32) + tid + h * BC_THREADS;
      prev8[h] = a.slab_ff[o8];
      if (REC) prev8z[h] = a.slab_rec[o8];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      d[bc_row(q, lane) * C32] = (a.accumulate ? old[q] : 0.f) + wacc[q];
      if (REC) dz[bc_row(q, lane) * C32] = (a.accumulate ? oldz[q] : 0.f) + zacc[q];
    }
    float* s_t8 = (float*)smem_raw;  // [8][1024], aliases the weights (done)
    auto reduce_t8 = [&](const f32x16& v8, float* slab, const float (&prev)[2]) {
#pragma unroll
      for (int q = 0; q < 16; ++q) s_t8[wv * (C32 * C32) + bc_row(q, lane) * C32 + i] = v8[q];
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = tid + h * BC_THREADS;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s_t8[w * (C32 * C32) + e];
        slab[(long)blk * (9 * C32 * C32) + 8 * (C32 * C32) + e] = (a.accumulate ? prev[h] : 0.f) + v;
      }
      __syncthreads();
    };
    reduce_t8(wacc8, a.slab_ff, prev8);
    if (REC) reduce_t8(zacc8, a.slab_rec, prev8z);
  }
  // ---- dleak / dthresh: lane = channel; the two half waves and the 8 waves meet in LDS
  sl += __shfl_xor(sl, 32, 64);
  st += __shfl_xor(st, 32, 64);
  if (lane < 32) {
    s_red[(0 * 8 + wv) * C32 + lane] = sl;
    s_red[(1 * 8 + wv) * C32 + lane] = st;
  }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, c = tid & 31;
    float v = 0.f;
    for (int w = 0; w < 8; ++w) v += s_red[(which * 8 + w) * C32 + c];
    if (which == 0) {
      const float l = bc_sigmoid(a.leak[c]);
      evf_atomic_add(a.g_leak + c, v * l * (1.0f - l));
    } else if (a.thresh[c] > 0.01f) {
      evf_atomic_add(a.g_thresh + c, v);
    }
  }
}

#define BC_LDS ((NFRAG * 64 + 3 * BC_HPP * 4 + 256) * 16 + 2 * BC_PLANE * 4 + 2 * 8 * C32 * 4)

static int bc_zb(int B, int H, int W) {
  const long tiles = (long)evf_cdiv(W, 32) * evf_cdiv(H, BC_ROWS);
  int zb = B;
  for (int z = 1; z < B; ++z)
    if (B % z == 0 && tiles * z <= 256 && tiles * z >= 192) zb = z;
  return zb;
}

// number of blocks = weight-gradient slabs evf_bwd_chain writes
extern "C" int evf_bwd_chain_slabs(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return evf_cdiv(W, 32) * evf_cdiv(H, BC_ROWS) * bc_zb(B, H, W);
}

extern "C" int evf_bwd_chain(const float* g_cur_hi, const void* wT_ff_hi, const void* wT_rec_hi, float* g_x2_hi,
                             const float* gz_add, const float* g_v_out, const float* v_out, const float* v_prev,
                             const uint32_t* z_prev, const uint32_t* xT, const uint32_t* zT_prev, const float* leak,
                             const float* thresh, int B, int H, int W, int hard_reset, int surrogate, float act_width,
                             float* g_cur, float* g_v_prev, float* g_leak, float* g_thresh, float* slab_ff,
                             float* slab_rec, int accumulate, void* stream) {
  if (!g_cur_hi || !wT_ff_hi || !v_out || !xT || !leak || !thresh || !g_cur || !g_v_prev || !g_leak || !g_thresh || !slab_ff ||
      B <= 0 || H <= 0 || W <= 0 || (long)B * H * W * C32 * 4 >= (1L << 31) || ((wT_rec_hi != nullptr) != (g_x2_hi != nullptr)) ||
      ((zT_prev != nullptr) != (slab_rec != nullptr)) || g_cur == g_cur_hi)
    return EVF_EINVAL;
  BcArgs a;
  a.g_hi = (const float4*)g_cur_hi, a.wt = (const uint4*)wT_ff_hi, a.wt2 = (const uint4*)wT_rec_hi, a.gx2 = g_x2_hi;
  a.gz_add = gz_add, a.g_v_out = g_v_out, a.v_out = v_out, a.v_prev = v_prev, a.z_prev = z_prev, a.xT = xT, a.zT_prev = zT_prev;
  a.leak = leak, a.thresh = thresh, a.g_cur = g_cur, a.g_v_prev = g_v_prev, a.g_leak = g_leak, a.g_thresh = g_thresh;
  a.slab_ff = slab_ff, a.slab_rec = slab_rec;
  a.B = B, a.H = H, a.W = W, a.hard_reset = hard_reset, a.surrogate = surrogate, a.accumulate = accumulate, a.width = act_width;
  dim3 grid(evf_cdiv(W, 32), evf_cdiv(H, BC_ROWS), bc_zb(B, H, W)), block(BC_THREADS);
  static bool a1 = false, a2 = false;
  if (zT_prev) {
    if (!a2) {
      (void)hipFuncSetAttribute((const void*)k_bwd_chain<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS);
      a2 = true;
    }
    hipLaunchKernelGGL(k_bwd_chain<true>, grid, block, BC_LDS, EVF_STREAM(stream), a);
  } else {
    if (!a1) {
      (void)hipFuncSetAttribute((const void*)k_bwd_chain<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS);
      a1 = true;
    }
    hipLaunchKernelGGL(k_bwd_chain<false>, grid, block, BC_LDS, EVF_STREAM(stream), a);
  }
  return evf_status();
}
