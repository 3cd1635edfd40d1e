// Forward spiking conv cell on the bf16 matrix cores with fp32-equivalent
// numerics ("bf16x3"): the inputs of every 32->32 layer are binary spikes,
// which bf16 represents exactly, and an fp32 weight is the exact sum of three
// bf16 values  w = hi + mid + lo  (8+8+8 mantissa bits).  So
//     sum_k z_k * w_k = sum_k z_k*hi_k + sum_k z_k*mid_k + sum_k z_k*lo_k
// with exact products and fp32 accumulation inside v_mfma_f32_32x32x16_bf16:
// three MFMAs of 32 cycles cover K = 16, against eight fp32 MFMAs of 64 cycles
// (5.3x fewer matrix-core cycles); the rounding error is that of an fp32
// accumulation, like any re-ordered fp32 convolution.  The kernel becomes
// HBM bound (v in / v out), which is where a fused elementwise update belongs.
//
// A operand: lane (i, kg) needs the 8 channels 16m+8kg .. +7 of pixel i as
// bf16: one byte of the pixel's spike word -> 16 bytes through a 256-entry
// LDS table (one ds_read_b128, broadcast for the all-zero byte).
// B operand: packed per (tap, m, term) as 64 lanes x 16 B (k_pack_conv_weight_b3).
#include "evf_fwd.h"

#ifndef TH
#define TH 8
#endif
#define FW_WAVES (TH / 2)        // a wave owns two tile rows
#define FW_THREADS (64 * FW_WAVES)
#define HALO_H (TH + 2)

__device__ __forceinline__ int b3_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ uint32_t bf16_rne(float f) {  // round-to-nearest-even, finite inputs
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// dst[((tau*2+m)*3+s)*64 + lane] (uint4 = 8 bf16): element e = term s of w[co=lane&31][ci=16m+8(lane>>5)+e][tau]
__global__ void k_pack_conv_weight_b3(const float* __restrict__ w, uint4* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (tau*2+m)*64 + lane
  if (idx >= 18 * 64) return;
  const int lane = idx & 63, tm = idx >> 6, m = tm & 1, tau = tm >> 1;
  const int j = lane & 31, kg = lane >> 5;
  uint32_t t[3][8];
  for (int e = 0; e < 8; ++e) {
    const float v = w[(j * C32 + (16 * m + 8 * kg + e)) * 9 + tau];
    const uint32_t hi = bf16_rne(v);
    const float r1 = v - __uint_as_float(hi << 16);
    const uint32_t mid = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(mid << 16);
    const uint32_t lo = bf16_rne(r2);
    t[0][e] = hi, t[1][e] = mid, t[2][e] = lo;
  }
  for (int s = 0; s < 3; ++s)
    dst[(tm * 3 + s) * 64 + lane] = make_uint4(t[s][0] | (t[s][1] << 16), t[s][2] | (t[s][3] << 16),
                                               t[s][4] | (t[s][5] << 16), t[s][6] | (t[s][7] << 16));
}

extern "C" int evf_pack_conv_weight_b3(const float* w, int Cout, int Cin, void* dst, void* stream) {
  if (!w || !dst || Cout != C32 || Cin != C32) return EVF_EINVAL;
  hipLaunchKernelGGL(k_pack_conv_weight_b3, dim3(evf_cdiv(18 * 64, 256)), dim3(256), 0, EVF_STREAM(stream), w,
                     (uint4*)dst);
  return evf_status();
}


// PLIF (spiking_submodules.py:191-227, :618-657): a per-channel pre-synaptic trace
//   pt' = pt*sigma(leak_pt) + (1 - sigma(leak_pt)) * AvgPool3x3(mean_c |input|)
// is subtracted from the current, cur = ff (+ rec) - sigma(add_pt) * pt'.  For binary inputs
// mean_c|x| = popcount(word)/32, pooled here from the spike-word halo already in LDS.
struct PlifArgs {
  const float* leak_pt;
  const float* add_pt;
  const float* pt_prev;  // [B,H,W,32] or NULL
  float* pt_out;         // [B,H,W,32]
  float* P_out;          // [B,H,W] pooled pre-synaptic activity (saved for the backward)
  int xl;                // 1: XLIF cell (spiking_submodules.py:337-435, :771-875): add_pt = t1, thresh = t0; threshold t0 + t1 * pt', the
                         // current stays ff (+ rec).  2: ALIF cell (:230-334, :660-768): the same, the trace (leak_pt = leak_t) driven by
                         // the cell's own previous spikes.  Entry points: bits 1-2 of `hard_reset`
};

#ifdef EVF_SPAN  // start / end of every block in the chip-wide 100 MHz counter (probe build through EVF_LIB)
__device__ unsigned long long fw_span[2 * 4096];
extern "C" int evf_debug_fw_span(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(fw_span), sizeof(fw_span))); }
#define FW_SPAN_MARK(w)                                                                                                     \
  do {                                                                                                                      \
    const int bid_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                                        \
    if (threadIdx.x == 0 && bid_ < 4096) fw_span[2 * bid_ + (w)] = __builtin_amdgcn_s_memrealtime();                        \
  } while (0)
#else
#define FW_SPAN_MARK(w) do {} while (0)
#endif

template <bool REC, bool PLIF>
__device__ __forceinline__ void fwd_b3_body(const int b, const uint32_t* __restrict__ x, const uint4* __restrict__ wff,
                                                         const uint4* __restrict__ wrec,
                                                         const float* __restrict__ leak,
                                                         const float* __restrict__ thresh,
                                                         const float* __restrict__ v_prev,
                                                         const uint32_t* __restrict__ z_prev, int B, int H, int W,
                                                         int hard_reset, float* __restrict__ v_out,
                                                         uint32_t* __restrict__ z_out,
                                                         uint32_t* __restrict__ zT_out, PlifArgs pl, PredArgs pr) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_w = (uint4*)smem_raw;    // NFRAG*64
  uint4* s_lut = s_w + NFRAG * 64;  // 256
  uint32_t* s_x = (uint32_t*)(s_lut + 256);
  uint32_t* s_z = s_x + HALO_H * HALO_W;
  float* s_P = (float*)(s_z + HALO_H * HALO_W);  // TH*TW (PLIF)
  float* s_pw = s_P + TH * TW;                   // 2*32 + 2 prediction-head weights and bias
  float* s_par = s_pw + 2 * C32 + 2;             // [5][32]: sigmoid(leak), clamped thresh, sigmoid(leak_pt), alpha, beta (PLIF: alpha =
                                                 // sigmoid(add_pt), beta = 0; XLIF: alpha = 0, beta = max(t1, 0))
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;  // b: the sample (blockIdx.z of the one-layer kernel)
  FW_SPAN_MARK(0);

  // per-channel parameters: requested FIRST and by every thread (no load under a branch).  Memory returns in order: as
  // `if (tid < 32) s_par[tid] = sigmoid(leak[tid])` below the halo loads, wave 0 waited for them there and issued its
  // state prefetch one round trip after the other seven waves.
  const float par_leak = leak[tid & 31], par_thresh = thresh[tid & 31];
  const float par_lpt = PLIF ? pl.leak_pt[tid & 31] : 0.f, par_apt = PLIF ? pl.add_pt[tid & 31] : 0.f;
#ifndef PROBE_NO_WEIGHT_DMA  // (probe build: what does staging the 54 KiB of weights per block cost?)
  for (int u = wv; u < NFRAG; u += FW_WAVES) b3_glds16(wff + u * 64 + lane, s_w + u * 64);
#endif
  {
    const float* psrc = pr.w ? pr.w : leak;  // no load under a branch: dummy source, the values are unused then
    const float* bsrc = pr.w ? pr.bias : leak;
    const float pwv = psrc[min(tid, 2 * C32 - 1) % (pr.w ? 2 * C32 : C32)], pbv = bsrc[tid & 1];
    if (tid < 2 * C32) s_pw[tid] = pwv;
    if (tid < 2) s_pw[2 * C32 + tid] = pbv;
  }
  if (tid < 256) {  // byte -> 8 x bf16 {0, 1.0}
    const uint32_t t = tid;
    auto pr = [&](int e) { return ((t >> e) & 1u) * 0x3F80u | (((t >> (e + 1)) & 1u) * 0x3F80u) << 16; };
    s_lut[tid] = make_uint4(pr(0), pr(2), pr(4), pr(6));
  }
  {
    const uint32_t* zsrc = z_prev ? z_prev : x;  // no load under a branch: clamped address, select afterwards
    for (int i = tid; i < HALO_H * HALO_W; i += FW_THREADS) {
      const int yy = y0 + i / HALO_W - 1, xx = x0 + i % HALO_W - 1;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const long p = ((long)b * H + min(max(yy, 0), H - 1)) * W + min(max(xx, 0), W - 1);
      const uint32_t vx = x[p], vz = zsrc[p];
      s_x[i] = in ? vx : 0u;
      s_z[i] = (in && z_prev) ? vz : 0u;
    }
  }
  const int i = lane & 31, kg = lane >> 5, j = lane & 31;
  const int r0 = 2 * wv;
  // The matrix phase computes the TRANSPOSED tile (weights as the A operand): a lane owns PIXEL i of its two rows
  // and the 16 channels 8q + 4kg .. +3 (q = 0..3), so every state tensor moves as float4 -- 4 memory instructions per
  // tensor, row and lane instead of 16 (the dword-per-lane epilogue was bound by the texture addresser: 128
  // instructions per wave).  Prefetch of the previous state: in flight during the MFMAs.
  float4 vp[2][4];
  float4 ptp[PLIF ? 2 : 1][PLIF ? 4 : 1];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const long pq = ((long)b * H + min(y0 + r0 + m, H - 1)) * W + min(x0 + i, W - 1);
    const float* src = v_prev ? v_prev : v_out;  // unconditional (clamped) loads, selected afterwards
    const float* psrc = (PLIF && pl.pt_prev) ? pl.pt_prev : v_out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 val = *(const float4*)(src + pq * C32 + 8 * q + 4 * kg);
      vp[m][q] = v_prev ? val : make_float4(0.f, 0.f, 0.f, 0.f);
      if (PLIF) {
        const float4 pv4 = *(const float4*)(psrc + pq * C32 + 8 * q + 4 * kg);
        ptp[m][q] = pl.pt_prev ? pv4 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  if (tid < C32) {  // per-channel constants, once per block (operands: top of the kernel)
    s_par[tid] = b3_sigmoid(par_leak);             // torch.sigmoid(self.leak)     spiking_submodules.py:111/:536
    s_par[C32 + tid] = fmaxf(par_thresh, 0.01f);   // self.thresh.clamp_min(0.01)  :108/:533
    s_par[2 * C32 + tid] = PLIF ? evf_plif_sigmoid(par_lpt) : 0.f;
    // cur = (ff + rec) - alpha * pt', threshold = thresh + beta * pt': one of the two is zero (x - 0 * pt' = x, th + 0 * pt' = th exactly)
    s_par[3 * C32 + tid] = (PLIF && !pl.xl) ? evf_plif_sigmoid(par_apt) : 0.f;
    s_par[4 * C32 + tid] = (PLIF && pl.xl) ? fmaxf(par_apt, 0.f) : 0.f;  // self.t1.clamp_min(0)  :365/:810
  }
  // The weight DMA (invisible to the compiler's counters) was issued before everything else and memory
  // returns in order: once at most the state prefetches issued above (8, or 16 with the PLIF trace) are still
  // outstanding, the DMA has landed -- the matrix phase does not wait for v_prev.
  if (PLIF)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  if (PLIF) {  // pooled pre-synaptic activity of the tile's 256 pixels (one per thread)
    const int py = tid >> 5, px = tid & 31;
    int cnt = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) cnt += __popc(s_x[(py + dy) * HALO_W + px + dx]);
    const float P = ((float)cnt / 32.0f) / 9.0f;  // sum of the 9 means (exact), then AvgPool's division
    s_P[tid] = P;
    if (y0 + py < H && x0 + px < W) pl.P_out[((long)b * H + y0 + py) * W + x0 + px] = P;
    __syncthreads();
  }

  f32x16 acc0 = {0}, acc1 = {0};
  auto conv_phase = [&](const uint32_t* __restrict__ sb) {
#pragma unroll 1
    for (int tau = 0; tau < 9; ++tau) {
      const int dy = tau / 3, dx = tau % 3;
      const uint32_t w0 = sb[(r0 + dy) * HALO_W + i + dx] >> (8 * kg);
      const uint32_t w1 = sb[(r0 + 1 + dy) * HALO_W + i + dx] >> (8 * kg);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const uint4 a0u = s_lut[(w0 >> (16 * m)) & 0xFFu], a1u = s_lut[(w1 >> (16 * m)) & 0xFFu];
        const bf16x8 a0 = *(const bf16x8*)&a0u, a1 = *(const bf16x8*)&a1u;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const uint4 bu = s_w[((tau * 2 + m) * 3 + s) * 64 + lane];
          const bf16x8 bw = *(const bf16x8*)&bu;
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, a0, acc0, 0, 0, 0);  // weights as A: transposed product
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, a1, acc1, 0, 0, 0);
        }
      }
    }
  };
  conv_phase(s_x);
  if (REC) {
    __syncthreads();  // every wave is done with the ff weights
    for (int u = wv; u < NFRAG; u += FW_WAVES) b3_glds16(wrec + u * 64 + lane, s_w + u * 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    conv_phase(s_z);
  }

  // Epilogue (transposed tile): lane = pixel (x0 + i) of rows y0 + r0, y0 + r0 + 1; channel c = 8q + e + 4kg.
  // v_out leaves through a wave-private staging tile [32 pixels][FW_SP floats] laid over the (now dead) weights, so that a
  // store instruction writes 8 FULL 128-byte lines and can be non-temporal: the state does not stay dirty in the L2s for
  // the end-of-kernel write-back (see evf_dgrad_b3.hip; the blocks of this kernel were done 3 us before it retired).
  __syncthreads();  // every wave is done with the weights
  float* stg = (float*)s_w + wv * (32 * FW_SP);
  const int nW = (W + 31) / 32;
  const int rj = (j & 3) + 4 * (j >> 3), kgj = (j >> 2) & 1;  // (r, half) whose ballot holds channel j's bit plane
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const f32x16& acc = m ? acc1 : acc0;
    const int row = y0 + r0 + m;
    const bool ok = row < H && x0 + i < W;
    const long pix = ((long)b * H + min(row, H - 1)) * W + min(x0 + i, W - 1);
    const uint32_t zw = s_z[(r0 + m + 1) * HALO_W + 1 + i];  // previous output spikes of this pixel
    const float Pq = PLIF ? s_P[(r0 + m) * TW + i] : 0.f;
    uint32_t bits = 0u, myplane = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v4[4] = {vp[m][q].x, vp[m][q].y, vp[m][q].z, vp[m][q].w};
      const float p4[4] = {PLIF ? ptp[m][q].x : 0.f, PLIF ? ptp[m][q].y : 0.f, PLIF ? ptp[m][q].z : 0.f, PLIF ? ptp[m][q].w : 0.f};
      float vo4[4], po4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e, c = 8 * q + e + 4 * kg;
        const float lam = s_par[c], th = s_par[C32 + c];
        const float z = (float)((zw >> c) & 1u);
        float cur = acc[r];
        float pto = 0.f, th_e = th, th_p = th;  // threshold of this element now / at the previous pass (soft reset)
        if (PLIF) {
          const float lpt = s_par[2 * C32 + c], apt = s_par[3 * C32 + c], bet = s_par[4 * C32 + c];
          pto = evf_plif_trace(p4[e], lpt, pl.xl == 2 ? z : Pq);  // :212 / :642; ALIF: t * leak_t + (1 - leak_t) * z, :311 / :744
          cur = cur - apt * pto;                  // (ff + rec) - add_pt * pt_out, :220 / :650
          th_e = th + bet * pto;                  // XLIF: thresh = t0 + t1 * pt_out, :419 / :864
          th_p = th + bet * p4[e];                // XLIF soft reset: - z * (t0 + t1 * pt), :430 / :871
        }
        // both reset rules evaluated, one selected: no per-element branch on the (uniform) flag
        const float vo_hard = (v4[e] * lam) * (1.0f - z) + (1.0f - lam) * cur;  // :119/:544
        const float vo_soft = v4[e] * lam + (1.0f - lam) * cur - z * th_p;      // :121/:546
        const float vo = hard_reset ? vo_hard : vo_soft;
        const bool spike = ok && (vo - th_e) > 0.f;
        vo4[e] = vo, po4[e] = pto;
        bits |= (spike ? 1u : 0u) << c;
        // channel-major bit plane of channel c over the tile's 32 pixels = this ballot (low half: kg = 0)
        const unsigned long long mk = __ballot(spike);
        myplane = (r == rj) ? (kgj ? (uint32_t)(mk >> 32) : (uint32_t)mk) : myplane;
      }
      *(float4*)(stg + i * FW_SP + 8 * q + 4 * kg) = make_float4(vo4[0], vo4[1], vo4[2], vo4[3]);
      if (ok && PLIF) *(float4*)(pl.pt_out + pix * C32 + 8 * q + 4 * kg) = make_float4(po4[0], po4[1], po4[2], po4[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
      const float4 v = *(const float4*)(stg + p * FW_SP + c4);
      if (row < H && x0 + p < W) evf_store_nt(v_out + (((long)b * H + row) * W + x0 + p) * C32 + c4, v);
    }
    __builtin_amdgcn_wave_barrier();  // (the tile is rewritten for the wave's second row)
    const uint32_t word = bits | __shfl_xor(bits, 32, 64);  // the pixel's 32 output spikes
    if (ok && kg == 0) z_out[pix] = word;
    if (pr.w) {  // (block-uniform) the prediction head on this pixel's spike word, summed like evf_pred_fwd
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int c = 0; c < C32; ++c) {
        const float z = (float)((word >> c) & 1u);
        s0 += z * s_pw[c];
        s1 += z * s_pw[C32 + c];
      }
      if (ok && kg == 0) {
        const long hw = (long)H * W, qq = (long)row * W + x0 + i;
        pr.flow[(long)b * 2 * hw + qq] = tanhf(s0 + s_pw[2 * C32]);
        pr.flow[((long)b * 2 + 1) * hw + qq] = tanhf(s1 + s_pw[2 * C32 + 1]);
      }
    }
    if (zT_out && row < H && lane < 32) zT_out[(((long)b * H + row) * C32 + j) * nW + x0 / 32] = myplane;
  }
  FW_SPAN_MARK(1);
}

template <bool REC, bool PLIF>
__global__ __launch_bounds__(FW_THREADS) void k_conv_lif_fwd_b3(const uint32_t* __restrict__ x, const uint4* __restrict__ wff,
                                                         const uint4* __restrict__ wrec,
                                                         const float* __restrict__ leak,
                                                         const float* __restrict__ thresh,
                                                         const float* __restrict__ v_prev,
                                                         const uint32_t* __restrict__ z_prev, int B, int H, int W,
                                                         int hard_reset, float* __restrict__ v_out,
                                                         uint32_t* __restrict__ z_out,
                                                         uint32_t* __restrict__ zT_out, PlifArgs pl, PredArgs pr) {
  fwd_b3_body<REC, PLIF>(blockIdx.z, x, wff, wrec, leak, thresh, v_prev, z_prev, B, H, W, hard_reset, v_out, z_out, zT_out, pl,
                         pr);
}

// ---- several independent (pass, layer) cells of a window in ONE launch ------------------------------------------------
// Cell (t, l) of the fused stack needs (t, l-1) and (t-1, l) only: all cells with the same t + l are independent.  A
// window's forward is recorded by evf_fwd_defer_* (below) and launched diagonal by diagonal: P + L - 1 launches of
// up to L cells instead of P x L launches -- every launch pays its fixed cost (cold first fetch, kernel boundary:
// ~8 of the ~15 us of a single cell) once.  blockIdx.z = cell * B + sample; same body, same results.
__global__ __launch_bounds__(FW_THREADS) void k_fwd_diag(FwJobs jobs, int B, int H, int W) {
  const int jb = blockIdx.z / B, b = blockIdx.z - jb * B;
  const FwJob& J = jobs.j[jb];
  const PlifArgs none{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (J.wrec)
    fwd_b3_body<true, false>(b, J.x, J.wff, J.wrec, J.leak, J.thresh, J.v_prev, J.z_prev, B, H, W, J.hard_reset, J.v_out,
                             J.z_out, J.zT_out, none, J.pr);
  else
    fwd_b3_body<false, false>(b, J.x, J.wff, J.wrec, J.leak, J.thresh, J.v_prev, J.z_prev, B, H, W, J.hard_reset, J.v_out,
                              J.z_out, J.zT_out, none, J.pr);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_fwd_diag_p: the cells of a diagonal as ONE PERSISTENT launch (256 blocks x 8 waves), the forward counterpart of
// k_dgrad_diag_dma.
//
// k_fwd_diag above runs one 8-row x 32-pixel tile per 4-wave block: every block stages 54 KiB of split weights by LDS-DMA
// (the CU's DMA path takes ~30 B/clk: 1.8 k cycles, twice for a recurrent cell) for 3.7 k cycles of MFMAs, builds the byte
// table, and its four waves run matrix phase and epilogue in lockstep -- the matrix pipe is busy a third of the time
// (a launch of 5.3 products: 61 us against 22 us of MFMAs).  Here
//   * a block takes a contiguous, cost-weighted range of the launch's (cell, strip) list (a recurrent cell's strip counts
//     twice), stages the weights of a cell ONCE -- feed-forward and recurrent set side by side, 108 KiB -- and the table once;
//   * the unit of work is a STRIP of 2 rows x 32 pixels owned by ONE wave: the wave loads its own 4 x 34 halo words
//     (x and previous z), prefetches the previous state, runs its 108 (216) MFMAs and its epilogue, with wave-level
//     synchronisation only.  The eight waves of a block drift apart, so one wave's epilogue (element-wise LIF update,
//     ballots, stores) runs under the MFMAs of the other wave of its SIMD.
// Same products in the same order per accumulator as fwd_b3_body: bit-identical results.  Default neuron path (LIF); PLIF
// cells keep the per-tile kernel.
// ---------------------------------------------------------------------------------------------------------------------
#define FP_WAVES 8
#define FP_THREADS (64 * FP_WAVES)
#define FP_HALO (4 * HALO_W)  // words of a strip's halo: rows y0 - 1 .. y0 + 2
#define FP_LDS ((size_t)2 * WB3_BYTES + 256 * 16 + 4 * C32 * 4 + (2 * C32 + 2 + 2) * 4 + (size_t)FP_WAVES * (2 * FP_HALO * 4 + 32 * FW_SP * 4))

struct FpPlan {
  int njobs;
  int ntx, nyy;      // strips per row of tiles / strip rows per sample
  int nstrips;       // per cell
  int weight[FW_MAX_JOBS];  // relative cost of a strip: 3 feed-forward, 4 recurrent
  int total;         // sum of nstrips * weight
};

#ifdef FP_STAMPS  // phase stamps (debug build through EVF_LIB): [block < 16][wave 0 / 4][128] shader-clock values
__device__ unsigned long long fp_stamps[16 * 2 * 128];
extern "C" int evf_debug_fp_stamps(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(fp_stamps), sizeof(fp_stamps))); }
#define FP_STAMP()                                                                                   \
  do {                                                                                               \
    if (blockIdx.x < 16 && lane == 0 && (wv & 3) == 0 && nst < 128)                                  \
      fp_stamps[(blockIdx.x * 2 + (wv >> 2)) * 128 + nst++] = __builtin_readcyclecounter();         \
  } while (0)
#else
#define FP_STAMP() do {} while (0)
#endif

template <bool HARD>  // the reset rule of every cell of the launch (fixed at compile time: as a per-element select BOTH rules were evaluated, and
                      // a vector instruction beside a saturated matrix pipe issues every ~8 cycles instead of every 2: tools/probes/mma_probe)
__global__ __launch_bounds__(FP_THREADS) void k_fwd_diag_p(FwJobs jobs, FpPlan plan, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_wff = (uint4*)smem_raw;
  uint4* s_wrec = s_wff + NFRAG * 64;
  uint4* s_lut = s_wrec + NFRAG * 64;
  float* s_par = (float*)(s_lut + 256);   // [4][32]
  float* s_pw = s_par + 4 * C32;          // 2*32 + 2 (+2 pad)
  uint32_t* s_halo = (uint32_t*)(s_pw + 2 * C32 + 4);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t* s_x = s_halo + wv * (2 * FP_HALO);
  uint32_t* s_z = s_x + FP_HALO;
  float* stg = (float*)(s_halo + FP_WAVES * 2 * FP_HALO) + wv * (32 * FW_SP);
  const int i = lane & 31, kg = lane >> 5, j = lane & 31;
  const int nW = (W + 31) / 32;
  const int rj = (j & 3) + 4 * (j >> 3), kgj = (j >> 2) & 1;  // (r, half) whose ballot holds channel j's bit plane
  int nst = 0;
  (void)nst;
  FP_STAMP();
  if (tid < 256) {  // byte -> 8 x bf16 {0, 1.0}
    const uint32_t t = tid;
    auto pr8 = [&](int e) { return ((t >> e) & 1u) * 0x3F80u | (((t >> (e + 1)) & 1u) * 0x3F80u) << 16; };
    s_lut[tid] = make_uint4(pr8(0), pr8(2), pr8(4), pr8(6));
  }
  const long lo = ((long)blockIdx.x * plan.total) / gridDim.x, hi = ((long)(blockIdx.x + 1) * plan.total) / gridDim.x;
  long cell0 = 0;  // weighted start of the cell
  bool first = true;
  for (int c = 0; c < plan.njobs; ++c) {
    const int wgt = plan.weight[c];
    const long cw = (long)plan.nstrips * wgt;
    // strips of this cell whose weighted start S = cell0 + i * wgt lies in [lo, hi)
    long a0 = lo - cell0, a1 = hi - cell0;
    cell0 += cw;
    int i0 = a0 <= 0 ? 0 : (int)((a0 + wgt - 1) / wgt), i1 = a1 <= 0 ? 0 : (int)((a1 + wgt - 1) / wgt);
    i0 = min(i0, plan.nstrips), i1 = min(i1, plan.nstrips);
    if (i0 >= i1) continue;  // (block-uniform)
    const FwJob& J = jobs.j[c];
    const bool rec = J.wrec != nullptr;
    if (!first) __syncthreads();  // every wave is done with the previous cell's weights and parameters
    first = false;
    for (int u = wv; u < NFRAG; u += FP_WAVES) b3_glds16(J.wff + u * 64 + lane, s_wff + u * 64);
    if (rec)
      for (int u = wv; u < NFRAG; u += FP_WAVES) b3_glds16(J.wrec + u * 64 + lane, s_wrec + u * 64);
    if (tid < C32) {
      s_par[tid] = b3_sigmoid(J.leak[tid]);           // torch.sigmoid(self.leak)     spiking_submodules.py:111/:536
      s_par[C32 + tid] = fmaxf(J.thresh[tid], 0.01f);  // self.thresh.clamp_min(0.01)  :108/:533
    }
    if (J.pr.w) {
      if (tid < 2 * C32) s_pw[tid] = J.pr.w[tid];
      if (tid < 2) s_pw[2 * C32 + tid] = J.pr.bias[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (Measured and dropped: starting the second wave of every SIMD 60 / 110 x 64 cycles late, so that one wave's matrix phase
    //  meets the other's epilogue: 3.55 / 3.59 against 3.51 ms per step.  A wave alone issues an MFMA every ~62 cycles, its
    //  own LDS-read -> expand -> MFMA chain; two of them together just fill the pipe.)
    FP_STAMP();
    const uint32_t* __restrict__ x = J.x;
    const uint32_t* __restrict__ z_prev = J.z_prev;
    const float* __restrict__ v_prev = J.v_prev;
    float* __restrict__ v_out = J.v_out;
    uint32_t hx[3], hz[3];
    bool hin[3];
    auto halo_fetch = [&](int sj) {  // (past the range: the last strip again, never committed)
      const int sk = min(sj, i1 - 1);
      const int tx = sk % plan.ntx, rr = sk / plan.ntx, yy = rr % plan.nyy, b = rr / plan.nyy;
      const int y0 = 2 * yy, x0 = tx * TW;
      const uint32_t* zsrc = z_prev ? z_prev : x;  // no load under a branch: clamped address, select afterwards
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int l = min(lane + 64 * q, FP_HALO - 1), hr = l / HALO_W, hc = l - hr * HALO_W;
        const int ya = y0 + hr - 1, xa = x0 + hc - 1;
        hin[q] = ya >= 0 && ya < H && xa >= 0 && xa < W;
        const long p = ((long)b * H + min(max(ya, 0), H - 1)) * W + min(max(xa, 0), W - 1);
        hx[q] = x[p], hz[q] = zsrc[p];
      }
    };
    // (Measured and dropped: all eight waves taking their matrix phases together and their epilogues together, one barrier
    //  between -- a vector instruction issues every ~2 cycles while the matrix pipe idles and only every ~8 beside a saturated
    //  one, tools/probes/mma_probe -- 3.78 ms per step either way: the epilogue's ~10 k cycles are mostly waits (previous
    //  state, staging round trips, stores), not vector issue.)
    for (int si = i0 + wv; si < i1; si += FP_WAVES) {
      const bool valid = true;
      const int tx = si % plan.ntx, rr = si / plan.ntx, yy = rr % plan.nyy, b = rr / plan.nyy;
      const int y0 = 2 * yy, x0 = tx * TW;
      FP_STAMP();
      // ---- the strip's halo words (x and the cell's previous output spikes), 3 per lane and array: requested one strip
      // ahead (`hx`, `hz`, issued behind the previous strip's state prefetch), committed to the wave's LDS rows here
      if (si == i0 + wv) halo_fetch(si);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int l = lane + 64 * q;
        if (l < FP_HALO) s_x[l] = hin[q] ? hx[q] : 0u, s_z[l] = (hin[q] && z_prev) ? hz[q] : 0u;
      }
      // ---- previous state of the strip's two rows: in flight during the MFMAs
      float4 vp[2][4];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const long pq = ((long)b * H + min(y0 + m, H - 1)) * W + min(x0 + i, W - 1);
        const float* src = v_prev ? v_prev : v_out;  // unconditional (clamped) loads, selected afterwards
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 val = *(const float4*)(src + pq * C32 + 8 * q + 4 * kg);
          vp[m][q] = v_prev ? val : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      halo_fetch(si + FP_WAVES);  // the next strip's words: land during this strip's MFMAs
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f32x16 acc0 = {0}, acc1 = {0};
      // Matrix phase, software pipelined: the 18 spike words of the lane's two pixels (9 taps x 2 rows) are read up front, the
      // two A fragments (table look-ups) and the three weight fragments of stage g + 1 are requested BEFORE the 6 MFMAs of stage g
      // and pinned there -- as `ds_read; s_waitcnt; v_mfma` per tap (the rolled loop of fwd_b3_body) the look-up chain
      // word -> table -> MFMA was a dependent LDS round trip per tap that two waves per SIMD cannot hide.
      auto conv_phase = [&](const uint32_t* __restrict__ sb, const uint4* __restrict__ sw) {
        uint32_t wd[9][2];
#pragma unroll
        for (int tau = 0; tau < 9; ++tau) {
          const int dy = tau / 3, dx = tau % 3;
          wd[tau][0] = sb[dy * HALO_W + i + dx] >> (8 * kg);
          wd[tau][1] = sb[(1 + dy) * HALO_W + i + dx] >> (8 * kg);
        }
        uint4 af[2][2], wf[2][3];  // [stage parity][row] / [stage parity][term]; a stage = (tap, K half m): 6 MFMAs
        auto fetch = [&](int g) {
          const int sp = g & 1, tau = g >> 1, m = g & 1;
          af[sp][0] = s_lut[(wd[tau][0] >> (16 * m)) & 0xFFu];
          af[sp][1] = s_lut[(wd[tau][1] >> (16 * m)) & 0xFFu];
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) wf[sp][t3] = sw[(g * 3 + t3) * 64 + lane];
        };
        fetch(0);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
          const int sp = g & 1;
          if (g + 1 < 18) fetch(g + 1);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 a0 = *(const bf16x8*)&af[sp][0], a1 = *(const bf16x8*)&af[sp][1];
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) {
            const bf16x8 bw = *(const bf16x8*)&wf[sp][t3];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, a0, acc0, 0, 0, 0);  // weights as A: transposed product
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, a1, acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      FP_STAMP();
      if (valid) conv_phase(s_x, s_wff);
      FP_STAMP();
      if (valid && rec) conv_phase(s_z, s_wrec);
      FP_STAMP();
      // ---- epilogue (transposed tile, as fwd_b3_body): lane = pixel x0 + i of rows y0, y0 + 1; channel c = 8q + e + 4kg.
      // The lane's 16 leak / threshold values come in as eight 16-byte reads up front (as `s_par[c]` beside each use the
      // element loop was 48 dependent LDS round trips per strip: 6-9 k cycles of epilogue against 4.2 k of MFMAs, phase stamps).
      float lam[16], th[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 l4 = *(const float4*)(s_par + 8 * q + 4 * kg), t4 = *(const float4*)(s_par + C32 + 8 * q + 4 * kg);
        lam[4 * q] = l4.x, lam[4 * q + 1] = l4.y, lam[4 * q + 2] = l4.z, lam[4 * q + 3] = l4.w;
        th[4 * q] = t4.x, th[4 * q + 1] = t4.y, th[4 * q + 2] = t4.z, th[4 * q + 3] = t4.w;
      }
      const uint32_t zw2[2] = {s_z[HALO_W + 1 + i], s_z[2 * HALO_W + 1 + i]};  // previous output spikes of the lane's two pixels
      __builtin_amdgcn_sched_barrier(0);  // (left alone the reads above are sunk next to their uses, one wait each)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x16& acc = m ? acc1 : acc0;
        const int row = y0 + m;
        const bool ok = row < H && x0 + i < W;
        const long pix = ((long)b * H + min(row, H - 1)) * W + min(x0 + i, W - 1);
        const uint32_t zw = zw2[m];
        uint32_t bits = 0u, myplane = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v4[4] = {vp[m][q].x, vp[m][q].y, vp[m][q].z, vp[m][q].w};
          float vo4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e, cc = 8 * q + e + 4 * kg;
            const float z = (float)((zw >> cc) & 1u);
            const float cur = acc[r];
            const float vo = HARD ? (v4[e] * lam[r]) * (1.0f - z) + (1.0f - lam[r]) * cur   // :119/:544
                                  : v4[e] * lam[r] + (1.0f - lam[r]) * cur - z * th[r];     // :121/:546
            const bool spike = ok && (vo - th[r]) > 0.f;
            vo4[e] = vo;
            bits |= (spike ? 1u : 0u) << cc;
            const unsigned long long mk = __ballot(spike);
            myplane = (r == rj) ? (kgj ? (uint32_t)(mk >> 32) : (uint32_t)mk) : myplane;
          }
          *(float4*)(stg + i * FW_SP + 8 * q + 4 * kg) = make_float4(vo4[0], vo4[1], vo4[2], vo4[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        FP_STAMP();
        float4 ev[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[r] = *(const float4*)(stg + (8 * r + (lane >> 3)) * FW_SP + (lane & 7) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
          if (row < H && x0 + p < W) evf_store_nt(v_out + (((long)b * H + row) * W + x0 + p) * C32 + c4, ev[r]);
        }
        __builtin_amdgcn_wave_barrier();  // (the tile is rewritten for the wave's second row)
        FP_STAMP();
        const uint32_t word = bits | __shfl_xor(bits, 32, 64);  // the pixel's 32 output spikes
        if (ok && kg == 0) J.z_out[pix] = word;
        if (J.pr.w) {  // (cell-uniform) the prediction head on this pixel's spike word, summed like evf_pred_fwd
          float s0 = 0.f, s1 = 0.f;
#pragma unroll 1
          for (int c4 = 0; c4 < C32; c4 += 4) {  // (rolled: fully unrolled its 64 table reads were the register peak of the kernel)
            const float4 pa = *(const float4*)(s_pw + c4), pb = *(const float4*)(s_pw + C32 + c4);
            const float pa4[4] = {pa.x, pa.y, pa.z, pa.w}, pb4[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = (float)((word >> (c4 + e)) & 1u);
              s0 += z * pa4[e];
              s1 += z * pb4[e];
            }
          }
          if (ok && kg == 0) {
            const long hw = (long)H * W, qq = (long)row * W + x0 + i;
            J.pr.flow[(long)b * 2 * hw + qq] = tanhf(s0 + s_pw[2 * C32]);
            J.pr.flow[((long)b * 2 + 1) * hw + qq] = tanhf(s1 + s_pw[2 * C32 + 1]);
          }
        }
        if (J.zT_out && row < H && lane < 32) J.zT_out[(((long)b * H + row) * C32 + j) * nW + x0 / 32] = myplane;
      }
      __builtin_amdgcn_wave_barrier();  // (the halo words are rewritten by this wave's next strip)
      FP_STAMP();
    }
  }
  FP_STAMP();
}

#define FW_MAX_DIAGS 96
#define FW_SLOT_JOBS FW_WIN_MAX  // cells an index holds: up to FW_MAX_JOBS independent ones per launch, or a chain of a feed-forward
                                 // layer's passes (evf_fwd_win_is_chain: one k_fwd_win_t launch)
// (one recorder per recording context = per stream, evf_common.h)
struct FwDefer {
  bool active = false;
  int slot = 0;
  int B = 0, H = 0, W = 0;
  int n[FW_MAX_DIAGS] = {0};
  FwJob job[FW_MAX_DIAGS][FW_SLOT_JOBS];
};
static FwDefer fw_tab[EVF_CTX_MAX];

static size_t fw_lds_bytes() {
  return WB3_BYTES + 256 * 16 + 2 * HALO_H * HALO_W * 4 + TH * TW * 4 + (2 * C32 + 2) * 4 + 5 * C32 * 4;
}

static int fw_diag_select = -1;  // -1 environment / default, 0 k_fwd_diag (a tile per block), 1 k_fwd_diag_p, 2 k_fwd_diag_t
extern "C" int evf_fwd_diag_select(int which) {
  if (which < -1 || which > 2) return EVF_EINVAL;
  fw_diag_select = which;
  return EVF_OK;
}

static int launch_fwd_b3_now(const uint32_t* x, const void* wb_ff, const void* wb_rec, const float* leak,
                             const float* thresh, const float* v_prev, const uint32_t* z_prev, int B, int H, int W,
                             int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, const PlifArgs* plif,
                             void* stream, PredArgs pd);

// Launch what has been recorded (diagonals in increasing order) and keep recording.
static int fw_defer_launch(FwDefer& fw_defer, void* stream) {
  {  // the head layer's recorded cells first: every diagonal cell of pass t reads (through its layers below) the head of pass t
    const int rc = evf_hf_defer_launch((int)(&fw_defer - fw_tab), stream);
    if (rc) return rc;
  }
  const size_t lds = fw_lds_bytes();
  static bool attr_set = false;
  if (!attr_set) {
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_fwd_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_fwd_diag_p<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FP_LDS);
    (void)hipFuncSetAttribute((const void*)k_fwd_diag_p<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FP_LDS);
    attr_set = true;
  }
  // EVF_FWD_DIAG=tile|persistent|teams (A/B measurements, the bit-identity tests); default: teams (k_fwd_diag_t, evf_fwd_teams.hip)
  static const int env_mode = []() {
    const char* e = getenv("EVF_FWD_DIAG");
    return !e ? 2 : (e[0] == 't' && e[1] == 'i' ? 0 : (e[0] == 'p' ? 1 : 2));
  }();
  const int mode = fw_diag_select < 0 ? env_mode : fw_diag_select;
  const bool persistent = mode >= 1;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  if (evf_prof_mode() == 2) {  // an empty bracket: what the two event nodes cost by themselves inside the replayed graph
    evf_prof_mark(7, 0, stream);
    evf_prof_mark(7, 1, stream);
  }
  for (int d = 0; d < FW_MAX_DIAGS; ++d) {
    const int ntot = fw_defer.n[d];
    if (!ntot) continue;
    if (evf_fwd_win_is_chain(fw_defer.job[d], ntot)) {  // the passes of a feed-forward layer: one launch, state in registers
      evf_prof_mark(6, 0, stream);
      const int rc = evf_fwd_win_t_launch(fw_defer.job[d], ntot, fw_defer.B, fw_defer.H, fw_defer.W, stream);
      evf_prof_mark(6, 1, stream);
      if (rc) return rc;
      fw_defer.n[d] = 0;
      continue;
    }
    // cells under one index that are neither independent nor one chain (a window interrupted by a state reset, a chain longer
    // than a launch holds): one launch per cell, in the order recorded
    bool dependent = false;
    for (int k = 0; k < ntot && !dependent; ++k)
      for (int j = 0; j < ntot; ++j)
        if (j != k && fw_defer.job[d][k].v_prev && fw_defer.job[d][k].v_prev == fw_defer.job[d][j].v_out) dependent = true;
    const int per_launch = dependent ? 1 : FW_MAX_JOBS;
    for (int k0 = 0; k0 < ntot; k0 += per_launch) {  // (more than FW_MAX_JOBS independent cells under one index: several launches)
    const int n = ntot - k0 < per_launch ? ntot - k0 : per_launch;
    FwJobs jobs;
    for (int k = 0; k < FW_MAX_JOBS; ++k) jobs.j[k] = fw_defer.job[d][k0 + (k < n ? k : 0)];
    int nhard = 0, nplif = 0;
    for (int k = 0; k < n; ++k) nhard += jobs.j[k].hard_reset ? 1 : 0, nplif += jobs.j[k].leak_pt ? 1 : 0;
    evf_prof_mark(0, 0, stream);
    int rc_t = EVF_EINVAL;  // (the teams launch refuses geometries beyond its index arithmetic: those take the next path)
    if ((mode == 2 || nplif) && (nhard == 0 || nhard == n)) {
      rc_t = evf_fwd_diag_t_launch(jobs, n, fw_defer.B, fw_defer.H, fw_defer.W, stream);
      if (rc_t && rc_t != EVF_EINVAL) return rc_t;
    }
    if (rc_t == EVF_OK) {
    } else if (nplif) {  // (only the teams kernel and the one-cell kernel know the trace: cell by cell)
      for (int k = 0; k < n; ++k) {
        const FwJob& J = jobs.j[k];
        const PlifArgs pa{J.leak_pt, J.add_pt, J.pt_prev, J.pt_out, J.P_out, J.xl};
        const int rc = launch_fwd_b3_now(J.x, J.wff, J.wrec, J.leak, J.thresh, J.v_prev, J.z_prev, fw_defer.B, fw_defer.H, fw_defer.W,
                                         J.hard_reset, J.v_out, J.z_out, J.zT_out, J.leak_pt ? &pa : nullptr, stream, J.pr);
        if (rc) return rc;
      }
    } else if (persistent && (nhard == 0 || nhard == n)) {  // (cells of both reset rules in one index: the per-tile kernel)
      FpPlan plan;
      plan.njobs = n, plan.ntx = evf_cdiv(fw_defer.W, TW), plan.nyy = evf_cdiv(fw_defer.H, 2);
      plan.nstrips = plan.ntx * plan.nyy * fw_defer.B;
      plan.total = 0;
      for (int k = 0; k < FW_MAX_JOBS; ++k) {
        plan.weight[k] = (k < n && jobs.j[k].wrec) ? 4 : 3;  // (a recurrent strip costs ~19 k cycles, a feed-forward one ~14.6 k: phase stamps)
        if (k < n) plan.total += plan.nstrips * plan.weight[k];
      }
      const int nblk = plan.total / 24 < ncu ? evf_cdiv(plan.total, 24) : ncu;  // (tiny launches: at least ~8 strips per block)
      if (nhard)
        hipLaunchKernelGGL(k_fwd_diag_p<true>, dim3(nblk), dim3(FP_THREADS), FP_LDS, EVF_STREAM(stream), jobs, plan, fw_defer.B,
                           fw_defer.H, fw_defer.W);
      else
        hipLaunchKernelGGL(k_fwd_diag_p<false>, dim3(nblk), dim3(FP_THREADS), FP_LDS, EVF_STREAM(stream), jobs, plan, fw_defer.B,
                           fw_defer.H, fw_defer.W);
    } else {
      dim3 grid(evf_cdiv(fw_defer.W, TW), evf_cdiv(fw_defer.H, TH), fw_defer.B * n), block(FW_THREADS);
      hipLaunchKernelGGL(k_fwd_diag, grid, block, lds, EVF_STREAM(stream), jobs, fw_defer.B, fw_defer.H, fw_defer.W);
    }
    evf_prof_mark(0, 1, stream);
    }
    fw_defer.n[d] = 0;
  }
  return evf_status();
}

// evf_fwd_defer_begin(): from now on evf_conv_lif_fwd_b3 / evf_conv_lif_fwd_b3_pred RECORD their launch under the
// diagonal index last set by evf_fwd_defer_slot() instead of launching.  evf_fwd_defer_flush() launches the recorded
// cells (one k_fwd_diag per non-empty diagonal, increasing index) and ends the recording; it must run before anything
// reads the cells' outputs.  The caller guarantees that cells recorded under one index are independent and that a cell's
// operands come from lower indices (or from launches issued before).  One recorder per stream (recording contexts, evf_common.h).
static bool fw_poison = false;  // evf_defer_poison: process-wide debug switch
int evf_defer_poisoned() { return fw_poison ? 1 : 0; }
int evf_fwd_defer_active(int ctx) { return fw_tab[ctx].active ? 1 : 0; }
extern "C" int evf_defer_poison(int on) {
  fw_poison = on != 0;
  return EVF_OK;
}

extern "C" int evf_fwd_defer_begin(void* stream) {
  const int c = evf_ctx_find(stream);
  if (c >= 0 && fw_tab[c].active) return EVF_EINVAL;  // one forward recording per stream
  const int ctx = evf_ctx_acquire(stream);
  if (ctx < 0) return EVF_EINVAL;
  FwDefer& fw_defer = fw_tab[ctx];
  fw_defer.active = true;
  fw_defer.slot = 0;
  fw_defer.B = fw_defer.H = fw_defer.W = 0;
  for (int d = 0; d < FW_MAX_DIAGS; ++d) fw_defer.n[d] = 0;
  evf_hf_defer_reset(ctx);
  return EVF_OK;
}
extern "C" int evf_fwd_defer_slot(int d, void* stream) {
  const int c = evf_ctx_find(stream);
  if (c < 0 || !fw_tab[c].active || d < 0 || d >= FW_MAX_DIAGS) return EVF_EINVAL;
  fw_tab[c].slot = d;
  return EVF_OK;
}
extern "C" int evf_fwd_defer_pending(void* stream) {
  const int c = evf_ctx_find(stream);
  if (c < 0 || !fw_tab[c].active) return 0;
  FwDefer& fw_defer = fw_tab[c];
  int n = evf_hf_defer_count(c);  // (head cells)
  for (int d = 0; d < FW_MAX_DIAGS; ++d) n += fw_defer.n[d];
  return n;
}
extern "C" int evf_fwd_defer_flush(void* stream) {
  const int c = evf_ctx_find(stream);
  if (c < 0 || !fw_tab[c].active) return EVF_OK;
  const int rc = fw_defer_launch(fw_tab[c], stream);
  fw_tab[c].active = false;
  evf_ctx_drop(c);
  return rc;
}

static int launch_fwd_b3(const uint32_t* x, const void* wb_ff, const void* wb_rec, const float* leak,
                         const float* thresh, const float* v_prev, const uint32_t* z_prev, int B, int H, int W,
                         int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, const PlifArgs* plif,
                         void* stream, const PredArgs* pred = nullptr) {
  PredArgs pd = pred ? *pred : PredArgs{nullptr, nullptr, nullptr};
  const int fctx = evf_ctx_find(stream);
  FwDefer& fw_defer = fw_tab[fctx < 0 ? 0 : fctx];
  if (fctx >= 0 && fw_defer.active) {  // recorded, launched by evf_fwd_defer_flush (or when a diagonal is full / the geometry changes)
    if (fw_defer.B && (fw_defer.B != B || fw_defer.H != H || fw_defer.W != W)) {
      const int rc = fw_defer_launch(fw_defer, stream);
      if (rc) return rc;
    }
    if (fw_defer.n[fw_defer.slot] == FW_SLOT_JOBS) {  // (the caller opens a new recording before an index overflows)
      const int rc = fw_defer_launch(fw_defer, stream);
      if (rc) return rc;
    }
    fw_defer.B = B, fw_defer.H = H, fw_defer.W = W;
    fw_defer.job[fw_defer.slot][fw_defer.n[fw_defer.slot]++] =
        FwJob{x, (const uint4*)wb_ff, (const uint4*)wb_rec, leak, thresh, v_prev, z_prev, v_out, z_out, zT_out, pd, hard_reset, plif ? plif->xl : 0,
              plif ? plif->leak_pt : nullptr, plif ? plif->add_pt : nullptr, plif ? plif->pt_prev : nullptr,
              plif ? plif->pt_out : nullptr, plif ? plif->P_out : nullptr};
    if (fw_poison) {  // debug aid: the outputs hold conspicuous garbage until the flush has run the cell
      const size_t npix = (size_t)B * H * W;
      int rc = evf_hip(evf_memset_async(v_out, 0xFF, npix * C32 * sizeof(float), EVF_STREAM(stream)));  // 0xFFFFFFFF = NaN
      if (!rc && z_out) rc = evf_hip(evf_memset_async(z_out, 0xFF, npix * sizeof(uint32_t), EVF_STREAM(stream)));
      if (!rc && pd.flow) rc = evf_hip(evf_memset_async(pd.flow, 0xFF, npix * 2 * sizeof(float), EVF_STREAM(stream)));
      if (!rc && plif) rc = evf_hip(evf_memset_async(plif->pt_out, 0xFF, npix * C32 * sizeof(float), EVF_STREAM(stream)));
      if (rc) return rc;
    }
    return EVF_OK;
  }
  return launch_fwd_b3_now(x, wb_ff, wb_rec, leak, thresh, v_prev, z_prev, B, H, W, hard_reset, v_out, z_out, zT_out, plif, stream, pd);
}

// one cell, one launch (k_conv_lif_fwd_b3: one 8 x 32 tile per block)
static int launch_fwd_b3_now(const uint32_t* x, const void* wb_ff, const void* wb_rec, const float* leak,
                             const float* thresh, const float* v_prev, const uint32_t* z_prev, int B, int H, int W,
                             int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, const PlifArgs* plif,
                             void* stream, PredArgs pd) {
  dim3 grid(evf_cdiv(W, TW), evf_cdiv(H, TH), B), block(FW_THREADS);
  hipStream_t st = EVF_STREAM(stream);
  const size_t lds = fw_lds_bytes();
  PlifArgs pa = plif ? *plif : PlifArgs{nullptr, nullptr, nullptr, nullptr, nullptr};
#define EVF_FWD(REC_, PLIF_)                                                                                           \
  do {                                                                                                                 \
  if (lds > 65536) {                                                                                                   \
    static bool attr_set = false;                                                                                      \
    if (!attr_set) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)k_conv_lif_fwd_b3<REC_, PLIF_>,                                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
  }                                                                                                                    \
  hipLaunchKernelGGL((k_conv_lif_fwd_b3<REC_, PLIF_>), grid, block, lds, st, x, (const uint4*)wb_ff,                   \
                     (const uint4*)wb_rec, leak, thresh, v_prev, z_prev, B, H, W, hard_reset, v_out, z_out, zT_out, pa,   \
                     pd);                                                                                              \
  } while (0)
  if (plif) {
    if (wb_rec) EVF_FWD(true, true); else EVF_FWD(false, true);
  } else {
    if (wb_rec) EVF_FWD(true, false); else EVF_FWD(false, false);
  }
#undef EVF_FWD
  return evf_status();
}

extern "C" int evf_conv_lif_fwd_b3(const uint32_t* x, const void* wb_ff, const void* wb_rec, const float* leak,
                                   const float* thresh, const float* v_prev, const uint32_t* z_prev, int B, int H,
                                   int W, int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, void* stream) {
  if (!x || !wb_ff || !leak || !thresh || !v_out || !z_out || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  return launch_fwd_b3(x, wb_ff, wb_rec, leak, thresh, v_prev, z_prev, B, H, W, hard_reset, v_out, z_out, zT_out, nullptr,
                       stream);
}

// evf_conv_lif_fwd_b3 for the layer under the prediction head, with the head (evf_pred_fwd) in its epilogue:
// pred_w [2][32], pred_b [2], flow [B,2,H,W] (written).
extern "C" int evf_conv_lif_fwd_b3_pred(const uint32_t* x, const void* wb_ff, const void* wb_rec, const float* leak,
                                        const float* thresh, const float* v_prev, const uint32_t* z_prev, int B, int H,
                                        int W, int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out,
                                        const float* pred_w, const float* pred_b, float* flow, void* stream) {
  if (!x || !wb_ff || !leak || !thresh || !v_out || !z_out || !pred_w || !pred_b || !flow || B <= 0 || H <= 0 || W <= 0)
    return EVF_EINVAL;
  const PredArgs pd{pred_w, pred_b, flow};
  return launch_fwd_b3(x, wb_ff, wb_rec, leak, thresh, v_prev, z_prev, B, H, W, hard_reset, v_out, z_out, zT_out, nullptr,
                       stream, &pd);
}

extern "C" int evf_conv_plif_fwd_b3(const uint32_t* x, const void* wb_ff, const void* wb_rec, const float* leak_v,
                                    const float* leak_pt, const float* add_pt, const float* thresh, const float* v_prev,
                                    const uint32_t* z_prev, const float* pt_prev, int B, int H, int W, int hard_reset,
                                    float* v_out, uint32_t* z_out, uint32_t* zT_out, float* pt_out, float* P_out,
                                    void* stream) {
  if (!x || !wb_ff || !leak_v || !leak_pt || !add_pt || !thresh || !v_out || !z_out || !pt_out || !P_out || B <= 0 ||
      H <= 0 || W <= 0)
    return EVF_EINVAL;
  PlifArgs pa{leak_pt, add_pt, pt_prev, pt_out, P_out, (hard_reset >> 1) & 3};  // (bits 1-2: 1 an XLIF cell, 2 an ALIF cell)
  hard_reset &= 1;
  return launch_fwd_b3(x, wb_ff, wb_rec, leak_v, thresh, v_prev, z_prev, B, H, W, hard_reset, v_out, z_out, zT_out, &pa,
                       stream);
}

// evf_conv_plif_fwd_b3 for the layer under the prediction head, with the head (evf_pred_fwd) in its epilogue -- what makes a
// PLIF network's window recordable (no launch between the last cell of a pass and the first of the next).
extern "C" int evf_conv_plif_fwd_b3_pred(const uint32_t* x, const void* wb_ff, const void* wb_rec, const float* leak_v,
                                         const float* leak_pt, const float* add_pt, const float* thresh, const float* v_prev,
                                         const uint32_t* z_prev, const float* pt_prev, int B, int H, int W, int hard_reset,
                                         float* v_out, uint32_t* z_out, uint32_t* zT_out, float* pt_out, float* P_out,
                                         const float* pred_w, const float* pred_b, float* flow, void* stream) {
  if (!x || !wb_ff || !leak_v || !leak_pt || !add_pt || !thresh || !v_out || !z_out || !pt_out || !P_out || !pred_w || !pred_b ||
      !flow || B <= 0 || H <= 0 || W <= 0)
    return EVF_EINVAL;
  PlifArgs pa{leak_pt, add_pt, pt_prev, pt_out, P_out, (hard_reset >> 1) & 3};  // (bits 1-2: 1 an XLIF cell, 2 an ALIF cell)
  hard_reset &= 1;
  const PredArgs pd{pred_w, pred_b, flow};
  return launch_fwd_b3(x, wb_ff, wb_rec, leak_v, thresh, v_prev, z_prev, B, H, W, hard_reset, v_out, z_out, zT_out, &pa,
                       stream, &pd);
}
