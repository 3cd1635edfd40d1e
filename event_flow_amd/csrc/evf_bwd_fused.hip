// Fused neuron backward + weight-gradient conv (gfx950).
//
// Per layer and pass the backward needs (a) the elementwise neuron backward
// (evf_lif_bwd: 6 x 128 B/pixel of HBM traffic) and (b) the weight gradients
//   dW_ff [tau][ci][co] = sum_pix x     [pix+tau][ci] * g_cur[pix][co]
//   dW_rec[tau][ci][co] = sum_pix z_prev[pix+tau][ci] * g_cur[pix][co]
// This kernel does both in one pass over the row segment: the threads that
// stream g_z, g_v, v', v through registers compute g_cur / g_v_prev, write
// them out, and drop the exact 3-way bf16 split of g_cur into LDS in MFMA
// B-operand order; the matrix cores then contract it with the binary spike
// operands (bf16-exact) -- "bf16x3": products exact, fp32 accumulation, so the
// result has fp32 round-off like the fp32-MFMA kernel at 1/5 of the matrix
// cycles.  The kernel is HBM bound; the MFMAs ride under the memory stream.
//
// GEMM: M = ci, N = co, K = pixels (16 per v_mfma_f32_32x32x16_bf16).
//   A[ci][pix]: 8 consecutive pixels of one channel = one byte of the
//               channel-major spike bit plane (written by the forward kernels),
//               expanded to 8 bf16 through a 256-entry LDS table;
//   B[pix][co]: 8 consecutive pixels of one channel of g_cur, as hi/mid/lo bf16.
// Tap per wave (8 waves own taps 0..7 over all pixels of the block, the ninth
// tap is shared round-robin) -> no atomics, one slab [9][32][32] per block and
// input, accumulated over the passes of a window.
#include <stdio.h>

#include "evf_common.h"
#include "evf_split.h"
#include <stdlib.h>
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define C32 32
#define FB_CW 64    // pixels per unit (row segment): one float4 of every tensor per thread
#define FB_UNITS 8       // units per block of a one-cell launch = slab rows per (cell, input): evf_lif_bwd_wgrad_slabs
#define FB_UNITS_MAX 64  // a launch may give a block up to this many (fb_blocks_per_cell): one lane of the geometry table each
#define FB_THREADS 512
#define FB_NW (FB_CW / 32 + 2)  // plane words per (row, channel): segment + one halo word each side
#define FB_R0 32768             // LDS region 0: operand double buffer (24 KiB), later the tap-8 reduction (32 KiB)

__device__ __forceinline__ int fb_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ float fb_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ uint32_t fb_bf16(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float fb_surrogate(int kind, float x, float width) {
  switch (kind) {  // models/spiking_util.py:38-43, 55-65, 74-79, 88-93
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
      return __builtin_amdgcn_rcpf(1.0f + width * x * x);  // v_rcp_f32 (1 ulp): the kernel is VALU bound
  }
}

#ifdef FB_SPAN  // start / end of EVERY block in the chip-wide 100 MHz counter (debug build through EVF_LIB)
__device__ unsigned long long fb_span[2 * 1024];
extern "C" int evf_debug_fb_span(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(fb_span), sizeof(fb_span))); }
#define FB_SPAN_MARK(w)                                                                                      \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 1024) fb_span[2 * blockIdx.x + (w)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define FB_SPAN_MARK(w) do {} while (0)
#endif

#ifdef FB_STAMPS  // phase stamps (debug build loaded through EVF_LIB): [block < 16][wave 0 / wave 7][96] shader-clock values
__device__ unsigned long long fb_stamps[16 * 2 * 96];
extern "C" int evf_debug_fb_stamps(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(fb_stamps), sizeof(fb_stamps))); }
#define FB_STAMP()                                                                                   \
  do {                                                                                               \
    if (blockIdx.x < 16 && lane == 0 && (wv == 0 || wv == 7) && nst < 96)                            \
      fb_stamps[(blockIdx.x * 2 + (wv ? 1 : 0)) * 96 + nst++] = __builtin_readcyclecounter();       \
  } while (0)
#else
#define FB_STAMP() do {} while (0)
#endif

struct FbStage {
  float4 gz, gv, vo, vp;
  float4 gz2;            // the second part of dL/d(spikes) (from the cell's own recurrent input gradient, one pass later)
  float f0, f1, q0, q1;  // TOP: flow and dL/dflow of the pixel (x, y components)
  uint32_t zo;           // TOP: the layer's own output spikes (input of the prediction head)
  uint32_t zw;
  uint32_t px, pz, pin;  // one word of the x / z_prev bit planes (threads < 3*32*FB_NW) + in-image mask
};
struct FbStagePlif {  // PLIF cells: the trace backward's operands of the same (pixel, channel quad)
  float4 gk, pp;  // dL/d(pt') carried from pass t + 1, pt of pass t - 1
  float P;        // pooled input activity of the pixel
};

// PLIF (spiking_submodules.py:634-652): the presynaptic trace's backward inside team E (evf_plif_trace_bwd as its own pass read
// g_cur back from HBM and was one more launch per cell).  NULL g_pt_prev = a LIF cell.
struct FbPlif {
  const float4* gpt_carry;  // [B,H,W,32] or NULL (last pass of the window)
  const float4* pt_prev;    // [B,H,W,32] or NULL (zero state)
  const float* P;           // [B,H,W]
  const float *leak_pt, *add_pt;  // [32] raw parameters
  float4* g_pt_prev;        // [B,H,W,32] -> carry of pass t - 1 (may alias gpt_carry: same thread reads, then writes)
  float* g_P;               // [B,H,W] dL/d(pooled activity), raw (the input-gradient kernel applies the pooling's adjoint)
  float *g_leak_pt, *g_add_pt;  // per-block rows (row_ld) or dense (atomics)
  int xl;  // 1: an XLIF cell (spiking_submodules.py:337-435, :771-875; entry points: bits 1-2 of `hard_reset` / of `accumulate`): add_pt =
           // t1, thresh = t0, threshold t0 + t1 * pt' -- the trace takes -t1 * dL/d(thresh) instead of -sigma(add_pt) * dL/d(current).
           // 2: an ALIF cell (:230-334, :660-768): the same, with the trace (leak_pt = leak_t) driven by the cell's OWN previous spikes z
           // (un-detached, :311): no pooled activity, and (1 - sigma(leak_t)) * dL/d(t') flows into dL/d(spikes) of the pass BEFORE --
           // a window launch carries it in registers, a one-pass launch writes it to g_zx (the caller hands it back as g_z_out2)
  float4* g_zx;  // ALIF, one-pass launches: [B,H,W,32] out (may alias g_z_out2: the same thread reads, then writes)
};

// TOP: the (non-recurrent) layer under the 1x1 tanh prediction head (models/model.py:197-199, :265).  The head's
// backward (evf_pred_bwd) runs inside this kernel: per pixel gpre = g_flow * (1 - flow^2); the layer's dL/d(spikes)
// row is gpre_x * Wp[0][c] + gpre_y * Wp[1][c] (never written to HBM), and dWp[o][c] += gpre_o * z[c], db[o] += gpre_o.
struct FbTop {
  const float* flow;    // [B,2,H,W] tanh output of the head
  const float* g_flow;  // [B,2,H,W]
  const float* pred_w;  // [2][32]
  const uint32_t* z_out;  // [B,H,W] this layer's output spikes
  float* dw;            // [2][32] accumulated
  float* db;            // [2]     accumulated
};

// FAST: the reference's default neuron (arctan surrogate, hard reset: configs/train_SNN.yml) fixed at compile time.  With
// the kind and the reset mode as run-time values every channel of the element-wise part carries a `switch` and an `if`:
// uniform branches, but branches -- the four channels' dependent chains (v_rcp, the products behind it) cannot be
// interleaved across them, and the loop body holds the code of all four surrogates (805 VALU instructions).
template <bool REC, bool TOP, bool FAST>
__device__ __forceinline__ void fb_body(
    const int bid, const int nblk_,  // this block and the number of blocks of its cell (blockIdx.x / gridDim.x of a one-cell launch)
    const float4* __restrict__ g_z_out, const float4* __restrict__ g_z_out2, const float4* __restrict__ g_v_out,
    const float4* __restrict__ v_out,
    const float4* __restrict__ v_prev, const uint32_t* __restrict__ z_prev, const uint32_t* __restrict__ xT,
    const uint32_t* __restrict__ zT, const float* __restrict__ leak, const float* __restrict__ thresh, int B, int H,
    int W, int nchunk, long nunits, int hard_reset_rt, int surrogate_rt, float width, int accumulate, int nrows_total,
    float4* __restrict__ g_cur, uint2* __restrict__ g_split, float4* __restrict__ g_v_prev,
    float* __restrict__ g_leak, float* __restrict__ g_thresh, float* __restrict__ slab_ff,
    float* __restrict__ slab_rec, FbTop top, int row_ld) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short* s_b = (unsigned short*)smem_raw;           // [2][3][FB_CW*32] bf16 (region of FB_R0 bytes)
  uint32_t* s_px = (uint32_t*)(smem_raw + FB_R0);             // [2][3][32][FB_NW]
  uint32_t* s_pz = s_px + 2 * 3 * C32 * FB_NW;                // same (REC)
  uint4* s_lut = (uint4*)(s_pz + 2 * 3 * C32 * FB_NW);        // [256]
  float* s_red = (float*)(s_lut + 256);                       // [2][8][32]
  const int hard_reset = FAST ? 1 : hard_reset_rt, surrogate = FAST ? EVF_ARCTAN : surrogate_rt;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 31, kg = lane >> 5;
  const int cg = tid & 7;  // channel group of the elementwise part: channels 4cg..4cg+3
  const int p = tid >> 3;  // pixel of the elementwise part within the 64-pixel unit
  const int nW = (W + 31) / 32;
  int nst = 0;
  (void)nst;
  FB_STAMP();
  FB_SPAN_MARK(0);

  if (tid < 256) {
    const uint32_t t = tid;
    auto pr = [&](int e) { return ((t >> e) & 1u) * 0x3F80u | (((t >> (e + 1)) & 1u) * 0x3F80u) << 16; };
    s_lut[tid] = make_uint4(pr(0), pr(2), pr(4), pr(6));
  }
  float lam[4], th[4], oml[4], inv_oml[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lam[k] = fb_sigmoid(leak[4 * cg + k]);
    th[k] = fmaxf(thresh[4 * cg + k], 0.01f);
    oml[k] = 1.0f - lam[k];
    inv_oml[k] = 1.0f / oml[k];  // per-channel constant: no division in the element loop
  }
  float sl[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0};
  float pwa[4] = {0, 0, 0, 0}, pwb[4] = {0, 0, 0, 0}, dwa[4] = {0, 0, 0, 0}, dwb[4] = {0, 0, 0, 0}, dba = 0.f, dbb = 0.f;
  if (TOP) {
#pragma unroll
    for (int k = 0; k < 4; ++k) pwa[k] = top.pred_w[4 * cg + k], pwb[k] = top.pred_w[C32 + 4 * cg + k];
  }

  // Units are dealt round-robin: block x takes units x, x + gridDim.x, ...  At any moment the
  // resident blocks then stream one contiguous region (consecutive 8 KiB units), which spreads
  // over all HBM channels; a contiguous chunk per block would make every block hit the same
  // few channels at the same time (64 KiB stride between blocks).
  const int nblk = nblk_;
  const int nu = (int)((nunits - (long)bid + nblk - 1) / nblk);
  // Geometry of the block's (<= FB_UNITS_MAX) units: lane k holds unit k's (sample, row, first column), computed ONCE
  // here; a unit's geometry then is three v_readlane into SGPRs.  As two integer divisions by run-time values per call
  // (v_rcp + fix-up chains with VALU -> SALU hops), four calls per unit, it was the "0.75 k cycles of load issue" of the
  // phase stamps.
  static_assert(FB_UNITS_MAX <= 64, "geometry table: one lane per unit of the block");
  int g_b, g_y, g_x0;
  {
    const int u = bid + min(lane, nu - 1) * nblk;  // nunits < 2^31
    const int row = u / nchunk;
    g_b = row / H;
    g_y = row - g_b * H;
    g_x0 = (u - row * nchunk) * FB_CW;
  }
  auto geom = [&](int k, int& b, int& y, int& x0, int& cw) {  // k: block-uniform, < nu
    b = __builtin_amdgcn_readlane(g_b, k);
    y = __builtin_amdgcn_readlane(g_y, k);
    x0 = __builtin_amdgcn_readlane(g_x0, k);
    cw = min(FB_CW, W - x0);
  };
  // stage 1: issue the global loads of unit k (one float4 of each tensor per thread).
  // Straight-line code on purpose: every load is unconditional (indices clamped into valid
  // memory, optional tensors redirected to v_out and zeroed by a select afterwards), so the
  // compiler can keep them in flight behind counted s_waitcnt instead of draining vmcnt(0)
  // at every divergent branch.
  const float4* pgz = g_z_out ? g_z_out : v_out;
  const float4* pgz2 = g_z_out2 ? g_z_out2 : v_out;  // (not under the prediction head)
  const bool has_gz2 = !TOP && g_z_out2 != nullptr;
  const float4* pgv = g_v_out ? g_v_out : v_out;
  const float4* pvp = v_prev ? v_prev : v_out;
  const uint32_t* pzw = z_prev ? z_prev : xT;
  const uint32_t* pzt = REC ? zT : xT;
  const bool has_gz = g_z_out != nullptr, has_gv = g_v_out != nullptr, has_vp = v_prev != nullptr,
             has_zw = z_prev != nullptr;
  const int tpl = min(tid, 3 * C32 * FB_NW - 1);
  const int pl_wq = tpl % FB_NW, pl_c = (tpl / FB_NW) % C32, pl_dy = tpl / (FB_NW * C32);
  auto issue_loads = [&](int k, FbStage& s) {
    int b, y, x0, cw;
    geom(min(k, nu - 1), b, y, x0, cw);
    const long pix0 = ((long)b * H + y) * W + x0;
    const int pc = min(p, cw - 1);
    const long ge = (pix0 + pc) * 8 + cg;
    s.vo = v_out[ge];
    if (TOP) {
      const long hw = (long)H * W, q = (long)y * W + x0 + pc;
      s.f0 = top.flow[(long)b * 2 * hw + q], s.f1 = top.flow[((long)b * 2 + 1) * hw + q];
      s.q0 = top.g_flow[(long)b * 2 * hw + q], s.q1 = top.g_flow[((long)b * 2 + 1) * hw + q];
      s.zo = top.z_out[pix0 + pc];
    } else {
      s.gz = pgz[ge];
      s.gz2 = pgz2[ge];
    }
    s.gv = pgv[ge];
    s.vp = pvp[ge];
    s.zw = pzw[z_prev ? pix0 + pc : 0];
    const int yy = y + pl_dy - 1, xw = x0 / 32 - 1 + pl_wq;
    const bool in = yy >= 0 && yy < H && xw >= 0 && xw < nW;
    const long src = in ? (((long)b * H + yy) * C32 + pl_c) * nW + xw : 0;
    s.px = xT[src];
    s.pz = pzt[src];
    s.pin = in ? 0xFFFFFFFFu : 0u;
  };
  // stage 2: neuron backward in registers, results to HBM, split g_cur + spike planes to LDS
  auto commit = [&](int k, const FbStage& s, int buf) {
    int b, y, x0, cw;
    geom(min(k, nu - 1), b, y, x0, cw);
    const long pix0 = ((long)b * H + y) * W + x0;
    unsigned short* sb = s_b + buf * (3 * FB_CW * C32);
    const bool ok = p < cw && k < nu;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float gp0 = 0.f, gp1 = 0.f;
    if (TOP) {
      gp0 = s.q0 * (1.0f - s.f0 * s.f0);  // tanh' (evf_pred_bwd)
      gp1 = s.q1 * (1.0f - s.f1 * s.f1);
    }
    const float4 gz4 = TOP ? make_float4(gp0 * pwa[0] + gp1 * pwb[0], gp0 * pwa[1] + gp1 * pwb[1], gp0 * pwa[2] + gp1 * pwb[2],
                                         gp0 * pwa[3] + gp1 * pwb[3])
                           : (has_gz ? s.gz : z4);
    float4 gzb4 = z4;
    if (!TOP) gzb4 = has_gz2 ? s.gz2 : z4;
    const float4 gv4 = has_gv ? s.gv : z4, vp4 = has_vp ? s.vp : z4;
    if (TOP && ok) {
      const uint32_t zo = s.zo >> (4 * cg);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool on = (zo >> c) & 1u;
        dwa[c] += on ? gp0 : 0.f;
        dwb[c] += on ? gp1 : 0.f;
      }
      if (cg == 0) dba += gp0, dbb += gp1;
    }
    const float vo[4] = {s.vo.x, s.vo.y, s.vo.z, s.vo.w};
    // (the two parts in the order the accumulating input gradient added them: feed-forward part + recurrent part)
    const float gz[4] = {!TOP ? gz4.x + gzb4.x : gz4.x, !TOP ? gz4.y + gzb4.y : gz4.y, !TOP ? gz4.z + gzb4.z : gz4.z,
                         !TOP ? gz4.w + gzb4.w : gz4.w};
    const float gvo[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, vp[4] = {vp4.x, vp4.y, vp4.z, vp4.w};
    const uint32_t zw = (has_zw ? s.zw : 0u) >> (4 * cg);
    float gc[4], gp[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // autograd of spiking_submodules.py:103-126 / :523-551 (see evf_lif_bwd)
      const float z = (float)((zw >> c) & 1u);
      const float sg = fb_surrogate(surrogate, vo[c] - th[c], width);
      const float gsp = gz[c] * sg;
      const float gv = gvo[c] + gsp;
      gc[c] = gv * oml[c];
      float cur, dlam;
      if (hard_reset) {
        gp[c] = gv * lam[c] * (1.0f - z);
        cur = (vo[c] - (vp[c] * lam[c]) * (1.0f - z)) * inv_oml[c];
        dlam = vp[c] * (1.0f - z) - cur;
      } else {
        gp[c] = gv * lam[c];
        cur = (vo[c] - vp[c] * lam[c] + z * th[c]) * inv_oml[c];
        dlam = vp[c] - cur;
        if (ok) st[c] -= gv * z;
      }
      if (ok) {
        sl[c] += gv * dlam;
        st[c] -= gsp;
      }
    }
    if (ok) {
#ifdef FB_NT_STORES  // (probe build: g_cur is read by the very next kernel, g_v_prev only a pass later)
      if (g_cur) g_cur[pix0 * 8 + tid] = make_float4(gc[0], gc[1], gc[2], gc[3]);
      evf_store_nt(g_v_prev + pix0 * 8 + tid, make_float4(gp[0], gp[1], gp[2], gp[3]));
#else
      if (g_cur) g_cur[pix0 * 8 + tid] = make_float4(gc[0], gc[1], gc[2], gc[3]);
      g_v_prev[pix0 * 8 + tid] = make_float4(gp[0], gp[1], gp[2], gp[3]);
#endif
    }
    // exact split g = hi + mid + lo (bf16 each), stored in B-operand order:
    // pixel p = 16*kq + 8*kgp + ee, element ((kq*2 + kgp)*32 + j)*8 + ee
    // exact split g = hi + mid + lo (bf16 each; two channels per v_cvt_pk_bf16_f32, evf_split.h), stored in B-operand order:
    // pixel p = 16*kq + 8*kgp + ee, element ((kq*2 + kgp)*32 + j)*8 + ee
    uint32_t tp[3][2];  // [term][channel pair]: low half = channel 2e, high half = channel 2e+1
    const int base = ((p >> 3) * C32) * 8 + (p & 7);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      evf_split3_pair(ok ? gc[2 * e] : 0.f, ok ? gc[2 * e + 1] : 0.f, tp[0][e], tp[1][e], tp[2][e]);
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) {
        const int o = t3 * FB_CW * C32 + base + (4 * cg + 2 * e) * 8;
        sb[o] = (unsigned short)tp[t3][e];              // ds_write_b16
        sb[o + 8] = (unsigned short)(tp[t3][e] >> 16);  // ds_write_b16_d16_hi
      }
    }
    if (ok && g_split) {  // the same split as three bf16 planes [term][pix][32] for evf_conv_dgrad_b3
      const long ps = (long)B * H * W * 8;
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) g_split[t3 * ps + pix0 * 8 + tid] = make_uint2(tp[t3][0], tp[t3][1]);
    }
    if (tid < 3 * C32 * FB_NW) {
      s_px[buf * (3 * C32 * FB_NW) + tid] = s.px & s.pin;
      if (REC) s_pz[buf * (3 * C32 * FB_NW) + tid] = s.pz & s.pin;
    }
  };

  f32x16 acc = {0}, acc8 = {0}, accz = {0}, accz8 = {0};
  const int dy = wv / 3, dx = wv % 3;  // taps 0..7; tap 8 = (2, 2) is shared
  // stage 3: matrix cores on the staged unit
  auto mfma_unit = [&](int par) {  // par = parity of the unit = its LDS buffer
    const int buf = par;
    const uint4* sbh = (const uint4*)(s_b + buf * (3 * FB_CW * C32));
    const uint32_t* px = s_px + buf * (3 * C32 * FB_NW);
    const uint32_t* pz = s_pz + buf * (3 * C32 * FB_NW);
    auto afrag = [&](const uint32_t* planes, int ddy, int ddx, int kq) -> bf16x8 {
      const int q = 32 + 16 * kq + 8 * kg + ddx - 1;  // bit offset of the first of the 8 pixels
      const uint32_t* wr = planes + (ddy * C32 + i) * FB_NW + (q >> 5);
      const uint32_t byte = __funnelshift_r(wr[0], wr[1], q & 31) & 0xFFu;
      const uint4 a = s_lut[byte];
      return *(const bf16x8*)&a;
    };
#pragma unroll
    for (int kq = 0; kq < FB_CW / 16; ++kq) {
      const int fo = (kq * 2 + kg) * C32 + i;  // uint4 index of this lane's 8 pixels of channel i (= co)
      const uint4 uh = sbh[fo], um = sbh[FB_CW * C32 / 8 + fo], ul = sbh[2 * FB_CW * C32 / 8 + fo];
      const bf16x8 bh = *(const bf16x8*)&uh, bm = *(const bf16x8*)&um, bl = *(const bf16x8*)&ul;
      const bf16x8 a = afrag(px, dy, dx, kq);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
      if (REC) {
        const bf16x8 az = afrag(pz, dy, dx, kq);
        accz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bh, accz, 0, 0, 0);
        accz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bm, accz, 0, 0, 0);
        accz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bl, accz, 0, 0, 0);
      }
      if (kq + (FB_CW / 16) * par == wv) {  // this wave's share of the ninth tap
        const bf16x8 a8 = afrag(px, 2, 2, kq);
        acc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bh, acc8, 0, 0, 0);
        acc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bm, acc8, 0, 0, 0);
        acc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bl, acc8, 0, 0, 0);
        if (REC) {
          const bf16x8 az8 = afrag(pz, 2, 2, kq);
          accz8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az8, bh, accz8, 0, 0, 0);
          accz8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az8, bm, accz8, 0, 0, 0);
          accz8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az8, bl, accz8, 0, 0, 0);
        }
      }
    }
  };

  // software pipeline with two units of loads always in flight:
  //   unit k: loads(k+2) -> registers | MFMA(k) from LDS[k&1] | commit(k+1) -> LDS[(k+1)&1]
  // The loop is unrolled by two and the two register stages swap ROLES instead of contents: `s_nxt = s_new` at the end
  // of an iteration was 22 v_mov per unit -- and a move of a loaded register is a wait for that load, i.e. the loads of
  // unit k+2 had to land within iteration k instead of by the commit of iteration k+1.  An odd unit count runs one more
  // half-iteration on a unit of zeros (commit of a unit >= nu writes zeros: the MFMAs add nothing).
  FbStage s_cur, s_nxt, s_new;
  issue_loads(0, s_cur);
  issue_loads(1, s_nxt);
  // Operands of the EPILOGUE, requested now (behind the first two units' loads): the previous partial sums of this
  // wave's slab tile, of the ninth tap and of this block's row of per-channel sums.  Loaded after the unit loop they
  // were two dependent HBM round trips (slab read-modify-write, then the row) at the end of every block's life.
  // (REC: the recurrent slab's 16 words per lane stay after the loop -- the kernel is at 222 VGPRs -- but are issued
  //  before the LDS reductions and consumed after them.)
  const long slab_off = (long)bid * (9 * C32 * C32) + wv * (C32 * C32) + i;
  float old[16], prev8[2], prev8z[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 16; ++q) old[q] = slab_ff[slab_off + fb_row(q, lane) * C32];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const long o8 = (long)bid * (9 * C32 * C32) + 8 * (C32 * C32) + tid + h * FB_THREADS;
    prev8[h] = slab_ff[o8];
    if (REC) prev8z[h] = slab_rec[o8];
  }
  const size_t row_off = (size_t)bid * row_ld;
  const int row_c = tid & 31, row_which = (tid >> 5) & 1;
  const float row_prev = (row_which ? g_thresh : g_leak)[row_off + row_c];          // (read by threads < 64)
  float top_prev = 0.f;
  if (TOP) top_prev = tid < 64 ? top.dw[row_off + row_which * C32 + row_c] : top.db[row_off + (tid & 1)];  // (threads < 66)
  FB_STAMP();
  commit(0, s_cur, 0);
  FB_STAMP();
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < nu; k += 2) {
    FB_STAMP();
    issue_loads(k + 2, s_new);
    FB_STAMP();
    mfma_unit(0);
    FB_STAMP();
    commit(k + 1, s_nxt, 1);
    FB_STAMP();
    __syncthreads();
    issue_loads(k + 3, s_nxt);
    mfma_unit(1);
    commit(k + 2, s_new, 0);
    __syncthreads();
  }
  FB_STAMP();

  // ---- weight-gradient slabs: taps 0..7 straight from the owning wave (previous partial sums: see the prologue;
  // as `acc ? *p + a : a` every load sat under a branch and was its own HBM round trip -- s_memtime showed the
  // epilogue taking 30 % (ff) / 42 % (rec) of the kernel; loaded together after the loop it still was one round trip)
  float oldz[REC ? 16 : 1];
  if (REC) {
#pragma unroll
    for (int q = 0; q < 16; ++q) oldz[q] = slab_rec[slab_off + fb_row(q, lane) * C32];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) slab_ff[slab_off + fb_row(q, lane) * C32] = ((accumulate & 1) ? old[q] : 0.f) + acc[q];
  // ---- tap 8: sum the 8 partial tiles through LDS (aliases the operand buffers)
  float* s_t8 = (float*)smem_raw;  // [8][1024]
  auto reduce_t8 = [&](const f32x16& a, float* slab, const float (&prev)[2]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) s_t8[wv * (C32 * C32) + fb_row(q, lane) * C32 + i] = a[q];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // C32*C32 = 2 * FB_THREADS
      const int e = tid + h * FB_THREADS;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += s_t8[w * (C32 * C32) + e];
      slab[(long)bid * (9 * C32 * C32) + 8 * (C32 * C32) + e] = ((accumulate & 1) ? prev[h] : 0.f) + v;
    }
    __syncthreads();
  };
  reduce_t8(acc8, slab_ff, prev8);
  if (REC) {
    reduce_t8(accz8, slab_rec, prev8z);
#pragma unroll
    for (int q = 0; q < 16; ++q) slab_rec[slab_off + fb_row(q, lane) * C32] = ((accumulate & 1) ? oldz[q] : 0.f) + accz[q];
  }

  // ---- first touch of the slabs in this window by a launch with FEWER blocks than slab rows (a diagonal launch whose blocks
  // take 16 units): the rows this launch does not write start at zero, so that later launches -- with either block count --
  // and the window's reduction find every row initialised
  if (!(accumulate & 1) && nblk < nrows_total) {
    for (int r = nblk + bid; r < nrows_total; r += nblk) {
      float4* z0 = (float4*)(slab_ff + (long)r * (9 * C32 * C32));
      float4* z1 = REC ? (float4*)(slab_rec + (long)r * (9 * C32 * C32)) : nullptr;
      for (int e = tid; e < 9 * C32 * C32 / 4; e += FB_THREADS) {
        z0[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (REC) z1[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }

  // ---- per-channel sums for leak / thresh: lanes with equal (lane & 7) share channels
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      sl[c] += __shfl_xor(sl[c], o, 64);
      st[c] += __shfl_xor(st[c], o, 64);
    }
  if (lane < 8) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s_red[(0 * 8 + wv) * C32 + 4 * lane + c] = sl[c];
      s_red[(1 * 8 + wv) * C32 + 4 * lane + c] = st[c];
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, c = tid & 31;
    float v = 0.f;
    for (int w = 0; w < 8; ++w) v += s_red[(which * 8 + w) * C32 + c];
    // row_ld > 0: every block owns ROW blockIdx.x of a [blocks][row_ld] buffer of per-block partial sums (plain
    // read-modify-write, summed once per window by evf_sum_rows).  256 blocks adding atomically into the same 64 words
    // kept the kernel alive 4.5 us after its last block was done (34.5 -> 29.9 us without them).
    const size_t ro = row_off;  // (the row's previous value was requested in the prologue: row_prev)
    if (which == 0) {
      const float l = fb_sigmoid(leak[c]), t = v * l * (1.0f - l);
      if (row_ld) g_leak[ro + c] = row_prev + t;
      else evf_atomic_add(g_leak + c, t);
    } else if (thresh[c] > 0.01f) {
      if (row_ld) g_thresh[ro + c] = row_prev + v;
      else evf_atomic_add(g_thresh + c, v);
    }
  }
  FB_STAMP();
  if (TOP) {  // prediction-head weight / bias gradients, reduced the same way
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        dwa[c] += __shfl_xor(dwa[c], o, 64);
        dwb[c] += __shfl_xor(dwb[c], o, 64);
      }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      dba += __shfl_xor(dba, o, 64);
      dbb += __shfl_xor(dbb, o, 64);
    }
    if (lane < 8) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s_red[(0 * 8 + wv) * C32 + 4 * lane + c] = dwa[c];
        s_red[(1 * 8 + wv) * C32 + 4 * lane + c] = dwb[c];
      }
    }
    float* s_b2 = (float*)smem_raw;  // operand buffers are free now
    if (lane == 0) s_b2[2 * wv] = dba, s_b2[2 * wv + 1] = dbb;
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float v = 0.f;
      for (int w = 0; w < 8; ++w) v += s_red[(which * 8 + w) * C32 + c];
      if (row_ld) top.dw[row_off + which * C32 + c] = top_prev + v;
      else evf_atomic_add(top.dw + which * C32 + c, v);
    } else if (tid < 66) {
      float v = 0.f;
      for (int w = 0; w < 8; ++w) v += s_b2[2 * w + (tid - 64)];
      if (row_ld) top.db[row_off + (tid - 64)] = top_prev + v;
      else evf_atomic_add(top.db + (tid - 64), v);
    }
  }
  FB_SPAN_MARK(1);
}

// ---- the same cell with the block's eight waves in two TEAMS (k_bwd_diag_ws) ---------------------------------------------
// In fb_body every wave runs load -> neuron backward -> stage -> matrix phase one after the other and the block's barrier
// keeps all eight in the same phase: per 64-pixel unit the vector pipes (~2.0 k cycles per SIMD), the matrix pipes (~1.2 k)
// and the LDS (~2.2 k) are busy one AFTER the other, 5.5 k cycles (PMC: SQ_INSTS_VALU / MFMA / LDS_IDX_ACTIVE per unit).
// Here waves 0-3 (team E) stream the tensors, do the neuron backward, write the results and stage the operands of unit
// k + 1, while waves 4-7 (team M) contract unit k on the matrix cores: one wave of each team per SIMD, so a SIMD's vector
// and matrix pipes work at the same time.  Team M: wave m owns taps 2m and 2m + 1 (the B fragments of a K step are read
// once for both) and the ninth tap's K step m of every unit; team E: a thread takes one float4 of every tensor per HALF
// unit (32 pixels), four half units of loads in flight.  Default neuron only (arctan surrogate, hard reset).
// Same products in the same order per tap as fb_body; the ninth tap's partial tiles are grouped differently (4 instead of
// 8) and the per-channel sums run over other thread subsets: equal to fp32 round-off, not bit for bit.
// EW = waves of team E: 4 (a 512-thread block, one wave of each team per SIMD; a thread takes a float4 per HALF unit, four
// half units of loads in flight) or 8 (a 768-thread block: two E waves per SIMD hide each other's dependent chains -- with
// one, team E needed 4.4 k cycles per unit against team M's 3.6 k (phase stamps); a thread takes a float4 per unit).
// WIN (k_bwd_win_plif): ALL passes of a window of one feed-forward cell in one launch.  A block's iteration space is (unit, pass)
// with the passes of a unit back to back, last pass first: dL/dv, dL/d(pt) and the potential of the pass before stay in REGISTERS
// from one iteration to the next (the thread keeps its pixel and channel quad), the matrix team's accumulators run over passes and
// units alike.  Per pixel and pass the kernel then reads dL/dz, v_prev (+ pt_prev, P) and writes dL/d(current) (+ dL/dP): 640 B
// instead of the 1152 B of a one-pass PLIF cell.  Same arithmetic per element; sums of the weight gradient in another order.
#define FB_WIN_MAX 16
struct FbWin {
  int np;  // passes; index s = 0 is the window's LAST pass (backward order)
  const float4* gz[FB_WIN_MAX];   // dL/d(spikes) of pass s (NULL: none)
  const float4* vo[FB_WIN_MAX];   // potential after the pass (read at s = 0 only: vo[s] = vp[s - 1])
  const float4* vp[FB_WIN_MAX];   // potential before the pass (NULL: zero state)
  const uint32_t* zp[FB_WIN_MAX]; // spike words before the pass (NULL: none)
  const uint32_t* xT[FB_WIN_MAX]; // input spike planes of the pass
  float4* gcur[FB_WIN_MAX];       // out: dL/d(current) of the pass (fp32; NULL: not wanted)
  uint2* gsp[FB_WIN_MAX];         // out: its exact 3-way bf16 split, three planes [term][pix][32] (NULL: not wanted) -- k_dgrad_diag_dma
  const float4* pp[FB_WIN_MAX];   // PLIF: trace before the pass (NULL: zero)
  const float* P[FB_WIN_MAX];     // PLIF: pooled activity of the pass
  float* gP[FB_WIN_MAX];          // out: dL/d(pooled activity) of the pass (raw)
  // TOP (the layer under the prediction head, FbTop: pred_w / dw / db of the launch's FbTop): per pass
  const float* flow[FB_WIN_MAX];     // [B,2,H,W]
  const float* g_flow[FB_WIN_MAX];   // [B,2,H,W]
  const uint32_t* z_out[FB_WIN_MAX]; // [B,H,W] the layer's own output spikes
};
// AL (PLIF instantiations): ALIF cells, see FbPlif::xl -- a template parameter, so that the PLIF / XLIF kernels keep their registers
template <bool REC, bool TOP, int EW, bool PLIF = false, bool WIN = false, bool AL = false>
__device__ __forceinline__ void fb_body_ws(
    const int bid, const int nblk_, const float4* __restrict__ g_z_out, const float4* __restrict__ g_z_out2,
    const float4* __restrict__ g_v_out, const float4* __restrict__ v_out, const float4* __restrict__ v_prev,
    const uint32_t* __restrict__ z_prev, const uint32_t* __restrict__ xT, const uint32_t* __restrict__ zT,
    const float* __restrict__ leak, const float* __restrict__ thresh, int B, int H, int W, int nchunk, long nunits, float width,
    int accumulate, int nrows_total, float4* __restrict__ g_cur, uint2* __restrict__ g_split, float4* __restrict__ g_v_prev,
    float* __restrict__ g_leak, float* __restrict__ g_thresh, float* __restrict__ slab_ff, float* __restrict__ slab_rec, FbTop top,
    int row_ld, const FbPlif pl = FbPlif{}, const FbWin* wp = nullptr) {
  static_assert(!PLIF || EW == 8, "PLIF cells: whole-unit stages only");
  static_assert(!AL || PLIF, "ALIF cells run the PLIF body");
  static_assert(!(AL && TOP && !WIN), "ALIF under the prediction head: window launches only (one pass: evf_pred_bwd + the plain cell)");
  static_assert(!WIN || (!REC && EW == 8), "window launches: feed-forward cells");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short* s_b = (unsigned short*)smem_raw;           // [2][3][FB_CW*32] bf16 (region of FB_R0 bytes)
  uint32_t* s_px = (uint32_t*)(smem_raw + FB_R0);             // [2][3][32][FB_NW]
  uint32_t* s_pz = s_px + 2 * 3 * C32 * FB_NW;                // same (REC)
  uint4* s_lut = (uint4*)(s_pz + 2 * 3 * C32 * FB_NW);        // [256]
  float* s_red = (float*)(s_lut + 256);                       // [2][8][32]
  float* s_red2 = s_red + 2 * 8 * C32;                        // [2][8][32] (PLIF: the trace parameters' sums)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NTHR = 64 * (EW + 4);  // threads of the block
  constexpr int ETHR = 64 * EW;        // ... of team E
  const bool team_e = wv < EW;
  int nst = 0;
  (void)nst;
#ifdef FB_STAMPS  // (team E: wave 0, team M: wave 4; before and after every barrier of the unit loop)
#define FBW_STAMP()                                                                                  \
  do {                                                                                               \
    if (blockIdx.x < 16 && lane == 0 && (wv == 0 || wv == EW) && nst < 96)                           \
      fb_stamps[(blockIdx.x * 2 + (wv ? 1 : 0)) * 96 + nst++] = __builtin_readcyclecounter();       \
  } while (0)
#else
#define FBW_STAMP() do {} while (0)
#endif
  const int i = lane & 31, kg = lane >> 5;
  const int et = tid & (ETHR - 1);  // team E: thread within the team
  const int cg = et & 7;            // ... channel group (channels 4cg..4cg+3) ...
  const int pe = et >> 3;           // ... and pixel within the (half) unit
  const int mw = (wv - EW) & 3;     // team M: wave within the team
  const int nW = (W + 31) / 32;
  if (tid < 256) {
    const uint32_t t = tid;
    auto pr = [&](int e) { return ((t >> e) & 1u) * 0x3F80u | (((t >> (e + 1)) & 1u) * 0x3F80u) << 16; };
    s_lut[tid] = make_uint4(pr(0), pr(2), pr(4), pr(6));
  }
  const int nblk = nblk_;
  const int nu = (int)((nunits - (long)bid + nblk - 1) / nblk);
  const int T = WIN ? wp->np : 1;  // passes per unit (WIN)
  const int nit = nu * T;          // iterations of the block: (unit, pass), the passes of a unit back to back
  static_assert(FB_UNITS_MAX <= 64, "geometry table: one lane per unit of the block");
  static_assert(FB_NW == 4 && FB_CW == 64 && (EW == 4 || EW == 8), "team E: (half) units of 8 EW pixels, 384 plane words per unit");
  int g_b, g_y, g_x0;
  {
    const int u = bid + min(lane, nu - 1) * nblk;
    const int row = u / nchunk;
    g_b = row / H;
    g_y = row - g_b * H;
    g_x0 = (u - row * nchunk) * FB_CW;
  }
  auto geom = [&](int k, int& b, int& y, int& x0, int& cw) {
    b = __builtin_amdgcn_readlane(g_b, k);
    y = __builtin_amdgcn_readlane(g_y, k);
    x0 = __builtin_amdgcn_readlane(g_x0, k);
    cw = min(FB_CW, W - x0);
  };
  // operands of the epilogue, requested now (see fb_body): the ninth tap's previous partial sums (all threads), the
  // block's row of per-channel sums (threads < 66 of team E)
  // (the slab tiles' previous partial sums: the accumulators START from them, see team M)
  float row_prev, top_prev = 0.f;
  const size_t row_off = (size_t)bid * row_ld;
  {
    const int row_c = tid & 31, row_which = (tid >> 5) & 1;
    row_prev = (row_which ? g_thresh : g_leak)[row_off + row_c];  // (read by threads < 64)
    if (TOP) top_prev = tid < 64 ? top.dw[row_off + row_which * C32 + row_c] : top.db[row_off + (tid & 1)];  // (threads < 66)
  }
  float rowp_prev = 0.f;  // PLIF: the block's previous sums for leak_pt / add_pt (threads 128 .. 191)
  if (PLIF && tid >= 128 && tid < 192) rowp_prev = ((tid >> 5) & 1 ? pl.g_add_pt : pl.g_leak_pt)[row_off + (tid & 31)];

  // team E state
  float sl[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0};
  float slp[4] = {0, 0, 0, 0}, sap[4] = {0, 0, 0, 0};  // PLIF: sums for leak_pt / add_pt
  float4 gvc = make_float4(0.f, 0.f, 0.f, 0.f), voc = gvc, gkc = gvc;  // WIN: dL/dv, potential, dL/d(pt) carried to the next iteration
  float4 gzxc = gvc;  // WIN, ALIF: (1 - sigma(leak_t)) * dL/d(t') of this pass = part of dL/d(spikes) of the pass before
  float dwa[4] = {0, 0, 0, 0}, dwb[4] = {0, 0, 0, 0}, dba = 0.f, dbb = 0.f;
  // team M state: taps t0 = 2 mw, t1 = 2 mw + 1, the ninth tap's K step mw
  f32x16 acc0 = {0}, acc1 = {0}, accz0 = {0}, accz1 = {0}, acc8 = {0}, accz8 = {0};
  const int t0 = 2 * mw, t1 = 2 * mw + 1;
  const long slab_off0 = (long)bid * (9 * C32 * C32) + t0 * (C32 * C32) + i, slab_off1 = slab_off0 + C32 * C32;

  if (team_e) {
    float lam[4], th[4], oml[4], inv_oml[4];
    float pwa[4] = {0, 0, 0, 0}, pwb[4] = {0, 0, 0, 0};
    float lpt[4] = {0, 0, 0, 0}, apt[4] = {0, 0, 0, 0};  // PLIF: sigma(leak_pt), sigma(add_pt)
    // (computed BEHIND the first units' loads: four exp, four divisions and the head's weights are not needed to request them)
    auto constants = [&]() {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lam[k] = fb_sigmoid(leak[4 * cg + k]);
        th[k] = fmaxf(thresh[4 * cg + k], 0.01f);
        oml[k] = 1.0f - lam[k];
        inv_oml[k] = 1.0f / oml[k];
      }
      if (TOP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pwa[k] = top.pred_w[4 * cg + k], pwb[k] = top.pred_w[C32 + 4 * cg + k];
      }
      if (PLIF) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          lpt[k] = evf_plif_sigmoid(pl.leak_pt[4 * cg + k]),
          apt[k] = pl.xl ? fmaxf(pl.add_pt[4 * cg + k], 0.f) : evf_plif_sigmoid(pl.add_pt[4 * cg + k]);  // (XLIF: t1.clamp_min(0), :365/:810)
      }
    };
    const float4* pgz = g_z_out ? g_z_out : v_out;
    const float4* pgz2 = g_z_out2 ? g_z_out2 : v_out;
    const bool has_gz2 = !TOP && g_z_out2 != nullptr;
    const float4* pgv = g_v_out ? g_v_out : v_out;
    const float4* pvp = v_prev ? v_prev : v_out;
    const uint32_t* pzw = z_prev ? z_prev : xT;
    const uint32_t* pzt = REC ? zT : xT;
    const bool has_gz = g_z_out != nullptr, has_gv = g_v_out != nullptr, has_vp = v_prev != nullptr, has_zw = z_prev != nullptr;
    const float4* pgk = (PLIF && pl.gpt_carry) ? pl.gpt_carry : v_out;  // (optional tensors: a valid dummy, selected afterwards)
    const float4* ppp = (PLIF && pl.pt_prev) ? pl.pt_prev : v_out;
    const bool has_gk = PLIF && pl.gpt_carry != nullptr, has_pp = PLIF && pl.pt_prev != nullptr;
    // stage 1: the global loads of half h of unit k (straight-line, unconditional, clamped: see fb_body)
    auto issue = [&](const int k, const int h, FbStage& s, FbStagePlif& sp) {
      int b, y, x0, cw;
      int ku = k, ks = 0;  // unit of the block, pass (WIN)
      if (WIN) {
        const int kk = min(k, nit - 1);
        ku = kk / T, ks = kk - ku * T;
      }
      geom(min(ku, nu - 1), b, y, x0, cw);
      const long pix0 = ((long)b * H + y) * W + x0;
      const int pc = min(8 * EW * h + pe, cw - 1);
      const long ge = (pix0 + pc) * 8 + cg;
      if (WIN) {  // (per-pass pointers: scalar loads from the argument table; optional tensors from a valid dummy)
        const float4* wvo = wp->vo[ks];
        const float4 *wgz = wp->gz[ks], *wvp = wp->vp[ks], *wpp = wp->pp[ks];
        const uint32_t* wzp = wp->zp[ks];
        if (ks == 0) s.vo = wvo[ge];  // (block-uniform: later passes take the potential the previous iteration loaded as v_prev)
        if (TOP) {
          const long hw = (long)H * W, q = (long)y * W + x0 + pc;
          const float *wf = wp->flow[ks], *wg = wp->g_flow[ks];
          s.f0 = wf[(long)b * 2 * hw + q], s.f1 = wf[((long)b * 2 + 1) * hw + q];
          s.q0 = wg[(long)b * 2 * hw + q], s.q1 = wg[((long)b * 2 + 1) * hw + q];
          s.zo = wp->z_out[ks][pix0 + pc];
        } else
          s.gz = (wgz ? wgz : wvo)[ge];
        s.vp = (wvp ? wvp : wvo)[ge];
        s.zw = (wzp ? wzp : wp->xT[ks])[wzp ? pix0 + pc : 0];
        if (PLIF) {
          sp.pp = (wpp ? wpp : wvo)[ge];
          sp.P = wp->P[ks][pix0 + pc];
        }
        const int tpl = min(et + ETHR * h, 3 * C32 * FB_NW - 1);
        const int pl_wq = tpl % FB_NW, pl_c = (tpl / FB_NW) % C32, pl_dy = tpl / (FB_NW * C32);
        const int yy = y + pl_dy - 1, xw = x0 / 32 - 1 + pl_wq;
        const bool in = yy >= 0 && yy < H && xw >= 0 && xw < nW;
        const long src = in ? (((long)b * H + yy) * C32 + pl_c) * nW + xw : 0;
        s.px = wp->xT[ks][src];
        s.pz = 0u;
        s.pin = in ? 0xFFFFFFFFu : 0u;
        return;
      }
      s.vo = v_out[ge];
      if (TOP) {
        const long hw = (long)H * W, q = (long)y * W + x0 + pc;
        s.f0 = top.flow[(long)b * 2 * hw + q], s.f1 = top.flow[((long)b * 2 + 1) * hw + q];
        s.q0 = top.g_flow[(long)b * 2 * hw + q], s.q1 = top.g_flow[((long)b * 2 + 1) * hw + q];
        s.zo = top.z_out[pix0 + pc];
      } else {
        s.gz = pgz[ge];
        s.gz2 = pgz2[ge];
      }
      s.gv = pgv[ge];
      s.vp = pvp[ge];
      s.zw = pzw[z_prev ? pix0 + pc : 0];
      if (PLIF) {
        sp.gk = pgk[ge];
        sp.pp = ppp[ge];
        sp.P = pl.P[pix0 + pc];
      }
      // this (half) unit's share of the unit's 384 plane words (3 rows x 32 channels x 4 words): EW = 4: 256 + 128
      const int tpl = min(et + ETHR * h, 3 * C32 * FB_NW - 1);
      const int pl_wq = tpl % FB_NW, pl_c = (tpl / FB_NW) % C32, pl_dy = tpl / (FB_NW * C32);
      const int yy = y + pl_dy - 1, xw = x0 / 32 - 1 + pl_wq;
      const bool in = yy >= 0 && yy < H && xw >= 0 && xw < nW;
      const long src = in ? (((long)b * H + yy) * C32 + pl_c) * nW + xw : 0;
      s.px = xT[src];
      s.pz = pzt[src];
      s.pin = in ? 0xFFFFFFFFu : 0u;
    };
    // stage 2: neuron backward in registers, results to HBM, split g_cur + spike planes to LDS buffer `buf`
    auto commit = [&](const int k, const int h, const FbStage& s, const FbStagePlif& sp, const int buf) {
      int b, y, x0, cw;
      int ku = k, ks = 0;
      if (WIN) {
        const int kk = min(k, nit - 1);
        ku = kk / T, ks = kk - ku * T;
      }
      geom(min(ku, nu - 1), b, y, x0, cw);
      const long pix0 = ((long)b * H + y) * W + x0;
      unsigned short* sb = s_b + buf * (3 * FB_CW * C32);
      const int p = 8 * EW * h + pe;
      const bool ok = p < cw && k < nit;
      // WIN: which of the pass's optional tensors exist (block-uniform scalars), where its outputs go
      const bool w_gz = WIN && wp->gz[ks] != nullptr, w_vp = WIN && wp->vp[ks] != nullptr, w_zp = WIN && wp->zp[ks] != nullptr,
                 w_pp = WIN && wp->pp[ks] != nullptr;
      float4* const w_gcur = WIN ? wp->gcur[ks] : nullptr;
      uint2* const w_gsp = WIN ? wp->gsp[ks] : nullptr;
      float* const w_gP = WIN ? wp->gP[ks] : nullptr;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float gp0 = 0.f, gp1 = 0.f;
      if (TOP) {
        gp0 = s.q0 * (1.0f - s.f0 * s.f0);  // tanh' (evf_pred_bwd)
        gp1 = s.q1 * (1.0f - s.f1 * s.f1);
      }
      const float4 gz4 = TOP ? make_float4(gp0 * pwa[0] + gp1 * pwb[0], gp0 * pwa[1] + gp1 * pwb[1], gp0 * pwa[2] + gp1 * pwb[2],
                                           gp0 * pwa[3] + gp1 * pwb[3])
                             : ((WIN ? w_gz : has_gz) ? s.gz : z4);
      float4 gzb4 = z4;
      if (!TOP && !WIN) gzb4 = has_gz2 ? s.gz2 : z4;
      if (AL && WIN) gzb4 = ks > 0 ? gzxc : z4;
      const float4 gv4 = WIN ? (ks > 0 ? gvc : z4) : (has_gv ? s.gv : z4), vp4 = (WIN ? w_vp : has_vp) ? s.vp : z4;
      if (TOP && ok) {
        const uint32_t zo = s.zo >> (4 * cg);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool on = (zo >> c) & 1u;
          dwa[c] += on ? gp0 : 0.f;
          dwb[c] += on ? gp1 : 0.f;
        }
        if (cg == 0) dba += gp0, dbb += gp1;
      }
      const float4 vo4 = (WIN && ks > 0) ? voc : s.vo;
      const float vo[4] = {vo4.x, vo4.y, vo4.z, vo4.w};
      constexpr bool two = !TOP || (AL && WIN);  // dL/d(spikes) has a second part
      const float gz[4] = {two ? gz4.x + gzb4.x : gz4.x, two ? gz4.y + gzb4.y : gz4.y, two ? gz4.z + gzb4.z : gz4.z,
                           two ? gz4.w + gzb4.w : gz4.w};
      const float gvo[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, vp[4] = {vp4.x, vp4.y, vp4.z, vp4.w};
      const uint32_t zw = ((WIN ? w_zp : has_zw) ? s.zw : 0u) >> (4 * cg);
      float gc[4], gp[4], gsv[4], pov[4] = {0.f, 0.f, 0.f, 0.f};
      const bool xl = PLIF && pl.xl != 0;  // (cell-uniform)
      if (PLIF) {  // pt' of the forward pass, recomputed (an XLIF cell: it is part of the threshold, t0 + t1 * pt', :419 / :864)
        const float4 pp4 = (WIN ? w_pp : has_pp) ? sp.pp : z4;
        const float pp[4] = {pp4.x, pp4.y, pp4.z, pp4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) pov[c] = evf_plif_trace(pp[c], lpt[c], AL ? (float)((zw >> c) & 1u) : sp.P);  // (ALIF: t * leak_t + (1 - leak_t) * z, :311)
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {  // autograd of spiking_submodules.py:103-126 / :523-551 (hard reset, arctan surrogate)
        const float z = (float)((zw >> c) & 1u);
        const float sg = fb_surrogate(EVF_ARCTAN, vo[c] - (xl ? th[c] + apt[c] * pov[c] : th[c]), width);
        const float gsp = gz[c] * sg;
        gsv[c] = gsp;
        const float gv = gvo[c] + gsp;
        gc[c] = gv * oml[c];
        gp[c] = gv * lam[c] * (1.0f - z);
        const float cur = (vo[c] - (vp[c] * lam[c]) * (1.0f - z)) * inv_oml[c];
        const float dlam = vp[c] * (1.0f - z) - cur;
        if (ok) {
          sl[c] += gv * dlam;
          st[c] -= gsp;
        }
      }
      const long eo = pix0 * 8 + ETHR * h + et;  // = (pix0 + p) * 8 + cg
      if (WIN) {
        gvc = make_float4(gp[0], gp[1], gp[2], gp[3]);
        voc = vp4;
        if (ok) {
          if (w_gcur) w_gcur[eo] = make_float4(gc[0], gc[1], gc[2], gc[3]);
          if (ks == T - 1 && g_v_prev) g_v_prev[eo] = gvc;  // (the gradient on the state entering the window, when wanted)
        }
      } else if (ok) {
        if (g_cur) g_cur[eo] = make_float4(gc[0], gc[1], gc[2], gc[3]);
        g_v_prev[eo] = make_float4(gp[0], gp[1], gp[2], gp[3]);
      }
      if (PLIF) {  // trace backward (the expressions of k_plif_trace_bwd, evf_network.hip: the same bits per element)
        const float4 gk4 = WIN ? (ks > 0 ? gkc : z4) : (has_gk ? sp.gk : z4), pp4 = (WIN ? w_pp : has_pp) ? sp.pp : z4;
        const float gk[4] = {gk4.x, gk4.y, gk4.z, gk4.w}, pp[4] = {pp4.x, pp4.y, pp4.z, pp4.w};
        const float Pv = sp.P;
        float gq[4], gPp = 0.f, gzx[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float po = pov[c];                // pt' of the forward pass, recomputed
          const float gx = xl ? gsv[c] : gc[c];  // what the trace scaled in the forward pass: the threshold's / the current's gradient (negated)
          const float g = gk[c] - apt[c] * gx;
          gq[c] = g * lpt[c];
          gzx[c] = g * (1.0f - lpt[c]);
          gPp += gzx[c];
          if (ok) {
            slp[c] += g * (pp[c] - (AL ? (float)((zw >> c) & 1u) : Pv));
            sap[c] -= gx * po;
          }
        }
        if constexpr (AL) {  // the trace's drive was the cell's own previous spikes: their gradient, to the pass before
          if (WIN) gzxc = make_float4(gzx[0], gzx[1], gzx[2], gzx[3]);
          else if (ok && pl.g_zx) pl.g_zx[eo] = make_float4(gzx[0], gzx[1], gzx[2], gzx[3]);
        }
        if (WIN) gkc = make_float4(gq[0], gq[1], gq[2], gq[3]);
        if (ok && (!WIN || (ks == T - 1 && pl.g_pt_prev))) pl.g_pt_prev[eo] = make_float4(gq[0], gq[1], gq[2], gq[3]);
        gPp += __shfl_xor(gPp, 1, 64);  // the 8 lanes of a pixel hold its 32 channels
        gPp += __shfl_xor(gPp, 2, 64);
        gPp += __shfl_xor(gPp, 4, 64);
        if (!AL && ok && cg == 0) (WIN ? w_gP : pl.g_P)[pix0 + p] = gPp;  // (ALIF: no pooled activity)
      }
      // exact split g = hi + mid + lo in B-operand order: 16-byte chunk (pixel group G = p >> 3, channel j) = the 8 pixels of
      // the group, chunk index G * 32 + (j ^ (j >> 4)).  A lane holds 4 channels of ONE pixel; as 12 two-byte stores a wave
      // instruction hit 16 distinct words four times over (two lanes per word, channels j and j + 16 on one bank): half of the
      // LDS cycles of the kernel were bank conflicts (PMC), and team M's reads queued behind them.  Instead the lanes of a
      // pixel pair (lane ^ 8) swap half of their channels: the even pixel's lane ends up with channels 4cg, 4cg+1 of both
      // pixels, the odd one's with 4cg+2, 4cg+3 -- 6 four-byte stores per lane, and with the swap of the channel halves
      // 0-15 / 16-31 in the chunk index (j ^ (j >> 4)) the 64 lanes of a store hit 64 different banks.
      uint32_t tp[3][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) evf_split3_pair(ok ? gc[2 * e] : 0.f, ok ? gc[2 * e + 1] : 0.f, tp[0][e], tp[1][e], tp[2][e]);
      {
        uint32_t* sbw = (uint32_t*)sb;
        const bool odd = (p & 1) != 0;
        const int j0 = 4 * cg + (odd ? 2 : 0);  // first of the two channels this lane stores (for pixels p & ~1, p | 1)
        const int d0 = ((p >> 3) * C32 + (j0 ^ (j0 >> 4))) * 4 + ((p & 7) >> 1);
        const int d1 = ((p >> 3) * C32 + ((j0 + 1) ^ (j0 >> 4))) * 4 + ((p & 7) >> 1);
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
          const uint32_t send = odd ? tp[t3][0] : tp[t3][1];
          const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x128, 0xF, 0xF, true);  // row_ror:8 = lane ^ 8
          const uint32_t ev = odd ? recv : tp[t3][0], od = odd ? tp[t3][1] : recv;  // the even / the odd pixel's channel pair
          sbw[t3 * (FB_CW * C32 / 2) + d0] = __builtin_amdgcn_perm(od, ev, 0x05040100u);  // channel j0:     [even px | odd px]
          sbw[t3 * (FB_CW * C32 / 2) + d1] = __builtin_amdgcn_perm(od, ev, 0x07060302u);  // channel j0 + 1
        }
      }
      if (ok && (WIN ? w_gsp != nullptr : g_split != nullptr)) {
        uint2* const gs = WIN ? w_gsp : g_split;
        const long ps = (long)B * H * W * 8;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) gs[t3 * ps + eo] = make_uint2(tp[t3][0], tp[t3][1]);
      }
      const int widx = et + ETHR * h;
      if (widx < 3 * C32 * FB_NW) {
        s_px[buf * (3 * C32 * FB_NW) + widx] = s.px & s.pin;
        if (REC) s_pz[buf * (3 * C32 * FB_NW) + widx] = s.pz & s.pin;
      }
    };
    if constexpr (EW == 4) {
      // half units j = 2k + h through four register stages (stage j % 4): the loads of j + 4 go out as soon as commit(j) has
      // consumed the stage -- four half units (two units) of loads in flight, like fb_body's two units
      FbStage s0, s1, s2, s3;
      FbStagePlif pd;  // (LIF only)
      issue(0, 0, s0, pd);
      issue(0, 1, s1, pd);
      issue(1, 0, s2, pd);
      issue(1, 1, s3, pd);
      constants();
      commit(0, 0, s0, pd, 0);
      issue(2, 0, s0, pd);
      commit(0, 1, s1, pd, 0);
      issue(2, 1, s1, pd);
      FBW_STAMP();
      __syncthreads();  // unit 0 staged
      FBW_STAMP();
#pragma unroll 1
      for (int k = 0; k < nu; k += 2) {
        commit(k + 1, 0, s2, pd, 1);
        issue(k + 3, 0, s2, pd);
        commit(k + 1, 1, s3, pd, 1);
        issue(k + 3, 1, s3, pd);
        FBW_STAMP();
        __syncthreads();  // unit k + 1 staged in buffer 1; team M is done with buffer 0 (unit k)
        FBW_STAMP();
        commit(k + 2, 0, s0, pd, 0);
        issue(k + 4, 0, s0, pd);
        commit(k + 2, 1, s1, pd, 0);
        issue(k + 4, 1, s1, pd);
        FBW_STAMP();
        __syncthreads();
        FBW_STAMP();
      }
    } else {
      // whole units through three register stages that swap roles (fb_body's pipeline): two units of loads in flight
      FbStage s_cur, s_nxt, s_new;
      FbStagePlif p_cur, p_nxt, p_new;
      issue(0, 0, s_cur, p_cur);
      issue(1, 0, s_nxt, p_nxt);
      constants();
      commit(0, 0, s_cur, p_cur, 0);
      FBW_STAMP();
      __syncthreads();  // unit 0 staged
      FBW_STAMP();
#pragma unroll 1
      for (int k = 0; k < nit; k += 2) {
        issue(k + 2, 0, s_new, p_new);
        commit(k + 1, 0, s_nxt, p_nxt, 1);
        FBW_STAMP();
        __syncthreads();  // unit k + 1 staged in buffer 1; team M is done with buffer 0 (unit k)
        FBW_STAMP();
        issue(k + 3, 0, s_nxt, p_nxt);
        commit(k + 2, 0, s_new, p_new, 0);
        FBW_STAMP();
        __syncthreads();
        FBW_STAMP();
      }
    }
  } else {
    // previous partial sums of this wave's two feed-forward slab tiles (consumed in the epilogue)
    // The accumulators START from the block's previous partial sums (an accumulating launch): the loads land while team E
    // stages unit 0, and the epilogue is stores only -- as `previous + sum` it began with an HBM round trip at the end of
    // every block's life.  (Sum order: ((previous + p1) + p2) + ... instead of previous + (p1 + p2 + ...).)  The ninth tap's
    // previous tile goes into wave 0's partial.
    if (accumulate & 1) {
      const long o8 = (long)bid * (9 * C32 * C32) + 8 * (C32 * C32) + i;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc0[q] = slab_ff[slab_off0 + fb_row(q, lane) * C32];
        acc1[q] = slab_ff[slab_off1 + fb_row(q, lane) * C32];
        if (mw == 0) acc8[q] = slab_ff[o8 + fb_row(q, lane) * C32];
        if (REC) {
          accz0[q] = slab_rec[slab_off0 + fb_row(q, lane) * C32];
          accz1[q] = slab_rec[slab_off1 + fb_row(q, lane) * C32];
          if (mw == 0) accz8[q] = slab_rec[o8 + fb_row(q, lane) * C32];
        }
      }
    }
    const int dy0 = t0 / 3, dx0 = t0 % 3, dy1 = t1 / 3, dx1 = t1 % 3;
    const int isw = i ^ (i >> 4);  // chunk of channel i within a pixel group (team E's layout)
    // One wave per SIMD has nobody to hide its LDS latencies behind: the unit is a hand-ordered pipeline (pinned with
    // sched_barrier).  (1) The four plane words of a (row, channel) are 16 bytes: one ds_read_b128 per row gives this lane every
    // 8-pixel byte of the unit for that row (all K steps, all column offsets) -- instead of two ds_read_b32 plus a dependent
    // table lookup in front of every group of MFMAs.  (2) From them the table addresses of all 18 A fragments of the unit.
    // (3) Five steps (four K steps, then the ninth tap's K step): the MFMAs of a step run tap by tap -- per accumulator still
    // hi, mid, lo --, and as soon as a tap's three MFMAs are issued its A registers take the NEXT step's fragment; the B
    // fragments alternate between two register sets, the next set requested in the middle of the step.
    auto mfma_unit = [&](const int buf) {
      const uint4* sbh = (const uint4*)(s_b + buf * (3 * FB_CW * C32));
      const uint32_t* px = s_px + buf * (3 * C32 * FB_NW);
      const uint32_t* pz = s_pz + buf * (3 * C32 * FB_NW);
      // table address (LUT entry) of the byte at bits q .. q + 7 of a row, q = 32 + 16 kq + 8 kg + ddx - 1
      auto abits = [&](uint32_t lo, uint32_t hi, int sh) -> uint32_t {  // ((hi:lo) >> sh) & 0xFF as a table address (x 16 bytes)
        const unsigned long long v = ((unsigned long long)hi << 32) | lo;  // (sh <= 41: the byte lies inside the pair)
        return ((uint32_t)(v >> sh) & 0xFFu) * 16u;
      };
      auto aaddr = [&](const uint4& r, const int ddx, const int kq) -> uint32_t {  // kq: compile-time constant
        const int a = (31 + 16 * kq) >> 5;  // first word that can hold bit q: 0, 1, 1, 2
        return abits(a == 0 ? r.x : (a == 1 ? r.y : r.z), a == 0 ? r.y : (a == 1 ? r.z : r.w), 32 + 16 * kq - 32 * a + 8 * kg + ddx - 1);
      };
      // the ninth tap's K step is the wave's number: its word pair is picked by ADDRESS (a run-time pick among the registers of
      // a row became a store to scratch and a dynamic load back)
      auto aaddr8 = [&](const uint32_t* planes) -> uint32_t {
        const int a = (31 + 16 * mw) >> 5;
        const uint32_t* wr = planes + (2 * C32 + i) * FB_NW + a;
        return abits(wr[0], wr[1], 32 + 16 * mw - 32 * a + 8 * kg + 2 - 1);
      };
      auto lut = [&](uint32_t addr) -> bf16x8 {
        const uint4 a = *(const uint4*)((const char*)s_lut + addr);
        return *(const bf16x8*)&a;
      };
      auto mma3 = [&](f32x16& c, const bf16x8& a, const bf16x8& bh, const bf16x8& bm, const bf16x8& bl) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, c, 0, 0, 0);
      };
      // (two 12-bit table addresses per register: [tap 1 | tap 0] -- 168 registers per wave with EW = 8)
      uint32_t adx[4], adz[4] = {0, 0, 0, 0}, ad8;
      {
        const uint4 rx0 = *(const uint4*)(px + (dy0 * C32 + i) * FB_NW), rx1 = *(const uint4*)(px + (dy1 * C32 + i) * FB_NW);
        uint4 rz0 = rx0, rz1 = rx1;
        if (REC) rz0 = *(const uint4*)(pz + (dy0 * C32 + i) * FB_NW), rz1 = *(const uint4*)(pz + (dy1 * C32 + i) * FB_NW);
        ad8 = aaddr8(px);
        if (REC) ad8 |= aaddr8(pz) << 16;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          adx[kq] = aaddr(rx0, dx0, kq) | aaddr(rx1, dx1, kq) << 16;
          if (REC) adz[kq] = aaddr(rz0, dx0, kq) | aaddr(rz1, dx1, kq) << 16;
          // (pinned: left alone the compiler carries the 64-bit shift results -- two registers per address -- to the use)
          asm volatile("" : "+v"(adx[kq]));
          if (REC) asm volatile("" : "+v"(adz[kq]));
        }
        asm volatile("" : "+v"(ad8));
      }
      auto lo16 = [](uint32_t v) { return v & 0xFFFFu; };
      auto hi16 = [](uint32_t v) { return v >> 16; };
      auto bload = [&](const int kq, bf16x8& bh, bf16x8& bm, bf16x8& bl) {  // this lane's 8 pixels of channel i (= co), K step kq
        const int fo = (kq * 2 + kg) * C32 + isw;
        const uint4 uh = sbh[fo], um = sbh[FB_CW * C32 / 8 + fo], ul = sbh[2 * FB_CW * C32 / 8 + fo];
        bh = *(const bf16x8*)&uh, bm = *(const bf16x8*)&um, bl = *(const bf16x8*)&ul;
      };
      bf16x8 a0 = lut(lo16(adx[0])), a1 = lut(hi16(adx[0])), az0 = a0, az1 = a1;
      if (REC) az0 = lut(lo16(adz[0])), az1 = lut(hi16(adz[0]));
      auto bload1 = [&](const int kq, const int term, bf16x8& b) {
        const uint4 u = sbh[term * (FB_CW * C32 / 8) + (kq * 2 + kg) * C32 + isw];
        b = *(const bf16x8*)&u;
      };
      bf16x8 bAh, bAm, bAl, bBh, bBm, bBl;
      bload(0, bAh, bAm, bAl);
      __builtin_amdgcn_sched_barrier(0);
      // K step kq on the B set (ch, cm, cl); requests the next step's fragments: A into the registers just used; B into the other
      // set (nh, nm, nl) in the middle of the step (EW = 4: 256 registers per wave), or -- one set only, EW = 8: 168 registers --
      // term by term behind the step's last three MFMAs
      auto step = [&](const int kq, bf16x8& ch, bf16x8& cm, bf16x8& cl, bf16x8& nh, bf16x8& nm, bf16x8& nl,
                      const uint32_t nx, const uint32_t nz) {  // nx / nz: packed table addresses of the next step's fragments
        constexpr bool TWO = EW == 4;
        const int nkq = kq < 3 ? kq + 1 : mw;
        mma3(acc0, a0, ch, cm, cl);
        a0 = lut(lo16(nx));  // (after the last K step: the ninth tap's fragment)
        __builtin_amdgcn_sched_barrier(0);
        if (REC || TWO) {
          mma3(acc1, a1, ch, cm, cl);
          if (kq < 3 || REC) a1 = lut(hi16(nx));  // (after the last K step: the ninth tap's recurrent fragment)
          if (TWO) bload(nkq, nh, nm, nl);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (REC) {
          mma3(accz0, az0, ch, cm, cl);
          if (kq < 3) az0 = lut(lo16(nz));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (TWO) {
          if (REC) {
            mma3(accz1, az1, ch, cm, cl);
            if (kq < 3) az1 = lut(hi16(nz));
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {  // the step's last tap, its B registers refilled one by one
          f32x16& c = REC ? accz1 : acc1;
          const bf16x8& al = REC ? az1 : a1;
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ch, c, 0, 0, 0);
          bload1(nkq, 0, ch);
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, cm, c, 0, 0, 0);
          bload1(nkq, 1, cm);
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, cl, c, 0, 0, 0);
          bload1(nkq, 2, cl);
          if (REC) {
            if (kq < 3) az1 = lut(hi16(nz));
          } else if (kq < 3) {
            a1 = lut(hi16(nx));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (EW == 4) {
        step(0, bAh, bAm, bAl, bBh, bBm, bBl, adx[1], adz[1]);
        step(1, bBh, bBm, bBl, bAh, bAm, bAl, adx[2], adz[2]);
        step(2, bAh, bAm, bAl, bBh, bBm, bBl, adx[3], adz[3]);
        step(3, bBh, bBm, bBl, bAh, bAm, bAl, ad8, 0u);
      } else {
        step(0, bAh, bAm, bAl, bAh, bAm, bAl, adx[1], adz[1]);
        step(1, bAh, bAm, bAl, bAh, bAm, bAl, adx[2], adz[2]);
        step(2, bAh, bAm, bAl, bAh, bAm, bAl, adx[3], adz[3]);
        step(3, bAh, bAm, bAl, bAh, bAm, bAl, ad8, 0u);
      }
      // the ninth tap (2, 2): this wave's K step of the unit (fragments in a0 / a1, B in the first set again)
      mma3(acc8, a0, bAh, bAm, bAl);
      if (REC) mma3(accz8, a1, bAh, bAm, bAl);
    };
    FBW_STAMP();
    __syncthreads();  // unit 0 staged
    FBW_STAMP();
#pragma unroll 1
    for (int k = 0; k < nit; k += 2) {
      mfma_unit(0);
      FBW_STAMP();
      __syncthreads();
      FBW_STAMP();
      mfma_unit(1);
      FBW_STAMP();
      __syncthreads();
      FBW_STAMP();
    }
    // ---- this wave's slab tiles (taps t0, t1)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      slab_ff[slab_off0 + fb_row(q, lane) * C32] = acc0[q];
      slab_ff[slab_off1 + fb_row(q, lane) * C32] = acc1[q];
      if (REC) {
        slab_rec[slab_off0 + fb_row(q, lane) * C32] = accz0[q];
        slab_rec[slab_off1 + fb_row(q, lane) * C32] = accz1[q];
      }
    }
  }

  // ---- tap 8: the four partial tiles of team M through LDS (aliases the operand buffers: both teams are past the loop's
  // last barrier), summed and written by all 512 threads
  float* s_t8 = (float*)smem_raw;  // [ff, rec][4][1024] (32 KiB = region 0)
  if (!team_e) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      s_t8[mw * (C32 * C32) + fb_row(q, lane) * C32 + i] = acc8[q];
      if (REC) s_t8[(4 + mw) * (C32 * C32) + fb_row(q, lane) * C32 + i] = accz8[q];
    }
  } else {  // per-channel sums for leak / thresh: lanes with equal (lane & 7) share channels
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        sl[c] += __shfl_xor(sl[c], o, 64);
        st[c] += __shfl_xor(st[c], o, 64);
        if (PLIF) {
          slp[c] += __shfl_xor(slp[c], o, 64);
          sap[c] += __shfl_xor(sap[c], o, 64);
        }
      }
    if (lane < 8) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s_red[(0 * 8 + wv) * C32 + 4 * lane + c] = sl[c];
        s_red[(1 * 8 + wv) * C32 + 4 * lane + c] = st[c];
        if (PLIF) {
          s_red2[(0 * 8 + wv) * C32 + 4 * lane + c] = slp[c];
          s_red2[(1 * 8 + wv) * C32 + 4 * lane + c] = sap[c];
        }
      }
    }
  }
  __syncthreads();  // (one barrier for both: the partial tiles of team M and the per-wave channel sums of team E)
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // C32 * C32 <= 2 * NTHR
    const int e = tid + h * NTHR;
    if (e < C32 * C32) {
      const long o8 = (long)bid * (9 * C32 * C32) + 8 * (C32 * C32) + e;
      slab_ff[o8] = (s_t8[e] + s_t8[C32 * C32 + e]) + (s_t8[2 * C32 * C32 + e] + s_t8[3 * C32 * C32 + e]);
      if (REC) {
        const float* z = s_t8 + 4 * C32 * C32;
        slab_rec[o8] = (z[e] + z[C32 * C32 + e]) + (z[2 * C32 * C32 + e] + z[3 * C32 * C32 + e]);
      }
    }
  }
  if (tid < 64) {
    const int which = tid >> 5, c = tid & 31;
    float v = 0.f;
    for (int w = 0; w < EW; ++w) v += s_red[(which * 8 + w) * C32 + c];
    if (which == 0) {
      const float l = fb_sigmoid(leak[c]), t = v * l * (1.0f - l);
      if (row_ld) g_leak[row_off + c] = row_prev + t;
      else evf_atomic_add(g_leak + c, t);
    } else if (thresh[c] > 0.01f) {
      if (row_ld) g_thresh[row_off + c] = row_prev + v;
      else evf_atomic_add(g_thresh + c, v);
    }
  } else if (PLIF && tid >= 128 && tid < 192) {
    const int which = (tid >> 5) & 1, c = tid & 31;
    float v = 0.f;
    for (int w = 0; w < EW; ++w) v += s_red2[(which * 8 + w) * C32 + c];
    const float sgm = fb_sigmoid(which == 0 ? pl.leak_pt[c] : pl.add_pt[c]);
    float* dst = which == 0 ? pl.g_leak_pt : pl.g_add_pt;
    // chain factor of the raw parameter: sigmoid' -- or, XLIF's t1.clamp_min(0), one where the clamp is inactive
    const float dv = (pl.xl && which == 1) ? (pl.add_pt[c] > 0.f ? v : 0.f) : v * sgm * (1.0f - sgm);
    if (row_ld) dst[row_off + c] = rowp_prev + dv;
    else evf_atomic_add(dst + c, dv);
  }

  // ---- first touch of the slabs by a launch with fewer blocks than slab rows: the other rows start at zero (see fb_body)
  if (!(accumulate & 1) && nblk < nrows_total) {
    for (int r = nblk + bid; r < nrows_total; r += nblk) {
      float4* z0 = (float4*)(slab_ff + (long)r * (9 * C32 * C32));
      float4* z1 = REC ? (float4*)(slab_rec + (long)r * (9 * C32 * C32)) : nullptr;
      for (int e = tid; e < 9 * C32 * C32 / 4; e += NTHR) {
        z0[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (REC) z1[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  if (TOP) {  // prediction-head weight / bias gradients, reduced the same way
    __syncthreads();
    if (team_e) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
          dwa[c] += __shfl_xor(dwa[c], o, 64);
          dwb[c] += __shfl_xor(dwb[c], o, 64);
        }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        dba += __shfl_xor(dba, o, 64);
        dbb += __shfl_xor(dbb, o, 64);
      }
      if (lane < 8) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s_red[(0 * 8 + wv) * C32 + 4 * lane + c] = dwa[c];
          s_red[(1 * 8 + wv) * C32 + 4 * lane + c] = dwb[c];
        }
      }
    }
    float* s_b2 = (float*)smem_raw;  // operand buffers are free now
    if (team_e && lane == 0) s_b2[2 * wv] = dba, s_b2[2 * wv + 1] = dbb;
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float v = 0.f;
      for (int w = 0; w < EW; ++w) v += s_red[(which * 8 + w) * C32 + c];
      if (row_ld) top.dw[row_off + which * C32 + c] = top_prev + v;
      else evf_atomic_add(top.dw + which * C32 + c, v);
    } else if (tid < 66) {
      float v = 0.f;
      for (int w = 0; w < EW; ++w) v += s_b2[2 * w + (tid - 64)];
      if (row_ld) top.db[row_off + (tid - 64)] = top_prev + v;
      else evf_atomic_add(top.db + (tid - 64), v);
    }
  }
  FBW_STAMP();
}

template <bool REC, bool TOP, bool FAST>
__global__ __launch_bounds__(FB_THREADS) void k_lif_bwd_wgrad(
    const float4* __restrict__ g_z_out, const float4* __restrict__ g_z_out2, const float4* __restrict__ g_v_out,
    const float4* __restrict__ v_out, const float4* __restrict__ v_prev, const uint32_t* __restrict__ z_prev,
    const uint32_t* __restrict__ xT, const uint32_t* __restrict__ zT, const float* __restrict__ leak,
    const float* __restrict__ thresh, int B, int H, int W, int nchunk, long nunits, int hard_reset_rt, int surrogate_rt,
    float width, int accumulate, int nrows_total, float4* __restrict__ g_cur, uint2* __restrict__ g_split,
    float4* __restrict__ g_v_prev, float* __restrict__ g_leak, float* __restrict__ g_thresh, float* __restrict__ slab_ff,
    float* __restrict__ slab_rec, FbTop top, int row_ld) {
  fb_body<REC, TOP, FAST>(blockIdx.x, gridDim.x, g_z_out, g_z_out2, g_v_out, v_out, v_prev, z_prev, xT, zT, leak, thresh, B, H, W,
                          nchunk, nunits, hard_reset_rt, surrogate_rt, width, accumulate, nrows_total, g_cur, g_split, g_v_prev, g_leak,
                          g_thresh, slab_ff, slab_rec, top, row_ld);
}

// ---- several independent backward cells of a window in ONE launch (see k_fwd_diag in evf_fwd_b3.hip) --------------------
// The backward of a pass is a chain of 13 steps (fused backward of layer 6, its input gradient, layer 5, ... , head); step
// s of pass t needs step s - 1 of pass t and step s (+1 for the recurrent input gradient) of pass t + 1: with the index
// 2 (P - 1 - t) + s all cells under one index are independent AND of one kind (s even: fused backward, s odd: input
// gradient).  evf_bwd_defer_* record the cells; the flush launches index after index: one k_bwd_diag (this file) or one
// k_dgrad_diag (evf_dgrad_b3.hip) per index, the head layer's cells as their own launches in between.  Default neuron only
// (the FAST bodies).  blockIdx.x = cell * nblk + block.
#define FB_MAX_JOBS 8
struct FbJob {
  const float4 *g_z, *g_z2, *g_v, *v_out, *v_prev;
  const uint32_t *z_prev, *xT, *zT;
  const float *leak, *thresh;
  float4* g_cur;   // fp32 dL/d(current) (NULL when only the split planes are wanted)
  uint2* g_split;  // its exact 3-way bf16 split as three planes [term][pix][32] (NULL: not written) -- what k_dgrad_diag_dma reads
  float4* g_v_prev;
  float *g_leak, *g_thresh, *slab_ff, *slab_rec;
  FbTop top;
  float width;
  int accumulate;
  int kind;  // 0 feed-forward, 1 recurrent, 2 under the prediction head; + 3: the same as PLIF cells (k_bwd_diag_ws_plif)
  int blk0;  // first block of the cell in a launch of several (INT_MAX: unused entry)
  int nblk;  // its number of blocks
  int pad_;
  FbPlif pl;  // (kind >= 3)
};

struct FbJobs {
  FbJob j[FB_MAX_JOBS];
};
// the cell of a block: cells own consecutive block ranges of DIFFERENT lengths (fb_split_blocks)
__device__ __forceinline__ int fb_job_of_block(const FbJobs& jobs, int blk) {
  int jb = 0;
#pragma unroll
  for (int k = 1; k < FB_MAX_JOBS; ++k) jb += blk >= jobs.j[k].blk0 ? 1 : 0;
  return jb;
}
__global__ __launch_bounds__(FB_THREADS) void k_bwd_diag(FbJobs jobs, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                         int nrows_total) {
  const int jb = fb_job_of_block(jobs, (int)blockIdx.x);
  const FbJob& J = jobs.j[jb];
  const int bid = (int)blockIdx.x - J.blk0, nblk = J.nblk;
  if (J.kind == 1)
    fb_body<true, false, true>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                               nchunk, nunits, 1, EVF_ARCTAN, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak,
                               J.g_thresh, J.slab_ff, J.slab_rec, J.top, row_ld);
  else if (J.kind == 2)
    fb_body<false, true, true>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                               nchunk, nunits, 1, EVF_ARCTAN, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak,
                               J.g_thresh, J.slab_ff, J.slab_rec, J.top, row_ld);
  else
    fb_body<false, false, true>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                                nchunk, nunits, 1, EVF_ARCTAN, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak,
                                J.g_thresh, J.slab_ff, J.slab_rec, J.top, row_ld);
}

template <int EW>
__global__ __launch_bounds__(64 * (EW + 4)) void k_bwd_diag_ws(FbJobs jobs, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                             int nrows_total) {
  const int jb = fb_job_of_block(jobs, (int)blockIdx.x);
  const FbJob& J = jobs.j[jb];
  const int bid = (int)blockIdx.x - J.blk0, nblk = J.nblk;
  if (J.kind == 1)
    fb_body_ws<true, false, EW>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                                nchunk, nunits, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak, J.g_thresh,
                                J.slab_ff, J.slab_rec, J.top, row_ld);
  else if (J.kind == 2)
    fb_body_ws<false, true, EW>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                                nchunk, nunits, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak, J.g_thresh,
                                J.slab_ff, J.slab_rec, J.top, row_ld);
  else
    fb_body_ws<false, false, EW>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                                 nchunk, nunits, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak, J.g_thresh,
                                 J.slab_ff, J.slab_rec, J.top, row_ld);
}

// PLIF cells (kind 3 .. 5): team E carries the trace backward as well; a kernel of its own, so that the LIF kernel's register
// allocation (167 of the 168 a 768-thread block may have) stays what it is
__global__ __launch_bounds__(768) void k_bwd_diag_ws_plif(FbJobs jobs, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                          int nrows_total) {
  const int jb = fb_job_of_block(jobs, (int)blockIdx.x);
  const FbJob& J = jobs.j[jb];
  const int bid = (int)blockIdx.x - J.blk0, nblk = J.nblk;
  if (J.kind == 4)
    fb_body_ws<true, false, 8, true>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                                     nchunk, nunits, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak,
                                     J.g_thresh, J.slab_ff, J.slab_rec, J.top, row_ld, J.pl);
  else if (J.kind == 5)
    fb_body_ws<false, true, 8, true>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                                     nchunk, nunits, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak,
                                     J.g_thresh, J.slab_ff, J.slab_rec, J.top, row_ld, J.pl);
  else
    fb_body_ws<false, false, 8, true>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh, B, H, W,
                                      nchunk, nunits, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev, J.g_leak,
                                      J.g_thresh, J.slab_ff, J.slab_rec, J.top, row_ld, J.pl);
}

// ALIF cells (FbPlif::xl == 2): the same three launches with the trace driven by the cell's own previous spikes (fb_body_ws<.., AL>)
__global__ __launch_bounds__(768) void k_bwd_diag_ws_alif(FbJobs jobs, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                          int nrows_total) {
  const int jb = fb_job_of_block(jobs, (int)blockIdx.x);
  const FbJob& J = jobs.j[jb];
  const int bid = (int)blockIdx.x - J.blk0, nblk = J.nblk;
  if (J.kind == 4)
    fb_body_ws<true, false, 8, true, false, true>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh,
                                                  B, H, W, nchunk, nunits, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev,
                                                  J.g_leak, J.g_thresh, J.slab_ff, J.slab_rec, J.top, row_ld, J.pl);
  else
    fb_body_ws<false, false, 8, true, false, true>(bid, nblk, J.g_z, J.g_z2, J.g_v, J.v_out, J.v_prev, J.z_prev, J.xT, J.zT, J.leak, J.thresh,
                                                   B, H, W, nchunk, nunits, J.width, J.accumulate, nrows_total, J.g_cur, J.g_split, J.g_v_prev,
                                                   J.g_leak, J.g_thresh, J.slab_ff, J.slab_rec, J.top, row_ld, J.pl);
}
__global__ __launch_bounds__(768) void k_bwd_win_alif(FbJob J, FbWin Wn, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                      int nrows_total) {
  fb_body_ws<false, false, 8, true, true, true>((int)blockIdx.x, (int)gridDim.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                nullptr, J.leak, J.thresh, B, H, W, nchunk, nunits, J.width, J.accumulate, nrows_total, nullptr,
                                                nullptr, J.g_v_prev, J.g_leak, J.g_thresh, J.slab_ff, nullptr, J.top, row_ld, J.pl, &Wn);
}
__global__ __launch_bounds__(768) void k_bwd_win_alif_top(FbJob J, FbWin Wn, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                          int nrows_total) {
  fb_body_ws<false, true, 8, true, true, true>((int)blockIdx.x, (int)gridDim.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                               nullptr, J.leak, J.thresh, B, H, W, nchunk, nunits, J.width, J.accumulate, nrows_total, nullptr,
                                               nullptr, J.g_v_prev, J.g_leak, J.g_thresh, J.slab_ff, nullptr, J.top, row_ld, J.pl, &Wn);
}

// All passes of a window of ONE feed-forward PLIF cell (fb_body_ws<.., WIN>): a launch of its own
__global__ __launch_bounds__(768) void k_bwd_win_plif(FbJob J, FbWin Wn, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                      int nrows_total) {
  fb_body_ws<false, false, 8, true, true>((int)blockIdx.x, (int)gridDim.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                          nullptr, J.leak, J.thresh, B, H, W, nchunk, nunits, J.width, J.accumulate, nrows_total, nullptr,
                                          nullptr, J.g_v_prev, J.g_leak, J.g_thresh, J.slab_ff, nullptr, J.top, row_ld, J.pl, &Wn);
}

// ... of the layer under the prediction head (the head's backward inside, per pass)
__global__ __launch_bounds__(768) void k_bwd_win_plif_top(FbJob J, FbWin Wn, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                          int nrows_total) {
  fb_body_ws<false, true, 8, true, true>((int)blockIdx.x, (int)gridDim.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                         nullptr, J.leak, J.thresh, B, H, W, nchunk, nunits, J.width, J.accumulate, nrows_total, nullptr,
                                         nullptr, J.g_v_prev, J.g_leak, J.g_thresh, J.slab_ff, nullptr, J.top, row_ld, J.pl, &Wn);
}

// ... LIF cells (the feed-forward layers on top of a LIF-FireNet: their dL/d(spikes) of every pass is known before anything
// below them has run)
__global__ __launch_bounds__(768) void k_bwd_win_lif(FbJob J, FbWin Wn, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                     int nrows_total) {
  fb_body_ws<false, false, 8, false, true>((int)blockIdx.x, (int)gridDim.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                           nullptr, J.leak, J.thresh, B, H, W, nchunk, nunits, J.width, J.accumulate, nrows_total, nullptr,
                                           nullptr, J.g_v_prev, J.g_leak, J.g_thresh, J.slab_ff, nullptr, J.top, row_ld, J.pl, &Wn);
}
__global__ __launch_bounds__(768) void k_bwd_win_lif_top(FbJob J, FbWin Wn, int B, int H, int W, int nchunk, long nunits, int row_ld,
                                                         int nrows_total) {
  fb_body_ws<false, true, 8, false, true>((int)blockIdx.x, (int)gridDim.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                          nullptr, J.leak, J.thresh, B, H, W, nchunk, nunits, J.width, J.accumulate, nrows_total, nullptr,
                                          nullptr, J.g_v_prev, J.g_leak, J.g_thresh, J.slab_ff, nullptr, J.top, row_ld, J.pl, &Wn);
}

static long fb_units(int B, int H, int W) { return (long)B * H * ((W + FB_CW - 1) / FB_CW); }
#define FB_LDS (FB_R0 + 2 * (2 * 3 * C32 * FB_NW * 4) + 256 * 16 + 2 * (2 * 8 * C32 * 4))  // (the second [2][8][32]: PLIF)

// Slab rows per (cell, input) = the most blocks a launch may give a cell.  One row per FB_UNITS units up to one block per CU's
// worth (256); beyond that a launch never takes more blocks than that anyway (fb_blocks_per_cell: 260 x 346 x B4 = 6240 units runs
// on 250 blocks) -- with 780 rows the other 530 were zero-filled by every first-touch launch and read back by the reduction
// (k_grads_finalize: 115 us per step).  At least cdiv(units, FB_UNITS_MAX): a block holds at most 64 units.
static int fb_rows(long nunits) {
  const long full = (nunits + FB_UNITS - 1) / FB_UNITS, need = (nunits + FB_UNITS_MAX - 1) / FB_UNITS_MAX;
  const long cap = need > 256 ? need : 256;
  return (int)(full < cap ? full : cap);
}
extern "C" int evf_lif_bwd_wgrad_slabs(int B, int H, int W) { return fb_rows(fb_units(B, H, W)); }

EvfBwdDefer evf_bwd_defer_tab[EVF_CTX_MAX] = {};

// recording contexts (evf_common.h): stream -> index, reference counted by the open recordings
static struct {
  std::mutex mu;
  void* stream[EVF_CTX_MAX];
  int refs[EVF_CTX_MAX];
} evf_ctx = {};
int evf_ctx_find(void* stream) {
  std::lock_guard<std::mutex> g(evf_ctx.mu);
  for (int c = 0; c < EVF_CTX_MAX; ++c)
    if (evf_ctx.refs[c] > 0 && evf_ctx.stream[c] == stream) return c;
  return -1;
}
int evf_ctx_acquire(void* stream) {
  std::lock_guard<std::mutex> g(evf_ctx.mu);
  int free_c = -1;
  for (int c = 0; c < EVF_CTX_MAX; ++c) {
    if (evf_ctx.refs[c] > 0 && evf_ctx.stream[c] == stream) {
      ++evf_ctx.refs[c];
      return c;
    }
    if (evf_ctx.refs[c] == 0 && free_c < 0) free_c = c;
  }
  if (free_c >= 0) evf_ctx.stream[free_c] = stream, evf_ctx.refs[free_c] = 1;
  return free_c;
}
void evf_ctx_drop(int ctx) {
  std::lock_guard<std::mutex> g(evf_ctx.mu);
  if (ctx >= 0 && ctx < EVF_CTX_MAX && evf_ctx.refs[ctx] > 0) --evf_ctx.refs[ctx];
}

#define EVF_PROF_MAX 1024
#define EVF_PROF_STAMPS 256  // brackets of a captured step (mode 2)
static struct {
  int mode;  // 0 off, 1 eager brackets (HIP events), 2 brackets CAPTURED into a hipGraph as timestamp kernels (read after replays)
  int n;
  int kind[EVF_PROF_MAX];
  hipEvent_t e0[EVF_PROF_MAX], e1[EVF_PROF_MAX];
  int made;
  unsigned long long* stamps;  // device [EVF_PROF_STAMPS][2]: wall-clock reading before / after the bracketed launch
  bool stamped;                // the recorded brackets are timestamp pairs
} evf_prof = {0, 0, {0}, {}, {}, 0, nullptr, false};
// One thread reads the constant-rate wall clock (s_memrealtime; hipDeviceAttributeWallClockRate kHz).  As ordinary kernel
// nodes these are what a capture can carry on every runtime (external event-record nodes inside torch's captures were refused
// with hipErrorInvalidValue on ROCm 7.2); each replay overwrites the stamps.
__global__ void k_prof_stamp(unsigned long long* dst) { *dst = wall_clock64(); }
void evf_prof_mark(int kind, int end, void* stream) {
  if (!evf_prof.mode || (!end && evf_prof.n >= EVF_PROF_MAX)) return;
  if (evf_prof.mode == 2) {
    if (!evf_prof.stamps || evf_prof.n >= EVF_PROF_STAMPS) return;
    if (!end) evf_prof.kind[evf_prof.n] = kind;
    hipLaunchKernelGGL(k_prof_stamp, dim3(1), dim3(1), 0, EVF_STREAM(stream), evf_prof.stamps + 2 * evf_prof.n + (end ? 1 : 0));
    if (end) ++evf_prof.n;
    return;
  }
  if (!end) {
    if (evf_prof.n >= evf_prof.made) {
      (void)hipEventCreate(&evf_prof.e0[evf_prof.made]);
      (void)hipEventCreate(&evf_prof.e1[evf_prof.made]);
      ++evf_prof.made;
    }
    evf_prof.kind[evf_prof.n] = kind;
    (void)hipEventRecord(evf_prof.e0[evf_prof.n], EVF_STREAM(stream));
  } else if (evf_prof.n < EVF_PROF_MAX) {
    (void)hipEventRecord(evf_prof.e1[evf_prof.n], EVF_STREAM(stream));
    ++evf_prof.n;
  }
}
// evf_defer_profile(1): time every dispatcher launch of the following flushes; evf_defer_profile_read: device sync, then
// ms[k] = summed duration and count[k] = number of launches of kind k < 16 (0 k_fwd_diag, 1 k_bwd_diag, 2 k_dgrad_diag, 3 head
// backward pass by pass, 4 k_head_lif_fwd_win, 5 k_head_bwd_win, 6 k_fwd_win_t, 7 an EMPTY bracket, 8 k_bwd_win_* (a hidden layer's backward of a window in one launch), 9 evf_conv_dgrad_b3_multi = the bracket's own cost) since it was
// switched on (event-bracket overhead included: ~1.6 us per launch); switches it off.
// evf_defer_profile(2): the same brackets while the step is CAPTURED into a hipGraph -- a one-thread timestamp kernel in front
// of and behind every dispatcher launch (and one empty bracket per forward flush, kind 7); evf_defer_profile(0) after the
// capture stops the recording and KEEPS the brackets, the graph is replayed, and evf_defer_profile_read returns the durations
// of the LAST replay: the kernels as they run back to back inside the replayed step (clocks, caches), not as eager launches.
// A bracket there = the launch + one inter-kernel gap + the empty bracket (kind 7).
extern "C" int evf_defer_profile(int on) {
  if (on == 0) {  // stop; what is recorded stays readable (mode 2: after the replays)
    evf_prof.mode = 0;
    return EVF_OK;
  }
  evf_prof.mode = on == 2 ? 2 : 1;
  evf_prof.n = 0;
  evf_prof.stamped = on == 2;
  if (on == 2 && !evf_prof.stamps) {  // (nothing may be allocated once the capture is open)
    const int rc = evf_hip(hipMalloc((void**)&evf_prof.stamps, sizeof(unsigned long long) * 2 * EVF_PROF_STAMPS));
    if (rc) {
      evf_prof.stamps = nullptr, evf_prof.mode = 0;
      return rc;
    }
  }
  return EVF_OK;
}
extern "C" int evf_defer_profile_read(float* ms, int* count) {
  if (!ms || !count) return EVF_EINVAL;
  evf_prof.mode = 0;
  { const int rc = evf_hip(hipDeviceSynchronize()); if (rc) return rc; }
  for (int k = 0; k < 16; ++k) ms[k] = 0.f, count[k] = 0;
  if (evf_prof.stamped) {
    static unsigned long long host[2 * EVF_PROF_STAMPS];
    int dev = 0, khz = 0;
    if (evf_prof.n > 0) {
      const int rc = evf_hip(hipMemcpy(host, evf_prof.stamps, sizeof(unsigned long long) * 2 * evf_prof.n, hipMemcpyDeviceToHost));
      if (rc) return rc;
    }
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
      khz = 100000;  // 100 MHz
    for (int i = 0; i < evf_prof.n; ++i) {
      ms[evf_prof.kind[i] & 15] += (float)((double)(host[2 * i + 1] - host[2 * i]) / (double)khz);
      ++count[evf_prof.kind[i] & 15];
    }
    evf_prof.n = 0, evf_prof.stamped = false;
    return EVF_OK;
  }
  for (int i = 0; i < evf_prof.n; ++i) {
    float t = 0.f;
    const int rc = evf_hip(hipEventElapsedTime(&t, evf_prof.e0[i], evf_prof.e1[i]));
    if (rc) {
      (void)hipGetLastError();  // (a failed measurement must not poison the next launch's status)
      evf_prof.n = 0;
      return rc;
    }
    ms[evf_prof.kind[i] & 15] += t;
    ++count[evf_prof.kind[i] & 15];
  }
  evf_prof.n = 0;
  return EVF_OK;
}
int evf_prof_mode() { return evf_prof.mode; }
struct FbDefer {
  int B, H, W, row_ld;
  int n[EVF_BWD_DIAGS];
  FbJob job[EVF_BWD_DIAGS][FB_MAX_JOBS];
};
static FbDefer fb_tab[EVF_CTX_MAX];

// Blocks per cell: a block takes u = 8..64 units (the slab rows a launch does not write are zero-filled on first touch, see
// fb_body), and a launch of n cells runs in whole rounds of one block per CU -- a block costs ~21 k cycles of prologue +
// epilogue and ~5.5 k per unit (phase stamps, 128 x 128 x B8).  The u with the least rounds x (21 + 5.5 u): one cell of
// 128 x 128 x B8 (2048 units): u = 8, 256 blocks; three cells: u = 25, 3 x 82 blocks = ONE round; six cells: u = 49, 6 x 42 blocks;
// one cell of 260 x 346 x B4 (6240 units): u = 25, 250 blocks instead of 780 = 4 rounds.  EVF_BWD_UNITS=8..64 fixes u.
static int fb_blocks_per_cell(long nunits, int n, int per_unit = 11) {  // per_unit: cycles per unit / 500 (11 fused, 8 two teams)
  static const int mode = []() {
    const char* e = getenv("EVF_BWD_UNITS");
    const int v = e ? atoi(e) : 0;
    return (v >= FB_UNITS && v <= FB_UNITS_MAX) ? v : 0;
  }();
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  const int rows = fb_rows(nunits);
  if (mode) return evf_cdiv(nunits, mode) < rows ? evf_cdiv(nunits, mode) : rows;
  long best = -1;
  int best_nb = rows;
  for (int u = FB_UNITS; u <= FB_UNITS_MAX; ++u) {
    const int nb = evf_cdiv(nunits, u);
    if (nb > rows) continue;
    const long cost = (long)evf_cdiv((long)n * nb, ncu) * (42 + per_unit * u);  // (x2: integers)
    if (best < 0 || cost < best) best = cost, best_nb = nb;
  }
  return best_nb;
}

// The n cells of a launch share n * nblk blocks IN PROPORTION TO WHAT A UNIT OF THEIR KIND COSTS: a recurrent cell contracts two
// weight gradients per unit (its matrix team is the slower team), the cell under the prediction head carries the head's backward;
// with equal block counts their blocks were the tail of every launch.  Weights per kind (feed-forward, recurrent, top) from
// EVF_BWD_W=a,b,c (measurements); 0,0,0 = equal shares.  A block takes 8..64 units (fb_body*: one lane per unit in the geometry
// table, one slab row per block).  Sets blk0 / nblk of every entry; returns the number of blocks of the launch.
static int fb_split_blocks(FbJobs& jobs, int n, int nblk, long nunits, bool teams8) {
  static int w[3] = {-1, 0, 0};
  if (w[0] < 0) {
    w[0] = 10, w[1] = 13, w[2] = 11;
    const char* e = getenv("EVF_BWD_W");
    int a = 0, b = 0, c = 0;
    if (e && sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a >= 0 && b >= 0 && c >= 0 && a < 256 && b < 256 && c < 256) w[0] = a, w[1] = b, w[2] = c;
  }
  const int lo = evf_cdiv(nunits, FB_UNITS_MAX), hi = fb_rows(nunits);
  int nb[FB_MAX_JOBS], wj[FB_MAX_JOBS];
  long ws = 0;
  for (int k = 0; k < n; ++k) wj[k] = (teams8 && w[0] > 0 && w[1] > 0 && w[2] > 0) ? w[jobs.j[k].kind % 3] : 1, ws += wj[k];
  const long tot = (long)nblk * n;
  long used = 0;
  for (int k = 0; k < n; ++k) {
    long v = tot * wj[k] / ws;
    nb[k] = (int)(v < lo ? lo : (v > hi ? hi : v));
    used += nb[k];
  }
  // left-over blocks one at a time to the cell whose blocks are the slowest; too many (clamping): from the fastest
  auto cost = [&](int k, int b) { return (long)wj[k] * evf_cdiv(nunits, b); };
  while (used < tot) {
    int best = -1;
    for (int k = 0; k < n; ++k)
      if (nb[k] < hi && (best < 0 || cost(k, nb[k]) > cost(best, nb[best]))) best = k;
    if (best < 0) break;
    ++nb[best], ++used;
  }
  while (used > tot) {
    int best = -1;
    for (int k = 0; k < n; ++k)
      if (nb[k] > lo && (best < 0 || cost(k, nb[k] - 1) < cost(best, nb[best] - 1))) best = k;
    if (best < 0) break;
    --nb[best], --used;
  }
  int b0 = 0;
  for (int k = 0; k < FB_MAX_JOBS; ++k) {
    if (k < n) jobs.j[k].blk0 = b0, jobs.j[k].nblk = nb[k], b0 += nb[k];
    else jobs.j[k].blk0 = 0x7fffffff, jobs.j[k].nblk = 1;
  }
  return b0;
}

static int fb_diag_select = -1;  // -1 environment / default, 0 k_bwd_diag, 1 k_bwd_diag_ws<4>, 2 k_bwd_diag_ws<8>
extern "C" int evf_bwd_diag_select(int which) {
  if (which < -1 || which > 2) return EVF_EINVAL;
  fb_diag_select = which;
  return EVF_OK;
}

static int fb_defer_launch(FbDefer& fb_defer, int d, void* stream) {
  const int n = fb_defer.n[d];
  if (!n) return EVF_OK;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_bwd_diag, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_diag_ws<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_diag_ws<8>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_diag_ws_plif, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_diag_ws_alif, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    attr_set = true;
  }
  // EVF_BWD_DIAG=fused: every wave through all phases (k_bwd_diag); teams4: 4 + 4 waves; default (teams): 8 + 4 waves
  static const int teams_env = []() {
    const char* e = getenv("EVF_BWD_DIAG");
    if (e && e[0] == 'f') return 0;
    return (e && e[0] == 't' && e[5] == '4') ? 1 : 2;
  }();
  const bool plif = fb_defer.job[d][0].kind >= 3;  // (a recording holds cells of one neuron model: fb_launch)
  const int teams = plif ? 2 : (fb_diag_select < 0 ? teams_env : fb_diag_select);
  FbJobs jobs;
  for (int k = 0; k < FB_MAX_JOBS; ++k) jobs.j[k] = fb_defer.job[d][k < n ? k : 0];
  const long nunits = fb_units(fb_defer.B, fb_defer.H, fb_defer.W);
  const int nrows = fb_rows(nunits), nchunk = (fb_defer.W + FB_CW - 1) / FB_CW;
  static const int cost_env = []() { const char* e = getenv("EVF_BWD_COST"); return e ? atoi(e) : 0; }();  // (A/B measurements)
  const int nblk = fb_blocks_per_cell(nunits, n, cost_env > 0 ? cost_env : (teams == 2 ? 8 : 11));  // (k_bwd_diag_ws<8>: ~4.0 k cycles per unit, phase stamps)
  const int ntot = fb_split_blocks(jobs, n, nblk, nunits, teams == 2);
  evf_prof_mark(1, 0, stream);
  if (plif && jobs.j[0].pl.xl == 2)
    hipLaunchKernelGGL(k_bwd_diag_ws_alif, dim3(ntot), dim3(768), FB_LDS, EVF_STREAM(stream), jobs, fb_defer.B, fb_defer.H,
                       fb_defer.W, nchunk, nunits, fb_defer.row_ld, nrows);
  else if (plif)
    hipLaunchKernelGGL(k_bwd_diag_ws_plif, dim3(ntot), dim3(768), FB_LDS, EVF_STREAM(stream), jobs, fb_defer.B, fb_defer.H,
                       fb_defer.W, nchunk, nunits, fb_defer.row_ld, nrows);
  else if (teams == 2)
    hipLaunchKernelGGL(k_bwd_diag_ws<8>, dim3(ntot), dim3(768), FB_LDS, EVF_STREAM(stream), jobs, fb_defer.B, fb_defer.H,
                       fb_defer.W, nchunk, nunits, fb_defer.row_ld, nrows);
  else if (teams == 1)
    hipLaunchKernelGGL(k_bwd_diag_ws<4>, dim3(ntot), dim3(512), FB_LDS, EVF_STREAM(stream), jobs, fb_defer.B, fb_defer.H,
                       fb_defer.W, nchunk, nunits, fb_defer.row_ld, nrows);
  else
    hipLaunchKernelGGL(k_bwd_diag, dim3(ntot), dim3(FB_THREADS), FB_LDS, EVF_STREAM(stream), jobs, fb_defer.B, fb_defer.H,
                       fb_defer.W, nchunk, nunits, fb_defer.row_ld, nrows);
  evf_prof_mark(1, 1, stream);
  fb_defer.n[d] = 0;
  return evf_status();
}

// (Round 3, measured and dropped: the head layer's backward cells on a side stream forked inside this loop -- they form a chain
//  of their own, but the next input-gradient launch rewrites the one buffer they read (dL/d(spikes) of the head layer), so each
//  can only hide under ONE fused-backward launch; as parallel branches of the replayed graph: 4.21 against 4.20 ms per step.)
// (Later in round 3: with one dL/d(spikes) buffer PER PASS the head cells have no tie to the indices at all and run after
//  the last one, all passes in one launch: evf_hd_defer_launch_window, evf_network.hip.)
// final = false (a call that cannot be recorded runs everything recorded first) with evf_bwd_defer_hold_heads on: the head layer's
// cells stay recorded -- the caller has promised that nothing recorded or launched later touches what they read or write (one
// dL/d(spikes) buffer per pass), so they run at the recording's end, all passes in one launch.  PLIF networks: their hidden cells'
// input gradients are not recordable (the pooling's adjoint), i.e. every pass flushes.
int evf_bwd_defer_flush_now(int ctx, void* stream, bool final) {
  const bool hold = !final && evf_bwd_defer_tab[ctx].hold_heads;
  const bool heads_last = hold || evf_hd_defer_window_ok(ctx) != 0;
  for (int d = 0; d < EVF_BWD_DIAGS; ++d) {
    int rc = fb_defer_launch(fb_tab[ctx], d, stream);
    if (!rc) rc = evf_dg_defer_launch(ctx, d, stream);
    if (!rc && !heads_last) rc = evf_hd_defer_launch(ctx, d, stream);
    if (rc) return rc;
  }
  if (hold) return EVF_OK;
  return heads_last ? evf_hd_defer_launch_window(ctx, stream) : EVF_OK;
}

extern "C" int evf_bwd_defer_begin(void* stream) {
  const int c = evf_ctx_find(stream);
  if (c >= 0 && evf_bwd_defer_tab[c].active) return EVF_EINVAL;  // one backward recording per stream
  const int ctx = evf_ctx_acquire(stream);
  if (ctx < 0) return EVF_EINVAL;
  evf_bwd_defer_tab[ctx].active = true;
  evf_bwd_defer_tab[ctx].slot = 0;
  evf_bwd_defer_tab[ctx].hold_heads = false;
  return EVF_OK;
}
extern "C" int evf_bwd_defer_hold_heads(int on, void* stream) {
  const int c = evf_ctx_find(stream);
  if (c < 0 || !evf_bwd_defer_tab[c].active) return EVF_EINVAL;
  evf_bwd_defer_tab[c].hold_heads = on != 0;
  return EVF_OK;
}
extern "C" int evf_bwd_defer_slot(int d, void* stream) {
  const int c = evf_ctx_find(stream);
  if (c < 0 || !evf_bwd_defer_tab[c].active || d < 0 || d >= EVF_BWD_DIAGS) return EVF_EINVAL;
  evf_bwd_defer_tab[c].slot = d;
  return EVF_OK;
}
extern "C" int evf_bwd_defer_pending(void* stream) {
  const int c = evf_ctx_find(stream);
  if (c < 0 || !evf_bwd_defer_tab[c].active) return 0;
  int n = evf_dg_defer_count(c) + evf_hd_defer_count(c);
  for (int d = 0; d < EVF_BWD_DIAGS; ++d) n += fb_tab[c].n[d];
  return n;
}
extern "C" int evf_bwd_defer_flush(void* stream) {
  const int c = evf_ctx_find(stream);
  if (c < 0 || !evf_bwd_defer_tab[c].active) return EVF_OK;
  const int rc = evf_bwd_defer_flush_now(c, stream, true);
  evf_bwd_defer_tab[c].active = false;
  evf_ctx_drop(c);
  return rc;
}

static int fb_launch(const float* g_z_out, const float* g_z_out2, const FbTop* topp, const float* g_v_out, const float* v_out, const float* v_prev,
                     const uint32_t* z_prev, const uint32_t* xT, const uint32_t* zT_prev, const float* leak,
                     const float* thresh, int B, int H, int W, int hard_reset, int surrogate, float act_width,
                     float* g_cur, void* g_split, float* g_v_prev, float* g_leak, float* g_thresh, float* slab_ff,
                     float* slab_rec, int accumulate, void* stream, const FbPlif* plp = nullptr) {
  if (plp && !(hard_reset != 0 && surrogate == EVF_ARCTAN)) return EVF_ENOTSUP;  // (the trace backward lives in the two-team body)
  if (plp && plp->xl == 2 && topp) return EVF_ENOTSUP;  // (an ALIF cell under the prediction head, one pass: evf_pred_bwd + evf_plif_bwd_wgrad2)
  if (!v_out || !xT || !leak || !thresh || (!g_cur && !g_split) || !g_v_prev || !g_leak || !g_thresh || !slab_ff || B <= 0 || H <= 0 ||
      W <= 0 || ((zT_prev != nullptr) != (slab_rec != nullptr)) || (topp && (g_z_out || zT_prev)) || (topp && g_z_out2))
    return EVF_EINVAL;
  const int row_ld = accumulate >> 8;  // pitch of the per-block parameter-gradient rows (0: dense outputs, atomics)
  accumulate &= 1;
  const long nunits = fb_units(B, H, W);
  const int nchunk = (W + FB_CW - 1) / FB_CW;
  const int nrows_all = fb_rows(nunits);
  dim3 grid(fb_blocks_per_cell(nunits, 1)), block(FB_THREADS);
  hipStream_t st = EVF_STREAM(stream);
  const FbTop top = topp ? *topp : FbTop{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  static bool attr[12] = {false};
  const bool fast = hard_reset != 0 && surrogate == EVF_ARCTAN;
  const int bctx = evf_ctx_find(stream);
  const EvfBwdDefer evf_bwd_defer = bctx >= 0 ? evf_bwd_defer_tab[bctx] : EvfBwdDefer{false, 0, false};
  FbDefer& fb_defer = fb_tab[bctx < 0 ? 0 : bctx];
  if (evf_bwd_defer.active) {
    bool any = false;
    for (int d = 0; d < EVF_BWD_DIAGS && !any; ++d) any = fb_defer.n[d] != 0;
    bool same = !any || (fb_defer.B == B && fb_defer.H == H && fb_defer.W == W && fb_defer.row_ld == row_ld);
    for (int d = 0; d < EVF_BWD_DIAGS && same; ++d)
      if (fb_defer.n[d])  // one neuron model per recording
        same = (fb_defer.job[d][0].kind >= 3) == (plp != nullptr) && (!plp || fb_defer.job[d][0].pl.xl == plp->xl);
    if (fast && (g_cur || g_split) && same && fb_defer.n[evf_bwd_defer.slot] < FB_MAX_JOBS) {
      fb_defer.B = B, fb_defer.H = H, fb_defer.W = W, fb_defer.row_ld = row_ld;
      FbJob& J = fb_defer.job[evf_bwd_defer.slot][fb_defer.n[evf_bwd_defer.slot]++];
      J = FbJob{(const float4*)g_z_out, (const float4*)g_z_out2, (const float4*)g_v_out, (const float4*)v_out,
                (const float4*)v_prev, z_prev, xT, zT_prev, leak, thresh, (float4*)g_cur, (uint2*)g_split, (float4*)g_v_prev, g_leak, g_thresh,
                slab_ff, slab_rec, top, act_width, accumulate, (topp ? 2 : (zT_prev ? 1 : 0)) + (plp ? 3 : 0), 0, 0, 0,
                plp ? *plp : FbPlif{}};
      return EVF_OK;
    }
    const int rc = evf_bwd_defer_flush_now(bctx, stream);  // not recordable: everything recorded runs first
    if (rc) return rc;
  }
  // One cell of the default neuron through the two-team body as well (k_bwd_diag_ws<8> with a one-entry table;
  // EVF_BWD_ONE=fused: k_lif_bwd_wgrad, every wave through all phases)
  static const bool one_teams = []() {
    const char* e = getenv("EVF_BWD_ONE");
    return !(e && e[0] == 'f');
  }();
  if (fast && (one_teams || plp) && (g_cur || g_split)) {
    static bool attr_ws = false;
    if (!attr_ws) {
      (void)hipFuncSetAttribute((const void*)k_bwd_diag_ws<8>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
      (void)hipFuncSetAttribute((const void*)k_bwd_diag_ws_plif, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
      (void)hipFuncSetAttribute((const void*)k_bwd_diag_ws_alif, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
      attr_ws = true;
    }
    FbJobs jobs;
    const FbJob J{(const float4*)g_z_out, (const float4*)g_z_out2, (const float4*)g_v_out, (const float4*)v_out, (const float4*)v_prev,
                  z_prev, xT, zT_prev, leak, thresh, (float4*)g_cur, (uint2*)g_split, (float4*)g_v_prev, g_leak, g_thresh, slab_ff,
                  slab_rec, top, act_width, accumulate, (topp ? 2 : (zT_prev ? 1 : 0)) + (plp ? 3 : 0), 0, 0, 0,
                  plp ? *plp : FbPlif{}};
    const int nblk = fb_blocks_per_cell(nunits, 1, 8);
    for (int k = 0; k < FB_MAX_JOBS; ++k) jobs.j[k] = J, jobs.j[k].blk0 = k ? 0x7fffffff : 0, jobs.j[k].nblk = nblk;
    if (plp && plp->xl == 2)
      hipLaunchKernelGGL(k_bwd_diag_ws_alif, dim3(nblk), dim3(768), FB_LDS, st, jobs, B, H, W, nchunk, nunits, row_ld, nrows_all);
    else if (plp)
      hipLaunchKernelGGL(k_bwd_diag_ws_plif, dim3(nblk), dim3(768), FB_LDS, st, jobs, B, H, W, nchunk, nunits, row_ld, nrows_all);
    else
      hipLaunchKernelGGL(k_bwd_diag_ws<8>, dim3(nblk), dim3(768), FB_LDS, st, jobs, B, H, W, nchunk, nunits, row_ld, nrows_all);
    return evf_status();
  }
#define FB_GO(REC_, TOP_, FAST_, slot)                                                                                    \
  do {                                                                                                                    \
    if (!attr[slot]) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)k_lif_bwd_wgrad<REC_, TOP_, FAST_>,                                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);                                      \
      attr[slot] = true;                                                                                                  \
    }                                                                                                                     \
    hipLaunchKernelGGL((k_lif_bwd_wgrad<REC_, TOP_, FAST_>), grid, block, FB_LDS, st, (const float4*)g_z_out,             \
                       (const float4*)g_z_out2, (const float4*)g_v_out, (const float4*)v_out, (const float4*)v_prev, z_prev, xT, zT_prev, leak,    \
                       thresh, B, H, W, nchunk, nunits, hard_reset, surrogate, act_width, accumulate, nrows_all, (float4*)g_cur,     \
                       (uint2*)g_split, (float4*)g_v_prev, g_leak, g_thresh, slab_ff, slab_rec, top, row_ld);             \
  } while (0)
#define FB_GO2(REC_, TOP_, slot)            \
  do {                                      \
    if (fast)                               \
      FB_GO(REC_, TOP_, true, 2 * (slot));  \
    else                                    \
      FB_GO(REC_, TOP_, false, 2 * (slot) + 1); \
  } while (0)
  if (topp)
    FB_GO2(false, true, 2);
  else if (zT_prev)
    FB_GO2(true, false, 1);
  else
    FB_GO2(false, false, 0);
#undef FB_GO2
#undef FB_GO
  return evf_status();
}

extern "C" int evf_lif_bwd_wgrad(const float* g_z_out, const float* g_v_out, const float* v_out, const float* v_prev,
                                 const uint32_t* z_prev, const uint32_t* xT, const uint32_t* zT_prev, const float* leak,
                                 const float* thresh, int B, int H, int W, int hard_reset, int surrogate,
                                 float act_width, float* g_cur, void* g_split, float* g_v_prev, float* g_leak,
                                 float* g_thresh, float* slab_ff, float* slab_rec, int accumulate, void* stream) {
  return fb_launch(g_z_out, nullptr, nullptr, g_v_out, v_out, v_prev, z_prev, xT, zT_prev, leak, thresh, B, H, W, hard_reset,
                   surrogate, act_width, g_cur, g_split, g_v_prev, g_leak, g_thresh, slab_ff, slab_rec, accumulate, stream);
}

// The same with dL/d(output spikes) in TWO parts, g_z_out + g_z_out2 (either may be NULL): the part from
// the layer above (this pass) and the part from the cell's own recurrent input gradient (one pass later) stay in separate
// buffers and meet here -- the order of those two launches then is free (evf_bwd_defer_*), and the input gradient of the layer
// above no longer reads and rewrites the other part (its accumulating form).
extern "C" int evf_lif_bwd_wgrad2(const float* g_z_out, const float* g_z_out2, const float* g_v_out, const float* v_out,
                                  const float* v_prev, const uint32_t* z_prev, const uint32_t* xT, const uint32_t* zT_prev,
                                  const float* leak, const float* thresh, int B, int H, int W, int hard_reset, int surrogate,
                                  float act_width, float* g_cur, void* g_split, float* g_v_prev, float* g_leak,
                                  float* g_thresh, float* slab_ff, float* slab_rec, int accumulate, void* stream) {
  return fb_launch(g_z_out, g_z_out2, nullptr, g_v_out, v_out, v_prev, z_prev, xT, zT_prev, leak, thresh, B, H, W, hard_reset,
                   surrogate, act_width, g_cur, g_split, g_v_prev, g_leak, g_thresh, slab_ff, slab_rec, accumulate, stream);
}

// The non-recurrent layer directly under the prediction head, with the head's backward (evf_pred_bwd) inside:
// flow / g_flow [B,2,H,W], pred_w [2][32], z_out = this layer's output spikes [B,H,W]; d_pred_w [2][32] and
// d_pred_b [2] are accumulated.  One launch and 2 x 128 B/pixel of HBM traffic less than evf_pred_bwd + evf_lif_bwd_wgrad.
extern "C" int evf_lif_bwd_wgrad_top(const float* flow, const float* g_flow, const float* pred_w, const uint32_t* z_out,
                                     float* d_pred_w, float* d_pred_b, const float* g_v_out, const float* v_out,
                                     const float* v_prev, const uint32_t* z_prev, const uint32_t* xT, const float* leak,
                                     const float* thresh, int B, int H, int W, int hard_reset, int surrogate,
                                     float act_width, float* g_cur, void* g_split, float* g_v_prev, float* g_leak,
                                     float* g_thresh, float* slab_ff, int accumulate, void* stream) {
  if (!flow || !g_flow || !pred_w || !z_out || !d_pred_w || !d_pred_b) return EVF_EINVAL;
  const FbTop top{flow, g_flow, pred_w, z_out, d_pred_w, d_pred_b};
  return fb_launch(nullptr, nullptr, &top, g_v_out, v_out, v_prev, z_prev, xT, nullptr, leak, thresh, B, H, W, hard_reset, surrogate,
                   act_width, g_cur, g_split, g_v_prev, g_leak, g_thresh, slab_ff, nullptr, accumulate, stream);
}

// PLIF cells (spiking_submodules.py:557-655): evf_lif_bwd_wgrad2 / _top with the presynaptic trace's backward in the same pass
// (what evf_plif_trace_bwd computes from g_cur as a pass of its own): g_pt_carry [B,H,W,32] or NULL, pt_prev [B,H,W,32] or
// NULL, P [B,H,W]; out: g_pt_prev [B,H,W,32] (may alias g_pt_carry), g_P_raw [B,H,W], g_leak_pt / g_add_pt (rows of pitch
// accumulate >> 8, like g_leak).  Default neuron only (hard reset, arctan surrogate): EVF_ENOTSUP otherwise.
extern "C" int evf_plif_bwd_wgrad2(const float* g_z_out, const float* g_z_out2, const float* g_v_out, const float* v_out,
                                   const float* v_prev, const uint32_t* z_prev, const uint32_t* xT, const uint32_t* zT_prev,
                                   const float* leak, const float* thresh, int B, int H, int W, int hard_reset, int surrogate,
                                   float act_width, float* g_cur, void* g_split, float* g_v_prev, float* g_leak, float* g_thresh,
                                   float* slab_ff, float* slab_rec, int accumulate, const float* g_pt_carry, const float* pt_prev,
                                   const float* P, const float* leak_pt, const float* add_pt, float* g_pt_prev, float* g_P_raw,
                                   float* g_leak_pt, float* g_add_pt, void* stream) {
  if (!P || !leak_pt || !add_pt || !g_pt_prev || !g_P_raw || !g_leak_pt || !g_add_pt) return EVF_EINVAL;
  const int xl = (hard_reset >> 1) & 3;  // (bits 1-2 of hard_reset: 1 an XLIF cell, 2 an ALIF cell -- g_P_raw then is g_zx [B,H,W,32])
  const FbPlif pl{(const float4*)g_pt_carry, (const float4*)pt_prev, P, leak_pt, add_pt, (float4*)g_pt_prev, g_P_raw, g_leak_pt, g_add_pt,
                  xl, xl == 2 ? (float4*)g_P_raw : nullptr};
  return fb_launch(g_z_out, g_z_out2, nullptr, g_v_out, v_out, v_prev, z_prev, xT, zT_prev, leak, thresh, B, H, W, hard_reset & 1,
                   surrogate, act_width, g_cur, g_split, g_v_prev, g_leak, g_thresh, slab_ff, slab_rec, accumulate, stream, &pl);
}
extern "C" int evf_plif_bwd_wgrad_top(const float* flow, const float* g_flow, const float* pred_w, const uint32_t* z_out,
                                      float* d_pred_w, float* d_pred_b, const float* g_v_out, const float* v_out,
                                      const float* v_prev, const uint32_t* z_prev, const uint32_t* xT, const float* leak,
                                      const float* thresh, int B, int H, int W, int hard_reset, int surrogate, float act_width,
                                      float* g_cur, void* g_split, float* g_v_prev, float* g_leak, float* g_thresh, float* slab_ff,
                                      int accumulate, const float* g_pt_carry, const float* pt_prev, const float* P,
                                      const float* leak_pt, const float* add_pt, float* g_pt_prev, float* g_P_raw, float* g_leak_pt,
                                      float* g_add_pt, void* stream) {
  if (!flow || !g_flow || !pred_w || !z_out || !d_pred_w || !d_pred_b) return EVF_EINVAL;
  if (!P || !leak_pt || !add_pt || !g_pt_prev || !g_P_raw || !g_leak_pt || !g_add_pt) return EVF_EINVAL;
  const FbTop top{flow, g_flow, pred_w, z_out, d_pred_w, d_pred_b};
  const FbPlif pl{(const float4*)g_pt_carry, (const float4*)pt_prev, P, leak_pt, add_pt, (float4*)g_pt_prev, g_P_raw, g_leak_pt, g_add_pt,
                  (hard_reset >> 1) & 3, nullptr};
  return fb_launch(nullptr, nullptr, &top, g_v_out, v_out, v_prev, z_prev, xT, nullptr, leak, thresh, B, H, W, hard_reset & 1, surrogate,
                   act_width, g_cur, g_split, g_v_prev, g_leak, g_thresh, slab_ff, nullptr, accumulate, stream, &pl);
}

// A feed-forward PLIF hidden cell, ALL passes of a window in one launch (k_bwd_win_plif): the arrays hold np device pointers, index 0
// = the window's LAST pass (backward order).  Per pass: g_z (dL/d(spikes), may be NULL), v_out / v_prev (v_prev NULL: zero state;
// v_out[s] must be v_prev[s - 1]: only v_out[0] is read), z_prev (spike words before the pass, NULL: none), xT (input spike planes),
// pt_prev (NULL: zero), P; out per pass: g_cur, g_P_raw.  dL/dv and dL/d(pt) are carried in registers from pass to pass and start at
// zero behind the window's last pass; g_v_prev / g_pt_prev (may be NULL): the gradients on the state entering the window.  Slab and
// per-channel sums as evf_plif_bwd_wgrad2 (accumulate: bit 0 = add to the slab, bits 8.. = pitch of the per-block rows).
static int fb_window_launch(int np, const void* const* g_z, const void* const* flow, const void* const* g_flow, const void* const* z_out,
                            const float* pred_w, float* d_pred_w, float* d_pred_b, const void* const* v_out, const void* const* v_prev,
                            const void* const* z_prev, const void* const* xT, void* const* g_cur, void* const* g_split,
                            const void* const* pt_prev, const void* const* P, void* const* g_P_raw, const float* leak, const float* thresh,
                            const float* leak_pt,
                            const float* add_pt, int B, int H, int W, float act_width, float* g_v_prev, float* g_pt_prev, float* g_leak,
                            float* g_thresh, float* g_leak_pt, float* g_add_pt, float* slab_ff, int accumulate, void* stream) {
  const bool top = flow != nullptr, plif = leak_pt != nullptr;
  if (np < 1 || np > FB_WIN_MAX || (!top && !g_z) || !v_out || !v_prev || !z_prev || !xT || (!g_cur && !g_split) || !leak || !thresh ||
      !g_leak || !g_thresh || !slab_ff || B <= 0 || H <= 0 || W <= 0 || (top && (!g_flow || !z_out || !pred_w || !d_pred_w || !d_pred_b)) ||
      (plif && (!pt_prev || !P || !g_P_raw || !add_pt || !g_leak_pt || !g_add_pt)))
    return EVF_EINVAL;
  FbWin Wn;
  Wn.np = np;
  for (int s = 0; s < FB_WIN_MAX; ++s) {
    const int q = s < np ? s : 0;
    if (!v_out[q] || !xT[q] || (plif && (!P[q] || !g_P_raw[q])) || (top && (!flow[q] || !g_flow[q] || !z_out[q])) ||
        !((g_cur && g_cur[q]) || (g_split && g_split[q])))
      return EVF_EINVAL;
    Wn.gz[s] = top ? nullptr : (const float4*)g_z[q];
    Wn.vo[s] = (const float4*)v_out[q], Wn.vp[s] = (const float4*)v_prev[q];
    Wn.zp[s] = (const uint32_t*)z_prev[q], Wn.xT[s] = (const uint32_t*)xT[q], Wn.gcur[s] = g_cur ? (float4*)g_cur[q] : nullptr;
    Wn.gsp[s] = g_split ? (uint2*)g_split[q] : nullptr;
    Wn.pp[s] = plif ? (const float4*)pt_prev[q] : nullptr, Wn.P[s] = plif ? (const float*)P[q] : nullptr;
    Wn.gP[s] = plif ? (float*)g_P_raw[q] : nullptr;
    Wn.flow[s] = top ? (const float*)flow[q] : nullptr, Wn.g_flow[s] = top ? (const float*)g_flow[q] : nullptr;
    Wn.z_out[s] = top ? (const uint32_t*)z_out[q] : nullptr;
  }
  const int row_ld = accumulate >> 8;
  const long nunits = fb_units(B, H, W);
  const int nchunk = (W + FB_CW - 1) / FB_CW;
  FbJob J{};
  J.leak = leak, J.thresh = thresh, J.g_v_prev = (float4*)g_v_prev, J.g_leak = g_leak, J.g_thresh = g_thresh, J.slab_ff = slab_ff;
  J.width = act_width, J.accumulate = accumulate & 1, J.kind = (top ? 2 : 0) + (plif ? 3 : 0);
  J.top = FbTop{nullptr, nullptr, pred_w, nullptr, d_pred_w, d_pred_b};
  J.pl = FbPlif{nullptr, nullptr, nullptr, leak_pt, add_pt, (float4*)g_pt_prev, nullptr, g_leak_pt, g_add_pt,
                (accumulate >> 1) & 3, nullptr};  // (bits 1-2 of accumulate: 1 an XLIF cell, 2 an ALIF cell)
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_bwd_win_plif, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_win_plif_top, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_win_alif, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_win_alif_top, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_win_lif, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)k_bwd_win_lif_top, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    attr = true;
  }
  // blocks: as a one-cell launch whose units cost np times as much (whole rounds of one block per CU)
  const int nblk = fb_blocks_per_cell(nunits, 1, 8 * np);
  evf_prof_mark(8, 0, stream);
#define FB_WIN_GO(K_) hipLaunchKernelGGL(K_, dim3(nblk), dim3(768), FB_LDS, EVF_STREAM(stream), J, Wn, B, H, W, nchunk, nunits, row_ld, fb_rows(nunits))
  if (plif && J.pl.xl == 2) {
    if (top) FB_WIN_GO(k_bwd_win_alif_top); else FB_WIN_GO(k_bwd_win_alif);
  } else if (plif) {
    if (top) FB_WIN_GO(k_bwd_win_plif_top); else FB_WIN_GO(k_bwd_win_plif);
  } else {
    if (top) FB_WIN_GO(k_bwd_win_lif_top); else FB_WIN_GO(k_bwd_win_lif);
  }
#undef FB_WIN_GO
  evf_prof_mark(8, 1, stream);
  return evf_status();
}

extern "C" int evf_plif_bwd_wgrad_window(int np, const void* const* g_z, const void* const* v_out, const void* const* v_prev,
                                         const void* const* z_prev, const void* const* xT, void* const* g_cur, void* const* g_split,
                                         const void* const* pt_prev, const void* const* P, void* const* g_P_raw, const float* leak,
                                         const float* thresh, const float* leak_pt, const float* add_pt, int B, int H, int W,
                                         float act_width, float* g_v_prev, float* g_pt_prev, float* g_leak, float* g_thresh,
                                         float* g_leak_pt, float* g_add_pt, float* slab_ff, int accumulate, void* stream) {
  if (!leak_pt) return EVF_EINVAL;
  return fb_window_launch(np, g_z, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, v_out, v_prev, z_prev, xT, g_cur, g_split, pt_prev,
                          P, g_P_raw, leak, thresh, leak_pt, add_pt, B, H, W, act_width, g_v_prev, g_pt_prev, g_leak, g_thresh, g_leak_pt,
                          g_add_pt, slab_ff, accumulate, stream);
}
// ... of the (feed-forward) layer under the prediction head, with the head's backward inside (evf_plif_bwd_wgrad_top per pass):
// flow / g_flow [B,2,H,W] and z_out [B,H,W] per pass instead of g_z; d_pred_w [2][32] / d_pred_b [2] rows like g_leak.
extern "C" int evf_plif_bwd_wgrad_window_top(int np, const void* const* flow, const void* const* g_flow, const float* pred_w,
                                             const void* const* z_out, float* d_pred_w, float* d_pred_b, const void* const* v_out,
                                             const void* const* v_prev, const void* const* z_prev, const void* const* xT,
                                             void* const* g_cur, void* const* g_split, const void* const* pt_prev, const void* const* P,
                                             void* const* g_P_raw, const float* leak, const float* thresh, const float* leak_pt,
                                             const float* add_pt, int B, int H, int W, float act_width, float* g_v_prev,
                                             float* g_pt_prev, float* g_leak, float* g_thresh, float* g_leak_pt, float* g_add_pt,
                                             float* slab_ff, int accumulate, void* stream) {
  if (!flow || !leak_pt) return EVF_EINVAL;
  return fb_window_launch(np, nullptr, flow, g_flow, z_out, pred_w, d_pred_w, d_pred_b, v_out, v_prev, z_prev, xT, g_cur, g_split, pt_prev, P,
                          g_P_raw, leak, thresh, leak_pt, add_pt, B, H, W, act_width, g_v_prev, g_pt_prev, g_leak, g_thresh, g_leak_pt,
                          g_add_pt, slab_ff, accumulate, stream);
}

// LIF feed-forward hidden cells (evf_lif_bwd_wgrad2 / _top per pass), all passes of a window in one launch: as the PLIF forms
// without the trace; g_cur (fp32) and / or g_split (three bf16 planes per pass, what evf_conv_dgrad_b3 reads) per pass.
// flow == NULL: a hidden cell with g_z per pass; else the layer under the prediction head (flow / g_flow / z_out per pass).
extern "C" int evf_lif_bwd_wgrad_window(int np, const void* const* g_z, const void* const* flow, const void* const* g_flow,
                                        const float* pred_w, const void* const* z_out, float* d_pred_w, float* d_pred_b,
                                        const void* const* v_out, const void* const* v_prev, const void* const* z_prev,
                                        const void* const* xT, void* const* g_cur, void* const* g_split, const float* leak,
                                        const float* thresh, int B, int H, int W, float act_width, float* g_v_prev, float* g_leak,
                                        float* g_thresh, float* slab_ff, int accumulate, void* stream) {
  return fb_window_launch(np, g_z, flow, g_flow, z_out, pred_w, d_pred_w, d_pred_b, v_out, v_prev, z_prev, xT, g_cur, g_split, nullptr,
                          nullptr, nullptr, leak, thresh, nullptr, nullptr, B, H, W, act_width, g_v_prev, nullptr, g_leak, g_thresh, nullptr,
                          nullptr, slab_ff, accumulate, stream);
}
