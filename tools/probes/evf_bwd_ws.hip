// LAB NOTEBOOK, not part of the library build: fused neuron backward + weight-gradient conv,
// wave-specialised form (gfx950).  Verified against k_lif_bwd_wgrad (all kernel / network tests) and measured in the
// hipGraph-replayed train step with rocprofv3: 33.1-34.3 us against 34.0-34.2 us (feed-forward form), 38.3-39.3 against
// 37.6-38.0 us (top form) -- no gain, although a block's life drops from 66 k to 55 k cycles.  To build it again: copy it
// into event_flow_amd/csrc/, declare evf_bwd_ws_launch in evf_common.h and call it from fb_launch (evf_bwd_fused.hip).
//
//
// Same arithmetic, same operand layouts and the same slab layout as k_lif_bwd_wgrad (evf_bwd_fused.hip; read its header for
// the GEMM shapes).  There every wave did everything -- request unit k+2, MFMAs of unit k, neuron backward + split of unit
// k+1, barrier -- and phase stamps showed each wave's SERIAL chain bounding the kernel (per 64-pixel unit: 0.75 k cycles of
// load issue, 1.4 k of MFMAs + their dependent LDS reads, 2.2 k of element-wise work, 1.2 k at the barrier = 5.6 k, against
// ~2 k of HBM time; dropping a whole input tensor or a third of the VALU work changed nothing).  Here the block is two teams:
//
//   waves 0..3  MATRIX team, one wave per SIMD: two taps each (+ a quarter of the ninth), spike-plane loads, slab epilogue;
//   waves 4..7  ELEMENT team: stream g_z, g_v, v', v (two pixels of a unit per thread), neuron backward, exact bf16 split of
//               g_cur into the B-operand buffer, g_cur / g_v_prev out, per-channel leak / threshold sums.
//
// The two chains run side by side on every SIMD (matrix pipe and VALU are separate issue ports) and meet at ONE barrier per
// unit: LDS operands double buffered, unit k+1 is produced while unit k is contracted.  Both teams request unit k+2 before
// they work (register stages alternate by unrolling, never by moves: a move of a loaded register is a wait for the load).
#include "evf_common.h"
#include "evf_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define C32 32
#define BW_CW 64     // pixels per unit (row segment)
#define BW_UNITS 8   // units per block (must match evf_lif_bwd_wgrad_slabs)
#define BW_EW 8                       // element-team waves (4 or 8): 64 pixels x 8 channel groups over BW_EW * 64 threads
#define BW_PPT (8 / BW_EW)            // pixels of a unit per element thread
#define BW_THREADS (256 + 64 * BW_EW)
#define BW_NW (BW_CW / 32 + 2)  // plane words per (row, channel): segment + one halo word each side
#define BW_PW (3 * C32 * BW_NW) // plane words per unit and input (384)
#define BW_R0 32768             // LDS region 0: B-operand double buffer (24 KiB), later the tap-8 reduction (16 KiB)

__device__ __forceinline__ int bw_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ float bw_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float bw_surrogate(int kind, float x, float width) {
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
      return __builtin_amdgcn_rcpf(1.0f + width * x * x);
  }
}

#ifdef BW_STAMPS  // phase stamps (debug build through EVF_LIB): [block < 16][matrix wave 0 / element wave 4][96]
__device__ unsigned long long bw_stamps[16 * 2 * 96];
extern "C" int evf_debug_fb_stamps(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(bw_stamps), sizeof(bw_stamps))); }
#define BW_STAMP()                                                                                   \
  do {                                                                                               \
    if (blockIdx.x < 16 && lane == 0 && (wv == 0 || wv == 4) && nst < 96)                            \
      bw_stamps[(blockIdx.x * 2 + (wv ? 1 : 0)) * 96 + nst++] = __builtin_readcyclecounter();       \
  } while (0)
#else
#define BW_STAMP() do {} while (0)
#endif

struct BwTop {  // the prediction head above the top layer (see FbTop in evf_bwd_fused.hip)
  const float* flow;
  const float* g_flow;
  const float* pred_w;
  const uint32_t* z_out;
  float* dw;
  float* db;
};

struct BwElem {  // one pixel (4 channels of it) of a unit in flight
  float4 gz, gv, vo, vp;
  float f0, f1, q0, q1;
  uint32_t zo, zw;
};
struct BwStageE {
  BwElem h[BW_PPT];  // pixels p0 + 32 h of the unit (BW_PPT = 2), or pixel p0 (BW_PPT = 1)
};
struct BwStageM {
  uint32_t px[2], pz[2], pin[2];  // plane words tid and tid + 256 (< BW_PW) of the unit + their in-image masks
};

template <bool REC, bool TOP>
__global__ __launch_bounds__(BW_THREADS) void k_lif_bwd_wgrad_ws(
    const float4* __restrict__ g_z_out, const float4* __restrict__ g_v_out, const float4* __restrict__ v_out,
    const float4* __restrict__ v_prev, const uint32_t* __restrict__ z_prev, const uint32_t* __restrict__ xT,
    const uint32_t* __restrict__ zT, const float* __restrict__ leak, const float* __restrict__ thresh, int B, int H, int W,
    int nchunk, int nunits, int hard_reset, int surrogate, float width, int accumulate, float4* __restrict__ g_cur,
    uint2* __restrict__ g_split, float4* __restrict__ g_v_prev, float* __restrict__ g_leak, float* __restrict__ g_thresh,
    float* __restrict__ slab_ff, float* __restrict__ slab_rec, BwTop top) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short* s_b = (unsigned short*)smem_raw;            // [2][3][BW_CW*32] bf16 (region of BW_R0 bytes)
  uint32_t* s_px = (uint32_t*)(smem_raw + BW_R0);             // [2][BW_PW]
  uint32_t* s_pz = s_px + 2 * BW_PW;                          // same (REC)
  uint4* s_lut = (uint4*)(s_pz + 2 * BW_PW);                  // [256]
  float* s_red = (float*)(s_lut + 256);                       // [2][BW_EW][32]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool elem = wv >= 4;
  const int et = tid - 256;      // 0..255 inside the element team
  const int cg = et & 7;         // channel group: channels 4cg..4cg+3
  const int p0 = (et >> 3) & (64 / BW_PPT - 1);  // first pixel of this thread in the unit
  const int i = lane & 31, kg = lane >> 5;
  const int nW = (W + 31) / 32;
  int nst = 0;
  (void)nst;
  BW_STAMP();

  // unit -> (sample, row, first column): units are dealt round-robin over the blocks (see evf_bwd_fused.hip), walked
  // incrementally (one unit step = nblk units further; no division in the loop)
  const int nblk = gridDim.x;
  const int nu = (nunits - (int)blockIdx.x + nblk - 1) / nblk;
  const int rows = B * H;
  const int step_row = nblk / nchunk, step_chk = nblk - step_row * nchunk;
  struct Pos {
    int row, chk;
  };
  auto first_pos = [&]() {
    Pos q;
    q.row = (int)blockIdx.x / nchunk;
    q.chk = (int)blockIdx.x - q.row * nchunk;
    return q;
  };
  auto advance = [&](Pos q) {
    q.chk += step_chk;
    q.row += step_row;
    if (q.chk >= nchunk) q.chk -= nchunk, ++q.row;
    return q;
  };
  auto clampp = [&](Pos q) {  // past the last unit: a valid position whose results are discarded
    if (q.row >= rows) q.row = rows - 1, q.chk = 0;
    return q;
  };

  // ------------------------------------------------------------------ element team
  float lam[4], th[4], oml[4], inv_oml[4];
  float sl[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0};
  float pwa[4] = {0, 0, 0, 0}, pwb[4] = {0, 0, 0, 0}, dwa[4] = {0, 0, 0, 0}, dwb[4] = {0, 0, 0, 0}, dba = 0.f, dbb = 0.f;
  const float4* pgz = g_z_out ? g_z_out : v_out;
  const float4* pgv = g_v_out ? g_v_out : v_out;
  const float4* pvp = v_prev ? v_prev : v_out;
  const uint32_t* pzw = z_prev ? z_prev : xT;
  const uint32_t* pzt = REC ? zT : xT;
  const bool has_gz = g_z_out != nullptr, has_gv = g_v_out != nullptr, has_vp = v_prev != nullptr, has_zw = z_prev != nullptr;
  if (!elem && tid < 256) {
    const uint32_t t = tid;
    auto pr = [&](int e) { return ((t >> e) & 1u) * 0x3F80u | (((t >> (e + 1)) & 1u) * 0x3F80u) << 16; };
    s_lut[tid] = make_uint4(pr(0), pr(2), pr(4), pr(6));
  }

  // all loads unconditional (clamped addresses, optional tensors redirected to v_out and zeroed by selects afterwards)
  auto issue_e = [&](Pos q, BwStageE& s) {
    q = clampp(q);
    const int b = q.row / H, y = q.row - b * H;  // (one division per unit and thread: H is not a compile-time constant)
    const int x0 = q.chk * BW_CW, cw = min(BW_CW, W - x0);
    // 32-bit element offsets from the (uniform) tensor bases: one address register per load instead of two -- the launcher
    // checks that B*H*W*32 fits
    const unsigned pix0 = (unsigned)q.row * (unsigned)W + (unsigned)x0;
#pragma unroll
    for (int h = 0; h < BW_PPT; ++h) {
      BwElem& e = s.h[h];
      const unsigned pc = (unsigned)min(p0 + 32 * h, cw - 1);
      const unsigned ge = (pix0 + pc) * 8u + (unsigned)cg;
      e.vo = v_out[ge];
      if (TOP) {
        const unsigned hw = (unsigned)H * (unsigned)W, qq = (unsigned)y * (unsigned)W + (unsigned)x0 + pc;
        e.f0 = top.flow[(unsigned)b * 2u * hw + qq], e.f1 = top.flow[((unsigned)b * 2u + 1u) * hw + qq];
        e.q0 = top.g_flow[(unsigned)b * 2u * hw + qq], e.q1 = top.g_flow[((unsigned)b * 2u + 1u) * hw + qq];
        e.zo = top.z_out[pix0 + pc];
      } else {
        e.gz = pgz[ge];
      }
      e.gv = pgv[ge];
      e.vp = pvp[ge];
      e.zw = pzw[z_prev ? pix0 + pc : 0u];
    }
  };
  // neuron backward of unit at `q` (valid = it exists), results to HBM, exact split of g_cur into B-operand buffer `buf`
  auto commit_e = [&](Pos q, bool valid, const BwStageE& s, int buf) {
    q = clampp(q);
    const int x0 = q.chk * BW_CW, cw = min(BW_CW, W - x0);
    const unsigned pix0 = (unsigned)q.row * (unsigned)W + (unsigned)x0;
    unsigned short* sb = s_b + buf * (3 * BW_CW * C32);
#pragma unroll
    for (int h = 0; h < BW_PPT; ++h) {
      const BwElem& e = s.h[h];
      const int p = p0 + 32 * h;
      const bool ok = p < cw && valid;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float gp0 = 0.f, gp1 = 0.f;
      if (TOP) {
        gp0 = e.q0 * (1.0f - e.f0 * e.f0);  // tanh' (evf_pred_bwd)
        gp1 = e.q1 * (1.0f - e.f1 * e.f1);
      }
      const float4 gz4 = TOP ? make_float4(gp0 * pwa[0] + gp1 * pwb[0], gp0 * pwa[1] + gp1 * pwb[1], gp0 * pwa[2] + gp1 * pwb[2],
                                           gp0 * pwa[3] + gp1 * pwb[3])
                             : (has_gz ? e.gz : z4);
      const float4 gv4 = has_gv ? e.gv : z4, vp4 = has_vp ? e.vp : z4;
      if (TOP && ok) {
        const uint32_t zo = e.zo >> (4 * cg);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool on = (zo >> c) & 1u;
          dwa[c] += on ? gp0 : 0.f;
          dwb[c] += on ? gp1 : 0.f;
        }
        if (cg == 0) dba += gp0, dbb += gp1;
      }
      const float vo[4] = {e.vo.x, e.vo.y, e.vo.z, e.vo.w}, gz[4] = {gz4.x, gz4.y, gz4.z, gz4.w};
      const float gvo[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, vp[4] = {vp4.x, vp4.y, vp4.z, vp4.w};
      const uint32_t zw = (has_zw ? e.zw : 0u) >> (4 * cg);
      float gc[4], gp[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // autograd of spiking_submodules.py:103-126 / :523-551 (the expressions of k_lif_bwd_wgrad, term for term)
        const float z = (float)((zw >> c) & 1u);
        const float sg = bw_surrogate(surrogate, vo[c] - th[c], width);
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
        const unsigned o = (pix0 + (unsigned)p) * 8u + (unsigned)cg;
        if (g_cur) g_cur[o] = make_float4(gc[0], gc[1], gc[2], gc[3]);
        g_v_prev[o] = make_float4(gp[0], gp[1], gp[2], gp[3]);
      }
      // exact split g = hi + mid + lo (evf_split.h), stored in B-operand order:
      // pixel p = 16*kq + 8*kgp + ee, element ((kq*2 + kgp)*32 + j)*8 + ee
      uint32_t tp[3][2];
      const int base = ((p >> 3) * C32) * 8 + (p & 7);
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        evf_split3_pair(ok ? gc[2 * e2] : 0.f, ok ? gc[2 * e2 + 1] : 0.f, tp[0][e2], tp[1][e2], tp[2][e2]);
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
          const int o = t3 * BW_CW * C32 + base + (4 * cg + 2 * e2) * 8;
          sb[o] = (unsigned short)tp[t3][e2];
          sb[o + 8] = (unsigned short)(tp[t3][e2] >> 16);
        }
      }
      if (ok && g_split) {  // the same split as three bf16 planes [term][pix][32] for evf_conv_dgrad_b3
        const long ps = (long)B * H * W * 8;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) g_split[t3 * ps + (long)((pix0 + (unsigned)p) * 8u + (unsigned)cg)] = make_uint2(tp[t3][0], tp[t3][1]);
      }
    }
  };

  // ------------------------------------------------------------------ matrix team
  // plane words of the unit: (dy, channel, word) = tid and tid + 256
  auto issue_m = [&](Pos q, BwStageM& s) {
    q = clampp(q);
    const int b = q.row / H, y = q.row - b * H;
    const int xw0 = q.chk * (BW_CW / 32) - 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = min(tid + 256 * h, BW_PW - 1);
      const int wq = t % BW_NW, c = (t / BW_NW) % C32, dyy = t / (BW_NW * C32);
      const int yy = y + dyy - 1, xw = xw0 + wq;
      const bool in = yy >= 0 && yy < H && xw >= 0 && xw < nW;
      const unsigned src = in ? ((unsigned)(b * H + yy) * C32 + (unsigned)c) * (unsigned)nW + (unsigned)xw : 0u;
      s.px[h] = xT[src];
      s.pz[h] = pzt[src];
      s.pin[h] = in ? 0xFFFFFFFFu : 0u;  // (applied in commit_m: an AND here would be a wait for the load)
    }
  };
  auto commit_m = [&](const BwStageM& s, int buf) {
    s_px[buf * BW_PW + tid] = s.px[0] & s.pin[0];
    if (REC) s_pz[buf * BW_PW + tid] = s.pz[0] & s.pin[0];
    if (tid + 256 < BW_PW) {
      s_px[buf * BW_PW + tid + 256] = s.px[1] & s.pin[1];
      if (REC) s_pz[buf * BW_PW + tid + 256] = s.pz[1] & s.pin[1];
    }
  };
  f32x16 acc[2] = {{0}, {0}}, acc8 = {0}, accz[2] = {{0}, {0}}, accz8 = {0};
  float old_ff[2][16], prev8[4] = {0.f, 0.f, 0.f, 0.f};  // previous slab sums (feed-forward form: prefetched in the prologue)
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int q = 0; q < 16; ++q) old_ff[t2][q] = 0.f;
  auto mfma_unit = [&](int buf) {  // wave wv (0..3): taps 2 wv, 2 wv + 1 over the whole unit, tap 8 for pixel group kq == wv
    const uint4* sbh = (const uint4*)(s_b + buf * (3 * BW_CW * C32));
    const uint32_t* px = s_px + buf * BW_PW;
    const uint32_t* pz = s_pz + buf * BW_PW;
    auto afrag = [&](const uint32_t* planes, int ddy, int ddx, int kq) -> bf16x8 {
      const int q = 32 + 16 * kq + 8 * kg + ddx - 1;  // bit offset of the first of the 8 pixels
      const uint32_t* wr = planes + (ddy * C32 + i) * BW_NW + (q >> 5);
      const uint32_t byte = __funnelshift_r(wr[0], wr[1], q & 31) & 0xFFu;
      const uint4 a = s_lut[byte];
      return *(const bf16x8*)&a;
    };
#pragma unroll
    for (int kq = 0; kq < BW_CW / 16; ++kq) {
      const int fo = (kq * 2 + kg) * C32 + i;  // uint4 index of this lane's 8 pixels of channel i (= co)
      const uint4 uh = sbh[fo], um = sbh[BW_CW * C32 / 8 + fo], ul = sbh[2 * BW_CW * C32 / 8 + fo];
      const bf16x8 bh = *(const bf16x8*)&uh, bm = *(const bf16x8*)&um, bl = *(const bf16x8*)&ul;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int tap = 2 * wv + t2, dy = tap / 3, dx = tap - 3 * dy;
        const bf16x8 a = afrag(px, dy, dx, kq);
        acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc[t2], 0, 0, 0);
        acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bm, acc[t2], 0, 0, 0);
        acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc[t2], 0, 0, 0);
        if (REC) {
          const bf16x8 az = afrag(pz, dy, dx, kq);
          accz[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bh, accz[t2], 0, 0, 0);
          accz[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bm, accz[t2], 0, 0, 0);
          accz[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bl, accz[t2], 0, 0, 0);
        }
      }
      if (kq == wv) {  // this wave's share of the ninth tap
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
      // (keep the scheduler from hoisting every pixel group's operands to the top of the unit: with the accumulators and the
      //  prefetched slab sums live that spills)
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ------------------------------------------------------------------ the pipeline
  //   step k:  ELEMENT: request unit k+2 | backward + split of unit k+1 -> buffer (k+1)&1
  //            MATRIX : request planes of unit k+2 | MFMAs of unit k from buffer k&1 | planes of unit k+1 -> buffer (k+1)&1
  //            barrier
  // The two teams run SEPARATE loops with the same number of barriers (1 + nu + 2), so that the register allocator can
  // overlay the element team's stages with the matrix team's accumulators; each loop is unrolled by two so that the register
  // stages alternate without moves.
  const Pos q0 = first_pos();
  if (elem) {
    // THREE stages: while unit k+1 is worked on, the requests of units k+2 and k+3 are in flight (one step is ~3 k cycles and a
    // loaded HBM round trip was longer than that: with two stages the element team sat in s_waitcnt for ~1.5 k cycles per unit)
    BwStageE e0, e1, e2;
    Pos qa = advance(q0), qb = advance(qa), qc = advance(qb);  // units k+1, k+2, k+3
    issue_e(q0, e0);
    issue_e(qa, e1);
    issue_e(qb, e2);
    // (per-channel constants AFTER the first requests: their loads and the expf chains hide under the cold HBM round trip)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lam[k] = bw_sigmoid(leak[4 * cg + k]);
      th[k] = fmaxf(thresh[4 * cg + k], 0.01f);
      oml[k] = 1.0f - lam[k];
      inv_oml[k] = 1.0f / oml[k];
    }
    if (TOP) {
#pragma unroll
      for (int k = 0; k < 4; ++k) pwa[k] = top.pred_w[4 * cg + k], pwb[k] = top.pred_w[C32 + 4 * cg + k];
    }
    BW_STAMP();
    commit_e(q0, nu > 0, e0, 0);
    BW_STAMP();
    __syncthreads();
#define BW_ESTEP(RQ, RS)                        \
  {                                             \
    BW_STAMP();                                 \
    issue_e(qc, RQ);                            \
    BW_STAMP();                                 \
    commit_e(qa, k + 1 < nu, RS, (k + 1) & 1);  \
    BW_STAMP();                                 \
    __syncthreads();                            \
    qa = qb, qb = qc, qc = advance(qc);         \
    if (++k >= nu) break;                       \
  }
    for (int k = 0; k < nu;) {  // unrolled by three: the register stages rotate without moves
      BW_ESTEP(e0, e1)
      BW_ESTEP(e1, e2)
      BW_ESTEP(e2, e0)
    }
#undef BW_ESTEP
  } else {
    BwStageM ma, mb;
    Pos q1 = advance(q0), q2 = advance(q1);
    issue_m(q0, ma);
    issue_m(q1, mb);
    // the previous partial sums of this block's slab rows (read-modify-write at the end) are requested NOW: the matrix team
    // idles through the prologue anyway, and at the end they would be one more cold round trip (the epilogue was 8 k cycles)
    if (!REC && (accumulate & 1)) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          old_ff[t2][q] = slab_ff[(long)blockIdx.x * (9 * C32 * C32) + (2 * wv + t2) * (C32 * C32) + i + bw_row(q, lane) * C32];
    }
    BW_STAMP();
    commit_m(ma, 0);
    BW_STAMP();
    __syncthreads();
    for (int k = 0; k < nu; k += 2) {
      BW_STAMP();
      issue_m(q2, ma);
      BW_STAMP();
      mfma_unit(0);
      commit_m(mb, 1);
      BW_STAMP();
      __syncthreads();
      if (k + 1 >= nu) break;
      const Pos q3 = advance(q2);
      BW_STAMP();
      issue_m(q3, mb);
      BW_STAMP();
      mfma_unit(1);
      commit_m(ma, 0);
      BW_STAMP();
      __syncthreads();
      q1 = q3;
      q2 = advance(q3);
    }
  }

  // ------------------------------------------------------------------ epilogue
  BW_STAMP();
  if (!elem) {
    // weight-gradient slabs: taps 2 wv, 2 wv + 1 straight from the owning wave (previous partial sums loaded together, selected
    // afterwards: a load under a branch is its own round trip)
    float prev8z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const long off = (long)blockIdx.x * (9 * C32 * C32) + (2 * wv + t2) * (C32 * C32) + i;
      float* d = slab_ff + off;
      if (!REC) {  // previous sums were requested in the prologue (zeros when this launch starts the slab)
#pragma unroll
        for (int q = 0; q < 16; ++q) d[bw_row(q, lane) * C32] = old_ff[t2][q] + acc[t2][q];
      } else {
        float* dz = slab_rec + off;
        float old[16], oldz[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          old[q] = d[bw_row(q, lane) * C32];
          oldz[q] = dz[bw_row(q, lane) * C32];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          d[bw_row(q, lane) * C32] = ((accumulate & 1) ? old[q] : 0.f) + acc[t2][q];
          dz[bw_row(q, lane) * C32] = ((accumulate & 1) ? oldz[q] : 0.f) + accz[t2][q];
        }
      }
    }
    // the ninth-tap tile's previous sums (four words per thread and slab), consumed after the LDS reduction below
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const long o8 = (long)blockIdx.x * (9 * C32 * C32) + 8 * (C32 * C32) + tid + h * 256;
      const float a = slab_ff[o8], az = REC ? slab_rec[o8] : 0.f;
      prev8[h] = (accumulate & 1) ? a : 0.f;
      prev8z[h] = (accumulate & 1) ? az : 0.f;
    }
    // tap 8: sum the 4 partial tiles through LDS (aliases the operand buffers: every wave is past the last barrier)
    float* s_t8 = (float*)smem_raw;  // [REC ? 2 : 1][4][1024]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      s_t8[wv * (C32 * C32) + bw_row(q, lane) * C32 + i] = acc8[q];
      if (REC) s_t8[(4 + wv) * (C32 * C32) + bw_row(q, lane) * C32 + i] = accz8[q];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();        // (first of two barriers the element team mirrors below)
#pragma unroll
    for (int h = 0; h < 4; ++h) {  // C32*C32 = 4 * 256
      const int e = tid + h * 256;
      float v = 0.f, vz = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        v += s_t8[w * (C32 * C32) + e];
        if (REC) vz += s_t8[(4 + w) * (C32 * C32) + e];
      }
      const long o8 = (long)blockIdx.x * (9 * C32 * C32) + 8 * (C32 * C32) + e;
      slab_ff[o8] = prev8[h] + v;
      if (REC) slab_rec[o8] = prev8z[h] + vz;
    }
    __builtin_amdgcn_s_barrier();
  } else {
    // per-channel sums for leak / thresh: lanes with equal (lane & 7) share channels
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        sl[c] += __shfl_xor(sl[c], o, 64);
        st[c] += __shfl_xor(st[c], o, 64);
      }
    if (TOP) {
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
    }
    float* s_top = s_red + 2 * BW_EW * C32;  // [2][BW_EW][32] + [BW_EW][2]
    if (lane < 8) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s_red[(0 * BW_EW + (wv - 4)) * C32 + 4 * lane + c] = sl[c];
        s_red[(1 * BW_EW + (wv - 4)) * C32 + 4 * lane + c] = st[c];
        if (TOP) {
          s_top[(0 * BW_EW + (wv - 4)) * C32 + 4 * lane + c] = dwa[c];
          s_top[(1 * BW_EW + (wv - 4)) * C32 + 4 * lane + c] = dwb[c];
        }
      }
    }
    if (TOP && lane == 0) s_top[2 * BW_EW * C32 + 2 * (wv - 4)] = dba, s_top[2 * BW_EW * C32 + 2 * (wv - 4) + 1] = dbb;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (et < 64) {
      const int which = et >> 5, c = et & 31;
      float v = 0.f;
      for (int w = 0; w < BW_EW; ++w) v += s_red[(which * BW_EW + w) * C32 + c];
      if (which == 0) {
        const float l = bw_sigmoid(leak[c]);
        evf_atomic_add(g_leak + c, v * l * (1.0f - l));
      } else if (thresh[c] > 0.01f) {
        evf_atomic_add(g_thresh + c, v);
      }
    } else if (TOP && et < 128) {
      const int which = (et - 64) >> 5, c = et & 31;
      float v = 0.f;
      for (int w = 0; w < BW_EW; ++w) v += s_top[(which * BW_EW + w) * C32 + c];
      evf_atomic_add(top.dw + which * C32 + c, v);
    } else if (TOP && et < 130) {
      float v = 0.f;
      for (int w = 0; w < BW_EW; ++w) v += s_top[2 * BW_EW * C32 + 2 * w + (et - 128)];
      evf_atomic_add(top.db + (et - 128), v);
    }
    __builtin_amdgcn_s_barrier();
  }
  BW_STAMP();
}

// LDS: B-operand double buffer (32 KiB region, reused by the tap-8 reduction: 16 / 32 KiB) + planes + LUT + the small sums
#define BW_LDS (BW_R0 + 2 * (2 * BW_PW * 4) + 256 * 16 + (2 * BW_EW * C32 + 2 * BW_EW * C32 + 2 * BW_EW + 8) * 4)

int evf_bwd_ws_launch(const float* g_z_out, const void* topp, const float* g_v_out, const float* v_out, const float* v_prev,
                      const uint32_t* z_prev, const uint32_t* xT, const uint32_t* zT_prev, const float* leak, const float* thresh,
                      int B, int H, int W, int hard_reset, int surrogate, float act_width, float* g_cur, void* g_split,
                      float* g_v_prev, float* g_leak, float* g_thresh, float* slab_ff, float* slab_rec, int accumulate,
                      void* stream) {
  const long nunits = (long)B * H * ((W + BW_CW - 1) / BW_CW);
  if ((long)B * H * W * C32 >= (1L << 31)) return EVF_EINVAL;  // 32-bit element offsets (the caller falls back to k_lif_bwd_wgrad)
  const int nchunk = (W + BW_CW - 1) / BW_CW;
  dim3 grid(evf_cdiv(nunits, BW_UNITS)), block(BW_THREADS);
  hipStream_t st = EVF_STREAM(stream);
  const BwTop top = topp ? *(const BwTop*)topp : BwTop{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  static bool a1 = false, a2 = false, a3 = false;
#define BW_GO(REC_, TOP_, flag)                                                                                            \
  do {                                                                                                                     \
    if (!flag) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)k_lif_bwd_wgrad_ws<REC_, TOP_>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                BW_LDS);                                                                                   \
      flag = true;                                                                                                         \
    }                                                                                                                      \
    hipLaunchKernelGGL((k_lif_bwd_wgrad_ws<REC_, TOP_>), grid, block, BW_LDS, st, (const float4*)g_z_out,                  \
                       (const float4*)g_v_out, (const float4*)v_out, (const float4*)v_prev, z_prev, xT, zT_prev, leak,     \
                       thresh, B, H, W, nchunk, (int)nunits, hard_reset, surrogate, act_width, accumulate, (float4*)g_cur, \
                       (uint2*)g_split, (float4*)g_v_prev, g_leak, g_thresh, slab_ff, slab_rec, top);                      \
  } while (0)
  if (topp)
    BW_GO(false, true, a3);
  else if (zT_prev)
    BW_GO(true, false, a2);
  else
    BW_GO(false, false, a1);
#undef BW_GO
  return evf_status();
}
