// General 2-D convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
// k in {1,3,5,7}, stride in {1,2}, padding k/2, any channel counts, NHWC fp32 with
// explicit pixel strides.  Used by every layer that is not one of the fused
// 32->32 binary-spike kernels: the spiking EV-FlowNet encoders / residual
// blocks / decoders (reference models/unet.py:418-465), the ANN FireNet
// (models/submodules.py:64-83,377-418), the 1x1 prediction heads and the
// stand-alone Conv{LIF,PLIF,ALIF,XLIF}[Recurrent] cells.
//
//   forward / input gradient : implicit GEMM, M = output pixels (flattened
//       b,oy,ox), N = output channels, K = taps x input channels.  One wave =
//       32 pixels x (32 NT) channels.  The A operand is read straight from
//       global memory -- a lane holds 8 consecutive channels of its pixel and
//       feeds them to 8 MFMAs (the K order inside a 16-channel chunk is a
//       permutation the packed weights share), no LDS, no VALU.  The weights
//       of one (tap, 64-channel group) are staged through LDS in fragment
//       order (ds_read_b128, conflict free), double buffered, one barrier
//       per 32 NT MFMAs per wave.
//       The input gradient of a stride-s conv is the same kernel in
//       "transposed" addressing (mode 1) with weights packed [tap][co][ci].
//   weight gradient : GEMM over pixels, gw[tap][ci][co] = sum_px x[px+tap][ci] g[px][co].
//       For the 32x32x2 fp32 MFMA a lane holds ONE k value, so the natural
//       [pixel][channel] tiles in LDS are already in fragment order (no
//       transpose).  Pixels are split over blocks (split-K), partial tiles
//       are reduced across the 4 waves in LDS and added atomically.
//
// fp32 products, fp32 accumulation: the results equal a CPU fp32 convolution
// up to summation order.
#include "evf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CG_BM 128  // output pixels per block (4 waves x 32)
#define CG_KG 64   // input channels per stage

struct ConvGeo {
  int B, SH, SW, K;  // source image dims, contraction channels
  int OH, OW, N;     // output image dims, output channels
  int ksz, stride, mode;
  int lds, ldo;      // pixel strides (floats) of source / output
};

__device__ __forceinline__ int cg_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }


// ---------------------------------------------------------------------------
// weight packing.  w is the torch layout [Cout][Cin][k][k].  transpose = 0:
// K = Cin, N = Cout (forward); transpose = 1: K = Cout, N = Cin (input grad).
// dst float4 index (((nt*T + tap)*G + g)*8 + ch*2 + h)*64 + lane (nt = 32-wide N tile), element j:
//   k = g*64 + ch*16 + 8*(lane>>5) + 4h + j,   n = nt*32 + (lane&31)
// (independent of how many N tiles a block of the consumer kernel covers)
// ---------------------------------------------------------------------------
__global__ void k_pack_conv2d(const float* __restrict__ w, int Cout, int Cin, int T, int transpose, long total,
                              int cin_total, int cin_off, float4* __restrict__ dst) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int K = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
  const int G = (K + CG_KG - 1) / CG_KG;
  const int lane = idx & 63;
  long q = idx >> 6;
  const int chh = q & 7;
  q >>= 3;
  const int g = q % G;
  q /= G;
  const int tap = q % T;
  const int nt = q / T;
  const int ch = chh >> 1, h = chh & 1;
  const int n = nt * 32 + (lane & 31);
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = g * CG_KG + ch * 16 + 8 * (lane >> 5) + 4 * h + j;
    float x = 0.f;
    if (k < K && n < N) {
      const int co = transpose ? k : n, ci = transpose ? n : k;
      // input channels past the weight's own (cin_off + ci >= cin_total) are alignment padding of the activation: zero
      if (cin_off + ci < cin_total) x = w[((long)co * cin_total + cin_off + ci) * T + tap];
    }
    v[j] = x;
  }
  dst[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

static long cg_packed_float4(int Cout, int Cin, int ksz, int transpose) {
  const int K = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
  const int NTILES = evf_cdiv(N, 32), G = evf_cdiv(K, CG_KG), T = ksz * ksz;
  return (long)NTILES * T * G * 512;
}

extern "C" int64_t evf_conv2d_packed_size(int Cout, int Cin, int ksz, int transpose) {
  if (Cout <= 0 || Cin <= 0 || !EVF_KSZ_OK(ksz)) return 0;
  return cg_packed_float4(Cout, Cin, ksz, transpose) * 4;
}

extern "C" int evf_pack_conv2d_weight(const float* w, int Cout, int Cin, int ksz, int transpose, int cin_total, int cin_off,
                                      float* dst, void* stream) {
  if (!w || !dst || Cout <= 0 || Cin <= 0 || !EVF_KSZ_OK(ksz) || cin_off < 0 || cin_off >= cin_total) return EVF_EINVAL;
  const long total = cg_packed_float4(Cout, Cin, ksz, transpose);
  hipLaunchKernelGGL(k_pack_conv2d, dim3(evf_cdiv(total, 256)), dim3(256), 0, EVF_STREAM(stream), w, Cout, Cin, ksz * ksz,
                     transpose, total, cin_total, cin_off, (float4*)dst);
  return evf_status();
}

// ---------------------------------------------------------------------------
// forward / input-gradient kernel
// ---------------------------------------------------------------------------
// PAR (input gradient of a stride-2 3x3 conv): blockIdx.z = parity class (oy & 1, ox & 1) of the output
// pixels of this block.  All pixels of a class share the taps that can reach them (1, 2, 2 or 4 of the 9), so no
// MFMA runs on structurally-zero taps: 2.25 taps per pixel on average instead of 9.
// TR: weights as the A operand -> transposed tile (lane = pixel, float4 epilogue); chosen when an output pixel row is a
// whole number of 128-byte lines (else the 16-byte pieces of neighbouring pixels share lines and the plain form wins)
template <int NT, int VEC, bool PAR, bool TR>
__global__ __launch_bounds__(256) void k_conv2d_f32(const float* __restrict__ src, const float4* __restrict__ wp,
                                                    const float* __restrict__ bias, float* __restrict__ out, ConvGeo g,
                                                    int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float4* s_b = (float4*)smem_raw;  // 2 stages x NT*512
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane & 31, kg = lane >> 5;
  const int py = PAR ? (int)(blockIdx.z >> 1) : 0, px = PAR ? (int)(blockIdx.z & 1) : 0;
  const int CH = PAR ? (g.OH - py + 1) / 2 : g.OH, CW = PAR ? (g.OW - px + 1) / 2 : g.OW;  // class image
  const long M = (long)g.B * CH * CW;
  const long m = (long)blockIdx.x * CG_BM + wv * 32 + row;
  const bool mok = m < M;
  const long mc = mok ? m : (M > 0 ? M - 1 : 0);
  const int cx = (int)(mc % CW);
  const long t1 = mc / CW;
  const int ox = PAR ? 2 * cx + px : cx, oy = PAR ? 2 * (int)(t1 % CH) + py : (int)(t1 % CH), b = (int)(t1 / CH);
  const int T = g.ksz * g.ksz, pad = g.ksz >> 1, G = (g.K + CG_KG - 1) / CG_KG;
  const int ntx = PAR ? (px ? 2 : 1) : g.ksz, nty = PAR ? (py ? 2 : 1) : g.ksz;
  const int S = (PAR ? ntx * nty : T) * G;
  (void)T;
  const int ntiles = (g.N + 31) >> 5;
  const float4* wblk[NT];  // this block's N tiles (a tile past the end re-reads the last one; never stored)
#pragma unroll
  for (int t = 0; t < NT; ++t) wblk[t] = wp + (long)min((int)blockIdx.y * NT + t, ntiles - 1) * (T * G) * 512;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // unconditional loads from clamped addresses + selects (a load inside a divergent
  // branch makes the compiler drain vmcnt(0) after it)
  auto load_a = [&](int s, float(&a)[32]) {
    const int ti = s / G, cgi = s - ti * G;
    int dy, dx;
    if (PAR) {  // the class's taps: dy = 1 (even rows) or {0, 2} (odd rows), same for dx
      const int iy = ti / ntx, ix = ti - iy * ntx;
      dy = py ? 2 * iy : 1, dx = px ? 2 * ix : 1;
    } else {
      dy = ti / g.ksz, dx = ti - dy * g.ksz;
    }
    int sy, sx;
    bool ok;
    if (g.mode == 0) {
      sy = oy * g.stride + dy - pad;
      sx = ox * g.stride + dx - pad;
      ok = mok && sy >= 0 && sy < g.SH && sx >= 0 && sx < g.SW;
    } else {
      const int ty = oy + pad - dy, tx = ox + pad - dx;
      sy = ty / g.stride;
      sx = tx / g.stride;
      ok = mok && ty >= 0 && tx >= 0 && sy * g.stride == ty && sx * g.stride == tx && sy < g.SH && sx < g.SW;
    }
    sy = min(max(sy, 0), g.SH - 1);
    sx = min(max(sx, 0), g.SW - 1);
    const float* p = src + (((long)b * g.SH + sy) * g.SW + sx) * g.lds;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const int c0 = cgi * CG_KG + ch * 16 + 8 * kg;
      if (VEC == 4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = c0 + 4 * q;
          const bool v = ok && c + 4 <= g.K;
          const float4 t4 = *(const float4*)(p + (c + 4 <= g.K ? c : 0));
          a[ch * 8 + 4 * q + 0] = v ? t4.x : 0.f;
          a[ch * 8 + 4 * q + 1] = v ? t4.y : 0.f;
          a[ch * 8 + 4 * q + 2] = v ? t4.z : 0.f;
          a[ch * 8 + 4 * q + 3] = v ? t4.w : 0.f;
        }
      } else if (VEC == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + 2 * q;
          const bool v = ok && c + 2 <= g.K;
          const float2 t2 = *(const float2*)(p + (c + 2 <= g.K ? c : 0));
          a[ch * 8 + 2 * q + 0] = v ? t2.x : 0.f;
          a[ch * 8 + 2 * q + 1] = v ? t2.y : 0.f;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int c = c0 + q;
          const float t = p[c < g.K ? c : 0];
          a[ch * 8 + q] = (ok && c < g.K) ? t : 0.f;
        }
      }
    }
  };

  // packed-weight stage (tap, 64-channel group) of pipeline stage s
  auto wstage = [&](int s) -> long {
    if (!PAR) return s;
    const int ti = s / G, cgi = s - ti * G, iy = ti / ntx, ix = ti - iy * ntx;
    return (long)((py ? 2 * iy : 1) * 3 + (px ? 2 * ix : 1)) * G + cgi;
  };
  float a_cur[32], a_nxt[32];
  float4 b_reg[2 * NT];
  if (S == 0 || M == 0) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) s_b[t * 512 + tid + 256 * i] = wblk[t][wstage(0) * 512 + tid + 256 * i];
  load_a(0, a_cur);
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    const int sn = min(s + 1, S - 1);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) b_reg[t * 2 + i] = wblk[t][wstage(sn) * 512 + tid + 256 * i];
    load_a(sn, a_nxt);
    const float4* sb = s_b + (s & 1) * (NT * 512);
    // 16-channel chunks of this stage that hold real channels (the last 64-channel group of K = 2, 130, 258 ...
    // is mostly padding): uniform per stage, straight-line code per case
    const int kleft = g.K - (s % G) * CG_KG;
    const int nchunk = kleft >= CG_KG ? 4 : (kleft + 15) >> 4;
    auto chunk = [&](int ch) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 bq = sb[(t * 8 + ch * 2 + h) * 64 + lane];
          acc[t] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(bq.x, a_cur[ch * 8 + 4 * h + 0], acc[t], 0, 0, 0)
                      : __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ch * 8 + 4 * h + 0], bq.x, acc[t], 0, 0, 0);
          acc[t] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(bq.y, a_cur[ch * 8 + 4 * h + 1], acc[t], 0, 0, 0)
                      : __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ch * 8 + 4 * h + 1], bq.y, acc[t], 0, 0, 0);
          acc[t] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(bq.z, a_cur[ch * 8 + 4 * h + 2], acc[t], 0, 0, 0)
                      : __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ch * 8 + 4 * h + 2], bq.z, acc[t], 0, 0, 0);
          acc[t] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(bq.w, a_cur[ch * 8 + 4 * h + 3], acc[t], 0, 0, 0)
                      : __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ch * 8 + 4 * h + 3], bq.w, acc[t], 0, 0, 0);
        }
    };
    chunk(0);
    if (nchunk > 1) {
      chunk(1);
      if (nchunk > 2) {
        chunk(2);
        if (nchunk > 3) chunk(3);
      }
    }
    float4* sbn = s_b + ((s + 1) & 1) * (NT * 512);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) sbn[t * 512 + tid + 256 * i] = b_reg[t * 2 + i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; ++i) a_cur[i] = a_nxt[i];
  }

  if (TR) {
    // epilogue.  The weights are the A operand of the MFMAs, so the tile is TRANSPOSED: this lane owns ONE output pixel and the 16 channels 8q + 4kg .. +3 of every N tile:
    // the stores (and the accumulate loads, issued together before them) are float4s, 4 instead of 16 memory
    // instructions per tile and lane (a dword-per-lane epilogue is bound by the texture addresser).
    float* orow = out + ((long)(b * g.OH + oy) * g.OW + ox) * g.ldo;  // (b, oy, ox): this lane's output pixel (computed above)
    const bool vec = (g.ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0;  // uniform
  #pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n0 = (blockIdx.y * NT + t) * 32 + 4 * kg;  // channels n0 + 8q + e
      float4 oldv[4], bv[4];
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 8 * q, nq = min(n, max(g.N - 4, 0));
        const bool full = n + 4 <= g.N;
        if (vec) {  // (uniform branch; both sides load unconditionally from clamped addresses)
          const float4 o = *(const float4*)(orow + nq);
          oldv[q] = (accumulate && full) ? o : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          float o[4];
  #pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = orow[min(n + e, g.N - 1)];
          oldv[q] = accumulate ? make_float4(o[0], o[1], o[2], o[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float bb[4];
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float bl = (bias ? bias : out)[bias ? min(n + e, g.N - 1) : 0];
          bb[e] = (bias && n + e < g.N) ? bl : 0.f;
        }
        bv[q] = make_float4(bb[0], bb[1], bb[2], bb[3]);
      }
      if (mok) {
  #pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + 8 * q;
          const float v[4] = {(acc[t][4 * q + 0] + bv[q].x) + oldv[q].x, (acc[t][4 * q + 1] + bv[q].y) + oldv[q].y,
                              (acc[t][4 * q + 2] + bv[q].z) + oldv[q].z, (acc[t][4 * q + 3] + bv[q].w) + oldv[q].w};
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
    return;
  }
  // epilogue.  Output pixel of each accumulator row once (PAR needs divisions); when accumulating, all old values
  // are read first from clamped addresses in one straight-line block (a load under `if (mr < M)` is followed by its
  // own s_waitcnt vmcnt(0): 16 NT serial round trips)
  int pixr[16];  // pixel index (B*OH*OW < 2^31), -1 = past the end
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long mr = (long)blockIdx.x * CG_BM + wv * 32 + cg_row(r, lane);
    int pix = (int)mr;
    if (PAR) {
      const int mi = (int)(mr < M ? mr : M - 1), rcx = mi % CW, rt = mi / CW;
      pix = ((rt / CH) * g.OH + 2 * (rt % CH) + py) * g.OW + 2 * rcx + px;
    }
    pixr[r] = mr < M ? pix : -1;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = (blockIdx.y * NT + t) * 32 + row, nq = min(n, g.N - 1);
    const float bv = (bias && n < g.N) ? bias[n] : 0.f;
    float oldv[16];
    if (accumulate) {  // uniform
#pragma unroll
      for (int r = 0; r < 16; ++r) oldv[r] = out[(long)max(pixr[r], 0) * g.ldo + nq];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) oldv[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (pixr[r] >= 0 && n < g.N) out[(long)pixr[r] * g.ldo + n] = (acc[t][r] + bv) + oldv[r];
  }
}

template <int NT, bool PAR>
static int cg_launch_vec(const float* src, const float* wp, const float* bias, float* out, const ConvGeo& g, int accumulate,
                         hipStream_t st) {
  // PAR: the largest parity class (even rows, even columns) sizes the grid; blocks past a smaller class exit
  const long M = PAR ? (long)g.B * ((g.OH + 1) / 2) * ((g.OW + 1) / 2) : (long)g.B * g.OH * g.OW;
  dim3 grid(evf_cdiv(M, CG_BM), evf_cdiv(g.N, 32 * NT), PAR ? 4 : 1), block(256);
  const size_t smem = 2 * NT * 512 * sizeof(float4);
  const bool a16 = ((uintptr_t)src & 15) == 0, a8 = ((uintptr_t)src & 7) == 0;
  const bool tr = (g.ldo % 32) == 0 && (((uintptr_t)out) & 127) == 0;
#define CG_GO(V_)                                                                                                           \
  do {                                                                                                                      \
    if (tr)                                                                                                                 \
      hipLaunchKernelGGL((k_conv2d_f32<NT, V_, PAR, true>), grid, block, smem, st, src, (const float4*)wp, bias, out, g,    \
                         accumulate);                                                                                       \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_conv2d_f32<NT, V_, PAR, false>), grid, block, smem, st, src, (const float4*)wp, bias, out, g,   \
                         accumulate);                                                                                       \
  } while (0)
  if (g.K % 4 == 0 && g.lds % 4 == 0 && a16)
    CG_GO(4);
  else if (g.K % 2 == 0 && g.lds % 2 == 0 && a8)
    CG_GO(2);
  else
    CG_GO(1);
#undef CG_GO
  return evf_status();
}

static int cg_launch(const float* src, const float* wp, const float* bias, float* out, const ConvGeo& g, int accumulate,
                     void* stream) {
  // two N tiles per wave halve the A traffic, but only pay when the grid still fills the 256 CUs twice over
  const long mblocks = evf_cdiv((long)g.B * g.OH * g.OW, CG_BM), ntiles = evf_cdiv(g.N, 32);
  const bool two = ntiles >= 2 && mblocks * ((ntiles + 1) / 2) >= 512;
  if (g.mode == 1 && g.stride == 2 && g.ksz == 3) {
    if (two) return cg_launch_vec<2, true>(src, wp, bias, out, g, accumulate, EVF_STREAM(stream));
    return cg_launch_vec<1, true>(src, wp, bias, out, g, accumulate, EVF_STREAM(stream));
  }
  if (two) return cg_launch_vec<2, false>(src, wp, bias, out, g, accumulate, EVF_STREAM(stream));
  return cg_launch_vec<1, false>(src, wp, bias, out, g, accumulate, EVF_STREAM(stream));
}

static inline int cg_out_dim(int n, int ksz, int stride) { return (n + 2 * (ksz >> 1) - ksz) / stride + 1; }

extern "C" int evf_conv2d_fwd(const float* x, int ldx, const float* w_packed, const float* bias, float* y, int ldy, int B,
                              int H, int W, int Cin, int Cout, int ksz, int stride, int accumulate, void* stream) {
  if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !EVF_KSZ_OK(ksz) ||
      (stride != 1 && stride != 2) || ldx < Cin || ldy < Cout)
    return EVF_EINVAL;
  ConvGeo g;
  g.B = B, g.SH = H, g.SW = W, g.K = Cin;
  g.OH = cg_out_dim(H, ksz, stride), g.OW = cg_out_dim(W, ksz, stride), g.N = Cout;
  g.ksz = ksz, g.stride = stride, g.mode = 0, g.lds = ldx, g.ldo = ldy;
  return cg_launch(x, w_packed, bias, y, g, accumulate, stream);
}

// g_x [B,H,W,Cin] (+)= conv^T(g_y [B,Ho,Wo,Cout]); wT_packed from evf_pack_conv2d_weight(transpose = 1)
extern "C" int evf_conv2d_dgrad(const float* g_y, int ldg, const float* wT_packed, float* g_x, int ldx, int B, int H, int W,
                                int Cin, int Cout, int ksz, int stride, int accumulate, void* stream) {
  if (!g_y || !wT_packed || !g_x || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !EVF_KSZ_OK(ksz) ||
      (stride != 1 && stride != 2) || ldx < Cin || ldg < Cout)
    return EVF_EINVAL;
  ConvGeo g;
  g.B = B, g.SH = cg_out_dim(H, ksz, stride), g.SW = cg_out_dim(W, ksz, stride), g.K = Cout;
  g.OH = H, g.OW = W, g.N = Cin;
  g.ksz = ksz, g.stride = stride, g.mode = 1, g.lds = ldg, g.ldo = ldx;
  return cg_launch(g_y, wT_packed, nullptr, g_x, g, accumulate, stream);
}
