// General 2-D convolution on the bf16 matrix cores with EXACT operand splits ("bf16x3", as the fused 32 -> 32 FireNet kernels):
// the same implicit GEMM, tiling, addressing and epilogues as k_conv2d_f32 (evf_conv_gen.hip -- read its header), but
//   * the weights are packed as three bf16 planes w = hi + mid + lo (exact: 8 + 8 + 8 mantissa bits),
//   * the activation fragment (8 consecutive channels of the lane's pixel) is converted on the fly to its bf16 head plus
//     the exact fp32 residual.  A wave-uniform vote (ballot over the residuals of the wave's 32 pixels x 16 channels)
//     picks the product: all residuals zero -- binary spikes, event counts < 256, residual sums {0,1,2}, bilinear x2
//     blends of those (multiples of 1/16): most of a spiking network's activations -- 3 MFMAs, nothing dropped; else the
//     exact 3-way split and the 6 terms above 2^-24 of the leading one.  The vote never changes a result (the 6-term
//     product of an exactly representable input adds exact zeros to the 3-term one); it needs no promise from the caller.
//   * products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 or 6 MFMAs of 32 cycles per 16 channels against
//     8 fp32 MFMAs of 64 cycles -- 5.3x / 2.7x fewer matrix cycles at fp32 round-off.
#include <stdlib.h>

#include "evf_common.h"
#include "evf_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b3_bf16x8 __attribute__((ext_vector_type(8)));

#define CG_BM 128  // output pixels per block (4 waves x 32)
#define CG_KG 64   // input channels per stage
#define B3_STAGE (4 * 3 * 64)  // uint4 per (N tile, tap, 64-channel group): [chunk 4][term 3][lane 64]

struct ConvGeo {
  int B, SH, SW, K;  // source image dims, contraction channels
  int OH, OW, N;     // output image dims, output channels
  int ksz, stride, mode;
  int lds, ldo;      // pixel strides (floats) of source / output
};

__device__ __forceinline__ int cg_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------
// weight packing.  w is the torch layout [Cout][Cin][k][k].  transpose = 0: K = Cin, N = Cout (forward); 1: K = Cout,
// N = Cin (input gradient).  dst uint4 index ((((nt*T + tap)*G + g)*4 + ch)*3 + term)*64 + lane, bf16 element e:
//   k = g*64 + ch*16 + 8*(lane>>5) + e,   n = nt*32 + (lane&31)
// ---------------------------------------------------------------------------
__global__ void k_pack_conv2d_b3(const float* __restrict__ w, int Cout, int Cin, int T, int transpose, long total, int cin_total,
                                 int cin_off, uint4* __restrict__ dst) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one (nt, tap, g, ch, lane): all three terms
  if (idx >= total) return;
  const int K = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
  const int G = (K + CG_KG - 1) / CG_KG;
  const int lane = idx & 63;
  long q = idx >> 6;
  const int ch = q & 3;
  q >>= 2;
  const int g = q % G;
  q /= G;
  const int tap = q % T;
  const int nt = q / T;
  const int n = nt * 32 + (lane & 31);
  uint32_t t3[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = g * CG_KG + ch * 16 + 8 * (lane >> 5) + 2 * e + j;
      float x = 0.f;
      if (k < K && n < N) {
        const int co = transpose ? k : n, ci = transpose ? n : k;
        if (cin_off + ci < cin_total) x = w[((long)co * cin_total + cin_off + ci) * T + tap];
      }
      v[j] = x;
    }
    evf_split3_pair(v[0], v[1], t3[0][e], t3[1][e], t3[2][e]);
  }
  const long base = ((((long)nt * T + tap) * G + g) * 4 + ch) * 3;
#pragma unroll
  for (int s = 0; s < 3; ++s) dst[(base + s) * 64 + lane] = make_uint4(t3[s][0], t3[s][1], t3[s][2], t3[s][3]);
}

static long b3_packed_uint4(int Cout, int Cin, int ksz, int transpose) {
  const int K = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
  return (long)evf_cdiv(N, 32) * ksz * ksz * evf_cdiv(K, CG_KG) * B3_STAGE;
}

extern "C" int64_t evf_conv2d_b3_packed_size(int Cout, int Cin, int ksz, int transpose) {
  if (Cout <= 0 || Cin <= 0 || !EVF_KSZ_OK(ksz)) return 0;
  return b3_packed_uint4(Cout, Cin, ksz, transpose) * 4;  // in floats (4 bytes), like evf_conv2d_packed_size
}

// All packed operands of a network in ONE launch (after an optimizer step every weight changed: 2 packs per layer were
// ~45 launches of ~5 us each in the LIF-EV-FlowNet step).
#define B3_MULTI 48
struct B3PackMulti {
  const float* w[B3_MULTI];
  uint4* dst[B3_MULTI];
  int Cout[B3_MULTI], Cin[B3_MULTI], T[B3_MULTI], tr[B3_MULTI], cin_total[B3_MULTI], cin_off[B3_MULTI];
  int blk0[B3_MULTI + 1];  // first block of each tensor
  int n;
};
// One block = one (N tile of 32, 64-channel K group) of one tensor, all taps: the 32 x 64 x T source weights are read in whole
// contiguous runs of the torch layout [co][ci][tap] (64 ci x T floats per co, or 32 ci x T per co when transposed) into LDS, and the
// fragments are built from there.  (A thread per fragment reading its eight weights straight from memory touched eight different
// lines 36 bytes .. 18 KB apart: 130 us for the 20 M weights of LIF-EV-FlowNet in both layouts, every step.)
#define B3P_MAXT 9  // taps of the LDS path (3x3 and 1x1); larger kernels take the per-fragment loads
__global__ __launch_bounds__(256) void k_pack_conv2d_b3_multi(B3PackMulti m) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [32 n][64 k x T + 1] (72 KiB for 3x3)
  int k = 0;
  while (k + 1 < m.n && (int)blockIdx.x >= m.blk0[k + 1]) ++k;  // (uniform; n <= 48)
  const int Cout = m.Cout[k], Cin = m.Cin[k], T = m.T[k], transpose = m.tr[k], cin_total = m.cin_total[k], cin_off = m.cin_off[k];
  const float* __restrict__ w = m.w[k];
  uint4* __restrict__ dst = m.dst[k];
  const int K = transpose ? Cout : Cin;
  const int G = (K + CG_KG - 1) / CG_KG;
  const int rel = (int)blockIdx.x - m.blk0[k], g = rel % G, nt = rel / G;
  const int tid = threadIdx.x;
  // ---- source tile -> LDS as s_w[n_l * NP + k_l * T + tap], NP = 64 T + 1 (odd: the 32 rows a fragment pass reads lie on 32 banks)
  const int NP = 64 * T + 1;
  if (!transpose) {  // n = co, k = ci: per co a run of 64 ci x T floats
    const int run = 64 * T;
    for (int r = tid; r < run; r += 256) {  // (the divisions once per column, not per element)
      const int k_l = r / T, ci = g * 64 + k_l;
      const bool cok = ci < Cin && cin_off + ci < cin_total;
#pragma unroll 8
      for (int n_l = 0; n_l < 32; ++n_l) {
        const int co = nt * 32 + n_l;
        const bool ok = cok && co < Cout;
        const float v = w[((long)(ok ? co : 0) * cin_total + cin_off + (ok ? g * 64 : 0)) * T + (ok ? r : 0)];
        s_w[n_l * NP + r] = ok ? v : 0.f;
      }
    }
  } else {  // n = ci, k = co: per co a run of 32 ci x T floats
    const int run = 32 * T;
    for (int r = tid; r < run; r += 256) {
      const int n_l = r / T, tap = r - n_l * T, ci = nt * 32 + n_l;
      const bool cok = ci < Cin && cin_off + ci < cin_total;
#pragma unroll 8
      for (int k_l = 0; k_l < 64; ++k_l) {
        const int co = g * 64 + k_l;
        const bool ok = cok && co < Cout;
        const float v = w[((long)(ok ? co : 0) * cin_total + cin_off + (ok ? nt * 32 : 0)) * T + (ok ? r : 0)];
        s_w[n_l * NP + k_l * T + tap] = ok ? v : 0.f;
      }
    }
  }
  __syncthreads();
  // ---- fragments: (tap, chunk, lane) -> three uint4 (hi, mid, lo planes of 8 consecutive k of row n)
  for (int f = tid; f < T * 4 * 64; f += 256) {
    const int lane = f & 63, ch = (f >> 6) & 3, tap = f >> 8;
    const int n_l = lane & 31;
    const float* row = s_w + n_l * NP + (ch * 16 + 8 * (lane >> 5)) * T + tap;
    uint32_t t3[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) evf_split3_pair(row[(2 * e) * T], row[(2 * e + 1) * T], t3[0][e], t3[1][e], t3[2][e]);
    const long base = ((((long)nt * T + tap) * G + g) * 4 + ch) * 3;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) dst[(base + s3) * 64 + lane] = make_uint4(t3[s3][0], t3[s3][1], t3[s3][2], t3[s3][3]);
  }
}

// meta: 6 ints per tensor (Cout, Cin, ksz, transpose, cin_total, cin_off) -- the arguments of evf_pack_conv2d_weight_b3
extern "C" int evf_pack_conv2d_weight_b3(const float* w, int Cout, int Cin, int ksz, int transpose, int cin_total, int cin_off,
                                         void* dst, void* stream);
extern "C" int evf_pack_conv2d_weights_b3_multi(const void* const* w, void* const* dst, const int* meta, int n, void* stream) {
  if (!w || !dst || !meta || n <= 0) return EVF_EINVAL;
  static bool once = false;
  const size_t smem = sizeof(float) * 32 * (64 * B3P_MAXT + 1);
  if (!once) {
    (void)hipFuncSetAttribute((const void*)k_pack_conv2d_b3_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    once = true;
  }
  int i = 0;
  while (i < n) {
    B3PackMulti m;
    m.n = 0;
    int blk = 0;
    for (; i < n && m.n < B3_MULTI; ++i) {
      const int* t = meta + 6 * i;
      if (!w[i] || !dst[i] || t[0] <= 0 || t[1] <= 0 || !EVF_KSZ_OK(t[2]) || t[5] < 0 || t[5] >= t[4]) return EVF_EINVAL;
      if (t[2] * t[2] > B3P_MAXT) {  // 5x5 / 7x7: the single-tensor kernel (per-fragment loads)
        const int rc = evf_pack_conv2d_weight_b3((const float*)w[i], t[0], t[1], t[2], t[3], t[4], t[5], dst[i], stream);
        if (rc) return rc;
        continue;
      }
      const int k = m.n++;
      m.w[k] = (const float*)w[i], m.dst[k] = (uint4*)dst[i];
      m.Cout[k] = t[0], m.Cin[k] = t[1], m.T[k] = t[2] * t[2], m.tr[k] = t[3], m.cin_total[k] = t[4], m.cin_off[k] = t[5];
      m.blk0[k] = blk;
      const int Kk = t[3] ? t[0] : t[1], Nn = t[3] ? t[1] : t[0];
      blk += evf_cdiv(Nn, 32) * evf_cdiv(Kk, CG_KG);  // one block per (N tile, 64-channel group)
    }
    m.blk0[m.n] = blk;
    if (m.n > 0) hipLaunchKernelGGL(k_pack_conv2d_b3_multi, dim3(blk), dim3(256), smem, EVF_STREAM(stream), m);
  }
  return evf_status();
}

extern "C" int evf_pack_conv2d_weight_b3(const float* w, int Cout, int Cin, int ksz, int transpose, int cin_total, int cin_off,
                                         void* dst, void* stream) {
  if (!w || !dst || Cout <= 0 || Cin <= 0 || !EVF_KSZ_OK(ksz) || cin_off < 0 || cin_off >= cin_total) return EVF_EINVAL;
  const long total = b3_packed_uint4(Cout, Cin, ksz, transpose) / 3;
  hipLaunchKernelGGL(k_pack_conv2d_b3, dim3(evf_cdiv(total, 256)), dim3(256), 0, EVF_STREAM(stream), w, Cout, Cin, ksz * ksz,
                     transpose, total, cin_total, cin_off, (uint4*)dst);
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
__global__ __launch_bounds__(256) void k_conv2d_b3(const float* __restrict__ src, const uint4* __restrict__ wp,
                                                    const float* __restrict__ bias, float* __restrict__ out, ConvGeo g,
                                                    int accumulate, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_b = (uint4*)smem_raw;  // 2 stages x NT x [4 chunks][3 terms][64 lanes] 16-byte fragments
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
  const uint4* wblk[NT];  // this block's N tiles (a tile past the end re-reads the last one; never stored)
#pragma unroll
  for (int t = 0; t < NT; ++t) wblk[t] = wp + (long)min((int)blockIdx.y * NT + t, ntiles - 1) * (T * G) * B3_STAGE;

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
  uint4 b_reg[3 * NT];
  if (S == 0 || M == 0) return;
  // split-K (not PAR): blockIdx.z owns the stages [s_lo, s_hi) and writes its partial sums to its own slab
  // out + z * M * ldo (the launcher points `out` at the scratch slabs and sums them with k_b3_reduce)
  int s_lo = 0, s_hi = S;
  if (!PAR && nsplit > 1) {
    const int per = (S + nsplit - 1) / nsplit;
    s_lo = min((int)blockIdx.z * per, S), s_hi = min(s_lo + per, S);
    out += (long)blockIdx.z * M * g.ldo;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      s_b[t * B3_STAGE + tid + 256 * i] = wblk[t][wstage(min(s_lo, S - 1)) * B3_STAGE + tid + 256 * i];
  load_a(min(s_lo, S - 1), a_cur);
  __syncthreads();

#pragma unroll 1
  for (int s = s_lo; s < s_hi; ++s) {
    const int sn = min(s + 1, s_hi - 1);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 3; ++i) b_reg[t * 3 + i] = wblk[t][wstage(sn) * B3_STAGE + tid + 256 * i];
    load_a(sn, a_nxt);
    const uint4* sb = s_b + ((s - s_lo) & 1) * (NT * B3_STAGE);
    // 16-channel chunks of this stage that hold real channels (the last 64-channel group of K = 2, 130, 258 ...
    // is mostly padding): uniform per stage, straight-line code per case
    const int kleft = g.K - (s % G) * CG_KG;
    const int nchunk = kleft >= CG_KG ? 4 : (kleft + 15) >> 4;
    auto chunk = [&](int ch) {
      // this lane's 8 consecutive channels of the chunk (k = 8 kg + e): bf16 head + exact fp32 residual
      uint32_t xh[4];
      float r[8];
      uint32_t nz = 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = a_cur[ch * 8 + 2 * e], a1 = a_cur[ch * 8 + 2 * e + 1];
        xh[e] = evf_pk_bf16(a0, a1);
        r[2 * e] = a0 - __uint_as_float(xh[e] << 16);
        r[2 * e + 1] = a1 - __uint_as_float(xh[e] & 0xFFFF0000u);
        nz |= __float_as_uint(r[2 * e]) | __float_as_uint(r[2 * e + 1]);
      }
      const uint4 uh = make_uint4(xh[0], xh[1], xh[2], xh[3]);
      const b3_bf16x8 ah = *(const b3_bf16x8*)&uh;
      // wave-uniform: does ANY of the wave's 32 x 16 activations need more than its bf16 head?
      const bool inexact = __builtin_amdgcn_ballot_w64((nz & 0x7FFFFFFFu) != 0u) != 0ull;
#define B3_MMA(X_, W_) acc[t] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(W_, X_, acc[t], 0, 0, 0) \
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(X_, W_, acc[t], 0, 0, 0)
      if (!inexact) {  // (spikes, counts, {0,1,2} sums, bilinear blends: three products, nothing dropped)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const uint4 q0 = sb[t * B3_STAGE + (ch * 3 + 0) * 64 + lane], q1 = sb[t * B3_STAGE + (ch * 3 + 1) * 64 + lane],
                      q2 = sb[t * B3_STAGE + (ch * 3 + 2) * 64 + lane];
          const b3_bf16x8 wh = *(const b3_bf16x8*)&q0, wm = *(const b3_bf16x8*)&q1, wl = *(const b3_bf16x8*)&q2;
          B3_MMA(ah, wl);  // smallest first
          B3_MMA(ah, wm);
          B3_MMA(ah, wh);
        }
      } else {
        uint32_t xm[4], xl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xm[e] = evf_pk_bf16(r[2 * e], r[2 * e + 1]);
          xl[e] = evf_pk_bf16(r[2 * e] - __uint_as_float(xm[e] << 16), r[2 * e + 1] - __uint_as_float(xm[e] & 0xFFFF0000u));
        }
        const uint4 um = make_uint4(xm[0], xm[1], xm[2], xm[3]), ul = make_uint4(xl[0], xl[1], xl[2], xl[3]);
        const b3_bf16x8 am = *(const b3_bf16x8*)&um, al = *(const b3_bf16x8*)&ul;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const uint4 q0 = sb[t * B3_STAGE + (ch * 3 + 0) * 64 + lane], q1 = sb[t * B3_STAGE + (ch * 3 + 1) * 64 + lane],
                      q2 = sb[t * B3_STAGE + (ch * 3 + 2) * 64 + lane];
          const b3_bf16x8 wh = *(const b3_bf16x8*)&q0, wm = *(const b3_bf16x8*)&q1, wl = *(const b3_bf16x8*)&q2;
          // the six terms above 2^-24 of the leading one (as the 32 -> 32 input-gradient kernel), smallest first; with
          // mid = lo = 0 this adds exact zeros to the three products of the branch above: the branch is an optimisation,
          // never a different result
          B3_MMA(am, wm);
          B3_MMA(ah, wl);
          B3_MMA(al, wh);
          B3_MMA(ah, wm);
          B3_MMA(am, wh);
          B3_MMA(ah, wh);
        }
      }
#undef B3_MMA
    };
    chunk(0);
    if (nchunk > 1) {
      chunk(1);
      if (nchunk > 2) {
        chunk(2);
        if (nchunk > 3) chunk(3);
      }
    }
    uint4* sbn = s_b + ((s - s_lo + 1) & 1) * (NT * B3_STAGE);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 3; ++i) sbn[t * B3_STAGE + tid + 256 * i] = b_reg[t * 3 + i];
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


// out[pix * ldo + n] = (accumulate ? out : 0) + bias[n] + sum_z slab[z][pix][n]   (deterministic split-K reduction)
__global__ __launch_bounds__(256) void k_b3_reduce(const float* __restrict__ slab, int nsplit, long M, int N,
                                                   const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                   int accumulate) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;  // one float4 of (pix, n..n+3); N % 4 == 0
  const int q = N >> 2;
  if (e >= M * q) return;
  const long pix = e / q;
  const int n = (int)(e - pix * q) * 4;
  float4 s = *(const float4*)(slab + pix * N + n);
  for (int z = 1; z < nsplit; ++z) {
    const float4 v = *(const float4*)(slab + ((long)z * M + pix) * N + n);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  if (bias) s.x += bias[n], s.y += bias[n + 1], s.z += bias[n + 2], s.w += bias[n + 3];
  float* o = out + pix * ldo + n;
  if (accumulate) s.x += o[0], s.y += o[1], s.z += o[2], s.w += o[3];
  o[0] = s.x, o[1] = s.y, o[2] = s.z, o[3] = s.w;
}

template <int NT, bool PAR>
static int b3_launch_vec(const float* src, const void* wp, const float* bias, float* out, const ConvGeo& g, int accumulate,
                         hipStream_t st, int nsplit = 1) {
  // PAR: the largest parity class (even rows, even columns) sizes the grid; blocks past a smaller class exit
  const long M = PAR ? (long)g.B * ((g.OH + 1) / 2) * ((g.OW + 1) / 2) : (long)g.B * g.OH * g.OW;
  dim3 grid(evf_cdiv(M, CG_BM), evf_cdiv(g.N, 32 * NT), PAR ? 4 : nsplit), block(256);
  const size_t smem = 2 * NT * B3_STAGE * sizeof(uint4);
  const bool a16 = ((uintptr_t)src & 15) == 0, a8 = ((uintptr_t)src & 7) == 0;
  const bool tr = (g.ldo % 32) == 0 && (((uintptr_t)out) & 127) == 0;
#define CG_GO(V_)                                                                                                           \
  do {                                                                                                                      \
    if (tr)                                                                                                                 \
      hipLaunchKernelGGL((k_conv2d_b3<NT, V_, PAR, true>), grid, block, smem, st, src, (const uint4*)wp, bias, out, g, \
                         accumulate, nsplit);                                                                               \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_conv2d_b3<NT, V_, PAR, false>), grid, block, smem, st, src, (const uint4*)wp, bias, out, g, \
                         accumulate, nsplit);                                                                               \
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

// -1 by shape (default; EVF_CONV_TILE=0 / 2 at load time), 0 never, 2 whenever the operands are aligned for it (tests)
static int g_tile_mode = -1;
static int b3_tile_mode() {
  if (g_tile_mode < 0) {
    const char* e = getenv("EVF_CONV_TILE");
    g_tile_mode = e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : 1;
  }
  return g_tile_mode;
}
extern "C" int evf_conv_tile_select(int mode) {
  if (mode < -1 || mode > 2) return EVF_EINVAL;
  g_tile_mode = mode;
  return EVF_OK;
}

static int g_split_force = -1;  // EVF_CONV_SPLIT=n: force n K splits wherever the scratch allows (0 = by shape)
static int b3_split_force() {
  if (g_split_force < 0) {
    const char* e = getenv("EVF_CONV_SPLIT");
    g_split_force = e ? atoi(e) : 0;
  }
  return g_split_force;
}

// EVF_CONV_NSTREAM=2 (tests): with the tile family forced (evf_conv_tile_select(2)) take the N-streaming kernel wherever its operands
// allow, instead of the tile kernel
static bool b3_nstream_forced() {
  const char* e = getenv("EVF_CONV_NSTREAM");
  return e && e[0] == '2';
}

extern "C" int evf_conv_split_select(int n) {  // n > 0: force n K splits wherever the scratch allows; 0: by shape
  if (n < 0 || n > 64) return EVF_EINVAL;
  g_split_force = n;
  return EVF_OK;
}

// ws: caller's scratch of ws_floats floats (may be null): split-K slabs of the layers whose output tiles alone cannot
// fill the chip (low-resolution, many-channel layers at small batch)
// parts (may be null): the caller takes the K split's partial sums itself -- when the plan splits, ws [nparts][M][N] is left
// unreduced and *parts = the number of slabs (bias / accumulate must be off); *parts = 0: the result is in `out`
static int b3_launch(const float* src, const void* wp, const float* bias, float* out, const ConvGeo& g, int accumulate,
                     float* ws, long ws_floats, void* stream, int* parts = nullptr) {
  hipStream_t st = EVF_STREAM(stream);
  if (parts) *parts = 0;
  // bit 2: the source is exactly representable in bf16 BY CONSTRUCTION (hip_ops.spike_tag) from channel (flags >> 4) & 31 on
  const bool exact = (accumulate & 4) != 0;
  const int exact_from = (accumulate >> 4) & 31;
  accumulate &= 1;
  const long M = (long)g.B * g.OH * g.OW;
  const long cap = (ws && (g.N & 3) == 0 && (((uintptr_t)ws) & 15) == 0) ? ws_floats / (M * g.N) : 0;  // slabs that fit
  // spike-valued input: the single-plane kernel of evf_conv_b3small.hip (EVF_CONV_SMALL=0: off)
  static const bool small_ok = !(getenv("EVF_CONV_SMALL") && getenv("EVF_CONV_SMALL")[0] == '0');
  if (exact && small_ok && g.ksz == 3 && g.stride == 1 && g.mode == 0 && b3_split_force() == 0 && b3_tile_mode() != 0) {
    const int ks = evf_conv3_b3x_plan(src, g.B, g.OH, g.OW, g.K, g.N, g.lds, exact_from, cap);
    if (ks == 1 && !bias && !accumulate)  // raw sums = the result
      return evf_conv3_b3x_launch(src, g.lds, wp, out, g.ldo, g.B, g.OH, g.OW, g.K, g.N, exact_from, cap, st);
    if (ks >= 1 && cap >= ks) {
      const int rc = evf_conv3_b3x_launch(src, g.lds, wp, ws, g.N, g.B, g.OH, g.OW, g.K, g.N, exact_from, cap, st);
      if (rc != EVF_OK) return rc;
      if (parts) {
        *parts = ks;
        return EVF_OK;
      }
      hipLaunchKernelGGL(k_b3_reduce, dim3(evf_cdiv(M * (g.N >> 2), 256)), dim3(256), 0, st, ws, ks, M, g.N, bias, out, g.ldo,
                         accumulate);
      return evf_status();
    }
  }
  // low-resolution many-channel 3x3 stride-1 layers (images of at most 16 x 16): one block per image x 64 output channels x K
  // split (evf_conv_b3img.hip); EVF_CONV_IMG=0 disables
  if (g.ksz == 3 && g.stride == 1 && b3_tile_mode() != 0) {
    const int ks = evf_conv3_b3i_plan(src, g.B, g.OH, g.OW, g.K, g.N, g.lds, b3_tile_mode() == 2, (int)min(cap, 8L), b3_split_force());
    if (ks == 1)
      return evf_conv3_b3i_launch(src, g.lds, wp, bias, out, g.ldo, g.B, g.OH, g.OW, g.K, g.N, g.mode, accumulate, 1, st);
    if (ks > 1) {
      const int rc = evf_conv3_b3i_launch(src, g.lds, wp, nullptr, ws, g.N, g.B, g.OH, g.OW, g.K, g.N, g.mode, 0, ks, st);
      if (rc != EVF_OK) return rc;
      if (parts) {
        *parts = ks;
        return EVF_OK;
      }
      hipLaunchKernelGGL(k_b3_reduce, dim3(evf_cdiv(M * (g.N >> 2), 256)), dim3(256), 0, st, ws, ks, M, g.N, bias, out, g.ldo,
                         accumulate);
      return evf_status();
    }
  }
  // 3x3 stride-1 layers with three or more 32-channel output tiles per input tile (the decoders' input gradients): the halo staged
  // once per channel group, the output channels streaming past it (evf_conv_b3n.hip); EVF_CONV_NSTREAM=0 disables
  if (g.ksz == 3 && g.stride == 1 && b3_tile_mode() != 0) {
    const int ks = evf_conv3_b3n_plan(src, g.B, g.OH, g.OW, g.K, g.N, g.lds, b3_tile_mode() == 2 && b3_nstream_forced(), (int)min(cap, 8L),
                                      b3_split_force());
    if (ks == 1)
      return evf_conv3_b3n_launch(src, g.lds, wp, bias, out, g.ldo, g.B, g.OH, g.OW, g.K, g.N, g.mode, accumulate, 1, st);
    if (ks > 1) {
      const int rc = evf_conv3_b3n_launch(src, g.lds, wp, nullptr, ws, g.N, g.B, g.OH, g.OW, g.K, g.N, g.mode, 0, ks, st);
      if (rc != EVF_OK) return rc;
      if (parts) {
        *parts = ks;
        return EVF_OK;
      }
      hipLaunchKernelGGL(k_b3_reduce, dim3(evf_cdiv(M * (g.N >> 2), 256)), dim3(256), 0, st, ws, ks, M, g.N, bias, out, g.ldo,
                         accumulate);
      return evf_status();
    }
  }
  // wide high-resolution 3x3 stride-1 layers: the spatially tiled kernel (evf_conv_b3tile.hip); EVF_CONV_TILE=0 disables
  if (g.ksz == 3 && g.stride == 1 && b3_tile_mode() != 0) {
    const int ks = evf_conv3_b3t_plan(src, g.B, g.OH, g.OW, g.K, g.N, g.lds, b3_tile_mode() == 2, (int)min(cap, 8L),
                                      b3_split_force());
    if (ks == 1)
      return evf_conv3_b3t_launch(src, g.lds, wp, bias, out, g.ldo, g.B, g.OH, g.OW, g.K, g.N, g.mode, accumulate, 1, st);
    if (ks > 1) {
      const int rc = evf_conv3_b3t_launch(src, g.lds, wp, nullptr, ws, g.N, g.B, g.OH, g.OW, g.K, g.N, g.mode, 0, ks, st);
      if (rc != EVF_OK) return rc;
      if (parts) {
        *parts = ks;
        return EVF_OK;
      }
      hipLaunchKernelGGL(k_b3_reduce, dim3(evf_cdiv(M * (g.N >> 2), 256)), dim3(256), 0, st, ws, ks, M, g.N, bias, out, g.ldo,
                         accumulate);
      return evf_status();
    }
  }
  const bool par = g.mode == 1 && g.stride == 2 && g.ksz == 3;
  const long mblocks = evf_cdiv(M, CG_BM), ntiles = evf_cdiv(g.N, 32);
  const int S = g.ksz * g.ksz * evf_cdiv(g.K, CG_KG);
  const long maxsplit = par ? 1 : max(1L, min(min(cap, 8L), (long)(S / 8)));
  // two N tiles per wave halve the activation traffic and the conversion work, but only pay when the grid (times the K
  // splits available) still fills the 256 CUs twice over
  const bool two = ntiles >= 2 && mblocks * ((ntiles + 1) / 2) * maxsplit >= 512;
  const long blocks = mblocks * (two ? (ntiles + 1) / 2 : ntiles);
  int nsplit = 1;
  if (blocks < 512) nsplit = (int)min(maxsplit, evf_cdiv(768L, blocks));
  if (b3_split_force() > 0) nsplit = (int)min((long)b3_split_force(), max(1L, min(cap, (long)S)));
  if (par) {
    if (two) return b3_launch_vec<2, true>(src, wp, bias, out, g, accumulate, st);
    return b3_launch_vec<1, true>(src, wp, bias, out, g, accumulate, st);
  }
  if (nsplit > 1) {
    ConvGeo g2 = g;
    g2.ldo = g.N;
    const int rc = two ? b3_launch_vec<2, false>(src, wp, nullptr, ws, g2, 0, st, nsplit)
                       : b3_launch_vec<1, false>(src, wp, nullptr, ws, g2, 0, st, nsplit);
    if (rc != EVF_OK) return rc;
    if (parts) {
      *parts = nsplit;
      return EVF_OK;
    }
    hipLaunchKernelGGL(k_b3_reduce, dim3(evf_cdiv(M * (g.N >> 2), 256)), dim3(256), 0, st, ws, nsplit, M, g.N, bias, out,
                       g.ldo, accumulate);
    return evf_status();
  }
  if (two) return b3_launch_vec<2, false>(src, wp, bias, out, g, accumulate, st);
  return b3_launch_vec<1, false>(src, wp, bias, out, g, accumulate, st);
}

static inline int b3_out_dim(int n, int ksz, int stride) { return (n + 2 * (ksz >> 1) - ksz) / stride + 1; }

// Scratch (floats) that lets the two products below split their contraction when the output tiles alone cannot fill the
// chip: up to 8 slabs of the output.  0 = never needed for this shape.
extern "C" int64_t evf_conv2d_b3_ws(int B, int Ho, int Wo, int Cout) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return 0;
  const long M = (long)B * Ho * Wo;
  if (evf_cdiv(M, 512) * evf_cdiv(Cout, 64) >= 256) return 0;  // (the tiled kernel's blocks fill the chip unsplit)
  return min(8L, max(2L, (1L << 26) / (M * Cout))) * M * Cout;  // at most 256 MiB
}

// y [B,Ho,Wo,Cout] (+)= conv(x [B,H,W,Cin], w) + bias; w_packed from evf_pack_conv2d_weight_b3(transpose = 0).
extern "C" int evf_conv2d_fwd_b3(const float* x, int ldx, const void* w_packed, const float* bias, float* y, int ldy, int B,
                                 int H, int W, int Cin, int Cout, int ksz, int stride, int accumulate, float* ws,
                                 int64_t ws_floats, void* stream) {
  if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !EVF_KSZ_OK(ksz) ||
      (stride != 1 && stride != 2) || ldx < Cin || ldy < Cout)
    return EVF_EINVAL;
  ConvGeo g;
  g.B = B, g.SH = H, g.SW = W, g.K = Cin;
  g.OH = b3_out_dim(H, ksz, stride), g.OW = b3_out_dim(W, ksz, stride), g.N = Cout;
  g.ksz = ksz, g.stride = stride, g.mode = 0, g.lds = ldx, g.ldo = ldy;
  return b3_launch(x, w_packed, bias, y, g, accumulate, ws, ws_floats, stream);
}

// The same product with the K split's partial sums LEFT IN PARTS for a consumer that adds them itself (evf_lif_fwd_parts): no
// bias, no accumulation; flags bit 2 = exact input as above.  *nparts = 0: y holds the result; n > 0: ws holds n slabs [B*Ho*Wo][Cout].
extern "C" int evf_conv2d_fwd_b3_parts(const float* x, int ldx, const void* w_packed, float* y, int ldy, int B, int H, int W, int Cin,
                                       int Cout, int ksz, int stride, int flags, float* ws, int64_t ws_floats, int* nparts,
                                       void* stream) {
  if (!x || !w_packed || !y || !nparts || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !EVF_KSZ_OK(ksz) ||
      (stride != 1 && stride != 2) || ldx < Cin || ldy < Cout)
    return EVF_EINVAL;
  ConvGeo g;
  g.B = B, g.SH = H, g.SW = W, g.K = Cin;
  g.OH = b3_out_dim(H, ksz, stride), g.OW = b3_out_dim(W, ksz, stride), g.N = Cout;
  g.ksz = ksz, g.stride = stride, g.mode = 0, g.lds = ldx, g.ldo = ldy;
  return b3_launch(x, w_packed, nullptr, y, g, flags & (4 | (31 << 4)), ws, ws_floats, stream, nparts);
}

// g_x [B,H,W,Cin] (+)= conv^T(g_y [B,Ho,Wo,Cout]); wT_packed from evf_pack_conv2d_weight_b3(transpose = 1).
extern "C" int evf_conv2d_dgrad_b3(const float* g_y, int ldg, const void* wT_packed, float* g_x, int ldx, int B, int H, int W,
                                   int Cin, int Cout, int ksz, int stride, int accumulate, float* ws, int64_t ws_floats,
                                   void* stream) {
  if (!g_y || !wT_packed || !g_x || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !EVF_KSZ_OK(ksz) ||
      (stride != 1 && stride != 2) || ldx < Cin || ldg < Cout)
    return EVF_EINVAL;
  ConvGeo g;
  g.B = B, g.SH = b3_out_dim(H, ksz, stride), g.SW = b3_out_dim(W, ksz, stride), g.K = Cout;
  g.OH = H, g.OW = W, g.N = Cin;
  g.ksz = ksz, g.stride = stride, g.mode = 1, g.lds = ldg, g.ldo = ldx;
  return b3_launch(g_y, wT_packed, nullptr, g_x, g, accumulate, ws, ws_floats, stream);
}
