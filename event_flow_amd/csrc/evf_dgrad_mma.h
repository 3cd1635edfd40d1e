// Matrix phase shared by the two input-gradient kernels (evf_dgrad_b3.hip, evf_dgrad_ws.hip): one wave = one row of 32
// pixels x 32 input channels, K = 9 taps x 32 output channels, six exact bf16 terms per K step (108 MFMAs).
//   s_w  : split transposed weights, fragment ((tau*2+m)*3+term)*64 + lane
//   pa   : the three gradient planes [term][halo pixel][4 slots of 16 B], chunk c of pixel p in slot c ^ ((p >> 2) & 3)
//   hp0  : halo-pixel index of (row of this wave - 1... i.e. dy = 0, dx = 0) for this lane = (wave_row) * 34 + (lane & 31)
// TWO accumulators, even / odd K groups alternating: consecutive MFMAs never depend on each other, so a wave that has a SIMD's
// matrix pipe to itself (the wave-specialised kernel) issues them back to back instead of waiting out the dependent-accumulator
// latency of a 108-long chain.  Both kernels share this order, so they stay bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float dgm_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 dgm_bf16x8 __attribute__((ext_vector_type(8)));

#define DGM_HW 34

struct DgmFrags {
  dgm_bf16x8 wh, wm, wl, ah, am, al;  // 6 x 16 bytes per lane: one K group = (tap, 16 output channels)
};

// group g = tau * 2 + m
template <bool MASK>
__device__ __forceinline__ void dg_load_frags(DgmFrags& f, int g, const uint4* __restrict__ s_w, const uint4* __restrict__ pa,
                                              int plane, int hp0, int lane, const uint32_t (&msk)[9]) {
  const int kg = lane >> 5;
  const int tau = g >> 1, m = g & 1;
  const int dy = tau / 3, dx = tau - 3 * dy;
  const int hp = hp0 + dy * DGM_HW + dx, sw = (hp >> 2) & 3;
  const uint4* wf = s_w + (g * 3) * 64 + lane;
  const uint4 w0 = wf[0], w1 = wf[64], w2 = wf[128];
  const int slot = hp * 4 + ((2 * m + kg) ^ sw);
  uint4 u0 = pa[slot], u1 = pa[plane + slot], u2 = pa[2 * plane + slot];
  if (MASK) {  // out-of-image taps (the one-phase kernel clamps its halo loads and masks at use)
    const uint32_t k = msk[tau];
    u0.x &= k, u0.y &= k, u0.z &= k, u0.w &= k;
    u1.x &= k, u1.y &= k, u1.z &= k, u1.w &= k;
    u2.x &= k, u2.y &= k, u2.z &= k, u2.w &= k;
  }
  f.wh = *(const dgm_bf16x8*)&w0, f.wm = *(const dgm_bf16x8*)&w1, f.wl = *(const dgm_bf16x8*)&w2;
  f.ah = *(const dgm_bf16x8*)&u0, f.am = *(const dgm_bf16x8*)&u1, f.al = *(const dgm_bf16x8*)&u2;
}

// The operands of group g+1 are requested (6 ds_read_b128) BEFORE the 6 MFMAs of group g and pinned there with a
// scheduling barrier: left alone, hipcc sinks every read next to its use ("ds_read; s_waitcnt lgkmcnt(0); v_mfma"), which a
// wave that owns a SIMD's matrix pipe pays in full -- 46 instead of 32 cycles per MFMA, measured with phase stamps.
// Even groups accumulate in a0, odd ones in a1 (two independent chains).
template <bool MASK>
__device__ __forceinline__ dgm_f32x16 dg_matrix_phase(const uint4* __restrict__ s_w, const uint4* __restrict__ pa, int plane,
                                                      int hp0, int lane, const uint32_t (&msk)[9]) {
  dgm_f32x16 acc[2] = {{0}, {0}};
  DgmFrags fr[2];
  dg_load_frags<MASK>(fr[0], 0, s_w, pa, plane, hp0, lane, msk);
#pragma unroll
  for (int g = 0; g < 18; ++g) {
    if (g + 1 < 18) dg_load_frags<MASK>(fr[(g + 1) & 1], g + 1, s_w, pa, plane, hp0, lane, msk);
    __builtin_amdgcn_sched_barrier(0);
    const DgmFrags& f = fr[g & 1];
    dgm_f32x16 a = acc[g & 1];
    // smallest terms first.  Weights as the A operand, gradient as B: the product comes out TRANSPOSED (lane = pixel, 16
    // channels in groups of four), so the epilogue moves float4s
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wm, f.am, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh, f.al, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wl, f.ah, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh, f.am, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wm, f.ah, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh, f.ah, a, 0, 0, 0);
    acc[g & 1] = a;
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc[0] + acc[1];
}
