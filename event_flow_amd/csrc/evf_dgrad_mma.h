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

// ---------------------------------------------------------------------------------------------------------------------
// Second form of the matrix phase (k_dgrad_diag_ws, evf_dgrad_diag.hip): the same 108 products in the same order per
// accumulator (acc0: the m = 0 groups in tap order, acc1: the m = 1 groups), so the result is bit-identical to
// dg_matrix_phase -- only where the operands come from and when they are requested differ:
//  * DPPX: in the transposed product a lane is a PIXEL, so the gradient fragment of tap (dy, 1) is the fragment of (dy, 0)
//    moved down by one lane (and the fragment of (dy, 2) moved up by one).  Per tap row and K half only the dx = 0 and
//    dx = 2 fragments are read from LDS; the middle one is two DPP moves per dword: `wave_shl:1` of the dx = 0 fragment
//    (right for every lane but 31 and 63, whose neighbour belongs to the other K half / does not exist) patched by
//    `wave_shr:1` of the dx = 2 fragment under row_mask 0xA, bank_mask 0x8 (lanes 28..31 and 60..63).  12 instead of 18
//    gradient reads per tap row: 90 instead of 108 fragment reads per phase.
//  * PF: the weight fragments are requested PF groups ahead (the gradient fragments one tap ROW ahead under DPPX).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dgm_dpp_mid1(uint32_t f0, uint32_t f2) {
  uint32_t r = (uint32_t)__builtin_amdgcn_mov_dpp((int)f0, 0x130, 0xF, 0xF, true);  // wave_shl:1   lane i <- lane i + 1 (lane 63: 0)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)f2, 0x138, 0xA, 0x8, false);   // wave_shr:1   lanes 28..31, 60..63 <- lane i - 1
}
__device__ __forceinline__ uint4 dgm_dpp_mid(const uint4 f0, const uint4 f2) {
  return make_uint4(dgm_dpp_mid1(f0.x, f2.x), dgm_dpp_mid1(f0.y, f2.y), dgm_dpp_mid1(f0.z, f2.z), dgm_dpp_mid1(f0.w, f2.w));
}

struct DgmW {
  uint4 h, m, l;
};
struct DgmG {
  uint4 h, m, l;
};
__device__ __forceinline__ void dgm_load_w(DgmW& w, int g, const uint4* __restrict__ s_w, int lane) {
  const uint4* wf = s_w + (g * 3) * 64 + lane;
  w.h = wf[0], w.m = wf[64], w.l = wf[128];
}
__device__ __forceinline__ void dgm_load_g(DgmG& a, int dy, int dx, int m, const uint4* __restrict__ pa, int plane, int hp0, int lane) {
  const int kg = lane >> 5;
  const int hp = hp0 + dy * DGM_HW + dx, sw = (hp >> 2) & 3;
  const int slot = hp * 4 + ((2 * m + kg) ^ sw);
  a.h = pa[slot], a.m = pa[plane + slot], a.l = pa[2 * plane + slot];
}
// ... with the halo-pixel index given (a tap row of a RING of halo rows has no fixed distance to the next one)
// COLSWZ: the chunk swizzle follows the pixel's COLUMN in its halo row (col = (lane & 31) + dx) instead of its pixel index, so
// that the per-lane part of an address does not depend on the ring row and stays out of the item loop.
template <bool COLSWZ>
__device__ __forceinline__ void dgm_load_g_at(DgmG& a, int hp, int dx, int m, const uint4* __restrict__ pa, int plane, int lane) {
  const int kg = lane >> 5;
  const int sw = COLSWZ ? ((((lane & 31) + dx) >> 2) & 3) : ((hp >> 2) & 3);
  const int slot = hp * 4 + ((2 * m + kg) ^ sw);
  a.h = pa[slot], a.m = pa[plane + slot], a.l = pa[2 * plane + slot];
}
__device__ __forceinline__ dgm_f32x16 dgm_six(dgm_f32x16 a, const DgmW& w, const DgmG& g) {
  const dgm_bf16x8 wh = *(const dgm_bf16x8*)&w.h, wm = *(const dgm_bf16x8*)&w.m, wl = *(const dgm_bf16x8*)&w.l;
  const dgm_bf16x8 ah = *(const dgm_bf16x8*)&g.h, am = *(const dgm_bf16x8*)&g.m, al = *(const dgm_bf16x8*)&g.l;
  a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, a, 0, 0, 0);  // (the order of dg_matrix_phase: smallest terms first)
  a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, a, 0, 0, 0);
  return a;
}

template <int PF, bool DPPX>
__device__ __forceinline__ dgm_f32x16 dg_matrix_phase2(const uint4* __restrict__ s_w, const uint4* __restrict__ pa, int plane,
                                                       int hp0, int lane) {
  static_assert(PF >= 1 && PF <= 3, "weight prefetch distance in K groups");
  dgm_f32x16 acc[2] = {{0}, {0}};
  DgmW w[PF + 1];
#pragma unroll
  for (int p = 0; p < PF; ++p) dgm_load_w(w[p], p, s_w, lane);
  if (DPPX) {
    // r0[m] / r2[m]: the dx = 0 and dx = 2 fragments of the current tap row.  Each set is re-requested for the NEXT tap row
    // right behind its last use (r0[0] is dead once the middle fragment of (dx 1, m 0) is built, ...), three to four groups
    // ahead of its first use there: one set of registers, no second buffer.
    // The DPP moves that build a middle fragment are issued BETWEEN the six MFMAs of the group before it (a wave issues in
    // order: behind the last MFMA of a dependent chain they would leave the matrix pipe idle for ~100 cycles per tap row
    // and K half); scheduling barriers pin that order.
    DgmG r0[2], r2[2], mid;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      dgm_load_g(r0[m], 0, 0, m, pa, plane, hp0, lane);
      dgm_load_g(r2[m], 0, 2, m, pa, plane, hp0, lane);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int g = (dy * 3 + dx) * 2 + m;
          if (g + PF < 18) dgm_load_w(w[(g + PF) % (PF + 1)], g + PF, s_w, lane);
          const DgmG cur = dx == 0 ? r0[m] : (dx == 2 ? r2[m] : mid);
          // the group after this one is (dx, m ^ 1 ...) in the order dx-major: next = (dx + (m == 1), m ^ 1)
          const int ndx = dx + (m == 1 ? 1 : 0), nm = m ^ 1;
          const bool build = ndx == 1;  // the next group multiplies a middle fragment: build it under this group's MFMAs
          if (dy > 0 && dx == 0 && m == 0) dgm_load_g(r2[1], dy, 2, 1, pa, plane, hp0, lane);
          if (dy < 2) {  // next tap row: the set whose last use lies behind us (r0[.] is dead once its middle fragment is built)
            if (dx == 2 && m == 0) dgm_load_g(r0[0], dy + 1, 0, 0, pa, plane, hp0, lane);
            if (dx == 2 && m == 1) {
              dgm_load_g(r0[1], dy + 1, 0, 1, pa, plane, hp0, lane);
              dgm_load_g(r2[0], dy + 1, 2, 0, pa, plane, hp0, lane);
            }
          }
          const dgm_bf16x8 wh = *(const dgm_bf16x8*)&w[g % (PF + 1)].h, wm = *(const dgm_bf16x8*)&w[g % (PF + 1)].m,
                           wl = *(const dgm_bf16x8*)&w[g % (PF + 1)].l;
          const dgm_bf16x8 ah = *(const dgm_bf16x8*)&cur.h, am = *(const dgm_bf16x8*)&cur.m, al = *(const dgm_bf16x8*)&cur.l;
          DgmG nmid = mid;
          dgm_f32x16 a = acc[m];
          __builtin_amdgcn_sched_barrier(0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, a, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (build) nmid.h.x = dgm_dpp_mid1(r0[nm].h.x, r2[nm].h.x), nmid.h.y = dgm_dpp_mid1(r0[nm].h.y, r2[nm].h.y);
          __builtin_amdgcn_sched_barrier(0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, a, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (build) nmid.h.z = dgm_dpp_mid1(r0[nm].h.z, r2[nm].h.z), nmid.h.w = dgm_dpp_mid1(r0[nm].h.w, r2[nm].h.w);
          __builtin_amdgcn_sched_barrier(0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, a, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (build) nmid.m.x = dgm_dpp_mid1(r0[nm].m.x, r2[nm].m.x), nmid.m.y = dgm_dpp_mid1(r0[nm].m.y, r2[nm].m.y);
          __builtin_amdgcn_sched_barrier(0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, a, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (build) nmid.m.z = dgm_dpp_mid1(r0[nm].m.z, r2[nm].m.z), nmid.m.w = dgm_dpp_mid1(r0[nm].m.w, r2[nm].m.w);
          __builtin_amdgcn_sched_barrier(0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, a, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (build) nmid.l.x = dgm_dpp_mid1(r0[nm].l.x, r2[nm].l.x), nmid.l.y = dgm_dpp_mid1(r0[nm].l.y, r2[nm].l.y);
          if (build) nmid.l.z = dgm_dpp_mid1(r0[nm].l.z, r2[nm].l.z), nmid.l.w = dgm_dpp_mid1(r0[nm].l.w, r2[nm].l.w);
          __builtin_amdgcn_sched_barrier(0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, a, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          acc[m] = a;
          mid = nmid;
        }
      }
    }
  } else {
    DgmG a[2];
    dgm_load_g(a[0], 0, 0, 0, pa, plane, hp0, lane);
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      if (g + PF < 18) dgm_load_w(w[(g + PF) % (PF + 1)], g + PF, s_w, lane);
      if (g + 1 < 18) {
        const int tau = (g + 1) >> 1;
        dgm_load_g(a[(g + 1) & 1], tau / 3, tau % 3, (g + 1) & 1, pa, plane, hp0, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[g & 1] = dgm_six(acc[g & 1], w[g % (PF + 1)], a[g & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  return acc[0] + acc[1];
}

// ---------------------------------------------------------------------------------------------------------------------
// Third form (k_dgrad_diag_dma, evf_dgrad_diag.hip: ONE wave per SIMD, operands brought in by LDS-DMA, no producer team).
// Same 108 products, same order per accumulator (bit-identical results), but the two accumulators now ALTERNATE MFMA by
// MFMA: acc0 takes the m = 0 half of a tap, acc1 the m = 1 half, and consecutive matrix instructions never depend on each
// other.  (A 6-long chain on one accumulator costs ~40 instead of 32 cycles per 32x32x16 MFMA when the wave has the SIMD's
// matrix pipe to itself: phase stamps of the two earlier forms, 4.2-4.5 k cycles per 108 MFMAs.)
// `side(slot)`, slot = 0..107, is called behind every MFMA: the caller's other work of the phase (LDS-DMA pieces of the next
// tile, the previous tile's epilogue) is issued there, a few instructions at a time, pinned by scheduling barriers -- a wave
// issues in order, so anything placed behind the last MFMA would leave the matrix pipe idle.
// DPPX as above; the middle fragments of a tap row are built behind the MFMAs of its dx = 0 tap.
// ---------------------------------------------------------------------------------------------------------------------
// hp_row(dy): halo-pixel index of this lane's pixel in tap row dy at dx = 0 (hp0 + dy * 34 for a plain tile; a ring of rows
// wraps).
template <bool DPPX, bool COLSWZ, class Side, class HpRow>
__device__ __forceinline__ dgm_f32x16 dg_matrix_phase3h(const uint4* __restrict__ s_w, const uint4* __restrict__ pa, int plane,
                                                        HpRow&& hp_row, int lane, Side&& side) {
  auto dgm_load_g = [&](DgmG& a, int dy, int dx, int m, const uint4* __restrict__ pa_, int plane_, int, int lane_) {
    dgm_load_g_at<COLSWZ>(a, hp_row(dy) + dx, dx, m, pa_, plane_, lane_);
  };
  const int hp0 = 0;
  (void)hp0;
  dgm_f32x16 acc0 = {0}, acc1 = {0};
  DgmW w[2][2];  // [tap parity][m]
  dgm_load_w(w[0][0], 0, s_w, lane);
  dgm_load_w(w[0][1], 1, s_w, lane);
  DgmG r0[2], r2[2][2], mid[2], gq[2][2];  // DPPX: r0[m], r2[dy parity][m], mid[m];  else gq[tap parity][m]
  if (DPPX) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      dgm_load_g(r0[m], 0, 0, m, pa, plane, hp0, lane);
      dgm_load_g(r2[0][m], 0, 2, m, pa, plane, hp0, lane);
    }
  } else {
    dgm_load_g(gq[0][0], 0, 0, 0, pa, plane, hp0, lane);
    dgm_load_g(gq[0][1], 0, 0, 1, pa, plane, hp0, lane);
  }
#pragma unroll
  for (int tau = 0; tau < 9; ++tau) {
    const int dy = tau / 3, dx = tau - 3 * dy, tp = tau & 1;
    // operands of the next tap (and, DPPX, of the next tap row), requested before this tap's 12 MFMAs
    if (tau + 1 < 9) {
      dgm_load_w(w[tp ^ 1][0], 2 * (tau + 1), s_w, lane);
      dgm_load_w(w[tp ^ 1][1], 2 * (tau + 1) + 1, s_w, lane);
      if (!DPPX) {
        dgm_load_g(gq[tp ^ 1][0], (tau + 1) / 3, (tau + 1) % 3, 0, pa, plane, hp0, lane);
        dgm_load_g(gq[tp ^ 1][1], (tau + 1) / 3, (tau + 1) % 3, 1, pa, plane, hp0, lane);
      } else if (dx == 1 && dy < 2) {  // r0 is dead (its middle fragments are built), r2 of the next row goes to the other set
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          dgm_load_g(r0[m], dy + 1, 0, m, pa, plane, hp0, lane);
          dgm_load_g(r2[(dy + 1) & 1][m], dy + 1, 2, m, pa, plane, hp0, lane);
        }
      }
    }
    const DgmG c0 = DPPX ? (dx == 0 ? r0[0] : (dx == 1 ? mid[0] : r2[dy & 1][0])) : gq[tp][0];
    const DgmG c1 = DPPX ? (dx == 0 ? r0[1] : (dx == 1 ? mid[1] : r2[dy & 1][1])) : gq[tp][1];
    const DgmW &w0 = w[tp][0], &w1 = w[tp][1];
    const dgm_bf16x8 w0h = *(const dgm_bf16x8*)&w0.h, w0m = *(const dgm_bf16x8*)&w0.m, w0l = *(const dgm_bf16x8*)&w0.l;
    const dgm_bf16x8 w1h = *(const dgm_bf16x8*)&w1.h, w1m = *(const dgm_bf16x8*)&w1.m, w1l = *(const dgm_bf16x8*)&w1.l;
    const dgm_bf16x8 a0h = *(const dgm_bf16x8*)&c0.h, a0m = *(const dgm_bf16x8*)&c0.m, a0l = *(const dgm_bf16x8*)&c0.l;
    const dgm_bf16x8 a1h = *(const dgm_bf16x8*)&c1.h, a1m = *(const dgm_bf16x8*)&c1.m, a1l = *(const dgm_bf16x8*)&c1.l;
    DgmG nm0 = mid[0], nm1 = mid[1];
    const bool build = DPPX && dx == 0;
    // one of the 24 dwords of the two middle fragments per call (6 per MFMA gap over the first 4 gaps of each accumulator)
    auto bld = [&](int e) {
      if (!build) return;
      const int m = e / 12, q = e % 12, pl = q / 4, d = q % 4;
      const DgmG &a = r0[m], &b = r2[dy & 1][m];
      DgmG& o = m ? nm1 : nm0;
      const uint4& fa = pl == 0 ? a.h : (pl == 1 ? a.m : a.l);
      const uint4& fb = pl == 0 ? b.h : (pl == 1 ? b.m : b.l);
      uint4& fo = pl == 0 ? o.h : (pl == 1 ? o.m : o.l);
      const uint32_t va = d == 0 ? fa.x : (d == 1 ? fa.y : (d == 2 ? fa.z : fa.w));
      const uint32_t vb = d == 0 ? fb.x : (d == 1 ? fb.y : (d == 2 ? fb.z : fb.w));
      const uint32_t r = dgm_dpp_mid1(va, vb);
      if (d == 0) fo.x = r; else if (d == 1) fo.y = r; else if (d == 2) fo.z = r; else fo.w = r;
    };
#define DGM3_STEP(J, WA0, GA0, WA1, GA1)                                       \
  __builtin_amdgcn_sched_barrier(0);                                           \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA0, GA0, acc0, 0, 0, 0);     \
  __builtin_amdgcn_sched_barrier(0);                                           \
  bld(4 * (J)), bld(4 * (J) + 1);                                              \
  side(tau * 12 + 2 * (J));                                                    \
  __builtin_amdgcn_sched_barrier(0);                                           \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA1, GA1, acc1, 0, 0, 0);     \
  __builtin_amdgcn_sched_barrier(0);                                           \
  bld(4 * (J) + 2), bld(4 * (J) + 3);                                          \
  side(tau * 12 + 2 * (J) + 1);
    DGM3_STEP(0, w0m, a0m, w1m, a1m)  // (the term order of dg_matrix_phase: smallest first)
    DGM3_STEP(1, w0h, a0l, w1h, a1l)
    DGM3_STEP(2, w0l, a0h, w1l, a1h)
    DGM3_STEP(3, w0h, a0m, w1h, a1m)
    DGM3_STEP(4, w0m, a0h, w1m, a1h)
    DGM3_STEP(5, w0h, a0h, w1h, a1h)
#undef DGM3_STEP
    __builtin_amdgcn_sched_barrier(0);
    mid[0] = nm0, mid[1] = nm1;
  }
  return acc0 + acc1;
}

template <bool DPPX, class Side>
__device__ __forceinline__ dgm_f32x16 dg_matrix_phase3(const uint4* __restrict__ s_w, const uint4* __restrict__ pa, int plane,
                                                       int hp0, int lane, Side&& side) {
  return dg_matrix_phase3h<DPPX, false>(s_w, pa, plane, [&](int dy) { return hp0 + dy * DGM_HW; }, lane, side);
}
