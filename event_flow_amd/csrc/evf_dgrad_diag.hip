// The input-gradient cells of one backward index (evf_bwd_defer_*, see evf_bwd_fused.hip) as ONE persistent,
// wave-specialised launch: k_dgrad_diag_ws.
//
//   g_x[pix][ci] = sum_{tap,co} g[pix + tap][co] * Wt[tap][co][ci]      six-term exact bf16 split, evf_dgrad_mma.h
//
// k_dgrad_diag (evf_dgrad_b3.hip) ran the cells with the one-phase-after-the-other block of k_conv_dgrad_b3_lds: request the
// halo, wait, split, 108 MFMAs per wave, store -- ONE block per CU (LDS), so nothing overlapped and a launch of 4 cells took
// as long as 4 launches (0.22 of the HBM roofline, matrix pipe 38 % busy).  The cells of an index are independent, so here
// the launch is a flat list of PRODUCTS (gradient tensor, weight set, output; a recurrent cell contributes two products
// that share the gradient) x 4-row x 32-pixel tiles, cut into one contiguous range per block (256 blocks = 256 CUs):
//
//   waves 0..3  CONSUMERS, one per SIMD: the 108 MFMAs of one 32-pixel row each, operands from LDS; epilogue through a
//               wave-private LDS tile so that the wave stores full 128-byte lines (non-temporal);
//   waves 4..7  PRODUCERS: fetch the fp32 gradient halo two items ahead, do the exact 3-way bf16 split on the VALU and
//               write the three planes into the other half of a double buffer (out-of-image pixels as zeros).
//
// One barrier per item.  A block's range crosses a product boundary at most once or twice; there all eight waves bring
// the next weight set in by LDS-DMA (54 KiB) behind one more barrier.  Consecutive items of a block are horizontally, then
// vertically adjacent tiles of one sample, so the halo rows two tiles share are L2 hits.
// Same accumulation order per output element as the other input-gradient kernels: bit-identical results.
#include "evf_common.h"
#include "evf_dgrad_mma.h"
#include "evf_split.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void wd_lds_void;
typedef __attribute__((address_space(1))) const void wd_glb_void;

#define C32 32
#define WD_ROWS 4
#define WD_HW 34
#define WD_HP ((WD_ROWS + 2) * WD_HW)  // 204 halo pixels
#define WD_NFRAG 54
#define WD_PLANE (WD_HP * 4)  // uint4 per plane
#define WD_BUF (3 * WD_PLANE)
#define WD_ITEMS (WD_HP * 4)
#define WD_NIT ((WD_ITEMS + 255) / 256)
#define WD_SP 36  // floats per pixel of the epilogue staging tile
#define WD_LDS ((size_t)(WD_NFRAG * 64 + 2 * WD_BUF) * sizeof(uint4) + (size_t)WD_ROWS * 32 * WD_SP * 4)

#ifndef WD_PF
#define WD_PF 2
#endif
#ifndef WD_DPPX
#define WD_DPPX 1
#endif

struct WdTile {
  int prod, b, y0, x0;
};

#ifdef WD_STAMPS  // phase stamps (debug build loaded through EVF_LIB): [block < 16][team 2][128] shader-clock values
__device__ unsigned long long wd_stamps[16 * 2 * 128];
extern "C" int evf_debug_wd_stamps(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(wd_stamps), sizeof(wd_stamps))); }
#define WD_STAMP()                                                                                       \
  do {                                                                                                   \
    if (blockIdx.x < 16 && lane == 0 && (wv == 0 || wv == WD_ROWS) && nst < 128)                         \
      wd_stamps[(blockIdx.x * 2 + (wv ? 1 : 0)) * 128 + nst++] = __builtin_readcyclecounter();          \
  } while (0)
#else
#define WD_STAMP() do {} while (0)
#endif

__global__ __launch_bounds__(512) void k_dgrad_diag_ws(EvfDgProds P, int H, int W, int ntx, int nty, unsigned ntiles,
                                                       unsigned total) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_w = (uint4*)smem_raw;     // [54][64]
  uint4* s_a = s_w + WD_NFRAG * 64;  // [2][3][WD_HP][4], chunk c of pixel p in slot c ^ ((p >> 2) & 3)
  float* s_stage = (float*)(s_a + 2 * WD_BUF);  // [4 consumer waves][32 pixels][WD_SP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool producer = wv >= WD_ROWS;
  const int i = lane & 31;
  int nst = 0;
  (void)nst;
  WD_STAMP();

  // this block's range of (product, tile) items
  const unsigned lo = (unsigned)(((unsigned long long)blockIdx.x * total) / gridDim.x);
  const unsigned hi = (unsigned)(((unsigned long long)(blockIdx.x + 1) * total) / gridDim.x);
  const int nitem = (int)(hi - lo);
  if (nitem <= 0) return;
  const float rnt = 1.0f / (float)ntiles, rntx = 1.0f / (float)ntx, rnty = 1.0f / (float)nty;
  auto divmod = [](unsigned n, unsigned d, float rd, unsigned& q, unsigned& r) {  // exact for n < 2^22
    q = (unsigned)((float)n * rd);
    int rr = (int)n - (int)(q * d);
    if (rr < 0) --q, rr += (int)d;
    if (rr >= (int)d) ++q, rr -= (int)d;
    r = (unsigned)rr;
  };
  auto item_of = [&](int k, WdTile& t) -> bool {
    const bool ok = k < nitem;
    const unsigned id = lo + (unsigned)(ok ? k : nitem - 1);  // (a valid item either way: the prefetch past the end is ignored)
    unsigned pr, tl, r, tx, b, ty;
    divmod(id, ntiles, rnt, pr, tl);
    divmod(tl, (unsigned)ntx, rntx, r, tx);
    divmod(r, (unsigned)nty, rnty, b, ty);
    // block-uniform values: keep them in SGPRs (scalar address arithmetic, a scalar branch at the product boundary)
    t.prod = __builtin_amdgcn_readfirstlane((int)pr), t.x0 = __builtin_amdgcn_readfirstlane((int)tx * 32);
    t.b = __builtin_amdgcn_readfirstlane((int)b), t.y0 = __builtin_amdgcn_readfirstlane((int)ty * WD_ROWS);
    return ok;
  };
  auto load_weights8 = [&](const uint4* src) {  // all 8 waves
    for (int u = wv; u < WD_NFRAG; u += 8)
      __builtin_amdgcn_global_load_lds((wd_glb_void*)(src + u * 64 + lane), (wd_lds_void*)(s_w + u * 64), 16, 0, 0);
  };

  WdTile t0, t1, t2;
  bool h0 = item_of(0, t0), h1 = item_of(1, t1), h2;
  int wprod = t0.prod;  // the weight set in LDS
  if (!producer) {
    const uint4* wt = (const uint4*)P.p[wprod].wt;
    for (int u = wv; u < WD_NFRAG; u += WD_ROWS)
      __builtin_amdgcn_global_load_lds((wd_glb_void*)(wt + u * 64 + lane), (wd_lds_void*)(s_w + u * 64), 16, 0, 0);
  } else {
    __builtin_amdgcn_s_setprio(3);  // the producers' few hundred instructions per item go first (see evf_dgrad_ws.hip)
  }

  struct Regs {
    float4 lo4[WD_NIT], hi4[WD_NIT];
  };
  const int ptid = tid - WD_ROWS * 64;  // 0..255 among the producers
  int ihr[WD_NIT], ihc[WD_NIT];
#pragma unroll
  for (int n = 0; n < WD_NIT; ++n) {
    const int p = min(ptid + n * 256, WD_ITEMS - 1) >> 2;
    ihr[n] = p / WD_HW, ihc[n] = p - ihr[n] * WD_HW;
  }
  auto fetch = [&](Regs& r, const WdTile& t) {
    const float4* gf = (const float4*)P.p[t.prod].g;
#pragma unroll
    for (int n = 0; n < WD_NIT; ++n) {
      const int c = (ptid + n * 256) & 3;
      const int yr = t.y0 - 1 + ihr[n], xr = t.x0 - 1 + ihc[n];
      const int yy = min(max(yr, 0), H - 1), xx = min(max(xr, 0), W - 1);  // loads stay unconditional; zeroed in split_store
      const float4* src = gf + (((long)t.b * H + yy) * W + xx) * 8 + 2 * c;
      r.lo4[n] = src[0], r.hi4[n] = src[1];
    }
  };
  auto split_store = [&](const Regs& r, int buf, const WdTile& t) {
    uint4* dst = s_a + buf * WD_BUF;
#pragma unroll
    for (int n = 0; n < WD_NIT; ++n) {
      const int it = ptid + n * 256;
      if (it < WD_ITEMS) {
        const int p = it >> 2, c = it & 3;
        const int yr = t.y0 - 1 + ihr[n], xr = t.x0 - 1 + ihc[n];
        const bool pin = yr >= 0 && yr < H && xr >= 0 && xr < W;  // out-of-image halo pixels are zeros
        const float v[8] = {r.lo4[n].x, r.lo4[n].y, r.lo4[n].z, r.lo4[n].w, r.hi4[n].x, r.hi4[n].y, r.hi4[n].z, r.hi4[n].w};
        uint32_t t3[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e)  // g = hi + mid + lo, two channels per step (evf_split.h)
          evf_split3_pair(pin ? v[2 * e] : 0.f, pin ? v[2 * e + 1] : 0.f, t3[0][e], t3[1][e], t3[2][e]);
        const int slot = p * 4 + (c ^ ((p >> 2) & 3));
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) dst[sp * WD_PLANE + slot] = make_uint4(t3[sp][0], t3[sp][1], t3[sp][2], t3[sp][3]);
      }
    }
  };

  // The two teams run SEPARATE loops with the same barrier sequence (one per item, one more where the range crosses into
  // the next product): in one shared loop both teams' registers would be live everywhere (the producers' two fetch sets, the
  // consumers' operand pipeline) and the kernel spills.
  auto weight_switch = [&](const WdTile& cur) {
    if (cur.prod != wprod) {  // (block-uniform) every wave is past the previous item's barrier, i.e. done with the old set
      wprod = cur.prod;
      load_weights8((const uint4*)P.p[wprod].wt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  if (producer) {
    // item k: request item k+2 into `rq`, split item k+1 (requested one step ago into `rs`) into buffer (k+1) & 1
    auto pstep = [&](int k, const WdTile& cur, const WdTile& n1, bool have1, const WdTile& n2, bool have2, Regs& rq, const Regs& rs) {
      WD_STAMP();
      weight_switch(cur);
      fetch(rq, have2 ? n2 : cur);  // (unconditional: a conditional fetch makes the register set a phi and the loads synchronous)
      WD_STAMP();
      if (have1) split_store(rs, (k & 1) ^ 1, n1);
      WD_STAMP();
      __syncthreads();  // planes of item k+1 complete; planes of item k free
    };
    Regs ra, rb;
    fetch(ra, t0);
    split_store(ra, 0, t0);
    fetch(rb, h1 ? t1 : t0);
    WD_STAMP();
    __syncthreads();  // weights and the planes of item 0 are in LDS (the requests for item 1 stay in flight)
    for (int k = 0; h0; k += 2) {  // unrolled by two: the register sets alternate without moves
      h2 = item_of(k + 2, t2);
      pstep(k, t0, t1, h1, t2, h2, ra, rb);
      if (!h1) break;
      h0 = item_of(k + 3, t0);
      pstep(k + 1, t1, t2, h2, t0, h0, rb, ra);
      t1 = t0;
      h1 = h0;
      t0 = t2;
      h0 = h2;
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight DMA of this wave has landed
    WD_STAMP();
    __syncthreads();
    float* st = s_stage + wv * (32 * WD_SP);
    const int kg = lane >> 5;
    for (int k = 0; k < nitem; ++k) {
      WdTile cur;
      item_of(k, cur);
      WD_STAMP();
      weight_switch(cur);
      const int y = cur.y0 + wv;
      const f32x16 acc = dg_matrix_phase2<WD_PF, WD_DPPX != 0>(s_w, s_a + (k & 1) * WD_BUF, WD_PLANE, wv * WD_HW + i, lane);
      WD_STAMP();
      // epilogue through the wave-private LDS tile [32 pixels][32 channels] (144-byte pixel pitch): the MFMA layout gives a
      // lane 4 x 16 bytes of its pixel's 128-byte line; read back as 8 pixels x 128 bytes per instruction the wave stores
      // FULL lines, non-temporal
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(float4*)(st + i * WD_SP + 8 * q + 4 * kg) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float* dstp = P.p[cur.prod].gx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
        const float4 v = *(const float4*)(st + p * WD_SP + c4);
        if (y < H && cur.x0 + p < W) evf_store_nt(dstp + (((long)cur.b * H + y) * W + cur.x0 + p) * C32 + c4, v);
      }
      __builtin_amdgcn_wave_barrier();
      WD_STAMP();
      __syncthreads();
    }
  }
  WD_STAMP();
}

// ---------------------------------------------------------------------------------------------------------------------
// k_dgrad_diag_dma: the same flat (product, tile) list, gradient PRE-SPLIT in HBM.
//
// What the stamps of k_dgrad_diag_ws said (6 products, 24 items per block; cycles per item): consumers 4.4 k matrix phase +
// 1.3 k epilogue + 0.7 k barrier, producers 1-2 k fetch issue + 4-4.5 k split: the producer wave of a SIMD needs ~550
// instructions per item for addresses and the exact split, the consumer wave ~350, and a SIMD issues ONE vector instruction
// per 4 cycles -- the two teams take each other's issue slots, and the matrix pipe idles during the epilogue and the barrier.
// The split is free in the kernel that makes g_cur (evf_lif_bwd_wgrad has it in registers for its own weight gradient and
// can write the three bf16 planes, `g_split`): with the planes in HBM this kernel needs no VALU work to stage a tile --
// the halo comes in by LDS-DMA (global_load_lds_dwordx4, no VGPRs, source address per lane: out-of-image pixels read a
// page of zeros), issued by the four waves themselves BETWEEN their MFMAs.
//
//   4 waves (one per SIMD, up to 512 VGPRs each), wave w = row w of the 4-row x 32-pixel tile;
//   item k: [behind the MFMAs: DMA pieces of item k+1 into the other buffer; epilogue of item k-1 from registers]
//           108 MFMAs, the two accumulators alternating (dg_matrix_phase3) -> s_waitcnt vmcnt(0) -> ONE barrier.
// ---------------------------------------------------------------------------------------------------------------------
#define WM_UPP 13                  // 16-pixel DMA units per plane (208 pixel slots, 204 used)
#define WM_PLANE (WM_UPP * 64)     // uint4 per plane
#define WM_BUF (3 * WM_PLANE)      // uint4 per halo buffer
#define WM_NU (3 * WM_UPP)         // DMA pieces per item
#define WM_NJ ((WM_NU + 3) / 4)    // pieces per wave
#define WM_LDS ((size_t)(WD_NFRAG * 64 + 2 * WM_BUF) * sizeof(uint4) + (size_t)4 * 32 * WD_SP * 4)
#ifndef WM_STAGE
#define WM_STAGE 0  // 1: the loader team stages a tile through registers instead of by LDS-DMA pieces (measured: slower, see load_piece)
#endif
#ifndef WM_PIPE
#define WM_PIPE 1  // the epilogue of item k-1 behind the MFMAs of item k (0: after the item's own matrix phase)
#endif

__device__ uint4 wm_zero_page[16];  // 256 bytes of zeros: the DMA source of out-of-image halo pixels

#ifdef WD_STAMPS
__device__ unsigned long long wm_stamps[16 * 4 * 128];
extern "C" int evf_debug_wm_stamps(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(wm_stamps), sizeof(wm_stamps))); }
#define WM_STAMP()                                                                              \
  do {                                                                                          \
    if (blockIdx.x < 16 && lane == 0 && (wv & 3) == 0 && nst < 128)                             \
      wm_stamps[(blockIdx.x * 4 + (wv >> 2)) * 128 + nst++] = __builtin_readcyclecounter();     \
  } while (0)
#else
#define WM_STAMP() do {} while (0)
#endif

struct WmTile {
  int prod, b, y0, x0;
  const char* g;  // the product's gradient planes / output: scalar loads at item_of time, outside the matrix phase (a load
  float* gx;      // from the argument table behind an MFMA would wait for lgkmcnt(0), i.e. for every operand read in flight)
  const float* gP;     // PLIFT: the product's raw dL/d(pooled activity) map (NULL: no trace term) ...
  const uint32_t* xb;  // ... and input spike words
};

// PLIFT: products may carry the PLIF trace term (EvfDgProd.gP / .xb): the matrix wave requests the nine dL/dP values and the spike
// word of its pixel a few MFMAs before it stores a line group and adds AvgPool3x3^T(gP) / 32 to the channels whose input spike is
// set -- the expression of k_conv_dgrad_ws<.., PLIF, ..> (acc + term): the same bits.
template <bool FULL, bool PLIFT = false>  // FULL: H % 4 == 0 and W % 32 == 0 -- every output pixel of every tile exists, the stores need no test
__global__ __launch_bounds__(512) void k_dgrad_diag_dma(EvfDgProds P, unsigned plane_bytes, int H, int W, int ntx, int nty,
                                                        unsigned ntiles, unsigned total) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_w = (uint4*)smem_raw;     // [54][64]
  uint4* s_a = s_w + WD_NFRAG * 64;  // [2][3][WM_UPP * 16 pixels][4], chunk c of pixel p in slot c ^ ((p >> 2) & 3)
  float* s_stage = (float*)(s_a + 2 * WM_BUF);  // [4 waves][32 pixels][WD_SP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 31, kg = lane >> 5;
  int nst = 0;
  (void)nst;
  WM_STAMP();
  const unsigned lo = (unsigned)(((unsigned long long)blockIdx.x * total) / gridDim.x);
  const unsigned hi = (unsigned)(((unsigned long long)(blockIdx.x + 1) * total) / gridDim.x);
  const int nitem = (int)(hi - lo);
  if (nitem <= 0) return;
  const float rnt = 1.0f / (float)ntiles, rntx = 1.0f / (float)ntx, rnty = 1.0f / (float)nty;
  auto divmod = [](unsigned n, unsigned d, float rd, unsigned& q, unsigned& r) {  // exact for n < 2^22
    q = (unsigned)((float)n * rd);
    int rr = (int)n - (int)(q * d);
    if (rr < 0) --q, rr += (int)d;
    if (rr >= (int)d) ++q, rr -= (int)d;
    r = (unsigned)rr;
  };
  auto item_of = [&](int k, WmTile& t) {  // (past the end: the last item again -- its DMA is issued and never used)
    const unsigned id = lo + (unsigned)min(k, nitem - 1);
    unsigned pr, tl, r, tx, b, ty;
    divmod(id, ntiles, rnt, pr, tl);
#ifdef WM_PROBE_COLMAJOR  // (probe build: the items of a product column by column, as k_dgrad_diag_ring takes them)
    divmod(tl, (unsigned)nty, rnty, r, ty);
    divmod(r, (unsigned)ntx, rntx, b, tx);
#else
    divmod(tl, (unsigned)ntx, rntx, r, tx);
    divmod(r, (unsigned)nty, rnty, b, ty);
#endif
    t.prod = __builtin_amdgcn_readfirstlane((int)pr), t.x0 = __builtin_amdgcn_readfirstlane((int)tx * 32);
    t.b = __builtin_amdgcn_readfirstlane((int)b), t.y0 = __builtin_amdgcn_readfirstlane((int)ty * WD_ROWS);
    t.g = (const char*)P.p[t.prod].g, t.gx = P.p[t.prod].gx;
    if (PLIFT) t.gP = P.p[t.prod].gP, t.xb = P.p[t.prod].xb;
  };
  auto load_weights4 = [&](const uint4* src) {  // the four loader waves
    for (int u = wv - 4; u < WD_NFRAG; u += 4)
      __builtin_amdgcn_global_load_lds((wd_glb_void*)(src + u * 64 + lane), (wd_lds_void*)(s_w + u * 64), 16, 0, 0);
  };
  // a loader wave's DMA pieces: piece q = (wv - 4) + 4 j covers plane q / 13, pixels 16 u .. 16 u + 15 (u = q % 13); tile independent:
  // halo row / column of this lane's pixel, the byte offset of its 16-byte chunk inside the plane, the LDS unit.  The 40th
  // piece (wave 3, j = 9) repeats piece 38: same bytes to the same unit, so that no piece sits under a branch.
  int p_hr[WM_NJ], p_hc[WM_NJ], p_lds[WM_NJ];
  unsigned p_off[WM_NJ];
#pragma unroll
  for (int j = 0; j < WM_NJ; ++j) {
    const int q = min((wv & 3) + 4 * j, WM_NU - 1), pl = q / WM_UPP, u = q - pl * WM_UPP;
    const int p = 16 * u + (lane >> 2), pc = min(p, WD_HP - 1);
    p_hr[j] = pc / WD_HW, p_hc[j] = pc - p_hr[j] * WD_HW;
    p_off[j] = (unsigned)pl * plane_bytes + (unsigned)(((lane & 3) ^ ((p >> 2) & 3)) * 16);
    p_lds[j] = __builtin_amdgcn_readfirstlane(pl * WM_PLANE + u * 64);
  }
  const char* zero_page = (const char*)wm_zero_page + (lane & 3) * 16;
  auto dma_piece = [&](int j, const WmTile& t, int buf) {  // (j: compile-time after unrolling)
    // ~10 VALU instructions, no branch: 32-bit offset arithmetic (the launcher bounds 3 planes below 4 GiB), bitwise select
    const int y = t.y0 - 1 + p_hr[j], x = t.x0 - 1 + p_hc[j];
    const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const unsigned off = ((unsigned)(t.b * H + y) * (unsigned)W + (unsigned)x) * 64u + p_off[j];
    const unsigned long long m = in ? ~0ull : 0ull;
    const unsigned long long a = (((unsigned long long)t.g + off) & m) | ((unsigned long long)zero_page & ~m);
    __builtin_amdgcn_global_load_lds((wd_glb_void*)a, (wd_lds_void*)(s_a + buf * WM_BUF + p_lds[j]), 16, 0, 0);
  };
  // What the loader team costs the MATRIX team (probe builds, us per launch): loader idle (barriers only) 57.1; LDS-DMA pieces
  // 71.9-72.8 (the default); the same pieces through registers (WM_STAGE=1: a plain 16-byte load per lane, one ds_write_b128
  // once it has landed) 76.6-77.8 -- of which loads without the stores 63.1, stores without the loads 65.1.  Both halves
  // count: the requests with their address arithmetic on the SIMD that also feeds a matrix wave, and the 39 KB per item
  // written into the LDS the matrix waves read their operands from.  Fewer BYTES per item is what helps.
#if WM_STAGE
  auto load_piece = [&](int j, const WmTile& t) -> uint4 {
    const int y = t.y0 - 1 + p_hr[j], x = t.x0 - 1 + p_hc[j];
    const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const unsigned off = ((unsigned)(t.b * H + y) * (unsigned)W + (unsigned)x) * 64u + p_off[j];
    const unsigned long long m = in ? ~0ull : 0ull;
    const unsigned long long a = (((unsigned long long)t.g + off) & m) | ((unsigned long long)zero_page & ~m);
    return *(const uint4*)a;
  };
  auto store_piece = [&](int j, int buf, const uint4& v) { s_a[buf * WM_BUF + p_lds[j] + lane] = v; };
#endif
  float* st = s_stage + (wv & 3) * (32 * WD_SP);
  // epilogue through the wave-private LDS tile [32 pixels][32 channels] (144-byte pixel pitch): the MFMA layout gives a lane
  // 4 x 16 bytes of its pixel's line; read back as 8 pixels x 128 bytes per instruction the wave stores FULL lines,
  // non-temporal.  In parts, so that it can ride behind another item's MFMAs.
  auto epi_write = [&](const f32x16& a) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(float4*)(st + i * WD_SP + 8 * q + 4 * kg) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  };
  float4 ev[4];  // the tile read back row-wise: 8 pixels x 128 bytes per instruction
  auto epi_read = [&](int r) {
    const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
    ev[r] = *(const float4*)(st + p * WD_SP + c4);
  };
  float gpv[9];  // PLIFT: the 3 x 3 neighbourhood of dL/dP of this lane's pixel of the line group about to be stored ...
  uint32_t xbv = 0u;  // ... and its input spike word
  auto plif_load = [&](int r, const WmTile& t) {
    if (PLIFT && t.gP) {  // (block-uniform; clamped addresses: partial tiles read in-image values they never store)
      const int p = 8 * r + (lane >> 3);
      const int y = min(t.y0 + wv, H - 1), x = min(t.x0 + p, W - 1);
      evf_plif_gp_load(t.gP, 1, t.b, y, x, H, W, gpv);
      xbv = t.xb[((long)t.b * H + y) * W + x];
    }
  };
  auto epi_store = [&](int r, const WmTile& t) {
    const int y = t.y0 + wv;
    const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
    float* dst = t.gx + ((unsigned)((t.b * H + y) * W + t.x0 + p) * (unsigned)C32 + (unsigned)c4);
    if (PLIFT && t.gP) {
      const float pv = evf_plif_gp_sum(1, min(y, H - 1), min(t.x0 + p, W - 1), H, W, gpv);
      const uint32_t xq = xbv >> c4;
      ev[r].x += (xq & 1u) ? pv : 0.f, ev[r].y += (xq & 2u) ? pv : 0.f, ev[r].z += (xq & 4u) ? pv : 0.f, ev[r].w += (xq & 8u) ? pv : 0.f;
    }
    if (FULL || (y < H && t.x0 + p < W)) evf_store_nt(dst, ev[r]);
  };
  // Two teams, SEPARATE loops with the same barrier sequence (one per item, one more at a product boundary):
  //   waves 0..3  matrix waves, one per SIMD: 108 MFMAs of row w of the tile; behind them the epilogue of the previous item;
  //   waves 4..7  loader waves, one per SIMD: the 39 LDS-DMA pieces of the NEXT item into the other buffer (last read during
  //               item k - 1: every wave is past that barrier), `s_waitcnt vmcnt(0)`, barrier.
  // tools/probes/mma_probe.hip (cycles per 108-MFMA phase, one wave per SIMD, 1.77 GHz under this load): bare MFMAs 3.68 k;
  // + the 90 operand reads and the DPP moves 4.15 k; + the epilogue 4.48 k; + 10 DMA pieces issued by the SAME wave 6.2 k --
  // a piece costs its issuing wave 130-160 cycles (45 among bare MFMAs), and a wave issues in order, so the pieces must come
  // from waves that have nothing else to do.
  WmTile cur, nxt, prv;
  int wprod;
  const bool loader = wv >= 4;
  item_of(0, cur);
  wprod = cur.prod;
  if (loader) {
#ifndef WM_NOPRIO
    __builtin_amdgcn_s_setprio(3);  // (a few hundred instructions per item: ahead of the MFMA wave of the SIMD in the arbitration)
#endif
    load_weights4((const uint4*)P.p[wprod].wt);
#if WM_STAGE
    {
      uint4 rg[WM_NJ];
#pragma unroll
      for (int j = 0; j < WM_NJ; ++j) rg[j] = load_piece(j, cur);
#pragma unroll
      for (int j = 0; j < WM_NJ; ++j) store_piece(j, 0, rg[j]);
    }
#else
#pragma unroll
    for (int j = 0; j < WM_NJ; ++j) dma_piece(j, cur, 0);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WM_STAMP();
    __syncthreads();
    for (int k = 0; k < nitem; ++k) {
      WM_STAMP();
      item_of(k + 1, nxt);  // (past the end: the last item again, never used)
#ifndef WM_PROBE_NOLOAD  // (probe build: the loader team only keeps the barriers -- what do its pieces cost the matrix team?)
#if WM_STAGE
      uint4 rg[WM_NJ];
#ifndef WM_PROBE_NOGLOAD
#pragma unroll
      for (int j = 0; j < WM_NJ; ++j) rg[j] = load_piece(j, nxt);
#else
#pragma unroll
      for (int j = 0; j < WM_NJ; ++j) rg[j] = make_uint4(k, j, lane, 0);
#endif
      WM_STAMP();
#ifndef WM_PROBE_NOSTORE
#pragma unroll
      for (int j = 0; j < WM_NJ; ++j) store_piece(j, (k + 1) & 1, rg[j]);
#else
#pragma unroll
      for (int j = 0; j < WM_NJ; ++j) asm volatile("" ::"v"(rg[j].x), "v"(rg[j].y), "v"(rg[j].z), "v"(rg[j].w));
#endif
#else
#pragma unroll
      for (int j = 0; j < WM_NJ; ++j) dma_piece(j, nxt, (k + 1) & 1);
      WM_STAMP();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#endif
      WM_STAMP();
      __syncthreads();
      if (k + 1 < nitem && nxt.prod != wprod) {  // (block-uniform) next product: every matrix wave is done with the old weights
        wprod = nxt.prod;
        load_weights4((const uint4*)P.p[wprod].wt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
  } else {
    f32x16 acc_prev = {0};
    auto run_item = [&](int k, auto epi_tag) {
      constexpr bool EPI = decltype(epi_tag)::value;
      WM_STAMP();
      auto side = [&](int slot) {
        if (slot == 70) item_of(k + 1, nxt);  // (~60 scalar-ish instructions: behind an MFMA, not between the barrier and the first one)
        if (EPI) {
          // accumulators -> LDS tile right away, read back a few MFMAs later (the LDS pipe is in order: these reads then sit
          // BEHIND this tap's operand reads and cost no extra wait), stored one line group at a time far behind that
          if (slot == 1) epi_write(acc_prev);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (slot == 13 + r) epi_read(r);
            if (PLIFT && slot == 32 + 12 * r) plif_load(r, prv);
            if (slot == 40 + 12 * r) epi_store(r, prv);
          }
        }
      };
      const f32x16 acc = dg_matrix_phase3<WD_DPPX != 0>(s_w, s_a + (k & 1) * WM_BUF, WM_PLANE, wv * WD_HW + i, lane, side);
      WM_STAMP();
      if (!WM_PIPE) {
        epi_write(acc);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) epi_read(r);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          plif_load(r, cur);
          epi_store(r, cur);
        }
        __builtin_amdgcn_wave_barrier();
      }
      acc_prev = acc;
      prv = cur;
      WM_STAMP();
      __syncthreads();
      if (k + 1 < nitem && nxt.prod != wprod) {
        wprod = nxt.prod;
        __syncthreads();
      }
      cur = nxt;
    };
    WM_STAMP();
    __syncthreads();
    prv = cur;
    run_item(0, std::false_type{});
    for (int k = 1; k < nitem; ++k) run_item(k, std::integral_constant<bool, WM_PIPE != 0>{});
    if (WM_PIPE) {  // the last item's epilogue
      epi_write(acc_prev);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) epi_read(r);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        plif_load(r, prv);
        epi_store(r, prv);
      }
    }
  }
  WM_STAMP();
}

// ---------------------------------------------------------------------------------------------------------------------
// k_dgrad_diag_ring: k_dgrad_diag_dma with the halo rows in a RING.  Vertically adjacent 4-row tiles share two of their six halo
// rows; the items of a product run column by column (ty fastest), and a tile that continues the previous one keeps those two
// rows in LDS: the loader team brings in four new rows (27-30 LDS-DMA pieces) instead of six (39) -- fewer bytes requested
// and fewer bytes written into the LDS the matrix team reads from, which is what the loader team costs it (see dma_piece above).
//   ring   12 rows x 36 pixel slots (34 used: 36 x 12 = 27 units of 16 pixels, so the ring closes on a unit boundary) per plane;
//          tile k occupies rows r0 .. r0 + 5 (mod 12), halo row h of the tile = image row y0 - 1 + h;
//          next tile continues it:  r0' = r0 + 4, rows r0' + 2 .. r0' + 5 are loaded (the four free rows r0 + 6 .. r0 + 9);
//          otherwise:               r0' = r0 + 6, all six rows are loaded (the six free rows).
//   A piece is a 16-pixel unit of one plane; the first / last unit of a region also holds pixels of the neighbouring rows: those
//   lanes are switched off (the row before belongs to the tile in use).
// Same products in the same order per output element: bit-identical to the other input-gradient kernels.
// MEASURED (128 x 128 x B8, us per launch): 76.3-78.7 against 72.4-73.6 for k_dgrad_diag_dma -- NOT the default.  The probe builds
// say why the bytes do not matter: with the loader team idle the dma kernel runs 57.1 and this one 60.4 (three row bases per
// item instead of addresses that never change: +3), and the loader team's pieces cost ~15 us in BOTH -- with 39 pieces of ~25
// vector instructions (here, generic path), 28 pieces of 2 (here, table path) or 39 of 10 (dma); the dma kernel's items taken
// column by column run 72.4.  What the matrix team pays for is that the loader team requests data at all while it computes
// (the launch is power-limited at 1.77 GHz; an idle loader team also means no HBM traffic), not how many requests that takes.
// Kept behind EVF_DGRAD_RING=1 / evf_dgrad_diag_select(2) and under the bit-identity test.
// ---------------------------------------------------------------------------------------------------------------------
extern int evf_dgrad_ring_select;
#define WR_ROWS 12
#define WR_PITCH 36
#define WR_UPP 27                  // 16-pixel units per plane: 12 x 36 / 16
#define WR_PLANE (WR_UPP * 64)     // uint4 per plane
#define WR_BUF (3 * WR_PLANE)      // uint4 of the ring
#define WR_NJ 11                   // pieces per loader wave at most: ceil(3 planes x 14 units / 4 waves)
#define WR_LDS ((size_t)(WD_NFRAG * 64 + WR_BUF) * sizeof(uint4) + (size_t)4 * 32 * WD_SP * 4)

template <bool FULL>
__global__ __launch_bounds__(512) void k_dgrad_diag_ring(EvfDgProds P, unsigned plane_bytes, int H, int W, int ntx, int nty,
                                                         unsigned ntiles, unsigned total) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_w = (uint4*)smem_raw;     // [54][64]
  uint4* s_a = s_w + WD_NFRAG * 64;  // [3][12 rows x 36 pixels][4], chunk c of the pixel in column x of its row in slot c ^ ((x >> 2) & 3)
  float* s_stage = (float*)(s_a + WR_BUF);  // [4 waves][32 pixels][WD_SP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kg = lane >> 5;
  const unsigned lo = (unsigned)(((unsigned long long)blockIdx.x * total) / gridDim.x);
  const unsigned hi = (unsigned)(((unsigned long long)(blockIdx.x + 1) * total) / gridDim.x);
  const int nitem = (int)(hi - lo);
  if (nitem <= 0) return;
  const float rnt = 1.0f / (float)ntiles, rntx = 1.0f / (float)ntx, rnty = 1.0f / (float)nty;
  auto divmod = [](unsigned n, unsigned d, float rd, unsigned& q, unsigned& r) {  // exact for n < 2^22
    q = (unsigned)((float)n * rd);
    int rr = (int)n - (int)(q * d);
    if (rr < 0) --q, rr += (int)d;
    if (rr >= (int)d) ++q, rr -= (int)d;
    r = (unsigned)rr;
  };
  auto item_of = [&](int k, WmTile& t) {  // (past the end: the last item again -- its pieces are issued and never used)
    const unsigned id = lo + (unsigned)min(k, nitem - 1);
    unsigned pr, tl, r, tx, b, ty;
    divmod(id, ntiles, rnt, pr, tl);
    divmod(tl, (unsigned)nty, rnty, r, ty);  // column by column: ty fastest
    divmod(r, (unsigned)ntx, rntx, b, tx);
    t.prod = __builtin_amdgcn_readfirstlane((int)pr), t.x0 = __builtin_amdgcn_readfirstlane((int)tx * 32);
    t.b = __builtin_amdgcn_readfirstlane((int)b), t.y0 = __builtin_amdgcn_readfirstlane((int)ty * WD_ROWS);
    t.g = (const char*)P.p[t.prod].g, t.gx = P.p[t.prod].gx;
  };
  auto continues = [&](const WmTile& a, const WmTile& b) {  // b is the tile below a, same product / sample / column
    return b.prod == a.prod && b.b == a.b && b.x0 == a.x0 && b.y0 == a.y0 + WD_ROWS;
  };
  auto mod12 = [](int r) { return r >= WR_ROWS ? r - WR_ROWS : r; };
  auto load_weights4 = [&](const uint4* src) {  // the four loader waves
    for (int u = wv - 4; u < WD_NFRAG; u += 4)
      __builtin_amdgcn_global_load_lds((wd_glb_void*)(src + u * 64 + lane), (wd_lds_void*)(s_w + u * 64), 16, 0, 0);
  };
  const char* zero_page = (const char*)wm_zero_page + (lane & 3) * 16;
  // the loader team's pieces of tile t whose ring rows start at r0t: its rows 2 .. 5 (t continues the tile in use) or all six
  auto load_region = [&](const WmTile& t, int r0t, bool cont) {
    const int a = mod12(r0t + (cont ? 2 : 0)), n = cont ? 4 : 6;
    const int ustart = (a * WR_PITCH) >> 4, nu = (((a + n) * WR_PITCH + 15) >> 4) - ustart;  // 9, 10 | 14 units per plane
#pragma unroll
    for (int j = 0; j < WR_NJ; ++j) {
      const int e = (wv & 3) + 4 * j;
      if (e < 3 * nu) {  // (wave-uniform)
        const int pl = (e >= nu ? 1 : 0) + (e >= 2 * nu ? 1 : 0);
        int u = ustart + e - pl * nu;
        u = u >= WR_UPP ? u - WR_UPP : u;
        const int q = 16 * u + (lane >> 2);        // ring pixel slot of this lane
        const int rr = (q * 1821) >> 16;           // = q / 36 for q < 432
        const int c = q - rr * WR_PITCH;
        int h = rr - r0t;                          // halo row of the tile that ring row rr holds
        h += h < 0 ? WR_ROWS : 0;
        int hrel = rr - a;
        hrel += hrel < 0 ? WR_ROWS : 0;
        const int y = t.y0 - 1 + h, x = t.x0 - 1 + c;
        const bool in = c < WD_HW && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const unsigned off = ((unsigned)(t.b * H + y) * (unsigned)W + (unsigned)x) * 64u + (unsigned)pl * plane_bytes +
                             (unsigned)(((lane & 3) ^ ((c >> 2) & 3)) * 16);  // (swizzle by column: dgm_load_g_at<true>)
        const unsigned long long m = in ? ~0ull : 0ull;
        const unsigned long long adr = (((unsigned long long)t.g + off) & m) | ((unsigned long long)zero_page & ~m);
        const int dst = __builtin_amdgcn_readfirstlane(pl * WR_PLANE + u * 64);
        if (hrel < n)  // (lanes on a neighbouring row of the first / last unit keep what the ring holds)
          __builtin_amdgcn_global_load_lds((wd_glb_void*)adr, (wd_lds_void*)(s_a + dst), 16, 0, 0);
      }
    }
  };
  // FULL shapes (H % 4 == 0, W % 32 == 0): the same pieces from TABLES.  What the loader team costs the matrix team is, for
  // about half, the vector instructions of its address arithmetic on the SIMD both share (~25 per piece above); here a piece is
  // two of them.  Per lane and piece, fixed for a (region alignment, continues / full) pair -- which changes once per column --:
  // the byte offset of the lane's chunk from the tile's halo origin (scalar base + 32-bit offset addressing) and five flag
  // bits (in the region, left / right / top / bottom halo line).  Per item a scalar origin and the tile's border bits; a lane
  // loads iff (flags & (border | 1)) == 1, and on a border tile the lanes of the lines outside the image store zeros.
  uint32_t tb_off[WR_NJ], tb_flg[WR_NJ];
  int tb_key = -1, tb_nu = 0;
  auto build_tables = [&](int aph, bool cont) {  // aph = a & 3 (0 or 2)
    const int sh = (aph * WR_PITCH) & 15, n = cont ? 4 : 6;
    tb_nu = (sh + n * WR_PITCH + 15) >> 4;
#pragma unroll
    for (int j = 0; j < WR_NJ; ++j) {
      const int e = (wv & 3) + 4 * j;
      const int pl = min((e >= tb_nu ? 1 : 0) + (e >= 2 * tb_nu ? 1 : 0), 2), ui = e - pl * tb_nu;
      const int qrel = 16 * ui + (lane >> 2) - sh;            // pixel slot relative to the region's first one
      const int hrel = qrel < 0 ? -1 : (qrel * 1821) >> 16;   // (qrel < 16 * 14 + 16)
      const int c = qrel - hrel * WR_PITCH, h = hrel + (cont ? 2 : 0);
      const bool live = qrel >= 0 && hrel < n && c < WD_HW && e < 3 * tb_nu;
      tb_off[j] = (unsigned)(h * W + c) * 64u + (unsigned)(((lane & 3) ^ ((c >> 2) & 3)) * 16) + (unsigned)pl * plane_bytes;
      tb_flg[j] = (live ? 1u : 0u) | (c == 0 ? 2u : 0u) | (c == WD_HW - 1 ? 4u : 0u) | (h == 0 ? 8u : 0u) | (h == WD_ROWS + 1 ? 16u : 0u);
    }
  };
  auto load_region_full = [&](const WmTile& t, int r0t, bool cont) {
    const int a = mod12(r0t + (cont ? 2 : 0));
    const int key = (a & 3) * 2 + (cont ? 1 : 0);
    if (key != tb_key) {  // (wave-uniform; once per column)
      build_tables(a & 3, cont);
      tb_key = key;
    }
    const int ustart = (a * WR_PITCH) >> 4;
    const unsigned tf = (t.x0 == 0 ? 2u : 0u) | (t.x0 + 32 == W ? 4u : 0u) | (t.y0 == 0 ? 8u : 0u) | (t.y0 + WD_ROWS == H ? 16u : 0u);
    // halo origin (row y0 - 1, column x0 - 1; before the tensor on a top / left border tile: never dereferenced there)
    const unsigned long long org = (unsigned long long)t.g + (unsigned long long)((long)(t.b * H + t.y0 - 1) * W + t.x0 - 1) * 64ull;
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int j = 0; j < WR_NJ; ++j) {
      const int e = (wv & 3) + 4 * j;
      if (e < 3 * tb_nu) {  // (wave-uniform)
        const int pl = (e >= tb_nu ? 1 : 0) + (e >= 2 * tb_nu ? 1 : 0);
        int u = ustart + e - pl * tb_nu;
        u = u >= WR_UPP ? u - WR_UPP : u;
        const int dst = pl * WR_PLANE + u * 64;  // (scalar)
        const unsigned mk = tb_flg[j] & (tf | 1u);
        if (mk == 1u) {
          unsigned keep;
          const unsigned ldsb = (unsigned)(uintptr_t)(s_a + dst);
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(tb_off[j]), "s"(org), "s"(ldsb)
                       : "memory");
        }
        if (tf && mk > 1u) s_a[dst + lane] = z4;  // (a halo line outside the image)
      }
    }
  };
  float* st = s_stage + (wv & 3) * (32 * WD_SP);
  auto epi_write = [&](const f32x16& a) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(float4*)(st + i * WD_SP + 8 * q + 4 * kg) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  };
  float4 ev[4];
  auto epi_read = [&](int r) {
    const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
    ev[r] = *(const float4*)(st + p * WD_SP + c4);
  };
  auto epi_store = [&](int r, const WmTile& t) {
    const int y = t.y0 + wv;
    const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
    float* dst = t.gx + ((unsigned)((t.b * H + y) * W + t.x0 + p) * (unsigned)C32 + (unsigned)c4);
    if (FULL || (y < H && t.x0 + p < W)) evf_store_nt(dst, ev[r]);
  };
  WmTile cur, nxt, prv;
  int wprod, r0 = 0;  // r0: first ring row of the tile in use (both teams keep the same sequence)
  const bool loader = wv >= 4;
  item_of(0, cur);
  wprod = cur.prod;
  if (loader) {
    __builtin_amdgcn_s_setprio(3);
    load_weights4((const uint4*)P.p[wprod].wt);
    if (FULL) load_region_full(cur, 0, false); else load_region(cur, 0, false);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < nitem; ++k) {
      item_of(k + 1, nxt);  // (past the end: the last item again, never used)
      const bool cont = k + 1 < nitem && continues(cur, nxt);
      const int r0n = mod12(r0 + (cont ? 4 : 6));
#ifndef WR_PROBE_NOLOAD  // (probe build: the loader team only keeps the barriers)
      if (FULL) load_region_full(nxt, r0n, cont); else load_region(nxt, r0n, cont);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (k + 1 < nitem && nxt.prod != wprod) {  // (block-uniform) next product: every matrix wave is done with the old weights
        wprod = nxt.prod;
        load_weights4((const uint4*)P.p[wprod].wt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      cur = nxt, r0 = r0n;
    }
  } else {
    f32x16 acc_prev = {0};
    auto run_item = [&](int k, auto epi_tag) {
      constexpr bool EPI = decltype(epi_tag)::value;
      auto side = [&](int slot) {
        if (slot == 70) item_of(k + 1, nxt);
        if (EPI) {
          if (slot == 1) epi_write(acc_prev);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (slot == 13 + r) epi_read(r);
            if (slot == 40 + 12 * r) epi_store(r, prv);
          }
        }
      };
      int hpd[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        int rb = r0 + wv + d;  // (< 24)
        rb = rb >= WR_ROWS ? rb - WR_ROWS : rb;
        hpd[d] = rb * WR_PITCH + i;
      }
      const f32x16 acc = dg_matrix_phase3h<WD_DPPX != 0, true>(s_w, s_a, WR_PLANE, [&](int dy) { return hpd[dy]; }, lane, side);
      acc_prev = acc;
      prv = cur;
      __syncthreads();
      const bool cont = k + 1 < nitem && continues(cur, nxt);
      if (k + 1 < nitem && nxt.prod != wprod) {
        wprod = nxt.prod;
        __syncthreads();
      }
      r0 = mod12(r0 + (cont ? 4 : 6));
      cur = nxt;
    };
    __syncthreads();
    prv = cur;
    run_item(0, std::false_type{});
    for (int k = 1; k < nitem; ++k) run_item(k, std::true_type{});
    epi_write(acc_prev);  // the last item's epilogue
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) epi_read(r);
#pragma unroll
    for (int r = 0; r < 4; ++r) epi_store(r, prv);
  }
}

bool evf_dgrad_diag_fits(int split, int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return false;
  const long ntiles = (long)evf_cdiv(W, 32) * evf_cdiv(H, WD_ROWS) * B;
  if (ntiles * EVF_DG_MAX_PROD >= (1L << 22)) return false;
  if (split && (3L * B * H * W * 64 >= (1L << 32) || (long)B * H * W * C32 >= (1L << 30))) return false;
  return true;
}

int evf_dgrad_diag_dma_launch(const EvfDgProds& P, int nprod, int B, int H, int W, void* stream) {
  if (nprod <= 0 || nprod > EVF_DG_MAX_PROD || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  const int ntx = evf_cdiv(W, 32), nty = evf_cdiv(H, WD_ROWS);
  const long ntiles = (long)ntx * nty * B, total = ntiles * nprod;
  const long plane_bytes = (long)B * H * W * 64;
  if (total >= (1L << 22) || 3 * plane_bytes >= (1L << 32) || (long)B * H * W * C32 >= (1L << 30)) return EVF_EINVAL;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_dgrad_diag_dma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WM_LDS);
    (void)hipFuncSetAttribute((const void*)k_dgrad_diag_dma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WM_LDS);
    attr = true;
  }
  const int nblk = (int)(total < ncu ? total : ncu);
  bool plift = false;
  for (int k = 0; k < nprod; ++k) {
    plift = plift || P.p[k].gP != nullptr;
    if (P.p[k].gP && !P.p[k].xb) return EVF_EINVAL;
  }
  // default: k_dgrad_diag_dma (every tile's six halo rows loaded, items row by row); EVF_DGRAD_RING=1 / evf_dgrad_diag_select(2):
  // k_dgrad_diag_ring -- measured 76.3-78.7 against 72.4-73.6 us per launch at 128 x 128 x B8 (see the kernel's header)
  static const bool ring_env = []() {
    const char* e = getenv("EVF_DGRAD_RING");
    return e && e[0] == '1';
  }();
  const bool ring = !plift && (evf_dgrad_ring_select < 0 ? ring_env : evf_dgrad_ring_select == 1);
  if (ring) {
    static bool rattr = false;
    if (!rattr) {
      (void)hipFuncSetAttribute((const void*)k_dgrad_diag_ring<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS);
      (void)hipFuncSetAttribute((const void*)k_dgrad_diag_ring<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS);
      rattr = true;
    }
    if (H % WD_ROWS == 0 && W % 32 == 0)
      hipLaunchKernelGGL(k_dgrad_diag_ring<true>, dim3(nblk), dim3(512), WR_LDS, EVF_STREAM(stream), P, (unsigned)plane_bytes, H, W,
                         ntx, nty, (unsigned)ntiles, (unsigned)total);
    else
      hipLaunchKernelGGL(k_dgrad_diag_ring<false>, dim3(nblk), dim3(512), WR_LDS, EVF_STREAM(stream), P, (unsigned)plane_bytes, H, W,
                         ntx, nty, (unsigned)ntiles, (unsigned)total);
    return evf_status();
  }
  if (plift) {  // (a product with the PLIF trace term: the epilogue's own instantiation, so that the LIF kernel stays what it is)
    static bool pattr = false;
    if (!pattr) {
      (void)hipFuncSetAttribute((const void*)k_dgrad_diag_dma<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WM_LDS);
      (void)hipFuncSetAttribute((const void*)k_dgrad_diag_dma<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WM_LDS);
      pattr = true;
    }
    if (H % WD_ROWS == 0 && W % 32 == 0)
      hipLaunchKernelGGL((k_dgrad_diag_dma<true, true>), dim3(nblk), dim3(512), WM_LDS, EVF_STREAM(stream), P, (unsigned)plane_bytes, H, W,
                         ntx, nty, (unsigned)ntiles, (unsigned)total);
    else
      hipLaunchKernelGGL((k_dgrad_diag_dma<false, true>), dim3(nblk), dim3(512), WM_LDS, EVF_STREAM(stream), P, (unsigned)plane_bytes, H, W,
                         ntx, nty, (unsigned)ntiles, (unsigned)total);
    return evf_status();
  }
  if (H % WD_ROWS == 0 && W % 32 == 0)
    hipLaunchKernelGGL(k_dgrad_diag_dma<true>, dim3(nblk), dim3(512), WM_LDS, EVF_STREAM(stream), P, (unsigned)plane_bytes, H, W, ntx,
                       nty, (unsigned)ntiles, (unsigned)total);
  else
    hipLaunchKernelGGL(k_dgrad_diag_dma<false>, dim3(nblk), dim3(512), WM_LDS, EVF_STREAM(stream), P, (unsigned)plane_bytes, H, W, ntx,
                       nty, (unsigned)ntiles, (unsigned)total);
  return evf_status();
}

// (internal: reached through evf_dg_defer_launch in evf_dgrad_b3.hip)
int evf_dgrad_diag_ws_launch(const EvfDgProds& P, int nprod, int B, int H, int W, void* stream) {
  if (nprod <= 0 || nprod > EVF_DG_MAX_PROD || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  const int ntx = evf_cdiv(W, 32), nty = evf_cdiv(H, WD_ROWS);
  const long ntiles = (long)ntx * nty * B, total = ntiles * nprod;
  if (total >= (1L << 22)) return EVF_EINVAL;  // (the float-reciprocal index arithmetic of the kernel)
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_dgrad_diag_ws, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WD_LDS);
    attr = true;
  }
  const int nblk = (int)(total < ncu ? total : ncu);  // one block per CU (149 KiB of LDS each)
  hipLaunchKernelGGL(k_dgrad_diag_ws, dim3(nblk), dim3(512), WD_LDS, EVF_STREAM(stream), P, H, W, ntx, nty, (unsigned)ntiles,
                     (unsigned)total);
  return evf_status();
}

// Input gradients of up to 16 products (gradient planes, weight set, output) in ONE persistent launch (k_dgrad_diag_dma), straight
// from the caller instead of through a backward recording: host arrays of nprod device pointers.  g_split[k]: the three bf16 planes
// [term][B,H,W,32] of dL/d(current) (what the fused backward kernels write); g_x[k] [B,H,W,32] is WRITTEN (no accumulation);
// g_P_raw / x_bits (arrays may be NULL, entries may be NULL): the PLIF trace term of that product (evf_conv_dgrad_b3 with
// `accumulate | 2`).  Bit-identical to one evf_conv_dgrad_b3 call per product.
extern "C" int evf_conv_dgrad_b3_multi(int nprod, const void* const* g_split, const void* const* wT_b3, void* const* g_x,
                                       const void* const* g_P_raw, const void* const* x_bits, int B, int H, int W, void* stream) {
  if (nprod <= 0 || nprod > EVF_DG_MAX_PROD || !g_split || !wT_b3 || !g_x) return EVF_EINVAL;
  if (!evf_dgrad_diag_fits(1, B, H, W)) return EVF_ENOTSUP;
  EvfDgProds P;
  for (int k = 0; k < EVF_DG_MAX_PROD; ++k) {
    const int q = k < nprod ? k : 0;
    if (!g_split[q] || !wT_b3[q] || !g_x[q]) return EVF_EINVAL;
    const float* gp = g_P_raw ? (const float*)g_P_raw[q] : nullptr;
    const uint32_t* xb = x_bits ? (const uint32_t*)x_bits[q] : nullptr;
    if ((gp != nullptr) != (xb != nullptr)) return EVF_EINVAL;
    P.p[k] = EvfDgProd{g_split[q], wT_b3[q], (float*)g_x[q], gp, xb};
  }
  evf_prof_mark(9, 0, stream);
  const int rc = evf_dgrad_diag_dma_launch(P, nprod, B, H, W, stream);
  evf_prof_mark(9, 1, stream);
  return rc;
}
extern "C" int evf_conv_dgrad_b3_multi_fits(int B, int H, int W) { return evf_dgrad_diag_fits(1, B, H, W) ? 1 : 0; }
