// Input-gradient conv, wave-specialised form (gfx950) -- the default behind evf_conv_dgrad_b3_f32[_pair].
//
//   g_x[pix][ci] (+)= sum_{tap,co} g[pix + tap][co] * Wt[tap][co][ci]        (same six-term exact bf16 split as
//   evf_dgrad_b3.hip: gh*wh, gh*wm, gm*wh, gh*wl, gl*wh, gm*wm; fp32 accumulation in v_mfma_f32_32x32x16_bf16)
//
// k_conv_dgrad_b3_lds (evf_dgrad_b3.hip) runs its phases back to back -- request the halo, wait, split, 108 MFMAs per wave,
// store -- with ONE block per CU, so the matrix cores idle while the halo arrives and HBM idles during the matrix phase
// (phase stamps: 44 % load issue + wait, 28 % matrix, 14 % stores).  Here a 512-thread block is two teams:
//
//   waves 0..3  CONSUMERS, one per SIMD: the 108 MFMAs of one 32-pixel row each, operands from LDS, epilogue stores;
//   waves 4..7  PRODUCERS: fetch the NEXT tile's fp32 gradient halo, do the exact 3-way bf16 split on the VALU and write
//               the three planes into the other half of a double buffer.
//
// A tile is 4 rows x 32 pixels (6 x 34 halo pixels x 3 planes x 64 B = 38 KiB per buffer; + 54 KiB of split weights =
// 131 KiB of LDS), blocks are persistent over tiles (tile = block + k * grid), ONE barrier per tile.  The matrix pipe and
// the VALU are separate issue ports, so a producer wave's split and a consumer wave's MFMAs on the same SIMD overlap, and
// the producers' HBM latency hides under the consumers' matrix phase.  Out-of-image halo pixels are written as zeros by
// the producers (the consumers need no masks).  Accumulation order per output element is that of k_conv_dgrad_b3_lds:
// the two kernels give bit-identical results.
#include "evf_common.h"
#include "evf_dgrad_mma.h"
#include "evf_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void ws_lds_void;
typedef __attribute__((address_space(1))) const void ws_glb_void;

#define C32 32
#define WS_ROWS 4                    // tile rows = consumer waves
#define WS_HW 34                     // halo width
#define WS_HP ((WS_ROWS + 2) * WS_HW)  // 204 halo pixels (51 groups of 4: the slot swizzle works on groups of 4)
#define WS_NFRAG 54                  // weight fragments of 1 KiB: [tap 9][m 2][term 3]
#define WS_PLANE (WS_HP * 4)         // uint4 per plane
#define WS_BUF (3 * WS_PLANE)        // uint4 per halo buffer (hi, mid, lo)
#define WS_ITEMS (WS_HP * 4)         // (halo pixel, 8-channel chunk) items per tile
#define WS_NIT ((WS_ITEMS + 255) / 256)

__device__ __forceinline__ uint32_t ws_bf16(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

struct WsTile {
  int b, y0, x0;
};

#ifdef WS_STAMPS  // phase stamps (debug build loaded through EVF_LIB): [block < 16][team 2][64] shader-clock values
__device__ unsigned long long ws_stamps[16 * 2 * 64];
extern "C" int evf_debug_ws_stamps(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(ws_stamps), sizeof(ws_stamps))); }
#define WS_STAMP()                                                                                       \
  do {                                                                                                   \
    if (blockIdx.x < 16 && lane == 0 && (wv == 0 || wv == WS_ROWS) && nst < 64)                          \
      ws_stamps[(blockIdx.x * 2 + (wv ? 1 : 0)) * 64 + nst++] = __builtin_readcyclecounter();           \
  } while (0)
#else
#define WS_STAMP() do {} while (0)
#endif

template <bool ACC, bool PLIF, bool PAIR>
__global__ __launch_bounds__(512) void k_conv_dgrad_ws(const float4* __restrict__ gf, const uint4* __restrict__ wt,
                                                       float* __restrict__ gx, int B, int H, int W, int ntx, int nty,
                                                       long ntiles, const float* __restrict__ gPb,
                                                       const uint32_t* __restrict__ xbits, const uint4* __restrict__ wt2,
                                                       float* __restrict__ gx2, int plif_raw) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_w = (uint4*)smem_raw;          // [54][64]
  uint4* s_a = s_w + WS_NFRAG * 64;       // [2][3][WS_HP][4], chunk c of pixel p in slot c ^ ((p >> 2) & 3)
  float* s_gp = (float*)(s_a + 2 * WS_BUF);  // PLIF: [2][4 rows x 32 pixels] d loss / d(input spike) through the trace, per tile
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool producer = wv >= WS_ROWS;
  const int i = lane & 31, kg = lane >> 5;
  int nst = 0;
  (void)nst;
  WS_STAMP();

  // XCD-aware tile order: consecutive blocks sit on different XCDs (block b -> XCD b % 8, observed), so give every XCD
  // one contiguous eighth of the tile sequence -- vertically adjacent tiles (which share two halo rows) then meet in the
  // same L2.  Speed only: any mapping that visits every tile once is correct.
  const unsigned nblk = gridDim.x, nx = min(8u, nblk), nt = (unsigned)ntiles;  // (the launcher keeps ntiles < 2^31)
  const unsigned per_xcd = (nt + nx - 1) / nx;
  const unsigned xcd = blockIdx.x % nx, rank = blockIdx.x / nx, nrank = (nblk - xcd + nx - 1) / nx;  // blocks of this XCD: rank 0..nrank-1
  // quotient / remainder by a small runtime divisor without the ~40-instruction integer division (the producers compute a tile
  // index per step on their critical path): float reciprocal + one-step fix-up, exact for n < 2^22
  const float rntx = 1.0f / (float)ntx, rnty = 1.0f / (float)nty;
  auto divmod = [](unsigned n, unsigned d, float rd, unsigned& q, unsigned& r) {
    q = (unsigned)((float)n * rd);
    int rr = (int)n - (int)(q * d);
    if (rr < 0) --q, rr += (int)d;
    if (rr >= (int)d) ++q, rr -= (int)d;
    r = (unsigned)rr;
  };
  auto tile_of = [&](int k, WsTile& t) -> bool {
    const unsigned local = rank + (unsigned)k * nrank;
    const unsigned idx = xcd * per_xcd + local;
    const bool ok = local < per_xcd && idx < nt;
    const unsigned id = ok ? idx : 0u;  // (a valid tile either way: the prefetch past the end is issued and ignored)
    unsigned r, tx, b, ty;
    divmod(id, (unsigned)ntx, rntx, r, tx);
    divmod(r, (unsigned)nty, rnty, b, ty);
    t.x0 = (int)tx * 32;
    t.b = (int)b;
    t.y0 = (int)ty * WS_ROWS;
    return ok;
  };

  auto load_weights = [&](const uint4* src) {  // all 8 waves (the weight swap of the PAIR form)
    for (int u = wv; u < WS_NFRAG; u += 8)
      __builtin_amdgcn_global_load_lds((ws_glb_void*)(src + u * 64 + lane), (ws_lds_void*)(s_w + u * 64), 16, 0, 0);
  };
  // prologue: the CONSUMERS bring the weights in (LDS-DMA) while the producers fetch and split the first tile (memory
  // returns in order per wave: a producer that queued weight pieces first would see its halo only after them).
  // (Measured alternative: the same 54 KiB through registers -- 14 loads + 14 ds_write_b128 per consumer wave -- lands at
  //  cycle 13 k instead of 10 k.)
  if (!producer) {
    for (int u = wv; u < WS_NFRAG; u += WS_ROWS)
      __builtin_amdgcn_global_load_lds((ws_glb_void*)(wt + u * 64 + lane), (ws_lds_void*)(s_w + u * 64), 16, 0, 0);
  } else {
    // the producers' few hundred instructions per tile go first; the MFMA wave of the SIMD fills every other issue slot
    // (at equal priority the older MFMA wave wins the arbitration and the producer needed ~20 cycles per instruction)
    __builtin_amdgcn_s_setprio(3);
  }
  // ---- producer: fp32 halo of a tile -> registers (one of two sets: the requests run TWO tiles ahead of the consumers,
  // so that an HBM round trip hides under a whole matrix phase) -> exact split -> planes of buffer `buf`
  // PLIF: the producers also fetch, with the halo, the nine values of dL/d(pooled activity) around each of the tile's 128
  // pixels (threads 0..127) and leave their pooled, scaled sum in LDS at split time -- in the consumers' prologue those nine
  // loads and ~40 address instructions sat in front of every matrix phase (+7 us per launch at 260 x 346 x B4).
  struct Regs {
    float4 lo4[WS_NIT], hi4[WS_NIT];
    float gp[PLIF ? 9 : 1];
  };
  const int ptid = tid - WS_ROWS * 64;  // 0..255 among the producers
  int ihr[WS_NIT], ihc[WS_NIT];  // halo row / column of this thread's items (tile independent)
#pragma unroll
  for (int n = 0; n < WS_NIT; ++n) {
    const int p = min(ptid + n * 256, WS_ITEMS - 1) >> 2;
    ihr[n] = p / WS_HW, ihc[n] = p - ihr[n] * WS_HW;
  }
  auto fetch = [&](Regs& r, const WsTile& t) {
#pragma unroll
    for (int n = 0; n < WS_NIT; ++n) {
      const int c = (ptid + n * 256) & 3;  // (256 is a multiple of 4: the clamped item keeps its chunk)
      const int yr = t.y0 - 1 + ihr[n], xr = t.x0 - 1 + ihc[n];
      const int yy = min(max(yr, 0), H - 1), xx = min(max(xr, 0), W - 1);  // loads stay unconditional; zeroed in split_store
      const float4* src = gf + (((long)t.b * H + yy) * W + xx) * 8 + 2 * c;
      r.lo4[n] = src[0], r.hi4[n] = src[1];
    }
    if constexpr (PLIF) evf_plif_gp_load(gPb, plif_raw, t.b, t.y0 + ((ptid & 127) >> 5), t.x0 + (ptid & 31), H, W, r.gp);
  };
  auto split_store = [&](const Regs& r, int buf, const WsTile& t) {
    uint4* dst = s_a + buf * WS_BUF;
    if constexpr (PLIF) {
      const float pv = evf_plif_gp_sum(plif_raw, t.y0 + ((ptid & 127) >> 5), t.x0 + (ptid & 31), H, W, r.gp);
      if (ptid < 128) s_gp[buf * 128 + ptid] = pv;
    }
#pragma unroll
    for (int n = 0; n < WS_NIT; ++n) {
      const int it = ptid + n * 256;
      if (it < WS_ITEMS) {
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
        for (int sp = 0; sp < 3; ++sp)
          dst[sp * WS_PLANE + slot] = make_uint4(t3[sp][0], t3[sp][1], t3[sp][2], t3[sp][3]);
      }
    }
  };

  const uint32_t nomask[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int wset = 0;  // PAIR: which weight set sits in LDS
  // One tile: the consumers run tile k out of buffer k & 1; the producers request tile k+2 into `rq` and split tile k+1
  // (requested one iteration ago into `rs`) into the other buffer.  ONE barrier per tile (+2 around the weight swap of PAIR).
  auto step = [&](int k, const WsTile& cur, const WsTile& t1, bool have1, const WsTile& t2, bool have2, Regs& rq, const Regs& rs) {
    const int buf = k & 1;
    WS_STAMP();
    if (producer) {
      // (unconditional: past the last tile the current one is requested again and never used -- a conditional fetch makes the
      //  register set a phi of "old" and "new" and hipcc then waits for the loads right here to copy them)
#ifndef WS_NOPRODUCE
      fetch(rq, have2 ? t2 : cur);
      WS_STAMP();
      if (have1) split_store(rs, buf ^ 1, t1);
#endif  // (the consumers finished reading that half before the previous barrier)
      WS_STAMP();
      if (PAIR) {  // the consumers swap the weight set between their two matrix phases: take part in both barriers
        __syncthreads();
        load_weights(wset ? wt : wt2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    } else {
      const int y = cur.y0 + wv;
      const long pixq = ((long)cur.b * H + min(y, H - 1)) * W + min(cur.x0 + i, W - 1);
      float4 oldv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) oldv[q] = ACC ? *(const float4*)(gx + pixq * C32 + 8 * q + 4 * kg) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float pv = PLIF ? s_gp[buf * 128 + wv * 32 + i] : 0.f;  // (left by the producers with the tile's planes)
      const uint32_t xb = PLIF ? xbits[pixq] : 0u;
      const bool ok = y < H && cur.x0 + i < W;
      const long pix = ((long)cur.b * H + y) * W + cur.x0 + i;
#pragma unroll
      for (int ph = 0; ph < (PAIR ? 2 : 1); ++ph) {
        const int set = PAIR ? (wset ^ ph) : 0;
        if (PAIR && ph == 1) {
          __syncthreads();  // every consumer is done with the current weight set
          load_weights(set ? wt2 : wt);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
        const f32x16 acc = dg_matrix_phase<false>(s_w, s_a + buf * WS_BUF, WS_PLANE, wv * WS_HW + i, lane, nomask);
        WS_STAMP();
        if (ok) {
          if (set == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              // PLIF: the pooled pre-synaptic trace also reads the input spikes (see k_conv_dgrad_b3_lds)
              const uint32_t xq = xb >> (8 * q + 4 * kg);
              const float o[4] = {oldv[q].x, oldv[q].y, oldv[q].z, oldv[q].w};
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e] + o[e] + (((xq >> e) & 1u) ? pv : 0.f);
              *(float4*)(gx + pix * C32 + 8 * q + 4 * kg) = make_float4(v[0], v[1], v[2], v[3]);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *(float4*)(gx2 + pix * C32 + 8 * q + 4 * kg) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
          }
        }
      }
    }
    if (PAIR) wset ^= 1;
    WS_STAMP();
    __syncthreads();  // planes of tile k+1 complete; planes of tile k free
  };

  Regs ra, rb;
  WsTile t0, t1, t2;
  bool h0 = tile_of(0, t0), h1 = tile_of(1, t1), h2;
  if (producer) {
    fetch(ra, t0);
    if (h0) split_store(ra, 0, t0);
    fetch(rb, h1 ? t1 : t0);
  }
  WS_STAMP();
  if (!producer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight DMA of this wave has landed
  WS_STAMP();
  __syncthreads();  // weights and the planes of tile 0 are in LDS (the producers' requests for tile 1 stay in flight)
  for (int k = 0; h0; k += 2) {  // unrolled by two: the register sets alternate without moves
    h2 = tile_of(k + 2, t2);
    step(k, t0, t1, h1, t2, h2, ra, rb);
    if (!h1) break;
    h0 = tile_of(k + 3, t0);
    step(k + 1, t1, t2, h2, t0, h0, rb, ra);
    t1 = t0;
    h1 = h0;
    t0 = t2;
    h0 = h2;
  }
  WS_STAMP();
}

#define WS_LDS ((size_t)(WS_NFRAG * 64 + 2 * WS_BUF) * sizeof(uint4) + 2 * 128 * sizeof(float))

// (internal: reached through evf_conv_dgrad_b3_f32[_pair], see dg_launch in evf_dgrad_b3.hip)
int evf_dgrad_ws_launch(const float* g_cur, const void* wT_b3, float* g_x, int accumulate, int B, int H, int W,
                                   const float* g_P, const uint32_t* x_bits, const void* wT2_b3, float* g_x2, int max_blocks,
                                   void* stream) {
  if (!g_cur || !wT_b3 || !g_x || B <= 0 || H <= 0 || W <= 0 || ((g_P != nullptr) != (x_bits != nullptr)) ||
      ((wT2_b3 != nullptr) != (g_x2 != nullptr)) || (long)B * evf_cdiv(H, WS_ROWS) * evf_cdiv(W, 32) >= (1L << 31))
    return EVF_EINVAL;
  const int ntx = evf_cdiv(W, 32), nty = evf_cdiv(H, WS_ROWS);
  const long ntiles = (long)ntx * nty * B;
  if (max_blocks <= 0) max_blocks = 256;  // one block per CU (131 KiB of LDS each)
  const int nblk = (int)(ntiles < max_blocks ? ntiles : max_blocks);
  dim3 grid(nblk), block(512);
  const bool acc = (accumulate & 1) != 0, plif = g_P != nullptr, pair = wT2_b3 != nullptr;
  const int plif_raw = (accumulate & 2) ? 1 : 0;  // g_P is evf_plif_trace_bwd's RAW map: AvgPool3x3^T / 32 applied in the kernel
#define WS_GO(A_, P_, R_)                                                                                                     \
  do {                                                                                                                        \
    static bool attr = false;                                                                                                 \
    if (!attr) {                                                                                                              \
      (void)hipFuncSetAttribute((const void*)k_conv_dgrad_ws<A_, P_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                (int)WS_LDS);                                                                                 \
      attr = true;                                                                                                            \
    }                                                                                                                         \
    hipLaunchKernelGGL((k_conv_dgrad_ws<A_, P_, R_>), grid, block, WS_LDS, EVF_STREAM(stream), (const float4*)g_cur,          \
                       (const uint4*)wT_b3, g_x, B, H, W, ntx, nty, ntiles, g_P, x_bits, (const uint4*)wT2_b3, g_x2,          \
                       plif_raw);                                                                                             \
  } while (0)
#define WS_AP(R_)                  \
  do {                             \
    if (acc && plif)               \
      WS_GO(true, true, R_);       \
    else if (acc)                  \
      WS_GO(true, false, R_);      \
    else if (plif)                 \
      WS_GO(false, true, R_);      \
    else                           \
      WS_GO(false, false, R_);     \
  } while (0)
  if (pair)
    WS_AP(true);
  else
    WS_AP(false);
#undef WS_AP
#undef WS_GO
  return evf_status();
}
