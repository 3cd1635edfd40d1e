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
