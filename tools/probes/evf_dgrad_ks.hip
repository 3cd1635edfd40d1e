// LAB NOTEBOOK, not built: the K-split form of the persistent input-gradient launch (round 3).
//
// Two matrix waves per SIMD split the K range of a tile row (wave m runs the 54 MFMAs of accumulator acc[m] of
// dg_matrix_phase, the sum acc0 + acc1 after an exchange through LDS: bit-identical, verified by
// test_recorded_input_gradient_cells_are_bit_identical), the 27 weight fragments of a wave live in registers (no weight image
// in LDS), no loader team.  Measured in the replayed train step (128 x 128 x B8, per k_dgrad_diag launch):
//   all ten LDS-DMA pieces on the m = 1 waves   81.1 us  (m = 1 phase 5.2 k cycles, m = 0 phase 3.7 k + 2.2 k at the barrier)
//   five pieces per wave                        87.6 us  (both phases 5.5 k cycles)
//   k_dgrad_diag_dma (4 matrix + 4 loader waves) 74.7-76.3 us  <- shipped
// A global_load_lds piece issued between the MFMAs of a wave costs that wave ~350 cycles, not the ~140 it costs a wave that
// does nothing else: LDS-DMA belongs on waves without a matrix stream.  The part that goes into evf_dgrad_mma.h comes
// first, the kernel (part of evf_dgrad_diag.hip) second.
#if 0
// ---------------------------------------------------------------------------------------------------------------------
// Fourth form (k_dgrad_diag_ks, evf_dgrad_diag.hip): the K range of a tile row split between TWO waves of one SIMD.  Wave
// m (0 / 1) owns the m-th K half of every tap -- exactly the groups the accumulator acc[m] of dg_matrix_phase takes, in the
// same order -- so its 54 MFMAs produce acc[m] bit for bit; the caller adds the two (acc0 + acc1, the same addition) after
// an exchange through LDS.  The wave's 27 weight fragments (tap x term, 16 bytes per lane each) live in REGISTERS for the
// whole product (108 VGPRs: no weight image in LDS, no weight reads in the phase); the gradient fragments come from LDS,
// the middle tap of a row by DPP moves as above.  `side(slot)`, slot = 0..53, behind every MFMA.
// ---------------------------------------------------------------------------------------------------------------------
struct DgmWreg {
  uint4 f[27];  // [tap][term hi, mid, lo]
};
__device__ __forceinline__ void dgm_load_wreg(DgmWreg& w, const uint4* __restrict__ wt, int m, int lane) {
#pragma unroll
  for (int tau = 0; tau < 9; ++tau)
#pragma unroll
    for (int t = 0; t < 3; ++t) w.f[tau * 3 + t] = wt[((2 * tau + m) * 3 + t) * 64 + lane];
}

template <class Side>
__device__ __forceinline__ dgm_f32x16 dg_half_phase(const DgmWreg& w, const uint4* __restrict__ pa, int plane, int hp0, int lane,
                                                    int m, Side&& side) {
  dgm_f32x16 acc = {0};
  DgmG r0, r2[2], mid;
  dgm_load_g(r0, 0, 0, m, pa, plane, hp0, lane);
  dgm_load_g(r2[0], 0, 2, m, pa, plane, hp0, lane);
  mid = r0;
#pragma unroll
  for (int tau = 0; tau < 9; ++tau) {
    const int dy = tau / 3, dx = tau - 3 * dy;
    if (dx == 1 && dy < 2) {  // r0 is dead (its middle fragment is built); the next row's dx = 2 set goes to the other registers
      dgm_load_g(r0, dy + 1, 0, m, pa, plane, hp0, lane);
      dgm_load_g(r2[(dy + 1) & 1], dy + 1, 2, m, pa, plane, hp0, lane);
    }
    const DgmG c = dx == 0 ? r0 : (dx == 1 ? mid : r2[dy & 1]);
    const dgm_bf16x8 wh = *(const dgm_bf16x8*)&w.f[tau * 3], wm = *(const dgm_bf16x8*)&w.f[tau * 3 + 1],
                     wl = *(const dgm_bf16x8*)&w.f[tau * 3 + 2];
    const dgm_bf16x8 ah = *(const dgm_bf16x8*)&c.h, am = *(const dgm_bf16x8*)&c.m, al = *(const dgm_bf16x8*)&c.l;
    DgmG nm = mid;
    const bool build = dx == 0;
    auto bld = [&](int q) {  // one of the 12 dwords of the middle fragment
      if (!build) return;
      const int pl = q / 4, d = q % 4;
      const DgmG &a = r0, &b = r2[dy & 1];
      const uint4& fa = pl == 0 ? a.h : (pl == 1 ? a.m : a.l);
      const uint4& fb = pl == 0 ? b.h : (pl == 1 ? b.m : b.l);
      uint4& fo = pl == 0 ? nm.h : (pl == 1 ? nm.m : nm.l);
      const uint32_t va = d == 0 ? fa.x : (d == 1 ? fa.y : (d == 2 ? fa.z : fa.w));
      const uint32_t vb = d == 0 ? fb.x : (d == 1 ? fb.y : (d == 2 ? fb.z : fb.w));
      const uint32_t r = dgm_dpp_mid1(va, vb);
      if (d == 0) fo.x = r; else if (d == 1) fo.y = r; else if (d == 2) fo.z = r; else fo.w = r;
    };
#define DGM4_STEP(J, WA, GA)                                                 \
  __builtin_amdgcn_sched_barrier(0);                                         \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WA, GA, acc, 0, 0, 0);       \
  __builtin_amdgcn_sched_barrier(0);                                         \
  bld(2 * (J)), bld(2 * (J) + 1);                                            \
  side(tau * 6 + (J));
    DGM4_STEP(0, wm, am)  // (the term order of dg_matrix_phase: smallest first)
    DGM4_STEP(1, wh, al)
    DGM4_STEP(2, wl, ah)
    DGM4_STEP(3, wh, am)
    DGM4_STEP(4, wm, ah)
    DGM4_STEP(5, wh, ah)
#undef DGM4_STEP
    __builtin_amdgcn_sched_barrier(0);
    mid = nm;
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_dgrad_diag_ks: as k_dgrad_diag_dma, with TWO matrix waves per SIMD and no loader team.
//
// Stamps of k_dgrad_diag_dma (4 matrix + 4 loader waves; cycles per item at 1.77 GHz): 5.1 k matrix phase + 0.4 k barrier
// against 3.7 k for the bare MFMAs -- one wave per SIMD exposes every operand-read latency, the epilogue instructions and
// the barrier.  Here waves w and w + 4 (same SIMD) split the K range of tile row w: wave m = 0 / 1 runs the 54 MFMAs of
// accumulator acc[m] (dg_half_phase), so the SIMD's matrix pipe is fed by two instruction streams, and
//   * the weights live in registers (27 fragments per wave): LDS holds only the two halo buffers, the exchange tiles and
//     the epilogue tiles -- 128 KiB -- and a phase has 18 operand reads per wave instead of 90;
//   * the m = 1 waves issue the LDS-DMA pieces of the next item behind their MFMAs (the ~140 cycles a piece costs its
//     issuing wave are the other wave's MFMA time), then park acc[1] in an exchange tile X[item parity][row];
//   * the m = 0 waves, behind the MFMAs of the NEXT item, add acc[0] + X (the addition of dg_matrix_phase: bit-identical
//     results) and write the tile out through their staging tile (full 128-byte lines, non-temporal).
// One barrier per item orders everything: X[k & 1] is written before barrier k, read after it, and written again only
// after barrier k + 1, which its reader passes after the read.  A product boundary costs 27 register loads per wave.
// ---------------------------------------------------------------------------------------------------------------------
#define WK_LDS ((size_t)(2 * WM_BUF) * sizeof(uint4) + (size_t)2 * 4 * 4 * 64 * 16 + (size_t)4 * 32 * WD_SP * 4)

template <bool FULL>
__global__ __launch_bounds__(512) void k_dgrad_diag_ks(EvfDgProds P, unsigned plane_bytes, int H, int W, int ntx, int nty,
                                                       unsigned ntiles, unsigned total) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_a = (uint4*)smem_raw;        // [2][3][WM_UPP * 16 pixels][4], chunk c of pixel p in slot c ^ ((p >> 2) & 3)
  float4* s_x = (float4*)(s_a + 2 * WM_BUF);  // [2 parities][4 rows][4][64 lanes]: acc[1] of the m = 1 wave
  float* s_stage = (float*)(s_x + 2 * 4 * 4 * 64);  // [4 rows][32 pixels][WD_SP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = wv & 3, m = wv >> 2;
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
    divmod(tl, (unsigned)ntx, rntx, r, tx);
    divmod(r, (unsigned)nty, rnty, b, ty);
    t.prod = __builtin_amdgcn_readfirstlane((int)pr), t.x0 = __builtin_amdgcn_readfirstlane((int)tx * 32);
    t.b = __builtin_amdgcn_readfirstlane((int)b), t.y0 = __builtin_amdgcn_readfirstlane((int)ty * WD_ROWS);
    t.g = (const char*)P.p[t.prod].g, t.gx = P.p[t.prod].gx;
  };
  WmTile cur, nxt, prv;
  item_of(0, cur);
  int wprod = cur.prod;
  DgmWreg wreg;
  dgm_load_wreg(wreg, (const uint4*)P.p[wprod].wt, m, lane);
  const int hp0 = row * WD_HW + i;
  const char* zero_page = (const char*)wm_zero_page + (lane & 3) * 16;

  // LDS-DMA pieces of an item: 39 (+1 repeat), five per wave: wave (row, m) takes pieces row + 4 (2 jj + m), jj = 0..4 -- a piece
  // costs its issuing wave 130-160 cycles (the CU's DMA path takes ~30 B/clk) during which the OTHER wave of the SIMD issues
  // MFMAs; all ten on the m = 1 waves made those the critical path (5.2 k against 3.7 k cycles per item).
  int p_hr[5], p_hc[5], p_lds[5];
  unsigned p_off[5];
#pragma unroll
  for (int jj = 0; jj < 5; ++jj) {  // (piece 39 repeats piece 38: same bytes to the same unit)
    const int q = min(row + 4 * (2 * jj + m), WM_NU - 1), pl = q / WM_UPP, u = q - pl * WM_UPP;
    const int p = 16 * u + (lane >> 2), pc = min(p, WD_HP - 1);
    p_hr[jj] = pc / WD_HW, p_hc[jj] = pc - p_hr[jj] * WD_HW;
    p_off[jj] = (unsigned)pl * plane_bytes + (unsigned)(((lane & 3) ^ ((p >> 2) & 3)) * 16);
    p_lds[jj] = __builtin_amdgcn_readfirstlane(pl * WM_PLANE + u * 64);
  }
  auto dma_piece = [&](int j, const WmTile& t, int buf) {
    const int y = t.y0 - 1 + p_hr[j], x = t.x0 - 1 + p_hc[j];
    const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const unsigned off = ((unsigned)(t.b * H + y) * (unsigned)W + (unsigned)x) * 64u + p_off[j];
    const unsigned long long mk = in ? ~0ull : 0ull;
    const unsigned long long a = (((unsigned long long)t.g + off) & mk) | ((unsigned long long)zero_page & ~mk);
    __builtin_amdgcn_global_load_lds((wd_glb_void*)a, (wd_lds_void*)(s_a + buf * WM_BUF + p_lds[j]), 16, 0, 0);
  };
#pragma unroll
  for (int jj = 0; jj < 5; ++jj) dma_piece(jj, cur, 0);

  if (m == 1) {
    // ---- K half 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WM_STAMP();
    __syncthreads();
    for (int k = 0; k < nitem; ++k) {
      WM_STAMP();
      item_of(k + 1, nxt);
      const int nb = (k + 1) & 1;
      auto side = [&](int slot) {
#pragma unroll
        for (int jj = 0; jj < 5; ++jj)
          if (slot == 3 + 10 * jj) dma_piece(jj, nxt, nb);
      };
      const f32x16 acc = dg_half_phase(wreg, s_a + (k & 1) * WM_BUF, WM_PLANE, hp0, lane, 1, side);
      WM_STAMP();
      float4* xo = s_x + (((k & 1) * 4 + row) * 4) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) xo[q * 64] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the next item have landed
      WM_STAMP();
      __syncthreads();
      if (k + 1 < nitem && nxt.prod != wprod) {  // next product: its weight fragments into the registers
        wprod = nxt.prod;
        dgm_load_wreg(wreg, (const uint4*)P.p[wprod].wt, 1, lane);
      }
    }
  } else {
    // ---- K half 0 + the sum and the epilogue of the previous item
    float* st = s_stage + row * (32 * WD_SP);
    float4 ev[4], xv[4];
    f32x16 acc_prev = {0}, fin = {0};
    auto x_read = [&](int par) {
      const float4* xi = s_x + ((par * 4 + row) * 4) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[q] = xi[q * 64];
    };
    auto x_add = [&]() {  // acc[0] + acc[1], element by element as dg_matrix_phase returns it
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fin[4 * q] = acc_prev[4 * q] + xv[q].x, fin[4 * q + 1] = acc_prev[4 * q + 1] + xv[q].y;
        fin[4 * q + 2] = acc_prev[4 * q + 2] + xv[q].z, fin[4 * q + 3] = acc_prev[4 * q + 3] + xv[q].w;
      }
    };
    auto epi_write = [&]() {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(float4*)(st + i * WD_SP + 8 * q + 4 * kg) = make_float4(fin[4 * q], fin[4 * q + 1], fin[4 * q + 2], fin[4 * q + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    auto epi_read = [&](int r) { ev[r] = *(const float4*)(st + (8 * r + (lane >> 3)) * WD_SP + (lane & 7) * 4); };
    auto epi_store = [&](int r, const WmTile& t) {
      const int y = t.y0 + row;
      const int p = 8 * r + (lane >> 3), c4 = (lane & 7) * 4;
      float* dst = t.gx + ((unsigned)((t.b * H + y) * W + t.x0 + p) * (unsigned)C32 + (unsigned)c4);
      if (FULL || (y < H && t.x0 + p < W)) evf_store_nt(dst, ev[r]);
    };
    auto run_item = [&](int k, auto epi_tag) {
      constexpr bool EPI = decltype(epi_tag)::value;
      WM_STAMP();
      item_of(k + 1, nxt);
      const int par = (k + 1) & 1;  // parity of item k - 1 = buffer of item k + 1
      auto side = [&](int slot) {
#pragma unroll
        for (int jj = 0; jj < 5; ++jj)
          if (slot == 7 + 10 * jj) dma_piece(jj, nxt, par);
        if (EPI) {
          if (slot == 0) x_read(par);
          if (slot == 4) x_add();
          if (slot == 5) epi_write();
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (slot == 9 + r) epi_read(r);
            if (slot == 22 + 8 * r) epi_store(r, prv);
          }
        }
      };
      const f32x16 acc = dg_half_phase(wreg, s_a + (k & 1) * WM_BUF, WM_PLANE, hp0, lane, 0, side);
      WM_STAMP();
      acc_prev = acc;
      prv = cur;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the next item have landed
      WM_STAMP();
      __syncthreads();
      if (k + 1 < nitem && nxt.prod != wprod) {
        wprod = nxt.prod;
        dgm_load_wreg(wreg, (const uint4*)P.p[wprod].wt, 0, lane);
      }
      cur = nxt;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WM_STAMP();
    __syncthreads();
    prv = cur;
    run_item(0, std::false_type{});
    for (int k = 1; k < nitem; ++k) run_item(k, std::true_type{});
    x_read((nitem - 1) & 1);  // the last item
    x_add();
    epi_write();
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) epi_read(r);
#pragma unroll
    for (int r = 0; r < 4; ++r) epi_store(r, prv);
  }
  WM_STAMP();
}


#endif
