// k_fwd_diag_t: the forward cells of one diagonal index as ONE persistent launch whose blocks run two wave TEAMS -- the forward
// counterpart of k_bwd_diag_ws (evf_bwd_fused.hip).  Replaces, for the recorded default-neuron cells, k_fwd_diag_p
// (evf_fwd_b3.hip), whose waves each ran load -> matrix phase -> epilogue for their own strip (8.3 vector instructions per MFMA,
// matrix pipe 26 % busy).
//
//   team M  (4 waves, one per SIMD): a STRIP of 2 rows x 32 pixels per wave and round -- the strip's 4 x 34 halo words (input
//           spikes, previous output spikes; requested one round ahead, committed to LDS already split into the four table
//           addresses of a pixel), the 108 (+108 recurrent) v_mfma_f32_32x32x16_bf16 of fwd_b3_body in the same order per
//           accumulator (bit-identical currents), software pipelined as in k_fwd_diag_p; the two accumulator tiles go to LDS as
//           [pixel][channel] rows.  ~50 vector instructions per 216 MFMAs.
//   team E  (8 waves, two per SIMD): one ROW of 32 pixels per wave and round.  A lane owns 4 channels of 4 pixels (pixel =
//           lane / 8 + 8 k): the previous potential arrives as FULL 128-byte lines (requested two rounds ahead into one of two
//           register sets), the current comes from team M's LDS tile in the same shape, the new potential leaves as full lines
//           (non-temporal) -- no staging round trip, 4 leak / threshold values per lane instead of 32, packed fp32 arithmetic,
//           the spike nibble by compare + add-with-carry.  The pixel's spike word is the OR of its 8 lanes' nibbles (three DPP
//           steps); the channel-major bit planes of the row (the weight gradient's operand) are the 32 x 32 bit TRANSPOSE of the
//           row's words: five butterfly stages (DPP, v_permlane16/32_swap, v_alignbit, v_bfi -- vector unit only) instead of a
//           ballot and two selects per element.  ~100 vector instructions per row; the tile read is its only LDS access.
// Hand-over per (team M wave, its two team E waves) through two LDS counters, no block barrier inside a cell: `full` = tiles the
// M wave has published (written after the tile: the LDS pipe runs a wave's instructions in order), `empty` = reads of them by
// the E waves (counted right behind their tile reads); M polls `empty` before it overwrites the tile -- the read is issued
// during the last matrix phase --, E polls `full` before it reads.  (First version: two s_barriers per round.  Ablation builds
// showed what they cost: team M waited at the early one for team E's tile read, ~1.8 k cycles per round.)
// Same arithmetic, element for element, as fwd_b3_body (spiking_submodules.py:96-126, :516-551).
#include "evf_fwd.h"

#define FT_EW 8
#ifndef FT_DEPTH2
#define FT_DEPTH2 1
#endif
#ifndef FT_ESLEEP
#define FT_ESLEEP 1  // team E's poll interval for a published tile, in 64-cycle units
#endif
#define FT_WAVES (4 + FT_EW)
#define FT_THREADS (64 * FT_WAVES)
#define FT_HW (4 * HALO_W)                              // halo pixels of a strip: rows y0 - 1 .. y0 + 2
#define FT_OFF_LUT 0                                    // byte -> 8 x bf16 {0, 1}: at LDS address 0, so (byte << 4) IS the address
#define FT_OFF_WFF 4096
#define FT_OFF_WREC (FT_OFF_WFF + WB3_BYTES)
#define FT_OFF_PAR (FT_OFF_WREC + WB3_BYTES)            // sigmoid(leak)[32], clamped thresh[32]
#define FT_OFF_PW (FT_OFF_PAR + 2 * C32 * 4)            // prediction head: 2 x 32 weights, 2 biases (+2 pad)
#define FT_OFF_HALO (FT_OFF_PW + (2 * C32 + 4) * 4)     // [4 M waves][x | z][FT_HW pixels][2 words]
#define FT_OFF_ACC (FT_OFF_HALO + 4 * 2 * FT_HW * 8)    // [4 strips][64 pixels][FW_SP floats]
#define FT_OFF_FLAG (FT_OFF_ACC + 4 * 64 * FW_SP * 4)   // full[4]: tiles team M's wave has published; empty[4][2]: tiles read by each of the strip's two team E waves
#define FT_OFF_PAR2 (FT_OFF_FLAG + 48)                  // PLIF: sigmoid(leak_pt)[32], sigmoid(add_pt)[32]
#define FT_OFF_P (FT_OFF_PAR2 + 2 * C32 * 4)            // PLIF: pooled pre-synaptic activity [2 tiles][4 strips][64 pixels]
#define FT_LDS (FT_OFF_P + 2 * 4 * 64 * 4)

__device__ uint4 ft_zero_page[256];  // 4 KiB of zeros: what a cell without previous state reads (no load under a branch, no select)

struct FtPlan {
  int njobs;
  int ntx, nyy;   // strips per row of tiles / strip rows per sample
  int nstrips;    // per cell
  int nquads;     // per cell: rounds of four strips
  int weight[FW_MAX_JOBS];  // relative cost of a round of the cell
  int total;      // sum of nquads * weight
  int ilv;        // k_fwd_win_t: 1 = a block's rounds of strips are blockIdx + k * gridDim (at any moment the blocks write ONE
                  // contiguous region of a pass's tape), 0 = a contiguous range per block
};

#ifdef FT_STAMPS  // phase stamps (debug build -DFT_STAMPS=<cells> through EVF_LIB; launches of that many cells): [blocks 0, 80, 160, 240][wave][128] shader-clock values
__device__ unsigned long long ft_stamps[4 * FT_WAVES * 128];
extern "C" int evf_debug_ft_stamps(void* dst) { return evf_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(ft_stamps), sizeof(ft_stamps))); }
#define FT_STAMP()                                                                                     \
  do {                                                                                                 \
    if (plan.njobs == FT_STAMPS && blockIdx.x % 80 == 0 && blockIdx.x < 320 && lane == 0 && nst < 128) \
      ft_stamps[((blockIdx.x / 80) * FT_WAVES + wv) * 128 + nst++] = __builtin_readcyclecounter();     \
  } while (0)
#else
#define FT_STAMP() do {} while (0)
#endif


// HARD: the reset rule of every cell of the launch; FULL: H even and W a multiple of 32 (no partial strips); PLIF: every cell
// carries the pre-synaptic trace (spiking_submodules.py:191-227, :618-657) -- team M pools the strip's input spike counts from
// its halo words (mean_c |x| = popcount / 32, AvgPool3x3 over the zero-padded halo) into a tile beside the accumulators, team E
// reads the previous trace like the previous potential, updates it, subtracts sigma(add_pt) * pt' from the current and stores
// pt' and the pooled activity (the backward's operands).  Same expressions as fwd_b3_body<.., true>: bit-identical.
//
// WIN (k_fwd_win_t): ONE feed-forward layer, the passes of a window back to back -- the iteration space is (round of four strips,
// pass) with the pass running fastest.  A feed-forward cell (t, l) needs layer l - 1 at pass t (an earlier launch) and its OWN
// pixels' state at t - 1 only, so team E keeps potential, trace and the pixel's previous spikes in REGISTERS from one pass to the
// next (the register set the diagonal form loads every round is loaded once per round of strips, two rounds of strips ahead) and
// writes the tape only: per pixel and pass 128 (+128 + 4 PLIF) bytes written and the 4-byte input words read, instead of the
// potential (+ trace) read back as well.  Team M is the diagonal form's with the halo source and the pass advancing per round.
// Same expressions in the same order: bit-identical to the cells launched one by one.
// XL (PLIF instantiations with the hard reset): XLIF cells -- a compile-time switch; as a run-time (cell-uniform) branch it cost the
// PLIF chain kernel 3 % (the round holds both formulas' values live: 168 registers, spills).
// XL = 2: ALIF cells (FwJob::xl): the XLIF arithmetic with the trace driven by the lane's own previous spike bits instead of the pooled activity.
template <bool HARD, bool FULL, bool PLIF, bool WIN, class JOBS, class WT, int XL = 0>
__device__ __forceinline__ void ft_body(const JOBS& jobs, const FtPlan& plan, const int B, const int H, const int W, const WT& wt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* s_lut = (uint4*)(smem + FT_OFF_LUT);
  uint4* s_wff = (uint4*)(smem + FT_OFF_WFF);
  uint4* s_wrec = (uint4*)(smem + FT_OFF_WREC);
  float* s_par = (float*)(smem + FT_OFF_PAR);
  float* s_pw = (float*)(smem + FT_OFF_PW);
  float* s_par2 = (float*)(smem + FT_OFF_PAR2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nW = (W + 31) / 32;
  int nst = 0;
  (void)nst;
  FT_STAMP();
  if (tid < 256) {  // byte -> 8 x bf16 {0, 1.0}
    const uint32_t t = tid;
    auto pr8 = [&](int e) { return ((t >> e) & 1u) * 0x3F80u | (((t >> (e + 1)) & 1u) * 0x3F80u) << 16; };
    s_lut[tid] = make_uint4(pr8(0), pr8(2), pr8(4), pr8(6));
  }
  if (tid < 12) ((uint32_t*)(smem + FT_OFF_FLAG))[tid] = 0u;
  int tile0 = 0;  // tiles (rounds) of this block so far: the flag counters run on across its cells
  bool first_cell = true;
  const long lo = ((long)blockIdx.x * plan.total) / gridDim.x, hi = ((long)(blockIdx.x + 1) * plan.total) / gridDim.x;
  long cell0 = 0;  // weighted start of the cell
  for (int c = 0; c < plan.njobs; ++c) {
    const int wgt = plan.weight[c];
    const long cw = (long)plan.nquads * wgt;
    // rounds of this cell whose weighted start S = cell0 + i * wgt lies in [lo, hi)
    const long a0 = lo - cell0, a1 = hi - cell0;
    cell0 += cw;
    int i0 = a0 <= 0 ? 0 : (int)((a0 + wgt - 1) / wgt), i1 = a1 <= 0 ? 0 : (int)((a1 + wgt - 1) / wgt);
    i0 = min(i0, plan.nquads), i1 = min(i1, plan.nquads);
    // round of strips number k of this block (k in [i0, i1)) as an index of the cell's rounds
    const bool ilv = WIN && plan.ilv != 0;
    if (ilv) i0 = 0, i1 = ((int)blockIdx.x < plan.nquads) ? (plan.nquads - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int q_mul = ilv ? (int)gridDim.x : 1, q_add = ilv ? (int)blockIdx.x : 0;
    auto qa = [&](int k) { return q_add + k * q_mul; };
    if (i0 >= i1) continue;  // (block-uniform)
    const FwJob& J = jobs.j[c];
    const bool rec = !WIN && J.wrec != nullptr;
    // A feed-forward cell leaves the recurrent weights' 54 KiB unused: its tiles alternate between FT_OFF_ACC and that region,
    // so team M may run TWO rounds ahead of team E (with one tile per strip the two teams took turns waiting for each other:
    // phase stamps showed team M 3-6 k cycles per round at the `empty` poll although team E idles half of its round).
    const int depth = (FT_DEPTH2 && !rec) ? 2 : 1;
    if (!first_cell) __syncthreads();  // every wave is done with the previous cell's weights, parameters and tiles
    first_cell = false;
    for (int u = wv; u < NFRAG; u += FT_WAVES) b3_glds16(J.wff + u * 64 + lane, s_wff + u * 64);
    if (rec)
      for (int u = wv; u < NFRAG; u += FT_WAVES) b3_glds16(J.wrec + u * 64 + lane, s_wrec + u * 64);
    if (tid < C32) {
      s_par[tid] = b3_sigmoid(J.leak[tid]);            // torch.sigmoid(self.leak)     spiking_submodules.py:111/:536
      s_par[C32 + tid] = fmaxf(J.thresh[tid], 0.01f);  // self.thresh.clamp_min(0.01)  :108/:533
    }
    if (PLIF && tid < C32) {
      s_par2[tid] = evf_plif_sigmoid(J.leak_pt[tid]);
      // (XLIF cell: the slot holds max(t1, 0) -- self.t1.clamp_min(0), spiking_submodules.py:365/:810 -- and the cell's `xl` flag, a
      // block-uniform scalar, picks the formula: the two neuron models share every register)
      s_par2[C32 + tid] = XL != 0 ? fmaxf(J.add_pt[tid], 0.f) : evf_plif_sigmoid(J.add_pt[tid]);
    }
    if (J.pr.w) {
      if (tid < 2 * C32) s_pw[tid] = J.pr.w[tid];
      if (tid < 2) s_pw[2 * C32 + tid] = J.pr.bias[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FT_STAMP();
    int NP = 1;  // WIN: passes of the window (round r = (round of strips r / NP, pass r % NP))
    if constexpr (WIN) NP = wt.np;
    const int n = (i1 - i0) * NP;  // rounds of this cell in this block; round r: team M strips 4 (i0 + r) + 0..3, team E those of round r - 1
    const uint32_t* __restrict__ x = J.x;
    const uint32_t* __restrict__ z_prev = J.z_prev;
    const float* __restrict__ v_prev = J.v_prev;
    float* __restrict__ v_out = J.v_out;
    // (every field of the cell the round loops touch, read from the argument table ONCE: as `J.z_out` beside its use each was a
    // scalar load + s_waitcnt lgkmcnt(0) per round -- four dependent round trips in team E's epilogue)
    uint32_t* __restrict__ z_out = J.z_out;
    uint32_t* __restrict__ zT_out = J.zT_out;
    float* __restrict__ flow_out = J.pr.flow;
    const float* __restrict__ pt_prev = PLIF ? J.pt_prev : nullptr;
    float* __restrict__ pt_out = PLIF ? J.pt_out : nullptr;
    float* __restrict__ P_out = PLIF ? J.P_out : nullptr;
    const bool has_pred = J.pr.w != nullptr;
    constexpr bool xl = PLIF && XL != 0, al = PLIF && XL == 2;  // (XLIF / ALIF cells, see FwJob: the launcher picks the instantiation by the cells' flag)
    const int nstrips = plan.nstrips;

    if (wv < 4) {
      // =============================================== team M ===============================================================
#ifdef FT_MPRIO  // (A/B builds)
      __builtin_amdgcn_s_setprio(FT_MPRIO);
#endif
      const int i = lane & 31, kg = lane >> 5;
      uint32_t* s_hx = (uint32_t*)(smem + FT_OFF_HALO) + wv * (2 * FT_HW * 2);
      uint32_t* s_hz = s_hx + FT_HW * 2;
      float* s_acc0 = (float*)(smem + FT_OFF_ACC) + wv * (64 * FW_SP);
      float* s_acc1 = depth == 2 ? (float*)(smem + FT_OFF_WREC) + wv * (64 * FW_SP) : s_acc0;
      float* s_Pm = (float*)(smem + FT_OFF_P) + wv * 64;
      int hro[3], hco[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int l = min(lane + 64 * q, FT_HW - 1);
        hro[q] = l / HALO_W, hco[q] = l - hro[q] * HALO_W;
      }
      uint32_t hx[3], hz[3];
      bool hin[3];
      const uint32_t* zsrc = z_prev ? z_prev : x;  // no load under a branch: clamped address, select afterwards
      // The strip whose halo is requested next, as scalars: two integer divisions by run-time values once per cell, then + 4
      // strips per round by compares (a strip index past the cell's last one keeps a valid address: it is never committed).
      // The requests themselves -- ~15 vector instructions and two loads per third of the halo -- ride behind the MFMAs of the
      // matrix phase (conv_phase's `side`): in front of it they were 0.5-1.1 k cycles of every round of the team that sets the
      // launch's pace (phase stamps).
      int gtx, gyy, gb;
      {
        const int sk = min(4 * qa(i0) + wv, nstrips - 1);
        gtx = sk % plan.ntx;
        const int rr = sk / plan.ntx;
        gyy = rr % plan.nyy, gb = rr / plan.nyy;
      }
      int gq = i0;  // WIN: the round of strips the request stream is at
      auto geom_adv = [&]() {
        gtx += 4;
        const int c = (gtx >= plan.ntx ? 1 : 0) + (gtx >= 2 * plan.ntx ? 1 : 0) + (gtx >= 3 * plan.ntx ? 1 : 0) + (gtx >= 4 * plan.ntx ? 1 : 0);
        gtx -= c * plan.ntx, gyy += c;
        const int d = (gyy >= plan.nyy ? 1 : 0) + (gyy >= 2 * plan.nyy ? 1 : 0) + (gyy >= 3 * plan.nyy ? 1 : 0) + (gyy >= 4 * plan.nyy ? 1 : 0);
        gyy -= d * plan.nyy, gb += d;
      };
      const uint32_t* __restrict__ xf = x;  // the input words the next halo request reads (WIN: those of pass ft)
      int ft = 0, mq = 0, mt = 0;            // WIN: pass of the next request; round of strips / pass of the round in work
      if constexpr (WIN) xf = wt.x[0];
      auto halo_fetch_q = [&](int q) {
        const int y0 = 2 * gyy, x0 = gtx * TW, b = min(gb, B - 1);
        const int ya = y0 + hro[q] - 1, xa = x0 + hco[q] - 1;
        hin[q] = ya >= 0 && ya < H && xa >= 0 && xa < W;
        const long p = ((long)b * H + min(max(ya, 0), H - 1)) * W + min(max(xa, 0), W - 1);
        hx[q] = xf[p];
        if constexpr (WIN) hz[q] = 0u; else hz[q] = zsrc[p];
      };
      auto round_adv = [&]() {  // the request stream's next round: WIN = the same strips at the next pass, the next strips after the last
        if constexpr (WIN) {
          ft = ft + 1 < NP ? ft + 1 : 0;
          xf = wt.x[ft];
          if (ft != 0) return;
          if (ilv) {  // (two divisions per NP rounds)
            ++gq;
            const int sk = min(4 * qa(gq) + wv, nstrips - 1);
            gtx = sk % plan.ntx;
            const int rr = sk / plan.ntx;
            gyy = rr % plan.nyy, gb = rr / plan.nyy;
            return;
          }
        }
        geom_adv();
      };
#pragma unroll
      for (int q = 0; q < 3; ++q) halo_fetch_q(q);
      round_adv();
      const unsigned fl_full = FT_OFF_FLAG + 4 * wv, fl_empty = FT_OFF_FLAG + 16 + 8 * wv;  // (LDS byte addresses; `empty`: one counter per row's wave)
      for (int r = 0; r < n; ++r) {
        FT_STAMP();
        const bool valid = 4 * qa(i0 + (WIN ? mq : r)) + wv < nstrips;  // (wave-uniform)
        if constexpr (WIN) {
          if (++mt == NP) mt = 0, ++mq;
        }
        // ---- commit the strip's halo words: per pixel two words = the table ADDRESSES (byte << 4) of its four channel
        // bytes, [kg][K half m] -- the matrix phase reads one word per (row, tap) and needs one vector instruction per look-up
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int l = lane + 64 * q;
          const uint32_t wx = hin[q] ? hx[q] : 0u, wz = (hin[q] && z_prev) ? hz[q] : 0u;
          if (l < FT_HW) {
            *(uint2*)(s_hx + 2 * l) = make_uint2((wx & 0x00FF00FFu) << 4, ((wx >> 8) & 0x00FF00FFu) << 4);
            if constexpr (!WIN) *(uint2*)(s_hz + 2 * l) = make_uint2((wz & 0x00FF00FFu) << 4, ((wz >> 8) & 0x00FF00FFu) << 4);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x16 acc0 = {0}, acc1 = {0};
        unsigned long long ev = 0ull;  // team E's two read counters of this wave's tiles, requested during the last matrix phase
        // Matrix phase, software pipelined (as k_fwd_diag_p): the two A fragments (table look-ups) and the three weight
        // fragments of stage g + 1 are requested BEFORE the 6 MFMAs of stage g and pinned there.
        auto conv_phase = [&](const uint32_t* __restrict__ sh, const uint4* __restrict__ sw, auto last_tag, auto side_tag) {
          constexpr bool LAST = decltype(last_tag)::value, SIDE = decltype(side_tag)::value;
          uint32_t hw[4][3];
#pragma unroll
          for (int rho = 0; rho < 4; ++rho)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) hw[rho][dx] = sh[2 * (rho * HALO_W + i + dx) + kg];
#ifndef FT_PF
#define FT_PF 2  // operand sets in flight: stage g + FT_PF - 1 is requested before the MFMAs of stage g
#endif
          uint4 af[FT_PF][2], wf[FT_PF][3];  // [stage slot][row] / [stage slot][term]; a stage = (tap, K half m): 6 MFMAs
          auto fetch = [&](int g) {
            const int sp = g % FT_PF, tau = g >> 1, m = g & 1, dy = tau / 3, dx = tau % 3;
#ifdef FT_PROBE_NOLUT  // (probe build: every lane reads table entry 0 -- no data-dependent address, no bank conflict: 38.8
                       //  against 44.9 us per launch.  Reading a third of the fragments less -- the upper row's fragment of
                       //  tap (dy, dx, m) is the lower row's of tap (dy - 1, dx, m), kept in a ring of six register sets --
                       //  gained 0.3 us: it is the chain commit -> word -> look-up -> first MFMA at the head of a phase, not
                       //  the number of look-ups.  Committing and reading the NEXT round's words behind this round's last
                       //  stages (second halo buffer on feed-forward cells) needs ~40 more registers in the matrix phase:
                       //  60-80 spilled at the 168 a wave of this 12-wave block may hold.)
            const uint32_t a0 = 0u, a1 = 0u;
            asm volatile("" ::"v"(hw[dy][dx]), "v"(hw[dy + 1][dx]));
#else
            const uint32_t a0 = m ? (hw[dy][dx] >> 16) : (hw[dy][dx] & 0xFFFFu);
            const uint32_t a1 = m ? (hw[dy + 1][dx] >> 16) : (hw[dy + 1][dx] & 0xFFFFu);
#endif
            af[sp][0] = *(const uint4*)(smem + FT_OFF_LUT + a0);
            af[sp][1] = *(const uint4*)(smem + FT_OFF_LUT + a1);
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) wf[sp][t3] = sw[(g * 3 + t3) * 64 + lane];
          };
#pragma unroll
          for (int g = 0; g < FT_PF - 1; ++g) fetch(g);
#pragma unroll
          for (int g = 0; g < 18; ++g) {
            const int sp = g % FT_PF;
            if (g + FT_PF - 1 < 18) fetch(g + FT_PF - 1);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 a0 = *(const bf16x8*)&af[sp][0], a1 = *(const bf16x8*)&af[sp][1];
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
              const bf16x8 bw = *(const bf16x8*)&wf[sp][t3];
#ifndef FT_PROBE_NOMFMA
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, a0, acc0, 0, 0, 0);  // weights as A: transposed product
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, a1, acc1, 0, 0, 0);
#else  // (probe build: the operand reads stay, the matrix pipe is idle)
              asm volatile("" ::"v"(bw), "v"(a0), "v"(a1));
#endif
            }
            // (the counter read rides in the LDS queue behind the operand reads: its latency is covered by the last six stages)
            if (LAST && g == 11) asm volatile("ds_read_b64 %0, %1" : "=v"(ev) : "v"(fl_empty) : "memory");
            // the next round's halo words, a third at a time behind this stage's MFMAs: they land during the rest of the phase
            if (SIDE && (g == 2 || g == 5 || g == 8)) halo_fetch_q((g - 2) / 3);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if (valid) {
#ifdef FT_PROBE_NOCONV  // (probe build: team M only requests halos and publishes tiles -- team E with the LDS pipe to itself)
#pragma unroll
          for (int q = 0; q < 3; ++q) halo_fetch_q(q);
          asm volatile("ds_read_b64 %0, %1" : "=v"(ev) : "v"(fl_empty) : "memory");
#else
          if (rec) {
            conv_phase(s_hx, s_wff, std::false_type{}, std::true_type{});
            conv_phase(s_hz, s_wrec, std::true_type{}, std::false_type{});
          } else {
            conv_phase(s_hx, s_wff, std::true_type{}, std::true_type{});
          }
#endif
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ev)::"memory");
        } else {
#pragma unroll
          for (int q = 0; q < 3; ++q) halo_fetch_q(q);
        }
        round_adv();
        FT_STAMP();
        // ---- the tile this round's buffer held before (round r - depth) must have been read by both of team E's waves
        // (one counter per wave: with their sum and two tiles in flight, a wave two reads ahead would cover for the other one)
        const uint32_t need = (uint32_t)(tile0 + max(r - (depth - 1), 0));
        float* s_acc = (r & 1) ? s_acc1 : s_acc0;
        while (min((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ev),
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ev >> 32))) < need) {
          __builtin_amdgcn_s_sleep(1);
          asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ev) : "v"(fl_empty) : "memory");
        }
        if (valid) {
          // ---- the two accumulator tiles -> LDS, [pixel][channel] rows: lane = pixel i of rows y0, y0 + 1; register
          // r = 4q + e is channel 8q + e + 4kg
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const f32x16& acc = m ? acc1 : acc0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *(float4*)(s_acc + (m * 32 + i) * FW_SP + 8 * q + 4 * kg) =
                  make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
          }
          if (PLIF) {  // lane = pixel (row lane / 32, column lane % 32) of the strip: the 9 halo words around it, still committed
            const int pm = lane >> 5;
            int cnt = 0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) {
                const uint2 hw2 = *(const uint2*)(s_hx + 2 * ((pm + dy) * HALO_W + i + dx));
                cnt += __popc(hw2.x) + __popc(hw2.y);
              }
            s_Pm[((depth == 2 && (r & 1)) ? 256 : 0) + lane] = ((float)cnt / 32.0f) / 9.0f;  // as fwd_b3_body
          }
        }
        // publish: the LDS pipe executes a wave's instructions in order, so the counter is written after the tile
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        {
          const uint32_t done = (uint32_t)(tile0 + r + 1);
          asm volatile("ds_write_b32 %0, %1" ::"v"(fl_full), "v"(done) : "memory");
        }
        FT_STAMP();
      }
    } else {
      // =============================================== team E ===============================================================
#ifdef FT_EPRIO  // (A/B builds)
      __builtin_amdgcn_s_setprio(FT_EPRIO);
#endif
      const int e = wv - 4, sidx = e >> 1, m = e & 1;     // strip slot (= team M's wave) and row of the strip
      const int p8 = lane >> 3, c4 = (lane & 7) * 4;       // pixels p8 + 8k (k = 0..3), channels c4 .. c4 + 3
      const float* s_accE0 = (const float*)(smem + FT_OFF_ACC) + sidx * (64 * FW_SP) + (m * 32) * FW_SP;
      const float* s_accE1 = depth == 2 ? (const float*)(smem + FT_OFF_WREC) + sidx * (64 * FW_SP) + (m * 32) * FW_SP : s_accE0;
      float lam[4], th[4], oml[4];
      {
        const float4 l4 = *(const float4*)(s_par + c4), t4 = *(const float4*)(s_par + C32 + c4);
        lam[0] = l4.x, lam[1] = l4.y, lam[2] = l4.z, lam[3] = l4.w;
        th[0] = t4.x, th[1] = t4.y, th[2] = t4.z, th[3] = t4.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) oml[q] = 1.0f - lam[q];
      }
      const float* s_PE = (const float*)(smem + FT_OFF_P) + sidx * 64 + m * 32;
      // After the element loop the words of the wave's 32 pixels sit in the lanes 8g + 4 + kq (g = lane / 8, kq = lane & 3): lane
      // 8g + 4 + kq PUBLISHES pixel vi = g + 8 kq -- its z_out word, its flow, and (after the transpose) bit plane vi.  No LDS
      // on the way (team M's operand reads keep the LDS pipe ~85 % busy, and every trip through it -- memory or crossbar -- is one
      // more long latency in the epilogue's dependent chain): DPP and v_permlane*_swap stay on the vector unit.
      const int kq = lane & 3, vi = p8 + 8 * kq;
      const bool pub = (lane & 4) != 0;
      // butterfly constants of the 32 x 32 bit transpose: stage s swaps, between pixels i and i ^ s, the bit blocks (c & s) != (i & s)
      uint32_t bf_keep[5], bf_amt[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const int s = 16 >> t;
        const uint32_t mlo = s == 16 ? 0x0000FFFFu : s == 8 ? 0x00FF00FFu : s == 4 ? 0x0F0F0F0Fu : s == 2 ? 0x33333333u : 0x55555555u;
        bf_keep[t] = (vi & s) ? ~mlo : mlo;
        bf_amt[t] = (vi & s) ? (uint32_t)s : (uint32_t)(32 - s);
      }
#ifdef FT_PROBE_NOVPREV  // (probe build: every cell reads the zero page instead of its previous potential)
      const bool has_v = false, has_z = z_prev != nullptr;
#else
      const bool has_v = v_prev != nullptr, has_z = z_prev != nullptr;
#endif
      const bool has_pt = PLIF && pt_prev != nullptr;
      const unsigned lane_off = has_v ? (unsigned)(p8 * C32 + c4) : 0u;  // floats
      const unsigned zlane = has_z ? (unsigned)p8 : 0u;
      auto geom = [&](int si, int& b, int& row, int& tx) {
        tx = si % plan.ntx;
        const int rr = si / plan.ntx, yy = rr % plan.nyy;
        b = rr / plan.nyy, row = 2 * yy + m;
      };
      // previous potential (full lines) and previous spike words of the row, requested TWO rounds ahead into one of two register
      // sets (requested one round ahead the loads had ~0.3 of a round to land and the epilogue waited for them: 8.5 k cycles per
      // round against 7.3 k of MFMAs, phase stamps)
      auto e_fetch = [&](int qd, float4 (&vp)[4], uint32_t (&zq)[4], float4 (&pq)[PLIF ? 4 : 1]) {
        const int sk = min(4 * qa(qd) + sidx, nstrips - 1);
        int b, row, tx;
        geom(sk, b, row, tx);
        const long pb = ((long)b * H + (FULL ? row : min(row, H - 1))) * W + tx * TW;
        const uint32_t* zs = has_z ? z_prev + pb : (const uint32_t*)ft_zero_page;
        const float* vs = has_v ? v_prev + pb * C32 : (const float*)ft_zero_page;
        const float* ps = has_pt ? pt_prev + pb * C32 : (const float*)ft_zero_page;
        // the previous spike words of the lane's four pixels: one dword load each (8 lanes per word; through a word per lane and
        // ds_bpermute they were one more LDS round trip in the epilogue's dependent chain)
        if (FULL) {
#pragma unroll
          for (int k = 0; k < 4; ++k) zq[k] = zs[zlane + (has_z ? 8 * k : 0)];
#pragma unroll
          for (int k = 0; k < 4; ++k) vp[k] = *(const float4*)(vs + lane_off + k * (8 * C32));
          if (PLIF) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pq[k] = *(const float4*)(ps + (has_pt ? (unsigned)(p8 * C32 + c4 + k * (8 * C32)) : 0u));
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) zq[k] = zs[has_z ? (unsigned)min(p8 + 8 * k, W - 1 - tx * TW) : 0u];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int pe = min(p8 + 8 * k, W - 1 - tx * TW);
            vp[k] = *(const float4*)(vs + (has_v ? (unsigned)(pe * C32 + c4) : 0u));
            if (PLIF) pq[k] = *(const float4*)(ps + (has_pt ? (unsigned)(pe * C32 + c4) : 0u));
          }
        }
      };
      const unsigned fl_full = FT_OFF_FLAG + 4 * sidx, fl_empty = FT_OFF_FLAG + 16 + 8 * sidx + 4 * m;  // (LDS byte addresses)
      // the epilogue of round r's strips (r = 0 .. n - 1); WIN: round of strips rq, pass t (r = rq * NP + t), and the state of
      // the pass stays in the register set for the next one
      auto e_round = [&](int r, int rq, int t, float4 (&vp)[4], uint32_t (&zq)[4], float4 (&pq)[PLIF ? 4 : 1]) {
        FT_STAMP();
        float* __restrict__ v_out_r = v_out;
        uint32_t* __restrict__ z_out_r = z_out;
        uint32_t* __restrict__ zT_out_r = zT_out;
        float* __restrict__ flow_out_r = flow_out;
        float* __restrict__ pt_out_r = pt_out;
        float* __restrict__ P_out_r = P_out;
        if constexpr (WIN) {
          v_out_r = wt.v_out[t], z_out_r = wt.z_out[t], zT_out_r = wt.zT_out[t], flow_out_r = wt.flow[t];
          if constexpr (PLIF) pt_out_r = wt.pt_out[t], P_out_r = wt.P_out[t];
        }
        const int si = 4 * qa(i0 + rq) + sidx;
        const bool valid = si < nstrips;  // (wave-uniform)
        int b, row, tx;
        geom(min(si, nstrips - 1), b, row, tx);
        const int x0 = tx * TW;
        const long pb = ((long)b * H + row) * W + x0;  // the row's first pixel
        {  // ---- wait for team M's wave to publish the tile of this round
          const uint32_t need = (uint32_t)(tile0 + r + 1);
          uint32_t fv;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fv) : "v"(fl_full) : "memory");
          while ((uint32_t)__builtin_amdgcn_readfirstlane((int)fv) < need) {
            __builtin_amdgcn_s_sleep(FT_ESLEEP);
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fv) : "v"(fl_full) : "memory");
          }
        }
        // PLIF: the per-channel constants are re-read from LDS and the transpose's masks rebuilt every round instead of living in
        // ~30 registers across it -- what lets TWO sets of previous potential + trace (2 x 36 registers) stay in flight without
        // spilling at the 168 registers a wave of this block may hold (one set: team E's loads covered half of a round,
        // 3.1 TB/s; phase stamps: 3-11 k cycles per round at the issue of the next set's loads)
        float lamL[4], thL[4], omlL[4], lptL[4] = {0.f, 0.f, 0.f, 0.f}, aptL[4] = {0.f, 0.f, 0.f, 0.f};
        int viL = vi;
        if constexpr (PLIF) {
          unsigned offL = (unsigned)c4;
          asm volatile("" : "+v"(offL), "+v"(viL));  // (opaque per round: neither the reads nor the masks are hoisted out of the loop)
          const float4 l4 = *(const float4*)(s_par + offL), t4 = *(const float4*)(s_par + C32 + offL);
          const float4 p4_ = *(const float4*)(s_par2 + offL), a4_ = *(const float4*)(s_par2 + C32 + offL);
          lamL[0] = l4.x, lamL[1] = l4.y, lamL[2] = l4.z, lamL[3] = l4.w;
          thL[0] = t4.x, thL[1] = t4.y, thL[2] = t4.z, thL[3] = t4.w;
          lptL[0] = p4_.x, lptL[1] = p4_.y, lptL[2] = p4_.z, lptL[3] = p4_.w;
          aptL[0] = a4_.x, aptL[1] = a4_.y, aptL[2] = a4_.z, aptL[3] = a4_.w;
#pragma unroll
          for (int q = 0; q < 4; ++q) omlL[q] = 1.0f - lamL[q];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) lamL[q] = lam[q], thL[q] = th[q], omlL[q] = oml[q];
        }
        float4 a4[4];
        uint32_t zw[4];
        const float* s_accE = (r & 1) ? s_accE1 : s_accE0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a4[k] = *(const float4*)(s_accE + (p8 + 8 * k) * FW_SP + c4);
          zw[k] = zq[k];  // previous word of pixel p8 + 8k
        }
        float Pk[4] = {0.f, 0.f, 0.f, 0.f}, Pvi = 0.f;  // pooled activity of the lane's four pixels / of the pixel it publishes
        if (PLIF) {
          const float* sp = s_PE + ((depth == 2 && (r & 1)) ? 256 : 0);
#pragma unroll
          for (int k = 0; k < 4; ++k) Pk[k] = sp[p8 + 8 * k];
          Pvi = sp[vi];
        }
        // the tile is in (or on its way into) registers: the LDS pipe runs a wave's instructions in order, so the counter moves
        // after the reads above have taken their data -- team M may overwrite the tile once both rows' waves have counted
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane == 0) {
          const uint32_t one = 1u;
          asm volatile("ds_add_u32 %0, %1" ::"v"(fl_empty), "v"(one) : "memory");
        }
        FT_STAMP();
#ifdef FT_PROBE_NOE  // (probe build: team E only keeps the barriers and its loads)
        if (false) {
#else
        if (valid) {
#endif
          const bool rowok = FULL || row < H;
          uint32_t wk[4];
#ifdef FT_PROBE_NOELEM
          if (true) {
#pragma unroll
            for (int k = 0; k < 4; ++k) wk[k] = zw[k] ^ __float_as_uint(a4[k].x) ^ __float_as_uint(vp[k].x);
          } else
#endif
          if (HARD && FULL && !PLIF) {
            // The default cell on whole rows, written for the vector-issue budget (a vector instruction beside team M's MFMA
            // stream issues every ~8 cycles: this loop, not the matrix pipe, set the round time): packed fp32 pairs, the spike nibble by compare + add-with-carry (two
            // instructions per element, no subtraction: vo - th > 0 <=> vo > th for finite values), stores relative to a
            // uniform base.  Same products and sums in the same order as the general form below.
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 lam01 = {lam[0], lam[1]}, lam23 = {lam[2], lam[3]}, oml01 = {oml[0], oml[1]}, oml23 = {oml[2], oml[3]};
            float* vo_base = v_out_r + pb * C32;
            const unsigned st_off = (unsigned)(p8 * C32 + c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t zn = zw[k] >> c4;
              const float4 omz = make_float4(1.0f - (float)(zn & 1u), 1.0f - (float)((zn >> 1) & 1u), 1.0f - (float)((zn >> 2) & 1u),
                                             1.0f - (float)((zn >> 3) & 1u));
              const f2 v01 = {vp[k].x, vp[k].y}, v23 = {vp[k].z, vp[k].w}, c01 = {a4[k].x, a4[k].y}, c23 = {a4[k].z, a4[k].w};
              const f2 z01 = {omz.x, omz.y}, z23 = {omz.z, omz.w};
              const f2 o01 = (v01 * lam01) * z01 + oml01 * c01;  // :119/:544
              const f2 o23 = (v23 * lam23) * z23 + oml23 * c23;
              uint32_t nib = 0u;
              asm volatile(
                  "v_cmp_gt_f32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                  "v_cmp_gt_f32 vcc, %2, %6\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                  "v_cmp_gt_f32 vcc, %3, %7\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                  "v_cmp_gt_f32 vcc, %4, %8\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                  : "+v"(nib)
                  : "v"(o23.y), "v"(o23.x), "v"(o01.y), "v"(o01.x), "v"(th[3]), "v"(th[2]), "v"(th[1]), "v"(th[0])
                  : "vcc");
#ifndef FT_PROBE_NOSTORE
              evf_store_nt(vo_base + st_off + k * (8 * C32), make_float4(o01.x, o01.y, o23.x, o23.y));
#else
              asm volatile("" ::"v"(o01.x), "v"(o01.y), "v"(o23.x), "v"(o23.y));
#endif
              uint32_t w = nib << c4;
              if constexpr (WIN) vp[k] = make_float4(o01.x, o01.y, o23.x, o23.y), zq[k] = w;  // (the lane's own four channels of the word)
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x114, 0xF, 0xF, true);  // row_shr:4
              wk[k] = w;
            }
          } else if (HARD && PLIF) {
            // The PLIF cell with the hard reset in the same form: team E's vector instructions, not the bytes, set the pace of a
            // PLIF round (phase stamps of the window form: ~6 k cycles in this loop against 5 k of MFMAs) -- packed fp32 pairs for
            // the trace, the current and the potential, (1 - z) as a bit pattern (two instructions instead of convert + subtract),
            // the spike nibble by compare + add-with-carry.  Same products and sums in the same order as the general form below.
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 lam01 = {lamL[0], lamL[1]}, lam23 = {lamL[2], lamL[3]}, oml01 = {omlL[0], omlL[1]}, oml23 = {omlL[2], omlL[3]};
            const f2 lpt01 = {lptL[0], lptL[1]}, lpt23 = {lptL[2], lptL[3]}, apt01 = {aptL[0], aptL[1]}, apt23 = {aptL[2], aptL[3]};
            const f2 olp01 = {1.0f - lptL[0], 1.0f - lptL[1]}, olp23 = {1.0f - lptL[2], 1.0f - lptL[3]};
            float* vo_base = v_out_r + pb * C32;
            float* po_base = pt_out_r + pb * C32;
            const unsigned st_off = (unsigned)(p8 * C32 + c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const bool ok = FULL || (rowok && x0 + p8 + 8 * k < W);
              const uint32_t nzn = ~(zw[k] >> c4);  // 1 - z = 1.0f where the previous spike bit is clear
              const f2 z01 = {__uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)nzn, 0, 1) & 0x3F800000u),
                              __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)nzn, 1, 1) & 0x3F800000u)};
              const f2 z23 = {__uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)nzn, 2, 1) & 0x3F800000u),
                              __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)nzn, 3, 1) & 0x3F800000u)};
              const f2 P2 = {Pk[k], Pk[k]};
              const f2 pp01 = {pq[PLIF ? k : 0].x, pq[PLIF ? k : 0].y}, pp23 = {pq[PLIF ? k : 0].z, pq[PLIF ? k : 0].w};
              f2 po01, po23;
              if constexpr (al) {  // t * leak_t + (1 - leak_t) * z (:311 / :744): z01 / z23 hold 1 - z
                const f2 one2 = {1.0f, 1.0f};
                po01 = pp01 * lpt01 + olp01 * (one2 - z01), po23 = pp23 * lpt23 + olp23 * (one2 - z23);
              } else {
                po01 = pp01 * lpt01 + olp01 * P2, po23 = pp23 * lpt23 + olp23 * P2;  // evf_plif_trace, :212 / :642
              }
              const f2 a01 = {a4[k].x, a4[k].y}, a23 = {a4[k].z, a4[k].w};
              f2 c01, c23, t01 = {thL[0], thL[1]}, t23 = {thL[2], thL[3]};
              if constexpr (xl) {  // XLIF: the current stays ff + rec, the trace raises the threshold: t0 + t1 * pt_out, :419 / :864
                c01 = a01, c23 = a23;
                t01 = t01 + apt01 * po01, t23 = t23 + apt23 * po23;
              } else {
                c01 = a01 - apt01 * po01, c23 = a23 - apt23 * po23;                          // (ff + rec) - add_pt * pt_out, :220 / :650
              }
              const f2 v01 = {vp[k].x, vp[k].y}, v23 = {vp[k].z, vp[k].w};
              const f2 o01 = (v01 * lam01) * z01 + oml01 * c01;                             // :119/:544
              const f2 o23 = (v23 * lam23) * z23 + oml23 * c23;
              uint32_t nib = 0u;
              asm volatile(
                  "v_cmp_gt_f32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                  "v_cmp_gt_f32 vcc, %2, %6\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                  "v_cmp_gt_f32 vcc, %3, %7\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                  "v_cmp_gt_f32 vcc, %4, %8\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                  : "+v"(nib)
                  : "v"(o23.y), "v"(o23.x), "v"(o01.y), "v"(o01.x), "v"(t23.y), "v"(t23.x), "v"(t01.y), "v"(t01.x)
                  : "vcc");
              if (!FULL) nib = ok ? nib : 0u;
#ifndef FT_PROBE_NOSTORE
              if (ok) {
                evf_store_nt(vo_base + st_off + k * (8 * C32), make_float4(o01.x, o01.y, o23.x, o23.y));
                evf_store_nt(po_base + st_off + k * (8 * C32), make_float4(po01.x, po01.y, po23.x, po23.y));
              }
#else
              asm volatile("" ::"v"(o01.x), "v"(o01.y), "v"(o23.x), "v"(o23.y), "v"(po01.x), "v"(po01.y), "v"(po23.x), "v"(po23.y));
#endif
              uint32_t w = nib << c4;
              if constexpr (WIN) {  // the next pass's previous state (the lane's own four channels of the word)
                vp[k] = make_float4(o01.x, o01.y, o23.x, o23.y), zq[k] = w;
                pq[PLIF ? k : 0] = make_float4(po01.x, po01.y, po23.x, po23.y);
              }
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x114, 0xF, 0xF, true);  // row_shr:4
              wk[k] = w;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int p = p8 + 8 * k;
              const bool ok = FULL || (rowok && x0 + p < W);
              const float cu[4] = {a4[k].x, a4[k].y, a4[k].z, a4[k].w};
              const float v4[4] = {vp[k].x, vp[k].y, vp[k].z, vp[k].w};
              const float p4[4] = {PLIF ? pq[PLIF ? k : 0].x : 0.f, PLIF ? pq[PLIF ? k : 0].y : 0.f, PLIF ? pq[PLIF ? k : 0].z : 0.f,
                                   PLIF ? pq[PLIF ? k : 0].w : 0.f};
              const uint32_t zn = zw[k] >> c4;
              float vo4[4], po4[4];
              uint32_t nib = 0u;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float z = (float)((zn >> q) & 1u);
                float cur = cu[q];
                float th_e = thL[q], th_p = thL[q];  // threshold of the element now / at the previous pass (soft reset)
                po4[q] = 0.f;
                if (PLIF) {
                  po4[q] = evf_plif_trace(p4[q], lptL[q], al ? z : Pk[k]);  // :212 / :642 (ALIF: the own previous spike, :311 / :744)
                  if constexpr (xl) {  // XLIF: thresh = t0 + t1 * pt_out, :419 / :864; soft reset - z * (t0 + t1 * pt), :430 / :871
                    th_e = thL[q] + aptL[q] * po4[q];
                    th_p = thL[q] + aptL[q] * p4[q];
                  } else {
                    cur = cur - aptL[q] * po4[q];                       // (ff + rec) - add_pt * pt_out, :220 / :650
                  }
                }
                const float vo = HARD ? (v4[q] * lamL[q]) * (1.0f - z) + omlL[q] * cur    // :119/:544
                                      : v4[q] * lamL[q] + omlL[q] * cur - z * th_p;       // :121/:546
                const bool spike = ok && (vo - th_e) > 0.f;
                vo4[q] = vo;
                nib |= (spike ? 1u : 0u) << q;
              }
#ifndef FT_PROBE_NOSTORE
              if (ok) evf_store_nt(v_out_r + (pb + p) * C32 + c4, make_float4(vo4[0], vo4[1], vo4[2], vo4[3]));
              if (PLIF && ok) evf_store_nt(pt_out_r + (pb + p) * C32 + c4, make_float4(po4[0], po4[1], po4[2], po4[3]));
#else  // (probe build: the new potential is computed and dropped)
              asm volatile("" ::"v"(vo4[0]), "v"(vo4[1]), "v"(vo4[2]), "v"(vo4[3]));
#endif
              // the pixel's word = OR of its 8 lanes' nibbles: complete in lanes 4..7 of the group after three DPP steps
              uint32_t w = nib << c4;
              if constexpr (WIN) {  // the next pass's previous state (the lane's own four channels of the word)
                vp[k] = make_float4(vo4[0], vo4[1], vo4[2], vo4[3]), zq[k] = w;
                if constexpr (PLIF) pq[k] = make_float4(po4[0], po4[1], po4[2], po4[3]);
              }
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x114, 0xF, 0xF, true);  // row_shr:4
              wk[k] = w;
            }
          }
          FT_STAMP();
          const uint32_t word = kq == 0 ? wk[0] : kq == 1 ? wk[1] : kq == 2 ? wk[2] : wk[3];  // (publishing lanes: pixel vi)
          const bool okx = pub && rowok && (FULL || x0 + vi < W);
#ifndef FT_PROBE_NOZOUT
          if (okx) z_out_r[pb + vi] = word;
          if (PLIF && okx) P_out_r[pb + vi] = Pvi;
#else
          asm volatile("" ::"v"(word));
#endif
          if (has_pred) {  // (cell-uniform) the prediction head on this pixel's spike word, summed like evf_pred_fwd
            float s0 = 0.f, s1 = 0.f;
#pragma unroll 1
            for (int cc = 0; cc < C32; cc += 4) {
              const float4 pa = *(const float4*)(s_pw + cc), pb4 = *(const float4*)(s_pw + C32 + cc);
              const float pa4[4] = {pa.x, pa.y, pa.z, pa.w}, pb44[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float z = (float)((word >> (cc + q)) & 1u);
                s0 += z * pa4[q];
                s1 += z * pb44[q];
              }
            }
            if (okx) {
              const long hw = (long)H * W, qq = (long)row * W + x0 + vi;
              flow_out_r[(long)b * 2 * hw + qq] = tanhf(s0 + s_pw[2 * C32]);
              flow_out_r[((long)b * 2 + 1) * hw + qq] = tanhf(s1 + s_pw[2 * C32 + 1]);
            }
          }
          FT_STAMP();
#ifdef FT_PROBE_NOZT
          if (false) {
#else
          if (zT_out_r) {  // channel-major bit planes of the row = the transpose of its 32 words
#endif
            uint32_t a = word;
            uint32_t bf_keepL[5], bf_amtL[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
              if constexpr (PLIF) {
                const int s = 16 >> t;
                const uint32_t mlo = s == 16 ? 0x0000FFFFu : s == 8 ? 0x00FF00FFu : s == 4 ? 0x0F0F0F0Fu : s == 2 ? 0x33333333u : 0x55555555u;
                bf_keepL[t] = (viL & s) ? ~mlo : mlo;
                bf_amtL[t] = (viL & s) ? (uint32_t)s : (uint32_t)(32 - s);
              } else {
                bf_keepL[t] = bf_keep[t], bf_amtL[t] = bf_amt[t];
              }
            }
#define FT_BFLY(t_, partner_)                                                                                       \
  do {                                                                                                             \
    const uint32_t p_ = (uint32_t)(partner_);                                                                      \
    const uint32_t rot_ = __builtin_amdgcn_alignbit(p_, p_, bf_amtL[t_]);                                          \
    a = (a & bf_keepL[t_]) | (rot_ & ~bf_keepL[t_]);                                                               \
  } while (0)
            // pixel vi = p8 + 8 kq: vi ^ 16, vi ^ 8 flip kq (lanes ^ 2, ^ 1: quad permutes); vi ^ 4, ^ 2, ^ 1 flip p8 (lanes ^ 32, ^ 16, ^ 8)
            FT_BFLY(0, __builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
            FT_BFLY(1, __builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            {  // lanes ^ 32 and ^ 16 by v_permlane32_swap / v_permlane16_swap of two copies (vector unit only: no LDS-pipe round trip)
              const auto s32 = __builtin_amdgcn_permlane32_swap(a, a, false, false);  // [0] = (low half, low half), [1] = (high, high)
              FT_BFLY(2, (lane & 32) ? s32[0] : s32[1]);
              const auto s16 = __builtin_amdgcn_permlane16_swap(a, a, false, false);  // [0] = rows (0, 0, 2, 2), [1] = rows (1, 1, 3, 3)
              FT_BFLY(3, (lane & 16) ? s16[0] : s16[1]);
            }
            FT_BFLY(4, __builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xF, 0xF, true));  // row_ror:8
#undef FT_BFLY
            if (pub && rowok) zT_out_r[(((long)b * H + row) * C32 + vi) * nW + tx] = a;
          }
        }
        FT_STAMP();
        // this register set's next use: round r + 2.  Unconditional (past the range: the last round again, never used) -- under
        // `if (r + 1 < n)` the compiler's counter bookkeeping at the join made round r + 1 wait for THESE loads as well
#ifndef FT_PROBE_NOFETCH
        if constexpr (WIN) {  // (the set's next use: the first pass of the round of strips after the next)
          if (t == NP - 1) e_fetch(min(i0 + rq + 2, i1 - 1), vp, zq, pq);
        } else {
          e_fetch(min(i0 + r + 2, i1 - 1), vp, zq, pq);
        }
#endif
        FT_STAMP();
      };
      float4 vpA[4], vpB[4], pqA[PLIF ? 4 : 1], pqB[PLIF ? 4 : 1];
      uint32_t zqA[4], zqB[4];
      e_fetch(i0, vpA, zqA, pqA);
      e_fetch(min(i0 + 1, i1 - 1), vpB, zqB, pqB);
      if constexpr (WIN) {  // a round of strips keeps its register set for all its passes
        const int nq = i1 - i0;
        int r = 0;
        for (int rq = 0; rq < nq; rq += 2) {
#pragma unroll 1
          for (int t = 0; t < NP; ++t, ++r) e_round(r, rq, t, vpA, zqA, pqA);
          if (rq + 1 < nq) {
#pragma unroll 1
            for (int t = 0; t < NP; ++t, ++r) e_round(r, rq + 1, t, vpB, zqB, pqB);
          }
        }
      } else {
        for (int r = 0; r < n; r += 2) {  // (unrolled by two: the register sets swap roles, no moves of loaded registers)
          e_round(r, r, 0, vpA, zqA, pqA);
          if (r + 1 < n) e_round(r + 1, r + 1, 0, vpB, zqB, pqB);
        }
      }
    }
    tile0 += n;
  }
  FT_STAMP();
}

template <bool HARD, bool FULL, bool PLIF, int XL = 0>
__global__ __launch_bounds__(FT_THREADS) void k_fwd_diag_t(FwJobs jobs, FtPlan plan, int B, int H, int W) {
  ft_body<HARD, FULL, PLIF, false, FwJobs, FwWinNone, XL>(jobs, plan, B, H, W, FwWinNone{});
}

template <bool HARD, bool FULL, bool PLIF, int XL = 0>
__global__ __launch_bounds__(FT_THREADS) void k_fwd_win_t(FwJob1 job, FwWinTab wt, FtPlan plan, int B, int H, int W) {
  ft_body<HARD, FULL, PLIF, true, FwJob1, FwWinTab, XL>(job, plan, B, H, W, wt);
}

int evf_fwd_diag_t_launch(const FwJobs& jobs, int n, int B, int H, int W, void* stream) {
  if (n <= 0 || n > FW_MAX_JOBS || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  static bool attr_set = false;
  if (!attr_set) {
#define FT_ATTR(H_, F_, P_) \
  (void)hipFuncSetAttribute((const void*)k_fwd_diag_t<H_, F_, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS)
    FT_ATTR(true, true, false), FT_ATTR(true, false, false), FT_ATTR(false, true, false), FT_ATTR(false, false, false);
    FT_ATTR(true, true, true), FT_ATTR(true, false, true), FT_ATTR(false, true, true), FT_ATTR(false, false, true);
#undef FT_ATTR
    (void)hipFuncSetAttribute((const void*)k_fwd_diag_t<true, true, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS);
    (void)hipFuncSetAttribute((const void*)k_fwd_diag_t<true, false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS);
    (void)hipFuncSetAttribute((const void*)k_fwd_diag_t<true, true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS);
    (void)hipFuncSetAttribute((const void*)k_fwd_diag_t<true, false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS);
    attr_set = true;
  }
  // relative cost of a round: feed-forward / recurrent cell / feed-forward cell with the prediction head in team E's epilogue (its
  // ~320 vector instructions per row make team E the slower team there: 8 k cycles per round against 5.8 k, phase stamps)
  // (EVF_FT_W=ff,rec,pred: measurements)
  static int w_ff = 0, w_rec = 0, w_pred = 0;
  if (!w_ff) {
    w_ff = 8, w_rec = 14, w_pred = 13;
    const char* e = getenv("EVF_FT_W");
    int a = 0, b = 0, c = 0;
    if (e && sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0 && a < 64 && b < 64 && c < 64) w_ff = a, w_rec = b, w_pred = c;
  }
  int nhard = 0, nplif = 0;
  for (int k = 0; k < n; ++k) nhard += jobs.j[k].hard_reset ? 1 : 0, nplif += jobs.j[k].leak_pt ? 1 : 0;
  if ((nhard != 0 && nhard != n) || (nplif != 0 && nplif != n)) return EVF_EINVAL;
  for (int k = 0; k < n; ++k)  // XLIF cells: one kind per launch, hard reset (the caller then runs the cells one by one: evf_fwd_b3.hip)
    if (jobs.j[k].xl != jobs.j[0].xl || (jobs.j[k].xl && !jobs.j[k].hard_reset)) return EVF_EINVAL;
  FtPlan plan;
  plan.njobs = n, plan.ntx = evf_cdiv(W, TW), plan.nyy = evf_cdiv(H, 2);
  const long nstrips = (long)plan.ntx * plan.nyy * B;
  if (nstrips >= (1L << 28)) return EVF_EINVAL;
  plan.nstrips = (int)nstrips;
  plan.nquads = evf_cdiv(nstrips, 4);
  long total = 0;
  static const bool w_env = getenv("EVF_FT_W") != nullptr;
  // (PLIF cells: team E moves twice the state and sets the pace of every round -- recurrent and feed-forward cells cost nearly
  //  the same there: 8 : 10 : 10 measured best at 260 x 346 x B4, 171 against 186 us per launch)
  const int wf = (nplif && !w_env) ? 8 : w_ff, wr = (nplif && !w_env) ? 10 : w_rec, wp = (nplif && !w_env) ? 10 : w_pred;
  for (int k = 0; k < FW_MAX_JOBS; ++k) {
    plan.weight[k] = (k < n && jobs.j[k].wrec) ? wr : ((k < n && jobs.j[k].pr.w) ? wp : wf);
    if (k < n) total += (long)plan.nquads * plan.weight[k];
  }
  if (total >= (1L << 31)) return EVF_EINVAL;
  plan.total = (int)total;
  plan.ilv = 0;
  const long nq = (long)plan.nquads * n;
  const int nblk = (int)(nq / 2 < ncu ? (nq + 1) / 2 : ncu);  // (tiny launches: at least two rounds per block)
  const bool full = (H % 2 == 0) && (W % TW == 0);
  hipStream_t st = EVF_STREAM(stream);
#define FT_GO(HARD_, FULL_)                                                                                                  \
  do {                                                                                                                       \
    if (nplif && jobs.j[0].xl == 2)                                                                                          \
      hipLaunchKernelGGL((k_fwd_diag_t<true, FULL_, true, 2>), dim3(nblk), dim3(FT_THREADS), FT_LDS, st, jobs, plan, B, H, W); \
    else if (nplif && jobs.j[0].xl)                                                                                          \
      hipLaunchKernelGGL((k_fwd_diag_t<true, FULL_, true, 1>), dim3(nblk), dim3(FT_THREADS), FT_LDS, st, jobs, plan, B, H, W); \
    else if (nplif)                                                                                                          \
      hipLaunchKernelGGL((k_fwd_diag_t<HARD_, FULL_, true>), dim3(nblk), dim3(FT_THREADS), FT_LDS, st, jobs, plan, B, H, W);  \
    else                                                                                                                     \
      hipLaunchKernelGGL((k_fwd_diag_t<HARD_, FULL_, false>), dim3(nblk), dim3(FT_THREADS), FT_LDS, st, jobs, plan, B, H, W); \
  } while (0)
  if (nhard) {
    if (full) FT_GO(true, true); else FT_GO(true, false);
  } else {
    if (full) FT_GO(false, true); else FT_GO(false, false);
  }
#undef FT_GO
  return evf_status();
}

int evf_fwd_win_is_chain(const FwJob* c, int n) {
  if (n < 2 || n > FW_WIN_MAX) return 0;
  if (c[0].wrec) return 0;
  if (c[0].xl && !c[0].hard_reset) return 0;  // (XLIF / ALIF cells with the soft reset: the one-cell kernel)
  for (int k = 1; k < n; ++k) {
    const FwJob &a = c[k - 1], &b = c[k];
    if (b.wrec || b.wff != a.wff || b.leak != a.leak || b.thresh != a.thresh || b.hard_reset != a.hard_reset ||
        b.leak_pt != a.leak_pt || b.add_pt != a.add_pt || b.xl != a.xl || b.pr.w != a.pr.w || b.pr.bias != a.pr.bias)
      return 0;
    if (b.v_prev != a.v_out || b.z_prev != a.z_out || b.pt_prev != a.pt_out) return 0;
    if ((b.zT_out == nullptr) != (a.zT_out == nullptr)) return 0;
  }
  return 1;
}

// The chain `cells[0..n)` (evf_fwd_win_is_chain) as one launch: a block owns rounds of four strips and runs their n passes.
int evf_fwd_win_t_launch(const FwJob* cells, int n, int B, int H, int W, void* stream) {
  if (!cells || !evf_fwd_win_is_chain(cells, n) || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  static bool attr_set = false;
  if (!attr_set) {
#define FT_ATTR(H_, F_, P_) \
  (void)hipFuncSetAttribute((const void*)k_fwd_win_t<H_, F_, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS)
    FT_ATTR(true, true, false), FT_ATTR(true, false, false), FT_ATTR(false, true, false), FT_ATTR(false, false, false);
    FT_ATTR(true, true, true), FT_ATTR(true, false, true), FT_ATTR(false, true, true), FT_ATTR(false, false, true);
#undef FT_ATTR
    (void)hipFuncSetAttribute((const void*)k_fwd_win_t<true, true, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS);
    (void)hipFuncSetAttribute((const void*)k_fwd_win_t<true, false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS);
    (void)hipFuncSetAttribute((const void*)k_fwd_win_t<true, true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS);
    (void)hipFuncSetAttribute((const void*)k_fwd_win_t<true, false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS);
    attr_set = true;
  }
  FwJob1 job;
  job.j[0] = cells[0];
  FwWinTab wt;
  wt.np = n, wt.pad_ = 0;
  for (int k = 0; k < FW_WIN_MAX; ++k) {
    const FwJob& c = cells[k < n ? k : n - 1];
    wt.x[k] = c.x, wt.v_out[k] = c.v_out, wt.z_out[k] = c.z_out, wt.zT_out[k] = c.zT_out, wt.flow[k] = c.pr.flow;
    wt.pt_out[k] = c.pt_out, wt.P_out[k] = c.P_out;
  }
  const bool plif = cells[0].leak_pt != nullptr, hard = cells[0].hard_reset != 0;
  FtPlan plan;
  plan.njobs = 1, plan.ntx = evf_cdiv(W, TW), plan.nyy = evf_cdiv(H, 2);
  const long nstrips = (long)plan.ntx * plan.nyy * B;
  if (nstrips >= (1L << 28)) return EVF_EINVAL;
  plan.nstrips = (int)nstrips;
  plan.nquads = evf_cdiv(nstrips, 4);
  for (int k = 0; k < FW_MAX_JOBS; ++k) plan.weight[k] = 1;
  plan.total = plan.nquads;
  static const int ilv_env = []() {
    const char* e = getenv("EVF_FWD_WIN_ILV");
    return e ? atoi(e) : 1;
  }();
  plan.ilv = ilv_env;
  const int nblk = plan.nquads < ncu ? plan.nquads : ncu;
  const bool full = (H % 2 == 0) && (W % TW == 0);
  hipStream_t st = EVF_STREAM(stream);
#define FT_GO(HARD_, FULL_)                                                                                                     \
  do {                                                                                                                          \
    if (plif && cells[0].xl == 2)                                                                                               \
      hipLaunchKernelGGL((k_fwd_win_t<true, FULL_, true, 2>), dim3(nblk), dim3(FT_THREADS), FT_LDS, st, job, wt, plan, B, H, W); \
    else if (plif && cells[0].xl)                                                                                               \
      hipLaunchKernelGGL((k_fwd_win_t<true, FULL_, true, 1>), dim3(nblk), dim3(FT_THREADS), FT_LDS, st, job, wt, plan, B, H, W); \
    else if (plif)                                                                                                              \
      hipLaunchKernelGGL((k_fwd_win_t<HARD_, FULL_, true>), dim3(nblk), dim3(FT_THREADS), FT_LDS, st, job, wt, plan, B, H, W);   \
    else                                                                                                                        \
      hipLaunchKernelGGL((k_fwd_win_t<HARD_, FULL_, false>), dim3(nblk), dim3(FT_THREADS), FT_LDS, st, job, wt, plan, B, H, W);  \
  } while (0)
  if (hard) {
    if (full) FT_GO(true, true); else FT_GO(true, false);
  } else {
    if (full) FT_GO(false, true); else FT_GO(false, false);
  }
#undef FT_GO
  return evf_status();
}
