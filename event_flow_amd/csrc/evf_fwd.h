// Shared definitions of the fused forward kernels of the 32 -> 32 spiking conv cells (evf_fwd_b3.hip: one tile per block,
// k_fwd_diag, k_fwd_diag_p; evf_fwd_teams.hip: k_fwd_diag_t).
#pragma once
#include "evf_common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define C32 32
#define TW 32
#define HALO_W (TW + 2)
#define NFRAG 54                 // 9 taps x 2 k-halves x 3 terms
#define FW_SP 36                 // floats per pixel of the epilogue staging tile (32 + 4: conflict-free 16-byte writes)
#define WB3_BYTES (NFRAG * 1024)  // per conv

__device__ __forceinline__ float b3_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// LDS-DMA of one 1 KiB piece (64 lanes x 16 B, destination = wave-uniform LDS address + lane*16), invisible to the
// compiler's wait counting: completion is covered by the explicit s_waitcnt vmcnt(0) before the barrier that
// publishes the buffer.  Staging the 54 KiB of split weights through registers (load -> wait -> ds_write, 13.5 uint4
// per thread) cost 6.8 us of a 15.7 us block lifetime (s_memtime instrumentation); as DMA all pieces are in flight
// at once and no VGPRs are held.
__device__ __forceinline__ void b3_glds16(const void* gsrc, void* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// Prediction head (models/model.py:197-199, :265) fused into the epilogue of the layer it reads: flow = tanh(W z + b)
// from the spike word of each pixel, same summation order as evf_pred_fwd.  All NULL: no head here.
struct PredArgs {
  const float* w;     // [2][32]
  const float* bias;  // [2]
  float* flow;        // [B,2,H,W]
};

// ---- several independent (pass, layer) cells of a window in ONE launch (evf_fwd_defer_*) ----------------------------------
#define FW_MAX_JOBS 8
struct FwJob {
  const uint32_t* x;
  const uint4* wff;
  const uint4* wrec;  // NULL: feed-forward cell
  const float* leak;
  const float* thresh;
  const float* v_prev;
  const uint32_t* z_prev;
  float* v_out;
  uint32_t* z_out;
  uint32_t* zT_out;
  PredArgs pr;
  int hard_reset;
  int xl;  // 1: the PLIF fields describe an XLIF cell (spiking_submodules.py:337-435, :771-875): `add_pt` holds t1, `thresh` t0 -- the
           // trace raises the THRESHOLD (t0 + t1 * pt') instead of being subtracted from the current.  2: an ALIF cell (:230-334,
           // :660-768): the same with the trace driven by the cell's OWN previous spikes instead of the pooled input activity
  // PLIF cell (spiking_submodules.py:191-227, :618-657; leak_pt == NULL: LIF): per-channel trace parameters, previous trace
  // [B,H,W,32] or NULL, new trace, pooled pre-synaptic activity [B,H,W] (saved for the backward)
  const float* leak_pt;
  const float* add_pt;
  const float* pt_prev;
  float* pt_out;
  float* P_out;
};
struct FwJobs {
  FwJob j[FW_MAX_JOBS];
};

// evf_fwd_teams.hip: the n (<= FW_MAX_JOBS) default-neuron cells of one index as ONE persistent launch of matrix + element-wise
// wave teams; all cells of one reset rule.
int evf_fwd_diag_t_launch(const FwJobs& jobs, int n, int B, int H, int W, void* stream);

// ---- the passes of a window of ONE feed-forward layer in one launch (k_fwd_win_t, evf_fwd_teams.hip) ----------------------------
// Cells recorded under one index that form a CHAIN (cell k + 1 starts from cell k's state, same weights, no recurrent conv) are a
// window: the first cell carries weights, parameters and the state before the window, the table the per-pass operands.
#define FW_WIN_MAX 16
struct FwWinTab {
  int np, pad_;
  const uint32_t* x[FW_WIN_MAX];
  float* v_out[FW_WIN_MAX];
  uint32_t* z_out[FW_WIN_MAX];
  uint32_t* zT_out[FW_WIN_MAX];
  float* flow[FW_WIN_MAX];    // prediction head in the epilogue (cell 0's pr.w != NULL), else unused
  float* pt_out[FW_WIN_MAX];  // PLIF
  float* P_out[FW_WIN_MAX];   // PLIF
};
struct FwWinNone {};
struct FwJob1 {
  FwJob j[1];
};
// 1: the n cells are such a chain (n >= 2)
int evf_fwd_win_is_chain(const FwJob* cells, int n);
int evf_fwd_win_t_launch(const FwJob* cells, int n, int B, int H, int W, void* stream);
