// Shared helpers for the gfx950 kernels of libevflow_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/evflow.h"

#define EVF_STREAM(s) ((hipStream_t)(s))

static inline int evf_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? EVF_OK : -(1000 + (int)e);
}
static inline int evf_hip(hipError_t e) { return e == hipSuccess ? EVF_OK : -(1000 + (int)e); }

// kernel sizes of the general convolution path: odd, padding k/2 (the reference's layers; models/unet.py:51 defaults to 5)
#define EVF_KSZ_OK(k) ((k) == 1 || (k) == 3 || (k) == 5 || (k) == 7)

static inline int evf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipMemsetAsync as a KERNEL.  Inside a captured step a hipMemsetAsync is a memset node, which the runtime executes outside the
// pre-built AQL packets of the kernel nodes around it; on ROCm 7.2 replays of such graphs returned corrupted results after a
// hipDeviceSynchronize (train.GraphedWindowStep: loss inf on the first replay behind the synchronize, fine with
// DEBUG_CLR_GRAPH_PACKET_CAPTURE=0).  A fill kernel is a kernel node like its neighbours.  `bytes` any size, `dst` 4-byte aligned
// when bytes >= 4 (every caller clears float / uint32 tensors).
static __global__ void k_evf_fill(uint32_t* __restrict__ dst, uint32_t word, size_t nwords, unsigned char* __restrict__ tail, int ntail,
                                  unsigned char byte) {
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4, stride = (size_t)gridDim.x * blockDim.x * 4;
  const bool al16 = (((uintptr_t)dst) & 15) == 0;
  for (size_t i = i0; i < nwords; i += stride) {
    if (al16 && i + 4 <= nwords) {
      *(uint4*)(dst + i) = make_uint4(word, word, word, word);
    } else {
      for (size_t k = i; k < nwords && k < i + 4; ++k) dst[k] = word;
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = byte;
}
static inline hipError_t evf_memset_async(void* dst, int value, size_t bytes, hipStream_t st) {
  if (!bytes) return hipSuccess;
#ifdef EVF_MEMSET_RUNTIME  // (A/B builds)
  return hipMemsetAsync(dst, value, bytes, st);
#endif
  if (((uintptr_t)dst) & 3) return hipMemsetAsync(dst, value, bytes, st);  // (never taken by this library's callers)
  const unsigned char b = (unsigned char)value;
  const uint32_t w = 0x01010101u * b;
  const size_t nwords = bytes / 4;
  const int ntail = (int)(bytes & 3);
  const size_t quads = (nwords + 3) / 4;
  long nb = (long)((quads + 255) / 256);
  if (nb < 1) nb = 1;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_evf_fill, dim3((unsigned)nb), dim3(256), 0, st, (uint32_t*)dst, w, nwords, (unsigned char*)dst + nwords * 4, ntail, b);
  return hipGetLastError();
}

// PLIF / XLIF presynaptic trace update (reference models/spiking_submodules.py:212, :418, :642):
//   pt' = pt * sigma(leak_pt) + (1 - sigma(leak_pt)) * P
// ONE expression for every forward kernel that produces pt' (fwd_b3_body, k_fwd_diag_t, k_head_lif_fwd[_win], k_neuron_fwd) and for
// every backward that recomputes it from pt and P instead of reading it back (k_plif_trace_bwd, team E of k_bwd_diag_ws_plif,
// head_bwd_pass<.., PLIF>): the library is built with -ffp-contract=off, so this is mul, sub, mul, add wherever it is inlined --
// the recomputed trace equals the stored one bit for bit.  evf_plif_sigmoid: the one sigmoid of the trace parameters.
__device__ __forceinline__ float evf_plif_trace(float pt, float lpt, float P) { return pt * lpt + (1.0f - lpt) * P; }
__device__ __forceinline__ float evf_plif_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// hardware fp32 atomic add (global_atomic_add_f32 / ds_add_f32); plain
// atomicAdd would lower to a CAS loop without -munsafe-fp-atomics.
__device__ __forceinline__ void evf_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }

// Non-temporal 16-byte store (global_store_dwordx4 ... nt) for tensors that are streamed out once: the lines do not stay
// dirty in the XCD's L2, so the kernel does not end in a multi-microsecond L2 write-back.  Measured on MI355X (tools/probes/
// stream_probe.hip, 4 tensors in / 2 out of 16.8 MB each): 16.3 us with nt stores against 22.6 us with plain ones (sc1 and
// sc0 sc1 write-through: 21.7); a dependent pair of such kernels 33 against 37 us.  Non-temporal LOADS gained nothing.
// NOT for partial-line writers: the MFMA epilogues of the forward / input-gradient kernels put 32 bytes of a pixel's 128-byte
// line per instruction, and as nt stores those reach memory uncombined (dgrad 22.9 -> 27.2 us, forward 16.5 -> 20.9 us in
// the step); the fused backward (full lines) neither gained nor lost.  Kept for kernels that end in a large full-line output.
typedef float evf_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void evf_store_nt(float4* p, const float4 v) {
  const evf_v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, (evf_v4f*)p);
}
__device__ __forceinline__ void evf_store_nt(float* p, const float4 v) { evf_store_nt((float4*)p, v); }

// wave64 sum, result valid in every lane
__device__ __forceinline__ float evf_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of up to 1024 threads; result valid in thread 0
__device__ __forceinline__ float evf_block_sum(float v, float* smem /* >= 16 floats */) {
  v = evf_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < 64) {
    const int nw = (blockDim.x + 63) >> 6;
    r = (lane < nw) ? smem[lane] : 0.f;
    r = evf_wave_sum(r);
  }
  __syncthreads();
  return r;
}

// block-wide sum of up to 1024 threads, the same bits in EVERY thread (fixed order: wave sums, then the waves in index order)
__device__ __forceinline__ float evf_block_sum_all(float v, float* smem /* >= 16 floats */) {
  v = evf_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  float r = 0.f;
  for (int w = 0; w < nw; ++w) r += smem[w];
  __syncthreads();
  return r;
}

// evf_conv_b3tile.hip: spatially tiled 3x3 stride-1 convolution behind evf_conv2d_fwd_b3 / evf_conv2d_dgrad_b3
int evf_conv3_b3t_plan(const float* src, int B, int H, int W, int K, int N, int lds, bool force, int max_split, int force_split);
int evf_conv3_b3t_launch(const float* src, int lds, const void* wp, const float* bias, float* out, int ldo, int B, int H, int W,
                         int K, int N, int flip, int accumulate, int ksplit, hipStream_t st);

// evf_conv_b3n.hip: 3x3 stride-1 convolution with the halo tile staged once per channel group and up to 192 output channels
// streaming past it (layers with many output channels per input tile: the decoders' input gradients); same arguments as the tile kernel's
int evf_conv3_b3n_plan(const float* src, int B, int H, int W, int K, int N, int lds, bool force, int max_split, int force_split);
int evf_conv3_b3n_launch(const float* src, int lds, const void* wp, const float* bias, float* out, int ldo, int B, int H, int W,
                         int K, int N, int flip, int accumulate, int ksplit, hipStream_t st);

// evf_conv_b3img.hip: 3x3 stride-1 convolution of images of at most 16 x 16 pixels with many channels (one block per image x 64
// output channels x K split); same operands and meaning of the arguments as the tile kernel's pair above
int evf_conv3_b3i_plan(const float* src, int B, int H, int W, int K, int N, int lds, bool force, int max_split, int force_split);
int evf_conv3_b3i_launch(const float* src, int lds, const void* wp, const float* bias, float* out, int ldo, int B, int H, int W,
                         int K, int N, int flip, int accumulate, int ksplit, hipStream_t st);

// evf_conv_b3small.hip: 3x3 stride-1 forward product of an input exactly representable in bf16 from channel `exact_from` on (the
// caller's promise; NaN if broken)
int evf_conv3_b3x_plan(const float* src, int B, int H, int W, int K, int N, int lds, int exact_from, long slab_cap);
int evf_conv3_b3x_launch(const float* src, int lds, const void* wp, float* dst, int ldo, int B, int H, int W, int K, int N,
                         int exact_from, long slab_cap, hipStream_t st);

// evf_wgrad_b3gen.hip: bf16 matrix-core weight gradient of the general 3x3 convolution (stride 1 or 2) behind evf_conv2d_wgrad
bool evf_wgrad9_b3_ok(const float* x, const float* gy, int Cin, int Cout, int ldx, int ldg);
int evf_wgrad9_b3_launch(const float* x, int ldx, const float* gy, int ldg, float* slab, float* gbias, int* redo, int B, int H,
                         int W, int Cin, int Cout, int nsplit, int CT, int NT, hipStream_t st, float* fuse_gw = nullptr,
                         int* tickets = nullptr, int cin_total = 0, int cin_off = 0, int accumulate = 0, int promised = 0,
                         int stride = 1);

// evf_dgrad_ws.hip: wave-specialised input-gradient kernel behind evf_conv_dgrad_b3_f32[_pair]
// PLIF: d loss / d(input spike) through the pooled pre-synaptic trace at pixel (y, x) of sample b.  raw = 0: g_P is the boxed,
// scaled map (evf_plif_trace_bwd's g_P_in); raw = 1: g_P is its g_P_raw and AvgPool3x3^T / 32 is applied here -- the same sums
// in the same order as k_plif_box (a term outside the image adds 0.f).  Nine unconditional loads either way (no load under a
// branch; raw = 0 reads the centre nine times, an L1 hit).
// ... in two halves, for a caller that requests the nine values long before it needs their sum
__device__ __forceinline__ void evf_plif_gp_load(const float* __restrict__ gP, int raw, int b, int y, int x, int H, int W, float (&v)[9]) {
  const int yc = y < H - 1 ? y : H - 1, xc = x < W - 1 ? x : W - 1;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = yc + dy - 1, xx = xc + dx - 1;
      const int ya = raw ? (yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy)) : yc, xa = raw ? (xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx)) : xc;
      v[3 * dy + dx] = gP[((long)b * H + ya) * W + xa];
    }
}
__device__ __forceinline__ float evf_plif_gp_sum(int raw, int y, int x, int H, int W, const float (&v)[9]) {
  const int yc = y < H - 1 ? y : H - 1, xc = x < W - 1 ? x : W - 1;
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = yc + dy - 1, xx = xc + dx - 1;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      s += (raw ? in : (dy == 1 && dx == 1)) ? v[3 * dy + dx] : 0.f;
    }
  return raw ? (s / 9.0f) / 32.0f : s;
}
__device__ __forceinline__ float evf_plif_gp(const float* __restrict__ gP, int raw, int b, int y, int x, int H, int W) {
  const int yc = y < H - 1 ? y : H - 1, xc = x < W - 1 ? x : W - 1;
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = yc + dy - 1, xx = xc + dx - 1;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const int ya = raw ? (yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy)) : yc, xa = raw ? (xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx)) : xc;
      const float v = gP[((long)b * H + ya) * W + xa];
      s += (raw ? in : (dy == 1 && dx == 1)) ? v : 0.f;
    }
  return raw ? (s / 9.0f) / 32.0f : s;
}

int evf_dgrad_ws_launch(const float* g_cur, const void* wT_b3, float* g_x, int accumulate, int B, int H, int W, const float* g_P,
                        const uint32_t* x_bits, const void* wT2_b3, float* g_x2, int max_blocks, void* stream);

// evf_dgrad_diag.hip: the input-gradient cells of one backward index as a flat list of products (gradient, weight set,
// output) in one persistent wave-specialised launch (k_dgrad_diag_ws)
#define EVF_DG_MAX_PROD 16
struct EvfDgProd {
  const void* g;   // fp32 gradient [B,H,W,32] (k_dgrad_diag_ws) / its three bf16 planes [term][B,H,W,32] (k_dgrad_diag_dma)
  const void* wt;  // split transposed weights (evf_pack_conv_weight_b3t)
  float* gx;       // [B,H,W,32], written
  // PLIF (k_dgrad_diag_dma only; NULL: none): the raw map dL/d(pooled activity) [B,H,W] and the layer's input spike words [B,H,W] --
  // the product's output gets AvgPool3x3^T(gP) / 32 where the input spike is set (evf_plif_gp, `accumulate | 2` of evf_conv_dgrad_b3)
  const float* gP;
  const uint32_t* xb;
};
struct EvfDgProds {
  EvfDgProd p[EVF_DG_MAX_PROD];
};
int evf_dgrad_diag_ws_launch(const EvfDgProds& P, int nprod, int B, int H, int W, void* stream);
// ... and with the gradient pre-split in HBM (three bf16 planes [term][B,H,W,32]): LDS-DMA fed, one wave per SIMD (k_dgrad_diag_dma)
int evf_dgrad_diag_dma_launch(const EvfDgProds& P, int nprod, int B, int H, int W, void* stream);
// whether a cell of this shape may be RECORDED for the persistent launches (their index arithmetic: < 2^22 items with a full
// table of products, 32-bit plane offsets); a cell that does not fit launches directly instead of failing at the flush
bool evf_dgrad_diag_fits(int split, int B, int H, int W);

// Deferred backward cells (evf_bwd_defer_*, owner: evf_bwd_fused.hip): while `active`, the fused backward, the fp32 input
// gradient and the head backward of the default-neuron FireNet path RECORD their launch under index `slot`; the flush
// launches index after index -- the fused-backward cells of an index as one k_bwd_diag, the input-gradient cells as one
// k_dgrad_diag, head cells one by one.  A launcher that cannot record its call first flushes everything recorded so far
// (record order is a valid execution order) and then launches as usual.
// RECORDING CONTEXTS.  A recording (forward or backward) belongs to the HIP STREAM it was opened on: every recorder below is
// an array indexed by a context id, and a context is looked up by the `stream` argument every entry point has.  Two host
// threads driving two models on two streams therefore record and launch independently -- also through PyTorch's autograd,
// whose backward nodes run on ONE worker thread per device but on the stream of their forward.  (A process-global recorder
// mixed such cells; a thread_local one did as well, because of that shared worker thread.)  Two recordings of one kind on
// ONE stream are refused (EVF_EINVAL).  evf_ctx_* are implemented in evf_bwd_fused.hip.
#define EVF_CTX_MAX 16
int evf_ctx_find(void* stream);     // context of `stream`, or -1: no recording is open on it
int evf_ctx_acquire(void* stream);  // find or create, one more open recording on it; -1: all EVF_CTX_MAX contexts are in use
void evf_ctx_drop(int ctx);         // one recording of the context ended (at zero the context is free again)
#define EVF_BWD_DIAGS 96
struct EvfBwdDefer {
  bool active;
  int slot;
  bool hold_heads;  // evf_bwd_defer_hold_heads: the head layer's recorded cells wait for the recording's END (evf_bwd_defer_flush)
};
extern EvfBwdDefer evf_bwd_defer_tab[EVF_CTX_MAX];
int evf_bwd_defer_flush_now(int ctx, void* stream, bool final = false);  // launch what is recorded, keep recording
int evf_dg_defer_launch(int ctx, int d, void* stream);  // evf_dgrad_b3.hip: launch and clear the cells of index d
int evf_dg_defer_count(int ctx);
int evf_dg_defer_pending(int ctx, int d);  // cells recorded under index d
int evf_hd_defer_launch(int ctx, int d, void* stream);  // evf_network.hip (head layer)
int evf_hd_defer_count(int ctx);
int evf_hd_defer_pending(int ctx, int d);  // cells recorded under index d
int evf_hd_defer_window_ok(int ctx);  // the recorded head cells may run after the last index (no shared dL/d(spikes) buffer)
int evf_hd_defer_launch_window(int ctx, void* stream);  // ... all of them, consecutive passes in one launch
// head cells of a FORWARD recording (evf_network.hip): launched before the diagonals, consecutive passes in one launch
int evf_fwd_defer_active(int ctx);  // evf_fwd_b3.hip: a forward recording is open in this context
int evf_defer_poisoned();           // evf_defer_poison is on
int evf_hf_defer_launch(int ctx, void* stream);
int evf_hf_defer_count(int ctx);
void evf_hf_defer_reset(int ctx);
// Per-launch timing of the diagonal launches (evf_defer_profile, evf_bwd_fused.hip): HIP events around every dispatcher
// launch of a flush, by kind (0 k_fwd_diag, 1 k_bwd_diag, 2 k_dgrad_diag, 3 head backward).  No-ops unless switched on
// (never during a graph capture).
void evf_prof_mark(int kind, int end, void* stream);
int evf_prof_mode();  // 0 off, 1 eager brackets, 2 brackets captured into a hipGraph
