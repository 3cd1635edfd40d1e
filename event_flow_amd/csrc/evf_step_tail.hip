// The tail of a training step in two launches (it was eight graph nodes: two row sums, the slab reduction, the segment add,
// a fill, the squared norm, clip + Adam, and each node costs ~5 us of graph time however little it does):
//   evf_grads_finalize   every partial sum of the window's backward -> the optimizer's flat gradient buffer:
//                        * the conv weights' slabs [nslab][9][32][32] (k_reduce_wgrad_multi's work),
//                        * the per-channel / head / prediction gradients: small accumulator + per-block rows + the head
//                          layer's per-block weight-gradient rows, added to their segments of the flat buffer
//                          (evf_sum_rows x 2 + evf_add_segments), sources handed back zeroed;
//   evf_clip_adam_fused  squared norm (every block sums the whole gradient itself), clip + Adam (+ zero_grad behind a short
//                        hand-shake) -- k_sumsq, k_clip_adam and the fill of the norm word in one launch of 16 blocks.
// With several ranks the all-reduce of the flat buffer sits between the two (train.window_backward / window_apply).
// train_flow.py:157-164 (clip_grad_norm_, Adam.step, zero_grad).
#include "evf_common.h"

#define C32 32

struct GfArgs {
  const float* slab[16];  // [nslab][9*32*32] partial sums of a conv weight gradient ...
  float* slab_dst[16];    // ... added to this tensor (torch layout [co][ci][3][3])
  float* seg_dst[32];     // segment k: seg_dst[k][i] += total[seg_off[k] + i], i < seg_n[k]
  int seg_off[32];
  int seg_n[32];
  int seg_rows[32];       // rows of the per-block partials that can hold something for segment k (the others are zero: skipped)
  int seg_blk0[33];       // first block (of the segment part) of segment k; [nseg] = their number
};

#define GF_GROUPS 16
#define GF_SLAB_BLOCKS 36  // blocks per weight tensor of the slab part: 9216 outputs / (64 threads x 4)
#define GF_UNROLL 16  // independent row loads per thread and trip of the segment part (512 .. 1024 rows: 2 .. 4 trips)
// blocks [0, GF_SLAB_BLOCKS * nslabs): 256 outputs x 16 slab groups of tensor block / GF_SLAB_BLOCKS;
// blocks behind them: 64 columns x 16 row groups of one segment of the small gradients, all rows
__global__ __launch_bounds__(64 * GF_GROUPS) void k_grads_finalize(GfArgs a, int nslabs, int nslab, float* __restrict__ small,
                                                                   int clear_small, float* __restrict__ rows, int nrows, int ncols,
                                                                   const float* __restrict__ hrows, int nhrows, int nhcols, int hoff,
                                                                   int nseg) {
  __shared__ float4 red4[GF_GROUPS][64];
  float(*red)[64] = (float(*)[64])red4;  // (the segment part below: [GF_GROUPS][64] floats of the same array)
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int nb_slab = GF_SLAB_BLOCKS * nslabs;
  if ((int)blockIdx.x < nb_slab) {
    // A thread takes FOUR consecutive outputs (one float4 per slab row), a wave 1 KB of a row, the block's 16 waves 16 rows of the
    // same 1 KB span: with one float per thread a wave's load was 256 B of a row and the next one 36 KB away -- the launch ran at
    // 1.9 TB/s on its 80 MB.  Per output the same sums in the same order as before (rows ty, ty + 16, ... into four accumulators,
    // (s0 + s1) + (s2 + s3), then the 16 row groups in index order): the same bits.
    const int t = blockIdx.x / GF_SLAB_BLOCKS, bx = blockIdx.x - GF_SLAB_BLOCKS * t;
    const int e4 = bx * 64 + tx;  // float4 index; e = 4 e4 + c = (tau*32 + ci)*32 + co
    constexpr long N4 = 9 * C32 * C32 / 4;
    const float4* __restrict__ p = (const float4*)a.slab[t] + e4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s0 = z4, s1 = z4, s2 = z4, s3 = z4;
    auto add4 = [](float4& d, const float4& v) { d.x += v.x, d.y += v.y, d.z += v.z, d.w += v.w; };
    int k = ty;
    for (; k + 15 * GF_GROUPS < nslab; k += 16 * GF_GROUPS) {  // sixteen loads in flight (256 slab rows: one trip)
      float4 q[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) q[j] = p[(long)(k + j * GF_GROUPS) * N4];
#pragma unroll
      for (int j = 0; j < 16; j += 4) add4(s0, q[j]), add4(s1, q[j + 1]), add4(s2, q[j + 2]), add4(s3, q[j + 3]);
    }
    for (; k + 3 * GF_GROUPS < nslab; k += 4 * GF_GROUPS) {
      const float4 q0 = p[(long)k * N4], q1 = p[(long)(k + GF_GROUPS) * N4], q2 = p[(long)(k + 2 * GF_GROUPS) * N4],
                   q3 = p[(long)(k + 3 * GF_GROUPS) * N4];
      add4(s0, q0), add4(s1, q1), add4(s2, q2), add4(s3, q3);
    }
    for (; k < nslab; k += GF_GROUPS) add4(s0, p[(long)k * N4]);
    red4[ty][tx] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                               (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (ty < 4) {  // (wave c of the first four adds component c: 64 outputs each)
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < GF_GROUPS; ++g) {
        const float4 v = red4[g][tx];
        sum += ty == 0 ? v.x : (ty == 1 ? v.y : (ty == 2 ? v.z : v.w));
      }
      const int e = 4 * e4 + ty;
      const int co = e & 31, ci = (e >> 5) & 31, tau = e >> 10;
      a.slab_dst[t][(co * C32 + ci) * 9 + tau] += sum;
    }
    return;
  }
  // ---- small gradients: block = (segment k, 64 of its columns).  The block walks ALL rows of the per-block partials in a fixed
  // order (row group ty takes rows ty, ty + 16, ...; GF_UNROLL independent loads in flight per thread; the 16 row groups meet in
  // LDS and are added in index order) and is the only writer of its elements: no atomics, the same bits whatever the block
  // schedule, a zero or non-zero destination alike (the first fused version added one atomic per 64-row chunk).
  const int sb = blockIdx.x - nb_slab;
  int k = 0;
  while (k + 1 < nseg && sb >= a.seg_blk0[k + 1]) ++k;  // (block-uniform, <= 32 steps)
  const int cg = sb - a.seg_blk0[k];         // column group
  const int i = cg * 64 + tx;                // element of segment k
  const bool in = i < a.seg_n[k];
  const int e = a.seg_off[k] + (in ? i : 0);  // column of the small accumulator
  float v = 0.f;
  nrows = min(nrows, a.seg_rows[k]);
  if (rows && e < ncols) {  // (clamped rows + selects: every load of a trip is unconditional, i.e. in flight together)
    float* __restrict__ rp = rows + e;
    for (int r0 = 0; r0 < nrows; r0 += GF_GROUPS * GF_UNROLL) {
      float q[GF_UNROLL];
#pragma unroll
      for (int j = 0; j < GF_UNROLL; ++j) q[j] = rp[(long)min(r0 + ty + j * GF_GROUPS, nrows - 1) * ncols];
#pragma unroll
      for (int j = 0; j < GF_UNROLL; ++j) {
        const int r = r0 + ty + j * GF_GROUPS;
        if (r < nrows) {
          v += in ? q[j] : 0.f;
          if (in) rp[(long)r * ncols] = 0.f;  // (the persistent per-block rows are handed back zeroed)
        }
      }
    }
  }
  if (hrows && e >= hoff && e < hoff + nhcols) {
    const float* __restrict__ hp = hrows + (e - hoff);
    for (int r0 = 0; r0 < nhrows; r0 += GF_GROUPS * GF_UNROLL) {
      float q[GF_UNROLL];
#pragma unroll
      for (int j = 0; j < GF_UNROLL; ++j) q[j] = hp[(long)min(r0 + ty + j * GF_GROUPS, nhrows - 1) * nhcols];
#pragma unroll
      for (int j = 0; j < GF_UNROLL; ++j)
        if (in && r0 + ty + j * GF_GROUPS < nhrows) v += q[j];
    }
  }
  red[ty][tx] = v;
  __syncthreads();
  if (ty == 0 && in) {
    float s = small[e];
#pragma unroll
    for (int g = 0; g < GF_GROUPS; ++g) s += red[g][tx];
    a.seg_dst[k][i] += s;
    if (clear_small) small[e] = 0.f;
  }
}

extern "C" int evf_grads_finalize(const void* const* slabs, void* const* slab_dst, int nslabs, int nslab, float* small, int clear_small,
                                  float* rows, int nrows, int ncols, const float* head_rows, int nhrows, int nhcols, int head_off,
                                  void* const* seg_dst, const int* seg_off, const int* seg_n, const int* seg_rows, int nseg,
                                  void* stream) {
  if (nslabs < 0 || nslabs > 16 || nseg < 0 || nseg > 32 || (nslabs && (!slabs || !slab_dst || nslab <= 0)) ||
      (nseg && (!small || !seg_dst || !seg_off || !seg_n)) || (rows && (nrows <= 0 || ncols <= 0)) ||
      (head_rows && (nhrows <= 0 || nhcols <= 0 || head_off < 0)))
    return EVF_EINVAL;
  if (!nslabs && !nseg) return EVF_OK;
  GfArgs a;
  for (int i = 0; i < 16; ++i) {
    a.slab[i] = i < nslabs ? (const float*)slabs[i] : nullptr;
    a.slab_dst[i] = i < nslabs ? (float*)slab_dst[i] : nullptr;
    if (i < nslabs && (!a.slab[i] || !a.slab_dst[i])) return EVF_EINVAL;
  }
  int nb = 0;
  for (int k = 0; k < 32; ++k) {
    a.seg_dst[k] = k < nseg ? (float*)seg_dst[k] : nullptr;
    a.seg_off[k] = k < nseg ? seg_off[k] : 0;
    a.seg_n[k] = k < nseg ? seg_n[k] : 0;
    a.seg_rows[k] = (k < nseg && seg_rows && seg_rows[k] > 0) ? seg_rows[k] : 0x7fffffff;
    if (k < nseg && (!a.seg_dst[k] || a.seg_off[k] < 0 || a.seg_n[k] <= 0)) return EVF_EINVAL;
    a.seg_blk0[k] = nb;
    if (k < nseg) nb += evf_cdiv(a.seg_n[k], 64);
  }
  a.seg_blk0[32] = nb;
  for (int k = nseg; k < 32; ++k) a.seg_blk0[k] = nb;
  hipLaunchKernelGGL(k_grads_finalize, dim3(GF_SLAB_BLOCKS * nslabs + nb), dim3(64 * GF_GROUPS), 0, EVF_STREAM(stream), a, nslabs, nslab, small,
                     clear_small, rows, nrows, ncols, head_rows, nhrows, nhcols, head_off, nseg);
  return evf_status();
}

// ---- clip_grad_norm_ + Adam in ONE launch ----------------------------------------------------------------------------------
// For the parameter counts of the FireNets (75 k) the squared norm is cheap to compute REDUNDANTLY: CA_BLOCKS blocks of 1024
// threads each sum the whole gradient (300 KB out of the L2; the same order in every block, i.e. a reproducible norm -- no
// atomics, no word to clear) and then run clip + Adam on their own slice.  zero_grad is the one cross-block hazard (a block must
// not clear its slice while another still sums it).  No block ever waits for another one (the first version spun on
// an arrival ticket, which needs all blocks co-resident -- not guaranteed under a CU mask or beside other streams' blocks --
// and left stale tickets behind an aborted launch): nobody clears anything while working; every block takes a DEPARTURE ticket
// once it has read all it needs, and the block that draws the last ticket -- everybody else is provably done reading -- clears
// the whole gradient (75 floats per thread), publishes the norm, advances the step counter and resets the ticket.  Blocks
// that are scheduled late simply find the gradient still intact.  Large models keep the two-launch form
// (evf_clip_adam_step): n > CA_MAX_N.
// ws (>= 8 floats, zeroed once by the caller): [0] squared gradient norm of the last step, [1] the device-side step counter,
// [4] departure tickets (uint32, left zero; [3] unused, kept zero).
#define CA_BLOCKS 16
#define CA_MAX_N (1 << 20)
__global__ __launch_bounds__(1024) void k_clip_adam_fused(float* __restrict__ p, float* g, float* __restrict__ m,
                                                          float* __restrict__ v, long n, float max_norm, float lr, float b1, float b2,
                                                          float host_step_size, float host_bc2_sqrt, float eps, float* ws,
                                                          int device_step, int zero_grad) {
  __shared__ float red[16];
  __shared__ int s_last;
  // ---- the whole gradient's sum of squares, the same in every block (float4 trips + tail)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long n4 = n >> 2;
  const float4* g4 = (const float4*)g;
  for (long i = threadIdx.x; i < n4; i += blockDim.x) {
    const float4 q = g4[i];
    s0 += q.x * q.x, s1 += q.y * q.y, s2 += q.z * q.z, s3 += q.w * q.w;
  }
  for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s0 += g[i] * g[i];
  const float total = evf_block_sum_all((s0 + s1) + (s2 + s3), red);
  // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(1.f, max_norm / (sqrtf(total) + 1e-6f));
  float step_size = host_step_size, bc2_sqrt = host_bc2_sqrt;
  if (device_step) {  // bias corrections from the device-side counter (advanced by the last block to leave), in double
    const double t = (double)ws[1] + 1.0;
    step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  const long per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
  __syncthreads();  // (every thread of the block has read the gradient and ws[1])
  unsigned* tick = (unsigned*)(ws + 4);
  if (threadIdx.x == 0) {
    // (relaxed: what the last block needs is that every block has READ the gradient and the counter -- their values were consumed
    //  before the barrier above; an agent-scope release here would only write this XCD's dirty L2 back, once per block)
    const unsigned t = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t + 1u >= gridDim.x;  // (>=: a ticket left over by a launch that never finished cannot lock the tail out for good)
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last block to leave: everybody has read the gradient and the counter
  if (zero_grad) {  // optimizer.zero_grad() of the next step, without its fill kernel
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = threadIdx.x; i < n4; i += blockDim.x) ((float4*)g)[i] = z4;
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) g[i] = 0.f;
  }
  if (threadIdx.x == 0) {
    ws[0] = total;
    if (device_step) ws[1] += 1.0f;
    __hip_atomic_store(tick, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int evf_clip_adam_step_impl(float* param, float* grad, float* m, float* v, int64_t n, float max_norm, float lr, float beta1,
                            float beta2, float eps, int step, float* norm_ws, int zero_grad, void* stream);

extern "C" int evf_clip_adam_fused(float* param, float* grad, float* m, float* v, int64_t n, float max_norm, float lr, float beta1,
                                   float beta2, float eps, int step, float* ws, int zero_grad, void* stream) {
  if (!param || !grad || !m || !v || !ws || n <= 0) return EVF_EINVAL;
  if (n > CA_MAX_N || ((uintptr_t)grad & 15))  // large models: partial sums + a second launch (ws[0..1] mean the same there)
    return evf_clip_adam_step_impl(param, grad, m, v, n, max_norm, lr, beta1, beta2, eps, step, ws, zero_grad, stream);
  const int device_step = step <= 0;  // step <= 0: use (and advance) the counter in ws[1]
  double bc1 = 1.0, bc2 = 1.0;
  if (!device_step) {
    bc1 = 1.0 - pow((double)beta1, (double)step);
    bc2 = 1.0 - pow((double)beta2, (double)step);
  }
  const int nblk = (int)(n < 1024L * CA_BLOCKS ? (n + 1023) / 1024 : CA_BLOCKS);
  hipLaunchKernelGGL(k_clip_adam_fused, dim3(nblk), dim3(1024), 0, EVF_STREAM(stream), param, grad, m, v, (long)n, max_norm, lr,
                     beta1, beta2, (float)((double)lr / bc1), (float)sqrt(bc2), eps, ws, device_step, zero_grad);
  return evf_status();
}
