// The tail of a training step in two launches (it was eight graph nodes: two row sums, the slab reduction, the segment add,
// a fill, the squared norm, clip + Adam, and each node costs ~5 us of graph time however little it does):
//   evf_grads_finalize   every partial sum of the window's backward -> the optimizer's flat gradient buffer:
//                        * the conv weights' slabs [nslab][9][32][32] (k_reduce_wgrad_multi's work),
//                        * the per-channel / head / prediction gradients: small accumulator + per-block rows + the head
//                          layer's per-block weight-gradient rows, added to their segments of the flat buffer
//                          (evf_sum_rows x 2 + evf_add_segments), sources handed back zeroed;
//   evf_clip_adam_fused  squared norm, a grid-wide hand-shake, clip + Adam (+ zero_grad) -- k_sumsq, k_clip_adam and the
//                        fill of the norm word in one launch of <= one block per CU.
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
  int seg_blk0[33];       // first block (of the segment part) of segment k; [nseg] = their number
};

#define GF_GROUPS 16
// blocks [0, 144 * nslabs): 64 outputs x 16 slab groups of tensor block / 144 (as k_reduce_wgrad_multi);
// blocks behind them: 64 columns x 16 row groups of one segment of the small gradients
__global__ __launch_bounds__(64 * GF_GROUPS) void k_grads_finalize(GfArgs a, int nslabs, int nslab, float* __restrict__ small,
                                                                   int clear_small, float* __restrict__ rows, int nrows, int ncols,
                                                                   const float* __restrict__ hrows, int nhrows, int nhcols, int hoff,
                                                                   int nseg) {
  __shared__ float red[GF_GROUPS][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int nb_slab = 144 * nslabs;
  if ((int)blockIdx.x < nb_slab) {
    const int t = blockIdx.x / 144, bx = blockIdx.x - 144 * t;
    const int e = bx * 64 + tx;  // e = (tau*32 + ci)*32 + co;  9216 = 144 * 64
    const float* __restrict__ p = a.slab[t] + e;
    constexpr long N = 9 * C32 * C32;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = ty;
    for (; k + 3 * GF_GROUPS < nslab; k += 4 * GF_GROUPS) {  // four independent loads in flight
      s0 += p[(long)k * N];
      s1 += p[(long)(k + GF_GROUPS) * N];
      s2 += p[(long)(k + 2 * GF_GROUPS) * N];
      s3 += p[(long)(k + 3 * GF_GROUPS) * N];
    }
    for (; k < nslab; k += GF_GROUPS) s0 += p[(long)k * N];
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < GF_GROUPS; ++g) s += red[g][tx];
      const int co = e & 31, ci = (e >> 5) & 31, tau = e >> 10;
      a.slab_dst[t][(co * C32 + ci) * 9 + tau] += s;
    }
    return;
  }
  const int sb = blockIdx.x - nb_slab;
  int k = 0;
  while (k + 1 < nseg && sb >= a.seg_blk0[k + 1]) ++k;  // (block-uniform, <= 32 steps)
  const int i = (sb - a.seg_blk0[k]) * 64 + tx;  // element of segment k
  const bool in = i < a.seg_n[k];
  const int e = a.seg_off[k] + (in ? i : 0);  // column of the small accumulator
  float v = 0.f;
  if (in) {
    if (rows && e < ncols)
      for (int r = ty; r < nrows; r += GF_GROUPS) {
        v += rows[(long)r * ncols + e];
        rows[(long)r * ncols + e] = 0.f;  // (the persistent per-block rows are handed back zeroed)
      }
    if (hrows && e >= hoff && e < hoff + nhcols)
      for (int r = ty; r < nhrows; r += GF_GROUPS) v += hrows[(long)r * nhcols + (e - hoff)];
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
                                  void* const* seg_dst, const int* seg_off, const int* seg_n, int nseg, void* stream) {
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
    if (k < nseg && (!a.seg_dst[k] || a.seg_off[k] < 0 || a.seg_n[k] <= 0)) return EVF_EINVAL;
    a.seg_blk0[k] = nb;
    if (k < nseg) nb += evf_cdiv(a.seg_n[k], 64);
  }
  a.seg_blk0[32] = nb;
  for (int k = nseg; k < 32; ++k) a.seg_blk0[k] = nb;
  hipLaunchKernelGGL(k_grads_finalize, dim3(144 * nslabs + nb), dim3(64 * GF_GROUPS), 0, EVF_STREAM(stream), a, nslabs, nslab, small,
                     clear_small, rows, nrows, ncols, head_rows, nhrows, nhcols, head_off, nseg);
  return evf_status();
}

// ---- clip_grad_norm_ + Adam in ONE launch ----------------------------------------------------------------------------------
// ws (>= 8 floats, zeroed once by the caller): [0] squared gradient norm of the last step (for the host to read), [1] the
// device-side step counter, [2] this step's running sum, [3] / [4] arrival / departure tickets (uint32).  Every block adds its
// part of the squared norm to ws[2] and takes an arrival ticket; when all gridDim.x tickets are out the sum is complete --
// the grid is at most one block per CU, i.e. all blocks are resident and the wait cannot deadlock.  The last block to LEAVE
// publishes ws[0] and clears [2..4]: the next launch needs no fill.
__global__ __launch_bounds__(256) void k_clip_adam_fused(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long n, float max_norm, float lr, float b1, float b2,
                                                         float host_step_size, float host_bc2_sqrt, float eps, float* ws,
                                                         int device_step, int zero_grad) {
  __shared__ float red[16];
  __shared__ float s_total;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = evf_block_sum(s, red);
  unsigned* tick = (unsigned*)(ws + 3);
  if (threadIdx.x == 0) {
    evf_atomic_add(ws + 2, s);
    if (device_step && blockIdx.x == 0) ws[1] += 1.0f;  // single writer; read by everybody behind the hand-shake
    __threadfence();
    __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(tick, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(2);
    s_total = __uint_as_float(__hip_atomic_load((unsigned*)(ws + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    red[0] = __uint_as_float(__hip_atomic_load((unsigned*)(ws + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  __syncthreads();
  const float total = s_total;
  // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(1.f, max_norm / (sqrtf(total) + 1e-6f));
  float step_size = host_step_size, bc2_sqrt = host_bc2_sqrt;
  if (device_step) {  // bias corrections from the device-side counter, in double like the host path (k_clip_adam)
    const double t = (double)red[0];
    step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    if (zero_grad) g[i] = 0.f;  // optimizer.zero_grad() of the next step, without its fill kernel
  }
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(tick + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {  // everybody has read the sum and the counter
      ws[0] = total;
      ws[2] = 0.f;
      tick[0] = 0u;
      tick[1] = 0u;
    }
  }
}

extern "C" int evf_clip_adam_fused(float* param, float* grad, float* m, float* v, int64_t n, float max_norm, float lr, float beta1,
                                   float beta2, float eps, int step, float* ws, int zero_grad, void* stream) {
  if (!param || !grad || !m || !v || !ws || n <= 0) return EVF_EINVAL;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 64;
  }
  const int device_step = step <= 0;  // step <= 0: use (and advance) the counter in ws[1]
  const long want = (n + 255) / 256;
  const int nblk = (int)(want < ncu ? want : ncu);  // all blocks resident: the hand-shake cannot deadlock
  double bc1 = 1.0, bc2 = 1.0;
  if (!device_step) {
    bc1 = 1.0 - pow((double)beta1, (double)step);
    bc2 = 1.0 - pow((double)beta2, (double)step);
  }
  hipLaunchKernelGGL(k_clip_adam_fused, dim3(nblk), dim3(256), 0, EVF_STREAM(stream), param, grad, m, v, (long)n, max_norm, lr,
                     beta1, beta2, (float)((double)lr / bc1), (float)sqrt(bc2), eps, ws, device_step, zero_grad);
  return evf_status();
}
