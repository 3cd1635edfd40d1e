// Event-list kernels for gfx950: window encodings, IWE warp + splat, the
// contrast-maximisation loss (forward + backward) and the metric reductions.
//
// All of these are HBM / L2-atomic bound scan+scatter kernels: one thread per
// event, 16-byte coalesced event reads, two 4-byte flow gathers, fp32 hardware
// atomics into per-sample images that stay L2 resident (a 128x128x8-channel
// sample is 512 KiB).  Built with -ffp-contract=off: the warp
// `y + ((tref - t) * f) * S` must round exactly like the reference's separate
// torch ops (utils/iwe.py:37) for the rounded-index IWE to be bit-exact.
#include <stdlib.h>

#include "evf_common.h"
#include <mutex>

// --------------------------------------------------------------------------
// encodings
// --------------------------------------------------------------------------
__global__ void k_events_to_image(const float* __restrict__ xs, const float* __restrict__ ys,
                                  const float* __restrict__ vals, int n, int H, int W, int accumulate,
                                  float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // .long() truncates toward zero (dataloader/encodings.py:39-42)
  long x = (long)xs[i], y = (long)ys[i];
  if (x < 0 || x >= W || y < 0 || y >= H) return;  // the reference would raise IndexError
  float v = vals[i];
  if (accumulate)
    evf_atomic_add(out + y * W + x, v);
  else
    out[y * W + x] = v;
}

extern "C" int evf_events_to_image(const float* xs, const float* ys, const float* vals, int n, int H, int W,
                                   int accumulate, float* out, void* stream) {
  if (!out || H <= 0 || W <= 0 || n < 0 || (n > 0 && (!xs || !ys || !vals))) return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  int rc = evf_hip(evf_memset_async(out, 0, sizeof(float) * (size_t)H * W, st));
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_events_to_image, dim3(evf_cdiv(n, 256)), dim3(256), 0, st, xs, ys, vals, n, H, W, accumulate, out);
  return evf_status();
}

__global__ void k_encode_events(const float4* __restrict__ ev, int B, int N, int H, int W, int nb, int round_ts,
                                float* __restrict__ cnt, float* __restrict__ mask, float* __restrict__ voxel,
                                float2* __restrict__ pol) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * N) return;
  const int b = (int)(i / N);
  const float4 e = ev[i];  // (t, y, x, p)
  const float p = e.w;
  if (pol) pol[i] = make_float2(p > 0.f ? p : 0.f, p < 0.f ? -p : 0.f);  // base.py:210-222
  if (p == 0.f) return;  // padding
  const long x = (long)e.z, y = (long)e.y;
  if (x < 0 || x >= W || y < 0 || y >= H) return;
  const long HW = (long)H * W, px = y * W + x;
  if (cnt) evf_atomic_add(cnt + ((long)b * 2 + (p > 0.f ? 0 : 1)) * HW + px, p * p);  // encodings.py:77-83
  if (mask) mask[(long)b * HW + px] = fabsf(p);  // non-accumulating put, base.py:168-170
  if (voxel) {
    float t = e.x * (float)(nb - 1);  // encodings.py:56-59
    if (round_ts) t = rintf(t);
    for (int k = 0; k < nb; ++k) {
      float w = fmaxf(0.f, 1.0f - fabsf(t - (float)k));  // :63
      if (w != 0.f) evf_atomic_add(voxel + ((long)b * nb + k) * HW + px, p * w);
    }
  }
}

extern "C" int evf_encode_events(const float* ev, int B, int N, int H, int W, int num_bins, int round_ts, float* cnt,
                                 float* mask, float* voxel, float* pol, void* stream) {
  if (!ev || B <= 0 || N < 0 || H <= 0 || W <= 0 || (voxel && num_bins < 1)) return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  const size_t HW = (size_t)H * W;
  int rc = 0;
  if (cnt) rc |= evf_hip(evf_memset_async(cnt, 0, sizeof(float) * B * 2 * HW, st));
  if (mask) rc |= evf_hip(evf_memset_async(mask, 0, sizeof(float) * B * HW, st));
  if (voxel) rc |= evf_hip(evf_memset_async(voxel, 0, sizeof(float) * B * num_bins * HW, st));
  if (rc || N == 0) return rc;
  hipLaunchKernelGGL(k_encode_events, dim3(evf_cdiv((long)B * N, 256)), dim3(256), 0, st, (const float4*)ev, B, N, H, W,
                     num_bins, round_ts, cnt, mask, voxel, (float2*)pol);
  return evf_status();
}

// All passes of a BPTT window in one launch.  ev [B][P][N] rows (t, y, x, p) -- batch-major, so that the window's event
// list [B][P*N] the loss reads is the same memory.  The network inputs come out pass-major (cnt [P][B][2][H][W], voxel
// [P][B][nb][H][W]: one contiguous tensor per pass), the loss inputs batch-major (mask [B][P][H][W], pol [B][P*N][2]):
// no torch.stack / torch.cat between the binning and its consumers.
__global__ void k_encode_window(const float4* __restrict__ ev, int B, int P, int N, int H, int W, int nb, int round_ts,
                                float* __restrict__ cnt, float* __restrict__ mask, float* __restrict__ voxel,
                                float2* __restrict__ pol) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * P * N) return;
  const long bp = i / N;
  const int b = (int)(bp / P), pp = (int)(bp - (long)b * P);
  const long s = (long)pp * B + b;  // pass-major sample index of the network inputs
  const float4 e = ev[i];
  const float p = e.w;
  if (pol) pol[i] = make_float2(p > 0.f ? p : 0.f, p < 0.f ? -p : 0.f);
  if (p == 0.f) return;
  const long x = (long)e.z, y = (long)e.y;
  if (x < 0 || x >= W || y < 0 || y >= H) return;
  const long HW = (long)H * W, px = y * W + x;
  if (cnt) evf_atomic_add(cnt + (s * 2 + (p > 0.f ? 0 : 1)) * HW + px, p * p);
  if (mask) mask[bp * HW + px] = fabsf(p);
  if (voxel) {
    float t = e.x * (float)(nb - 1);
    if (round_ts) t = rintf(t);
    for (int k = 0; k < nb; ++k) {
      const float w = fmaxf(0.f, 1.0f - fabsf(t - (float)k));
      if (w != 0.f) evf_atomic_add(voxel + (s * nb + k) * HW + px, p * w);
    }
  }
}

// dense: ONE allocation [cnt | voxel | mask] (the parts `want` selects: 1 cnt, 2 voxel, 4 mask), zero-filled here in one go
extern "C" int evf_encode_window(const float* ev, int B, int P, int N, int H, int W, int num_bins, int round_ts, int want,
                                 float* dense, float* pol, void* stream) {
  if ((N > 0 && !ev) || B <= 0 || P <= 0 || N < 0 || H <= 0 || W <= 0 || ((want & 2) && num_bins < 1) || ((want & 7) && !dense))
    return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  const size_t HW = (size_t)H * W, S = (size_t)B * P;
  float* cnt = (want & 1) ? dense : nullptr;
  float* voxel = (want & 2) ? dense + ((want & 1) ? S * 2 * HW : 0) : nullptr;
  float* mask = (want & 4) ? dense + ((want & 1) ? S * 2 * HW : 0) + ((want & 2) ? S * num_bins * HW : 0) : nullptr;
  const size_t total = ((want & 1) ? S * 2 * HW : 0) + ((want & 2) ? S * num_bins * HW : 0) + ((want & 4) ? S * HW : 0);
  if (total) {
    const int rc = evf_hip(evf_memset_async(dense, 0, sizeof(float) * total, st));
    if (rc) return rc;
  }
  if (N == 0) return EVF_OK;
  hipLaunchKernelGGL(k_encode_window, dim3(evf_cdiv((long)B * P * N, 256)), dim3(256), 0, st, (const float4*)ev, B, P, N, H, W,
                     num_bins, round_ts, cnt, mask, voxel, (float2*)pol);
  return evf_status();
}

// --------------------------------------------------------------------------
// warp + splat of one event into up to four image channels
//   I0 += w*a0, I1 += w*a1, T0 += (w*tau)*a0, T1 += (w*tau)*a1
// ROUND: torch.round indices, weight 1 (utils/iwe.py:39-43); else the four
// bilinear corners (utils/iwe.py:48-62).  Out-of-image corners carry weight 0
// (purge_unfeasible, :4-17) and are simply skipped.
// --------------------------------------------------------------------------
struct Warp {
  float wy, wx;
};

__device__ __forceinline__ Warp evf_warp(float t, float y, float x, float fy, float fx, float tref, float S) {
  // events[:, :, 1:3] + (tref - t) * flow * flow_scaling   (left-to-right)
  const float dt = tref - t;
  Warp w;
  w.wy = y + (dt * fy) * S;
  w.wx = x + (dt * fx) * S;
  return w;
}

__device__ __forceinline__ void evf_put(float* __restrict__ img, long px, float v) {
  if (img && v != 0.f) evf_atomic_add(img + px, v);
}

template <bool ROUND>
__device__ __forceinline__ void evf_splat(const Warp w, int H, int W, float a0, float a1, float tau, float* I0,
                                          float* I1, float* T0, float* T1) {
  if (ROUND) {
    const float iy = rintf(w.wy), ix = rintf(w.wx);  // half-to-even like torch.round
    if (iy < 0.f || iy >= (float)H || ix < 0.f || ix >= (float)W) return;
    const long px = (long)(iy * (float)W + ix);
    evf_put(I0, px, a0);
    evf_put(I1, px, a1);
    evf_put(T0, px, tau * a0);
    evf_put(T1, px, tau * a1);
  } else {
    const float cy[2] = {floorf(w.wy), floorf(w.wy + 1.0f)};
    const float cx[2] = {floorf(w.wx), floorf(w.wx + 1.0f)};
    float ay[2], ax[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      ay[k] = fmaxf(0.f, 1.0f - fabsf(w.wy - cy[k]));
      ax[k] = fmaxf(0.f, 1.0f - fabsf(w.wx - cx[k]));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (cy[j] < 0.f || cy[j] >= (float)H || cx[i] < 0.f || cx[i] >= (float)W) continue;
        const float wt = ay[j] * ax[i];
        if (wt == 0.f) continue;
        const long px = (long)(cy[j] * (float)W + cx[i]);
        evf_put(I0, px, wt * a0);
        evf_put(I1, px, wt * a1);
        const float wtau = wt * tau;  // (fw_weights * ts_list) * polarity_mask, loss/flow.py:206-211
        evf_put(T0, px, wtau * a0);
        evf_put(T1, px, wtau * a1);
      }
  }
}

__device__ __forceinline__ void evf_event_flow(const float* __restrict__ flow, int map, int B, int b, long HW, float y,
                                               float x, int W, float& fy, float& fx) {
  // flow_idx = y*W + x in float, then .long()  (loss/flow.py:65-67)
  const long lin = (long)(y * (float)W + x);
  const float* f = flow + ((long)map * B + b) * 2 * HW;
  fx = f[lin];       // horizontal component, channel 0 (:76)
  fy = f[HW + lin];  // vertical component, channel 1 (:75)
}

template <bool ROUND>
__global__ void k_iwe_splat(const float* __restrict__ flow, const float4* __restrict__ ev,
                            const int32_t* __restrict__ map_of_event, const int32_t* __restrict__ ts_shift,
                            const float* __restrict__ w0, const float* __restrict__ w1, int wstride, int B, int M, int H,
                            int W, float S, float tref, float tref_ts, int mode, int nch, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * M) return;
  const int b = (int)(i / M), e = (int)(i - (long)b * M);
  const float4 q = ev[i];
  float t = q.x;
  if (ts_shift) t += (float)ts_shift[e];
  const long HW = (long)H * W;
  float fy = 0.f, fx = 0.f;
  evf_event_flow(flow, map_of_event ? map_of_event[e] : 0, B, b, HW, q.y, q.z, W, fy, fx);
  if (mode & 2) {  // `flow * 0` keeps the sign/NaN semantics of the reference
    fy *= 0.f;
    fx *= 0.f;
  }
  const float a0 = w0 ? w0[i * wstride] : 1.0f;
  const float a1 = w1 ? w1[i * wstride] : 0.0f;
  const float tau = (mode & 8) ? (tref_ts - t) : t;
  float* o = out + (long)b * nch * HW;
  float *I0 = o, *I1 = nullptr, *T0 = nullptr, *T1 = nullptr;
  if (nch >= 2) I1 = o + HW;
  if (nch == 4) {
    T0 = o + 2 * HW;
    T1 = o + 3 * HW;
  }
  evf_splat<ROUND>(evf_warp(t, q.y, q.z, fy, fx, tref, S), H, W, a0, a1, tau, I0, I1, T0, T1);
}

// LDS-privatised variant: one block owns a stripe of `rows` image rows of ONE sample (all
// nch channels) in LDS, scans every event of that sample, keeps the taps that land in its
// stripe with LDS atomics and finally writes the stripe with coalesced float4 stores.
// Device-scope fp32 atomics on global memory top out near 21 G atomics/s on MI355X (they
// are resolved at the memory side of the 8 non-coherent L2s); LDS atomics scale with the
// CUs.  Same arithmetic as k_iwe_splat (integer histograms stay bit-exact).
template <bool ROUND>
__global__ __launch_bounds__(1024) void k_iwe_splat_lds(const float* __restrict__ flow, const float4* __restrict__ ev,
                                                        const int32_t* __restrict__ map_of_event,
                                                        const int32_t* __restrict__ ts_shift,
                                                        const float* __restrict__ w0, const float* __restrict__ w1,
                                                        int wstride, int B, int M, int H, int W, float S, float tref,
                                                        float tref_ts, int mode, int nch, int rows,
                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* img = (float*)smem_raw;  // [nch][rows][W]
  const int b = blockIdx.y, r0 = blockIdx.x * rows;
  const int nr = min(rows, H - r0);
  const int plane = rows * W;
  for (int q = threadIdx.x; q < nch * plane; q += blockDim.x) img[q] = 0.f;
  __syncthreads();
  const long HW = (long)H * W;
  const float lo = (float)r0, hi = (float)(r0 + nr);
  auto put = [&](int ch, float cy, float cx, float v) {
    if (v != 0.f && cy >= lo && cy < hi) atomicAdd(&img[ch * plane + ((int)cy - r0) * W + (int)cx], v);
  };
  // events in batches of IW_U per thread: all event loads first, then all flow gathers (which
  // depend on them), then the splats -- the two dependent memory latencies are paid once per batch
#define IW_U 4
  const bool pair = w0 && w1 == w0 + 1 && wstride == 2 && (((uintptr_t)w0) & 7) == 0;
  for (int e0 = threadIdx.x; e0 < M; e0 += IW_U * blockDim.x) {
    float4 qs[IW_U];
    float a0s[IW_U], a1s[IW_U], ts[IW_U], fys[IW_U], fxs[IW_U];
    int maps[IW_U];
#pragma unroll
    for (int u = 0; u < IW_U; ++u) {
      const int e = min(e0 + u * (int)blockDim.x, M - 1);  // clamped: loads stay unconditional
      const long i = (long)b * M + e;
      qs[u] = ev[i];
      ts[u] = ts_shift ? (float)ts_shift[e] : 0.f;
      maps[u] = map_of_event ? map_of_event[e] : 0;
      if (pair) {  // the two weights are adjacent floats (the reference's [B,N,2] polarity mask): one 8-byte load
        const float2 a = *(const float2*)(w0 + i * 2);
        a0s[u] = a.x, a1s[u] = a.y;
      } else {
        a0s[u] = w0 ? w0[i * wstride] : 1.0f;
        a1s[u] = w1 ? w1[i * wstride] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < IW_U; ++u) evf_event_flow(flow, maps[u], B, b, HW, qs[u].y, qs[u].z, W, fys[u], fxs[u]);
#pragma unroll
    for (int u = 0; u < IW_U; ++u) {
    if (e0 + u * (int)blockDim.x >= M) continue;
    const float4 q = qs[u];
    const float t = q.x + ts[u];
    float fy = fys[u], fx = fxs[u];
    const float a0 = a0s[u], a1 = a1s[u];
    if (mode & 2) {
      fy *= 0.f;
      fx *= 0.f;
    }
    const float tau = (mode & 8) ? (tref_ts - t) : t;
    const Warp w = evf_warp(t, q.y, q.z, fy, fx, tref, S);
    if (ROUND) {
      const float iy = rintf(w.wy), ix = rintf(w.wx);
      if (iy < 0.f || iy >= (float)H || ix < 0.f || ix >= (float)W) continue;
      put(0, iy, ix, a0);
      if (nch >= 2) put(1, iy, ix, a1);
      if (nch == 4) {
        put(2, iy, ix, tau * a0);
        put(3, iy, ix, tau * a1);
      }
    } else {
      const float cy[2] = {floorf(w.wy), floorf(w.wy + 1.0f)};
      const float cx[2] = {floorf(w.wx), floorf(w.wx + 1.0f)};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (cy[j] < 0.f || cy[j] >= (float)H || cx[k] < 0.f || cx[k] >= (float)W) continue;
          const float wt = fmaxf(0.f, 1.0f - fabsf(w.wy - cy[j])) * fmaxf(0.f, 1.0f - fabsf(w.wx - cx[k]));
          if (wt == 0.f) continue;
          put(0, cy[j], cx[k], wt * a0);
          if (nch >= 2) put(1, cy[j], cx[k], wt * a1);
          if (nch == 4) {
            const float wtau = wt * tau;
            put(2, cy[j], cx[k], wtau * a0);
            put(3, cy[j], cx[k], wtau * a1);
          }
        }
    }
    }
  }
#undef IW_U
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    float* dst = out + ((long)b * nch + ch) * HW + (long)r0 * W;
    for (int q = threadIdx.x; q < nr * W; q += blockDim.x) dst[q] = img[ch * plane + q];
  }
}

// Single flow map, one image plane <= 64 KiB, M <= 16 Ki events per sample (compute_pol_iwe at the
// benchmark shape): the sample's events stay in registers (16 per thread) and ONE 64 KiB LDS plane is reused
// in turn for flow_x, flow_y and each IWE channel:
//   stage flow_x (coalesced) -> wx of every event;  stage flow_y -> wy;  per channel: zero, LDS-atomic splat,
//   coalesced write-out.
// The two random 4-byte gathers per event (which bound k_iwe_splat_lds: they miss the 32 KiB L1 and queue at
// the L2) become LDS reads; HBM sees every input byte once, coalesced.  Same arithmetic, same bit-exact
// integer histograms.
#define IWR_EPT 15
#define IWR_THREADS 1024
#define IWR_NQ 4  // float4 per thread per 64 KiB plane
// PAIRW: the two weights are adjacent floats (the reference's [B,N,2] polarity mask) -> one 8-byte load;
// otherwise no weights at all (w0 = 1, single channel).  No per-event runtime branches: every uniform
// condition inside the unrolled event loops costs a basic block (and a spill) per event.
// When every weight of the sample is 0 or 1 (polarity masks) the splat uses INTEGER LDS atomics on one plane
// holding both channels as 16-bit counters (ds_add_u32 runs ~5x faster than ds_add_f32 on gfx950, measured);
// counts < 2^16 are exact in fp32, so the result is bit-identical to float accumulation.
typedef __attribute__((address_space(1))) void iw_glb_void;
typedef __attribute__((address_space(3))) void iw_lds_void;

template <bool PAIRW>
__global__ __launch_bounds__(IWR_THREADS) void k_iwe_splat_reg(const float* __restrict__ flow,
                                                               const float4* __restrict__ ev,
                                                               const float2* __restrict__ wpair, int B, int M, int H,
                                                               int W, float S, float tref, float zero_flow, int nch,
                                                               float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* pl = (float*)smem_raw;  // [2][H*W]: flow_x, flow_y (+ 1 KiB slack for the last DMA piece); then the image
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int HW = H * W, HQ = HW / 4;
  // Both flow planes of the sample (contiguous [2][HW] in memory and in LDS) arrive by LDS-DMA: no VGPRs, no
  // register -> LDS copy phase, and they are in flight together with the event and weight loads below.
  {
    const float4* f = (const float4*)(flow + (long)b * 2 * HW);
    for (int q0 = wv * 64; q0 < 2 * HQ; q0 += IWR_THREADS)
      __builtin_amdgcn_global_load_lds((iw_glb_void*)(f + min(q0 + lane, 2 * HQ - 1)), (iw_lds_void*)((float4*)pl + q0), 16, 0,
                                       0);
  }
  float t[IWR_EPT], y[IWR_EPT], x[IWR_EPT];
  int lin[IWR_EPT];
  float a0[IWR_EPT], a1[IWR_EPT];
  const float4* evb = ev + (long)b * M;
#pragma unroll
  for (int u = 0; u < IWR_EPT; ++u) {
    const float4 q = evb[min(tid + u * IWR_THREADS, M - 1)];
    t[u] = tref - q.x;
    y[u] = q.y, x[u] = q.z;
    lin[u] = min(max((int)(q.y * (float)W + q.z), 0), HW - 1);  // flow_idx = y*W + x in float, .long() (loss/flow.py:65-67)
  }
#pragma unroll
  for (int u = 0; u < IWR_EPT; ++u) {
    if (PAIRW) {
      const float2 a = wpair[(long)b * M + min(tid + u * IWR_THREADS, M - 1)];
      a0[u] = a.x, a1[u] = a.y;
    } else {
      a0[u] = 1.0f, a1[u] = 0.f;
    }
  }
  __syncthreads();  // (the compiler drains vmcnt before this barrier: DMA, events and weights have landed)
  // destination pixel (or -1) of every event: rounded indices, torch.round = half-to-even
  int dst[IWR_EPT];
  bool bin = true;
#pragma unroll
  for (int u = 0; u < IWR_EPT; ++u) {
    const float wx = x[u] + (t[u] * (pl[lin[u]] * zero_flow)) * S;        // channel 0 = horizontal (loss/flow.py:76)
    const float wy = y[u] + (t[u] * (pl[HW + lin[u]] * zero_flow)) * S;   // channel 1 = vertical (:75)
    const float iy = rintf(wy), ix = rintf(wx);
    const bool in = !(iy < 0.f || iy >= (float)H || ix < 0.f || ix >= (float)W) && tid + u * IWR_THREADS < M;
    dst[u] = in ? (int)(iy * (float)W + ix) : -1;
    bin = bin && (a0[u] == 0.f || a0[u] == 1.f) && (a1[u] == 0.f || a1[u] == 1.f);
  }
  const bool packed = __syncthreads_and(bin) && M < 65536;  // also: every wave is done reading the flow planes
  if (packed) {
    for (int q = tid; q < HQ; q += IWR_THREADS) ((uint4*)pl)[q] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    unsigned* cnt = (unsigned*)pl;
#pragma unroll
    for (int u = 0; u < IWR_EPT; ++u) {
      const unsigned inc = (a0[u] != 0.f ? 1u : 0u) | (a1[u] != 0.f ? 0x10000u : 0u);
      if (dst[u] >= 0 && inc) atomicAdd(&cnt[dst[u]], inc);
    }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      float4* o = (float4*)(out + ((long)b * nch + ch) * HW);
      const int sh = 16 * ch;
      for (int q = tid; q < HQ; q += IWR_THREADS) {
        const uint4 c = ((const uint4*)pl)[q];
        o[q] = make_float4((float)((c.x >> sh) & 0xFFFFu), (float)((c.y >> sh) & 0xFFFFu), (float)((c.z >> sh) & 0xFFFFu),
                           (float)((c.w >> sh) & 0xFFFFu));
      }
    }
    return;
  }
  // general weights: one fp32 plane per channel (the two planes are free now)
  for (int ch = 0; ch < nch; ++ch) {
    float* im = pl + (ch & 1) * HW;
    for (int q = tid; q < HQ; q += IWR_THREADS) ((float4*)im)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < IWR_EPT; ++u) {
    if (dst[u] >= 0 && a0[u] != 0.f) atomicAdd(&pl[dst[u]], a0[u]);
    if (nch > 1 && dst[u] >= 0 && a1[u] != 0.f) atomicAdd(&pl[HW + dst[u]], a1[u]);
  }
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    float4* o = (float4*)(out + ((long)b * nch + ch) * HW);
    const float4* im = (const float4*)(pl + ch * HW);
    for (int q = tid; q < HQ; q += IWR_THREADS) o[q] = im[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_iwe_splat_one: compute_pol_iwe at the metric's own shape (8 x 15k events, 128 x 128: BASELINE "IWE-warp GB/s") in ONE
// launch, without a zero-fill launch and without global atomics on the image.
//
// What bounds the event-parallel kernel there (k_iwe_splat: 7.6-10 us by rocprof + a 2-5 us fill launch) is the rate of
// device-scope fp32 atomics (21.4 G/s measured, tools/probes/atomic_probe.hip: 120 k events = 5.6 us); the stripe kernels
// (k_iwe_splat_lds, one block per band of rows) re-scan a sample's events once per band and pay its two random flow gathers
// 8-16 times over (19-20 us, profiles/design_log_r01-r04.md).  Here every event is touched ONCE by an event-parallel phase and
// the image is built in LDS by ONE block per sample -- the block that happens to finish the sample's first phase last:
//
//   phase 1  grid (ceil(M / 1024), B): one event per thread -- event, flow gather, warp, rounded destination -- and ONE
//            4-byte entry per event [dst pixel | valid | a0 == 1 | a1 == 1 | weights not 0 / 1] written into the sample's own
//            region of `out` (M <= nch * H * W entries fit; nobody reads `out` before it is complete);
//   ticket   every block takes a DEPARTURE ticket of its sample (agent-scope acq_rel: the release writes the entries back
//            past this XCD's L2, the acquire lets the last block see the other XCDs' entries).  Nobody ever WAITS: no
//            co-residency assumption (ADVICE r04 on spinning tickets); all blocks but the last simply exit;
//   phase 2  the block that drew the sample's last ticket reads the M entries (16 per thread, coalesced) into registers,
//            clears a 64 KiB LDS plane, bins the entries with LDS atomics -- both polarity counts as the 16-bit halves of one
//            u32 plane when every weight is 0 / 1 (ds_add_u32; counts < 2^16 exact in fp32, bit-identical to float adds), one
//            fp32 plane per channel otherwise -- and writes the sample's nch x H x W image with coalesced 16-byte stores,
//            over the entries it has already consumed.  The ticket word is left at zero for the next call.
// Same per-event arithmetic as k_iwe_splat (evf_event_flow / evf_warp / rintf): integer histograms bit-exact.
//
// MEASURED (round 6, rocprofv3 kernel durations on two boxes, probe builds that cut the kernel short): phase 1 alone -- every
// event read, its flow gathered, one entry stored -- 4.2-5.0 us (3.4 at best): the floor of ANY kernel that touches each event
// once behind a dependent gather at this size; + the ticket 6.0 (relaxed) / 6.9 (acq_rel); + phase 2 10.0-11.3, of which the
// 15 k LDS atomics of the one block per sample 1.2 and its 128 KiB image store 2.1.  That equals the two launches it replaces
// (k_iwe_splat 9.1 + k_evf_fill 1.8; under bench.py's HIP-event bracket 15.5 against 13.3 us), so it is NOT the default:
// EVF_IWE_ONE=1 selects it (tests/test_gpu_events.py holds it bit-exact).  What would beat both is phase 2 spread over the
// sample's blocks (a row stripe each: ~1 us), which needs the blocks to WAIT for each other's entries -- an arrival barrier
// inside the launch, i.e. co-residency of the grid, which a CU mask or another stream's blocks can break; a
// departure ticket cannot hand the work to more than the one block that draws it last.
// ---------------------------------------------------------------------------------------------------------------------
#define IW1_SLOTS 64
#define IW1_MAXB 32
#define IW1_EPT 16
__device__ unsigned iw1_ticket[IW1_SLOTS * IW1_MAXB];  // departure tickets, one word per (call slot, sample); zero between calls

__global__ __launch_bounds__(1024) void k_iwe_splat_one(const float* __restrict__ flow, const float4* __restrict__ ev,
                                                        const int32_t* __restrict__ map_of_event,
                                                        const int32_t* __restrict__ ts_shift, const float* __restrict__ w0,
                                                        const float* __restrict__ w1, int wstride, int B, int M, int H, int W,
                                                        float S, float tref, int mode, int nch, unsigned* __restrict__ ticket,
                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int s_last;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int HW = H * W;
  unsigned* list = (unsigned*)(out + (long)b * nch * HW);  // the sample's entries, M <= nch * HW
  {
    const int e = blockIdx.x * 1024 + tid;
    if (e < M) {
      const long i = (long)b * M + e;
      const float4 q = ev[i];
      float t = q.x;
      if (ts_shift) t += (float)ts_shift[e];
      float fy, fx;
      evf_event_flow(flow, map_of_event ? map_of_event[e] : 0, B, b, (long)HW, q.y, q.z, W, fy, fx);
      if (mode & 2) {  // `flow * 0` keeps the sign / NaN semantics of the reference
        fy *= 0.f;
        fx *= 0.f;
      }
      const float a0 = w0 ? w0[i * wstride] : 1.0f;
      const float a1 = w1 ? w1[i * wstride] : 0.0f;
      const Warp w = evf_warp(t, q.y, q.z, fy, fx, tref, S);
      const float iy = rintf(w.wy), ix = rintf(w.wx);  // half-to-even like torch.round
      unsigned ent = 0u;
      if (!(iy < 0.f || iy >= (float)H || ix < 0.f || ix >= (float)W)) {
        const bool gen = !((a0 == 0.f || a0 == 1.f) && (a1 == 0.f || a1 == 1.f));
        ent = (unsigned)(iy * (float)W + ix) | (1u << 27) | (a0 == 1.f ? 1u << 28 : 0u) | (a1 == 1.f ? 1u << 29 : 0u) |
              (gen ? 1u << 30 : 0u);
      }
      // an agent-scope store (written through this XCD's L2) that has COMPLETED before the block's ticket is taken: the
      // workgroup barrier alone does not wait for stores in flight
      __hip_atomic_store(list + e, ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket + b, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t + 1u >= gridDim.x;  // (>=: a ticket left behind by a launch that never finished cannot lock the sample out)
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last block of the sample to leave phase 1: every entry of the sample is written and visible
  unsigned ent[IW1_EPT];
  bool gen = false;
#pragma unroll
  for (int u = 0; u < IW1_EPT; ++u) {
    const int e = tid + u * 1024;
    ent[u] = e < M ? __hip_atomic_load(list + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    gen = gen || (ent[u] >> 30 & 1u);
  }
  const int any_gen = __syncthreads_or(gen);  // (also: every thread holds its entries, `out` may be overwritten from here on)
  const int HQ = HW / 4;
  if (!any_gen) {
    uint4* pl = (uint4*)smem_raw;  // [HW] u32: low half = channel 0 count, high half = channel 1 count
    for (int q = tid; q < HQ; q += 1024) pl[q] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    unsigned* cnt = (unsigned*)smem_raw;
#pragma unroll
    for (int u = 0; u < IW1_EPT; ++u) {
      const unsigned inc = (ent[u] >> 28 & 1u) | (nch > 1 ? (ent[u] >> 29 & 1u) << 16 : 0u);
      if ((ent[u] >> 27 & 1u) && inc) atomicAdd(&cnt[ent[u] & 0x7FFFFFFu], inc);
    }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      float4* o = (float4*)(out + ((long)b * nch + ch) * HW);
      const int sh = 16 * ch;
      for (int q = tid; q < HQ; q += 1024) {
        const uint4 c = pl[q];
        o[q] = make_float4((float)((c.x >> sh) & 0xFFFFu), (float)((c.y >> sh) & 0xFFFFu), (float)((c.z >> sh) & 0xFFFFu),
                           (float)((c.w >> sh) & 0xFFFFu));
      }
    }
  } else {
    float* im = (float*)smem_raw;  // [nch][HW] fp32
    for (int q = tid; q < nch * HQ; q += 1024) ((float4*)im)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < IW1_EPT; ++u) {
      const int e = tid + u * 1024;
      if (ent[u] >> 27 & 1u) {
        const long i = (long)b * M + e;
        const float a0 = w0 ? w0[i * wstride] : 1.0f;
        const float a1 = w1 ? w1[i * wstride] : 0.0f;
        const int px = (int)(ent[u] & 0x7FFFFFFu);
        if (a0 != 0.f) atomicAdd(&im[px], a0);
        if (nch > 1 && a1 != 0.f) atomicAdd(&im[HW + px], a1);
      }
    }
    __syncthreads();
    float4* o = (float4*)(out + (long)b * nch * HW);
    for (int q = tid; q < nch * HQ; q += 1024) o[q] = ((const float4*)im)[q];
  }
  if (tid == 0) __hip_atomic_store(ticket + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

extern "C" int evf_iwe_splat(const float* flow, const float* ev, const int32_t* map_of_event, const int32_t* ts_shift,
                             const float* w0, const float* w1, int wstride, int B, int M, int H, int W,
                             float flow_scaling, float tref, float tref_ts, int mode, int nch, float* out, void* stream) {
  if (!flow || !out || B <= 0 || M < 0 || H <= 0 || W <= 0 || (nch != 1 && nch != 2 && nch != 4)) return EVF_EINVAL;
  if (nch == 4 && !(mode & 4)) return EVF_EINVAL;
  if (M > 0 && !ev) return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  if (M == 0) return evf_hip(evf_memset_async(out, 0, sizeof(float) * (size_t)B * nch * H * W, st));
  // stripe height: as tall as 128 KiB of LDS allows, shrunk (down to 8 rows) until the grid has
  // >= 256 blocks; every block re-reads the sample's events (L2 resident), so shorter stripes
  // trade redundant event reads for parallelism
  // register-resident events + one reused LDS plane: single flow map, plane <= 64 KiB, M <= 16 Ki events,
  // enough samples to fill the CUs
  // Register-resident events + one reused LDS plane (k_iwe_splat_reg): rounded indices, single flow map, no
  // timestamp images / shifts, plane <= 64 KiB and a multiple of 16 bytes, M <= 15 Ki events, polarity weights as
  // an interleaved pair (or none), enough samples to fill the CUs -- i.e. compute_pol_iwe at the benchmark shape.
  {
    static const int reg_min_b = getenv("EVF_IWE_REG_MIN_B") ? atoi(getenv("EVF_IWE_REG_MIN_B")) : 16;
    const bool pairw = w0 && w1 == w0 + 1 && wstride == 2 && (((uintptr_t)w0) & 7) == 0 && nch == 2;
    const bool now = !w0 && !w1 && nch == 1;
    if ((mode & 1) && !(mode & 12) && !map_of_event && !ts_shift && (pairw || now) && (long)H * W * 4 <= 64 * 1024 &&
        ((H * W) & 3) == 0 && H * W <= 4 * IWR_NQ * IWR_THREADS && M <= IWR_EPT * IWR_THREADS && B >= reg_min_b && (((uintptr_t)flow) & 15) == 0 &&
        (((uintptr_t)out) & 15) == 0) {
      const size_t lds = (size_t)H * W * 8 + 1024;  // two planes + slack for the last (clamped) DMA piece
      const float zf = (mode & 2) ? 0.f : 1.f;  // `flow * 0` (FWL/RSAT reference images) keeps NaN/sign semantics
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_iwe_splat_reg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 129 * 1024);
        (void)hipFuncSetAttribute((const void*)k_iwe_splat_reg<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 129 * 1024);
        attr_set = true;
      }
      if (pairw)
        hipLaunchKernelGGL(k_iwe_splat_reg<true>, dim3(B), dim3(IWR_THREADS), lds, st, flow, (const float4*)ev,
                           (const float2*)w0, B, M, H, W, flow_scaling, tref, zf, nch, out);
      else
        hipLaunchKernelGGL(k_iwe_splat_reg<false>, dim3(B), dim3(IWR_THREADS), lds, st, flow, (const float4*)ev,
                           (const float2*)nullptr, B, M, H, W, flow_scaling, tref, zf, nch, out);
      return evf_status();
    }
  }
  const int max_rows = (128 * 1024) / (nch * W * 4);
  // small problems (a few 100k events) are latency bound: there the one-launch kernel below (one entry per event, the image
  // built in LDS by the sample's last block) or, for shapes it does not serve, the plain global-atomic kernel behind a
  // zero-fill; the LDS stripe version wins once the device-scope atomics saturate
  static const long lds_min = getenv("EVF_IWE_LDS_MIN") ? atol(getenv("EVF_IWE_LDS_MIN")) : 400000;
  {
    // k_iwe_splat_one: rounded indices, count images only (no timestamp images), the sample's events fit 16 per thread and
    // its entries fit its own region of `out`, the image fits the LDS plane(s)
    // OPT-IN (EVF_IWE_ONE=1, read per call): measured on the MI355X (rocprofv3 kernel durations, 8 x 15k events, 128 x 128) it
    // is no faster than the two launches it replaces -- 10.0-11.3 us against 9.1 (scatter) + 1.8 (fill); see the kernel's header
    const char* one_env = getenv("EVF_IWE_ONE");
    const int one_on = one_env ? atoi(one_env) : 0;
    const long HWl = (long)H * W;
    if (one_on && (mode & 1) && !(mode & 12) && (nch == 1 || nch == 2) && M <= IW1_EPT * 1024 && (long)M <= nch * HWl &&
        HWl <= 16384 && (HWl & 3) == 0 && B <= IW1_MAXB && (long)B * M < lds_min && (((uintptr_t)out) & 15) == 0) {
      static unsigned* tick_base = nullptr;
      static unsigned slot = 0;
      static std::mutex mu;
      unsigned* tk;
      {
        std::lock_guard<std::mutex> g(mu);
        if (!tick_base) {
          const int rc = evf_hip(hipGetSymbolAddress((void**)&tick_base, HIP_SYMBOL(iw1_ticket)));
          if (rc) return rc;
          (void)hipFuncSetAttribute((const void*)k_iwe_splat_one, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        }
        // a slot of ticket words per call, round robin: calls of one stream never overlap, calls of different streams that do
        // overlap use different words (64 slots in flight)
        tk = tick_base + (size_t)(slot++ % IW1_SLOTS) * IW1_MAXB;
      }
      const size_t lds = (size_t)nch * HWl * 4 > 65536 ? (size_t)nch * HWl * 4 : 65536;
      hipLaunchKernelGGL(k_iwe_splat_one, dim3(evf_cdiv(M, 1024), B), dim3(1024), lds, st, flow, (const float4*)ev, map_of_event,
                         ts_shift, w0, w1, wstride, B, M, H, W, flow_scaling, tref, mode, nch, tk, out);
      return evf_status();
    }
  }
  if (max_rows >= 1 && (long)B * M >= lds_min) {
    int rows = max_rows < H ? max_rows : H;
    while (rows > 8 && (long)B * evf_cdiv(H, rows) < 256) rows = (rows + 1) / 2;
    const size_t lds = (size_t)nch * rows * W * 4;
    static bool attr[2] = {false, false};
    const int r = (mode & 1) ? 1 : 0;
    if (!attr[r]) {
      if (r)
        (void)hipFuncSetAttribute((const void*)k_iwe_splat_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      else
        (void)hipFuncSetAttribute((const void*)k_iwe_splat_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr[r] = true;
    }
    dim3 grid(evf_cdiv(H, rows), B), block(1024);
    if (r)
      hipLaunchKernelGGL(k_iwe_splat_lds<true>, grid, block, lds, st, flow, (const float4*)ev, map_of_event, ts_shift, w0, w1,
                         wstride, B, M, H, W, flow_scaling, tref, tref_ts, mode, nch, rows, out);
    else
      hipLaunchKernelGGL(k_iwe_splat_lds<false>, grid, block, lds, st, flow, (const float4*)ev, map_of_event, ts_shift, w0,
                         w1, wstride, B, M, H, W, flow_scaling, tref, tref_ts, mode, nch, rows, out);
    return evf_status();
  }
  // very wide images: global-atomic fallback kernel
  int rc = evf_hip(evf_memset_async(out, 0, sizeof(float) * (size_t)B * nch * H * W, st));
  if (rc) return rc;
  dim3 grid(evf_cdiv((long)B * M, 256)), block(256);
  if (mode & 1)
    hipLaunchKernelGGL(k_iwe_splat<true>, grid, block, 0, st, flow, (const float4*)ev, map_of_event, ts_shift, w0, w1,
                       wstride, B, M, H, W, flow_scaling, tref, tref_ts, mode, nch, out);
  else
    hipLaunchKernelGGL(k_iwe_splat<false>, grid, block, 0, st, flow, (const float4*)ev, map_of_event, ts_shift, w0, w1,
                       wstride, B, M, H, W, flow_scaling, tref, tref_ts, mode, nch, out);
  return evf_status();
}

// --------------------------------------------------------------------------
// materialising forms of get_interpolation / interpolate (API parity with
// utils/iwe.py:20-92; the fused kernels above never build these tensors)
// --------------------------------------------------------------------------
template <bool ROUND>
__global__ void k_get_interpolation(const float4* __restrict__ ev, const float2* __restrict__ evflow, int B, int N, int H,
                                    int W, float S, float tref, float* __restrict__ idx, float* __restrict__ wgt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * N) return;
  const int b = (int)(i / N), e = (int)(i - (long)b * N);
  const float4 q = ev[i];
  const float2 f = evflow[i];  // (fy, fx)
  const Warp w = evf_warp(q.x, q.y, q.z, f.x, f.y, tref, S);
  if (ROUND) {
    const float iy = rintf(w.wy), ix = rintf(w.wx);
    const bool ok = !(iy < 0.f || iy >= (float)H || ix < 0.f || ix >= (float)W);
    idx[i] = ok ? iy * (float)W + ix : 0.f;
    wgt[i] = ok ? 1.f : 0.f;
  } else {
    const float cy[2] = {floorf(w.wy), floorf(w.wy + 1.0f)};
    const float cx[2] = {floorf(w.wx), floorf(w.wx + 1.0f)};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float ay = fmaxf(0.f, 1.0f - fabsf(w.wy - cy[j])), ax = fmaxf(0.f, 1.0f - fabsf(w.wx - cx[k]));
        const bool ok = !(cy[j] < 0.f || cy[j] >= (float)H || cx[k] < 0.f || cx[k] >= (float)W);
        const long o = ((long)b * 4 + (j * 2 + k)) * N + e;  // corner blocks of N: tl, tr, bl, br (:53-57)
        idx[o] = ok ? cy[j] * (float)W + cx[k] : 0.f;
        wgt[o] = ok ? ay * ax : 0.f;
      }
  }
}

extern "C" int evf_get_interpolation(const float* ev, const float* evflow, int B, int N, int H, int W, float flow_scaling,
                                     float tref, int round_idx, float* idx, float* weights, void* stream) {
  if (!ev || !evflow || !idx || !weights || B <= 0 || N <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  dim3 grid(evf_cdiv((long)B * N, 256)), block(256);
  if (round_idx)
    hipLaunchKernelGGL(k_get_interpolation<true>, grid, block, 0, EVF_STREAM(stream), (const float4*)ev,
                       (const float2*)evflow, B, N, H, W, flow_scaling, tref, idx, weights);
  else
    hipLaunchKernelGGL(k_get_interpolation<false>, grid, block, 0, EVF_STREAM(stream), (const float4*)ev,
                       (const float2*)evflow, B, N, H, W, flow_scaling, tref, idx, weights);
  return evf_status();
}

__global__ void k_interpolate(const float* __restrict__ idx, const float* __restrict__ wgt, const float* __restrict__ pm,
                              int pstride, int B, int M, int HW, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * M) return;
  const int b = (int)(i / M);
  float w = wgt[i];
  if (pm) w *= pm[i * pstride];
  const long px = (long)idx[i];
  if (px < 0 || px >= HW) return;
  if (w != 0.f) evf_atomic_add(out + (long)b * HW + px, w);
}

extern "C" int evf_interpolate(const float* idx, const float* weights, const float* pol_mask, int pstride, int B, int M,
                               int H, int W, float* out, void* stream) {
  if (!idx || !weights || !out || B <= 0 || M <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  int rc = evf_hip(evf_memset_async(out, 0, sizeof(float) * (size_t)B * H * W, st));
  if (rc) return rc;
  hipLaunchKernelGGL(k_interpolate, dim3(evf_cdiv((long)B * M, 256)), dim3(256), 0, st, idx, weights, pol_mask, pstride,
                     B, M, H * W, out);
  return evf_status();
}

// --------------------------------------------------------------------------
// contrast-maximisation loss, forward
// --------------------------------------------------------------------------
// images [S][B][8][HW]: channel = dir*4 + {0: I_pos, 1: I_neg, 2: TS_pos, 3: TS_neg}
__global__ void k_cm_splat(const float* __restrict__ flow, const float4* __restrict__ ev, const float2* __restrict__ pol,
                           const int32_t* __restrict__ ev_pass, int S, int Pm, int P, int B, int M, int H, int W,
                           float Sc, float* __restrict__ images) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * M) return;
  const int s = blockIdx.y;
  const int b = (int)(i / M), e = (int)(i - (long)b * M);
  const float4 q = ev[i];
  const int pass = ev_pass[e];
  const float t = q.x + (float)pass;  // event_list[:, :, 0:1] += passes (loss/flow.py:90)
  const float2 pm = pol[i];
  const long HW = (long)H * W;
  float fy, fx;
  evf_event_flow(flow + (long)s * Pm * B * 2 * HW, Pm == 1 ? 0 : pass, B, b, HW, q.y, q.z, W, fy, fx);
  float* o = images + ((long)s * B + b) * 8 * HW;
  const float maxts = (float)P;
  // forward: t_ref = P, timestamp image of t          (loss/flow.py:196-211)
  evf_splat<false>(evf_warp(t, q.y, q.z, fy, fx, maxts, Sc), H, W, pm.x, pm.y, t, o, o + HW, o + 2 * HW, o + 3 * HW);
  // backward: t_ref = 0, timestamp image of (P - t)   (loss/flow.py:229-244)
  evf_splat<false>(evf_warp(t, q.y, q.z, fy, fx, 0.f, Sc), H, W, pm.x, pm.y, maxts - t, o + 4 * HW, o + 5 * HW,
                   o + 6 * HW, o + 7 * HW);
}

// ---- LDS-privatised splat (used when the caller passes a workspace) ----------------------------------------
// Device-scope fp32 atomics resolve at the L2s at ~21 G/s: 16 per event and scale made k_cm_splat 1.1 ms of the
// EV-FlowNet step (8 x 50 k events x 4 scales).  Here a pre-pass stores, per (scale, event), the warped
// coordinates of both directions next to the polarity weights (the two random flow gathers happen once); then one
// block per (scale, sample, direction, stripe of image rows) keeps its stripe of the four images in LDS, streams
// the sample's pre-warped events (16 B, coalesced) and accumulates with LDS atomics, and finally writes the
// stripe with plain coalesced stores -- no zero-fill of `images`, no global atomics.
__device__ __forceinline__ void cm_prewarp_body(int bx, int s, const float* __restrict__ flow, const float4* __restrict__ ev,
                                                const float2* __restrict__ pol, const int32_t* __restrict__ ev_pass, int S, int Pm,
                                                int P, int B, int M, int H, int W, float Sc, float4* __restrict__ warp,
                                                float* __restrict__ tabs, float* __restrict__ ys) {
  const long i = (long)bx * blockDim.x + threadIdx.x;
  if (i >= (long)B * M) return;
  const int b = (int)(i / M), e = (int)(i - (long)b * M);
  const float4 q = ev[i];
  const int pass = ev_pass[e];
  const float t = q.x + (float)pass;  // event_list[:, :, 0:1] += passes (loss/flow.py:90)
  const float2 pm = pol[i];
  const long HW = (long)H * W;
  float fy, fx;
  evf_event_flow(flow + (long)s * Pm * B * 2 * HW, Pm == 1 ? 0 : pass, B, b, HW, q.y, q.z, W, fy, fx);
  const Warp f = evf_warp(t, q.y, q.z, fy, fx, (float)P, Sc), g = evf_warp(t, q.y, q.z, fy, fx, 0.f, Sc);
  float4* o = warp + ((long)(s * B + b) * 2) * M + e;
  o[0] = make_float4(f.wy, f.wx, pm.x, pm.y);
  o[M] = make_float4(g.wy, g.wx, pm.x, pm.y);
  // the warped ROW once more, alone: a stripe block of k_cm_splat_lds decides on 4 bytes per event whether the event is its own
  float* oy = ys + ((long)(s * B + b) * 2) * M + e;
  oy[0] = f.wy, oy[M] = g.wy;
  if (s == 0) tabs[i] = t;
}
__global__ void k_cm_prewarp(const float* __restrict__ flow, const float4* __restrict__ ev, const float2* __restrict__ pol,
                             const int32_t* __restrict__ ev_pass, int S, int Pm, int P, int B, int M, int H, int W,
                             float Sc, float4* __restrict__ warp, float* __restrict__ tabs, float* __restrict__ ys) {
  cm_prewarp_body(blockIdx.x, blockIdx.y, flow, ev, pol, ev_pass, S, Pm, P, B, M, H, W, Sc, warp, tabs, ys);
}

struct CmFin {  // what the LAST block of k_cm_splat_lds needs to finish the loss (stats == null: separate launches do it)
  float* stats;          // [S][B][2][2], zeroed by k_cm_pre
  unsigned* ticket;      // zeroed by k_cm_pre
  const float* part;     // smoothness partials [S][nblk_per_scale]
  float* loss;
  int S, Pm, nblk_per_scale, comps, loss_scaling;
  float weight;
};
__device__ void cm_finalize_body(float* red, const float* stats, const float* __restrict__ part, int S, int B, int Pm,
                                 int nblk_per_scale, float weight, int comps, int loss_scaling, float* __restrict__ loss);

__global__ __launch_bounds__(1024) void k_cm_splat_lds(const float4* __restrict__ warp, const float* __restrict__ tabs,
                                                       const float* __restrict__ ys, int B, int M, int H, int W, int rows, float P,
                                                       float* __restrict__ images, CmFin fin) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* img = (float*)smem_raw;  // [4][rows*W]: I_pos, I_neg, TS_pos, TS_neg of this direction
  const int sbd = blockIdx.y, d = sbd & 1, b = (sbd >> 1) % B;
  const int r0 = blockIdx.x * rows, nr = min(rows, H - r0), plane = rows * W;
  for (int q = threadIdx.x; q < 4 * plane; q += blockDim.x) img[q] = 0.f;
  __syncthreads();
  const float4* __restrict__ wp = warp + (long)sbd * M;
  const float* __restrict__ tb = tabs + (long)b * M;
  const float lo = (float)r0, hi = (float)(r0 + nr), fW = (float)W;
  auto one = [&](const float4 w4, const float t) {
    const float tau = d ? P - t : t;  // timestamp images: t forward, (P - t) backward (loss/flow.py:196-211, 229-244)
    const float cy[2] = {floorf(w4.x), floorf(w4.x + 1.0f)};
    const float cx[2] = {floorf(w4.y), floorf(w4.y + 1.0f)};
    float ay[2], ax[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      ay[k] = fmaxf(0.f, 1.0f - fabsf(w4.x - cy[k]));
      ax[k] = fmaxf(0.f, 1.0f - fabsf(w4.y - cx[k]));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!(cy[j] >= lo && cy[j] < hi && cx[i] >= 0.f && cx[i] < fW)) continue;  // other stripes / outside (NaN too)
        const float wt = ay[j] * ax[i];
        if (wt == 0.f) continue;
        float* px = img + ((int)cy[j] - r0) * W + (int)cx[i];
        const float wtau = wt * tau;
        const float v0 = wt * w4.z, v1 = wt * w4.w, u0 = wtau * w4.z, u1 = wtau * w4.w;
#ifdef CM_PROBE_NOATOM  // (probe build: plain stores instead of the float atomics -- wrong sums, valid timing)
        if (v0 != 0.f) px[0] = v0;
        if (v1 != 0.f) px[plane] = v1;
        if (u0 != 0.f) px[2 * plane] = u0;
        if (u1 != 0.f) px[3 * plane] = u1;
#else
        if (v0 != 0.f) atomicAdd(px, v0);
        if (v1 != 0.f) atomicAdd(px + plane, v1);
        if (u0 != 0.f) atomicAdd(px + 2 * plane, u0);
        if (u1 != 0.f) atomicAdd(px + 3 * plane, u1);
#endif
      }
  };
  // The scan.  Every stripe block passes over all M records of its sample; one event in H / rows is its own.  The warped ROW alone
  // (4 bytes, `ys`) decides; CM_CH rows per thread are requested together, and the records of the accepted events are fetched CM_NB
  // at a time, all loads of a batch issued before the first is used (a load under a divergent branch is waited for on the spot).
  // Probe builds at 256 x 256 x 50 k events x 4 scales (512 blocks, us per launch): 189 complete = 39 without any event (fill,
  // write-out, statistics, finish) + 21 rows tested and records fetched + 56 the weights and LDS stores of `one` + 73 that the float
  // atomics cost over plain stores.  Tried on top, both level: the wave compacting its accepted events through an LDS queue so that
  // `one` runs on full waves (188) -- the LDS float atomic is paid per LANE, 0.78 per CU and ns (tools/probes/lds_atomic_probe.hip;
  // u32 14.8, u64 10.5) --, and 64-bit fixed-point slots (exact sums, 13x the atomic rate, but half the stripe height: every fixed
  // and per-block cost twice; 424 us before the scan was cheap).
#define CM_CH 16
#define CM_NB 4
  const float* __restrict__ yp = ys + (long)sbd * M;
#ifdef CM_PROBE_NOSCAN  // (probe build: no events at all)
  M = 0;
#endif
  auto mine = [&](float y) {  // (the row tests of `one`; NaN: no)
    const float c0 = floorf(y), c1 = floorf(y + 1.0f);
    return (c0 >= lo && c0 < hi) || (c1 >= lo && c1 < hi);
  };
  for (int e0 = threadIdx.x; e0 < M; e0 += CM_CH * (int)blockDim.x) {
    float y[CM_CH];
#pragma unroll
    for (int k = 0; k < CM_CH; ++k) y[k] = yp[min(e0 + k * (int)blockDim.x, M - 1)];  // (clamped: the loads stay unconditional)
    unsigned acc = 0u;
#pragma unroll
    for (int k = 0; k < CM_CH; ++k)
      if (e0 + k * (int)blockDim.x < M && mine(y[k])) acc |= 1u << k;
    while (__builtin_amdgcn_ballot_w64(acc != 0u) != 0ull) {  // (wave-uniform trip count: the loads below sit under no divergent branch)
      int idx[CM_NB];
      bool ok[CM_NB];
#pragma unroll
      for (int j = 0; j < CM_NB; ++j) {
        ok[j] = acc != 0u;
        const int k = ok[j] ? __builtin_ctz(acc) : 0;
        acc &= acc - 1u;  // (0 stays 0)
        idx[j] = min(e0 + k * (int)blockDim.x, M - 1);
      }
      float4 w4[CM_NB];
      float tt[CM_NB];
#pragma unroll
      for (int j = 0; j < CM_NB; ++j) w4[j] = wp[idx[j]], tt[j] = tb[idx[j]];
#ifdef CM_PROBE_NOONE  // (probe build: rows tested, records fetched, nothing accumulated)
#pragma unroll
      for (int j = 0; j < CM_NB; ++j) asm volatile("" ::"v"(w4[j].x), "v"(w4[j].y), "v"(w4[j].z), "v"(w4[j].w), "v"(tt[j]));
#else
#pragma unroll
      for (int j = 0; j < CM_NB; ++j)
        if (ok[j]) one(w4[j], tt[j]);
#endif
    }
  }
  __syncthreads();
  const long HW = (long)H * W;
  float* o = images + ((long)(sbd >> 1) * 8 + d * 4) * HW + (long)r0 * W;
  const int n = nr * W;
  for (int q = threadIdx.x; q < 4 * n; q += blockDim.x) {
    const int ch = q / n, r = q - ch * n;
    o[(long)ch * HW + r] = img[ch * plane + r];
  }
  if (!fin.stats) return;
  // ---- this stripe's part of the image statistics, straight from LDS (k_cm_reduce re-read the images from memory) ...
  __shared__ float red[16];
  __shared__ int s_last;
  float sq = 0.f, nz = 0.f;
  for (int p = threadIdx.x; p < n; p += blockDim.x) {
    const float ip = img[p], in = img[plane + p];
    const float ap = img[2 * plane + p] / (ip + 1e-9f) / P;  // loss/flow.py:212-215
    const float an = img[3 * plane + p] / (in + 1e-9f) / P;
    sq += ap * ap + an * an;
    nz += (ip + in > 0.f) ? 1.f : 0.f;
  }
  sq = evf_block_sum(sq, red);
  nz = evf_block_sum(nz, red);
  if (threadIdx.x == 0) {
    evf_atomic_add(fin.stats + sbd * 2, sq);
    evf_atomic_add(fin.stats + sbd * 2 + 1, nz);
    // The two sums are device-scope ATOMICS (performed at the memory side, never cached) and the last block reads them with
    // device-scope loads: it is enough that they have COMPLETED before the ticket is drawn -- a workgroup-scope release is that wait
    // and no cache maintenance.  An agent-scope release here (__threadfence, or an ACQ_REL ticket) writes the XCD's dirty L2 back --
    // the image stripes the blocks have just stored, which the finish never reads -- once per block: 51 us of the 205 us launch at
    // 256 x 256 x 4 scales x 8 samples were this.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const unsigned t = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.x * gridDim.y - 1;
  }
  __syncthreads();
  // ... and the block that arrives last, when every stripe has added its part, finishes the loss (k_cm_finalize)
  if (s_last) cm_finalize_body(red, fin.stats, fin.part, fin.S, B, fin.Pm, fin.nblk_per_scale, fin.weight, fin.comps,
                               fin.loss_scaling, fin.loss);
}

static int cm_lds_rows(int S, int B, int H, int W) {
  int rows = 8192 / W;  // 4 planes x rows x W floats <= 128 KiB of LDS
  if (rows > H) rows = H;
  while (rows > 8 && (long)S * B * 2 * evf_cdiv(H, rows) < 512) rows >>= 1;  // enough blocks to fill the CUs twice
  return rows;
}

// floats of workspace that make evf_cm_loss_fwd take the LDS-privatised splat (0: image rows too wide for LDS)
extern "C" int64_t evf_cm_loss_ws(int S, int B, int M, int H, int W) {
  if (S <= 0 || B <= 0 || M <= 0 || H <= 0 || W <= 0 || W > 2048) return 0;
  // pre-warped records, event times, the ticket of the merged launch, the warped rows alone
  return (int64_t)S * B * 2 * M * 4 + (int64_t)B * M + 4 + (int64_t)S * B * 2 * M;
}

// stats [S][B][2][2] += (sum over px of A_pos^2 + A_neg^2, #px with I_pos+I_neg > 0)
#define CM_RED_CHUNK 2048
__global__ void k_cm_reduce(const float* __restrict__ images, int HW, float P, float* __restrict__ stats) {
  __shared__ float red[16];
  const int sbd = blockIdx.y;  // (s*B + b)*2 + dir
  const float* im = images + ((long)(sbd >> 1) * 8 + (sbd & 1) * 4) * HW;
  float sq = 0.f, nz = 0.f;
  const int p0 = blockIdx.x * CM_RED_CHUNK;
  for (int p = p0 + threadIdx.x; p < min(HW, p0 + CM_RED_CHUNK); p += blockDim.x) {
    const float ip = im[p], in = im[HW + p];
    const float ap = im[2 * HW + p] / (ip + 1e-9f) / P;  // loss/flow.py:212-215
    const float an = im[3 * HW + p] / (in + 1e-9f) / P;
    sq += ap * ap + an * an;
    nz += (ip + in > 0.f) ? 1.f : 0.f;
  }
  sq = evf_block_sum(sq, red);
  nz = evf_block_sum(nz, red);
  if (threadIdx.x == 0) {
    evf_atomic_add(stats + sbd * 2, sq);
    evf_atomic_add(stats + sbd * 2 + 1, nz);
  }
}

// Charbonnier smoothness, loss/flow.py:183-190,261-294.  One thread per pixel
// of one (s, p, b) flow map; each thread owns the 4 spatial pairs anchored at
// its pixel and the temporal pair (p, p+1).
__device__ __forceinline__ float evf_charb(float fxa, float fya, float fxb, float fyb) {
  const float u = (fxa - fxb) + (fya - fyb);  // components summed BEFORE the square (q5)
  return sqrtf(u * u + 1e-6f);
}

#define SM_ROWS 8
__device__ __forceinline__ void cm_smooth_body(int bx, int map, int gx, float* red, const float* __restrict__ flow,
                                               const float* __restrict__ mask, int Pm, int Pk, int B, int H, int W, int use_mask,
                                               int with_dt, float* __restrict__ part) {
  // bx = row-chunk within a map (of gx), map = index over (s, p, b): (s*Pm + p)*B + b
  const int b = map % B, p = (map / B) % Pm;
  const long HW = (long)H * W;
  const float* fx = flow + (long)map * 2 * HW;
  const float* fy = fx + HW;
  const float* fxn = fx + (long)B * 2 * HW;  // next pass, same (s, b)
  const float* fyn = fxn + HW;
  const float* m = use_mask ? mask + ((long)b * Pk + (Pk == 1 ? 0 : p)) * HW : nullptr;
  const float* mn = (use_mask && Pk > 1) ? m + HW : m;
  float acc = 0.f;
  const int y0 = bx * SM_ROWS;
  // Branch-free: every neighbour is loaded from a clamped index (the pixel itself when the pair does not exist) and the
  // pair's term is selected afterwards -- a load under `if (xr)` ends its basic block with s_waitcnt vmcnt(0).
  const float* mp = m ? m : fx;    // dummy source without a mask
  const float* mnp = m ? mn : fx;
  for (int idx = threadIdx.x; idx < SM_ROWS * W; idx += blockDim.x) {
    const int y = y0 + idx / W, x = idx % W;
    if (y >= H) break;
    const long c = (long)y * W + x;
    const bool xr = x + 1 < W, yd = y + 1 < H, dg = xr && yd, dt = with_dt && p + 1 < Pm;
    const long cr = xr ? c + 1 : c, cd = yd ? c + W : c, cdr = dg ? c + W + 1 : c;
    const float ax = fx[c], ay = fy[c];
    const float rx = fx[cr], ry = fy[cr], dxv = fx[cd], dyv = fy[cd], ex = fx[cdr], ey = fy[cdr];
    // (the next pass's map does not exist behind the last pass: take the dummy from this map then)
    const float nx = (dt ? fxn : fx)[c], ny = (dt ? fyn : fy)[c];
    const float mc = mp[c], mr = mp[cr], md = mp[cd], mdr = mp[cdr], mnx = (dt ? mnp : mp)[c];
    {
      const float v = evf_charb(ax, ay, rx, ry);
      acc += xr ? (m ? (mc * mr) * v : v) : 0.f;
    }
    {
      const float v = evf_charb(ax, ay, dxv, dyv);
      acc += yd ? (m ? (mc * md) * v : v) : 0.f;
    }
    {
      const float v = evf_charb(ax, ay, ex, ey);  // [:-1,:-1] - [1:,1:]
      acc += dg ? (m ? (mc * mdr) * v : v) : 0.f;
      const float u = evf_charb(dxv, dyv, rx, ry);  // [1:,:-1] - [:-1,1:]
      acc += dg ? (m ? (md * mr) * u : u) : 0.f;
    }
    {
      const float v = evf_charb(ax, ay, nx, ny);
      acc += dt ? (m ? (mc * mnx) * v : v) : 0.f;
    }
  }
  acc = evf_block_sum(acc, red);
  if (threadIdx.x == 0) part[(long)map * gx + bx] = acc;
}
__global__ void k_cm_smooth(const float* __restrict__ flow, const float* __restrict__ mask, int Pm, int Pk, int B, int H,
                            int W, int use_mask, int with_dt, float* __restrict__ part) {
  __shared__ float red[16];
  cm_smooth_body(blockIdx.x, blockIdx.y, gridDim.x, red, flow, mask, Pm, Pk, B, H, W, use_mask, with_dt, part);
}

// The two loss passes that only read the flow maps -- the pre-warp of the events and the smoothness partials -- in ONE launch
// (block role by index; 256 threads either way); its first threads also clear `stats` and the ticket of k_cm_splat_lds, which
// the next launch accumulates into (no fill node).
__global__ __launch_bounds__(256) void k_cm_pre(const float* __restrict__ flow, const float4* __restrict__ ev,
                                                const float2* __restrict__ pol, const int32_t* __restrict__ ev_pass,
                                                const float* __restrict__ mask, int S, int Pm, int Pk, int P, int B, int M, int H,
                                                int W, float Sc, int use_mask, int with_dt, float4* __restrict__ warp,
                                                float* __restrict__ tabs, float* __restrict__ ys, float* __restrict__ part,
                                                float* __restrict__ stats, unsigned* __restrict__ ticket, int nbw) {
  __shared__ float red[16];
  const int bid = blockIdx.x;
  if (bid < nbw * S) {
    if (bid == 0) {
      for (int i = threadIdx.x; i < S * B * 4; i += blockDim.x) stats[i] = 0.f;
      if (threadIdx.x == 0) ticket[0] = 0u;
    }
    cm_prewarp_body(bid % nbw, bid / nbw, flow, ev, pol, ev_pass, S, Pm, P, B, M, H, W, Sc, warp, tabs, ys);
    return;
  }
  const int sb = bid - nbw * S, gx = (H + SM_ROWS - 1) / SM_ROWS;
  cm_smooth_body(sb % gx, sb / gx, gx, red, flow, mask, Pm, Pk, B, H, W, use_mask, with_dt, part);
}

__device__ void cm_finalize_body(float* red, const float* stats, const float* __restrict__ part, int S, int B, int Pm,
                                 int nblk_per_scale, float weight, int comps, int loss_scaling, float* __restrict__ loss) {
  float total = 0.f;
  for (int s = 0; s < S; ++s) {
    float v = 0.f;
    for (int i = threadIdx.x; i < nblk_per_scale; i += blockDim.x) v += part[(long)s * nblk_per_scale + i];
    v = evf_block_sum(v, red);  // thread 0
    float data = 0.f;
    for (int i = threadIdx.x; i < B * 2; i += blockDim.x) {
      // (device-scope loads: inside k_cm_splat_lds the sums were made by other blocks' atomics of the same launch)
      const unsigned* st = (const unsigned*)(stats + ((long)s * B * 2 + i) * 2);
      const float s0 = __uint_as_float(__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      const float s1 = __uint_as_float(__hip_atomic_load(st + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      data += loss_scaling ? s0 / s1 : s0;
    }
    data = evf_block_sum(data, red);
    if (threadIdx.x == 0) total += data + weight * (v / (float)comps / (float)Pm);
  }
  if (threadIdx.x == 0) loss[0] = total / (float)S;
}
__global__ void k_cm_finalize(const float* __restrict__ stats, const float* __restrict__ part, int S, int B, int Pm,
                              int nblk_per_scale, float weight, int comps, int loss_scaling, float* __restrict__ loss) {
  __shared__ float red[16];
  cm_finalize_body(red, stats, part, S, B, Pm, nblk_per_scale, weight, comps, loss_scaling, loss);
}

extern "C" int evf_cm_smooth_blocks(int B, int P, int H, int W) {
  (void)W;
  return evf_cdiv(H, SM_ROWS) * P * B;
}

static int cm_args_ok(const void* flow, const void* ev, const void* pol, const void* ev_pass, const void* mask, int S,
                      int P, int B, int M, int H, int W, int flags) {
  if (!flow || !ev || !pol || !ev_pass || S <= 0 || P <= 0 || B <= 0 || M <= 0 || H <= 1 || W <= 1) return 0;
  if ((flags & 1) && !mask) return 0;
  return 1;
}

static bool cm_merge_on = true;  // evf_cm_merge: process-wide switch (the equivalence test; EVF_CM_MERGE=0 in the environment: off)
extern "C" int evf_cm_merge(int on) {
  cm_merge_on = on != 0;
  return EVF_OK;
}
// evf_cm_bwd_lds: how dL/dflow of the events is summed -- -1 by size (LDS stripes from 8192 events per map on), 0 always with
// device-scope atomics, 1 LDS stripes whenever a stripe fits (process-wide; tests, A/B; environment EVF_CM_BWD_LDS at load)
static int cm_bwd_lds_mode = []() {
  const char* e = getenv("EVF_CM_BWD_LDS");
  return e ? atoi(e) : -1;
}();
extern "C" int evf_cm_bwd_lds(int mode) {
  cm_bwd_lds_mode = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
  return EVF_OK;
}

extern "C" int evf_cm_loss_fwd(const float* flow, const float* ev, const float* pol, const int32_t* ev_pass,
                               const float* mask, int S, int P, int B, int M, int H, int W, float flow_scaling,
                               float regul_weight, int flags, float* images, float* stats, float* smooth_part,
                               float* loss, float* ws, void* stream) {
  if (!cm_args_ok(flow, ev, pol, ev_pass, mask, S, P, B, M, H, W, flags) || !images || !stats || !smooth_part || !loss)
    return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  const int overwrite = (flags & 2) ? 1 : 0;
  const int Pm = overwrite ? 1 : P, Pk = Pm;
  const int HW = H * W;
  const int srows = evf_cdiv(H, SM_ROWS);
  // EVF_CM_MERGE=0: one launch per pass (fill, pre-warp, splat, reduce, smooth, finalize) -- A/B measurements, the equivalence test
  static const bool merge = []() {
    const char* e = getenv("EVF_CM_MERGE");
    return !(e && e[0] == '0');
  }();
  int rc = EVF_OK;
  if (ws && evf_cm_loss_ws(S, B, M, H, W) > 0) {
    float4* warp = (float4*)ws;
    float* tabs = ws + (size_t)S * B * 2 * M * 4;
    unsigned* ticket = (unsigned*)(tabs + (size_t)B * M);
    float* ys = tabs + (size_t)B * M + 4;  // [S][B][2][M]: the warped rows alone
    const int rows = cm_lds_rows(S, B, H, W);
    const size_t lds = (size_t)4 * rows * W * sizeof(float);
    static size_t lds_set = 0;
    if (lds > lds_set) {
      (void)hipFuncSetAttribute((const void*)k_cm_splat_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      lds_set = lds;
    }
    if (merge && cm_merge_on) {
      // two launches: [pre-warp | smoothness partials | stats and ticket cleared], [striped splat + image statistics + the
      // last block finishes the loss]
      const int nbw = evf_cdiv((long)B * M, 256);
      hipLaunchKernelGGL(k_cm_pre, dim3(nbw * S + srows * S * Pm * B), dim3(256), 0, st, flow, (const float4*)ev, (const float2*)pol,
                         ev_pass, mask, S, Pm, Pk, P, B, M, H, W, flow_scaling, flags & 1, overwrite ? 0 : 1, warp, tabs, ys, smooth_part,
                         stats, ticket, nbw);
      const CmFin fin{stats, ticket, smooth_part, loss, S, Pm, srows * Pm * B, overwrite ? 4 : 5, (flags & 4) ? 1 : 0, regul_weight};
      hipLaunchKernelGGL(k_cm_splat_lds, dim3(evf_cdiv(H, rows), S * B * 2), dim3(1024), lds, st, (const float4*)warp, tabs, ys, B, M, H, W,
                         rows, (float)P, images, fin);
      return evf_status();
    }
    rc = evf_hip(evf_memset_async(stats, 0, sizeof(float) * (size_t)S * B * 4, st));
    if (rc) return rc;
    hipLaunchKernelGGL(k_cm_prewarp, dim3(evf_cdiv((long)B * M, 256), S), dim3(256), 0, st, flow, (const float4*)ev,
                       (const float2*)pol, ev_pass, S, Pm, P, B, M, H, W, flow_scaling, warp, tabs, ys);
    const CmFin none{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0.f};
    hipLaunchKernelGGL(k_cm_splat_lds, dim3(evf_cdiv(H, rows), S * B * 2), dim3(1024), lds, st, (const float4*)warp, tabs, ys, B,
                       M, H, W, rows, (float)P, images, none);
  } else {
    rc = evf_hip(evf_memset_async(stats, 0, sizeof(float) * (size_t)S * B * 4, st));
    if (rc) return rc;
    rc = evf_hip(evf_memset_async(images, 0, sizeof(float) * (size_t)S * B * 8 * HW, st));
    if (rc) return rc;
    hipLaunchKernelGGL(k_cm_splat, dim3(evf_cdiv((long)B * M, 256), S), dim3(256), 0, st, flow, (const float4*)ev,
                       (const float2*)pol, ev_pass, S, Pm, P, B, M, H, W, flow_scaling, images);
  }
  hipLaunchKernelGGL(k_cm_reduce, dim3(evf_cdiv(HW, CM_RED_CHUNK), S * B * 2), dim3(256), 0, st, images, HW, (float)P,
                     stats);
  hipLaunchKernelGGL(k_cm_smooth, dim3(srows, S * Pm * B), dim3(256), 0, st, flow, mask, Pm, Pk, B, H, W, flags & 1,
                     overwrite ? 0 : 1, smooth_part);
  hipLaunchKernelGGL(k_cm_finalize, dim3(1), dim3(256), 0, st, stats, smooth_part, S, B, Pm, srows * Pm * B, regul_weight,
                     overwrite ? 4 : 5, (flags & 4) ? 1 : 0, loss);
  return evf_status();
}

// --------------------------------------------------------------------------
// contrast-maximisation loss, backward
// --------------------------------------------------------------------------
// gimages = dL/d images (scaled by grad_out / S).  Scratch layout [S][B][2 directions][H*W][4]: the four planes of a direction
// (positive / negative event image, positive / negative timestamp image) INTERLEAVED per pixel, so that the event gather of
// k_cm_event_bwd takes one 16-byte load per bilinear tap instead of two dependent-looking 4-byte loads from planes 256 KiB
// apart (round 6: 254 -> see DESIGN 4.2 at 8 x 50 k events x 4 scales; the same sums per event).
__device__ __forceinline__ void cm_gimages_body(int bx, int sbd, int gx, const float* __restrict__ images,
                                                const float* __restrict__ stats, const float* __restrict__ grad_out, int S, int HW,
                                                float P, int loss_scaling, float* __restrict__ gim) {
  const long base = ((long)(sbd >> 1) * 8 + (sbd & 1) * 4) * HW;
  const float* im = images + base;
  float4* g = (float4*)(gim + base);
  const float sumsq = stats[sbd * 2], nnz = loss_scaling ? stats[sbd * 2 + 1] : 1.f;
  const float c = grad_out[0] / (float)S;
  for (int p = bx * blockDim.x + threadIdx.x; p < HW; p += gx * blockDim.x) {
    const float ip = im[p], in = im[HW + p];
    const float dp = ip + 1e-9f, dn = in + 1e-9f;
    const float ap = im[2 * HW + p] / dp / P, an = im[3 * HW + p] / dn / P;
    // d(#nonzero)/dI = 1 only where I_pos + I_neg is exactly 0 (masked assign, loss/flow.py:222-225)
    const float dnz = (loss_scaling && !(ip + in > 0.f)) ? -sumsq / (nnz * nnz) : 0.f;
    g[p] = make_float4(c * (-2.f * ap * ap / dp / nnz + dnz), c * (-2.f * an * an / dn / nnz + dnz), c * (2.f * ap / (dp * P) / nnz),
                       c * (2.f * an / (dn * P) / nnz));
  }
}
__global__ void k_cm_gimages(const float* __restrict__ images, const float* __restrict__ stats,
                             const float* __restrict__ grad_out, int S, int HW, float P, int loss_scaling,
                             float* __restrict__ gim) {
  cm_gimages_body(blockIdx.x, blockIdx.y, gridDim.x, images, stats, grad_out, S, HW, P, loss_scaling, gim);
}

// d max(0, 1-|d|) / d w with torch's sub-gradients: |.|'(0) = 0, and the
// max(0, v) tie v == 0 passes half the gradient.
__device__ __forceinline__ void evf_tent(float w, float c, float& val, float& dval) {
  const float d = w - c;
  const float v = 1.0f - fabsf(d);
  const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
  if (v > 0.f) {
    val = v;
    dval = -sg;
  } else if (v == 0.f) {
    val = 0.f;
    dval = -0.5f * sg;
  } else {
    val = 0.f;
    dval = 0.f;
  }
}

// g4: the direction's interleaved gradient image [H*W] of (pos, neg, pos ts, neg ts).  The four taps' loads are issued
// unconditionally (clamped address) and selected afterwards: eight 16-byte gathers in flight per event and scale.
__device__ __forceinline__ void evf_dir_grad(const Warp w, int H, int W, float a0, float a1, float tau,
                                             const float4* __restrict__ g4, float& gwy, float& gwx) {
  const float cy[2] = {floorf(w.wy), floorf(w.wy + 1.0f)};
  const float cx[2] = {floorf(w.wx), floorf(w.wx + 1.0f)};
  float ay[2], day[2], ax[2], dax[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    evf_tent(w.wy, cy[k], ay[k], day[k]);
    evf_tent(w.wx, cx[k], ax[k], dax[k]);
  }
  float4 G[2][2];
  bool in[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      in[j][i] = !(cy[j] < 0.f || cy[j] >= (float)H || cx[i] < 0.f || cx[i] >= (float)W);  // (false for NaN as well: mask = 0)
      const long px = in[j][i] ? (long)(cy[j] * (float)W + cx[i]) : 0;
      G[j][i] = g4[px];
    }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!in[j][i]) continue;
      // dL/d(weight of this tap)
      float gw = 0.f;
      if (a0 != 0.f) gw += a0 * (G[j][i].x + tau * G[j][i].z);
      if (a1 != 0.f) gw += a1 * (G[j][i].y + tau * G[j][i].w);
      gwy += gw * (day[j] * ax[i]);
      gwx += gw * (ay[j] * dax[i]);
    }
}

__global__ void k_cm_event_bwd(const float* __restrict__ flow, const float4* __restrict__ ev,
                               const float2* __restrict__ pol, const int32_t* __restrict__ ev_pass, int S, int Pm, int P,
                               int B, int M, int H, int W, float Sc, const float* __restrict__ gim,
                               float* __restrict__ dflow) {
  // Blocks are dealt round-robin to the eight XCDs (block n -> XCD n % 8, observed; gridDim.x is a multiple of 8): block x takes
  // chunk x % 8 of the event range, so that ONE XCD walks a contiguous eighth of the events -- one sample at batch 8 -- and its
  // 4 MiB L2 holds that sample's gradient image, flow and dL/dflow maps (3 MB at 256 x 256) instead of every sample's.
  const long vb = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long i = vb * blockDim.x + threadIdx.x;
  if (i >= (long)B * M) return;
  const int s = blockIdx.y;
  const int b = (int)(i / M), e = (int)(i - (long)b * M);
  const float4 q = ev[i];
  const int pass = ev_pass[e];
  const float t = q.x + (float)pass;
  const float2 pm = pol[i];
  const long HW = (long)H * W;
  const int map = Pm == 1 ? 0 : pass;
  float fy, fx;
  evf_event_flow(flow + (long)s * Pm * B * 2 * HW, map, B, b, HW, q.y, q.z, W, fy, fx);
  const float4* g = (const float4*)(gim + ((long)s * B + b) * 8 * HW);
  const float maxts = (float)P;
  float gfy = 0.f, gfx = 0.f;
  {
    float gwy = 0.f, gwx = 0.f;
    evf_dir_grad(evf_warp(t, q.y, q.z, fy, fx, maxts, Sc), H, W, pm.x, pm.y, t, g, gwy, gwx);
    const float k = (maxts - t) * Sc;  // d warped / d flow
    gfy += gwy * k;
    gfx += gwx * k;
  }
  {
    float gwy = 0.f, gwx = 0.f;
    evf_dir_grad(evf_warp(t, q.y, q.z, fy, fx, 0.f, Sc), H, W, pm.x, pm.y, maxts - t, g + HW, gwy, gwx);
    const float k = (0.f - t) * Sc;
    gfy += gwy * k;
    gfx += gwx * k;
  }
  const long lin = (long)(q.y * (float)W + q.z);
  float* d = dflow + (((long)s * Pm + map) * B + b) * 2 * HW;
  if (gfx != 0.f) evf_atomic_add(d + lin, gfx);
  if (gfy != 0.f) evf_atomic_add(d + HW + lin, gfy);
}

// ---- the same gather without global atomics (round 6) -------------------------------------------------------------------
// dL/dflow is added at the event's SOURCE pixel; with 8 x 50 k events x 4 scales k_cm_event_bwd issues 3.2 M device-scope
// float atomics, which resolve at the memory side at ~21 G/s: ~150 of its 254 us (config 4; 168 with the interleaved gradient
// image).  Here one block owns a stripe of rows of ONE dL/dflow map (scale, map, sample) in LDS: it passes over the sample's
// events in chunks of CMB_CHUNK, the waves COMPACT the events of the stripe (and map) into an LDS queue -- one in H / rows is
// the block's own, and the event body is four dependent memory round trips: it has to run on full waves --, the queue is worked
// off one event per thread (the arithmetic of k_cm_event_bwd, sums by LDS float atomics), and the stripe is finally ADDED to the
// smoothness gradient k_cm_bwd_pre stored there (every pixel has one owner: plain loads and stores).  Blocks of one sample share
// blockIdx.x % 8, i.e. (observed placement, speed only) an XCD and its L2.  Measured at 256 x 256 x 50 k x 4 scales x 8 samples
// (rocprofv3): 123 us with 512 blocks of 512 threads (131 with 256 x 1024 and chunks of 16 k events: chunk size and block shape
// hardly matter); the probe build -DCMB_PROBE_NOBODY (scan + queue alone) runs 48 us, i.e. ~75 us are the 12 divergent gathers
// per event (record, polarity, two flow components, eight 16-byte taps: 19 M lines through the CUs' texture addressers).
#define CMB_THREADS 512
#define CMB_CHUNK 4096  // events per pass over the queue: CMB_CHUNK / CMB_THREADS coordinate pairs in flight per thread
__global__ __launch_bounds__(CMB_THREADS) void k_cm_event_bwd_lds(const float* __restrict__ flow, const float4* __restrict__ ev,
                                                                   const float2* __restrict__ pol, const int32_t* __restrict__ ev_pass,
                                                                   int S, int Pm, int P, int B, int M, int H, int W, int rows, float Sc,
                                                                   const float* __restrict__ gim, float* __restrict__ dflow) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* acc = (float*)smem_raw;                       // [2][rows * W]: x component, y component
  int* queue = (int*)(acc + 2 * rows * W);             // [CMB_CHUNK] event numbers of the sample
  __shared__ int s_qn[2];  // queue length, one counter per chunk parity (reset two barriers before its next use)
  const int smb = blockIdx.x, b = smb % B, map = (smb / B) % Pm, s = smb / (B * Pm);
  const int r0 = blockIdx.y * rows, nr = min(rows, H - r0), plane = rows * W;
  const long HW = (long)H * W;
  for (int q = threadIdx.x; q < 2 * plane; q += CMB_THREADS) acc[q] = 0.f;
  if (threadIdx.x < 2) s_qn[threadIdx.x] = 0;
  const float4* __restrict__ evb = ev + (long)b * M;
  const float2* __restrict__ polb = pol + (long)b * M;
  const float* __restrict__ fl = flow + (((long)s * Pm + map) * B + b) * 2 * HW;
  const float4* __restrict__ g = (const float4*)(gim + ((long)s * B + b) * 8 * HW);
  const long lo = (long)r0 * W, hi = (long)(r0 + nr) * W;
  const float maxts = (float)P;
  constexpr int NL = CMB_CHUNK / CMB_THREADS;
  const int lane = threadIdx.x & 63;
  float qy[NL], qx[NL];
  int ps[NL];
  const float* __restrict__ evf = (const float*)evb;
  auto request = [&](int e0) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int e = min(e0 + k * CMB_THREADS + (int)threadIdx.x, M - 1);  // (clamped: the loads stay unconditional)
      qy[k] = evf[4 * (long)e + 1], qx[k] = evf[4 * (long)e + 2];
      ps[k] = 0;
      if (Pm != 1) ps[k] = ev_pass[e];  // (uniform branch)
    }
  };
  request(0);
  __syncthreads();
  for (int e0 = 0, par = 0; e0 < M; e0 += CMB_CHUNK, par ^= 1) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int e = e0 + k * CMB_THREADS + (int)threadIdx.x;
      const long lin = (long)(qy[k] * (float)W + qx[k]);  // flow_idx (loss/flow.py:65-67), as evf_event_flow
      const bool mine = e < M && ps[k] == map && lin >= lo && lin < hi;  // (NaN coordinates: lin is not in the stripe)
      const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
      int base = 0;
      if (lane == 0 && m) base = atomicAdd(&s_qn[par], __builtin_popcountll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      if (mine) queue[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = e;
    }
    if (e0 + CMB_CHUNK < M) request(e0 + CMB_CHUNK);  // (the next chunk's records travel while the queue is worked off)
    __syncthreads();
    const int qn = s_qn[par];
#ifdef CMB_PROBE_NOBODY  // (probe build: the scan and the queue alone)
    for (int j = threadIdx.x; j < qn; j += CMB_THREADS) acc[j & 1023] = (float)queue[j];
    if (0)
#endif
    for (int j = threadIdx.x; j < qn; j += CMB_THREADS) {
      const int e = queue[j];
      const float4 q = evb[e];
      const float2 pm = polb[e];
      const float t = q.x + (float)(Pm == 1 ? ev_pass[e] : map);  // event_list[:, :, 0:1] += passes (loss/flow.py:90)
      const long lin = (long)(q.y * (float)W + q.z);
      const float fx = fl[lin], fy = fl[HW + lin];
      float gfy = 0.f, gfx = 0.f;
      {
        float gwy = 0.f, gwx = 0.f;
        evf_dir_grad(evf_warp(t, q.y, q.z, fy, fx, maxts, Sc), H, W, pm.x, pm.y, t, g, gwy, gwx);
        const float k = (maxts - t) * Sc;  // d warped / d flow
        gfy += gwy * k;
        gfx += gwx * k;
      }
      {
        float gwy = 0.f, gwx = 0.f;
        evf_dir_grad(evf_warp(t, q.y, q.z, fy, fx, 0.f, Sc), H, W, pm.x, pm.y, maxts - t, g + HW, gwy, gwx);
        const float k = (0.f - t) * Sc;
        gfy += gwy * k;
        gfx += gwx * k;
      }
      const int o = (int)(lin - lo);
      if (gfx != 0.f) atomicAdd(acc + o, gfx);
      if (gfy != 0.f) atomicAdd(acc + plane + o, gfy);
    }
    __syncthreads();
    if (threadIdx.x == 0) s_qn[par] = 0;  // (everybody has read it; pushed to again two barriers from here)
  }
  float* d = dflow + (((long)s * Pm + map) * B + b) * 2 * HW + lo;
  const int n = nr * W;
  for (int q = threadIdx.x; q < 2 * n; q += CMB_THREADS) {
    const int c = q >= n, r = q - c * n;
    d[(long)c * HW + r] += acc[c * plane + r];
  }
}

__device__ __forceinline__ float evf_dcharb(float fxa, float fya, float fxb, float fyb) {
  const float u = (fxa - fxb) + (fya - fyb);
  return u / sqrtf(u * u + 1e-6f);
}

// writes dflow = d(weight * smoothness)/dflow (gather form, no atomics)
__device__ __forceinline__ void cm_smooth_bwd_body(int bx, int map, int gx, const float* __restrict__ flow,
                                                   const float* __restrict__ mask, int S, int Pm, int Pk, int B, int H, int W,
                                                   int use_mask, int with_dt, const float* __restrict__ grad_out, float scale,
                                                   float* __restrict__ dflow) {
  const int b = map % B, p = (map / B) % Pm;
  const long HW = (long)H * W;
  const float* fx = flow + (long)map * 2 * HW;
  const float* fy = fx + HW;
  const long pstride = (long)B * 2 * HW;
  const float* m = use_mask ? mask + ((long)b * Pk + (Pk == 1 ? 0 : p)) * HW : nullptr;
  const long mstride = (use_mask && Pk > 1) ? HW : 0;
  const float c = grad_out[0] * scale;
  const float* mp = m ? m : fx;  // dummy source without a mask
  for (int idx = bx * blockDim.x + threadIdx.x; idx < H * W; idx += gx * blockDim.x) {
    const int y = idx / W, x = idx % W;
    const long q = idx;
    const float ax = fx[q], ay = fy[q];
    const float mc = mp[q];
    float g = 0.f;
    // branch-free: the partner is loaded from q + dq when the pair exists and from q itself otherwise; the term is
    // selected afterwards (ten partners x three loads in flight instead of ten dependent round trips)
    auto pair = [&](long dq, bool ok, bool minuend) {
      const long qn = ok ? q + dq : q;
      const float bx = fx[qn], by = fy[qn], mm = mp[qn];
      const float w = m ? mc * mm : 1.f;
      const float v = minuend ? evf_dcharb(ax, ay, bx, by) : evf_dcharb(bx, by, ax, ay);
      const float t = ok ? w * v : 0.f;
      g = minuend ? g + t : g - t;
    };
    const bool xl = x > 0, xr = x + 1 < W, yu = y > 0, yd = y + 1 < H;
    pair(1, xr, true);              // dx anchored here
    pair(-1, xl, false);            // dx anchored at (y, x-1)
    pair(W, yd, true);              // dy
    pair(-W, yu, false);
    pair(W + 1, xr && yd, true);    // diag down-right
    pair(-W - 1, xl && yu, false);
    pair(-W + 1, xr && yu, true);   // up-right: a = (y, x) is the lower-left of the pair anchored at (y-1, x)
    pair(W - 1, xl && yd, false);   // up-right anchored at (y, x-1): a = (y+1, x-1), b = (y, x)
    {
      const bool nx = with_dt && p + 1 < Pm, pv = with_dt && p > 0;
      const float bx = fx[nx ? q + pstride : q], by = fy[nx ? q + pstride : q], mm = mp[nx ? q + mstride : q];
      const float t = nx ? (m ? mc * mm : 1.f) * evf_dcharb(ax, ay, bx, by) : 0.f;
      g += t;
      const float cx = fx[pv ? q - pstride : q], cy = fy[pv ? q - pstride : q], mq = mp[pv ? q - mstride : q];
      const float u = pv ? (m ? mc * mq : 1.f) * evf_dcharb(cx, cy, ax, ay) : 0.f;
      g -= u;
    }
    g *= c;
    float* d = dflow + (long)map * 2 * HW;
    d[q] = g;       // u depends on fx + fy symmetrically
    d[HW + q] = g;
  }
}
__global__ void k_cm_smooth_bwd(const float* __restrict__ flow, const float* __restrict__ mask, int S, int Pm, int Pk,
                                int B, int H, int W, int use_mask, int with_dt, const float* __restrict__ grad_out,
                                float scale, float* __restrict__ dflow) {
  cm_smooth_bwd_body(blockIdx.x, blockIdx.y, gridDim.x, flow, mask, S, Pm, Pk, B, H, W, use_mask, with_dt, grad_out, scale, dflow);
}
// The two backward passes in front of the event gather -- dL/dflow of the smoothness term (plain stores into dflow, which the
// gather then adds to) and dL/d(images) -- are independent of each other: one launch, block role by index.
__global__ __launch_bounds__(256) void k_cm_bwd_pre(const float* __restrict__ flow, const float* __restrict__ mask, int S, int Pm,
                                                    int Pk, int B, int H, int W, int use_mask, int with_dt,
                                                    const float* __restrict__ grad_out, float scale, float* __restrict__ dflow,
                                                    const float* __restrict__ images, const float* __restrict__ stats, float P,
                                                    int loss_scaling, float* __restrict__ gim) {
  const int gx = (H * W + 255) / 256, nsm = gx * S * Pm * B, bid = blockIdx.x;
  if (bid < nsm)
    cm_smooth_bwd_body(bid % gx, bid / gx, gx, flow, mask, S, Pm, Pk, B, H, W, use_mask, with_dt, grad_out, scale, dflow);
  else
    cm_gimages_body((bid - nsm) % gx, (bid - nsm) / gx, gx, images, stats, grad_out, S, H * W, P, loss_scaling, gim);
}

extern "C" int evf_cm_loss_bwd(const float* flow, const float* ev, const float* pol, const int32_t* ev_pass,
                               const float* mask, int S, int P, int B, int M, int H, int W, float flow_scaling,
                               float regul_weight, int flags, const float* images, const float* stats,
                               const float* grad_out, float* gimages, float* dflow, void* stream) {
  if (!cm_args_ok(flow, ev, pol, ev_pass, mask, S, P, B, M, H, W, flags) || !images || !stats || !grad_out ||
      !gimages || !dflow)
    return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  const int overwrite = (flags & 2) ? 1 : 0;
  const int Pm = overwrite ? 1 : P, Pk = Pm;
  const int HW = H * W;
  const int comps = overwrite ? 4 : 5;
  static const bool merge = []() {
    const char* e = getenv("EVF_CM_MERGE");
    return !(e && e[0] == '0');
  }();
  if (merge && cm_merge_on) {
    const int gx = evf_cdiv(HW, 256);
    hipLaunchKernelGGL(k_cm_bwd_pre, dim3(gx * S * Pm * B + gx * S * B * 2), dim3(256), 0, st, flow, mask, S, Pm, Pk, B, H, W, flags & 1,
                       overwrite ? 0 : 1, grad_out, regul_weight / (float)comps / (float)Pm / (float)S, dflow, images, stats, (float)P,
                       (flags & 4) ? 1 : 0, gimages);
  } else {
    hipLaunchKernelGGL(k_cm_smooth_bwd, dim3(evf_cdiv(HW, 256), S * Pm * B), dim3(256), 0, st, flow, mask, S, Pm, Pk, B, H,
                       W, flags & 1, overwrite ? 0 : 1, grad_out, regul_weight / (float)comps / (float)Pm / (float)S,
                       dflow);
    hipLaunchKernelGGL(k_cm_gimages, dim3(evf_cdiv(HW, 256), S * B * 2), dim3(256), 0, st, images, stats, grad_out, S, HW,
                       (float)P, (flags & 4) ? 1 : 0, gimages);
  }
  // many events per dL/dflow map: LDS stripes instead of device-scope atomics (EVF_CM_BWD_LDS=0: never, =1: whenever it fits)
  const int lds_mode = cm_bwd_lds_mode;
  int rows = W <= 8192 ? (H < 8192 / W ? H : 8192 / W) : 0;  // 2 components x rows x W floats <= 64 KiB of LDS
  while (rows > 8 && (long)S * Pm * B * evf_cdiv(H, rows) < 512) rows >>= 1;  // two blocks per CU at least
  const bool lds = rows > 0 && lds_mode != 0 && (lds_mode > 0 || (long)M >= 8192L * Pm);
  if (lds) {
    const size_t bytes = sizeof(float) * 2 * rows * W + sizeof(int) * CMB_CHUNK;
    static std::once_flag once;
    std::call_once(once, []() {
      (void)hipFuncSetAttribute((const void*)k_cm_event_bwd_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + (int)sizeof(int) * CMB_CHUNK);
    });
    hipLaunchKernelGGL(k_cm_event_bwd_lds, dim3(S * Pm * B, evf_cdiv(H, rows)), dim3(CMB_THREADS), bytes, st, flow, (const float4*)ev,
                       (const float2*)pol, ev_pass, S, Pm, P, B, M, H, W, rows, flow_scaling, gimages, dflow);
    return evf_status();
  }
  hipLaunchKernelGGL(k_cm_event_bwd, dim3(8 * evf_cdiv((long)B * M, 256 * 8), S), dim3(256), 0, st, flow, (const float4*)ev,
                     (const float2*)pol, ev_pass, S, Pm, P, B, M, H, W, flow_scaling, gimages, dflow);
  return evf_status();
}

// --------------------------------------------------------------------------
// metric reductions
// --------------------------------------------------------------------------
__global__ void k_image_variance(const float* __restrict__ img, int HW, float* __restrict__ out) {
  __shared__ float red[16];
  __shared__ float mean_s;
  const float* im = img + (long)blockIdx.x * HW;
  float s = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) s += im[p];
  s = evf_block_sum(s, red);
  if (threadIdx.x == 0) mean_s = s / (float)HW;
  __syncthreads();
  const float mean = mean_s;
  float v = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const float d = im[p] - mean;
    v += d * d;
  }
  v = evf_block_sum(v, red);
  if (threadIdx.x == 0) out[blockIdx.x] = v / (float)(HW - 1);  // unbiased, loss/flow.py:13-23
}

extern "C" int evf_image_variance(const float* img, int B, int HW, float* out, void* stream) {
  if (!img || !out || B <= 0 || HW <= 1) return EVF_EINVAL;
  hipLaunchKernelGGL(k_image_variance, dim3(B), dim3(1024), 0, EVF_STREAM(stream), img, HW, out);
  return evf_status();
}

__global__ void k_avg_ts_ratio(const float* __restrict__ images, int HW, float P, float* __restrict__ out) {
  __shared__ float red[16];
  const float* im = images + (long)blockIdx.x * 4 * HW;
  float sq = 0.f, nz = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const float ip = im[p], in = im[HW + p];
    const float ap = im[2 * HW + p] / (ip + 1e-9f) / P, an = im[3 * HW + p] / (in + 1e-9f) / P;
    sq += ap * ap + an * an;
    nz += (ip + in > 0.f) ? 1.f : 0.f;
  }
  sq = evf_block_sum(sq, red);
  nz = evf_block_sum(nz, red);
  if (threadIdx.x == 0) out[blockIdx.x] = sq / nz;
}

extern "C" int evf_avg_ts_ratio(const float* images, int B, int HW, float P, float* out, void* stream) {
  if (!images || !out || B <= 0 || HW <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_avg_ts_ratio, dim3(B), dim3(1024), 0, EVF_STREAM(stream), images, HW, P, out);
  return evf_status();
}

__global__ void k_aee(const float* __restrict__ flow, const float* __restrict__ gt, const float* __restrict__ mask,
                      const float* __restrict__ ratio, int HW, float S, float* __restrict__ out) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const float* f = flow + (long)b * 2 * HW;
  const float* g = gt + (long)b * 2 * HW;
  const float* m = mask + (long)b * HW;
  const float r = ratio[b];
  float se = 0.f, nv = 0.f, no = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const float fx = f[p] * S * r, fy = f[HW + p] * S * r;  // loss/flow.py:597-598
    const float gx = g[p], gy = g[HW + p];
    const bool valid = (m[p] != 0.f) && !((gx == 0.f) && (gy == 0.f));  // :605-614
    const float err = valid ? sqrtf((fx - gx) * (fx - gx) + (fy - gy) * (fy - gy)) : 0.f;
    const float mag = valid ? sqrtf(fx * fx + fy * fy) : 0.f;
    se += err;
    nv += valid ? 1.f : 0.f;
    no += ((err > 3.0f) && (err > 0.05f * mag)) ? 1.f : 0.f;  // :625
  }
  se = evf_block_sum(se, red);
  nv = evf_block_sum(nv, red);
  no = evf_block_sum(no, red);
  if (threadIdx.x == 0) {
    out[b * 3] = se;
    out[b * 3 + 1] = nv;
    out[b * 3 + 2] = no;
  }
}

extern "C" int evf_aee(const float* flow, const float* gt, const float* mask, const float* ratio, int B, int H, int W,
                       float flow_scaling, float* out, void* stream) {
  if (!flow || !gt || !mask || !ratio || !out || B <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_aee, dim3(B), dim3(1024), 0, EVF_STREAM(stream), flow, gt, mask, ratio, H * W, flow_scaling, out);
  return evf_status();
}

extern "C" int evf_version(void) { return 100; }
extern "C" int evf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// --------------------------------------------------------------------------
// hot-pixel mask applied to the encodings of a batch (dataloader/h5.py:289-295): x[b][c][q] *= mask[b][q]
// --------------------------------------------------------------------------
__global__ void k_apply_pixel_mask(float* __restrict__ x, const float* __restrict__ m, long BC, int C, long HW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC * HW) return;
  const long bc = i / HW, q = i - bc * HW;
  x[i] *= m[(bc / C) * HW + q];
}
extern "C" int evf_apply_pixel_mask(float* x, const float* mask, int B, int C, int H, int W, void* stream) {
  if (!x || !mask || B <= 0 || C <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  const long HW = (long)H * W, BC = (long)B * C;
  hipLaunchKernelGGL(k_apply_pixel_mask, dim3(evf_cdiv(BC * HW, 256)), dim3(256), 0, EVF_STREAM(stream), x, mask, BC, C, HW);
  return evf_status();
}

// --------------------------------------------------------------------------
// window masks (loss/flow.py:149-150, 443-452)
// --------------------------------------------------------------------------
// out[b][q] = min(sum_p mask[b][p][q], 1)
__global__ void k_mask_union(const float* __restrict__ m, int B, int P, long HW, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  const long b = i / HW, q = i - b * HW;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += m[(b * P + p) * HW + q];
  out[i] = fminf(s, 1.0f);
}
extern "C" int evf_mask_union(const float* masks, int B, int P, int H, int W, float* out, void* stream) {
  if (!masks || !out || B <= 0 || P <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  const long HW = (long)H * W;
  hipLaunchKernelGGL(k_mask_union, dim3(evf_cdiv(B * HW, 256)), dim3(256), 0, EVF_STREAM(stream), masks, B, P, HW, out);
  return evf_status();
}
// out[b][c][q] = sum_p maps[b][p][c][q] * mask[b][p][q] / (sum_p mask[b][p][q] + 1e-9)
__global__ void k_masked_flow_mean(const float* __restrict__ maps, const float* __restrict__ m, int B, int P, long HW,
                                   float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  const long b = i / HW, q = i - b * HW;
  float sx = 0.f, sy = 0.f, sm = 0.f;
  for (int p = 0; p < P; ++p) {
    const float w = m[(b * P + p) * HW + q];
    sx += maps[((b * P + p) * 2 + 0) * HW + q] * w;
    sy += maps[((b * P + p) * 2 + 1) * HW + q] * w;
    sm += w;
  }
  out[(b * 2 + 0) * HW + q] = sx / (sm + 1e-9f);
  out[(b * 2 + 1) * HW + q] = sy / (sm + 1e-9f);
}
extern "C" int evf_masked_flow_mean(const float* maps, const float* masks, int B, int P, int H, int W, float* out,
                                    void* stream) {
  if (!maps || !masks || !out || B <= 0 || P <= 0 || H <= 0 || W <= 0) return EVF_EINVAL;
  const long HW = (long)H * W;
  hipLaunchKernelGGL(k_masked_flow_mean, dim3(evf_cdiv(B * HW, 256)), dim3(256), 0, EVF_STREAM(stream), maps, masks, B, P, HW,
                     out);
  return evf_status();
}

// --------------------------------------------------------------------------
// norm_input (models/model.py:247-252): x[x != 0] = (x[x != 0] - mean) / std over the NON-ZERO entries of the
// whole tensor, std unbiased (torch.std).  Two passes over x with double accumulators (no host sync, no
// boolean-index temporaries): (count, sum), then sum of squared deviations, then the element-wise update.
// --------------------------------------------------------------------------
__global__ void k_nz_sum(const float* __restrict__ x, long n, double* __restrict__ ws) {
  __shared__ double red[2][16];
  double s = 0.0, c = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (v != 0.f) s += (double)v, c += 1.0;
  }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64), c += __shfl_xor(c, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[0][wv] = s, red[1][wv] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tc = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) ts += red[0][w], tc += red[1][w];
    atomicAdd(ws, ts);
    atomicAdd(ws + 1, tc);
  }
}
__global__ void k_nz_dev(const float* __restrict__ x, long n, double* __restrict__ ws) {
  __shared__ double red[16];
  const double mean = ws[0] / ws[1];
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (v != 0.f) {
      const double d = (double)v - mean;
      s += d * d;
    }
  }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) ts += red[w];
    atomicAdd(ws + 2, ts);
  }
}
__global__ void k_nz_apply(const float* __restrict__ x, long n, const double* __restrict__ ws, float* __restrict__ out) {
  const float mean = (float)(ws[0] / ws[1]);
  const float sd = (float)sqrt(ws[2] / (ws[1] - 1.0));  // unbiased; one non-zero entry -> NaN like torch
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    out[i] = v != 0.f ? (v - mean) / sd : 0.f;
  }
}

extern "C" int evf_norm_nonzero(const float* x, int64_t n, float* out, double* ws3, void* stream) {
  if (!x || !out || !ws3 || n <= 0) return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  const int rc = evf_hip(evf_memset_async(ws3, 0, 3 * sizeof(double), st));
  if (rc) return rc;
  const int nb = (int)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024);
  hipLaunchKernelGGL(k_nz_sum, dim3(nb), dim3(256), 0, st, x, (long)n, ws3);
  hipLaunchKernelGGL(k_nz_dev, dim3(nb), dim3(256), 0, st, x, (long)n, ws3);
  hipLaunchKernelGGL(k_nz_apply, dim3(nb), dim3(256), 0, st, x, (long)n, ws3, out);
  return evf_status();
}
