// Batch / instance normalisation of NHWC fp32 activations (gfx950): the two per-element passes of
// nn.BatchNorm2d / nn.InstanceNorm2d(track_running_stats=True) as the reference's ANN layers use them
// (models/submodules.py:46-56, 122-132, 169-180, 273-301).  The per-channel arithmetic between the passes (mean, 1/sqrt(var +
// eps), running statistics, affine coefficients) is a few hundred floats and stays in torch (models/hip_ops.py: norm2d).
//
// Statistics group g = 0 .. G-1 covers npg consecutive pixels (batch norm: G = 1 over B*H*W pixels; instance norm: G = B over
// H*W pixels); tensors are [G*npg][C] with pixel stride ld.  HBM bound element-wise / reduction kernels.
#include "evf_common.h"

// out[g*C + c] += sum over the group's pixels of
//   mode 0:  x
//   mode 1:  (x - center)^2
//   mode 2:  y * (x - center) * scale          (y = upstream gradient: sum g * xhat)
// `out` must be zeroed by the caller.  Block = 256 threads = (256 / CT) pixel lanes x CT channel lanes, CT = min(C, 64)
// rounded to a power of two; a block walks `ppb` pixels of ONE group.
__global__ __launch_bounds__(256) void k_chan_reduce(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ center, const float* __restrict__ scale, int mode,
                                                     int G, long npg, int C, int ldx, int ldy, int ct, int ppb,
                                                     float* __restrict__ out) {
  __shared__ float s[256];
  const int tid = threadIdx.x, cl = tid % ct, pl = tid / ct, npl = 256 / ct;
  const long blocks_per_group = (npg + ppb - 1) / ppb;
  const int g = (int)(blockIdx.x / blocks_per_group);
  const long p0 = (blockIdx.x % blocks_per_group) * ppb, p1 = min(p0 + (long)ppb, npg);
  for (int c0 = 0; c0 < C; c0 += ct) {
    const int c = c0 + cl;
    float acc = 0.f;
    if (c < C) {
      const float ce = center ? center[g * C + c] : 0.f, sc = scale ? scale[g * C + c] : 1.f;
      for (long p = p0 + pl; p < p1; p += npl) {
        const float v = x[((long)g * npg + p) * ldx + c];
        if (mode == 0)
          acc += v;
        else if (mode == 1)
          acc += (v - ce) * (v - ce);
        else
          acc += y[((long)g * npg + p) * ldy + c] * ((v - ce) * sc);
      }
    }
    s[tid] = acc;
    __syncthreads();
    if (pl == 0 && c < C) {
      float t = 0.f;
      for (int q = 0; q < npl; ++q) t += s[q * ct + cl];
      evf_atomic_add(out + g * C + c, t);
    }
    __syncthreads();
  }
}

extern "C" int evf_chan_reduce(const float* x, int ldx, const float* y, int ldy, const float* center, const float* scale, int mode,
                               int G, int64_t npg, int C, float* out, void* stream) {
  if (!x || !out || G <= 0 || npg <= 0 || C <= 0 || ldx < C || mode < 0 || mode > 2 || (mode == 2 && (!y || ldy < C)))
    return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  int rc = evf_hip(evf_memset_async(out, 0, sizeof(float) * (size_t)G * C, st));
  if (rc) return rc;
  int ct = 1;
  while (ct < C && ct < 64) ct <<= 1;
  // enough blocks to fill the chip, at least 64 pixels per pixel lane of a block
  long bpg = evf_cdiv(2048, G);
  const long maxb = evf_cdiv(npg, (long)(256 / ct) * 8);
  if (bpg > maxb) bpg = maxb;
  if (bpg < 1) bpg = 1;
  const int ppb = (int)evf_cdiv(npg, bpg);
  bpg = evf_cdiv(npg, ppb);
  hipLaunchKernelGGL(k_chan_reduce, dim3((unsigned)(bpg * G)), dim3(256), 0, st, x, y, center, scale, mode, G, (long)npg, C, ldx,
                     ldy, ct, ppb, out);
  return evf_status();
}

// out = (g ? A[gc] * g : 0) + Bc[gc] * x + Cc[gc]       per (group, channel) coefficients
//   forward:  g = NULL, Bc = w * rstd, Cc = b - mean * w * rstd
//   backward: A = w * rstd, Bc = -rstd^2 * w * S2 / n, Cc = rstd^2 * w * S2 / n * mean - rstd * w * S1 / n   (eval: Bc = Cc = 0)
__global__ void k_chan_affine(const float* __restrict__ gin, const float* __restrict__ x, const float* __restrict__ A,
                              const float* __restrict__ Bc, const float* __restrict__ Cc, long npg, int C, int ldg, int ldx,
                              int ldo, long total, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long p = idx / C;
  const int gc = (int)(p / npg) * C + c;
  float v = Bc[gc] * x[p * ldx + c] + Cc[gc];
  if (gin) v += A[gc] * gin[p * ldg + c];
  out[p * ldo + c] = v;
}

extern "C" int evf_chan_affine(const float* g, int ldg, const float* x, int ldx, const float* A, const float* Bc, const float* Cc,
                               int G, int64_t npg, int C, float* out, int ldo, void* stream) {
  if (!x || !Bc || !Cc || !out || G <= 0 || npg <= 0 || C <= 0 || ldx < C || ldo < C || (g && (!A || ldg < C))) return EVF_EINVAL;
  const long total = (long)G * npg * C;
  hipLaunchKernelGGL(k_chan_affine, dim3(evf_cdiv(total, 256)), dim3(256), 0, EVF_STREAM(stream), g, x, A, Bc, Cc, (long)npg, C,
                     ldg, ldx, ldo, total, out);
  return evf_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight normalisation of a conv weight (norm = "weight" cells, spiking_submodules.py:87-88, :502-504: nn.utils.weight_norm,
// dim = 0):  w[o, :] = v[o, :] * g[o] / ||v[o, :]||, the norm over the n = Cin * k * k elements of an output channel.
// One block per output channel; the norms are kept for the backward:
//   d v[o, j] = (g / nrm) * gw[o, j] - (g * <gw[o], v[o]> / nrm^3) * v[o, j],    d g[o] = <gw[o], v[o]> / nrm
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wn_block_sum(float s, float* s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_red[wv] = s;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void k_weight_norm_fwd(const float* __restrict__ v, const float* __restrict__ g, int n,
                                                         float* __restrict__ w, float* __restrict__ nrm) {
  __shared__ float s_red[4];
  const long o = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float t = v[o * n + j];
    s += t * t;
  }
  const float nr = sqrtf(wn_block_sum(s, s_red));
  const float sc = g[o] / nr;
  for (int j = threadIdx.x; j < n; j += blockDim.x) w[o * n + j] = v[o * n + j] * sc;
  if (threadIdx.x == 0) nrm[o] = nr;
}

__global__ __launch_bounds__(256) void k_weight_norm_bwd(const float* __restrict__ gw, const float* __restrict__ v,
                                                         const float* __restrict__ g, const float* __restrict__ nrm, int n,
                                                         float* __restrict__ gv, float* __restrict__ gg) {
  __shared__ float s_red[4];
  const long o = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) s += gw[o * n + j] * v[o * n + j];
  const float dot = wn_block_sum(s, s_red);
  const float nr = nrm[o], a = g[o] / nr, b = g[o] * dot / (nr * nr * nr);
  for (int j = threadIdx.x; j < n; j += blockDim.x) gv[o * n + j] = a * gw[o * n + j] - b * v[o * n + j];
  if (threadIdx.x == 0) gg[o] = dot / nr;
}

extern "C" int evf_weight_norm_fwd(const float* v, const float* g, int Cout, int n, float* w, float* nrm, void* stream) {
  if (!v || !g || !w || !nrm || Cout <= 0 || n <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_weight_norm_fwd, dim3(Cout), dim3(256), 0, EVF_STREAM(stream), v, g, n, w, nrm);
  return evf_status();
}

extern "C" int evf_weight_norm_bwd(const float* gw, const float* v, const float* g, const float* nrm, int Cout, int n, float* gv,
                                   float* gg, void* stream) {
  if (!gw || !v || !g || !nrm || !gv || !gg || Cout <= 0 || n <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_weight_norm_bwd, dim3(Cout), dim3(256), 0, EVF_STREAM(stream), gw, v, g, nrm, n, gv, gg);
  return evf_status();
}
