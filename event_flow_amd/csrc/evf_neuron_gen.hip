// Element-wise kernels of the general (any channel count, NHWC fp32) path:
//   * neuron state update + surrogate-gradient backward for the four spiking
//     cells (reference models/spiking_submodules.py: LIF :96-126/:516-551,
//     PLIF :191-227/:618-657, ALIF :299-334/:730-768, XLIF :399-435/:836-875)
//   * pooled pre-synaptic activity of PLIF/XLIF (:212,:418,:642)
//   * bilinear x2 / nearest xf up-sampling (spiking_submodules.py:1011,
//     models/model.py:529-539), tanh / sigmoid / relu, ConvGRU gate algebra
//     (models/submodules.py:400-418)
// All HBM bound: one float4 (4 channels of one pixel) per thread, per-channel
// parameter gradients reduced per block in LDS, then one atomic per channel.
#include <stdlib.h>

#include "evf_common.h"

__device__ __forceinline__ float ng_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float ng_surrogate(int kind, float x, float width) {
  // models/spiking_util.py:38-43 (superspike), :55-65 (multi-gauss), :74-79 (triangle), :88-93 (arctan)
  switch (kind) {
    case EVF_SUPERSPIKE: {
      const float d = 1.0f + width * fabsf(x);
      return 1.0f / (d * d);
    }
    case EVF_TRIANGLE:
      return fmaxf(0.f, 1.0f - width * fabsf(x));
    case EVF_MULTIGAUSS: {
      const float s1 = width, s2 = 6.f * width;
      const float k = 0.3989422804014327f;  // 1/sqrt(2*pi)
      auto gs = [&](float v, float mu, float sg) { return expf(-((v - mu) * (v - mu)) / (2.f * sg * sg)) / sg * k; };
      return 1.15f * gs(x, 0.f, s1) - 0.15f * gs(x, width, s2) - 0.15f * gs(x, -width, s2);
    }
    default:
      return 1.0f / (1.0f + width * x * x);
  }
}

struct NgParams {
  const float* p[4];
  float* g[4];
};
// parameter slots per kind:
//   LIF : p0 leak      p1 thresh
//   PLIF: p0 leak_v    p1 thresh   p2 leak_pt  p3 add_pt
//   ALIF: p0 leak_v    p1 t0       p2 t1       p3 leak_t
//   XLIF: p0 leak_v    p1 t0       p2 t1       p3 leak_pt

// optional tensor: read a valid dummy instead of branching around the load (a branch ends the basic block with
// s_waitcnt vmcnt(0)), select zero afterwards
__device__ __forceinline__ float4 ng_ld(const float4* p, const float4* dummy, long e) {
  const float4 v = (p ? p : dummy)[e];
  return p ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int KIND, bool PREV, bool RES>
__global__ void k_neuron_fwd(const float4* __restrict__ cur, const float4* __restrict__ v_prev,
                             const float4* __restrict__ z_prev, const float4* __restrict__ aux_prev,
                             const float* __restrict__ P, const float4* __restrict__ residual, NgParams prm, long npix, int C,
                             int hard, float4* __restrict__ v_out, float4* __restrict__ z_out, float4* __restrict__ aux_out,
                             float4* __restrict__ out) {
  const int Q = C >> 2;
  const long total = npix * Q, stride = (long)gridDim.x * blockDim.x;
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int cq = (int)(e % Q);
  float lam[4], a1[4], a2[4], a3[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * cq + k;
    lam[k] = ng_sigmoid(prm.p[0][c]);
    a1[k] = fmaxf(prm.p[1][c], 0.01f);  // thresh / t0 .clamp_min(0.01)
    if (KIND == EVF_PLIF) a2[k] = evf_plif_sigmoid(prm.p[2][c]), a3[k] = evf_plif_sigmoid(prm.p[3][c]);
    if (KIND == EVF_ALIF || KIND == EVF_XLIF) a2[k] = fmaxf(prm.p[2][c], 0.f), a3[k] = ng_sigmoid(prm.p[3][c]);
  }
  for (; e < total; e += stride) {
    const long pix = e / Q;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);  // (absent operands: compile-time zeros, no dummy loads)
    const float4 c4 = cur[e], v4 = PREV ? ng_ld(v_prev, cur, e) : zero4, z4 = PREV ? ng_ld(z_prev, cur, e) : zero4,
                 x4 = (PREV && KIND != EVF_LIF) ? ng_ld(aux_prev, cur, e) : zero4;
    const float Pv = (KIND == EVF_PLIF || KIND == EVF_XLIF) ? P[pix] : 0.f;
    const float cu[4] = {c4.x, c4.y, c4.z, c4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w}, z[4] = {z4.x, z4.y, z4.z, z4.w};
    const float ax[4] = {x4.x, x4.y, x4.z, x4.w};
    float vo[4], zo[4], ao[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float th = a1[k], c = cu[k], soft_th = a1[k];
      if (KIND == EVF_PLIF) {
        ao[k] = evf_plif_trace(ax[k], a2[k], Pv);  // pt' (:212)
        c = c - a3[k] * ao[k];                         // ff [+ rec] - add_pt * pt' (:220)
      } else if (KIND == EVF_ALIF) {
        ao[k] = ax[k] * a3[k] + (1.0f - a3[k]) * z[k];  // t' (:317)
        th = a1[k] + a2[k] * ao[k];                      // t0 + t1 * t' (:319)
        soft_th = a1[k] + a2[k] * ax[k];                 // soft reset uses the OLD trace (:329)
      } else if (KIND == EVF_XLIF) {
        ao[k] = evf_plif_trace(ax[k], a3[k], Pv);  // pt' (:418)
        th = a1[k] + a2[k] * ao[k];
        soft_th = a1[k] + a2[k] * ax[k];
      }
      if (hard)
        vo[k] = v[k] * lam[k] * (1.0f - z[k]) + (1.0f - lam[k]) * c;
      else
        vo[k] = v[k] * lam[k] + (1.0f - lam[k]) * c - z[k] * soft_th;
      zo[k] = (vo[k] - th) > 0.f ? 1.0f : 0.f;
    }
    v_out[e] = make_float4(vo[0], vo[1], vo[2], vo[3]);
    z_out[e] = make_float4(zo[0], zo[1], zo[2], zo[3]);
    if (KIND != EVF_LIF) aux_out[e] = make_float4(ao[0], ao[1], ao[2], ao[3]);
    if (out) {
      const float4 r4 = RES ? residual[e] : zero4;
      out[e] = make_float4(zo[0] + r4.x, zo[1] + r4.y, zo[2] + r4.z, zo[3] + r4.w);
    }
  }
}

// blockDim is a multiple of Q = C/4 so that a thread keeps its channel quad over the grid-stride loop
static int ng_block(int Q) { return Q >= 256 ? 256 : (256 / Q) * Q; }

// LIF update on a current that arrives IN PARTS: cur = sum_z a[z] (+ sum_z b[z]) -- the K-split partial sums of the feed-forward
// conv (and of the recurrent conv) as the conv kernels left them in scratch, added here in index order instead of by a
// k_b3_reduce launch per conv that writes `cur` only for this kernel to read it back (LIF-EV-FlowNet step: 13 + 7 launches).
// na / nb = 1 with the plain tensor when a conv did not split; b null = no recurrent part.
template <bool PREV, bool RES>
__global__ void k_lif_fwd_parts(const float4* __restrict__ a, int na, long sa, const float4* __restrict__ b, int nb, long sb,
                                const float4* __restrict__ v_prev, const float4* __restrict__ z_prev,
                                const float4* __restrict__ residual, const float* __restrict__ leak, const float* __restrict__ thresh,
                                long npix, int C, int hard, float4* __restrict__ v_out, float4* __restrict__ z_out,
                                float4* __restrict__ out) {
  const int Q = C >> 2;
  const long total = npix * Q, stride = (long)gridDim.x * blockDim.x;
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int cq = (int)(e % Q);
  float lam[4], th[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) lam[k] = ng_sigmoid(leak[4 * cq + k]), th[k] = fmaxf(thresh[4 * cq + k], 0.01f);
  for (; e < total; e += stride) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v4 = PREV ? ng_ld(v_prev, a, e) : zero4, z4 = PREV ? ng_ld(z_prev, a, e) : zero4, r4 = RES ? residual[e] : zero4;
    float4 c4 = a[e];
#pragma unroll 4
    for (int z = 1; z < na; ++z) {
      const float4 t = a[(long)z * sa + e];
      c4.x += t.x, c4.y += t.y, c4.z += t.z, c4.w += t.w;
    }
    if (b) {  // (uniform)
      float4 d4 = b[e];
#pragma unroll 4
      for (int z = 1; z < nb; ++z) {
        const float4 t = b[(long)z * sb + e];
        d4.x += t.x, d4.y += t.y, d4.z += t.z, d4.w += t.w;
      }
      c4.x = d4.x + c4.x, c4.y = d4.y + c4.y, c4.z = d4.z + c4.z, c4.w = d4.w + c4.w;  // (rec sum) + ff, as the accumulating conv adds
    }
    const float cu[4] = {c4.x, c4.y, c4.z, c4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w}, z[4] = {z4.x, z4.y, z4.z, z4.w};
    float vo[4], zo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (hard)
        vo[k] = v[k] * lam[k] * (1.0f - z[k]) + (1.0f - lam[k]) * cu[k];
      else
        vo[k] = v[k] * lam[k] + (1.0f - lam[k]) * cu[k] - z[k] * th[k];
      zo[k] = (vo[k] - th[k]) > 0.f ? 1.0f : 0.f;
    }
    v_out[e] = make_float4(vo[0], vo[1], vo[2], vo[3]);
    z_out[e] = make_float4(zo[0], zo[1], zo[2], zo[3]);
    if (out) out[e] = make_float4(zo[0] + r4.x, zo[1] + r4.y, zo[2] + r4.z, zo[3] + r4.w);
  }
}

extern "C" int evf_lif_fwd_parts(const float* a, int na, int64_t a_stride, const float* b, int nb, int64_t b_stride,
                                 const float* v_prev, const float* z_prev, const float* residual, const float* leak,
                                 const float* thresh, int64_t npix, int C, int hard_reset, float* v_out, float* z_out, float* out,
                                 void* stream) {
  if (!a || na < 1 || (b && nb < 1) || !leak || !thresh || !v_out || !z_out || npix <= 0 || C <= 0 || (C & 3) || C > 1024 ||
      (a_stride & 3) || (b_stride & 3) || (((uintptr_t)a | (uintptr_t)b) & 15))
    return EVF_EINVAL;
  const int Q = C >> 2, bs = ng_block(Q);
  const long total = npix * Q;
  const int nblk = (int)((total + bs - 1) / bs < 4096 ? (total + bs - 1) / bs : 4096);
  const bool prev = v_prev || z_prev, res = residual != nullptr;
#define LP_GO(P_, R_)                                                                                                          \
  hipLaunchKernelGGL((k_lif_fwd_parts<P_, R_>), dim3(nblk), dim3(bs), 0, EVF_STREAM(stream), (const float4*)a, na, (long)(a_stride >> 2), \
                     (const float4*)b, nb, (long)(b_stride >> 2), (const float4*)v_prev, (const float4*)z_prev,                  \
                     (const float4*)residual, leak, thresh, (long)npix, C, hard_reset, (float4*)v_out, (float4*)z_out, (float4*)out)
  if (prev && res) LP_GO(true, true);
  else if (prev) LP_GO(true, false);
  else if (res) LP_GO(false, true);
  else LP_GO(false, false);
#undef LP_GO
  return evf_status();
}

extern "C" int evf_neuron_fwd(int kind, const float* cur, const float* v_prev, const float* z_prev, const float* aux_prev,
                              const float* P, const float* residual, const float* p0, const float* p1, const float* p2,
                              const float* p3, int64_t npix, int C, int hard_reset, float* v_out, float* z_out,
                              float* aux_out, float* out, void* stream) {
  if (!cur || !p0 || !p1 || !v_out || !z_out || npix <= 0 || C <= 0 || (C & 3) || C > 1024 || kind < 0 || kind > 3)
    return EVF_EINVAL;
  if (kind != EVF_LIF && (!p2 || !p3 || !aux_out)) return EVF_EINVAL;
  if ((kind == EVF_PLIF || kind == EVF_XLIF) && !P) return EVF_EINVAL;
  NgParams prm = {{p0, p1, p2, p3}, {nullptr, nullptr, nullptr, nullptr}};
  const int Q = C >> 2, bs = ng_block(Q);
  const long total = npix * Q;
  const int nblk = (int)((total + bs - 1) / bs < 4096 ? (total + bs - 1) / bs : 4096);
  const bool prev = v_prev || z_prev || aux_prev, res = residual != nullptr;
#define NG_FWD_(K, P_, R_)                                                                                              \
  hipLaunchKernelGGL((k_neuron_fwd<K, P_, R_>), dim3(nblk), dim3(bs), 0, EVF_STREAM(stream), (const float4*)cur,        \
                     (const float4*)v_prev, (const float4*)z_prev, (const float4*)aux_prev, P, (const float4*)residual, \
                     prm, (long)npix, C, hard_reset, (float4*)v_out, (float4*)z_out, (float4*)aux_out, (float4*)out)
#define NG_FWD(K)                                 \
  do {                                            \
    if (prev && res) NG_FWD_(K, true, true);      \
    else if (prev) NG_FWD_(K, true, false);       \
    else if (res) NG_FWD_(K, false, true);        \
    else NG_FWD_(K, false, false);                \
  } while (0)
  switch (kind) {
    case EVF_LIF: NG_FWD(EVF_LIF); break;
    case EVF_PLIF: NG_FWD(EVF_PLIF); break;
    case EVF_ALIF: NG_FWD(EVF_ALIF); break;
    default: NG_FWD(EVF_XLIF); break;
  }
#undef NG_FWD
#undef NG_FWD_
  return evf_status();
}

// backward.  Saved: v_out, aux_out, v_prev, z_prev, aux_prev, P.  Upstream: g_v_out (state carry),
// g_z_out + g_z_out2 (through the layer output and through the z entry of the state), g_aux_out (trace carry).
// g_P [npix] = d loss / d pooled activity (PLIF/XLIF), g_z_prev only for ALIF (threshold trace reads z).
// GST: a state gradient arrives from the next pass (g_v_out / g_z_out2 / g_aux_out); PREV: there is a previous state.
// Absent groups are compile-time zeros: no dummy loads (each costs a texture-addresser slot), no wasted stores.
#define NG_REP 32
template <int KIND, bool GST, bool PREV>
__global__ void k_neuron_bwd(const float4* __restrict__ g_v_out, const float4* __restrict__ g_z_out,
                             const float4* __restrict__ g_z_out2, const float4* __restrict__ g_aux_out, const float4* __restrict__ v_out,
                             const float4* __restrict__ aux_out, const float4* __restrict__ v_prev,
                             const float4* __restrict__ z_prev, const float4* __restrict__ aux_prev,
                             const float* __restrict__ P, NgParams prm, long npix, int C, int hard, int surrogate,
                             float width, float4* __restrict__ g_cur, float4* __restrict__ g_v_prev,
                             float4* __restrict__ g_z_prev, float4* __restrict__ g_aux_prev, float* __restrict__ g_P,
                             float* __restrict__ ws) {
  extern __shared__ float s_acc[];  // [4][C]
  const int Q = C >> 2;
  const long total = npix * Q, stride = (long)gridDim.x * blockDim.x;
  for (int i = threadIdx.x; i < 4 * C; i += blockDim.x) s_acc[i] = 0.f;
  __syncthreads();
  const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cq = (int)(gtid % Q);
  float lam[4], a1[4], a2[4], a3[4], m1[4], m2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * cq + k;
    lam[k] = ng_sigmoid(prm.p[0][c]);
    a1[k] = fmaxf(prm.p[1][c], 0.01f);
    m1[k] = prm.p[1][c] >= 0.01f ? 1.f : 0.f;  // clamp_min passes the gradient where p >= min
    a2[k] = a3[k] = m2[k] = 0.f;
    if (KIND == EVF_PLIF) a2[k] = evf_plif_sigmoid(prm.p[2][c]), a3[k] = evf_plif_sigmoid(prm.p[3][c]);
    if (KIND == EVF_ALIF || KIND == EVF_XLIF) {
      a2[k] = fmaxf(prm.p[2][c], 0.f), a3[k] = ng_sigmoid(prm.p[3][c]);
      m2[k] = prm.p[2][c] >= 0.f ? 1.f : 0.f;
    }
  }
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};
  const int tpp = Q < 64 ? Q : 64;  // threads of one pixel inside a wave (Q is then a power of two <= 64) -- see host check
  for (long base = 0; base < total; base += stride) {  // uniform trip count: the pixel reduction shuffles
    const long e = base + gtid;
    const bool ok = e < total;
    const long ec = ok ? e : total - 1, pix = ec / Q;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 gv4 = GST ? ng_ld(g_v_out, v_out, ec) : zero4, gza = ng_ld(g_z_out, v_out, ec),
                 gzb = GST ? ng_ld(g_z_out2, v_out, ec) : zero4, ga4 = (GST && KIND != EVF_LIF) ? ng_ld(g_aux_out, v_out, ec) : zero4;
    const float4 gz4 = make_float4(gza.x + gzb.x, gza.y + gzb.y, gza.z + gzb.z, gza.w + gzb.w);
    const float4 vo4 = v_out[ec], v4 = PREV ? ng_ld(v_prev, v_out, ec) : zero4, z4 = PREV ? ng_ld(z_prev, v_out, ec) : zero4,
                 x4 = (PREV && KIND != EVF_LIF) ? ng_ld(aux_prev, v_out, ec) : zero4;
    const float4 ao4 = (KIND != EVF_LIF) ? aux_out[ec] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float Pv = (KIND == EVF_PLIF || KIND == EVF_XLIF) ? P[pix] : 0.f;
    const float gvo[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, gz[4] = {gz4.x, gz4.y, gz4.z, gz4.w};
    const float ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, vo[4] = {vo4.x, vo4.y, vo4.z, vo4.w};
    const float v[4] = {v4.x, v4.y, v4.z, v4.w}, z[4] = {z4.x, z4.y, z4.z, z4.w};
    const float ax[4] = {x4.x, x4.y, x4.z, x4.w}, ao[4] = {ao4.x, ao4.y, ao4.z, ao4.w};
    float gc[4], gp[4], gzp[4], gap[4], gPp = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float oml = 1.0f - lam[k];
      float th = a1[k], soft_th = a1[k];
      if (KIND == EVF_ALIF || KIND == EVF_XLIF) th = a1[k] + a2[k] * ao[k], soft_th = a1[k] + a2[k] * ax[k];
      const float sg = ng_surrogate(surrogate, vo[k] - th, width);
      const float gsp = gz[k] * sg;   // d z'/d(v' - th)
      const float G = gvo[k] + gsp;   // total gradient on v'
      const float gth = -gsp;         // on the threshold used by the spike function
      float cT, dlam, gsoft = 0.f;
      if (hard) {
        gp[k] = G * lam[k] * (1.0f - z[k]);  // z detached in the reset
        cT = (vo[k] - (v[k] * lam[k]) * (1.0f - z[k])) / oml;
        dlam = v[k] * (1.0f - z[k]) - cT;
      } else {
        gp[k] = G * lam[k];
        cT = (vo[k] - v[k] * lam[k] + z[k] * soft_th) / oml;
        dlam = v[k] - cT;
        gsoft = -z[k] * G;  // gradient on the soft-reset threshold
      }
      const float gcT = G * oml;
      gc[k] = gcT;
      gzp[k] = 0.f, gap[k] = 0.f;
      if (ok) s0[k] += G * dlam;
      if (KIND == EVF_LIF) {
        if (ok) s1[k] += gth + gsoft;
      } else if (KIND == EVF_PLIF) {
        const float gpt = ga[k] - a3[k] * gcT;  // on pt'
        gap[k] = gpt * a2[k];
        gPp += gpt * (1.0f - a2[k]);
        if (ok) {
          s1[k] += gth + gsoft;
          s2[k] += gpt * (ax[k] - Pv);  // d pt'/d sigma(leak_pt)
          s3[k] -= gcT * ao[k];          // d / d sigma(add_pt)
        }
      } else {  // ALIF / XLIF: th = t0 + t1 * trace'
        const float gtr = ga[k] + gth * a2[k];                 // on trace'
        gap[k] = gtr * a3[k] + gsoft * a2[k];                  // on the old trace (soft reset reads it)
        const float drive = (KIND == EVF_ALIF) ? z[k] : Pv;    // what the trace integrates
        if (KIND == EVF_ALIF) gzp[k] = gtr * (1.0f - a3[k]);
        else gPp += gtr * (1.0f - a3[k]);
        if (ok) {
          s1[k] += gth + gsoft;                       // t0
          s2[k] += gth * ao[k] + gsoft * ax[k];       // t1
          s3[k] += gtr * (ax[k] - drive);             // sigma(leak_t / leak_pt)
        }
      }
    }
    if (ok) {
      g_cur[e] = make_float4(gc[0], gc[1], gc[2], gc[3]);
      if (g_v_prev) {  // (null: the previous state takes no gradient -- uniform)
        g_v_prev[e] = make_float4(gp[0], gp[1], gp[2], gp[3]);
        if (KIND != EVF_LIF) g_aux_prev[e] = make_float4(gap[0], gap[1], gap[2], gap[3]);
        if (KIND == EVF_ALIF) g_z_prev[e] = make_float4(gzp[0], gzp[1], gzp[2], gzp[3]);
      }
    }
    if (KIND == EVF_PLIF || KIND == EVF_XLIF) {
      if (Q <= 64) {
        // the Q threads of a pixel are consecutive lanes of one wave
        for (int o = 1; o < tpp; o <<= 1) gPp += __shfl_xor(gPp, o, 64);
        if (ok && cq == 0) g_P[pix] = gPp;
      } else if (ok) {
        evf_atomic_add(g_P + pix, gPp);  // g_P zeroed by the host wrapper
      }
    }
  }
  // lanes l, l + Q, l + 2Q ... of a wave own the same channels (Q a power of two below 64): meet in registers first, so
  // that one lane per channel quad and wave issues the LDS float atomics (slow on gfx950)
  if (Q < 64) {
    for (int o = Q; o < 64; o <<= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s0[k] += __shfl_xor(s0[k], o, 64), s1[k] += __shfl_xor(s1[k], o, 64);
        if (KIND != EVF_LIF) s2[k] += __shfl_xor(s2[k], o, 64), s3[k] += __shfl_xor(s3[k], o, 64);
      }
    }
  }
  if (Q >= 64) {
    // blockDim / Q threads own a channel quad (threads t, t + Q, ...): they add in turns with plain LDS read-modify-writes --
    // ds_add_f32 is ~5x slower than an integer LDS atomic on gfx950, and 2048 of them per block were the fixed cost of this
    // kernel on the many-channel layers (14 us of a 22 us launch)
    const int turns = (int)blockDim.x / Q, mine = (int)threadIdx.x / Q;
    for (int t = 0; t < turns; ++t) {
      if (t == mine) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = 4 * cq + k;
          s_acc[0 * C + c] += s0[k] * lam[k] * (1.0f - lam[k]);
          s_acc[1 * C + c] += s1[k] * m1[k];
          if (KIND == EVF_PLIF) {
            s_acc[2 * C + c] += s2[k] * a2[k] * (1.0f - a2[k]);
            s_acc[3 * C + c] += s3[k] * a3[k] * (1.0f - a3[k]);
          } else if (KIND != EVF_LIF) {
            s_acc[2 * C + c] += s2[k] * m2[k];
            s_acc[3 * C + c] += s3[k] * a3[k] * (1.0f - a3[k]);
          }
        }
      }
      __syncthreads();
    }
  } else if ((int)(threadIdx.x & 63) < Q) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * cq + k;
    atomicAdd(&s_acc[0 * C + c], s0[k] * lam[k] * (1.0f - lam[k]));
    atomicAdd(&s_acc[1 * C + c], s1[k] * m1[k]);
    if (KIND == EVF_PLIF) {
      atomicAdd(&s_acc[2 * C + c], s2[k] * a2[k] * (1.0f - a2[k]));
      atomicAdd(&s_acc[3 * C + c], s3[k] * a3[k] * (1.0f - a3[k]));
    } else if (KIND != EVF_LIF) {
      atomicAdd(&s_acc[2 * C + c], s2[k] * m2[k]);
      atomicAdd(&s_acc[3 * C + c], s3[k] * a3[k] * (1.0f - a3[k]));
    }
  }
  }
  __syncthreads();
  const int np = KIND == EVF_LIF ? 2 : 4;
  if (!ws) {  // (no scratch: every block adds straight into the 2..4 x C outputs -- up to 1024 atomics per address)
    for (int i = threadIdx.x; i < np * C; i += blockDim.x) {
      const int p = i / C, c = i - p * C;
      if (prm.g[p]) evf_atomic_add(prm.g[p] + c, s_acc[i]);
    }
    return;
  }
  // Same-address atomics serialise at the memory side (1024 blocks on 128 words: +20..30 us per launch): the blocks are
  // spread over NG_REP replicas in scratch; the last block (below) sums the replicas into the outputs and hands the
  // scratch back zeroed.  ws: [NG_REP][4 * 1024] floats, zero on entry and on exit.
  float* mine = ws + (size_t)(blockIdx.x % NG_REP) * 4096;
  for (int i = threadIdx.x; i < np * C; i += blockDim.x) evf_atomic_add(mine + i, s_acc[i]);
  // ... the block that draws the LAST ticket (word NG_REP * 4096 of the scratch) does what k_ng_finish did as a launch of its own:
  // replicas -> outputs, scratch and ticket handed back zeroed (17 launches of ~4 us + their boundaries per EV-FlowNet step)
  __shared__ int s_fin;
  // The replica sums are device-scope ATOMICS (performed at the memory side, never cached): it is enough that every thread's
  // atomics have completed (workgroup-scope release = a wait, no cache maintenance) before thread 0 draws the ticket.  An
  // agent-scope release fence here writes the XCD's whole dirty L2 back -- the 134 MB of g_cur / g_v_prev this kernel has just
  // stored -- once per block: +45 us on the 256 x 256 layer when tried.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  int* ticket = (int*)(ws + (size_t)NG_REP * 4096);
  if (threadIdx.x == 0) {
    const int t = atomicAdd(ticket, 1);
    s_fin = t + 1 == (int)gridDim.x;
    if (s_fin) {
      *ticket = 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  if (!s_fin) return;
  const int nrep = (int)gridDim.x < NG_REP ? (int)gridDim.x : NG_REP;
  for (int i = threadIdx.x; i < np * C; i += blockDim.x) {
    float v[NG_REP];
#pragma unroll
    for (int r = 0; r < NG_REP; ++r) v[r] = __builtin_nontemporal_load(ws + (size_t)(r < nrep ? r : 0) * 4096 + i);
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < NG_REP; ++r) {
      if (r < nrep) {
        t += v[r];
        ws[(size_t)r * 4096 + i] = 0.f;
      }
    }
    const int p = i / C, c = i - p * C;
    if (prm.g[p]) prm.g[p][c] += t;
  }
}

extern "C" int evf_neuron_bwd(int kind, const float* g_v_out, const float* g_z_out, const float* g_z_out2,
                              const float* g_aux_out,
                              const float* v_out, const float* aux_out, const float* v_prev, const float* z_prev,
                              const float* aux_prev, const float* P, const float* p0, const float* p1, const float* p2,
                              const float* p3, int64_t npix, int C, int hard_reset, int surrogate, float act_width,
                              float* g_cur, float* g_v_prev, float* g_z_prev, float* g_aux_prev, float* g_P, float* g_p0,
                              float* g_p1, float* g_p2, float* g_p3, float* ws, void* stream) {
  if (!v_out || !p0 || !p1 || !g_cur || npix <= 0 || C <= 0 || (C & 3) || C > 1024 || kind < 0 || kind > 3)
    return EVF_EINVAL;
  const int Q = C >> 2;
  if (Q < 64 && (Q & (Q - 1))) return EVF_EINVAL;  // the in-wave pixel reduction needs a power of two
  if (kind != EVF_LIF && (!p2 || !p3 || !aux_out || (g_v_prev && !g_aux_prev))) return EVF_EINVAL;
  if ((kind == EVF_PLIF || kind == EVF_XLIF) && (!P || !g_P)) return EVF_EINVAL;
  if (kind == EVF_ALIF && g_v_prev && !g_z_prev) return EVF_EINVAL;
  hipStream_t st = EVF_STREAM(stream);
  if ((kind == EVF_PLIF || kind == EVF_XLIF) && Q > 64) {
    const int rc = evf_hip(evf_memset_async(g_P, 0, sizeof(float) * (size_t)npix, st));
    if (rc) return rc;
  }
  NgParams prm = {{p0, p1, p2, p3}, {g_p0, g_p1, g_p2, g_p3}};
  const int bs = ng_block(Q);
  const long total = npix * Q;
  // at least ~4 float4 per thread before a block pays its per-channel reduction (LDS float atomics + np * C global atomics): the
  // 512-channel 16 x 16 layers ran ONE float4 per thread in 1024 blocks -- 1 M global atomics for 1 M elements, 22 us per launch
  long want = (total + (long)bs * 4 - 1) / ((long)bs * 4);
  if (want < 64) want = (total + bs - 1) / bs < 64 ? (total + bs - 1) / bs : 64;
  const int nblk = (int)(want < 1024 ? want : 1024);
  // few blocks: straight into the outputs (<= 256 adds per address); many: NG_REP replicas + the last block's finish
  if (nblk <= 256) ws = nullptr;  // (replicas from 32 blocks on: LIF-EV-FlowNet step 6.56 against 6.53 ms)
  const size_t smem = sizeof(float) * 4 * (size_t)C;
  const bool gst = g_v_out || g_z_out2 || g_aux_out, prev = v_prev || z_prev || aux_prev;
#define NG_BWD_(K, G_, P_)                                                                                                 \
  hipLaunchKernelGGL((k_neuron_bwd<K, G_, P_>), dim3(nblk), dim3(bs), smem, st, (const float4*)g_v_out, (const float4*)g_z_out, \
                     (const float4*)g_z_out2, (const float4*)g_aux_out, (const float4*)v_out, (const float4*)aux_out, (const float4*)v_prev,        \
                     (const float4*)z_prev, (const float4*)aux_prev, P, prm, (long)npix, C, hard_reset, surrogate,         \
                     act_width, (float4*)g_cur, (float4*)g_v_prev, (float4*)g_z_prev, (float4*)g_aux_prev, g_P, ws)
#define NG_BWD(K)                                   \
  do {                                              \
    if (gst && prev) NG_BWD_(K, true, true);        \
    else if (gst) NG_BWD_(K, true, false);          \
    else if (prev) NG_BWD_(K, false, true);         \
    else NG_BWD_(K, false, false);                  \
  } while (0)
  switch (kind) {
    case EVF_LIF: NG_BWD(EVF_LIF); break;
    case EVF_PLIF: NG_BWD(EVF_PLIF); break;
    case EVF_ALIF: NG_BWD(EVF_ALIF); break;
    default: NG_BWD(EVF_XLIF); break;
  }
#undef NG_BWD
#undef NG_BWD_
  return evf_status();
}

// ---------------------------------------------------------------------------
// leaky (non-spiking) cells of the ANN comparisons -- models/submodules.py:502-554 (ConvLeaky) and :454-499
// (ConvLeakyRecurrent):  mix = prev * sigmoid(leak) + (1 - sigmoid(leak)) * (cur [+ residual]),  out = act(mix).
// ConvLeaky keeps `mix` as its state and returns act(mix); ConvLeakyRecurrent keeps tanh(mix) as state.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float ng_act(int kind, float v) {
  if (kind == 1) return tanhf(v);
  if (kind == 2) return 1.0f / (1.0f + expf(-v));
  if (kind == 3) return fmaxf(v, 0.f);
  return v;
}
__device__ __forceinline__ float ng_dact(int kind, float o) {  // through the output
  if (kind == 1) return 1.0f - o * o;
  if (kind == 2) return o * (1.0f - o);
  if (kind == 3) return o > 0.f ? 1.f : 0.f;
  return 1.f;
}

__global__ void k_leaky_fwd(const float4* __restrict__ cur, const float4* __restrict__ prev, const float4* __restrict__ residual,
                            const float* __restrict__ leak, int act, long npix, int C, float4* __restrict__ mix,
                            float4* __restrict__ out) {
  const int Q = C >> 2;
  const long total = npix * Q, stride = (long)gridDim.x * blockDim.x;
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int cq = (int)(e % Q);
  float lam[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) lam[k] = ng_sigmoid(leak[4 * cq + k]);
  for (; e < total; e += stride) {
    const float4 c4 = cur[e], p4 = ng_ld(prev, cur, e), r4 = ng_ld(residual, cur, e);
    const float c[4] = {c4.x + r4.x, c4.y + r4.y, c4.z + r4.z, c4.w + r4.w}, p[4] = {p4.x, p4.y, p4.z, p4.w};
    float m[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      m[k] = p[k] * lam[k] + (1.0f - lam[k]) * c[k];
      o[k] = ng_act(act, m[k]);
    }
    mix[e] = make_float4(m[0], m[1], m[2], m[3]);
    if (out) out[e] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int evf_leaky_fwd(const float* cur, const float* prev, const float* residual, const float* leak, int act,
                             int64_t npix, int C, float* mix, float* out, void* stream) {
  if (!cur || !leak || !mix || npix <= 0 || C <= 0 || (C & 3) || C > 1024 || act < 0 || act > 3) return EVF_EINVAL;
  const int Q = C >> 2, bs = ng_block(Q);
  const long total = npix * Q;
  const int nblk = (int)((total + bs - 1) / bs < 4096 ? (total + bs - 1) / bs : 4096);
  hipLaunchKernelGGL(k_leaky_fwd, dim3(nblk), dim3(bs), 0, EVF_STREAM(stream), (const float4*)cur, (const float4*)prev,
                     (const float4*)residual, leak, act, (long)npix, C, (float4*)mix, (float4*)out);
  return evf_status();
}

// g_mix = g_state + g_out * act'(act(mix));  g_cur (= g_residual) = g_mix (1 - lam);  g_prev = g_mix lam;
// g_leak[c] += sum g_mix (prev - cur) lam (1 - lam), with cur recovered from mix as (mix - prev lam) / (1 - lam)
__device__ float g_leaky_rep[NG_REP * 1024];  // zero at load, zero after every evf_leaky_bwd

__global__ void k_leaky_bwd(const float4* __restrict__ g_out, const float4* __restrict__ g_state, const float4* __restrict__ mix,
                            const float4* __restrict__ prev, const float* __restrict__ leak, int act, long npix, int C,
                            float4* __restrict__ g_cur, float4* __restrict__ g_prev, float* __restrict__ g_leak) {
  extern __shared__ float s_acc[];  // [C]
  const int Q = C >> 2;
  const long total = npix * Q, stride = (long)gridDim.x * blockDim.x;
  for (int i = threadIdx.x; i < C; i += blockDim.x) s_acc[i] = 0.f;
  __syncthreads();
  const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cq = (int)(gtid % Q);
  float lam[4], s0[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) lam[k] = ng_sigmoid(leak[4 * cq + k]);
  for (long e = gtid; e < total; e += stride) {
    const float4 go4 = ng_ld(g_out, mix, e), gs4 = ng_ld(g_state, mix, e), m4 = mix[e], p4 = ng_ld(prev, mix, e);
    const float go[4] = {go4.x, go4.y, go4.z, go4.w}, gs[4] = {gs4.x, gs4.y, gs4.z, gs4.w};
    const float m[4] = {m4.x, m4.y, m4.z, m4.w}, p[4] = {p4.x, p4.y, p4.z, p4.w};
    float gc[4], gp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float oml = 1.0f - lam[k];
      const float G = gs[k] + go[k] * ng_dact(act, ng_act(act, m[k]));
      const float c = (m[k] - p[k] * lam[k]) / oml;
      gc[k] = G * oml;
      gp[k] = G * lam[k];
      s0[k] += G * (p[k] - c);
    }
    g_cur[e] = make_float4(gc[0], gc[1], gc[2], gc[3]);
    if (g_prev) g_prev[e] = make_float4(gp[0], gp[1], gp[2], gp[3]);
  }
  if (g_leak) {
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(&s_acc[4 * cq + k], s0[k] * lam[k] * (1.0f - lam[k]));
    __syncthreads();
    // (1024 blocks adding into the same C words serialise at the memory side: 32 replicas, summed by k_leaky_finish)
    float* mine = g_leaky_rep + (blockIdx.x % NG_REP) * 1024;
    for (int i = threadIdx.x; i < C; i += blockDim.x) evf_atomic_add(mine + i, s_acc[i]);
  }
}

__global__ void k_leaky_finish(int nrep, int C, float* __restrict__ g_leak) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  float t = 0.f;
  for (int r = 0; r < nrep; ++r) {
    t += g_leaky_rep[r * 1024 + i];
    g_leaky_rep[r * 1024 + i] = 0.f;  // (handed back zeroed; calls are stream-ordered on one stream per device)
  }
  g_leak[i] += t;
}

extern "C" int evf_leaky_bwd(const float* g_out, const float* g_state, const float* mix, const float* prev, const float* leak,
                             int act, int64_t npix, int C, float* g_cur, float* g_prev, float* g_leak, void* stream) {
  if (!mix || !leak || !g_cur || (!g_out && !g_state) || npix <= 0 || C <= 0 || (C & 3) || C > 1024 || act < 0 || act > 3)
    return EVF_EINVAL;
  const int Q = C >> 2, bs = ng_block(Q);
  const long total = npix * Q;
  const int nblk = (int)((total + bs - 1) / bs < 1024 ? (total + bs - 1) / bs : 1024);
  hipLaunchKernelGGL(k_leaky_bwd, dim3(nblk), dim3(bs), sizeof(float) * (size_t)C, EVF_STREAM(stream), (const float4*)g_out,
                     (const float4*)g_state, (const float4*)mix, (const float4*)prev, leak, act, (long)npix, C,
                     (float4*)g_cur, (float4*)g_prev, g_leak);
  if (g_leak)
    hipLaunchKernelGGL(k_leaky_finish, dim3(evf_cdiv(C, 64)), dim3(64), 0, EVF_STREAM(stream), nblk < NG_REP ? nblk : NG_REP, C,
                       g_leak);
  return evf_status();
}

// ---------------------------------------------------------------------------
// pooled pre-synaptic activity: P = AvgPool_k(mean_c |x|), stride s, pad k/2,
// count_include_pad (F.avg_pool2d default).  spiking_submodules.py:212
// ---------------------------------------------------------------------------
__global__ void k_absmean(const float* __restrict__ x, long npix, int C, int ld, float* __restrict__ m) {
  // one wave per pixel group: tpp lanes share one pixel
  const int tpp = C >= 64 ? 64 : (C >= 32 ? 32 : (C >= 16 ? 16 : (C >= 8 ? 8 : (C >= 4 ? 4 : (C >= 2 ? 2 : 1)))));
  const int ppw = 64 / tpp, lane = threadIdx.x & 63, sub = lane % tpp;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long pix = wave * ppw + lane / tpp;
  const bool ok = pix < npix;
  const float* p = x + (ok ? pix : npix - 1) * ld;
  float s = 0.f;
  for (int c = sub; c < C; c += tpp) s += fabsf(p[c]);
  for (int o = 1; o < tpp; o <<= 1) s += __shfl_xor(s, o, 64);
  if (ok && sub == 0) m[pix] = s / (float)C;
}

__global__ void k_boxpool(const float* __restrict__ m, int B, int H, int W, int OH, int OW, int ksz, int stride,
                          float* __restrict__ P) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * OH * OW) return;
  const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH), b = (int)(idx / ((long)OW * OH)), pad = ksz >> 1;
  float s = 0.f;
  for (int dy = 0; dy < ksz; ++dy)
    for (int dx = 0; dx < ksz; ++dx) {
      const int yy = oy * stride + dy - pad, xx = ox * stride + dx - pad;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += m[((long)b * H + yy) * W + xx];
    }
  P[idx] = s / (float)(ksz * ksz);
}

extern "C" int evf_pretrace_fwd(const float* x, int ldx, int B, int H, int W, int C, int ksz, int stride, float* absmean_ws,
                                float* P, void* stream) {
  if (!x || !absmean_ws || !P || B <= 0 || H <= 0 || W <= 0 || C <= 0 || !EVF_KSZ_OK(ksz) || stride < 1)
    return EVF_EINVAL;
  const long npix = (long)B * H * W;
  const int tpp = C >= 64 ? 64 : (C >= 32 ? 32 : (C >= 16 ? 16 : (C >= 8 ? 8 : (C >= 4 ? 4 : (C >= 2 ? 2 : 1)))));
  const long waves = (npix + (64 / tpp) - 1) / (64 / tpp);
  hipLaunchKernelGGL(k_absmean, dim3(evf_cdiv(waves * 64, 256)), dim3(256), 0, EVF_STREAM(stream), x, npix, C, ldx,
                     absmean_ws);
  const int OH = (H + 2 * (ksz >> 1) - ksz) / stride + 1, OW = (W + 2 * (ksz >> 1) - ksz) / stride + 1;
  hipLaunchKernelGGL(k_boxpool, dim3(evf_cdiv((long)B * OH * OW, 256)), dim3(256), 0, EVF_STREAM(stream), absmean_ws, B, H,
                     W, OH, OW, ksz, stride, P);
  return evf_status();
}

// g_x[pix][c] (+)= sign(x) / C * sum_{windows covering pix} g_P[o] / k^2
__global__ void k_pretrace_bwd(const float* __restrict__ x, int ldx, const float* __restrict__ g_P, int B, int H, int W,
                               int C, int OH, int OW, int ksz, int stride, float* __restrict__ g_x, int ldg,
                               int accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long pix = idx / C;
  const int xx = (int)(pix % W), yy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H)), pad = ksz >> 1;
  float s = 0.f;
  for (int dy = 0; dy < ksz; ++dy)
    for (int dx = 0; dx < ksz; ++dx) {
      const int ty = yy + pad - dy, tx = xx + pad - dx;
      if (ty < 0 || tx < 0 || ty % stride || tx % stride) continue;
      const int oy = ty / stride, ox = tx / stride;
      if (oy < OH && ox < OW) s += g_P[((long)b * OH + oy) * OW + ox];
    }
  const float xv = x[pix * ldx + c];
  const float sgn = xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f);
  const float v = sgn * (s / (float)(ksz * ksz)) / (float)C;
  float* d = g_x + pix * ldg + c;
  *d = accumulate ? *d + v : v;
}

extern "C" int evf_pretrace_bwd(const float* x, int ldx, const float* g_P, int B, int H, int W, int C, int ksz, int stride,
                                float* g_x, int ldg, int accumulate, void* stream) {
  if (!x || !g_P || !g_x || B <= 0 || H <= 0 || W <= 0 || C <= 0 || !EVF_KSZ_OK(ksz) || stride < 1) return EVF_EINVAL;
  const int OH = (H + 2 * (ksz >> 1) - ksz) / stride + 1, OW = (W + 2 * (ksz >> 1) - ksz) / stride + 1;
  const long total = (long)B * H * W * C;
  hipLaunchKernelGGL(k_pretrace_bwd, dim3(evf_cdiv(total, 256)), dim3(256), 0, EVF_STREAM(stream), x, ldx, g_P, B, H, W, C,
                     OH, OW, ksz, stride, g_x, ldg, accumulate);
  return evf_status();
}

// ---------------------------------------------------------------------------
// bilinear x2 up-sampling, align_corners = False (F.interpolate default),
// NHWC.  spiking_submodules.py:1011 / submodules.py:149
// ---------------------------------------------------------------------------
__device__ __forceinline__ void up2_src(int o, int n, int& i0, int& i1, float& w1) {
  float s = 0.5f * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  w1 = s - (float)i0;
}

// one thread = V consecutive channels of one pixel (V = 4 / 2 / 1 by divisibility of C: the decoder inputs of the
// EV-FlowNet have 2C+2 channels); all taps are loaded unconditionally (border taps carry weight 0)
template <int V>
struct UpVec;
template <>
struct UpVec<4> { typedef float4 T; };
template <>
struct UpVec<2> { typedef float2 T; };
template <>
struct UpVec<1> { typedef float T; };
__device__ __forceinline__ float4 up_fma(float w, float4 a, float4 acc) { return make_float4(acc.x + w * a.x, acc.y + w * a.y, acc.z + w * a.z, acc.w + w * a.w); }
__device__ __forceinline__ float2 up_fma(float w, float2 a, float2 acc) { return make_float2(acc.x + w * a.x, acc.y + w * a.y); }
__device__ __forceinline__ float up_fma(float w, float a, float acc) { return acc + w * a; }
__device__ __forceinline__ float4 up_mix(float wa, float4 a, float wb, float4 b) { return make_float4(wa * a.x + wb * b.x, wa * a.y + wb * b.y, wa * a.z + wb * b.z, wa * a.w + wb * b.w); }
__device__ __forceinline__ float2 up_mix(float wa, float2 a, float wb, float2 b) { return make_float2(wa * a.x + wb * b.x, wa * a.y + wb * b.y); }
__device__ __forceinline__ float up_mix(float wa, float a, float wb, float b) { return wa * a + wb * b; }

template <int V>
__global__ void k_up2_fwd(const typename UpVec<V>::T* __restrict__ x, int B, int H, int W, int Q,
                          typename UpVec<V>::T* __restrict__ y) {
  typedef typename UpVec<V>::T T;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int OH = 2 * H, OW = 2 * W;
  if (idx >= (long)B * OH * OW * Q) return;
  const int q = (int)(idx % Q);
  const long pix = idx / Q;
  const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((long)OW * OH));
  int y0, y1, x0, x1;
  float wy, wx;
  up2_src(oy, H, y0, y1, wy);
  up2_src(ox, W, x0, x1, wx);
  const T a = x[(((long)b * H + y0) * W + x0) * Q + q], bq = x[(((long)b * H + y0) * W + x1) * Q + q];
  const T c = x[(((long)b * H + y1) * W + x0) * Q + q], d = x[(((long)b * H + y1) * W + x1) * Q + q];
  // same association as ATen's upsample_bilinear2d: h0 (w0 a + w1 b) + h1 (w0 c + w1 d)
  y[idx] = up_mix(1.f - wy, up_mix(1.f - wx, a, wx, bq), wy, up_mix(1.f - wx, c, wx, d));
}

// gather form of the transpose: input pixel iy receives from output rows 2iy-1 .. 2iy+2 (clamped addresses, the
// weight is 0 where the output pixel does not read this input pixel: no branch around the loads)
template <int V>
__global__ void k_up2_bwd(const typename UpVec<V>::T* __restrict__ gy, int B, int H, int W, int Q,
                          typename UpVec<V>::T* __restrict__ gx) {
  typedef typename UpVec<V>::T T;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W * Q) return;
  const int OH = 2 * H, OW = 2 * W;
  const int q = (int)(idx % Q);
  const long pix = idx / Q;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
  float cy[4], cx[4];
  int oyc[4], oxc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oy = 2 * iy - 1 + k, ox = 2 * ix - 1 + k;
    int a0, a1;
    float w;
    oyc[k] = min(max(oy, 0), OH - 1);
    up2_src(oyc[k], H, a0, a1, w);
    cy[k] = (oy >= 0 && oy < OH) ? ((a0 == iy ? 1.f - w : 0.f) + (a1 == iy ? w : 0.f)) : 0.f;
    oxc[k] = min(max(ox, 0), OW - 1);
    up2_src(oxc[k], W, a0, a1, w);
    cx[k] = (ox >= 0 && ox < OW) ? ((a0 == ix ? 1.f - w : 0.f) + (a1 == ix ? w : 0.f)) : 0.f;
  }
  T acc = T();
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = up_fma(cy[j] * cx[k], gy[(((long)b * OH + oyc[j]) * OW + oxc[k]) * Q + q], acc);
  gx[idx] = acc;
}

extern "C" int evf_upsample2x_fwd(const float* x, int B, int H, int W, int C, float* y, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EVF_EINVAL;
  const bool a16 = (((uintptr_t)x | (uintptr_t)y) & 15) == 0, a8 = (((uintptr_t)x | (uintptr_t)y) & 7) == 0;
  const int V = (C % 4 == 0 && a16) ? 4 : ((C % 2 == 0 && a8) ? 2 : 1);
  const long total = (long)B * 4 * H * W * (C / V);
  dim3 grid(evf_cdiv(total, 256)), block(256);
  hipStream_t st = EVF_STREAM(stream);
  if (V == 4)
    hipLaunchKernelGGL(k_up2_fwd<4>, grid, block, 0, st, (const float4*)x, B, H, W, C / 4, (float4*)y);
  else if (V == 2)
    hipLaunchKernelGGL(k_up2_fwd<2>, grid, block, 0, st, (const float2*)x, B, H, W, C / 2, (float2*)y);
  else
    hipLaunchKernelGGL(k_up2_fwd<1>, grid, block, 0, st, x, B, H, W, C, y);
  return evf_status();
}

extern "C" int evf_upsample2x_bwd(const float* g_y, int B, int H, int W, int C, float* g_x, void* stream) {
  if (!g_y || !g_x || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EVF_EINVAL;
  const bool a16 = (((uintptr_t)g_y | (uintptr_t)g_x) & 15) == 0, a8 = (((uintptr_t)g_y | (uintptr_t)g_x) & 7) == 0;
  const int V = (C % 4 == 0 && a16) ? 4 : ((C % 2 == 0 && a8) ? 2 : 1);
  const long total = (long)B * H * W * (C / V);
  dim3 grid(evf_cdiv(total, 256)), block(256);
  hipStream_t st = EVF_STREAM(stream);
  if (V == 4)
    hipLaunchKernelGGL(k_up2_bwd<4>, grid, block, 0, st, (const float4*)g_y, B, H, W, C / 4, (float4*)g_x);
  else if (V == 2)
    hipLaunchKernelGGL(k_up2_bwd<2>, grid, block, 0, st, (const float2*)g_y, B, H, W, C / 2, (float2*)g_x);
  else
    hipLaunchKernelGGL(k_up2_bwd<1>, grid, block, 0, st, g_y, B, H, W, C, g_x);
  return evf_status();
}

// channel concatenation of up to 6 NHWC activations (decoder input cat(prediction, x, skip[, zero padding]),
// models/unet.py:303-306; a null source = zeros): one pass, every output float written once
struct CatParts {
  const float* src[6];
  int C[6], ld[6], off[7];  // channels, pixel stride (floats), first output channel of each part
  int n;
};
__global__ void k_concat_channels(CatParts p, long npix, int Ctot, float* __restrict__ out, int ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npix * Ctot) return;
  const long pix = idx / Ctot;
  const int c = (int)(idx - pix * Ctot);
  int k = 0;
#pragma unroll
  for (int j = 1; j < 6; ++j) k += (j < p.n && c >= p.off[j]) ? 1 : 0;
  const float* s = p.src[k];
  out[pix * ldo + c] = s ? s[pix * p.ld[k] + (c - p.off[k])] : 0.f;
}
extern "C" int evf_concat_channels(const void* const* src, const int* C, const int* ld, int n, int64_t npix, float* out, int ldo,
                                   void* stream) {
  if (!src || !C || !ld || !out || n <= 0 || n > 6 || npix <= 0) return EVF_EINVAL;
  CatParts p;
  int tot = 0;
  for (int k = 0; k < 6; ++k) {
    p.src[k] = k < n ? (const float*)src[k] : nullptr;
    p.C[k] = k < n ? C[k] : 0;
    p.ld[k] = k < n ? ld[k] : 0;
    p.off[k] = tot;
    if (k < n && (C[k] <= 0 || (src[k] && ld[k] < C[k]))) return EVF_EINVAL;
    tot += p.C[k];
  }
  p.off[6] = tot, p.n = n;
  if (ldo < tot) return EVF_EINVAL;
  hipLaunchKernelGGL(k_concat_channels, dim3(evf_cdiv(npix * tot, 256L)), dim3(256), 0, EVF_STREAM(stream), p, (long)npix, tot, out,
                     ldo);
  return evf_status();
}

// The same concatenation READ THROUGH the bilinear x2 of a decoder (spiking_submodules.py:1011 on the cat of unet.py:303-306): the
// low-resolution concatenation is never written -- a thread blends the four source pixels of its two output channels straight from
// the part they live in (all part offsets of the decoder inputs are even: [2 flow | C | C | 2 zeros]).  Same association as
// k_up2_fwd / ATen's upsample_bilinear2d.
__device__ __forceinline__ float2 up_sel(bool c, float2 a, float2 b) { return c ? a : b; }
__global__ void k_concat_up2_fwd(CatParts p, int B, int H, int W, int Ctot, float* __restrict__ out, int ldo) {
  // One thread = FOUR output channels of the 2 x 2 output pixels of ONE source pixel (i, j): its 3 x 3 source neighbourhood is read
  // once (18 eight-byte loads for four 16-byte stores; a thread per output pixel issued 8 per store, and the output -- 4x the
  // input -- is what this kernel moves).  The four channels are two pairs that may come from different parts (the part offsets
  // are even, not multiples of four).  Rows: A = i0(2i), Bm = i1(2i), C = i1(2i + 1), and i0(2i + 1) is A for i = 0 and Bm
  // otherwise (up2_src); columns alike -- the SAME operands in the same expression as k_up2_fwd, bit for bit.
  const int Q = Ctot >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W * Q) return;
  const int c = 4 * (int)(idx % Q);
  const long pix = idx / Q;
  const int j = (int)(pix % W), i = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
  int ya, yb, yb0, yc, xa, xb, xb0, xc;
  float wy0, wy1, wx0, wx1;
  up2_src(2 * i, H, ya, yb, wy0);
  up2_src(2 * i + 1, H, yb0, yc, wy1);
  up2_src(2 * j, W, xa, xb, wx0);
  up2_src(2 * j + 1, W, xb0, xc, wx1);
  const bool ry = yb0 == ya, rx = xb0 == xa;  // (i == 0 / j == 0: the odd output's first tap is slot A, else slot B)
  const int ys[3] = {ya, yb, yc}, xs[3] = {xa, xb, xc};
  float2 r[2][2][2];  // [output row][output column][channel pair]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int ch = c + 2 * h;
    int k = 0;
#pragma unroll
    for (int jj = 1; jj < 6; ++jj) k += (jj < p.n && ch >= p.off[jj]) ? 1 : 0;
    const float* s = p.src[k];
    // (a null part = zero padding: a valid dummy address, selected away -- no load under a branch)
    const float* q = (s ? s : p.src[0]) + (s ? ch - p.off[k] : 0);
    const long ld = s ? p.ld[k] : p.ld[0];
    float2 v[3][3];
#pragma unroll
    for (int yy = 0; yy < 3; ++yy)
#pragma unroll
      for (int xx = 0; xx < 3; ++xx) v[yy][xx] = *(const float2*)(q + (((long)b * H + ys[yy]) * W + xs[xx]) * ld);
    // output (0, 0): rows A, Bm; columns A, Bm
    const float2 o00 = up_mix(1.f - wy0, up_mix(1.f - wx0, v[0][0], wx0, v[0][1]), wy0, up_mix(1.f - wx0, v[1][0], wx0, v[1][1]));
    // output (0, 1): columns (rx ? A : Bm), C
    const float2 t0 = up_sel(rx, v[0][0], v[0][1]), t1 = up_sel(rx, v[1][0], v[1][1]), t2 = up_sel(rx, v[2][0], v[2][1]);
    const float2 o01 = up_mix(1.f - wy0, up_mix(1.f - wx1, t0, wx1, v[0][2]), wy0, up_mix(1.f - wx1, t1, wx1, v[1][2]));
    // output (1, 0): rows (ry ? A : Bm), C
    const float2 u0 = up_sel(ry, v[0][0], v[1][0]), u1 = up_sel(ry, v[0][1], v[1][1]), u2 = up_sel(ry, v[0][2], v[1][2]);
    const float2 o10 = up_mix(1.f - wy1, up_mix(1.f - wx0, u0, wx0, u1), wy1, up_mix(1.f - wx0, v[2][0], wx0, v[2][1]));
    // output (1, 1)
    const float2 w0 = up_sel(rx, u0, u1);
    const float2 o11 = up_mix(1.f - wy1, up_mix(1.f - wx1, w0, wx1, u2), wy1, up_mix(1.f - wx1, t2, wx1, v[2][2]));
    const float2 z2 = make_float2(0.f, 0.f);
    r[0][0][h] = s ? o00 : z2, r[0][1][h] = s ? o01 : z2, r[1][0][h] = s ? o10 : z2, r[1][1][h] = s ? o11 : z2;
  }
  const int OW = 2 * W;
  float* o = out + (((long)b * 2 * H + 2 * i) * OW + 2 * j) * ldo + c;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
      *(float4*)(o + ((long)a * OW + bb) * ldo) = make_float4(r[a][bb][0].x, r[a][bb][0].y, r[a][bb][1].x, r[a][bb][1].y);
}
// parts: NHWC [B,H,W,C[k]] with pixel stride ld[k]; out [B,2H,2W,sum C] with pixel stride ldo.  Every C[k], ld[k] even, sum C and
// ldo multiples of four, part pointers 8-byte and `out` 16-byte aligned, the first part not null.
extern "C" int evf_concat_up2_fwd(const void* const* src, const int* C, const int* ld, int n, int B, int H, int W, float* out, int ldo,
                                  void* stream) {
  if (!src || !C || !ld || !out || n <= 0 || n > 6 || !src[0] || B <= 0 || H <= 0 || W <= 0 || (ldo & 3) || (((uintptr_t)out) & 15))
    return EVF_EINVAL;
  CatParts p;
  int tot = 0;
  for (int k = 0; k < 6; ++k) {
    p.src[k] = k < n ? (const float*)src[k] : nullptr;
    p.C[k] = k < n ? C[k] : 0;
    p.ld[k] = k < n ? ld[k] : 0;
    p.off[k] = tot;
    if (k < n && (C[k] <= 0 || (C[k] & 1) || (src[k] && (ld[k] < C[k] || (ld[k] & 1) || (((uintptr_t)src[k]) & 7))))) return EVF_EINVAL;
    tot += p.C[k];
  }
  p.off[6] = tot, p.n = n;
  if (ldo < tot || (tot & 3)) return EVF_EINVAL;
  const long total = (long)B * H * W * (tot / 4);  // (a thread per source pixel and channel quad: its 2 x 2 output pixels)
  hipLaunchKernelGGL(k_concat_up2_fwd, dim3(evf_cdiv(total, 256L)), dim3(256), 0, EVF_STREAM(stream), p, B, H, W, tot, out, ldo);
  return evf_status();
}

// nearest up-sampling by an integer factor of planes [n][h][w] (flow maps, NCHW): models/model.py:529-539
__global__ void k_upnear_fwd(const float* __restrict__ x, long n, int h, int w, int f, float* __restrict__ y) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int OW = w * f, OH = h * f;
  if (idx >= n * OH * OW) return;
  const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH);
  const long p = idx / ((long)OW * OH);
  y[idx] = x[(p * h + oy / f) * w + ox / f];
}
__global__ void k_upnear_bwd(const float* __restrict__ gy, long n, int h, int w, int f, float* __restrict__ gx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * h * w) return;
  const int ix = (int)(idx % w), iy = (int)((idx / w) % h);
  const long p = idx / ((long)w * h);
  const int OW = w * f, OH = h * f;
  float s = 0.f;
  for (int dy = 0; dy < f; ++dy)
    for (int dx = 0; dx < f; ++dx) s += gy[(p * OH + iy * f + dy) * OW + ix * f + dx];
  gx[idx] = s;
}
extern "C" int evf_upsample_nearest_fwd(const float* x, int64_t planes, int h, int w, int factor, float* y, void* stream) {
  if (!x || !y || planes <= 0 || h <= 0 || w <= 0 || factor < 1) return EVF_EINVAL;
  const long total = planes * h * w * factor * factor;
  hipLaunchKernelGGL(k_upnear_fwd, dim3(evf_cdiv(total, 256)), dim3(256), 0, EVF_STREAM(stream), x, (long)planes, h, w,
                     factor, y);
  return evf_status();
}
extern "C" int evf_upsample_nearest_bwd(const float* g_y, int64_t planes, int h, int w, int factor, float* g_x,
                                        void* stream) {
  if (!g_y || !g_x || planes <= 0 || h <= 0 || w <= 0 || factor < 1) return EVF_EINVAL;
  hipLaunchKernelGGL(k_upnear_bwd, dim3(evf_cdiv(planes * h * w, 256)), dim3(256), 0, EVF_STREAM(stream), g_y,
                     (long)planes, h, w, factor, g_x);
  return evf_status();
}

// ---------------------------------------------------------------------------
// activations and ConvGRU gate algebra (flat element-wise)
// ---------------------------------------------------------------------------
// kind: 0 identity, 1 tanh, 2 sigmoid, 3 relu.   y = act(x [+ r])
__global__ void k_act_fwd(int kind, const float* x, const float* __restrict__ r, long n, float* y) {  // y may alias x
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i] + (r ? r[i] : 0.f);
  if (kind == 1) v = tanhf(v);
  else if (kind == 2) v = 1.0f / (1.0f + expf(-v));
  else if (kind == 3) v = fmaxf(v, 0.f);
  y[i] = v;
}
// g_x = g_y * act'(.) expressed through the output y
__global__ void k_act_bwd(int kind, const float* __restrict__ y, const float* __restrict__ gy, long n, float* __restrict__ gx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float o = y[i];
  float d = 1.f;
  if (kind == 1) d = 1.0f - o * o;
  else if (kind == 2) d = o * (1.0f - o);
  else if (kind == 3) d = o > 0.f ? 1.f : 0.f;
  gx[i] = gy[i] * d;
}
extern "C" int evf_act_fwd(int kind, const float* x, const float* residual, int64_t n, float* y, void* stream) {
  if (!x || !y || n <= 0 || kind < 0 || kind > 3) return EVF_EINVAL;
  hipLaunchKernelGGL(k_act_fwd, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), kind, x, residual, (long)n, y);
  return evf_status();
}
extern "C" int evf_act_bwd(int kind, const float* y, const float* g_y, int64_t n, float* g_x, void* stream) {
  if (!y || !g_y || !g_x || n <= 0 || kind < 0 || kind > 3) return EVF_EINVAL;
  hipLaunchKernelGGL(k_act_bwd, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), kind, y, g_y, (long)n, g_x);
  return evf_status();
}

// ConvGRU (submodules.py:412-416).  Stage 1: u = sigmoid(cu), r = sigmoid(cr), hr = h * r.
// Stage 2: o = tanh(co), new = h * (1 - u) + o * u.
__global__ void k_gru_gates_fwd(const float* __restrict__ cu, const float* __restrict__ cr, const float* __restrict__ h,
                                long n, float* __restrict__ u, float* __restrict__ r, float* __restrict__ hr) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float uu = 1.0f / (1.0f + expf(-cu[i])), rr = 1.0f / (1.0f + expf(-cr[i]));
  u[i] = uu, r[i] = rr, hr[i] = (h ? h[i] : 0.f) * rr;
}
__global__ void k_gru_out_fwd(const float* __restrict__ co, const float* __restrict__ h, const float* __restrict__ u, long n,
                              float* __restrict__ o, float* __restrict__ hn) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float oo = tanhf(co[i]), hh = h ? h[i] : 0.f;
  o[i] = oo, hn[i] = hh * (1.0f - u[i]) + oo * u[i];
}
// backward of stage 2: g_new -> g_co, g_u (pre-sigmoid: g_cu), partial g_h
__global__ void k_gru_out_bwd(const float* __restrict__ g_new, const float* __restrict__ h, const float* __restrict__ u,
                              const float* __restrict__ o, long n, float* __restrict__ g_co, float* __restrict__ g_cu,
                              float* __restrict__ g_h) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = g_new[i], uu = u[i], oo = o[i], hh = h ? h[i] : 0.f;
  g_co[i] = g * uu * (1.0f - oo * oo);
  g_cu[i] = g * (oo - hh) * uu * (1.0f - uu);
  g_h[i] = g * (1.0f - uu);
}
// backward of stage 1: g_hr (from the out-gate conv) -> g_cr, g_h += g_hr * r
__global__ void k_gru_gates_bwd(const float* __restrict__ g_hr, const float* __restrict__ h, const float* __restrict__ r,
                                long n, float* __restrict__ g_cr, float* __restrict__ g_h) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = g_hr[i], rr = r[i], hh = h ? h[i] : 0.f;
  g_cr[i] = g * hh * rr * (1.0f - rr);
  g_h[i] += g * rr;
}
extern "C" int evf_gru_gates_fwd(const float* cu, const float* cr, const float* h, int64_t n, float* u, float* r, float* hr,
                                 void* stream) {
  if (!cu || !cr || !u || !r || !hr || n <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_gru_gates_fwd, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), cu, cr, h, (long)n, u, r, hr);
  return evf_status();
}
extern "C" int evf_gru_out_fwd(const float* co, const float* h, const float* u, int64_t n, float* o, float* h_new,
                               void* stream) {
  if (!co || !u || !o || !h_new || n <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_gru_out_fwd, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), co, h, u, (long)n, o, h_new);
  return evf_status();
}
extern "C" int evf_gru_out_bwd(const float* g_new, const float* h, const float* u, const float* o, int64_t n, float* g_co,
                               float* g_cu, float* g_h, void* stream) {
  if (!g_new || !u || !o || !g_co || !g_cu || !g_h || n <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_gru_out_bwd, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), g_new, h, u, o, (long)n, g_co,
                     g_cu, g_h);
  return evf_status();
}
extern "C" int evf_gru_gates_bwd(const float* g_hr, const float* h, const float* r, int64_t n, float* g_cr, float* g_h,
                                 void* stream) {
  if (!g_hr || !r || !g_cr || !g_h || n <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_gru_gates_bwd, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), g_hr, h, r, (long)n, g_cr,
                     g_h);
  return evf_status();
}

// ConvLSTM gate algebra (models/submodules.py:357-374).  gates [npix, 4*Ch] = conv(cat(x, h)) with the channel chunks
// in | remember | out | cell;  i, r, o = sigmoid, cg = tanh;  cell' = r * cell + i * cg;  hidden = o * tanh(cell').
// The activated gates overwrite `gates` (saved for the backward).
__global__ void k_lstm_fwd(float* __restrict__ gates, const float* __restrict__ prev_cell, long npix, int Ch,
                           float* __restrict__ cell, float* __restrict__ hidden) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * Ch) return;
  const long pix = i / Ch;
  const int c = (int)(i - pix * Ch);
  float* gp = gates + pix * 4 * Ch + c;
  const float ig = 1.0f / (1.0f + expf(-gp[0])), rg = 1.0f / (1.0f + expf(-gp[Ch])), og = 1.0f / (1.0f + expf(-gp[2 * Ch])),
              cg = tanhf(gp[3 * Ch]);
  const float cn = rg * (prev_cell ? prev_cell[i] : 0.f) + ig * cg;
  gp[0] = ig, gp[Ch] = rg, gp[2 * Ch] = og, gp[3 * Ch] = cg;
  cell[i] = cn;
  hidden[i] = og * tanhf(cn);
}
__global__ void k_lstm_bwd(const float* __restrict__ g_hidden, const float* __restrict__ g_cell, const float* __restrict__ gates,
                           const float* __restrict__ cell, const float* __restrict__ prev_cell, long npix, int Ch,
                           float* __restrict__ g_gates, float* __restrict__ g_prev_cell) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * Ch) return;
  const long pix = i / Ch;
  const int c = (int)(i - pix * Ch);
  const float* gp = gates + pix * 4 * Ch + c;
  const float ig = gp[0], rg = gp[Ch], og = gp[2 * Ch], cg = gp[3 * Ch];
  const float tc = tanhf(cell[i]), gh = g_hidden ? g_hidden[i] : 0.f, pc = prev_cell ? prev_cell[i] : 0.f;
  const float gc = (g_cell ? g_cell[i] : 0.f) + gh * og * (1.0f - tc * tc);
  float* go = g_gates + pix * 4 * Ch + c;
  go[0] = gc * cg * ig * (1.0f - ig);
  go[Ch] = gc * pc * rg * (1.0f - rg);
  go[2 * Ch] = gh * tc * og * (1.0f - og);
  go[3 * Ch] = gc * ig * (1.0f - cg * cg);
  if (g_prev_cell) g_prev_cell[i] = gc * rg;
}
extern "C" int evf_lstm_fwd(float* gates, const float* prev_cell, int64_t npix, int Ch, float* cell, float* hidden,
                            void* stream) {
  if (!gates || !cell || !hidden || npix <= 0 || Ch <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_lstm_fwd, dim3(evf_cdiv(npix * Ch, 256)), dim3(256), 0, EVF_STREAM(stream), gates, prev_cell, (long)npix,
                     Ch, cell, hidden);
  return evf_status();
}
extern "C" int evf_lstm_bwd(const float* g_hidden, const float* g_cell, const float* gates, const float* cell,
                            const float* prev_cell, int64_t npix, int Ch, float* g_gates, float* g_prev_cell, void* stream) {
  if ((!g_hidden && !g_cell) || !gates || !cell || !g_gates || npix <= 0 || Ch <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_lstm_bwd, dim3(evf_cdiv(npix * Ch, 256)), dim3(256), 0, EVF_STREAM(stream), g_hidden, g_cell, gates,
                     cell, prev_cell, (long)npix, Ch, g_gates, g_prev_cell);
  return evf_status();
}

// ---------------------------------------------------------------------------
// stand-alone spike functions (models/spiking_util.py:13-109): z = (x - thresh > 0) as fp32,
// backward g * surrogate(x - thresh).  thresh is one scalar (device pointer).
// ---------------------------------------------------------------------------
__global__ void k_spike_fwd(const float* __restrict__ x, const float* __restrict__ thresh, int tstride, long n,
                            float* __restrict__ z) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) z[i] = (x[i] - thresh[i * tstride]) > 0.f ? 1.0f : 0.f;
}
__global__ void k_spike_bwd(int kind, const float* __restrict__ x, const float* __restrict__ thresh, int tstride,
                            const float* __restrict__ g, float width, long n, float* __restrict__ gx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gx[i] = g[i] * ng_surrogate(kind, x[i] - thresh[i * tstride], width);
}
extern "C" int evf_spike_fwd(const float* x, const float* thresh, int thresh_per_element, int64_t n, float* z, void* stream) {
  if (!x || !thresh || !z || n <= 0) return EVF_EINVAL;
  hipLaunchKernelGGL(k_spike_fwd, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), x, thresh,
                     thresh_per_element ? 1 : 0, (long)n, z);
  return evf_status();
}
extern "C" int evf_spike_bwd(int surrogate, const float* x, const float* thresh, int thresh_per_element, const float* g,
                             float width, int64_t n, float* g_x, void* stream) {
  if (!x || !thresh || !g || !g_x || n <= 0 || surrogate < 0 || surrogate > 3) return EVF_EINVAL;
  hipLaunchKernelGGL(k_spike_bwd, dim3(evf_cdiv(n, 256)), dim3(256), 0, EVF_STREAM(stream), surrogate, x, thresh,
                     thresh_per_element ? 1 : 0, g, width, (long)n, g_x);
  return evf_status();
}
