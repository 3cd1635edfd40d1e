// Micro-benchmark: what one wave per SIMD pays for the parts of the input-gradient matrix phase on MI355X (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mma_probe.hip -o tools/probes/bin/mma_probe && tools/probes/bin/mma_probe
// One block of 4 waves per CU (LDS-limited like k_dgrad_diag_dma), every wave runs `reps` phases of 9 taps x 12
// v_mfma_f32_32x32x16_bf16 and reports shader cycles per phase.  Template flags switch the other work of the phase on and off:
//   W   the 54 weight-fragment reads (ds_read_b128), requested one tap ahead
//   G   the gradient-fragment reads: 36 (with D) or 54 (without)
//   D   the middle fragments built by DPP moves behind the dx = 0 tap's MFMAs (144 v_mov_b32_dpp)
//   C   CHAINED accumulators: the six MFMAs of a K group on one accumulator back to back (the order of dg_matrix_phase)
//       instead of alternating acc0 / acc1
//   L   10 LDS-DMA pieces (global_load_lds_dwordx4) per phase issued behind MFMAs (+ their address arithmetic)
//   E   an epilogue (4 ds_write_b128, 4 ds_read_b128, 4 global_store_dwordx4) behind MFMAs
//   X   no MFMAs at all (what the rest costs alone)
//   P   waves per SIMD: 2 blocks' worth is impossible (LDS); `two` = launch 512 threads with the SAME work per wave
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

enum { FW = 1, FG = 2, FD = 4, FC = 8, FL = 16, FE = 32, FX = 64 };
#define HW 34
#define PLANE (13 * 64)
#define BUF (3 * PLANE)

__device__ __forceinline__ uint32_t dpp_mid1(uint32_t f0, uint32_t f2) {
  uint32_t r = (uint32_t)__builtin_amdgcn_mov_dpp((int)f0, 0x130, 0xF, 0xF, true);
  return (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)f2, 0x138, 0xA, 0x8, false);
}
struct Fr3 {
  uint4 h, m, l;
};

template <int F>
__global__ __launch_bounds__(512) void k_probe(const uint4* __restrict__ gsrc, float* __restrict__ out,
                                               unsigned long long* __restrict__ cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* s_w = (uint4*)smem_raw;
  uint4* s_a = s_w + 54 * 64;
  float* s_st = (float*)(s_a + 2 * BUF);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, w4 = wv & 3;
  const int i = lane & 31, kg = lane >> 5;
  for (int k = tid; k < 54 * 64 + 2 * BUF; k += blockDim.x) s_w[k] = make_uint4(0x3F803F80u, 0x3C003C00u, 0x3F803F80u, 0x38003800u);
  __syncthreads();
  const int hp0 = w4 * HW + i;
  f32x16 acc0 = {0}, acc1 = {0}, accp = {0};
  uint32_t keep = 0;
  float* st = s_st + w4 * (32 * 36);  // (512 threads: waves w and w + 4 share a staging tile -- timing only)
  auto ldw = [&](Fr3& w, int g) {
    const uint4* wf = s_w + (g * 3) * 64 + lane;
    w.h = wf[0], w.m = wf[64], w.l = wf[128];
  };
  auto ldg = [&](Fr3& a, int dy, int dx, int m, int buf) {
    const int hp = hp0 + dy * HW + dx, sw = (hp >> 2) & 3, slot = hp * 4 + ((2 * m + kg) ^ sw);
    const uint4* pa = s_a + buf * BUF;
    a.h = pa[slot], a.m = pa[PLANE + slot], a.l = pa[2 * PLANE + slot];
  };
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < reps; ++rep) {
    const int buf = rep & 1;
    Fr3 w[2][2], r0[2], r2[2][2], mid[2], gq[2][2];
    ldw(w[0][0], 0), ldw(w[0][1], 1);
    if (F & FD) {
      for (int m = 0; m < 2; ++m) ldg(r0[m], 0, 0, m, buf), ldg(r2[0][m], 0, 2, m, buf);
    } else {
      ldg(gq[0][0], 0, 0, 0, buf), ldg(gq[0][1], 0, 0, 1, buf);
    }
    mid[0] = r0[0], mid[1] = r0[1];
    float4 ev[4];
#pragma unroll
    for (int tau = 0; tau < 9; ++tau) {
      const int dy = tau / 3, dx = tau - 3 * dy, tp = tau & 1;
      if (tau + 1 < 9) {
        if (F & FW) ldw(w[tp ^ 1][0], 2 * (tau + 1)), ldw(w[tp ^ 1][1], 2 * (tau + 1) + 1);
        else w[tp ^ 1][0] = w[tp][0], w[tp ^ 1][1] = w[tp][1];
        if (!(F & FD)) {
          if (F & FG) ldg(gq[tp ^ 1][0], (tau + 1) / 3, (tau + 1) % 3, 0, buf), ldg(gq[tp ^ 1][1], (tau + 1) / 3, (tau + 1) % 3, 1, buf);
          else gq[tp ^ 1][0] = gq[tp][0], gq[tp ^ 1][1] = gq[tp][1];
        } else if (dx == 1 && dy < 2 && (F & FG)) {
          for (int m = 0; m < 2; ++m) ldg(r0[m], dy + 1, 0, m, buf), ldg(r2[(dy + 1) & 1][m], dy + 1, 2, m, buf);
        } else if (dx == 1 && dy < 2) {
          r2[(dy + 1) & 1][0] = r2[dy & 1][0], r2[(dy + 1) & 1][1] = r2[dy & 1][1];
        }
      }
      const Fr3 c0 = (F & FD) ? (dx == 0 ? r0[0] : (dx == 1 ? mid[0] : r2[dy & 1][0])) : gq[tp][0];
      const Fr3 c1 = (F & FD) ? (dx == 0 ? r0[1] : (dx == 1 ? mid[1] : r2[dy & 1][1])) : gq[tp][1];
      const Fr3 &w0 = w[tp][0], &w1 = w[tp][1];
      const bf16x8 w0h = *(const bf16x8*)&w0.h, w0m = *(const bf16x8*)&w0.m, w0l = *(const bf16x8*)&w0.l;
      const bf16x8 w1h = *(const bf16x8*)&w1.h, w1m = *(const bf16x8*)&w1.m, w1l = *(const bf16x8*)&w1.l;
      const bf16x8 a0h = *(const bf16x8*)&c0.h, a0m = *(const bf16x8*)&c0.m, a0l = *(const bf16x8*)&c0.l;
      const bf16x8 a1h = *(const bf16x8*)&c1.h, a1m = *(const bf16x8*)&c1.m, a1l = *(const bf16x8*)&c1.l;
      Fr3 nm0 = mid[0], nm1 = mid[1];
      const bool build = (F & FD) && dx == 0;
      auto bld = [&](int e) {
        if (!build) return;
        const int m = e / 12, q = e % 12, pl = q / 4, d = q % 4;
        const Fr3 &a = r0[m], &b = r2[dy & 1][m];
        Fr3& o = m ? nm1 : nm0;
        const uint4& fa = pl == 0 ? a.h : (pl == 1 ? a.m : a.l);
        const uint4& fb = pl == 0 ? b.h : (pl == 1 ? b.m : b.l);
        uint4& fo = pl == 0 ? o.h : (pl == 1 ? o.m : o.l);
        const uint32_t va = d == 0 ? fa.x : (d == 1 ? fa.y : (d == 2 ? fa.z : fa.w));
        const uint32_t vb = d == 0 ? fb.x : (d == 1 ? fb.y : (d == 2 ? fb.z : fb.w));
        const uint32_t r = dpp_mid1(va, vb);
        if (d == 0) fo.x = r; else if (d == 1) fo.y = r; else if (d == 2) fo.z = r; else fo.w = r;
      };
      auto side = [&](int slot) {
        if (F & FL) {
#pragma unroll
          for (int j = 0; j < 10; ++j)
            if (slot == 2 + 6 * j) {
              // same address arithmetic shape as k_dgrad_diag_dma::dma_piece
              const int y = rep + j + (lane >> 4), x = (lane >> 2) + j;
              const bool in = (unsigned)y < 1000000u && (unsigned)x < 1000000u;
              const unsigned off = ((unsigned)(blockIdx.x * 4 + w4) * 13u + (unsigned)(j % 13)) * 1024u + (unsigned)lane * 16u;
              const unsigned long long mk = in ? ~0ull : 0ull;
              const unsigned long long a = (((unsigned long long)gsrc + off) & mk) | ((unsigned long long)gsrc & ~mk);
              __builtin_amdgcn_global_load_lds((glb_void*)a, (lds_void*)(s_a + (buf ^ 1) * BUF + ((w4 + 4 * j) % 39) * 64), 16, 0, 0);
            }
        }
        if (F & FE) {
          if (slot == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *(float4*)(st + i * 36 + 8 * q + 4 * kg) = make_float4(accp[4 * q], accp[4 * q + 1], accp[4 * q + 2], accp[4 * q + 3]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (slot == 13 + r) ev[r] = *(const float4*)(st + (8 * r + (lane >> 3)) * 36 + (lane & 7) * 4);
            if (slot == 40 + 12 * r)
              {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f t4 = {ev[r].x, ev[r].y, ev[r].z, ev[r].w};
                __builtin_nontemporal_store(t4, (v4f*)(out + ((size_t)(blockIdx.x * 8 + wv) * 4 + r) * 256 + lane * 4));
              }
          }
        }
      };
#define MF(ACC, A_, B_)                                                        \
  __builtin_amdgcn_sched_barrier(0);                                           \
  if (!(F & FX)) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC, 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
      if (F & FC) {  // chained: six on acc0, then six on acc1
        MF(acc0, w0m, a0m) bld(0), bld(1); side(tau * 12 + 0);
        MF(acc0, w0h, a0l) bld(2), bld(3); side(tau * 12 + 1);
        MF(acc0, w0l, a0h) bld(4), bld(5); side(tau * 12 + 2);
        MF(acc0, w0h, a0m) bld(6), bld(7); side(tau * 12 + 3);
        MF(acc0, w0m, a0h) bld(8), bld(9); side(tau * 12 + 4);
        MF(acc0, w0h, a0h) bld(10), bld(11); side(tau * 12 + 5);
        MF(acc1, w1m, a1m) bld(12), bld(13); side(tau * 12 + 6);
        MF(acc1, w1h, a1l) bld(14), bld(15); side(tau * 12 + 7);
        MF(acc1, w1l, a1h) bld(16), bld(17); side(tau * 12 + 8);
        MF(acc1, w1h, a1m) bld(18), bld(19); side(tau * 12 + 9);
        MF(acc1, w1m, a1h) bld(20), bld(21); side(tau * 12 + 10);
        MF(acc1, w1h, a1h) bld(22), bld(23); side(tau * 12 + 11);
      } else {
        MF(acc0, w0m, a0m) bld(0), bld(1); side(tau * 12 + 0);
        MF(acc1, w1m, a1m) bld(2), bld(3); side(tau * 12 + 1);
        MF(acc0, w0h, a0l) bld(4), bld(5); side(tau * 12 + 2);
        MF(acc1, w1h, a1l) bld(6), bld(7); side(tau * 12 + 3);
        MF(acc0, w0l, a0h) bld(8), bld(9); side(tau * 12 + 4);
        MF(acc1, w1l, a1h) bld(10), bld(11); side(tau * 12 + 5);
        MF(acc0, w0h, a0m) bld(12), bld(13); side(tau * 12 + 6);
        MF(acc1, w1h, a1m) bld(14), bld(15); side(tau * 12 + 7);
        MF(acc0, w0m, a0h) bld(16), bld(17); side(tau * 12 + 8);
        MF(acc1, w1m, a1h) bld(18), bld(19); side(tau * 12 + 9);
        MF(acc0, w0h, a0h) bld(20), bld(21); side(tau * 12 + 10);
        MF(acc1, w1h, a1h) bld(22), bld(23); side(tau * 12 + 11);
      }
#undef MF
      __builtin_amdgcn_sched_barrier(0);
      mid[0] = nm0, mid[1] = nm1;
    }
    accp = acc0 + acc1;
    keep += mid[0].h.x + mid[1].l.w;
    if (F & FL) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
  for (int q = 0; q < 16; ++q) out[(size_t)(blockIdx.x * blockDim.x + tid) * 16 + q] = accp[q] + (float)keep;
}


// ---- two waves per SIMD with DIFFERENT roles: waves 0..3 run `reps` phases of 108 bare MFMAs, waves 4..7 run `reps` blocks of
// NV independent VALU instructions (8 chains of v_fma_f32) -- how much does each slow the other?  role_mask: bit 0 = the
// MFMA team runs, bit 1 = the VALU team runs.
template <int NV>
__global__ __launch_bounds__(512) void k_mix(float* __restrict__ out, unsigned long long* __restrict__ cyc, int reps, int role_mask) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wv < 4) {
    if (role_mask & 1) {
      f32x16 acc0 = {0}, acc1 = {0};
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) a[e] = (__bf16)1.0f, b[e] = (__bf16)0.5f;
      for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int k = 0; k < 54; ++k) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
      }
      for (int q = 0; q < 16; ++q) out[(size_t)(blockIdx.x * 512 + tid) * 16 + q] = acc0[q] + acc1[q];
    }
  } else if (role_mask & 2) {
    float r[8];
    for (int e = 0; e < 8; ++e) r[e] = (float)(lane + e);
    const float c = 1.0000001f, d = 1e-6f;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
      for (int k = 0; k < NV / 8; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = __builtin_fmaf(r[e], c, d);
      }
    }
    float sum = 0.f;
    for (int e = 0; e < 8; ++e) sum += r[e];
    out[(size_t)(blockIdx.x * 512 + tid) * 16] = sum;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int NV>
void run_mix(int role_mask, float* out, unsigned long long* cyc, int reps) {
  hipLaunchKernelGGL(k_mix<NV>, dim3(256), dim3(512), 0, 0, out, cyc, reps, role_mask);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_mix<NV>, dim3(256), dim3(512), 0, 0, out, cyc, reps, role_mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> m, v;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back((double)h[b * 8 + w] / reps);
  std::sort(m.begin(), m.end()), std::sort(v.begin(), v.end());
  printf("mix NV=%4d roles %d: MFMA team %7.0f cycles per 108 MFMAs (%5.1f each)   VALU team %7.0f cycles per %d VALU (%5.2f each)   kernel %.1f us\n", NV,
         role_mask, m[m.size() / 2], m[m.size() / 2] / 108.0, v[v.size() / 2], NV, v[v.size() / 2] / NV, ms * 1e3);
}

template <int F>
void run(const char* name, int threads, const uint4* gsrc, float* out, unsigned long long* cyc, int reps) {
  const size_t lds = (size_t)(54 * 64 + 2 * BUF) * 16 + (size_t)4 * 32 * 36 * 4;
  hipFuncSetAttribute((const void*)k_probe<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int nblk = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k_probe<F>, dim3(nblk), dim3(threads), lds, 0, gsrc, out, cyc, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_probe<F>, dim3(nblk), dim3(threads), lds, 0, gsrc, out, cyc, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nblk * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> v;
  for (int b = 0; b < nblk; ++b)
    for (int w = 0; w < threads / 64; ++w) v.push_back((double)h[b * 8 + w] / reps);
  std::sort(v.begin(), v.end());
  const hipError_t err = hipGetLastError();
  printf("%-34s waves/SIMD %d  cycles/phase min %7.0f med %7.0f max %7.0f   per MFMA %5.1f   kernel %.1f us (%.0f ns/phase)  %s\n", name,
         threads / 256, v.front(), v[v.size() / 2], v.back(), v[v.size() / 2] / 108.0, ms * 1e3, ms * 1e6 / reps,
         err == hipSuccess ? "" : hipGetErrorString(err));
}

int main() {
  uint4* gsrc;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&gsrc, 64 << 20);
  hipMemset(gsrc, 0x3c, 64 << 20);
  hipMalloc(&out, (size_t)256 * 512 * 16 * 4 + (1 << 20));
  hipMalloc(&cyc, 256 * 8 * 8);
  const int reps = 24;
#define RUN(F, T) run<F>(#F, T, gsrc, out, cyc, reps)
  RUN(0, 256);
  RUN(FC, 256);
  RUN(FW, 256);
  RUN(FW | FC, 256);
  RUN(FW | FG, 256);
  RUN(FW | FG | FC, 256);
  RUN(FW | FG | FD, 256);
  RUN(FW | FG | FD | FC, 256);
  RUN(FD, 256);
  RUN(FW | FG | FD | FL, 256);
  RUN(FW | FG | FD | FE, 256);
  RUN(FW | FG | FD | FL | FE, 256);
  RUN(FW | FG | FL | FE, 256);
  RUN(FL, 256);
  RUN(FE, 256);
  RUN(FX | FW | FG | FD | FL | FE, 256);
  RUN(FX | FW | FG, 256);
  RUN(FX | FL, 256);
  RUN(FX | FD, 256);
  // two waves per SIMD sharing the LDS image (each does the full work: twice the matrix work per CU)
  RUN(0, 512);
  RUN(FC, 512);
  RUN(FW | FG | FD, 512);
  RUN(FW | FG | FD | FC, 512);
  RUN(FW | FG, 512);
  RUN(FW | FG | FD | FL | FE, 512);
  for (int roles : {1, 2, 3}) run_mix<800>(roles, out, cyc, reps);
  for (int roles : {2, 3}) run_mix<3200>(roles, out, cyc, reps);
  return 0;
}
