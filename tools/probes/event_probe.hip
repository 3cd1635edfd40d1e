// Can HIP events be recorded INTO a stream capture as external event-record nodes and read after a replay?
// (bench.py wants the per-kernel durations of the replayed step graph.)  hipcc --offload-arch=gfx950 event_probe.hip -o bin/event_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); printf("%-60s -> %d (%s)\n", #x, (int)e_, hipGetErrorString(e_)); } while (0)
__global__ void spin(float* p, int n) {
  float a = p[threadIdx.x];
  for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
  p[threadIdx.x] = a;
}
int main() {
  float* d;
  CK(hipMalloc(&d, 4096));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (int variant = 0; variant < 3; ++variant) {
    printf("---- variant %d (0: hipEventRecordWithFlags external, 1: plain hipEventRecord in capture, 2: explicit graph nodes)\n", variant);
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));
    hipGraph_t g;
    hipGraphExec_t ge;
    if (variant < 2) {
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      if (variant == 0) CK(hipEventRecordWithFlags(e0, st, hipEventRecordExternal)); else CK(hipEventRecord(e0, st));
      spin<<<64, 64, 0, st>>>(d, 20000);
      if (variant == 0) CK(hipEventRecordWithFlags(e1, st, hipEventRecordExternal)); else CK(hipEventRecord(e1, st));
      spin<<<64, 64, 0, st>>>(d, 60000);
      if (variant == 0) CK(hipEventRecordWithFlags(e2, st, hipEventRecordExternal)); else CK(hipEventRecord(e2, st));
      CK(hipStreamEndCapture(st, &g));
    } else {
      CK(hipGraphCreate(&g, 0));
      hipGraphNode_t n0, k0, n1, k1, n2;
      CK(hipGraphAddEventRecordNode(&n0, g, nullptr, 0, e0));
      void* args0[2]; int c0 = 20000, c1 = 60000; args0[0] = &d; args0[1] = &c0;
      hipKernelNodeParams kp = {};
      kp.func = (void*)spin; kp.gridDim = dim3(64); kp.blockDim = dim3(64); kp.kernelParams = args0;
      CK(hipGraphAddKernelNode(&k0, g, &n0, 1, &kp));
      CK(hipGraphAddEventRecordNode(&n1, g, &k0, 1, e1));
      void* args1[2]; args1[0] = &d; args1[1] = &c1; kp.kernelParams = args1;
      CK(hipGraphAddKernelNode(&k1, g, &n1, 1, &kp));
      CK(hipGraphAddEventRecordNode(&n2, g, &k1, 1, e2));
    }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) {
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      float t01 = -1, t12 = -1;
      CK(hipEventElapsedTime(&t01, e0, e1));
      CK(hipEventElapsedTime(&t12, e1, e2));
      printf("replay %d: e0->e1 %.3f ms, e1->e2 %.3f ms\n", r, t01, t12);
    }
    (void)hipGetLastError();
  }
  return 0;
}
