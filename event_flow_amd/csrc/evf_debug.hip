// Debugging aid: fill the LDS of every CU with a bit pattern (default: a quiet NaN).  LDS is not cleared between kernels -- a
// kernel that reads a word it never wrote gets whatever the previous kernel on that CU left there, i.e. usually small finite
// numbers, and the result looks right until the chip idles or another kernel runs in between.  With the pattern in place such
// a read shows up as NaN in the output (0 x NaN = NaN).  tests/test_gpu_lds_poison.py runs the hot path with a poison launch in
// front of every library call (EVF_DEBUG_POISON_LDS=1 in event_flow_amd/_lib.py).
#include "evf_common.h"

#define PZ_BYTES (160 * 1024)  // LDS per CU (gfx950)

__global__ __launch_bounds__(1024) void k_poison_lds(uint32_t pattern, unsigned long long spin) {
  extern __shared__ __attribute__((aligned(16))) uint32_t pz[];
  for (int i = threadIdx.x; i < PZ_BYTES / 4; i += blockDim.x) pz[i] = pattern;
  __syncthreads();
  // stay resident for a moment so that the blocks of the launch spread over all CUs (one block fills a CU's LDS)
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (pz[threadIdx.x] != pattern) __builtin_trap();  // (keeps the stores alive)
}

extern "C" int evf_debug_poison_lds(uint32_t pattern, void* stream) {
  static bool attr = false;
  if (!attr) {
    const int rc = evf_hip(hipFuncSetAttribute((const void*)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, PZ_BYTES));
    if (rc) return rc;
    attr = true;
  }
  int dev = 0, ncu = 256;
  hipDeviceProp_t pr;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
    ncu = pr.multiProcessorCount;
  // two rounds of one block per CU, ~20 us resident each (wall clock: 100 MHz)
  hipLaunchKernelGGL(k_poison_lds, dim3(2 * ncu), dim3(1024), PZ_BYTES, EVF_STREAM(stream), pattern ? pattern : 0x7FC00000u, 2000ull);
  return evf_status();
}

// hipMemsetAsync as a kernel launch (evf_memset_async, evf_common.h): what the Python host uses for every buffer it clears on a
// stream that may be capturing (event_flow_amd/_lib.py: zeros / zero_).
extern "C" int evf_memset(void* dst, int value, size_t bytes, void* stream) {
  if (!dst && bytes) return EVF_EINVAL;
  return evf_hip(evf_memset_async(dst, value, bytes, EVF_STREAM(stream)));
}
