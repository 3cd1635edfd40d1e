#include <hip/hip_runtime.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  // lane i of a 16-lane group: row i/4 (pitch 96 elements), piece i%4 (4 elements)
  const unsigned short* p = s + ((l & 15) >> 2) * 96 + (l & 3) * 4 + (l >> 4) * 16;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[4096], *d, *o, r[256];
  for (int i = 0; i < 4096; ++i) h[i] = i;
  hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, r[4*l], r[4*l+1], r[4*l+2], r[4*l+3]);
  return 0;
}
