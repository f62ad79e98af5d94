#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xsum_builtin(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v), u2 = u;
  asm volatile("" : "+v"(u2));
  auto r = __builtin_amdgcn_permlane16_swap(u, u2, false, false);
  float s = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
  unsigned w = __builtin_bit_cast(unsigned, s), w2 = w;
  asm volatile("" : "+v"(w2));
  auto r2 = __builtin_amdgcn_permlane32_swap(w, w2, false, false);
  return __builtin_bit_cast(float, r2[0]) + __builtin_bit_cast(float, r2[1]);
}
__device__ __forceinline__ float xsum_asm(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));
  float s = a + b;
  a = s; b = s;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(const float* in, float* o1, float* o2) {
  int l = threadIdx.x;
  o1[l] = xsum_builtin(in[l]);
  o2[l] = xsum_asm(in[l]);
}
int main() {
  float h[64], r1[64], r2[64]; for (int i = 0; i < 64; ++i) h[i] = (i & 15) + 100.f * (i >> 4);
  float *d, *a, *b; hipMalloc(&d, 256); hipMalloc(&a, 256); hipMalloc(&b, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, b);
  hipMemcpy(r1, a, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, b, 256, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int i = 0; i < 64; ++i) { float want = 4.f * (i & 15) + 600.f; bad1 += r1[i] != want; bad2 += r2[i] != want; }
  printf("builtin: bad=%d (lane0 %g lane17 %g lane63 %g)  asm: bad=%d (lane0 %g lane17 %g lane63 %g) want lane0 %g lane17 %g\n", bad1, r1[0], r1[17], r1[63], bad2, r2[0], r2[17], r2[63], 600.f, 604.f);
  return 0;
}
