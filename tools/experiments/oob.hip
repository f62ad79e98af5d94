#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* E, float* out, int nbytes, float* st) {
  int l = threadIdx.x;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)E, 0, nbytes, 0x00020000);
  f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, nbytes - 16 + 4 * l, 0, 0));
  out[4*l+0]=v.x; out[4*l+1]=v.y; out[4*l+2]=v.z; out[4*l+3]=v.w;
  // store test: b128 store straddling the end of a smaller window
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, 64, 0x00020000);
  f4 w = {100.f + l, 200.f + l, 300.f + l, 400.f + l};
  if (l < 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, w), ro, 64 - 16 + 4 * l + 64 * 0, 0, 0);
}
int main() {
  const int N = 64; float h[N]; for (int i = 0; i < N; ++i) h[i] = i + 1;
  float *d, *o, *st; hipMalloc(&d, N * 4 + 64); hipMalloc(&o, 4 * 8 * 4); hipMalloc(&st, 128 * 4);
  hipMemset(d, 0x7f, N * 4 + 64); hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
  hipMemset(st, 0, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, N * 4, st);
  float r[32]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int l = 0; l < 8; ++l) printf("lane %d off=end-16+%d: %g %g %g %g\n", l, 4 * l, r[4*l], r[4*l+1], r[4*l+2], r[4*l+3]);
  float s[32]; hipMemcpy(s, st, sizeof(s), hipMemcpyDeviceToHost);
  for (int i = 10; i < 22; ++i) printf("st[%d]=%g ", i, s[i]); printf("\n");
  return 0;
}
