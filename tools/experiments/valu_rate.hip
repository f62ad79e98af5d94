// VALU / MFMA issue-rate microbenchmark: cycles per wave-instruction per SIMD at W waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define N 16   // independent accumulators
template <int KIND> __global__ void k(float* out, int iters, float a, float b) {
  float x[N]; f2 p[N]; f4 m[4];
  for (int i = 0; i < N; ++i) { x[i] = threadIdx.x + i; p[i] = (f2){x[i], x[i] + 1}; }
  for (int i = 0; i < 4; ++i) m[i] = (f4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (KIND == 0) x[i] = __builtin_fmaf(x[i], a, b);                       // v_fma_f32
      if (KIND == 1) { f2 aa = {a, a}, bb = {b, b}; p[i] = p[i] * aa + bb; }   // v_pk_fma_f32
      if (KIND == 2) x[i] = x[i] * a;                                         // v_mul_f32
      if (KIND == 3) x[i] = __builtin_fmaxf(x[i], a);                         // v_max_f32
      if (KIND == 4) { f2 aa = {a, a}; p[i] = p[i] * aa; }                    // v_pk_mul_f32
      if (KIND == 5) x[i] = x[i] + a;                                         // v_add_f32
      if (KIND == 6) m[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[i & 3], 0, 0, 0);
      if (KIND == 7) { m[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[i & 3], 0, 0, 0); x[i] = __builtin_fmaf(x[i], a, b); }
    }
  }
  float s = 0; for (int i = 0; i < N; ++i) s += x[i] + p[i].x + p[i].y; for (int i = 0; i < 4; ++i) s += m[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, float* d, int per) {
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 256 * wps;  // 256-thread blocks = 4 waves = 1 per SIMD per block
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)iters * N * per * wps;   // wave-instructions per SIMD
    printf("%-14s waves/SIMD=%d  %.2f ms  -> %.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / instr);
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_fma_f32", d, 1); run<1>("v_pk_fma_f32", d, 1); run<2>("v_mul_f32", d, 1); run<3>("v_max_f32", d, 1);
  run<4>("v_pk_mul_f32", d, 1); run<5>("v_add_f32", d, 1); run<6>("mfma16x16x4", d, 1); run<7>("mfma+fma", d, 1);
  return 0;
}
