// Attainable HBM read bandwidth with a minimal kernel: every lane streams 16-byte loads,
// `unroll` independent loads in flight, grid-stride.  hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ x, size_t n4, float *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += x[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U>
static void run(const f4 *x, size_t n4, float *out, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_read<U>, dim3(blocks), dim3(256), 0, 0, x, n4, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_read<U>, dim3(blocks), dim3(256), 0, 0, x, n4, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    printf("unroll %d blocks %6d: %.3f ms  %.2f TB/s\n", U, blocks, ms / 10, n4 * 16.0 / (ms / 10 * 1e-3) / 1e12);
}

int main() {
    const size_t n4 = (size_t)1536 * 1000 * 1000 / 4;      // 6.144 GB
    f4 *x; float *out;
    hipMalloc(&x, n4 * 16); hipMalloc(&out, 4);
    hipMemset(x, 0, n4 * 16);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        run<1>(x, n4, out, blocks);
        run<4>(x, n4, out, blocks);
        run<8>(x, n4, out, blocks);
    }
    return 0;
}
