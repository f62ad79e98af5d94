// What the HBM system delivers for the apply kernels' access pattern: every wave owns 16 streams
// ("chains") that are `gap` bytes apart and visits each stream R rows (R*60 bytes) at a time, the
// visits of all resident waves interleaved in time.  Reads only, or reads + writes of the same shape to a
// second buffer.  Variants: TILE (lane (g,n) loads 16 bytes of chain n's row at +16g, row by row: the
// MFMA tile layout) and PIECES (a chain's R*60 bytes as consecutive 16-byte pieces dealt to lanes).
//   hipcc --offload-arch=gfx950 -O3 stream_pattern.hip -o stream_pattern && ./stream_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int R, bool PIECES, bool WRITE, bool DESC>
__global__ __launch_bounds__(256) void k(const char *__restrict__ x, char *__restrict__ y, long long nwaves, int T,
                                         float *__restrict__ out) {
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const long long gap = (long long)T * 60;
    const char *base = x + wave * 16 * gap;
    char *wbase = y + wave * 16 * gap;
    const int nblk = T / R;
    constexpr int P = R * 60 / 16;                // pieces per chain per block
    constexpr int NL = PIECES ? (16 * P + 63) / 64 : R;
    f4 acc = {0, 0, 0, 0};
    f4 cur[NL], nxt[NL];
    auto load = [&](int blk, f4 (&v)[NL]) {
        const int b = DESC ? nblk - 1 - blk : blk;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            long long off;
            if (PIECES) {
                const int id = i * 64 + lane;
                const int c = id / P, kk = id - c * P;
                off = (c < 16) ? c * gap + (long long)b * R * 60 + kk * 16 : 0;
            } else {
                off = n * gap + ((long long)b * R + i) * 60 + 16 * g;
            }
            v[i] = *reinterpret_cast<const f4u *>(base + off);
        }
    };
    load(0, nxt);
    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
        for (int i = 0; i < NL; ++i) cur[i] = nxt[i];
        if (blk + 1 < nblk) load(blk + 1, nxt);
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += cur[i];
        if (WRITE) {
            const int b = DESC ? nblk - 1 - blk : blk;
            constexpr int WL = (16 * P + 63) / 64;
#pragma unroll
            for (int i = 0; i < WL; ++i) {
                const int id = i * 64 + lane;
                const int c = id / P, kk = id - c * P;
                if (c < 16) *reinterpret_cast<f4u *>(wbase + c * gap + (long long)b * R * 60 + kk * 16) = cur[i % NL] + acc;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int R, bool PIECES, bool WRITE, bool DESC>
static void run(const char *x, char *y, float *out, int T, long long bytes) {
    const long long nwaves = bytes / (16ll * T * 60);
    const unsigned blocks = (unsigned)((nwaves + 3) / 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<R, PIECES, WRITE, DESC>), dim3(blocks), dim3(256), 0, 0, x, y, nwaves, T, out);
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<R, PIECES, WRITE, DESC>), dim3(blocks), dim3(256), 0, 0, x, y, nwaves, T, out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    const double moved = (double)nwaves * 16 * T * 60 * (WRITE ? 2 : 1);
    printf("R=%2d %-6s %-5s %-4s T=%4d: %.3f ms  %.2f TB/s\n", R, PIECES ? "pieces" : "tile", WRITE ? "r+w" : "read",
           DESC ? "desc" : "asc", T, ms / 5, moved / (ms / 5 * 1e-3) / 1e12);
}

int main() {
    const long long bytes = 6144000000ll;
    char *x, *y; float *out;
    (void)hipMalloc(&x, bytes + 4096); (void)hipMalloc(&y, bytes + 4096); (void)hipMalloc(&out, 4);
    (void)hipMemset(x, 0, bytes + 4096);
    const int T = 512;
    run<8, false, false, false>(x, y, out, T, bytes);
    run<8, false, false, true>(x, y, out, T, bytes);
    run<8, true, false, false>(x, y, out, T, bytes);
    run<16, false, false, false>(x, y, out, T, bytes);
    run<16, true, false, false>(x, y, out, T, bytes);
    run<32, true, false, false>(x, y, out, T, bytes);
    run<64, true, false, false>(x, y, out, T, bytes);
    run<32, true, false, true>(x, y, out, T, bytes);
    run<8, false, true, true>(x, y, out, T, bytes);
    run<8, true, true, true>(x, y, out, T, bytes);
    run<16, true, true, true>(x, y, out, T, bytes);
    run<32, true, true, true>(x, y, out, T, bytes);
    run<64, true, true, false>(x, y, out, T, bytes);
    return 0;
}
