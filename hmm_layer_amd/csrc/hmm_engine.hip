// hmm_engine.hip — MI355X (gfx950 / CDNA4) HMM forward / backward / posterior engine.
//
// The per-timestep recurrence of the reference (hmm_layer/MsaHmmCell.py:73-106, driven by
// the Python loop hmm_layer/BaseRNN.py:217-227) is restructured as a three-phase scan
// over time chunks whose combine is a states x states matrix product on the f32 MFMA
// (v_mfma_f32_16x16x4_f32, exact f32 fma chain):
//
//   reduce  one wave per (sequence, chunk): the chunk operator  X <- diag(E_t) A^T X
//           (16 conditional forward vectors = the columns of one 16x16 MFMA tile),
//           columns rescaled by exact powers of two with integer exponent carry.
//           This is the reference's parallel_factor mode (hmm_layer/MsaHmmCell.py:122-142)
//           in linear space; one operator serves both directions.
//   scan    one wave per sequence: chunk-level prefix (alpha_hat entering each chunk)
//           and suffix (beta leaving each chunk) vectors — the role of
//           TotalProbabilityCell (hmm_layer/TotalProbabilityCell.py:30-49).
//   apply   one wave per 16 (sequence, chunk) pairs (the 16 columns of the MFMA tile):
//           exact cell-step semantics from the true prefix / suffix; forward pass writes
//           alpha_hat checkpoints every 16 steps, backward pass recomputes alpha_hat per
//           16-step block in registers and emits posteriors.
//
// MFMA operand trick: D = Aop * X accumulates over k in 4 steps; lane (g = lane>>4,
// n = lane&15) supplies k = 4g + kk at step kk instead of the canonical g + 4kk.  The
// permutation is applied to both operands, so the product is unchanged, and the output
// layout (row 4g + r, col n in register r) IS the next step's B-operand layout: the
// recurrence chains through registers with no transposes and no LDS.
//
// Layouts: everything the caller passes is row-major fp32 (k,b,L,q), q <= 16.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <vector>
#include "hmm_engine.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

#ifndef HMM_REDUCE_PF
#define HMM_REDUCE_PF 8        // emission rows in flight ahead of the reduce recurrence
#endif
#ifndef HMM_NT_STORE
#define HMM_NT_STORE 1  // posteriors / log alpha / log beta / dE leave through non-temporal 16-byte stores
#endif
#ifndef HMM_PSI
#define HMM_PSI 1       // experiments only: 0 compiles the clamp-born-mass sums out of k_backward (nothing is routed then)
#endif
#ifndef HMM_BWD_FULL
#define HMM_BWD_FULL 0  // k_backward: waves whose chains all own a full chunk skip the per-lane step masks — A/B in one
                        // process: 3.27 ms with, 2.97 without (two copies of the unrolled block)
#endif
#ifndef HMM_FWD_PF2
#define HMM_FWD_PF2 0
#endif
#ifndef HMM_CK_LD_NT
#define HMM_CK_LD_NT 0
#endif
#ifndef HMM_LD_AUX
#define HMM_LD_AUX 0   // cache-policy bits of the apply kernels' emission loads (experiments: 2 = nt)
#endif
#define QP 16          // padded state count = MFMA tile edge (scan kernels: q <= 16)
#define HMM_LARGEQ_MAX 4096   // serial-in-time GEMM path for 16 < q <= this
#ifndef HMM_SUB
#define HMM_SUB 8      // steps per apply block = alpha_hat checkpoint spacing
#endif
#define SUB HMM_SUB
#ifndef HMM_COALESCE_F
#define HMM_COALESCE_F 1   // apply kernels load emission rows in the coalesced loader layout (see permute_rows)
#endif
#ifndef HMM_COALESCE_B
#define HMM_COALESCE_B 0
#endif
#ifndef SCAN_PF
#define SCAN_PF 1      // hops of operator loads in flight in the group-level scan kernels.  These kernels are
                       // bandwidth-bound, not latency-bound (k_scan_inner reads the 345 MB of chunk operators in 65 us):
                       // four hops in flight changed nothing in k_scan_inner and cost k_scan_compose 50 -> 73 us
#endif
#define SCAN2_MIN_C 32  // chunks per sequence from which the chunk-level scan runs in two levels
#define MAX_T 512      // longest chunk (512 beat 1024 and 256 on b=1024 x L=1e5: more apply waves, short scan)
#define LN2 0.69314718055994530942

// ---- tuning / test options.  Explicit process-wide settings (hmm_set_option); the HMM_ENGINE_*
// environment variables only seed them, once, the first time any option is read.  The defaults
// are the measured best; results never depend on anything else outside a call's arguments.
#include <atomic>
static std::atomic<int> g_opt[HMM_OPT_COUNT];
static std::once_flag g_opt_once;
static void opt_seed() {
    static const char *names[HMM_OPT_COUNT] = {"HMM_ENGINE_CHUNK", "HMM_ENGINE_FORCE_DENSE", "HMM_ENGINE_SCAN2",
                                               "HMM_ENGINE_GROUPS", "HMM_ENGINE_EXACT", "HMM_ENGINE_PGCHUNK",
                                               "HMM_ENGINE_VGROUPS"};
    static const int defaults[HMM_OPT_COUNT] = {0, 0, 1, 1, HMM_EXACT_AUTO, 1, 0};
    for (int i = 0; i < HMM_OPT_COUNT; ++i) {
        const char *v = getenv(names[i]);
        g_opt[i].store(v ? atoi(v) : defaults[i]);
    }
}
static int opt(int which) {
    std::call_once(g_opt_once, opt_seed);
    return g_opt[which].load(std::memory_order_relaxed);
}

struct Plan {
    int k, b, L, q;
    int NB;            // k*b sequences
    int T, C;          // chunk length (multiple of SUB), chunks per sequence
    int nsub;          // T / SUB
    long long nchains; // NB * C
    int cpw;           // (sequence, chunk) pairs per apply wave: 16 = the tile's columns (1: exact plan of a very long sequence)
    int seq_start;     // 1: position 0 of the tensor is the first observation of its sequence (no transition into it);
                       // 0: a later time slab of a sequence-sharded call (hmm_seqshard_*)
    // workspace offsets (bytes)
    int G, gsize;      // two-level chunk scan: G groups of gsize chunks per sequence (G = 0: single level)
    size_t o_ops, o_exps, o_prefix, o_llpre, o_suffix, o_lsuf, o_ckpt, o_loglik, o_topo, total;
    size_t o_phi, o_nexact;   // exact-clamp routing: per-chain certificate sums psi [nchains], counter of routed sequences
    size_t o_flags;           // ... and the per-sequence verdict k_exact_select derives from them (ROUTE_*)
    size_t o_xend, o_rstart;  // alpha_hat after / R before every chain, as the scan plan's apply kernels stepped them
    size_t o_wtab, o_wlist, o_wcnt, o_dfix;   // window table [seq][WIN_STRIDE], sequences with windows, counters, loglik shifts
    size_t o_wshift;                          // per window: its log-likelihood shift and last chunk [seq][WSH_STRIDE doubles] (log alpha)
    size_t o_upi;                             // a uniform start distribution (k,q) (hmm_backward's certificate)
    size_t o_gops, o_gexps, o_gprefix, o_gllpre, o_gsuffix, o_glsuf;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
#define PLAN_WIN_STRIDE 34    // = WIN_STRIDE (window table, see k_exact_select)
#define PLAN_WSH_STRIDE 24    // = WSH_STRIDE: WIN_MAX doubles (shifts) + WIN_MAX ints (last chunks)

static int choose_T(long long NB, int L) {
    {                                                        // tuning knob, multiple of 16
        const int t = opt(HMM_OPT_CHUNK);
        if (t >= 16 && t <= MAX_T && t % 16 == 0) return t;
    }
    // enough (sequence, chunk) pairs to fill 256 CUs x 4 SIMDs in the apply kernels (16 pairs per
    // wave, two waves per SIMD: 32 768 pairs), chunks no longer than MAX_T, at least 16.
    long long t = (NB * (long long)L) / 32768;
    // ... but with few sequences that rule shreds a long sequence into tens of thousands of chunks
    // while a chunk is still a serial walk for its wave.  With the two-level chunk scans the serial
    // parts of a pass are T in-chunk steps and ~3 sqrt(L / T) scan hops of about one step's latency
    // each: T + 3 sqrt(L / T) is smallest at T = (1.5 sqrt(L))^(2/3), i.e. T^3 = 2.25 L — never go below
    // that (L = 1e6: 144, L = 1e5: 64; measured optimum for one sequence of 1e6: 128-256, for two of 1e5: 64,
    // tools/experiments/long_seq_chunks.py).
    long long tb = 16;
    while (tb * tb * tb * 4 < 9ll * L && tb < MAX_T) tb += 16;
    if (t < tb) t = tb;
    t = ((t + 16 - 1) / 16) * 16;
    if (t < 16) t = 16;
    if (t > MAX_T) t = MAX_T;
    long long lmax = (((long long)L + 16 - 1) / 16) * 16;
    if (t > lmax) t = lmax;
    return (int)t;
}

static int make_plan(int op, int k, int b, int L, int q, Plan *p, int T_fixed = 0) {
    if (k < 1 || b < 1 || L < 1 || q < 1) return HMM_ERR_BAD_SHAPE;
    if (q > QP) return HMM_ERR_Q_UNSUPPORTED;
    if ((long long)k * b > (1ll << 30) / 64) return HMM_ERR_BAD_SHAPE;
    p->k = k; p->b = b; p->L = L; p->q = q;
    p->NB = k * b;
    p->T = T_fixed ? T_fixed : choose_T(p->NB, L);
    p->C = (L + p->T - 1) / p->T;
    p->nsub = p->T / SUB;
    p->nchains = (long long)p->NB * p->C;
    p->cpw = 16;
    p->seq_start = 1;
    size_t off = 0;
    p->o_ops = off;    off = align_up(off + (size_t)p->nchains * QP * QP * sizeof(float));
    p->o_exps = off;   off = align_up(off + (size_t)p->nchains * QP * sizeof(int));
    p->o_prefix = off; off = align_up(off + (size_t)p->nchains * QP * sizeof(float));
    p->o_llpre = off;  off = align_up(off + (size_t)p->nchains * sizeof(double));
    p->o_suffix = off; off = align_up(off + (size_t)p->nchains * QP * sizeof(float));
    p->o_lsuf = off;   off = align_up(off + (size_t)p->nchains * sizeof(double));
    p->o_loglik = off; off = align_up(off + (size_t)p->NB * sizeof(double));
    p->o_topo = off;   off = align_up(off + (size_t)p->k * sizeof(int));
    p->o_phi = off;    off = align_up(off + (size_t)p->nchains * sizeof(float));
    p->o_nexact = off; off = align_up(off + sizeof(int));
    p->o_flags = off;  off = align_up(off + (size_t)p->NB * sizeof(int));
    p->o_xend = off;   off = align_up(off + (size_t)p->nchains * QP * sizeof(float));
    p->o_rstart = off; off = align_up(off + (size_t)p->nchains * QP * sizeof(float));
    p->o_wtab = off;   off = align_up(off + (size_t)p->NB * PLAN_WIN_STRIDE * sizeof(int));
    p->o_wlist = off;  off = align_up(off + (size_t)p->NB * sizeof(int));
    p->o_wcnt = off;   off = align_up(off + 4 * sizeof(int));
    p->o_dfix = off;   off = align_up(off + (size_t)p->NB * sizeof(double));
    p->o_wshift = off; off = align_up(off + ((op == HMM_OP_FORWARD || op == HMM_OP_BACKWARD) ? (size_t)p->NB * PLAN_WSH_STRIDE * sizeof(double) : 0));
    p->o_upi = off;    off = align_up(off + (size_t)p->k * p->q * sizeof(float));
    // two-level scan once the serial chain is long enough to matter (see k_scan_compose)
    p->G = 0; p->gsize = 0;
    if (p->C >= SCAN2_MIN_C) {
        int gs = 1;
        while (gs * gs < p->C) ++gs;
        p->gsize = gs;
        p->G = (p->C + gs - 1) / gs;
    }
    const size_t ng = (size_t)p->NB * (p->G > 0 ? p->G : 1);
    p->o_gops = off;    off = align_up(off + ng * QP * QP * sizeof(float));
    p->o_gexps = off;   off = align_up(off + ng * QP * sizeof(int));
    p->o_gprefix = off; off = align_up(off + ng * QP * sizeof(float));
    p->o_gllpre = off;  off = align_up(off + ng * sizeof(double));
    p->o_gsuffix = off; off = align_up(off + ng * QP * sizeof(float));
    p->o_glsuf = off;   off = align_up(off + ng * sizeof(double));
    // alpha_hat checkpoints, one QP-float row per (chain, SUB-step block), stored wave by wave:
    // [apply wave][block][chain in wave][QP] — a wave writes (and later reads) one contiguous
    // 1 KB piece per block.  Sized for the scan plan's waves and for the serial exact-clamp plan's
    // (make_xplan: one chunk of ceil(L/16)*16 steps per sequence), which reuses the region.
    p->o_ckpt = off;
    if (op == HMM_OP_POSTERIOR) {
        const size_t scan_rows = (size_t)(((long long)p->b * p->C + 15) / 16) * 16 * p->nsub;
        const size_t exact_rows = (size_t)((p->b + 15) / 16) * 16 * (size_t)(((p->L + 15) / 16) * 16 / SUB);
        off = align_up(off + (size_t)p->k * (scan_rows > exact_rows ? scan_rows : exact_rows) * QP * sizeof(float));
    }
    p->total = off;
    return HMM_OK;
}

// The plan of the serial exact-clamp kernels for the same problem: ONE chunk per sequence, walked
// by the apply kernels with the cell's exact step semantics, 16 sequences per wave.  It shares the
// scan plan's workspace (flags, log-likelihoods; its checkpoints nest inside the scan plan's
// region: NB * ceil(L/16)*2 rows <= nchains * nsub rows).  The apply kernels address a wave's
// chains with 32-bit byte offsets from the wave's base, so 16 whole sequences must span < 2 GB;
// beyond that a wave takes a single sequence.
static int make_xplan(const Plan &p, Plan *x) {
    *x = p;
    x->T = ((p.L + 15) / 16) * 16;
    x->C = 1;
    x->nsub = x->T / SUB;
    x->nchains = p.NB;
    x->G = 0; x->gsize = 0;
    const long long seq_bytes = (long long)x->T * p.q * (long long)sizeof(float);
    x->cpw = 16 * seq_bytes < (1ll << 31) - 4096 ? 16 : 1;
    if (opt(HMM_OPT_EXACT) == HMM_EXACT_ALWAYS_NARROW) x->cpw = 1;     // test hook: the very-long-sequence layout at any size
    if (seq_bytes >= (1ll << 31) - 4096) return HMM_ERR_BAD_SHAPE;      // one sequence of more than 2 GB
    return HMM_OK;
}

// ------------------------------------------------------------------ device helpers

__device__ __forceinline__ f4 mfma4(const float (&a)[4], f4 x) {
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], x.x, z, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], x.y, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], x.z, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], x.w, d, 0, 0, 0);
    return d;
}

// the same product with vector operands and an accumulator: c += Aop(a) * Bop(x)
__device__ __forceinline__ f4 mfma4v(f4 a, f4 x, f4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x.w, c, 0, 0, 0);
    return c;
}

// sum / max over the four lanes (n, n+16, n+32, n+48) that hold one tile column.
// gfx950's v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows between two
// VGPRs in the VALU: with both operands = v the two results are v's even and odd rows
// broadcast pairwise, so their sum is the xor-16 (xor-32) butterfly — no LDS round trip and
// no s_waitcnt in the recurrence's dependency chain (ds_bpermute costs two of each per step).
// Inline asm because this toolchain's __builtin_amdgcn_permlane{16,32}_swap alias their two
// results (verified wrong on hardware); "s_nop 1" covers the VALU-write -> permlane-read hazard.
__device__ __forceinline__ void swap16(float &a, float &b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float &a, float &b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float col_sum(float v) {
    float a = v, b = v;
    swap16(a, b);
    a += b; b = a;
    swap32(a, b);
    return a + b;
}
__device__ __forceinline__ float col_max(float v) {
    float a = v, b = v;
    swap16(a, b);
    a = fmaxf(a, b); b = a;
    swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float hsum(f4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float hmax(f4 v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned long long bytes) {
    unsigned n = bytes > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)n, 0x00020000);
}
// Load N consecutive rows' states 4g..4g+3 for this lane: one 16-byte buffer load per row
// at 4-byte alignment.  Elements past a row's q states are the next row's first values (or,
// at the very end of the tensor, out of the descriptor's range: gfx950 range-checks raw
// buffer accesses per dword and returns 0 there — verified on hardware); clampE() zeroes
// them either way, so no byte outside the (k,b,L,q) tensor is ever dereferenced.
template <int N>
__device__ __forceinline__ void ld_rows(__amdgpu_buffer_rsrc_t r, int voff, int rowb, f4 (&e)[N]) {
#pragma unroll
    for (int s = 0; s < N; ++s)
        e[s] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + s * rowb, 0, HMM_LD_AUX));
}

// per-lane clamp bounds: valid state -> [eps, +inf), padded state -> [0, 0]
struct Bounds { f4 lo, hi; };
__device__ __forceinline__ Bounds make_bounds(int g, int q, float eps) {
    Bounds bd;
    const float inf = __builtin_inff();
    bd.lo.x = (4 * g + 0 < q) ? eps : 0.f;  bd.hi.x = (4 * g + 0 < q) ? inf : 0.f;
    bd.lo.y = (4 * g + 1 < q) ? eps : 0.f;  bd.hi.y = (4 * g + 1 < q) ? inf : 0.f;
    bd.lo.z = (4 * g + 2 < q) ? eps : 0.f;  bd.hi.z = (4 * g + 2 < q) ? inf : 0.f;
    bd.lo.w = (4 * g + 3 < q) ? eps : 0.f;  bd.hi.w = (4 * g + 3 < q) ? inf : 0.f;
    return bd;
}
__device__ __forceinline__ f4 clampE(f4 e, const Bounds &bd) {
    f4 r;
    r.x = __builtin_amdgcn_fmed3f(e.x, bd.lo.x, bd.hi.x);
    r.y = __builtin_amdgcn_fmed3f(e.y, bd.lo.y, bd.hi.y);
    r.z = __builtin_amdgcn_fmed3f(e.z, bd.lo.z, bd.hi.z);
    r.w = __builtin_amdgcn_fmed3f(e.w, bd.lo.w, bd.hi.w);
    return r;
}
__device__ __forceinline__ f4 fmax4(f4 v, float s) {
    f4 r = {fmaxf(v.x, s), fmaxf(v.y, s), fmaxf(v.z, s), fmaxf(v.w, s)};
    return r;
}
__device__ __forceinline__ f4 sel4(bool c, f4 a, f4 b) { return c ? a : b; }

// MFMA A-operands for lane (g, n).  fwd: Aop = A^T (dst n <- src 4g+kk);  bwd: Aop = A.
__device__ __forceinline__ void load_A(const float *A, int q, int g, int n, float (&af)[4], float (&ab)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        int s = 4 * g + kk;
        bool ok = (s < q) && (n < q);
        af[kk] = ok ? A[s * q + n] : 0.f;
        ab[kk] = ok ? A[n * q + s] : 0.f;
    }
}

// ------------------------------------------------------------------ reduce

// One wave per (sequence, chunk).  ops[chain][i][k] (i = state at the chunk's last step,
// k = state just before the chunk), exps[chain][k].
__device__ __forceinline__ void reduce_chain(const float *__restrict__ A, const float *__restrict__ E,
                                             float *__restrict__ ops, int *__restrict__ exps,
                                             const int *__restrict__ topo, const Plan &p, float eps, long long chain) {
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int seq = (int)(chain / p.C), c = (int)(chain - (long long)seq * p.C);
    const int m = seq / p.b;
    if (topo && topo[m] != 0) return;      // this model's A fits a sparse topology: k_reduce_sparse has it
    const int t0 = c * p.T;
    const int len = min(p.T, p.L - t0);
    const int q = p.q;

    float af[4], ab[4];
    load_A(A + (size_t)m * q * q, q, g, n, af, ab);
    const Bounds bd = make_bounds(g, q, eps);

    const float *base = E + ((size_t)seq * p.L + t0) * q;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (unsigned long long)len * q * sizeof(float));
    const int rowb = q * (int)sizeof(float);
    int voff = g * 16;

    f4 X;
    X.x = (4 * g + 0 == n) ? 1.f : 0.f;
    X.y = (4 * g + 1 == n) ? 1.f : 0.f;
    X.z = (4 * g + 2 == n) ? 1.f : 0.f;
    X.w = (4 * g + 3 == n) ? 1.f : 0.f;
    int ex = 0;        // column n holds X[:,n] * 2^-ex

    // The operator is the exactly linear product of the steps A diag(max(E, eps)): the cell's clamp of
    // the state MIXTURE (MsaHmmCell.py:88) has no per-column form, and leaving it out is what keeps the
    // scan closest to the serial recursion (its whole deviation is the posterior mass of clamp-born
    // paths, which the apply kernels measure: see backward_body's psi).
    // rescale the column by the power of two that brings its sum into [0.5, 1): exact
    auto rescale = [&](f4 &V) {
        float s = col_sum(hsum(V));
        int xe = __builtin_amdgcn_frexp_expf(s);
        float sc = __builtin_amdgcn_ldexpf(1.0f, -xe);     // exact power of two
        V = V * sc;
        ex += xe;
    };

    int t = 0;
    if (c == 0 && p.seq_start) {   // first observation of the sequence: no transition (MsaHmmCell.py:78-79)
        f4 e0[1];
        ld_rows<1>(rs, voff, rowb, e0);
        X = X * clampE(e0[0], bd);
        rescale(X);
        voff += rowb;
        t = 1;
    }
    // software-pipelined emission stream: HMM_REDUCE_PF rows in flight ahead of the recurrence
    constexpr int PF = HMM_REDUCE_PF;
    f4 en[PF];
    ld_rows<PF>(rs, voff, rowb, en);
    for (; t < len; t += PF) {
        f4 ec[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) ec[u] = en[u];
        voff += PF * rowb;
        ld_rows<PF>(rs, voff, rowb, en);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (t + u < len) {          // wave-uniform
                X = mfma4(af, X) * clampE(ec[u], bd);
                rescale(X);
            }
        }
    }
    float *o = ops + (size_t)chain * QP * QP;
    o[(4 * g + 0) * QP + n] = X.x;
    o[(4 * g + 1) * QP + n] = X.y;
    o[(4 * g + 2) * QP + n] = X.z;
    o[(4 * g + 3) * QP + n] = X.w;
    if (g == 0) exps[(size_t)chain * QP + n] = ex;
}

// Waves walk the chains with a grid stride: when every model is served by a sparse-topology kernel
// (decided on the device) the launch costs a few thousand blocks that find nothing to do, not one
// block per four chains (27 us at 200 000 chains).
__global__ __launch_bounds__(256) void k_reduce(const float *__restrict__ A, const float *__restrict__ E,
                                                float *__restrict__ ops, int *__restrict__ exps,
                                                const int *__restrict__ topo, Plan p, float eps) {
    const long long stride = (long long)gridDim.x * 4;
    for (long long chain = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
         chain < p.nchains; chain += stride)
        reduce_chain(A, E, ops, exps, topo, p, eps, chain);
}

// ------------------------------------------------------------------ reduce, sparse topologies

// The gene-prediction models' A has 23 (15-state) / 15 (7-state) non-zeros.  On gfx950 the f32
// MFMA runs on the FP32 vector lanes (it does not overlap VALU work: measured per-step cost =
// 4 x 32 MFMA cycles + 4 x #VALU), so for a known support it is cheaper to keep one operator
// COLUMN per lane (15 registers) and do only the structural non-zero fma's: lane = (chain,
// start state), 4 chains per wave, ~70 VALU per 4 positions instead of 4 MFMA + ~33 VALU per
// position.  Results (ops/exps) have the dense kernel's format.  Which kernel serves a model is
// decided on the device from A itself (k_topo_check): the dense MFMA kernel stays the generic path.

struct TopoGene15 {      // hmm_layer/gene_pred_hmm_transitioner.py:200-221, 279-303 (k = 1)
    static constexpr int Q = 15, NE = 23, ID = 1;
    // CSR by destination state: sources of state j are src[start[j] .. start[j+1])
    static constexpr int start[Q + 1] = {0, 2, 4, 6, 8, 10, 13, 15, 16, 17, 18, 19, 20, 21, 22, 23};
    static constexpr int src[NE] = {0, 14, 8, 1, 9, 2, 10, 3, 11, 6, 7, 4, 12, 5, 13, 0, 4, 5, 6, 1, 2, 3, 5};
};
struct TopoGene7 {       // hmm_layer/gene_pred_hmm_transitioner.py:132-148
    static constexpr int Q = 7, NE = 15, ID = 2;
    static constexpr int start[Q + 1] = {0, 2, 4, 6, 8, 11, 13, 15};
    static constexpr int src[NE] = {0, 6, 4, 1, 5, 2, 6, 3, 0, 6, 3, 4, 1, 5, 2};
};

struct TopoGene29 {      // two copies of the 14 gene states around one intergenic state:
                         // hmm_layer/gene_pred_hmm_transitioner.py:263-308 (GenePredMultiHMMTransitioner, k = 2)
    static constexpr int Q = 29, NE = 45, ID = 3;
    static constexpr int start[Q + 1] = {0, 3, 5, 7, 9, 11, 13, 15, 17, 19, 22, 25, 27, 29, 30, 31, 32, 33, 34, 35, 36,
                                         37, 38, 39, 40, 41, 42, 43, 44, 45};
    static constexpr int src[NE] = {0, 27, 28, 15, 1, 16, 2, 17, 3, 18, 4, 19, 5, 20, 6, 21, 11, 22, 12, 13, 7, 23, 14,
                                    8, 24, 9, 25, 10, 26, 0, 0, 7, 8, 9, 10, 11, 12, 1, 2, 3, 4, 5, 6, 9, 10};
};

// number of edges of T that leave state i
template <class T>
__host__ __device__ constexpr int out_degree(int i) {
    int n = 0;
    for (int e = 0; e < T::NE; ++e) n += (T::src[e] == i) ? 1 : 0;
    return n;
}
// Every state of T with a single outgoing edge carries weight exactly 1 on it in A (what a softmax over
// one edge gives: the gene models' START, EI, IE and STOP states): k_reduce_sparse then never forms
// x_i = y_i * e_i for those states but feeds (e_i, y_i) straight into the successor's fma.
#define TOPO_UNIT 16
// entry (i -> j) of A lies inside topology T's support
template <class T>
__device__ __forceinline__ bool edge_in(int i, int j) {
    bool ok = false;
    for (int e = T::start[j]; e < T::start[j + 1]; ++e) ok = ok || (T::src[e] == i);
    return ok;
}

// topo[m] = ID of the sparse topology that contains the support of A[m], 0 = none (dense kernel),
// TOPO_EXACT = the chunked scan must not serve this model at all.
// One wave per model, lanes over the q*q entries (a single thread walking them took 32 us).
//
// TOPO_EXACT.  The cell clamps the predicted state mixture at eps every step
// (hmm_layer/MsaHmmCell.py:87-88); chunk operators can only floor each conditional column, which is
// the same thing up to O(eps) as long as no state's mass ever lives on the floor alone.  When the
// support of A (entries > eps: a smaller entry cannot lift its target above the clamp) is not
// PRIMITIVE — reducible chains, A = I, periodic chains, states without incoming edges, all-zero
// rows as in the reference's as-shipped matrices (hmm_layer/Transitioner.py:366-367) — forward
// and backward evidence can contradict each other outright and the answer is decided by the floors:
// those models are served by the serial exact-clamp kernels.  Primitive <=> B^n > 0 for every
// n >= (q-1)^2 + 1 (Wielandt); 8 boolean squarings give B^256, enough for q <= 16.
#define TOPO_EXACT 255
__global__ __launch_bounds__(64) void k_topo_check(const float *__restrict__ A, int *__restrict__ topo, int k, int q,
                                                   int force_dense, int exact_mode, float eps, int *__restrict__ nexact,
                                                   int *__restrict__ wcnt = nullptr) {
    const int m = blockIdx.x;
    const float *Am = A + (size_t)m * q * q;
    if (m == 0 && threadIdx.x == 0) *nexact = 0;
    if (m == 0 && threadIdx.x < 4 && wcnt) wcnt[threadIdx.x] = 0;
    bool exact = exact_mode == HMM_EXACT_ALWAYS || exact_mode == HMM_EXACT_ALWAYS_NARROW;
    if (exact_mode == HMM_EXACT_AUTO) {
        int row = 0;                                            // lane i < q: row i of the support as a bit mask
        if ((int)threadIdx.x < q)
            for (int j = 0; j < q; ++j) row |= (Am[threadIdx.x * q + j] > eps) ? (1 << j) : 0;
        for (int it = 0; it < 8; ++it) {
            int nr = 0;
            for (int j = 0; j < q; ++j) {
                const int rj = __builtin_amdgcn_readlane(row, j);
                nr |= ((row >> j) & 1) ? rj : 0;
            }
            row = nr;
        }
        const bool notfull = (int)threadIdx.x < q && row != (1 << q) - 1;
        exact = __ballot(notfull) != 0ull;
    }
    if (exact) {
        if (threadIdx.x == 0) topo[m] = TOPO_EXACT;
        return;
    }
    bool bad15 = q != TopoGene15::Q, bad7 = q != TopoGene7::Q;
    bool nonunit15 = false, nonunit7 = false;      // an edge out of a single-successor state that is not exactly 1
    if (!force_dense && (!bad15 || !bad7))
        for (int e = threadIdx.x; e < q * q; e += 64) {
            const float a = Am[e];
            const int i = e / q, j = e - i * q;
            if (!bad15) {
                const bool in = edge_in<TopoGene15>(i, j);
                bad15 = a != 0.f && !in;
                nonunit15 = nonunit15 || (in && a != 1.0f && out_degree<TopoGene15>(i) == 1);
            }
            if (!bad7) {
                const bool in = edge_in<TopoGene7>(i, j);
                bad7 = a != 0.f && !in;
                nonunit7 = nonunit7 || (in && a != 1.0f && out_degree<TopoGene7>(i) == 1);
            }
        }
    // (lanes over the entries: a single lane walking the topology's edges for the unit test took 10 of the
    // kernel's 20 us — a tenth of a whole log-likelihood call at b = 256 x L = 1e4)
    const bool any15 = __ballot(bad15) != 0ull, any7 = __ballot(bad7) != 0ull;
    const bool nu15 = __ballot(nonunit15) != 0ull, nu7 = __ballot(nonunit7) != 0ull;
    if (threadIdx.x == 0) {
        int id = force_dense ? 0 : (!any15 ? TopoGene15::ID : (!any7 ? TopoGene7::ID : 0));
        if (id == TopoGene15::ID && !nu15) id |= TOPO_UNIT;
        if (id == TopoGene7::ID && !nu7) id |= TOPO_UNIT;
        topo[m] = id;
    }
}

#define SP_TILE 16     // steps staged per LDS tile

#ifndef HMM_RS_AHEAD
#define HMM_RS_AHEAD 0     // sparse reduce: read each step's emission row from LDS one step ahead (A/B: the
                           // registers are worth more as a fourth wave per SIMD: 2.10 -> 2.00 ms with HMM_RS_WPE 4)
#endif
#ifndef HMM_RS_UNI
#define HMM_RS_UNI 1       // sparse reduce: wave-uniform transition weights in SGPRs when the wave holds one model
#endif
#ifndef HMM_POST_BLOCK
#define HMM_POST_BLOCK 16  // block length (checkpoint spacing) of hmm_posterior's scan-plan pair in probability mode
#endif
#ifndef HMM_RS_SUM2
#define HMM_RS_SUM2 1      // sparse reduce: the column sum (and the rescale test) every second step
#endif
#ifndef HMM_RS_WPE
#define HMM_RS_WPE 4       // waves per SIMD the register allocator of the 16-lane sparse reduce is held to (0: its own choice)
#endif
#if HMM_RS_WPE
#define RS_ATTR __attribute__((amdgpu_waves_per_eu(HMM_RS_WPE, HMM_RS_WPE)))
#else
#define RS_ATTR
#endif
// Lane layout of the sparse reduce: W lanes per chain (the padded state count: 16, or 32 for the 29-state
// model), 64 / W chains per wave.  One wave's staging: [buffer][chain in wave][step x W floats (+pad)].
template <class T> struct RsCfg {
    static constexpr int W = T::Q <= 16 ? 16 : 32;
    static constexpr int CPW = 64 / W;
    static constexpr int PIECES = SP_TILE * T::Q / 4;        // 16-byte pieces of one chain's 16-step tile
    static constexpr int RPC = (PIECES + 63) / 64;           // load rounds per chain per tile
    typedef float Lds[2][CPW][SP_TILE * W + 16];
};

// UNIT: every single-out-edge state of T has weight 1 on its edge (TOPO_UNIT, for all of the wave's chains)
// UNI:  all chains of the wave belong to one model (always, unless k > 1 and the wave straddles two): the
//       transition weights are then wave-uniform and live in SGPRs — 15 to 23 vector registers that the column
//       and the emission row need (held to 128 VGPRs for four waves per SIMD, the kernel spilled 26 dwords)
template <class T, bool UNIT, bool UNI>
__device__ __forceinline__ void reduce_sparse_wave(const float *__restrict__ A, const float *__restrict__ E,
                                                   float *__restrict__ ops, int *__restrict__ exps, const Plan &p,
                                                   float eps, typename RsCfg<T>::Lds &ldsw, long long wchain0,
                                                   bool mine, int m, long long chain, int c) {
    constexpr int Q = T::Q, W = RsCfg<T>::W, CPW = RsCfg<T>::CPW, RPC = RsCfg<T>::RPC;
    const int lane = threadIdx.x & 63, cl = lane / W, kc = lane % W;
    const int t0 = c * p.T;
    const int len = mine ? min(p.T, p.L - t0) : 0;
    const bool first = (c == 0) && p.seq_start;

    // transition weights of this lane's model, one per structural non-zero
    float a[T::NE];
    {
        const float *Am = A + (size_t)(UNI ? __builtin_amdgcn_readfirstlane(m) : m) * Q * Q;
#pragma unroll
        for (int j = 0; j < Q; ++j)
#pragma unroll
            for (int e = T::start[j]; e < T::start[j + 1]; ++e) a[e] = Am[T::src[e] * Q + j];
    }

    // ---- emission staging: the wave's 4 chains are adjacent in memory; each chain's 16-step
    // tile is 16*Q contiguous floats = (16*Q*4/16) 16-byte pieces, one per lane
    const int seq0 = (int)(wchain0 / p.C), c0 = (int)(wchain0 - (long long)seq0 * p.C);
    const long long row0 = (long long)seq0 * p.L + (long long)c0 * p.T;
    const unsigned long long total = (unsigned long long)p.NB * p.L * Q * sizeof(float);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(E + row0 * Q, total - (unsigned long long)row0 * Q * sizeof(float));
    constexpr int PIECES = RsCfg<T>::PIECES;               // 60 for Q = 15, 28 for Q = 7, 116 for Q = 29
    static_assert(SP_TILE * Q % 4 == 0, "tile must be a whole number of 16-byte pieces");
    int coff[CPW];                                          // byte offset of chain j's chunk from the wave base
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        long long ch = wchain0 + j;
        if (ch >= p.nchains) ch = wchain0;
        const long long sq = ch / p.C;
        const long long rw = sq * p.L + (ch - sq * p.C) * (long long)p.T;
        coff[j] = (int)((rw - row0) * Q * (long long)sizeof(float));
    }
    int loff[RPC][4];                                       // where this lane's 4 floats of a piece go in a tile
#pragma unroll
    for (int rr = 0; rr < RPC; ++rr)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = 4 * (rr * 64 + lane) + u;
            loff[rr][u] = (idx / Q) * W + (idx % Q);
        }
    for (int i = lane; i < 2 * CPW * (SP_TILE * W + 16); i += 64) (&ldsw[0][0][0])[i] = 0.f;   // pad columns = 0

    f4 r[CPW * RPC];
    auto fetch = [&](int tile) {
#pragma unroll
        for (int j = 0; j < CPW; ++j)
#pragma unroll
            for (int rr = 0; rr < RPC; ++rr)
                r[j * RPC + rr] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs, coff[j] + tile * (SP_TILE * Q * 4) + (rr * 64 + lane) * 16, 0, 0));
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int rr = 0; rr < RPC; ++rr) {
            if (rr * 64 + lane < PIECES) {
#pragma unroll
                for (int j = 0; j < CPW; ++j) {
                    float *dst = &ldsw[buf][j][0];
                    const f4 v = r[j * RPC + rr];
                    dst[loff[rr][0]] = fmaxf(v.x, eps);     // the cell's max(E, eps), once per value
                    dst[loff[rr][1]] = fmaxf(v.y, eps);
                    dst[loff[rr][2]] = fmaxf(v.z, eps);
                    dst[loff[rr][3]] = fmaxf(v.w, eps);
                }
            }
        }
    };

    // x[j]: the column.  UNIT: for a state j with a single outgoing edge x[j] holds the PRE-emission value
    // y_j and pend[j] the emission it still has to be multiplied with (the true entry is x[j] * pend[j]):
    // its only consumer multiplies by a weight of 1, so (pend[j], x[j]) go straight into that fma and the
    // product is never formed — 8 of the 15 emission multiplies of the 15-state model.
    float x[Q], pend[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) { x[j] = (j == kc) ? 1.f : 0.f; pend[j] = 1.f; }
    int ex = 0;
    float cs = (kc < Q) ? 1.f : 0.f;     // sum of the column as currently scaled
    auto deferred = [](int j) constexpr { return UNIT && out_degree<T>(j) == 1; };

    // exact power-of-two rescale of the column (sum back into [0.5, 1)).  The exponent is clamped
    // so that the factor cannot overflow when a column has underflowed to a denormal or to zero
    // (a start state from which the chunk is impossible even through the eps clamps: its weight
    // in the scan is then 0, as it should be).
    // risk: the column's sum has been found below 2^-100 at a rescale — it has then been through the denormal range
    // (two observations in a row that every path survives only at the emission floor take 2^-106 off between two sums)
    // and its entries are no longer good to fp32's precision relative to each other.  Nothing to do with the clamps of
    // the state mixture, so psi does not see it: the chain is marked (pad lane of its exponent row) and k_exact_select
    // treats it as flagged.
    bool risk = false;
    auto rescale = [&]() {
        risk = risk || (kc < Q && cs < 0x1p-100f);
        int xe = max(__builtin_amdgcn_frexp_expf(cs), -100);
        float sc = __builtin_amdgcn_ldexpf(1.0f, -xe);
#pragma unroll
        for (int j = 0; j < Q; ++j) x[j] *= sc;
        cs *= sc;
        ex += xe;
    };

    // finalise and store this lane's operator column (called once, at the lane's last step)
    auto column_sum = [&]() {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < Q; ++j) s += deferred(j) ? x[j] * pend[j] : x[j];
        return s;
    };
    auto finish = [&]() {
        cs = kc < Q ? column_sum() : 0.f;            // (the running sum may be a step old)
        rescale();
        float *o = ops + (size_t)chain * W * W;
#pragma unroll
        for (int j = 0; j < Q; ++j) o[j * W + kc] = (kc < Q) ? (deferred(j) ? x[j] * pend[j] : x[j]) : 0.f;
#pragma unroll
        for (int j = Q; j < W; ++j) o[j * W + kc] = 0.f;
        static_assert(Q < W, "the exponent row's last lane is a pad lane");
        const unsigned long long rb = __builtin_amdgcn_ballot_w64(risk) >> (W == 64 ? 0 : W * cl);
        const bool chain_risk = (W == 64 ? rb : (rb & ((1ull << (W & 63)) - 1ull))) != 0ull;
        exps[(size_t)chain * W + kc] = (kc < Q) ? ex : ((kc == W - 1 && chain_risk) ? 1 : 0);
    };
    // one recurrence step on the column: x <- max(E,eps) * (A^T x): exactly linear.  (Rounds 1-2 added
    // eps * sum(x) to every state, a per-column stand-in for the cell's clamp of the state mixture; on the
    // gene models' own emissions that floor costs more accuracy than it buys — the floored scan is off by
    // eps * sum_t 1/<alpha_hat_t, R_t>, up to 1.6e-4 in a posterior for peaked class probabilities, the
    // floor-free one by the posterior mass of clamp-born paths, 1e-8 .. 1e-7 there; tools/experiments/cert_study.py.)
    // e: the clamped emission row of this step, already in registers
    // SUM (compile-time): also form the column's sum and rescale if it has become small; the callers ask for it
    // every second step — the sum is 15 of a step's 48 instructions and nothing but the rescaling needs it since
    // the operators carry no floor.  From the threshold two steps of the smallest emissions a realistic path
    // sees (2^-32: a free state's 1/4096 times a class probability of 1e-6) stay normal numbers; a column that
    // loses more than that in two steps has met two impossible observations and weighs nothing.
    auto step = [&](const float (&e)[Q], auto sumc) {
        constexpr bool SUM = decltype(sumc)::value;
        float y[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int ed = T::start[j]; ed < T::start[j + 1]; ++ed) {
                const int i = T::src[ed];
                const float w = deferred(i) ? pend[i] : a[ed];               // weight 1: the pending emission instead
                acc = (ed == T::start[j]) ? w * x[i] : fmaf(w, x[i], acc);
            }
            y[j] = acc;
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            if (SUM) s = fmaf(y[j], e[j], s);
            if (deferred(j)) { x[j] = y[j]; pend[j] = e[j]; }
            else x[j] = y[j] * e[j];
        }
        if (SUM) {
            // (32-lane rows — the 29-state model, whose one-directional entry points have no certificate of their own:
            // a column that loses more than 2^-45 between two sums has met an observation that everything survives at
            // the emission floor only; such sequences go to the serial kernels, k32_select)
            if (W >= 32) {                      // ... EVERY column of the chain, whatever the start state
                const unsigned long long keep = __builtin_amdgcn_ballot_w64(kc < Q && !(s < cs * 0x1p-45f));
                const unsigned long long mine = W == 64 ? ~0ull : (((1ull << (W & 63)) - 1ull) << (W * cl));
                risk = risk || (keep & mine) == 0ull;
            }
            cs = s;
            // rescale when any column of the wave has shrunk below 2^-40 (wave-uniform branch; every
            // ~4th step at gene-model emission magnitudes, every ~12th for E ~ 0.5).  0 < cs < 2^-40 as ONE
            // unsigned compare on the bit pattern (cs >= 0; 0 wraps to the top): the ballot is then the compare.
            const unsigned cb = __builtin_bit_cast(unsigned, cs) - 1u;
            if (__builtin_amdgcn_ballot_w64(cb < 0x2B800000u - 1u) != 0) rescale();
        }
    };

    // exactly Q floats of a staged row: whole 16-byte reads plus one 12- / 8- / 4-byte read.  (Reading
    // the pad too leaves a dead destination register that the allocator hands out again at once: the
    // prefetch then has to be waited for on the spot.)
    auto ldrow = [&](const float *rowp, float (&r)[Q]) {
#pragma unroll
        for (int v = 0; v + 4 <= Q; v += 4) {
            const f4 t = *reinterpret_cast<const f4 *>(rowp + v);
            r[v] = t.x; r[v + 1] = t.y; r[v + 2] = t.z; r[v + 3] = t.w;
        }
        constexpr int R0 = Q & ~3;
        if constexpr (Q - R0 == 3) {
            typedef float f3 __attribute__((ext_vector_type(3)));
            const f3 t = *reinterpret_cast<const f3 *>(rowp + R0);
            r[R0] = t.x; r[R0 + 1] = t.y; r[R0 + 2] = t.z;
        } else if constexpr (Q - R0 == 2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 t = *reinterpret_cast<const f2 *>(rowp + R0);
            r[R0] = t.x; r[R0 + 1] = t.y;
        } else if constexpr (Q - R0 == 1) {
            r[R0] = rowp[R0];
        }
    };

    // The main loop is NOT predicated per lane (a per-lane `if` becomes 15 selects per step):
    // every lane runs to the longest chunk of the wave — rows past a lane's own chunk are
    // finite, clamped emissions of whatever follows — and each lane's result is captured at its
    // own last step.  Only a wave that holds the tail chunk of a sequence (len < T) needs to look
    // for that step as it goes (CHECK, a wave-uniform branch per step); every other wave finishes
    // all its lanes after the last tile.  Each step's emission row is read from LDS one step ahead
    // of its use, so the LDS latency sits under the previous step's ~55 VALU instructions.
    const int ntiles = p.T / SP_TILE;
    const int last = len - 1;                       // -1 for lanes that are not ours
    const bool ragged = __builtin_amdgcn_ballot_w64(mine && len != p.T) != 0;
    auto tile_steps = [&](auto checked, int tile, int buf) {
        constexpr bool CHECK = decltype(checked)::value;
        const float *tp = &ldsw[buf][cl][0];
        // (the 29-state column already fills the register file: no look-ahead copy of the next row there)
        constexpr bool AHEAD = W == 16 && HMM_RS_AHEAD;
        float c[Q], nx[AHEAD ? Q : 1];
        ldrow(tp, c);
        if constexpr (AHEAD) ldrow(tp + W, nx);
        if (tile == 0) {
            // step 0 of a sequence's first chunk has no transition (MsaHmmCell.py:78-79):
            // X = diag(E_0); everyone else takes the generic step
            float xs[Q];
#pragma unroll
            for (int j = 0; j < Q; ++j) xs[j] = x[j];
            step(c, std::true_type());
            if (first) {
                const float e0 = tp[kc < Q ? kc : 0];
#pragma unroll
                for (int j = 0; j < Q; ++j) { x[j] = xs[j] * e0; pend[j] = 1.f; }      // xs = unit column kc
                cs = (kc < Q) ? e0 : 0.f;
                ex = 0;
                rescale();
            }
        } else {
            step(c, std::integral_constant<bool, !HMM_RS_SUM2>());
        }
        if (CHECK) { if (__builtin_amdgcn_ballot_w64(tile * SP_TILE == last) != 0) { if (tile * SP_TILE == last) finish(); } }
        // steps 1 .. SP_TILE - 1: the odd ones (the tile's last one among them) form the column sum
        auto one = [&](int sidx, auto sumc) {
            if constexpr (AHEAD) {
#pragma unroll
                for (int u = 0; u < Q; ++u) c[u] = nx[u];
                if (sidx + 1 < SP_TILE) ldrow(tp + (sidx + 1) * W, nx);
            } else {
                ldrow(tp + sidx * W, c);
            }
            step(c, sumc);
            if (CHECK) {
                const int t = tile * SP_TILE + sidx;
                if (__builtin_amdgcn_ballot_w64(t == last) != 0) { if (t == last) finish(); }
            }
        };
#pragma unroll
        for (int sidx = 1; sidx < SP_TILE; sidx += 2) {
            one(sidx, std::true_type());
            if (sidx + 1 < SP_TILE) one(sidx + 1, std::integral_constant<bool, !HMM_RS_SUM2>());
        }
    };
    fetch(0);
    if (ragged) {
        for (int tile = 0; tile < ntiles; ++tile) {
            stage(tile & 1);
            if (tile + 1 < ntiles) fetch(tile + 1);
            tile_steps(std::true_type(), tile, tile & 1);
        }
    } else {
        for (int tile = 0; tile < ntiles; ++tile) {
            stage(tile & 1);
            if (tile + 1 < ntiles) fetch(tile + 1);
            tile_steps(std::false_type(), tile, tile & 1);
        }
        if (mine) finish();
    }
}

// MIXED: the launch for waves that straddle two models (k > 1 only); every other wave belongs to the main launch
template <class T, bool MIXED>
__device__ __forceinline__ void reduce_sparse_block(const float *__restrict__ A, const float *__restrict__ E,
                                                    float *__restrict__ ops, int *__restrict__ exps,
                                                    const int *__restrict__ topo, const Plan &p, float eps) {
    // [wave][buffer][chain in wave][step][16 floats: one clamped emission row, 64-byte stride]
    // chain images are 1 KB; +16 floats of padding puts the 4 chains of a wave on different banks
    __shared__ __attribute__((aligned(16))) typename RsCfg<T>::Lds lds[4];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, cl = lane / RsCfg<T>::W;
    const long long wchain0 = ((long long)blockIdx.x * 4 + w) * RsCfg<T>::CPW;          // first chain of the wave
    if (wchain0 >= p.nchains) return;
    const long long chain = wchain0 + cl;
    const bool inrange = chain < p.nchains;
    const long long chn = inrange ? chain : wchain0;
    const int seq = (int)(chn / p.C), c = (int)(chn - (long long)seq * p.C);
    const int m = seq / p.b;
    const int tp = topo[m];
    const bool mine = inrange && (tp & (TOPO_UNIT - 1)) == T::ID && tp != TOPO_EXACT;
    // wave-uniform early out when none of the four chains is ours
    const unsigned long long mmask = __builtin_amdgcn_ballot_w64(mine);
    if (mmask == 0) return;
    // the unit-weight variant when every chain of the wave that is ours qualifies (models can mix in a wave)
    const bool unit = __builtin_amdgcn_ballot_w64(mine && (tp & TOPO_UNIT)) == mmask;
    const bool uni = HMM_RS_UNI && __builtin_amdgcn_ballot_w64(m != __builtin_amdgcn_readfirstlane(m)) == 0ull;
    if (uni == MIXED) return;                    // the other launch has this wave
    if (unit) reduce_sparse_wave<T, true, !MIXED>(A, E, ops, exps, p, eps, lds[w], wchain0, mine, m, chain, c);
    else reduce_sparse_wave<T, false, !MIXED>(A, E, ops, exps, p, eps, lds[w], wchain0, mine, m, chain, c);
}

// 16 lanes per chain (7 / 15 states): held to HMM_RS_WPE waves per SIMD
template <class T, bool MIXED = false>
__global__ __launch_bounds__(256) RS_ATTR void k_reduce_sparse(const float *__restrict__ A, const float *__restrict__ E,
                                                       float *__restrict__ ops, int *__restrict__ exps,
                                                       const int *__restrict__ topo, Plan p, float eps) {
    reduce_sparse_block<T, MIXED>(A, E, ops, exps, topo, p, eps);
}
// 32 lanes per chain (29 states): the column alone takes ~230 registers, two waves per SIMD
template <class T, bool MIXED = false>
__global__ __launch_bounds__(256) void k_reduce_sparse_wide(const float *__restrict__ A, const float *__restrict__ E,
                                                            float *__restrict__ ops, int *__restrict__ exps,
                                                            const int *__restrict__ topo, Plan p, float eps) {
    reduce_sparse_block<T, MIXED>(A, E, ops, exps, topo, p, eps);
}

// ------------------------------------------------------------------ scan

// 16-lane (DPP row) all-reduce and lane broadcast without LDS: ds_bpermute shuffles made every
// hop of the scan ~1.6 us of pure latency (24 dependent shuffles); these are plain VALU ops.
template <class Op>
__device__ __forceinline__ int row_allreduce_i(int v, Op op) {
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}
__device__ __forceinline__ float row_sum_f(float x) {
    int v = __builtin_bit_cast(int, x);
    auto add = [](int a, int b) { return __builtin_bit_cast(int, __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b)); };
    return __builtin_bit_cast(float, row_allreduce_i(v, add));
}
__device__ __forceinline__ int row_max_i(int x) {
    return row_allreduce_i(x, [](int a, int b) { return a > b ? a : b; });
}
__device__ __forceinline__ float lane_bcast(float x, int srclane) {      // srclane wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), srclane));
}
// lane K of this lane's 16-lane row, in every lane of the row (DPP row_newbcast): four independent
// 16-lane problems per wave
template <int K>
__device__ __forceinline__ float row_bcast_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + K, 0xF, 0xF, false));
}
// sum_k c[k] * (lane k of v's row)
__device__ __forceinline__ float row_dot16(const float (&c)[16], float v) {
    float a0 = 0.f, a1 = 0.f;
    a0 = fmaf(c[0], row_bcast_f<0>(v), a0);   a1 = fmaf(c[1], row_bcast_f<1>(v), a1);
    a0 = fmaf(c[2], row_bcast_f<2>(v), a0);   a1 = fmaf(c[3], row_bcast_f<3>(v), a1);
    a0 = fmaf(c[4], row_bcast_f<4>(v), a0);   a1 = fmaf(c[5], row_bcast_f<5>(v), a1);
    a0 = fmaf(c[6], row_bcast_f<6>(v), a0);   a1 = fmaf(c[7], row_bcast_f<7>(v), a1);
    a0 = fmaf(c[8], row_bcast_f<8>(v), a0);   a1 = fmaf(c[9], row_bcast_f<9>(v), a1);
    a0 = fmaf(c[10], row_bcast_f<10>(v), a0); a1 = fmaf(c[11], row_bcast_f<11>(v), a1);
    a0 = fmaf(c[12], row_bcast_f<12>(v), a0); a1 = fmaf(c[13], row_bcast_f<13>(v), a1);
    a0 = fmaf(c[14], row_bcast_f<14>(v), a0); a1 = fmaf(c[15], row_bcast_f<15>(v), a1);
    return a0 + a1;
}

// One 128-thread block per sequence: wave 0 runs the forward prefix chain, wave 1 the backward
// suffix chain (separate waves so the two dependent chains run concurrently), 16 lanes each.
// C hops per chain, every hop a 16x16 mat-vec.
__global__ __launch_bounds__(128) void k_scan(const float *__restrict__ pi, const float *__restrict__ ops,
                                             const int *__restrict__ exps, float *__restrict__ prefix,
                                             double *__restrict__ llpre, float *__restrict__ suffix,
                                             double *__restrict__ lsuf, double *__restrict__ loglik,
                                             const int *__restrict__ topo, Plan p, float eps,
                                             const float *__restrict__ pre_in = nullptr,
                                             const double *__restrict__ ll_in = nullptr,
                                             const float *__restrict__ suf_in = nullptr,
                                             const double *__restrict__ ls_in = nullptr) {
    // pre_in / ll_in / suf_in / ls_in (sequence-sharded calls): the vectors entering this time slab
    // from the slabs before / after it, in place of the start distribution and of ones
    const int seq = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15;
    const int q = p.q, C = p.C;
    const int m = seq / p.b;
    const size_t chain0 = (size_t)seq * C;
    if (lane >= 16) return;
    if (topo[m] == TOPO_EXACT) return;          // served by the serial exact-clamp kernels
    if (dir == 0) {
        float praw = pre_in ? pre_in[(size_t)seq * QP + n] : ((n < q) ? pi[(size_t)m * q + n] : 0.f);
        float a = pre_in ? praw : ((n < q) ? fmaxf(praw, eps) : 0.f);
        double ll = ll_in ? ll_in[seq] : 0.0;
        prefix[chain0 * QP + n] = praw;
        if (n == 0) llpre[chain0] = ll;
        // operator rows are prefetched one hop ahead: the hop itself is ~100 cycles of math, a
        // dependent 1 KB load per hop would make the scan a chain of C memory round trips
        const f4 *row = reinterpret_cast<const f4 *>(ops + chain0 * QP * QP + n * QP);
        f4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
        int xe = exps[chain0 * QP + n];
        for (int c = 0; c < C; ++c) {
            const f4 q0 = r0, q1 = r1, q2 = r2, q3 = r3;
            const int xec = xe;
            if (c + 1 < C) {
                const f4 *nx = reinterpret_cast<const f4 *>(ops + (chain0 + c + 1) * QP * QP + n * QP);
                r0 = nx[0]; r1 = nx[1]; r2 = nx[2]; r3 = nx[3];
                xe = exps[(chain0 + c + 1) * QP + n];
            }
            int we = (a > 0.f) ? __builtin_amdgcn_frexp_expf(a) + xec : -(1 << 28);
            const int emax = row_max_i(we);
            int sh = xec - emax;
            sh = sh < -300 ? -300 : sh;
            float w = __builtin_amdgcn_ldexpf(a, sh);
            float acc = 0.f;
            float xr[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = fmaf(xr[kk], lane_bcast(w, kk), acc);
            const float S = row_sum_f(acc);
            a = acc / S;
            ll += (double)__logf(S) + (double)emax * LN2;
            if (c + 1 < C) {
                prefix[(chain0 + c + 1) * QP + n] = a;
                if (n == 0) llpre[chain0 + c + 1] = ll;
            }
        }
        if (n == 0) loglik[seq] = ll;
    } else {
        float v = suf_in ? suf_in[(size_t)seq * QP + n] : ((n < q) ? 1.f : 0.f);
        double lb = ls_in ? ls_in[seq] : 0.0;
        // column n of the operator (stride QP), prefetched one hop ahead like the forward chain
        float col[16];
        int xe = 0;
        if (C > 1) {
            const float *X = ops + (chain0 + C - 1) * QP * QP;
#pragma unroll
            for (int j = 0; j < 16; ++j) col[j] = X[j * QP + n];
            xe = exps[(chain0 + C - 1) * QP + n];
        }
        for (int c = C - 1; c >= 0; --c) {
            suffix[(chain0 + c) * QP + n] = v;
            if (n == 0) lsuf[chain0 + c] = lb;
            if (c == 0) break;
            float cc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) cc[j] = col[j];
            const int xec = xe;
            if (c - 1 > 0) {
                const float *X = ops + (chain0 + c - 1) * QP * QP;
#pragma unroll
                for (int j = 0; j < 16; ++j) col[j] = X[j * QP + n];
                xe = exps[(chain0 + c - 1) * QP + n];
            }
            float u = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) u = fmaf(cc[j], lane_bcast(v, j), u);
            int we = (u > 0.f) ? __builtin_amdgcn_frexp_expf(u) + xec : -(1 << 28);
            const int emax = row_max_i(we);
            int sh = xec - emax;
            sh = sh < -300 ? -300 : sh;
            v = __builtin_amdgcn_ldexpf(u, sh);
            lb += (double)emax * LN2;
        }
    }
}

// ---- two-level chunk scan.  The hops above are linear (no clamps at this level), so operators
// compose exactly up to rounding: groups of ~sqrt(C) chunks are composed in parallel
// (k_scan_compose), k_scan itself runs over the G group operators, and k_scan_inner walks the
// chunks of every group in parallel from the group's entry vectors.  Serial depth ~3 sqrt(C) hops
// instead of C (196 -> 42 at b = 1024 x L = 1e5; 1954 -> 135 for one sequence of 1e6).
__device__ __forceinline__ int col_max_i(int v) {       // max over the four lanes of a tile column
    float a = __builtin_bit_cast(float, v), b2 = a;
    swap16(a, b2);
    v = max(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b2));
    a = __builtin_bit_cast(float, v); b2 = a;
    swap32(a, b2);
    return max(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b2));
}

// one wave per (sequence, group): X <- Op_c X over the group's chunks, X = identity at the start.
// Tile layout of k_reduce: lane (g, n) holds rows 4g..4g+3 of column n, column n scaled by 2^-ex.
__global__ __launch_bounds__(256) void k_scan_compose(const float *__restrict__ ops, const int *__restrict__ exps,
                                                      float *__restrict__ gops, int *__restrict__ gexps,
                                                      const int *__restrict__ topo, Plan p) {
    const long long wv = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv >= (long long)p.NB * p.G) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int seq = (int)(wv / p.G), grp = (int)(wv - (long long)seq * p.G);
    if (topo[seq / p.b] == TOPO_EXACT) return;
    const int c0 = grp * p.gsize, c1 = min(p.C, c0 + p.gsize);
    f4 X = {4 * g + 0 == n ? 1.f : 0.f, 4 * g + 1 == n ? 1.f : 0.f, 4 * g + 2 == n ? 1.f : 0.f, 4 * g + 3 == n ? 1.f : 0.f};
    int ex = 0;
    // A-operand: lane (g, i = n) supplies Op_c[i][4g + kk]; exponents of the contraction rows 4g + r.
    // The operands of the next SCAN_PF hops are in flight during this one.
    const size_t chb = (size_t)seq * p.C;
    f4 an[SCAN_PF];
    i4 en[SCAN_PF];
#pragma unroll
    for (int u = 0; u < SCAN_PF; ++u) {
        const int c = min(c0 + u, c1 - 1);
        an[u] = *reinterpret_cast<const f4 *>(ops + (chb + c) * QP * QP + n * QP + 4 * g);
        en[u] = *reinterpret_cast<const i4 *>(exps + (chb + c) * QP + 4 * g);
    }
    for (int cb = c0; cb < c1; cb += SCAN_PF)
#pragma unroll
      for (int u = 0; u < SCAN_PF; ++u) {
        const int c = cb + u;
        if (c >= c1) break;                                   // wave-uniform
        const f4 a4 = an[u];
        const i4 e4 = en[u];
        if (c + SCAN_PF < c1) {
            an[u] = *reinterpret_cast<const f4 *>(ops + (chb + c + SCAN_PF) * QP * QP + n * QP + 4 * g);
            en[u] = *reinterpret_cast<const i4 *>(exps + (chb + c + SCAN_PF) * QP + 4 * g);
        }
        // align the rows of X to a common exponent per column
        int we = -(1 << 28);
        we = X.x > 0.f ? max(we, __builtin_amdgcn_frexp_expf(X.x) + e4.x) : we;
        we = X.y > 0.f ? max(we, __builtin_amdgcn_frexp_expf(X.y) + e4.y) : we;
        we = X.z > 0.f ? max(we, __builtin_amdgcn_frexp_expf(X.z) + e4.z) : we;
        we = X.w > 0.f ? max(we, __builtin_amdgcn_frexp_expf(X.w) + e4.w) : we;
        const int emax = col_max_i(we);
        f4 W;
        W.x = __builtin_amdgcn_ldexpf(X.x, max(e4.x - emax, -300));
        W.y = __builtin_amdgcn_ldexpf(X.y, max(e4.y - emax, -300));
        W.z = __builtin_amdgcn_ldexpf(X.z, max(e4.z - emax, -300));
        W.w = __builtin_amdgcn_ldexpf(X.w, max(e4.w - emax, -300));
        const float af[4] = {a4.x, a4.y, a4.z, a4.w};
        X = mfma4(af, W);
        // bring the column sum back into [0.5, 1): exact power of two
        const float sden = col_sum(hsum(X));
        const int xe = __builtin_amdgcn_frexp_expf(sden);
        X = X * __builtin_amdgcn_ldexpf(1.0f, -xe);
        ex += emax + xe;
    }
    float *o = gops + (size_t)wv * QP * QP;
    o[(4 * g + 0) * QP + n] = X.x;
    o[(4 * g + 1) * QP + n] = X.y;
    o[(4 * g + 2) * QP + n] = X.z;
    o[(4 * g + 3) * QP + n] = X.w;
    if (g == 0) gexps[(size_t)wv * QP + n] = ex;
}

// one block per (sequence, group): the hops of k_scan over the group's chunks, started from the
// group's entry vectors (gprefix / gllpre forward, gsuffix / glsuf backward) that k_scan left.
__global__ __launch_bounds__(128) void k_scan_inner(const float *__restrict__ ops, const int *__restrict__ exps,
                                                   const float *__restrict__ gprefix, const double *__restrict__ gllpre,
                                                   const float *__restrict__ gsuffix, const double *__restrict__ glsuf,
                                                   float *__restrict__ prefix, double *__restrict__ llpre,
                                                   float *__restrict__ suffix, double *__restrict__ lsuf,
                                                   const int *__restrict__ topo, Plan p, float eps) {
    // wave 0 of the block: forward prefixes, wave 1: backward suffixes; a 16-lane row per (sequence, group),
    // FOUR of them per wave (row sums / maxima and the row broadcast are DPP operations inside the row), the
    // loops run to the longest row's count
    const int lane = threadIdx.x & 63;
    const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15;
    const int q = p.q;
    const long long blk = (long long)blockIdx.x * 4 + (lane >> 4);
    bool valid = blk < (long long)p.NB * p.G;
    const long long bl = valid ? blk : 0;
    const int seq = (int)(bl / p.G), grp = (int)(bl - (long long)seq * p.G);
    valid = valid && topo[seq / p.b] != TOPO_EXACT;
    const int c0 = grp * p.gsize, c1 = min(p.C, c0 + p.gsize);
    const size_t chain0 = (size_t)seq * p.C;
    int hops = valid ? c1 - 1 - c0 : 0;
    for (int o = 32; o > 0; o >>= 1) hops = max(hops, __shfl_xor(hops, o));
    if (dir == 0) {
        const float pin = valid ? gprefix[(size_t)bl * QP + n] : 0.f;
        // the sequence's very first vector is the raw start distribution: clamped for the recursion,
        // stored raw (the apply kernel clamps it itself), as in k_scan
        float a = (grp == 0 && p.seq_start) ? ((n < q) ? fmaxf(pin, eps) : 0.f) : pin;
        double ll = valid ? gllpre[bl] : 0.0;
        if (valid) {
            prefix[(chain0 + c0) * QP + n] = pin;
            if (n == 0) llpre[chain0 + c0] = ll;
        }
        float rw[SCAN_PF][16];
        int xe[SCAN_PF];
        auto fetch = [&](int u, int c) {                      // row n of operator c, exponent of source state n -> slot u
            const f4 *row = reinterpret_cast<const f4 *>(ops + (chain0 + c) * QP * QP + n * QP);
            const f4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
            rw[u][0] = r0.x; rw[u][1] = r0.y; rw[u][2] = r0.z; rw[u][3] = r0.w; rw[u][4] = r1.x; rw[u][5] = r1.y;
            rw[u][6] = r1.z; rw[u][7] = r1.w; rw[u][8] = r2.x; rw[u][9] = r2.y; rw[u][10] = r2.z; rw[u][11] = r2.w;
            rw[u][12] = r3.x; rw[u][13] = r3.y; rw[u][14] = r3.z; rw[u][15] = r3.w;
            xe[u] = exps[(chain0 + c) * QP + n];
        };
#pragma unroll
        for (int u = 0; u < SCAN_PF; ++u) {
            xe[u] = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) rw[u][k] = 0.f;
            if (valid && c0 + u + 1 < c1) fetch(u, c0 + u);
        }
        for (int hb = 0; hb < hops; hb += SCAN_PF)
#pragma unroll
          for (int u = 0; u < SCAN_PF; ++u) {
            const int h = hb + u;
            if (h >= hops) break;                             // wave-uniform
            const int c = c0 + h;
            const bool on = valid && c + 1 < c1;
            float cur[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) cur[k] = rw[u][k];
            const int xec = xe[u];
            if (valid && c + SCAN_PF + 1 < c1) fetch(u, c + SCAN_PF);
            int we = (a > 0.f) ? __builtin_amdgcn_frexp_expf(a) + xec : -(1 << 28);
            const int emax = row_max_i(we);
            int sh = xec - emax;
            sh = sh < -300 ? -300 : sh;
            const float w = __builtin_amdgcn_ldexpf(a, sh);
            const float acc = row_dot16(cur, w);
            const float S = row_sum_f(acc);
            if (on) {
                a = acc / S;
                ll += (double)__logf(S) + (double)emax * LN2;
                prefix[(chain0 + c + 1) * QP + n] = a;
                if (n == 0) llpre[chain0 + c + 1] = ll;
            }
        }
    } else {
        float v = valid ? gsuffix[(size_t)bl * QP + n] : 0.f;
        double lb = valid ? glsuf[bl] : 0.0;
        float col[SCAN_PF][16];
        int xe[SCAN_PF];
        auto fetch = [&](int u, int c) {                      // column n of operator c -> slot u
            const float *X = ops + (chain0 + c) * QP * QP;
#pragma unroll
            for (int j = 0; j < 16; ++j) col[u][j] = X[j * QP + n];
            xe[u] = exps[(chain0 + c) * QP + n];
        };
        if (valid) {
            suffix[(chain0 + c1 - 1) * QP + n] = v;
            if (n == 0) lsuf[chain0 + c1 - 1] = lb;
        }
#pragma unroll
        for (int u = 0; u < SCAN_PF; ++u) {
            xe[u] = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) col[u][j] = 0.f;
            if (valid && c1 - 1 - u > c0) fetch(u, c1 - 1 - u);
        }
        for (int hb = 0; hb < hops; hb += SCAN_PF)
#pragma unroll
          for (int u = 0; u < SCAN_PF; ++u) {
            const int h = hb + u;
            if (h >= hops) break;                             // wave-uniform
            const int c = c1 - 1 - h;                         // through operator c: the suffix of chunk c - 1
            const bool on = valid && c > c0;
            float cc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) cc[j] = col[u][j];
            const int xec = xe[u];
            if (valid && c - SCAN_PF > c0) fetch(u, c - SCAN_PF);
            const float un = row_dot16(cc, v);
            int we = (un > 0.f) ? __builtin_amdgcn_frexp_expf(un) + xec : -(1 << 28);
            const int emax = row_max_i(we);
            int sh = xec - emax;
            sh = sh < -300 ? -300 : sh;
            if (on) {
                v = __builtin_amdgcn_ldexpf(un, sh);
                lb += (double)emax * LN2;
                suffix[(chain0 + c - 1) * QP + n] = v;
                if (n == 0) lsuf[chain0 + c - 1] = lb;
            }
        }
    }
}

// ------------------------------------------------------------------ apply (shared pieces)

struct Tile {                 // what one wave of an apply kernel works on: 16 chains
    long long wave;           // index of the wave in its plan
    long long chain;          // this lane's chain (column n)
    bool valid;
    bool first;               // chunk 0 of its sequence
    int len;                  // steps in this chain's chunk (0 if !valid)
    int voff;                 // byte offset of element (t0, 4g) relative to the wave base
    __amdgpu_buffer_rsrc_t rsE;
    const float *baseE;       // wave base pointer (for the matching output base)
};

__device__ __forceinline__ Tile make_tile(const float *E, const Plan &p, long long wave, int g, int n,
                                          int *model, long long *wchain0) {
    // waves never straddle models: each model owns ceil(b*C/16) waves
    const long long per_model = (long long)p.b * p.C;
    const long long wpm = (per_model + p.cpw - 1) / p.cpw;
    const int m = (int)(wave / wpm);
    const long long w = wave - (long long)m * wpm;
    const long long c0 = (long long)m * per_model + w * p.cpw;       // first chain of the wave
    Tile tl;
    tl.wave = wave;
    long long rel = w * p.cpw + n;
    tl.valid = n < p.cpw && rel < per_model;
    tl.chain = c0 + (tl.valid ? n : 0);
    const long long seq = tl.chain / p.C;
    const int c = (int)(tl.chain - seq * p.C);
    const long long seq0 = c0 / p.C;
    const int cc0 = (int)(c0 - seq0 * p.C);
    tl.first = (c == 0) && p.seq_start;
    tl.len = tl.valid ? min(p.T, p.L - c * p.T) : 0;
    const long long row0 = seq0 * p.L + (long long)cc0 * p.T;        // wave base row
    const long long row = seq * p.L + (long long)c * p.T;
    tl.voff = (int)((row - row0) * p.q * (long long)sizeof(float)) + g * 16;
    const unsigned long long total = (unsigned long long)p.NB * p.L * p.q * sizeof(float);
    const unsigned long long boff = (unsigned long long)row0 * p.q * sizeof(float);
    tl.baseE = E + row0 * p.q;
    tl.rsE = make_rsrc(tl.baseE, total - boff);
    *model = m;
    *wchain0 = c0;
    return tl;
}

// ---- output staging.  A wave produces, per step, 16 rows (one per chain) of q floats spread
// over its lanes as 16-byte pieces; storing those directly touches 16 different cache lines
// with 16/12-byte fragments per instruction and made the posterior kernel store-bound
// (3.9 ms with, 2.1 ms without its stores).  Instead the rows are collected in wave-private LDS —
// per chain one contiguous run of rows, exactly its image in HBM — and flushed as full 16-byte
// pieces of contiguous memory, OUT_ROWS rows of every chain at a time.  The size of those bursts
// is what the HBM system rewards in a read + write stream (tools/experiments/stream_pattern.hip, the
// kernels' own access pattern with no arithmetic: 480-byte visits per chain 3.97 TB/s, 960-byte
// 4.74, 1920-byte 5.11); 32 rows of 60 bytes are also exactly 15 cache lines, so the groups of a
// line-aligned chunk share no line with their neighbours.
#ifndef HMM_OUT_ROWS
#define HMM_OUT_ROWS 16                                // rows per chain staged in LDS per flush (multiple of SUB)
#endif
#define OUT_GB (HMM_OUT_ROWS / SUB)                    // apply blocks per flush group
#define OUT_STRIDE (HMM_OUT_ROWS * QP + 4)             // floats per chain in LDS (16-byte multiple)
#define OUT_SEG (16 * OUT_STRIDE + 32)                 // floats per wave: the staged rows + the chain table

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
struct __attribute__((packed, aligned(4))) P4 { float a, b, c, d; };
struct __attribute__((packed, aligned(4))) P3 { float a, b, c; };
struct __attribute__((packed, aligned(4))) P2 { float a, b; };

struct OutStage {
    float *seg;                 // this wave's LDS region: 16 chains x OUT_STRIDE floats, then the chain table
    char *base;                 // global base pointer of the wave (same origin as the E descriptor)
    int q;
};

// chain table at the end of the segment: [c] byte offset of chain c's chunk start, [16 + c] its valid steps
__device__ __forceinline__ OutStage make_outstage(float *seg, char *base, int q, int lane, int voff0, int len) {
    OutStage o;
    o.seg = seg; o.base = base; o.q = q;
    int *tab = reinterpret_cast<int *>(seg + 16 * OUT_STRIDE);
    if (lane < 16) { tab[lane] = voff0; tab[16 + lane] = len; }        // lane c (g = 0) owns chain c
    __builtin_amdgcn_wave_barrier();
    return o;
}

// put states 4g..4g+3 of chain n's row `s` (relative to the flush group) into the stage
__device__ __forceinline__ void stage_row(const OutStage &o, int n, int g, int s, f4 v) {
    float *p = o.seg + n * OUT_STRIDE + s * o.q + 4 * g;
    const int nv = o.q - 4 * g;
    if (nv >= 1) p[0] = v.x;
    if (nv >= 2) p[1] = v.y;
    if (nv >= 3) p[2] = v.z;
    if (nv >= 4) p[3] = v.w;
}

// write the staged rows row0 .. row0+nrows-1 (chunk-relative; nrows a multiple of 4, at most
// OUT_ROWS) of every chain to global memory: consecutive lanes take consecutive 16-byte pieces
__device__ __forceinline__ void flush_rows(const OutStage &o, int lane, int row0, int nrows) {
    const int ppc = nrows * o.q / 4;                                   // pieces per chain
    const float inv = 1.0f / (float)ppc;
    const int *tab = reinterpret_cast<const int *>(o.seg + 16 * OUT_STRIDE);
    __builtin_amdgcn_wave_barrier();
    for (int pc = lane; pc < 16 * ppc; pc += 64) {
        const int c = (int)(((float)pc + 0.5f) * inv);                 // pc / ppc, exact for these ranges
        const int kk = pc - c * ppc;
        int rows = tab[16 + c] - row0;
        rows = rows > nrows ? nrows : rows;
        const int nfl = rows * o.q - 4 * kk;                           // floats of this piece that exist
        if (nfl <= 0) continue;
        const f4 v = *reinterpret_cast<const f4 *>(o.seg + c * OUT_STRIDE + 4 * kk);
        char *dst = o.base + tab[c] + (row0 * o.q + 4 * kk) * (int)sizeof(float);
#if HMM_NT_STORE
        // the output is never read back by these kernels: streaming stores leave L2 to the emission rows and
        // checkpoints (k_backward 2.97 -> 2.86 ms; the same hint on the emission LOADS costs 20-45 %)
        if (nfl >= 4) { __builtin_nontemporal_store(v, reinterpret_cast<f4u *>(dst)); }
#else
        if (nfl >= 4) { P4 t = {v.x, v.y, v.z, v.w}; *reinterpret_cast<P4 *>(dst) = t; }
#endif
        else if (nfl == 3) { P3 t = {v.x, v.y, v.z}; *reinterpret_cast<P3 *>(dst) = t; }
        else if (nfl == 2) { P2 t = {v.x, v.y}; *reinterpret_cast<P2 *>(dst) = t; }
        else { *reinterpret_cast<float *>(dst) = v.x; }
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ f4 log4(f4 v) {
    f4 r = {__logf(v.x), __logf(v.y), __logf(v.z), __logf(v.w)};
    return r;
}

// ---- coalesced emission loads.  In the tile layout the four lanes that hold one chain's row
// window (64 bytes) are 16 lanes apart, so a row load is 64 separate 16-byte L1 accesses
// (TCP_TOTAL_CACHE_ACCESSES: 501 M per k_forward launch against 111 M for the sparse reduce reading
// the same tensor).  Loader layout instead: lane -> (chain lane >> 2, piece lane & 3), adjacent
// lanes on adjacent 16-byte pieces, and the block is turned into the tile layout through
// wave-private LDS (one ds_write_b128 + one ds_read_b128 per row; chains IN_STRIDE floats apart
// puts both patterns on distinct banks).
#define IN_STRIDE (SUB * QP + 4)

__device__ __forceinline__ int loader_voff(const Tile &tl, int lane) {
    // byte offset of (chain lane >> 2, first row, piece lane & 3): lane c (g = 0) owns chain c's offset
    return __shfl(tl.voff, lane >> 2) + 16 * (lane & 3);
}
// rows[] (loader layout) -> e[] (tile layout: states 4g..4g+3 of chain n), through `seg`
template <int BS>
__device__ __forceinline__ void permute_rows(float *seg, int lane, int g, int n, const f4 (&rows)[BS], f4 (&e)[BS]) {
    constexpr int STRIDE = BS * QP + 4;
    float *wr = seg + (lane >> 2) * STRIDE + 4 * (lane & 3);
#pragma unroll
    for (int s = 0; s < BS; ++s) *reinterpret_cast<f4 *>(wr + s * QP) = rows[s];
    __builtin_amdgcn_wave_barrier();
    const float *rd = seg + n * STRIDE + 4 * g;
#pragma unroll
    for (int s = 0; s < BS; ++s) e[s] = *reinterpret_cast<const f4 *>(rd + s * QP);
    __builtin_amdgcn_wave_barrier();
}

// one exact forward cell step on the tile: X <- normalise(max(E,eps) * max(X A, eps))
__device__ __forceinline__ f4 fwd_step(const float (&af)[4], f4 X, f4 e, bool init, float eps, float *Sout) {
    f4 D = mfma4(af, X);
    f4 R = fmax4(sel4(init, X, D), eps);
    f4 sf = R * e;
    float S = col_sum(hsum(sf));
    float inv = __builtin_amdgcn_rcpf(S);
    *Sout = S;
    return sf * inv;
}

// ------------------------------------------------------------------ forward apply

// The apply kernels serve three plans.
//   KIND_SCAN   a wave owns 16 (sequence, chunk) pairs and starts from the chunk scan's prefix / suffix
//               vectors; models routed to the serial path (topo[m] == TOPO_EXACT) are skipped.
//   KIND_EXACT  (make_xplan) one chunk = the whole sequence, 16 sequences per wave, started from pi / ones
//               exactly as the cell's get_initial_state does (hmm_layer/MsaHmmCell.py:114-119) — the serial
//               recursion of the reference, step for step; only sequences whose verdict in `flags` is
//               ROUTE_WHOLE are computed and written.
//   KIND_WIN    a wave owns the windows (runs of chunks) of ONE flagged sequence, one window per tile
//               column: the same serial recursion over the window only, started from the exact vectors
//               the scan plan's kernels left at the window's two ends (see k_window_posterior).
//
// Which sequences leave the scan plan is decided on the device from psi, the posterior mass of
// clamp-born paths that the scan plan's backward kernel sums per chunk (backward_body); every serial
// kernel of a call reads the same verdict (k_exact_select): deterministic.
#define KIND_SCAN 0
#define KIND_EXACT 1
#define KIND_WIN 2
#define EXACT_DELTA 2e-6f       // a tenth of the posteriors' stated tolerance (2e-5)
#define ROUTE_NONE 0
#define ROUTE_WINDOWS 1
#define ROUTE_WHOLE 2
#define WIN_MAX 16                      // windows per sequence = tile columns (more: the whole sequence is redone)
#define WIN_STRIDE (2 * WIN_MAX + 2)    // ints per sequence in the window table: count, spare, (first chunk, chunks | lead << 24) pairs
#define WIN_LEAD (1 << 24)              // the window's forward pass starts one chunk early (at the flagged chunk itself)
#define WSH_STRIDE (WIN_MAX + WIN_MAX / 2)  // doubles per sequence in the shift table: WIN_MAX shifts, then WIN_MAX last chunks (ints)
static_assert(WSH_STRIDE == PLAN_WSH_STRIDE && WIN_STRIDE == PLAN_WIN_STRIDE, "plan strides");
static_assert(WIN_STRIDE == PLAN_WIN_STRIDE, "window table stride");
#define WIN_TOL 2e-6f                   // a window is accepted when its far-end vectors meet the scan plan's to this
#define WIN_MARGIN_STEPS 192            // how far past a flagged chunk a window reaches (rounded up to chunks)
struct Routing {
    const int *topo;
    int exact_mode;
    int *nexact;          // counter of routed sequences (kernels without a k_exact_select count themselves), or null
    const int *flags;     // per-sequence verdict (ROUTE_*) of k_exact_select, or null: per-model routing only
};
template <bool EXACT>
__device__ __forceinline__ bool route_tile(Tile &tl, int m, int g, const Routing &rt, float eps) {
    if (!EXACT) return rt.topo[m] != TOPO_EXACT;                     // wave-uniform: waves never straddle models
    bool need = tl.valid && rt.topo[m] == TOPO_EXACT;                // chain == sequence in the exact plan
    if (rt.flags) need = tl.valid && rt.flags[tl.chain] == ROUTE_WHOLE;
    tl.valid = need;
    tl.len = need ? tl.len : 0;
    if (rt.nexact && need && g == 0) atomicAdd(rt.nexact, 1);
    return __builtin_amdgcn_ballot_w64(need) != 0ull;
}

// checkpoint row of (this wave, block 0, chain n), states 4g..4g+3; block j is j * ckpt_block(p) floats on
__device__ __forceinline__ size_t ckpt_origin(const Tile &tl, const Plan &p, int g, int n) {
    return (((size_t)tl.wave * p.nsub) * p.cpw + n) * QP + 4 * g;
}
__device__ __forceinline__ size_t ckpt_block(const Plan &p) { return (size_t)p.cpw * QP; }

// a checkpoint row (read exactly once, by the wave that wrote it)
__device__ __forceinline__ f4 ld_ckpt(const float *ptr) {
#if HMM_CK_LD_NT
    return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(ptr));
#else
    return *reinterpret_cast<const f4 *>(ptr);
#endif
}

// states 4g..4g+3 of a q-vector in the tile layout (0 beyond q)
__device__ __forceinline__ f4 ld_state4(const float *v, int q, int g) {
    f4 r;
    r.x = 4 * g + 0 < q ? v[4 * g + 0] : 0.f;
    r.y = 4 * g + 1 < q ? v[4 * g + 1] : 0.f;
    r.z = 4 * g + 2 < q ? v[4 * g + 2] : 0.f;
    r.w = 4 * g + 3 < q ? v[4 * g + 3] : 0.f;
    return r;
}
__device__ __forceinline__ f4 ones4(int q, int g) {
    f4 r = {4 * g + 0 < q ? 1.f : 0.f, 4 * g + 1 < q ? 1.f : 0.f, 4 * g + 2 < q ? 1.f : 0.f, 4 * g + 3 < q ? 1.f : 0.f};
    return r;
}
__device__ __forceinline__ f4 abs4(f4 v) {
    f4 r = {__builtin_fabsf(v.x), __builtin_fabsf(v.y), __builtin_fabsf(v.z), __builtin_fabsf(v.w)};
    return r;
}
// max(u, eps) with the sign bit set where the clamp was active (u <= eps): the flag costs one instruction more
// than the clamp itself
__device__ __forceinline__ f4 clamp_flag4(f4 u, float eps) {
    f4 r = {u.x > eps ? u.x : -eps, u.y > eps ? u.y : -eps, u.z > eps ? u.z : -eps, u.w > eps ? u.w : -eps};
    return r;
}
// component-wise with |.| as source modifiers (the packed multiply has none: written per component on purpose)
__device__ __forceinline__ f4 mul_abs4(f4 a, f4 b) {
    f4 r = {__builtin_fabsf(a.x) * b.x, __builtin_fabsf(a.y) * b.y, __builtin_fabsf(a.z) * b.z, __builtin_fabsf(a.w) * b.w};
    return r;
}
__device__ __forceinline__ float hsum_abs(f4 v) {
    return (__builtin_fabsf(v.x) + __builtin_fabsf(v.y)) + (__builtin_fabsf(v.z) + __builtin_fabsf(v.w));
}
// sum over the flagged (negative) components of a of |a| * w
__device__ __forceinline__ float flagged_dot(f4 a, f4 w) {
    float s = fmaxf(-a.x, 0.f) * w.x;
    s = fmaf(fmaxf(-a.y, 0.f), w.y, s);
    s = fmaf(fmaxf(-a.z, 0.f), w.z, s);
    return fmaf(fmaxf(-a.w, 0.f), w.w, s);
}
// sum of the negative components' magnitudes
__device__ __forceinline__ float hsum_neg(f4 v) {
    return (fmaxf(-v.x, 0.f) + fmaxf(-v.y, 0.f)) + (fmaxf(-v.z, 0.f) + fmaxf(-v.w, 0.f));
}

// WRITE_CKPT: alpha_hat entering every SUB-step block -> ck + j * ckb (posterior pipeline)
// WRITE_LOGA: log alpha -> out (forward_recursion)
// X: the vector entering the tile's first step (raw pi for a sequence's first chunk); ll0: the log-likelihood
//    up to there (log alpha output)
// KIND_EXACT / KIND_WIN also accumulate the log-likelihood of the steps walked, which is what the function returns
//    (every lane of a tile column holds it): the product of the normalisers c_t in fp64 with its exponent split off
//    once per block and ONE logarithm at the end — a window's value replaces the chunk scan's for its span, so it
//    has to be good to better than the clamp-born mass it accounts for (1e-6); per-step fp32 logarithms were not.
// Xend: alpha_hat after the tile's last step — after the last block (KIND_SCAN: meaningful for full chunks, the
//    only ones anybody reads), or captured at the block that ends the column's own window (KIND_WIN)
// CERT (scan plan of hmm_forward, which has no backward pass to sum psi in): the part of alpha_hat that descends
//    from the forward cell's clamps INSIDE this chunk is carried along as a second vector (Fv: the same step,
//    four more MFMAs) and weighed at the chunk's last position with the chunk scan's suffix vector there:
//    *cert = <Fv, suffix> / <alpha_hat, suffix> = the posterior mass of the paths born in this chunk — a path's
//    posterior mass is the same wherever it is measured, so the sum over a sequence's chunks is psi's forward half,
//    which is all that the log-likelihood and log alpha depend on.  WRITE_LOGA adds the clamp-born share of
//    alpha_hat itself at the chunk's end (log alpha is a statement about the filtered vector).
// BS: steps per block = checkpoint spacing (BS; the posterior's scan-plan pair may use 16)
template <bool WRITE_CKPT, bool WRITE_LOGA, int KIND, bool CERT = false, int BS = SUB>
__device__ __forceinline__ double forward_body(const float *__restrict__ A, const float *__restrict__ E, f4 X, double ll0,
                                             float *__restrict__ ck, size_t ckb, float *__restrict__ out,
                                             const Tile &tl, int m, float *seg, const Plan &p, float eps,
                                             f4 *Xend = nullptr, float *cert = nullptr, const float *sufv = nullptr) {
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int q = p.q;
    float af[4], ab[4];
    load_A(A + (size_t)m * q * q, q, g, n, af, ab);
    const Bounds bd = make_bounds(g, q, eps);
    const int rowb = q * (int)sizeof(float);
    constexpr bool ACC = WRITE_LOGA || KIND != KIND_SCAN;

    // `seg`: one LDS segment per wave: input permutation, and (log alpha) output staging — used in turn
    OutStage os;
    if (WRITE_LOGA)
        os = make_outstage(seg, reinterpret_cast<char *>(out + (tl.baseE - E)), q, lane, tl.voff - g * 16, tl.len);
    double llb = ll0;                                                 // log-likelihood up to the current block
    double dm = 1.0;                                                  // product of the normalisers, mantissa / exponent
    int de = 0;
    const f4 zero4c = {0.f, 0.f, 0.f, 0.f};
    f4 Fv = zero4c, Xc = X, Fc = zero4c;                              // CERT: clamp-born part of X; both at the chain's last step
    float shmax = 0.f;
    // the coalesced loader layout permutes through the LDS segment, which the log alpha variant
    // needs for its staged rows: that variant loads in the tile layout
    constexpr bool COAL = HMM_COALESCE_F && !WRITE_LOGA;
    int voff = COAL ? loader_voff(tl, lane) : tl.voff;
    f4 xe = X;

    // the next block's emission rows are in flight while the current block is computed
    // (the serial plans run one or two waves per SIMD with nothing else to hide a load behind: three blocks in flight)
    constexpr bool PF2 = HMM_FWD_PF2 || KIND != KIND_SCAN;
    f4 en[BS], en2[BS], en3[BS];
    ld_rows<BS>(tl.rsE, voff, rowb, en);
    if (PF2) {
        ld_rows<BS>(tl.rsE, voff + BS * rowb, rowb, en2);
        ld_rows<BS>(tl.rsE, voff + 2 * BS * rowb, rowb, en3);
    }
    for (int j = 0; j < p.nsub; ++j) {
#ifdef HMM_NT_CKPT
        if (WRITE_CKPT && tl.valid && j * BS < tl.len) __builtin_nontemporal_store(X, reinterpret_cast<f4 *>(ck + (size_t)j * ckb));
#else
        if (WRITE_CKPT && tl.valid && j * BS < tl.len) *reinterpret_cast<f4 *>(ck + (size_t)j * ckb) = X;
#endif
        f4 e[BS];
        if (COAL) {
            permute_rows<BS>(seg, lane, g, n, en, e);
        } else {
#pragma unroll
            for (int s = 0; s < BS; ++s) e[s] = en[s];
        }
        if (PF2) {
#pragma unroll
            for (int s = 0; s < BS; ++s) { en[s] = en2[s]; en2[s] = en3[s]; }
            if (j + 3 < p.nsub) ld_rows<BS>(tl.rsE, voff + 3 * BS * rowb, rowb, en3);
        } else {
            if (j + 1 < p.nsub) ld_rows<BS>(tl.rsE, voff + BS * rowb, rowb, en);
        }
        float lacc = 0.f;
#pragma unroll
        for (int s = 0; s < BS; ++s) {
            float S;
            if (CERT) {
                const bool init = tl.first && j == 0 && s == 0;
                const f4 ec = clampE(e[s], bd);
                const f4 D = mfma4(af, X), Df = mfma4(af, Fv);
                const f4 sf = fmax4(sel4(init, X, D), eps) * ec;
                f4 Rb = {D.x > eps ? Df.x : eps, D.y > eps ? Df.y : eps, D.z > eps ? Df.z : eps, D.w > eps ? Df.w : eps};
                Rb = sel4(init, Fv, Rb);                    // (the start distribution's own clamp: the scan has it)
                S = col_sum(hsum(sf));
                const float inv = __builtin_amdgcn_rcpf(S);
                X = sf * inv;
                Fv = Rb * ec * inv;
                const bool last = j * BS + s + 1 == tl.len;
                Xc = sel4(last, X, Xc);
                Fc = sel4(last, Fv, Fc);
            } else {
                X = fwd_step(af, X, clampE(e[s], bd), tl.first && j == 0 && s == 0, eps, &S);
            }
            if (WRITE_LOGA) {
                lacc += (j * BS + s < tl.len) ? __logf(S) : 0.f;
                float base = (float)(llb + (double)lacc);
                stage_row(os, n, g, (j % (HMM_OUT_ROWS / BS)) * BS + s, log4(X) + base);
            }
            if (KIND != KIND_SCAN) dm *= (j * BS + s < tl.len) ? (double)S : 1.0;
        }
        if (WRITE_LOGA) llb += (double)lacc;
        if (KIND != KIND_SCAN) { de += __builtin_amdgcn_frexp_exp(dm); dm = __builtin_amdgcn_frexp_mant(dm); }
        if (KIND == KIND_WIN) xe = sel4((j + 1) * BS == tl.len, X, xe);
        if (WRITE_LOGA && ((j + 1) % (HMM_OUT_ROWS / BS) == 0 || j + 1 == p.nsub))
            flush_rows(os, lane, (j / (HMM_OUT_ROWS / BS)) * HMM_OUT_ROWS, (j % (HMM_OUT_ROWS / BS) + 1) * BS);
        voff += BS * rowb;
    }
    if (Xend) *Xend = KIND == KIND_WIN ? xe : X;
    if (CERT) {
        const f4 sv = *reinterpret_cast<const f4 *>(sufv + 4 * g);
        const float num = col_sum(hsum(Fc * sv)), den = col_sum(hsum(Xc * sv));
        // log alpha: also the clamp-born share of alpha_hat itself where the chunk hands over to the next one (what
        // was born and forgotten inside the chunk the in-chunk steps have exactly; what is still there at the end is
        // missing from the next chunk's prefix)
        if (WRITE_LOGA) shmax = col_sum(hsum(Fc));
        // ... and what that share becomes one step on, under the NEXT observation (the first row of the chunk after,
        // whose kernel starts from the chunk scan's prefix vector: alpha_hat without its clamp-born part) — the mirror
        // image of backward_body's shnext
        float shnext = 0.f;
        if (WRITE_LOGA) {
            f4 en1 = {0.f, 0.f, 0.f, 0.f};
            const bool more = tl.valid && tl.chain % p.C != p.C - 1;             // (not the sequence's last chunk)
            if (more) {
                f4 r1[1];
                ld_rows<1>(tl.rsE, tl.voff + tl.len * rowb, rowb, r1);
                en1 = clampE(r1[0], bd);
            }
            const f4 Dx = fmax4(mfma4(af, Xc), eps), Df = mfma4(af, Fc);
            const float dn = col_sum(hsum(en1 * Dx));
            shnext = (more && dn > 0.f) ? col_sum(hsum(en1 * Df)) * __builtin_amdgcn_rcpf(dn) : 0.f;
        }
        *cert = fmaxf(fmaxf(num * __builtin_amdgcn_rcpf(den), shmax), shnext);
    }
    // the serial kernels of every entry point return the same value for the same sequence
    if (KIND != KIND_SCAN) return ll0 + log(dm) + (double)de * LN2;
    return llb;
}

// xend (scan plan, posterior pipeline): alpha_hat after every chain's last step, [chain][QP] — what a window of
// the serial recomputation starts from and is checked against
// CERT: psi[chain] = the chunk's clamp-born posterior mass (see forward_body), weighed with suffix[chain]
template <bool WRITE_CKPT, bool WRITE_LOGA, bool EXACT, bool CERT = false, int BS = SUB>
__global__ __launch_bounds__(256) void k_forward(const float *__restrict__ A, const float *__restrict__ pi,
                                                 const float *__restrict__ E,
                                                 const float *__restrict__ prefix, const double *__restrict__ llpre,
                                                 float *__restrict__ ckpt, float *__restrict__ out,
                                                 double *__restrict__ loglik, float *__restrict__ xend, Routing rt,
                                                 Plan p, float eps, long long nwaves, float *__restrict__ psi = nullptr,
                                                 const float *__restrict__ suffix = nullptr) {
    const long long wave = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    int m; long long wc0;
    Tile tl = make_tile(E, p, wave, g, n, &m, &wc0);
    if (!route_tile<EXACT>(tl, m, g, rt, eps)) return;
    // LDS per wave: the staged log alpha rows, or only the input permutation's block
    constexpr int SEG = WRITE_LOGA ? OUT_SEG : 16 * (BS * QP + 4);
    __shared__ __attribute__((aligned(16))) float ostage[4 * SEG];
    float *seg = ostage + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * SEG;
    const f4 X0 = EXACT ? ld_state4(pi + (size_t)m * p.q, p.q, g)
                        : *reinterpret_cast<const f4 *>(prefix + (size_t)tl.chain * QP + 4 * g);
    const double ll0 = (!EXACT && WRITE_LOGA) ? llpre[tl.chain] : 0.0;
    float *ck = ckpt + ckpt_origin(tl, p, g, n);
    f4 xe;
    float cert = 0.f;
    const double ll = forward_body<WRITE_CKPT, WRITE_LOGA, EXACT ? KIND_EXACT : KIND_SCAN, CERT, BS>(
        A, E, X0, ll0, ck, ckpt_block(p), out, tl, m, seg, p, eps, &xe, &cert,
        CERT ? suffix + (size_t)tl.chain * QP : nullptr);
    if (EXACT && tl.valid && g == 0) loglik[tl.chain] = ll;
    if (!EXACT && xend && tl.valid) *reinterpret_cast<f4 *>(xend + (size_t)tl.chain * QP + 4 * g) = xe;
    if (CERT && tl.valid && g == 0) psi[tl.chain] = cert;
}

// ------------------------------------------------------------------ backward apply

// MODE 0: gamma, 1: log gamma, 2: log gamma + loglik, 3: log beta (no forward part)
//
// psi (KIND_SCAN, MODE < 3): per chain, the posterior mass of CLAMP-BORN paths.  The chunk operators are the
// exactly linear products of A diag(E_t); the cell additionally lifts every component of the predicted state
// mixture to eps (hmm_layer/MsaHmmCell.py:87-88) in both directions.  The mass a clamp creates at (t, j) is
// part of alpha_hat_t[j] (forward cell) or of R_t[j] (reverse cell); paths through it carry the posterior weight
// gamma_t[j] at most, and that weight is the same at every position of the sequence (a path's posterior mass does
// not depend on where it is measured).  So psi = sum_t sum_{j: a prediction into (t, j) was clamped} gamma_t[j]
// bounds, for the WHOLE sequence, how far the serial recursion's posteriors and log-likelihood can be from the
// clamp-free scan's — and on the fp64 model the bound is attained (tools/experiments/cert_study.py: |d gamma| = psi
// to two digits).  The flags travel in the sign bits of the recomputed alpha_hat and of R, so the sign of their
// product marks a component that exactly one of the two clamps touched (one touched by both weighs eps^2 against
// sums of order eps at least: nothing), and psi's share of a step is the sum of the negative products: ~20 VALU
// per step in a kernel that waits for memory.
// Rend: R after the tile's first position has been walked = the vector leaving the chunk before.
// CERT3 (scan plan of hmm_backward, MODE 3: no forward part to take gamma from): the mirror image of forward_body's
// CERT — the part of R that descends from the reverse cell's clamps inside this chunk is carried along (Gv) and
// weighed at the chunk's FIRST position with alpha_hat there, one forward step from the chunk scan's prefix vector
// (prev): psi[chain] = <alpha_hat, Gv> / <alpha_hat, R> there, or the clamp-born share of the backward vector
// itself where the chunk hands over to the one before, if that is larger (log beta is a statement about that vector).
template <int MODE, int KIND, bool CERT3 = false, int BS = SUB>
__device__ __forceinline__ void backward_body(const float *__restrict__ A, const float *__restrict__ E, f4 Rv, double lbb0,
                                              float llf, const float *__restrict__ ck, size_t ckb,
                                              float *__restrict__ out, float *__restrict__ psi, const Tile &tl, int m,
                                              float *seg, const Plan &p, float eps, f4 *Rend = nullptr,
                                              const float *prev = nullptr, double *lbb_end = nullptr) {
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int q = p.q;
    float af[4], ab[4];
    load_A(A + (size_t)m * q * q, q, g, n, af, ab);
    const Bounds bd = make_bounds(g, q, eps);
    const int rowb = q * (int)sizeof(float);
    const OutStage os = make_outstage(seg, reinterpret_cast<char *>(out + (tl.baseE - E)), q, lane,
                                      tl.voff - g * 16, tl.len);
    constexpr bool PSI = HMM_PSI && KIND == KIND_SCAN && MODE != 3;

    double lbb = lbb0;                                                // log scale of beta after the current block
    float ps = 0.f;
    if (MODE != 2) llf = 0.f;
    const f4 zero4g = {0.f, 0.f, 0.f, 0.f};
    f4 Gv = zero4g, Rc = Rv, Gc = zero4g, ec0 = zero4g;               // CERT3: clamp-born part of R; R, Gv, E at the first position
    float shmax = 0.f;

#if HMM_COALESCE_B
    const int lvoff = loader_voff(tl, lane);
#else
    const int lvoff = tl.voff;
#endif
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 Xn = zero4;              // checkpoint of the block about to be processed, also prefetched
    if (MODE != 3 && tl.valid && (p.nsub - 1) * BS < tl.len)
        Xn = ld_ckpt(ck + (size_t)(p.nsub - 1) * ckb);

    // one BS-step block: recompute alpha_hat from the block's checkpoint, walk the backward steps,
    // stage the outputs; er = the block's raw emission rows
    // FULL: every chain of the wave owns all p.nsub * BS steps (all waves but those holding a sequence's last
    // chunk): no per-lane "is this step mine" selects
    auto block = [&](auto fullc, int j, const f4 *er) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(fullc)::value;
        const int srow = (j % (HMM_OUT_ROWS / BS)) * BS;               // where this block's rows sit in the staged group
        f4 e[BS];
#pragma unroll
        for (int s = 0; s < BS; ++s) e[s] = clampE(er[s], bd);
        const f4 Xc = Xn;
        if (j > 0 && MODE != 3 && tl.valid && (j - 1) * BS < tl.len)
            Xn = ld_ckpt(ck + (size_t)(j - 1) * ckb);
        f4 fa[BS];
        if (MODE != 3) {
            f4 X = Xc;
#pragma unroll
            for (int s = 0; s < BS; ++s) {
                const bool init = tl.first && j == 0 && s == 0;
                const f4 D = mfma4(af, X);
                if (PSI) {
                    // (the start distribution's own clamp is part of the scan's first vector: no flag)
                    const f4 Rf = sel4(init, fmax4(X, eps), clamp_flag4(D, eps));
                    const f4 sf = Rf * e[s];
                    fa[s] = sf * __builtin_amdgcn_rcpf(col_sum(hsum_abs(sf)));
                    X = abs4(fa[s]);
                } else {
                    const f4 sf = fmax4(sel4(init, X, D), eps) * e[s];
                    X = sf * __builtin_amdgcn_rcpf(col_sum(hsum(sf)));
                    fa[s] = X;
                }
            }
        }
        float lacc = 0.f;
#pragma unroll
        for (int s = BS - 1; s >= 0; --s) {
            const bool act = FULL || j * BS + s < tl.len;
            if (MODE == 3) {
                float base = (float)(lbb + (double)lacc);
                stage_row(os, n, g, srow + s, log4(Rv) + base);
            } else {
                f4 gm = fa[s] * Rv;                     // PSI: negative where exactly one of the two clamps was active
                float Sg = col_sum(PSI ? hsum_abs(gm) : hsum(gm));
                const float ig = __builtin_amdgcn_rcpf(Sg);
                if (PSI) ps = fmaf(hsum_neg(gm), act ? ig : 0.f, ps);
                if (MODE == 0) {
                    gm = PSI ? mul_abs4(gm, f4{ig, ig, ig, ig}) : gm * ig;
                } else {
                    gm = log4(PSI ? abs4(gm) : gm) - (__logf(Sg) - llf);
                }
                stage_row(os, n, g, srow + s, gm);
            }
            if (CERT3 && s == 0) { Rc = Rv; Gc = Gv; ec0 = e[0]; }   // (the last block executed is the chunk's first)
            f4 sf = PSI ? mul_abs4(Rv, e[s]) : e[s] * Rv;
            float S = col_sum(hsum(sf));
            const float iS = __builtin_amdgcn_rcpf(S);
            f4 bh = sf * iS;
            const f4 U = mfma4(ab, bh);
            const f4 Rn = PSI ? clamp_flag4(U, eps) : fmax4(U, eps);
            if (CERT3) {
                const f4 Ug = mfma4(ab, e[s] * Gv * iS);
                const f4 Gn = {U.x > eps ? Ug.x : eps, U.y > eps ? Ug.y : eps, U.z > eps ? Ug.z : eps, U.w > eps ? Ug.w : eps};
                Gv = sel4(act, Gn, Gv);
            }
            Rv = sel4(act, Rn, Rv);
            if (MODE == 3) lacc += act ? __logf(S) : 0.f;
        }
        if (MODE == 3) lbb += (double)lacc;
        if (j % (HMM_OUT_ROWS / BS) == 0) {                 // the group's earliest block is done: rows j*BS .. (top group: fewer)
            const int top = p.nsub - j;
            flush_rows(os, lane, j * BS, (top < (HMM_OUT_ROWS / BS) ? top : (HMM_OUT_ROWS / BS)) * BS);
        }
    };

    const bool full = HMM_BWD_FULL && KIND == KIND_SCAN &&
                      __builtin_amdgcn_ballot_w64(tl.valid && tl.len != p.nsub * BS) == 0ull;
    // the previous (earlier-in-time) block's emission rows are in flight while this one is computed
    constexpr bool PF2 = KIND != KIND_SCAN;       // serial plans: two blocks in flight (see forward_body)
    f4 en[BS], en2[BS];
    ld_rows<BS>(tl.rsE, lvoff + (p.nsub - 1) * BS * rowb, rowb, en);
    if (PF2 && p.nsub > 1) ld_rows<BS>(tl.rsE, lvoff + (p.nsub - 2) * BS * rowb, rowb, en2);
    for (int j = p.nsub - 1; j >= 0; --j) {
        f4 e[BS];
#if HMM_COALESCE_B
        permute_rows<BS>(seg, lane, g, n, en, e);
#else
#pragma unroll
        for (int s = 0; s < BS; ++s) e[s] = en[s];
#endif
        if (PF2) {
#pragma unroll
            for (int s = 0; s < BS; ++s) en[s] = en2[s];
            if (j > 1) ld_rows<BS>(tl.rsE, lvoff + (j - 2) * BS * rowb, rowb, en2);
        } else if (j > 0) ld_rows<BS>(tl.rsE, lvoff + (j - 1) * BS * rowb, rowb, en);
        if (full) block(std::true_type(), j, e);
        else block(std::false_type(), j, e);
    }
    if (PSI) {
        ps = col_sum(ps);
        if (g == 0 && tl.valid) psi[tl.chain] = ps;
    } else if (KIND == KIND_SCAN && MODE != 3 && psi && g == 0 && tl.valid) {
        psi[tl.chain] = 0.f;
    }
    if (CERT3) {
        // alpha_hat at the chunk's first position, up to scale: one forward step from the vector entering the chunk
        const f4 P = *reinterpret_cast<const f4 *>(prev + 4 * g);
        const f4 a0 = fmax4(sel4(tl.first, P, mfma4(af, P)), eps) * ec0;
        const float num = col_sum(hsum(a0 * Gc)), den = col_sum(hsum(a0 * Rc));
        shmax = col_sum(hsum(Gv)) * __builtin_amdgcn_rcpf(col_sum(hsum(Rv)));      // what the chunk before does not get
        // ... and what that share becomes under the NEXT observation (the last row of the chunk before, whose kernel starts
        // from the chunk scan's suffix vector: the same direction without the floor).  A floor component that weighs 1e-16
        // here is all there is after a row that only its state can emit.
        float shnext = 0.f;
        {
            f4 ep = {0.f, 0.f, 0.f, 0.f};
            if (tl.valid && tl.chain % p.C != 0) {                  // (not the sequence's first chunk: the row exists)
                const char *pe = reinterpret_cast<const char *>(tl.baseE) + (tl.voff - rowb);
                const f4u raw = *reinterpret_cast<const f4u *>(pe);
                ep = clampE((f4){raw.x, raw.y, raw.z, raw.w}, bd);
            }
            const float dn = col_sum(hsum(ep * Rv));
            shnext = dn > 0.f ? col_sum(hsum(ep * Gv)) * __builtin_amdgcn_rcpf(dn) : 0.f;
        }
        const float c = fmaxf(fmaxf(num * __builtin_amdgcn_rcpf(den), shmax), shnext);
        if (g == 0 && tl.valid) psi[tl.chain] = c;
    }
    if (Rend) *Rend = PSI ? abs4(Rv) : Rv;
    if (MODE == 3 && lbb_end) *lbb_end = lbb;                 // log scale of beta after the tile's first position
}

// psi: [nchains]; rstart: R after every chain's first position, [chain][QP]
template <int MODE, bool EXACT, bool CERT3 = false, int BS = SUB>
__global__ __launch_bounds__(256) void k_backward(const float *__restrict__ A, const float *__restrict__ E,
                                                  const float *__restrict__ ckpt, const float *__restrict__ suffix,
                                                  const double *__restrict__ lsuf, const double *__restrict__ loglik,
                                                  float *__restrict__ out, float *__restrict__ psi,
                                                  float *__restrict__ rstart, Routing rt,
                                                  Plan p, float eps, long long nwaves,
                                                  const float *__restrict__ prefix = nullptr) {
    const long long wave = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    int m; long long wc0;
    Tile tl = make_tile(E, p, wave, g, n, &m, &wc0);
    if (!route_tile<EXACT>(tl, m, g, rt, eps)) return;
    __shared__ __attribute__((aligned(16))) float ostage[4 * OUT_SEG];
    float *seg = ostage + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * OUT_SEG;
    // the reverse cell's initial state: ones (hmm_layer/MsaHmmCell.py:115-116), or the chunk scan's suffix
    const f4 R0 = EXACT ? ones4(p.q, g) : *reinterpret_cast<const f4 *>(suffix + (size_t)tl.chain * QP + 4 * g);
    const double lbb0 = (MODE == 3 && !EXACT) ? lsuf[tl.chain] : 0.0;
    const float llf = MODE == 2 ? (float)loglik[tl.chain / p.C] : 0.f;
    const float *ck = ckpt + ckpt_origin(tl, p, g, n);
    f4 re;
    backward_body<MODE, EXACT ? KIND_EXACT : KIND_SCAN, CERT3, BS>(A, E, R0, lbb0, llf, ck, ckpt_block(p), out, psi, tl, m, seg,
                                                               p, eps, &re, CERT3 ? prefix + (size_t)tl.chain * QP : nullptr);
    if (!EXACT && rstart && tl.valid) *reinterpret_cast<f4 *>(rstart + (size_t)tl.chain * QP + 4 * g) = re;
}

// ---- the per-sequence verdict, one wave per sequence.
//   ROUTE_WHOLE    model routed by k_topo_check; or too much of the sequence is flagged
//   ROUTE_WINDOWS  psi (summed over the sequence's chunks) above EXACT_DELTA: the chunks that carry the
//                  clamp-born mass are found, each is widened by `margin` chunks on either side (forward-born mass
//                  changes alpha_hat downstream, backward-born mass R upstream), overlapping runs are merged -> the
//                  window table; the rest of the sequence keeps the scan's values
//   ROUTE_NONE     everything else
// wcnt: [0] sequences with windows (= entries of wlist), [1] sequences redone whole because of their psi or of
// windows that ran into each other, [2] windows, [3] chunks the windows walked; zeroed by k_topo_check.
// exps (or null): the chunk operators' exponent rows; the sparse reduce marks chains whose columns went through the
// denormal range there (reduce_sparse_wave's `risk`) — such a chunk counts as flagged whatever its psi
__global__ __launch_bounds__(64) void k_exact_select(const int *__restrict__ topo, float *__restrict__ psi, Plan p,
                                                     int exact_mode, int margin, int *__restrict__ flags,
                                                     int *__restrict__ nexact, int *__restrict__ wtab,
                                                     int *__restrict__ wlist, int *__restrict__ wcnt,
                                                     const int *__restrict__ exps = nullptr) {
    const int seq = blockIdx.x, lane = threadIdx.x;
    const int C = p.C;
    const int tpv = topo[seq / p.b];
    if (tpv == TOPO_EXACT) {
        if (lane == 0) { flags[seq] = ROUTE_WHOLE; atomicAdd(nexact, 1); }
        return;
    }
    float *pc = psi + (size_t)seq * C;
    if (exps && tpv != 0 && exact_mode == HMM_EXACT_AUTO)          // (every later loop visits chunk c from the same lane)
        for (int c = lane; c < C; c += 64)
            if (exps[((size_t)seq * C + c) * QP + QP - 1] != 0) pc[c] = 1.f;
    float s = 0.f;
    if (exact_mode == HMM_EXACT_AUTO) {
        for (int c = lane; c < C; c += 64) s += pc[c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    }
    if (s <= EXACT_DELTA) {                                           // (NaN / inf fall through)
        if (lane == 0) flags[seq] = ROUTE_NONE;
        return;
    }
    if (margin < 0) {                                                 // this entry point has no windows
        if (lane == 0) { flags[seq] = ROUTE_WHOLE; atomicAdd(nexact, 1); atomicAdd(wcnt + 1, 1); }
        return;
    }
    // the largest per-chunk threshold that leaves at most EXACT_DELTA / 2 outside the flagged chunks
    float thr = EXACT_DELTA * 0.125f;
    for (int it = 0; it < 8; ++it) {
        float rem = 0.f;
        for (int c = lane; c < C; c += 64) { const float v = pc[c]; rem += v <= thr ? v : 0.f; }
        for (int o = 32; o > 0; o >>= 1) rem += __shfl_xor(rem, o);
        if (rem <= 0.5f * EXACT_DELTA) break;
        thr *= 0.125f;
    }
    // Windows.  A flagged chunk c standing alone is itself right as the scan plan computed it — its in-chunk steps
    // apply the clamps, from vectors that are right when nothing flagged precedes / follows within reach; what its
    // births change is alpha_hat AFTER it and R BEFORE it.  It therefore gets two windows, [c - margin, c - 1] and
    // [c + 1, c + margin] (two tile columns of the window kernel: they run side by side); the second one's forward
    // pass starts at c itself (WIN_LEAD) so that the window's log-likelihood covers the births inside c.  A run of
    // adjacent flagged chunks becomes one window with the margin on either side.  Windows that touch are merged.
    __shared__ int wlo[WIN_MAX], whi[WIN_MAX], wld[WIN_MAX];
    int nw = 0;
    bool over = false;
    auto push = [&](int lo, int hi, int lead) {                       // lane 0 only
        while (nw > 0 && lo <= whi[nw - 1] + 1) {
            const int fstart = min(lo - lead, wlo[nw - 1] - wld[nw - 1]);     // where the merged forward pass starts
            lo = min(lo, wlo[nw - 1]); hi = max(hi, whi[nw - 1]); --nw;
            lead = lo - fstart;
        }
        if (nw == WIN_MAX) { over = true; lo = min(lo, wlo[nw - 1]); hi = max(hi, whi[nw - 1]); lead = wld[nw - 1]; --nw; }
        wlo[nw] = lo; whi[nw] = hi; wld[nw] = lead; ++nw;
    };
    auto flush_run = [&](int c, int d) {
        if (c < 0) return;
        if (d > c) { push(max(0, c - margin), min(C - 1, d + margin), 0); return; }
        if (c > 0) push(max(0, c - margin), c - 1, 0);
        push(c + 1, min(C - 1, c + margin), 1);                       // (c == C - 1: no chunks, the forward pass over c alone)
    };
    int run_lo = -1, run_hi = -2;
    for (int base = 0; base < C; base += 64) {
        const int c = base + lane;
        const bool hot = c < C && !(pc[c] <= thr);
        const unsigned long long hmask = __builtin_amdgcn_ballot_w64(hot);
        if (lane == 0) {
            unsigned long long any = hmask;
            while (any) {
                const int bit = __builtin_ctzll(any);
                any &= any - 1;
                const int cc = base + bit;
                if (cc == run_hi + 1) { run_hi = cc; }
                else { flush_run(run_lo, run_hi); run_lo = run_hi = cc; }
            }
        }
    }
    if (lane == 0) {
        flush_run(run_lo, run_hi);
        int tot = 0;
        for (int i = 0; i < nw; ++i) tot += whi[i] - wlo[i] + 1 + wld[i];
        const bool whole = over || 4ll * tot >= 3ll * C;
        flags[seq] = whole ? ROUTE_WHOLE : ROUTE_WINDOWS;
        atomicAdd(nexact, 1);
        if (whole) {
            atomicAdd(wcnt + 1, 1);
        } else {
            int *wt = wtab + (size_t)seq * WIN_STRIDE;
            wt[0] = nw;
            for (int i = 0; i < nw; ++i) { wt[2 + 2 * i] = wlo[i]; wt[3 + 2 * i] = (whi[i] - wlo[i] + 1) | (wld[i] ? WIN_LEAD : 0); }
            wlist[atomicAdd(wcnt, 1)] = seq;
            atomicAdd(wcnt + 2, nw);
        }
    }
}

// ---- windows: the serial recursion over the flagged runs of chunks only.  One wave per sequence with windows
// (grid stride over wlist), window i in tile column i.  A window [lo, hi] walks the cell's exact steps
//   forward   from alpha_hat at the end of chunk lo - 1 (xend: what the scan plan's forward kernel stepped to from a
//             prefix that no flagged chunk precedes within reach), and on PAST hi, a growing number of chunks at a
//             time, until the posterior it arrives with at a chunk's last position (its alpha_hat times the scan
//             plan's R there) equals the scan plan's to WIN_TOL — the births inside the window have then been
//             forgotten by the recursion, as far as the future can tell — or the sequence ends;
//   backward  from R at the start of chunk hi + 1 (rstart) over everything the forward pass walked, and on BELOW lo
//             (recomputing the scan's forward vectors there for the checkpoints) until the posterior before the
//             window (the scan plan's alpha_hat times its R) meets the scan plan's;
// and writes the outputs of every position it walked.  Everything outside stands as the scan computed it; the
// window's own log-likelihood takes the place of the chunk scan's for its forward span.  A window may grow up to
// its neighbours; one that has not met the scan's vectors by then (the windows would have to be merged) sends its
// sequence to the whole-sequence kernel that follows.  The cost of a flagged sequence is therefore the flagged
// chunks plus the model's forgetting time, not its length.  Checkpoints: rows [seq][chunk][block] of the checkpoint
// region (the scan plan's checkpoints have been consumed by k_backward).
// What a window does on its way back is the caller's (Pol): the posterior kernels' backward body, the gradient's
// (hmm_grad.inc), or nothing at all (log-likelihood only: MODE 4, the forward half alone — the windows' log-likelihood
// in place of the chunk scan's; the future's weights in the far-end check are then the chunk scan's suffix vectors).
//   Pol::BACKWARD, Pol::CKPT                       compile-time: is there a backward half, are checkpoints written
//   pol.backward(tl, m, R, ck, pw, &re)            walk tile tl backward from R (checkpoint rows ck + j * QP)
//   pol.finish(seq, good)                          after the sequence's windows (good: none ran into a neighbour)
// The final extents of the windows are written back to the table (the gradient's second scan-plan launch leaves
// exactly these chains out).
template <class Pol>
__device__ __forceinline__ void window_walk(Pol &pol, const float *__restrict__ A, const float *__restrict__ E,
                                            const float *__restrict__ prefix, const double *__restrict__ llpre,
                                            const float *__restrict__ suffix, const float *__restrict__ xend,
                                            const float *__restrict__ rstart, float *__restrict__ ckpt,
                                            double *__restrict__ loglik, int *__restrict__ wtab,
                                            const int *__restrict__ wlist, int *__restrict__ wcnt, int *__restrict__ flags,
                                            double *__restrict__ dfix, const Plan &p, float eps, int ext0, float *seg,
                                            double *__restrict__ wshift = nullptr) {
    constexpr bool LLONLY = !Pol::BACKWARD;
    constexpr bool LOGA = Pol::LOGA;       // the forward half also writes log alpha of what it walks (pol.loga)
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nlist = wcnt[0];
    const int C = p.C;
    const unsigned long long total = (unsigned long long)p.NB * p.L * p.q * sizeof(float);
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto wave_max = [](int v) {
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
        return __builtin_amdgcn_readfirstlane(v);
    };
    for (int it = blockIdx.x * 4 + w; it < nlist; it += gridDim.x * 4) {
        const int seq = wlist[it];
        const int m = seq / p.b;
        int *wt = wtab + (size_t)seq * WIN_STRIDE;
        const int nwin = wt[0];
        const bool valid = n < nwin;
        int lo = valid ? wt[2 + 2 * n] : 0;
        const int lead = valid && (wt[3 + 2 * n] & WIN_LEAD) ? 1 : 0;                 // the forward pass starts at lo - lead
        int hi = valid ? lo + (wt[3 + 2 * n] & (WIN_LEAD - 1)) - 1 : -1;
        const int lo_first = lo - lead;
        // where the next window's forward pass starts: this one's has to have met the scan's vectors by then
        const int lo_next = (valid && n + 1 < nwin) ? wt[2 + 2 * (n + 1)] - ((wt[3 + 2 * (n + 1)] & WIN_LEAD) ? 1 : 0) : C;
        const float *baseE = E + (size_t)seq * p.L * p.q;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(baseE, total - (unsigned long long)seq * p.L * p.q * sizeof(float));
        const size_t chs = (size_t)seq * C;                                          // chain of the sequence's chunk 0
        float *ckq = ckpt + chs * p.nsub * QP + 4 * g;
        const double ll_scan = loglik[seq];
        pol.begin(seq, m, (float)ll_scan);
        // what a column walks in one call: chunks [a, a + nch) (nothing when !on)
        auto seg_tile = [&](bool on, int a, int nch) {
            Tile tl;
            tl.wave = 0;
            tl.chain = (long long)(chs + (on ? a : 0));
            tl.valid = on;
            tl.first = on && a == 0 && p.seq_start;
            tl.len = on ? min(nch * p.T, p.L - a * p.T) : 0;
            tl.voff = (on ? a : 0) * p.T * p.q * (int)sizeof(float) + g * 16;
            tl.baseE = baseE;
            tl.rsE = rs;
            return tl;
        };
        auto ld4 = [&](bool on, const float *ptr) { return on ? *reinterpret_cast<const f4 *>(ptr + 4 * g) : zero4; };
        auto start_vec = [&](bool on, int a) {        // alpha_hat entering chunk a, as the scan plan has it
            return ld4(on, a == 0 ? prefix + chs * QP : xend + (chs + (on ? a : 1) - 1) * QP);
        };
        bool conflict = false;
        // ---- forward: the window, then on until the scan's vector is met
        f4 X = start_vec(valid, lo_first);
        double llw = 0.0;
        // log alpha_t = log alpha_hat_t + the log-likelihood up to t: the window's rows carry the chunk scan's value at
        // the window's start plus the window's own steps; what earlier windows of the sequence changed is added to
        // every later row afterwards (k_window_shift_loga)
        double llrun = (LOGA && valid) ? llpre[chs + lo_first] : 0.0;
        {
            bool on = valid, merged = !valid;
            int a = lo_first, nch = hi - lo_first + 1, ext = ext0;
            while (__builtin_amdgcn_ballot_w64(on) != 0ull) {
                const Tile tl = seg_tile(on, a, nch);
                Plan pw = p;
                pw.nsub = (wave_max(tl.len) + SUB - 1) / SUB;
                f4 xe;
                const double l = forward_body<Pol::CKPT, LOGA, KIND_WIN>(A, E, X, llrun, ckq + (size_t)(on ? a : 0) * p.nsub * QP,
                                                                        (size_t)QP, pol.loga(), tl, m, seg, pw, eps, &xe) - llrun;
                if (on) { X = xe; llw += l; llrun += l; hi = a + nch - 1; }
                const bool tail = hi + 1 >= C;
                // met = the POSTERIORS at the chunk's last position agree (the scan plan's R there weighs the
                // difference: a component that is tiny in alpha_hat may be all the future cares about — comparing
                // the filtered vectors alone accepted windows that were off by 6e-2 downstream)
                const bool chk = on && !tail;
                const f4 xs = ld4(chk, xend + (chs + (on ? hi : 0)) * QP);
                const f4 rw = ld4(chk, LLONLY ? suffix + (chs + (on ? hi : 0)) * QP : rstart + (chs + (on ? hi : 0) + 1) * QP);
                const f4 ge = X * rw, gs = xs * rw;
                const float ie = __builtin_amdgcn_rcpf(col_sum(hsum(ge))), is = __builtin_amdgcn_rcpf(col_sum(hsum(gs)));
                float d = col_max(hmax(abs4(ge * ie - gs * is)));
                if (LOGA) d = fmaxf(d, col_max(hmax(abs4(X - xs))));    // log alpha is a statement about alpha_hat itself
                if (on) merged = tail || d <= WIN_TOL;
                on = valid && !merged;
                if (on) {
                    a = hi + 1;
                    const int bnd = min(min(C - 1, lo_next - 1), a + ext - 1);
                    if (bnd < a) { conflict = true; on = false; }
                    nch = bnd - a + 1;
                }
                ext *= 2;
            }
        }
        double dll = 0.0;
        if (valid) dll = llw - ((hi + 1 >= C ? ll_scan : llpre[chs + hi + 1]) - llpre[chs + lo_first]);
        __threadfence();
        // ---- backward: everything the forward pass walked, then on below the window until the scan's R is met
        if (Pol::BACKWARD) {
            f4 R = ld4(valid, hi + 1 >= C ? suffix + (chs + max(hi, 0)) * QP : rstart + (chs + hi + 1) * QP);
            const bool nonempty = valid && hi >= lo;             // (a lead-only window at the sequence's end has no chunks)
            {
                const Tile tl = seg_tile(nonempty, lo, hi - lo + 1);
                Plan pw = p;
                pw.nsub = (wave_max(tl.len) + SUB - 1) / SUB;
                f4 re;
                pol.backward(A, E, tl, m, R, ckq + (size_t)(nonempty ? lo : 0) * p.nsub * QP, pw, eps, seg, &re);
                if (nonempty) R = re;
            }
            const int hi_prev = __shfl(hi, (lane + 63) & 63);        // the previous window's last chunk after its forward pass
            const int lob = (valid && n > 0) ? hi_prev + 1 : 0;
            bool merged = !nonempty;
            int ext = ext0;
            while (true) {
                // met = the posteriors at the last position BEFORE the window agree (weighed by the scan plan's
                // alpha_hat there)
                const bool chk = nonempty && lo > 0;
                const f4 rsv = ld4(chk, rstart + (chs + lo) * QP);
                const f4 aw = ld4(chk, xend + (chs + max(lo, 1) - 1) * QP);
                const f4 ge = aw * R, gs = aw * rsv;
                const float i1 = __builtin_amdgcn_rcpf(col_sum(hsum(ge))), i2 = __builtin_amdgcn_rcpf(col_sum(hsum(gs)));
                const float d = col_max(hmax(abs4(ge * i1 - gs * i2)));
                if (nonempty && !merged) merged = lo == 0 || d <= WIN_TOL;
                bool on = nonempty && !merged && !conflict;
                int a = 0, nch = 0;
                if (on) {
                    a = max(lob, lo - ext);
                    if (a > lo - 1) { conflict = true; on = false; }
                    nch = lo - a;
                }
                if (__builtin_amdgcn_ballot_w64(on) == 0ull) break;
                const Tile tl = seg_tile(on, a, nch);
                Plan pw = p;
                pw.nsub = (wave_max(tl.len) + SUB - 1) / SUB;
                float *ck = ckq + (size_t)(on ? a : 0) * p.nsub * QP;
                forward_body<true, false, KIND_WIN>(A, E, start_vec(on, a), 0.0, ck, (size_t)QP, nullptr, tl, m, seg, pw, eps);
                __threadfence();
                f4 re;
                pol.backward(A, E, tl, m, R, ck, pw, eps, seg, &re);
                if (on) { R = re; lo = a; }
                ext *= 2;
            }
        }
        const bool good = __builtin_amdgcn_ballot_w64(valid && conflict) == 0ull;
        // the sequence's log-likelihood: the windows' own sums in place of the chunk scan's, in window order
        double dsum = 0.0;
        int walked = 0;
        for (int i = 0; i < nwin; ++i) {
            const long long bits = __builtin_bit_cast(long long, dll);
            const int lo32 = __builtin_amdgcn_readlane((int)bits, i), hi32 = __builtin_amdgcn_readlane((int)(bits >> 32), i);
            dsum += __builtin_bit_cast(double, ((long long)hi32 << 32) | (unsigned int)lo32);
            walked += __builtin_amdgcn_readlane(hi - lo + 1, i);
        }
        if (valid && g == 0) { wt[2 + 2 * n] = lo; wt[3 + 2 * n] = hi - lo + 1; }     // the final extents (outputs written)
        if (wshift && valid && g == 0) {
            double *ws_ = wshift + (size_t)seq * WSH_STRIDE;
            ws_[n] = dll;
            reinterpret_cast<int *>(ws_ + WIN_MAX)[n] = hi;
        }
        // (a chunk scan that broke down altogether — a non-finite log-likelihood — cannot be repaired by differences)
        const bool fine = good && __builtin_isfinite(ll_scan + dsum);
        if (lane == 0) {
            atomicAdd(wcnt + 3, walked);
            if (fine) {
                loglik[seq] = ll_scan + dsum;
                dfix[seq] = dsum;
            } else {
                flags[seq] = ROUTE_WHOLE;
                atomicAdd(wcnt + 1, 1);
            }
        }
        pol.finish(seq, m, fine);
    }
}

// MODE 0..2: the posterior's output modes; MODE 4: the log-likelihood alone (hmm_forward without log alpha)
// MODE 5: log alpha (hmm_forward): forward half only, writing log alpha of every position walked
template <int MODE>
struct PostWindows {
    static constexpr bool BACKWARD = MODE < 4, CKPT = MODE < 4, LOGA = MODE == 5;
    float *out;
    float llf;
    __device__ __forceinline__ float *loga() const { return LOGA ? out : nullptr; }
    __device__ __forceinline__ void begin(int, int, float ll_scan) { llf = ll_scan; }
    __device__ __forceinline__ void backward(const float *A, const float *E, const Tile &tl, int m, f4 R, const float *ck,
                                             const Plan &pw, float eps, float *seg, f4 *re) {
        backward_body<MODE >= 4 ? 0 : MODE, KIND_WIN>(A, E, R, 0.0, llf, ck, (size_t)QP, out, nullptr, tl, m, seg, pw, eps, re);
    }
    __device__ __forceinline__ void finish(int, int, bool) {}
};

template <int MODE>
__global__ __launch_bounds__(256) void k_window_posterior(const float *__restrict__ A, const float *__restrict__ E,
                                                          const float *__restrict__ prefix, const double *__restrict__ llpre,
                                                          const float *__restrict__ suffix, const float *__restrict__ xend,
                                                          const float *__restrict__ rstart, float *__restrict__ ckpt,
                                                          double *__restrict__ loglik, float *__restrict__ out,
                                                          int *__restrict__ wtab, const int *__restrict__ wlist,
                                                          int *__restrict__ wcnt, int *__restrict__ flags,
                                                          double *__restrict__ dfix, Plan p, float eps, int ext0,
                                                          double *__restrict__ wshift = nullptr) {
    constexpr int SEG = MODE == 4 ? 16 * IN_STRIDE : OUT_SEG;
    __shared__ __attribute__((aligned(16))) float ostage[4 * SEG];
    float *seg = ostage + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * SEG;
    PostWindows<MODE> pol;
    pol.out = out;
    window_walk(pol, A, E, prefix, llpre, suffix, xend, rstart, ckpt, loglik, wtab, wlist, wcnt, flags, dfix, p, eps, ext0, seg,
                wshift);
}

// log beta (hmm_backward) of the sequences k_exact_select gave windows: the backward half alone.  What is born in a
// flagged chunk changes R BEFORE it; every window of the table is walked backward from the scan plan's
// R at its upper end (rstart) with the chunk scan's log scale there, writing log beta, and on below the
// window until the vector it arrives with equals the scan plan's — weighed with alpha_hat there as the chunk scan has
// it (hmm_backward's uniform start), and unweighed, log beta being a statement about the vector itself.  The rows
// BELOW a window then move with its log scale (k_window_shift_logb).  Windows that run into their neighbours send
// the sequence to the whole-sequence kernel.
__global__ __launch_bounds__(256) void k_window_logbeta(const float *__restrict__ A, const float *__restrict__ E,
                                                        const float *__restrict__ prefix, const float *__restrict__ suffix,
                                                        const double *__restrict__ lsuf, const float *__restrict__ rstart,
                                                        float *__restrict__ out, int *__restrict__ wtab,
                                                        const int *__restrict__ wlist, int *__restrict__ wcnt,
                                                        int *__restrict__ flags, double *__restrict__ wshift, Plan p,
                                                        float eps, int ext0) {
    __shared__ __attribute__((aligned(16))) float ostage[4 * OUT_SEG];
    float *seg = ostage + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * OUT_SEG;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nlist = wcnt[0];
    const int C = p.C;
    const unsigned long long total = (unsigned long long)p.NB * p.L * p.q * sizeof(float);
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto wave_max = [](int v) {
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
        return __builtin_amdgcn_readfirstlane(v);
    };
    for (int it = blockIdx.x * 4 + w; it < nlist; it += gridDim.x * 4) {
        const int seq = wlist[it];
        const int m = seq / p.b;
        int *wt = wtab + (size_t)seq * WIN_STRIDE;
        const int nwin = wt[0];
        const bool valid = n < nwin;
        int lo = valid ? wt[2 + 2 * n] : 0;
        const int hi_tab = valid ? lo + (wt[3 + 2 * n] & (WIN_LEAD - 1)) - 1 : -1;
        // (every window is walked: one that only follows a flagged chunk meets the scan's R at once, but merged windows
        // carry the lead mark of their first part)
        const bool act = valid && hi_tab >= lo;
        // the pass starts one chunk ABOVE the window: a flagged chunk standing alone sits there (its own rows are
        // right as the scan computed them, but the log scale it hands down has to be the in-chunk one; the mirror
        // image of WIN_LEAD) — any other chunk is simply walked again with the same result
        const int hi = act ? min(hi_tab + 1, C - 1) : hi_tab;
        const float *baseE = E + (size_t)seq * p.L * p.q;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(baseE, total - (unsigned long long)seq * p.L * p.q * sizeof(float));
        const size_t chs = (size_t)seq * C;
        auto seg_tile = [&](bool on, int a, int nch) {
            Tile tl;
            tl.wave = 0;
            tl.chain = (long long)(chs + (on ? a : 0));
            tl.valid = on;
            tl.first = on && a == 0 && p.seq_start;
            tl.len = on ? min(nch * p.T, p.L - a * p.T) : 0;
            tl.voff = (on ? a : 0) * p.T * p.q * (int)sizeof(float) + g * 16;
            tl.baseE = baseE;
            tl.rsE = rs;
            return tl;
        };
        auto ld4 = [&](bool on, const float *ptr) { return on ? *reinterpret_cast<const f4 *>(ptr + 4 * g) : zero4; };
        // the window itself, from R at the last position of chunk hi as the chunk above left it (rstart: the in-chunk
        // steps' normalisation, every component lifted to eps as the cell does — the chunk scan's suffix vector is the
        // same direction WITHOUT the floor, which is all that is left of some states inside a stretch); its log scale
        // is the suffix vector's, moved by the ratio of the two vectors' sums.  The sequence's last chunk starts from
        // the suffix vector itself (ones).
        const bool top = hi + 1 >= C;
        const f4 sv0 = ld4(act, suffix + (chs + max(hi, 0)) * QP);
        f4 R = top ? sv0 : ld4(act, rstart + (chs + hi + 1) * QP);
        double lbrun = 0.0;
        {
            const float s0 = col_sum(hsum(sv0)), s1 = col_sum(hsum(R));
            if (act) lbrun = lsuf[chs + hi] + (top ? 0.0 : (double)__logf(s0) - (double)__logf(s1));
        }
        bool conflict = false;
        {
            const Tile tl = seg_tile(act, lo, hi - lo + 1);
            Plan pw = p;
            pw.nsub = (wave_max(tl.len) + SUB - 1) / SUB;
            f4 re;
            double le = 0.0;
            backward_body<3, KIND_WIN>(A, E, R, lbrun, 0.f, nullptr, (size_t)QP, out, nullptr, tl, m, seg, pw, eps, &re, nullptr, &le);
            if (act) { R = re; lbrun = le; }
        }
        // how far down a window may grow: to the top chunk of the nearest window below it that is walked at all
        int lob = 0;
        {
            const int top_act = act ? hi : -1;
            for (int i = 0; i < nwin; ++i) {
                const int v = __builtin_amdgcn_readlane(top_act, i);
                lob = (i < n) ? max(lob, v + 1) : lob;
            }
        }
        bool merged = !act;
        int ext = ext0, walked = act ? hi - lo + 1 : 0;
        while (true) {
            const bool chk = act && lo > 0;
            const f4 rsv = ld4(chk, rstart + (chs + lo) * QP);
            const f4 aw = ld4(chk, prefix + (chs + lo) * QP);
            const f4 ge = aw * R, gs = aw * rsv;
            const float i1 = __builtin_amdgcn_rcpf(col_sum(hsum(ge))), i2 = __builtin_amdgcn_rcpf(col_sum(hsum(gs)));
            float d = col_max(hmax(abs4(ge * i1 - gs * i2)));
            d = fmaxf(d, col_max(hmax(abs4(R - rsv))));
            if (act && !merged) merged = lo == 0 || d <= WIN_TOL;
            bool on = act && !merged && !conflict;
            int a = 0, nch = 0;
            if (on) {
                a = max(lob, lo - ext);
                if (a > lo - 1) { conflict = true; on = false; }
                nch = lo - a;
            }
            if (__builtin_amdgcn_ballot_w64(on) == 0ull) break;
            const Tile tl = seg_tile(on, a, nch);
            Plan pw = p;
            pw.nsub = (wave_max(tl.len) + SUB - 1) / SUB;
            f4 re;
            double le = 0.0;
            backward_body<3, KIND_WIN>(A, E, R, lbrun, 0.f, nullptr, (size_t)QP, out, nullptr, tl, m, seg, pw, eps, &re, nullptr, &le);
            if (on) { R = re; lbrun = le; lo = a; walked += nch; }
            ext *= 2;
        }
        bool good = __builtin_amdgcn_ballot_w64(valid && conflict) == 0ull;
        // what the rows below the window move by: the window's log scale at its lower end against the chunk scan's
        // (log beta of the last position below the window, both ways: the vectors agree up to scale there)
        double dlb = 0.0;
        {
            const bool dn = act && lo > 0;
            const f4 sv = ld4(dn, suffix + (chs + max(lo, 1) - 1) * QP);
            const float sr = col_sum(hsum(R)), ss = col_sum(hsum(sv));
            if (dn) dlb = (lbrun + (double)__logf(sr)) - (lsuf[chs + lo - 1] + (double)__logf(ss));
        }
        good = good && __builtin_amdgcn_ballot_w64(valid && !__builtin_isfinite(dlb)) == 0ull;   // (a chunk scan that broke down)
        if (valid && g == 0) {
            double *ws_ = wshift + (size_t)seq * WSH_STRIDE;
            ws_[n] = dlb;
            reinterpret_cast<int *>(ws_ + WIN_MAX)[n] = act ? lo : C;       // (inactive: above every chunk, shifts nothing)
        }
        int wsum = 0;
        for (int i = 0; i < nwin; ++i) wsum += __builtin_amdgcn_readlane(walked, i);
        if (lane == 0) {
            atomicAdd(wcnt + 3, wsum);
            if (!good) { flags[seq] = ROUTE_WHOLE; atomicAdd(wcnt + 1, 1); }
        }
    }
}

// log beta after the windows: every row of chunk c moves by the log-scale shifts of the windows that BEGIN above c
__global__ __launch_bounds__(256) void k_window_shift_logb(float *__restrict__ out, const int *__restrict__ wtab,
                                                           const int *__restrict__ wlist, const int *__restrict__ wcnt,
                                                           const int *__restrict__ flags, const double *__restrict__ wshift,
                                                           Plan p) {
    const int nlist = wcnt[0];
    const size_t rowsz = (size_t)p.q;
    for (int it = blockIdx.y; it < nlist; it += gridDim.y) {
        const int seq = wlist[it];
        if (flags[seq] != ROUTE_WINDOWS) continue;
        const int nwin = wtab[(size_t)seq * WIN_STRIDE];
        const double *d = wshift + (size_t)seq * WSH_STRIDE;
        const int *los = reinterpret_cast<const int *>(d + WIN_MAX);
        int top = 0;                                                 // nothing moves at or above the highest window's start
        for (int w = 0; w < nwin; ++w) top = (d[w] != 0.0 && los[w] > top) ? los[w] : top;
        float *o = out + (size_t)seq * p.L * rowsz;
        const size_t total = min((size_t)top * p.T, (size_t)p.L) * rowsz;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const int c = (int)(i / (rowsz * p.T));
            double sh = 0.0;
            for (int w = 0; w < nwin; ++w) sh += los[w] > c ? d[w] : 0.0;
            o[i] += (float)sh;
        }
    }
}

// log alpha after the windows: every row of chunk c moves by the log-likelihood shifts of the windows that END before
// c (rows inside a window carry the chunk scan's value at the window's start already).  One block column per sequence
// of the window list; sequences that go on to the whole-sequence kernel are rewritten there.
__global__ __launch_bounds__(256) void k_window_shift_loga(float *__restrict__ out, const int *__restrict__ wtab,
                                                           const int *__restrict__ wlist, const int *__restrict__ wcnt,
                                                           const int *__restrict__ flags, const double *__restrict__ wshift,
                                                           Plan p) {
    const int nlist = wcnt[0];
    const size_t rowsz = (size_t)p.q;
    for (int it = blockIdx.y; it < nlist; it += gridDim.y) {
        const int seq = wlist[it];
        if (flags[seq] != ROUTE_WINDOWS) continue;
        const int nwin = wtab[(size_t)seq * WIN_STRIDE];
        const double *d = wshift + (size_t)seq * WSH_STRIDE;
        const int *his = reinterpret_cast<const int *>(d + WIN_MAX);
        const int c0 = his[0] + 1;                                   // nothing moves before the first window's end
        float *o = out + (size_t)seq * p.L * rowsz;
        const size_t first = (size_t)c0 * p.T * rowsz, total = (size_t)p.L * rowsz;
        for (size_t i = first + (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const int c = (int)(i / (rowsz * p.T));
            double sh = 0.0;
            for (int w = 0; w < nwin; ++w) sh += his[w] < c ? d[w] : 0.0;
            o[i] += (float)sh;
        }
    }
}

// HMM_POST_LOG_NO_LL (log gamma + loglik): the log-likelihood the windows corrected enters every position of the sequence
__global__ __launch_bounds__(256) void k_window_fixll(float *__restrict__ out, const int *__restrict__ wlist,
                                                      const int *__restrict__ wcnt, const int *__restrict__ flags,
                                                      const double *__restrict__ dfix, Plan p) {
    const int nlist = wcnt[0];
    const size_t per = (size_t)p.L * p.q;
    for (int it = blockIdx.y; it < nlist; it += gridDim.y) {
        const int seq = wlist[it];
        if (flags[seq] != ROUTE_WINDOWS) continue;
        const float d = (float)dfix[seq];
        if (d == 0.f) continue;
        float *o = out + (size_t)seq * per;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) o[i] += d;
    }
}

// The whole-sequence serial posterior in ONE launch (its waves exit at once when nothing is routed, so
// the common case pays one empty launch): forward pass writing checkpoints and the log-likelihood,
// then the backward pass of the same wave reading them back (same lanes, same addresses).
template <int MODE>
__global__ __launch_bounds__(256) void k_exact_posterior(const float *__restrict__ A, const float *__restrict__ pi,
                                                         const float *__restrict__ E, float *__restrict__ ckpt,
                                                         double *__restrict__ loglik, float *__restrict__ out,
                                                         Routing rt, Plan p, float eps, long long nwaves) {
    const long long wave = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    int m; long long wc0;
    Tile tl = make_tile(E, p, wave, g, n, &m, &wc0);
    if (!route_tile<true>(tl, m, g, rt, eps)) return;
    __shared__ __attribute__((aligned(16))) float ostage[4 * OUT_SEG];
    float *seg = ostage + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * OUT_SEG;
    float *ck = ckpt + ckpt_origin(tl, p, g, n);
    const double ll = forward_body<true, false, KIND_EXACT>(A, E, ld_state4(pi + (size_t)m * p.q, p.q, g), 0.0, ck,
                                                            ckpt_block(p), nullptr, tl, m, seg, p, eps);
    if (tl.valid && g == 0) loglik[tl.chain] = ll;
    __threadfence();
    backward_body<MODE, KIND_EXACT>(A, E, ones4(p.q, g), 0.0, (float)ll, ck, ckpt_block(p), out, nullptr, tl, m, seg, p, eps);
}

// ------------------------------------------------------------------ small kernels

__global__ void k_copy_loglik(const double *__restrict__ src, double *__restrict__ dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// (sum_b w*loglik, sum_b w) per model; one block per model, deterministic tree
__global__ __launch_bounds__(256) void k_loglik_partials(const double *__restrict__ ll, const float *__restrict__ w,
                                                         int b, double *__restrict__ partial) {
    __shared__ double s1[256], s2[256];
    const int m = blockIdx.x;
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < b; i += 256) {
        double wi = w ? (double)w[(size_t)m * b + i] : 1.0;
        a += wi * ll[(size_t)m * b + i];
        c += wi;
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * m] = s1[0]; partial[2 * m + 1] = s2[0]; }
}

// ------------------------------------------------------------------ host side

static int check_launch() { return hipGetLastError() == hipSuccess ? HMM_OK : HMM_ERR_LAUNCH; }

// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
struct Profile {
    struct Span { int kernel; hipEvent_t a, b; };
    std::vector<Span> spans;
};
struct Timed {   // brackets one launch when a profile is attached
    Profile *pr; hipStream_t st; Profile::Span sp;
    Timed(Profile *pr_, int kernel, hipStream_t st_) : pr(pr_), st(st_) {
        if (!pr) return;
        sp.kernel = kernel;
        (void)hipEventCreate(&sp.a); (void)hipEventCreate(&sp.b);
        (void)hipEventRecord(sp.a, st);
    }
    ~Timed() {
        if (!pr) return;
        (void)hipEventRecord(sp.b, st);
        pr->spans.push_back(sp);
    }
};

// chunk operators of every (sequence, chunk): ws + o_ops / o_exps
static void run_reduce(const float *A, const float *E, const Plan &p, float eps, char *ws, hipStream_t st,
                       Profile *pr, int exact_mode) {
    float *ops = (float *)(ws + p.o_ops);
    int *exps = (int *)(ws + p.o_exps);
    const unsigned nb = (unsigned)((p.nchains + 3) / 4);
    int *topo = (int *)(ws + p.o_topo);
    const int force_dense = opt(HMM_OPT_FORCE_DENSE) == 1 ? 1 : 0;
    hipLaunchKernelGGL(k_topo_check, dim3(p.k), dim3(64), 0, st, A, topo, p.k, p.q, force_dense, exact_mode,
                       eps, (int *)(ws + p.o_nexact), (int *)(ws + p.o_wcnt));
    {
        // every (sequence, chunk) is served by exactly one of the two kernels, chosen on the
        // device from the support of its model's A; the other kernel's waves exit at once
        Timed t(pr, HMM_KERNEL_REDUCE, st);
        const unsigned nbs = (unsigned)((p.nchains + 15) / 16);
        if (p.q == TopoGene15::Q) {
            hipLaunchKernelGGL(k_reduce_sparse<TopoGene15>, dim3(nbs), dim3(256), 0, st, A, E, ops, exps, topo, p, eps);
            if (p.k > 1 || !HMM_RS_UNI)        // waves that straddle two models
                hipLaunchKernelGGL((k_reduce_sparse<TopoGene15, true>), dim3(nbs), dim3(256), 0, st, A, E, ops, exps, topo, p, eps);
        } else if (p.q == TopoGene7::Q) {
            hipLaunchKernelGGL(k_reduce_sparse<TopoGene7>, dim3(nbs), dim3(256), 0, st, A, E, ops, exps, topo, p, eps);
            if (p.k > 1 || !HMM_RS_UNI)
                hipLaunchKernelGGL((k_reduce_sparse<TopoGene7, true>), dim3(nbs), dim3(256), 0, st, A, E, ops, exps, topo, p, eps);
        }
        // the dense kernel: every chain its own wave, unless a sparse kernel may have taken the model
        const bool maybe_sparse = p.q == TopoGene15::Q || p.q == TopoGene7::Q;
        const unsigned nbd = (maybe_sparse && nb > 4096u) ? 4096u : nb;
        hipLaunchKernelGGL(k_reduce, dim3(nbd), dim3(256), 0, st, A, E, ops, exps, (const int *)topo, p, eps);
    }
}

// chunk-level prefix / suffix vectors from the chunk operators.  pre_in .. ls_in (sequence-sharded
// calls): the vectors entering this time slab, in place of the start distribution and of ones.
static void run_scan(const float *pi, const Plan &p, float eps, char *ws, hipStream_t st, Profile *pr,
                     const float *pre_in = nullptr, const double *ll_in = nullptr, const float *suf_in = nullptr,
                     const double *ls_in = nullptr) {
    float *ops = (float *)(ws + p.o_ops);
    int *exps = (int *)(ws + p.o_exps);
    int *topo = (int *)(ws + p.o_topo);
    {
        Timed t(pr, HMM_KERNEL_SCAN, st);
        const bool two = p.G > 0 && opt(HMM_OPT_SCAN2) != 0;
        if (!two) {
            hipLaunchKernelGGL(k_scan, dim3(p.NB), dim3(128), 0, st, pi, ops, exps, (float *)(ws + p.o_prefix),
                               (double *)(ws + p.o_llpre), (float *)(ws + p.o_suffix), (double *)(ws + p.o_lsuf),
                               (double *)(ws + p.o_loglik), (const int *)topo, p, eps, pre_in, ll_in, suf_in, ls_in);
        } else {
            float *gops = (float *)(ws + p.o_gops);
            int *gexps = (int *)(ws + p.o_gexps);
            const long long nwv = (long long)p.NB * p.G;
            hipLaunchKernelGGL(k_scan_compose, dim3((unsigned)((nwv + 3) / 4)), dim3(256), 0, st, (const float *)ops,
                               (const int *)exps, gops, gexps, (const int *)topo, p);
            Plan pg = p;                  // the same scan, over the group operators
            pg.C = p.G;
            hipLaunchKernelGGL(k_scan, dim3(p.NB), dim3(128), 0, st, pi, (const float *)gops, (const int *)gexps,
                               (float *)(ws + p.o_gprefix), (double *)(ws + p.o_gllpre), (float *)(ws + p.o_gsuffix),
                               (double *)(ws + p.o_glsuf), (double *)(ws + p.o_loglik), (const int *)topo, pg, eps,
                               pre_in, ll_in, suf_in, ls_in);
            hipLaunchKernelGGL(k_scan_inner, dim3((unsigned)((nwv + 3) / 4)), dim3(128), 0, st, (const float *)ops, (const int *)exps,
                               (const float *)(ws + p.o_gprefix), (const double *)(ws + p.o_gllpre),
                               (const float *)(ws + p.o_gsuffix), (const double *)(ws + p.o_glsuf),
                               (float *)(ws + p.o_prefix), (double *)(ws + p.o_llpre), (float *)(ws + p.o_suffix),
                               (double *)(ws + p.o_lsuf), (const int *)topo, p, eps);
        }
    }
}

static int run_reduce_scan(const float *A, const float *pi, const float *E, const Plan &p, float eps,
                           char *ws, hipStream_t st, Profile *pr = nullptr) {
    run_reduce(A, E, p, eps, ws, st, pr, opt(HMM_OPT_EXACT));
    run_scan(pi, p, eps, ws, st, pr);
    return check_launch();
}

static Routing routing(const Plan &p, char *ws, bool, bool count) {
    Routing rt;
    rt.topo = (const int *)(ws + p.o_topo);
    rt.exact_mode = opt(HMM_OPT_EXACT);
    rt.nexact = count ? (int *)(ws + p.o_nexact) : nullptr;
    rt.flags = nullptr;
    return rt;
}

static int win_margin(const Plan &p) { return (WIN_MARGIN_STEPS + p.T - 1) / p.T; }

static long long apply_waves(const Plan &p) {
    const long long per_model = (long long)p.b * p.C;
    return (long long)p.k * ((per_model + p.cpw - 1) / p.cpw);
}

// ---- batch groups for the posterior pipeline.  The sparse reduce kernel is VALU-bound and the
// apply kernels are HBM-bound (measured: removing all arithmetic from them changes their time by
// 3 %), so a large batch is cut into groups and reduce(g+1) runs on a second stream underneath
// forward/backward(g).  Groups are independent sub-problems (sequences never interact); all use
// the chunk length of the whole problem, so results do not depend on the grouping.
#ifndef HMM_MAX_GROUPS
#define HMM_MAX_GROUPS 16
#endif
#define MAX_GROUPS HMM_MAX_GROUPS
#ifndef HMM_GROUP_MIN_SEQ
#define HMM_GROUP_MIN_SEQ 64
#endif
struct Groups {
    int n;                      // number of groups (1 = no pipelining)
    int T;                      // chunk length shared by all groups
    int b0[MAX_GROUPS + 1];     // group g owns sequences [b0[g], b0[g+1])
    Plan plan[MAX_GROUPS];
    size_t off[MAX_GROUPS];     // workspace offset of group g
    size_t total;
};

static int plan_groups(int k, int b, int L, int q, Groups *G) {
    Plan whole;
    int rc = make_plan(HMM_OP_POSTERIOR, k, b, L, q, &whole);
    if (rc) return rc;
    int n = 1;
    if (k == 1 && (long long)b * L >= (1ll << 24)) {
        // Measured on MI355X (b=1024, L=1e5): the kernels of the two streams do overlap, but each
        // slows down by as much as it overlaps (7.67 ms with 1 group, 7.62 / 7.80 / 7.92 with
        // 2 / 4 / 8), so the pipeline is off by default and kept as an opt-in knob.
        n = opt(HMM_OPT_GROUPS);
        if (n > b / HMM_GROUP_MIN_SEQ) n = b / HMM_GROUP_MIN_SEQ;
        if (n > MAX_GROUPS) n = MAX_GROUPS;
        if (n < 1) n = 1;
    }
    G->n = n;
    G->T = whole.T;
    size_t off = 0;
    for (int g = 0; g < n; ++g) {
        G->b0[g] = (int)((long long)b * g / n);
        G->b0[g + 1] = (int)((long long)b * (g + 1) / n);
        if ((rc = make_plan(HMM_OP_POSTERIOR, k, G->b0[g + 1] - G->b0[g], L, q, &G->plan[g], whole.T))) return rc;
        G->off[g] = off;
        off += G->plan[g].total;
    }
    G->total = off;
    return HMM_OK;
}

// two helper streams per device, created on first use and kept for the life of the process
static hipStream_t *helper_streams() {
    static hipStream_t pool[64][2];
    static bool ready[64];
    static std::mutex mu;                      // entry points may be called from several host threads
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ready[dev]) {
        if (hipStreamCreateWithFlags(&pool[dev][0], hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipStreamCreateWithFlags(&pool[dev][1], hipStreamNonBlocking) != hipSuccess) return nullptr;
        ready[dev] = true;
    }
    return pool[dev];
}

static int check_ws(const Plan &p, void *ws, size_t bytes) {
    if (!ws) return HMM_ERR_NULL_POINTER;
    if (bytes < p.total || ((uintptr_t)ws & 255)) return HMM_ERR_WORKSPACE;
    return HMM_OK;
}

#include "hmm_largeq.inc"
#include "hmm_midq.inc"
#include "hmm_scan32.inc"
#include "hmm_scan64.inc"

extern "C" {

const char *hmm_strerror(int code) {
    switch (code) {
        case HMM_OK: return "ok";
        case HMM_ERR_BAD_SHAPE: return "bad shape (k, b, L, q must be >= 1)";
        case HMM_ERR_Q_UNSUPPORTED: return "number of states not supported by this build (q <= 4096; Viterbi q <= 16)";
        case HMM_ERR_NULL_POINTER: return "null pointer";
        case HMM_ERR_WORKSPACE: return "workspace too small or not 256-byte aligned";
        case HMM_ERR_LAUNCH: return "HIP kernel launch failed";
        case HMM_ERR_BAD_ARGUMENT: return "bad argument";
        case HMM_ERR_NO_DEVICE: return "no HIP device";
        case HMM_ERR_NO_RCCL: return "ncclAllReduce not found: the host process has not loaded RCCL";
        default: return "unknown error";
    }
}

int hmm_abi_version(void) { return HMM_ENGINE_ABI_VERSION; }

int hmm_set_option(int option, int value) {
    if (option < 0 || option >= HMM_OPT_COUNT) return HMM_ERR_BAD_ARGUMENT;
    (void)opt(option);                                       // seeds the table on first use
    return g_opt[option].exchange(value);
}
int hmm_get_option(int option) {
    if (option < 0 || option >= HMM_OPT_COUNT) return HMM_ERR_BAD_ARGUMENT;
    return opt(option);
}
int hmm_max_states(void) { return HMM_LARGEQ_MAX; }
int hmm_scan_max_states(void) { return QP; }

int hmm_largeq_tile_cols(int b, int q) {
    if (b < 1 || q <= MQ_MAX || q > HMM_LARGEQ_MAX) return 0;
    return 16 * lq_tile_width(b, q);
}

int hmm_chunk_len(int k, int b, int L, int q) {
    if (q > QP) return q > HMM_LARGEQ_MAX ? HMM_ERR_Q_UNSUPPORTED : 0;     // 0: serial in time, no chunks
    Plan p;
    int rc = make_plan(HMM_OP_LOGLIK, k, b, L, q, &p);
    return rc ? rc : p.T;
}

size_t hmm_workspace_bytes(int op, int k, int b, int L, int q) {
    if (q > QP) {
        LqPlan lp;
        if (make_lqplan(k, b, L, q, &lp)) return 0;
        if (q <= Q32 && op != HMM_OP_VITERBI) {                               // + the chunked scan's region
            Plan32 p32;
            if (make_plan32(op, k, b, L, q, &p32)) return 0;
            return lp.total + p32.total;
        }
        if (scan64_wanted(k, b, L, q) && op != HMM_OP_VITERBI) {               // few sequences of 33..64 states: likewise
            Plan64 p64;
            if (make_plan64(op, k, b, L, q, &p64)) return 0;
            return lp.total + p64.total;
        }
        return lp.total;
    }
    if (op == HMM_OP_POSTERIOR) {
        Groups G;
        if (plan_groups(k, b, L, q, &G)) return 0;
        return G.total;
    }
    Plan p;
    if (make_plan(op, k, b, L, q, &p)) return 0;
    return p.total;
}

static int lq_check(const LqPlan &lp, void *ws, size_t bytes) {
    if (!ws) return HMM_ERR_NULL_POINTER;
    if (bytes < lp.total || ((uintptr_t)ws & 255)) return HMM_ERR_WORKSPACE;
    return HMM_OK;
}

int hmm_forward(const float *A, const float *pi, const float *E, int k, int b, int L, int q, float eps,
                float *log_alpha, double *loglik, void *workspace, size_t workspace_bytes, void *stream) {
    if (q > QP) {
        LqPlan lp;
        int rc = make_lqplan(k, b, L, q, &lp);
        if (rc) return rc;
        if (!A || !pi || !E || !loglik) return HMM_ERR_NULL_POINTER;
        if ((rc = lq_check(lp, workspace, workspace_bytes))) return rc;
        char *ws = (char *)workspace;
        if (q <= Q32) {
            // 17..32 states: chunk operators + chunk scan (+ the forward apply kernel for log alpha) for the
            // models the chunked path serves (decided on the device), one wave per sequence for the others
            Plan32 p32;
            if ((rc = make_plan32(log_alpha ? HMM_OP_FORWARD : HMM_OP_LOGLIK, k, b, L, q, &p32))) return rc;
            if (workspace_bytes < lp.total + p32.total) return HMM_ERR_WORKSPACE;
            if (log_alpha) scan32_forward(A, pi, E, p32, eps, log_alpha, ws + lp.total, (hipStream_t)stream);
            else scan32_loglik(A, pi, E, p32, eps, ws + lp.total, (hipStream_t)stream);
            mq_forward(A, pi, E, k, b, L, q, eps, nullptr, log_alpha, (double *)(ws + lp.total + p32.o_loglik),
                       (hipStream_t)stream, (const int *)(ws + lp.total + p32.o_need), (MqSp *)(ws + lp.o_sp));
            hipLaunchKernelGGL(k_copy_loglik, dim3((lp.NB + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                               (const double *)(ws + lp.total + p32.o_loglik), loglik, lp.NB);
            return check_launch();
        }
        if (scan64_wanted(k, b, L, q)) {
            // 33..64 states, few sequences: the chunked scan for primitive models, one wave per sequence for the rest
            Plan64 p64;
            if ((rc = make_plan64(log_alpha ? HMM_OP_FORWARD : HMM_OP_LOGLIK, k, b, L, q, &p64))) return rc;
            if (workspace_bytes < lp.total + p64.total) return HMM_ERR_WORKSPACE;
            if (log_alpha) scan64_forward(A, pi, E, p64, eps, log_alpha, ws + lp.total, (hipStream_t)stream);
            else scan64_loglik(A, pi, E, p64, eps, ws + lp.total, (hipStream_t)stream);
            mq_forward(A, pi, E, k, b, L, q, eps, nullptr, log_alpha, (double *)(ws + lp.total + p64.o_loglik),
                       (hipStream_t)stream, (const int *)(ws + lp.total + p64.o_need), (MqSp *)(ws + lp.o_sp));
            hipLaunchKernelGGL(k_copy_loglik, dim3((lp.NB + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                               (const double *)(ws + lp.total + p64.o_loglik), loglik, lp.NB);
            return check_launch();
        }
        if (q <= MQ_MAX)                                     // one wave per sequence, no launches per step
            mq_forward(A, pi, E, k, b, L, q, eps, nullptr, log_alpha, (double *)(ws + lp.o_ll), (hipStream_t)stream, nullptr,
                       (MqSp *)(ws + lp.o_sp));
        else
            lq_forward(A, pi, E, lp, eps, ws, log_alpha, (hipStream_t)stream);
        hipLaunchKernelGGL(k_copy_loglik, dim3((lp.NB + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const double *)(ws + lp.o_ll), loglik, lp.NB);
        return check_launch();
    }
    Plan p;
    int rc = make_plan(log_alpha ? HMM_OP_FORWARD : HMM_OP_LOGLIK, k, b, L, q, &p);
    if (rc) return rc;
    if (!A || !pi || !E || !loglik) return HMM_ERR_NULL_POINTER;
    if ((rc = check_ws(p, workspace, workspace_bytes))) return rc;
    char *ws = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    Plan px;
    if ((rc = make_xplan(p, &px))) return rc;
    if ((rc = run_reduce_scan(A, pi, E, p, eps, ws, st))) return rc;
    double *wll = (double *)(ws + p.o_loglik);
    // Routing.  Per model: k_topo_check.  Per sequence: there is no backward pass here, so the scan plan's forward
    // kernel itself carries the clamp-born part of alpha_hat along and weighs it with the chunk scan's suffix
    // vectors (forward_body's CERT); sequences whose sum is above EXACT_DELTA are walked whole by the serial plan.
    // The log-likelihood alone comes out of the chunk scan, so for it the kernel runs for the verdict only.
    const Routing rt = routing(p, ws, false, false);
    Routing rtx = routing(p, ws, false, false);
    const long long nw = apply_waves(p), nwx = apply_waves(px);
    const dim3 grid((unsigned)((nw + 3) / 4)), gx((unsigned)((nwx + 3) / 4));
    float *psi = (float *)(ws + p.o_phi);
    int *flags = (int *)(ws + p.o_flags);
    const float *pre = (const float *)(ws + p.o_prefix), *suf = (const float *)(ws + p.o_suffix);
    const double *llp = (const double *)(ws + p.o_llpre);
    const bool cert = rt.exact_mode == HMM_EXACT_AUTO;
    if (log_alpha) {
        if (cert)
            hipLaunchKernelGGL((k_forward<false, true, false, true>), grid, dim3(256), 0, st, A, pi, E, pre, llp, (float *)nullptr,
                               log_alpha, wll, (float *)(ws + p.o_xend), rt, p, eps, nw, psi, suf);
        else
            hipLaunchKernelGGL((k_forward<false, true, false>), grid, dim3(256), 0, st, A, pi, E, pre, llp, (float *)nullptr,
                               log_alpha, wll, (float *)nullptr, rt, p, eps, nw);
    } else if (cert) {
        hipLaunchKernelGGL((k_forward<false, false, false, true>), grid, dim3(256), 0, st, A, pi, E, pre, llp, (float *)nullptr,
                           (float *)nullptr, wll, (float *)(ws + p.o_xend), rt, p, eps, nw, psi, suf);
    }
    // Routed sequences: windows, forward half only — for the log-likelihood alone, and for log alpha, whose rows
    // after a window then move with the window's log-likelihood (k_window_shift_loga); what the windows cannot
    // settle is walked whole
    hipLaunchKernelGGL(k_exact_select, dim3(p.NB), dim3(64), 0, st, rt.topo, psi, p, rt.exact_mode,
                       win_margin(p), flags, (int *)(ws + p.o_nexact), (int *)(ws + p.o_wtab),
                       (int *)(ws + p.o_wlist), (int *)(ws + p.o_wcnt), (const int *)(ws + p.o_exps));
    if (log_alpha) {
        const unsigned gw = (unsigned)((p.NB < 4096 ? p.NB : 4096) + 3) / 4;
        double *wsh = (double *)(ws + p.o_wshift);
        hipLaunchKernelGGL((k_window_posterior<5>), dim3(gw), dim3(256), 0, st, A, E, pre, llp, suf,
                           (const float *)(ws + p.o_xend), (const float *)nullptr, (float *)nullptr, wll, log_alpha,
                           (int *)(ws + p.o_wtab), (const int *)(ws + p.o_wlist), (int *)(ws + p.o_wcnt), flags,
                           (double *)(ws + p.o_dfix), p, eps, win_margin(p), wsh);
        hipLaunchKernelGGL(k_window_shift_loga, dim3(64, 64), dim3(256), 0, st, log_alpha, (const int *)(ws + p.o_wtab),
                           (const int *)(ws + p.o_wlist), (const int *)(ws + p.o_wcnt), (const int *)flags,
                           (const double *)wsh, p);
    } else {
        const unsigned gw = (unsigned)((p.NB < 4096 ? p.NB : 4096) + 3) / 4;
        hipLaunchKernelGGL((k_window_posterior<4>), dim3(gw), dim3(256), 0, st, A, E, pre, llp, suf,
                           (const float *)(ws + p.o_xend), (const float *)nullptr, (float *)nullptr, wll, (float *)nullptr,
                           (int *)(ws + p.o_wtab), (const int *)(ws + p.o_wlist), (int *)(ws + p.o_wcnt), flags,
                           (double *)(ws + p.o_dfix), p, eps, win_margin(p));
    }
    rtx.flags = flags;
    if (log_alpha)
        hipLaunchKernelGGL((k_forward<false, true, true>), gx, dim3(256), 0, st, A, pi, E, (const float *)nullptr,
                           (const double *)nullptr, (float *)nullptr, log_alpha, wll, (float *)nullptr, rtx, px, eps, nwx);
    else
        hipLaunchKernelGGL((k_forward<false, false, true>), gx, dim3(256), 0, st, A, pi, E, (const float *)nullptr,
                           (const double *)nullptr, (float *)nullptr, (float *)nullptr, wll, (float *)nullptr, rtx, px,
                           eps, nwx);
    hipLaunchKernelGGL(k_copy_loglik, dim3((p.NB + 255) / 256), dim3(256), 0, st, (const double *)wll, loglik, p.NB);
    return check_launch();
}

int hmm_backward(const float *A, const float *E, int k, int b, int L, int q, float eps, float *log_beta,
                 void *workspace, size_t workspace_bytes, void *stream) {
    if (q > QP) {
        LqPlan lp;
        int rc = make_lqplan(k, b, L, q, &lp);
        if (rc) return rc;
        if (!A || !E || !log_beta) return HMM_ERR_NULL_POINTER;
        if ((rc = lq_check(lp, workspace, workspace_bytes))) return rc;
        if (q <= Q32) {
            Plan32 p32;
            if ((rc = make_plan32(HMM_OP_BACKWARD, k, b, L, q, &p32))) return rc;
            if (workspace_bytes < lp.total + p32.total) return HMM_ERR_WORKSPACE;
            char *w32 = (char *)workspace + lp.total;
            scan32_backward(A, E, p32, eps, log_beta, w32, (hipStream_t)stream);
            mq_backward(A, E, k, b, L, q, eps, log_beta, nullptr, 3, (hipStream_t)stream, (const int *)(w32 + p32.o_need),
                        (MqSp *)((char *)workspace + lp.o_sp));
            return check_launch();
        }
        if (scan64_wanted(k, b, L, q)) {
            Plan64 p64;
            if ((rc = make_plan64(HMM_OP_BACKWARD, k, b, L, q, &p64))) return rc;
            if (workspace_bytes < lp.total + p64.total) return HMM_ERR_WORKSPACE;
            char *w64 = (char *)workspace + lp.total;
            scan64_backward(A, E, p64, eps, log_beta, w64, (hipStream_t)stream);
            mq_backward(A, E, k, b, L, q, eps, log_beta, nullptr, 3, (hipStream_t)stream, (const int *)(w64 + p64.o_need),
                        (MqSp *)((char *)workspace + lp.o_sp));
            return check_launch();
        }
        if (q <= MQ_MAX)
            mq_backward(A, E, k, b, L, q, eps, log_beta, nullptr, 3, (hipStream_t)stream, nullptr,
                        (MqSp *)((char *)workspace + lp.o_sp));
        else
            lq_backward(A, E, lp, eps, (char *)workspace, log_beta, (hipStream_t)stream);
        return check_launch();
    }
    Plan p;
    int rc = make_plan(HMM_OP_BACKWARD, k, b, L, q, &p);
    if (rc) return rc;
    if (!A || !E || !log_beta) return HMM_ERR_NULL_POINTER;
    if ((rc = check_ws(p, workspace, workspace_bytes))) return rc;
    char *ws = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    // hmm_backward has no start distribution: the chunk scan's forward half (whose vectors weigh the certificate,
    // backward_body's CERT3) starts from the uniform one
    Plan px;
    if ((rc = make_xplan(p, &px))) return rc;
    float *upi = (float *)(ws + p.o_upi);
    {
        const float u = 1.0f / (float)q;
        if (hipMemsetD32Async((hipDeviceptr_t)upi, __builtin_bit_cast(int, u), (size_t)k * q, st) != hipSuccess)
            return HMM_ERR_LAUNCH;
    }
    if ((rc = run_reduce_scan(A, upi, E, p, eps, ws, st))) return rc;
    const Routing rt = routing(p, ws, false, false);
    Routing rtx = routing(p, ws, false, false);
    const long long nw = apply_waves(p), nwx = apply_waves(px);
    float *psi = (float *)(ws + p.o_phi);
    int *flags = (int *)(ws + p.o_flags);
    if (rt.exact_mode == HMM_EXACT_AUTO)
        hipLaunchKernelGGL((k_backward<3, false, true>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, A, E,
                           (const float *)nullptr, (const float *)(ws + p.o_suffix), (const double *)(ws + p.o_lsuf),
                           (const double *)(ws + p.o_loglik), log_beta, psi, (float *)(ws + p.o_rstart), rt, p, eps, nw,
                           (const float *)(ws + p.o_prefix));
    else
        hipLaunchKernelGGL((k_backward<3, false>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, A, E,
                           (const float *)nullptr, (const float *)(ws + p.o_suffix), (const double *)(ws + p.o_lsuf),
                           (const double *)(ws + p.o_loglik), log_beta, (float *)nullptr, (float *)nullptr, rt, p, eps, nw);
    hipLaunchKernelGGL(k_exact_select, dim3(p.NB), dim3(64), 0, st, rt.topo, psi, p, rt.exact_mode,
                       win_margin(p), flags, (int *)(ws + p.o_nexact), (int *)(ws + p.o_wtab), (int *)(ws + p.o_wlist),
                       (int *)(ws + p.o_wcnt), (const int *)(ws + p.o_exps));
    if (rt.exact_mode == HMM_EXACT_AUTO) {                  // routed sequences: windows (k_window_logbeta), the rest whole
        const unsigned gw = (unsigned)((p.NB < 4096 ? p.NB : 4096) + 3) / 4;
        double *wsh = (double *)(ws + p.o_wshift);
        hipLaunchKernelGGL(k_window_logbeta, dim3(gw), dim3(256), 0, st, A, E, (const float *)(ws + p.o_prefix),
                           (const float *)(ws + p.o_suffix), (const double *)(ws + p.o_lsuf),
                           (const float *)(ws + p.o_rstart), log_beta, (int *)(ws + p.o_wtab),
                           (const int *)(ws + p.o_wlist), (int *)(ws + p.o_wcnt), flags, wsh, p, eps, win_margin(p));
        hipLaunchKernelGGL(k_window_shift_logb, dim3(64, 64), dim3(256), 0, st, log_beta, (const int *)(ws + p.o_wtab),
                           (const int *)(ws + p.o_wlist), (const int *)(ws + p.o_wcnt), (const int *)flags,
                           (const double *)wsh, p);
    }
    rtx.flags = flags;
    hipLaunchKernelGGL((k_backward<3, true>), dim3((unsigned)((nwx + 3) / 4)), dim3(256), 0, st, A, E,
                       (const float *)nullptr, (const float *)nullptr, (const double *)nullptr,
                       (const double *)(ws + p.o_loglik), log_beta, (float *)nullptr, (float *)nullptr, rtx, px, eps, nwx);
    return check_launch();
}

static int launch_apply(const float *A, const float *pi, const float *E, const Plan &p, float eps, int mode, char *ws,
                        float *out, double *loglik, hipStream_t st, Profile *pr, bool allow_exact = true) {
    Plan px;
    int rc = make_xplan(p, &px);
    if (rc) return rc;
    const long long nw = apply_waves(p), nwx = apply_waves(px);
    const dim3 grid((unsigned)((nw + 3) / 4)), gx((unsigned)((nwx + 3) / 4));
    float *ckpt = (float *)(ws + p.o_ckpt);
    float *psi = (float *)(ws + p.o_phi);
    float *xend = (float *)(ws + p.o_xend), *rstart = (float *)(ws + p.o_rstart);
    double *ll = (double *)(ws + p.o_loglik);
    const Routing rt = routing(p, ws, false, false);
    // The scan plan's forward / backward pair agrees on its own block length: checkpoints every HMM_POST_BLOCK
    // steps for the probability output (half the checkpoint traffic: k_forward 1.38 -> 1.24 ms in a one-process A/B;
    // 16 recomputed alpha_hat rows fit k_backward<0>'s register file at two waves per SIMD, the log modes' do not)
    const bool wide = HMM_POST_BLOCK != SUB && mode == HMM_POST_PROB && p.T % HMM_POST_BLOCK == 0;
    Plan pb = p;
    if (wide) pb.nsub = p.T / HMM_POST_BLOCK;
    {
        Timed t(pr, HMM_KERNEL_FORWARD, st);
        if (wide)
            hipLaunchKernelGGL((k_forward<true, false, false, false, HMM_POST_BLOCK>), grid, dim3(256), 0, st, A, pi, E,
                               (const float *)(ws + p.o_prefix), (const double *)(ws + p.o_llpre), ckpt, (float *)nullptr,
                               ll, xend, rt, pb, eps, nw);
        else
            hipLaunchKernelGGL((k_forward<true, false, false>), grid, dim3(256), 0, st, A, pi, E,
                               (const float *)(ws + p.o_prefix), (const double *)(ws + p.o_llpre), ckpt, (float *)nullptr,
                               ll, xend, rt, p, eps, nw);
    }
    const float *sx = (const float *)(ws + p.o_suffix);
    const double *ls = (const double *)(ws + p.o_lsuf);
    {
        Timed t(pr, HMM_KERNEL_BACKWARD, st);
        if (mode == HMM_POST_PROB && wide)
            hipLaunchKernelGGL((k_backward<0, false, false, HMM_POST_BLOCK>), grid, dim3(256), 0, st, A, E, (const float *)ckpt,
                               sx, ls, (const double *)ll, out, psi, rstart, rt, pb, eps, nw);
        else if (mode == HMM_POST_PROB)
            hipLaunchKernelGGL((k_backward<0, false>), grid, dim3(256), 0, st, A, E, (const float *)ckpt, sx, ls,
                               (const double *)ll, out, psi, rstart, rt, p, eps, nw);
        else if (mode == HMM_POST_LOG)
            hipLaunchKernelGGL((k_backward<1, false>), grid, dim3(256), 0, st, A, E, (const float *)ckpt, sx, ls,
                               (const double *)ll, out, psi, rstart, rt, p, eps, nw);
        else
            hipLaunchKernelGGL((k_backward<2, false>), grid, dim3(256), 0, st, A, E, (const float *)ckpt, sx, ls,
                               (const double *)ll, out, psi, rstart, rt, p, eps, nw);
    }
    if (allow_exact) {
        // the serial kernels: per model as k_topo_check decided, per sequence from the clamp-born mass the
        // backward kernel just summed; their waves exit at once when nothing is routed
        Timed t(pr, HMM_KERNEL_EXACT, st);
        Routing rtx = routing(p, ws, true, false);
        int *flags = (int *)(ws + p.o_flags);
        int *wtab = (int *)(ws + p.o_wtab), *wlist = (int *)(ws + p.o_wlist), *wcnt = (int *)(ws + p.o_wcnt);
        double *dfix = (double *)(ws + p.o_dfix);
        hipLaunchKernelGGL(k_exact_select, dim3(p.NB), dim3(64), 0, st, rtx.topo, psi, p, rtx.exact_mode,
                           win_margin(p), flags, (int *)(ws + p.o_nexact), wtab, wlist, wcnt, (const int *)(ws + p.o_exps));
        rtx.flags = flags;
        const unsigned gw = (unsigned)((p.NB < 4096 ? p.NB : 4096) + 3) / 4;
        const float *pre = (const float *)(ws + p.o_prefix);
        const double *llp = (const double *)(ws + p.o_llpre);
        if (mode == HMM_POST_PROB) {
            hipLaunchKernelGGL((k_window_posterior<0>), dim3(gw), dim3(256), 0, st, A, E, pre, llp, sx, (const float *)xend,
                               (const float *)rstart, ckpt, ll, out, wtab, (const int *)wlist, wcnt, flags, dfix, p, eps, win_margin(p));
            hipLaunchKernelGGL((k_exact_posterior<0>), gx, dim3(256), 0, st, A, pi, E, ckpt, ll, out, rtx, px, eps, nwx);
        } else if (mode == HMM_POST_LOG) {
            hipLaunchKernelGGL((k_window_posterior<1>), dim3(gw), dim3(256), 0, st, A, E, pre, llp, sx, (const float *)xend,
                               (const float *)rstart, ckpt, ll, out, wtab, (const int *)wlist, wcnt, flags, dfix, p, eps, win_margin(p));
            hipLaunchKernelGGL((k_exact_posterior<1>), gx, dim3(256), 0, st, A, pi, E, ckpt, ll, out, rtx, px, eps, nwx);
        } else {
            hipLaunchKernelGGL((k_window_posterior<2>), dim3(gw), dim3(256), 0, st, A, E, pre, llp, sx, (const float *)xend,
                               (const float *)rstart, ckpt, ll, out, wtab, (const int *)wlist, wcnt, flags, dfix, p, eps, win_margin(p));
            hipLaunchKernelGGL(k_window_fixll, dim3(64, 64), dim3(256), 0, st, out, (const int *)wlist, (const int *)wcnt,
                               (const int *)flags, (const double *)dfix, p);
            hipLaunchKernelGGL((k_exact_posterior<2>), gx, dim3(256), 0, st, A, pi, E, ckpt, ll, out, rtx, px, eps, nwx);
        }
    }
    if (loglik)
        hipLaunchKernelGGL(k_copy_loglik, dim3((p.NB + 255) / 256), dim3(256), 0, st, (const double *)ll, loglik, p.NB);
    return HMM_OK;
}

static int posterior_impl(const float *A, const float *pi, const float *E, int k, int b, int L, int q, float eps,
                          int mode, float *out, double *loglik, void *workspace, size_t workspace_bytes,
                          void *stream, Profile *pr) {
    if (q > QP) {
        LqPlan lp;
        int rc = make_lqplan(k, b, L, q, &lp);
        if (rc) return rc;
        if (!A || !pi || !E || !out) return HMM_ERR_NULL_POINTER;
        if (mode < HMM_POST_PROB || mode > HMM_POST_LOG_NO_LL) return HMM_ERR_BAD_ARGUMENT;
        if ((rc = lq_check(lp, workspace, workspace_bytes))) return rc;
        char *ws = (char *)workspace;
        hipStream_t st = (hipStream_t)stream;
        if (q <= Q32) {
            // 17..32 states: the chunked scan for the models it serves, the serial kernels for the rest
            // and for the sequences the certificate flags (all decided on the device)
            Plan32 p32;
            if ((rc = make_plan32(HMM_OP_POSTERIOR, k, b, L, q, &p32))) return rc;
            if (workspace_bytes < lp.total + p32.total) return HMM_ERR_WORKSPACE;
            char *w32 = ws + lp.total;
            scan32_posterior(A, pi, E, p32, eps, mode, out, w32, st);
            const int *need = (const int *)(w32 + p32.o_need);
            double *ll = (double *)(w32 + p32.o_loglik);
            if (mode != HMM_POST_LOG_NO_LL && L >= 2) {
                mq_posterior2(A, pi, E, k, b, L, q, eps, out, ll, mode, st, need, (MqSp *)(ws + lp.o_sp));
            } else {
                mq_forward(A, pi, E, k, b, L, q, eps, out, nullptr, ll, st, need, (MqSp *)(ws + lp.o_sp));
                mq_backward(A, E, k, b, L, q, eps, out, (const double *)ll, mode, st, need, (MqSp *)(ws + lp.o_sp));
            }
            if (loglik)
                hipLaunchKernelGGL(k_copy_loglik, dim3((lp.NB + 255) / 256), dim3(256), 0, st, (const double *)ll, loglik,
                                   lp.NB);
            return check_launch();
        }
        if (scan64_wanted(k, b, L, q)) {
            Plan64 p64;
            if ((rc = make_plan64(HMM_OP_POSTERIOR, k, b, L, q, &p64))) return rc;
            if (workspace_bytes < lp.total + p64.total) return HMM_ERR_WORKSPACE;
            char *w64 = ws + lp.total;
            scan64_posterior(A, pi, E, p64, eps, mode, out, w64, st);
            const int *need = (const int *)(w64 + p64.o_need);
            double *ll = (double *)(w64 + p64.o_loglik);
            if (mode != HMM_POST_LOG_NO_LL)
                mq_posterior2(A, pi, E, k, b, L, q, eps, out, ll, mode, st, need, (MqSp *)(ws + lp.o_sp));
            else {
                mq_forward(A, pi, E, k, b, L, q, eps, out, nullptr, ll, st, need, (MqSp *)(ws + lp.o_sp));
                mq_backward(A, E, k, b, L, q, eps, out, (const double *)ll, mode, st, need, (MqSp *)(ws + lp.o_sp));
            }
            if (loglik)
                hipLaunchKernelGGL(k_copy_loglik, dim3((lp.NB + 255) / 256), dim3(256), 0, st, (const double *)ll, loglik,
                                   lp.NB);
            return check_launch();
        }
        if (q <= MQ_MAX && mode != HMM_POST_LOG_NO_LL && L >= 2) {
            // forward and backward waves side by side, meeting in the middle
            mq_posterior2(A, pi, E, k, b, L, q, eps, out, (double *)(ws + lp.o_ll), mode, st, nullptr, (MqSp *)(ws + lp.o_sp));
        } else if (q <= MQ_MAX) {
            mq_forward(A, pi, E, k, b, L, q, eps, out, nullptr, (double *)(ws + lp.o_ll), st, nullptr, (MqSp *)(ws + lp.o_sp));   // alpha_hat parked in `out`
            mq_backward(A, E, k, b, L, q, eps, out, (const double *)(ws + lp.o_ll), mode, st, nullptr, (MqSp *)(ws + lp.o_sp));
        } else {
            hipStream_t *hs = helper_streams();               // the two recursions side by side (lq_posterior)
            lq_posterior(A, pi, E, lp, eps, ws, out, mode, st, hs ? hs[0] : nullptr);
        }
        if (loglik)
            hipLaunchKernelGGL(k_copy_loglik, dim3((lp.NB + 255) / 256), dim3(256), 0, st,
                               (const double *)(ws + lp.o_ll), loglik, lp.NB);
        return check_launch();
    }
    Groups G;
    int rc = plan_groups(k, b, L, q, &G);
    if (rc) return rc;
    if (!A || !pi || !E || !out) return HMM_ERR_NULL_POINTER;
    if (mode < HMM_POST_PROB || mode > HMM_POST_LOG_NO_LL) return HMM_ERR_BAD_ARGUMENT;
    if (!workspace) return HMM_ERR_NULL_POINTER;
    if (workspace_bytes < G.total || ((uintptr_t)workspace & 255)) return HMM_ERR_WORKSPACE;
    char *ws = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
#ifdef HMM_GROUPS_SERIAL
    hipStream_t *hs = nullptr;
#else
    hipStream_t *hs = G.n > 1 ? helper_streams() : nullptr;
#endif
    if (G.n == 1 || !hs) {
        // single group (or no helper streams available): everything in order on the caller's stream
        for (int g = 0; g < G.n; ++g) {
            const Plan &p = G.plan[g];
            const size_t row = (size_t)G.b0[g] * L * q;
            if ((rc = run_reduce_scan(A, pi, E + row, p, eps, ws + G.off[g], st, pr))) return rc;
            if ((rc = launch_apply(A, pi, E + row, p, eps, mode, ws + G.off[g], out + row,
                                   loglik ? loglik + G.b0[g] : nullptr, st, pr))) return rc;
        }
        return check_launch();
    }
    // fork: helper stream 0 runs reduce+scan of every group back to back, helper stream 1 runs
    // forward+backward of group g as soon as its reduce+scan is done; join back into `st`
    hipEvent_t ev_fork, ev_red[MAX_GROUPS], ev_join[2];
    (void)hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
    (void)hipEventRecord(ev_fork, st);
    (void)hipStreamWaitEvent(hs[0], ev_fork, 0);
    (void)hipStreamWaitEvent(hs[1], ev_fork, 0);
    int ngroups = 0;                            // groups whose event exists (all of them unless a launch failed)
    for (int g = 0; g < G.n && rc == HMM_OK; ++g) {
        const Plan &p = G.plan[g];
        const size_t row = (size_t)G.b0[g] * L * q;
        rc = run_reduce_scan(A, pi, E + row, p, eps, ws + G.off[g], hs[0], pr);
        (void)hipEventCreateWithFlags(&ev_red[g], hipEventDisableTiming);
        (void)hipEventRecord(ev_red[g], hs[0]);
        ngroups = g + 1;
        if (rc != HMM_OK) break;                // still join the helper streams and release the events below
        (void)hipStreamWaitEvent(hs[1], ev_red[g], 0);
        rc = launch_apply(A, pi, E + row, p, eps, mode, ws + G.off[g], out + row, loglik ? loglik + G.b0[g] : nullptr,
                          hs[1], pr);
    }
    for (int i = 0; i < 2; ++i) {
        (void)hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming);
        (void)hipEventRecord(ev_join[i], hs[i]);
        (void)hipStreamWaitEvent(st, ev_join[i], 0);
        (void)hipEventDestroy(ev_join[i]);
    }
    (void)hipEventDestroy(ev_fork);
    for (int g = 0; g < ngroups; ++g) (void)hipEventDestroy(ev_red[g]);
    return rc != HMM_OK ? rc : check_launch();
}

int hmm_posterior(const float *A, const float *pi, const float *E, int k, int b, int L, int q, float eps,
                  int mode, float *out, double *loglik, void *workspace, size_t workspace_bytes, void *stream) {
    return posterior_impl(A, pi, E, k, b, L, q, eps, mode, out, loglik, workspace, workspace_bytes, stream, nullptr);
}

long long hmm_exact_count(int op, int k, int b, int L, int q, const void *workspace, size_t workspace_bytes) {
    if (q > QP && q <= Q32 && op != HMM_OP_VITERBI) {
        // sequences of the last call that the one-wave-per-sequence kernels served
        LqPlan lp;
        Plan32 p32;
        if (make_lqplan(k, b, L, q, &lp) || make_plan32(op, k, b, L, q, &p32)) return HMM_ERR_BAD_SHAPE;
        if (!workspace) return HMM_ERR_NULL_POINTER;
        if (workspace_bytes < lp.total + p32.total) return HMM_ERR_WORKSPACE;
        int v = 0;
        if (hipMemcpy(&v, (const char *)workspace + lp.total + p32.o_nex, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            return HMM_ERR_LAUNCH;
        return v;
    }
    if (scan64_wanted(k, b, L, q) && op != HMM_OP_VITERBI) {
        LqPlan lp;
        Plan64 p64;
        if (make_lqplan(k, b, L, q, &lp) || make_plan64(op, k, b, L, q, &p64)) return HMM_ERR_BAD_SHAPE;
        if (!workspace) return HMM_ERR_NULL_POINTER;
        if (workspace_bytes < lp.total + p64.total) return HMM_ERR_WORKSPACE;
        int v = 0;
        if (hipMemcpy(&v, (const char *)workspace + lp.total + p64.o_nex, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            return HMM_ERR_LAUNCH;
        return v;
    }
    if (q > QP) return 0;                                    // the serial-in-time paths are exact throughout
    if (!workspace) return HMM_ERR_NULL_POINTER;
    long long total = 0;
    auto read = [&](const Plan &p, size_t off) -> int {
        if (workspace_bytes < off + p.total) return HMM_ERR_WORKSPACE;
        int v = 0;
        if (hipMemcpy(&v, (const char *)workspace + off + p.o_nexact, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            return HMM_ERR_LAUNCH;
        total += v;
        return HMM_OK;
    };
    int rc;
    if (op == HMM_OP_POSTERIOR) {
        Groups G;
        if ((rc = plan_groups(k, b, L, q, &G))) return rc;
        for (int g = 0; g < G.n; ++g)
            if ((rc = read(G.plan[g], G.off[g]))) return rc;
        return total;
    }
    Plan p;
    if ((rc = make_plan(op, k, b, L, q, &p))) return rc;
    if ((rc = read(p, 0))) return rc;
    return total;
}

int hmm_exact_detail(int k, int b, int L, int q, const void *workspace, size_t workspace_bytes, long long *detail) {
    if (!workspace || !detail) return HMM_ERR_NULL_POINTER;
    if (q > QP) return HMM_ERR_Q_UNSUPPORTED;
    Groups G;
    int rc = plan_groups(k, b, L, q, &G);
    if (rc) return rc;
    for (int i = 0; i < 5; ++i) detail[i] = 0;
    for (int g = 0; g < G.n; ++g) {
        const Plan &p = G.plan[g];
        if (workspace_bytes < G.off[g] + p.total) return HMM_ERR_WORKSPACE;
        int nx = 0, wc[4] = {0, 0, 0, 0};
        const char *ws = (const char *)workspace + G.off[g];
        if (hipMemcpy(&nx, ws + p.o_nexact, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(wc, ws + p.o_wcnt, sizeof(wc), hipMemcpyDeviceToHost) != hipSuccess)
            return HMM_ERR_LAUNCH;
        detail[0] += nx; detail[1] += wc[0]; detail[2] += wc[2]; detail[3] += wc[1]; detail[4] += wc[3];
    }
    return HMM_OK;
}

int hmm_exact_detail_op(int op, int k, int b, int L, int q, const void *workspace, size_t workspace_bytes,
                        long long *detail) {
    if (op == HMM_OP_POSTERIOR) return hmm_exact_detail(k, b, L, q, workspace, workspace_bytes, detail);
    if (!workspace || !detail) return HMM_ERR_NULL_POINTER;
    if (q > QP) return HMM_ERR_Q_UNSUPPORTED;
    if (op != HMM_OP_LOGLIK && op != HMM_OP_FORWARD && op != HMM_OP_BACKWARD) return HMM_ERR_BAD_ARGUMENT;
    Plan p;
    int rc = make_plan(op, k, b, L, q, &p);
    if (rc) return rc;
    if (workspace_bytes < p.total) return HMM_ERR_WORKSPACE;
    int nx = 0, wc[4] = {0, 0, 0, 0};
    const char *ws = (const char *)workspace;
    if (hipMemcpy(&nx, ws + p.o_nexact, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(wc, ws + p.o_wcnt, sizeof(wc), hipMemcpyDeviceToHost) != hipSuccess)
        return HMM_ERR_LAUNCH;
    detail[0] = nx; detail[1] = wc[0]; detail[2] = wc[2]; detail[3] = wc[1]; detail[4] = wc[3];
    return HMM_OK;
}

int hmm_window_table(int op, int k, int b, int L, int q, const void *workspace, size_t workspace_bytes, int seq,
                     int *table, double *shifts, float *psi, int npsi) {
    if (!workspace || !table) return HMM_ERR_NULL_POINTER;
    if (q > QP) return HMM_ERR_Q_UNSUPPORTED;
    Plan p;
    int rc = make_plan(op, k, b, L, q, &p);
    if (rc) return rc;
    if (workspace_bytes < p.total || seq < 0 || seq >= p.NB) return HMM_ERR_WORKSPACE;
    const char *ws = (const char *)workspace;
    if (hipMemcpy(table, ws + p.o_wtab + (size_t)seq * WIN_STRIDE * sizeof(int), WIN_STRIDE * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        return HMM_ERR_LAUNCH;
    if (shifts && (op == HMM_OP_FORWARD || op == HMM_OP_BACKWARD) &&
        hipMemcpy(shifts, ws + p.o_wshift + (size_t)seq * WSH_STRIDE * sizeof(double), WSH_STRIDE * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return HMM_ERR_LAUNCH;
    if (psi && npsi > 0) {
        const int n = npsi < p.C ? npsi : p.C;
        if (hipMemcpy(psi, ws + p.o_phi + (size_t)seq * p.C * sizeof(float), (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            return HMM_ERR_LAUNCH;
    }
    return p.C;
}

void *hmm_profile_create(void) { return new Profile(); }

void hmm_profile_destroy(void *profile) {
    Profile *pr = (Profile *)profile;
    if (!pr) return;
    for (auto &s : pr->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    delete pr;
}

int hmm_posterior_profiled(const float *A, const float *pi, const float *E, int k, int b, int L, int q, float eps,
                           int mode, float *out, double *loglik, void *workspace, size_t workspace_bytes,
                           void *stream, void *profile) {
    return posterior_impl(A, pi, E, k, b, L, q, eps, mode, out, loglik, workspace, workspace_bytes, stream,
                          (Profile *)profile);
}

int hmm_profile_read(void *profile, double *ms, long long *launches) {
    Profile *pr = (Profile *)profile;
    if (!pr || !ms || !launches) return HMM_ERR_NULL_POINTER;
    for (int i = 0; i < HMM_KERNEL_COUNT; ++i) { ms[i] = 0.0; launches[i] = 0; }
    for (auto &s : pr->spans) {
        if (hipEventSynchronize(s.b) != hipSuccess) return HMM_ERR_LAUNCH;
        float t = 0.f;
        if (hipEventElapsedTime(&t, s.a, s.b) != hipSuccess) return HMM_ERR_LAUNCH;
        ms[s.kernel] += t;
        launches[s.kernel] += 1;
        (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b);
    }
    pr->spans.clear();
    return HMM_OK;
}

// The one collective of the path for hosts without torch.distributed: all-reduce(sum) of the (k,2)
// partials over an RCCL communicator the HOST created.  RCCL is not linked: ncclAllReduce is looked up
// among the libraries the process has already loaded (the host's own RCCL, the one its communicator
// belongs to), so the engine never brings a second copy of the library into the process.
#include <dlfcn.h>
int hmm_loglik_allreduce(void *comm, double *partial, int k, void *stream) {
    typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    if (k < 1) return HMM_ERR_BAD_SHAPE;
    if (!comm || !partial) return HMM_ERR_NULL_POINTER;
    static std::atomic<allreduce_fn> cached{nullptr};
    allreduce_fn fn = cached.load();
    if (!fn) {
        fn = (allreduce_fn)dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!fn) {
            // a host that loaded RCCL without global symbol visibility (dlopen(..., RTLD_LOCAL), as Python extension
            // modules do): ask for the handle of the copy that is ALREADY in the process — RTLD_NOLOAD never loads one
            static const char *names[] = {"librccl.so", "librccl.so.1", "libnccl.so", "libnccl.so.2"};
            for (const char *nm : names) {
                void *h = dlopen(nm, RTLD_NOLOAD | RTLD_NOW);
                if (h && (fn = (allreduce_fn)dlsym(h, "ncclAllReduce"))) break;
            }
        }
        if (!fn) return HMM_ERR_NO_RCCL;
        cached.store(fn);
    }
    const int ncclFloat64 = 8, ncclSum = 0;                  // rccl.h: ncclDataType_t, ncclRedOp_t
    return fn(partial, partial, (size_t)2 * k, ncclFloat64, ncclSum, comm, (hipStream_t)stream) == 0 ? HMM_OK
                                                                                                     : HMM_ERR_LAUNCH;
}

int hmm_loglik_partials(const double *loglik, const float *weights, int k, int b, double *partial, void *stream) {
    if (k < 1 || b < 1) return HMM_ERR_BAD_SHAPE;
    if (!loglik || !partial) return HMM_ERR_NULL_POINTER;
    hipLaunchKernelGGL(k_loglik_partials, dim3(k), dim3(256), 0, (hipStream_t)stream, loglik, weights, b, partial);
    return check_launch();
}

}  // extern "C"

#include "hmm_seqshard.inc"
#include "hmm_viterbi.inc"
#include "hmm_emitter.inc"
#include "hmm_grad.inc"
#include "hmm_postgrad.inc"
