"""Every BASELINE.json configuration on the GPU at its own shape, plus the instantiations and
producers that the shape-generic tests do not reach:

  configs[1]  15-state gene model, b = 256 x L = 10 000, forward only (log-likelihood, log alpha)
  configs[4]  1027-state profile-HMM size at its per-GPU batch b = 1024 (the 80-column GEMM tile),
              and shapes that select the 64- and 96-column tiles
  emitter     hmm_gene_emissions on the reference's own fixture input (tests/golden/emitter.npz)
  transitioner  A built on the device == tests/golden/transitioner.npz bit for bit (incl. the as-shipped D1 mode)
  Viterbi     the Q16 path is within L * 2^-16 * 2 of the unquantised fp64 optimum
(configs[0] is CPU-only: tests/test_modules_cpu.py; configs[2], [3]: test_full_size_* in
tests/test_engine_gpu.py and tests/test_viterbi_gpu.py.)
"""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
from hmm_layer_amd.gene_pred_hmm_transitioner import (
    GenePredHMMTransitioner, GenePredMultiHMMTransitioner, SimpleGenePredHMMTransitioner)
from oracle import build as obuild
from oracle import params, textbook, viterbi as oviterbi

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CODONS = dict(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
              intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)],
              intron_end_pattern=[("AGN", .99), ("ACN", .01)])


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype, device=DEV)


def test_config2_forward_only_b256_L10000():
    """BASELINE configs[1] exactly: the chunk length this shape selects (80) is one no other test uses."""
    torch.manual_seed(2)
    b, L, q = 256, 10000, 15
    assert engine.chunk_len(1, b, L, q) == 80
    A = params.intended_A15().to(DEV)[None]
    pi = torch.full((q,), 1 / q, device=DEV)
    E = torch.rand((1, b, L, q), device=DEV) * 0.9 + 0.05
    _, ll = engine.forward(A, pi, E, want_log_alpha=False)
    la, ll2 = engine.forward(A, pi, E)
    _, ll3 = engine.forward(A, pi, E, want_log_alpha=False)
    la2, _ = engine.forward(A, pi, E)
    torch.cuda.synchronize()
    assert torch.equal(ll, ll2) and torch.equal(ll, ll3) and torch.equal(la, la2)      # deterministic, consistent
    assert bool(torch.isfinite(la).all())
    # the last log alpha row sums (in probability space) to the likelihood
    lse = torch.logsumexp(la[0, :, -1].double(), -1)
    assert float((lse - ll[0]).abs().max()) <= 2e-7 * float(ll.abs().max()) + 3e-4
    idx = [0, 1, 100, 255]
    An, pin = A[0].cpu().numpy(), pi.cpu().numpy()
    Es = E[0, idx].cpu().numpy()
    la64, ll64 = textbook.log_alpha(An, pin, Es)
    assert np.all(np.abs(ll[0, idx].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64))
    got = la[0, idx].cpu().numpy()
    assert np.all(np.abs(got - la64) <= 3e-4 + 2e-7 * np.abs(la64))
    # and the same batch through the posterior pipeline agrees on the log-likelihood
    _, llp = engine.posterior(A, pi, E)
    assert torch.equal(llp, ll)


def profile_like_model(rng, q):
    A = rng.random((q, q)).astype(np.float32) ** 8
    A *= (np.abs(np.subtract.outer(np.arange(q), np.arange(q))) < 40) + 1e-4
    A /= A.sum(-1, keepdims=True)
    pi = rng.random(q).astype(np.float32)
    return A, pi / pi.sum()


@pytest.mark.parametrize("b,q,L,cols", [(1024, 1027, 6, 80), (3328, 344, 5, 96), (192, 1027, 7, 64)])
def test_config5_per_gpu_shape_and_every_gemm_tile(b, q, L, cols):
    """BASELINE configs[4] per GPU: q = 2*512+3 = 1027 states, b = 1024 sequences (the shape bench.py
    times; it selects the 80-column GEMM tile, whose second B-piece staging path the small-batch tests
    never run), plus shapes selecting the 96- and 64-column tiles.  Forward log-likelihood, log alpha,
    log beta and posteriors on the whole batch; the fp64 oracle checks a sample of sequences (the
    sequences of a batch are independent rows of the GEMM)."""
    assert engine.largeq_tile_cols(b, q) == cols
    rng = np.random.default_rng(q + b)
    A, pi = profile_like_model(rng, q)
    E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[rng.random(E.shape) < 0.02] = 0.0
    Ad, pid, Ed = dev(A)[None], dev(pi), dev(E)[None]
    out, ll = engine.posterior(Ad, pid, Ed)
    la, ll2 = engine.forward(Ad, pid, Ed)
    _, ll3 = engine.forward(Ad, pid, Ed, want_log_alpha=False)
    lb = engine.backward(Ad, Ed)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()) and float((out.sum(-1) - 1).abs().max()) <= 2e-5
    assert torch.equal(ll, ll2) and torch.equal(ll, ll3)
    idx = np.unique(np.concatenate([np.arange(0, b, max(1, b // 24)), [b - 1, b - 2, 63, 64, 65]]))
    g64, ll64 = obuild.posterior(A, pi, E[idx])
    assert np.abs(out[0, idx].cpu().numpy() - g64).max() <= 2e-5
    assert np.all(np.abs(ll[0, idx].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    la64, _ = textbook.log_alpha(A, pi, E[idx[:6]])
    lb64 = textbook.log_beta(A, E[idx[:6]])
    got = la[0, idx[:6]].cpu().numpy()
    m = la64 > -30
    assert np.all(np.abs(got - la64)[m] <= 3e-4 + 2e-7 * np.abs(la64[m]))
    got = lb[0, idx[:6]].cpu().numpy()
    m = lb64 > -30
    assert np.all(np.abs(got - lb64)[m] <= 3e-4 + 2e-7 * np.abs(lb64[m]))


def test_fused_emitter_on_the_reference_fixture(golden):
    """hmm_gene_emissions on the input the fixtures were captured with: x -> E (inference), E_training
    (the +1e-7 offset), E_c2 (two copies, unshared intron rows, 29 states) and E_as_shipped (doubled N
    mass in right 3-mers, defect D5) — the reference's own outputs, fp32."""
    g = golden("emitter")
    x = dev(g["x"])
    cases = [(dict(), "kernel", False, "E"), (dict(), "kernel", True, "E_training"),
             (dict(n_mass_compat=True), "kernel", False, "E_as_shipped"),
             (dict(num_copies=2, share_intron_parameters=False), "kernel_c2", False, "E_c2")]
    for kw, kern, training, want in cases:
        em = GenePredHMMEmitter(**CODONS, **kw)
        em.build((1, 2, 40, 15))
        with torch.no_grad():
            em.emission_kernel.copy_(torch.as_tensor(g[kern]))
        em = em.to(DEV)
        em.recurrent_init()
        assert em.can_fuse(x)
        got = em.forward_fused(x, training=training).cpu().numpy()
        ref = g[want]
        assert got.shape == ref.shape, want
        # relative to each row's largest entry: the kernel's MFMA and table products round differently
        # from the reference's matmul over 64 3-mer classes (fp32 both)
        scale = np.abs(ref).max(-1, keepdims=True)
        assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max(), want
        assert (np.abs(got - ref) <= 2e-5 * scale + 1e-12).all(), want


def test_transition_matrices_built_on_the_device(golden):
    """make_A() with the parameters resident on the GPU (a handful of ATen kernels on device index
    buffers, no host round trip) reproduces the matrices captured from the imported reference — the
    intended ones and, with zero_logit_is_absent=True, the as-shipped ones (D1): identical support
    (every structural zero, incl. the all-zero rows) and values to 1 ulp (the device's exp is not the
    CPU's; the same module on CPU tensors is bit-identical, tests/test_producers_cpu.py)."""
    g = golden("transitioner")

    def same(got, ref, key=""):
        got = got.detach().cpu().numpy()
        assert np.array_equal(got == 0, ref == 0), key
        assert np.abs(got - ref).max() <= 1.2e-7 * np.abs(ref).max(), key
        return True

    t7 = SimpleGenePredHMMTransitioner().to(DEV)
    t15 = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000,
                                       starting_distribution_init="zeros").to(DEV)
    assert same(t7.make_A()[0], g["A7"]) and same(t15.make_A()[0], g["A15"])
    assert same(t15.make_initial_distribution().reshape(-1), g["pi15"])
    for cls, kw, key in ((SimpleGenePredHMMTransitioner, {}, "A7_as_shipped"),
                         (GenePredHMMTransitioner, {}, "A15_single_as_shipped"),
                         (GenePredMultiHMMTransitioner, dict(k=2, init_component_sd=0.0), "A29_as_shipped"),
                         (GenePredMultiHMMTransitioner, dict(initial_exon_len=200, initial_intron_len=4500,
                                                             initial_ir_len=10000), "A15_as_shipped")):
        t = cls(zero_logit_is_absent=True, **kw).to(DEV)
        A = t.make_A()
        assert A.is_cuda
        assert same(A[0], g[key], key)
    # log A for the Viterbi entry point comes from the same device-side producer
    logA = t15.make_log_A()
    assert logA.is_cuda and float(logA[t15.make_A() == 0].max()) == -1000.0


def test_viterbi_q16_path_is_optimal_up_to_quantisation():
    """Ties the Q16 definition to real arithmetic: the engine's path, scored under the UNQUANTISED
    fp64 model, is within L * 2^-16 * 2 of the fp64 Viterbi optimum (each of the 2L terms of a path
    score moves by at most 2^-17 under Q, for the optimum and for the returned path alike)."""
    rng = np.random.default_rng(14)
    A = params.intended_A15().numpy().astype(np.float64)
    q, b, L = 15, 4, 5000
    pi = np.full(q, 1 / q)
    E = (rng.random((b, L, q)) * 0.9 + 0.05) / 4096
    dead = rng.random(E.shape) < 0.3
    dead[..., :6] = False
    E[dead] = 1e-12
    with np.errstate(divide="ignore"):
        logA = np.maximum(np.log(A), -1000.0)
    logpi, logE = np.log(pi), np.log(E)
    path, score = engine.viterbi(dev(logA)[None], dev(logpi)[None], dev(logE)[None])
    path, score = path[0].cpu().numpy(), score[0].cpu().numpy()
    # fp64 Viterbi (plain max-plus recursion, no quantisation); absent edges at the engine's -1024 clamp
    lA = np.maximum(logA, -1024.0)
    for s in range(b):
        d = logpi + logE[s, 0]
        for t in range(1, L):
            d = (d[:, None] + lA).max(0) + logE[s, t]
        best = d.max()
        mine = logpi[path[s, 0]] + logE[s, 0, path[s, 0]]
        for t in range(1, L):
            mine += lA[path[s, t - 1], path[s, t]] + logE[s, t, path[s, t]]
        assert mine <= best + 1e-6                    # (two summation orders of 2L fp64 terms)
        assert best - mine <= L * 2.0 ** -16 * 2, (s, best - mine)
        # and the engine's own (quantised) score of that path is the real score up to the same bound
        assert abs(score[s] - mine) <= L * 2.0 ** -16 * 2
        # float32 inputs: the engine saw fp32 roundings of logA / logE (1e-7 relative on |x| <= 30)
