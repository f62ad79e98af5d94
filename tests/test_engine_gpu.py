"""Parity of the HIP engine (through the C ABI) with the oracle.  Needs an MI355X.

Tolerances (fp32 engine vs float64 oracle with the reference's eps clamps):
  posteriors          |gamma - gamma64|      <= 2e-5 absolute (probability space; the eps clamps
                      are non-linear, so values that exist only through clamp paths (< 1e-12 or
                      so) are compared in probability space, not in log space)
  log-likelihood      |ll - ll64|            <= 1e-6 * |ll64| + 2e-4
  log alpha/log beta  |x - x64|              <= 3e-4 + 2e-7*|x64|  where x64 > -30
and against the fixtures captured from the imported reference cell (fp32):  <= 3e-4.
"""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from oracle import params, textbook

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype, device=DEV)


def run_post(A, pi, E, mode=engine.POST_PROB):
    out, ll = engine.posterior(dev(A).reshape(-1, A.shape[-1], A.shape[-1]), dev(pi), dev(E), mode=mode)
    torch.cuda.synchronize()
    return out.cpu().numpy(), ll.cpu().numpy()


def rand_model(rng, q, dense=True):
    A = rng.random((q, q)) ** 3 + 1e-3
    if not dense:
        A *= rng.random((q, q)) < 0.3
        A += np.eye(q) * 0.5
    A /= A.sum(-1, keepdims=True)
    pi = rng.random(q) + 0.1
    pi /= pi.sum()
    return A.astype(np.float32), pi.astype(np.float32)


def assert_log_close_in_probability_space(x, x64, tag=""):
    """log alpha / log beta, EVERY component, compared as probabilities relative to the row's largest value:
    |p - p64| <= 2e-5 + p64 * (3e-4 + 2e-7 |x64|) — the log-space tolerance where a value carries weight, the
    posteriors' absolute tolerance where it exists through the eps clamps only (1e-16 relative or so)."""
    ref = x64.max(-1, keepdims=True)
    with np.errstate(over="ignore", under="ignore"):
        p, p64 = np.exp(np.minimum(x - ref, 50.0)), np.exp(x64 - ref)
    tol = 2e-5 + p64 * (3e-4 + 2e-7 * np.abs(ref))
    bad = np.abs(p - p64) > tol
    assert not bad.any(), (tag, float(np.abs(p - p64).max()), int(bad.sum()))


def check_all(A, pi, E, tag=""):
    """E (b,L,q).  Compares every output of the engine with the fp64 oracle."""
    g64, ll64 = textbook.posterior(A, pi, E)
    la64, _ = textbook.log_alpha(A, pi, E)
    lb64 = textbook.log_beta(A, E)
    E4 = E[None]
    gam, ll = run_post(A, pi, E4)
    assert np.isfinite(gam).all(), tag
    assert np.abs(gam[0] - g64).max() <= 2e-5, (tag, np.abs(gam[0] - g64).max())
    assert np.all(np.abs(ll[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), (tag, ll[0], ll64)
    lg, _ = run_post(A, pi, E4, engine.POST_LOG)
    assert np.abs(np.exp(lg[0]) - g64).max() <= 2e-5, tag
    m = g64 > 1e-4          # log space only where eps-clamp paths cannot dominate the value
    assert np.abs(lg[0] - np.log(np.maximum(g64, 1e-300)))[m].max() <= 1e-3, tag
    la, ll2 = engine.forward(dev(A)[None], dev(pi), dev(E4))
    la, ll2 = la.cpu().numpy()[0], ll2.cpu().numpy()[0]
    m = la64 > -30
    assert np.all(np.abs(la - la64)[m] <= 3e-4 + 2e-7 * np.abs(la64[m])), (tag, np.abs(la - la64)[m].max())
    assert_log_close_in_probability_space(la, la64, tag)
    assert np.array_equal(ll2, ll[0]), tag
    _, ll3 = engine.forward(dev(A)[None], dev(pi), dev(E4), want_log_alpha=False)
    # (identical unless the sequence is routed: the entry points then recompute different stretches of it)
    assert np.allclose(ll3.cpu().numpy()[0], ll[0], rtol=1e-10, atol=0), tag
    lb = engine.backward(dev(A)[None], dev(E4)).cpu().numpy()[0]
    m = lb64 > -30
    assert np.all(np.abs(lb - lb64)[m] <= 3e-4 + 2e-7 * np.abs(lb64[m])), (tag, np.abs(lb - lb64)[m].max())
    assert_log_close_in_probability_space(lb, lb64, tag)
    return gam, ll


@pytest.mark.parametrize("name", ["kat", "cell_q3", "cell_q7", "cell_q15", "cell_q15z"])
def test_golden_fixtures(golden, name):
    """Same inputs as the fixtures captured from the imported reference cell."""
    g = golden(name)
    gam, ll = check_all(g["A"], g["pi"], g["E"], name)
    # against the reference's own fp32 outputs
    assert np.abs(ll[0] - g["loglik"]).max() <= 3e-4
    ref_la = g["fwd"][..., :-1] + g["fwd"][..., -1:]
    ref_lb = g["bwd"][..., :-1] + g["bwd"][..., -1:]
    la, _ = engine.forward(dev(g["A"])[None], dev(g["pi"]), dev(g["E"])[None])
    lb = engine.backward(dev(g["A"])[None], dev(g["E"])[None])
    m = ref_la > -30
    assert np.abs(la.cpu().numpy()[0] - ref_la)[m].max() <= 3e-4
    m = ref_lb > -30
    assert np.abs(lb.cpu().numpy()[0] - ref_lb)[m].max() <= 3e-4
    ref_gam = np.exp(ref_la + ref_lb - g["loglik"][:, None, None])
    assert np.abs(gam[0] - ref_gam).max() <= 1e-4        # the reference's own fp32 formula


def test_known_answer_toy(golden):
    g = golden("kat")
    gam, ll = run_post(g["A"], g["pi"], g["E"][None])
    assert abs(ll[0, 0] - (-3.407610614)) < 2e-6
    np.testing.assert_allclose(gam[0, 0, 0], [0.66464191, 0.10683619, 0.22852191], atol=2e-6)
    np.testing.assert_allclose(gam[0, 0, 3], [0.35321482, 0.50980066, 0.13698451], atol=2e-6)


@pytest.mark.parametrize("q", [1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 15, 16])
def test_state_counts(q):
    rng = np.random.default_rng(q)
    A, pi = rand_model(rng, q)
    E = (rng.random((3, 70, q)) * 0.9 + 0.05).astype(np.float32)
    check_all(A, pi, E, "q=%d" % q)


@pytest.mark.parametrize("b,L", [(1, 1), (1, 2), (2, 15), (1, 16), (3, 17), (5, 31), (2, 33), (17, 100),
                                 (64, 257), (3, 1000), (2, 1025), (1, 3000), (33, 2049)])
def test_ragged_lengths_and_batches(b, L):
    """Lengths that are not multiples of the 16-step block or of the chunk length."""
    rng = np.random.default_rng(1000 * b + L)
    A, pi = rand_model(rng, 15, dense=False)
    E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32)
    check_all(A, pi, E, "b=%d L=%d" % (b, L))


def test_multiple_models():
    rng = np.random.default_rng(7)
    k, b, L, q = 3, 5, 90, 7
    As, pis = zip(*[rand_model(rng, q) for _ in range(k)])
    A, pi = np.stack(As), np.stack(pis)
    E = (rng.random((k, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    out, ll = engine.posterior(dev(A), dev(pi), dev(E))
    out, ll = out.cpu().numpy(), ll.cpu().numpy()
    for m in range(k):
        g64, ll64 = textbook.posterior(A[m], pi[m], E[m])
        assert np.abs(out[m] - g64).max() <= 2e-5
        assert np.abs(ll[m] - ll64).max() <= 2e-4
    # (1,k,q) start distribution as make_initial_distribution() returns it
    out2, _ = engine.posterior(dev(A), dev(pi)[None], dev(E))
    assert torch.equal(out2.cpu(), torch.as_tensor(out))


def test_zero_emissions_hit_the_eps_clamp():
    """Exact zeros in E go through max(E, 1e-16).  The gene emitter produces them all the time
    for the codon-constrained states 6-14 (START/STOP/EI/IE/E2 emit 0 unless the 3-mer fits,
    hmm_layer/gene_pred_hmm_emitter.py:247-258) while states 0-5 always keep mass alive."""
    rng = np.random.default_rng(11)
    A = params.intended_A15().numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((4, 600, 15)) * 0.9 + 0.05).astype(np.float32) / 4096
    dead = rng.random(E.shape) < 0.6
    dead[..., :6] = False
    E[dead] = 0.0
    check_all(A, pi, E, "zeros")


def test_impossible_observations_match_the_serial_recursion():
    """Zeros everywhere (25 % of all entries): whole stretches are impossible under the model
    and every path survives only through the 1e-16 clamps.  The clamp of the mixture is not linear,
    so a scan over chunk operators cannot reproduce the serial recursion there (the reference's own
    parallel_factor > 1 mode differs from its serial mode in the same way); the
    backward kernel's floor-transition certificate sends such sequences to the serial exact-clamp
    kernels (tests/test_exact_gpu.py), so the result is the serial oracle's on EVERY sequence."""
    rng = np.random.default_rng(12)
    A = params.intended_A15().numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((4, 600, 15)) * 0.9 + 0.05).astype(np.float32)
    E[rng.random(E.shape) < 0.25] = 0.0
    gam, ll = run_post(A, pi, E[None])
    assert np.isfinite(gam).all() and np.isfinite(ll).all()
    assert np.abs(gam.sum(-1) - 1).max() < 1e-5 and gam.min() >= 0
    g64, ll64 = textbook.posterior(A, pi, E)
    assert np.abs(gam[0] - g64).max() <= 2e-5
    assert np.all(np.abs(ll[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)


def test_gene_model_long_sequences_vs_fp64():
    """Gene-model magnitudes (E ~ 1e-5 => loglik ~ -1e5 at L = 1e4): the regime in which the
    reference's own fp32 posterior formula breaks down."""
    rng = np.random.default_rng(5)
    A = params.intended_A15().numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    b, L = 6, 10000
    E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32) / 4096
    g64, ll64 = textbook.posterior(A, pi, E)
    gam, ll = run_post(A, pi, E[None])
    assert np.abs(gam[0] - g64).max() <= 2e-5
    assert np.abs(gam[0].sum(-1) - 1).max() <= 1e-5
    assert np.all(np.abs(ll[0] - ll64) <= 1e-6 * np.abs(ll64))
    assert (gam[0].argmax(-1) == g64.argmax(-1)).mean() > 0.9999


def test_log_modes_are_consistent():
    rng = np.random.default_rng(2)
    A, pi = rand_model(rng, 15)
    E = (rng.random((1, 4, 300, 15)) * 0.9 + 0.05).astype(np.float32)
    p, ll = run_post(A, pi, E, engine.POST_PROB)
    lp, _ = run_post(A, pi, E, engine.POST_LOG)
    lq, _ = run_post(A, pi, E, engine.POST_LOG_NO_LL)
    assert np.abs(np.exp(lp) - p).max() < 1e-6
    assert np.abs((lq - lp) - ll[..., None, None]).max() < 1e-3 * (1 + np.abs(ll).max() * 1e-4)


def test_deterministic_and_reentrant():
    rng = np.random.default_rng(3)
    A, pi = rand_model(rng, 15)
    E = dev((rng.random((1, 40, 5000, 15)) * 0.9 + 0.05).astype(np.float32))
    a1, l1 = engine.posterior(dev(A)[None], dev(pi), E)
    a2, l2 = engine.posterior(dev(A)[None], dev(pi), E)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        a3, l3 = engine.posterior(dev(A)[None], dev(pi), E)
    s.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(l1, l2)
    assert torch.equal(a1, a3) and torch.equal(l1, l3)


def test_loglik_partials():
    rng = np.random.default_rng(4)
    ll = rng.standard_normal((3, 1000)) * 50 - 1e5
    w = rng.random((3, 1000)).astype(np.float32)
    p = engine.loglik_partials(dev(ll, torch.float64), dev(w)).cpu().numpy()
    np.testing.assert_allclose(p[:, 0], (ll * w.astype(np.float64)).sum(1), rtol=1e-12)
    np.testing.assert_allclose(p[:, 1], w.astype(np.float64).sum(1), rtol=1e-12)
    p = engine.loglik_partials(dev(ll, torch.float64)).cpu().numpy()
    np.testing.assert_allclose(p[:, 0] / p[:, 1], ll.mean(1), rtol=1e-12)


def test_errors():
    E = np.random.rand(1, 2, 20, 17).astype(np.float32)
    with pytest.raises(ValueError, match="exceeds"):
        engine.posterior(torch.zeros((1, 4097, 4097), device=DEV), torch.zeros(4097, device=DEV),
                         torch.zeros((1, 1, 2, 4097), device=DEV))
    A, pi = rand_model(np.random.default_rng(0), 5)
    with pytest.raises(ValueError, match="shape"):
        engine.posterior(dev(A)[None], dev(pi), dev(E[..., :4]))
    with pytest.raises(engine.EngineError, match="HIP device"):
        engine.posterior(dev(A)[None], dev(pi), torch.rand(1, 2, 20, 5))


def test_full_size_properties():
    """BASELINE config 3: b = 1024 x L = 100 000 x q = 15 fwd-bwd posteriors.
    Size-independent properties on the whole output + fp64 parity on a sample of sequences."""
    torch.manual_seed(0)
    b, L, q = 1024, 100000, 15
    A = params.intended_A15().to(DEV)[None]
    pi = torch.full((q,), 1 / q, device=DEV)
    E = torch.rand((1, b, L, q), device=DEV) * 0.9 + 0.05
    out, ll = engine.posterior(A, pi, E)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    rows = out.sum(-1)
    assert float((rows - 1).abs().max()) <= 2e-5
    assert float(out.min()) >= 0.0
    out2, ll2 = engine.posterior(A, pi, E)
    assert torch.equal(out, out2) and torch.equal(ll, ll2)          # deterministic
    _, ll3 = engine.forward(A, pi, E, want_log_alpha=False)
    assert torch.equal(ll3, ll)
    idx = [0, 1, 511, 1023]
    Es = E[0, idx].cpu().numpy()
    g64, ll64 = textbook.posterior(A[0].cpu().numpy(), pi.cpu().numpy(), Es)
    got = out[0, idx].cpu().numpy()
    assert np.abs(got - g64).max() <= 2e-5
    assert np.all(np.abs(ll[0, idx].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64))
    # a shard of the batch gives the same per-sequence results (what multi-GPU sharding relies
    # on); not bitwise, because the chunk length is chosen from the shard's size
    half, llh = engine.posterior(A, pi, E[:, 256:768].contiguous())
    assert float((llh - ll[:, 256:768]).abs().max()) <= 1e-6 * float(ll.abs().max())
    assert float((half - out[:, 256:768]).abs().max()) <= 1e-5


# ---------------------------------------------------------------- sparse-topology reduce kernel

def gene7():
    ed = params.edges_simple()
    return params.dense_A(ed, np.where(params.init_logits(ed, 1) == 0, 1e-30, params.init_logits(ed, 1)), 7).numpy()


@pytest.mark.parametrize("b,L", [(1, 1), (1, 17), (3, 100), (4, 600), (5, 1031), (2, 4099), (37, 333)])
def test_gene_topology_ragged(b, L):
    """15-state gene model (served by the topology-specialised reduce kernel), lengths and batch
    sizes that leave partial tiles, partial waves and sequences starting mid-wave."""
    rng = np.random.default_rng(b * 7919 + L)
    A = params.intended_A15().numpy()
    pi = rng.random(15).astype(np.float32) + 0.1
    pi /= pi.sum()
    E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32) / 4096
    dead = rng.random(E.shape) < 0.5
    dead[..., :6] = False
    E[dead] = 0.0
    check_all(A, pi, E, "gene15 b=%d L=%d" % (b, L))


def test_seven_state_topology():
    rng = np.random.default_rng(70)
    A = gene7()
    pi = np.full(7, 1 / 7, dtype=np.float32)
    E = (rng.random((5, 700, 7)) * 0.9 + 0.05).astype(np.float32)
    check_all(A, pi, E, "gene7")


def test_sparse_and_dense_reduce_agree():
    """The same inputs through the sparse-topology kernel and (forced) through the dense MFMA
    kernel: two implementations of the same chunk operators."""
    rng = np.random.default_rng(71)
    A = params.intended_A15().numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    E = dev((rng.random((1, 9, 5000, 15)) * 0.9 + 0.05).astype(np.float32) / 4096)
    with engine.option(engine.OPT_FORCE_DENSE, 0):
        g1, l1 = engine.posterior(dev(A)[None], dev(pi), E)
        la1, _ = engine.forward(dev(A)[None], dev(pi), E)
    with engine.option(engine.OPT_FORCE_DENSE, 1):
        g2, l2 = engine.posterior(dev(A)[None], dev(pi), E)
        la2, _ = engine.forward(dev(A)[None], dev(pi), E)
    assert float((g1 - g2).abs().max()) <= 2e-6
    assert float(((l1 - l2) / l2).abs().max()) <= 1e-7
    assert float((la1 - la2).abs().max()) <= 0.2          # fp32 ulp at |log alpha| ~ 5e4 is 4e-3
    # and both are the oracle's answer
    g64, ll64 = textbook.posterior(A, pi, E[0, :2].cpu().numpy())
    assert np.abs(g1[0, :2].cpu().numpy() - g64).max() <= 2e-5


def test_mixed_models_dispatch_per_model():
    """k = 3 models in one call: gene topology, a dense matrix, gene topology with other edge
    weights — the kernel serving each model is chosen on the device from A's support.  (The same
    with an edge deleted, which leaves a state reachable only through the eps clamp:
    tests/test_exact_gpu.py::test_deleted_edge_leaves_a_state_without_incoming_edges.)"""
    rng = np.random.default_rng(72)
    q, b, L = 15, 6, 900
    A0 = params.intended_A15().numpy()
    A1, _ = rand_model(rng, q)
    A2 = params.intended_A15(50, 300, 900).numpy()
    assert ((A2 != 0) == (A0 != 0)).all()
    A = np.stack([A0, A1, A2])
    pi = np.stack([rand_model(rng, q)[1] for _ in range(3)])
    E = (rng.random((3, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    out, ll = engine.posterior(dev(A), dev(pi), dev(E))
    out, ll = out.cpu().numpy(), ll.cpu().numpy()
    for m in range(3):
        g64, ll64 = textbook.posterior(A[m], pi[m], E[m])
        assert np.abs(out[m] - g64).max() <= 2e-5, m
        assert np.all(np.abs(ll[m] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), m


def test_matrix_outside_the_topology_uses_dense_kernel():
    """A 15-state matrix with one extra edge (Ir -> E0) must not be treated as the gene topology."""
    rng = np.random.default_rng(73)
    A = params.intended_A15().numpy().copy()
    A[0, 4] = 0.01
    A[0] /= A[0].sum()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((3, 500, 15)) * 0.9 + 0.05).astype(np.float32)
    check_all(A, pi, E, "extra edge")


def test_group_pipeline_is_invisible():
    """Large batches are processed in groups on two internal streams (reduce of group g+1 under
    forward/backward of group g).  Results must not depend on the grouping, must be ordered
    after the caller's stream, and the caller's stream must wait for them."""
    torch.manual_seed(5)
    b, L, q = 256, 70000, 15                      # b*L >= 2^24: large enough to be grouped
    A = params.intended_A15().to(DEV)[None]
    pi = torch.full((q,), 1 / q, device=DEV)
    E = torch.rand((1, b, L, q), device=DEV) * 0.9 + 0.05
    outs = []
    for groups in (1, 2, 4):
        engine.release_workspaces()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())    # E is produced on the default stream
        with engine.option(engine.OPT_GROUPS, groups), torch.cuda.stream(s):
            E2 = E * 1.0                          # produced on s just before the call
            out, ll = engine.posterior(A, pi, E2)
            chk = out.sum(-1)                     # consumed on s right after the call
        s.synchronize()
        assert float((chk - 1).abs().max()) <= 2e-5
        outs.append((out, ll))
    for out, ll in outs[1:]:
        assert torch.equal(out, outs[0][0]) and torch.equal(ll, outs[0][1])
    g64, ll64 = textbook.posterior(A[0].cpu().numpy(), pi.cpu().numpy(), E[0, [0, 255]].cpu().numpy())
    assert np.abs(outs[2][0][0, [0, 255]].cpu().numpy() - g64).max() <= 2e-5
    assert np.all(np.abs(outs[2][1][0, [0, 255]].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64))


# ---------------------------------------------------------------- large-q (serial GEMM) path

@pytest.mark.parametrize("q,b,L", [(17, 3, 40), (29, 5, 120), (64, 4, 50), (100, 70, 33), (257, 9, 25)])
def test_large_q_path_all_outputs(q, b, L):
    """q > 16: serial in time, one f32-MFMA GEMM per position + exact cell semantics per row."""
    rng = np.random.default_rng(q)
    A, pi = rand_model(rng, q, dense=False)
    E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[rng.random(E.shape) < 0.1] = 0.0
    check_all(A, pi, E, "largeq q=%d" % q)


@pytest.mark.parametrize("L", [1, 2, 3, 4, 7])
def test_large_q_short_sequences(L):
    """The posterior's two recursions run from both ends and pass each other in the middle: the lengths
    where one of their four stages is empty, a single step or the initial step."""
    rng = np.random.default_rng(L)
    A, pi = rand_model(rng, 100, dense=False)
    E = (rng.random((5, L, 100)) * 0.9 + 0.05).astype(np.float32)
    E[rng.random(E.shape) < 0.1] = 0.0
    check_all(A, pi, E, "largeq L=%d" % L)
    lgl, _ = run_post(A, pi, E[None], engine.POST_LOG_NO_LL)
    g64, ll64 = textbook.posterior(A, pi, E)
    assert np.abs(np.exp(lgl[0] - ll64[:, None, None]) - g64).max() <= 2e-5 + 2.4e-7 * np.abs(ll64).max()


def test_large_q_two_streams_are_invisible():
    """lq_posterior forks onto an internal stream and joins again: a caller's side stream sees an ordinary
    in-order call (input produced just before it, output consumed right after it), results are bitwise
    reproducible, and the call can be captured into a HIP graph and replayed on new data."""
    rng = np.random.default_rng(8)
    q, b, L = 130, 96, 41
    A, pi = rand_model(rng, q, dense=False)
    A, pi = dev(A)[None], dev(pi)
    E = torch.rand((1, b, L, q), device=DEV) * 0.9 + 0.05
    ref, llref = engine.posterior(A, pi, E)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        E2 = E * 1.0                                  # produced on s just before the call
        out, ll = engine.posterior(A, pi, E2)
        chk = out.sum(-1)                             # consumed on s right after the call
    s.synchronize()
    assert torch.equal(out, ref) and torch.equal(ll, llref)
    assert float((chk - 1).abs().max()) <= 2e-5
    g64, ll64 = textbook.posterior(A[0].cpu().numpy(), pi.cpu().numpy(), E[0, :4].cpu().numpy())
    assert np.abs(ref[0, :4].cpu().numpy() - g64).max() <= 2e-5
    # graph capture and replay on new data in the same buffers
    outg = torch.empty_like(E)
    with torch.cuda.stream(s):
        engine.posterior(A, pi, E, out=outg)          # warm-up on s: workspace allocated outside the capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            _, llg = engine.posterior(A, pi, E, out=outg)
    E3 = torch.rand((1, b, L, q), device=DEV) * 0.9 + 0.05
    want, llwant = engine.posterior(A, pi, E3)
    torch.cuda.synchronize()
    E.copy_(E3)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(outg, want) and torch.equal(llg, llwant)


def test_profile_hmm_size_config5():
    """BASELINE config 5 shape per GPU, scaled in L: q = 2*512+3 = 1027 states, dense A with a
    profile-like band, b = 64, forward log-likelihood + posteriors vs the fp64 oracle."""
    rng = np.random.default_rng(1027)
    q, b, L = 1027, 64, 48
    A = rng.random((q, q)).astype(np.float32) ** 8
    A *= (np.abs(np.subtract.outer(np.arange(q), np.arange(q))) < 40) + 1e-4
    A /= A.sum(-1, keepdims=True)
    pi = rng.random(q).astype(np.float32)
    pi /= pi.sum()
    E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
    from oracle import build as obuild
    g64, ll64 = obuild.posterior(A, pi, E)
    out, ll = engine.posterior(dev(A)[None], dev(pi), dev(E)[None])
    assert np.abs(out.cpu().numpy()[0] - g64).max() <= 2e-5
    assert np.all(np.abs(ll.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    _, ll2 = engine.forward(dev(A)[None], dev(pi), dev(E)[None], want_log_alpha=False)
    assert np.all(np.abs(ll2.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    # two models in one call
    A2 = np.stack([A, A.T / A.T.sum(-1, keepdims=True)])
    pi2 = np.stack([pi, pi[::-1].copy()])
    E2 = np.stack([E[:8], E[8:16]])
    out2, ll3 = engine.posterior(dev(A2), dev(pi2), dev(E2))
    for m in range(2):
        g, l = obuild.posterior(A2[m], pi2[m], E2[m])
        assert np.abs(out2[m].cpu().numpy() - g).max() <= 2e-5 and np.abs(ll3[m].cpu().numpy() - l).max() <= 1e-3


def test_engine_calls_are_graph_capturable():
    """No hidden synchronisation, allocation or host-side state in the launch path: posterior,
    forward and Viterbi are captured into a HIP graph and replayed on new data in the same buffers."""
    rng = np.random.default_rng(21)
    A = params.intended_A15().to(DEV)[None]
    pi = torch.full((15,), 1 / 15, device=DEV)
    E = torch.rand((1, 8, 3000, 15), device=DEV) * 0.9 + 0.05
    out = torch.empty_like(E)
    logA, logpi = torch.log(A), torch.log(pi)[None]
    # warm-up allocates the cached workspaces outside the capture
    engine.posterior(A, pi, E, out=out)
    engine.forward(A, pi, E, want_log_alpha=False)
    engine.viterbi(logA, logpi, torch.log(E))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            _, ll = engine.posterior(A, pi, E, out=out)
            _, ll2 = engine.forward(A, pi, E, want_log_alpha=False)
            path, score = engine.viterbi(logA, logpi, torch.log(E))
    E2 = torch.rand((1, 8, 3000, 15), device=DEV) * 0.9 + 0.05
    E.copy_(E2)
    g.replay()
    torch.cuda.synchronize()
    g64, ll64 = textbook.posterior(A[0].cpu().numpy(), pi.cpu().numpy(), E2[0].cpu().numpy())
    assert np.abs(out[0].cpu().numpy() - g64).max() <= 2e-5
    assert np.all(np.abs(ll[0].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    assert torch.equal(ll, ll2)
    from oracle import build as obuild
    wp, ws = obuild.viterbi(logA[0].cpu().numpy(), logpi[0].cpu().numpy(), torch.log(E2)[0].cpu().numpy())
    assert np.array_equal(path[0].cpu().numpy(), wp) and np.array_equal(score[0].cpu().numpy(), ws)


def test_two_level_chunk_scan_matches_single_level():
    """From 32 chunks per sequence on, the chunk-level scan composes groups of ~sqrt(C) operators in
    parallel (k_scan_compose -> k_scan over groups -> k_scan_inner).  The hops are linear, so both
    orders agree to rounding; both are held to the fp64 oracle."""
    rng = np.random.default_rng(77)
    for (b, L, q, dense) in ((1, 5000, 15, False), (2, 3001, 7, True), (3, 2500, 16, True)):
        A, pi = rand_model(rng, q, dense=dense)
        if q == 15 and not dense:
            A = params.intended_A15().numpy()
            pi = np.full(15, 1 / 15, dtype=np.float32)
        E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
        assert engine.chunk_len(1, b, L, q) * 32 <= L                      # at least 32 chunks
        outs = {}
        for flag in ("1", "0"):
            with engine.option(engine.OPT_SCAN2, int(flag)):
                gam, ll = run_post(A, pi, E[None])
                la, _ = engine.forward(dev(A)[None], dev(pi), dev(E[None]))
                lb = engine.backward(dev(A)[None], dev(E[None]))
            outs[flag] = (gam, ll, la.cpu().numpy(), lb.cpu().numpy())
        g64, ll64 = textbook.posterior(A, pi, E)
        for flag in ("1", "0"):
            assert np.abs(outs[flag][0][0] - g64).max() <= 2e-5, flag
            assert np.all(np.abs(outs[flag][1][0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), flag
        assert np.abs(outs["1"][0] - outs["0"][0]).max() <= 2e-6
        assert np.abs(outs["1"][1] - outs["0"][1]).max() <= 1e-7 * np.abs(ll64).max() + 1e-5
        assert np.abs(outs["1"][2] - outs["0"][2]).max() <= 2e-3 and np.abs(outs["1"][3] - outs["0"][3]).max() <= 2e-3


def test_mid_size_two_models_in_one_call():
    """17..64 states with k = 2 (the meet-in-the-middle posterior kernel indexes models per sequence)."""
    rng = np.random.default_rng(291)
    q, b, L = 29, 3, 57
    Ms = [rand_model(rng, q, dense=False) for _ in range(2)]
    A = np.stack([m[0] for m in Ms]); pi = np.stack([m[1] for m in Ms])
    E = (rng.random((2, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    for mode in (engine.POST_PROB, engine.POST_LOG, engine.POST_LOG_NO_LL):
        out, ll = engine.posterior(dev(A), dev(pi), dev(E), mode=mode)
        for m in range(2):
            g64, ll64 = textbook.posterior(A[m], pi[m], E[m])
            got = out[m].cpu().numpy()
            if mode == engine.POST_LOG_NO_LL:
                got = got - ll[m].cpu().numpy()[:, None, None]
            if mode != engine.POST_PROB:
                got = np.exp(got)
            assert np.abs(got - g64).max() <= 3e-5, (mode, m)
            assert np.all(np.abs(ll[m].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
