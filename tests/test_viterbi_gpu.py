"""hmm_viterbi against the Q16 fixed-point oracle: state paths and scores must be BIT-EXACT
(the reference has no Viterbi — parity unpinned; oracle/viterbi.py defines the semantics)."""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from oracle import build as obuild
from oracle import params, viterbi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=DEV)


def run(logA, logpi, logE):
    path, score = engine.viterbi(dev(logA)[None], dev(logpi)[None], dev(logE)[None])
    torch.cuda.synchronize()
    return path.cpu().numpy()[0], score.cpu().numpy()[0]


def gene_logs(rng, b, L, zero_frac=0.0, scale=1.0 / 4096):
    A = params.intended_A15().numpy()
    with np.errstate(divide="ignore"):
        logA = np.log(A).astype(np.float32)                   # -inf for absent edges
    logpi = np.log(np.full(15, 1 / 15, dtype=np.float32))
    E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32) * np.float32(scale)
    if zero_frac:
        dead = rng.random(E.shape) < zero_frac
        dead[..., :6] = False
        E[dead] = 0.0
    logE = np.log(np.maximum(E, np.float32(1e-16))).astype(np.float32)
    return logA, logpi, logE


def check(logA, logpi, logE, tag=""):
    want_path, want_score = obuild.viterbi(logA, logpi, logE)
    got_path, got_score = run(logA, logpi, logE)
    assert np.array_equal(got_score, want_score), (tag, got_score[:3], want_score[:3])
    bad = np.argwhere(got_path != want_path)
    assert len(bad) == 0, (tag, len(bad), bad[:5].tolist())


def test_numpy_definition_small():
    rng = np.random.default_rng(0)
    logA, logpi, logE = gene_logs(rng, 3, 40)
    want_path, want_score = viterbi.viterbi(logA, logpi, logE)
    got_path, got_score = run(logA, logpi, logE)
    assert np.array_equal(got_path, want_path) and np.array_equal(got_score, want_score)


@pytest.mark.parametrize("b,L", [(1, 1), (1, 2), (2, 15), (1, 16), (3, 17), (5, 100), (4, 600), (2, 1031),
                                 (37, 333), (3, 4099), (16, 5000)])
def test_gene_model_paths_bit_exact(b, L):
    rng = np.random.default_rng(b * 1000 + L)
    check(*gene_logs(rng, b, L, zero_frac=0.5), tag="b=%d L=%d" % (b, L))


@pytest.mark.parametrize("q", [1, 2, 3, 5, 7, 8, 12, 15, 16])
def test_dense_matrices_all_state_counts(q):
    rng = np.random.default_rng(q)
    logA = np.log(rng.dirichlet(np.ones(q), size=q)).astype(np.float32)
    logpi = np.log(rng.dirichlet(np.ones(q))).astype(np.float32)
    logE = (-4 * rng.random((3, 300, q))).astype(np.float32)
    check(logA, logpi, logE, "dense q=%d" % q)


def test_ties_lowest_index_and_uniform_matrix():
    q, L = 6, 700
    logA = np.zeros((q, q), dtype=np.float32)             # every entry is the matrix minimum: no edges at all
    logpi = np.zeros(q, dtype=np.float32)
    logE = np.zeros((2, L, q), dtype=np.float32)
    logE[0, 300, 0] = -1.0
    logE[1, ::7, :3] = -0.5
    check(logA, logpi, logE, "ties")
    path, _ = run(logA, logpi, logE)
    assert path[0, 299] == 0 and path[0, 300] == 1 and path[0, 301] == 0


def test_coarse_scores_force_many_ties():
    """Scores that are multiples of 0.25: long exact ties between paths, across chunk boundaries."""
    rng = np.random.default_rng(3)
    q, b, L = 15, 4, 3000
    logA = (-0.25 * rng.integers(0, 8, (q, q))).astype(np.float32)
    logA[rng.random((q, q)) < 0.5] = -np.inf
    logpi = (-0.25 * rng.integers(0, 4, q)).astype(np.float32)
    logE = (-0.25 * rng.integers(0, 6, (b, L, q))).astype(np.float32)
    check(logA, logpi, logE, "coarse")


def test_multiple_models_and_clamping():
    rng = np.random.default_rng(4)
    k, b, L, q = 3, 5, 400, 7
    logA = np.log(rng.dirichlet(np.ones(q), size=(k, q))).astype(np.float32)
    logA[1][rng.random((q, q)) < 0.4] = -np.inf
    logA[2] = np.clip(logA[2] * 500, -5000, 0)            # values far below the -1024 clamp
    logpi = np.log(rng.dirichlet(np.ones(q), size=k)).astype(np.float32)
    logE = (-6 * rng.random((k, b, L, q))).astype(np.float32)
    logE[rng.random(logE.shape) < 0.05] = -np.inf
    path, score = engine.viterbi(dev(logA), dev(logpi), dev(logE))
    path, score = path.cpu().numpy(), score.cpu().numpy()
    for m in range(k):
        wp, ws = obuild.viterbi(logA[m], logpi[m], logE[m])
        assert np.array_equal(path[m], wp) and np.array_equal(score[m], ws), m


def test_path_is_consistent_with_posterior_decoding_on_easy_data():
    """Sanity link to the forward-backward engine: with strongly informative emissions the Viterbi
    path (best joint path, must follow the model's edges) and the posterior argmax (per-position
    marginals) agree on most positions; the random labels below ignore the topology, so they
    cannot agree everywhere."""
    rng = np.random.default_rng(5)
    A = params.intended_A15().numpy()
    q, b, L = 15, 2, 2000
    truth = rng.integers(0, 6, (b, L))
    E = np.full((b, L, q), 1e-3, dtype=np.float32)
    np.put_along_axis(E, truth[..., None], 1.0, axis=-1)
    pi = np.full(q, 1 / q, dtype=np.float32)
    with np.errstate(divide="ignore"):
        path, _ = run(np.log(A).astype(np.float32), np.log(pi), np.log(E))
    post, _ = engine.posterior(dev(A)[None], dev(pi), dev(E)[None])
    agree = (post[0].argmax(-1).cpu().numpy() == path).mean()
    assert agree > 0.8


def test_full_size_config4():
    """BASELINE config 4: b = 1024 x L = 100 000 x q = 15.  Determinism on the whole batch and
    bit-exactness against the CPU oracle on a sample of sequences."""
    torch.manual_seed(0)
    b, L, q = 1024, 100000, 15
    A = params.intended_A15().to(DEV)
    logA = torch.log(A)[None]
    logpi = torch.log(torch.full((1, q), 1 / q, device=DEV))
    logE = torch.log(torch.rand((1, b, L, q), device=DEV) * 0.9 + 0.05)
    p1, s1 = engine.viterbi(logA, logpi, logE)
    p2, s2 = engine.viterbi(logA, logpi, logE)
    torch.cuda.synchronize()
    assert torch.equal(p1, p2) and torch.equal(s1, s2)
    assert int(p1.min()) >= 0 and int(p1.max()) < q
    idx = [0, 1, 511, 1023]
    wp, ws = obuild.viterbi(logA[0].cpu().numpy(), logpi[0].cpu().numpy(), logE[0, idx].cpu().numpy())
    assert np.array_equal(p1[0, idx].cpu().numpy(), wp)
    assert np.array_equal(s1[0, idx].cpu().numpy(), ws)
    # every transition on the returned paths is an existing edge of the model
    src, dst = p1[0, :, :-1].reshape(-1)[:5_000_000].long(), p1[0, :, 1:].reshape(-1)[:5_000_000].long()
    flat = (p1[0, :8, :-1].long() * q + p1[0, :8, 1:].long()).reshape(-1)
    assert bool((A.reshape(-1)[flat] > 0).all())


def test_specialised_and_generic_reduce_agree():
    """Gene-topology models are served by the register-resident max-plus reduce, anything else
    (and everything when forced) by the generic edge-list kernel: identical integers either way."""
    rng = np.random.default_rng(9)
    logA, logpi, logE = gene_logs(rng, 6, 2500, zero_frac=0.4)
    with engine.option(engine.OPT_FORCE_DENSE, 0):
        p1, s1 = run(logA, logpi, logE)
    with engine.option(engine.OPT_FORCE_DENSE, 1):
        p2, s2 = run(logA, logpi, logE)
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2)
    wp, ws = obuild.viterbi(logA, logpi, logE)
    assert np.array_equal(p1, wp) and np.array_equal(s1, ws)
    # a gene-like matrix with one extra finite edge must fall back to the generic kernel and stay exact
    logA2 = logA.copy()
    logA2[0, 4] = -3.0
    check(logA2, logpi, logE, "extra edge")
    # 7-state topology
    A7 = params.dense_A(params.edges_simple(), np.where(params.init_logits(params.edges_simple(), 1) == 0, 1e-30,
                                                          params.init_logits(params.edges_simple(), 1)), 7).numpy()
    with np.errstate(divide="ignore"):
        check(np.log(A7).astype(np.float32), np.log(np.full(7, 1 / 7, dtype=np.float32)),
              (-5 * rng.random((4, 900, 7))).astype(np.float32), "gene7")


@pytest.mark.parametrize("frac_dead", [1.0, 0.97, 0.5])
def test_gene_model_extreme_values_stay_exact(frac_dead):
    """Worst case for the 32-bit relative scores of the register-resident kernels: emissions at the
    -1024 clamp (log 0) almost everywhere, so every step costs up to 2^26 for the emission plus 2^26
    for a transition through the matrix minimum, for hundreds of steps — frames that move only every
    few steps must not overflow, and the result must stay bit-exact."""
    rng = np.random.default_rng(int(frac_dead * 100))
    logA, logpi, logE = gene_logs(rng, 5, 1300)
    dead = rng.random(logE.shape) < frac_dead
    logE = np.where(dead, -np.inf, logE).astype(np.float32)
    check(logA, logpi, logE, "extreme %.2f" % frac_dead)


@pytest.mark.parametrize("q,b,L", [(17, 3, 70), (29, 4, 333), (48, 2, 129), (64, 3, 200), (33, 1, 1)])
def test_mid_size_models_one_wave_per_sequence(q, b, L):
    """17..64 states: lane = state, all q candidates per step; bit-exact paths and scores, incl.
    absent edges (-inf), ties and a multi-model call."""
    rng = np.random.default_rng(q * 1000 + L)
    logA = np.log(rng.dirichlet(np.ones(q), size=q)).astype(np.float32)
    logA[rng.random((q, q)) < 0.5] = -np.inf
    logA[np.arange(q), np.arange(q)] = np.float32(np.log(0.5))
    logpi = np.log(rng.dirichlet(np.ones(q))).astype(np.float32)
    logE = (-6 * rng.random((b, L, q))).astype(np.float32)
    logE[rng.random(logE.shape) < 0.05] = -np.inf
    check(logA, logpi, logE, "midq q=%d" % q)
    # coarse scores: many exact ties, lowest index must win
    check(np.round(logA), np.round(logpi), np.round(logE), "midq ties q=%d" % q)
    if b >= 2:
        la2 = np.stack([logA, logA.T.copy()]); lp2 = np.stack([logpi, logpi[::-1].copy()])
        le2 = np.stack([logE, logE[::-1].copy()])
        path, score = engine.viterbi(dev(la2), dev(lp2), dev(le2))
        for m in range(2):
            wp, ws = obuild.viterbi(la2[m], lp2[m], le2[m])
            assert np.array_equal(path[m].cpu().numpy(), wp) and np.array_equal(score[m].cpu().numpy(), ws)


def gene_k_logs(k):
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    tr = GenePredMultiHMMTransitioner(k=k, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    with torch.no_grad():
        A = tr.make_A()[0].numpy().copy()
        pi = tr.make_initial_distribution().reshape(-1).numpy().copy()
    with np.errstate(divide="ignore"):
        return np.log(A).astype(np.float32), np.log(pi).astype(np.float32)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_sparse_mid_size_models_visit_their_predecessors_only(k):
    """The gene models of 29 / 43 / 57 states (at most 3 / 4 / 5 edges into a state): k_mq_viterbi_sparse gathers a
    state's explicit predecessors and covers every absent edge with one candidate.  Bit-exact against the oracle's
    all-candidates loop, with dead emissions, coarse scores (ties between explicit and absent edges) and the dense
    loop forced on the same input."""
    rng = np.random.default_rng(300 + k)
    logA, logpi = gene_k_logs(k)
    q = 1 + 14 * k
    for b, L in ((1, 1), (2, 2), (3, 257), (2, 3001)):
        logE = (-8 * rng.random((b, L, q))).astype(np.float32)
        logE[rng.random(logE.shape) < 0.2] = -np.inf
        check(logA, logpi, logE, "sparse k=%d b=%d L=%d" % (k, b, L))
        # coarse: every score a multiple of 4 (incl. -1024 for the absent edges): ties everywhere
        cA = np.where(np.isfinite(logA), 4 * np.round(logA / 4), -np.inf).astype(np.float32)
        cE = np.where(np.isfinite(logE), 4 * np.round(logE / 4), -np.inf).astype(np.float32)
        check(cA, 4 * np.round(logpi / 4), cE, "sparse ties k=%d" % k)
        # everything impossible for a stretch: only absent-edge candidates survive the clamp at -1024
        dE = logE.copy()
        dE[:, L // 3:L // 3 + 7] = -np.inf
        check(logA, logpi, dE, "sparse dead stretch k=%d" % k)
    got = run(logA, logpi, logE)
    with engine.option(engine.OPT_FORCE_DENSE, 1):
        dense = run(logA, logpi, logE)
    assert np.array_equal(got[0], dense[0]) and np.array_equal(got[1], dense[1])


@pytest.mark.parametrize("q,deg", [(17, 1), (32, 4), (40, 5), (64, 8), (50, 9)])
def test_sparse_mid_size_random_topologies(q, deg):
    """In-degree up to 4 / up to 8 / above (dense loop); some states without any explicit predecessor; two models of
    different kinds in one call."""
    rng = np.random.default_rng(q * 10 + deg)
    logA = np.full((q, q), -np.inf, dtype=np.float32)
    for jj in range(q):
        n = 0 if jj % 7 == 3 else int(rng.integers(1, deg + 1))
        if jj == 0:
            n = deg
        for i in rng.choice(q, size=n, replace=False):
            logA[i, jj] = np.float32(-3 * rng.random())
    logpi = np.log(rng.dirichlet(np.ones(q))).astype(np.float32)
    b, L = 3, 500
    logE = (-6 * rng.random((b, L, q))).astype(np.float32)
    logE[rng.random(logE.shape) < 0.05] = -np.inf
    check(logA, logpi, logE, "random sparse q=%d deg=%d" % (q, deg))
    check(np.round(logA), np.round(logpi), np.round(logE), "random sparse ties q=%d" % q)
    dense = np.log(rng.dirichlet(np.ones(q), size=q)).astype(np.float32)
    la2 = np.stack([logA, dense]); lp2 = np.stack([logpi, logpi[::-1].copy()]); le2 = np.stack([logE, logE[::-1].copy()])
    path, score = engine.viterbi(dev(la2), dev(lp2), dev(le2))
    for m in range(2):
        wp, ws = obuild.viterbi(la2[m], lp2[m], le2[m])
        assert np.array_equal(path[m].cpu().numpy(), wp) and np.array_equal(score[m].cpu().numpy(), ws)


def test_two_level_scans_match_single_level_and_the_oracle():
    """From 32 chunks per sequence on, both chunk-level scans of the Viterbi pipeline run in two levels
    (k_vit_scan_compose -> k_vit_scan_fwd over groups -> k_vit_scan_inner; the same for the backpointer
    maps).  Integer max-plus composes exactly: identical integers either way, and the oracle's."""
    rng = np.random.default_rng(23)
    for (b, L, chunk, zero_frac) in ((1, 40000, 0, 0.3), (3, 20011, 16, 0.5), (2, 5000, 48, 0.0), (5, 1100, 16, 0.6)):
        logA, logpi, logE = gene_logs(rng, b, L, zero_frac=zero_frac)
        with engine.option(engine.OPT_CHUNK, chunk):
            assert L >= 32 * engine.chunk_len(1, b, L, 15)
            with engine.option(engine.OPT_SCAN2, 1):
                p2, s2 = run(logA, logpi, logE)
            with engine.option(engine.OPT_SCAN2, 0):
                p1, s1 = run(logA, logpi, logE)
        assert np.array_equal(p1, p2) and np.array_equal(s1, s2)
        wp, ws = obuild.viterbi(logA, logpi, logE)
        assert np.array_equal(p2, wp) and np.array_equal(s2, ws)
    # a dense 7-state model (generic kernels) with ties everywhere
    q = 7
    logA = np.log(rng.dirichlet(np.ones(q), size=q)).astype(np.float32)
    logpi = np.log(rng.dirichlet(np.ones(q))).astype(np.float32)
    logE = (-0.25 * rng.integers(0, 6, (2, 9000, q))).astype(np.float32)
    with engine.option(engine.OPT_CHUNK, 32):
        check(logA, logpi, logE, "two-level dense")


def test_batch_groups_do_not_change_anything():
    """hmm_viterbi pipelines groups of sequences over two streams (HMM_OPT_VGROUPS): identical paths and scores
    for every grouping, on the caller's side stream, input produced just before and output consumed right after."""
    rng = np.random.default_rng(5)
    b, L, q = 70, 3000, 15
    logA = torch.log(params.intended_A15().to(DEV))[None]
    logpi = torch.log(torch.full((1, q), 1 / q, device=DEV))
    E = torch.as_tensor(rng.random((1, b, L, q)) * 0.9 + 0.05, dtype=torch.float32, device=DEV)
    want = None
    s = torch.cuda.Stream()
    for n in (1, 2, 3, 4):
        with engine.option(engine.OPT_VGROUPS, n):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                logE = torch.log(E)                       # produced on s just before the call
                path, score = engine.viterbi(logA, logpi, logE)
                chk = path.sum()                          # consumed on s right after the call
            s.synchronize()
        if want is None:
            want = (path.clone(), score.clone(), int(chk))
            from oracle import build as obuild
            wp, ws = obuild.viterbi(logA[0].cpu().numpy(), logpi[0].cpu().numpy(), torch.log(E)[0, :5].cpu().numpy())
            assert np.array_equal(path[0, :5].cpu().numpy(), wp) and np.array_equal(score[0, :5].cpu().numpy(), ws)
        assert torch.equal(path, want[0]) and torch.equal(score, want[1]) and int(chk) == want[2]
    # two models in one call (the groups of one model are joined before the next reuses the workspace)
    logA2 = torch.cat([logA, torch.log(torch.softmax(torch.randn((1, q, q), device=DEV), -1))])
    logpi2 = torch.cat([logpi, logpi])
    logE2 = torch.log(torch.cat([E, E.flip(1)]))
    ref = engine.viterbi(logA2, logpi2, logE2)
    with engine.option(engine.OPT_VGROUPS, 3):
        got = engine.viterbi(logA2, logpi2, logE2)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert torch.equal(ref[0][0], want[0][0]) and torch.equal(ref[1][0], want[1][0])
