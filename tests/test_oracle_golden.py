"""The CPU oracle against fixtures captured from the imported reference
(tests/golden/make_golden.py) and against the TF outputs recorded in the
reference's tests/test_tf.ipynb.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import params, ref_cell, textbook


def _p(g):
    return ref_cell.HmmParams(g["A"], g["pi"])


def _E(g):
    return torch.as_tensor(g["E"]).unsqueeze(0)


def test_known_answer_toy(golden):
    g = golden("kat")
    p = _p(g)
    fo, st = ref_cell.forward_outputs(p, _E(g))
    assert np.array_equal(fo.numpy(), g["fwd"])
    assert abs(float(st[1]) - (-3.4076104164)) < 1e-6
    gam, ll = textbook.posterior(g["A"], g["pi"], g["E"])
    assert abs(ll[0] - (-3.407610614)) < 1e-6
    np.testing.assert_allclose(gam[0, 0], [0.66464191, 0.10683619, 0.22852191], atol=2e-7)
    np.testing.assert_allclose(gam[0, 3], [0.35321482, 0.50980066, 0.13698451], atol=2e-7)


@pytest.mark.parametrize("name", ["cell_q3", "cell_q7", "cell_q15", "cell_q15z"])
def test_cell_steps_bit_exact(golden, name):
    """Op-for-op restatement == reference HmmCell.forward, both directions, every step."""
    g = golden(name)
    p = _p(g)
    fo, st = ref_cell.forward_outputs(p, _E(g))
    bo, _ = ref_cell.backward_outputs(p, _E(g))
    assert np.array_equal(fo.numpy(), g["fwd"])
    assert np.array_equal(bo.numpy(), g["bwd"])
    assert np.array_equal(st[1].reshape(-1).numpy(), g["loglik"])


@pytest.mark.parametrize("name", ["cell_q3", "cell_q7", "cell_q15", "cell_q15z"])
def test_fp64_textbook_agrees_with_reference_cell(golden, name):
    g = golden(name)
    la, ll = textbook.log_alpha(g["A"], g["pi"], g["E"])
    lb = textbook.log_beta(g["A"], g["E"])
    ref_la = g["fwd"][..., :-1] + g["fwd"][..., -1:]
    ref_lb = g["bwd"][..., :-1] + g["bwd"][..., -1:]
    m = la > -30
    assert np.abs(la - ref_la)[m].max() < 2e-4
    assert np.abs(lb - ref_lb)[lb > -30].max() < 2e-4
    assert np.abs(ll - g["loglik"]).max() < 2e-4
    # posterior from the reference's scaled variables vs fp64
    gam, _ = textbook.posterior(g["A"], g["pi"], g["E"])
    ref_g, _ = ref_cell.posterior_scaled(_p(g), _E(g))
    assert np.abs(gam - ref_g[0].numpy()).max() < 5e-6


@pytest.mark.parametrize("q", ["q3", "q15"])
@pytest.mark.parametrize("pf", [2, 4, 8])
def test_chunk_parallel_mode(golden, q, pf):
    g = golden("chunk_%s_pf%d" % (q, pf))
    p = _p(g)
    E = _E(g)
    b, L, _ = g["E"].shape
    st = ref_cell.initial_state(p, b * pf, parallel_factor=pf)
    assert np.array_equal(st[0].numpy(), g["init_f"])
    rows = E.reshape(b * pf, L // pf, -1)
    st = ref_cell.initial_state(p, b * pf, reverse=True, parallel_factor=pf, chunk_emissions=rows)
    assert np.array_equal(st[0].numpy(), g["init_b"])
    fwd, bwd = ref_cell.chunked_outputs(p, E, pf)
    assert np.array_equal(fwd.numpy(), g["fwd"])
    assert np.array_equal(bwd.numpy(), g["bwd"])
    post, ll, la, lb = ref_cell.posterior_log_probs_chunked(p, E, pf)
    np.testing.assert_allclose(la[0].numpy(), g["log_alpha"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(lb[0].numpy(), g["log_beta"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ll[0].numpy(), g["loglik"], rtol=0, atol=2e-5)
    # and the chunked result equals the serial fp64 one
    gam, ll64 = textbook.posterior(g["A"], g["pi"], g["E"])
    assert np.abs(np.exp(post[0].numpy()) - gam).max() < 3e-4
    assert np.abs(ll64 - g["loglik"]).max() < 1e-4


def test_transitioner_matrices(golden):
    g = golden("transitioner")
    for tag, q in (("7", 7), ("15", 15)):
        A = params.dense_A(g["edges" + tag], g["logits" + tag], q)
        assert np.array_equal(A.numpy(), g["A" + tag])
        np.testing.assert_allclose(A.sum(-1).numpy(), 1.0, atol=1e-6)
    # the as-shipped matrices (defect D1: zero logits dropped) via the compatibility flag
    ed7, ed15 = params.edges_simple(), params.edges_multi(1)
    assert np.array_equal(ed7, g["edges7"]) and np.array_equal(ed15, g["edges15"])
    assert np.array_equal(params.edges_15(), g["edges15_single"])
    assert np.array_equal(params.edges_multi(2), g["edges29"])
    l7 = params.init_logits(ed7, 1)
    A = params.dense_A(ed7, l7, 7, zero_logit_is_absent=True)
    assert np.array_equal(A.numpy(), g["A7_as_shipped"])
    l15 = params.init_logits(ed15, 1, 200, 4500, 10000)
    A = params.dense_A(ed15, l15, 15, zero_logit_is_absent=True)
    assert np.array_equal(A.numpy(), g["A15_as_shipped"])
    assert int((g["A15_as_shipped"] != 0).sum()) == 9
    A = params.dense_A(params.edges_15(), g["logits15_single"], 15, zero_logit_is_absent=True)
    assert np.array_equal(A.numpy(), g["A15_single_as_shipped"])
    A = params.dense_A(params.edges_multi(2), g["logits29"], 29, zero_logit_is_absent=True)
    assert np.array_equal(A.numpy(), g["A29_as_shipped"])
    np.testing.assert_allclose(params.init_logits(ed15, 1, 200, 4500, 10000),
                               np.where(g["logits15"] == np.float32(1e-30), 0, g["logits15"]), rtol=1e-6)
    # intended matrix quoted in SURVEY.md section 8(c)
    A = g["A15"]
    assert abs(A[0, 0] - .99989998) < 1e-7 and abs(A[0, 7] - 1e-4) < 1e-8
    assert abs(A[4, 5] - .995) < 1e-6 and abs(A[5, 9] - .0025) < 1e-6 and A[14, 0] == 1
    np.testing.assert_allclose(params.start_distribution(np.zeros(15)).numpy(), g["pi15"])


def test_emitters(golden):
    g = golden("emitter")
    x = torch.as_tensor(g["x"])
    tab = params.codon_table(**params.DEFAULT_CODONS)
    assert np.array_equal(tab.numpy(), g["codon_probs"])
    E = params.gene_emissions(x, g["kernel"], tab)
    assert np.array_equal(E.numpy(), g["E"])
    E = params.gene_emissions(x, g["kernel"], tab, d5_compat=True)
    assert np.array_equal(E.numpy(), g["E_as_shipped"])
    E = params.gene_emissions(x, g["kernel"], tab, training=True)
    assert np.array_equal(E.numpy(), g["E_training"])
    E = params.gene_emissions(x, g["kernel_c2"], tab, copies=2, share_intron=False)
    assert np.array_equal(E.numpy(), g["E_c2"])
    E = params.class_emissions(x[..., :15], g["kernel_simple"])
    assert np.array_equal(E.numpy(), g["E_simple"])


def test_kmers_against_tf_notebook_outputs(golden):
    g = golden("kmer")
    x = torch.as_tensor(g["tf_input"])
    keep = x.clone()
    assert np.array_equal(params.make_k_mers(x, 3, True).numpy(), g["tf_k_mers_left"])
    assert np.array_equal(params.make_k_mers(x, 3, False).numpy(), g["tf_k_mers_right"])
    assert torch.equal(x, keep)                      # no in-place mutation (defect D5)
    assert np.array_equal(params.encode_kmer_string("ACGN", True).numpy(), g["tf_encoded_kmer_left"])
    assert np.array_equal(params.encode_kmer_string("ACGN", False).numpy(), g["tf_encoded_kmer_right"])
    nuc = torch.as_tensor(g["nuc"])
    assert np.array_equal(params.make_k_mers(nuc, 3, True).numpy(), g["left"])
    assert np.array_equal(params.make_k_mers(nuc, 3, False).numpy(), g["right"])


def test_brute_force_property():
    rng = np.random.default_rng(0)
    A = rng.random((3, 3)); A /= A.sum(-1, keepdims=True)
    pi = rng.random(3); pi /= pi.sum()
    E = rng.random((2, 6, 3))
    np.testing.assert_allclose(textbook.loglik(A, pi, E, clamp=False),
                               textbook.brute_force_loglik(A, pi, E), rtol=1e-12)


def test_gradient_oracles_agree():
    """fp64 Baum-Welch gradients == autograd through the restated reference loop == finite differences."""
    from oracle import ref_cell, textbook
    rng = np.random.default_rng(0)
    q, b, L = 5, 3, 40
    A = rng.random((q, q)) + 0.05
    A /= A.sum(1, keepdims=True)
    pi = rng.random(q) + 0.1
    pi /= pi.sum()
    E = rng.random((b, L, q)) * 0.9 + 0.05
    w = rng.random(b) + 0.5
    dA, dpi, dE = textbook.loglik_grad(A, pi, E, w)
    gA, gpi, gE, _ = ref_cell.loglik_grad(A[None], pi[None], E[None], w[None])
    assert np.abs(dA - gA[0].numpy()).max() <= 1e-5 * np.abs(dA).max()
    assert np.abs(dpi - gpi[0].numpy()).max() <= 1e-5 * np.abs(dpi).max()
    assert np.abs(dE - gE[0].numpy()).max() <= 1e-5 * np.abs(dE).max()

    def f(A_, pi_, E_):
        return (textbook.loglik(A_, pi_, E_) * w).sum()

    h = 1e-6
    A2 = A.copy(); A2[1, 2] += h
    assert abs((f(A2, pi, E) - f(A, pi, E)) / h - dA[1, 2]) <= 1e-4 * abs(dA[1, 2])
    E2 = E.copy(); E2[1, 7, 2] += h
    assert abs((f(A, pi, E2) - f(A, pi, E)) / h - dE[1, 7, 2]) <= 1e-4 * abs(dE[1, 7, 2])


@pytest.mark.parametrize("name", ["grad_q5", "grad_q15"])
def test_gradient_oracles_pinned_by_reference_autograd(golden, name):
    """tests/golden/grad_*.npz = autograd through the IMPORTED reference's cell loop
    (tests/golden/make_golden_grad.py).  The restated loop reproduces it to fp32 rounding and the
    fp64 Baum-Welch oracle agrees with both."""
    from oracle import ref_cell, textbook
    g = golden(name)
    dA, dpi, dE, ll = ref_cell.loglik_grad(g["A"][None], g["pi"][None], g["E"][None], g["w"][None])
    assert np.abs(ll[0].numpy() - g["loglik"]).max() <= 1e-4
    for got, want in ((dA[0].numpy(), g["dA"]), (dpi[0].numpy(), g["dpi"]), (dE[0].numpy(), g["dE"])):
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    tA, tpi, tE = textbook.loglik_grad(g["A"], g["pi"], g["E"], g["w"])
    assert np.abs(tA - g["dA"]).max() <= 2e-4 * np.abs(g["dA"]).max()
    assert np.abs(tpi - g["dpi"]).max() <= 2e-4 * np.abs(g["dpi"]).max()
    assert np.abs(tE - g["dE"]).max() <= 2e-4 * np.abs(g["dE"]).max()


def test_posterior_gradient_oracle_is_pinned():
    """oracle/torch64.py: fp64 torch restatement used as the yardstick for hmm_posterior_grad.  Its
    posteriors equal oracle/textbook.py's, and its gradients equal fp32 autograd through the restated
    reference formula log alpha + log beta - loglik (oracle/ref_cell.py, itself bit-pinned to the
    imported reference cell by the cell_* fixtures)."""
    import torch
    from oracle import ref_cell, textbook, torch64
    rng = np.random.default_rng(0)
    q, b, L = 5, 2, 30
    A = rng.random((q, q)) + 0.05
    A /= A.sum(1, keepdims=True)
    pi = rng.random(q) + 0.1
    pi /= pi.sum()
    E = rng.random((b, L, q)) * 0.9 + 0.05
    g64, ll64 = textbook.posterior(A, pi, E)
    gam, ll = torch64.posterior(torch.tensor(A), torch.tensor(pi), torch.tensor(E))
    assert np.abs(gam.numpy() - g64).max() <= 1e-14 and np.abs(ll.numpy() - ll64).max() <= 1e-12
    G = rng.standard_normal((b, L, q))
    dA, dpi, dE, _ = torch64.posterior_grad(A, pi, E, G, log=True)
    At = torch.tensor(A, dtype=torch.float32, requires_grad=True)
    pit = torch.tensor(pi, dtype=torch.float32, requires_grad=True)
    Et = torch.tensor(E[None], dtype=torch.float32, requires_grad=True)
    lp, _ = ref_cell.posterior_log_probs(ref_cell.HmmParams(At, pit), Et)            # (1, b, L, q)
    (lp[0] * torch.tensor(G, dtype=torch.float32)).sum().backward()
    assert np.abs(dA - At.grad.numpy()).max() <= 2e-5 * np.abs(dA).max()
    assert np.abs(dE - Et.grad[0].numpy()).max() <= 2e-5 * np.abs(dE).max()
    assert np.abs(dpi - pit.grad.numpy().reshape(-1)).max() <= 2e-5 * np.abs(dpi).max()
