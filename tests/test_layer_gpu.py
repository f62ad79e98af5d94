"""MsaHmmLayer / _impl functions on a HIP device: the drop-in surface of the reference's layer,
with the gene-prediction emitter and transitioner feeding the engine.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from hmm_layer_amd.MsaHmmCell import HmmCell
from hmm_layer_amd import MsaHMMLayer as L5
from hmm_layer_amd.MsaHMMLayer import MsaHmmLayer
from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
from oracle import params, ref_cell, textbook

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CODONS = dict(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
              intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)],
              intron_end_pattern=[("AGN", .99), ("ACN", .01)])


def gene_setup(b, L, seed=0, device=None):
    device = DEV if device is None else device
    g = torch.Generator().manual_seed(seed)
    cls = torch.softmax(2 * torch.randn((1, b, L, 15), generator=g), -1)
    nuc = torch.nn.functional.one_hot(torch.randint(0, 5, (1, b, L), generator=g), 5).float()
    x = torch.cat([cls, nuc], -1)
    em = GenePredHMMEmitter(**CODONS)
    em.build((1, b, L, 15))
    with torch.no_grad():
        em.emission_kernel.copy_(torch.randn(em.emission_kernel.shape, generator=g))
    tr = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    cell = HmmCell([15], 15, em, tr).to(device)
    # oracle side: same parameters through the CPU restatement of the producers
    tab = params.codon_table(**params.DEFAULT_CODONS)
    E = params.gene_emissions(x, em.emission_kernel.detach().cpu(), tab).numpy()[0]
    A = params.intended_A15(200, 4500, 10000).numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    return cell, x.to(device), A, pi, E


def test_layer_against_oracle_gene_model():
    b, L = 3, 700
    cell, x, A, pi, E = gene_setup(b, L)
    layer = MsaHmmLayer(cell, use_prior=False, parallel_factor=7)
    layer.build(x.shape)
    g64, ll64 = textbook.posterior(A, pi, E)
    la64, _ = textbook.log_alpha(A, pi, E)
    lb64 = textbook.log_beta(A, E)
    post = layer.state_posterior_log_probs(x).detach()      # carries a graph under default autograd mode
    assert post.shape == (1, b, L, 15) and post.is_cuda
    assert np.abs(np.exp(post.cpu().numpy()[0]) - g64).max() <= 2e-5
    probs, ll = layer.state_posterior_probs(x)
    probs, ll = probs.detach(), ll.detach()
    assert np.abs(probs.cpu().numpy()[0] - g64).max() <= 2e-5
    assert np.all(np.abs(ll.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    la, ll32 = layer.forward_recursion(x)
    m = la64 > -30
    assert np.all(np.abs(la.cpu().numpy()[0] - la64)[m] <= 3e-4 + 2e-7 * np.abs(la64[m]))
    assert ll32.dtype == torch.float32 and ll32.shape == (1, b)
    lb = layer.backward_recursion(x)
    m = lb64 > -30
    assert np.all(np.abs(lb.cpu().numpy()[0] - lb64)[m] <= 3e-4 + 2e-7 * np.abs(lb64[m]))
    nol = layer.state_posterior_log_probs(x, no_loglik=True)
    assert float((nol - post - ll.to(torch.float32)[..., None, None]).abs().max()) < 0.05
    loglik, mean = layer(x)
    assert abs(float(mean.detach()) - ll64.mean()) <= 1e-6 * abs(ll64.mean()) + 1e-3
    assert torch.allclose(loglik, ll.to(torch.float32))


def test_impl_functions_keep_reference_signatures():
    cell, x, A, pi, E = gene_setup(2, 300, seed=1)
    rc = cell.make_reverse_direction_offspring()
    la, ll, prior, aux = L5._forward_recursion_impl(x, cell, None, None, end_hints=None, return_prior=True,
                                                    training=False, parallel_factor=1)
    assert la.shape == (1, 2, 300, 15) and ll.shape == (1, 2) and aux == 0.0
    lb = L5._backward_recursion_impl(x, cell, rc, None, None, parallel_factor=3)
    with torch.no_grad():
        post, prior, aux = L5._state_posterior_log_probs_impl(x, cell, rc, None, None, None, return_prior=True,
                                                              parallel_factor=99)
    got = (la + lb - ll[..., None, None]).cpu().numpy()
    m = post.cpu().numpy() > -12
    assert np.abs(got - post.cpu().numpy())[m].max() < 2e-3          # fp32 log alpha + log beta - loglik
    # the reference's own fp32 CPU path on the same E
    p = ref_cell.HmmParams(A, pi)
    ref_post, ref_ll = ref_cell.posterior_scaled(p, torch.as_tensor(E)[None])
    assert np.abs(np.exp(post.cpu().numpy()) - ref_post.numpy()).max() < 1e-4
    assert np.abs(ll.cpu().numpy() - ref_ll.numpy()).max() < 1e-3 + 1e-6 * np.abs(ref_ll.numpy()).max()


def test_end_hints_and_training_flag():
    cell, x, A, pi, E = gene_setup(2, 64, seed=2)
    hints = torch.zeros(1, 2, 2, 15, device=DEV)
    hints[..., 0, 0] = 1.0            # left end: intergenic
    hints[..., 1, 0] = 1.0            # right end: intergenic
    layer = MsaHmmLayer(cell, use_prior=False)
    post = torch.exp(layer.state_posterior_log_probs(x, end_hints=hints)).detach()
    assert float(post[0, :, 0, 0].min()) > 1 - 1e-5 and float(post[0, :, -1, 0].min()) > 1 - 1e-5
    Eh = E.copy()
    Eh[:, 0, 1:] = 0
    Eh[:, -1, 1:] = 0
    g64, _ = textbook.posterior(A, pi, Eh)
    assert np.abs(post.cpu().numpy()[0] - g64).max() <= 2e-5
    tr_post = layer.state_posterior_log_probs(x, training=True).detach()
    assert torch.isfinite(tr_post).all()


def test_sequence_weights_and_prior_outputs():
    cell, x, A, pi, E = gene_setup(4, 200, seed=3)
    w = np.array([1.0, 2.0, 0.5, 4.0, 3.0])
    layer = MsaHmmLayer(cell, num_seqs=5, use_prior=True, sequence_weights=w).to(DEV)
    idx = torch.tensor([[4, 0, 2, 1]], device=DEV)
    loglik, mean, prior, aux = layer(x, indices=idx)
    ll64 = textbook.loglik(A, pi, E)
    want = (ll64 * w[[4, 0, 2, 1]]).sum() / w[[4, 0, 2, 1]].sum()
    assert abs(float(mean.detach()) - want) <= 1e-6 * abs(want) + 1e-3
    assert float(prior) == 0.0 and aux == 0.0


def test_fused_emitter_matches_torch_emitter():
    """hmm_gene_emissions (one HIP kernel) vs GenePredHMMEmitter.forward in torch vs the oracle's
    CPU restatement, incl. N nucleotides, sequence borders, training offset, copies, no sharing."""
    from hmm_layer_amd import engine
    g = torch.Generator().manual_seed(9)
    b, L = 5, 333
    cls = torch.softmax(2 * torch.randn((1, b, L, 15), generator=g), -1)
    nuc = torch.nn.functional.one_hot(torch.randint(0, 5, (1, b, L), generator=g), 5).float()
    x = torch.cat([cls, nuc], -1)
    tab = params.codon_table(**params.DEFAULT_CODONS)
    for kw in (dict(), dict(num_copies=2, share_intron_parameters=False), dict(n_mass_compat=True)):
        em = GenePredHMMEmitter(**CODONS, **kw)
        em.build((1, b, L, 15))
        with torch.no_grad():
            em.emission_kernel.copy_(torch.randn(em.emission_kernel.shape, generator=g))
        em = em.to(DEV)
        em.recurrent_init()
        xd = x.to(DEV)
        assert em.can_fuse(xd)
        for training in (False, True):
            want = em(xd, training=training)
            got = em.forward_fused(xd, training=training)
            assert got.shape == want.shape
            assert float(((got - want.detach()).abs() / (want.detach().abs() + 1e-30)).max()) < 2e-5
        if not kw:
            cpu = params.gene_emissions(x, em.emission_kernel.detach().cpu(), tab).numpy()
            assert np.abs(em.forward_fused(xd).cpu().numpy() - cpu).max() <= 1e-6 * np.abs(cpu).max()
        hints = torch.rand((1, b, 2, em.num_states), generator=g).to(DEV)
        assert torch.allclose(em.forward_fused(xd, end_hints=hints), em(xd, end_hints=hints), rtol=2e-5, atol=0)


@pytest.mark.parametrize("b,L", [(3, 5), (7, 16), (4, 37), (2, 1100)])
def test_fused_emitter_soft_nucleotides_and_short_sequences(b, L):
    """The generic path of the fused kernel: soft nucleotide distributions, N flags that are not
    exactly 1, an N flag next to a base, and sequences shorter than a 16-position tile (every window
    crosses a border) — against the torch emitter."""
    g = torch.Generator().manual_seed(100 * b + L)
    cls = torch.softmax(2 * torch.randn((1, b, L, 15), generator=g), -1)
    idx = torch.randint(0, 5, (1, b, L), generator=g)
    nuc = torch.nn.functional.one_hot(idx, 5).float()
    kind = torch.rand((1, b, L), generator=g)
    soft = torch.softmax(torch.randn((1, b, L, 5), generator=g), -1)
    nuc = torch.where((kind < 0.15)[..., None], soft, nuc)                    # genuinely soft rows
    nuc[..., 4] = torch.where((kind >= 0.15) & (kind < 0.2), torch.full_like(kind, 0.5), nuc[..., 4])   # N flag != 1
    both = (kind >= 0.2) & (kind < 0.25)
    nuc[..., 4] = torch.where(both, torch.ones_like(kind), nuc[..., 4])       # N == 1 next to a base
    x = torch.cat([cls, nuc], -1).to(DEV)
    for kw in (dict(), dict(n_mass_compat=True)):
        em = GenePredHMMEmitter(**CODONS, **kw)
        em.build((1, b, L, 15))
        with torch.no_grad():
            em.emission_kernel.copy_(torch.randn(em.emission_kernel.shape, generator=g))
        em = em.to(DEV)
        em.recurrent_init()
        with torch.no_grad():
            want = em(x)
        got = em.forward_fused(x)
        assert float(((got - want).abs() / (want.abs() + 1e-30)).max()) < 2e-5


def test_layer_uses_fused_emitter_and_scales():
    """The layer path (fused emitter -> engine) at a size where the torch emitter's (b,L,64)
    intermediates would be 10x the input: b = 64 x L = 50 000."""
    cell, x, A, pi, E = gene_setup(2, 400, seed=4)
    layer = MsaHmmLayer(cell, use_prior=False)
    with torch.no_grad():                              # inference: fused emitter -> engine
        post = layer.state_posterior_log_probs(x)
    g64, _ = textbook.posterior(A, pi, E)
    assert np.abs(np.exp(post.cpu().numpy()[0]) - g64).max() <= 2e-5
    big = torch.cat([torch.softmax(torch.randn((1, 64, 50000, 15), device=DEV), -1),
                     torch.nn.functional.one_hot(torch.randint(0, 5, (1, 64, 50000), device=DEV), 5).float()], -1)
    with torch.no_grad():
        probs, ll = layer.state_posterior_probs(big)
    assert bool(torch.isfinite(probs).all()) and float((probs.sum(-1) - 1).abs().max()) < 2e-5


def test_layer_forward_is_trainable():
    """loss.backward() through MsaHmmLayer.forward: ONE autograd node whose backward is the engine's
    analytic gradient.  Checked against the reference's mechanism — autograd through the restated
    cell loop (oracle.ref_cell.loglik_grad) chained into the same producers on the CPU."""
    b, L = 4, 400
    cell, x, A, pi, E = gene_setup(b, L, seed=3)
    ccell = gene_setup(b, L, seed=3, device="cpu")[0]  # same seed: identical parameters, on the CPU
    w = torch.tensor([0.5, 1.0, 2.0, 1.5])
    layer = MsaHmmLayer(cell, use_prior=False, sequence_weights=w)
    layer.build(x.shape)
    idx = torch.arange(b, device=DEV)
    loglik, mean = layer(x, indices=idx, training=True)
    assert mean.requires_grad
    (-mean).backward()
    got = {n: p.grad.detach().cpu() for n, p in cell.named_parameters() if p.grad is not None}
    assert "transitioner.transition_kernel" in got and any("emission_kernel" in n for n in got)

    # reference mechanism on the CPU
    ccell.recurrent_init()
    Ec = ccell.emission_probs(x.cpu(), end_hints=None, training=True).to(torch.float32)
    Ac = ccell.A
    pic = ccell.init_dist.reshape(1, 15)
    gl = (-(w / w.sum()))[None]                                          # d(-weighted mean)/d loglik
    dA, dpi, dE, ll_ref = ref_cell.loglik_grad(Ac.detach(), pic.detach(), Ec.detach(), gl)
    plist = [(n, p) for n, p in ccell.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad([Ac, pic, Ec], [p for _, p in plist], [dA, dpi, dE], allow_unused=True)
    assert np.abs(loglik.detach().cpu().numpy() - ll_ref.numpy()).max() <= 1e-6 * np.abs(ll_ref.numpy()).max() + 2e-3
    checked = 0
    for (n, _), gr in zip(plist, grads):
        if gr is None:
            assert n not in got or float(got[n].abs().max()) == 0.0
            continue
        scale = float(gr.abs().max())
        assert float((got[n] - gr).abs().max()) <= 5e-4 * scale + 1e-7, (n, float((got[n] - gr).abs().max()), scale)
        checked += 1
    assert checked >= 2


def test_inference_calls_build_no_graph():
    cell, x, A, pi, E = gene_setup(2, 200, seed=4)
    layer = MsaHmmLayer(cell, use_prior=False)
    layer.build(x.shape)
    with torch.no_grad():
        loglik, mean = layer(x)
    assert not mean.requires_grad and not loglik.requires_grad
    loglik2, mean2 = layer(x)                         # grad mode: same values through the autograd node
    assert mean2.requires_grad
    assert torch.allclose(loglik, loglik2.detach(), rtol=0, atol=2e-3)


def test_two_copy_gene_model_29_states():
    """GenePredMultiHMMTransitioner with two gene copies: 29 states — the fused emitter's generic
    tile path feeding the one-wave-per-sequence kernels (17..64 states), against the fp64 oracle on
    the producers' own A, pi, E."""
    b, L = 3, 450
    g = torch.Generator().manual_seed(12)
    cls = torch.softmax(2 * torch.randn((1, b, L, 15), generator=g), -1)
    nuc = torch.nn.functional.one_hot(torch.randint(0, 5, (1, b, L), generator=g), 5).float()
    x = torch.cat([cls, nuc], -1).to(DEV)
    em = GenePredHMMEmitter(**CODONS, num_copies=2)
    em.build((1, b, L, 15))
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    cell = HmmCell([29], 15, em, tr).to(DEV)
    layer = MsaHmmLayer(cell, use_prior=False)
    layer.build(x.shape)
    with torch.no_grad():
        probs, ll = layer.state_posterior_probs(x)
        cell.recurrent_init()
        A = cell.A[0].double().cpu().numpy()
        pi = cell.init_dist.reshape(-1).double().cpu().numpy()
        E = cell.emission_probs(x, end_hints=None, training=False)[0].double().cpu().numpy()
    assert probs.shape == (1, b, L, 29)
    g64, ll64 = textbook.posterior(A, pi, E)
    assert np.abs(probs.cpu().numpy()[0] - g64).max() <= 2e-5
    assert np.all(np.abs(ll.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    la, _ = layer.forward_recursion(x)
    la64, _ = textbook.log_alpha(A, pi, E)
    m = la64 > -30
    assert np.all(np.abs(la.cpu().numpy()[0] - la64)[m] <= 3e-4 + 2e-7 * np.abs(la64[m]))


def test_two_copy_gene_model_is_trainable():
    """loss.backward() through the 29-state model: the analytic backward of the one-wave-per-sequence
    path against autograd through the restated reference loop on the same A, pi, E."""
    b, L = 2, 160
    g = torch.Generator().manual_seed(13)
    cls = torch.softmax(2 * torch.randn((1, b, L, 15), generator=g), -1)
    nuc = torch.nn.functional.one_hot(torch.randint(0, 4, (1, b, L), generator=g), 5).float()
    x = torch.cat([cls, nuc], -1).to(DEV)
    em = GenePredHMMEmitter(**CODONS, num_copies=2)
    em.build((1, b, L, 15))
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    cell = HmmCell([29], 15, em, tr).to(DEV)
    layer = MsaHmmLayer(cell, use_prior=False)
    layer.build(x.shape)
    loglik, mean = layer(x, training=True)
    (-mean).backward()
    gk = cell.transitioner.transition_kernel.grad
    assert gk is not None and bool(torch.isfinite(gk).all()) and float(gk.abs().max()) > 0
    # reference mechanism on the same producers' outputs
    cell.recurrent_init()
    E = cell.emission_probs(x, end_hints=None, training=True).to(torch.float32)
    A, pi = cell.A, cell.init_dist.reshape(1, 29)
    gl = torch.full((1, b), -1.0 / b)
    dA, dpi, dE, ll_ref = ref_cell.loglik_grad(A.detach().cpu(), pi.detach().cpu(), E.detach().cpu(), gl)
    want = torch.autograd.grad([A, E], [cell.transitioner.transition_kernel], [dA.to(DEV), dE.to(DEV)])[0]
    assert float((gk - want).abs().max()) <= 5e-4 * float(want.abs().max()) + 1e-7
    assert np.abs(loglik.detach().cpu().numpy() - ll_ref.numpy()).max() <= 2e-3


def test_viterbi_wrapper_on_the_gene_model():
    """hmm_layer_amd.Viterbi.viterbi(inputs, cell): log A / log pi / log E built like the layer builds
    its engine inputs, one hmm_viterbi call; bit-exact against the Q16 oracle on the same logs."""
    from hmm_layer_amd import Viterbi
    from oracle import build as obuild
    cell, x, A, pi, E = gene_setup(3, 900, seed=6)
    path, score = Viterbi.viterbi(x, cell)
    assert path.shape == (1, 3, 900) and path.dtype == torch.int32 and score.shape == (1, 3)
    At, pit, Et = L5._engine_inputs(x, cell, None, False)
    logE = torch.log(torch.clamp_min(Et, cell.epsilon))[0].cpu().numpy()
    logA = torch.log(At)[0].cpu().numpy()
    logpi = torch.log(torch.clamp_min(pit, cell.epsilon))[0].cpu().numpy()
    wp, ws = obuild.viterbi(logA, logpi, logE)
    assert np.array_equal(path[0].cpu().numpy(), wp) and np.array_equal(score[0].cpu().numpy(), ws)
    # every step of the path is an edge of the model
    p = path[0].cpu().numpy()
    assert bool((A[p[:, :-1], p[:, 1:]] > 0).all())


def test_training_through_state_posteriors():
    """A loss on state_posterior_log_probs(training=True) — what the reference's own test script
    drives (tests/parallel_rnn_forward.py:70-80) — back-propagated by the engine's analytic backward;
    parameter gradients against autograd through the restated reference formula on the CPU."""
    b, L = 3, 250
    cell, x, A, pi, E = gene_setup(b, L, seed=8)
    ccell = gene_setup(b, L, seed=8, device="cpu")[0]
    g = torch.Generator().manual_seed(21)
    target = torch.softmax(torch.randn((1, b, L, 15), generator=g), -1)      # a soft labelling to fit
    layer = MsaHmmLayer(cell, use_prior=False)
    layer.build(x.shape)
    logp = layer.state_posterior_log_probs(x, training=True)
    assert logp.requires_grad
    loss = -(target.to(DEV) * logp).sum() / (b * L)                             # cross-entropy
    loss.backward()
    got = {n: p.grad.detach().cpu() for n, p in cell.named_parameters() if p.grad is not None}
    # reference mechanism: autograd through the restated loops (oracle.ref_cell.posterior_log_probs)
    ccell.recurrent_init()
    Ec = ccell.emission_probs(x.cpu(), end_hints=None, training=True).to(torch.float32)
    lp = ref_cell.posterior_log_probs(ref_cell.HmmParams(ccell.A, ccell.init_dist.reshape(1, 15)), Ec)
    lp = lp[0] if isinstance(lp, tuple) else lp
    ref_loss = -(target * lp).sum() / (b * L)
    plist = [(n, p) for n, p in ccell.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(ref_loss, [p for _, p in plist], allow_unused=True)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-4 * abs(float(ref_loss.detach())) + 1e-4
    checked = 0
    for (n, _), gr in zip(plist, grads):
        if gr is None:
            continue
        scale = float(gr.abs().max())
        assert float((got[n] - gr).abs().max()) <= 2e-3 * scale + 1e-7, (n, float((got[n] - gr).abs().max()), scale)
        checked += 1
    assert checked >= 2
