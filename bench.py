#!/usr/bin/env python3
"""Headline benchmark: HMM cell-updates/s (batch x len x states / time) of one forward-backward
posterior pass of the 15-state gene model on b = 1024 sequences of L = 100 000 per GPU
(BASELINE.json configs[2]), inputs resident in HBM, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W --scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself:
the parent never touches the GPU, it runs torch.distributed.run (one process per GPU, rendezvous on
127.0.0.1) as a child process and passes its output and exit code on.

A step = one hmm_posterior pass over the rank's batch + the log-likelihood aggregate
(MsaHmmLayer.apply_sequence_weights): per-model (sum w*ll, sum w) on device and, for N > 1,
ONE RCCL all-reduce of those two numbers — the only collective on the path.  Sequences are
independent, so ranks own disjoint batches and nothing else is exchanged.  --scaling weak (default):
every rank owns --batch sequences; --scaling strong: the --batch sequences are split over the ranks
(1024 / 8 = 128 per GPU).

Rank 0 prints one JSON line with the contract's fields plus
  roofline      the dominant kernel's algorithmic-bytes rate from HIP events recorded on the
                launch stream inside the timed region (all kernels are listed under "kernels")
  cpu_baseline  the oracle's PyTorch-CPU port of the reference path timed on this host
  accuracy      (N = 1, outside the timed region; the oracle is the checker here, nothing measured)
                BASELINE.md section 3's gates on sampled sequences of the timed batch: max |gamma - gamma64|,
                relative log-likelihood error, Viterbi paths bit-exact against the CPU oracle
  variants      (N = 1 only, outside the timed region) the other passes SURVEY.md 8(d) names, on
                the same batch: log-likelihood only, Viterbi, log-likelihood gradients; plus posterior
                gradients at the reference's test size, the
                fused emitter (E producer), the 1027-state profile-HMM shape (configs[4]) and
                `pipeline_input`: the posterior pass on EMITTER-GENERATED emissions (class probabilities +
                one-hot nucleotides -> hmm_gene_emissions -> hmm_posterior, SURVEY.md 8(d)'s parity recipe) with
                the number of sequences the device routed to the serial recomputation and their accuracy
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured achievable
MFMA_F32_PEAK_TFLOPS = 157.3

# algorithmic HBM bytes per cell-update (one (sequence, position, state)), fp32 — SURVEY.md 8(d)
ALG_BYTES = {"reduce": 4.0,      # read E once
             "forward": 4.0,     # read E once (checkpoints are 1/16 of a row per step)
             "backward": 8.0,    # read E + write gamma
             "scan": 0.0,
             "exact": 0.0}       # serial exact-clamp kernels: empty launches unless the device routes sequences there
ALG_BYTES_JOB = 8.0              # whole fwd-bwd posterior: read E once + write gamma once
# useful flops per (sequence, position) in the reduce kernel: one 16x16x16 product
REDUCE_FLOPS_PER_STEP = 2.0 * 16 * 16 * 16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="sequences per GPU")
    ap.add_argument("--len", type=int, default=100000)
    ap.add_argument("--states", type=int, default=15)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch sequences per GPU; strong: --batch sequences split over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--cpu-len", type=int, default=20000, help="sequence length of the CPU baseline sample")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="per-kernel HBM bytes per launch from a separate rocprofv3 --pmc pass")
    return ap.parse_args()


def spawn_ranks(args):
    """N > 1 without a launcher: start one process per GPU through torch.distributed.run as a CHILD
    process (this parent has not touched the GPU and never will) and hand its result on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def gene_model(q, device):
    """Intended 23-edge 15-state A (tests/parallel_rnn_forward.py kwargs) + uniform start."""
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    tr = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000,
                                      starting_distribution_init="zeros").to(device)
    with torch.no_grad():
        A = tr.make_A().contiguous()
        pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
    assert A.shape[-1] == q
    return A, pi


def cpu_baseline(L_cpu, batch, q):
    """The reference CPU path (oracle/ref_cell.py: op-for-op PyTorch-CPU restatement of the
    HmmCell loop, forward + reverse + posterior assembly) on bounded samples: the headline batch at
    a reduced length (the recurrence is O(L): cells/s is stable in L), and the two sizes SURVEY.md
    8(d) names (b=1024 x L=2000; b=32 x L=9999 = the reference's own test size)."""
    from oracle import ref_cell, params
    torch.manual_seed(0)
    # the eager loop runs ~20 small ATen ops per step: more threads than the box's CPU share
    # for one GPU (16) only adds fork/join cost
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    A = params.intended_A15()
    p = ref_cell.HmmParams(A, torch.full((q,), 1.0 / q))

    def run(b, L):
        E = torch.rand((1, b, L, q)) * 0.9 + 0.05
        t0 = time.perf_counter()
        ref_cell.posterior_scaled(p, E)
        dt = time.perf_counter() - t0
        return {"batch": b, "len": L, "seconds": dt, "cell_updates_per_s": b * L * q / dt}

    ref_cell.posterior_scaled(p, torch.rand((1, batch, 50, q)) * 0.9 + 0.05)            # warm-up
    main = run(batch, L_cpu)
    others = [run(1024, 2000), run(32, 9999)]
    return {"value": main["cell_updates_per_s"], "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": "b=%d x L=%d x q=%d fwd-bwd posterior, PyTorch-CPU eager loop over the cell step "
                      "(oracle/ref_cell.py), %.1f s" % (batch, L_cpu, q, main["seconds"]),
            "other_samples": others}


def accuracy(engine, A, pi, E, out, ll, nsample=4):
    """BASELINE.md section 3's accuracy gates on sampled sequences of the timed batch, outside the timed
    region.  The oracle (fp64 C twin of oracle/textbook.py, Q16 Viterbi) is the CHECKER here."""
    from oracle import build as obuild
    import numpy as np
    _, b, L, q = E.shape
    idx = sorted(set(int(i) for i in np.linspace(0, b - 1, nsample)))
    An, pin = A[0].cpu().numpy(), pi.reshape(-1).cpu().numpy()
    Es = E[0, idx].cpu().numpy()
    g64, ll64 = obuild.posterior(An, pin, Es)
    got, gll = out[0, idx].cpu().numpy(), ll[0, idx].cpu().numpy()
    with np.errstate(divide="ignore"):
        logA, logpi = np.log(An).astype(np.float32), np.log(pin).astype(np.float32)
    logE = torch.log(E[:, idx].contiguous())
    path, score = engine.viterbi(torch.as_tensor(logA, device=E.device)[None], torch.as_tensor(logpi, device=E.device)[None], logE)
    wp, ws = obuild.viterbi(logA, logpi, logE[0].cpu().numpy())
    return {"sampled_sequences": idx, "len": L,
            "max_abs_gamma_err_vs_fp64": float(np.abs(got - g64).max()),
            "max_rel_loglik_err_vs_fp64": float(np.max(np.abs(gll - ll64) / np.abs(ll64))),
            "max_abs_row_sum_err": float(np.abs(got.sum(-1) - 1).max()),
            "viterbi_paths_bit_exact": bool(np.array_equal(path[0].cpu().numpy(), wp)),
            "viterbi_scores_bit_exact": bool(np.array_equal(score[0].cpu().numpy(), ws)),
            "tolerances": {"gamma": 2e-5, "loglik_rel": 1e-6},
            "checker": "oracle/hmm_oracle.c (fp64 scaled forward-backward with the reference's clamps; Q16 Viterbi)"}


def variants(engine, A, pi, E, reps=3):
    """Secondary passes on the same resident batch: ms per pass and cell-updates/s, with the
    algorithmic-bytes rate of each (loglik only 4 B/cell, Viterbi 4 + 4/q B/cell,
    gradients 8 B/cell: read E, write dE)."""
    _, b, L, q = E.shape
    cells = float(b) * L * q

    def timed(fn):
        # warm-up: at least three calls and 30 ms of GPU work (the accuracy block before this leaves the device
        # idle for a second of CPU work, and the first kernels after an idle spell run at a lower clock)
        t0 = time.perf_counter()
        n = 0
        while n < 3 or time.perf_counter() - t0 < 0.03:
            fn()
            torch.cuda.synchronize()
            n += 1
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def entry(dt, alg_bytes):
        return {"ms": dt * 1e3, "cell_updates_per_s": cells / dt, "alg_bytes_per_cell": alg_bytes,
                "alg_GBps": alg_bytes * cells / dt / 1e9, "hbm_frac": alg_bytes * cells / dt / 1e9 / HBM_PEAK_GBS}

    res = {}
    res["loglik_only"] = entry(timed(lambda: engine.forward(A, pi, E, want_log_alpha=False)), 4.0)
    res["loglik_grad"] = entry(timed(lambda: engine.loglik_grad(A, pi, E)), 8.0)
    logE = torch.log(E)
    logA = torch.log(A.clamp_min(1e-30))
    logpi = torch.log(pi)
    res["viterbi"] = entry(timed(lambda: engine.viterbi(logA, logpi, logE)), 4.0 + 4.0 / q)
    del logE
    # BASELINE configs[1]: b = 256 x L = 10 000, forward only (log-likelihood, and with log alpha)
    E2 = E[:, :256, :10000].contiguous()
    c2 = float(256) * 10000 * q
    t = timed(lambda: engine.forward(A, pi, E2, want_log_alpha=False))
    t2 = timed(lambda: engine.forward(A, pi, E2))
    res["config2_forward_only_b256_L10000"] = {"loglik_ms": t * 1e3, "loglik_cell_updates_per_s": c2 / t,
                                                "log_alpha_ms": t2 * 1e3, "log_alpha_cell_updates_per_s": c2 / t2}
    # the strong-scaling shard of the headline config at 8 GPUs: b = 128 x L = 100 000
    E3 = E[:, :128].contiguous()
    t = timed(lambda: engine.posterior(A, pi, E3))
    res["strong_scaling_shard_b128"] = {"ms": t * 1e3, "cell_updates_per_s": float(128) * L * q / t,
                                        "chunk_len": engine.chunk_len(1, 128, L, q)}
    del E2, E3
    res["two_copy_gene_model_q29"] = two_copy_variant(engine, timed)
    res["three_copy_gene_model_q43"] = three_copy_variant(engine, timed)
    res["posterior_grad_train_shape"] = postgrad_variant(engine, A, pi, timed)
    res["gene_emitter"] = emitter_variant(engine, b, L, timed)
    res["pipeline_input"] = pipeline_variant(engine, A, pi, b, L, timed)
    res["profile_hmm_q1027"] = largeq_variant(engine, timed)
    return res


def two_copy_variant(engine, timed, b=1024, L=100000):
    """The 29-state two-copy gene model (GenePredMultiHMMTransitioner(k=2)) at the headline batch: the chunked
    32-state scan (sparse 45-edge reduce, two-tile MFMA apply kernels); 8 B per cell-update algorithmic."""
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    dev = torch.device("cuda", torch.cuda.current_device())
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
    with torch.no_grad():
        A = tr.make_A().contiguous()
        pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
    q = A.shape[-1]
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    out = torch.empty_like(E)
    dt = timed(lambda: engine.posterior(A, pi, E, out=out))
    nserial = engine.exact_count(engine.OP_POSTERIOR, (1, b, L, q))       # sequences the certificate sent to the serial kernels
    dl = timed(lambda: engine.forward(A, pi, E, want_log_alpha=False))
    cells = float(b) * L * q
    del out
    # Viterbi (bit-exact Q16, one wave per sequence with the sparse step: hmm_viterbi.inc, k_mq_viterbi)
    logA, logpi = torch.log(A), torch.log(pi)
    E.log_()
    dv = timed(lambda: engine.viterbi(logA, logpi, E))
    del E
    # training at the reference's own test size (b = 32, L = 9 999): both gradients per chunk of the 32-state scan
    # plan, and as whole-sequence sweeps beside them
    bt, Lt = 32, 9999
    Et = torch.rand((1, bt, Lt, q), device=dev) * 0.9 + 0.05
    gam, _ = engine.posterior(A, pi, Et, mode=engine.POST_PROB)
    lab = torch.multinomial(gam.reshape(-1, q).clamp_min(0) + 1e-30, 1).reshape(1, bt, Lt, 1)
    G = torch.zeros((1, bt, Lt, q), device=dev).scatter_(3, lab, -1.0)
    del gam, lab
    train = {"batch": bt, "len": Lt}
    for how, tag in ((1, ""), (0, "_whole_sequence_sweeps")):
        with engine.option(engine.OPT_PGCHUNK, how):
            train["loglik_grad%s_ms" % tag] = timed(lambda: engine.loglik_grad(A, pi, Et)) * 1e3
            if how: train["loglik_grad_serial_sequences"] = engine.loglik_grad_serial_count((1, bt, Lt, q))
            train["posterior_grad%s_ms" % tag] = timed(lambda: engine.posterior_grad(A, pi, Et, G, mode=engine.POST_LOG)) * 1e3
            if how: train["posterior_grad_serial_sequences"] = engine.posterior_grad_serial_count((1, bt, Lt, q))
    return {"ms": dt * 1e3, "loglik_ms": dl * 1e3, "viterbi_ms": dv * 1e3, "batch": b, "len": L, "states": q,
            "cell_updates_per_s": cells / dt,
            "alg_GBps": 8.0 * cells / dt / 1e9, "hbm_frac": 8.0 * cells / dt / 1e9 / HBM_PEAK_GBS,
            "serial_sequences": nserial, "train_shape": train}


def three_copy_variant(engine, timed, b=1024, L=100000):
    """The 43-state three-copy gene model (GenePredMultiHMMTransitioner(k=3)) at the headline batch: one wave per
    sequence with the sparse step (hmm_midq.inc; the chunked 64-state scan serves up to 56 sequences of it)."""
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    dev = torch.device("cuda", torch.cuda.current_device())
    tr = GenePredMultiHMMTransitioner(k=3, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
    with torch.no_grad():
        A = tr.make_A().contiguous()
        pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
    q = A.shape[-1]
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    out = torch.empty_like(E)
    dt = timed(lambda: engine.posterior(A, pi, E, out=out))
    dl = timed(lambda: engine.forward(A, pi, E, want_log_alpha=False))
    del out
    logA, logpi = torch.log(A), torch.log(pi)
    E.log_()
    dv = timed(lambda: engine.viterbi(logA, logpi, E))
    del E
    cells = float(b) * L * q
    return {"ms": dt * 1e3, "loglik_ms": dl * 1e3, "viterbi_ms": dv * 1e3, "batch": b, "len": L, "states": q,
            "cell_updates_per_s": cells / dt, "alg_GBps": 8.0 * cells / dt / 1e9,
            "hbm_frac": 8.0 * cells / dt / 1e9 / HBM_PEAK_GBS}


def postgrad_variant(engine, A, pi, timed, b=32, L=9999):
    """Backward of a loss on log posteriors (hmm_posterior_grad) at the reference's own test size
    (b = 32, L = 9999, tests/parallel_rnn_forward.py:19-23); upstream gradient of a cross-entropy on
    log gamma against a labelling drawn from the posterior.  Per chunk of the scan plan where the device-side
    routing allows (serial_sequences = how many were redone by the whole-sequence sweeps), and the
    whole-sequence sweeps alone beside it."""
    q = A.shape[-1]
    E = torch.rand((1, b, L, q), device=A.device) * 0.9 + 0.05
    gam, _ = engine.posterior(A, pi, E, mode=engine.POST_PROB)
    lab = torch.multinomial(gam.reshape(-1, q).clamp_min(0) + 1e-30, 1).reshape(1, b, L, 1)
    G = torch.zeros((1, b, L, q), device=A.device).scatter_(3, lab, -1.0)
    del gam, lab
    dt = timed(lambda: engine.posterior_grad(A, pi, E, G, mode=engine.POST_LOG))
    nserial = engine.posterior_grad_serial_count((1, b, L, q))
    with engine.option(engine.OPT_PGCHUNK, 0):
        dw = timed(lambda: engine.posterior_grad(A, pi, E, G, mode=engine.POST_LOG))
    df = timed(lambda: engine.posterior(A, pi, E, mode=engine.POST_LOG))
    return {"ms": dt * 1e3, "whole_sequence_sweeps_ms": dw * 1e3, "forward_ms": df * 1e3, "batch": b, "len": L,
            "states": q, "serial_sequences": nserial, "cell_updates_per_s": float(b) * L * q / dt}


def emitter_variant(engine, b, L, timed):
    """E producer (hmm_gene_emissions): (b,L,15+5) class probabilities + one-hot nucleotides -> E (b,L,15).
    HBM-bound: 80 B in + 60 B out per position."""
    from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
    dev = torch.device("cuda", torch.cuda.current_device())
    em = GenePredHMMEmitter(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
                            intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)],
                            intron_end_pattern=[("AGN", .99), ("ACN", .01)])
    em.build((1, b, L, 15))
    em = em.to(dev)
    em.recurrent_init()
    x = torch.empty((1, b, L, 20), device=dev)
    x[..., :15] = torch.softmax(torch.randn((1, b, L, 15), device=dev), -1)
    idx = torch.where(torch.rand((1, b, L), device=dev) < 0.01, torch.full((1, b, L), 4, device=dev),
                      torch.randint(0, 4, (1, b, L), device=dev))               # 1 % N
    x[..., 15:] = torch.nn.functional.one_hot(idx, 5).float()
    del idx
    dt = timed(lambda: em.forward_fused(x))
    nbytes = float(b) * L * (20 + 15) * 4
    return {"ms": dt * 1e3, "positions_per_s": b * L / dt, "alg_bytes_per_position": 140.0,
            "alg_GBps": nbytes / dt / 1e9, "hbm_frac": nbytes / dt / 1e9 / HBM_PEAK_GBS}


def gene_emitter(dev, b, L):
    from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
    em = GenePredHMMEmitter(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
                            intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)],
                            intron_end_pattern=[("AGN", .99), ("ACN", .01)])
    em.build((1, b, L, 15))
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        em.emission_kernel.copy_(torch.randn(em.emission_kernel.shape, generator=g))
    em = em.to(dev)
    em.recurrent_init()
    return em


def pipeline_variant(engine, A, pi, b, L, timed, nsample=3):
    """The pass on the pipeline's own kind of input (reference tests/parallel_rnn_forward.py:19-40):
    class probabilities softmax(scale * randn) (2 = the survey's recipe, 6 = peaked, what a trained classifier
    emits), one-hot nucleotides with 1 % N -> hmm_gene_emissions -> hmm_posterior.  47 % of these emissions are
    exact zeros; the device decides per sequence whether the cell's eps clamps matter (clamp-born posterior mass
    above 2e-6) and recomputes those serially in windows.  Reported: ms per posterior pass, how many sequences were
    routed (and how), the same pass with the routing off, and sampled sequences — routed ones first — against the
    fp64 oracle with the reference's clamps."""
    import numpy as np
    from oracle import build as obuild
    dev = A.device
    em = gene_emitter(dev, b, L)
    q = A.shape[-1]
    res = {"batch": b, "len": L, "n_fraction": 0.01, "tolerance_gamma": 2e-5}
    for scale in (2.0, 6.0):
        g = torch.Generator(device=dev).manual_seed(7)
        x = torch.empty((1, b, L, 20), device=dev)
        x[..., :15] = torch.softmax(scale * torch.randn((1, b, L, 15), device=dev, generator=g), -1)
        idx = torch.where(torch.rand((1, b, L), device=dev, generator=g) < 0.01, torch.full((1, b, L), 4, device=dev),
                          torch.randint(0, 4, (1, b, L), device=dev, generator=g))
        x[..., 15:] = torch.nn.functional.one_hot(idx, 5).float()
        del idx
        E = em.forward_fused(x).contiguous()
        del x
        out = torch.empty_like(E)
        dt = timed(lambda: engine.posterior(A, pi, E, out=out))
        det = engine.exact_detail((1, b, L, q))
        _, ll = engine.posterior(A, pi, E, out=out)
        with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
            d0 = timed(lambda: engine.posterior(A, pi, E))
        # sample: routed sequences first (the verdicts are in the workspace: compare against the routing-off pass)
        with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
            off, _ = engine.posterior(A, pi, E)
        changed = ((off - out).abs().amax(dim=(2, 3))[0] > 0).nonzero().reshape(-1).tolist()
        del off
        pick = (changed[:nsample] + [i for i in (0, b // 2, b - 1) if i not in changed])[:nsample]
        Es = E[0, pick].cpu().numpy()
        g64, ll64 = obuild.posterior(A[0].cpu().numpy(), pi.reshape(-1).cpu().numpy(), Es)
        err = float(np.abs(out[0, pick].cpu().numpy() - g64).max())
        lerr = float(np.max(np.abs(ll[0, pick].cpu().numpy() - ll64) / np.abs(ll64)))
        res["class_scale_%g" % scale] = {
            "ms": dt * 1e3, "cell_updates_per_s": float(b) * L * q / dt, "routing_off_ms": d0 * 1e3,
            "zero_emissions_fraction": float((E == 0).float().mean()),
            "routed_sequences": det["routed"], "window_sequences": det["window_sequences"], "windows": det["windows"],
            "whole_sequence_recomputations": det["whole"], "sampled_sequences": pick,
            "sampled_routed": [i for i in pick if i in changed],
            "max_abs_gamma_err_vs_fp64": err, "max_rel_loglik_err_vs_fp64": lerr}
        del E, out
    return res


def largeq_variant(engine, timed, q=1027, b=1024, L=64):
    """BASELINE configs[4] per-GPU shape (q = 2*512+3 states, 1024 sequences), forward log-likelihood:
    serial in time, one f32-MFMA GEMM (with the cell step in its epilogue) per position and direction;
    MFMA-bound (2 b q^2 flop per position and direction)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    A = torch.rand((1, q, q), device=dev) ** 4
    A = A / A.sum(-1, keepdim=True)
    pi = torch.full((1, q), 1.0 / q, device=dev)
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    dt = timed(lambda: engine.forward(A, pi, E, want_log_alpha=False))
    tf = 2.0 * b * q * q * L / dt / 1e12
    # posteriors: the forward and the backward recursion side by side on two streams (4 b q^2 flop per position)
    dp = timed(lambda: engine.posterior(A, pi, E))
    tp = 4.0 * b * q * q * L / dp / 1e12
    return {"ms": dt * 1e3, "us_per_position": dt / L * 1e6, "cell_updates_per_s": float(b) * L * q / dt,
            "batch": b, "len": L, "states": q, "TFLOPs": tf, "mfma_f32_frac": tf / MFMA_F32_PEAK_TFLOPS,
            "posterior_ms": dp * 1e3, "posterior_us_per_position": dp / L * 1e6,
            "posterior_cell_updates_per_s": float(b) * L * q / dp, "posterior_TFLOPs": tp,
            "posterior_mfma_f32_frac": tp / MFMA_F32_PEAK_TFLOPS}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from hmm_layer_amd import engine
    b, L, q = args.batch, args.len, args.states
    if args.scaling == "strong" and args.batch < world:
        raise SystemExit("--scaling strong needs at least one sequence per rank (--batch %d < %d ranks)" % (args.batch, world))
    if args.scaling == "strong":                  # the --batch sequences split over the ranks (contiguous shards)
        from hmm_layer_amd.distributed import shard_bounds
        lo, hi = shard_bounds(args.batch, rank, world)
        b = hi - lo
    torch.manual_seed(1234 + rank)
    A, pi = gene_model(q, dev)
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    out = torch.empty_like(E)
    prof = engine.Profile()

    def step(profile=None):
        _, ll = engine.posterior(A, pi, E, out=out, profile=profile)
        part = engine.loglik_partials(ll)                     # (1,2): sum ll, count
        if dist is not None:
            dist.all_reduce(part)                             # RCCL over xGMI, 16 bytes
        return part

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        part = step(prof)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern = prof.read()
    reduce_is_mfma = False       # the gene model is served by the sparse-topology (VALU) reduce kernel
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    mean_ll = float(part[0, 0] / part[0, 1])
    nseq_all = float(part[0, 1])                     # sequences of ALL ranks (the all-reduced weight sum)
    if dist is not None:
        world = dist.get_world_size()                # what RCCL actually runs on

    if rank == 0:
        cells_rank = float(b) * L * q
        cells_all = nseq_all * L * q
        value = cells_all * args.steps / dt
        kernels = {}
        for name, (ms, n) in kern.items():
            if n == 0:
                continue
            avg = ms / n
            kernels[name] = {"avg_ms": avg, "launches": n,
                             "alg_GBps": ALG_BYTES.get(name, 0.0) * cells_rank / (avg * 1e-3) / 1e9}
        # the dominant kernel among those that move the data (the scan and the routing launches have no
        # algorithmic bytes of their own)
        dom = max((k for k in kernels if ALG_BYTES.get(k, 0.0) > 0), key=lambda k: kernels[k]["avg_ms"])
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                traffic = json.load(open(args.traffic_json)).get(dom, {}).get("hbm_bytes")
            except Exception:
                traffic = None
        ach = kernels[dom]["alg_GBps"]
        roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": "%s: separate rocprofv3 --pmc passes of this command (tools/collect_traffic.sh), "
                                      "NOT measured in this run" % os.path.relpath(args.traffic_json, ROOT)}
        if dom == "reduce" and reduce_is_mfma:
            # the dense chunk-operator kernel is bounded by the f32 MFMA, not by HBM
            tf = REDUCE_FLOPS_PER_STEP * b * L / (kernels["reduce"]["avg_ms"] * 1e-3) / 1e12
            roofline.update({"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / MFMA_F32_PEAK_TFLOPS, "hbm_alg_GBps": ach})
        roofline.update({"alg_bytes_per_cell": ALG_BYTES[dom], "cells_per_launch": cells_rank,
                         "job_alg_GBps_per_gpu": ALG_BYTES_JOB * cells_rank * args.steps / dt / 1e9,
                         "job_frac": ALG_BYTES_JOB * cells_rank * args.steps / dt / 1e9 / HBM_PEAK_GBS,
                         "kernels": kernels})
        line = {
            "metric": "HMM cell-updates/sec (batch x len x states) fwd-bwd, 15-state model",
            "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Bidirectional fwd-bwd posteriors, 15-state gene model, "
                                   "batch=%d x len=%d %s (BASELINE configs[2])"
                                   % (args.batch, L, "per GPU" if args.scaling == "weak" else "in total, split over the GPUs"),
                       "batch_per_gpu": b, "batch_total": int(nseq_all), "seq_len": L, "states": q,
                       "chunk_len": engine.chunk_len(1, b, L, q),
                       "parallelism": "batch-sharded x%d, loglik all-reduce only" % world,
                       "mean_loglik": mean_ll},
            "roofline": roofline,
        }
        # everything below runs after the timed region, on rank 0 (its own shard of the batch); the other
        # ranks wait at the barrier that follows
        if not args.no_accuracy:
            _, ll = engine.posterior(A, pi, E, out=out)
            line["accuracy"] = accuracy(engine, A, pi, E, out, ll)
        if not args.no_variants and world == 1:
            del out
            line["variants"] = variants(engine, A, pi, E)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_len, b, q)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
