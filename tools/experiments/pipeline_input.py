"""Posterior pass on emitter-generated gene-model input (SURVEY section 8(d)'s parity recipe): class probabilities
softmax(scale * randn), one-hot nucleotides with a share of N -> hmm_gene_emissions -> hmm_posterior.  Reports ms per
pass, how many sequences left the scan (windows / whole) and the error of sampled sequences against the fp64 oracle.
    python tools/experiments/pipeline_input.py [b] [L] [scale ...]"""
import os
import sys
import time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hmm_layer_amd import engine  # noqa: E402
from oracle import params, textbook  # noqa: E402


def gene_x(b, L, scale, pN, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    cls = torch.softmax(scale * torch.randn((b, L, 15), generator=g, device=dev), -1)
    idx = torch.randint(0, 4, (b, L), generator=g, device=dev)
    isn = torch.rand((b, L), generator=g, device=dev) < pN
    idx = torch.where(isn, torch.full_like(idx, 4), idx)
    nuc = torch.nn.functional.one_hot(idx, 5).float()
    return torch.cat([cls, nuc], -1)


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    scales = [float(v) for v in sys.argv[3:]] or [2.0, 6.0]
    dev = torch.device("cuda:0")
    from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
    em = GenePredHMMEmitter(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
                            intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)],
                            intron_end_pattern=[("AGN", .99), ("ACN", .01)])
    em.build((1, b, L, 15))
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        em.emission_kernel.copy_(torch.randn(em.emission_kernel.shape, generator=g))
    em = em.to(dev)
    A = params.intended_A15(200, 4500, 10000).to(dev).unsqueeze(0)
    pi = torch.full((1, 15), 1 / 15, device=dev)
    for scale in scales:
        x = gene_x(b, L, scale, 0.01, dev)
        with torch.no_grad():
            em.recurrent_init()
            E = em.forward_fused(x.unsqueeze(0))
        E = E.reshape(1, b, L, 15).contiguous()
        del x
        out = torch.empty_like(E)
        for mode_name, xm in (("auto", engine.EXACT_AUTO), ("off", engine.EXACT_OFF)):
            with engine.option(engine.OPT_EXACT, xm):
                engine.posterior(A, pi, E, out=out)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    _, ll = engine.posterior(A, pi, E, out=out)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 5 * 1e3
                det = engine.exact_detail((1, b, L, 15))
            # accuracy of a few sequences against the fp64 serial recursion
            idx = [0, b // 3, b - 1] if b >= 3 else list(range(b))
            Es = E[0, idx].cpu().numpy()
            g64, ll64 = textbook.posterior(A[0].cpu().numpy(), pi[0].cpu().numpy(), Es)
            err = float(np.abs(out[0, idx].cpu().numpy() - g64).max())
            lerr = float(np.abs(ll[0, idx].cpu().numpy() - ll64).max())
            print("scale %g b %d L %d routing %-4s: %.3f ms  %s  max|dgamma| %.2e  max|dll| %.2e (ll ~ %.3g)" % (
                scale, b, L, mode_name, ms, det, err, lerr, float(np.abs(ll64).max())), flush=True)
        del E, out


if __name__ == "__main__":
    main()
