"""The path's only collective, over gloo with world_size 2 on CPU: sharded batches give the same
aggregated log-likelihood as the unsharded batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hmm_layer_amd import distributed as hd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ll, w, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = hd.shard_bounds(ll.shape[1], rank, world)
    got = hd.aggregate_loglik(ll[:, lo:hi], None if w is None else w[:, lo:hi])
    out[rank] = float(got)
    dist.destroy_process_group()


def _run(world, ll, w):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ll, w, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


def test_shard_bounds_cover_batch():
    for b in (1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [hd.shard_bounds(b, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == b
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_aggregate_equals_unsharded_gloo():
    torch.manual_seed(0)
    ll = (torch.randn(2, 37, dtype=torch.float64) * 30 - 1e5)
    w = torch.rand(2, 37)
    want_w = float(((ll * w.double()).sum(1) / w.double().sum(1)).mean())
    want = float(ll.mean(1).mean())
    for got in _run(2, ll, w):
        assert abs(got - want_w) < 1e-9 * abs(want_w)
    for got in _run(2, ll, None):
        assert abs(got - want) < 1e-9 * abs(want)
    assert abs(float(hd.aggregate_loglik(ll, w)) - want_w) < 1e-9 * abs(want_w)      # no process group
    p = hd.loglik_partials(ll, w).numpy()
    np.testing.assert_allclose(p[:, 1], w.double().sum(1).numpy())


def _grad_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    ll = -100 * torch.rand((2, 10), generator=g, dtype=torch.float64)
    w = torch.rand((2, 10), generator=g) + 0.5
    lo, hi = hd.shard_bounds(10, rank, world)
    mine = ll[:, lo:hi].clone().requires_grad_(True)
    mean = hd.aggregate_loglik(mine, w[:, lo:hi])
    mean.backward()
    out[rank] = (float(mean), mine.grad.numpy(), lo, hi)
    dist.destroy_process_group()


def test_differentiable_aggregate_gloo():
    """Value = global weighted mean on every rank; gradient = w / sum_all(w) / k on the local shard."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g = torch.Generator().manual_seed(7)
    ll = -100 * torch.rand((2, 10), generator=g, dtype=torch.float64)
    w = (torch.rand((2, 10), generator=g) + 0.5).to(torch.float64)
    want = float(((w * ll).sum(1) / w.sum(1)).mean())
    wantg = (w / w.sum(1, keepdim=True) / 2).numpy()
    for rank in range(2):
        mean, grad, lo, hi = out[rank]
        assert abs(mean - want) <= 1e-12 * abs(want)
        assert np.abs(grad - wantg[:, lo:hi]).max() <= 1e-12


def _param_grad_worker(rank, world, port, reduction, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(11)
    c = -50 * torch.rand((2, 9), generator=g, dtype=torch.float64)
    w = torch.rand((2, 9), generator=g) + 0.5
    theta = torch.tensor([1.3, 0.7], dtype=torch.float64, requires_grad=True)     # a "parameter" every rank replicates
    lo, hi = hd.shard_bounds(9, rank, world)
    ll = theta[:, None] * c[:, lo:hi]
    hd.aggregate_loglik(ll, w[:, lo:hi], grad_reduction=reduction).backward()
    out[rank] = theta.grad.numpy()
    dist.destroy_process_group()


def test_parameter_gradient_under_summing_and_averaging_wrappers():
    """The rank gradients combine to the single-process parameter gradient: summed with
    grad_reduction="sum", averaged (what DistributedDataParallel does) with grad_reduction="mean"."""
    g = torch.Generator().manual_seed(11)
    c = -50 * torch.rand((2, 9), generator=g, dtype=torch.float64)
    w = torch.rand((2, 9), generator=g) + 0.5
    theta = torch.tensor([1.3, 0.7], dtype=torch.float64, requires_grad=True)
    hd.aggregate_loglik(theta[:, None] * c, w).backward()
    want = theta.grad.numpy()
    for reduction, combine in (("sum", np.sum), ("mean", np.mean)):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_param_grad_worker, args=(2, _free_port(), reduction, out), nprocs=2, join=True)
        got = combine(np.stack([out[0], out[1]]), axis=0)
        np.testing.assert_allclose(got, want, rtol=1e-12)
