"""bench.py's contract on the GPU box: one JSON line with the driver's fields, the roofline /
cpu_baseline / accuracy objects, and — when the box has at least two GPUs — the self-spawned
multi-rank path (RCCL all-reduce of the log-likelihood aggregate, max-over-ranks timing)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_small_shape():
    line = run_bench("--steps", "2", "--warmup", "1", "--batch", "16", "--len", "3000", "--no-variants",
                     "--cpu-len", "200")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "accuracy"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["scaling"] == "weak" and line["dtype"] == "f32"
    assert abs(line["value"] - 16 * 3000 * 15 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and "traffic_source" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    acc = line["accuracy"]
    assert acc["max_abs_gamma_err_vs_fp64"] <= 2e-5 and acc["max_rel_loglik_err_vs_fp64"] <= 1e-6
    assert acc["viterbi_paths_bit_exact"] and acc["viterbi_scores_bit_exact"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's scaling run has them)")
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_self_spawned(scaling):
    line = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--len", "3000",
                     "--scaling", scaling, "--cpu-len", "200")
    assert line["n_gpus"] == 2 and line["scaling"] == scaling
    # an N > 1 line stands alone as a record: every object the N = 1 line has, except the N = 1 extras
    for key in ("roofline", "cpu_baseline", "accuracy"):
        assert key in line, key
    assert line["accuracy"]["max_abs_gamma_err_vs_fp64"] <= 2e-5 and line["cpu_baseline"]["kind"] == "port"
    total = 32 if scaling == "weak" else 16
    assert line["config"]["batch_total"] == total
    assert abs(line["value"] - total * 3000 * 15 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]


def _nccl_worker(rank, world, port, out):
    import torch.distributed as dist
    from hmm_layer_amd import distributed as hd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    g = torch.Generator().manual_seed(3)
    ll = (torch.randn((2, 33), generator=g, dtype=torch.float64) * 30 - 1e5)
    w = torch.rand((2, 33), generator=g)
    lo, hi = hd.shard_bounds(33, rank, world)
    got = hd.aggregate_loglik(ll[:, lo:hi].cuda(), w[:, lo:hi].cuda())
    out[rank] = float(got)
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_aggregate_loglik_over_rccl():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_nccl_worker, args=(2, port, out), nprocs=2, join=True)
    g = torch.Generator().manual_seed(3)
    ll = (torch.randn((2, 33), generator=g, dtype=torch.float64) * 30 - 1e5)
    w = torch.rand((2, 33), generator=g).double()
    want = float(((ll * w).sum(1) / w.sum(1)).mean())
    for r in range(2):
        assert abs(out[r] - want) <= 1e-9 * abs(want)


def test_loglik_allreduce_through_the_c_abi():
    """hmm_loglik_allreduce with a communicator the host made itself (one rank: what a box with one GPU
    can show — the values come back unchanged through RCCL's all-reduce on the caller's stream)."""
    import ctypes
    import glob
    from hmm_layer_amd import engine
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so"]
    rccl = ctypes.CDLL([c for c in cands if os.path.exists(c)][0], mode=ctypes.RTLD_GLOBAL)
    uid = ctypes.create_string_buffer(128)
    assert rccl.ncclGetUniqueId(uid) == 0
    comm = ctypes.c_void_p()

    class Uid(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
    u = Uid.from_buffer_copy(uid.raw)
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, u, 0) == 0
    try:
        ll = torch.randn((3, 50), dtype=torch.float64, device="cuda:0") * 10 - 1e4
        w = torch.rand((3, 50), device="cuda:0")
        part = engine.loglik_partials(ll, w)
        want = part.clone()
        engine.loglik_allreduce(comm, part)
        torch.cuda.synchronize()
        assert torch.equal(part, want)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
