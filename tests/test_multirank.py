"""N > 1 path: world_size-2 runs over gloo (127.0.0.1).  CPU: the reduce-buffer protocol + host logic;
GPU box: the real sharded ICP through the C ABI with the callback transport (2 ranks share the one GPU)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_workers(mode, nproc=2, timeout=300):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "mr_worker.py"), mode]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout


def test_sharded_kabsch_protocol_world2_gloo():
    out = run_workers("protocol", 2)
    assert out.count("protocol ok") == 2


def test_sharded_kabsch_protocol_world3_gloo():
    out = run_workers("protocol", 3)
    assert out.count("protocol ok") == 3


@pytest.mark.gpu
def test_sharded_icp_two_ranks_one_gpu():
    out = run_workers("gpu", 2)
    assert out.count("gpu sharded icp ok") == 2


@pytest.mark.gpu
def test_sharded_icp_four_ranks_one_gpu():
    """uneven shards (20 003 points over 4 ranks): four ranks share the one GPU over gloo.  The GPU box allows 6 processes
    on the card and the pytest process itself holds it too, so 4 workers is the safe maximum here.  The pose must be
    identical on every rank and equal to the single-rank pose."""
    out = run_workers("gpu", 4, timeout=600)
    assert out.count("gpu sharded icp ok") == 4


@pytest.mark.gpu
def test_sharded_icp_two_ranks_two_gpus_rccl_device_path(pcr):
    """The native transport with MORE than one rank: one process per GPU, ncclAllReduce enqueued on the context stream between
    icp_reduce_slots and icp_update_from_sums (csrc/icp.cpp).  Needs two GPUs in this box; the single-GPU boxes skip it
    (the driver's 8-GPU scaling run is then the first multi-rank RCCL execution)."""
    if pcr.device_count() < 2:               # (no torch in this process: it would bring a second HIP runtime + RCCL with it)
        pytest.skip("one GPU visible: the RCCL device path needs one GPU per rank")
    out = run_workers("rccl", 2)
    assert out.count("rccl sharded icp ok") == 2


@pytest.mark.gpu
def test_native_rccl_communicator_single_rank(pcr, synth):
    """RCCL is bound with dlopen (ncclGetUniqueId / ncclCommInitRank with a by-value 128-byte id / ncclAllReduce):
    a one-rank communicator exercises that ABI end to end, and ICP runs unchanged with it attached."""
    import numpy as np
    ctx = pcr.Context(0)
    try:
        uid = pcr.comm_unique_id()
        assert len(uid) == 128
        for bad in (25, 64, 1000):               # the reduce buffer holds 16 + 2 * nranks <= 64 f64 (PCR_MAX_RANKS = 24)
            with pytest.raises(pcr.PcrError):
                ctx.comm_init_rccl(bad, 0, uid)
            with pytest.raises(pcr.PcrError):
                ctx.comm_init_callback(bad, 0, lambda a: None)
        ctx.comm_init_rccl(1, 0, uid)
        ctx.comm_selftest()
        src, tgt = synth.kitti_like_pair(5000, seed_target=91, seed_pair=92)
        cs, ct = ctx.cloud(src), ctx.cloud(tgt)
        T1, st1 = ctx.icp_point2point(cs, ct, max_iter=6)
        # the multi-rank paths with a real ncclAllReduce per iteration (sum over one rank): reduce slots -> all-reduce ->
        # update, in the device-resident loop (1) and in the synchronous loop (-1), for both correspondence searches
        ctx.tune("icp_force_slots", 1)
        for method in (1, 2):
            ctx.tune("nn_method", method)
            for pipe in (1, -1):
                ctx.tune("icp_pipeline", pipe)
                T2, st2 = ctx.icp_point2point(cs, ct, max_iter=6)
                assert np.array_equal(T1, T2) and st1["last_pairs"] == st2["last_pairs"], (method, pipe)
        for k in ("icp_force_slots", "nn_method", "icp_pipeline"):
            ctx.tune(k, 0)
        ctx.comm_destroy()
        T3, st3 = ctx.icp_point2point(cs, ct, max_iter=6)
        assert np.array_equal(T1, T3) and st1["last_pairs"] == st3["last_pairs"]
    finally:
        ctx.close()
