"""Rehearsal of the command the scaling driver runs (`bench.py --gpus N`), on ONE GPU: PCR_BENCH_BACKEND=gloo lets N ranks share the
card and carries the per-iteration collective over torch.distributed / gloo (the host-callback transport), so everything but RCCL
itself executes: the self-launch, the rank / shard bookkeeping, the barriers and max-over-ranks timing, the `weak` and `c5` blocks
under world > 1, and rank 0's one JSON line.  The pose of the sharded runs must equal the one-rank pose BIT FOR BIT (the Kabsch
moments are exact integer limbs: csrc/numerics.hpp) — north_star: "one RCCL all-reduce of centroids + 3x3 covariance per ICP
iteration", SURVEY.md 8(e)."""
import json
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(gpus, *flags, timeout=420):
    # a rank that is still running after 300 s dumps its Python stacks and exits (bench.py PCR_BENCH_WATCHDOG_S): a deadlock says where;
    # the launcher and its ranks form one process group that is killed as a whole if the call does not come back in time — no rank may
    # outlive its test on the GPU (the box allows six processes on the card)
    env = dict(os.environ, PCR_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PCR_BENCH_WATCHDOG_S="300")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *flags]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        stdout, stderr = p.communicate(timeout=timeout)
    except BaseException:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        stdout, stderr = p.communicate()
        raise AssertionError(f"bench.py --gpus {gpus} did not finish within {timeout} s\n" + stdout[-2000:] + "\n" + stderr[-6000:])
    assert p.returncode == 0, stdout[-3000:] + "\n" + stderr[-3000:]
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]            # rank 0 prints ONE line, the other ranks none
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def one_rank():
    return run_bench(1, "--c5-points", "200000", "--no-c4")


def check_line(line, n):
    assert line["n_gpus"] == n and line["steps"] == 3 and line["warmup"] == 1
    assert line["metric"].startswith("M correspondences/sec") and line["unit"] == "M corr/s" and line["value"] > 0
    assert line["scaling"] == "strong" and line["higher_is_better"] is True
    cfg = line["config"]
    assert cfg["sharding"].startswith(f"sources x{n} ")
    # (spatial shards, the default: whole runs of the index-ordered cloud, 64 per rank — the sizes differ by at most one run)
    assert cfg["n_src_total"] == 120000 and abs(cfg["n_src_this_rank"] - 120000 / n) <= 120000 / (64 * n) + 1
    assert cfg["collective"] == ("none" if n == 1 else "torch")
    assert cfg["pose_err_vs_gt_fro"] < 0.5          # three iterations in: moving towards the known pose (identity start: 0.54)
    assert line["roofline"]["frac"] > 0 and line["roofline"]["avg_launch_ms"] > 0


def test_bench_one_rank_line(one_rank):
    check_line(one_rank, 1)
    assert "weak" not in one_rank
    assert one_rank["roofline"]["kernel_family"] == "strack3"
    assert one_rank["roofline"]["survey_8d_convention"]["ratio"] > 0
    assert one_rank["roofline"]["shader_clock"]["mhz"] > 500
    assert one_rank["c5"]["config"]["n_src_total"] == 200000


def test_bench_two_ranks_gloo_rehearsal(one_rank):
    line = run_bench(2, "--c5-points", "200000")
    check_line(line, 2)
    assert line["weak"]["scaling"] == "weak" and line["weak"]["n_src_per_rank"] == 120000 and line["weak"]["value"] > 0
    assert "c4" not in line                         # configs[3] is a one-GPU block
    # the pose does not depend on the number of ranks, bit for bit — brute force (120 k) and exact grid (the c5 block)
    assert line["config"]["pose_bits"] == one_rank["config"]["pose_bits"]
    assert line["exact_grid"]["pose_bit_identical_to_brute_force"] is True
    c5 = line["c5"]
    assert c5["n_gpus"] == 2 and c5["config"]["collective"] == "torch" and c5["config"]["sharding"].startswith("sources x2")
    assert c5["config"]["pose_bits"] == one_rank["c5"]["config"]["pose_bits"]
    assert c5["config"]["kept_pairs_last_iter"] == one_rank["c5"]["config"]["kept_pairs_last_iter"]


def test_bench_four_ranks_gloo_rehearsal_no_c5(one_rank):
    line = run_bench(4, "--no-c5")
    check_line(line, 4)
    assert "c5" not in line and line["weak"]["value"] > 0
    assert line["config"]["pose_bits"] == one_rank["config"]["pose_bits"]
    assert line["config"]["kept_pairs_last_iter"] == one_rank["config"]["kept_pairs_last_iter"]
