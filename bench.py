#!/usr/bin/env python3
"""bench.py — headline benchmark of the k-NN correspondence + ICP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N = 1: runs in this process.  N > 1 without a launcher (WORLD_SIZE unset): starts its own ranks —
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a
        CHILD process before anything touches the GPU, and exits with the child's code (rank 0's JSON line goes to stdout).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W     (the driver's launch)

A *step* is one point-to-point ICP iteration on the BASELINE.json configs[1]/[2] workload: 1-NN correspondence of the source
points against the 120 000-point target (exhaustive / brute force) + Kabsch accumulation + (N > 1: one RCCL all-reduce of
56 + 2N f64) + 3x3 SVD + in-place transform.  W untimed iterations, then exactly K iterations timed between
barrier + torch.cuda.synchronize() on both sides; max over ranks.

value            = correspondences of the whole job per second (M corr/s) on ONE 120 000 x 120 000 pair; with N > 1 ranks the
                   120 000 sources are SHARDED N ways (contiguous blocks, target replicated): "scaling": "strong" — the reading of
                   BASELINE.json.metric ("120k-pt KITTI pair, 1/2/4/8 MI355X").  "weak" (extra key, N > 1) is the other reading:
                   every rank registers its own 120 000-point shard of a denser source scan.
roofline         = the dominant kernel (nn1_btrack_kernel, f16 form): matrix flops the kernel's algorithm needs per launch / average launch
                   duration measured live with HIP events on the kernel's own stream (inside libpcr_hip.so) / 2 500 TFLOP/s; the
                   same launch priced as the f32 filter (6 flop per pair against 157.3 TFLOP/s) under fp32_equivalent.
kernels          = the same for the exact-only kernel (9-op convention of SURVEY.md 8d) and the two HBM streaming kernels.
one_shot         = the cold configs[1] search (no previous correspondences), fresh target and indexed target.
c4 / c5          = BASELINE configs[3] (plane count + radius-NN on the 120k scan) and configs[4] (10 M x 10 M pair, exact grid,
                   sources sharded) measured in the same invocation (--no-c4 / --no-c5 skip them; --workload c4|c5 makes one the
                   headline instead).
cpu_baseline     = the reference's own nanoflann 1-NN (oracle/_ref, kind "reference") or, if that binary is absent, the oracle's
                   scalar brute force (kind "port"), timed on this box's host cores (rank 0, N = 1 only).
Inputs are resident in HBM before the timed region starts; data = synthetic (no dataset ships, no network).
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "hands-on-point-cloud-processing_amd"

VALU_PEAK_TFLOPS = 157.3        # MI355X vector f32, FMA = 2 flop (MI355X_MICROARCH.md)
VALU_PEAK_TOPS_NOFMA = 78.6     # the same issue rate counted one op per lane-slot: the bound of arithmetic without FMA (A1 is unfused)
HBM_PEAK_GBS = 8000.0
OPS_PER_PAIR = 9                # 3 sub + 3 mul + 2 add + 1 compare per (query, target) pair — the exact kernel's work
ETRACK_FLOPS_PER_PAIR = 6       # the f32 filter's algorithm: 3 FMAs per (query, target) pair (csrc/nn1_brute.hip, ETRACK)
STRACK_FLOPS_PER_PAIR = 28      # the sign form of the f16 filter: the same 14 data products as HTRACK (the two K-slots that carry the query's threshold are
                                # bookkeeping, not data: they count under executed_slots — ADVICE r3: keeps roofline.frac comparable with round 2's)
HTRACK_FLOPS_PER_PAIR = 28      # the f16 filter's algorithm: 14 f16 multiply-adds per pair that carry data (3 coordinates x 4 piece products + 2 pieces of
                                # |t''|^2) of the 16 K-slots ONE v_mfma_f32_32x32x16_f16 provides (csrc/nn1_brute.hip, HTRACK)
BTRACK_FLOPS_PER_PAIR = 54      # the bf16 filter's algorithm: 27 bf16 multiply-adds per pair that carry data (3 coordinates x 8 piece products + 3 pieces
                                # of |t''|^2) of the 32 K-slots two v_mfma_f32_32x32x16_bf16 provide (csrc/nn1_brute.hip, BTRACK)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak (MI355X_MICROARCH.md)
METRIC = "M correspondences/sec + ICP iter/sec, 120k-pt KITTI pair, 1/2/4/8 MI355X"


# ------------------------------------------------------------------------------------------------------------ self-launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """--gpus N > 1 without a launcher: one rank per GPU through torch.distributed.run, as a child (never exec: this process
    may not replace itself once anything GPU-related is loaded, and a child keeps the rule trivially true)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode        # stdout / stderr inherited: rank 0's JSON line is the child's line


# ------------------------------------------------------------------------------------------------------------ CPU baseline
def host_threads():
    """threads the CPU baseline may use: the cores this process is allowed on, capped by the cgroup CPU quota when there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:   # noqa: BLE001
        quota = None
    return (min(n, quota) if quota else n), n, quota


def cpu_baseline(src, tgt):
    """Rank 0, N = 1 only.  Bounded: one kd-tree build + 120k queries (~0.1-0.3 s) x 3 repetitions, single- and multi-threaded."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    nthr, affinity, quota = host_threads()
    n = src.shape[1]
    if orc.have_ref():
        best1, bestn = None, None
        for _ in range(3):
            _, _, b, q = orc.ref_nano_nn1_f32(tgt, src, leaf=2, threads=1)
            best1 = (b + q) if best1 is None else min(best1, b + q)
        for _ in range(3):
            _, _, b, q = orc.ref_nano_nn1_f32(tgt, src, leaf=2, threads=nthr)
            bestn = (b + q) if bestn is None else min(bestn, b + q)
        # one CPU ICP iteration = the reference's queries against the tree built once before the loop (registration.cpp:903-934)
        # + the Kabsch accumulation / solve / transform of the oracle restatement (hw9 itself cannot be built here)
        idx, d2, _, q_ms = orc.ref_nano_nn1_f32(tgt, src, leaf=2, threads=1)
        t0 = time.perf_counter()
        sums, _ = orc.kabsch_accumulate(src, tgt, idx, d2, 1.0)
        _, R, t = orc.kabsch_solve(sums)
        orc.transform_f32(src, R, t)
        rest_ms = (time.perf_counter() - t0) * 1e3
        return {"value": n / best1 / 1e3, "unit": "M corr/s", "cores": 1, "kind": "reference",
                "icp_iter_per_s": 1e3 / (q_ms + rest_ms),
                "icp_iter_note": f"queries {q_ms:.1f} ms (reference nanoflann, tree built once per ICP) + Kabsch sums / SVD / transform "
                                 f"{rest_ms:.1f} ms (oracle restatement), 1 thread",
                "sample": f"vendored nanoflann 1.3.2 f32 leaf 2 (ICP's configuration), build + {n} queries, "
                          f"best of 3, 1 thread as the reference runs it",
                "all_cores": {"value": n / bestn / 1e3, "cores": nthr,
                              "note": "same tree (built on one thread), queries split over std::thread; cores = every core this process may "
                                      f"use (affinity {affinity}, cgroup quota {quota})"},
                "host_cpus": os.cpu_count()}
    # port: scalar brute force of the oracle on a bounded sample
    m = 400
    t0 = time.perf_counter()
    orc.nn1_f32(tgt, src[:, :m].copy())
    dt = time.perf_counter() - t0
    return {"value": m / dt / 1e6, "unit": "M corr/s", "cores": 1, "kind": "port",
            "sample": f"oracle scalar brute force, {m} of {n} queries against {tgt.shape[1]} targets",
            "host_cpus": os.cpu_count()}


def lib_sha16(pcr):
    try:
        return hashlib.sha256(open(pcr.LIB_PATH, "rb").read()).hexdigest()[:16]
    except Exception:   # noqa: BLE001
        return None


def load_pmc(name, sha):
    """HBM-side traffic from the separate rocprofv3 --pmc passes of tools/gpu_check.sh (PMC counters cannot be read from inside
    this process).  Only trusted when it was collected with the very library that is loaded now (sha of libpcr_hip.so)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        return d if d.get("lib_sha16") == sha and sha else None
    except Exception:   # noqa: BLE001
        return None


# ------------------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=0, help="points of the pair (target, and source of the whole job); 0 = workload default")
    ap.add_argument("--workload", choices=["c2", "c4", "c5"], default="c2",
                    help="headline of the JSON line.  c2 (default, BASELINE configs[1]/[2]): ONE 120k x 120k pair, sources sharded over the "
                         "GPUs (strong scaling), brute force; c5 (configs[4]): ONE 10M x 10M pair, sources sharded, exact grid; "
                         "c4 (configs[3]): 80-hypothesis plane count + radius-NN r = 1 on the 120k scan (one GPU)")
    ap.add_argument("--collective", choices=["rccl", "torch"], default="rccl")
    ap.add_argument("--shard", choices=["spatial", "contiguous"], default="spatial",
                    help="N > 1: how the sources of the ONE pair are dealt to the ranks.  spatial (default): pcr_cloud_shard_spatial — the cloud in the "
                         "order of the target's index cut into 64 runs per rank, dealt round-robin (every rank keeps the scene's local density); "
                         "contiguous: blocks of the caller's (shuffled) order, pcr_shard_range.  Same pose bits either way (exact sums)")
    ap.add_argument("--no-predict", action="store_true", help="skip the one-GPU shard timings behind predicted_scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the configs[3] block of the default line")
    ap.add_argument("--no-c5", action="store_true", help="skip the configs[4] block of the default line")
    ap.add_argument("--no-extras", action="store_true", help="only the timed loop and its roofline (profiling runs)")
    ap.add_argument("--c5-points", type=int, default=10_000_000)
    ap.add_argument("--tiles-per-slice", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0, help="nn1 kernel variant (0 = library default; 1 FTRACK, 2 exact TRACK, 4 ETRACK, 6 BTRACK, 7 HTRACK)")
    ap.add_argument("--nn", choices=["brute", "grid"], default="brute",
                    help="correspondence search of the c2 headline: brute = BASELINE configs[1] (exhaustive), grid = exact grid index")
    args = ap.parse_args()

    if args.gpus > 1 and not os.environ.get("WORLD_SIZE"):
        sys.exit(self_launch(args.gpus))                # before torch / HIP are even imported

    # PCR_BENCH_WATCHDOG_S=<seconds>: a rank that is still running after that long dumps the Python stack of every thread to stderr and
    # exits (a deadlocked multi-rank run then says WHERE each rank waits instead of hanging until the caller's timeout)
    # Default 1 200 s (the whole default line takes one to two minutes); 0 = off.
    watchdog_s = int(os.environ.get("PCR_BENCH_WATCHDOG_S", "1200"))
    if watchdog_s > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog_s, exit=True)

    import numpy as np
    import torch   # device sync + torch.distributed (RCCL) plumbing; loaded BEFORE libpcr_hip.so so that the
                   # HIP runtime (libamdhip64.so.7) is shared
    pcr = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one rank per GPU (the driver's launch); PCR_BENCH_BACKEND=gloo lets several ranks REHEARSE the multi-rank code
    # path on a box with fewer GPUs (ranks then share devices and the collective runs over gloo / torch.distributed)
    backend = os.environ.get("PCR_BENCH_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    masked = any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"))
    if backend == "nccl" and world > n_dev and not masked:     # a launcher that masks one GPU per rank shows 1 device to each rank: fine
        raise SystemExit(f"{world} ranks but {n_dev} GPU(s): RCCL needs one GPU per rank (PCR_BENCH_BACKEND=gloo rehearses)")
    device_index = local_rank % n_dev
    dist = None
    torch.cuda.set_device(device_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
            if args.collective == "rccl":
                args.collective = "torch"

    ctx = pcr.Context(device_index)
    sha = lib_sha16(pcr)
    tunes_env = {}
    for kv in filter(None, os.environ.get("PCR_TUNE", "").split(",")):     # experiments: PCR_TUNE="key=value,key=value"
        k, v = kv.split("=")
        ctx.tune(k.strip(), int(v))
        tunes_env[k.strip()] = int(v)
    if args.tiles_per_slice:
        ctx.tune("nn1_tiles_per_slice", args.tiles_per_slice)
    if args.variant:
        ctx.tune("nn1_variant", args.variant)

    collective = "none"
    if world > 1:
        collective = args.collective
        if collective == "rccl":
            # native transport: the library's own RCCL communicator (one ncclAllReduce of 56 + 2N f64 per iteration, enqueued
            # on the context stream).  Every step below is collective-safe: rank 0 ALWAYS broadcasts (the id or None),
            # and the ranks agree on the outcome before anybody uses the communicator.
            uid = [None]
            if rank == 0:
                try:
                    uid = [pcr.comm_unique_id()]
                except Exception as e:   # noqa: BLE001
                    print(f"[rank 0] pcr_comm_unique_id failed: {e}", file=sys.stderr)
            dist.broadcast_object_list(uid, src=0)

            def all_agree(ok):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1

            ok = False
            if uid[0] is not None:
                try:
                    ctx.comm_init_rccl(world, rank, uid[0])     # collective rendezvous
                    ok = True
                except Exception as e:   # noqa: BLE001
                    print(f"[rank {rank}] native RCCL communicator failed: {e}", file=sys.stderr)
            if all_agree(ok):
                try:
                    ctx.comm_selftest()                          # one real ncclAllReduce, result checked
                except Exception as e:   # noqa: BLE001
                    ok = False
                    print(f"[rank {rank}] RCCL self-test failed: {e}", file=sys.stderr)
                ok = all_agree(ok)
            else:
                ok = False
            if not ok:
                # same collective over a different (slower) transport — never a different computation
                ctx.comm_destroy()
                collective = "torch"
        if collective == "torch":
            def allreduce(arr):
                if backend == "nccl":
                    t = torch.from_numpy(arr.copy()).cuda()
                    dist.all_reduce(t)
                    arr[:] = t.cpu().numpy()
                else:
                    dist.all_reduce(torch.from_numpy(arr))      # in place on the host buffer
            ctx.comm_init_callback(world, rank, allreduce)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if dist is None:
            return dt
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def shard_of(full_np, ct):
        """this rank's sources of the ONE pair: (device cloud, host copy of it)"""
        if world == 1:
            return ctx.cloud(full_np), full_np
        if args.shard == "contiguous":
            b, e = pcr.shard_range(full_np.shape[1], world, rank)
            part = np.ascontiguousarray(full_np[:, b:e])
            return ctx.cloud(part), part
        full = ctx.cloud(full_np)
        sh = ctx.shard_spatial(ct, full, world, rank)
        full.free()
        return sh, sh.numpy()

    def predicted_scaling(full_np, ct, method, iters, T_final, counts=(2, 4, 8)):
        """What ONE rank of an N-rank strong-scaling run has to do, timed on THIS GPU for every rank's shard (no collective: the message is
        <= 832 B, latency-bound): ms per iteration of an `iters`-iteration ICP from the start pose (wall, no event pairs), and the steady
        search at the final pose (HIP events).  predicted efficiency = T_1 / (N x the slowest rank's T) — the figure the driver's 8-GPU run
        is to be held against; the all-reduce (~25 us per iteration) is NOT in it."""
        def measure(cloud):
            ctx.tune("nn_method", method); ctx.tune("prof", 0)
            ctx.icp_point2point(cloud, ct, max_corr=1.0, max_iter=2, eps=0.0)
            best = None
            for _ in range(3):
                ctx.sync(); t0 = time.perf_counter()
                ctx.icp_point2point(cloud, ct, max_corr=1.0, max_iter=iters, eps=0.0)
                dt = (time.perf_counter() - t0) * 1e3 / iters
                best = dt if best is None else min(best, dt)
            ctx.tune("prof", 1); ctx.prof_reset()
            ctx.icp_point2point(cloud, ct, init_T=T_final, max_corr=1.0, max_iter=8, eps=0.0)
            each = np.concatenate([ctx.prof_get_each("nn1_brute"), ctx.prof_get_each("nn1_grid")])
            ctx.tune("prof", 0)
            return best, float(each[2:].mean()) if each.size > 2 else float("nan")
        full = ctx.cloud(full_np)
        t1, s1 = measure(full)
        out = {"iterations": iters, "one_rank": {"ms_per_iteration": t1, "steady_search_ms": s1}, "sharding": "spatial (pcr_cloud_shard_spatial, 64 runs per rank)",
               "note": "every rank's shard of the ONE pair run on this one GPU, one after the other; efficiency = T_1 / (N x slowest rank); the "
                       "per-iteration all-reduce (56 + 2N f64, latency-bound) is not included; no multi-GPU hardware was involved"}
        for N in counts:
            per, srch, sizes = [], [], []
            for r in range(N):
                sh = ctx.shard_spatial(ct, full, N, r)
                a, b = measure(sh)
                per.append(a); srch.append(b); sizes.append(len(sh))
                sh.free()
            out[str(N)] = {"ms_per_iteration_slowest_rank": max(per), "ms_per_iteration_by_rank": per, "efficiency": t1 / (N * max(per)),
                           "steady_search_ms_slowest_rank": max(srch), "steady_search_efficiency": s1 / (N * max(srch)), "points_by_rank": sizes}
        full.free()
        return out

    def prepare(cs, ct, method, src_np):
        """one-time preparation, whatever --warmup says: the index over the (replicated) target and the code objects of every kernel
        of the loop — a 2-iteration ICP of a small slice of this rank's sources; initialisation, not a step"""
        ctx.tune("nn_method", method)
        n_prep = min(4096, src_np.shape[1])
        c_prep = ctx.cloud(np.ascontiguousarray(src_np[:, :n_prep]))
        ctx.icp_point2point(c_prep, ct, max_corr=1.0, max_iter=2, eps=0.0)
        c_prep.free()

    def timed_icp(cs, ct, method, src_np, prof=1, steps=None, warmup=None):
        """W untimed + exactly K timed ICP iterations with the given correspondence search; max over ranks.
        prof = 1: every correspondence launch of the timed region is bracketed by a HIP event pair on the library's stream
        (the kernel's average duration for the roofline); the pairs cost ~13 us of stream time per iteration."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        ctx.tune("prof", prof)
        prepare(cs, ct, method, src_np)
        if warmup > 0:
            ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=warmup, eps=0.0)
        ctx.prof_reset()
        barrier()
        t0 = time.perf_counter()
        T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=steps, eps=0.0)   # eps = 0: never early-exits
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        assert st["iters_run"] == steps, st
        return T, st, dt

    def kernel_breakdown(cs, ct, method, steps):
        """a second, fully profiled run of the same loop (an event pair around EVERY kernel): per-kernel average durations"""
        ctx.tune("nn_method", method)
        ctx.tune("prof", 2)
        ctx.prof_reset()
        ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=steps, eps=0.0)
        out = {}
        for name in ("nn1_brute", "nn1_grid", "kabsch_partial", "kabsch_final", "icp_update", "transform"):
            k, ms = ctx.prof_get(name)
            if k:
                out[name] = (int(k), ms / k)
        ctx.tune("prof", 0)
        return out

    def stream_kernels(bd, n_q, kept):
        """the two HBM streaming kernels of an iteration against the 8 TB/s peak (28 B per kept pair, 24 B per point)"""
        out = {}
        if "kabsch_partial" in bd:
            k, ms = bd["kabsch_partial"]
            b = 28.0 * kept
            out["kabsch_partial"] = {"bytes": b, "ms": ms, "GB/s": b / ms / 1e6, "frac_of_8TB/s": b / ms / 1e6 / HBM_PEAK_GBS, "launches": k,
                                     "algorithmic": "28 B per kept pair: 12 B source + 8 B key + 12 B gathered target - 4 (the key carries the index)"}
        if "transform" in bd:
            k, ms = bd["transform"]
            b = 24.0 * n_q
            out["transform_state"] = {"bytes": b, "ms": ms, "GB/s": b / ms / 1e6, "frac_of_8TB/s": b / ms / 1e6 / HBM_PEAK_GBS, "launches": k,
                                      "algorithmic": "24 B per point: 12 B read + 12 B written in place"}
        for name in ("kabsch_final", "icp_update"):
            if name in bd:
                out[name] = {"ms": bd[name][1], "launches": bd[name][0], "note": "one workgroup: block partials -> 16 moments -> 3x3 SVD (latency-bound)"}
        return out

    out = None
    # ==================================================================================================== c2 headline (+ extras)
    if args.workload == "c2":
        n = args.points or 120000
        full_src, tgt = synth.kitti_like_pair(n)
        ct = ctx.cloud(tgt)
        ctx.tune("nn_method", 2)                                  # (the shard plan orders the sources by the target's cell index, whatever search follows)
        cs, src = shard_of(full_src, ct)
        ctx.tune("nn_method", 0)
        main_method = 1 if args.nn == "brute" else 2
        # the K timed steps run WITHOUT event pairs (a pair around every search costs ~10 us of stream time per step — 15 % of a 70 us iteration);
        # the same K steps are then repeated with the pairs for the kernel's duration (roofline): same pose bits, `ms_per_step_with_event_pairs` beside it
        T, st, dt = timed_icp(cs, ct, main_method, src, prof=0)
        Tp, stp, dt_pairs = timed_icp(cs, ct, main_method, src, prof=1)
        assert np.array_equal(T.view(np.uint32), Tp.view(np.uint32)) and stp["iters_run"] == st["iters_run"]
        nn_name = "nn1_brute" if args.nn == "brute" else "nn1_grid"
        nn_launches, nn_ms = ctx.prof_get(nn_name)
        family = ctx.mfma_check()["last_nn1_kernel"]        # which kernel family the library's dispatcher took for the timed searches
        kern_s = nn_ms / 1e3 / max(nn_launches, 1)
        n_q, n_t = src.shape[1], tgt.shape[1]
        pairs = float(n_q) * float(n_t)
        default_kernels = not args.variant
        extras = not args.no_extras
        bd = kernel_breakdown(cs, ct, main_method, min(args.steps, 10)) if extras else {}

        # ---- the shader clock the chip holds under the headline kernel's own load: s_memtime / s_memrealtime stamps of a diagnostics
        # launch (tune grid_stats) that follows 40 back-to-back launches of the same seeded search; median of 3
        clock_mhz = None
        if extras and args.nn == "brute" and family in ("strack3", "strack", "htrack", "btrack"):
            ca = cs.clone(); ctx.transform(ca, T)
            ctx.tune("nn_method", 1); ctx.tune("prof", 0); ctx.tune("nn1_async_in_loop", 1)
            clocks = []
            for _ in range(3):
                for _ in range(40):
                    ctx.nn1_async(ct, ca)
                ctx.tune("grid_stats", 1); ctx.nn1_async(ct, ca); ctx.tune("grid_stats", 0)
                w = ctx.nn1_stats()
                if w[5]:
                    clocks.append(w[4] / w[5] * 100.0)
            ctx.tune("nn1_async_in_loop", 0)
            ca.free()
            clock_mhz = sorted(clocks)[len(clocks) // 2] if clocks else None

        # ---- STRACK3 (the sign filter over three levels of bounding spheres): how many matrix instructions a search at the final pose executes — the last
        # search of a 9-iteration loop from that pose (the loop sorts its working cloud along the target's order, as the timed loop does), diagnostics launch
        s2_counts = None
        if args.nn == "brute" and family == "strack3":
            ctx.tune("nn_method", 1); ctx.tune("prof", 0)
            ctx.tune("grid_stats", 1)
            ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=9, eps=0.0)
            ctx.tune("grid_stats", 0)
            w = ctx.nn1_stats()
            s2_counts = {"level0_mfma": int(w[7]), "level1_tiles_flagged": int(w[3]),
                         "level1_mfma": int(w[8]), "level2_mfma": int(w[10]), "level2_tiles_flagged": int(w[9]), "chunks_evaluated_exactly": int(w[6])}

        # ---- the exact-only kernel (the 9-op convention's own kernel), the cold searches: a few launches each, HIP-event timed
        exact_line, one_shot = None, None
        if extras and args.nn == "brute":
            def time_search(tgt_cloud, reps, **tunes):
                for k, v in tunes.items():
                    ctx.tune(k, v)
                ctx.tune("nn_method", 1); ctx.tune("prof", 1)
                ctx.nn1_async(tgt_cloud, cs); ctx.sync(); ctx.prof_reset()
                for _ in range(reps):
                    ctx.nn1_async(tgt_cloud, cs)
                k, ms = ctx.prof_get("nn1_brute")
                for key in tunes:
                    ctx.tune(key, 0)
                return ms / max(k, 1)
            ms_exact = time_search(ct, 5, nn1_variant=2)
            tf = OPS_PER_PAIR * pairs / (ms_exact / 1e3) / 1e12
            exact_line = {"kernel": "pcr::nn1_track_kernel<2, 16> (nn1_variant 2: the exact unfused arithmetic for every pair, no filter)",
                          "bound": "valu", "achieved": tf, "peak": VALU_PEAK_TOPS_NOFMA, "unit": "T lane-ops/s", "frac": tf / VALU_PEAK_TOPS_NOFMA,
                          "avg_launch_ms": ms_exact, "kernel_M_corr_per_s": n_q / ms_exact / 1e3,
                          "algorithmic": f"{OPS_PER_PAIR} f32 lane-ops per (query, target) pair (SURVEY.md 8d: 3 sub + 3 mul + 2 add + 1 compare) x {pairs:.3e} "
                                         "pairs; peak = 157.3 TF/s / 2 (the exact arithmetic has no FMA: one op per lane per issue slot)"}
            # a target nobody has searched before: what a first, one-shot call sees — WALL time of the call (launch to completion, the
            # index the default kernel builds on a target's first search included), a new cloud for every timed call
            ctx.tune("nn_method", 1); ctx.tune("prof", 1)
            warm_cloud = ctx.cloud(tgt); ctx.nn1_async(warm_cloud, cs); ctx.sync(); warm_cloud.free(); ctx.prof_reset()
            walls = []
            for _ in range(7):
                fresh = ctx.cloud(tgt); ctx.sync()
                t0 = time.perf_counter(); ctx.nn1_async(fresh, cs); ctx.sync(); walls.append((time.perf_counter() - t0) * 1e3)
                fresh.free()
            k_fresh, ms_fresh_total = ctx.prof_get("nn1_brute")
            ms_fresh = sorted(walls)[len(walls) // 2]
            ms_fresh_kernel = ms_fresh_total / max(k_fresh, 1)
            ctx.tune("nn1_bf16", 2)                          # the same first call with the f32 kernels (FTRACK: no index at all)
            walls_f32 = []
            for _ in range(7):
                fresh = ctx.cloud(tgt); ctx.sync()
                t0 = time.perf_counter(); ctx.nn1_async(fresh, cs); ctx.sync(); walls_f32.append((time.perf_counter() - t0) * 1e3)
                fresh.free()
            ctx.tune("nn1_bf16", 0)
            ms_fresh_f32 = sorted(walls_f32)[len(walls_f32) // 2]
            ctx.tune("nn_method", 0)                         # the same first call as the library itself dispatches it (nn1_auto_grid)
            walls_auto = []
            for _ in range(7):
                fresh = ctx.cloud(tgt); ctx.sync()
                t0 = time.perf_counter(); ctx.nn1_async(fresh, cs); ctx.sync(); walls_auto.append((time.perf_counter() - t0) * 1e3)
                fresh.free()
            ctx.tune("nn_method", 1)
            ms_fresh_auto = sorted(walls_auto)[len(walls_auto) // 2]
            ms_indexed = time_search(ct, 5)
            cold_family = ctx.mfma_check()["last_nn1_kernel"].upper()      # what the dispatcher took for the cold, self-seeded search
            one_shot = {"note": "BASELINE configs[1]: ONE 1-NN search of the pair with no earlier correspondences (no seed)",
                        "fresh_target": {"ms": ms_fresh, "M_corr_per_s": n_q / ms_fresh / 1e3, "kernel_ms": ms_fresh_kernel,
                                         "kernel": "the FIRST search of a target cloud, wall time of the call (median of 7, host launch + completion wait "
                                                   "included): the Morton-ordered bf16 operands are built (one bounding-box round trip, radix sort) and "
                                                   f"the default kernel ({cold_family}, behind the cold seed of bt_seed_kernel) runs",
                                         "f32_kernels_ms": ms_fresh_f32,
                                         "f32_kernels": "the same call with nn1_bf16 = 2: pcr::nn1_ftrack_kernel<2, 16>, which needs no index",
                                         "auto_dispatch_ms": ms_fresh_auto,
                                         "auto_dispatch": "the same call with nn_method 0, the library's own choice: at this size (queries x targets > 2e9) "
                                                          "the exact grid — index build + search, same answers; NOT the brute-force workload of BASELINE configs[1]"},
                        "indexed_target": {"ms": ms_indexed, "M_corr_per_s": n_q / ms_indexed / 1e3,
                                           "kernel": f"the default kernel ({cold_family}), cold — no earlier correspondences; the search seeds itself from the nearest super-tile (bt_seed_kernel, inside the timed scope) — kernel time (the target's operands exist: any earlier search built them)"}}
            ms_f32 = time_search(ct, 5, nn1_bf16=2)
            ms_bf16 = time_search(ct, 5, nn1_f16=2)
            one_shot["indexed_target_bf16_filter"] = {"ms": ms_bf16, "M_corr_per_s": n_q / ms_bf16 / 1e3,
                                                      "kernel": "BTRACK (tune nn1_f16 = 2: the filter as two bf16 MFMAs per tile; what a cloud outside f16's range gets), cold"}
            one_shot["indexed_target_f32_filter"] = {"ms": ms_f32, "M_corr_per_s": n_q / ms_f32 / 1e3,
                                                     "kernel": "pcr::nn1_etrack_kernel<4> (tune nn1_bf16 = 2: the same filter as 3 vector FMAs per pair), cold and unseeded"}

        # ---- second, separately timed pass with the exact grid index (same answers, different search): extra info only
        grid_extra = None
        if extras and args.nn == "brute":
            Tg, stg, dtg = timed_icp(cs, ct, 2, src, prof=0)       # throughput without the event pairs (11 % of this much shorter step) ...
            ctx.tune("prof", 1); ctx.prof_reset()                   # ... and the kernel time from a second, profiled run of the same loop
            ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=args.steps, eps=0.0)
            gl, gms = ctx.prof_get("nn1_grid")
            grid_extra = {"value": n * args.steps / dtg / 1e6, "unit": "M corr/s", "icp_iter_per_s": args.steps / dtg,
                          "ms_per_step": dtg * 1e3 / args.steps, "avg_nn_kernel_ms": gms / max(gl, 1),
                          "pose_bit_identical_to_brute_force": bool(np.array_equal(T.view(np.uint32), Tg.view(np.uint32))),
                          "note": "same ICP with pcr nn_method = grid (exact uniform-grid index, csrc/grid.hip); NOT the "
                                  "BASELINE configs[1] workload, reported for information"}

        # ---- weak scaling (N > 1): every rank its own 120k-point shard of a denser source scan
        weak = None
        if extras and world > 1:
            wsrc, _ = synth.kitti_like_pair(n, n_src=n, shard=rank)
            cw = ctx.cloud(wsrc)
            Tw, stw, dtw = timed_icp(cw, ct, main_method, wsrc, prof=0)
            weak = {"value": world * n * args.steps / dtw / 1e6, "unit": "M corr/s", "icp_iter_per_s": args.steps / dtw, "ms_per_step": dtw * 1e3 / args.steps,
                    "scaling": "weak", "n_src_per_rank": n, "note": "every rank owns a different 120k-point shard of the source scan, target replicated"}
            cw.free()

        # what the 1 / 2 / 4 / 8 reading of the metric should show: every rank's shard timed on this one GPU (world == 1 only: no collective inside)
        pred_c2 = predicted_scaling(full_src, ct, main_method, args.steps, T) if (extras and world == 1 and not args.no_predict) else None
        # (grid_roofline runs ICP loops — collectives under world > 1 — so EVERY rank calls it, not only the one that prints)
        groof_c2 = grid_roofline(ctx, pcr, np, cs, ct, T, n_q, n_t, kern_s, sha) if args.nn == "grid" else None
        if rank == 0:
            gt_err = float(np.linalg.norm(T.astype(np.float64) - synth.gt_pose()))
            if args.nn == "brute":
                two = three = family == "strack3"
                sign = family in ("strack", "strack3")
                f16 = family in ("strack3", "strack", "htrack")
                bf16 = family in ("strack3", "strack", "htrack", "btrack")
                slots_pp = 32 if f16 else 64                 # flop per pair of ALL K-slots the matrix instruction(s) execute (16 / 2 x 16 multiply-adds)
                flops_pp = STRACK_FLOPS_PER_PAIR if sign else HTRACK_FLOPS_PER_PAIR if f16 else BTRACK_FLOPS_PER_PAIR if bf16 else ETRACK_FLOPS_PER_PAIR
                peak_tf = MFMA_BF16_PEAK_TFLOPS if bf16 else VALU_PEAK_TFLOPS
                achieved_tflops = flops_pp * pairs / kern_s / 1e12
                if two and s2_counts:
                    # the two-level form does NOT run every (query, record) pair through the matrix pipe: its algorithmic work is what it executes —
                    # 14 data-carrying K-slots x 32 x 32 x 2 flop per MFMA of either level (counted by a diagnostics launch at the final pose)
                    achieved_tflops = (s2_counts["level0_mfma"] + s2_counts["level1_mfma"] + s2_counts["level2_mfma"]) * 32 * 32 * 14 * 2 / kern_s / 1e12
                compulsory_bytes = 12.0 * n_t + 12.0 * n_q + 8.0 * n_q        # targets + sources + (idx, d2) key
                pmc = load_pmc("latest_pmc.json", sha) if (default_kernels and n_q == 120000 == n_t) else None
                roofline = {
                    "bound": "mfma" if bf16 else "valu", "achieved": achieved_tflops, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": achieved_tflops / peak_tf,
                    "traffic": (pmc["fetch_bytes_per_launch_corrected_x2"] + pmc["write_bytes_per_launch"]) if pmc else None,
                    "traffic_note": (f"HBM-side bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) from separate rocprofv3 --pmc passes of this "
                                     f"kernel with this very libpcr_hip.so (sha {sha}), {pmc['source']}") if pmc else
                                    "null: PMC counters need a rocprofv3 wrapper (tools/gpu_check.sh); none was collected with the library loaded now",
                    "kernel": (("pcr::nn1_strack3_kernel<1> = STRACK3, the SIGN form of the f16 matrix-core filter at THREE LEVELS (csrc/nn1_sphere.hpp), one wave per 32 "
                               "queries: level 0 — one MFMA row per level-1 TILE of 512 records (its bounding sphere, in the scale of a level-0 super-tile of 131 072 "
                               "records: |r - c| <= sqrt(thr) + rho as a sum of 16 K-slot products whose sign answers), over EVERY row of the target; level 1 — the chunk "
                               "rows (16 records each) of the level-1 tiles level 0 flagged; level 2 — the per-record rows of the tiles of 32 records that hold a flagged "
                               "chunk; flagged (query, 16-record chunk) pairs evaluated with the exact unfused arithmetic, four lanes per chunk; thresholds = the exact "
                               "distance of the best candidate so far (the previous correspondence re-evaluated by the move, then whatever the scan finds).  Exhaustive "
                               "in the brute-force contract's sense: no record is skipped without a computed sign that says it cannot matter")
                               if two else (("pcr::nn1_strack_kernel<4> = STRACK, the SIGN form of the f16 matrix-core filter, operands staged through LDS per workgroup "
                                "(exhaustive: the expanded-form lower bound of ALL (query, target) pairs from ONE v_mfma_f32_32x32x16_f16 per 32 queries "
                                "x 32 targets — operands scaled per 256-target super-tile and cut into two f16 pieces, every piece product exact in f32 — "
                                "with the query's threshold (the exact distance of its best candidate so far: the previous correspondence re-evaluated, "
                                "then whatever the scan finds) folded into the two remaining K-slots, so that an accumulator is bound - threshold and "
                                "its SIGN says whether the record can matter; the vector ALU ORs the 64 accumulators a lane holds for a tile's four query groups in "
                                "one chain (32 v_or3_b32 — a half-rate instruction like v_min3: what the sign form saves is the minimum tracking) and the "
                                "wave tests one word per tile; flagged (query, 16-record chunk) pairs go to a wave-private LDS list and are evaluated "
                                "together with the exact unfused arithmetic, 16 lanes per chunk, the thresholds fall before the next super-tile's operands "
                                "are built; one operand setup per query and super-tile, the halves exchanged by v_permlane32_swap)") if sign else
                               ("pcr::nn1_btrack_kernel<4, true> = HTRACK, operands staged through LDS per workgroup (exhaustive scan of every (query, 16-target chunk): the expanded-form lower "
                                "bound of ALL pairs on the f16 matrix pipe — operands scaled per 256-target super-tile and cut into two f16 pieces each, "
                                "every piece product exact in f32, ONE v_mfma_f32_32x32x16_f16 per 32 queries x 32 targets; the vector ALU takes the "
                                "minimum of the 16 accumulators per lane (8 v_min3) and tracks first / second minimum branch-free; only the winning "
                                "chunk is evaluated with the exact unfused arithmetic — a query it does not settle gets its slice filtered again and "
                                "every chunk at or below its threshold evaluated; the previous correspondence of each query, re-evaluated "
                                "exactly, seeds the bound)") if f16 else
                               ("pcr::nn1_btrack_kernel<4, false> = BTRACK (exhaustive scan of every (query, 16-target chunk): the expanded-form lower bound of ALL "
                                "pairs on the bf16 matrix pipe — f32 operands cut into three bf16 pieces each, every piece product exact in f32, "
                                "two v_mfma_f32_32x32x16_bf16 per 32 queries x 32 targets; the vector ALU takes the minimum of the 16 accumulators "
                                "per lane (8 v_min3) and tracks first / second minimum branch-free; only the winning chunk is evaluated with the "
                                "exact unfused arithmetic; the previous correspondence of each query, re-evaluated exactly, seeds the bound)") if bf16 else
                               ("pcr::nn1_etrack_kernel<4> (exhaustive scan of every (query, 16-target chunk); chunk-centred targets broadcast through the "
                                "scalar cache; expanded-form lower bound = 3 FMAs per pair (v_pk_fma_f32), min-tree + first/second minimum tracked "
                                "branch-free; only the winning chunk is evaluated with the exact unfused arithmetic; the previous correspondence of "
                                "each query, re-evaluated exactly, seeds the bound)"))) if default_kernels else f"nn1 variant={args.variant}",
                    "launches": int(nn_launches), "avg_launch_ms": kern_s * 1e3, "kernel_M_corr_per_s": n_q / kern_s / 1e6,
                    "algorithmic": (((f"the matrix work the hierarchical filter EXECUTES at the final pose: {s2_counts['level0_mfma']} level-0 + {s2_counts['level1_mfma']} level-1 + "
                                     f"{s2_counts['level2_mfma']} level-2 v_mfma_f32_32x32x16_f16 per launch x 32 x 32 x 14 data-carrying K-slots x 2 flop (a diagnostics launch "
                                     f"counts them; every {'level-1 tile' if three else 'chunk'} of the target gets its row for every query: {pairs / (512 if three else 16):.3e} "
                                     f"(query, row) pairs) — NOT {pairs:.3e} pairs x 28 flop: the finer rows run only where a sphere reaches the query's ball.  The launch has "
                                     "left both roofs: it executes a few GFLOP and moves a few MB (hbm_view), and its time is the chain of dependent steps of ONE wave — "
                                     "operands -> level 0 -> listed level-1 tiles -> listed level-2 tiles -> exact evaluation -> thresholds, each a trip to memory (DESIGN.md 5: "
                                     "wave lives 10-35 us, all 3 750 waves resident at once).  pairs_per_s prices the same launch in (query, target) pairs settled per second") if (two and s2_counts) else
                                    (f"{flops_pp} f16 flop per (query, target) pair (the 14 piece products that carry data; the two K-slots with the "
                                     f"pieces of the query's threshold count under executed_slots) x {pairs:.3e} pairs per launch; peak = 2 500 TF/s dense f16.  Per 1024 pairs: one MFMA "
                                     "(32 cycles of the SIMD's matrix pipe) and 13 vector instructions all told (PMC SQ_INSTS_VALU / SQ_INSTS_MFMA: 8 v_or3_b32 in the tile "
                                     "loop, 3 of the operand setup per super-tile, the rest prologue / lists); the two pipes share the SIMD's issue port, so the "
                                     "launch is issue-bound, not matrix-bound (DESIGN.md 5, profiles/r03_ubench_sign_filter.txt)"))
                                    if sign else
                                    f"{flops_pp} {'f16' if f16 else 'bf16'} flop per (query, target) pair ({flops_pp // 2} piece products that carry data, of "
                                    f"the {16 if f16 else 32} K-slots executed) x {pairs:.3e} pairs per launch; peak = 2 500 TF/s dense f16 / bf16.  On this chip "
                                    "the matrix instructions and the vector instructions of one SIMD's waves take turns in this loop (measured: "
                                    "tools/ubench/mfma_filter.hip, profiles/r02_mfma_filter_experiments.txt), so the launch time is MFMA time "
                                    f"({1 if f16 else 2} x 32 cycles per 1024 pairs) PLUS vector time (about 20 instructions per 1024 pairs, the larger "
                                    "share): fp32_equivalent prices the same launch in the f32 filter's 6 flop per pair against the 157.3 TF/s vector "
                                    "peak, kernels.nn1_exact_track is SURVEY.md 8d's 9-op convention")
                                   if bf16 else
                                   (f"{ETRACK_FLOPS_PER_PAIR} flop (3 FMAs) per (query, target) pair x {pairs:.3e} pairs per launch — the arithmetic of the "
                                    "kernel that ran; the min-tree, the per-chunk prologue (|q - C|^2, 11 ops per 16 targets) and the exact "
                                    "re-evaluation of the winning chunk are overhead, not numerator.  peak = 157.3 TF/s vector f32 (FMA = 2).  "
                                    "SURVEY.md 8d's 9-op-per-pair convention describes the EXACT kernel: kernels.nn1_exact_track"),
                    "kernel_family": family,
                    "two_level": ({"counts_at_the_final_pose": s2_counts, "pairs_per_s": pairs / kern_s,
                                   "equivalent_strack_frac": STRACK_FLOPS_PER_PAIR * pairs / kern_s / 1e12 / peak_tf,
                                   "note": "equivalent_strack_frac = what roofline.frac would read if every (query, target) pair had gone through the per-record "
                                           "filter (28 flop) in this launch's time — an equivalent for comparison with earlier rounds (0.33-0.38), not a bound"}
                                  if (two and s2_counts) else None),
                    "issue_view": ({k: pmc.get(k) for k in ("valu_insts_per_launch", "salu_insts_per_launch", "mfma_insts_per_launch", "waves_per_launch", "valu_busy_frac",
                                                              "mfma_busy_frac", "wave_life_us")} | {
                                       "note": "from the same --pmc passes as `traffic` (mean over the launches of a 9-iteration loop, its cold first search included): what the launch is "
                                               "made of — instructions issued, the share of the launch the vector / matrix pipes were busy, a wave's mean life"}) if pmc else None,
                    "hbm_view": {"algorithmic_bytes": compulsory_bytes, "achieved": compulsory_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": compulsory_bytes / kern_s / 1e9 / HBM_PEAK_GBS,
                                 "note": "targets + sources + keys once per launch against the 8 TB/s roof: the search is nowhere near it either"},
                    "survey_8d_convention": {"achieved": OPS_PER_PAIR * pairs / kern_s / 1e12, "peak": VALU_PEAK_TOPS_NOFMA, "unit": "T lane-ops/s",
                                             "ratio": OPS_PER_PAIR * pairs / kern_s / 1e12 / VALU_PEAK_TOPS_NOFMA,
                                             "note": "SURVEY.md 8d's own figure: 9 f32 lane-ops per (query, target) pair / 78.6 T lane-ops/s.  It "
                                                     "exceeds 1 for the matrix-core kernels because the products run on the matrix pipe (every pair "
                                                     "is still evaluated: SQ_INSTS_MFMA = pairs / 1024 per MFMA of the form) — a ratio, not a bound; "
                                                     "the kernel this convention describes is kernels.nn1_exact_track"},
                    "executed_slots": ({"flop_per_pair": slots_pp, "achieved": slots_pp * pairs / kern_s / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": slots_pp * pairs / kern_s / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                        "note": "every K-slot the matrix instructions execute (16 multiply-adds per pair and MFMA), data-carrying or not"}
                                       if bf16 else None),
                    "shader_clock": ({"mhz": clock_mhz, "nominal_mhz": 2400.0,
                                      "peak_at_measured_clock": peak_tf * clock_mhz / 2400.0,
                                      "frac_at_measured_clock": achieved_tflops / (peak_tf * clock_mhz / 2400.0),
                                      "note": "sum of s_memtime cycles / sum of s_memrealtime ticks (100 MHz) over the workgroups of a diagnostics "
                                              "launch that follows 40 back-to-back launches of the same seeded search (median of 3): the peak the "
                                              "chip could reach at the clock it actually holds under this kernel"} if clock_mhz else None),
                    "reproduce": (f"achieved = {flops_pp} x {n_q} x {n_t} / (avg_launch_ms / 1e3) / 1e12; frac = achieved / {peak_tf:g}; avg_launch_ms = HIP "
                                  "events around every search of the timed region (same kernel's average in profiles/*_bench_rocprof_summary.md); "
                                  "traffic = 2 x FETCH_SIZE + WRITE_SIZE per launch of the --pmc passes stamped with lib_sha16"),
                    "fp32_equivalent": {"achieved": ETRACK_FLOPS_PER_PAIR * pairs / kern_s / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "ratio": ETRACK_FLOPS_PER_PAIR * pairs / kern_s / 1e12 / VALU_PEAK_TFLOPS,
                                        "note": "the f32 filter's 3 FMAs per pair over this launch's time, against the vector f32 peak — an "
                                                "EQUIVALENT, not a bound (for the f16 / bf16 kernels the matrix cores do these multiplications: "
                                                "the ratio may pass 1); the roofline fraction of the kernel that ran is roofline.frac"},
                    "issue": ({"executed_lane_ops_per_pair": pmc["valu_insts_per_launch"] * 64 / pairs,
                               "issue_frac": pmc["valu_insts_per_launch"] * 64 / kern_s / 1e12 / VALU_PEAK_TOPS_NOFMA,
                               "note": "SQ_INSTS_VALU x 64 lanes / time / 78.6 T issue slots per second (same PMC passes)"} if pmc else None),
                    "hbm_literal": {"bound": "hbm", "achieved": compulsory_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": compulsory_bytes / kern_s / 1e9 / HBM_PEAK_GBS,
                                    "note": "compulsory bytes (32 B/point); brute force is VALU-bound, see DESIGN.md"}}
                workload = (f"point-to-point ICP iteration = exhaustive 1-NN correspondence of ONE {n} x {n_t} pair + Kabsch + transform; BASELINE.json "
                            "configs[1]/[2] ('LDS-tiled brute force': the default kernel stages the target's matrix-core operands through LDS, one "
                            "256-target super-tile per workgroup and barrier)")
            else:
                roofline, workload = groof_c2, \
                    f"point-to-point ICP iteration on ONE {n_t} x {n} pair = exact grid 1-NN + Kabsch + transform"
            out = {
                "metric": METRIC, "value": n * args.steps / dt / 1e6, "unit": "M corr/s", "icp_iter_per_s": args.steps / dt,
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": workload, "nn": args.nn, "n_src_this_rank": n_q, "n_src_total": n, "n_tgt": n_t, "max_corr": 1.0,
                           "sharding": f"sources x{world} ({'spatial: pcr_cloud_shard_spatial, 64 runs of the index-ordered cloud per rank dealt round-robin' if args.shard == 'spatial' else 'contiguous blocks of the ONE pair: pcr_shard_range'}), target replicated",
                           "collective": collective,
                           "pose_err_vs_gt_fro": gt_err, "kept_pairs_last_iter": int(st["last_pairs"]), "lib_sha16": sha,
                           "first_timed_iteration": "seeded by the warm-up run's correspondences (as every iteration after the first of an ICP is)",
                           "timing": "the K timed steps carry no HIP-event pairs; an identical repetition with a pair around every search gives the kernel durations (roofline)",
                           "M_corr_per_s_three_readings": ({"warm_icp_iteration (= value)": n * args.steps / dt / 1e6,
                                                            "one_shot_indexed_target (kernel, cold)": one_shot["indexed_target"]["M_corr_per_s"],
                                                            "one_shot_fresh_target (wall, index build included)": one_shot["fresh_target"]["M_corr_per_s"]}
                                                           if (one_shot and world == 1) else None),
                           "pose_bits": "".join(f"{int(v):08x}" for v in np.ascontiguousarray(T, np.float32).view(np.uint32).ravel())},
                "roofline": roofline,
                "kernels": dict(({"nn1_exact_track": exact_line} if exact_line else {}), **stream_kernels(bd, n_q, int(st["last_pairs"]) * n_q // max(n, 1))),
            }
            # the fixed tail of an iteration (sums + solve + move + the bench's own event pairs): what caps strong scaling on a small pair
            out["tail_us_per_iteration"] = (dt * 1e3 / args.steps - kern_s * 1e3) * 1e3
            out["ms_per_step_with_event_pairs"] = dt_pairs * 1e3 / args.steps        # (the repetition the kernel durations of `roofline` come from)
            if pred_c2 is not None:
                out["predicted_scaling"] = pred_c2
            if one_shot:
                out["one_shot"] = one_shot
            if grid_extra is not None:
                out["exact_grid"] = grid_extra
            if weak is not None:
                out["weak"] = weak
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(src, tgt)
        cs.free(); ct.free()
        del full_src

    # ==================================================================================================== c4: configs[3]
    if (args.workload == "c4" or (args.workload == "c2" and not args.no_c4 and not args.no_extras)) and world == 1:
        c4 = bench_c4(ctx, pcr, synth, np, args)
        if args.workload == "c4":
            out = c4
        elif out is not None:
            out["c4"] = c4

    # ==================================================================================================== c5: configs[4]
    if args.workload == "c5" or (args.workload == "c2" and not args.no_c5 and not args.no_extras):
        n5 = (args.points or 10_000_000) if args.workload == "c5" else args.c5_points
        full_src, tgt = synth.kitti_like_pair(n5)
        ct = ctx.cloud(tgt)
        ctx.tune("nn_method", 2)
        cs, src = shard_of(full_src, ct)
        steps5 = args.steps if args.workload == "c5" else min(args.steps, 10)
        warm5 = args.warmup if args.workload == "c5" else min(args.warmup, 2)
        T, st, dt = timed_icp(cs, ct, 2, src, prof=1, steps=steps5, warmup=warm5)
        gl, gms = ctx.prof_get("nn1_grid")
        kern_s = gms / 1e3 / max(gl, 1)
        bd = kernel_breakdown(cs, ct, 2, min(steps5, 5))
        n_q, n_t = src.shape[1], tgt.shape[1]
        groof = grid_roofline(ctx, pcr, np, cs, ct, T, n_q, n_t, kern_s, sha)      # (every rank: its ICP loops are collective)
        pred_c5 = predicted_scaling(full_src, ct, 2, 20, T) if (world == 1 and not args.no_predict) else None
        del full_src
        if rank == 0:
            c5 = {"metric": METRIC, "value": n5 * steps5 / dt / 1e6, "unit": "M corr/s", "icp_iter_per_s": steps5 / dt, "n_gpus": world, "steps": steps5,
                  "warmup": warm5, "ms_per_step": dt * 1e3 / steps5, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                  "data": "synthetic",
                  "config": {"workload": f"point-to-point ICP iteration on ONE {n_t} x {n5} pair = exact grid 1-NN + Kabsch + transform; BASELINE.json "
                                         "configs[4] (sources sharded over the GPUs, one all-reduce of 56 + 2N f64 per iteration)",
                             "nn": "grid", "n_src_this_rank": n_q, "n_src_total": n5, "n_tgt": n_t, "max_corr": 1.0,
                             "sharding": f"sources x{world} ({args.shard}), target replicated", "collective": collective,
                             "pose_err_vs_gt_fro": float(np.linalg.norm(T.astype(np.float64) - synth.gt_pose())),
                             "kept_pairs_last_iter": int(st["last_pairs"]), "lib_sha16": sha,
                             "pose_bits": "".join(f"{int(v):08x}" for v in np.ascontiguousarray(T, np.float32).view(np.uint32).ravel())},
                  "roofline": groof,
                  "kernels": stream_kernels(bd, n_q, int(st["last_pairs"]) * n_q // max(n5, 1))}
            if pred_c5 is not None:
                c5["predicted_scaling"] = pred_c5
            if args.workload == "c5":
                out = c5
            elif out is not None:
                out["c5"] = c5
        cs.free(); ct.free()

    if rank == 0 and out is not None:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def grid_roofline(ctx, pcr, np, cs, ct, T, n_q, n_t, kern_s, sha):
    """steady state of the exact grid search: the searches of an ICP loop that STARTS at the final pose (HIP-event duration of every
    search of the loop: the cold first one and the first seeded one are skipped), and one more loop with the diagnostics counters for
    the algorithmic bytes (SURVEY.md 8d "1-NN exact grid").  Large targets take the sign tile search (csrc/grid_stile.hpp), which needs the
    loop's sorted working cloud — a bare sequence of pcr_nn1_f32_async calls would measure the cell walk alone."""
    ctx.tune("nn_method", 2); ctx.tune("prof", 1); ctx.prof_reset()
    ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=8, eps=0.0)
    each = ctx.prof_get_each("nn1_grid")
    family = ctx.mfma_check()["last_nn1_kernel"]
    steady = each[2:] if each.size > 2 else each
    gl, steady_s = int(steady.size), float(steady.mean()) / 1e3
    ctx.tune("grid_stats", 1)
    ctx.icp_point2point(cs, ct, init_T=T, max_corr=1.0, max_iter=3, eps=0.0)
    w = ctx.nn1_stats()
    ctx.tune("grid_stats", 0); ctx.tune("prof", 0)
    tile = family == "grid-stile"
    gpmc = load_pmc("latest_pmc_grid.json", sha)
    if gpmc and (gpmc.get("n") != n_t or n_q != n_t):
        gpmc = None
    # bytes the search has to move through L2: per query its point, winner position and key; per opened x-row two bounds; per tested
    # sphere 16 B; per record LOADED 16 B — the tile search loads a record once for the (up to 32) queries of its pass, the cell walk
    # once per query that visits it
    rec_loads = w[10] if tile else w[0]                  # (tile: the walk's loads for the few deferred queries are not in [10])
    alg_bytes = n_q * (12.0 + 4.0 + 8.0) + 8.0 * w[1] + 16.0 * w[2] + 16.0 * rec_loads + (16.0 * 16.0 * n_q if tile else 0.0)
    L2_PEAK_GBS = 34500.0    # aggregate L2 bandwidth measured on MI355X (MI355X_MICROARCH.md, L2 section)
    return {
        "bound": "l2-gather", "achieved": alg_bytes / steady_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
        "frac": alg_bytes / steady_s / 1e9 / L2_PEAK_GBS,
        "traffic": (gpmc["fetch_bytes_per_launch_corrected_x2"] + gpmc["write_bytes_per_launch"]) if gpmc else None,
        "traffic_note": (f"HBM-side bytes per search (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, summed over the launches of a search) from separate "
                         f"rocprofv3 --pmc passes at {gpmc['n']} x {gpmc['n']} with this very library (sha {sha}), {gpmc['source']}") if gpmc
                        else "null: no PMC pass of this search at this size with the library loaded now (tools/gpu_pmc_grid.sh)",
        "compulsory_bytes": n_q * 24.0 + n_t * 16.0,
        "vs_hbm_stream_peak": alg_bytes / steady_s / 1e9 / HBM_PEAK_GBS,
        "compulsory_vs_hbm_peak": (n_q * 24.0 + n_t * 16.0) / steady_s / 1e9 / HBM_PEAK_GBS,
        "kernel": (("pcr::nn1_stile_kernel<false> (csrc/grid_stile.hpp, the sign tile search: one wave per 64 consecutive queries of the sorted working cloud, a lane "
                    "per query; quarter boxes -> coarse Morton cells -> tile spheres -> a wave-private list of tiles of 32 records; every listed tile through ONE "
                    "v_mfma_f32_32x32x16_f16 per half-wave with the query's threshold folded into the free K-slots — the accumulator's sign says whether the record "
                    "can matter (the exhaustive search's own exact-decision bound) — flagged (query, 16-record chunk) pairs evaluated exactly, four lanes per chunk) "
                    "+ pcr::nn1_grid_kernel<16, false, 2, true> (the cell walk over the queue of deferred query segments), at the converged pose, seeded by the previous winners")
) if tile else
                  "pcr::nn1_grid_kernel (exact uniform-grid 1-NN, cell walk) at the converged pose, seeded by the previous correspondences as inside the loop",
        "kernel_family": family,
        "launches": gl, "avg_launch_ms": steady_s * 1e3, "kernel_M_corr_per_s": n_q / steady_s / 1e6,
        "avg_launch_ms_over_the_timed_icp": kern_s * 1e3,
        "pair_evaluations_per_query": w[0] / max(n_q, 1), "rows_per_query": w[1] / max(n_q, 1), "sphere_tests_per_query": w[2] / max(n_q, 1),
        "records_loaded_per_query": rec_loads / max(n_q, 1), "deferred_to_the_walk_frac": w[6] / max(n_q, 1) if tile else None,
        "algorithmic": f"per query 12 B point + 4 B winner position + 8 B key, 8 B per opened x-row ({w[1] / max(n_q, 1):.2f}/query), 16 B per bounding "
                       f"sphere tested ({w[2] / max(n_q, 1):.1f}/query), 16 B per record loaded ({rec_loads / max(n_q, 1):.1f}/query; every loaded record is "
                       f"evaluated against up to 32 queries: {w[0] / max(n_q, 1):.0f} lower-bound / exact pair evaluations per query) and 256 B for the exact "
                       "evaluation of the winning run, counted by the kernels' diagnostics build.  Neighbouring groups need the same records: the bound is the "
                       "L2 gather rate; the search is bound by vector issue and dependent L2 round trips, not by bandwidth (profiles/r04_grid_pmc.md)"}


def bench_c4(ctx, pcr, synth, np, args):
    """BASELINE configs[3]: RANSAC ground plane (Homework4, ground_detection_ransac.py:131-153) + radius-NN on ONE 120k-point scan.
    step = the 80-hypothesis inlier count (40 per x-segment, :54,71-72) in one pass + the radius search r = 1 of every point of
    the scan against the scan (benchmark.hpp:14,66-70; iss_detector.cpp:48-56).  Kernel times from HIP events; the radius results
    (12 B per neighbour) come back over PCIe at this boundary — that time is reported apart, never in `value`."""
    n = args.points or 120000
    scan = synth.kitti_like_scan(n)
    ground = np.where(np.abs(scan[2] + 1.73) < 0.3)[0]
    pick = (synth.splitmix64(9, np.arange(240, dtype=np.uint64)) % np.uint64(ground.size)).astype(np.int64).reshape(80, 3)
    hw4 = importlib.import_module(PKG + ".hw4")
    planes = np.stack([hw4.estimate_plane_params(scan[:, ground[p]].T.astype(np.float64)) for p in pick])
    planes = planes[np.isfinite(planes).all(axis=1)]
    c = ctx.cloud(scan)
    ctx.tune("prof", 2)
    ctx.plane_count(c, planes, 0.15); ctx.prof_reset()
    reps = max(args.steps, 5)
    t0 = time.perf_counter()
    for _ in range(reps):
        counts = ctx.plane_count(c, planes, 0.15)
    call_ms = (time.perf_counter() - t0) * 1e3 / reps
    k, ms = ctx.prof_get("plane_count")
    plane_ms = ms / max(k, 1)
    plane_bytes = 12.0 * n
    # radius search: resident database, count-only pass and the full (count, fill, sort, distances) pass
    db = np.ascontiguousarray(scan.T.astype(np.float64))
    d = ctx.db64(db)
    L = pcr.lib()
    import ctypes as C
    row = np.zeros(n + 1, np.int64)
    L.pcr_db64_radius(ctx.h, d.h, db.ctypes.data, n, C.c_double(1.0), row.ctypes.data, None, None)      # builds + warms
    ctx.prof_reset()
    t0 = time.perf_counter()
    L.pcr_db64_radius(ctx.h, d.h, db.ctypes.data, n, C.c_double(1.0), row.ctypes.data, None, None)
    count_call_ms = (time.perf_counter() - t0) * 1e3
    total = int(row[-1])
    idx = np.zeros(max(total, 1), np.int32); dist = np.zeros(max(total, 1), np.float64)
    ctx.prof_reset()
    t0 = time.perf_counter()
    rc = L.pcr_db64_radius(ctx.h, d.h, db.ctypes.data, n, C.c_double(1.0), row.ctypes.data, idx.ctypes.data, dist.ctypes.data)
    full_call_ms = (time.perf_counter() - t0) * 1e3
    assert rc == 0
    # the same search with its rows kept in HBM (pcr_rows) and REDUCED there: what a consumer that only needs neighbour counts / sums /
    # moments pays at the boundary — the search, m + 1 offsets and m doubles over PCIe instead of 12 B per neighbour
    R = d.radius_rows(None, 1.0); R.free()                               # (code objects, allocator warm-up)
    t0 = time.perf_counter()
    R = d.radius_rows(None, 1.0)
    rows_search_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    cnt_r = R.reduce(R.COUNT); sum_r = R.reduce(R.SUM_DIST)
    rows_reduce_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    R.moments()
    rows_moments_ms = (time.perf_counter() - t0) * 1e3
    assert R.total == total and np.array_equal(cnt_r, np.diff(row).astype(np.float64))
    assert np.allclose(sum_r[:64], [dist[row[i]:row[i + 1]].sum() for i in range(64)], rtol=1e-12)
    t0 = time.perf_counter()
    blk_i, blk_d = R.fetch(0, min(n, 2048), row)                         # a bounded block of rows (the iterate access)
    rows_block_ms = (time.perf_counter() - t0) * 1e3
    R.free()
    kern = {}
    for name in ("radius_grid_build", "radius_count", "radius_emit", "radius_fill", "radius_sort", "radius_dist"):
        k, ms = ctx.prof_get(name)
        if k:
            kern[name] = ms / k
    radius_kernel_ms = sum(kern.values())
    ctx.tune("prof", 0)
    d.free(); c.free()
    step_ms = plane_ms + radius_kernel_ms
    out_bytes = 12.0 * total + 8.0 * n
    c4pmc = load_pmc("latest_pmc_c4.json", lib_sha16(pcr)) if n == 120000 else None
    res = {
        "metric": "M radius queries/sec + M point-hypothesis evaluations/sec, 120k-pt KITTI scan, 1 MI355X (BASELINE configs[3])",
        "value": n / step_ms / 1e3, "unit": "M scan points/s through (80-plane inlier count + radius-NN r = 1)", "n_gpus": 1, "steps": reps,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"RANSAC plane-inlier count ({planes.shape[0]} hypotheses, thr 0.15) + radius-NN r = 1.0 of all {n} points against the scan",
                   "n_points": n, "neighbours_reported": total, "best_plane_inliers": int(counts.max())},
        "plane_count": {"ms": plane_ms, "call_ms_incl_h2d_d2h": call_ms, "bytes": plane_bytes, "GB/s": plane_bytes / plane_ms / 1e6,
                        "frac_of_8TB/s": plane_bytes / plane_ms / 1e6 / HBM_PEAK_GBS,
                        "G_point_hypothesis_per_s": n * planes.shape[0] / plane_ms / 1e6,
                        "algorithmic": "12 B per point, ONE pass for all hypotheses (the reference reads the points once per hypothesis); 7 f64 ops per "
                                       "point x hypothesis: 1.44 MB is launch-latency-bound at this size"},
        "radius": {"kernel_ms": kern, "kernels_total_ms": radius_kernel_ms, "M_queries_per_s": n / radius_kernel_ms / 1e3,
                   "G_neighbours_per_s": total / radius_kernel_ms / 1e6, "count_only_call_ms": count_call_ms, "full_call_ms_incl_d2h": full_call_ms,
                   "result_bytes": out_bytes, "result_GB/s_of_kernels": out_bytes / radius_kernel_ms / 1e6,
                   "rows_handle": {"search_call_ms": rows_search_ms, "reduce_count_and_sum_ms": rows_reduce_ms, "moments_ms": rows_moments_ms,
                                   "search_plus_reduce_ms": rows_search_ms + rows_reduce_ms,
                                   "vs_kernels": (rows_search_ms + rows_reduce_ms) / radius_kernel_ms if radius_kernel_ms else None,
                                   "fetch_block_of_2048_rows_ms": rows_block_ms, "block_bytes": int(12 * blk_i.size),
                                   "note": "pcr_db64_radius_rows (self-query, the CSR stays in HBM) + pcr_rows_reduce x 2: wall time of the calls "
                                           "against the kernels' 'kernels_total_ms'; the full D2H of the rows is 'full_call_ms_incl_d2h'"},
                   "note": "results are 12 B per reported neighbour (i32 index + f64 distance): the call is D2H-bound at this boundary; kernels: grid build, "
                           "count, fill, per-row ascending-index sort, distances"},
        "roofline": {"bound": "hbm", "achieved": out_bytes / radius_kernel_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": out_bytes / radius_kernel_ms / 1e6 / HBM_PEAK_GBS,
                     "traffic": ((c4pmc["fetch_bytes_per_launch_corrected_x2"] + c4pmc["write_bytes_per_launch"]) if c4pmc else None),
                     "traffic_note": (f"HBM-side bytes of one filled call (FETCH_SIZE x 2 + WRITE_SIZE over its radius kernels), {c4pmc['source']}" if c4pmc else
                                      "null: no PMC pass of the radius kernels with the library loaded now (tools/gpu_pmc_c4.sh)"),
                     "kernel": "radius pipeline (count + fill + sort + dist)",
                     "algorithmic": "compulsory output bytes: 12 B per reported neighbour + 8 B row pointer per query"},
    }
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc
        t0 = time.perf_counter()
        oc = orc.plane_count(scan, planes, 0.15)
        cpu_plane_ms = (time.perf_counter() - t0) * 1e3
        assert np.array_equal(oc, counts)
        sel = np.arange(0, n, max(1, n // 600))
        base = {"plane_count_ms": cpu_plane_ms, "plane_kind": "port (oracle restatement of ground_detection_ransac.py:138-139, one core)", "cores": 1}
        if orc.have_ref():
            qs = np.ascontiguousarray(db[sel])
            rp = np.zeros(sel.size + 1, np.int64)
            t0 = time.perf_counter()
            orc.ref().ref_hw2_kd_radius(db, n, 3, qs, sel.size, 1.0, 1, rp, None, None)      # one build + the searches
            dtr = time.perf_counter() - t0
            base.update({"radius_M_queries_per_s": sel.size / dtr / 1e6, "kind": "reference",
                         "sample": f"the reference's hw2 kd-tree (leaf 1, build + radius 1.0 of {sel.size} of the {n} points, one core, benchmark.hpp:66-70)"})
        else:
            t0 = time.perf_counter()
            orc.radius_f64(db, db[sel[:100]], 1.0)
            dtr = time.perf_counter() - t0
            base.update({"radius_M_queries_per_s": 100 / dtr / 1e6, "kind": "port", "sample": f"oracle exhaustive radius, 100 of {n} queries, one core"})
        base["value"] = base["radius_M_queries_per_s"]; base["unit"] = "M radius queries/s"
        res["cpu_baseline"] = base
    return res


if __name__ == "__main__":
    main()
