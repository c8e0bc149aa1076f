#!/usr/bin/env python3
"""bench.py — headline benchmark of the k-NN correspondence + ICP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one point-to-point ICP iteration on the BASELINE.json configs[1]/[2] workload: 1-NN correspondence
of 120 000 source points against a 120 000-point target (exhaustive / brute force) + Kabsch accumulation +
(N > 1: one all-reduce of 16 f64 moments) + 3x3 SVD + in-place transform.  W untimed iterations, then
exactly K iterations timed between barrier + torch.cuda.synchronize() on both sides; max over ranks.
Weak scaling: every rank owns its own 120 000-point shard of the source cloud, the target is replicated.

value            = correspondences of the whole job per second (M corr/s) = N * n_src * K / t
icp_iter_per_s   = K / t
roofline         = the dominant kernel (nn1_brute): ALGORITHMIC work per launch / average launch duration
                   measured live with HIP events on the kernel's own stream (inside libpcr_hip.so)
cpu_baseline     = the reference's own nanoflann 1-NN (oracle/_ref, kind "reference") or, if that binary is
                   absent, the oracle's scalar brute force (kind "port"), timed on this box's host cores.
Inputs are resident in HBM before the timed region starts; data = synthetic (no dataset ships, no network).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "hands-on-point-cloud-processing_amd"

VALU_PEAK_TFLOPS_NOFMA = 78.6   # MI355X: 157.3 TF/s vector f32 counts FMA as 2; the contract forbids FMA
HBM_PEAK_GBS = 8000.0
OPS_PER_PAIR = 9                # 3 sub + 3 mul + 2 add + 1 compare per (query, target) pair


def cpu_baseline(src, tgt):
    """Rank 0, N = 1 only.  Bounded: one kd-tree build + 120k queries (~0.1-0.3 s) x 3 repetitions."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    cores = os.cpu_count() or 1
    n = src.shape[1]
    if orc.have_ref():
        best1, bestn = None, None
        for _ in range(3):
            _, _, b, q = orc.ref_nano_nn1_f32(tgt, src, leaf=2, threads=1)
            best1 = (b + q) if best1 is None else min(best1, b + q)
        nthr = min(cores, 16)
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
                "multi_thread": {"value": n / bestn / 1e3, "cores": nthr,
                                 "note": "same tree, queries split over std::thread"},
                "host_cpus": cores}
    # port: scalar brute force of the oracle on a bounded sample
    m = 400
    t0 = time.perf_counter()
    orc.nn1_f32(tgt, src[:, :m].copy())
    dt = time.perf_counter() - t0
    return {"value": m / dt / 1e6, "unit": "M corr/s", "cores": 1, "kind": "port",
            "sample": f"oracle scalar brute force, {m} of {n} queries against {tgt.shape[1]} targets",
            "host_cpus": cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=0, help="target points (and source points per GPU for c2); 0 = workload default")
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2 (default, BASELINE configs[1]/[2]): 120k x 120k pair per GPU, weak scaling, brute force; "
                         "c5 (BASELINE configs[4]): ONE 10M x 10M pair, sources sharded over the GPUs (strong scaling), exact grid")
    ap.add_argument("--collective", choices=["rccl", "torch"], default="rccl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--qpl", type=int, default=0)
    ap.add_argument("--tiles-per-slice", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0, help="nn1 kernel variant (0 = library default FTRACK; 2 = exact TRACK; 3 = exact TRACK through LDS tiles)")
    ap.add_argument("--no-grid-extra", action="store_true", help="skip the additional exact-grid pass")
    ap.add_argument("--nn", choices=["brute", "grid"], default="brute",
                    help="correspondence search: brute = BASELINE configs[1] (LDS-tiled brute force), grid = exact grid index")
    args = ap.parse_args()

    import numpy as np
    import torch   # device sync + torch.distributed (RCCL) plumbing; loaded BEFORE libpcr_hip.so so that the
                   # HIP runtime (libamdhip64.so.7) is shared
    pcr = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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

    if args.workload == "c5":
        # one pair for the whole job: every rank builds the same source cloud and keeps its contiguous shard
        n = args.points or 10_000_000
        args.nn = "grid"
        full_src, tgt = synth.kitti_like_pair(n)
        b, e = pcr.shard_range(n, world, rank)
        src = np.ascontiguousarray(full_src[:, b:e])
        del full_src
        scaling, total_src = "strong", n
    else:
        n = args.points or 120000
        src, tgt = synth.kitti_like_pair(n, n_src=n, shard=rank)
        scaling, total_src = "weak", world * n
    ctx = pcr.Context(device_index)
    for kv in filter(None, os.environ.get("PCR_TUNE", "").split(",")):     # experiments: PCR_TUNE="key=value,key=value"
        k, v = kv.split("=")
        ctx.tune(k.strip(), int(v))
    if args.qpl:
        ctx.tune("nn1_qpl", args.qpl)
    if args.tiles_per_slice:
        ctx.tune("nn1_tiles_per_slice", args.tiles_per_slice)
    if args.variant:
        ctx.tune("nn1_variant", args.variant)
    cs, ct = ctx.cloud(src), ctx.cloud(tgt)

    collective = "none"
    if world > 1:
        collective = args.collective
        if collective == "rccl":
            # native transport: the library's own RCCL communicator (one ncclAllReduce of <= 32 f64 per iteration, enqueued
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

    def timed_icp(method, prof=1):
        """W untimed + exactly K timed ICP iterations with the given correspondence search; max over ranks.
        prof = 1: every correspondence launch of the timed region is bracketed by a HIP event pair on the library's stream
        (the kernel's average duration for the roofline); the pairs cost ~13 us of stream time per iteration."""
        ctx.tune("nn_method", method)
        ctx.tune("prof", prof)
        # one-time preparation, whatever --warmup says: the index over the (replicated) target and the code objects of every kernel
        # of the loop — a 2-iteration ICP of a small slice of this rank's sources; initialisation, not a step
        n_prep = min(4096, src.shape[1])
        c_prep = ctx.cloud(np.ascontiguousarray(src[:, :n_prep]))
        ctx.icp_point2point(c_prep, ct, max_corr=1.0, max_iter=2, eps=0.0)
        c_prep.free()
        if args.warmup > 0:
            ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=args.warmup, eps=0.0)
        ctx.prof_reset()
        barrier()
        t0 = time.perf_counter()
        T, st = ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=args.steps, eps=0.0)   # eps = 0: never early-exits
        barrier()
        dt = time.perf_counter() - t0
        assert st["iters_run"] == args.steps, st
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return T, st, dt

    main_method = 1 if args.nn == "brute" else 2
    T, st, dt = timed_icp(main_method)
    nn_name = "nn1_brute" if args.nn == "brute" else "nn1_grid"
    nn_launches, nn_ms = ctx.prof_get(nn_name)
    # second, separately timed pass with the exact grid index (same answers, different search): extra info only
    grid_extra = None
    if args.nn == "brute" and not args.no_grid_extra:
        Tg, stg, dtg = timed_icp(2, prof=0)       # throughput without the event pairs (11 % of this much shorter step) ...
        ctx.tune("prof", 1); ctx.prof_reset()       # ... and the kernel time from a second, profiled run of the same loop
        ctx.icp_point2point(cs, ct, max_corr=1.0, max_iter=args.steps, eps=0.0)
        gl, gms = ctx.prof_get("nn1_grid")
        grid_extra = {"value": total_src * args.steps / dtg / 1e6, "unit": "M corr/s", "icp_iter_per_s": args.steps / dtg,
                      "ms_per_step": dtg * 1e3 / args.steps, "avg_nn_kernel_ms": gms / max(gl, 1),
                      "pose_bit_identical_to_brute_force": bool(np.array_equal(T.view(np.uint32), Tg.view(np.uint32))),
                      "note": "same ICP with pcr nn_method = grid (exact uniform-grid index, csrc/grid.hip); NOT the "
                              "BASELINE configs[1] workload, reported for information"}
    if rank == 0:
        pmc = None
        try:   # HBM traffic of the same kernel from a separate rocprofv3 --pmc pass (tools/gpu_check.sh), committed
            pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        except Exception:   # noqa: BLE001
            pmc = None
        total_corr = total_src * args.steps
        kern_s = nn_ms / 1e3 / max(nn_launches, 1)
        n_q, n_t = src.shape[1], tgt.shape[1]
        gt_err = float(np.linalg.norm(T.astype(np.float64) - synth.gt_pose()))
        if args.nn == "brute":
            pairs = float(n_q) * float(n_t)
            achieved_tflops = OPS_PER_PAIR * pairs / kern_s / 1e12
            compulsory_bytes = 12.0 * n_t + 12.0 * n_q + 8.0 * n_q        # targets + sources + (idx, d2) key
            have_pmc = pmc and pmc.get("valu_insts_per_launch") and "etrack" in pmc.get("kernel", "") and n_q == 120000 == n_t
            default_kernels = not (args.qpl or args.variant)
            roofline = {
                "bound": "valu", "achieved": achieved_tflops, "peak": VALU_PEAK_TFLOPS_NOFMA, "unit": "TFLOP/s",
                "frac": achieved_tflops / VALU_PEAK_TFLOPS_NOFMA,
                "traffic": (pmc["fetch_bytes_per_launch_corrected_x2"] + pmc["write_bytes_per_launch"]) if have_pmc else None,
                "traffic_note": ("HBM-side bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) from a separate "
                                 "rocprofv3 --pmc pass of this kernel, " + pmc["source"]) if have_pmc else "not collected",
                "clock_ghz_profiled": pmc.get("clock_ghz_profiled") if have_pmc else None,
                "kernel": ("pcr::nn1_etrack_kernel<4> (exhaustive scan; chunk-centred targets broadcast through the scalar cache; "
                           "expanded-form lower bound, 3 FMAs per pair, tracked branch-free; the winner decided with the exact unfused "
                           "arithmetic; the previous correspondence of each query — of the previous iteration, or of the warm-up run for "
                           "the first timed iteration — re-evaluated exactly, seeds the bound); a search without any earlier "
                           "correspondences (the very first of the preparation) runs the same kernel without a seed; avg_launch_ms "
                           "averages all launches of the timed region")
                          if default_kernels else f"nn1 variant={args.variant} qpl={args.qpl}",
                "executed_lane_ops_per_pair": (pmc["valu_insts_per_launch"] * 64 / pairs) if have_pmc else None,
                "issue_frac": (pmc["valu_insts_per_launch"] * 64 / kern_s / 1e12 / VALU_PEAK_TFLOPS_NOFMA) if have_pmc else None,
                # the warm kernel's instruction mix per (query, chunk of 16 targets) priced with the measured issue costs of
                # each instruction form (tools/ubench/valu_rate.hip -> profiles/r01_ubench_valu_rate.txt): 80.6 ns per wave
                "mix_bound_ms": (pairs / 16 / 64 * 80.6e-9 / 1024 * 1e3) if default_kernels else None,
                "mix_frac": (pairs / 16 / 64 * 80.6e-9 / 1024 / kern_s) if default_kernels else None,
                "launches": int(nn_launches), "avg_launch_ms": kern_s * 1e3, "kernel_M_corr_per_s": n_q / kern_s / 1e6,
                "algorithmic": f"{OPS_PER_PAIR} f32 lane-ops per (query,target) pair (SURVEY.md 8d) x {pairs:.3e} pairs/launch; "
                               "peak = 157.3 TF/s / 2: one op per lane per issue slot (the exact arithmetic has no FMA). "
                               "frac can exceed 1 because the kernels need fewer than 9 issue slots per pair: the hot loop "
                               "evaluates a cheaper filter (3 FMAs on chunk-centred targets; fused 6-op form in the cold "
                               "kernel) and only the winning chunk exactly; issue_frac = executed lane-ops of the warm kernel "
                               "(PMC) / time / peak counts every wave-instruction as one slot; v_pk_fma_f32 and min/max/med3 "
                               "take about two (microbenchmark), so mix_frac = mix_bound_ms / avg_launch_ms — the loop's "
                               "instruction mix priced per form, 1 024 SIMDs — is the share of the VALU actually used",
                "hbm_literal": {"bound": "hbm", "achieved": compulsory_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": compulsory_bytes / kern_s / 1e9 / HBM_PEAK_GBS,
                                "note": "compulsory bytes (32 B/point); brute force is VALU-bound, see DESIGN.md"}}
            workload = ("point-to-point ICP iteration = exhaustive 1-NN correspondence 120k x 120k + Kabsch + transform; BASELINE.json "
                        "configs[1]/[2] ('LDS-tiled brute force': the default kernels broadcast the target tiles through the scalar "
                        "cache, faster than the LDS-tiled variant, which --variant 3 selects; same results)")
        else:
            # steady state of the exact grid search: the source at the final pose, a few launches timed with HIP events,
            # one more launch with the diagnostics counters for the algorithmic bytes (SURVEY.md 8d "1-NN exact grid")
            ca = cs.clone()
            ctx.transform(ca, T)
            ctx.tune("nn_method", 2)
            ctx.nn1_async(ct, ca); ctx.sync(); ctx.prof_reset()
            for _ in range(5):
                ctx.nn1_async(ct, ca)
            gl, gms = ctx.prof_get("nn1_grid")
            ctx.tune("grid_stats", 1); ctx.nn1_async(ct, ca); ctx.sync(); gs = ctx.grid_stats(); ctx.tune("grid_stats", 0)
            ca.free()
            steady_s = gms / 1e3 / max(gl, 1)
            gpmc = None
            try:
                gpmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc_grid.json")))
                if gpmc.get("n") != n_t or n_q != n_t:
                    gpmc = None
            except Exception:   # noqa: BLE001
                gpmc = None
            alg_bytes = n_q * (12.0 + 4.0 + 8.0) + 8.0 * gs["fine_rows"] + 16.0 * gs["candidates"]
            L2_PEAK_GBS = 34500.0    # aggregate L2 bandwidth measured on MI355X (MI355X_MICROARCH.md, L2 section)
            roofline = {
                "bound": "l2-gather", "achieved": alg_bytes / steady_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / steady_s / 1e9 / L2_PEAK_GBS,
                "traffic": (gpmc["fetch_bytes_per_launch_corrected_x2"] + gpmc["write_bytes_per_launch"]) if gpmc else None,
                "traffic_note": (f"HBM-side bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) from separate rocprofv3 --pmc passes of this "
                                 f"kernel at {gpmc['n']} x {gpmc['n']}, {gpmc['source']}: the candidate records come out of L2, not HBM") if gpmc
                                else "collected at 10M x 10M only (profiles/latest_pmc_grid.json)",
                "vs_hbm_stream_peak": alg_bytes / steady_s / 1e9 / HBM_PEAK_GBS,
                "kernel": "pcr::nn1_grid_kernel<16, false> (exact uniform-grid 1-NN) at the converged pose",
                "launches": int(gl), "avg_launch_ms": steady_s * 1e3, "kernel_M_corr_per_s": n_q / steady_s / 1e6,
                "avg_launch_ms_over_the_timed_icp": kern_s * 1e3,
                "algorithmic": f"per query 12 B point + 4 B order + 8 B key, 8 B per opened x-row ({gs['fine_rows'] / n_q:.1f}/query) and "
                               f"16 B per visited candidate ({gs['candidates'] / n_q:.1f}/query), counted by the kernel's diagnostics "
                               "build.  Neighbouring queries visit the same cells, so most candidate records are served by L2 (the algorithmic "
                               "byte rate exceeds the 8 TB/s HBM stream peak): the bound is the L2 gather rate"}
            workload = (f"point-to-point ICP iteration on ONE {n_t} x {total_src} pair = exact grid 1-NN + Kabsch + transform"
                        + ("; BASELINE.json configs[4] (sources sharded over the GPUs)" if args.workload == "c5" else ""))
        out = {
            "metric": "M correspondences/sec + ICP iter/sec, 120k-pt KITTI pair, 1/2/4/8 MI355X",
            "value": total_corr / dt / 1e6, "unit": "M corr/s",
            "icp_iter_per_s": args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "nn": args.nn, "n_src_this_rank": n_q, "n_src_total": total_src, "n_tgt": n_t,
                       "max_corr": 1.0, "sharding": f"sources x{world}, target replicated",
                       "collective": collective, "pose_err_vs_gt_fro": gt_err,
                       "kept_pairs_last_iter": int(st["last_pairs"])},
            "roofline": roofline,
        }
        if grid_extra is not None:
            out["exact_grid"] = grid_extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(src, tgt)
        print(json.dumps(out))
    cs.free(); ct.free()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
