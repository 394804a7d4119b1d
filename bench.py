#!/usr/bin/env python3
"""bench.py — scan-to-map registrations/s on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch: (re)index the resident 200k-point local map
and register B = 1024 synthetic VLP-16 scans against it (2 outer iterations x [exact 5-NN
association + line/plane fit -> Ceres-semantics LM(<=6)+Huber(0.1)]), inputs already resident in
HBM.  With N > 1 every rank registers its own B scans (weak scaling) against a replicated map and
the poses are gathered with one RCCL all_gather per step.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
ASSOC_KERNEL_PREFIX = "knn5_scan2map_split_kernel"   # the dominant kernel (rocprofv3 name prefix): the whole-batch form of the 5-NN search


def build_inputs(B, map_points, seed_offset, extractor=None, scan_ids=None, with_map=True):
    """Synthetic world/map (shared by all ranks) + this rank's scans -> feature clouds.
    Weak scaling: B scans seeded by the rank (`seed_offset`).  Strong scaling: `scan_ids` = this rank's block of GLOBAL
    scan numbers; pose, guess and sensor noise of scan g depend on g alone, so the job is the same whatever the number
    of ranks.  with_map=False: the caller receives the map by broadcast (rank 0 builds it)."""
    from msf_loam_amd import synth
    world = synth.World(ground_half=synth.ground_half_for_target(map_points))
    map_corner, map_surf = synth.make_map(world) if with_map else (np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))
    if scan_ids is None:
        poses = synth.random_poses(B, synth.SEED + 2 + 7919 * seed_offset)
        rng = np.random.default_rng(synth.SEED + 3 + 7919 * seed_offset)
        guesses = np.stack([synth.perturb_pose(p, rng) for p in poses])
        scan_seeds = [synth.SEED + 100 + i + 100003 * seed_offset for i in range(B)]
    else:
        scan_ids = [int(g) for g in scan_ids]
        B = len(scan_ids)
        poses = np.stack([synth.random_poses(1, synth.SEED + 20000 + g)[0] for g in scan_ids]) if B else np.zeros((0, 7))
        guesses = np.stack([synth.perturb_pose(poses[i], np.random.default_rng(synth.SEED + 40000 + g)) for i, g in enumerate(scan_ids)]) if B else np.zeros((0, 7))
        scan_seeds = [synth.SEED + 60000 + g for g in scan_ids]
    corner, surf, co, so = [], [], [0], [0]
    raw = []
    for i in range(B):
        pts, ring, kind = synth.make_scan(world, poses[i], scan_seeds[i], with_kind=True)
        raw.append((pts, ring, kind))
    if extractor is not None:
        feats = extractor(raw) if raw else []
    else:
        feats = [synth.direct_features(p, k) for p, _, k in raw]
    for c, s in feats:
        corner.append(c); surf.append(s)
        co.append(co[-1] + len(c)); so.append(so[-1] + len(s))
    z = np.zeros((0, 4), np.float32)
    return dict(map_corner=map_corner, map_surf=map_surf, truth=poses, guesses=guesses, raw=raw, world=world,
                corner=np.concatenate(corner) if corner else z, surf=np.concatenate(surf) if surf else z,
                corner_off=np.array(co, np.int32), surf_off=np.array(so, np.int32))


def product_extractor(handle):
    """Feature clouds through the product's own GPU path: msfl_extract_features_batch +
    msfl_voxel_downsample_batch (0.2 m corner / 0.4 m surf, laser_mapping.cc:264-270), 256 scans per call."""
    def run(raw):
        out = []
        chunk = 256
        for s in range(0, len(raw), chunk):
            part = raw[s:s + chunk]
            pts = np.concatenate([p for p, _, _ in part])
            ring = np.concatenate([r for _, r, _ in part])
            off = np.cumsum([0] + [len(p) for p, _, _ in part]).astype(np.int32)
            feats = handle.extract_features_batch(pts, ring, off)
            # the layout msfl_features_batch uses on the device: per-scan regions at the input offsets
            full = np.zeros((off[-1], 4), np.float32)
            idx = {k: np.zeros(off[-1], np.int32) for k in ("less_sharp", "less_flat")}
            cnt = {k: np.zeros(len(part), np.int32) for k in ("less_sharp", "less_flat")}
            for b, f in enumerate(feats):
                full[off[b]:off[b] + len(f["full"])] = f["full"]
                for k in idx:
                    idx[k][off[b]:off[b] + len(f[k])] = f[k]
                    cnt[k][b] = len(f[k])
            c, c_off = handle.voxel_downsample_batch(full, off, 0.2, idx=idx["less_sharp"], count=cnt["less_sharp"])
            sf, s_off = handle.voxel_downsample_batch(full, off, 0.4, idx=idx["less_flat"], count=cnt["less_flat"])
            for b in range(len(part)):
                out.append((c[c_off[b]:c_off[b + 1]], sf[s_off[b]:s_off[b + 1]]))
        return out
    return run


def cpu_baseline(inp, sample, threads, reps=5):
    """Oracle (CPU restatement of the reference algorithm) timed on the box's host cores, BASELINE.md §3 protocol: one
    warm-up run, then the MEDIAN of `reps` runs, for both cost structures:
      value           kd-trees rebuilt for every registration (the reference: mapping_scan_matcher.cc:66-73 builds them per call)
      value_one_tree  one pair of kd-trees for the whole sample, built OUTSIDE the timed region (the GPU step indexes the
                      map once per batch; a serial tree build inside the region would be most of the time on many cores)
    The sample is the whole batch by default (>= 4 scans per hardware thread on a 256-thread host), so no thread idles."""
    from oracle import oracle as orc
    orc.build()
    n = min(sample, len(inp["guesses"]))
    co, so = inp["corner_off"][:n + 1], inp["surf_off"][:n + 1]
    corner, surf, guesses = inp["corner"][:co[-1]], inp["surf"][:so[-1]], inp["guesses"][:n]
    n1 = max(1, min(n, 16))
    t0 = time.perf_counter()
    _, _, stages = orc.match_scan2map_batch_timed(inp["map_corner"], inp["map_surf"], inp["corner"][:co[n1]], co[:n1 + 1],
                                                  inp["surf"][:so[n1]], so[:n1 + 1], inp["guesses"][:n1])
    single = n1 / (time.perf_counter() - t0)
    stages_ms = {k: 1e3 * v / n1 for k, v in stages.items()}          # per registration, one thread

    def timed(fn):
        fn()                                                           # warm-up (page faults, thread pool)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return out, float(np.median(ts)), ts

    (poses, status), t_rebuild, all_rebuild = timed(lambda: orc.match_scan2map_batch(
        inp["map_corner"], inp["map_surf"], corner, co, surf, so, guesses, threads=threads, rebuild_tree_per_scan=True))
    t0 = time.perf_counter()
    tc, ts_ = orc.KdTree(inp["map_corner"]), orc.KdTree(inp["map_surf"])
    t_trees = time.perf_counter() - t0
    _, t_one, all_one = timed(lambda: orc.match_scan2map_batch_trees(tc, ts_, corner, co, surf, so, guesses, threads=threads))
    return dict(value=n / t_rebuild, value_one_tree=n / t_one, single_thread_value=single, n=n, n_single=n1, poses=poses,
                stages_ms=stages_ms, reps=reps, tree_build_s=t_trees, runs_s=all_rebuild, runs_one_tree_s=all_one,
                threads_busy=min(n, threads))


def self_launch(n):
    """Re-run this command under torch.distributed.run with n local ranks; returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the kernel whose name starts with `kernel_prefix`, from the committed rocprofv3
    --pmc profile of this command (PMC counters cannot be sampled from inside the process).  A missing file
    means "no profile yet" (None); a profile WITHOUT the kernel is an error, never a silent null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        pt = json.load(f)
    # `<..., true>` is the candidate-counting instantiation of the 5-NN kernel (one extra, untimed step): not the product launch
    hits = [k for k in pt["kernels"] if k.startswith(kernel_prefix) and not k.replace(" ", "").endswith(",true>")]
    if not hits:
        raise KeyError("profiles/pmc_traffic.json has no kernel starting with %r (keys: %s)" % (kernel_prefix, sorted(pt["kernels"])))
    return sum(pt["kernels"][k]["hbm_bytes_per_launch"] for k in hits), pt["source"]


def valu_roof(kernel_prefix, launch_ms):
    """VALU issue roof of the dominant kernel (profiles/r03_valu_rate.md): wave-instructions per launch from the committed
    PMC profile x the kernel's static mix of 2.25-clock and 4.3-clock instructions (tools/valu_model.py from the ISA, issue
    rates measured by tools/ubench/valu_rate.hip), over the SIMD-clocks one launch has: 1 024 SIMDs x 2.4 GHz x its duration.
    None when either committed file is missing or does not hold the kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        with open(os.path.join(ROOT, "profiles", "r04b_valu_mix.json")) as f:
            mix = json.load(f)
    except OSError:
        return None
    hit = [k for k in pt["kernels"] if k.startswith(kernel_prefix) and not k.replace(" ", "").endswith(",true>") and "sq_insts_valu" in pt["kernels"][k]]
    mk = [k for k in mix["kernels"] if k.replace(" ", "") in (kernel_prefix, kernel_prefix + "<false>", kernel_prefix + "<false,false>", kernel_prefix + "_split_kernel", kernel_prefix + "<128>")] or \
         [k for k in mix["kernels"] if k.startswith(kernel_prefix) and not k.replace(" ", "").endswith(",true>")]
    if not hit or not mk or launch_ms <= 0:
        return None
    insts = sum(pt["kernels"][k]["sq_insts_valu"] for k in hit)      # a kernel class launched in several instantiations per pass (the fit's two)
    m = mix["kernels"][mk[0]]
    clk = m["clocks_per_inst_static_mix"]
    simd_clocks = 1024 * 2.4e9 * launch_ms * 1e-3
    return {"wave_insts_per_launch": insts, "clocks_per_inst": clk, "clocks_per_inst_class": mix["clocks"],
            "static_mix": {"fast_2clk": m["fast_2clk"], "four_clk": m["four_clk"], "slow": m["slow"]},
            "simd_clocks_needed": insts * clk, "simd_clocks_available": simd_clocks, "frac": insts * clk / simd_clocks,
            "source": "SQ_INSTS_VALU: %s; mix: profiles/r04b_valu_mix.json (static ISA counts); rates: profiles/r03_valu_rate.md; "
                      "clock 2.4 GHz x 256 CUs x 4 SIMDs" % pt["source"].split(" (")[0]}


def vmem_roof(kernel_prefix):
    """Vector-memory issue of a kernel from the committed PMC profile: how busy the texture addresser (the unit every
    global load / store instruction goes through, one per CU) was, and what kept it busy.  TA_BUSY_avr is the per-CU average of
    busy cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs.  None when the profile does not hold the counters."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
    except OSError:
        return None
    hit = [k for k in pt["kernels"] if k.startswith(kernel_prefix) and not k.replace(" ", "").endswith(",true>") and "ta_busy_avr" in pt["kernels"][k]]
    if not hit:
        return None
    k = pt["kernels"][hit[0]]
    cyc = k.get("grbm_gui_active", 0.0) / 8.0
    waves = k.get("sq_waves", 0.0)
    loads = k.get("sq_insts_vmem_rd", 0.0)
    if cyc <= 0 or waves <= 0:
        return None
    return {"ta_busy_frac": k["ta_busy_avr"] / cyc, "ta_busy_frac_max_cu": k.get("ta_busy_max", 0.0) / cyc,
            "vmem_loads_per_wavefront": loads / waves, "vmem_stores_per_wavefront": k.get("sq_insts_vmem_wr", 0.0) / waves,
            "l1_accesses_per_load": (k.get("tcp_total_cache_accesses_sum", 0.0) / loads) if loads else None,
            "valu_insts_per_wavefront": k.get("sq_insts_valu", 0.0) / waves, "lds_insts_per_wavefront": k.get("sq_insts_lds", 0.0) / waves,
            "source": pt["source"].split(" (")[0] + " (TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8 XCDs); SQ_INSTS_* / SQ_WAVES)"}


def kernel_roofs(launch_ms, alg_bytes):
    """Both roofs for each of the step's three kernel classes: {name: {ms, valu_frac, fetched_GBps, ...}}.  `launch_ms`:
    {kernel name prefix: measured average launch duration}; counters and instruction mixes come from the committed
    profiles (valid for the profiled workload only).  `fetched` is what the L2 missed ((2 x FETCH_SIZE + WRITE_SIZE) per
    launch: HBM plus Infinity-Cache hits), not algorithmic bytes."""
    out = {}
    for prefix, ms in launch_ms.items():
        if ms <= 0:
            continue
        try:
            traffic, _ = pmc_traffic(prefix)
        except KeyError:
            traffic = None
        v = valu_roof(prefix, ms)
        row = {"avg_launch_ms": ms, "valu_frac": None if v is None else v["frac"],
               "fetched_bytes_per_launch": traffic,
               "fetched_GBps": None if traffic is None else traffic / (ms * 1e-3) / 1e9,
               "fetched_frac_of_hbm_peak": None if traffic is None else traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if prefix in alg_bytes:
            row["algorithmic_bytes_per_launch"] = alg_bytes[prefix]
            row["algorithmic_GBps"] = alg_bytes[prefix] / (ms * 1e-3) / 1e9
        vm = vmem_roof(prefix)
        row["ta_busy_frac"] = None if vm is None else vm["ta_busy_frac"]
        row["vmem_loads_per_wavefront"] = None if vm is None else vm["vmem_loads_per_wavefront"]
        fr = [(row["valu_frac"] or 0.0, "valu"), (row["fetched_frac_of_hbm_peak"] or 0.0, "memory (L2 misses)"),
              (row["ta_busy_frac"] or 0.0, "vector-memory instruction issue (texture addresser)")]
        row["nearer_roof"] = max(fr)[1]
        out[prefix] = row
    return out


def host_buffer_rate(h, inp, B, reps=5):
    """PCIe-inclusive rate (batch wall time incl. H2D of the features, excl. the one-time map
    upload / index): features, guesses and results in pinned host memory, the map resident and indexed."""
    import torch
    c, s, g = (torch.from_numpy(inp[k]).pin_memory().numpy() for k in ("corner", "surf", "guesses"))
    co, so = inp["corner_off"], inp["surf_off"]
    for _ in range(2):
        h.match_scan2map_batch(c, co, s, so, g)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        h.match_scan2map_batch(c, co, s, so, g)
        ts.append(time.perf_counter() - t0)
    # the MEDIAN call: the host and its PCIe root are shared with the pod's other GPU slots, and one stalled copy among five calls
    # made a round-4 run report 80 k instead of 432 k registrations/s (gpurun_out/r04b/bench_driver_shape.json)
    dt = sorted(ts)[len(ts) // 2]
    return B / dt, 1e3 * dt, int(c.nbytes + s.nbytes + g.nbytes), [1e3 * t for t in ts], B / (sum(ts) / len(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scans", type=int, default=1024, help="scans per GPU per step (BASELINE configs[1]: 1024)")
    ap.add_argument("--map-points", type=int, default=200000)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --scans per GPU (the default; BASELINE configs[1] per rank); strong: --total-scans split over the ranks "
                         "in contiguous blocks (the shape of configs[3]: one fixed batch sharded across the GPUs)")
    ap.add_argument("--total-scans", type=int, default=0, help="strong scaling: scans of the whole job (default: --scans)")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="scans timed on the CPU oracle (0 disables; default: the whole batch)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="diagnostic: leave the per-kernel HIP events off during the timed steps (roofline.achieved is then 0)")
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="steps of the steady-state companion figure (value_steady) run AFTER the timed region; 0 disables it")
    ap.add_argument("--no-worlds", action="store_true", help="skip stages.worlds / config3_share / config4_share / pairs (round 5)")
    ap.add_argument("--world-scans", type=int, default=256, help="scans per world in stages.worlds")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive leg (value_incl_h2d)")
    ap.add_argument("--no-stages", action="store_true", help="skip the `stages` leg (extraction, voxel filters, scan-to-scan, raw-scan pipeline, SLAM step)")
    ap.add_argument("--stage-slam-scans", type=int, default=100, help="scans of the SLAM-step replay in `stages` (0 disables it)")
    ap.add_argument("--features", choices=["product", "direct"], default="product",
                    help="product: features from the GPU extraction + voxel kernels; direct: from ray-cast hit kinds")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, rendezvous on
        # 127.0.0.1) and hand their output through; rank 0 of the child job prints the one JSON line
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != world_size:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world_size))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # test hook (1-GPU boxes): MSFL_BENCH_SHARED_GPU=1 puts every rank on cuda:0 and swaps RCCL for gloo, so that the
    # N > 1 control flow (sharding by rank, pose gather, max-over-ranks timing, rank-0 reporting) can be exercised
    # where only one GPU exists; the numbers of such a run mean nothing
    shared_gpu = os.environ.get("MSFL_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ            # launched by torch.distributed.run (also exercises N=1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)

    from msf_loam_amd import capi, dist as mdist
    h = capi.Handle(local_rank)
    strong = args.scaling == "strong"
    total_scans = (args.total_scans or args.scans) if strong else world_size * args.scans
    extractor = None
    feature_source = args.features
    if feature_source == "product":
        extractor = product_extractor(h)
    t_prep = time.perf_counter()
    # rank 0 builds the local map, every other rank receives it by broadcast over the process group (scans
    # sharded, map replicated); the scans are this rank's own (weak) or its block of the job's scans (strong)
    if strong:
        lo, hi = mdist.shard_bounds(total_scans, rank, world_size)
        inp = build_inputs(0, args.map_points, rank, extractor, scan_ids=range(lo, hi), with_map=(rank == 0))
    else:
        inp = build_inputs(args.scans, args.map_points, rank, extractor, with_map=(rank == 0))
    B = len(inp["guesses"])
    # MSFL_BENCH_FORCE_COLLECTIVES=1: run the map broadcast and the ragged gather even in a one-rank group, so that a 1-GPU box
    # executes every RCCL call of the N-GPU path (init with device_id, broadcast, all_gather_into_tensor on device tensors)
    force_coll = use_dist and os.environ.get("MSFL_BENCH_FORCE_COLLECTIVES") == "1"
    if use_dist and (world_size > 1 or force_coll):
        bdev = torch.device("cpu") if shared_gpu else dev
        mc_t, ms_t = mdist.broadcast_map(inp["map_corner"], inp["map_surf"], src=0, device=bdev, force=force_coll)
        inp["map_corner"], inp["map_surf"] = mc_t.cpu().numpy(), ms_t.cpu().numpy()
    t_prep = time.perf_counter() - t_prep

    stream = torch.cuda.current_stream(dev)
    h.set_stream(stream.cuda_stream)
    d_map_c = torch.from_numpy(inp["map_corner"]).to(dev)
    d_map_s = torch.from_numpy(inp["map_surf"]).to(dev)
    d_corner = torch.from_numpy(inp["corner"]).to(dev)
    d_surf = torch.from_numpy(inp["surf"]).to(dev)
    d_guess = torch.from_numpy(inp["guesses"]).to(dev)
    # the pose gather sends equal blocks: in strong scaling the ranks' shares differ by at most one scan, padded to `cap`
    cap = (total_scans + world_size - 1) // world_size if strong else B
    d_poses = torch.zeros((cap, 7), dtype=torch.float64, device=dev)
    d_status = torch.zeros(cap, dtype=torch.int32, device=dev)
    gather = mdist.PoseGather(cap, dev) if use_dist else None
    n_mc, n_ms = len(inp["map_corner"]), len(inp["map_surf"])
    co, so = inp["corner_off"], inp["surf_off"]

    def step():
        d_poses[:B].copy_(d_guess)                                   # fresh initial guesses
        h.set_map(d_map_c, d_map_s, n_mc, n_ms, capi.MEM_DEVICE)      # kd-tree build equivalent, per batch
        h.match_scan2map_batch_device(B, d_corner, co, d_surf, so, d_poses, d_status)
        if gather is not None:
            if os.environ.get("MSFL_GATHER_ASYNC") == "1":
                gather.all_gather_async(d_poses, d_status)           # on a side stream, under the next step's compute
            else:
                gather.all_gather(d_poses, d_status)                 # RCCL pose gather on the compute stream (measured
                                                                     # faster at N=1: 1.844 vs 1.864 ms/step)

    def barrier():
        if gather is not None:
            gather.wait()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # a generation-2 garbage collection of the Python harness is a ~40 ms pause (measured): keep it out of the timed region -- and, since
    # round 6, out of the gap between the warm-up and the timed region: the device's clocks fall back within milliseconds of idling
    # (profiles/r04_step_ramp.json), so a 40 ms host pause right after the warm-up steps undid them
    import gc
    gc.collect()
    gc.disable()
    # timed region: HIP events around the dominant (5-NN) kernel only; every event pair costs ~6 us of
    # stream time, the other kernel classes are timed in a few extra, untimed steps afterwards
    h.set_timing(0 if args.no_kernel_timing else 2)
    for _ in range(args.warmup):
        step()
    barrier()
    h.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    timing = h.get_timing(reset=True)
    h.set_timing(1)
    for _ in range(3):
        step()
    barrier()
    timing_all = h.get_timing(reset=True)
    # one more step with the counting 5-NN instantiation: map points whose distance was evaluated (SURVEY.md 8d's
    # secondary figure; the HBM roofline is not what binds this kernel)
    h.set_timing(3)
    h.get_timing(reset=True)
    step()
    barrier()
    t_knn = h.get_timing(reset=True)
    knn_first, knn_seeded = t_knn.knn_candidates, t_knn.knn_candidates_seeded
    knn_candidates = knn_first + knn_seeded
    h.set_timing(0)
    # steady-state companion: the timed region above is `--steps` steps from wherever the clocks were after the
    # warm-up (the driver's 20 steps sit on the ramp, profiles/r04_step_ramp.json); 200 more steps, same bracket, never `value`
    steady_steps = args.steady_steps
    elapsed_steady = None
    if steady_steps > 0:
        barrier()
        t0 = time.perf_counter()
        for _ in range(steady_steps):
            step()
        barrier()
        elapsed_steady = time.perf_counter() - t0
    gc.enable()
    rank_ms = [1e3 * elapsed / args.steps]
    rccl_ranks = 1
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if not shared_gpu else "cpu")
        every = [torch.zeros_like(t) for _ in range(world_size)]
        dist.all_gather(every, t)
        rank_ms = [1e3 * float(e.item()) / args.steps for e in every]
        elapsed = max(float(e.item()) for e in every)          # MAX over ranks
        rccl_ranks = dist.get_world_size()
        if elapsed_steady is not None:
            t = torch.tensor([elapsed_steady], dtype=torch.float64, device=dev if not shared_gpu else "cpu")
            every = [torch.zeros_like(t) for _ in range(world_size)]
            dist.all_gather(every, t)
            elapsed_steady = max(float(e.item()) for e in every)

    poses_gpu = d_poses[:B].cpu().numpy()
    status_gpu = d_status[:B].cpu().numpy()
    poses_sha1 = None
    if strong:
        # every rank's block back in scan order (one more gather, outside the timed region): the job's result must not
        # depend on how many ranks shared it
        import hashlib
        if use_dist and (world_size > 1 or force_coll):
            t_p = torch.from_numpy(poses_gpu).to(dev if not shared_gpu else "cpu"); t_s = torch.from_numpy(status_gpu).to(dev if not shared_gpu else "cpu")
            all_p, all_s = mdist.gather_ragged(t_p, t_s, total_scans, force=force_coll)
            all_p = all_p.cpu().numpy()
        else:
            all_p = poses_gpu
        poses_sha1 = hashlib.sha1(np.ascontiguousarray(all_p).tobytes()).hexdigest()

    if rank == 0:
        from msf_loam_amd import synth
        total_regs = total_scans * args.steps
        value = total_regs / elapsed
        F_total = int(co[-1] + so[-1])
        n_outer = 2
        # ALGORITHMIC bytes: per association launch = F_total*(16 query + 5*16 neighbours)
        # + the map read once per batch, split over the n_outer launches + one pose per scan.
        alg_bytes_assoc = F_total * 96 + (n_mc + n_ms) * 16 / n_outer + B * 56
        assoc_ms = timing.ms_assoc / max(timing.launches_assoc, 1)
        solve_ms = timing_all.ms_solve / max(timing_all.launches_solve, 1)
        index_ms = timing_all.ms_index / 3.0          # per STEP (both maps): the three extra steps above; round 6: one timed span per msfl_set_map (pair build), two before
        achieved = alg_bytes_assoc / (assoc_ms * 1e-3) / 1e9 if assoc_ms > 0 else 0.0
        # HBM traffic of the dominant kernel: PMC counters cannot be sampled from inside this process, so
        # the figure comes from the committed rocprofv3 --pmc profile of this same command
        # (profiles/pmc_traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch); null only when the
        # workload is not the profiled one.
        traffic, traffic_src = None, "not the profiled workload (B=1024, 200k-pt map)"
        if B == 1024 and args.map_points == 200000:
            traffic, traffic_src = pmc_traffic(ASSOC_KERNEL_PREFIX)
        # whole step: every outer iteration reads each feature and its five neighbours once, the map once per batch,
        # one pose in and out per scan; kernels in between communicate through HBM, which is not
        # algorithmic
        valu = valu_roof(ASSOC_KERNEL_PREFIX, assoc_ms) if (B == 1024 and args.map_points == 200000) else None
        roof_bound = "valu" if (valu is not None and valu["frac"] > achieved / HBM_PEAK_GBS) else "hbm"
        alg_bytes_step = n_outer * F_total * 96 + (n_mc + n_ms) * 16 + B * 112
        step_s = elapsed / args.steps
        out = {
            "metric": "scan-to-map registrations/s", "value": value, "unit": "registrations/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "value_steady": None if elapsed_steady is None else total_scans * steady_steps / elapsed_steady,
            "ms_per_step_steady": None if elapsed_steady is None else 1e3 * elapsed_steady / steady_steps, "steady_steps": steady_steps,
            "rccl_ranks": rccl_ranks, "ms_per_step_per_rank": rank_ms,
            "vs_baseline": None, "dtype": "f64 (f32 kNN distances)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch of %d VLP-16 scans (16x1800) vs %d-pt local map, "
                                   "2 outer x [exact 5-NN + line/plane fit, LM<=6 + Huber 0.1]" % (B, n_mc + n_ms),
                       "scans_per_gpu": B if not strong else cap, "total_scans": total_scans, "map_points": n_mc + n_ms, "map_corner": n_mc, "map_surf": n_ms,
                       "features_per_scan": F_total / B, "feature_source": feature_source,
                       "index_rebuilt_per_step": True, "parallelism": "scan-sharded x%d, map replicated" % world_size},
            "roofline": {"bound": roof_bound, "kernel": ASSOC_KERNEL_PREFIX, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes_assoc, "avg_launch_ms": assoc_ms, "valu": valu,
                         "vmem": vmem_roof(ASSOC_KERNEL_PREFIX) if (B == 1024 and args.map_points == 200000) else None,
                         "bound_note": "achieved / peak / frac are the HBM figures on ALGORITHMIC bytes (SURVEY.md 8d); `bound` names the roof the "
                                       "kernel is nearer to: its working set is L2-resident (traffic < algorithmic bytes) and its VALU issue "
                                       "fraction (`valu.frac`) is the larger one; `vmem.ta_busy_frac` is the second resource at ~0.8: a "
                                       "multi-dword load instruction costs the CU's texture addresser ~16 clocks whatever it fetches (profiles/r04b_vmem_rate.md)",
                         "kernels": (kernel_roofs({ASSOC_KERNEL_PREFIX: assoc_ms, "fit_scan2map": timing_all.ms_fit / max(timing_all.launches_fit, 1),
                                                   "lm_solve_kernel": solve_ms},
                                                  {ASSOC_KERNEL_PREFIX: alg_bytes_assoc, "fit_scan2map": F_total * (5 * 16 + 20 + 40),
                                                   "lm_solve_kernel": F_total * 48})
                                     if (B == 1024 and args.map_points == 200000) else None),
                         "step": {"algorithmic_bytes_per_step": alg_bytes_step, "achieved": alg_bytes_step / step_s / 1e9,
                                  "unit": "GB/s", "frac": alg_bytes_step / step_s / 1e9 / HBM_PEAK_GBS}},
            "kernels_ms": {"assoc": assoc_ms,
                           "assoc_pass1": (timing.ms_assoc - timing.ms_assoc_seeded) / max(timing.launches_assoc - timing.launches_assoc_seeded, 1),
                           "assoc_pass2": timing.ms_assoc_seeded / max(timing.launches_assoc_seeded, 1) if timing.launches_assoc_seeded else None, "fit": timing_all.ms_fit / max(timing_all.launches_fit, 1), "solve": solve_ms,
                           "index_build": index_ms,
                           "note": "assoc: HIP events inside the timed region; fit / solve / index_build: 3 extra steps after it; per launch, index_build per step (both maps)",
                           "launches": {"assoc": timing.launches_assoc, "solve": timing_all.launches_solve,
                                        "index": timing_all.launches_index}},
            "knn": {"candidates_per_launch": knn_candidates // 2, "candidates_per_query": knn_candidates / 2 / max(F_total, 1),
                    "candidates_per_query_pass1": knn_first / max(F_total, 1), "candidates_per_query_pass2": knn_seeded / max(F_total, 1),
                    "pass2_seeded": os.environ.get("MSFL_KNN_SEED", "0") == "1",
                    "distance_evals_per_s": (knn_candidates / 2) / (assoc_ms * 1e-3) if assoc_ms > 0 else 0.0,
                    "note": "map points whose f32 distance one launch evaluates (counting instantiation, one extra step); "
                            "rate = count / the timed launches' average duration"},
            "prep_s": t_prep,
            "n_failed": int((status_gpu != 0).sum()),
        }
        if strong:
            out["poses_sha1"] = poses_sha1
            out["config"]["map_source"] = "built on rank 0, broadcast over the process group" if (world_size > 1 or force_coll) else "built on rank 0"
            out["config"]["process_group_backend"] = dist.get_backend() if use_dist else None
        # fetched / algorithmic bytes of the LM solve per instantiation, from the committed PMC profiles
        try:
            with open(os.path.join(ROOT, "profiles", "r06_lm_traffic.json")) as f:
                out["roofline"]["lm_instantiations"] = {k: v for k, v in json.load(f).items() if k != "note"}
        except OSError:
            out["roofline"]["lm_instantiations"] = None
        out["value_incl_h2d"] = None               # N=1 only: the host-buffer call is a separate, untimed-for-`value` leg
        if world_size == 1 and not args.no_h2d:
            v, ms, nbytes, ms_calls, v_mean = host_buffer_rate(h, inp, B)
            out["value_incl_h2d"] = {"value": v, "value_is": "median call", "value_mean_of_calls": v_mean, "unit": "registrations/s", "ms_per_batch": ms,
                                     "ms_per_call": ms_calls, "host_bytes_in": nbytes,
                                     "note": "`value` = the MEDIAN of the listed calls (rounds 1-3 reported the mean: `value_mean_of_calls`); same batch with features, guesses and results in pinned host memory (PCIe "
                                             "staging inside the call), map resident and indexed; never `value`"}
        out["cpu_baseline"] = None                 # timed on rank 0 at N=1 only (the other ranks would idle behind it)
        if args.cpu_sample > 0 and world_size == 1:
            cores = os.cpu_count() or 1
            cb = cpu_baseline(inp, args.cpu_sample, cores)
            dts, drs = zip(*[synth.pose_error(poses_gpu[i], cb["poses"][i]) for i in range(cb["n"])])
            out["cpu_baseline"] = {"value": cb["value"], "unit": "registrations/s", "cores": cores, "kind": "port",
                                   "threads_busy": cb["threads_busy"],
                                   "single_thread_value": cb["single_thread_value"],
                                   "value_one_tree_per_batch": cb["value_one_tree"],
                                   "protocol": "warm-up 1, median of %d runs; runs (s): per-registration trees %s, one tree pair %s; "
                                               "the shared kd-trees of value_one_tree_per_batch are built outside the timed region (%.3f s, serial)"
                                               % (cb["reps"], ["%.3f" % t for t in cb["runs_s"]], ["%.3f" % t for t in cb["runs_one_tree_s"]], cb["tree_build_s"]),
                                   "like_for_like": "`value` rebuilds the kd-tree for every registration (the reference's cost "
                                                    "structure, mapping_scan_matcher.cc:66-73); `value_one_tree_per_batch` builds it "
                                                    "once for the sample, which is what the GPU step (one index build per batch) does",
                                   "sample": "%d of the %d scans of rank 0, oracle/msfl_oracle.c with per-registration "
                                             "kd-tree rebuild, OpenMP over scans on %d threads (single-thread figure on %d scans)"
                                             % (cb["n"], B, cores, cb["n_single"])}
            out["pose_delta_vs_oracle"] = {"max_m": max(dts), "max_rad": max(drs), "n": cb["n"], "tolerance": 1e-4}
            # the reference's LOG_STEP_TIME stages (mapping_scan_matcher.cc:73,248,264,275), CPU port per registration on one
            # thread next to the GPU's share of a batch divided by its scans
            k = out["kernels_ms"]
            gpu = {"build tree": k["index_build"], "Data association": 2 * (k["assoc"] + k["fit"]), "Solver time": 2 * k["solve"]}
            cpu = dict(cb["stages_ms"])
            for t in (gpu, cpu):
                t["Optimization twice"] = t["Data association"] + t["Solver time"]
            out["stages_MAP"] = {"cpu_port_ms_per_registration_1_thread": cpu,
                                 "gpu_ms_per_batch": gpu, "gpu_us_per_registration": {n_: 1e3 * v / B for n_, v in gpu.items()}}
        # the other stages of the hot path (the raw-scan pipeline, configs[2]'s per-scan step): measured on
        # this batch's own raw scans AFTER everything that feeds `value`, each spot-checked against the oracle (checker use only)
        out["stages"] = None
        if world_size == 1 and not args.no_stages and not strong and inp.get("raw"):
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_stages
            checker = None
            if args.cpu_sample > 0:
                from oracle import oracle as checker_mod
                checker_mod.build()
                checker = checker_mod
            t_st = time.perf_counter()
            out["stages"] = bench_stages.measure(inp["raw"], inp["world"], inp["map_corner"], inp["map_surf"], inp["truth"], inp["guesses"],
                                                 device=local_rank, slam_scans=args.stage_slam_scans, checker=checker)
            if not args.no_worlds:
                import bench_worlds
                out["stages"]["worlds"] = bench_worlds.measure_worlds(device=local_rank, checker=checker, scans=args.world_scans)
                out["stages"].update(bench_worlds.measure_shares(device=local_rank, checker=checker))
            out["stages"]["wall_s"] = time.perf_counter() - t_st
        print(json.dumps(out))
    h.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
