#!/usr/bin/env python
"""bench.py — throughput of the per-locus likelihood engine on synthetic pileups (BASELINE.json metric).

One "step" = one pass of the hot path (vlr_batch_run) over the rank's resident batch of candidate
loci + the all-gather of the full result records when N > 1.  Workload at N=1: BASELINE configs[2]
(tumor-normal with contamination, SNV+indel, 100x, 1M loci); weak scaling: every rank gets its own
shard of that size.  Prints ONE JSON line on rank 0.

  --workload config2|config3|config4|config5   BASELINE configs[1..4] (locus counts of the config, or --loci)
  --workload realign                          second hot path: read-vs-allele pair HMM, pairs/s
  --workload cli                              end to end through the process boundary: observation BCFs -> call variants -> calls BCF
  --afd / --no-afd                            (default on) additionally time the step WITH the AFD lists (the reference always
                                              computes AFD, calling.rs:889-928) and report it as `with_afd` in the same line;
                                              `value` stays the BASELINE metric (plain step), the roofline is the plain kernel's
  --gpus N                                    N > 1 without a torchrun environment: bench.py starts its own ranks
                                              (python -m torch.distributed.run --nproc-per-node N on 127.0.0.1) and relays rank 0's line
  --dry-run                                   launcher / collective check without a GPU (gloo, synthetic result records)
"""
import argparse
import ctypes as C
import math
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6  # half of the 157.3 TF f32 vector peak (SURVEY.md §8d secondary figure)
FLOP_PER_TERM = 3  # one observation term of one VAF point: fma(q, alpha, c) and the running product
FLOP_PER_CELL = 13  # pair-HMM cell in linear space: 5 multiplies + 4 fused multiply-adds
DEFAULT_LOCI = {"config2": 100_000, "config3": 1_000_000, "config4": 1_250_000, "config5": 625_000}  # configs 4/5: 10 M / 5 M over 8 GPUs


def effective_cpus():
    """CPUs this process may use: affinity mask and cgroup CPU quota (a container reports the host's hardware threads)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _gen_chunk(args):
    name, n, chunk = args
    from varlociraptor_amd import synth
    cfg = synth.CONFIGS[name]()
    return synth.generate(cfg, n, chunk=chunk)


def generate(name, n_loci, rank, chunk_loci=50000, workers=None):
    from varlociraptor_amd.batch import PileupBatch
    chunks = []
    left, k = n_loci, 0
    while left > 0:
        m = min(left, chunk_loci)
        chunks.append((name, m, rank * 100000 + k))
        left -= m
        k += 1
    if len(chunks) == 1:
        return _gen_chunk(chunks[0])
    workers = workers or min(len(chunks), max(1, (os.cpu_count() or 8) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    with ProcessPoolExecutor(max_workers=workers) as ex:
        parts = list(ex.map(_gen_chunk, chunks))
    return PileupBatch.concat(parts)


def _gen_pairs(args):
    n_reads, seed, window = args
    from varlociraptor_amd import realign_synth
    pb, _ = realign_synth.generate(n_reads, seed=seed, window=window)
    return pb.x, pb.y, pb.q, pb.band


def traffic_for(workload, n_units, build_id):
    """HBM bytes per launch from the PMC passes kept under profiles/ — only if they were taken from THIS build and size."""
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if not os.path.exists(tpath):
        return None
    with open(tpath) as fh:
        tj = json.load(fh)
    if tj.get("n_units", tj.get("n_loci")) != n_units or tj.get("build_id") != build_id:
        return None
    return tj.get("hbm_bytes_per_launch")


SIMDS = 256 * 4          # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
SHADER_CLOCK_HZ = 2.4e9  # peak engine clock
VALU_CYCLES_PER_INST = 4  # a wave64 VALU instruction occupies its SIMD's 16 lanes for four cycles (f64 FMA / MUL / ADD: full rate on this part)


def valu_for(workload, build_id, n_loci, kernel_ms):
    """The binding resource of the call kernel from the PMC stamp under profiles/ (tools/valu_stamp.py) — only if it was taken from
    THIS build: VALU issue cycles of the launch over the SIMD cycles it had (valu_busy), the share of f64 FMA / MUL / ADD among the
    VALU instructions (f64_share), and the instruction counts they come from.  None without a stamp of this build."""
    vpath = os.path.join(ROOT, "profiles", "valu_%s.json" % workload)
    if not os.path.exists(vpath):
        return None
    with open(vpath) as fh:
        vj = json.load(fh)
    if vj.get("build_id") != build_id:
        return None
    per = vj["per_locus"]
    busy = per["valu_insts"] * VALU_CYCLES_PER_INST * n_loci / (SIMDS * SHADER_CLOCK_HZ * kernel_ms * 1e-3)
    return {"valu_busy": busy, "f64_share": vj["f64_share"], "valu_insts_per_locus": per["valu_insts"], "salu_insts_per_locus": per["salu_insts"],
            "lane_utilisation": vj.get("lane_utilisation"), "stamp": "profiles/valu_%s.json (%d loci)" % (workload, vj["n_units"])}


def reference_binary_baseline(cfg, batch, n_loci, cores):
    """BASELINE.md §2 / SURVEY §8(d): if a `varlociraptor` executable is on the box, time the REFERENCE's own `call variants` on
    observation BCFs written from the first `n_loci` loci of the same synthetic batch, one process per effective CPU over
    contiguous shards of the records.  Returns None when no binary is present (the only state this image has been seen in)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("varlociraptor")
    if exe is None:
        return None
    from varlociraptor_amd import ingest
    names = cfg.scenario.sample_names
    tumor_normal = cfg.purity is not None and sorted(names) == ["normal", "tumor"]
    if not tumor_normal and len(names) != 1:
        return {"value": None, "note": "binary found at %s, but only the tumor-normal and single-sample scenarios have a command line here" % exe}
    tmp = tempfile.mkdtemp(prefix="vlr_refbin_")
    try:
        version = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=30).stdout.strip()
        per = -(-n_loci // cores)
        cmds = []
        for k in range(cores):
            lo, hi = min(n_loci, k * per), min(n_loci, (k + 1) * per)
            if hi <= lo:
                continue
            sub = batch.select(np.arange(lo, hi))
            paths = {}
            for s, name in enumerate(names):
                paths[name] = os.path.join(tmp, "%s_%d.bcf" % (name, k))
                ingest.write_observations(paths[name], sub, s)
            out = os.path.join(tmp, "calls_%d.bcf" % k)
            if tumor_normal:
                cmd = [exe, "call", "variants", "tumor-normal", "--tumor", paths["tumor"], "--normal", paths["normal"], "--purity", str(cfg.purity)]
            else:
                sc = os.path.join(tmp, "scenario.yaml")
                if not os.path.exists(sc):
                    with open(sc, "w") as f:
                        f.write("samples:\n  %s:\n    resolution: 0.01\n    universe: \"[0.0,1.0]\"\nevents:\n  present: \"%s:]0.0,1.0]\"\n" % (names[0], names[0]))
                cmd = [exe, "call", "variants", "generic", "--scenario", sc, "--obs", "%s=%s" % (names[0], paths[names[0]])]
            cmds.append((cmd, out))
        t0 = time.perf_counter()
        procs = [subprocess.Popen(c, stdout=open(o, "wb"), stderr=subprocess.PIPE) for c, o in cmds]
        errs = [p.communicate()[1] for p in procs]
        dt = time.perf_counter() - t0
        bad = [e.decode(errors="replace")[-300:] for p, e in zip(procs, errs) if p.returncode != 0]
        if bad:
            return {"value": None, "binary": exe, "version": version, "note": "call variants failed: " + bad[0]}
        return {"value": n_loci / dt, "unit": "loci/s", "cores": len(cmds), "kind": "reference", "binary": exe, "version": version,
                "sample": "first %d loci of the same batch written as observation BCFs, %d processes x contiguous shards, %.1f s (process start-up and BCF I/O included)" % (n_loci, len(cmds), dt)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bench_realign(args, rank, world, local_rank, dev):
    import torch
    import torch.distributed as dist
    from varlociraptor_amd import engine, realign
    n_reads = 4000 if args.loci is None else max(50, args.loci // 2)
    jobs = [(250, 1000 * rank + k, args.read_window) for k in range(max(1, n_reads // 250))]
    with ProcessPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 8)) as ex:
        parts = list(ex.map(_gen_pairs, jobs))
    base = realign.PairBatch()
    for x, y, q, band in parts:
        base.x += x; base.y += y; base.q += q; base.band += band
    tile = 64 if args.loci is None else 1  # the same pairs many times over: the kernel's work per pair does not depend on its neighbours
    pb = realign.PairBatch()
    pb.x, pb.y, pb.q, pb.band = base.x * tile, base.y * tile, base.q * tile, base.band * tile
    dp = realign.DevicePairs(pb, dev)
    gap = realign.GapParams()
    mode = args.mode
    hop = None
    if mode == "homopolymer":  # nanopore-like run-length error rates (HopParams, realignment/pairhmm.rs:207-295; the default is all zero)
        hop = realign.HopParams([math.log(0.02)] * 4, [math.log(0.03)] * 4, [math.log(0.3)] * 4, [math.log(0.3)] * 4)
    if mode == "fast":
        L = realign._bind()
        L.vlr_realign_fast_batch.restype = C.c_int
        L.vlr_realign_fast_batch.argtypes = [C.c_int, C.POINTER(realign.RealignDesc), C.c_void_p, C.c_void_p]

    def run_once():
        if mode == "exact":
            dp.run(gap, local_rank, stream)
        elif mode == "homopolymer":
            dp.run_homopolymer(gap, hop, local_rank, stream)
        else:
            p = [t.data_ptr() for t in dp.t]
            d = realign.RealignDesc(dp.n, p[0], p[1], p[2], p[3], p[4], None, gap.as_array())
            rc = L.vlr_realign_fast_batch(local_rank, C.byref(d), dp.out.data_ptr(), stream)
            assert rc == 0, rc
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(args.warmup):
        run_once()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)  # same stream as the launches
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        run_once()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the edit-distance pre-filter of the same pairs (calc_best_hit): its kernel time, and that the band it yields is the one the
    # synthetic batch carries (host routine)
    band_host = dp.t[5].clone()
    dp.band_from_hits(local_rank, stream)
    ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev2.record()
    for _ in range(args.steps):
        dp.band_from_hits(local_rank, stream)
    ev3.record()
    torch.cuda.synchronize()
    edit_ms = ev2.elapsed_time(ev3) / args.steps
    bands_equal = bool(torch.equal(band_host, dp.t[5]))
    if rank != 0:
        return None
    got = dp.out.cpu().numpy()
    n_pairs = len(pb)
    parity = cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        cores = effective_cpus()
        n_cpu = min(len(base), args.cpu_loci or 40 * cores)
        sub = realign.PairBatch()
        sub.x, sub.y, sub.q, sub.band = base.x[:n_cpu], base.y[:n_cpu], base.q[:n_cpu], base.band[:n_cpu]
        oracle.lib()
        tc = time.perf_counter()
        if mode == "exact":
            ref = oracle.pairhmm_batch(sub, gap, threads=cores)
        elif mode == "homopolymer":
            ref = oracle.homopoly_batch(sub, gap, hop)
        else:
            g4 = [gap.prob_insertion_artifact, gap.prob_deletion_artifact, gap.prob_insertion_extend_artifact, gap.prob_deletion_extend_artifact]
            ref = np.array([oracle.pathhmm_best(sub.x[k], sub.y[k], sub.q[k], g4) for k in range(n_cpu)])
        t_cpu = time.perf_counter() - tc
        d = np.abs(got[:n_cpu] - ref)
        parity = {"n_checked": int(n_cpu), "max_abs_dlnprob": float(np.nanmax(d)), "vs": "CPU restatement (oracle/vlr_realign_oracle.cpp), parity unpinned for the third-party recursion"}
        cpu = {"value": n_cpu / t_cpu, "unit": "pairs/s", "cores": cores if mode == "exact" else 1, "kind": "port",
               "sample": "first %d pairs of the same batch, %d threads, %.1f s" % (n_cpu, cores if mode == "exact" else 1, t_cpu)}
    achieved = dp.bytes / (kernel_ms * 1e-3) / 1e9
    cells = dp.cells
    return {
        "metric": "read-allele pairs/sec (pair HMM, whole node)" if mode == "exact" else "read-allele pairs/sec (pair HMM mode %s, whole node)" % mode, "value": n_pairs * world * args.steps / elapsed, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "realign: %d read-allele pairs/GPU (%d distinct, SNV/MNV/insertion/deletion loci, read windows %d..%d bases, reference windows %d bases, banded)" % (n_pairs, len(base), args.read_window, min(128, 2 * args.read_window), 3 * args.read_window),
                   "parallelism": "pairs sharded x%d, no collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic_for("realign", n_pairs, engine.build_id()), "algorithmic_bytes_per_launch": int(dp.bytes), "kernel_ms": kernel_ms,
                     "note": "recurrence, f64 VALU bound: see valu",
                     "valu": {"cells_per_s": cells / (kernel_ms * 1e-3), "flop_per_cell": FLOP_PER_CELL,
                              "achieved_tflops": cells * FLOP_PER_CELL / (kernel_ms * 1e-3) / 1e12, "peak_tflops": F64_VALU_PEAK_TFLOPS,
                              "frac": cells * FLOP_PER_CELL / (kernel_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TFLOPS}},
        "edit_distance_prefilter": {"kernel_ms": edit_ms, "pairs_per_s": n_pairs / (edit_ms * 1e-3), "bands_equal_host_routine": bands_equal,
                                    "note": "vlr_edit_distance_batch + band update on the resident pairs, not part of `value`"},
        "cpu_baseline": cpu, "parity": parity, "build_id": engine.build_id(),
    }


def self_launch(argv, n):
    """`python bench.py --gpus N` outside torchrun: start one rank per GPU on this node and relay their output."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """No GPU: the ranks rendezvous over gloo, shard a synthetic record set like the timed step does and all-gather it."""
    import torch
    import torch.distributed as dist
    from varlociraptor_amd.dist import all_gather_records, shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n_loci = args.loci or 1000
    n_total = n_loci * world
    width = 9
    mine = torch.arange(rank * n_loci, (rank + 1) * n_loci, dtype=torch.float64)[:, None].repeat(1, width)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = all_gather_records(mine, n_total, world) if world > 1 else mine
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    ok = bool(torch.equal(full[:, 0], torch.arange(n_total, dtype=torch.float64)))
    assert shard_range(n_total, rank, world) == (rank * n_loci, (rank + 1) * n_loci)
    if rank == 0:
        print(json.dumps({"metric": "candidate loci/sec (whole node)", "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "gathered_in_order": ok, "ms_per_step": el / args.steps * 1e3, "value": None}))
    if world > 1:
        dist.destroy_process_group()


def bench_cli(args, rank, world, local_rank, dev):
    """End to end through the process boundary: observation BCFs on disk -> `call variants` (native BGZF/BCF/v15
    ingest, vlr_batch_run_host with AFD lists, native calls writer) -> calls BCF on disk.  One step = one whole run of
    cli.call_variants over the rank's files; the synthetic observation files are written once, outside the timed region."""
    import tempfile
    import torch
    import torch.distributed as dist
    from varlociraptor_amd import cli, engine, ingest, synth
    n_loci = args.loci or 200_000
    cfg = synth.CONFIGS["config3"]()
    batch = generate("config3", n_loci, rank)
    tmp = tempfile.mkdtemp(prefix="vlr_cli_%d_" % rank, dir=os.environ.get("VLR_BENCH_TMP", None))
    names = cfg.scenario.sample_names
    paths = {}
    t0 = time.perf_counter()
    for s, name in enumerate(names):
        paths[name] = os.path.join(tmp, "%s.bcf" % name)
        ingest.write_observations(paths[name], batch, s)
    t_write_obs = time.perf_counter() - t0
    obs_bytes = sum(os.path.getsize(p) for p in paths.values())
    out_path = os.path.join(tmp, "calls.bcf")
    del batch
    tm = {}

    def step():
        cli.call_variants(cfg.scenario, paths, output=out_path, device=local_rank, afd_capacity=args.afd_capacity, timings=tm)
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    stages = {"read_s": 0.0, "call_s": 0.0, "write_s": 0.0, "setup_s": 0.0, "drain_s": 0.0, "drain_writer_s": 0.0, "drain_reader_close_s": 0.0, "drain_plans_close_s": 0.0}
    ingest.total_timings(reset=True)
    ingest.device_timings(reset=True)
    for _ in range(args.steps):
        step()
        for k in stages:
            stages[k] += tm.get(k, 0.0)
    native = {k: v / args.steps for k, v in ingest.total_timings().items()}
    # the device reader (csrc/vlr_decode.hip) keeps its own stage clock: inflate / split / decode kernels, copies, host side
    # (seconds per step for feed_inflate / split_scan / decode / copy_back / host_table / total / inflate_kernel; bytes, records and
    # serial walks per step for the counters)
    native_dev = {k: v / args.steps for k, v in ingest.device_timings().items()}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    calls_bytes = os.path.getsize(out_path)
    n_obs = tm["n_obs"]
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    if rank != 0:
        return None
    per = {k: v / args.steps for k, v in stages.items()}
    return {
        "metric": "candidate loci/sec end to end (observation BCF -> call variants -> calls BCF, whole node)", "value": n_loci * world * args.steps / elapsed,
        "unit": "loci/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "cli: tumor-normal 100x (config3 pileups), %d records/GPU in two observation BCFs (format v15, BGZF), AFD lists of %d entries, calls written as BCF" % (n_loci, args.afd_capacity),
                   "parallelism": "records sharded x%d, one process per GPU" % world, "host_threads": os.cpu_count(), "effective_cpus": effective_cpus()},
        "stages_s": per, "stages_note": "seconds per step inside the reader / evaluation / writer threads of cli.call_variants; the three overlap across chunks of %s records (%d chunks per step), so their sum exceeds ms_per_step" % (os.environ.get("VLR_CLI_CHUNK", "16384"), tm.get("chunks", 1)),
        "stage_rates": {"read_records_per_s": n_loci / per["read_s"], "read_uncompressed_GBps": None, "call_loci_per_s": n_loci / per["call_s"], "write_records_per_s": n_loci / per["write_s"]},
        "native_stage_seconds_per_step": native,
        "device_reader_seconds_per_step": native_dev,
        "process_cpu": {"cpu_seconds_per_step": cpu_s / args.steps, "busy_cpus": cpu_s / elapsed, "effective_cpus": effective_cpus(),
                        "note": "user + system time of this process over the timed steps / wall time: how much of the CPU quota the pipeline uses"},
        "native_stage_note": "inside read_s: inflate and parse_decode are summed over the sample files (which run side by side), files_wall is their wall time, merge + strings build the table; inside write_s: encode = record formatting, deflate_write = BGZF + file",
        "files": {"observation_bcf_bytes": obs_bytes, "calls_bcf_bytes": calls_bytes, "observations": int(n_obs), "observation_write_s_untimed": t_write_obs,
                  # the asymmetry between what is read and what is written: inputs deflated like htslib's default,
                  # calls at level 1 (written once, read once; VLR_BGZF_LEVEL selects another level)
                  "observation_bgzf_level": 6, "calls_bgzf_level": int(os.environ.get("VLR_BGZF_LEVEL") or 1)},
        "roofline": None, "cpu_baseline": None, "build_id": engine.build_id(),
    }


def bench_ingest(args, rank, world, local_rank, dev):
    """The device reader alone: observation BCFs on disk -> the SoA columns of vlr_batch in device memory
    (BGZF inflate, record split and v15 decode as kernels).  One step = one pass over the rank's files.  roofline: the inflate kernel,
    the dominant one — algorithmic bytes = compressed bytes read + inflated bytes written, over its HIP-event time."""
    import tempfile
    import torch
    import torch.distributed as dist
    from varlociraptor_amd import engine, ingest, synth
    n_loci = args.loci or 200_000
    cfg = synth.CONFIGS["config3"]()
    batch = generate("config3", n_loci, rank)
    tmp = tempfile.mkdtemp(prefix="vlr_ingest_%d_" % rank, dir=os.environ.get("VLR_BENCH_TMP", None))
    paths = []
    for s, name in enumerate(cfg.scenario.sample_names):
        paths.append(os.path.join(tmp, "%s.bcf" % name))
        ingest.write_observations(paths[-1], batch, s)
    obs_bytes = sum(os.path.getsize(p) for p in paths)
    n_obs = int(batch.n_obs)
    del batch
    chunk = int(os.environ.get("VLR_CLI_CHUNK", "0")) or 32768

    def one_pass(device):
        rd = ingest.ObsReader(paths, chunk_records=chunk, device=device)
        k = 0
        for b, _ in rd:
            k += b.n_loci
        rd.close()
        return k
    for _ in range(max(1, args.warmup)):
        assert one_pass(local_rank) == n_loci
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ingest.device_timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass(local_rank)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tm = ingest.device_timings()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    host = None
    if rank == 0 and not args.no_cpu_baseline:   # the host reader on the same files and cores (product code, not the oracle): one pass
        th = time.perf_counter()
        one_pass(None)
        host = n_loci / (time.perf_counter() - th)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    if rank != 0:
        return None
    algo = (tm["inflated_bytes"] + tm["compressed_bytes"]) / args.steps
    k_s = tm["inflate_kernel"] / args.steps
    return {
        "metric": "observation records/sec through the device reader (BGZF inflate + record split + v15 decode, whole node)", "value": n_loci * world * args.steps / elapsed,
        "unit": "records/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "ingest: tumor-normal 100x (config3 pileups), %d records/GPU in two observation BCFs (format v15, BGZF), requests of %d records" % (n_loci, chunk),
                   "parallelism": "files sharded x%d, one process per GPU" % world, "effective_cpus": effective_cpus()},
        "stages_s": {k: v / args.steps for k, v in tm.items() if k not in ("inflated_bytes", "compressed_bytes", "records", "serial_walks")},
        "files": {"observation_bcf_bytes": obs_bytes, "inflated_bytes": tm["inflated_bytes"] / args.steps, "observations": n_obs, "serial_walks": tm["serial_walks"]},
        "roofline": {"bound": "hbm", "achieved": algo / k_s / 1e9 if k_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (algo / k_s / 1e9 / HBM_PEAK_GBS) if k_s > 0 else None,
                     "traffic": None, "algorithmic_bytes_per_step": algo, "kernel_ms": k_s * 1e3,
                     "note": "vlr_inflate_kernel: one serial DEFLATE decoder per wave, fourteen waves per CU (4 KiB ring + tables in LDS), followed by the CRC32 kernel — bound by the issue rate and latencies of single waves, not by bandwidth (DESIGN 3e)"},
        "cpu_baseline": {"value": host, "unit": "records/s", "cores": effective_cpus(), "kind": "host reader (csrc/vlr_ingest.cpp: libdeflate + v15 decode on all cores; product code, not the oracle)", "sample": "one pass over the same files"} if host else None,
        "build_id": engine.build_id(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config3", choices=["config2", "config3", "config4", "config5", "realign", "cli", "ingest"])
    ap.add_argument("--loci", type=int, default=None, help="loci per GPU (default: the config's size); realign: pairs per GPU")
    ap.add_argument("--afd", dest="afd", action="store_true", default=True, help="(default) also time the step with the AFD lists")
    ap.add_argument("--no-afd", dest="afd", action="store_false", help="only the plain step")
    ap.add_argument("--dry-run", action="store_true", help="launcher and collective check without a GPU (gloo)")
    ap.add_argument("--afd-capacity", type=int, default=96)
    ap.add_argument("--cpu-loci", type=int, default=None, help="loci of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive and observation-BCF -> calls-BCF legs of the default line")
    ap.add_argument("--mode", default="exact", choices=["exact", "fast", "homopolymer"], help="realign workload: --pairhmm-mode of the reference (cli.rs:912-947)")
    ap.add_argument("--read-window", type=int, default=64, help="realign workload: realignment window; read windows are window..2*window bases (32: short reads, two pairs per wave)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if args.dry_run:
        assert int(os.environ.get("WORLD_SIZE", "1")) == args.gpus
        return dry_run(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))

    import torch
    import torch.distributed as dist
    from varlociraptor_amd import engine, synth
    from varlociraptor_amd.dist import all_gather_records, pack_full_records

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # VLR_BENCH_FORCE_DIST=1: go through RCCL even with a single rank (tests/test_gpu_cli_end_to_end.py: the collective path on a
    # one-GPU box) — the all-gather of the result records is then part of the timed step at world size 1 as well
    force_dist = os.environ.get("VLR_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d needs cuda:%d, %d device(s) visible (no CPU path)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    if args.workload in ("realign", "cli", "ingest"):
        line = {"realign": bench_realign, "cli": bench_cli, "ingest": bench_ingest}[args.workload](args, rank, world, local_rank, dev)
        if rank == 0:
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()
        return

    n_loci = args.loci or DEFAULT_LOCI[args.workload]
    cfg = synth.CONFIGS[args.workload]()
    t0 = time.time()
    batch = generate(args.workload, n_loci, rank)
    t_gen = time.time() - t0
    dbatch = engine.DeviceBatch(batch, dev)
    plan = engine.Plan(cfg.scenario, device=local_rank)
    # the caller knows its pileups: size the kernel's LDS coefficient area to the deepest locus of the batch
    plan.fit_max_obs(batch.obs_offset)   # LDS budget from the batch (vlr_plan_fit_max_obs: the deepest locus or the 16-workgroup budget)
    plan.reserve(batch.n_loci, with_afd=args.afd_capacity if args.afd else 0)  # vlr_batch_run then only enqueues work on the stream
    out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, dev)
    stream = torch.cuda.current_stream().cuda_stream
    n_total = n_loci * world

    def step(o=out):
        plan.call_device(dbatch, o, stream)
        if world > 1 or force_dist:
            # the FULL fixed-size record travels: posteriors, marginal, MAP VAFs, bias codes, best event, status
            return all_gather_records(pack_full_records(o), n_total, world)
        return None

    def timed(fn):
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1 or force_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    plan.work_counters(reset=True)
    elapsed = timed(step)
    # per-launch kernel duration from HIP events recorded on the launch stream inside vlr_batch_run
    # (the last launch's events; all launches process the same batch)
    last_ms = plan.last_kernel_ms()
    n_eval, n_terms = plan.work_counters()

    with_afd = None
    if args.afd:
        out_afd = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, dev, afd_capacity=args.afd_capacity)
        step(out_afd)
        el_afd = timed(lambda: step(out_afd))
        cnt = out_afd.afd_count.cpu().numpy()
        with_afd = {"value": n_total * args.steps / el_afd, "unit": "loci/s", "ms_per_step": el_afd / args.steps * 1e3,
                    "kernel_ms_both_launches": plan.last_kernel_ms(), "afd_capacity": args.afd_capacity,
                    "mean_afd_points_per_sample": float(np.minimum(cnt, args.afd_capacity).mean()),
                    "truncated_lists": int((cnt > args.afd_capacity).sum()),
                    "ratio_to_plain": (n_total * args.steps / el_afd) / (n_total * args.steps / elapsed)}
        del out_afd

    # like-for-like with the reference's per-record loop, which starts from host memory and always computes the AFD lists
    # (calling.rs:889-928): host buffers in and out through vlr_batch_run_host (page-locked arrays, staging pipelined with the
    # kernels), without and with AFD lists; and the whole process boundary (observation BCFs -> call variants -> calls BCF)
    pcie = e2e = None
    if world == 1 and not force_dist and args.workload == "config3" and not args.no_end_to_end:
        import types
        pb = engine.pin_batch(batch)
        from varlociraptor_amd.batch import CallResults
        r_plain = CallResults(batch.n_loci, plan.n_out, plan.n_samples, 0, alloc=engine.host_array)
        r_afd = CallResults(batch.n_loci, plan.n_out, plan.n_samples, args.afd_capacity, alloc=engine.host_array)
        plan.call_host(pb, results=r_plain)
        t0 = time.perf_counter(); plan.call_host(pb, results=r_plain); t_plain_locked = time.perf_counter() - t0
        # ... and into ordinary arrays: a third faster on the boxes of round 6 (tools/pcie_probe.py: the result copies into page-locked
        # memory are DMA transfers that queue up with the uploads of the next chunk; into pageable memory the runtime stages them)
        plan.call_host(pb)
        t0 = time.perf_counter(); plan.call_host(pb); t_plain = time.perf_counter() - t0
        plan.call_host(pb, afd_capacity=args.afd_capacity, results=r_afd)
        t0 = time.perf_counter(); plan.call_host(pb, afd_capacity=args.afd_capacity, results=r_afd); t_afd = time.perf_counter() - t0
        del pb, r_plain, r_afd
        pcie = {"value": batch.n_loci / t_plain, "page_locked_results": batch.n_loci / t_plain_locked, "with_afd": batch.n_loci / t_afd, "unit": "loci/s",
                "note": "vlr_batch_run_host: host arrays (page-locked) in, results out, one call each; value: results into ordinary arrays, "
                        "page_locked_results / with_afd (AFD lists of %d entries): results into page-locked arrays" % args.afd_capacity}
        try:
            a2 = types.SimpleNamespace(**vars(args))
            # the whole of BASELINE configs[2] (1 M records) per step since round 5: a run of the CLI has 0.07 s of fixed costs (opening and
            # indexing the files, the first request nothing overlaps with, the writer's last chunk, closing), which a 200 000-record step
            # counted as a quarter of its time (`--workload cli` defaults to such steps and reports both halves)
            a2.loci, a2.steps, a2.warmup = min(1_000_000, batch.n_loci), 2, 1
            cl = bench_cli(a2, 0, 1, local_rank, dev)
            e2e = {"value": cl["value"], "unit": "records/s", "records": a2.loci, "stages_s": cl["stages_s"], "native_stage_seconds": cl["native_stage_seconds_per_step"],
                   "device_reader_seconds": cl["device_reader_seconds_per_step"],
                   "host_threads_effective": cl["config"]["effective_cpus"], "files": cl["files"],
                   "note": "observation BCFs (format v15) -> cli.call_variants (native ingest, AFD lists, native calls writer) -> calls BCF; reader, evaluation and writer overlap; `python bench.py --workload cli` is the same with more steps"}
        except Exception as ex:  # the front door must not take the kernel line down
            e2e = {"value": None, "note": "end-to-end leg failed: %r" % (ex,)}

    res = out.to_host()
    line = None
    if rank == 0:
        alg_bytes = dbatch.algorithmic_bytes() + res.ln_posterior.nbytes + res.map_vaf.nbytes + res.status.nbytes
        achieved = alg_bytes / (last_ms * 1e-3) / 1e9
        # ---- parity + CPU baseline on a bounded sample of the same workload (rank 0, N = 1 only)
        parity = None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            from parity import compare
            cores = effective_cpus()
            # about 10 s of CPU work on the effective cores (863 / 97 k / 32 k loci/s for config3 / 2 / 5 on the box's 16 CPUs)
            per_core = {"config2": 50000, "config5": 15000}.get(args.workload, 600)
            n_cpu = args.cpu_loci or min(batch.n_loci, per_core * cores)
            sub = batch.select(np.arange(n_cpu))
            bounds = np.linspace(0, n_cpu, cores + 1).astype(int)
            oracle.lib()
            sc = cfg.scenario
            tc = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                parts = list(ex.map(lambda i: oracle.call(sc, sub, begin=int(bounds[i]), end=int(bounds[i + 1]), want_events=True), range(cores)))
            t_cpu = time.perf_counter() - tc
            from varlociraptor_amd.batch import CallResults
            ref = CallResults(n_cpu, plan.n_out, plan.n_samples)
            ref.event_ln_posterior = np.full((n_cpu, 1 + 2 * len(sc.event_names)), np.nan)
            for i, p in enumerate(parts):
                lo, hi = int(bounds[i]), int(bounds[i + 1])
                for f in ("ln_posterior", "map_vaf", "map_bias", "best_event", "status"):
                    getattr(ref, f)[lo:hi] = getattr(p, f)[lo:hi]
                ref.event_ln_posterior[lo:hi] = p.event_ln_posterior
            got = CallResults(n_cpu, plan.n_out, plan.n_samples)
            for f in ("ln_posterior", "map_vaf", "map_bias", "best_event", "status"):
                getattr(got, f)[:] = getattr(res, f)[:n_cpu]
            m = compare(got, ref)
            parity = {"n_checked": int(n_cpu), "max_abs_dposterior": m["max_dpost"], "max_abs_dmap_vaf": m["max_dvaf"],
                      "max_dln": m["max_dln"], "ln_entries_broken": m["n_ln_broken"], "loci_failing_ln": m["n_ln_fail"],
                      "ln_criterion": "|d ln posterior| <= 1e-6 max(1, |ln posterior|) where both finite, -inf only where the oracle has -inf (PHRED f32 of the calls record equal to >= 6 digits)",
                      "frac_within_1e-6": m["frac_within"], "exact_event_ties": m["n_ties"], "vs": "CPU restatement of the reference (oracle/)"}
            cpu = {"value": n_cpu / t_cpu, "unit": "loci/s", "cores": cores, "kind": "port",
                   "sample": "first %d loci of the same batch, %d threads x contiguous shards, %.1f s" % (n_cpu, cores, t_cpu),
                   "note": "fidelity oracle (reference operation order, log-space transcendentals, caches), not a tuned CPU implementation"}
            # the same algorithm with the pileup likelihood in affine product form, no allocation in the term
            # loop, AVX2/FMA build: the CPU path somebody tried to make fast (oracle/vlr_oracle.cpp, Ctx::tuned)
            per_core_t = {"config2": 250000, "config5": 40000}.get(args.workload, 15000)
            n_t = min(batch.n_loci, per_core_t * cores)
            sub_t = batch.select(np.arange(n_t)) if n_t != n_cpu else sub
            bounds_t = np.linspace(0, n_t, cores + 1).astype(int)
            oracle.lib_tuned()
            tt = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                parts_t = list(ex.map(lambda i: oracle.call(sc, sub_t, begin=int(bounds_t[i]), end=int(bounds_t[i + 1]), tuned=True), range(cores)))
            t_tuned = time.perf_counter() - tt
            dev_t = 0.0
            for i, p in enumerate(parts_t):
                lo, hi = int(bounds_t[i]), min(int(bounds_t[i + 1]), n_cpu)
                if hi > lo:
                    with np.errstate(invalid="ignore"):
                        d = np.abs(np.exp(p.ln_posterior[lo:hi]) - np.exp(ref.ln_posterior[lo:hi]))
                    dev_t = max(dev_t, float(np.nanmax(d)) if d.size else 0.0)
            cpu["tuned"] = {"value": n_t / t_tuned, "unit": "loci/s", "cores": cores, "kind": "tuned",
                            "sample": "first %d loci of the same batch, %d threads x contiguous shards, %.1f s" % (n_t, cores, t_tuned),
                            "max_abs_dposterior_vs_port": dev_t,
                            "note": "affine product form of the pileup likelihood (one log per pileup evaluation), -O3 -march=x86-64-v3; tree walk, prior, integrator and caches shared with the port"}
            # the reference itself, if somebody put its binary on the box (never seen so far: null)
            try:
                cpu["reference_binary"] = reference_binary_baseline(cfg, sub, n_cpu, cores)
            except Exception as ex:  # a probe must not take the bench line down
                cpu["reference_binary"] = {"value": None, "note": "probe failed: %r" % (ex,)}
        # posteriors must be normalised at full size (size-independent property)
        ps = np.exp(res.ln_posterior)
        ok = (res.status & 0xF) == 0
        norm_err = float(np.abs(ps[ok].sum(axis=1) - 1.0).max()) if ok.any() else None
        terms_per_launch = n_terms / max(1, args.steps)
        tflops = terms_per_launch * FLOP_PER_TERM / (last_ms * 1e-3) / 1e12
        vstamp = valu_for(args.workload, engine.build_id(), n_loci, last_ms) or {}
        line = {
            "metric": "candidate loci/sec (whole node)", "value": n_total * args.steps / elapsed, "unit": "loci/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s, %d loci/GPU, mean depth %.0fx/sample, %s%s; variant types %s; truth classes %s" % (
                           args.workload, cfg.name, n_loci, cfg.depth, "%d samples" % plan.n_samples,
                           (", purity %.2f (contamination %.2f)" % (cfg.purity, 1.0 - cfg.purity)) if cfg.purity is not None else "",
                           "/".join("%s %.0f%%" % ({0: "SNV", 1: "MNV", 2: "indel", 3: "SV/BND", 4: "other"}.get(k, str(k)), 100 * v) for k, v in cfg.type_mix.items()),
                           "/".join("%s %.0f%%" % (c[0], 100 * c[1]) for c in cfg.classes)),
                       "scenario_events": cfg.scenario.event_names,
                       "parallelism": "loci sharded x%d, all-gather of the full result records (posteriors, marginal, MAP VAFs, bias codes, best event, status)" % world,
                       "gen_seconds": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic_for(args.workload, n_loci, engine.build_id()),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": last_ms,
                         "note": "path is f64-VALU bound, not bandwidth bound (SURVEY §8d): valu_busy = VALU issue cycles / SIMD cycles of the launch, f64_share = f64 FMA+MUL+ADD / VALU instructions (PMC stamp of this build under profiles/, null without one), f64_flops_frac = 3 flops per observation term / 78.6 TFLOP/s",
                         "valu_busy": vstamp.get("valu_busy"), "f64_share": vstamp.get("f64_share"), "f64_flops_frac": tflops / F64_VALU_PEAK_TFLOPS,
                         "valu_insts_per_locus": vstamp.get("valu_insts_per_locus"), "salu_insts_per_locus": vstamp.get("salu_insts_per_locus"),
                         "lane_utilisation": vstamp.get("lane_utilisation"), "valu_stamp": vstamp.get("stamp"),
                         "valu": {"pileup_evals_per_launch": n_eval // max(1, args.steps), "obs_terms_per_launch": int(terms_per_launch),
                                  "terms_per_s": terms_per_launch / (last_ms * 1e-3), "flop_per_term": FLOP_PER_TERM,
                                  "achieved_tflops": tflops, "peak_tflops": F64_VALU_PEAK_TFLOPS, "frac": tflops / F64_VALU_PEAK_TFLOPS}},
            "with_afd": with_afd, "pcie_inclusive": pcie, "end_to_end": e2e,
            "cpu_baseline": cpu if world == 1 else {"value": None, "note": "N=1 only (rank 0 times the CPU restatement on a bounded sample of the same workload)"},
            "parity": parity if world == 1 else {"value": None, "note": "N=1 only (the sample compared with the oracle is taken at N=1; the N>1 path is covered by tests/test_distributed_cpu.py and tests/test_gpu_node.py)"},
            "posterior_normalisation_max_err": norm_err,
            "n1_only": None if world == 1 else "cpu_baseline, parity, pcie_inclusive and end_to_end are measured at N=1 only (rank 0, a bounded sample): null here by design; this line carries the whole-job rate, the per-rank kernel roofline and the collective",
            "status_counts": {str(k): int(v) for k, v in zip(*np.unique(res.status, return_counts=True))},
            "collective": ("rccl all_gather_into_tensor, world size %d" % world) if (world > 1 or force_dist) else None,
            "build_id": engine.build_id(),
        }
        print(json.dumps(line))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
