"""N > 1 path on CPU: world_size 2 over gloo.  Each rank evaluates its contiguous shard (here with the
oracle standing in for the device engine — the sharding/all-gather plumbing is what is under test) and
the all-gathered records must equal the single-process result in input order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_loci, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from varlociraptor_amd import synth
    from types import SimpleNamespace
    from varlociraptor_amd.dist import all_gather_records, pack_full_records, shard_range
    cfg = synth.config2()
    batch = synth.generate(cfg, n_loci)
    lo, hi = shard_range(n_loci, rank, world)
    res = oracle.call(cfg.scenario, batch, begin=lo, end=hi)
    # the record bench.py and the CLI send: every fixed-size field of the result (stand-in for engine.DeviceResults)
    shard = SimpleNamespace(ln_posterior=torch.from_numpy(res.ln_posterior[lo:hi]), ln_marginal=torch.from_numpy(res.ln_marginal[lo:hi]),
                            map_vaf=torch.from_numpy(res.map_vaf[lo:hi]), map_bias=torch.from_numpy(res.map_bias[lo:hi]),
                            best_event=torch.from_numpy(res.best_event[lo:hi]), status=torch.from_numpy(res.status[lo:hi].astype(np.int64)))
    full = all_gather_records(pack_full_records(shard), n_loci, world)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _rows_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from varlociraptor_amd.ingest import _gather_rows
    out = _gather_rows(np.arange(2 * 7, dtype=np.int64).reshape(2, 7) + 100 * rank)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_rows_of_the_sharded_reader_are_gathered_in_rank_order(world):
    """The one exchange of the sharded front door (ingest.ObsReader(shard=...)): every rank's [n_files][7] int64 rows, in rank order,
    on every rank — over gloo as over RCCL (a flat tensor: gloo's all_gather_into_tensor takes no higher-rank output)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29620 + world
    ps = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=300) for _ in range(world))
    [p.join(timeout=120) for p in ps]
    want = np.stack([np.arange(14, dtype=np.int64).reshape(2, 7) + 100 * r for r in range(world)])
    for r in range(world):
        assert got[r].shape == (world, 2, 7) and np.array_equal(got[r], want)


@pytest.mark.parametrize("n_loci", [37, 64])
def test_two_rank_gather_matches_single_process(n_loci):
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from varlociraptor_amd import synth
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n_loci
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_loci, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = synth.config2()
    batch = synth.generate(cfg, n_loci)
    ref = oracle.call(cfg.scenario, batch)
    n_out = cfg.scenario.n_out
    from varlociraptor_amd.dist import unpack_full_records
    assert full.shape == (n_loci, n_out + 1 + 1 + 6 + 2)
    got = unpack_full_records(full, n_out, 1)
    assert np.array_equal(got["ln_posterior"], ref.ln_posterior)
    assert np.array_equal(got["ln_marginal"], ref.ln_marginal, equal_nan=True)
    assert np.array_equal(got["map_vaf"], ref.map_vaf, equal_nan=True)
    assert np.array_equal(got["map_bias"], ref.map_bias)
    assert np.array_equal(got["best_event"], ref.best_event)
    assert np.array_equal(got["status"], ref.status)


def _worker_full(rank, world, port, n_loci, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from varlociraptor_amd import synth
    from varlociraptor_amd.batch import CallResults
    from varlociraptor_amd.dist import gather_call_results, shard_range
    cfg = synth.config2()
    batch = synth.generate(cfg, n_loci)
    lo, hi = shard_range(n_loci, rank, world)
    res = oracle.call(cfg.scenario, batch.select(range(lo, hi)), afd_capacity=96) if hi > lo else CallResults(0, cfg.scenario.n_out, 1, 96)
    full = gather_call_results(res, lo, hi, n_loci, cfg.scenario.n_out, 1, afd_capacity=96)
    if rank == 1:  # every rank holds the full result
        q.put({k: getattr(full, k) for k in ("ln_posterior", "ln_marginal", "map_vaf", "map_bias", "best_event", "status", "afd_count", "afd_vaf", "afd_lnprob")})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_loci,world", [(41, 2), (5, 3)])
def test_gather_call_results_including_afd_lists(n_loci, world):
    """Fixed-size fields in one all-gather, the ragged AFD lists as counts + packed pairs in a second one; ranks with an
    empty shard take part (5 loci on 3 ranks)."""
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from varlociraptor_amd import synth
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 500) + n_loci
    procs = [ctx.Process(target=_worker_full, args=(r, world, port, n_loci, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = synth.config2()
    ref = oracle.call(cfg.scenario, synth.generate(cfg, n_loci), afd_capacity=96)
    for k in ("ln_posterior", "ln_marginal", "map_vaf", "map_bias", "best_event", "status", "afd_count"):
        assert np.array_equal(full[k], getattr(ref, k), equal_nan=True), k
    m = np.arange(96)[None, None, :] < ref.afd_count[:, :, None]
    assert np.array_equal(full["afd_vaf"][m], ref.afd_vaf[m]) and np.array_equal(full["afd_lnprob"][m], ref.afd_lnprob[m])


@pytest.mark.parametrize("n", [2, 8])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` (the driver's command shape) must work outside torchrun.  Dry run over gloo:
    bench.py re-executes itself under torch.distributed.run with N ranks on 127.0.0.1, the ranks all-gather their record
    blocks in input order and rank 0 prints one JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "0", "--loci", "777", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    # (8 = the node the north star names: the launcher has been seen with eight ranks before an 8-GPU node ever appears)
    assert line["n_gpus"] == n and line["gathered_in_order"] is True and line["steps"] == 2


def test_bench_without_a_device_fails_loudly():
    """No GPU here: the real (non dry-run) bench must not fall back to anything."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--loci", "10", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)
