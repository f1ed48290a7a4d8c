"""Multi-GPU plumbing: static contiguous sharding of loci + one all-gather of fixed-size result records.

Loci are independent given the plan, so there is no data-path collective; the only
exchange is the reassembly of per-locus result records in input order for emission (north star:
"RCCL all-gather over xGMI only to reassemble posteriors for VCF emission").  Works with any
torch.distributed backend ("nccl" = RCCL on ROCm for GPUs, "gloo" for the CPU tests).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_range(n_loci: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of ceil(n/world) loci per rank, in input order (last ranks may be short/empty)."""
    per = -(-n_loci // world)
    lo = min(n_loci, rank * per)
    hi = min(n_loci, lo + per)
    return lo, hi


def pack_records(ln_posterior, map_vaf, status):
    """[n, n_out + S + 1] float64 record per locus (status carried as a float64 bit-exact small integer)."""
    import torch
    return torch.cat([ln_posterior, map_vaf, status.to(torch.float64).unsqueeze(1)], dim=1).contiguous()


def pack_full_records(out):
    """[n, n_out + 1 + S + 6 + 2] float64 record per locus from an engine.DeviceResults: ln posteriors, ln marginal, MAP
    VAFs, the six bias codes, best event and status (small integers travel bit-exactly as float64) — everything the calls
    writer needs besides the ragged AFD lists."""
    import torch
    f = torch.float64
    return torch.cat([out.ln_posterior, out.ln_marginal.unsqueeze(1), out.map_vaf, out.map_bias.to(f),
                      out.best_event.to(f).unsqueeze(1), out.status.to(f).unsqueeze(1)], dim=1).contiguous()


def unpack_full_records(rec, n_out: int, n_samples: int):
    """Inverse of pack_full_records on a host array: dict of numpy arrays."""
    rec = np.asarray(rec)
    o = 0
    res = {"ln_posterior": rec[:, o:o + n_out]}; o += n_out
    res["ln_marginal"] = rec[:, o]; o += 1
    res["map_vaf"] = rec[:, o:o + n_samples]; o += n_samples
    res["map_bias"] = rec[:, o:o + 6].astype(np.uint8); o += 6
    res["best_event"] = rec[:, o].astype(np.int32); o += 1
    res["status"] = rec[:, o].astype(np.int64).astype(np.uint32)
    return res


def all_gather_records(records, n_total: int, world: int):
    """All-gather equally sized per-rank record blocks (padded to ceil(n/world) rows) and trim to n_total rows."""
    import torch
    import torch.distributed as dist
    per = -(-n_total // world)
    if records.shape[0] < per:
        pad = torch.full((per - records.shape[0], records.shape[1]), float("nan"), dtype=records.dtype, device=records.device)
        records = torch.cat([records, pad], dim=0)
    out = torch.empty((per * world, records.shape[1]), dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(out, records.contiguous())
    return out[:n_total]


def gather_call_results(local, lo: int, hi: int, n_total: int, n_out: int, n_samples: int, afd_capacity: int = 0):
    """Reassemble CallResults of all ranks in input order (every rank gets the full result).

    `local` holds the loci [lo, hi) of this rank (rows 0..hi-lo).  Fixed-size fields travel in one all-gather of f64
    records; the variable-length AFD lists travel as counts (inside the record) plus a second all-gather of the packed
    (vaf, ln prob) pairs, padded to the longest rank."""
    import torch
    import torch.distributed as dist
    from .batch import CallResults
    world = dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = hi - lo
    S = n_samples
    cols = [local.ln_posterior[:n], local.ln_marginal[:n, None], local.map_vaf[:n], local.map_bias[:n].astype(np.float64),
            local.best_event[:n, None].astype(np.float64), local.status[:n, None].astype(np.float64)]
    if afd_capacity:
        cols.append(np.minimum(local.afd_count[:n], afd_capacity).astype(np.float64))
    rec = torch.from_numpy(np.ascontiguousarray(np.concatenate(cols, axis=1))).to(dev)
    full = all_gather_records(rec, n_total, world).cpu().numpy()
    out = CallResults(n_total, n_out, S, afd_capacity)
    o = 0
    out.ln_posterior[:] = full[:, o:o + n_out]; o += n_out
    out.ln_marginal[:] = full[:, o]; o += 1
    out.map_vaf[:] = full[:, o:o + S]; o += S
    out.map_bias[:] = full[:, o:o + 6].astype(np.uint8); o += 6
    out.best_event[:] = full[:, o].astype(np.int32); o += 1
    out.status[:] = full[:, o].astype(np.uint32); o += 1
    if afd_capacity:
        out.afd_count[:] = full[:, o:o + S].astype(np.int32)
        cnt = np.minimum(local.afd_count[:n], afd_capacity)
        m = np.arange(afd_capacity)[None, None, :] < cnt[:, :, None]
        packed = np.stack([local.afd_vaf[:n][m], local.afd_lnprob[:n][m]], axis=1) if n else np.zeros((0, 2))
        sizes = torch.tensor([packed.shape[0]], dtype=torch.int64, device=dev)
        all_sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        all_sizes = [int(x.item()) for x in all_sizes]
        mx = max(max(all_sizes), 1)
        buf = torch.zeros((mx, 2), dtype=torch.float64, device=dev)
        if packed.shape[0]:
            buf[:packed.shape[0]] = torch.from_numpy(np.ascontiguousarray(packed)).to(dev)
        gathered = torch.empty((mx * world, 2), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(gathered, buf)
        gathered = gathered.cpu().numpy()
        per = -(-n_total // world)
        for r in range(world):
            l0, l1 = min(n_total, r * per), min(n_total, r * per + per)
            if l1 <= l0:
                continue
            c = out.afd_count[l0:l1]
            mm = np.arange(afd_capacity)[None, None, :] < c[:, :, None]
            block = gathered[r * mx:r * mx + all_sizes[r]]
            out.afd_vaf[l0:l1][mm] = block[:, 0]
            out.afd_lnprob[l0:l1][mm] = block[:, 1]
    return out
