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
