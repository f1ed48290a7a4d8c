"""PileupBatch: host-side SoA container for vlr_batch (numpy), plus result buffers.

Layout = include/vlr.h `vlr_batch`: pileup p = locus*S + sample covers observation rows
[obs_offset[p], obs_offset[p+1]) of every column (the AoS Vec<ReadObservation> per sample of
variants/evidence/observations/pileup.rs:5-11 turned into columns for coalesced device loads).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import abi


class PileupBatch:
    def __init__(self, n_samples: int, obs_offset: np.ndarray, columns: Dict[str, np.ndarray],
                 locus: Dict[str, np.ndarray]):
        self.n_samples = int(n_samples)
        oo = np.asarray(obs_offset)
        if oo.size and int(oo.max()) > 0xFFFFFFFF:
            raise ValueError("batch of %d observations exceeds the 32-bit observation offsets of vlr_batch: split it" % int(oo.max()))
        self.obs_offset = np.ascontiguousarray(oo, dtype=np.uint32)
        assert (len(self.obs_offset) - 1) % self.n_samples == 0
        self.n_loci = (len(self.obs_offset) - 1) // self.n_samples
        self.n_obs = int(self.obs_offset[-1])
        self.columns = {}
        for name, dt in abi.OBS_COLUMNS:
            a = columns.get(name)
            if a is None:
                if name in ("prob_hp_artifact", "prob_hp_variant"):
                    a = np.full(self.n_obs, np.nan, dtype=np.float32)
                else:
                    raise KeyError(name)
            a = np.ascontiguousarray(a, dtype=dt)
            assert a.shape == (self.n_obs,), (name, a.shape, self.n_obs)
            self.columns[name] = a
        self.locus = {}
        defaults = {"locus_flags": abi.BIAS_ALL, "variant_type": abi.VT_SNV, "ref_base": ord("A"), "alt_base": ord("C")}
        for name, dt in abi.LOCUS_COLUMNS:
            a = locus.get(name)
            if a is None:
                a = np.full(self.n_loci, defaults[name], dtype=dt)
            a = np.ascontiguousarray(a, dtype=dt)
            assert a.shape == (self.n_loci,)
            self.locus[name] = a

    # ---- views
    def pileup_slice(self, locus: int, sample: int) -> slice:
        p = locus * self.n_samples + sample
        return slice(int(self.obs_offset[p]), int(self.obs_offset[p + 1]))

    def depth(self) -> np.ndarray:
        return np.diff(self.obs_offset.astype(np.int64)).reshape(self.n_loci, self.n_samples)

    def select(self, loci: Sequence[int]) -> "PileupBatch":
        """Sub-batch with the given loci (copy)."""
        loci = np.asarray(loci, dtype=np.int64)
        S = self.n_samples
        off = self.obs_offset.astype(np.int64)
        starts = off[loci * S]
        totals = off[loci * S + S] - starts
        # rows of a locus are contiguous: one arange over the total, shifted per locus (no Python loop over loci)
        first = np.concatenate(([0], np.cumsum(totals)))[:-1]
        idx = np.repeat(starts - first, totals) + np.arange(int(totals.sum()), dtype=np.int64)
        lens = np.diff(off).reshape(self.n_loci, S)[loci]
        new_off = np.concatenate(([0], np.cumsum(lens.ravel())))
        cols = {k: v[idx] for k, v in self.columns.items()}
        loc = {k: v[loci] for k, v in self.locus.items()}
        return PileupBatch(S, np.asarray(new_off, np.uint32), cols, loc)

    @staticmethod
    def concat(batches: List["PileupBatch"]) -> "PileupBatch":
        S = batches[0].n_samples
        offs = [np.zeros(1, np.int64)]
        base = 0
        for b in batches:
            offs.append(b.obs_offset[1:].astype(np.int64) + base)
            base += b.n_obs
        cols = {k: np.concatenate([b.columns[k] for b in batches]) for k, _ in abi.OBS_COLUMNS}
        loc = {k: np.concatenate([b.locus[k] for b in batches]) for k, _ in abi.LOCUS_COLUMNS}
        return PileupBatch(S, np.concatenate(offs).astype(np.uint32), cols, loc)

    # ---- C struct with HOST pointers (for vlr_batch_run_host and the oracle)
    def as_struct(self) -> abi.Batch:
        b = abi.Batch()
        b.n_loci, b.n_samples, b.n_obs = self.n_loci, self.n_samples, self.n_obs
        b.obs_offset = self.obs_offset.ctypes.data
        for name, _ in abi.OBS_COLUMNS:
            setattr(b, name, self.columns[name].ctypes.data)
        for name, _ in abi.LOCUS_COLUMNS:
            setattr(b, name, self.locus[name].ctypes.data)
        return b

    def algorithmic_bytes(self) -> int:
        """Bytes one pass must read: all observation columns + offsets + per-locus columns."""
        per_obs = sum(np.dtype(dt).itemsize for _, dt in abi.OBS_COLUMNS)
        return self.n_obs * per_obs + self.obs_offset.nbytes + sum(v.nbytes for v in self.locus.values())


class CallResults:
    """Host result buffers of include/vlr.h `vlr_results`."""

    def __init__(self, n_loci: int, n_out: int, n_samples: int, afd_capacity: int = 0, alloc=None, afd_text_capacity: int = 0):
        """`alloc(shape, dtype)`: allocator of the arrays (engine.host_array: page-locked memory, so that vlr_batch_run_host copies the
        results back by direct DMA — the AFD lists are gigabytes for a million loci); default: ordinary initialised numpy arrays.
        `afd_text_capacity` > 0: the FORMAT/AFD text of every list is formatted on the device into `afd_text` / `afd_text_span`
        (vlr_results.afd_text) and the lists themselves only come down when a list could not be formatted."""
        self.n_loci, self.n_out, self.n_samples, self.afd_capacity = n_loci, n_out, n_samples, afd_capacity
        self.afd_text = self.afd_text_span = None
        custom = alloc is not None
        if alloc is None:
            def alloc(shape, dtype, fill=0):
                return np.full(shape, fill, dtype)
        else:
            raw = alloc

            def alloc(shape, dtype, fill=0):
                a = raw(shape, dtype)
                if fill is not None and np.prod(shape) < (1 << 24):   # (the large AFD arrays are overwritten up to afd_count: not touched here)
                    a[...] = fill
                return a
        self.ln_posterior = alloc((n_loci, n_out), np.float64, np.nan)
        self.ln_marginal = alloc((n_loci,), np.float64, np.nan)
        self.map_vaf = alloc((n_loci, n_samples), np.float64, np.nan)
        self.map_bias = alloc((n_loci, abi.N_BIAS), np.uint8)
        self.best_event = alloc((n_loci,), np.int32, -1)
        self.status = alloc((n_loci,), np.uint32)
        if afd_capacity > 0:
            self.afd_count = alloc((n_loci, n_samples), np.int32)
            self.afd_vaf = alloc((n_loci, n_samples, afd_capacity), np.float64, None if custom else 0)
            self.afd_lnprob = alloc((n_loci, n_samples, afd_capacity), np.float64, None if custom else 0)
            if afd_text_capacity > 0:
                self.afd_text = alloc((int(afd_text_capacity),), np.uint8, None if custom else 0)
                self.afd_text_span = alloc((n_loci, n_samples, 2), np.uint32)
        else:
            self.afd_count = self.afd_vaf = self.afd_lnprob = None

    def afd_strings(self, locus: int, sample: int):
        """The device-formatted AFD text of one list, or None when the list was left to the arrays (or no text was requested)."""
        if self.afd_text is None:
            return None
        off, n = (int(x) for x in self.afd_text_span[locus, sample])
        return None if n == 0xffffffff else bytes(self.afd_text[off:off + n]).decode()

    def as_struct(self) -> abi.Results:
        r = abi.Results()
        r.n_loci, r.n_out, r.n_samples = self.n_loci, self.n_out, self.n_samples
        r.ln_posterior = self.ln_posterior.ctypes.data
        r.ln_marginal = self.ln_marginal.ctypes.data
        r.map_vaf = self.map_vaf.ctypes.data
        r.map_bias = self.map_bias.ctypes.data
        r.best_event = self.best_event.ctypes.data
        r.status = self.status.ctypes.data
        r.afd_capacity = self.afd_capacity
        if self.afd_capacity > 0:
            r.afd_count = self.afd_count.ctypes.data
            r.afd_vaf = self.afd_vaf.ctypes.data
            r.afd_lnprob = self.afd_lnprob.ctypes.data
            if self.afd_text is not None:
                r.afd_text = self.afd_text.ctypes.data
                r.afd_text_capacity = int(self.afd_text.size)
                r.afd_text_span = self.afd_text_span.ctypes.data
        return r

    def phred(self) -> np.ndarray:
        """PROB_* as written to the BCF: PHREDProb::from(prob).abs() as f32 (calling/variants/mod.rs:463-466)."""
        with np.errstate(invalid="ignore"):
            return np.abs(-10.0 / np.log(10.0) * self.ln_posterior).astype(np.float32)
