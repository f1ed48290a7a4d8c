"""Native ingest / emission at the process boundary (csrc/vlr_ingest.cpp behind include/vlr.h): observation BCF/VCF files
-> PileupBatch whose columns are views of the table's page-locked arrays; results -> calls BCF/VCF.

Replaces, on the product path, the Python decoder (obsfmt.py), BCF reader/writer (bcfio.py) and record formatter
(callsfmt.py), which stay as the independent restatement the tests compare against (reference: calling.rs:306-339,
preprocessing/mod.rs:818-1038, calling/variants/mod.rs:178-600).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import abi, engine
from .batch import CallResults, PileupBatch


class ObsSites(C.Structure):
    _fields_ = [
        ("n_loci", C.c_int64), ("n_contigs", C.c_int32), ("_pad", C.c_int32),
        ("contig_names", C.POINTER(C.c_char_p)), ("contig", C.c_void_p), ("pos", C.c_void_p), ("strings", C.c_void_p),
        ("id_offset", C.c_void_p), ("ref_offset", C.c_void_p), ("alt_offset", C.c_void_p),
        ("group_representative", C.c_void_p), ("heterozygosity_ln", C.c_void_p), ("somatic_effective_mutation_rate_ln", C.c_void_p),
        ("third_allele_evidence", C.c_void_p), ("imprecise", C.c_void_p), ("group_key", C.c_void_p),
    ]


def _lib():
    L = engine.lib()
    if not getattr(L, "_ingest_typed", False):
        L.vlr_obs_read.restype = C.c_int
        L.vlr_obs_read.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.vlr_obs_table_free.restype = None
        L.vlr_obs_table_free.argtypes = [C.c_void_p]
        L.vlr_obs_table_batch.restype = C.c_int
        L.vlr_obs_table_batch.argtypes = [C.c_void_p, C.POINTER(abi.Batch)]
        L.vlr_obs_table_sites.restype = C.c_int
        L.vlr_obs_table_sites.argtypes = [C.c_void_p, C.POINTER(ObsSites)]
        L.vlr_obs_write.restype = C.c_int
        L.vlr_obs_write.argtypes = [C.c_char_p, C.POINTER(abi.Batch), C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.vlr_calls_write.restype = C.c_int
        L.vlr_calls_write.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.POINTER(abi.Results), C.POINTER(C.c_char_p), C.c_int]
        L._ingest_typed = True
    return L


def _check(rc):
    if rc != 0:
        raise engine.EngineError(rc, (engine.lib().vlr_last_error() or b"").decode())


def _view(ptr, n, dtype):
    dt = np.dtype(dtype)
    if n == 0 or not ptr:
        return np.zeros(0, dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n)


class Sites:
    """Per-locus site data of a table: arrays (contig index, position, group representative, prior overrides) and strings on
    demand.  Indexing gives the (chrom, pos, ref, alt) tuple the Python formatter uses."""

    def __init__(self, table: "ObsTable"):
        s = ObsSites()
        _check(_lib().vlr_obs_table_sites(table.handle, C.byref(s)))
        self._table = table
        n = int(s.n_loci)
        self.n_loci = n
        self.contig_names = [s.contig_names[i].decode() for i in range(s.n_contigs)]
        self.contig = _view(s.contig, n, np.int32)
        self.pos = _view(s.pos, n, np.int64)
        self.group_representative = _view(s.group_representative, n, np.int64)
        self.heterozygosity_ln = _view(s.heterozygosity_ln, n, np.float64)
        self.somatic_effective_mutation_rate_ln = _view(s.somatic_effective_mutation_rate_ln, n, np.float64)
        self.imprecise = _view(s.imprecise, n, np.uint8)
        self.group_key = _view(s.group_key, n, np.uint64)
        self._strings = s.strings
        self._id, self._ref, self._alt = (_view(p, n, np.uint64) for p in (s.id_offset, s.ref_offset, s.alt_offset))
        self.third_allele_evidence = _view(s.third_allele_evidence, table.n_obs, np.int32)

    def _str(self, off) -> str:
        return C.string_at(self._strings + int(off)).decode()

    def chrom(self, l: int) -> str:
        c = int(self.contig[l])
        return self.contig_names[c] if 0 <= c < len(self.contig_names) else str(c)

    def __len__(self):
        return self.n_loci

    def __getitem__(self, l: int) -> Tuple[str, int, str, str]:
        return (self.chrom(l), int(self.pos[l]), self._str(self._ref[l]), self._str(self._alt[l]))

    def record_id(self, l: int) -> str:
        return self._str(self._id[l])


class ObsTable:
    def __init__(self, handle):
        self.handle = handle
        b = abi.Batch()
        _check(_lib().vlr_obs_table_batch(handle, C.byref(b)))
        self.n_loci, self.n_samples, self.n_obs = int(b.n_loci), int(b.n_samples), int(b.n_obs)
        self._struct = b
        self.on_device = False   # set by a device reader: device_batch() holds the same batch in device memory

    def batch(self) -> PileupBatch:
        b = self._struct
        cols = {name: _view(getattr(b, name), self.n_obs, dt) for name, dt in abi.OBS_COLUMNS}
        locus = {name: _view(getattr(b, name), self.n_loci, dt) for name, dt in abi.LOCUS_COLUMNS}
        pb = PileupBatch(self.n_samples, _view(b.obs_offset, self.n_loci * self.n_samples + 1, np.uint32), cols, locus)
        pb._table = self  # the views live as long as the table
        return pb

    def device_batch(self) -> "abi.Batch":
        """vlr_obs_table_device_batch: the same batch in device memory (tables of a device reader only) — what vlr_batch_run takes."""
        b = abi.Batch()
        L = _lib()
        L.vlr_obs_table_device_batch.restype = C.c_int
        L.vlr_obs_table_device_batch.argtypes = [C.c_void_p, C.POINTER(abi.Batch)]
        _check(L.vlr_obs_table_device_batch(self.handle, C.byref(b)))
        return b

    def summaries(self):
        """vlr_obs_table_summaries: (the calls writer formats this table from device summaries, pileups left to the columns)."""
        L = engine.lib()
        L.vlr_obs_table_summaries.restype = C.c_int
        L.vlr_obs_table_summaries.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        n = C.c_int64(0)
        on = L.vlr_obs_table_summaries(self.handle, C.byref(n))
        return bool(on), int(n.value)

    def fetch_columns(self):
        """vlr_obs_table_fetch_columns: bring the observation columns of a device reader's table down to the host (no-op when they are)."""
        L = _lib()
        L.vlr_obs_table_fetch_columns.restype = C.c_int
        L.vlr_obs_table_fetch_columns.argtypes = [C.c_void_p]
        _check(L.vlr_obs_table_fetch_columns(self.handle))

    def close(self):
        if self.handle:
            _lib().vlr_obs_table_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_observations(paths: Sequence[str], omit_bias_mask: int = 0, threads: int = 0):
    """vlr_obs_read: one observation file per sample (sample-index order) -> (PileupBatch, Sites).  batch.extra carries what the
    Python decoder's does (third_allele_evidence, haplotype groups as representatives, prior overrides as arrays)."""
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    h = C.c_void_p()
    _check(_lib().vlr_obs_read(len(paths), arr, int(omit_bias_mask), int(threads), C.byref(h)))
    return _wrap_table(h)


def _wrap_table(h):
    table = ObsTable(h)
    batch = table.batch()
    sites = Sites(table)
    batch.extra = {"third_allele_evidence": sites.third_allele_evidence, "group_representative": sites.group_representative, "group_key": sites.group_key,
                   "prior_het_ln": sites.heterozygosity_ln, "prior_som_ln": sites.somatic_effective_mutation_rate_ln, "native_table": table}
    return batch, sites


def total_timings(reset: bool = False) -> dict:
    """Stage times summed over all streaming reader / writer calls since the last reset (vlr_ingest_total_timings)."""
    a = (C.c_double * 16)()
    L = _lib()
    L.vlr_ingest_total_timings.restype = None
    L.vlr_ingest_total_timings.argtypes = [C.POINTER(C.c_double), C.c_int]
    L.vlr_ingest_total_timings(a, int(reset))
    k = ["file_read", "inflate", "parse_decode", "files_wall", "merge", "strings", "read_total", None, "encode", "deflate_write", "write_total"]
    return {n: a[i] for i, n in enumerate(k) if n}


def bgzf_inflate(data: bytes, device: int = 0) -> bytes:
    """vlr_bgzf_inflate: a sequence of BGZF members inflated by the device kernel (csrc/vlr_inflate.hip)."""
    L = _lib()
    L.vlr_bgzf_inflate.restype = C.c_int
    L.vlr_bgzf_inflate.argtypes = [C.c_int, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    n = C.c_int64(0)
    # first call sizes the output (ISIZE sum), second fills it
    rc = L.vlr_bgzf_inflate(int(device), data, len(data), None, 0, C.byref(n))
    if n.value == 0:
        _check(rc)
        return b""
    out = np.empty(n.value, np.uint8)
    _check(L.vlr_bgzf_inflate(int(device), data, len(data), out.ctypes.data, n.value, C.byref(n)))
    return out.tobytes()


def device_timings(reset: bool = False) -> dict:
    """Stage times of the device reader summed since the last reset (vlr_ingest_device_timings)."""
    a = (C.c_double * 16)()
    L = _lib()
    L.vlr_ingest_device_timings.restype = None
    L.vlr_ingest_device_timings.argtypes = [C.POINTER(C.c_double), C.c_int]
    L.vlr_ingest_device_timings(a, int(reset))
    k = ["decode_offsets_up", "decode_launch", "feed_inflate", "split_scan", None, "decode", "copy_back", "host_table", "total", "inflated_bytes", "compressed_bytes", "records", "serial_walks", "inflate_kernel", "feed_call", "decode_wait"]
    return {n: a[i] for i, n in enumerate(k) if n}


def _gather_rows(mine: np.ndarray) -> np.ndarray:
    """The shard rows of all ranks in rank order: one all_gather of a few int64 per file over the default process group."""
    import torch
    import torch.distributed as tdist
    world = tdist.get_world_size()
    backend = tdist.get_backend()
    dev = "cuda:%d" % torch.cuda.current_device() if backend == "nccl" else "cpu"
    t = torch.from_numpy(np.ascontiguousarray(mine)).reshape(-1).to(dev)
    out = torch.empty(world * t.numel(), dtype=t.dtype, device=dev)
    tdist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().reshape((world,) + tuple(mine.shape))


def node_readers(node, paths: Sequence[str], omit_bias_mask: int = 0, threads: int = 0, chunk_records: int = 32768) -> List["ObsReader"]:
    """vlr_node_obs_readers_open: one sharded device reader per device of an engine.Node (one process, N devices); reader r delivers
    the records of shard r in file order, its tables hold the batch on node.devices[r]."""
    L = _lib()
    L.vlr_node_obs_readers_open.restype = C.c_int
    L.vlr_node_obs_readers_open.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    arr = (C.c_char_p * len(paths))(*[p_.encode() for p_ in paths])
    hs = (C.c_void_p * node.n_devices)()
    _check(L.vlr_node_obs_readers_open(node._h, len(paths), arr, int(omit_bias_mask), int(threads), hs))
    out = []
    for r in range(node.n_devices):
        rd = ObsReader.__new__(ObsReader)
        rd._h, rd.chunk_records, rd.device, rd.first_record, rd.n_records = C.c_void_p(hs[r]), int(chunk_records), node.devices[r], None, None
        L.vlr_obs_reader_next.restype = C.c_int
        L.vlr_obs_reader_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        L.vlr_obs_reader_close.restype = None
        L.vlr_obs_reader_close.argtypes = [C.c_void_p]
        out.append(rd)
    return out


class ObsReader:
    """vlr_obs_reader: the observation files a bounded number of records at a time.  Iterating yields (PileupBatch, Sites)."""

    def __init__(self, paths: Sequence[str], omit_bias_mask: int = 0, threads: int = 0, chunk_records: int = 250_000, device: Optional[int] = None,
                 host_columns: bool = True, async_columns: bool = False, shard: Optional[Tuple[int, int]] = None, gather=None):
        """device = None: the host reader.  device = k: vlr_obs_reader_open_device — BGZF inflate, record split and v15 decode as kernels on
        device k; the tables then also hold the batch in device memory (ObsTable.device_batch)."""
        L = _lib()
        L.vlr_obs_reader_open.restype = C.c_int
        L.vlr_obs_reader_open.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.vlr_obs_reader_open_device.restype = C.c_int
        L.vlr_obs_reader_open_device.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        self.device = device
        L.vlr_obs_reader_next.restype = C.c_int
        L.vlr_obs_reader_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        L.vlr_obs_reader_close.restype = None
        L.vlr_obs_reader_close.argtypes = [C.c_void_p]
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = C.c_void_p()
        self.first_record, self.n_records = 0, None
        if device is None:
            if shard is not None:
                raise ValueError("the sharded reader is the device reader (device=k)")
            _check(L.vlr_obs_reader_open(len(paths), arr, int(omit_bias_mask), int(threads), C.byref(h)))
        elif shard is not None:
            # shard = (k, N): this reader inflates and decodes about 1 / N of every file (vlr_obs_reader_open_device_shard); `gather` takes
            # this shard's int64 row array and returns the rows of all shards in shard order, shape (N, n_files, row) — by default one
            # torch.distributed all_gather over the default process group
            L.vlr_obs_reader_open_device_shard.restype = C.c_int
            L.vlr_obs_reader_open_device_shard.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
            L.vlr_obs_reader_shard_counts.restype = C.c_int
            L.vlr_obs_reader_shard_counts.argtypes = [C.c_void_p, C.c_void_p]
            L.vlr_obs_reader_shard_assign.restype = C.c_int
            L.vlr_obs_reader_shard_assign.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
            L.vlr_obs_reader_shard_row_size.restype = C.c_int
            k, n = int(shard[0]), int(shard[1])
            # A rank that cannot open (or count) its shard still takes part in the exchange, with a row of -1, and every rank raises:
            # leaving the collective to the others alone would hold them until the RCCL timeout.
            w = L.vlr_obs_reader_shard_row_size()
            mine = np.full((len(paths), w), -1, np.int64)
            own_error = None
            try:
                _check(L.vlr_obs_reader_open_device_shard(int(device), len(paths), arr, int(omit_bias_mask), int(threads), k, n, C.byref(h)))
                counted = np.zeros((len(paths), w), np.int64)
                _check(L.vlr_obs_reader_shard_counts(h, counted.ctypes.data))
                mine = counted
            except Exception as ex:  # noqa: BLE001 (raised below, after the exchange)
                own_error = ex
            try:
                self.shard_rows = mine
                rows = np.ascontiguousarray((gather or _gather_rows)(mine), np.int64)
                if own_error is not None:
                    raise own_error
                if rows.shape != (n, len(paths), w):
                    raise ValueError("gather returned rows of shape %r, expected %r" % (rows.shape, (n, len(paths), w)))
                if (rows < 0).any():
                    bad = sorted(set(int(r_) for r_ in np.nonzero((rows < 0).reshape(n, -1).any(axis=1))[0]))
                    raise engine.EngineError(abi.ERR_INVALID_ARGUMENT,
                                             "shard(s) %s of the sharded reader could not be opened" % ", ".join(str(b_) for b_ in bad))
                fr, nr = C.c_int64(0), C.c_int64(0)
                _check(L.vlr_obs_reader_shard_assign(h, rows.ctypes.data, C.byref(fr), C.byref(nr)))
                self.first_record, self.n_records = int(fr.value), int(nr.value)
                self.total_records = int(rows[:, 0, 0].sum())
            except Exception:
                if h:
                    L.vlr_obs_reader_close.restype = None
                    L.vlr_obs_reader_close.argtypes = [C.c_void_p]
                    L.vlr_obs_reader_close(h)
                raise
        else:
            _check(L.vlr_obs_reader_open_device(int(device), len(paths), arr, int(omit_bias_mask), int(threads), C.byref(h)))
        if device is not None:
            if async_columns:
                # next() returns while the columns are still on their way to the host (the device batch is complete): the calls writer and
                # ObsTable.fetch_columns() wait for them; do not read the numpy views of the columns before fetch_columns()
                L.vlr_obs_reader_set_async_columns.restype = C.c_int
                L.vlr_obs_reader_set_async_columns.argtypes = [C.c_void_p, C.c_int]
                _check(L.vlr_obs_reader_set_async_columns(h, 1))
            if not host_columns:
                # the observation columns stay on the device; the calls writer gets per-pileup summaries (vlr_obs_reader_set_host_columns).
                # ObsTable.fetch_columns() fills the numpy views of the columns on demand.
                L.vlr_obs_reader_set_host_columns.restype = C.c_int
                L.vlr_obs_reader_set_host_columns.argtypes = [C.c_void_p, C.c_int]
                _check(L.vlr_obs_reader_set_host_columns(h, 0))
        self._h, self.chunk_records = h, int(chunk_records)

    def next(self, max_records: Optional[int] = None):
        h = C.c_void_p()
        _check(_lib().vlr_obs_reader_next(self._h, int(max_records or self.chunk_records), C.byref(h)))
        if not h.value:
            return None
        item = _wrap_table(h)
        item[0].extra["native_table"].on_device = self.device is not None
        return item

    def __iter__(self):
        while True:
            item = self.next()
            if item is None:
                return
            yield item

    def close(self):
        if self._h:
            _lib().vlr_obs_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CallsWriter:
    """vlr_calls_writer: the calls file written one chunk of records at a time."""

    def __init__(self, path: str, header_text: str, part: Optional[Tuple[int, int]] = None):
        """part = (k, N): this writer writes part k of a file that N writers produce (concat_parts assembles it)."""
        L = _lib()
        L.vlr_calls_writer_open.restype = C.c_int
        L.vlr_calls_writer_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.vlr_calls_writer_append.restype = C.c_int
        L.vlr_calls_writer_append.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.Results), C.POINTER(C.c_char_p), C.c_int]
        L.vlr_calls_writer_close.restype = C.c_int
        L.vlr_calls_writer_close.argtypes = [C.c_void_p]
        h = C.c_void_p()
        _check(L.vlr_calls_writer_open(path.encode(), header_text.encode(), C.byref(h)))
        self._h = h
        if part is not None:
            L.vlr_calls_writer_set_part.restype = C.c_int
            L.vlr_calls_writer_set_part.argtypes = [C.c_void_p, C.c_int, C.c_int]
            _check(L.vlr_calls_writer_set_part(h, 1 if int(part[0]) == 0 else 0, 0))

    def append(self, table: ObsTable, results: CallResults, out_names: List[str], threads: int = 0):
        rs = results.as_struct()
        names = (C.c_char_p * len(out_names))(*[n.encode() for n in out_names])
        _check(_lib().vlr_calls_writer_append(self._h, table.handle, C.byref(rs), names, int(threads)))

    def close(self):
        if self._h:
            h, self._h = self._h, None
            _check(_lib().vlr_calls_writer_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def concat_parts(path: str, parts: Sequence[str]):
    """vlr_calls_concat_parts: the part files of a sharded run, in shard order, as one calls BCF (the parts are removed)."""
    L = _lib()
    L.vlr_calls_concat_parts.restype = C.c_int
    L.vlr_calls_concat_parts.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
    arr = (C.c_char_p * len(parts))(*[p.encode() for p in parts])
    _check(L.vlr_calls_concat_parts(path.encode(), arr, len(parts)))


def write_observations(path: str, batch: PileupBatch, sample: int, threads: int = 0, third_allele_evidence: Optional[np.ndarray] = None):
    """vlr_obs_write: the observation BCF of one sample of a host batch (synthetic sites)."""
    bs = batch.as_struct()
    third = None
    if third_allele_evidence is not None:
        third = np.ascontiguousarray(third_allele_evidence, np.int32)
    _check(_lib().vlr_obs_write(path.encode(), C.byref(bs), int(sample), None, third.ctypes.data if third is not None else None, int(threads)))


def write_calls(path: str, header_text: str, table: ObsTable, results: CallResults, out_names: List[str], threads: int = 0):
    """vlr_calls_write: calls BCF (path ends in .bcf) or text VCF for the loci of `table`."""
    rs = results.as_struct()
    names = (C.c_char_p * len(out_names))(*[n.encode() for n in out_names])
    _check(_lib().vlr_calls_write(path.encode(), header_text.encode(), table.handle, C.byref(rs), names, int(threads)))


def last_timings() -> dict:
    """Stage times of the last read_observations / write_calls (vlr_ingest_last_timings)."""
    a = (C.c_double * 16)()
    L = _lib()
    L.vlr_ingest_last_timings.restype = None
    L.vlr_ingest_last_timings.argtypes = [C.POINTER(C.c_double)]
    L.vlr_ingest_last_timings(a)
    k = ["file_read", "inflate", "parse_decode", "files_wall", "merge", "strings", "read_total", None, "encode", "deflate_write", "write_total"]
    return {n: a[i] for i, n in enumerate(k) if n}
