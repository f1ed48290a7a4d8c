"""ctypes binding of the engine's C ABI (varlociraptor_amd/libvlr.so, include/vlr.h).

Host-side mirror of the reference interface for the hot path: a `Caller` is configured with a
scenario (like CallerBuilder, src/calling/variants/calling.rs:52-81) and `call()` evaluates a batch
of records (Caller::call_record, calling.rs:720-842).  The library must be built (`build()`); there
is no Python/CPU fallback — importing works without a GPU, creating a plan does not.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from . import abi
from .batch import CallResults, PileupBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLR_LIB", os.path.join(_HERE, "libvlr.so"))  # VLR_LIB: tuning variants only
_LIB = None

EXPORTS = [
    "vlr_abi_version", "vlr_build_id", "vlr_last_error", "vlr_plan_create", "vlr_plan_destroy", "vlr_plan_n_out",
    "vlr_plan_n_samples", "vlr_plan_set_max_depth", "vlr_plan_set_max_obs", "vlr_plan_fit_max_obs", "vlr_plan_reserve", "vlr_batch_run", "vlr_batch_run_host", "vlr_batch_run_device_in",
    "vlr_plan_last_kernel_ms", "vlr_plan_work_counters", "vlr_host_alloc", "vlr_host_free",
    "vlr_node_create", "vlr_node_destroy", "vlr_node_n_devices", "vlr_node_device", "vlr_node_plan", "vlr_node_set_max_depth", "vlr_node_set_max_obs", "vlr_node_shard_range", "vlr_node_batch_run_host",
    "vlr_realign_batch", "vlr_realign_batch_host", "vlr_realign_fast_batch", "vlr_realign_fast_batch_host", "vlr_realign_homopolymer_batch", "vlr_realign_homopolymer_batch_host", "vlr_edit_distance_batch", "vlr_edit_distance_batch_host", "vlr_fdr_threshold", "vlr_selftest_math", "vlr_selftest_stream", "vlr_selftest_format_fixed", "vlr_selftest_afd_text",
    "vlr_obs_read", "vlr_obs_table_free", "vlr_obs_table_batch", "vlr_obs_table_sites", "vlr_obs_write", "vlr_calls_write", "vlr_ingest_last_timings", "vlr_ingest_total_timings",
    "vlr_obs_reader_open", "vlr_obs_reader_open_device", "vlr_obs_table_device_batch", "vlr_obs_reader_set_host_columns", "vlr_obs_reader_set_async_columns", "vlr_obs_table_fetch_columns", "vlr_obs_table_summaries", "vlr_bgzf_inflate", "vlr_ingest_device_timings", "vlr_ingest_device_trim", "vlr_obs_reader_open_device_shard", "vlr_obs_reader_shard_row_size", "vlr_obs_reader_shard_counts", "vlr_obs_reader_shard_assign", "vlr_node_obs_readers_open", "vlr_obs_reader_next", "vlr_obs_reader_close", "vlr_calls_writer_open", "vlr_calls_writer_append", "vlr_calls_writer_close", "vlr_calls_writer_set_part", "vlr_calls_concat_parts", "vlr_calls_filter_fdr",
]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vlr error %d: %s" % (code, msg))
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in ("vlr_kernels.hip", "vlr_kernels_deep.hip", "vlr_kernels_wide.hip", "vlr_kernels_widedeep.hip", "vlr_realign.hip", "vlr_fdr.hip", "vlr_inflate.hip", "vlr_decode.hip", "vlr_host.cpp", "vlr_ingest.cpp", "vlr_plan.h", "vlr_gpuio.h")] + [os.path.join(_HERE, "..", "include", "vlr.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", src_dir] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB_PATH


def build_id() -> str:
    """Source identity of the loaded library (vlr_build_id)."""
    return lib().vlr_build_id().decode()


def source_id() -> str:
    """The id a build of the current sources would carry (same recipe as csrc/Makefile)."""
    import hashlib
    h = hashlib.sha1()
    for f in ("csrc/vlr_kernels.hip", "csrc/vlr_kernels_deep.hip", "csrc/vlr_kernels_wide.hip", "csrc/vlr_kernels_widedeep.hip", "csrc/vlr_realign.hip", "csrc/vlr_fdr.hip", "csrc/vlr_inflate.hip", "csrc/vlr_decode.hip", "csrc/vlr_host.cpp", "csrc/vlr_ingest.cpp", "csrc/vlr_plan.h", "csrc/vlr_gpuio.h", "../include/vlr.h", "../include/vlr_detmath.h"):
        with open(os.path.join(_HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_all(force: bool = False) -> None:
    """The engine and its build-matrix variants in ONE parallel make (four hipcc jobs)."""
    src_dir = os.path.join(_HERE, "csrc")
    subprocess.check_call(["make", "-C", src_dir, "-j", "4", "all", "matrix"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)


MAX_OBS_LDS = 7680  # kept observations of one locus whose coefficient pairs fit the 120 kB LDS budget (vlr_plan_set_max_obs)

MATRIX_DIR = os.path.join(_HERE, "matrix")
MATRIX_LIBS = ("stress", "O1", "O2", "sync", "ilp")


def build_matrix(force: bool = False) -> str:
    """Compile the build-matrix variants of the engine (csrc/Makefile `matrix`; tests/test_gpu_build_matrix.py)."""
    src_dir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", src_dir, "-j", "3", "matrix"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return MATRIX_DIR


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(abi.ERR_NO_DEVICE, "libvlr.so is not built (run varlociraptor_amd.engine.build()); no fallback path exists")
        # One process, one HIP runtime: torch ships its own libamdhip64, libvlr.so is linked against /opt/rocm's.  If the
        # engine initialises HIP first, a later torch.cuda initialisation finds no devices; loading torch's runtime first
        # lets the dynamic loader resolve libvlr.so's dependency to the copy that is already mapped.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.vlr_abi_version.restype = C.c_int
        L.vlr_last_error.restype = C.c_char_p
        L.vlr_build_id.restype = C.c_char_p
        L.vlr_plan_create.restype = C.c_int
        L.vlr_plan_create.argtypes = [C.POINTER(abi.ScenarioDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.vlr_plan_destroy.restype = None
        L.vlr_plan_destroy.argtypes = [C.c_void_p]
        L.vlr_plan_n_out.restype = C.c_int
        L.vlr_plan_n_out.argtypes = [C.c_void_p]
        L.vlr_plan_n_samples.restype = C.c_int
        L.vlr_plan_n_samples.argtypes = [C.c_void_p]
        L.vlr_plan_set_max_depth.restype = C.c_int
        L.vlr_plan_set_max_depth.argtypes = [C.c_void_p, C.c_int]
        L.vlr_plan_set_max_obs.restype = C.c_int
        L.vlr_plan_set_max_obs.argtypes = [C.c_void_p, C.c_int]
        L.vlr_batch_run.restype = C.c_int
        L.vlr_batch_run.argtypes = [C.c_void_p, C.POINTER(abi.Batch), C.POINTER(abi.Results), C.c_void_p]
        L.vlr_batch_run_host.restype = C.c_int
        L.vlr_batch_run_host.argtypes = [C.c_void_p, C.POINTER(abi.Batch), C.POINTER(abi.Results)]
        L.vlr_plan_last_kernel_ms.restype = C.c_int
        L.vlr_plan_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.vlr_host_alloc.restype = C.c_void_p
        L.vlr_host_alloc.argtypes = [C.c_size_t]
        L.vlr_host_free.restype = None
        L.vlr_host_free.argtypes = [C.c_void_p]
        L.vlr_plan_work_counters.restype = C.c_int
        L.vlr_plan_work_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
        if L.vlr_abi_version() != abi.ABI_VERSION:
            raise EngineError(abi.ERR_INVALID_ARGUMENT, "ABI version mismatch")
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise EngineError(rc, lib().vlr_last_error().decode())


class _HostBlock:
    """Page-locked host memory from vlr_host_alloc, exposed to numpy through the array interface."""

    def __init__(self, nbytes: int):
        self.ptr = lib().vlr_host_alloc(max(1, int(nbytes)))
        if not self.ptr:
            raise EngineError(abi.ERR_OOM, lib().vlr_last_error().decode())
        self.__array_interface__ = {"shape": (max(1, int(nbytes)),), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        if getattr(self, "ptr", None) and _LIB is not None:
            _LIB.vlr_host_free(self.ptr)
            self.ptr = None


def host_array(shape, dtype) -> np.ndarray:
    """Uninitialised page-locked numpy array (include/vlr.h vlr_host_alloc); freed with its last view."""
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    flat = np.asarray(_HostBlock(n))[:n]
    return flat.view(dtype).reshape(shape)


def pin_batch(batch: PileupBatch) -> PileupBatch:
    """Copy of `batch` whose arrays live in page-locked memory, so that call_host stages them by direct DMA."""
    def pin(a):
        out = host_array(a.shape, a.dtype)
        out[...] = a
        return out
    return PileupBatch(batch.n_samples, pin(batch.obs_offset), {k: pin(v) for k, v in batch.columns.items()},
                       {k: pin(v) for k, v in batch.locus.items()})


class Plan:
    """vlr_plan: a scenario compiled for one contig on one device."""

    def __init__(self, scenario, device: int = 0, max_depth: Optional[int] = None):
        self.scenario = scenario
        self.device = device
        self._h = C.c_void_p()
        desc = scenario.desc()
        _check(lib().vlr_plan_create(C.byref(desc), device, C.byref(self._h)))
        self.n_out = lib().vlr_plan_n_out(self._h)
        self.n_samples = lib().vlr_plan_n_samples(self._h)
        if max_depth is not None:
            self.set_max_depth(max_depth)

    def set_max_depth(self, depth: int):
        _check(lib().vlr_plan_set_max_depth(self._h, int(depth)))

    def set_max_obs(self, n: int):
        _check(lib().vlr_plan_set_max_obs(self._h, int(n)))

    def fit_max_obs(self, obs_offset) -> int:
        """vlr_plan_fit_max_obs: LDS budget from the batch's pileup offsets (host array of n_loci * n_samples + 1 uint32) — the
        deepest locus, or the budget that lets sixteen workgroups share a CU when at most 0.5 % of the loci exceed it."""
        off = np.ascontiguousarray(obs_offset, np.uint32)
        L = lib()
        L.vlr_plan_fit_max_obs.restype = C.c_int
        L.vlr_plan_fit_max_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        rc = L.vlr_plan_fit_max_obs(self._h, off.ctypes.data, (len(off) - 1) // self.n_samples)
        if rc < 0:
            _check(rc)
        return int(rc)

    def close(self):
        if self._h:
            lib().vlr_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host buffers in, host buffers out (stages through the device; synchronous)
    def call_host(self, batch: PileupBatch, afd_capacity: int = 0, results: Optional[CallResults] = None) -> CallResults:
        """`results`: caller-owned buffers to fill (e.g. CallResults(..., alloc=engine.host_array): page-locked, reusable)."""
        res = results if results is not None else CallResults(batch.n_loci, self.n_out, self.n_samples, afd_capacity)
        bs, rs = batch.as_struct(), res.as_struct()
        _check(lib().vlr_batch_run_host(self._h, C.byref(bs), C.byref(rs)))
        return res

    def call_table_device(self, table, afd_capacity: int = 0, results: Optional[CallResults] = None) -> CallResults:
        """vlr_batch_run_device_in: evaluate the device-resident batch of a table read by a device reader (ingest.ObsReader(device=k));
        results in host buffers."""
        L = lib()
        L.vlr_batch_run_device_in.restype = C.c_int
        L.vlr_batch_run_device_in.argtypes = [C.c_void_p, C.POINTER(abi.Batch), C.c_void_p, C.POINTER(abi.Results)]
        db = table.device_batch()
        res = results if results is not None else CallResults(int(db.n_loci), self.n_out, self.n_samples, afd_capacity)
        rs = res.as_struct()
        _check(L.vlr_batch_run_device_in(self._h, C.byref(db), C.c_void_p(table._struct.obs_offset), C.byref(rs)))
        return res

    # ---- device pointers (torch tensors) in/out, stream-ordered
    def reserve(self, n_loci: int, with_afd=False):
        """Size the plan's device buffers once so that call_device never allocates (vlr_plan_reserve).  `with_afd`: False, or
        the AFD capacity (entries per locus and sample) the result buffers will have."""
        L = lib()
        L.vlr_plan_reserve.restype = C.c_int
        L.vlr_plan_reserve.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        _check(L.vlr_plan_reserve(self._h, int(n_loci), int(with_afd)))

    def call_device(self, dbatch: "DeviceBatch", out: "DeviceResults", stream=None):
        bs, rs = dbatch.as_struct(), out.as_struct()
        _check(lib().vlr_batch_run(self._h, C.byref(bs), C.byref(rs), C.c_void_p(stream) if stream else None))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(lib().vlr_plan_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def work_counters(self, reset: bool = False):
        out = (C.c_ulonglong * 2)()
        _check(lib().vlr_plan_work_counters(self._h, out, int(reset)))
        return int(out[0]), int(out[1])


def shard_range(n_loci: int, n_shards: int, shard: int):
    """vlr_node_shard_range: the contiguous block of loci shard `shard` of `n_shards` evaluates (no device needed)."""
    L = lib()
    L.vlr_node_shard_range.restype = C.c_int
    L.vlr_node_shard_range.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    l0, l1 = C.c_int64(), C.c_int64()
    _check(L.vlr_node_shard_range(int(n_loci), int(n_shards), int(shard), C.byref(l0), C.byref(l1)))
    return int(l0.value), int(l1.value)


class Node:
    """vlr_gpu_node: one scenario compiled for several devices of this node; call_host shards a host batch over them in one
    C-ABI call (vlr_node_batch_run_host) and returns the records in input order."""

    def __init__(self, scenario, devices=None, max_depth: Optional[int] = None):
        L = lib()
        L.vlr_node_create.restype = C.c_int
        L.vlr_node_create.argtypes = [C.POINTER(abi.ScenarioDesc), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.vlr_node_destroy.restype = None
        L.vlr_node_destroy.argtypes = [C.c_void_p]
        L.vlr_node_n_devices.restype = C.c_int
        L.vlr_node_n_devices.argtypes = [C.c_void_p]
        L.vlr_node_device.restype = C.c_int
        L.vlr_node_device.argtypes = [C.c_void_p, C.c_int]
        L.vlr_node_plan.restype = C.c_void_p
        L.vlr_node_plan.argtypes = [C.c_void_p, C.c_int]
        L.vlr_node_set_max_depth.restype = C.c_int
        L.vlr_node_set_max_depth.argtypes = [C.c_void_p, C.c_int]
        L.vlr_node_set_max_obs.restype = C.c_int
        L.vlr_node_set_max_obs.argtypes = [C.c_void_p, C.c_int]
        L.vlr_node_batch_run_host.restype = C.c_int
        L.vlr_node_batch_run_host.argtypes = [C.c_void_p, C.POINTER(abi.Batch), C.POINTER(abi.Results)]
        self.scenario = scenario
        self._h = C.c_void_p()
        desc = scenario.desc()
        if devices is None:
            _check(L.vlr_node_create(C.byref(desc), 0, None, C.byref(self._h)))
        else:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            _check(L.vlr_node_create(C.byref(desc), len(devices), arr, C.byref(self._h)))
        self.n_devices = L.vlr_node_n_devices(self._h)
        self.devices = [L.vlr_node_device(self._h, r) for r in range(self.n_devices)]
        p0 = L.vlr_node_plan(self._h, 0)
        self.n_out = L.vlr_plan_n_out(p0)
        self.n_samples = L.vlr_plan_n_samples(p0)
        if max_depth is not None:
            _check(L.vlr_node_set_max_depth(self._h, int(max_depth)))

    def set_max_obs(self, n: int):
        _check(lib().vlr_node_set_max_obs(self._h, int(n)))

    def call_host(self, batch: PileupBatch, afd_capacity: int = 0) -> CallResults:
        res = CallResults(batch.n_loci, self.n_out, self.n_samples, afd_capacity)
        bs, rs = batch.as_struct(), res.as_struct()
        _check(lib().vlr_node_batch_run_host(self._h, C.byref(bs), C.byref(rs)))
        return res

    def close(self):
        if self._h:
            lib().vlr_node_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBatch:
    """vlr_batch whose columns are torch tensors resident in HBM (plumbing only)."""

    def __init__(self, batch: PileupBatch, device="cuda:0"):
        import torch
        self.n_loci, self.n_samples, self.n_obs = batch.n_loci, batch.n_samples, batch.n_obs
        self.t = {}
        self.t["obs_offset"] = torch.from_numpy(batch.obs_offset.view(np.int32)).to(device)
        for name, _ in abi.OBS_COLUMNS:
            a = batch.columns[name]
            self.t[name] = torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(device)
        for name, _ in abi.LOCUS_COLUMNS:
            self.t[name] = torch.from_numpy(batch.locus[name]).to(device)
        self._algorithmic_bytes = batch.algorithmic_bytes()

    def algorithmic_bytes(self):
        return self._algorithmic_bytes

    def as_struct(self) -> abi.Batch:
        b = abi.Batch()
        b.n_loci, b.n_samples, b.n_obs = self.n_loci, self.n_samples, self.n_obs
        b.obs_offset = self.t["obs_offset"].data_ptr()
        for name, _ in abi.OBS_COLUMNS:
            setattr(b, name, self.t[name].data_ptr())
        for name, _ in abi.LOCUS_COLUMNS:
            setattr(b, name, self.t[name].data_ptr())
        return b


class DeviceResults:
    def __init__(self, n_loci, n_out, n_samples, device="cuda:0", afd_capacity=0):
        import torch
        self.n_loci, self.n_out, self.n_samples = n_loci, n_out, n_samples
        self.afd_capacity = afd_capacity
        if afd_capacity:
            self.afd_count = torch.zeros((n_loci, n_samples), dtype=torch.int32, device=device)
            self.afd_vaf = torch.empty((n_loci, n_samples, afd_capacity), dtype=torch.float64, device=device)
            self.afd_lnprob = torch.empty((n_loci, n_samples, afd_capacity), dtype=torch.float64, device=device)
        self.ln_posterior = torch.empty((n_loci, n_out), dtype=torch.float64, device=device)
        self.ln_marginal = torch.empty(n_loci, dtype=torch.float64, device=device)
        self.map_vaf = torch.empty((n_loci, n_samples), dtype=torch.float64, device=device)
        self.map_bias = torch.empty((n_loci, abi.N_BIAS), dtype=torch.uint8, device=device)
        self.best_event = torch.empty(n_loci, dtype=torch.int32, device=device)
        self.status = torch.empty(n_loci, dtype=torch.int32, device=device)

    def as_struct(self) -> abi.Results:
        r = abi.Results()
        r.n_loci, r.n_out, r.n_samples = self.n_loci, self.n_out, self.n_samples
        r.ln_posterior = self.ln_posterior.data_ptr()
        r.ln_marginal = self.ln_marginal.data_ptr()
        r.map_vaf = self.map_vaf.data_ptr()
        r.map_bias = self.map_bias.data_ptr()
        r.best_event = self.best_event.data_ptr()
        r.status = self.status.data_ptr()
        if self.afd_capacity:
            r.afd_capacity = self.afd_capacity
            r.afd_count = self.afd_count.data_ptr()
            r.afd_vaf = self.afd_vaf.data_ptr()
            r.afd_lnprob = self.afd_lnprob.data_ptr()
        return r

    def to_host(self) -> CallResults:
        res = CallResults(self.n_loci, self.n_out, self.n_samples)
        res.ln_posterior[:] = self.ln_posterior.cpu().numpy()
        res.ln_marginal[:] = self.ln_marginal.cpu().numpy()
        res.map_vaf[:] = self.map_vaf.cpu().numpy()
        res.map_bias[:] = self.map_bias.cpu().numpy()
        res.best_event[:] = self.best_event.cpu().numpy()
        res.status[:] = self.status.cpu().numpy().view(np.uint32)
        return res
