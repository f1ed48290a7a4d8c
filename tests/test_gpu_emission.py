"""Calls emission on the device (Call::write_final_record, calling/variants/mod.rs:447-559): the FORMAT/AFD text of
vlr_results.afd_text (afd_text_kernel) against printf's "%.3f=%.2f" on the same numbers, and calls files written from device text
against files written from the numbers."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from varlociraptor_amd import callsfmt, engine, ingest, synth
from varlociraptor_amd.batch import CallResults

pytestmark = pytest.mark.gpu


def _device_text(count, vaf, lnprob, text_capacity=None):
    L = engine.lib()
    n, cap = vaf.shape
    text_capacity = int(text_capacity if text_capacity is not None else max(64, n * cap * 24))
    text = np.zeros(text_capacity, np.uint8)
    span = np.zeros((n, 2), np.uint32)
    unf = C.c_uint32(0)
    L.vlr_selftest_afd_text.restype = C.c_int
    L.vlr_selftest_afd_text.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32)]
    count = np.ascontiguousarray(count, np.int32)
    vaf = np.ascontiguousarray(vaf, np.float64)
    lnprob = np.ascontiguousarray(lnprob, np.float64)
    rc = L.vlr_selftest_afd_text(0, n, cap, count.ctypes.data, vaf.ctypes.data, lnprob.ctypes.data, text.ctypes.data, text_capacity, span.ctypes.data, C.byref(unf))
    assert rc == 0, L.vlr_last_error().decode()
    out = []
    for i in range(n):
        off, ln = int(span[i, 0]), int(span[i, 1])
        out.append(None if ln == 0xffffffff else bytes(text[off:off + ln]).decode())
    return out, int(unf.value)


def _printf_text(count, vaf, lnprob):
    out = []
    ln10 = math.log(10.0)
    for i in range(len(count)):
        n = min(int(count[i]), vaf.shape[1])
        order = np.argsort(vaf[i, :n], kind="stable")
        out.append(",".join("%.3f=%.2f" % (vaf[i, j], -10.0 * lnprob[i, j] / ln10 + 0.0) for j in order))
    return out


def test_afd_text_is_printf_on_random_lists_ties_and_special_values():
    rng = np.random.default_rng(5)
    n, cap = 4000, 96
    count = rng.integers(0, cap + 20, n).astype(np.int32)   # (counts above the capacity are clamped like by the writer)
    count[:8] = [0, 1, 2, 63, 64, 65, cap, cap + 7]
    vaf = rng.random((n, cap))
    lnprob = -rng.exponential(20.0, (n, cap))
    # allele frequencies on a grid: equal values (stable order), exact ties of the third decimal (k + 1/2) / 1000 and neighbours
    g = rng.integers(0, 2001, (n // 2, cap)) / 2000.0
    vaf[: n // 2] = g
    vaf[100:200] = np.nextafter(vaf[100:200], 0.0)
    vaf[200:300] = np.nextafter(vaf[200:300], 2.0)
    vaf[300:310, :4] = [0.0, 1.0, 0.0005, 0.9995]
    # PHRED values on ties of the second decimal: p = -(k + 1/2) / 100 * ln10 / 10 up to rounding, zero, tiny, large, infinite
    k = rng.integers(0, 400000, (200, cap))
    lnprob[400:600] = -((k + 0.5) / 100.0) * math.log(10.0) / 10.0
    lnprob[600:610, :6] = [0.0, -0.0, -1e-300, -1e5, -np.inf, -2.5e10]
    got, unformatted = _device_text(count, vaf, lnprob)
    want = _printf_text(count, vaf, lnprob)
    assert unformatted == 0
    bad = [i for i in range(n) if got[i] != want[i]]
    assert not bad, (bad[:5], got[bad[0]], want[bad[0]])


def test_afd_text_leaves_what_it_does_not_format_to_the_numbers():
    rng = np.random.default_rng(6)
    cap = 1100
    count = np.array([5, 5, 5, 1100, 1024, 3], np.int32)
    vaf = rng.random((6, cap))
    lnprob = -rng.random((6, cap)) * 50
    lnprob[0, 2] = -1e13           # PHRED beyond 1e12: printf's exact expansion stays on the host
    vaf[1, 1] = np.nan             # no order
    got, unformatted = _device_text(count, vaf, lnprob)
    want = _printf_text(count, vaf, lnprob)
    assert got[0] is None and got[1] is None and got[3] is None   # (list 3 has more than 1 024 entries)
    assert unformatted == 3
    for i in (2, 4, 5):
        assert got[i] == want[i]
    # a text buffer that holds the first lists only: the rest is left to the numbers, nothing is cut
    got2, unformatted2 = _device_text(count[[2, 5, 4]], vaf[[2, 5, 4]], lnprob[[2, 5, 4]], text_capacity=200)
    assert got2[0] == want[2] and got2[1] == want[5] and got2[2] is None and unformatted2 == 1


@pytest.mark.parametrize("name", ["config3", "config5"])
def test_calls_written_from_device_text_equal_calls_written_from_numbers(name, tmp_path):
    """vlr_batch_run_device_in with vlr_results.afd_text: the lists stay on the device, the writer takes the text — the file must be
    the one written from the numbers, also with a text buffer that is too small for all lists (the numbers follow then)."""
    cfg = synth.CONFIGS[name]()
    b = synth.generate(cfg, 2500, seed=23)
    paths = []
    for s in range(b.n_samples):
        p = str(tmp_path / ("%s_%d.bcf" % (name, s)))
        ingest.write_observations(p, b, s)
        paths.append(p)
    rd = ingest.ObsReader(paths, chunk_records=1 << 20, device=0, host_columns=False)
    db, dsites = rd.next()
    table = db.extra["native_table"]
    plan = engine.Plan(cfg.scenario, device=0)
    names = cfg.scenario.out_names()
    header = callsfmt.header(names, cfg.scenario.sample_names, list(dsites.contig_names))
    L, S, cap = b.n_loci, b.n_samples, 48
    plain = plan.call_table_device(table, afd_capacity=cap)
    ref = str(tmp_path / "numbers.vcf")
    ingest.write_calls(ref, header, table, plain, names)
    want = open(ref).read()
    assert want.count("=") > 4 * L   # (there are lists)
    used = 0
    for tag in ("roomy", "tight"):
        tcap = L * S * cap * 16 if tag == "roomy" else used // 2
        res = CallResults(L, len(names), S, cap, afd_text_capacity=tcap)
        res.afd_vaf[:] = np.nan   # (numbers that are not copied stay NaN)
        plan.call_table_device(table, afd_capacity=cap, results=res)
        n_text = int((res.afd_text_span[:, :, 1] != 0xffffffff).sum())
        if tag == "roomy":
            assert n_text == L * S and np.isnan(res.afd_vaf).all()
            used = int(res.afd_text_span[:, :, 1].astype(np.int64).sum())
        else:
            assert 0 < n_text < L * S and not np.isnan(res.afd_vaf[:, :, 0]).all()
        out = str(tmp_path / (tag + ".vcf"))
        ingest.write_calls(out, header, table, res, names)
        assert open(out).read() == want
    rd.close()
    plan.close()


def test_device_text_through_the_chunked_host_path_and_the_node():
    """vlr_batch_run_host cuts the loci into staging chunks and vlr_node_batch_run_host into shards: every chunk / shard formats into
    its share of the caller's text buffer and the spans come back as offsets into the whole buffer."""
    cfg = synth.config2()
    b = synth.generate(cfg, 40000, seed=29)
    plan = engine.Plan(cfg.scenario, device=0)
    n_out, S, cap = len(cfg.scenario.out_names()), b.n_samples, 32
    plain = plan.call_host(b, afd_capacity=cap)
    want = _printf_text(plain.afd_count.reshape(-1), plain.afd_vaf.reshape(-1, cap), plain.afd_lnprob.reshape(-1, cap))
    assert sum(len(w) for w in want) > 40000 * 8

    def check(res):
        assert int((res.afd_text_span[:, :, 1] != 0xffffffff).sum()) == b.n_loci * S
        got = [res.afd_strings(l, s) for l in range(b.n_loci) for s in range(S)]
        bad = [i for i in range(len(want)) if got[i] != want[i]]
        assert not bad, (bad[:5], got[bad[0]], want[bad[0]])
    os.environ["VLR_HOST_CHUNK_MB"] = "4"
    try:
        res = CallResults(b.n_loci, n_out, S, cap, afd_text_capacity=b.n_loci * S * cap * 16)
        plan.call_host(b, afd_capacity=cap, results=res)
        check(res)
    finally:
        del os.environ["VLR_HOST_CHUNK_MB"]
    plan.close()
    node = engine.Node(cfg.scenario, devices=[0, 0, 0])
    res = CallResults(b.n_loci, n_out, S, cap, afd_text_capacity=b.n_loci * S * cap * 16)
    bs, rs = b.as_struct(), res.as_struct()
    engine._check(engine.lib().vlr_node_batch_run_host(node._h, C.byref(bs), C.byref(rs)))
    check(res)
    node.close()
