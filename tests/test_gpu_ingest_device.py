"""Device front door (csrc/vlr_inflate.hip, csrc/vlr_decode.hip behind vlr_obs_reader_open_device / vlr_bgzf_inflate, include/vlr.h):
the BGZF inflate kernel against zlib on every DEFLATE block type, and the device reader against the host reader (csrc/vlr_ingest.cpp,
itself held against the Python restatement and the reference's files in tests/test_ingest.py) — column by column, flag by flag,
string by string; the device-resident batch evaluated by vlr_batch_run against the host batch through vlr_batch_run_host."""
import os
import struct
import zlib

import numpy as np
import pytest

from varlociraptor_amd import engine, ingest, synth

pytestmark = pytest.mark.gpu


def _member(payload: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_at=()) -> bytes:
    """One BGZF member (SAM spec 4.1) around a raw DEFLATE stream of `payload`."""
    assert len(payload) <= 65536
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    out, last = b"", 0
    for cut in list(flush_at) + [len(payload)]:
        out += co.compress(payload[last:cut])
        if cut != len(payload):
            out += co.flush(zlib.Z_FULL_FLUSH)   # ends the block and emits an empty stored block
        last = cut
    out += co.flush()
    bsize = 18 + len(out) + 8
    assert bsize <= 65536, bsize
    head = b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return head + out + struct.pack("<II", zlib.crc32(payload), len(payload))


def _payloads(rng):
    ints = (rng.integers(0, 70000, 12000).astype(np.int32)).tobytes()            # the typed INFO vectors of a record: X Y 0 0 patterns
    text = (b"the quick brown fox jumps over the lazy dog " * 1400)[:60000]
    return {
        "empty": b"",
        "one byte": b"A",
        "zeros (distance 1, length 258)": bytes(65536),
        "random (literals, long codes)": rng.integers(0, 256, 40000, dtype=np.uint8).tobytes(),
        "int32 words": ints[:48000],
        "text": text,
        "period 3": (b"abc" * 22000)[:65000],
        "period 200": (bytes(range(200)) * 400)[:64000],
        "skewed alphabet": rng.choice(np.arange(256, dtype=np.uint8), 50000, p=np.r_[[0.7], np.full(255, 0.3 / 255)]).tobytes(),
    }


@pytest.fixture(params=["0", "1"], ids=["symbol loop", "speculative batches"])
def inflate_mode(request):
    """Both symbol loops of vlr_inflate_kernel (VLR_INFLATE_BATCH is read at every launch)."""
    old = os.environ.get("VLR_INFLATE_BATCH")
    os.environ["VLR_INFLATE_BATCH"] = request.param
    yield request.param
    if old is None:
        del os.environ["VLR_INFLATE_BATCH"]
    else:
        os.environ["VLR_INFLATE_BATCH"] = old


@pytest.fixture(params=["mapped", "descriptor", "descriptor, ring of 1.25 MB"])
def reader_mode(request):
    """The two ways the device reader takes a file: mapped (small files, sharded readers) and descriptor mode (pread into the page-locked
    ring, member chain indexed there by the stager thread; files above 32 MB by default — VLR_INGEST_STAGE_MIN_MB=0 sends the tests'
    small files through it)."""
    keys = ("VLR_INGEST_STAGE_MIN_MB", "VLR_INGEST_STAGE_SEG_KB")
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ.pop(k, None)
    if request.param != "mapped":
        os.environ["VLR_INGEST_STAGE_MIN_MB"] = "0"
    if "ring" in request.param:
        # twenty segments of 64 KB: the files wrap around the ring many times, a request takes several feeds, the reader has to give
        # ranges back before the stager can go on (a request of 131 072 records against the default ring hung before this was handled)
        os.environ["VLR_INGEST_STAGE_SEG_KB"] = "64"
    yield request.param
    for k in keys:
        os.environ.pop(k, None)
        if old[k] is not None:
            os.environ[k] = old[k]


def test_inflate_kernel_equals_zlib_on_every_block_type(inflate_mode):
    rng = np.random.default_rng(5)
    members, plain = [], []
    for name, pl in _payloads(rng).items():
        for level, strategy in [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)]:
            body = pl if level or len(pl) <= 60000 else pl[:60000]   # (stored blocks do not shrink: keep the member under 64 KiB)
            if level and name.startswith("random"):
                body = pl[:30000]
            try:
                m = _member(body, level, strategy)
            except AssertionError:
                continue
            members.append(m); plain.append(body)
    # several DEFLATE blocks inside one member, with the empty stored blocks Z_FULL_FLUSH leaves between them
    body = _payloads(rng)["text"][:40000] + bytes(3000) + rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    members.append(_member(body, 6, flush_at=(100, 101, 20000, 43000))); plain.append(body)
    # each member alone (a failure names the case), then all of them as one stream (output offsets that are not multiples of 16)
    for m, pl in zip(members, plain):
        assert ingest.bgzf_inflate(m) == pl
    assert ingest.bgzf_inflate(b"".join(members)) == b"".join(plain)
    # the EOF member htslib appends
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    assert ingest.bgzf_inflate(members[3] + eof) == plain[3]


def test_inflate_kernel_refuses_damaged_members(inflate_mode):
    rng = np.random.default_rng(6)
    pl = _payloads(rng)["text"]
    m = bytearray(_member(pl, 6))
    ok = bytes(m)
    assert ingest.bgzf_inflate(ok) == pl
    bad_isize = bytearray(ok); bad_isize[-4:] = struct.pack("<I", len(pl) - 1)
    with pytest.raises(engine.EngineError):
        ingest.bgzf_inflate(bytes(bad_isize))
    # a flipped byte inside the DEFLATE stream is an error — a broken code, or bytes whose CRC32 is not the trailer's (vlr_crc_kernel;
    # htslib refuses such members, bgzf.c inflate_block) — never a crash; bytes that differ from the original never come back
    for at in range(30, len(ok) - 12, 97):
        x = bytearray(ok); x[at] ^= 0x5a
        try:
            got = ingest.bgzf_inflate(bytes(x))
            assert got == pl, "a damaged member was accepted (byte %d)" % at   # (the flip did not change what the stream decodes to)
        except engine.EngineError:
            pass
    with pytest.raises(engine.EngineError):
        ingest.bgzf_inflate(b"not a gzip member at all, not even close....")


def test_member_crc32_is_checked_on_the_device(inflate_mode):
    """a flipped bit in a literal or a stored byte decodes to another VALID stream of the same length; only the member's
    CRC32 (RFC 1952 trailer) catches it.  Stored blocks (level 0) make every payload byte such a case; the trailer itself too."""
    rng = np.random.default_rng(8)
    for name, pl in _payloads(rng).items():
        if not pl:
            continue
        body = pl[:60000]
        for level in (0, 1, 6):
            ok = _member(body if level == 0 else pl[:30000] if name.startswith("random") else pl, level)
            want = ingest.bgzf_inflate(ok)
            bad = bytearray(ok); bad[-8] ^= 0x01                       # the CRC32 field itself
            with pytest.raises(engine.EngineError, match="CRC32|corrupt"):
                ingest.bgzf_inflate(bytes(bad))
            if level == 0:                                             # stored: header 5 bytes, then the payload as it is
                for at in (18 + 5, 18 + 5 + len(want) // 2, 18 + 5 + len(want) - 1):
                    bad = bytearray(ok); bad[at] ^= 0x40
                    with pytest.raises(engine.EngineError, match="CRC32|corrupt"):
                        ingest.bgzf_inflate(bytes(bad))
    # lengths around the lane split of the kernel (lane 0 takes n - 63 C bytes, C a multiple of four): all checked against zlib's CRC
    for n in (1, 2, 3, 4, 63, 64, 65, 255, 256, 257, 259, 1000, 4095, 4096, 4099, 65535, 65536):
        pl = rng.integers(0, 256, n, dtype=np.uint8).tobytes() if n <= 60000 else (rng.integers(0, 256, 977, dtype=np.uint8).tobytes() * 68)[:n]
        for level in (0, 6):
            if level == 0 and n > 60000:
                continue   # (a stored member holds at most 65 536 - 31 bytes)
            assert ingest.bgzf_inflate(_member(pl, level)) == pl


def test_readers_refuse_a_member_with_a_wrong_crc(tmp_path, reader_mode):
    """Both readers (device: vlr_crc_kernel; host: libdeflate / zlib CRC behind inflate_raw) fail on an observation file with one
    flipped payload bit that still inflates."""
    cfg = synth.config2()
    b = synth.generate(cfg, 40, seed=3)
    path = str(tmp_path / "obs.bcf")
    ingest.write_observations(path, b, 0)
    raw = bytearray(open(path, "rb").read())
    raw_ok = bytes(raw)
    # the trailer of the first member that holds record data: CRC32 sits 8 bytes before the member's end
    off, k = 0, 0
    while off < len(raw_ok):
        bsize = struct.unpack_from("<H", raw_ok, off + 16)[0] + 1
        if k == 1 or off + bsize >= len(raw_ok) - 28:
            raw[off + bsize - 8] ^= 0x10
            break
        off += bsize; k += 1
    bad_path = str(tmp_path / "bad.bcf")
    open(bad_path, "wb").write(bytes(raw))
    for device in (True, False):
        with pytest.raises(Exception, match="CRC32|corrupt"):
            r = ingest.ObsReader([bad_path], device=0 if device else None)
            while r.next(1000) is not None:
                pass


def _tables_equal(a, sa, b, sb):
    assert a.n_samples == b.n_samples and a.n_loci == b.n_loci
    assert np.array_equal(a.obs_offset, b.obs_offset)
    for k in a.columns:
        x, y = a.columns[k], b.columns[k]
        assert x.dtype == y.dtype and np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y), k
    for k in a.locus:
        assert np.array_equal(a.locus[k], b.locus[k]), k
    for k in ("third_allele_evidence", "group_representative", "group_key"):
        assert np.array_equal(a.extra[k], b.extra[k]), k
    for k in ("prior_het_ln", "prior_som_ln"):
        assert np.array_equal(a.extra[k], b.extra[k], equal_nan=True), k
    assert len(sa) == len(sb)
    for i in range(len(sa)):
        assert sa[i] == sb[i], i


def _read_all(paths, device, chunk):
    rd = ingest.ObsReader(paths, chunk_records=chunk, device=device)
    out = list(rd)
    rd.close()
    return out


def _rows(chunks):
    rows = []
    for b, sites in chunks:
        S = b.n_samples
        for l in range(b.n_loci):
            lo, hi = int(b.obs_offset[l * S]), int(b.obs_offset[(l + 1) * S])
            rows.append((tuple(int(x) for x in b.obs_offset[l * S:(l + 1) * S + 1] - lo),
                         {k: v[lo:hi].tobytes() for k, v in b.columns.items()}, {k: v[l].tobytes() for k, v in b.locus.items()},
                         b.extra["third_allele_evidence"][lo:hi].tobytes(), int(b.extra["group_key"][l]), sites[l]))
    return rows


def _concat_check(host_chunks, dev_chunks, first=None):
    """Chunk boundaries differ between the readers (the device reader delivers what its buffers hold): compare record by record."""
    h, d = _rows(host_chunks), _rows(dev_chunks)
    if first is not None:
        h = h[:first]
    assert len(h) == len(d)
    for i, (x, y) in enumerate(zip(h, d)):
        assert x == y, "record %d differs" % i


@pytest.mark.parametrize("name", ["config3", "config4", "config5"])
def test_device_reader_equals_host_reader(name, tmp_path, inflate_mode, reader_mode):
    cfg = synth.CONFIGS[name]()
    b = synth.generate(cfg, 2500, seed=31)
    third = np.where(np.arange(b.n_obs) % 5 == 0, np.arange(b.n_obs) % 4, -1).astype(np.int32)
    paths = []
    for s in range(b.n_samples):
        p = str(tmp_path / ("%s_%d.bcf" % (name, s)))
        ingest.write_observations(p, b, s, third_allele_evidence=third)
        paths.append(p)
    host = _read_all(paths, None, 1 << 20)
    dev = _read_all(paths, 0, 1 << 20)
    assert len(host) == 1 and len(dev) == 1
    _tables_equal(host[0][0], host[0][1], dev[0][0], dev[0][1])
    # small requests: many chunks, records that straddle the buffered bytes, the carry between calls
    for chunk in (1, 7, 333):
        if chunk == 1:
            d1 = ingest.ObsReader(paths, chunk_records=1, device=0)
            first = [d1.next() for _ in range(40)]
            d1.close()
            assert all(x is not None and x[0].n_loci == 1 for x in first)
            _concat_check(host, first, first=40)
            continue
        _concat_check(host, _read_all(paths, 0, chunk))
    # the verified walk and the serial walk agree
    os.environ["VLR_INGEST_SERIAL_WALK"] = "1"
    try:
        _concat_check(host, _read_all(paths, 0, 900))
    finally:
        del os.environ["VLR_INGEST_SERIAL_WALK"]
    t = ingest.device_timings()
    assert t["records"] > 0


def test_device_reader_on_files_of_the_reference(golden_dir, tmp_path, inflate_mode, reader_mode):
    """normal.bcf of the reference's flamegraph_profiling fixture was written by varlociraptor preprocess through htslib (other member
    sizes, int8 / int16 typed vectors, its own header); the fourteen format-v15 testcases are re-encoded as BCF by the Python writer."""
    import glob
    from varlociraptor_amd import bcfio
    files = [os.path.join(golden_dir, "flamegraph_profiling", "normal.bcf")]
    for i, v in enumerate(sorted(glob.glob(os.path.join(golden_dir, "testcases", "*", "observations.vcf")))):
        out = str(tmp_path / ("t%d.bcf" % i))
        if bcfio.vcf_to_bcf(v, out):
            files.append(out)
    for f in files:
        h = _read_all([f], None, 1 << 20)
        d = _read_all([f], 0, 1 << 20)
        _concat_check(h, d)


def test_device_batch_evaluates_like_the_host_batch(tmp_path):
    import ctypes as C
    from varlociraptor_amd import abi
    cfg = synth.config3()
    b = synth.generate(cfg, 3000, seed=33)
    paths = []
    for s in range(b.n_samples):
        p = str(tmp_path / ("s%d.bcf" % s))
        ingest.write_observations(p, b, s)
        paths.append(p)
    (hb, _), = _read_all(paths, None, 1 << 20)
    (db, _), = _read_all(paths, 0, 1 << 20)
    plan = engine.Plan(cfg.scenario, device=0)
    ref = plan.call_host(hb, afd_capacity=32)
    got = plan.call_table_device(db.extra["native_table"], afd_capacity=32)
    for k in ("ln_posterior", "map_vaf", "status", "best_event"):
        x, y = getattr(ref, k), getattr(got, k)
        assert np.array_equal(x, y, equal_nan=True), k
    assert np.array_equal(ref.afd_count, got.afd_count)
    valid = np.arange(32)[None, None, :] < ref.afd_count[:, :, None]   # (entries beyond the count are not written)
    assert np.array_equal(ref.afd_vaf[valid], got.afd_vaf[valid]) and np.array_equal(ref.afd_lnprob[valid], got.afd_lnprob[valid])
    plan.close()


def test_device_reader_refuses_what_it_does_not_read(tmp_path, golden_dir, reader_mode):
    with pytest.raises(engine.EngineError) as e:
        ingest.ObsReader([os.path.join(golden_dir, "flamegraph_profiling", "normal.vcf")], device=0)
    assert e.value.code == -2   # VLR_ERR_UNSUPPORTED: the host reader takes text VCF / plain gzip
    cfg = synth.config3()
    b = synth.generate(cfg, 300, seed=3)
    p = str(tmp_path / "a.bcf")
    ingest.write_observations(p, b, 0)
    raw = open(p, "rb").read()
    cut = str(tmp_path / "cut.bcf")
    # whole members only, the last ones missing: the stream ends inside a record
    off, offs = 0, []
    while off < len(raw):
        offs.append(off)
        off += struct.unpack_from("<H", raw, off + 16)[0] + 1
    open(cut, "wb").write(raw[:offs[len(offs) // 2]])
    with pytest.raises(engine.EngineError, match="truncated"):
        _read_all([cut], 0, 1 << 20)


@pytest.mark.parametrize("name", ["config3", "config5"])
def test_writer_from_device_summaries_equals_writer_from_columns(name, tmp_path):
    """host_columns=False: the observation columns stay in device memory, obs_text_kernel (one wave per pileup) writes the OBS text and
    counts what the calls writer formats (SAOBS / SROBS / DP) and the writer works from those summaries — also for the synthetic
    pileups, which have almost one distinct observation key per observation.  The file must be the one written from the columns —
    text VCF, so every character is compared — and fetch_columns must deliver the host reader's columns."""
    from varlociraptor_amd import callsfmt
    cfg = synth.CONFIGS[name]()
    b = synth.generate(cfg, 3000, seed=37)
    third = np.where(np.arange(b.n_obs) % 3 == 0, np.arange(b.n_obs) % 6, -1).astype(np.int32)
    paths = []
    for s in range(b.n_samples):
        p = str(tmp_path / ("%s_%d.bcf" % (name, s)))
        ingest.write_observations(p, b, s, third_allele_evidence=third)
        paths.append(p)
    (hb, hsites), = _read_all(paths, None, 1 << 20)
    rd = ingest.ObsReader(paths, chunk_records=1 << 20, device=0, host_columns=False)
    db, dsites = rd.next()
    table = db.extra["native_table"]
    plan = engine.Plan(cfg.scenario, device=0)
    res = plan.call_table_device(table, afd_capacity=24)
    names = cfg.scenario.out_names()
    header = callsfmt.header(names, cfg.scenario.sample_names, list(dsites.contig_names))
    a, c = str(tmp_path / "from_summaries.vcf"), str(tmp_path / "from_columns.vcf")
    assert table.summaries() == (True, 0)
    ingest.write_calls(a, header, table, res, names)
    assert table.summaries() == (True, 0)   # (the writer did not have to fetch the columns)
    table.fetch_columns()
    assert table.summaries() == (False, 0)
    ingest.write_calls(c, header, table, res, names)
    ta, tc = open(a).read(), open(c).read()
    assert ta == tc
    assert ta.count("\n") > b.n_loci
    _tables_equal(hb, hsites, db, dsites)
    # and the host reader's table gives the same file
    ref = plan.call_host(hb, afd_capacity=24)
    h = str(tmp_path / "host.vcf")
    ingest.write_calls(h, header, hb.extra["native_table"], ref, names)
    assert open(h).read() == ta
    rd.close()
    plan.close()


def test_device_reader_records_larger_than_a_segment_and_tiny_files(tmp_path, reader_mode):
    """Records of 0.2 to 2 MB (pileups of thousands of observations) span many 64 KiB segments of the inflated stream: segments
    without any record start, anchors found far behind the boundary; plus the degenerate files (no record, one record)."""
    base = synth.config3()
    deep = synth.config3()
    deep.depth, deep.max_depth = 6000.0, 20000
    b = synth.generate(deep, 9, seed=5)
    assert b.n_obs > 50000
    for name, batch in (("deep", b), ("one", synth.generate(base, 1, seed=6)), ("none", synth.generate(base, 4, seed=7).select(np.arange(0)))):
        paths = []
        for s in range(batch.n_samples):
            p = str(tmp_path / ("%s_%d.bcf" % (name, s)))
            ingest.write_observations(p, batch, s)
            paths.append(p)
        host = _read_all(paths, None, 1 << 20)
        for chunk in (1 << 20, 2):
            dev = _read_all(paths, 0, chunk)
            _concat_check(host, dev)
    # mixed: deep records between ordinary ones, several requests
    mix = synth.generate(base, 400, seed=8)
    paths = []
    for s in range(mix.n_samples):
        p = str(tmp_path / ("mix_%d.bcf" % s))
        ingest.write_observations(p, mix, s)
        paths.append(p)
    _concat_check(_read_all(paths, None, 1 << 20), _read_all(paths, 0, 37))


def test_device_reader_on_damaged_records(tmp_path, reader_mode):
    """Bytes flipped INSIDE the inflated records (length words, typed descriptors, vector payloads), re-packed as valid BGZF: the device
    reader must behave like the host reader — the same table or an error, never a crash, a hang or a silently different table."""
    import gzip
    from varlociraptor_amd import bcfio
    cfg = synth.config3()
    b = synth.generate(cfg, 40, seed=11)
    p = str(tmp_path / "a.bcf")
    ingest.write_observations(p, b, 0)
    raw = bytearray(gzip.open(p, "rb").read())
    l_text = struct.unpack_from("<I", raw, 5)[0]
    first = 9 + l_text
    rng = np.random.default_rng(12)
    n_err = n_same = 0
    for trial in range(60):
        x = bytearray(raw)
        # a handful of positions: the start of a record (length words / fixed fields) or anywhere behind the header
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(first, len(x))) if trial % 3 else first + int(rng.integers(0, 64))
            x[at] ^= int(rng.integers(1, 256))
        q = str(tmp_path / ("d%d.bcf" % trial))
        with open(q, "wb") as fh:
            for o in range(0, len(x), 0xff00):
                fh.write(bcfio._bgzf_block(bytes(x[o:o + 0xff00])))
            fh.write(bcfio._BGZF_EOF)
        try:
            h = _read_all([q], None, 1 << 20)
        except engine.EngineError:
            h = None
        try:
            d = _read_all([q], 0, 1 << 20)
        except engine.EngineError:
            d = None
        if h is None or d is None:
            assert h is None and d is None, "trial %d: one reader refused the file, the other did not" % trial
            n_err += 1
        else:
            _concat_check(h, d)
            n_same += 1
    assert n_err > 0 and n_same > 0


def test_async_columns_are_there_when_they_are_read(tmp_path):
    """async_columns=True: next() returns while the column copy to the host is in flight; fetch_columns() and the calls writer wait."""
    from varlociraptor_amd import callsfmt
    cfg = synth.config3()
    b = synth.generate(cfg, 6000, seed=41)
    paths = []
    for s in range(b.n_samples):
        p = str(tmp_path / ("s%d.bcf" % s))
        ingest.write_observations(p, b, s)
        paths.append(p)
    host = _read_all(paths, None, 1 << 20)
    rd = ingest.ObsReader(paths, chunk_records=2000, device=0, async_columns=True)
    chunks, tables = [], []
    plan = engine.Plan(cfg.scenario, device=0)
    names = cfg.scenario.out_names()
    out = str(tmp_path / "calls.vcf")
    w = None
    for db, sites in rd:
        t = db.extra["native_table"]
        res = plan.call_table_device(t, afd_capacity=8)
        if w is None:
            w = ingest.CallsWriter(out, callsfmt.header(names, cfg.scenario.sample_names, list(sites.contig_names)))
        w.append(t, res, names)          # (waits for the columns of this table)
        t.fetch_columns()
        chunks.append((db, sites))
    w.close()
    _concat_check(host, chunks)
    ref = plan.call_host(host[0][0], afd_capacity=8)
    h = str(tmp_path / "host.vcf")
    ingest.write_calls(h, callsfmt.header(names, cfg.scenario.sample_names, list(host[0][1].contig_names)), host[0][0].extra["native_table"], ref, names)
    assert open(h).read() == open(out).read()
    rd.close(); plan.close()


def test_device_reader_refuses_sample_files_that_do_not_match(tmp_path, reader_mode):
    """calling.rs:369-390: the sample files must hold the same records in the same order — one file shorter, or a different site at
    some record, is an error of the device reader as it is of the host reader."""
    cfg = synth.config3()
    b = synth.generate(cfg, 900, seed=17)
    full = []
    for s in range(b.n_samples):
        p = str(tmp_path / ("s%d.bcf" % s))
        ingest.write_observations(p, b, s)
        full.append(p)
    short = str(tmp_path / "short.bcf")
    ingest.write_observations(short, b.select(np.arange(700)), 1)
    other = str(tmp_path / "other.bcf")
    ingest.write_observations(other, synth.generate(cfg, 900, seed=18), 1)
    for device in (None, 0):
        with pytest.raises(engine.EngineError, match="inconsistent observations"):
            _read_all([full[0], short], device, 256)
        with pytest.raises(engine.EngineError, match="inconsistent observations"):
            _read_all([full[0], other], device, 256)


def _drain(reader):
    parts = []
    for batch, sites in reader:
        parts.append((batch, sites))
    return parts


def test_sharded_reader_partitions_the_file_and_inflates_its_share_only(tmp_path):
    """VERDICT r04 missing #2: N readers (ranks / devices) each inflate and decode about 1 / N of every file
    (vlr_obs_reader_open_device_shard: byte shares of the members, a guessed first record start confirmed by the neighbour's landing,
    record numbers from an exchange of counts).  Three shards in one process (threads stand in for ranks, a barrier for the
    all-gather): their tables, in shard order, are the unsharded reader's records column by column, and together they inflate
    little more than the file once — not three times."""
    import threading
    cfg = synth.config3()
    cfg.depth = 30.0
    b = synth.generate(cfg, 6000, seed=9)
    paths = []
    for s in range(2):
        p = str(tmp_path / ("s%d.bcf" % s))
        ingest.write_observations(p, b, s)
        paths.append(p)
    ingest.device_timings(reset=True)
    whole = _drain(ingest.ObsReader(paths, device=0, chunk_records=1500))
    t_whole = ingest.device_timings(reset=True)
    N = 3
    rows, out, errs = [None] * N, [None] * N, []
    bar = threading.Barrier(N)

    def run(k):
        def gather(mine):
            rows[k] = np.array(mine)
            bar.wait(timeout=120)
            return np.stack(rows)
        try:
            r = ingest.ObsReader(paths, device=0, chunk_records=700, shard=(k, N), gather=gather)
            out[k] = (r.first_record, r.n_records, _drain(r))
        except Exception as ex:   # noqa: BLE001
            errs.append(ex)
            try:
                bar.abort()
            except Exception:
                pass
    th = [threading.Thread(target=run, args=(k,)) for k in range(N)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    t_shards = ingest.device_timings(reset=True)
    # the ranges partition the records in shard order
    at = 0
    for k in range(N):
        assert out[k][0] == at
        at += out[k][1]
        assert sum(bt.n_loci for bt, _ in out[k][2]) == out[k][1]
    assert at == b.n_loci
    assert min(o[1] for o in out) > 0.2 * b.n_loci / N
    # column by column, locus by locus
    def cat(parts, f):
        return np.concatenate([f(bt, st) for bt, st in parts])
    sharded = [x for k in range(N) for x in out[k][2]]
    for name in whole[0][0].columns:
        assert np.array_equal(cat(whole, lambda bt, st: bt.columns[name].view(np.uint32) if bt.columns[name].dtype == np.float32 else bt.columns[name]),
                              cat(sharded, lambda bt, st: bt.columns[name].view(np.uint32) if bt.columns[name].dtype == np.float32 else bt.columns[name])), name
    assert np.array_equal(cat(whole, lambda bt, st: np.diff(bt.obs_offset)), cat(sharded, lambda bt, st: np.diff(bt.obs_offset)))
    for name in whole[0][0].locus:
        assert np.array_equal(cat(whole, lambda bt, st: bt.locus[name]), cat(sharded, lambda bt, st: bt.locus[name])), name
    assert np.array_equal(cat(whole, lambda bt, st: np.asarray(st.pos)), cat(sharded, lambda bt, st: np.asarray(st.pos)))
    # the shards together inflate the file once, plus lead-ins and tails (1 / 32 of a share each)
    assert t_shards["inflated_bytes"] < 1.25 * t_whole["inflated_bytes"], (t_shards["inflated_bytes"], t_whole["inflated_bytes"])
    # a shard whose neighbour reports another landing than its first start is refused (the check that catches a wrong guess)
    bad = np.stack(rows).copy()
    bad[1, 0, 2] += 1
    with pytest.raises(engine.EngineError, match="do not meet"):
        ingest.ObsReader(paths, device=0, shard=(1, N), gather=lambda mine: bad)


def test_summaries_of_deep_and_filtered_pileups(tmp_path):
    """Pileups of some hundred observations with repeated keys, runs of several prob_mapping values and loci that drop non-standard
    alignments (pileup.rs:26-43) are written from the summaries; a table with a FEW pileups of more than 1 024 observations keeps its
    summaries and the writer fetches the columns for it; a file of such pileups only turns the summaries off.  The calls text must be the
    host reader's, character by character."""
    from varlociraptor_amd import callsfmt
    from varlociraptor_amd.batch import PileupBatch

    def pileups(depth, n, seed):
        cfg = synth.config3()
        cfg.depth, cfg.max_depth = depth, 4000
        return cfg, synth.generate(cfg, n, seed=seed)
    cfg, mid = pileups(400.0, 40, 11)
    _, deep = pileups(1500.0, 40, 12)
    _, shallow = pileups(60.0, 300, 13)
    mixed = PileupBatch.concat([shallow.select(np.arange(150)), deep.select(np.arange(2)), shallow.select(np.arange(150, 300))])
    for tag, b, expect in (("mid", mid, (True, False, True)), ("mixed", mixed, (True, True, False)), ("deep", deep, (False, False, False))):
        # few distinct keys and runs: quantise the evidence columns and repeat the flags
        c = b.columns
        c["prob_alt"][:] = np.round(c["prob_alt"] * 2) / 2
        c["prob_ref"][:] = np.round(c["prob_ref"] * 2) / 2
        c["flags"][:] = c["flags"][(np.arange(b.n_obs) // 7) * 7 % b.n_obs]
        c["prob_mapping"][:] = np.where((np.arange(b.n_obs) // 5) % 2 == 0, np.float32(-0.001), np.float32(-0.25))
        third = np.where(np.arange(b.n_obs) % 4 == 0, (np.arange(b.n_obs) % 3) * 1234567, -1).astype(np.int32)
        paths = []
        for s in range(b.n_samples):
            p = str(tmp_path / ("%s_%d.bcf" % (tag, s)))
            ingest.write_observations(p, b, s, third_allele_evidence=third)
            paths.append(p)
        (hb, hsites), = _read_all(paths, None, 1 << 20)
        rd = ingest.ObsReader(paths, chunk_records=1 << 20, device=0, host_columns=False)
        db, dsites = rd.next()
        table = db.extra["native_table"]
        on, n_over = table.summaries()
        assert (on, n_over > 0) == expect[:2], (tag, on, n_over)
        plan = engine.Plan(cfg.scenario, device=0)
        res = plan.call_table_device(table, afd_capacity=16)
        names = cfg.scenario.out_names()
        header = callsfmt.header(names, cfg.scenario.sample_names, list(dsites.contig_names))
        a, h = str(tmp_path / (tag + "_dev.vcf")), str(tmp_path / (tag + "_host.vcf"))
        ingest.write_calls(a, header, table, res, names)
        assert table.summaries()[0] == expect[2], tag   # (the writer took the columns where a pileup was left to them)
        ref = plan.call_host(hb, afd_capacity=16)
        ingest.write_calls(h, header, hb.extra["native_table"], ref, names)
        assert open(a).read() == open(h).read()
        rd.close()
        plan.close()


def test_descriptor_mode_reports_a_damaged_member_chain(tmp_path, reader_mode):
    """A file whose member chain breaks in the middle (a BSIZE field that points nowhere, a file cut inside a member): the records in
    front of the damage are delivered, then the reader fails — with the stager's index (descriptor mode) like with the index pass over the
    mapping."""
    cfg = synth.config3()
    b = synth.generate(cfg, 400, seed=17)
    p = str(tmp_path / "a.bcf")
    ingest.write_observations(p, b, 0)
    raw = bytearray(open(p, "rb").read())
    offs, off = [], 0
    while off < len(raw):
        offs.append(off)
        off += struct.unpack_from("<H", raw, off + 16)[0] + 1
    assert len(offs) > 20
    # (a) the magic of a member in the middle
    bad = bytearray(raw)
    bad[offs[len(offs) // 2]] = 0x1e
    q = str(tmp_path / "magic.bcf")
    open(q, "wb").write(bad)
    # (b) the file cut inside a member
    c = str(tmp_path / "cut.bcf")
    open(c, "wb").write(raw[:offs[len(offs) // 2] + 40])
    for path in (q, c):
        with pytest.raises(engine.EngineError):
            _read_all([path], 0, 64)
