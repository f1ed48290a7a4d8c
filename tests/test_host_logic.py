"""CPU tests of the host side: scenario/VAF-tree builder, codecs, synthetic generator, C-ABI library
loading (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.batch import PileupBatch
from varlociraptor_amd.dist import shard_range
from varlociraptor_amd.scenario import (Sample, Scenario, VAFRange, VAFSet, parse_formula, parse_universe, single_sample,
                                        tumor_normal)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parse_universe():
    u = parse_universe("[0.0,0.5[ | 0.5 | 1.0")
    assert u == [VAFSet((0.5,)), VAFSet((1.0,)), VAFRange(0.0, 0.5, False, True)]  # Sets sort before Ranges
    assert parse_universe("[0.0,1.0]") == [VAFRange(0.0, 1.0, False, False)]
    assert parse_universe("{0.0,0.5,1.0}") == [VAFSet((0.0, 0.5, 1.0))]
    with pytest.raises(ValueError):
        parse_universe("[0,1]")  # formula.pest: vaf needs a decimal point


def test_tumor_normal_trees():
    """normal = 0, tumor = 1; every event's root is the normal atom, child the tumor range."""
    sc = tumor_normal(0.75)
    assert sc.sample_names == ["normal", "tumor"]
    assert sc.event_names == ["germline_het", "germline_hom", "somatic_normal", "somatic_tumor"]
    d = sc.desc()
    assert d.n_events == 4 and d.n_nodes == 8
    for e in range(4):
        assert d.event_root_offset[e + 1] - d.event_root_offset[e] == 1
        root = d.nodes[d.root_index[d.event_root_offset[e]]]
        assert root.kind == abi.NODE_SAMPLE and root.sample == 0 and root.n_children == 1
        child = d.nodes[d.child_index[root.child_offset]]
        assert child.sample == 1 and child.vafs.kind == abi.SPECTRUM_RANGE and child.n_children == 0
        assert (child.vafs.start, child.vafs.end, child.vafs.left_exclusive, child.vafs.right_exclusive) == (0.0, 1.0, 1, 0)
    assert d.contaminated_by[1] == 0 and d.contaminated_by[0] == -1
    assert d.contamination_fraction[1] == pytest.approx(0.25)
    assert (d.resolution[0], d.resolution[1]) == (0.1, 0.01)


def test_missing_samples_are_added_from_the_universe():
    """grammar/vaftree.rs:246-295 add_missing_samples: one child per universe spectrum."""
    sc = Scenario({"a": Sample(universe="[0.0,1.0]"), "b": Sample(universe="[0.0,0.5[ | 0.5 | 1.0")}, {"ev": "a:0.5"})
    roots = sc.vaftree("ev")
    assert len(roots) == 1 and roots[0].sample == 0
    assert [type(c.vafs) for c in roots[0].children] == [VAFSet, VAFSet, VAFRange]
    assert all(c.sample == 1 for c in roots[0].children)


def test_formula_parser():
    f = parse_formula("(tumor:]0.0,1.0] & normal:0.0) | l2fc(tumor,normal) >= 1.5")
    assert type(f).__name__ == "Disj" and len(f.operands) == 2
    assert parse_formula("A>T").refbase == "A"
    assert type(parse_formula("!tumor:0.5")).__name__ == "Neg"
    assert type(parse_formula("$loh & tumor:0.5").operands[0]).__name__ == "ExprRef"
    with pytest.raises(ValueError):
        parse_formula("$ & tumor:0.5")


def test_negation_uses_the_universe():
    """Formula::negate on atoms (formula.rs:757-858): complement within the sample's universe."""
    sc = Scenario({"m": Sample(ploidy=2, germline_mutation_rate=1e-3), "t": Sample(universe="[0.0,1.0]")},
                  {"a": "!m:0.0 & t:0.5", "b": "!t:[0.2,0.5[ & m:0.0"},
                  species=__import__("varlociraptor_amd.scenario", fromlist=["Species"]).Species(heterozygosity=1e-3))
    roots = sc.vaftree("a")
    assert len(roots) == 1 and roots[0].sample == 0 and roots[0].vafs == VAFSet((0.5, 1.0))
    roots = sc.vaftree("b")
    # the normal form is a disjunction of conjunctions: one root per complement piece, each m:0.0 -> t:<piece>
    assert [r.sample for r in roots] == [0, 0]
    kids = [r.children[0] for r in roots]
    # VAFRange::split_at always makes the right part left-exclusive (formula.rs:1099-1129), so 0.5 itself is dropped
    assert sorted((k.vafs for k in kids), key=lambda v: v.start) == [VAFRange(0.0, 0.2, False, True), VAFRange(0.5, 1.0, True, False)]


def test_synth_is_deterministic_and_well_formed():
    cfg = synth.config3()
    a = synth.generate(cfg, 50, chunk=3)
    b = synth.generate(cfg, 50, chunk=3)
    for k in a.columns:
        assert np.array_equal(a.columns[k], b.columns[k], equal_nan=True)
    assert a.n_samples == 2 and a.n_loci == 50
    # realigned observations are normalised, SNV ones come from the base-quality table
    pa, pr = a.columns["prob_alt"].astype(np.float64), a.columns["prob_ref"].astype(np.float64)
    assert np.all(pa <= 0) and np.all(pr <= 0)
    # prob_mapping constant within a pileup (pileup-mean adjustment)
    for l in range(5):
        for s in range(2):
            sl = a.pileup_slice(l, s)
            assert np.unique(a.columns["prob_mapping"][sl]).size <= 1
    c = synth.generate(cfg, 50, chunk=4)
    assert not np.array_equal(a.columns["prob_alt"][:100], c.columns["prob_alt"][:100])


def test_minilogprob_round():
    x = np.array([-0.5, -10.5, -11.0, -1000.0, -np.inf, 0.0])
    r = synth.minilogprob_round(x)
    assert r[0] == np.float32(-0.5) and r[5] == 0.0 and np.isneginf(r[4])
    assert r[2] == -11.0 and r[3] == -1000.0  # exactly representable in f16
    assert r[1] == np.float32(np.float16(-10.5))


def test_batch_select_concat_roundtrip():
    cfg = synth.config2()
    a = synth.generate(cfg, 30)
    parts = [a.select(range(0, 10)), a.select(range(10, 30))]
    b = PileupBatch.concat(parts)
    assert np.array_equal(a.obs_offset, b.obs_offset)
    for k in a.columns:
        assert np.array_equal(a.columns[k], b.columns[k], equal_nan=True)
    e = a.select([])
    assert e.n_loci == 0 and e.n_obs == 0


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def test_node_shard_range_of_the_c_abi_is_the_harness_rule():
    """vlr_node_shard_range (pure arithmetic, no device) == dist.shard_range: the in-process node entry and the multi-process
    harness cut a batch at the same loci."""
    for n in (0, 1, 7, 8, 5003, 1000003):
        for w in (1, 2, 3, 4, 8):
            for k in range(w):
                assert engine.shard_range(n, w, k) == shard_range(n, k, w)
    with pytest.raises(engine.EngineError):
        engine.shard_range(10, 0, 0)


def test_node_create_without_a_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    with pytest.raises(engine.EngineError) as e:
        engine.Node(single_sample(0.01))
    assert e.value.code == abi.ERR_NO_DEVICE


# ---- C ABI: library loads, exports every symbol include/vlr.h declares, fails loudly without a GPU
def declared_functions():
    text = open(os.path.join(ROOT, "include", "vlr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vlr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_all_declared_symbols():
    engine.build()
    L = engine.lib()
    names = declared_functions()
    assert "vlr_plan_create" in names and "vlr_batch_run" in names and len(names) >= 10
    for n in names:
        assert hasattr(L, n), n
    assert sorted(engine.EXPORTS) == names
    assert L.vlr_abi_version() == abi.ABI_VERSION


def test_plan_create_validates_before_touching_the_device():
    L = engine.lib()
    sc = single_sample(0.01)
    d = sc.desc()
    d.resolution[0] = 1.5
    h = C.c_void_p()
    assert L.vlr_plan_create(C.byref(d), 0, C.byref(h)) == abi.ERR_INVALID_ARGUMENT
    assert b"resolution" in L.vlr_last_error()
    # invalid prior configuration: mendelian inheritance without germline mutation rate (prior.rs:812-816)
    from varlociraptor_amd.scenario import Inheritance, Species
    sc = Scenario({"c": Sample(inheritance=Inheritance(abi.INHERIT_MENDELIAN, ("m", "f"))), "m": Sample(), "f": Sample()},
                  {"e": "c:0.5"}, species=Species(heterozygosity=0.001, ploidy=2))
    d = sc.desc()
    assert L.vlr_plan_create(C.byref(d), 0, C.byref(h)) == abi.ERR_INVALID_PRIOR


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.EngineError) as ei:
        engine.Plan(single_sample(0.01))
    assert ei.value.code == abi.ERR_NO_DEVICE


def test_device_front_door_without_a_device_is_an_error_not_a_fallback(golden_dir):
    """vlr_obs_reader_open_device and vlr_bgzf_inflate have no host path behind them: without a GPU they fail (the host reader is a
    separate entry point the caller chooses); a file the device reader does not take is refused before any device call."""
    import torch
    from varlociraptor_amd import ingest
    d = os.path.join(golden_dir, "flamegraph_profiling")
    with pytest.raises(engine.EngineError) as ei:   # text VCF: not BGZF-compressed BCF, whatever the machine
        ingest.ObsReader([os.path.join(d, "normal.vcf")], device=0)
    assert ei.value.code == abi.ERR_UNSUPPORTED
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.EngineError) as ei:
        ingest.ObsReader([os.path.join(d, "normal.bcf")], device=0)
    assert ei.value.code in (abi.ERR_NO_DEVICE, abi.ERR_HIP)
    with pytest.raises(engine.EngineError) as ei:
        ingest.bgzf_inflate(open(os.path.join(d, "normal.bcf"), "rb").read())
    assert ei.value.code in (abi.ERR_NO_DEVICE, abi.ERR_HIP)
    # the host reader takes the same file
    assert sum(b.n_loci for b, _ in ingest.ObsReader([os.path.join(d, "normal.bcf")])) > 0


def test_bcf_reader_agrees_with_the_text_vcf(golden_dir):
    """The binary observation BCF of the reference's fixture decodes to exactly the batch of its text twin."""
    from varlociraptor_amd import obsfmt
    d = os.path.join(golden_dir, "flamegraph_profiling")
    a, sa = obsfmt.read_observation_vcf([os.path.join(d, "normal.vcf")])
    b, sb = obsfmt.read_observation_vcf([os.path.join(d, "normal.bcf")])
    assert sa == sb
    assert np.array_equal(a.obs_offset, b.obs_offset)
    for k in a.columns:
        assert np.array_equal(a.columns[k], b.columns[k], equal_nan=True), k
    for k in a.locus:
        assert np.array_equal(a.locus[k], b.locus[k]), k
    assert np.array_equal(a.extra["third_allele_evidence"], b.extra["third_allele_evidence"])


def test_bcf_reader_reads_the_reference_calls_file(golden_dir):
    from varlociraptor_amd.bcfio import BcfReader
    d = os.path.join(golden_dir, "flamegraph_profiling")
    recs = list(BcfReader(os.path.join(d, "calls.bcf")))
    assert len(recs) == 11 and recs[0]["pos"] == 10469 and recs[0]["alt"] == "<METH>"
    assert recs[0]["info"]["PROB_ABSENT"][0] == pytest.approx(285.541, rel=1e-6)
    assert recs[0]["info"]["PROB_PRESENT"][0] == 0.0


def test_bcf_writer_matches_reference_encoding(tmp_path):
    """Re-encoding the reference's calls.vcf text with the reference calls.bcf header gives records of identical
    layout (descriptor bytes, integer widths, string padding, dictionary indices); only f32 payload bytes may differ
    because the text carries 6 significant digits."""
    import struct
    from varlociraptor_amd.bcfio import BcfReader, BcfWriter
    g = os.path.join(os.path.dirname(__file__), "golden", "flamegraph_profiling")
    recs = [l for l in open(os.path.join(g, "calls.vcf")).read().split("\n") if l and not l.startswith("#")]
    ref = BcfReader(os.path.join(g, "calls.bcf"))
    path = str(tmp_path / "re.bcf")
    with BcfWriter(path, ref.header_text) as w:
        for r in recs:
            w.write_line(r)
    mine = BcfReader(path)

    def raw(r):
        p, out = r.pos, []
        while p + 8 <= len(r.buf):
            a, b = struct.unpack_from("<II", r.buf, p)
            out.append(r.buf[p:p + 8 + a + b])
            p += 8 + a + b
        return out
    A, B = raw(ref), raw(mine)
    assert len(A) == len(B) == 11
    for x, y in zip(A, B):
        assert len(x) == len(y)
        assert sum(1 for i in range(len(x)) if x[i] != y[i]) <= 6  # low bytes of the PROB_* floats
    a, b = list(ref), list(mine)
    for ra, rb in zip(a, b):
        assert ra["format"] == rb["format"] and ra["pos"] == rb["pos"] and ra["alt"] == rb["alt"]
        for k in ra["info"]:
            va, vb = ra["info"][k], rb["info"][k]
            assert all((x == y) or abs(x - y) <= 1e-3 * max(1.0, abs(x)) for x, y in zip(va, vb))


# ---- Formula::normalize: the reference's own unit tests (grammar/formula.rs:1601-1735) -----------------------------
def test_formula_range_conjunction():
    from varlociraptor_amd.scenario import Conj, Sample, Scenario, parse_formula
    sc = Scenario({"normal": Sample(resolution=0.01, universe="[0.0,1.0]")}, {"full": "normal:[0.0,1.0]"})
    conj = Conj([parse_formula("normal:[0.0,0.7]"), parse_formula("normal:[0.3,1.0]")])
    assert sc.canonical(conj) == sc.canonical("normal:[0.3,0.7]")
    assert sc.canonical(conj) != sc.canonical("normal:[0.0,1.0]")


def test_formula_nested_range_disjunction():
    from varlociraptor_amd.scenario import Sample, Scenario
    sc = Scenario({"normal": Sample(resolution=0.01, universe="[0.0,1.0]")}, {"full": "normal:[0.0,1.0]"})
    full = "(normal:[0.0, 0.25] | normal:[0.5,0.75]) | (normal:[0.25,0.5] | normal:[0.75,1.0]) | normal:[0.1,0.4] | normal:0.1"
    assert sc.canonical(full) == sc.canonical("normal:[0.0,1.0]")


def test_formula_two_separate_range_disjunctions():
    from varlociraptor_amd.scenario import Sample, Scenario
    sc = Scenario({"normal": Sample(resolution=0.01, universe="[0.0,1.0]")}, {"full": "normal:[0.0,1.0]"})
    full = "(normal:[0.0, 0.25] | normal:[0.5,0.6]) | ((normal:[0.25,0.5] | normal:[0.7,0.9]) | normal:[0.9,1.0]) | normal:]0.8,0.9[ | normal:0.75"
    assert sc.canonical(full) == sc.canonical("normal:[0.0,0.6] | normal:[0.7,1.0]")


def test_formula_merge_atoms_with_expressions_and_negation():
    from varlociraptor_amd import abi
    from varlociraptor_amd.scenario import Contamination, Inheritance, Sample, Scenario, Species
    sp = Species(heterozygosity=0.001, germline_mutation_rate=1e-3, somatic_effective_mutation_rate=None, ploidy=2)
    sc = Scenario(
        {"tumor": Sample(resolution=0.01, somatic_effective_mutation_rate=1e-6, inheritance=Inheritance(abi.INHERIT_CLONAL, ("normal",), False),
                         contamination=Contamination("normal", 0.11)),
         "normal": Sample(resolution=0.01, somatic_effective_mutation_rate=1e-10)},
        {"germline": "(normal:0.5 | normal:1.0) & !($loh | $loh_or_amplification)",
         "expected": "(normal:0.5 & tumor:{0.0, 0.5}) | (normal:0.5 & tumor:]0.0,0.5[) | (normal:0.5 & tumor:]0.5,0.9[) | normal:1.0"},
        species=sp, expressions={"loh": "normal:0.5 & tumor:1.0", "loh_or_amplification": "normal:0.5 & tumor:[0.9,1.0["})
    assert sc.canonical(sc.events["germline"]) == sc.canonical(sc.events["expected"])
    with pytest.raises(ValueError):
        sc.canonical("$undefined & normal:0.5")


def test_scenario_yaml_contig_and_sex_specific_definitions(tmp_path):
    """UniverseDefinition / PloidyDefinition maps and SexPloidyDefinition (grammar/mod.rs:280-345, 503-593)."""
    from varlociraptor_amd import cli
    y = tmp_path / "s.yaml"
    y.write_text("""
species:
  heterozygosity: 0.001
  ploidy:
    male: {all: 2, X: 1, Y: 1}
    female: {all: 2, X: 2, Y: 0}
samples:
  father: {sex: male}
  mother: {sex: female, ploidy: {all: 2, MT: 1}}
  tumor:
    sex: female
    resolution: 0.05
    universe: {all: "[0.0,1.0]", X: "{0.0,1.0}"}
events:
  e: "father:0.5"
""")
    sc = cli.scenario_from_yaml(str(y), "1")
    assert (sc.samples["father"].ploidy, sc.samples["mother"].ploidy, sc.samples["tumor"].universe) == (2, 2, "[0.0,1.0]")
    sc = cli.scenario_from_yaml(str(y), "X")
    assert (sc.samples["father"].ploidy, sc.samples["mother"].ploidy, sc.samples["tumor"].universe) == (1, 2, "{0.0,1.0}")
    assert cli.scenario_from_yaml(str(y), "Y").samples["father"].ploidy == 1
    assert cli.scenario_from_yaml(str(y), "MT").samples["mother"].ploidy == 1
    y.write_text("species: {ploidy: {male: 2, female: 2}}\nsamples: {a: {}}\nevents: {e: 'a:0.5'}\n")
    with pytest.raises(ValueError):
        cli.scenario_from_yaml(str(y), "1")  # sex specific ploidy but no sex in the sample
    y.write_text("samples: {a: {universe: {X: '[0.0,1.0]'}}}\nevents: {e: 'a:0.5'}\n")
    with pytest.raises(ValueError):
        cli.scenario_from_yaml(str(y), "1")  # UniverseContigNotFound


def test_haplotype_identifier_and_groups():
    """HaplotypeIdentifier::from (variants/model/mod.rs:87-133) and the breakend-group fan-out (calling.rs:569-580)."""
    from varlociraptor_amd.obsfmt import haplotype_groups, haplotype_identifier
    assert haplotype_identifier({"EVENT": "ev7", "MATEID": "b"}) == "ev7"
    assert haplotype_identifier({"MATEID": "bnd_U", "__ID": "bnd_W"}) == "bnd_U-bnd_W"
    assert haplotype_identifier({"SVTYPE": "BND"}) is None
    with pytest.raises(ValueError):
        haplotype_identifier({"MATEID": "x", "__ID": "."})
    reps, source = haplotype_groups([None, "e1", None, "e1", "e2", "e1", "e2"])
    assert reps == [0, 1, 2, 4] and source == [0, 1, 2, 1, 3, 1, 3]


def test_minilogprob_vectors_decode_the_same_on_the_strided_and_the_element_path():
    """obsfmt._vec_minilogprob: all-f16 and all-f32 vectors take a strided numpy view, mixed ones the element loop
    (bincode: u64 length, then per element a u32 variant tag + f16 | f32; utils/mod.rs:449-474)."""
    import struct
    from varlociraptor_amd import obsfmt
    rng = np.random.default_rng(3)
    vals = (-rng.random(37) * 50).astype(np.float32)

    def enc(tags):
        b = struct.pack("<Q", len(vals))
        for v, t in zip(vals, tags):
            b += struct.pack("<I", t) + (np.float16(v).tobytes() if t == 0 else struct.pack("<f", v))
        return b

    want16 = vals.astype(np.float16).astype(np.float32)
    assert np.array_equal(obsfmt._vec_minilogprob(enc([0] * 37)), want16)
    assert np.array_equal(obsfmt._vec_minilogprob(enc([1] * 37)), vals)
    tags = rng.integers(0, 2, 37)
    tags[0], tags[1] = 0, 1
    mixed = obsfmt._vec_minilogprob(enc(tags))
    assert np.array_equal(mixed, np.where(tags == 0, want16, vals))
    assert len(obsfmt._vec_minilogprob(struct.pack("<Q", 0))) == 0
    assert np.array_equal(obsfmt._vec_enum(struct.pack("<Q", 3) + struct.pack("<III", 2, 0, 8)), np.array([2, 0, 8], dtype=np.uint32))


def test_model_mode_has_only_the_four_reference_flags():
    """calling.rs:413-418: the model (and with it the variant-specific prior of a contig's first record) is keyed by the
    orientation / position / softclip / homopolymer checks only; strand and alt-locus checks follow `is_precise` and must not
    split a contig's records into two models."""
    import numpy as np
    from varlociraptor_amd import abi, cli
    precise_indel = abi.BIAS_STRAND | abi.BIAS_HOMOPOLYMER | abi.BIAS_ALTLOCUS
    imprecise_sv = abi.BIAS_HOMOPOLYMER | abi.BIAS_ALTLOCUS
    snv = abi.BIAS_ALL
    m = cli.model_modes(np.array([precise_indel, imprecise_sv, snv, imprecise_sv & ~abi.BIAS_ALTLOCUS], dtype=np.uint8))
    assert m[0] == m[1] == m[3] and m[2] != m[0]
    assert cli.MODEL_MODE_MASK == 0x1E


def test_bench_reference_binary_probe(tmp_path, monkeypatch):
    """bench.py times the reference's own `call variants` when a varlociraptor binary is on the box: absent ->
    None; present (here: a stand-in script that only checks its command line) -> one process per shard over observation BCFs
    written from the same batch."""
    import stat
    import sys
    sys.path.insert(0, ROOT)
    import bench
    cfg = synth.config3()
    batch = synth.generate(cfg, 40, seed=3)
    monkeypatch.setenv("PATH", str(tmp_path))
    assert bench.reference_binary_baseline(cfg, batch, 40, 4) is None
    exe = tmp_path / "varlociraptor"
    exe.write_text("#!/bin/sh\nif [ \"$1\" = --version ]; then echo 'varlociraptor 8.9.3'; exit 0; fi\n"
                   "[ \"$1 $2 $3\" = 'call variants tumor-normal' ] || exit 3\n"
                   "for a in \"$@\"; do case $a in *.bcf) [ -s \"$a\" ] || exit 4;; esac; done\nprintf BCF\n")
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + ":/usr/bin:/bin")
    r = bench.reference_binary_baseline(cfg, batch, 40, 4)
    assert r["kind"] == "reference" and r["value"] > 0 and r["cores"] == 4 and r["version"] == "varlociraptor 8.9.3"
