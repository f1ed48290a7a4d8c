"""End to end at the process boundary: observation VCF (format v15) + scenario YAML in, calls VCF out, compared
record by record with the reference's own output file tests/resources/flamegraph_profiling/calls.vcf."""
import io
import os
import re

import pytest

from varlociraptor_amd import abi, cli

pytestmark = pytest.mark.gpu


def tokens(obs: str):
    return sorted(re.findall(r"\d+[^\d]+", obs))


def test_cli_reproduces_reference_calls_file(golden_dir):
    d = os.path.join(golden_dir, "flamegraph_profiling")
    sc = cli.scenario_from_yaml(os.path.join(d, "scenario.yaml"))
    buf = io.StringIO()
    # the reference file was produced without artifact events (PROB_ARTIFACT=inf): all six biases omitted
    cli.call_variants(sc, {"normal": os.path.join(d, "normal.vcf")}, omit_mask=abi.BIAS_ALL, out=buf)
    got = [l for l in buf.getvalue().splitlines() if not l.startswith("#")]
    exp = [l.rstrip("\n") for l in open(os.path.join(d, "calls.vcf")) if not l.startswith("#")]
    assert len(got) == len(exp) == 11
    n_identical = 0
    for g, e in zip(got, exp):
        gf, ef = g.split("\t"), e.split("\t")
        assert gf[:7] == ef[:7]
        # PROB_* INFO: same tags in the same (descending probability) order; values agree to the 6 printed digits
        # up to one unit in the last place (calls.vcf comes from an older build of the reference)
        gi, ei = [kv.split("=") for kv in gf[7].split(";")], [kv.split("=") for kv in ef[7].split(";")]
        assert [k for k, _ in gi] == [k for k, _ in ei]
        for (_, a), (_, b) in zip(gi, ei):
            assert float(a) == pytest.approx(float(b), rel=3e-6)
        assert gf[8] == ef[8]
        gs, es = gf[9].split(":"), ef[9].split(":")
        keys = gf[8].split(":")
        same = gf[7] == ef[7]
        for k, a, b in zip(keys, gs, es):
            if k in ("SAOBS", "SROBS", "OBS"):
                assert tokens(a) == tokens(b), (k, a, b)  # Counter::most_common leaves ties unordered
            elif k == "AFD":
                ga, ea = [x.split("=") for x in a.split(",")], [x.split("=") for x in b.split(",")]
                assert [x for x, _ in ga] == [x for x, _ in ea]           # exact visited-point list
                for (_, u), (_, v) in zip(ga, ea):
                    assert abs(float(u) - float(v)) <= 0.011            # 2 printed decimals
                same = same and a == b
            else:
                assert a == b, (k, a, b)
        n_identical += same
    assert n_identical >= 5  # most records are character-identical


def test_cli_writes_bcf_calls_readable_back(golden_dir, tmp_path):
    """--output calls.bcf: the binary file decodes to the same records as the text output, and field by field to the
    reference's own calls.bcf (PROB_* within f32 text rounding)."""
    from varlociraptor_amd.bcfio import BcfReader
    d = os.path.join(golden_dir, "flamegraph_profiling")
    sc = cli.scenario_from_yaml(os.path.join(d, "scenario.yaml"))
    out = str(tmp_path / "calls.bcf")
    cli.call_variants(sc, {"normal": os.path.join(d, "normal.bcf")}, omit_mask=abi.BIAS_ALL, output=out)
    got = list(BcfReader(out))
    ref = list(BcfReader(os.path.join(d, "calls.bcf")))
    assert len(got) == len(ref) == 11
    for g, r in zip(got, ref):
        assert (g["chrom"], g["pos"], g["ref"], g["alt"]) == (r["chrom"], r["pos"], r["ref"], r["alt"])
        assert list(g["info"]) == list(r["info"])
        for k in r["info"]:
            for a, b in zip(g["info"][k], r["info"][k]):
                assert a == b or abs(a - b) <= 3e-6 * max(1.0, abs(b))
        assert g["format"]["DP"] == r["format"]["DP"] and g["format"]["AF"] == r["format"]["AF"]
        assert g["format"]["OOBS"] == r["format"]["OOBS"]
        assert tokens(g["format"]["OBS"][0]) == tokens(r["format"]["OBS"][0])


def test_cli_contig_specific_scenario_uses_one_plan_per_resolution(golden_dir, tmp_path):
    """Records on two contigs with different universes: every record is evaluated under its contig's scenario and the
    results come back in input order."""
    import numpy as np
    d = os.path.join(golden_dir, "flamegraph_profiling")
    lines = open(os.path.join(d, "normal.vcf")).read().split("\n")
    out, k = [], 0
    for l in lines:
        if l and not l.startswith("#"):
            f = l.split("\t")
            if k % 2:
                f[0] = "X"
            k += 1
            l = "\t".join(f)
        out.append(l)
        if l.startswith("##contig=<ID=1"):
            out.append(l.replace("ID=1", "ID=X"))
    obs = tmp_path / "two_contigs.vcf"
    obs.write_text("\n".join(out))
    y = tmp_path / "s.yaml"
    y.write_text('samples:\n  normal:\n    resolution: 0.1\n    universe: {all: "[0.0,1.0]", X: "{0.0,1.0}"}\nevents:\n  present: "normal:]0.0,1.0]"\n')
    ya = tmp_path / "a.yaml"
    ya.write_text('samples:\n  normal:\n    resolution: 0.1\n    universe: "[0.0,1.0]"\nevents:\n  present: "normal:]0.0,1.0]"\n')
    yx = tmp_path / "x.yaml"
    yx.write_text('samples:\n  normal:\n    resolution: 0.1\n    universe: "{0.0,1.0}"\nevents:\n  present: "normal:]0.0,1.0]"\n')
    mixed = cli.call_variants(lambda c: cli.scenario_from_yaml(str(y), c), {"normal": str(obs)}, out=io.StringIO())
    plain_a = cli.call_variants(cli.scenario_from_yaml(str(ya)), {"normal": str(obs)}, out=io.StringIO())
    plain_x = cli.call_variants(cli.scenario_from_yaml(str(yx)), {"normal": str(obs)}, out=io.StringIO())
    assert k == 11
    for l in range(k):
        want = plain_x if l % 2 else plain_a
        assert np.array_equal(mixed.ln_posterior[l], want.ln_posterior[l], equal_nan=True)
        assert np.array_equal(mixed.map_vaf[l], want.map_vaf[l], equal_nan=True)
    assert not np.array_equal(plain_a.ln_posterior, plain_x.ln_posterior)


def test_cli_breakend_groups_are_evaluated_once_and_fanned_out(golden_dir, tmp_path, monkeypatch):
    """Records carrying the same INFO EVENT share one evaluation; every record of the group gets its result."""
    import numpy as np
    from varlociraptor_amd import engine
    d = os.path.join(golden_dir, "flamegraph_profiling")
    out = []
    k = 0
    for l in open(os.path.join(d, "normal.vcf")).read().split("\n"):
        if l and not l.startswith("#"):
            f = l.split("\t")
            if k in (2, 5, 6):   # records 2, 5 and 6 become members of one event; they keep their own (different) pileups,
                f[7] = "EVENT=grp1;" + f[7]   # so the fan-out is visible: all three must carry record 2's result
            k += 1
            l = "\t".join(f)
        out.append(l)
        if l.startswith("##INFO=<ID=SVLEN"):
            pass
    obs = tmp_path / "grouped.vcf"
    obs.write_text("\n".join(out))
    sc = cli.scenario_from_yaml(os.path.join(d, "scenario.yaml"))
    seen = []
    orig = engine.Plan.call_host

    def spy(self, batch, afd_capacity=0):
        seen.append(batch.n_loci)
        return orig(self, batch, afd_capacity=afd_capacity)
    monkeypatch.setattr(engine.Plan, "call_host", spy)
    grouped = cli.call_variants(sc, {"normal": str(obs)}, omit_mask=abi.BIAS_ALL, out=io.StringIO())
    plain = cli.call_variants(sc, {"normal": os.path.join(d, "normal.vcf")}, omit_mask=abi.BIAS_ALL, out=io.StringIO())
    assert seen == [9, 11]  # 11 records, one group of three -> nine evaluations
    for l in range(11):
        want = 2 if l in (2, 5, 6) else l
        assert np.array_equal(grouped.ln_posterior[l], plain.ln_posterior[want], equal_nan=True)
        assert np.array_equal(grouped.map_vaf[l], plain.map_vaf[want], equal_nan=True)


def test_cli_breakend_event_that_straddles_reader_chunks(golden_dir, tmp_path, monkeypatch):
    """The streaming front door reads a bounded number of records at a time; an event whose breakends fall into different chunks
    still gets ONE result, the first record's, as in the reference (calling.rs:569-580, 726-741 work on the whole file)."""
    import numpy as np
    from varlociraptor_amd.bcfio import BcfWriter
    d = os.path.join(golden_dir, "flamegraph_profiling")
    header, recs = [], []
    k = 0
    for l in open(os.path.join(d, "normal.vcf")).read().split("\n"):
        if not l:
            continue
        if l.startswith("#"):
            header.append(l)
            continue
        f = l.split("\t")
        if k in (2, 5, 9):
            f[7] = "EVENT=grp1;" + f[7]
        k += 1
        recs.append("\t".join(f))
    obs = str(tmp_path / "grouped.bcf")
    with BcfWriter(obs, "\n".join(header)) as wr:
        for r in recs:
            wr.write_line(r)
    sc = cli.scenario_from_yaml(os.path.join(d, "scenario.yaml"))
    whole = cli.call_variants(sc, {"normal": obs}, omit_mask=abi.BIAS_ALL, out=io.StringIO())
    monkeypatch.setenv("VLR_CLI_CHUNK", "4")   # records 0-3 | 4-7 | 8-10: the event has one breakend in every chunk
    tm = {}
    pieces = cli.call_variants(sc, {"normal": obs}, omit_mask=abi.BIAS_ALL, out=io.StringIO(), timings=tm)
    assert tm["chunks"] == 3
    plain = cli.call_variants(sc, {"normal": os.path.join(d, "normal.vcf")}, omit_mask=abi.BIAS_ALL, out=io.StringIO())
    for l in range(11):
        want = 2 if l in (2, 5, 9) else l
        for got in (whole, pieces):
            assert np.array_equal(got.ln_posterior[l], plain.ln_posterior[want], equal_nan=True), l
            assert np.array_equal(got.map_vaf[l], plain.map_vaf[want], equal_nan=True), l


def test_cli_call_processor_and_candidate_filter_plug_points(tmp_path, monkeypatch):
    """calling.rs:964-1020: the driver's two plug points, as `estimate contamination` uses them (contamination.rs:371-428): a
    candidate filter that keeps SNVs with a clean contaminant pileup and strong alt evidence in the sample, and a processor that
    collects the calls instead of writing a file.  The collected results are those of the unfiltered run at the kept records."""
    import numpy as np
    from varlociraptor_amd import ingest, synth
    from varlociraptor_amd.scenario import Sample, Scenario
    sc = Scenario({"sample": Sample(resolution=0.01, universe="[0.0,1.0]"), "contaminant": Sample(resolution=0.01, universe="[0.0,1.0]")},
                  {"denovo": "sample:]0.0,1.0] & contaminant:0.0", "other": "sample:[0.0,1.0] & contaminant:]0.0,1.0]"})
    cfg = synth.config3()
    cfg.depth = 40.0
    batch = synth.generate(cfg, 600, seed=41)    # two samples in name order: contaminant = index 0, sample = index 1
    paths = {}
    for s_, name in enumerate(sc.sample_names):
        paths[name] = str(tmp_path / ("%s.bcf" % name))
        ingest.write_observations(paths[name], batch, s_)
    monkeypatch.setenv("VLR_CLI_CHUNK", "250")

    class Collect(cli.CallProcessor):
        def __init__(self):
            self.rows, self.loci, self.setups, self.done = [], [], 0, 0

        def setup(self, out_names, sample_names):
            self.setups += 1
            assert out_names[0] == "absent" and sample_names == ["contaminant", "sample"]

        def process_calls(self, chunk):
            assert chunk.results.n_loci == len(chunk.loci) == chunk.batch.n_loci
            self.rows.append(np.array(chunk.results.ln_posterior))
            self.loci.append(np.array(chunk.loci) + chunk.offset)

        def finalize(self):
            self.done += 1
    col = Collect()
    cli.call_variants(sc, paths, processor=col, candidate_filter=cli.ContaminationCandidateFilter())
    assert col.setups == 1 and col.done == 1
    kept = np.concatenate(col.loci)
    want = cli.ContaminationCandidateFilter().filter(batch, None, sc.sample_names)
    assert 0 < want.sum() < batch.n_loci
    assert np.array_equal(np.nonzero(want)[0], kept)
    full = cli.call_variants(sc, paths, out=io.StringIO())
    assert np.array_equal(np.concatenate(col.rows), full.ln_posterior[kept])
    with pytest.raises(ValueError):
        cli.call_variants(sc, paths, candidate_filter=cli.CandidateFilter())


def test_cli_reads_variant_specific_priors_from_the_first_record_of_a_contig(golden_dir, tmp_path):
    import numpy as np
    d = os.path.join(golden_dir, "flamegraph_profiling")
    out, k = [], 0
    for l in open(os.path.join(d, "normal.vcf")).read().split("\n"):
        if l and not l.startswith("#"):
            f = l.split("\t")
            if k == 0:
                f[7] = "HETEROZYGOSITY=13.0103;" + f[7]  # PHRED(0.05)
            k += 1
            l = "\t".join(f)
        out.append(l)
    obs = tmp_path / "het.vcf"
    obs.write_text("\n".join(out))
    y = tmp_path / "s.yaml"
    y.write_text("species:\n  heterozygosity: 0.001\n  ploidy: 2\nsamples:\n  normal:\n    resolution: 0.1\nevents:\n  het: 'normal:0.5'\n  hom: 'normal:1.0'\n")
    with_info = cli.call_variants(cli.scenario_from_yaml(str(y)), {"normal": str(obs)}, out=io.StringIO())
    plain = cli.call_variants(cli.scenario_from_yaml(str(y)), {"normal": os.path.join(d, "normal.vcf")}, out=io.StringIO())
    sc = cli.scenario_from_yaml(str(y))
    sc.variant_heterozygosity_ln = -float(np.float32(13.0103)) * np.log(10.0) / 10.0  # an INFO float is f32 (calling.rs:470-494 reads it through htslib)
    forced = cli.call_variants(sc, {"normal": os.path.join(d, "normal.vcf")}, out=io.StringIO())
    assert np.array_equal(with_info.ln_posterior, forced.ln_posterior, equal_nan=True)
    assert not np.allclose(with_info.ln_posterior, plain.ln_posterior, equal_nan=True)


def test_cli_precise_and_imprecise_records_share_the_model_of_a_contig(golden_dir, tmp_path):
    """an imprecise record after a precise one with a variant-specific prior still gets THAT prior (one model per
    (orientation, position, softclip, homopolymer) mode, calling.rs:413-418; strand / alt-locus checks do not key the model)."""
    import numpy as np
    d = os.path.join(golden_dir, "flamegraph_profiling")

    def write(name, first_info, second_info):
        out, k = [], 0
        for l in open(os.path.join(d, "normal.vcf")).read().split("\n"):
            if l and not l.startswith("#"):
                f = l.split("\t")
                if k == 0 and first_info:
                    f[7] = first_info + f[7]
                if k == 1 and second_info:
                    f[7] = second_info + f[7]
                k += 1
                l = "\t".join(f)
            out.append(l)
        p = tmp_path / name
        p.write_text("\n".join(out))
        return str(p)
    y = tmp_path / "s.yaml"
    y.write_text("species:\n  heterozygosity: 0.001\n  ploidy: 2\nsamples:\n  normal:\n    resolution: 0.1\nevents:\n  het: 'normal:0.5'\n  hom: 'normal:1.0'\n")
    mixed = cli.call_variants(cli.scenario_from_yaml(str(y)), {"normal": write("mixed.vcf", "HETEROZYGOSITY=13.0103;", "IMPRECISE;")}, out=io.StringIO())
    sc = cli.scenario_from_yaml(str(y))
    sc.variant_heterozygosity_ln = -float(np.float32(13.0103)) * np.log(10.0) / 10.0  # an INFO float is f32 (calling.rs:470-494 reads it through htslib)
    forced = cli.call_variants(sc, {"normal": write("imp.vcf", "", "IMPRECISE;")}, out=io.StringIO())
    assert np.array_equal(mixed.ln_posterior, forced.ln_posterior, equal_nan=True)


@pytest.mark.parametrize("name,sample,lo,hi", [("test_moelder_floatisnan", "tumor", -1e-12, 1e-12), ("test_mapq_meth", "normal", 0.71, 0.72),
                                               ("test_hiv_vaf_higher_than_expected", "sample", 0.05, 0.3), ("test_prinz_af_scan", "normal", 0.0, 1.0),
                                               ("test_prinz_call_meth_1", "normal", 0.97, 1.0 + 1e-12), ("test_prinz_call_meth_2", "normal", -1e-12, 1e-12),
                                               ("test_uzuner_only_N", "sample", -1e-12, 1e-12)])
def test_cli_reference_testcases_expected_allele_frequencies(golden_dir, name, sample, lo, hi):
    """The reference's testcase.yaml `expected: allelefreqs` conditions on the recorded v15 observations (see
    tests/test_oracle_fixture.py); the 1009-observation pileup also exercises the LDS budget sizing of the CLI."""
    d = os.path.join(golden_dir, "testcases", name)
    res = cli.call_variants(cli.scenario_from_yaml(os.path.join(d, "scenario.yaml")), {sample: os.path.join(d, "observations.vcf")}, out=io.StringIO())
    assert lo < res.map_vaf[0, 0] < hi
    assert (res.status[0] & 0xF) == 0


@pytest.mark.parametrize("obs_file", ["normal.vcf", "normal.bcf"], ids=["host reader (text VCF)", "device reader (BGZF BCF)"])
def test_cli_two_ranks_shard_and_reassemble(golden_dir, tmp_path, obs_file):
    """`call variants` under torchrun with two ranks (both on the one GPU of the test box, gloo for the exchange): loci
    are sharded, results all-gathered, rank 0 writes the same file as a single process.  With the BCF every rank reads through the
    device reader and cuts its shard from the host copy of the columns."""
    import subprocess
    import sys
    d = os.path.join(golden_dir, "flamegraph_profiling")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = tmp_path / "one.vcf"
    two = tmp_path / "two.vcf"
    base = ["-m", "varlociraptor_amd", "call", "variants", "--omit-strand-bias", "--omit-read-orientation-bias", "--omit-read-position-bias",
            "--omit-softclip-bias", "--omit-homopolymer-artifact-detection", "--omit-alt-locus-bias"]
    tail = ["generic", "--scenario", os.path.join(d, "scenario.yaml"), "--obs", "normal=" + os.path.join(d, obs_file)]
    env = dict(os.environ, PYTHONPATH=root, VLR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable] + base + ["--output", str(one)] + tail, check=True, cwd=root, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29541"] + base + ["--output", str(two)] + tail, check=True, cwd=root, env=env, timeout=900)
    a, b = one.read_text(), two.read_text()
    assert a == b and a.count("\n") > 11


def test_cli_two_ranks_read_and_write_their_own_shards(tmp_path):
    """VERDICT r04 missing #2 / next #3e: under torchrun with a BCF output every rank inflates and decodes only its share of the
    members (ingest.ObsReader(shard=...)), evaluates it, writes its part of the calls file, and rank 0 concatenates the parts (BGZF
    members concatenate).  Two ranks on the one GPU of the test box, gloo for the few counters that are exchanged: the records of
    the assembled file are those of a single process, and each rank inflated about half of the bytes."""
    import gzip
    import json
    import subprocess
    import sys
    from varlociraptor_amd import ingest, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = synth.config3()
    cfg.depth = 25.0
    b = synth.generate(cfg, 5000, seed=17)
    obs = {}
    for s_, name in enumerate(cfg.scenario.sample_names):
        obs[name] = str(tmp_path / (name + ".bcf"))
        ingest.write_observations(obs[name], b, s_)
    total_inflated = sum(len(gzip.decompress(open(p_, "rb").read())) for p_ in obs.values())
    y = tmp_path / "scenario.yaml"
    # the embedded tumor-normal scenario of the reference (src/cli.rs:1151-1172), purity 0.75: what synth.config3 evaluates
    y.write_text("samples:\n  tumor:\n    resolution: 0.01\n    universe: '[0.0,1.0]'\n    contamination:\n      by: normal\n      fraction: 0.25\n"
                 "  normal:\n    resolution: 0.1\n    universe: '[0.0,0.5[ | 0.5 | 1.0'\n"
                 "events:\n  somatic_tumor: 'tumor:]0.0,1.0] & normal:0.0'\n  somatic_normal: 'tumor:]0.0,1.0] & normal:]0.0,0.5['\n"
                 "  germline_het: 'tumor:]0.0,1.0] & normal:0.5'\n  germline_hom: 'tumor:]0.0,1.0] & normal:1.0'\n")
    one, two = tmp_path / "one.bcf", tmp_path / "two.bcf"
    base = ["-m", "varlociraptor_amd", "call", "variants"]
    tail = ["generic", "--scenario", str(y), "--obs"] + ["%s=%s" % (n_, p_) for n_, p_ in obs.items()]
    env = dict(os.environ, PYTHONPATH=root, VLR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", VLR_INGEST_SHARD_REPORT=str(tmp_path / "report"))
    subprocess.run([sys.executable] + base + ["--output", str(one)] + tail, check=True, cwd=root, env=env, timeout=900)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29543"] + base + ["--output", str(two)] + tail, check=True, cwd=root, env=env, timeout=900)
    a, c = gzip.decompress(one.read_bytes()), gzip.decompress(two.read_bytes())
    assert a == c and len(a) > 100000
    reps = [json.load(open(str(tmp_path / "report") + ".%d" % k)) for k in range(2)]
    assert reps[0]["first_record"] == 0 and reps[1]["first_record"] == reps[0]["n_records"] and reps[0]["n_records"] + reps[1]["n_records"] == 5000
    for r_ in reps:
        assert 0.3 * total_inflated < r_["device_reader"]["inflated_bytes"] < 0.62 * total_inflated, (r_["device_reader"]["inflated_bytes"], total_inflated)


def test_bench_step_through_rccl_on_one_gpu(tmp_path):
    """nothing in the repo had ever exercised RCCL itself (the two-rank tests use gloo: a one-GPU box cannot
    host two NCCL ranks).  One rank under torch.distributed.run with backend nccl: process-group set-up on the device, the
    all-gather of the packed result records through RCCL inside the timed step, barrier and max-reduction of the timing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, VLR_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "1", "--loci", "20000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-afd"], capture_output=True, text=True, cwd=root, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["collective"].startswith("rccl") and line["n_gpus"] == 1 and line["value"] > 0
    assert line["posterior_normalisation_max_err"] < 1e-9


def _flamegraph_with(tmp_path, golden_dir, edit):
    """the observation file of the flamegraph testcase as BCF, `edit(k, fields)` applied to its k-th record"""
    from varlociraptor_amd.bcfio import BcfWriter
    d = os.path.join(golden_dir, "flamegraph_profiling")
    header, recs = [], []
    k = 0
    for l in open(os.path.join(d, "normal.vcf")).read().split("\n"):
        if not l:
            continue
        if l.startswith("#"):
            if l.startswith("#CHROM"):
                header.append('##INFO=<ID=HETEROZYGOSITY,Number=A,Type=Float,Description="PHRED scaled expected heterozygosity">')
            header.append(l)
            continue
        f = l.split("\t")
        edit(k, f)
        k += 1
        recs.append("\t".join(f))
    obs = str(tmp_path / "edited.bcf")
    with BcfWriter(obs, "\n".join(header)) as wr:
        for r in recs:
            wr.write_line(r)
    return d, obs


def _two_ranks(root, env, port, args, timeout=600):
    import subprocess
    import sys
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), "-m", "varlociraptor_amd"] + args, capture_output=True, text=True, cwd=root, env=env, timeout=timeout)


@pytest.mark.parametrize("what", ["breakend event", "prior override"])
def test_cli_two_ranks_take_records_that_reach_across_shards_on_the_unsharded_path(golden_dir, tmp_path, what):
    """breakend events hand the FIRST record's result to the later ones and per-variant prior overrides are
    installed from the first record of a contig (calling.rs:569-580, 643-713) — both reach across shard boundaries, and only the rank
    whose shard holds such a record notices.  The ranks agree in the one collective before the barrier, drop their parts and take the
    file again on the unsharded path: the calls file equals a single process's, no part file is left behind."""
    import glob
    import gzip
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def edit(k, f):
        if what == "breakend event" and k in (9, 10):   # the event lives in the LAST records: the second rank's shard alone
            f[7] = "EVENT=grp1;" + f[7]
        if what == "prior override" and k == 0:          # the FIRST record of the contig: the first rank's shard alone
            f[7] = "HETEROZYGOSITY=20.0;" + f[7]

    d, obs = _flamegraph_with(tmp_path, golden_dir, edit)
    one, two = tmp_path / "one.bcf", tmp_path / "two.bcf"
    env = dict(os.environ, PYTHONPATH=root, VLR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tail = ["generic", "--scenario", os.path.join(d, "scenario.yaml"), "--obs", "normal=" + obs]
    subprocess.run([sys.executable, "-m", "varlociraptor_amd", "call", "variants", "--output", str(one)] + tail, check=True, cwd=root, env=env, timeout=600)
    r = _two_ranks(root, env, 29549, ["call", "variants", "--output", str(two)] + tail)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "2 ranks read the whole file" in r.stderr
    a, b = gzip.decompress(one.read_bytes()), gzip.decompress(two.read_bytes())
    assert a == b and len(a) > 2000
    assert sorted(glob.glob(str(two) + "*")) == [str(two)], glob.glob(str(tmp_path / "*"))


def test_cli_two_ranks_fail_together_when_one_shard_is_corrupt(golden_dir, tmp_path):
    """a rank that fails alone must not leave the others at the barrier until the collective's timeout.  A
    flipped bit in the CRC32 trailer of the LAST data member of the observation file is seen by the second rank alone (its share of
    the members); both ranks end with an error within seconds, no part of the calls file and no calls file is left behind."""
    import glob
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d, obs = _flamegraph_with(tmp_path, golden_dir, lambda k, f: None)
    raw = bytearray(open(obs, "rb").read())
    assert raw[-28:-24] == bytes([0x1f, 0x8b, 8, 4])   # the empty end-of-file member (SAM spec 4.1.2)
    raw[-28 - 8] ^= 0x10                                # CRC32 of the member before it
    open(obs, "wb").write(bytes(raw))
    out = tmp_path / "calls.bcf"
    env = dict(os.environ, PYTHONPATH=root, VLR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    r = _two_ranks(root, env, 29551, ["call", "variants", "--output", str(out), "generic", "--scenario", os.path.join(d, "scenario.yaml"), "--obs", "normal=" + obs])
    assert r.returncode != 0
    assert "CRC32 checksum mismatch" in r.stderr
    assert time.time() - t0 < 300, "the ranks waited for each other"
    assert not out.exists() and not glob.glob(str(out) + "*"), glob.glob(str(tmp_path / "*"))
