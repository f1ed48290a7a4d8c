"""`python -m varlociraptor_amd call variants generic --scenario S.yaml --obs name=path.vcf ... [> calls.vcf]`

Mirror of the reference's `call variants` surface (src/cli.rs:684-735) for text observation VCFs (format v15):
`generic` with a scenario YAML (grammar/mod.rs:129-144) and `tumor-normal --tumor --normal --purity`
(src/cli.rs:1151-1172).  Model evaluation runs on the GPU engine; there is no CPU path.
"""
from __future__ import annotations

import argparse
import sys
from typing import Dict, List

import numpy as np

from . import abi, callsfmt, engine, obsfmt
from .batch import CallResults
from .scenario import Contamination, Inheritance, Sample, Scenario, Species, tumor_normal


def _contig_value(defn, contig: str, what: str):
    """PloidyDefinition / UniverseDefinition (grammar/mod.rs:280-312, 503-523): a plain value or a contig map with `all`."""
    if isinstance(defn, dict):
        if contig in defn:
            return defn[contig]
        if "all" in defn:
            return defn["all"]
        raise ValueError("%s for contig %r not found and no 'all' entry" % (what, contig))  # Ploidy/UniverseContigNotFound
    return defn


def _num(v):
    """YAML 1.1 (PyYAML) reads `1e-3` as a string, serde_yaml (the reference) as a float: accept both."""
    return None if v is None else float(v)


def scenario_from_yaml(path: str, contig: str = "all") -> Scenario:
    """Scenario as `Caller::configure_model` sees it on `contig` (calling.rs:632-718): contig maps of universes and
    ploidies and sex-specific species ploidies (grammar/mod.rs:314-345) are resolved here."""
    import yaml
    with open(path) as fh:
        y = yaml.safe_load(fh)
    species = None
    sp = y.get("species") or None
    if sp:
        vf = sp.get("variant-fractions", {}) or {}
        species = Species(heterozygosity=_num(sp.get("heterozygosity")), germline_mutation_rate=_num(sp.get("germline-mutation-rate")),
                          somatic_effective_mutation_rate=_num(sp.get("somatic-effective-mutation-rate")), ploidy=None,
                          fraction_indel=vf.get("indel", 0.0125), fraction_mnv=vf.get("mnv", 0.001), fraction_sv=vf.get("sv", 0.01))
    samples: Dict[str, Sample] = {}
    for name, sd in y["samples"].items():
        sd = sd or {}
        cont = sd.get("contamination")
        inh = sd.get("inheritance")
        inheritance = None
        if inh:
            if "mendelian" in inh:
                inheritance = Inheritance(abi.INHERIT_MENDELIAN, tuple(inh["mendelian"]["from"]))
            elif "clonal" in inh:
                inheritance = Inheritance(abi.INHERIT_CLONAL, (inh["clonal"]["from"],), bool(inh["clonal"]["somatic"]))
            elif "subclonal" in inh:
                inheritance = Inheritance(abi.INHERIT_SUBCLONAL, (inh["subclonal"]["from"],))
        universe = sd.get("universe")
        if universe is not None:
            universe = _contig_value(universe, contig, "universe")
        # Sample::contig_ploidy (grammar/mod.rs:581-593): the sample's own definition wins over the species'
        ploidy = sd.get("ploidy")
        if ploidy is not None:
            ploidy = int(_contig_value(ploidy, contig, "ploidy"))
        elif sp and sp.get("ploidy") is not None:
            pd = sp["ploidy"]
            if isinstance(pd, dict) and set(pd) <= {"male", "female"}:  # SexPloidyDefinition::Specific
                sex = sd.get("sex")
                if sex is None:
                    raise ValueError("sex specific ploidy definition found but no sex specified in sample %r" % name)
                if sex not in pd:
                    raise ValueError("ploidy definition for %s not found" % sex)
                pd = pd[sex]
            ploidy = int(_contig_value(pd, contig, "ploidy"))
        samples[name] = Sample(
            resolution=float(sd.get("resolution", 0.01)), universe=universe,
            contamination=Contamination(cont["by"], float(cont["fraction"])) if cont else None, ploidy=ploidy,
            somatic_effective_mutation_rate=_num(sd.get("somatic-effective-mutation-rate")),
            germline_mutation_rate=_num(sd.get("germline-mutation-rate")), inheritance=inheritance)
    sc = Scenario(samples, dict(y["events"]), species=species, expressions=dict(y.get("expressions") or {}))
    sc.validate()  # Scenario::vaftrees -> validate (grammar/mod.rs:206-279): OverlappingEvents
    return sc


# model_mode of the reference (calling.rs:413-418): (check_read_orientation_bias, check_read_position_bias, check_softclip_bias,
# check_homopolymer_artifact_detection).  check_strand_bias and the alt-locus check are NOT part of it: a precise indel and an
# imprecise SV of one contig share a model, its `last_rid` and therefore the variant-specific prior of the contig's first record.
MODEL_MODE_MASK = abi.BIAS_ORIENTATION | abi.BIAS_POSITION | abi.BIAS_SOFTCLIP | abi.BIAS_HOMOPOLYMER


def model_modes(locus_flags):
    import numpy as np
    return (np.asarray(locus_flags) & MODEL_MODE_MASK).astype(int)


def _scenario_signature(sc: Scenario):
    return tuple((n, s.universe, s.ploidy) for n, s in sc.samples.items()) + (sc.variant_heterozygosity_ln, sc.variant_somatic_effective_mutation_rate_ln)


class CallChunk:
    """What a CallProcessor sees of one chunk of records, in input order: the pileups (`batch`), the site columns (`sites`:
    contig, pos, ref / alt through sites.ref(l) ...), the results (`results`: ln_posterior[l][k] for out_names[k], map_vaf,
    map_bias, status, AFD lists) and `loci`, the indices of these records within the chunk the reader delivered."""

    def __init__(self, batch, sites, results, out_names, sample_names, loci, offset=0):
        self.batch, self.sites, self.results, self.out_names, self.sample_names, self.loci = batch, sites, results, out_names, sample_names, loci
        self.offset = offset   # records of the file(s) delivered before this chunk: offset + loci = record numbers


class CallProcessor:
    """Plug point of the driver, calling.rs:964-975 (`CallProcessor{setup, process_call, finalize}`): what is done with the calls.
    The default (processor=None) is the reference's CallWriter (calling.rs:977-1006): the calls file.  `estimate contamination`
    (estimation/contamination.rs:371-399) plugs a collector in here.  process_calls is the batched form of process_call: one
    invocation per chunk of the streaming reader, records in input order."""

    def setup(self, out_names, sample_names):
        return None

    def process_calls(self, chunk: "CallChunk"):
        raise NotImplementedError

    def finalize(self):
        return None


class CandidateFilter:
    """calling.rs:1008-1020 (`CandidateFilter::filter(work_item, sample_names) -> bool`), vectorised over a chunk: returns a boolean
    array, True = the record is processed (evaluated and handed to the processor), False = skipped as in calling.rs:409."""

    def filter(self, batch, sites, sample_names):
        import numpy as np
        return np.ones(batch.n_loci, bool)


class ContaminationCandidateFilter(CandidateFilter):
    """estimation/contamination.rs:404-428: SNVs whose contaminant pileup has >= 10 observations that are all ref support
    (prob_ref > prob_alt, read_observation.rs:439-441) and whose sample pileup has >= 10 observations with at least one strong
    alt support (Bayes factor alt:ref above 20, read_observation.rs:429-432)."""

    def filter(self, batch, sites, sample_names):
        import numpy as np
        S = batch.n_samples
        ci, si = list(sample_names).index("contaminant"), list(sample_names).index("sample")
        off = batch.obs_offset.astype(np.int64)
        pa, pr = batch.columns["prob_alt"].astype(np.float64), batch.columns["prob_ref"].astype(np.float64)
        ref_support = pr > pa
        with np.errstate(invalid="ignore"):   # (-inf) - (-inf)
            strong_alt = (pa - pr) > np.log(20.0)
        cs = np.concatenate([[0], np.cumsum(~ref_support)])
        sa = np.concatenate([[0], np.cumsum(strong_alt)])
        L = batch.n_loci
        c0, c1 = off[np.arange(L) * S + ci], off[np.arange(L) * S + ci + 1]
        s0, s1 = off[np.arange(L) * S + si], off[np.arange(L) * S + si + 1]
        has_snv = (batch.locus["locus_flags"] & abi.LOCUS_HAS_SNV) != 0
        return has_snv & (c1 - c0 >= 10) & (cs[c1] - cs[c0] == 0) & (s1 - s0 >= 10) & (sa[s1] - sa[s0] > 0)


class _PinnedResults:
    """Result buffers of one chunk carved out of recycled page-locked blocks (engine.host_array): the AFD lists of a chunk are hundreds of
    megabytes — allocating and first-touching them per chunk costs more than the kernel, and from page-locked memory the device
    copies them by direct DMA."""

    def __init__(self):
        import threading
        self.free, self.lock = [], threading.Lock()

    def results(self, n_loci, n_out, n_samples, afd_capacity, afd_text_capacity=0):
        need = (n_loci * (8 * (n_out + 1 + n_samples) + abi.N_BIAS + 8) + (n_loci * n_samples * (4 + 16 * afd_capacity) if afd_capacity else 0)
                + (afd_text_capacity + n_loci * n_samples * 8 if afd_capacity else 0) + 64 * 16)
        block = None
        with self.lock:
            for i, b in enumerate(self.free):
                if b.size >= need:
                    block = self.free.pop(i)
                    break
            if block is None and len(self.free) >= 4:
                self.free.pop(0)
        if block is None:
            block = engine.host_array(int(need * 1.15) + 4096, np.uint8)
        at = [0]

        def alloc(shape, dtype):
            n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
            off = at[0]
            at[0] = (off + n + 63) & ~63
            return block[off:off + n].view(dtype).reshape(shape)
        res = CallResults(n_loci, n_out, n_samples, afd_capacity, alloc=alloc, afd_text_capacity=afd_text_capacity)
        res._pool_block = block
        return res

    def release(self, res):
        block = getattr(res, "_pool_block", None)
        if block is not None:
            res._pool_block = None
            with self.lock:
                self.free.append(block)


_RESULT_POOL = None


def _shared_result_pool():
    global _RESULT_POOL
    if _RESULT_POOL is None:
        _RESULT_POOL = _PinnedResults()
    return _RESULT_POOL


def _fixed_fields(res):
    """Copy of the fixed-size result fields (the AFD lists stay with the chunk's own buffers)."""
    out = CallResults(res.n_loci, res.n_out, res.n_samples, 0)
    for f in ("ln_posterior", "ln_marginal", "map_vaf", "map_bias", "best_event", "status"):
        getattr(out, f)[...] = getattr(res, f)
    return out


def _part_path(output: str, rank: int) -> str:
    """Part file of a sharded run (ends in .bcf: the native writer picks the container by the suffix)."""
    return "%s.part%d.bcf" % (output, rank)


class _CrossesShards(Exception):
    """a chunk of a sharded run holds records whose evaluation needs records of other shards"""


def call_variants(scenario, obs_paths: Dict[str, str], omit_mask: int = 0, afd_capacity: int = 128, out=sys.stdout,
                  device: int = 0, output: str = None, ingest: str = None, timings: dict = None,
                  processor: "CallProcessor" = None, candidate_filter: "CandidateFilter" = None, _allow_shards: bool = True):
    """`scenario`: a Scenario, or a callable contig -> Scenario (contig-specific universes / ploidies: one plan per
    distinct resolution, as the reference re-configures its model on contig change, calling.rs:343-356).
    `processor` / `candidate_filter`: the driver's two plug points (calling.rs:964-1020); with a processor no calls file is
    written (the processor IS the consumer, as CallWriter is in the reference); a candidate filter needs a processor."""
    if candidate_filter is not None and processor is None:
        raise ValueError("a candidate filter comes with its own call processor (calling.rs:1022-1040: call_generic takes both)")
    per_contig = scenario if callable(scenario) else (lambda contig: scenario)
    scen: Dict[str, Scenario] = {}

    # The reference keeps one model (and one `last_rid`) per model mode = the tuple of check_* flags of the work item
    # (calling.rs:414-443): each mode installs the variant-specific prior of ITS first record on a contig.
    first_of_contig: Dict[tuple, tuple] = {}

    def resolve(contig, mode=0):
        key = (contig, mode)
        if key not in scen:
            sc = per_contig(contig)
            if callable(scenario) or key in first_of_contig:
                import copy
                sc = copy.copy(sc)
            # variant-specific priors are installed with the contig's model, from its first record (calling.rs:643-713)
            het, som = first_of_contig.get(key, (None, None))
            if het is not None:
                sc.variant_heterozygosity_ln = het
            if som is not None:
                sc.variant_somatic_effective_mutation_rate_ln = som
            for name in sc.sample_names:
                if name not in obs_paths:
                    raise SystemExit("no observations given for sample %r" % name)
            for name in obs_paths:
                if name not in sc.sample_names:
                    raise SystemExit("invalid observation sample name %r" % name)  # errors::Error::InvalidObservationSampleName
            scen[key] = sc
        return scen[key]

    # samples are ordered by name (BTreeMap, grammar/mod.rs:137), independent of the contig
    sample_order = scenario.sample_names if not callable(scenario) else sorted(obs_paths)
    paths = [obs_paths[name] for name in sample_order if name in obs_paths]
    if not callable(scenario):
        resolve("all")
        scen.clear()
    import numpy as np
    import os
    import time
    from .batch import CallResults
    native = (ingest or os.environ.get("VLR_INGEST", "native")) == "native"
    if processor is not None and not native:
        raise ValueError("call processors plug into the native streaming driver (VLR_INGEST=native)")
    world, rank = 1, 0
    try:
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            world, rank = tdist.get_world_size(), tdist.get_rank()
    except ImportError:
        pass
    # Sharded front door (several ranks, device reader, BCF output): every rank inflates, decodes, evaluates and WRITES its own
    # contiguous share of the records (ingest.ObsReader(shard=...): about 1 / N of the members of every file per rank) and rank 0 puts
    # the parts of the calls file together — no rank ever holds the whole file, nothing but a few counters crosses between the ranks.
    # Set below, once the reader is open; VLR_INGEST_SHARDED=0 keeps the older path (every rank reads everything, results all-gathered).
    shard_state = {"on": False}
    plans: Dict[tuple, "engine.Plan"] = {}   # one plan per scenario signature, kept across the chunks of a run
    reserve_loci = [0]
    FIELDS = ("ln_posterior", "ln_marginal", "map_vaf", "map_bias", "best_event", "status", "afd_count", "afd_vaf", "afd_lnprob")
    # (one pool per process: page-locking its blocks again for every run costs tens of milliseconds)
    result_pool = _shared_result_pool() if (native and processor is None and rank == 0 and world == 1 and output) else None

    # Breakend events whose first record sat in an EARLIER chunk of the streaming reader: the reference hands the first breakend's
    # event probabilities and sample infos to every later record of the event across the whole file (calling.rs:569-580,
    # 726-741); key = vlr_obs_sites.group_key, value = that record's result row
    carried: Dict[int, dict] = {}

    def evaluate(batch, contig_names, contig_of, het, som, group_rep, group_key=None):
        """Results of one batch of records (the whole file, or one chunk of the streaming reader)."""
        L = batch.n_loci
        modes = model_modes(batch.locus["locus_flags"]) if L else np.zeros(0, np.int64)
        key = contig_of * 256 + modes  # (contig, model mode): one model, one `last_rid`, one variant-specific prior (calling.rs:413-443)
        for k_, first in zip(*np.unique(key, return_index=True)):
            first_of_contig.setdefault((contig_names[int(k_) // 256] if contig_names else "all", int(k_) % 256), (
                None if het[first] != het[first] else float(het[first]), None if som[first] != som[first] else float(som[first])))
        # breakends of one event share a pileup and a result: evaluate the first record of every event only and copy its
        # event probabilities / sample info to the others (calling.rs:569-580, 726-741, 820-839)
        reps = np.nonzero(group_rep == np.arange(L))[0]
        groups: Dict[tuple, List] = {}
        sig_scenario: Dict[tuple, Scenario] = {}
        for k_ in np.unique(key[reps]) if len(reps) else []:
            sc = resolve(contig_names[int(k_) // 256] if contig_names else "all", int(k_) % 256)
            sig = _scenario_signature(sc)
            sig_scenario.setdefault(sig, sc)
            groups.setdefault(sig, []).append(reps[key[reps] == k_])
        res, names = None, None
        for sig, parts in groups.items():
            loci = np.sort(np.concatenate(parts))
            sc = sig_scenario[sig]
            n_out_, S_ = sc.n_out, len(sc.sample_names)
            if world > 1 and not shard_state["on"]:
                # loci shard across the ranks (one process per GPU); the results are reassembled by one all-gather of
                # fixed-size records (+ one for the AFD lists), every rank ends up with the full result
                from . import dist as vdist
                lo, hi = vdist.shard_range(len(loci), rank, world)
                mine = loci[lo:hi]
            else:
                lo, hi, mine = 0, len(loci), loci
            if len(mine):
                if sig not in plans:
                    while len(plans) >= 4:  # a plan owns device buffers (staging slots, scratch rows, AFD log): keep a handful
                        plans.pop(next(iter(plans))).close()
                    plans[sig] = engine.Plan(sc, device=device)
                    if native and reserve_loci[0]:   # size the plan's buffers once for the reader's request size: no growth (hipFree + hipMalloc) between chunks
                        plans[sig].reserve(reserve_loci[0], afd_capacity)
                plan = plans[sig]
                table = batch.extra.get("native_table") if getattr(batch, "extra", None) else None
                on_dev = table is not None and getattr(table, "on_device", False)
                if on_dev and len(mine) != L:
                    # several models (or ranks) in one chunk: the sub-batches are cut from the HOST columns, which a device reader with
                    # detached copies may still be filling — wait for them before the first select
                    table.fetch_columns()
                sub = batch if len(mine) == L else batch.select(mine)
                # observation files are already capped by preprocess's --max-depth: size the LDS budget to the deepest record
                # (deeper records than the LDS holds take the deep launch); the depth comes from the offsets, never from the columns
                plan.fit_max_obs(sub.obs_offset)
                if len(mine) == L and on_dev:
                    # the columns were decoded on the device (device reader): nothing to stage but the results, which land in recycled
                    # page-locked memory when the calls writer is the only consumer (it hands the block back after the chunk is written)
                    # the FORMAT/AFD text is written on the device (vlr_results.afd_text, 8 bytes per slot of the lists' capacity; a list
                    # that does not fit comes down as numbers) unless results are copied between records afterwards — breakend events
                    # (group fan-out, rows carried across chunks) index the arrays.  VLR_AFD_TEXT=0: numbers only.
                    text_cap = 0
                    if (afd_capacity and len(reps) == L and (group_key is None or not np.any(group_key)) and os.environ.get("VLR_AFD_TEXT", "1") != "0"):
                        text_cap = min(L * S_ * 8 * afd_capacity, 0xffff0000)
                    # (device text only where the native calls writer is the sole consumer of the lists — the pooled path and the sharded
                    #  writer: a list that was formatted on the device is not copied down as numbers, and a caller of call_variants() that
                    #  gets the CallResults of a single chunk back reads afd_vaf / afd_lnprob)
                    buf = (result_pool.results(L, n_out_, S_, afd_capacity, text_cap) if result_pool is not None
                           else (CallResults(L, n_out_, S_, afd_capacity, afd_text_capacity=text_cap) if (text_cap and shard_state["on"]) else None))
                    r = plan.call_table_device(table, afd_capacity=afd_capacity, results=buf)
                else:
                    r = plan.call_host(sub, afd_capacity=afd_capacity)
            else:
                r = CallResults(0, n_out_, S_, afd_capacity)
            if world > 1 and not shard_state["on"]:
                from . import dist as vdist
                r = vdist.gather_call_results(r, lo, hi, len(loci), n_out_, S_, afd_capacity)
            if names is None:
                names = sc.out_names()
            if len(loci) == L:
                res = r
                break
            if res is None:
                res = CallResults(L, r.n_out, r.n_samples, afd_capacity)
            for f in FIELDS:
                a = getattr(res, f)
                if a is not None:
                    a[loci] = getattr(r, f)
        if res is not None and len(reps) < L:  # fan the group results out to every record of the group
            for f in FIELDS:
                a = getattr(res, f)
                if a is not None:
                    a[:] = a[group_rep]
        if res is not None and group_key is not None and L:
            grouped = np.nonzero(group_key != 0)[0]
            if len(grouped):
                for l in grouped:
                    k_ = int(group_key[l])
                    row = carried.get(k_)
                    if row is None:   # first record of the event in the file (a representative of this chunk)
                        carried[k_] = {f: np.array(getattr(res, f)[l]) for f in FIELDS if getattr(res, f) is not None}
                    else:             # the event began in an earlier chunk: its first record's result
                        for f, v in row.items():
                            getattr(res, f)[l] = v
        if res is not None and (rank == 0 or shard_state["on"]):
            # the reference panics on NaN (assert!(!p.is_nan())): say so instead of writing `.` silently
            hard = res.status & (abi.LOCUS_NAN | abi.LOCUS_UNDERFLOW | abi.LOCUS_TABLE_FULL | abi.LOCUS_TOO_DEEP)
            for bit, what in ((abi.LOCUS_NAN, "a likelihood became NaN"), (abi.LOCUS_UNDERFLOW, "an observation likelihood is outside the f64 range"),
                              (abi.LOCUS_TABLE_FULL, "visited-point table overflow"), (abi.LOCUS_TOO_DEEP, "pileup above the LDS budget and the deep pool")):
                n_bad = int(((hard & bit) != 0).sum())
                if n_bad:
                    print("warning: %d record(s) without a result: %s" % (n_bad, what), file=sys.stderr)
        return res, names

    def close_plans():
        for p_ in plans.values():
            p_.close()
        plans.clear()

    def header_for(names, contigs):
        scenario0 = resolve(contigs[0] if contigs else "all")
        return callsfmt.header(names or scenario0.out_names(), scenario0.sample_names, sorted(set(contigs))), scenario0

    t_begin = time.perf_counter()
    if native:
        # product path: BGZF inflate, BCF/VCF parse, the v15 decoder and the calls writer in native code (csrc/vlr_ingest.cpp), a
        # bounded number of records at a time; reader, evaluation and writer of consecutive chunks overlap (three threads: the
        # native calls release the GIL)
        import queue
        import threading
        from . import ingest as vingest
        is_text = any(not (p_.endswith(".bcf") or p_.endswith(".bcf.gz")) for p_ in paths)
        chunk = int(os.environ.get("VLR_CLI_CHUNK", "0")) or (1 << 62 if is_text else 16384)  # text VCF: contigs are only known at the end
        reader = None
        if not is_text and os.environ.get("VLR_INGEST_HOST", "0") == "0":
            # BGZF inflate, record split and v15 decode as kernels (csrc/vlr_inflate.hip, csrc/vlr_decode.hip): the compressed members
            # cross PCIe, the columns are born in device memory and the evaluation reads them there.  Files it does not read (plain
            # gzip, uncompressed BCF) go to the host reader.
            want_shards = (world > 1 and processor is None and candidate_filter is None and bool(output) and str(output).endswith(".bcf")
                           and os.environ.get("VLR_INGEST_SHARDED", "1") != "0" and _allow_shards)
            try:
                # records per request: 32 768, and 65 536 for inputs above a gigabyte (tools/cli_sweep.sh, 1 M records per step: 1.66 -> 1.75 M
                # records/s; 131 072: 1.42 M — too few chunks for the three stages to overlap —, 16 384: 1.25 M)
                big = sum(os.path.getsize(p_) for p_ in paths) >= (1 << 30)
                reader = vingest.ObsReader(paths, omit_bias_mask=omit_mask, chunk_records=int(os.environ.get("VLR_CLI_CHUNK", "0")) or (65536 if big else 32768), device=device,
                                           shard=(rank, world) if want_shards else None,
                                           # the observation columns stay on the device and the calls writer takes the OBS text, the SAOBS / SROBS letters
                                           # and the DP runs of every pileup from obs_text_kernel (vlr_obs_reader_set_host_columns(0)); a processor, a
                                           # candidate filter and the unsharded multi-rank path read the host columns.  VLR_INGEST_SUMMARIES=0: columns.
                                           host_columns=(processor is not None or candidate_filter is not None or not (world == 1 or want_shards)
                                                         or os.environ.get("VLR_INGEST_SUMMARIES", "1") == "0"),
                                           # the evaluation reads the device side of a table; only the writer (which waits) needs the host copy of the columns
                                           # (several ranks: every rank inflates and decodes the files on its own device instead of sharing the
                                           # node's CPUs between N host readers; its shard is cut from the host copy of the columns)
                                           async_columns=(processor is None and candidate_filter is None and (world == 1 or want_shards)))
                shard_state["on"] = want_shards
            except engine.EngineError as ex:
                if ex.code != abi.ERR_UNSUPPORTED:
                    raise
        if reader is None:
            reader = vingest.ObsReader(paths, omit_bias_mask=omit_mask, chunk_records=chunk)
        else:
            reserve_loci[0] = reader.chunk_records
        q_in: "queue.Queue" = queue.Queue(maxsize=int(os.environ.get("VLR_CLI_QUEUE", "2")))
        q_out: "queue.Queue" = queue.Queue(maxsize=int(os.environ.get("VLR_CLI_QUEUE", "2")))
        stage = {"read_s": 0.0, "call_s": 0.0, "write_s": 0.0, "n_loci": 0, "n_obs": 0}
        errors: List[BaseException] = []

        stop = threading.Event()

        def read_loop():
            try:
                while not stop.is_set():
                    t0 = time.perf_counter()
                    item = reader.next()
                    stage["read_s"] += time.perf_counter() - t0
                    while not stop.is_set():
                        try:
                            q_in.put(item, timeout=0.2)
                            break
                        except queue.Full:
                            pass
                    if item is None:
                        return
            except BaseException as ex:  # noqa: BLE001 (handed to the main thread)
                errors.append(ex)
                q_in.put(None)

        writer_state = {"w": None, "tmp": None}

        def write_loop():
            try:
                while True:
                    item = q_out.get()
                    if item is None:
                        return
                    table, res_, names_, contigs_ = item
                    t0 = time.perf_counter()
                    if writer_state["w"] is None:
                        hdr, _ = header_for(names_, contigs_)
                        target = output
                        if not target:
                            import tempfile
                            writer_state["tmp"] = tempfile.TemporaryDirectory()
                            target = os.path.join(writer_state["tmp"].name, "calls.vcf")
                        if shard_state["on"]:
                            target = _part_path(output, rank)
                        writer_state["w"] = vingest.CallsWriter(target, hdr, part=(rank, world) if shard_state["on"] else None)
                        writer_state["path"] = target
                    writer_state["w"].append(table, res_, list(names_))
                    if result_pool is not None:
                        result_pool.release(res_)
                    stage["write_s"] += time.perf_counter() - t0
            except BaseException as ex:  # noqa: BLE001
                errors.append(ex)
                while q_out.get() is not None:
                    pass

        proc_state = {"setup": False}
        tr = threading.Thread(target=read_loop, daemon=True)
        tw = threading.Thread(target=write_loop, daemon=True) if ((rank == 0 or shard_state["on"]) and processor is None) else None
        stage["setup_s"] = time.perf_counter() - t_begin
        tr.start()
        if tw:
            tw.start()
        collected = []
        names = None
        used_contigs: List[str] = []
        loop_exc: List[BaseException] = []   # sharded run: what went wrong on THIS rank (the others must hear of it before anybody waits)
        try:
          try:
              while True:
                  item = q_in.get()
                  if item is None or errors:
                      break
                  batch, sites = item
                  t0 = time.perf_counter()
                  contig_names = list(sites.contig_names)
                  contig_of = np.asarray(sites.contig, np.int64)
                  het_, som_ = batch.extra["prior_het_ln"], batch.extra["prior_som_ln"]
                  grep_, gkey_ = np.asarray(batch.extra["group_representative"], np.int64), np.asarray(batch.extra["group_key"], np.uint64)
                  if shard_state["on"] and ((gkey_ != 0).any() or np.isfinite(np.asarray(het_, np.float64)).any() or np.isfinite(np.asarray(som_, np.float64)).any()):
                      # breakend events hand the FIRST record's result to the later ones, and per-variant prior overrides are installed from
                      # the first record of a contig (calling.rs:569-580, 643-713): both reach across shard boundaries.  This rank stops
                      # here; after the collective below ALL ranks drop their parts and take the file again on the unsharded path
                      # (every rank reads everything, results all-gathered), which carries both.
                      raise _CrossesShards()
                  loci_ = np.arange(batch.n_loci)
                  ebatch = batch
                  if candidate_filter is not None:   # calling.rs:409: work items the filter rejects are not processed at all
                      keep = np.asarray(candidate_filter.filter(batch, sites, sample_order), bool)
                      if not keep.all():
                          loci_ = np.nonzero(keep)[0]
                          ebatch = batch.select(loci_)
                          contig_of, het_, som_, gkey_ = contig_of[loci_], np.asarray(het_)[loci_], np.asarray(som_)[loci_], gkey_[loci_]
                          # representatives among the records that are left: the first kept record of every group
                          grep_ = np.arange(len(loci_))
                          first_of: Dict[int, int] = {}
                          for j_, k_ in enumerate(gkey_):
                              if k_:
                                  grep_[j_] = first_of.setdefault(int(k_), j_)
                  res, nm = (evaluate(ebatch, contig_names, contig_of, het_, som_, grep_, gkey_) if ebatch.n_loci else (None, None))
                  names = names or nm
                  stage["call_s"] += time.perf_counter() - t0
                  stage["n_loci"] += batch.n_loci
                  stage["n_obs"] += batch.n_obs
                  if not used_contigs:
                      used_contigs = contig_names if not is_text else [contig_names[int(c_)] for c_ in np.unique(np.asarray(sites.contig))]
                  collected.append(_fixed_fields(res) if (result_pool is not None and getattr(res, "_pool_block", None) is not None) else res)
                  if processor is not None:
                      if res is not None and rank == 0:
                          if not proc_state["setup"]:
                              processor.setup(list(names), list(sample_order))
                              proc_state["setup"] = True
                          processor.process_calls(CallChunk(ebatch, sites, res, list(names), list(sample_order), loci_, stage["n_loci"] - batch.n_loci))
                  elif tw and res is not None:
                      q_out.put((batch.extra["native_table"], res, names, used_contigs))
          except BaseException as ex:  # noqa: BLE001
            if not shard_state["on"]:
                raise
            loop_exc.append(ex)
        finally:
            t_loop_end = time.perf_counter()
            if tw:
                q_out.put(None)
                tw.join()
            stage["drain_writer_s"] = time.perf_counter() - t_loop_end   # the writer's last chunk(s)
            stop.set()
            tr.join()
            if shard_state["on"] and os.environ.get("VLR_INGEST_SHARD_REPORT"):
                # what this rank read (tests, tools): its record range and the bytes it inflated
                import json
                with open("%s.%d" % (os.environ["VLR_INGEST_SHARD_REPORT"], rank), "w") as fh_:
                    json.dump({"rank": rank, "world": world, "first_record": reader.first_record, "n_records": reader.n_records,
                               "total_records": getattr(reader, "total_records", None), "device_reader": vingest.device_timings()}, fh_)
            t_c0 = time.perf_counter()
            reader.close()
            stage["drain_reader_close_s"] = time.perf_counter() - t_c0
            t_c0 = time.perf_counter()
            close_plans()
            stage["drain_plans_close_s"] = time.perf_counter() - t_c0
        if shard_state["on"]:
            # One collective tells every rank whether ALL ranks came through and what the file's header holds (a rank that
            # fails alone leaves the others at the barrier below until the RCCL timeout; a rank without records cannot know the contigs
            # and output names the other parts index).  Nothing is written unless every rank succeeded; parts are removed otherwise.
            import torch.distributed as tdist
            crosses = bool(loop_exc) and isinstance(loop_exc[0], _CrossesShards)
            mine_failed = bool(errors or loop_exc) and not crosses
            info: List = [None] * world
            tdist.all_gather_object(info, (mine_failed, list(names) if names else None, list(used_contigs), crosses))
            if any(i_[3] for i_ in info) and not any(i_[0] for i_ in info):
                # records that reach across shard boundaries (seen by at least one rank): the whole file again, unsharded
                if writer_state["w"] is not None:
                    writer_state["w"].close()
                try:
                    os.remove(_part_path(output, rank))
                except OSError:
                    pass
                if rank == 0:
                    print("note: breakend events or per-variant prior overrides in the input: %d ranks read the whole file (results all-gathered) "
                          "instead of a share each" % world, file=sys.stderr)
                tdist.barrier()
                return call_variants(scenario, obs_paths, omit_mask=omit_mask, afd_capacity=afd_capacity, out=out, device=device, output=output,
                                     ingest=ingest, timings=timings, processor=processor, candidate_filter=candidate_filter, _allow_shards=False)
            if any(i_[0] for i_ in info):
                if writer_state["w"] is not None:
                    try:
                        writer_state["w"].close()
                    except Exception:  # noqa: BLE001 (already failing)
                        pass
                try:
                    os.remove(_part_path(output, rank))
                except OSError:
                    pass
                if loop_exc and not crosses:
                    raise loop_exc[0]
                if errors:
                    raise errors[0]
                raise SystemExit("rank(s) %s of the sharded run failed: no calls file was written" % ", ".join(str(k_) for k_, i_ in enumerate(info) if i_[0]))
            # the same header on every rank: names and contigs of the first rank that saw records
            names = next((i_[1] for i_ in info if i_[1]), names)
            used_contigs = next((i_[2] for i_ in info if i_[2]), used_contigs)
        if errors:
            raise errors[0]
        if processor is not None:
            if rank == 0:
                if not proc_state["setup"]:
                    processor.setup(list(names or resolve(used_contigs[0] if used_contigs else "all").out_names()), list(sample_order))
                processor.finalize()
        elif shard_state["on"]:
            # every rank closes its part (a rank without records writes an empty one: the header, on rank 0); rank 0 assembles the file
            if writer_state["w"] is None:
                hdr, _ = header_for(names, used_contigs)
                vingest.CallsWriter(_part_path(output, rank), hdr, part=(rank, world)).close()
            else:
                writer_state["w"].close()
            tdist.barrier()
            if rank == 0:
                vingest.concat_parts(output, [_part_path(output, k_) for k_ in range(world)])
            tdist.barrier()
        elif rank == 0:
            if writer_state["w"] is None:  # no records at all: the header alone
                hdr, _ = header_for(names, used_contigs)
                if output:
                    vingest.CallsWriter(output, hdr).close()
                else:
                    print(hdr, file=out)
            else:
                writer_state["w"].close()
                if not output:
                    with open(writer_state["path"]) as fh:
                        out.write(fh.read())
                    writer_state["tmp"].cleanup()
        if timings is not None:
            timings.update(dict(stage, wall_s=time.perf_counter() - t_begin, chunks=len(collected), drain_s=time.perf_counter() - t_loop_end))
        collected = [r_ for r_ in collected if r_ is not None]
        if len(collected) == 1:
            return collected[0]
        if not collected:
            return None
        # several chunks: the fixed-size fields concatenated (the AFD lists went to the file chunk by chunk)
        tot = CallResults(sum(r_.n_loci for r_ in collected), collected[0].n_out, collected[0].n_samples, 0)
        o = 0
        for r_ in collected:
            for f in FIELDS[:6]:
                getattr(tot, f)[o:o + r_.n_loci] = getattr(r_, f)
            o += r_.n_loci
        return tot

    # the Python restatement of the same decoder and formatter (obsfmt.py / bcfio.py / callsfmt.py); the tests compare the two
    batch, sites = obsfmt.read_observation_vcf(paths, omit_bias_mask=omit_mask)
    contig_names = sorted(set(s_[0] for s_ in sites))
    cidx = {c: i for i, c in enumerate(contig_names)}
    contig_of = np.array([cidx[s_[0]] for s_ in sites], np.int64)
    pri = batch.extra.get("prior_overrides") or []
    het = np.array([np.nan if p_[0] is None else p_[0] for p_ in pri], np.float64)
    som = np.array([np.nan if p_[1] is None else p_[1] for p_ in pri], np.float64)
    reps_, source_ = obsfmt.haplotype_groups(batch.extra.get("haplotype") or [None] * batch.n_loci)
    group_rep = np.asarray(reps_, np.int64)[np.asarray(source_, np.int64)] if batch.n_loci else np.zeros(0, np.int64)
    L = batch.n_loci
    t_read = time.perf_counter()
    try:
        res, names = evaluate(batch, contig_names, contig_of, het, som, group_rep)
    finally:
        close_plans()
    t_call = time.perf_counter()
    header, scenario0 = header_for(names, contig_names if L else [])
    names = names or scenario0.out_names()
    if rank != 0:
        return res
    if output and output.endswith(".bcf"):  # binary calls file (reference: bcf::Writer, calling.rs:296-304)
        from .bcfio import BcfWriter
        with BcfWriter(output, header) as wr:
            for l in range(L):
                wr.write_line(callsfmt.format_record(sites[l], batch, res, l, names, scenario0.sample_names))
    else:
        if output:
            out = open(output, "w")
        print(header, file=out)
        for l in range(L):
            print(callsfmt.format_record(sites[l], batch, res, l, names, scenario0.sample_names), file=out)
        if output:
            out.close()
    if timings is not None:
        t_end = time.perf_counter()
        timings.update({"read_s": t_read - t_begin, "call_s": t_call - t_read, "write_s": t_end - t_call, "n_loci": L, "n_obs": batch.n_obs, "wall_s": t_end - t_begin, "chunks": 1})
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(prog="varlociraptor_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    call = sub.add_parser("call").add_subparsers(dest="what", required=True)
    variants = call.add_parser("variants")
    for flag, bit in (("--omit-strand-bias", abi.BIAS_STRAND), ("--omit-read-orientation-bias", abi.BIAS_ORIENTATION),
                      ("--omit-read-position-bias", abi.BIAS_POSITION), ("--omit-softclip-bias", abi.BIAS_SOFTCLIP),
                      ("--omit-homopolymer-artifact-detection", abi.BIAS_HOMOPOLYMER), ("--omit-alt-locus-bias", abi.BIAS_ALTLOCUS)):
        variants.add_argument(flag, action="store_const", const=bit, default=0)
    variants.add_argument("--full-prior", action="store_true")
    variants.add_argument("--device", type=int, default=0)
    variants.add_argument("--output", help="calls file (.bcf = BCF2, anything else text VCF; default stdout)")
    mode = variants.add_subparsers(dest="mode", required=True)
    g = mode.add_parser("generic")
    g.add_argument("--scenario", required=True)
    g.add_argument("--obs", nargs="+", required=True, metavar="NAME=PATH")
    t = mode.add_parser("tumor-normal")
    t.add_argument("--tumor", required=True)
    t.add_argument("--normal", required=True)
    t.add_argument("--purity", type=float, required=True)
    fc = sub.add_parser("filter-calls").add_subparsers(dest="what", required=True)
    cf = fc.add_parser("control-fdr")  # cli.rs FilterMethod::ControlFDR
    cf.add_argument("calls")
    cf.add_argument("--events", nargs="+", required=True)
    cf.add_argument("--fdr", type=float, required=True)
    cf.add_argument("--mode", choices=["local-smart", "local-strict", "global-smart", "global-strict"], default="local-smart")
    cf.add_argument("--smart-retain-artifacts", action="store_true")
    cf.add_argument("--var", choices=["SNV", "MNV", "INS", "DEL", "BND", "INV", "DUP", "REP"])
    cf.add_argument("--minlen", type=int)
    cf.add_argument("--maxlen", type=int)
    cf.add_argument("--device", default="cpu")
    cf.add_argument("--output", "-o", help="BCF file for the kept records (default: a CHROM/POS/ID/REF/ALT table on stdout)")
    a = ap.parse_args(argv)
    import os
    if a.cmd == "call" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # one process per GPU (torchrun): `nccl` is RCCL on ROCm; VLR_DIST_BACKEND=gloo for hosts with fewer GPUs than ranks
        import torch
        import torch.distributed as tdist
        backend = os.environ.get("VLR_DIST_BACKEND", "nccl")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        a.device = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(a.device)
        tdist.init_process_group(backend, **({"device_id": torch.device("cuda", a.device)} if backend == "nccl" else {}))
    if a.cmd == "filter-calls":
        from . import fdr
        vartype = None
        if a.var:
            rng = (a.minlen or 0, a.maxlen if a.maxlen is not None else 1 << 62) if (a.minlen is not None or a.maxlen is not None) else None
            vartype = (a.var, rng)
        if a.output and str(a.device).startswith("cuda") and os.environ.get("VLR_INGEST", "native") != "python":
            # the whole command in the engine (vlr_calls_filter_fdr): BCF in, kept records out, threshold search on the device
            dev = int(str(a.device).split(":")[1]) if ":" in str(a.device) else 0
            kept_n, total_n = fdr.filter_calls_native(a.calls, a.output, a.events, a.fdr, vartype=vartype, local=a.mode.startswith("local"),
                                                      smart=a.mode.endswith("smart"), smart_retain_artifacts=a.smart_retain_artifacts, device=dev)
            print(f"{kept_n} of {total_n} records kept", file=sys.stderr)
            return
        from .bcfio import BcfReader
        r = BcfReader(a.calls)
        recs = list(r)
        tags = [l.split("ID=")[1].split(",")[0] for l in r.header_lines if l.startswith("##INFO") and "ID=PROB_" in l]
        kept = fdr.control_fdr(recs, a.events, a.fdr, vartype=vartype, local=a.mode.startswith("local"), smart=a.mode.endswith("smart"),
                               smart_retain_artifacts=a.smart_retain_artifacts, header_tags=tags, device=a.device)
        # utils::filter_calls (filtration/fdr.rs:58-62, utils/mod.rs:288-374): the kept records go out as BCF with the input's
        # header.  Records pass through in their original encoding (calls written by `call variants` carry one ALT per record,
        # so there are no ALT alleles to trim; a multi-ALT record is kept whole when any of its alleles is kept).
        if a.output:
            from .bcfio import BcfWriter
            with BcfWriter(a.output, r.header_text) as w:
                for rec in kept:
                    w.write_raw(rec["raw"])
        else:
            print("#CHROM\tPOS\tID\tREF\tALT")
            for rec in kept:
                print("\t".join(str(rec[k]) for k in ("chrom", "pos", "id", "ref", "alt")))
        print(f"{len(kept)} of {len(recs)} records kept", file=sys.stderr)
        return
    omit = (a.omit_strand_bias | a.omit_read_orientation_bias | a.omit_read_position_bias | a.omit_softclip_bias |
            a.omit_homopolymer_artifact_detection | a.omit_alt_locus_bias)
    if a.mode == "generic":
        full_prior = a.full_prior

        def sc(contig, _path=a.scenario):
            r = scenario_from_yaml(_path, contig)
            r.full_prior = full_prior
            return r
        obs = dict(kv.split("=", 1) for kv in a.obs)
    else:
        sc = tumor_normal(a.purity)
        obs = {"tumor": a.tumor, "normal": a.normal}
    if not callable(sc):
        sc.full_prior = a.full_prior
    call_variants(sc, obs, omit_mask=omit, device=a.device, output=a.output)


if __name__ == "__main__":
    main()
