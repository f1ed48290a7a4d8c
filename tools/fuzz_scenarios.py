"""Randomised scenario fuzzing: GPU engine vs oracle on random event formulas (sets, ranges, negation, disjunctions,
l2fc terms, contamination) over small random pileups.  usage: python tools/fuzz_scenarios.py [n_scenarios] [seed]"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from oracle import oracle
from parity import compare, describe
from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.scenario import Contamination, Sample, Scenario

SPECTRA = ["0.0", "0.5", "1.0", "{0.0,0.5}", "{0.5,1.0}", "]0.0,1.0]", "]0.0,0.5[", "[0.5,1.0]", "]0.5,1.0[", "[0.0,1.0]", "]0.0,0.25]", "[0.1,0.4]", "]0.2,0.8["]


def random_scenario(rng):
    S = int(os.environ.get("FUZZ_S") or rng.choice([1, 2, 2, 3]))
    names = ["a", "b", "c", "d", "e", "f"][:S]
    samples = {}
    for i, n in enumerate(names):
        cont = None
        if S > 1 and i == 0 and rng.random() < 0.4:
            cont = Contamination(names[1], float(rng.choice([0.1, 0.25, 0.5])))
        samples[n] = Sample(resolution=float(rng.choice([0.1, 0.05, 0.02] if S < 3 else [0.1, 0.2])), universe="[0.0,1.0]", contamination=cont)

    def atom():
        n = names[int(rng.integers(S))]
        f = "%s:%s" % (n, SPECTRA[int(rng.integers(len(SPECTRA)))])
        return "!" + f if rng.random() < 0.15 else f

    def conj():
        k = int(rng.integers(1, S + 1))
        used, parts = set(), []
        for _ in range(k):
            a = atom()
            n = a.lstrip("!").split(":")[0]
            if n in used:
                continue
            used.add(n)
            parts.append(a)
        if S >= 2 and rng.random() < 0.2 and os.environ.get("FUZZ_NO_LFC") != "1":
            x, y = rng.choice(S, 2, replace=False)
            parts.append("l2fc(%s,%s) %s %s" % (names[x], names[y], rng.choice([">", ">=", "<", "<="]), rng.choice(["0.5", "1.0", "-1.0"])))
        return " & ".join(parts)

    events = {}
    # FUZZ_MANY_EVENTS=1: 33..48 events (masks of event groups beyond one word: the wide build of the kernels)
    for e in range(int(rng.integers(33, 49)) if os.environ.get("FUZZ_MANY_EVENTS") == "1" else int(rng.integers(2, 5))):
        f = conj()
        if rng.random() < 0.3:
            f = "(%s) | (%s)" % (f, conj())
        events[("ev%02d" if os.environ.get("FUZZ_MANY_EVENTS") == "1" else "ev%d") % e] = f
    return Scenario(samples, events), names


def random_scenario_prior(rng):
    """Ploidy-derived universes with germline / somatic / Mendelian / clonal priors (prior.rs), variant nodes, one fine
    resolution; mirrors the shapes of tests/resources/prior/scenarios."""
    from varlociraptor_amd.scenario import Inheritance, Species
    kind = int(rng.integers(4))
    species = Species(heterozygosity=float(rng.choice([0.001, 0.01])), germline_mutation_rate=1e-3, ploidy=2,
                      somatic_effective_mutation_rate=(1e-6 if rng.random() < 0.5 else None))
    if kind == 0:   # trio
        mend = Inheritance(abi.INHERIT_MENDELIAN, ("m", "f"))
        samples = {"m": Sample(), "f": Sample(), "k": Sample(inheritance=mend)}
        events = {"denovo": "(k:0.5 | k:1.0) & m:0.0 & f:0.0", "inherited": "!m:0.0 | !f:0.0", "het_all": "k:0.5 & m:0.5 & f:0.5"}
        names = ["f", "k", "m"]
    elif kind == 1:  # tumor/normal with somatic rates and clonal inheritance
        samples = {"n": Sample(somatic_effective_mutation_rate=1e-10, resolution=float(rng.choice([0.1, 0.05]))),
                   "t": Sample(somatic_effective_mutation_rate=1e-6, resolution=float(rng.choice([0.05, 0.02])),
                               inheritance=Inheritance(abi.INHERIT_CLONAL, ("n",), bool(rng.random() < 0.5)),
                               contamination=Contamination("n", float(rng.choice([0.1, 0.3]))) if rng.random() < 0.6 else None)}
        events = {"germline": "n:0.5 | n:1.0", "somatic_t": "n:0.0 & t:]0.0,1.0]", "somatic_n": "n:]0.0,0.5[", "loh": "n:0.5 & t:1.0"}
        if rng.random() < 0.5:
            del events["loh"]
        names = ["n", "t"]
    elif kind == 2:  # single sample, germline + somatic, fine resolution, variant nodes
        samples = {"s": Sample(somatic_effective_mutation_rate=1e-6, resolution=float(rng.choice([0.01, 0.02, 0.005])))}
        events = {"het": "s:0.5", "hom": "s:1.0", "low": "s:]0.0,0.5[", "high": "s:]0.5,1.0["}
        if rng.random() < 0.5:
            events = {"ct_het": "C>T & s:0.5", "other_het": "!C>T & s:0.5", "hom": "s:1.0", "sub": "s:]0.0,0.5[ | s:]0.5,1.0["}
        names = ["s"]
    else:           # subclonal relapse
        samples = {"n": Sample(somatic_effective_mutation_rate=1e-10, resolution=0.1),
                   "p": Sample(somatic_effective_mutation_rate=1e-6, resolution=0.05, inheritance=Inheritance(abi.INHERIT_CLONAL, ("n",), True)),
                   "r": Sample(somatic_effective_mutation_rate=1e-6, resolution=0.05, inheritance=Inheritance(abi.INHERIT_SUBCLONAL, ("p",)))}
        events = {"germline": "n:0.5 | n:1.0", "primary_only": "n:0.0 & p:]0.0,1.0] & r:0.0", "relapse": "n:0.0 & r:]0.0,1.0]"}
        names = ["n", "p", "r"]
    sc = Scenario(samples, events, species=species, full_prior=bool(rng.random() < 0.5))
    return sc, names


LAST_STATS = {}  # counters of the most recent main() run, for the tests: nothing may be skipped or excused silently


def main(argv=None):
    argv = sys.argv if argv is None else argv
    stats = {"generated": 0, "front_end_rejected": 0, "plans_rejected": 0, "run": 0, "mismatching": 0, "knife_edge_loci": 0, "flat_loci": 0, "afd_bad_lists": 0, "afd_map_tie_loci": 0}
    LAST_STATS.clear()
    LAST_STATS.update(stats)
    n_sc = int(argv[1]) if len(argv) > 1 else 50
    seed = int(argv[2]) if len(argv) > 2 else 1
    only = int(argv[3]) if len(argv) > 3 else -1
    prior_mode = os.environ.get("FUZZ_PRIOR") == "1"
    rng = np.random.default_rng(seed)
    bad = 0
    done = 0
    for it in range(n_sc):
        try:
            sc, names = (random_scenario_prior(rng) if prior_mode else random_scenario(rng))
            sc.desc()
        except Exception as ex:  # invalid formula for the front-end (e.g. empty spectrum): not a kernel case
            if prior_mode:
                print("scenario rejected:", ex)
            LAST_STATS["front_end_rejected"] += 1
            continue
        LAST_STATS["generated"] += 1
        S = len(names)
        classes = []
        for _ in range(4):
            classes.append(("c", 0.25, tuple((float(v), float(v + w)) for v, w in zip(rng.choice([0.0, 0.1, 0.5, 1.0], S), rng.choice([0.0, 0.0, 0.2], S)))))
        classes = [(l, f, tuple((lo, min(hi, 1.0)) for lo, hi in spec)) for l, f, spec in classes]
        if os.environ.get("FUZZ_TYPES") == "1":
            type_mix = {abi.VT_SNV: 0.4, abi.VT_MNV: 0.15, abi.VT_INDEL: 0.3, abi.VT_SV: 0.15}
            bias_mask = int(rng.choice([abi.BIAS_ALL, abi.BIAS_ALL, abi.BIAS_STRAND | abi.BIAS_ORIENTATION, abi.BIAS_POSITION | abi.BIAS_SOFTCLIP | abi.BIAS_HOMOPOLYMER, abi.BIAS_ALTLOCUS, 0]))
        else:
            type_mix, bias_mask = {abi.VT_SNV: 0.7, abi.VT_INDEL: 0.3}, abi.BIAS_ALL
        cfg = synth.SynthConfig(name="fuzz", config_id=50, scenario=sc, depth=float(os.environ.get("FUZZ_DEPTH") or rng.choice([4.0, 12.0, 30.0] if not prior_mode else [8.0, 25.0, 60.0])), type_mix=type_mix,
                                classes=classes, purity=None)
        b = synth.generate(cfg, 24, seed=int(rng.integers(1 << 30)), bias_mask=bias_mask)
        if only >= 0 and it != only:
            continue
        if only >= 0 and os.environ.get("FUZZ_EVENTS"):
            from varlociraptor_amd.scenario import Conj, Atom, Lfc, VAFSet, VAFRange, parse_formula
            env = dict(Conj=Conj, Atom=Atom, Lfc=Lfc, VAFSet=VAFSet, VAFRange=VAFRange, parse_formula=parse_formula, abi=abi)
            sc = Scenario(sc.samples, eval(os.environ["FUZZ_EVENTS"], env))
        try:
            plan = engine.Plan(sc)
        except Exception as ex:
            print("plan rejected:", ex, sc.events)
            LAST_STATS["plans_rejected"] += 1
            continue
        afd_cap = int(os.environ.get("FUZZ_AFD_CAP") or 256) if os.environ.get("FUZZ_AFD") == "1" else 0
        got = plan.call_host(b, afd_capacity=afd_cap)
        plan.close()
        ref = oracle.call(sc, b, afd_capacity=afd_cap, want_events=True)
        m = compare(got, ref, label="fuzz %d" % it)
        done += 1
        # flat-likelihood samples (at most three observations, typically neutralised by the singleton adjustment): every
        # VAF of that sample has the same joint up to rounding noise, so the MAP among them is arbitrary in the reference
        # too (HashMap order).  Tolerated when the posteriors agree and only such samples' MAP VAFs differ.
        pg, pr_ = np.exp(got.ln_posterior), np.exp(ref.ln_posterior)
        post_ok = np.nan_to_num(np.abs(pg - pr_), nan=0.0).max(axis=1) <= 1e-6
        dv = np.nan_to_num(np.abs(got.map_vaf - ref.map_vaf), nan=0.0)
        shallow = b.depth() <= 3
        flat = post_ok & np.all((dv <= 1e-6) | shallow, axis=1) & (got.best_event == ref.best_event)
        real_bad = [l for l in m["bad"] if not flat[l]]
        LAST_STATS["flat_loci"] += len(m["bad"]) - len(real_bad)
        # knife-edge loci: an argmax of the adaptive integrator sits on a (near-)tie, so the REFERENCE's own posterior jumps
        # between discrete levels when the inputs move by 1e-7 relative (seed 164 / scenario 31 / locus 4: levels 6e-5 and 3e-4
        # apart).  A posterior-only deviation smaller than that spread, with equal MAP and best event, is not a defect.
        knife = []
        for l in list(real_bad):
            if not (np.all(dv[l] <= 1e-6) and got.best_event[l] == ref.best_event[l]):
                continue
            dev = float(np.nan_to_num(np.abs(got.ln_posterior[l] - ref.ln_posterior[l]), nan=0.0, posinf=0.0).max())
            prng = np.random.default_rng(12345 + l)
            vals = []
            for _ in range(10):
                s2 = b.select([l])
                for col in ("prob_alt", "prob_ref", "prob_mapping"):
                    s2.columns[col] = (s2.columns[col].astype(np.float64) * (1 + 1e-7 * prng.standard_normal(s2.n_obs))).astype(np.float32)
                vals.append(oracle.call(sc, s2).ln_posterior[0])
            vals = np.nan_to_num(np.array(vals), nan=0.0, neginf=-1e300)
            spread = float(np.max(np.ptp(vals, axis=0)[np.isfinite(ref.ln_posterior[l])])) if np.any(np.isfinite(ref.ln_posterior[l])) else 0.0
            if spread >= dev and dev < 1e-2:
                knife.append(l)
        if knife:
            print("  knife-edge loci (reference chaotic under 1e-7 input perturbation):", knife)
            real_bad = [l for l in real_bad if l not in knife]
            LAST_STATS["knife_edge_loci"] += len(knife)
        ok = len(real_bad) == 0 and m["bias_equal"] and m["status_equal"]
        if ok and afd_cap:
            n_afd_bad = 0
            for l in range(b.n_loci):
                if flat[l] and l in m["bad"]:
                    continue
                # MAP on a near-tie: two visited points one rounding error apart (a bisection point and a tail point of the same
                # chain) have joints that differ in the last bits, and the engine's product-of-mantissas likelihood may order them
                # the other way round than the oracle's sum of logarithms.  The MAP VAFs then agree to 1e-6 (the parity bar) but
                # not bit for bit, and the lists — "operand sets EQUAL to the MAP in all other samples" — are those of another chain.
                if not np.array_equal(got.map_vaf[l], ref.map_vaf[l], equal_nan=True) and np.all(dv[l] <= 1e-9):
                    LAST_STATS["afd_map_tie_loci"] += 1
                    continue
                for si in range(S):
                    ng, nr = int(got.afd_count[l, si]), int(ref.afd_count[l, si])
                    if ng != nr or ng > afd_cap:
                        n_afd_bad += 1
                        if n_afd_bad <= 3:
                            print("  AFD count differs: locus %d sample %d gpu %d ref %d" % (l, si, ng, nr))
                        continue
                    og, orr = np.argsort(got.afd_vaf[l, si, :ng], kind="stable"), np.argsort(ref.afd_vaf[l, si, :nr], kind="stable")
                    if not (np.array_equal(got.afd_vaf[l, si, :ng][og], ref.afd_vaf[l, si, :nr][orr]) and
                            np.allclose(np.sort(got.afd_lnprob[l, si, :ng]), np.sort(ref.afd_lnprob[l, si, :nr]), atol=1e-6, equal_nan=True)):
                        n_afd_bad += 1
                        if n_afd_bad <= 3:
                            print("  AFD list differs: locus %d sample %d" % (l, si))
            LAST_STATS["afd_bad_lists"] += n_afd_bad
            if n_afd_bad:
                ok = False
                print("  AFD mismatches:", n_afd_bad)
                if only >= 0:
                    np.set_printoptions(precision=5, linewidth=220)
                    for l in range(b.n_loci):
                        if any(int(got.afd_count[l, si]) != int(ref.afd_count[l, si]) for si in range(S)):
                            print(" locus", l, "depth", b.depth()[l], "best", got.best_event[l], ref.best_event[l], "map", got.map_vaf[l], ref.map_vaf[l], "post", ref.ln_posterior[l])
                            for si in range(S):
                                ng, nr = int(got.afd_count[l, si]), int(ref.afd_count[l, si])
                                print("   s%d gpu" % si, np.sort(got.afd_vaf[l, si, :ng]))
                                print("   s%d ref" % si, np.sort(ref.afd_vaf[l, si, :nr]))
                            break
        if not ok:
            bad += 1
            print("MISMATCH", it, {k: (v.universe, v.resolution, v.contamination) for k, v in sc.samples.items()}, sc.events)
            print(describe(m))
            if only >= 0:
                np.set_printoptions(precision=6, linewidth=200)
                print("out names", sc.out_names(), "univ events", sc.event_names)
                for l in np.nonzero((got.map_bias != ref.map_bias).any(axis=1))[0][:4]:
                    print(" bias differs at locus", l, "depth", b.depth()[l], "got", got.map_bias[l], "ref", ref.map_bias[l], "map", got.map_vaf[l], ref.map_vaf[l],
                          "best", got.best_event[l], ref.best_event[l])
                    print("  got post", got.ln_posterior[l])
                    print("  ref post", ref.ln_posterior[l], "events", ref.event_ln_posterior[l])
                for l in m["bad"][:4]:
                    print(" locus", l, "depth", b.depth()[l], "status got %x ref %x" % (got.status[l], ref.status[l]))
                    print("  got post", got.ln_posterior[l], "map", got.map_vaf[l], "bias", got.map_bias[l], "best", got.best_event[l])
                    print("  ref post", ref.ln_posterior[l], "map", ref.map_vaf[l], "bias", ref.map_bias[l], "best", ref.best_event[l])
                    print("  ref events", ref.event_ln_posterior[l])
    LAST_STATS["run"], LAST_STATS["mismatching"] = done, bad
    print("scenarios run %d, mismatching %d; %s" % (done, bad, LAST_STATS))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
