"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on the same inputs."""
import os

import numpy as np
import pytest

from varlociraptor_amd import abi, engine, obsfmt, synth
from varlociraptor_amd.scenario import Sample, Scenario

from parity import TOL, compare, describe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return engine.lib()


def run_both(oracle, scenario, batch):
    plan = engine.Plan(scenario)
    got = plan.call_host(batch)
    ref = oracle.call(scenario, batch, want_events=True)
    plan.close()
    return got, ref


def test_fixture_matches_reference_calls(oracle, golden_dir, lib):
    d = os.path.join(golden_dir, "flamegraph_profiling")
    batch, _ = obsfmt.read_observation_vcf([os.path.join(d, "normal.vcf")], omit_bias_mask=abi.BIAS_ALL)
    sc = Scenario({"normal": Sample(resolution=0.1, universe="[0.0,1.0]")}, {"present": "normal:]0.0,1.0]"})
    got, ref = run_both(oracle, sc, batch)
    m = compare(got, ref, label="fixture")
    print(describe(m))
    assert m["frac_within"] == 1.0
    # against the reference's own output file (PHRED f32, 6 significant digits)
    expected_absent = [285.541, 260.289, 273.284, 339.982, 262.097, 383.273, 528.672, 452.981, 526.504, 521.16, 517.726]
    assert np.allclose(got.phred()[:, 0], expected_absent, rtol=2e-6)
    assert np.allclose(got.map_vaf[:, 0], 1.0)


@pytest.mark.parametrize("cfg_name,n", [("config2", 400), ("config3", 150)])
def test_synthetic_parity(oracle, lib, cfg_name, n):
    cfg = synth.CONFIGS[cfg_name]()
    batch = synth.generate(cfg, n)
    got, ref = run_both(oracle, cfg.scenario, batch)
    m = compare(got, ref, label=cfg_name)
    print(describe(m))
    assert m["frac_within"] == 1.0, describe(m)
    assert m["bias_equal"]
    assert m["status_equal"]
