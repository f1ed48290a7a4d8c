"""FDR control ("next" row §8(f)#4) against the reference's own fixtures and expected call counts
(tests/lib.rs:261-402).  The binary fixtures (data, tests/golden/fdr/) are read with the htslib-free BCF reader."""
import os

import pytest

from varlociraptor_amd import fdr
from varlociraptor_amd.bcfio import BcfReader

RES = os.path.join(os.path.dirname(__file__), "golden", "fdr")

DEL_1_30 = ("DEL", (1, 30))

CASES = [
    # (fixture, events, alpha, local, smart, retain_artifacts, vartype, expected)  -- tests/lib.rs
    ("test_fdr_ev_2", ["SOMATIC"], 0.05, False, False, False, DEL_1_30, 985),
    ("test_fdr_ev_2", ["ABSENT"], 0.001, False, False, False, DEL_1_30, 0),
    ("test_fdr_ev_4", ["SOMATIC_TUMOR"], 0.05, False, False, False, DEL_1_30, 0),
    ("test_fdr_local1", ["SOMATIC"], 0.05, True, False, False, DEL_1_30, 0),
    ("test_fdr_local2", ["SOMATIC"], 0.25, True, False, False, DEL_1_30, 1),
    ("test_fdr_local2_smart", ["SOMATIC"], 0.08, True, True, False, DEL_1_30, 1),
    ("test_fdr_local2_smart", ["SOMATIC"], 0.08, True, True, True, DEL_1_30, 1),
    ("test_fdr_local3", ["GERMLINE", "SOMATIC_TUMOR_LOW"], 0.05, True, False, False, None, 0),
]


@pytest.mark.parametrize("fixture,events,alpha,local,smart,retain,vartype,expected", CASES)
def test_reference_fdr_counts(fixture, events, alpha, local, smart, retain, vartype, expected):
    r = BcfReader(os.path.join(RES, fixture.replace("test_fdr_", "") + ".bcf"))
    recs = list(r)
    tags = [l.split("ID=")[1].split(",")[0] for l in r.header_lines if "ID=PROB_" in l]
    kept = fdr.control_fdr(recs, events, alpha, vartype=vartype, local=local, smart=smart, smart_retain_artifacts=retain, header_tags=tags)
    if expected > 50:
        assert abs(len(kept) - expected) <= 1  # assert_call_number allows +-1 (tests/lib.rs:241-247)
    else:
        assert len(kept) == expected


def test_unknown_events_rejected():
    r = BcfReader(os.path.join(RES, "local1.bcf"))
    recs = list(r)
    with pytest.raises(ValueError):
        fdr.control_fdr(recs, ["NOSUCH"], 0.05, header_tags=["PROB_SOMATIC"])


@pytest.mark.gpu
def test_threshold_on_device_matches_cpu():
    import math
    r = BcfReader(os.path.join(RES, "ev_2.bcf"))
    desc = fdr.collect_prob_dist(list(r), ["PROB_SOMATIC"], DEL_1_30)[::-1]
    assert fdr.fdr_threshold(desc, math.log(0.05), device="cuda") == pytest.approx(fdr.fdr_threshold(desc, math.log(0.05)), abs=1e-12)
