"""Known-answer vectors of SURVEY.md Appendix B (tests/golden/survey_appendix_b.json).

They were derived by the survey's own scratch restatement — independent of both restatements
in oracle/ — and include multi-topic Context carry-over, failure cases and the quirks Q5-Q8.
"""
import json
import os

import pytest

from impls import ALL_IMPLS, IMPLS
from kafka_assigner_amd import assigner as A
from kafka_assigner_amd.flatten import java_string_hashcode

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_appendix_b.json")))


def _int_keys(d):
    return {int(k): v for k, v in d.items()}


def test_java_hashcodes():
    for name, h in G["hashes"].items():
        assert java_string_hashcode(name) == h


@pytest.mark.parametrize("impl", ALL_IMPLS)
@pytest.mark.parametrize("case", G["ktat"], ids=[c["name"] for c in G["ktat"]])
def test_ktat_exact_outputs(impl, case):
    new = IMPLS[impl]().generate_assignment(
        case["topic"], _int_keys(case["current"]), set(case["brokers"]),
        _int_keys(case["racks"]), case["desired_rf"])
    assert new == _int_keys(case["expected"])


C1 = G["config1"]


@pytest.mark.parametrize("impl", ALL_IMPLS)
@pytest.mark.parametrize("case", C1["cases"], ids=[c["name"] for c in C1["cases"]])
def test_config1_three_topics_one_context(impl, case):
    """BASELINE.json config 1: 3 topics x 12 partitions, 6 brokers / 3 racks, RF 3, one
    assigner (= one Context, KTA:19-23) across the topics as in KAG:172-184."""
    assigner = IMPLS[impl]()
    brokers = set(case["brokers"])
    racks = _int_keys(case["racks"])
    if "fails" in case:
        # CLI semantics: the run aborts at the first failing topic
        with pytest.raises(A.IllegalStateException) as e:
            assigner.generate_assignment(C1["topics"][0], _int_keys(C1["current"][0]), brokers, racks, -1)
        assert str(e.value) == "Partition %d could not be fully assigned!" % case["fails"]["partition"]
        # and each topic fails standalone at the listed partition
        for t, p in enumerate(case["fails"]["standalone_fail_partitions"]):
            with pytest.raises(A.IllegalStateException) as e:
                IMPLS[impl]().generate_assignment(C1["topics"][t], _int_keys(C1["current"][t]),
                                                  brokers, racks, -1)
            assert str(e.value) == "Partition %d could not be fully assigned!" % p
        return
    for t, topic in enumerate(C1["topics"]):
        cur = _int_keys(C1["current"][t])
        new = assigner.generate_assignment(topic, cur, brokers, racks, -1)
        assert new == _int_keys(case["expected"][t]), f"topic {topic}"
        moved = sum(len(set(new[p]) - set(cur[p])) for p in cur)
        assert moved == case["moved_replicas"][t]
    if "final_context" in case:
        want = {int(n): {int(r): c for r, c in m.items()} for n, m in case["final_context"].items()}
        assert assigner.context == want


@pytest.mark.parametrize("impl", ALL_IMPLS)
@pytest.mark.parametrize("case", G["quirks"], ids=[c["name"] for c in G["quirks"]])
def test_quirks(impl, case):
    args = (case["topic"], _int_keys(case["current"]), set(case["brokers"]),
            _int_keys(case["racks"]), case["desired_rf"])
    if case.get("error") == "index":
        with pytest.raises(A.ArrayIndexOutOfBoundsException):
            IMPLS[impl]().generate_assignment(*args)
    else:
        assert IMPLS[impl]().generate_assignment(*args) == _int_keys(case["expected"])


def test_golden_file_is_what_the_literal_restatement_derives():
    """tests/golden/make_golden.py recomputes every expected entry of the (hand-transcribed) JSON
    with oracle/literal_ref.py; the committed file must be exactly that."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "golden", "make_golden.py")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
