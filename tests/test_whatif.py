"""What-if front end (kafka_assigner_amd/whatif.py): one snapshot, many broker-set variants, one
batch over a shared current-assignment table.  CPU: the flattening against the oracle and the
reference-shaped mirror; GPU: the same batch through the HIP path."""
import json
import os

import pytest

from kafka_assigner_amd import abi
from kafka_assigner_amd.whatif import Variant, WhatIf
from oracle_lib import oracle_solve

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_appendix_b.json")))
C1 = G["config1"]


def _plan():
    brokers = {b: "abc"[b % 3] for b in range(6)}
    topics = {name: {int(p): r for p, r in C1["current"][t].items()} for t, name in enumerate(C1["topics"])}
    return WhatIf(brokers, topics)


VARIANTS = [Variant(label="no-op"),
            Variant(remove=[5], add={6: "c"}, label="replace 5->6 (rack c)"),
            Variant(add={6: "c", 7: "a", 8: "b"}, label="add 6(c),7(a),8(b)"),
            Variant(remove=[5], label="remove 5, rack-aware"),
            Variant(remove=[5], rack_aware=False, label="remove 5, rack-awareness disabled")]


def _check(results):
    by = {c["name"]: c for c in C1["cases"]}
    for r in results:
        case = by[r.label]
        if "fails" in case:
            assert r.status == abi.KAS_FAIL_UNASSIGNABLE and r.fail_topic == C1["topics"][0]
            assert r.fail_partition == case["fails"]["partition"]
            with pytest.raises(Exception) as e:
                r.raise_for_status()
            assert "Partition %d could not be fully assigned!" % r.fail_partition in str(e.value)
            continue
        assert r.status == abi.KAS_OK
        assert r.moved_replicas == sum(case["moved_replicas"])
        for t, name in enumerate(C1["topics"]):
            assert r.assignment(name) == {int(p): v for p, v in case["expected"][t].items()}, (r.label, name)


def test_all_appendix_b_variants_in_one_batch_share_one_cur_table():
    plan = _plan()
    fb = plan.flat_batch(VARIANTS)
    assert fb.n_scenarios == 5 and fb.cur.size == 3 * 12 * 3            # one copy of the three topics
    assert set(fb.topics["cur_off"][:3]) == set(fb.topics["cur_off"][3:6])
    _check(plan.solve(VARIANTS, solve_fn=oracle_solve))


@pytest.mark.gpu
def test_what_if_batch_on_the_gpu():
    from kafka_assigner_amd import native
    plan = _plan()
    _check(plan.solve(VARIANTS))
    fb = plan.flat_batch(VARIANTS)
    from parity_util import assert_same_outputs
    assert_same_outputs(fb, oracle_solve(fb), native.solve_host(fb), "what-if batch")
