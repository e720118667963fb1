"""KTA:47-69 through the C ABI ("missing" 6): kas_resolve_replication_factor resolves the replication factor the way
KafkaTopicAssigner.generateAssignment does - the first entry of the caller's map fixes it, the first later entry of another size
fails the topic with ITS partition id and list size - and kas_failure_text renders the reference's exception messages character
for character.  Held against the Python mirror (assigner.resolve_replication_factor / raise_for_status, which the reference's own
JUnit cases run through in tests/test_reference_junit.py) on seeded random maps and against the literal texts of KTA / KAS.
Host arithmetic: needs the library, not a GPU."""
import numpy as np
import pytest

from kafka_assigner_amd import abi, assigner, native


def _mirror(topic, cur, n_brokers, desired):
    try:
        return assigner.resolve_replication_factor(topic, cur, n_brokers, desired), None
    except assigner.IllegalStateException as e:
        return None, str(e)


def _abi(topic, cur, n_brokers, desired):
    res = native.resolve_replication_factor(list(cur.keys()), [len(v) for v in cur.values()], desired, n_brokers)
    if res.status == abi.KAS_OK:
        return res.rf, None
    return None, native.failure_text(topic, res.status, res.fail_partition, res.rf, res.fail_list_size)


def test_literal_texts_of_the_reference():
    assert native.failure_text("test", abi.KAS_FAIL_UNASSIGNABLE, 17) == "Partition 17 could not be fully assigned!"                      # KAS:183-184
    assert native.failure_text("t-1", abi.KAS_FAIL_RF_MISMATCH, 3, 2, 4) == "Topic t-1 has partition 3 with unexpected replication factor 4"  # KTA:58-60
    assert native.failure_text("t-1", abi.KAS_FAIL_RF_NOT_POSITIVE) == "Topic t-1 does not have a positive replication factor!"             # KTA:65-66
    assert native.failure_text("t-1", abi.KAS_FAIL_RF_GT_BROKERS, -1, 5) == "Topic t-1 has a higher replication factor (5) than available brokers!"  # KTA:67-69
    assert native.failure_text("größe-topic", abi.KAS_FAIL_RF_NOT_POSITIVE).startswith("Topic größe-topic does not")
    assert native.failure_text(None, abi.KAS_FAIL_RF_NOT_POSITIVE).startswith("Topic null does not")     # (Java prints a null String so)
    assert native.failure_text("t", abi.KAS_OK) == "" and native.failure_text("t", abi.KAS_FAIL_HASH_INDEX) == ""
    assert native.failure_text("x" * 2000, abi.KAS_FAIL_RF_NOT_POSITIVE) == ("Topic " + "x" * 2000)[:1023]   # truncated, terminated


def test_the_cases_generate_assignment_distinguishes():
    cur = {0: [10, 11], 1: [11, 12], 2: [12, 10], 3: [10, 12]}
    assert _abi("test", cur, 5, -1) == (2, None) == _mirror("test", cur, 5, -1)
    assert _abi("test", cur, 5, 3) == (3, None) == _mirror("test", cur, 5, 3)           # a desired factor wins, sizes are not compared
    ragged = {0: [10, 11], 1: [11], 2: [12, 10, 13]}
    assert _abi("test", ragged, 5, 2) == (2, None) == _mirror("test", ragged, 5, 2)
    a = _abi("test", ragged, 5, -1)
    assert a == (None, "Topic test has partition 1 with unexpected replication factor 1") == _mirror("test", ragged, 5, -1)
    # map order decides which entry fixes the factor and which one fails (KTA:50: entrySet() order)
    other = {2: [12, 10, 13], 0: [10, 11], 1: [11]}
    assert _abi("test", other, 5, -1) == (None, "Topic test has partition 0 with unexpected replication factor 2") == _mirror("test", other, 5, -1)
    assert _abi("test", {}, 5, -1) == (None, "Topic test does not have a positive replication factor!") == _mirror("test", {}, 5, -1)
    assert _abi("test", {0: [], 1: []}, 5, -1) == (None, "Topic test does not have a positive replication factor!") == _mirror("test", {0: [], 1: []}, 5, -1)
    assert _abi("test", cur, 5, 0) == (None, "Topic test does not have a positive replication factor!") == _mirror("test", cur, 5, 0)
    assert _abi("test", cur, 1, -1) == (None, "Topic test has a higher replication factor (2) than available brokers!") == _mirror("test", cur, 1, -1)
    assert _abi("test", cur, 2, 3) == (None, "Topic test has a higher replication factor (3) than available brokers!") == _mirror("test", cur, 2, 3)
    # an empty first list with longer ones behind it: the factor becomes 0 at the first entry and the second entry mismatches
    z = {5: [], 6: [1, 2]}
    assert _abi("test", z, 5, -1) == (None, "Topic test has partition 6 with unexpected replication factor 2") == _mirror("test", z, 5, -1)


def test_random_maps_against_the_python_mirror():
    rng = np.random.default_rng(20260930)
    seen = set()
    for case in range(3000):
        n = int(rng.integers(0, 12))
        base = int(rng.integers(0, 5))
        keys = rng.permutation(50)[:n].tolist()
        cur = {int(k): list(range(base if rng.random() < 0.85 else int(rng.integers(0, 6)))) for k in keys}
        desired = int(rng.choice([-1, -1, -1, -7, 0, 1, 2, 3, 6]))
        n_brokers = int(rng.integers(0, 7))
        a, m = _abi("topic-%d" % case, cur, n_brokers, desired), _mirror("topic-%d" % case, cur, n_brokers, desired)
        assert a == m, (cur, desired, n_brokers, a, m)
        seen.add("ok" if a[1] is None else a[1].split(" ", 3)[2])
    assert {"ok", "has", "does"} <= seen, seen


def test_refusals():
    res = abi.RfResult()
    L = native.load()
    assert L.kas_resolve_replication_factor(None, None, 3, -1, 5, res) == abi.KAS_E_INVALID_ARG
    assert L.kas_resolve_replication_factor(None, None, 0, 2, 5, res) == 0 and res.status == abi.KAS_OK and res.rf == 2
    assert L.kas_resolve_replication_factor(None, None, 0, 2, 5, None) == abi.KAS_E_INVALID_ARG
