"""The reference's own unit tests, assertion for assertion.

KafkaTopicAssignerTest.java (KTAT) lines 18-187, re-expressed against the reference-shaped
interface of every implementation: the two CPU restatements (always) and the HIP product path
(on the GPU box).  These are the only result pins the reference itself ships.
"""
import pytest

from impls import ALL_IMPLS, IMPLS


def verify_partitions_and_build_replica_counts(current, new, minimal_movement_threshold):
    """KTAT:159-187."""
    counts = {}
    for partition, replicas in new.items():
        assert len(replicas) == len(set(replicas))          # no broker twice (KTAT:167-168)
        for b in replicas:
            counts[b] = counts.get(b, 0) + 1
        prev = set(current[partition])
        assert len(set(replicas) & prev) >= minimal_movement_threshold   # KTAT:179-184
    return counts


CUR_A = {0: [10, 11], 1: [11, 12], 2: [12, 10], 3: [10, 12]}


@pytest.mark.parametrize("impl", ALL_IMPLS)
def test_rack_aware_expansion(impl):
    """KTAT:18-57."""
    racks = {10: "a", 11: "b", 12: "c", 13: "a", 14: "b"}
    new = IMPLS[impl]().generate_assignment("test", CUR_A, {10, 11, 12, 13, 14}, racks, -1)
    counts = verify_partitions_and_build_replica_counts(CUR_A, new, 1)
    assert sum(1 for v in counts.values() if v == 1) == 2
    assert sum(1 for v in counts.values() if v == 2) == 3


@pytest.mark.parametrize("impl", ALL_IMPLS)
def test_cluster_expansion(impl):
    """KTAT:59-82."""
    new = IMPLS[impl]().generate_assignment("test", CUR_A, {10, 11, 12, 13}, {}, -1)
    counts = verify_partitions_and_build_replica_counts(CUR_A, new, 1)
    for v in counts.values():
        assert v == 2


@pytest.mark.parametrize("impl", ALL_IMPLS)
def test_decommission(impl):
    """KTAT:84-122."""
    cur = {0: [10, 11], 1: [11, 12], 2: [12, 13], 3: [13, 10]}
    new = IMPLS[impl]().generate_assignment("test", cur, {10, 11, 13}, {}, -1)
    counts = verify_partitions_and_build_replica_counts(cur, new, 1)
    assert 12 not in counts
    assert all(v in (2, 3) for v in counts.values())
    assert sum(1 for v in counts.values() if v == 2) == 1
    assert sum(1 for v in counts.values() if v == 3) == 2


@pytest.mark.parametrize("impl", ALL_IMPLS)
def test_replacement(impl):
    """KTAT:124-157 — includes the one exact pin: partition 0 unchanged INCLUDING order."""
    new = IMPLS[impl]().generate_assignment("test", CUR_A, {10, 11, 13}, {}, -1)
    counts = verify_partitions_and_build_replica_counts(CUR_A, new, 1)
    assert 12 not in counts
    assert new[0] == CUR_A[0]
    assert 11 in new[1] and (10 in new[1] or 13 in new[1])
    assert 10 in new[2] and (11 in new[2] or 13 in new[2])
    assert 10 in new[3] and (11 in new[3] or 13 in new[3])
