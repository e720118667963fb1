"""Host-side mirror of the reference's operator interface for the hot path.

  KafkaTopicAssigner.generateAssignment            KafkaTopicAssigner.java:42-72
  KafkaAssignmentStrategy.getRackAwareAssignment   KafkaAssignmentStrategy.java:40-63

Same names (snake_case), same argument meaning, same error behaviour: the reference's
Preconditions failures surface as IllegalStateException with the reference's message text.
The solve itself goes through the C ABI (native.py -> csrc/libkas_hip.so -> HIP kernels);
there is no CPU path here.  (The C++ twin of this file is host/kafka_assigner.hpp.)
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Set

from . import abi
from .flatten import Scenario, Topic, flatten, unflatten_context, unflatten_topic

INT_MIN = -(2 ** 31)


class IllegalStateException(RuntimeError):
    """Guava Preconditions.checkState failure (KAS:183-184; KTA:58-69)."""


class ArrayIndexOutOfBoundsException(RuntimeError):
    """KAS:190-192 with topic.hashCode() == Integer.MIN_VALUE."""


def resolve_replication_factor(topic, current_assignment: Dict[int, Sequence[int]],
                               n_brokers: int, desired_replication_factor: int) -> int:
    """KTA:49-69: derive the replication factor and apply the two precondition checks."""
    replication_factor = desired_replication_factor
    for partition, replicas in current_assignment.items():
        if replication_factor < 0:
            replication_factor = len(replicas)
        elif desired_replication_factor < 0:
            if replication_factor != len(replicas):                       # KTA:58-60
                raise IllegalStateException(
                    "Topic " + str(topic) + " has partition " + str(partition) +
                    " with unexpected replication factor " + str(len(replicas)))
    if not replication_factor > 0:                                         # KTA:65-66
        raise IllegalStateException(
            "Topic " + str(topic) + " does not have a positive replication factor!")
    if not replication_factor <= n_brokers:                                # KTA:67-69
        raise IllegalStateException(
            "Topic " + str(topic) + " has a higher replication factor (" +
            str(replication_factor) + ") than available brokers!")
    return replication_factor


def raise_for_status(topic, status: int, fail_partition: int, replication_factor=None) -> None:
    """Turn a kas_topic_result status back into the exception the reference throws."""
    if status == abi.KAS_OK:
        return
    if status == abi.KAS_FAIL_UNASSIGNABLE:                                # KAS:183-184
        raise IllegalStateException(
            "Partition " + str(fail_partition) + " could not be fully assigned!")
    if status == abi.KAS_FAIL_RF_NOT_POSITIVE:
        raise IllegalStateException(
            "Topic " + str(topic) + " does not have a positive replication factor!")
    if status == abi.KAS_FAIL_RF_GT_BROKERS:                               # KTA:67-69
        raise IllegalStateException(
            "Topic " + str(topic) + " has a higher replication factor (" +
            str(replication_factor if replication_factor is not None else "?") +
            ") than available brokers!")
    if status == abi.KAS_FAIL_HASH_INDEX:                                  # KAS:190-192
        raise ArrayIndexOutOfBoundsException("negative node processing index")
    raise RuntimeError("solver status " + abi.STATUS_NAMES.get(status, str(status)))


class Context:
    """KafkaAssignmentStrategy.Context (KAS:360-369): leader/follower counters per broker."""

    def __init__(self):
        self.counter: Dict[int, Dict[int, int]] = {}


def _solve_one(solve_host, topic, current_assignment, node_rack_assignment, nodes, partitions,
               replication_factor, context: Optional[Context]):
    sc = Scenario(brokers=nodes, racks=dict(node_rack_assignment),
                  topics=[Topic(topic, {int(p): list(v) for p, v in current_assignment.items()},
                                replication_factor,
                                None if partitions is None else set(partitions))],
                  context=(context.counter if context is not None else None),
                  want_context=context is not None)
    fb = flatten([sc])
    ho = solve_host(fb)
    tr = ho.topic_results[0]
    raise_for_status(topic, int(tr["status"]), int(tr["fail_partition"]), replication_factor)
    if context is not None:
        # counters of brokers outside `nodes` are untouched by this call (they are not in the
        # flat table), exactly as the reference only touches nodes it iterates
        new = unflatten_context(fb, ho.ctx, 0)
        for b in set(int(x) for x in nodes):
            if b in new:
                context.counter[b] = new[b]
            elif b in context.counter:
                context.counter[b] = {}
    return unflatten_topic(fb, ho.out, 0), tr


class KafkaAssignmentStrategy:
    """Static entry point mirroring KAS:40-63, solved on the GPU through the C ABI.

    The C ABI is a generateAssignment-level boundary (include/kas_abi.h): the solver applies
    KafkaTopicAssigner's two replication-factor preconditions (KTA:65-69) itself.  The reference's
    getRackAwareAssignment has no such checks — called directly with rf > |nodes| it throws
    "Partition p could not be fully assigned!" from KAS:183-184 and with rf <= 0 it returns the
    sticky-filled map — so a DIRECT caller of this mirror gets KafkaTopicAssigner's message for
    those two argument ranges instead.  Every in-range call (what KTA:70-71 can pass) is
    identical."""

    @staticmethod
    def get_rack_aware_assignment(topic_name, current_assignment: Dict[int, Sequence[int]],
                                  node_rack_assignment: Dict[int, str], nodes: Set[int],
                                  partitions: Set[int], replication_factor: int,
                                  context: Optional[Context]) -> Dict[int, List[int]]:
        import os
        from . import native
        from .flatten import cells16_to_ids

        def solve16(fb):
            """kas_solve_host16 (ABI v5): replicas travel as positions in the sorted broker list, half the bytes of the
            per-topic call; mapped back here.  KAS_CELLS32=1: kas_solve_host with int32 broker ids."""
            ho = native.solve_host16(fb)
            ho.out = cells16_to_ids(fb, ho.out)
            return ho

        cells32 = os.environ.get("KAS_CELLS32", "") == "1" or len(set(nodes)) > 32767
        result, _ = _solve_one(native.solve_host if cells32 else solve16, topic_name, current_assignment,
                               node_rack_assignment, nodes, partitions, replication_factor,
                               context)
        return result


class KafkaTopicAssigner:
    """Mirror of KafkaTopicAssigner (KTA:18-72): one Context per instance (KTA:19-23)."""

    def __init__(self):
        self.assignment_context = Context()

    def generate_assignment(self, topic, current_assignment: Dict[int, Sequence[int]],
                            brokers: Set[int], rack_assignment: Dict[int, str],
                            desired_replication_factor: int) -> Dict[int, List[int]]:
        replication_factor = resolve_replication_factor(
            topic, current_assignment, len(set(brokers)), desired_replication_factor)
        partitions = set(int(p) for p in current_assignment.keys())          # KTA:50-54
        return KafkaAssignmentStrategy.get_rack_aware_assignment(
            topic, current_assignment, rack_assignment, set(brokers), partitions,
            replication_factor, self.assignment_context)
