"""literal_ref.py — line-by-line Python restatement of the reference solver.  TEST INFRASTRUCTURE.

Second, independent restatement of
  KafkaAssignmentStrategy.java (KAS) lines 40-370 and KafkaTopicAssigner.java (KTA) lines 42-72,
written against Python dicts / sorted containers so that every statement can be laid next to the
Java it mirrors (TreeMap -> iterate sorted keys, TreeSet -> sorted list, HashMap -> dict).  It is
slow (pure Python loops) and only meant for small cases: it cross-checks oracle/kas_oracle.c
(under hypothesis-generated inputs) and re-derives the golden vectors of tests/golden/ (which were
transcribed by hand from SURVEY.md Appendix B) in tests/golden/make_golden.py.  Nothing in the
product imports it.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Set

INT_MIN = -(2 ** 31)


class IllegalStateException(Exception):
    """Guava Preconditions.checkState failure (KAS:183-184, KTA:58-69)."""


class ArrayIndexOutOfBoundsException(Exception):
    """Negative index at KAS:192 when topic.hashCode() == Integer.MIN_VALUE."""


def java_string_hashcode(s: str) -> int:
    """java.lang.String.hashCode(): s[0]*31^(n-1) + ... over UTF-16 code units, int32 wrap."""
    h = 0
    data = s.encode("utf-16-be")
    for i in range(0, len(data), 2):
        unit = (data[i] << 8) | data[i + 1]
        h = (31 * h + unit) & 0xFFFFFFFF
    return h - (1 << 32) if h >= (1 << 31) else h


def java_abs(x: int) -> int:
    """Math.abs(int): identity on Integer.MIN_VALUE."""
    return x if x == INT_MIN else abs(x)


def java_rem(a: int, n: int) -> int:
    """Java % on ints: truncating division, result takes the sign of the dividend."""
    return int(math.fmod(a, n))


def java_int_mul(a: int, b: int) -> int:
    v = (a * b) & 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


class Rack:  # KAS:337-355
    def __init__(self, rack_id: str):
        self.id = rack_id
        self.assigned_partitions: Set[int] = set()

    def can_accept(self, partition: int) -> bool:  # KAS:346-348
        return partition not in self.assigned_partitions

    def accept(self, partition: int) -> None:  # KAS:350-354
        assert self.can_accept(partition)
        self.assigned_partitions.add(partition)


class Node:  # KAS:307-332
    def __init__(self, node_id: int, capacity: int, rack: Rack):
        self.id = node_id
        self.capacity = capacity
        self.rack = rack
        self.assigned_partitions: Set[int] = set()

    def can_accept(self, partition: int) -> bool:  # KAS:320-324
        return (partition not in self.assigned_partitions
                and len(self.assigned_partitions) < self.capacity
                and self.rack.can_accept(partition))

    def accept(self, partition: int) -> None:  # KAS:326-331
        assert self.can_accept(partition)
        self.assigned_partitions.add(partition)
        self.rack.accept(partition)


class Context:  # KAS:360-369
    def __init__(self):
        self.counter: Dict[int, Dict[int, int]] = {}


def get_max_replicas_per_node(nodes, partitions, replication_factor: int) -> int:  # KAS:65-71
    total_replicas = float(java_int_mul(len(partitions), replication_factor))
    return int(math.ceil(total_replicas / len(nodes)))


def create_node_map(node_rack_assignment: Dict[int, str], nodes: Iterable[int],
                    max_replicas: int) -> Dict[int, Node]:  # KAS:73-99
    rack_map: Dict[str, Rack] = {}
    node_map: Dict[int, Node] = {}
    for node_id in nodes:
        assert node_id not in node_map  # KAS:80
        rack_id = node_rack_assignment.get(node_id)
        if rack_id is None:
            rack_id = str(node_id)  # KAS:82-86
        rack = rack_map.get(rack_id)
        if rack is None:
            rack = Rack(rack_id)
            rack_map[rack_id] = rack
        node_map[node_id] = Node(node_id, max_replicas, rack)
    return node_map  # iterate with sorted(node_map) wherever Java iterates the TreeMap


def fill_nodes_from_assignment(assignment: Dict[int, List[int]],
                               node_map: Dict[int, Node]) -> None:  # KAS:101-131
    iterators = {p: iter(list(v)) for p, v in assignment.items()}  # TreeMap: use sorted keys
    filled = False
    while not filled:
        for partition in sorted(list(iterators.keys())):  # KAS:113-114
            node_it = iterators[partition]
            node_id = next(node_it, None)
            if node_id is not None:  # nodeIt.hasNext()
                node = node_map.get(node_id)
                if node is not None and node.can_accept(partition):  # KAS:120
                    node.accept(partition)
            else:
                del iterators[partition]  # roundRobin.remove()
        filled = len(iterators) == 0


def get_orphaned_replicas(node_map: Dict[int, Node], partitions: Iterable[int],
                          replication_factor: int) -> Dict[int, int]:  # KAS:133-160
    partition_counter: Dict[int, int] = {}
    for node_id in sorted(node_map):
        for partition in sorted(node_map[node_id].assigned_partitions):
            partition_counter[partition] = partition_counter.get(partition, 0) + 1
    orphaned: Dict[int, int] = {}
    for partition in sorted(partitions):  # Set<Integer> partitions is a TreeSet via KTA:50
        remaining = replication_factor
        if partition in partition_counter:
            remaining -= partition_counter[partition]
        if remaining > 0:
            orphaned[partition] = remaining
    return orphaned


def get_node_processing_order(topic_hash: int, node_ids: List[int]) -> List[int]:  # KAS:188-200
    n = len(node_ids)
    order: List[Optional[int]] = [None] * n
    index = java_rem(java_abs(topic_hash), n)
    for node_id in node_ids:
        if index < 0 or index >= n:
            raise ArrayIndexOutOfBoundsException(str(index))
        order[index] = node_id
        index += 1
        if index == n:
            index = 0
    return order  # type: ignore[return-value]


def assign_orphans(topic_hash: int, node_map: Dict[int, Node],
                   orphaned: Dict[int, int]) -> None:  # KAS:162-186
    order = get_node_processing_order(topic_hash, sorted(node_map))  # KAS:168
    for partition in sorted(orphaned):  # TreeMap entrySet
        remaining = orphaned[partition]
        it = iter(order)  # KAS:175: restarts at the head for every orphan
        for node_id in it:
            if not remaining > 0:
                break
            node = node_map[node_id]
            if node.can_accept(partition):
                node.accept(partition)
                remaining -= 1
        if remaining != 0:  # KAS:183-184
            raise IllegalStateException(
                "Partition " + str(partition) + " could not be fully assigned!")


def _ensure_count(counters: Dict[int, Dict[int, int]], node_id: int, replica_id: int) -> int:
    replica_count = counters.get(node_id)  # KAS:289-301
    if replica_count is None:
        replica_count = {}
        counters[node_id] = replica_count
    current = replica_count.get(replica_id)
    if current is None:
        current = 0
        replica_count[replica_id] = current
    return current


def get_least_seen_node_for_replica_id(topic_hash: int, counters, replica_id: int,
                                       nodes: List[int]) -> int:  # KAS:263-278
    min_count = None
    min_node = None
    for node_id in get_node_processing_order(topic_hash, sorted(nodes)):
        count = _ensure_count(counters, node_id, replica_id)
        if min_count is None or count < min_count:
            min_count = count
            min_node = node_id
    assert min_node is not None
    return min_node


def compute_preference_lists(topic_hash: int, node_map: Dict[int, Node],
                             context: Context) -> Dict[int, List[int]]:  # KAS:202-239
    unordered: Dict[int, List[int]] = {}
    for node_id in sorted(node_map):
        for partition in sorted(node_map[node_id].assigned_partitions):
            unordered.setdefault(partition, []).append(node_id)
    counters = context.counter
    preferences: Dict[int, List[int]] = {}
    for partition_id in sorted(unordered):
        preference_list = unordered[partition_id]
        ordered: List[int] = []
        replication_factor = len(preference_list)  # KAS:227: the list's OWN size
        node_set = sorted(set(preference_list))
        for replica in range(replication_factor):
            node_to_select = get_least_seen_node_for_replica_id(
                topic_hash, counters, replica, node_set)
            node_set.remove(node_to_select)
            ordered.append(node_to_select)
        preferences[partition_id] = ordered
        replica = 0  # updateCountersFromList KAS:254-261
        for node_id in ordered:
            current = _ensure_count(counters, node_id, replica)
            counters[node_id][replica] = current + 1
            replica += 1
    return preferences


def get_rack_aware_assignment(topic_name, current_assignment: Dict[int, List[int]],
                              node_rack_assignment: Dict[int, str], nodes: Set[int],
                              partitions: Set[int], replication_factor: int,
                              context: Optional[Context]) -> Dict[int, List[int]]:  # KAS:40-63
    topic_hash = topic_name if isinstance(topic_name, int) else java_string_hashcode(topic_name)
    max_replicas = get_max_replicas_per_node(nodes, partitions, replication_factor)
    node_map = create_node_map(node_rack_assignment, nodes, max_replicas)
    fill_nodes_from_assignment(current_assignment, node_map)
    orphaned = get_orphaned_replicas(node_map, partitions, replication_factor)
    assign_orphans(topic_hash, node_map, orphaned)
    if context is None:
        context = Context()
    return compute_preference_lists(topic_hash, node_map, context)


class KafkaTopicAssigner:  # KTA:18-72
    def __init__(self):
        self.assignment_context = Context()  # KTA:21-23

    def generate_assignment(self, topic, current_assignment: Dict[int, List[int]],
                            brokers: Set[int], rack_assignment: Dict[int, str],
                            desired_replication_factor: int) -> Dict[int, List[int]]:
        replication_factor = desired_replication_factor  # KTA:49
        partitions: Set[int] = set()
        for partition, replicas in current_assignment.items():  # KTA:51-62
            partitions.add(partition)
            if replication_factor < 0:
                replication_factor = len(replicas)
            elif desired_replication_factor < 0:
                if replication_factor != len(replicas):
                    raise IllegalStateException(
                        "Topic " + str(topic) + " has partition " + str(partition) +
                        " with unexpected replication factor " + str(len(replicas)))
        if not replication_factor > 0:  # KTA:65-66
            raise IllegalStateException(
                "Topic " + str(topic) + " does not have a positive replication factor!")
        if not replication_factor <= len(brokers):  # KTA:67-69
            raise IllegalStateException(
                "Topic " + str(topic) + " has a higher replication factor (" +
                str(replication_factor) + ") than available brokers!")
        return get_rack_aware_assignment(topic, current_assignment, rack_assignment, brokers,
                                         partitions, replication_factor,
                                         self.assignment_context)
