"""What-if planning over ONE cluster snapshot: many broker-set variants, one batch.

The front end SURVEY.md 8(f) N3 describes: the reference's broker-set flags
(--integer_broker_ids / --broker_hosts_to_remove / --disable_rack_awareness,
KafkaAssignmentGenerator.java:137-151, 238-250) decide the solver's `nodes` and rack map; here a
list of such variants is solved against the same current assignment in one launch.  Every topic's
current table is uploaded once and shared by all variants (cur_off of every scenario points at
the same rows), so S variants cost S out tables but one cur table.

    plan = WhatIf(brokers={id: rack or None}, topics={"t": {partition: [replicas]}})
    results = plan.solve([Variant(remove=[5]), Variant(add={9: "c"}), Variant(rack_aware=False)])
    results[0].moved_replicas, results[0].assignment("t")
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import abi
from .assigner import IllegalStateException, raise_for_status
from .flatten import FlatBatch, java_string_hashcode


@dataclass
class Variant:
    """One what-if: brokers to take out, brokers to add (id -> rack or None), rack awareness."""
    remove: Iterable[int] = ()
    add: Dict[int, Optional[str]] = field(default_factory=dict)
    rack_aware: bool = True                      # False = --disable_rack_awareness (KAG:241)
    label: str = ""


@dataclass
class VariantResult:
    label: str
    status: int                                  # KAS_OK or the first failing topic's status
    fail_topic: Optional[str]
    fail_partition: int
    moved_replicas: int
    moved_partitions: int
    digest: int
    _plan: "WhatIf" = field(repr=False, default=None)
    _index: int = field(repr=False, default=0)
    _out: np.ndarray = field(repr=False, default=None)

    def raise_for_status(self):
        """The exception the reference's CLI run would have died with (KAS:183-184 ...)."""
        rf = None
        if self.fail_topic is not None and self._plan is not None:
            rf = self._plan.rfs[self._plan.topic_names.index(self.fail_topic)]
        raise_for_status(self.fail_topic, self.status, self.fail_partition, rf)

    def assignment(self, topic: str) -> Dict[int, List[int]]:
        """partition -> new replica list of `topic` under this variant."""
        return self._plan._rows(self._index, topic, self._out)


class WhatIf:
    def __init__(self, brokers: Dict[int, Optional[str]], topics: Dict[str, Dict[int, Sequence[int]]],
                 desired_replication_factor: int = -1):
        self.brokers = dict(brokers)
        self.topic_names = list(topics)                                   # KAG:155-157: input order
        self.part_ids, self.widths, self.rfs, tables = [], [], [], []
        for name in self.topic_names:
            cur = topics[name]
            pids = sorted(cur)
            lens = {len(cur[p]) for p in pids}
            rf = desired_replication_factor
            if rf < 0:                                                     # KTA:49-62
                if len(lens) > 1:
                    raise IllegalStateException("Topic " + name + " has partitions with different replication factors")
                rf = lens.pop() if lens else -1
            w = max([len(cur[p]) for p in pids] + [0])
            t = np.full((len(pids), max(w, 1)), -1, dtype=np.int32)
            for i, p in enumerate(pids):
                t[i, :len(cur[p])] = cur[p]
            if any(len(cur[p]) != w for p in pids):
                raise ValueError("what-if tables need uniform replica lists per topic (use flatten() otherwise)")
            self.part_ids.append(np.asarray(pids, dtype=np.int32))
            self.widths.append(w)
            self.rfs.append(rf)
            tables.append(t.reshape(-1))
        self.cur_off = np.concatenate([[0], np.cumsum([t.size for t in tables])[:-1]]).astype(np.int64)
        self.cur = np.concatenate(tables) if tables else np.zeros(1, np.int32)

    # ---- batch construction -----------------------------------------------------------------
    def flat_batch(self, variants: Sequence[Variant]) -> FlatBatch:
        S, T = len(variants), len(self.topic_names)
        scen = np.zeros(S, dtype=abi.SCENARIO_DESC_DTYPE)
        topics = np.zeros(S * T, dtype=abi.TOPIC_DESC_DTYPE)
        node_ids, node_racks, aux = [], [], []
        node_off, out_off, aux_off = 0, 0, 0
        part_off = []
        for pids in self.part_ids:                                         # shared part_id arrays
            part_off.append(aux_off); aux.append(pids); aux_off += pids.size
        for s, v in enumerate(variants):
            live = {b: r for b, r in self.brokers.items() if b not in set(v.remove)}
            live.update(v.add)
            ids = sorted(live)
            index: Dict[str, int] = {}
            racks = []
            for b in ids:                                                  # KAS:82-86
                r = live[b] if (v.rack_aware and live[b] is not None) else str(b)
                racks.append(index.setdefault(r, len(index)))
            scen[s] = (len(ids), s * T, T, 0, node_off, -1)
            node_ids.append(np.asarray(ids, dtype=np.int32)); node_racks.append(np.asarray(racks, dtype=np.int32))
            node_off += len(ids)
            for t, name in enumerate(self.topic_names):
                P, w, rf = self.part_ids[t].size, self.widths[t], self.rfs[t]
                ow = max(w, rf, 1)
                topics[s * T + t] = (java_string_hashcode(name), P, w, rf, ow, 0, int(self.cur_off[t]), out_off,
                                     -1, -1, part_off[t])
                out_off += P * ow
        return FlatBatch(scen=scen, topics=topics,
                         node_id=np.concatenate(node_ids) if S else np.zeros(0, np.int32),
                         node_rack=np.concatenate(node_racks) if S else np.zeros(0, np.int32),
                         cur=self.cur, aux=np.concatenate(aux) if aux else np.zeros(0, np.int32),
                         ctx=np.zeros(0, np.int32), out_len=out_off)

    # ---- solve ---------------------------------------------------------------------------------
    def solve(self, variants: Sequence[Variant], solve_fn=None) -> List[VariantResult]:
        """Solve every variant in one batch (default: the HIP path through the C ABI)."""
        if solve_fn is None:
            from . import native
            solve_fn = native.solve_host
        fb = self.flat_batch(variants)
        ho = solve_fn(fb)
        self._fb = fb
        T = len(self.topic_names)
        res = []
        for s, v in enumerate(variants):
            sr = ho.scenario_results[s]
            ft = int(sr["fail_topic"])
            res.append(VariantResult(label=v.label, status=int(sr["status"]),
                                     fail_topic=self.topic_names[ft] if ft >= 0 else None,
                                     fail_partition=int(sr["fail_partition"]),
                                     moved_replicas=int(sr["moved_replicas"]),
                                     moved_partitions=int(sr["moved_partitions"]), digest=int(sr["digest"]),
                                     _plan=self, _index=s, _out=ho.out))
        return res

    def _rows(self, s: int, topic: str, out: np.ndarray) -> Dict[int, List[int]]:
        t = self.topic_names.index(topic)
        td = self._fb.topics[s * len(self.topic_names) + t]
        P, ow, off = int(td["n_partitions"]), int(td["out_width"]), int(td["out_off"])
        rows = out[off:off + P * ow].reshape(P, ow)
        return {int(p): [int(b) for b in rows[i] if b >= 0] for i, p in enumerate(self.part_ids[t])
                if (rows[i] >= 0).any()}
