"""Flattening between the reference's argument shapes and the flat int32 tables of the C ABI.

The reference hands KafkaAssignmentStrategy.getRackAwareAssignment boxed collections
(KafkaAssignmentStrategy.java:40-43).  This module turns those into the pools and descriptors
of include/kas_abi.h and back:

  Map<Integer,List<Integer>> currentAssignment -> cur[P][cur_width] rows in ascending partition id
  Set<Integer> nodes                            -> node_id[N] ascending
  Map<Integer,String> nodeRackAssignment        -> node_rack[N]: dense index per distinct rack
                                                   string; a broker without a rack uses its own
                                                   decimal id as the rack string (KAS:82-86), so a
                                                   real rack named "12" merges with rack-less
                                                   broker 12 exactly as in the reference
  String topicName                              -> Java String.hashCode()
  Context                                       -> counter[N][ctx_width]
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Set, Union

import numpy as np

from . import abi


def java_string_hashcode(s: str) -> int:
    """java.lang.String.hashCode() (UTF-16 code units, int32 wrap-around)."""
    h = 0
    data = s.encode("utf-16-be")
    for i in range(0, len(data), 2):
        h = (31 * h + ((data[i] << 8) | data[i + 1])) & 0xFFFFFFFF
    return h - (1 << 32) if h >= (1 << 31) else h


@dataclass
class Topic:
    """One generateAssignment call's worth of input (KafkaTopicAssigner.java:42-44)."""
    name: Union[str, int]                     # topic name, or its Java hashCode directly
    current: Dict[int, Sequence[int]]         # partition -> current replica list
    rf: int                                   # replication factor handed to KAS:40-43
    partitions: Optional[Set[int]] = None     # None = keys(current) (what KTA:50-54 passes)

    @property
    def name_hash(self) -> int:
        return self.name if isinstance(self.name, int) else java_string_hashcode(self.name)


@dataclass
class Scenario:
    """One cluster snapshot: broker set, rack map and the topics solved against one Context."""
    brokers: Iterable[int]
    racks: Dict[int, str]
    topics: List[Topic]
    context: Optional[Dict[int, Dict[int, int]]] = None   # Context.counter (KAS:361)
    want_context: bool = False                             # write the final counters back


@dataclass
class FlatBatch:
    scen: np.ndarray                 # SCENARIO_DESC_DTYPE [S]
    topics: np.ndarray               # TOPIC_DESC_DTYPE   [T]
    node_id: np.ndarray              # int32 node pool
    node_rack: np.ndarray            # int32 rack pool
    cur: np.ndarray                  # int32 cur pool
    aux: np.ndarray                  # int32 aux pool
    ctx: np.ndarray                  # int32 ctx pool (in/out)
    out_len: int
    row_ids: List[np.ndarray] = field(default_factory=list)   # per topic: partition id per row

    @property
    def n_scenarios(self) -> int:
        return int(self.scen.shape[0])

    @property
    def n_topics(self) -> int:
        return int(self.topics.shape[0])

    def algorithmic_bytes(self) -> int:
        """Bytes the path must move: 4*P*(cur_width+out_width) per topic + 8*N per scenario
        (+ 8*N*ctx_width when a Context goes in and out)."""
        t = self.topics
        b = int((4 * t["n_partitions"].astype(np.int64)
                 * (t["cur_width"].astype(np.int64) + t["out_width"].astype(np.int64))).sum())
        s = self.scen
        b += int((8 * s["n_nodes"].astype(np.int64)).sum())
        has_ctx = s["ctx_off"] >= 0
        b += int((8 * s["n_nodes"].astype(np.int64) * s["ctx_width"].astype(np.int64))[has_ctx].sum())
        return b


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(C.POINTER(typ))


def batch_desc(fb: FlatBatch) -> abi.BatchDesc:
    """ctypes kas_batch_desc over the FlatBatch's host arrays (which must stay alive)."""
    bd = abi.BatchDesc()
    bd.n_scenarios = fb.n_scenarios
    bd.n_topics = fb.n_topics
    bd.scenarios = C.cast(fb.scen.ctypes.data, C.POINTER(abi.ScenarioDesc))
    bd.topics = C.cast(fb.topics.ctypes.data, C.POINTER(abi.TopicDesc))
    bd.node_id = _ptr(fb.node_id, C.c_int32)
    bd.node_rack = _ptr(fb.node_rack, C.c_int32)
    bd.node_pool_len = int(fb.node_id.shape[0])
    return bd


@dataclass
class HostOutputs:
    out: np.ndarray                  # int32 out pool
    topic_results: np.ndarray        # TOPIC_RESULT_DTYPE [T]
    scenario_results: np.ndarray     # SCENARIO_RESULT_DTYPE [S]
    ctx: np.ndarray                  # int32 ctx pool after the solve


def host_tables(fb: FlatBatch, out_len: Optional[int] = None) -> "tuple[abi.Tables, HostOutputs]":
    """Allocate host outputs and build a kas_tables of HOST pointers (out_len: cells of the out array when
    only selected scenarios' rows come back, kas_solve_host_select)."""
    n_out = fb.out_len if out_len is None else out_len
    ho = HostOutputs(
        out=np.full(max(n_out, 1), -2, dtype=np.int32),
        topic_results=np.zeros(max(fb.n_topics, 1), dtype=abi.TOPIC_RESULT_DTYPE),
        scenario_results=np.zeros(max(fb.n_scenarios, 1), dtype=abi.SCENARIO_RESULT_DTYPE),
        ctx=fb.ctx.copy(),
    )
    t = abi.Tables()
    t.cur = fb.cur.ctypes.data
    t.out = ho.out.ctypes.data
    t.aux = fb.aux.ctypes.data if fb.aux.size else None
    t.ctx = ho.ctx.ctypes.data if ho.ctx.size else None
    t.topic_results = ho.topic_results.ctypes.data
    t.scenario_results = ho.scenario_results.ctypes.data
    t.cur_len = int(fb.cur.shape[0])
    t.out_len = int(n_out)
    t.aux_len = int(fb.aux.shape[0])
    t.ctx_len = int(ho.ctx.shape[0])
    return t, ho


def flatten(scenarios: Sequence[Scenario], ctx_width: int = abi.KAS_MAX_WIDTH) -> FlatBatch:
    """Flatten reference-shaped scenarios into the ABI's pools and descriptors."""
    scen = np.zeros(len(scenarios), dtype=abi.SCENARIO_DESC_DTYPE)
    tdescs = []
    node_id: List[int] = []
    node_rack: List[int] = []
    cur: List[int] = []
    aux: List[int] = []
    ctx: List[int] = []
    row_ids: List[np.ndarray] = []
    out_len = 0
    for si, sc in enumerate(scenarios):
        ids = sorted(set(int(b) for b in sc.brokers))
        rack_index: Dict[str, int] = {}
        racks = []
        for b in ids:
            name = sc.racks.get(b)
            if name is None:
                name = str(b)                      # KAS:82-86
            racks.append(rack_index.setdefault(name, len(rack_index)))
        scen[si]["n_nodes"] = len(ids)
        scen[si]["topic_begin"] = len(tdescs)
        scen[si]["topic_count"] = len(sc.topics)
        scen[si]["node_off"] = len(node_id)
        node_id.extend(ids)
        node_rack.extend(racks)
        if sc.context is not None or sc.want_context:
            scen[si]["ctx_width"] = ctx_width
            scen[si]["ctx_off"] = len(ctx)
            table = np.zeros((len(ids), ctx_width), dtype=np.int64)
            for n, b in enumerate(ids):
                for r, v in (sc.context or {}).get(b, {}).items():
                    if r >= ctx_width:
                        raise ValueError("context replica index beyond ctx_width")
                    table[n, r] = v
            ctx.extend(int(v) for v in table.reshape(-1))
        else:
            scen[si]["ctx_width"] = 0
            scen[si]["ctx_off"] = -1
        for tp in sc.topics:
            keys = set(int(p) for p in tp.current.keys())
            parts = keys if tp.partitions is None else set(int(p) for p in tp.partitions)
            rows = sorted(keys | parts)
            P = len(rows)
            lens = [len(tp.current.get(p, ())) for p in rows]
            cur_width = max(lens) if lens else 0
            out_width = max(cur_width, int(tp.rf), 1)
            td = np.zeros((), dtype=abi.TOPIC_DESC_DTYPE)
            td["name_hash"] = tp.name_hash
            td["n_partitions"] = P
            td["cur_width"] = cur_width
            td["rf"] = int(tp.rf)
            td["out_width"] = out_width
            td["cur_off"] = len(cur)
            td["out_off"] = out_len
            for p, ln in zip(rows, lens):
                reps = [int(x) for x in tp.current.get(p, ())]
                cur.extend(reps + [-1] * (cur_width - ln))
            if any(ln != cur_width for ln in lens):
                td["cur_len_off"] = len(aux)
                aux.extend(lens)
            else:
                td["cur_len_off"] = -1
            if parts != set(rows):
                td["in_partitions_off"] = len(aux)
                aux.extend(1 if p in parts else 0 for p in rows)
            else:
                td["in_partitions_off"] = -1
            if rows != list(range(P)):
                td["part_id_off"] = len(aux)
                aux.extend(rows)
            else:
                td["part_id_off"] = -1
            out_len += P * out_width
            tdescs.append(td)
            row_ids.append(np.asarray(rows, dtype=np.int32))
    topics = (np.stack(tdescs) if tdescs else np.zeros(0, dtype=abi.TOPIC_DESC_DTYPE))
    return FlatBatch(
        scen=scen, topics=topics.astype(abi.TOPIC_DESC_DTYPE),
        node_id=np.asarray(node_id, dtype=np.int32), node_rack=np.asarray(node_rack, dtype=np.int32),
        cur=np.asarray(cur if cur else [0], dtype=np.int32)[: max(len(cur), 1)],
        aux=np.asarray(aux, dtype=np.int32), ctx=np.asarray(ctx, dtype=np.int32),
        out_len=out_len, row_ids=row_ids)


def uniform_batch(cur: np.ndarray, node_id: np.ndarray, node_rack: np.ndarray, rf: int,
                  name_hash: Union[int, Sequence[int]] = 3644,
                  shared_cur: bool = False) -> FlatBatch:
    """Batch of S single-topic scenarios of identical shape, straight from arrays.

    cur:       int32 [S, P, W]  (or [P, W] with shared_cur=True: every scenario reads the same
               base assignment — the what-if mode of SURVEY.md 8d/C4)
    node_id:   int32 [S, N] ascending per row;  node_rack: int32 [S, N]
    name_hash: one Java hashCode for all topics (default "t0".hashCode() == 3644) or one per scenario
    """
    node_id = np.ascontiguousarray(node_id, dtype=np.int32)
    node_rack = np.ascontiguousarray(node_rack, dtype=np.int32)
    S, N = node_id.shape
    if shared_cur:
        P, W = cur.shape
    else:
        assert cur.shape[0] == S
        _, P, W = cur.shape
    cur = np.ascontiguousarray(cur, dtype=np.int32)
    ow = max(W, rf, 1)
    scen = np.zeros(S, dtype=abi.SCENARIO_DESC_DTYPE)
    scen["n_nodes"] = N
    scen["topic_begin"] = np.arange(S)
    scen["topic_count"] = 1
    scen["ctx_width"] = 0
    scen["node_off"] = np.arange(S, dtype=np.int64) * N
    scen["ctx_off"] = -1
    topics = np.zeros(S, dtype=abi.TOPIC_DESC_DTYPE)
    topics["name_hash"] = np.asarray(name_hash, dtype=np.int32)
    topics["n_partitions"] = P
    topics["cur_width"] = W
    topics["rf"] = rf
    topics["out_width"] = ow
    topics["cur_off"] = 0 if shared_cur else np.arange(S, dtype=np.int64) * (P * W)
    topics["out_off"] = np.arange(S, dtype=np.int64) * (P * ow)
    topics["cur_len_off"] = -1
    topics["in_partitions_off"] = -1
    topics["part_id_off"] = -1
    return FlatBatch(scen=scen, topics=topics, node_id=node_id.reshape(-1),
                     node_rack=node_rack.reshape(-1), cur=cur.reshape(-1),
                     aux=np.zeros(0, dtype=np.int32), ctx=np.zeros(0, dtype=np.int32),
                     out_len=S * P * ow,
                     row_ids=[])


def node_set_batch(node_ids: Sequence[np.ndarray], node_racks: Sequence[np.ndarray], P: int, W: int,
                   rf: int, name_hash: int = 3644, shared_cur: bool = False,
                   cur: Optional[np.ndarray] = None) -> FlatBatch:
    """Descriptors for S single-topic scenarios whose broker sets differ in size (each scenario
    has its own node table) over cur tables of one shape [P, W].  `cur` may be None when the
    table lives only in HBM (bench); offsets assume scenario s at s*P*W (or 0 if shared_cur)."""
    S = len(node_ids)
    ow = max(W, rf, 1)
    scen = np.zeros(S, dtype=abi.SCENARIO_DESC_DTYPE)
    topics = np.zeros(S, dtype=abi.TOPIC_DESC_DTYPE)
    lens = np.asarray([len(x) for x in node_ids], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]) if S else np.zeros(0, np.int64)
    scen["n_nodes"] = lens
    scen["topic_begin"] = np.arange(S)
    scen["topic_count"] = 1
    scen["ctx_width"] = 0
    scen["node_off"] = offs
    scen["ctx_off"] = -1
    topics["name_hash"] = name_hash
    topics["n_partitions"] = P
    topics["cur_width"] = W
    topics["rf"] = rf
    topics["out_width"] = ow
    topics["cur_off"] = 0 if shared_cur else np.arange(S, dtype=np.int64) * (P * W)
    topics["out_off"] = np.arange(S, dtype=np.int64) * (P * ow)
    topics["cur_len_off"] = -1
    topics["in_partitions_off"] = -1
    topics["part_id_off"] = -1
    return FlatBatch(
        scen=scen, topics=topics,
        node_id=np.concatenate(node_ids).astype(np.int32) if S else np.zeros(0, np.int32),
        node_rack=np.concatenate(node_racks).astype(np.int32) if S else np.zeros(0, np.int32),
        cur=(np.ascontiguousarray(cur, dtype=np.int32).reshape(-1) if cur is not None
             else np.zeros(1, np.int32)),
        aux=np.zeros(0, np.int32), ctx=np.zeros(0, np.int32), out_len=S * P * ow)


def unflatten_topic(fb: FlatBatch, out: np.ndarray, topic_index: int) -> Dict[int, List[int]]:
    """Rebuild the reference's return value (TreeMap partition -> preference list, KAS:221-238)
    for one topic: rows that hold no replica at all are not keys of the map."""
    td = fb.topics[topic_index]
    P, ow = int(td["n_partitions"]), int(td["out_width"])
    rows = out[int(td["out_off"]): int(td["out_off"]) + P * ow].reshape(P, ow)
    ids = fb.row_ids[topic_index] if fb.row_ids else np.arange(P, dtype=np.int32)
    result: Dict[int, List[int]] = {}
    for i in range(P):
        lst = [int(v) for v in rows[i] if v != -1]
        if lst:
            result[int(ids[i])] = lst
    return result


def unflatten_context(fb: FlatBatch, ctx: np.ndarray, scenario_index: int) -> Dict[int, Dict[int, int]]:
    """Context counters back as {broker id: {replica index: count}} (zero entries dropped)."""
    sd = fb.scen[scenario_index]
    if sd["ctx_off"] < 0:
        return {}
    N, cw = int(sd["n_nodes"]), int(sd["ctx_width"])
    tab = ctx[int(sd["ctx_off"]): int(sd["ctx_off"]) + N * cw].reshape(N, cw)
    ids = fb.node_id[int(sd["node_off"]): int(sd["node_off"]) + N]
    return {int(ids[n]): {r: int(tab[n, r]) for r in range(cw) if tab[n, r]}
            for n in range(N) if tab[n].any()}


# ---- 16-bit cells (kas_solve_host16, include/kas_abi.h): cur / out as node indices ---------------------------------
def _topic_scenarios(fb: FlatBatch) -> np.ndarray:
    """scenario index per topic"""
    owner = np.full(fb.n_topics, -1, dtype=np.int64)
    for s in range(fb.n_scenarios):
        b, c = int(fb.scen["topic_begin"][s]), int(fb.scen["topic_count"][s])
        owner[b:b + c] = s
    return owner


def to_cells16(fb: FlatBatch) -> np.ndarray:
    """The cur pool as uint16 node indices: cell i = position of the broker id in its scenario's (ascending) node table,
    KAS_CELL16_NONE for a broker that is not in the scenario's broker set (and for the cells behind a short row).
    A cur table shared by scenarios with different broker sets (the what-if layout) has no such form: ValueError."""
    cur16 = np.full(fb.cur.shape[0], abi.KAS_CELL16_NONE, dtype=np.uint16)
    owner = _topic_scenarios(fb)
    done: Dict[int, int] = {}
    for t in range(fb.n_topics):
        td, s = fb.topics[t], int(owner[t])
        if s < 0:
            continue
        n, off = int(fb.scen["n_nodes"][s]), int(fb.scen["node_off"][s])
        if n > 32767:
            raise ValueError("more than 32,767 brokers do not fit 16-bit cells (KAS_N_LIMIT)")
        ids = fb.node_id[off:off + n]
        lo, cells = int(td["cur_off"]), int(td["n_partitions"]) * int(td["cur_width"])
        if cells <= 0:
            continue
        if lo in done:
            o = done[lo]
            po, pn = int(fb.scen["node_off"][o]), int(fb.scen["n_nodes"][o])
            if pn != n or not np.array_equal(fb.node_id[po:po + pn], ids):
                raise ValueError("a cur table shared by scenarios with different broker sets has no node-index form")
            continue
        done[lo] = s
        c = fb.cur[lo:lo + cells]
        pos = np.searchsorted(ids, c)
        hit = (pos < n) & (ids[np.minimum(pos, max(n - 1, 0))] == c) if n else np.zeros(cells, bool)
        cur16[lo:lo + cells] = np.where(hit, pos, abi.KAS_CELL16_NONE).astype(np.uint16)
    return cur16


def cells16_to_ids(fb: FlatBatch, out16: np.ndarray) -> np.ndarray:
    """An out pool of node indices back as int32 broker ids (KAS_CELL16_NONE -> -1, cells no topic owns -> -2 as
    host_tables leaves them)."""
    out = np.full(out16.shape[0], -2, dtype=np.int32)
    owner = _topic_scenarios(fb)
    for t in range(fb.n_topics):
        td, s = fb.topics[t], int(owner[t])
        if s < 0:
            continue
        n, off = int(fb.scen["n_nodes"][s]), int(fb.scen["node_off"][s])
        ids = np.concatenate([fb.node_id[off:off + n], np.zeros(1, np.int32)])
        lo, cells = int(td["out_off"]), int(td["n_partitions"]) * int(td["out_width"])
        c = out16[lo:lo + cells].astype(np.int64)
        pad = c == abi.KAS_CELL16_NONE
        out[lo:lo + cells] = np.where(pad, -1, ids[np.where(pad, n, np.minimum(c, n))])
    return out


def index_form(fb: FlatBatch) -> FlatBatch:
    """The batch a 16-bit call solves, in int32: node i of every scenario has id i and cur holds node indices (-1 for a
    broker that is not in the broker set).  kas_solve_host on it returns the cells and the digests kas_solve_host16
    returns for `fb` (tests/test_cells16.py holds both to the oracle)."""
    c16 = to_cells16(fb)
    ident = np.zeros_like(fb.node_id)
    for s in range(fb.n_scenarios):
        n, off = int(fb.scen["n_nodes"][s]), int(fb.scen["node_off"][s])
        ident[off:off + n] = np.arange(n, dtype=np.int32)
    cur = np.where(c16 == abi.KAS_CELL16_NONE, -1, c16.astype(np.int32)).astype(np.int32)
    return FlatBatch(scen=fb.scen, topics=fb.topics, node_id=ident, node_rack=fb.node_rack, cur=cur, aux=fb.aux,
                     ctx=fb.ctx, out_len=fb.out_len, row_ids=fb.row_ids)


def host_tables16(fb: FlatBatch, cur16: np.ndarray, out_len: Optional[int] = None) -> "tuple[abi.Tables, HostOutputs]":
    """host_tables for kas_solve_host16: HostOutputs.out is a uint16 pool of node indices."""
    t, ho = host_tables(fb, out_len=out_len)
    ho.out = np.full(max(int(t.out_len), 1), 0xFFFE, dtype=np.uint16)
    t.cur = cur16.ctypes.data
    t.out = ho.out.ctypes.data
    t.cur_len = int(cur16.shape[0])
    return t, ho
