"""Property test: the C oracle and the line-by-line Python restatement agree on random small
clusters — including ragged current lists, duplicate/dead brokers in a list, rack-less brokers,
`partitions` != keys(current) (direct KAS callers), RF changes (Q5/Q6), several
topics sharing one Context, and the failure paths."""
import os
import sys

from hypothesis import HealthCheck, given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import literal_ref  # noqa: E402
from kafka_assigner_amd import abi  # noqa: E402
from kafka_assigner_amd.flatten import (Scenario, Topic, flatten, unflatten_context,  # noqa: E402
                                        unflatten_topic)
from oracle_lib import oracle_solve  # noqa: E402


@st.composite
def scenarios(draw):
    n_brokers = draw(st.integers(1, 9))
    base = draw(st.sampled_from([0, 10, 1000]))
    stride = draw(st.sampled_from([1, 1, 3]))
    brokers = [base + stride * i for i in range(n_brokers)]
    n_racks = draw(st.integers(1, 4))
    rack_mode = draw(st.sampled_from(["all", "none", "some", "numeric"]))
    racks = {}
    for b in brokers:
        if rack_mode == "all" or (rack_mode == "some" and draw(st.booleans())):
            racks[b] = "r%d" % draw(st.integers(0, n_racks - 1))
        elif rack_mode == "numeric" and draw(st.booleans()):
            racks[b] = str(draw(st.sampled_from(brokers)))     # Q3: collides with a fallback id
    universe = brokers + [base - 1, base + 100]                  # includes dead brokers
    n_topics = draw(st.integers(1, 3))
    topics = []
    for t in range(n_topics):
        n_parts = draw(st.integers(0, 12))
        ids = sorted(draw(st.sets(st.integers(0, 20), min_size=n_parts, max_size=n_parts)))
        w = draw(st.integers(0, 4))
        ragged = draw(st.booleans())
        cur = {}
        for p in ids:
            ln = draw(st.integers(0, w)) if ragged else w
            cur[p] = [draw(st.sampled_from(universe)) for _ in range(ln)]
        rf = draw(st.integers(0, 5))
        parts = None
        if draw(st.integers(0, 3)) == 0:
            parts = set(draw(st.sets(st.integers(0, 22), max_size=12)))
        name = draw(st.sampled_from(["test", "t0", "topic-%d" % t, "polygenelubricants", "x"]))
        topics.append((name, cur, rf, parts))
    return brokers, racks, topics


def literal_run(brokers, racks, topics):
    """Run the literal restatement with CLI semantics; returns per-topic (status, fail, map)."""
    ctx = literal_ref.Context()
    results = []
    failed = False
    for name, cur, rf, parts in topics:
        if failed:
            results.append((abi.KAS_SKIPPED, -1, None))
            continue
        partitions = set(cur.keys()) if parts is None else set(parts)
        if not rf > 0:
            results.append((abi.KAS_FAIL_RF_NOT_POSITIVE, -1, None)); failed = True; continue
        if not rf <= len(brokers):
            results.append((abi.KAS_FAIL_RF_GT_BROKERS, -1, None)); failed = True; continue
        try:
            new = literal_ref.get_rack_aware_assignment(name, cur, racks, set(brokers),
                                                        partitions, rf, ctx)
            results.append((abi.KAS_OK, -1, new))
        except literal_ref.IllegalStateException as e:
            p = int(str(e).split()[1])
            results.append((abi.KAS_FAIL_UNASSIGNABLE, p, None)); failed = True
        except literal_ref.ArrayIndexOutOfBoundsException:
            results.append((abi.KAS_FAIL_HASH_INDEX, -1, None)); failed = True
    return results, ctx.counter, failed


@settings(max_examples=400, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenarios())
def test_oracle_equals_literal(sc):
    brokers, racks, topics = sc
    want, want_ctx, failed = literal_run(brokers, racks, topics)
    fb = flatten([Scenario(brokers=brokers, racks=racks, want_context=True,
                           topics=[Topic(n, c, rf, parts) for n, c, rf, parts in topics])])
    ho = oracle_solve(fb)
    moved_r = moved_p = 0
    for t, (status, fail_p, new) in enumerate(want):
        tr = ho.topic_results[t]
        assert int(tr["status"]) == status, (t, tr)
        assert int(tr["fail_partition"]) == fail_p
        got = unflatten_topic(fb, ho.out, t)
        if status == abi.KAS_OK:
            assert got == new
            cur = topics[t][1]
            rows = set(cur) | (set(topics[t][3]) if topics[t][3] is not None else set())
            mr = sum(len(set(new.get(p, [])) - set(cur.get(p, []))) for p in rows)
            mp = sum(1 for p in rows if set(new.get(p, [])) != set(cur.get(p, [])))
            assert (int(tr["moved_replicas"]), int(tr["moved_partitions"])) == (mr, mp)
            moved_r += mr; moved_p += mp
        else:
            assert got == {}
    sr = ho.scenario_results[0]
    assert int(sr["moved_replicas"]) == moved_r and int(sr["moved_partitions"]) == moved_p
    first_bad = next((t for t, w in enumerate(want) if w[0] not in (abi.KAS_OK,)), -1)
    assert int(sr["fail_topic"]) == first_bad
    # digest = sum of per-cell contributions over the emitted lists
    d = 0
    for t, (status, _, new) in enumerate(want):
        if status != abi.KAS_OK:
            continue
        ids = list(fb.row_ids[t])
        for p, lst in new.items():
            for r, b in enumerate(lst):
                d = (d + abi.digest_cell(t, ids.index(p), r, b)) & ((1 << 64) - 1)
    assert int(sr["digest"]) == d
    hash_fail = any(w[0] == abi.KAS_FAIL_HASH_INDEX for w in want)
    if not hash_fail:   # a KAS:190 failure inside P5 leaves counters partially updated
        got_ctx = unflatten_context(fb, ho.ctx, 0)
        want_clean = {n: {r: c for r, c in m.items() if c} for n, m in want_ctx.items()}
        want_clean = {n: m for n, m in want_clean.items() if m}
        assert got_ctx == want_clean
