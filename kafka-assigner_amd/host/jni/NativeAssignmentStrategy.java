// Java side of the JNI boundary (see INTEGRATION.md).  Not compiled in this repository's image
// (there is no JDK); a maintainer of the reference adds this file next to
// KafkaAssignmentStrategy.java and routes getRackAwareAssignment (KAS:40-63) through solve(), or
// a whole PRINT_REASSIGNMENT run / a set of what-if broker sets through solveScenarios().
package siftscience.kafka.tools;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.Map;
import java.util.Set;
import java.util.TreeMap;
import java.util.TreeSet;

final class NativeAssignmentStrategy {
    static { System.loadLibrary("kas_jni"); }

    static final int LAYOUT = 4;         // KAS_JNI_LAYOUT of kas_jni.cpp
    static final int WIDTH = 8;          // KAS_MAX_WIDTH
    static final int HEADER_INTS = 12;   // {LAYOUT, S, T, nodePoolLen, curLen, auxLen, ctxLen, outLen, device, nSelect, cellBits, 0}
    /** 16-bit cells (kas_solve_host16, ABI v5): cur[] and out[] travel as positions in the scenario's ascending broker
     *  list, two bytes a cell — half the bytes over the host link.  Used whenever no cur table is shared by several
     *  scenarios (a what-if batch shares ONE table among different broker sets, which has no such form) and every
     *  broker set has at most 32,767 members; false = int32 broker ids always. */
    static volatile boolean cells16 = true;
    /** HIP device the calls of this JVM run on (one native context per device). */
    static volatile int device = 0;

    /** One batch: S scenarios, each a broker set + rack map and an ordered run of topics that share
     *  one Context.  Layout of `in` (int32 / int64 fields, native order): header[12] (HEADER_INTS), S scenario
     *  descriptors (32 bytes: nNodes, topicBegin, topicCount, ctxWidth, long nodeOff, long ctxOff),
     *  T topic descriptors (64 bytes: nameHash, nPartitions, curWidth, rf, outWidth, reserved,
     *  long curOff, outOff, curLenOff, inPartitionsOff, partIdOff), nodeId[], nodeRack[], cur[],
     *  aux[], ctx[], select[].  Layout of `out`: T topic results (status, failPartition, movedReplicas,
     *  movedPartitions), S scenario results (32 bytes), out[], ctx[].  nSelect = -1: out[] holds every
     *  row; nSelect >= 0 (what-if): only the rows of the scenarios select[] names, packed in that order.
     *  cellBits = 16: cur[] / out[] hold unsigned 16-bit node indices (0xFFFF: not in the broker set / pad), two to an
     *  int, (len + 1) / 2 ints each; lengths and offsets count cells. */
    static native int solveBatch(ByteBuffer in, ByteBuffer out);

    /** One topic of a scenario, in the reference's own argument types (KAS:40-43). */
    static final class TopicRequest {
        final String topic;
        final Map<Integer, List<Integer>> currentAssignment;
        final Set<Integer> partitions;       // null = keys(currentAssignment), as KTA:50-54 passes
        final int replicationFactor;
        TopicRequest(String topic, Map<Integer, List<Integer>> cur, Set<Integer> partitions, int rf) {
            this.topic = topic; this.currentAssignment = cur; this.partitions = partitions;
            this.replicationFactor = rf;
        }
    }

    /** One scenario: what one PRINT_REASSIGNMENT run solves (KAG:172-184). */
    static final class ScenarioRequest {
        final Set<Integer> nodes;
        final Map<Integer, String> racks;
        final List<TopicRequest> topics = new ArrayList<TopicRequest>();
        /** Context counters in/out (KAS:360-369); null = start empty, do not report back. */
        Map<Integer, Map<Integer, Integer>> counters;
        ScenarioRequest(Set<Integer> nodes, Map<Integer, String> racks) { this.nodes = nodes; this.racks = racks; }
    }

    /** Outcome of one topic: the new assignment, or the exception the reference would have thrown. */
    static final class TopicOutcome {
        int status, failPartition, movedReplicas, movedPartitions;
        String topic;            // for the exception texts of KTA:65-69
        int replicationFactor;
        Map<Integer, List<Integer>> assignment;
        /** The statuses of include/kas_abi.h as the exceptions (and texts) the reference throws. */
        RuntimeException failure() {
            switch (status) {
                case 0: return null;
                case 1:         // KAS:183-184
                    return new IllegalStateException("Partition " + failPartition + " could not be fully assigned!");
                case 2:         // KTA:65-66
                    return new IllegalStateException("Topic " + topic + " does not have a positive replication factor!");
                case 3:         // KTA:67-69
                    return new IllegalStateException("Topic " + topic + " has a higher replication factor ("
                            + replicationFactor + ") than available brokers!");
                case 4:         // KAS:190 with hashCode() == Integer.MIN_VALUE
                    return new ArrayIndexOutOfBoundsException();
                case 5:         // KTA:58-60 (raised while resolving rf on the host, never by the kernels).  The reference's
                                // text names the partition and its list size ("Topic <t> has partition <p> with unexpected
                                // replication factor <n>"); a topic status carries neither — the C ABI gives both
                                // (kas_resolve_replication_factor -> kas_rf_result) and the exact text (kas_failure_text),
                                // and the Java mirror below resolves rf itself and throws the exact text, so this
                                // approximation is never reached through either.
                    return new IllegalStateException("Topic " + topic + " has a partition with unexpected replication factor");
                case 6: return new IllegalStateException("skipped: an earlier topic of the run failed");
                case 7: return new IllegalArgumentException("broker ids must be non-negative and distinct; at most 32768 racks");
                default: return new IllegalStateException("solver status " + status);
            }
        }
    }

    /** Drop-in body for getRackAwareAssignment: one scenario, one topic. */
    static Map<Integer, List<Integer>> solve(String topic, Map<Integer, List<Integer>> cur,
            Map<Integer, String> racks, Set<Integer> nodes, Set<Integer> partitions, int rf,
            Map<Integer, Map<Integer, Integer>> counters) {
        ScenarioRequest sc = new ScenarioRequest(nodes, racks);
        sc.counters = counters;
        sc.topics.add(new TopicRequest(topic, cur, partitions, rf));
        List<ScenarioRequest> batch = new ArrayList<ScenarioRequest>();
        batch.add(sc);
        TopicOutcome o = solveScenarios(batch).get(0).get(0);
        RuntimeException e = o.failure();
        if (e != null) throw e;
        return o.assignment;
    }

    /** The batch path: every scenario is independent, the topics of a scenario run in order
     *  against its Context and the first failure skips the rest (the CLI run aborts there). */
    static List<List<TopicOutcome>> solveScenarios(List<ScenarioRequest> batch) {
        return solveScenarios(batch, null);
    }

    /** The what-if form (KAG:131-187: one snapshot, many broker sets, ONE assignment printed): every
     *  scenario is solved and reports status and movement counts, but only the scenarios whose index is
     *  in `select` (ascending) get their `assignment` maps — the rows of the others never leave the GPU.
     *  select == null: every scenario's rows come back. */
    static List<List<TopicOutcome>> solveScenarios(List<ScenarioRequest> batch, int[] select) {
        final int S = batch.size();
        int T = 0;
        long nodePool = 0, curLen = 0, auxLen = 0, ctxLen = 0, outLen = 0;
        // rows of a topic: keys(cur) UNION partitions, ascending (a member of `partitions` without a
        // current list is an all-orphan row, KAS:150-157)
        // A what-if batch hands the SAME currentAssignment (and partitions) objects to every scenario: such
        // topics share one cur table and one set of aux arrays in the payload (identity, not equality — no
        // table is ever compared), so S broker sets over one snapshot upload the snapshot once.
        List<List<TreeMap<Integer, List<Integer>>>> rowsOf = new ArrayList<List<TreeMap<Integer, List<Integer>>>>();
        java.util.IdentityHashMap<Object, java.util.IdentityHashMap<Object, TreeMap<Integer, List<Integer>>>> seen =
            new java.util.IdentityHashMap<Object, java.util.IdentityHashMap<Object, TreeMap<Integer, List<Integer>>>>();
        java.util.IdentityHashMap<Object, long[]> placed = new java.util.IdentityHashMap<Object, long[]>();   // rows -> {curOff, auxOff}
        final Object NO_PARTITIONS = new Object();
        boolean anyShared = false, small = true;
        for (ScenarioRequest sc : batch) {
            small = small && sc.nodes.size() <= 32767;
            List<TreeMap<Integer, List<Integer>>> perTopic = new ArrayList<TreeMap<Integer, List<Integer>>>();
            for (TopicRequest t : sc.topics) {
                java.util.IdentityHashMap<Object, TreeMap<Integer, List<Integer>>> byParts = seen.get(t.currentAssignment);
                if (byParts == null) {
                    byParts = new java.util.IdentityHashMap<Object, TreeMap<Integer, List<Integer>>>();
                    seen.put(t.currentAssignment, byParts);
                }
                final Object partsKey = t.partitions != null ? t.partitions : NO_PARTITIONS;
                TreeMap<Integer, List<Integer>> rows = byParts.get(partsKey);
                final boolean shared = rows != null;
                anyShared = anyShared || shared;
                if (!shared) {
                    rows = new TreeMap<Integer, List<Integer>>(t.currentAssignment);
                    byParts.put(partsKey, rows);
                }
                if (!shared && t.partitions != null)
                    for (int part : t.partitions)
                        if (!rows.containsKey(part)) rows.put(part, new ArrayList<Integer>());
                int cw = 0;
                for (List<Integer> l : rows.values()) cw = Math.max(cw, l.size());
                int ow = Math.max(Math.max(cw, Math.min(t.replicationFactor, sc.nodes.size())), 1);
                if (ow > WIDTH) throw new IllegalStateException("replica lists longer than " + WIDTH);
                perTopic.add(rows);
                if (!shared) { curLen += (long) rows.size() * cw; auxLen += 3L * rows.size(); }
                outLen += (long) rows.size() * ow;
                ++T;
            }
            rowsOf.add(perTopic);
            nodePool += sc.nodes.size();
            if (sc.counters != null) ctxLen += (long) sc.nodes.size() * WIDTH;
        }
        boolean[] wanted = new boolean[S];
        long selLen = 0;
        if (select != null) {
            int prev = -1;
            for (int s : select) {
                if (s <= prev || s >= S) throw new IllegalArgumentException("select must be ascending scenario indices");
                prev = s; wanted[s] = true;
            }
        }
        {   // rows of the selected scenarios (all of them when select == null)
            int s = 0;
            for (ScenarioRequest sc : batch) {
                List<TreeMap<Integer, List<Integer>>> perTopic = rowsOf.get(s);
                for (int k = 0; k < sc.topics.size(); ++k) {
                    TreeMap<Integer, List<Integer>> rows = perTopic.get(k);
                    int cw = 0;
                    for (List<Integer> l : rows.values()) cw = Math.max(cw, l.size());
                    int ow = Math.max(Math.max(cw, Math.min(sc.topics.get(k).replicationFactor, sc.nodes.size())), 1);
                    if (select == null || wanted[s]) selLen += (long) rows.size() * ow;
                }
                ++s;
            }
        }
        final int nSelect = select == null ? -1 : select.length;
        final long retLen = select == null ? outLen : selLen;       // cells of out[] that come back
        final boolean c16 = cells16 && !anyShared && small;
        final long curInts = c16 ? (curLen + 1) / 2 : curLen, retInts = c16 ? (retLen + 1) / 2 : retLen;
        long inInts = HEADER_INTS + 8L * S + 16L * T + 2 * nodePool + curInts + auxLen + ctxLen + Math.max(nSelect, 0);
        long outInts = 4L * T + 8L * S + retInts + ctxLen;
        // (a direct ByteBuffer holds < 2 GiB: a larger batch must be split by the caller, not truncated)
        ByteBuffer in = ByteBuffer.allocateDirect(Math.toIntExact(4 * inInts)).order(ByteOrder.nativeOrder());
        ByteBuffer out = ByteBuffer.allocateDirect(Math.toIntExact(4 * outInts)).order(ByteOrder.nativeOrder());
        in.putInt(LAYOUT).putInt(S).putInt(T).putInt(Math.toIntExact(nodePool)).putInt(Math.toIntExact(curLen))
          .putInt(Math.toIntExact(auxLen)).putInt(Math.toIntExact(ctxLen)).putInt(Math.toIntExact(retLen))
          .putInt(device).putInt(nSelect).putInt(c16 ? 16 : 32).putInt(0);
        // ---- descriptors
        long nodeOff = 0, ctxOff = 0;
        int topicBegin = 0;
        for (ScenarioRequest sc : batch) {
            in.putInt(sc.nodes.size()).putInt(topicBegin).putInt(sc.topics.size())
              .putInt(sc.counters != null ? WIDTH : 0).putLong(nodeOff).putLong(sc.counters != null ? ctxOff : -1L);
            nodeOff += sc.nodes.size(); topicBegin += sc.topics.size();
            if (sc.counters != null) ctxOff += (long) sc.nodes.size() * WIDTH;
        }
        long curOff = 0, outOff = 0, auxOff = 0;
        for (int s = 0; s < S; ++s) {
            ScenarioRequest sc = batch.get(s);
            for (int k = 0; k < sc.topics.size(); ++k) {
                TopicRequest t = sc.topics.get(k);
                TreeMap<Integer, List<Integer>> rows = rowsOf.get(s).get(k);
                int p = rows.size(), cw = 0;
                for (List<Integer> l : rows.values()) cw = Math.max(cw, l.size());
                int ow = Math.max(Math.max(cw, Math.min(t.replicationFactor, sc.nodes.size())), 1);
                long[] at = placed.get(rows);                       // a shared table keeps the offsets of its first use
                if (at == null) {
                    at = new long[] {curOff, auxOff};
                    placed.put(rows, at);
                    curOff += (long) p * cw; auxOff += 3L * p;
                }
                in.putInt(t.topic.hashCode()).putInt(p).putInt(cw).putInt(t.replicationFactor).putInt(ow).putInt(0)
                  .putLong(at[0]).putLong(outOff)
                  .putLong(at[1] + p)          // curLen[P]
                  .putLong(at[1] + 2L * p)     // inPartitions[P]
                  .putLong(at[1]);             // partId[P]
                outOff += (long) p * ow;
            }
        }
        // ---- node pools: ids ascending; rack = dense index of the rack string, a broker without a
        // rack is its own rack named by its id (KAS:82-86, string equality as in KAS:90-94)
        List<TreeSet<Integer>> nodeSets = new ArrayList<TreeSet<Integer>>();
        for (ScenarioRequest sc : batch) {
            TreeSet<Integer> ns = new TreeSet<Integer>(sc.nodes);
            nodeSets.add(ns);
            for (int id : ns) in.putInt(id);
        }
        for (int s = 0; s < S; ++s) {
            Map<String, Integer> rackIndex = new HashMap<String, Integer>();
            for (int id : nodeSets.get(s)) {
                Map<Integer, String> racks = batch.get(s).racks;
                String r = racks != null && racks.containsKey(id) ? racks.get(id) : Integer.toString(id);
                Integer k = rackIndex.get(r);
                if (k == null) { k = rackIndex.size(); rackIndex.put(r, k); }
                in.putInt(k);
            }
        }
        // ---- cur pool, then aux pool (per topic: partId[P], curLen[P], inPartitions[P]); a shared table is
        // written where it was first placed and nowhere else
        java.util.IdentityHashMap<Object, Boolean> written = new java.util.IdentityHashMap<Object, Boolean>();
        // (16-bit cells: the position of a broker in its scenario's ascending list; no table is shared then)
        List<int[]> idOf = new ArrayList<int[]>();
        for (int s = 0; s < S; ++s) {
            Map<Integer, Integer> indexOf = null;
            if (c16) {
                indexOf = new HashMap<Integer, Integer>();
                int[] ids = new int[nodeSets.get(s).size()];
                int n = 0;
                for (int id : nodeSets.get(s)) { indexOf.put(id, n); ids[n++] = id; }
                idOf.add(ids);
            }
            for (TreeMap<Integer, List<Integer>> rows : rowsOf.get(s)) {
                if (written.put(rows, Boolean.TRUE) != null) continue;
                int cw = 0;
                for (List<Integer> l : rows.values()) cw = Math.max(cw, l.size());
                for (List<Integer> l : rows.values())
                    for (int k = 0; k < cw; ++k) {
                        if (!c16) { in.putInt(k < l.size() ? l.get(k) : -1); continue; }
                        Integer at = k < l.size() ? indexOf.get(l.get(k)) : null;
                        in.putShort((short) (at != null ? at : 0xFFFF));
                    }
            }
        }
        if (c16 && (curLen & 1) != 0) in.putShort((short) 0);
        written.clear();
        for (int s = 0; s < S; ++s)
            for (int k = 0; k < batch.get(s).topics.size(); ++k) {
                TopicRequest t = batch.get(s).topics.get(k);
                TreeMap<Integer, List<Integer>> rows = rowsOf.get(s).get(k);
                if (written.put(rows, Boolean.TRUE) != null) continue;
                for (int part : rows.keySet()) in.putInt(part);
                for (List<Integer> l : rows.values()) in.putInt(l.size());
                for (int part : rows.keySet()) in.putInt(t.partitions == null || t.partitions.contains(part) ? 1 : 0);
            }
        // ---- Context counters in
        for (int s = 0; s < S; ++s) {
            Map<Integer, Map<Integer, Integer>> counters = batch.get(s).counters;
            if (counters == null) continue;
            for (int id : nodeSets.get(s))
                for (int k = 0; k < WIDTH; ++k) {
                    Map<Integer, Integer> c = counters.get(id);
                    Integer v = c != null ? c.get(k) : null;
                    in.putInt(v != null ? v : 0);
                }
        }
        if (select != null) for (int s : select) in.putInt(s);
        int rc = solveBatch(in, out);
        if (rc != 0) throw new IllegalStateException("native solver error " + rc);
        // ---- results
        List<List<TopicOutcome>> result = new ArrayList<List<TopicOutcome>>();
        int ti = 0;
        long rowBase = 0;                    // cell index inside out[], which starts at byte 4 * (4T + 8S) of `out`
        final long rowBytes = 4 * (4L * T + 8L * S);
        long ctxBase = 4L * T + 8L * S + retInts;
        for (int s = 0; s < S; ++s) {
            ScenarioRequest sc = batch.get(s);
            List<TopicOutcome> outcomes = new ArrayList<TopicOutcome>();
            for (int k = 0; k < sc.topics.size(); ++k, ++ti) {
                TreeMap<Integer, List<Integer>> rows = rowsOf.get(s).get(k);
                int cw = 0;
                for (List<Integer> l : rows.values()) cw = Math.max(cw, l.size());
                int ow = Math.max(Math.max(cw, Math.min(sc.topics.get(k).replicationFactor, sc.nodes.size())), 1);
                TopicOutcome o = new TopicOutcome();
                o.topic = sc.topics.get(k).topic; o.replicationFactor = sc.topics.get(k).replicationFactor;
                o.status = out.getInt(4 * (4 * ti)); o.failPartition = out.getInt(4 * (4 * ti + 1));
                o.movedReplicas = out.getInt(4 * (4 * ti + 2)); o.movedPartitions = out.getInt(4 * (4 * ti + 3));
                final boolean haveRows = select == null || wanted[s];
                if (o.status == 0 && haveRows) {
                    o.assignment = new TreeMap<Integer, List<Integer>>();
                    long row = 0;
                    for (int part : rows.keySet()) {
                        List<Integer> l = new ArrayList<Integer>();
                        for (int c = 0; c < ow; ++c) {
                            final long cellAt = rowBase + row * ow + c;
                            if (c16) {
                                int v = out.getShort((int) (rowBytes + 2 * cellAt)) & 0xFFFF;
                                if (v != 0xFFFF) l.add(idOf.get(s)[v]);
                            } else {
                                int b = out.getInt((int) (rowBytes + 4 * cellAt));
                                if (b >= 0) l.add(b);
                            }
                        }
                        if (!l.isEmpty()) o.assignment.put(part, l);   // a row nobody holds is not a key (KAS:205-214)
                        ++row;
                    }
                }
                if (haveRows) rowBase += (long) rows.size() * ow;
                outcomes.add(o);
            }
            if (sc.counters != null) {
                for (int id : nodeSets.get(s)) {
                    Map<Integer, Integer> c = new HashMap<Integer, Integer>();
                    for (int k = 0; k < WIDTH; ++k) {
                        int v = out.getInt((int) (4 * (ctxBase + k)));
                        if (v != 0) c.put(k, v);
                    }
                    sc.counters.put(id, c);
                    ctxBase += WIDTH;
                }
            }
            result.add(outcomes);
        }
        return result;
    }
}
