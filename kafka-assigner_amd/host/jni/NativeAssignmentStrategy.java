// Java side of the JNI boundary (see INTEGRATION.md).  Not compiled in this repository's image
// (there is no JDK); a maintainer of the reference adds this file next to
// KafkaAssignmentStrategy.java and routes getRackAwareAssignment (KAS:40-63) through solve().
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

    static final int WIDTH = 8;          // KAS_MAX_WIDTH
    static final int HEADER_INTS = 8;    // {nameHash, P, curWidth, rf, outWidth, N, hasCtx, reserved}

    /** Layout of `in` (int32, native order): header[8], nodeId[N], nodeRack[N], partId[P],
     *  curLen[P], inPartitions[P], cur[P*curWidth], ctx[N*8].
     *  Layout of `out`: {status, failPartition, movedReplicas, movedPartitions}, out[P*outWidth],
     *  ctx[N*8]. */
    static native int solveBatch(ByteBuffer in, ByteBuffer out);

    static Map<Integer, List<Integer>> solve(String topic, Map<Integer, List<Integer>> cur,
            Map<Integer, String> racks, Set<Integer> nodes, Set<Integer> partitions, int rf,
            Map<Integer, Map<Integer, Integer>> counters) {
        TreeSet<Integer> nodeSet = new TreeSet<Integer>(nodes);
        TreeMap<Integer, List<Integer>> rows = new TreeMap<Integer, List<Integer>>(cur);
        int n = nodeSet.size(), p = rows.size(), cw = 0;
        for (List<Integer> l : rows.values()) cw = Math.max(cw, l.size());
        int ow = Math.max(Math.max(cw, rf), 1);
        if (ow > WIDTH) throw new IllegalStateException("replica lists longer than " + WIDTH);
        ByteBuffer in = ByteBuffer.allocateDirect(4 * (HEADER_INTS + 2 * n + 3 * p + p * cw + n * WIDTH))
                .order(ByteOrder.nativeOrder());
        in.putInt(topic.hashCode()).putInt(p).putInt(cw).putInt(rf).putInt(ow).putInt(n)
          .putInt(counters != null ? 1 : 0).putInt(0);
        for (int id : nodeSet) in.putInt(id);
        Map<String, Integer> rackIndex = new HashMap<String, Integer>();
        for (int id : nodeSet) {                       // KAS:82-86: missing rack = own id string
            String r = racks.containsKey(id) ? racks.get(id) : Integer.toString(id);
            Integer k = rackIndex.get(r);
            if (k == null) { k = rackIndex.size(); rackIndex.put(r, k); }
            in.putInt(k);
        }
        for (int part : rows.keySet()) in.putInt(part);
        for (List<Integer> l : rows.values()) in.putInt(l.size());
        for (int part : rows.keySet()) in.putInt(partitions.contains(part) ? 1 : 0);
        for (List<Integer> l : rows.values())
            for (int k = 0; k < cw; ++k) in.putInt(k < l.size() ? l.get(k) : -1);
        for (int id : nodeSet)
            for (int k = 0; k < WIDTH; ++k) {
                Map<Integer, Integer> c = counters != null ? counters.get(id) : null;
                Integer v = c != null ? c.get(k) : null;
                in.putInt(v != null ? v : 0);
            }
        ByteBuffer out = ByteBuffer.allocateDirect(4 * (4 + p * ow + n * WIDTH)).order(ByteOrder.nativeOrder());
        int rc = solveBatch(in, out);
        if (rc != 0) throw new IllegalStateException("native solver error " + rc);
        int status = out.getInt(0), failPartition = out.getInt(4);
        if (status == 1)
            throw new IllegalStateException("Partition " + failPartition + " could not be fully assigned!");
        if (status == 4) throw new ArrayIndexOutOfBoundsException();
        if (status != 0) throw new IllegalStateException("solver status " + status);
        Map<Integer, List<Integer>> result = new TreeMap<Integer, List<Integer>>();
        int row = 0;
        for (int part : rows.keySet()) {
            List<Integer> l = new ArrayList<Integer>();
            for (int k = 0; k < ow; ++k) {
                int b = out.getInt(4 * (4 + row * ow + k));
                if (b >= 0) l.add(b);
            }
            if (!l.isEmpty()) result.put(part, l);
            ++row;
        }
        if (counters != null) {
            int i = 0;
            for (int id : nodeSet) {
                Map<Integer, Integer> c = new HashMap<Integer, Integer>();
                for (int k = 0; k < WIDTH; ++k) {
                    int v = out.getInt(4 * (4 + p * ow + i * WIDTH + k));
                    if (v != 0) c.put(k, v);
                }
                counters.put(id, c);
                ++i;
            }
        }
        return result;
    }
}
