package siftscience.kafka.tools;

import java.io.IOException;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.nio.file.Paths;
import java.util.ArrayList;
import java.util.List;
import java.util.Map;
import java.util.Set;
import java.util.TreeMap;
import java.util.TreeSet;

import org.json.JSONArray;
import org.json.JSONObject;

/**
 * CPU baseline on a real JVM: times the untouched reference solver
 * (KafkaTopicAssigner.generateAssignment, KafkaTopicAssigner.java:42-72) on the scenarios that
 * tools/export_scenarios.py writes (the same seeded G(seed, P, N, R, RF) inputs and broker-set
 * actions bench.py solves on the MI355X), one snapshot file per scenario.  Not run in the build
 * image (no JDK there): bench.py's cpu_baseline is the C restatement, kind "port".
 *
 *   java -cp ... siftscience.kafka.tools.JavaRefBench scen_0000.json scen_0001.json ...
 *
 * Each file: {"brokers":[{"id","rack"}...], "solve_brokers":[ids the scenario solves with],
 * "partitions":[{"topic","partition","replicas"}...]}.  Prints scenarios/s (single thread, after
 * one untimed warm-up pass over the first file) and per-file moved-replica counts to compare with
 * the GPU records.
 */
public final class JavaRefBench {
    private JavaRefBench() {}

    private static long solve(String path, boolean print) throws IOException {
        JSONObject snap = new JSONObject(new String(Files.readAllBytes(Paths.get(path)), StandardCharsets.UTF_8));
        Map<Integer, String> racks = new TreeMap<Integer, String>();
        JSONArray bs = snap.getJSONArray("brokers");
        for (int i = 0; i < bs.length(); ++i) {
            JSONObject b = bs.getJSONObject(i);
            if (b.has("rack") && !b.isNull("rack")) {
                racks.put(b.getInt("id"), b.getString("rack"));
            }
        }
        Set<Integer> brokers = new TreeSet<Integer>();
        JSONArray sb = snap.getJSONArray("solve_brokers");
        for (int i = 0; i < sb.length(); ++i) {
            brokers.add(sb.getInt(i));
        }
        Map<String, Map<Integer, List<Integer>>> cur = new TreeMap<String, Map<Integer, List<Integer>>>();
        List<String> order = new ArrayList<String>();
        JSONArray ps = snap.getJSONArray("partitions");
        for (int i = 0; i < ps.length(); ++i) {
            JSONObject p = ps.getJSONObject(i);
            String topic = p.getString("topic");
            if (!cur.containsKey(topic)) {
                cur.put(topic, new TreeMap<Integer, List<Integer>>());
                order.add(topic);
            }
            List<Integer> reps = new ArrayList<Integer>();
            JSONArray rs = p.getJSONArray("replicas");
            for (int k = 0; k < rs.length(); ++k) {
                reps.add(rs.getInt(k));
            }
            cur.get(topic).put(p.getInt("partition"), reps);
        }
        long t0 = System.nanoTime();
        KafkaTopicAssigner assigner = new KafkaTopicAssigner();
        long moved = 0;
        String failure = null;
        for (String topic : order) {
            Map<Integer, List<Integer>> before = cur.get(topic);
            try {
                Map<Integer, List<Integer>> after = assigner.generateAssignment(topic, before, brokers, racks, -1);
                for (Map.Entry<Integer, List<Integer>> e : after.entrySet()) {
                    Set<Integer> was = new TreeSet<Integer>(before.get(e.getKey()));
                    for (Integer b : e.getValue()) {
                        if (!was.contains(b)) {
                            moved += 1;
                        }
                    }
                }
            } catch (IllegalStateException e) {
                failure = e.getMessage();
                break;
            }
        }
        long dt = System.nanoTime() - t0;
        if (print) {
            System.out.println(path + "\tmoved_replicas=" + (failure == null ? moved : 0)
                    + "\t" + (failure == null ? "OK" : failure) + "\t" + (dt / 1e6) + " ms");
        }
        return dt;
    }

    public static void main(String[] args) throws IOException {
        if (args.length < 1) {
            System.err.println("usage: JavaRefBench scenario.json...");
            System.exit(2);
        }
        solve(args[0], false);                              // JIT warm-up, untimed
        long total = 0;
        for (String a : args) {
            total += solve(a, true);
        }
        System.out.println("scenarios/s (1 thread, solve only): " + (args.length / (total / 1e9)));
    }
}
