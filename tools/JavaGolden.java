package siftscience.kafka.tools;

import java.io.IOException;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.nio.file.Paths;
import java.util.ArrayList;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;
import java.util.Set;
import java.util.TreeMap;
import java.util.TreeSet;

import org.json.JSONArray;
import org.json.JSONObject;

/**
 * Golden-vector generator for sites that have a JDK: drives the UNTOUCHED reference classes
 * (KafkaTopicAssigner.generateAssignment, KafkaTopicAssigner.java:42-72, one instance = one Context
 * across topics like KafkaAssignmentGenerator.java:172-184) on a cluster snapshot in the format
 * kafka-assigner_amd/host/kas_cli.cpp reads, and prints what PRINT_REASSIGNMENT would print for it:
 *
 *   { "brokers":    [ {"id": 1, "host": "h1", "port": 9092, "rack": "a"}, ... ],
 *     "partitions": [ {"topic": "t", "partition": 0, "replicas": [1, 2, 3]}, ... ] }
 *
 * Not compiled in the build image (no JDK there).  Place it in the reference tree under
 * src/main/java/siftscience/kafka/tools/ (it uses only that tree's own dependencies: org.json),
 * then
 *
 *   mvn -q package && java -cp target/classes:... siftscience.kafka.tools.JavaGolden snapshot.json \
 *       [--brokers 1,2,3] [--disable_rack_awareness] [--desired_replication_factor N] > java.json
 *   python tools/compare_golden.py snapshot.json java.json          # against the MI355X solver
 *
 * Output: one JSON object {"version":1,"partitions":[...],"failed":{"topic":..,"message":..}?};
 * compare as PARSED JSON (org.json's key order is JVM dependent, SURVEY.md Q11).
 */
public final class JavaGolden {
    private JavaGolden() {}

    public static void main(String[] args) throws IOException {
        if (args.length < 1) {
            System.err.println("usage: JavaGolden snapshot.json [--brokers a,b,c] "
                    + "[--disable_rack_awareness] [--desired_replication_factor N]");
            System.exit(2);
        }
        String text = new String(Files.readAllBytes(Paths.get(args[0])), StandardCharsets.UTF_8);
        JSONObject snap = new JSONObject(text);
        Set<Integer> brokers = new TreeSet<Integer>();
        boolean rackAware = true;
        int desiredRf = -1;
        Set<Integer> brokerOverride = null;
        for (int i = 1; i < args.length; ++i) {
            if (args[i].equals("--disable_rack_awareness")) {
                rackAware = false;
            } else if (args[i].equals("--desired_replication_factor")) {
                desiredRf = Integer.parseInt(args[++i]);
            } else if (args[i].equals("--brokers")) {
                brokerOverride = new TreeSet<Integer>();
                for (String s : args[++i].split(",")) {
                    brokerOverride.add(Integer.parseInt(s.trim()));
                }
            }
        }
        Map<Integer, String> racks = new TreeMap<Integer, String>();
        JSONArray bs = snap.getJSONArray("brokers");
        for (int i = 0; i < bs.length(); ++i) {
            JSONObject b = bs.getJSONObject(i);
            int id = b.getInt("id");
            brokers.add(id);
            if (rackAware && b.has("rack") && !b.isNull("rack")) {
                racks.put(id, b.getString("rack"));
            }
        }
        if (brokerOverride != null) {
            brokers = brokerOverride;
        }
        // topics in first-appearance order, partitions ascending (KafkaAssignmentGenerator.java:172-184)
        Map<String, Map<Integer, List<Integer>>> cur =
                new LinkedHashMap<String, Map<Integer, List<Integer>>>();
        JSONArray ps = snap.getJSONArray("partitions");
        for (int i = 0; i < ps.length(); ++i) {
            JSONObject p = ps.getJSONObject(i);
            String topic = p.getString("topic");
            Map<Integer, List<Integer>> m = cur.get(topic);
            if (m == null) {
                m = new TreeMap<Integer, List<Integer>>();
                cur.put(topic, m);
            }
            List<Integer> reps = new ArrayList<Integer>();
            JSONArray rs = p.getJSONArray("replicas");
            for (int k = 0; k < rs.length(); ++k) {
                reps.add(rs.getInt(k));
            }
            m.put(p.getInt("partition"), reps);
        }

        KafkaTopicAssigner assigner = new KafkaTopicAssigner();   // one Context for the whole run
        JSONObject out = new JSONObject();
        out.put("version", 1);
        JSONArray parts = new JSONArray();
        for (Map.Entry<String, Map<Integer, List<Integer>>> t : cur.entrySet()) {
            Map<Integer, List<Integer>> result;
            try {
                result = assigner.generateAssignment(t.getKey(), t.getValue(), brokers, racks, desiredRf);
            } catch (RuntimeException e) {
                // the CLI would die here with a stack trace; record what it died of and stop
                JSONObject f = new JSONObject();
                f.put("topic", t.getKey());
                f.put("exception", e.getClass().getName());
                f.put("message", String.valueOf(e.getMessage()));
                out.put("failed", f);
                break;
            }
            for (Map.Entry<Integer, List<Integer>> e : new TreeMap<Integer, List<Integer>>(result).entrySet()) {
                JSONObject pj = new JSONObject();
                pj.put("topic", t.getKey());
                pj.put("partition", e.getKey().intValue());
                pj.put("replicas", new JSONArray(e.getValue()));
                parts.put(pj);
            }
        }
        out.put("partitions", parts);
        System.out.println(out.toString());
    }
}
