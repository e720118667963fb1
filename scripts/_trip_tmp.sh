#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O; export TMPDIR=/tmp
val() { grep -o "\"$1\": [0-9.]*" "$2" | head -1 | cut -d' ' -f2; }
run() { n=$1; lib=$2; shift; shift; KAS_HIP_LIB=variants/libkas_hip_$lib.so timeout 400 python bench.py --no-cpu --check 0 --repeats 3 --no-extras "$@" > $O/b_$n.log 2>&1; echo "$n [$lib $*]: value $(val value $O/b_$n.log) $(grep -o '"values": \[[^]]*' $O/b_$n.log | cut -c1-90) | $(grep -o '"in_flight_launch": {[^}]*' $O/b_$n.log | cut -c22-100)"; }
run cur cur --in-flight 8
run sp1 sp1 --in-flight 8
run sp2 sp2 --in-flight 8
run k8 k8 --in-flight 8
run nap1 nap1 --in-flight 8
run nap8 nap8 --in-flight 8
run cur_b cur --in-flight 8
