#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
timeout 200 tools/issue_probe 300 > $O/issue_probe.log 2>&1; echo "issue_probe exit $?"
export TMPDIR=/tmp; cd /tmp
timeout 60 rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1; echo "rocprofv3 -L exit $?"
