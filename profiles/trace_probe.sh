#!/bin/bash
# Development aid: rocprofv3 kernel trace (durations per kernel) of profiles/team_probe.py.  usage: trace_probe.sh <outdir> <probe args...>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$1; shift
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tp -- python profiles/team_probe.py "$@" > $O/probe.txt 2>&1
tail -8 $O/probe.txt
python - <<PY
import csv,glob
for f in glob.glob("$O/prof/*tp_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:100], "calls", r["Calls"], "total_ns", r["TotalDurationNs"], "avg_ns", r["AverageNs"], "max_ns", r["MaxNs"])
PY
rm -rf $O/prof
