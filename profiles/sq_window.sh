#!/bin/bash
# Development aid: SQ instruction counters of the timed k_run launch for one bench window (separate --pmc pass, kernel trace only).
# usage: bash profiles/sq_window.sh <outdir> <steps> <warmup> [more bench args]; prints per-launch totals of the K-step launch.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$1; K=$2; W=$3; shift 3
mkdir -p $O
B="python bench.py --steps $K --warmup $W --no-cpu-baseline --no-extra --no-dense-leg $@"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES -d $O/pmc -o sq_${K}_${W} -- $B > /dev/null 2>&1
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmc/*sq_${K}_${W}_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_run" in r["Kernel_Name"] and "rebuild" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print("steps $K warmup $W", k, "launches", ["%.4g" % x for x in v])
PY
rm -rf $O/pmc
