#!/bin/bash
# SQ instruction / wait counters of the k_run launches of bench.py (separate --pmc passes, kernel trace only).
# usage: bash profiles/sq_run.sh <outdir> [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$1; shift
mkdir -p $O
B="python bench.py --steps 1000 --warmup 20 --fused 2 --no-cpu-baseline --no-extra --no-dense-leg $@"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $O/pmc -o sq1 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES -d $O/pmc -o sq2 -- $B > /dev/null 2>&1
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmc/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_run" in r["Kernel_Name"] and "rebuild" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(f.split("/")[-1], k, "launches", len(v), "top3", ["%.4g" % x for x in sorted(v)[-3:]])
PY
