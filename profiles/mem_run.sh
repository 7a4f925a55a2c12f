#!/bin/bash
# L1 (TCP) / TA / L2 (TCC) counters of the k_run launches of bench.py (separate --pmc passes, kernel trace only).
# usage: bash profiles/mem_run.sh <outdir> [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$1; shift
mkdir -p $O
B="python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extra $@"
rocprofv3 -L > $O/counters_available.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $O/pmc -o m1 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $O/pmc -o m2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/pmc -o m3 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum -d $O/pmc -o m4 -- $B > /dev/null 2>&1
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmc/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_run" in r["Kernel_Name"] and "rebuild" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(f.split("/")[-1], k, "launches", len(v), "max %.4g" % max(v))
PY
rm -f $O/pmc/*.csv
