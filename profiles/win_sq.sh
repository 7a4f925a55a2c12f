#!/bin/bash
# SQ counters of k_run launches of 4 and 16 updates after a reset (window phase): per-update figures = (16 - 4) / 12.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/win_sq; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $O/pmc -o sq1 -- python profiles/win_sq.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/pmc -o sq2 -- python profiles/win_sq.py "$@" > /dev/null 2>&1
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmc/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_run" in r["Kernel_Name"] and "rebuild" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        if len(v) >= 4:
            per = (v[3] - v[2]) / 12.0
            print("%-22s 4 updates %.4g   16 updates %.4g   per update %.4g   per update and env %.1f" % (k, v[2], v[3], per, per / 256))
PY
rm -rf $O/pmc
