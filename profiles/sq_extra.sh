#!/bin/bash
# extra SQ counter passes for k_step / k_select (C3, default bench run); prints means over the timed steps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r01x; mkdir -p $O
B="python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extra --no-dense-leg"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $O -o p1 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_CYCLES -d $O -o p2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_PENDING_STALL_CYCLES TA_TA_BUSY TCP_CACHE_MISS TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES -d $O -o p3 -- $B > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
O="gpurun_out/r01x"
for f in sorted(glob.glob(O+"/*_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn=r["Kernel_Name"]
        kn="k_step" if "k_step" in kn else ("k_select" if "k_select" in kn else None)
        if kn: acc[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,c),v in sorted(acc.items()):
        v=v[20:1020]
        print(f"{k},{c},{sum(v)/len(v):.1f}")
PY
rm -f $O/*.csv
