#!/bin/bash
# builds nothing; run from the repo root on the GPU box: bash profiles/collect_pmc.sh
# rocprofv3 passes for profiles/ (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r01
mkdir -p $O
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra --no-dense-leg"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc -o sparse_fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc -o sparse_write -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc -o dense_fetch -- $B --dense > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc -o dense_write -- $B --dense > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $O/pmc -o sparse_sq -- $B > /dev/null 2>&1
# keep only the small summaries
python - <<'PY'
import csv, glob, collections, json, os
O="gpurun_out/r01"
def per_kernel(path, skip=20, n=300):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        kn=r["Kernel_Name"]; kn="k_step" if "k_step" in kn else ("k_select" if "k_select" in kn else kn[:40]); acc[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc
out={}
for tag in ("sparse","dense"):
    tot={}
    for cn in ("fetch","write"):
        acc=per_kernel(f"{O}/pmc/{tag}_{cn}_counter_collection.csv")
        s=0.0
        for (k,c),v in acc.items():
            if "k_step" in k or "k_select" in k:
                v=v[20:320]          # the timed 300 steps (after 20 warm-up launches)
                s+=sum(v)/len(v)
        tot[cn]=s
    out[tag]=tot
json.dump(out, open(f"{O}/pmc_summary_raw.json","w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
for f in glob.glob(f"{O}/pmc/*_counter_collection.csv")+glob.glob(f"{O}/pmc/*_kernel_trace.csv")+glob.glob(f"{O}/pmc/*agent_info.csv"):
    if "sparse_sq" in f and "counter_collection" in f: continue
    os.remove(f)
PY
