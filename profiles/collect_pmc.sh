#!/bin/bash
# rocprofv3 passes behind the numbers in DESIGN.md / bench.py ("roofline.traffic").
# Run on the GPU box from the repo root:  bash profiles/collect_pmc.sh
# Counters are collected in their own runs (separate --pmc passes, no trace domains besides
# --kernel-trace), as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r01
mkdir -p $O
B="python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extra --no-dense-leg"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/sf_mb profiles/streaming_microbench.hip
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o sparse -- $B > $O/bench_sparse.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o dense -- $B --dense > $O/bench_dense.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc -o sparse_fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc -o sparse_write -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc -o dense_fetch -- $B --dense > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc -o dense_write -- $B --dense > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc -o calib_fetch -- /tmp/sf_mb > $O/calib.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $O/pmc -o sparse_sq -- $B > /dev/null 2>&1
rm -f $O/stats/*_kernel_trace.csv
python - <<'PY'
import csv, glob, collections, json, os
O="gpurun_out/r01"
def per_kernel(path):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        kn=r["Kernel_Name"]
        kn="k_step" if "k_step" in kn else ("k_select" if "k_select" in kn else kn[:48])
        acc[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc
out={}
for tag in ("sparse","dense"):
    tot={}
    for cn in ("fetch","write"):
        acc=per_kernel(f"{O}/pmc/{tag}_{cn}_counter_collection.csv")
        s=0.0
        for (k,c),v in acc.items():
            if k in ("k_step","k_select"):
                v=v[20:1020]         # the timed 1000 steps (after 20 warm-up launches)
                s+=sum(v)/len(v)
        tot[cn]=s
    out[tag]=tot
cal=per_kernel(f"{O}/pmc/calib_fetch_counter_collection.csv")
out["calibration_fetch_kb"]={k[0]:sum(v)/len(v) for k,v in cal.items()}
json.dump(out, open(f"{O}/pmc_summary_raw.json","w"), indent=1)
acc=per_kernel(f"{O}/pmc/sparse_sq_counter_collection.csv")
with open(f"{O}/sq_counters_sparse.csv","w") as f:
    f.write("kernel,counter,mean_over_timed_steps\n")
    for (k,c),v in sorted(acc.items()):
        if k in ("k_step","k_select"):
            v=v[20:1020]; f.write(f"{k},{c},{sum(v)/len(v):.1f}\n")
print(json.dumps(out, indent=1))
for f in glob.glob(f"{O}/pmc/*.csv"): os.remove(f)
PY
