#!/bin/bash
# rocprofv3 passes behind the numbers in DESIGN.md / bench.py ("roofline.traffic").
# Run on the GPU box from the repo root:  bash profiles/collect_pmc.sh [round-tag]
# Counters are collected in their own runs (separate --pmc passes, no trace domains besides
# --kernel-trace), as MI355X_MICROARCH.md prescribes.  usage: collect_pmc.sh [tag] [steps] [warmup]
# (default window = bench.py's default, --steps 1000 --warmup 20; the driver's window is 20 5).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r02}
STEPS=${2:-1000}
WARM=${3:-20}
O=gpurun_out/$R
mkdir -p $O
B="python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extra --repeats 2"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/sf_mb profiles/streaming_microbench.hip
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o default -- $B > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o perstep -- $B --fused 0 > $O/bench_under_rocprof_perstep.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc -o fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc -o write -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc -o calib_fetch -- /tmp/sf_mb > $O/calib.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $O/pmc -o sq -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $O/pmc -o sq2 -- $B > /dev/null 2>&1
# the per-dispatch rows of the resident launch (k_run) out of the kernel trace: the timed launch's duration can be read off a tracked file
python - <<PY2
import csv, glob
for f in glob.glob("$O/stats/default_kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "k_run" in r["Kernel_Name"] and "rebuild" not in r["Kernel_Name"]]
    with open("$O/kernel_trace_k_run.csv", "w") as o:
        o.write("dispatch,kernel,start_ns,end_ns,duration_us,grid,workgroup,lds_bytes,vgpr,sgpr\n")
        for i, r in enumerate(rows):
            o.write("%d,%s,%s,%s,%.2f,%s,%s,%s,%s,%s\n" % (i, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(", ", " ")[-40:], r["Start_Timestamp"], r["End_Timestamp"],
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                    r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("SGPR_Count", "")))
PY2
rm -f $O/stats/*_kernel_trace.csv
R=$R STEPS=$STEPS WARM=$WARM python - <<'PY'
import csv, glob, collections, json, os
R=os.environ["R"]; O=f"gpurun_out/{R}"
def launches(path, want):
    """counter values of the launches of kernel `want`, in launch order"""
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        kn=r["Kernel_Name"]
        if want in kn and "rebuild" not in kn and "k_run_tiles" not in kn:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
bench=json.load(open(f"{O}/bench_under_rocprof.json"))
kernel=bench["roofline"]["kernel"]
# bench.py launches k_run for: the dress rehearsal (W + K steps), the warm-up (W), the timed rollout (K), then measure(): W, K, W, K (counted).
# The K-step launches are the large ones: take the largest value per counter (they are identical rollouts).
out={"workload": bench["config"]["workload"], "kernel": kernel, "steps": bench["steps"], "warmup": bench["warmup"],
     "command": "bash profiles/collect_pmc.sh <tag> %s %s (rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE | WRITE_SIZE in separate passes -- python bench.py --steps %s --warmup %s --no-cpu-baseline --no-extra)" % (os.environ["STEPS"], os.environ["WARM"], os.environ["STEPS"], os.environ["WARM"]),
     "per_launch": "one launch of k_run = the whole K-step rollout of all environments (the window bench.py times)"}
raw={}
for cn,f in (("fetch","fetch"),("write","write")):
    acc=launches(f"{O}/pmc/{f}_counter_collection.csv", "k_run")
    for k,v in acc.items(): raw[k]=v[3] if len(v)>=8 else max(v)      # launches: rehearsal W, K; W, K (the wall-timed one); measure(): W, K, W, K
out["raw_kb"]=raw
cal=collections.defaultdict(list)
for r in csv.DictReader(open(f"{O}/pmc/calib_fetch_counter_collection.csv")):
    cal[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
out["calibration_fetch_kb"]={k:sum(v)/len(v) for k,v in cal.items()}
out["correction"]="FETCH_SIZE x 2 (gfx950, 16 B/lane loads: MI355X_MICROARCH.md HBM section, re-checked by profiles/streaming_microbench.hip); WRITE_SIZE as is (uncalibrated); counters are in KB"
out["hbm_bytes_per_launch"]=(raw.get("FETCH_SIZE",0)*2+raw.get("WRITE_SIZE",0))*1024
out["hbm_bytes_per_step"]=out["hbm_bytes_per_launch"]/bench["steps"]
out["algorithmic_bytes_per_launch"]=bench["roofline"]["algorithmic_bytes_per_launch"]
out["note"]="FETCH_SIZE counts L2 fills from the fabric (HBM or the 256 MB memory-side cache)"
json.dump(out, open(f"{O}/pmc_traffic.json","w"), indent=1)
with open(f"{O}/sq_counters.csv","w") as f:
    f.write("kernel,counter,value_of_the_K_step_launch\n")
    for p in ("sq","sq2"):
        for k,v in sorted(launches(f"{O}/pmc/{p}_counter_collection.csv","k_run").items()):
            f.write(f"{kernel},{k},{(v[3] if len(v)>=8 else max(v)):.0f}\n")
print(json.dumps(out, indent=1))
for f in glob.glob(f"{O}/pmc/*.csv"): os.remove(f)
PY
