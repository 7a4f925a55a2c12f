cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/icache; mkdir -p $O
for wl in c3 c5; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $O/$wl -o ic -- python bench.py --workload $wl --steps 300 --warmup 20 --no-cpu-baseline --no-extra --no-dense-leg > /dev/null 2>$O/$wl.err
  python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open("$O/$wl/ic_counter_collection.csv")):
        kn=r["Kernel_Name"].replace("(anonymous namespace)::","")[:40]
        if "k_run" in kn: acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn,d in acc.items():
        print("$wl", kn, {k:max(v) for k,v in d.items()})
except Exception as e:
    print("ERR", e); print(open("$O/$wl.err").read()[-800:])
PY
done
