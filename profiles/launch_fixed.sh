#!/bin/bash
# the fixed part of a resident launch: events, the kernel's own clock, rocprofv3's kernel duration - side by side
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lfx && rocprofv3 --kernel-trace -d /tmp/lfx -o lfx --output-format csv -- python $R/profiles/launch_fixed_probe.py "$@" 2>/dev/null | grep LAUNCH > /tmp/lfx_lines.txt
python3 - <<'PY'
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("/tmp/lfx/**/*kernel_trace.csv", recursive=True)[0])) if "k_run" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
lines = open("/tmp/lfx_lines.txt").read().splitlines()
# the probe ends with 4 launches per n (six values of n) of which the last 3 are printed
dur = dur[-24:]
k = 0
for i in range(len(dur)):
    if i % 4 == 0:
        continue
    print(lines[k], "  rocprofv3 kernel %.2f us" % dur[i]); k += 1
PY
