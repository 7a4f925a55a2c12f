#!/bin/bash
# Development aid: the register / spill / scratch table of every kernel of the product build (the compiler's own report,
# -Rpass-analysis=kernel-resource-usage), one translation unit after the other.  usage: bash profiles/resource_usage.sh > profiles/r06_resource_usage.txt
cd "$(dirname "$0")/.."
echo "# kernel resource usage, product build (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage), round 6"
echo "# $(/opt/rocm/bin/hipcc --version | grep 'HIP version')"
echo "# k_run<MAXD, ATT, DIAG, MIT, TEAM>: MIT 0 = sf_step (with the window phase where MAXD <= 2, TEAM != 2), -1 = sf_step_mitigated, -2 = the closed loop;"
echo "#   TEAM 1 = teams of a size fixed for the launch, 2 = teams that grow inside the launch (NOTEBOOK.md 5.8)"
echo "# The argument block: by value in k_run<*, *, *, 0, 0>, <2, *, *, *, 1>, <*, *, *, -2, *>, <*, *, *, *, 2>; read through the kernel-argument segment in the others (sf_run_kernels.h)"
echo "# translation unit | kernel | VGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | waves/SIMD"
echo
for u in simfire_hip simfire_hip_run2 simfire_hip_run3 simfire_hip_run4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -o /tmp/_ru.o simfire_amd/csrc/$u.hip -Rpass-analysis=kernel-resource-usage 2>&1 | \
    python3 -c "
import re, subprocess, sys
unit = '$u.hip'
rows, cur = [], None
for ln in sys.stdin:
    m = re.search(r'remark: +(.*?) \[-Rpass', ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*\)$', '', n)
    if 'Occupancy [waves/SIMD]' not in r:
        continue
    print('%-22s | %-44s | %4s | %4s | %4s | %3s | %4s | %s' % (unit, n, r.get('VGPRs', '?'), r.get('TotalSGPRs', r.get('SGPRs', '?')), r.get('SGPRs Spill', '?'),
          r.get('VGPRs Spill', '?'), r.get('ScratchSize [bytes/lane]', '?'), r.get('Occupancy [waves/SIMD]', '?')))
"
done
rm -f /tmp/_ru.o
