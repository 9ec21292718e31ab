#!/bin/bash
# kernel durations of the split attention instances (rocprofv3 --kernel-trace --stats of tools/attn_bench.py --split); prints the stats rows
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/attn_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attn_prof -- python3 $ROOT/tools/attn_bench.py --split > /tmp/attn_prof.log 2>&1
python3 - <<'P'
import csv,glob
f=glob.glob('/tmp/attn_prof/**/*kernel_stats.csv',recursive=True)
if not f:
    print(open('/tmp/attn_prof.log').read()[-2000:])
for r in csv.DictReader(open(f[0])) if f else []:
    if 'relattn' in r['Name'] or 'attn_pack' in r['Name']:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
P
