#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/attn_scan
rocprofv3 --kernel-trace --output-format csv -d /tmp/attn_scan -- python3 $ROOT/tools/attn_scan.py "$@" > /tmp/attn_scan.log 2>&1
python3 $ROOT/tools/attn_scan.py --read /tmp/attn_scan "$@" || tail -20 /tmp/attn_scan.log
