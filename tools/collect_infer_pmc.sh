#!/bin/bash
# HBM traffic of the inference batch: two counter passes over tools/infer_prof.py (1 eager + 7 replayed passes; the
# capture pass itself executes nothing).  Usage: tools/collect_infer_pmc.sh <tag>  -> gpurun_out/<tag>_pmc_infer_traffic.json
set -u
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pi1 /tmp/pi2
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pi1 -- python3 $R/tools/infer_prof.py > /dev/null 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pi2 -- python3 $R/tools/infer_prof.py > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_infer.py /tmp/pi1 /tmp/pi2 8 > $O/${tag}_pmc_infer_traffic.json || exit 1
echo collected
