#!/bin/bash
# One SQ counter pass over any tool script: tools/pmc_any.sh <tag> <script> <kernel patterns> <counter> ...
set -u
tag=$1; script=$2; pat=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pq -- python3 $R/$script > /tmp/pq.log 2>&1 || { tail -5 /tmp/pq.log; exit 1; }
python3 $R/tools/pmc_table.py /tmp/pq "$pat" > $R/gpurun_out/$tag.json || exit 1
echo collected
