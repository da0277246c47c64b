#!/bin/bash
# One SQ counter pass over the training step (run from the repository root through gpurun):
#   tools/pmc_sq.sh <tag> <kernel patterns> <counter> [<counter> ...]   -> gpurun_out/<tag>.json
set -u
tag=$1; pat=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pq -- python3 $R/tools/step_prof.py > /tmp/pq.log 2>&1 || { tail -5 /tmp/pq.log; exit 1; }
python3 $R/tools/pmc_table.py /tmp/pq "$pat" > $R/gpurun_out/$tag.json || exit 1
echo collected
