#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repository root through gpurun):
#   kernel traces of the training step and of the inference batch (per-kernel summary of the last repetition, timeline),
#   three counter passes over the training step (MFMA busy; HBM fetch; HBM write), each in a run of its own.
# Usage: tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*.txt|json
set -u
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr /tmp/ti /tmp/p1 /tmp/p2 /tmp/p3
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/tools/step_prof.py > /dev/null 2>&1 || exit 1
python3 $R/tools/prof_last.py /tmp/tr adam_kernel 3 80 > $O/${tag}_train_step_b128_kernels.txt || exit 1
python3 $R/tools/step_timeline.py /tmp/tr 100 > $O/${tag}_train_step_b128_timeline.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ti -- python3 $R/tools/infer_prof.py > /dev/null 2>&1 || exit 1
python3 $R/tools/prof_last.py /tmp/ti ssim_fwd_kernel 1 60 > $O/${tag}_infer_b32_512_kernels.txt || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p1 -- python3 $R/tools/step_prof.py > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_mfma.py /tmp/p1 conv_glds,conv_halo,wgrad_glds,igemm,attn_ > $O/${tag}_pmc_mfma_busy.json || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -- python3 $R/tools/step_prof.py > /dev/null 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -- python3 $R/tools/step_prof.py > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_summary.py /tmp/p2 /tmp/p3 conv_glds,conv_halo > $O/${tag}_pmc_conv_traffic.json || exit 1
echo collected
