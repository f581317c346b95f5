#!/bin/bash
# rocprofv3 kernel trace of a python tool; the CSV is left under gpurun_out/tr for local analysis
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 "$@" > gpurun_out/tr.log 2>&1
tail -3 gpurun_out/tr.log
