#!/bin/bash
# rocprofv3 kernel-trace + stats of an arbitrary python tool on the GPU box; prints the top kernels
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/ts; rm -rf $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > gpurun_out/ts.log 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
head -25 $f | cut -c1-160
