#!/bin/bash
# runs each argument as a command under its own timeout and appends output to gpurun_out/ab_step.log as it goes, so a hang in
# one step is visible and costs only that step's limit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/ab_step.log
for cmd in "$@"; do
  echo "== $cmd" >> gpurun_out/ab_step.log
  timeout ${STEP_TIMEOUT:-180} bash -c "$cmd" >> gpurun_out/ab_step.log 2>&1
  echo "== rc=$?" >> gpurun_out/ab_step.log
done
tail -${TAIL:-60} gpurun_out/ab_step.log
