#!/bin/bash
# runs an arbitrary command line on the GPU box with stdin closed and a hard timeout; output to gpurun_out/<tag>.log
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 bash -c "$*" < /dev/null > gpurun_out/${tag}.log 2>&1
echo "rc=$?" >> gpurun_out/${tag}.log
tail -40 gpurun_out/${tag}.log
