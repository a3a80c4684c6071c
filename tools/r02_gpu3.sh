#!/bin/bash
# round 2, GPU session 3: multi-rank comm rehearsal on one GPU + new tests
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s3
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_comm.py -x -q -m gpu > $O/pytest_comm.log 2>&1; echo "pytest comm rc=$?"; tail -30 $O/pytest_comm.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_determinism.py tests/test_gpu_shards.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -5 $O/pytest_new.log
