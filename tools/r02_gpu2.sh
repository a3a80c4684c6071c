#!/bin/bash
# round 2, GPU session 2: new reduction kernels — determinism + shard parity tests, obj timing sweep
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s2
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_determinism.py tests/test_gpu_shards.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -5 $O/pytest_new.log
for w in 256 512 1024 2048; do
  timeout -k 10 200 python3 tools/eval_loop.py --workload quadrotor --supports 1000000 --opt obj_wgs=$w > $O/loop_quad_1e6_objwgs$w.json 2>>$O/loop.err || echo "fail $w"
done
timeout -k 10 200 python3 tools/eval_loop.py --workload quadrotor --supports 16000 > $O/loop_quad_16000.json 2>>$O/loop.err
timeout -k 10 200 python3 tools/eval_loop.py --workload opf --supports 16000 > $O/loop_opf_16000.json 2>>$O/loop.err
timeout -k 10 200 python3 tools/eval_loop.py --workload farmer --supports 100000 > $O/loop_farmer_1e5.json 2>>$O/loop.err
timeout -k 10 200 python3 tools/eval_loop.py --workload farmer --supports 100000 --opt det_shared=0 > $O/loop_farmer_1e5_atomics.json 2>>$O/loop.err
timeout -k 10 200 python3 tools/eval_loop.py --workload opf --supports 10000 > $O/loop_opf_1e4.json 2>>$O/loop.err
timeout -k 10 200 python3 tools/eval_loop.py --workload opf --supports 10000 --opt det_shared=0 > $O/loop_opf_1e4_atomics.json 2>>$O/loop.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/loop_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), {k: round(v*1e3,2) for k,v in j["ms"].items()}, "loop_us", round(j["loop_ms"]*1e3,1))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -- python3 $R/tools/eval_loop.py --workload quadrotor --supports 1000000 --iters 50 > $O/stats_quad.log 2>&1 || echo "rocprof failed"
find $O/stats_quad -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_quad_1e6.csv
head -12 $O/kernel_stats_quad_1e6.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad16k -- python3 $R/tools/eval_loop.py --workload quadrotor --supports 16000 --iters 50 > $O/stats_quad16k.log 2>&1 || echo "rocprof failed"
find $O/stats_quad16k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_quad_16000.csv
head -12 $O/kernel_stats_quad_16000.csv
rm -rf $O/stats_quad $O/stats_quad16k
