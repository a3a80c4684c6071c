#!/bin/bash
# One parametrised GPU session (run through gpurun): `tools/gpu_session.sh <name> <step> [<step> ...]`.
# Every step writes under gpurun_out/<name>/ ; steps are joined with && semantics (a failed / timed-out GPU step ends the
# session).  Steps:
#   smoke            __graft_entry__.smoke()
#   pytest[:EXPR]    pytest -m gpu (optionally -k EXPR)
#   bench[:ARGS]     python3 bench.py ARGS   (ARGS with '+' for spaces, e.g. bench:--fused+--steps+100)
#   ab:ENV:VARIANTS  tools/ab_inproc.py VARIANTS (';'-separated) with ENV (','-separated A=B)
#   py:SCRIPT[:ARGS] python3 SCRIPT ARGS
#   prof             tools/profile_gpu.sh (kernel-trace + PMC passes of the default bench)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
O=$R/gpurun_out/$NAME
mkdir -p $O
cd $R
n=0
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" == "$step" ] && rest=""
  tag=$(printf "%02d_%s" $n $kind)
  case $kind in
    smoke)  timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/$tag.log 2>&1 ;;
    pytest) if [ -n "$rest" ]; then timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu -k "$rest" > $O/$tag.log 2>&1; else timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/$tag.log 2>&1; fi ;;
    bench)  timeout -k 10 600 python3 bench.py ${rest//+/ } > $O/$tag.json 2> $O/$tag.err ;;
    ab)     envs=${rest%%:*}; vars=${rest#*:}; IFS=';' read -ra V <<< "$vars"
            env ${envs//,/ } timeout -k 10 900 python3 tools/ab_inproc.py "${V[@]}" > $O/$tag.log 2>&1 ;;
    py)     script=${rest%%:*}; a=${rest#*:}; [ "$a" == "$rest" ] && a=""
            timeout -k 10 900 python3 $script ${a//+/ } > $O/$tag.log 2>&1 ;;
    prof)   bash tools/profile_gpu.sh > $O/$tag.log 2>&1 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
  rc=$?
  echo "[$tag] rc=$rc  ($step)"
  for f in $O/$tag.log $O/$tag.json; do [ -f $f ] && tail -c 1500 $f | tail -4 | cut -c1-600; done
  if [ $rc -ne 0 ]; then [ -f $O/$tag.err ] && tail -5 $O/$tag.err; exit $rc; fi
done
