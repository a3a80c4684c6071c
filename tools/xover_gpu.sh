mkdir -p gpurun_out
# A/B recipe behind profiles/r01_ab_small_grids.txt: lane-fused vs templates side by side, 16k..512k supports (run through gpurun).
for wl in quadrotor opf quadrotor_oc3 pandemic; do
for n in 16000 32000 64000 128000 256000 512000; do
for mode in "fuse_groups=1" "no_fuse=1 --opt fuse_groups=2"; do
  if [ $wl = pandemic ]; then extra="--nt $((n/100 - 10)) --nxi 100"; else extra="--supports $n"; fi
  timeout -k 10 250 python tools/eval_loop.py --workload $wl $extra --iters 50 --opt $mode 2>>gpurun_out/el.err | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl $n $mode', round(r['loop_ms'],4), round(r['graph_loop_ms'],4), {k: round(v,4) for k,v in r['ms'].items()})"
done; done; done
