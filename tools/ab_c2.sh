for i in 1 2; do
  unset TXM_LIBRARY
  python3 bench.py --config c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step'],3), d['step_breakdown_ms']['bootstrap_call'], d['step_breakdown_ms']['sampler_tile_counts'])"
  export TXM_LIBRARY=$PWD/tools/build/libtxmom_${ABLIB:-oldw8}.so
  python3 bench.py --config c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('${ABLIB:-oldw8}', round(d['ms_per_step'],3), d['step_breakdown_ms']['bootstrap_call'], d['step_breakdown_ms']['sampler_tile_counts'])"
done
