#!/bin/bash
# A/B of environment switches on the bench line.  usage (GPU box): bash tools/ab_env.sh "<bench args>" NAME=VALUE [NAME=VALUE ...]
# (each setting is run twice, interleaved with the default; prints Msamples/s and ms per frame)
set -e
ARGS=$1; shift
mkdir -p gpurun_out/abenv
for rnd in 1 2; do
  for kv in default "$@"; do
    if [ "$kv" = default ]; then e=A=1; else e=$kv; fi
    env $e timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-alone $ARGS > gpurun_out/abenv/o.json 2> gpurun_out/abenv/err.log
    python3 -c "
import json;j=json.load(open('gpurun_out/abenv/o.json'));print('%-28s r$rnd %8.1f Msamples/s %8.2f ms'%('$kv', j['value'], j['ms_per_step']))"
  done
done
