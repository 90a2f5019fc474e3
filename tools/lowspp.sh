#!/bin/bash
# Frame rate of the headline scene at low sample counts (a rank's share of a strong-scaling job), for several pass sizes.
# usage (GPU box): bash tools/lowspp.sh [log2 of the paths per pass ...]   (no argument: the library's default)
set -e
mkdir -p gpurun_out/lowspp
for spp in 16 32 64 128 256; do
  for pl in ${@:-default}; do
    if [ "$pl" = default ]; then unset RT_HIP_PATHS_IN_FLIGHT; else export RT_HIP_PATHS_IN_FLIGHT=$pl; fi
    timeout -k 10 120 python3 bench.py --steps 6 --warmup 2 --spp $spp --no-cpu-baseline --no-alone > gpurun_out/lowspp/o.json 2> gpurun_out/lowspp/o.err
    python3 -c "
import json;j=json.load(open('gpurun_out/lowspp/o.json'));print('spp $spp paths per pass 2^$pl: %.0f Msamples/s, %.2f ms, %.0f passes'%(j['value'],j['ms_per_step'],j['passes_per_step']))"
  done
done
