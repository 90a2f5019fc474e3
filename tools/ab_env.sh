#!/bin/bash
# In-run A/B of an environment switch: alternates `python bench.py` without and with the given assignment(s).
# usage (on the GPU box): tools/ab_env.sh "RT_HIP_BLOCKING_POLL=1" [bench args...]
set -u
SW=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
one() {  # label, env assignment or empty
  local out
  out=$(env $2 timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline --no-alone "${@:3}") || return 1
  echo "$out" | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.read()); print('%-28s value=%8.1f ms/step=%7.2f' % ('$1', j['value'], j['ms_per_step']))"
}
for i in 1 2; do
  one "default" "RT_AB_DUMMY=0" "$@" || exit 1
  one "$SW" "$SW" "$@" || exit 1
done
