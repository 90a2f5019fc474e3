#!/bin/bash
# In-run A/B of several environment settings: runs `python bench.py` once per setting per round, two rounds.
# usage (on the GPU box): tools/ab_multi.sh "A=1" "B=2 C=3" ... -- [bench args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ $# -gt 0 ] && shift
one() {  # env assignments (may be "RT_AB_DUMMY=0")
  local out
  out=$(env $1 timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline --no-alone "${@:2}") || return 1
  echo "$out" | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.read()); print('%-44s value=%8.1f ms/step=%7.2f launches/step=%5.0f' % ('$1', j['value'], j['ms_per_step'], j.get('launches_per_step', 0)))"
}
for i in 1 2; do
  one "RT_AB_DUMMY=0" "$@" || exit 1
  for s in "${SETS[@]}"; do one "$s" "$@" || exit 1; done
done
