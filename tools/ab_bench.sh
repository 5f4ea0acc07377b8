#!/bin/bash
# Quick same-box A/B of the headline step: bash tools/ab_bench.sh "VAR=a" "VAR=b" ...   (each argument: env assignments for one run, or "-" for none)
R=${GRAFT_REPO_ROOT:-$PWD}
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  env $cfg python $R/bench.py --mode infer --no-cpu-baseline --no-f32 --no-fast --no-targets70 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('step_breakdown_ms',{})
print('%-28s %.2f ms/step %.2f scenes/s | '%('$cfg' or 'default', d['ms_per_step'], d['value']) + ' '.join('%s %.1f'%(k,v) for k,v in b.items()))"
done
