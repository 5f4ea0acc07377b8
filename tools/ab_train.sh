#!/bin/bash
# Same-box A/B of the 24-scene f16 training step: tools/ab_train.sh "ENV=a" "ENV=b" ...   ("-" = no switch)
for e in "$@"; do
  [ "$e" = "-" ] && e="VS_NOOP=1"
  env $e timeout 400 python tools/bench_train.py --scenes 24 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json
t=json.loads(sys.stdin.read()); print('$e', 'ms_per_step', t['ms_per_step'], 'loss', t.get('loss'), 'gnorm', t.get('grad_norm'))"
done
