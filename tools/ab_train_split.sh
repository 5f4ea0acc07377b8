#!/bin/bash
# Same-box A/B of the 8-scene split-class training step: tools/ab_train_split.sh "ENV=a" "ENV=b" ...   ("-" = no switch)
for e in "$@"; do
  [ "$e" = "-" ] && e="VS_NOOP=1"
  env $e timeout 400 python bench.py --mode train --dtype split --train-steps 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('$e', 'ms_per_step', t['ms_per_step'], 'loss', t.get('loss'), 'gnorm', t.get('grad_norm'), 'mem', t.get('peak_mem_gb'))"
done
