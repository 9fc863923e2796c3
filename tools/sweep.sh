#!/bin/bash
# usage: tools/sweep.sh "<vx list>" "<zu list>" "<zc list>" [extra bench args]
for vx in $1; do for zu in $2; do for zc in $3; do
timeout -k 10 120 python -u bench.py --no-cpu-baseline --vx $vx --zu $zu --z-chunk $zc --steps 100 ${4:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vx',$vx,'zu',$zu,'zc',$zc, d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac'])" || echo "vx $vx zu $zu zc $zc FAILED"
done; done; done
