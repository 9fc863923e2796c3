set -u
for o in 3 0 1; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --order $o > gpurun_out/b_$o.json 2>gpurun_out/b_$o.err || { tail -3 gpurun_out/b_$o.err; exit 1; }
  python - $o <<'PY'
import json,sys
d=json.loads(open("gpurun_out/b_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("order", sys.argv[1], "S0 kernel", d["roofline"]["kernel_ms"], "orbit", d["orbit_views"]["kernel_ms_mean"], d["orbit_views"]["kernel_ms_per_view"])
PY
done
