# GPU box: bench.py under unusual but legal argument combinations; prints value / ms per step for each
for args in "--steps 1 --warmup 0" "--steps 2 --warmup 0 --pipeline-depth 1" "--steps 20 --warmup 5" "--steps 3 --warmup 1 --max-batch 8"; do
  out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs $args 2>/dev/null | tail -1)
  python - "$args" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); print(f"{sys.argv[1]:45s} -> {d['value']:8.1f} xRT  {d['ms_per_step']:7.2f} ms/step  steps={d['steps']} warmup={d['warmup']}")
except Exception as e:
    print(f"{sys.argv[1]:45s} -> FAILED ({e}): {sys.argv[2][:200]}")
PY
done
