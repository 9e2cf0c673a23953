cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 5 90 python tools/block_fixed_cost.py 2>&1 | grep -v "b9 sampler"
for r in 1 2 3; do timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/s41.log 2>&1; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/s41.log") if l.startswith("{")][-1])
print("bench steps %d value %.3e ms/step %.4f launch_us %.2f ratio %.3f" % (d["steps"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["timed_region_breakdown"]["ms_per_step_over_launch_period"]))
PY
done
