cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/time_step.py C2 C3 C4
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/s29_bench.log 2>&1; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/s29_bench.log") if l.startswith("{")][-1])
print("bench value %.3e ms/step %.4f launch_us %.2f accept %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["accept_rate"]))
PY
