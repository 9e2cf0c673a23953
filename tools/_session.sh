cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/time_step.py C0 C1 C2 C3 C4 F16 > $O/s21_time.log 2>&1; cat $O/s21_time.log
for c in C2 C3; do echo "== $c"; B9_HIP_LIB=build/variants/lib_gantt_l0.so timeout -k 10 120 python tools/gantt_step.py $c > $O/s21.log 2>&1; sed -n 3,5p $O/s21.log; grep -A2 "heavy role phases" $O/s21.log | tail -2; grep -A2 "inside one heavy" $O/s21.log; done
