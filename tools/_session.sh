set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/s11_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/s11_tests.log
timeout -k 10 300 python tools/time_step.py C0 C1 C2 C3 C4 F16 F16P2 F4 > $O/s11_time_default.log 2>&1 || echo "time default failed"
for r in 1 2; do timeout -k 10 120 python tools/time_marg.py 50000 4 4 8 >> $O/s11_marg.log 2>&1 || echo fail; done
B9_HIP_LIB=build/variants/lib_gantt.so timeout -k 10 120 python tools/gantt_step.py C2 > $O/s11_gantt_C2.log 2>&1 || echo "gantt failed"
tail -3 $O/s11_tests.log; cat $O/s11_time_default.log $O/s11_marg.log; sed -n 1,6p $O/s11_gantt_C2.log
