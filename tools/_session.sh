cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_sampler.py -m gpu -x -q > $O/s13_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/s13_tests.log
B9_SAMPLER_TRACE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/s13_bench20.log 2>&1; echo "rc=$?"
grep "b9 sampler" $O/s13_bench20.log | tail -12
