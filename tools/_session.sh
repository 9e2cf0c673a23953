cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_host.py -m gpu -x -q 2>&1 | grep -B30 "AssertionError" | tail -40
