cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | tail -6
