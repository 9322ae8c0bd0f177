python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -3
