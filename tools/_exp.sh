set -x
timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --workload bert_base_train --no-traffic 2>&1 | grep metric | cut -c1-330
