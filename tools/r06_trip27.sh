#!/bin/bash
for i in 1 2 3 4; do python -m pytest tests/test_gpu_grad_overlap.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | head -12; done
