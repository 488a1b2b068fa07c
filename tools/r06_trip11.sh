#!/bin/bash
python -m pytest tests/test_gpu_fused_head.py -m gpu -q -x 2>&1 | tail -30
