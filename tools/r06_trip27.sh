#!/bin/bash
for i in 1 2 3; do python -m pytest tests/test_gpu_dense_unet.py -m gpu -q -k "bitwise_invisible" 2>&1 | grep -E "Error|passed|failed|assert" | cut -c1-300; done
