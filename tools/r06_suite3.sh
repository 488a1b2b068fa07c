#!/bin/bash
# three runs of the whole GPU suite: rare flakes show here, not on the driver's box
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | tail -5; done
