#!/bin/bash
# flakiness check: the whole GPU suite twice on one box
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for i in 1 2; do timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -10 | cut -c1-400; done
