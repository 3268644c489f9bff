#!/bin/bash
# round 6, GPU call 1: the poison hunt (tools/poison_hunt.py) + the GPU suite under the poisoned-allocator fixture, then with the engine's own poison on
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "1 1" "2 0" "1 2"; do
  timeout 600 python tools/poison_hunt.py $cfg > gpurun_out/r6_hunt_$(echo $cfg | tr ' ' _).txt 2>&1
  echo "hunt $cfg rc $?"; grep "\[hunt\]" gpurun_out/r6_hunt_$(echo $cfg | tr ' ' _).txt | tail -12
done
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/r6_suite_fixture.txt 2>&1
echo "suite(fixture) rc $?"; tail -5 gpurun_out/r6_suite_fixture.txt
HN_POISON_WS=1 timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/r6_suite_engine_poison.txt 2>&1
echo "suite(engine poison) rc $?"; tail -5 gpurun_out/r6_suite_engine_poison.txt
grep -n "FAILED\|^ERROR" gpurun_out/r6_suite_fixture.txt gpurun_out/r6_suite_engine_poison.txt | head -40
