#!/bin/bash
# One GPU-box pass: parity tests, accuracy report, bench.  Outputs under gpurun_out/.  A step that times out or
# errors (rc >= 2) stops the chain; plain test failures (rc 1) do not.
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
if [ -n "${FIRST}" ]; then
echo "== first: ${FIRST}"; timeout -k 10 240 python -m pytest ${FIRST} -m gpu -v -x --timeout=120 2>&1 | tee gpurun_out/pytest_first.log | grep -E "PASS|FAIL|ERROR|passed|failed|^E " | tail -n 40
rc=${PIPESTATUS[0]}; [ $rc -ge 1 ] && { echo "first rc $rc"; exit $rc; }
fi
if [ "${SKIP_TESTS}" != "1" ]; then
echo "== pytest -m gpu"; timeout -k 10 ${T_TEST:-800} python -m pytest tests -m gpu -v --timeout=300 ${PYTEST_ARGS} 2>&1 | tee gpurun_out/pytest_gpu.log | grep -E "PASSED|FAILED|ERROR|passed|failed"
rc=${PIPESTATUS[0]}
grep -E "^E  |^FAILED|sections|oracle sub-batch" gpurun_out/pytest_gpu.log | tail -n 40
[ $rc -ge 2 ] && { echo "pytest rc $rc"; exit $rc; }
fi
if [ "${SKIP_ACC}" != "1" ]; then
echo "== accuracy"; timeout -k 10 500 python tools/accuracy_report.py --full 2> gpurun_out/accuracy.err | tee gpurun_out/accuracy.txt || { echo "accuracy failed"; tail -5 gpurun_out/accuracy.err; exit 3; }
fi
if [ "${SKIP_BENCH}" != "1" ]; then
echo "== bench"; timeout -k 10 300 python bench.py ${BENCH_ARGS} > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err || { echo "bench failed"; tail -5 gpurun_out/bench_c3.err; exit 4; }
tail -c 4000 gpurun_out/bench_c3.json
fi
