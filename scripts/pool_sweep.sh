mkdir -p gpurun_out/r04w
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stage or intermediate or n64 or f16x3 or batch" > gpurun_out/r04w/pytest.log 2>&1; tail -3 gpurun_out/r04w/pytest.log
for R in 0 22 11 4 2 1; do
  echo "== SE3TN_POOL_ROWS=$R"
  SE3TN_POOL_ROWS=$R timeout 120 bash scripts/ktrace.sh r04_pool$R 2>&1 | grep -i "maxpool"
  python -c "
import json;d=json.loads(open('gpurun_out/ktrace_r04_pool$R/bench.json').read().strip().splitlines()[-1]);print(d['single_stream'])"
done
