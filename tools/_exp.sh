set -e
mkdir -p gpurun_out/e6
rm -f gpurun_out/e6/res.log
run() { tag=$1; shift; env "$@" timeout -k 10 120 python3 bench.py --no-roofline --no-cpu-baseline --steps 300 --warmup 20 2>gpurun_out/e6/$tag.err | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$tag /" >> gpurun_out/e6/res.log; }
run new A=1
run new2 A=1
cat gpurun_out/e6/res.log
timeout -k 10 900 python3 -u -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/e6/tests.log 2>&1; tail -5 gpurun_out/e6/tests.log
