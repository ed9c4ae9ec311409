#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "g4p or lnfold or geglu or sqp" > gpurun_out/r05/verify_tests.txt 2>&1; tail -3 gpurun_out/r05/verify_tests.txt
python3 bench.py --no-cpu-baseline --no-emulate-world > gpurun_out/r05/verify_bench.json 2> gpurun_out/r05/verify_bench.err; echo "bench rc=$?"; tail -c 900 gpurun_out/r05/verify_bench.json; tail -3 gpurun_out/r05/verify_bench.err
