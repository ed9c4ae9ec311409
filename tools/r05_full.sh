#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05/gputests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/gputests.txt
tail -5 gpurun_out/r05/gputests.txt
timeout -k 10 600 python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err; echo "bench rc=$?"
cat gpurun_out/r05/bench.json
