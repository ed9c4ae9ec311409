#!/bin/bash
# HBM traffic of the bench step: separate --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with tracing)
set -e
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-fifo --no-emulate-world > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-fifo --no-emulate-world > $OUT/write.log 2>&1
