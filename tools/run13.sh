#!/bin/bash
set -o pipefail
O=gpurun_out/r03_run13; mkdir -p $O
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== gpu tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
echo "== soak settle"; timeout -k 10 600 python tools/soak_settle.py 400 2 > $O/soak_settle.txt 2>&1; tail -4 $O/soak_settle.txt
echo "== rehearsal N=2 bench on one GPU (gloo)"; RWH_BENCH_REHEARSAL=1 timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > $O/bench_n2.json 2> $O/bench_n2.err; tail -c 1500 $O/bench_n2.json; tail -2 $O/bench_n2.err
