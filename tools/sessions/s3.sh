#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_resident.py -x -q -m gpu > $O/resident.log 2>&1
tail -15 $O/resident.log
timeout 600 python -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_parity.py -x -q -m gpu -k "sweep or cfg5 or lab_frame or lindblad" > $O/other.log 2>&1
tail -5 $O/other.log
timeout 300 python tools/bench_sweep_sizes.py > $O/sweep_sizes.txt 2>&1; cat $O/sweep_sizes.txt
