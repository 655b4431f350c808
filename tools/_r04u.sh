#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04u
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python tests/soak_wave_vs_lane.py --minutes 6 --seed 404 2>/dev/null | tail -3 | cut -c1-1500 | tee gpurun_out/r04u/soak_wave_vs_lane_long.json
