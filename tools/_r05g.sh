#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 800 python tests/soak_parity.py --minutes 11 --seed 91 2>/dev/null | tail -1 | cut -c1-1500 | tee $O/soak_parity.json
timeout 800 python tests/soak_wave_vs_lane.py --minutes 11 --seed 707 2>/dev/null | tail -1 | cut -c1-1500 | tee $O/soak_wave_vs_lane.json
