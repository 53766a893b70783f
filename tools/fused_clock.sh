#!/bin/bash
# tools/fused_clock.sh -- per-phase time of the fused solve kernel (ALTRO_HIP_FUSED_CLOCK) on the C2 / C3 batches, plus the
# fused / sequenced wall times of tools/solve_ab.py; output: gpurun_out/fused_clock.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/fused_clock.txt
echo "# python tools/solve_ab.py 7   (MI355X; wall time of altro_hip_ilqr_solve, fused / launch-sequenced alternating in one process)" > $out
timeout 300 python tools/solve_ab.py 7 2>/dev/null | grep -v amdgpu.ids >> $out
echo >> $out
echo "# ALTRO_HIP_FUSED_CLOCK=1 python -u tools/solve_ab.py 1   (two fused solves per case, then the case's line; 100 MHz wall_clock64 per phase)" >> $out
ALTRO_HIP_FUSED_CLOCK=1 timeout 300 python -u tools/solve_ab.py 1 2>&1 | grep -v "amdgpu.ids\|sequenced" >> $out
