#!/bin/bash
# The whole GPU suite under every diagnostic / A-B switch (each must pass: the switches select other kernels, not other answers).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/switch_matrix; rm -rf $O; mkdir -p $O
run() { name=$1; shift; (env "$@" timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $O/$name.log; echo "$name: $(tail -1 $O/$name.log | cut -c1-120)"; }
run default LSR_NOP=1
run nn_coop0 LSR_NN_COOP=0
run gicp_fused0 LSR_GICP_FUSED=0
run gicp_ball0 LSR_GICP_BALL=0
run gicp_ball_cells3 LSR_GICP_BALL_CELLS=3
run gicp_ball_cells8 LSR_GICP_BALL_CELLS=8
run ndt_quad0 LSR_NDT_QUAD=0
run wait_sleep LSR_WAIT_MODE=sleep
run wait_yield LSR_WAIT_MODE=yield
run nn_prefetch0 LSR_NN_PREFETCH=0
run nn_from_grid0 LSR_NN_FROM_GRID=0
run nn_fine_rings1 LSR_NN_FINE_RINGS=1
run fit_group_quad LSR_FIT_GROUP_FORM=1
run table_dense LSR_NDT_TABLE_MODE=0
run table_tile LSR_NDT_TABLE_MODE=3
run ndt_lane512 LSR_NDT_QUAD=0 LSR_NDT_WORKGROUP=512
run ndt_widen0 LSR_NDT_WIDEN=0
