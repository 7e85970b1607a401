#!/bin/bash
# filtered-source entries without the final stream synchronisation: frontend tests, probes A/B, the bench's frontend + N1 legs
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
O=gpurun_out/s11.txt; : > $O
timeout 900 python -m pytest -x -q -m gpu tests/test_voxelgrid_gpu.py tests/test_frontend_stream_gpu.py tests/test_loop_closure_gpu.py tests/test_host_cpu.py tests/test_edge_cases_gpu.py 2>&1 | tail -4 >> $O
for v in 0 1; do LSR_SOURCE_SYNC=$v timeout 300 python tools/preprocess_probe.py 2>&1 | tail -1 >> $O; done
for v in 0 1; do echo "[LSR_SOURCE_SYNC=$v]" >> $O; LSR_SOURCE_SYNC=$v timeout 900 python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d['frontend_stream']; print('frontend', f['scan_in_to_pose_out'], 'host', f['scan_in_to_pose_out_host_payload_pcie_inclusive']['median_ms'], 'map', f['map_update_ms'], 'amort', f['ms_per_scan_with_map_update_amortised'], f['parity_vs_cpu_over_the_stream'])
print('n1', d['next_rows']['source_preprocess'])
print('value', d['value'])
" >> $O 2>&1; done
cat $O
