set -x
python -m pytest tests/test_nn_gpu.py tests/test_gicp_gpu.py tests/test_loop_closure_gpu.py -x -q 2>&1 | tail -3
python tools/gicp_probe.py 2>&1 | tail -2
python tools/gicp_probe.py 2>&1 | tail -1
timeout 300 python tools/cfg4_stage_probe.py 2>&1 | tail -2
for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1 | cut -c1-260; done
for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1 | cut -c1-260; done
