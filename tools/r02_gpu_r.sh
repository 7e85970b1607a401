#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r02_pytest_full.log; cat gpurun_out/r02_pytest_full.log
bash tools/round_profiles.sh r02 2>&1 | tail -30
