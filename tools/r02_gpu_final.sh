#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r02_pytest_full.log; cat gpurun_out/r02_pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/round_profiles.sh r02 2>&1 | grep -v "^-rw\|^total\|^drwx" | tail -14
