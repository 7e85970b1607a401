#!/bin/bash
bash /root/repo/tools/r02_gpu_o.sh 2>&1 | grep -v "^W2\|^E2" | head -16
bash /root/repo/tools/r02_gpu_i.sh 2>&1 | grep -E "cfg4 x|nn1|nnb|bbox|fitness|rocprim|vg_|lds_pack|deinter|fillBuffer|copyBuffer"
