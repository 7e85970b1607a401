# kernel trace of the fitness stage alone (MODE=stages: the four C entries one after the other) for the slowest share
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_share
FIRST=24 MODE=${MODE:-stages} REPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_share -o t -- python $REPO/tools/share_probe.py > /dev/null 2>&1
python $REPO/tools/timeline.py /tmp/tr_share 300 2000 | grep -v "ndt_eval_lane" | tail -40
