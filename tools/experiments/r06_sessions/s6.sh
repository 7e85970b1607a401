#!/bin/bash
# CU-masked side stream for the share of 8 candidates (slowest share: FIRST=24; fastest: FIRST=0)
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
for F in 24 0; do
echo "== FIRST=$F baseline"; FIRST=$F python tools/share_probe.py 2>&1 | tail -1 | cut -c1-260
for K in 32 64 96 128; do for G in 1 4 12; do
echo "== FIRST=$F LSR_SIDE_CUS=$K LSR_FIT_GROUP_MIN=$G"; LSR_SIDE_CUS=$K LSR_FIT_GROUP_MIN=$G FIRST=$F python tools/share_probe.py 2>&1 | tail -1 | cut -c1-260
done; done
for G in 1 4; do echo "== FIRST=$F unmasked LSR_FIT_GROUP_MIN=$G"; LSR_FIT_GROUP_MIN=$G FIRST=$F python tools/share_probe.py 2>&1 | tail -1 | cut -c1-260; done
done
