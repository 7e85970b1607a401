cd $GRAFT_REPO_ROOT; export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1200 python -m pytest tests/test_multigpu_gpu.py -x -q -m gpu 2>&1 | tail -12
