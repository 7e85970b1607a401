cd $GRAFT_REPO_ROOT; export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 900 python -m pytest tests/test_frontend_stream_gpu.py tests/test_voxelgrid_gpu.py tests/test_host_cpu.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_check.json 2> gpurun_out/r05_bench_check.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_check.json'))
f=d['frontend_stream']
print({k:f[k] for k in f if k not in ('what',)})
PY
