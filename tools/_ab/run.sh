python -m pytest tests/test_gpu_raster.py tests/test_gpu_losses.py -m gpu -x -q 2>&1 | tail -1
for r in 1 2; do for v in a b; do cp tools/_ab/lib_$v.so dimo_amd/csrc/libdimo_hip.so; echo "== $v"; DIMO_EXEC_STREAMS=0 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_launch']; print('serial', round(d['value']), 'bwd', round(k['blend_bwd'],4))"; python tools/ab_steps.py 2 40 | tail -1; done; done
