cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_synth10m.py -x -q > gpurun_out/pytest_gpu4.txt 2>&1; tail -5 gpurun_out/pytest_gpu4.txt
bash tools/gpu_bench3.sh
COPER_TAIL_UNFUSED=1 timeout 300 python bench.py --no-cpu-baseline --no-scale --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused tail: value %.3fM ms/step %.4f tail %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['tail_frac']))"
