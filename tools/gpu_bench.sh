cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline --no-scale > gpurun_out/bench_quick_$i.json 2>/dev/null; python3 -c "
import json
d=json.load(open('gpurun_out/bench_quick_$i.json'))
r=d['roofline']
print('value %.3fM ms/step %.4f median %.4f min %.4f | score %.4f dense %.4f tail %.3f | pcie %.4f' % (d['value']/1e6, d['ms_per_step'], d['timing']['ms_per_step_median'], d['timing']['ms_per_step_min'], r['avg_launch_ms'], r['all_kernels']['k_dense_fused_bf16x3']['avg_launch_ms'], r['tail_frac'], d['pcie_inclusive']['ms_per_step_median']))
"; done
timeout 300 python bench.py --no-cpu-baseline --no-scale --profile-every 1000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('uninstrumented timed region: value %.3fM ms/step %.4f' % (d['value']/1e6, d['ms_per_step']))"
