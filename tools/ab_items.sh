cd ${GRAFT_REPO_ROOT:-/root/repo}
X="--no-cpu-baseline --no-extras --workload synth10m_cpg --mode entity --steps 5 --warmup 2"
for R in 0 16 32 64; do
  for T in 0 10; do
    COPER_SC3_ITEM_ROWS=$R timeout 600 python bench.py $X --topk $T > gpurun_out/ab_items_${R}_${T}.json 2>/dev/null
    python3 -c "
import json
d=json.load(open('gpurun_out/ab_items_${R}_${T}.json')); r=d.get('roofline',{})
print('item_rows $R topk $T: %.4f ms/step; count %.4f ms; sha %s' % (d['ms_per_step'], r.get('avg_launch_ms',0), d.get('ranks_sha1', d.get('config',{}).get('ranks_sha1'))))
"
  done
done
