#!/bin/bash
# bench.py lines of the other BASELINE configs + training-step timings (run on the GPU box); usage: tools/collect_lines.sh <tag>
TAG=$1
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
X="--no-cpu-baseline --no-scale"
timeout 600 python bench.py $X --workload wn18rr_cpg > gpurun_out/bench_${TAG}_wn.json 2>/dev/null
timeout 600 python bench.py $X --workload fb15k237_plain > gpurun_out/bench_${TAG}_plain.json 2>/dev/null
timeout 600 python bench.py $X --workload nations_cpg > gpurun_out/bench_${TAG}_nations.json 2>/dev/null
timeout 600 python bench.py $X --score-mode f32 > gpurun_out/bench_${TAG}_f32.json 2>/dev/null
timeout 900 python bench.py $X --no-extras --workload synth10m_cpg --mode entity --steps 5 --warmup 2 > gpurun_out/bench_${TAG}_10m.json 2>/dev/null
timeout 900 python bench.py $X --no-extras --workload synth10m_cpg --mode entity --topk 10 --steps 5 --warmup 2 > gpurun_out/bench_${TAG}_10m_top10.json 2>/dev/null
timeout 900 python bench.py $X --no-extras --workload synth10m_cpg --mode entity --queries 128 --steps 5 --warmup 2 > gpurun_out/bench_${TAG}_10m_q128.json 2>/dev/null
for w in fb15k237_cpg wn18rr_cpg fb15k237_plain; do timeout 300 python tools/bench_train.py $w 2>/dev/null | tail -1; done > gpurun_out/bench_${TAG}_train.json
timeout 600 python tools/bench_api.py > gpurun_out/bench_${TAG}_api.txt 2>&1
for f in wn plain nations f32 10m 10m_top10 10m_q128; do python3 -c "
import json
d=json.load(open('gpurun_out/bench_${TAG}_$f.json')); r=d.get('roofline',{})
print('$f: %.4gM triples/s, %.4f ms/step; %s frac %.3f (%.4f ms)%s' % (d['value']/1e6, d['ms_per_step'], r.get('kernel'), r.get('frac',0), r.get('avg_launch_ms',0), ''.join('; %s %.3f (%.4f ms)' % (k, v['frac'], v['avg_launch_ms']) for k, v in r.get('all_kernels',{}).items())))
"; done
cat gpurun_out/bench_${TAG}_train.json; tail -3 gpurun_out/bench_${TAG}_api.txt
