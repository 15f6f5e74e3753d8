cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu3.txt 2>&1; tail -5 gpurun_out/pytest_gpu3.txt
timeout 900 python bench.py > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; tail -2 gpurun_out/bench_r02b.err; cut -c1-400 gpurun_out/bench_r02b.json
# two ranks sharing the one GPU over gloo: the entity-sharded ranks must not depend on the world size (bench asserts it)
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r02b_gloo2.json 2> gpurun_out/bench_r02b_gloo2.err
echo "gloo2 rc=$?"; tail -3 gpurun_out/bench_r02b_gloo2.err; python3 -c "
import json
d=json.loads(open('gpurun_out/bench_r02b_gloo2.json').read().strip().splitlines()[-1])
print('world 2 (gloo, one GPU): scale ranks_independent_of_world =', d['scale'].get('ranks_independent_of_world'), d['scale']['ranks_sha1'], d['scale']['mean_rank'])
"
