cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2; done
bash tools/collect_profiles.sh r02j > gpurun_out/collect_r02j.log 2>&1
tail -2 gpurun_out/collect_r02j.log
python bench.py > gpurun_out/bench_r02j_line.json 2> gpurun_out/bench_r02j.err
bash tools/collect_lines.sh r02j > gpurun_out/lines_r02j.txt 2>&1
tail -14 gpurun_out/lines_r02j.txt
