cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r02k > gpurun_out/collect_r02k.log 2>&1
tail -1 gpurun_out/collect_r02k.log
python bench.py > gpurun_out/bench_r02k_line.json 2> gpurun_out/bench_r02k.err
bash tools/collect_lines.sh r02k > gpurun_out/lines_r02k.txt 2>&1
tail -14 gpurun_out/lines_r02k.txt
bash tools/trace_pass.sh 2>&1 | tail -8
