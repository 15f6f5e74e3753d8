cd $GRAFT_REPO_ROOT
for v in default tl8 tl2 default tl8; do
if [ $v = default ]; then unset COPER_HIP_LIB; else export COPER_HIP_LIB=$PWD/build/ab/lib_$v.so; fi
timeout 300 python bench.py --no-cpu-baseline --no-scale --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: value %.3fM ms/step %.4f tail %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['tail_frac']))"
done
