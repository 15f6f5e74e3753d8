#!/usr/bin/env python3
"""MFMA-heavy loops of a kernel in a hipcc -S listing: tools/isa_loops.py file.s <kernel-name-substring> [min_mfma]
prints (first line, last line, MFMAs, vector ops, LDS reads, global loads, barriers, s_nops) per back-edge; feed a range to isa_trace.py."""
import re, sys
fn, key = sys.argv[1], sys.argv[2]
min_m = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(fn).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for si, a in enumerate(starts):
    if key not in lines[a]:
        continue
    b = starts[si + 1] if si + 1 < len(starts) else len(lines)
    print(lines[a][:110])
    labels = {}
    for i in range(a, b):
        m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
        if m: labels[m.group(1)] = i
    for i in range(a, b):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', lines[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            j = labels[m.group(1)]
            seg = lines[j:i]
            nm = sum('v_mfma' in l for l in seg)
            if nm >= min_m:
                print("  lines %d-%d: mfma %d, valu %d, ds_read %d, ds_write %d, global_load %d, barrier %d, s_nop %d, accvgpr %d" % (
                    j, i, nm, sum(l.strip().startswith('v_') and 'mfma' not in l for l in seg), sum('ds_read' in l for l in seg),
                    sum('ds_write' in l for l in seg), sum('global_load' in l for l in seg), sum('s_barrier' in l for l in seg),
                    sum('s_nop' in l for l in seg), sum('accvgpr' in l for l in seg)))
