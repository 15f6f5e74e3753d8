import re, sys, textwrap
fn, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lines=open(fn).read().split('\n')
seg=lines[a:b]
out=[]
for l in seg:
    m=re.match(r'\s+([a-z_0-9]+)\s*(.*)',l)
    if not m: continue
    op=m.group(1); rest=m.group(2)
    if 'mfma' in op: t='M'
    elif op.startswith('ds_read'): t='r'
    elif op.startswith('ds_write'): t='w'
    elif 'accvgpr' in op: t='a'
    elif op.startswith('global_load'): t='G'
    elif op.startswith('global_store'): t='S'
    elif op=='s_waitcnt': t='['+rest.split(';')[0].strip().replace('vmcnt','v').replace('lgkmcnt','l')+']'
    elif op=='s_barrier': t='|B|'
    elif op.startswith('v_'): t='v'
    elif op.startswith('s_'): t='s'
    else: t='?'
    out.append(t)
print('\n'.join(textwrap.wrap(''.join(out),160)))
