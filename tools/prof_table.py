import json,sys
d=json.load(open(sys.argv[1]))
n=int(sys.argv[2]) if len(sys.argv)>2 else 45
rows=sorted(d['per_tag'].items(), key=lambda kv:-kv[1]['total_ms'])
tot=sum(v['total_ms'] for k,v in rows)
print('total kernel ms',tot, 'wall s', d['seconds'])
for k,v in rows[:n]:
    print('%-58s %8.2f ms %5.1f%%  n=%4d  %7.1f TF/s %7.1f GB/s'%(k,v['total_ms'],100*v['total_ms']/tot,v['launches'],v['flops']/v['total_ms']/1e9 if v['total_ms'] else 0, v['bytes']/v['total_ms']/1e6 if v['total_ms'] else 0))
print('--- per kernel')
for k,v in sorted(d['per_kernel'].items(), key=lambda kv:-kv[1]['total_ms'])[:12]:
    print('%-40s %8.2f ms %5.1f%% n=%5d %7.1f TF/s'%(k,v['total_ms'],100*v['total_ms']/tot,v['launches'],v['flops']/v['total_ms']/1e9 if v['total_ms'] else 0))
