import csv,collections,glob,re,sys
d=sys.argv[1]; skip_eager=int(sys.argv[2]) if len(sys.argv)>2 else 0
f=glob.glob(f'{d}/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'blend_backward' in r['Kernel_Name']]
# graph-replay window: bench = enable_graph(3 eager + 2 cap + capture) + 5 warm + 10 timed (graph) + 10 eager instrumented
# take blend_bwd launches [a:b) that belong to the timed graph region
a=int(sys.argv[3]) if len(sys.argv)>3 else 0; b=int(sys.argv[4]) if len(sys.argv)>4 else len(idx)
sub=rows[idx[a]:idx[b-1]+1]; nsteps=(b-a)/2
t0=int(sub[0]['Start_Timestamp']); t1=int(sub[-1]['End_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in sub)
print('window ms/step', (t1-t0)/1e6/nsteps, 'sum-of-kernels ms/step', busy/1e6/nsteps, 'kernels/step', len(sub)/nsteps)
cat=collections.defaultdict(lambda:[0,0.0])
def c(n):
    if 'instag' in n:
        m=re.search(r'(\w+_kernel)',n); return 'instag:'+(m.group(1) if m else n[:30])
    if n.startswith('Cijk'): return 'hipblaslt gemm'
    if 'miopen' in n.lower() or 'naive_conv' in n or 'Conv' in n or 'igemm' in n or 'transpose' in n: return 'miopen conv'
    if 'rocprim' in n: return 'rocprim'
    if 'multi_tensor' in n: return 'fused adam/foreach'
    if 'rocclr' in n: return 'memset/copy'
    if 'at::native' in n: return 'aten elementwise/reduce'
    return 'other:'+n[:50]
for r in sub:
    k=c(r['Kernel_Name']); cat[k][0]+=1; cat[k][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for k,v in sorted(cat.items(), key=lambda kv:-kv[1][1])[:22]:
    print(f"{k:50s} calls/step {v[0]/nsteps:7.1f}  ms/step {v[1]/nsteps:7.3f}")
