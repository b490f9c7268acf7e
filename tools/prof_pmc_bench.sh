# PMC counters for the phase kernels of the bench workload (separate passes, kernel-trace only)
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/prof_pmc
mkdir -p $O
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$tag -o $tag --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/$tag.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<PY
import csv, collections, glob
out=[]
for tag in ('sq1','sq2','fetch','write'):
    for f in glob.glob('$O/%s/*counter_collection.csv' % tag):
        rows=list(csv.DictReader(open(f)))
        agg=collections.OrderedDict()
        for r in rows:
            k=r['Kernel_Name'].split('(')[0][:40]
            if not any(x in k for x in ('p1_kernel','p2_kernel','p2_fast','psi1_kernel')): continue
            a=agg.setdefault(k,collections.OrderedDict())
            a.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        for k,v in agg.items():
            out.append('%-40s %s' % (k, ' '.join('%s=%.4g' % (c, sum(x)/len(x)) for c,x in v.items())))
open('$O/summary.txt','w').write('\n'.join(out)+'\n')
print('\n'.join(out))
PY
rm -rf $O/*/*kernel_trace.csv
python3 - <<PY
import csv, glob, json
def avg(tag, counter, key):
    vals=[]
    for f in glob.glob('$O/%s/*counter_collection.csv' % tag):
        for r in csv.DictReader(open(f)):
            if key in r['Kernel_Name'] and r['Counter_Name']==counter: vals.append(float(r['Counter_Value']))
    return sum(vals)/len(vals) if vals else None
out={}
for key,name in (('p2_fast_kernel','p2_kernel'),('p1_kernel','p1_kernel'),('psi1_kernel','psi1_kernel')):
    f=avg('fetch','FETCH_SIZE',key); w=avg('write','WRITE_SIZE',key)
    if f is not None and w is not None:
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads -> x2
        out[name+'_hbm_bytes_per_launch']=(2.0*f+w)*1024.0
        out[name+'_fetch_kb_raw']=f; out[name+'_write_kb_raw']=w
json.dump(out, open('$O/traffic.json','w'), indent=1)
print(out)
PY
