#!/bin/bash
# round-3 profile of the bench workload: kernel stats, PMC passes (separate runs), HBM traffic.  Output: gpurun_out/r03_prof[_TAG]
# (kernel name and shape in traffic.json are taken from the bench line of the SAME command)
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r03_prof${1:+_$1}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 $ARGS > $O/stats.log 2>&1
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$tag -o $tag --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $ARGS > $O/$tag.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<PY
import csv, collections, glob, json, datetime
out=[]
for f in glob.glob('$O/stats/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    out += ['%-64s calls=%5s total_ms=%10.3f avg_us=%10.1f pct=%s' % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']) for r in rows[:40]]
out.append('')
KEYS=('p1_kernel','p1v2_kernel','p2_kernel','p2_fast','psi1_kernel','psi2_')
for tag in ('sq1','sq2','fetch','write'):
    for f in glob.glob('$O/%s/*counter_collection.csv' % tag):
        agg=collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:40]
            if not any(x in k for x in KEYS): continue
            agg.setdefault(k,collections.OrderedDict()).setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        for k,v in agg.items():
            out.append('%-40s %s' % (k, ' '.join('%s=%.4g' % (c, sum(x)/len(x)) for c,x in v.items())))
bench=[json.loads(l) for l in open('$O/stats.log') if l.startswith('{')][-1]
out.append('')
out.append('bench line of the profiled command: ' + json.dumps({k: bench[k] for k in ('value','ms_per_step','roofline')}))
open('$O/summary.txt','w').write('\n'.join(out)+'\n')
print('\n'.join(out))
def avg(tag, counter, key):
    vals=[float(r['Counter_Value']) for f in glob.glob('$O/%s/*counter_collection.csv' % tag) for r in csv.DictReader(open(f)) if key in r['Kernel_Name'] and r['Counter_Name']==counter]
    return sum(vals)/len(vals) if vals else None
tr={}
kname=bench['roofline']['kernel']                       # e.g. gp::p2_fast8_kernel<3>
for key,name in ((kname.split('::')[-1].split('<')[0],'p2_kernel'),('p1v2_kernel','p1_kernel'),('psi1_kernel','psi1_kernel')):
    f=avg('fetch','FETCH_SIZE',key); w=avg('write','WRITE_SIZE',key)
    if f is not None and w is not None:
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads -> x2 (MI355X_MICROARCH.md, HBM)
        tr[name+'_hbm_bytes_per_launch']=(2.0*f+w)*1024.0; tr[name+'_fetch_kb_raw']=f; tr[name+'_write_kb_raw']=w
c=bench['config']
tr['kernel']=kname; tr['N']=c['N_per_gpu']; tr['D']=c['D']; tr['M']=c['M']; tr['Q']=c['Q']
tr['date']=datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%MZ')
tr['command']='tools/r03_prof.sh (bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'
json.dump(tr, open('$O/traffic.json','w'), indent=1)
print(tr)
PY
rm -rf $O/*/*kernel_trace.csv $O/*/*agent_info.csv $O/*/*counter_collection.csv
cd $R
