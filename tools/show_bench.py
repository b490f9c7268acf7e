"""One-screen summary of a bench.py JSON line: headline, device times, configs[1]'s steady-state wall clock, the fractions of the free-embedding extras.
usage: python tools/show_bench.py gpurun_out/r05/bench_c.json"""
import json
import sys

text = open(sys.argv[1]).read().strip()
try:
    d = json.loads(text)                         # a pretty-printed copy under profiles/
except ValueError:
    d = json.loads(text.splitlines()[-1])        # bench.py's own output: the last line
print('%.2f evaluations/s  %.3f ms per step  roofline %.3f  device %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('device_ms')))
for e in d.get('extra', []):
    if 'ms_per_eval_wall' in e:
        print('%s: wall %.4f ms (blocks %s)  device %.4f ms  host share %.3f  frac %.3f' % (
            e['workload'][:24], e['ms_per_eval_wall'], e.get('ms_per_eval_wall_blocks'), e['device_ms'], e['host_enqueue_share'], e['frac']))
    else:
        print('%s: %s' % (e['workload'][:70], {k: v for k, v in e.items() if k in ('ms_per_eval', 'frac')}))
if 'cpu_baseline' in d:
    print('cpu baseline %.3f evaluations/s on %d logical CPUs (%s)' % (d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind']))
