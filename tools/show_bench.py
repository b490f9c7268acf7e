import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["device_ms"])
e=d["extra"][0]; print(e["ms_per_eval_wall"], e["ms_per_eval_wall_blocks"], e["device_ms"], e["host_enqueue_share"], e["frac"])
for e in d["extra"][1:]: print({k:v for k,v in e.items() if k in ("ms_per_eval","frac","fraction","frac_WB")})
