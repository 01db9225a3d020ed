import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    print(d['config'].get('workload','')[:40], round(d['ms_per_step']*1000,2), {k:round(v,2) for k,v in d['roofline']['avg_launch_us'].items()})
