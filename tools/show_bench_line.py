"""Print step time and per-kernel times of bench.py JSON lines (files named on the command line, else stdin)."""
import json, sys
src = [l for f in sys.argv[1:] for l in open(f)] if len(sys.argv) > 1 else sys.stdin
for line in src:
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    print(d['config'].get('workload', '')[:40], round(d['ms_per_step'] * 1000, 2), {k: round(v, 2) for k, v in d['roofline']['avg_launch_us'].items()})
