ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/psu; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
MMN_BENCH_PS_UNIFORM=1 python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --workload c5 --steps 40 --warmup 10 2>/dev/null | python3 $ROOT/tools/show_bench_line.py
MMN_BENCH_PS_UNIFORM=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o u -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --workload c5 --no-graph --steps 20 --warmup 5 > /dev/null 2>&1
python3 - <<'P'
import csv,glob,collections
f=glob.glob('/tmp/../'+__import__('os').environ['GRAFT_REPO_ROOT']+'/gpurun_out/psu/f/**/u_counter_collection.csv',recursive=True) or glob.glob(__import__('os').environ['GRAFT_REPO_ROOT']+'/gpurun_out/psu/f/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r['Counter_Name']=='FETCH_SIZE': acc[r['Kernel_Name'].split('(')[0][-30:]].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, len(v), 'FETCH KB/launch', sum(v)/len(v))
P
rm -rf $OUT
