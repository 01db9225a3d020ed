import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names=[r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:24] for r in rows]
# find a place where fb9,wgrad,reduce,adam_acc,fb9 repeats
idx=None
for i in range(len(rows)-12, 12, -1):
    if names[i].startswith('k_fb9') and names[i+1].startswith('k_wgrad') and names[i+2].startswith('k_reduce') and names[i+3].startswith('k_adam_acc') and names[i+4].startswith('k_fb9') and names[i-1].startswith('k_adam_acc') and names[i-4].startswith('k_fb9'):
        idx=i; break
print('idx', idx, 'of', len(rows))
t0=int(rows[idx-8]["Start_Timestamp"]); pe=None
for r,n in zip(rows[idx-8:idx+12], names[idx-8:idx+12]):
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{(s-t0)/1000:9.1f} {(e-s)/1000:7.1f} {'' if pe is None else f'{(s-pe)/1000:6.1f}':>7} {n}")
    pe=e
