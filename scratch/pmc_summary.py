import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-72:]][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in agg.items():
    if 'lr_' not in k:
        continue
    res[k] = {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()}
    print(k, {c: round(x["mean"], 1) for c, x in res[k].items()})
json.dump(res, open(out + '/pmc_summary.json', 'w'), indent=1)
