import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
for kn in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for r in rows:
        if kn in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(kn, {k: round(sum(v) / len(v) / 1e6, 3) for k, v in agg.items()})
