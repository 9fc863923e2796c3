import csv, glob, sys, collections, os
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)):
    print('==', f)
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ('Name','Calls','TotalDurationNs','AverageNs','Percentage','MinNs','MaxNs')})
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'][:100]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in agg.items():
    print('--', k)
    for c, v in sorted(cs.items()):
        print('   %-28s n=%d mean=%.6g' % (c, len(v), sum(v)/len(v)))
