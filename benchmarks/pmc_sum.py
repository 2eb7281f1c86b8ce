import csv, sys, glob, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name'][:40]
            agg[kn][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(kn, r['Counter_Name'])] += 1
        for kn, c in agg.items():
            if 'split' in kn and 'scan' in kn:
                print(kn, {k: round(v / cnt[(kn, k)]) for k, v in c.items()})
