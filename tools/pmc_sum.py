import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in d.items():
    if not any(s in k for s in sys.argv[2].split(',')): continue
    print(k)
    for n, v in c.items(): print('   %-28s %14.0f  (n=%d)' % (n, sum(v)/len(v), len(v)))
