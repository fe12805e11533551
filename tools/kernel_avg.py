"""Average duration of the kernels whose name contains one of the given substrings, from the newest rocprofv3
kernel-trace csv under a directory.  usage: kernel_avg.py <dir> name [name ...]"""
import csv, glob, os, sys, collections
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    for n in sys.argv[2:]:
        if n in r['Kernel_Name']:
            acc[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in acc.items():
    v = v[len(v) // 2:]          # second half: past warm-up
    print(f"{n:28s} calls {len(v):5d}  avg {sum(v)/len(v):7.1f} us  min {min(v):7.1f}")
