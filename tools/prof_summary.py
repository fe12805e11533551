"""Per-step kernel breakdown from a rocprofv3 --kernel-trace csv of `bench.py --leg train` (the last
`steps` graph replays are taken as the timed region)."""
import csv, glob, collections, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = max(glob.glob(d + '/**/*kernel_trace.csv', recursive=True), key=__import__('os').path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the timed region: the last `steps` occurrences of the AdamW kernel close a step each
idx = [i for i, r in enumerate(rows) if 'k_adamw' in r['Kernel_Name']]
first = idx[-steps - 1] + 1
sel = rows[first:idx[-1] + 1]
wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / steps / 1e3
c = collections.Counter(); t = collections.Counter()
for r in sel:
    n = r['Kernel_Name']; c[n] += 1; t[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(t.values()) / steps / 1e3
print(f"kernels/step {len(sel)/steps:.0f}  wall {wall:.0f} us/step  kernel sum {tot:.0f} us/step")
small = sum(v for n, v in t.items() if v / c[n] < 12000) / steps / 1e3
print(f"kernels < 12 us: {sum(c[n] for n in c if t[n]/c[n] < 12000)/steps:.0f}/step, {small:.0f} us/step")
for n, v in sorted(t.items(), key=lambda x: -x[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{c[n]/steps:6.1f}/step {v/c[n]/1e3:8.1f} us  {v/steps/1e3:7.1f} us/step  {n[:110]}")
