"""Timeline of the last training step in a rocprofv3 kernel trace (start us, duration us, '||' when it
started before an earlier kernel ended, idle gap before it)."""
import csv, glob, re, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=__import__('os').path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_final_sum_step' in r['Kernel_Name']]      # once per step: a step is shown from its
# predecessor's update (norm, AdamW) to its own gradient-norm launch
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1            # which step from the end (1 = last)
sel = rows[idx[-back - 1] + 1:idx[-back] + 1]
t0 = int(sel[0]['Start_Timestamp'])
def short(n):
    n = re.sub(r'void |at::native::|\(anonymous namespace\)::|_GLOBAL__N_', '', n)
    n = re.sub(r'vectorized_elementwise_kernel<\d, ', 'vec<', n)
    n = re.sub(r'elementwise_kernel_manual_unroll<128, 4, gpu_kernel_impl(_nocast)?<', 'ew<', n)
    return n[:70]
prev_end = 0; idle = 0
for i, r in enumerate(sel):
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    flag = '||' if s < prev_end - 0.5 else '  '
    gap = max(0, s - prev_end); idle += gap
    if len(sys.argv) < 3 or (e - s) > float(sys.argv[2]) or gap > 8:
        print(f"{i:3d} q{r.get('Queue_Id', '?'):>2} {s:7.0f} {e-s:6.1f} {flag} gap{gap:5.1f} {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
print("step", prev_end, "us; idle", idle)
