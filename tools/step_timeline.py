"""One replayed step out of a rocprofv3 --kernel-trace CSV as a timeline: offset, duration, idle gap before each kernel.
usage: step_timeline.py kernel_trace.csv [which_step_from_the_end=2]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marks = [i for i, r in enumerate(rows) if 'grid_project_kernel' in r['Kernel_Name']]
a, b = marks[-back - 1], marks[-back]
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n[:70]
t0 = int(rows[a]['Start_Timestamp']); prev_end = None; busy = 0; gaps = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if prev_end is None else s - prev_end
    print("%9.1f  dur %7.1f  gap %6.1f  %-70s grid=%s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short(r['Kernel_Name']), r['Grid_Size_X']))
    busy += e - s; gaps += max(gap, 0); prev_end = max(e, prev_end or e)
print("kernels %d  busy %.1f us  gaps %.1f us  span %.1f us" % (b - a, busy / 1e3, gaps / 1e3, (prev_end - t0) / 1e3))
