# One decode step of a `rocprofv3 --kernel-trace` CSV as a timeline: kernel, hardware queue, start, end (us), grid, workgroup, LDS, VGPRs; and the
# per-kernel totals of that step.  Usage: python3 profiles/timeline.py <trace dir> [shortest kernel shown, us]
# timeline of the last decode step in a rocprofv3 kernel trace: kernel, queue, start, end; and per-kernel sums of that step
import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'copyBuffer' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps are separated by gaps: find step starts as first 'zstd_entropy'/'decompress'/... after the summary kernel
ends = [i for i, r in enumerate(rows) if 'summary_to_host' in r['Kernel_Name']]
# group kernels by step: a step ends at the LAST summary_to_host within a burst; use time gaps > 300us w/o kernels? simpler: split at gaps
steps = []
cur = [rows[0]]
mx_end = int(rows[0]['End_Timestamp'])
for r in rows[1:]:
    if int(r['Start_Timestamp']) - mx_end > 150000:  # 150 us of nothing running
        steps.append(cur); cur = []
    cur.append(r); mx_end = max(mx_end, int(r['End_Timestamp']))
steps.append(cur)
big = [s for s in steps if (max(int(r['End_Timestamp']) for r in s) - int(s[0]['Start_Timestamp'])) > 0.5 * max((max(int(r['End_Timestamp']) for r in t) - int(t[0]['Start_Timestamp'])) for t in steps)]
s = big[-1]
t0 = int(s[0]['Start_Timestamp'])
print("step span %.1f us, %d kernels; %d big steps" % ((max(int(r['End_Timestamp']) for r in s) - t0) / 1e3, len(s), len(big)))
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
for r in s:
    du = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if du >= minus:
        print("%-36s q%-2s %9.1f -> %9.1f (%8.1f) grid %8s wg %4s lds %6s vgpr %s" % (r['Kernel_Name'].split('(')[0][-36:], r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, du, r['Grid_Size_X'], r['Workgroup_Size_X'], r.get('LDS_Block_Size'), r.get('VGPR_Count')))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in s:
    k = r['Kernel_Name'].split('(')[0][-40:]
    acc[k][0] += 1; acc[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("--- per kernel (this step)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-42s %4d %10.1f us" % (k, v[0], v[1]))
