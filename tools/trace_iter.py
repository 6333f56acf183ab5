"""Print the kernel timeline of one iteration from a rocprofv3 --kernel-trace CSV: python tools/trace_iter.py DIR [N [ANCHOR]]
(anchored at the Nth scaling_kernel -- the first kernel of an iteration on dense handles -- or prepare_kernel)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
anchor = sys.argv[3] if len(sys.argv) > 3 else ('scaling_kernel' if any('scaling_kernel' in r['Kernel_Name'] for r in rows) else 'prepare_kernel')
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = idx[nth], idx[nth + 1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    nm = r['Kernel_Name'].split('(')[0].replace('ipm::', '').replace('void ', '')[:48]
    print("%8.1f -> %8.1f  dur %7.1f  q%-3s grid %7s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r['Queue_Id'], r['Grid_Size_X'], nm))
