import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'maxdiag' in r['Kernel_Name']]
a = idx[3]
t0 = int(rows[a]['Start_Timestamp'])
n = 0
for r in rows[a:a + int(sys.argv[2])]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    nm = r['Kernel_Name'].split('(')[0].replace('ipm::', '').replace('void ', '')[:44]
    print("%8.1f -> %8.1f  dur %6.1f  q%s grid %6s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r['Queue_Id'], r['Grid_Size_X'], nm))
