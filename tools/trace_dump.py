"""Dump the backward part of the last full step of a rocprofv3 kernel trace: main-queue kernels with duration, gap and
what runs on the other queue meanwhile.  usage: python tools/trace_dump.py <rocprof dir> [t_from_ms] [t_to_ms]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)
t_from = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t_to = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step = rows[adam[-2] + 1:adam[-1] + 1]
t0 = step[0]['s']
nm = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mednet::', '').replace('_ZN6mednet', '')[:34]
q1 = [r for r in step if r['Queue_Id'] == step[0]['Queue_Id']]
q2 = [r for r in step if r['Queue_Id'] != step[0]['Queue_Id']]
print('step %.2f ms; main queue %d kernels busy %.2f ms; side queue %d kernels busy %.2f ms' % (
    (step[-1]['e'] - t0) / 1e6, len(q1), sum(r['e'] - r['s'] for r in q1) / 1e6, len(q2), sum(r['e'] - r['s'] for r in q2) / 1e6))
for i, r in enumerate(q1):
    t = (r['s'] - t0) / 1e6
    if t < t_from or t > t_to:
        continue
    gap = (r['s'] - q1[i - 1]['e']) / 1e3 if i else 0.0
    ov = [nm(k) + ' g' + k['Grid_Size_X'] for k in q2 if k['s'] < r['e'] and k['e'] > r['s']]
    print('%7.3f %8.1fus gap %5.1f  %-36s g=%-8s | %s' % (t, (r['e'] - r['s']) / 1e3, gap, nm(r), r['Grid_Size_X'], ';'.join(ov)[:70]))
print()
for k in q2:
    t = (k['s'] - t0) / 1e6
    if t_from <= t <= t_to:
        print('side %7.3f %8.1fus %-36s g=%s' % (t, (k['e'] - k['s']) / 1e3, nm(k), k['Grid_Size_X']))
