"""Timeline of the LAST training step in a rocprofv3 kernel trace (csv): start offset, kernel, duration, queue.
usage: python tools/step_timeline.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['End_Timestamp'])
tot = {}
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name'].split('(')[0].replace('void mednet::', '').replace('mednet::', '')[:44]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[n] = tot.get(n, 0) + d
    if d > min_us:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} ms  {n:44s} {d:8.1f} us  q={r['Queue_Id']}")
print("step wall %.2f ms" % ((int(rows[b]['End_Timestamp']) - t0) / 1e6))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {k:44s} {v / 1e3:7.2f} ms")
