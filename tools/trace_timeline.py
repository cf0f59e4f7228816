"""Summarise one training step from a rocprofv3 kernel trace: per-queue busy time, overlap, and the longest kernels."""
import csv, glob, sys, collections
import os
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# steps are delimited by adam_kernel
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a0, a1 = adam[-2], adam[-1]
step = rows[a0 + 1:a1 + 1]
t0 = step[0]['s']; t1 = step[-1]['e']
print('step wall %.2f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step: byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = sum(r['e'] - r['s'] for r in rs)
    print('queue', q, 'kernels', len(rs), 'busy %.2f ms' % (busy / 1e6), 'span %.2f..%.2f ms' % ((rs[0]['s'] - t0) / 1e6, (rs[-1]['e'] - t0) / 1e6))
# union busy
ev = sorted([(r['s'], 1) for r in step] + [(r['e'], -1) for r in step])
cur = 0; last = t0; idle = 0; both = 0
for t, d in ev:
    if cur == 0: idle += t - last
    if cur >= 2: both += t - last
    cur += d; last = t
print('idle (no kernel running) %.2f ms, >=2 kernels concurrently %.2f ms' % (idle / 1e6, both / 1e6))
# forward/backward boundary: dice_fwd
for i, r in enumerate(step):
    if 'dice_fwd' in r['Kernel_Name']:
        print('forward ends at %.2f ms' % ((r['s'] - t0) / 1e6)); break
# per-name totals within the step
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mednet::', '')[:60]
    agg[k][0] += 1; agg[k][1] += r['e'] - r['s']
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print('%-62s %3d %7.3f ms' % (k, c, t / 1e6))
