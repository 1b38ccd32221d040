"""timeline of ONE replayed training step from a rocprofv3 kernel trace: per-stream lanes, segments (head / MMT fwd / MMT bwd / tail / optimizer)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:70]
# steps delimited by adam_kernel
adam = [i for i, r in enumerate(rows) if "sumsq_final" in r["Kernel_Name"]]
lo, hi = adam[-2] + 1, adam[-1] + 1
seg = rows[lo:hi]
t0 = int(seg[0]['Start_Timestamp'])
print("kernels in step:", len(seg), "span ms:", (int(seg[-1]['End_Timestamp']) - t0) / 1e6)
streams = collections.OrderedDict()
for r in seg:
    streams.setdefault(r['Queue_Id'] + '/' + r['Stream_Id'], []).append(r)
print("streams:", {k: len(v) for k, v in streams.items()})
mode = sys.argv[2] if len(sys.argv) > 2 else 'all'
busy_union = 0
last_end = t0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if mode == 'all' or (mode == 'small' and (e - s) < 30000):
        print("%8.1f %8.1f  %6.1f us  q%s  %s  grid %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r['Queue_Id'], short(r['Kernel_Name']), r['Grid_Size_X']))
