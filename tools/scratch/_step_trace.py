"""Per-phase kernel time of one train step from a rocprofv3 kernel trace: groups kernels by name family and prints the
sequence of the last step with gaps (diagnostic)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = after the last k_adamw but one
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_adamw')]
lo, hi = idx[-2] + 1, idx[-1] + 1
step = rows[lo:hi]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, launches {len(step)}")
fam = collections.Counter(); cnt = collections.Counter()
for r in step:
    n = r['Kernel_Name']
    key = n.split('(')[0][:64]
    fam[key] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[key] += 1
for k, v in fam.most_common(60):
    print(f"{v / 1e3:9.1f} us {cnt[k]:5d}  {k}")
if len(sys.argv) > 2:
    prev = None
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1e3 if prev else 0
        print(f"{(s - t0) / 1e3:10.1f} +{gap:6.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:90]}")
        prev = e
