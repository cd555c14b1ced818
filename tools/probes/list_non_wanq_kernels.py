import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
win = float(sys.argv[2]) * 1e6
sel = [r for r in rows if int(r["Start_Timestamp"]) >= end - win]
c = collections.Counter(); t = collections.Counter()
for r in sel:
    n = r["Kernel_Name"]
    c[n] += 1; t[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = 0
for n, k in sorted(c.items(), key=lambda kv: -t[kv[0]]):
    if "wanq::" in n: continue
    tot += t[n]
    print(f"{k:5d} {t[n]/1e3:9.1f} us  {n[:110]}")
print("non-wanq total us:", tot / 1e3, " launches:", sum(k for n, k in c.items() if "wanq::" not in n), " all launches:", len(sel))
