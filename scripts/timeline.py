"""Per-kernel totals and the serial chain of one create() from a rocprofv3 --kernel-trace CSV: the LAST create in the
trace is cut at its kbuild launch (small models: at the small_factor_kernel launch); for every kernel name: calls, total, average; plus busy / idle time of the window."""
import csv, glob, os, sys
src = sys.argv[1]
f = max(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "kbuild_kernel" in r["Kernel_Name"] or "small_factor_kernel" in r["Kernel_Name"]]
i0 = starts[-1]
win = rows[i0:]
t0, t1 = win[0]["s"], max(r["e"] for r in win)
agg = {}
for r in win:
    nm = r["Kernel_Name"].split("(")[0].replace("void gpx::", "")
    a = agg.setdefault(nm, [0, 0])
    a[0] += 1
    a[1] += r["e"] - r["s"]
# union of busy intervals
iv = sorted((r["s"], r["e"]) for r in win)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("# last create() in %s: window %.3f ms, some kernel running %.3f ms, idle %.3f ms" % (os.path.basename(f), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
print("%-90s %6s %10s %10s" % ("kernel", "calls", "total_ms", "avg_us"))
for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-90s %6d %10.3f %10.1f" % (nm[:90], c, t / 1e6, t / c / 1e3))
