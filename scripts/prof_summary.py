"""Condense a rocprofv3 --kernel-trace --stats CSV dir into a small text summary for profiles/."""
import csv, glob, os, sys

def main(src, dst, note=""):
    ks = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if not ks:
        raise SystemExit("no kernel_stats.csv under " + src)
    ks.sort(key=os.path.getmtime, reverse=True)  # newest run first
    rows = list(csv.DictReader(open(ks[0])))
    with open(dst, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats summary\n# source: %s\n# %s\n" % (os.path.basename(ks[0]), note))
        fh.write("%-100s %8s %14s %14s %8s\n" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in rows:
            fh.write("%-100s %8s %14.3f %14.1f %8s\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, r["Percentage"][:7]))
    print(open(dst).read()[:3000])

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
