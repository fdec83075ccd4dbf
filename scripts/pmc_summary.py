"""Average per-launch counter values of one kernel from rocprofv3 --pmc passes (scripts/pmc_pass.sh output dirs).
Usage: python scripts/pmc_summary.py <kernel-name-substring> <dir> [<dir> ...]"""
import collections, csv, glob, os, sys
pat, dirs = sys.argv[1], sys.argv[2:]
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("%-28s launches %3d   avg %.4g" % (k, len(v), sum(v) / len(v)))
