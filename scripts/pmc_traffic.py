"""Build profiles/<out>.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py."""
import collections, csv, glob, json, os, sys

def main(fetch_dir, write_dir, out, cmd):
    res = {"cmd": cmd,
           "units": ("FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reports half of the bytes of wide "
                     "(16 B/lane) streaming reads (MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024; "
                     "WRITE_SIZE is exact for 16 B/lane stores (check: kbuild writes 516 MiB algorithmic)"),
           "kernels": {}}
    vals = collections.defaultdict(dict)
    for d, ctr in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        f = max(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)  # newest run
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "gpx" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            vals[k][ctr] = sum(v) / len(v)
            vals[k]["dispatches"] = len(v)
    for k, d in sorted(vals.items()):
        fe, wr = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
        res["kernels"][k] = {"dispatches": d["dispatches"], "fetch_size_kib_avg": fe, "write_size_kib_avg": wr,
                             "hbm_bytes_per_dispatch": (2 * fe + wr) * 1024}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print("%-90s n=%4d  %.3f GB" % (k[:90], v["dispatches"], v["hbm_bytes_per_dispatch"] / 1e9))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], " ".join(sys.argv[4:]))
