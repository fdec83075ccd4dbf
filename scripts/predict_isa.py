"""VALU instructions per (query, training point) pair in the inner loop of predict_kernel<double, KID, GRAD> (the mean /
gradient kernel, VALU-issue bound), counted in the gfx950 ISA hipcc emits for the shipped source.  The inner loop
handles 4 points x 2 queries = 8 pairs per iteration.  Runs without a GPU:
    python scripts/predict_isa.py > profiles/r02_predict_isa.txt"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gaussian-object-modelling_amd", "csrc")
KNAME = {0: "gaussian / laplace", 2: "thinplate", 3: "matern32", 4: "matern52"}
with tempfile.TemporaryDirectory() as t:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", SRC, "-c",
                    os.path.join(SRC, "gpx_predict.hip"), "-o", os.path.join(t, "p.o"), "-save-temps"], check=True, cwd=t,
                   capture_output=True)
    asm = open(os.path.join(t, "gpx_predict-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
print("# predict_kernel<double, KID, GRAD>: inner loop (4 training points x 2 queries per iteration), gfx950 ISA")
print("# %-22s %5s %10s %12s   %s" % ("kernel", "grad", "VALU/iter", "VALU/pair", "instruction mix (per iteration)"))
table = {}
for kid in (0, 2, 3, 4):
    for grad in (0, 1):
        m = re.search(r"^_ZN3gpx14predict_kernelIdLi%dELb%dELb1EEE\S*:.*?s_endpgm" % (kid, grad), asm, re.S | re.M)
        body = m.group(0)
        # the innermost loop: from the Depth=2 header to its back edge
        i = body.index("Depth=2")
        loop = body[i:]
        loop = loop[:loop.index("s_cbranch_scc0")]
        ins = [ln.split()[0] for ln in loop.splitlines()[1:] if ln.strip() and not ln.strip().startswith((";", "."))]
        valu = [x for x in ins if x.startswith("v_")]
        mix = collections.Counter(x.replace("_e32", "").replace("_e64", "") for x in valu)
        table[(kid, grad)] = len(valu) / 8.0
        print("  %-22s %5d %10d %12.1f   %s | ds_read %d" % (KNAME[kid], grad, len(valu), len(valu) / 8.0,
              " ".join("%s:%d" % kv for kv in mix.most_common(8)), sum(x.startswith("ds_read") for x in ins)))
print("# MEAN_VALU_PER_PAIR for bench.py:", {KNAME[k].split(" / ")[0]: v for (k, g), v in table.items() if g == 0})
