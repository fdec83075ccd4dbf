"""Registers, spills, scratch and static LDS of every kernel in libgpx.so (code-object metadata; runs without a GPU).
Usage: python scripts/kernel_resources.py [lib]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import codeobj
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gaussian-object-modelling_amd", "lib", "libgpx.so")
with tempfile.TemporaryDirectory() as t:
    ks = codeobj.kernels(lib, t)
names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in ks), capture_output=True, text=True).stdout.split("\n")
print("%-100s %5s %5s %5s %6s %6s %7s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
for k, nm in sorted(zip(ks, names), key=lambda t: t[1]):
    nm = nm.replace("void gpx::", "").replace("(anonymous namespace)::", "").split("(")[0]
    print("%-100s %5d %5d %5d %6d %6d %7d %7d" % (nm[:100], k["vgpr_count"], k["agpr_count"], k["sgpr_count"], k["vgpr_spill_count"],
                                             k["sgpr_spill_count"], k["private_segment_fixed_size"], k["group_segment_fixed_size"]))
