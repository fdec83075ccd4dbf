import importlib, sys, os
sys.path.insert(0, '/root/repo'); os.chdir('/root/repo')
import torch
import bench
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
dev = torch.device("cuda", 0)
def show(tag):
    r = bench.small_model_roofline64(torch, gpx, ds, dev, 0)
    print(tag, {n: round(v["kernel_ms"], 3) for n, v in r["sizes"].items()}, flush=True)
show("fresh process          ")
bench.sharded_call_config(gpx, ds, 0)
show("after the sharded leg  ")
gpx.trim()
show("after gpx.trim()       ")
bench.small_model_roofline(torch, gpx, ds, dev, 0)
show("after the fp32 leg     ")
