"""Node-scale models (C1 / C5 shapes): mean + variance over the 128^3 grid through the host-pointer entry."""
import importlib, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
qx,qy,qz = ds.query_grid(128)
for n,prec,name in ((277,gpx.F64,'f64'),(277,gpx.F32,'f32'),(724,gpx.F32,'f32'),(300,gpx.F32_SPLIT,'split')):
    x,y,z,lab,s2 = ds.fibonacci_training_set(n)
    m = gpx.Model(gpx.make_kernel("thinplate",2.0),x,y,z,lab,s2,precision=prec)
    m.evaluate(qx,qy,qz,want_v=True)
    t=time.perf_counter(); m.evaluate(qx,qy,qz,want_v=True); dt=time.perf_counter()-t
    print("N=%d %s 128^3 grid mean+variance (host arrays): %.2f ms; device stages mean %.2f var %.2f" % (n,name,dt*1e3,m.stats["t_mean_ms"],m.stats["t_var_ms"]))
    m.close()
