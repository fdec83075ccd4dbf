"""Agreement of the small-model variance shapes (gpx_varcols*.hip) with the general path (GPX_VAR_COLS=0): evaluate(f, v) of
fp32-mode models on 40000 lattice queries; the switches are read once per process, so the script re-runs itself per setting."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIZES = [int(a) for a in os.environ.get("VC_SIZES", "100,277,300,336,500,512,724,1000").split(",")]
if len(sys.argv) > 1:
    import numpy as np
    gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
    ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
    qx, qy, qz = ds.query_grid(34)
    out = {}
    for kn, par in (("matern52", (1.0, 1.0)), ("gaussian", (1.0, 1.0)), ("thinplate", (4.0,))):
        for n in SIZES:
            x, y, z, lab, s2 = ds.fibonacci_training_set(n)
            m = gpx.Model(gpx.make_kernel(kn, *par), x, y, z, lab, s2, precision=gpx.F32)
            out["%s/%d" % (kn, n)] = m.evaluate(qx, qy, qz, want_v=True)["v"]
            m.close()
    np.savez(sys.argv[1], **out)
else:
    import numpy as np
    tmp = "/tmp/vc_check"
    os.makedirs(tmp, exist_ok=True)
    cfgs = [("ref", {"GPX_VAR_COLS": "0"})]
    for shape in ("0", "2", "3"):
        for gen in ("1", "0"):
            cfgs.append(("shape%s gen%s" % (shape, gen), {"GPX_VAR_COLS_SHAPE": shape, "GPX_VAR_COLS_GEN": gen}))
    res = {}
    for name, env in cfgs:
        f = os.path.join(tmp, name.replace(" ", "_") + ".npz")
        subprocess.run([sys.executable, os.path.abspath(__file__), f], env=dict(os.environ, **env), check=True)
        res[name] = np.load(f)
    for name, _ in cfgs[1:]:
        worst = 0.0
        line = []
        for k in res["ref"].files:
            e = float(np.max(np.abs(res[name][k] - res["ref"][k])) / np.max(np.abs(res["ref"][k])))
            worst = max(worst, e)
            if e > 3e-6:
                line.append("%s %.2e" % (k, e))
        print("%-12s worst %.2e %s" % (name, worst, "| " + ", ".join(line) if line else "ok"), flush=True)
