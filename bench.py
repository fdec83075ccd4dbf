#!/usr/bin/env python3
"""Headline benchmark of the GP-regression hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One step = one pass of the hot path over one batch of synthetic input:
    train   : gpx_model_create  (kbuild -> blocked LDL^T on MFMA -> alpha + fp64-residual refinement ->
              inverse factor)                         N_train = 16384, fp32, Matern-5/2(1,1), sigma2 = 0.1
    predict : gpx_model_evaluate_device (mean + variance) over N_query = 2^20 lattice points already
              resident in HBM (the x < 0 half of the 128^3 grid on [-1.01, 1.01]^3).
Metric (BASELINE.json): GP train+predict time (ms per step) and query-points/s (value).

Multi-GPU (weak scaling, the north star's "independent GP models"): every rank owns one GPU and one
model of the same shape (its own jitter seed) and its own 2^20 queries; no data-path collective.  value =
queries of all ranks / max-over-ranks time.  `--mode shard` instead runs ONE model whose query grid is
x-slab sharded (strong scaling): rank 0 factorises and broadcasts the read-only state (points, alpha, D,
inverse factor) with RCCL (torch.distributed broadcast over xGMI).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TRAIN = 16384
GRID = 128
NQ = 1 << 20
KERNEL = ("matern52", (1.0, 1.0))
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_F64_MFMA_TFLOPS = 78.6
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16 matrix peak (the opt-in split contraction runs three fp16 products per fp32 one)
# VALU instructions per (query, training point) pair in the inner loop of predict_kernel<double, KID, false> (mean only),
# counted in the gfx950 ISA of the shipped library (scripts/predict_isa.py -> profiles/r02_predict_isa.txt)
# Vector-ALU work of the two small-model variance kernels per QUERY at the three sizes of roofline_small / roofline_small64, as
# wave-instructions: (vector instructions that are not MFMAs, of which transcendentals, fp64 MFMAs of the fit's add-back) --
# properties of the shipped code, counted by rocprofv3 --pmc (SQ_INSTS_VALU - SQ_INSTS_MFMA, SQ_INSTS_VALU_TRANS_*,
# SQ_INSTS_VALU_MFMA_F64 over 2^21 queries; profiles/r06_pmc_small_kernels.txt).  On gfx950 these MFMAs never co-execute with the
# vector ALU (SQ_VALU_MFMA_COEXEC_CYCLES = 0 there): a kernel's issue-slot bound is the SUM of its MFMA and vector cycles.
SMALL_VECTOR_WORK = {"f32": {277: (119.4, 12.0, 4.5), 512: (251.5, 30.0, 8.0), 724: (431.1, 56.0, 11.5)},
                     "f64": {277: (174.5, 4.5, 0.0), 512: (394.3, 10.5, 0.0), 724: (669.4, 18.0, 0.0)}}
NOMINAL_CLOCK_HZ = 2.4e9  # the clock the MFMA peaks are quoted at
N_SIMD = 1024


def issue_bound(prec, n, nq, kernel_ms, alg_mfma_cycles_per_query):
    """(algorithmic MFMA cycles + the fit's fp64 MFMAs + 4 cycles per vector instruction, 16 per transcendental) / measured cycles
    of a SIMD at the nominal clock: how far the kernel is from the bound that binds it (MFMA and vector work in series)."""
    valu, trans, fit = SMALL_VECTOR_WORK[prec][n]
    vec = (valu - trans) * 4.0 + trans * 16.0
    bound = (alg_mfma_cycles_per_query + fit * 64.0 + vec) * nq / N_SIMD
    measured = kernel_ms * 1e-3 * NOMINAL_CLOCK_HZ
    return {"issue_bound_frac": bound / measured, "issue_bound": {"algorithmic_mfma_cycles_per_query": alg_mfma_cycles_per_query,
            "fit_mfma_cycles_per_query": fit * 64.0, "vector_issue_cycles_per_query": vec,
            "what": "SIMD cycles per query; frac = their sum x queries / 1024 SIMDs / (kernel ms x 2.4 GHz)"}}


MEAN_VALU_PER_PAIR = {"gaussian": 26.125, "laplace": 26.125, "thinplate": 16.125, "matern32": 28.125, "matern52": 29.125}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["independent", "shard"], default="independent")
    ap.add_argument("--state", choices=["broadcast", "recompute"], default="broadcast",
                    help="--mode shard: how the ranks get the model -- rank 0 factorises and broadcasts the two state blobs "
                         "over RCCL, or every rank builds the model itself (no communication at all); SURVEY 8e: measure both")
    ap.add_argument("--n-train", type=int, default=N_TRAIN)
    ap.add_argument("--nq", type=int, default=NQ)
    ap.add_argument("--precision", choices=["f32", "f64", "mixed", "f32split"], default="f32")
    ap.add_argument("--kernel", default=KERNEL[0])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--query-batch", type=int, default=0, help="queries per variance batch (0 = library default)")
    ap.add_argument("--mean-only", action="store_true", help="diagnostic: skip the variance")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra F32_SPLIT / F64 / host-API legs (not part of value)")
    ap.add_argument("--no-configs", action="store_true", help="skip the record of the other BASELINE configs (C1, C2, C4 slab, C5)")
    return ap.parse_args()


def cpu_baseline(n_train, nq, kernel_name, kernel_par):
    """Oracle (CPU fp64 port of the reference algorithm) on bounded samples: one core (the reference's create is
    single-threaded and its evaluate parallelism is the caller's threads) and all host cores (OpenMP build)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    orc = importlib.import_module("gp_oracle")
    ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
    kern = orc.make_kernel(kernel_name, *kernel_par)
    ncores = os.cpu_count() or 1

    def sample(ns, nqs, omp):
        x, y, z, lab, s2 = ds.fibonacci_training_set(ns)
        qx, qy, qz = ds.query_grid(11)
        qx, qy, qz = qx[:nqs], qy[:nqs], qz[:nqs]
        t0 = time.perf_counter()
        m = orc.Model(kern, x, y, z, lab, s2, omp=omp)
        t1 = time.perf_counter()
        m.evaluate(qx, qy, qz, want_v=True)
        t2 = time.perf_counter()
        t_create, t_q = t1 - t0, (t2 - t1) / nqs
        # cost model of the reference algorithm: create ~ N^3 (unblocked LDL^T), evaluate ~ N^2 per query
        t_full = t_create * (n_train / ns) ** 3 + t_q * (n_train / ns) ** 2 * nq
        return t_create, t_q, t_full

    ns1, nq1 = 3072, 1024
    c1, q1, full1 = sample(ns1, nq1, False)
    # The same one-core path timed ONCE at the full size in the build container (scripts/cpu_full_size.py ->
    # profiles/r04_cpu_full_size.txt): measured / extrapolated from N = 3072 there
    FULL_SIZE = {"n_train": 16384, "create_s": 1443.4, "ms_per_query": 360.158, "create_ratio": 1.834, "query_ratio": 1.599,
                 "where": "profiles/r04_cpu_full_size.txt (build container, one core, 256 queries)"}
    out = {
        "value": nq / full1, "unit": "query-points/s", "cores": 1, "kind": "port", "measured_on": "this box (bounded sample, scaled)",
        "sample": ("oracle/gp_oracle.c (fp64 restatement of gp_regressor.hpp, unblocked LDL^T, per-query solve) "
                   "timed at N_train=%d (create %.2fs) and %d queries (%.2f ms/query), scaled by N^3 / N^2 to "
                   "N_train=%d, N_query=%d -> %.0f s per step.  The sample's 72 MiB matrix is cache-resident, the "
                   "2 GiB one of N_train=%d is not: the scaling flatters the CPU -- timed once at the full size on "
                   "the build container's core the create took %.2fx and a query %.2fx the extrapolation from the "
                   "same sample there (%s)" %
                   (ns1, c1, nq1, q1 * 1e3, n_train, nq, full1, n_train, FULL_SIZE["create_ratio"],
                    FULL_SIZE["query_ratio"], FULL_SIZE["where"])),
    }
    if n_train == FULL_SIZE["n_train"]:
        # the headline field is the MEASURED full-size figure; the (flattering) extrapolation of this run's sample stays beside it
        measured = nq / (FULL_SIZE["create_s"] + FULL_SIZE["ms_per_query"] * 1e-3 * nq)
        out["extrapolated_from_sample"] = {"value": out["value"], "unit": "query-points/s", "what": "this run's N_train=%d sample scaled by N^3 / N^2" % ns1}
        out["full_size_measured"] = dict(FULL_SIZE, value=measured,
                                         unit="query-points/s at the measured one-core rates (create once + N_query per-query solves)")
        out["value"] = measured
        out["measured_on"] = "build container (one core, once, at the full size); this box's own bounded sample: extrapolated_from_sample"
        out["sample"] = ("oracle/gp_oracle.c timed ONCE at the full size N_train=%d on one core of the build container (%s): create %.0f s, "
                         "%.1f ms per query -> %.0f s per step of %d queries.  This run's own bounded sample on this box: " %
                         (n_train, FULL_SIZE["where"], FULL_SIZE["create_s"], FULL_SIZE["ms_per_query"],
                          FULL_SIZE["create_s"] + FULL_SIZE["ms_per_query"] * 1e-3 * nq, nq)) + out["sample"]
    try:
        nsa, nqa = 6144, 1024
        ca, qa, fulla = sample(nsa, nqa, True)
        out["all_cores"] = {
            "value": nq / fulla, "unit": "query-points/s", "cores": ncores, "kind": "port",
            "sample": ("oracle/libgp_oracle_omp.so (OpenMP over the rows of each rank-1 update and over the queries) on %d "
                       "cores, timed at N_train=%d (create %.2fs) and %d queries (%.2f ms/query), scaled by N^3 / N^2 -> "
                       "%.0f s per step" % (ncores, nsa, ca, nqa, qa * 1e3, fulla))}
    except Exception as e:  # the one-core figure is the contract; never lose it to the extra sample
        out["all_cores"] = {"error": str(e)}
    return out


# committed PMC passes per variance tile (GPX_VAR_TILE): file, kernel-name prefix of that tile's instantiation
PMC_TRAFFIC = {"6": ("r06_pmc_traffic_w1.json", "gpx::var_w1_kernel<true"),    # one-wave tile (default; the round-6 build)
               "3": ("r03_pmc_traffic.json", "gpx::gemm_kernel<float, false, 2, 4, 4, 2, 2, 64")}   # LDS-staged fallback


def pmc_traffic(args, n_train, q_per_launch):
    """(bytes, source): HBM bytes per launch of the variance GEMM from the COMMITTED PMC passes of this bench (separate
    --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/pmc_pass.sh; FETCH_SIZE doubled as the MI355X guide prescribes for wide
    streaming reads, + WRITE_SIZE) -- counters cannot be collected inside an un-profiled run, so this figure is NOT
    measured by the run that prints it; only valid for the shape and the tile those passes were taken on."""
    tile = "3" if os.environ.get("GPX_VAR_TILE") == "3" else "6"
    if not (args.precision == "f32" and n_train == N_TRAIN and q_per_launch == 8192 and tile in PMC_TRAFFIC):
        return None, None
    fn, prefix = PMC_TRAFFIC[tile]
    try:
        for name, k in json.load(open(os.path.join(ROOT, "profiles", fn)))["kernels"].items():
            if name.startswith(prefix):
                return k["hbm_bytes_per_dispatch"], "profiles/%s (committed rocprofv3 --pmc pass of this command, not this run)" % fn
    except Exception:
        pass
    return None, None


def accuracy_record(torch, f_a, v_a, f_ref, v_ref, k0):
    """Error of one precision mode against the fp64 pipeline over ALL timed queries, in both normalisations of the
    variance error: SURVEY 8d's max|dv| / max|v_ref| and the k(0)-scaled one (v = k(0) - quadratic form)."""
    dv = (v_a - v_ref).abs()
    df = (f_a - f_ref).abs()
    vmax = float(v_ref.abs().max().item())
    fmax = float(f_ref.abs().max().item())
    return {"n_queries": int(v_ref.numel()), "reference": "GPX_PREC_F64 pipeline on the same queries",
            "v_err_over_max_v": float(dv.max().item()) / vmax, "v_err_over_max_v_k0": float(dv.max().item()) / max(vmax, k0),
            "v_err_mean_over_max_v": float(dv.mean().item()) / vmax, "max_abs_v_ref": vmax, "min_v_ref": float(v_ref.min().item()),
            "k0": k0, "f_err_over_max_f": float(df.max().item()) / fmax,
            "tolerance": "1e-5 (north star, SURVEY 8d: max|dv| / max|v_ref|)"}


def eigen_on_box():
    """Is Eigen 3 -- the third-party library the reference's arithmetic lives in (gp_regressor.hpp:9-12, CMakeLists.txt:25) --
    installed on this box?  Decides whether the oracle's LDL^T can be pinned to the real Eigen::LDLT (BASELINE.md section 4
    item 4): header search in the usual prefixes plus what the compiler itself finds."""
    import glob
    import tempfile
    cands = []
    for pat in ("/usr/include/eigen3", "/usr/local/include/eigen3", "/opt/*/include/eigen3", "/usr/include/Eigen",
                "/usr/local/include/Eigen", "/opt/rocm*/include/eigen3", "/usr/share/eigen3"):
        cands += [d for d in glob.glob(pat) if os.path.exists(os.path.join(d, "Eigen", "Core")) or os.path.exists(os.path.join(d, "Core"))]
    rec = {"present": bool(cands), "path": cands[0] if cands else None, "version": None}
    try:
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "e.cpp")
            open(src, "w").write("#include <Eigen/Core>\n#include <cstdio>\nint main(){std::printf(\"%d.%d.%d\", EIGEN_WORLD_VERSION, "
                                 "EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);}\n")
            inc = ["-I" + cands[0]] if cands else []
            r = subprocess.run(["g++", "-std=c++14"] + inc + [src, "-o", os.path.join(td, "e")], capture_output=True, text=True, timeout=60)
            if r.returncode == 0:
                rec["present"] = True
                rec["version"] = subprocess.run([os.path.join(td, "e")], capture_output=True, text=True, timeout=10).stdout
            else:
                rec["compiler"] = (r.stderr.strip().splitlines() or ["?"])[0][-160:]
    except Exception as e:
        rec["probe_error"] = str(e)
    return rec


def small_model_roofline(torch, gpx, ds, dev, local_rank):
    """The variance contraction at the reference's own model sizes (N = 166 .. 724): Matern-5/2 fp32-mode
    models of 277 / 512 / 724 points on the Fibonacci cloud, evaluate(f, v) on 2^19 lattice queries; the kernel's HIP-event time
    against the fp32 MFMA peak on the ALGORITHMIC N^2 flop per query, and the whole variance stage (fit + kernel) beside it."""
    g = 80
    t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
    idx = torch.arange(0, 2 ** 19, device=dev)
    q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
    nq = int(idx.numel())
    f = torch.empty(nq, dtype=torch.float64, device=dev)
    v = torch.empty_like(f)
    out = {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "n_query": nq,
           "traffic": "0.35 GB per 2^21-query launch (queries + fit coefficients read, v written; the operand never exists in memory): "
                      "profiles/r04_pmc_var_cols.txt (committed rocprofv3 --pmc passes of scripts/c5_stages.py, not this run)",
           "kernel": "var_cols_kernel<2 x 12 fragments, operand formed in the wave> (gpx_varcols_kernel.hpp)",
           "issue_note": "issue_bound_frac: (algorithmic MFMA + fit MFMA + vector-instruction cycles) / measured cycles at the NOMINAL 2.4 GHz; the "
                         "kernel runs at 2.02-2.16 GHz under power, where the PMC passes show 94-96 % of the SIMD cycles busy with MFMAs or vector "
                         "issue (profiles/r06_pmc_small_kernels.txt)",
           "note": "the fp32-input MFMA issues on the vector ALU's slots (profiles/r04_mfma_filler_probe.txt): the fp64 add-back of "
                   "the fit, the accumulator reads and the in-wave evaluation of the operand add to the MFMA time instead of hiding "
                   "behind it -- at N = 277 they are as long as the algorithmic MFMAs (DESIGN.md section 4)", "sizes": {}}
    for n in (277, 512, 724):
        x, y, z, lab, s2 = ds.fibonacci_training_set(n)
        m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), x, y, z, lab, s2, precision=gpx.F32, prepare_variance=True, device=local_rank)
        runs = []
        t_w = time.perf_counter()  # warm-up: 40 ms of the same evaluations (a launch of 0.6 ms behind an idle gap runs at a lower clock)
        while time.perf_counter() - t_w < 0.04:
            m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
            m.sync()
        for _ in range(5):
            m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
            m.sync()
            runs.append(dict(m.stats))
        m.close()
        # the MEDIAN of five evaluations (one slow evaluation in four moved a mean by 10 % on a round-6 box); mean and minimum beside it
        mean = {k: float(sorted(r[k] for r in runs)[len(runs) // 2]) for k in ("t_var_gemm_ms", "t_var_ms", "t_mean_ms")}
        flops = float(n) ** 2 * nq
        a = flops / (mean["t_var_gemm_ms"] * 1e-3) / 1e12
        out["sizes"][str(n)] = {"kernel_ms": mean["t_var_gemm_ms"], "launches": runs[-1]["var_gemm_launches"], "achieved": a,
                                "frac": a / PEAK_F32_MFMA_TFLOPS, "variance_stage_ms": mean["t_var_ms"],
                                # fp32 16x16x4 MFMA: 1024 MAC in 32 cycles -> N^2 / 2 MAC per query = N^2 / 64 cycles
                                **issue_bound("f32", n, nq, mean["t_var_gemm_ms"], float(n) ** 2 / 64.0),
                                "variance_stage_frac": flops / (mean["t_var_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                "mean_ms": mean["t_mean_ms"], "timing": "median of 5 evaluations after 40 ms of warm-up evaluations",
                                "kernel_ms_min": min(r["t_var_gemm_ms"] for r in runs),
                                "kernel_ms_mean": sum(r["t_var_gemm_ms"] for r in runs) / len(runs)}
    return out


def small_model_roofline64(torch, gpx, ds, dev, local_rank):
    """The same at the reference's own arithmetic: fp64 models (what the header shim creates by default) of 277 / 512 / 724 points,
    evaluate(f, v) on 2^19 lattice queries; var_cols64_kernel (gpx_varcols64.hip: variance AND mean in one launch) against the fp64
    MFMA peak on the algorithmic flop per query -- the triangle of X per 16 x 16 fragment, 256 F (F + 1), F = ceil(N / 16)."""
    g = 80
    t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
    idx = torch.arange(0, 2 ** 19, device=dev)
    q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
    nq = int(idx.numel())
    f = torch.empty(nq, dtype=torch.float64, device=dev)
    v = torch.empty_like(f)
    out = {"bound": "mfma", "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "n_query": nq, "traffic": None,
           "kernel": "var_cols64_kernel<two waves per SIMD: 1 column x 22 row fragments each, operand formed in the wave, mean carried along> (gpx_varcols64.hip)",
           "note": "the fp64 MFMA runs at the fp64 vector rate and never co-executes with the vector ALU (SQ_VALU_MFMA_COEXEC_CYCLES = 0, "
                   "profiles/r06_pmc_small_kernels.txt): MFMAs and operand evaluation add up -- issue_bound_frac prices the kernel against "
                   "that sum; the requests for X cost what the L2 behind them costs (profiles/r06_var64_parts.txt); round 5's "
                   "one-wave-per-SIMD form: 0.61 / 0.68 / 0.71 of the MFMA peak",
           "sizes": {}}
    for n in (277, 512, 724):
        m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), *ds.fibonacci_training_set(n), precision=gpx.F64, prepare_variance=True,
                      device=local_rank)
        runs = []
        t_w = time.perf_counter()  # warm-up: 40 ms of the same evaluations, as above
        while time.perf_counter() - t_w < 0.04:
            m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
            m.sync()
        for i in range(5):
            m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
            m.sync()
            runs.append(dict(m.stats))
        m.close()
        mean = {k: float(sorted(r[k] for r in runs)[len(runs) // 2]) for k in ("t_var_gemm_ms", "t_var_ms", "t_mean_ms")}  # median, as above
        nf = (n + 15) // 16
        flops = 256.0 * nf * (nf + 1) * nq
        a = flops / (mean["t_var_gemm_ms"] * 1e-3) / 1e12
        out["sizes"][str(n)] = {"kernel_ms": mean["t_var_gemm_ms"], "launches": runs[-1]["var_gemm_launches"], "achieved": a,
                                "frac": a / PEAK_F64_MFMA_TFLOPS, "variance_stage_ms": mean["t_var_ms"], "mean_ms": mean["t_mean_ms"],
                                # fp64 16x16x4 MFMA: 2048 flop in 64 cycles -> 256 F (F + 1) flop per query = 8 F (F + 1) cycles
                                **issue_bound("f64", n, nq, mean["t_var_gemm_ms"], 8.0 * nf * (nf + 1)),
                                "timing": "median of 5 evaluations after 40 ms of warm-up evaluations",
                                "kernel_ms_min": min(r["t_var_gemm_ms"] for r in runs),
                                "kernel_ms_mean": sum(r["t_var_gemm_ms"] for r in runs) / len(runs)}
    return out


def surface_config(torch, gpx, ds, dev, local_rank, kern, data, n_train):
    """SURVEY 8f.2 on the headline model: the node's fakeDeterministicSampling (src/gp_node.cpp:998-1100) as ONE call --
    gpx_model_sample_surface over the full 128^3 lattice: mean everywhere, |f| <= 0.01 compacted on the device, variance of
    the survivors only.  Host arrays in and out (the C ABI's form), so the PCIe copies of 3 x 8 B per lattice point are in."""
    import numpy as np
    tt = np.linspace(-1.01, 1.01, GRID)
    gx, gy, gz = np.meshgrid(tt, tt, tt, indexing="ij")
    qx, qy, qz = gx.ravel(), gy.ravel(), gz.ravel()
    m = gpx.Model(kern, *data, precision=gpx.F32, prepare_variance=True, device=local_rank)
    m.sync()
    m.sample_surface(qx[:4096], qy[:4096], qz[:4096], f_tol=0.01)
    t0 = time.perf_counter()
    o = m.sample_surface(qx, qy, qz, f_tol=0.01)
    dt = time.perf_counter() - t0
    st = m.stats
    m.close()
    ns = int(o["n_total"])
    return {"workload": "C3 model (N=%d fp32 matern52) + gpx_model_sample_surface over the whole 128^3 lattice, |f| <= 0.01, "
                        "variance on the survivors only (src/gp_node.cpp:1066-1100 as one call); host arrays in and out" % n_train,
            "n_query": int(qx.size), "survivors": ns, "ms": dt * 1e3, "value": qx.size / dt, "unit": "lattice points/s",
            "survivor_variance_ms": st["t_var_ms"], "survivor_mean_ms": st["t_mean_ms"]}


def sharded_call_config(gpx, ds, local_rank):
    """gpx_model_evaluate_sharded / gpx_model_sample_surface_sharded with TWO replicas on this one ordinal (what a one-GPU box can
    run: no scaling claim, a second ordinal has not run) against the single calls: are the results equal bit for bit, and what does
    the cut cost on one device.  A model of the reference's own size (N = 724, fp32 mode: the small-model kernels -- the two
    replicas run at the same time, and with the headline model their launches would sit in the profiler's average of the dominant
    kernel), host arrays in and out; 2^20 queries (four 2^18-query slices: two per replica) for evaluate(f, v), the 128^3 lattice
    for sampleSurface."""
    import numpy as np
    n = 724
    m = gpx.Model(gpx.make_kernel("gaussian", 1.0, 1.0), *ds.fibonacci_training_set(n), precision=gpx.F32, prepare_variance=True, device=local_rank)
    reps = [m] + m.replicate([local_rank])
    tt = np.linspace(-1.01, 1.01, GRID)
    gx, gy, gz = np.meshgrid(tt[:64], tt, tt, indexing="ij")
    q = (gx.ravel().copy(), gy.ravel().copy(), gz.ravel().copy())
    m.evaluate(q[0][:4096], q[1][:4096], q[2][:4096], want_v=True)
    t0 = time.perf_counter()
    one = m.evaluate(*q, want_v=True)
    t1 = time.perf_counter()
    many = gpx.evaluate_sharded(reps, *q, want_v=True)
    t2 = time.perf_counter()
    same_eval = bool(np.array_equal(one["f"], many["f"]) and np.array_equal(one["v"], many["v"]))
    sx, sy, sz = (a.ravel().copy() for a in np.meshgrid(tt, tt, tt, indexing="ij"))
    t3 = time.perf_counter()
    s_one = m.sample_surface(sx, sy, sz, f_tol=0.01)
    t4 = time.perf_counter()
    s_many = gpx.sample_surface_sharded(reps, sx, sy, sz, f_tol=0.01)
    t5 = time.perf_counter()
    same_surf = bool(np.array_equal(s_one["idx"], s_many["idx"]) and np.array_equal(s_one["f"], s_many["f"]) and
                     np.array_equal(s_one["v"], s_many["v"]))
    for r in reps[1:]:
        r.close()
    m.close()
    return {"workload": "N=%d Gaussian(1,1) fp32 mode: gpx_model_evaluate_sharded(f, v) on %d queries and gpx_model_sample_surface_sharded "
                        "on the 128^3 lattice over 2 replicas on ONE ordinal, against the single calls (host arrays in and out)" % (n, q[0].size),
            "replicas": 2, "ordinals": 1, "evaluate_bit_identical": same_eval, "surface_bit_identical": same_surf,
            "evaluate_ms": {"single": (t1 - t0) * 1e3, "sharded": (t2 - t1) * 1e3},
            "surface_ms": {"single": (t4 - t3) * 1e3, "sharded": (t5 - t4) * 1e3, "survivors": int(s_one["n_total"])},
            "ms": (t2 - t1) * 1e3}


def extra_configs(torch, gpx, ds, sharding, dev, local_rank):
    """The other BASELINE.json configurations on this one GPU, after the timed region and never part of `value`:
    ms per train+predict(mean+variance) step and query-points/s each (one warm-up, then timed once or twice)."""
    import numpy as np
    out = {}

    def lattice(g, lo=0, hi=None, scale=1.01):
        hi = g ** 3 if hi is None else hi
        t = torch.linspace(-scale, scale, g, dtype=torch.float64, device=dev)
        idx = torch.arange(lo, hi, device=dev, dtype=torch.int64)
        return t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()

    def run(name, what, kern, data, prec, q, reps):
        qx, qy, qz = q
        nq = int(qx.numel())
        f = torch.empty(nq, dtype=torch.float64, device=dev)
        v = torch.empty(nq, dtype=torch.float64, device=dev)

        def step():
            m = gpx.Model(kern, *data, precision=prec, prepare_variance=True, device=local_rank)
            m.evaluate_device(nq, qx.data_ptr(), qy.data_ptr(), qz.data_ptr(), f.data_ptr(), v.data_ptr())
            m.sync()
            st = m.stats
            m.close()
            return st
        for _ in range(3 if reps >= 10 else 1):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fb = 0
        for _ in range(reps):
            st = step()
            fb += int(st["solve_fallbacks"])
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"workload": what, "ms_per_step": dt * 1e3, "value": nq / dt, "unit": "query-points/s",
                     "n_train": int(st["n"]), "n_query": nq, "fallbacks": fb,
                     "stages_ms": {k: st[k] for k in ("t_kbuild_ms", "t_factor_ms", "t_solve_ms", "t_inverse_ms", "t_mean_ms", "t_var_ms")}}

    pcd_dir = os.path.join(ROOT, "tests", "golden", "pcd")
    try:  # C1: the reference's own CPU-runnable case (tests/test_gp.cpp shape): mugD, Gaussian, 32^3 grid, fp64
        pts = gpx.pcd_read(os.path.join(pcd_dir, "mugD.pcd"))
        run("C1", "resources/mugD.pcd -> node training set (262 + 15 points), Gaussian(1,1), fp64, 32^3 grid",
            gpx.make_kernel("gaussian", 1.0, 1.0), gpx.node_training_set(pts), gpx.F64, lattice(32), 50)
    except Exception as e:
        out["C1"] = {"error": str(e)}
    try:  # C2
        run("C2", "synthetic sphere N=4096, Gaussian(1,1), fp64, 64^3 grid", gpx.make_kernel("gaussian", 1.0, 1.0),
            ds.fibonacci_training_set(4096), gpx.F64, lattice(64), 2)
    except Exception as e:
        out["C2"] = {"error": str(e)}
    try:  # C5: the eight objects of scripts/perform.sh one after the other, Gaussian(1,1), fp32, 128^3 grid each
        names = ["bowlA", "bowlB", "containerA", "containerB", "jug", "kettle", "pot", "mugD"]
        sets = [gpx.node_training_set(gpx.pcd_read(os.path.join(pcd_dir, nm + ".pcd"))) for nm in names]
        kern = gpx.make_kernel("gaussian", 1.0, 1.0)
        q = lattice(128)
        nq = int(q[0].numel())
        f = torch.empty(nq, dtype=torch.float64, device=dev)
        v = torch.empty(nq, dtype=torch.float64, device=dev)

        def all_objects(prec=gpx.F32):
            for d_ in sets:
                m = gpx.Model(kern, *d_, precision=prec, prepare_variance=True, device=local_rank)
                m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
                m.sync()
                m.close()
        all_objects()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        all_objects()
        dt = time.perf_counter() - t0
        all_objects(gpx.F64)  # the same in the reference's own arithmetic (what the header shim creates by default)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        all_objects(gpx.F64)
        dt64 = time.perf_counter() - t0
        all_objects(gpx.F32_SPLIT)  # ... and in the opt-in split-fp16 mode (small models: gpx_varcols16.hip)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        all_objects(gpx.F32_SPLIT)
        dts = time.perf_counter() - t0
        out["C5_split"] = {"workload": "configs.C5 in the opt-in GPX_PREC_F32_SPLIT mode (fp16 matrix cores, hi / lo halves, fp32 accuracy: "
                                       "gpx_varcols16.hip for the models of <= 1024 points)",
                           "ms_per_step": dts * 1e3, "ms_per_object": dts * 1e3 / len(sets), "value": nq * len(sets) / dts,
                           "unit": "query-points/s", "n_query": nq * len(sets)}
        out["C5_f64"] = {"workload": "configs.C5 with fp64 models (small-model fp64 kernel, gpx_varcols64.hip: variance and mean in one launch)",
                         "ms_per_step": dt64 * 1e3, "ms_per_object": dt64 * 1e3 / len(sets), "value": nq * len(sets) / dt64,
                         "unit": "query-points/s", "n_query": nq * len(sets)}
        out["C5"] = {"workload": "8 objects (%s; N = %s), Gaussian(1,1), fp32 mode, 128^3 grid each, one after "
                                 "the other on ONE GPU (the 8-GPU form is one object per rank: bench.py --gpus 8)"
                                 % (", ".join(names), ", ".join(str(len(d_[0])) for d_ in sets)),
                     "ms_per_step": dt * 1e3, "ms_per_object": dt * 1e3 / len(sets), "value": nq * len(sets) / dt,
                     "unit": "query-points/s", "n_query": nq * len(sets)}
    except Exception as e:
        out["C5"] = {"error": str(e)}
    try:  # C4: one rank's share of the 256^3 grid sharded over 8 GPUs, on a shell committed from the state blobs
        n = N_TRAIN
        kern = gpx.make_kernel("thinplate", 4.0)
        data = ds.fibonacci_training_set(n)
        world4, rank4 = 8, 3
        lo, hi = sharding.slab_range(256 ** 3, rank4, world4)
        q = lattice(256, lo, hi)
        nq = hi - lo
        f = torch.empty(nq, dtype=torch.float64, device=dev)
        v = torch.empty(nq, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        src = gpx.Model(kern, *data, precision=gpx.F32, prepare_variance=True, device=local_rank)
        src.sync()
        t1 = time.perf_counter()
        sh = gpx.Model.shell(kern, n, precision=gpx.F32, device=local_rank)
        for part in (0, 1):
            a = sharding.device_blob_as_tensor(torch, *src.state_blob(part), dev)
            b = sharding.device_blob_as_tensor(torch, *sh.state_blob(part), dev)
            b.copy_(a)  # stands in for the RCCL broadcast of the same bytes
        torch.cuda.synchronize()
        sh.commit(with_variance=True)
        t2 = time.perf_counter()
        sh.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
        sh.sync()
        t3 = time.perf_counter()
        st = src.stats
        src.close()
        sh.close()
        out["C4_slab"] = {"workload": "N=16384 thin-plate R=4 fp32 mode; slab %d of %d of the 256^3 grid (%d queries) evaluated on a shell "
                                      "committed from the two state blobs (device-to-device copy in place of the RCCL broadcast)" % (rank4, world4, nq),
                          "t_train_ms": (t1 - t0) * 1e3, "t_state_copy_commit_ms": (t2 - t1) * 1e3, "t_predict_ms": (t3 - t2) * 1e3,
                          "ms_per_step": (t3 - t0) * 1e3, "value": nq / (t3 - t0), "unit": "query-points/s (one rank's slab)",
                          "n_query": nq, "v_min": float(v.min().item()), "v_max": float(v.max().item()),
                          "train_stages_ms": {k: st[k] for k in ("t_kbuild_ms", "t_factor_ms", "t_solve_ms", "t_inverse_ms")}}
    except Exception as e:
        out["C4_slab"] = {"error": str(e)}
    return out


C5_OBJECTS = ["bowlA", "bowlB", "containerA", "containerB", "jug", "kettle", "pot", "mugD"]  # scripts/perform.sh of the reference


class _DeviceBackend:
    """libgpx on this rank's GPU behind the backend protocol of sharding.sharded_grid_step / objects_per_rank_step."""

    def __init__(self, torch, gpx, sharding, dev, local_rank, kern, data, prec):
        self.torch, self.gpx, self.sharding, self.dev, self.local_rank = torch, gpx, sharding, dev, local_rank
        self.kern, self.data, self.prec = kern, data, prec
        self.q = self.f = self.v = None

    def lattice(self, g, lo, hi, scale=1.01):
        """Lattice points [lo, hi) of the g^3 grid (x slowest, z fastest) resident in HBM, plus their output arrays."""
        torch = self.torch
        t = torch.linspace(-scale, scale, g, dtype=torch.float64, device=self.dev)
        idx = torch.arange(lo, hi, device=self.dev, dtype=torch.int64)
        self.q = (t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous())
        del idx
        self.f = torch.empty(hi - lo, dtype=torch.float64, device=self.dev)
        self.v = torch.empty(hi - lo, dtype=torch.float64, device=self.dev)
        self.range = (g, lo, hi)

    def sync(self):
        self.torch.cuda.synchronize()

    def train(self):
        return self.gpx.Model(self.kern, *self.data, precision=self.prec, prepare_variance=True, device=self.local_rank)

    def shell(self):
        return self.gpx.Model.shell(self.kern, len(self.data[0]), precision=self.prec, device=self.local_rank)

    def blobs(self, m):
        return [self.sharding.device_blob_as_tensor(self.torch, *m.state_blob(part), self.dev) for part in (0, 1)]

    def commit(self, m):
        m.commit(with_variance=True)

    def evaluate(self, m, nq=None):
        nq = int(self.f.numel()) if nq is None else nq
        m.evaluate_device(nq, self.q[0].data_ptr(), self.q[1].data_ptr(), self.q[2].data_ptr(), self.f.data_ptr(), self.v.data_ptr())
        m.sync()

    def predict(self, m, g, x_lo, x_hi):
        if self.range != (g, x_lo * g * g, x_hi * g * g):
            raise RuntimeError("backend lattice %s was prepared for another slab than planes [%d, %d)" % (self.range, x_lo, x_hi))
        self.evaluate(m)
        f, v = self.f, self.v
        return int(f.numel()), float(f.sum().item()), float(v.sum().item()), float(v.min().item()), float(v.max().item())

    def close(self, m):
        m.close()


def multi_gpu_configs(torch, gpx, ds, sharding, dist, rank, world, dev, local_rank, grid4=256, grid5=128):
    """BASELINE configs 4 and 5 in their multi-GPU form, run by EVERY rank after the timed region (never part of `value`):
      C4_shard    -- N = 16384 thin plate R = 4, fp32 mode, the real 256^3 lattice cut into x-slabs by
                     sharding.grid_x_slab; once with the state broadcast over RCCL, once with every rank rebuilding the model;
      C5_per_rank -- the eight PCD objects, object o on rank o mod world (one per GPU at world 8), Gaussian(1,1), fp32 mode,
                     128^3 lattice each, no data-path collective.
    The rank logic is sharding.sharded_grid_step / objects_per_rank_step (the world-2 gloo tests run the same functions)."""
    out = {}
    clock = time.perf_counter
    try:
        n = N_TRAIN
        be = _DeviceBackend(torch, gpx, sharding, dev, local_rank, gpx.make_kernel("thinplate", 4.0), ds.fibonacci_training_set(n), gpx.F32)
        x_lo, x_hi, lo, hi = sharding.grid_x_slab(grid4, rank, world)
        be.lattice(grid4, lo, hi)
        m = be.train()  # warm-up: code objects, the fp64 training temporaries and the pool of this shape
        if hi > lo:
            be.evaluate(m, min(hi - lo, 8192))
        be.close(m)
        rec = {"workload": "C4: N=%d thin-plate R=4 fp32 mode, ONE model, the %d^3 lattice (%d queries) in x-slabs over %d rank(s)"
                           % (n, grid4, grid4 ** 3, world)}
        for state in ("broadcast", "recompute"):
            rec[state] = sharding.sharded_grid_step(dist, torch, rank, world, grid4, state, be, dev, clock)
        out["C4_shard"] = rec
        del be
        gpx.trim()
        torch.cuda.empty_cache()
    except Exception as e:
        if world > 1:
            raise  # a rank that left the collective sequence would hang the others: fail the job loudly instead
        out["C4_shard"] = {"error": str(e)}
    try:
        pcd_dir = os.path.join(ROOT, "tests", "golden", "pcd")
        kern = gpx.make_kernel("gaussian", 1.0, 1.0)
        mine = sharding.objects_of_rank(len(C5_OBJECTS), rank, world)
        sets = {o: gpx.node_training_set(gpx.pcd_read(os.path.join(pcd_dir, C5_OBJECTS[o] + ".pcd"))) for o in mine}
        be = _DeviceBackend(torch, gpx, sharding, dev, local_rank, kern, None, gpx.F32)
        be.lattice(grid5, 0, grid5 ** 3)

        def run_object(o):
            be.data = sets[o]
            m = be.train()
            be.evaluate(m)
            m.close()
            return len(sets[o][0]), int(be.f.numel()), float(be.f.sum().item()), float(be.v.sum().item())

        sharding.objects_per_rank_step(dist, torch, rank, world, len(C5_OBJECTS), run_object, be, dev, clock)  # warm-up
        rec = sharding.objects_per_rank_step(dist, torch, rank, world, len(C5_OBJECTS), run_object, be, dev, clock)
        for o in rec["objects"]:
            o["name"] = C5_OBJECTS[o["object"]]
        rec["workload"] = ("C5: %d independent objects (%s), Gaussian(1,1), fp32 mode, %d^3 lattice each; object o on rank o mod %d"
                           % (len(C5_OBJECTS), ", ".join(C5_OBJECTS), grid5, world))
        out["C5_per_rank"] = rec
    except Exception as e:
        if world > 1:
            raise
        out["C5_per_rank"] = {"error": str(e)}
    return out


def node_pattern_config():
    """The UNCHANGED drop-in caller (row 8b): the reference node's own sampling loop (src/gp_node.cpp:998-1100) -- 29 x-slabs
    of 841 std::threads, each one evaluate(f, v) call with ONE query point on a ThinPlate(2.0) model of resources/mugD.pcd --
    compiled against the header shim (scripts/node_pattern_bench.cpp -> lib/node_pattern_bench, built by build()) and run
    as a child process; beside it the same grid as one batched evaluate and as one sampleSurface call, and the oracle
    doing the same per-point solves on this box's host cores (one core; all cores) as the CPU baseline of THAT pattern."""
    import numpy as np
    exe = os.path.join(ROOT, "gaussian-object-modelling_amd", "lib", "node_pattern_bench")
    pcd = os.path.join(ROOT, "tests", "golden", "pcd", "mugD.pcd")
    if not os.path.exists(exe):
        return {"error": "lib/node_pattern_bench not built (python -c 'import __graft_entry__ as g; g.build()')"}
    r = subprocess.run([exe, pcd, "--json"], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return {"error": "node_pattern_bench rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
    rec = json.loads(lines[-1])
    out = {"workload": "gp_node.cpp's fakeDeterministicSampling as written: 29^3 = %d single-point evaluate(f, v) calls, one std::thread "
                       "each (joined per x-slab of 841), header shim -> C ABI -> libgpx; mugD.pcd node training set (N = %d), ThinPlate(2.0)"
                       % (rec["calls"], rec["n_train"]),
           "ms_per_grid": rec["node_ms"], "us_per_call": rec["us_per_call"], "threads_only_ms": rec["threads_only_ms"],
           "create_ms": rec["create_ms"], "one_batched_evaluate_ms": rec["batched_evaluate_ms"],
           "one_sample_surface_call_ms": rec["sample_surface_ms"], "surface_points": rec["survivors"],
           "same_surface_as_batched": rec["survivors"] == rec["kept_by_node_loop"] and r.returncode == 0,
           "value": rec["calls"] / (rec["node_ms"] * 1e-3), "unit": "single-point evaluate(f, v) calls/s"}
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        orc = importlib.import_module("gp_oracle")
        gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
        data = gpx.node_training_set(gpx.pcd_read(pcd))
        ax, v = [], -1.01
        while v <= 1.01:  # the node's accumulating loop (src/gp_node.cpp:1025-1036)
            ax.append(v)
            v += 0.07
        g = np.array(ax)
        qx, qy, qz = (a.ravel().copy() for a in np.meshgrid(g, g, g, indexing="ij"))
        cpu = {}
        for name, omp in (("one_core", False), ("all_cores", True)):
            om = orc.Model(orc.make_kernel("thinplate", 2.0), *data, omp=omp)
            t0 = time.perf_counter()
            om.evaluate(qx, qy, qz, want_v=True)
            dt = time.perf_counter() - t0
            cpu[name] = {"ms_per_grid": dt * 1e3, "us_per_call": dt * 1e6 / len(qx), "cores": (os.cpu_count() or 1) if omp else 1}
        out["cpu_baseline"] = dict(cpu, kind="port", sample="oracle/gp_oracle.c: the same %d per-point evaluate(f, v) solves "
                                   "(k_q, one LDL^T solve each) on the whole grid, no thread creation" % len(qx))
    except Exception as e:
        out["cpu_baseline"] = {"error": str(e)}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # not under torch.distributed.run: start it as a child (nothing has touched the GPU yet)
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The contract is ONE JSON line on stdout.  Libraries underneath write banners to file descriptor 1 (RCCL prints
    # its version block at the first communicator), so everything but the result line goes to stderr from here on.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
    ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
    sharding = importlib.import_module("gaussian-object-modelling_amd.sharding")
    gpx.lib()  # fail loudly if the HIP extension is missing
    if not torch.cuda.is_available() or gpx.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (libgpx has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("GPX_BENCH_FORCE_DIST") == "1":  # the env switch rehearses the RCCL path on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")  # only missing in the one-process rehearsal (GPX_BENCH_FORCE_DIST=1)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    prec = {"f32": gpx.F32, "f64": gpx.F64, "mixed": gpx.MIXED, "f32split": gpx.F32_SPLIT}[args.precision]
    kpar = (4.0,) if args.kernel == "thinplate" else (1.0, 1.0)
    kern = gpx.make_kernel(args.kernel, *kpar)
    n_train, nq = args.n_train, args.nq
    shard = args.mode == "shard" and dist is not None

    # ---- synthetic inputs (SURVEY.md 8d recipe); queries resident in HBM before the timed region ----
    seed = 20151106 + (0 if shard else rank)
    x, y, z, lab, s2 = ds.fibonacci_training_set(n_train, seed=seed)
    t = torch.linspace(-1.01, 1.01, GRID, dtype=torch.float64, device=dev)
    if shard:
        # x-slab of the first nq lattice points (strong scaling: total work fixed)
        lo, hi = sharding.slab_range(nq, rank, world)
    else:
        lo, hi = 0, nq
    idx = torch.arange(lo, hi, device=dev, dtype=torch.int64)
    qx = t[(idx // (GRID * GRID)) % GRID].contiguous()
    qy = t[(idx // GRID) % GRID].contiguous()
    qz = t[idx % GRID].contiguous()
    nq_local = int(idx.numel())
    f = torch.empty(nq_local, dtype=torch.float64, device=dev)
    v = torch.empty(nq_local, dtype=torch.float64, device=dev)
    want_v = not args.mean_only

    model = [None]
    stats_acc = []
    phases = []  # per step: (t_model, t_exchange, t_commit, t_predict) of this rank, seconds

    def step():
        if model[0] is not None:
            model[0].close()
        t0 = time.perf_counter()
        t2 = t3 = None
        if shard and args.state == "broadcast":
            if rank == 0:
                m = gpx.Model(kern, x, y, z, lab, s2, precision=prec, prepare_variance=want_v, device=local_rank,
                              query_batch=args.query_batch)
            else:
                m = gpx.Model.shell(kern, n_train, precision=prec, device=local_rank)
            t1 = time.perf_counter()
            bufs = [sharding.device_blob_as_tensor(torch, *m.state_blob(part), dev)
                    for part in ((0, 1) if want_v else (0,))]
            sharding.broadcast_state(dist, bufs, src=0)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if rank != 0:
                m.commit(with_variance=want_v)
            t3 = time.perf_counter()
        else:  # independent models, or the sharded grid with every rank building the (same) model itself
            m = gpx.Model(kern, x, y, z, lab, s2, precision=prec, prepare_variance=want_v, device=local_rank,
                          query_batch=args.query_batch)
            t1 = t2 = t3 = time.perf_counter()
        model[0] = m
        m.evaluate_device(nq_local, qx.data_ptr(), qy.data_ptr(), qz.data_ptr(), f.data_ptr(),
                          v.data_ptr() if want_v else None)
        m.sync()
        phases.append((t1 - t0, t2 - t1, t3 - t2, time.perf_counter() - t3))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    phases.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        stats_acc.append(model[0].stats)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    shard_rec = None
    if shard:  # phase times of every rank (means over the timed steps) -> the `shard` record on rank 0
        mine = torch.tensor([sum(p[i] for p in phases) / len(phases) for i in range(4)], dtype=torch.float64, device=dev)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        shard_rec = sharding.shard_record(args.state, [tuple(p.tolist()) for p in allp])
        shard_rec["state_bytes"] = int(sum(model[0].state_blob(part)[1] for part in ((0, 1) if want_v else (0,))))
    total_q = nq if shard else nq * world
    value = total_q / (elapsed / args.steps)

    mg = None
    if dist is not None and not shard and not args.no_configs and args.precision == "f32" and n_train == N_TRAIN:
        # BASELINE configs 4 and 5 in their multi-GPU form: collective legs, every rank takes part (after the timed region)
        if model[0] is not None:
            model[0].close()
            model[0] = None
        gpx.trim()
        mg = multi_gpu_configs(torch, gpx, ds, sharding, dist, rank, world, dev, local_rank)

    if rank == 0:
        st = {k: float(np.mean([s[k] for s in stats_acc])) for k in stats_acc[0]}
        peak = PEAK_F64_MFMA_TFLOPS if prec == gpx.F64 else PEAK_F32_MFMA_TFLOPS
        gemm_t = "f64" if prec == gpx.F64 else "f32"
        roof = None
        if want_v and st["var_gemm_launches"] > 0:
            launches = st["var_gemm_launches"]
            avg_ms = st["t_var_gemm_ms"] / launches
            q_per_launch = nq_local / launches
            flops_per_launch = float(n_train) ** 2 * q_per_launch  # N^2 flop per query
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            vkernel, vpeak = "gemm_kernel<%s,NT,COLSQ> (predict_var)" % gemm_t, peak
            if prec != gpx.F64 and os.environ.get("GPX_VAR_TILE", "6") != "3":
                vkernel = "var_w1_kernel<fit added back in fp64> (predict_var; one wave per workgroup, 128x128 tile, no LDS)"
            if prec == gpx.F32_SPLIT:  # three fp16 MFMA products per algorithmic multiply-add: price against the fp16 peak
                vkernel, vpeak, achieved = "vsplit_gemm_kernel (3 fp16 MFMA products per fp32 product, LDS-DMA staged)", PEAK_F16_MFMA_TFLOPS, 3 * achieved
            roof = {"bound": "mfma", "kernel": vkernel,
                    "achieved": achieved, "peak": vpeak, "unit": "TFLOP/s", "frac": achieved / vpeak,
                    "traffic": pmc_traffic(args, n_train, q_per_launch)[0],
                    "traffic_source": pmc_traffic(args, n_train, q_per_launch)[1],
                    "avg_launch_ms": avg_ms, "launches_per_step": launches,
                    "algorithmic_flops_per_launch": flops_per_launch}
        out = {
            "metric": "GP train+predict query-points/s (N_train=%d, N_query=%d)" % (n_train, nq),
            "value": value, "unit": "query-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "f16 hi/lo split operands, f32 accumulate" if prec == gpx.F32_SPLIT else args.precision, "data": "synthetic",
            "config": {"workload": "C3-headline: N_train=%d %s %s(%s) sigma2=0.1, train (kbuild+LDL^T+alpha+inverse "
                                   "factor) + predict mean%s over N_query=%d lattice points of the 128^3 grid per %s"
                                   % (n_train, args.precision, args.kernel, ",".join(str(p) for p in kpar),
                                      "+variance" if want_v else "", nq, "job" if shard else "GPU"),
                       "mode": args.mode, "n_train": n_train, "n_query": nq,
                       "parallelism": "%s x%d" % (args.mode + ("/" + args.state if shard else ""), world)},
            "stages_ms": {k: st[k] for k in ("t_kbuild_ms", "t_factor_ms", "t_factor_gemm_ms", "t_solve_ms",
                                             "t_inverse_ms", "t_mean_ms", "t_var_ms", "t_var_gemm_ms")},
            "alpha_residual": st["alpha_residual"],
        }
        # a dataflow launch that gave up and was redone by the launch chain (gpx_stats.solve_fallbacks) would only look like a
        # slower create: the count over the timed steps is on the line, next to `value`
        fallbacks = int(sum(s_["solve_fallbacks"] for s_ in stats_acc))
        out["fallbacks"] = fallbacks
        out["stages_ms"]["solve_fallbacks"] = fallbacks
        if fallbacks:
            out["fallbacks_note"] = ("%d of %d timed creates fell back from a one-launch dataflow kernel to the launch chain "
                                     "(a wait gave up on its clock budget): `value` is NOT the product path's number" % (fallbacks, args.steps))
            sys.stderr.write("bench.py: WARNING: %s\n" % out["fallbacks_note"])
        if shard_rec:
            out["shard"] = shard_rec
        if roof:
            out["roofline"] = roof
            fg = st["factor_gemm_launches"]
            if fg > 0 and st["t_factor_gemm_ms"] > 0:
                # two figures: the event-timed trailing-update launches against their OWN flops (lower tiles x 2 x 128^2 x
                # K; the look-ahead's 256-column strip products and the panel chain are outside these events), and the
                # whole factorisation, N^3/3 over everything between kbuild and the alpha solve
                tr = st["factor_gemm_flops"] / (st["t_factor_gemm_ms"] * 1e-3) / 1e12
                whole = (n_train ** 3 / 3.0) / (st["t_factor_ms"] * 1e-3) / 1e12
                out["roofline_factor"] = {
                    "bound": "mfma", "kernel": "gemm_kernel<%s,NT,STORE> (LDL^T trailing update, K = panel width)" % gemm_t,
                    "achieved": tr, "peak": peak, "unit": "TFLOP/s", "frac": tr / peak,
                    "timed_launches": fg, "timed_flops": st["factor_gemm_flops"], "timed_ms": st["t_factor_gemm_ms"],
                    "whole_ldlt": {"what": "N^3/3 flop / t_factor_ms (diagonal blocks, panel solves, strip and trailing updates)",
                                   "achieved": whole, "frac": whole / peak, "ms": st["t_factor_ms"]}}
            elif st["t_factor_ms"] > 0:
                # kernel matrix + LDL^T as ONE dataflow launch (csrc/gpx_dataflow.hpp; fp32 up to 16384 rows): there is no separate
                # trailing-update launch to time -- the figure is the whole launch against N^3/3 flop (the kernel-matrix entries it
                # also forms are not counted)
                whole = (n_train ** 3 / 3.0) / (st["t_factor_ms"] * 1e-3) / 1e12
                out["roofline_factor"] = {
                    "bound": "mfma",
                    "kernel": ("wide_factor_kernel<%s> (kernel matrix + LDL^T as one dataflow launch over 128 x 128 tiles, csrc/gpx_dataflow_wide.hpp)"
                               if n_train >= 8192 else
                               "mid_factor_kernel<%s> (kernel matrix + LDL^T as one dataflow launch over 64 x 64 tiles, csrc/gpx_dataflow.hpp)") % gemm_t,
                    "achieved": whole, "peak": peak, "unit": "TFLOP/s", "frac": whole / peak, "ms": st["t_factor_ms"],
                    "what": "N^3/3 flop / t_factor_ms (everything between the upload of the points and the alpha solve)",
                    "whole_ldlt": {"what": "the same figure (one launch)", "achieved": whole, "frac": whole / peak, "ms": st["t_factor_ms"]}}
            if want_v and st["t_inverse_ms"] > 0:
                # the inverse factor X = L^-1 (recursive doubling, assembled in fp64 in every mode when the fp64 temporaries
                # fit): N^3/3 flop over the whole stage, casts and the small levels on the LDS tiles included
                inv = (n_train ** 3 / 3.0) / (st["t_inverse_ms"] * 1e-3) / 1e12
                out["roofline_inverse"] = {
                    "bound": "mfma", "kernel": "w1_f64_nn_kernel (levels with K >= 1024) + gemm_kernel<f64,NN,STORE> (small levels), fp64",
                    "achieved": inv, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": inv / PEAK_F64_MFMA_TFLOPS,
                    "ms": st["t_inverse_ms"], "what": "N^3/3 flop / t_inverse_ms"}
        esz = 8 if prec == gpx.F64 else 4
        npad = int(st["n_padded"])
        k0 = 64.0 if args.kernel == "thinplate" else (2.0 if args.kernel == "laplace" else 1.0)
        f_timed, v_timed = f.clone(), v.clone()  # results of the last timed step (the legs below overwrite f, v)
        HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md
        if world == 1 and not shard:
            # The two HBM-write-bound stages last ~0.1 ms per launch: a HIP event pair around ONE such launch mostly
            # times the events.  They are therefore measured here on their own, after the timed region and never part
            # of `value`: the same kernels through the stand-alone stage entries (gpx_dev_kbuild / gpx_dev_kqp) on the
            # same shapes, 20 launches back to back between two events on the launching stream.
            try:
                import ctypes as C
                L = gpx.lib()
                tdt = torch.float64 if prec == gpx.F64 else torch.float32
                stage_prec = gpx.F64 if prec == gpx.F64 else gpx.F32
                pts = [torch.zeros(npad, dtype=tdt, device=dev) for _ in range(3)]
                for t_, a_ in zip(pts, (x, y, z)):
                    t_[:n_train] = torch.from_numpy(a_).to(dev).to(tdt)
                s2t = torch.zeros(npad, dtype=tdt, device=dev)
                s2t[:n_train] = 0.1
                strm = torch.cuda.current_stream()
                vp = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else None
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 20

                def timed(fn):
                    fn()
                    torch.cuda.synchronize()
                    e0.record(strm)
                    for _ in range(reps):
                        fn()
                    e1.record(strm)
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / reps

                Kbuf = torch.empty(npad * npad, dtype=tdt, device=dev)
                kb = lambda: gpx._check(L.gpx_dev_kbuild(C.byref(kern), stage_prec, n_train, npad, vp(pts[0]), vp(pts[1]),
                                                         vp(pts[2]), vp(s2t), vp(Kbuf), None, C.c_void_p(strm.cuda_stream)))
                ms = timed(kb)
                nt = npad // 128
                kb_bytes = nt * (nt + 1) // 2 * 128 * 128 * esz + 4 * npad * esz  # lower block-triangle written + points read
                out["roofline_kbuild"] = {"bound": "hbm", "kernel": "kbuild_kernel<%s>" % gemm_t, "bytes": kb_bytes,
                                          "avg_launch_ms": ms, "launches_timed": reps, "achieved": kb_bytes / (ms * 1e-3) / 1e9,
                                          "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": kb_bytes / (ms * 1e-3) / HBM_PEAK,
                                          "in_create_ms": st["t_kbuild_ms"]}
                del Kbuf
                qb = 8192
                fab = torch.zeros(3 * qb, dtype=torch.float64, device=dev)
                fab[:qb] = 0.3
                fab[qb:2 * qb] = -0.1  # a representative per-query fit (values do not change the work)
                fab[2 * qb:] = 0.01
                np_rows = min(npad, (n_train + 127) // 128 * 128)
                Kq = torch.empty(qb * npad, dtype=tdt, device=dev)
                # the library forms k - fit in fp64 from the fp64 points for the thin plate (and for fp64 models), in fp32 from
                # the centred fp32 points for the exponential kernels: time the kernel the timed step actually ran
                op64 = args.kernel == "thinplate" or prec == gpx.F64
                if op64:
                    p64 = [t_.double() for t_ in pts]
                    kq = lambda: gpx._check(L.gpx_dev_kqp(C.byref(kern), stage_prec, n_train, npad, vp(p64[0]), vp(p64[1]), vp(p64[2]),
                                                          qb, vp(qx), vp(qy), vp(qz), vp(fab) if prec != gpx.F64 else None, vp(Kq),
                                                          C.c_void_p(strm.cuda_stream)))
                else:
                    cen = torch.zeros(8, dtype=torch.float64, device=dev)
                    kq = lambda: gpx._check(L.gpx_dev_kqp_f32(C.byref(kern), n_train, npad, vp(pts[0]), vp(pts[1]), vp(pts[2]), vp(cen),
                                                              qb, vp(qx), vp(qy), vp(qz), vp(fab), vp(Kq), C.c_void_p(strm.cuda_stream)))
                ms = timed(kq)
                kq_bytes = qb * np_rows * esz + 3 * npad * 8 + qb * 48  # Kqp written + points + queries and fit read
                out["roofline_kqp"] = {"bound": "hbm", "kernel": "kqp_kernel<%s, formed in %s> (kernel operand of one variance batch of %d queries)" % (gemm_t, "fp64" if op64 else "fp32", qb),
                                       "bytes_per_launch": kq_bytes, "avg_launch_ms": ms, "launches_timed": reps,
                                       "achieved": kq_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                       "frac": kq_bytes / (ms * 1e-3) / HBM_PEAK,
                                       "in_evaluate_ms_per_launch": (st["t_var_kqp_ms"] / st["var_gemm_launches"]) if want_v and st["var_gemm_launches"] > 0 else None}
                del Kq
            except Exception as e:
                out["roofline_kbuild"] = {"error": str(e)}
        if st["t_mean_ms"] > 0:
            pairs = float(nq_local) * npad
            rate = pairs / (st["t_mean_ms"] * 1e-3)
            lane_rate = 256 * 4 * 16 * 2.4e9  # lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes at 2.4 GHz
            out["roofline_mean"] = {
                "bound": "valu", "kernel": "predict_kernel<f64> (mean; fp64 in every mode)",
                "pair_evals": pairs, "ms": st["t_mean_ms"], "achieved": rate / 1e9, "unit": "G pair-evals/s",
                "valu_instr_per_pair": MEAN_VALU_PER_PAIR[args.kernel] if args.kernel in MEAN_VALU_PER_PAIR else None,
                "frac": (rate * MEAN_VALU_PER_PAIR[args.kernel] / lane_rate) if args.kernel in MEAN_VALU_PER_PAIR else None,
                "note": "VALU-issue fraction = pair-evals/s x VALU instructions per pair (counted in the ISA of the inner loop, "
                        "profiles/r02_predict_isa.txt) / 3.93e13 lane-instructions/s; HBM traffic is 32 B per query"}
        if world == 1 and want_v and args.precision == "f32" and not args.no_fast_mode and not shard:
            # informative only, measured AFTER the timed region and never part of `value`: the same step with the
            # variance contraction on the fp16 matrix cores (hi/lo operand halves, GPX_PREC_F32_SPLIT)
            try:
                def split_step():
                    ms = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F32_SPLIT, prepare_variance=True,
                                   device=local_rank)
                    ms.evaluate_device(nq_local, qx.data_ptr(), qy.data_ptr(), qz.data_ptr(), f.data_ptr(), v.data_ptr())
                    ms.sync()
                    return ms
                split_step().close()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ms = split_step()
                dt = time.perf_counter() - t1
                sst = ms.stats
                ms.close()
                f_split, v_split = f.clone(), v.clone()
                out["fast_mode"] = {
                    "precision": "f32split (3 fp16 MFMA products on hi/lo halves, fp32 accumulation; opt-in)",
                    "ms_per_step": dt * 1e3, "value": nq / dt, "unit": "query-points/s",
                    "variance_gemm_avg_launch_ms": sst["t_var_gemm_ms"] / max(1, sst["var_gemm_launches"]),
                    "variance_accuracy": "hi halves on a shared quantum per MFMA k-group (exact fixed-point product sums); "
                                         "measured on this run's queries against the fp64 pipeline: fast_mode.accuracy "
                                         "(the native fp32 path: accuracy)"}
            except Exception as e:  # never let the extra line break the contract line
                out["fast_mode"] = {"error": str(e)}
        if world == 1 and want_v and args.precision == "f32" and not args.no_fast_mode and not shard:
            # informative only, after the timed region, never `value`: the same step in the reference's own arithmetic
            # (fp64 throughout), driver-visible: ms per step and the fp64 variance GEMM against the fp64 MFMA peak
            try:
                def f64_step():
                    m64 = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F64, prepare_variance=True, device=local_rank)
                    m64.evaluate_device(nq_local, qx.data_ptr(), qy.data_ptr(), qz.data_ptr(), f.data_ptr(), v.data_ptr())
                    m64.sync()
                    return m64
                f64_step().close()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sts = []
                for _ in range(2):
                    m64 = f64_step()
                    sts.append(m64.stats)
                    m64.close()
                dt = (time.perf_counter() - t1) / 2
                s64 = {k: float(np.mean([s_[k] for s_ in sts])) for k in sts[0]}
                l64 = max(1.0, s64["var_gemm_launches"])
                a64 = float(n_train) ** 2 * (nq_local / l64) / (s64["t_var_gemm_ms"] / l64 * 1e-3) / 1e12
                out["f64"] = {"what": "the same step with GPX_PREC_F64 (the reference's arithmetic), 2 steps after 1 warm-up",
                              "ms_per_step": dt * 1e3, "value": nq / dt, "unit": "query-points/s",
                              "variance_gemm": {"achieved": a64, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                "frac": a64 / PEAK_F64_MFMA_TFLOPS, "avg_launch_ms": s64["t_var_gemm_ms"] / l64},
                              "stages_ms": {k: s64[k] for k in ("t_kbuild_ms", "t_factor_ms", "t_solve_ms", "t_inverse_ms",
                                                                "t_mean_ms", "t_var_ms")},
                              "alpha_residual": s64["alpha_residual"],
                              "whole_ldlt": {"what": "N^3/3 flop / t_factor_ms, fp64", "ms": s64["t_factor_ms"],
                                             "achieved": (n_train ** 3 / 3.0) / (s64["t_factor_ms"] * 1e-3) / 1e12,
                                             "frac": (n_train ** 3 / 3.0) / (s64["t_factor_ms"] * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS}}
                # accuracy of the TIMED workload: f, v now hold the fp64 pipeline's results for the same 2^20 queries
                out["accuracy"] = accuracy_record(torch, f_timed, v_timed, f, v, k0)
                if "fast_mode" in out and "error" not in out["fast_mode"]:
                    out["fast_mode"]["accuracy"] = accuracy_record(torch, f_split, v_split, f, v, k0)
            except Exception as e:
                out["f64"] = {"error": str(e)}
        if world == 1 and want_v and not shard and not args.no_fast_mode and model[0] is not None:
            # informative only (never `value`): the same prediction through the HOST-pointer entry gpx_model_evaluate,
            # i.e. including the PCIe copies of 3 x 8 B in and 2 x 8 B out per query (1 MiB-query slices)
            try:
                hq = [t.cpu().numpy() for t in (qx, qy, qz)]
                model[0].evaluate(hq[0][:4096], hq[1][:4096], hq[2][:4096], want_v=True)
                t1 = time.perf_counter()
                model[0].evaluate(hq[0], hq[1], hq[2], want_v=True)
                dt = time.perf_counter() - t1
                out["host_api"] = {"what": "gpx_model_evaluate(f, v) on host arrays of the trained model (predict only, PCIe copies included)",
                                   "ms": dt * 1e3, "value": nq_local / dt, "unit": "query-points/s"}
            except Exception as e:
                out["host_api"] = {"error": str(e)}
        if world == 1 and not shard and not args.no_fast_mode and not args.no_configs:
            try:
                if model[0] is not None:
                    model[0].close()
                    model[0] = None
                gpx.trim()
                out["configs"] = extra_configs(torch, gpx, ds, sharding, dev, local_rank)
            except Exception as e:
                out["configs"] = {"error": str(e)}
            try:  # the node-shaped end-to-end number on the headline model (informative, after the timed region)
                full_variance_ms = st["t_var_ms"] * (GRID ** 3 / float(nq_local)) if want_v else None
                out["configs"]["C3_surface"] = surface_config(torch, gpx, ds, dev, local_rank, kern, (x, y, z, lab, s2), n_train)
                out["configs"]["C3_surface"]["full_variance_equivalent_ms"] = full_variance_ms
            except Exception as e:
                out["configs"]["C3_surface"] = {"error": str(e)}
            try:  # ONE evaluate / sampleSurface call cut over replicas from the C boundary: two replicas on THIS ordinal
                out["configs"]["sharded_call"] = sharded_call_config(gpx, ds, local_rank)
            except Exception as e:
                out["configs"]["sharded_call"] = {"error": str(e)}
            try:
                out["configs"]["C1_node"] = node_pattern_config()
            except Exception as e:
                out["configs"]["C1_node"] = {"error": str(e)}
            try:
                out["roofline_small"] = small_model_roofline(torch, gpx, ds, dev, local_rank)
            except Exception as e:
                out["roofline_small"] = {"error": str(e)}
            try:
                out["roofline_small64"] = small_model_roofline64(torch, gpx, ds, dev, local_rank)
            except Exception as e:
                out["roofline_small64"] = {"error": str(e)}
        if mg:
            out.setdefault("configs", {}).update(mg)
        if "roofline" in out:
            # the driver's record keeps the `roofline` object and only the NAMES of the other top-level keys: the secondary
            # rooflines are therefore nested here as well (compact: the figures, not the prose); the top-level copies stay
            def compact(r):
                if not isinstance(r, dict):
                    return r
                keep = ("bound", "achieved", "peak", "unit", "frac", "ms", "avg_launch_ms", "kernel_ms", "issue_bound_frac",
                        "issue_bound", "error")
                c = {k: r[k] for k in keep if k in r}
                if "whole_ldlt" in r:
                    c["whole_ldlt_frac"] = r["whole_ldlt"]["frac"]
                if "sizes" in r:
                    c["sizes"] = {n: compact(v_) for n, v_ in r["sizes"].items()}
                return c
            out["roofline"]["stages"] = {name: compact(out["roofline_" + name]) for name in
                                         ("factor", "inverse", "kbuild", "kqp", "mean", "small", "small64") if "roofline_" + name in out}
            if "f64" in out and "whole_ldlt" in out["f64"]:
                out["roofline"]["stages"]["factor_f64"] = {"frac": out["f64"]["whole_ldlt"]["frac"], "ms": out["f64"]["whole_ldlt"]["ms"]}
            cfg = out.get("configs") or {}
            out["roofline"]["configs_ms"] = {k: (v_.get("ms_per_step", v_.get("ms", v_.get("ms_per_grid"))) if isinstance(v_, dict) else None)
                                             for k, v_ in cfg.items()}
            out["roofline"]["fallbacks"] = out["fallbacks"]
        if world == 1:
            out["eigen_on_box"] = eigen_on_box()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n_train, nq, args.kernel, kpar)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if model[0] is not None:
        model[0].close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
