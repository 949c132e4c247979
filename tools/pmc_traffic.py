"""Turns the FETCH_SIZE / WRITE_SIZE rocprofv3 passes of `bench.py` into profiles/r2_pmc_traffic.json (read back by bench.py's
roofline.traffic as long as the kernel source and the launch shape match).

usage: pmc_traffic.py <fetch_csv> <write_csv> <calib_fetch_csv> <out.json> <n1> <n2_local> <m>
The counters are reported in KB per dispatch.  Calibration: MI355X_MICROARCH.md says FETCH_SIZE counts 128-B requests at
64 B for wide (16 B/lane) streaming loads, i.e. reads 1/2 of the bytes; our kernels load 8 B/lane, so the factor is
measured here on vg_sumsq_kernel reading a buffer of known size (calib pass), not assumed."""
import collections, csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_sha


def per_kernel(f, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
calib, _ = per_kernel(sys.argv[3], "FETCH_SIZE")
n1, n2, m = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
cal_bytes = float(os.environ.get("CALIB_BYTES", 0))
k = next(n for n in ("vg_gemm_project_deep_kernel", "vg_gemm_gram_project_wide_kernel", "vg_gemm_gram_project_kernel") if n in fetch)
ck = "vg_sumsq_kernel"
factor = cal_bytes / (calib[ck] * 1024.0)
alg = 8 * (n1 * n2 + 2 * m * n2 + 2 * m * n1)
out = {
    "kernel": k, "dispatches_averaged": nf[k], "shape": [n1, n2, m], "source_sha": source_sha(),
    "FETCH_SIZE_KB_reported": fetch[k], "WRITE_SIZE_KB_reported": write[k],
    "fetch_correction_factor": factor,
    "calibration": f"{ck}: reads {cal_bytes:.0f} B, FETCH_SIZE reported {calib[ck]:.1f} KB",
    "hbm_read_bytes": fetch[k] * 1024.0 * factor, "hbm_write_bytes": write[k] * 1024.0,
    "bytes": fetch[k] * 1024.0 * factor + write[k] * 1024.0,
    "algorithmic_bytes": alg, "bytes_over_algorithmic": (fetch[k] * 1024.0 * factor + write[k] * 1024.0) / alg,
    "other_kernels_KB": {kk: {"FETCH_SIZE_KB_reported": fetch.get(kk), "WRITE_SIZE_KB_reported": write.get(kk)}
                         for kk in sorted(set(fetch) | set(write)) if kk.startswith("vg_")},
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k_: v for k_, v in out.items() if k_ != "other_kernels_KB"}))
