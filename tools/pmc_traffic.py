"""Turns the FETCH_SIZE / WRITE_SIZE rocprofv3 passes of `bench.py` into profiles/r1_pmc_traffic.json.

usage: pmc_traffic.py <fetch_dir> <write_dir> <calib_fetch_dir> <out.json>
The counters are reported in KB per dispatch.  Calibration: MI355X_MICROARCH.md says FETCH_SIZE counts 128-B requests at
64 B for wide (16 B/lane) streaming loads, i.e. reads 1/2 of the bytes; our kernels load 8 B/lane, so the factor is
measured here on vg_sumsq_kernel reading a buffer of known size (calib dir), not assumed."""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
calib, _ = per_kernel(sys.argv[3], "FETCH_SIZE")
cal_bytes = float(os.environ.get("CALIB_BYTES", 0))
k = "vg_gemm_gram_project_wide_kernel" if "vg_gemm_gram_project_wide_kernel" in fetch else "vg_gemm_gram_project_kernel"
ck = "vg_sumsq_kernel"
factor = cal_bytes / (calib[ck] * 1024.0)
out = {
    "kernel": k, "dispatches_averaged": nf[k],
    "FETCH_SIZE_KB_reported": fetch[k], "WRITE_SIZE_KB_reported": write[k],
    "fetch_correction_factor": factor,
    "calibration": f"{ck}: reads {cal_bytes:.0f} B, FETCH_SIZE reported {calib[ck]:.1f} KB",
    "hbm_read_bytes": fetch[k] * 1024.0 * factor, "hbm_write_bytes": write[k] * 1024.0,
    "bytes": fetch[k] * 1024.0 * factor + write[k] * 1024.0,
    "algorithmic_bytes": None,
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
