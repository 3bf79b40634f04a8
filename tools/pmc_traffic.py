"""HBM-side traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench run, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (separate passes; FETCH_SIZE counts 64 B per 128-B request of a
wide streaming read: doubled; both counters are in kilobytes), written to profiles/traffic_<kernel>.json together with the hash
of the kernel source it was measured on -- bench.py reports `roofline.traffic` only while that hash still matches.

usage (on the GPU box, from the repository root):  python tools/pmc_traffic.py [bench.py arguments ...]
"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get("PMC_KERNEL", "k_factor_product_x3")
SOURCE = os.environ.get("PMC_SOURCE", "nmfgpu_amd/csrc/kernels_x3.hip")
OUT = os.environ.get("PMC_OUT", "profiles/traffic_factor_product_x3.json")


def one_pass(counter, extra):
    d = os.path.join(ROOT, "gpurun_out", f"pmc_{counter}")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
           "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-events", *extra]
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, timeout=170, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                per[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per.items()}, {k: len(v) for k, v in per.items()}


def main():
    extra = sys.argv[1:]
    fetch, nf = one_pass("FETCH_SIZE", extra)
    write, _ = one_pass("WRITE_SIZE", extra)
    forms = {}
    for k in fetch:
        forms[k] = {"fetch_size_kb": fetch[k], "write_size_kb": write.get(k, 0.0), "launches": nf[k],
                    "hbm_bytes_per_launch": 2.0 * fetch[k] * 1024 + write.get(k, 0.0) * 1024}
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in forms.values()) / max(1, sum(v["launches"] for v in forms.values()))
    out = {"kernel": KERNEL, "hbm_bytes_per_launch": total, "forms": forms,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 20 --warmup 3`; "
                     "bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (gfx950: FETCH_SIZE tallies 64 B per 128-B request), average per launch over all forms",
           "source": SOURCE, "source_sha256": hashlib.sha256(open(os.path.join(ROOT, SOURCE), "rb").read()).hexdigest(),
           "bench_arguments": extra}
    # (gpurun brings back gpurun_out/ only: keep a copy there to commit as profiles/... from the CPU container)
    for path in (os.path.join(ROOT, OUT), os.path.join(ROOT, "gpurun_out", os.path.basename(OUT))):
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
