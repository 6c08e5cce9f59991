"""Turns the rocprofv3 PMC passes of `python3 bench.py ...` into profiles/conv_traffic.json.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...   (separate pass: TCC slots)
  python profiles/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write

Corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B as reported by
rocprofv3's derived metric (TCC_EA0_RDREQ x 64 B / 1024); on gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B per lane) streaming reads, so it is doubled; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter, kernel):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    kernel = "spconv_fwd"        # spconv_fwd2_kernel, spconv_fwd3_kernel, spconv_fwd_kernel: the family bench.py times
    fetch, nf = per_launch(fetch_dir, "FETCH_SIZE", kernel)
    write, nw = per_launch(write_dir, "WRITE_SIZE", kernel)
    out = {"kernel": kernel, "launches_sampled": [nf, nw], "FETCH_SIZE_per_launch_raw": fetch,
           "WRITE_SIZE_per_launch_raw": write, "unit_bytes": 1024, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": (fetch * 2.0 + write) * 1024.0,
           "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
