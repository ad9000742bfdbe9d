"""Turn the summary of `PMC=16 bash tools/prof_bench.sh <tag>` (three rocprofv3 --pmc passes on the dominant kernel's dominant launch)
into profiles/pmc_traffic.json, stamped with the build of the library that was measured -- bench.py reports `roofline.traffic` from
that record only when the stamp equals the running library's.   usage: python tools/pmc_record.py gpurun_out/<tag>/summary.txt"""
import ast
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals, us = {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) launches (\d+) avg_us ([0-9.]+) (\{.*\})", line.strip())
    if m:
        d = ast.literal_eval(m.group(4))
        vals.update({k: float(v) for k, v in d.items()})
        us[m.group(1)] = float(m.group(3))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
nbytes = int((2 * fetch + write) * 1024)
stamp = open(os.path.join(ROOT, "council-gan_amd", "lib", "libcouncilgan_hip.so.stamp")).read().strip()[:16]
cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
t = us.get("SQ_VALU_MFMA_BUSY_CYCLES", us.get("FETCH_SIZE"))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
rec = json.load(open(path))
k = "conv_fwd_x3w_kernel<256,256,fast>"
rec[k].update({"fetch_size_kib": int(fetch), "write_size_kib": int(write), "bytes_per_launch": nbytes,
               "ratio": round(nbytes / rec[k]["algorithmic_bytes_per_launch"], 2), "avg_us_under_profiler": us.get("FETCH_SIZE"),
               "matrix_pipe": "SQ_VALU_MFMA_BUSY_CYCLES %.4g of GRBM_GUI_ACTIVE/8 x 1024 SIMDs = %.4g x 1024 cycles: %.0f %% busy at %.2f GHz"
                              % (vals["SQ_VALU_MFMA_BUSY_CYCLES"], cyc, 100 * busy, cyc / t / 1e3),
               "build_stamp": stamp, "source": "gpurun_out/%s" % os.path.relpath(os.path.abspath(sys.argv[1]), os.path.join(ROOT, "gpurun_out")),
               "how": "AB_ACT=0 PMC=16 bash tools/prof_bench.sh: three rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* / GRBM_*), --kernel-trace "
                      "only; gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes -> x 2 (MI355X_MICROARCH.md); written by tools/pmc_record.py"})
json.dump(rec, open(path, "w"), indent=1)
print(json.dumps(rec[k], indent=1))
