"""Raw FETCH_SIZE / WRITE_SIZE (KiB per launch, as rocprofv3 reports them) of every kernel of one bench.py configuration:
the calibration runs behind bench.py's FETCH_FACTORS (variants of a kernel that leave one kind of read out are built
with tools/ab_variants.sh MODE=tool TOOL="tools/pmc_raw.py c5").   usage: pmc_raw.py c2|c3|c5 [steps]"""
import csv
import glob
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
steps = sys.argv[2] if len(sys.argv) > 2 else "40"
cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", steps, "--warmup", "4"] + (
    ["--headline-only"] if cfg == "c2" else ["--only", cfg])
rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
tmp = tempfile.mkdtemp(prefix="pmc_raw_", dir="/tmp")
res = defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    subprocess.run([rocprof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", f"{tmp}/{ctr}", "-o", "p", "--"] + cmd,
                   cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    acc = defaultdict(list)
    for f in glob.glob(f"{tmp}/{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == ctr:
                acc[row["Kernel_Name"].split("(")[0].replace("void bh::", "")].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k][ctr] = (sum(v) / len(v), len(v))
shutil.rmtree(tmp, ignore_errors=True)
for k, d in sorted(res.items()):
    f, w = d.get("FETCH_SIZE", (0, 0)), d.get("WRITE_SIZE", (0, 0))
    if f[1] >= 4:
        print(f"pmc_raw {cfg} {k:55s} FETCH_SIZE {f[0]:12.1f} KiB  WRITE_SIZE {w[0]:12.1f} KiB  ({f[1]} launches)")
