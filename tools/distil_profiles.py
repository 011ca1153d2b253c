"""Distil gpurun_out/prof_<round>/ (tools/collect_profiles.sh <round>, run on the GPU box) into the committed files
under profiles/: <round>_kernel_stats_{c2,c3,c3lazy,c4,c5,wheat}.csv, <round>_pmc.json and <round>_pmc_traffic.json
(per-kernel memory-side bytes per launch, corrected as MI355X_MICROARCH.md prescribes; bench.py names it as the source of
roofline.traffic when its own live passes fail).   usage: distil_profiles.py r03"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
sys.path.insert(0, str(ROOT))
import bench  # fetch_factor: one table for the line and for the committed file
src, dst = ROOT / "gpurun_out" / f"prof_{TAG}", ROOT / "profiles"


def newest(pattern):
    f = sorted(glob.glob(str(src / pattern), recursive=True), key=lambda x: Path(x).stat().st_mtime)
    return f[-1] if f else None


def short(name):
    return name.split("(")[0].replace("void bh::", "").replace("bh::", "")


for cfg in ("c2", "c3", "c3lazy", "c4", "c5", "wheat"):
    f = newest(f"kt_{cfg}/**/*kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(dst / f"{TAG}_kernel_stats_{cfg}.csv", "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(rows)
        print(cfg, *(f"{short(r['Name'])[:60]:60s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:9.2f} us" for r in rows[:6]), sep="\n  ")


def counters(sub):
    f = newest(f"{sub}/**/*counter_collection.csv")
    out = {}
    if not f:
        return out
    acc = defaultdict(lambda: defaultdict(list))
    disp = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        disp[k].add(r["Dispatch_Id"])
    for k, d in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in d.items()}
        out[k]["dispatches"] = len(disp[k])
    return out


pmc = {f"{kind}_{cfg}": counters(f"{kind}_{cfg}") for cfg in ("c2", "c3", "c5") for kind in ("fetch", "write", "sq")}
(dst / f"{TAG}_pmc.json").write_text(json.dumps(pmc, indent=1, sort_keys=True))

# HBM-side bytes per launch: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a
# wide coalesced (16 B/lane) streaming read -> doubled; WRITE_SIZE is exact for 16-B/lane stores.  The 2-byte genotype
# words and 4-byte table loads of these kernels are widths the guide has not calibrated (small next to the stores).
kernels = []
for cfg in ("c2", "c3", "c5"):
    for k, d in pmc[f"fetch_{cfg}"].items():
        if not any(t in k for t in ("packed_cross_kernel", "packed_long_kernel", "mask_finish", "gebv_bits_kernel", "gebv_bits_regs_kernel",
                                    "gebv_bytes_kernel", "gebv_finish_kernel", "gebv_mfma_kernel", "tail_kernel")):
            continue
        fk = d.get("FETCH_SIZE", 0.0)
        wk = pmc[f"write_{cfg}"].get(k, {}).get("WRITE_SIZE", 0.0)
        # per-kernel calibration of the FETCH_SIZE correction: bench.py fetch_factor / FETCH_FACTORS
        ff = bench.fetch_factor(k)
        kernels.append({"config": cfg, "kernel": k, "FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB_raw": wk, "fetch_factor": ff,
                        "hbm_bytes_per_launch": ff * fk * 1024 + wk * 1024, "dispatches": d.get("dispatches")})
(dst / f"{TAG}_pmc_traffic.json").write_text(json.dumps({
    "correction": "fetch_factor x FETCH_SIZE + WRITE_SIZE, KiB -> bytes (MI355X_MICROARCH.md, HBM: 2 for wide coalesced reads; per-kernel calibrations in bench.py FETCH_FACTORS); separate --pmc passes; memory-side bytes: Infinity-Cache hits are counted",
    "made_by": f"tools/collect_profiles.sh {TAG} + tools/distil_profiles.py {TAG}", "kernels": kernels}, indent=1))
for k in kernels:
    print(f"{k['config']} {k['kernel'][:50]:50s} {k['hbm_bytes_per_launch']/1e6:10.1f} MB per launch")
