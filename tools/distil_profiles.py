"""Distil gpurun_out/prof_<tag>/ (made by tools/collect_profiles.sh on the GPU box) into the small,
committed files under profiles/:  <tag>_kernel_stats.csv, <tag>_pmc.json, and pmc_traffic.json
(the file bench.py reads for roofline.traffic)."""
import glob
import json
import sys
from pathlib import Path

import pandas as pd

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)


def one(pattern):
    # gpurun merges new files next to those of earlier calls: take the newest
    f = sorted(glob.glob(str(src / pattern)), key=lambda x: Path(x).stat().st_mtime)
    return f[-1] if f else None


ks = one("kt/*/*kernel_stats.csv")  # the headline configuration alone
stats = pd.read_csv(ks)
stats.to_csv(dst / f"{tag}_kernel_stats.csv", index=False)
print(stats[["Name", "Calls", "AverageNs", "Percentage"]].to_string())
ka = one("kt_all/*/*kernel_stats.csv")  # every configuration bench.py runs (kernel names are shared between some)
if ka:
    pd.read_csv(ka).to_csv(dst / f"{tag}_kernel_stats_all_configs.csv", index=False)


def counters(sub):
    f = one(f"{sub}/*/*counter_collection.csv")
    if not f:
        return {}
    df = pd.read_csv(f)
    out = {}
    for name, g in df.groupby("Kernel_Name"):
        short = name.split("(")[0].replace("void bh::", "")
        # per launch: every counter has one row per dispatch (summed over XCDs by rocprofv3)
        out[short] = {c: float(v) for c, v in g.groupby("Counter_Name")["Counter_Value"].mean().items()}
        out[short]["dispatches"] = int(g["Dispatch_Id"].nunique())
    return out


pmc = {"fetch": counters("fetch"), "write": counters("write"), "sq": counters("sq"), "tcc": counters("tcc"),
       "sq_all_configs": counters("sq_all")}
(dst / f"{tag}_pmc.json").write_text(json.dumps(pmc, indent=1, sort_keys=True))

# HBM traffic of the hot kernel per launch, corrected as MI355X_MICROARCH.md "HBM" prescribes:
# FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
# wide coalesced (16 B/lane) streaming read -> double it; WRITE_SIZE is exact for 16-B/lane stores.
hot = [k for k in pmc["fetch"] if k.startswith("packed_cross_kernel<1, 4")]  # TT=1, MW=4: the headline launch
hot = hot or [k for k in pmc["fetch"] if k.startswith("cross_kernel<1, 4")]
hot = hot or [k for k in pmc["fetch"] if k.startswith("cross_kernel")]
if hot:
    k = hot[0]
    fetch_kib = pmc["fetch"][k]["FETCH_SIZE"]
    write_kib = pmc["write"].get(k, {}).get("WRITE_SIZE", 0.0)
    traffic = {
        "kernel": k,
        "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
        "fetch_bytes_corrected": 2 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
        "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B for 16-B/lane streams), KiB -> bytes; "
                      "for packed_cross_kernel the reads are 2-byte genotype words (uncalibrated width per the guide) and "
                      "16-B/lane effect vectors, together under a third of the total either way; WRITE_SIZE is exact "
                      "for its 16-B/lane observation stores",
        "profile": f"profiles/{tag}_pmc.json",
    }
    (dst / "pmc_traffic.json").write_text(json.dumps(traffic, indent=1))
    print(json.dumps(traffic, indent=1))
