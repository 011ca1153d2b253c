#!/bin/bash
# PMC counters of the ablated cross_kernel builds (run on the GPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in FULL NO_ROWROLE NO_MASKROLE; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ABLATE_$v -Wno-unused-result \
    -o gpurun_out/ablate/ab_$v tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip 2>/dev/null
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d gpurun_out/ablate/pmc_$v -- ./gpurun_out/ablate/ab_$v > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA \
    --output-format csv -d gpurun_out/ablate/pmc2_$v -- ./gpurun_out/ablate/ab_$v > /dev/null 2>&1
done
python3 - <<'PY'
import pandas as pd, glob
for v in ("FULL","NO_ROWROLE","NO_MASKROLE"):
    out = {}
    for d in (f"gpurun_out/ablate/pmc_{v}", f"gpurun_out/ablate/pmc2_{v}"):
        f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
        df = pd.read_csv(f)
        k = df[df["Kernel_Name"].str.contains("cross_kernel")]
        out.update(k.groupby("Counter_Name")["Counter_Value"].mean().to_dict())
    print(v, {a: f"{b:.3g}" for a, b in sorted(out.items())})
PY
rm -rf gpurun_out/ablate
