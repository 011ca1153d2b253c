from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = ["golden_small_shared", "golden_small_independent", "golden_odd61", "golden_sample", "golden_zero_recomb"]


def _unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(tuple(shape)).astype(np.uint8)


def load_case(name):
    z = np.load(GOLDEN / f"{name}.npz")
    return dict(
        pool=_unpack(z["pool_bits"], z["pool_shape"]), idx=z["idx"], r=z["r"], effects=z["effects"],
        seed=int(z["seed"]), rng_mode=int(z["rng_mode"]), out=_unpack(z["out_bits"], z["out_shape"]),
        masks=_unpack(z["masks_bits"], z["masks_shape"]), gebv32=z["gebv32"], gebv64=z["gebv64"],
        new_key=z["new_key"],
    )
