#!/usr/bin/env python3
"""Golden cell-array file written by the reference's own converter statements (oracle/_ref/dat_harness = hdf42bin.f90 with
its HDF4 input replaced; make -C oracle ref).  Stores the file's bytes and its inputs: tests/golden/cellarray_dat.npz."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "dat_harness")


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build oracle/_ref/dat_harness first: make -C oracle ref")
    amr = np.load(os.path.join(HERE, "amr6_scattered_level2.npz"))
    n, level = int(amr["n"]), amr["level"].astype(np.int32)
    rng = np.random.default_rng(77)
    ncell = level.size
    fields = {k: (10 ** rng.uniform(-8, -2, ncell)).astype(np.float32) for k in ("HI", "HeI", "HeII", "temperature", "density")}
    box = 1200.0 * float(np.float32(1.0e3)) * float(np.float32(3.08568025e18))
    with tempfile.TemporaryDirectory() as tmp:
        case = os.path.join(tmp, "case.bin")
        with open(case, "wb") as f:
            f.write(struct.pack("<2i", n, ncell))
            f.write(struct.pack("<d", box))
            f.write(level.astype("<i4").tobytes())
            for k in ("HI", "HeI", "HeII", "temperature", "density"):
                f.write(fields[k].astype("<f4").tobytes())
        res = subprocess.run([HARNESS, case, tmp + "/"], capture_output=True, text=True)
        out = os.path.join(tmp, "cellArray.dat")
        if res.returncode != 0 or not os.path.exists(out):
            raise RuntimeError(res.stdout[-500:] + res.stderr[-500:])
        raw = np.fromfile(out, dtype=np.uint8)
    path = os.path.join(HERE, "cellarray_dat.npz")
    np.savez_compressed(path, n=n, level=level, box=box, dat_bytes=raw, **fields)
    print(f"cellarray_dat: {os.path.getsize(path) / 1024:.0f} KiB, file of {raw.size} bytes")


if __name__ == "__main__":
    main()
