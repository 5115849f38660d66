#!/usr/bin/env python3
"""Golden vectors for the group cross-sections: the reference's uvbBetaTable (uvbBetaTable.f90) called by
oracle/_ref/uvb_harness (make -C oracle ref) for a few power-law slopes.  Writes tests/golden/uvb_beta_table.npz."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "uvb_harness")


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build oracle/_ref/uvb_harness first: make -C oracle ref")
    alphas = np.array([[1.8, 1.5, 1.2], [5.0, 5.0, 5.0], [0.5, 2.25, 1.75], [1.5, 1.5, 1.5]])
    out, uni = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for a in alphas:
            path = os.path.join(tmp, "o.bin")
            res = subprocess.run([HARNESS, repr(float(a[0])), repr(float(a[1])), repr(float(a[2])), path], capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(res.stdout + res.stderr)
            raw = np.fromfile(path, dtype="<f8")
            out.append(raw[:27].reshape(3, 3, 3))   # [beta|ksi|gamma][group][24,25,26 | HI,HeI,HeII]
            uni.append(raw[27:39].reshape(2, 2, 3))   # uniformTable(alpha1, alpha2): [ksi|gamma][quasar, stellar][3]
            coll = raw[39:39 + 768].reshape(2, 64, 6)  # Fortran kout(6,64,2): [case A, case B][temperature][k1..k6]
            coll_t = raw[39 + 768:]                     # the temperatures as the harness formed them
    out, uni = np.array(out), np.array(uni)
    np.savez_compressed(os.path.join(HERE, "uvb_beta_table.npz"), alpha=alphas, beta=out[:, 0], ksi=out[:, 1], gamma=out[:, 2],
                        uniform_ksi=uni[:, 0], uniform_gamma=uni[:, 1], coll_temperature=coll_t,
                        coll_rates=coll)
    print("uvb_beta_table:", out.shape)


if __name__ == "__main__":
    main()
