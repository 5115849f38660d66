"""Round 3: how long ONE brick takes.  One group of directions of one izone on one frequency group, one stream: the first stages of the
sweep hold 1, 3, 6, ... bricks, so the kernel trace's duration by launch width gives the latency of a brick alone (what the
dependency path of a one-group rank is made of) and what more bricks per launch add.
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/experiments/r03_latency.py run <form> <chunk> <ndir>
  python3 tools/experiments/r03_latency.py read DIR"""
import sys, glob, csv, collections
import numpy as np


def run(form, chunk, ndir, ablate=0):
    import radiativetransfer_amd as rt
    from radiativetransfer_amd import synthetic
    n = 256
    kappa, uvb, box = synthetic.uniform_workload(n, 1, seed=1, tau_median=0.1)
    phi, theta, w = np.array([0.3, 0.5, 0.7]), np.array([0.3, 0.35, 0.4]), np.array([0.3, 0.3, 0.4])  # one octant, one leading axis
    pick = list(range(ndir))
    with rt.DiffuseTransfer() as e:
        e.set_uniform_grid(n, box)
        e.set_option("team", form)
        e.set_option("chunk", chunk)
        e.set_option("lanes", 1)
        e.set_option("ablate", ablate)
        e.set_opacity(kappa)
        for _ in range(3):
            e.transport(phi[pick], theta[pick], w[pick], uvb)
        print("directions", len(pick), "form", e.counter("brick_form"), "groups", e.counter("brick_groups"))


def read(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        if "brick" not in r["Kernel_Name"]:
            continue
        wg = int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) // int(r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r["Workgroup_Size"])
        by[wg].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for wg in sorted(by):
        v = np.array(by[wg])
        print(f"{wg:6d} workgroups: {len(v):4d} launches, duration median {np.median(v):7.1f} us, min {v.min():7.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*[int(x) for x in sys.argv[2:]])
    else:
        read(sys.argv[2])
