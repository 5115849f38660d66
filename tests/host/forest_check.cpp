// forest_check.cpp -- CPU test of the product's AMR planner (csrc/ftte_amr.cpp): evaluates the segment forest it
// builds, depth by depth, with the device arithmetic (csrc/ftte_math.h) -- i.e. exactly what amr_level_kernel and
// amr_combine_kernel do, serially -- and writes J, to be compared bit for bit with the oracle's tree sweep.
//
//   forest_check <case.bin> <out.bin>
// case.bin: int32 n, ncell, ndir, nnu; double box; double uvb[nnu]; int32 level[ncell]; double kappa[nnu][ncell];
//           double phi[ndir], theta[ndir], w[ndir]       out.bin: double J[nnu][ncell]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../radiativetransfer_amd/csrc/ftte_amr.h"
#include "../../radiativetransfer_amd/csrc/ftte_geometry.h"
#include "../../radiativetransfer_amd/csrc/ftte_math.h"

using namespace ftte;

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[4];
    if (std::fread(hdr, 4, 4, f) != 4) return 2;
    const int n = hdr[0], ncell = hdr[1], ndir = hdr[2], nnu = hdr[3];
    double box;
    std::vector<double> uvb(nnu), kappa((size_t)nnu * ncell), phi(ndir), theta(ndir), w(ndir);
    std::vector<int32_t> level(ncell);
    bool ok = std::fread(&box, 8, 1, f) == 1 && std::fread(uvb.data(), 8, nnu, f) == (size_t)nnu &&
              std::fread(level.data(), 4, ncell, f) == (size_t)ncell &&
              std::fread(kappa.data(), 8, kappa.size(), f) == kappa.size() && std::fread(phi.data(), 8, ndir, f) == (size_t)ndir &&
              std::fread(theta.data(), 8, ndir, f) == (size_t)ndir && std::fread(w.data(), 8, ndir, f) == (size_t)ndir;
    std::fclose(f);
    if (!ok) return 2;

    AmrTree tree;
    const std::string terr = tree.build(n, ncell, level.data());
    if (!terr.empty()) { std::fprintf(stderr, "%s\n", terr.c_str()); return 3; }

    static const ftte_consts K = FTTE_CONSTS_INIT;
    std::vector<double> J((size_t)nnu * ncell, 0.0), Iout((size_t)3 * ncell * nnu), mean((size_t)3 * ncell * nnu);
    for (int d = 0; d < ndir; ++d) {
        double p, t;
        int z;
        if (fold_direction(phi[d], theta[d], &p, &t, &z)) return 4;
        AmrForest F;
        std::string err;
        if (build_forest(tree, p, t, z, box, &F, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 5; }
        for (size_t depth = 0; depth + 1 < F.depth_off.size(); ++depth)
            for (int64_t e = F.depth_off[depth]; e < F.depth_off[depth + 1]; ++e) {
                const int seg = F.order[(size_t)e];
                for (int g = 0; g < nnu; ++g) {
                    double I;
                    if (F.up[seg] < 0) I = uvb[g];
                    else {
                        I = Iout[(size_t)F.up[seg] * nnu + g];
                        if (F.up2[seg] >= 0) I = 0.5 * (I + Iout[(size_t)F.up2[seg] * nnu + g]);
                    }
                    const double m = ftte_segment(&K, &I, kappa[(size_t)g * ncell + seg / 3] * F.dpath[seg]);
                    Iout[(size_t)seg * nnu + g] = I;
                    mean[(size_t)seg * nnu + g] = m;
                }
            }
        for (int64_t c = 0; c < ncell; ++c)
            for (int g = 0; g < nnu; ++g) {
                double acc = mean[(size_t)(3 * c) * nnu + g];
                int nseg = 1;
                if (F.up[3 * c + 1] != AmrForest::kInactive) { acc += mean[(size_t)(3 * c + 1) * nnu + g]; ++nseg; }
                if (F.up[3 * c + 2] != AmrForest::kInactive) { acc += mean[(size_t)(3 * c + 2) * nnu + g]; ++nseg; }
                J[(size_t)g * ncell + c] += ftte_cell_mean(acc, nseg, w[d]);
            }
    }
    f = std::fopen(argv[2], "wb");
    std::fwrite(J.data(), 8, J.size(), f);
    std::fclose(f);
    return 0;
}
