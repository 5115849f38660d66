// ftte_geometry.h -- host-side ray geometry (see ftte_geometry.cpp)
#pragma once
#include <cstdint>

#include "../../include/ftte.h"

namespace ftte {

extern const double kPi, kHalfPi, kTwoPi;

// Which sweep index (0 = i the march axis, 1 = j, 2 = k) feeds each storage index
// (0 = icell, 1 = jcell, 2 = kcell) for an izone, and whether it is mirrored (n+1-x).
struct ZoneMap {
    int src[3];
    bool mirror[3];
};
bool zone_map(int izone, ZoneMap *m);

int rotate_indices(int i, int j, int k, int nx, int ny, int nz, int izone, int *ic, int *jc, int *kc);
int pix2ang_nest(int nside, int64_t ipix, double *phi, double *theta);
int fold_direction(double phi_in, double theta_in, double *phi, double *theta, int *izone);
int set_pattern(ftte_pattern *P, double phi, double theta);
int layer_patterns(int n, double phi, double theta, ftte_pattern *layers);

} // namespace ftte
