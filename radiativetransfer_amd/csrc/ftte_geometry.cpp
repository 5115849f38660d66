// ftte_geometry.cpp -- host-side ray geometry: the O(ndir * n) part of the sweep that the
// reference computes per direction before touching any cell.  Everything here must reproduce
// the reference's binary64 decisions (which octant, which face a ray leaves through), so the
// expressions keep the reference's operation order and its float32-widened literals; citations
// are to /root/reference files.
#include "ftte_geometry.h"

#include <cmath>
#include <cstring>

namespace ftte {

// definitionsModule.f90:8-10.  `pi = 3.141592654` is a default-real literal: the binary64
// constant is the float32 value 3.1415927410125732, and halfPi/twoPi derive from it.
const double kPi = static_cast<double>(3.141592654f);
const double kHalfPi = 0.5 * kPi;
const double kTwoPi = 2.0 * kPi;

// ---------------------------------------------------------------------------------------------
// rotateIndicesModule.f90:14-111 as data: for each izone, which sweep index feeds each storage
// index and whether it is mirrored.  Encoding per storage component: 0/1/2 = i/j/k, +4 = mirrored.
static const unsigned char kZoneTable[12][3] = {
    {0, 1, 2},         //  1 (i, j, k)
    {1, 2, 0},         //  2 (j, k, i)
    {2, 0, 1},         //  3 (k, i, j)
    {0, 2, 1 | 4},     //  4 (i, k, nz+1-j)
    {1, 0, 2 | 4},     //  5 (j, i, nz+1-k)
    {2, 1, 0 | 4},     //  6 (k, j, nz+1-i)
    {0, 1 | 4, 2 | 4}, //  7 (i, ny+1-j, nz+1-k)
    {1, 2 | 4, 0 | 4}, //  8 (j, ny+1-k, nz+1-i)
    {2, 0 | 4, 1 | 4}, //  9 (k, ny+1-i, nz+1-j)
    {0, 2 | 4, 1},     // 10 (i, ny+1-k, j)
    {1, 0 | 4, 2},     // 11 (j, ny+1-i, k)
    {2, 1 | 4, 0},     // 12 (k, ny+1-j, i)
};

bool zone_map(int izone, ZoneMap *m)
{
    if (izone < 1 || izone > 24) return false;
    const unsigned char *row = kZoneTable[(izone - 1) % 12];
    for (int c = 0; c < 3; ++c) {
        m->src[c] = row[c] & 3;
        m->mirror[c] = (row[c] & 4) != 0;
    }
    if (izone > 12) m->mirror[0] = !m->mirror[0]; // zones 13-24: first component reflected
    return true;
}

int rotate_indices(int i, int j, int k, int nx, int ny, int nz, int izone, int *ic, int *jc, int *kc)
{
    ZoneMap m;
    if (!zone_map(izone, &m)) return -1;
    const int in[3] = {i, j, k};
    const int ext[3] = {nx, ny, nz};
    int out[3];
    for (int c = 0; c < 3; ++c) out[c] = m.mirror[c] ? ext[c] + 1 - in[m.src[c]] : in[m.src[c]];
    *ic = out[0];
    *jc = out[1];
    *kc = out[2];
    return 0;
}

// ---------------------------------------------------------------------------------------------
static double arcsin_clamped(double x) // equiSources.f90:2277-2295
{
    if (x > 1.0) return kHalfPi;
    if (x < -1.0) return -kHalfPi;
    return std::asin(x);
}

static double angle_from(double c, double s) // getAngle, equiSources.f90:2337-2361
{
    const double a = arcsin_clamped(s);
    if (c > 0.0) return s > 0.0 ? a : kTwoPi + a;
    return kPi - a;
}

// rotateAngles, equiSources.f90:2297-2335: 0.111 rad about x then 0.222 rad about y (both
// default-real literals), so that no HEALPix direction is parallel to a grid axis.
static void tilt(double *phi, double *theta)
{
    {
        const double a = static_cast<double>(0.111f), p0 = *phi, t0 = *theta;
        const double t = arcsin_clamped(std::cos(t0) * std::sin(p0) * std::sin(a) + std::sin(t0) * std::cos(a));
        const double c = std::cos(t0) * std::cos(p0) / std::cos(t);
        const double s = (std::cos(t0) * std::sin(p0) * std::cos(a) - std::sin(t0) * std::sin(a)) / std::cos(t);
        *phi = angle_from(c, s);
        *theta = t;
    }
    {
        const double a = static_cast<double>(0.222f), p0 = *phi, t0 = *theta;
        const double t = arcsin_clamped(std::cos(t0) * std::cos(p0) * std::sin(a) + std::sin(t0) * std::cos(a));
        const double c = (std::cos(t0) * std::cos(p0) * std::cos(a) - std::sin(t0) * std::sin(a)) / std::cos(t);
        const double s = std::cos(t0) * std::sin(p0) / std::cos(t);
        *phi = angle_from(c, s);
        *theta = t;
    }
}

int pix2ang_nest(int nside, int64_t ipix, double *phi_out, double *theta_out) // equiSources.f90:2118-2231
{
    if (nside < 1 || nside > 4 * 8192) return -1;
    const int64_t per_face = static_cast<int64_t>(nside) * nside;
    if (ipix < 0 || ipix >= 12 * per_face) return -1;

    // ring index of the southernmost corner of each base face (units of nside) and its
    // azimuthal index (units of nside/2): the jrll / jpll tables of :2144-2146
    static const int ring_of_face[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    static const int azim_of_face[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};

    const int face = static_cast<int>(ipix / per_face);
    int64_t in_face = ipix % per_face;
    int x = 0, y = 0;
    for (int b = 0; in_face; ++b, in_face >>= 2) { // even bits -> x, odd bits -> y (:2233-2275)
        x |= static_cast<int>(in_face & 1) << b;
        y |= static_cast<int>((in_face >> 1) & 1) << b;
    }

    const double fn = static_cast<double>(static_cast<float>(nside));
    const double polar_scale = 1.0 / (3.0 * fn * fn);
    const double belt_scale = 2.0 / (3.0 * fn);
    const int ring = ring_of_face[face] * nside - (x + y) - 1; // 1 .. 4 nside - 1
    int in_ring = nside, shift = (ring - nside) % 2;
    double z = static_cast<double>(static_cast<float>(2 * nside - ring)) * belt_scale;
    if (ring < nside) { // north cap
        in_ring = ring;
        z = 1.0 - static_cast<double>(static_cast<float>(in_ring) * static_cast<float>(in_ring)) * polar_scale;
        shift = 0;
    } else if (ring > 3 * nside) { // south cap
        in_ring = 4 * nside - ring;
        z = -1.0 + static_cast<double>(static_cast<float>(in_ring) * static_cast<float>(in_ring)) * polar_scale;
        shift = 0;
    }
    double theta = std::acos(z) - kHalfPi;

    int jp = (azim_of_face[face] * in_ring + (x - y) + 1 + shift) / 2;
    if (jp > 4 * nside) jp -= 4 * nside;
    if (jp < 1) jp += 4 * nside;
    double phi = static_cast<double>(static_cast<float>(jp) - static_cast<float>(shift + 1) * 0.5f) * kHalfPi /
                 static_cast<double>(static_cast<float>(in_ring));
    while (phi > kTwoPi) phi -= kTwoPi;
    while (phi < 0.0) phi += kTwoPi;

    tilt(&phi, &theta);
    if (phi > 2.0 * kPi) return -1;
    *phi_out = phi;
    *theta_out = theta;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// equiSources.f90:1395-1454.  Returns 0, or 1/2/3 for the three `stop`s.
int fold_direction(double phi_in, double theta_in, double *phi, double *theta, int *izone)
{
    int quadrant = -1;
    const double edges[5] = {0.0, 0.5 * kPi, kPi, 1.5 * kPi, 2.0 * kPi};
    for (int q = 0; q < 4; ++q)
        if (phi_in > edges[q] && phi_in < edges[q + 1]) quadrant = q;
    if (quadrant < 0) return 1;
    const double p = phi_in - edges[quadrant];

    double t;
    int zone = 1 + 3 * quadrant;
    if (theta_in > 0.0 && theta_in < 0.5 * kPi) t = theta_in;
    else if (theta_in > -0.5 * kPi && theta_in < 0.0) { t = -theta_in; zone += 12; }
    else return 2;

    // distance to leave the unit cube through the z / x / y face: the smallest names the march axis
    const double along_z = 1.0 / std::sin(t);
    const double along_x = 1.0 / (std::cos(p) * std::cos(t));
    const double along_y = 1.0 / (std::sin(p) * std::cos(t));
    if (along_z < std::fmin(along_x, along_y)) {
        *theta = t;
        *phi = p;
    } else if (along_x < std::fmin(along_z, along_y)) {
        *theta = arcsin_clamped(std::cos(t) * std::cos(p));
        *phi = arcsin_clamped(std::sin(t) / std::cos(*theta));
        zone += 1;
    } else if (along_y < std::fmin(along_z, along_x)) {
        *theta = arcsin_clamped(std::cos(t) * std::sin(p));
        *phi = std::acos(std::sin(t) / std::cos(*theta));
        zone += 2;
    } else return 3;
    *izone = zone;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// setPattern, transportRoutinesModule.f90:7-85.  A ray enters the unit cell through the bottom
// face at (x0,y0) and is cut at the first face it meets; a piece cut by a side face re-enters
// from the opposite side (same cell pattern, neighbouring cell) until the top is reached.
int set_pattern(ftte_pattern *P, double phi, double theta)
{
    const double sin_t = std::sin(theta), cos_t = std::cos(theta);
    const double to_top = 1.0 / sin_t;
    const double to_x1 = (1.0 - P->xy_x0) / (std::cos(phi) * cos_t);
    const double to_y1 = (1.0 - P->xy_y0) / (std::sin(phi) * cos_t);

    P->xz_active = P->yz_active = 0;
    P->xz_top = P->yz_top = 0;

    if (to_top < std::fmin(to_x1, to_y1)) {
        P->xy_len = to_top;
        P->xy_top = 1;
        return 0;
    }
    if (to_x1 < std::fmin(to_top, to_y1)) {
        // out through x = 1: yz piece
        P->xy_len = to_x1;
        P->yz_active = 1;
        P->yz_y0 = (1.0 - P->xy_x0) * std::tan(phi) + P->xy_y0;
        P->yz_z0 = P->xy_len * sin_t;
        if (P->yz_y0 > 1.0 || P->yz_z0 > 1.0) return 1;
        const double up = (1.0 - P->yz_z0) / sin_t;
        const double side = (1.0 - P->yz_y0) / (std::sin(phi) * cos_t);
        P->yz_top = 1; // the xy piece ends on the x = 1 face
        if (up < side) {
            P->yz_len = up;
            P->xy_top = 2;
        } else { // and then out through y = 1 as well: xz piece
            P->yz_len = side;
            P->xz_active = 1;
            P->xz_x0 = (1.0 - P->yz_y0) / std::tan(phi);
            P->xz_z0 = P->yz_z0 + side * sin_t;
            P->xz_len = (1.0 - P->xz_z0) / sin_t;
            P->xy_top = 3;
            P->xz_top = 2;
        }
        return 0;
    }
    // out through y = 1: xz piece
    P->xy_len = to_y1;
    P->xz_active = 1;
    P->xz_x0 = (1.0 - P->xy_y0) / std::tan(phi) + P->xy_x0;
    P->xz_z0 = to_y1 * sin_t;
    if (P->xz_x0 > 1.0 || P->xz_z0 > 1.0) return 1;
    const double up = (1.0 - P->xz_z0) / sin_t;
    const double side = (1.0 - P->xz_x0) / (std::cos(phi) * cos_t);
    P->xz_top = 1;
    if (up < side) {
        P->xz_len = up;
        P->xy_top = 3;
    } else { // and then out through x = 1: yz piece
        P->xz_len = side;
        P->yz_active = 1;
        P->yz_y0 = (1.0 - P->xz_x0) * std::tan(phi);
        P->yz_z0 = P->xz_len * sin_t + P->xz_z0;
        P->yz_len = (1.0 - P->yz_z0) / sin_t;
        P->xy_top = 2;
        P->yz_top = 3;
    }
    return 0;
}

// equiSources.f90:1495-1534: layer 1 starts at the cell centre of the bottom face, layer i where
// the top-ending piece of layer i-1 leaves.
int layer_patterns(int n, double phi, double theta, ftte_pattern *L)
{
    std::memset(L, 0, sizeof(ftte_pattern) * static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        if (i == 0) {
            L[i].xy_x0 = 0.5;
            L[i].xy_y0 = 0.5;
        } else {
            const ftte_pattern &b = L[i - 1];
            if (b.xy_top == 1) {
                L[i].xy_x0 = b.xy_x0 + std::cos(phi) / std::tan(theta);
                L[i].xy_y0 = b.xy_y0 + std::sin(phi) / std::tan(theta);
            } else if (b.xy_top == 3) {
                L[i].xy_x0 = b.xz_x0 + b.xz_len * std::cos(theta) * std::cos(phi);
                L[i].xy_y0 = b.xz_len * std::cos(theta) * std::sin(phi);
            } else if (b.xy_top == 2) {
                L[i].xy_x0 = b.yz_len * std::cos(theta) * std::cos(phi);
                L[i].xy_y0 = b.yz_y0 + b.yz_len * std::cos(theta) * std::sin(phi);
            } else return 1;
            if (L[i].xy_x0 > 1.0 || L[i].xy_y0 > 1.0) return 1;
        }
        if (set_pattern(&L[i], phi, theta)) return 1;
    }
    return 0;
}

} // namespace ftte
