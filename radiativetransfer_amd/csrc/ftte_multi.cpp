// ftte_multi.cpp -- several devices behind ONE context (ftte_create with ndev > 1): what SURVEY.md 8(b) specifies for the Fortran
// host, which is one serial process (equiSources.f90:1385-1391 is its direction loop; the only coupling of the directions is the sum
// transportRoutinesModule.f90:953-955).  One single-device context per device; host arrays in, host arrays out:
//   * frequency groups first, then directions -- the split of radiativetransfer_amd/distributed.py (Shard2D): with r_nu = gcd(devices,
//     groups) frequency slices and r_dir = devices / r_nu direction slices, device k sweeps the groups of slice k mod r_nu for the
//     directions of slice k div r_nu.  Frequency groups never meet inside the sweep, so a frequency-sharded J_nu is complete where it
//     is computed: with as many groups as devices (8 and 8) nothing is exchanged at all, every device sends its own groups home;
//   * only where the directions are split as well is J summed -- a reduce-scatter over the r_dir devices that hold the same groups
//     (RCCL, grouped calls from the one host thread; every device ends with 1/r_dir of its groups' cells and sends that piece home).
// RCCL is loaded on first need (librccl.so is half a gigabyte: a single-device process never maps it).  Where it cannot serve --
// two contexts on one physical device, as on a one-GPU test box; no library -- the same pieces are summed by a kernel that reads
// the partners' buffers directly (one process: device memory of a peer is addressable), in device order.
#include <dlfcn.h>

#include <numeric>
#include <thread>

#include "ftte_context.h"

namespace ftte {

namespace {

// the few entry points of rccl.h this file needs (ncclResult_t 0 = success; ncclDouble = 8, ncclSum = 0: rccl.h:448,467)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*ReduceScatter)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why; // why it is not there
    bool load()
    {
        if (lib) return true;
        if (!why.empty()) return false;
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { why = std::string("librccl.so not loadable: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        ReduceScatter = (decltype(ReduceScatter))dlsym(lib, "ncclReduceScatter");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !ReduceScatter || !GetErrorString) {
            why = "librccl.so lacks an entry point";
            dlclose(lib); lib = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;

void bounds(int count, int part, int parts, int *lo, int *hi) // shard_bounds of distributed.py
{
    const int base = count / parts, extra = count % parts;
    *lo = part * base + std::min(part, extra);
    *hi = *lo + base + (part < extra ? 1 : 0);
}

} // namespace

struct Multi {
    std::vector<ftte_ctx *> sub;       // one single-device context per device, in the caller's order
    std::vector<int> dev;
    int nnu = 0, r_nu = 1, r_dir = 1;
    std::vector<double *> J_part, J_red; // per device: its sweep's J [groups][ncell] (padded to r_dir pieces); its piece of the sum
    std::vector<size_t> J_cap, red_cap;
    std::vector<hipEvent_t> swept;     // per device: its sweep is enqueued up to here
    std::vector<void *> comm;          // per device: its communicator inside its group of r_dir devices, or empty
    int comm_r_dir = 0, comm_r_nu = 0;
    bool rccl = false;
    int reduce_opt = 0;                // option "multi_reduce": 0 RCCL where it can serve, 1 always the peer kernel
    std::string how;                   // how the last direction-split sweep was summed, in words
};

static int adopt(ftte_ctx *c, ftte_ctx *sub, int rc)
{
    if (rc) c->err = sub->err;
    return rc;
}

// fn(k) for every device, each on a thread of its own (uploads and downloads have a PCIe link each; plan builds are host work)
template <typename F> static int on_every_device(ftte_ctx *c, F fn)
{
    Multi &M = *c->multi;
    const size_t nd = M.sub.size();
    std::vector<int> rc(nd, 0);
    std::vector<std::thread> pool;
    for (size_t k = 1; k < nd; ++k) pool.emplace_back([&, k] { (void)hipSetDevice(M.dev[k]); rc[k] = fn((int)k); });
    (void)hipSetDevice(M.dev[0]);
    rc[0] = fn(0);
    for (auto &t : pool) t.join();
    for (size_t k = 0; k < nd; ++k)
        if (rc[k]) return adopt(c, M.sub[k], rc[k]);
    return FTTE_OK;
}

static void drop_comms(Multi &M)
{
    for (void *q : M.comm) if (q && g_rccl.lib) (void)g_rccl.CommDestroy(q);
    M.comm.clear();
    M.rccl = false;
}

int multi_create(ftte_ctx **out, int ndev, const int *dev_ids)
{
    if (ndev < 2 || ndev > 64) return fail(nullptr, FTTE_ERR_ARG, "ftte_create: ndev must be 1..64");
    if (!dev_ids) return fail(nullptr, FTTE_ERR_ARG, "ftte_create: ndev > 1 needs the device ordinals");
    ftte_ctx *c = new ftte_ctx;
    c->multi = new Multi;
    Multi &M = *c->multi;
    for (int k = 0; k < ndev; ++k) {
        ftte_ctx *s = nullptr;
        const int rc = ftte_create(&s, 1, dev_ids + k);
        if (rc) { // g_create_error holds the reason
            for (ftte_ctx *q : M.sub) (void)ftte_destroy(q);
            delete c->multi; delete c;
            return rc;
        }
        M.sub.push_back(s);
        M.dev.push_back(dev_ids[k]);
    }
    c->device = dev_ids[0];
    M.J_part.assign((size_t)ndev, nullptr); M.J_red.assign((size_t)ndev, nullptr);
    M.J_cap.assign((size_t)ndev, 0); M.red_cap.assign((size_t)ndev, 0);
    M.swept.assign((size_t)ndev, nullptr);
    *out = c;
    return FTTE_OK;
}

int multi_destroy(ftte_ctx *c)
{
    Multi &M = *c->multi;
    for (size_t k = 0; k < M.sub.size(); ++k) {
        (void)hipSetDevice(M.dev[k]);
        (void)hipDeviceSynchronize();
        if (M.J_part[k]) (void)hipFree(M.J_part[k]);
        if (M.J_red[k]) (void)hipFree(M.J_red[k]);
        if (M.swept[k]) (void)hipEventDestroy(M.swept[k]);
    }
    drop_comms(M);
    for (ftte_ctx *s : M.sub) (void)ftte_destroy(s);
    delete c->multi;
    delete c;
    return FTTE_OK;
}

int multi_set_grid(ftte_ctx *c, int nx, int ny, int nz, int64_t ncell, const int32_t *level, double box_cm)
{
    Multi &M = *c->multi;
    const int rc = on_every_device(c, [&](int k) -> int { return ftte_set_grid(M.sub[(size_t)k], nx, ny, nz, ncell, level, box_cm); });
    if (rc) return rc;
    c->grid_set = true; c->n = nx; c->ncell = ncell; c->box = box_cm;
    return FTTE_OK;
}

static void split(Multi &M, int nnu)
{
    const int nd = (int)M.sub.size();
    M.nnu = nnu;
    M.r_nu = std::gcd(nd, nnu);
    M.r_dir = nd / M.r_nu;
}

static void groups_of(const Multi &M, int k, int *lo, int *hi) { bounds(M.nnu, k % M.r_nu, M.r_nu, lo, hi); }

int multi_set_opacity(ftte_ctx *c, int nnu, const double *kappa)
{
    if (!c->grid_set) return fail(c, FTTE_ERR_STATE, "ftte_set_grid has not been called");
    if (nnu < 1 || !kappa) return fail(c, FTTE_ERR_ARG, "ftte_set_opacity: bad argument");
    Multi &M = *c->multi;
    split(M, nnu);
    c->nnu = nnu;
    return on_every_device(c, [&](int k) -> int {
        int lo, hi;
        groups_of(M, k, &lo, &hi);
        return ftte_set_opacity(M.sub[(size_t)k], hi - lo, kappa + (size_t)lo * (size_t)c->ncell);
    });
}

int multi_set_emission(ftte_ctx *c, int mode, const double *values)
{
    Multi &M = *c->multi;
    if (values && !M.nnu) return fail(c, FTTE_ERR_STATE, "no opacities: call ftte_set_opacity first (it fixes the frequency groups of every device)");
    return on_every_device(c, [&](int k) -> int {
        int lo = 0, hi = 0;
        if (values) groups_of(M, k, &lo, &hi);
        const double *mine = values ? values + (size_t)lo * (size_t)c->ncell : nullptr;
        return mode == 1 ? ftte_set_emissivity(M.sub[(size_t)k], mine) : ftte_set_source_function(M.sub[(size_t)k], mine);
    });
}

int multi_set_option(ftte_ctx *c, const char *key, int value)
{
    Multi &M = *c->multi;
    if (!std::strcmp(key, "multi_reduce")) {
        if (value < 0 || value > 1) return fail(c, FTTE_ERR_ARG, "multi_reduce must be 0 (RCCL where it can serve) or 1 (always the kernel that reads the partners' buffers)");
        M.reduce_opt = value;
        return FTTE_OK;
    }
    for (size_t k = 0; k < M.sub.size(); ++k) {
        const int rc = ftte_set_option(M.sub[k], key, value);
        if (rc) return adopt(c, M.sub[k], rc);
    }
    return FTTE_OK;
}

// the communicators of the current split: one clique per frequency slice, its r_dir devices in direction-slice order
static bool make_comms(ftte_ctx *c)
{
    Multi &M = *c->multi;
    if (M.rccl && M.comm_r_dir == M.r_dir && M.comm_r_nu == M.r_nu) return true;
    drop_comms(M);
    M.comm_r_dir = M.r_dir; M.comm_r_nu = M.r_nu;
    if (M.reduce_opt == 1) { M.how = "summed by the peer kernel (option multi_reduce = 1)"; return false; }
    std::vector<int> sorted(M.dev);
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
        M.how = "summed by the peer kernel: two contexts share a physical device, which RCCL's communicators cannot (one rank per device)";
        return false;
    }
    if (!g_rccl.load()) { M.how = "summed by the peer kernel: " + g_rccl.why; return false; }
    M.comm.assign(M.sub.size(), nullptr);
    for (int a = 0; a < M.r_nu; ++a) {
        std::vector<int> devs((size_t)M.r_dir);
        std::vector<void *> comms((size_t)M.r_dir, nullptr);
        for (int b = 0; b < M.r_dir; ++b) devs[(size_t)b] = M.dev[(size_t)(a + M.r_nu * b)];
        const int rc = g_rccl.CommInitAll(comms.data(), M.r_dir, devs.data());
        if (rc) {
            M.how = std::string("summed by the peer kernel: ncclCommInitAll: ") + g_rccl.GetErrorString(rc);
            for (int b = 0; b < M.r_dir; ++b) M.comm[(size_t)(a + M.r_nu * b)] = comms[(size_t)b];
            drop_comms(M);
            return false;
        }
        for (int b = 0; b < M.r_dir; ++b) M.comm[(size_t)(a + M.r_nu * b)] = comms[(size_t)b];
    }
    M.rccl = true;
    M.how = "reduce-scatter over RCCL, one communicator clique per frequency slice";
    return true;
}

int multi_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J)
{
    Multi &M = *c->multi;
    if (!c->grid_set) return fail(c, FTTE_ERR_STATE, "ftte_set_grid has not been called");
    if (!M.nnu) return fail(c, FTTE_ERR_STATE, "no opacities: call ftte_set_opacity first");
    if (ndir < 0 || (ndir > 0 && (!phi || !theta || !w)) || !uvb || !J) return fail(c, FTTE_ERR_ARG, "ftte_diffuse_sweep: bad argument");
    const int nd = (int)M.sub.size();
    const size_t ncell = (size_t)c->ncell;

    // every device sweeps its directions for its groups into its own J (asynchronous from here on; the first call plans, on the host)
    int rc = on_every_device(c, [&](int k) -> int {
        ftte_ctx *s = M.sub[(size_t)k];
        int lo, hi, dlo, dhi;
        groups_of(M, k, &lo, &hi);
        bounds(ndir, k / M.r_nu, M.r_dir, &dlo, &dhi);
        const size_t total = (size_t)(hi - lo) * ncell, piece = (total + (size_t)M.r_dir - 1) / (size_t)M.r_dir;
        if (M.J_cap[(size_t)k] < piece * (size_t)M.r_dir) {
            if (M.J_part[(size_t)k]) FTTE_HIP(s, hipFree(M.J_part[(size_t)k]));
            M.J_part[(size_t)k] = nullptr; M.J_cap[(size_t)k] = 0;
            FTTE_HIP(s, hipMalloc((void **)&M.J_part[(size_t)k], sizeof(double) * piece * (size_t)M.r_dir));
            FTTE_HIP(s, hipMemset(M.J_part[(size_t)k], 0, sizeof(double) * piece * (size_t)M.r_dir)); // (the padding of the last piece stays zero)
            M.J_cap[(size_t)k] = piece * (size_t)M.r_dir;
        }
        if (M.r_dir > 1 && M.red_cap[(size_t)k] < piece) {
            if (M.J_red[(size_t)k]) FTTE_HIP(s, hipFree(M.J_red[(size_t)k]));
            M.J_red[(size_t)k] = nullptr; M.red_cap[(size_t)k] = 0;
            FTTE_HIP(s, hipMalloc((void **)&M.J_red[(size_t)k], sizeof(double) * piece));
            M.red_cap[(size_t)k] = piece;
        }
        if (!M.swept[(size_t)k]) FTTE_HIP(s, hipEventCreateWithFlags(&M.swept[(size_t)k], hipEventDisableTiming));
        const int src = ftte_diffuse_sweep_device(s, dhi - dlo, phi + dlo, theta + dlo, w + dlo, uvb + lo, M.J_part[(size_t)k], nullptr);
        if (src) return src;
        FTTE_HIP(s, hipEventRecord(M.swept[(size_t)k], s->stream));
        return FTTE_OK;
    });
    if (rc) return rc;

    if (M.r_dir > 1) {
        // J_nu = sum over the direction slices: device (a, b) ends with piece b of slice a's groups
        const bool by_rccl = make_comms(c);
        if (by_rccl) {
            int nrc = g_rccl.GroupStart();
            for (int k = 0; k < nd && !nrc; ++k) {
                int lo, hi;
                groups_of(M, k, &lo, &hi);
                const size_t total = (size_t)(hi - lo) * ncell, piece = (total + (size_t)M.r_dir - 1) / (size_t)M.r_dir;
                (void)hipSetDevice(M.dev[(size_t)k]);
                nrc = g_rccl.ReduceScatter(M.J_part[(size_t)k], M.J_red[(size_t)k], piece, /*ncclDouble*/ 8, /*ncclSum*/ 0, M.comm[(size_t)k], M.sub[(size_t)k]->stream);
            }
            const int erc = g_rccl.GroupEnd();
            if (nrc || erc) return fail(c, FTTE_ERR_NO_DEVICE, std::string("ncclReduceScatter: ") + g_rccl.GetErrorString(nrc ? nrc : erc));
        } else {
            for (int k = 0; k < nd; ++k) {
                int lo, hi;
                groups_of(M, k, &lo, &hi);
                const int a = k % M.r_nu, b = k / M.r_nu;
                const size_t total = (size_t)(hi - lo) * ncell, piece = (total + (size_t)M.r_dir - 1) / (size_t)M.r_dir;
                ftte_ctx *s = M.sub[(size_t)k];
                FTTE_HIP(c, hipSetDevice(M.dev[(size_t)k]));
                const double *parts[64];
                for (int q = 0; q < M.r_dir; ++q) {
                    const int peer = a + M.r_nu * q;
                    parts[q] = M.J_part[(size_t)peer] + (size_t)b * piece;
                    if (peer != k) {
                        if (M.dev[(size_t)peer] != M.dev[(size_t)k]) {
                            const hipError_t e = hipDeviceEnablePeerAccess(M.dev[(size_t)peer], 0);
                            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                                return fail(c, FTTE_ERR_NO_DEVICE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                            (void)hipGetLastError();
                        }
                        FTTE_HIP(c, hipStreamWaitEvent(s->stream, M.swept[(size_t)peer], 0));
                    }
                }
                if (launch_sum_parts(parts, M.r_dir, M.J_red[(size_t)k], (long)piece, s->stream))
                    return fail(c, FTTE_ERR_NO_DEVICE, "sum kernel launch failed");
            }
        }
    } else M.how = "nothing to sum: every device holds all directions of its frequency groups";

    // home: every device its own piece of the caller's array
    rc = on_every_device(c, [&](int k) -> int {
        ftte_ctx *s = M.sub[(size_t)k];
        int lo, hi;
        groups_of(M, k, &lo, &hi);
        const size_t total = (size_t)(hi - lo) * ncell;
        int drc;
        if (M.r_dir == 1) drc = download(s, J + (size_t)lo * ncell, M.J_part[(size_t)k], sizeof(double) * total);
        else {
            const size_t piece = (total + (size_t)M.r_dir - 1) / (size_t)M.r_dir, at = (size_t)(k / M.r_nu) * piece;
            const size_t count = at < total ? std::min(piece, total - at) : 0;
            drc = count ? download(s, J + (size_t)lo * ncell + at, M.J_red[(size_t)k], sizeof(double) * count) : FTTE_OK;
            if (!drc) FTTE_HIP(s, hipStreamSynchronize(s->stream));
        }
        if (drc) return drc;
        return wait_sweep(s);
    });
    return rc; // (every stream has drained: the partners' buffers the sums read are free for the next sweep)
}

int multi_iteration(ftte_ctx *c, int nnu, const double *kappa, int ndir, const double *phi, const double *theta, const double *w,
                    const double *uvb, double *J)
{
    const int rc = multi_set_opacity(c, nnu, kappa);
    if (rc) return rc;
    return multi_sweep(c, ndir, phi, theta, w, uvb, J);
}

// The calling sequence of the direction sum -- ncclCommInitAll, a grouped ncclReduceScatter of doubles on the sweep's stream,
// ncclCommDestroy -- on a clique of ONE rank, the first device: what a box with a single GPU can check of the RCCL branch (that the
// library loads, that the entry points take these arguments, that the result lands where the sweep expects it).
// 1: the piece came back unchanged; 0: it did not; -(1000 + r): RCCL returned r; -1: librccl is not loadable; -2: a HIP call failed.
static long long rccl_selftest(const Multi &M)
{
    if (!g_rccl.load()) return -1;
    const int dev = M.dev[0];
    if (hipSetDevice(dev) != hipSuccess) return -2;
    const size_t n = 4096;
    std::vector<double> in(n), out(n, 0.0);
    for (size_t i = 0; i < n; ++i) in[i] = 1.0 / (double)(i + 3);
    double *d_in = nullptr, *d_out = nullptr;
    void *comm = nullptr;
    long long verdict = -2;
    hipStream_t stream = M.sub[0]->stream;
    if (hipMalloc((void **)&d_in, n * sizeof(double)) == hipSuccess && hipMalloc((void **)&d_out, n * sizeof(double)) == hipSuccess &&
        hipMemcpy(d_in, in.data(), n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
        hipMemset(d_out, 0, n * sizeof(double)) == hipSuccess) {
        int rc = g_rccl.CommInitAll(&comm, 1, &dev);
        if (!rc) {
            rc = g_rccl.GroupStart();
            if (!rc) rc = g_rccl.ReduceScatter(d_in, d_out, n, /*ncclDouble*/ 8, /*ncclSum*/ 0, comm, stream);
            const int erc = g_rccl.GroupEnd();
            if (!rc) rc = erc;
        }
        if (rc) verdict = -(1000 + rc);
        else if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(out.data(), d_out, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess)
            verdict = std::memcmp(in.data(), out.data(), n * sizeof(double)) == 0 ? 1 : 0;
    }
    if (comm) (void)g_rccl.CommDestroy(comm);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    (void)hipGetLastError();
    return verdict;
}

long long multi_counter(const ftte_ctx *c, const char *name)
{
    const Multi &M = *c->multi;
    if (!std::strcmp(name, "rccl_selftest")) return rccl_selftest(M);
    if (!std::strcmp(name, "devices")) return (long long)M.sub.size();
    if (!std::strcmp(name, "multi_rccl")) return M.rccl ? 1 : 0;
    if (!std::strcmp(name, "rccl_loadable")) return g_rccl.load() ? 1 : 0; // librccl.so is there and has the entry points this file calls
    if (!std::strcmp(name, "frequency_slices")) return M.r_nu;
    if (!std::strcmp(name, "direction_slices")) return M.r_dir;
    return ftte_counter(M.sub[0], name);
}

const char *multi_how(const ftte_ctx *c) { return c->multi->how.c_str(); }
ftte_ctx *multi_first(const ftte_ctx *c) { return c->multi->sub[0]; }

int multi_host_register(ftte_ctx *c, void *ptr, size_t bytes, bool on)
{
    // pinned memory is pinned for the process: the first device's context does it, the others learn the range
    Multi &M = *c->multi;
    const int rc = on ? ftte_host_register(M.sub[0], ptr, bytes) : ftte_host_unregister(M.sub[0], ptr);
    if (rc) return adopt(c, M.sub[0], rc);
    for (size_t k = 1; k < M.sub.size(); ++k) {
        auto &R = M.sub[k]->registered_elsewhere;
        if (on) R.push_back({(const char *)ptr, bytes});
        else R.erase(std::remove_if(R.begin(), R.end(), [&](const ftte_ctx::HostRange &r) { return r.base == (const char *)ptr; }), R.end());
    }
    return FTTE_OK;
}

} // namespace ftte
