// membench.hip -- what the MI355X memory system gives for the access mix of the sweep (kappa read, J read, J write:
// 24 bytes per element) with plain streaming kernels, as the practical ceiling to compare the sweep kernel against.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// j += k, VEC doubles per lane per access, UNROLL accesses in flight per lane
template <int VEC, int UNROLL> __global__ void __launch_bounds__(256) rmw(const double *__restrict__ k, double *__restrict__ j, long n)
{
    const long chunk = (long)gridDim.x * blockDim.x * VEC;
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    for (; i + (UNROLL - 1) * chunk + VEC <= n; i += UNROLL * chunk) {
        double a[UNROLL][VEC], b[UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { a[u][e] = k[i + u * chunk + e]; b[u][e] = j[i + u * chunk + e]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) j[i + u * chunk + e] = a[u][e] + b[u][e];
    }
}

// j += k with the add done by the L2 (fp64 atomic without return): no J load, no J register, same HBM traffic
template <int UNROLL> __global__ void __launch_bounds__(256) rmw_atomic(const double *__restrict__ k, double *__restrict__ j, long n)
{
    const long chunk = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * chunk < n; i += UNROLL * chunk) {
        double a[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) a[u] = k[i + u * chunk];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) unsafeAtomicAdd(&j[i + u * chunk], a[u]);
    }
}

template <int VEC> __global__ void __launch_bounds__(256) copy(const double *__restrict__ k, double *__restrict__ j, long n)
{
    const long chunk = (long)gridDim.x * blockDim.x * VEC;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i + VEC <= n; i += chunk)
#pragma unroll
        for (int e = 0; e < VEC; ++e) j[i + e] = k[i + e];
}

template <int VEC> __global__ void __launch_bounds__(256) readonly(const double *__restrict__ k, double *__restrict__ out, long n)
{
    const long chunk = (long)gridDim.x * blockDim.x * VEC;
    double s = 0;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i + VEC <= n; i += chunk)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += k[i + e];
    if (s == 123.456) out[0] = s;
}

// The brick kernel's way of asking for the same bytes: one wavefront per workgroup takes a brick of 64 x 8 x `chunk` cells of an
// (n+2)^3 frame and, layer after layer, loads its eight rows of k and of j (512 B each, a row of the frame apart), adds, and stores the
// rows of j -- no arithmetic to speak of, no dependencies between bricks, `lds` bytes of dynamic LDS to set the residency.
__global__ void __launch_bounds__(64) bricks(const double *__restrict__ k, double *__restrict__ j, int n, int chunk, int nnu, int rmw_j, int tiled)
{
    extern __shared__ double pad[];
    const int ntu = n / 64, ntv = n / 8, nti = n / chunk;
    long b = blockIdx.x;
    const int nu = (int)(b % nnu); b /= nnu;
    const int tu = (int)(b % ntu); b /= ntu;
    const int tv = (int)(b % ntv); b /= ntv;
    const int ti = (int)b;
    if (ti >= nti) return;
    // tiled 0: the frame as the library keeps it; 1: a brick's layer in one piece of 4 KB (rows 512 B apart), the layers a plane of
    // such pieces apart; 2: the whole brick in one piece
    long row = n + 2, plane = row * row, group = plane * row;
    long base = nu * group + (long)(ti * chunk + 1) * plane + (long)(tv * 8 + 1) * row + tu * 64 + 1 + threadIdx.x;
    if (tiled == 1) { row = 64; plane = (long)ntu * ntv * 512; base = nu * group + (long)(ti * chunk) * plane + ((long)tv * ntu + tu) * 512 + threadIdx.x; }
    if (tiled == 2) { row = 64; plane = 512; base = nu * group + (((long)ti * ntv + tv) * ntu + tu) * 512l * chunk + threadIdx.x; }
    double next[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) next[r] = k[base + r * row];
    for (int i = 0; i < chunk; ++i) {
        double kap[8], acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { kap[r] = next[r]; acc[r] = 0.0; }
        if (rmw_j)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = __builtin_nontemporal_load(&j[base + i * plane + r * row]);
        if (i + 1 < chunk)
#pragma unroll
            for (int r = 0; r < 8; ++r) next[r] = k[base + (i + 1) * plane + r * row];
#pragma unroll
        for (int r = 0; r < 8; ++r) __builtin_nontemporal_store(acc[r] + kap[r], &j[base + i * plane + r * row]);
    }
    if (pad[0] == 123.456) j[0] = 0;
}

int main()
{
    {
        const int n = 256, nnu = 8;
        const long cells = (long)(n + 2) * (n + 2) * (n + 2) * nnu;
        double *k, *j;
        CHECK(hipMalloc(&k, cells * 8)); CHECK(hipMalloc(&j, cells * 8));
        CHECK(hipMemset(k, 0, cells * 8)); CHECK(hipMemset(j, 0, cells * 8));
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float ms;
        printf("-- brick pattern: 256^3 x 8 groups, one wavefront per 64 x 8 x chunk brick, all bricks in one launch\n");
        for (int chunk : {16, 4})
            for (int rmw_j : {0, 1})
                for (int tiled : {0, 1, 2}) {
                    const int lds = 8192;
                    const unsigned grid = (unsigned)((n / 64) * (n / 8) * (n / chunk) * nnu);
                    const double bytes = (double)n * n * n * nnu * (rmw_j ? 24.0 : 16.0);
                    for (int rep = 0; rep < 3; ++rep) {
                        CHECK(hipEventRecord(a));
                        hipLaunchKernelGGL(bricks, dim3(grid), dim3(64), lds, 0, k, j, n, chunk, nnu, rmw_j, tiled);
                        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                        CHECK(hipEventElapsedTime(&ms, a, b));
                        if (rep == 2) printf("chunk %2d %s %-28s %7.3f ms  %6.0f GB/s\n", chunk, rmw_j ? "k,J->J" : "k->J  ", tiled == 0 ? "rows of the frame" : tiled == 1 ? "4 KB per brick and layer" : "the brick in one piece", ms, bytes / ms / 1e6);
                    }
                }
        CHECK(hipFree(k)); CHECK(hipFree(j));
    }

    const long n = 1l << 28; // 2 GiB of doubles per array
    double *k, *j;
    CHECK(hipMalloc(&k, n * 8)); CHECK(hipMalloc(&j, n * 8));
    CHECK(hipMemset(k, 0, n * 8)); CHECK(hipMemset(j, 0, n * 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms;
#define RUN(label, bytes, ...)                                                                         \
    for (int rep = 0; rep < 3; ++rep) {                                                                \
        CHECK(hipEventRecord(a)); __VA_ARGS__; CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); \
        CHECK(hipEventElapsedTime(&ms, a, b));                                                         \
        if (rep == 2) printf("%-44s %7.3f ms  %6.0f GB/s\n", label, ms, (bytes) / ms / 1e6);            \
    }
    for (int blocks : {2048, 8192}) {
        printf("-- %d blocks of 256\n", blocks);
        RUN("read        8 B/lane", n * 8.0, hipLaunchKernelGGL(readonly<1>, dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("read       16 B/lane", n * 8.0, hipLaunchKernelGGL(readonly<2>, dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("copy        8 B/lane", n * 16.0, hipLaunchKernelGGL(copy<1>, dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("copy       16 B/lane", n * 16.0, hipLaunchKernelGGL(copy<2>, dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("rmw (k,J->J)  8 B/lane x1", n * 24.0, hipLaunchKernelGGL((rmw<1, 1>), dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("rmw (k,J->J)  8 B/lane x4", n * 24.0, hipLaunchKernelGGL((rmw<1, 4>), dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("rmw (k,J->J) 16 B/lane x1", n * 24.0, hipLaunchKernelGGL((rmw<2, 1>), dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("rmw (k,J->J) 16 B/lane x4", n * 24.0, hipLaunchKernelGGL((rmw<2, 4>), dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("rmw atomic (k -> J += k) x1", n * 24.0, hipLaunchKernelGGL((rmw_atomic<1>), dim3(blocks), dim3(256), 0, 0, k, j, n));
        RUN("rmw atomic (k -> J += k) x4", n * 24.0, hipLaunchKernelGGL((rmw_atomic<4>), dim3(blocks), dim3(256), 0, 0, k, j, n));
    }
    return 0;
}
