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

int main()
{
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
