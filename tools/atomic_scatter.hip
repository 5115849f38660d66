// scattered fp64 atomic adds: device scope into one array vs workgroup scope into a per-XCD private copy
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE> __global__ void __launch_bounds__(64) scatter(double *a, long ncell, int steps, int per)
{
    const unsigned tid = blockIdx.x * 64 + threadIdx.x;
    unsigned xcc = 0;
    if (MODE == 1) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; }
    double *base = a + (MODE == 1 ? (long)xcc * ncell * 8 : 0);
    for (int k = 0; k < steps; ++k) {
        const long c = hash(tid * 9781u + k) % (unsigned long)ncell;
        for (int r = 0; r < per; ++r) {
            if (MODE == 0) unsafeAtomicAdd(base + c * 8 + r, 1.0);
            else __hip_atomic_fetch_add(base + c * 8 + r, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

int main()
{
    const long ncell = 1l << 24;
    double *a;
    CHECK(hipMalloc(&a, ncell * 8 * 8 * sizeof(double)));
    CHECK(hipMemset(a, 0, ncell * 8 * 8 * sizeof(double)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 98304, steps = 64;
    for (int per : {1, 6}) {
        for (int mode = 0; mode < 2; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(scatter<0>, dim3(blocks), dim3(64), 0, 0, a, ncell, steps, per);
                else hipLaunchKernelGGL(scatter<1>, dim3(blocks), dim3(64), 0, 0, a, ncell, steps, per);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double n = (double)blocks * 64 * steps * per;
            printf("%d atomics per cell, %s: %8.2f ms  %.3e atomics/s\n", per, mode ? "workgroup scope, per-XCD copy" : "device scope, one array      ", ms, n / ms * 1e3);
        }
    }
    // check the per-XCD sums
    double *h = (double *)malloc(8 * 8 * sizeof(double));
    CHECK(hipMemcpy(h, a, 64 * sizeof(double), hipMemcpyDeviceToHost));
    printf("cell 0 of copy 0: %g %g\n", h[0], h[5]);
    return 0;
}
