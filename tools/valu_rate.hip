// valu_rate.hip -- issue rates of the vector instructions the brick kernel is made of, on the GPU at hand (gfx950):
// cycles per wavefront instruction for chains of dependent and of independent operations, at 1, 2 and 4 wavefronts per SIMD.
// The sweep's arithmetic floor is priced with these numbers.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kUnroll = 32;

// OP: 0 fma_f64 dependent chain, 1 fma_f64 four independent chains, 2 mul_f64 dep, 3 add_f64 dep, 4 fma_f32 dep, 5 v_mov_b32 (dpp shift) dep,
// 6 cndmask_b32 dep, 7 ldexp_f64 dep, 8 rndne_f64 dep, 9 rcp_f64 dep, 10 v_mov_b64 dep, 11 v_pk_fma_f32 dep, 12 fma_f64 with an SGPR operand
template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(double *out, int iters, double seed)
{
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9, a1 = a + 1, a2 = a + 2, a3 = a + 3;
    float fa = (float)a, fb = 1.0000001f, fc = 1e-9f;
    int ia = threadIdx.x, ib = 1;
    double s = seed; // wave-uniform: lives in SGPRs
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            if (OP == 1) {
                if ((k & 3) == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                if ((k & 3) == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                if ((k & 3) == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                if ((k & 3) == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
            }
            if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
            if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));
            if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));
            if (OP == 5) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ia));
            if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib) : );
            if (OP == 7) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(ib));
            if (OP == 8) asm volatile("v_rndne_f64 %0, %0" : "+v"(a));
            if (OP == 9) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));
            if (OP == 10) asm volatile("v_mov_b64 %0, %0" : "+v"(a));
            if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            if (OP == 12) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "s"(s), "v"(c));
        }
    }
    if (a + a1 + a2 + a3 + fa + ia == 12345.678) out[0] = a; // keeps the chains alive
}

template <int OP>
double run(const char *name, int waves_per_simd, double clock_ghz, double *out)
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 4 * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<OP>), dim3(blocks), dim3(64), 0, 0, out, 1000, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<OP>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // cycles of a SIMD per instruction of one of its wavefronts
    const double cyc = ms * 1e-3 * clock_ghz * 1e9 / ((double)iters * kUnroll * waves_per_simd);
    std::printf("%-34s %d waves/SIMD: %7.3f ms  %6.2f cycles per wavefront instruction\n", name, waves_per_simd, ms, cyc);
    return cyc;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    std::printf("%s, %d CUs, clock %.3f GHz (device property; the running clock may be lower)\n", p.name, p.multiProcessorCount, ghz);
    double *out;
    CHECK(hipMalloc((void **)&out, 64));
    for (int w : {1, 2, 4}) {
        run<4>("v_fma_f32 dependent", w, ghz, out);
        run<0>("v_fma_f64 dependent", w, ghz, out);
        run<1>("v_fma_f64 four chains", w, ghz, out);
        run<12>("v_fma_f64 dependent, SGPR operand", w, ghz, out);
        run<2>("v_mul_f64 dependent", w, ghz, out);
        run<3>("v_add_f64 dependent", w, ghz, out);
        run<10>("v_mov_b64 dependent", w, ghz, out);
        run<11>("v_pk_fma_f32 dependent", w, ghz, out);
        run<5>("v_mov_b32 dpp wave_shr dependent", w, ghz, out);
        run<6>("v_cndmask_b32 dependent", w, ghz, out);
        run<7>("v_ldexp_f64 dependent", w, ghz, out);
        run<8>("v_rndne_f64 dependent", w, ghz, out);
        run<9>("v_rcp_f64 dependent", w, ghz, out);
    }
    return 0;
}
