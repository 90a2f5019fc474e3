// valu_rate — issue cost (cycles per wave64 instruction, one wave on a SIMD, independent instructions) of the VALU
// operations the traversal kernels are made of: f64 fma / add / mul, f32 fma, f64 <-> f32 conversions, v_rcp_f64.
// usage: valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double seed, int reps) {
    double a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3; f[i] = (float)a[i]; }
    const double m = 1.0000001, c = 1e-9;
    const float mf = 1.0000001f, cf = 1e-9f;
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = fma(a[i], m, c);
            if (OP == 1) a[i] = a[i] + c;
            if (OP == 2) a[i] = a[i] * m;
            if (OP == 3) f[i] = fmaf(f[i], mf, cf);
            if (OP == 4) { f[i] = (float)a[i]; a[i] = a[i] + (double)0; asm volatile("" : "+v"(f[i])); }
            if (OP == 5) { a[i] = (double)f[i]; asm volatile("" : "+v"(a[i])); }
            if (OP == 6) { asm volatile("v_rcp_f64 %0, %1" : "=v"(a[i]) : "v"(a[i])); }
            if (OP == 7) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i])); }
            if (OP == 8) { asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(f[i])); }
            if (OP == 9) { asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(m), "v"(c)); }
            if (OP == 10) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[i]), "v"(mf), "v"(cf)); }
            if (OP == 11) { asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c)); }
            if (OP == 12) { asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(m)); }
            if (OP == 13) { asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[i]), "v"(mf), "v"(cf)); }
            if (OP == 14) { asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a[i]), "v"(m) : "vcc"); }
            if (OP == 15) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(f[i]), "v"(mf) : "vcc"); }
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = c1 - c0; cyc[2 * blockIdx.x + 1] = t1 - t0; }
}

template <int OP>
int run(const char* name, int waves_per_simd) {
    double* out; long long* cyc;
    const int reps = 4096;
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int grid = waves_per_simd == 0 ? 1 : pr.multiProcessorCount * 4 * waves_per_simd;
    CHECK(hipMalloc(&out, sizeof(double) * 64 * grid)); CHECK(hipMalloc(&cyc, sizeof(long long) * 2 * grid));
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0, 16);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0, reps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h[2]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    const double ns = (double)h[1] / khz * 1e6;
    // per wave instruction: shader clocks by s_memtime ticks are not core clocks on every part, so also ns from the wall clock
    printf("%-18s waves/SIMD %d: %7.2f ns per 8 instr per wave -> %5.2f ns/instr (x %d waves: SIMD busy %5.2f ns per wave-instr); kernel %.3f ms\n", name,
           waves_per_simd, ns / reps, ns / reps / 8.0, waves_per_simd ? waves_per_simd : 1, ns / reps / 8.0 / (waves_per_simd ? waves_per_simd : 1), ms);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
    return 0;
}
int main() {
    for (int w : {0, 1, 4}) {
        run<10>("v_fma_f32", w); run<9>("v_fma_f64", w); run<11>("v_add_f64", w); run<12>("v_mul_f64", w);
        run<7>("v_cvt_f32_f64", w); run<5>("cvt f64<-f32", w); run<8>("v_cvt_f32_ubyte1", w); run<6>("v_rcp_f64", w);
        run<13>("v_max3_f32", w); run<14>("v_cmp_lt_f64", w); run<15>("v_cndmask_b32", w);
    }
    return 0;
}
