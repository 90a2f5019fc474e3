// valu_rate2 — issue cost of the instructions the traversal kernels are made of, one table: ns of SIMD time per wave64
// instruction with 4 waves launched per SIMD (the walk's occupancy) and 8 independent chains per wave, so that what is
// measured is the issue rate, not the latency; timed by the kernel's duration (HIP events) over a long loop.  Selects, compares (to VCC and to an SGPR pair), min/max, integer and
// bit-field operations, conversions, the f64 division helpers, packed f32, LDS and lane reads, next to v_fma_f32/f64.
// usage: valu_rate2 [waves per SIMD = 4] [repetitions = 2048] [name filter]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef void (*kern_t)(double*, long long*, double, int);

#define K(NAME, STMT)                                                                                              \
    __global__ __launch_bounds__(64) void NAME(double* out, long long* cyc, double seed, int reps) {               \
        __shared__ uint32_t lds[64 * 8];                                                                           \
        double a[8]; float f[8]; uint32_t u[8]; float2 p[8]; float g2[8], h2[8];                                                       \
        for (int i = 0; i < 8; ++i) {                                                                              \
            a[i] = seed + i + threadIdx.x * 1e-3; f[i] = (float)a[i]; u[i] = (uint32_t)(a[i] * 1000.0) + i;        \
            p[i] = make_float2(f[i], f[i] + 1.0f); lds[i * 64 + threadIdx.x] = u[i]; g2[i] = f[i] + 2.0f; h2[i] = f[i] + 3.0f;                               \
        }                                                                                                          \
        const double m = 1.0000001, c = 1e-9;                                                                      \
        const float mf = 1.0000001f, cf = 1e-9f;                                                                   \
        const float2 pm = make_float2(mf, mf);                                                                     \
        const uint32_t x = 0x01020304u + threadIdx.x;                                                              \
        const uint32_t la = threadIdx.x * 4;                                                                       \
        uint64_t sm = __ballot(threadIdx.x & 1);                                                                   \
        (void)m; (void)c; (void)mf; (void)cf; (void)pm; (void)x; (void)la; (void)sm;                                 \
        asm volatile("s_mov_b64 vcc, %0" ::"s"(sm) : "vcc");                                                       \
        const long long t0 = wall_clock64();                                                                       \
        for (int r = 0; r < reps; ++r) {                                                                           \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { STMT; }                                                \
        }                                                                                                          \
        const long long t1 = wall_clock64();                                                                       \
        double s = (double)sm;                                                                                     \
        for (int i = 0; i < 8; ++i) s += a[i] + f[i] + u[i] + p[i].x + p[i].y + g2[i] + h2[i];                                     \
        out[blockIdx.x * 64 + threadIdx.x] = s + lds[threadIdx.x];                                                 \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                           \
    }

// ---- f32 / integer, one VGPR per chain
K(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(mf), "v"(cf)))
K(k_mov_b32, asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(x)))
K(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(x)))
K(k_and_b32, asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(x)))
K(k_lshlrev_b32, asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i])))
K(k_lshl_or_b32, asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(x)))
K(k_and_or_b32, asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(x)))
K(k_bfe_u32, asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i])))
K(k_bfi_b32, asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u[i]) : "v"(x)))
K(k_perm_b32, asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(x)))
K(k_min_u32, asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(x)))
K(k_min_f32, asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(mf)))
K(k_max3_f32, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(mf), "v"(cf)))
K(k_med3_f32, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(mf), "v"(cf)))
K(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(x)))
K(k_mul_hi_u32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(x)))
K(k_mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(x)))
K(k_cvt_f32_u32, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u[i])))
K(k_cvt_f32_ubyte1, asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(u[i])))
K(k_cndmask_vcc, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(x)))
K(k_cndmask_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(x), "s"(sm)))
K(k_cmp_lt_f32_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(f[i]), "v"(mf) : "vcc"))
K(k_cmp_lt_f32_sgpr, asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm) : "v"(f[i]), "v"(mf)))
K(k_cmp_lt_u32_vcc, asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(u[i]), "v"(x) : "vcc"))
K(k_cmp_then_cndmask_f32, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(mf) : "vcc"))
K(k_cmp_then_2cndmask_f32, asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(f[i]), "+v"(u[i]) : "v"(mf), "v"(x) : "vcc"))
K(k_cmpx_lt_f32, asm volatile("v_cmpx_lt_f32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(f[i]), "v"(mf) : "exec"))
// ---- selects in the sequences compilers emit: one compare, several selects on its result
K(k_cmp_4cnd_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %4\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc"
                               : "+v"(f[i]), "+v"(u[i]), "+v"(g2[i]), "+v"(h2[i]) : "v"(mf), "v"(x) : "vcc"))
K(k_cmp_4cnd_sgpr, asm volatile("v_cmp_lt_f32_e64 %4, %0, %5\n\tv_cndmask_b32_e64 %0, %0, %5, %4\n\tv_cndmask_b32_e64 %1, %1, %6, %4\n\tv_cndmask_b32_e64 %2, %2, %5, %4\n\tv_cndmask_b32_e64 %3, %3, %5, %4"
                                : "+v"(f[i]), "+v"(u[i]), "+v"(g2[i]), "+v"(h2[i]), "=&s"(sm) : "v"(mf), "v"(x)))
K(k_cndmask_vcc_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(x)))
K(k_cnd_vcc_then_fma, asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_fma_f32 %1, %1, %3, %3" : "+v"(u[i]), "+v"(f[i]) : "v"(x), "v"(mf)))
K(k_cmp_cnd_fma_cnd_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_fma_f32 %2, %2, %3, %3\n\tv_cndmask_b32 %1, %1, %4, vcc"
                                      : "+v"(f[i]), "+v"(u[i]), "+v"(g2[i]) : "v"(mf), "v"(x) : "vcc"))
K(k_min_max_pair_f32, asm volatile("v_min_f32 %2, %0, %1\n\tv_max_f32 %1, %0, %1\n\tv_mov_b32 %0, %2" : "+v"(f[i]), "+v"(g2[i]), "+v"(h2[i])))
K(k_sand_vcc_then_cnd, asm volatile("s_and_b64 vcc, %2, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(x), "s"(sm) : "vcc", "scc"))
K(k_3cnd_vcc_then_fma, asm volatile("v_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_fma_f32 %2, %2, %4, %4" : "+v"(u[i]), "+v"(f[i]), "+v"(g2[i]) : "v"(x), "v"(mf)))
K(k_2cnd_vcc_then_fma, asm volatile("v_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_fma_f32 %2, %2, %4, %4" : "+v"(u[i]), "+v"(f[i]), "+v"(g2[i]) : "v"(x), "v"(mf)))
K(k_cmp64_6cnd_vcc, asm volatile("v_cmp_lt_f64 vcc, %3, %4\n\tv_cndmask_b32 %0, %0, %5, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc\n\tv_cndmask_b32 %2, %2, %5, vcc" : "+v"(u[i]), "+v"(f[i]), "+v"(g2[i]) : "v"(a[i]), "v"(m), "v"(x) : "vcc"))
// ---- packed f32
K(k_pk_fma_f32, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pm)))
K(k_pk_mul_f32, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm)))
K(k_pk_add_f32, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm)))
// ---- f64
K(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)))
K(k_add_f64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c)))
K(k_mul_f64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m)))
K(k_min_f64, asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m)))
K(k_max_f64, asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m)))
K(k_cmp_lt_f64_vcc, asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(a[i]), "v"(m) : "vcc"))
K(k_cmp_lt_f64_sgpr, asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(sm) : "v"(a[i]), "v"(m)))
K(k_cmp_class_f64, asm volatile("v_cmp_class_f64 vcc, %0, %1" ::"v"(a[i]), "v"(x) : "vcc"))
K(k_rcp_f64, asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i])))
K(k_rsq_f64, asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i])))
K(k_sqrt_f64, asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i])))
K(k_rcp_f32, asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i])))
K(k_sqrt_f32, asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i])))
K(k_div_scale_f64, asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(m) : "vcc"))
K(k_div_fmas_f64, asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)))
K(k_div_fixup_f64, asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)))
K(k_ldexp_f64, asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x)))
K(k_frexp_mant_f64, asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[i])))
K(k_trunc_f64, asm volatile("v_trunc_f64 %0, %0" : "+v"(a[i])))
K(k_floor_f64, asm volatile("v_floor_f64 %0, %0" : "+v"(a[i])))
K(k_fract_f64, asm volatile("v_fract_f64 %0, %0" : "+v"(a[i])))
K(k_cvt_f64_u32, asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(u[i])))
K(k_cvt_u32_f64, asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[i]) : "v"(a[i])))
K(k_cvt_f32_f64, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i])))
K(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i])))
K(k_mul_f64_neg_abs, asm volatile("v_mul_f64 %0, -|%0|, %1" : "+v"(a[i]) : "v"(m)))
K(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a[i]) : "v"(x) : "vcc"))
K(k_lshlrev_b64, asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i])))
// ---- lanes, LDS, scalar
K(k_readfirstlane, uint32_t sl; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sl) : "v"(u[i])); sm += sl)
K(k_readlane, uint32_t sl; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sl) : "v"(u[i])); sm += sl)
K(k_mov_dpp_shr1, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i])))
K(k_ds_bpermute, asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(u[i]) : "v"(la)))
K(k_ds_read_b32, asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(u[i]) : "v"(la) : "memory"))
K(k_ds_read_b32_x8, asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"(la) : "memory"); if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"))
K(k_ds_write_b32_x8, asm volatile("ds_write_b32 %1, %0" ::"v"(u[i]), "v"(la) : "memory"); if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"))
K(k_s_and_b64, asm volatile("s_and_b64 %0, %0, vcc" : "+s"(sm) : : "scc"))
K(k_s_bcnt1, asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(u[i]) : "s"(sm) : "scc"))

struct Entry { const char* name; kern_t fn; };
#define E(n) {#n, n}

int run(const Entry& e, int waves_per_simd, int reps) {
    double* out; long long* cyc;
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int grid = pr.multiProcessorCount * 4 * waves_per_simd;
    CHECK(hipMalloc(&out, sizeof(double) * 64 * grid)); CHECK(hipMalloc(&cyc, sizeof(long long) * grid));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(e.fn, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0, 64);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(e.fn, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0, reps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h[64]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    double ticks = 0; for (long long v : h) ticks += (double)v / 64.0;          // mean over the first 64 waves
    // SIMD time per wave instruction from the KERNEL's duration (every SIMD ran waves_per_simd x reps x 8 entries); the time
    // one wave spent in its loop is printed beside it (the waves of a grid do not all start together, so a wave's own
    // clock readings see fewer neighbours than the grid has and understate the cost)
    const double per_simd = (double)ms * 1e6 / ((double)waves_per_simd * reps * 8.0);
    const double per_wave = ticks / khz * 1e6 / reps / 8.0;
    printf("%-26s %6.2f ns of SIMD time per entry (kernel %8.3f ms); a wave's own loop: %6.2f ns per entry\n", e.name + 2, per_simd, ms, per_wave);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
    return 0;
}

int main(int argc, char** argv) {
    const Entry all[] = {
        E(k_fma_f32), E(k_mov_b32), E(k_add_u32), E(k_and_b32), E(k_lshlrev_b32), E(k_lshl_or_b32), E(k_and_or_b32), E(k_bfe_u32), E(k_bfi_b32),
        E(k_perm_b32), E(k_min_u32), E(k_min_f32), E(k_max3_f32), E(k_med3_f32), E(k_mul_lo_u32), E(k_mul_hi_u32), E(k_mad_u32_u24),
        E(k_cvt_f32_u32), E(k_cvt_f32_ubyte1), E(k_cndmask_vcc), E(k_cndmask_sgpr), E(k_cmp_lt_f32_vcc), E(k_cmp_lt_f32_sgpr),
        E(k_cmp_lt_u32_vcc), E(k_cmp_then_cndmask_f32), E(k_cmp_then_2cndmask_f32), E(k_cmpx_lt_f32), E(k_cmp_4cnd_vcc), E(k_cmp_4cnd_sgpr), E(k_cndmask_vcc_e64), E(k_cnd_vcc_then_fma), E(k_cmp_cnd_fma_cnd_vcc), E(k_min_max_pair_f32), E(k_sand_vcc_then_cnd), E(k_3cnd_vcc_then_fma), E(k_2cnd_vcc_then_fma), E(k_cmp64_6cnd_vcc), E(k_pk_fma_f32), E(k_pk_mul_f32), E(k_pk_add_f32),
        E(k_fma_f64), E(k_add_f64), E(k_mul_f64), E(k_min_f64), E(k_max_f64), E(k_cmp_lt_f64_vcc), E(k_cmp_lt_f64_sgpr), E(k_cmp_class_f64),
        E(k_rcp_f64), E(k_rsq_f64), E(k_sqrt_f64), E(k_rcp_f32), E(k_sqrt_f32), E(k_div_scale_f64), E(k_div_fmas_f64), E(k_div_fixup_f64),
        E(k_ldexp_f64), E(k_frexp_mant_f64), E(k_trunc_f64), E(k_floor_f64), E(k_fract_f64), E(k_cvt_f64_u32), E(k_cvt_u32_f64), E(k_cvt_f32_f64),
        E(k_cvt_f64_f32), E(k_mul_f64_neg_abs), E(k_mad_u64_u32), E(k_lshlrev_b64), E(k_readfirstlane), E(k_readlane), E(k_mov_dpp_shr1),
        E(k_ds_bpermute), E(k_ds_read_b32), E(k_ds_read_b32_x8), E(k_ds_write_b32_x8), E(k_s_and_b64), E(k_s_bcnt1)};
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int waves = argc > 1 ? atoi(argv[1]) : 4;
    printf("SIMD time per wave64 instruction, %d waves per SIMD, 8 independent chains per wave, %d x 8 entries per wave (entries of two or three instructions: per ENTRY)\n", waves, argc > 2 ? atoi(argv[2]) : 16384);
    const int reps = argc > 2 ? atoi(argv[2]) : 16384;            // (long kernels: launch overhead out of the picture)
    for (const Entry& e : all)
        if ((argc <= 3 || strstr(e.name, argv[3])) && run(e, waves, reps)) return 1;
    return 0;
}
