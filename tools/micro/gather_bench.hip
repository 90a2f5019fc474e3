// gather_bench — what the vector-memory pipe of one MI355X sustains for the access pattern of a per-lane BVH
// traversal: every lane chases its own pointer through a table (dependent loads, 32 B = one Node16 per step).
// Varies the table size (L2 / Infinity Cache / HBM resident), the active lanes per wave and the waves per CU.
// usage: gather_bench            prints one line per configuration: lane-fetches/s in total and per CU per clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BYTES>
__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ table, uint32_t mask, int steps, uint32_t active_lanes,
                                            uint32_t* sink) {
    const uint32_t lane = threadIdx.x;
    uint32_t idx = (blockIdx.x * 64u + lane) * 2654435761u;
    uint32_t acc = 0;
    if (lane < active_lanes) {
        for (int s = 0; s < steps; ++s) {
            const uint4* p = table + (size_t)(idx & mask) * (BYTES / 16);
            uint4 a = p[0];
            acc += a.y;
            if (BYTES >= 32) { uint4 b = p[1]; acc += b.w; }
            if (BYTES >= 64) { uint4 c = p[2], d = p[3]; acc += c.x + d.z; }
            if (BYTES >= 128) { uint4 e = p[4], f = p[5], g = p[6], h = p[7]; acc += e.x + f.y + g.z + h.w; }
            uint32_t h = a.x ^ acc ^ ((uint32_t)s * 0x9E3779B9u);   // next node depends on the loaded data (murmur3 finaliser)
            h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
            idx = h;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int BYTES>
static void run(size_t table_bytes, int waves_per_cu, uint32_t lanes, int cus, double clock_hz) {
    const size_t n = table_bytes / BYTES;                     // power of two
    std::vector<uint32_t> h(table_bytes / 4);
    std::mt19937 gen(12345u);
    for (auto& v : h) v = gen();
    uint4* d; uint32_t* sink;
    CHECK(hipMalloc(&d, table_bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemcpy(d, h.data(), table_bytes, hipMemcpyHostToDevice));
    const int steps = 512, grid = cus * waves_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chase<BYTES>, dim3(grid), dim3(64), 0, 0, d, (uint32_t)(n - 1), steps, lanes, sink);   // warm
    CHECK(hipEventRecord(e0));
    const int reps = 4;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(chase<BYTES>, dim3(grid), dim3(64), 0, 0, d, (uint32_t)(n - 1), steps, lanes, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double fetches = (double)reps * grid * lanes * steps, per_s = fetches / (ms * 1e-3);
    printf("bytes/fetch %3d table %6.0f MiB waves/CU %2d lanes %2u : %8.2f G lane-fetches/s  %6.3f per CU per clk  %7.1f GB/s  step latency %6.0f ns\n",
           BYTES, table_bytes / 1048576.0, waves_per_cu, lanes, per_s / 1e9, per_s / cus / clock_hz, per_s * BYTES / 1e9,
           ms * 1e6 / reps / steps);
    fflush(stdout);
    CHECK(hipFree(d)); CHECK(hipFree(sink));
}

int main(int argc, char** argv) {
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount; const double clk = pr.clockRate * 1e3;
    printf("%s: %d CUs, %.0f MHz\n", pr.name, cus, clk / 1e6);
    const size_t MiB = 1048576;
    if (argc > 3 && std::string(argv[1]) == "calib") {
        // calibration of the FETCH_SIZE counter for random gathers (tools/calib_fetch.sh): ONE configuration, run under
        // rocprofv3 --pmc; prints the number of lane fetches of all its launches (1 warm-up + 4 timed) so that the counter
        // can be set against known bytes: lane fetches x bytes per fetch x (L2 miss share, from TCC_MISS / TCC_REQ)
        const int bytes = atoi(argv[2]);
        const size_t tb = (size_t)atoi(argv[3]) * MiB;
        if (bytes == 32) run<32>(tb, 16, 64, cus, clk); else if (bytes == 64) run<64>(tb, 16, 64, cus, clk); else run<128>(tb, 16, 64, cus, clk);
        printf("calib: bytes_per_fetch %d table_mib %zu lane_fetches_all_launches %.0f\n", bytes == 32 || bytes == 64 ? bytes : 128, tb / MiB,
               5.0 * cus * 16 * 64 * 512);
        return 0;
    }
    if (argc > 1) {      // sweep of the table size: where the fetch rate falls off the per-XCD L2 (4 MiB) and the Infinity Cache
        for (size_t mb : {1, 2, 4, 8, 16, 32, 64, 256}) {
            run<32>(mb * MiB, 16, 64, cus, clk); run<64>(mb * MiB, 16, 64, cus, clk); run<128>(mb * MiB, 16, 64, cus, clk);
        }
        return 0;
    }
    for (size_t tb : {(size_t)16 * MiB, (size_t)128 * MiB, (size_t)1024 * MiB})
        for (int w : {8, 16, 32})
            for (uint32_t lanes : {64u, 32u, 16u}) run<32>(tb, w, lanes, cus, clk);
    for (int w : {16, 32}) { run<16>(16 * MiB, w, 64, cus, clk); run<64>(16 * MiB, w, 64, cus, clk); }
    return 0;
}
