// rt_api.hip — the C ABI of librt_hip.so (include/rt_hip.h): scene flattening + upload, and the
// host side of the wavefront loop.  No CPU render path exists here: every compute entry point
// needs a HIP device and fails with RT_ERR_NO_DEVICE / RT_ERR_HIP otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bvh_build.h"
#include "rt_kernels.h"
#include "rt_layout.h"

static_assert(sizeof(rt_sphere) == 64 && sizeof(rt_triangle) == 152 && sizeof(rt_material) == 32 &&
                  sizeof(rt_texture) == 64 && sizeof(rt_sky) == 144 && sizeof(rt_camera) == 184 &&
                  sizeof(rt_opts) == 40 && sizeof(rt_stats) == 176 && sizeof(rt_ray) == 48 && sizeof(rt_hit) == 16 &&
                  sizeof(rt_scene_info) == 72,
              "ABI struct sizes (mirrored in rt_amd/_abi.py)");

using namespace rt;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (void)hipGetLastError();                                                             \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorName(e_), __FILE__, __LINE__);                                \
        }                                                                                        \
    } while (0)

constexpr int MAX_DEVICES = 64;
std::mutex g_device_mu[MAX_DEVICES];

// Every entry point that needs a device makes it current through a DeviceGuard, which puts the caller's
// current device back on return: a single-process multi-GPU host (or torch with another current device)
// keeps its own selection across calls into the library.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) { (void)hipSetDevice(prev); } }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(RT_ERR_NO_DEVICE, "no HIP device available (%s); librt_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorName(e));
    }
    if (device < 0 || device >= n || device >= MAX_DEVICES)
        return fail(RT_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    return RT_OK;
}

constexpr int MAX_PIPES = 8;

// ---- debug / test switches ------------------------------------------------------------------------------------------
// The environment is read in ONE place, into this struct: when a scene is created (builder, node format, walk mode) and
// again at the start of every entry point that traces or renders — once per call, never per pass or per launch — so that
// a test can flip a switch between two calls on the same scene.  None of them changes a result; the tests that compare
// the variants bit for bit are named beside each switch.
struct Tuning {
    // scene creation
    std::string builder;          // RT_HIP_BUILDER=lbvh: the device builder (tests/test_gpu_parity.py: both builders, same hits)
    std::string node_boxes;       // RT_HIP_NODE_BOXES=f64|f32|q16|w4|w4f|w8: force a node format (test_gpu_parity.py: every format)
    int all_lanes_load = -1;      // RT_HIP_ALL_LANES_LOAD=0|1: every-lane-fetches walk off / on (default: off)
    int groups = -1;              // RT_HIP_GROUPS=0|1: two-level build over the scene's separate meshes off / on (default: by scene)
    int bvh8 = -1;                // RT_HIP_BVH8=0|1: the BVH8 as the default format of the per-lane walk off / on (default: by scene size)
    // per call
    bool no_packet = false;       // RT_HIP_NO_PACKET: camera rays through the per-lane walk (test_gpu_configs.py: kernel variants)
    bool no_fuse = false;         // RT_HIP_NO_FUSE: depth-0 shading in k_shade instead of the camera-ray kernel
    bool primary_packet = false;  // RT_HIP_PRIMARY_PACKET: rt_trace_primary through the coherent-wave kernel
    bool debug = false;           // RT_HIP_DEBUG_UTIL: diagnostics on stderr
    bool stream_probe = true;     // RT_HIP_STREAM_PROBE=0: do not check that the pipes' streams sit on queues of their own
    int pipes = 0;                // RT_HIP_PIPES=<1..8>: passes in flight (default 4)
    int tail_depth = 0;           // RT_HIP_TAIL_DEPTH=<d>: depth from which the persistent kernel takes over (default: adaptive)
    int stagger = 2;              // RT_HIP_STAGGER=0|1|2 (default 2)
    int group = 0;                // RT_HIP_GROUP=1|4|16|64: cap of the samples of a pixel that share a wave
    int paths_log2 = 0;           // RT_HIP_PATHS_IN_FLIGHT=<log2>: default cap of the paths of a pass
    int hw_queues = 4;            // GPU_MAX_HW_QUEUES (HIP's own variable; 4 is its default)
    bool no_refill = false;       // RT_HIP_NO_REFILL=1: one chunk per wave instead of persistent walk launches
    int refill_slots = 0;         // RT_HIP_REFILL_SLOTS=<n>: number of persistent waves
    int ovf_lds_rows = 0;         // RT_HIP_OVF_LDS_ROWS=<4..32>: LDS rows of the overflow-stack walk
};
Tuning read_tuning() {
    auto str = [](const char* n) { const char* e = getenv(n); return std::string(e ? e : ""); };
    auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e && *e ? atoi(e) : dflt; };
    auto set = [](const char* n) { return getenv(n) != nullptr; };
    Tuning t;
    t.builder = str("RT_HIP_BUILDER");
    t.node_boxes = str("RT_HIP_NODE_BOXES");
    t.all_lanes_load = set("RT_HIP_ALL_LANES_LOAD") ? (num("RT_HIP_ALL_LANES_LOAD", 0) != 0 ? 1 : 0) : -1;
    t.groups = set("RT_HIP_GROUPS") ? (num("RT_HIP_GROUPS", 0) != 0 ? 1 : 0) : -1;
    t.bvh8 = set("RT_HIP_BVH8") ? (num("RT_HIP_BVH8", 0) != 0 ? 1 : 0) : -1;
    t.no_packet = set("RT_HIP_NO_PACKET");
    t.no_fuse = set("RT_HIP_NO_FUSE");
    t.primary_packet = set("RT_HIP_PRIMARY_PACKET");
    t.debug = set("RT_HIP_DEBUG_UTIL");
    t.stream_probe = num("RT_HIP_STREAM_PROBE", 1) != 0;
    t.pipes = num("RT_HIP_PIPES", 0);
    t.tail_depth = num("RT_HIP_TAIL_DEPTH", 0);
    t.stagger = num("RT_HIP_STAGGER", 2);
    t.group = num("RT_HIP_GROUP", 0);
    t.paths_log2 = num("RT_HIP_PATHS_IN_FLIGHT", 0);
    t.hw_queues = std::max(1, num("GPU_MAX_HW_QUEUES", 4));
    t.no_refill = num("RT_HIP_NO_REFILL", 0) == 1;
    t.refill_slots = std::max(0, num("RT_HIP_REFILL_SLOTS", 0));
    t.ovf_lds_rows = num("RT_HIP_OVF_LDS_ROWS", 0);
    return t;
}

// ---- scratch of rt_trace_closest_noalloc (include/rt_hip.h) -------------------------------------
// The ray queue (o, d: 6 doubles + path id) and the hit buffer (t, bu, bv + primitive id) of one bounded batch live
// in a file-scope device array instead of a hipMalloc'ed workspace: 80 B per queue slot, RT_NOALLOC_MAX_RAYS rays +
// QUEUE_SLACK slots of shard padding = 21.3 MB, held by every GPU the library runs on until the process ends.
// `g_noalloc_scratch` is only a host-side stand-in for the symbol: host code uses the per-device ADDRESS obtained
// once per device with hipGetSymbolAddress (noalloc_scratch_for), never the array's name.
constexpr size_t NOALLOC_SLOTS = (size_t)RT_NOALLOC_MAX_RAYS + QUEUE_SLACK;
constexpr size_t NOALLOC_DOUBLES = 9 * NOALLOC_SLOTS;      // o[3], d[3], t, bu, bv
constexpr size_t NOALLOC_U32S = 2 * NOALLOC_SLOTS;         // path id, primitive id
constexpr size_t NOALLOC_BYTES = NOALLOC_DOUBLES * sizeof(double) + NOALLOC_U32S * sizeof(uint32_t);
static __device__ __align__(256) unsigned char g_noalloc_scratch[NOALLOC_BYTES];
unsigned char* g_noalloc_ptr[MAX_DEVICES] = {};            // device address of the array on each device (under g_device_mu)

// call with `device` current and g_device_mu[device] held
int noalloc_scratch_for(int device, unsigned char** out) {
    if (!g_noalloc_ptr[device]) {
        void* p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch)));
        if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress returned NULL for the no-alloc scratch");
        g_noalloc_ptr[device] = (unsigned char*)p;
    }
    *out = g_noalloc_ptr[device];
    return RT_OK;
}

struct Workspace {
    size_t capacity = 0;          // paths
    char* base = nullptr;         // one device allocation, carved below
    size_t bytes = 0;
    RayQueue q[2];
    HitBuf hb;
    double* path_radiance = nullptr;
    uint32_t* counters = nullptr;                  // 2 sets of NSHARD queue counters, 128 B apart (device only: no host copy exists)
    uint32_t* hist = nullptr;                      // HIST_DEPTHS u32 right behind them: queue sizes the kernels report per pass
    uint32_t* h_hist = nullptr;                    // pinned copy, read when the pipe is reused (launch estimates only)
    double* jitter = nullptr;                      // Halton table of the pass in flight (2 doubles per local sample)
    size_t jitter_cap = 0;                         // in samples
    StackOverflow ovf = {nullptr, 0u, 0u};         // traversal-stack rows beyond the LDS part (deep trees only)
    unsigned long long* stat_counters = nullptr;   // STAT_GROUPS groups of STAT_GROUP u64
    unsigned long long* h_stat = nullptr;          // pinned copy of them
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // phase boundaries of the pass in flight
};

}  // namespace

struct rt_scene {
    int device = 0;
    DScene d{};
    std::vector<void*> allocs;
    rt_scene_info info{};
    std::mutex mu;
    bool device_built = false;       // BVH from bvh_gpu.hip
    Workspace ws[MAX_PIPES];         // one per in-flight pass (DEFAULT_PIPES passes are pipelined by default)
    hipStream_t aux_stream[MAX_PIPES] = {};      // the pipes' streams when several passes are in flight (ensure_pipe_streams)
    int n_aux = 0;                               // how many of them exist
    hipStream_t pad_stream[3] = {};              // idle streams that only shift where HIP puts the pipes' streams (same place)
    hipEvent_t ev_fork = nullptr, ev_acc[MAX_PIPES] = {}, ev_cam[MAX_PIPES] = {};
    // grow-only device buffers for the host-facing entry points (the frame of rt_render, the ray / hit staging of
    // rt_trace_closest and rt_trace_primary): no hipMalloc / hipFree per call once they have their size
    void* io[2] = {nullptr, nullptr};
    size_t io_cap[2] = {0, 0};
    // progressive preview state (rt_render_pass): per-pass sums, the running float frame and the 8-bit linear frame
    double* pv_sum = nullptr;
    double* pv_mean = nullptr;
    uint8_t* pv_rgba8 = nullptr;
    size_t pv_pixels = 0;
    uint64_t pv_total = 0;
    // share of a pass's paths that is still alive at depth d, as the last passes reported it (0 = not known yet): sizes
    // the shading launches and nothing else
    double alive_ratio[HIST_DEPTHS] = {};
    // depth from which the persistent trace-and-shade kernel takes over; follows the reported queue sizes (render_into)
    uint32_t tail_depth = 4;
    // rt_render_multi: this device's radiance sums (f64), their f32 copy for RT_OPT_REDUCE_F32, and the stream the render
    // and the reduce are queued on — grow-only, made on first use, freed with the scene (no hipMalloc / stream per call)
    double* mg_sum = nullptr;
    float* mg_sum32 = nullptr;
    size_t mg_cap = 0, mg_cap32 = 0;       // in values
    hipStream_t mg_stream = nullptr;
    Tuning tune;                     // the switches as the running call read them (refresh_tuning)
    LaunchCtx lctx;                  // what the walk launchers keep for this scene's device
    // RT_OPT_PIPELINED: passes of a frame that was queued but not waited for (render_into).  While `valid`, the pipes'
    // streams may still be running them: every entry point that touches the scene drains them first (drain_pending),
    // except a pipelined render of the same shape, which queues its passes behind them.
    struct Pending {
        bool valid = false;
        int n_pipes = 0;
        uint32_t per_pass = 0, max_depth = 0;
        size_t npix = 0;
        uint32_t spp = 0, cap = 0;                           // what the pass size and the pipe count were derived from,
        int tune_pipes = 0, tune_paths = 0;                  // ... besides npix
        bool want_stats = false;
        int prev_pipe = -1;                                  // pipe of the frame's last pass (its ev_acc ends the frame)
        struct { bool in_flight = false; uint32_t n = 0, tail_depth = 0; } pipe[MAX_PIPES];
    } pend;
};

namespace {

// start of every tracing / rendering entry point, with the device's mutex held
void refresh_tuning(rt_scene* sc) {
    sc->tune = read_tuning();
    sc->lctx.debug = sc->tune.debug;
    sc->lctx.refill = !sc->tune.no_refill;
    sc->lctx.refill_slots = (unsigned)sc->tune.refill_slots;
    const int r = sc->tune.ovf_lds_rows;
    sc->lctx.ovf_lds_rows = r >= 4 && r <= (int)OVF_LDS_ROWS ? (uint32_t)r : OVF_LDS_ROWS;
}

void free_workspace(Workspace& w) {
    if (w.base) (void)hipFree(w.base);
    if (w.h_stat) (void)hipHostFree(w.h_stat);
    if (w.h_hist) (void)hipHostFree(w.h_hist);
    if (w.jitter) (void)hipFree(w.jitter);
    if (w.ovf.rows) (void)hipFree(w.ovf.rows);
    for (auto& e : w.ev)
        if (e) (void)hipEventDestroy(e);
    w = Workspace();
}

constexpr size_t HIST_BYTES = 256;                 // >= HIST_DEPTHS * 4, keeps the alignment of what follows

size_t workspace_bytes(size_t paths) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t slots = paths + QUEUE_SLACK;     // sharded queues pad every shard to a multiple of 64
    const size_t dsz = up(slots * sizeof(double)), usz = up(slots * sizeof(uint32_t));
    return 2 * (9 * dsz + usz) + (3 * dsz + usz) + up(3 * paths * sizeof(double)) + 2 * COUNTER_BYTES + HIST_BYTES + STAT_BYTES;
}

int ensure_workspace(rt_scene* sc, size_t paths, int which = 0) {
    Workspace& w = sc->ws[which];
    if (!w.h_stat) {
        HIPCHK(hipHostMalloc((void**)&w.h_stat, STAT_BYTES, hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void**)&w.h_hist, HIST_BYTES, hipHostMallocDefault));
        memset(w.h_hist, 0, HIST_BYTES);
        for (auto& e : w.ev) HIPCHK(hipEventCreate(&e));
    }
    // deep trees: the rows of the traversal stack beyond its LDS part, a column per lane of a persistent launch
    // (16 waves per CU is what the registers of the traversal kernel allow at most)
    const uint32_t ovf_rows = stack_overflow_rows(sc->d, sc->lctx.ovf_lds_rows);
    if (ovf_rows > w.ovf.rows_cap) {
        if (w.ovf.rows) (void)hipFree(w.ovf.rows);
        w.ovf = {nullptr, 0u, 0u};
        hipDeviceProp_t pr;
        HIPCHK(hipGetDeviceProperties(&pr, sc->device));
        const uint32_t waves = (uint32_t)pr.multiProcessorCount * 16u;
        uint32_t* rows = nullptr;
        if (hipMalloc((void**)&rows, (size_t)ovf_rows * waves * 64u * sizeof(uint32_t)) == hipSuccess) w.ovf = {rows, ovf_rows, waves};
        else (void)hipGetLastError();              // not fatal: the kernels then keep the whole stack in LDS
    }
    if (paths <= w.capacity) return RT_OK;
    if (w.base) {
        (void)hipFree(w.base);
        w.base = nullptr;
        w.capacity = 0;
        w.bytes = 0;
    }
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t slots = paths + QUEUE_SLACK;
    const size_t dsz = up(slots * sizeof(double)), usz = up(slots * sizeof(uint32_t));
    const size_t total = workspace_bytes(paths);
    char* p = nullptr;
    hipError_t e = hipMalloc((void**)&p, total);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(RT_ERR_OOM, "hipMalloc(%zu bytes) for %zu paths in flight failed: %s", total, paths, hipGetErrorName(e));
    }
    w.base = p;
    w.bytes = total;
    auto take = [&](size_t n) { char* r = p; p += n; return r; };
    for (int k = 0; k < 2; ++k) {
        for (int c = 0; c < 3; ++c) w.q[k].o[c] = (double*)take(dsz);
        for (int c = 0; c < 3; ++c) w.q[k].d[c] = (double*)take(dsz);
        for (int c = 0; c < 3; ++c) w.q[k].thr[c] = (double*)take(dsz);
        w.q[k].pid = (uint32_t*)take(usz);
    }
    w.hb.t = (double*)take(dsz);
    w.hb.bu = (double*)take(dsz);
    w.hb.bv = (double*)take(dsz);
    w.hb.prim = (uint32_t*)take(usz);
    w.path_radiance = (double*)take(up(3 * paths * sizeof(double)));
    w.counters = (uint32_t*)take(2 * COUNTER_BYTES);
    w.hist = (uint32_t*)take(HIST_BYTES);
    w.stat_counters = (unsigned long long*)take(STAT_BYTES);
    w.capacity = paths;
    return RT_OK;
}

// Rotation3::from_euler_angles(roll, pitch, yaw) = Rz(yaw) * Ry(pitch) * Rx(roll)
void euler_matrix(double roll, double pitch, double yaw, double m[9]) {
    const double sr = std::sin(roll), cr = std::cos(roll);
    const double sp = std::sin(pitch), cp = std::cos(pitch);
    const double sy = std::sin(yaw), cy = std::cos(yaw);
    m[0] = cy * cp; m[1] = cy * sp * sr - sy * cr; m[2] = cy * sp * cr + sy * sr;
    m[3] = sy * cp; m[4] = sy * sp * sr + cy * cr; m[5] = sy * sp * cr - cy * sr;
    m[6] = -sp;     m[7] = cp * sr;                m[8] = cp * cr;
}
// unit_sphere_uv_facing (src/hittable.rs:392-400) + the matrix product of unit_sphere_uv (:376-377)
void sphere_texture_rotation(const double f[3], double r[9]) {
    const double pitch = std::atan2(f[2], std::sqrt(f[1] * f[1] + f[0] * f[0]));
    const double yaw = std::atan2(f[1], f[0]);
    double a[9], b[9];
    euler_matrix(0.0, pitch, 0.0, a);
    euler_matrix(0.0, 0.0, -yaw, b);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}

// Scenes whose build looks for separate meshes first (RT_HIP_GROUPS=0|1 overrides)
bool default_groups(size_t n_triangles) {
    (void)n_triangles;
    return false;
}

// Scenes that walk the BVH8 by default (RT_HIP_NODE_BOXES=w8 / w4 / ... overrides either way)
bool default_bvh8(size_t n_prims) {
    (void)n_prims;
    return false;
}

// Connected components of the triangle soup (triangles that share a vertex position, bit for bit): the separate objects a
// scene is made of — what the reference's mesh_scene / from_gltf loads one by one (src/hittable.rs:497-633).  Components of
// min_size .. max_size triangles become groups 1 .. n for build_bvh_grouped; everything else (the terrain, loose
// triangles) stays 0.  group_of has one entry per primitive (spheres first).  Returns the number of groups.
uint32_t find_mesh_groups(const rt_triangle* tris, size_t n_tris, size_t n_spheres, size_t min_size, size_t max_size,
                          std::vector<uint32_t>& group_of, size_t& grouped_tris) {
    group_of.assign(n_spheres + n_tris, 0u);
    grouped_tris = 0;
    if (n_tris < 2 * min_size || n_tris >= (size_t(1) << 30)) return 0;
    std::vector<uint32_t> parent(n_tris);
    for (size_t i = 0; i < n_tris; ++i) parent[i] = (uint32_t)i;
    auto find = [&](uint32_t x) {
        while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
        return x;
    };
    size_t cap = 1;
    while (cap < 6 * n_tris) cap <<= 1;
    std::vector<uint32_t> table(cap, 0xFFFFFFFFu);          // corner index (3 * triangle + k) of the first vertex with that position
    auto corner = [&](uint32_t c) -> const double* { const rt_triangle& t = tris[c / 3]; return c % 3 == 0 ? t.a : c % 3 == 1 ? t.b : t.c; };
    for (size_t i = 0; i < n_tris; ++i)
        for (uint32_t k = 0; k < 3; ++k) {
            const uint32_t c = (uint32_t)(3 * i + k);
            const double* v = corner(c);
            uint64_t b[3];
            memcpy(b, v, 24);
            for (uint64_t& w : b) if (w == 0x8000000000000000ull) w = 0;      // -0.0 and +0.0 are the same place
            uint64_t h = b[0] * 0x9E3779B97F4A7C15ull;
            h = (h ^ (h >> 29)) + b[1] * 0xBF58476D1CE4E5B9ull;
            h = (h ^ (h >> 32)) + b[2] * 0x94D049BB133111EBull;
            h ^= h >> 31;
            for (size_t slot = h & (cap - 1);; slot = (slot + 1) & (cap - 1)) {
                const uint32_t o = table[slot];
                if (o == 0xFFFFFFFFu) { table[slot] = c; break; }
                const double* w = corner(o);
                if (w[0] == v[0] && w[1] == v[1] && w[2] == v[2]) {
                    const uint32_t ra = find((uint32_t)i), rb = find(o / 3);
                    if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
                    break;
                }
            }
        }
    std::vector<uint32_t> size(n_tris, 0), id(n_tris, 0);
    for (size_t i = 0; i < n_tris; ++i) size[find((uint32_t)i)]++;
    uint32_t n_groups = 0;
    for (size_t i = 0; i < n_tris; ++i)
        if (size[i] >= min_size && size[i] <= max_size) { id[i] = ++n_groups; grouped_tris += size[i]; }      // (roots only have a size)
    if (!n_groups) return 0;
    for (size_t i = 0; i < n_tris; ++i) group_of[n_spheres + i] = id[find((uint32_t)i)];
    return n_groups;
}

bool finite3(const double* p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

template <class T>
int upload(rt_scene* sc, const std::vector<T>& v, const T** out) {
    void* p = nullptr;
    // 256 bytes behind every array: the traversal kernels read up to 112 bytes from a record's address whatever its kind
    size_t bytes = v.size() * sizeof(T) + 256;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(RT_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorName(e));
    }
    sc->allocs.push_back(p);
    sc->info.device_bytes += bytes;
    if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RT_OK;
}

// sc->io[which] with at least `bytes` bytes (called with the scene's mutex held and its device current)
template <class T>
int scene_io(rt_scene* sc, int which, size_t count, T** out) {
    const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    if (sc->io_cap[which] < bytes) {
        if (sc->io[which]) (void)hipFree(sc->io[which]);
        sc->io[which] = nullptr;
        sc->io_cap[which] = 0;
        hipError_t e = hipMalloc(&sc->io[which], bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            sc->io[which] = nullptr;
            return fail(RT_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorName(e));
        }
        sc->io_cap[which] = bytes;
    }
    *out = (T*)sc->io[which];
    return RT_OK;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int check_camera(const rt_camera* cam) {
    if (!cam) return fail(RT_ERR_INVALID, "camera is NULL");
    if (cam->width == 0 || cam->height == 0) return fail(RT_ERR_INVALID, "camera has zero size");
    if ((uint64_t)cam->width * cam->height > (1ull << 31)) return fail(RT_ERR_LIMIT, "image larger than 2^31 pixels");
    return RT_OK;
}

// 2^26 paths = 32 samples per pixel and pass at 1080p: 204 B of path state each, four workspaces (four passes in
// flight) = 55 GB of the 288 GB.  Large launches amortise the tails, several passes in flight fill what one pass leaves idle.
constexpr size_t DEFAULT_PATHS_IN_FLIGHT = size_t(1) << 26;
constexpr int DEFAULT_PIPES = 4;
constexpr size_t PRIMARY_MAX_PIXELS = size_t(1) << 25;

// ---- streams of the passes in flight ---------------------------------------------------------------------------------
// A HIP process has GPU_MAX_HW_QUEUES hardware queues (4 here) and every stream is put on the one that carries the fewest
// streams at that moment: four streams made in a row sit on four different queues only if the streams the host made
// before (the null stream, torch's RCCL stream, a side stream for copies ...) were spread evenly.  Two pipes on one queue
// run one after the other, and a 256-spp frame takes 4-11 % longer (measured: 71.4 ms with 4 or 8 streams in the process,
// 76.5 / 80.6 / 74.5 ms with 5 / 6 / 7).  So the set-up checks: one wave per pipe stream waits PROBE_US; side by side all are
// done after about PROBE_US, with two of them on one queue after twice that.  When they are not side by side, one more
// idle stream is parked (at most three: with the fourth the count is even again) and the pipes' streams are made anew.
constexpr unsigned PROBE_US = 200;
__global__ void k_stream_probe(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 18); ++i) {            // bounded whatever the clock does (s_sleep 8 = 512 cycles a turn)
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

int pipes_side_by_side(rt_scene* sc, int n, bool* ok) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, sc->device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
    const unsigned long long ticks = (unsigned long long)khz * PROBE_US / 1000ull;
    hipEvent_t start = nullptr, end[MAX_PIPES] = {};
    HIPCHK(hipEventCreate(&start));
    for (int k = 0; k < n; ++k) HIPCHK(hipEventCreate(&end[k]));
    float worst = 0.f;
    for (int round = 0; round < 2; ++round) {        // the first round only warms the streams up (queue creation, code load)
        HIPCHK(hipEventRecord(start, sc->aux_stream[0]));
        for (int k = 0; k < n; ++k) {
            hipLaunchKernelGGL(k_stream_probe, dim3(1), dim3(64), 0, sc->aux_stream[k], ticks);
            HIPCHK(hipEventRecord(end[k], sc->aux_stream[k]));
        }
        for (int k = 0; k < n; ++k) HIPCHK(hipStreamSynchronize(sc->aux_stream[k]));
        worst = 0.f;
        for (int k = 0; k < n; ++k) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, start, end[k]));
            worst = std::max(worst, ms);
        }
    }
    (void)hipEventDestroy(start);
    for (int k = 0; k < n; ++k) (void)hipEventDestroy(end[k]);
    *ok = worst < 1.6f * (float)PROBE_US * 1e-3f;
    if (sc->tune.debug) fprintf(stderr, "[rt_hip] %d pipe streams: probe waves done after %.0f us (%s)\n", n, worst * 1e3f, *ok ? "side by side" : "queued behind each other");
    return RT_OK;
}

int ensure_pipe_streams(rt_scene* sc, int n_pipes) {
    if (n_pipes <= 1 || sc->n_aux >= n_pipes) return RT_OK;
    auto drop = [&]() {                              // every slot, so that a partly failed creation loop is cleaned up too
        for (int k = 0; k < MAX_PIPES; ++k)
            if (sc->aux_stream[k]) { (void)hipStreamDestroy(sc->aux_stream[k]); sc->aux_stream[k] = nullptr; }
        sc->n_aux = 0;
    };
    drop();                                          // (no pass is in flight here: the caller holds the scene)
    // a render that needs more pipes than the one before starts the placement over: the idle streams parked for the old
    // set go too, so that their number is again what the probe below finds necessary
    for (auto& ps : sc->pad_stream)
        if (ps) { (void)hipStreamDestroy(ps); ps = nullptr; }
    const bool probe = n_pipes <= sc->tune.hw_queues && sc->tune.stream_probe;
    for (int attempt = 0;; ++attempt) {
        for (int k = 0; k < n_pipes; ++k) {
            hipError_t e = hipStreamCreateWithFlags(&sc->aux_stream[k], hipStreamNonBlocking);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                sc->aux_stream[k] = nullptr;
                drop();
                return fail(RT_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorName(e));
            }
        }
        sc->n_aux = n_pipes;
        if (!probe || attempt == 3) return RT_OK;
        bool ok = false;
        int rc = pipes_side_by_side(sc, n_pipes, &ok);
        if (rc) return rc;
        if (ok) return RT_OK;
        drop();
        HIPCHK(hipStreamCreateWithFlags(&sc->pad_stream[attempt], hipStreamNonBlocking));
    }
}

// The wavefront loop.  Adds this call's per-pixel radiance sums to d_rgb_sum (device).
//
// Event timings of the pass a pipe has finished (phases: camera rays | first-bounce traversal | its shading | every
// deeper bounce: stand-alone pairs, then the persistent kernel | sums), what it reported about its queues (sizes the
// next launches), and the switch depth of the persistent kernel that follows from them.  Waits for the pass.
constexpr uint32_t TAIL_MAX_PATHS = 1u << 15;
int collect_pass(rt_scene* sc, Workspace& w, bool& in_flight, uint32_t n_paths, uint32_t td, int td_env, uint32_t max_depth, rt_stats& st) {
    if (!in_flight) return RT_OK;
    HIPCHK(hipEventSynchronize(w.ev[5]));
    float ms[5];
    for (int i = 0; i < 5; ++i) HIPCHK(hipEventElapsedTime(&ms[i], w.ev[i], w.ev[i + 1]));
    st.seconds_trace_primary += ms[0] * 1e-3;
    st.seconds_trace_bounce += ms[1] * 1e-3;
    st.seconds_trace += (ms[0] + ms[1]) * 1e-3;
    st.seconds_shade += (ms[2] + ms[4]) * 1e-3;
    st.seconds_tail += ms[3] * 1e-3;
    for (int d = 1; d < HIST_DEPTHS; ++d)          // what this pass reported: the next launches are sized by it
        if (w.h_hist[d] && n_paths) sc->alive_ratio[d] = (double)w.h_hist[d] / (double)n_paths;
    if (td_env <= 0 && td == sc->tail_depth && td < (uint32_t)HIST_DEPTHS) {
        const uint32_t cap = std::min<uint32_t>((uint32_t)HIST_DEPTHS - 1u, std::max(2u, max_depth));
        if (w.h_hist[td] > TAIL_MAX_PATHS && td < cap) {
            uint32_t more = 1;
            double r = 0.5;                                                            // paths that survive a bounce
            if (td >= 2 && w.h_hist[td - 1] > w.h_hist[td]) {
                r = (double)w.h_hist[td] / (double)w.h_hist[td - 1];
                more = (uint32_t)std::ceil(std::log((double)TAIL_MAX_PATHS / (double)w.h_hist[td]) / std::log(r));
            }
            sc->tail_depth = std::min(cap, td + std::max(1u, more));
            // the stand-alone launches of the depths in between are sized by the same extrapolation until they report
            for (uint32_t d = td + 1; d <= sc->tail_depth && d < (uint32_t)HIST_DEPTHS; ++d)
                if (!(sc->alive_ratio[d] > 0.0)) sc->alive_ratio[d] = sc->alive_ratio[d - 1] * r;
        } else if (td > 2 && w.h_hist[td - 1] < TAIL_MAX_PATHS / 4) {
            sc->tail_depth = td - 1;
        }
    }
    in_flight = false;
    return RT_OK;
}

// The pipes' segment counters (and, with collect_stats, the visit counters), read back into h_stat, added to st.
void add_counters(rt_scene* sc, int n_pipes, bool want_stats, rt_stats& st) {
    for (int k = 0; k < n_pipes; ++k) {
        const unsigned long long* h = sc->ws[k].h_stat;
        const unsigned long long* hp = h + STAT_GROUP * STAT_GROUP_PRIMARY;
        const unsigned long long* hb_ = h + STAT_GROUP * STAT_GROUP_BOUNCE;
        const unsigned long long* ht = h + STAT_GROUP * STAT_GROUP_TAIL;
        const unsigned long long* hd = h + STAT_GROUP * STAT_GROUP_DEEP;
        st.bounce_segments += hb_[STAT_SEGMENTS];
        st.tail_segments += ht[STAT_SEGMENTS] + hd[STAT_SEGMENTS];
        st.ray_segments += hb_[STAT_SEGMENTS] + ht[STAT_SEGMENTS] + hd[STAT_SEGMENTS];
        if (want_stats) {
            st.nodes_visited += hp[STAT_NODES] + hb_[STAT_NODES] + ht[STAT_NODES] + hd[STAT_NODES];
            st.prims_tested += hp[STAT_PRIMS] + hb_[STAT_PRIMS] + ht[STAT_PRIMS] + hd[STAT_PRIMS];
            st.nodes_visited_primary += hp[STAT_NODES];
            st.prims_tested_primary += hp[STAT_PRIMS];
            st.nodes_visited_bounce += hb_[STAT_NODES];
            st.prims_tested_bounce += hb_[STAT_PRIMS];
            st.wave_node_fetches_primary += hp[STAT_WAVE_NODES];
            st.wave_prim_fetches_primary += hp[STAT_WAVE_PRIMS];
            if (sc->tune.debug) {      // bounce launches: lane steps / (64 x wave loop iterations)
                fprintf(stderr, "[rt_hip] pipe %d camera rays: %llu node records fetched per wave-launch, %llu of them dead ends "
                        "(no ray entered a child), %llu primitive records\n", k, hp[STAT_WAVE_NODES], hp[STAT_LEAVES], hp[STAT_WAVE_PRIMS]);
                fprintf(stderr, "[rt_hip] pipe %d bounce: node steps %llu, leaf steps %llu, wave iterations %llu -> lane occupancy %.3f\n",
                        k, hb_[STAT_NODES], hb_[STAT_LEAVES], hb_[STAT_ITERS],
                        hb_[STAT_ITERS] ? (double)(hb_[STAT_NODES] + hb_[STAT_LEAVES]) / (64.0 * (double)hb_[STAT_ITERS]) : 0.0);
            }
        }
    }
}

// Waits for the passes a pipelined frame left in flight (rt_scene::Pending) and adds their timings and counters to
// *st (may be null).  With the scene's device current and its mutex held; every entry point that touches the scene's
// workspaces or streams calls it first.  On a HIP error the pipes are still synchronised one by one and the state cleared.
int drain_pending(rt_scene* sc, rt_stats* st_out) {
    if (!sc->pend.valid) return RT_OK;
    rt_scene::Pending& pd = sc->pend;
    pd.valid = false;
    rt_stats st;
    memset(&st, 0, sizeof(st));
    int first_err = RT_OK;
    for (int k = 0; k < pd.n_pipes; ++k) {
        hipError_t e = hipMemcpyAsync(sc->ws[k].h_stat, sc->ws[k].stat_counters, STAT_BYTES, hipMemcpyDeviceToHost, sc->aux_stream[k]);
        if (e == hipSuccess) e = hipStreamSynchronize(sc->aux_stream[k]);
        if (e != hipSuccess && !first_err) {
            (void)hipGetLastError();
            first_err = fail(RT_ERR_HIP, "pipelined render: %s on pipe %d", hipGetErrorName(e), k);
        }
    }
    if (first_err) return first_err;
    for (int k = 0; k < pd.n_pipes; ++k) {
        bool in_flight = pd.pipe[k].in_flight;
        if (int rc = collect_pass(sc, sc->ws[k], in_flight, pd.pipe[k].n, pd.pipe[k].tail_depth, sc->tune.tail_depth, pd.max_depth, st)) return rc;
        pd.pipe[k].in_flight = false;
    }
    add_counters(sc, pd.n_pipes, pd.want_stats, st);
    if (st_out) {
        st_out->seconds_trace_primary += st.seconds_trace_primary;
        st_out->seconds_trace_bounce += st.seconds_trace_bounce;
        st_out->seconds_trace += st.seconds_trace;
        st_out->seconds_shade += st.seconds_shade;
        st_out->seconds_tail += st.seconds_tail;
        st_out->bounce_segments += st.bounce_segments;
        st_out->tail_segments += st.tail_segments;
        st_out->ray_segments += st.ray_segments;
        st_out->nodes_visited += st.nodes_visited;
        st_out->prims_tested += st.prims_tested;
        st_out->nodes_visited_primary += st.nodes_visited_primary;
        st_out->prims_tested_primary += st.prims_tested_primary;
        st_out->nodes_visited_bounce += st.nodes_visited_bounce;
        st_out->prims_tested_bounce += st.prims_tested_bounce;
        st_out->wave_node_fetches_primary += st.wave_node_fetches_primary;
        st_out->wave_prim_fetches_primary += st.wave_prim_fetches_primary;
    }
    return RT_OK;
}

// A pass = per_pass samples of every pixel.  Its kernels are queued on one stream WITHOUT any host round trip:
//   depth 0      k_trace_packet<GEN, FUSE>   camera rays: generate, trace (coherent waves), shade, compact -> queue + counts A
//   depth 1      k_walk                      first bounce: per-lane BVH4 traversal of that queue (the dominant kernel)
//                k_shade                     shade + Russian roulette, compact -> queue + counts B
//   depth >= 2   k_walk<SHADE>               persistent: every remaining bounce of every surviving path inside one launch
//   k_accumulate                             per-pixel sums in sample order (passes ordered by events), clears the counts
// The entry counts of the queues never leave the device (QueueGeom); every consumer reads them there.  Up to four passes are
// in flight on four streams; the host only waits when it reuses a pipe (to read its event timings) and at the end.
// RT_HIP_TAIL_DEPTH=<d> (tests, A/B): depth from which the persistent kernel takes over (default 2; 1 = straight after the
// camera rays; larger = more stand-alone k_walk + k_shade pairs first).  The image does not depend on it.
int render_into(rt_scene* sc, const rt_camera* cam, const rt_opts* opts, double* d_rgb_sum, hipStream_t stream,
                rt_stats* stats) {
    const uint32_t stride = opts->sample_stride ? opts->sample_stride : 1;
    if (opts->spp == 0) return fail(RT_ERR_INVALID, "spp must be >= 1");
    if (opts->precision != 0 && opts->precision != RT_PRECISION_F64)
        return fail(RT_ERR_INVALID, "precision %u not supported (use RT_PRECISION_F64)", opts->precision);
    const uint64_t last = (uint64_t)opts->sample_offset + (uint64_t)(opts->spp - 1) * stride;
    if (last >= RT_MAX_SAMPLE_INDEX)
        return fail(RT_ERR_LIMIT, "sample index %llu exceeds the Halton table limit %u (src/camera.rs:208-210)",
                    (unsigned long long)last, RT_MAX_SAMPLE_INDEX);
    const size_t npix = (size_t)cam->width * cam->height;
    const Tuning& tune = sc->tune;
    rt_stats st;
    memset(&st, 0, sizeof(st));
    const bool want_stats = opts->collect_stats != 0;
    int rc;
    // Passes of a pipelined frame may still be running (below): another pipelined frame of the same shape goes behind
    // them, anything else waits for them first — before any workspace is looked at, let alone outgrown and freed.
    if (sc->pend.valid && !((opts->flags & RT_OPT_PIPELINED) && sc->pend.npix == npix && sc->pend.spp == opts->spp &&
                            sc->pend.cap == opts->max_paths_in_flight && sc->pend.tune_pipes == tune.pipes &&
                            sc->pend.tune_paths == tune.paths_log2 && sc->pend.want_stats == want_stats &&
                            sc->pend.max_depth == cam->max_depth))
        if ((rc = drain_pending(sc, &st))) return rc;
    const int pfe = tune.paths_log2;                               // experiments: log2 of the default cap
    size_t dflt = pfe >= 10 && pfe <= 30 ? size_t(1) << pfe : DEFAULT_PATHS_IN_FLIGHT;
    // Frames of more than 4 M pixels: a default pass would hold fewer than 16 samples per pixel, and 16 is what the layout
    // of a pass is built around (a wave of camera rays = 16 samples of a 2 x 2 pixel block: coherent rays, k_accumulate16).
    // The pass grows to 16 samples per pixel where four such workspaces take at most 40 % of the device's memory (4K:
    // 133 M paths, 4 x 27 GB): config 5's frame (3840 x 2160, 20 M triangles) 3880-3920 -> 4020-4070 Msamples/s.
    // (2^27 paths at 1080p as well: S4 +1.5 %, S3 +-1 % from run to run — not worth 110 GB of workspaces per scene.)
    if (!(pfe >= 10 && pfe <= 30) && 16 * npix > dflt && 16 * npix < (size_t(1) << 32)) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (4 * workspace_bytes(16 * npix) <= total_b / 10 * 4) dflt = 16 * npix;
        } else {
            (void)hipGetLastError();
        }
    }
    // A frame of at most three passes (a rank's share of a strong-scaling job, a preview) cannot keep four
    // passes in flight: with passes of half the size (and half again) the tail of one hides behind the bulk of the next again.  1080p,
    // Msamples/s with 2^26 / 2^25 / 2^24 paths per pass: 32 spp 6010 / 6680 / 6350, 64 spp 7060 / 7270 / 6740, 128 spp 7550 /
    // 7570 / 6940, 256 spp 7750 / 7640 / 6900.  (The image does not depend on the pass size: sums are in sample order.)
    // Pipelined frames (below) of 16 / 32 / 64 / 128 spp: 6930 / 7330 / 7850 / 8000 with this rule; 64 spp in two passes of 32: 7560.
    const bool want_pipelined = (opts->flags & RT_OPT_PIPELINED) != 0;
    while (!(pfe >= 10 && pfe <= 30) && npix * (size_t)opts->spp <= 3 * dflt && (dflt >> 1) >= std::max<size_t>(16 * npix, size_t(1) << 25))
        dflt >>= 1;      // (never below 16 samples per pixel, nor below the 2^25 paths that fill the GPU)
    const size_t cap = opts->max_paths_in_flight ? opts->max_paths_in_flight : dflt;
    uint32_t per_pass = (uint32_t)std::max<size_t>(1, std::min<size_t>(opts->spp, cap / npix));
    if (npix * per_pass >= (1ull << 32)) return fail(RT_ERR_LIMIT, "too many paths in flight");
    // Several passes are kept in flight, one stream each: the tail of a pass (few paths, long dependent chains) hides
    // behind the other passes' bulk work.  RT_HIP_PIPES=<n> sets the number (1 disables the overlap).
    const int pe = tune.pipes;
    const uint32_t want_pipes = pe >= 1 ? (uint32_t)std::min(pe, MAX_PIPES) : (uint32_t)DEFAULT_PIPES;
    uint32_t n_passes = (opts->spp + per_pass - 1) / per_pass;
    // (a pipelined frame of fewer passes than pipes still gets every pipe: the frames behind it go on where it stopped)
    int n_pipes = want_pipelined ? (int)want_pipes : (int)std::min<uint32_t>(n_passes, want_pipes);
    // Workspaces: n_pipes x (per_pass x pixels) paths.  When the device cannot hold them (a shared or smaller-HBM GPU) the
    // render goes on with fewer passes in flight, then with smaller passes, instead of failing.
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // what the workspaces would take beyond what this scene already holds (an outgrown one is freed first)
            auto extra = [&](int pipes_, uint32_t per_) {
                size_t add = 0;
                for (int k = 0; k < pipes_; ++k)
                    if (sc->ws[k].capacity < npix * per_) add += workspace_bytes(npix * per_) - std::min(workspace_bytes(npix * per_), sc->ws[k].bytes);
                return add;
            };
            // (smaller passes first, down to the 2^25 paths that still fill the GPU, then fewer passes in flight)
            while (extra(n_pipes, per_pass) > free_b - free_b / 32 && !(per_pass == 1 && n_pipes == 1)) {
                if (per_pass > 1 && npix * (size_t)per_pass > (size_t(1) << 25)) per_pass = (per_pass + 1) / 2;
                else if (n_pipes > 1) --n_pipes;
                else per_pass = (per_pass + 1) / 2;
            }
        } else {
            (void)hipGetLastError();
        }
        for (;;) {
            int got = 0;
            rc = RT_OK;
            for (int k = 0; k < n_pipes && !rc; ++k)
                if (!(rc = ensure_workspace(sc, npix * per_pass, k))) ++got;
            if (!rc) break;
            if (rc != RT_ERR_OOM) return rc;
            if (got >= 1) { n_pipes = got; break; }
            if (per_pass == 1) return rc;
            per_pass = (per_pass + 1) / 2;
        }
        n_passes = (opts->spp + per_pass - 1) / per_pass;
        if (!want_pipelined) n_pipes = (int)std::min<uint32_t>(n_passes, (uint32_t)n_pipes);
    }
    for (int k = 0; k < n_pipes; ++k) {          // per-pass Halton tables (16 B per sample of a pass; optional: a failed allocation only disables them)
        Workspace& w = sc->ws[k];
        if (w.jitter_cap < per_pass) {
            if (w.jitter) (void)hipFree(w.jitter);
            w.jitter = nullptr;
            w.jitter_cap = 0;
            if (hipMalloc((void**)&w.jitter, (size_t)per_pass * 2 * sizeof(double)) == hipSuccess) w.jitter_cap = per_pass;
            else (void)hipGetLastError();
        }
    }
    if (!sc->ev_fork) {
        HIPCHK(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
        for (auto& e : sc->ev_acc) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : sc->ev_cam) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // (One stream per pass in flight and no more: beyond four hardware queues per process the queues are time-sliced on
    // this platform and a render stalls for seconds, so the short phases of a pass get no high-priority side stream.)
    // With several passes in flight every pipe runs on a stream of the scene's own, placed on hardware queues of their own
    // (ensure_pipe_streams): with pipe 0 on the caller's stream, whatever other streams the host had made decided whether
    // two pipes shared a queue.
    if ((rc = ensure_pipe_streams(sc, n_pipes))) return rc;
    // several passes in flight: the 96-register walk (rt_kernels.hip: k_walk5); the other entry points launch one walk at a time
    struct FiveGuard {
        LaunchCtx& c;
        ~FiveGuard() { c.five = false; }
    } five_guard{sc->lctx};
    sc->lctx.five = n_pipes > 1;

    PassParams pp0;
    pp0.cam = *cam;
    pp0.seed = opts->seed;
    pp0.n_pixels = (uint32_t)npix;
    pp0.sample_offset = opts->sample_offset;
    pp0.sample_stride = stride;
    pp0.pass_base = 0;
    pp0.jitter = 1;
    pp0.tiled = 1;
    pp0.group = 1;
    pp0.jitter_table = nullptr;
    // samples of a pixel that share a wave of camera rays: as many as the pass and the frame allow
    // (RT_HIP_GROUP=1|4|16|64 caps it; 1 = one sample of an 8x8 tile per wave)
    const int ge = tune.group;
    const uint32_t group_cap = ge > 0 ? (uint32_t)ge : 16u;

    const bool use_packet = !tune.no_packet && packet_supported(sc->d);
    const bool fuse_shade = use_packet && !tune.no_fuse;      // depth-0 shading inside the camera-ray kernel
    // Depth from which the persistent trace-and-shade kernel takes over.  Stand-alone k_walk + k_shade launches are the
    // efficient way through a queue of any size (full occupancy, half the registers), the persistent kernel is the way
    // through the few thousand paths that are left after some bounces (one launch, no matter how deep they go): the
    // switch follows the sizes the passes report: it goes to the first depth whose queue has at most TAIL_MAX_PATHS paths —
    // read from the reported sizes where a stand-alone pair ran, extrapolated with the last decay ratio beyond them (the
    // persistent kernel reports only the size it starts with) — and comes back one depth when the last stand-alone pair
    // saw fewer than a quarter of that.  RT_HIP_TAIL_DEPTH=<d> fixes it (tests, A/B).  The image does not depend on it.
    const int td_env = tune.tail_depth;
    auto tail_depth_now = [&]() -> uint32_t { return td_env > 0 ? (uint32_t)td_env : sc->tail_depth; };
    const int stagger_mode = tune.stagger;
    const bool stagger = stagger_mode != 0;

    struct Pipe {
        Workspace* w = nullptr;
        hipStream_t s = nullptr;      // the stream every launch of the pipe's passes goes to
        bool in_flight = false;
        uint32_t n = 0;               // paths of the pass in flight
        uint32_t tail_depth = 0;      // and the switch depth it ran with
    } pipes[MAX_PIPES];
    for (int k = 0; k < n_pipes; ++k) {
        pipes[k].w = &sc->ws[k];
        pipes[k].s = n_pipes > 1 ? sc->aux_stream[k] : stream;
    }
    // RT_OPT_PIPELINED (include/rt_hip.h): the frame is queued and not waited for, and a frame of the same shape queued
    // behind it starts on the pipes while the last passes of this one are still in their low-occupancy bounces — the
    // ~1.5 ms at the end of every frame in which the GPU is nearly idle.  Only the sums touch the caller's buffer: the
    // frame's first k_accumulate waits for what the caller had queued on `stream` (the others follow it in pass order),
    // and at the end `stream` is made to wait for the last one.
    const bool pipelined = (opts->flags & RT_OPT_PIPELINED) != 0 && n_pipes > 1;
    if (sc->pend.valid && !(pipelined && sc->pend.n_pipes == n_pipes && sc->pend.per_pass == per_pass))      // (memory ran short meanwhile)
        if ((rc = drain_pending(sc, &st))) return rc;
    const bool continuing = sc->pend.valid;      // (then: same pipes, same pass size, their streams order the workspaces' reuse)
    int prev_pipe = -1;
    uint32_t pipe0 = 0;                          // pipe of this frame's first pass
    if (continuing) {
        for (int k = 0; k < n_pipes; ++k) {
            pipes[k].in_flight = sc->pend.pipe[k].in_flight;
            pipes[k].n = sc->pend.pipe[k].n;
            pipes[k].tail_depth = sc->pend.pipe[k].tail_depth;
        }
        prev_pipe = sc->pend.prev_pipe;
        pipe0 = (uint32_t)(prev_pipe + 1) % (uint32_t)n_pipes;      // the pipe after the one the frame before ended on
        sc->pend.valid = false;                  // (this call owns the passes now; set again when it leaves them in flight)
    }
    if (n_pipes > 1) {   // work already queued on the caller's stream (e.g. zeroing d_rgb_sum) comes first
        HIPCHK(hipEventRecord(sc->ev_fork, stream));
        if (!pipelined)
            for (int k = 0; k < n_pipes; ++k) HIPCHK(hipStreamWaitEvent(sc->aux_stream[k], sc->ev_fork, 0));
    }
    if (!continuing)
        for (int k = 0; k < n_pipes; ++k) {
            // queue counters start at zero (every pass leaves them cleared again: k_accumulate), statistics too
            // (a frame queued behind a pipelined one finds both as it needs them: the statistics run on until they are read)
            HIPCHK(hipMemsetAsync(pipes[k].w->counters, 0, 2 * COUNTER_BYTES + HIST_BYTES, pipes[k].s));
            HIPCHK(hipMemsetAsync(pipes[k].w->stat_counters, 0, STAT_BYTES, pipes[k].s));
        }
    const double t0 = now_s();

    auto collect = [&](Pipe& P) -> int {
        return collect_pass(sc, *P.w, P.in_flight, P.n, P.tail_depth, td_env, cam->max_depth, st);
    };
    auto estimate = [&](uint32_t depth, uint32_t n) -> uint32_t {
        if (depth >= (uint32_t)HIST_DEPTHS || !(sc->alive_ratio[depth] > 0.0)) return 0u;
        return (uint32_t)std::min<double>((double)n, sc->alive_ratio[depth] * (double)n + 1.0);
    };

    // queues one pass on its pipe; a failure in here leaves the other pipes' work in flight: the caller drains them all
    auto queue_pass = [&](uint32_t pass) -> int {
        const int k = (int)((pipe0 + pass) % (uint32_t)n_pipes);
        Pipe& P = pipes[k];
        if (int crc = collect(P)) return crc;        // (also: the pipe's events are free to be recorded again)
        Workspace& w = *P.w;
        const uint32_t base = pass * per_pass;
        const uint32_t nl = std::min(per_pass, opts->spp - base);
        const uint32_t n = (uint32_t)(npix * nl);
        PassParams pp = pp0;
        pp.pass_base = base;
        for (uint32_t g = 64; g > 1; g >>= 2)
            if (g <= group_cap && nl % g == 0 && npix % (64u / g) == 0) { pp.group = g; break; }
        if (w.jitter_cap >= nl) {
            pp.jitter_table = w.jitter;
            launch_jitter_table(pp, nl, w.jitter, P.s);
            st.launches++;
        }
        finish_pass_params(pp);
        const uint32_t scap = shard_cap_for(n);
        const QueueGeom dense = {nullptr, scap, n};
        uint32_t* cnt[2] = {w.counters, w.counters + COUNTER_BYTES / 4};
        unsigned long long* sg_primary = w.stat_counters + STAT_GROUP * STAT_GROUP_PRIMARY;
        unsigned long long* sg_bounce = w.stat_counters + STAT_GROUP * STAT_GROUP_BOUNCE;
        unsigned long long* sg_tail = w.stat_counters + STAT_GROUP * STAT_GROUP_TAIL;
        unsigned long long* sg_deep = w.stat_counters + STAT_GROUP * STAT_GROUP_DEEP;

        // Camera chain: a pass starts when the pass before it (on the neighbouring pipe) has traced its camera rays, so the
        // pipes run out of phase — the coherent camera rays of one pass share the GPU with the incoherent bounces of the
        // others, and two camera-ray kernels never compete.  RT_HIP_STAGGER=2 (default): every pass waits; =1: only the
        // first pass of every pipe in a frame (a ramp, after which the pipes run free); =0: no waiting.  With pipelined
        // frames, Msamples/s for 0 / 1 / 2: S3 7770 / 7980 / 8035, S4 2355 / 2377 / 2390, S5 - / 3842 / 3909, S1 - / 7528 / 7508.
        if (stagger && prev_pipe >= 0 && prev_pipe != k && (pass < (uint32_t)n_pipes || stagger_mode == 2))
            HIPCHK(hipStreamWaitEvent(P.s, sc->ev_cam[prev_pipe], 0));
        HIPCHK(hipEventRecord(w.ev[0], P.s));
        // ---- depth 0: survivors -> q[1], counts cnt[0]
        if (use_packet) {
            PacketFuse f = {w.q[1], cnt[0], w.path_radiance};
            launch_trace_packet(sc->d, &pp, w.q[0], dense, cam->t_min, cam->t_max, w.hb, want_stats, sg_primary, P.s, fuse_shade ? &f : nullptr);
            st.launches++;
        } else {
            launch_raygen(pp, dense, w.q[0], P.s);
            launch_walk(sc->lctx, sc->d, w.q[0], dense, n, cam->t_min, cam->t_max, w.hb, want_stats, sg_primary, P.s, 0, &w.ovf);
            st.launches += 2;
        }
        if (!fuse_shade) {
            launch_shade(sc->d, pp, 0u, w.q[0], dense, n, 0u, w.hb, w.q[1], cnt[0], w.path_radiance, nullptr, P.s);
            st.launches++;
        }
        st.trace_launches++;
        HIPCHK(hipEventRecord(w.ev[1], P.s));
        HIPCHK(hipEventRecord(sc->ev_cam[k], P.s));
        // ---- depths 1 .. tail_depth-1: stand-alone traversal + shading launches
        int cur = 1, cset = 0;
        uint32_t depth = 1;
        bool ev2 = false;
        hipStream_t ws = P.s;
        const uint32_t tail_depth = tail_depth_now();
        for (; depth < tail_depth && depth <= cam->max_depth; ++depth) {
            const QueueGeom qg = {cnt[cset], scap, 0u};
            launch_walk(sc->lctx, sc->d, w.q[cur], qg, n, cam->t_min, cam->t_max, w.hb, want_stats, depth == 1 ? sg_bounce : sg_deep, ws, estimate(depth, n), &w.ovf);
            st.trace_launches++;
            if (depth == 1) st.bounce_launches++;
            if (!ev2) { HIPCHK(hipEventRecord(w.ev[2], ws)); ev2 = true; }
            if (depth >= 2) HIPCHK(hipMemsetAsync(cnt[cset ^ 1], 0, COUNTER_BYTES, ws));      // (set last read two kernels ago)
            const uint32_t est = estimate(depth, n);
            launch_shade(sc->d, pp, depth, w.q[cur], qg, n, est, w.hb, w.q[cur ^ 1], cnt[cset ^ 1], w.path_radiance,
                         depth < (uint32_t)HIST_DEPTHS ? w.hist + depth : nullptr, ws);
            st.launches += est && est < n ? 3 : 2;
            if (depth == 1) HIPCHK(hipEventRecord(w.ev[3], ws));
            cur ^= 1;
            cset ^= 1;
        }
        if (!ev2) HIPCHK(hipEventRecord(w.ev[2], ws));
        if (depth == 1) HIPCHK(hipEventRecord(w.ev[3], ws));      // (no stand-alone pair ran)
        // ---- every remaining bounce inside one persistent launch
        if (depth <= cam->max_depth) {
            const QueueGeom qg = {cnt[cset], scap, 0u};
            launch_walk_shade(sc->lctx, sc->d, pp, depth, w.q[cur], qg, n, cam->t_min, cam->t_max, w.hb, w.path_radiance,
                              depth < (uint32_t)HIST_DEPTHS ? w.hist + depth : nullptr, want_stats, sg_tail, ws, estimate(depth, n), &w.ovf);
            st.launches++;
        }
        HIPCHK(hipMemcpyAsync(w.h_hist, w.hist, HIST_DEPTHS * sizeof(uint32_t), hipMemcpyDeviceToHost, ws));
        HIPCHK(hipEventRecord(w.ev[4], ws));
        // ---- radiance sums are added pass by pass in pass order (bit-reproducible), whichever pipe is first
        if (prev_pipe >= 0 && prev_pipe != k) HIPCHK(hipStreamWaitEvent(ws, sc->ev_acc[prev_pipe], 0));
        if (pipelined && pass == 0) HIPCHK(hipStreamWaitEvent(ws, sc->ev_fork, 0));      // the caller's queued work (see above)
        launch_accumulate(pp, w.path_radiance, nl, d_rgb_sum, w.counters, (uint32_t)((2 * COUNTER_BYTES + HIST_BYTES) / 4), ws);
        HIPCHK(hipEventRecord(sc->ev_acc[k], ws));
        HIPCHK(hipEventRecord(w.ev[5], ws));
        HIPCHK(hipGetLastError());
        st.launches++;
        st.passes++;
        st.samples += (uint64_t)n;
        st.primary_segments += n;
        P.in_flight = true;
        P.n = n;
        P.tail_depth = tail_depth;
        prev_pipe = k;
        return RT_OK;
    };
    int loop_rc = RT_OK;
    std::string loop_err;
    for (uint32_t pass = 0; pass < n_passes && !loop_rc; ++pass)
        if ((loop_rc = queue_pass(pass))) loop_err = g_err;
    if (pipelined && !loop_rc) {
        // The frame is complete when its last k_accumulate is (the sums run in pass order, and a pass's sum is its last
        // kernel): the caller's stream waits for that one, and the passes stay in flight for the next call to find.
        hipError_t e = hipStreamWaitEvent(stream, sc->ev_acc[prev_pipe], 0);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) {
            rt_scene::Pending& pd = sc->pend;
            pd.valid = true;
            pd.n_pipes = n_pipes; pd.per_pass = per_pass; pd.max_depth = cam->max_depth; pd.npix = npix;
            pd.spp = opts->spp; pd.cap = opts->max_paths_in_flight; pd.tune_pipes = tune.pipes; pd.tune_paths = tune.paths_log2;
            pd.want_stats = want_stats; pd.prev_pipe = prev_pipe;
            for (int k = 0; k < n_pipes; ++k) {
                pd.pipe[k].in_flight = pipes[k].in_flight; pd.pipe[k].n = pipes[k].n; pd.pipe[k].tail_depth = pipes[k].tail_depth;
            }
            st.seconds_total = now_s() - t0;          // (host time to queue the frame)
            st.ray_segments += st.primary_segments;   // (the bounce counters come with the call that waits: rt_scene_sync or any other)
            if (stats) *stats = st;
            return RT_OK;
        }
        (void)hipGetLastError();
        loop_rc = fail(RT_ERR_HIP, "render: %s", hipGetErrorName(e));
        loop_err = g_err;
    }
    // Drain EVERY pipe before returning, whatever happened while queueing or on another pipe: the kernels in flight read
    // and write the scene's workspaces and the caller's sum buffer, and the next call on this scene reuses both.
    int first_err = RT_OK;
    for (int k = 0; k < n_pipes; ++k) {
        hipError_t e = loop_rc ? hipSuccess
                               : hipMemcpyAsync(pipes[k].w->h_stat, pipes[k].w->stat_counters, STAT_BYTES, hipMemcpyDeviceToHost, pipes[k].s);
        if (e == hipSuccess) e = hipStreamSynchronize(pipes[k].s);
        if (e != hipSuccess && !first_err) {
            (void)hipGetLastError();
            first_err = fail(RT_ERR_HIP, "render: %s on pipe %d", hipGetErrorName(e), k);
        }
    }
    if (loop_rc) {                           // the first failure is the one reported
        (void)hipGetLastError();
        g_err = loop_err;
        return loop_rc;
    }
    if (first_err) return first_err;
    for (int k = 0; k < n_pipes; ++k)
        if ((rc = collect(pipes[k]))) return rc;
    HIPCHK(hipGetLastError());
    st.seconds_total = now_s() - t0;
    if (tune.debug) {
        fprintf(stderr, "[rt_hip] tail depth %u; queue sizes by depth (last pass of pipe 0):", sc->tail_depth);
        for (int d = 1; d < HIST_DEPTHS && pipes[0].w->h_hist[d]; ++d) fprintf(stderr, " %u", pipes[0].w->h_hist[d]);
        fprintf(stderr, "\n");
    }
    st.ray_segments += st.primary_segments;      // (+=: a drained pipelined frame may have left its counters here)
    add_counters(sc, n_pipes, want_stats, st);
    if (stats) *stats = st;
    return RT_OK;
}

// The body of rt_trace_closest_noalloc: with the scene's device current and its mutex held, queues the three launches
// on `stream`.  Calls nothing that allocates once the scene's creation has warmed it up.
int trace_noalloc_queued(rt_scene* sc, const rt_ray* d_rays, size_t n, double t_min, double t_max, rt_hit* d_hits, hipStream_t stream) {
    int rc;
    unsigned char* base = nullptr;                      // the device address kept at set-up (rt_scene_create)
    if ((rc = noalloc_scratch_for(sc->device, &base))) return rc;
    // carve the array: every kernel below gets these pointers as arguments, like the workspace of rt_trace_closest
    double* dp = (double*)base;
    uint32_t* up = (uint32_t*)(base + NOALLOC_DOUBLES * sizeof(double));
    RayQueue q{};
    HitBuf hb{};
    for (int c = 0; c < 3; ++c) q.o[c] = dp + (size_t)c * NOALLOC_SLOTS;
    for (int c = 0; c < 3; ++c) q.d[c] = dp + (size_t)(3 + c) * NOALLOC_SLOTS;
    for (int c = 0; c < 3; ++c) q.thr[c] = nullptr;     // the traversal never touches the throughput
    hb.t = dp + 6 * NOALLOC_SLOTS;
    hb.bu = dp + 7 * NOALLOC_SLOTS;
    hb.bv = dp + 8 * NOALLOC_SLOTS;
    q.pid = up;
    hb.prim = up + NOALLOC_SLOTS;
    // Nothing has to start at zero: k_prepare_rays writes every queue slot the traversal reads, the traversal writes t and
    // prim of every one of them (bu / bv are not read by k_pack_hits), and no counter is involved (stats are off).
    const QueueGeom qg = {nullptr, shard_cap_for(n), (uint32_t)n};
    launch_prepare_rays(d_rays, qg, q, stream);
    launch_walk(sc->lctx, sc->d, q, qg, (uint32_t)n, t_min, t_max, hb, false, nullptr, stream);
    launch_pack_hits(hb, qg, d_hits, stream);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

}  // namespace

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
const char* rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(int* count) {
    if (!count) return fail(RT_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        return fail(RT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorName(e));
    }
    *count = n;
    return RT_OK;
}

int rt_scene_create(const rt_sphere* spheres, size_t n_spheres, const rt_triangle* triangles, size_t n_triangles,
                    const rt_material* materials, size_t n_materials, const rt_texture* textures, size_t n_textures,
                    const uint8_t* texel_pool, size_t texel_pool_bytes, const rt_sky* sky, int device, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!sky) return fail(RT_ERR_INVALID, "sky is NULL");
    if ((n_spheres && !spheres) || (n_triangles && !triangles) || (n_materials && !materials) ||
        (n_textures && !textures) || (texel_pool_bytes && !texel_pool))
        return fail(RT_ERR_INVALID, "NULL array with non-zero count");
    const size_t n_prims = n_spheres + n_triangles;
    if (n_prims >= REF_FIRST_MASK) return fail(RT_ERR_LIMIT, "more than 2^28-1 primitives");
    // ---- validate references (a bad index would be an out-of-bounds read on the GPU)
    for (size_t i = 0; i < n_textures; ++i) {
        const rt_texture& t = textures[i];
        if (t.kind > RT_TEX_IMAGE) return fail(RT_ERR_INVALID, "texture %zu: bad kind %u", i, t.kind);
        if (t.kind == RT_TEX_CHECKER && (t.even >= i || t.odd >= i))
            return fail(RT_ERR_INVALID, "texture %zu: checker children must precede it", i);
        if (t.kind == RT_TEX_IMAGE) {
            if (t.texel_format > RT_TEXEL_RGBA16) return fail(RT_ERR_INVALID, "texture %zu: bad texel format", i);
            const size_t bpt = t.texel_format == RT_TEXEL_RGBA16 ? 8 : 4;
            // (no product or sum that could wrap: width * height fits 64 bits, the rest is divided out)
            if (!t.width || !t.height || (t.texel_offset % bpt) != 0 || t.texel_offset > (uint64_t)texel_pool_bytes ||
                (uint64_t)t.width * t.height > ((uint64_t)texel_pool_bytes - t.texel_offset) / bpt)
                return fail(RT_ERR_INVALID, "texture %zu: image outside the texel pool", i);
        }
    }
    for (size_t i = 0; i < n_materials; ++i) {
        const rt_material& m = materials[i];
        if (m.kind > RT_MAT_DIELECTRIC) return fail(RT_ERR_INVALID, "material %zu: bad kind %u", i, m.kind);
        if (m.kind != RT_MAT_DIELECTRIC && m.texture >= n_textures)
            return fail(RT_ERR_INVALID, "material %zu: texture %u out of range", i, m.texture);
    }
    for (size_t i = 0; i < n_spheres; ++i) {
        if (spheres[i].material >= n_materials) return fail(RT_ERR_INVALID, "sphere %zu: material out of range", i);
        if (!finite3(spheres[i].center) || !std::isfinite(spheres[i].radius) || spheres[i].radius < 0.0)
            return fail(RT_ERR_INVALID, "sphere %zu: non-finite or negative geometry", i);
    }
    try {   // (in pieces on the build threads; the first offending triangle is the one reported, as a plain loop would)
        constexpr size_t PIECE = 65536;
        const size_t n_pieces = (n_triangles + PIECE - 1) / PIECE;
        std::vector<size_t> first_bad(n_pieces, SIZE_MAX);
        for_each_piece(n_pieces, [&](size_t piece) {
            for (size_t i = piece * PIECE; i < std::min(n_triangles, (piece + 1) * PIECE); ++i)
                if (triangles[i].material >= n_materials || !finite3(triangles[i].a) || !finite3(triangles[i].b) || !finite3(triangles[i].c)) {
                    first_bad[piece] = i;
                    break;
                }
        });
        for (size_t i : first_bad) {
            if (i == SIZE_MAX) continue;
            if (triangles[i].material >= n_materials) return fail(RT_ERR_INVALID, "triangle %zu: material out of range", i);
            return fail(RT_ERR_INVALID, "triangle %zu: non-finite vertex", i);
        }
    } catch (const std::exception& e) {
        return fail(RT_ERR_OOM, "scene validation failed: %s", e.what());
    }
    DeviceGuard dg;
    int rc = select_device(device);
    if (rc) return rc;
    {   // set-up of the allocation-free entry point: this device's address of the scratch array, once per device
        std::lock_guard<std::mutex> dl(g_device_mu[device]);
        unsigned char* scratch = nullptr;
        if ((rc = noalloc_scratch_for(device, &scratch))) return rc;
    }

    rt_scene* sc = new (std::nothrow) rt_scene();
    if (!sc) return fail(RT_ERR_OOM, "out of host memory");
    sc->device = device;
    refresh_tuning(sc);
    struct Guard {
        rt_scene* s;
        ~Guard() { if (s) rt_scene_destroy(s); }
    } guard{sc};

    try {
        // ---- bounds (Bounded::aabb: src/hittable.rs:242-248, :299-306) and BVH
        const double tb0 = now_s();
        std::vector<PrimBox> boxes(n_prims);
        for (size_t i = 0; i < n_spheres; ++i)
            for (int k = 0; k < 3; ++k) {
                boxes[i].lo[k] = spheres[i].center[k] - spheres[i].radius;
                boxes[i].hi[k] = spheres[i].center[k] + spheres[i].radius;
            }
        // (the per-triangle loops of this function run in pieces on the build threads: at 20 M triangles they were a
        // third of the time to the first pixel)
        constexpr size_t TRI_PIECE = 65536;
        const size_t tri_pieces = (n_triangles + TRI_PIECE - 1) / TRI_PIECE;
        auto tri_piece = [&](size_t piece, size_t& i0, size_t& i1) { i0 = piece * TRI_PIECE; i1 = std::min(n_triangles, i0 + TRI_PIECE); };
        for_each_piece(tri_pieces, [&](size_t piece) {
            size_t i0, i1;
            tri_piece(piece, i0, i1);
            for (size_t i = i0; i < i1; ++i) {
                const rt_triangle& t = triangles[i];
                for (int k = 0; k < 3; ++k) {
                    boxes[n_spheres + i].lo[k] = std::min(std::min(t.a[k], t.b[k]), t.c[k]);
                    boxes[n_spheres + i].hi[k] = std::max(std::max(t.a[k], t.b[k]), t.c[k]);
                }
            }
        });
        BvhResult bvh;
        // RT_HIP_BUILDER=lbvh: Morton-order build on the device (milliseconds; lower tree quality); default: binned SAH
        // on the host.  The device build declines tiny scenes and trees deeper than the traversal stacks.
        bool built = false;
        if (sc->tune.builder == "lbvh") {
            std::string berr;
            double phase_s[2] = {0.0, 0.0};
            const int brc = build_bvh_device(boxes, n_spheres, bvh, berr, phase_s);
            if (brc < 0) return fail(RT_ERR_HIP, "device BVH build: %s", berr.c_str());
            if (sc->tune.debug) fprintf(stderr, "[rt_hip] device builder: bounds %.3f s, upload + kernels + download %.3f s, host finish %.3f s\n", now_s() - tb0 - phase_s[0] - phase_s[1], phase_s[0], phase_s[1]);
            built = brc == 0;
        }
        sc->device_built = built;
        if (!built) {
            // Scenes made of many separate meshes (RT_HIP_GROUPS=1, or by default from GROUPS_MIN_TRIS triangles on when a
            // fifth of them sit in such meshes): a tree per mesh under a top tree over the meshes' boxes.
            std::vector<uint32_t> group_of;
            size_t grouped = 0;
            uint32_t n_groups = 0;
            const bool want = sc->tune.groups >= 0 ? sc->tune.groups != 0 : default_groups(n_triangles);
            if (want) n_groups = find_mesh_groups(triangles, n_triangles, n_spheres, 16, 65536, group_of, grouped);
            if (n_groups && (sc->tune.groups > 0 || grouped * 5 >= n_triangles)) build_bvh_grouped(boxes, n_spheres, group_of, n_groups, bvh);
            else build_bvh(boxes, n_spheres, bvh);
            if (sc->tune.debug) fprintf(stderr, "[rt_hip] builder: %u mesh groups holding %zu of %zu triangles\n", n_groups, grouped, n_triangles);
        }
        if (bvh.depth > (uint32_t)BVH_MAX_DEPTH)
            return fail(RT_ERR_LIMIT, "internal: BVH depth %u exceeds the traversal stack (%d)", bvh.depth, BVH_MAX_DEPTH);
        if (bvh.tri_order.size() != n_triangles || bvh.sph_order.size() != n_spheres)
            return fail(RT_ERR_INVALID, "internal: BVH lost primitives");
        boxes.clear();
        boxes.shrink_to_fit();
        const double tb_bvh = now_s();

        // ---- flatten
        std::vector<TriGeom> tg(n_triangles);
        std::vector<TriShade> tsh(n_triangles);
        for_each_piece(tri_pieces, [&](size_t piece) {
            size_t i0, i1;
            tri_piece(piece, i0, i1);
            for (size_t k = i0; k < i1; ++k) {
                const rt_triangle& t = triangles[bvh.tri_order[k]];
                for (int c = 0; c < 3; ++c) {
                    tg[k].a[c] = t.a[c];
                    tg[k].e1[c] = t.b[c] - t.a[c];           // a_to_b (src/hittable.rs:412)
                    tg[k].e2[c] = t.c[c] - t.a[c];           // a_to_c (:413)
                }
                tg[k].prim = (uint32_t)(n_spheres + bvh.tri_order[k]);
                tg[k].kind = PRIM_KIND_TRIANGLE;
            }
            for (size_t i = i0; i < i1; ++i) {
                const rt_triangle& t = triangles[i];
                for (int c = 0; c < 3; ++c) tsh[i].n[c] = t.n[c];
                // f64::min/max chain (src/hittable.rs:469-476)
                const double left = std::fmin(std::fmin(t.uv[0][0], t.uv[1][0]), t.uv[2][0]);
                const double right = std::fmax(std::fmax(t.uv[0][0], t.uv[1][0]), t.uv[2][0]);
                const double bot = std::fmin(std::fmin(t.uv[0][1], t.uv[1][1]), t.uv[2][1]);
                const double top = std::fmax(std::fmax(t.uv[0][1], t.uv[1][1]), t.uv[2][1]);
                tsh[i].uv_left = left; tsh[i].uv_width = right - left;
                tsh[i].uv_bot = bot; tsh[i].uv_height = top - bot;
                tsh[i].material = t.material;
                tsh[i]._pad = 0;
            }
        });
        std::vector<SphShade> ssh(n_spheres);
        for (size_t i = 0; i < n_spheres; ++i) {
            const rt_sphere& s = spheres[i];
            for (int c = 0; c < 3; ++c) ssh[i].c[c] = s.center[c];
            ssh[i].r = s.radius;
            sphere_texture_rotation(s.face, ssh[i].rot);
            ssh[i].material = s.material;
            ssh[i]._pad = 0;
        }
        std::vector<SphGeom> sg(n_spheres);
        for (size_t k = 0; k < n_spheres; ++k) {
            const uint32_t i = bvh.sph_order[k];
            for (int c = 0; c < 3; ++c) {
                sg[k].c[c] = spheres[i].center[c];
                sg[k].rot2[c] = ssh[i].rot[6 + c];
            }
            sg[k].r = spheres[i].radius;
            sg[k].prim = i;
            for (uint32_t& w : sg[k]._pad) w = 0;
            sg[k].kind = PRIM_KIND_SPHERE;
        }
        const double tb1 = now_s();

        // ---- upload
        std::vector<ShadeMat> mats(n_materials);
        for (size_t i = 0; i < n_materials; ++i) {
            const rt_material& m = materials[i];
            ShadeMat& r = mats[i];
            memset(&r, 0, sizeof(r));
            r.kind = m.kind; r.has_fuzz = m.has_fuzz ? 1u : 0u; r.fuzz = m.fuzz; r.ior = m.ior;
            r.texture = m.texture;
            r.tex_kind = RT_TEX_SOLID;
            if (m.kind != RT_MAT_DIELECTRIC) {
                r.tex_kind = textures[m.texture].kind;
                for (int c = 0; c < 3; ++c) r.rgb[c] = textures[m.texture].rgb[c];
            }
        }
        std::vector<rt_texture> texs(textures, textures + n_textures);
        std::vector<uint8_t> pool(texel_pool, texel_pool + texel_pool_bytes);
        // Node format: conservative f32 boxes (default; 64 B/node) or exact f64 boxes (128 B/node).
        // RT_HIP_NODE_BOXES=f64 forces the latter (the two must give identical hits; tests compare).
        // Coordinates beyond f32 range also fall back to f64 boxes.
        const std::string& nb = sc->tune.node_boxes;
        bool use32 = nb != "f64";
        for (const Node64& nd : bvh.nodes)
            for (int k = 0; k < 3 && use32; ++k) {
                const double m = std::max(std::max(std::fabs(nd.lo_l[k]), std::fabs(nd.hi_l[k])),
                                          nd.right == REF_EMPTY ? 0.0 : std::max(std::fabs(nd.lo_r[k]), std::fabs(nd.hi_r[k])));
                if (!(m < 1e30)) use32 = false;
            }
        // ---- BVH8 (Node8): the per-lane walk's format when asked for (RT_HIP_NODE_BOXES=w8) or by default (below).  Its
        // record pool holds the nodes AND the primitives, in the order of its own layout; the BVH2 arrays (camera rays)
        // share the primitives, so their leaf references are re-expressed as pool records first.
        sc->d.pool8 = nullptr;
        sc->d.depth8 = 0;
        bool use8 = false;
        size_t n_nodes8 = 0;
        if (use32 && n_prims && (nb == "w8" || (nb.empty() && (sc->tune.bvh8 >= 0 ? sc->tune.bvh8 != 0 : default_bvh8(n_prims))))) {
            Wide8 w8;
            if (widen_nodes8(bvh.nodes, n_triangles, n_spheres, w8)) {
                static_assert(sizeof(Node8) == PRIM_RECORD_BYTES && sizeof(TriGeom) == PRIM_RECORD_BYTES && sizeof(SphGeom) == PRIM_RECORD_BYTES,
                              "one stride for every record of the BVH8 pool");
                std::vector<Node8> pool8(w8.n_records);
                constexpr size_t NODE_PIECE8 = 16384;
                for_each_piece((w8.nodes.size() + NODE_PIECE8 - 1) / NODE_PIECE8, [&](size_t piece) {
                    for (size_t i = piece * NODE_PIECE8; i < std::min(w8.nodes.size(), (piece + 1) * NODE_PIECE8); ++i) pool8[w8.node_rec[i]] = w8.nodes[i];
                });
                for_each_piece(tri_pieces, [&](size_t piece) {
                    size_t i0, i1;
                    tri_piece(piece, i0, i1);
                    for (size_t k = i0; k < i1; ++k) memcpy(&pool8[w8.tri_rec[k]], &tg[k], sizeof(TriGeom));
                });
                for (size_t k = 0; k < n_spheres; ++k) memcpy(&pool8[w8.sph_rec[k]], &sg[k], sizeof(SphGeom));
                // BVH2 leaf references: `first` = the pool record of the leaf's first primitive (the rest follow it)
                constexpr size_t N2_PIECE = 32768;
                for_each_piece((bvh.nodes.size() + N2_PIECE - 1) / N2_PIECE, [&](size_t piece) {
                    auto fix = [&](uint32_t ref) -> uint32_t {
                        if (ref == REF_EMPTY || !(ref & REF_LEAF)) return ref;
                        const uint32_t first = ref & REF_FIRST_MASK;
                        return (ref & ~REF_FIRST_MASK) | ((ref & REF_SPHERE) ? w8.sph_rec[first] : w8.tri_rec[first]);
                    };
                    for (size_t i = piece * N2_PIECE; i < std::min(bvh.nodes.size(), (piece + 1) * N2_PIECE); ++i) {
                        bvh.nodes[i].left = fix(bvh.nodes[i].left);
                        bvh.nodes[i].right = fix(bvh.nodes[i].right);
                    }
                });
                const Node8* pool_d = nullptr;
                if ((rc = upload(sc, pool8, &pool_d))) return rc;
                sc->d.pool8 = reinterpret_cast<const char*>(pool_d);
                sc->d.depth8 = w8.depth;
                sc->d.tris = reinterpret_cast<const TriGeom*>(pool_d);
                sc->d.sphs = reinterpret_cast<const SphGeom*>(pool_d);
                n_nodes8 = w8.nodes.size();
                use8 = true;
                if (sc->tune.debug) fprintf(stderr, "[rt_hip] BVH8: %zu nodes from %zu BVH2 nodes, %zu records, %u levels\n", n_nodes8, bvh.nodes.size(), w8.n_records, w8.depth);
            } else if (sc->tune.debug) {
                fprintf(stderr, "[rt_hip] BVH8: not representable for this scene, falling back\n");
            }
        }
        sc->d.nodes64 = nullptr;
        sc->d.nodes32 = nullptr;
        sc->d.nodes16 = nullptr;
        sc->d.nodes4 = nullptr;
        sc->d.nodes4f = nullptr;
        sc->d.stack4 = 0;
        bool use16 = false, use4 = false, use4f = false;
        size_t n_nodes4 = 0;
        std::vector<Node4> n4;          // uploaded below, as the head of the BVH4 walk's record pool
        std::vector<Node4f> n4f;
        if (use32) {
            std::vector<Node32> n32;
            narrow_nodes(bvh.nodes, n32);
            if ((rc = upload(sc, n32, &sc->d.nodes32))) return rc;
            // 16-bit grid nodes for the per-lane (incoherent) traversal, when the grid is fine
            // enough for this scene (RT_HIP_NODE_BOXES=f32 disables; =q16 forces the grid and keeps the
            // BVH2; =w4 forces the grid and the BVH4), and on the same grid the BVH4 that bounce rays use
            std::vector<Node16> n16;
            const bool force4 = nb == "w4";
            const bool force16 = nb == "q16" || force4, no16 = nb == "f32";
            if (!use8 && !no16 && quantize_nodes(n32, n16, sc->d.grid, force16 ? 1e300 : 1.25)) {
                for (int k = 0; k < 3; ++k) sc->d.grid_inv_cell[k] = 1.0 / (double)sc->d.grid.cell[k];
                if ((rc = upload(sc, n16, &sc->d.nodes16))) return rc;
                use16 = true;
                uint32_t need = 0;
                if ((!force16 || force4) && widen_nodes(bvh.nodes, sc->d.grid, n4, need) && need <= (uint32_t)BVH4_MAX_STACK) {
                    sc->d.stack4 = need;
                    n_nodes4 = n4.size();
                    use4 = true;
                }
                if (sc->tune.debug) {
                    fprintf(stderr, "[rt_hip] BVH4: %zu nodes from %zu BVH2 nodes, stack need %u, %s\n", n4.size(), bvh.nodes.size(), need,
                            use4 ? "used" : "not used");
                }
            }
            // no usable grid: the BVH4 with f32 boxes (RT_HIP_NODE_BOXES=f32 keeps the BVH2 for A/B, =w4f forces this one)
            const bool force4f = nb == "w4f";
            if (!use8 && ((!use16 && !no16) || force4f)) {
                uint32_t need = 0;
                if (widen_nodes_f32(bvh.nodes, n4f, need) && need <= (uint32_t)BVH4_MAX_STACK) {
                    sc->d.stack4 = need;
                    n_nodes4 = n4f.size();
                    use4 = true;
                    use4f = true;
                }
            }
        } else {
            if ((rc = upload(sc, bvh.nodes, &sc->d.nodes64))) return rc;
        }
        // ---- the BVH4 walk's record pool: [nodes4 or nodes4f | tris | sphs] in one allocation, child references of the
        // BVH4 nodes re-expressed as pool offsets in 16-byte units (DScene::pool4).  A pool beyond 4 GB (some 40 M
        // triangles) cannot be addressed with 28 bits: such a scene is walked through its BVH2.
        sc->d.pool4 = nullptr;
        if (use4) {
            const size_t node_bytes = use4f ? n4f.size() * sizeof(Node4f) : n4.size() * sizeof(Node4);
            const size_t tri_off = (node_bytes + 127) & ~size_t(127), sph_off = tri_off + tg.size() * sizeof(TriGeom);
            const size_t total = sph_off + sg.size() * sizeof(SphGeom);
            if (total + 4096 >= (size_t(1) << 32)) {
                use4 = use4f = false;
                sc->d.stack4 = 0;
                n_nodes4 = 0;
            } else {
                const uint32_t node_units = (uint32_t)((use4f ? sizeof(Node4f) : sizeof(Node4)) / 16);
                const uint32_t tri_units = (uint32_t)(sizeof(TriGeom) / 16), sph_units = (uint32_t)(sizeof(SphGeom) / 16);
                const uint32_t tri_base = (uint32_t)(tri_off / 16), sph_base = (uint32_t)(sph_off / 16);
                const size_t n_nodes = use4f ? n4f.size() : n4.size();
                bool ok = true;
                auto recode = [&](uint32_t ref) -> uint32_t {
                    if (ref == REF_EMPTY) return ref;
                    const uint64_t first = ref & REF_FIRST_MASK;
                    uint64_t unit;
                    if (!(ref & REF_LEAF)) { ok = ok && first < n_nodes; unit = first * node_units; }
                    else if (ref & REF_SPHERE) { ok = ok && first + ((ref >> 28) & 3u) < sg.size(); unit = sph_base + first * sph_units; }
                    else { ok = ok && first + ((ref >> 28) & 3u) < tg.size(); unit = tri_base + first * tri_units; }
                    ok = ok && unit <= REF_FIRST_MASK - 8u;
                    return (ref & ~REF_FIRST_MASK) | (uint32_t)(unit & REF_FIRST_MASK);
                };
                if (use4f) { for (auto& nd : n4f) for (auto& c : nd.child) c = recode(c); }
                else { for (auto& nd : n4) for (auto& c : nd.child) c = recode(c); }
                if (!ok) return fail(RT_ERR_INVALID, "internal: a BVH4 child reference points outside its arrays");
                char* pool_d = nullptr;
                hipError_t e = hipMalloc((void**)&pool_d, total + 256);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(RT_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", total + 256, hipGetErrorName(e));
                }
                sc->allocs.push_back(pool_d);
                sc->info.device_bytes += total + 256;
                if (tri_off > node_bytes) HIPCHK(hipMemset(pool_d + node_bytes, 0, tri_off - node_bytes));
                HIPCHK(hipMemset(pool_d + total, 0, 256));
                if (use4f) HIPCHK(hipMemcpy(pool_d, n4f.data(), node_bytes, hipMemcpyHostToDevice));
                else HIPCHK(hipMemcpy(pool_d, n4.data(), node_bytes, hipMemcpyHostToDevice));
                if (!tg.empty()) HIPCHK(hipMemcpy(pool_d + tri_off, tg.data(), tg.size() * sizeof(TriGeom), hipMemcpyHostToDevice));
                if (!sg.empty()) HIPCHK(hipMemcpy(pool_d + sph_off, sg.data(), sg.size() * sizeof(SphGeom), hipMemcpyHostToDevice));
                sc->d.pool4 = pool_d;
                if (use4f) sc->d.nodes4f = reinterpret_cast<const Node4f*>(pool_d);
                else sc->d.nodes4 = reinterpret_cast<const Node4*>(pool_d);
                sc->d.tris = reinterpret_cast<const TriGeom*>(pool_d + tri_off);
                sc->d.sphs = reinterpret_cast<const SphGeom*>(pool_d + sph_off);
            }
        }
        if (!sc->d.pool4 && !use8) {
            if ((rc = upload(sc, tg, &sc->d.tris))) return rc;
            if ((rc = upload(sc, sg, &sc->d.sphs))) return rc;
        }
        if ((rc = upload(sc, tsh, &sc->d.tri_shade))) return rc;
        if ((rc = upload(sc, ssh, &sc->d.sph_shade))) return rc;
        if ((rc = upload(sc, mats, &sc->d.mats))) return rc;
        if ((rc = upload(sc, texs, &sc->d.texs))) return rc;
        if ((rc = upload(sc, pool, &sc->d.texels))) return rc;
        sc->d.n_spheres = (uint32_t)n_spheres;
        sc->d.n_prims = (uint32_t)n_prims;
        sc->d.root = n_prims ? 0u : REF_EMPTY;
        sc->d.bvh_depth = bvh.depth;
        sc->d.sky = *sky;
        {
            // Every-lane-fetches walk (walk_body, MODE ALL): +1 % in round 2 on scenes inside the 256 MB Infinity Cache; with this
            // round's leaner walk it costs 0.5-1 % instead (cover scene 7404-7436 with it, 7475-7506 without), so it is off unless
            // RT_HIP_ALL_LANES_LOAD=1 asks for it (the kernel-variant tests do)
            sc->d.all_lanes_load = sc->tune.all_lanes_load > 0 ? 1u : 0u;
        }
        HIPCHK(hipDeviceSynchronize());
        const double tb2 = now_s();
        if (sc->tune.debug)
            fprintf(stderr, "[rt_hip] scene creation: bounds + BVH %.3f s, flatten %.3f s, node formats + upload %.3f s\n", tb_bvh - tb0, tb1 - tb_bvh, tb2 - tb1);
        {   // The allocation-free entry point must not be the first to run its kernels on this device: the first launch of a
            // kernel loads its code object and the first launch of a walk kernel asks the runtime for its occupancy
            // (LaunchCtx::walk_slots), and either may allocate.  One ray through the same three launches, here, where
            // allocating is allowed; the scratch array is scratch.
            std::lock_guard<std::mutex> dl(g_device_mu[device]);
            void* tmp = nullptr;
            HIPCHK(hipMalloc(&tmp, 256));
            struct Free { void* p; ~Free() { (void)hipFree(p); } } fr{tmp};
            HIPCHK(hipMemset(tmp, 0, 256));
            const rt_ray one = {{0.0, 0.0, 0.0}, {0.0, 0.0, 1.0}};
            HIPCHK(hipMemcpy(tmp, &one, sizeof(one), hipMemcpyHostToDevice));
            if ((rc = trace_noalloc_queued(sc, (const rt_ray*)tmp, 1, 0.001, 1.0, (rt_hit*)((char*)tmp + 128), nullptr))) return rc;
            HIPCHK(hipDeviceSynchronize());
        }

        sc->info.n_prims = n_prims;
        sc->info.n_nodes = bvh.nodes.size();
        sc->info.bvh_depth = bvh.depth;
        sc->info.max_leaf_size = bvh.max_leaf;
        sc->info.node_bytes = use8 ? sizeof(Node8) : use4f ? sizeof(Node4f) : use4 ? sizeof(Node4) : use16 ? sizeof(Node16) : (use32 ? sizeof(Node32) : sizeof(Node64));
        sc->info.bvh_width = use8 ? 8 : use4 ? 4 : 2;
        (void)n_nodes8;
        sc->info.builder = sc->device_built ? 1u : 0u;
        (void)n_nodes4;
        sc->info.tri_bytes = sizeof(TriGeom);
        sc->info.sphere_bytes = sizeof(SphGeom);
        sc->info.build_seconds = tb1 - tb0;
        sc->info.upload_seconds = tb2 - tb1;
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_OOM, "out of host memory while flattening the scene");
    } catch (const std::exception& e) {       // (nothing may cross the C ABI: a thread that cannot start, a length error ...)
        return fail(RT_ERR_INVALID, "scene creation failed: %s", e.what());
    }
    guard.s = nullptr;
    *out = sc;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* sc) {
    if (!sc) return;
    DeviceGuard dg;
    if (hipSetDevice(sc->device) == hipSuccess) {
        if (sc->pend.valid)                              // passes of a pipelined frame nobody waited for
            for (int k = 0; k < sc->pend.n_pipes; ++k) (void)hipStreamSynchronize(sc->aux_stream[k]);
        for (auto& w : sc->ws) free_workspace(w);
        for (auto& a : sc->pad_stream)
            if (a) (void)hipStreamDestroy(a);
        for (auto& a : sc->aux_stream)
            if (a) (void)hipStreamDestroy(a);
        if (sc->ev_fork) (void)hipEventDestroy(sc->ev_fork);
        for (auto& e : sc->ev_acc)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : sc->ev_cam)
            if (e) (void)hipEventDestroy(e);
        if (sc->pv_sum) (void)hipFree(sc->pv_sum);
        if (sc->pv_mean) (void)hipFree(sc->pv_mean);
        if (sc->pv_rgba8) (void)hipFree(sc->pv_rgba8);
        for (void* p : sc->io)
            if (p) (void)hipFree(p);
        if (sc->mg_sum) (void)hipFree(sc->mg_sum);
        if (sc->mg_sum32) (void)hipFree(sc->mg_sum32);
        if (sc->mg_stream) (void)hipStreamDestroy(sc->mg_stream);
        for (void* p : sc->allocs) (void)hipFree(p);
    }
    (void)hipGetLastError();
    delete sc;
}

int rt_scene_get_info(const rt_scene* sc, rt_scene_info* out) {
    if (!sc || !out) return fail(RT_ERR_INVALID, "NULL argument");
    *out = sc->info;
    return RT_OK;
}

int rt_trace_closest(const rt_scene* csc, const rt_ray* rays, size_t n, double t_min, double t_max, rt_hit* out,
                     rt_stats* stats) {
    rt_scene* sc = const_cast<rt_scene*>(csc);
    if (!sc) return fail(RT_ERR_INVALID, "scene is NULL");
    if (n && (!rays || !out)) return fail(RT_ERR_INVALID, "rays/out is NULL");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!n) return RT_OK;
    std::lock_guard<std::mutex> dl(g_device_mu[sc->device]);
    std::lock_guard<std::mutex> sl(sc->mu);
    DeviceGuard dg;
    int rc = select_device(sc->device);
    if (rc) return rc;
    refresh_tuning(sc);
    if ((rc = drain_pending(sc, nullptr))) return rc;      // (a pipelined frame's passes hold the workspaces)
    const size_t chunk = std::min<size_t>(n, size_t(1) << 22);
    if ((rc = ensure_workspace(sc, chunk))) return rc;
    Workspace& w = sc->ws[0];
    rt_ray* d_rays = nullptr;
    rt_hit* d_hits = nullptr;
    if ((rc = scene_io(sc, 0, chunk, &d_rays))) return rc;
    if ((rc = scene_io(sc, 1, chunk, &d_hits))) return rc;
    hipStream_t stream = nullptr;
    const bool want_stats = stats != nullptr;
    unsigned long long* sg = w.stat_counters + STAT_GROUP * STAT_GROUP_BOUNCE;
    if (want_stats) HIPCHK(hipMemsetAsync(w.stat_counters, 0, STAT_BYTES, stream));
    const double t0 = now_s();
    for (size_t off = 0; off < n; off += chunk) {
        const uint32_t m = (uint32_t)std::min(chunk, n - off);
        HIPCHK(hipMemcpyAsync(d_rays, rays + off, m * sizeof(rt_ray), hipMemcpyHostToDevice, stream));
        const QueueGeom qg = {nullptr, shard_cap_for(m), m};
        launch_prepare_rays(d_rays, qg, w.q[0], stream);
        HIPCHK(hipEventRecord(w.ev[0], stream));
        launch_walk(sc->lctx, sc->d, w.q[0], qg, m, t_min, t_max, w.hb, want_stats, want_stats ? sg : nullptr, stream, 0, &w.ovf);
        HIPCHK(hipEventRecord(w.ev[1], stream));
        launch_pack_hits(w.hb, qg, d_hits, stream);
        HIPCHK(hipMemcpyAsync(out + off, d_hits, m * sizeof(rt_hit), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipGetLastError());
        if (stats) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, w.ev[0], w.ev[1]));
            stats->seconds_trace += ms * 1e-3;
            stats->trace_launches++;
            stats->launches += 3;
        }
    }
    if (stats) {
        HIPCHK(hipMemcpy(w.h_stat, w.stat_counters, STAT_BYTES, hipMemcpyDeviceToHost));
        const unsigned long long* h = w.h_stat + STAT_GROUP * STAT_GROUP_BOUNCE;
        stats->ray_segments = n;
        stats->bounce_segments = h[STAT_SEGMENTS];
        stats->nodes_visited = stats->nodes_visited_bounce = h[STAT_NODES];
        stats->prims_tested = stats->prims_tested_bounce = h[STAT_PRIMS];
        stats->seconds_trace_bounce = stats->seconds_trace;
        stats->bounce_launches = stats->trace_launches;
        stats->seconds_total = now_s() - t0;
    }
    return RT_OK;
}

int rt_trace_closest_noalloc(const rt_scene* csc, const rt_ray* d_rays, size_t n, double t_min, double t_max,
                             rt_hit* d_hits, void* stream_) {
    rt_scene* sc = const_cast<rt_scene*>(csc);
    if (!sc) return fail(RT_ERR_INVALID, "scene is NULL");
    if (n && (!d_rays || !d_hits)) return fail(RT_ERR_INVALID, "d_rays/d_hits is NULL");
    if (n > (size_t)RT_NOALLOC_MAX_RAYS)
        return fail(RT_ERR_LIMIT, "rt_trace_closest_noalloc: %zu rays exceed the scratch array's limit of %u per call", n,
                    (unsigned)RT_NOALLOC_MAX_RAYS);
    if (!n) return RT_OK;
    std::lock_guard<std::mutex> dl(g_device_mu[sc->device]);
    DeviceGuard dg;
    int rc = select_device(sc->device);
    if (rc) return rc;
    return trace_noalloc_queued(sc, d_rays, n, t_min, t_max, d_hits, (hipStream_t)stream_);
}

int rt_trace_primary(const rt_scene* csc, const rt_camera* cam, uint32_t sample_index, int jitter, uint64_t seed,
                     rt_hit* out, rt_ray* rays_out) {
    rt_scene* sc = const_cast<rt_scene*>(csc);
    if (!sc || !out) return fail(RT_ERR_INVALID, "NULL argument");
    int rc = check_camera(cam);
    if (rc) return rc;
    if (sample_index >= RT_MAX_SAMPLE_INDEX) return fail(RT_ERR_LIMIT, "sample index beyond the Halton table");
    const size_t npix = (size_t)cam->width * cam->height;
    if (npix > PRIMARY_MAX_PIXELS) return fail(RT_ERR_LIMIT, "rt_trace_primary: more than 2^25 pixels");
    std::lock_guard<std::mutex> dl(g_device_mu[sc->device]);
    std::lock_guard<std::mutex> sl(sc->mu);
    DeviceGuard dg;
    if ((rc = select_device(sc->device))) return rc;
    refresh_tuning(sc);
    if ((rc = drain_pending(sc, nullptr))) return rc;      // (a pipelined frame's passes hold the workspaces)
    if ((rc = ensure_workspace(sc, npix))) return rc;
    Workspace& w = sc->ws[0];
    rt_hit* d_hits = nullptr;
    rt_ray* d_rays = nullptr;
    if ((rc = scene_io(sc, 1, npix, &d_hits))) return rc;
    if (rays_out && (rc = scene_io(sc, 0, npix, &d_rays))) return rc;
    PassParams pp;
    pp.cam = *cam;
    pp.seed = seed;
    pp.n_pixels = (uint32_t)npix;
    pp.sample_offset = sample_index;
    pp.sample_stride = 1;
    pp.pass_base = 0;
    pp.jitter = jitter ? 1 : 0;
    pp.tiled = 0;
    pp.group = 1;
    pp.jitter_table = nullptr;
    finish_pass_params(pp);
    hipStream_t stream = nullptr;
    const QueueGeom qm = {nullptr, shard_cap_for(npix), (uint32_t)npix};
    launch_raygen(pp, qm, w.q[0], stream);
    // RT_HIP_PRIMARY_PACKET=1 (tests): the coherent-wave kernel that rt_render uses for camera rays, on the same rays
    if (sc->tune.primary_packet && packet_supported(sc->d))
        launch_trace_packet(sc->d, nullptr, w.q[0], qm, cam->t_min, cam->t_max, w.hb, false, nullptr, stream, nullptr);
    else
        launch_walk(sc->lctx, sc->d, w.q[0], qm, (uint32_t)npix, cam->t_min, cam->t_max, w.hb, false, nullptr, stream, 0, &w.ovf);
    launch_pack_hits(w.hb, qm, d_hits, stream);
    HIPCHK(hipMemcpyAsync(out, d_hits, npix * sizeof(rt_hit), hipMemcpyDeviceToHost, stream));
    if (rays_out) {
        launch_pack_rays(w.q[0], qm, d_rays, stream);
        HIPCHK(hipMemcpyAsync(rays_out, d_rays, npix * sizeof(rt_ray), hipMemcpyDeviceToHost, stream));
    }
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    return RT_OK;
}

int rt_render_accumulate_device(const rt_scene* csc, const rt_camera* cam, const rt_opts* opts, double* d_rgb_sum,
                                void* stream, rt_stats* stats) {
    rt_scene* sc = const_cast<rt_scene*>(csc);
    if (!sc || !opts || !d_rgb_sum) return fail(RT_ERR_INVALID, "NULL argument");
    int rc = check_camera(cam);
    if (rc) return rc;
    if (opts->device != sc->device)
        return fail(RT_ERR_INVALID, "opts.device (%d) differs from the scene's device (%d)", opts->device, sc->device);
    std::lock_guard<std::mutex> dl(g_device_mu[sc->device]);
    std::lock_guard<std::mutex> sl(sc->mu);
    DeviceGuard dg;
    if ((rc = select_device(sc->device))) return rc;
    refresh_tuning(sc);
    return render_into(sc, cam, opts, d_rgb_sum, (hipStream_t)stream, stats);
}

int rt_scene_sync(const rt_scene* csc, rt_stats* stats) {
    rt_scene* sc = const_cast<rt_scene*>(csc);
    if (!sc) return fail(RT_ERR_INVALID, "scene is NULL");
    if (stats) memset(stats, 0, sizeof(*stats));
    std::lock_guard<std::mutex> dl(g_device_mu[sc->device]);
    std::lock_guard<std::mutex> sl(sc->mu);
    DeviceGuard dg;
    if (int rc = select_device(sc->device)) return rc;
    return drain_pending(sc, stats);
}

int rt_resolve_device(const double* d_rgb_sum, double* d_rgb_mean, size_t n_values, uint64_t total_spp, int device,
                      void* stream) {
    if (!d_rgb_sum || !d_rgb_mean) return fail(RT_ERR_INVALID, "NULL argument");
    if (!total_spp) return fail(RT_ERR_INVALID, "total_spp must be >= 1");
    DeviceGuard dg;
    int rc = select_device(device);
    if (rc) return rc;
    launch_resolve(d_rgb_sum, d_rgb_mean, n_values, (double)total_spp, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

int rt_render(const rt_scene* csc, const rt_camera* cam, const rt_opts* opts, double* out_rgb_mean, rt_stats* stats) {
    rt_scene* sc = const_cast<rt_scene*>(csc);
    if (!sc || !opts || !out_rgb_mean) return fail(RT_ERR_INVALID, "NULL argument");
    int rc = check_camera(cam);
    if (rc) return rc;
    if (opts->device != sc->device)
        return fail(RT_ERR_INVALID, "opts.device (%d) differs from the scene's device (%d)", opts->device, sc->device);
    std::lock_guard<std::mutex> dl(g_device_mu[sc->device]);
    std::lock_guard<std::mutex> sl(sc->mu);
    DeviceGuard dg;
    if ((rc = select_device(sc->device))) return rc;
    refresh_tuning(sc);
    const size_t nval = (size_t)cam->width * cam->height * 3;
    double* d_sum = nullptr;
    if ((rc = scene_io(sc, 0, nval, &d_sum))) return rc;
    hipStream_t stream = nullptr;
    HIPCHK(hipMemsetAsync(d_sum, 0, nval * sizeof(double), stream));
    rt_opts o = *opts;
    o.flags &= ~RT_OPT_PIPELINED;                         // (this entry returns the finished image)
    if ((rc = render_into(sc, cam, &o, d_sum, stream, stats))) return rc;
    launch_resolve(d_sum, d_sum, nval, (double)opts->spp, stream);
    HIPCHK(hipMemcpyAsync(out_rgb_mean, d_sum, nval * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    return RT_OK;
}

int rt_encode_gamma8_device(const double* d_rgb_mean, size_t n_values, uint8_t* d_out, int device, void* stream) {
    if (n_values && (!d_rgb_mean || !d_out)) return fail(RT_ERR_INVALID, "NULL argument");
    DeviceGuard dg;
    int rc = select_device(device);
    if (rc) return rc;
    launch_gamma8(d_rgb_mean, n_values, d_out, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

int rt_render_pass(const rt_scene* csc, const rt_camera* cam, uint32_t n_samples, uint64_t total_samples, uint64_t seed,
                   uint32_t flags, uint8_t* out_rgba8, double* out_rgb_mean, rt_stats* stats) {
    rt_scene* sc = const_cast<rt_scene*>(csc);
    if (!sc) return fail(RT_ERR_INVALID, "scene is NULL");
    int rc = check_camera(cam);
    if (rc) return rc;
    if (!n_samples || total_samples < n_samples) return fail(RT_ERR_INVALID, "need 1 <= n_samples <= total_samples");
    if (flags & ~(uint32_t)RT_PASS_QUANTISE) return fail(RT_ERR_INVALID, "unknown flag bits 0x%x", flags);
    std::lock_guard<std::mutex> dl(g_device_mu[sc->device]);
    std::lock_guard<std::mutex> sl(sc->mu);
    DeviceGuard dg;
    if ((rc = select_device(sc->device))) return rc;
    refresh_tuning(sc);
    const size_t npix = (size_t)cam->width * cam->height;
    const bool first = total_samples == n_samples;               // sweep 1: the old frame has weight 0 (src/window.rs:293-295)
    if (sc->pv_pixels != npix) {                                 // frame buffers live with the scene: no allocation per pass
        if (!first) return fail(RT_ERR_INVALID, "rt_render_pass: frame size changed in the middle of a progression (start with total_samples == n_samples)");
        if (sc->pv_sum) (void)hipFree(sc->pv_sum);
        if (sc->pv_mean) (void)hipFree(sc->pv_mean);
        if (sc->pv_rgba8) (void)hipFree(sc->pv_rgba8);
        sc->pv_sum = sc->pv_mean = nullptr; sc->pv_rgba8 = nullptr; sc->pv_pixels = 0;
        HIPCHK(hipMalloc((void**)&sc->pv_sum, npix * 3 * sizeof(double)));
        HIPCHK(hipMalloc((void**)&sc->pv_mean, npix * 3 * sizeof(double)));
        HIPCHK(hipMalloc((void**)&sc->pv_rgba8, npix * 4));
        sc->pv_pixels = npix;
    }
    hipStream_t stream = nullptr;
    if (first) {
        HIPCHK(hipMemsetAsync(sc->pv_mean, 0, npix * 3 * sizeof(double), stream));
        HIPCHK(hipMemsetAsync(sc->pv_rgba8, 0, npix * 4, stream));
    } else if (total_samples - n_samples != sc->pv_total) {
        return fail(RT_ERR_INVALID, "rt_render_pass: total_samples - n_samples = %llu but the frame holds %llu samples",
                    (unsigned long long)(total_samples - n_samples), (unsigned long long)sc->pv_total);
    }
    HIPCHK(hipMemsetAsync(sc->pv_sum, 0, npix * 3 * sizeof(double), stream));
    rt_opts o;
    memset(&o, 0, sizeof(o));
    o.spp = n_samples;                     // render_pixel(x, y, n): Halton points 0 .. n-1 again in every pass (src/window.rs:277)
    o.sample_offset = 0;
    o.sample_stride = 1;
    o.precision = RT_PRECISION_F64;
    o.seed = seed;
    o.device = sc->device;
    if ((rc = render_into(sc, cam, &o, sc->pv_sum, stream, stats))) return rc;
    launch_blend_pass(sc->pv_sum, sc->pv_mean, sc->pv_rgba8, npix, n_samples, total_samples, (flags & RT_PASS_QUANTISE) != 0, stream);
    if (out_rgba8) HIPCHK(hipMemcpyAsync(out_rgba8, sc->pv_rgba8, npix * 4, hipMemcpyDeviceToHost, stream));
    if (out_rgb_mean) HIPCHK(hipMemcpyAsync(out_rgb_mean, sc->pv_mean, npix * 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    sc->pv_total = total_samples;
    return RT_OK;
}

// ---- single-process multi-GPU render: one host thread per device, RCCL sum-reduce over xGMI ----------------------
namespace {
// librccl is loaded on first use with RTLD_LOCAL: hosts that never call rt_render_multi do not depend on it, and a
// process that already holds another copy of RCCL (torch ships its own) keeps its symbols to itself.
struct Rccl {
    void* h = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*Reduce)(const void* send, void* recv, size_t count, int datatype, int op, int root, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
std::mutex g_rccl_mu;
Rccl g_rccl;
struct CommSet { std::vector<int> devices; std::vector<void*> comms; };
std::vector<CommSet> g_comm_sets;      // communicators are kept for the life of the process (creation takes ~a second)

int rccl_load() {
    if (g_rccl.h) return RT_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(RT_ERR_HIP, "rt_render_multi needs RCCL: dlopen(librccl.so.1) failed: %s", dlerror());
    Rccl r;
    r.h = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))dlsym(h, "ncclReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.Reduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        dlclose(h);
        return fail(RT_ERR_HIP, "librccl lacks an expected symbol");
    }
    g_rccl = r;
    return RT_OK;
}
constexpr int NCCL_FLOAT = 7, NCCL_DOUBLE = 8, NCCL_SUM = 0;      // ncclFloat32, ncclFloat64, ncclSum (rccl.h)
}  // namespace

static int render_multi_impl(rt_scene* const* scenes, uint32_t n_devices, const rt_camera* cam, const rt_opts* opts,
                             double* out_rgb_mean, rt_stats* stats);
int rt_render_multi(rt_scene* const* scenes, uint32_t n_devices, const rt_camera* cam, const rt_opts* opts,
                    double* out_rgb_mean, rt_stats* stats) {
    try {
        return render_multi_impl(scenes, n_devices, cam, opts, out_rgb_mean, stats);
    } catch (const std::exception& e) {              // (host allocation of the bookkeeping vectors; nothing may cross the C ABI)
        return fail(RT_ERR_OOM, "rt_render_multi: %s", e.what());
    }
}
static int render_multi_impl(rt_scene* const* scenes, uint32_t n_devices, const rt_camera* cam, const rt_opts* opts,
                             double* out_rgb_mean, rt_stats* stats) {
    if (!scenes || !opts || !out_rgb_mean) return fail(RT_ERR_INVALID, "NULL argument");
    if (n_devices < 1 || n_devices > 64) return fail(RT_ERR_INVALID, "n_devices must be 1..64");
    int rc = check_camera(cam);
    if (rc) return rc;
    if (!opts->spp) return fail(RT_ERR_INVALID, "spp must be >= 1");
    std::vector<int> devs(n_devices);
    for (uint32_t g = 0; g < n_devices; ++g) {
        if (!scenes[g]) return fail(RT_ERR_INVALID, "scenes[%u] is NULL", g);
        devs[g] = scenes[g]->device;
        for (uint32_t j = 0; j < g; ++j)
            if (devs[j] == devs[g]) return fail(RT_ERR_INVALID, "scenes[%u] and scenes[%u] are on the same device %d", j, g, devs[g]);
    }
    DeviceGuard dg;
    // ---- communicators (one per device, created together once per device set)
    std::vector<void*> comms;
    {
        std::lock_guard<std::mutex> rl(g_rccl_mu);
        if ((rc = rccl_load())) return rc;
        for (const CommSet& cs : g_comm_sets)
            if (cs.devices == devs) comms = cs.comms;
        if (comms.empty()) {
            comms.assign(n_devices, nullptr);
            const int r = g_rccl.CommInitAll(comms.data(), (int)n_devices, devs.data());
            if (r != 0) return fail(RT_ERR_HIP, "ncclCommInitAll over %u devices failed: %s", n_devices, g_rccl.GetErrorString(r));
            g_comm_sets.push_back({devs, comms});
        }
    }
    const size_t nval = (size_t)cam->width * cam->height * 3;
    const uint32_t stride0 = opts->sample_stride ? opts->sample_stride : 1;
    if (opts->flags & ~(uint32_t)RT_OPT_REDUCE_F32) return fail(RT_ERR_INVALID, "unknown rt_opts.flags bits 0x%x", opts->flags);
    const bool f32 = (opts->flags & RT_OPT_REDUCE_F32) != 0;
    // device g renders the frame's samples g, g + N, g + 2N, ... (scene replicated, sample-index partition) into the sum
    // buffer its scene owns, on the stream its scene owns; every worker holds its device's and its scene's mutex until the
    // exchange and the read-back are through, so the buffers cannot be reused under a running reduce
    struct PerDev { int rc = RT_OK; std::string err; rt_stats st{}; uint32_t spp = 0; };
    std::vector<PerDev> pd(n_devices);
    std::vector<std::unique_lock<std::mutex>> dev_locks(n_devices), scene_locks(n_devices);
    for (uint32_t g = 0; g < n_devices; ++g) {      // in device order: two concurrent calls cannot deadlock on each other
        std::vector<uint32_t> order(n_devices);
        for (uint32_t j = 0; j < n_devices; ++j) order[j] = j;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return devs[a] < devs[b]; });
        const uint32_t k = order[g];
        dev_locks[k] = std::unique_lock<std::mutex>(g_device_mu[devs[k]]);
        scene_locks[k] = std::unique_lock<std::mutex>(scenes[k]->mu);
    }
    auto worker = [&](uint32_t g) {
        PerDev& P = pd[g];
        rt_scene* sc = scenes[g];
        auto bail = [&](int code) { P.rc = code; P.err = g_err; };
        try {
            if (int r = select_device(sc->device)) return bail(r);
            refresh_tuning(sc);
            if (!sc->mg_stream && hipStreamCreateWithFlags(&sc->mg_stream, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                sc->mg_stream = nullptr;
                return bail(fail(RT_ERR_HIP, "hipStreamCreate failed on device %d", sc->device));
            }
            if (sc->mg_cap < nval) {
                if (sc->mg_sum) (void)hipFree(sc->mg_sum);
                sc->mg_sum = nullptr; sc->mg_cap = 0;
                if (hipMalloc((void**)&sc->mg_sum, nval * sizeof(double)) != hipSuccess) {
                    (void)hipGetLastError();
                    sc->mg_sum = nullptr;
                    return bail(fail(RT_ERR_OOM, "hipMalloc of the radiance sums failed on device %d", sc->device));
                }
                sc->mg_cap = nval;
            }
            if (f32 && sc->mg_cap32 < nval) {
                if (sc->mg_sum32) (void)hipFree(sc->mg_sum32);
                sc->mg_sum32 = nullptr; sc->mg_cap32 = 0;
                if (hipMalloc((void**)&sc->mg_sum32, nval * sizeof(float)) != hipSuccess) {
                    (void)hipGetLastError();
                    sc->mg_sum32 = nullptr;
                    return bail(fail(RT_ERR_OOM, "hipMalloc of the f32 sums failed on device %d", sc->device));
                }
                sc->mg_cap32 = nval;
            }
            if (hipMemsetAsync(sc->mg_sum, 0, nval * sizeof(double), sc->mg_stream) != hipSuccess) { (void)hipGetLastError(); return bail(fail(RT_ERR_HIP, "hipMemsetAsync failed")); }
            P.spp = opts->spp > g ? (opts->spp - g + n_devices - 1) / n_devices : 0;
            if (P.spp) {
                rt_opts o = *opts;
                o.spp = P.spp;
                o.sample_offset = opts->sample_offset + g * stride0;
                o.sample_stride = stride0 * n_devices;
                o.device = sc->device;
                o.flags = 0;
                if (int r = render_into(sc, cam, &o, sc->mg_sum, sc->mg_stream, &P.st)) return bail(r);
            }
            if (f32) {
                launch_narrow_f32(sc->mg_sum, sc->mg_sum32, nval, sc->mg_stream);
                if (hipGetLastError() != hipSuccess) return bail(fail(RT_ERR_HIP, "narrowing the sums failed on device %d", sc->device));
            }
        } catch (const std::exception& e) {          // (nothing may leave a worker thread, nor the C ABI)
            bail(fail(RT_ERR_OOM, "device %d: %s", sc->device, e.what()));
        }
    };
    {
        std::vector<std::thread> th;
        uint32_t started = 1;
        try {
            for (uint32_t g = 1; g < n_devices; ++g, ++started) th.emplace_back(worker, g);
        } catch (const std::system_error&) {         // no thread to be had: the caller's thread takes the remaining devices in turn
        }
        worker(0);
        for (uint32_t g = started; g < n_devices; ++g) worker(g);
        for (auto& t : th) t.join();
    }
    // whatever happened, nothing of this call may still run on a scene's stream when the locks go
    auto drain = [&]() {
        for (uint32_t g = 0; g < n_devices; ++g) {
            if (hipSetDevice(devs[g]) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (scenes[g]->mg_stream) (void)hipStreamSynchronize(scenes[g]->mg_stream);
        }
        (void)hipGetLastError();
    };
    int first_bad = -1;
    for (uint32_t g = 0; g < n_devices; ++g)
        if (pd[g].rc && first_bad < 0) first_bad = (int)g;
    if (first_bad >= 0) {
        drain();
        g_err = "device " + std::to_string(devs[first_bad]) + ": " + pd[first_bad].err;
        return pd[first_bad].rc;
    }
    // ---- the exchange step: sum the per-device buffers into device devs[0] (extends `.sum::<Vec3>()`, src/camera.rs:323)
    {
        std::lock_guard<std::mutex> rl(g_rccl_mu);
        int r = g_rccl.GroupStart();
        for (uint32_t g = 0; g < n_devices && r == 0; ++g) {
            if (hipSetDevice(devs[g]) != hipSuccess) { (void)hipGetLastError(); r = -1; break; }
            rt_scene* sc = scenes[g];
            r = f32 ? g_rccl.Reduce(sc->mg_sum32, sc->mg_sum32, nval, NCCL_FLOAT, NCCL_SUM, 0, comms[g], sc->mg_stream)
                    : g_rccl.Reduce(sc->mg_sum, sc->mg_sum, nval, NCCL_DOUBLE, NCCL_SUM, 0, comms[g], sc->mg_stream);
        }
        const int r2 = g_rccl.GroupEnd();
        if (r == 0) r = r2;
        if (r != 0) {
            drain();
            return fail(RT_ERR_HIP, "ncclReduce failed: %s", r > 0 ? g_rccl.GetErrorString(r) : "hipSetDevice");
        }
    }
    rc = select_device(devs[0]);
    if (!rc) {
        rt_scene* sc = scenes[0];
        if (f32) launch_resolve_f32(sc->mg_sum32, sc->mg_sum, nval, (double)opts->spp, sc->mg_stream);
        else launch_resolve(sc->mg_sum, sc->mg_sum, nval, (double)opts->spp, sc->mg_stream);
        hipError_t e = hipMemcpyAsync(out_rgb_mean, sc->mg_sum, nval * sizeof(double), hipMemcpyDeviceToHost, sc->mg_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(sc->mg_stream);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = fail(RT_ERR_HIP, "read-back of the reduced frame failed: %s", hipGetErrorName(e)); }
    }
    if (stats)
        for (uint32_t g = 0; g < n_devices; ++g) stats[g] = pd[g].st;
    const std::string keep = g_err;
    drain();
    if (rc) g_err = keep;
    return rc;
}

int rt_encode_gamma8(const double* rgb_mean, size_t n_values, uint8_t* out) {
    if (n_values && (!rgb_mean || !out)) return fail(RT_ERR_INVALID, "NULL argument");
    const double gamma = 1.0 / 2.2;                      // as_gamma_vec (src/vec3.rs:39-42)
    for (size_t i = 0; i < n_values; ++i) {
        double c = rgb_mean[i];
        if (!(c >= 0.0)) c = 0.0;                        // the reference panics outside [0,1] (:71-90); clamp instead
        if (c > 1.0) c = 1.0;
        out[i] = (uint8_t)std::round(std::pow(c, gamma) * 255.0);
    }
    return RT_OK;
}

}  // extern "C"
