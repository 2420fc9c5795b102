// Where does a write -> read round trip of an intermediate go: HBM or the 256 MiB Infinity Cache?  
//
// rocprofv3's FETCH_SIZE / WRITE_SIZE sit on the L2's fabric side: a hit in the memory-side Infinity Cache counts like a trip to
// HBM.  What does tell them apart is (a) bandwidth -- the cache delivers more than the 6.3 TB/s a copy from HBM reaches -- and (b)
// JOULES: the board is at its power cap in the forward pass, so what a byte costs is what matters.
//
// For a footprint F: kernel W writes F bytes (plain 16-byte stores, as gemm_split_kernel writes zx), kernel R reads them back once
// (non-temporal 16-byte loads, as lstm32_kernel<false> reads zx), alternating for ~`seconds`; time per kernel from HIP events,
// board power from the GPU's hwmon file sampled by a thread of this program.  Modes:
//   wr   : W(F) then R(F)             -- the round trip of the two-launch path: everything written before anything is read
//   lag  : the buffer in chunks of C bytes, R(chunk k - 1) right after W(chunk k)   -- a consumer one chunk behind its producer
//   r    : R(F) only, W once          -- re-reading data that was not just written
// usage: mall_probe [seconds per point = 0.5] [chunk MiB for lag = 16]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void write_kernel(f32x4 *p, size_t n_vec, float v) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) p[i] = (f32x4){v, v + 1.f, v + 2.f, (float)i};
}

__global__ __launch_bounds__(256) void read_kernel(const f32x4 *p, size_t n_vec, float *sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) acc += __builtin_nontemporal_load(p + i);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = 1.f;   // never true: keeps the loads
}

static std::string power_file(int device) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) return "";
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    const std::string base = std::string("/sys/bus/pci/devices/") + bdf + "/hwmon";
    DIR *d = opendir(base.c_str());
    if (!d) return "";
    std::string found;
    while (dirent *e = readdir(d)) {
        if (strncmp(e->d_name, "hwmon", 5)) continue;
        for (const char *leaf : {"power1_average", "power1_input"}) {
            const std::string f = base + "/" + e->d_name + "/" + leaf;
            if (FILE *t = fopen(f.c_str(), "r")) { fclose(t); found = f; break; }
        }
        if (!found.empty()) break;
    }
    closedir(d);
    return found;
}

struct PowerSampler {
    std::string path;
    std::atomic<bool> stop{false};
    std::atomic<long> sum_uw{0}, n{0};
    std::thread th;
    void start() {
        sum_uw = 0; n = 0; stop = false;
        th = std::thread([this] {
            while (!stop) {
                if (FILE *f = fopen(path.c_str(), "r")) {
                    long v = 0;
                    if (fscanf(f, "%ld", &v) == 1) { sum_uw += v / 1000; n += 1; }   // mW
                    fclose(f);
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
        });
    }
    double finish() { stop = true; th.join(); return n ? (double)sum_uw / (double)n / 1e3 : -1.0; }   // W
};

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 0.5;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 16) << 20;
    CHECK(hipSetDevice(0));
    PowerSampler ps;
    ps.path = power_file(0);
    printf("# power from %s\n", ps.path.empty() ? "(no hwmon file found)" : ps.path.c_str());
    const size_t max_bytes = (size_t)2048 << 20;
    f32x4 *buf;
    float *sink;
    CHECK(hipMalloc(&buf, max_bytes));
    CHECK(hipMalloc(&sink, 4));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    const int grid = 256 * 8;
    // idle power first
    double idle_w = -1;
    if (!ps.path.empty()) { CHECK(hipDeviceSynchronize()); ps.start(); std::this_thread::sleep_for(std::chrono::milliseconds(700)); idle_w = ps.finish(); }
    printf("# idle %.0f W\n# mode footprint_MiB  write_GBs read_GBs  power_W  pJ_per_byte_above_idle (bytes = written + read)\n", idle_w);
    const int sizes_mib[] = {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 768, 1024, 2048};
    for (const char *mode : {"wr", "lag", "r"}) {
        for (int mib : sizes_mib) {
            const size_t bytes = (size_t)mib << 20, n_vec = bytes / 16;
            if (!strcmp(mode, "lag") && bytes < 2 * chunk) continue;
            // warm
            hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, st, buf, n_vec, 1.f);
            hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, (const f32x4 *)buf, n_vec, sink);
            CHECK(hipStreamSynchronize(st));
            double w_ms = 0, r_ms = 0, moved = 0;
            int reps = 0;
            if (!ps.path.empty()) ps.start();
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                if (!strcmp(mode, "wr")) {
                    for (int k = 0; k < 4; ++k) {
                        CHECK(hipEventRecord(e0, st));
                        hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, st, buf, n_vec, (float)k);
                        CHECK(hipEventRecord(e1, st));
                        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, (const f32x4 *)buf, n_vec, sink);
                        CHECK(hipEventRecord(e2, st));
                        CHECK(hipEventSynchronize(e2));
                        float a, b;
                        CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
                        w_ms += a; r_ms += b; moved += 2.0 * bytes; ++reps;
                    }
                } else if (!strcmp(mode, "lag")) {
                    const size_t nchunks = bytes / chunk, cvec = chunk / 16;
                    CHECK(hipEventRecord(e0, st));
                    for (size_t k = 0; k <= nchunks; ++k) {
                        if (k < nchunks) hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, st, buf + k * cvec, cvec, (float)k);
                        if (k > 0) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, (const f32x4 *)(buf + (k - 1) * cvec), cvec, sink);
                    }
                    CHECK(hipEventRecord(e2, st));
                    CHECK(hipEventSynchronize(e2));
                    float a;
                    CHECK(hipEventElapsedTime(&a, e0, e2));
                    w_ms += a / 2; r_ms += a / 2; moved += 2.0 * (double)(nchunks * chunk); ++reps;
                } else {
                    for (int k = 0; k < 4; ++k) {
                        CHECK(hipEventRecord(e1, st));
                        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, (const f32x4 *)buf, n_vec, sink);
                        CHECK(hipEventRecord(e2, st));
                        CHECK(hipEventSynchronize(e2));
                        float b;
                        CHECK(hipEventElapsedTime(&b, e1, e2));
                        r_ms += b; moved += (double)bytes; ++reps;
                    }
                }
            }
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double watts = ps.path.empty() ? -1 : ps.finish();
            const double per = !strcmp(mode, "r") ? (double)bytes : (double)bytes;
            const double w_gbs = w_ms > 0 ? per * reps / (w_ms * 1e-3) / 1e9 : 0, r_gbs = r_ms > 0 ? per * reps / (r_ms * 1e-3) / 1e9 : 0;
            const double pj = watts > 0 && idle_w > 0 ? (watts - idle_w) * wall / moved * 1e12 : -1;
            printf("%-3s %5d  %8.0f %8.0f  %6.0f  %6.1f\n", mode, mib, w_gbs, r_gbs, watts, pj);
            fflush(stdout);
        }
    }
    return 0;
}
