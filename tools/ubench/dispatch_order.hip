// micro-benchmark: where does the dispatcher put the LAST (partial) round of workgroups of a launch whose workgroups fit two per CU?
// Question behind it: a 200-row band of the 1600x1600 image is 1300 workgroups on 512 slots = 2.54 rounds.
// If the 276 workgroups of the last round land two-per-CU on 138 CUs the band costs 3 full rounds; if they land one-per-CU they run
// alone on their CU.  The kernel has the final stage's footprint (256 threads, 72 KiB dynamic LDS, <= 256 VGPRs: two per CU), spins on a
// dependent fma chain for a fixed number of iterations and records (XCC id, HW_ID, start, end) per workgroup.
// usage: dispatch_order_ub [n_workgroups ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { uint32_t xcc, hwid; unsigned long long t0, t1; };

__global__ __launch_bounds__(256, 2) void k_spin(int iters, float seed, Rec *rec, float *sink) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = seed;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = lds[threadIdx.x], b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fmaf(a, 1.0001f, 1e-3f); b = __builtin_fmaf(b, 0.9999f, 1e-3f);
        c = __builtin_fmaf(c, 1.0001f, -1e-3f); d = __builtin_fmaf(d, 0.9999f, -1e-3f);
    }
    if (a + b + c + d == 12345.678f) sink[threadIdx.x] = a;
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { rec[blockIdx.x] = Rec{xcc & 0xfu, hwid, t0, t1}; }
}

int main(int argc, char **argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {512, 625, 768, 1250, 1300, 2500};
    const size_t lds = 72 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spin, 256, lds);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("occupancy API: %d workgroups per CU; wall clock %d kHz\n", occ, wall_khz);
    Rec *drec; float *sink;
    hipMalloc(&drec, sizeof(Rec) * 8192); hipMalloc(&sink, 4096);
    const int iters = 200000;
    for (int n : sizes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_spin, dim3(n), dim3(256), lds, 0, iters, 1.0f, drec, sink);
            hipDeviceSynchronize();
        }
        std::vector<Rec> r(n);
        hipMemcpy(r.data(), drec, sizeof(Rec) * n, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (auto &x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
        // a workgroup belongs to the last round if it started after the first workgroup of the launch had finished its own slot's
        // predecessor, i.e. its start is later than (n / 512) whole rounds: classify by start order instead -- the last (n mod 512)
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return r[a].t0 < r[b].t0; });
        const int tail = n % 512 ? n % 512 : 0;
        std::map<uint32_t, int> per_cu_all, per_cu_tail;
        auto key = [&](const Rec &x) { return (x.xcc << 16) | ((x.hwid >> 8) & 0xffu); };     // xcc | se_id, sh_id, cu_id
        for (int i = 0; i < n; ++i) per_cu_all[key(r[i])]++;
        for (int i = n - tail; i < n; ++i) per_cu_tail[key(r[order[i]])]++;
        int hist[8] = {0};
        for (auto &kv : per_cu_tail) hist[std::min(kv.second, 7)]++;
        int mn = 1 << 30, mx = 0;
        for (auto &kv : per_cu_all) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
        double one = 0; for (auto &x : r) one += (double)(x.t1 - x.t0); one /= n;
        printf("n=%5d  CUs seen %3zu  workgroups per CU min %d max %d | last round: %3d workgroups on %3zu CUs (1 per CU: %d, 2 per CU: %d, 3+: %d) | "
               "launch %.3f ms, mean workgroup %.3f ms\n", n, per_cu_all.size(), mn, mx, tail, per_cu_tail.size(), hist[1], hist[2], hist[3] + hist[4] + hist[5] + hist[6] + hist[7],
               (double)(tmax - tmin) / wall_khz, one / wall_khz);
        // duration of tail workgroups by how many share their CU
        double d1 = 0, d2 = 0; int c1 = 0, c2 = 0;
        for (int i = n - tail; i < n; ++i) {
            const Rec &x = r[order[i]];
            if (per_cu_tail[key(x)] == 1) { d1 += (double)(x.t1 - x.t0); ++c1; } else { d2 += (double)(x.t1 - x.t0); ++c2; }
        }
        if (tail) printf("         tail workgroups alone on their CU: %d, mean %.3f ms; sharing: %d, mean %.3f ms\n", c1, c1 ? d1 / c1 / wall_khz : 0.0, c2, c2 ? d2 / c2 / wall_khz : 0.0);
    }
    return 0;
}
