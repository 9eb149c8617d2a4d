// How many bits per onesweep pass suit the grid-backward sort (16.8 M / 21 M (u32 key, u32 value) pairs, 19..23 key bits)?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o sort_bits_ub sort_bits.hip ; ./sort_bits_ub
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class Config>
static float run(const char *name, uint32_t *k0, uint32_t *k1, uint32_t *v0, uint32_t *v1, size_t n, int bits) {
    size_t bytes = 0;
    OK(rocprim::radix_sort_pairs<Config>(nullptr, bytes, k0, k1, v0, v1, n, 0, bits, 0));
    void *tmp; OK(hipMalloc(&tmp, bytes));
    hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) OK(rocprim::radix_sort_pairs<Config>(tmp, bytes, k0, k1, v0, v1, n, 0, bits, 0));
    OK(hipEventRecord(a));
    const int it = 10;
    for (int i = 0; i < it; ++i) OK(rocprim::radix_sort_pairs<Config>(tmp, bytes, k0, k1, v0, v1, n, 0, bits, 0));
    OK(hipEventRecord(b)); OK(hipEventSynchronize(b));
    float ms; OK(hipEventElapsedTime(&ms, a, b));
    OK(hipFree(tmp));
    printf("  %-34s %2d key bits  n=%zu : %.1f us\n", name, bits, n, ms / it * 1e3f);
    return ms / it;
}

template <unsigned BITS, unsigned BS, unsigned IPT>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS,
                                                                           rocprim::block_radix_rank_algorithm::match>>;

int main() {
    for (size_t n : {(size_t)16777216, (size_t)20971520}) {
        for (int bits : {19, 23}) {
            std::vector<uint32_t> h(n);
            uint32_t s = 12345u;
            for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 7) & ((1u << bits) - 1u); }
            uint32_t *k0, *k1, *v0, *v1;
            OK(hipMalloc(&k0, n * 4)); OK(hipMalloc(&k1, n * 4)); OK(hipMalloc(&v0, n * 4)); OK(hipMalloc(&v1, n * 4));
            OK(hipMemcpy(k0, h.data(), n * 4, hipMemcpyHostToDevice)); OK(hipMemcpy(v0, h.data(), n * 4, hipMemcpyHostToDevice));
            run<rocprim::default_config>("default", k0, k1, v0, v1, n, bits);
            run<cfg<8, 512, 12>>("onesweep 8 bits 512x12", k0, k1, v0, v1, n, bits);
            run<cfg<7, 512, 12>>("onesweep 7 bits 512x12", k0, k1, v0, v1, n, bits);
            run<cfg<10, 512, 12>>("onesweep 10 bits 512x12", k0, k1, v0, v1, n, bits);
            run<cfg<10, 1024, 8>>("onesweep 10 bits 1024x8", k0, k1, v0, v1, n, bits);
            run<cfg<10, 256, 12>>("onesweep 10 bits 256x12", k0, k1, v0, v1, n, bits);
            run<cfg<9, 512, 12>>("onesweep 9 bits 512x12", k0, k1, v0, v1, n, bits);
            OK(hipFree(k0)); OK(hipFree(k1)); OK(hipFree(v0)); OK(hipFree(v1));
        }
    }
    return 0;
}
