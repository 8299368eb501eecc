// how long does rocprim::radix_sort_pairs take for the 63-bit octree path keys (uint64 keys, uint32 values)?
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
    for (size_t n : {500000ul, 1000000ul, 1100000ul, 2000000ul, 4000000ul}) {
        for (unsigned end_bit : {63u, 48u}) {
            std::vector<uint64_t> hk(n);
            uint64_t s = 88172645463325252ull;
            for (auto &k : hk) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; k = s >> 1; }
            uint64_t *k0, *k1; uint32_t *v0, *v1;
            hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
            hipMemcpy(k0, hk.data(), n * 8, hipMemcpyHostToDevice);
            size_t tb = 0;
            rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, n, 0u, end_bit, (hipStream_t)0);
            void *tmp; hipMalloc(&tmp, tb);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int w = 0; w < 2; w++) rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0u, end_bit, (hipStream_t)0);
            hipEventRecord(a);
            for (int w = 0; w < 10; w++) rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0u, end_bit, (hipStream_t)0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("n %zu bits 0..%u: %.1f us per sort (temp %zu bytes)\n", n, end_bit, ms * 100.0, tb);
            hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
        }
    }
    return 0;
}
