// Microbenchmark: a divergent gather of 32- / 96-byte records through the vector-memory path (what the pair kernels do)
// against the same gather out of an LDS tile that the workgroup staged with coalesced loads first.
//   hipcc -O3 --offload-arch=gfx950 gather_lds.hip -o gather_lds && ./gather_lds
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

constexpr int VIS = 64;
constexpr int TILE = 1152;        // records staged per workgroup: the block's own 256 +- 448

template <int PIECES>             // record = PIECES x 32 bytes
__global__ __launch_bounds__(256) void gather_global(const double2 *__restrict__ rec, const short *__restrict__ off, int n, double *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int base = max(0, min((int)blockIdx.x * 256 - 448, n - TILE));
    double s = 0.0;
    for (int k = 0; k < VIS; k++) {
        const int j = base + off[(size_t)k * n + i];
        const double2 *r = rec + (size_t)j * (2 * PIECES);
#pragma unroll
        for (int p = 0; p < 2 * PIECES; p++) { const double2 v = r[p]; s += v.x * v.y; }
    }
    out[i] = s;
}

template <int PIECES>
__global__ __launch_bounds__(256) void gather_tile(const double2 *__restrict__ rec, const short *__restrict__ off, int n, double *out) {
    extern __shared__ double2 tile[];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int base = max(0, min((int)blockIdx.x * 256 - 448, n - TILE));
    for (int t = threadIdx.x; t < TILE * 2 * PIECES; t += 256) tile[t] = rec[(size_t)base * (2 * PIECES) + t];
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < VIS; k++) {
        const int j = off[(size_t)k * n + i];
        const double2 *r = tile + j * (2 * PIECES);
#pragma unroll
        for (int p = 0; p < 2 * PIECES; p++) { const double2 v = r[p]; s += v.x * v.y; }
    }
    out[i] = s;
}

// the same with the tile stored piece by piece (16-byte piece p of record j at tile[p * TILE + j]): a 96-byte stride maps
// the 64 lanes of a read onto 8 groups of LDS banks, a 16-byte stride onto 16
template <int PIECES>
__global__ __launch_bounds__(256) void gather_tile_soa(const double2 *__restrict__ rec, const short *__restrict__ off, int n, double *out) {
    extern __shared__ double2 tile[];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int base = max(0, min((int)blockIdx.x * 256 - 448, n - TILE));
    for (int t = threadIdx.x; t < TILE * 2 * PIECES; t += 256) {
        const int j = t / (2 * PIECES), p = t - j * (2 * PIECES);
        tile[p * TILE + j] = rec[(size_t)base * (2 * PIECES) + t];
    }
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < VIS; k++) {
        const int j = off[(size_t)k * n + i];
#pragma unroll
        for (int p = 0; p < 2 * PIECES; p++) { const double2 v = tile[p * TILE + j]; s += v.x * v.y; }
    }
    out[i] = s;
}

int main() {
    const int n = 1 << 20;
    std::vector<short> off((size_t)VIS * n);
    srand(7);
    for (size_t t = 0; t < off.size(); t++) off[t] = (short)(rand() % TILE);
    short *d_off; double2 *d_rec; double *d_out;
    CK(hipMalloc(&d_off, off.size() * sizeof(short)));
    CK(hipMalloc(&d_rec, (size_t)n * 96));
    CK(hipMalloc(&d_out, (size_t)n * 8));
    CK(hipMemcpy(d_off, off.data(), off.size() * sizeof(short), hipMemcpyHostToDevice));
    std::vector<double> rec((size_t)n * 12);
    for (size_t t = 0; t < rec.size(); t++) rec[t] = (double)(t % 977) * 1e-3;
    CK(hipMemcpy(d_rec, rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> oa(n), ob(n);
    auto run = [&](const char *name, auto kern, size_t lds, std::vector<double> &o) -> int {
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int w = 0; w < 2; w++) kern<<<n / 256, 256, lds>>>(d_rec, d_off, n, d_out);
        CK(hipGetLastError());
        CK(hipEventRecord(e0));
        for (int w = 0; w < 10; w++) kern<<<n / 256, 256, lds>>>(d_rec, d_off, n, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(o.data(), d_out, (size_t)n * 8, hipMemcpyDeviceToHost));
        printf("%-40s %.3f ms / launch  (%d visits per lane)\n", name, ms / 10, VIS);
        return 0;
    };
    if (run("vector memory, 32-byte records", gather_global<1>, 0, oa)) return 1;
    if (run("LDS tile (36 KB), 32-byte records", gather_tile<1>, (size_t)TILE * 32, ob)) return 1;
    double d = 0; for (int i = 0; i < n; i++) d = fmax(d, fabs(oa[i] - ob[i]));
    printf("  max difference = %g\n", d);
    if (run("vector memory, 96-byte records", gather_global<3>, 0, oa)) return 1;
    if (run("LDS tile (108 KB), 96-byte records", gather_tile<3>, (size_t)TILE * 96, ob)) return 1;
    d = 0; for (int i = 0; i < n; i++) d = fmax(d, fabs(oa[i] - ob[i]));
    printf("  max difference = %g\n", d);
    if (run("LDS tile by pieces, 96-byte records", gather_tile_soa<3>, (size_t)TILE * 96, ob)) return 1;
    d = 0; for (int i = 0; i < n; i++) d = fmax(d, fabs(oa[i] - ob[i]));
    printf("  max difference = %g\n", d);
    if (run("LDS tile by pieces, 32-byte records", gather_tile_soa<1>, (size_t)TILE * 32, ob)) return 1;
    return 0;
}
