// Microbenchmark: does the texture-addresser path charge a divergent 16-byte gather per LANE or per distinct cache line /
// sector of neighbouring lanes?  Variant A = every lane reads the pieces of its own record (what the pair kernels do);
// variant B = the two lanes of a pair read the two halves of the SAME 32-byte chunk in one instruction and swap them with
// DPP afterwards.  Same bytes fetched per lane, half as many distinct chunks per instruction.
//   hipcc -O3 --offload-arch=gfx950 gather_pair.hip -o gather_pair && ./gather_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

constexpr int VIS = 64;

__device__ __forceinline__ double swap_pair(double v) {       // value of lane ^ 1
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int swap_pair_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); }

template <int PIECES>   // record = PIECES x 32 bytes (1: density-like, 3: forces-like)
__global__ __launch_bounds__(256) void gather_own(const double2 *__restrict__ rec, const int *__restrict__ idx, int n, double *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < VIS; k++) {
        const int j = idx[(size_t)k * n + i];
        const double2 *r = rec + (size_t)j * (2 * PIECES);
#pragma unroll
        for (int p = 0; p < 2 * PIECES; p++) { const double2 v = r[p]; s += v.x * v.y; }
    }
    out[i] = s;
}

template <int PIECES>
__global__ __launch_bounds__(256) void gather_pair(const double2 *__restrict__ rec, const int *__restrict__ idx, int n, double *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;           // n is a multiple of 256
    const int odd = threadIdx.x & 1;
    double s = 0.0;
    for (int k = 0; k < VIS; k++) {
        const int j = idx[(size_t)k * n + i];
        const int jp = swap_pair_i(j);
        const int je = odd ? jp : j, jo = odd ? j : jp;        // the even lane's and the odd lane's neighbour
        const double2 *re = rec + (size_t)je * (2 * PIECES) + odd;
        const double2 *ro = rec + (size_t)jo * (2 * PIECES) + odd;
#pragma unroll
        for (int p = 0; p < PIECES; p++) {
            const double2 a = re[2 * p], b = ro[2 * p];      // even lane: first halves, odd lane: second halves
            // what the partner needs from me: even lane gives b (first half of the odd lane's record), odd lane gives a
            const double gx = odd ? a.x : b.x, gy = odd ? a.y : b.y;
            const double tx = swap_pair(gx), ty = swap_pair(gy);
            const double2 mine0 = odd ? make_double2(tx, ty) : a;     // first half of my record
            const double2 mine1 = odd ? b : make_double2(tx, ty);     // second half of my record
            s += mine0.x * mine0.y; s += mine1.x * mine1.y;
        }
    }
    out[i] = s;
}

int main() {
    const int n = 1 << 20;
    std::vector<int> idx((size_t)VIS * n);
    srand(7);
    for (int k = 0; k < VIS; k++)
        for (int i = 0; i < n; i++) {
            int j = i + (rand() % 1201) - 600;       // neighbours in the sorted order: a few cells away
            idx[(size_t)k * n + i] = j < 0 ? 0 : (j >= n ? n - 1 : j);
        }
    int *d_idx; double2 *d_rec; double *d_out;
    CK(hipMalloc(&d_idx, idx.size() * sizeof(int)));
    CK(hipMalloc(&d_rec, (size_t)n * 96));
    CK(hipMalloc(&d_out, (size_t)n * 8));
    CK(hipMemcpy(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<double> rec((size_t)n * 12);
    for (size_t t = 0; t < rec.size(); t++) rec[t] = (double)(t % 977) * 1e-3;
    CK(hipMemcpy(d_rec, rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> oa(n), ob(n);
    auto run = [&](const char *name, auto kern, std::vector<double> &o) -> int {
        for (int w = 0; w < 2; w++) kern<<<n / 256, 256>>>(d_rec, d_idx, n, d_out);
        CK(hipEventRecord(e0));
        for (int w = 0; w < 10; w++) kern<<<n / 256, 256>>>(d_rec, d_idx, n, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(o.data(), d_out, (size_t)n * 8, hipMemcpyDeviceToHost));
        printf("%-28s %.3f ms / launch  (%d visits per lane)\n", name, ms / 10, VIS);
        return 0;
    };
    if (run("own, 32-byte records", gather_own<1>, oa)) return 1;
    if (run("paired, 32-byte records", gather_pair<1>, ob)) return 1;
    double d = 0; for (int i = 0; i < n; i++) d = fmax(d, fabs(oa[i] - ob[i]));
    printf("  max |own - paired| = %g\n", d);
    if (run("own, 96-byte records", gather_own<3>, oa)) return 1;
    if (run("paired, 96-byte records", gather_pair<3>, ob)) return 1;
    d = 0; for (int i = 0; i < n; i++) d = fmax(d, fabs(oa[i] - ob[i]));
    printf("  max |own - paired| = %g\n", d);
    return 0;
}
