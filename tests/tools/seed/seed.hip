#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *rsq, double *rcp, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { rsq[i] = __builtin_amdgcn_rsq(x[i]); rcp[i] = __builtin_amdgcn_rcp(x[i]); }
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-20.0, 20.0);
    for (auto &v : x) v = std::exp2(u(g));
    double *dx, *da, *db;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    long double er = 0, ec = 0;
    for (int i = 0; i < n; i++) {
        long double xr = 1.0L / sqrtl((long double)x[i]), xc = 1.0L / (long double)x[i];
        er = fmaxl(er, fabsl((a[i] - xr) / xr)); ec = fmaxl(ec, fabsl((b[i] - xc) / xc));
    }
    printf("rsq seed max rel err 2^%.2f   rcp seed max rel err 2^%.2f\n", (double)log2l(er), (double)log2l(ec));
    return 0;
}
