// accuracy of v_rcp_f64 and of one / two Newton steps on it, against the IEEE quotient: tools/probes/probe_rcp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const double *q, double *o0, double *o1, double *o2, double *oq, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x0 = __builtin_amdgcn_rcp(q[i]);
    const double x1 = fma(fma(-q[i], x0, 1.0), x0, x0);
    const double x2 = fma(fma(-q[i], x1, 1.0), x1, x1);
    o0[i] = x0; o1[i] = x1; o2[i] = x2; oq[i] = 1.0 / q[i];
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> q(n), a(n), b(n), c(n), d(n);
    srand(7);
    for (int i = 0; i < n; i++) q[i] = ldexp(1.0 + rand() / (double)RAND_MAX + rand() / ((double)RAND_MAX * RAND_MAX), rand() % 80 - 47);
    double *dq, *d0, *d1, *d2, *dd;
    hipMalloc(&dq, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&dd, n * 8);
    hipMemcpy(dq, q.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dq, d0, d1, d2, dd, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, ed = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / (long double)q[i];
        const double u = ldexp(1.0, ilogb((double)t) - 52);
        e0 = fmax(e0, fabs((double)((long double)a[i] - t)) / u);
        e1 = fmax(e1, fabs((double)((long double)b[i] - t)) / u);
        e2 = fmax(e2, fabs((double)((long double)c[i] - t)) / u);
        ed = fmax(ed, fabs((double)((long double)d[i] - t)) / u);
    }
    printf("max error in ulps of 1/q over %d samples: v_rcp_f64 %.3g, +1 Newton %.4f, +2 Newton %.4f, IEEE division %.4f\n", n, e0, e1, e2, ed);
    return 0;
}
