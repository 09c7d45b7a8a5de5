// w <- w + x + y on three arrays of 2^27 doubles (1 GiB each) in the single-vector kernels' shape: a workgroup owns
// 16 KiB tiles, a thread four 16-byte pieces 4 KiB apart; plain and non-temporal stores; 1, 2, 4 workgroups a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <bool NT, int R>
__global__ void __launch_bounds__(256) k3(double *w, const double *x, long stride, long tiles) {
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        d2v *p = reinterpret_cast<d2v *>(w + t * 2048) + threadIdx.x;
        d2v a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = p[u * 256];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const d2v *q = reinterpret_cast<const d2v *>(x + r * stride + t * 2048) + threadIdx.x;
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += q[u * 256];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NT) __builtin_nontemporal_store(a[u], p + u * 256); else p[u * 256] = a[u];
        }
    }
}
int main() {
    const long n = 1L << 27, tiles = n / 2048;
    double *buf;
    CK(hipMalloc(&buf, sizeof(double) * n * 4));
    CK(hipMemset(buf, 0, sizeof(double) * n * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](const char *name, int g, double passes, auto launch) {
        launch(256 * g);
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) launch(256 * g);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s groups/CU %d  %.3f ms  %.0f GB/s\n", name, g, ms / 5, passes * n * 8 / (ms / 5 * 1e-3) / 1e9);
    };
    for (int g : {1, 2, 4}) {
        run("w += x          plain", g, 3, [&](int G) { hipLaunchKernelGGL((k3<false, 1>), dim3(G), dim3(256), 0, 0, buf, buf + n, n, tiles); });
        run("w += x          nt", g, 3, [&](int G) { hipLaunchKernelGGL((k3<true, 1>), dim3(G), dim3(256), 0, 0, buf, buf + n, n, tiles); });
        run("w += x + y      plain", g, 4, [&](int G) { hipLaunchKernelGGL((k3<false, 2>), dim3(G), dim3(256), 0, 0, buf, buf + n, n, tiles); });
        run("w += x + y      nt", g, 4, [&](int G) { hipLaunchKernelGGL((k3<true, 2>), dim3(G), dim3(256), 0, 0, buf, buf + n, n, tiles); });
        run("w += x + y + z  nt", g, 5, [&](int G) { hipLaunchKernelGGL((k3<true, 3>), dim3(G), dim3(256), 0, 0, buf, buf + n, n, tiles); });
    }
    return 0;
}
