// Experiment: a thin, latency-bound kernel (few registers, no LDS, one long dependent fp64 chain per lane) to run
// beside the covariance kernel on another stream: does the GPU overlap the two?  (tools/exp_overlap.py)
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(128) void dummy_chain(double* x, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    for (int k = 0; k < iters; ++k) {
        double s, c;
        sincos(v, &s, &c);
        v = fma(s, 0.5, c * 0.25) + 1e-3 * v;
    }
    x[i] = v;
}
extern "C" int launch_dummy(double* x, int n, int iters, void* stream) {
    hipLaunchKernelGGL(dummy_chain, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, x, n, iters);
    return (int)hipGetLastError();
}
