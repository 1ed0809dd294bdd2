// build & run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/dpp_micro.hip -o /tmp/dpp_micro && /tmp/dpp_micro
// micro-checks for the DPP-broadcast update: semantics of row_newbcast on f64, permlane16_swap, and issue rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

__global__ void sem(double* out, int* outi) {
    const int l = threadIdx.x;
    double col = 100.0 + l, acc = 0.0, one = 1.0, b = -1.0;
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(col), "v"(one));
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(col));
    out[l] = acc; out[64 + l] = b;
    int x = 1000 + l, y = 2000 + l;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    outi[l] = x; outi[64 + l] = y;
}

template <int MODE>
__global__ void rate(double* out, int iters) {
    const int l = threadIdx.x;
    double a0 = l, a1 = l + 1, a2 = l + 2, a3 = l + 3, a4 = l + 4, a5 = l + 5, a6 = l + 6, a7 = l + 7;
    double c = 1e-9 * l, k = 1.0000001;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile("v_fmac_f64_e32 %0, %8, %9\n v_fmac_f64_e32 %1, %8, %9\n v_fmac_f64_e32 %2, %8, %9\n v_fmac_f64_e32 %3, %8, %9\n"
                         "v_fmac_f64_e32 %4, %8, %9\n v_fmac_f64_e32 %5, %8, %9\n v_fmac_f64_e32 %6, %8, %9\n v_fmac_f64_e32 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(k));
        } else {
            asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(k));
        }
    }
    out[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    double* d; int* di;
    hipMalloc(&d, 1 << 24); hipMalloc(&di, 4096);
    sem<<<1, 64>>>(d, di);
    std::vector<double> h(128); std::vector<int> hi(128);
    hipMemcpy(h.data(), d, 128 * 8, hipMemcpyDeviceToHost); hipMemcpy(hi.data(), di, 128 * 4, hipMemcpyDeviceToHost);
    printf("fmac_dpp newbcast:3  lanes 0,5,17,40,63 -> %g %g %g %g %g (expect 103 103 119 135 151)\n", h[0], h[5], h[17], h[40], h[63]);
    printf("mov_dpp newbcast:5   lanes 0,5,17,40,63 -> %g %g %g %g %g (expect 105 105 121 137 153)\n", h[64], h[69], h[81], h[104], h[127]);
    printf("permlane16_swap x: lanes 0,16,32,48 -> %d %d %d %d ; y: %d %d %d %d\n", hi[0], hi[16], hi[32], hi[48], hi[64], hi[80], hi[96], hi[112]);
    const int iters = 20000, blocks = 256 * 8;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            if (mode == 0) rate<0><<<blocks, 64>>>(d, iters); else rate<1><<<blocks, 64>>>(d, iters);
            hipDeviceSynchronize();
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("mode %d (%s): %.3f ms, %.2f TFLOP/s\n", mode, mode ? "fmac_f64_dpp" : "fmac_f64", s * 1e3,
                   2.0 * 64 * 8 * iters * blocks / s / 1e12);
        }
    return 0;
}
