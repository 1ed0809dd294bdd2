// build & run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/pk_micro.hip -o /tmp/pk_micro && /tmp/pk_micro
// What "fp32 designed as fp32" could buy the row update p[j] += c_j * k (c_j uniform over the instance's lanes, k per lane):
//   mode 0  v_fmac_f32            plain, operand already in the lane                 (the ceiling of one multiply-add per lane and instruction)
//   mode 1  v_fmac_f32_dpp        row_newbcast: the operand comes from another lane  (what the kernels issue today)
//   mode 2  v_pk_fma_f32          two columns per instruction, operand pair already in the lane (VOP3P takes no DPP modifier)
//   mode 3  ds_read2_b32 + v_pk_fma_f32   the operand pair delivered to every lane through LDS (all lanes of a half-wave read the same words)
//   mode 4  v_mov_b32_dpp x2 + v_pk_fma_f32   the operand pair delivered by two DPP moves
// Each block is 16 wavefronts (4 per SIMD), 8 independent accumulators (pairs) per lane; the figure is multiply-adds per
// clock and CU at the clock the run reached (from s_memtime), next to the wall-clock TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void rate(float* out, unsigned long long* ticks, int iters) {
    __shared__ float sh[2048];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 1.0f + 1e-7f * i;
    __syncthreads();
    f2 a0 = {1.f * l, 2.f}, a1 = {1.f, l + 1.f}, a2 = {2.f, 1.f * l}, a3 = {3.f, 1.f}, a4 = {4.f, 2.f}, a5 = {5.f, 3.f}, a6 = {6.f, 1.f}, a7 = {7.f, 2.f};
    f2 c = {1e-9f * l, 2e-9f * l}, k = {1.0000001f, 0.9999999f};
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)sh + 64u * (unsigned)(threadIdx.x >> 5);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {
#define OP(A) "v_fmac_f32_e32 %" #A ", %8, %9\n"
            asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(c.x), "v"(k.x));
#undef OP
        } else if constexpr (MODE == 1) {
#define OP(A) "v_fmac_f32_dpp %" #A ", %8, %9 row_newbcast:" #A " row_mask:0xf bank_mask:0xf\n"
            asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(c.x), "v"(k.x));
#undef OP
        } else if constexpr (MODE == 2) {
#define OP(A) "v_pk_fma_f32 %" #A ", %8, %9, %" #A "\n"
            asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(k));
#undef OP
        } else if constexpr (MODE == 3) {
            f2 c0, c1, c2, c3, c4, c5, c6, c7;
            asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:1\n ds_read2_b32 %1, %8 offset0:2 offset1:3\n ds_read2_b32 %2, %8 offset0:4 offset1:5\n"
                         "ds_read2_b32 %3, %8 offset0:6 offset1:7\n ds_read2_b32 %4, %8 offset0:8 offset1:9\n ds_read2_b32 %5, %8 offset0:10 offset1:11\n"
                         "ds_read2_b32 %6, %8 offset0:12 offset1:13\n ds_read2_b32 %7, %8 offset0:14 offset1:15\n s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4), "=&v"(c5), "=&v"(c6), "=&v"(c7) : "v"(lds_addr) : "memory");
            asm volatile("v_pk_fma_f32 %0, %8, %16, %0\n v_pk_fma_f32 %1, %9, %16, %1\n v_pk_fma_f32 %2, %10, %16, %2\n v_pk_fma_f32 %3, %11, %16, %3\n"
                         "v_pk_fma_f32 %4, %12, %16, %4\n v_pk_fma_f32 %5, %13, %16, %5\n v_pk_fma_f32 %6, %14, %16, %6\n v_pk_fma_f32 %7, %15, %16, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7), "v"(k));
        } else {
            f2 cc;
#define OP(A, L0, L1) "v_mov_b32_dpp %[lo], %[src] row_newbcast:" #L0 " row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %[hi], %[src] row_newbcast:" #L1 \
                      " row_mask:0xf bank_mask:0xf\n s_nop 0\n v_pk_fma_f32 %" #A ", %[cc], %[k], %" #A "\n"
            asm volatile(OP(0, 0, 1) OP(1, 2, 3) OP(2, 4, 5) OP(3, 6, 7) OP(4, 8, 9) OP(5, 10, 11) OP(6, 12, 13) OP(7, 14, 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [cc] "=&v"(cc), [lo] "=&v"(cc.x), [hi] "=&v"(cc.y)
                         : [src] "v"(c.x), [k] "v"(k));
#undef OP
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int mads_per_lane_per_iter, float* d, unsigned long long* dt, int iters, int blocks) {
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        rate<MODE><<<blocks, 1024>>>(d, dt, iters);
        hipDeviceSynchronize();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), dt, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= blocks;
        if (rep)
            printf("mode %d %-34s %8.3f ms  %6.2f TFLOP/s  %7.1f multiply-adds per clock and CU  (%.2f clocks per wavefront instruction group of 8)\n", MODE, name,
                   s * 1e3, 2.0 * 1024 * mads_per_lane_per_iter * (double)iters * blocks / s / 1e12, 1024.0 * mads_per_lane_per_iter * iters / mean,
                   mean / iters);
    }
}

int main() {
    float* d; unsigned long long* dt;
    const int iters = 20000, blocks = 256;     // one block of 16 wavefronts per CU
    hipMalloc(&d, (size_t)blocks * 1024 * sizeof(float)); hipMalloc(&dt, blocks * sizeof(unsigned long long));
    run<0>("v_fmac_f32", 8, d, dt, iters, blocks);
    run<1>("v_fmac_f32_dpp row_newbcast", 8, d, dt, iters, blocks);
    run<2>("v_pk_fma_f32 (operands in the lane)", 16, d, dt, iters, blocks);
    run<3>("ds_read2_b32 + v_pk_fma_f32", 16, d, dt, iters, blocks);
    run<4>("2 v_mov_b32_dpp + v_pk_fma_f32", 16, d, dt, iters, blocks);
    return 0;
}
