// Synthetic-input noise for the batch workloads ("RNG = Philox4x32-10,
// key = seed, counter = (instance, step, lane); normals via Box-Muller").
//
// One standard normal per (instance, step, lane) triple, a pure function of (seed, triple):
// the stream of an instance does not depend on the batch it is generated in, on the rank that
// holds it or on the launch geometry -- a sharded Monte-Carlo run sees the numbers of the
// unsharded one.  Philox4x32-10 is the counter-based generator of Salmon et al., "Parallel
// random numbers: as easy as 1, 2, 3" (SC'11); oracle/philox.py restates it and pins it to
// the published known-answer vectors.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "eskf_device.h"

namespace eskf {
namespace {

struct Philox4 { uint32_t v[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// counter = (instance lo, instance hi, step, lane), key = (seed lo, seed hi)
__device__ __forceinline__ Philox4 draw(uint64_t seed, uint64_t instance, uint32_t step, uint32_t lane) {
    return philox4x32_10((uint32_t)instance, (uint32_t)(instance >> 32), step, lane, (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}

// Box-Muller on two 53-bit uniforms: u1 in (0,1], u2 in [0,1); the cosine branch only, so that one
// triple gives one number
__device__ __forceinline__ double normal_from(const Philox4& r) {
    const uint64_t k1 = ((uint64_t)r.v[0] << 21) | (r.v[1] >> 11);
    const uint64_t k2 = ((uint64_t)r.v[2] << 21) | (r.v[3] >> 11);
    const double u1 = (double)(k1 + 1) * 0x1.0p-53;
    const double u2 = (double)k2 * 0x1.0p-53;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

template <typename T>
__global__ void __launch_bounds__(256)
eskf_randn_kernel(uint64_t seed, int64_t first_instance, int64_t n_instances, int64_t first_step, int64_t n_steps,
                  int lanes, T* __restrict__ out) {
    const int64_t total = n_steps * n_instances * lanes;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t si = e / lanes;
        const uint32_t lane = (uint32_t)(e - si * lanes);
        const int64_t s = si / n_instances;
        const int64_t i = si - s * n_instances;
        out[e] = (T)normal_from(draw(seed, (uint64_t)(first_instance + i), (uint32_t)(first_step + s), lane));
    }
}

__global__ void __launch_bounds__(256)
eskf_philox_bits_kernel(uint64_t seed, int64_t first_instance, int64_t n_instances, int64_t first_step, int64_t n_steps,
                        int lanes, uint32_t* __restrict__ out) {
    const int64_t total = n_steps * n_instances * lanes;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t si = e / lanes;
        const uint32_t lane = (uint32_t)(e - si * lanes);
        const int64_t s = si / n_instances;
        const int64_t i = si - s * n_instances;
        const Philox4 r = draw(seed, (uint64_t)(first_instance + i), (uint32_t)(first_step + s), lane);
        *reinterpret_cast<uint4*>(out + 4 * e) = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    }
}

}  // namespace

hipError_t launch_randn(int kind, uint64_t seed, int64_t first_instance, int64_t n_instances, int64_t first_step,
                        int64_t n_steps, int lanes, void* out, hipStream_t stream) {
    const int64_t total = n_steps * n_instances * lanes;
    if (total <= 0) return hipSuccess;
    const int64_t want = (total + 255) / 256;
    const unsigned blocks = (unsigned)(want > 262144 ? 262144 : want);
    if (kind == 0)
        hipLaunchKernelGGL(eskf_randn_kernel<double>, dim3(blocks), dim3(256), 0, stream, seed, first_instance,
                           n_instances, first_step, n_steps, lanes, (double*)out);
    else if (kind == 1)
        hipLaunchKernelGGL(eskf_randn_kernel<float>, dim3(blocks), dim3(256), 0, stream, seed, first_instance,
                           n_instances, first_step, n_steps, lanes, (float*)out);
    else
        hipLaunchKernelGGL(eskf_philox_bits_kernel, dim3(blocks), dim3(256), 0, stream, seed, first_instance,
                           n_instances, first_step, n_steps, lanes, (uint32_t*)out);
    return hipGetLastError();
}

}  // namespace eskf
