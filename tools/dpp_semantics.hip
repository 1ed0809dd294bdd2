// lane patterns of v_permlane16_swap / v_permlane32_swap (with the wait states inline assembly has to provide itself)
// build & run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/dpp_semantics.hip -o /tmp/dpp_sem && /tmp/dpp_sem
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void sem(int* outi, int nop) {
    const int l = threadIdx.x;
    int x = 1000 + l, y = 2000 + l;
    asm volatile("s_nop 7\n s_nop 7\n v_permlane16_swap_b32 %0, %1\n s_nop 7\n s_nop 7" : "+v"(x), "+v"(y));
    outi[l] = x; outi[64 + l] = y;
    int u = 1000 + l, w = 2000 + l;
    asm volatile("s_nop 7\n s_nop 7\n v_permlane32_swap_b32 %0, %1\n s_nop 7\n s_nop 7" : "+v"(u), "+v"(w));
    outi[128 + l] = u; outi[192 + l] = w;
}
int main() {
    int* di; (void)hipMalloc(&di, 4096);
    sem<<<1, 64>>>(di, 1);
    int h[256]; (void)hipMemcpy(h, di, 1024, hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) { printf(r == 0 ? "x16: " : r == 1 ? "y16: " : r == 2 ? "x32: " : "y32: "); for (int l = 0; l < 64; l += 4) printf("%d ", h[r * 64 + l]); printf("\n"); }
    return 0;
}
