#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned* out) {
    extern __shared__ float lds[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < 400; ++i) __builtin_amdgcn_s_sleep(100);
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hwid; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    if (lds[threadIdx.x] == 123.f) out[0] = 0;
}
int main() {
    const int blocks = 512;
    unsigned* d; (void)hipMalloc(&d, blocks * 4 * 2 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 49152, 0, d);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 8);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < 20; ++b) {
        printf("block %3d:", b);
        for (int w = 0; w < 4; ++w) { unsigned x = h[(b * 4 + w) * 2]; printf("  [xcc %u se %u sh %u cu %2u simd %u wave %u]", h[(b*4+w)*2+1] & 15, (x >> 13) & 7, (x >> 12) & 1, (x >> 8) & 15, (x >> 4) & 3, x & 15); }
        printf("\n");
    }
    // histogram: for each (xcc, se, sh, cu, simd): which blocks have waves there
    std::map<unsigned, std::vector<int>> m;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) { unsigned x = h[(b*4+w)*2]; unsigned key = ((h[(b*4+w)*2+1] & 15) << 16) | (x & 0xfff0); m[key].push_back(b); }
    int same = 0, diff = 0, other = 0;
    for (auto& kv : m) { if (kv.second.size() == 2) { if (kv.second[0] == kv.second[1]) ++same; else ++diff; } else ++other; }
    printf("SIMDs holding 2 waves of the SAME block: %d, of two DIFFERENT blocks: %d, other counts: %d (of %zu SIMDs)\n", same, diff, other, m.size());
    int simds_per_block[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b) { std::map<unsigned, int> s; for (int w = 0; w < 4; ++w) s[h[(b*4+w)*2] & 0x30]++; simds_per_block[s.size()]++; }
    printf("blocks whose 4 waves sit on 1/2/3/4 distinct SIMDs: %d %d %d %d\n", simds_per_block[1], simds_per_block[2], simds_per_block[3], simds_per_block[4]);
    return 0;
}
