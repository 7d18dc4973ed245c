// Developer tool: where does the dispatcher put the workgroups / waves of a persistent launch?
//   hipcc --offload-arch=gfx950 tools/placement_probe.hip -o /tmp/probe && /tmp/probe
// Prints, for a 1024 x 256-thread launch with 32 KB of LDS per workgroup (the K1p shape), the
// XCC / SE / CU / SIMD / wave-slot of every wave, summarised.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned int* out, unsigned long long* spin)
{
    extern __shared__ unsigned char lds[];
    lds[threadIdx.x] = 1;
    __syncthreads();
    // hold the wave for a while so that all 1024 workgroups are resident together
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000ull) { if (spin[0] == 12345ull) break; }
    if ((threadIdx.x & 63) == 0) {
        const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        const unsigned int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
}

int main()
{
    const int blocks = 1024, waves = blocks * 4;
    unsigned int* d; unsigned long long* s;
    hipMalloc(&d, waves * 8); hipMalloc(&s, 8); hipMemset(s, 0, 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d, 0xFF, waves * 8);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 32768, 0, d, s);
        hipDeviceSynchronize();
    }
    std::vector<unsigned int> h(waves * 2);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    // per block: (xcc, se, cu); per wave: simd, slot
    std::map<unsigned, std::vector<int>> cu_blocks;
    int slot_hist[16] = {0}, simd_of_wave[4][4] = {{0}};
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[2 * (b * 4 + w)], xcc = h[2 * (b * 4 + w) + 1] & 15;
            const unsigned slot = hw & 15, simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            slot_hist[slot]++;
            simd_of_wave[w][simd]++;
            if (w == 0) cu_blocks[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
            if (b < 24) printf("block %4d wave %d: xcc %u se %u sh %u cu %2u simd %u slot %u\n", b, w, xcc, se, sh, cu, simd, slot);
        }
    }
    printf("distinct CUs hosting wave 0 of a block: %zu\n", cu_blocks.size());
    int shown = 0;
    for (auto& kv : cu_blocks) {
        if (shown++ < 12) {
            printf("cu key %05x:", kv.first);
            for (int b : kv.second) printf(" %d", b);
            printf("\n");
        }
    }
    std::map<size_t, int> per_cu;
    for (auto& kv : cu_blocks) per_cu[kv.second.size()]++;
    for (auto& kv : per_cu) printf("%d CUs host %zu blocks\n", kv.second, kv.first);
    printf("wave-slot histogram:");
    for (int k = 0; k < 16; ++k) printf(" %d", slot_hist[k]);
    printf("\nsimd of wave-in-block w (rows w, cols simd):\n");
    for (int w = 0; w < 4; ++w) printf("  %d %d %d %d\n", simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
    return 0;
}
