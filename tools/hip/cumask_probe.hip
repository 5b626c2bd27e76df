// Which CU-mask bit is which XCD / CU?  hipExtStreamCreateWithCUMask with one 32-bit word set at a time; the kernel reports the
// XCC_ID and HW_ID registers of every workgroup.   hipcc --offload-arch=gfx950 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ void k_where(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
    }
    // stay a little so that workgroups spread over the allowed CUs
    unsigned long long t0 = clock64();
    while (clock64() - t0 < 200000ull) {}
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    unsigned* d; hipMalloc(&d, 2 * 512 * 4);
    std::vector<unsigned> h(2 * 512);
    for (int w = 0; w < words; ++w) {
        for (int pat = 0; pat < 3; ++pat) {
            std::vector<uint32_t> mask(words, 0u);
            if (pat == 0) mask[w] = 0xFFFFFFFFu;            // one whole word
            else if (pat == 1) mask[w] = 0x000000FFu;        // its low 8 bits
            else { for (int k = 0; k < words; ++k) mask[k] = 0x01010101u << w; }  // every 8th bit, offset w
            hipStream_t s;
            hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
            if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 1; }
            hipMemsetAsync(d, 0xFF, 2 * 512 * 4, s);
            k_where<<<64, 64, 0, s>>>(d);
            hipMemcpyAsync(h.data(), d, 2 * 64 * 4, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            std::map<unsigned, int> xc; std::map<unsigned, int> cus;
            for (int b = 0; b < 64; ++b) { xc[h[2 * b] & 0xF]++; cus[(h[2 * b] & 0xF) << 16 | (h[2 * b + 1] & 0xFFFF)]++; }
            printf("word %d pattern %d:", w, pat);
            for (auto& kv : xc) printf(" xcd%u:%d", kv.first, kv.second);
            printf("  distinct (xcd,hw_id) %zu\n", cus.size());
            hipStreamDestroy(s);
        }
    }
    return 0;
}
