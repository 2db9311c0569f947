// Which CUs does a stream created with hipExtStreamCreateWithCUMask use?  Every workgroup records its XCC and its
// (SE, SH, CU) from the hardware id registers; usage: cumask_probe <first bit set> <bits set> [stride]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include <map>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void where_kernel(unsigned *out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // burn a little time so that workgroups spread over all permitted CUs
    float x = threadIdx.x;
    for (int i = 0; i < 20000; i++) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xcc & 0xf) | (x == 0.123f ? 16u : 0u); }
}
int main(int argc, char **argv)
{
    int first = argc > 1 ? atoi(argv[1]) : 0, count = argc > 2 ? atoi(argv[2]) : 256, stride = argc > 3 ? atoi(argv[3]) : 1;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> mask(words, 0u);
    int set = 0;
    for (int i = first; i < ncu && set < count; i += stride) { mask[i / 32] |= 1u << (i % 32); set++; }
    hipStream_t s; CHECK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
    int const blocks = 8192;
    unsigned *d; CHECK(hipMalloc(&d, blocks * 8));
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(64), 0, s, d);
    CHECK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * blocks); CHECK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> cus;     // xcc -> set of (se, sh, cu)
    for (int b = 0; b < blocks; b++) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;    // gfx9 HW_ID layout
        cus[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    printf("mask: bits %d.. x%d (stride %d) of %d CUs -> distinct CUs used per XCC:", first, set, stride, ncu);
    int total = 0;
    for (auto &kv : cus) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  total %d\n", total);
    return 0;
}
