// Vector-issue calibration for gfx950: how many shader cycles does one wave64 VALU
// instruction hold a SIMD for, as a function of the number of waves resident on that SIMD?
//
// Each wave runs ITERS x 32 independent instructions of one kind between two s_memtime reads.  A workgroup
// of 256*W threads puts W waves on each of a CU's four SIMDs; one workgroup per CU.  Output per (kind, W):
// median cycles per wave, and  cycles per wave-instruction per SIMD = cycles / (W * ITERS * 32).
// The same binary under `rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` gives the counter view.
//
// build + run (GPU box):  hipcc -O2 --offload-arch=gfx950 tools/valu_calib.hip -o /tmp/valu_calib && /tmp/valu_calib
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(OP)  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define PKMIN(i)  "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define MIN3(i)   "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define ADDU(i)   "v_add_u32 %" #i ", %" #i ", %8\n"
#define PKADD(i)  "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define DOT4(i)   "v_dot4_u32_u8 %" #i ", %" #i ", %8, %" #i "\n"
#define BCNT(i)   "v_bcnt_u32_b32 %" #i ", %8, %" #i "\n"
#define ADDE64(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define MINU16(i) "v_min_u16_e32 %" #i ", %" #i ", %8\n"
#define XORB(i)   "v_xor_b32_e32 %" #i ", %" #i ", %8\n"
#define SDWA(i)   "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
#define PERM(i)   "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define MIXED(i)  "v_min_u16_e32 %" #i ", %" #i ", %8\nv_pk_min_u16 %" #i ", %" #i ", %9\n"

template <int KIND>
__global__ void __launch_bounds__(1024) k_issue(uint64_t* out, int iters, uint32_t seed)
{
    uint32_t a0 = threadIdx.x ^ seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed * 2654435761u + threadIdx.x, c = b ^ 0x55aa55aa;
    __builtin_amdgcn_s_barrier();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define BODY(OP) asm volatile(REP8(OP) REP8(OP) REP8(OP) REP8(OP) \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))
        if (KIND == 0) BODY(PKMIN);
        if (KIND == 1) BODY(MIN3);
        if (KIND == 2) BODY(ADDU);
        if (KIND == 3) BODY(PKADD);
        if (KIND == 4) BODY(DOT4);
        if (KIND == 5) BODY(BCNT);
        if (KIND == 6) BODY(ADDE64);
        if (KIND == 7) BODY(MINU16);
        if (KIND == 8) BODY(XORB);
        if (KIND == 9) BODY(SDWA);
        if (KIND == 10) BODY(PERM);
        if (KIND == 11) BODY(MIXED);
    }
    asm volatile("s_nop 0" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if ((threadIdx.x & 63) == 0) {
        size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[w] = (t1 - t0) | ((uint64_t)(sink == 0xdeadbeef) << 63);
    }
}

template <int KIND>
static int run(const char* name, uint64_t* d, int cus)
{
    const int iters = 4096;
    CK(hipFuncSetAttribute((const void*)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    const double per_body = KIND == 11 ? 64 : 32;     // instructions per loop body
    for (int wps : {1, 2, 4, 8}) {                    // waves per SIMD
        // <= 4 waves per SIMD: one workgroup of 256*wps threads per CU; 8: two workgroups of 1024 per CU.  The dynamic LDS
        // request (96 KiB resp. 72 KiB of the CU's 160 KiB) is what makes the dispatcher place exactly that many per CU.
        int threads = 256 * std::min(wps, 4), per_cu = wps > 4 ? wps / 4 : 1, grid = cus * per_cu;
        size_t lds = per_cu == 1 ? 96 * 1024 : 72 * 1024;
        int nw = grid * threads / 64;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(threads), lds, 0, d, 16, 1u);          // warm
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(threads), lds, 0, d, iters, 1u);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h(nw);
        CK(hipMemcpy(h.data(), d, nw * sizeof(uint64_t), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        double med = (double)h[nw / 2], insts = (double)iters * per_body;
        printf("%-14s waves/SIMD=%d  cycles/wave median=%.0f (min %.0f max %.0f)  cycles per wave-instruction: one wave %.3f, per SIMD %.3f"
               "  | kernel %.3f ms => %.2f G wave-inst/s/SIMD-equivalent clock %.0f MHz\n",
               name, wps, med, (double)h[0], (double)h[nw - 1], med / insts, med / (insts * wps), ms,
               insts * nw / (ms * 1e-3) / 1e9 / (cus * 4), med / (ms * 1e-3) / 1e6);
    }
    return 0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    printf("# %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    uint64_t* d; CK(hipMalloc(&d, sizeof(uint64_t) * cus * 64));
    if (run<0>("v_pk_min_u16", d, cus)) return 1;
    if (run<1>("v_min3_u32", d, cus)) return 1;
    if (run<2>("v_add_u32", d, cus)) return 1;
    if (run<3>("v_pk_add_u16", d, cus)) return 1;
    if (run<4>("v_dot4_u32_u8", d, cus)) return 1;
    if (run<5>("v_bcnt_u32_b32", d, cus)) return 1;
    if (run<6>("v_add_u32_e64", d, cus)) return 1;
    if (run<7>("v_min_u16_e32", d, cus)) return 1;
    if (run<8>("v_xor_b32_e32", d, cus)) return 1;
    if (run<9>("v_add_u32_sdwa", d, cus)) return 1;
    if (run<10>("v_perm_b32", d, cus)) return 1;
    if (run<11>("e32+pk mixed", d, cus)) return 1;
    CK(hipFree(d));
    return 0;
}
