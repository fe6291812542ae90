// Probe (round 4): does the dispatcher place workgroups of a SECOND queue on a CU whose registers a resident GEMM workgroup leaves free?
// spin_kernel: 256-thread workgroups that do nothing but spin for `us` microseconds (s_memrealtime, 100 MHz) - no memory traffic, no LDS,
// no matrix work - with a chosen VGPR footprint (REGS = 24 or 104: the decode attention's is 102).  Launched on its own stream beside
// back-to-back gemm_tile256_kernel launches (tools/probes/spin_beside_gemm.py): if it co-resides, its launch takes ~`us` whatever the GEMM
// does; if it has to wait for whole CUs, it takes the GEMM tiles' time.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/probes/libspin_probe.so tools/probes/spin_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int REGS>
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long* out, int us) {
    if (REGS > 64) asm volatile("v_mov_b32 v100, 0" ::: "v100");          // raises the kernel's VGPR allocation to 104
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long until = t0 + (unsigned long long)us * 100ull;
    unsigned long long t = t0;
    while (t < until) { __builtin_amdgcn_s_sleep(8); t = __builtin_amdgcn_s_memrealtime(); }
    if (threadIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 4 + 0] = t0; out[blockIdx.x * 4 + 1] = t; out[blockIdx.x * 4 + 2] = hw; out[blockIdx.x * 4 + 3] = xcc;
    }
}

extern "C" int spin_launch(void* out, int n_wg, int us, int regs, void* stream) {
    if (regs > 64) spin_kernel<104><<<n_wg, 256, 0, (hipStream_t)stream>>>((unsigned long long*)out, us);
    else spin_kernel<24><<<n_wg, 256, 0, (hipStream_t)stream>>>((unsigned long long*)out, us);
    return (int)hipGetLastError();
}
