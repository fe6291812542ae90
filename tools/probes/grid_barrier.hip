// Probe: cost of a grid-wide barrier (256 workgroups x 512 threads, one per CU) with agent-scope release / acquire, the primitive a
// persistent decode-layer kernel would need between its phases.  Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512) void barrier_loop(unsigned* counter, float* data, int rounds, int touch) {
    const unsigned nwg = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        if (touch) {   // every workgroup writes a line and, after the barrier, reads its neighbour's (forces the release / acquire to matter)
            data[((size_t)blockIdx.x * 512 + threadIdx.x)] = (float)(r + blockIdx.x);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * nwg;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (touch) {
            const float v = data[((size_t)((blockIdx.x + 1) % nwg) * 512 + threadIdx.x)];
            if (v != (float)(r + (blockIdx.x + 1) % nwg)) data[(size_t)nwg * 512] = -1.f;     // stale read detector
        }
    }
}

int main() {
    unsigned* counter; float* data;
    hipMalloc(&counter, 4); hipMalloc(&data, (256 * 512 + 16) * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int touch = 0; touch < 2; ++touch)
        for (int rounds : {1, 101, 1001}) {
            hipMemset(counter, 0, 4); hipMemset(data, 0, (256 * 512 + 16) * sizeof(float));
            hipDeviceSynchronize();
            hipEventRecord(a);
            barrier_loop<<<256, 512>>>(counter, data, rounds, touch);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            float flag; hipMemcpy(&flag, data + 256 * 512, 4, hipMemcpyDeviceToHost);
            printf("touch=%d rounds=%4d: %8.1f us total, %6.2f us per barrier%s\n", touch, rounds, ms * 1e3, ms * 1e3 / rounds, flag < 0 ? "  STALE READ SEEN" : "");
        }
    return 0;
}
