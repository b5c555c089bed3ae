// How many 256-thread workgroups with L bytes of dynamic LDS and a given register budget does the runtime say fit one CU?  (round 4: the generation-3
// launches were resident two per CU where the arithmetic said four.)   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_occupancy_query tools/ubench_occupancy_query.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ char smem[];
template <int W> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W))) void kern(float* out, int n)
{
    float acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = out[threadIdx.x + i * 256];
    for (int j = 0; j < n; ++j) for (int i = 0; i < 24; ++i) acc[i] = acc[i] * acc[(i + 1) % 24] + smem[(threadIdx.x * 4 + j) & 1023];
    float s = 0; for (int i = 0; i < 24; ++i) s += acc[i];
    out[threadIdx.x] = s;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("{\"sharedMemPerBlock\": %zu, \"maxSharedMemoryPerMultiProcessor\": %zu, \"regsPerBlock\": %d, \"multiProcessorCount\": %d, \"occupancy\": {", p.sharedMemPerBlock,
           p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.multiProcessorCount);
    const int lds[] = { 0, 16384, 32768, 33280, 36864, 40960, 49152, 65536 };
    for (int i = 0; i < 8; ++i)
    {
        int n4 = -1, n5 = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, kern<4>, 256, lds[i]);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n5, kern<5>, 256, lds[i]);
        printf("%s\"%d\": [%d, %d]", i ? ", " : "", lds[i], n4, n5);
    }
    printf("}}\n");
    return 0;
}
