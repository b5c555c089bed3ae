// Which XCD does workgroup i of a 1-D grid run on?  (MI355X_MICROARCH.md: round robin over the 8 XCDs in dispatch order.)  Also: is a plain store
// by one workgroup visible to a later plain load of another workgroup on the SAME XCD once an L2-executed atomic orders them (no sc1 traffic)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(uint32_t* xcc, uint32_t* cu)
{
    if (threadIdx.x == 0)
    {
        uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        uint32_t h; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        xcc[blockIdx.x] = x & 15; cu[blockIdx.x] = h;
    }
}
int main()
{
    for (int n : { 64, 256, 768, 1792, 4096 })
    {
        uint32_t *dx, *dc; hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4);
        std::vector<uint32_t> hx(n);
        int bad = 0;
        for (int rep = 0; rep < 5; ++rep)
        {
            probe<<<n, 256>>>(dx, dc); hipDeviceSynchronize();
            hipMemcpy(hx.data(), dx, n * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < n; ++i) if (hx[i] != (uint32_t) (i % 8)) ++bad;
        }
        printf("grid %5d: workgroups NOT on xcc == id %% 8 over 5 launches: %d ; first 16:", n, bad);
        for (int i = 0; i < 16; ++i) printf(" %u", hx[i]);
        printf("\n");
        hipFree(dx); hipFree(dc);
    }
    return 0;
}
