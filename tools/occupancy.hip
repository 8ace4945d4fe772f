// Diagnostic: resident workgroups per CU the runtime grants each BAQ kernel (VGPR- and LDS-limited).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 tools/occupancy.hip -o tools/occupancy
#include "../secphase_amd/csrc/spx_kernels.hip"
#include <stdio.h>
template <class K>
static void show(const char *name, K k)
{
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, 0);
    hipFuncAttributes a;
    hipFuncGetAttributes(&a, (const void *)k);
    printf("%-34s blocks/CU %d  regs %d  lds %zu  scratch %zu  (%s)\n", name, n, a.numRegs, a.sharedSizeBytes, a.localSizeBytes, hipGetErrorString(e));
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s CUs %d lds/CU %zu lds/block %zu regs/block %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerBlock, p.clockRate);
    show("baq_fwd1_kernel<41>", baq_fwd1_kernel<41>);
    show("baq_fwd1_kernel<43>", baq_fwd1_kernel<43>);
    show("baq_fwd1_kernel<45>", baq_fwd1_kernel<45>);
    show("baq_bwd_kernel<2,21,41,false>", baq_bwd_kernel<2, 21, 41, false>);
    show("baq_bwd_kernel<2,22,43,false>", baq_bwd_kernel<2, 22, 43, false>);
    show("baq_fwd_kernel<4,16,0,false>", baq_fwd_kernel<4, 16, 0, false>);
    show("baq_fwd_kernel<8,16,0,false>", baq_fwd_kernel<8, 16, 0, false>);
    show("map_kernel<6,8>", map_kernel<6, 8>);
    show("map_kernel<16,8>", map_kernel<16, 8>);
    return 0;
}
