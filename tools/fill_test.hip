// Does a 64-byte store issued by one quad (4 lanes x 16 B, one instruction) avoid the line fill that four
// 16-byte stores from ONE lane (four instructions) cause?  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fill_test.hip -o tools/fill_test
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void own_line(double2 *buf, size_t stride_lines, int iters)
{   // lane l writes the whole 64-byte line number (tid + it*total): four 16-byte stores
    size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, total = gridDim.x * (size_t)blockDim.x;
    for (int it = 0; it < iters; ++it) {
        double2 *p = buf + (tid + it * total) * 4;
        p[0] = make_double2(1, 2); p[1] = make_double2(3, 4); p[2] = make_double2(5, 6); p[3] = make_double2(7, 8);
    }
}
__global__ void quad_line(double2 *buf, size_t stride_lines, int iters)
{   // the four lanes of a quad write the four pieces of ONE line per instruction; four instructions cover the quad's four lines
    size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, total = gridDim.x * (size_t)blockDim.x;
    size_t q = tid & 3, base = tid & ~(size_t)3;
    for (int it = 0; it < iters; ++it)
        for (int k = 0; k < 4; ++k) buf[(base + k + it * total) * 4 + q] = make_double2(1 + k, 2 + q);
}
int main()
{
    const int blocks = 4096, threads = 64, iters = 32;
    size_t lines = (size_t)blocks * threads * iters;
    double2 *buf;
    hipMalloc(&buf, lines * 64);
    hipMemset(buf, 0, lines * 64);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(own_line, dim3(blocks), dim3(threads), 0, 0, buf, 0, iters);
        hipLaunchKernelGGL(quad_line, dim3(blocks), dim3(threads), 0, 0, buf, 0, iters);
    }
    hipDeviceSynchronize();
    printf("wrote %zu MB per kernel\n", lines * 64 >> 20);
    return 0;
}
