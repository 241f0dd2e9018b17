// Per-CU store throughput of the two epilogue shapes (one workgroup of 8 waves per CU, as the GEMM kernels):
//   dword:   a lane stores one column of a 32 x 32 fp32 tile, 16 rows (MFMA accumulator layout): 16 x buffer_store_dword
//   dwordx4: a lane stores 4 consecutive columns of 4 rows (LDS-transposed layout): 4 x buffer_store_dwordx4
// hipcc --offload-arch=gfx950 -O3 tools/micro/store_rate.hip -o tools/micro/store_rate && tools/micro/store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float fx4 __attribute__((ext_vector_type(4)));
template <int WIDE>
__global__ __launch_bounds__(512) void store_kernel(float* out, int ld, int tiles_per_wg, float v) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    for (int t = 0; t < tiles_per_wg; ++t) {
        float* base = out + ((size_t)(blockIdx.x * tiles_per_wg + t) * 256) * ld;       // a 256 x 256 output tile
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
        for (int tile = 0; tile < 8; ++tile) {
            const int mt = tile >> 1, nt = tile & 1;
            if (WIDE) {
                const int voff = ((wm * 128 + (lane >> 3)) * ld + wn * 64 + 4 * (lane & 7)) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fx4 o = {v + j, v, v, v};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), rc, voff,
                                                           ((mt * 32 + 8 * j) * ld + nt * 32) * 4, 0);
                }
            } else {
                const int voff = ((wm * 128 + 4 * kh) * ld + wn * 64 + li) * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + r), rc, voff,
                                                          ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ld + nt * 32) * 4, 0);
            }
        }
    }
}
int main(int argc, char** argv) {
    const int ld = 256, tiles = 64; const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    float* out;
    hipMalloc(&out, (size_t)wgs * tiles * 256 * ld * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wide = 0; wide < 2; ++wide)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (wide) hipLaunchKernelGGL(store_kernel<1>, dim3(wgs), dim3(512), 0, 0, out, ld, tiles, 1.0f);
            else hipLaunchKernelGGL(store_kernel<0>, dim3(wgs), dim3(512), 0, 0, out, ld, tiles, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)wgs * tiles * 256 * 256 * 4;
            printf("%s stores: %.3f ms, %.2f TB/s total, %.1f B/clk/CU at 2.0 GHz\n", wide ? "dwordx4" : "dword  ", ms, bytes / ms / 1e9,
                   bytes / wgs / (ms * 1e-3) / 2.0e9);
        }
    return 0;
}
