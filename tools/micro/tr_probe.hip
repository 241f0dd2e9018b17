// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds ushort i at byte 2 i; lane l reads with address 8 l (contiguous) or a row-major
// [16 rows][stride] pattern; prints, per lane, the four 16-bit values it received.   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds;
    unsigned addr;
    if (mode == 0) addr = base + 8 * l;                                  // lane-linear 8-byte pieces
    else addr = base + ((l & 15) * 64 + (l >> 4) * 4) * 2;               // row (l & 15) of 64 ushorts, columns 4 (l >> 4) ..
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xFFFF; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xFFFF; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
