// Probe of the gfx950 semantics the fp16 two-plane gradient arithmetic (csrc/gemm_grad.hip) relies on:
//   (1) v_cvt_pkrtz_f16_f32 on values beyond the fp16 range: saturates to +-65504 (round toward zero) or inf?
//   (2) v_fma_mixlo_f16 / mixhi: m = rn_f16(x * s - h) with an fp16 source operand;
//   (3) v_mfma_f32_32x32x16_f16 with SUBNORMAL fp16 inputs: honoured or flushed to zero?
//   (4) the operand / accumulator layout of v_mfma_f32_32x32x16_f16 (must equal the bf16 form's).
//     hipcc --offload-arch=gfx950 -O2 -o /tmp/f16_probe tools/micro/f16_probe.hip && /tmp/f16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void probe_cvt(const float* x, float s, uint32_t* h, uint32_t* m) {
    const int i = threadIdx.x;
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t hh = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a * s, b * s));
    uint32_t mm;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(mm) : "v"(a), "v"(s), "v"(hh));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(mm) : "v"(b), "v"(s), "v"(hh));
    h[i] = hh;
    m[i] = mm;
}

// one wave: C (32 x 32) = A (32 x 16) * B^T (32 x 16), A / B given as fp16 row-major [row][k]
__global__ void probe_mfma(const _Float16* A, const _Float16* B, float* C) {
    const int lane = threadIdx.x, li = lane & 31, kh = lane >> 5;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[li * 16 + kh * 8 + j]; b[j] = B[li * 16 + kh * 8 + j]; }
    floatx16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + li] = c[r];
}

static float h2f(uint16_t h) {
    const int s = h >> 15, e = (h >> 10) & 31, f = h & 1023;
    float v = e == 0 ? std::ldexp((float)f, -24) : e == 31 ? (f ? NAN : INFINITY) : std::ldexp((float)(f | 1024), e - 25);
    return s ? -v : v;
}

int main() {
    // (1) + (2)
    std::vector<float> x = {1.0f, -1.0f, 3.14159274f, -2.71828175f, 70000.0f, -70000.0f, 1e9f, -1e9f, 65504.0f, 65519.9f,
                            1.00048828125f, 0.333333343f, 1e-3f, 6.1e-5f, 3.0e-6f, 5.9e-8f, 2.9e-8f, 0.0f, 123.456f, -0.1f,
                            4097.5f, 8191.999f, 2049.25f, 1e-7f};
    const int n = (int)x.size() / 2;
    float* dx; uint32_t *dh, *dm;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dh, n * 4); hipMalloc(&dm, n * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_cvt, dim3(1), dim3(n), 0, 0, dx, 1.0f, dh, dm);
    std::vector<uint32_t> h(n), m(n);
    hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
    printf("(1)/(2) x -> h = rtz_f16(x), m = rn_f16(x - h), x - (h + m) relative\n");
    for (int i = 0; i < (int)x.size(); ++i) {
        const uint16_t hh = (uint16_t)(h[i / 2] >> (16 * (i & 1))), mm = (uint16_t)(m[i / 2] >> (16 * (i & 1)));
        const double rec = (double)h2f(hh) + (double)h2f(mm);
        printf("  x = % .9g  h = % .9g (0x%04x)  m = % .9g (0x%04x)  rel err %.3g\n", x[i], h2f(hh), hh, h2f(mm), mm,
               x[i] != 0.f ? std::fabs(rec - x[i]) / std::fabs(x[i]) : std::fabs(rec));
    }
    // (3) + (4)
    std::vector<uint16_t> A(32 * 16), B(32 * 16);
    auto f2h_exact = [](int mant, int e) { return (uint16_t)((e << 10) | mant); };
    for (int r = 0; r < 32; ++r)
        for (int k = 0; k < 16; ++k) {
            A[r * 16 + k] = f2h_exact((r * 16 + k) & 1023, 15);                  // 1.xxx
            B[r * 16 + k] = f2h_exact((r * 7 + k * 3) & 1023, 14);               // 0.5 .. 1
        }
    // row 0 of A: subnormals (exponent field 0) against B = 2^14 so the products are visible in fp32
    for (int k = 0; k < 16; ++k) A[k] = (uint16_t)(k + 1);                       // (k + 1) * 2^-24
    _Float16 *dA, *dB; float* dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double worst = 0.0; int sub_ok = 1;
    for (int r = 0; r < 32; ++r)
        for (int c = 0; c < 32; ++c) {
            double ref = 0.0;
            for (int k = 0; k < 16; ++k) ref += (double)h2f(A[r * 16 + k]) * (double)h2f(B[c * 16 + k]);
            const double e = std::fabs(C[r * 32 + c] - ref) / std::fabs(ref);
            if (r == 0) { if (e > 1e-6) sub_ok = 0; } else if (e > worst) worst = e;
        }
    printf("(4) layout: worst relative error of rows 1..31 vs fp64: %.3g (expect ~1e-7)\n", worst);
    printf("(3) subnormal fp16 inputs (row 0): %s (C[0][0] = %.9g)\n", sub_ok ? "HONOURED" : "FLUSHED or wrong", C[0]);
    return 0;
}
