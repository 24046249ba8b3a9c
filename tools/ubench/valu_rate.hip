// Micro-benchmark: issue rate of float32 vector ALU instruction kinds on gfx950 (per SIMD, per wave count).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2000
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, const float* __restrict__ sc) {
    float x[16];
    f2 y[8];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) y[i] = (f2)(threadIdx.x * 0.001f + i);
    float s0 = sc[0], s1 = sc[1];
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0) {          // v_fma_f32, VGPR operands
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
            } else if (KIND == 1) {   // v_pk_fma_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = __builtin_elementwise_fma(y[i], (f2)(a), (f2)(b));
            } else if (KIND == 2) {   // v_fma with SGPR operand
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], s0, x[(i + 1) & 15]);
            } else if (KIND == 3) {   // v_rcp_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_rcpf(x[i]);
            } else if (KIND == 4) {   // v_min3_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fminf(fminf(x[i], x[(i + 1) & 15]), x[(i + 2) & 15]) + 0.0f * s1;
            } else if (KIND == 5) {   // v_cmp + v_cndmask
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = (x[i] > x[(i + 1) & 15]) ? a : x[i];
            } else if (KIND == 6) {   // v_mul_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = x[i] * a;
            } else if (KIND == 7) {   // v_pk_fma with SGPR broadcast operand
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = __builtin_elementwise_fma(y[i], (f2)(s0), y[(i + 1) & 7]);
            }
        }
    }
    float acc = 0;
    for (int i = 0; i < 16; ++i) acc += x[i];
    for (int i = 0; i < 8; ++i) acc += y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int KIND>
void run(const char* name, int instr_per_iter, float* d_out, const float* d_sc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {       // waves per SIMD
        int blocks = 256 * wps;                   // 256 CUs, 4 waves per block = 1 per SIMD
        k<KIND><<<blocks, 256>>>(d_out, 1.0001f, 0.5f, d_sc);
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(d_out, 1.0001f, 0.5f, d_sc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instr_per_simd = (double)ITERS * 4 * instr_per_iter * wps;
        double cycles = ms * 1e-3 * 2.4e9;
        printf("%-28s waves/SIMD %d  %.3f ms  cycles/wave-instr/SIMD (at 2.4 GHz) %.2f\n", name, wps, ms, cycles / wave_instr_per_simd);
    }
}
int main() {
    float* d_out; hipMalloc(&d_out, 256 * 8 * 256 * 4 * sizeof(float));
    float h[2] = {1.0001f, 0.25f}; float* d_sc; hipMalloc(&d_sc, 8); hipMemcpy(d_sc, h, 8, hipMemcpyHostToDevice);
    run<0>("v_fma_f32 (vgpr)", 16, d_out, d_sc);
    run<1>("v_pk_fma_f32", 8, d_out, d_sc);
    run<2>("v_fma_f32 (sgpr operand)", 16, d_out, d_sc);
    run<7>("v_pk_fma_f32 (sgpr bcast)", 8, d_out, d_sc);
    run<6>("v_mul_f32", 16, d_out, d_sc);
    run<3>("v_rcp_f32", 16, d_out, d_sc);
    run<4>("v_min3_f32 (+fma)", 32, d_out, d_sc);
    run<5>("v_cmp+v_cndmask", 32, d_out, d_sc);
    return 0;
}
