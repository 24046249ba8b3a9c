// Micro-benchmark: the two-triangles-per-step interior-test loop of the closest-hit kernel in isolation
// (scalar-loaded rows, packed FP32, hit bits in a float), cycles per triangle pair per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float s) { return (f2)(s); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 step01(f2 m, f2 big) {
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(m), "s"(big));
    return r;
}

// One pair of triangles from the 16 SGPRs starting at s[B]: rows (a_s, aw, a_t, -), (b_s, bw, b_t, -) twice.
// v[112:113] = Ps, v[114:115] = Pt, v[116:117] = hit bits A, temporaries v[110:111], v[118:127], %7 = (2^126, 2^126).
#define PAIR(b0, b2, b4, b6, b8, b10, b12, b14)                                                              \
    "v_pk_fma_f32 v[118:119], s[" b0 "], v[112:113], s[" b0 "] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n"            \
    "v_pk_fma_f32 v[120:121], s[" b4 "], v[112:113], s[" b4 "] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n"            \
    "v_pk_fma_f32 v[122:123], s[" b8 "], v[112:113], s[" b8 "] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n"            \
    "v_pk_fma_f32 v[124:125], s[" b12 "], v[112:113], s[" b12 "] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n"          \
    "v_pk_fma_f32 v[118:119], s[" b2 "], v[114:115], v[118:119] op_sel_hi:[0,1,1]\n"                          \
    "v_pk_fma_f32 v[120:121], s[" b6 "], v[114:115], v[120:121] op_sel_hi:[0,1,1]\n"                          \
    "v_pk_fma_f32 v[122:123], s[" b10 "], v[114:115], v[122:123] op_sel_hi:[0,1,1]\n"                         \
    "v_pk_fma_f32 v[124:125], s[" b14 "], v[114:115], v[124:125] op_sel_hi:[0,1,1]\n"                         \
    "v_pk_add_f32 v[126:127], v[118:119], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"                    \
    "v_pk_add_f32 v[110:111], v[122:123], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"                    \
    "v_pk_add_f32 v[126:127], v[126:127], v[120:121] neg_lo:[0,1] neg_hi:[0,1]\n"                             \
    "v_pk_add_f32 v[110:111], v[110:111], v[124:125] neg_lo:[0,1] neg_hi:[0,1]\n"                             \
    "s_nop 0\n"                                                                                               \
    "v_min3_f32 v118, v118, v120, v126\n"                                                                     \
    "v_min3_f32 v119, v119, v121, v127\n"                                                                     \
    "v_min3_f32 v122, v122, v124, v110\n"                                                                     \
    "v_min3_f32 v123, v123, v125, v111\n"                                                                     \
    "v_pk_mul_f32 v[118:119], v[118:119], %7 clamp\n"                                                         \
    "v_pk_mul_f32 v[122:123], v[122:123], %7 clamp\n"                                                         \
    "s_nop 0\n"                                                                                               \
    "v_pk_fma_f32 v[118:119], v[118:119], 2.0, v[122:123] op_sel_hi:[1,0,1]\n"                                \
    "s_nop 0\n"                                                                                               \
    "v_pk_fma_f32 v[116:117], v[116:117], 4.0, v[118:119] op_sel_hi:[1,0,1]\n"
#define PAIR_A PAIR("36:37", "38:39", "40:41", "42:43", "44:45", "46:47", "48:49", "50:51")
#define PAIR_B PAIR("52:53", "54:55", "56:57", "58:59", "60:61", "62:63", "64:65", "66:67")
// %4 = row pointer, %1 = pairs left (consumed), %2 / %3 = byte offsets of the next rows for s[36:51] / s[52:67] (advanced)
#define T3D_TRI_PAIR_LOOP                                 \
    "s_load_dwordx16 s[36:51], %4, %2\n"                  \
    "1:\n"                                                \
    "s_waitcnt lgkmcnt(0)\n"                              \
    "s_cmp_le_i32 %1, 1\n"                                \
    "s_cbranch_scc1 2f\n"                                 \
    "s_load_dwordx16 s[52:67], %4, %3\n"                  \
    "s_add_u32 %2, %2, 0x80\n"                            \
    PAIR_A                                                \
    "s_waitcnt lgkmcnt(0)\n"                              \
    "s_cmp_le_i32 %1, 2\n"                                \
    "s_cbranch_scc1 3f\n"                                 \
    "s_load_dwordx16 s[36:51], %4, %2\n"                  \
    "s_add_u32 %3, %3, 0x80\n"                            \
    PAIR_B                                                \
    "s_sub_i32 %1, %1, 2\n"                               \
    "s_branch 1b\n"                                       \
    "2:\n"                                                \
    PAIR_A                                                \
    "s_branch 4f\n"                                       \
    "3:\n"                                                \
    PAIR_B                                                \
    "4:\n"
#define T3D_TRI_PAIR_CLOBBERS                                                                                            \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", \
        "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "v110",   \
        "v111", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "scc", "memory"
#define REPS 400
template <int VARIANT>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tri_uv, int n_tris, float* out) {
    const f2 big = splat(0x1p126f);
    f2 ps = {threadIdx.x * 0.001f, threadIdx.x * 0.002f}, pt = {0.3f + threadIdx.x * 0.0005f, 0.1f};
    f2 A = splat(0.0f);
    for (int rep = 0; rep < REPS; ++rep) {
        if (VARIANT == 2) {
            // software pipeline: the rows of the NEXT pair are requested before the current pair is evaluated
            float4 n0 = tri_uv[0], n1 = tri_uv[1], n2 = tri_uv[2], n3 = tri_uv[3];
            for (int k = 0; k + 1 < n_tris; k += 2) {
                const float4 ra0 = n0, rb0 = n1, ra1 = n2, rb1 = n3;
                n0 = tri_uv[2 * k + 4]; n1 = tri_uv[2 * k + 5]; n2 = tri_uv[2 * k + 6]; n3 = tri_uv[2 * k + 7];
                const f2 u0 = fma2(splat(ra0.z), pt, fma2(splat(ra0.x), ps, splat(ra0.y)));
                const f2 v0 = fma2(splat(rb0.z), pt, fma2(splat(rb0.x), ps, splat(rb0.y)));
                const f2 w0 = (splat(1.0f) - u0) - v0;
                const f2 m0 = {fminf(fminf(u0.x, v0.x), w0.x), fminf(fminf(u0.y, v0.y), w0.y)};
                const f2 u1 = fma2(splat(ra1.z), pt, fma2(splat(ra1.x), ps, splat(ra1.y)));
                const f2 v1 = fma2(splat(rb1.z), pt, fma2(splat(rb1.x), ps, splat(rb1.y)));
                const f2 w1 = (splat(1.0f) - u1) - v1;
                const f2 m1 = {fminf(fminf(u1.x, v1.x), w1.x), fminf(fminf(u1.y, v1.y), w1.y)};
                A = fma2(A, splat(4.0f), fma2(step01(m0, big), splat(2.0f), step01(m1, big)));
            }
            ps += A * 1e-30f;
            continue;
        }
        if (VARIANT == 3) {
            // hand-scheduled loop: rows of the next pair requested (s[52:67] / s[36:51] in turn) before the
            // current pair is evaluated, one s_waitcnt per pair that finds its data already there
            int npairs = n_tris / 2, off_a = 0, off_b = 0x40;
            asm volatile(T3D_TRI_PAIR_LOOP
                         : "+{v[116:117]}"(A), "+s"(npairs), "+s"(off_a), "+s"(off_b)
                         : "s"(tri_uv), "{v[112:113]}"(ps), "{v[114:115]}"(pt), "s"(big)
                         : T3D_TRI_PAIR_CLOBBERS);
            ps += A * 1e-30f;
            continue;
        }
        if (VARIANT == 4 || VARIANT == 5) {
            // rows staged in LDS (copied once per block), read back with wave-uniform addresses (every lane gets the same
            // 16 bytes: a broadcast, no bank conflict) into VGPRs; 4: as the compiler schedules it, 5: one pair ahead
            __shared__ float4 rows[2 * 72];
            if (rep == 0) {
                for (int i = threadIdx.x; i < 2 * n_tris + 8; i += blockDim.x) rows[i] = tri_uv[i];
                __syncthreads();
            }
            float4 n0 = rows[0], n1 = rows[1], n2 = rows[2], n3 = rows[3];
            for (int k = 0; k + 1 < n_tris; k += 2) {
                float4 ra0, rb0, ra1, rb1;
                if (VARIANT == 5) {
                    ra0 = n0; rb0 = n1; ra1 = n2; rb1 = n3;
                    n0 = rows[2 * k + 4]; n1 = rows[2 * k + 5]; n2 = rows[2 * k + 6]; n3 = rows[2 * k + 7];
                } else {
                    ra0 = rows[2 * k + 0]; rb0 = rows[2 * k + 1]; ra1 = rows[2 * k + 2]; rb1 = rows[2 * k + 3];
                }
                const f2 u0 = fma2(splat(ra0.z), pt, fma2(splat(ra0.x), ps, splat(ra0.y)));
                const f2 v0 = fma2(splat(rb0.z), pt, fma2(splat(rb0.x), ps, splat(rb0.y)));
                const f2 w0 = (splat(1.0f) - u0) - v0;
                const f2 m0 = {fminf(fminf(u0.x, v0.x), w0.x), fminf(fminf(u0.y, v0.y), w0.y)};
                const f2 u1 = fma2(splat(ra1.z), pt, fma2(splat(ra1.x), ps, splat(ra1.y)));
                const f2 v1 = fma2(splat(rb1.z), pt, fma2(splat(rb1.x), ps, splat(rb1.y)));
                const f2 w1 = (splat(1.0f) - u1) - v1;
                const f2 m1 = {fminf(fminf(u1.x, v1.x), w1.x), fminf(fminf(u1.y, v1.y), w1.y)};
                A = fma2(A, splat(4.0f), fma2(step01(m0, big), splat(2.0f), step01(m1, big)));
            }
            ps += A * 1e-30f;
            continue;
        }
        for (int k = 0; k + 1 < n_tris; k += 2) {
            float4 ra0 = tri_uv[2 * k + 0], rb0 = tri_uv[2 * k + 1], ra1 = tri_uv[2 * k + 2], rb1 = tri_uv[2 * k + 3];
            if (VARIANT == 1) {  // rows do not depend on the loop index: loads hoisted, pure VALU + loop overhead
                ra0 = tri_uv[0]; rb0 = tri_uv[1]; ra1 = tri_uv[2]; rb1 = tri_uv[3];
            }
            const f2 u0 = fma2(splat(ra0.z), pt, fma2(splat(ra0.x), ps, splat(ra0.y)));
            const f2 v0 = fma2(splat(rb0.z), pt, fma2(splat(rb0.x), ps, splat(rb0.y)));
            const f2 w0 = (splat(1.0f) - u0) - v0;
            const f2 m0 = {fminf(fminf(u0.x, v0.x), w0.x), fminf(fminf(u0.y, v0.y), w0.y)};
            const f2 u1 = fma2(splat(ra1.z), pt, fma2(splat(ra1.x), ps, splat(ra1.y)));
            const f2 v1 = fma2(splat(rb1.z), pt, fma2(splat(rb1.x), ps, splat(rb1.y)));
            const f2 w1 = (splat(1.0f) - u1) - v1;
            const f2 m1 = {fminf(fminf(u1.x, v1.x), w1.x), fminf(fminf(u1.y, v1.y), w1.y)};
            A = fma2(A, splat(4.0f), fma2(step01(m0, big), splat(2.0f), step01(m1, big)));
        }
        ps += A * 1e-30f;  // keep the repetitions dependent
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = A.x + A.y;
}
template <int VARIANT>
void run(const char* name, const float4* d_rows, int n_tris, float* d_out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;
        k<VARIANT><<<blocks, 256>>>(d_rows, n_tris, d_out);
        hipEventRecord(e0);
        k<VARIANT><<<blocks, 256>>>(d_rows, n_tris, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double pairs_per_simd = (double)REPS * (n_tris / 2) * wps;
        printf("%-34s waves/SIMD %d  %.3f ms  cycles per triangle pair per SIMD (2.4 GHz) %.1f  (20 VALU: %.2f each)\n", name, wps, ms,
               ms * 1e-3 * 2.4e9 / pairs_per_simd, ms * 1e-3 * 2.4e9 / pairs_per_simd / 20);
    }
}
int main() {
    const int n_tris = 64;
    float4 h[2 * 72];
    for (int i = 0; i < 2 * 72; ++i) h[i] = make_float4(0.3f * ((i * 7) % 5 - 2), 0.2f + 0.01f * (i % 9), 0.25f * ((i * 3) % 7 - 3), 0.f);
    float4* d_rows; hipMalloc(&d_rows, sizeof h); hipMemcpy(d_rows, h, sizeof h, hipMemcpyHostToDevice);
    float* d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    {   // the hand-scheduled loop must give the compiler-generated loop's result
        const int n = 256 * 256;
        float* a = new float[n]; float* b = new float[n];
        k<0><<<256, 256>>>(d_rows, n_tris, d_out); hipMemcpy(a, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
        k<3><<<256, 256>>>(d_rows, n_tris, d_out); hipMemcpy(b, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
        int bad = 0, nz = 0;
        for (int i = 0; i < n; ++i) { bad += a[i] != b[i]; nz += a[i] != 0.0f; }
        printf("asm loop vs compiled loop: %d of %d values differ (%d non-zero)\n", bad, n, nz);
    }
    run<0>("scalar-loaded rows", d_rows, n_tris, d_out);
    run<2>("scalar rows, one pair ahead", d_rows, n_tris, d_out);
    run<3>("hand-scheduled asm loop", d_rows, n_tris, d_out);
    run<1>("rows in registers (no loads)", d_rows, n_tris, d_out);
    run<4>("rows staged in LDS", d_rows, n_tris, d_out);
    run<5>("rows staged in LDS, one pair ahead", d_rows, n_tris, d_out);
    return 0;
}
