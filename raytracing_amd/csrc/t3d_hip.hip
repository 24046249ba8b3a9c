// t3d_hip.hip -- libt3d_hip.so: the T3DR "Photon" ray-tracing path on MI355X (gfx950, wave64).
//
// Structure: a wavefront (queue based) tracer.  One render call processes a band of image rows
// level by level of the trace tree (T3DR.py:959-1001):
//
//   for depth d = 0 .. bounces:
//       k_level   every wave is a small scheduler over two kinds of closest-hit rounds, two queries per
//                 lane per round:
//                 trace round: 128 rays of level d -- closest hit over all primitives, classify
//                 (miss / emitter leaf / shaded node), append the node to the level's node array and the
//                 reflected / transmitted child rays to the next level's queue (wave64 ballot + prefix
//                 popcount, one atomic per counter per round, counters sharded 64 ways);
//                 shadow round: 128 (node, light) pairs from an LDS ring of the wave's shaded nodes --
//                 shadow-ray closest hit, direct-light intensity (T3DR.py:820-900)
//                 Scenes traced through the LBVH (t3d_lbvh.h) run a level as three launches instead:
//                 k_bvh_trace (closest hits), k_bvh_process (the same classification), k_bvh_shadow;
//                 see the comment above k_bvh_trace.
//   for depth d = bounces .. 0:
//       k_resolve one thread per node: computeColor (T3DR.py:1004-1083) from the two child slots
//                 and the direct light, pushes the colour into the parent's slot
//   k_finish      per pixel: mean over samples, floor, clamp, BGR8 store (T3DR.py:1219-1221); with one sample
//                 per pixel the kernel that finishes the sample (k_level at a leaf, k_resolve at level 0)
//                 stores the pixel itself and k_finish is not launched
//
// The colour of a node is a clamped (non-linear) function of its children, so the tree is kept in
// HBM and resolved bottom-up; a throughput-accumulating path tracer would not give the same image.
//
// All arithmetic is float32.  What bounds the closest-hit loop on gfx950 is vector-ALU ISSUE: a wave64
// v_fma_f32 occupies its SIMD for ~4 cycles and so does a v_pk_fma_f32 doing two FMAs per lane (measured,
// tools/ubench/valu_rate.hip), i.e. the FP32 vector peak is a packed-math peak.  Hence:
//   * two rays per lane, held in the two halves of 64-bit register pairs, all dot products / hit points
//     / barycentrics in packed instructions;
//   * triangles grouped by supporting plane on the host: plane distance t and hit point P are computed
//     once per plane (like Ray.getTPlane, T3DE.py:133-142), each triangle of the plane only adds its
//     two barycentric rows u = a.P + aw, v = b.P + bw; the strict interior test of
//     Photon.pointWithinTriangle (T3DR.py:601-621) becomes min3(u, v, 1-u-v) > 0;
//   * plane and triangle rows are read with wave-uniform addresses, so they travel through the scalar
//     data path (SGPR operands) and cost no vector memory or LDS bandwidth.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/t3d.h"

#define T3D_BLOCK 256
#define T3D_WAVES (T3D_BLOCK / 64)
#define T3D_MAX_LIGHTS 4096 /* every emissive face is a light (T3DR.py:789-816); 4 bytes per (node, light) */
#define T3D_MAX_BOUNCES 14 /* heap index of the deepest node must fit 16 bits of the RNG key */
#define T3D_SHARDS 64      /* queue shards per level; every launch holds a multiple of 64 waves */
#define T3D_PAIRS 1024     /* (node, light) pairs per wave in the brute-force kernels' pair list: < 128 pending + <= 128 nodes x 7 lights */
#define T3D_LIVE_SHIFT 20  /* node records of such scenes: shader index in bits 0..19, the listed lights in bits 20..26, the side in bit 30 */
#define T3D_PAIR_LIGHTS 7  /* scenes with up to this many lights keep a list of the pairs whose light term can be non-zero */
#define T3D_RING 256       /* nodes per wave in the brute-force kernels: < 128 pairs pending before a trace round, which adds <= 128 nodes */

// ------------------------------------------------------------------------------------------------
// device-side scene
// ------------------------------------------------------------------------------------------------
struct DevShader {
    float r, g, b, rough;
    float refl, emis, pad0, pad1;
};

struct DevSphere {
    float cx, cy, cz, r2;  // centre, radius^2
    int obj;               // index into scene.objects
    int is_light;          // Light objects are visible only to rays traced with lights = true
    int shader;            // shaders[0] (for lights: colour holder)
    float strength;        // Light.strength
};

struct DevLight {
    float x, y, z, strength;
};

struct DevCamera {
    float pos[3];
    float m[9];  // Rz*Ry*Rx
    float zlen;
};

// Geometry streamed by every closest-hit query; passed as separate __restrict__ kernel arguments so
// that the compiler may use scalar loads for it.
//   planes      [6G]  per coplanar group a 96-byte record: (n.x, -n.v0, n.y, n.z) of its first triangle, un-normalised
//                     like MathematicalPlane.normal (T3DE.py:251-263); (first, end) of its triangle range as int
//                     bits, a parallelogram flag and the range once more as first | count << 26 (what a ray keeps of the
//                     group it hit); the tri_uv rows of its first two triangles (zeros where the
//                     group has only one; for a parallelogram only the first triangle's, see t3d_upload_scene)
//   plane_first [G+1] triangle range of each group (no kernel reads it any more: a group's range comes with its record)
//   tri_uv      [2T]  per triangle (a_s, aw, a_t, 0), (b_s, bw, b_t, 0): barycentric rows in the two in-plane
//                     coordinates (s, t) left after dropping the group's dominant axis, prepared in float64
#define GEO_PARAMS const float4* __restrict__ planes, const int* __restrict__ plane_first, \
                   const float4* __restrict__ tri_uv, const DevSphere* __restrict__ spheres
#define GEO_ARGS planes, plane_first, tri_uv, spheres

struct DevGeo {
    int axis_end[3];  // plane groups [0, axis_end[0]) have their largest normal component along x, then y, then z
    int n_tris, n_spheres;
    float box_lo[3], box_hi[3];  // bounds of every triangle and sphere, padded outward
};
struct DevBvh {
    int* spill;            // [traversal waves][T3D_BVH_DEEP_STACK][64]: stack entries beyond the LDS part, entry e of a lane at 64 e + lane
    const float4* nodes;   // [n_prims - 1] x 4 rows, see t3d_lbvh.h; null when the scene is traced by brute force
    const float4* leaves;  // [n_prims] x 4 rows in sorted order, see k_lbvh_leaves; triangles: (n.xyz, -n.v0), (a.xyz, aw),
                           // (b.xyz, bw) with u = a.P + aw, v = b.P + bw
    int n_prims;
};
struct DevScene {
    DevGeo geo;
    DevBvh bvh;
    int n_planes, n_tris, n_spheres, n_lights, n_shaders;
    const float4* tri_n;       // [T]: the triangle's own un-normalised normal, w = -(n . v0)
    const int4* tri_meta;      // [T]: (object, face-in-mesh, shader, unused); triangles in plane-group order
    const DevLight* lights;    // [L]
    const DevShader* shaders;  // [K]
    const float* obj_ior;      // [O]
    const float4* prim_info;   // [4 (T + S)]: what the classification of a hit needs about its primitive, gathered in one go
                               //   row 0: the triangle's unit normal (sphere: its centre), w = ior of the primitive's object
                               //   row 1: int bits: object, face in its mesh (sphere: -1), shader, kind (bit 0 emitter, bit 1 sphere)
                               //   row 2: roughness, reflectivity, emission strength (Light: its strength),
                               //          visibility a shadow ray is left with when this primitive is the first thing it hits
                               //   row 3: colour of its shader (what an emitter shows, T3DR.py:1014-1041)
    DevCamera cam;
    float light_color[3];  // sum over lights of colour * 0.6 with saturating adds (T3DR.py:893)
};

struct RenderParams {
    int width, height, row0, nrows, row_stride;
    int vrow0, frame_rows;   // several frames per call: local row k of the band is row (vrow0 + k) % frame_rows of frame (vrow0 + k) / frame_rows; frame_rows 0: one frame, row vrow0 + k
    uint32_t seed_step;      // seed of frame f: seed + f * seed_step
    int bounces, samples, lights;
    uint32_t seed;
    float ambient[3];
    int resident_waves;  // waves of k_level the device holds at once
    int n0;          // primary samples of this launch (rows * width * samples)
    int cap;         // capacity of one level's ray queue / node array (multiple of T3D_SHARDS * 64)
    int want_debug;  // record t3d_node entries (t3d_trace_pixel)
    int px0, py0, sample0;  // single-sample mode (t3d_trace_pixel / t3d_trace_ray): key of the random numbers
    int single;             // 0 image band, 1 one pixel sample (camera ray), 2 one explicit ray
    int full_rounds;        // always 128 rays per trace round (t3d_set_rounds)
    int tiled;              // level 0 walks the band in 16 x 8 pixel tiles (width % 16 == 0, rows % 8 == 0)
    float ray_o[3], ray_d[3], ray_ior;  // single == 2: the initial ray (unit direction) ...
    int ray_obj;                        // ... and the object it travels inside (-1 none)
    int live_masks;  // the node records list the lights whose shadow rays were traced (brute-force path, few lights)
    int shard_bits;  // log2 of the number of queue shards in use: 6 for image bands, 0 for the tree of a single ray
    int* sticky;     // two host-visible words (pinned memory): [0] set when a queue or node array overflows, [1] when a
                     // traversal stack does; see t3d_ctx
};

// One queued ray (64 B).
struct RayRec {
    float4 o;  // origin, w = ior of the medium the ray travels in
    float4 d;  // unit direction, w = int bits: object the ray is inside (-1 none)
    float4 p;  // parent node's ray.start (attenuation origin, T3DR.py:1062-1064), w = int bits: parent node
    uint4 k;   // x = pixel-sample index, y = sample << 16 | heap index, z = side (0 = reflected slot, 1 = transmitted), w = pixel hash of the random-number key
};

// Node arrays (36 B per node; hit point and normal of a node only live in the wave's LDS ring)
struct NodePool {
    float4* c;    // reflected slot rgb (child colour * attenuation), w = this node's attenuation towards its parent
    float4* d;    // transmitted slot rgb, w = int bits: shader | side << 30
    int* parent;  // index of the parent node (level 0: of the primary sample)
    int* dbg;   // debug record index of the node (only in single-pixel mode)
    float* li;  // [levels * cap * L] direct-light intensity per (node, light), written by k_level
};

#include "t3d_lbvh.h"
#include "t3d_post.h"

// Queue counters.  Same-address device-scope atomics retire at roughly one per 7 ns on this chip, so one
// counter per level would serialise the kernel: every level has T3D_SHARDS counters, each in its own
// 64-byte line, and wave w only ever touches shard w % T3D_SHARDS.  Workgroups go to the eight XCDs in turn and a
// grid holds a multiple of 16 workgroups (64 waves), so shard s is read and written on XCD (s / 4) % 8 in every launch:
// the rays and nodes one level appends are found in the same XCD's L2 by the next level and by k_resolve.
struct alignas(64) PaddedCount {
    int v;
    int pad[15];
};
struct Counters {
    PaddedCount n_rays[T3D_MAX_BOUNCES + 2][T3D_SHARDS];   // rays queued for each level, per shard
    PaddedCount n_nodes[T3D_MAX_BOUNCES + 2][T3D_SHARDS];  // shaded nodes of each level, per shard
    int overflow;
    int n_dbg;
    int bvh_stack_overflow;
    int pad_[13];
    unsigned long long bvh_stat[T3D_SHARDS][8];  // per shard: nodes visited, triangle tests, sphere tests (one 64-byte line each)
};

// Counters of a whole render call that was cut into several bands and is not waited for: the two counter sets are used
// in turn and a band's last kernel clears the set of the band before, so every band's sums are added up here by a
// small kernel queued behind it (k_add_counters), and t3d_last_call_stats reads these instead of the last band's set.
struct CallTotals {
    unsigned long long rays[T3D_MAX_BOUNCES + 2];   // tree rays per level (level 0: the primary samples)
    unsigned long long nodes[T3D_MAX_BOUNCES + 2];  // shaded nodes per level
    unsigned long long bvh[8];                      // the bvh_stat columns
};

// ------------------------------------------------------------------------------------------------
// small vector helpers
// ------------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float k) { return mk(a.x * k, a.y * k, a.z * k); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 madd(V3 a, V3 d, float t) {  // a + d * t
    return mk(fmaf(d.x, t, a.x), fmaf(d.y, t, a.y), fmaf(d.z, t, a.z));
}
__device__ __forceinline__ V3 normalize(V3 a) { return a * __builtin_amdgcn_rsqf(dot(a, a)); }

__device__ __forceinline__ float attenuation(float dist2) {  // getAttenuation(d), T3DR.py:782-785, from d^2
    float a = __builtin_amdgcn_rcpf(fmaf(0.01f, dist2, 1.0f));
    return fmaxf(0.0f, fminf(1.0f, a));
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float s) { return (f2)(s); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// ------------------------------------------------------------------------------------------------
// keyed random numbers (raytracing_amd/keyed_rng.py)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// The key is hashed in two steps: the part that belongs to the pixel once per primary ray (it then travels in
// the ray records), the part that belongs to the ray being generated per draw.
__device__ __forceinline__ uint32_t keyed_pixel_hash(uint32_t seed, uint32_t px, uint32_t py) {
    uint32_t h = mix32(seed + 0x9E3779B9u);
    h = mix32(h ^ px);
    return mix32(h ^ py);
}
__device__ __forceinline__ V3 keyed_cube(uint32_t pixel_hash, uint32_t sample_heap) {  // sample << 16 | heap
    uint32_t h = mix32(pixel_hash ^ sample_heap);
    uint32_t a = mix32(h ^ 0x9E3779B9u) >> 8, b = mix32(h ^ 0x3C6EF372u) >> 8, c = mix32(h ^ 0xDAA66D2Bu) >> 8;
    const float s = 1.0f / 8388608.0f;  // 2^-23: -1 + 2 * bits / 2^24, exact in float32
    return mk(fmaf((float)a, s, -1.0f), fmaf((float)b, s, -1.0f), fmaf((float)c, s, -1.0f));
}

// ------------------------------------------------------------------------------------------------
// closest hit: Photon.getFirstCollision (T3DR.py:654-730) over every primitive, two rays per lane
// ------------------------------------------------------------------------------------------------
#ifdef T3D_COUNT_GROUPS
// instrumented build only: how many plane groups the closest-hit loops visit, enter and how many triangle steps they run
struct GroupCount { uint32_t calls, visited, entered, steps; };
__device__ unsigned long long g_group_count[8];
#define GC_ADD(gc, field, n) do { if (gc) (gc)->field += (n); } while (0)
#else
struct GroupCount;
#define GC_ADD(gc, field, n) do { } while (0)
#endif

#ifdef T3D_WAVE_TRACE
// instrumented build only: when every wave of a k_level launch started and ended (s_memrealtime, 100 MHz), how many
// rounds it ran and where it ran (HW_ID register)
#define T3D_WAVE_TRACE_MAX (1 << 16)
__device__ unsigned long long g_wave_trace[T3D_MAX_BOUNCES + 1][T3D_WAVE_TRACE_MAX][4];
extern "C" int t3d_debug_wave_trace(int level, unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), sizeof(unsigned long long) * 4 * (size_t)n,
                               sizeof(unsigned long long) * 4 * T3D_WAVE_TRACE_MAX * (size_t)level) == hipSuccess ? 0 : -1;
}
#endif

struct Hit {
    float t;
    int prim;  // < 0 miss; [0, T) triangle (plane-group order); T + s sphere s
    int back;  // sphere: 1 = hit from inside
};

// 1 where m * 2^126 >= 1, 0 where m <= 0, per component: one packed multiply with the clamp modifier (the
// compiler does not fold a clamp into packed FP32 instructions; it does add the wait state a packed FP32
// result needs before its next use after the asm block).
__device__ __forceinline__ f2 step01(f2 m, f2 big) {
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(m), "s"(big));
    return r;
}

// Plane groups whose normal is largest along axis AX (getFaceCollision, T3DR.py:624-651).
// Per group:  d.n,  q = n.o - n.v0,  and the plane-level rejections folded into the divisor:
//     ok = step01(min((d.n)^2 - 1e-12, cull - d.n))      |d.n| > 1e-6 and (d.n < 0 or both sides count)
//     t  = -q / (d.n * ok)                               rejected planes divide by zero: t = +-inf or NaN
// (cull = 0 when back faces are rejected, +inf otherwise; packed instructions have no |x| modifier, hence the
// square).  Distances are compared as UNSIGNED integer bit patterns: bits(t) < bits(best) holds exactly for
// 0 <= t < best and fails for negative, infinite and NaN values, so neither "t >= 0" nor the rejections need a
// test of their own.  A group whose plane cannot beat the current best of any lane is skipped by the whole wave.
// Per triangle of the group, in the two coordinates (s, t) that remain after dropping axis AX:
//     u = a.(Ps, Pt) + aw,  v = b.(Ps, Pt) + bw,  w = 1 - u - v;   inside <=> min3(u, v, w) > 0 (strict: edges miss)
// The hit flags of a group (at most 24 triangles) are collected as the bits of an integer kept in a float:
//     A = 2 A + step01(min3(u, v, w))
// so its highest set bit is the FIRST triangle hit, which is the one the reference keeps (T3DR.py:715).
template <int AX, bool NO_CULL>
__device__ __forceinline__ void plane_groups(const float4* __restrict__ planes, const int* __restrict__ plane_first,
                                             const float4* __restrict__ tri_uv, int g_begin, int g_end, f2 ox, f2 oy, f2 oz,
                                             f2 dx, f2 dy, f2 dz, f2 cull, f2 big, uint32_t& best0, uint32_t& best1, float& A0,
                                             float& A1, int& bg0, int& bg1, GroupCount* gc = nullptr) {
    const f2 os = AX == 0 ? oy : (AX == 1 ? oz : ox), ot = AX == 0 ? oz : (AX == 1 ? ox : oy);
    const f2 ds = AX == 0 ? dy : (AX == 1 ? dz : dx), dt = AX == 0 ? dz : (AX == 1 ? dx : dy);
    for (int g = g_begin; g < g_end; ++g) {
        // One 96-byte record per group: plane row, triangle range, and the rows of its first two triangles --
        // requested together, so a group of up to two triangles costs a single wait on scalar memory.
        const float4 rn = planes[6 * g], rk = planes[6 * g + 1];
        const float4 fa0 = planes[6 * g + 2], fb0 = planes[6 * g + 3], fa1 = planes[6 * g + 4], fb1 = planes[6 * g + 5];
        const f2 dn = fma2(splat(rn.w), dz, fma2(splat(rn.z), dy, splat(rn.x) * dx));
        const f2 q = fma2(splat(rn.w), oz, fma2(splat(rn.z), oy, fma2(splat(rn.x), ox, splat(rn.y))));  // n.o - n.v0
        const f2 e = fma2(dn, dn, splat(-1e-12f));
        f2 mn = e;
        if constexpr (!NO_CULL) {   // shadow rays count both sides of every face: no cull term for them
            const f2 c = cull - dn;
            mn = f2{fminf(e.x, c.x), fminf(e.y, c.y)};
        }
        const f2 dnv = dn * step01(mn, big);
        const f2 inv = {__builtin_amdgcn_rcpf(dnv.x), __builtin_amdgcn_rcpf(dnv.y)};
        const f2 t = -q * inv;
        const uint32_t tp0 = __float_as_uint(t.x), tp1 = __float_as_uint(t.y);
        // four of five groups are entered: their range and first rows are asked for together with the plane row (this empty
        // statement needs them, so the requests stay in front of the branch)
        asm volatile("" :: "s"(rk.w), "s"(fa0.w), "s"(fb0.w));
        GC_ADD(gc, visited, 1);
        if (!__any((tp0 < best0) | (tp1 < best1))) continue;  // wave-uniform
        GC_ADD(gc, entered, 1);
        GC_ADD(gc, steps, (__float_as_int(rk.y) - __float_as_int(rk.x) + 1) / 2);
        const int k0 = __float_as_int(rk.x), k1 = __float_as_int(rk.y);
        const f2 ps = fma2(ds, t, os), pt = fma2(dt, t, ot);
        auto inside = [&](float4 ra, float4 rb) -> f2 {
            const f2 u = fma2(splat(ra.z), pt, fma2(splat(ra.x), ps, splat(ra.y)));
            const f2 v = fma2(splat(rb.z), pt, fma2(splat(rb.x), ps, splat(rb.y)));
            const f2 w = (splat(1.0f) - u) - v;
            const f2 m = {fminf(fminf(u.x, v.x), w.x), fminf(fminf(u.y, v.y), w.y)};
            return step01(m, big);
        };
        // two triangles per step; the first step always covers two bit positions (a group of one triangle has
        // an all-zero second row pair, which is never inside)
        f2 A;
        if (__float_as_int(rk.z) != 0) {   // wave-uniform: the first two triangles form a parallelogram
            const f2 u = fma2(splat(fa0.z), pt, fma2(splat(fa0.x), ps, splat(fa0.y)));
            const f2 v = fma2(splat(fb0.z), pt, fma2(splat(fb0.x), ps, splat(fb0.y)));
            const f2 w = (splat(1.0f) - u) - v;
            const f2 u1 = splat(1.0f) - u, v1 = splat(1.0f) - v;
            const f2 ma = {fminf(fminf(u.x, v.x), w.x), fminf(fminf(u.y, v.y), w.y)};
            const f2 mb = {fminf(fminf(u1.x, v1.x), -w.x), fminf(fminf(u1.y, v1.y), -w.y)};
            A = fma2(step01(ma, big), splat(2.0f), step01(mb, big));
        } else {
            A = fma2(inside(fa0, fb0), splat(2.0f), inside(fa1, fb1));
        }
        int k = k0 + 2;
        for (; k + 1 < k1; k += 2) {
            const float4 ra0 = tri_uv[2 * k + 0], rb0 = tri_uv[2 * k + 1], ra1 = tri_uv[2 * k + 2], rb1 = tri_uv[2 * k + 3];
            const f2 f0 = inside(ra0, rb0), f1 = inside(ra1, rb1);
            A = fma2(A, splat(4.0f), fma2(f0, splat(2.0f), f1));
        }
        if (k < k1) A = fma2(A, splat(2.0f), inside(tri_uv[2 * k + 0], tri_uv[2 * k + 1]));
        // closer = some triangle of the group was hit AND the plane is nearer (two compares and a scalar AND)
        const bool cl0 = (A.x >= 1.0f) & (tp0 < best0), cl1 = (A.y >= 1.0f) & (tp1 < best1);
        // (what is kept of the group is its triangle range as the record packs it: first triangle | count << 26)
        const int range = __float_as_int(rk.w);
        best0 = cl0 ? tp0 : best0; A0 = cl0 ? A.x : A0; bg0 = cl0 ? range : bg0;
        best1 = cl1 ? tp1 : best1; A1 = cl1 ? A.y : A1; bg1 = cl1 ? range : bg1;
    }
}

// Triangle index from the hit bits of its group: bit (n-1-j) stands for the j-th triangle of the group.
// (range: the group's first triangle | its triangle count << 26, from its record)
__device__ __forceinline__ int first_hit_of_group(int range, float A) {
    const int k0 = range & 0x3FFFFFF, n = (int)((uint32_t)range >> 26);
    const int width = n > 2 ? n : 2;  // bit positions used by the group: the first step always takes two
    int j = width - 1 - ((int)(__float_as_uint(A) >> 23) - 127);
    j = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
    return k0 + j;
}

// BOX_FIRST (camera rays): when no ray of the wave meets the scene's bounding box, nothing is tested at all.  The
// reference culls per object with padded boxes (T3DR.py:670-705); like there, the cull is conservative.
// Does the ray meet the (padded) bounding box of all primitives?  Slab test with 1/d clamped to a huge finite value
// (0 * inf would be NaN on a box face).  A ray that does not, hits nothing.
__device__ __forceinline__ bool ray_meets_scene_box(const DevGeo& geo, V3 o, V3 d) {
    const float BIGF = 1e30f;
    const float ix = fminf(fmaxf(__builtin_amdgcn_rcpf(d.x), -BIGF), BIGF), iy = fminf(fmaxf(__builtin_amdgcn_rcpf(d.y), -BIGF), BIGF),
                iz = fminf(fmaxf(__builtin_amdgcn_rcpf(d.z), -BIGF), BIGF);
    const float ax = (geo.box_lo[0] - o.x) * ix, bx = (geo.box_hi[0] - o.x) * ix;
    const float ay = (geo.box_lo[1] - o.y) * iy, by = (geo.box_hi[1] - o.y) * iy;
    const float az = (geo.box_lo[2] - o.z) * iz, bz = (geo.box_hi[2] - o.z) * iz;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000004f;
    return tn <= tf;
}

template <bool BOX_FIRST = false, bool NO_CULL = false>
__device__ __forceinline__ void closest_hit2(GEO_PARAMS, const DevGeo geo, V3 o0, V3 d0, V3 o1, V3 d1,
                                             bool lights, bool back_cull0, bool back_cull1, Hit& out0, Hit& out1, GroupCount* gc = nullptr,
                                             float bound0 = __builtin_inff(), float bound1 = __builtin_inff()) {
    // bound0 / bound1: only hits nearer than this count (non-negative; +inf: no bound).  The closest hit below the bound is
    // the closest hit, and a plane group that no ray of the wave meets below its bound is skipped like one that no ray
    // meets below its best distance.
    GC_ADD(gc, calls, 1);
    const uint32_t INF_BITS = 0x7f800000u;
    const f2 ox = {o0.x, o1.x}, oy = {o0.y, o1.y}, oz = {o0.z, o1.z};
    const f2 dx = {d0.x, d1.x}, dy = {d0.y, d1.y}, dz = {d0.z, d1.z};
    bool any_inside = true;   // wave-uniform
    if constexpr (BOX_FIRST) any_inside = __any((int)ray_meets_scene_box(geo, o0, d0) | (int)ray_meets_scene_box(geo, o1, d1));
    const f2 cull = {back_cull0 ? 0.0f : __builtin_inff(), back_cull1 ? 0.0f : __builtin_inff()};
    const f2 big = splat(0x1p126f);
    uint32_t best0 = __float_as_uint(bound0), best1 = __float_as_uint(bound1);
    float A0 = 0.0f, A1 = 0.0f;
    int bg0 = 0, bg1 = 0;
    int prim0 = -1, prim1 = -1, back0 = 0, back1 = 0;
    if (any_inside) {
    plane_groups<0, NO_CULL>(planes, plane_first, tri_uv, 0, geo.axis_end[0], ox, oy, oz, dx, dy, dz, cull, big, best0, best1, A0, A1, bg0, bg1, gc);
    plane_groups<1, NO_CULL>(planes, plane_first, tri_uv, geo.axis_end[0], geo.axis_end[1], ox, oy, oz, dx, dy, dz, cull, big, best0, best1, A0, A1, bg0, bg1, gc);
    plane_groups<2, NO_CULL>(planes, plane_first, tri_uv, geo.axis_end[1], geo.axis_end[2], ox, oy, oz, dx, dy, dz, cull, big, best0, best1, A0, A1, bg0, bg1, gc);
    if (A0 >= 1.0f) prim0 = first_hit_of_group(bg0, A0);   // some group was hit (its hit bits and its triangle range are kept with it)
    if (A1 >= 1.0f) prim1 = first_hit_of_group(bg1, A1);
    // Spheres and lights.  getSphereCollision (T3DR.py:572-598), Ray.getTSphere (T3DE.py:145-163).
    const int n_tris = geo.n_tris;
    for (int s = 0; s < geo.n_spheres; ++s) {
        const DevSphere sp = spheres[s];
        if (sp.is_light && !lights) continue;  // wave-uniform
        const f2 Lx = ox - splat(sp.cx), Ly = oy - splat(sp.cy), Lz = oz - splat(sp.cz);
        const f2 b = fma2(Lz, dz, fma2(Ly, dy, Lx * dx));
        // b^2 - (L.L - r^2) evaluated as r^2 - |L - b d|^2 (d is a unit vector): the textbook form cancels to
        // a few digits in float32 once the sphere is many radii away
        const f2 qx = fma2(-b, dx, Lx), qy = fma2(-b, dy, Ly), qz = fma2(-b, dz, Lz);
        const f2 disc = splat(sp.r2) - fma2(qz, qz, fma2(qy, qy, qx * qx));
        const f2 root = {__builtin_amdgcn_sqrtf(fmaxf(disc.x, 0.0f)), __builtin_amdgcn_sqrtf(fmaxf(disc.y, 0.0f))};
        const f2 t1 = -b - root, t2 = -b + root;
        {
            const bool front = t1.x > 0.0f;
            const bool inside = (!front) & (t2.x > 0.0f) & (!back_cull0) & (sp.is_light == 0);
            const bool ok = (disc.x >= 0.0f) & (front | inside);
            const uint32_t tt = ok ? __float_as_uint(front ? t1.x : t2.x) : INF_BITS;
            const bool closer = tt < best0;
            best0 = closer ? tt : best0; prim0 = closer ? n_tris + s : prim0; back0 = closer ? (front ? 0 : 1) : back0;
        }
        {
            const bool front = t1.y > 0.0f;
            const bool inside = (!front) & (t2.y > 0.0f) & (!back_cull1) & (sp.is_light == 0);
            const bool ok = (disc.y >= 0.0f) & (front | inside);
            const uint32_t tt = ok ? __float_as_uint(front ? t1.y : t2.y) : INF_BITS;
            const bool closer = tt < best1;
            best1 = closer ? tt : best1; prim1 = closer ? n_tris + s : prim1; back1 = closer ? (front ? 0 : 1) : back1;
        }
    }
    }
    out0.t = prim0 >= 0 ? __uint_as_float(best0) : __builtin_inff(); out0.prim = prim0; out0.back = back0;
    out1.t = prim1 >= 0 ? __uint_as_float(best1) : __builtin_inff(); out1.prim = prim1; out1.back = back1;
}

// ------------------------------------------------------------------------------------------------
// closest hit through the LBVH (t3d_lbvh.h): one ray per lane, near child first; the first T3D_BVH_LDS_STACK entries
// of a lane's stack live in LDS, deeper ones (rare) in a spill area in global memory that belongs to the wave.
// Same acceptance rules as the brute-force loop; equal distances go to the primitive that comes first in
// the scene's object / face order, whatever order the traversal meets them in.
// ------------------------------------------------------------------------------------------------
#ifndef T3D_BVH_STACK
#define T3D_BVH_STACK 96
#endif
#ifndef T3D_BVH_LDS_STACK
#define T3D_BVH_LDS_STACK 24   // entries per lane in LDS: 4 waves x 24 x 64 lanes x 4 B = 24 KB per block, 6 blocks per CU
#endif
#define T3D_BVH_DEEP_STACK (T3D_BVH_STACK - T3D_BVH_LDS_STACK)
#ifndef T3D_BVH_WAVES
#define T3D_BVH_WAVES 6   // waves per SIMD the traversal kernels are compiled for
#endif
#define BVH_DONE 0x7fffffff
struct BvhCount {
    uint32_t nodes, tris, spheres, deep;
    uint32_t culled = 0;   // shadow rays answered without a traversal (shadow_dead)
};

// Traversal state of one ray (one lane).  bvh_begin sets it up, bvh_step advances it by one node record or one
// leaf and says whether the ray is finished, so that a caller may hand the lane its next ray at once instead of
// waiting for the slowest ray of the wave.
struct BvhLane {
    V3 o, d, inv, oi;
    float best;
    int prim, rank, back;
    int sp, top;   // top mirrors stack[sp - 1]: a pop uses it at once and only then fetches the entry below,
    int node;      // so the stack read is off the critical path; BVH_DONE: nothing left to visit
    int pend;      // a leaf link (negative) met but not tested yet, 0: none
    bool lights, back_cull;
};   // the stack is a separate array: inside the structure it would drag every field into private memory

__device__ __forceinline__ void bvh_begin(BvhLane& s, V3 o, V3 d, bool lights, bool back_cull) {
    s.o = o; s.d = d;
    // 1/d with zero components replaced by a huge finite value: 0 * inf would be NaN on a box face
    const float BIGF = 1e30f;
    s.inv = mk(fminf(fmaxf(__builtin_amdgcn_rcpf(d.x), -BIGF), BIGF), fminf(fmaxf(__builtin_amdgcn_rcpf(d.y), -BIGF), BIGF),
               fminf(fmaxf(__builtin_amdgcn_rcpf(d.z), -BIGF), BIGF));
    s.oi = mk(-o.x * s.inv.x, -o.y * s.inv.y, -o.z * s.inv.z);
    s.best = __builtin_inff();
    s.prim = -1; s.back = 0; s.rank = 0x7fffffff;
    s.sp = 0; s.top = 0; s.node = 0; s.pend = 0;
    s.lights = lights; s.back_cull = back_cull;
}

// One step: a node record (four slab tests, entered slots pushed far to near, nearest descended into) or a leaf
// (one triangle or sphere).  Returns true when nothing is left on the stack.
// lds: this lane's column of the wave's LDS stack (entry e at lds[64 e]); deep: this lane's column of the wave's spill
// area for the entries beyond it (entry e at deep[64 (e - T3D_BVH_LDS_STACK)])
__device__ __forceinline__ bool bvh_step(const DevScene& sc, BvhLane& s, int* lds, int* __restrict__ deep, BvhCount& bc,
                                         int* stack_overflow, int* sticky) {
    const V3 o = s.o, d = s.d, inv = s.inv, oi = s.oi;
    auto pop = [&]() {
        if (s.sp == 0) { s.node = BVH_DONE; return; }
        s.node = s.top;
        --s.sp;
        const int e = s.sp > 0 ? s.sp - 1 : 0;
        s.top = e < T3D_BVH_LDS_STACK ? lds[64 * e] : deep[64 * (e - T3D_BVH_LDS_STACK)];
    };
    if (s.node >= 0 && s.node != BVH_DONE) {
        const float4* __restrict__ rec = sc.bvh.nodes + 4 * (size_t)s.node;
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        bc.nodes++;
        const uint32_t meta = __float_as_uint(r0.w);
        // plane = origin + q * step, so t = (plane - o) / d = q * (step * inv) + (origin * inv + oi)
        const float sx = __uint_as_float((meta & 255u) << 23) * inv.x, sy = __uint_as_float(((meta >> 8) & 255u) << 23) * inv.y,
                    sz = __uint_as_float(((meta >> 16) & 255u) << 23) * inv.z;
        const float cx = fmaf(r0.x, inv.x, oi.x), cy = fmaf(r0.y, inv.y, oi.y), cz = fmaf(r0.z, inv.z, oi.z);
        const uint32_t qlx = __float_as_uint(r1.x), qhx = __float_as_uint(r1.y), qly = __float_as_uint(r1.z), qhy = __float_as_uint(r1.w);
        const uint32_t qlz = __float_as_uint(r2.x), qhz = __float_as_uint(r2.y);
        const int base01 = __float_as_int(r2.z), base23 = __float_as_int(r2.w);
        // which plane of a slab the ray meets first depends on the sign of its direction only: the near planes of all
        // four slots are the lo bytes where the ray travels up the axis and the hi bytes where it travels down
        const uint32_t nx4 = inv.x < 0.0f ? qhx : qlx, fx4 = inv.x < 0.0f ? qlx : qhx;
        const uint32_t ny4 = inv.y < 0.0f ? qhy : qly, fy4 = inv.y < 0.0f ? qly : qhy;
        const uint32_t nz4 = inv.z < 0.0f ? qhz : qlz, fz4 = inv.z < 0.0f ? qlz : qhz;
        // a slot is entered while its slab interval reaches the best distance so far (<=: equal distances must all
        // be seen); empty slots never are
        float tn[4];
        int ln[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ax = fmaf((float)((nx4 >> (8 * k)) & 255u), sx, cx), bx = fmaf((float)((fx4 >> (8 * k)) & 255u), sx, cx);
            const float ay = fmaf((float)((ny4 >> (8 * k)) & 255u), sy, cy), by = fmaf((float)((fy4 >> (8 * k)) & 255u), sy, cy);
            const float az = fmaf((float)((nz4 >> (8 * k)) & 255u), sz, cz), bz = fmaf((float)((fz4 >> (8 * k)) & 255u), sz, cz);
            const float near = fmaxf(fmaxf(ax, ay), fmaxf(az, 0.0f));
            const float far = fminf(fminf(bx, by), fminf(bz, s.best)) * 1.0000004f;
            const bool present = (meta & (1u << (24 + k))) != 0u;
            tn[k] = (near <= far && present) ? near : __builtin_inff();   // +inf: not entered
            const int leaf = ((int)(meta << (3 - k))) >> 31;                 // bit 28 + k, spread: -1 for a leaf, else 0
            ln[k] = ((k < 2 ? base01 : base23) + (k & 1)) ^ leaf;            // leaves travel as negative links (~number)
        }
        // order the slots by entry distance (5 compare-exchanges), nearest first
        auto cswap = [&](int a, int b) {
            const bool sw = tn[b] < tn[a];
            const float ta = sw ? tn[b] : tn[a], tb = sw ? tn[a] : tn[b];
            const int la = sw ? ln[b] : ln[a], lb = sw ? ln[a] : ln[b];
            tn[a] = ta; tn[b] = tb; ln[a] = la; ln[b] = lb;
        };
        cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
        // push the entered slots 3, 2, 1 (farthest first), descend into slot 0.  Entered slots come first after the
        // sort, so a slot that is not entered may be written all the same: the next push or a later one overwrites it.
        const bool e1 = tn[1] < __builtin_inff(), e2 = tn[2] < __builtin_inff(), e3 = tn[3] < __builtin_inff();
        if (s.sp + 3 <= T3D_BVH_LDS_STACK) {
            lds[64 * s.sp] = ln[3]; s.sp += e3;
            lds[64 * s.sp] = ln[2]; s.sp += e2;
            lds[64 * s.sp] = ln[1]; s.sp += e1;
            s.top = e1 ? ln[1] : s.top;
        } else {
#pragma unroll
            for (int k = 3; k >= 1; --k) {
                if (tn[k] < __builtin_inff()) {
                    if (s.sp < T3D_BVH_LDS_STACK) { s.top = ln[k]; lds[64 * s.sp++] = s.top; }
                    else if (s.sp < T3D_BVH_STACK) { s.top = ln[k]; deep[64 * (s.sp++ - T3D_BVH_LDS_STACK)] = s.top; bc.deep++; }
                    else { *stack_overflow = 1; if (sticky) sticky[1] = 1; }
                }
            }
        }
        if (tn[0] < __builtin_inff()) s.node = ln[0];
        else pop();
    }
    // A leaf is not tested at once.  The leaf code is issued for the whole wave whenever one lane needs it, and at any
    // moment only a few lanes do: the link waits in the lane's pending slot while the lane goes on with its stack
    // (against a best distance that may be one leaf out of date, which can only cost visits), and the leaves of
    // all lanes are tested together as soon as one lane cannot go on without its own -- it met a second leaf, or
    // its stack is empty.
    if (s.node < 0 && s.pend == 0) { s.pend = s.node; pop(); }
    const bool blocked = s.pend != 0 && (s.node < 0 || s.node == BVH_DONE);
    if (__ballot(blocked) != 0ull && s.pend != 0) {
        // all rows are requested at once: one memory latency per leaf instead of two on the hit path
        const float4* __restrict__ rec = sc.bvh.leaves + 4 * (size_t)(~s.pend);
        s.pend = 0;
        const float4 rn = rec[0], ra = rec[1], rb = rec[2], rm = rec[3];
        const int p = __float_as_int(rm.x), rk = __float_as_int(rm.w);  // rk: place in the scene's object / face order, spheres last
        if (__float_as_int(rm.y) == 0) {
            // getFaceCollision (T3DR.py:624-651) for one triangle
            bc.tris++;
            const float dn = fmaf(rn.z, d.z, fmaf(rn.y, d.y, rn.x * d.x));
            if (fabsf(dn) >= 1e-6f && (!s.back_cull || dn < 0.0f)) {
                const float q = fmaf(rn.z, o.z, fmaf(rn.y, o.y, fmaf(rn.x, o.x, rn.w)));
                const float t = -q * __builtin_amdgcn_rcpf(dn);
                if (t >= 0.0f && t <= s.best) {
                    const V3 P = madd(o, d, t);
                    const float u = fmaf(ra.z, P.z, fmaf(ra.y, P.y, fmaf(ra.x, P.x, ra.w)));
                    const float v = fmaf(rb.z, P.z, fmaf(rb.y, P.y, fmaf(rb.x, P.x, rb.w)));
                    const float w = (1.0f - u) - v;
                    if (fminf(fminf(u, v), w) > 0.0f) {
                        if (t < s.best || rk < s.rank) { s.best = t; s.prim = p; s.rank = rk; s.back = 0; }
                    }
                }
            }
        } else {
            // getSphereCollision (T3DR.py:572-598)
            const int is_light = __float_as_int(ra.x);
            if (!is_light || s.lights) {
                bc.spheres++;
                const V3 Lv = mk(o.x - rn.x, o.y - rn.y, o.z - rn.z);
                const float b = dot(Lv, d);
                const V3 qv = madd(Lv, d, -b);  // L - b d: see closest_hit2
                const float disc = rn.w - dot(qv, qv);
                if (disc >= 0.0f) {
                    const float root = __builtin_amdgcn_sqrtf(disc);
                    const float t1 = -b - root, t2 = -b + root;
                    const bool front = t1 > 0.0f;
                    const bool inside = (!front) & (t2 > 0.0f) & (!s.back_cull) & (is_light == 0);
                    const float t = front ? t1 : t2;
                    if ((front | inside) && (t < s.best || (t == s.best && rk < s.rank))) { s.best = t; s.prim = p; s.rank = rk; s.back = front ? 0 : 1; }
                }
            }
        }
    }
    if (s.node < 0 && s.pend == 0) { s.pend = s.node; pop(); }   // the leaf that blocked the lane
    return s.node == BVH_DONE && s.pend == 0;
}

// Closest hit of one ray per lane; every lane runs until its own ray is finished.
__device__ __forceinline__ void bvh_closest(const DevScene& sc, const DevSphere* __restrict__ spheres, V3 o, V3 d, bool active,
                                            bool lights, bool back_cull, Hit& out, BvhCount& bc, int* stack_overflow, int* lds, int* deep) {
    out.t = __builtin_inff(); out.prim = -1; out.back = 0;
    if (!active) return;
    BvhLane s;
    bvh_begin(s, o, d, lights, back_cull);
    while (!bvh_step(sc, s, lds, deep, bc, stack_overflow, nullptr)) {}
    out.t = s.best; out.prim = s.prim; out.back = s.back;
}

// ------------------------------------------------------------------------------------------------
// k_level helpers
// ------------------------------------------------------------------------------------------------
struct TraceOut {
    float4* sample_color;  // [N0] rgb of a finished sample (w unused)
    int32_t* primary;      // [npix*2] or null
    float* primary_t;      // [npix] or null
    t3d_node* dbg;         // debug records or null
    int dbg_cap;
    uint8_t* bgr;          // one sample per pixel: a finished sample is a finished pixel and goes straight to the
    float* radiance;       // outputs (bgr null otherwise; radiance may be null)
};

// A pixel from the sum of its samples: floor-mean, clamp, BGR8 (T3DE.py:401-415, T3DR.py:1219-1221)
__device__ __forceinline__ void store_pixel(uint8_t* __restrict__ bgr, float* __restrict__ radiance, int p, float r, float g, float b, float n) {
    const float inv = 1.0f / n;
    if (radiance) { radiance[3 * p] = r * inv; radiance[3 * p + 1] = g * inv; radiance[3 * p + 2] = b * inv; }
    // fmaxf also turns NaN into 0: the conversion to uint8 is only defined for values in range (negative colours or
    // strengths can make a sum negative; the reference would wrap such a value when it stores it into its uint8 array)
    const float fr = fmaxf(fminf(floorf(r / n), 255.0f), 0.0f), fg = fmaxf(fminf(floorf(g / n), 255.0f), 0.0f), fb = fmaxf(fminf(floorf(b / n), 255.0f), 0.0f);
    bgr[3 * p] = (uint8_t)fb; bgr[3 * p + 1] = (uint8_t)fg; bgr[3 * p + 2] = (uint8_t)fr;
}

__device__ __forceinline__ void camera_ray(const DevCamera& cam, const RenderParams& rp, int px, int py, V3& o, V3& d) {
    // pixelToVector (T3DR.py:528-542): pixel corner on the image plane, rotated, translated
    float w = (float)rp.width, h = (float)rp.height;
    float ratio = w / h;
    float x = ((float)px / w) * ratio - ratio * 0.5f;
    float y = 0.5f - (float)py / h;
    float z = cam.zlen;
    // getRay (T3DR.py:545-551): direction = (pixelVec - camera.position), i.e. the rotated offset
    V3 v = mk(cam.m[0] * x + cam.m[1] * y + cam.m[2] * z, cam.m[3] * x + cam.m[4] * y + cam.m[5] * z,
              cam.m[6] * x + cam.m[7] * y + cam.m[8] * z);
    o = mk(cam.pos[0], cam.pos[1], cam.pos[2]);
    d = normalize(v);
}

// Direct-light term of a (node, light) pair given the closest hit of its shadow ray: getDirectLight for
// one light (T3DR.py:839-894) with getVisibility (T3DR.py:762-778).
struct ShadowRay {
    V3 P, N, to, Ld, start;   // to = light - P (not normalised), Ld = its direction
    float rough, dist2, strength;
    float reach;   // a little more than the distance from the ray's start to the light's position
};
__device__ __forceinline__ ShadowRay shadow_ray(const DevLight lt, float4 ra, float4 rb) {
    ShadowRay s;
    s.P = mk(ra.x, ra.y, ra.z); s.N = mk(rb.x, rb.y, rb.z); s.rough = ra.w;
    s.to = mk(lt.x - s.P.x, lt.y - s.P.y, lt.z - s.P.z);
    s.dist2 = dot(s.to, s.to);
    s.Ld = s.to * __builtin_amdgcn_rsqf(s.dist2);
    s.start = madd(s.P, s.Ld, 0.2f);  // T3DR.py:840
    s.strength = lt.strength;
    s.reach = fmaf(fmaxf(s.dist2 * __builtin_amdgcn_rsqf(s.dist2) - 0.2f, 0.0f), 1.0001f, 1e-3f);
    return s;
}
__device__ __forceinline__ ShadowRay shadow_ray(const DevScene& sc, float4 ra, float4 rb, int l) { return shadow_ray(sc.lights[l], ra, rb); }
// The part of the light term that does not depend on what the shadow ray hits: diffuse and specular response of the
// surface to the light's direction (T3DR.py:850-857, 880).  Where it is zero -- the surface faces away from the light
// and shows no highlight to the camera -- the term is zero whatever the visibility, and the ray need not be traced.
// Worked out from the un-normalised vector to the light, `to`, and the un-normalised vector to the camera, V, so that the
// test at node creation (which asks only whether the response is zero) needs no square root at all:
//   diffuse   max(0, L^.N) = to.N / |to| where to.N > 0, else exactly 0
//   specular  R^.V^ >= 0.96 with R = 2 (L^.N) N - L^ (unit, like L^ and N):  R' = 2 (to.N) N - to = |to| R, so the test is
//             R'.V >= 0.96 |to| |V|, i.e. R'.V > 0 and (R'.V)^2 >= 0.9216 |to|^2 |V|^2
// Both users call the same code, so a pair is culled exactly when its intensity would be multiplied by an exact zero.
struct LightResponse {
    float tn;     // to . N
    bool spec;    // the camera sees the light's mirror image at the point (T3DR.py:857)
};
__device__ __forceinline__ LightResponse light_response(V3 to, float dist2, V3 N, V3 V, float vv) {
    LightResponse r;
    r.tn = dot(to, N);
    const float two_tn = r.tn + r.tn;
    const float rv = fmaf(fmaf(N.z, two_tn, -to.z), V.z, fmaf(fmaf(N.y, two_tn, -to.y), V.y, fmaf(N.x, two_tn, -to.x) * V.x));
    r.spec = rv > 0.0f && rv * rv >= 0.9216f * dist2 * vv;
    return r;
}
// is the response zero?  (no division, no square root)
__device__ __forceinline__ bool response_zero(const LightResponse& r, float rough, float strength) {
    const bool diffuse = r.tn > 0.0f && rough != 0.0f, mirror = r.spec && rough != 1.0f;
    return !(diffuse || mirror) || strength == 0.0f;
}
__device__ __forceinline__ bool light_dead(const DevLight lt, V3 P, V3 N, V3 V, float vv, float rough) {
    const V3 to = mk(lt.x - P.x, lt.y - P.y, lt.z - P.z);   // as shadow_ray forms it
    return response_zero(light_response(to, dot(to, to), N, V, vv), rough, lt.strength);
}
__device__ __forceinline__ float shadow_facing(const DevScene& sc, const ShadowRay& s) {
    const V3 V = mk(sc.cam.pos[0] - s.P.x, sc.cam.pos[1] - s.P.y, sc.cam.pos[2] - s.P.z);
    const LightResponse r = light_response(s.to, s.dist2, s.N, V, dot(V, V));
    const float diffuse = r.tn > 0.0f && s.rough != 0.0f ? r.tn * __builtin_amdgcn_rsqf(s.dist2) * s.rough : 0.0f;
    const float mirror = r.spec && s.rough != 1.0f ? 1.0f - s.rough : 0.0f;
    return diffuse + mirror;
}
__device__ __forceinline__ bool shadow_dead(const DevScene& sc, const ShadowRay& s) {
    const V3 V = mk(sc.cam.pos[0] - s.P.x, sc.cam.pos[1] - s.P.y, sc.cam.pos[2] - s.P.z);
    return response_zero(light_response(s.to, s.dist2, s.N, V, dot(V, V)), s.rough, s.strength);
}
__device__ __forceinline__ float shadow_intensity(const DevScene& sc, const DevSphere* __restrict__ spheres, const ShadowRay& s,
                                                  const Hit& h) {
    // getVisibility (T3DR.py:762-778): 1 - reflectivity of the first thing hit unless that is a light or an emissive face
    const float vis = h.prim >= 0 ? sc.prim_info[4 * (size_t)h.prim + 2].w : 1.0f;
    return shadow_facing(sc, s) * s.strength * vis * attenuation(s.dist2);
}

// computeColor of a shaded node (T3DR.py:1054-1083) from its two child slots, its summed direct-light intensity and its
// shader, with the clamp of every RGB addition (T3DE.py:372-377)
__device__ __forceinline__ void node_color(const DevScene& sc, const RenderParams& rp, const DevShader& sh, float4 c, float4 d, float I, float (&res)[3]) {
    const float refl = sh.refl, tf = 1.0f - refl;
    const float col[3] = {sh.r, sh.g, sh.b};
    const float rs[3] = {c.x, c.y, c.z}, ts[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float indirect = fminf(rs[ch] * refl + ts[ch] * tf * 1.4f, 255.0f);
        float direct = fminf(col[ch] * I + sc.light_color[ch] * I * (2.0f / 255.0f), 255.0f) * refl;
        float v = fminf(indirect + direct, 255.0f);
        res[ch] = fminf(v + col[ch] * rp.ambient[ch] * (1.0f / 255.0f), 255.0f);
    }
}

// What a wave needs to know about itself
struct WaveCtx {
    int lane, wid, shard, shard_cap, level, L;
    size_t node_base;
    unsigned long long lt_mask;
};

// One ray of a trace round
struct RayIn {
    V3 o, d, pstart;
    float ior;
    int inside_obj, parent, side;
    uint32_t ps, heap;   // pixel-sample index of the root; sample << 16 | heap index of this ray in its tree
    uint32_t pixel_hash; // keyed_pixel_hash of the root's pixel
    int px, py, smp;     // level 0 only
    int i;  // index in the level's queue (level 0: sample index)
    bool active;
};

// What the closest hit of a ray leads to (per lane, no communication): classification (miss / emitter leaf / shaded
// node), the colour of an emitter leaf, and whether a shaded node gets a reflected and a transmitted child.  The child rays
// themselves are worked out later (child_rays), once the wave has reserved their slots, so that the classification of
// several rays of a lane can be kept in registers side by side.
struct Outcome {
    bool hit, emitter, shaded, want_t, want_b;
    V3 P, N;                      // hit point; unit normal as bounce / transmit / direct light use it
    V3 Nraw, emit;                // normal as the reference stores it (debug records); colour of an emitter leaf
    int obj, face, shader, dirflag;
    float att_parent, rough, obj_ior;
};
struct ChildRays {
    V3 tdir, bdir, tstart, bstart;
    float t_ior;
    int t_obj;
    uint32_t child_heap;          // heap key of the bounce child (2n); the transmit child is 2n + 1
};

// The medium on the far side of the surface a ray meets (T3DR.py:915-927): leaving the object the ray travels in, the
// normal is flipped and the ray enters air
__device__ __forceinline__ void transmit_side(const Outcome& r, const RayIn& in, V3& Nn, float& n2, int& nobj) {
    Nn = r.N; n2 = r.obj_ior; nobj = r.obj;
    if (r.obj == in.inside_obj) { Nn = r.N * -1.0f; n2 = 1.0f; nobj = -1; }
}

// level0: the ray is a primary ray (its leaf colours go to the pixel, its hit to the primary-id outputs);
// may_spawn: the ray's level is below options['bounces'] (T3DR.py:976-977)
// ONE: the launch may be the tree of a single ray (t3d_trace_pixel / t3d_trace_ray); kernels that only ever trace image
// bands say so, and the parameters of that mode are not kept in registers
// The record of the primitive a ray hit (DevScene::prim_info).  Fetched for every lane, hit or not (a miss reads record 0 and
// ignores it), so that the requests of a lane's rays are not tied to the rays' own outcomes and can all be in flight
// before the first of them is classified.
struct PrimRows { float4 i0, i1, i2, i3; };
__device__ __forceinline__ PrimRows fetch_prim_rows(const DevScene& sc, const Hit& h) {
    const float4* __restrict__ info = sc.prim_info + 4 * (size_t)(h.prim > 0 ? h.prim : 0);
    return PrimRows{info[0], info[1], info[2], info[3]};
}

template <bool ONE = true>
__device__ __forceinline__ Outcome hit_outcome(const DevScene& sc, const RenderParams& rp, const RayIn& in, const Hit& h, const PrimRows& pr,
                                               bool level0, bool may_spawn, const TraceOut& out) {
    const int single = ONE ? rp.single : 0;
    Outcome r;
    const V3 o = in.o, d = in.d;
    r.hit = in.active && h.prim >= 0;
    // ---- classify ---------------------------------------------------------------------
    r.P = madd(o, d, h.t);
    r.N = mk(0, 0, 1);
    r.Nraw = mk(0, 0, 1);
    r.obj = -1; r.face = -1; r.shader = 0; r.dirflag = 0;
    r.emitter = false;
    r.emit = mk(0, 0, 0);
    float strength = 1.0f, refl = 1.0f;
    r.rough = 1.0f; r.obj_ior = 1.0f;
    if (r.hit) {
        const float4 i0 = pr.i0, i1 = pr.i1, i2 = pr.i2, i3 = pr.i3;
        const int kind = __float_as_int(i1.w);
        r.obj = __float_as_int(i1.x); r.face = __float_as_int(i1.y); r.shader = __float_as_int(i1.z);
        if (!(kind & 2)) {
            // the record holds the unit normal (normalised in float64 at upload); the normal as the reference stores it,
            // un-normalised, only goes into debug records
            r.N = mk(i0.x, i0.y, i0.z);
            r.Nraw = r.N;
            if (ONE && rp.want_debug) { const float4 rn = sc.tri_n[h.prim]; r.Nraw = mk(rn.x, rn.y, rn.z); }
            r.dirflag = (dot(r.N, d) < 0.0f) ? 0 : 1;
        } else {
            r.dirflag = h.back;
            r.N = normalize(mk(r.P.x - i0.x, r.P.y - i0.y, r.P.z - i0.z));
            r.Nraw = r.N;
        }
        r.emitter = (kind & 1) != 0;   // an emissive face or a Light
        strength = i2.z;
        r.rough = i2.x; refl = i2.y; r.obj_ior = i0.w;
        if (r.emitter) {
            // computeColor for lights / emissive faces (T3DR.py:1014-1041): distance to the camera
            V3 pc = mk(r.P.x - sc.cam.pos[0], r.P.y - sc.cam.pos[1], r.P.z - sc.cam.pos[2]);
            float att = attenuation(dot(pc, pc));
            r.emit = mk(fminf(i3.x * strength * att, 255.0f), fminf(i3.y * strength * att, 255.0f),
                        fminf(i3.z * strength * att, 255.0f));
        }
    }
    if (level0 && in.active) {
        if (in.smp == 0 || single) {
            int pix = single ? 0 : in.i / rp.samples;
            if (out.primary) { out.primary[2 * pix] = r.obj; out.primary[2 * pix + 1] = r.face; }
            if (out.primary_t) out.primary_t[pix] = r.hit ? h.t : __builtin_nanf("");
        }
    }
    // attenuation of this node's colour on its way into the parent: |P - parent.ray.start|
    V3 pp = r.P - in.pstart;
    r.att_parent = level0 ? 1.0f : attenuation(dot(pp, pp));

    // ---- shaded node: which children it gets (T3DR.py:979-995) --------------------------------------
    r.shaded = r.hit && !r.emitter;
    r.want_t = false; r.want_b = false;
    if (r.shaded && may_spawn) {
        if (refl != 1.0f) {
            // transmit (T3DR.py:904-956): no child on total internal reflection
            V3 Nn; float n2; int nobj;
            transmit_side(r, in, Nn, n2, nobj);
            const float cosI = -dot(d, Nn);
            const float ratio = in.ior * __builtin_amdgcn_rcpf(n2);
            const float sin2T = ratio * ratio * (1.0f - cosI * cosI);
            r.want_t = !(sin2T > 1.0f);
        }
        r.want_b = refl != 0.0f;
    }
    return r;
}

// The bounce and transmit rays of a shaded node (those its classification asked for).
__device__ __forceinline__ ChildRays child_rays(const Outcome& r, const RayIn& in) {
    ChildRays c;
    const V3 d = in.d, N = r.N, P = r.P;
    c.tdir = mk(0, 0, 1); c.bdir = mk(0, 0, 1); c.tstart = P; c.bstart = P;
    c.t_ior = 1.0f; c.t_obj = -1;
    c.child_heap = (in.heap & 0xFFFF0000u) | ((2u * in.heap) & 0xFFFFu);  // bounce child 2n, transmit child 2n + 1
    if (r.want_t) {
        // transmit (T3DR.py:904-956)
        V3 Nn; float n2; int nobj;
        transmit_side(r, in, Nn, n2, nobj);
        const float cosI = -dot(d, Nn);
        const float ratio = in.ior * __builtin_amdgcn_rcpf(n2);
        const float sin2T = ratio * ratio * (1.0f - cosI * cosI);
        const float cosT = __builtin_amdgcn_sqrtf(1.0f - sin2T);
        V3 spec = madd(d * ratio, Nn, ratio * cosI - cosT);
        V3 diff = normalize(keyed_cube(in.pixel_hash, c.child_heap + 1u));
        if (dot(diff, Nn) > 0.0f) diff = diff * -1.0f;
        V3 dir = madd(spec * (1.0f - r.rough), diff, r.rough);
        c.tstart = madd(P, dir, 0.1f);
        c.tdir = normalize(dir);
        c.t_ior = n2; c.t_obj = nobj;
    }
    if (r.want_b) {
        // bounce (T3DR.py:733-759)
        V3 diff = normalize(keyed_cube(in.pixel_hash, c.child_heap));
        if (dot(diff, N) < 0.0f) diff = diff * -1.0f;
        V3 spec = madd(d, N, -2.0f * dot(d, N));
        V3 dir = madd(spec * (1.0f - r.rough), diff, r.rough);
        c.bstart = madd(P, dir, 0.1f);
        c.bdir = normalize(dir);
    }
    return c;
}

// The wave's list of shadow rays still to trace, when the scene has few lights (PAIRS): (ring slot of the node, light),
// node after node in ring order, only the pairs whose light term can be non-zero (shadow_dead).
struct PairList {
    unsigned short* ring;   // T3D_PAIRS entries in LDS: slot | light << 8
    int tail;               // pairs appended so far (wave-uniform)
    uint32_t culled;        // pairs answered without a ray, this lane
};

// Everything that follows the closest hits of the NR rays of a lane (wave-collective: ballots and atomics):
// classification, emitter / miss leaves, bounce and transmit rays, node records, queue appends, ring entries.
// The rays of a lane are classified first -- the gathers of all of them are in flight together -- and the wave then
// reserves its node and ray slots with ONE atomic per counter: nodes of ray 0 of every lane in lane order, then those of
// ray 1; child rays as (transmitted, reflected) of ray 0, then of ray 1.
template <bool LEVEL0, int RING, bool PAIRS = false, bool ONE = true, int NR = 1>
__device__ __forceinline__ void process_rays(const DevScene& sc, const DevLight* __restrict__ lights, const RenderParams& rp,
                                             const WaveCtx& wc, const RayIn (&in)[NR], const Hit (&h)[NR], RayRec* __restrict__ rays_out,
                                             const NodePool& pool, Counters* cnt, const TraceOut& out, float4* ring_a,
                                             float4* ring_b, int& ring_nodes, PairList* pl = nullptr) {
    const bool want_debug = ONE && rp.want_debug;
    Outcome oc[NR];
    PrimRows rows[NR];
    bool any_hit = false;   // wave-uniform: a round of camera rays that all leave the scene (two in five on the benchmark frame) fetches and tests nothing
#pragma unroll
    for (int r = 0; r < NR; ++r) any_hit |= __any((int)(in[r].active && h[r].prim >= 0)) != 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) rows[r] = any_hit ? fetch_prim_rows(sc, h[r]) : PrimRows{};
#pragma unroll
    for (int r = 0; r < NR; ++r) oc[r] = hit_outcome<ONE>(sc, rp, in[r], h[r], rows[r], LEVEL0, wc.level < rp.bounces, out);

    // ---- debug records (single-pixel mode) ----------------------------------------------
    int dbg_idx[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        dbg_idx[r] = -1;
        if (want_debug && oc[r].hit) {
            dbg_idx[r] = atomicAdd(&cnt->n_dbg, 1);
            if (dbg_idx[r] < out.dbg_cap) {
                t3d_node& nd = out.dbg[dbg_idx[r]];
                const V3 P = oc[r].P, o = in[r].o, d = in[r].d, emit = oc[r].emit;
                nd.heap = (int)(in[r].heap & 0xFFFFu); nd.obj = oc[r].obj; nd.face = oc[r].face; nd.direction = oc[r].dirflag; nd.t = h[r].t;
                nd.point[0] = P.x; nd.point[1] = P.y; nd.point[2] = P.z;
                nd.normal[0] = oc[r].Nraw.x; nd.normal[1] = oc[r].Nraw.y; nd.normal[2] = oc[r].Nraw.z;
                nd.ray_start[0] = o.x; nd.ray_start[1] = o.y; nd.ray_start[2] = o.z;
                nd.ray_dir[0] = d.x; nd.ray_dir[1] = d.y; nd.ray_dir[2] = d.z;
                nd.ray_ior = in[r].ior; nd.ray_obj = in[r].inside_obj;
                nd.color[0] = emit.x; nd.color[1] = emit.y; nd.color[2] = emit.z;
            }
        }
    }

    // ---- leaves -------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (in[r].active && (!oc[r].hit || oc[r].emitter)) {
            const V3 emit = oc[r].emit;
            if (LEVEL0) {
                V3 c = oc[r].hit ? emit : mk(rp.ambient[0], rp.ambient[1], rp.ambient[2]);  // T3DR.py:1006-1007
                if (out.bgr) store_pixel(out.bgr, out.radiance, in[r].i, c.x, c.y, c.z, 1.0f);
                else out.sample_color[in[r].i] = make_float4(c.x, c.y, c.z, 0.0f);
            } else if (oc[r].hit) {
                const float att_parent = oc[r].att_parent;
                float4* slot = in[r].side ? pool.d : pool.c;
                *(float3*)&slot[in[r].parent] = make_float3(emit.x * att_parent, emit.y * att_parent, emit.z * att_parent);
            }
        }
    }

    // ---- append: node slots and child-ray slots (wave64 ballot + prefix popcount) --------------------
    unsigned long long m_n[NR], m_t[NR], m_b[NR];
    int c_n = 0, c_r = 0;
    int off_n[NR], off_t[NR], off_b[NR];   // where this lane's entries go, relative to the wave's reservation
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        m_n[r] = __ballot(oc[r].shaded); m_t[r] = __ballot(oc[r].want_t); m_b[r] = __ballot(oc[r].want_b);
        off_n[r] = c_n + __popcll(m_n[r] & wc.lt_mask); c_n += __popcll(m_n[r]);
        off_t[r] = c_r + __popcll(m_t[r] & wc.lt_mask); c_r += __popcll(m_t[r]);
        off_b[r] = c_r + __popcll(m_b[r] & wc.lt_mask); c_r += __popcll(m_b[r]);
    }
    int base_n = 0, base_r = 0;
    if (wc.lane == 0) {   // both atomics are issued before either result is needed
        if (c_n) base_n = atomicAdd(&cnt->n_nodes[wc.level][wc.shard].v, c_n);
        if (c_r) base_r = atomicAdd(&cnt->n_rays[wc.level + 1][wc.shard].v, c_r);
    }
    // Few lights (PAIRS): which of them can add anything at a node?  Only those (node, light) pairs are listed for the
    // shadow rounds, node after node; the node record says which, so the others' intensities are never read.
    // (worked out while the atomics are on their way)
    uint32_t live[NR];
    int pair_pos[NR], pairs_total = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        live[r] = 0; pair_pos[r] = 0;
        if constexpr (RING > 0 && PAIRS) {
            const V3 V = mk(sc.cam.pos[0] - oc[r].P.x, sc.cam.pos[1] - oc[r].P.y, sc.cam.pos[2] - oc[r].P.z);
            const float vv = dot(V, V);
            pair_pos[r] = pairs_total;
            for (int l = 0; any_hit && l < wc.L; ++l) {
                // lights: the light array as a kernel argument of its own, so that this wave-uniform read goes through the
                // scalar cache (through the pointer in the scene structure it would be a vector load, waited for per light)
                const bool lv = oc[r].shaded && !light_dead(lights[l], oc[r].P, oc[r].N, V, vv, oc[r].rough);
                const unsigned long long m = __ballot(lv);
                pair_pos[r] += __popcll(m & wc.lt_mask);
                pairs_total += __popcll(m);
                live[r] |= (uint32_t)lv << l;
            }
        }
    }
    base_n = __shfl(base_n, 0);
    base_r = __shfl(base_r, 0);
    int node[NR];
    bool stored[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int nloc = base_n + off_n[r];                            // index within the shard
        node[r] = wc.shard * wc.shard_cap + nloc;                      // index within the level
        stored[r] = oc[r].shaded && nloc < wc.shard_cap;
        if (oc[r].shaded && !stored[r]) { cnt->overflow = 1; rp.sticky[0] = 1; }
        if (stored[r]) {
            const float k = 2.0f / 255.0f;  // absent children count as ambient * 2/255 (T3DR.py:1054-1055)
            const size_t g = wc.node_base + (size_t)node[r];
            pool.c[g] = make_float4(rp.ambient[0] * k, rp.ambient[1] * k, rp.ambient[2] * k, oc[r].att_parent);
            pool.d[g] = make_float4(rp.ambient[0] * k, rp.ambient[1] * k, rp.ambient[2] * k,
                                    __int_as_float(oc[r].shader | (int)(live[r] << T3D_LIVE_SHIFT) | (in[r].side << 30)));
            pool.parent[g] = in[r].parent;
            if (want_debug) pool.dbg[g] = dbg_idx[r];
        }
        const V3 o = in[r].o;
        const ChildRays ch = child_rays(oc[r], in[r]);
        if (oc[r].want_t && stored[r]) {
            const int tloc = base_r + off_t[r];
            if (tloc < wc.shard_cap) {
                RayRec q;
                q.o = make_float4(ch.tstart.x, ch.tstart.y, ch.tstart.z, ch.t_ior);
                q.d = make_float4(ch.tdir.x, ch.tdir.y, ch.tdir.z, __int_as_float(ch.t_obj));
                q.p = make_float4(o.x, o.y, o.z, __int_as_float((int)(wc.node_base + node[r])));
                q.k = make_uint4(in[r].ps, ch.child_heap + 1u, 1u, in[r].pixel_hash);
                rays_out[wc.shard * wc.shard_cap + tloc] = q;
            } else { cnt->overflow = 1; rp.sticky[0] = 1; }
        }
        if (oc[r].want_b && stored[r]) {
            const int bloc = base_r + off_b[r];
            if (bloc < wc.shard_cap) {
                RayRec q;
                q.o = make_float4(ch.bstart.x, ch.bstart.y, ch.bstart.z, in[r].ior);
                q.d = make_float4(ch.bdir.x, ch.bdir.y, ch.bdir.z, __int_as_float(in[r].inside_obj));
                q.p = make_float4(o.x, o.y, o.z, __int_as_float((int)(wc.node_base + node[r])));
                q.k = make_uint4(in[r].ps, ch.child_heap, 0u, in[r].pixel_hash);
                rays_out[wc.shard * wc.shard_cap + bloc] = q;
            } else { cnt->overflow = 1; rp.sticky[0] = 1; }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const V3 P = oc[r].P, N = oc[r].N;
        const float rough = oc[r].rough;
        if constexpr (RING > 0 && PAIRS) {
            // only nodes with a listed pair enter the ring
            const unsigned long long m_r = __ballot(live[r] != 0);
            if (live[r]) {
                const int slot = (ring_nodes + __popcll(m_r & wc.lt_mask)) % RING;
                ring_a[slot] = make_float4(P.x, P.y, P.z, rough);
                ring_b[slot] = make_float4(N.x, N.y, N.z, __int_as_float(stored[r] ? node[r] : 0));
                const int pos = pl->tail + pair_pos[r];
                for (int l = 0; l < wc.L; ++l)
                    if ((live[r] >> l) & 1u) pl->ring[(pos + __popc(live[r] & ((1u << l) - 1u))) % T3D_PAIRS] = (unsigned short)(slot | (l << 8));
            }
            if (oc[r].shaded) pl->culled += (uint32_t)(wc.L - __popc(live[r]));
            ring_nodes += __popcll(m_r);
        } else if constexpr (RING > 0) {
            // shaded nodes enter the wave's ring in lane order
            if (oc[r].shaded && wc.L > 0) {
                const int slot = (ring_nodes + __popcll(m_n[r] & wc.lt_mask)) % RING;
                ring_a[slot] = make_float4(P.x, P.y, P.z, rough);
                ring_b[slot] = make_float4(N.x, N.y, N.z, __int_as_float(stored[r] ? node[r] : 0));
            }
            ring_nodes += __popcll(m_n[r]);
        } else {
            // no ring: ring_a / ring_b are arrays in global memory indexed by the node's position in its level
            if (stored[r] && wc.L > 0) {
                ring_a[node[r]] = make_float4(P.x, P.y, P.z, rough);
                ring_b[node[r]] = make_float4(N.x, N.y, N.z, 0.0f);
            }
        }
    }
    if constexpr (RING > 0 && PAIRS) pl->tail += pairs_total;
}

template <bool LEVEL0, bool ONE = true>
__device__ __forceinline__ void load_ray(const DevScene& sc, const RenderParams& rp, const RayRec* __restrict__ rays_in,
                                         int idx, int n_in, int queue_base, RayIn& in, const RayRec* fetched = nullptr) {
    const int single = ONE ? rp.single : 0;
    in.active = idx < n_in;
    in.i = LEVEL0 ? idx : queue_base + idx;
    in.o = mk(0, 0, 0); in.d = mk(0, 0, 1); in.pstart = mk(0, 0, 0);
    in.ior = 1.0f; in.inside_obj = -1; in.parent = -1; in.side = 0;
    in.ps = 0; in.heap = 1; in.pixel_hash = 0; in.px = 0; in.py = 0; in.smp = 0;
    if constexpr (!LEVEL0) {
        // The record is fetched by every lane (a lane without a ray reads the first entry of the shard's queue and ignores
        // it): the requests of a lane's two rays are then not tied to the rays' own conditions and are all in flight together.
        // (`fetched`: the caller has done that already, see k_level)
        const RayRec r = fetched ? *fetched : rays_in[in.active ? in.i : queue_base];
        if (in.active) {
            in.o = mk(r.o.x, r.o.y, r.o.z); in.ior = r.o.w;
            in.d = mk(r.d.x, r.d.y, r.d.z); in.inside_obj = __float_as_int(r.d.w);
            in.pstart = mk(r.p.x, r.p.y, r.p.z); in.parent = __float_as_int(r.p.w);
            // a level that overflowed its queue (the frame is then traced again with more room) may hand on entries
            // that nobody wrote: whatever they hold, the parent they name stays inside the node arrays
            in.parent = min(max(in.parent, 0), (rp.bounces + 1) * rp.cap - 1);
            in.ps = r.k.x; in.heap = r.k.y; in.side = (int)r.k.z; in.pixel_hash = r.k.w;
        }
        return;
    }
    if (in.active) {
        if (LEVEL0) {
            if (rp.tiled) {
                // work order: 16 x 8 pixel tiles, so that the 128 rays of a round stay close together
                const int wpix = in.i / rp.samples, wsmp = in.i - wpix * rp.samples;
                const int t = wpix >> 7, within = wpix & 127, tiles_x = rp.width >> 4;
                const int ty = t / tiles_x, tx = t - ty * tiles_x;
                in.i = (((ty << 3) + (within >> 4)) * rp.width + (tx << 4) + (within & 15)) * rp.samples + wsmp;
            }
            in.ps = (uint32_t)in.i;
            in.parent = in.i;
            int pix = in.i / rp.samples;
            in.smp = in.i - pix * rp.samples;
            int ly = pix / rp.width;
            in.px = pix - ly * rp.width;
            int vrow = rp.vrow0 + ly;
            uint32_t seed = rp.seed;
            if (rp.frame_rows) { const int frame = vrow / rp.frame_rows; vrow -= frame * rp.frame_rows; seed += (uint32_t)frame * rp.seed_step; }
            in.py = rp.row0 + vrow * rp.row_stride;
            if (single) { in.px = rp.px0; in.py = rp.py0; in.smp = rp.sample0; }
            in.pixel_hash = keyed_pixel_hash(seed, (uint32_t)in.px, (uint32_t)in.py);
            in.heap = (((uint32_t)in.smp & 0xFFFFu) << 16) | 1u;
            if (single == 2) {
                in.o = mk(rp.ray_o[0], rp.ray_o[1], rp.ray_o[2]); in.d = mk(rp.ray_d[0], rp.ray_d[1], rp.ray_d[2]);
                in.ior = rp.ray_ior; in.inside_obj = rp.ray_obj;
            } else {
                camera_ray(sc.cam, rp, in.px, in.py, in.o, in.d);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_level: everything one tree level needs, one launch.
//
// Every WAVE is an autonomous scheduler over two kinds of closest-hit work, two queries per lane per round:
//   trace round   the next 128 rays of the level's queue: closest hit, classification, child rays,
//                 node records; the shaded nodes also enter the wave's ring buffer in LDS
//   shadow round  128 (node, light) pairs from the ring: shadow-ray closest hit -> li[node][light]
// A shadow round runs whenever 128 pairs are pending (or the queue is exhausted), so lanes stay busy even
// though only part of the rays produce shaded nodes, and node data for shading never leaves LDS.
// There is no block-level synchronisation at all.
//
// Queues are sharded T3D_SHARDS ways: wave w appends to shard w % T3D_SHARDS (own counter in its own
// 64-byte line, own slice of the arrays) with wave64 ballot + prefix-popcount compaction and ONE atomic
// per counter per round of 128 rays (k_bvh_process: per 64).
// ------------------------------------------------------------------------------------------------
template <bool LEVEL0, int WAVES, bool PAIRS>
__global__ __launch_bounds__(64 * WAVES, 4) void k_level(DevScene sc, GEO_PARAMS, const DevLight* __restrict__ light_rows, RenderParams rp, int level,
                                                      const RayRec* __restrict__ rays_in, RayRec* __restrict__ rays_out,
                                                      NodePool pool, Counters* cnt, TraceOut out) {
    constexpr int RING = T3D_RING;
    constexpr bool ONE = WAVES != 1;   // the tree of a single ray is traced by one workgroup of several waves; image bands by one-wave workgroups
    __shared__ float4 ring_a_all[WAVES][RING];  // P.xyz, roughness
    __shared__ float4 ring_b_all[WAVES][RING];  // unit normal, int bits: node index within the level
    __shared__ unsigned short pair_ring_all[PAIRS ? WAVES : 1][PAIRS ? T3D_PAIRS : 1];
    WaveCtx wc;
    wc.lane = threadIdx.x & 63; wc.wid = threadIdx.x >> 6;
    wc.lt_mask = (1ull << wc.lane) - 1ull;
    const int gw = blockIdx.x * WAVES + wc.wid;  // global wave id
    const int nw = gridDim.x * WAVES;            // multiple of T3D_SHARDS (host guarantees)
    wc.shard = gw & ((1 << rp.shard_bits) - 1);
    wc.shard_cap = rp.cap >> rp.shard_bits;
    wc.level = level;
    wc.node_base = (size_t)level * rp.cap;
    wc.L = sc.n_lights;
    float4* ring_a = ring_a_all[wc.wid];
    float4* ring_b = ring_b_all[wc.wid];
    const int L = wc.L;
    // input: level 0 is the dense list of primary samples; deeper levels read this wave's shard of the queue
    const int n_in = LEVEL0 ? rp.n0 : min(cnt->n_rays[level][wc.shard].v, wc.shard_cap);
    const int queue_base = wc.shard * wc.shard_cap;
    int chunk = LEVEL0 ? gw : gw >> rp.shard_bits;  // in units of cs rays
    const int chunk_step = LEVEL0 ? nw : nw >> rp.shard_bits;
    // Rays per trace round.  A launch with little work (a thin band of rows, one rank's share of a frame) is
    // bound by the number of rounds one wave runs back to back -- a trace round and the shadow rounds of the
    // nodes it creates -- not by throughput: such a launch is cut into more, smaller chunks (64 or 32 rays,
    // one per lane) as long as every chunk still finds a resident wave.  A caller that keeps several bands in flight on
    // several contexts hides that latency itself and asks for full rounds, which waste no slots (t3d_set_rounds).
    const int resident_all = nw < rp.resident_waves ? nw : rp.resident_waves;
    const int resident = LEVEL0 ? resident_all : resident_all >> rp.shard_bits;
    const int cs = rp.full_rounds || n_in > 64 * resident ? 128 : (n_in > 32 * resident ? 64 : 32);
    int ring_nodes = 0;  // nodes appended to the ring so far      (wave-uniform)
    int items_done = 0;  // (node, light) pairs already traced     (wave-uniform)
    const float inv_L = 1.0f / (float)(L > 0 ? L : 1);
    PairList pl{PAIRS ? pair_ring_all[wc.wid] : nullptr, 0, 0u};
#ifdef T3D_WAVE_TRACE
    const unsigned long long wt_start = __builtin_amdgcn_s_memrealtime();
    unsigned wt_trace = 0, wt_shadow = 0;
#endif
#ifdef T3D_COUNT_GROUPS
    GroupCount gct{0, 0, 0, 0}, gcs{0, 0, 0, 0};
    GroupCount *gcp_t = &gct, *gcp_s = &gcs;
#else
    GroupCount *gcp_t = nullptr, *gcp_s = nullptr;
#endif
    for (;;) {
        const int n_items = PAIRS ? pl.tail : ring_nodes * L;
        const int pending = n_items - items_done;
        const bool more = (long long)chunk * cs < n_in;
        // nodes whose pairs are all done leave the ring: its 256 entries always hold 127 pending + 128 new nodes
        if (pending >= 128 || (!more && pending > 0)) {
            // ---------------- shadow round: pairs items_done + lane and items_done + 64 + lane ----------------
            const int item0 = items_done + wc.lane, item1 = item0 + 64;
            const bool run0 = item0 < n_items, run1 = item1 < n_items;
            int nd0, nd1, l0, l1;
            if constexpr (PAIRS) {
                // a lane without a pair of its own repeats the round's first one and stores nothing
                const int e0 = pl.ring[(run0 ? item0 : items_done) % T3D_PAIRS], e1 = pl.ring[(run1 ? item1 : items_done) % T3D_PAIRS];
                nd0 = e0 & 255; l0 = e0 >> 8; nd1 = e1 & 255; l1 = e1 >> 8;
            } else {
                // item = node * L + light.  items_done is wave-uniform, so its quotient and remainder by L are scalar
                // work; per lane only a small offset (< L + 128) is divided, exactly, in float.
                const int nd_base = __builtin_amdgcn_readfirstlane(items_done / L), l_base = __builtin_amdgcn_readfirstlane(items_done % L);
                const int x0 = l_base + (run0 ? wc.lane : 0), x1 = l_base + (run1 ? wc.lane + 64 : 0);
                const int q0 = (int)(((float)x0 + 0.5f) * inv_L), q1 = (int)(((float)x1 + 0.5f) * inv_L);
                nd0 = nd_base + q0; nd1 = nd_base + q1;
                l0 = x0 - q0 * L; l1 = x1 - q1 * L;
            }
            const float4 rb0 = ring_b[nd0 % RING], rb1 = ring_b[nd1 % RING];
            const ShadowRay s0 = shadow_ray(sc, ring_a[nd0 % RING], rb0, l0);
            const ShadowRay s1 = shadow_ray(sc, ring_a[nd1 % RING], rb1, l1);
            // A shadow ray is aimed at the very position of its light: a Light's centre, the centroid of an emissive face
            // (T3DR.py:789-816).  Unless something blocks it, the first thing it hits is that Light or that face, so the
            // search is first bounded just behind the light's position -- the closest hit below the bound is the closest
            // hit, and the plane groups that lie behind the light for every ray of the wave are skipped.  The reference
            // has no distance check (T3DR.py:762-778): a ray that finds nothing up to there (a Light of radius zero, a
            // start inside the Light) goes on through the whole scene.
            Hit h0, h1;
            closest_hit2<false, true>(GEO_ARGS, sc.geo, s0.start, s0.Ld, s1.start, s1.Ld, true, false, false, h0, h1, gcp_s, s0.reach, s1.reach);
            if (__any((int)(h0.prim < 0) | (int)(h1.prim < 0))) {   // wave-uniform
                Hit g0, g1;
                closest_hit2<false, true>(GEO_ARGS, sc.geo, s0.start, s0.Ld, s1.start, s1.Ld, true, false, false, g0, g1, gcp_s);
                if (h0.prim < 0) h0 = g0;
                if (h1.prim < 0) h1 = g1;
            }
            const float I0 = shadow_intensity(sc, spheres, s0, h0), I1 = shadow_intensity(sc, spheres, s1, h1);
            if (run0) pool.li[(wc.node_base + (size_t)__float_as_int(rb0.w)) * L + l0] = I0;
            if (run1) pool.li[(wc.node_base + (size_t)__float_as_int(rb1.w)) * L + l1] = I1;
            items_done += pending < 128 ? pending : 128;
#ifdef T3D_WAVE_TRACE
            ++wt_shadow;
#endif
        } else if (more) {
#ifdef T3D_WAVE_TRACE
            ++wt_trace;
#endif
            // ---------------- trace round: rays chunk*cs + lane and (cs = 128) chunk*cs + 64 + lane ----------------
            RayIn in[2];
            if constexpr (LEVEL0) {
                load_ray<LEVEL0, ONE>(sc, rp, rays_in, chunk * cs + wc.lane, wc.lane < cs ? n_in : 0, queue_base, in[0]);
                load_ray<LEVEL0, ONE>(sc, rp, rays_in, chunk * cs + 64 + wc.lane, cs == 128 ? n_in : 0, queue_base, in[1]);
            } else {
                // both records of the lane are requested before either is looked at (a lane without a ray reads the first
                // entry of its shard's queue and ignores it)
                const int i0 = chunk * cs + wc.lane, i1 = i0 + 64, n0 = wc.lane < cs ? n_in : 0, n1 = cs == 128 ? n_in : 0;
                const RayRec q0 = rays_in[queue_base + (i0 < n0 ? i0 : 0)], q1 = rays_in[queue_base + (i1 < n1 ? i1 : 0)];
                asm volatile("" :: "v"(q0.o.x), "v"(q0.d.x), "v"(q0.p.x), "v"(q0.k.x), "v"(q1.o.x), "v"(q1.d.x), "v"(q1.p.x), "v"(q1.k.x));
                load_ray<LEVEL0, ONE>(sc, rp, rays_in, i0, n0, queue_base, in[0], &q0);
                load_ray<LEVEL0, ONE>(sc, rp, rays_in, i1, n1, queue_base, in[1], &q1);
            }
            const bool lights = LEVEL0 ? (rp.lights != 0) : true;
            Hit h[2];  // back faces culled iff ray.ior == 1 (T3DR.py:961)
            closest_hit2<LEVEL0>(GEO_ARGS, sc.geo, in[0].o, in[0].d, in[1].o, in[1].d, lights, in[0].ior == 1.0f, in[1].ior == 1.0f, h[0], h[1], gcp_t);
            process_rays<LEVEL0, RING, PAIRS, ONE, 2>(sc, light_rows, rp, wc, in, h, rays_out, pool, cnt, out, ring_a, ring_b, ring_nodes, &pl);
            chunk += chunk_step;
            // the ring is private to this wave; LDS operations of one wave complete in order, the fences
            // only stop the compiler from moving the next round's reads above these writes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            break;
        }
    }
#ifdef T3D_WAVE_TRACE
    if (wc.lane == 0 && gw < T3D_WAVE_TRACE_MAX) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wave_trace[level][gw][0] = wt_start;
        g_wave_trace[level][gw][1] = __builtin_amdgcn_s_memrealtime();
        g_wave_trace[level][gw][2] = (unsigned long long)wt_trace | ((unsigned long long)wt_shadow << 32);
        g_wave_trace[level][gw][3] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
#endif
#ifdef T3D_COUNT_GROUPS
    if (wc.lane == 0) {
        atomicAdd(&g_group_count[0], (unsigned long long)gct.calls); atomicAdd(&g_group_count[1], (unsigned long long)gct.visited);
        atomicAdd(&g_group_count[2], (unsigned long long)gct.entered); atomicAdd(&g_group_count[3], (unsigned long long)gct.steps);
        atomicAdd(&g_group_count[4], (unsigned long long)gcs.calls); atomicAdd(&g_group_count[5], (unsigned long long)gcs.visited);
        atomicAdd(&g_group_count[6], (unsigned long long)gcs.entered); atomicAdd(&g_group_count[7], (unsigned long long)gcs.steps);
    }
#endif
    if constexpr (PAIRS) {
        // shadow rays answered without being traced: wave sum, one atomic on the wave's own counter line
        unsigned long long f = pl.culled;
        for (int off = 32; off > 0; off >>= 1) f += __shfl_xor(f, off);
        if (wc.lane == 0 && f) atomicAdd(&cnt->bvh_stat[wc.shard][4], f);
    }
}

// ------------------------------------------------------------------------------------------------
// LBVH path: one tree level = three launches.
//   k_bvh_trace    closest hit of every ray of the level -> hits[]
//   k_bvh_process  classification, child rays, node records (process_rays) from the stored hits; the shading
//                  inputs of every new node (hit point, normal, roughness) go to src_a / src_b
//   k_bvh_shadow   closest hit of the shadow ray of every (node, light) pair -> li[node][light]
// Rays of one wave need very different numbers of traversal steps, and a wave lasts as long as its slowest lane.
// The two traversal kernels therefore never wait for a whole batch: a wave takes 64-ray chunks of its shard's list
// from a counter (one atomic per chunk), so waves that get cheap rays simply take more, and hands the next ray of
// its chunk to a lane as soon as the lane's ray is finished.  Refills are collective (ballot + prefix count) and
// happen once T3D_REFILL_IDLE lanes are idle, so that the ray set-up is not run for one lane at a time.
// ------------------------------------------------------------------------------------------------
#ifndef T3D_REFILL_IDLE
#define T3D_REFILL_IDLE 16
#endif

// The part of a list a wave hands out to its lanes (all wave-uniform).
struct RayFeed {
    long long pos = 0, end = 0;   // entries pos .. end - 1 of the current chunk are still to be handed out
    bool exhausted = false;       // the list has no chunk left
};
// Takes the next chunk when the current one is used up.  Called by all 64 lanes.  Lists of level 0 are not sharded:
// there shard s owns chunks s, s + 64, ... of the one list (INTERLEAVED).
template <bool INTERLEAVED>
__device__ __forceinline__ void feed_refresh(RayFeed& f, PaddedCount* counter, int lane, int shard, int shard_bits, long long n_in) {
    if (f.pos < f.end || f.exhausted) return;
    int j = 0;
    if (lane == 0) j = atomicAdd(&counter->v, 1);
    j = __builtin_amdgcn_readfirstlane(j);
    const long long base = (INTERLEAVED ? ((long long)j << shard_bits) + shard : (long long)j) * 64;
    if (base >= n_in) f.exhausted = true;
    else { f.pos = base; f.end = base + 64 < n_in ? base + 64 : n_in; }
}

__device__ __forceinline__ void bvh_stats_flush(const BvhCount& bc, Counters* cnt, int lane, int shard) {
    // traversal statistics: wave sum, one atomic per counter on the wave's own counter line
    unsigned long long a = bc.nodes, b = bc.tris, c = bc.spheres, e = bc.deep, f = bc.culled;
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off); b += __shfl_xor(b, off); c += __shfl_xor(c, off); e += __shfl_xor(e, off); f += __shfl_xor(f, off);
    }
    if (lane == 0) {
        atomicAdd(&cnt->bvh_stat[shard][0], a);
        atomicAdd(&cnt->bvh_stat[shard][1], b);
        atomicAdd(&cnt->bvh_stat[shard][2], c);
        if (e) atomicAdd(&cnt->bvh_stat[shard][3], e);
        if (f) atomicAdd(&cnt->bvh_stat[shard][4], f);
    }
}

template <bool LEVEL0>
__global__ __launch_bounds__(T3D_BLOCK, T3D_BVH_WAVES) void k_bvh_trace(DevScene sc, const DevSphere* __restrict__ spheres, RenderParams rp,
                                                                         int level, const RayRec* __restrict__ rays_in,
                                                                         uint2* __restrict__ hits, Counters* cnt, PaddedCount* fetch) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int gw = blockIdx.x * T3D_WAVES + (threadIdx.x >> 6);
    const int shard = gw & ((1 << rp.shard_bits) - 1);   // the grid holds a multiple of that many waves (host guarantees)
    const int shard_cap = rp.cap >> rp.shard_bits;
    const int n_in = LEVEL0 ? rp.n0 : min(cnt->n_rays[level][shard].v, shard_cap);
    const int queue_base = shard * shard_cap;
    const bool lights = LEVEL0 ? (rp.lights != 0) : true;
    __shared__ int lds_stack[T3D_WAVES][T3D_BVH_LDS_STACK][64];
    int* lds = &lds_stack[threadIdx.x >> 6][0][lane];
    BvhLane st;
    int* deep = sc.bvh.spill + ((size_t)gw * T3D_BVH_DEEP_STACK) * 64 + lane;
    BvhCount bc{0, 0, 0, 0, 0};
    bool have = false;     // a ray is in flight in this lane
    bool pending = false;  // a finished ray's result is still in st
    int mine = 0;          // where this lane's result goes
    RayFeed feed;
    for (;;) {
        const int n_idle = __popcll(__ballot(!have));
        if ((!feed.exhausted && n_idle >= T3D_REFILL_IDLE) || n_idle == 64) {
            // service point, shared by all idle lanes: results out, next rays in
            const unsigned long long idle = __ballot(!have);
            if (pending) {
                hits[mine] = make_uint2(__float_as_uint(st.best) | ((uint32_t)st.back << 31), (uint32_t)st.prim);
                pending = false;
            }
            feed_refresh<LEVEL0>(feed, &fetch[shard], lane, shard, rp.shard_bits, n_in);
            if (feed.exhausted && n_idle == 64) break;  // nothing in flight and nothing left to hand out
            const long long idx = feed.pos + __popcll(idle & lt_mask);
            if (!have && idx < feed.end) {
                RayIn in;
                load_ray<LEVEL0>(sc, rp, rays_in, (int)idx, n_in, queue_base, in);
                mine = in.i;
                bvh_begin(st, in.o, in.d, lights, in.ior == 1.0f);  // back faces culled iff ray.ior == 1 (T3DR.py:961)
                have = true;
            }
            feed.pos = feed.pos + n_idle < feed.end ? feed.pos + n_idle : feed.end;
        }
        if (have && bvh_step(sc, st, lds, deep, bc, &cnt->bvh_stack_overflow, rp.sticky)) { have = false; pending = true; }
    }
    bvh_stats_flush(bc, cnt, lane, shard);
}

template <bool LEVEL0>
__global__ __launch_bounds__(T3D_BLOCK) void k_bvh_process(DevScene sc, const DevSphere* __restrict__ spheres, RenderParams rp, int level,
                                                           const RayRec* __restrict__ rays_in, RayRec* __restrict__ rays_out,
                                                           const uint2* __restrict__ hits, NodePool pool, Counters* cnt, TraceOut out,
                                                           float4* __restrict__ src_a, float4* __restrict__ src_b) {
    WaveCtx wc;
    wc.lane = threadIdx.x & 63; wc.wid = threadIdx.x >> 6;
    wc.lt_mask = (1ull << wc.lane) - 1ull;
    const int gw = blockIdx.x * T3D_WAVES + wc.wid;
    const int nw = gridDim.x * T3D_WAVES;
    wc.shard = gw & ((1 << rp.shard_bits) - 1);
    wc.shard_cap = rp.cap >> rp.shard_bits;
    wc.level = level;
    wc.node_base = (size_t)level * rp.cap;
    wc.L = sc.n_lights;
    const int n_in = LEVEL0 ? rp.n0 : min(cnt->n_rays[level][wc.shard].v, wc.shard_cap);
    const int queue_base = wc.shard * wc.shard_cap;
    const int chunk_step = LEVEL0 ? nw : nw >> rp.shard_bits;
    int unused = 0;
    for (int chunk = LEVEL0 ? gw : gw >> rp.shard_bits; (long long)chunk * 64 < n_in; chunk += chunk_step) {
        RayIn in[1];
        load_ray<LEVEL0>(sc, rp, rays_in, chunk * 64 + wc.lane, n_in, queue_base, in[0]);
        Hit h[1];
        h[0].t = __builtin_inff(); h[0].prim = -1; h[0].back = 0;
        if (in[0].active) {
            const uint2 r = hits[in[0].i];
            h[0].t = __uint_as_float(r.x & 0x7fffffffu); h[0].back = (int)(r.x >> 31); h[0].prim = (int)r.y;
        }
        process_rays<LEVEL0, 0>(sc, sc.lights, rp, wc, in, h, rays_out, pool, cnt, out, src_a, src_b, unused);
    }
}

__global__ __launch_bounds__(T3D_BLOCK, T3D_BVH_WAVES) void k_bvh_shadow(DevScene sc, const DevSphere* __restrict__ spheres, RenderParams rp,
                                                                          int level, const float4* __restrict__ src_a,
                                                                          const float4* __restrict__ src_b, NodePool pool, Counters* cnt,
                                                                          PaddedCount* fetch) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int gw = blockIdx.x * T3D_WAVES + (threadIdx.x >> 6);
    const int shard = gw & ((1 << rp.shard_bits) - 1);
    const int shard_cap = rp.cap >> rp.shard_bits;
    const int L = sc.n_lights;
    const long long n_in = (long long)min(cnt->n_nodes[level][shard].v, shard_cap) * L;  // (node, light) pairs of this shard
    const size_t node_base = (size_t)level * rp.cap;
    __shared__ int lds_stack[T3D_WAVES][T3D_BVH_LDS_STACK][64];
    int* lds = &lds_stack[threadIdx.x >> 6][0][lane];
    BvhLane st;
    int* deep = sc.bvh.spill + ((size_t)gw * T3D_BVH_DEEP_STACK) * 64 + lane;
    BvhCount bc{0, 0, 0, 0, 0};
    bool have = false, pending = false;
    int my_node = 0, my_light = 0;
    RayFeed feed;
    for (;;) {
        const int n_idle = __popcll(__ballot(!have));
        if ((!feed.exhausted && n_idle >= T3D_REFILL_IDLE) || n_idle == 64) {
            const unsigned long long idle = __ballot(!have);
            if (pending) {
                // the shading inputs are read again rather than kept in registers across the traversal
                const ShadowRay s = shadow_ray(sc, src_a[my_node], src_b[my_node], my_light);
                Hit h;
                h.t = st.best; h.prim = st.prim; h.back = st.back;
                pool.li[(node_base + (size_t)my_node) * L + my_light] = shadow_intensity(sc, spheres, s, h);
                pending = false;
            }
            feed_refresh<false>(feed, &fetch[shard], lane, shard, rp.shard_bits, n_in);
            if (feed.exhausted && n_idle == 64) break;
            const long long idx = feed.pos + __popcll(idle & lt_mask);
            if (!have && idx < feed.end) {
                const int nd = (int)(idx / L);
                my_light = (int)(idx - (long long)nd * L);
                my_node = shard * shard_cap + nd;
                const ShadowRay s = shadow_ray(sc, src_a[my_node], src_b[my_node], my_light);
                if (shadow_dead(sc, s)) {
                    // this light adds nothing at this node whatever lies between them: answered here, the lane stays idle
                    pool.li[(node_base + (size_t)my_node) * L + my_light] = 0.0f;
                    bc.culled++;
                } else {
                    bvh_begin(st, s.start, s.Ld, true, false);
                    have = true;
                }
            }
            feed.pos = feed.pos + n_idle < feed.end ? feed.pos + n_idle : feed.end;
        }
        if (have && bvh_step(sc, st, lds, deep, bc, &cnt->bvh_stack_overflow, rp.sticky)) { have = false; pending = true; }
    }
    bvh_stats_flush(bc, cnt, lane, shard);
}

// ------------------------------------------------------------------------------------------------
// k_resolve: computeColor (T3DR.py:1043-1083) for one level, pushing the result to the parent
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T3D_BLOCK) void k_resolve(DevScene sc, RenderParams rp, int level, NodePool pool,
                                                       const Counters* __restrict__ cnt, float4* sample_color,
                                                       t3d_node* dbg, int dbg_cap, uint8_t* __restrict__ bgr, float* __restrict__ radiance,
                                                       uint4* __restrict__ zero, int n_zero) {
    // level 0 also clears the counters of the NEXT band (the context has two sets, used in turn): a band needs no fill
    if (level == 0)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += gridDim.x * blockDim.x) zero[i] = make_uint4(0u, 0u, 0u, 0u);
    const size_t node_base = (size_t)level * rp.cap;
    const int shard_cap = rp.cap >> rp.shard_bits;
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;  // nw: multiple of the shard count
    const int shard = gw & ((1 << rp.shard_bits) - 1);
    const int n = min(cnt->n_nodes[level][shard].v, shard_cap);
    for (int k = (gw >> rp.shard_bits) * 64 + lane; k < n; k += (nw >> rp.shard_bits) * 64) {
        const size_t g = node_base + (size_t)shard * shard_cap + (size_t)k;
        const float4 c = pool.c[g], d = pool.d[g];
        const int meta = __float_as_int(d.w);
        // a scene with few lights on the brute-force path: the record lists the lights that can add anything at the node,
        // the shadow rays of the others were never traced and their intensities (zero) never written
        const int shader = meta & (rp.live_masks ? (1 << T3D_LIVE_SHIFT) - 1 : 0x3FFFFFFF), side = (meta >> 30) & 1;
        const uint32_t live = rp.live_masks ? (uint32_t)(meta >> T3D_LIVE_SHIFT) & ((1u << T3D_PAIR_LIGHTS) - 1u) : 0xFFFFFFFFu;
        const int parent = pool.parent[g];
        DevShader sh = sc.shaders[shader];
        float I = 0.0f;
        for (int l = 0; l < sc.n_lights; ++l)   // light order, T3DR.py:894
            if (!rp.live_masks || ((live >> l) & 1u)) I += pool.li[g * sc.n_lights + l];
        float res[3];
        node_color(sc, rp, sh, c, d, I, res);
        if (rp.want_debug) {
            int di = pool.dbg[g];
            if (di >= 0 && di < dbg_cap) { dbg[di].color[0] = res[0]; dbg[di].color[1] = res[1]; dbg[di].color[2] = res[2]; }
        }
        if (level == 0) {
            if (bgr) store_pixel(bgr, radiance, parent, res[0], res[1], res[2], 1.0f);
            else sample_color[parent] = make_float4(res[0], res[1], res[2], 0.0f);
        } else {
            const float att = c.w;
            float4* slot = side ? pool.d : pool.c;
            *(float3*)&slot[parent] = make_float3(res[0] * att, res[1] * att, res[2] * att);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_finish: RGB.mean + clamp + BGR8 (T3DE.py:401-415, T3DR.py:1219-1221)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T3D_BLOCK) void k_finish(RenderParams rp, const float4* __restrict__ sample_color,
                                                      uint8_t* __restrict__ bgr, float* __restrict__ radiance) {
    const int npix = rp.nrows * rp.width;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        float r = 0, g = 0, b = 0;
        for (int k = 0; k < rp.samples; ++k) {
            float4 c = sample_color[(size_t)p * rp.samples + k];
            r += c.x; g += c.y; b += c.z;
        }
        store_pixel(bgr, radiance, p, r, g, b, (float)rp.samples);
    }
}

// One band's counters added to the totals of its call (one workgroup of 64 threads, queued behind the band).
__global__ __launch_bounds__(64) void k_add_counters(const Counters* __restrict__ cnt, CallTotals* tot, int levels, unsigned long long n0) {
    const int t = threadIdx.x;
    if (t < levels) {
        unsigned long long r = t == 0 ? n0 : 0ull, n = 0ull;
        for (int sh = 0; sh < T3D_SHARDS; ++sh) {
            if (t > 0) r += (unsigned long long)cnt->n_rays[t][sh].v;
            n += (unsigned long long)cnt->n_nodes[t][sh].v;
        }
        tot->rays[t] += r; tot->nodes[t] += n;
    } else if (t >= 32 && t < 40) {
        unsigned long long a = 0ull;
        for (int sh = 0; sh < T3D_SHARDS; ++sh) a += cnt->bvh_stat[sh][t - 32];
        tot->bvh[t - 32] += a;
    }
}

// closest hits for arbitrary rays (t3d_closest_hits); the second ray slot of every lane repeats the first
__global__ __launch_bounds__(T3D_BLOCK) void k_closest(DevScene sc, GEO_PARAMS, int n, const float* __restrict__ origins,
                                                       const float* __restrict__ dirs, int lights, int back_cull,
                                                       int32_t* out_obj, int32_t* out_face, float* out_t, int32_t* out_dir,
                                                       float* out_normal) {
    __shared__ int lds_stack[T3D_WAVES][T3D_BVH_LDS_STACK][64];
    // the grid is bounded (every wave owns a piece of the traversal spill area): blocks walk the rays in strides
    for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const int i = base + threadIdx.x;
        const bool active = i < n;
        const int j = active ? i : 0;
        V3 o = mk(origins[3 * j], origins[3 * j + 1], origins[3 * j + 2]);
        V3 d = normalize(mk(dirs[3 * j], dirs[3 * j + 1], dirs[3 * j + 2]));
        Hit h, h_dup;
        if (sc.bvh.nodes) {  // uniform over the launch
            BvhCount bc{0, 0, 0, 0, 0};
            int overflow = 0;
            const int gw = blockIdx.x * T3D_WAVES + (threadIdx.x >> 6);
            bvh_closest(sc, spheres, o, d, active, lights != 0, back_cull != 0, h, bc, &overflow, &lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63],
                        sc.bvh.spill + ((size_t)gw * T3D_BVH_DEEP_STACK) * 64 + (threadIdx.x & 63));
        } else {
            closest_hit2(GEO_ARGS, sc.geo, o, d, o, d, lights != 0, back_cull != 0, back_cull != 0, h, h_dup);
        }
        if (!active) continue;
        int obj = -1, face = -1, dirflag = 0;
        V3 N = mk(0, 0, 0);
        if (h.prim >= 0) {
            if (h.prim < sc.n_tris) {
                int4 meta = sc.tri_meta[h.prim];
                float4 rn = sc.tri_n[h.prim];
                obj = meta.x; face = meta.y; N = mk(rn.x, rn.y, rn.z);
                dirflag = (dot(N, d) < 0.0f) ? 0 : 1;
            } else {
                DevSphere sp = spheres[h.prim - sc.n_tris];
                V3 P = madd(o, d, h.t);
                obj = sp.obj; dirflag = h.back;
                N = normalize(mk(P.x - sp.cx, P.y - sp.cy, P.z - sp.cz));
            }
        }
        out_obj[i] = obj; out_face[i] = face; out_dir[i] = dirflag;
        out_t[i] = h.prim >= 0 ? h.t : __builtin_nanf("");
        if (out_normal) { out_normal[3 * i] = N.x; out_normal[3 * i + 1] = N.y; out_normal[3 * i + 2] = N.z; }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Work buffers for callers that must not allocate (t3d_render_rows_device_static): one array per GPU, part of the
// library's code object, alive from the moment the library is loaded on a device until the process ends.  In host code
// the array's NAME is only a stand-in; the address of a device's copy comes from hipGetSymbolAddress with that device
// current (static_scratch_for, called by t3d_create) and that pointer is what every launch and memset is given.
#define T3D_STATIC_SCRATCH_BYTES ((size_t)128 << 20)
static __device__ __align__(256) unsigned char t3d_static_scratch[T3D_STATIC_SCRATCH_BYTES];
#define T3D_MAX_DEVICES 64
static unsigned char* g_static_scratch[T3D_MAX_DEVICES];   // per device: address of its copy of the array, null until asked for
static std::mutex g_static_scratch_mutex;

// Address of the CURRENT device's copy (the caller has made `device` current); null with *err set on failure.
static unsigned char* static_scratch_for(int device, hipError_t* err) {
    *err = hipSuccess;
    if (device < 0 || device >= T3D_MAX_DEVICES) { *err = hipErrorInvalidDevice; return nullptr; }
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    if (!g_static_scratch[device]) {
        void* p = nullptr;
        *err = hipGetSymbolAddress(&p, HIP_SYMBOL(t3d_static_scratch));
        if (*err != hipSuccess) return nullptr;
        g_static_scratch[device] = (unsigned char*)p;
    }
    return g_static_scratch[device];
}

// Work buffers of one render call (a band of rows, or the tree of a single ray).
struct Work {
    size_t cap = 0;      // entries per level in the ray queues and node arrays
    int levels = 0;
    int li_lights = 0;   // lights the li array was sized for
    RayRec* rays[2] = {nullptr, nullptr};
    NodePool pool{};
    // LBVH path only: closest hit of every ray of the current level; shading inputs of its new nodes
    uint2* hits = nullptr;
    float4* src_a[2] = {nullptr, nullptr}; float4* src_b[2] = {nullptr, nullptr};  // by level parity: see render_band
    size_t n0 = 0;       // primary samples sample_color holds
    float4* sample_color = nullptr;
    uint64_t gen = 0;    // counts (re)allocations: a launch graph holds the addresses
};

// A surface-area hierarchy being cut on a host thread while the scene is already traced through the device's tree
// (t3d_set_background_tree).  The thread touches nothing but this object; the context looks at `state` (acquire) and
// joins the thread before it reads the tree or lets the object go.
struct BackgroundTree {
    std::thread th;
    std::atomic<int> state{0};       // 0: building, 1: the tree is there, 2: the build failed (out of memory)
    std::vector<float> lo, hi;       // [3n] boxes of the primitives as uploaded
    int n = 0;
    HostTree tree;
};

struct t3d_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t upload_stream = nullptr;   // exists while t3d_upload_scene runs (UploadStream)
    hipEvent_t ev[(T3D_MAX_BOUNCES + 1) + 4] = {};
    std::string err;
    bool has_scene = false;
    int n_sm = 256;
    // scene buffers
    DevScene sc{};
    void* d_planes = nullptr; void* d_plane_first = nullptr; void* d_tri_uv = nullptr; void* d_tri_n = nullptr;
    void* d_bvh_nodes = nullptr; void* d_bvh_leaves = nullptr; void* d_bvh_spill = nullptr;
    int accel = T3D_ACCEL_AUTO;   // t3d_set_accel
    int host_tree_max = T3D_HOST_TREE_MAX_PRIMS;   // t3d_set_host_tree_limit
    bool use_bvh = false;         // the uploaded scene is traced through the LBVH
    bool background_tree = false; // t3d_set_background_tree
    int tree_source = T3D_TREE_NONE;            // T3D_TREE_*: where the uploaded scene's hierarchy comes from
    std::unique_ptr<BackgroundTree> bg_tree;    // a build that has not been adopted or dropped yet
    void* d_tri_meta = nullptr; void* d_spheres = nullptr; void* d_lights = nullptr;
    void* d_shaders = nullptr; void* d_obj_ior = nullptr; void* d_prim_info = nullptr;
    int n_objects = 0, n_sphere_objs = 0, n_light_objs = 0;
    // kept for t3d_update_scene (the same scene with spheres, Lights or the camera moved)
    std::vector<int> obj_types;               // T3D_OBJ_* of every object as uploaded
    std::vector<DevSphere> h_spheres;         // the spheres and Lights as uploaded (object, shader, kind)
    double tri_lo[3] = {0, 0, 0}, tri_hi[3] = {0, 0, 0};   // bounds of the triangles
    int up_tris = 0, up_shaders = 0, up_lights = 0;
    std::vector<float4> up_rows;              // t3d_update_scene: host staging that must outlive its asynchronous copies
    std::vector<DevLight> up_light_rows;
    LbvhWork lbvh_work;                       // working arrays of the hierarchy build (LBVH scenes)
    void* d_lbvh_verts = nullptr; void* d_lbvh_tri_ab = nullptr;
    // work buffers: one set for image bands, a small one for the tree of a single ray (t3d_trace_pixel / t3d_trace_ray), so
    // that a picked pixel neither frees the band buffers nor invalidates the launch graphs that hold their addresses
    Work work, single_work;
    hipStream_t shadow_stream = nullptr;             // k_bvh_shadow of level d runs beside the tracing of level d + 1
    hipStream_t capture_stream = nullptr;            // launch graphs are recorded here, never on a caller's stream: other threads of
                                                     // the process (a collective library's watchdog) may query events of that stream
                                                     // at any moment, which HIP refuses while the stream records
    hipEvent_t ev_traced[T3D_MAX_BOUNCES + 2] = {};  // level d classified: its shadow rays may start
    hipEvent_t ev_shadowed[T3D_MAX_BOUNCES + 2] = {};  // shadow rays of level d done
    Counters* d_cnt = nullptr;            // counters of the current / most recent band: one of the pair
    Counters* d_cnt_pair[2] = {nullptr, nullptr};
    int cnt_idx = 0;                      // d_cnt_pair[cnt_idx] is all zero between bands ...
    bool cnt_clean = true;                // ... as long as this says so
    Counters* h_cnt = nullptr;  // pinned
    PaddedCount* d_fetch = nullptr;  // LBVH path: chunk counters of the traversal kernels, [level][trace, shadow][shard]
    // staging for host outputs
    size_t out_alloc = 0;
    void* h_stage = nullptr; size_t stage_alloc = 0;  // pinned: device -> pinned -> caller's (pageable) buffers
    uint8_t* d_bgr = nullptr; float* d_rad = nullptr; int32_t* d_prim = nullptr; float* d_prim_t = nullptr;
    size_t mem_budget = (size_t)48 << 30;
    // the most recent band, for t3d_last_call_stats
    bool timing = true;       // record an event after every k_level launch (t3d_set_timing)
    bool last_timed = true;
    bool last_events = true;   // the band recorded its start and end
    bool last_valid = false; int last_levels = 0; size_t last_n0 = 0; int last_lights = 1; uint64_t last_launches = 0;
    hipStream_t last_stream = nullptr;
    // Launch graphs of repeated asynchronous bands: the same band (same options, rows, buffers, scene) asked for again
    // is one graph launch instead of ten kernel launches (0.1 ms of host time per call otherwise).
    // The seed of the random numbers is not part of a graph's key: only the first kernel of a band reads it (the key of
    // a pixel's random numbers is hashed once per primary ray and travels with the ray records), so a band asked for
    // again with another seed -- the next frame of an animation -- replays the same graph after that one kernel node's
    // parameters have been set anew (hipGraphExecKernelNodeSetParams).
    struct FrameGraph {
        bool valid = false; char key[320] = {}; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint64_t launches = 0;
        hipGraphNode_t level0 = nullptr;   // the k_level launch of level 0; null: the graph serves its own seed only
        uint32_t seed = 0;                 // the seed its level-0 node holds at the moment
    };
    FrameGraph graphs[4];
    int graph_next = 0;
    uint64_t graphs_captured = 0, graphs_replayed = 0, graphs_reseeded = 0;   // t3d_graph_counters
    char graph_miss[2][320] = {};   // bands that were launched kernel by kernel most recently: a graph is made at the second request
    int graph_miss_next = 0;
    bool graphs_enabled = true;
    bool full_rounds = false;       // t3d_set_rounds
    uint64_t scene_gen = 0;         // counts uploads
    // Overflow that nobody was waiting for.  An asynchronous render call (no stats) returns before its kernels have run, so
    // a trace tree that outgrows the queues or node arrays cannot be reported by that call.  The kernels raise these two
    // words in pinned host memory ([0] capacity, [1] traversal stack) besides the per-band counters; they stay raised until
    // the host looks: at the start of the next render call on the context and in t3d_last_call_stats, either of which then
    // returns T3D_E_CAPACITY for the earlier call (whose output is invalid) and lets later asynchronous calls start with
    // larger buffers.  Kernels only ever write there on an overflow.
    int* h_sticky = nullptr;
    size_t async_min_cap = 0;       // entries per level that asynchronous calls start with: twice what the last one that overflowed had
    size_t last_async_cap = 0;      // entries per level of the most recent asynchronous band
    bool last_async_static = false; // ... which was a band of t3d_render_rows_device_static
    int static_async_factor = 2;    // rays per sample and level that asynchronous static calls provide for: doubled after an overflow
    // cancellation (Photon.render's cancelCallback, T3DR.py:1227-1237)
    int (*cancel_cb)(void*) = nullptr; void* cancel_user = nullptr;
    // t3d_closest_hits: pinned staging the kernel reads its rays from and writes its answers to
    void* h_pick = nullptr; size_t pick_alloc = 0;
    t3d_node* d_dbg = nullptr; int dbg_alloc = 0;   // node records of a single ray's tree (t3d_trace_pixel / t3d_trace_ray)
    unsigned char* static_scratch = nullptr;        // this device's copy of t3d_static_scratch (t3d_render_rows_device_static)
    bool counters_on_host = false;   // h_cnt holds (or is about to receive) the counters of the most recent band
    // a call that is not waited for and is cut into several bands adds every band's counters up on the device
    CallTotals* d_totals = nullptr;
    CallTotals* h_totals = nullptr;   // pinned
    int call_bands = 1;               // bands of the most recent render call (totals are kept when > 1 and nobody waited)
    bool call_totals = false;         // d_totals holds the counters of the most recent call
    uint64_t call_n0 = 0;             // its primary samples
    hipEvent_t ev_call0 = nullptr;    // start of its first band (recorded when events are)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // t3d_render_rows_device_split: the caller's stream -> this context's, and back
};

static thread_local std::string g_create_err;
// HIP refuses legacy-stream calls of one host thread while another records a launch graph (and the recording is lost):
// the library's only such call (t3d_create) and its recordings take turns.
static std::mutex g_capture_mutex;
static std::once_flag g_legacy_touched[T3D_MAX_DEVICES];

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return T3D_E_DEVICE;                                                                       \
        }                                                                                              \
    } while (0)

extern "C" int t3d_abi_version(void) { return T3D_ABI_VERSION; }

extern "C" int t3d_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return T3D_E_DEVICE;
    return n;
}

extern "C" const char* t3d_last_error(const t3d_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" void t3d_destroy(t3d_ctx* c);

extern "C" int t3d_create(int device, t3d_ctx** out) {
    if (!out) return T3D_E_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_err = "no HIP device visible"; return T3D_E_DEVICE; }
    if (device < 0 || device >= n) { g_create_err = "device index out of range"; return T3D_E_INVALID; }
    t3d_ctx* c = new t3d_ctx();
    c->device = device;
    // every failure below goes through t3d_destroy, which releases whatever exists by then (all handles start out null)
    auto fail = [&](int code, const std::string& what) { g_create_err = "t3d_create: " + what; t3d_destroy(c); return code; };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    if ((e = hipStreamCreateWithFlags(&c->shadow_stream, hipStreamNonBlocking)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    if ((e = hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    for (auto& ev : c->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    for (auto& ev : c->ev_traced) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    for (auto& ev : c->ev_shadowed) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_sm = prop.multiProcessorCount;
    if (getenv("T3D_NO_GRAPHS")) c->graphs_enabled = false;
    if (const char* bg = getenv("T3D_BACKGROUND_TREE")) c->background_tree = atoi(bg) != 0;   // t3d_set_background_tree's initial value
    // working-set budget of one band in MiB (default 48 GiB): small values make small frames take the paths of huge ones
    if (const char* mb = getenv("T3D_MEM_BUDGET_MB")) { const long v = atol(mb); if (v > 0) c->mem_budget = (size_t)v << 20; }
    if (hipMalloc((void**)&c->d_cnt_pair[0], sizeof(Counters)) != hipSuccess || hipMalloc((void**)&c->d_cnt_pair[1], sizeof(Counters)) != hipSuccess ||
        // (copies and fills never go through the legacy stream in this library, see UploadStream)
        hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMemsetAsync(c->d_cnt_pair[0], 0, sizeof(Counters), c->upload_stream) != hipSuccess ||
        hipMemsetAsync(c->d_cnt_pair[1], 0, sizeof(Counters), c->upload_stream) != hipSuccess ||
        hipStreamSynchronize(c->upload_stream) != hipSuccess ||
        hipMalloc((void**)&c->d_fetch, sizeof(PaddedCount) * (T3D_MAX_BOUNCES + 2) * 2 * T3D_SHARDS) != hipSuccess ||
        hipHostMalloc((void**)&c->h_cnt, sizeof(Counters), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_sticky, 64, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&c->d_totals, sizeof(CallTotals)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_totals, sizeof(CallTotals), hipHostMallocDefault) != hipSuccess)
        return fail(T3D_E_NOMEM, "counter allocation failed");
    hipStreamDestroy(c->upload_stream); c->upload_stream = nullptr;
    // The first context on a device puts one small fill on the device's legacy stream.  HIP gives a stream its place on
    // one of the process's few hardware queues when it first carries work; with the legacy stream placed first, four
    // streams of a caller that renders four chains at once end up on queues of their own (one rank's share of an 8-GPU
    // frame: 0.069 ms per frame, against 0.113 when the first work of the process went through other streams).
    if (device >= 0 && device < T3D_MAX_DEVICES)
        std::call_once(g_legacy_touched[device], [&] {
            std::lock_guard<std::mutex> lk(g_capture_mutex);
            (void)hipMemset(c->d_cnt_pair[0], 0, 64);
        });
    if ((e = hipEventCreate(&c->ev_call0)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    if ((e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    if ((e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess) return fail(T3D_E_DEVICE, hipGetErrorString(e));
    c->h_sticky[0] = c->h_sticky[1] = 0;
    c->static_scratch = static_scratch_for(device, &e);   // the device is current
    if (!c->static_scratch) return fail(T3D_E_DEVICE, std::string("address of the static work array: ") + hipGetErrorString(e));
    *out = c;
    return T3D_OK;
}

static void drop_background_tree(t3d_ctx* c) {
    if (!c->bg_tree) return;
    if (c->bg_tree->th.joinable()) c->bg_tree->th.join();   // (a build cannot be interrupted: at most a few tenths of a second)
    c->bg_tree.reset();
}

static void free_scene(t3d_ctx* c) {
    drop_background_tree(c);
    c->tree_source = T3D_TREE_NONE;
    c->lbvh_work.release();
    hipFree(c->d_lbvh_verts); hipFree(c->d_lbvh_tri_ab);
    c->d_lbvh_verts = c->d_lbvh_tri_ab = nullptr;
    hipFree(c->d_planes); hipFree(c->d_plane_first); hipFree(c->d_tri_uv); hipFree(c->d_tri_n);
    hipFree(c->d_bvh_nodes); hipFree(c->d_bvh_leaves); hipFree(c->d_bvh_spill);
    c->d_bvh_nodes = c->d_bvh_leaves = c->d_bvh_spill = nullptr; c->use_bvh = false;
    hipFree(c->d_tri_meta); hipFree(c->d_spheres); hipFree(c->d_lights);
    hipFree(c->d_shaders); hipFree(c->d_obj_ior); hipFree(c->d_prim_info);
    c->d_prim_info = nullptr;
    c->d_planes = c->d_plane_first = c->d_tri_uv = c->d_tri_n = nullptr;
    c->d_tri_meta = c->d_spheres = c->d_lights = c->d_shaders = c->d_obj_ior = nullptr;
    c->has_scene = false;
}
static void free_work(Work& w) {
    const uint64_t gen = w.gen + 1;
    hipFree(w.rays[0]); hipFree(w.rays[1]); hipFree(w.pool.parent); hipFree(w.pool.c);
    hipFree(w.pool.d); hipFree(w.pool.dbg); hipFree(w.pool.li);
    hipFree(w.hits);
    for (int k = 0; k < 2; ++k) { hipFree(w.src_a[k]); hipFree(w.src_b[k]); }
    hipFree(w.sample_color);
    w = Work{};
    w.gen = gen;
}

extern "C" void t3d_destroy(t3d_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->shadow_stream) hipStreamSynchronize(c->shadow_stream);
    free_scene(c); free_work(c->work); free_work(c->single_work);
    if (c->h_sticky) hipHostFree(c->h_sticky);
    if (c->h_pick) hipHostFree(c->h_pick);
    hipFree(c->d_dbg);
    hipFree(c->d_cnt_pair[0]); hipFree(c->d_cnt_pair[1]); hipFree(c->d_fetch);
    if (c->h_cnt) hipHostFree(c->h_cnt);
    hipFree(c->d_totals);
    if (c->h_totals) hipHostFree(c->h_totals);
    if (c->ev_call0) hipEventDestroy(c->ev_call0);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    for (auto& g : c->graphs) if (g.valid) { hipGraphExecDestroy(g.exec); hipGraphDestroy(g.graph); }
    hipFree(c->d_bgr); hipFree(c->d_rad); hipFree(c->d_prim); hipFree(c->d_prim_t);
    if (c->h_stage) hipHostFree(c->h_stage);
    for (auto& ev : c->ev) if (ev) hipEventDestroy(ev);
    for (auto& ev : c->ev_traced) if (ev) hipEventDestroy(ev);
    for (auto& ev : c->ev_shadowed) if (ev) hipEventDestroy(ev);
    if (c->shadow_stream) hipStreamDestroy(c->shadow_stream);
    if (c->capture_stream) hipStreamDestroy(c->capture_stream);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->upload_stream) hipStreamDestroy(c->upload_stream);
    delete c;
}

// Host staging of the large per-triangle records: not filled on allocation (a std::vector would be, on one thread, touching
// every page first), so that the threads which write the records are the ones that touch their pages.  Every entry
// must be written by the code that fills it.
template <typename T>
struct HostArray {
    T* p; size_t n;
    explicit HostArray(size_t count) : p((T*)malloc(sizeof(T) * (count ? count : 1))), n(count) { if (!p) throw std::bad_alloc(); }
    ~HostArray() { free(p); }
    HostArray(const HostArray&) = delete;
    HostArray& operator=(const HostArray&) = delete;
    T& operator[](size_t i) { return p[i]; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};

template <typename V>
static int upload(t3d_ctx* c, void** dst, const V& v) {
    typedef typename std::remove_cv<typename std::remove_reference<decltype(*v.data())>::type>::type T;
    size_t bytes = sizeof(T) * (v.empty() ? 1 : v.size());
    HIPCHK(c, hipMalloc(dst, bytes));
    if (!v.empty()) {
        hipStream_t st = c->upload_stream ? c->upload_stream : c->stream;
        HIPCHK(c, hipMemcpyAsync(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    return T3D_OK;
}

// The copies of an upload go through a stream that exists for the duration of the call.  Not the legacy stream (the
// synchronous copy functions use it, and HIP refuses them while ANOTHER host thread records a launch graph for another
// context), and not the context's own stream either: a stream that has carried work keeps its place on one of the
// process's few hardware queues, and a caller that renders on streams of its own -- four chains on four streams --
// then shares those queues with the contexts' idle streams (one rank's share of an 8-GPU frame, four calls in flight:
// 0.113 ms per frame against 0.068).
// The per-triangle loops of an upload on several host threads: f(first, last) over slices of [0, n).  Small inputs, and
// slices whose thread cannot be started, run on the calling thread; f must not throw.
template <typename F>
static void parallel_slices(size_t n, F f) {
    unsigned n_threads = n >= 65536 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (const char* env = getenv("T3D_UPLOAD_THREADS")) n_threads = n >= 2 ? (unsigned)std::max(1, atoi(env)) : 1u;
    if (n_threads <= 1) { f((size_t)0, n); return; }
    auto slice = [&](unsigned w) { f(n * w / n_threads, n * (w + 1) / n_threads); };
    std::vector<std::thread> pool;
    unsigned started = 1;
    try { for (; started < n_threads; ++started) pool.emplace_back(slice, started); } catch (const std::system_error&) {}
    slice(0);
    for (unsigned w = started; w < n_threads; ++w) slice(w);
    for (std::thread& th : pool) th.join();
}

struct UploadStream {
    t3d_ctx* c;
    explicit UploadStream(t3d_ctx* ctx) : c(ctx) {
        if (hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->upload_stream = nullptr; }
    }
    ~UploadStream() {
        if (c->upload_stream) { hipStreamSynchronize(c->upload_stream); hipStreamDestroy(c->upload_stream); c->upload_stream = nullptr; }
    }
};

// Pieces of the device scene that t3d_upload_scene and t3d_update_scene both make.
// scene bounds for the camera-ray pre-test, padded outward by 1e-4 of their size (tri_lo / tri_hi: bounds of the triangles)
static void set_scene_box(t3d_ctx* c, const std::vector<DevSphere>& spheres) {
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = c->tri_lo[a]; hi[a] = c->tri_hi[a]; }
    for (const DevSphere& sp : spheres) {
        const double r = sqrt((double)sp.r2) * 1.000001, cc[3] = {sp.cx, sp.cy, sp.cz};
        for (int a = 0; a < 3; ++a) { lo[a] = fmin(lo[a], cc[a] - r); hi[a] = fmax(hi[a], cc[a] + r); }
    }
    for (int a = 0; a < 3; ++a) {
        const double pad = 1e-4 * fmax(fmax(fabs(lo[a]), fabs(hi[a])), hi[a] - lo[a]) + 1e-30;
        c->sc.geo.box_lo[a] = lo[a] <= hi[a] ? (float)(lo[a] - pad) : 1.0f;   // no primitives: an empty box
        c->sc.geo.box_hi[a] = lo[a] <= hi[a] ? (float)(hi[a] + pad) : -1.0f;
    }
}
static void set_camera(DevScene& d, const t3d_scene_desc* s) {
    for (int k = 0; k < 3; ++k) d.cam.pos[k] = (float)s->cam_pos[k];
    // Euler.toMatrix (T3DE.py:70-94) in float64, rounded once
    const double cx = cos(s->cam_euler[0]), sx = sin(s->cam_euler[0]), cy = cos(s->cam_euler[1]), sy = sin(s->cam_euler[1]);
    const double cz = cos(s->cam_euler[2]), sz = sin(s->cam_euler[2]);
    const double m[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                         sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                         -sy, cy * sx, cy * cx};
    for (int k = 0; k < 9; ++k) d.cam.m[k] = (float)m[k];
    d.cam.zlen = (float)s->cam_zlen;
}
// the light list, and the sum over the lights of colour * 0.6 with saturating adds (T3DR.py:893)
static std::vector<DevLight> make_lights(DevScene& d, const t3d_scene_desc* s) {
    std::vector<DevLight> lights((size_t)s->n_lights);
    double lc[3] = {0, 0, 0};
    for (int l = 0; l < s->n_lights; ++l) {
        lights[l] = DevLight{(float)s->light_pos[3 * l], (float)s->light_pos[3 * l + 1], (float)s->light_pos[3 * l + 2], (float)s->light_strength[l]};
        for (int ch = 0; ch < 3; ++ch) { lc[ch] += s->light_color[3 * l + ch] * 0.6; if (lc[ch] > 255) lc[ch] = 255; }
    }
    for (int k = 0; k < 3; ++k) d.light_color[k] = (float)lc[k];
    return lights;
}
// the four prim_info rows of a sphere or Light
static void sphere_rows(const DevSphere& sp, const t3d_scene_desc* s, float4* rows) {
    auto ib = [](int v) { return __builtin_bit_cast(float, v); };
    const float rough = (float)s->shader_rough[sp.shader], refl = (float)s->shader_refl[sp.shader];
    rows[0] = make_float4(sp.cx, sp.cy, sp.cz, (float)s->obj_ior[sp.obj]);
    rows[1] = make_float4(ib(sp.obj), ib(-1), ib(sp.shader), ib(2 | (sp.is_light ? 1 : 0)));
    rows[2] = make_float4(rough, refl, sp.strength, sp.is_light ? 1.0f : 1.0f - refl);
    rows[3] = make_float4((float)s->shader_color[3 * sp.shader], (float)s->shader_color[3 * sp.shader + 1], (float)s->shader_color[3 * sp.shader + 2], 0.f);
}
static const char* not_finite(const t3d_scene_desc* s, bool triangles) {
    auto finite = [](const double* p, size_t n) {
        if (!p) return true;
        std::atomic<bool> all{true};
        parallel_slices(n, [&](size_t first, size_t last) {
            bool good = true;
            for (size_t i = first; i < last; ++i) good &= std::isfinite(p[i]);
            if (!good) all.store(false, std::memory_order_relaxed);
        });
        return all.load();
    };
    return triangles && !finite(s->tri_verts, (size_t)s->n_tris * 9) ? "a triangle corner"
         : !finite(s->obj_center, (size_t)s->n_objects * 3) ? "an object position"
         : !finite(s->obj_radius, (size_t)s->n_objects) ? "an object radius"
         : !finite(s->light_pos, (size_t)s->n_lights * 3) ? "a light position"
         : !finite(s->cam_pos, 3) || !finite(s->cam_euler, 3) || !finite(&s->cam_zlen, 1) ? "the camera"
         : !finite(s->obj_ior, (size_t)s->n_objects) || !finite(s->obj_strength, (size_t)s->n_objects) ? "an object's ior or strength"
         : !finite(s->shader_color, (size_t)s->n_shaders * 3) || !finite(s->shader_rough, (size_t)s->n_shaders) ||
           !finite(s->shader_refl, (size_t)s->n_shaders) || !finite(s->shader_emis, (size_t)s->n_shaders) ? "a shader value"
         : !finite(s->light_strength, (size_t)s->n_lights) || !finite(s->light_color, (size_t)s->n_lights * 3) ? "a light's strength or colour"
         : nullptr;
}

static int upload_scene(t3d_ctx* c, const t3d_scene_desc* s) {
    if (!c) return T3D_E_INVALID;
    if (!s || s->n_objects < 0 || s->n_tris < 0 || s->n_shaders < 0 || s->n_lights < 0) { c->err = "t3d_upload_scene: bad description"; return T3D_E_INVALID; }
    if (s->n_lights > T3D_MAX_LIGHTS) { c->err = "t3d_upload_scene: too many lights (limit 4096)"; return T3D_E_CAPACITY; }
    {
        // an array that the counts say has entries must exist
        const char* missing = s->n_objects > 0 && (!s->obj_type || !s->obj_tri_begin || !s->obj_tri_count || !s->obj_center || !s->obj_radius ||
                                                   !s->obj_ior || !s->obj_strength || !s->obj_shader0) ? "an object array"
                            : s->n_tris > 0 && (!s->tri_verts || !s->tri_shader) ? "a triangle array"
                            : s->n_shaders > 0 && (!s->shader_color || !s->shader_rough || !s->shader_refl || !s->shader_emis) ? "a shader array"
                            : s->n_lights > 0 && (!s->light_pos || !s->light_strength || !s->light_color) ? "a light array" : nullptr;
        if (missing) { c->err = std::string("t3d_upload_scene: ") + missing + " is null"; return T3D_E_INVALID; }
    }
    // geometry that is not a finite number has no place in a hierarchy or a plane group: refuse it by name
    // T3D_UPLOAD_TRACE: where an upload's time goes, phase by phase, on stderr
    const bool trace_phases = getenv("T3D_UPLOAD_TRACE") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (!trace_phases) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "t3d_upload_scene: %-28s %8.2f ms\n", name, std::chrono::duration<double, std::milli>(now - t_phase).count());
        t_phase = now;
    };
    if (const char* bad = not_finite(s, true)) { c->err = std::string("t3d_upload_scene: ") + bad + " is not finite"; return T3D_E_INVALID; }
    phase("arguments checked");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    UploadStream upload_stream(c);
    phase("previous scene released");

    // Triangles are grouped by supporting plane (same unit normal to 1e-7 and same offset to 1e-6, float64):
    // groups in order of their first triangle, triangles inside a group in their original order.
    struct Tri { int obj, face, shader; double n[3], nv0; };
    HostArray<Tri> tris((size_t)s->n_tris);   // every triangle belongs to exactly one object (checked below), which writes its entry
    std::vector<std::vector<int>> groups;
    std::map<std::tuple<long long, long long, long long, long long>, int> plane_index;
    std::vector<DevSphere> spheres;
    std::vector<float> obj_ior((size_t)s->n_objects);
    std::vector<char> seen((size_t)s->n_tris, 0);
    c->n_sphere_objs = c->n_light_objs = 0;
    // Large scenes are traced through an LBVH over all triangles and spheres, built on the device; the plane groups
    // of the brute-force loop are not built for them (every triangle stays where it is in the description).
    int n_prims = s->n_tris;
    for (int o = 0; o < s->n_objects; ++o) n_prims += s->obj_type[o] == T3D_OBJ_SPHERE || s->obj_type[o] == T3D_OBJ_LIGHT;
    const bool want_bvh = n_prims >= 2 && (c->accel == T3D_ACCEL_LBVH || (c->accel == T3D_ACCEL_AUTO && n_prims > T3D_BRUTE_FORCE_MAX_PRIMS));
    if (!want_bvh && s->n_tris >= (1 << 26)) { c->err = "t3d_upload_scene: too many triangles for the brute-force path"; return T3D_E_CAPACITY; }
    int n_owned = 0;
    for (int o = 0; o < s->n_objects; ++o) {
        obj_ior[o] = (float)s->obj_ior[o];
        int ty = s->obj_type[o];
        if (ty == T3D_OBJ_MESH) {
            int b0 = s->obj_tri_begin[o], n = s->obj_tri_count[o];
            if (b0 < 0 || n < 0 || b0 + n > s->n_tris) { c->err = "t3d_upload_scene: triangle range out of bounds"; return T3D_E_INVALID; }
            if (want_bvh) {
                // (the triangles of one object are distinct entries: slices of them can be filled side by side)
                std::atomic<int> bad{0};
                parallel_slices((size_t)n, [&](size_t first, size_t last) {
                    for (size_t k = first; k < last; ++k) {
                        const int t = b0 + (int)k;
                        const double* v = s->tri_verts + 9 * (size_t)t;
                        const double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
                        const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
                        Tri& T = tris[t];
                        T.n[0] = nx; T.n[1] = ny; T.n[2] = nz; T.nv0 = nx * v[0] + ny * v[1] + nz * v[2];
                        T.obj = o; T.face = (int)k; T.shader = s->tri_shader[t];
                        if (T.shader < 0 || T.shader >= s->n_shaders) bad.store(1, std::memory_order_relaxed);
                        if (seen[t]) bad.store(2, std::memory_order_relaxed);
                        seen[t] = 1;
                    }
                });
                if (bad.load() == 1) { c->err = "t3d_upload_scene: triangle shader index out of range"; return T3D_E_INVALID; }
                if (bad.load() == 2) { c->err = "t3d_upload_scene: triangle ranges of two objects overlap"; return T3D_E_INVALID; }
                n_owned += n;
                continue;
            }
            for (int k = 0; k < n; ++k) {
                int t = b0 + k;
                const double* v = s->tri_verts + 9 * (size_t)t;
                double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
                double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
                double nn = nx * nx + ny * ny + nz * nz;
                Tri& T = tris[t];
                T.n[0] = nx; T.n[1] = ny; T.n[2] = nz; T.nv0 = nx * v[0] + ny * v[1] + nz * v[2];
                T.obj = o; T.face = k; T.shader = s->tri_shader[t];
                if (T.shader < 0 || T.shader >= s->n_shaders) { c->err = "t3d_upload_scene: triangle shader index out of range"; return T3D_E_INVALID; }
                if (seen[t]) { c->err = "t3d_upload_scene: triangle ranges of two objects overlap"; return T3D_E_INVALID; }
                seen[t] = 1;
                ++n_owned;
                int g = -1;
                if (nn > 0) {
                    double len = sqrt(nn), off = T.nv0 / len;
                    auto key = std::make_tuple(llround(nx / len * 1e7), llround(ny / len * 1e7), llround(nz / len * 1e7),
                                               llround(off * 1e6));
                    auto it = plane_index.find(key);
                    if (it == plane_index.end()) { g = (int)groups.size(); plane_index.emplace(key, g); groups.emplace_back(); }
                    else g = it->second;
                } else { g = (int)groups.size(); groups.emplace_back(); }  // degenerate triangle: never hit, own group
                groups[g].push_back(t);
            }
        } else if (ty == T3D_OBJ_SPHERE || ty == T3D_OBJ_LIGHT) {
            DevSphere sp;
            sp.cx = (float)s->obj_center[3 * o]; sp.cy = (float)s->obj_center[3 * o + 1]; sp.cz = (float)s->obj_center[3 * o + 2];
            sp.r2 = (float)(s->obj_radius[o] * s->obj_radius[o]);
            sp.obj = o; sp.is_light = ty == T3D_OBJ_LIGHT; sp.shader = s->obj_shader0[o];
            sp.strength = (float)s->obj_strength[o];
            if (sp.shader < 0 || sp.shader >= s->n_shaders) { c->err = "t3d_upload_scene: sphere/light without shader"; return T3D_E_INVALID; }
            spheres.push_back(sp);
            if (sp.is_light) c->n_light_objs++; else c->n_sphere_objs++;
        }
    }
    if (n_owned != s->n_tris) { c->err = "t3d_upload_scene: triangles not owned by any mesh object"; return T3D_E_INVALID; }
    phase("normals, plane groups");
    // Device order of the groups: by the axis of the largest normal component (that axis is dropped for the
    // in-plane coordinates), groups of one axis in order of their first triangle; groups of more than 24
    // triangles are cut into pieces (the hit flags of a group are the bits of a float32 integer).
    struct Piece { int group, first, count, axis; };
    std::vector<Piece> pieces;
    int axis_count[3] = {0, 0, 0};
    for (int ax = 0; ax < 3; ++ax) {
        for (size_t g = 0; g < groups.size(); ++g) {
            const Tri& F = tris[groups[g][0]];
            int gax = 0;
            if (fabs(F.n[1]) > fabs(F.n[gax])) gax = 1;
            if (fabs(F.n[2]) > fabs(F.n[gax])) gax = 2;
            if (gax != ax) continue;
            for (int first = 0; first < (int)groups[g].size(); first += 24) {
                int cnt = (int)groups[g].size() - first;
                pieces.push_back(Piece{(int)g, first, cnt > 24 ? 24 : cnt, ax});
                axis_count[ax]++;
            }
        }
    }
    // Within an axis the pieces are visited in the order of how often they hold the closest hit, most often first: a plane
    // that cannot beat the best distance found so far is skipped by the whole wave, so the sooner the near surfaces are
    // tested, the more of the others are skipped.  The frequencies come from a few thousand rays between random points
    // of the triangles (float64, deterministic).  Planes that coincide -- the pieces of one group, two faces back to back
    // -- keep their original order, which decides who wins a tie in distance (T3DR.py:715).
    if (!want_bvh && pieces.size() > 2 && !getenv("T3D_KEEP_GROUP_ORDER")) {
        const size_t P = pieces.size();
        std::vector<int> piece_of_tri((size_t)s->n_tris, -1);
        for (size_t q = 0; q < P; ++q)
            for (int j = 0; j < pieces[q].count; ++j) piece_of_tri[groups[pieces[q].group][pieces[q].first + j]] = (int)q;
        std::vector<double> weight(P, 0.0);
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        auto uni = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng >> 11) * (1.0 / 9007199254740992.0); };
        auto point_on = [&](int t, double* p) {
            const double* v = s->tri_verts + 9 * (size_t)t;
            double a = uni(), b = uni();
            if (a + b > 1) { a = 1 - a; b = 1 - b; }
            for (int k = 0; k < 3; ++k) p[k] = v[k] + a * (v[3 + k] - v[k]) + b * (v[6 + k] - v[k]);
        };
        const int n_samples = 4096;
        for (int it = 0; it < n_samples && s->n_tris > 1; ++it) {
            const int ta = (int)(uni() * s->n_tris) % s->n_tris, tb = (int)(uni() * s->n_tris) % s->n_tris;
            double o[3], e[3], d[3];
            point_on(ta, o); point_on(tb, e);
            double len = 0;
            for (int k = 0; k < 3; ++k) { d[k] = e[k] - o[k]; len += d[k] * d[k]; }
            if (!(len > 1e-20)) continue;
            len = sqrt(len);
            for (int k = 0; k < 3; ++k) { d[k] /= len; o[k] += 1e-3 * len * d[k]; }
            double best = INFINITY; int hit = -1;
            for (int t = 0; t < s->n_tris; ++t) {
                const Tri& T = tris[t];
                const double dn = T.n[0] * d[0] + T.n[1] * d[1] + T.n[2] * d[2];
                if (fabs(dn) < 1e-12) continue;
                const double tt = (T.nv0 - (T.n[0] * o[0] + T.n[1] * o[1] + T.n[2] * o[2])) / dn;
                if (!(tt > 0) || tt >= best) continue;
                const double* v = s->tri_verts + 9 * (size_t)t;
                double pnt[3], c0[3], c1[3], c2[3];
                for (int k = 0; k < 3; ++k) pnt[k] = o[k] + tt * d[k];
                auto edge = [&](const double* a, const double* b, double* out) {   // (b - a) x (P - a)
                    const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {pnt[0] - a[0], pnt[1] - a[1], pnt[2] - a[2]};
                    out[0] = u[1] * w[2] - u[2] * w[1]; out[1] = u[2] * w[0] - u[0] * w[2]; out[2] = u[0] * w[1] - u[1] * w[0];
                };
                edge(v, v + 3, c0); edge(v + 3, v + 6, c1); edge(v + 6, v, c2);
                if (c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2] > 0 && c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2] > 0) { best = tt; hit = t; }
            }
            if (hit >= 0 && piece_of_tri[hit] >= 0) weight[piece_of_tri[hit]] += 1.0;
        }
        // classes of coinciding planes (union-find over the pieces)
        std::vector<int> cls(P);
        for (size_t q = 0; q < P; ++q) cls[q] = (int)q;
        std::function<int(int)> find = [&](int x) { while (cls[x] != x) x = cls[x] = cls[cls[x]]; return x; };
        auto unit_plane = [&](size_t q, double* n, double& off) {
            const Tri& F = tris[groups[pieces[q].group][0]];
            const double len = sqrt(F.n[0] * F.n[0] + F.n[1] * F.n[1] + F.n[2] * F.n[2]);
            if (!(len > 0)) return false;
            for (int k = 0; k < 3; ++k) n[k] = F.n[k] / len;
            off = F.nv0 / len;
            return true;
        };
        for (size_t a = 0; a < P; ++a) {
            double na[3], oa;
            if (!unit_plane(a, na, oa)) continue;
            for (size_t b = a + 1; b < P; ++b) {
                if (pieces[a].axis != pieces[b].axis) continue;
                double nb[3], ob;
                if (!unit_plane(b, nb, ob)) continue;
                const double dt = na[0] * nb[0] + na[1] * nb[1] + na[2] * nb[2];
                const double scale = fmax(1.0, fmax(fabs(oa), fabs(ob)));
                if (pieces[a].group == pieces[b].group || (fabs(dt) > 1 - 1e-9 && fabs(oa - (dt > 0 ? ob : -ob)) < 1e-5 * scale)) cls[find((int)b)] = find((int)a);
            }
        }
        std::vector<double> cls_weight(P, 0.0);
        std::vector<int> cls_first(P, (int)P);
        for (size_t q = 0; q < P; ++q) { const int r = find((int)q); cls_weight[r] += weight[q]; cls_first[r] = std::min(cls_first[r], (int)q); }
        std::vector<int> idx(P);
        for (size_t q = 0; q < P; ++q) idx[q] = (int)q;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
            if (pieces[a].axis != pieces[b].axis) return pieces[a].axis < pieces[b].axis;
            const int ra = find(a), rb = find(b);
            if (ra == rb) return a < b;
            if (cls_weight[ra] != cls_weight[rb]) return cls_weight[ra] > cls_weight[rb];
            return cls_first[ra] < cls_first[rb];
        });
        std::vector<Piece> sorted(P);
        for (size_t q = 0; q < P; ++q) sorted[q] = pieces[idx[q]];
        pieces.swap(sorted);
    }
    const size_t G = pieces.size();
    std::vector<float4> planes(6 * (G + 1), make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<int> plane_first(G + 1, 0);
    std::vector<float4> tri_uv((size_t)2 * ((want_bvh ? 0 : s->n_tris) + 1), make_float4(0.f, 0.f, 0.f, 0.f));   // in-plane rows: the brute-force loop's
    HostArray<float4> tri_n((size_t)s->n_tris + 1);
    HostArray<int4> meta((size_t)s->n_tris + 1);
    HostArray<int> order((size_t)s->n_tris + 1);  // device position -> index in the description
    tri_n[(size_t)s->n_tris] = make_float4(0.f, 0.f, 0.f, 0.f); meta[(size_t)s->n_tris] = make_int4(0, 0, 0, 0); order[(size_t)s->n_tris] = 0;   // (one spare entry each)
    {
        int k = 0;
        for (size_t g = 0; g < G; ++g) {
            const Piece& pc = pieces[g];
            const Tri& F = tris[groups[pc.group][0]];
            // rows are stored (x, offset, y, z): the first fma of a dot product then takes its factor and its
            // addend from one aligned SGPR pair; "+ 0.0f" turns a -0 offset into +0
            planes[6 * g] = make_float4((float)F.n[0], -(float)F.nv0 + 0.0f, (float)F.n[1], (float)F.n[2]);
            planes[6 * g + 1] = make_float4(__builtin_bit_cast(float, k), __builtin_bit_cast(float, k + pc.count), 0.f,
                                            __builtin_bit_cast(float, k | (pc.count << 26)));
            plane_first[g] = k;
            const int i1 = (pc.axis + 1) % 3, i2 = (pc.axis + 2) % 3;  // in-plane coordinates (s, t)
            for (int j = 0; j < pc.count; ++j) {
                const int t = groups[pc.group][pc.first + j];
                const Tri& T = tris[t];
                const double* v = s->tri_verts + 9 * (size_t)t;
                const double e1s = v[3 + i1] - v[i1], e1t = v[3 + i2] - v[i2], e2s = v[6 + i1] - v[i1], e2t = v[6 + i2] - v[i2];
                const double det = e1s * e2t - e2s * e1t;
                if (det != 0.0) {
                    // (Ps - v0s, Pt - v0t) = u (e1s, e1t) + v (e2s, e2t), solved in float64
                    const double as = e2t / det, at = -e2s / det, bs = -e1t / det, bt = e1s / det;
                    const double aw = -(as * v[i1] + at * v[i2]), bw = -(bs * v[i1] + bt * v[i2]);
                    tri_uv[2 * (size_t)k + 0] = make_float4((float)as, (float)aw + 0.0f, (float)at, 0.f);
                    tri_uv[2 * (size_t)k + 1] = make_float4((float)bs, (float)bw + 0.0f, (float)bt, 0.f);
                }   // else: degenerate triangle, rows stay zero and min3(u, v, w) = 0 never counts as inside
                if (j < 2) { planes[6 * g + 2 + 2 * j] = tri_uv[2 * (size_t)k + 0]; planes[6 * g + 3 + 2 * j] = tri_uv[2 * (size_t)k + 1]; }
                tri_n[k] = make_float4((float)T.n[0], (float)T.n[1], (float)T.n[2], -(float)T.nv0 + 0.0f);
                order[k] = t;
                meta[k] = make_int4(T.obj, T.face, T.shader, t);
                ++k;
            }
        }
        plane_first[G] = k;
        if (want_bvh) {
            parallel_slices((size_t)s->n_tris, [&](size_t first, size_t last) {
                for (size_t t = first; t < last; ++t) {
                    const Tri& T = tris[t];
                    tri_n[t] = make_float4((float)T.n[0], (float)T.n[1], (float)T.n[2], -(float)T.nv0 + 0.0f);
                    order[t] = (int)t;
                    meta[t] = make_int4(T.obj, T.face, T.shader, (int)t);
                }
            });
        }
        // Parallelograms: when the first two triangles of a group share an edge and their other two corners are
        // opposite corners of a parallelogram, the second triangle's barycentrics are the mirror image of the
        // first one's (u' = 1 - u, v' = 1 - v, w' = -w, origin at the corner opposite the shared edge), and the
        // kernel derives both interior tests from one (u, v).  The record then holds the first triangle's rows
        // with that origin and a flag.
        for (size_t g = 0; g < G; ++g) {
            const Piece& pc = pieces[g];
            if (pc.count < 2) continue;
            const double* A = s->tri_verts + 9 * (size_t)groups[pc.group][pc.first];
            const double* B = s->tri_verts + 9 * (size_t)groups[pc.group][pc.first + 1];
            int shared_a = 0, shared_b = 0, n_shared = 0;   // bit masks of the corners that coincide exactly
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    if (A[3 * i] == B[3 * j] && A[3 * i + 1] == B[3 * j + 1] && A[3 * i + 2] == B[3 * j + 2]) { shared_a |= 1 << i; shared_b |= 1 << j; ++n_shared; }
            if (n_shared != 2 || __builtin_popcount(shared_a) != 2 || __builtin_popcount(shared_b) != 2) continue;
            const int ia = shared_a == 6 ? 0 : (shared_a == 5 ? 1 : 2), ib = shared_b == 6 ? 0 : (shared_b == 5 ? 1 : 2);  // the lone corners
            const int i1a = (ia + 1) % 3, i2a = (ia + 2) % 3;
            double scale = 0, err = 0;
            for (int c3 = 0; c3 < 3; ++c3) {
                const double sum_diag = A[3 * i1a + c3] + A[3 * i2a + c3], sum_lone = A[3 * ia + c3] + B[3 * ib + c3];
                err = fmax(err, fabs(sum_diag - sum_lone));
                scale = fmax(scale, fmax(fabs(sum_diag), fabs(A[3 * i1a + c3] - A[3 * i2a + c3])));
            }
            if (!(err <= 1e-10 * scale)) continue;
            const int i1 = (pc.axis + 1) % 3, i2 = (pc.axis + 2) % 3;
            const double* v0 = A + 3 * ia; const double* v1 = A + 3 * i1a; const double* v2 = A + 3 * i2a;
            const double e1s = v1[i1] - v0[i1], e1t = v1[i2] - v0[i2], e2s = v2[i1] - v0[i1], e2t = v2[i2] - v0[i2];
            const double det = e1s * e2t - e2s * e1t;
            if (det == 0.0) continue;
            const double as = e2t / det, at = -e2s / det, bs = -e1t / det, bt = e1s / det;
            const double aw = -(as * v0[i1] + at * v0[i2]), bw = -(bs * v0[i1] + bt * v0[i2]);
            planes[6 * g + 2] = make_float4((float)as, (float)aw + 0.0f, (float)at, 0.f);
            planes[6 * g + 3] = make_float4((float)bs, (float)bw + 0.0f, (float)bt, 0.f);
            planes[6 * g + 1].z = __builtin_bit_cast(float, 1);
        }
    }
    phase("group and triangle rows");
    std::vector<DevShader> shaders((size_t)s->n_shaders);
    for (int k = 0; k < s->n_shaders; ++k) {
        shaders[k] = DevShader{(float)s->shader_color[3 * k], (float)s->shader_color[3 * k + 1], (float)s->shader_color[3 * k + 2],
                               (float)s->shader_rough[k], (float)s->shader_refl[k], (float)s->shader_emis[k], 0.f, 0.f};
    }
    std::vector<DevLight> lights = make_lights(c->sc, s);
    HostArray<float4> prim_info(4 * std::max<size_t>(1, (size_t)s->n_tris + spheres.size()));   // never empty: a miss reads record 0
    if ((size_t)s->n_tris + spheres.size() == 0) for (int q = 0; q < 4; ++q) prim_info[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        auto ib = [](int v) { return __builtin_bit_cast(float, v); };
        parallel_slices((size_t)s->n_tris, [&](size_t first, size_t last) {
        for (size_t k = first; k < last; ++k) {
            const int4 m = meta[k];
            const DevShader& sh = shaders[m.z];
            const bool emitter = sh.emis > 0.0f;
            {   // unit normal, normalised in float64 (a degenerate triangle is never hit: any finite value will do)
                const Tri& T = tris[order[k]];
                const double len = sqrt(T.n[0] * T.n[0] + T.n[1] * T.n[1] + T.n[2] * T.n[2]);
                const double inv = len > 0 ? 1.0 / len : 0.0;
                prim_info[4 * (size_t)k + 0] = make_float4((float)(T.n[0] * inv), (float)(T.n[1] * inv), (float)(T.n[2] * inv), obj_ior[m.x]);
            }
            prim_info[4 * (size_t)k + 1] = make_float4(ib(m.x), ib(m.y), ib(m.z), ib(emitter ? 1 : 0));
            prim_info[4 * (size_t)k + 2] = make_float4(sh.rough, sh.refl, sh.emis, emitter ? 1.0f : 1.0f - sh.refl);
            prim_info[4 * (size_t)k + 3] = make_float4(sh.r, sh.g, sh.b, 0.f);
        }
        });
        for (size_t q = 0; q < spheres.size(); ++q) sphere_rows(spheres[q], s, &prim_info[4 * ((size_t)s->n_tris + q)]);
    }
    phase("primitive records");
    int rc;
    if ((rc = upload(c, &c->d_prim_info, prim_info)) || (rc = upload(c, &c->d_planes, planes)) || (rc = upload(c, &c->d_plane_first, plane_first)) || (rc = upload(c, &c->d_tri_uv, tri_uv)) ||
        (rc = upload(c, &c->d_tri_n, tri_n)) || (rc = upload(c, &c->d_tri_meta, meta)) || (rc = upload(c, &c->d_spheres, spheres)) ||
        (rc = upload(c, &c->d_lights, lights)) || (rc = upload(c, &c->d_shaders, shaders)) || (rc = upload(c, &c->d_obj_ior, obj_ior)))
        return rc;
    phase("copies to the device");
    if (want_bvh) {
        // 3-D barycentric rows  a = (e2 x n) / |n|^2, b = (n x e1) / |n|^2 :  P - v0 = u e1 + v e2  (float64, rounded once)
        HostArray<float4> tri_ab((size_t)4 * (s->n_tris + 1));
        HostArray<float> verts((size_t)9 * (s->n_tris + 1));
        for (int q = 0; q < 4; ++q) tri_ab[4 * (size_t)s->n_tris + q] = make_float4(0.f, 0.f, 0.f, 0.f);   // (one spare entry each)
        for (int q = 0; q < 9; ++q) verts[9 * (size_t)s->n_tris + q] = 0.f;
        parallel_slices((size_t)s->n_tris, [&](size_t first, size_t last) {
        for (size_t k = first; k < last; ++k) {
            const double* v = s->tri_verts + 9 * (size_t)order[k];
            for (int q = 0; q < 9; ++q) verts[9 * (size_t)k + q] = (float)v[q];
            const double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
            const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
            const double nn = nx * nx + ny * ny + nz * nz;
            tri_ab[4 * (size_t)k + 0] = tri_n[k];
            tri_ab[4 * (size_t)k + 3] = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, order[k]));
            tri_ab[4 * (size_t)k + 1] = tri_ab[4 * (size_t)k + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(nn > 0)) continue;  // degenerate: rows stay zero, min3(u, v, w) = 0 is never inside
            const double a[3] = {(e2[1] * nz - e2[2] * ny) / nn, (e2[2] * nx - e2[0] * nz) / nn, (e2[0] * ny - e2[1] * nx) / nn};
            const double b[3] = {(ny * e1[2] - nz * e1[1]) / nn, (nz * e1[0] - nx * e1[2]) / nn, (nx * e1[1] - ny * e1[0]) / nn};
            tri_ab[4 * (size_t)k + 1] = make_float4((float)a[0], (float)a[1], (float)a[2], -(float)(a[0] * v[0] + a[1] * v[1] + a[2] * v[2]) + 0.0f);
            tri_ab[4 * (size_t)k + 2] = make_float4((float)b[0], (float)b[1], (float)b[2], -(float)(b[0] * v[0] + b[1] * v[1] + b[2] * v[2]) + 0.0f);
        }
        });
        phase("leaf rows");
        void* d_verts = nullptr;
        void* d_tri_ab = nullptr;   // records in device order; the build copies them into sorted leaf order
        if ((rc = upload(c, &d_tri_ab, tri_ab)) || (rc = upload(c, &d_verts, verts))) { hipFree(d_verts); hipFree(d_tri_ab); return rc; }
        phase("their copies");
        float4* nodes = nullptr;
        float4* leaves = nullptr;
        // Few primitives: the hierarchy is built here, top-down by surface area (t3d_host_tree.h, t3d_set_host_tree_limit), at about 0.3 us a primitive;
        // many: in Morton order on the device, which takes a tenth of that and gives a tree some 5 % (uniform soup) to 15 %
        // (a mesh in a room) slower to traverse.  T3D_HOST_TREE_MAX moves the line (0: always on the device): worth raising
        // for a scene that is rendered many times.
        HostTree host_tree;
        const char* env_max = getenv("T3D_HOST_TREE_MAX");
        const long host_max = env_max ? atol(env_max) : c->host_tree_max;
        bool host_built = n_prims >= 2 && n_prims <= host_max;
        auto prim_boxes = [&](std::vector<float>& blo, std::vector<float>& bhi) {   // boxes of the primitives in device order
            blo.resize(3 * (size_t)n_prims); bhi.resize(3 * (size_t)n_prims);
            parallel_slices((size_t)s->n_tris, [&](size_t first, size_t last) {
                for (size_t k = first; k < last; ++k)
                    for (int a = 0; a < 3; ++a) {
                        const float* v = &verts[9 * k];
                        blo[3 * k + a] = fminf(fminf(v[a], v[3 + a]), v[6 + a]);
                        bhi[3 * k + a] = fmaxf(fmaxf(v[a], v[3 + a]), v[6 + a]);
                    }
            });
            for (size_t q = 0; q < spheres.size(); ++q) {
                const DevSphere& sp = spheres[q];
                const float r = sqrtf(sp.r2), ce[3] = {sp.cx, sp.cy, sp.cz};
                for (int a = 0; a < 3; ++a) { blo[3 * (s->n_tris + q) + a] = ce[a] - r; bhi[3 * (s->n_tris + q) + a] = ce[a] + r; }
            }
        };
        if (host_built) {
            std::vector<float> blo, bhi;
            prim_boxes(blo, bhi);
            host_tree = host_tree_build(blo.data(), bhi.data(), n_prims);
            // a surface-area tree has no bound on its depth (sizes in geometric progression give a chain); the traversal
            // stack is sized for the Morton tree's (30 key bits + ties): deeper than that, take the Morton tree
            host_built = host_tree.depth <= T3D_BVH_STACK / 2;
        }
        hipError_t e = lbvh_build(c->stream, (const float*)d_verts, (const float4*)d_tri_ab, s->n_tris, (const DevSphere*)c->d_spheres,
                                  (int)spheres.size(), &nodes, &leaves, &c->lbvh_work, host_built ? &host_tree : nullptr);
        if (e != hipSuccess) { hipFree(d_verts); hipFree(d_tri_ab); c->err = std::string("t3d_upload_scene: LBVH build failed: ") + hipGetErrorString(e); return T3D_E_DEVICE; }
        phase("hierarchy");
        c->d_lbvh_verts = d_verts; c->d_lbvh_tri_ab = d_tri_ab;   // kept for a refit (t3d_update_scene)
        c->d_bvh_nodes = nodes; c->d_bvh_leaves = leaves;
        // deep ends of traversal stacks, one piece per wave the traversal kernels can have in flight (never touched by the
        // benchmark workloads, whose stacks stay within the LDS part)
        HIPCHK(c, hipMalloc(&c->d_bvh_spill, sizeof(int) * 64 * T3D_BVH_DEEP_STACK * (size_t)c->n_sm * T3D_BVH_WAVES * T3D_WAVES));
        c->use_bvh = true;
        c->tree_source = host_built ? T3D_TREE_HOST : T3D_TREE_DEVICE;
        if (!host_built && c->background_tree && n_prims >= 2) {
            // the surface-area tree of the same primitives, cut on a thread of its own while the caller renders through the
            // device's tree; adopt_background_tree puts it in place.  No thread to be had: the device's tree stays.
            std::unique_ptr<BackgroundTree> bg(new BackgroundTree());
            bg->n = n_prims;
            prim_boxes(bg->lo, bg->hi);
            BackgroundTree* b = bg.get();
            try {
                bg->th = std::thread([b] {
                    try { b->tree = host_tree_build(b->lo.data(), b->hi.data(), b->n); b->state.store(1, std::memory_order_release); }
                    catch (...) { b->state.store(2, std::memory_order_release); }
                });
                c->bg_tree = std::move(bg);
                c->tree_source = T3D_TREE_BUILDING;
            } catch (const std::system_error&) {}
        }
    }
    DevScene& d = c->sc;
    d.bvh.nodes = (const float4*)c->d_bvh_nodes; d.bvh.leaves = (const float4*)c->d_bvh_leaves; d.bvh.n_prims = n_prims;
    d.bvh.spill = (int*)c->d_bvh_spill;
    d.geo.axis_end[0] = axis_count[0]; d.geo.axis_end[1] = axis_count[0] + axis_count[1]; d.geo.axis_end[2] = (int)G;
    d.geo.n_tris = s->n_tris; d.geo.n_spheres = (int)spheres.size();
    for (int a = 0; a < 3; ++a) { c->tri_lo[a] = INFINITY; c->tri_hi[a] = -INFINITY; }
    {
        std::mutex box_mutex;
        parallel_slices((size_t)s->n_tris * 3, [&](size_t first, size_t last) {   // corners first .. last - 1
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (size_t p = first; p < last; ++p)
                for (int a = 0; a < 3; ++a) { const double x = s->tri_verts[3 * p + a]; lo[a] = fmin(lo[a], x); hi[a] = fmax(hi[a], x); }
            std::lock_guard<std::mutex> lk(box_mutex);
            for (int a = 0; a < 3; ++a) { c->tri_lo[a] = fmin(c->tri_lo[a], lo[a]); c->tri_hi[a] = fmax(c->tri_hi[a], hi[a]); }
        });
    }
    set_scene_box(c, spheres);
    d.n_planes = (int)G; d.n_tris = s->n_tris; d.n_spheres = (int)spheres.size(); d.n_lights = s->n_lights; d.n_shaders = s->n_shaders;
    d.tri_n = (const float4*)c->d_tri_n; d.tri_meta = (const int4*)c->d_tri_meta;
    d.lights = (const DevLight*)c->d_lights; d.shaders = (const DevShader*)c->d_shaders; d.obj_ior = (const float*)c->d_obj_ior;
    d.prim_info = (const float4*)c->d_prim_info;
    set_camera(d, s);
    c->n_objects = s->n_objects;
    c->obj_types.assign(s->obj_type, s->obj_type + s->n_objects);
    c->h_spheres = spheres;
    c->up_tris = s->n_tris; c->up_shaders = s->n_shaders; c->up_lights = s->n_lights;
    c->has_scene = true;
    c->scene_gen++;
    phase("bounds, bookkeeping");
    return T3D_OK;
}

// The host side of an upload works in containers (and, for large host-built hierarchies, threads) of the C++ library:
// what they throw ends here, as an error code like any other.
extern "C" int t3d_upload_scene(t3d_ctx* c, const t3d_scene_desc* s) {
    try {
        return upload_scene(c, s);
    } catch (const std::bad_alloc&) {
        c->err = "t3d_upload_scene: out of host memory"; c->has_scene = false;
        return T3D_E_CAPACITY;
    } catch (const std::exception& e) {
        c->err = std::string("t3d_upload_scene: ") + e.what(); c->has_scene = false;
        return T3D_E_DEVICE;
    }
}

// Puts the hierarchy that a host thread has cut since the upload (BackgroundTree) in place of the device's: the order of
// the leaves, the links and the parents are written over the build's working arrays, then the boxes are refitted and the
// records packed into the buffers the kernels already read -- no address changes, launch graphs stay valid.  Called at
// the start of every entry point that traces; does nothing (one atomic read) while the build runs, unless `wait`.
static int adopt_background_tree(t3d_ctx* c, bool wait = false) {
    if (!c->bg_tree) return T3D_OK;
    BackgroundTree& b = *c->bg_tree;
    if (!wait && b.state.load(std::memory_order_acquire) == 0) return T3D_OK;
    if (b.th.joinable()) b.th.join();
    std::unique_ptr<BackgroundTree> done = std::move(c->bg_tree);   // gone whatever happens below
    c->tree_source = T3D_TREE_DEVICE;
    const int n = c->up_tris + (int)c->h_spheres.size();
    if (done->state.load(std::memory_order_acquire) != 1 || !c->use_bvh || !c->has_scene || done->n != n ||
        done->tree.depth > T3D_BVH_STACK / 2)   // (deeper than the traversal stack provides for: see upload_scene)
        return T3D_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // nothing that reads the records may be in flight: the context's streams and the stream of the most recent call
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->shadow_stream));
    // (a caller's stream that has been destroyed since is an error of no consequence here)
    if (c->last_stream && c->last_stream != c->stream && hipStreamSynchronize(c->last_stream) != hipSuccess) (void)hipGetLastError();
    LbvhWork& w = c->lbvh_work;
    const HostTree& t = done->tree;
    // from the first copy on a failure leaves order, links and records that do not belong together: the scene is gone then
    hipError_t e = hipMemcpyAsync(w.vals[w.sorted], t.order.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(w.kids, t.kids.data(), sizeof(HostTreeKids) * (size_t)(n - 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(w.parent, t.parent.data(), sizeof(int) * (2 * (size_t)n - 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = lbvh_refit(c->stream, w, (const float*)c->d_lbvh_verts, (const float4*)c->d_lbvh_tri_ab, c->up_tris, (const DevSphere*)c->d_spheres,
                       (int)c->h_spheres.size(), (float4*)c->d_bvh_nodes, (float4*)c->d_bvh_leaves, true);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // the tree's vectors go with `done`; callers' streams start after this
    if (e != hipSuccess) { c->err = std::string("background tree: could not be put in place: ") + hipGetErrorString(e); c->has_scene = false; return T3D_E_DEVICE; }
    c->tree_source = T3D_TREE_BACKGROUND;
    return T3D_OK;
}

extern "C" int t3d_set_background_tree(t3d_ctx* c, int on) {
    if (!c) return T3D_E_INVALID;
    c->background_tree = on != 0;
    return T3D_OK;
}

extern "C" int t3d_tree_status(t3d_ctx* c, int wait) {
    if (!c) return T3D_E_INVALID;
    if (wait || (c->bg_tree && c->bg_tree->state.load(std::memory_order_acquire) != 0)) {
        const int rc = adopt_background_tree(c, true);
        if (rc != T3D_OK) return rc;
    }
    return c->has_scene && c->use_bvh ? c->tree_source : T3D_TREE_NONE;
}

// The uploaded scene again with spheres, Lights or the camera moved (centres, radii, strengths, light positions, the
// camera): nothing is grouped or built again, the hierarchy of a large scene is refitted.  The caller vouches for the
// rest of the description -- the triangles, the shaders, the objects' kinds -- being what was uploaded; what can be
// checked cheaply (the counts, the kinds) is.
extern "C" int t3d_update_scene(t3d_ctx* c, const t3d_scene_desc* s) {
    if (!c) return T3D_E_INVALID;
    if (!c->has_scene) { c->err = "t3d_update_scene: no scene uploaded"; return T3D_E_INVALID; }
    if (!s || s->n_objects != c->n_objects || s->n_tris != c->up_tris || s->n_shaders != c->up_shaders || s->n_lights != c->up_lights) {
        c->err = "t3d_update_scene: not the uploaded scene (counts differ): upload it";
        return T3D_E_INVALID;
    }
    if ((s->n_objects > 0 && (!s->obj_type || !s->obj_center || !s->obj_radius || !s->obj_ior || !s->obj_strength)) ||
        (s->n_lights > 0 && (!s->light_pos || !s->light_strength || !s->light_color)) ||
        (s->n_shaders > 0 && (!s->shader_rough || !s->shader_refl || !s->shader_color || !s->shader_emis))) {
        c->err = "t3d_update_scene: an array is null";
        return T3D_E_INVALID;
    }
    for (int o = 0; o < s->n_objects; ++o)
        if (s->obj_type[o] != c->obj_types[o]) { c->err = "t3d_update_scene: not the uploaded scene (an object's kind differs): upload it"; return T3D_E_INVALID; }
    if (const char* bad = not_finite(s, false)) { c->err = std::string("t3d_update_scene: ") + bad + " is not finite"; return T3D_E_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->shadow_stream));
    // (the stream is idle now: the staging vectors of the previous update are free again)
    std::vector<DevSphere>& spheres = c->h_spheres;
    std::vector<float4>& rows = c->up_rows;
    rows.resize(4 * spheres.size());
    for (size_t q = 0; q < spheres.size(); ++q) {
        DevSphere& sp = spheres[q];
        const int o = sp.obj;
        sp.cx = (float)s->obj_center[3 * o]; sp.cy = (float)s->obj_center[3 * o + 1]; sp.cz = (float)s->obj_center[3 * o + 2];
        sp.r2 = (float)(s->obj_radius[o] * s->obj_radius[o]);
        sp.strength = (float)s->obj_strength[o];
        sphere_rows(sp, s, &rows[4 * q]);
    }
    c->up_light_rows = make_lights(c->sc, s);
    const std::vector<DevLight>& lights = c->up_light_rows;
    if (!spheres.empty()) {
        HIPCHK(c, hipMemcpyAsync(c->d_spheres, spheres.data(), sizeof(DevSphere) * spheres.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync((float4*)c->d_prim_info + 4 * (size_t)c->up_tris, rows.data(), sizeof(float4) * rows.size(), hipMemcpyHostToDevice, c->stream));
    }
    if (!lights.empty()) HIPCHK(c, hipMemcpyAsync(c->d_lights, lights.data(), sizeof(DevLight) * lights.size(), hipMemcpyHostToDevice, c->stream));
    set_scene_box(c, spheres);
    set_camera(c->sc, s);
    if (c->use_bvh) {
        hipError_t e = lbvh_refit(c->stream, c->lbvh_work, (const float*)c->d_lbvh_verts, (const float4*)c->d_lbvh_tri_ab, c->up_tris,
                                  (const DevSphere*)c->d_spheres, (int)spheres.size(), (float4*)c->d_bvh_nodes, (float4*)c->d_bvh_leaves);
        if (e != hipSuccess) { c->err = std::string("t3d_update_scene: refit failed: ") + hipGetErrorString(e); return T3D_E_DEVICE; }
    }
    // not waited for: whatever the caller does next on this context is queued behind the copies and the refit
    c->scene_gen++;   // launch graphs hold the camera and the bounds by value
    return T3D_OK;
}

static int ensure_work(t3d_ctx* c, Work& w, size_t cap, int levels, size_t n0, bool dbg) {
    if (cap > w.cap || levels > w.levels || (dbg && !w.pool.dbg) || c->sc.n_lights > w.li_lights || (c->use_bvh && !w.hits)) {
        const size_t keep_n0 = w.n0;
        float4* keep_color = w.sample_color;
        w.sample_color = nullptr;
        free_work(w);
        w.sample_color = keep_color; w.n0 = keep_n0;
        size_t cap2 = cap, lv = (size_t)levels;
        HIPCHK(c, hipMalloc((void**)&w.rays[0], sizeof(RayRec) * cap2));
        HIPCHK(c, hipMalloc((void**)&w.rays[1], sizeof(RayRec) * cap2));
        HIPCHK(c, hipMalloc((void**)&w.pool.parent, sizeof(int) * cap2 * lv));
        HIPCHK(c, hipMalloc((void**)&w.pool.c, sizeof(float4) * cap2 * lv));
        HIPCHK(c, hipMalloc((void**)&w.pool.d, sizeof(float4) * cap2 * lv));
        if (dbg) HIPCHK(c, hipMalloc((void**)&w.pool.dbg, sizeof(int) * cap2 * lv));
        if (c->use_bvh) {
            HIPCHK(c, hipMalloc((void**)&w.hits, sizeof(uint2) * cap2));
            for (int k = 0; k < 2; ++k) {
                HIPCHK(c, hipMalloc((void**)&w.src_a[k], sizeof(float4) * cap2));
                HIPCHK(c, hipMalloc((void**)&w.src_b[k], sizeof(float4) * cap2));
            }
        }
        HIPCHK(c, hipMalloc((void**)&w.pool.li, sizeof(float) * cap2 * lv * (size_t)(c->sc.n_lights > 0 ? c->sc.n_lights : 1)));
        w.li_lights = c->sc.n_lights;
        w.cap = cap2; w.levels = levels;
    }
    if (n0 > w.n0) {
        hipFree(w.sample_color); w.sample_color = nullptr; w.n0 = 0;
        w.gen++;
        HIPCHK(c, hipMalloc((void**)&w.sample_color, sizeof(float4) * n0));
        w.n0 = n0;
    }
    return T3D_OK;
}

static int check_opts(t3d_ctx* c, const t3d_opts* o, int row0, int nrows) {
    if (!c) return T3D_E_INVALID;
    if (!c->has_scene) { c->err = "no scene uploaded"; return T3D_E_INVALID; }
    if (!o || o->width <= 0 || o->height <= 0 || o->bounces < 0 || o->samples < 1) { c->err = "bad options"; return T3D_E_INVALID; }
    if (o->bounces > T3D_MAX_BOUNCES) { c->err = "bounces above the supported maximum (14)"; return T3D_E_CAPACITY; }
    if (o->samples > 65535) { c->err = "samples above the supported maximum (65535)"; return T3D_E_CAPACITY; }
    if (!std::isfinite(o->ambient[0]) || !std::isfinite(o->ambient[1]) || !std::isfinite(o->ambient[2])) { c->err = "ambient colour is not finite"; return T3D_E_INVALID; }
    if (o->rng_mode != T3D_RNG_KEYED) { c->err = "the HIP library implements the keyed random mode only"; return T3D_E_INVALID; }
    if (o->frames < 0 || (o->frames > 1 && (long)o->frames * nrows > 0x3FFFFFFF)) { c->err = "bad number of frames per call"; return T3D_E_INVALID; }
    {
        long stride = o->row_stride > 1 ? o->row_stride : 1;
        if (row0 < 0 || nrows <= 0 || row0 + (long)(nrows - 1) * stride >= o->height) { c->err = "row range outside the image"; return T3D_E_INVALID; }
    }
    return T3D_OK;
}

// Counters and event timings of the most recent band -- or, with `whole_call`, the counters of every band of the most
// recent call as k_add_counters summed them (timings: the whole call's device time, the last band's per-kernel times);
// the stream must have been synchronised.
static void fill_stats(t3d_ctx* c, t3d_stats* stats, bool whole_call = false) {
    const int levels = c->last_levels;
    const int n_ev = c->last_timed ? levels + 2 : 2;
    float ms = 0;
    if (c->last_events) hipEventElapsedTime(&ms, whole_call ? c->ev_call0 : c->ev[0], c->ev[n_ev - 1]);
    stats->ms_device += ms;
    if (c->last_timed) {
        for (int d = 0; d < levels; ++d) {
            hipEventElapsedTime(&ms, c->ev[d], c->ev[d + 1]);
            stats->ms_k_level += ms;
            stats->level_ms[d] += ms;
        }
        hipEventElapsedTime(&ms, c->ev[levels], c->ev[n_ev - 1]);
        stats->ms_resolve += ms;
    }
    CallTotals sums{};
    if (whole_call) sums = *c->h_totals;
    else {
        for (int d = 0; d < levels; ++d) {
            if (d == 0) sums.rays[0] = c->last_n0;
            for (int sh = 0; sh < T3D_SHARDS; ++sh) {
                if (d > 0) sums.rays[d] += (uint64_t)c->h_cnt->n_rays[d][sh].v;
                sums.nodes[d] += (uint64_t)c->h_cnt->n_nodes[d][sh].v;
            }
        }
        for (int sh = 0; sh < T3D_SHARDS; ++sh) for (int k = 0; k < 8; ++k) sums.bvh[k] += c->h_cnt->bvh_stat[sh][k];
    }
    const uint64_t n0 = whole_call ? c->call_n0 : c->last_n0;
    uint64_t rays = 0, nodes = 0;
    for (int d = 0; d < levels; ++d) {
        rays += sums.rays[d]; nodes += sums.nodes[d];
        stats->level_rays[d] += sums.rays[d]; stats->level_nodes[d] += sums.nodes[d];
    }
    uint64_t shadow = nodes * (uint64_t)c->sc.n_lights;
    uint64_t q = rays + shadow;
    stats->rays_api += n0 + shadow;
    stats->queries += q;
    stats->trace_queries += rays;
    stats->shadow_queries += shadow;
    const uint64_t culled = sums.bvh[4];
    stats->shadow_culled += culled;
    if (c->use_bvh) {
        stats->bvh_deep_pushes += sums.bvh[3];
        stats->aabb_tests += 4 * sums.bvh[0];  // every record visited holds four boxes
        stats->tri_tests += sums.bvh[1];
        stats->sphere_tests += sums.bvh[2];
    } else {
        stats->tri_tests += (q - culled) * (uint64_t)c->sc.n_tris;
        stats->sphere_tests += (q - culled) * (uint64_t)c->sc.n_spheres - (c->last_lights ? 0 : n0 * (uint64_t)c->n_light_objs);
    }
    stats->nodes += nodes;  // shaded nodes only; emitter leaves are folded into their parents
#ifdef T3D_COUNT_GROUPS
    {
        unsigned long long g[8] = {0}, z[8] = {0};
        hipMemcpyFromSymbol(g, HIP_SYMBOL(g_group_count), sizeof(g));
        hipMemcpyToSymbol(HIP_SYMBOL(g_group_count), z, sizeof(z));
        fprintf(stderr, "group counts  trace: rounds %llu groups visited %llu entered %llu triangle steps %llu   shadow: rounds %llu visited %llu entered %llu steps %llu\n",
                g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
    }
#endif
    stats->launches += (whole_call ? (uint64_t)c->call_bands : 1) * c->last_launches;
    stats->n_k_level += (uint64_t)levels * (whole_call ? (uint64_t)c->call_bands : 1);
}

// Trace one band whose outputs are device pointers, with the work buffers `w` (grown as needed unless `fixed_work`:
// then `w` has been laid out by the caller and must already hold `cap_factor` entries per primary sample).  Returns
// T3D_E_CAPACITY if a level overflowed its queue (the caller retries with a larger capacity factor or fewer rows).
static int render_band(t3d_ctx* c, Work& w, const t3d_opts* o, int row0, int frame_rows, int vrow0, int nrows, int cap_factor, uint8_t* d_bgr, float* d_rad,
                       int32_t* d_prim, float* d_prim_t, hipStream_t st, t3d_stats* stats, const RenderParams* single,
                       t3d_node* d_dbg, int dbg_cap, bool fixed_work = false, size_t min_cap = 0) {
    const int S = o->bounces > 1 ? o->samples : 1;  // T3DR.py:1209
    const size_t n0 = single ? 1 : (size_t)nrows * o->width * S;
    const int levels = o->bounces + 1;
    const int shard_bits = single ? 0 : 6;   // the tree of a single ray lives in one shard, worked on by one workgroup
    static_assert((1 << 6) == T3D_SHARDS, "shard_bits");
    const size_t shards = (size_t)1 << shard_bits;
    size_t cap = n0 * (size_t)cap_factor;
    if (!single && cap < 64 * 1024) cap = 64 * 1024;   // at least 1024 entries per queue shard
    if (cap < min_cap && min_cap * (size_t)levels <= (size_t)0x7FFFFFF0) cap = min_cap;
    cap = (cap + shards * 64 - 1) / (shards * 64) * (shards * 64);
    if (cap * (size_t)levels > (size_t)0x7FFFFFF0) { c->err = "band too large for 32-bit node indices"; return T3D_E_CAPACITY; }
    if (fixed_work) {
        if (cap > w.cap || levels > w.levels || n0 > w.n0) { c->err = "render_band: fixed work buffers too small"; return T3D_E_CAPACITY; }
    } else {
        int rc = ensure_work(c, w, cap, levels, n0, single != nullptr);
        if (rc) return rc;
    }

    RenderParams rp{};
    rp.width = o->width; rp.height = o->height; rp.row0 = row0; rp.nrows = single ? 1 : nrows;
    rp.row_stride = o->row_stride > 1 ? o->row_stride : 1;
    rp.vrow0 = single ? 0 : vrow0; rp.frame_rows = !single && o->frames > 1 ? frame_rows : 0; rp.seed_step = o->frame_seed_step;
    rp.bounces = o->bounces; rp.samples = single ? 1 : S; rp.lights = o->lights; rp.seed = o->seed;
    for (int k = 0; k < 3; ++k) rp.ambient[k] = (float)o->ambient[k];
    rp.cap = (int)(fixed_work ? cap : w.cap);
    rp.n0 = (int)n0;
    rp.resident_waves = c->n_sm * 16;   // 4 waves per SIMD at this kernel's register count
    rp.tiled = !single && o->width % 16 == 0 && nrows % 8 == 0;
    rp.full_rounds = c->full_rounds && !single;
    rp.shard_bits = shard_bits;
    // few lights on the brute-force path: k_level lists the (node, light) pairs worth a shadow ray (PAIRS)
    const bool pairs = !c->use_bvh && c->sc.n_lights >= 1 && c->sc.n_lights <= T3D_PAIR_LIGHTS && c->sc.n_shaders < (1 << T3D_LIVE_SHIFT);
    rp.live_masks = pairs;
    rp.sticky = c->h_sticky;
    if (single) {
        rp.single = single->single; rp.px0 = single->px0; rp.py0 = single->py0; rp.sample0 = single->sample0; rp.want_debug = 1;
        for (int k = 0; k < 3; ++k) { rp.ray_o[k] = single->ray_o[k]; rp.ray_d[k] = single->ray_d[k]; }
        rp.ray_ior = single->ray_ior; rp.ray_obj = single->ray_obj;
    }
    const size_t cap_used = (size_t)rp.cap;

    static_assert(sizeof(Counters) % sizeof(uint4) == 0, "k_resolve clears the counters in 16-byte pieces");
    c->d_cnt = c->d_cnt_pair[c->cnt_idx];               // all zero: cleared by the band before (or at creation) ...
    if (!c->cnt_clean) HIPCHK(c, hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), st));   // ... unless that band was cut short
    Counters* next_cnt = c->d_cnt_pair[c->cnt_idx ^ 1];
    c->cnt_idx ^= 1;
    c->cnt_clean = false;
    if (c->use_bvh) HIPCHK(c, hipMemsetAsync(c->d_fetch, 0, sizeof(PaddedCount) * (size_t)levels * 2 * T3D_SHARDS, st));
    // one sample per pixel: the kernels that finish a sample write the pixel themselves, and k_finish is not launched
    const bool direct = !single && rp.samples == 1;
    TraceOut out{w.sample_color, d_prim, d_prim_t, d_dbg, dbg_cap, direct ? d_bgr : nullptr, direct ? d_rad : nullptr};
    const int max_blocks = c->n_sm * 8;
    auto blocks_for = [&](size_t n) { size_t b = (n + T3D_BLOCK - 1) / T3D_BLOCK; return (int)(b < 1 ? 1 : (b > (size_t)max_blocks ? (size_t)max_blocks : b)); };
    // grids of the sharded kernels hold a multiple of T3D_SHARDS waves
    const size_t grid_unit = T3D_SHARDS / (T3D_BLOCK / 64);
    auto trace_blocks_for = [&](size_t n) {
        // up to four times as many waves as the device holds at once: a wave then has one chunk of rays at most, and the
        // hardware hands out workgroups as slots fall free, which balances better than chunks dealt to resident waves
        // (measured: 8, 12, 16 workgroups of 256 per CU give k_level launches of 121, 119, 117 us; 4 give 124)
        size_t b = (n + T3D_BLOCK - 1) / T3D_BLOCK, mx = (size_t)c->n_sm * (4096 / T3D_BLOCK);
        b = b > mx ? mx : b;
        b = (b + grid_unit - 1) / grid_unit * grid_unit;
        return (int)b;
    };
    uint64_t launches = 0;
    // per-launch timing: event d+1 after k_level(d); read by fill_stats
    int n_timed = 0;
    auto tick = [&]() -> hipError_t { return hipEventRecord(c->ev[n_timed++], st); };
    const float4* planes = (const float4*)c->d_planes;
    const int* pfirst = (const int*)c->d_plane_first;
    const float4* uv = (const float4*)c->d_tri_uv;
    const DevSphere* sph = (const DevSphere*)c->d_spheres;
    // the traversal kernels of the LBVH path keep as many waves as the device holds; they take their rays in chunks from counters
    auto resident_blocks_for = [&](size_t n) {
        size_t b = (n + T3D_BLOCK - 1) / T3D_BLOCK, mx = (size_t)c->n_sm * T3D_BVH_WAVES;
        b = b > mx ? mx : b;
        b = (b + grid_unit - 1) / grid_unit * grid_unit;
        return (int)b;
    };
    // A caller that wants to be able to stop a render is asked between the levels: the stream is drained after every
    // level's launches, which costs the overlap of host and device and is only done while a cancel callback is installed.
    const bool cancellable = c->cancel_cb != nullptr && !single;
    bool cancelled = false;
    // per-level timing wants the levels one after the other
    const bool overlap = c->use_bvh && !c->timing && !single && !cancellable;
    // a call that stays asynchronous and is not being timed gets no events at all: each one is a stop in the stream
    const bool events = c->timing || stats || single;
    // ---- launch graph of a repeated band (brute-force path, nothing but kernels in the stream) ----
    char key[sizeof c->graphs[0].key];
    static_assert(sizeof(RenderParams) + 12 * sizeof(void*) <= sizeof key, "graph key");
    const bool graphable = c->graphs_enabled && !events && !single && !c->use_bvh && !cancellable && !fixed_work;
    bool capturing = false;
    std::unique_lock<std::mutex> capture_lock(g_capture_mutex, std::defer_lock);   // held while this thread records
    hipStream_t ls = st;   // where the brute-force path's kernels are queued: the call's stream, or the recording stream
    t3d_ctx::FrameGraph* slot = nullptr;
    // level 0 of the brute-force path: the launch a graph re-parameterises for another seed
    void* level0_fn = nullptr;
    const int level0_grid = single ? 0 : trace_blocks_for(n0) * T3D_WAVES;
    RayRec* level0_in = w.rays[0];
    RayRec* level0_out = w.rays[1];
    int level0_depth = 0;
    if (graphable) {
        memset(key, 0, sizeof key);
        RenderParams key_rp = rp;
        key_rp.seed = 0;
        memcpy(key, &key_rp, sizeof key_rp);
        const void* ptrs[12] = {d_bgr, d_rad, d_prim, d_prim_t, c->d_cnt, next_cnt, w.rays[0], w.rays[1], w.pool.c, w.pool.li,
                                w.sample_color, (const void*)(uintptr_t)((c->scene_gen << 32) ^ w.gen)};
        memcpy(key + sizeof rp, ptrs, sizeof ptrs);
        for (auto& g : c->graphs)
            if (g.valid && memcmp(g.key, key, sizeof key) == 0 && (g.seed == rp.seed || g.level0)) {
                if (g.seed != rp.seed) {
                    // another frame of the same band: the level-0 node gets this call's parameters (only the seed differs)
                    level0_fn = pairs ? (void*)k_level<true, 1, true> : (void*)k_level<true, 1, false>;
                    void* kargs[] = {(void*)&c->sc, (void*)&planes, (void*)&pfirst, (void*)&uv, (void*)&sph, (void*)&c->sc.lights, (void*)&rp,
                                     (void*)&level0_depth, (void*)&level0_in, (void*)&level0_out, (void*)&w.pool, (void*)&c->d_cnt, (void*)&out};
                    hipKernelNodeParams np{};
                    np.func = level0_fn; np.gridDim = dim3(level0_grid); np.blockDim = dim3(64); np.sharedMemBytes = 0;
                    np.kernelParams = kargs; np.extra = nullptr;
                    if (hipGraphExecKernelNodeSetParams(g.exec, g.level0, &np) != hipSuccess) {
                        (void)hipGetLastError();
                        g.level0 = nullptr;      // this runtime cannot do it: the graph keeps serving its own seed
                        break;
                    }
                    g.seed = rp.seed;
                    c->graphs_reseeded++;
                }
                HIPCHK(c, hipGraphLaunch(g.exec, st));
                c->graphs_replayed++;
                launches = g.launches;
                goto issued;
            }
        if (memcmp(c->graph_miss[0], key, sizeof key) == 0 || memcmp(c->graph_miss[1], key, sizeof key) == 0) {
            capture_lock.lock();
            if (hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) { capturing = true; ls = c->capture_stream; }
            else { (void)hipGetLastError(); c->graphs_enabled = false; capture_lock.unlock(); }
        } else {
            memcpy(c->graph_miss[c->graph_miss_next], key, sizeof key);
            c->graph_miss_next ^= 1;
        }
    }
    if (events) HIPCHK(c, tick());  // ev[0]: start of the call
    for (int d = 0; d < levels; ++d) {
        RayRec* in = w.rays[d & 1];
        RayRec* outq = w.rays[(d + 1) & 1];
        const int grid_n = single ? 1 : trace_blocks_for(d == 0 ? n0 : cap_used);
        if (c->use_bvh) {
            const int grid_r = single ? 1 : resident_blocks_for(d == 0 ? n0 : cap_used);
            PaddedCount* fetch_t = c->d_fetch + (size_t)d * 2 * T3D_SHARDS;
            float4* src_a = w.src_a[d & 1];
            float4* src_b = w.src_b[d & 1];
            // the classification of level d overwrites the shading inputs that the shadow rays of level d - 2 read
            if (overlap && d >= 2 && c->sc.n_lights > 0) HIPCHK(c, hipStreamWaitEvent(st, c->ev_shadowed[d - 2], 0));
            if (d == 0) {
                hipLaunchKernelGGL(k_bvh_trace<true>, dim3(grid_r), dim3(T3D_BLOCK), 0, st, c->sc, sph, rp, d, in, w.hits, c->d_cnt, fetch_t);
                hipLaunchKernelGGL(k_bvh_process<true>, dim3(grid_n), dim3(T3D_BLOCK), 0, st, c->sc, sph, rp, d, in, outq, w.hits, w.pool,
                                   c->d_cnt, out, src_a, src_b);
            } else {
                hipLaunchKernelGGL(k_bvh_trace<false>, dim3(grid_r), dim3(T3D_BLOCK), 0, st, c->sc, sph, rp, d, in, w.hits, c->d_cnt, fetch_t);
                hipLaunchKernelGGL(k_bvh_process<false>, dim3(grid_n), dim3(T3D_BLOCK), 0, st, c->sc, sph, rp, d, in, outq, w.hits, w.pool,
                                   c->d_cnt, out, src_a, src_b);
            }
            launches += 2;
            if (c->sc.n_lights > 0) {
                // The shadow rays of level d need its classification only, and the rays of level d + 1 do not need
                // them: on a stream of their own they fill the device while the next level's kernels drain and start.
                hipStream_t ss = overlap ? c->shadow_stream : st;
                if (overlap) {
                    HIPCHK(c, hipEventRecord(c->ev_traced[d], st));
                    HIPCHK(c, hipStreamWaitEvent(ss, c->ev_traced[d], 0));
                }
                const size_t pairs = (d == 0 ? n0 : cap_used) * (size_t)c->sc.n_lights;
                hipLaunchKernelGGL(k_bvh_shadow, dim3(single ? 1 : resident_blocks_for(pairs)), dim3(T3D_BLOCK), 0, ss, c->sc, sph, rp,
                                   d, src_a, src_b, w.pool, c->d_cnt, fetch_t + T3D_SHARDS);
                if (overlap) HIPCHK(c, hipEventRecord(c->ev_shadowed[d], ss));
                launches += 1;
            }
        } else {
            // waves of k_level never talk to each other: one wave per workgroup, so that a finished wave's slot is free at
            // once instead of when the slowest of four is done (1.5 % on the benchmark frame)
#define T3D_LAUNCH_LEVEL(L0, W, P, GRID, BLOCK)                                                                                   \
    hipLaunchKernelGGL((k_level<L0, W, P>), dim3(GRID), dim3(BLOCK), 0, ls, c->sc, planes, pfirst, uv, sph, c->sc.lights, rp, d, in, outq, w.pool, \
                       c->d_cnt, out)
            // Level 0 is launched with up to four times as many waves as the device holds at once (trace_blocks_for); a
            // deeper level with as many as it holds, every wave walking its shard's queue in strides: alone on the device
            // such a launch takes as long as the larger one, and when other frames are in flight on other streams their
            // kernels get the slots its finished waves leave (with three frames in flight 0.40 -> 0.37 ms per frame;
            // 75 % of the resident count is slower alone, 200 % and more lose the overlap).
            const int grid_0 = grid_n * T3D_WAVES;
            const int grid_w = std::min(grid_0, (rp.resident_waves + T3D_SHARDS - 1) / T3D_SHARDS * T3D_SHARDS);
            if (!single) {
                if (d == 0) { if (pairs) T3D_LAUNCH_LEVEL(true, 1, true, grid_0, 64); else T3D_LAUNCH_LEVEL(true, 1, false, grid_0, 64); }
                else        { if (pairs) T3D_LAUNCH_LEVEL(false, 1, true, grid_w, 64); else T3D_LAUNCH_LEVEL(false, 1, false, grid_w, 64); }
            } else {
                if (d == 0) { if (pairs) T3D_LAUNCH_LEVEL(true, T3D_WAVES, true, grid_n, T3D_BLOCK); else T3D_LAUNCH_LEVEL(true, T3D_WAVES, false, grid_n, T3D_BLOCK); }
                else        { if (pairs) T3D_LAUNCH_LEVEL(false, T3D_WAVES, true, grid_n, T3D_BLOCK); else T3D_LAUNCH_LEVEL(false, T3D_WAVES, false, grid_n, T3D_BLOCK); }
            }
#undef T3D_LAUNCH_LEVEL
            launches += 1;
        }
        if (c->timing) HIPCHK(c, tick());  // ev[1 + d]: level d traced
        if (cancellable) {
            HIPCHK(c, hipStreamSynchronize(st));
            if (c->cancel_cb(c->cancel_user)) { cancelled = true; break; }
        }
    }
    if (cancelled) {
        // the counters of this band and of the next one are in an unknown state: the next band clears its own at its start
        // (cnt_clean stays false) and, in its last kernel, the pair's other set
        c->last_valid = false;
        c->err = "render cancelled by the caller's cancel callback";
        return T3D_E_CANCELLED;
    }
    if (overlap && c->sc.n_lights > 0)   // the resolve reads every level's light intensities
        for (int d = levels >= 2 ? levels - 2 : 0; d < levels; ++d) HIPCHK(c, hipStreamWaitEvent(st, c->ev_shadowed[d], 0));
    for (int d = levels - 1; d >= 0; --d) {
        // eight workgroups per CU at most: a grid twice that size is no faster for these short, memory-bound launches
        int rgrid = trace_blocks_for(cap_used);
        { int mx = c->n_sm * 8; mx = (mx + (int)grid_unit - 1) / (int)grid_unit * (int)grid_unit; if (rgrid > mx) rgrid = mx; }
        hipLaunchKernelGGL(k_resolve, dim3(single ? 1 : rgrid), dim3(T3D_BLOCK), 0, ls, c->sc, rp, d, w.pool, c->d_cnt,
                           w.sample_color, d_dbg, dbg_cap, out.bgr, out.radiance, (uint4*)next_cnt, (int)(sizeof(Counters) / sizeof(uint4)));
        launches++;
    }
    if (!single && !direct) {
        hipLaunchKernelGGL(k_finish, dim3(blocks_for((size_t)nrows * o->width)), dim3(T3D_BLOCK), 0, ls, rp, w.sample_color, d_bgr, d_rad);
        launches++;
    }
    if (capturing) {
        // what was queued above was recorded, not run: make the graph, keep it, run it
        slot = &c->graphs[c->graph_next];
        c->graph_next = (c->graph_next + 1) % 4;
        if (slot->valid) { hipGraphExecDestroy(slot->exec); hipGraphDestroy(slot->graph); slot->valid = false; }
        hipError_t e = hipStreamEndCapture(c->capture_stream, &slot->graph);
        capture_lock.unlock();
        if (e == hipSuccess) {
            e = hipGraphInstantiate(&slot->exec, slot->graph, nullptr, nullptr, 0);
            if (e != hipSuccess) hipGraphDestroy(slot->graph);
        }
        if (e != hipSuccess) {
            c->graphs_enabled = false;   // this band is lost; the caller may simply ask again, kernels are launched one by one from now on
            c->err = std::string("launch graph: ") + hipGetErrorString(e);
            return T3D_E_DEVICE;
        }
        memcpy(slot->key, key, sizeof key);
        slot->launches = launches;
        slot->valid = true;
        slot->seed = rp.seed;
        slot->level0 = nullptr;
        {
            // the one launch that reads the seed is the first of the band: the graph's only root
            size_t n_root = 0;
            hipGraphNode_t root = nullptr;
            hipGraphNodeType ty;
            hipKernelNodeParams np{};
            const void* want = pairs ? (const void*)k_level<true, 1, true> : (const void*)k_level<true, 1, false>;
            if (hipGraphGetRootNodes(slot->graph, nullptr, &n_root) == hipSuccess && n_root == 1 &&
                hipGraphGetRootNodes(slot->graph, &root, &n_root) == hipSuccess && hipGraphNodeGetType(root, &ty) == hipSuccess &&
                ty == hipGraphNodeTypeKernel && hipGraphKernelNodeGetParams(root, &np) == hipSuccess && np.func == want &&
                np.gridDim.x == (unsigned)level0_grid)
                slot->level0 = root;
            (void)hipGetLastError();
        }
        c->graphs_captured++;
        HIPCHK(c, hipGraphLaunch(slot->exec, st));
    }
issued:
    c->cnt_clean = true;
    if (events) HIPCHK(c, tick());  // last event: end of the call
    HIPCHK(c, hipGetLastError());
    // a call that stays asynchronous leaves its counters on the device until somebody asks (t3d_last_call_stats)
    c->counters_on_host = stats || single;
    if (c->counters_on_host) HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st));
    c->last_timed = c->timing;
    c->last_events = events;
    c->last_levels = levels; c->last_n0 = n0; c->last_lights = o->lights; c->last_launches = launches; c->last_stream = st;
    c->last_valid = true;
    if (!stats && !single) { c->last_async_cap = cap_used; c->last_async_static = fixed_work; return T3D_OK; }  // asynchronous use: an overflow raises the context's sticky words, see t3d_ctx
    HIPCHK(c, hipStreamSynchronize(st));
    if (c->h_cnt->overflow) { c->h_sticky[0] = 0; return T3D_E_CAPACITY; }   // seen and dealt with by the caller: not left standing
    if (c->h_cnt->bvh_stack_overflow) { c->h_sticky[1] = 0; c->err = "LBVH traversal stack overflow"; return T3D_E_DEVICE; }
    if (stats) fill_stats(c, stats);
    return T3D_OK;
}

// Overflow left behind by an earlier asynchronous call on this context (see t3d_ctx::h_sticky): reported once, then cleared.
static int take_sticky(t3d_ctx* c) {
    if (c->h_sticky[1]) {
        c->h_sticky[0] = c->h_sticky[1] = 0;
        c->err = "an earlier asynchronous render call overflowed the LBVH traversal stack; its output is invalid";
        return T3D_E_DEVICE;
    }
    if (c->h_sticky[0]) {
        c->h_sticky[0] = 0;
        if (c->last_async_static) { if (c->static_async_factor < (1 << T3D_MAX_BOUNCES)) c->static_async_factor *= 2; }
        else c->async_min_cap = 2 * (c->last_async_cap > c->async_min_cap ? c->last_async_cap : c->async_min_cap);
        c->err = "an earlier asynchronous render call overflowed its node buffers (its output is invalid); asynchronous calls use larger buffers from now on";
        return T3D_E_CAPACITY;
    }
    return T3D_OK;
}

// A render call that is not waited for (no stats) and takes several bands keeps whole-call counters on the device:
// cleared here, every band's added by k_add_counters behind it, read by t3d_last_call_stats.
static int call_totals_begin(t3d_ctx* c, hipStream_t st, bool keep) {
    c->call_bands = 1; c->call_totals = false; c->call_n0 = 0;
    if (!keep) return T3D_OK;
    HIPCHK(c, hipMemsetAsync(c->d_totals, 0, sizeof(CallTotals), st));
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev_call0, st));
    return T3D_OK;
}
static int call_totals_add(t3d_ctx* c, hipStream_t st, int levels, size_t n0, int bands_done) {
    hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(64), 0, st, c->d_cnt, c->d_totals, levels, (unsigned long long)n0);
    HIPCHK(c, hipGetLastError());
    c->call_n0 += n0; c->call_bands = bands_done; c->call_totals = true;
    return T3D_OK;
}

extern "C" int t3d_render_rows_device(t3d_ctx* c, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_device,
                                      float* radiance_device, int32_t* primary_device, float* primary_t_device,
                                      void* stream, t3d_stats* stats) {
    int rc = check_opts(c, o, row0, nrows);
    if (rc) return rc;
    if (!bgr_device) { c->err = "bgr output is required"; return T3D_E_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = take_sticky(c)) != T3D_OK) return rc;
    if ((rc = adopt_background_tree(c)) != T3D_OK) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (stats) memset(stats, 0, sizeof *stats);
    const int S = o->bounces > 1 ? o->samples : 1;
    const int levels = o->bounces + 1;
    // rows per band: the optimistic working set (capacity factor 2 per level) must fit the budget
    const size_t per_sample = (size_t)2 * (2 * sizeof(RayRec) + (c->use_bvh ? sizeof(uint2) + 4 * sizeof(float4) : 0) +
                                           (size_t)levels * (36 + 4 * (size_t)(c->sc.n_lights > 0 ? c->sc.n_lights : 1))) + sizeof(float4);
    size_t max_samples = c->mem_budget / per_sample;
    size_t max_idx = (size_t)0x7FFFFFF0 / ((size_t)levels * 2);
    if (max_samples > max_idx) max_samples = max_idx;
    int band = (int)(max_samples / ((size_t)o->width * S));
    if (band < 1) band = 1;
    // the rows of the call, all its frames one after the other (bands may straddle frames)
    const int vrows = nrows * (o->frames > 1 ? o->frames : 1);
    if (band > vrows) band = vrows;
    // a level holds at most 2^bounces rays per sample; a band of few samples is worked on by few waves, whose
    // rays all land in those waves' shards, so the bound on the factor counts every shard
    const int factor_limit = (1 << o->bounces) * T3D_SHARDS;
    const bool keep_totals = !stats && band < vrows;
    if ((rc = call_totals_begin(c, st, keep_totals)) != T3D_OK) return rc;
    int bands_done = 0;
    for (int r = 0; r < vrows;) {
        if (c->cancel_cb && c->cancel_cb(c->cancel_user)) { c->last_valid = false; c->err = "render cancelled by the caller's cancel callback"; return T3D_E_CANCELLED; }
        int nb = band < vrows - r ? band : vrows - r;
        size_t off = (size_t)r * o->width;
        // a call without stats cannot see an overflow and try again: it starts from what earlier overflows have taught
        int factor = 2;
        const size_t min_cap = stats ? 0 : c->async_min_cap;
        for (;;) {
            t3d_stats band_stats{};
            // capacity overflow can only be seen after a synchronisation; without stats the call stays asynchronous
            rc = render_band(c, c->work, o, row0, nrows, r, nb, factor, bgr_device + 3 * off, radiance_device ? radiance_device + 3 * off : nullptr,
                             primary_device ? primary_device + 2 * off : nullptr, primary_t_device ? primary_t_device + off : nullptr, st,
                             stats ? &band_stats : nullptr, nullptr, nullptr, 0, false, min_cap);
            if (rc == T3D_E_CAPACITY && factor < factor_limit) {
                const size_t next_cap = (size_t)nb * o->width * S * (size_t)factor * 2;   // entries per level after doubling
                size_t need = next_cap * per_sample / 2;
                // fewer rows instead, where doubling would leave the budget or the 32-bit node indices
                if ((need > c->mem_budget || next_cap * (size_t)levels > (size_t)0x7FFFFFF0) && nb > 1) { nb = (nb + 1) / 2; continue; }
                factor *= 2;
                continue;
            }
            if (rc) { if (rc == T3D_E_CAPACITY) c->err = "trace tree does not fit the node buffers"; return rc; }
            if (keep_totals && (rc = call_totals_add(c, st, levels, (size_t)nb * o->width * S, ++bands_done)) != T3D_OK) return rc;
            if (stats) {
                uint64_t* dst = &stats->rays_api; const uint64_t* src = &band_stats.rays_api;
                for (int k = 0; k < T3D_STATS_N_COUNTERS; ++k) dst[k] += src[k];
                stats->ms_device += band_stats.ms_device; stats->ms_k_level += band_stats.ms_k_level;
                stats->ms_resolve += band_stats.ms_resolve;
                for (int k = 0; k < T3D_STATS_MAX_LEVELS; ++k) {
                    stats->level_rays[k] += band_stats.level_rays[k]; stats->level_nodes[k] += band_stats.level_nodes[k];
                    stats->level_ms[k] += band_stats.level_ms[k];
                }
                stats->bvh_deep_pushes += band_stats.bvh_deep_pushes; stats->shadow_culled += band_stats.shadow_culled;
            }
            break;
        }
        r += nb;
    }
    return T3D_OK;
}

// Lays the work buffers of a band out in `base` (`bytes` long); false if they do not fit.
static bool carve_work(unsigned char* base, size_t bytes, size_t cap, int levels, size_t n0, int n_lights, bool bvh, Work& w) {
    size_t off = 0;
    auto take = [&](size_t n) -> unsigned char* {
        unsigned char* p = base + off;
        off += (n + 255) / 256 * 256;
        return off <= bytes ? p : nullptr;
    };
    const size_t lv = (size_t)levels, L = (size_t)(n_lights > 0 ? n_lights : 1);
    w = Work{};
    bool ok = true;
    auto get = [&](auto*& dst, size_t n) { dst = (std::remove_reference_t<decltype(dst)>)take(n); ok = ok && dst != nullptr; };
    get(w.rays[0], sizeof(RayRec) * cap); get(w.rays[1], sizeof(RayRec) * cap);
    get(w.pool.parent, sizeof(int) * cap * lv); get(w.pool.c, sizeof(float4) * cap * lv); get(w.pool.d, sizeof(float4) * cap * lv);
    get(w.pool.li, sizeof(float) * cap * lv * L);
    if (bvh) {
        get(w.hits, sizeof(uint2) * cap);
        for (int k = 0; k < 2; ++k) { get(w.src_a[k], sizeof(float4) * cap); get(w.src_b[k], sizeof(float4) * cap); }
    }
    get(w.sample_color, sizeof(float4) * n0);
    if (!ok) { w = Work{}; return false; }
    w.cap = cap; w.levels = levels; w.li_lights = n_lights; w.n0 = n0;
    return true;
}

// t3d_render_rows_device for callers that cannot allocate: no hipMalloc / hipFree, no launch graph; every work buffer
// is a piece of the device's static array, the frame is cut into as many bands as that takes.
extern "C" int t3d_render_rows_device_static(t3d_ctx* c, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_device,
                                             float* radiance_device, int32_t* primary_device, float* primary_t_device,
                                             void* stream, t3d_stats* stats) {
    int rc = check_opts(c, o, row0, nrows);
    if (rc) return rc;
    if (!bgr_device) { c->err = "bgr output is required"; return T3D_E_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = take_sticky(c)) != T3D_OK) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (stats) memset(stats, 0, sizeof *stats);
    const int S = o->bounces > 1 ? o->samples : 1;
    const int levels = o->bounces + 1;
    const size_t row_samples = (size_t)o->width * S;
    auto cap_for = [&](size_t n0, int factor) {
        size_t cap = n0 * (size_t)factor;
        if (cap < 64 * 1024) cap = 64 * 1024;
        return (cap + (size_t)T3D_SHARDS * 64 - 1) / ((size_t)T3D_SHARDS * 64) * ((size_t)T3D_SHARDS * 64);
    };
    Work w;
    auto fits = [&](int rows, int factor) {
        return carve_work(c->static_scratch, T3D_STATIC_SCRATCH_BYTES, cap_for((size_t)rows * row_samples, factor), levels,
                          (size_t)rows * row_samples, c->sc.n_lights, c->use_bvh, w);
    };
    // a call without stats cannot see an overflow and try again: it starts from what earlier overflows have taught
    const int factor0 = stats ? 2 : c->static_async_factor;
    if (!fits(1, factor0)) {
        c->err = "t3d_render_rows_device_static: one image row of this width, depth and light count does not fit the static work array (128 MiB)";
        return T3D_E_CAPACITY;
    }
    const int vrows = nrows * (o->frames > 1 ? o->frames : 1);   // the rows of the call, all its frames one after the other
    int band = 1;   // the largest number of rows whose optimistic working set (two rays per sample and level) fits
    for (int lo = 1, hi = vrows; lo <= hi;) { const int mid = lo + (hi - lo) / 2; if (fits(mid, factor0)) { band = mid; lo = mid + 1; } else hi = mid - 1; }
    const bool keep_totals = !stats && band < vrows;
    if ((rc = call_totals_begin(c, st, keep_totals)) != T3D_OK) return rc;
    int bands_done = 0;
    // a level holds at most 2^bounces rays per sample, and the rays of a band of few samples land in few shards
    const int factor_limit = (1 << o->bounces) * T3D_SHARDS;
    for (int r = 0; r < vrows;) {
        if (c->cancel_cb && c->cancel_cb(c->cancel_user)) { c->last_valid = false; c->err = "render cancelled by the caller's cancel callback"; return T3D_E_CANCELLED; }
        int nb = band < vrows - r ? band : vrows - r;
        int factor = factor0;
        const size_t off = (size_t)r * o->width;
        for (;;) {
            if (!fits(nb, factor)) { c->err = "t3d_render_rows_device_static: trace tree does not fit the static work array (128 MiB)"; return T3D_E_CAPACITY; }
            t3d_stats band_stats{};
            rc = render_band(c, w, o, row0, nrows, r, nb, factor, bgr_device + 3 * off,
                             radiance_device ? radiance_device + 3 * off : nullptr, primary_device ? primary_device + 2 * off : nullptr,
                             primary_t_device ? primary_t_device + off : nullptr, st, stats ? &band_stats : nullptr, nullptr, nullptr, 0, true);
            if (rc == T3D_E_CAPACITY) {
                // only a call with stats gets here (it has waited for the band): fewer rows, and for a single row a larger factor
                if (nb > 1) { nb = (nb + 1) / 2; continue; }
                // a single row: the first factor that really gives more room than the capacity just tried (narrow rows sit
                // on the minimum capacity, which doubling the factor would not leave for many rounds)
                const size_t tried = cap_for(row_samples, factor);
                while (factor < factor_limit && cap_for(row_samples, factor) <= tried) factor *= 2;
                if (cap_for(row_samples, factor) <= tried) { c->err = "t3d_render_rows_device_static: trace tree does not fit the node buffers"; return T3D_E_CAPACITY; }
                continue;
            }
            if (rc) return rc;
            if (keep_totals && (rc = call_totals_add(c, st, levels, (size_t)nb * row_samples, ++bands_done)) != T3D_OK) return rc;
            if (stats) {
                uint64_t* dst = &stats->rays_api; const uint64_t* src = &band_stats.rays_api;
                for (int k = 0; k < T3D_STATS_N_COUNTERS; ++k) dst[k] += src[k];
                stats->ms_device += band_stats.ms_device; stats->ms_k_level += band_stats.ms_k_level;
                stats->ms_resolve += band_stats.ms_resolve;
                for (int k = 0; k < T3D_STATS_MAX_LEVELS; ++k) {
                    stats->level_rays[k] += band_stats.level_rays[k]; stats->level_nodes[k] += band_stats.level_nodes[k];
                    stats->level_ms[k] += band_stats.level_ms[k];
                }
                stats->bvh_deep_pushes += band_stats.bvh_deep_pushes; stats->shadow_culled += band_stats.shadow_culled;
            }
            break;
        }
        r += nb;
    }
    return T3D_OK;
}

extern "C" int t3d_set_accel(t3d_ctx* c, int mode) {
    if (!c) return T3D_E_INVALID;
    if (mode != T3D_ACCEL_AUTO && mode != T3D_ACCEL_BRUTE_FORCE && mode != T3D_ACCEL_LBVH) { c->err = "t3d_set_accel: unknown mode"; return T3D_E_INVALID; }
    c->accel = mode;
    return T3D_OK;
}

extern "C" int t3d_set_host_tree_limit(t3d_ctx* c, int max_prims) {
    if (!c) return T3D_E_INVALID;
    if (max_prims < 0) { c->err = "t3d_set_host_tree_limit: negative limit"; return T3D_E_INVALID; }
    c->host_tree_max = max_prims;
    return T3D_OK;
}

extern "C" int t3d_set_rounds(t3d_ctx* c, int mode) {
    if (!c) return T3D_E_INVALID;
    if (mode != T3D_ROUNDS_ADAPTIVE && mode != T3D_ROUNDS_FULL) { c->err = "t3d_set_rounds: unknown mode"; return T3D_E_INVALID; }
    c->full_rounds = mode == T3D_ROUNDS_FULL;
    return T3D_OK;
}

extern "C" int t3d_set_timing(t3d_ctx* c, int per_kernel) {
    if (!c) return T3D_E_INVALID;
    c->timing = per_kernel != 0;
    return T3D_OK;
}

extern "C" int t3d_last_call_stats(t3d_ctx* c, t3d_stats* stats) {
    if (!c || !stats) return T3D_E_INVALID;
    if (!c->last_valid) { c->err = "no render call to report on"; return T3D_E_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->counters_on_host) {
        HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, c->last_stream));
        c->counters_on_host = true;
    }
    // a call of several bands: the sums of all of them (k_add_counters); d_cnt holds its last band only
    if (c->call_totals) HIPCHK(c, hipMemcpyAsync(c->h_totals, c->d_totals, sizeof(CallTotals), hipMemcpyDeviceToHost, c->last_stream));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    memset(stats, 0, sizeof *stats);
    // the sticky words cover every band of the call (and any earlier asynchronous call nobody has asked about)
    int rc = take_sticky(c);
    if (rc != T3D_OK) return rc;
    if (c->h_cnt->overflow) { c->err = "trace tree did not fit the node buffers"; return T3D_E_CAPACITY; }
    fill_stats(c, stats, c->call_totals);
    return T3D_OK;
}

extern "C" int t3d_graph_counters(t3d_ctx* c, uint64_t* captured, uint64_t* replayed, uint64_t* reseeded) {
    if (!c) return T3D_E_INVALID;
    if (captured) *captured = c->graphs_captured;
    if (replayed) *replayed = c->graphs_replayed;
    if (reseeded) *reseeded = c->graphs_reseeded;
    return T3D_OK;
}

extern "C" int t3d_set_cancel(t3d_ctx* c, int (*poll)(void*), void* user) {
    if (!c) return T3D_E_INVALID;
    c->cancel_cb = poll; c->cancel_user = user;
    return T3D_OK;
}

// ------------------------------------------------------------------------------------------------
// Post.upscale / Post.denoise (T3DR.py:1600-1647) on images in device memory, and on host images through it
// ------------------------------------------------------------------------------------------------
extern "C" int t3d_post_upscale_device(t3d_ctx* c, const uint8_t* in_device, int height, int width, int channels, int factor,
                                       uint8_t* out_device, void* stream) {
    if (!c) return T3D_E_INVALID;
    if (!in_device || !out_device || height <= 0 || width <= 0 || channels <= 0 || factor <= 0) { c->err = "t3d_post_upscale: bad argument"; return T3D_E_INVALID; }
    if (height > 65535 || (long long)width * factor * channels > (1ll << 31)) { c->err = "t3d_post_upscale: output too large"; return T3D_E_CAPACITY; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    const unsigned row_bytes = (unsigned)width * factor * channels;
    const int vec = (row_bytes % 16 == 0 && (uintptr_t)out_device % 16 == 0) ? 16 : (row_bytes % 4 == 0 && (uintptr_t)out_device % 4 == 0) ? 4 : 1;
    const dim3 grid((row_bytes / vec + 255) / 256, (unsigned)height);
    if (vec == 16) hipLaunchKernelGGL(k_post_upscale<16>, grid, dim3(256), 0, st, in_device, width, channels, factor, row_bytes, out_device);
    else if (vec == 4) hipLaunchKernelGGL(k_post_upscale<4>, grid, dim3(256), 0, st, in_device, width, channels, factor, row_bytes, out_device);
    else hipLaunchKernelGGL(k_post_upscale<1>, grid, dim3(256), 0, st, in_device, width, channels, factor, row_bytes, out_device);
    HIPCHK(c, hipGetLastError());
    return T3D_OK;
}

extern "C" int t3d_post_denoise_device(t3d_ctx* c, const uint8_t* in_device, int height, int width, int kernel_size, uint8_t* out_device,
                                       void* stream) {
    if (!c) return T3D_E_INVALID;
    if (!in_device || !out_device || in_device == out_device || height <= 0 || width <= 0 || kernel_size < 0) { c->err = "t3d_post_denoise: bad argument"; return T3D_E_INVALID; }
    const int pad = kernel_size / 2;
    if (pad > POST_MAX_PAD) { c->err = "t3d_post_denoise: kernel size above the supported maximum (15)"; return T3D_E_CAPACITY; }
    if ((height + POST_TILE - 1) / POST_TILE > 65535) { c->err = "t3d_post_denoise: image too tall"; return T3D_E_CAPACITY; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    const dim3 grid((unsigned)((width + POST_TILE - 1) / POST_TILE), (unsigned)((height + POST_TILE - 1) / POST_TILE)), block(POST_TILE, POST_TILE);
    switch (pad) {
#define POST_DENOISE_CASE(P) case P: hipLaunchKernelGGL(k_post_denoise<P>, grid, block, 0, st, in_device, height, width, out_device); break;
        POST_DENOISE_CASE(1) POST_DENOISE_CASE(2) POST_DENOISE_CASE(3) POST_DENOISE_CASE(4) POST_DENOISE_CASE(5) POST_DENOISE_CASE(6) POST_DENOISE_CASE(7)
#undef POST_DENOISE_CASE
        default:   // a 1 x 1 window: the image itself
            HIPCHK(c, hipMemcpyAsync(out_device, in_device, (size_t)height * width * 3, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(c, hipGetLastError());
    return T3D_OK;
}

// host image in, host image out: staged through device memory of the call's own (allocated and freed here)
static int post_on_host(t3d_ctx* c, const uint8_t* in, size_t in_bytes, uint8_t* out, size_t out_bytes,
                        const std::function<int(const uint8_t*, uint8_t*)>& run) {
    if (!c) return T3D_E_INVALID;
    if (!in || !out) { c->err = "t3d_post: null image"; return T3D_E_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    int rc = T3D_OK;
    hipError_t e = hipMalloc((void**)&d_in, in_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, out_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) rc = run(d_in, d_out);
    if (e == hipSuccess && rc == T3D_OK) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == T3D_OK) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    hipFree(d_in); hipFree(d_out);
    if (e != hipSuccess) { c->err = std::string("t3d_post: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? T3D_E_NOMEM : T3D_E_DEVICE; }
    return rc;
}

extern "C" int t3d_post_upscale(t3d_ctx* c, const uint8_t* in, int height, int width, int channels, int factor, uint8_t* out) {
    if (!c) return T3D_E_INVALID;
    if (height <= 0 || width <= 0 || channels <= 0 || factor <= 0) { c->err = "t3d_post_upscale: bad argument"; return T3D_E_INVALID; }
    const size_t nin = (size_t)height * width * channels;
    return post_on_host(c, in, nin, out, nin * factor * factor,
                        [&](const uint8_t* d_in, uint8_t* d_out) { return t3d_post_upscale_device(c, d_in, height, width, channels, factor, d_out, c->stream); });
}

extern "C" int t3d_post_denoise(t3d_ctx* c, const uint8_t* in, int height, int width, int kernel_size, uint8_t* out) {
    if (!c) return T3D_E_INVALID;
    if (height <= 0 || width <= 0) { c->err = "t3d_post_denoise: bad argument"; return T3D_E_INVALID; }
    const size_t n = (size_t)height * width * 3;
    return post_on_host(c, in, n, out, n,
                        [&](const uint8_t* d_in, uint8_t* d_out) { return t3d_post_denoise_device(c, d_in, height, width, kernel_size, d_out, c->stream); });
}

// device-side outputs of the entry points that hand back host memory
static int ensure_outputs(t3d_ctx* c, size_t npix) {
    if (npix > c->out_alloc) {
        hipFree(c->d_bgr); hipFree(c->d_rad); hipFree(c->d_prim); hipFree(c->d_prim_t);
        c->d_bgr = nullptr; c->d_rad = nullptr; c->d_prim = nullptr; c->d_prim_t = nullptr; c->out_alloc = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_bgr, npix * 3));
        HIPCHK(c, hipMalloc((void**)&c->d_rad, npix * 3 * sizeof(float)));
        HIPCHK(c, hipMalloc((void**)&c->d_prim, npix * 2 * sizeof(int32_t)));
        HIPCHK(c, hipMalloc((void**)&c->d_prim_t, npix * sizeof(float)));
        c->out_alloc = npix;
    }
    return T3D_OK;
}

extern "C" int t3d_render_rows(t3d_ctx* c, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_out, float* radiance_out,
                               int32_t* primary_out, float* primary_t_out, t3d_stats* stats) {
    int rc = check_opts(c, o, row0, nrows);
    if (rc) return rc;
    if (!bgr_out) { c->err = "bgr output is required"; return T3D_E_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    size_t npix = (size_t)nrows * (o->frames > 1 ? o->frames : 1) * o->width;
    if ((rc = ensure_outputs(c, npix)) != T3D_OK) return rc;
    t3d_stats local{};
    rc = t3d_render_rows_device(c, o, row0, nrows, c->d_bgr, radiance_out ? c->d_rad : nullptr, primary_out ? c->d_prim : nullptr,
                                primary_t_out ? c->d_prim_t : nullptr, nullptr, &local);
    if (rc) return rc;
    // A copy into pageable memory runs at a few GB/s; through a pinned staging buffer the link runs at its own rate
    // and the last hop is a host memcpy.
    const size_t sz_bgr = npix * 3, sz_rad = radiance_out ? npix * 3 * sizeof(float) : 0;
    const size_t sz_prim = primary_out ? npix * 2 * sizeof(int32_t) : 0, sz_pt = primary_t_out ? npix * sizeof(float) : 0;
    const size_t off_rad = (sz_bgr + 255) / 256 * 256, off_prim = off_rad + (sz_rad + 255) / 256 * 256, off_pt = off_prim + (sz_prim + 255) / 256 * 256;
    const size_t need = off_pt + sz_pt;
    if (need > c->stage_alloc) {
        if (c->h_stage) hipHostFree(c->h_stage);
        c->h_stage = nullptr; c->stage_alloc = 0;
        HIPCHK(c, hipHostMalloc(&c->h_stage, need, hipHostMallocDefault));
        c->stage_alloc = need;
    }
    char* hs = (char*)c->h_stage;
    HIPCHK(c, hipMemcpyAsync(hs, c->d_bgr, sz_bgr, hipMemcpyDeviceToHost, c->stream));
    if (sz_rad) HIPCHK(c, hipMemcpyAsync(hs + off_rad, c->d_rad, sz_rad, hipMemcpyDeviceToHost, c->stream));
    if (sz_prim) HIPCHK(c, hipMemcpyAsync(hs + off_prim, c->d_prim, sz_prim, hipMemcpyDeviceToHost, c->stream));
    if (sz_pt) HIPCHK(c, hipMemcpyAsync(hs + off_pt, c->d_prim_t, sz_pt, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(bgr_out, hs, sz_bgr);
    if (sz_rad) memcpy(radiance_out, hs + off_rad, sz_rad);
    if (sz_prim) memcpy(primary_out, hs + off_prim, sz_prim);
    if (sz_pt) memcpy(primary_t_out, hs + off_pt, sz_pt);
    if (stats) *stats = local;
    return T3D_OK;
}

extern "C" int t3d_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return T3D_E_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? T3D_OK : T3D_E_NOMEM;
}

extern "C" int t3d_host_free(void* p) {
    if (!p) return T3D_OK;
    return hipHostFree(p) == hipSuccess ? T3D_OK : T3D_E_INVALID;
}

extern "C" int t3d_render_rows_async(t3d_ctx* c, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_pinned) {
    int rc = check_opts(c, o, row0, nrows);
    if (rc) return rc;
    if (!bgr_pinned) { c->err = "bgr output is required"; return T3D_E_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)nrows * (o->frames > 1 ? o->frames : 1) * o->width;
    if ((rc = ensure_outputs(c, npix)) != T3D_OK) return rc;
    rc = t3d_render_rows_device(c, o, row0, nrows, c->d_bgr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(bgr_pinned, c->d_bgr, npix * 3, hipMemcpyDeviceToHost, c->stream));
    return T3D_OK;
}

// The rows of one call as n contiguous bands, band k traced by context k on that context's own stream, all of them at
// once: the launches of one band leave the device part empty while they start and while their last waves finish, and
// another band's launches fill it (one 1080p frame: 0.52 ms on one context, 0.44 ms as two bands).  The contexts' streams
// are forked from the call's stream and joined to it again by events, so to the caller the call behaves like
// t3d_render_rows_device without stats on `st`.
static int render_rows_split(t3d_ctx* const* ctxs, int n, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_device, hipStream_t st) {
    t3d_ctx* c0 = ctxs[0];
    for (int k = 0; k < n; ++k) {
        if (!ctxs[k]) { c0->err = "t3d_render_rows_device_split: null context"; return T3D_E_INVALID; }
        if (ctxs[k]->device != c0->device) { c0->err = "t3d_render_rows_device_split: the contexts must share a device"; return T3D_E_INVALID; }
        for (int j = 0; j < k; ++j) if (ctxs[j] == ctxs[k]) { c0->err = "t3d_render_rows_device_split: the same context twice"; return T3D_E_INVALID; }
    }
    if (o && o->frames > 1) { c0->err = "t3d_render_rows_device_split: one frame per call"; return T3D_E_INVALID; }
    int rc = check_opts(c0, o, row0, nrows);
    if (rc) return rc;
    if (!bgr_device) { c0->err = "bgr output is required"; return T3D_E_INVALID; }
    HIPCHK(c0, hipSetDevice(c0->device));
    const int stride = o->row_stride > 1 ? o->row_stride : 1;
    // bands of whole 8-row blocks (the unit of the pixel tiles level 0 walks) while there are enough rows for that
    const int unit = nrows >= 8 * n ? 8 : 1;
    auto bound = [&](int k) { return k >= n ? nrows : (int)((long long)nrows * k / n / unit * unit); };
    if (n > 1) HIPCHK(c0, hipEventRecord(c0->ev_fork, st));
    int first_error = T3D_OK;
    for (int k = 0; k < n; ++k) {
        t3d_ctx* c = ctxs[k];
        const int b0 = bound(k), b1 = bound(k + 1);
        if (b1 <= b0) continue;
        hipStream_t sk = k == 0 ? st : c->stream;
        if (k > 0) HIPCHK(c0, hipStreamWaitEvent(sk, c0->ev_fork, 0));
        rc = t3d_render_rows_device(c, o, row0 + b0 * stride, b1 - b0, bgr_device + (size_t)b0 * o->width * 3, nullptr, nullptr, nullptr, sk, nullptr);
        if (rc != T3D_OK && first_error == T3D_OK) { first_error = rc; if (k) c0->err = "band " + std::to_string(k) + ": " + c->err; }
        if (k > 0) {
            HIPCHK(c0, hipEventRecord(c->ev_join, sk));
            HIPCHK(c0, hipStreamWaitEvent(st, c->ev_join, 0));
        }
    }
    return first_error;
}

extern "C" int t3d_render_rows_device_split(t3d_ctx* const* ctxs, int n, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_device, void* stream) {
    if (!ctxs || n < 1 || n > 16 || !ctxs[0]) return T3D_E_INVALID;
    return render_rows_split(ctxs, n, o, row0, nrows, bgr_device, stream ? (hipStream_t)stream : ctxs[0]->stream);
}

extern "C" int t3d_render_rows_async_split(t3d_ctx* const* ctxs, int n, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_pinned) {
    if (!ctxs || n < 1 || n > 16 || !ctxs[0]) return T3D_E_INVALID;
    t3d_ctx* c = ctxs[0];
    int rc = check_opts(c, o, row0, nrows);
    if (rc) return rc;
    if (!bgr_pinned) { c->err = "bgr output is required"; return T3D_E_INVALID; }
    if (o->frames > 1) { c->err = "t3d_render_rows_async_split: one frame per call"; return T3D_E_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)nrows * o->width;
    if ((rc = ensure_outputs(c, npix)) != T3D_OK) return rc;
    rc = render_rows_split(ctxs, n, o, row0, nrows, c->d_bgr, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(bgr_pinned, c->d_bgr, npix * 3, hipMemcpyDeviceToHost, c->stream));
    return T3D_OK;
}

// One frame on several devices (or several contexts of one device): context k renders image rows k, k + n, k + 2n, ...
// (pixels are independent and the random numbers are keyed by absolute pixel coordinates, so the picture does not depend
// on the partition) and copies them, with one strided copy on its own stream, straight to their places in the frame:
// the reference's stripes per thread and the np.vstack that joins them (T3DR.py:1111-1165).  One host thread per
// context, so the launches of the devices do not queue up behind each other on the host.
extern "C" int t3d_render_frame_multi(t3d_ctx* const* ctxs, int n, const t3d_opts* o, uint8_t* bgr_frame, int frame_on_device,
                                      t3d_stats* stats) {
    if (!ctxs || n < 1 || !ctxs[0]) return T3D_E_INVALID;
    t3d_ctx* c0 = ctxs[0];
    for (int k = 0; k < n; ++k) if (!ctxs[k]) { c0->err = "t3d_render_frame_multi: null context"; return T3D_E_INVALID; }
    for (int k = 0; k < n; ++k) for (int j = 0; j < k; ++j) if (ctxs[j] == ctxs[k]) { c0->err = "t3d_render_frame_multi: the same context twice"; return T3D_E_INVALID; }
    if (!o || !bgr_frame) { c0->err = "t3d_render_frame_multi: bad argument"; return T3D_E_INVALID; }
    if (o->height <= 0 || o->height % n != 0) { c0->err = "Number of GPUs must be a factor of image height"; return T3D_E_INVALID; }
    const int rows = o->height / n;
    std::vector<int> rcs((size_t)n, T3D_OK);
    std::vector<t3d_stats> sts((size_t)n);
    auto work = [&](int k) {
        t3d_ctx* c = ctxs[k];
        t3d_opts ok = *o;
        ok.row_stride = n; ok.frames = 1;
        int rc = check_opts(c, &ok, k, rows);
        if (rc == T3D_OK && hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; rc = T3D_E_DEVICE; }
        if (rc == T3D_OK) rc = ensure_outputs(c, (size_t)rows * o->width);
        const size_t line = (size_t)o->width * 3;
        auto place = [&]() -> int {   // the gather: tile rows to frame rows k, k + n, ...
            hipError_t e = hipMemcpy2DAsync(bgr_frame + (size_t)k * line, line * n, c->d_bgr, line, line, rows,
                                            frame_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) { c->err = std::string("t3d_render_frame_multi: tile copy: ") + hipGetErrorString(e); return T3D_E_DEVICE; }
            return T3D_OK;
        };
        if (rc == T3D_OK) rc = t3d_render_rows_device(c, &ok, k, rows, c->d_bgr, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc == T3D_OK) rc = place();
        if (rc == T3D_OK) rc = t3d_last_call_stats(c, &sts[k]);
        if (rc == T3D_E_CAPACITY) {
            // the trees outgrew the buffers of the unchecked call: once more, checked, which grows them
            rc = t3d_render_rows_device(c, &ok, k, rows, c->d_bgr, nullptr, nullptr, nullptr, nullptr, &sts[k]);
            if (rc == T3D_OK) rc = place();
            if (rc == T3D_OK && hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "t3d_render_frame_multi: synchronise failed"; rc = T3D_E_DEVICE; }
        }
        rcs[k] = rc;
    };
    if (n == 1) work(0);
    else {
        std::vector<std::thread> threads;
        for (int k = 0; k < n; ++k) threads.emplace_back(work, k);
        for (auto& t : threads) t.join();
    }
    for (int k = 0; k < n; ++k)
        if (rcs[k] != T3D_OK) { if (k) c0->err = "context " + std::to_string(k) + ": " + ctxs[k]->err; return rcs[k]; }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int k = 0; k < n; ++k) {
            uint64_t* dst = &stats->rays_api; const uint64_t* src = &sts[k].rays_api;
            for (int q = 0; q < T3D_STATS_N_COUNTERS; ++q) dst[q] += src[q];
            // the devices work side by side: the frame's device time is the slowest tile's
            stats->ms_device = fmax(stats->ms_device, sts[k].ms_device); stats->ms_k_level = fmax(stats->ms_k_level, sts[k].ms_k_level);
            stats->ms_resolve = fmax(stats->ms_resolve, sts[k].ms_resolve);
            for (int q = 0; q < T3D_STATS_MAX_LEVELS; ++q) {
                stats->level_rays[q] += sts[k].level_rays[q]; stats->level_nodes[q] += sts[k].level_nodes[q];
                stats->level_ms[q] = fmax(stats->level_ms[q], sts[k].level_ms[q]);
            }
            stats->bvh_deep_pushes += sts[k].bvh_deep_pushes; stats->shadow_culled += sts[k].shadow_culled;
        }
    }
    return T3D_OK;
}

// One initial ray through every level; its tree in depth-first order (bounce child before transmit child).
static int trace_single(t3d_ctx* c, const t3d_opts* o, const RenderParams& single, int row, t3d_node* nodes, int capacity) {
    HIPCHK(c, hipSetDevice(c->device));
    int rc = take_sticky(c);
    if (rc != T3D_OK) return rc;
    if ((rc = adopt_background_tree(c)) != T3D_OK) return rc;
    const int worst = (1 << (o->bounces + 1)) - 1;   // a full binary tree: T3D_MAX_BOUNCES keeps this below 2^15
    const int dbg_cap = worst;
    if (dbg_cap > c->dbg_alloc) {
        hipFree(c->d_dbg); c->d_dbg = nullptr; c->dbg_alloc = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_dbg, sizeof(t3d_node) * dbg_cap));
        c->dbg_alloc = dbg_cap;
    }
    t3d_node* d_dbg = c->d_dbg;
    t3d_stats st{};
    // The whole tree lives in one queue shard (render_band uses a single shard and a single workgroup for it), and a
    // level of it holds at most 2^bounces rays: that is all the capacity a level needs.
    const int factor = 1 << o->bounces;
    c->call_totals = false; c->call_bands = 1;
    rc = render_band(c, c->single_work, o, row, 1, 0, 1, factor, nullptr, nullptr, nullptr, nullptr, c->stream, &st, &single, d_dbg, dbg_cap);
    int n = 0;
    if (!rc) {
        n = c->h_cnt->n_dbg < dbg_cap ? c->h_cnt->n_dbg : dbg_cap;
        std::vector<t3d_node> tmp((size_t)(n > 0 ? n : 1));
        if (n > 0 && (hipMemcpyAsync(tmp.data(), d_dbg, sizeof(t3d_node) * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                      hipStreamSynchronize(c->stream) != hipSuccess)) { rc = T3D_E_DEVICE; c->err = "trace: copy of the node records failed"; }
        if (!rc) {
            // depth-first order, bounce child (2n) before transmit child (2n+1); heap index -> record
            std::vector<int> where((size_t)worst + 1, -1);
            for (int i = 0; i < n; ++i) if (tmp[i].heap >= 1 && tmp[i].heap <= worst) where[tmp[i].heap] = i;
            std::vector<int> order;
            std::vector<int> stack;
            stack.push_back(1);
            while (!stack.empty()) {
                int hidx = stack.back(); stack.pop_back();
                if (hidx > worst || where[hidx] < 0) continue;
                order.push_back(where[hidx]);
                stack.push_back(2 * hidx + 1); stack.push_back(2 * hidx);
            }
            int m = (int)order.size();
            if (m > capacity) { rc = T3D_E_CAPACITY; c->err = "trace: node buffer too small"; }
            else { for (int i = 0; i < m; ++i) nodes[i] = tmp[order[i]]; n = m; }
        }
    }
    return rc ? rc : n;
}

extern "C" int t3d_trace_pixel(t3d_ctx* c, const t3d_opts* o, int x, int y, int sample, t3d_node* nodes, int capacity) {
    int rc = check_opts(c, o, 0, 1);
    if (rc) return rc;
    if (!nodes || capacity <= 0 || x < 0 || y < 0 || x >= o->width || y >= o->height || sample < 0) { c->err = "t3d_trace_pixel: bad argument"; return T3D_E_INVALID; }
    RenderParams single{};
    single.single = 1; single.px0 = x; single.py0 = y; single.sample0 = sample;
    return trace_single(c, o, single, y, nodes, capacity);
}

extern "C" int t3d_trace_ray(t3d_ctx* c, const t3d_opts* o, const float* origin, const float* direction, float ior, int inside_object,
                             int key_x, int key_y, int key_sample, t3d_node* nodes, int capacity) {
    int rc = check_opts(c, o, 0, 1);
    if (rc) return rc;
    if (!nodes || capacity <= 0 || !origin || !direction || key_x < 0 || key_y < 0 || key_sample < 0 || inside_object < -1 ||
        inside_object >= c->n_objects) { c->err = "t3d_trace_ray: bad argument"; return T3D_E_INVALID; }
    const double len = sqrt((double)direction[0] * direction[0] + (double)direction[1] * direction[1] + (double)direction[2] * direction[2]);
    if (!(len > 0)) { c->err = "t3d_trace_ray: zero direction"; return T3D_E_INVALID; }
    RenderParams single{};
    single.single = 2; single.px0 = key_x; single.py0 = key_y; single.sample0 = key_sample;
    for (int k = 0; k < 3; ++k) { single.ray_o[k] = origin[k]; single.ray_d[k] = (float)(direction[k] / len); }  // Ray.__init__ normalises (T3DE.py:126-130)
    single.ray_ior = ior; single.ray_obj = inside_object;
    return trace_single(c, o, single, 0, nodes, capacity);
}

extern "C" int t3d_closest_hits(t3d_ctx* c, int n, const float* origins, const float* dirs, int lights, int back_cull,
                                int32_t* out_obj, int32_t* out_face, float* out_t, int32_t* out_dirflag, float* out_normal) {
    if (!c) return T3D_E_INVALID;
    if (!c->has_scene) { c->err = "no scene uploaded"; return T3D_E_INVALID; }
    if (n < 0 || !origins || !dirs || !out_obj || !out_face || !out_t || !out_dirflag) { c->err = "t3d_closest_hits: bad argument"; return T3D_E_INVALID; }
    if (n == 0) return T3D_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = adopt_background_tree(c)) return rc;
    // The rays and the answers live in one pinned host buffer of the context that the kernel reads and writes in place
    // (44 bytes in, 32 bytes out per ray over the host link): a picked pixel is a launch and a wait, with no device
    // allocation and no staged copies.
    const size_t n3 = (size_t)n * 3 * sizeof(float), n1 = (size_t)n * sizeof(float);
    const size_t need = 3 * n3 + 4 * n1;
    if (need > c->pick_alloc) {
        if (c->h_pick) hipHostFree(c->h_pick);
        c->h_pick = nullptr; c->pick_alloc = 0;
        const size_t want = need < 4096 ? 4096 : need + need / 2;
        HIPCHK(c, hipHostMalloc(&c->h_pick, want, hipHostMallocDefault));
        c->pick_alloc = want;
    }
    char* base = (char*)c->h_pick;
    float* h_o = (float*)base; float* h_d = (float*)(base + n3); float* h_n = (float*)(base + 2 * n3);
    float* h_t = (float*)(base + 3 * n3);
    int32_t* h_obj = (int32_t*)(base + 3 * n3 + n1); int32_t* h_face = (int32_t*)(base + 3 * n3 + 2 * n1); int32_t* h_dir = (int32_t*)(base + 3 * n3 + 3 * n1);
    memcpy(h_o, origins, n3);
    memcpy(h_d, dirs, n3);
    const int pick_blocks = (int)std::min<size_t>(((size_t)n + T3D_BLOCK - 1) / T3D_BLOCK, (size_t)c->n_sm * T3D_BVH_WAVES);
    hipLaunchKernelGGL(k_closest, dim3(pick_blocks), dim3(T3D_BLOCK), 0, c->stream, c->sc, (const float4*)c->d_planes,
                       (const int*)c->d_plane_first, (const float4*)c->d_tri_uv, (const DevSphere*)c->d_spheres, n, h_o, h_d, lights, back_cull, h_obj, h_face, h_t, h_dir, h_n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(out_obj, h_obj, n1); memcpy(out_face, h_face, n1); memcpy(out_t, h_t, n1); memcpy(out_dirflag, h_dir, n1);
    if (out_normal) memcpy(out_normal, h_n, n3);
    return T3D_OK;
}
