// t3d_lbvh.h -- linear bounding-volume hierarchy built on the GPU (Morton codes, radix sort, Karras'
// hierarchy, bottom-up refit) over the triangles and spheres of a scene, for scenes too large for the
// brute-force plane-group loop.
//
// The reference culls with a per-mesh median-split BVH of padded boxes (Object.buildBVH T3DE.py:985-1022,
// Photon.getFaces T3DR.py:553-569, Ray.entersAABB T3DE.py:166-204).  That cull is purely conservative: the
// closest hit it returns is the brute-force closest hit.  This hierarchy keeps the same contract (boxes are
// padded outward, a node is entered whenever its slab interval reaches the current best distance) and is
// otherwise free to differ: one tree over the whole scene, built from Morton order.
//
// Included by t3d_hip.hip (one translation unit, one shared library).
#pragma once
#include "t3d_host_tree.h"

#define LBVH_BLOCK 256

struct LbvhWork {
    float4* prim_lo = nullptr;   // [N] box of primitive p (w unused)
    float4* prim_hi = nullptr;
    uint32_t* keys[2] = {nullptr, nullptr};
    int* vals[2] = {nullptr, nullptr};
    int* hist = nullptr;
    int* sums = nullptr;         // [1024] tile totals of the scan
    int* parent = nullptr;       // [2N-1]: internal nodes 0..N-2, leaves N-1..2N-2 (sorted position + N-1)
    int2* kids = nullptr;        // [N-1]
    float4* box_lo = nullptr;    // [N-1] boxes of internal nodes
    float4* box_hi = nullptr;
    int* flags = nullptr;        // [N-1] arrival counters of the refit
    int* bounds = nullptr;       // [6] scene bounds as order-preserving ints
    int sorted = 0;              // keys[sorted] / vals[sorted] hold the sorted order (kept for lbvh_refit)
    int rounds = 0;              // launches of k_lbvh_refit_round this tree's boxes take (its height); 0: not known yet, < 0: the climbing kernel instead
    void release() {
        hipFree(prim_lo); hipFree(prim_hi); hipFree(keys[0]); hipFree(keys[1]); hipFree(vals[0]); hipFree(vals[1]);
        hipFree(hist); hipFree(sums); hipFree(parent); hipFree(kids); hipFree(box_lo); hipFree(box_hi); hipFree(flags); hipFree(bounds);
        *this = LbvhWork();
    }
};

// float <-> int that keeps the order (for atomicMin / atomicMax on floats)
__device__ __forceinline__ int lbvh_f2o(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float lbvh_o2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// Primitive boxes, padded outward so that float32 rounding of a hit point can never leave the box of the
// primitive it lies on, and the scene bounds.
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_bounds(const float* __restrict__ verts, int n_tris, const DevSphere* __restrict__ spheres,
                                                            int n_spheres, float4* __restrict__ prim_lo, float4* __restrict__ prim_hi,
                                                            int* __restrict__ bounds, int first = 0) {
    // first: the primitives from this one on only (a refit: the triangles have not moved, first = n_tris)
    const int n = n_tris + n_spheres;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int p = first + blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        float blo[3], bhi[3];
        if (p < n_tris) {
            const float* v = verts + 9 * (size_t)p;
            for (int a = 0; a < 3; ++a) {
                blo[a] = fminf(fminf(v[a], v[3 + a]), v[6 + a]);
                bhi[a] = fmaxf(fmaxf(v[a], v[3 + a]), v[6 + a]);
            }
        } else {
            const DevSphere sp = spheres[p - n_tris];
            const float r = __builtin_sqrtf(sp.r2) * 1.000001f;
            blo[0] = sp.cx - r; blo[1] = sp.cy - r; blo[2] = sp.cz - r;
            bhi[0] = sp.cx + r; bhi[1] = sp.cy + r; bhi[2] = sp.cz + r;
        }
        float ext = 0.0f;
        for (int a = 0; a < 3; ++a) ext = fmaxf(ext, fmaxf(fmaxf(fabsf(blo[a]), fabsf(bhi[a])), bhi[a] - blo[a]));
        const float pad = 4e-6f * ext + 1e-30f;
        for (int a = 0; a < 3; ++a) {
            blo[a] -= pad; bhi[a] += pad;
            lo[a] = fminf(lo[a], blo[a]); hi[a] = fmaxf(hi[a], bhi[a]);
        }
        prim_lo[p] = make_float4(blo[0], blo[1], blo[2], 0.f);
        prim_hi[p] = make_float4(bhi[0], bhi[1], bhi[2], 0.f);
    }
    // wave reduction, then one atomic per wave and bound
    for (int a = 0; a < 3; ++a) {
        float l = lo[a], h = hi[a];
        for (int off = 32; off > 0; off >>= 1) {
            l = fminf(l, __shfl_xor(l, off));
            h = fmaxf(h, __shfl_xor(h, off));
        }
        if ((threadIdx.x & 63) == 0 && l <= h) {
            atomicMin(&bounds[a], lbvh_f2o(l));
            atomicMax(&bounds[3 + a], lbvh_f2o(h));
        }
    }
}

__device__ __forceinline__ uint32_t lbvh_spread10(uint32_t v) {  // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// 30-bit Morton code of the box centre inside the scene bounds.
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_morton(const float4* __restrict__ prim_lo, const float4* __restrict__ prim_hi, int n,
                                                            const int* __restrict__ bounds, uint32_t* __restrict__ keys, int* __restrict__ vals) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 lo = prim_lo[p], hi = prim_hi[p];
    const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float l = lbvh_o2f(bounds[a]), h = lbvh_o2f(bounds[3 + a]);
        const float ext = h - l;
        float x = ext > 0.0f ? (c[a] - l) / ext : 0.0f;
        x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
        q[a] = (uint32_t)x;
    }
    keys[p] = (lbvh_spread10(q[0]) << 2) | (lbvh_spread10(q[1]) << 1) | lbvh_spread10(q[2]);
    vals[p] = p;
}

// Stable LSD radix sort, 8 bits per pass.  Thread t owns the contiguous chunk [t*chunk, (t+1)*chunk): it
// counts its digits into hist[digit * nt + t], an exclusive scan over the digit-major table gives every (digit, thread)
// its first output slot, and the thread scatters its chunk in order.  Not a fast sort; the build runs once
// per scene.
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_rs_count(const uint32_t* __restrict__ keys, int n, int shift, int chunk, int nt,
                                                              int* __restrict__ hist) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const int b = t * chunk, e = min(n, b + chunk);
    for (int i = b; i < e; ++i) hist[((keys[i] >> shift) & 255u) * nt + t] += 1;
}
// Exclusive scan of the table in three launches: every block scans one tile of 4096 entries (4 per thread) and
// leaves its total in sums[]; one block scans the totals; every tile then adds its offset.  m <= 1024 tiles.
#define LBVH_SCAN_TILE 4096
__device__ __forceinline__ int lbvh_block_exclusive(int v, int* part) {   // exclusive prefix of v over the 1024 threads of a block
    const int t = threadIdx.x;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int u = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += u;
        __syncthreads();
    }
    return part[t] - v;
}
__global__ __launch_bounds__(1024) void k_lbvh_scan_tiles(int* __restrict__ hist, int m, int* __restrict__ sums) {
    __shared__ int part[1024];
    const int base = blockIdx.x * LBVH_SCAN_TILE + threadIdx.x * 4;
    int v[4];
    for (int j = 0; j < 4; ++j) v[j] = base + j < m ? hist[base + j] : 0;
    const int total = v[0] + v[1] + v[2] + v[3];
    int run = lbvh_block_exclusive(total, part);
    for (int j = 0; j < 4; ++j) { if (base + j < m) hist[base + j] = run; run += v[j]; }
    if (threadIdx.x == 1023) sums[blockIdx.x] = run;   // the last thread's running sum is the tile's total
}
__global__ __launch_bounds__(1024) void k_lbvh_scan_sums(int* __restrict__ sums, int n_tiles) {
    __shared__ int part[1024];
    const int v = (int)threadIdx.x < n_tiles ? sums[threadIdx.x] : 0;
    const int ex = lbvh_block_exclusive(v, part);
    if ((int)threadIdx.x < n_tiles) sums[threadIdx.x] = ex;
}
__global__ __launch_bounds__(1024) void k_lbvh_scan_add(int* __restrict__ hist, int m, const int* __restrict__ sums) {
    const int base = blockIdx.x * LBVH_SCAN_TILE + threadIdx.x * 4;
    const int off = sums[blockIdx.x];
    for (int j = 0; j < 4; ++j) if (base + j < m) hist[base + j] += off;
}
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_rs_scatter(const uint32_t* __restrict__ keys, const int* __restrict__ vals, int n, int shift,
                                                                int chunk, int nt, int* __restrict__ hist, uint32_t* __restrict__ keys_out,
                                                                int* __restrict__ vals_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const int b = t * chunk, e = min(n, b + chunk);
    for (int i = b; i < e; ++i) {
        const uint32_t k = keys[i];
        const int slot = ((k >> shift) & 255u) * nt + t;
        const int pos = hist[slot];
        hist[slot] = pos + 1;
        keys_out[pos] = k;
        vals_out[pos] = vals[i];
    }
}

// Length of the common prefix of the keys at sorted positions i and j; equal keys are told apart by
// their positions (Karras 2012), out-of-range j gives -1.
__device__ __forceinline__ int lbvh_delta(const uint32_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    return a != b ? __clz(a ^ b) : 32 + __clz((uint32_t)i ^ (uint32_t)j);
}

// One thread per internal node: its key range, split position and children.
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_hierarchy(const uint32_t* __restrict__ keys, int n, int2* __restrict__ kids,
                                                               int* __restrict__ parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    // node ids: internal g -> g, leaf at sorted position g -> (n - 1) + g
    const int left = lo == gamma ? (n - 1) + gamma : gamma;
    const int right = hi == gamma + 1 ? (n - 1) + gamma + 1 : gamma + 1;
    kids[i] = make_int2(left, right);
    parent[left] = i;
    parent[right] = i;
    if (i == 0) parent[0] = -1;
}

// Bottom-up boxes: every leaf climbs; the first thread to reach a node stops, the second one (whose
// sibling subtree is therefore complete) computes the node's box and goes on.
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_refit(const int* __restrict__ vals, int n, const float4* __restrict__ prim_lo,
                                                           const float4* __restrict__ prim_hi, const int2* __restrict__ kids,
                                                           const int* __restrict__ parent, float4* box_lo, float4* box_hi, int* flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cur = parent[(n - 1) + i];
    while (cur >= 0) {
        // release our subtree's boxes, acquire the sibling's
        if (__hip_atomic_fetch_add(&flags[cur], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        const int2 k = kids[cur];
        float4 l0, h0, l1, h1;
        if (k.x >= n - 1) { const int p = vals[k.x - (n - 1)]; l0 = prim_lo[p]; h0 = prim_hi[p]; }
        else { l0 = box_lo[k.x]; h0 = box_hi[k.x]; }
        if (k.y >= n - 1) { const int p = vals[k.y - (n - 1)]; l1 = prim_lo[p]; h1 = prim_hi[p]; }
        else { l1 = box_lo[k.y]; h1 = box_hi[k.y]; }
        box_lo[cur] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.f);
        box_hi[cur] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.f);
        cur = parent[cur];
    }
}

// The same boxes level by level, without a hand-over inside a launch: launch number r (1, 2, ...) finishes every node
// whose children are leaves or were finished by an EARLIER launch (their flag holds that launch's number; a flag written
// in this launch reads as r or as 0 and counts as not ready, so nothing depends on what one wave sees of another's
// stores).  As many launches as the tree is high -- 30 to 45 for a million primitives, a few microseconds each once
// most nodes are done and a thread leaves after one coalesced load -- against one launch of k_lbvh_refit, whose
// agent-scope acquire-release per node costs an L2 write-back and invalidate each: 3.8 ms for a million primitives.
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_refit_round(const int* __restrict__ vals, int n, const float4* __restrict__ prim_lo,
                                                                 const float4* __restrict__ prim_hi, const int2* __restrict__ kids,
                                                                 float4* box_lo, float4* box_hi, int* flags, int round) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || flags[i] != 0) return;
    const int2 k = kids[i];
    const bool leaf0 = k.x >= n - 1, leaf1 = k.y >= n - 1;
    const int f0 = leaf0 ? 1 : flags[k.x], f1 = leaf1 ? 1 : flags[k.y];
    if (!((leaf0 || (f0 != 0 && f0 < round)) && (leaf1 || (f1 != 0 && f1 < round)))) return;
    float4 l0, h0, l1, h1;
    if (leaf0) { const int p = vals[k.x - (n - 1)]; l0 = prim_lo[p]; h0 = prim_hi[p]; }
    else { l0 = box_lo[k.x]; h0 = box_hi[k.x]; }
    if (leaf1) { const int p = vals[k.y - (n - 1)]; l1 = prim_lo[p]; h1 = prim_hi[p]; }
    else { l1 = box_lo[k.y]; h1 = box_hi[k.y]; }
    box_lo[i] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.f);
    box_hi[i] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.f);
    flags[i] = round;
}

// Traversal nodes, 4 wide and quantised: the record of binary node i describes its GRANDchildren (a child that is
// a leaf takes one slot itself), so one fetch decides four subtrees and a ray makes half as many dependent fetches
// on its way down.  The traversal is bound by the number of 16-byte accesses a lane makes, not by arithmetic, so a
// record is three rows (48 bytes used of a 64-byte stride, never straddling a cache line):
//   row 0  origin.x, origin.y, origin.z of the record's own frame; meta = ex | ey << 8 | ez << 16 | used << 24 | leaf << 28
//          (e*: biased exponent of the power-of-two grid step of the axis; used, leaf: one bit per slot)
//   row 1  lo.x, hi.x, lo.y, hi.y of the four slots, one byte per slot (slot k in bits 8k..8k+7)
//   row 2  lo.z, hi.z the same way; base01, base23
// A slot's box is origin + q * step, lo rounded down and hi rounded up, so it contains the exact (padded) box.
// Links need no storage of their own: in this hierarchy the children of a node split at position g are node or
// leaf g and node or leaf g + 1, so slots 0, 1 are base01, base01 + 1 and slots 2, 3 are base23, base23 + 1 (a child
// that is a leaf itself: slot 0 / 2 is that leaf, slot 1 / 3 empty).  Leaves are numbered by sorted position.
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_pack(const int* __restrict__ vals, int n, const float4* __restrict__ prim_lo,
                                                          const float4* __restrict__ prim_hi, const int2* __restrict__ kids,
                                                          const float4* __restrict__ box_lo, const float4* __restrict__ box_hi,
                                                          float4* __restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    float lo[4][3], hi[4][3];
    int kind[4] = {0, 0, 0, 0}, base[2] = {0, 0};
    auto put = [&](int slot, int id) -> int {   // id: internal node (< n - 1) or leaf (n - 1 + sorted position); returns its number
        float4 l, h;
        const bool leaf = id >= n - 1;
        if (leaf) { const int p = vals[id - (n - 1)]; l = prim_lo[p]; h = prim_hi[p]; }
        else { l = box_lo[id]; h = box_hi[id]; }
        lo[slot][0] = l.x; lo[slot][1] = l.y; lo[slot][2] = l.z;
        hi[slot][0] = h.x; hi[slot][1] = h.y; hi[slot][2] = h.z;
        kind[slot] = leaf ? 2 : 1;
        return leaf ? id - (n - 1) : id;
    };
    const int2 k = kids[i];
    const int child[2] = {k.x, k.y};
    for (int c = 0; c < 2; ++c) {
        if (child[c] >= n - 1) base[c] = put(2 * c, child[c]);
        else { const int2 g = kids[child[c]]; base[c] = put(2 * c, g.x); put(2 * c + 1, g.y); }   // g.y's number is g.x's + 1
    }
    float org[3];
    uint32_t eb[3], qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) {
        float o = __builtin_inff(), top = -__builtin_inff();
        for (int s = 0; s < 4; ++s) if (kind[s]) { o = fminf(o, lo[s][a]); top = fmaxf(top, hi[s][a]); }
        const float ext = top - o;
        int e = -126;
        if (ext > 0.0f) frexpf(ext * (1.0f / 254.0f), &e);   // ext / 254 = m 2^e, m in [0.5, 1): 254 steps of 2^e cover the extent
        e = e < -126 ? -126 : (e > 126 ? 126 : e);
        const float step = __int_as_float((e + 127) << 23), inv_step = __int_as_float((127 - e) << 23);
        org[a] = o; eb[a] = (uint32_t)(e + 127);
        for (int s = 0; s < 4; ++s) {
            if (!kind[s]) continue;
            int ql = (int)floorf((lo[s][a] - o) * inv_step), qh = (int)ceilf((hi[s][a] - o) * inv_step);
            ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
            while (ql > 0 && fmaf((float)ql, step, o) > lo[s][a]) --ql;     // the subtraction above was rounded
            while (qh < 255 && fmaf((float)qh, step, o) < hi[s][a]) ++qh;
            qlo[a] |= (uint32_t)ql << (8 * s); qhi[a] |= (uint32_t)qh << (8 * s);
        }
    }
    uint32_t used = 0, leaf = 0;
    for (int s = 0; s < 4; ++s) { used |= (uint32_t)(kind[s] != 0) << s; leaf |= (uint32_t)(kind[s] == 2) << s; }
    const uint32_t meta = eb[0] | eb[1] << 8 | eb[2] << 16 | used << 24 | leaf << 28;
    float4* out = nodes + 4 * (size_t)i;
    out[0] = make_float4(org[0], org[1], org[2], __uint_as_float(meta));
    out[1] = make_float4(__uint_as_float(qlo[0]), __uint_as_float(qhi[0]), __uint_as_float(qlo[1]), __uint_as_float(qhi[1]));
    out[2] = make_float4(__uint_as_float(qlo[2]), __uint_as_float(qhi[2]), __int_as_float(base[0]), __int_as_float(base[1]));
    out[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Leaf records in sorted order, 64 bytes each, so that a leaf's number is all a link has to carry:
//   triangle  (n.xyz, -n.v0), (a.xyz, aw), (b.xyz, bw), (primitive, 0, -, index in the scene's object / face order)
//   sphere    (centre, r^2), (is_light, -, -, -), -, (primitive, 1, -, 0x40000000 + sphere index)
// primitive p < n_tris: triangle p in device order; otherwise sphere p - n_tris.  All integers as bit patterns.
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_leaves(const int* __restrict__ vals, int n, int n_tris, const float4* __restrict__ tri_ab,
                                                            const DevSphere* __restrict__ spheres, float4* __restrict__ leaves) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const int p = vals[g];
    float4* out = leaves + 4 * (size_t)g;
    if (p < n_tris) {
        out[0] = tri_ab[4 * (size_t)p]; out[1] = tri_ab[4 * (size_t)p + 1]; out[2] = tri_ab[4 * (size_t)p + 2];
        out[3] = make_float4(__int_as_float(p), __int_as_float(0), 0.f, tri_ab[4 * (size_t)p + 3].w);
    } else {
        const DevSphere sp = spheres[p - n_tris];
        out[0] = make_float4(sp.cx, sp.cy, sp.cz, sp.r2);
        out[1] = make_float4(__int_as_float(sp.is_light), 0.f, 0.f, 0.f);
        out[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        out[3] = make_float4(__int_as_float(p), __int_as_float(1), 0.f, __int_as_float(0x40000000 + (p - n_tris)));
    }
}

// Boxes of the inner nodes from the boxes of the primitives.  `learn`: the tree is new -- rounds are launched in batches
// until the root is done (its flag then holds the tree's height, kept in w.rounds; waits for the stream); a tree higher
// than LBVH_MAX_ROUNDS gets the one-launch climbing kernel, now and at every refit.  Otherwise: w.rounds launches, not
// waited for.
#define LBVH_MAX_ROUNDS 128
#define LBVH_ROUNDS_MIN_PRIMS 16384
static hipError_t lbvh_node_boxes(hipStream_t st, LbvhWork& w, const int* vals, int n, bool learn) {
    auto blocks = [](int k) { return dim3((unsigned)((k + LBVH_BLOCK - 1) / LBVH_BLOCK)); };
    hipError_t e = hipMemsetAsync(w.flags, 0, sizeof(int) * (n - 1), st);
    if (e != hipSuccess) return e;
    auto round = [&](int r) {
        hipLaunchKernelGGL(k_lbvh_refit_round, blocks(n - 1), dim3(LBVH_BLOCK), 0, st, vals, n, w.prim_lo, w.prim_hi, w.kids, w.box_lo, w.box_hi,
                           w.flags, r);
    };
    // (few primitives: the one climbing launch is done before a tree's twenty-odd levels could be launched)
    if (learn && n >= LBVH_ROUNDS_MIN_PRIMS && !getenv("T3D_LBVH_CLIMB")) {
        w.rounds = -1;
        for (int r = 0; r < LBVH_MAX_ROUNDS && w.rounds < 0;) {
            for (int q = 0; q < 16; ++q) round(++r);
            int root = 0;
            if ((e = hipMemcpyAsync(&root, w.flags, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
            if (root > 0) w.rounds = root;
        }
        if (w.rounds > 0) return hipGetLastError();
        if ((e = hipMemsetAsync(w.flags, 0, sizeof(int) * (n - 1), st)) != hipSuccess) return e;
    } else if (learn) {
        w.rounds = -1;
    }
    if (w.rounds > 0) {
        for (int r = 1; r <= w.rounds; ++r) round(r);
    } else {
        hipLaunchKernelGGL(k_lbvh_refit, blocks(n), dim3(LBVH_BLOCK), 0, st, vals, n, w.prim_lo, w.prim_hi, w.kids, w.parent, w.box_lo, w.box_hi,
                           w.flags);
    }
    return hipGetLastError();
}

// Build the hierarchy over n_tris triangles (vertices in device order, 9 floats each; d_tri_ab their 64-byte
// intersection records in the same order) and n_spheres spheres; *out_nodes receives n - 1 traversal nodes and
// *out_leaves n leaf records of 64 bytes each (hipMalloc, owned by the caller).  Needs n >= 2.
// `keep`: where the build's working arrays go instead of being released, for lbvh_refit.
// host: a hierarchy to use instead of the Morton-order one (null: build that one here)
static hipError_t lbvh_build(hipStream_t st, const float* d_verts, const float4* d_tri_ab, int n_tris, const DevSphere* d_spheres,
                             int n_spheres, float4** out_nodes, float4** out_leaves, LbvhWork* keep = nullptr, const HostTree* host = nullptr) {
    const int n = n_tris + n_spheres;
    LbvhWork w;
    hipError_t e = hipSuccess;
    float4* nodes = nullptr;
    float4* leaves = nullptr;
    const int nt = n < 16384 * 16 ? (n + 15) / 16 : 16384;  // sort threads
    const int chunk = (n + nt - 1) / nt;
    auto blocks = [](int k) { return dim3((unsigned)((k + LBVH_BLOCK - 1) / LBVH_BLOCK)); };
#define LB(x) do { e = (x); if (e != hipSuccess) { w.release(); hipFree(nodes); hipFree(leaves); return e; } } while (0)
    LB(hipMalloc(&w.prim_lo, sizeof(float4) * n)); LB(hipMalloc(&w.prim_hi, sizeof(float4) * n));
    for (int k = 0; k < 2; ++k) { LB(hipMalloc(&w.keys[k], sizeof(uint32_t) * n)); LB(hipMalloc(&w.vals[k], sizeof(int) * n)); }
    LB(hipMalloc(&w.hist, sizeof(int) * 256 * (size_t)nt));
    LB(hipMalloc(&w.sums, sizeof(int) * 1024));
    LB(hipMalloc(&w.parent, sizeof(int) * (2 * (size_t)n - 1))); LB(hipMalloc(&w.kids, sizeof(int2) * (n - 1)));
    LB(hipMalloc(&w.box_lo, sizeof(float4) * (n - 1))); LB(hipMalloc(&w.box_hi, sizeof(float4) * (n - 1)));
    LB(hipMalloc(&w.flags, sizeof(int) * (n - 1))); LB(hipMalloc(&w.bounds, sizeof(int) * 6));
    LB(hipMalloc(&nodes, sizeof(float4) * 4 * (size_t)(n - 1)));
    LB(hipMalloc(&leaves, sizeof(float4) * 4 * (size_t)n));
    const int init_bounds[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    LB(hipMemcpyAsync(w.bounds, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, st));
    LB(hipStreamSynchronize(st));  // init_bounds is a stack array
    int nb = (n + LBVH_BLOCK - 1) / LBVH_BLOCK;
    hipLaunchKernelGGL(k_lbvh_bounds, dim3((unsigned)(nb > 1024 ? 1024 : nb)), dim3(LBVH_BLOCK), 0, st, d_verts, n_tris, d_spheres, n_spheres,
                       w.prim_lo, w.prim_hi, w.bounds);
    int cur = 0;
    if (host) {
        LB(hipMemcpyAsync(w.vals[0], host->order.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
        LB(hipMemcpyAsync(w.kids, host->kids.data(), sizeof(HostTreeKids) * (n - 1), hipMemcpyHostToDevice, st));
        LB(hipMemcpyAsync(w.parent, host->parent.data(), sizeof(int) * (2 * (size_t)n - 1), hipMemcpyHostToDevice, st));
        LB(hipStreamSynchronize(st));   // the caller's vectors may go
    } else
    hipLaunchKernelGGL(k_lbvh_morton, blocks(n), dim3(LBVH_BLOCK), 0, st, w.prim_lo, w.prim_hi, n, w.bounds, w.keys[0], w.vals[0]);
    for (int shift = 0; !host && shift < 32; shift += 8) {
        LB(hipMemsetAsync(w.hist, 0, sizeof(int) * 256 * (size_t)nt, st));
        hipLaunchKernelGGL(k_lbvh_rs_count, blocks(nt), dim3(LBVH_BLOCK), 0, st, w.keys[cur], n, shift, chunk, nt, w.hist);
        const int m = 256 * nt, n_tiles = (m + LBVH_SCAN_TILE - 1) / LBVH_SCAN_TILE;   // nt <= 16384: at most 1024 tiles
        hipLaunchKernelGGL(k_lbvh_scan_tiles, dim3((unsigned)n_tiles), dim3(1024), 0, st, w.hist, m, w.sums);
        hipLaunchKernelGGL(k_lbvh_scan_sums, dim3(1), dim3(1024), 0, st, w.sums, n_tiles);
        hipLaunchKernelGGL(k_lbvh_scan_add, dim3((unsigned)n_tiles), dim3(1024), 0, st, w.hist, m, w.sums);
        hipLaunchKernelGGL(k_lbvh_rs_scatter, blocks(nt), dim3(LBVH_BLOCK), 0, st, w.keys[cur], w.vals[cur], n, shift, chunk, nt, w.hist,
                           w.keys[cur ^ 1], w.vals[cur ^ 1]);
        cur ^= 1;
    }
    if (!host) hipLaunchKernelGGL(k_lbvh_hierarchy, blocks(n - 1), dim3(LBVH_BLOCK), 0, st, w.keys[cur], n, w.kids, w.parent);
    LB(lbvh_node_boxes(st, w, w.vals[cur], n, true));
    hipLaunchKernelGGL(k_lbvh_pack, blocks(n - 1), dim3(LBVH_BLOCK), 0, st, w.vals[cur], n, w.prim_lo, w.prim_hi, w.kids, w.box_lo, w.box_hi,
                       nodes);
    hipLaunchKernelGGL(k_lbvh_leaves, blocks(n), dim3(LBVH_BLOCK), 0, st, w.vals[cur], n, n_tris, d_tri_ab, d_spheres, leaves);
    LB(hipGetLastError());
    LB(hipStreamSynchronize(st));
#undef LB
    w.sorted = cur;
    if (keep) *keep = w; else w.release();
    *out_nodes = nodes;
    *out_leaves = leaves;
    return hipSuccess;
}

// The same hierarchy after primitives have moved (spheres, Lights: the triangles are the caller's promise not to have):
// boxes of the primitives again, boxes of the nodes bottom-up, records packed again.  The tree keeps its shape -- a sphere
// that has travelled far sits in a larger box than a new build would give it, which costs visits, never hits.
// new_tree: order, links and parents have been replaced (a hierarchy adopted from the host): its height is learnt again,
// which waits for the stream.
static hipError_t lbvh_refit(hipStream_t st, LbvhWork& w, const float* d_verts, const float4* d_tri_ab, int n_tris, const DevSphere* d_spheres,
                             int n_spheres, float4* nodes, float4* leaves, bool new_tree = false) {
    const int n = n_tris + n_spheres;
    auto blocks = [](int k) { return dim3((unsigned)((k + LBVH_BLOCK - 1) / LBVH_BLOCK)); };
    // the boxes of the spheres and Lights again; those of the triangles are the build's (the scene bounds, which only the
    // Morton codes of a build use, take the new boxes in as well: harmless)
    if (n_spheres > 0) {
        const int nb = (n_spheres + LBVH_BLOCK - 1) / LBVH_BLOCK;
        hipLaunchKernelGGL(k_lbvh_bounds, dim3((unsigned)(nb > 1024 ? 1024 : nb)), dim3(LBVH_BLOCK), 0, st, d_verts, n_tris, d_spheres, n_spheres,
                           w.prim_lo, w.prim_hi, w.bounds, n_tris);
    }
    hipError_t e = lbvh_node_boxes(st, w, w.vals[w.sorted], n, new_tree);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_lbvh_pack, blocks(n - 1), dim3(LBVH_BLOCK), 0, st, w.vals[w.sorted], n, w.prim_lo, w.prim_hi, w.kids, w.box_lo, w.box_hi,
                       nodes);
    hipLaunchKernelGGL(k_lbvh_leaves, blocks(n), dim3(LBVH_BLOCK), 0, st, w.vals[w.sorted], n, n_tris, d_tri_ab, d_spheres, leaves);
    return hipGetLastError();
}
