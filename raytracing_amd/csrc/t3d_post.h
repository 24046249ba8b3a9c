// t3d_post.h -- the image filters of the reference's Post class (T3DR.py:1600-1647) on 8-bit images that are already in
// device memory: a frame can be filtered where t3d_render_rows_device left it, before it crosses the PCIe link.  Byte
// work, bit-exact against the reference; bound by HBM (upscale) and by LDS reads (median).
//
// Included by t3d_hip.hip (one translation unit, one shared library).
#pragma once

#define POST_TILE 16          // output pixels per block edge of the median filter
#define POST_MAX_PAD 7        // kernelSize up to 15: (16 + 14)^2 x 3 bytes of LDS per block

// Post.upscale (T3DR.py:1600-1614): every pixel becomes a factor x factor block.  grid.y is the source row; a thread
// puts together VEC consecutive bytes of the widened row, walking (channel, copy of the pixel, source pixel) along
// them so the two divisions that place its first byte are shared by all of them, and stores them to the `factor`
// output rows that row becomes.  VEC is the widest of 16 / 4 / 1 that divides the output row's length with the output
// base aligned to it; the re-reads of the source bytes hit L1, HBM sees the input once and the output once.
template <int VEC>
__global__ __launch_bounds__(256) void k_post_upscale(const uint8_t* __restrict__ in, int width, int channels, int factor,
                                                      unsigned row_bytes, uint8_t* __restrict__ out) {
    const unsigned b0 = (blockIdx.x * 256u + threadIdx.x) * VEC;
    if (b0 >= row_bytes) return;
    const unsigned iy = blockIdx.y;
    const unsigned ox = b0 / (unsigned)channels, ix = ox / (unsigned)factor;
    int c = (int)(b0 - ox * channels), copy = (int)(ox - ix * factor);
    const uint8_t* sp = in + ((size_t)iy * width + ix) * channels;
    uint8_t* dst = out + (size_t)iy * factor * row_bytes + b0;
    uint32_t word[VEC >= 4 ? VEC / 4 : 1] = {};
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        word[j >> 2] |= (uint32_t)sp[c] << (8 * (j & 3));
        if (++c == channels) { c = 0; if (++copy == factor) { copy = 0; sp += channels; } }
    }
    for (int r = 0; r < factor; ++r, dst += row_bytes) {
        if constexpr (VEC == 16) *reinterpret_cast<uint4*>(dst) = make_uint4(word[0], word[1], word[2], word[3]);
        else if constexpr (VEC == 4) *reinterpret_cast<uint32_t*>(dst) = word[0];
        else *dst = (uint8_t)word[0];
    }
}

__device__ __forceinline__ int post_min3(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int post_max3(int a, int b, int c) { return max(a, max(b, c)); }
__device__ __forceinline__ int post_med3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// the m-th smallest (from 0) of v[0..N): res grows by every bit t for which at most m values are below res | t
template <int N>
__device__ __forceinline__ int post_select(const int (&v)[N], int m) {
    int res = 0;
#pragma unroll
    for (int bit = 128; bit > 0; bit >>= 1) {
        const int t = res | bit;
        int below = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) below += v[i] < t;
        if (below <= m) res = t;
    }
    return res;
}

// Post.denoise (T3DR.py:1616-1647): per channel the median of the (2 pad + 1)^2 window around a pixel, pad =
// kernelSize // 2; pixels within pad of the image border are (0, 0, 0).  The median is sorted(values)[n // 2], the true
// median of the odd-sized window.  A block loads its tile and halo into LDS once.  PAD 1: the three columns of the
// window are sorted (min3 / med3 / max3) and the median of nine is med3(max of the lows, med3 of the middles, min of
// the highs).  PAD 2, 3: the window goes to registers once and the median is found bit by bit (eight counting passes).
// PAD 4 .. 7: the same selection, counting over the window in LDS (row offsets are immediates of the reads).
template <int PAD>
__global__ __launch_bounds__(POST_TILE * POST_TILE) void k_post_denoise(const uint8_t* __restrict__ in, int height, int width,
                                                                        uint8_t* __restrict__ out) {
    __shared__ uint8_t tile[3][POST_TILE + 2 * POST_MAX_PAD][POST_TILE + 2 * POST_MAX_PAD + 2];
    constexpr int pad = PAD;
    const int x0 = blockIdx.x * POST_TILE, y0 = blockIdx.y * POST_TILE;
    const int span = POST_TILE + 2 * pad;
    for (int k = threadIdx.y * POST_TILE + threadIdx.x; k < span * span; k += POST_TILE * POST_TILE) {
        const int ty = k / span, tx = k - ty * span;
        const int gy = y0 + ty - pad, gx = x0 + tx - pad;
        uint8_t b = 0, g = 0, r = 0;
        if (gy >= 0 && gy < height && gx >= 0 && gx < width) {
            const uint8_t* p = in + ((size_t)gy * width + gx) * 3;
            b = p[0]; g = p[1]; r = p[2];
        }
        tile[0][ty][tx] = b; tile[1][ty][tx] = g; tile[2][ty][tx] = r;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= width || y >= height) return;
    uint8_t* dst = out + ((size_t)y * width + x) * 3;
    if (y < pad || y >= height - pad || x < pad || x >= width - pad) { dst[0] = dst[1] = dst[2] = 0; return; }
    const int k = 2 * pad + 1, m = (k * k) / 2;   // index of the median in sorted order
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int res = 0;
        if constexpr (PAD == 1) {
            int lo[3], mid[3], hi[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int a = tile[c][threadIdx.y][threadIdx.x + dx], b = tile[c][threadIdx.y + 1][threadIdx.x + dx],
                          d = tile[c][threadIdx.y + 2][threadIdx.x + dx];
                lo[dx] = post_min3(a, b, d); mid[dx] = post_med3(a, b, d); hi[dx] = post_max3(a, b, d);
            }
            res = post_med3(post_max3(lo[0], lo[1], lo[2]), post_med3(mid[0], mid[1], mid[2]), post_min3(hi[0], hi[1], hi[2]));
        } else if constexpr (PAD <= 3) {
            constexpr int K = 2 * PAD + 1;
            int v[K * K];
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int dx = 0; dx < K; ++dx) v[dy * K + dx] = tile[c][threadIdx.y + dy][threadIdx.x + dx];
            res = post_select<K * K>(v, m);
        } else {
            constexpr int K = 2 * PAD + 1;
            const uint8_t* w = &tile[c][threadIdx.y][threadIdx.x];
            constexpr int PITCH = POST_TILE + 2 * POST_MAX_PAD + 2;
            for (int bit = 128; bit > 0; bit >>= 1) {
                const int t = res | bit;
                int below = 0;
                for (int dy = 0; dy < K; ++dy) {
#pragma unroll
                    for (int dx = 0; dx < K; ++dx) below += w[dy * PITCH + dx] < t;
                }
                if (below <= m) res = t;   // at most m values are smaller: the m-th smallest is at least t
            }
        }
        dst[c] = (uint8_t)res;
    }
}
