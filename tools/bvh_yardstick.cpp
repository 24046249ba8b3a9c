// bvh_yardstick.cpp -- how many hierarchy records a closest-hit query of a large scene has to visit, on the host, for
// several ways of building the hierarchy: a yardstick for the device-built LBVH of raytracing_amd/csrc/t3d_lbvh.h.
//
//   builders   lbvh   30-bit Morton codes of the box centres, sorted, split at the highest differing bit (the tree
//                     Karras' construction gives, equal codes told apart by position) -- what the device builds
//              sah    top-down binned surface-area heuristic (16 bins per axis), one primitive per leaf
//   collapse   grand  a 4-wide record holds the four grandchildren of a binary node (a child that is a leaf takes one
//                     slot) -- what the device packs
//              area   start from the two children and keep replacing the slot of largest surface area by its two
//                     children until there are four
//   traversal  the device's: the slots of a record are slab-tested against [0, best], entered slots are visited nearest
//              first, leaves are tested, postponed nodes are NOT re-tested against the best distance when popped
//              (also counted: with that re-test)
//
// The queries are those of a small path tracer over the same scene and camera (every `step`-th pixel of the frame;
// tree rays to depth `bounces`, shadow rays towards every light the surface faces): close to, not identical with,
// the queries of the device's frame; the device's own count for its own hierarchy calibrates the comparison.
//
// usage: bvh_yardstick scene.bin        (scene.bin written by tools/bvh_yardstick.py)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct Box {
    float lo[3], hi[3];
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const { const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2]; return 2.0f * (x * y + y * z + z * x); }
};
static Box empty_box() { Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; } return b; }

struct Scene {
    int T = 0, S = 0, L = 0;
    std::vector<float> tri, sph, lights;   // 9 per triangle; (cx, cy, cz, r) per sphere; 3 per light
    std::vector<float> rough, refl;        // per primitive
    float cam[3], zlen;
    int W, H, step, bounces;
    std::vector<Box> box;                  // per primitive, padded like the device's
};

struct Node2 { Box b; int left, right; };   // child >= 0: internal node; < 0: leaf ~prim
struct Bvh2 { std::vector<Node2> nodes; int root = 0; };

static uint32_t spread10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static Bvh2 build_lbvh(const Scene& sc) {
    const int n = sc.T + sc.S;
    Box all = empty_box();
    for (const Box& b : sc.box) all.grow(b);
    std::vector<uint64_t> key(n);
    for (int p = 0; p < n; ++p) {
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * (sc.box[p].lo[a] + sc.box[p].hi[a]), ext = all.hi[a] - all.lo[a];
            float x = ext > 0 ? (c - all.lo[a]) / ext : 0.0f;
            x = std::min(std::max(x * 1024.0f, 0.0f), 1023.0f);
            q[a] = (uint32_t)x;
        }
        key[p] = ((uint64_t)((spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2])) << 32) | (uint32_t)p;
    }
    std::sort(key.begin(), key.end());   // stable in the primitive index, like the device's radix sort
    std::vector<int> prim(n);
    for (int i = 0; i < n; ++i) { prim[i] = (int)(key[i] & 0xffffffffu); key[i] = (key[i] & 0xffffffff00000000ull) | (uint32_t)i; }   // ties: by position
    Bvh2 t;
    t.nodes.reserve(n);
    std::function<int(int, int)> rec = [&](int lo, int hi) -> int {
        if (lo == hi) return ~prim[lo];
        const uint64_t diff = key[lo] ^ key[hi];
        const int bit = 63 - __builtin_clzll(diff);
        int a = lo, b = hi;   // first index whose key has the bit set
        while (a < b) { const int m = (a + b) / 2; if ((key[m] >> bit) & 1) b = m; else a = m + 1; }
        const int id = (int)t.nodes.size();
        t.nodes.push_back(Node2{});
        const int l = rec(lo, a - 1), r = rec(a, hi);
        Node2& nd = t.nodes[id];
        nd.left = l; nd.right = r;
        nd.b = l < 0 ? sc.box[~l] : t.nodes[l].b;
        nd.b.grow(r < 0 ? sc.box[~r] : t.nodes[r].b);
        return id;
    };
    t.root = rec(0, n - 1);
    return t;
}

static Bvh2 build_sah(const Scene& sc) {
    const int n = sc.T + sc.S;
    std::vector<int> prim(n);
    for (int i = 0; i < n; ++i) prim[i] = i;
    std::vector<float> cen(3 * (size_t)n);
    for (int p = 0; p < n; ++p) for (int a = 0; a < 3; ++a) cen[3 * (size_t)p + a] = 0.5f * (sc.box[p].lo[a] + sc.box[p].hi[a]);
    Bvh2 t;
    t.nodes.reserve(n);
    const int BINS = 16;
    std::function<int(int, int)> rec = [&](int lo, int hi) -> int {   // [lo, hi)
        if (hi - lo == 1) return ~prim[lo];
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = lo; i < hi; ++i) for (int a = 0; a < 3; ++a) { clo[a] = std::min(clo[a], cen[3 * (size_t)prim[i] + a]); chi[a] = std::max(chi[a], cen[3 * (size_t)prim[i] + a]); }
        int best_axis = -1, best_split = 0;
        float best_cost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            const float ext = chi[a] - clo[a];
            if (!(ext > 0)) continue;
            Box bb[BINS]; int cnt[BINS];
            for (int k = 0; k < BINS; ++k) { bb[k] = empty_box(); cnt[k] = 0; }
            const float scale = BINS / ext;
            for (int i = lo; i < hi; ++i) {
                int k = (int)((cen[3 * (size_t)prim[i] + a] - clo[a]) * scale);
                k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
                bb[k].grow(sc.box[prim[i]]); cnt[k]++;
            }
            float right_area[BINS]; int right_cnt[BINS];
            Box acc = empty_box(); int c = 0;
            for (int k = BINS - 1; k >= 1; --k) { acc.grow(bb[k]); c += cnt[k]; right_area[k] = c ? acc.area() : 0; right_cnt[k] = c; }
            acc = empty_box(); c = 0;
            for (int k = 0; k + 1 < BINS; ++k) {
                acc.grow(bb[k]); c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const float cost = acc.area() * c + right_area[k + 1] * right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k + 1; }
            }
        }
        int mid;
        if (best_axis < 0) mid = (lo + hi) / 2;   // all centres coincide
        else {
            const float ext = chi[best_axis] - clo[best_axis], scale = BINS / ext;
            mid = (int)(std::partition(prim.begin() + lo, prim.begin() + hi, [&](int p) {
                int k = (int)((cen[3 * (size_t)p + best_axis] - clo[best_axis]) * scale);
                k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
                return k < best_split;
            }) - prim.begin());
            if (mid == lo || mid == hi) mid = (lo + hi) / 2;
        }
        const int id = (int)t.nodes.size();
        t.nodes.push_back(Node2{});
        const int l = rec(lo, mid), r = rec(mid, hi);
        Node2& nd = t.nodes[id];
        nd.left = l; nd.right = r;
        nd.b = l < 0 ? sc.box[~l] : t.nodes[l].b;
        nd.b.grow(r < 0 ? sc.box[~r] : t.nodes[r].b);
        return id;
    };
    t.root = rec(0, n);
    return t;
}

struct Node4 { Box b[4]; int child[4]; int n; };   // child >= 0: record; < 0: leaf ~prim
struct Bvh4 { std::vector<Node4> nodes; };

static Bvh4 collapse(const Scene& sc, const Bvh2& t, bool by_area) {
    Bvh4 out;
    std::vector<int> where(t.nodes.size(), -1);
    std::vector<int> todo;
    auto box_of = [&](int id) { return id < 0 ? sc.box[~id] : t.nodes[id].b; };
    auto record_of = [&](int id2) {
        if (where[id2] < 0) { where[id2] = (int)out.nodes.size(); out.nodes.push_back(Node4{}); todo.push_back(id2); }
        return where[id2];
    };
    record_of(t.root);
    while (!todo.empty()) {
        const int id2 = todo.back(); todo.pop_back();
        int slots[4], ns = 0;
        slots[ns++] = t.nodes[id2].left; slots[ns++] = t.nodes[id2].right;
        if (by_area) {
            while (ns < 4) {
                int pick = -1; float best = -1;
                for (int k = 0; k < ns; ++k) if (slots[k] >= 0 && box_of(slots[k]).area() > best) { best = box_of(slots[k]).area(); pick = k; }
                if (pick < 0) break;
                const int id = slots[pick];
                slots[pick] = t.nodes[id].left; slots[ns++] = t.nodes[id].right;
            }
        } else {
            int g[4], ng = 0;
            for (int k = 0; k < 2; ++k) {
                if (slots[k] < 0) g[ng++] = slots[k];
                else { g[ng++] = t.nodes[slots[k]].left; g[ng++] = t.nodes[slots[k]].right; }
            }
            ns = ng; for (int k = 0; k < ng; ++k) slots[k] = g[k];
        }
        const int me = where[id2];
        Node4 nd{};
        nd.n = ns;
        for (int k = 0; k < ns; ++k) { nd.b[k] = box_of(slots[k]); nd.child[k] = slots[k] < 0 ? slots[k] : record_of(slots[k]); }
        out.nodes[me] = nd;
    }
    return out;
}

struct Ray { double o[3], d[3]; };
struct HitRec { double t; int prim; };
struct Count { uint64_t queries = 0, records = 0, records_recheck = 0, tri = 0, sph = 0; };

static bool hit_prim(const Scene& sc, int p, const Ray& r, double& t) {
    if (p < sc.T) {
        const float* v = &sc.tri[9 * (size_t)p];
        const double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
        const double pv[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
        const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
        if (std::fabs(det) < 1e-14) return false;
        const double inv = 1.0 / det, tv[3] = {r.o[0] - v[0], r.o[1] - v[1], r.o[2] - v[2]};
        const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
        if (u <= 0 || u >= 1) return false;
        const double qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
        const double w = (r.d[0] * qv[0] + r.d[1] * qv[1] + r.d[2] * qv[2]) * inv;
        if (w <= 0 || u + w >= 1) return false;
        t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
        return t >= 0;
    }
    const float* s = &sc.sph[4 * (size_t)(p - sc.T)];
    const double Lv[3] = {r.o[0] - s[0], r.o[1] - s[1], r.o[2] - s[2]};
    const double b = Lv[0] * r.d[0] + Lv[1] * r.d[1] + Lv[2] * r.d[2];
    const double disc = b * b - (Lv[0] * Lv[0] + Lv[1] * Lv[1] + Lv[2] * Lv[2] - (double)s[3] * s[3]);
    if (disc < 0) return false;
    const double root = std::sqrt(disc), t1 = -b - root, t2 = -b + root;
    if (t1 > 0) { t = t1; return true; }
    if (t2 > 0) { t = t2; return true; }
    return false;
}

static HitRec closest(const Scene& sc, const Bvh4& h, const Ray& r, Count& c) {
    double inv[3];
    for (int a = 0; a < 3; ++a) { inv[a] = 1.0 / r.d[a]; if (!std::isfinite(inv[a])) inv[a] = r.d[a] < 0 ? -1e30 : 1e30; }
    HitRec best{INFINITY, -1};
    struct Item { int node; double tn; };
    Item stack[256];
    int sp = 0;
    stack[sp++] = Item{0, 0.0};
    c.queries++;
    while (sp) {
        const Item it = stack[--sp];
        c.records++;
        if (it.tn <= best.t) c.records_recheck++;   // a traversal that re-tests postponed nodes when it pops them would stop here otherwise
        const Node4& nd = h.nodes[it.node];
        double tn[4]; int ord[4], ne = 0;
        for (int k = 0; k < nd.n; ++k) {
            double lo = 0, hi = best.t;
            for (int a = 0; a < 3; ++a) {
                double t0 = (nd.b[k].lo[a] - r.o[a]) * inv[a], t1 = (nd.b[k].hi[a] - r.o[a]) * inv[a];
                if (t0 > t1) std::swap(t0, t1);
                lo = std::max(lo, t0); hi = std::min(hi, t1);
            }
            if (lo <= hi * 1.0000004) { tn[k] = lo; ord[ne++] = k; }
        }
        std::sort(ord, ord + ne, [&](int a, int b) { return tn[a] < tn[b]; });
        for (int q = 0; q < ne; ++q) {   // leaves are tested (nearest first)
            const int k = ord[q];
            if (nd.child[k] >= 0) continue;
            const int p = ~nd.child[k];
            if (p < sc.T) c.tri++; else c.sph++;
            double t;
            if (hit_prim(sc, p, r, t) && t < best.t) best = HitRec{t, p};
        }
        for (int q = ne - 1; q >= 0; --q) {   // nodes: farthest pushed first
            const int k = ord[q];
            if (nd.child[k] >= 0 && sp < 255) stack[sp++] = Item{nd.child[k], tn[k]};
        }
    }
    return best;
}

static uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

static void normal_at(const Scene& sc, int p, const double* P, double* n) {
    if (p < sc.T) {
        const float* v = &sc.tri[9 * (size_t)p];
        const double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
        n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
    } else {
        const float* s = &sc.sph[4 * (size_t)(p - sc.T)];
        n[0] = P[0] - s[0]; n[1] = P[1] - s[1]; n[2] = P[2] - s[2];
    }
    const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int a = 0; a < 3; ++a) n[a] /= len > 0 ? len : 1;
}

// the queries of a small path tracer: tree rays to depth `bounces`, shadow rays towards the lights a surface faces
static void trace_frame(const Scene& sc, const Bvh4& h, Count& tree, Count& shadow) {
    std::vector<Ray> cur, next;
    std::vector<uint32_t> cur_key, next_key;
    const double ratio = (double)sc.W / sc.H;
    for (int py = 0; py < sc.H; py += sc.step)
        for (int px = 0; px < sc.W; px += sc.step) {
            Ray r;
            const double v[3] = {((double)px / sc.W) * ratio - ratio / 2, 0.5 - (double)py / sc.H, sc.zlen};
            const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            for (int a = 0; a < 3; ++a) { r.o[a] = sc.cam[a]; r.d[a] = v[a] / len; }
            cur.push_back(r); cur_key.push_back(mix32((uint32_t)(py * sc.W + px)));
        }
    for (int level = 0; level <= sc.bounces; ++level) {
        next.clear(); next_key.clear();
        for (size_t i = 0; i < cur.size(); ++i) {
            const Ray& r = cur[i];
            const HitRec hit = closest(sc, h, r, tree);
            if (hit.prim < 0) continue;
            double P[3], n[3];
            for (int a = 0; a < 3; ++a) P[a] = r.o[a] + hit.t * r.d[a];
            normal_at(sc, hit.prim, P, n);
            for (int l = 0; l < sc.L; ++l) {
                double to[3] = {sc.lights[3 * l] - P[0], sc.lights[3 * l + 1] - P[1], sc.lights[3 * l + 2] - P[2]};
                const double len = std::sqrt(to[0] * to[0] + to[1] * to[1] + to[2] * to[2]);
                if (!(to[0] * n[0] + to[1] * n[1] + to[2] * n[2] > 0) || sc.rough[hit.prim] == 0) continue;   // the light term is zero: not traced
                Ray s;
                for (int a = 0; a < 3; ++a) { s.d[a] = to[a] / len; s.o[a] = P[a] + 0.2 * s.d[a]; }
                closest(sc, h, s, shadow);
            }
            if (level == sc.bounces) continue;
            const double dn = r.d[0] * n[0] + r.d[1] * n[1] + r.d[2] * n[2];
            uint32_t k = cur_key[i];
            double diff[3];
            for (int a = 0; a < 3; ++a) { k = mix32(k + 0x9E3779B9u); diff[a] = (double)(k >> 8) / 8388608.0 - 1.0; }
            double dl = std::sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
            double sgn = (diff[0] * n[0] + diff[1] * n[1] + diff[2] * n[2]) < 0 ? -1 : 1;
            const double rough = sc.rough[hit.prim];
            Ray b;
            double bl = 0;
            for (int a = 0; a < 3; ++a) { b.d[a] = (r.d[a] - 2 * dn * n[a]) * (1 - rough) + sgn * diff[a] / dl * rough; bl += b.d[a] * b.d[a]; }
            bl = std::sqrt(bl);
            for (int a = 0; a < 3; ++a) { b.o[a] = P[a] + 0.1 * b.d[a]; b.d[a] /= bl; }
            next.push_back(b); next_key.push_back(mix32(k ^ 0x5bd1e995u));
            if (sc.refl[hit.prim] != 1.0f) {   // a transmitted ray too (straight on: the direction statistics are what matters here)
                Ray t2;
                for (int a = 0; a < 3; ++a) { t2.d[a] = r.d[a]; t2.o[a] = P[a] + 0.1 * r.d[a]; }
                next.push_back(t2); next_key.push_back(mix32(k ^ 0x1b873593u));
            }
        }
        cur.swap(next); cur_key.swap(next_key);
    }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s scene.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    Scene sc;
    int hdr[7];
    if (fread(hdr, sizeof(int), 7, f) != 7) return 1;
    sc.T = hdr[0]; sc.S = hdr[1]; sc.L = hdr[2]; sc.W = hdr[3]; sc.H = hdr[4]; sc.step = hdr[5]; sc.bounces = hdr[6];
    float camz[4];
    if (fread(camz, sizeof(float), 4, f) != 4) return 1;
    for (int a = 0; a < 3; ++a) sc.cam[a] = camz[a];
    sc.zlen = camz[3];
    const int n = sc.T + sc.S;
    sc.tri.resize(9 * (size_t)sc.T); sc.sph.resize(4 * (size_t)sc.S); sc.lights.resize(3 * (size_t)sc.L); sc.rough.resize(n); sc.refl.resize(n);
    if (fread(sc.tri.data(), sizeof(float), sc.tri.size(), f) != sc.tri.size() || fread(sc.sph.data(), sizeof(float), sc.sph.size(), f) != sc.sph.size() ||
        fread(sc.lights.data(), sizeof(float), sc.lights.size(), f) != sc.lights.size() || fread(sc.rough.data(), sizeof(float), n, f) != (size_t)n ||
        fread(sc.refl.data(), sizeof(float), n, f) != (size_t)n) { fprintf(stderr, "short file\n"); return 1; }
    fclose(f);
    sc.box.resize(n);
    for (int p = 0; p < n; ++p) {
        Box b = empty_box();
        if (p < sc.T) {
            const float* v = &sc.tri[9 * (size_t)p];
            for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], v[3 * k + a]); b.hi[a] = std::max(b.hi[a], v[3 * k + a]); }
        } else {
            const float* s = &sc.sph[4 * (size_t)(p - sc.T)];
            for (int a = 0; a < 3; ++a) { b.lo[a] = s[a] - s[3] * 1.000001f; b.hi[a] = s[a] + s[3] * 1.000001f; }
        }
        float ext = 0;
        for (int a = 0; a < 3; ++a) ext = std::max(ext, std::max(std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])), b.hi[a] - b.lo[a]));
        const float pad = 4e-6f * ext + 1e-30f;
        for (int a = 0; a < 3; ++a) { b.lo[a] -= pad; b.hi[a] += pad; }
        sc.box[p] = b;
    }
    printf("scene: %d triangles, %d spheres, %d lights; queries of every %d-th pixel of %d x %d, %d bounces\n", sc.T, sc.S, sc.L, sc.step, sc.W, sc.H, sc.bounces);
    struct Variant { const char* name; bool sah, by_area; };
    const Variant variants[] = {{"lbvh (Morton order), grandchildren  [the device's]", false, false}, {"lbvh (Morton order), slots by area", false, true},
                                {"binned SAH, grandchildren", true, false}, {"binned SAH, slots by area", true, true}};
    Bvh2 lbvh = build_lbvh(sc), sah = build_sah(sc);
    auto sah_cost = [&](const Bvh2& t) { double c = 0; for (const Node2& nd : t.nodes) c += nd.b.area(); return c / t.nodes[t.root].b.area(); };
    printf("surface-area cost of the binary trees (sum of node areas / root area): lbvh %.1f, sah %.1f\n", sah_cost(lbvh), sah_cost(sah));
    for (const Variant& v : variants) {
        const Bvh4 h = collapse(sc, v.sah ? sah : lbvh, v.by_area);
        Count tree, shadow;
        trace_frame(sc, h, tree, shadow);
        const uint64_t q = tree.queries + shadow.queries;
        printf("%-52s records %8zu | per query: records %6.2f (tree rays %6.2f, shadow rays %6.2f; with a re-test at pop %6.2f) triangle tests %5.2f sphere tests %5.2f | queries %llu (%llu shadow)\n",
               v.name, h.nodes.size(), (double)(tree.records + shadow.records) / q, (double)tree.records / tree.queries,
               shadow.queries ? (double)shadow.records / shadow.queries : 0.0, (double)(tree.records_recheck + shadow.records_recheck) / q,
               (double)(tree.tri + shadow.tri) / q, (double)(tree.sph + shadow.sph) / q, (unsigned long long)q, (unsigned long long)shadow.queries);
        fflush(stdout);
    }
    return 0;
}
