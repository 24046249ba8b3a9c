// host_tree_check.cpp -- builds the host-side hierarchy (raytracing_amd/csrc/t3d_host_tree.h) over generated boxes and
// checks that it is a tree in the device's numbering: every primitive at exactly one leaf position, node ranges nested,
// child ids at the cut, parents consistent.  Prints "ok <case> n=<n> <milliseconds> cost <cost> depth <depth> tree <hash of order and children>" per case; exit status 1 on a violation.
//   g++ -O2 -std=c++17 -o host_tree_check tools/host_tree_check.cpp && ./host_tree_check [n ...]
#include "../raytracing_amd/csrc/t3d_host_tree.h"
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd() {   // xorshift, [0, 1)
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) * (1.0 / 16777216.0));
}

static bool check(const HostTree& t, int n, const char* what) {
    auto fail = [&](const char* why, int at) { fprintf(stderr, "%s n=%d: %s at %d\n", what, n, why, at); return false; };
    if ((int)t.order.size() != n) return fail("order size", 0);
    std::vector<char> seen(n, 0);
    for (int i = 0; i < n; ++i) {
        if (t.order[i] < 0 || t.order[i] >= n || seen[t.order[i]]) return fail("order is no permutation", i);
        seen[t.order[i]] = 1;
    }
    if (n < 2) return true;
    if (t.parent[0] != -1) return fail("root has a parent", 0);
    struct Span { int node, first, last; };
    std::vector<Span> todo{{0, 0, n - 1}};
    std::vector<char> visited(n - 1, 0);
    int inner = 0, leaves = 0;
    while (!todo.empty()) {
        const Span s = todo.back();
        todo.pop_back();
        if (s.node < 0 || s.node >= n - 1 || visited[s.node]) return fail("node id", s.node);
        if (s.node != s.first && s.node != s.last) return fail("node id not at an end of its range", s.node);
        visited[s.node] = 1; ++inner;
        const int l = t.kids[s.node].left, r = t.kids[s.node].right;
        const int g = l >= n - 1 ? l - (n - 1) : l;
        if (g < s.first || g >= s.last) return fail("cut outside the range", s.node);
        if (l >= n - 1 && g != s.first) return fail("left leaf is not the range's first position", s.node);
        if (r >= n - 1 ? r - (n - 1) != g + 1 || g + 1 != s.last : r != g + 1) return fail("right child is not behind the cut", s.node);
        if (t.parent[l] != s.node || t.parent[r] != s.node) return fail("parent", s.node);
        if (l >= n - 1) ++leaves; else todo.push_back({l, s.first, g});
        if (r >= n - 1) ++leaves; else todo.push_back({r, g + 1, s.last});
    }
    if (inner != n - 1 || leaves != n) return fail("node count", inner);
    return true;
}

// sum over internal nodes of area(node) -- proportional to the expected number of nodes a random ray enters
static double tree_cost(const HostTree& t, const std::vector<float>& lo, const std::vector<float>& hi, int n) {
    using host_tree_detail::Box;
    if (n < 2) return 0;
    std::vector<Box> box(2 * (size_t)n - 1, Box::none());
    std::vector<int> pending(n - 1, 0);
    double cost = 0;
    for (int g = 0; g < n; ++g) {
        int node = (n - 1) + g;
        box[node].grow(&lo[3 * (size_t)t.order[g]], &hi[3 * (size_t)t.order[g]]);
        for (int p = t.parent[node]; p >= 0; p = t.parent[node]) {
            box[p].join(box[node]);
            if (++pending[p] < 2) break;
            cost += box[p].area();
            node = p;
        }
    }
    return cost / box[0].area();
}

int main(int argc, char** argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {0, 1, 2, 3, 5, 64, 1000, 20000};
    bool ok = true;
    for (int n : sizes)
        for (int kind = 0; kind < 5; ++kind) {
            const char* names[5] = {"soup", "room+mesh", "coincident", "line of duplicates", "graded sizes"};
            std::vector<float> lo(3 * (size_t)n + 3), hi(3 * (size_t)n + 3);
            for (int i = 0; i < n; ++i) {
                float c[3], e[3];
                if (kind == 0) for (int a = 0; a < 3; ++a) { c[a] = 20 * rnd() - 10; e[a] = 0.2f * rnd(); }
                else if (kind == 1) { const bool wall = i < 12; for (int a = 0; a < 3; ++a) { c[a] = wall ? 0.0f : rnd() - 0.5f; e[a] = wall ? (a == i % 3 ? 0.0f : 5.0f) : 0.02f * rnd(); } if (wall) c[i % 3] = i % 2 ? 5.0f : -5.0f; }
                else if (kind == 2) for (int a = 0; a < 3; ++a) { c[a] = 1.5f; e[a] = 0.5f; }
                else if (kind == 3) { c[0] = (float)(i / 3); c[1] = c[2] = 0; e[0] = e[1] = e[2] = 0.25f; }
                else { c[0] = c[1] = c[2] = 100.0f * powf(15.0f / 16.0f, (float)(i % 1200)); e[0] = e[1] = e[2] = 0.4f * c[0]; }   // peels one box per level
                for (int a = 0; a < 3; ++a) { lo[3 * (size_t)i + a] = c[a] - e[a]; hi[3 * (size_t)i + a] = c[a] + e[a]; }
            }
            const auto t0 = std::chrono::steady_clock::now();
            const HostTree t = host_tree_build(lo.data(), hi.data(), n);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const bool good = check(t, n, names[kind]);
            ok &= good;
            uint64_t h = 1469598103934665603ull;   // FNV-1a over the leaf order and the children of every node
            auto mix = [&](int v) { h = (h ^ (uint32_t)v) * 1099511628211ull; };
            for (int v : t.order) mix(v);
            for (const HostTreeKids& k : t.kids) { mix(k.left); mix(k.right); }
            printf("%s %-18s n=%d %.3f ms cost %.2f depth %d tree %016llx\n", good ? "ok" : "BAD", names[kind], n, ms, tree_cost(t, lo, hi, n), t.depth,
                   (unsigned long long)h);
        }
    return ok ? 0 : 1;
}
