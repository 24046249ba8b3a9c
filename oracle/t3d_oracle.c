/*
 * t3d_oracle.c -- CPU restatement of the reference "Photon" ray tracer, float64, single thread
 * per call.  TEST INFRASTRUCTURE ONLY: it is the checker for the HIP library and the CPU
 * baseline leg of bench.py; nothing in the product path may call it.
 *
 * Every function follows one function of the reference (file:line in its comment) operation by
 * operation, in the same evaluation order, so that results are bit-identical to CPython's
 * float arithmetic on the same libm:  x**2 is pow(x, 2.0) as CPython's float_pow does, sums
 * associate left to right, and the file must be compiled with -ffp-contract=off.
 * Parity pinned by the golden fixtures in tests/golden (produced by running the reference).
 */
#include "t3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Vec3 (T3DE.py:464-577)
 * ---------------------------------------------------------------------------------------- */
typedef struct { double x, y, z; } V3;

static V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
static V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }            /* :470 */
static V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }            /* :473 */
static V3 vmulk(V3 a, double k) { return v3(a.x * k, a.y * k, a.z * k); }              /* :476 */
static V3 vdivk(V3 a, double k) { return v3(a.x / k, a.y / k, a.z / k); }              /* :484 */
static double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           /* :493 */
static double vmag(V3 a) { return sqrt(pow(a.x, 2.0) + pow(a.y, 2.0) + pow(a.z, 2.0)); } /* :515 */
static V3 vnorm(V3 a) { return vdivk(a, vmag(a)); }                                     /* :547 */
static V3 vcross(V3 a, V3 b) {                                                          /* :519 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static double vget(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* Vec3.flip(direction) T3DE.py:567-577 with component/project T3DE.py:501-512 */
static V3 vflip(V3 self, V3 direction) {
    V3 unit = vnorm(direction);                 /* :571 */
    /* project(unit): self.component(unit.normalise()) * unit.normalise()   :510 */
    V3 on = vnorm(unit);
    double comp = vdot(self, vnorm(on));        /* component(other) = self.dot(other.normalise()) :503 */
    V3 projection = vmulk(on, comp);            /* float * Vec3 via the __rmul__ shim = Vec3 * float */
    return vsub(self, vmulk(projection, 2.0));  /* :575 */
}

/* Euler.toMatrix T3DE.py:70-94 and Matrix9.__mul__ T3DE.py:20-43 */
static void mat_mul(const double* e, const double* o, double* r) {
    r[0] = e[0] * o[0] + e[1] * o[3] + e[2] * o[6];
    r[1] = e[0] * o[1] + e[1] * o[4] + e[2] * o[7];
    r[2] = e[0] * o[2] + e[1] * o[5] + e[2] * o[8];
    r[3] = e[3] * o[0] + e[4] * o[3] + e[5] * o[6];
    r[4] = e[3] * o[1] + e[4] * o[4] + e[5] * o[7];
    r[5] = e[3] * o[2] + e[4] * o[5] + e[5] * o[8];
    r[6] = e[6] * o[0] + e[7] * o[3] + e[8] * o[6];
    r[7] = e[6] * o[1] + e[7] * o[4] + e[8] * o[7];
    r[8] = e[6] * o[2] + e[7] * o[5] + e[8] * o[8];
}
void t3do_euler_matrix(const double* euler, double* m) {
    double cx = cos(euler[0]), sx = sin(euler[0]);
    double cy = cos(euler[1]), sy = sin(euler[1]);
    double cz = cos(euler[2]), sz = sin(euler[2]);
    double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double zy[9];
    mat_mul(Rz, Ry, zy);
    mat_mul(zy, Rx, m);
}
/* Vec3.applyTransforms T3DE.py:530-544: rotate, then scale, then translate */
void t3do_apply_transforms(const double* v, const double* pos, const double* scale,
                           const double* euler, double* out) {
    double m[9];
    t3do_euler_matrix(euler, m);
    double rx = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double ry = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double rz = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    out[0] = rx * scale[0] + pos[0];
    out[1] = ry * scale[1] + pos[1];
    out[2] = rz * scale[2] + pos[2];
}

/* ------------------------------------------------------------------------------------------
 * Random numbers
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } MT;

static void mt_init_genrand(MT* m, uint32_t s) {
    m->mt[0] = s;
    for (int i = 1; i < 624; i++) m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
/* CPython random.seed(int): init_by_array over the 32-bit digits of abs(seed) (_randommodule.c) */
static void mt_seed(MT* m, uint32_t seed) {
    uint32_t key[1] = {seed};
    int key_len = 1;
    mt_init_genrand(m, 19650218u);
    int i = 1, j = 0;
    for (int k = 624; k; k--) {
        m->mt[i] = (m->mt[i] ^ ((m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { m->mt[0] = m->mt[623]; i = 1; }
        if (j >= key_len) j = 0;
    }
    for (int k = 623; k; k--) {
        m->mt[i] = (m->mt[i] ^ ((m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { m->mt[0] = m->mt[623]; i = 1; }
    }
    m->mt[0] = 0x80000000u;
}
static uint32_t mt_u32(MT* m) {
    if (m->idx >= 624) {
        uint32_t* mt = m->mt;
        for (int k = 0; k < 624; k++) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        m->idx = 0;
    }
    uint32_t y = m->mt[m->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
static double mt_random(MT* m) { /* random.random(): 53-bit */
    uint32_t a = mt_u32(m) >> 5, b = mt_u32(m) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
uint32_t t3do_keyed_bits24(uint32_t seed, uint32_t px, uint32_t py, uint32_t sample, uint32_t heap,
                           uint32_t comp) {
    uint32_t h = mix32(seed + 0x9E3779B9u);
    h = mix32(h ^ px);
    h = mix32(h ^ py);
    h = mix32(h ^ (((sample & 0xFFFFu) << 16) | (heap & 0xFFFFu)));
    h = mix32(h ^ ((comp + 1u) * 0x9E3779B9u));
    return h >> 8;
}

/* ------------------------------------------------------------------------------------------
 * Scene, BVH (T3DE.py:879-902, 985-1022, 1074-1103)
 * ---------------------------------------------------------------------------------------- */
typedef struct BVH {
    V3 bmin, bmax;
    struct BVH *left, *right;
    int* faces;   /* leaf: global triangle indices in list order */
    int n_faces;  /* 0 for inner nodes */
} BVH;

typedef struct {
    const t3d_scene_desc* s;
    const t3d_opts* o;
    BVH** bvh;          /* per object */
    MT mt;
    /* key of the random draw under construction */
    uint32_t kx, ky, ks;
    t3d_stats st;
    /* optional node recorder */
    t3do_node* rec; int rec_cap; int rec_n; int rec_enabled;
    /* fragility of the primary hit under evaluation */
    double frag;
} Ctx;

static V3 tri_v(const t3d_scene_desc* s, int t, int k) {
    const double* p = s->tri_verts + 9 * (size_t)t + 3 * k;
    return v3(p[0], p[1], p[2]);
}

/* Object.boundingBox for a mesh face list (T3DE.py:1075-1094); transforms are identity after
 * setTransforms so the transformed vertices are the baked ones. */
static void mesh_box(const t3d_scene_desc* s, const int* faces, int n, V3* bmin, V3* bmax) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            V3 p = tri_v(s, faces[i], k);
            for (int a = 0; a < 3; a++) {
                double c = vget(p, a);
                if (c < mn[a]) mn[a] = c;
                if (c > mx[a]) mx[a] = c;
            }
        }
    *bmin = v3(mn[0] - 0.06, mn[1] - 0.06, mn[2] - 0.06);
    *bmax = v3(mx[0] + 0.06, mx[1] + 0.06, mx[2] + 0.06);
}

/* Mesh.centroid T3DE.py:766-774 */
static double centroid_axis(const t3d_scene_desc* s, int t, int axis) {
    V3 a = tri_v(s, t, 0), b = tri_v(s, t, 1), c = tri_v(s, t, 2);
    return (vget(a, axis) + vget(b, axis) + vget(c, axis)) / 3;
}

/* stable merge sort by key (list.sort is stable, T3DE.py:1008) */
static void stable_sort(int* idx, double* key, int n, int* tmp_i, double* tmp_k) {
    if (n < 2) return;
    int h = n / 2;
    stable_sort(idx, key, h, tmp_i, tmp_k);
    stable_sort(idx + h, key + h, n - h, tmp_i, tmp_k);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (key[j] < key[i]) { tmp_i[k] = idx[j]; tmp_k[k++] = key[j++]; }
        else { tmp_i[k] = idx[i]; tmp_k[k++] = key[i++]; }
    }
    while (i < h) { tmp_i[k] = idx[i]; tmp_k[k++] = key[i++]; }
    while (j < n) { tmp_i[k] = idx[j]; tmp_k[k++] = key[j++]; }
    memcpy(idx, tmp_i, sizeof(int) * n);
    memcpy(key, tmp_k, sizeof(double) * n);
}

/* Object.buildBVH T3DE.py:985-1022 (mesh case) */
static BVH* build_bvh(const t3d_scene_desc* s, int* faces, int n, int depth) {
    BVH* node = (BVH*)calloc(1, sizeof(BVH));
    V3 bmin, bmax;
    mesh_box(s, faces, n, &bmin, &bmax);
    int small = 0;
    for (int a = 0; a < 3; a++)
        if (fabs(vget(bmin, a) - vget(bmax, a)) <= 0.0006) small = 1;
    if (n <= 8 || small || depth >= 100) {
        node->bmin = bmin; node->bmax = bmax;
        node->faces = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
        memcpy(node->faces, faces, sizeof(int) * n);
        node->n_faces = n;
        return node;
    }
    int axis = depth % 3;
    double* key = (double*)malloc(sizeof(double) * n);
    double* tk = (double*)malloc(sizeof(double) * n);
    int* ti = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) key[i] = centroid_axis(s, faces[i], axis);
    stable_sort(faces, key, n, ti, tk);
    free(key); free(tk); free(ti);
    int mid = n / 2;
    node->left = build_bvh(s, faces, mid, depth + 1);
    node->right = build_bvh(s, faces + mid, n - mid, depth + 1);
    /* BVHNode.merge T3DE.py:888-902 */
    node->bmin = v3(fmin(node->left->bmin.x, node->right->bmin.x), fmin(node->left->bmin.y, node->right->bmin.y),
                    fmin(node->left->bmin.z, node->right->bmin.z));
    node->bmax = v3(fmax(node->left->bmax.x, node->right->bmax.x), fmax(node->left->bmax.y, node->right->bmax.y),
                    fmax(node->left->bmax.z, node->right->bmax.z));
    return node;
}
static void free_bvh(BVH* b) {
    if (!b) return;
    free_bvh(b->left); free_bvh(b->right); free(b->faces); free(b);
}

/* buildBVH for every object (T3DR.py:1104-1107): sphere/light boxes T3DE.py:1095-1099,
 * camera/empty boxes T3DE.py:1100-1101 (never consulted). */
static BVH** build_all(const t3d_scene_desc* s) {
    BVH** all = (BVH**)calloc(s->n_objects > 0 ? s->n_objects : 1, sizeof(BVH*));
    for (int o = 0; o < s->n_objects; o++) {
        int ty = s->obj_type[o];
        if (ty == T3D_OBJ_MESH) {
            int n = s->obj_tri_count[o];
            int* faces = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
            for (int i = 0; i < n; i++) faces[i] = s->obj_tri_begin[o] + i;
            all[o] = build_bvh(s, faces, n, 0);
            free(faces);
        } else {
            BVH* b = (BVH*)calloc(1, sizeof(BVH));
            V3 c = v3(s->obj_center[3 * o], s->obj_center[3 * o + 1], s->obj_center[3 * o + 2]);
            if (ty == T3D_OBJ_SPHERE || ty == T3D_OBJ_LIGHT) {
                double r = s->obj_radius[o];
                b->bmin = vsub(vsub(c, v3(r, r, r)), v3(0.06, 0.06, 0.06));
                b->bmax = vadd(vadd(c, v3(r, r, r)), v3(0.06, 0.06, 0.06));
            } else { b->bmin = c; b->bmax = c; }
            all[o] = b;
        }
    }
    return all;
}

/* ------------------------------------------------------------------------------------------
 * Ray (T3DE.py:125-208)
 * ---------------------------------------------------------------------------------------- */
typedef struct { V3 start, dir; double ior; int obj; } Ray;

static Ray make_ray(V3 start, V3 direction, double ior, int obj) { /* Ray.__init__ :126-130 */
    Ray r; r.start = start; r.dir = vnorm(direction); r.ior = ior; r.obj = obj; return r;
}

/* Ray.entersAABB T3DE.py:166-204 */
static int enters_aabb(Ctx* c, const Ray* ray, V3 bmin, V3 bmax) {
    if (c) c->st.aabb_tests++;
    double te[3], tx[3];
    for (int a = 0; a < 3; a++) {
        double oc = vget(ray->start, a), dc = vget(ray->dir, a);
        if (dc == 0) {
            if (oc < vget(bmin, a) || oc > vget(bmax, a)) return 0;
            te[a] = -INFINITY; tx[a] = INFINITY;
            continue;
        }
        double t0 = (vget(bmin, a) - oc) / dc;
        double t1 = (vget(bmax, a) - oc) / dc;
        if (t0 > t1) { double t = t0; t0 = t1; t1 = t; }
        te[a] = t0; tx[a] = t1;
    }
    /* max()/min() over a list keep the first of equal elements; values only matter */
    double tEnter = te[0]; if (te[1] > tEnter) tEnter = te[1]; if (te[2] > tEnter) tEnter = te[2];
    double tExit = tx[0]; if (tx[1] < tExit) tExit = tx[1]; if (tx[2] < tExit) tExit = tx[2];
    return tEnter <= tExit && tExit > 0;
}
int t3do_enters_aabb(const double* start, const double* dir, const double* bmin, const double* bmax) {
    Ray r = make_ray(v3(start[0], start[1], start[2]), v3(dir[0], dir[1], dir[2]), 1, -1);
    return enters_aabb(NULL, &r, v3(bmin[0], bmin[1], bmin[2]), v3(bmax[0], bmax[1], bmax[2]));
}

typedef struct { int hit; V3 point; double t; V3 normal; int direction; } Hit;

/* Photon.getSphereCollision T3DR.py:572-598 with Ray.getTSphere T3DE.py:145-163 */
static Hit sphere_collision(Ctx* c, const Ray* ray, V3 center, double radius, int is_light, int backCull) {
    Hit h; memset(&h, 0, sizeof h);
    V3 L = vsub(ray->start, center);
    double disc = pow(vdot(L, ray->dir), 2.0) - (vdot(L, L) - pow(radius, 2.0));
    if (c && radius > 0) {
        /* fragility (test aid): how far the ray passes from the sphere's silhouette, relative to the radius squared --
         * a ray that grazes the sphere may hit it in one precision and miss it in another, like a ray on a triangle's edge */
        double r = fabs(disc) / (radius * radius);
        if (r < c->frag) c->frag = r;
    }
    if (disc < 0) return h;
    double t1 = (vdot(L, ray->dir) * -1) - sqrt(disc);
    double t2 = (vdot(L, ray->dir) * -1) + sqrt(disc);
    double t; int direction;
    if (t1 > 0) { t = t1; direction = 0; }
    else if (t2 > 0 && !backCull && !is_light) { t = t2; direction = 1; }
    else return h;
    h.hit = 1; h.t = t; h.direction = direction;
    h.point = vadd(ray->start, vmulk(ray->dir, t));
    h.normal = vnorm(vsub(h.point, center));
    return h;
}

/* Photon.pointWithinTriangle T3DR.py:601-621; *margin gets min of the two products (test aid) */
static int point_within_triangle(V3 p, V3 a, V3 b, V3 cc, double* m1, double* m2) {
    V3 ab = vsub(b, a), bc = vsub(cc, b), ca = vsub(a, cc);
    V3 ap = vsub(p, a), bp = vsub(p, b), cp = vsub(p, cc);
    V3 crossA = vcross(ab, ap), crossB = vcross(bc, bp), crossC = vcross(ca, cp);
    double d1 = vdot(crossA, crossB), d2 = vdot(crossB, crossC);
    if (m1) *m1 = d1;
    if (m2) *m2 = d2;
    return (d1 > 0) && (d2 > 0);
}

/* Photon.getFaceCollision T3DR.py:624-651, MathematicalPlane T3DE.py:251-263, getTPlane :133-142 */
static Hit face_collision(Ctx* c, const Ray* ray, V3 a, V3 b, V3 cc) {
    Hit h; memset(&h, 0, sizeof h);
    if (c) c->st.tri_tests++;
    V3 normal = vcross(vsub(b, a), vsub(cc, a));
    double dn = vdot(ray->dir, normal);
    if (-0.000001 < dn && dn < 0.000001) return h;
    double t = (vdot(normal, a) - vdot(normal, ray->start)) / vdot(normal, ray->dir);
    if (c) {
        /* fragility (test aid): a ray that starts within a few float32 steps of the face's plane (a child ray that leaves its
         * surface at a grazing angle, seen from that surface's plane or from a face coplanar with it) is on either side of the
         * t >= 0 cut depending on how its start point was rounded */
        double dist = fabs(vdot(normal, ray->start) - vdot(normal, a)) / sqrt(vdot(normal, normal));
        double mag = fabs(ray->start.x) > fabs(ray->start.y) ? fabs(ray->start.x) : fabs(ray->start.y);
        if (fabs(ray->start.z) > mag) mag = fabs(ray->start.z);
        if (dist < 4e-6 * (mag > 1 ? mag : 1)) c->frag = 0;
    }
    if (t < 0) {
        if (c && t > -1e-4) c->frag = 0; /* hit distance on the t >= 0 boundary */
        return h;
    }
    V3 p = vadd(ray->start, vmulk(ray->dir, t));
    double m1, m2;
    int inside = point_within_triangle(p, a, b, cc, &m1, &m2);
    if (c) {
        /* relative distance of the strict inside test from its decision boundary: the cross
         * products scale with (twice the triangle area)^2 = |normal|^2 */
        double scale = vdot(normal, normal);
        double r1 = fabs(m1) / scale, r2 = fabs(m2) / scale;
        double r = r1 < r2 ? r1 : r2;
        /* a sign flip of the smaller product changes the outcome only if the other is positive
         * or equally marginal; keeping it simple and conservative: record the smaller */
        if (r < c->frag) c->frag = r;
        if (t < 1e-4) c->frag = 0;
    }
    if (!inside) return h;
    h.hit = 1; h.t = t; h.point = p; h.normal = normal;
    h.direction = (vdot(ray->dir, normal) < 0) ? 0 : 1;
    return h;
}

typedef struct { int obj; V3 point; double t; V3 normal; int direction; int face; } Collision;

static void first_collision_faces(Ctx* c, const Ray* ray, int o, const int* faces, int n, int backCull,
                                  Collision* best) {
    const t3d_scene_desc* s = c->s;
    for (int i = 0; i < n; i++) {
        int t = faces[i];
        Hit h = face_collision(c, ray, tri_v(s, t, 0), tri_v(s, t, 1), tri_v(s, t, 2));
        if (h.hit && (h.direction == 0 || !backCull)) {
            /* fragility (test aid): two surfaces met at all but the same distance -- the ray runs through the line in which
             * two faces cross -- may change places in another precision, like a ray on a triangle's edge */
            if (best->obj >= 0) {
                double near = h.t < best->t ? h.t : best->t;
                double m = fabs(h.t - best->t) / (near > 1 ? near : 1);
                if (m > 0 && m < c->frag) c->frag = m;   /* (faces that coincide tie exactly in every precision) */
            }
            if (best->obj < 0 || h.t < best->t) {
                best->obj = o; best->point = h.point; best->t = h.t; best->normal = h.normal;
                best->direction = h.direction; best->face = t;
            }
        }
    }
}

/* Photon.getFaces T3DR.py:553-569: the concatenated face list is consumed in order, so the
 * candidates are visited directly instead of materialising the list. */
static void visit_faces(Ctx* c, const Ray* ray, int o, const BVH* b, int backCull, Collision* best) {
    if (!enters_aabb(c, ray, b->bmin, b->bmax)) return;
    if (b->n_faces) { first_collision_faces(c, ray, o, b->faces, b->n_faces, backCull, best); return; }
    if (b->left) visit_faces(c, ray, o, b->left, backCull, best);
    if (b->right) visit_faces(c, ray, o, b->right, backCull, best);
}

/* Photon.getFirstCollision T3DR.py:654-730.  `face` is a global triangle index (-1 none) and,
 * as in the reference, is not cleared when a sphere wins later. */
static Collision first_collision(Ctx* c, const Ray* ray, int lights, int backCull) {
    const t3d_scene_desc* s = c->s;
    Collision best; memset(&best, 0, sizeof best); best.obj = -1; best.face = -1;
    c->st.queries++;
    for (int o = 0; o < s->n_objects; o++) {
        int ty = s->obj_type[o];
        if (ty != T3D_OBJ_SPHERE && ty != T3D_OBJ_MESH && ty != T3D_OBJ_LIGHT) continue;
        if (c->o->use_aabb || c->o->use_bvh) {
            if (!enters_aabb(c, ray, c->bvh[o]->bmin, c->bvh[o]->bmax)) continue;
        }
        if (ty == T3D_OBJ_SPHERE || (lights && ty == T3D_OBJ_LIGHT)) {
            c->st.sphere_tests++;
            V3 ctr = v3(s->obj_center[3 * o], s->obj_center[3 * o + 1], s->obj_center[3 * o + 2]);
            Hit h = sphere_collision(c, ray, ctr, s->obj_radius[o], ty == T3D_OBJ_LIGHT, backCull);
            if (h.hit && best.obj >= 0) {   /* fragility: see first_collision_faces */
                double near = h.t < best.t ? h.t : best.t;
                double m = fabs(h.t - best.t) / (near > 1 ? near : 1);
                if (m > 0 && m < c->frag) c->frag = m;   /* (faces that coincide tie exactly in every precision) */
            }
            if (h.hit && (best.obj < 0 || h.t < best.t)) {
                best.obj = o; best.point = h.point; best.t = h.t; best.normal = h.normal;
                best.direction = h.direction;
            }
        } else if (ty == T3D_OBJ_MESH) {
            if (c->o->use_bvh) visit_faces(c, ray, o, c->bvh[o], backCull, &best);
            else {
                int n = s->obj_tri_count[o], b0 = s->obj_tri_begin[o];
                for (int i = 0; i < n; i++) {
                    int t = b0 + i;
                    first_collision_faces(c, ray, o, &t, 1, backCull, &best);
                }
            }
        }
    }
    return best;
}

/* Object.getShader T3DE.py:1046-1054: mesh -> face shader, sphere -> shaders[0], light -> None (-1) */
static int get_shader(const t3d_scene_desc* s, int obj, int face) {
    int ty = s->obj_type[obj];
    if (ty == T3D_OBJ_MESH) return s->tri_shader[face];
    if (ty == T3D_OBJ_SPHERE) return s->obj_shader0[obj];
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * Random direction triple for the ray with heap index `heap`
 * ---------------------------------------------------------------------------------------- */
static V3 draw3(Ctx* c, int heap) {
    double u[3];
    for (int k = 0; k < 3; k++) {
        if (c->o->rng_mode == T3D_RNG_MT_SERIAL) u[k] = -1 + (1 - -1) * mt_random(&c->mt);
        else {
            double r = t3do_keyed_bits24(c->o->seed, c->kx, c->ky, c->ks, (uint32_t)heap, (uint32_t)k) / 16777216.0;
            u[k] = -1.0 + (1.0 - -1.0) * r;
        }
    }
    return v3(u[0], u[1], u[2]);
}

/* Photon.bounce T3DR.py:733-759 */
static Ray bounce(Ctx* c, const Ray* ray, const Collision* ci, int heap_child) {
    V3 normal = vnorm(ci->normal);
    V3 diffuse = vnorm(draw3(c, heap_child));
    if (vdot(diffuse, normal) < 0) diffuse = vmulk(diffuse, -1);
    V3 specular = vflip(ray->dir, normal);
    int sh = get_shader(c->s, ci->obj, ci->face);
    double rough = sh >= 0 ? c->s->shader_rough[sh] : 1;
    V3 dir = vadd(vmulk(specular, 1 - rough), vmulk(diffuse, rough));
    V3 start = vadd(ci->point, vmulk(dir, 0.1));
    return make_ray(start, dir, ray->ior, ray->obj);
}

/* Photon.transmit T3DR.py:904-956; returns 0 for total internal reflection (no draw happens) */
static int transmit(Ctx* c, const Ray* ray, const Collision* ci, int heap_child, Ray* out) {
    V3 normal = vnorm(ci->normal);
    V3 direction = vnorm(ray->dir);
    double n1 = ray->ior;
    int tobj = ci->obj;
    double n2 = c->s->obj_ior[tobj];
    if (tobj == ray->obj) { normal = vmulk(normal, -1); n2 = 1; tobj = -1; }
    double cosI = vdot(direction, normal) * -1;
    double sin2I = 1 - pow(cosI, 2.0);
    double ratio = n1 / n2;
    double sin2T = pow(ratio, 2.0) * sin2I;
    if (sin2T > 1) return 0;
    double cosT = sqrt(1 - sin2T);
    V3 par = vmulk(direction, ratio);
    V3 perp = vmulk(normal, (ratio * cosI) - cosT);
    V3 specular = vadd(par, perp);
    V3 diffuse = vnorm(draw3(c, heap_child));
    if (vdot(diffuse, normal) > 0) diffuse = vmulk(diffuse, -1);
    int sh = get_shader(c->s, ci->obj, ci->face);
    double rough = sh >= 0 ? c->s->shader_rough[sh] : 1;
    V3 dir = vadd(vmulk(specular, 1 - rough), vmulk(diffuse, rough));
    V3 start = vadd(ci->point, vmulk(dir, 0.1));
    *out = make_ray(start, dir, n2, tobj);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Trace tree (T3DE.py:273-280) and Photon.trace T3DR.py:959-1001
 * ---------------------------------------------------------------------------------------- */
typedef struct Node { Ray ray; Collision ci; struct Node *left, *right; int heap; } Node;

static void free_tree(Node* n) { if (!n) return; free_tree(n->left); free_tree(n->right); free(n); }

static Node* trace(Ctx* c, const Ray* ray, int bounces, int lights, int heap) {
    c->st.trace_queries++;
    Collision ci = first_collision(c, ray, lights, ray->ior == 1);
    if (ci.obj < 0) return NULL;
    c->st.nodes++;
    Node* node = (Node*)calloc(1, sizeof(Node));
    node->ray = *ray; node->ci = ci; node->heap = heap;
    const t3d_scene_desc* s = c->s;
    int sh = get_shader(s, ci.obj, ci.face);
    int ty = s->obj_type[ci.obj];
    if (lights && (ty == T3D_OBJ_LIGHT || (ty == T3D_OBJ_MESH && s->shader_emis[sh] > 0))) return node;
    if (bounces == 0) return node;
    double refl = sh >= 0 ? s->shader_refl[sh] : 1;
    Node* tnode = NULL;
    if (refl != 1) {
        Ray tr;
        if (transmit(c, ray, &ci, 2 * heap + 1, &tr)) tnode = trace(c, &tr, bounces - 1, 1, 2 * heap + 1);
    }
    Node* bnode = NULL;
    if (refl != 0) {
        Ray br = bounce(c, ray, &ci, 2 * heap);
        bnode = trace(c, &br, bounces - 1, 1, 2 * heap);
    }
    node->left = bnode; node->right = tnode;
    return node;
}

/* ------------------------------------------------------------------------------------------
 * Shading: RGB (T3DE.py:366-423), getAttenuation T3DR.py:782-785, getVisibility :762-778,
 * getDirectLight :820-900, computeColor :1004-1083
 * ---------------------------------------------------------------------------------------- */
typedef struct { double r, g, b; } RGBd;
static RGBd rgb(double r, double g, double b) { RGBd c = {r, g, b}; return c; }
static RGBd rgb_clamp(RGBd c) { return rgb(c.r < 255 ? c.r : 255, c.g < 255 ? c.g : 255, c.b < 255 ? c.b : 255); }
/* min(x, 255) in Python returns x unless 255 < x; NaN never occurs on this path */
static RGBd rgb_add(RGBd a, RGBd b) { return rgb_clamp(rgb(a.r + b.r, a.g + b.g, a.b + b.b)); }
static RGBd rgb_mulk(RGBd a, double k) { return rgb(a.r * k, a.g * k, a.b * k); }
static RGBd rgb_mul(RGBd a, RGBd b) { return rgb(a.r * b.r, a.g * b.g, a.b * b.b); }

static double attenuation(double distance) {
    double att = 1 / (1 + 0.01 * pow(distance, 2.0));
    double m = att < 1 ? att : 1;   /* min(1, att) */
    return m > 0 ? m : 0;           /* max(0, .) */
}

static RGBd shader_color(const t3d_scene_desc* s, int sh) {
    return rgb(s->shader_color[3 * sh], s->shader_color[3 * sh + 1], s->shader_color[3 * sh + 2]);
}

static double visibility(Ctx* c, const Ray* ray) {
    const t3d_scene_desc* s = c->s;
    c->st.shadow_queries++;
    Collision ci = first_collision(c, ray, 1, 0);
    if (ci.obj >= 0 && s->obj_type[ci.obj] != T3D_OBJ_LIGHT) {
        int sh = get_shader(s, ci.obj, ci.face);
        if (!(s->obj_type[ci.obj] == T3D_OBJ_MESH && s->shader_emis[sh] > 0))
            return 1 - s->shader_refl[sh];
    }
    return 1;
}

static void direct_light(Ctx* c, const Collision* ci, RGBd* color_out, double* intensity_out) {
    const t3d_scene_desc* s = c->s;
    RGBd total = rgb(0, 0, 0);
    double totalI = 0;
    V3 cam = v3(s->cam_pos[0], s->cam_pos[1], s->cam_pos[2]);
    for (int l = 0; l < s->n_lights; l++) {
        V3 pos = v3(s->light_pos[3 * l], s->light_pos[3 * l + 1], s->light_pos[3 * l + 2]);
        double strength = s->light_strength[l];
        RGBd color = rgb(s->light_color[3 * l], s->light_color[3 * l + 1], s->light_color[3 * l + 2]);
        V3 bdir = vnorm(vsub(pos, ci->point));
        V3 bstart = vadd(ci->point, vmulk(bdir, 0.2));
        Ray br = make_ray(bstart, bdir, 1, -1);
        c->st.rays_api++;
        V3 lightVector = vnorm(br.dir);
        V3 normal = vnorm(ci->normal);
        double ldn = vdot(lightVector, normal);
        double diffuse = ldn > 0 ? ldn : 0;
        V3 refl = vsub(vmulk(vmulk(normal, vdot(lightVector, normal)), 2), lightVector);
        V3 view = vsub(cam, ci->point);
        double spec = (vdot(vnorm(refl), vnorm(view)) < 0.96) ? 0 : 1;
        int sh = get_shader(s, ci->obj, ci->face);
        double rough = sh >= 0 ? s->shader_rough[sh] : 1;
        double vis = visibility(c, &br);
        double att = attenuation(vmag(vsub(pos, ci->point)));
        double intensity = ((diffuse * rough) + (spec * (1 - rough))) * strength * vis * att;
        /* not part of the reference: how many shadow rays cannot change anything, the factor in front of the visibility
           being zero -- or below 1e-6, where float32 may see a zero (a light in the plane of the surface it lights).
           The reference and this oracle trace them all the same; the HIP library skips those whose factor is zero in
           float32, and its count (t3d_stats.shadow_culled) must not exceed this one */
        if ((diffuse * rough) + (spec * (1 - rough)) <= 1e-6 || strength == 0) c->st.shadow_culled++;
        total = rgb_add(total, rgb_mulk(color, 0.6));
        totalI += intensity;
    }
    *color_out = total; *intensity_out = totalI;
}

static void record_node(Ctx* c, const Node* n, RGBd col) {
    if (!c->rec_enabled) return;
    if (c->rec_n < c->rec_cap) {
        t3do_node* r = &c->rec[c->rec_n];
        const t3d_scene_desc* s = c->s;
        r->px = (int32_t)c->kx; r->py = (int32_t)c->ky; r->sample = (int32_t)c->ks; r->heap = n->heap;
        r->obj = n->ci.obj;
        r->face = (s->obj_type[n->ci.obj] == T3D_OBJ_MESH) ? n->ci.face - s->obj_tri_begin[n->ci.obj] : -1;
        r->direction = n->ci.direction; r->ray_obj = n->ray.obj;
        r->t = n->ci.t; r->ray_ior = n->ray.ior;
        r->point[0] = n->ci.point.x; r->point[1] = n->ci.point.y; r->point[2] = n->ci.point.z;
        r->normal[0] = n->ci.normal.x; r->normal[1] = n->ci.normal.y; r->normal[2] = n->ci.normal.z;
        r->ray_start[0] = n->ray.start.x; r->ray_start[1] = n->ray.start.y; r->ray_start[2] = n->ray.start.z;
        r->ray_dir[0] = n->ray.dir.x; r->ray_dir[1] = n->ray.dir.y; r->ray_dir[2] = n->ray.dir.z;
        r->color[0] = col.r; r->color[1] = col.g; r->color[2] = col.b;
    }
    c->rec_n++;
}

static RGBd compute_color(Ctx* c, const Node* n) {
    const t3d_scene_desc* s = c->s;
    RGBd ambient = rgb(c->o->ambient[0], c->o->ambient[1], c->o->ambient[2]);
    if (!n) return ambient;
    int obj = n->ci.obj;
    V3 cam = v3(s->cam_pos[0], s->cam_pos[1], s->cam_pos[2]);
    if (s->obj_type[obj] == T3D_OBJ_LIGHT) {
        RGBd col = shader_color(s, s->obj_shader0[obj]);
        col = rgb_mulk(col, s->obj_strength[obj]);
        col = rgb_mulk(col, attenuation(vmag(vsub(n->ci.point, cam))));
        col = rgb_clamp(col);
        record_node(c, n, col);
        return col;
    }
    int sh = get_shader(s, obj, n->ci.face);
    if (s->obj_type[obj] == T3D_OBJ_MESH && s->shader_emis[sh] > 0) {
        RGBd col = rgb_mulk(shader_color(s, sh), s->shader_emis[sh]);
        col = rgb_mulk(col, attenuation(vmag(vsub(n->ci.point, cam))));
        col = rgb_clamp(col);
        record_node(c, n, col);
        return col;
    }
    RGBd color = shader_color(s, sh);
    /* the record order is depth-first with the bounce child first, like the recursion below */
    int my_slot = c->rec_n;
    if (c->rec_enabled) c->rec_n++; /* reserve so that parents precede children */
    RGBd reflected = n->left ? compute_color(c, n->left) : rgb_mulk(ambient, 2.0 / 255);
    RGBd transmitted = n->right ? compute_color(c, n->right) : rgb_mulk(ambient, 2.0 / 255);
    double refl = s->shader_refl[sh];
    double tf = 1 - refl;
    double rd = n->left ? vmag(vsub(n->left->ci.point, n->ray.start)) : 0;
    double ra = attenuation(rd);
    double td = n->right ? vmag(vsub(n->right->ci.point, n->ray.start)) : 0;
    double ta = attenuation(td);
    RGBd indirect = rgb_add(rgb_mulk(rgb_mulk(reflected, refl), ra),
                            rgb_mulk(rgb_mulk(rgb_mulk(transmitted, tf), ta), 1.4));
    RGBd lcol; double I;
    direct_light(c, &n->ci, &lcol, &I);
    RGBd direct = rgb_mulk(rgb_add(rgb_mulk(color, I), rgb_mulk(rgb_mulk(lcol, I), 2.0 / 255)), refl);
    RGBd out = rgb_add(rgb_add(indirect, direct), rgb_mulk(rgb_mul(color, ambient), 1.0 / 255));
    out = rgb_clamp(out);
    if (c->rec_enabled) {
        int save = c->rec_n; c->rec_n = my_slot; record_node(c, n, out); c->rec_n = save;
    }
    return out;
}

/* ------------------------------------------------------------------------------------------
 * Camera: Photon.pixelToVector T3DR.py:528-542, getRay T3DR.py:545-551
 * ---------------------------------------------------------------------------------------- */
static V3 pixel_to_vector(const t3d_scene_desc* s, const t3d_opts* o, int px, int py) {
    double width = o->width, height = o->height;
    double ratio = width / height;
    double x = ((px / width) * ratio) - (ratio / 2);
    double y = 0.5 - (py / height);
    double v[3] = {x, y, s->cam_zlen}, one[3] = {1, 1, 1}, out[3];
    t3do_apply_transforms(v, s->cam_pos, one, s->cam_euler, out);
    return v3(out[0], out[1], out[2]);
}
static Ray camera_ray(const t3d_scene_desc* s, const t3d_opts* o, int px, int py) {
    V3 pv = pixel_to_vector(s, o, px, py);
    V3 start = v3(s->cam_pos[0], s->cam_pos[1], s->cam_pos[2]);
    return make_ray(start, vnorm(vsub(pv, start)), 1, -1);
}
void t3do_camera_ray(const t3d_scene_desc* s, const t3d_opts* o, int px, int py, double* pixvec,
                     double* start, double* dir) {
    V3 pv = pixel_to_vector(s, o, px, py);
    Ray r = camera_ray(s, o, px, py);
    pixvec[0] = pv.x; pixvec[1] = pv.y; pixvec[2] = pv.z;
    start[0] = r.start.x; start[1] = r.start.y; start[2] = r.start.z;
    dir[0] = r.dir.x; dir[1] = r.dir.y; dir[2] = r.dir.z;
}

/* float.__floordiv__ (CPython floatobject.c float_floor_div / _float_div_mod) for RGB.mean
 * T3DE.py:410-412 */
static double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod) { if ((wx < 0) != (mod < 0)) { mod += wx; div -= 1.0; } }
    double floordiv;
    if (div) { floordiv = floor(div); if (div - floordiv > 0.5) floordiv += 1.0; }
    else floordiv = copysign(0.0, vx / wx);
    return floordiv;
}

static void ctx_init(Ctx* c, const t3d_scene_desc* s, const t3d_opts* o) {
    memset(c, 0, sizeof *c);
    c->s = s; c->o = o; c->bvh = build_all(s);
    if (o->rng_mode == T3D_RNG_MT_SERIAL) mt_seed(&c->mt, o->seed);
}
static void ctx_free(Ctx* c) {
    for (int i = 0; i < c->s->n_objects; i++) free_bvh(c->bvh[i]);
    free(c->bvh);
}

/* Photon.renderStripe T3DR.py:1183-1242 for rows [row0, row0+nrows) */
int t3do_render_rows(const t3d_scene_desc* s, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_out,
                     double* radiance_out, int32_t* primary_out, double* primary_t_out,
                     double* fragility_out, t3d_stats* stats, t3do_node* nodes_out, int nodes_cap,
                     int64_t* nodes_written) {
    if (!s || !o || !bgr_out || o->width <= 0 || o->height <= 0) return T3D_E_INVALID;
    Ctx c; ctx_init(&c, s, o);
    c.rec = nodes_out; c.rec_cap = nodes_cap; c.rec_enabled = nodes_out != NULL;
    int W = o->width;
    int nsamp = o->bounces > 1 ? o->samples : 1;
    RGBd* samples = (RGBd*)malloc(sizeof(RGBd) * (nsamp > 0 ? nsamp : 1));
    const int rstride = o->row_stride > 1 ? o->row_stride : 1;
    for (int y = 0; y < nrows; y++)
        for (int x = 0; x < W; x++) {
            int py = row0 + y * rstride;
            Ray ray = camera_ray(s, o, x, py);
            c.kx = (uint32_t)x; c.ky = (uint32_t)py;
            size_t pix = (size_t)y * W + x;
            c.frag = INFINITY;
            for (int k = 0; k < nsamp; k++) {
                c.ks = (uint32_t)k;
                uint64_t q0 = c.st.queries;
                Node* tree = trace(&c, &ray, o->bounces, o->lights, 1);
                (void)q0;
                c.st.rays_api++;
                if (k == 0) {
                    if (primary_out) {
                        if (tree) {
                            primary_out[2 * pix] = tree->ci.obj;
                            primary_out[2 * pix + 1] = s->obj_type[tree->ci.obj] == T3D_OBJ_MESH
                                                           ? tree->ci.face - s->obj_tri_begin[tree->ci.obj] : -1;
                        } else { primary_out[2 * pix] = -1; primary_out[2 * pix + 1] = -1; }
                    }
                    if (primary_t_out) primary_t_out[pix] = tree ? tree->ci.t : NAN;
                }
                samples[k] = compute_color(&c, tree);
                free_tree(tree);
            }
            if (fragility_out) fragility_out[pix] = c.frag; /* smallest margin met by any ray of any sample of the pixel */
            /* RGB.mean T3DE.py:401-415 */
            double tr = 0, tg = 0, tb = 0;
            for (int k = 0; k < nsamp; k++) { tr += samples[k].r; tg += samples[k].g; tb += samples[k].b; }
            if (radiance_out) {
                radiance_out[3 * pix] = tr / nsamp; radiance_out[3 * pix + 1] = tg / nsamp; radiance_out[3 * pix + 2] = tb / nsamp;
            }
            RGBd mean = rgb_clamp(rgb(py_floordiv(tr, nsamp), py_floordiv(tg, nsamp), py_floordiv(tb, nsamp)));
            /* toTuple('bgr') then a uint8 store (truncation) T3DR.py:1219-1221 */
            bgr_out[3 * pix] = (uint8_t)mean.b; bgr_out[3 * pix + 1] = (uint8_t)mean.g; bgr_out[3 * pix + 2] = (uint8_t)mean.r;
        }
    free(samples);
    if (stats) *stats = c.st;
    if (nodes_written) *nodes_written = c.rec_n;
    ctx_free(&c);
    return T3D_OK;
}

/* Primary-hit fragility only: for every pixel, the smallest relative margin of any inside test
 * (and of the t >= 0 cut, and of the silhouette of any sphere) met by the PRIMARY ray.  Pixels whose margin is below a float32-sized
 * threshold are the "edge-tie pixels" where a float32 tracer may legitimately pick another face. */
int t3do_primary_fragility(const t3d_scene_desc* s, const t3d_opts* o, int row0, int nrows, double* frag_out) {
    if (!s || !o || !frag_out) return T3D_E_INVALID;
    Ctx c; ctx_init(&c, s, o);
    int W = o->width;
    for (int y = 0; y < nrows; y++)
        for (int x = 0; x < W; x++) {
            Ray ray = camera_ray(s, o, x, row0 + y * (o->row_stride > 1 ? o->row_stride : 1));
            c.frag = INFINITY;
            (void)first_collision(&c, &ray, o->lights, 1);
            frag_out[(size_t)y * W + x] = c.frag;
        }
    ctx_free(&c);
    return T3D_OK;
}

/* Single closest-hit query (Photon.getFirstCollision), for known-answer tests and picking checks. */
int t3do_first_collision(const t3d_scene_desc* s, const t3d_opts* o, const double* start, const double* dir,
                         int lights, int backCull, int32_t* obj, int32_t* face, double* t, double* point,
                         double* normal, int32_t* direction) {
    Ctx c; ctx_init(&c, s, o);
    Ray r = make_ray(v3(start[0], start[1], start[2]), v3(dir[0], dir[1], dir[2]), 1, -1);
    Collision ci = first_collision(&c, &r, lights, backCull);
    *obj = ci.obj;
    *face = (ci.obj >= 0 && s->obj_type[ci.obj] == T3D_OBJ_MESH) ? ci.face - s->obj_tri_begin[ci.obj] : -1;
    *t = ci.obj >= 0 ? ci.t : NAN;
    if (ci.obj >= 0) {
        point[0] = ci.point.x; point[1] = ci.point.y; point[2] = ci.point.z;
        normal[0] = ci.normal.x; normal[1] = ci.normal.y; normal[2] = ci.normal.z;
        *direction = ci.direction;
    }
    ctx_free(&c);
    return T3D_OK;
}

/* Known-answer helpers: one reference function each */
int t3do_face_collision(const double* start, const double* dir, const double* verts, double* t,
                        double* point, double* normal, int32_t* direction) {
    Ray r = make_ray(v3(start[0], start[1], start[2]), v3(dir[0], dir[1], dir[2]), 1, -1);
    Hit h = face_collision(NULL, &r, v3(verts[0], verts[1], verts[2]), v3(verts[3], verts[4], verts[5]),
                           v3(verts[6], verts[7], verts[8]));
    if (!h.hit) return 0;
    *t = h.t; *direction = h.direction;
    point[0] = h.point.x; point[1] = h.point.y; point[2] = h.point.z;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    return 1;
}
int t3do_sphere_collision(const double* start, const double* dir, const double* center, double radius,
                          int is_light, int backCull, double* t, double* point, double* normal,
                          int32_t* direction) {
    Ray r = make_ray(v3(start[0], start[1], start[2]), v3(dir[0], dir[1], dir[2]), 1, -1);
    Hit h = sphere_collision(NULL, &r, v3(center[0], center[1], center[2]), radius, is_light, backCull);
    if (!h.hit) return 0;
    *t = h.t; *direction = h.direction;
    point[0] = h.point.x; point[1] = h.point.y; point[2] = h.point.z;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    return 1;
}
/* leaf order of the BVH of object `obj`: face indices (within the mesh) left to right */
int t3do_bvh_leaf_order(const t3d_scene_desc* s, int obj, int32_t* out, int cap, int32_t* leaf_sizes,
                        int leaf_cap, int32_t* n_leaves) {
    BVH** all = build_all(s);
    int n = 0, nl = 0;
    BVH* stack[256]; int sp = 0;
    stack[sp++] = all[obj];
    while (sp) {
        BVH* b = stack[--sp];
        if (b->n_faces || (!b->left && !b->right)) {
            for (int i = 0; i < b->n_faces && n < cap; i++) out[n++] = b->faces[i] - s->obj_tri_begin[obj];
            if (nl < leaf_cap) leaf_sizes[nl] = b->n_faces;
            nl++;
        } else {
            if (b->right) stack[sp++] = b->right;
            if (b->left) stack[sp++] = b->left;
        }
    }
    *n_leaves = nl;
    for (int i = 0; i < s->n_objects; i++) free_bvh(all[i]);
    free(all);
    return n;
}
void t3do_object_box(const t3d_scene_desc* s, int obj, double* bmin, double* bmax) {
    BVH** all = build_all(s);
    bmin[0] = all[obj]->bmin.x; bmin[1] = all[obj]->bmin.y; bmin[2] = all[obj]->bmin.z;
    bmax[0] = all[obj]->bmax.x; bmax[1] = all[obj]->bmax.y; bmax[2] = all[obj]->bmax.z;
    for (int i = 0; i < s->n_objects; i++) free_bvh(all[i]);
    free(all);
}
double t3do_attenuation(double d) { return attenuation(d); }
double t3do_py_floordiv(double a, double b) { return py_floordiv(a, b); }
void t3do_mt_uniforms(uint32_t seed, int n, double a, double b, double* out) {
    MT m; mt_seed(&m, seed);
    for (int i = 0; i < n; i++) out[i] = a + (b - a) * mt_random(&m);
}
