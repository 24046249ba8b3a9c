/*
 * t3d_oracle.h -- entry points of the float64 CPU restatement of the reference Photon renderer.
 * TEST INFRASTRUCTURE ONLY (checker + CPU baseline); see t3d_oracle.c.
 * Scene / options / stats structures are the ones of the product ABI (include/t3d.h).
 */
#ifndef T3D_ORACLE_H
#define T3D_ORACLE_H

#include <stdint.h>

#include "../include/t3d.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct t3do_node {
    int32_t px, py, sample, heap;
    int32_t obj, face, direction, ray_obj;
    double t, ray_ior;
    double point[3], normal[3], ray_start[3], ray_dir[3], color[3];
} t3do_node;

int t3do_render_rows(const t3d_scene_desc* s, const t3d_opts* o, int row0, int nrows, uint8_t* bgr_out,
                     double* radiance_out, int32_t* primary_out, double* primary_t_out,
                     double* fragility_out, t3d_stats* stats, t3do_node* nodes_out, int nodes_cap,
                     int64_t* nodes_written);
int t3do_primary_fragility(const t3d_scene_desc* s, const t3d_opts* o, int row0, int nrows, double* frag_out);
int t3do_first_collision(const t3d_scene_desc* s, const t3d_opts* o, const double* start, const double* dir,
                         int lights, int backCull, int32_t* obj, int32_t* face, double* t, double* point,
                         double* normal, int32_t* direction);
int t3do_face_collision(const double* start, const double* dir, const double* verts, double* t,
                        double* point, double* normal, int32_t* direction);
int t3do_sphere_collision(const double* start, const double* dir, const double* center, double radius,
                          int is_light, int backCull, double* t, double* point, double* normal,
                          int32_t* direction);
int t3do_enters_aabb(const double* start, const double* dir, const double* bmin, const double* bmax);
int t3do_bvh_leaf_order(const t3d_scene_desc* s, int obj, int32_t* out, int cap, int32_t* leaf_sizes,
                        int leaf_cap, int32_t* n_leaves);
void t3do_object_box(const t3d_scene_desc* s, int obj, double* bmin, double* bmax);
void t3do_camera_ray(const t3d_scene_desc* s, const t3d_opts* o, int px, int py, double* pixvec,
                     double* start, double* dir);
void t3do_euler_matrix(const double* euler, double* m);
void t3do_apply_transforms(const double* v, const double* pos, const double* scale, const double* euler,
                           double* out);
uint32_t t3do_keyed_bits24(uint32_t seed, uint32_t px, uint32_t py, uint32_t sample, uint32_t heap,
                           uint32_t comp);
double t3do_attenuation(double d);
double t3do_py_floordiv(double a, double b);
void t3do_mt_uniforms(uint32_t seed, int n, double a, double b, double* out);

#ifdef __cplusplus
}
#endif
#endif
