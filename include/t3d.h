/*
 * t3d.h -- C ABI of libt3d_hip.so, the MI355X (gfx950) implementation of the T3DR "Photon"
 * ray-tracing path.
 *
 * The reference renderer is pure Python and has no FFI: its boundary is the Python class API
 * (Photon(scene, options).render(), T3DR.py:501-1242).  Each entry point below names the
 * reference function(s) it takes over.  Conventions: every function returns 0 on success or a
 * negative T3D_E_* code; t3d_last_error() gives the text; no exception crosses the ABI; all
 * pointers are plain host pointers unless the name says "device"; the library copies what it
 * needs during the call, the caller keeps ownership of everything it passes in.
 * One host thread per context, and one stream at a time: all calls of a context share its work buffers and
 * counters, so a caller that moves from one stream to another (or from t3d_render_rows_device on its own stream
 * to the entry points that use the context's stream) lets the first stream finish before it does.
 */
#ifndef T3D_H
#define T3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T3D_ABI_VERSION 9

enum {
    T3D_OK = 0,
    T3D_E_INVALID = -1,   /* bad argument (null pointer, size out of range, no scene)      */
    T3D_E_DEVICE = -2,    /* HIP runtime error                                              */
    T3D_E_NOMEM = -3,     /* device or host allocation failed                               */
    T3D_E_CAPACITY = -4,  /* a fixed capacity (trace depth, node buffer) would be exceeded  */
    T3D_E_CANCELLED = -5  /* the cancel callback asked to stop                              */
};

/* object kinds: Object.type strings of T3DE.py:960, 1218, 1251, 1261, 1279 */
enum { T3D_OBJ_MESH = 0, T3D_OBJ_SPHERE = 1, T3D_OBJ_LIGHT = 2, T3D_OBJ_OTHER = 3 /* camera, empty */ };

/* random-number modes for bounce()/transmit() (T3DR.py:740, T3DR.py:937) */
enum {
    T3D_RNG_KEYED = 0,     /* counter-based, see raytracing_amd/keyed_rng.py; the only mode the GPU has */
    T3D_RNG_MT_SERIAL = 1  /* CPython random.seed(seed) stream, depth-first row-major (CPU oracle only) */
};

/*
 * Flattened scene: what Photon.render() sees after deepcopy + Object.setTransforms()
 * (T3DR.py:1090-1100, T3DE.py:1057-1071), in scene.objects order.  Geometry is float64 exactly
 * as the reference holds it; the HIP library rounds to float32 when it builds device buffers.
 */
typedef struct t3d_scene_desc {
    int32_t n_objects;
    const int32_t* obj_type;      /* [O]   T3D_OBJ_*                                               */
    const int32_t* obj_tri_begin; /* [O]   first triangle of a mesh object in tri_*                */
    const int32_t* obj_tri_count; /* [O]   number of faces (0 unless mesh)                         */
    const double* obj_center;     /* [O*3] Object.position (sphere/light centre)                   */
    const double* obj_radius;     /* [O]   scale.x / 2            (T3DR.py:574)                    */
    const double* obj_ior;        /* [O]   material.ior           (T3DR.py:911)                    */
    const double* obj_strength;   /* [O]   Light.strength         (T3DR.py:812, 1018)              */
    const int32_t* obj_shader0;   /* [O]   index of shaders[0] in shader_*; -1 if none             */

    int32_t n_tris;
    const double* tri_verts;      /* [T*9] v0,v1,v2 world space (baked), faces in mesh.faces order */
    const int32_t* tri_shader;    /* [T]   shader of the face, Object.getShader rule T3DE.py:1046  */

    int32_t n_shaders;
    const double* shader_color;   /* [K*3] Shader.color r,g,b                                      */
    const double* shader_rough;   /* [K]   roughness                                               */
    const double* shader_refl;    /* [K]   reflectivity                                            */
    const double* shader_emis;    /* [K]   emissionStrength                                        */

    int32_t n_lights;             /* Photon.getAllLightsPos() list, T3DR.py:789-816                */
    const double* light_pos;      /* [L*3]                                                         */
    const double* light_strength; /* [L]                                                           */
    const double* light_color;    /* [L*3]                                                         */

    double cam_pos[3];            /* Camera.position                                               */
    double cam_euler[3];          /* Camera.rotation (x,y,z radians), matrix = Rz*Ry*Rx T3DE.py:70 */
    double cam_zlen;              /* camera.length / 20          (T3DR.py:540)                     */
} t3d_scene_desc;

/* Photon options that influence pixels (T3DR.py:519-525) plus the additive keys seed / rng_mode. */
typedef struct t3d_opts {
    int32_t width, height;
    int32_t bounces;
    int32_t samples;    /* used only when bounces > 1 (T3DR.py:1209)                               */
    int32_t lights;     /* options['lights']: primary rays see Light objects / stop at emitters    */
    int32_t rng_mode;   /* T3D_RNG_*                                                               */
    uint32_t seed;
    int32_t use_bvh;    /* options['bvh'];  results do not depend on it (conservative culling)     */
    int32_t use_aabb;   /* options['aabb']; likewise                                               */
    int32_t row_stride; /* local row k of a render call is image row row0 + k * row_stride; 0 means 1  */
    int32_t frames;     /* frames per render call (0 means 1).  A call with frames = B renders the same rows B
                           times, frame f with the seed `seed + f * frame_seed_step`, and every output array
                           holds the B results one after the other ([B][nrows][W]...): the pixels are those of B
                           calls with one frame each, but the launches are B times as large -- for callers whose
                           share of a frame is too small to fill the device (one rank of several).  Honoured by
                           t3d_render_rows, _device, _device_static and _async; t3d_render_frame_multi and the
                           single-ray entry points render one frame whatever it says.  No counterpart in the
                           reference                                                                      */
    uint32_t frame_seed_step;
    double ambient[3];  /* options['ambient'] r,g,b                                                */
} t3d_opts;

#define T3D_STATS_N_COUNTERS 10
#define T3D_STATS_MAX_LEVELS 16
typedef struct t3d_stats {
    /* T3D_STATS_N_COUNTERS integer counters first, then the timings */
    uint64_t rays_api;       /* Rendered.rays: primary samples + shadow rays (T3DR.py:842, 1211)     */
    uint64_t queries;        /* closest-hit queries: primary + bounce + transmit + shadow            */
    uint64_t trace_queries;  /* of which tree rays (primary, bounce, transmit)                        */
    uint64_t shadow_queries; /* of which shadow rays (one per light per shaded node)                  */
    uint64_t tri_tests;      /* ray-triangle tests: counted by the LBVH traversal; on the brute-force path
                                queries x triangles (plane groups a whole wave skips are not subtracted)  */
    uint64_t sphere_tests;   /* ray-sphere tests, likewise                                           */
    uint64_t aabb_tests;     /* ray-box tests executed                                               */
    uint64_t nodes;          /* shaded trace-tree nodes (emitter leaves are folded into the parent)  */
    uint64_t launches;       /* kernel launches issued                                               */
    uint64_t n_k_level;      /* tree levels traced (one k_level launch each; LBVH scenes: three kernels) */
    double ms_device;        /* device time of the whole call (HIP events on the launch stream)      */
    double ms_k_level;       /* time spent tracing the levels (k_level, or the three LBVH kernels)   */
    double ms_resolve;       /* colour resolve + image store                                         */
    /* per tree level (0 = primary rays), summed over the bands of the call; levels above bounces stay 0 */
    uint64_t level_rays[T3D_STATS_MAX_LEVELS];   /* tree rays traced at the level                    */
    uint64_t level_nodes[T3D_STATS_MAX_LEVELS];  /* shaded nodes created at the level                */
    double level_ms[T3D_STATS_MAX_LEVELS];       /* time spent tracing the level                     */
    uint64_t bvh_deep_pushes; /* LBVH path: traversal-stack entries that did not fit the per-lane LDS part   */
    uint64_t shadow_culled;   /* shadow rays answered without being traced: the light term of getDirectLight
                                 (T3DR.py:880) is zero whatever the ray hits when the surface faces away from the
                                 light and shows the camera no highlight, or the light's strength is zero.  They are
                                 counted in shadow_queries, queries and rays_api, as the reference counts them   */
} t3d_stats;

/* One trace-tree node as t3d_trace_pixel returns it (TraceTreeNode, T3DE.py:273-280). */
typedef struct t3d_node {
    int32_t heap;        /* 1 = primary, 2n = bounce child (left), 2n+1 = transmit child (right)   */
    int32_t obj;         /* index into scene.objects                                               */
    int32_t face;        /* face index in the mesh, -1 for spheres and lights                      */
    int32_t direction;   /* 0 = front hit, 1 = back hit                                            */
    float t;
    float point[3];
    float normal[3];     /* as stored by the reference: un-normalised for faces, unit for spheres  */
    float ray_start[3];
    float ray_dir[3];
    float ray_ior;
    int32_t ray_obj;     /* object the ray travels inside, -1 = none                               */
    float color[3];      /* computeColor() of the sub-tree (T3DR.py:1004-1083)                     */
} t3d_node;

typedef struct t3d_ctx t3d_ctx;

/* number of HIP devices visible to the process (<0: error code). */
int t3d_device_count(void);

/* Create a context bound to one device; takes over "Photon.__init__" state (T3DR.py:501-506). */
int t3d_create(int device, t3d_ctx** out);
void t3d_destroy(t3d_ctx* ctx);
const char* t3d_last_error(const t3d_ctx* ctx); /* ctx may be NULL: error of a failed t3d_create */

/*
 * Flattening step, device side: converts the description to float32 SoA device buffers
 * (pre-transformed triangles, spheres, lights, shaders, camera).  Replaces the per-render
 * deepcopy / setTransforms / buildBVH preparation of Photon.render (T3DR.py:1090-1107).
 */
int t3d_upload_scene(t3d_ctx* ctx, const t3d_scene_desc* scene);

/*
 * The uploaded scene again after spheres, Lights or the camera have moved: their centres, radii (scale), strengths, the
 * light positions and the camera are taken from `scene`, everything else stays as uploaded -- no plane grouping, no
 * hierarchy build; the hierarchy of a large scene is refitted (its boxes computed again bottom-up, its shape kept).
 * For callers that look at a live scene before every query (picking while the user drags a light, explorer.py:1346-1392,
 * T3DR.py:671-705).  `scene` must describe the uploaded scene otherwise -- same objects of the same kinds, same
 * triangles, same shaders; differing counts or kinds are refused with T3D_E_INVALID, changed triangles are not noticed.
 */
int t3d_update_scene(t3d_ctx* ctx, const t3d_scene_desc* scene);

/*
 * Render nrows image rows starting at row0 (consecutive, or every opts->row_stride-th row for
 * interleaved multi-GPU tiles): Photon.renderStripe (T3DR.py:1183-1242) with everything
 * under it (pixelToVector, getRay, trace, computeColor, getDirectLight, getFirstCollision).
 *   bgr_out       [nrows*W*3] uint8, B,G,R per pixel like the stripe array (T3DR.py:1219-1221)
 *   radiance_out  [nrows*W*3] float, mean sample colour before the floor (nullable)
 *   primary_out   [nrows*W*2] int32 (object index, face index) of sample 0's primary hit,
 *                 -1,-1 for a miss, face -1 for spheres/lights (nullable)
 *   primary_t_out [nrows*W]   float distance of that hit, NaN for a miss (nullable)
 * The result does not depend on how rows are split into calls (keyed random numbers).
 */
int t3d_render_rows(t3d_ctx* ctx, const t3d_opts* opts, int row0, int nrows, uint8_t* bgr_out,
                    float* radiance_out, int32_t* primary_out, float* primary_t_out,
                    t3d_stats* stats);

/*
 * Same work, outputs left in device memory owned by the caller (e.g. a torch tensor's data
 * pointer) and launched on `stream` (a hipStream_t passed as void*, NULL = the context's own
 * stream).  Does not synchronise unless `stats` is non-NULL.  Only a call with `stats` can see that a
 * trace tree outgrew the work buffers and trace the band again with larger ones.  A call without `stats`
 * cannot: its kernels raise a flag in host-visible memory instead, which stays raised until the host looks,
 * which it does at the start of every later render / trace call on the context and in t3d_last_call_stats.
 * Whichever comes first returns T3D_E_CAPACITY once -- for the EARLIER call, whose output is invalid -- and
 * from then on asynchronous calls on the context start with buffers twice as large (a later call that returns
 * T3D_OK did run).  The flag covers every band of a frame that was cut into several.
 */
int t3d_render_rows_device(t3d_ctx* ctx, const t3d_opts* opts, int row0, int nrows,
                           uint8_t* bgr_device, float* radiance_device, int32_t* primary_device,
                           float* primary_t_device, void* stream, t3d_stats* stats);

/*
 * The same rows, started and not waited for: the kernels and the copy of the finished pixels into `bgr_pinned`
 * are queued on the context's stream and the call returns.  `bgr_pinned` [nrows*W*3] should be page-locked
 * host memory (t3d_host_alloc) so that the copy runs at the link's rate and really is asynchronous.
 * t3d_last_call_stats waits for the frame, returns its counters and reports T3D_E_CAPACITY if a trace tree did
 * not fit the work buffers (then ask again through t3d_render_rows, which grows them).  Lets a caller prepare
 * the next thing on the host (Photon.render: flatten the scene again to see whether it changed) while the device
 * renders.  No counterpart in the reference, whose render() blocks (T3DR.py:1086-1181).
 */
int t3d_render_rows_async(t3d_ctx* ctx, const t3d_opts* opts, int row0, int nrows, uint8_t* bgr_pinned);

/*
 * One frame on n contexts at once -- normally one per GPU of the node; several contexts of one device work too.
 * Context k renders image rows k, k + n, k + 2n, ... (opts->row_stride is ignored) and one strided copy per context
 * puts them in their places in `bgr_frame` [H*W*3]: host memory (page-locked for full speed), or, with
 * frame_on_device != 0, device memory that every context's device can reach (hipMemcpyDeviceToDevice, peer copies
 * over xGMI between GPUs).  opts->height must be a multiple of n -- the reference's rule for its stripe threads
 * (T3DR.py:1111-1112) -- else T3D_E_INVALID.  Stands where Photon.render starts one thread per stripe and stacks the
 * stripes (T3DR.py:1127-1165).  Every context must hold the same scene.  `stats` (nullable): counters summed over the
 * contexts, times the maximum.  Errors are reported through ctxs[0].
 */
int t3d_render_frame_multi(t3d_ctx* const* ctxs, int n, const t3d_opts* opts, uint8_t* bgr_frame, int frame_on_device,
                           t3d_stats* stats);

/*
 * The rows of one call as n contiguous bands of rows, band k traced by ctxs[k] -- contexts of ONE device that hold the
 * same scene -- all bands at once, each on its context's own stream.  For the latency of a single frame: the launches
 * of one band leave part of the device empty while they start and while their last waves finish, and another band's
 * launches fill it (1920x1080, 4 bounces: 0.52 ms on one context, 0.44 ms as two bands; more bands gain nothing, and a
 * caller that keeps several frames in flight on several contexts already has this overlap).  Towards the caller the call
 * behaves like t3d_render_rows_device without `stats` on `stream` (NULL: the stream of ctxs[0]): the bands' streams
 * are forked from it and joined to it by events, the pixels are those of one call, nothing is waited for;
 * t3d_last_call_stats of every context gives the counters of its band (their sum is the call's), and reports a band
 * whose trace trees did not fit its buffers, as after t3d_render_rows_device.  opts->frames must be 0 or 1.
 * t3d_render_rows_async_split is t3d_render_rows_async in the same way: the finished rows are copied to `bgr_pinned` on
 * the stream of ctxs[0].  Stands where Photon.render starts its stripe threads (T3DR.py:1111-1137).
 */
int t3d_render_rows_device_split(t3d_ctx* const* ctxs, int n, const t3d_opts* opts, int row0, int nrows,
                                 uint8_t* bgr_device, void* stream);
int t3d_render_rows_async_split(t3d_ctx* const* ctxs, int n, const t3d_opts* opts, int row0, int nrows, uint8_t* bgr_pinned);

/* Page-locked host memory for t3d_render_rows_async outputs (thin wrappers; no counterpart in the reference). */
int t3d_host_alloc(size_t bytes, void** out);
int t3d_host_free(void* p);

/*
 * t3d_render_rows_device for callers that cannot allocate device memory at the time of the call: code running under
 * HIP graph capture, tight serving loops, harnesses that forbid it.  Same arguments, same pixels.  It calls no
 * hipMalloc / hipFree and builds no launch graph: every work buffer (ray queues, node arrays, light intensities,
 * sample colours, the hit list of the LBVH path) is a piece of ONE array of T3D_STATIC_SCRATCH_MIB MiB that is part
 * of the library's code object, exists once per GPU from the moment the library is loaded there until the process
 * ends, and is shared by every context and every call on that GPU: calls that overlap on one GPU race (one at a time,
 * whatever their streams).  The frame is cut into as many bands of rows as the array takes (1920x1080 at 4 bounces
 * and 3 lights: 12 bands).  T3D_E_CAPACITY with a message naming the limit when even one image row does not fit
 * (very wide rows, thousands of lights, deep trees); with `stats` a band whose trees overflow is halved and traced
 * again, without `stats` the overflow is reported by the next call as for t3d_render_rows_device.  Nothing in the
 * array needs clearing between calls: the queue counters (in the context, allocated by t3d_create) say which entries
 * are valid.  No counterpart in the reference.
 */
#define T3D_STATIC_SCRATCH_MIB 128
int t3d_render_rows_device_static(t3d_ctx* ctx, const t3d_opts* opts, int row0, int nrows, uint8_t* bgr_device,
                                  float* radiance_device, int32_t* primary_device, float* primary_t_device,
                                  void* stream, t3d_stats* stats);

/*
 * Trace one pixel sample and return its tree (Photon.trace + computeColor for one initial ray;
 * serves Rendered.traces[(x, y)][sample] lazily, T3DR.py:1210-1216).  Returns the number of
 * nodes written (>= 0), nodes in depth-first order with the bounce child before the transmit
 * child, or a negative error.
 * One deviation from the reference, here and in t3d_trace_ray and the render calls: an emitter (Light object or
 * emissive face) hit by the initial ray is a leaf whatever opts->lights says.  The reference makes it one only when
 * `lights` is true (T3DR.py:973) and otherwise traces children below it that computeColor never looks at
 * (T3DR.py:1014-1041 return first): pixels and Rendered.rays are the same, the tree and the counters queries /
 * trace_queries / nodes / level_rays lack those never-shaded rays.
 */
int t3d_trace_pixel(t3d_ctx* ctx, const t3d_opts* opts, int x, int y, int sample, t3d_node* nodes,
                    int capacity);

/*
 * The same for an explicit initial ray: Photon.trace(ray, bounces, lights) followed by computeColor
 * (T3DR.py:959-1083), the pair Debug.inspect calls for a picked pixel (T3DR.py:455-469).  `direction` is
 * normalised like Ray.__init__ does (T3DE.py:126-130); `ior` / `inside_object` are ray.ior and the index of
 * ray.object in the scene's object list (-1 = none); opts->bounces and opts->lights play the parts of the
 * `bounces` and `lights` arguments; (key_x, key_y, key_sample) select the random numbers of the diffuse
 * directions the way a pixel position and sample number do in a render.
 */
int t3d_trace_ray(t3d_ctx* ctx, const t3d_opts* opts, const float* origin, const float* direction, float ior,
                  int inside_object, int key_x, int key_y, int key_sample, t3d_node* nodes, int capacity);

/*
 * Closest hit for n arbitrary rays (Photon.getFirstCollision, T3DR.py:654-730; used for picking,
 * explorer.py:1351-1355).  origins/dirs [n*3] (dirs are normalised by the library like
 * Ray.__init__, T3DE.py:128); out_obj/out_face [n] (-1 = miss), out_t [n], out_dirflag [n]
 * (0 front, 1 back), out_normal [n*3] (nullable).
 */
int t3d_closest_hits(t3d_ctx* ctx, int n, const float* origins, const float* dirs, int lights,
                     int back_cull, int32_t* out_obj, int32_t* out_face, float* out_t,
                     int32_t* out_dirflag, float* out_normal);

/*
 * Counters and timings of the most recent t3d_render_rows_device / _device_static / _async call: synchronises that
 * call's stream first.  Lets a caller run render calls asynchronously (stats == NULL) and still read counters and
 * kernel durations afterwards.  The counters and ms_device are those of the whole call, however many bands of rows
 * it was cut into (an asynchronous call of several bands sums its bands' counters on the device); ms_k_level,
 * ms_resolve and level_ms are those of its last band.
 * T3D_E_CAPACITY reports that the asynchronous call (any of its bands, or an earlier asynchronous call nobody
 * has asked about since) overflowed its node buffers (output invalid); see t3d_render_rows_device.
 * No counterpart in the reference (its timing is time.time() around the threads, T3DR.py:1119-1161).
 */
int t3d_last_call_stats(t3d_ctx* ctx, t3d_stats* stats);

/*
 * Per-kernel timing on (default) / off.  When on, an event is recorded after every k_level launch so
 * that t3d_stats::ms_k_level and ms_resolve are filled; each event costs a few microseconds of stream
 * idle time between two launches.  When off only ms_device (whole call) is measured and the other two
 * stay 0; a call that is also asynchronous (no `stats`) records no event at all and reports ms_device 0.
 * No counterpart in the reference.
 */
int t3d_set_timing(t3d_ctx* ctx, int per_kernel);

/*
 * Launch graphs.  A band of rows asked for again through t3d_render_rows_device without `stats` -- same options, rows,
 * output buffers and scene -- is replayed as one launch graph from its third request on (brute-force path): one
 * submission instead of ten.  The seed is not part of what must be the same: a band asked for again with another
 * opts->seed (the next frame of an animation) replays the same graph after the parameters of its first kernel node,
 * the only launch that reads the seed, have been set anew.  The counters say how often each of that happened on this
 * context (any pointer may be NULL).  No counterpart in the reference.
 */
int t3d_graph_counters(t3d_ctx* ctx, uint64_t* captured, uint64_t* replayed, uint64_t* reseeded);

/*
 * How a launch with little work (a thin band, one rank's share of a frame) is cut up.  ADAPTIVE (default): into
 * more, smaller trace rounds, so that more waves share it: shortest time for one band alone on the device.
 * FULL: always 128 rays per round, no slot wasted: highest throughput when the caller keeps several bands in
 * flight (several contexts on several streams), which hide each other's latency.  Same pixels either way.
 * No counterpart in the reference.
 */
enum { T3D_ROUNDS_ADAPTIVE = 0, T3D_ROUNDS_FULL = 1 };
int t3d_set_rounds(t3d_ctx* ctx, int mode);

/*
 * Acceleration structure used by scenes uploaded AFTER this call.
 *   T3D_ACCEL_AUTO         brute force over plane groups up to T3D_BRUTE_FORCE_MAX_PRIMS triangles + spheres,
 *                          the LBVH above (default)
 *   T3D_ACCEL_BRUTE_FORCE  always the brute-force loop
 *   T3D_ACCEL_LBVH         always the device-built LBVH (needs at least two primitives)
 * Stands where the reference has options['bvh'] / options['aabb'] (T3DR.py:659-707): there, as here, the
 * hierarchy is a conservative cull and the hit found is the brute-force closest hit.
 */
enum { T3D_ACCEL_AUTO = 0, T3D_ACCEL_BRUTE_FORCE = 1, T3D_ACCEL_LBVH = 2 };
#define T3D_BRUTE_FORCE_MAX_PRIMS 176   /* where the two meet in a closed room; in a sparse scene the LBVH is ahead from 80 on */
int t3d_set_accel(t3d_ctx* ctx, int mode);

/*
 * Who builds the hierarchy of scenes uploaded AFTER this call.  Up to `max_prims` triangles + spheres it is cut on the
 * host, top-down by surface area: the better tree (a mesh in a room reads 6-13 % fewer node records per ray, a uniform
 * soup 5 %) at about 0.3 us per primitive of host time (0.15 us from 32768 primitives on: several threads).  Above it, on
 * the device in Morton order, in a tenth of that time.  Default T3D_HOST_TREE_MAX_PRIMS, where the host build still costs less than it saves in a 4K frame; raise it
 * for a large scene that is rendered more than a hundred times; 0: always the device.  The environment variable
 * T3D_HOST_TREE_MAX, when set, takes precedence.  The pictures do not depend on it, t3d_stats.aabb_tests does.
 * Stands where the reference has Object.buildBVH (T3DE.py:985-1022), always a median split of each mesh.
 */
#define T3D_HOST_TREE_MAX_PRIMS 16384
int t3d_set_host_tree_limit(t3d_ctx* ctx, int max_prims);

/*
 * The better tree without the wait.  With `on` non-zero, a scene uploaded AFTER this call whose hierarchy is built on the
 * device (more primitives than the limit above) is traced through that Morton-order tree at once, while a host thread
 * of the library cuts the surface-area tree of the same primitives (0.15 us per primitive on up to eight threads); the
 * first render, trace or closest-hit call that finds it finished waits for the context's streams, writes the new order
 * and links over the old ones, refits and packs (a few milliseconds, once) and goes on through the better tree --
 * buffers and addresses stay what they were, so do the pictures (either tree is only a cull); t3d_stats.aabb_tests
 * drops.  For a caller that renders an unchanged or merely moving scene (t3d_update_scene) many times.  Off by default:
 * a caller that renders one frame per upload gains nothing (the environment variable T3D_BACKGROUND_TREE, read by
 * t3d_create, gives the initial value).  A too deep surface-area tree is dropped, as at upload.
 * t3d_upload_scene and t3d_destroy wait for a build that is still running.  t3d_render_rows_device_static never adopts
 * (the swap copies from pageable host memory): it goes on through the tree that is in place.
 *
 * t3d_tree_status says where the uploaded scene's hierarchy comes from; with `wait` non-zero it first waits for a
 * running build and adopts its tree (what a benchmark calls before it starts its clock).
 * Stands where the reference has Object.buildBVH (T3DE.py:985-1022), which always runs before the first pixel.
 */
enum {
    T3D_TREE_NONE = 0,        /* no scene, or one that is traced without a hierarchy */
    T3D_TREE_DEVICE = 1,      /* Morton order, built on the device */
    T3D_TREE_HOST = 2,        /* surface area, built on the host during t3d_upload_scene */
    T3D_TREE_BUILDING = 3,    /* the device's, while the host's is being built in the background */
    T3D_TREE_BACKGROUND = 4   /* the host's, adopted after the upload */
};
int t3d_set_background_tree(t3d_ctx* ctx, int on);
int t3d_tree_status(t3d_ctx* ctx, int wait);   /* T3D_TREE_* or a negative T3D_E_* */

/*
 * Cancellation: Photon.render polls cancelCallback() once per pixel and stops the stripe when it answers true
 * (T3DR.py:1227-1237).  Here `poll(user)` is asked before every band of rows and after every tree level of a
 * band (the stream is drained after each level while a callback is installed, so rendering with one costs the
 * overlap of host and device); a non-zero answer makes the render call return T3D_E_CANCELLED at once with
 * the output unfinished.  poll == NULL removes the callback.  The callback runs on the calling thread.
 */
int t3d_set_cancel(t3d_ctx* ctx, int (*poll)(void* user), void* user);

/*
 * The image filters of the reference's Post class on 8-bit images (byte-exact).
 *   upscale  Post.upscale (T3DR.py:1600-1614): every pixel of the [height][width][channels] image becomes a
 *            factor x factor block of the [height * factor][width * factor][channels] output.
 *   denoise  Post.denoise (T3DR.py:1616-1647): per channel the median -- sorted(values)[n / 2] -- of the
 *            (2 p + 1) x (2 p + 1) window around a pixel of a 3-channel image, p = kernel_size / 2 (at most 7); the
 *            pixels within p of the image border come out (0, 0, 0).  Input and output must not be the same buffer.
 * The _device forms take device pointers and queue their kernel on `stream` (NULL: the context's stream) without
 * waiting: a frame can be filtered where t3d_render_rows_device left it.  The host forms copy the image to the device
 * and the result back (device memory allocated and freed inside the call) and return when the result is there.
 */
int t3d_post_upscale_device(t3d_ctx* ctx, const uint8_t* in_device, int height, int width, int channels, int factor,
                            uint8_t* out_device, void* stream);
int t3d_post_denoise_device(t3d_ctx* ctx, const uint8_t* in_device, int height, int width, int kernel_size,
                            uint8_t* out_device, void* stream);
int t3d_post_upscale(t3d_ctx* ctx, const uint8_t* in, int height, int width, int channels, int factor, uint8_t* out);
int t3d_post_denoise(t3d_ctx* ctx, const uint8_t* in, int height, int width, int kernel_size, uint8_t* out);

int t3d_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* T3D_H */
