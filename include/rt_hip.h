/* rt_hip.h - C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * Drop-in boundary.  The reference renderer has no plugin or FFI interface; the
 * seam this library replaces is the WorkOrder queue consumed by
 * `bool RaytraceWork(WorkQueue*)` (main.cpp:244-259, 336-400), i.e. everything
 * the thread pool of main.cpp:432-461 does for all rows of an image:
 *   camera ray generation      main.cpp:350-388
 *   the bounce loop            main.cpp:265-334   (RaytraceWorld)
 *   world intersection         main.cpp:26-161    (IntersectWorldWide; 163-242 IntersectWorld)
 *   xorshift32 random stream   math_util.h:32-50
 *   Refract / Schlick          math_util.h:52-93
 *   sRGB transfer + ARGB pack  math_util.h:102-125, main.cpp:393
 *   ray counter                main.cpp:397
 * The caller keeps the reference's scene and image model: it builds the AoS
 * arrays `World` points to (scene.h:177-191), hands them over once, and receives
 * packed 0xAARRGGBB pixels, row-major top-down, as RaytraceWork writes them.
 *
 * Conventions: plain C, no exceptions, every call returns an rt_status; one
 * rt_context per GPU; calls on one context must be serialised by the caller,
 * distinct contexts may be used from distinct threads.  The library never
 * falls back to the CPU: without a usable gfx950 device rt_create fails.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1u

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = 1,
    RT_ERR_NO_DEVICE = 2,      /* no HIP device, or ordinal out of range */
    RT_ERR_HIP = 3,            /* a HIP runtime call failed; see rt_error_string */
    RT_ERR_NO_WORLD = 4,       /* rt_render* before rt_upload_world */
    RT_ERR_TOO_LARGE = 5,      /* planes + rectangles + materials exceed the LDS of one workgroup (spheres are not limited) */
    RT_ERR_OUT_OF_MEMORY = 6
} rt_status;

/* ---- scene records: byte-for-byte the reference's AoS types ------------- */
typedef struct rt_material {    /* Material, scene.h:13-18 (32 bytes) */
    float color[3];
    float refractive_index;     /* 0 = opaque */
    float emit_color[3];
    float reflection;           /* 0 = diffuse, 1 = mirror */
} rt_material;

typedef struct rt_sphere {      /* Sphere, scene.h:20-24 (20 bytes) */
    float position[3];
    float radius;
    uint32_t material_index;
} rt_sphere;

typedef struct rt_plane {       /* Plane, scene.h:32-36 (20 bytes) */
    float normal[3];
    float d;
    uint32_t material_index;
} rt_plane;

typedef struct rt_rectangle {   /* RectangleXY, scene.h:43-47 (80 bytes) */
    float inverse_transform[4][4]; /* row-major, ALREADY inverted (scene.h:408-411) */
    float normal[3];
    uint32_t material_index;
} rt_rectangle;

typedef struct rt_camera {      /* Camera, scene.h:163-168 (48 bytes) */
    float position[3];
    float z_axis[3];
    float y_axis[3];
    float x_axis[3];
} rt_camera;

typedef struct rt_world {       /* World, scene.h:177-191, without the CPU lane mirrors */
    uint32_t material_count;    /* >= 1; material 0 is the sky (main.cpp:280,326) */
    const rt_material* materials;
    uint32_t plane_count;
    const rt_plane* planes;
    uint32_t sphere_count;
    const rt_sphere* spheres;
    uint32_t rectangle_count;
    const rt_rectangle* rectangles;
    const rt_camera* camera;
} rt_world;

/* ---- render parameters --------------------------------------------------- */
enum {  /* rt_render_params.rng_mode */
    /* The reference's streams: one xorshift32 state per WorkOrder of
     * rows_per_order rows, seeded seed_constant*(startRow + endRow*36) and consumed
     * serially across x, samples and bounces (main.cpp:370-392).  One GPU lane per
     * WorkOrder; reproduces the reference's random numbers exactly. */
    RT_RNG_ROW_SERIAL = 0,
    /* One stream per (pixel, sample batch): the reference's seed for row y is the
     * root, x and the batch index are hashed in.  lanes_per_pixel lanes share a
     * pixel, each tracing spp/lanes_per_pixel consecutive samples; their sums are
     * combined by a fixed pairwise tree.  The throughput mode. */
    RT_RNG_PIXEL = 1
};

enum {  /* rt_render_params.intersect_mode */
    RT_INTERSECT_WIDE8 = 0,   /* arithmetic and tie-breaks of IntersectWorldWide (main.cpp:26-161) at LANE_WIDTH 8 */
    RT_INTERSECT_SCALAR = 1,  /* arithmetic and order of IntersectWorld (main.cpp:163-242) */
    RT_INTERSECT_WIDE4 = 2    /* IntersectWorldWide of the reference built at LANE_WIDTH 4 (simd.h:17-26, simd_sse.h): the same
                                 arithmetic, primitives dealt to four lanes, so closest-hit ties between primitives may fall
                                 to another one than at width 8 */
};

enum {  /* rt_render_params.flags */
    RT_FLAG_OUTPUT_SUM = 1u,    /* radiance receives the un-normalised sample sum instead of sum/spp */
    RT_FLAG_TIME_KERNEL = 2u,   /* bracket the render kernels with HIP events on their stream (rt_last_kernel_ms) */
    /* RT_RNG_PIXEL only (ignored otherwise, and for very low sample counts): a wave whose lanes have mostly
     * finished hands its last, nearly complete streams to a second launch that packs them densely; a third
     * small kernel adds the per-stream sums of each pixel.  Same results bit for bit, fewer idle lanes.  Uses
     * a workspace owned by the context (16 B per stream + 40 B per lane), allocated by the first render that
     * needs it or by rt_reserve: that call allocates, later ones do not.  The workspace is per context: at most
     * ONE compacted render of a context may be in flight at a time (two of them enqueued on different streams,
     * or two captured graphs replayed concurrently, race on it and give wrong pixels); use one context per
     * concurrent stream. */
    RT_FLAG_COMPACT_TAILS = 4u,
    /* Progressive rendering: radiance (which must be given; pixels must be NULL) is not overwritten but added
     * to: radiance[i] = radiance[i] + this render's sample sum, as a host loop `sum += pass` would do.  With
     * the device entry point the buffer is the caller's.  With rt_render / rt_render_multi the running sum
     * lives in the context: it is started by a render of the same rows with RT_FLAG_OUTPUT_SUM (or set with
     * rt_load_accumulator), every RT_FLAG_ACCUMULATE render adds to it, and `radiance` (may be NULL) receives
     * a copy.  Resolve the image at the end with rt_resolve / rt_resolve_device. */
    RT_FLAG_ACCUMULATE = 8u,
    /* RT_RNG_PIXEL renders skip, in iterations whose rays are all primary rays, the rectangles that no primary
     * ray of the wave's pixel tile can reach (decided once per wave, conservatively: same pixels, same ray
     * count).  This flag switches the skip off, for A/B measurements and tests. */
    RT_FLAG_NO_PRIMARY_CULL = 16u,
    /* Rectangles whose inverse transform holds the exact zeros of a rotation about the x or the y axis are
     * tested without the multiply-adds of those entries (same pixels, same ray count: a term with a zero factor
     * changes nothing a comparison can see).  This flag makes every rectangle use the full 3x4 transform, for A/B
     * measurements and tests. */
    RT_FLAG_DENSE_TRANSFORMS = 32u
};

typedef struct rt_render_params {
    uint32_t struct_size;        /* sizeof(rt_render_params), for ABI checking */
    uint32_t width, height;      /* full image size: defines film coordinates (main.cpp:350-377) */
    uint32_t samples_per_pixel;  /* WorkOrder.sampleSize (main.cpp:249) */
    uint32_t row_begin, row_end; /* rows rendered, global coordinates (WorkOrder start/endRowIndex) */
    /* Row interleave for multi-GPU sharding: rows [row_begin,row_end) are cut into
     * bands of band_rows rows and this render owns bands band_index, band_index +
     * band_count, ...  band_rows = 0 renders every row.  Outputs are compact:
     * rt_local_row_count() rows of `width` pixels, in increasing global row order. */
    uint32_t band_rows, band_index, band_count;
    uint32_t max_bounces;        /* reference: 8      (main.cpp:275) */
    float    min_hit_distance;   /* reference: 0.001  (main.cpp:29)  */
    float    plane_tolerance;    /* reference: 0.001  (main.cpp:28)  */
    uint32_t seed_constant;      /* reference: 262346 (main.cpp:370) */
    uint32_t rng_mode;           /* RT_RNG_* */
    uint32_t rows_per_order;     /* ROW_SERIAL: rows per WorkOrder; reference: 1 (main.cpp:433); 0 = one order */
    uint32_t lanes_per_pixel;    /* PIXEL: power of two <= 64 dividing samples_per_pixel */
    uint32_t batch_offset;       /* PIXEL: index of the first sample batch (distinct per pass / per GPU) */
    uint32_t intersect_mode;     /* RT_INTERSECT_* */
    uint32_t flags;              /* RT_FLAG_* */
} rt_render_params;

typedef struct rt_device_info {
    char     name[128];
    char     arch[32];                 /* e.g. "gfx950" */
    uint32_t compute_units;
    uint32_t clock_khz;                /* peak engine clock */
    uint32_t lds_bytes_per_workgroup;  /* opt-in maximum */
    uint64_t global_memory_bytes;
    double   fp32_vector_peak_tflops;  /* compute_units * 4 SIMDs * 32 FMA lanes * 2 flop * clock */
} rt_device_info;

typedef struct rt_context rt_context;

uint32_t rt_abi_version(void);
const char* rt_status_string(rt_status status);

/* Context = one GPU.  Replaces the thread pool set-up of main.cpp:450-454. */
rt_status rt_create(int device_ordinal, rt_context** out_context);
void rt_destroy(rt_context* context);
/* Message of the last failure on this context ("" if none). */
const char* rt_error_string(const rt_context* context);
rt_status rt_get_device_info(const rt_context* context, rt_device_info* out_info);

/* Fills *params with the reference's constants for a full-image render. */
void rt_default_params(rt_render_params* params, uint32_t width, uint32_t height, uint32_t samples_per_pixel);
/* Number of rows a render with these parameters produces / their global index. */
uint32_t rt_local_row_count(const rt_render_params* params);
uint32_t rt_global_row(const rt_render_params* params, uint32_t local_row);

/* Copies the world to the device (the arrays the reference's GPU front end
 * uploads raw, win32gpu/main_win32.cpp:227-253).  The caller keeps ownership of
 * its arrays; a later call replaces the world. */
rt_status rt_upload_world(rt_context* context, const rt_world* world);

/* Renders the selected rows, as running every WorkOrder through RaytraceWork
 * would (main.cpp:336-400), and returns when the outputs are in host memory.
 *   pixels   : local_rows*width uint32 0xAARRGGBB sRGB (main.cpp:393), or NULL
 *   radiance : local_rows*width*3 float, linear colour/spp before tonemapping, or NULL
 *   rays     : number of world intersections executed (main.cpp:278,397), or NULL */
rt_status rt_render(rt_context* context, const rt_render_params* params,
                    uint32_t* pixels, float* radiance, uint64_t* rays);

/* Same, with caller-owned DEVICE buffers, asynchronous on `hip_stream`
 * (a hipStream_t; NULL = the default stream).  *d_rays is zeroed by the call and
 * then accumulated into.  Allocates nothing (but see RT_FLAG_COMPACT_TAILS / rt_reserve) and changes no
 * kernel attribute: the call only enqueues kernels on the stream, so it may be recorded into a HIP graph
 * (stream capture) and replayed. */
rt_status rt_render_device(rt_context* context, const rt_render_params* params,
                           uint32_t* d_pixels, float* d_radiance, uint64_t* d_rays,
                           void* hip_stream);

/* rt_render_device for callers that cannot allocate device memory at all (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc): same arguments, same kernels, same results bit
 * for bit, but the workspace of an RT_FLAG_COMPACT_TAILS render is a fixed 32 MiB array that is part of
 * the library's code object (one copy per GPU, held until the process ends) instead of memory the context
 * allocates, so the call never allocates or frees anything and needs no rt_reserve.  It has no counterpart
 * in the reference.  Limit: with RT_FLAG_COMPACT_TAILS at most 2^20 pixels and 2^20 streams
 * (local rows x width x lanes_per_pixel) per call, e.g. 1024x1024 at one lane per pixel or 640x400 at four;
 * larger renders return RT_ERR_TOO_LARGE (without the flag there is no workspace and no limit).  The pool
 * of handed-over streams takes what is left of the array (196,000 entries); when a render has more waves
 * than that allows at the usual hand-over threshold, the threshold is lowered, which changes the schedule
 * only.  Every call on a GPU shares that GPU's array, whatever its context or stream: calls that overlap
 * in time on one GPU race on it and give wrong pixels; run them one after another. */
rt_status rt_render_device_noalloc(rt_context* context, const rt_render_params* params,
                                   uint32_t* d_pixels, float* d_radiance, uint64_t* d_rays,
                                   void* hip_stream);

/* Allocates now whatever device workspace rt_render_device would allocate on its first call with these
 * parameters (RT_FLAG_COMPACT_TAILS), so that the render calls themselves allocate nothing. */
rt_status rt_reserve(rt_context* context, const rt_render_params* params);

/* Sets the running sum that RT_FLAG_ACCUMULATE renders through rt_render add to (a sum loaded back from a
 * checkpoint): local_rows*width*3 floats, the rows of `params` in the compact order rt_render returns. */
rt_status rt_load_accumulator(rt_context* context, const rt_render_params* params, const float* radiance_sum);

/* Multi-GPU render into HOST memory: the rows [row_begin,row_end) of `params` are cut into bands of
 * 8 rows, context i renders bands i, i + n, ... concurrently (one host thread per context, the
 * replacement of the reference's worker threads, main.cpp:450-461, at GPU granularity) and the
 * bands are interleaved back into full-size outputs: (row_end-row_begin) x width pixels / radiance,
 * *rays = sum over GPUs.  Every context must hold the same world.  params->band_* are ignored.
 * The same context may appear more than once (its shares are then rendered one after another), except with
 * RT_FLAG_OUTPUT_SUM / RT_FLAG_ACCUMULATE: a context keeps one running sum, so that is refused. */
rt_status rt_render_multi(rt_context* const* contexts, uint32_t context_count, const rt_render_params* params,
                          uint32_t* pixels, float* radiance, uint64_t* rays);

/* Adds `part_count` (1..64) radiance sums of `value_count` floats each, part k starting at
 * d_sums + k * part_stride, into d_out (which may be part 0) in a FIXED order: neighbouring parts first
 * (0+1, 2+3, ...), then neighbouring pairs, left operand first - the order in which the render kernel adds
 * the lanes_per_pixel streams of a pixel.  This is how the sample-sharded multi-GPU mode composes its
 * image: the per-GPU sums (RT_FLAG_OUTPUT_SUM renders with batch_offset = gpu * lanes_per_pixel) are
 * gathered and added here, so that N = 2^k GPUs give the image of one GPU rendering N * lanes_per_pixel
 * streams per pixel, bit for bit, whatever order a transport's reduction would have used.  The reference
 * has one process and no counterpart; its row sums are serial (main.cpp:379-392).  Device pointers,
 * asynchronous on hip_stream. */
rt_status rt_combine_device(rt_context* context, const float* d_sums, uint32_t part_count, uint64_t part_stride,
                            uint64_t value_count, float* d_out, void* hip_stream);

/* Tonemap + pack of an accumulated radiance sum (sum over passes or GPUs of
 * RT_FLAG_OUTPUT_SUM renders): pixels[i] = pack(sum[i] / total_samples), the
 * epilogue of main.cpp:393.  Device pointers, asynchronous on hip_stream. */
rt_status rt_resolve_device(rt_context* context, const float* d_radiance_sum, uint64_t pixel_count,
                            uint32_t total_samples, uint32_t* d_pixels, float* d_radiance_mean,
                            void* hip_stream);

/* Same with HOST buffers, for callers that keep the accumulated sum in host memory (a progressive
 * render that adds one pass after another, a sum loaded back from a checkpoint file).  pixels and
 * radiance_mean may each be NULL.  Returns when the outputs are written. */
rt_status rt_resolve(rt_context* context, const float* radiance_sum, uint64_t pixel_count,
                     uint32_t total_samples, uint32_t* pixels, float* radiance_mean);

/* Duration of the last render kernel launched with RT_FLAG_TIME_KERNEL, measured
 * with HIP events on the kernel's own stream.  Blocks until that kernel finished. */
rt_status rt_last_kernel_ms(rt_context* context, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
