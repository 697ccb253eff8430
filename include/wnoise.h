/*
 * wnoise.h -- C ABI of libwnoise_hip.so: MI355X (gfx950) evaluation of the reference's
 * per-sample wavelet / Perlin noise path.
 *
 * This is the drop-in boundary.  The reference (Jason9339/Wavelet-Noise-in-ray-tracing) has no
 * FFI: its boundary is the C++ class surface of WaveletNoise.h, perlin.h,
 * experient/PerlinNoise.hpp and texture.h.  Every entry point below names the reference
 * interface (file:line, relative to the reference checkout) it stands under; the C++ host
 * classes in wavelet-noise-in-ray-tracing_amd/host/ keep the reference's class names and
 * signatures and forward to these functions (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ / torch types.
 *   - Every function returns a wn_status (0 = WN_OK).  C cannot throw: on failure the message
 *     is kept per thread and read with wn_last_error().  There is NO CPU fallback: without a
 *     HIP device every compute entry point fails with WN_ERR_NO_DEVICE.
 *   - Pointers named *_dev are device (HBM) pointers, *_host are host pointers.  Caller owns
 *     all output buffers.  `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     compute entry points only enqueue work on it and never synchronise.
 *   - Handles (wn_tile, wn_perm) are immutable after creation, so evaluation is re-entrant,
 *     like the reference's const evaluate* / noise() members, and entry points may be called
 *     from several host threads and for several devices (per-device facts are kept in
 *     mutex-protected tables; a handle must be used on the device it was created on, otherwise
 *     WN_ERR_INVALID).
 *   - rand(): the HIP runtime draws from the C library's global rand() state while it launches.
 *     Every entry point parks the application's state while any thread is inside the library
 *     (depth-counted under a mutex) and restores it when the last call returns, so a caller
 *     that uses rand() between calls (the reference's renderer, main.cpp:184-185) sees the
 *     stream it would see without the library.  An application thread that calls rand()
 *     CONCURRENTLY with another thread's ABI call draws from the library's private state:
 *     rand() is one process-global stream.
 *   - Value-level conventions kept from the reference: an empty tile evaluates to 0.0f
 *     (WaveletNoise.cpp:112,186,219); an odd tile size is bumped to the next even size
 *     (WaveletNoise.cpp:22-25); NaN / |coordinate| >= 2^31 inputs are undefined as in the
 *     reference (unguarded float->int casts).
 *   - Numerics: Perlin (fp64) and every "points" / WN_GRID_EXACT wavelet path keep the
 *     reference's operation order with FMA contraction off and are bit-identical to it.  The
 *     default dense-grid wavelet path evaluates the same B-spline sum separably (x after y
 *     after z) and agrees within 1e-5 absolute (BASELINE.json north_star tolerance).
 */
#ifndef WNOISE_H
#define WNOISE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_API __attribute__((visibility("default")))

typedef enum wn_status {
    WN_OK = 0,
    WN_ERR_INVALID = 1,   /* bad argument (NULL handle, negative size, ...) */
    WN_ERR_NO_DEVICE = 2, /* no HIP device / runtime: the product has no CPU path */
    WN_ERR_HIP = 3,       /* a HIP call failed; wn_last_error() has hipGetErrorString */
    WN_ERR_ALLOC = 4
} wn_status;

WN_API const char *wn_last_error(void);
WN_API const char *wn_version(void);

/* ---- device plumbing (so a host without torch can drive the library) ------------------------ */
WN_API int wn_device_count(int *count);
WN_API int wn_device_set(int ordinal);
WN_API int wn_device_get(int *ordinal);
WN_API int wn_device_info(char *name, size_t name_len, int *compute_units, size_t *hbm_bytes);
WN_API int wn_dev_alloc(void **dptr, size_t bytes);
WN_API int wn_dev_free(void *dptr);
/* Host memory the device can address (pinned + mapped): `*dev_alias` is the same bytes seen from
 * kernels.  (Scalar value(p) calls go through wn_scalar_* below, which keep their own mailbox.) */
WN_API int wn_host_alloc_mapped(void **host_ptr, void **dev_alias, size_t bytes);
WN_API int wn_host_free_mapped(void *host_ptr);
WN_API int wn_copy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
WN_API int wn_copy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
WN_API int wn_stream_sync(void *stream);

/* HIP-event stopwatch on `stream` (bench.py measures the kernels on the stream they run on). */
typedef struct wn_timer wn_timer;
WN_API int wn_timer_create(wn_timer **t);
WN_API int wn_timer_start(wn_timer *t, void *stream);
WN_API int wn_timer_stop(wn_timer *t, void *stream);
WN_API int wn_timer_elapsed_ms(wn_timer *t, float *ms); /* synchronises on the stop event */
WN_API void wn_timer_destroy(wn_timer *t);

/* ---- setup streams: libstdc++ <random>, exactly the calls the reference makes --------------- */
/* mt19937(seed) + normal_distribution<float>(0,1), `count` draws in order
 * (WaveletNoise.cpp:21, 74-77, 146-147).  Host side: the stream is libstdc++-defined. */
WN_API int wn_gaussian_fill(uint32_t seed, size_t count, float *out_host);
/* iota(0..255), std::shuffle(mt19937(seed)), duplicated to 512 (perlin.h:34-39,
 * experient/PerlinNoise.hpp:29-34). */
WN_API int wn_perlin_permutation(uint32_t seed, int out512_host[512]);

/* ---- coefficient tiles (class WaveletNoise state, WaveletNoise.h:43-48) --------------------- */
typedef struct wn_tile wn_tile;
/* Even size the reference would use for `requested` (WaveletNoise.cpp:22-25). */
WN_API int wn_tile_even_size(int requested);
/* Upload ready-made coefficients: n^dims floats, x fastest (idx = x + y*n + z*n*n,
 * WaveletNoise.cpp:209).  n == 0 or coeffs == NULL gives an EMPTY tile (evaluates to 0). */
WN_API int wn_tile_create(int n, int dims, const float *coeffs_host, wn_tile **out);
/* WaveletNoise(n, seed) + generateNoiseTile2D/3D (WaveletNoise.cpp:20-26, 69-108, 142-183):
 * Gaussian fill on the host (wn_gaussian_fill), the separable 32-tap down / 4-tap up passes
 * and the subtraction as HIP kernels.  Bit-identical to the reference tile. */
WN_API int wn_tile_generate(int n, int dims, uint32_t seed, wn_tile **out);
/* The filter half alone, from a caller-supplied Gaussian field of n^dims floats. */
WN_API int wn_tile_generate_from_field(int n, int dims, const float *field_host, wn_tile **out);
WN_API int wn_tile_size(const wn_tile *t);          /* getTileSize(), WaveletNoise.h:40 */
WN_API int wn_tile_dims(const wn_tile *t);
WN_API size_t wn_tile_count(const wn_tile *t);      /* getNoiseCoefficients().size() */
WN_API const float *wn_tile_device_ptr(const wn_tile *t);
WN_API int wn_tile_download(const wn_tile *t, float *out_host); /* getNoiseCoefficients(), :39 */
WN_API void wn_tile_destroy(wn_tile *t);

/* ---- Perlin permutation tables (perlin::p, perlin.h:16) -------------------------------------- */
typedef struct wn_perm wn_perm;
WN_API int wn_perm_create(const int table512_host[512], wn_perm **out);
WN_API int wn_perm_create_seeded(uint32_t seed, wn_perm **out); /* perlin(seed), perlin.h:34 */
WN_API int wn_perm_download(const wn_perm *p, int out512_host[512]);
WN_API void wn_perm_destroy(wn_perm *p);

/* ---- dense grids (the loops of experient/main.cpp:11-129, extended to volumes) ------------- */
/* Sample (x, y, z) of the lattice has coordinate, per axis a with index i_a,
 *     c_a = (((float)i_a / (float)den) * base_range) * octave_scale * post_scale
 * evaluated in float in exactly this order (experient/main.cpp:20-26, 47-54, 102-104).  With
 * z_mode == WN_Z_CONST the third coordinate is z_const for every sample ("sliced" generators,
 * experient/main.cpp:50-54, 122) and the slab is one plane thick.  Output index is
 * x + nx*(y + ny*(z - z0)), i.e. a contiguous z-slab of the volume (shard-friendly). */
enum { WN_Z_LATTICE = 0, WN_Z_CONST = 1 };
enum {
    WN_GRID_DEFAULT = 0,
    WN_GRID_EXACT = 1 /* reference summation order, bit-identical; slower */
};
typedef struct wn_grid {
    int32_t den;        /* divisor of every axis (the reference's imageSize) */
    int32_t nx, ny;     /* extent computed in x and y: indices [0,nx) x [0,ny) */
    int32_t z0, z1;     /* z-planes [z0,z1) computed by this call (ignored for 2-D kernels) */
    float base_range;   /* 4.0f, experient/main.cpp:13 */
    float octave_scale; /* 2^octave, experient/main.cpp:14 */
    float post_scale;   /* 2.0f wavelet (experient/main.cpp:24-25), 1.0f Perlin */
    int32_t z_mode;     /* WN_Z_LATTICE | WN_Z_CONST */
    float z_const;      /* final third coordinate when z_mode == WN_Z_CONST */
    float out_scale;    /* result multiplied by this in float (inv_stddev, main.cpp:16,28) */
    int32_t flags;      /* WN_GRID_* */
} wn_grid;

/* evaluate3D over the lattice (WaveletNoise.cpp:185-215 under experient/main.cpp:38-64). */
WN_API int wn_eval3d_grid(const wn_tile *tile3d, const wn_grid *g, float *out_dev, void *stream);
/* evaluate2D over the lattice (WaveletNoise.cpp:111-140 under experient/main.cpp:11-36). */
WN_API int wn_eval2d_grid(const wn_tile *tile2d, const wn_grid *g, float *out_dev, void *stream);
/* evaluate3DProjected, one normal for the whole grid (WaveletNoise.cpp:218-265 under
 * experient/main.cpp:66-93). */
WN_API int wn_eval3d_projected_grid(const wn_tile *tile3d, const wn_grid *g,
                                    const float normal[3], float *out_dev, void *stream);
/* Cook & DeRose Appendix 2 WMultibandNoise (absent from the reference; normal == NULL branch):
 * for b < nbands while s+first_band+b < 0: q = 2*p*2^(first_band+b); acc += w[b]*evaluate3D(q);
 * acc /= sqrt(sum_b w[b]^2 * var_per_band).  p is the lattice coordinate c above
 * (use octave_scale = post_scale = 1); out_scale multiplies last. */
WN_API int wn_multiband3d_grid(const wn_tile *tile3d, const wn_grid *g, float s, int first_band,
                               int nbands, const float *w_host, float var_per_band,
                               float *out_dev, void *stream);
/* (float) perlin::noise(c_x, c_y, c_z) (perlin.h:42-62 under experient/main.cpp:95-129). */
WN_API int wn_perlin_grid(const wn_perm *perm, const wn_grid *g, float *out_dev, void *stream);
/* RTOW turb(p, depth) on the float lattice point (absent from the reference). */
WN_API int wn_perlin_turb_grid(const wn_perm *perm, const wn_grid *g, int depth, float *out_dev,
                               void *stream);
/* perlin::fractal_noise(p) (perlin.h:75-90) on the float lattice point. */
WN_API int wn_perlin_fractal_grid(const wn_perm *perm, const wn_grid *g, float *out_dev,
                                  void *stream);

/* ---- point lists (the scalar API batched: one call = n calls of the reference member) ------- */
WN_API int wn_eval3d_points(const wn_tile *tile3d, const float *xyz_dev, size_t n, float *out_dev,
                            void *stream); /* evaluate3D, WaveletNoise.h:33 */
WN_API int wn_eval2d_points(const wn_tile *tile2d, const float *xy_dev, size_t n, float *out_dev,
                            void *stream); /* evaluate2D, WaveletNoise.h:32 */
WN_API int wn_eval3d_projected_points(const wn_tile *tile3d, const float *xyz_dev,
                                      const float *normals_dev, size_t n, float *out_dev,
                                      void *stream); /* evaluate3DProjected, WaveletNoise.h:35 */
WN_API int wn_multiband3d_points(const wn_tile *tile3d, const float *xyz_dev, size_t n, float s,
                                 int first_band, int nbands, const float *w_host,
                                 float var_per_band, float *out_dev, void *stream);
/* The normal != NULL branch of the same Appendix-2 function: every band is WProjectedNoise =
 * evaluate3DProjected (WaveletNoise.cpp:218-265); the paper divides by sqrt(sum w^2 * 0.296), the constant the
 * reference uses for its projected grids (experient/main.cpp:72) -- pass it as var_per_band.  `normals_dev`
 * holds one normal per point, or ONE normal for all points when one_normal != 0.  Absent from the reference. */
WN_API int wn_multiband3d_projected_points(const wn_tile *tile3d, const float *xyz_dev,
                                           const float *normals_dev, int one_normal, size_t n, float s,
                                           int first_band, int nbands, const float *w_host,
                                           float var_per_band, float *out_dev, void *stream);
WN_API int wn_perlin_points(const wn_perm *perm, const double *xyz_dev, size_t n,
                            double *out_dev, void *stream); /* noise(x,y,z), perlin.h:42 */
/* noise(const point3&) / turb / fractal_noise on float vec3 points (perlin.h:70-90). */
WN_API int wn_perlin_points_vec3(const wn_perm *perm, const float *xyz_dev, size_t n,
                                 double *out_dev, void *stream);
WN_API int wn_perlin_turb_points(const wn_perm *perm, const float *xyz_dev, size_t n, int depth,
                                 double *out_dev, void *stream);
WN_API int wn_perlin_fractal_points(const wn_perm *perm, const float *xyz_dev, size_t n,
                                    double *out_dev, void *stream);

/* ---- texture adaptor (texture.h), batched over ray hit points -------------------------------- */
/* `active_dev` (may be NULL = all active): one byte per point, 0 = this hit is not on a
 * noise-textured surface.  Inactive points are skipped (their output is left untouched); the
 * kernel compacts active lanes with wavefront ballots before the gather loop.
 * Output: the grey level g with color(g,g,g) == texture::value(u,v,p) (texture.h:17). */
/* wavelet_texture::value, texture.h:67-107 (use_3d selects the :70-85 or :86-99 branch;
 * an empty tile gives the :101 branch = 0.5). */
WN_API int wn_wavelet_texture_points(const wn_tile *tile, int use_3d, double scale, int octave,
                                     const float *xyz_dev, const uint8_t *active_dev, size_t n,
                                     float *grey_dev, void *stream);
/* noise_texture::value, texture.h:37-43. */
WN_API int wn_noise_texture_points(const wn_perm *perm, double scale, int octave,
                                   const float *xyz_dev, const uint8_t *active_dev, size_t n,
                                   float *grey_dev, void *stream);

/* ---- scalar calls: the reference's scalar members, one value per call ------------------------------
 * evaluate2D/3D/3DProjected(p) (WaveletNoise.h:32-35), noise(x,y,z) / noise(point3) / fractal_noise(p)
 * (perlin.h:42-90), texture::value(u,v,p) (texture.h:17) as the reference's callers use them
 * (material.h:72, experient/main.cpp:28,56,85,104,122).  A kernel launch per call costs ~22 us; these
 * entry points hand the request to a resident one-wave kernel through a mailbox in pinned host memory
 * (csrc/wn_mailbox.hip): a few microseconds per call, results bit-identical to the batched entry points.
 * They block until the value is back and are serialised across host threads per device.  The resident
 * kernel ends by itself after 2 ms without a request, and after 20 ms in any case however many requests
 * keep arriving: a device-wide synchronise from another thread (hipDeviceSynchronize, the hipFree inside
 * wn_dev_free / wn_tile_destroy, torch.cuda.synchronize) never waits longer than ~20 ms, also in the middle
 * of a burst of scalar calls.  The next call restarts it (one launch).  depth of kind 1 is 0..64. */
WN_API int wn_scalar_eval3d(const wn_tile *tile3d, const float p[3], float *out);
WN_API int wn_scalar_eval2d(const wn_tile *tile2d, const float p[2], float *out);
WN_API int wn_scalar_eval3d_projected(const wn_tile *tile3d, const float p[3], const float normal[3],
                                      float *out);
WN_API int wn_scalar_perlin(const wn_perm *perm, double x, double y, double z, double *out);
/* kind 0: noise(const point3&), 1: turb(p, depth) (RTOW), 2: fractal_noise(p). */
WN_API int wn_scalar_perlin_vec3(const wn_perm *perm, const float p[3], int kind, int depth, double *out);
WN_API int wn_scalar_wavelet_texture(const wn_tile *tile, int use_3d, double scale, int octave,
                                     const float p[3], float *grey);
WN_API int wn_scalar_noise_texture(const wn_perm *perm, double scale, int octave, const float p[3],
                                   float *grey);
/* Scalar calls served so far and resident-kernel instances started for them (diagnostics). */
WN_API int wn_scalar_stats(unsigned long long *calls, unsigned long long *launches);
/* Waits for the resident instances to end and releases the mailboxes (pinned host memory, streams);
 * later scalar calls start over.  Optional: instances end by themselves. */
WN_API int wn_scalar_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* WNOISE_H */
