/*
 * gsr.h -- C ABI of the MI355X-native 3D-Gaussian-Splatting rasterizer (libgsr_hip.so).
 *
 * This is the drop-in boundary under the Python packages GSWorld imports
 * (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:11-13,22-26,266-267).  Each entry
 * point names the upstream pybind function of the un-vendored CUDA extension it replaces:
 *
 *   gsr_forward            <->  diff_gaussian_rasterization._C.rasterize_gaussians
 *   gsr_forward_batch      <->  the wrapper's double loop of render() calls per step (gs_world_wrapper.py:238-267)
 *   gsr_backward           <->  diff_gaussian_rasterization._C.rasterize_gaussians_backward
 *   gsr_mark_visible       <->  diff_gaussian_rasterization._C.mark_visible
 *   gsr_knn_dist2          <->  simple_knn._C.distCUDA2
 *   gsr_ssim_forward       <->  fused_ssim_cuda.fusedssim
 *   gsr_ssim_backward      <->  fused_ssim_cuda.fusedssim_backward
 *   gsr_transform_gaussians<->  the per-step rigid transform of GSWorld itself
 *                               (gs_world_wrapper.py:110-162,244-265 + gsworld/utils/gs_utils.py:283-385)
 *
 * Conventions: plain pointers and sizes only (no torch types); every data pointer is DEVICE memory on the
 * current HIP device unless marked [host]; all inputs are borrowed, contiguous, float32; `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Functions return 0 on success or a negative
 * GSR_E_* code; gsr_last_error() returns a human-readable message for the calling thread.
 * Work buffers are caller-owned and grown through a resize callback -- the same contract as upstream's
 * std::function<char*(size_t)> lambdas that resize_ a torch byte tensor and hand back its data_ptr.
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_E_INVALID (-1)   /* bad argument (shape / null / unsupported value) */
#define GSR_E_HIP (-2)       /* a HIP runtime call or kernel failed (debug mode checks every launch) */
#define GSR_E_ALLOC (-3)     /* a resize callback returned NULL */
#define GSR_E_OVERFLOW (-4)  /* num_rendered exceeded the binning capacity in no-sync mode */
#define GSR_E_TRUNCATED (-5) /* a cooperative quadrant of the compositor gave up on a hand-off: pixels of the frame are wrong */

#define GSR_TILE 16          /* BLOCK_X = BLOCK_Y of cuda_rasterizer/config.h */
#define GSR_NEAR_PLANE 0.05f /* /root/reference/README.md:33 (stock upstream 0.2f) */

/* GaussianRasterizationSettings (the 13-field NamedTuple) minus the tensors, plus GSWorld's near plane. */
typedef struct GsrSettings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;    /* active degree D, 0..3 */
    int32_t sh_coeffs;    /* M = coefficients stored per Gaussian (shs is (P, M, 3)) */
    int32_t prefiltered;
    int32_t antialiasing;
    int32_t debug;        /* sync + error check after every kernel */
    float near_plane;     /* cull p_view.z <= near_plane; GSR_NEAR_PLANE for GSWorld */
    /* A/B and test selectors; 0 = the library's default everywhere.  They travel with the call -- the library keeps
     * no mutable process state, so renderers on different threads or devices cannot disturb each other.  Every
     * choice produces the same point list and bit-identical image state.
     * binning_path: 0 = global depth sort + counting placement by tile rows (rows up to 256 tiles; by chunks of depth
     *   ranks beyond), 1 = global depth sort + emit + tile-id radix sort (always used for tile grids above 16384 tiles
     *   or wider than 2048 tiles), 2 = unordered binning + per-tile (depth, index) sort in LDS, 3 = global depth sort +
     *   counting placement by chunks of 256 depth ranks.
     * render_variant: 0 = wave-decoupled culling kernel, 1 = LDS-staged per tile (upstream's structure), 2 = batched
     *   tile kernel, 3 = the same with per-quadrant instance culling.
     * render_blocks_per_cu: 1..8 sizes the persistent grid of the compositing kernel (0 = 6). */
    int32_t binning_path;
    int32_t render_variant;
    int32_t render_blocks_per_cu;
    /* depth_sort: 0 = sample sort (5 launches: per-frame plan, compaction + classification, column sums, stable
     *   partition, per-bucket LDS radix), 1 = 3-pass LSD radix sort over all visible Gaussians (10 launches).  Same depth
     *   order. */
    int32_t depth_sort;
    /* render_split: what the compositor does with the quadrants that were costliest in the previous frame on the same
     *   state (image state bit-identical in every mode).  0 = default: inference frames give each of them a workgroup of
     *   its own -- three waves cull the quadrant's candidates, the fourth composites the survivors; 1 = two 8x4 halves on
     *   two waves (measured slower, kept for A/B); 2 = as 0; 3 = one wave per quadrant, always. */
    int32_t render_split;
    /* forward_only: 1 = inference frame (GSWorld's closed loop never runs a backward: gs_world_wrapper.py:266-270 keeps
     * ["render"] only).  The image is bit-identical to forward_only = 0; what changes is the work behind it:
     *   - preprocess writes nothing a backward would read (cov3D, SH clamp flags, tiles_touched) and GsrOutputs.radii
     *     may be NULL;
     *   - instances are binned per SUPER-TILE of 2 x 1 tiles (32 x 16 px) instead of per tile, and a Gaussian is listed
     *     only where it can colour a pixel: its tile rect is the reference's (getRect) cut down to the bounding box of
     *     the ellipse alpha >= 1/255 (a Gaussian below 1/255 everywhere is listed nowhere) -- 0.47 of the reference's
     *     instances to count, place and fetch at config 2.  The compositor applies that per-tile membership test to
     *     every candidate itself, so every pixel composites the depth-ordered list of ITS 16 x 16 tile minus entries
     *     every one of its pixels would have skipped: the same image bit for bit;
     *   - final_T / n_contrib (read by the backward only) are not written.
     * The state buffers of such a frame are NOT valid inputs of gsr_backward, and gsr_state_view / GsrFrameStats then
     * describe those lists (num_rendered = super-tile instances, num_visible = Gaussians listed somewhere).  Tile grids
     * wider or higher than 255 tiles keep per-tile lists. */
    int32_t forward_only;
} GsrSettings;

typedef struct GsrInputs {
    int32_t P;                   /* number of Gaussians */
    const float *background;     /* (3) */
    const float *means3D;        /* (P,3) */
    const float *shs;            /* (P,M,3) or NULL when colors_precomp is given */
    const float *colors_precomp; /* (P,3) or NULL */
    const float *opacities;      /* (P) */
    const float *scales;         /* (P,3) or NULL when cov3D_precomp is given */
    const float *rotations;      /* (P,4) (r,x,y,z), used un-normalised */
    const float *cov3D_precomp;  /* (P,6) or NULL */
    const float *viewmatrix;     /* (16) world->view, element [r][c] at m[c*4+r] */
    const float *projmatrix;     /* (16) full projection, same layout */
    const float *campos;         /* (3) */
    /* Optional split SH storage (forward only).  3DGS keeps the SH coefficients as two parameters, features_dc
     * (P,1,3) and features_rest (P,M-1,3), and upstream's render() concatenates them for every frame
     * (gaussian_renderer/__init__.py `shs = pc.get_features`: a 2 x 192 B/Gaussian copy, 564 MB of traffic at
     * 1.47 M Gaussians).  When shs_rest is not NULL, `shs` is features_dc and `shs_rest` is features_rest; the
     * colours are bit-identical to the concatenated call.  NULL = upstream layout. */
    const float *shs_rest;       /* (P,M-1,3) or NULL */
    /* Optional raw parameter space (forward only): OR of GSR_RAW_* -- the activations upstream's GaussianModel
     * getters apply per frame (scene/gaussian_model.py: torch.sigmoid / torch.exp / F.normalize, three passes over
     * the model) are then evaluated inside preprocess, in the float32 order fixed by oracle/gs_oracle.c
     * (gso_activate_params; exp = 2^n * Cephes polynomial, ~1 ulp).  0 = upstream contract (activated inputs). */
    int32_t param_space;
    /* Optional rigid transform of labelled Gaussians applied INSIDE preprocess (forward only): what
     * gsr_transform_gaussians computes, without materialising transformed (P,.) buffers -- a frame of environment e from
     * camera c reads the one base model plus environment e's 17-float pose table (see gsr_transform_gaussians below for
     * the table and the arithmetic; the results are bit-identical to transforming first and rendering the outputs).
     * part_labels NULL = off.  part_rescale ((K) bytes or NULL) marks the parts whose log-scales the reference rewrites
     * (needs GSR_RAW_SCALES). */
    const float *part_labels;      /* (P) semantic label per Gaussian, or NULL */
    const int32_t *part_lut;       /* (part_lut_size) label -> part index, -1 = not a moving part */
    int32_t part_lut_size;
    const float *part_transforms;  /* (part_count, 17) */
    int32_t part_count;
    const uint8_t *part_rescale;   /* (part_count) or NULL */
    /* Optional view-frustum culling by blocks of 256 consecutive Gaussians (gsworld_amd/layout.py builds both arrays
     * once per model).  The reference projects every Gaussian in front of the near plane and only then finds that its
     * tile rect is empty (forward.cu preprocessCUDA / getRect): at configs[1] that is 88 % of the model.  cull_blocks
     * holds, per block b of Gaussians [256 b, 256 b + 256), eight floats: the axis-aligned box of their centres
     * (lo.xyz, hi.xyz), rho >= sqrt(largest eigenvalue of every 3D covariance in the block, at scale_modifier 1) and
     * the block's common part label (NaN: labels differ -- such a block is never culled while part_labels is given).
     * A block is skipped only when the box, moved by its part's pose and grown by a rigorous bound of the projected
     * 3-sigma radius, lies wholly behind the near plane or beside the tile grid -- every Gaussian in it then has
     * radii == 0 in the reference as well, so images, radii and lists are bit-identical (preprocess.hip
     * prep_block_culled).  The bounds are the caller's promise: a box that does not contain its Gaussians drops them.
     * Correct for ANY order of the model; tight when neighbours in the model are neighbours in space, hence:
     * orig_index ((P) or NULL): the model arrays are a PERMUTED copy of the caller's model (Morton order per part) and
     * Gaussian i of them is number orig_index[i] of the original.  `radii` is written in ORIGINAL numbering and depth
     * ties resolve by original number, exactly as without the permutation, so the image is the same bit for bit; the
     * opaque state (and the point list in it) stays in the numbering of the permuted arrays.  forward_only frames on the
     * default sort / placement path only. */
    const float *cull_blocks;      /* (ceil(P / 256), 8) or NULL */
    const int32_t *orig_index;     /* (P) or NULL */
} GsrInputs;

#define GSR_RAW_OPACITY 1   /* opacities are logits:        opacity  = 1 / (1 + exp(-x)) */
#define GSR_RAW_SCALES 2    /* scales are log-scales:        scale    = exp(x) */
#define GSR_RAW_ROTATIONS 4 /* rotations are un-normalised:  rotation = q / max(|q|, 1e-12) */
/* Bits 8..31 of param_space: a MODEL VERSION, or 0.  Nonzero is the caller's promise that the model arrays (means3D, shs, shs_rest,
 * opacities, scales, rotations), part_labels, part_lut, part_rescale, cull_blocks and orig_index hold exactly what they held in the
 * previous frame on this state that carried the same version -- pick a fresh random value whenever any of them is written.
 * Inference frames (GsrSettings.forward_only, default path, radii == NULL) with part_labels and cull_blocks then leave a block of
 * 256 Gaussians as the previous frame on the state computed it when nothing the block's records depend on has changed: settings,
 * camera (view, projection, centre: bit for bit) and the pose row of the block's one part (csrc/preprocess.hip
 * prep_block_cached).  A fixed sensor camera over a scene in which only the robot moves recomputes the robot.  Frames are the same
 * bit for bit; the promise is the caller's: arrays changed under an unchanged version give stale blocks. */
#define GSR_MODEL_VERSION(v) ((int32_t)(((uint32_t)(v) & 0xFFFFFFu) << 8))
/* Bit 3 of param_space, read only beside a model version: the caller's promise that GsrOutputs.out_rgb8 still holds, byte for
 * byte, what the previous frame on this state wrote there (nobody drew into it, nobody else rendered into it).  An inference
 * frame whose ONLY output is out_rgb8 (out_color == out_invdepth == NULL) then leaves a 16 x 16 tile as it stands when camera,
 * background, settings and buffer are the previous frame's and no Gaussian of a recomputed block touches the tile now or touched
 * it then: its list holds the same records in the same order, so the compositor would write the bytes that are there
 * (csrc/render.hip "tile reuse"; preprocess marks the tiles of every recomputed Gaussian).  Frames are the same bit for bit. */
#define GSR_FRAME_KEPT 8

typedef struct GsrOutputs {
    /* The two float images.  Both may be NULL for an inference frame (GsrSettings.forward_only on the default path) that
     * hands over out_rgb8: GSWorld keeps the uint8 frame only (gs_world_wrapper.py:266-270 -- ["render"], x 255, clamp,
     * uint8), and 16 bytes per pixel that nobody reads are then not written (4.9 MB per 640 x 480 frame). */
    float *out_color;    /* (3,H,W) */
    float *out_invdepth; /* (1,H,W) */
    int32_t *radii;      /* (P) */
    /* Optional: the frame as GSWorld consumes it (gs_world_wrapper.py:268-270), (H,W,3) uint8 =
     * (uint8)clamp(255 * out_color, 0, 255), written by the compositor itself instead of a second pass over the
     * image (gsr_pack_rgb8 remains for callers that convert later).  NULL = not wanted. */
    uint8_t *out_rgb8;
    /* Optional: two words the HOST can read and the device can write -- the device-visible address of pinned host memory
     * (gsr_pinned_device_address) or plain device memory.  The frame's capacity check leaves there what GsrFrameStats
     * reports as overflow_frames, as (0x0F10F10F, count) -- (anything else, -) while the state has never overflowed:
     * a caller that runs ahead of the device learns of an overflowed frame without a copy behind every frame (8 bytes,
     * but a copy: ~3 us of the stream per frame) and without a wait.  NULL = not wanted. */
    uint32_t *overflow_mirror;
} GsrOutputs;

/* resize callback: must return a device pointer to at least `bytes` bytes (256-byte aligned), or NULL. */
typedef char *(*gsr_resize_fn)(void *user, size_t bytes);

typedef struct GsrBuffers {
    gsr_resize_fn geom_resize;    void *geom_user;    /* geometry state   (per Gaussian) */
    gsr_resize_fn binning_resize; void *binning_user; /* binning state    (per rendered instance) */
    gsr_resize_fn image_resize;   void *image_user;   /* image state      (per pixel / per tile) */
} GsrBuffers;

/* Frame statistics, all produced by the pipeline itself (N, V, R). */
typedef struct GsrFrameStats {
    int64_t num_visible;  /* V: Gaussians with radii > 0 */
    int64_t num_rendered; /* R: sum of tiles touched */
    int32_t overflow;     /* 1 if R exceeded the binning capacity (no-sync mode only) */
    int32_t overflow_frames; /* how many frames rendered on this geometry state overflowed since the first 256 bytes of
                              * the buffer were last zeroed (this package's resize callbacks zero them on new storage;
                              * a header that never held a count reads as 0) -- a rollout that never synchronises reads
                              * it once at the end to learn whether every frame was valid.  Occupies former padding. */
    int32_t truncated;       /* 1: a cooperative quadrant of THIS frame's compositor timed out waiting for a hand-off between
                              * its waves and was written truncated (cannot happen by the counters' construction; if it ever
                              * does the frame says so instead of showing a wrong quadrant silently: re-render) */
    int32_t coop_timeouts;   /* such quadrants in all frames on this state since the header was last zeroed */
} GsrFrameStats;

const char *gsr_last_error(void);
const char *gsr_version(void);
/* ABI handshake for bindings that mirror the structs above (ctypes, the compiled torch extension): sizes of
 * GsrSettings, GsrInputs, GsrOutputs, GsrBuffers, GsrBackwardInputs, GsrGrads as THIS library was compiled. */
void gsr_abi_sizes(int32_t out[6]);

/* Bytes needed for each state buffer. */
size_t gsr_geom_bytes(int32_t P, int32_t width, int32_t height);
size_t gsr_binning_bytes(int64_t r_capacity);
size_t gsr_image_bytes(int32_t width, int32_t height);

/*
 * Forward rasterization (upstream RasterizeGaussiansCUDA -> CudaRasterizer::Rasterizer::forward).
 *
 * r_capacity <= 0 : exact mode.  One host read-back of num_rendered in the middle of the frame (what
 *                   upstream does), binning state sized exactly; `stats` (may be NULL) is filled.
 * r_capacity  > 0 : no-sync mode.  Binning state sized for r_capacity instances, nothing is read back and the
 *                   call returns as soon as the work is enqueued.  Call gsr_frame_stats() later to learn
 *                   V, R and whether R overflowed the capacity (then the image is invalid: re-run bigger).
 */
int gsr_forward(const GsrSettings *settings, const GsrInputs *in, const GsrOutputs *out,
                const GsrBuffers *buffers, int64_t r_capacity, GsrFrameStats *stats, void *stream);

/*
 * B frames of ONE step in one set of launches: what GSWorld's render loop asks for per simulation step --
 * `for cam_name, cam_param in self.camera_params.items(): for i in range(self.num_envs): render(...)`
 * (gs_world_wrapper.py:238-267; SURVEY.md 8f-4 "render_batch(cameras[B])").  Frame k is described by settings[k], in[k],
 * out[k], buffers[k], r_capacity[k] exactly as gsr_forward takes them: its own camera, pose table, outputs and state
 * buffers; the model arrays are usually shared.  Consecutive frames that (a) take the default path (GsrSettings selectors
 * 0) with a capacity (r_capacity[k] > 0: no mid-frame read-back), (b) have the same P, image size, SH layout and
 * forward_only flag, go through every stage TOGETHER: each of the frame's eleven kernels is launched once with a grid
 * that spans the frames (the per-frame argument blocks travel in the kernarg segment, at most GSR_MAX_FRAMES_PER_LAUNCH
 * frames per set of launches; longer runs are cut into such sets).  The stages of a 640 x 480 frame that are chains of
 * dependent round trips on a few hundred workgroups (compaction, bucket sort, range scans) then fill the chip with
 * B x the workgroups instead of being overlapped by hand on B streams, and the compositor's tail -- its costliest
 * quadrants running alone -- is filled by the next frame's workgroups.  Any other frame (exact mode, A/B selectors,
 * P == 0, another image size) runs by itself as gsr_forward would run it, in the order given; every frame's outputs and
 * state are bit-identical to a gsr_forward call with the same arguments.  Nothing is read back; gsr_frame_stats() on each
 * frame's geometry state tells V, R and overflow afterwards.
 */
#define GSR_MAX_FRAMES_PER_LAUNCH 8
int gsr_forward_batch(int32_t B, const GsrSettings *settings, const GsrInputs *in, const GsrOutputs *out,
                      const GsrBuffers *buffers, const int64_t *r_capacity, void *stream);

/*
 * How gsr_forward would run a frame with these settings on a model of P Gaussians (`permuted`: GsrInputs.orig_index
 * given; r_capacity as for gsr_forward) -- the library's own decision, made on the host without touching a device:
 * out[0] = binning mode (1 = depth sort + counting placement, 2 = bin-then-sort, 0 = depth sort + radix), out[1] = the
 * counting placement (1 by tile rows, 2 by chunks, 0 neither), out[2] = 1: inference frame (forward_only honoured),
 * out[3] = 1: super-tile lists, out[4] = 1: lean state layout, out[5] = 1: LSD radix depth sort (depth_sort = 1, or a
 * model beyond the sample sort's 8 388 608 Gaussians), out[6] = 1: the compositor's order is dealt beside the depth
 * sort, out[7] = 1: exact mode.  Returns GSR_E_INVALID where gsr_forward would (a permuted model on a path that does not
 * take one).
 */
int gsr_plan_query(const GsrSettings *settings, int32_t P, int32_t permuted, int64_t r_capacity, int32_t out[8]);

/* Tuning aid: cycle stamps of the depth-sort kernels (meaningful in builds with -DGSR_SS_TIMING only).  A negative
 * height reads the state of a forward_only frame (the lean layout). */
int gsr_debug_ss_stamps(int32_t P, int32_t width, int32_t height, const void *geom, uint64_t *out64);

/* Tests and tools: what the sample sort of the last frame on `geom` did (synchronises `stream`).  out[0] = 1: it took
 * the kept splitters unchecked (fixed camera, scene standing still); out[1] = 1: it drew new splitters from samples;
 * (both 0: it checked the kept ones against samples and kept them); out[2] = 1: a depth bucket came out above what
 * quantiles of an unchanged scene give; out[3]: consecutive balanced frames on kept splitters before this one; out[4]:
 * depth buckets of the frame; out[5] = 2: it took every second entry of a kept table of twice as many quantiles;
 * out[6]: quadrants the frame's compositor handed to cooperative workgroups; out[7]: reserved (0). */
int gsr_debug_sort_state(const void *geom, int32_t out[8], void *stream);

/* Reads V / R / overflow of the frame whose geometry state is `geom` (synchronises `stream`).  Returns GSR_E_OVERFLOW for a
 * frame that exceeded its capacity, GSR_E_TRUNCATED for one with a timed-out cooperative quadrant (stats filled either way). */
int gsr_frame_stats(const void *geom, GsrFrameStats *stats, void *stream);

/* Views into the opaque state, for tests and tools (device pointers into the caller's buffers). */
typedef struct GsrStateView {
    const float *splat;           /* (P,12): xy, depth, 1/depth | conic xyz, opacity | rgb, radius */
    const float *cov3D;           /* (P,6) */
    const uint8_t *clamped;       /* (P,4) */
    const uint32_t *tiles_touched;/* (P) */
    const uint16_t *rects;        /* (P,4) min.x min.y max.x max.y */
    const uint32_t *depth_order;  /* (V) Gaussian indices, ascending (depth bits, index) */
    const uint32_t *point_list;   /* (R) Gaussian index per instance, tile-major / depth / index order */
    const uint32_t *point_tiles;  /* always NULL: tile ids follow from `ranges` */
    const uint32_t *ranges;       /* (tiles,2) */
    const float *final_T;         /* (H*W) */
    const uint32_t *n_contrib;    /* (H*W) */
} GsrStateView;
int gsr_state_view(int32_t P, int32_t width, int32_t height, int64_t r_capacity_or_R, const void *geom,
                   const void *binning, const void *image, GsrStateView *view);

/*
 * Optional stage timing with HIP events recorded on the caller's stream (what bench.py's roofline uses).
 * mode 0 = off, 1 = the compositing kernel only (2 events / frame), 2 = every stage (6 events / frame).
 * Stages: 0 preprocess, 1 depth sort (compaction, partition, buckets), 2 tile counts -> ranges (band ranges, count,
 * scan, tile starts; the fallback paths: tile offsets), 3 placement of the instances (fallback: emit + tile sort +
 * ranges), 4 render.
 * The recorder is per calling thread (enable, render and collect from the same thread); at most 4096 frames
 * are recorded between two collects.
 */
#define GSR_PROFILE_STAGES 5
typedef struct GsrProfile {
    int32_t frames;
    double stage_ms[GSR_PROFILE_STAGES]; /* summed over `frames` */
} GsrProfile;
int gsr_profile_enable(int mode);
int gsr_profile_collect(GsrProfile *out);

/*
 * Backward rasterization (upstream RasterizeGaussiansBackwardCUDA -> Rasterizer::backward).
 * `settings` / `in` are the forward's; the three state pointers are the buffers the forward filled (exact mode:
 * binning state carved for num_rendered instances).  Every gradient buffer is zeroed by the call.
 */
typedef struct GsrBackwardInputs {
    const float *dL_dout_color;    /* (3,H,W) */
    const float *dL_dout_invdepth; /* (1,H,W) or NULL */
    const int32_t *radii;          /* (P) from the forward */
    int64_t num_rendered;          /* R from the forward */
    const void *geom, *binning, *image;
} GsrBackwardInputs;

typedef struct GsrGrads {
    float *dL_dmeans2D;   /* (P,3)  pixel-NDC units, z unused */
    float *dL_dcolors;    /* (P,3) */
    float *dL_dopacity;   /* (P) */
    float *dL_dmeans3D;   /* (P,3) */
    float *dL_dcov3D;     /* (P,6) */
    float *dL_dsh;        /* (P,M,3) or NULL with colors_precomp */
    float *dL_dscales;    /* (P,3) or NULL with cov3D_precomp */
    float *dL_drots;      /* (P,4) or NULL with cov3D_precomp */
    float *dL_dconic;     /* unused (may be NULL): the compositor's per-Gaussian sums live in the geometry state */
    float *dL_dinvdepths; /* unused (may be NULL) */
    float *dL_dsh_rest;   /* (P,M-1,3) with GsrInputs.shs_rest (dL_dsh is then the (P,1,3) dc part), else NULL */
} GsrGrads;

int gsr_backward(const GsrSettings *settings, const GsrInputs *in, const GsrBackwardInputs *bw,
                 const GsrGrads *grads, void *stream);

/*
 * simple_knn._C.distCUDA2(points (P,3)) -> (P): mean of the squared distances to the 3 nearest OTHER points
 * (exact; self excluded by index, duplicates give 0).  `workspace` needs gsr_knn_workspace_bytes(P) bytes.
 */
size_t gsr_knn_workspace_bytes(int32_t P);
int gsr_knn_dist2(int32_t P, const float *points, float *mean_dist2, void *workspace, size_t workspace_bytes,
                  void *stream);

/*
 * fused_ssim_cuda.fusedssim / fusedssim_backward: images are (B,CH,H,W) contiguous; 11x11 Gaussian window
 * (sigma 1.5), zero "same" padding.  With train != 0 the three partial maps needed by the backward are stored.
 * The backward differentiates w.r.t. img1 only (as upstream).
 */
int gsr_ssim_forward(int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, const float *img1,
                     const float *img2, int32_t train, float *ssim_map, float *dm_dmu1, float *dm_dsigma1_sq,
                     float *dm_dsigma12, void *stream);
int gsr_ssim_backward(int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, const float *img1,
                      const float *img2, const float *dL_dmap, const float *dm_dmu1, const float *dm_dsigma1_sq,
                      const float *dm_dsigma12, float *dL_dimg1, void *stream);

/*
 * The photometric loss of the 3DGS training step, (1 - lambda) mean|x - t| + lambda (1 - mean ssim(x, t)) with
 * x = clamp(img, 0, 1) when clamp01 (what train.py builds from l1_loss, fused_ssim and render()'s clamp out of ~25
 * elementwise / reduction launches and their autograd), in three launches: the ssim forward pass with the sums folded
 * in, a one-workgroup sum in a fixed order (binary64), and -- gsr_photometric_loss_backward, after a forward with
 * keep_for_backward != 0 on the same scratch -- the ssim backward pass with the L1 sign term, the clamp mask and the
 * incoming gradient of the loss (grad_loss: one float on the device, NULL = 1) folded in.  img / target / dL_dimg:
 * (planes, H, W) float32; loss: one float on the device; dL_dimg = grad_loss x d loss / d img.  scratch:
 * gsr_photometric_loss_scratch_floats(planes, H, W) floats.
 * An extension (no upstream entry point); gsr_ssim_* above remain the drop-in for fused_ssim_cuda.
 */
size_t gsr_photometric_loss_scratch_floats(int32_t planes, int32_t H, int32_t W);
int gsr_photometric_loss(int32_t planes, int32_t H, int32_t W, const float *img, const float *target,
                         float lambda_dssim, int32_t clamp01, float *scratch, float *loss, int32_t keep_for_backward,
                         void *stream);
int gsr_photometric_loss_backward(int32_t planes, int32_t H, int32_t W, const float *img, const float *target,
                                  float lambda_dssim, int32_t clamp01, const float *scratch, const float *grad_loss,
                                  float *dL_dimg, void *stream);

/*
 * Fused per-step rigid transform of labelled Gaussians -- replaces GSWorld's per-link isin() mask / gather /
 * transform_gaussians / masked scatter passes (gs_world_wrapper.py:110-162,244-265; gs_utils.py:283-385) and the
 * full-model deepcopy with one pass.  label = (int)semantics[i]; k = lut[label] (or -1 / out of range: copy through);
 * transforms[k] = 17 floats: R row-major (9), t (3), uniform scale (1), quaternion of R as (w,x,y,z) (4).
 *   xyz' = R (scale * xyz) + t;   rot' = standardize(q_R (x) rot/|rot|) * |rot|.
 */
int gsr_transform_gaussians(int32_t P, const float *xyz, const float *rot, const float *semantics,
                            const int32_t *lut, int32_t lut_size, const float *transforms, int32_t K,
                            float *xyz_out, float *rot_out, void *stream);

/*
 * The same for E environments at once (gs_world_wrapper.py:239-242 renders `for i in range(num_envs)`, and
 * transform_gaussians returns (B,N,.) tensors for a batch of B poses): `transforms` is (E,K,17), the outputs are
 * (E,P,3) / (E,P,4), environment e under pose table e.  Optional scaling / scaling_out ((P,3) in, (E,P,3) out; both or
 * neither): the log-scale parameter of the parts flagged in `rescale` ((K) bytes, NULL = none) is rewritten the way the
 * reference rewrites it when it is given a per-environment scale vector (gs_utils.py: `inverse_sigmoid(exp(scaling) *
 * scale)` = log(x / (1 - x)), sic -- the tracked actors of the wrapper, gs_world_wrapper.py:146-157), copied otherwise.
 */
int gsr_transform_gaussians_batch(int32_t P, int32_t E, const float *xyz, const float *rot, const float *scaling,
                                  const float *semantics, const int32_t *lut, int32_t lut_size,
                                  const float *transforms, int32_t K, const uint8_t *rescale, float *xyz_out,
                                  float *rot_out, float *scaling_out, void *stream);

/*
 * Builds the 17-float transform table on the DEVICE from K row-major 4x4 rigid matrices (and optional uniform
 * scales, NULL = 1): the link / actor poses a GPU simulator already holds as device tensors
 * (gs_world_wrapper.py:118-120,146-156 compute them with torch on the simulation device).  The quaternion is
 * ManiSkill's matrix_to_quaternion (real part first, non-negative) in its float32 operation order.
 */
int gsr_pack_part_transforms(int32_t K, const float *matrices, const float *scales, float *table, void *stream);

/*
 * One closed-loop step's host values in ONE launch (gs_world_wrapper.py:110-162 computes the part poses per step, :238 the
 * cameras; a CPU simulator leaves both on the host): `src` is the DEVICE-VISIBLE address of pinned host memory
 * (gsr_pinned_device_address of a hipHostMalloc / torch pin_memory buffer), n floats; they are copied to `dst` (device),
 * and the K row-major 4x4 part matrices at src + mat_off (uniform scales at src + scale_off; scale_off < 0: all 1) are
 * packed into the 17-float pose table exactly as gsr_pack_part_transforms packs them.  Nothing but a kernel launch: the
 * call can be captured into a hipGraph, the step's host values then travel INSIDE the step's graph (no copy, no launch
 * between two replays).  The kernel reads the host buffer when it RUNS: keep it unchanged until then (a ring of slots).
 */
int gsr_pinned_device_address(const void *host, void **device);
int gsr_stage_step(int32_t n, const float *src, float *dst, int32_t K, int32_t mat_off, int32_t scale_off,
                   float *table, void *stream);

/* Self-test of the wave reductions used by the backward: out44[w] = sum(in256[64w .. 64w+63]) for the 4 waves, then
 * out44[4 + 10w + c] = sum over the wave's lanes l with l % (c + 2) == 0 of (c + 1) * in256[64w + l]  (ten different
 * per-lane values through the 10-component transpose-reduce). */
int gsr_selftest_wave_sum(const float *in256, float *out44, void *stream);

/* GSWorld's frame conversion (gs_world_wrapper.py:268-270): (3,H,W) float -> (H,W,3) uint8 with
 * (x*255).clamp(0,255) and a truncating cast.  `out` must be 4-byte aligned. */
int gsr_pack_rgb8(const float *color, int32_t width, int32_t height, uint8_t *out, void *stream);

/* upstream markVisible / checkFrustum: present[i] = p_view.z > near_plane. */
int gsr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, float near_plane,
                     uint8_t *present, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
