// api.hip -- C-ABI entry points of libgsr_hip.so (include/gsr.h) and the forward-pass orchestration
// (upstream rasterize_points.cu RasterizeGaussiansCUDA + rasterizer_impl.cu Rasterizer::forward).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "gsr_internal.h"

namespace {
thread_local char g_err[512] = "";

// ---- optional stage timing with HIP events on the caller's stream (bench / profiling only) ----------------
constexpr int kStages = GSR_PROFILE_STAGES;
constexpr int kMaxFrames = 4096;
struct Profiler {
    int mode = 0;  // 0 off, 1 render kernel only, 2 every stage
    int frames = 0;
    std::vector<hipEvent_t> ev;  // (kStages + 1) events per frame
    hipEvent_t &at(int frame, int k) { return ev[(size_t)frame * (kStages + 1) + k]; }
};
// one recorder per calling thread: renderers driven from different threads (one per device in a multi-GPU process)
// never share events or counters
thread_local Profiler g_prof;

inline void prof_mark(int k, hipStream_t stream) {
    if (g_prof.mode == 0 || g_prof.frames >= kMaxFrames) return;
    if (g_prof.mode == 1 && k < kStages - 1) return;
    (void)hipEventRecord(g_prof.at(g_prof.frames, k), stream);
}
inline void prof_end_frame() {
    if (g_prof.mode != 0 && g_prof.frames < kMaxFrames) g_prof.frames++;
}
}  // namespace

void gsr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int gsr_check_launch(const char *what, bool debug, hipStream_t stream) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && debug) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        gsr_set_error("%s: %s", what, hipGetErrorString(e));
        return GSR_E_HIP;
    }
    return GSR_OK;
}

extern "C" {

const char *gsr_last_error(void) { return g_err; }
// (0.4: GsrFrameStats grew by `truncated` / `coop_timeouts`, gsr_debug_sort_state writes out[8], gsr_plan_query /
//  gsr_stage_step are new -- callers compiled against 0.3's header must be rebuilt)
const char *gsr_version(void) {
    static char v[96] = "";
    if (v[0] == 0) snprintf(v, sizeof(v), "gsworld_amd-gsr 0.4 (gfx950; %s)", gsr_render_build_flags());
    return v;
}
void gsr_abi_sizes(int32_t out[6]) {
    out[0] = (int32_t)sizeof(GsrSettings);
    out[1] = (int32_t)sizeof(GsrInputs);
    out[2] = (int32_t)sizeof(GsrOutputs);
    out[3] = (int32_t)sizeof(GsrBuffers);
    out[4] = (int32_t)sizeof(GsrBackwardInputs);
    out[5] = (int32_t)sizeof(GsrGrads);
}

size_t gsr_geom_bytes(int32_t P, int32_t width, int32_t height) {
    return GeomState::required(P, gsr_div_up(width, GSR_TILE) * gsr_div_up(height, GSR_TILE), gsr_div_up(width, GSR_TILE));
}
size_t gsr_binning_bytes(int64_t r_capacity) { return BinningState::required(r_capacity); }
size_t gsr_image_bytes(int32_t width, int32_t height) { return ImageState::required(width, height); }

static int validate(const GsrSettings *st, const GsrInputs *in, const GsrOutputs *out, const GsrBuffers *buf) {
    if (!st || !in || !out || !buf) {
        gsr_set_error("gsr_forward: null argument struct");
        return GSR_E_INVALID;
    }
    if (st->image_width <= 0 || st->image_height <= 0) {
        gsr_set_error("gsr_forward: image size must be positive");
        return GSR_E_INVALID;
    }
    if (gsr_div_up(st->image_width, GSR_TILE) > 65535 || gsr_div_up(st->image_height, GSR_TILE) > 65535) {
        gsr_set_error("gsr_forward: image too large for 16-bit tile rects");
        return GSR_E_INVALID;
    }
    if (in->P < 0) {
        gsr_set_error("gsr_forward: negative P");
        return GSR_E_INVALID;
    }
    // (out_color / out_invdepth may be NULL together for an inference frame that hands over out_rgb8: make_plan checks the
    //  frame really takes the compositor that honours that)
    const bool no_float = !out->out_color && !out->out_invdepth && st->forward_only && out->out_rgb8;
    if (((!out->out_color || !out->out_invdepth) && !no_float) || !in->background) {
        gsr_set_error("gsr_forward: out_color, out_invdepth and background are required (the two images may be NULL "
                      "together for a forward_only frame with out_rgb8)");
        return GSR_E_INVALID;
    }
    if (in->P > 0) {
        if (!in->means3D || !in->opacities || !in->viewmatrix || !in->projmatrix || !in->campos ||
            (!out->radii && !st->forward_only)) {
            gsr_set_error("gsr_forward: means3D, opacities, viewmatrix, projmatrix, campos and radii are required "
                          "(radii may be NULL for forward_only frames)");
            return GSR_E_INVALID;
        }
        if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) {
            gsr_set_error("Please provide excatly one of either SHs or precomputed colors!");
            return GSR_E_INVALID;
        }
        const bool have_sr = in->scales != nullptr && in->rotations != nullptr;
        if (have_sr == (in->cov3D_precomp != nullptr) || (in->scales != nullptr) != (in->rotations != nullptr)) {
            gsr_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
            return GSR_E_INVALID;
        }
        if (in->shs) {
            if (st->sh_degree < 0 || st->sh_degree > 3 || st->sh_coeffs < (st->sh_degree + 1) * (st->sh_degree + 1)) {
                gsr_set_error("gsr_forward: sh_degree must be 0..3 and sh_coeffs >= (degree+1)^2");
                return GSR_E_INVALID;
            }
        }
        if (st->binning_path < 0 || st->binning_path > 4 || st->render_variant < 0 || st->render_variant > 3 ||
            st->render_blocks_per_cu < 0 || st->render_blocks_per_cu > 8 || st->depth_sort < 0 || st->depth_sort > 1) {
            gsr_set_error("gsr_forward: binning_path must be 0..4, render_variant 0..3, render_blocks_per_cu 0..8, depth_sort 0..1");
            return GSR_E_INVALID;
        }
        if (in->param_space & 0xF0) {  // (bits 0-2: GSR_RAW_*; bit 3: GSR_FRAME_KEPT; bits 8-31: GSR_MODEL_VERSION)
            gsr_set_error("gsr_forward: unknown bits in param_space");
            return GSR_E_INVALID;
        }
        if (in->part_labels) {
            if (!in->part_lut || in->part_lut_size <= 0 || in->part_count < 0 ||
                (in->part_count > 0 && !in->part_transforms) || in->cov3D_precomp) {
                gsr_set_error("gsr_forward: part_labels needs part_lut, part_transforms and scales / rotations");
                return GSR_E_INVALID;
            }
            if (in->part_rescale && !(in->param_space & GSR_RAW_SCALES)) {
                gsr_set_error("gsr_forward: part_rescale rewrites log-scales: it needs GSR_RAW_SCALES");
                return GSR_E_INVALID;
            }
        }
        if (in->shs_rest && (!in->shs || st->sh_coeffs < 2)) {
            gsr_set_error("gsr_forward: shs_rest needs shs (= features_dc) and sh_coeffs >= 2");
            return GSR_E_INVALID;
        }
    }
    if (!buf->geom_resize || !buf->binning_resize || !buf->image_resize) {
        gsr_set_error("gsr_forward: resize callbacks are required");
        return GSR_E_INVALID;
    }
    return GSR_OK;
}

}  // extern "C"

namespace {

// How one frame runs: everything gsr_forward decides from the settings, the image size and the inputs before it touches
// the device.  Frames with equal plans (and equal P / image size) can share their launches (gsr_forward_batch).
struct Plan {
    int W, H, tiles, tiles_x, tiles_y;
    int mode;          // 1 = depth sort + counting placement (default), 2 = bin-then-sort, 0 = depth sort + radix
    bool chunk, band;  // which counting placement
    bool exact;        // binning state sized from a read-back of num_rendered
    bool infer, super, lean, order_early, lean_bin;
    bool radix_depth;  // the 3-pass LSD radix depth sort (GsrSettings.depth_sort = 1, or a model beyond the sample sort)
    GsrSettings st_bin;  // the settings with the super-tile grid as image (inference frames on the default path)
};

int make_plan(const GsrSettings *st, const GsrInputs *in, int64_t r_capacity, Plan &p, const GsrOutputs *out = nullptr) {
    p.W = st->image_width;
    p.H = st->image_height;
    p.tiles_x = gsr_div_up(p.W, GSR_TILE);
    p.tiles = p.tiles_x * gsr_div_up(p.H, GSR_TILE);
    p.tiles_y = p.tiles / p.tiles_x;
    if (r_capacity > 0xFFFFFFFFll) {
        gsr_set_error("gsr_forward: r_capacity exceeds 32-bit instance offsets");
        return GSR_E_INVALID;
    }
    // binning path: 1 = depth sort + counting placement (default), 2 = bin-then-sort, 0 = depth sort + radix
    // (tile grids above GSR_MAX_COUNT_TILES, or wider than 2048 tiles -- one band row of counters must fit 64 KiB of
    // LDS -- always take 0)
    // GsrSettings.binning_path: 0 = default (mode 1, by tile rows where the grid allows), 1 = radix (mode 0),
    // 2 = bin-then-sort (mode 2), 3 = mode 1 by chunks of 256 depth ranks (round 1's counting placement)
    const int want = (st->binning_path == 0 || st->binning_path >= 3) ? 1 : (st->binning_path == 1 ? 0 : 2);
    p.mode = GeomState::counting(p.tiles) && p.tiles_x <= 2048 ? want : 0;
    // 4 = placement by chunks of the depth order (chunkplace.hip); grids it does not take fall back to the band placement
    p.chunk = p.mode == 1 && st->binning_path == 4 && gsr_chunk_supported(p.tiles_x, p.tiles / p.tiles_x);
    p.band = p.mode == 1 && (st->binning_path == 0 || (st->binning_path == 4 && !p.chunk)) &&
             gsr_band_supported(p.tiles_x);
    p.exact = r_capacity <= 0;
    // inference frames (GsrSettings.forward_only): binned per super-tile on the default path; the binning kernels take
    // their grid from a settings copy whose image is the super-tile grid
    // infer: preprocess writes nothing a backward would read (the depth / radius words of the record carry tau / the
    // tile rect instead, which only the super-tile compositor looks at).  super: lists per super-tile -- needs the
    // compositor that has every quadrant resident and its order computed early (grids up to 1536 tiles on 256 CUs), and
    // tile coordinates that fit a byte; larger images keep per-tile lists.
    p.radix_depth = st->depth_sort == 1 || !gsr_ss_supported(in->P);  // (sample sort: every block count in one LDS)
    p.infer = st->forward_only != 0 && (p.band || p.chunk) && !p.radix_depth;
    p.super = p.infer && st->render_variant == 0 && p.tiles_x <= 255 && p.tiles_y <= 255 &&
              gsr_render_uses_quad_order(*st, p.tiles) && gsr_render_split_blocks(*st, p.tiles) == 0;
    // (the chunk placement keeps its table in an array the lean layout drops)
    p.lean = p.infer && !p.chunk;
    p.st_bin = *st;
    if (p.super) {
        p.st_bin.image_width = gsr_div_up(p.tiles_x, 1 << GSR_SUPER_SX) * GSR_TILE;
        p.st_bin.image_height = gsr_div_up(p.tiles_y, 1 << GSR_SUPER_SY) * GSR_TILE;
    }
    if (in->orig_index != nullptr && !(p.infer && p.mode == 1)) {
        gsr_set_error("gsr_forward: orig_index (a permuted model) needs a forward_only frame on the default sort / "
                      "placement path");
        return GSR_E_INVALID;
    }
    if (out && !out->out_color && !p.super) {
        gsr_set_error("gsr_forward: out_color / out_invdepth may only be NULL for an inference frame on the default path "
                      "(super-tile lists, stream compositor)");
        return GSR_E_INVALID;
    }
    // the compositor's quadrant order depends on the previous frame only: a second workgroup of the depth sort's prepare
    // launch computes it
    p.order_early = (p.band || p.chunk) && !p.radix_depth && gsr_render_uses_quad_order(*st, p.tiles);
    // (the counting placements need the list only; the fallbacks their keys / ping-pong sides too)
    p.lean_bin = p.band || p.chunk;
    return GSR_OK;
}

// B frames through one set of launches.  B > 1: the caller (gsr_forward_batch) has checked that every frame takes the
// default path (sample sort + band placement + stream compositor) in no-sync mode with the same plan, P and image size.
int run_frames(int B, const GsrSettings *st, const GsrInputs *in, const GsrOutputs *out, const GsrBuffers *buf,
               const int64_t *r_capacity, const Plan *plans, GsrFrameStats *stats, hipStream_t stream) {
    const Plan &p = plans[0];
    const bool debug = st[0].debug != 0;
    const int mode = p.mode;
    const bool chunk = p.chunk, band = p.band, exact = p.exact, infer = p.infer, super = p.super;
    // cooperative quadrants: workgroups behind the compositor's main grid for last frame's costliest quadrants (render.hip);
    // the deal that names them rides in the depth sort's prepare launch
    const int coop = super && band && mode == 1 && !p.radix_depth && p.order_early ? gsr_render_coop_blocks(st[0], p.tiles, B) : 0;
    GsrFrame fr[GSR_MAX_BATCH];
    for (int k = 0; k < B; k++) {
        GsrFrame &f = fr[k];
        f.st = &st[k];
        f.st_bin = &plans[k].st_bin;
        f.in = &in[k];
        f.out = &out[k];
        char *geom_mem = buf[k].geom_resize(buf[k].geom_user, GeomState::required(in[k].P, p.tiles, p.tiles_x, p.lean));
        char *img_mem = buf[k].image_resize(buf[k].image_user, ImageState::required(p.W, p.H));
        if (!geom_mem || !img_mem) {
            gsr_set_error("gsr_forward: resize callback returned NULL");
            return GSR_E_ALLOC;
        }
        f.g = GeomState::carve(geom_mem, in[k].P, p.tiles, nullptr, p.tiles_x, p.lean);
        f.img = ImageState::carve(img_mem, p.W, p.H);
        // exact mode first counts with an unlimited capacity, reads R back, then sizes the binning state exactly
        f.cap32 = exact ? 0xFFFFFFFFu : (uint32_t)r_capacity[k];
        // block cache (preprocess.hip prep_block_cached): inference frames of the default path whose caller vouches for the model
        f.pc = GSR_PREP_BLOCK_CACHE && infer && band && mode == 1 && !p.radix_depth && ((uint32_t)in[k].param_space >> 8) != 0u &&
               in[k].part_labels != nullptr && in[k].cull_blocks != nullptr && out[k].radii == nullptr &&
               in[k].part_count <= GSR_PC_MAX_PARTS;
        // tile reuse (render.hip): ... whose only output is the uint8 frame, in a buffer the caller says nobody else writes
        f.td = GSR_TILE_REUSE && f.pc && super && (in[k].param_space & GSR_FRAME_KEPT) != 0 && out[k].out_rgb8 != nullptr &&
               out[k].out_color == nullptr && out[k].out_invdepth == nullptr;
    }
    const GeomState &g = fr[0].g;
    const ImageState &img = fr[0].img;

    if (mode == 2) {
        // header + per-tile totals are adjacent and both accumulated by preprocess; the overflow count at the header's
        // end outlives the frame
        const size_t head = offsetof(GsrHeader, of_magic);
        const size_t clear = (size_t)((char *)g.tile_cursor - (char *)g.tile_accum);
        if (hipMemsetAsync(g.hdr, 0, head, stream) != hipSuccess ||
            hipMemsetAsync(g.tile_accum, 0, clear, stream) != hipSuccess) {
            gsr_set_error("gsr_forward: hipMemsetAsync(header) failed");
            return GSR_E_HIP;
        }
    }
    prof_mark(0, stream);
    if (int e = gsr_launch_preprocess(B, fr, mode == 2, infer, stream)) return e;
    if (int e = gsr_check_launch("preprocess", debug, stream)) return e;
    prof_mark(1, stream);
    if (mode != 2) {
        // (the frame header is reset by the first kernel that writes it: the scan of the block counts)
        if (p.radix_depth) {
            if (int e = gsr_launch_compact_and_depth_sort(in[0].P, g, debug, stream)) return e;
            if (band || chunk)
                if (int e = gsr_launch_gather_rects(in[0].P, g, debug, stream)) return e;
        } else {
            if (int e = gsr_launch_sample_depth_sort(B, fr, p.order_early, 4 * p.tiles, super ? 1 : 0, coop, debug, stream))
                return e;
        }
    }
    prof_mark(2, stream);
    if (mode == 2) {
        if (int e = gsr_launch_bin_starts(st[0], g, img, fr[0].cap32, debug, stream)) return e;
    } else if (chunk) {
        if (int e = gsr_launch_chunk_count(p.st_bin, in[0].P, g, debug, stream)) return e;
        if (int e = gsr_launch_tile_starts(1, fr, p.order_early, debug, stream)) return e;
    } else if (band) {
        // (the ranges come out of the counting launches themselves when nothing else is asked of tile_starts_kernel)
        bool starts_done = false;
        if (int e = gsr_launch_band_count(B, fr, !p.radix_depth, p.order_early, &starts_done, debug, stream)) return e;
        if (!starts_done)
            if (int e = gsr_launch_tile_starts(B, fr, p.order_early, debug, stream)) return e;
    } else if (mode == 1) {
        if (int e = gsr_launch_tile_count(st[0], in[0].P, g, img, fr[0].cap32, debug, stream)) return e;
    } else {
        if (int e = gsr_launch_tile_offsets(in[0].P, g, fr[0].cap32, debug, stream)) return e;
    }
    prof_mark(3, stream);
    // GsrOutputs.overflow_mirror: the capacity check of the counting placements (band, chunk) writes it itself; the A/B
    // paths take the 8-byte copy their callers used to make
    if (!(band || chunk))
        for (int k = 0; k < B; k++)
            if (out[k].overflow_mirror &&
                hipMemcpyAsync(out[k].overflow_mirror, &fr[k].g.hdr->of_magic, 8, hipMemcpyDefault, stream) != hipSuccess) {
                gsr_set_error("gsr_forward: overflow mirror copy failed: %s", hipGetErrorString(hipGetLastError()));
                return GSR_E_HIP;
            }
    int64_t cap0 = r_capacity[0];
    if (exact) {
        GsrHeader h;
        if (hipMemcpyAsync(&h, g.hdr, sizeof(h), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
            gsr_set_error("gsr_forward: header read-back failed: %s", hipGetErrorString(hipGetLastError()));
            return GSR_E_HIP;
        }
        cap0 = h.R_raw;
        if (stats) {
            stats->num_visible = h.V;
            stats->num_rendered = h.R_raw;
            stats->overflow = 0;
            stats->overflow_frames = h.of_magic == GSR_OF_MAGIC ? (int32_t)h.overflow_frames : 0;
        }
    }
    for (int k = 0; k < B; k++) {
        const int64_t cap = k == 0 ? cap0 : r_capacity[k];
        char *bin_mem = buf[k].binning_resize(buf[k].binning_user, BinningState::required(cap, p.lean_bin));
        if (!bin_mem) {
            gsr_set_error("gsr_forward: binning resize callback returned NULL");
            return GSR_E_ALLOC;
        }
        fr[k].b = BinningState::carve(bin_mem, cap, nullptr, p.lean_bin);
    }
    const BinningState &b = fr[0].b;
    if (mode == 2) {
        if (int e = gsr_launch_bin_scatter_and_sort(st[0], in[0].P, g, b, img, debug, stream)) return e;
    } else if (chunk) {
        if (int e = gsr_launch_chunk_place(p.st_bin, in[0].P, g, b, img, debug, stream)) return e;
    } else if (band) {
        if (int e = gsr_launch_band_place(B, fr, debug, stream)) return e;
    } else if (mode == 1) {
        if (int e = gsr_launch_tile_place(st[0], in[0].P, g, b, img, debug, stream)) return e;
    } else {
        if (int e = gsr_launch_emit_and_tile_sort(st[0], in[0].P, g, b, img, cap0, debug, stream)) return e;
    }
    prof_mark(4, stream);
    // every binning path leaves the point list in gidx[0]; modes 1 and 2 also computed the tile order
    if (int e = gsr_launch_render(B, fr, mode != 0, mode == 1, super, coop, stream)) return e;
    prof_mark(5, stream);
    prof_end_frame();
    return gsr_check_launch("render", debug, stream);
}

int zero_outputs(const GsrSettings *st, const GsrOutputs *out, hipStream_t stream) {
    // upstream launches nothing for P == 0: the outputs keep their zero fill (NOT the background)
    const size_t n = (size_t)st->image_width * st->image_height;
    if ((out->out_color && hipMemsetAsync(out->out_color, 0, 3 * n * sizeof(float), stream) != hipSuccess) ||
        (out->out_invdepth && hipMemsetAsync(out->out_invdepth, 0, n * sizeof(float), stream) != hipSuccess) ||
        (out->out_rgb8 && hipMemsetAsync(out->out_rgb8, 0, 3 * n, stream) != hipSuccess)) {
        gsr_set_error("gsr_forward: hipMemsetAsync(outputs) failed");
        return GSR_E_HIP;
    }
    return GSR_OK;
}

}  // namespace

extern "C" {

int gsr_forward(const GsrSettings *st, const GsrInputs *in, const GsrOutputs *out, const GsrBuffers *buf,
                int64_t r_capacity, GsrFrameStats *stats, void *stream_) {
    if (int e = validate(st, in, out, buf)) return e;
    hipStream_t stream = (hipStream_t)stream_;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (in->P == 0) return zero_outputs(st, out, stream);
    Plan p;
    if (int e = make_plan(st, in, r_capacity, p, out)) return e;
    return run_frames(1, st, in, out, buf, &r_capacity, &p, stats, stream);
}

// B frames (include/gsr.h).  Runs of consecutive frames that take the default path in no-sync mode with the same model
// size, image size and selectors share their launches, GSR_MAX_BATCH at most per set; every other frame runs by itself,
// exactly as gsr_forward would run it.
int gsr_forward_batch(int32_t B, const GsrSettings *st, const GsrInputs *in, const GsrOutputs *out,
                      const GsrBuffers *buf, const int64_t *r_capacity, void *stream_) {
    if (B < 0 || (B > 0 && (!st || !in || !out || !buf || !r_capacity))) {
        gsr_set_error("gsr_forward_batch: null argument array or negative B");
        return GSR_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    for (int k = 0; k < B; k++)
        if (int e = validate(&st[k], &in[k], &out[k], &buf[k])) return e;
    Plan plans[GSR_MAX_BATCH];
    auto shares = [&](int a, int b, const Plan &pa, const Plan &pb) {
        const GsrSettings &x = st[a], &y = st[b];
        return in[a].P == in[b].P && pa.W == pb.W && pa.H == pb.H && pa.mode == pb.mode && pa.band == pb.band &&
               pa.infer == pb.infer && pa.super == pb.super && pa.lean == pb.lean && pa.order_early == pb.order_early &&
               x.sh_degree == y.sh_degree && x.sh_coeffs == y.sh_coeffs && x.debug == y.debug &&
               x.render_variant == y.render_variant && x.render_blocks_per_cu == y.render_blocks_per_cu &&
               x.render_split == y.render_split && x.depth_sort == y.depth_sort &&
               (in[a].colors_precomp == nullptr) == (in[b].colors_precomp == nullptr);
    };
    int k = 0;
    while (k < B) {
        if (in[k].P == 0) {
            if (int e = zero_outputs(&st[k], &out[k], stream)) return e;
            k++;
            continue;
        }
        if (int e = make_plan(&st[k], &in[k], r_capacity[k], plans[0], &out[k])) return e;
        int n = 1;
        // (what can share launches: the default path -- sample sort, band placement, stream compositor -- with a
        //  capacity; an exact-mode frame reads its instance count back in the middle of the frame)
        const bool batchable = plans[0].mode == 1 && plans[0].band && !plans[0].radix_depth && !plans[0].exact &&
                               st[k].render_variant == 0;
        while (batchable && n < GSR_MAX_BATCH && k + n < B && in[k + n].P != 0 && r_capacity[k + n] > 0) {
            if (int e = make_plan(&st[k + n], &in[k + n], r_capacity[k + n], plans[n], &out[k + n])) return e;
            if (!shares(k, k + n, plans[0], plans[n])) break;
            n++;
        }
        if (int e = run_frames(n, st + k, in + k, out + k, buf + k, r_capacity + k, plans, nullptr, stream)) return e;
        k += n;
    }
    return GSR_OK;
}

// How gsr_forward would run a frame: make_plan's answer without touching the device (tests; the Python side asks it
// whether a frame takes a permuted model instead of mirroring the rules).
int gsr_plan_query(const GsrSettings *st, int32_t P, int32_t permuted, int64_t r_capacity, int32_t out[8]) {
    if (!st || !out || P < 0 || st->image_width <= 0 || st->image_height <= 0) {
        gsr_set_error("gsr_plan_query: null argument, negative P or empty image");
        return GSR_E_INVALID;
    }
    GsrInputs in;
    memset(&in, 0, sizeof(in));
    in.P = P;
    static const int32_t one = 0;
    in.orig_index = permuted ? &one : nullptr;  // (only ever compared with NULL)
    Plan p;
    if (int e = make_plan(st, &in, r_capacity, p)) return e;
    out[0] = p.mode;
    out[1] = p.band ? 1 : (p.chunk ? 2 : 0);
    out[2] = p.infer ? 1 : 0;
    out[3] = p.super ? 1 : 0;
    out[4] = p.lean ? 1 : 0;
    out[5] = p.radix_depth ? 1 : 0;
    out[6] = p.order_early ? 1 : 0;
    out[7] = p.exact ? 1 : 0;
    return GSR_OK;
}

int gsr_profile_enable(int mode) {
    if (mode < 0 || mode > 2) {
        gsr_set_error("gsr_profile_enable: mode must be 0, 1 or 2");
        return GSR_E_INVALID;
    }
    if (mode != 0 && g_prof.ev.empty()) {
        g_prof.ev.resize((size_t)kMaxFrames * (kStages + 1));
        for (auto &e : g_prof.ev)
            if (hipEventCreate(&e) != hipSuccess) {
                gsr_set_error("gsr_profile_enable: hipEventCreate failed");
                return GSR_E_HIP;
            }
    }
    g_prof.mode = mode;
    g_prof.frames = 0;
    return GSR_OK;
}

int gsr_profile_collect(GsrProfile *out) {
    if (!out) {
        gsr_set_error("gsr_profile_collect: null output");
        return GSR_E_INVALID;
    }
    memset(out, 0, sizeof(*out));
    out->frames = g_prof.frames;
    for (int f = 0; f < g_prof.frames; f++) {
        for (int k = (g_prof.mode == 1 ? kStages - 1 : 0); k < kStages; k++) {
            if (hipEventSynchronize(g_prof.at(f, k + 1)) != hipSuccess) {
                gsr_set_error("gsr_profile_collect: hipEventSynchronize failed");
                return GSR_E_HIP;
            }
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, g_prof.at(f, k), g_prof.at(f, k + 1)) != hipSuccess) {
                gsr_set_error("gsr_profile_collect: hipEventElapsedTime failed");
                return GSR_E_HIP;
            }
            out->stage_ms[k] += (double)ms;
        }
    }
    g_prof.frames = 0;
    return GSR_OK;
}

// Tuning aid (libraries built with -DGSR_SS_TIMING): the 64 cycle stamps the depth-sort kernels left in the geometry
// state.  Synchronises the device.
int gsr_debug_ss_stamps(int32_t P, int32_t width, int32_t height, const void *geom, uint64_t *out64) {
    if (!geom || !out64) {
        gsr_set_error("gsr_debug_ss_stamps: null argument");
        return GSR_E_INVALID;
    }
    // (height < 0: the state of an inference frame -- GsrSettings.forward_only on the default path, the lean layout)
    const bool lean = height < 0;
    if (lean) height = -height;
    const GeomState g = GeomState::carve((char *)geom, P, gsr_div_up(width, GSR_TILE) * gsr_div_up(height, GSR_TILE),
                                         nullptr, gsr_div_up(width, GSR_TILE), lean);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(out64, g.ss_dbg, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) {
        gsr_set_error("gsr_debug_ss_stamps: copy failed");
        return GSR_E_HIP;
    }
    return GSR_OK;
}

// What the depth sort of the last frame on this state did with the splitters it found there (tests, tools).
int gsr_debug_sort_state(const void *geom, int32_t out[8], void *stream_) {
    if (!geom || !out) {
        gsr_set_error("gsr_debug_sort_state: null argument");
        return GSR_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    GsrHeader h;
    if (hipMemcpyAsync(&h, geom, sizeof(h), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        gsr_set_error("gsr_debug_sort_state: read-back failed: %s", hipGetErrorString(hipGetLastError()));
        return GSR_E_HIP;
    }
    out[0] = (int32_t)h.ss_blind;
    out[1] = (int32_t)h.ss_fresh;
    out[2] = (int32_t)h.ss_bad;
    out[3] = (int32_t)h.ss_trust;
    out[4] = (int32_t)h.ss_B;
    out[5] = (int32_t)h.ss_stride;
    out[6] = (int32_t)h.coop_quads;
    // (bit 0: this frame took the kept table unchecked under a view that moved a little; bit 1: its preprocess kept blocks of
    //  the previous frame -- the block cache)
    out[7] = (int32_t)((h.ss_near & 1u) | ((h.pc_hit_last & 1u) << 1) | ((h.td_skipped & 1u) << 2));  // (bit 2: tiles skipped)
    return GSR_OK;
}

int gsr_frame_stats(const void *geom, GsrFrameStats *stats, void *stream_) {
    if (!geom || !stats) {
        gsr_set_error("gsr_frame_stats: null argument");
        return GSR_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    GsrHeader h;
    if (hipMemcpyAsync(&h, geom, sizeof(h), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        gsr_set_error("gsr_frame_stats: read-back failed: %s", hipGetErrorString(hipGetLastError()));
        return GSR_E_HIP;
    }
    stats->num_visible = h.V;
    stats->num_rendered = h.R_raw;
    stats->overflow = (int32_t)h.overflow;
    stats->overflow_frames = h.of_magic == GSR_OF_MAGIC ? (int32_t)h.overflow_frames : 0;
    stats->truncated = h.coop_timeout_now != 0u ? 1 : 0;
    stats->coop_timeouts = h.of_magic == GSR_OF_MAGIC ? (int32_t)h.coop_timeouts : 0;
    return h.overflow ? GSR_E_OVERFLOW : (stats->truncated ? GSR_E_TRUNCATED : GSR_OK);
}

int gsr_state_view(int32_t P, int32_t width, int32_t height, int64_t r_capacity, const void *geom, const void *binning,
                   const void *image, GsrStateView *v) {
    if (!v) {
        gsr_set_error("gsr_state_view: null view");
        return GSR_E_INVALID;
    }
    memset(v, 0, sizeof(*v));
    if (geom) {
        const GeomState g = GeomState::carve((char *)geom, P, gsr_div_up(width, GSR_TILE) * gsr_div_up(height, GSR_TILE));
        v->splat = reinterpret_cast<const float *>(g.splat);
        v->cov3D = g.cov3D;
        v->clamped = reinterpret_cast<const uint8_t *>(g.clamped);
        v->tiles_touched = g.tiles_touched;
        v->rects = reinterpret_cast<const uint16_t *>(g.rects);
        v->depth_order = g.order;  // (not written by the bin-then-sort path, GsrSettings.binning_path = 2)
    }
    if (binning) {
        const BinningState b = BinningState::carve((char *)binning, r_capacity);
        v->point_list = b.gidx[0];  // both binning paths finish in side 0
        v->point_tiles = nullptr;   // tile ids follow from `ranges` (the counting path never materialises them)
    }
    if (image) {
        const ImageState s = ImageState::carve((char *)image, width, height);
        v->ranges = reinterpret_cast<const uint32_t *>(s.ranges);
        v->final_T = s.final_T;
        v->n_contrib = s.n_contrib;
    }
    return GSR_OK;
}

}  // extern "C"
