// preprocess.hip -- per-Gaussian projection stage (upstream forward.cu preprocessCUDA + auxiliary.h helpers;
// SURVEY.md 8a rows A3/A4/A9).  One thread per Gaussian, 256-thread workgroups.
//
// Arithmetic contract: compiled with -ffp-contract=off; every fused multiply-add below is an explicit
// __builtin_fmaf, division and sqrtf are the correctly rounded forms (hipcc default), so the depth keys, radii
// and tile rects are bit-identical to the canonical order fixed by oracle/gs_oracle.c.
//
// HBM traffic per Gaussian: 12 B xyz always; +28 B scale/quat, +4 B opacity once the near cull passed;
// +192 B of SH and 88 B of state written only for Gaussians that survive to a non-empty tile rect.
#include <string.h>

#include "gsr_internal.h"

#ifndef GSR_TIGHT_RECT
#define GSR_TIGHT_RECT 1
#endif

namespace {

struct PreprocessArgs {
    int P, D, M;
    int W, H, gx, gy;
    float tanfovx, tanfovy, fx, fy, scale_modifier, near_plane;
    int antialiasing;
    int param_space;  // GSR_RAW_* flags: activations evaluated here instead of three torch passes per frame
    int infer;        // GsrSettings.forward_only: nothing a backward would read is written; rec2.w = packed tile rect
    const float *means3D, *shs, *shs_rest, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *view, *proj, *campos;
    // optional rigid transform of labelled Gaussians (GsrInputs.part_*): same arithmetic as transform.hip
    const float *part_labels, *part_transforms;
    const int32_t *part_lut;
    const uint8_t *part_rescale;
    int part_lut_size, part_count;
    // optional block bounds (GsrInputs.cull_blocks) and the original numbering of a permuted model (GsrInputs.orig_index):
    // the state stays in the numbering of the ARRAYS (a block's records are neighbours in memory, and so are the records a
    // tile's list gathers); only `radii` -- the caller's array -- is written by original number, and the depth sort
    // breaks ties by it (depthsort.hip)
    const float *cull_blocks;
    const int32_t *orig_index;
    int32_t *radii;
    float4 *splat;
    float *cov3D;
    uint32_t *clamped;
    uint32_t *tiles_touched;
    uint2 *rects;
    uint32_t *block_counts;
    uint2 *block_recs;     // [P] (index, depth bits) of the block's visible Gaussians, compacted to the head of the block's
                           // own 256 slots (depthsort.hip gathers them: 8 B per VISIBLE Gaussian instead of a 4-byte key
                           // written and re-read for all N)

    // bin-then-sort path: per-tile instance totals and the visible count are accumulated here
    int num_tiles;
    uint32_t *tile_accum;
    GsrHeader *hdr;
    // block cache (prep_block_cached): the two slots that hold the last two frames' camera + pose table, or nullptr = this
    // frame recomputes every block; pc_sig: everything else the records depend on (settings, model identity, state layout)
    float *pc_slots;
    uint32_t pc_sig;
    // tile reuse (render.hip): the tiles a recomputed Gaussian touches, now or in the previous frame, are marked here with the
    // frame's token; nullptr = the frame composites every tile.  bg: the background colour is part of what a pixel depends on
    uint32_t *tile_dirty;
    const float *bg;
    uint32_t td_sig;
};

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
constexpr float kC2_0 = 1.0925484305920792f, kC2_1 = -1.0925484305920792f, kC2_2 = 0.31539156525252005f,
                kC2_3 = -1.0925484305920792f, kC2_4 = 0.5462742152960396f;
constexpr float kC3_0 = -0.5900435899266435f, kC3_1 = 2.890611442640554f, kC3_2 = -0.4570457994644658f,
                kC3_3 = 0.3731763325901154f, kC3_4 = -0.4570457994644658f, kC3_5 = 1.445305721320277f,
                kC3_6 = -0.5900435899266435f;

// real SH basis of a unit direction (coefficient signs of forward.cu computeColorFromSH)
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float *b) {
    b[0] = kC0;
    if (deg > 0) {
        b[1] = -(kC1 * y);
        b[2] = kC1 * z;
        b[3] = -(kC1 * x);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = kC2_0 * xy;
            b[5] = kC2_1 * yz;
            b[6] = kC2_2 * (fma_(2.f, zz, -xx) - yy);
            b[7] = kC2_3 * xz;
            b[8] = kC2_4 * (xx - yy);
            if (deg > 2) {
                b[9] = (kC3_0 * y) * fma_(3.f, xx, -yy);
                b[10] = (kC3_1 * xy) * z;
                b[11] = (kC3_2 * y) * (fma_(4.f, zz, -xx) - yy);
                b[12] = (kC3_3 * z) * fma_(-3.f, yy, fma_(-3.f, xx, 2.f * zz));
                b[13] = (kC3_4 * x) * (fma_(4.f, zz, -xx) - yy);
                b[14] = (kC3_5 * z) * (xx - yy);
                b[15] = (kC3_6 * x) * fma_(-3.f, yy, xx);
            }
        }
    }
}

// Position of Gaussian i after the optional rigid part transform (GsrInputs.part_*); xf = its pose row or nullptr.
__device__ __forceinline__ void prep_position(const PreprocessArgs &a, int i, float &px_, float &py_, float &pz_,
                                              const float *&xf_, int &part_) {
    float px = a.means3D[3 * (size_t)i], py = a.means3D[3 * (size_t)i + 1], pz = a.means3D[3 * (size_t)i + 2];
    // moving part?  (label -> part through the LUT; the reference compares labels after .long(): truncation)
    const float *xf = nullptr;
    int part = -1;
    if (a.part_labels != nullptr) {
        const int label = (int)a.part_labels[i];
        part = (label >= 0 && label < a.part_lut_size) ? a.part_lut[label] : -1;
        if (part >= 0 && part < a.part_count) {
            xf = a.part_transforms + (size_t)part * 17;
            // xyz' = R (s xyz) + t, in the operation order of transform.hip (plain multiplies and adds)
            const float s = xf[12];
            px *= s; py *= s; pz *= s;
            const float rx = xf[0] * px + xf[1] * py + xf[2] * pz + xf[9];
            const float ry = xf[3] * px + xf[4] * py + xf[5] * pz + xf[10];
            const float rz = xf[6] * px + xf[7] * py + xf[8] * pz + xf[11];
            px = rx; py = ry; pz = rz;
        }
    }
    px_ = px; py_ = py; pz_ = pz; xf_ = xf; part_ = part;
}

struct GeomOut {
    bool visible;
    float4 pos;      // xyz (after the part transform) + radius as float, handed to the colour phase
    uint32_t tiles;  // tiles touched
    uint32_t key;    // depth bits of a visible Gaussian, 0 otherwise
    uint2 rect;
    int radius;
};

// What a Gaussian is in the WORLD under one pose table and scale modifier -- everything of the per-Gaussian pass that no
// camera enters: position after the part transform, 3D covariance, raw opacity.  The frames of a step that share the
// model (gsr_forward_batch: cameras x environments) compute it once per Gaussian and pose table (round 6; until then
// every frame's workgroup read and transformed the model for itself).  The covariance half is filled in when the first
// frame needs it (a Gaussian behind every camera's near plane never loads its scales / rotation / opacity).
struct WorldGauss {
    float px, py, pz;
    int part;           // moving part of the Gaussian (-1: none): what the pose table can change
    bool have_cov;
    float c0, c1, c2, c3, c4, c5, opacity_raw;
};

__device__ __forceinline__ void prep_world_cov(const PreprocessArgs &a, int i, WorldGauss &w) {
    {
        // ---- 3D covariance: Sigma = R diag((mod*s)^2) R^T ------------------------------------------------
        // (requested with the covariance inputs: one round trip, not two.  Requesting all of them together with the
        //  POSITION in blocks that passed the frustum test -- nearly every Gaussian of such a block is in front of the
        //  camera -- was measured in round 4: no gain, 6 468 against 6 481 frames/s one at a time)
        const int part = w.part;
        const float *xf = (part >= 0 && part < a.part_count && a.part_labels != nullptr) ? a.part_transforms + (size_t)part * 17 : nullptr;
        const float opacity_raw = a.opacities[i];
        float c0, c1, c2, c3, c4, c5;
        if (a.cov3D_precomp) {
            const float *c = a.cov3D_precomp + 6 * (size_t)i;
            c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; c5 = c[5];
        } else {
            float4 rq = *reinterpret_cast<const float4 *>(a.rotations + 4 * (size_t)i);
            float sc0 = a.scales[3 * (size_t)i], sc1 = a.scales[3 * (size_t)i + 1], sc2 = a.scales[3 * (size_t)i + 2];
            if (xf != nullptr) {
                // rot' = standardize(q_R (x) rot / |rot|) * |rot|  (gs_utils.py:242-249; transform.hip)
                const float norm = sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
                const float bw = rq.x / norm, bx = rq.y / norm, by = rq.z / norm, bz = rq.w / norm;
                const float aw = xf[13], ax = xf[14], ay = xf[15], az = xf[16];
                float ow = aw * bw - ax * bx - ay * by - az * bz;
                float ox = aw * bx + ax * bw + ay * bz - az * by;
                float oy = aw * by - ax * bz + ay * bw + az * bx;
                float oz = aw * bz + ax * by - ay * bx + az * bw;
                if (ow < 0.f) { ow = -ow; ox = -ox; oy = -oy; oz = -oz; }
                rq = make_float4(ow * norm, ox * norm, oy * norm, oz * norm);
                if (a.part_rescale != nullptr && a.part_rescale[part]) {
                    // the reference's rewrite of a tracked actor's log-scales: inverse_sigmoid(exp(s) * scale)
                    const float s = xf[12];
                    const float x0 = expf(sc0) * s, x1 = expf(sc1) * s, x2 = expf(sc2) * s;
                    sc0 = logf(x0 / (1.0f - x0));
                    sc1 = logf(x1 / (1.0f - x1));
                    sc2 = logf(x2 / (1.0f - x2));
                }
            }
            if (a.param_space & GSR_RAW_ROTATIONS) {  // F.normalize: q / max(|q|, 1e-12)
                const float n2 = fma_(rq.w, rq.w, fma_(rq.z, rq.z, fma_(rq.y, rq.y, rq.x * rq.x)));
                const float d = fmaxf(sqrtf(n2), 1e-12f);
                rq = make_float4(rq.x / d, rq.y / d, rq.z / d, rq.w / d);
            }
            if (a.param_space & GSR_RAW_SCALES) {
                sc0 = exp_canonical(sc0);
                sc1 = exp_canonical(sc1);
                sc2 = exp_canonical(sc2);
            }
            const float r = rq.x, x = rq.y, y = rq.z, z = rq.w;
            const float s0 = a.scale_modifier * sc0;
            const float s1 = a.scale_modifier * sc1;
            const float s2 = a.scale_modifier * sc2;
            const float R00 = fma_(-2.f, fma_(z, z, y * y), 1.f);
            const float R01 = 2.f * fma_(-r, z, x * y);
            const float R02 = 2.f * fma_(r, y, x * z);
            const float R10 = 2.f * fma_(r, z, x * y);
            const float R11 = fma_(-2.f, fma_(z, z, x * x), 1.f);
            const float R12 = 2.f * fma_(-r, x, y * z);
            const float R20 = 2.f * fma_(-r, y, x * z);
            const float R21 = 2.f * fma_(r, x, y * z);
            const float R22 = fma_(-2.f, fma_(y, y, x * x), 1.f);
            // M[k][j] = s_k * R[j][k]
            const float M00 = s0 * R00, M01 = s0 * R10, M02 = s0 * R20;
            const float M10 = s1 * R01, M11 = s1 * R11, M12 = s1 * R21;
            const float M20 = s2 * R02, M21 = s2 * R12, M22 = s2 * R22;
            c0 = fma_(M20, M20, fma_(M10, M10, M00 * M00));
            c1 = fma_(M20, M21, fma_(M10, M11, M00 * M01));
            c2 = fma_(M20, M22, fma_(M10, M12, M00 * M02));
            c3 = fma_(M21, M21, fma_(M11, M11, M01 * M01));
            c4 = fma_(M21, M22, fma_(M11, M12, M01 * M02));
            c5 = fma_(M22, M22, fma_(M12, M12, M02 * M02));
        }
        w.c0 = c0; w.c1 = c1; w.c2 = c2; w.c3 = c3; w.c4 = c4; w.c5 = c5;
        w.opacity_raw = opacity_raw;
        w.have_cov = true;
    }
}

// The exact per-Gaussian geometry of ONE frame: cull, projection, covariances, conic, radius,
// rect.  Writes rec0 / rec1 / cov3D / rects of a visible Gaussian; the caller writes radii, tiles_touched and the
// block-local sort record.
__device__ __forceinline__ GeomOut prep_geometry(const PreprocessArgs &a, int i, WorldGauss &w) {
    bool visible = false;
    float4 mypos = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 my_rect = make_uint2(0u, 0u);
    const float px = w.px, py = w.py, pz = w.pz;
    const float *m = a.view;
    // transformPoint4x3: M[r][c] = m[c*4+r]
    const float vx = fma_(m[8], pz, fma_(m[4], py, m[0] * px)) + m[12];
    const float vy = fma_(m[9], pz, fma_(m[5], py, m[1] * px)) + m[13];
    const float vz = fma_(m[10], pz, fma_(m[6], py, m[2] * px)) + m[14];
    int radius = 0;
    uint32_t touched = 0;
    if (vz > a.near_plane) {  // in_frustum with GSWorld's near plane
        const float *q = a.proj;
        const float hx = fma_(q[8], pz, fma_(q[4], py, q[0] * px)) + q[12];
        const float hy = fma_(q[9], pz, fma_(q[5], py, q[1] * px)) + q[13];
        const float hw = fma_(q[11], pz, fma_(q[7], py, q[3] * px)) + q[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float ndc_x = hx * p_w, ndc_y = hy * p_w;
        if (!w.have_cov) prep_world_cov(a, i, w);
        const float c0 = w.c0, c1 = w.c1, c2 = w.c2, c3 = w.c3, c4 = w.c4, c5 = w.c5, opacity_raw = w.opacity_raw;

        // ---- EWA 2D covariance: (J W) Sigma (J W)^T ---------------------------------------------------------
        const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
        const float txtz = vx / vz, tytz = vy / vz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
        const float J00 = a.fx / vz, J02 = -(a.fx * tx) / (vz * vz);
        const float J11 = a.fy / vz, J12 = -(a.fy * ty) / (vz * vz);
        // A = J W with W[i][j] = m[j*4+i]
        const float A00 = fma_(J02, m[2], J00 * m[0]);
        const float A01 = fma_(J02, m[6], J00 * m[4]);
        const float A02 = fma_(J02, m[10], J00 * m[8]);
        const float A10 = fma_(J12, m[2], J11 * m[1]);
        const float A11 = fma_(J12, m[6], J11 * m[5]);
        const float A12 = fma_(J12, m[10], J11 * m[9]);
        // B = A Sigma
        const float B00 = fma_(A02, c2, fma_(A01, c1, A00 * c0));
        const float B01 = fma_(A02, c4, fma_(A01, c3, A00 * c1));
        const float B02 = fma_(A02, c5, fma_(A01, c4, A00 * c2));
        const float B10 = fma_(A12, c2, fma_(A11, c1, A10 * c0));
        const float B11 = fma_(A12, c4, fma_(A11, c3, A10 * c1));
        const float B12 = fma_(A12, c5, fma_(A11, c4, A10 * c2));
        float cxx = fma_(B02, A02, fma_(B01, A01, B00 * A00));
        const float cxy = fma_(B02, A12, fma_(B01, A11, B00 * A10));
        float cyy = fma_(B12, A12, fma_(B11, A11, B10 * A10));

        const float det_cov = fma_(-cxy, cxy, cxx * cyy);
        cxx += 0.3f;
        cyy += 0.3f;
        const float det = fma_(-cxy, cxy, cxx * cyy);
        float h_scale = 1.0f;
        if (a.antialiasing) h_scale = sqrtf(fmaxf(0.000025f, det_cov / det));
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float conic_x = cyy * det_inv, conic_y = -cxy * det_inv, conic_z = cxx * det_inv;
            const float mid = 0.5f * (cxx + cyy);
            const float root = sqrtf(fmaxf(0.1f, fma_(mid, mid, -det)));
            const float lambda1 = mid + root, lambda2 = mid - root;
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            // ndc2Pix is binary64 upstream (double literals)
            const float pix_x = (float)((((double)ndc_x + 1.0) * (double)a.W - 1.0) * 0.5);
            const float pix_y = (float)((((double)ndc_y + 1.0) * (double)a.H - 1.0) * 0.5);
            const int ir = (int)my_radius;
            const float fr = (float)ir;
            int rminx = (int)((pix_x - fr) / (float)GSR_TILE);
            int rminy = (int)((pix_y - fr) / (float)GSR_TILE);
            int rmaxx = (int)((pix_x + fr + (float)(GSR_TILE - 1)) / (float)GSR_TILE);
            int rmaxy = (int)((pix_y + fr + (float)(GSR_TILE - 1)) / (float)GSR_TILE);
            rminx = min(a.gx, max(0, rminx));
            rminy = min(a.gy, max(0, rminy));
            rmaxx = min(a.gx, max(0, rmaxx));
            rmaxy = min(a.gy, max(0, rmaxy));
            int area = (rmaxx - rminx) * (rmaxy - rminy);
            if (area != 0) radius = ir;  // (what the caller sees as radii: the reference's, whatever the lists keep)
            float opacity = 0.f, third = vz;
            if (area != 0) {
                float opacity_in = opacity_raw;
                if (a.param_space & GSR_RAW_OPACITY) opacity_in = sigmoid_canonical(opacity_in);
                opacity = opacity_in * h_scale;
                if (a.infer) {
                    // inference frames: the compositor never reads the depth word of the record; it carries what its
                    // per-quadrant cull would otherwise recompute for every candidate of every quadrant (render.hip
                    // quadrant_may_hit): tau = -ln(255 opacity), lowered by ~2 ulp (the cull may only err towards
                    // keeping), with the "conic is comfortably positive definite" flag in its last mantissa bit
                    const float tau = -0.6931471805599453f * __builtin_amdgcn_logf(255.0f * opacity);
                    const float low = tau - fma_(2e-7f, fabsf(tau), 1e-30f);
                    const float ac = conic_x * conic_z;
                    const bool safe = conic_x > 0.0f && conic_z > 0.0f && ac > 1e-20f && conic_y * conic_y <= 0.999f * ac;
                    third = __uint_as_float((__float_as_uint(low) & ~1u) | (safe ? 1u : 0u));
#if GSR_TIGHT_RECT
                    // ... and the tile rect shrinks to the tiles the splat can COLOUR.  The reference bins by the
                    // square of ceil(3 sqrt(lambda_max)) around the centre; a pixel is composited only where
                    // alpha >= 1/255, i.e. inside the ellipse  A dx^2 + 2 B dx dy + C dy^2 <= -2 tau, whose bounding box
                    // has the half-widths sqrt(-2 tau C / det), sqrt(-2 tau A / det).  For the flat, tilted splats of a
                    // scanned table that box is a fraction of the square, and every instance outside it is one the
                    // compositor would fetch, test and skip.  Only for a comfortably positive definite conic (the
                    // computed power is then <= 0 everywhere: stream_conic_is_safe), with 1e-4 relative + a quarter
                    // pixel to spare (the compositor's own float error on the power is ~4e-6 of its terms); never
                    // beyond the reference's rect, which cuts opaque splats at 3 sigma.  Inference frames only: no
                    // backward reads these lists, the image is the same bit for bit (tests, tools/fuzz_forward_only.py).
                    if (safe) {
                        const float k = -2.0f * low;  // (<= 0: opacity below 1/255, alpha never passes)
                        const float detc = fma_(-conic_y, conic_y, ac);
                        // (hardware reciprocal / square root, ~1 ulp each: the margins are a hundred times that)
                        const float rdet = __builtin_amdgcn_rcpf(detc);
                        const float kd = k * rdet;
                        // The margin follows the conic's conditioning.  The compositor's power is a sum of terms that
                        // cancel when B^2 approaches A C: its float error is ~4e-6 of sum |terms| (the slack of
                        // quadrant_may_hit), and on the ellipse's boundary sum |terms| <= ~2 k A C / det.  A pixel just
                        // outside the exact ellipse can therefore still compute alpha >= 1/255 when the exact power is
                        // within 8e-6 k A C / det of tau: the ellipse grows by the factor sqrt(1 + 1.6e-5 A C / det) at
                        // most.  2e-5 A C / det (2.5 x that, up to 2 % at the `safe` limit B^2 = 0.999 A C) + 1e-4
                        // relative + a quarter pixel.  (Round 3 used 1e-4 + 0.25 px whatever the conditioning: short by
                        // up to 0.4 % of the half-width for needle-shaped splats near the limit -- ADVICE round 3.)
                        const float rel = fma_(2e-5f, ac * rdet, 1.0001f);
                        const float hx = fma_(__builtin_amdgcn_sqrtf(kd * conic_z), rel, 0.25f);
                        const float hy = fma_(__builtin_amdgcn_sqrtf(kd * conic_x), rel, 0.25f);
                        if (!(k > 0.0f)) {
                            area = 0;
                        } else if (hx < 1e6f && hy < 1e6f) {  // (false for NaN / inf: the reference's rect stays)
                            rminx = max(rminx, (int)floorf((pix_x - hx) * (1.0f / GSR_TILE)));
                            rminy = max(rminy, (int)floorf((pix_y - hy) * (1.0f / GSR_TILE)));
                            rmaxx = min(rmaxx, (int)floorf((pix_x + hx) * (1.0f / GSR_TILE)) + 1);
                            rmaxy = min(rmaxy, (int)floorf((pix_y + hy) * (1.0f / GSR_TILE)) + 1);
                            area = max(rmaxx - rminx, 0) * max(rmaxy - rminy, 0);
                        }
                    }
#endif
                }
            }
            if (area != 0) {
                float4 *rec = a.splat + 3 * (size_t)i;
                rec[0] = make_float4(pix_x, pix_y, third, 1.0f / vz);
                rec[1] = make_float4(conic_x, conic_y, conic_z, opacity);
                if (!a.infer) {  // (the 3D covariance is kept for the backward only)
                    float2 *cv = reinterpret_cast<float2 *>(a.cov3D + 6 * (size_t)i);
                    cv[0] = make_float2(c0, c1);
                    cv[1] = make_float2(c2, c3);
                    cv[2] = make_float2(c4, c5);
                }
                my_rect = make_uint2((uint32_t)rminx | ((uint32_t)rminy << 16),
                                     (uint32_t)rmaxx | ((uint32_t)rmaxy << 16));
                a.rects[i] = my_rect;
                touched = (uint32_t)area;
                visible = true;
                // fourth word of the colour record: the radius (unused downstream), or -- inference frames -- the tile
                // rect as four bytes, which the compositor tests its tile against (super-tile binning, render.hip)
                mypos = make_float4(px, py, pz,
                                    a.infer ? __uint_as_float((uint32_t)rminx | ((uint32_t)rminy << 8) |
                                                              ((uint32_t)rmaxx << 16) | ((uint32_t)rmaxy << 24))
                                            : fr);
            }
        }
    }
    GeomOut o;
    o.visible = visible;
    o.pos = mypos;
    o.tiles = touched;
    o.key = visible ? __float_as_uint(vz) : 0u;  // (vz > near_plane > 0: never the 0 pattern)
    o.rect = my_rect;
    o.radius = radius;
    return o;
}

// SH -> RGB (or the precomputed colour) of visible Gaussian g at position pp.xyz; writes rec2 and the clamp flags.
#ifndef GSR_PREP_GROUP_HALVES
#define GSR_PREP_GROUP_HALVES 1  // (A/B: the multi-frame kernel's SH coefficients in two batches; see prep_colour)
#endif
// (HALVES: the coefficients are requested in two batches instead of one -- half the registers in flight, one more round
//  trip.  The multi-frame kernel takes it: its loop over the frames costs the compiler registers, and the second and
//  later frames of a group find the coefficients in this CU's cache.  Same products, same order: same colours.)
template <bool FAST_SH16, bool HALVES = false>
__device__ __forceinline__ void prep_colour(const PreprocessArgs &a, const int g, const float4 pp) {
    float cr, cg, cb;
    uint32_t clamp_bits = 0;
    if (a.colors_precomp) {
        cr = a.colors_precomp[3 * (size_t)g];
        cg = a.colors_precomp[3 * (size_t)g + 1];
        cb = a.colors_precomp[3 * (size_t)g + 2];
    } else {
        float dx = pp.x - a.campos[0], dy = pp.y - a.campos[1], dz = pp.z - a.campos[2];
        const float len = sqrtf(fma_(dz, dz, fma_(dy, dy, dx * dx)));
        dx = dx / len; dy = dy / len; dz = dz / len;
        float b[16];
        sh_basis(a.D, dx, dy, dz, b);
        if (a.shs_rest) {
            // split storage (features_dc | features_rest): coefficient 0 from one array, 1.. from the other
            const float *dc = a.shs + 3 * (size_t)g;
            const float *rest = a.shs_rest + (size_t)g * (a.M - 1) * 3;
            const int nb = (a.D + 1) * (a.D + 1);
            cr = b[0] * dc[0]; cg = b[0] * dc[1]; cb = b[0] * dc[2];
            if (a.D == 3 && HALVES) {
                float f[24];
#pragma unroll
                for (int k = 0; k < 24; k++) f[k] = rest[k];
#pragma unroll
                for (int k = 1; k < 9; k++) {
                    cr = fma_(b[k], f[3 * k - 3], cr);
                    cg = fma_(b[k], f[3 * k - 2], cg);
                    cb = fma_(b[k], f[3 * k - 1], cb);
                }
                asm volatile("" : "+v"(cr), "+v"(cg), "+v"(cb));  // (the second batch is requested behind the first's use)
#pragma unroll
                for (int k = 0; k < 21; k++) f[k] = rest[24 + k];
#pragma unroll
                for (int k = 9; k < 16; k++) {
                    cr = fma_(b[k], f[3 * k - 27], cr);
                    cg = fma_(b[k], f[3 * k - 26], cg);
                    cb = fma_(b[k], f[3 * k - 25], cb);
                }
            } else if (a.D == 3) {
                float f[45];
#pragma unroll
                for (int k = 0; k < 45; k++) f[k] = rest[k];  // 15 x dwordx3, all in flight together
#pragma unroll
                for (int k = 1; k < 16; k++) {
                    cr = fma_(b[k], f[3 * k - 3], cr);
                    cg = fma_(b[k], f[3 * k - 2], cg);
                    cb = fma_(b[k], f[3 * k - 1], cb);
                }
            } else {
                for (int k = 1; k < nb; k++) {
                    cr = fma_(b[k], rest[3 * k - 3], cr);
                    cg = fma_(b[k], rest[3 * k - 2], cg);
                    cb = fma_(b[k], rest[3 * k - 1], cb);
                }
            }
        } else if (FAST_SH16) {
            // D == 3, M == 16: 48 contiguous floats, 16-byte aligned -> 12 x dwordx4
            const float4 *sh4 = reinterpret_cast<const float4 *>(a.shs + (size_t)g * 48);
            if (HALVES) {
                float4 v[6];
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = sh4[k];
                const float *f = reinterpret_cast<const float *>(v);
                cr = b[0] * f[0]; cg = b[0] * f[1]; cb = b[0] * f[2];
#pragma unroll
                for (int k = 1; k < 8; k++) {
                    cr = fma_(b[k], f[3 * k], cr);
                    cg = fma_(b[k], f[3 * k + 1], cg);
                    cb = fma_(b[k], f[3 * k + 2], cb);
                }
                asm volatile("" : "+v"(cr), "+v"(cg), "+v"(cb));  // (the second batch is requested behind the first's use)
                float4 u[6];
#pragma unroll
                for (int k = 0; k < 6; k++) u[k] = sh4[6 + k];
                const float *h = reinterpret_cast<const float *>(u);
#pragma unroll
                for (int k = 8; k < 16; k++) {
                    cr = fma_(b[k], h[3 * k - 24], cr);
                    cg = fma_(b[k], h[3 * k - 23], cg);
                    cb = fma_(b[k], h[3 * k - 22], cb);
                }
            } else {
            float4 v[12];
#pragma unroll
            for (int k = 0; k < 12; k++) v[k] = sh4[k];
            const float *f = reinterpret_cast<const float *>(v);
            cr = b[0] * f[0]; cg = b[0] * f[1]; cb = b[0] * f[2];
#pragma unroll
            for (int k = 1; k < 16; k++) {
                cr = fma_(b[k], f[3 * k], cr);
                cg = fma_(b[k], f[3 * k + 1], cg);
                cb = fma_(b[k], f[3 * k + 2], cb);
            }
            }
        } else {
            const float *sh = a.shs + (size_t)g * a.M * 3;
            const int nb = (a.D + 1) * (a.D + 1);
            cr = b[0] * sh[0]; cg = b[0] * sh[1]; cb = b[0] * sh[2];
            for (int k = 1; k < nb; k++) {
                cr = fma_(b[k], sh[3 * k], cr);
                cg = fma_(b[k], sh[3 * k + 1], cg);
                cb = fma_(b[k], sh[3 * k + 2], cb);
            }
        }
        cr += 0.5f; cg += 0.5f; cb += 0.5f;
        clamp_bits = (cr < 0.f ? 1u : 0u) | (cg < 0.f ? 0x100u : 0u) | (cb < 0.f ? 0x10000u : 0u);
        cr = fmaxf(cr, 0.f); cg = fmaxf(cg, 0.f); cb = fmaxf(cb, 0.f);
    }
    a.splat[3 * (size_t)g + 2] = make_float4(cr, cg, cb, pp.w);
    if (!a.infer) a.clamped[g] = clamp_bits;
}

// ---------------------------------------------------------------------------------------------------------
// View-frustum test of one block of GSR_BLOCK consecutive Gaussians (GsrInputs.cull_blocks).  true = NO Gaussian of
// the block can have a non-empty tile rect in this frame (or pass the near test), so the reference writes radii = 0 for
// every one of them and the block's workgroup has nothing to do.  Has to err towards false only.
//
// The block's record: box of the centres (lo, hi), rho >= sqrt(lambda_max(Sigma)) of every member (scale modifier 1),
// the common part label.  Lane c < 8 of the calling wave takes corner c through the part pose (affine), the view
// matrix (affine) and the projection; the members' centres lie in the convex hull of the corners all the way:
//   * view depth in [min, max] of the corners' -> all behind the near plane: done;
//   * with every corner in front (w > 0) the pixel coordinates are linear-fractional with a positive denominator on the
//     hull, so they lie between the corners' minima and maxima;
//   * radius: my_radius = ceil(3 sqrt(lambda_1)), lambda_1 = mid + sqrt(max(0.1, mid^2 - det)) <= (largest eigenvalue of
//     the dilated 2D covariance) + sqrt(0.1) = lambda_max(A Sigma A^T) + 0.3 + 0.3163 with A = J W, and
//     lambda_max(A Sigma A^T) <= |J|_2^2 |W|_2^2 lambda_max(Sigma); |W|_2^2 <= 1 + |W^T W - I|_F;
//     J J^T = [[a^2 + b^2, b d], [b d, c^2 + d^2]] with a = fx / vz, |b| <= fx limx / vz, c = fy / vz, |d| <= fy limy / vz
//     (|tx / vz| is clamped to limx = 1.3 tan(fov / 2)), so |J|_2^2 <= (max(fx^2 (1 + limx^2), fy^2 (1 + limy^2)) +
//     fx fy limx limy) / vz^2 (Gershgorin);
//   * the rect of a Gaussian is empty when pix + r < 1 or pix - r >= 16 x tiles on an axis (getRect's clamping).
// Margins: 1e-3 relative on the squared radius, 2 px on the radius, 1 px on the box -- the float error of the corner
// chain is ~1e-6 of its terms.  A corner that is not finite (NaN / infinite box or pose): not culled.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool prep_block_culled(const PreprocessArgs &a, int block) {
    const float4 b0 = *reinterpret_cast<const float4 *>(a.cull_blocks + 8 * (size_t)block);
    const float4 b1 = *reinterpret_cast<const float4 *>(a.cull_blocks + 8 * (size_t)block + 4);
    const int lane = gsr_lane();
    float px = (lane & 1) ? b0.w : b0.x, py = (lane & 2) ? b1.x : b0.y, pz = (lane & 4) ? b1.y : b0.z;
    float rho = b1.z;
    if (a.part_labels != nullptr) {
        const float lf = b1.w;
        if (!(lf == lf)) return false;  // members carry different labels: no common pose
        const int label = (int)lf;
        const int part = (label >= 0 && label < a.part_lut_size) ? a.part_lut[label] : -1;
        if (part >= 0 && part < a.part_count) {
            if (a.part_rescale != nullptr && a.part_rescale[part]) return false;  // (rewritten scales: rho does not hold)
            const float *xf = a.part_transforms + (size_t)part * 17;
            const float s = xf[12];
            px *= s; py *= s; pz *= s;
            const float rx = xf[0] * px + xf[1] * py + xf[2] * pz + xf[9];
            const float ry = xf[3] * px + xf[4] * py + xf[5] * pz + xf[10];
            const float rz = xf[6] * px + xf[7] * py + xf[8] * pz + xf[11];
            px = rx; py = ry; pz = rz;
            // R(q) = (1 - n) I + n R(q / |q|), n = |q|^2: the pose quaternion multiplies every member's, so the rotation's
            // norm grows by at most max(1, 2 n - 1)
            const float n = xf[13] * xf[13] + xf[14] * xf[14] + xf[15] * xf[15] + xf[16] * xf[16];
            rho *= fmaxf(1.0f, 2.0f * n - 1.0f) * 1.0001f;
        }
    }
    const float *m = a.view, *q = a.proj;
    const float vz = fma_(m[10], pz, fma_(m[6], py, m[2] * px)) + m[14];
    const float hx = fma_(q[8], pz, fma_(q[4], py, q[0] * px)) + q[12];
    const float hy = fma_(q[9], pz, fma_(q[5], py, q[1] * px)) + q[13];
    const float hw = fma_(q[11], pz, fma_(q[7], py, q[3] * px)) + q[15] + 0.0000001f;
    const float inv = 1.0f / hw;
    const float cx = fma_(hx * inv + 1.0f, (float)a.W, -1.0f) * 0.5f;
    const float cy = fma_(hy * inv + 1.0f, (float)a.H, -1.0f) * 0.5f;
    // (fminf / fmaxf below drop a NaN operand: a corner that is not finite -- a NaN or infinite box, a pose with one -- has
    // to be seen before the reduction; every group of eight lanes holds the same eight corners)
    const bool finite = fabsf(vz) < 1e30f && fabsf(hw) < 1e30f && fabsf(cx) < 1e30f && fabsf(cy) < 1e30f && rho == rho;
    if (__builtin_amdgcn_ballot_w64(!finite) != 0ull) return false;
    float zlo = vz, zhi = vz, wlo = hw, xlo = cx, xhi = cx, ylo = cy, yhi = cy;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        zlo = fminf(zlo, __shfl_xor(zlo, o, 64)); zhi = fmaxf(zhi, __shfl_xor(zhi, o, 64));
        wlo = fminf(wlo, __shfl_xor(wlo, o, 64));
        xlo = fminf(xlo, __shfl_xor(xlo, o, 64)); xhi = fmaxf(xhi, __shfl_xor(xhi, o, 64));
        ylo = fminf(ylo, __shfl_xor(ylo, o, 64)); yhi = fmaxf(yhi, __shfl_xor(yhi, o, 64));
    }
    const float zmargin = 1e-5f * (fabsf(zlo) + fabsf(zhi) + 1.0f);
    if (zhi < a.near_plane - zmargin) return true;  // every member fails `vz > near_plane`
    if (!(zlo > zmargin && wlo > 1e-6f)) return false;
    // |W|_2^2 <= 1 + |W^T W - I|_F for the view matrix' 3 x 3 block (W[i][j] = m[j * 4 + i])
    const float g00 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2] - 1.0f, g11 = m[4] * m[4] + m[5] * m[5] + m[6] * m[6] - 1.0f;
    const float g22 = m[8] * m[8] + m[9] * m[9] + m[10] * m[10] - 1.0f;
    const float g01 = m[0] * m[4] + m[1] * m[5] + m[2] * m[6], g02 = m[0] * m[8] + m[1] * m[9] + m[2] * m[10];
    const float g12 = m[4] * m[8] + m[5] * m[9] + m[6] * m[10];
    const float wn2 = 1.0f + sqrtf(g00 * g00 + g11 * g11 + g22 * g22 + 2.0f * (g01 * g01 + g02 * g02 + g12 * g12));
    const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
    const float j22 = (fmaxf(a.fx * a.fx * (1.0f + limx * limx), a.fy * a.fy * (1.0f + limy * limy)) +
                       a.fx * a.fy * limx * limy) / ((zlo - zmargin) * (zlo - zmargin));
    const float sr = rho * fabsf(a.scale_modifier);
    const float rb = 3.0f * sqrtf(sr * sr * wn2 * j22 * 1.001f + 0.6163f) * 1.0001f + 2.0f;
    const float xend = (float)(GSR_TILE * a.gx) + 1.0f, yend = (float)(GSR_TILE * a.gy) + 1.0f;
    return xhi + rb < -1.0f || xlo - rb > xend || yhi + rb < -1.0f || ylo - rb > yend;
}

// ---------------------------------------------------------------------------------------------------------
// The kernel.  grid = (blocks of 256 Gaussians, GROUPS of frames): the frames of one launch (gsr_forward_batch) that read
// the SAME model -- the cameras x environments of a simulation step, gs_world_wrapper.py:238-242 -- form a group, and one
// workgroup takes a block of the model through all of them (round 6; SURVEY.md 8f-4 "shares the preprocess work"):
//   * the block's view-frustum test for every frame of the group up front (wave w: frames w, w + 4); a block no frame
//     sees costs one short workgroup per GROUP instead of one per frame (seven blocks in ten from a sensor camera);
//   * a Gaussian's position after the part transform, its 3D covariance and opacity are computed once per pose table
//     (WorldGauss: the two cameras of a closed-loop step share them; with E environments a Gaussian of the static scene is
//     transformed once for all of them) and its parameters are read once -- the first frame brings them, the others find
//     them in registers or, the SH coefficients of a Gaussian several cameras see, in this CU's cache;
//   * per frame: projection, EWA covariance, conic, radius, rect, block compaction, SH -> RGB, in exactly the operations of
//     the single-frame kernel: every frame's state is bit-identical to a launch of its own (tests/test_batch_gpu.py).
// A single frame is a group of one.  (Measured in round 5 and not kept: one workgroup per frame and block, the workgroups
// of a block neighbours on one XCD so that the model comes out of that L2: -4 % for four identical frames, +9 % for the
// two cameras of a closed-loop step.)
// ---------------------------------------------------------------------------------------------------------
struct PrepLaunch {
    GsrBatch<PreprocessArgs> bt;        // the frames, those of a group next to each other
    uint8_t first[GSR_MAX_BATCH];       // per group: its first frame in bt ...
    uint8_t count[GSR_MAX_BATCH];       // ... and how many
};
static_assert(sizeof(PrepLaunch) <= 4096, "kernel arguments travel in the 4 KiB kernarg segment");

// LDS of a workgroup: the block-compacted survivors of the frame in hand, and what is kept from frame to frame
// every tile of a rect (tile units, exclusive maxima; clamped to the grid: a Gaussian that has never been visible on this state
// holds whatever the buffer held) gets the frame's token
__device__ __forceinline__ void prep_mark_tiles(uint32_t *__restrict__ dirty, const uint2 rc, const int gx, const int gy,
                                                const uint32_t tok) {
    const int x0 = (int)(rc.x & 0xffffu), y0 = (int)(rc.x >> 16);
    const int x1 = min((int)(rc.y & 0xffffu), gx), y1 = min((int)(rc.y >> 16), gy);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) dirty[y * gx + x] = tok;
}

// (tile reuse, blocks whose members carry different labels -- never kept, a few dozen per model, and the ones that hold the
//  static scene's largest Gaussians where a size class of it borders on a part: has THIS Gaussian's own pose row changed
//  against the previous frame's?  If not, its recomputed records are the previous frame's and it marks nothing.)
__device__ __forceinline__ bool prep_gaussian_moved(const PreprocessArgs &a, const uint32_t idx) {
    const int label = (int)a.part_labels[idx];
    const int part = (label >= 0 && label < a.part_lut_size) ? a.part_lut[label] : -1;
    if (part < 0 || part >= a.part_count) return false;
    const uint32_t *prev = reinterpret_cast<const uint32_t *>(a.pc_slots + (size_t)(a.hdr->pc_parity & 1u) * GSR_PC_SLOT) + 40 + 17 * part;
    const float *cur = a.part_transforms + (size_t)part * 17;
    bool moved = false;
    for (int j = 0; j < 17; j++) moved = moved || __float_as_uint(cur[j]) != prev[j];
    return moved;
}

struct PrepShared {
    uint32_t w[4];
    float4 pos[GSR_BLOCK];
    int idx[GSR_BLOCK];
    int culled[GSR_MAX_BATCH];
    int cached[GSR_MAX_BATCH];  // (the block keeps the previous frame's records for this frame -- prep_block_cached)
};
typedef float PrepWorld[11][GSR_BLOCK];  // (multi-frame groups only: WorldGauss in LDS, each thread its own slots)

// One frame of the group on this workgroup's block.  have_world: an earlier frame of the group left the block's WorldGauss
// in LDS (computed under pose table w_table and scale modifier w_mod); more: a later frame will want it.
template <bool FAST_SH16, bool COUNT_TILES, bool MULTI, bool MARK = false>
__device__ __forceinline__ void prep_frame(const PreprocessArgs &a, const int i, const uint32_t blk, const bool have_world,
                                           const float *w_table, const float w_mod, const bool more, PrepShared &sh,
                                           PrepWorld *world, uint32_t *s_tcnt, const int mark = 0) {
    bool visible = false;
    float4 mypos = make_float4(0.f, 0.f, 0.f, 0.f);  // xyz + radius, handed to the colour phase
    uint32_t my_tiles = 0, my_key = 0;
    uint2 my_rect = make_uint2(0u, 0u);
    if (i < a.P) {
        // the Gaussian's number in the caller's arrays (radii)
        const int oi = (a.orig_index != nullptr && a.radii != nullptr) ? a.orig_index[i] : i;
        WorldGauss w;
        const int tid = (int)threadIdx.x;
        if (MULTI && have_world) {
            w.px = (*world)[0][tid]; w.py = (*world)[1][tid]; w.pz = (*world)[2][tid];
            const int pf = __float_as_int((*world)[3][tid]);
            w.part = pf >> 1;
            w.have_cov = (pf & 1) != 0;
            w.c0 = (*world)[4][tid]; w.c1 = (*world)[5][tid]; w.c2 = (*world)[6][tid];
            w.c3 = (*world)[7][tid]; w.c4 = (*world)[8][tid]; w.c5 = (*world)[9][tid];
            w.opacity_raw = (*world)[10][tid];
        } else {
            w.px = w.py = w.pz = 0.f;
            w.part = -1;
            w.have_cov = false;
            w.c0 = w.c1 = w.c2 = w.c3 = w.c4 = w.c5 = w.opacity_raw = 0.f;
        }
        const bool same_mod = MULTI && have_world && a.scale_modifier == w_mod;
        if (!same_mod || (a.part_transforms != w_table && w.part >= 0)) {
            // (first frame of the block, or a Gaussian of a moving part under another environment's poses)
            const float *xf;
            int part;
            prep_position(a, i, w.px, w.py, w.pz, xf, part);
            w.part = xf != nullptr ? part : -1;
            w.have_cov = false;
        }
        const GeomOut o = prep_geometry(a, i, w);
        if (MULTI && more) {  // (another frame of the group looks at this block)
            (*world)[0][tid] = w.px; (*world)[1][tid] = w.py; (*world)[2][tid] = w.pz;
            (*world)[3][tid] = __int_as_float((int)((uint32_t)w.part << 1) | (w.have_cov ? 1 : 0));
            (*world)[4][tid] = w.c0; (*world)[5][tid] = w.c1; (*world)[6][tid] = w.c2;
            (*world)[7][tid] = w.c3; (*world)[8][tid] = w.c4; (*world)[9][tid] = w.c5;
            (*world)[10][tid] = w.opacity_raw;
        }
        visible = o.visible;
        mypos = o.pos;
        my_tiles = o.tiles;
        my_rect = o.rect;
        // (tile reuse: the tiles this recomputed Gaussian touches NOW; the ones it touched before: prep_live_frames)
        if (MARK && mark != 0 && visible && (mark == 1 || prep_gaussian_moved(a, (uint32_t)i)))
            prep_mark_tiles(a.tile_dirty, my_rect, a.gx, a.gy, a.hdr->td_token + 1u);
        if (a.radii != nullptr) a.radii[oi] = o.radius;
        if (!a.infer) a.tiles_touched[i] = o.tiles;
        my_key = o.key;
    }
    // ---- phase 2: colour, on the block-compacted list of survivors ------------------------------------------
    // Typically only a fraction of the 256 lanes survive cull + rect; evaluating the SH (48 loads + ~110 VALU per
    // Gaussian) in place would run all 4 waves at that fraction of their lanes.  Dense lanes instead.
    uint32_t cnt;
    const uint32_t incl = gsr_block_incl_scan(visible ? 1u : 0u, sh.w, cnt);
    if (visible) {
        sh.pos[incl - 1u] = mypos;
        sh.idx[incl - 1u] = i;
        a.block_recs[(size_t)blk * GSR_BLOCK + (incl - 1u)] = make_uint2((uint32_t)i, my_key);
    }
    if (threadIdx.x == 0) {
        a.block_counts[blk] = cnt;  // consumed by the index-ordered compaction (depth-sorted paths)
        if (COUNT_TILES && cnt != 0u) atomicAdd(&a.hdr->V, cnt);
    }
    __syncthreads();
    if (COUNT_TILES) {
        // per-workgroup tile histogram in LDS, then one global add per touched tile
        uint32_t *cnt_lds = s_tcnt;
        gsr_for_each_tile(my_tiles, my_rect, a.gx, 0u, 0u,
                          [cnt_lds](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&cnt_lds[tile], 1u); });
        __syncthreads();
        for (int t = (int)threadIdx.x; t < a.num_tiles; t += GSR_BLOCK) {
            const uint32_t c = s_tcnt[t];
            if (c != 0u) atomicAdd(&a.tile_accum[(size_t)(blk % GSR_BIN_SLOTS) * a.num_tiles + t], c);
        }
    }
    if (threadIdx.x < cnt) {
        prep_colour<FAST_SH16, MULTI && GSR_PREP_GROUP_HALVES>(a, sh.idx[threadIdx.x], sh.pos[threadIdx.x]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Block cache (round 6).  A closed loop renders the same model from a FIXED sensor camera step after step, and most of what
// that camera sees -- table, walls, whatever is not a robot link or a tracked object -- has not moved: the records preprocess
// would write for such a block (splat, rect, block-local depth records, block count) are bit for bit the ones it wrote a
// step ago, and they are still in the state.  A block is therefore left as it is when (1) the caller vouches for the model
// (a model version in GsrInputs.param_space: the arrays, labels, LUT and block bounds hold what they held in the previous
// frame on this state that carried the same version), (2) settings, sizes and pointers are the previous frame's (pc_sig),
// (3) view matrix, projection and camera centre are the previous frame's bit for bit, and (4) the block's Gaussians carry one
// label (GsrInputs.cull_blocks names it) whose pose row is the previous frame's bit for bit, or no moving part at all.  The
// previous frame's camera and pose table live in one of two slots of the state (GeomState::pc_slots); this frame's go to the
// other one (workgroup 0), and the first kernel behind preprocess (ss_prepare) makes that one current -- nothing a workgroup
// of this launch reads is written during it.  Every other preprocess launch on the state (several frames per workgroup,
// training frames, the A/B paths) clears the mark.  What it saves on the configs[2] surrogate: the static right_cam's share of
// the two-frame launch, 39 -> 27 us per step.  Frames are bit-identical by construction (tests/test_closed_loop_gpu.py).
// ---------------------------------------------------------------------------------------------------------
// (one WAVE asks, a lane per compared word -- 35 of the camera, 17 of the pose row: two loads per lane; the first version had
//  every thread of every workgroup walk all 52 -- 39 -> 50 us instead of 39 -> 27)
// -> bit 0: the block keeps its records; bit 1: camera, background and everything in the signature are the previous frame's
// (what the tile reuse asks: a recomputed block then marks the tiles it touches instead of every tile being composited);
// bit 2: the block's members carry different labels
__device__ __forceinline__ uint32_t prep_block_cached(const PreprocessArgs &a, const uint32_t blk) {
    const GsrHeader *h = a.hdr;
    const int lane = gsr_lane();
    const uint32_t magic = h->pc_magic, sig = h->pc_sig, parity = h->pc_parity & 1u;
    const uint32_t *prev = reinterpret_cast<const uint32_t *>(a.pc_slots + (size_t)parity * GSR_PC_SLOT);
    const float lf = a.cull_blocks[8 * (size_t)blk + 7];
    const bool valid = magic == GSR_PC_MAGIC && sig == a.pc_sig;
    const bool one_label = lf == lf;  // (NaN: members carry different labels)
    const int label = (valid && one_label) ? (int)lf : -1;
    const int part = (label >= 0 && label < a.part_lut_size) ? a.part_lut[label] : -1;
    const bool moving = part >= 0 && part < a.part_count;
    uint32_t cur = 0u, was = 0u;
    if (lane < 16) {
        cur = __float_as_uint(a.view[lane]);
        was = prev[lane];
    } else if (lane < 32) {
        cur = __float_as_uint(a.proj[lane - 16]);
        was = prev[lane];
    } else if (lane < 35) {
        cur = __float_as_uint(a.campos[lane - 32]);
        was = prev[lane];
    } else if (lane < 38) {
        cur = __float_as_uint(a.bg[lane - 35]);
        was = prev[lane];
    } else if (lane == 38) {
        cur = a.td_sig;
        was = prev[38];
    } else if (lane < 56 && moving) {
        cur = __float_as_uint(a.part_transforms[(size_t)part * 17 + (lane - 39)]);
        was = prev[40 + 17 * part + (lane - 39)];
    }
    const uint64_t differ = __builtin_amdgcn_ballot_w64(cur != was);
    // (the records know nothing of background and output buffer -- lanes 35..38; the pixels nothing of the pose rows)
    const bool kept = valid && one_label && (differ & ~(0xFull << 35)) == 0ull;
    const bool frame_same = valid && (differ & ((1ull << 39) - 1ull)) == 0ull;
    return (kept ? 1u : 0u) | (frame_same ? 2u : 0u) | (one_label ? 0u : 4u);
}
// workgroup 0: this frame's camera and pose table into the slot that is NOT current
__device__ __forceinline__ void prep_cache_publish(const PreprocessArgs &a) {
    GsrHeader *h = a.hdr;
    // (a state without a valid mark: which slot is "current" does not matter, nobody reads either)
    const uint32_t parity = h->pc_magic == GSR_PC_MAGIC ? (h->pc_parity & 1u) : 0u;
    float *cur = a.pc_slots + (size_t)(parity ^ 1u) * GSR_PC_SLOT;
    const int tid = (int)threadIdx.x;
    if (tid < 16) {
        cur[tid] = a.view[tid];
        cur[16 + tid] = a.proj[tid];
    }
    if (tid < 3) {
        cur[32 + tid] = a.campos[tid];
        cur[35 + tid] = a.bg[tid];
    }
    if (tid == 3) reinterpret_cast<uint32_t *>(cur)[38] = a.td_sig;
    for (int k = tid; k < 17 * a.part_count; k += GSR_BLOCK) cur[40 + k] = a.part_transforms[k];
    if (tid == 0) {
        h->pc_sig_next = a.pc_sig;
        h->pc_pending = GSR_PC_MAGIC;
        if (h->pc_magic != GSR_PC_MAGIC) h->pc_parity = 0u;
    }
}

// the block's view-frustum test for every frame of the group -> the frames that have to look at it; the others get their
// zero count (and, where a caller reads them, zero radii) on the spot
template <bool CACHE = true, bool TD = CACHE>
__device__ __forceinline__ uint32_t prep_live_frames(const PrepLaunch &L, const int f0, const int nf, const uint32_t blk,
                                                     const int i, PrepShared &sh) {
    uint32_t live = (1u << nf) - 1u;
    if (L.bt.f[f0].cull_blocks == nullptr) return live;  // (the group shares the model: frame f0's block names it)
    // workgroup 0 leaves the camera and poses of the frames that take the block cache for their next frames
    if (CACHE && blk == 0u)
        for (int k = 0; k < nf; k++)
            if (L.bt.f[f0 + k].pc_slots != nullptr) prep_cache_publish(L.bt.f[f0 + k]);
    // (a wave evaluates the block's test for its frames, ~100 instructions each; the verdicts meet at a barrier.  A lone
    //  frame's cache question is asked by the last wave meanwhile, a group's frames ask theirs behind their frustum test.)
    constexpr int NW = GSR_BLOCK / GSR_WAVE;
    for (int k = gsr_wave(); k < nf; k += NW) {
        const PreprocessArgs a = L.bt.f[f0 + k];
        const bool c = prep_block_culled(a, (int)blk);
        const uint32_t h = (CACHE && nf > 1 && a.pc_slots != nullptr) ? prep_block_cached(a, blk) : 0u;
        if (gsr_lane() == 0) {
            sh.culled[k] = c ? 1 : 0;
            if (nf > 1) {
                sh.cached[k] = (int)h;
                if (TD && blk == 0u && a.tile_dirty != nullptr)  // (may this frame's compositor skip what nobody marks?)
                    a.hdr->td_reuse = ((h & 2u) != 0u && a.hdr->overflow == 0u && a.hdr->coop_timeout_now == 0u) ? 1u : 0u;
            }
        }
    }
    if (CACHE && nf == 1 && gsr_wave() == NW - 1) {
        const PreprocessArgs &a = L.bt.f[f0];
        const uint32_t h = a.pc_slots != nullptr ? prep_block_cached(a, blk) : 0u;
        if (gsr_lane() == 0) {
            sh.cached[0] = (int)h;
            if (TD && blk == 0u && a.tile_dirty != nullptr)
                a.hdr->td_reuse = ((h & 2u) != 0u && a.hdr->overflow == 0u && a.hdr->coop_timeout_now == 0u) ? 1u : 0u;
        }
    }
    __syncthreads();
    if (TD) {
        // Tile reuse: a block that IS recomputed under an unchanged camera first marks the tiles its Gaussians touched in the
        // previous frame -- the block's visible records of that frame are still in place (count, block-local records, rects) --
        // before anything of it is rewritten; the tiles they touch now are marked where the new rects appear (prep_frame).
        bool any = false;
        for (int k = 0; k < nf; k++) {
            const PreprocessArgs &a = L.bt.f[f0 + k];
            if ((sh.cached[k] & 3) != 2 || a.tile_dirty == nullptr) continue;  // (kept as it is, or every tile composited anyway)
            any = true;
            const uint32_t old_cnt = min(a.block_counts[blk], (uint32_t)GSR_BLOCK);
            if (threadIdx.x < old_cnt) {
                const uint32_t g = a.block_recs[(size_t)blk * GSR_BLOCK + threadIdx.x].x;
                if ((g >> 8) == blk && ((sh.cached[k] & 4) == 0 || prep_gaussian_moved(a, g)))
                    prep_mark_tiles(a.tile_dirty, a.rects[g], a.gx, a.gy, a.hdr->td_token + 1u);
            }
        }
        if (any) __syncthreads();  // (the old records are read: they may be rewritten)
    }
    for (int k = 0; k < nf; k++) {
        if (CACHE && (sh.cached[k] & 1) != 0) {  // (everything this frame would write for the block is in place: the previous frame's)
            live &= ~(1u << k);
            if (threadIdx.x == 0) L.bt.f[f0 + k].hdr->pc_hit = 1u;
            continue;
        }
        if (sh.culled[k] == 0) continue;
        live &= ~(1u << k);
        const PreprocessArgs &a = L.bt.f[f0 + k];
        if (i < a.P) {
            if (a.radii != nullptr) a.radii[a.orig_index != nullptr ? a.orig_index[i] : i] = 0;
            if (!a.infer) a.tiles_touched[i] = 0u;
        }
        if (threadIdx.x == 0) a.block_counts[blk] = 0u;
    }
    return live;
}

// groups of ONE frame (a single gsr_forward, frames of a batch that share nothing): no loop around the frame -- the loop
// alone costs the compiler 40 VGPRs (hoisted constants, lane predicates, addresses), 113 against 72
#ifndef GSR_PREP_MAX_WAVES
#define GSR_PREP_MAX_WAVES 0  // (A/B probe: cap the single-frame kernel's waves per SIMD -- what does occupancy buy it?)
#endif
// (TD: some frame of the launch marks tiles for the compositor's tile reuse -- an instance of its own: the marking costs the
//  single-frame kernel its 72nd register, i.e. a wave per SIMD, which frames without it must not pay)
template <bool FAST_SH16, bool COUNT_TILES, bool TD = false>
__global__ __launch_bounds__(GSR_BLOCK)
#if GSR_PREP_MAX_WAVES > 0
__attribute__((amdgpu_waves_per_eu(1, GSR_PREP_MAX_WAVES)))
#endif
void preprocess_kernel(const PrepLaunch L) {
    const int f0 = (int)L.first[blockIdx.y];
    const uint32_t blk = blockIdx.x;
    extern __shared__ uint32_t s_tcnt[];  // [num_tiles] when COUNT_TILES
    __shared__ PrepShared sh;
    const int i = (int)blk * GSR_BLOCK + (int)threadIdx.x;
    const PreprocessArgs a = L.bt.f[f0];  // (by value: every field is requested at the top, not at its first use)
    if (COUNT_TILES) {
        for (int t = (int)threadIdx.x; t < a.num_tiles; t += GSR_BLOCK) s_tcnt[t] = 0u;
    }
    if (a.pc_slots == nullptr && blk == 0u && threadIdx.x == 0)
        a.hdr->pc_magic = 0u;  // (this launch rewrites the records without leaving its camera: nothing to compare with next)
    if (prep_live_frames<true, TD>(L, f0, 1, blk, i, sh) == 0u) return;  // (+ the block cache: cull_blocks is there whenever pc_slots is)
    prep_frame<FAST_SH16, COUNT_TILES, false, TD>(a, i, blk, false, nullptr, 0.f, false, sh, nullptr, s_tcnt,
                                                  (TD && a.tile_dirty != nullptr && (sh.cached[0] & 2) != 0)
                                                      ? ((sh.cached[0] & 4) != 0 ? 2 : 1) : 0);
}

// groups of several frames
#ifndef GSR_PREP_GROUP_WAVES
#define GSR_PREP_GROUP_WAVES 6  // minimum waves per SIMD asked of the compiler for the multi-frame kernel: 80 VGPRs, no scratch (0: no limit -- 84; 7: 72 + 28 B of scratch)
#endif
// (CACHE: some frame of the launch takes the block cache -- an instance of its own: the questions cost the frame loop five
//  registers it does not have, 20 bytes of scratch per lane, which launches without the cache must not pay)
#ifndef GSR_PREP_GROUP_CACHE_WAVES
#define GSR_PREP_GROUP_CACHE_WAVES 5  // waves per SIMD asked for the CACHE instance: 85 VGPRs, no scratch (0 = the same six as
                                      // the plain instance, with the scratch: four environments 13.12 against 13.23-13.33 k)
#endif
template <bool FAST_SH16, bool CACHE = false>
__global__
#if GSR_PREP_GROUP_WAVES > 0
__launch_bounds__(GSR_BLOCK, (CACHE && GSR_PREP_GROUP_CACHE_WAVES > 0) ? GSR_PREP_GROUP_CACHE_WAVES : GSR_PREP_GROUP_WAVES)
#else
__launch_bounds__(GSR_BLOCK)
#endif
void preprocess_group_kernel(const PrepLaunch L) {
    const int f0 = (int)L.first[blockIdx.y], nf = (int)L.count[blockIdx.y];
    const uint32_t blk = blockIdx.x;
    __shared__ PrepShared sh;
    __shared__ PrepWorld s_world;
    const int i0 = (int)blk * GSR_BLOCK + (int)threadIdx.x;
    if (blk == 0u && (int)threadIdx.x < nf && L.bt.f[f0 + (int)threadIdx.x].pc_slots == nullptr)
        L.bt.f[f0 + (int)threadIdx.x].hdr->pc_magic = 0u;  // (a frame that rewrites its records without leaving its camera)
    const uint32_t live = prep_live_frames<CACHE>(L, f0, nf, blk, i0, sh);
    if (live == 0u) return;
    // ---- the frames that see the block, one after the other; what no camera enters is kept between them -- in LDS, each
    // thread its own slots (in registers the eleven words would stay live across the colour phase, whose 48 SH
    // coefficients in flight set the kernel's register count)
    bool have_world = false;            // (wave-uniform)
    const float *w_table = nullptr;     // pose table and scale modifier the kept values were computed under
    float w_mod = 0.f;
#pragma nounroll
    for (int k = 0; k < nf; k++) {
        if (!((live >> k) & 1u)) continue;
        const PreprocessArgs a = L.bt.f[f0 + k];  // (by value: every field is requested at the top, not at its first use)
        // (the Gaussian's number is made opaque per frame: otherwise every address of the body is hoisted out of the loop)
        int i = i0;
        asm volatile("" : "+v"(i));
        prep_frame<FAST_SH16, false, true, CACHE>(a, i, blk, have_world, w_table, w_mod, (live >> (k + 1)) != 0u, sh, &s_world,
                                                  nullptr, (CACHE && a.tile_dirty != nullptr && (sh.cached[k] & 2) != 0)
                                                               ? ((sh.cached[k] & 4) != 0 ? 2 : 1) : 0);
        have_world = true;
        w_table = a.part_transforms;
        w_mod = a.scale_modifier;
        __syncthreads();  // (the next frame's survivors take the same LDS slots)
    }
}

__global__ __launch_bounds__(GSR_BLOCK) void mark_visible_kernel(int P, const float *means3D, const float *m,
                                                                 float near_plane, uint8_t *present) {
    const int i = blockIdx.x * GSR_BLOCK + threadIdx.x;
    if (i >= P) return;
    const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1], pz = means3D[3 * (size_t)i + 2];
    const float vz = fma_(m[10], pz, fma_(m[6], py, m[2] * px)) + m[14];
    present[i] = vz > near_plane ? 1 : 0;
}

}  // namespace

#ifndef GSR_PREP_GROUPS
#define GSR_PREP_GROUPS 1  // (A/B: 0 = every frame of a launch its own group, workgroups per (block, frame) as in round 5)
#endif
// Frames that share a model form a group only from this many on.  The group kernel carries a loop over the frames, which
// costs the compiler 40 VGPRs (113 against the single-frame kernel's 72: four workgroups per CU instead of seven), and in a
// latency-bound launch residency is what counts: the two cameras of a closed-loop step are FASTER as 2 x 5 738 single-frame
// workgroups (39.5 us) than as 5 738 two-frame ones (44.4 us; the single-frame kernel capped at four waves per SIMD: 46.6 us,
// i.e. the sharing itself is worth 5 % at equal residency); eight frames per launch are faster shared.
#ifndef GSR_PREP_GROUP_MIN
#define GSR_PREP_GROUP_MIN 3
#endif
// two frames share a workgroup when everything but the camera, the pose TABLE and the outputs is the same
static bool prep_same_model(const PreprocessArgs &x, const PreprocessArgs &y) {
    return x.P == y.P && x.D == y.D && x.M == y.M && x.W == y.W && x.H == y.H && x.param_space == y.param_space &&
           x.infer == y.infer && x.means3D == y.means3D && x.shs == y.shs && x.shs_rest == y.shs_rest &&
           x.colors_precomp == y.colors_precomp && x.opacities == y.opacities && x.scales == y.scales &&
           x.rotations == y.rotations && x.cov3D_precomp == y.cov3D_precomp && x.part_labels == y.part_labels &&
           x.part_lut == y.part_lut && x.part_lut_size == y.part_lut_size && x.part_count == y.part_count &&
           x.part_rescale == y.part_rescale && x.cull_blocks == y.cull_blocks && x.orig_index == y.orig_index;
}

int gsr_launch_preprocess(int B, GsrFrame *fr, bool count_tiles, bool infer, hipStream_t stream) {
    PreprocessArgs args[GSR_MAX_BATCH];
    bool fast = true;
    for (int k = 0; k < B; k++) {
        const GsrSettings &st = *fr[k].st;
        const GsrInputs &in = *fr[k].in;
        const GeomState &g = fr[k].g;
        PreprocessArgs &a = args[k];
        a.infer = infer ? 1 : 0;
        a.P = in.P;
        a.D = st.sh_degree;
        a.M = st.sh_coeffs;
        a.W = st.image_width;
        a.H = st.image_height;
        a.gx = gsr_div_up(a.W, GSR_TILE);
        a.gy = gsr_div_up(a.H, GSR_TILE);
        a.tanfovx = st.tanfovx;
        a.tanfovy = st.tanfovy;
        a.fx = (float)a.W / (2.0f * st.tanfovx);
        a.fy = (float)a.H / (2.0f * st.tanfovy);
        a.scale_modifier = st.scale_modifier;
        a.near_plane = st.near_plane;
        a.antialiasing = st.antialiasing;
        a.means3D = in.means3D;
        a.shs = in.shs;
        a.shs_rest = in.shs_rest;
        a.param_space = in.param_space;
        a.colors_precomp = in.colors_precomp;
        a.opacities = in.opacities;
        a.scales = in.scales;
        a.rotations = in.rotations;
        a.cov3D_precomp = in.cov3D_precomp;
        a.view = in.viewmatrix;
        a.proj = in.projmatrix;
        a.campos = in.campos;
        a.part_labels = in.part_labels;
        a.part_lut = in.part_lut;
        a.part_lut_size = in.part_lut_size;
        a.part_transforms = in.part_transforms;
        a.part_count = in.part_count;
        a.part_rescale = in.part_rescale;
        a.cull_blocks = in.cull_blocks;
        a.orig_index = in.orig_index;
        a.radii = fr[k].out->radii;
        a.splat = g.splat;
        a.cov3D = g.cov3D;
        a.clamped = g.clamped;
        a.tiles_touched = g.tiles_touched;
        a.rects = g.rects;
        a.block_counts = g.block_counts;
        a.block_recs = g.block_recs;  // (the sort's compaction gathers from here into pair[0])

        a.num_tiles = a.gx * a.gy;
        a.tile_accum = g.tile_accum;
        a.hdr = g.hdr;
        a.pc_slots = nullptr;
        a.pc_sig = 0u;
        a.td_sig = 0u;
        a.tile_dirty = (fr[k].pc && fr[k].td) ? fr[k].img.tile_dirty : (uint32_t *)nullptr;
        a.bg = in.background;
        if (fr[k].pc) {
            a.pc_slots = g.pc_slots;
            // everything but camera and poses that the records depend on (FNV-1a over the values)
            uint32_t hsh = 2166136261u;
            auto mix = [&hsh](const void *p, size_t n) {
                const unsigned char *c = (const unsigned char *)p;
                for (size_t q = 0; q < n; q++) hsh = (hsh ^ c[q]) * 16777619u;
            };
            const uintptr_t ptrs[] = {(uintptr_t)in.means3D, (uintptr_t)in.shs, (uintptr_t)in.shs_rest, (uintptr_t)in.colors_precomp,
                                      (uintptr_t)in.opacities, (uintptr_t)in.scales, (uintptr_t)in.rotations,
                                      (uintptr_t)in.cov3D_precomp, (uintptr_t)in.part_labels, (uintptr_t)in.part_lut,
                                      (uintptr_t)in.part_rescale, (uintptr_t)in.cull_blocks, (uintptr_t)in.orig_index,
                                      (uintptr_t)((char *)g.block_recs - (char *)g.hdr)};
            const int32_t ints[] = {in.P, st.sh_degree, st.sh_coeffs, st.image_width, st.image_height, st.antialiasing,
                                    in.param_space, in.part_lut_size, in.part_count, infer ? 1 : 0};
            const float flts[] = {st.tanfovx, st.tanfovy, st.scale_modifier, st.near_plane};
            mix(ptrs, sizeof(ptrs));
            mix(ints, sizeof(ints));
            mix(flts, sizeof(flts));
            a.pc_sig = hsh;
            // ... and what the tile reuse depends on beyond camera and background: where the frame goes (a word of the slot
            // by itself: a caller that rotates its output buffers keeps its blocks, it just composites every tile)
            const uintptr_t outs[] = {(uintptr_t)fr[k].out->out_rgb8, (uintptr_t)fr[k].img.tile_dirty};
            mix(outs, sizeof(outs));
            a.td_sig = hsh;
        }
        // (the 12 x dwordx4 colour path needs every frame's SH array aligned)
        fast = fast && (in.colors_precomp == nullptr) && st.sh_degree == 3 && st.sh_coeffs == 16 &&
               ((reinterpret_cast<uintptr_t>(in.shs) & 15u) == 0);
    }
    // groups: the frames that read the same model, next to each other in the kernel's table (a frame's place in the table
    // means nothing: its argument block carries its own state and outputs)
    PrepLaunch L;
    memset(&L, 0, sizeof(L));
    int groups = 0, filled = 0;
    bool taken[GSR_MAX_BATCH] = {false};
    for (int k = 0; k < B; k++) {
        if (taken[k]) continue;
        int same = 0;
        for (int j = k; j < B; j++) same += (!taken[j] && prep_same_model(args[k], args[j])) ? 1 : 0;
        const bool share = GSR_PREP_GROUPS && !count_tiles && same >= GSR_PREP_GROUP_MIN;
        L.first[groups] = (uint8_t)filled;
        for (int j = k; j < B; j++)
            if (!taken[j] && (j == k || (share && prep_same_model(args[k], args[j])))) {
                taken[j] = true;
                L.bt.f[filled++] = args[j];
            }
        L.count[groups] = (uint8_t)(filled - (int)L.first[groups]);
        groups++;
    }
    const PreprocessArgs &a0 = args[0];
    const dim3 grid(GeomState::prep_blocks(a0.P), groups);
    const size_t lds = count_tiles ? (size_t)a0.num_tiles * sizeof(uint32_t) : 0;
    if (count_tiles) {  // (launches without the block cache: nobody publishes, so nobody may flip -- api.hip)
        for (int k = 0; k < B; k++) {
            L.bt.f[k].pc_slots = nullptr;
            L.bt.f[k].tile_dirty = nullptr;
            fr[k].pc = fr[k].td = false;
        }
    }
    if (groups < B) {  // (some frames share a model; count_tiles frames never do)
        bool cache = false;
        for (int k = 0; k < B; k++) cache = cache || L.bt.f[k].pc_slots != nullptr;
        if (fast && cache)
            hipLaunchKernelGGL((preprocess_group_kernel<true, true>), grid, dim3(GSR_BLOCK), 0, stream, L);
        else if (fast)
            hipLaunchKernelGGL((preprocess_group_kernel<true, false>), grid, dim3(GSR_BLOCK), 0, stream, L);
        else if (cache)
            hipLaunchKernelGGL((preprocess_group_kernel<false, true>), grid, dim3(GSR_BLOCK), 0, stream, L);
        else
            hipLaunchKernelGGL((preprocess_group_kernel<false, false>), grid, dim3(GSR_BLOCK), 0, stream, L);
    } else if (count_tiles) {
        if (fast)
            hipLaunchKernelGGL((preprocess_kernel<true, true>), grid, dim3(GSR_BLOCK), lds, stream, L);
        else
            hipLaunchKernelGGL((preprocess_kernel<false, true>), grid, dim3(GSR_BLOCK), lds, stream, L);
    } else {
        bool td = false;
        for (int k = 0; k < B; k++) td = td || L.bt.f[k].tile_dirty != nullptr;
        if (fast && td)
            hipLaunchKernelGGL((preprocess_kernel<true, false, true>), grid, dim3(GSR_BLOCK), 0, stream, L);
        else if (fast)
            hipLaunchKernelGGL((preprocess_kernel<true, false>), grid, dim3(GSR_BLOCK), 0, stream, L);
        else if (td)
            hipLaunchKernelGGL((preprocess_kernel<false, false, true>), grid, dim3(GSR_BLOCK), 0, stream, L);
        else
            hipLaunchKernelGGL((preprocess_kernel<false, false>), grid, dim3(GSR_BLOCK), 0, stream, L);
    }
    return GSR_OK;
}

extern "C" int gsr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, float near_plane,
                                uint8_t *present, void *stream) {
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) {
        gsr_set_error("gsr_mark_visible: null pointer or negative P");
        return GSR_E_INVALID;
    }
    if (P == 0) return GSR_OK;
    hipLaunchKernelGGL(mark_visible_kernel, dim3(gsr_div_up(P, GSR_BLOCK)), dim3(GSR_BLOCK), 0,
                       (hipStream_t)stream, P, means3D, viewmatrix, near_plane, present);
    return gsr_check_launch("mark_visible", false, (hipStream_t)stream);
}
