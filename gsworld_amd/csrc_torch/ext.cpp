// ext.cpp -- the compiled `_C` module of the drop-in `diff_gaussian_rasterization` package:
// what upstream's ext.cpp / rasterize_points.cu export through pybind, over the C ABI of libgsr_hip.so
// (include/gsr.h).  Same positional signatures, same return tuples, same error strings:
//
//   rasterize_gaussians            <->  RasterizeGaussiansCUDA           (rasterize_points.cu)
//   rasterize_gaussians_backward   <->  RasterizeGaussiansBackwardCUDA
//   mark_visible                   <->  markVisible
//
// plus `forward_frame`, the persistent-state entry gsworld_amd.renderer.FrameRenderer uses (caller-owned outputs and
// state tensors, optional no-sync capacity mode): one native call per frame instead of a ctypes struct marshalling and
// three Python resize callbacks.  No device code here -- this file is host glue; it is built by
// gsworld_amd/build_ext.py (g++ against the torch headers, linked to ../libgsr_hip.so) and is optional: without it
// gsworld_amd/_C.py falls back to the ctypes binding of the same library.
#include <torch/extension.h>
// (PyTorch-ROCm keeps the device type "cuda": the guard / stream types of that naming live in these headers)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cstring>
#include <tuple>

#include "../../include/gsr.h"

namespace {

char *resize_cb(void *user, size_t bytes) {
    auto *t = static_cast<torch::Tensor *>(user);
    t->resize_({static_cast<long long>(bytes)});
    return reinterpret_cast<char *>(t->data_ptr());
}

// The geometry state: storage the allocator hands out may be a freed state of another renderer, whose frame header
// still says how many of ITS frames overflowed -- on new storage that count (the header's last two words) is zeroed,
// enqueued on the frame's stream ahead of the frame's first kernel; the kept splitters / cuts of a recycled header
// are checked before use and stay (a training loop gets the block it freed a step ago back every step).
char *resize_geom_cb(void *user, size_t bytes) {
    auto *t = static_cast<torch::Tensor *>(user);
    const void *before = t->numel() ? t->data_ptr() : nullptr;
    t->resize_({static_cast<long long>(bytes)});
    if (t->data_ptr() != before && bytes >= 256) t->narrow(0, 248, 8).zero_();
    return reinterpret_cast<char *>(t->data_ptr());
}

const float *fptr(const torch::Tensor &t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

// float32, on `dev`, dense -- what upstream's `.contiguous().data<float>()` implies
torch::Tensor f32(const torch::Tensor &t, const torch::Device &dev, const char *what) {
    if (t.numel() == 0) return t;
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, what, " must be float32");
    return (t.device() == dev ? t : t.to(dev)).contiguous();
}

void require_gpu(const torch::Tensor &t, const char *what) {
    TORCH_CHECK(t.is_cuda(), what, " is on ", t.device(),
                ": the MI355X rasterizer has no CPU path (tensors must live on a HIP device)");
}

struct Tuning {
    int binning_path = 0, render_variant = 0, render_blocks_per_cu = 0, depth_sort = 0, render_split = 0,
        forward_only = 0;
};

GsrSettings settings(int H, int W, float tanfovx, float tanfovy, float scale_modifier, int degree, int M,
                     bool prefiltered, bool antialiasing, bool debug, float near_plane, const Tuning &tn) {
    GsrSettings st{};
    st.image_height = H;
    st.image_width = W;
    st.tanfovx = tanfovx;
    st.tanfovy = tanfovy;
    st.scale_modifier = scale_modifier;
    st.sh_degree = degree;
    st.sh_coeffs = M;
    st.prefiltered = prefiltered;
    st.antialiasing = antialiasing;
    st.debug = debug;
    st.near_plane = near_plane;
    st.binning_path = tn.binning_path;
    st.render_variant = tn.render_variant;
    st.render_blocks_per_cu = tn.render_blocks_per_cu;
    st.depth_sort = tn.depth_sort;
    st.render_split = tn.render_split;
    st.forward_only = tn.forward_only;
    return st;
}

Tuning tuning_from(const std::vector<int> &v) {
    Tuning t;
    if (v.size() >= 5) {
        t.binning_path = v[0];
        t.render_variant = v[1];
        t.render_blocks_per_cu = v[2];
        t.depth_sort = v[3];
        t.render_split = v[4];
    }
    if (v.size() >= 6) t.forward_only = v[5];
    return t;
}

void *current_stream(const torch::Device &dev) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream(); }

// ---- one frame into caller-owned tensors (FrameRenderer) ---------------------------------------------------------
// The argument structs of one frame (include/gsr.h) from tensors.  The state tensors are referenced by the resize
// callbacks: they must outlive the call into the library.
struct FrameCall {
    GsrSettings st;
    GsrInputs in;
    GsrOutputs out;
    GsrBuffers buf;
    torch::Tensor geom, binning, image;  // (handles: resized in place by the callbacks)
};

void fill_frame(FrameCall &f, int H, int W, double tanfovx, double tanfovy, double scale_modifier, int degree, int M,
                bool antialiasing, bool debug, double near_plane, const torch::Tensor &bg, const torch::Tensor &means3D,
                const torch::Tensor &colors, const torch::Tensor &opacity, const torch::Tensor &scales,
                const torch::Tensor &rotations, const torch::Tensor &cov3D_precomp, const torch::Tensor &viewmatrix,
                const torch::Tensor &projmatrix, const torch::Tensor &sh, const torch::Tensor &sh_rest,
                const torch::Tensor &campos, const torch::Tensor &out_color, const torch::Tensor &out_invdepth,
                const torch::Tensor &radii, const torch::Tensor &geom, const torch::Tensor &binning,
                const torch::Tensor &image, const torch::Tensor &rgb8_out, int param_space,
                const std::vector<int> &tuning, const torch::Tensor &part_labels, const torch::Tensor &part_lut,
                const torch::Tensor &part_table, const torch::Tensor &part_rescale, const torch::Tensor &cull_blocks,
                const torch::Tensor &orig_index, int64_t overflow_mirror) {
    f.st = settings(H, W, (float)tanfovx, (float)tanfovy, (float)scale_modifier, degree, M, false, antialiasing, debug,
                    (float)near_plane, tuning_from(tuning));
    GsrInputs in{};
    in.P = (int32_t)means3D.size(0);
    in.background = fptr(bg);
    in.means3D = fptr(means3D);
    in.shs = fptr(sh);
    in.colors_precomp = fptr(colors);
    in.opacities = fptr(opacity);
    in.scales = fptr(scales);
    in.rotations = fptr(rotations);
    in.cov3D_precomp = fptr(cov3D_precomp);
    in.viewmatrix = fptr(viewmatrix);
    in.projmatrix = fptr(projmatrix);
    in.campos = fptr(campos);
    in.shs_rest = fptr(sh_rest);
    in.param_space = param_space;
    if (part_labels.numel() != 0) {  // rigid transform of labelled Gaussians inside preprocess (GsrInputs.part_*)
        in.part_labels = part_labels.data_ptr<float>();
        in.part_lut = part_lut.data_ptr<int32_t>();
        in.part_lut_size = (int32_t)part_lut.numel();
        in.part_transforms = fptr(part_table);
        in.part_count = (int32_t)part_table.size(0);
        in.part_rescale = part_rescale.numel() ? part_rescale.data_ptr<uint8_t>() : nullptr;
    }
    if (cull_blocks.numel() != 0) {  // block bounds for view-frustum culling (GsrInputs.cull_blocks)
        TORCH_CHECK(cull_blocks.numel() == 8 * ((means3D.size(0) + 255) / 256), "cull_blocks must be (ceil(P / 256), 8)");
        in.cull_blocks = fptr(cull_blocks);
    }
    if (orig_index.numel() != 0) {  // the model arrays are a permuted copy (GsrInputs.orig_index)
        TORCH_CHECK(orig_index.numel() == means3D.size(0), "orig_index must be (P,)");
        in.orig_index = orig_index.data_ptr<int32_t>();
    }
    f.in = in;
    f.out = GsrOutputs{out_color.numel() ? out_color.data_ptr<float>() : nullptr,
                       out_invdepth.numel() ? out_invdepth.data_ptr<float>() : nullptr,
                       radii.numel() ? radii.data_ptr<int32_t>() : nullptr,
                       rgb8_out.numel() ? rgb8_out.data_ptr<uint8_t>() : nullptr,
                       reinterpret_cast<uint32_t *>((uintptr_t)overflow_mirror)};
    f.geom = geom;
    f.binning = binning;
    f.image = image;
    f.buf = GsrBuffers{resize_geom_cb, &f.geom, resize_cb, &f.binning, resize_cb, &f.image};
}

std::tuple<int64_t, int64_t, int64_t> forward_frame(
    int H, int W, double tanfovx, double tanfovy, double scale_modifier, int degree, int M, bool antialiasing,
    bool debug, double near_plane, const torch::Tensor &bg, const torch::Tensor &means3D, const torch::Tensor &colors,
    const torch::Tensor &opacity, const torch::Tensor &scales, const torch::Tensor &rotations,
    const torch::Tensor &cov3D_precomp, const torch::Tensor &viewmatrix, const torch::Tensor &projmatrix,
    const torch::Tensor &sh, const torch::Tensor &sh_rest, const torch::Tensor &campos, torch::Tensor out_color,
    torch::Tensor out_invdepth, torch::Tensor radii, torch::Tensor geom, torch::Tensor binning, torch::Tensor image,
    const torch::Tensor &rgb8_out, int64_t r_capacity, bool want_stats, int param_space, std::vector<int> tuning,
    const torch::Tensor &part_labels, const torch::Tensor &part_lut, const torch::Tensor &part_table,
    const torch::Tensor &part_rescale, const torch::Tensor &cull_blocks, const torch::Tensor &orig_index,
    int64_t overflow_mirror) {
    const auto dev = means3D.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    FrameCall f;
    fill_frame(f, H, W, tanfovx, tanfovy, scale_modifier, degree, M, antialiasing, debug, near_plane, bg, means3D, colors,
               opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, sh, sh_rest, campos, out_color,
               out_invdepth, radii, geom, binning, image, rgb8_out, param_space, tuning, part_labels, part_lut, part_table,
               part_rescale, cull_blocks, orig_index, overflow_mirror);
    GsrFrameStats stats{};
    const int rc = gsr_forward(&f.st, &f.in, &f.out, &f.buf, r_capacity, want_stats ? &stats : nullptr, current_stream(dev));
    TORCH_CHECK(rc == GSR_OK, "libgsr_hip error ", rc, ": ", gsr_last_error());
    return {stats.num_visible, stats.num_rendered, (int64_t)stats.overflow};
}

// ---- the frames of one step through gsr_forward_batch (gsworld_amd._C.forward_batch_raw) ----------------------------
// Every element of `frames` is the argument tuple of forward_frame without `want_stats` (nothing is read back).
void forward_batch(const py::list &frames) {
    const size_t B = frames.size();
    if (B == 0) return;
    std::vector<FrameCall> calls(B);  // (sized once: the callbacks keep pointers into it)
    std::vector<int64_t> caps(B);
    torch::Device dev(torch::kCPU);
    for (size_t k = 0; k < B; k++) {
        const py::tuple t = frames[k].cast<py::tuple>();
        TORCH_CHECK(t.size() == 39, "forward_batch: a frame is a tuple of 39 values, got ", t.size());
        auto T = [&](int i) { return t[i].cast<torch::Tensor>(); };
        if (k == 0) dev = T(11).device();
        caps[k] = t[29].cast<int64_t>();
        fill_frame(calls[k], t[0].cast<int>(), t[1].cast<int>(), t[2].cast<double>(), t[3].cast<double>(),
                   t[4].cast<double>(), t[5].cast<int>(), t[6].cast<int>(), t[7].cast<bool>(), t[8].cast<bool>(),
                   t[9].cast<double>(), T(10), T(11), T(12), T(13), T(14), T(15), T(16), T(17), T(18), T(19), T(20), T(21),
                   T(22), T(23), T(24), T(25), T(26), T(27), T(28), t[30].cast<int>(), t[31].cast<std::vector<int>>(),
                   T(32), T(33), T(34), T(35), T(36), T(37), t[38].cast<int64_t>());
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    std::vector<GsrSettings> st(B);
    std::vector<GsrInputs> in(B);
    std::vector<GsrOutputs> out(B);
    std::vector<GsrBuffers> buf(B);
    for (size_t k = 0; k < B; k++) {
        st[k] = calls[k].st;
        in[k] = calls[k].in;
        out[k] = calls[k].out;
        buf[k] = calls[k].buf;
    }
    const int rc = gsr_forward_batch((int32_t)B, st.data(), in.data(), out.data(), buf.data(), caps.data(),
                                     current_stream(dev));
    TORCH_CHECK(rc == GSR_OK, "libgsr_hip error ", rc, ": ", gsr_last_error());
}

// ---- the argument pack of a step, kept on this side (gsworld_amd._C.pack_batch) ---------------------------------------
// A loop that renders the SAME frames step after step (same tensors, settings and state; the per-step values live in device
// buffers the kernels read) parsed its 39-value tuples again on every call of forward_batch: ~10 us of host per step, and in
// the closed loop's eager route 15 us of idle device between the staging kernel and preprocess (kernel trace, round 6).  The
// pack keeps the filled argument structs; run() is gsr_forward_batch, run_staged() gsr_stage_step + gsr_forward_batch --
// one call from Python for everything a step enqueues.
struct StepPack {
    std::vector<FrameCall> calls;  // (sized once: the resize callbacks keep pointers into it)
    std::vector<int64_t> caps;
    std::vector<GsrSettings> st;
    std::vector<GsrInputs> in;
    std::vector<GsrOutputs> out;
    std::vector<GsrBuffers> buf;
    py::list keep;  // the tensors the raw pointers came from
    torch::Device dev{torch::kCPU};

    explicit StepPack(const py::list &frames) : keep(frames) {
        const size_t B = frames.size();
        TORCH_CHECK(B > 0, "StepPack: no frames");
        calls.resize(B);
        caps.resize(B);
        for (size_t k = 0; k < B; k++) {
            const py::tuple t = frames[k].cast<py::tuple>();
            TORCH_CHECK(t.size() == 39, "StepPack: a frame is a tuple of 39 values, got ", t.size());
            auto T = [&](int i) { return t[i].cast<torch::Tensor>(); };
            if (k == 0) dev = T(11).device();
            caps[k] = t[29].cast<int64_t>();
            fill_frame(calls[k], t[0].cast<int>(), t[1].cast<int>(), t[2].cast<double>(), t[3].cast<double>(),
                       t[4].cast<double>(), t[5].cast<int>(), t[6].cast<int>(), t[7].cast<bool>(), t[8].cast<bool>(),
                       t[9].cast<double>(), T(10), T(11), T(12), T(13), T(14), T(15), T(16), T(17), T(18), T(19), T(20), T(21),
                       T(22), T(23), T(24), T(25), T(26), T(27), T(28), t[30].cast<int>(), t[31].cast<std::vector<int>>(),
                       T(32), T(33), T(34), T(35), T(36), T(37), t[38].cast<int64_t>());
        }
        st.resize(B); in.resize(B); out.resize(B); buf.resize(B);
        for (size_t k = 0; k < B; k++) {
            st[k] = calls[k].st;
            in[k] = calls[k].in;
            out[k] = calls[k].out;
            buf[k] = calls[k].buf;
        }
    }
    void launch(void *stream) {
        const int rc = gsr_forward_batch((int32_t)calls.size(), st.data(), in.data(), out.data(), buf.data(), caps.data(), stream);
        TORCH_CHECK(rc == GSR_OK, "libgsr_hip error ", rc, ": ", gsr_last_error());
    }
    void run() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        launch(current_stream(dev));
    }
    // gsr_stage_step (include/gsr.h: n floats from the pinned slot `src` to `dst`, K matrices packed into `table`), then the frames
    void run_staged(int64_t n, int64_t src, int64_t dst, int64_t K, int64_t mat_off, int64_t scale_off, int64_t table) {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        void *stream = current_stream(dev);
        const int rc = gsr_stage_step((int32_t)n, reinterpret_cast<const float *>((uintptr_t)src),
                                      reinterpret_cast<float *>((uintptr_t)dst), (int32_t)K, (int32_t)mat_off,
                                      (int32_t)scale_off, reinterpret_cast<float *>((uintptr_t)table), stream);
        TORCH_CHECK(rc == GSR_OK, "libgsr_hip error ", rc, ": ", gsr_last_error());
        launch(stream);
    }
};

// The host values of a closed-loop step (part matrices, uniform scales, camera matrices -- gsworld_amd/closed_loop.py keeps them
// in one float32 staging vector): every (offset, tensor) pair into the host mirror, then the whole mirror into the pinned ring
// slot the step's staging kernel will read.  One call instead of a dozen numpy copies behind as many Python-level checks:
// with the policy in the loop the host's share of a step is on the critical path (13 -> 4 us of ~250, round 6).  Returns false
// -- nothing written -- if a tensor is not plain host float32 or does not fit: the caller takes its general path.
bool stage_host_values(torch::Tensor mirror, torch::Tensor slot, const py::list &segs) {
    TORCH_CHECK(mirror.device().is_cpu() && slot.device().is_cpu() && mirror.scalar_type() == torch::kFloat32 &&
                    slot.scalar_type() == torch::kFloat32 && mirror.is_contiguous() && slot.is_contiguous() &&
                    mirror.numel() == slot.numel(),
                "stage_host_values: mirror and slot must be host float32 vectors of one size");
    const int64_t n = mirror.numel();
    const size_t S = segs.size();
    std::vector<std::pair<int64_t, torch::Tensor>> parsed;
    parsed.reserve(S);
    for (size_t k = 0; k < S; k++) {
        const py::tuple t = segs[k].cast<py::tuple>();
        const int64_t off = t[0].cast<int64_t>();
        torch::Tensor src = t[1].cast<torch::Tensor>();
        if (!src.device().is_cpu() || src.scalar_type() != torch::kFloat32 || !src.is_contiguous() || src.requires_grad() ||
            off < 0 || off + src.numel() > n)
            return false;
        parsed.emplace_back(off, std::move(src));
    }
    float *m = mirror.data_ptr<float>();
    for (auto &pr : parsed) std::memcpy(m + pr.first, pr.second.data_ptr<float>(), (size_t)pr.second.numel() * sizeof(float));
    std::memcpy(slot.data_ptr<float>(), m, (size_t)n * sizeof(float));
    return true;
}

// hipStreamSynchronize of the device's current stream (what torch.cuda.current_stream(dev).synchronize() does, without the two
// Python-level objects in front of it: 2.3 -> 0.6 us in front of every waited-for step's return)
void sync_current_stream(int64_t device_index) {
    const auto s = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA((c10::DeviceIndex)device_index);
    py::gil_scoped_release nogil;
    s.synchronize();
}

std::tuple<int64_t, int64_t, int64_t> frame_stats(const torch::Tensor &geom) {
    c10::hip::HIPGuardMasqueradingAsCUDA guard(geom.device());
    GsrFrameStats s{};
    const int rc = gsr_frame_stats(geom.data_ptr(), &s, current_stream(geom.device()));
    TORCH_CHECK(rc == GSR_OK || rc == GSR_E_OVERFLOW, "libgsr_hip error ", rc, ": ", gsr_last_error());  // (a truncated frame: loud)
    return {s.num_visible, s.num_rendered, (int64_t)s.overflow};
}

// ---- upstream's three functions ------------------------------------------------------------------------------------
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians(const torch::Tensor &background, const torch::Tensor &means3D, const torch::Tensor &colors,
                    const torch::Tensor &opacity, const torch::Tensor &scales, const torch::Tensor &rotations,
                    double scale_modifier, const torch::Tensor &cov3D_precomp, const torch::Tensor &viewmatrix,
                    const torch::Tensor &projmatrix, double tan_fovx, double tan_fovy, int image_height,
                    int image_width, const torch::Tensor &sh, int degree, const torch::Tensor &campos, bool prefiltered,
                    bool antialiasing, bool debug, c10::optional<torch::Tensor> sh_rest, int param_space,
                    double near_plane, std::vector<int> tuning) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    require_gpu(means3D, "means3D");
    const auto dev = means3D.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int P = (int)means3D.size(0), H = image_height, W = image_width;
    auto fopt = means3D.options().dtype(torch::kFloat32);
    auto bopt = means3D.options().dtype(torch::kByte);
    // (a frame writes every pixel and every radius itself; upstream's zero fill is what P == 0 returns)
    auto iopt = means3D.options().dtype(torch::kInt32);
    torch::Tensor out_color = P != 0 ? torch::empty({3, H, W}, fopt) : torch::zeros({3, H, W}, fopt);
    torch::Tensor out_invdepth = P != 0 ? torch::empty({1, H, W}, fopt) : torch::zeros({1, H, W}, fopt);
    torch::Tensor radii = P != 0 ? torch::empty({P}, iopt) : torch::zeros({P}, iopt);
    torch::Tensor geom = torch::empty({0}, bopt), binning = torch::empty({0}, bopt), img = torch::empty({0}, bopt);
    int rendered = 0;
    if (P != 0) {
        int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
        torch::Tensor rest;
        if (sh_rest.has_value()) {
            TORCH_CHECK(sh.ndimension() == 3 && sh.size(1) == 1 && sh_rest->ndimension() == 3 &&
                            sh_rest->size(0) == sh.size(0),
                        "sh_rest needs sh = features_dc of shape (num_points, 1, 3)");
            rest = f32(*sh_rest, dev, "sh_rest");
            M = 1 + (int)rest.size(1);
        }
        const torch::Tensor bg = f32(background, dev, "background"), m3 = f32(means3D, dev, "means3D"),
                            col = f32(colors, dev, "colors"), op = f32(opacity, dev, "opacity"),
                            sc = f32(scales, dev, "scales"), rot = f32(rotations, dev, "rotations"),
                            cov = f32(cov3D_precomp, dev, "cov3D_precomp"), vm = f32(viewmatrix, dev, "viewmatrix"),
                            pm = f32(projmatrix, dev, "projmatrix"), shc = f32(sh, dev, "sh"),
                            cp = f32(campos, dev, "campos");
        auto stats = forward_frame(H, W, tan_fovx, tan_fovy, scale_modifier, degree, M, antialiasing, debug,
                                   near_plane, bg, m3, col, op, sc, rot, cov, vm, pm, shc,
                                   rest.defined() ? rest : torch::Tensor(torch::empty({0}, fopt)), cp, out_color,
                                   out_invdepth, radii, geom, binning, img, torch::empty({0}, bopt), 0, true,
                                   param_space, tuning, torch::empty({0}, fopt),
                                   torch::empty({0}, means3D.options().dtype(torch::kInt32)), torch::empty({0}, fopt),
                                   torch::empty({0}, bopt), torch::empty({0}, fopt),
                                   torch::empty({0}, means3D.options().dtype(torch::kInt32)), 0);
        (void)prefiltered;
        rendered = (int)std::get<1>(stats);
    }
    return {rendered, out_color, radii, geom, binning, img, out_invdepth};
}

std::vector<torch::Tensor> rasterize_gaussians_backward(
    const torch::Tensor &background, const torch::Tensor &means3D, const torch::Tensor &radii,
    const torch::Tensor &colors, const torch::Tensor &opacities, const torch::Tensor &scales,
    const torch::Tensor &rotations, double scale_modifier, const torch::Tensor &cov3D_precomp,
    const torch::Tensor &viewmatrix, const torch::Tensor &projmatrix, double tan_fovx, double tan_fovy,
    const torch::Tensor &dL_dout_color, c10::optional<torch::Tensor> dL_dout_invdepth, const torch::Tensor &sh,
    int degree, const torch::Tensor &campos, const torch::Tensor &geomBuffer, int64_t R,
    const torch::Tensor &binningBuffer, const torch::Tensor &imageBuffer, bool antialiasing, bool debug,
    c10::optional<torch::Tensor> sh_rest, int param_space, double near_plane) {
    const auto dev = means3D.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int64_t P = means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    int64_t M = sh.numel() != 0 ? sh.size(1) : 0;
    const bool split = sh_rest.has_value();
    if (split) {
        TORCH_CHECK(M == 1, "sh_rest needs sh = features_dc of shape (P,1,3)");
        M = 1 + sh_rest->size(1);
    }
    // One uninitialised arena sliced into the gradient buffers.  The SH gradients come first (16-byte aligned: the SH
    // backward kernel then writes every word of them itself, through whole lines); behind them, without gaps, the
    // buffers gsr_backward clears -- adjacent buffers cost ONE memset on the stream.
    auto pad4 = [](int64_t n) { return (n + 3) / 4 * 4; };
    const int64_t n_rot = 4 * P, n_sh = (split ? 1 : M) * 3 * P, n_rest = split ? (M - 1) * 3 * P : 0, n_cov = 6 * P,
                  n3 = 3 * P, n1 = P;
    auto arena = torch::empty({pad4(n_rest) + pad4(n_sh) + n_rot + n_cov + 4 * n3 + n1},
                              means3D.options().dtype(torch::kFloat32));
    int64_t off = 0;
    auto take = [&](int64_t n, std::vector<int64_t> shape, bool pad) {
        auto v = arena.narrow(0, off, n).view(shape);
        off += pad ? pad4(n) : n;
        return v;
    };
    auto dL_dsh_rest = take(n_rest, {P, split ? M - 1 : 0, 3}, true), dL_dsh = take(n_sh, {P, split ? 1 : M, 3}, true);
    auto dL_drot = take(n_rot, {P, 4}, false), dL_dcov = take(n_cov, {P, 6}, false), dL_dm3 = take(n3, {P, 3}, false),
         dL_dm2 = take(n3, {P, 3}, false), dL_dcol = take(n3, {P, 3}, false), dL_dsc = take(n3, {P, 3}, false),
         dL_dop = take(n1, {P, 1}, false);
    if (P != 0) {
        const torch::Tensor bg = f32(background, dev, "background"), m3 = f32(means3D, dev, "means3D"),
                            col = f32(colors, dev, "colors"), op = f32(opacities, dev, "opacities"),
                            sc = f32(scales, dev, "scales"), rot = f32(rotations, dev, "rotations"),
                            cov = f32(cov3D_precomp, dev, "cov3D_precomp"), vm = f32(viewmatrix, dev, "viewmatrix"),
                            pm = f32(projmatrix, dev, "projmatrix"), shc = f32(sh, dev, "sh"),
                            cp = f32(campos, dev, "campos"), dLc = f32(dL_dout_color, dev, "dL_dout_color");
        torch::Tensor rest, dLd;
        if (split) rest = f32(*sh_rest, dev, "sh_rest");
        if (dL_dout_invdepth.has_value() && dL_dout_invdepth->numel() != 0)
            dLd = f32(*dL_dout_invdepth, dev, "dL_dout_invdepth");
        GsrSettings st = settings(H, W, (float)tan_fovx, (float)tan_fovy, (float)scale_modifier, degree, (int)M, false,
                                  antialiasing, debug, (float)near_plane, Tuning{});
        GsrInputs in{};
        in.P = (int32_t)P;
        in.background = fptr(bg);
        in.means3D = fptr(m3);
        in.shs = fptr(shc);
        in.colors_precomp = fptr(col);
        in.opacities = fptr(op);
        in.scales = fptr(sc);
        in.rotations = fptr(rot);
        in.cov3D_precomp = fptr(cov);
        in.viewmatrix = fptr(vm);
        in.projmatrix = fptr(pm);
        in.campos = fptr(cp);
        in.shs_rest = split ? fptr(rest) : nullptr;
        in.param_space = param_space;
        GsrBackwardInputs bw{fptr(dLc), dLd.defined() ? fptr(dLd) : nullptr, radii.data_ptr<int32_t>(), R,
                             geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr()};
        auto mp = [](torch::Tensor &t) { return t.numel() ? t.data_ptr<float>() : nullptr; };
        GsrGrads gr{mp(dL_dm2), mp(dL_dcol), mp(dL_dop), mp(dL_dm3), mp(dL_dcov), mp(dL_dsh), mp(dL_dsc), mp(dL_drot),
                    nullptr, nullptr, split ? mp(dL_dsh_rest) : nullptr};
        const int rc = gsr_backward(&st, &in, &bw, &gr, current_stream(dev));
        TORCH_CHECK(rc == GSR_OK, "libgsr_hip error ", rc, ": ", gsr_last_error());
    }
    std::vector<torch::Tensor> out{dL_dm2, dL_dcol, dL_dop, dL_dm3, dL_dcov, dL_dsh, dL_dsc, dL_drot};
    if (split) out.push_back(dL_dsh_rest);
    return out;
}

torch::Tensor mark_visible(const torch::Tensor &means3D, const torch::Tensor &viewmatrix,
                           const torch::Tensor &projmatrix, double near_plane) {
    (void)projmatrix;  // unused upstream as well
    require_gpu(means3D, "means3D");
    const auto dev = means3D.device();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const int64_t P = means3D.size(0);
    auto present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        const torch::Tensor m3 = f32(means3D, dev, "means3D"), vm = f32(viewmatrix, dev, "viewmatrix");
        const int rc = gsr_mark_visible((int32_t)P, fptr(m3), fptr(vm), (float)near_plane,
                                        reinterpret_cast<uint8_t *>(present.data_ptr()), current_stream(dev));
        TORCH_CHECK(rc == GSR_OK, "libgsr_hip error ", rc, ": ", gsr_last_error());
    }
    return present;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled binding of libgsr_hip.so (MI355X 3DGS rasterizer) with upstream's _C signatures";
    {
        // ABI handshake: this file was compiled against include/gsr.h; the library it is linked to may have been
        // rebuilt from another revision since (struct layouts would then disagree silently).  Importing fails instead,
        // and gsworld_amd/_C.py falls back to the ctypes binding, which performs the same check.
        int32_t sizes[6] = {0, 0, 0, 0, 0, 0};
        gsr_abi_sizes(sizes);
        if (sizes[0] != (int32_t)sizeof(GsrSettings) || sizes[1] != (int32_t)sizeof(GsrInputs) ||
            sizes[2] != (int32_t)sizeof(GsrOutputs) || sizes[3] != (int32_t)sizeof(GsrBuffers) ||
            sizes[4] != (int32_t)sizeof(GsrBackwardInputs) || sizes[5] != (int32_t)sizeof(GsrGrads))
            throw py::import_error("gsworld_amd._C_ext was built against another revision of include/gsr.h than "
                                   "libgsr_hip.so: run gsworld_amd/build_ext.py again");
    }
    m.def("rasterize_gaussians", &rasterize_gaussians, py::arg("background"), py::arg("means3D"), py::arg("colors"),
          py::arg("opacity"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"),
          py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"),
          py::arg("tan_fovy"), py::arg("image_height"), py::arg("image_width"), py::arg("sh"), py::arg("degree"),
          py::arg("campos"), py::arg("prefiltered"), py::arg("antialiasing"), py::arg("debug"),
          py::arg("sh_rest") = py::none(), py::arg("param_space") = 0, py::arg("near_plane") = GSR_NEAR_PLANE,
          py::arg("tuning") = std::vector<int>{});
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward, py::arg("background"), py::arg("means3D"),
          py::arg("radii"), py::arg("colors"), py::arg("opacities"), py::arg("scales"), py::arg("rotations"),
          py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"),
          py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("dL_dout_color"), py::arg("dL_dout_invdepth"),
          py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("geomBuffer"), py::arg("R"),
          py::arg("binningBuffer"), py::arg("imageBuffer"), py::arg("antialiasing"), py::arg("debug"),
          py::arg("sh_rest") = py::none(), py::arg("param_space") = 0, py::arg("near_plane") = GSR_NEAR_PLANE);
    m.def("mark_visible", &mark_visible, py::arg("means3D"), py::arg("viewmatrix"), py::arg("projmatrix"),
          py::arg("near_plane") = GSR_NEAR_PLANE);
    m.def("forward_frame", &forward_frame);
    m.def("forward_batch", &forward_batch);
    m.def("stage_host_values", &stage_host_values);
    m.def("sync_current_stream", &sync_current_stream);
    py::class_<StepPack>(m, "StepPack")
        .def(py::init<const py::list &>())
        .def("run", &StepPack::run)
        .def("run_staged", &StepPack::run_staged);
    m.def("frame_stats", &frame_stats);
    m.def("version", []() { return std::string(gsr_version()); });
}
