// transform.hip -- fused per-step rigid transform of labelled Gaussians.
//
// What GSWorld does per step and per camera-frame (/root/reference/gsworld/mani_skill/utils/wrappers/
// gs_world_wrapper.py:110-162 and :244-265): deep-copy the whole model (352 MB at 1.47 M Gaussians), then for each
// of ~18 movable parts build an isin() mask over all N labels, gather, run transform_gaussians
// (gsworld/utils/gs_utils.py:283-385: scale -> rotate -> translate), and scatter xyz / rotation back through the
// same mask.  Here: ONE pass over N.  label -> transform index through a small LUT, then
//   xyz' = R (s * xyz) + t,   rot' = standardize(q_R (x) rot/|rot|) * |rot|
// written straight into the buffers the rasterizer reads; untouched labels are copied through.
// HBM traffic: 32 B read + 28 B written per Gaussian (88 MB at 1.47 M) -- a pure stream.
#include "gsr_internal.h"

namespace {

constexpr int kXf = 17;  // floats per transform: R (row-major 9), t (3), scale (1), q_R (w,x,y,z)

__global__ __launch_bounds__(GSR_BLOCK) void transform_kernel(int P, const float *__restrict__ xyz,
                                                              const float *__restrict__ rot,
                                                              const float *__restrict__ semantics,
                                                              const int32_t *__restrict__ lut, int lut_size,
                                                              const float *__restrict__ xf, int K,
                                                              float *__restrict__ xyz_out, float *__restrict__ rot_out) {
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    if (i >= P) return;
    float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    float4 q = *reinterpret_cast<const float4 *>(rot + 4 * (size_t)i);
    const int label = (int)semantics[i];  // the reference compares labels after .long() (truncation)
    const int k = (label >= 0 && label < lut_size) ? lut[label] : -1;
    if (k >= 0 && k < K) {
        const float *T = xf + (size_t)k * kXf;
        const float s = T[12];
        px *= s; py *= s; pz *= s;
        const float rx = T[0] * px + T[1] * py + T[2] * pz + T[9];
        const float ry = T[3] * px + T[4] * py + T[5] * pz + T[10];
        const float rz = T[6] * px + T[7] * py + T[8] * pz + T[11];
        px = rx; py = ry; pz = rz;
        // Gaussian quaternion: rotate the normalised quaternion, keep its norm (gs_utils.py:242-249)
        const float norm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        const float bw = q.x / norm, bx = q.y / norm, by = q.z / norm, bz = q.w / norm;
        const float aw = T[13], ax = T[14], ay = T[15], az = T[16];
        float ow = aw * bw - ax * bx - ay * by - az * bz;
        float ox = aw * bx + ax * bw + ay * bz - az * by;
        float oy = aw * by - ax * bz + ay * bw + az * bx;
        float oz = aw * bz + ax * by - ay * bx + az * bw;
        if (ow < 0.f) { ow = -ow; ox = -ox; oy = -oy; oz = -oz; }  // quaternion_multiply standardises the sign
        q = make_float4(ow * norm, ox * norm, oy * norm, oz * norm);
    }
    xyz_out[3 * (size_t)i] = px;
    xyz_out[3 * (size_t)i + 1] = py;
    xyz_out[3 * (size_t)i + 2] = pz;
    *reinterpret_cast<float4 *>(rot_out + 4 * (size_t)i) = q;
}

}  // namespace

extern "C" int gsr_transform_gaussians(int32_t P, const float *xyz, const float *rot, const float *semantics,
                                       const int32_t *lut, int32_t lut_size, const float *transforms, int32_t K,
                                       float *xyz_out, float *rot_out, void *stream) {
    if (P < 0 || K < 0 || lut_size < 0) {
        gsr_set_error("gsr_transform_gaussians: negative size");
        return GSR_E_INVALID;
    }
    if (P == 0) return GSR_OK;
    if (!xyz || !rot || !semantics || !xyz_out || !rot_out || (lut_size > 0 && !lut) || (K > 0 && !transforms)) {
        gsr_set_error("gsr_transform_gaussians: null pointer");
        return GSR_E_INVALID;
    }
    if ((reinterpret_cast<uintptr_t>(rot) | reinterpret_cast<uintptr_t>(rot_out)) & 15u) {
        gsr_set_error("gsr_transform_gaussians: rotation buffers must be 16-byte aligned");
        return GSR_E_INVALID;
    }
    hipLaunchKernelGGL(transform_kernel, dim3(gsr_div_up(P, GSR_BLOCK)), dim3(GSR_BLOCK), 0, (hipStream_t)stream, P, xyz,
                       rot, semantics, lut, lut_size, transforms, K, xyz_out, rot_out);
    return gsr_check_launch("transform_gaussians", false, (hipStream_t)stream);
}
