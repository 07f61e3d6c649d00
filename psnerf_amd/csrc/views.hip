// One training batch of stage 2 gathered ON THE DEVICE from view tables that stay resident in HBM.
//
// What it replaces: stage2/datasets/dataset.py:137-199 (__getitem__: the per-item light subset and pixel subset of a view --
// rgb = imgs[view][lidx] * object_mask, then every per-pixel tensor indexed with sampling_idx), the collation / un-batching of
// stage2/trainer.py:364-367 and the vis_plus selection of stage2/trainer.py:384-392 (vis_train_gt = vis_plus_v[sidx][:, sampling_idx]),
// which the reference evaluates on the host for every iteration and then uploads.  Here only the DRAWN INDICES travel to the device
// (a few hundred KB); the images (uint8 / uint16 as decoded, or float32), masks, surface points, normals and visibility maps of every
// view live in HBM (20 views x 96 lights x 612 x 512 x 3 uint8 = 1.8 GB of 288 GB), and each data-parallel rank gathers only ITS
// slice of the drawn pixel list.
//
// HBM-bound index work: one launch, every output element written once with coalesced stores (thread <-> output float), the reads
// are scattered 3- to 12-byte pieces.  Per 32768-pixel x 96-light batch: 37.7 MB of rgb + 12.6 MB of visibility written.
#include "common.h"

namespace psn {

// Jobs along blockIdx.y: [0, L) rgb row l | L: per-pixel attributes | (L, L + Lv] visibility row (Lv = L when a visibility output is
// requested, else 0) | (L + Lv, L + Lv + V] vis_train_gt row.
__global__ __launch_bounds__(256) void view_batch_kernel(const PsnViewBatch b) {
    __shared__ float lut_s[256];
    const int job = blockIdx.y;
    const int L = b.n_lights;
    const int64_t n = b.n;
    const int64_t base = (int64_t)blockIdx.x * 256;
    const int t = threadIdx.x;
    if (job < L) {
        // rgb [L, n, 3] = image[lidx[l], pix, :] * object_mask[pix]   (dataset.py:172 and :122: both multiplications are by 0 / 1)
        if (b.image_type == 1) {
            lut_s[t] = b.lut[t];
            __syncthreads();
        }
        const int64_t row = b.lidx[job];
        float* __restrict__ out = b.rgb + ((int64_t)job * n + base) * 3;
        const int64_t left = n - base;
        const int cnt = (int)(left < 256 ? left : 256) * 3;
        for (int e = t; e < cnt; e += 256) {
            const int p = e / 3, c = e - 3 * p;
            const int64_t px = b.pix != nullptr ? b.pix[base + p] : b.pix0 + base + p;
            const int64_t src = (row * b.hw + px) * 3 + c;
            float v;
            if (b.image_type == 0) v = ((const float*)b.images)[src];
            else if (b.image_type == 1) v = lut_s[((const unsigned char*)b.images)[src]];
            else v = b.lut[((const unsigned short*)b.images)[src]];
            out[e] = v * (b.object_mask[px] != 0 ? 1.0f : 0.0f);
        }
        return;
    }
    const int64_t i = base + t;
    if (job == L && b.light_direction_out != nullptr)  // light_direction[view][lidx] (dataset.py:166): L rows of three floats
        for (int64_t k = i; k < 3 * (int64_t)L; k += (int64_t)gridDim.x * 256) b.light_direction_out[k] = b.light_direction[3 * b.lidx[k / 3] + k % 3];
    if (i >= n) return;
    const int64_t px = b.pix != nullptr ? b.pix[i] : b.pix0 + i;
    if (job == L) {
        if (b.object_mask_out != nullptr) b.object_mask_out[i] = b.object_mask[px];
        if (b.surface_mask_out != nullptr) b.surface_mask_out[i] = b.surface_mask[px];
        if (b.uv != nullptr) {  // (x, y) = (column, row) of the pixel, as floats: the flipped np.mgrid of dataset.py:138-140
            const int64_t y = px / b.width;
            b.uv[2 * i] = (float)(px - y * b.width);
            b.uv[2 * i + 1] = (float)y;
        }
        if (b.points_out != nullptr)
            for (int c = 0; c < 3; ++c) b.points_out[3 * i + c] = b.points[3 * px + c];
        if (b.normal_out != nullptr)
            for (int c = 0; c < 3; ++c) b.normal_out[3 * i + c] = b.normal[3 * px + c];
        if (b.sampling_idx_out != nullptr) b.sampling_idx_out[i] = px;
        return;
    }
    const int Lv = b.visibility_out != nullptr ? L : 0;
    if (job <= L + Lv) {
        const int l = job - L - 1;
        b.visibility_out[(int64_t)l * n + i] = b.visibility[b.lidx[l] * b.hw + px];
        return;
    }
    const int v = job - L - Lv - 1;
    b.vis_train_gt[(int64_t)v * n + i] = b.vis_plus[b.vidx[v] * b.hw + px];
}

}  // namespace psn

extern "C" int psn_view_batch(const PsnViewBatch* b, void* stream) {
    PSN_CHECK_ARG(b != nullptr, "view_batch: null descriptor");
    PSN_CHECK_ARG(b->n >= 0 && b->n < (1ll << 31) && b->hw > 0 && b->width > 0, "view_batch: bad sizes (n %lld, hw %lld, width %d)",
                  (long long)b->n, (long long)b->hw, b->width);
    PSN_CHECK_ARG(b->n_lights >= 0 && b->n_vis >= 0 && 2 * b->n_lights + b->n_vis + 1 <= 65535, "view_batch: too many light rows");
    PSN_CHECK_ARG(b->image_type >= 0 && b->image_type <= 2, "view_batch: image_type must be 0 (float32), 1 (uint8) or 2 (uint16)");
    PSN_CHECK_ARG(b->n_lights == 0 || (b->images != nullptr && b->rgb != nullptr && b->lidx != nullptr), "view_batch: images / rgb / lidx missing");
    PSN_CHECK_ARG(b->image_type == 0 || b->lut != nullptr, "view_batch: integer images need the value table");
    PSN_CHECK_ARG(b->object_mask != nullptr, "view_batch: object_mask missing");
    PSN_CHECK_ARG((b->surface_mask_out == nullptr || b->surface_mask != nullptr) && (b->points_out == nullptr || b->points != nullptr) &&
                      (b->normal_out == nullptr || b->normal != nullptr),
                  "view_batch: an output is requested whose view table is missing");
    PSN_CHECK_ARG(b->visibility_out == nullptr || (b->visibility != nullptr && b->lidx != nullptr), "view_batch: visibility table missing");
    PSN_CHECK_ARG(b->light_direction_out == nullptr || (b->light_direction != nullptr && b->lidx != nullptr), "view_batch: light_direction table missing");
    PSN_CHECK_ARG(b->n_vis == 0 || (b->vis_plus != nullptr && b->vidx != nullptr && b->vis_train_gt != nullptr), "view_batch: vis_plus operands missing");
    PSN_CHECK_ARG(b->pix != nullptr || (b->pix0 >= 0 && b->pix0 + b->n <= b->hw), "view_batch: identity pixel range outside the view");
    if (b->n == 0) return PSN_OK;
    const int jobs = b->n_lights + 1 + (b->visibility_out != nullptr ? b->n_lights : 0) + b->n_vis;
    hipLaunchKernelGGL(psn::view_batch_kernel, dim3((unsigned)((b->n + 255) / 256), (unsigned)jobs), dim3(256), 0, (hipStream_t)stream, *b);
    PSN_CHECK_LAUNCH("view_batch");
    return PSN_OK;
}
