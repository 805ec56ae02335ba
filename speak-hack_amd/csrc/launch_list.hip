// spk_launch_list: enqueue a whole pre-built sequence of launches with ONE call across the C boundary.
//
// A decoder forward is ~25 launches of 10-700 us each; issued one by one from the reference's host language every launch
// also pays a descriptor build, a config / workspace query and a ctypes crossing (~50 us of Python each, 5.7 ms a step
// against 3.8 ms of GPU work).  The host side therefore builds the descriptors ONCE per (module, batch, device) -- packed
// weights, tile configs, split-K workspace, intermediate buffers -- and hands the array over per call: the same kernels
// with the same arguments as the one-by-one path, at ~4 us of hipLaunchKernel each.  Nothing is allocated or synchronised
// here, so a list may also be captured into a hipGraph.
#include "spk_common.hpp"

extern "C" int spk_launch_list(const spk_op* ops, int n_ops, uint32_t kind_mask, void* stream) {
    SPK_REQUIRE(ops && n_ops >= 0, "launch_list: bad arguments");
    for (int i = 0; i < n_ops; ++i) {
        const spk_op& op = ops[i];
        SPK_REQUIRE(op.kind > 0 && op.kind < 32 && op.desc, "launch_list: op %d: bad kind %d or null descriptor", i, op.kind);
        if (!(kind_mask & (1u << op.kind))) continue;
        int rc = SPK_EUNSUPPORTED;
        switch (op.kind) {
            case SPK_OP_CONV2D:
                rc = spk_conv2d_fwd(static_cast<const spk_conv2d_desc*>(op.desc), stream);
                break;
            case SPK_OP_FC: {
                const spk_fc_args* a = static_cast<const spk_fc_args*>(op.desc);
                rc = spk_fc_fwd(a->x, a->x_stride, a->w, a->bias, a->out, a->out_stride, a->B, a->I, a->O, a->wmul, a->bmul, a->slope, stream);
                break;
            }
            case SPK_OP_FC_GROUPED: {
                const spk_fc_grouped_args* a = static_cast<const spk_fc_grouped_args*>(op.desc);
                rc = spk_fc_grouped_fwd(a->groups, a->n_groups, a->B, stream);
                break;
            }
            case SPK_OP_BIAS_NOISE_STYLE: {
                const spk_bias_noise_style_args* a = static_cast<const spk_bias_noise_style_args*>(op.desc);
                rc = spk_bias_noise_style_fwd(a->x, a->x_batch_stride, a->bias, a->noise_w, a->noise, a->style, a->style_stride, a->y,
                                              a->B, a->C, a->HW, stream);
                break;
            }
            case SPK_OP_TORGB: {
                const spk_torgb_args* a = static_cast<const spk_torgb_args*>(op.desc);
                if (a->mod) rc = spk_torgb_mod_skip_fwd(a->x, a->w, a->mod, a->bias, a->skip, a->y, a->B, a->C, a->O, a->H, a->W, a->in_scale, stream);
                else if (a->skip) rc = spk::fail(SPK_EINVAL, "launch_list: op %d: a skip image needs a modulated toRGB", i);
                else rc = spk_conv1x1_small_fwd(a->x, a->w, a->bias, a->y, a->B, a->C, a->O, (int64_t)a->H * a->W, a->in_scale, stream);
                break;
            }
            case SPK_OP_DEMOD_GROUPED: {
                const spk_demod_grouped_args* a = static_cast<const spk_demod_grouped_args*>(op.desc);
                rc = spk_modconv_demod_grouped(a->groups, a->n_groups, a->B, a->eps, stream);
                break;
            }
            case SPK_OP_PIXELNORM: {
                const spk_pixelnorm_args* a = static_cast<const spk_pixelnorm_args*>(op.desc);
                rc = spk_pixelnorm_fwd(a->x, a->y, a->B, a->C, a->HW, a->eps, a->sqrt_form, stream);
                break;
            }
            case SPK_OP_UPSAMPLE2X: {
                const spk_upsample2x_args* a = static_cast<const spk_upsample2x_args*>(op.desc);
                rc = spk_upsample2x_fwd(a->x, a->y, a->planes, a->Hin, a->Win, a->zero_border, stream);
                break;
            }
            default:
                return spk::fail(SPK_EUNSUPPORTED, "launch_list: op %d: unknown kind %d", i, op.kind);
        }
        if (rc != SPK_OK) return rc;      // spk_last_error() holds the failing op's own message
    }
    return SPK_OK;
}
