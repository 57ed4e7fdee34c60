"""link_amd/modules.py -- the row-wise module wrappers the reference networks take from `torchsparse.nn`
(torchsparse/nn/modules/{norm,activation}.py, nn/utils/apply.py): a torch module applied to the feature
rows of a SparseTensor, coordinates / stride / cmaps / kmaps carried over by reference."""
from __future__ import annotations

from typing import Callable

import torch
from torch import nn

from .tensor import SparseTensor

__all__ = ["fapply", "BatchNorm", "ReLU", "LeakyReLU", "fuse_for_inference"]


def fapply(input: SparseTensor, fn: Callable[..., torch.Tensor], *args, **kwargs) -> SparseTensor:
    out = SparseTensor(fn(input.feats, *args, **kwargs), input.coords, input.stride)
    out.cmaps, out.kmaps = input.cmaps, input.kmaps
    return out


class _BatchNormTrain(torch.autograd.Function):
    """Training-mode BatchNorm over feature rows with the statistics on the HIP column-reduction kernels
    (include/link_amd.h section F): y = (x - mean) * scale + shift with batch statistics, running statistics updated in
    place; backward: grad_x = a * g + bq * (x - mean) + cq per channel, grad_weight = sum g * xhat, grad_bias = sum g."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu=False):
        """relu: the ReLU that follows the BatchNorm in the reference's blocks (linkunet.py:23-38) applied in the same pass,
        its gradient mask recomputed in the backward's two passes (round 5: one launch for the normalisation, one for the input
        gradient, instead of torch's x - mean / addcmul / clamp and threshold_backward / x - mean / addcmul / addcmul_)."""
        from . import _lib as L
        n, c = x.shape
        x = x.contiguous()
        lib, st = L.lib(), L.current_stream_handle()
        partial = torch.empty(int(lib.link_bn_partial_workgroups(n, c)) * 2 * c, dtype=torch.float64, device=x.device)
        vec = torch.empty((4, c), dtype=torch.float32, device=x.device)          # mean | invstd | scale | shift
        w = weight.detach().contiguous() if weight is not None else None
        b = bias.detach().contiguous() if bias is not None else None
        L.check(lib.link_bn_forward_stats(x.data_ptr(), n, c, float(eps), float(momentum), partial.data_ptr(), vec[0].data_ptr(),
                                          vec[1].data_ptr(), running_mean.data_ptr() if running_mean is not None else None,
                                          running_var.data_ptr() if running_var is not None else None,
                                          w.data_ptr() if w is not None else None, b.data_ptr() if b is not None else None,
                                          vec[2].data_ptr(), vec[3].data_ptr(), st), "link_bn_forward_stats")
        ctx.save_for_backward(x, vec, *((w,) if w is not None else ()))
        ctx.partial = partial
        ctx.has_wb = (weight is not None, bias is not None)
        ctx.relu = bool(relu)
        y = torch.empty_like(x)                                   # y = (x - mean) * scale + shift, centred first: no cancellation
        L.check(lib.link_bn_apply_forward(x.data_ptr(), vec[0].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), n, c,
                                          1 if relu else 0, y.data_ptr(), st), "link_bn_apply_forward")
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        x, vec = ctx.saved_tensors[:2]
        w = ctx.saved_tensors[2] if ctx.has_wb[0] else None
        n, c = x.shape
        g = g.contiguous()
        lib, st = L.lib(), L.current_stream_handle()
        out = torch.empty((5, c), dtype=torch.float32, device=x.device)           # sum_g | sum_gx | a | bq | cq
        g = g.float()
        if ctx.relu:                                          # g masked by y > 0 inside both passes (y recomputed from x)
            L.check(lib.link_bn_backward_reduce_relu(g.data_ptr(), x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                                     vec[3].data_ptr(), n, c, ctx.partial.data_ptr(), out[0].data_ptr(),
                                                     out[1].data_ptr(), w.data_ptr() if w is not None else None, out[2].data_ptr(), st),
                    "link_bn_backward_reduce_relu")
        else:
            L.check(lib.link_bn_backward_reduce(g.data_ptr(), x.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), n, c,
                                                ctx.partial.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                                w.data_ptr() if w is not None else None, out[2].data_ptr(), st),
                    "link_bn_backward_reduce")
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            L.check(lib.link_bn_apply_backward(g.data_ptr(), x.data_ptr(), vec[0].data_ptr(), out[2].data_ptr(),
                                               vec[2].data_ptr() if ctx.relu else None, vec[3].data_ptr() if ctx.relu else None,
                                               n, c, gx.data_ptr(), st), "link_bn_apply_backward")
        return gx, (out[1] if ctx.has_wb[0] else None), (out[0] if ctx.has_wb[1] else None), None, None, None, None, None


class BatchNorm(nn.BatchNorm1d):
    """torchsparse/nn/modules/norm.py:10-13.  Training mode on GPU rows runs the batch statistics through the HIP
    column reductions (_BatchNormTrain); everything else (eval, CPU, half rows, momentum=None) is nn.BatchNorm1d."""

    hip_stats = True

    def _hip_train_ok(self, x: torch.Tensor) -> bool:
        return (self.hip_stats and self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 1
                and x.shape[1] % 4 == 0 and 4 <= x.shape[1] <= 1024 and self.momentum is not None
                and (self.weight is None or self.weight.dtype == torch.float32))

    def forward(self, input: SparseTensor, relu: bool = False) -> SparseTensor:
        """relu=True (used by the fused containers of fuse_for_inference in TRAINING mode): the ReLU module that follows this
        BatchNorm runs inside its passes -- same values as BatchNorm then ReLU."""
        x = input.feats
        if self._hip_train_ok(x):
            rm, rv = (self.running_mean, self.running_var) if self.track_running_stats else (None, None)
            if self.track_running_stats and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            return fapply(input, lambda f: _BatchNormTrain.apply(f, self.weight, self.bias, rm, rv, self.momentum, self.eps, relu))
        out = fapply(input, super().forward)
        return fapply(out, torch.relu) if relu else out


class ReLU(nn.ReLU):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class _FusedSequential(nn.Sequential):
    """nn.Sequential whose [Conv3d, BatchNorm(, ReLU)] runs execute as one launch at inference (see
    fuse_for_inference).  Same children, same state_dict keys; with grad enabled or in training mode it is the
    plain nn.Sequential (BatchNorm then needs batch statistics and the autograd path)."""

    def forward(self, input):
        mods = list(self)
        if torch.is_grad_enabled() or self.training or not isinstance(input, SparseTensor) or not input.F.is_cuda:
            # training / autograd: the modules one by one -- except that a ReLU straight after a BatchNorm in training mode runs
            # inside the BatchNorm's passes (same values; hooks on either module keep the plain path)
            i = 0
            while i < len(mods):
                m = mods[i]
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if (isinstance(m, BatchNorm) and m.training and type(nxt) in (ReLU, nn.ReLU) and isinstance(input, SparseTensor)
                        and m._hip_train_ok(input.feats) and not (m._forward_hooks or m._forward_pre_hooks or nxt._forward_hooks
                                                                   or nxt._forward_pre_hooks)):
                    input = m(input, relu=True)
                    i += 2
                    continue
                input = m(input)
                i += 1
            return input
        from .elk import fold_batchnorm
        i = 0
        while i < len(mods):
            g = self._link_groups.get(i)
            if g is None:
                input = mods[i](input)
                i += 1
                continue
            conv, bn, relu, nxt = mods[i], mods[g[0]], g[1], g[2]
            sc, sh = fold_batchnorm(bn, conv.bias)
            input = conv.forward_affine(input, sc, sh, relu)
            i = nxt
        return input


def fuse_for_inference(model: nn.Module) -> nn.Module:
    """Rewrite, in place, every nn.Sequential of `model` so that runs of link_amd.Conv3d -> BatchNorm
    (-> ReLU) execute as ONE kernel launch at inference: the BatchNorm's running statistics and affine (and the
    convolution's bias) are folded into a per-channel scale / shift applied in the convolution's finish phase
    (include/link_amd.h: flag bit 1 of link_subm_conv_ln_add_relu / link_conv_centre_sum / link_conv_pairs_sum).
    That is the shape of the reference's BasicConvolutionBlock / ResidualBlock.net / stageK_tail / elkK_tail
    (linkunet.py:18-92,217-224), so it applies to the reference's own network classes built on the aliased
    surface.  Children, parameter names and state_dict keys are untouched (only the container's class changes);
    training mode and autograd run the modules one by one as before.  Returns `model`."""
    from .elk import Conv3d, can_fold_batchnorm
    for mod in list(model.modules()):
        if not isinstance(mod, nn.Sequential) or isinstance(mod, _FusedSequential):
            continue
        kids = list(mod)
        groups, i = {}, 0
        while i < len(kids):
            if isinstance(kids[i], Conv3d) and i + 1 < len(kids) and can_fold_batchnorm(kids[i + 1]):
                relu = i + 2 < len(kids) and isinstance(kids[i + 2], nn.ReLU)
                groups[i] = (i + 1, relu, i + (3 if relu else 2))
                i += 3 if relu else 2
            else:
                i += 1
        if groups and type(mod) is nn.Sequential:
            mod.__class__ = _FusedSequential
            mod._link_groups = groups
    return model
