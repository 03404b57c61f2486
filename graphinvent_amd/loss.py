"""
The training loss of the reference, ``Workflow.loss`` (Workflow.py:833-860):
``KLDivLoss(reduction="batchmean")(log_softmax(output, dim=1), target / target.sum(1, keepdim=True))``.

``apd_kl_loss`` dispatches on the device: CUDA logits go through ONE fused HIP kernel
(``gi_kl_loss``: log-softmax, target normalisation, KL and the gradient w.r.t. the logits in a
single pass over [B, APD] — SURVEY.md §8f row 2) instead of ~10 small torch kernels; anything else
uses the plain torch expression (``apd_kl_loss_torch``), which is also the reference the fused
kernel is tested against.  Rows whose target is all zero give NaN exactly as in the reference
(DataProcesser.py:268-269 padding rows); callers slice them off.
"""
import torch


def apd_kl_loss_torch(output: torch.Tensor, target_output: torch.Tensor) -> torch.Tensor:
    log_p = torch.log_softmax(output, dim=1)
    target = target_output / torch.sum(target_output, dim=1, keepdim=True)
    return torch.nn.functional.kl_div(log_p, target, reduction="batchmean")


class _FusedKL(torch.autograd.Function):
    """Two launches forward (row kernel + fixed-order batch mean), one backward (d_out scaled in place
    by the upstream scalar, read on the device) — no torch glue kernels on the step's critical path.
    Single backward only: the gradient buffer is consumed (a second `backward(retain_graph=True)` raises
    instead of returning a doubly scaled gradient)."""

    @staticmethod
    def forward(ctx, output, target):
        from . import lib as L
        lib = L.load()
        out = output.contiguous()
        if target.dtype == torch.int8:                       # HDF dtype, read directly
            tgt, tdt = target.contiguous(), L.DTYPE_I8
        else:
            tgt, tdt = target.contiguous().float(), L.DTYPE_F32
        B, W = out.shape
        if B == 0:               # the torch expression divides 0 by the batch size 0: NaN, zero-size gradient
            ctx.d_out = torch.empty_like(out) if ctx.needs_input_grad[0] else None
            return torch.full((), float("nan"), dtype=torch.float32, device=out.device)
        buf = torch.empty(B + 1, dtype=torch.float32, device=out.device)    # row losses | mean
        need_grad = ctx.needs_input_grad[0]
        d_out = torch.empty_like(out) if need_grad else None
        L.check(lib.gi_kl_loss(out.data_ptr(), out.stride(0), tgt.data_ptr(), tdt, tgt.stride(0), B, W,
                               buf.data_ptr(), d_out.data_ptr() if need_grad else None,
                               d_out.stride(0) if need_grad else 0, buf.data_ptr() + 4 * B,
                               torch.cuda.current_stream(out.device).cuda_stream), "gi_kl_loss")
        ctx.d_out = d_out
        return buf[B]

    @staticmethod
    def backward(ctx, grad):
        from . import lib as L
        d_out, ctx.d_out = ctx.d_out, None
        if d_out is None:
            raise RuntimeError("fused KL loss: backward called twice (its gradient buffer is "
                               "scaled in place)")
        if d_out.numel() == 0:
            return d_out, None
        if grad.data_ptr() in _UNIT_GRADS:     # a registered constant 1.0: scaling would change nothing
            if __debug__ and CHECK_UNIT_GRADS and float(grad) != 1.0:
                raise RuntimeError("fused KL loss: a tensor registered with register_unit_gradient no "
                                   "longer holds 1.0")
            return d_out, None
        g = grad.contiguous().float()
        L.check(L.load().gi_scale_by_scalar(d_out.data_ptr(), d_out.numel(), g.data_ptr(),
                                            torch.cuda.current_stream(d_out.device).cuda_stream),
                "gi_scale_by_scalar")
        return d_out, None


#: data pointers of device scalars that are known to hold exactly 1.0 for their whole life (kept alive
#: and never written by whoever registers them: ``dp.DataParallel``'s cached root gradient).  For these the
#: in-place scaling of the loss gradient — one launch per step — is skipped; the result is bit-identical.
_UNIT_GRADS = set()
#: set True to read every registered scalar back in backward and fail if it is not 1.0 (a host sync per
#: step — for debugging a training loop that might write to its cached root gradient)
CHECK_UNIT_GRADS = False


def register_unit_gradient(t: torch.Tensor) -> torch.Tensor:
    """Declare `t` (a scalar tensor the caller keeps alive and never modifies) to be the constant 1.0."""
    import weakref
    ptr = t.data_ptr()
    _UNIT_GRADS.add(ptr)
    weakref.finalize(t, _UNIT_GRADS.discard, ptr)      # the address may be recycled once `t` is gone
    return t


def apd_kl_loss(output: torch.Tensor, target_output: torch.Tensor) -> torch.Tensor:
    if output.is_cuda and output.dtype == torch.float32 and output.dim() == 2:
        return _FusedKL.apply(output, target_output)
    return apd_kl_loss_torch(output, target_output)
