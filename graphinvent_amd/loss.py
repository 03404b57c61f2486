"""
The training loss of the reference, ``Workflow.loss`` (Workflow.py:833-860):
``KLDivLoss(reduction="batchmean")(log_softmax(output, dim=1), target / target.sum(1, keepdim=True))``.

Runs on the device the logits live on (a handful of small torch kernels over [B, APD]; the fused
HIP version is SURVEY.md §8f row 2).  Rows whose target is all zero give 0/0 = NaN exactly as in
the reference (DataProcesser.py:268-269 padding rows); callers slice them off.
"""
import torch


def apd_kl_loss(output: torch.Tensor, target_output: torch.Tensor) -> torch.Tensor:
    log_p = torch.log_softmax(output, dim=1)
    target = target_output / torch.sum(target_output, dim=1, keepdim=True)
    return torch.nn.functional.kl_div(log_p, target, reduction="batchmean")
