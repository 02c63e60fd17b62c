"""Loss modules with the reference's names and return conventions, computed on LOGITS by HIP kernels.

Mirrors joeynmt/loss.py: XentLoss (:16-107) and XentCTCLoss (:110-177).  The reference applies
F.log_softmax in Model.forward (model.py:121,126) and then KLDivLoss / NLLLoss / CTCLoss on the log-probs;
here the log-softmax is folded into the loss kernels (one pass over the [N,V] logits, closed-form smoothed
target), so `forward` takes logits.  Values are identical up to fp32 rounding.
"""
from typing import Tuple

import torch
from torch import Tensor, nn

from joeys2t_amd import ops


_NAN = {}  # device -> 0-d NaN tensor (the placeholder gradient of _XentFn.backward's hand-over)


class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, trg, pad_index, smoothing):
        V = logits.shape[-1]
        l2 = logits.reshape(-1, V)
        if not l2.is_contiguous():
            l2 = l2.contiguous()
        t1 = trg.reshape(-1).contiguous()
        loss_rows, correct_rows, lse = ops.xent_fwd(l2, t1, pad_index, smoothing)
        ctx.l2, ctx.t1, ctx.lse = l2, t1, lse
        ctx.pad_index, ctx.smoothing, ctx.shape = pad_index, smoothing, logits.shape
        loss = ops.sum_f32(loss_rows)
        n_correct = ops.sum_f32(correct_rows)
        ctx.mark_non_differentiable(n_correct)
        return loss, n_correct

    @staticmethod
    def backward(ctx, g, _g2):
        g = g.contiguous().float()
        from joeys2t_amd import functional as Fn
        want = Fn.LOGIT_GRAD_DTYPE.pop(ctx.l2.data_ptr(), None)
        if want is not None and want != ctx.l2.dtype and ctx.l2.dtype == torch.float32:
            # the projection that made these f32 logits multiplies in bf16 (functional.LinearFn): its backward takes the gradient
            # in bf16 from the hand-over table - no f32 gradient [rows, V] is written and cast.  What autograd carries is a
            # placeholder of the logits' type that nothing reads (autograd insists on one): a NaN scalar expanded to their shape.
            d = ops.xent_bwd(ctx.l2, ctx.t1, ctx.lse, g, 1.0, ctx.pad_index, ctx.smoothing, out_dtype=want)
            nan = _NAN.get(ctx.l2.device)
            if nan is None:
                nan = _NAN[ctx.l2.device] = torch.full((), float("nan"), dtype=ctx.l2.dtype, device=ctx.l2.device)
            ph = nan.expand(ctx.shape)  # no memory, no kernel - and NaN everywhere should anything ever read it
            if len(Fn.LOGIT_GRAD_READY) > 16:  # (entries whose consumer never ran)
                Fn.LOGIT_GRAD_READY.clear()
            # keyed by the LOGITS (the projection's output storage), not by the placeholder: the cached NaN scalar has one address for
            # every call, and two losses of one backward pass would collide on it
            Fn.LOGIT_GRAD_READY[ctx.l2.data_ptr()] = (d, ph)
            return ph, None, None, None
        d = ops.xent_bwd(ctx.l2, ctx.t1, ctx.lse, g, 1.0, ctx.pad_index, ctx.smoothing)
        return d.view(ctx.shape), None, None, None


class _CtcFn(torch.autograd.Function):
    """pack (ops.PackedRows): the logits are the CTC projection of PACKED encoder rows, [1, pack.rows, V] - utterance b's frames are
    rows pack.seg[b] .. (the branch of a ragged batch on its live positions; the recursions index rows through js2t_ctc_alpha's
    row_offsets, alpha / beta keep their [B, T, S] layout)."""

    @staticmethod
    def forward(ctx, logits, targets, in_len, tgt_len, blank, zero_infinity, pack=None):
        ctx.pack, ctx.shape = pack, tuple(logits.shape)
        if pack is None:
            B, T, V = logits.shape
            l3 = logits.contiguous()
        else:
            V = logits.shape[-1]
            B, T = pack.B, pack.T
            l3 = logits.reshape(-1, V).contiguous()  # [rows, V]
        # f32 logits of a bf16 projection are registered for the cross-entropy's bf16 hand-over (functional.LOGIT_GRAD_DTYPE); this
        # loss takes its gradient the ordinary way, so the entry goes now - left behind, the address could be reused by f32 logits of
        # another producer within the step and send THEIR cross-entropy onto the hand-over path
        from joeys2t_amd import functional as Fn
        Fn.LOGIT_GRAD_DTYPE.pop(logits.data_ptr(), None)
        targets = targets.contiguous()
        in_len = in_len.to(torch.int64).contiguous()
        tgt_len = tgt_len.to(torch.int64).contiguous()
        lse, _ = ops.row_lse(l3.view(-1, V))
        # when a gradient will be asked for, the beta recursion runs beside alpha in the same launch
        alpha, nll, loss_rows, beta = ops.ctc_alpha(l3, lse, targets, in_len, tgt_len, blank, zero_infinity,
                                                    with_beta=logits.requires_grad, pack=pack)
        ctx.saved = (l3, lse, targets, in_len, tgt_len, alpha, nll, beta)
        ctx.blank, ctx.zero_infinity = blank, zero_infinity
        return ops.sum_f32(loss_rows)

    @staticmethod
    def backward(ctx, g):
        l3, lse, targets, in_len, tgt_len, alpha, nll, beta = ctx.saved
        g = g.contiguous().float()
        d = ops.ctc_bwd(l3, lse, targets, in_len, tgt_len, alpha, nll, g, 1.0, ctx.blank, ctx.zero_infinity, beta=beta, pack=ctx.pack)
        return d.view(ctx.shape), None, None, None, None, None, None


class XentLoss(nn.Module):
    """Cross-entropy with optional label smoothing (reference loss.py:16-107)."""

    def __init__(self, pad_index: int, smoothing: float = 0.0):
        super().__init__()
        self.smoothing = smoothing
        self.pad_index = pad_index
        self.require_ctc_layer = False
        self.takes_logits = True  # Model.forward hands us logits, not log-probs

    def xent(self, logits: Tensor, trg: Tensor) -> Tuple[Tensor, Tensor]:
        """(summed loss, n_correct) for logits [B,L,V] against trg [B,L]."""
        return _XentFn.apply(logits, trg, int(self.pad_index), float(max(self.smoothing, 0.0)))

    def forward(self, logits: Tensor, **kwargs) -> Tuple[Tensor]:
        assert "trg" in kwargs
        loss, _ = self.xent(logits, kwargs["trg"])
        return (loss, )

    def __repr__(self):
        return f"{self.__class__.__name__}(criterion=hip_ls_xent, smoothing={self.smoothing})"


class XentCTCLoss(XentLoss):
    """(1-w) * label-smoothed xent + w * CTC (reference loss.py:110-177); blank = BOS (model.py:84)."""

    def __init__(self, pad_index: int, bos_index: int, smoothing: float = 0.0, zero_infinity: bool = True,
                 ctc_weight: float = 0.3):
        super().__init__(pad_index=pad_index, smoothing=smoothing)
        self.require_ctc_layer = True
        self.bos_index = bos_index
        self.ctc_weight = ctc_weight
        self.zero_infinity = zero_infinity

    def ctc(self, ctc_logits: Tensor, trg: Tensor, input_lengths: Tensor, target_lengths: Tensor, pack=None) -> Tensor:
        return _CtcFn.apply(ctc_logits, trg, input_lengths, target_lengths, int(self.bos_index), bool(self.zero_infinity), pack)

    def forward(self, logits: Tensor, **kwargs) -> Tuple[Tensor, Tensor, Tensor]:
        assert "trg" in kwargs and "trg_length" in kwargs and "src_mask" in kwargs and "ctc_logits" in kwargs
        xent_loss, _ = self.xent(logits, kwargs["trg"])
        in_len = kwargs.get("ctc_input_lengths")
        if in_len is None:
            in_len = kwargs["src_mask"].squeeze(1).sum(dim=1)
        ctc_loss = self.ctc(kwargs["ctc_logits"], kwargs["trg"], in_len, kwargs["trg_length"])
        # interpolation of two 0-d tensors (loss.py:164); the reference's NaN / sign asserts (:166-167) force a
        # host sync per micro-batch and are left to the caller's logging cadence
        total_loss = ops.LinComb2Fn.apply(xent_loss, ctc_loss, 1.0 - self.ctc_weight, self.ctc_weight)
        return total_loss, xent_loss, ctc_loss

    def __repr__(self):
        return (f"{self.__class__.__name__}(criterion=hip_ls_xent, smoothing={self.smoothing}, "
                f"ctc=hip_ctc, ctc_weight={self.ctc_weight})")
