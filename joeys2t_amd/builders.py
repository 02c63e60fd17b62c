"""Activation selection (reference builders.py:24-41).  The returned nn.Module classes are placeholders inside
`pwff_layer` (so that repr()/indices match the reference: pwff_layer.0 / .3 are the Linears); the activation
itself is evaluated in the GEMM epilogue of libjoeys2t_hip.so."""
from typing import Callable

from torch import nn


class ConfigurationError(Exception):
    """Custom exception for misspecifications of configuration (reference config.py:22)."""


class _Tanh(nn.Module):
    def forward(self, x):  # pragma: no cover - placeholder, never called on the HIP path
        raise RuntimeError("placeholder activation module: the HIP GEMM epilogue applies tanh")


_ACTIVATIONS = {"relu": nn.ReLU, "gelu": nn.GELU, "tanh": _Tanh, "swish": nn.SiLU}


def build_activation(activation: str = "relu") -> Callable:
    """The module class behind a config's `activation` key (the HIP blocks only read its name: functional.BlockCfg.act)."""
    try:
        return _ACTIVATIONS[activation]
    except KeyError:
        raise ConfigurationError("Invalid activation function. Valid options: " + ", ".join(repr(k) for k in _ACTIVATIONS) + ".") from None


# ------------------------------------------------------------------------------------------------ update tail
import ctypes as _C  # noqa: E402
from typing import Dict, Optional  # noqa: E402

import torch  # noqa: E402


def trainable_ranges(store):
    """Contiguous [lo, hi) element ranges of the flat store that hold trainable parameters (alignment padding between
    two trainable parameters included; every boundary a multiple of 4 elements, as js2t_adamw's float4 passes need).
    A frozen parameter (requires_grad False; reference helpers.py:258-261 `freeze_params`) is left out entirely."""
    spans = sorted((store.offsets[id(p)], store.offsets[id(p)] + p.numel(), bool(p.requires_grad)) for p in store.params)
    if all(t for _, _, t in spans):
        return [(0, store.total)]
    ranges = []
    for i, (lo, hi, train) in enumerate(spans):
        if not train:
            continue
        nxt = spans[i + 1][0] if i + 1 < len(spans) else store.total
        hi = nxt if (i + 1 == len(spans) or spans[i + 1][2]) else (hi + 3) // 4 * 4  # padding rides with its predecessor
        if lo % 4 or hi % 4 or hi > nxt:
            raise ValueError("frozen and trainable parameters share a 16-byte granule of the flat store")
        if ranges and ranges[-1][1] == lo:
            ranges[-1] = (ranges[-1][0], hi)
        else:
            ranges.append((lo, hi))
    return ranges


def _off(t, lo: int):
    """device pointer of element `lo` of a flat buffer (None stays a null pointer)"""
    return None if t is None else _C.c_void_p(t.data_ptr() + lo * t.element_size())


import os as _os  # noqa: E402

FUSED_UPDATE = _os.environ.get("JS2T_FUSED_UPDATE", "1") != "0"  # js2t_adamw_items: update + transposed shadows + LayerNorm-fold weights in one pass (False: separate passes)


class SumsqCollector:
    """Where clip_grad_norm_'s sum of squares (builders.py:68-71) comes from when parts of it are left behind by the kernels
    that write the gradient: the un-split weight-gradient products store per-block sums of the values they write
    (js2t_gemm_desc.sumsq_partial) into segments of `partial`; finish() adds the pieces of the flat gradient nobody covered
    (js2t_sumsq_ranges) and turns everything into {norm, clip coefficient} (js2t_norm_clip)."""

    def __init__(self, grad: torch.Tensor, total: int):
        from joeys2t_amd._lib import lib
        self.grad, self.total = grad, total
        self.per_block = int(total // max(1, lib().js2t_sumsq_partials(_C.c_int64(total)))) or 1
        cap = 2 * int(lib().js2t_sumsq_partials(_C.c_int64(total))) + 8192
        self.partial = torch.zeros((cap, ), dtype=torch.float32, device=grad.device)
        self.pos, self.covered = 0, []
        self._tables: Dict[tuple, tuple] = {}

    def reset(self):
        self.pos, self.covered = 0, []

    def segment(self, n_blocks: int, spans):
        if self.pos + n_blocks > self.partial.numel():
            raise RuntimeError("SumsqCollector: partial buffer exhausted")
        seg = self.partial[self.pos:self.pos + n_blocks]
        self.pos += n_blocks
        self.covered.extend(spans)
        return seg

    def finish(self, max_norm: float, out2: torch.Tensor):
        from joeys2t_amd import ops
        from joeys2t_amd._lib import check, lib
        from joeys2t_amd.runtime import RangeSet
        key = (self.pos, tuple(sorted(self.covered)))
        hit = self._tables.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("SumsqCollector: a new coverage pattern during capture - run one eager step first")
            rest = RangeSet([(0, self.total)]).minus(self.covered)
            rows, pb = [], self.pos
            for lo, hi in rest:
                rows.append([lo, hi - lo, pb])
                pb += int(lib().js2t_sumsq_partials(_C.c_int64(hi - lo)))
            if pb > self.partial.numel():
                raise RuntimeError("SumsqCollector: partial buffer exhausted")
            table = torch.tensor(rows, dtype=torch.int64, device=self.grad.device) if rows else None
            hit = self._tables[key] = (table, len(rows), pb - self.pos, pb)
        table, n_ranges, n_blocks, n_partial = hit
        if n_ranges:
            check(lib().js2t_sumsq_ranges(ops._p(self.grad), ops._p(table), _C.c_int32(n_ranges), _C.c_int64(n_blocks), ops._p(self.partial),
                                          ops._stream()), "js2t_sumsq_ranges")
        check(lib().js2t_norm_clip(ops._p(self.partial), _C.c_int64(n_partial), _C.c_float(max_norm), ops._p(out2), ops._stream()),
              "js2t_norm_clip")
        self.reset()


class FlatAdamW:
    """AdamW over a runtime.ParamStore: one fused kernel (js2t_adamw) updates the fp32 master, both moments, the
    bf16 shadow and clears the gradient buffer.  Numerically the torch.optim.AdamW update the reference builds at
    builders.py:112-114 (betas from `adam_betas`, eps 1e-8, decoupled weight decay)."""

    def __init__(self, store, lr: float = 3e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.store = store
        self.param_groups = [{"lr": float(lr), "betas": tuple(betas), "eps": float(eps), "weight_decay": float(weight_decay)}]
        self.exp_avg = torch.zeros_like(store.flat)
        self.exp_avg_sq = torch.zeros_like(store.flat)
        self.t = 0
        self._retired_plans = []   # item / fold tables of earlier plans: captured graphs may still point at them (see _fused_plan)
        self.plan_generation = 0   # bumped whenever a plan that may have been captured is replaced
        from joeys2t_amd._lib import lib
        n = store.total
        self._partial = torch.empty((max(1, lib().js2t_sumsq_partials(_C.c_int64(n))), ), dtype=torch.float32,
                                    device=store.device)
        self.norm_clip = torch.ones((2, ), dtype=torch.float32, device=store.device)  # [grad norm, clip coefficient]
        # device-resident schedule state, used when the step is replayed from a captured hipGraph
        self.lr_dev = torch.full((1, ), float(lr), dtype=torch.float32, device=store.device)
        self.step_dev = torch.zeros((1, ), dtype=torch.int64, device=store.device)
        self.device_schedule = False
        self.update_ranges = trainable_ranges(store)
        self._plan_key, self._plan = None, None
        # set by TrainStep: runtime.RangeSet of gradient elements the update leaves un-cleared (their producers overwrite them,
        # runtime.WgradQueue) and the collector the weight-gradient epilogues leave their sums of squares with
        self.keep = None
        self.collector: Optional[SumsqCollector] = None


    def clip_and_step(self, max_norm: Optional[float], grad_scale: float = 1.0, zero_grad: bool = True):
        """clip_grad_norm_(max_norm) (builders.py:68-71) folded into the update: the coefficient stays on the device."""
        from joeys2t_amd import ops
        from joeys2t_amd._lib import check, lib
        st, g = self.store, self.param_groups[0]
        coef = None
        if max_norm is not None and max_norm > 0:
            if self.collector is not None and self.collector.covered:
                self.collector.finish(max_norm, self.norm_clip)  # the products' epilogues hold most of the sum already
            else:
                check(lib().js2t_grad_norm_clip(ops._p(st.flat_grad), _C.c_int64(st.total), _C.c_float(max_norm),
                                                ops._p(self._partial), ops._p(self.norm_clip), ops._stream()), "js2t_grad_norm_clip")
            coef = self.norm_clip[1:2]
        if self.collector is not None:
            self.collector.reset()
        self.t += 1
        lp = st.flat_lp
        lr_dev = step_dev = None
        if self.device_schedule:  # graph-replayable form: count and learning rate live on the device
            self.step_dev.add_(1)
            lr_dev, step_dev = self.lr_dev, self.step_dev
        plan = self._fused_plan()
        if plan is not None:
            # one table-driven kernel, launched twice: the 1-D parameters first (a LayerNorm fold reads the NEW gamma / beta /
            # bias), then the matrices - each block also writes the transposed bf16 image and the fold weights of its rows
            args = (_C.c_float(g["lr"]), _C.c_float(g["betas"][0]), _C.c_float(g["betas"][1]), _C.c_float(g["eps"]),
                    _C.c_float(g["weight_decay"]), _C.c_int64(self.t), ops._p(coef), _C.c_float(grad_scale), int(zero_grad),
                    ops._p(lr_dev), ops._p(step_dev), ops._stream())
            for table, n_items, n_units in plan["launches"]:
                check(lib().js2t_adamw_items(ops._p(st.flat), ops._p(st.flat_grad), ops._p(self.exp_avg), ops._p(self.exp_avg_sq),
                                             ops._p(lp), ops._p(st.flat_lp_t if lp is not None else None), ops._p(table),
                                             _C.c_int32(n_items), _C.c_int64(n_units), ops._p(plan["folds"]), *args), "js2t_adamw_items")
            ops.WEIGHT_VERSION += 1
            st.dirty = lp is None and st.dirty
            if plan["left_folds"] is not None:  # folds whose matrix the kernel does not cover (wider than a unit, part of a group)
                ops.fold_ln_weights(plan["left_folds"], plan["left_folds"].shape[0], plan["left_max_rows"])
            return
        # torch.optim.AdamW skips parameters without a gradient: frozen sub-networks (`freeze: True`) get no weight decay
        # and no moment update - one launch per contiguous trainable range (the whole store when nothing is frozen)
        for lo, hi in self.update_ranges:
            check(lib().js2t_adamw(_off(st.flat, lo), _off(st.flat_grad, lo), _off(self.exp_avg, lo), _off(self.exp_avg_sq, lo),
                                   _off(lp, lo), _C.c_int64(hi - lo), _C.c_float(g["lr"]), _C.c_float(g["betas"][0]),
                                   _C.c_float(g["betas"][1]), _C.c_float(g["eps"]), _C.c_float(g["weight_decay"]), _C.c_int64(self.t),
                                   ops._p(coef), _C.c_float(grad_scale), int(zero_grad), ops._p(lr_dev), ops._p(step_dev),
                                   ops._stream()), "js2t_adamw")
        ops.WEIGHT_VERSION += 1  # cached e4m3 copies of the weights (functional.FP8_FORWARD) are stale now
        st.dirty = lp is None and st.dirty
        if lp is not None and st.flat_lp_t is not None:
            st.refresh_t()  # transposed shadows follow the updated weights (one kernel; part of the captured step)
        if lp is not None:
            st.refresh_folds()  # gamma-scaled weights of the LayerNorm folds, likewise

    def _fused_plan(self):
        """Tables of js2t_adamw_items for the store as it is now (which matrices have a transposed shadow, which LayerNorm folds
        exist - both appear with the first forward pass), or None: frozen parameters / a matrix the kernel's 16-byte accesses
        do not fit -> the plain kernel + separate passes.  Rebuilt when the store's derived state has changed."""
        st = self.store
        if not FUSED_UPDATE or self.update_ranges != [(0, st.total)]:
            return None
        key = (len(st._fold_rows), st.flat_lp is not None, st.flat_lp_t is not None, None if self.keep is None else self.keep.version)
        if self._plan_key == key:
            return self._plan
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FlatAdamW: the store changed (new LayerNorm folds / shadows) after the last eager update: "
                               "run one eager step before capturing")
        # A rebuilt plan does not free the old one: hipGraphs captured earlier hold raw pointers to its tables (replay never comes
        # through here), and reused memory read as an item table would send the kernel's writes anywhere.  Every table ever handed to
        # a launch stays alive (a few KB each); `plan_generation` tells holders of captured graphs that their graphs carry the OLD
        # keep flags / folds and must be re-captured (graphed.GraphedTrainStep checks it before every replay).
        if self._plan is not None:
            self._retired_plans.append(self._plan)
            self.plan_generation += 1
        self._plan_key, self._plan = key, None
        from joeys2t_amd._lib import lib
        geo = [_C.c_int32() for _ in range(3)]
        lib().js2t_adamw_items_geometry(*[_C.byref(v) for v in geo])
        flat_unit, unit_rows, unit_cols = (int(v.value) for v in geo)
        mats = sorted((off, R, Cc) for off, R, Cc, _ in st._tgroups)
        if any(Cc % 4 or off % 4 for off, R, Cc in mats):
            return None
        base = st.flat.data_ptr()
        fold_of, left = {}, []
        for i, row in enumerate(st._fold_rows):  # {W, gamma, beta, bias, Wf, bias_f, N, K}
            hit = [(off, R, Cc) for off, R, Cc in mats if base + 4 * off == row[0] and R == row[6] and Cc == row[7]]
            if hit and row[7] <= unit_cols and hit[0][0] not in fold_of:
                fold_of[hit[0][0]] = i
            else:
                left.append(row)
        small, big, pos = [], [], 0
        for off, R, Cc in mats:
            if off > pos:
                small.append([0, pos, off - pos, 1, 0, -1, 0, 0])
            kept = int(self.keep is not None and self.keep.contains(off, off + R * Cc))  # overwritten by its producer: not cleared
            big.append([1, off, R, Cc, 0, fold_of.get(off, -1), kept, 0])
            pos = off + R * Cc
        if st.total > pos:
            small.append([0, pos, st.total - pos, 1, 0, -1, 0, 0])
        launches = []
        for items in (small, big):
            if not items:
                continue
            units = 0
            for it in items:
                it[4] = units
                units += -(-it[2] // flat_unit) if it[0] == 0 else -(-it[2] // unit_rows) * -(-it[3] // unit_cols)
            launches.append((torch.tensor(items, dtype=torch.int64, device=st.device), len(items), units))
        folds = torch.tensor(st._fold_rows, dtype=torch.int64, device=st.device) if st._fold_rows else None
        self._plan = {"launches": launches, "folds": folds,
                      "left_folds": torch.tensor(left, dtype=torch.int64, device=st.device) if left else None,
                      "left_max_rows": max((r[6] for r in left), default=0)}
        return self._plan

    def state_dict(self) -> Dict:
        return {"t": self.t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "param_groups": self.param_groups}

    def load_state_dict(self, sd: Dict):
        self.t = sd["t"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups = sd["param_groups"]
        self.step_dev.fill_(self.t)
        self.lr_dev.fill_(self.param_groups[0]["lr"])

    # ---- torch.optim.AdamW layout, as the reference's checkpoints carry it (training.py:166-177,251-252) ----------
    def torch_state_dict(self, params) -> Dict:
        """`params`: the model's parameters in model.parameters() order (= the index space of torch's optimizer state)."""
        st = self.store
        state = {}
        if self.t > 0:
            for i, p in enumerate(params):
                off, n = st.offsets[id(p)], p.numel()
                state[i] = {"step": torch.tensor(float(self.t)), "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_torch_state_dict(self, sd: Dict, params):
        st = self.store
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, p in enumerate(params):
            s = sd["state"].get(i)
            if s is None:
                continue
            off, n = st.offsets[id(p)], p.numel()
            self.exp_avg[off:off + n].view(p.shape).copy_(s["exp_avg"])
            self.exp_avg_sq[off:off + n].view(p.shape).copy_(s["exp_avg_sq"])
            steps.add(int(float(s["step"])))
        if len(steps) > 1:
            raise ValueError(f"optimizer state with differing step counts {sorted(steps)} cannot be held by one fused AdamW")
        self.t = steps.pop() if steps else 0
        g, src = self.param_groups[0], sd["param_groups"][0]
        g.update(lr=float(src["lr"]), betas=tuple(src["betas"]), eps=float(src["eps"]), weight_decay=float(src["weight_decay"]))
        self.step_dev.fill_(self.t)
        self.lr_dev.fill_(g["lr"])


class WarmupInverseSquareRootScheduler:
    """lr = step*peak/warmup while step < warmup, else peak*sqrt(warmup)/sqrt(step), floored at min_rate
    (reference builders.py:418-485).  `step(n)` sets the internal step to n+1 first (builders.py:278-284), so with the
    trainer calling scheduler.step(stats.steps) AFTER optimizer.step() the first update uses the configured LR."""

    def __init__(self, optimizer, peak_rate: float = 1.0e-3, warmup: int = 10000, min_rate: float = 1.0e-5):
        self.optimizer = optimizer
        self.warmup, self.min_rate, self.peak_rate = warmup, min_rate, peak_rate
        self.decay_rate = peak_rate * (warmup**0.5)
        self._step, self._rate = 0, 0

    def _compute_rate(self) -> float:
        step = self._step
        rate = step * self.peak_rate / self.warmup if step < self.warmup else self.decay_rate * (step**-0.5)
        return max(rate, self.min_rate)

    def step(self, step: int):
        self._step = step + 1
        self._rate = self._compute_rate()
        for g in self.optimizer.param_groups:
            g["lr"] = self._rate

    def state_dict(self):
        return {"step": self._step, "rate": self._rate, "warmup": self.warmup, "peak_rate": self.peak_rate,
                "decay_rate": self.decay_rate, "min_rate": self.min_rate}

    def load_state_dict(self, sd):
        """The reference stores {"step", "rate"} only (BaseScheduler, builders.py:265-276); the shape parameters come from the
        config in both implementations, extra keys of this class's own dumps are honoured when present."""
        self._step, self._rate = sd["step"], sd["rate"]
        self.warmup, self.peak_rate, self.min_rate = sd.get("warmup", self.warmup), sd.get("peak_rate", self.peak_rate), sd.get("min_rate", self.min_rate)
        self.decay_rate = sd.get("decay_rate", self.peak_rate * (self.warmup**0.5))
