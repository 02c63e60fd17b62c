"""Per-model runtime state for the HIP path: compute dtype, dropout RNG and the flat parameter store.

MI355X-first design notes
  * All parameters of a model live in ONE flat fp32 buffer (`ParamStore.flat`), their gradients in one flat
    fp32 buffer (`flat_grad`) and — for bf16 compute — a flat bf16 shadow (`flat_lp`).  `param.data` and
    `param.grad` are views into those buffers, so checkpoint names/shapes are unchanged
    (reference contract: transformer_layers.py:235-236) while
      - the DDP gradient exchange is a handful of large RCCL collectives on `flat_grad` (helpers_for_ddp),
      - AdamW + the bf16 re-cast is one fused kernel over the flat buffers,
      - q/k/v (and k/v) projection weights are adjacent, so a fused [3d,d] weight is a *view*.
  * With 288 GB of HBM per GPU there is no reason to shard or recompute any of this.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from joeys2t_amd import ops

_ALIGN = 64  # elements; keeps every group start 128/256-byte aligned in bf16/f32


class ParamStore:
    """Flat storage for a module's parameters (+ grads, + bf16 shadow)."""

    def __init__(self, module: nn.Module, device: torch.device):
        self.device = torch.device(device)
        # Parameters in registration (~forward) order; a parameter that belongs to a fusion group pulls its whole
        # group in at that point, so [k;v;q] weights are adjacent AND the flat order stays layer-by-layer (backward
        # fills the buffer from its tail towards its head -> contiguous DDP buckets become ready in order).
        group_of: Dict[int, List[nn.Parameter]] = {}
        for m in module.modules():
            fg = getattr(m, "fuse_groups", None)
            if fg is None:
                continue
            for grp in fg():
                grp = [p for p in grp if p is not None]
                if any(id(p) in group_of for p in grp):
                    continue
                for p in grp:
                    group_of[id(p)] = grp
        groups: List[List[nn.Parameter]] = []
        seen = set()
        for p in module.parameters():
            if id(p) in seen:
                continue
            grp = group_of.get(id(p), [p])
            for q in grp:
                seen.add(id(q))
            groups.append(grp)
        # Type-major order for the nn.Linear parameters: after everything else (registration order), the weight groups of one
        # shape from ALL layers are adjacent, followed by their bias groups.  The deferred weight-gradient products
        # (WgradQueue) run type by type, so each such range of the flat gradient is complete - and can be handed to RCCL -
        # while the next type's products still run.  `type_ranges` lists (lo, hi) of the prefix and of every type.
        # ... and SIDE-major above that: the nn.Linear parameters behind the encoder's output (everything registered under
        # `decoder.` / `trg_embed.`, unless the encoder side shares it) come after all the others.  Their gradients are complete
        # once the decoder's backward is - the backward pass is cut there (TrainStep.micro_step) and their ranges travel while the
        # encoder's backward still runs.  `late_ranges` names those ranges of `type_ranges`.
        late_ids = set()
        if hasattr(module, "named_parameters"):
            early_ids = {id(p) for n, p in module.named_parameters(remove_duplicate=False) if not n.startswith(("decoder.", "trg_embed."))}
            late_ids = {id(p) for n, p in module.named_parameters(remove_duplicate=False) if n.startswith(("decoder.", "trg_embed."))} - early_ids
        lin_key: Dict[int, Tuple] = {}
        for m in module.modules():
            if isinstance(m, nn.Linear):
                wg = group_of.get(id(m.weight), [m.weight])
                bg = [] if m.bias is None else group_of.get(id(m.bias), [m.bias])
                side = int(all(id(p) in late_ids for p in (*wg, *bg)))
                key = (side, sum(p.shape[0] for p in wg), wg[0].shape[1])
                for p in wg:
                    lin_key[id(p)] = key + (0, )
                for p in bg:
                    lin_key.setdefault(id(p), key + (1, ))
        order = sorted(range(len(groups)), key=lambda i: ((1, ) + lin_key[id(groups[i][0])] if id(groups[i][0]) in lin_key else (0, )) + (i, ))
        groups = [groups[i] for i in order]
        offsets: Dict[int, int] = {}
        total = 0
        self.type_ranges: List[Tuple[int, int]] = []
        self.late_ranges: List[Tuple[int, int]] = []
        cur_key, cur_lo = "prefix", 0
        for grp in groups:
            k = lin_key.get(id(grp[0]))
            k = "prefix" if k is None else k[:3]
            if k != cur_key:
                boundary = (total + _ALIGN - 1) // _ALIGN * _ALIGN
                if boundary > cur_lo:
                    self.type_ranges.append((cur_lo, boundary))
                    if cur_key != "prefix" and cur_key[0] == 1:
                        self.late_ranges.append((cur_lo, boundary))
                cur_key, cur_lo = k, boundary
            total = (total + _ALIGN - 1) // _ALIGN * _ALIGN
            for p in grp:
                offsets[id(p)] = total
                total += p.numel()
        total = (total + _ALIGN - 1) // _ALIGN * _ALIGN
        if total > cur_lo:
            self.type_ranges.append((cur_lo, total))
            if cur_key != "prefix" and cur_key[0] == 1:
                self.late_ranges.append((cur_lo, total))
        self.total = total
        self.offsets = offsets
        self.params: List[nn.Parameter] = [p for grp in groups for p in grp]
        # 2-D weight groups ([k;v;q] as ONE [3d, d] matrix, every other matrix alone): their transposed bf16 shadows make the
        # input gradient of every Linear a product of two k-contiguous operands (see refresh_t / view_t)
        self._tgroups: List[Tuple[int, int, int, List[nn.Parameter]]] = []
        self._tgroup_of: Dict[int, int] = {}
        for grp in groups:
            # (rows of a multiple of 16 bytes in fp32: what js2t_adamw_items takes; a [H, 2R+1] bias table is no nn.Linear weight)
            if all(p.dim() == 2 for p in grp) and len({p.shape[1] for p in grp}) == 1 and grp[0].shape[1] % 4 == 0:
                rows = sum(p.shape[0] for p in grp)
                for p in grp:
                    self._tgroup_of[id(p)] = len(self._tgroups)
                self._tgroups.append((offsets[id(grp[0])], rows, grp[0].shape[1], list(grp)))
        self.flat_lp_t: Optional[torch.Tensor] = None
        self._ttable = None
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_lp: Optional[torch.Tensor] = None
        self.dirty = True
        self.auto_refresh = True
        with torch.no_grad():
            for p in self.params:
                off, n = offsets[id(p)], p.numel()
                view = self.flat[off:off + n].view(p.shape)
                view.copy_(p.data.to(device=self.device, dtype=torch.float32))
                p.data = view
                p.grad = None
        self._cache: Dict[Tuple, torch.Tensor] = {}
        # LayerNorm folds (fold()): derived weights per (LayerNorm, Linear group) pair + the table js2t_fold_ln_weights walks
        self._folds: Dict[Tuple, LnFold] = {}
        self._fold_rows: List[List[int]] = []
        self._retired_fold_tables: List[torch.Tensor] = []
        self._fold_table: Optional[torch.Tensor] = None
        self._fold_max_rows = 0

    # ---- LayerNorm folded into the consuming product ----------------------------------------------
    def fold(self, weights: Sequence[nn.Parameter], biases: Sequence[nn.Parameter], gamma: nn.Parameter, beta: nn.Parameter):
        """The LnFold of LayerNorm(gamma, beta) followed by the Linear whose (adjacent) weights / biases are given, created on
        first use; None when the parameters are not laid out for it (not adjacent, N % 128 != 0, K % 8 != 0)."""
        key = tuple(id(p) for p in (*weights, *biases, gamma, beta))
        hit = self._folds.get(key)
        if hit is not None:
            return hit
        if key in self._folds:
            return None
        w32, b32 = self.view(list(weights), torch.float32), self.view(list(biases), torch.float32)
        ok = (w32 is not None and b32 is not None and w32.dim() == 2 and id(gamma) in self.offsets and id(beta) in self.offsets and
              w32.shape[0] % 128 == 0 and w32.shape[1] % 8 == 0 and gamma.numel() == w32.shape[1] and b32.numel() == w32.shape[0])
        if not ok or torch.cuda.is_current_stream_capturing():
            if not ok:
                self._folds[key] = None
            return None
        N, K = w32.shape
        f = LnFold(torch.empty((N, K), dtype=torch.bfloat16, device=self.device), torch.empty((N, ), dtype=torch.float32, device=self.device))
        self._folds[key] = f
        self._fold_rows.append([w32.data_ptr(), gamma.data.data_ptr(), beta.data.data_ptr(), b32.data_ptr(), f.w.data_ptr(),
                                f.bias.data_ptr(), N, K])
        self._fold_max_rows = max(self._fold_max_rows, N)
        if self._fold_table is not None:  # a captured refresh_folds() launch may still read it: kept, not freed (builders._fused_plan)
            self._retired_fold_tables.append(self._fold_table)
        self._fold_table = None
        row = torch.tensor([self._fold_rows[-1]], dtype=torch.int64, device=self.device)
        ops.fold_ln_weights(row, 1, N)  # this pair now; all pairs together after every update (refresh_folds)
        return f

    def refresh_folds(self):
        """Re-derive every fold from the fp32 masters (one launch): after an optimizer update / load_state_dict."""
        if not self._fold_rows:
            return
        if self._fold_table is None:
            if torch.cuda.is_current_stream_capturing():
                raise ops.Js2tError("LayerNorm folds were created after the last eager update: run one eager step before capturing")
            self._fold_table = torch.tensor(self._fold_rows, dtype=torch.int64, device=self.device)
        ops.fold_ln_weights(self._fold_table, len(self._fold_rows), self._fold_max_rows)

    # ---- gradient views -------------------------------------------------------------------------
    def attach_grads(self, zero: bool = True):
        """Point every param.grad at its slice of flat_grad (autograd then accumulates in place)."""
        if zero:
            self.flat_grad.zero_()
        for p in self.params:
            if p.requires_grad:
                off, n = self.offsets[id(p)], p.numel()
                p.grad = self.flat_grad[off:off + n].view(p.shape)

    def grad_view(self, params: Sequence[nn.Parameter]) -> Optional[torch.Tensor]:
        """fp32 view of flat_grad over adjacent parameters — the target kernels accumulate gradients into directly.
        None unless every parameter requires grad and its .grad currently IS its slice of flat_grad."""
        try:
            offs = [self.offsets[id(p)] for p in params]
        except KeyError:
            return None
        base_ptr = self.flat_grad.data_ptr()
        for i, p in enumerate(params):
            if not p.requires_grad or p.grad is None or p.grad.data_ptr() != base_ptr + 4 * offs[i]:
                return None
            if i and (offs[i] != offs[i - 1] + params[i - 1].numel() or p.shape[1:] != params[0].shape[1:]):
                return None
        rows = sum(p.shape[0] for p in params)
        n = sum(p.numel() for p in params)
        return self.flat_grad[offs[0]:offs[0] + n].view(rows, *params[0].shape[1:])

    # ---- compute-dtype views --------------------------------------------------------------------
    def refresh(self, force: bool = False):
        """Re-cast the bf16 shadow from the fp32 master (one kernel over the whole store)."""
        if self.flat_lp is None:
            self.flat_lp = torch.empty(self.total, dtype=torch.bfloat16, device=self.device)
            self._cache.clear()
            force = True
        if force or self.dirty:
            ops.cast(self.flat, torch.bfloat16, out=self.flat_lp)
            self.dirty = False
            if self.flat_lp_t is not None:
                self.refresh_t()
            self.refresh_folds()

    def refresh_t(self):
        """(Re)build the transposed bf16 shadows of all 2-D weight groups from the bf16 shadow: one kernel over a table."""
        if self.flat_lp is None:
            self.refresh()
        if self.flat_lp_t is None:
            self.flat_lp_t = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
            rows, t0 = [], 0
            for off, R, Cc, _ in self._tgroups:
                rows.append([off, R, Cc, t0])
                t0 += ((R + 63) // 64) * ((Cc + 63) // 64)
            self._ttable = (torch.tensor(rows, dtype=torch.int64, device=self.device), len(rows), t0)
            self._cache_t: Dict[Tuple, torch.Tensor] = {}
        table, n, tiles = self._ttable
        ops.transpose_groups(self.flat_lp, self.flat_lp_t, table, n, tiles)

    def view_t(self, params: Sequence[nn.Parameter]) -> Optional[torch.Tensor]:
        """bf16 [cols, sum rows] view of the transposed shadow over adjacent parameters of one 2-D group (a strided view
        when the parameters are only part of their group, e.g. the k|v rows of a [k;v;q] block), or None."""
        gi = self._tgroup_of.get(id(params[0]))
        if gi is None or any(self._tgroup_of.get(id(p)) != gi for p in params):
            return None
        if self.flat_lp_t is None:
            self.refresh_t()
        key = tuple(id(p) for p in params)
        hit = self._cache_t.get(key)
        if hit is not None:
            return hit
        off, R, Cc, members = self._tgroups[gi]
        r0 = (self.offsets[id(params[0])] - off) // Cc
        rows = sum(p.shape[0] for p in params)
        for a, b in zip(params[:-1], params[1:]):
            if self.offsets[id(b)] != self.offsets[id(a)] + a.numel():
                return None
        out = self.flat_lp_t[off:off + R * Cc].view(Cc, R)[:, r0:r0 + rows]
        self._cache_t[key] = out
        return out

    def mark_dirty(self):
        self.dirty = True

    def view(self, params: Sequence[nn.Parameter], dtype: torch.dtype) -> Optional[torch.Tensor]:
        """A [sum rows, ...] tensor over adjacent parameters in `dtype`, or None if they are not adjacent."""
        key = (tuple(id(p) for p in params), dtype)
        hit = self._cache.get(key)
        if hit is not None:
            return hit
        try:
            offs = [self.offsets[id(p)] for p in params]
        except KeyError:
            return None
        for i in range(1, len(params)):
            if offs[i] != offs[i - 1] + params[i - 1].numel() or params[i].shape[1:] != params[0].shape[1:]:
                return None
        rows = sum(p.shape[0] for p in params)
        n = sum(p.numel() for p in params)
        if dtype == torch.float32:
            base = self.flat
        else:
            if self.flat_lp is None:
                self.refresh()
            base = self.flat_lp
        out = base[offs[0]:offs[0] + n].view(rows, *params[0].shape[1:])
        self._cache[key] = out
        return out


class LnFold:
    """Derived operands of one LayerNorm -> nn.Linear pair for js2t_gemm's ln_partial mode: w = bf16(W gamma - rowmean(W gamma))
    [N, K] and bias f32[N] = b + W beta.  Owned and kept current by ParamStore.refresh_folds()."""
    __slots__ = ("w", "bias")

    def __init__(self, w, bias):
        self.w, self.bias = w, bias


class WgradQueue:
    """Weight-gradient products put off until the backward pass is over (or `flush_every` of them are waiting).

    A single dW = dY^T X of a 512/2048-wide layer has 16..64 output tiles for 256 CUs, so on its own it has to cut the
    token dimension into slices that are summed with f32 atomics - the atomics cost as much as the product.  All layers
    of one type together have enough tiles: queued products of one shape run as ONE grouped launch
    (ops.gemm_grouped) with no, or a much smaller, split.  The operands (dY, X) are kept alive by the queue; 288 GB of
    HBM make the extra ~2 GB irrelevant.  The bias gradient rides along (a_rowsum)."""

    def __init__(self, flush_every: Optional[int] = None):
        self.groups = {}
        self.pending = 0
        self.flush_every = flush_every
        self.after_flush = []  # callables run once the queued gradients are complete (DDP bucket bookkeeping)
        # set by TrainStep (all optional):
        #   cand: RangeSet of flat-gradient elements ONE product per micro-batch may overwrite (weights of nn.Linear modules that
        #         are not shared); an un-split product of the FIRST micro-batch of an update then runs with beta = 0 - no read of
        #         the old gradient - and its span joins `kept`, the pieces the update (FlatAdamW) no longer clears.  A kept piece
        #         that a first flush does not overwrite is cleared here, before the queued products run, and leaves the set.
        #         (Assumption: a weight gradient the queue has overwritten once is produced by the queue or not at all.)
        #   first: this flush belongs to the first micro-batch of an update; collector: builders.SumsqCollector that takes the
        #         sums of squares of the final gradients from the products' epilogues (last micro-batch, single GPU).
        self.cand = None
        self.kept = None
        self.first = False
        self.collector = None
        self.grad_base = None  # (flat gradient tensor) element offsets of the dW views are taken against it
        self._written = []     # spans overwritten by earlier (non-final) takes of this micro-batch

    def add(self, dz2d: torch.Tensor, x2d: torch.Tensor, dw_out: torch.Tensor, db_out: Optional[torch.Tensor]):
        N, K, M = dz2d.shape[1], x2d.shape[1], dz2d.shape[0]
        key = (N, K, M, dz2d.stride(0), x2d.stride(0), db_out is not None)
        self.groups.setdefault(key, []).append((dz2d, x2d, dw_out, db_out))
        self.pending += 1
        if self.flush_every is not None and self.pending >= self.flush_every:
            self.flush()

    def flush(self, on_group_done=None):
        """Run the queued products, one grouped launch per shape (largest first); `on_group_done(items)` is called after each
        launch with its (dY, X, dW, db) entries (the gradient exchange of completed ranges starts from there)."""
        self.run(self.take(), on_group_done)
        cbs, self.after_flush = self.after_flush, []
        for cb in cbs:
            cb()

    def _span(self, t: torch.Tensor):
        lo = (t.data_ptr() - self.grad_base.data_ptr()) // 4
        return (lo, lo + t.numel())

    def take(self, final: bool = True):
        """The queued products as a plan [(key + (mode, collector), items)], emptying the queue.  final=False: more products of
        this micro-batch will follow (the backward pass is cut at the encoder's output, TrainStep.micro_step): the pieces nothing
        overwrote are only settled by the last take.  A plan built while a hipGraph was
        captured stays valid for every replay (its tensors live in the graph's static pool): run(plan) re-issues the launches.
        mode: 1 = the products overwrite their dW (beta 0), 2 = their epilogues leave the sums of squares with the collector."""
        from joeys2t_amd.functional import wgrad_split
        # The order of the plan is the order in which the ranges of the flat gradient complete, i.e. the order of the all-reduces,
        # and that must be the same on every rank - while the token counts M differ from rank to rank and from batch to batch.
        # So nothing M-dependent may decide it: products are ordered by CLASS = (N, K, bias) - largest weight x members first, ties
        # in the order of arrival (the model's) - and inside a class the launches (one per distinct M / row stride: a grouped
        # launch has one token count) follow each other in the order of arrival.  Round 5 sorted the (N, K, M) groups themselves:
        # encoder and decoder products of one (N, K) were ONE group of 2n members on a rank whose B T' happened to equal B L and
        # two groups of n elsewhere - another sort key, another order of collectives.
        classes: Dict[tuple, list] = {}
        for key, items in self.groups.items():  # dicts keep insertion order: first arrival of a class, of a group inside it
            classes.setdefault((key[0], key[1], key[5]), []).append((key, items))
        plan = []
        for ck, members in sorted(classes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * sum(len(it) for _, it in kv[1])):
            plan.extend(members)
        self.groups = {}
        self.pending = 0
        cand, out, written = self.cand, [], []
        active = cand is not None and self.grad_base is not None and self.flush_every is None
        seen: Dict[int, int] = {}
        if active:
            for _, items in plan:  # a dW that two products add into (a layer applied twice) cannot be overwritten by either
                for it in items:
                    seen[it[2].data_ptr()] = seen.get(it[2].data_ptr(), 0) + 1
        for key, items in plan:
            mode = 0
            if active:
                N, K, M = key[:3]
                spans = [self._span(it[2]) for it in items]
                solo = all(seen[it[2].data_ptr()] == 1 and it[2].is_contiguous() and it[2].dtype == torch.float32 for it in items)
                # un-split whatever the number of tokens (enough output tiles for the chip), not just for this batch
                unsplit = wgrad_split(N, K, M, count=len(items)) == 1 and wgrad_split(N, K, 1 << 30, count=len(items)) == 1
                if solo and unsplit and all(cand.contains(*s) for s in spans):
                    if self.first:
                        mode |= 1
                        written.extend(spans)
                    if self.collector is not None and K % 128 == 0:
                        mode |= 2
            out.append((key + (mode, self.collector if mode & 2 else None), items))
        if active and self.first:
            written = self._written + written
            self._written = [] if final else written
            if final:
                for lo, hi in self.kept.minus(written):  # un-cleared, and nothing overwrites it this time
                    self.grad_base[lo:hi].zero_()
                    self.kept.remove(lo, hi)
            for lo, hi in written:
                if not self.kept.contains(lo, hi):
                    self.kept.add(lo, hi)
        return out

    @staticmethod
    def run(plan, on_group_done=None):
        from joeys2t_amd.functional import wgrad_split
        for key, items in plan:
            N, K, M, lda, ldb, has_db = key[:6]
            mode, collector = (key[6], key[7]) if len(key) > 6 else (0, None)
            n = len(items)
            sk = wgrad_split(N, K, M, count=n)
            part = None
            if mode & 2:
                lo = [(it[2].data_ptr() - collector.grad.data_ptr()) // 4 for it in items]
                part = collector.segment(ops.grouped_blocks(N, K, n), [(a, a + it[2].numel()) for a, it in zip(lo, items)])
            ops.gemm_grouped([it[0] for it in items], [it[1] for it in items], [it[2] for it in items], M=N, N=K, K=M, lda=lda,
                             ldb=ldb, ldc=K, split_k=sk, beta=0.0 if (sk > 1 or mode & 1) else 1.0,
                             a_rowsums=[it[3] for it in items] if has_db else None, sumsq_partial=part)
            if on_group_done is not None:
                on_group_done(items)


class RangeSet:
    """Disjoint, sorted [lo, hi) element ranges (adjacent ones merged)."""

    def __init__(self, ranges=()):
        self.r: List[Tuple[int, int]] = []
        for lo, hi in sorted(ranges):
            if self.r and lo <= self.r[-1][1]:
                self.r[-1] = (self.r[-1][0], max(hi, self.r[-1][1]))
            elif hi > lo:
                self.r.append((lo, hi))
        self.version = 0

    def contains(self, lo: int, hi: int) -> bool:
        return any(a <= lo and hi <= b for a, b in self.r)

    def minus(self, spans) -> List[Tuple[int, int]]:
        """the parts of this set that no span covers"""
        out = []
        cut = RangeSet(spans).r
        for a, b in self.r:
            pos = a
            for lo, hi in cut:
                if hi <= pos or lo >= b:
                    continue
                if lo > pos:
                    out.append((pos, lo))
                pos = max(pos, hi)
            if pos < b:
                out.append((pos, b))
        return out

    def remove(self, lo: int, hi: int):
        self.r = RangeSet(self.minus([(lo, hi)])).r
        self.version += 1

    def add(self, lo: int, hi: int):
        self.r = RangeSet(self.r + [(lo, hi)]).r
        self.version += 1

    def __bool__(self):
        return bool(self.r)


class Runtime:
    """What every HIP-backed module needs at call time."""

    def __init__(self, device=None, compute_dtype: torch.dtype = torch.float32):
        self.device = torch.device(device) if device is not None else None
        self.compute_dtype = compute_dtype
        self.store: Optional[ParamStore] = None
        self._rng: Optional[ops.DropoutRng] = None
        self.direct_grads = True  # kernels accumulate parameter gradients straight into the flat gradient buffer
        self.on_grads_ready = None  # callable(list of params): DDP bucket bookkeeping for directly written gradients
        self.wgrad_queue: Optional[WgradQueue] = None  # set (TrainStep) to defer + group the weight-gradient products
        self.grad_copies = None  # ops.GradCopies (TrainStep): LayerNorm parameter gradients accumulate into folded copies

    @property
    def rng(self) -> ops.DropoutRng:
        if self._rng is None:
            if self.device is None or self.device.type != "cuda":
                raise ops.Js2tError("dropout RNG needs a GPU runtime (model not finalized on a cuda device)")
            self._rng = ops.DropoutRng(self.device)
        return self._rng

    # weights in compute dtype: fused view when the store has the params adjacent, else gathered by kernels
    def side_stream(self) -> "torch.cuda.Stream":
        """Second stream of this runtime's device for work that is independent of the main chain (Model.overlap_ctc)."""
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        return self._side_stream

    def weight(self, params: Sequence[nn.Parameter]) -> torch.Tensor:
        dt = self.compute_dtype
        if self.store is not None:
            v = self.store.view(params, dt)
            if v is not None:
                return v
        ws = []
        for p in params:
            ws.append(p.data if p.dtype == dt else ops.cast(p.data, dt))
        return ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)

    def ln_fold(self, weights, biases, ln: Optional[nn.LayerNorm]) -> Optional[LnFold]:
        """Folded operands for `ln` -> Linear(weights, biases) (functional.LN_FOLD), or None: bf16 compute on a flat store only."""
        if ln is None or self.store is None or self.compute_dtype != torch.bfloat16 or any(b is None for b in biases):
            return None
        return self.store.fold(weights, biases, ln.weight, ln.bias)

    def weight_t(self, params: Sequence[nn.Parameter]) -> Optional[torch.Tensor]:
        """Transposed compute-dtype weight [in, out] for the input-gradient product, when the store keeps one (bf16 compute,
        gradients enabled); None otherwise (the caller then multiplies by the weight itself with trans_b)."""
        if self.store is None or self.compute_dtype != torch.bfloat16 or not torch.is_grad_enabled():
            return None
        return self.store.view_t(params)

    def bias(self, params: Sequence[Optional[nn.Parameter]]) -> Optional[torch.Tensor]:
        if any(p is None for p in params):
            return None
        if self.store is not None:
            v = self.store.view(params, torch.float32)
            if v is not None:
                return v
        bs = [p.data for p in params]
        return bs[0] if len(bs) == 1 else torch.cat(bs, dim=0)

    def grad_sink(self, params: Sequence[Optional[nn.Parameter]]) -> Optional[torch.Tensor]:
        if self.store is None or not self.direct_grads or any(p is None for p in params):
            return None
        return self.store.grad_view(params)

    def sinks(self, mapping) -> Optional[dict]:
        """{name: [params]} -> {name: flat-gradient view}, all or nothing (None when autograd should carry the
        parameter gradients instead: no flat store, grads not attached, frozen parameters, no_grad mode)."""
        if not torch.is_grad_enabled():
            return None
        out = {}
        for name, ps in mapping.items():
            v = self.grad_sink(ps)
            if v is None:
                return None
            out[name] = v
        if self.wgrad_queue is not None:
            out["_wq"] = self.wgrad_queue
        if self.grad_copies is not None:
            out["_copies"] = self.grad_copies
        return out

    def flush_wgrads(self, on_group_done=None):
        """Run the deferred weight-gradient products (call after backward, before anything reads the gradients)."""
        if self.wgrad_queue is not None:
            self.wgrad_queue.flush(on_group_done)

    def grads_ready(self, params):
        if self.on_grads_ready is not None:
            ps = [p for p in params if p is not None]
            q = self.wgrad_queue
            if q is not None and q.pending:
                q.after_flush.append(lambda: self.on_grads_ready(ps))  # their weight gradients are still queued
            else:
                self.on_grads_ready(ps)

    def act_in(self, x: torch.Tensor) -> torch.Tensor:
        """Bring an activation into the compute dtype (autocast-style entry cast)."""
        if x.dtype == self.compute_dtype:
            return x
        return ops.cast(x, self.compute_dtype)


_DEFAULT = Runtime()


def runtime_of(module: nn.Module) -> Runtime:
    return getattr(module, "_rt", None) or _DEFAULT


def install_runtime(root: nn.Module, rt: Runtime):
    for m in root.modules():
        object.__setattr__(m, "_rt", rt)
