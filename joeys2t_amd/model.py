"""Model container and builder with the reference's call surface (joeynmt/model.py): `Model.forward(return_type,
**vars(batch))` -> 4-tuple (:95-168), `DataParallelWrapper` (:323-363), `build_model(cfg, src_vocab, trg_vocab)`
(:366-506).  Only the S2T / Transformer path is provided (the reference asserts the same for S2T, :70-72).

What differs from the reference, by design:
  * `Model.finalize(device, compute_dtype)` moves all parameters into one flat HBM buffer (runtime.ParamStore)
    and binds the HIP runtime; forward on a CPU tensor or without libjoeys2t_hip.so raises.
  * log-softmax is fused into the loss kernels, so the loss modules receive logits (loss.py).
"""
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from joeys2t_amd import ops
from joeys2t_amd.builders import ConfigurationError
from joeys2t_amd.decoders import Decoder, TransformerDecoder
from joeys2t_amd.embeddings import Embeddings
from joeys2t_amd.encoders import ConformerEncoder, Encoder, TransformerEncoder
from joeys2t_amd.initialization import initialize_model
from joeys2t_amd.loss import XentCTCLoss, XentLoss
from joeys2t_amd.runtime import ParamStore, Runtime, install_runtime


def _mask_row_sums(src_mask: Tensor) -> Tensor:
    """src_mask.squeeze(1).sum(dim=1) (model.py:125); the sub-sampler leaves the numbers on the mask it built."""
    known = getattr(src_mask, "js2t_row_sums", None)
    return known if known is not None else src_mask.squeeze(1).sum(dim=1)


PACK_CTC = True  # the CTC branch of a ragged batch on packed encoder rows when the encoder packed them (tests flip this)


class Model(nn.Module):
    def __init__(self, encoder: Encoder, decoder: Decoder, src_embed: nn.Module, trg_embed: Embeddings, src_vocab,
                 trg_vocab, task: str = "S2T") -> None:
        super().__init__()
        self.src_embed = src_embed  # nn.Identity() for S2T
        self.trg_embed = trg_embed
        self.encoder = encoder
        self.decoder = decoder
        self.src_vocab = src_vocab
        self.trg_vocab = trg_vocab
        self.pad_index = trg_vocab.pad_index
        self.bos_index = trg_vocab.bos_index
        self.eos_index = trg_vocab.eos_index
        self.sep_index = trg_vocab.sep_index
        self.unk_index = trg_vocab.unk_index
        self.specials = [trg_vocab.lookup(t) for t in trg_vocab.specials]
        self.lang_tags = [trg_vocab.lookup(t) for t in trg_vocab.lang_tags]
        self._loss_function = None
        self.task = task
        if self.task == "S2T":
            assert isinstance(self.encoder, TransformerEncoder)
            assert isinstance(self.decoder, TransformerDecoder)
        self._rt_obj: Optional[Runtime] = None
        self._cut_tensor = None  # encoder output of the last training forward (the one tensor between the two halves of backward)
        self.overlap_ctc = False  # loss path: CTC branch on a second stream (TrainStep(overlap_ctc=True))

    # ------------------------------------------------------------------ HIP runtime binding
    def finalize(self, device, compute_dtype: torch.dtype = torch.float32, seed: int = 42) -> "Model":
        """Place the model on `device`, flatten its parameters into one HBM buffer and bind the kernel runtime."""
        device = torch.device(device)
        if device.type != "cuda":
            raise ops.Js2tError("Model.finalize: the HIP path needs a cuda (ROCm) device")
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise ops.Js2tError(f"compute dtype must be float32 or bfloat16, got {compute_dtype}")
        for name, buf in list(self.named_buffers()):
            mod = self
            *path, leaf = name.split(".")
            for part in path:
                mod = getattr(mod, part)
            mod._buffers[leaf] = buf.to(device)
        rt = Runtime(device, compute_dtype)
        rt.store = ParamStore(self, device)
        rt.rng.seed(seed)
        install_runtime(self, rt)
        object.__setattr__(self, "_rt_obj", rt)
        if compute_dtype == torch.bfloat16:
            rt.store.refresh(force=True)
        return self

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        """As nn.Module.load_state_dict (parameters stay views of the flat store); the compute-dtype shadows are re-derived
        before the next forward."""
        out = super().load_state_dict(state_dict, strict=strict, **kwargs)
        ops.WEIGHT_VERSION += 1
        if self._rt_obj is not None and self._rt_obj.store is not None:
            self._rt_obj.store.mark_dirty()
        return out

    @property
    def runtime(self) -> Runtime:
        if self._rt_obj is None:
            raise ops.Js2tError("Model.finalize(device, compute_dtype) must be called before use")
        return self._rt_obj

    def _prepare_step(self):
        rt = self.runtime
        if rt.compute_dtype == torch.bfloat16:
            # the bf16 shadow follows the fp32 master: re-cast before every training forward unless the optimizer
            # keeps it in sync itself (store.auto_refresh = False), and whenever the master was marked dirty
            rt.store.refresh(force=self.training and rt.store.auto_refresh)

    # ------------------------------------------------------------------ loss
    @property
    def loss_function(self):
        return self._loss_function

    @loss_function.setter
    def loss_function(self, cfg: Tuple):
        loss_type, label_smoothing, ctc_weight = cfg
        if loss_type == "crossentropy-ctc":
            loss_function = XentCTCLoss(pad_index=self.pad_index, bos_index=self.bos_index,  # bos -> blank
                                        smoothing=label_smoothing, ctc_weight=ctc_weight)
        elif loss_type == "crossentropy":
            loss_function = XentLoss(pad_index=self.pad_index, smoothing=label_smoothing)
            self.decoder.ctc_output_layer = None
        else:
            raise ConfigurationError(f"unknown loss type {loss_type}")
        self._loss_function = loss_function

    _ctc_pack = None  # set by _encode_decode(pack_ctc=True): the returned ctc logits are packed rows

    # ------------------------------------------------------------------ forward dispatch
    def forward(self, return_type: str = None, **kwargs) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        if return_type is None:
            raise ValueError("Please specify return_type: {`loss`, `loss_probs`, `encode`, `decode`, `decode_ctc`}.")
        self._prepare_step()
        if not torch.is_grad_enabled():
            # hand-over tables between residual blocks (LayerNorm statistics, dropout hints) are reset per training step by
            # TrainStep; inference forwards reset them here, so that no activation of an earlier call stays pinned in them
            from joeys2t_amd import functional as _Fn
            _Fn.reset_handover()
        if return_type.startswith("loss"):
            assert self.loss_function is not None
            assert "trg" in kwargs and "trg_mask" in kwargs
            lf = self.loss_function
            ctc_loss = None
            if (self.overlap_ctc and return_type == "loss" and lf.require_ctc_layer and kwargs["src"].is_cuda and
                    getattr(self.decoder, "ctc_output_layer", None) is not None):
                out, ctc_out, src_mask, ctc_loss = self._encode_decode_ctc_aside(lf, **kwargs)
            else:
                out, ctc_out, src_mask = self._encode_decode(pack_ctc=(return_type == "loss" and lf.require_ctc_layer), **kwargs)
            xent_loss, n_correct = lf.xent(out, kwargs["trg"])
            ret = [None, None, None, None]
            if lf.require_ctc_layer and isinstance(ctc_out, Tensor):
                if ctc_loss is None:
                    in_len = _mask_row_sums(src_mask)  # subsampled mask (model.py:125; loss.py:159)
                    ctc_loss = lf.ctc(ctc_out, kwargs["trg"], in_len, kwargs["trg_length"], pack=self._ctc_pack)
                else:  # computed on the side stream: join before the two losses meet
                    torch.cuda.current_stream().wait_stream(self.runtime.side_stream())
                    ctc_loss.record_stream(torch.cuda.current_stream())
                ret[0] = ops.LinComb2Fn.apply(xent_loss, ctc_loss, 1.0 - lf.ctc_weight, lf.ctc_weight)  # one launch
                ret[1], ret[2] = xent_loss, ctc_loss
            else:
                ret[0] = xent_loss
            ret[3] = n_correct.long()
            if return_type == "loss_probs":
                ret[1] = ops.log_softmax(out.detach())
                ret[2] = ops.log_softmax(ctc_out.detach()) if isinstance(ctc_out, Tensor) else None
            return tuple(ret)
        if return_type == "encode":
            encoder_output, encoder_hidden, src_mask = self._encode(**kwargs)
            return encoder_output, encoder_hidden, src_mask, None
        if return_type == "decode":
            kwargs.setdefault("compute_ctc", False)
            outputs, hidden, att_probs, att_vectors, _ = self._decode(**kwargs)
            return outputs, hidden, att_probs, att_vectors
        if return_type == "decode_ctc":
            outputs, hidden, att_probs, _, ctc_out = self._decode(**kwargs)
            return outputs, hidden, att_probs, ctc_out
        raise ValueError(f"unknown return_type {return_type}")

    def _ctc_packing(self, encoder_output):
        """The encoder's ops.PackedRows when the CTC branch can run on its packed rows (a ragged batch whose encoder packed them; bf16),
        else None: projection, row log-sum-exp, the recursions' inputs, the gradient and the projection's own gradients then stay on the
        live positions (13600 padded rows put the projection's input gradient - N = 512, K = 5000 - at 1.11 tiles per CU: two tile times)."""
        pk = getattr(self.encoder, "last_pack", None)
        if (PACK_CTC and pk is not None and self.runtime.compute_dtype == torch.bfloat16 and encoder_output.dim() == 3 and
                (pk.B, pk.T) == tuple(encoder_output.shape[:2])):
            return pk
        return None

    def _encode_decode(self, src: Tensor, trg_input: Tensor, src_mask: Tensor, src_length: Tensor, trg_mask: Tensor = None,
                       pack_ctc: bool = False, **kwargs):
        """pack_ctc: the caller only wants the CTC LOSS of the returned ctc logits (return_type "loss"): they may then come as the
        projection of the packed encoder rows, [1, rows, V], with `self._ctc_pack` saying so."""
        encoder_output, encoder_hidden, src_mask = self._encode(src=src, src_length=src_length, src_mask=src_mask, **kwargs)
        encoder_output = self._mark_cut(encoder_output)
        self._ctc_pack = self._ctc_packing(encoder_output) if pack_ctc else None
        decoder_output, _, _, _, ctc_output = self._decode(encoder_output=encoder_output, encoder_hidden=encoder_hidden,
                                                           src_mask=src_mask, trg_input=trg_input,
                                                           unroll_steps=trg_input.size(1), trg_mask=trg_mask,
                                                           memory_pack=getattr(self.encoder, "last_pack", None),
                                                           ctc_pack=self._ctc_pack, **kwargs)
        return decoder_output, ctc_output, src_mask

    def _encode_decode_ctc_aside(self, lf, src: Tensor, trg_input: Tensor, src_mask: Tensor, src_length: Tensor, trg_mask: Tensor = None,
                                 **kwargs):
        """_encode_decode with the CTC branch (projection of the encoder states + CTC loss, model.py:121-126 of the
        reference) on a second stream: it only needs the encoder output, its alpha/beta recursion keeps 32 of 256 CUs busy
        for ~0.25 ms and the decoder's kernels (2592 target rows) leave most of the chip idle as well.  Autograd runs the
        branch's backward on the same second stream.  Same kernels, same values."""
        encoder_output, encoder_hidden, src_mask = self._encode(src=src, src_length=src_length, src_mask=src_mask, **kwargs)
        encoder_output = self._mark_cut(encoder_output)
        cur, side = torch.cuda.current_stream(), self.runtime.side_stream()
        side.wait_stream(cur)
        pk = self._ctc_packing(encoder_output)  # ragged batch whose encoder ran on packed rows: so does the CTC branch
        with torch.cuda.stream(side):
            if pk is not None:
                from joeys2t_amd.functional import PackRowsFn
                enc_p = PackRowsFn.apply(self.runtime.act_in(encoder_output), pk)  # [1, rows, d]
                ctc_out = self.decoder.project(self.decoder.ctc_output_layer, enc_p, self.runtime.compute_dtype)  # [1, rows, V]
            else:
                ctc_out = self.decoder.project(self.decoder.ctc_output_layer, encoder_output, self.runtime.compute_dtype)
            in_len = _mask_row_sums(src_mask)
            ctc_loss = lf.ctc(ctc_out, kwargs["trg"], in_len, kwargs["trg_length"], pack=pk)
        for tns in (encoder_output, src_mask, kwargs["trg"], kwargs["trg_length"]):
            tns.record_stream(side)
        decoder_output, _, _, _, _ = self._decode(encoder_output=encoder_output, encoder_hidden=encoder_hidden, src_mask=src_mask,
                                                  trg_input=trg_input, unroll_steps=trg_input.size(1), trg_mask=trg_mask,
                                                  compute_ctc=False, memory_pack=getattr(self.encoder, "last_pack", None), **kwargs)
        return decoder_output, ctc_out, src_mask, ctc_loss

    def _mark_cut(self, encoder_output: Tensor) -> Tensor:
        """The encoder output of a training forward, passed through an identity node: the one tensor that connects the two halves
        of the backward pass, and where TrainStep may cut it (functional.CutFn says why the node is there)."""
        self._cut_tensor = None
        if torch.is_grad_enabled() and encoder_output.requires_grad:
            from joeys2t_amd.functional import CutFn
            encoder_output = self._cut_tensor = CutFn.apply(encoder_output)
        return encoder_output

    def _encode(self, src: Tensor, src_length: Tensor, src_mask: Tensor, **_kwargs):
        assert _kwargs.get("task", self.task) == self.task, (_kwargs.get("task"), self.task)
        if _kwargs.get("src_prompt_mask", None) is not None and isinstance(self.src_embed, Embeddings):
            _kwargs["src_prompt_mask"] = self.src_embed(_kwargs["src_prompt_mask"])
        return self.encoder(self.src_embed(src), src_length, src_mask, **_kwargs)

    def _decode(self, encoder_output: Tensor, encoder_hidden: Tensor, src_mask: Tensor, trg_input: Tensor,
                unroll_steps: int, decoder_hidden: Tensor = None, att_vector: Tensor = None, trg_mask: Tensor = None,
                **_kwargs):
        if _kwargs.get("trg_prompt_mask", None) is not None:
            assert self.sep_index is not None and self.sep_index in self.specials, "This model doesn't support prompting!"
            _kwargs["trg_prompt_mask"] = self.trg_embed(_kwargs["trg_prompt_mask"])
        return self.decoder(trg_embed=self.trg_embed(trg_input), encoder_output=encoder_output,
                            encoder_hidden=encoder_hidden, src_mask=src_mask, unroll_steps=unroll_steps,
                            hidden=decoder_hidden, prev_att_vector=att_vector, trg_mask=trg_mask, **_kwargs)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(task={self.task},\n\tencoder={self.encoder},\n\tdecoder={self.decoder},\n"
                f"\tsrc_embed={self.src_embed},\n\ttrg_embed={self.trg_embed},\n\tloss_function={self.loss_function})")


class DataParallelWrapper(nn.Module):
    """What TrainManager holds under DDP (contract of the reference's model.py:323-363): a shell around a data-parallel
    wrapper `module` whose own `.module` is the Model.  Attribute reads fall through shell -> wrapper -> Model, forward()
    goes through the wrapper (that is where the gradient exchange hooks in), and state_dict() / load_state_dict() talk to
    the Model directly, so checkpoints written under DDP carry un-prefixed keys."""

    def __init__(self, module: nn.Module):
        super().__init__()
        if not hasattr(module, "module"):
            raise AssertionError("DataParallelWrapper wraps a data-parallel module (something with a `.module`)")
        self.module = module

    @property
    def _model(self) -> nn.Module:
        return self._modules["module"].module

    def __getattr__(self, name):
        wrapper = self._modules["module"]
        for lookup in (lambda: nn.Module.__getattr__(self, name), lambda: getattr(wrapper, name), lambda: getattr(wrapper.module, name)):
            try:
                return lookup()
            except AttributeError:
                continue
        raise AttributeError(f"neither {type(self).__name__}, {type(wrapper).__name__} nor {type(wrapper.module).__name__} has {name!r}")

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        return self._model.state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return self._model.load_state_dict(*args, **kwargs)


def build_model(cfg: Dict = None, src_vocab=None, trg_vocab=None) -> Model:
    """Build and initialise the model from the `model` section of a JoeyNMT config (reference model.py:366-506)."""
    enc_cfg, dec_cfg = dict(cfg["encoder"]), dict(cfg["decoder"])
    task = "MT" if src_vocab is not None else "S2T"
    trg_pad_index = trg_vocab.pad_index
    if task == "MT":
        # text source (configs/transformer_small.yaml, the plumbing config): an embedding table instead of fbank frames, no
        # sub-sampler, no CTC head; everything behind the embedding is the same HIP path
        src_pad_index = src_vocab.pad_index
        src_embed = Embeddings(**enc_cfg["embeddings"], vocab_size=len(src_vocab), padding_idx=src_pad_index)
        if enc_cfg["embeddings"]["embedding_dim"] != enc_cfg["hidden_size"]:
            raise ConfigurationError("for transformer, emb_size must be the same as hidden_size.")
    else:
        src_pad_index = trg_pad_index
        src_embed = nn.Identity()
    if cfg.get("tied_embeddings", False):
        if task != "MT" or src_vocab != trg_vocab:
            raise ConfigurationError("Embedding cannot be tied since vocabularies differ.")
        trg_embed = src_embed
    else:
        trg_embed = Embeddings(**dec_cfg["embeddings"], vocab_size=len(trg_vocab), padding_idx=trg_pad_index)

    enc_type = enc_cfg.get("type", "transformer")
    if enc_type not in ("transformer", "conformer") or dec_cfg.get("type", "transformer") != "transformer":
        raise ConfigurationError("RNN model not supported for s2t task. use transformer.")
    enc_dropout = enc_cfg.get("dropout", 0.0)
    enc_emb_dropout = enc_cfg["embeddings"].get("dropout", enc_dropout)
    if enc_type == "conformer":
        # EXTENSION (BASELINE.json configs[4]): the reference's build_model accepts `recurrent` / `transformer` only
        # (model.py:417-421) although it ships the class (encoders.py:376-445).  Keys beside the transformer's:
        # `depthwise_conv_kernel_size` (default 31, transformer_layers.py:489) and `rel_pos_clip` (relative-position
        # attention bias, no counterpart in the reference); the sub-sampler is always on (encoders.py:431).
        if task != "S2T":
            raise ConfigurationError("conformer encoder: speech input only (it always sub-samples, encoders.py:431).")
        encoder = ConformerEncoder(**enc_cfg, emb_size=enc_cfg["embeddings"]["embedding_dim"], emb_dropout=enc_emb_dropout,
                                   pad_index=src_pad_index)
    else:
        encoder = TransformerEncoder(**enc_cfg, emb_size=enc_cfg["embeddings"]["embedding_dim"], emb_dropout=enc_emb_dropout,
                                     pad_index=src_pad_index)
    dec_dropout = dec_cfg.get("dropout", 0.0)
    dec_emb_dropout = dec_cfg["embeddings"].get("dropout", dec_dropout)
    if task == "S2T":
        dec_cfg["encoder_output_size_for_ctc"] = encoder.output_size
    decoder = TransformerDecoder(**dec_cfg, encoder=None, vocab_size=len(trg_vocab), emb_size=trg_embed.embedding_dim,
                                 emb_dropout=dec_emb_dropout)
    model = Model(encoder=encoder, decoder=decoder, src_embed=src_embed, trg_embed=trg_embed, src_vocab=src_vocab,
                  trg_vocab=trg_vocab, task=task)
    if cfg.get("tied_softmax", False):
        if trg_embed.lut.weight.shape == model.decoder.output_layer.weight.shape:
            model.decoder.output_layer.weight = trg_embed.lut.weight
        else:
            raise ConfigurationError("For tied_softmax, the decoder embedding_dim and decoder hidden_size must be the same.")
    initialize_model(model, cfg, src_pad_index, trg_pad_index)
    return model
