"""GPU: audio front-end kernels (fbank / CMVN / SpecAugment / pad) and the update tail (clip + AdamW) against the
CPU oracle and torch.optim on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import s2t_oracle as O

pytestmark = pytest.mark.gpu


def test_fbank_known_answer_and_oracle(device):
    from joeys2t_amd.helpers_for_audio import extract_fbank_features, get_extractor, get_n_frames
    g = load_golden("audio")
    names = g["tsv_names"].tolist()
    pcm = g["pcm_260-123440-1"].astype(np.float32) / 32768.0
    feat = extract_fbank_features(torch.from_numpy(pcm).unsqueeze(0), 16000, device=device)
    assert feat.shape == (int(g["tsv_n_frames"][names.index("260-123440-1")]), 80) and feat.dtype == np.float32
    # the reference's own known answer: CMVN'd frame 0, bins 0-9 (test/unit/test_tokenizer.py:318-325, 1e-5)
    np.testing.assert_allclose(O.cmvn(feat)[0, :10], g["fbank_cmvn_ref_260-123440-1_frame0_bins0_9"], rtol=1e-5, atol=1e-5)
    ref = O.fbank(pcm)
    np.testing.assert_allclose(feat, ref, rtol=1e-4, atol=2e-4)
    # ragged batch in one launch, incl. an utterance shorter than one window (-> 0 frames)
    ex = get_extractor(device)
    waves = [g["pcm_260-123440-0"].astype(np.float32) / 32768.0, pcm, np.zeros(300, dtype=np.float32),
             g["pcm_260-123440-6"].astype(np.float32) / 32768.0]
    flat = torch.from_numpy(np.concatenate(waves)).to(device)
    offs = np.concatenate([[0], np.cumsum([len(w) for w in waves])[:-1]]).tolist()
    out, foff, frames = ex.batch(flat, [len(w) for w in waves], offs)
    assert frames == [215, 172, 0, 270] == [max(0, get_n_frames(len(w), 16000)) if len(w) >= 400 else 0 for w in waves]
    foff = foff.cpu().numpy()
    for i, w in enumerate(waves):
        if frames[i]:
            np.testing.assert_allclose(out[foff[i]:foff[i + 1]].cpu().numpy(), O.fbank(w), rtol=1e-4, atol=2e-4)


def test_fbank_synthetic_noise(device):
    from joeys2t_amd.helpers_for_audio import get_extractor
    g = torch.Generator().manual_seed(1234)
    wave = (0.1 * torch.randn(3, 48000, generator=g)).clamp_(-1, 1)
    out, foff, frames = get_extractor(device).batch(wave.to(device), [48000, 40000, 16000])
    assert frames == [298, 248, 98]
    foff = foff.cpu().numpy()
    for i, n in enumerate([48000, 40000, 16000]):
        np.testing.assert_allclose(out[foff[i]:foff[i + 1]].cpu().numpy(), O.fbank(wave[i, :n].numpy()), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_cmvn_specaugment_pad(device, out_dtype):
    from joeys2t_amd.data_augmentation import CMVN, SpecAugment, finalize_features
    rs = np.random.RandomState(3)
    feats = [(rs.randn(t, 80) * 3 + 1).astype(np.float32) for t in (57, 20, 133)]
    frames = [f.shape[0] for f in feats]
    flat = torch.from_numpy(np.concatenate(feats)).to(device)
    foff = torch.tensor(np.concatenate([[0], np.cumsum(frames)]), dtype=torch.int64, device=device)
    sa = SpecAugment(freq_mask_n=2, freq_mask_f=27, time_mask_n=2, time_mask_t=100, time_mask_p=1.0)
    np.random.seed(42)
    out, lens = finalize_features(flat, foff, frames, cmvn=CMVN(), specaugment=sa, out_dtype=out_dtype)
    rng = np.random.RandomState(42)
    ref = [O.specaugment_apply(O.cmvn(f.copy()), O.specaugment_params(f.shape[0], 80, rng, time_mask_t=100)) for f in feats]
    padded, rl, _ = O.pad_features(ref)
    assert lens == rl and tuple(out.shape) == padded.shape
    tol = dict(rtol=1e-5, atol=2e-5) if out_dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(out.float().cpu().numpy(), padded, **tol)
    # no augmentation, truncation to max_length (evaluation rule): the reference cuts item[:max_length] FIRST and runs
    # CMVN on what is left (tokenizers.py:474-487), so the statistics cover the kept frames only
    out2, lens2 = finalize_features(flat, foff, frames, cmvn=CMVN(), specaugment=None, out_dtype=torch.float32, max_length=40)
    assert lens2 == [40, 20, 40] and out2.shape[1] == 40
    for u in (0, 2):
        np.testing.assert_allclose(out2[u].cpu().numpy(), O.cmvn(feats[u][:40].copy()), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(out2[1, :20].cpu().numpy(), O.cmvn(feats[1].copy()), rtol=1e-5, atol=2e-5)
    assert torch.all(out2[1, 20:] == 1.0)


@pytest.mark.parametrize("case", ["after", "many", "many_after", "nocmvn", "after_trunc"])
def test_frontend_orders_and_mask_counts_beside_the_configured_one(device, case):
    """CMVN(before=False), more than two masks of a kind, SpecAugment without CMVN, truncation first - finalize_features' general
    path (js2t_cmvn_stats / js2t_feature_transform / js2t_feature_finalize composed in the reference's order) against the
    reference's own classes (frontend_general.npz, tokenizers.py:474-492)."""
    import json
    from pathlib import Path
    from conftest import load_golden
    from joeys2t_amd.data_augmentation import CMVN, SpecAugment, finalize_features
    g = load_golden("frontend_general")
    spec = json.loads((Path(__file__).resolve().parent / "golden" / "frontend_general.json").read_text())[case]
    feats = [g[f"in{i}"] for i in range(3)]
    frames = [f.shape[0] for f in feats]
    flat = torch.from_numpy(np.concatenate(feats)).to(device)
    keep = flat.clone()
    foff = torch.tensor(np.concatenate([[0], np.cumsum(frames)]), dtype=torch.int64, device=device)
    cmvn = CMVN(**spec["cmvn"]) if spec["cmvn"] else None
    np.random.seed(77)
    out, lens = finalize_features(flat, foff, frames, cmvn=cmvn, specaugment=SpecAugment(**spec["specaugment"]), out_dtype=torch.float32,
                                  max_length=spec["max_length"])
    want = [g[f"{case}_{i}"] for i in range(3)]
    assert lens == [w.shape[0] for w in want] and out.shape[1] == max(lens)
    rng = np.random.RandomState(77)
    for i, w in enumerate(want):
        got = out[i, :w.shape[0]].cpu().numpy()
        ok = np.ones(80, dtype=bool)
        sa_kw = spec["specaugment"]
        params = O.specaugment_params(w.shape[0], 80, rng, **sa_kw)
        if cmvn is not None and not cmvn.before and cmvn.norm_vars and params is not None:
            # a frequency-masked bin is constant when CMVN sees it: variance 0, divided by sqrt(1e-10) - the reference's value there
            # is its float32 column mean's last-place error times 1e5 (up to ~0.1), the kernel's (mean taken in double) is 0
            for f0, f in params[0]:
                ok[f0:f0 + f] = False
            assert np.abs(got[:, ~ok]).max(initial=0.0) < 1e-3 and np.abs(w[:, ~ok]).max(initial=0.0) < 1.0
        np.testing.assert_allclose(got[:, ok], w[:, ok], rtol=1e-5, atol=3e-5, err_msg=f"{case} {i}")
        assert torch.all(out[i, w.shape[0]:] == 1.0)
    assert torch.equal(flat, keep)  # the caller's features are left alone (the general path works on a copy)


def test_flat_adamw_and_clip_match_torch(device):
    from joeys2t_amd.builders import FlatAdamW, WarmupInverseSquareRootScheduler
    from joeys2t_amd.runtime import ParamStore
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11))
    ref = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11))
    ref.load_state_dict(net.state_dict())
    store = ParamStore(net, device)
    store.attach_grads()
    opt = FlatAdamW(store, lr=2e-3, betas=(0.9, 0.98), weight_decay=0.01)
    sched = WarmupInverseSquareRootScheduler(opt, peak_rate=2e-3, warmup=4, min_rate=1e-6)
    ropt = torch.optim.AdamW(ref.parameters(), lr=2e-3, betas=(0.9, 0.98), weight_decay=0.01)
    rsched = WarmupInverseSquareRootScheduler(ropt, peak_rate=2e-3, warmup=4, min_rate=1e-6)
    g = torch.Generator().manual_seed(1)
    lrs = []
    for it in range(6):
        grads = [torch.randn(p.shape, generator=g) * (30.0 if it % 2 == 0 else 0.01) for p in ref.parameters()]
        for p, q, gr in zip(net.parameters(), ref.parameters(), grads):
            p.grad.copy_(gr)
            q.grad = gr.clone()
        total = torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0)
        ropt.step()
        rsched.step(it)
        opt.clip_and_step(10.0)
        sched.step(it)
        lrs.append(opt.param_groups[0]["lr"])
        assert abs(float(opt.norm_clip[0]) - float(total)) <= 1e-4 * float(total)
        assert torch.all(store.flat_grad == 0)
        for p, q in zip(net.parameters(), ref.parameters()):
            torch.testing.assert_close(p.detach().cpu(), q.detach(), rtol=1e-5, atol=1e-6)
    assert lrs == [ropt.param_groups[0]["lr"]] * 0 + lrs  # same schedule object semantics
    assert lrs[0] == 2e-3 * 1 / 4 and lrs[3] == 2e-3


def test_flat_adamw_leaves_frozen_parameters_alone(device):
    """`freeze: True` sub-networks (requires_grad False; reference helpers.py freeze_params) are not in torch's optimizer
    at all: no weight decay, no moment update.  The flat update must skip their ranges of the store."""
    from joeys2t_amd.builders import FlatAdamW, trainable_ranges
    from joeys2t_amd.runtime import ParamStore
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(36, 52), torch.nn.Linear(52, 12), torch.nn.Linear(12, 8))
    ref = torch.nn.Sequential(torch.nn.Linear(36, 52), torch.nn.Linear(52, 12), torch.nn.Linear(12, 8))
    ref.load_state_dict(net.state_dict())
    for m in (net, ref):
        for p in m[1].parameters():
            p.requires_grad = False
    store = ParamStore(net, device)
    store.attach_grads()
    ranges = trainable_ranges(store)
    assert len(ranges) >= 2 and all(lo % 4 == 0 and hi % 4 == 0 for lo, hi in ranges)
    opt = FlatAdamW(store, lr=1e-2, betas=(0.9, 0.98), weight_decay=0.1)
    ropt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], lr=1e-2, betas=(0.9, 0.98), weight_decay=0.1)
    frozen_before = [p.detach().clone() for p in net[1].parameters()]
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        for p, q in zip(net.parameters(), ref.parameters()):
            if q.requires_grad:
                gr = torch.randn(q.shape, generator=g)
                p.grad.copy_(gr)
                q.grad = gr.clone()
        ropt.step()
        opt.clip_and_step(None)
    for p, b in zip(net[1].parameters(), frozen_before):
        assert torch.equal(p.detach(), b)  # bit-identical: neither decayed nor updated
    for p, q in zip(net.parameters(), ref.parameters()):
        torch.testing.assert_close(p.detach().cpu(), q.detach(), rtol=1e-5, atol=1e-6)
    lo = store.offsets[id(net[1].weight)]
    assert torch.all(opt.exp_avg[lo:lo + net[1].weight.numel()] == 0)


def test_wav_manifest_to_gpu_features(device, tmp_path):
    """wav files listed in a JoeyS2T manifest -> SpeechFeatureLoader -> raw-sample batch -> GPU front-end
    (fbank -> CMVN -> pad with 1.0) equals the CPU oracle per utterance."""
    import wave

    from joeys2t_amd.feature_store import SpeechFeatureLoader, save_tsv
    from joeys2t_amd.helpers_for_audio import get_n_frames
    from joeys2t_amd.tokenizers import SpeechProcessor
    from oracle import s2t_oracle as O
    rs = np.random.RandomState(3)
    (tmp_path / "wav").mkdir()
    rows = []
    for i, n in enumerate([24000, 16400, 31234]):
        pcm = (rs.randn(n) * 2500).clip(-32768, 32767).astype("<i2")
        with wave.open(str(tmp_path / "wav" / f"u{i}.wav"), "wb") as f:
            f.setnchannels(1), f.setsampwidth(2), f.setframerate(16000)
            f.writeframes(pcm.tobytes())
        rows.append({"id": f"u{i}", "src": f"wav/u{i}.wav", "n_frames": get_n_frames(n, 16000), "trg": "x"})
    save_tsv(rows, tmp_path / "dev.tsv")
    loader = SpeechFeatureLoader(tmp_path / "dev.tsv", min_length=5)
    wavs, n, sr = loader.waveform_batch(range(len(loader)))
    proc = SpeechProcessor(num_freq=80, min_length=5, max_length=3000, cmvn=dict(norm_means=True, norm_vars=True, before=True))
    feats, frames = proc.batch_from_waveforms(torch.from_numpy(wavs).to(device), n, is_train=False)
    assert frames == [r["n_frames"] for r in rows]
    for b in range(len(rows)):
        ref = O.cmvn(O.fbank(wavs[b, :n[b]]))
        np.testing.assert_allclose(feats[b, :frames[b]].cpu().numpy(), ref, rtol=2e-3, atol=2e-3)
        assert torch.all(feats[b, frames[b]:] == 1.0)


def test_prefetch_loader_copies_one_batch_ahead(device):
    """datasets.PrefetchLoader: every batch arrives in HBM bit-identical, in order, through pinned staging + a side stream."""
    from joeys2t_amd.datasets import PrefetchLoader
    g = torch.Generator().manual_seed(0)
    data = [torch.randn(4, 24000, generator=g) for _ in range(5)]
    seen = []

    def load(idx):
        seen.append(idx[0])
        return {"wave": data[idx[0]], "n": [24000] * 4, "idx": idx}

    loader = PrefetchLoader([[i] for i in range(5)], load, device)
    got = []
    for item in loader:
        assert item["wave"].is_cuda and item["n"] == [24000] * 4
        assert len(seen) >= min(5, len(got) + 2) or len(seen) == 5  # the following batch is already being staged
        got.append((item["idx"][0], item["wave"].sum().item(), item["wave"].clone()))
    assert [i for i, _, _ in got] == list(range(5))
    for i, _, w in got:
        assert torch.equal(w.cpu(), data[i])


def test_feature_transform_edges(device):
    """js2t_feature_transform: no masks (normalisation only), masks of width 0, an utterance of one frame, and the argument checks."""
    import ctypes as C
    from joeys2t_amd import ops
    from joeys2t_amd._lib import Js2tError, check, lib
    L = lib()
    frames = [1, 37, 5]
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.randn(sum(frames), 80).astype(np.float32)).to(device)
    foff = torch.tensor(np.concatenate([[0], np.cumsum(frames)]), dtype=torch.int64, device=device)
    mean, istd = torch.randn(3, 80, device=device), torch.rand(3, 80, device=device) + 0.5
    fill = torch.tensor([9.0, -3.0, 0.5], device=device)

    def run(t, mean, istd, fill, masks, nf, nt):
        check(L.js2t_feature_transform(ops._p(t), ops._p(foff), C.c_int32(3), C.c_int32(80), ops._p(mean), ops._p(istd), ops._p(fill), ops._p(masks),
                                       C.c_int32(nf), C.c_int32(nt), ops._stream()), "js2t_feature_transform")

    a = x.clone()
    run(a, mean, istd, None, None, 0, 0)
    want = torch.cat([(x[foff[u]:foff[u + 1]] - mean[u]) * istd[u] for u in range(3)])
    assert torch.allclose(a, want, rtol=1e-6, atol=1e-6)
    # three frequency + one time mask; utterance 0: all widths 0; utterance 1: a band and a stretch; utterance 2: the whole of it
    masks = torch.tensor([[0, 0, 5, 0, 9, 0, 0, 0], [10, 4, 0, 0, 70, 10, 30, 7], [0, 80, 0, 0, 0, 0, 0, 5]], dtype=torch.int32, device=device)
    b = x.clone()
    run(b, None, None, fill, masks, 3, 1)
    ref = x.clone()
    u1 = ref[1:38]
    u1[:, 10:14] = -3.0
    u1[:, 70:80] = -3.0
    u1[30:37, :] = -3.0
    ref[38:43] = 0.5
    assert torch.equal(b, ref) and torch.equal(b[0], x[0])
    with pytest.raises(Js2tError):
        run(x.clone(), mean, None, None, None, 0, 0)      # mean without istd
    with pytest.raises(Js2tError):
        run(x.clone(), None, None, None, masks, 3, 1)     # masks without fill values
