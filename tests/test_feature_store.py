"""CPU: the reference's feature-store formats: npy-in-zip addressed by byte offset, wav reading, the TSV manifest
and its row filtering, padded batches."""
import wave

import numpy as np
import pytest

from joeys2t_amd.feature_store import (SpeechFeatureLoader, create_zip, get_features, get_zip_manifest, read_tsv, read_wav, save_tsv)
from joeys2t_amd.helpers_for_audio import get_n_frames


def _write_wav(path, samples, sr=16000, nch=1):
    with wave.open(str(path), "wb") as f:
        f.setnchannels(nch)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(samples.astype("<i2").tobytes())


def test_zip_manifest_offsets_and_features(tmp_path):
    rs = np.random.RandomState(0)
    feats = {f"utt{i}": rs.randn(50 + 7 * i, 80).astype(np.float32) for i in range(4)}
    (tmp_path / "npy").mkdir()
    for k, v in feats.items():
        np.save(tmp_path / "npy" / f"{k}.npy", v)
    create_zip(tmp_path / "npy", tmp_path / "fbank80.zip")
    manifest = get_zip_manifest(tmp_path / "fbank80.zip")
    assert set(manifest) == set(feats)
    for k, v in feats.items():
        name, off, size = manifest[k].split(":")
        assert name == "fbank80.zip" and int(size) == (tmp_path / "npy" / f"{k}.npy").stat().st_size
        assert np.array_equal(get_features(tmp_path, manifest[k]), v)                      # by byte offset
        assert np.array_equal(get_features(tmp_path / "npy", f"{k}.npy"), v)              # plain npy
    with pytest.raises(ValueError):
        get_features(tmp_path, f"fbank80.zip:{int(manifest['utt0'].split(':')[1]) + 1}:10")
    with pytest.raises(FileNotFoundError):
        get_features(tmp_path, "missing.npy")


def test_tsv_filtering_wav_and_batches(tmp_path):
    rs = np.random.RandomState(1)
    (tmp_path / "wav").mkdir()
    rows = []
    for i, n in enumerate([16000, 8000, 400, 24000]):
        pcm = (rs.randn(n) * 3000).clip(-32768, 32767)
        _write_wav(tmp_path / "wav" / f"u{i}.wav", pcm)
        rows.append({"id": f"u{i}", "src": f"wav/u{i}.wav", "n_frames": get_n_frames(n, 16000), "trg": f"text {i}" if i != 1 else " "})
    save_tsv(rows, tmp_path / "train.tsv")
    got = read_tsv(tmp_path / "train.tsv", min_length=10)
    # row 1 has an empty target, row 2 (400 samples -> 1 frame) is not above min_length: both dropped (datasets.py:604-609)
    assert [r["id"] for r in got] == ["u0", "u3"] and got[0]["n_frames"] == 98 and isinstance(got[0]["n_frames"], int)
    w, sr = read_wav(tmp_path / "wav" / "u0.wav")
    assert sr == 16000 and w.shape == (1, 16000) and w.dtype == np.float32 and np.abs(w).max() < 1.0
    loader = SpeechFeatureLoader(tmp_path / "train.tsv", min_length=10)
    batch, n, sr = loader.waveform_batch([0, 1])
    assert batch.shape == (2, 24000) and n == [16000, 24000] and sr == 16000 and np.all(batch[0, 16000:] == 0.0)
    fake = lambda wav, rate: np.full((get_n_frames(wav.shape[1], rate), 80), 0.5, dtype=np.float32)
    feats, lengths = loader.features([0, 1], extractor=fake)
    assert feats.shape == (2, 148, 80) and lengths == [98, 148]
    assert np.all(feats[0, 98:] == 1.0) and np.all(feats[0, :98] == 0.5)  # padding value 1.0 (helpers_for_audio.py:151-152)
    # two channels: channel-major array like torchaudio.load
    _write_wav(tmp_path / "st.wav", np.stack([np.arange(10), -np.arange(10)], 1).reshape(-1), nch=2)
    w2, _ = read_wav(tmp_path / "st.wav")
    assert w2.shape == (2, 10) and w2[0, 3] == pytest.approx(3 / 32768) and w2[1, 3] == pytest.approx(-3 / 32768)
