"""Feature store and loader for the reference's on-disk formats:

  * features as `.npy` files or inside an UNCOMPRESSED zip addressed as `name.zip:byte_offset:byte_size`
    (reference scripts/audiodata_utils.py:45-73, helpers_for_audio.py:70-122),
  * raw 16-bit PCM `.wav` files (the reference reads them with torchaudio; here the stdlib `wave` module),
  * the TSV manifest `id, src, n_frames, trg[, trg_prompt]` (datasets.py:583-630): rows with n_frames <= min_length or empty
    fields are dropped.

`SpeechFeatureLoader` turns manifest rows into padded training batches two ways: `features()` for pre-computed filterbanks
(the reference's path, host NumPy + padding with 1.0) and `waveform_batch()` which hands raw samples to the GPU front-end
(tokenizers.SpeechProcessor.batch_from_waveforms: fbank -> CMVN -> SpecAugment -> pad in HBM) - after the kernels are fast
the host loader bounds throughput, and a raw-waveform batch is 2x smaller than its 80-bin float features."""
import csv
import io
import wave
import zipfile
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np


def _is_npy_data(data: bytes) -> bool:
    return len(data) > 1 and data[0] == 147 and data[1] == 78  # b"\x93N" of the NumPy magic string


def read_wav(path: Path) -> Tuple[np.ndarray, int]:
    """16-bit PCM wav -> (float32 [channels, samples] in [-1, 1), sample rate), the layout torchaudio.load returns."""
    with wave.open(str(path), "rb") as f:
        if f.getsampwidth() != 2 or f.getcomptype() != "NONE":
            raise ValueError(f"{path}: only 16-bit PCM wav files are supported")
        nch, sr, n = f.getnchannels(), f.getframerate(), f.getnframes()
        pcm = np.frombuffer(f.readframes(n), dtype="<i2").reshape(-1, nch)
    return (pcm.astype(np.float32) / 32768.0).T.copy(), sr


def get_features(root_path: Path, fbank_path: str, extractor=None) -> np.ndarray:
    """Features [num_frames, num_freq] from `x.npy`, `x.zip:offset:size` or a wav file (helpers_for_audio.py:99-127).
    `extractor(waveform [C, N] float32, sample_rate) -> [T, F]` is required for wav input (the GPU fbank)."""
    _path, *extra = fbank_path.split(":")
    path = Path(root_path) / _path
    if not path.is_file():
        raise FileNotFoundError(f"File not found: {path}")
    if len(extra) == 0:
        if path.suffix == ".npy":
            features = np.load(path.as_posix())
        elif path.suffix == ".wav":
            if extractor is None:
                raise ValueError(f"{path}: a feature extractor is required for waveform input")
            features = extractor(*read_wav(path))
        else:
            raise ValueError(f"Invalid file type: {path}")
    elif len(extra) == 2:
        assert path.suffix == ".zip"
        offset, size = (int(i) for i in extra)
        with path.open("rb") as f:
            f.seek(offset)
            data = f.read(size)
        if not _is_npy_data(data):
            raise ValueError(f'Unknown file format for "{path}" [{offset}:{size}]')
        features = np.load(io.BytesIO(data))
    else:
        raise ValueError(f"Invalid path: {Path(root_path) / fbank_path}")
    assert features.ndim == 2, "spectrogram must be a 2-D array."
    return features


def create_zip(data_root: Path, zip_path: Path) -> None:
    """All `*.npy` of data_root into one STORED (uncompressed) zip, so members can be read by offset (audiodata_utils.py:64-73)."""
    with zipfile.ZipFile(zip_path, "w", zipfile.ZIP_STORED) as f:
        for path in sorted(Path(data_root).glob("*.npy")):
            f.write(path, arcname=path.name)


def get_zip_manifest(zip_path: Path) -> Dict[str, str]:
    """utt_id -> `zip_name:offset:size` (audiodata_utils.py:45-62): offset = local header offset + 30 + len(filename)."""
    zip_path = Path(zip_path)
    manifest = {}
    with zipfile.ZipFile(zip_path, mode="r") as f:
        info = f.infolist()
    with zip_path.open("rb") as f:
        for i in info:
            offset, size = i.header_offset + 30 + len(i.filename), i.file_size
            f.seek(offset)
            assert _is_npy_data(f.read(2)), (i.filename, size)
            manifest[Path(i.filename).stem] = f"{zip_path.name}:{offset}:{size}"
    return manifest


def read_tsv(file_path: Path, min_length: int = 0) -> List[Dict[str, str]]:
    """Rows of a JoeyS2T manifest (tab-separated, header line, no quoting, backslash escapes; datasets.py:590-612); rows whose
    n_frames is not above `min_length` or that have an empty field are dropped, `n_frames` comes back as int."""
    rows = []
    with open(file_path, encoding="utf-8", newline="") as f:
        reader = csv.DictReader(f, delimiter="\t", quoting=csv.QUOTE_NONE, escapechar="\\")
        assert "src" in reader.fieldnames, reader.fieldnames
        for r in reader:
            if any(v is None or not str(v).strip() for v in r.values()):
                continue
            r["n_frames"] = int(r["n_frames"])
            if r["n_frames"] <= min_length:
                continue
            rows.append(r)
    return rows


def save_tsv(rows: Iterable[Dict], path: Path, columns: Optional[List[str]] = None) -> None:
    rows = list(rows)
    columns = columns or list(rows[0].keys())
    with open(path, "w", encoding="utf-8", newline="") as f:
        w = csv.DictWriter(f, fieldnames=columns, delimiter="\t", quoting=csv.QUOTE_NONE, escapechar="\\", lineterminator="\n")
        w.writeheader()
        for r in rows:
            w.writerow({c: r[c] for c in columns})


class SpeechFeatureLoader:
    """Manifest rows -> model inputs.  `root`: directory the `src` column is relative to."""

    def __init__(self, tsv_path: Path, root: Optional[Path] = None, min_length: int = 0):
        self.tsv_path = Path(tsv_path)
        self.root = Path(root) if root is not None else self.tsv_path.parent
        self.rows = read_tsv(self.tsv_path, min_length)

    def __len__(self):
        return len(self.rows)

    def features(self, indices: Iterable[int], extractor=None) -> Tuple[np.ndarray, List[int]]:
        """Pre-computed features of the given rows, padded with 1.0 to the longest (helpers_for_audio.pad_features)."""
        feats = [get_features(self.root, self.rows[i]["src"], extractor).astype(np.float32) for i in indices]
        lengths = [f.shape[0] for f in feats]
        out = np.ones((len(feats), max(lengths), feats[0].shape[1]), dtype=np.float32)
        for b, f in enumerate(feats):
            out[b, :f.shape[0]] = f
        return out, lengths

    def waveform_batch(self, indices: Iterable[int]) -> Tuple[np.ndarray, List[int], int]:
        """Raw samples (first channel, as torchaudio's kaldi fbank takes it) of wav rows, zero-padded: float32 [B, Nmax],
        per-row sample counts, and the common sample rate."""
        waves, rates = [], set()
        for i in indices:
            w, sr = read_wav(self.root / self.rows[i]["src"])
            waves.append(w[0])
            rates.add(sr)
        if len(rates) != 1:
            raise ValueError(f"mixed sample rates {sorted(rates)}")
        n = [len(w) for w in waves]
        out = np.zeros((len(waves), max(n)), dtype=np.float32)
        for b, w in enumerate(waves):
            out[b, :len(w)] = w
        return out, n, rates.pop()
