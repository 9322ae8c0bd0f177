"""Feature-generation job (transform_clips caller side): memmap layout + trimming on CPU, GPU content vs oracle."""
import os

import numpy as np
import pytest

from nanowakeword_amd.features import generate_features, trim_mmap
from nanowakeword_amd.synth import synth_pcm


class FakeExtractor:
    def get_embedding_shape(self, sec, sr=16000):
        return (1 + int(sec * sr) // 160, 4)

    def embed_clips(self, x, batch_size=128, ncpu=1):
        f = self.get_embedding_shape(x.shape[1] / 16000)
        m = np.abs(x.astype(np.float32)).mean(axis=1)
        return np.broadcast_to(m[:, None, None], (x.shape[0],) + f).astype(np.float32).copy()


def test_memmap_layout_and_trim(tmp_path):
    path = os.path.join(tmp_path, "feats.npy")
    batches = [synth_pcm("noise", 8, 16000, seed=i) for i in range(3)]           # 24 clips available
    rows = generate_features(iter(batches), total_clips=40, output_path=path, extractor=FakeExtractor(), clip_seconds=1.0)
    assert rows == 24                                                            # 16 unwritten (zero) rows trimmed
    a = np.load(path, mmap_mode="r")
    assert a.shape == (24, 101, 4) and a.dtype == np.float32
    want = np.concatenate([FakeExtractor().embed_clips(b) for b in batches])
    assert np.array_equal(np.asarray(a), want)
    assert generate_features(iter(batches), 40, path, FakeExtractor(), 1.0, overwrite=False) is None
    rows = generate_features(iter(batches), total_clips=10, output_path=path, extractor=FakeExtractor(), clip_seconds=1.0)
    assert rows == 10 and np.load(path, mmap_mode="r").shape == (10, 101, 4)   # capped at total_clips
    # interior zero rows are kept, only the trailing run is dropped
    b = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(6, 2, 2))
    b[0] = 1; b[2] = 3; b.flush(); del b
    assert trim_mmap(path) == 3 and np.load(path).shape == (3, 2, 2)


@pytest.mark.gpu
def test_hip_features_job_matches_oracle(tmp_path, golden_frontend):
    import oracle
    from nanowakeword_amd.config import FrontendConfig
    from nanowakeword_amd.features import HipFeatures
    from parity import assert_frontend_close
    g = golden_frontend
    ex = HipFeatures(FrontendConfig(), window=g["window"], mel_fb=g["fb64"])
    assert ex.get_embedding_shape(1.0) == (101, 64)
    batches = [synth_pcm("speechlike", 32, 16000, seed=i) for i in range(4)]
    path = os.path.join(tmp_path, "positive_features_train.npy")
    rows = generate_features(iter(batches), total_clips=200, output_path=path, extractor=ex, clip_seconds=1.0)
    a = np.load(path, mmap_mode="r")
    assert rows == 128 and a.shape == (128, 101, 64)
    pcm = np.concatenate(batches)
    mel = oracle.mel_power(pcm, g["window"], g["fb64"])
    db = oracle.logmel_db(mel)
    got = np.asarray(a).transpose(0, 2, 1)
    assert np.abs(got - db)[mel > 1e-4 * mel.max(axis=1, keepdims=True)].max() <= 1e-4
    f = ex.embed_clips(pcm[:5], batch_size=2)                                    # ragged batching
    assert np.array_equal(f, np.asarray(a[:5]))
    with pytest.raises(ValueError):
        ex.embed_clips(np.zeros((2, 399), np.int16))
    ex.close()
