"""RawBoost mirror (slsmi/rawboost.py) against vectors produced by the reference's own RawBoost.py and
dispatcher (tests/golden/make_golden_rawboost.py): same seed of numpy's global generator, same waveform.
Tolerance 1e-12 relative to the peak: the arithmetic is float64 numpy / scipy on both sides."""
import os
import types
import warnings

import numpy as np
import pytest

from slsmi import rawboost as rb


@pytest.fixture(scope="module")
def cases(golden_dir):
    return np.load(os.path.join(golden_dir, "rawboost_cases.npz"))


def default_args():
    return rb.RawBoostParams()


@pytest.mark.parametrize("seed", [1234, 99])
@pytest.mark.parametrize("algo", list(range(9)))
def test_process_rawboost_feature_matches_reference(cases, seed, algo):
    x = cases["x"]
    np.random.seed(seed)
    y = rb.process_Rawboost_feature(x.copy(), 16000, default_args(), algo)
    want = cases[f"y_seed{seed}_algo{algo}"]
    assert y.shape == want.shape
    assert np.max(np.abs(y - want)) <= 1e-12 * max(1.0, np.max(np.abs(want)))
    if algo == 0:
        assert np.array_equal(y, x)
    else:
        assert not np.allclose(y, x)


def test_helpers_match_reference(cases):
    np.random.seed(5)
    b = rb.genNotchCoeffs(5, 20, 8000, 100, 1000, 10, 100, 0, 0, 16000)
    assert b.shape == cases["notch_seed5"].shape
    np.testing.assert_allclose(b, cases["notch_seed5"], rtol=0, atol=1e-14)
    y = rb.filterFIR(cases["x"], b)
    assert y.shape == cases["x"].shape          # the group delay is cut off: same length as the input
    np.testing.assert_allclose(y, cases["fir_seed5"], rtol=0, atol=1e-13)


def test_argparse_namespace_and_draw_order():
    """Any object with the reference's attribute names serves as args; a draw consumes exactly one
    uniform number, so interleaved use stays in step with the reference's stream."""
    ns = types.SimpleNamespace(**rb.RawBoostParams().__dict__)
    x = np.sin(np.arange(2000) * 0.05)
    np.random.seed(3)
    a = rb.process_Rawboost_feature(x, 16000, ns, 3)
    np.random.seed(3)
    b = rb.process_Rawboost_feature(x, 16000, rb.RawBoostParams(), 3)
    assert np.array_equal(a, b)
    np.random.seed(11)
    v = rb.randRange(10, 100, 1)
    np.random.seed(11)
    assert v == int(np.random.uniform(10, 100, size=(1,))[0]) and isinstance(v, int)
    assert np.max(np.abs(rb.normWav(np.array([0.5, -2.0]), 0))) == 1.0
    assert np.array_equal(rb.normWav(np.array([0.5, -0.25]), 0), np.array([0.5, -0.25]))
    assert np.max(np.abs(rb.normWav(np.array([0.5, -0.25]), 1))) == 1.0


def test_train_dataset_item(tmp_path):
    """Dataset_ASVspoof2019_train (data_utils_SSL.py:68-93): decode, augment, crop / tile, label."""
    import torch
    from slsmi import data
    from flac_writer import encode_flac
    rng = np.random.default_rng(0)
    os.makedirs(tmp_path / "flac")
    pcm = (rng.standard_normal(30000) * 3000).astype(np.int16)
    open(tmp_path / "flac" / "LA_T_1.flac", "wb").write(encode_flac(pcm))
    ds = data.Dataset_ASVspoof2019_train(rb.RawBoostParams(), ["LA_T_1"], {"LA_T_1": 1}, str(tmp_path) + "/", algo=3)
    np.random.seed(4)
    x, y = ds[0]
    assert isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.shape == (64600,) and y == 1
    np.random.seed(4)
    clean = pcm.astype(np.float32) / 32768.0
    want = rb.process_Rawboost_feature(clean, 16000, rb.RawBoostParams(), 3)
    reps = 64600 // 30000 + 1
    np.testing.assert_allclose(x.numpy(), np.tile(want, reps)[:64600].astype(np.float32), atol=1e-6)
    ds0 = data.Dataset_ASVspoof2019_train(rb.RawBoostParams(), ["LA_T_1"], {"LA_T_1": 0}, str(tmp_path) + "/", algo=0)
    x0, y0 = ds0[0]
    assert y0 == 0 and np.allclose(x0.numpy()[:30000], clean, atol=1e-7)


class _Toy(__import__("torch").utils.data.Dataset):
    """Stands in for Dataset_ASVspoof2019_train: the item is the first number the worker's numpy generator gives."""

    def __len__(self):
        return 16

    def __getitem__(self, i):
        import torch
        return torch.tensor([float(i), float(np.random.rand())]), i % 2


def test_train_loader_shards_and_seeds_workers():
    """Two ranks see disjoint halves of an epoch; worker processes draw different augmentation streams;
    the same (seed, epoch, rank) repeats exactly."""
    from slsmi import training
    seen, draws = [], []
    for rank in (0, 1):
        items = [b for b in training.train_loader(_Toy(), batch_size=2, num_workers=2, rank=rank, world=2, seed=5)]
        assert len(items) == 4 and all(x.shape == (2, 2) for x, _ in items)
        seen.append(sorted(int(v) for x, _ in items for v in x[:, 0]))
        draws.append([float(v) for x, _ in items for v in x[:, 1]])
    assert not set(seen[0]) & set(seen[1]) and sorted(seen[0] + seen[1]) == list(range(16))
    assert len(set(draws[0] + draws[1])) == 16          # no two items share a random draw
    again = [float(v) for x, _ in training.train_loader(_Toy(), batch_size=2, num_workers=2, rank=1, world=2, seed=5)
             for v in x[:, 1]]
    assert again == draws[1]
