"""CPU: the host logic of detect_to_track/training.py -- the DataManager protocol of the reference
(data/types.py:44-68), per-rank sharding of an epoch (north_star: frame pairs shard across ranks with no exchange) and
the anchor grid (utils.py:114-163).  The step itself needs the HIP ops: tests/test_model_graph.py (-m gpu)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd" / "detect_to_track"))
import training  # noqa: E402  (the package __init__ would load the HIP library: not needed here)


def test_build_anchors_matches_reference_formulas():
    a = training.build_anchors((2, 3), [0.01, 0.04], [0.5, 2.0])
    assert a.shape == (2 * 3 * 4, 4) and a.dtype == np.float32
    # cell (1, 2): centre ((1 + .5) / 2, (2 + .5) / 3); area 0.04, ratio h/w = 2 -> h = sqrt(.08), w = .04 / h
    k = (1 * 3 + 2) * 4 + 3
    np.testing.assert_allclose(a[k], [0.75, 2.5 / 3, np.sqrt(0.08), 0.04 / np.sqrt(0.08)], rtol=1e-6)
    np.testing.assert_allclose(a[:, 2] * a[:, 3], np.tile([0.01, 0.01, 0.04, 0.04], 6), rtol=1e-6)   # areas
    np.testing.assert_allclose(a[:, 2] / a[:, 3], np.tile([0.5, 2.0, 0.5, 2.0], 6), rtol=1e-5)       # h / w


def test_manager_is_a_pure_function_of_seed_and_index():
    m = training.SyntheticPairManager(10, (32, 48), 90, 6, 3, 30, torch.device("cpu"), seed=4)
    assert len(m) == 10
    a, b = m[7], m[7]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert tuple(a.frames.shape) == (2, 3, 32, 48) and tuple(a.o_star.shape) == (2, 90) and tuple(a.b_star.shape) == (2, 90, 4)
    assert tuple(a.c_star.shape) == (12,) and int(a.c_star.max()) <= 30 and tuple(a.r_star.shape) == (12, 4)
    assert tuple(a.track_rois.shape) == (3, 4) and tuple(a.t_star.shape) == (3, 4)
    r = a.track_rois.numpy()                                             # tracked boxes lie inside the frame (ROIPool: 0/0 outside)
    assert (r[:, :2] - r[:, 2:] / 2 >= -1e-6).all() and (r[:, :2] + r[:, 2:] / 2 <= 1 + 1e-6).all()
    assert not torch.equal(m[6].frames, a.frames)
    try:
        m[10]
    except IndexError:
        pass
    else:
        raise AssertionError("index past the end must raise")


def test_ranks_walk_disjoint_shards_of_every_epoch():
    class Ids:                                                            # a DataManager that returns its index
        def __len__(self):
            return 37
        def __getitem__(self, i):
            return i
    world, bs = 4, 2
    loaders = [training.BatchLoader(Ids(), bs, r, world, seed=3) for r in range(world)]
    for epoch in range(2):
        seen = [sum((list(b) for b in ld), []) for ld in loaders]
        assert all(len(s) == (37 // world) // bs * bs for s in seen)     # drop_last, like the reference's BatchSampler
        flat = sum(seen, [])
        assert len(set(flat)) == len(flat)                               # no pair on two ranks
        assert all(len(ld) == 4 for ld in loaders)
    first = [b for b in training.BatchLoader(Ids(), bs, 0, world, seed=3)]
    again = [b for b in training.BatchLoader(Ids(), bs, 0, world, seed=3)]
    assert first == again                                                 # seeded: every rank derives the same permutation
