"""GPU: MPO two-site compression sweeps (pytdscf_amd/mpo_compress.py: device SVD / QR) against the
reference's own outputs (tests/golden/mpo_compress.npz, produced by tests/golden/make_golden_mpo.py from
_mpo_cls.py's merge_mpos_twodot / sweep_qr / sweep_compress_twodot / _compress_block_by_block).
Compared: bond dimensions (exactly) and the represented operator on 600 sampled grid points (1e-10
relative to its largest sampled value; the truncated variants to the reference's truncated values)."""

import numpy as np
import pytest

from pytdscf_amd import mpo_compress as mc


def _inputs(g):
    nterm, nsite = int(g["nterm"]), int(g["nsite"])
    return [[g[f"in_{k}_{p}"].copy() for p in range(nsite)] for k in range(nterm)]


def _sample(cores, idx):
    t = cores[0]
    for c in cores[1:]:
        t = np.tensordot(t, c, axes=([-1], [0]))
    t = t.reshape(t.shape[1:-1])
    return t[tuple(idx.T)]


def test_guess_bond_dimension_matches_reference(golden):
    g = golden("mpo_compress.npz")
    got = [mc.guess_bond_dimension(g["gbd_svals"], r) for r in (0.5, 0.9, 0.99, 0.999999, 1.0)]
    assert got == g["gbd_ranks"].tolist()
    with pytest.raises(ValueError):
        mc.guess_bond_dimension(g["gbd_svals"], 1.5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,rate", [("tight", 0.999999999999), ("loose", 0.99999)])
def test_merge_canonicalise_compress(golden, tag, rate):
    g = golden("mpo_compress.npz")
    idx = g["sample_idx"]
    scale = np.abs(g["dense_sum"]).max()
    merged = mc.merge_mpos_twodot(_inputs(g), rate=rate)
    assert [c.shape[2] for c in merged[:-1]] == g[f"merge_{tag}_bonds"].tolist()
    assert np.abs(_sample(merged, idx) - g[f"merge_{tag}_dense"]).max() < 1e-10 * scale
    if tag == "tight":  # nothing was cut: the sum of the terms itself
        assert np.abs(_sample(merged, idx) - g["dense_sum"]).max() < 1e-9 * scale
    canon = mc.sweep_qr([c.copy() for c in merged])
    assert np.abs(_sample(canon, idx) - g[f"qr_{tag}_dense"]).max() < 1e-10 * scale
    for p, c in enumerate(canon[:-1]):  # left-orthogonal up to the sqrt(norm) factors the sweep spreads
        m = c.reshape(-1, c.shape[2])
        gram = m.T @ m
        assert np.abs(gram - np.diag(np.diag(gram))).max() < 1e-10 * np.abs(gram).max()
    comp = mc.sweep_compress_twodot([c.copy() for c in canon], rate=rate, left_to_right=False)
    assert [c.shape[2] for c in comp[:-1]] == g[f"comp_{tag}_bonds"].tolist()
    assert np.abs(_sample(comp, idx) - g[f"comp_{tag}_dense"]).max() < 1e-10 * scale
    comp2 = mc.sweep_compress_twodot([c.copy() for c in comp], rate=rate, left_to_right=True)
    assert [c.shape[2] for c in comp2[:-1]] == g[f"comp2_{tag}_bonds"].tolist()
    assert np.abs(_sample(comp2, idx) - g[f"comp2_{tag}_dense"]).max() < 1e-10 * scale


@pytest.mark.gpu
def test_compress_block_by_block_and_lq(golden):
    g = golden("mpo_compress.npz")
    idx = g["sample_idx"]
    scale = np.abs(g["dense_sum"]).max()
    blk = mc.compress_block_by_block(_inputs(g), 0.999999999, 1, 1000)
    assert [c.shape[2] for c in blk[:-1]] == g["block_bonds"].tolist()
    assert np.abs(_sample(blk, idx) - g["block_dense"]).max() < 1e-10 * scale
    # groups of 3 terms exercise the outer merge of already-compressed groups
    blk3 = mc.compress_block_by_block(_inputs(g), 0.999999999, 1, 1000, sub_mpo=3)
    assert np.abs(_sample(blk3, idx) - g["dense_sum"]).max() < 1e-7 * scale
    # right-orthogonalisation keeps the operator and makes the cores row-orthogonal
    lq = mc.sweep_lq([c.copy() for c in blk])
    assert np.abs(_sample(lq, idx) - g["block_dense"]).max() < 1e-10 * scale
    for c in lq[1:]:
        m = c.reshape(c.shape[0], -1)
        gram = m @ m.T
        assert np.abs(gram - np.diag(np.diag(gram))).max() < 1e-10 * np.abs(gram).max()


@pytest.mark.gpu
def test_complex_cores_round_trip():
    rng = np.random.default_rng(5)
    terms = [[rng.standard_normal((1 if p == 0 else 2, 5, 1 if p == 3 else 2)) + 1j * rng.standard_normal((1 if p == 0 else 2, 5, 1 if p == 3 else 2))
              for p in range(4)] for _ in range(3)]
    idx = rng.integers(0, 5, size=(200, 4))
    want = sum(_sample(t, idx) for t in terms)
    got = mc.compress_block_by_block([[c.copy() for c in t] for t in terms], 0.999999999999, 1)
    assert np.abs(_sample(got, idx) - want).max() < 1e-10 * np.abs(want).max()
