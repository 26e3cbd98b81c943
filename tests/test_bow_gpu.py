"""Row F3: vocabulary descent and ComputeDistinctiveDescriptors on the GPU vs the CPU oracle, on a synthetic tree of
ORBvoc's shape (the vocabulary file itself is not in the reference tree)."""
import numpy as np
import pytest

from motioncheck_ccm_slam_amd import synth
from motioncheck_ccm_slam_amd.matcher import ORBmatcher
from motioncheck_ccm_slam_amd.orb import ORBextractor
from motioncheck_ccm_slam_amd.vocabulary import ORBVocabulary, synthetic_tree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tree():
    return synthetic_tree(10, 4, seed=5, ragged=True)


def _features(tree, rng, n):
    par, desc, w = tree
    leaves = rng.integers(1, len(par), n)
    flips = np.packbits(rng.random((n, 256)) < 0.1, axis=1, bitorder="little")
    f = desc[leaves] ^ flips
    f[: n // 10] = rng.integers(0, 256, (n // 10, 32), dtype=np.uint8)
    return f


def test_transform_matches_oracle(ctx, oracle, tree):
    par, desc, w = tree
    voc = ORBVocabulary(10, 4, par, desc, w, ctx=ctx)
    ref = oracle.Voc(10, 4, par, desc, w)
    assert voc.size() == int((np.bincount(par[1:], minlength=len(par)) == 0)[1:].sum())
    feats = _features(tree, np.random.default_rng(0), 5000)
    for levelsup in (0, 2, 3, 4, 6):
        wid, ww, nid = voc.transform_features(feats, levelsup)
        rwid, rww, rnid = ref.transform_features(feats, levelsup)
        assert (wid == rwid).all() and (ww == rww).all() and (nid == rnid).all(), levelsup
    oid, oval, fv = voc.transform(feats[:1000], 2)
    r = oracle.bow_vector(*ref.transform_features(feats[:1000], 2))
    assert (oid == r[0]).all() and (oval == r[1]).all() and (fv == r[2]).all()
    assert (fv == -1).sum() > 0 and len(oid) > 300
    assert voc.transform_features(np.zeros((0, 32), np.uint8))[0].shape == (0,)


def test_degenerate_trees(ctx, oracle):
    d = np.zeros((1, 32), np.uint8)
    voc = ORBVocabulary(10, 6, [0], d, [0.0], ctx=ctx)                   # empty vocabulary: transform returns nothing
    wid, w, nid = voc.transform_features(np.ones((3, 32), np.uint8))
    assert (w == 0).all() and voc.size() == 0
    par, desc, ww = synthetic_tree(3, 7, seed=1, ragged=False)           # deep narrow tree, full
    voc = ORBVocabulary(3, 7, par, desc, ww, ctx=ctx)
    ref = oracle.Voc(3, 7, par, desc, ww)
    f = _features((par, desc, ww), np.random.default_rng(2), 700)
    a, b = voc.transform_features(f, 4), ref.transform_features(f, 4)
    assert all((x == y).all() for x, y in zip(a, b))


def test_compute_bow_feeds_search_by_bow(ctx, oracle, tree):
    """Frame::ComputeBoW -> SearchByBoW: extract two frames, transform both, match with the FeatureVector nodes."""
    par, desc, w = tree
    voc = ORBVocabulary(10, 4, par, desc, w, ctx=ctx)
    ref = oracle.Voc(10, 4, par, desc, w)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx)
    k1, d1 = ex(synth.frame(0)); k2, d2 = ex(synth.frame(1))
    _, _, fv1 = voc.transform(d1, 2); _, _, fv2 = voc.transform(d2, 2)
    assert (fv1 == oracle.bow_vector(*ref.transform_features(d1, 2))[2]).all()
    m = ORBmatcher(0.7, True, ctx=ctx)
    valid = np.ones(len(d1), np.uint8)
    n, m12 = m.SearchByBoW(d1, fv1, valid, k1["angle"], d2, fv2, k2["angle"])
    rn, r12 = oracle.match_bow(0.7, True, 50, False, d1, fv1, valid, k1["angle"], d2, fv2, None, k2["angle"])
    assert n == rn and (m12 == r12).all()


def test_full_batch_transform_properties(ctx, oracle, tree):
    """BASELINE config 2 size: 256 x 1000 descriptors in one call; spot-check against the oracle, determinism, and the
    invariant that the FeatureVector node is an ancestor of the word's node at the requested level."""
    par, desc, w = tree
    voc = ORBVocabulary(10, 4, par, desc, w, ctx=ctx)
    ref = oracle.Voc(10, 4, par, desc, w)
    feats = _features(tree, np.random.default_rng(9), 256000)
    wid, ww, nid = voc.transform_features(feats, 2)
    wid2, ww2, nid2 = voc.transform_features(feats, 2)
    assert (wid == wid2).all() and (nid == nid2).all()
    idx = np.random.default_rng(1).integers(0, 256000, 3000)
    r = ref.transform_features(feats[idx], 2)
    assert (wid[idx] == r[0]).all() and (ww[idx] == r[1]).all() and (nid[idx] == r[2]).all()
    depth = np.zeros(len(par), int)
    for i in range(1, len(par)):
        depth[i] = depth[par[i]] + 1
    assert ((depth[nid] == 2) | (nid == 0)).all()


def test_distinctive_descriptors(ctx, oracle, tree):
    par, desc, w = tree
    voc = ORBVocabulary(10, 4, par, desc, w, ctx=ctx)
    rng = np.random.default_rng(4)
    counts = np.concatenate([[0, 1, 2, 3, 64, 65, 130], rng.integers(2, 40, 300)]).astype("i4")
    first = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype("i8")
    total = int(counts.sum())
    base = rng.integers(0, 256, (len(counts), 32), dtype=np.uint8)
    d = np.zeros((total, 32), np.uint8)
    for p, (f, c) in enumerate(zip(first, counts)):
        flips = np.packbits(rng.random((c, 256)) < rng.uniform(0.02, 0.2), axis=1, bitorder="little")
        d[f:f + c] = base[p] ^ flips
    d[first[4]:first[4] + 5] = d[first[4]]                               # identical observations: ties
    best = voc.distinctive_descriptors(d, first, counts)
    for p, (f, c) in enumerate(zip(first, counts)):
        assert best[p] == oracle.distinctive_descriptor(d[f:f + c]), p
    assert best[0] == -1 and best[1] == 0
