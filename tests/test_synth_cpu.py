"""The synthetic generators are deterministic and their torch twin is bit-identical."""
import numpy as np

from motioncheck_ccm_slam_amd import synth


def test_descriptor_pairs_torch_equals_numpy():
    import torch
    a, b = synth.descriptor_pairs(7, 5)
    ta, tb = synth.descriptor_pairs_torch(7, 5, device="cpu", batch=2)
    assert (ta.numpy() == a).all() and (tb.numpy() == b).all()
    ta, tb = synth.descriptor_pairs_torch(9998, 2, device="cpu")
    a, b = synth.descriptor_pairs(9998, 2)
    assert (ta.numpy() == a).all() and (tb.numpy() == b).all()


def test_frames_are_deterministic():
    f = synth.frame(3)
    assert f.shape == (480, 752) and f.dtype == np.uint8 and (f == synth.frame(3)).all() and (f != synth.frame(4)).any()
