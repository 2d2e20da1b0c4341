"""A dict-backed stand-in for the few h5py calls the data readers make (h5py
is not installed in the build image).  Test infrastructure only: the fixture
generator hands it to the reference's readers as ``h5py.File`` and the parity
test hands the same tree to ``tike_amd.ptycho.io``."""
import numpy as np


class Dataset:
    def __init__(self, value, attrs=None):
        self._a = np.asarray(value)
        self.attrs = dict(attrs or {})

    shape = property(lambda self: self._a.shape)
    dtype = property(lambda self: self._a.dtype)

    def __len__(self):
        return len(self._a)

    def __getitem__(self, key):
        if isinstance(key, tuple) and key == ():
            return self._a[()] if self._a.ndim == 0 else self._a.copy()
        return np.array(self._a[key])  # a read is a copy, as with h5py


class File:
    """tree: nested dicts; a dict is a group, anything else a Dataset; None
    is a dangling external link (KeyError on access, listed by its group)."""

    def __init__(self, tree):
        self._tree = tree

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def __getitem__(self, path):
        node = self._tree
        for part in path.strip("/").split("/"):
            node = node[part]
            if node is None:
                raise KeyError(path)
        return node if isinstance(node, (dict, Dataset)) else Dataset(node)


def velociprobe_tree(frames, *, photon_energy, beam_center, distance,
                     pixel_size, chi):
    """frames: list of (F, H, W) integer arrays, one per linked file."""
    H, W = frames[0].shape[-2:]
    data = {f"data_{i:06d}": Dataset(f) for i, f in enumerate(frames)}
    data[f"data_{len(frames):06d}"] = None  # linked but never written
    return {
        "entry": {
            "data": data,
            "instrument": {"detector": {
                "beam_center_x": np.float64(beam_center[0]),
                "beam_center_y": np.float64(beam_center[1]),
                "detector_distance": np.float64(distance),
                "x_pixel_size": np.float64(pixel_size),
                "detectorSpecific": {
                    "photon_energy": np.float64(photon_energy),
                    "x_pixels_in_detector": np.int64(W),
                    "y_pixels_in_detector": np.int64(H),
                },
            }},
            "sample": {"goniometer": {"chi": np.array([chi])}},
        }
    }


def lynx_tree(frames, pixel_size):
    return {"entry": {"data": {"eiger_4": Dataset(
        frames, attrs={"Pixel_size": np.array([pixel_size])})}}}
