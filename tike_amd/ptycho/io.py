"""Loading raw ptychography data from disk (mirror of ``tike.ptycho.io``).

Host-side only: the readers return the ``(data, scan)`` pair that
``tike_amd.ptycho.reconstruct`` takes -- diffraction patterns cropped square
around the beam centre, optionally binned, FFT-shifted so that the peak sits at
the corners (what the forward model expects, reference ptycho.py:196-199), and
scan positions in object pixels.  ``data`` keeps the detector's integer type
(uint16 frames then stay 16-bit in HBM, see ``_arrays.data_to_device``).

Reference: src/tike/ptycho/io.py:20-58 (position_units_to_pixels), :61-285
(read_aps_velociprobe), :288-449 (read_aps_lynx).  HDF5 access needs ``h5py``
(with the lz4 filter plugin for Velociprobe master files); it is imported when
a reader is called.
"""
import logging
import warnings

import numpy as np

logger = logging.getLogger(__name__)

_PLANCK_KEV_S = 6.58211928e-19  # reduced Planck constant [keV s] (constants.py:67)
_LIGHT_CM_S = 299792458e2  # [cm / s] (constants.py:68)

__all__ = ["position_units_to_pixels", "read_aps_velociprobe", "read_aps_lynx"]


def _wavelength_cm(energy_kev):
    """constants.py:70-72."""
    return 2 * np.pi * _PLANCK_KEV_S * _LIGHT_CM_S / energy_kev


def position_units_to_pixels(positions, detector_distance,
                             detector_pixel_count, detector_pixel_width,
                             photon_energy):
    """Scan positions [m] -> object pixels for a far-field geometry
    (io.py:20-58): one object pixel is ``distance * wavelength /
    (pixel_width * pixel_count)`` metres wide.  photon_energy in eV."""
    wavelength_m = _wavelength_cm(photon_energy / 1000) / 100
    pixel_per_meter = ((detector_pixel_width * detector_pixel_count) /
                       (detector_distance * wavelength_m))
    logger.info("reconstruction pixel size %.3e m", 1 / pixel_per_meter)
    return positions * pixel_per_meter


def _open_h5(path):
    try:
        import h5py
    except ImportError as error:  # pragma: no cover - depends on the image
        raise ImportError(
            "reading HDF5 diffraction data needs h5py (and, for APS master "
            "files, the lz4 HDF5 filter plugin)") from error
    return h5py.File(path, "r")


def _crop_radius(beam_center_x, beam_center_y, width, height, max_crop):
    """Largest power-of-two half-width around the beam centre that stays on
    the detector and within max_crop (io.py:160-172)."""
    radius = 2
    while (radius <= max_crop // 2 and beam_center_x + radius < width
           and beam_center_y + radius < height and beam_center_x - radius >= 0
           and beam_center_y - radius >= 0):
        radius *= 2
    return radius // 2


def _binned_width(radius, binned_pix):
    width = (2 * radius) // binned_pix
    if width * binned_pix != 2 * radius:
        raise ValueError(
            f"Invalid pixel binning provided! {2 * radius} cannot be "
            f"evenly collected into bins of {binned_pix}.")
    return width


def _crop_bin_shift(frames, beam_center_x, beam_center_y, radius, binned_pix,
                    gap_value=None, block=256):
    """Frames (F, H, W) -> (F, w, w): crop, zero the detector gaps, sum
    binned_pix x binned_pix bins in the frames' own dtype, move the centre to
    the corners (io.py:183-206,376-402).  Read `block` frames at a time so that
    an HDF5 dataset is never materialised whole before cropping."""
    width = _binned_width(radius, binned_pix)
    count = frames.shape[0]
    out = None
    ys = slice(beam_center_y - radius, beam_center_y + radius)
    xs = slice(beam_center_x - radius, beam_center_x + radius)
    for lo in range(0, count, block):
        part = np.asarray(frames[lo:lo + block, ys, xs])
        if gap_value is not None:
            part = np.where(part == gap_value, 0, part).astype(part.dtype)
        if binned_pix > 1:
            part = part.reshape(len(part), width, binned_pix, width,
                                binned_pix).sum(axis=(2, 4), dtype=part.dtype)
        if out is None:
            out = np.empty((count, width, width), dtype=part.dtype)
        out[lo:lo + len(part)] = np.fft.ifftshift(part, axes=(-2, -1))
    if out is None:
        out = np.empty((0, width, width), dtype=getattr(frames, "dtype", float))
    return out


def _match_lengths(data, scan):
    if len(data) != len(scan):
        warnings.warn(
            f"The number of positions {scan.shape} and frames {data.shape}"
            " is not equal. One of the two will be truncated.")
        n = min(len(data), len(scan))
        data, scan = data[:n], scan[:n]
    return data, scan


def _check_counts(data):
    if data.dtype.kind == "f" and not np.all(np.isfinite(data)):
        warnings.warn("Some values in the diffraction data are not finite. "
                      "Photon counts must be >= 0 and finite.")
    if data.dtype.kind != "u" and np.any(data < 0):
        warnings.warn("Some values in the diffraction data are negative. "
                      "Photon counts must be >= 0 and finite.")


def _trigger_positions(raw):
    """(rows, 3) integer table [x, y, trigger] -> one position per trigger:
    the mean of the first and the last sample of every run of equal trigger
    numbers (io.py:236-253)."""
    starts = np.concatenate(([0], np.nonzero(np.diff(raw[:, -1]))[0] + 1))
    ends = np.concatenate((starts[1:], [len(raw)])) - 1
    return (raw[starts, :2] + raw[ends, :2]) / 2


def read_aps_velociprobe(diffraction_path, position_path, xy_columns=(5, 1),
                         trigger_column=7, max_crop=2048, binned_pix=1):
    """Load data of the APS Velociprobe (2-ID-D): an HDF5 master file
    (``/entry/data/data_00000N`` int[FRAME, WIDE, HIGH] plus the detector
    geometry under ``/entry/instrument/detector``) and one or several
    8-column CSV files of interferometer samples (io.py:61-285).

    Returns (data (FRAME, w, w) cropped / binned / shifted, scan (FRAME, 2)
    float32 in pixels, not centred)."""
    with _open_h5(diffraction_path) as f:
        det = "/entry/instrument/detector"
        photon_energy = f[det + "/detectorSpecific/photon_energy"][()]  # eV
        width = int(f[det + "/detectorSpecific/x_pixels_in_detector"][()])
        height = int(f[det + "/detectorSpecific/y_pixels_in_detector"][()])
        detector_dist = f[det + "/detector_distance"][()]  # m
        pixel_width = f[det + "/x_pixel_size"][()]  # m
        beam_center_x = int(f[det + "/beam_center_x"][()])
        beam_center_y = int(f[det + "/beam_center_y"][()])
        chi = float(f["entry/sample/goniometer/chi"][0])
        radius = _crop_radius(beam_center_x, beam_center_y, width, height,
                              max_crop)
        _binned_width(radius, binned_pix)
        logger.info("Velociprobe: chi %g deg, %g eV, crop %d, bin %d", chi,
                    photon_energy, 2 * radius, binned_pix)
        parts = []
        for name in f["/entry/data"]:
            try:
                frames = f[f"/entry/data/{name}"]
            except KeyError:
                break  # the master file links more files than were written
            try:
                parts.append(_crop_bin_shift(frames, beam_center_x,
                                             beam_center_y, radius,
                                             binned_pix))
            except OSError:
                warnings.warn(
                    "The HDF5 compression plugin is probably missing. "
                    "See the conda-forge hdf5-external-filter-plugins package.")
                raise
        data = np.concatenate(parts, axis=0)

    paths = position_path if isinstance(position_path, list) else [position_path]
    raw = np.concatenate([
        np.genfromtxt(p, usecols=(*xy_columns, trigger_column), delimiter=",",
                      dtype=np.int32).reshape(-1, 3) for p in paths
    ], axis=0)
    scan = _trigger_positions(raw)
    # stage geometry: nanometres; the horizontal stage rides on the rotation
    scan[:, 0] *= -1e-9
    scan -= np.mean(scan, axis=0, keepdims=True)
    scan[:, 1] *= 1e-9 * np.cos(chi / 180 * np.pi)
    logger.info("Loaded %d scan positions.", len(scan))
    data, scan = _match_lengths(data, scan)
    scan = position_units_to_pixels(scan, detector_dist, data.shape[-1],
                                    pixel_width * binned_pix, photon_energy)
    _check_counts(data)
    return data, scan.astype(np.float32)


def read_aps_lynx(diffraction_path, position_path, photon_energy,
                  beam_center_x, beam_center_y, detector_dist,
                  xy_columns=(6, 3), trigger_column=0, max_crop=2048,
                  gap_value=2**12 - 1, binned_pix=1):
    """Load data of APS LYNX (28-ID-C): ``/entry/data/eiger_4``
    uint16[FRAME, HIGH, WIDE] with the pixel size as an attribute, detector
    gaps encoded as `gap_value`, and a space-separated .dat file (two header
    rows) of positions in micrometres (io.py:288-449)."""
    with _open_h5(diffraction_path) as f:
        frames = f["/entry/data/eiger_4"]
        pixel_width = np.asarray(frames.attrs["Pixel_size"]).item()  # m
        _, height, width = frames.shape
        radius = _crop_radius(beam_center_x, beam_center_y, width, height,
                              max_crop)
        _binned_width(radius, binned_pix)
        logger.info("LYNX: %g eV, crop %d, bin %d", photon_energy, 2 * radius,
                    binned_pix)
        try:
            data = _crop_bin_shift(frames, beam_center_x, beam_center_y,
                                   radius, binned_pix, gap_value=gap_value)
        except OSError:
            warnings.warn(
                "The HDF5 compression plugin is probably missing. "
                "See the conda-forge hdf5-external-filter-plugins package.")
            raise
    raw = np.genfromtxt(position_path, usecols=(*xy_columns, trigger_column),
                        delimiter=" ", dtype=np.float32,
                        skip_header=2).reshape(-1, 3)
    scan = raw[:, :2] * -1e-6
    logger.info("Loaded %d scan positions.", len(scan))
    data, scan = _match_lengths(data, scan)
    scan = position_units_to_pixels(scan, detector_dist, data.shape[-1],
                                    pixel_width * binned_pix, photon_energy)
    _check_counts(data)
    return data, scan.astype(np.float32)
