"""Initial probes from a Fresnel model of zone-plate optics (mirror of
``tike.ptycho.fresnel``; host-side NumPy, used once before a reconstruction).

A zone plate of radius R and outermost zone width dr focuses wavelength l at
f = 2 R dr / l.  Its (thin, ideal) transmission -- a quadratic phase inside
the aperture, blocked by the central stop -- is sampled on the grid that a
single-FFT Fresnel transform maps onto the sample grid, and propagated to the
sample plane at f + defocus (fresnel.py:169-262).
"""
import numpy as np

__all__ = ["single_probe", "MW_probe"]

# radius, outermost zone width, beam-stop diameter [m]  (fresnel.py:196-212)
ZONE_PLATES = {
    "velo": dict(radius=90e-6, outmost=50e-9, beamstop=60e-6),
    "2idd": dict(radius=80e-6, outmost=70e-9, beamstop=60e-6),
    "lamni": dict(radius=114.8e-6 / 2, outmost=60e-9, beamstop=40e-6),
}


def _zone_plate_parameters(zone_plate_params):
    if isinstance(zone_plate_params, str):
        if zone_plate_params not in ZONE_PLATES:
            raise ValueError(
                f"{zone_plate_params} is not a known zone plate. "
                f"Choose one of {ZONE_PLATES.keys()} or provide a dictionary "
                "with custom zone plate parameters.")
        return ZONE_PLATES[zone_plate_params]
    return zone_plate_params


def _centred(n):
    """Integer coordinates -floor(n/2) .. ceil(n/2)-1."""
    return np.arange(-(n // 2), n - n // 2)


def _zone_plate(wavelength, defocus, width, dx, zone_plate_params):
    """(transmission (width, width), its pixel pitch, focal length): the pitch
    is the one whose Fresnel transform over focal + defocus has pitch dx."""
    zp = _zone_plate_parameters(zone_plate_params)
    focal = 2 * zp["radius"] * zp["outmost"] / wavelength
    pitch = wavelength * (focal + defocus) / width / dx
    x = -pitch * _centred(width)
    r2 = x[None, :]**2 + x[:, None]**2
    lens = np.exp(-1j * np.pi / wavelength * r2 / focal)
    aperture = (np.sqrt(r2) <= zp["radius"]) & (np.sqrt(r2) >=
                                                 zp["beamstop"] / 2)
    return lens * aperture, pitch, focal


def _fresnel_transform(field, pitch, z, wavelength):
    """Single-FFT Fresnel propagation of a centred square field over z (the
    output pitch is wavelength * |z| / (n * pitch)); z < 0 inverts it
    (fresnel.py:224-262)."""
    rows, cols = field.shape
    k = 2 * np.pi / wavelength
    gy, gx = _centred(rows), _centred(cols)
    # note the reference builds both meshes with `meshgrid(rows-axis, cols-axis)`
    XX, YY = np.meshgrid(gy * pitch, gx * pitch)
    out_pitch = wavelength * z / pitch
    UU, VV = np.meshgrid(gy * out_pitch / rows, gx * out_pitch / cols)
    chirp_in = np.exp(1j * k * (XX**2 + YY**2) / 2 / z)
    chirp_out = np.exp(1j * k * (UU**2 + VV**2) / 2 / z)
    carrier = np.exp(1j * k * z)
    if z > 0:
        spectrum = np.fft.fft2(np.fft.fftshift(field * chirp_in))
        return np.fft.fftshift(spectrum * np.fft.fftshift(carrier * chirp_out))
    back = np.fft.ifft2(np.fft.fftshift(field * chirp_out))
    return np.fft.fftshift(back) * (carrier * chirp_in)


def _unit_power(probe):
    return probe / np.sqrt(np.sum(np.abs(probe)**2))


def single_probe(probe_shape, lambda0, dx, dis_defocus, zone_plate_params):
    """One probe (1, 1, 1, W, W) complex64 for wavelength lambda0 [m], sample
    pixel size dx [m] and defocus dis_defocus [m] (fresnel.py:6-65);
    zone_plate_params: 'velo', '2idd', 'lamni' or a dict with radius, outmost,
    beamstop [m]."""
    plate, pitch, focal = _zone_plate(lambda0, dis_defocus, probe_shape, dx,
                                      zone_plate_params)
    probe = _unit_power(
        _fresnel_transform(plate, pitch, focal + dis_defocus, lambda0))
    return probe[None, None, None].astype(np.complex64)


def _gaussian_spectrum(lambda0, bandwidth, energy):
    """`energy` wavelengths across +-2 sigma of a Gaussian line of relative
    FWHM `bandwidth` (fresnel.py:160-167)."""
    sigma = lambda0 * bandwidth / 2.355
    step = sigma * 4 / (energy - 1)
    wavelengths = _centred(energy) * step + lambda0
    return np.stack(
        [wavelengths, np.exp(-(wavelengths - lambda0)**2 / sigma**2)], axis=1)


def _spectral_lines(lambda0, bandwidth, energy, spectrum):
    """`energy` (wavelength, weight) rows, brightest first: an even sub-sample
    of the supplied spectrum (ascending wavelengths), or a Gaussian line."""
    if spectrum is None:
        lines = _gaussian_spectrum(lambda0, bandwidth, energy)
    else:
        table = np.asarray(spectrum)
        lines = table[::len(table) // energy][:energy]
    brightest_first = np.argsort(-lines[:, 1])
    return lines[brightest_first][:energy]


def MW_probe(probe_shape, lambda0, dx, dis_defocus, zone_plate_params,
             energy=1, bandwidth=0.01, spectrum=None):
    """Multi-wavelength probes (1, 1, energy, W, W), brightest line first, each
    scaled by the square root of its spectral weight (fresnel.py:68-157).  All
    wavelengths are propagated to the plane that is `dis_defocus` behind the
    focus of the brightest one."""
    lines = _spectral_lines(lambda0, bandwidth, energy, spectrum)
    optics = (dis_defocus, probe_shape, dx, zone_plate_params)
    plane = _zone_plate(lines[0, 0], *optics)[2] + dis_defocus
    modes = []
    for wavelength, weight in lines:
        plate, pitch, _ = _zone_plate(wavelength, *optics)
        field = _fresnel_transform(plate, pitch, plane, wavelength)
        modes.append(np.sqrt(weight) * _unit_power(field))
    return np.stack(modes)[None, None].astype(np.complex64)
