"""
Oracle (test infrastructure): the healpy pieces the reference uses
(``hydra/vis_simulator.py:654-659,716``; ``hydra/sh_sampler.py:257``).

healpy is not installed: PARITY UNPINNED.  Restated from the HEALPix
definition (Gorski et al. 2005, RING scheme) and the healpy a_lm conventions:

  * ``Alm.getlm``: index i = m (2 lmax + 1 - m) / 2 + l, m-major, m >= 0 only.
  * ``pix2ang(nside, ipix)`` RING ordering -> (theta colatitude, phi).
  * ``alm2map`` for a real field, no pixel window:
      T = sum_l a_l0 Y_l0 + 2 sum_{m>0} sum_l Re(a_lm Y_lm),
    the imaginary part of a_l0 is ignored.

Known-answer tests: tests/test_oracle_geometry.py (closed forms for l <= 2,
orthonormality, scipy.special.sph_harm_y cross-check).
"""
import numpy as np


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def getlm(lmax):
    """(ell, m) arrays in healpy order (``hp.Alm().getlm(lmax)``)."""
    ell, em = [], []
    for m in range(lmax + 1):
        for l in range(m, lmax + 1):
            ell.append(l)
            em.append(m)
    return np.asarray(ell, dtype=np.int64), np.asarray(em, dtype=np.int64)


def getidx(lmax, l, m):
    return m * (2 * lmax + 1 - m) // 2 + l


def getlmax(nalm):
    """Inverse of the a_lm count with mmax = lmax (``hp.Alm.getlmax``)."""
    lmax = (-3.0 + np.sqrt(1.0 + 8.0 * nalm)) / 2.0
    if lmax != np.floor(lmax):
        return -1
    return int(lmax)


def pix2ang_ring(nside, ipix):
    """(theta, phi) of RING-ordered pixel centres."""
    nside = int(nside)
    ipix = np.asarray(ipix, dtype=np.int64)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    fact2 = 4.0 / npix
    fact1 = (2 * nside) * fact2
    theta = np.empty(ipix.shape, dtype=np.float64)
    phi = np.empty(ipix.shape, dtype=np.float64)

    north = ipix < ncap
    south = ipix >= npix - ncap
    equat = ~(north | south)

    def cap(p):
        # p counted from the pole, 0-based
        iring = (1 + np.floor(np.sqrt(1 + 2 * p.astype(np.float64))).astype(np.int64)) >> 1
        # guard against sqrt rounding
        iring = np.where(2 * iring * (iring - 1) > p, iring - 1, iring)
        iring = np.where(2 * (iring + 1) * iring <= p, iring + 1, iring)
        iphi = (p + 1) - 2 * iring * (iring - 1)
        tmp = (iring * iring) * fact2
        z = 1.0 - tmp
        sth = np.sqrt(tmp * (2.0 - tmp))
        ph = (iphi - 0.5) * (0.5 * np.pi) / iring
        return z, sth, ph

    if north.any():
        z, sth, ph = cap(ipix[north])
        theta[north] = np.where(z > 0.99, np.arctan2(sth, z), np.arccos(z))
        phi[north] = ph
    if equat.any():
        ip = ipix[equat] - ncap
        nl4 = 4 * nside
        iring = ip // nl4 + nside
        iphi = ip % nl4 + 1
        fodd = np.where(((iring + nside) & 1) == 1, 1.0, 0.5)
        z = (2 * nside - iring) * fact1
        theta[equat] = np.arccos(z)
        phi[equat] = (iphi - fodd) * np.pi * 0.75 * fact1
    if south.any():
        p = (npix - 1) - ipix[south]
        iring = (1 + np.floor(np.sqrt(1 + 2 * p.astype(np.float64))).astype(np.int64)) >> 1
        iring = np.where(2 * iring * (iring - 1) > p, iring - 1, iring)
        iring = np.where(2 * (iring + 1) * iring <= p, iring + 1, iring)
        iphi = 4 * iring + 1 - ((p + 1) - 2 * iring * (iring - 1))
        tmp = (iring * iring) * fact2
        z = tmp - 1.0
        sth = np.sqrt(tmp * (2.0 - tmp))
        theta[south] = np.where(z < -0.99, np.arctan2(sth, z), np.arccos(z))
        phi[south] = (iphi - 0.5) * (0.5 * np.pi) / iring
    return theta, phi


def lambda_lm(lmax, theta):
    """Normalised associated Legendre functions
    lambda_lm(theta) = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) P_lm(cos theta),
    Condon-Shortley phase included, in healpy (m-major) order.
    Returns shape (nalm, Npts).
    """
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    x = np.cos(theta)
    s = np.sin(theta)
    nalm = (lmax + 1) * (lmax + 2) // 2
    out = np.zeros((nalm, theta.size), dtype=np.float64)
    pmm = np.full(theta.shape, np.sqrt(1.0 / (4.0 * np.pi)))
    for m in range(lmax + 1):
        if m > 0:
            pmm = -pmm * s * np.sqrt((2.0 * m + 1.0) / (2.0 * m))
        i0 = getidx(lmax, m, m)
        out[i0] = pmm
        if m == lmax:
            break
        p1 = x * np.sqrt(2.0 * m + 3.0) * pmm
        out[i0 + 1] = p1
        p0 = pmm
        for l in range(m + 2, lmax + 1):
            a = np.sqrt((4.0 * l * l - 1.0) / (l * l - m * m))
            b = np.sqrt(((l - 1.0) ** 2 - m * m) / (4.0 * (l - 1.0) ** 2 - 1.0))
            p2 = a * (x * p1 - b * p0)
            out[i0 + (l - m)] = p2
            p0, p1 = p1, p2
    return out


def alm2map(alm, nside, lmax=None):
    """Real-field synthesis on the RING pixel centres (``hp.alm2map`` without
    pixel window).  ``alm`` in healpy order with mmax = lmax."""
    alm = np.asarray(alm, dtype=np.complex128)
    if lmax is None:
        lmax = getlmax(alm.size)
    theta, phi = pix2ang_ring(nside, np.arange(nside2npix(nside)))
    lam = lambda_lm(lmax, theta)
    ell, em = getlm(lmax)
    out = np.zeros(theta.size, dtype=np.float64)
    for m in range(lmax + 1):
        sel = em == m
        # sum over l of a_lm lambda_lm  -> complex function of theta
        f = alm[sel] @ lam[sel]
        if m == 0:
            out += f.real
        else:
            out += 2.0 * (f.real * np.cos(m * phi) - f.imag * np.sin(m * phi))
    return out


def unit_mode_maps(lmax, nside):
    """The 2*nalm maps ``alm2map(e_n * val)`` for val in (1, 1j), stacked in the
    reference's output order (first nalm: real-part modes, last nalm:
    imaginary-part modes; ``vis_simulator.py:702-730``).  Shape (2*nalm, Npix).
    """
    theta, phi = pix2ang_ring(nside, np.arange(nside2npix(nside)))
    lam = lambda_lm(lmax, theta)
    ell, em = getlm(lmax)
    nalm = ell.size
    maps = np.zeros((2 * nalm, theta.size), dtype=np.float64)
    for n in range(nalm):
        m = em[n]
        if m == 0:
            maps[n] = lam[n]
        else:
            maps[n] = 2.0 * lam[n] * np.cos(m * phi)
            maps[n + nalm] = -2.0 * lam[n] * np.sin(m * phi)
    return maps


def _ring_info(nside, ring):
    """(startpix, ringpix, theta, shifted) of RING-scheme iso-latitude ring 1 .. 4 nside - 1
    (HEALPix get_ring_info2)."""
    nside = int(nside)
    ring = np.asarray(ring, dtype=np.int64)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    fact2 = 4.0 / npix
    fact1 = (2 * nside) * fact2
    north = np.where(ring > 2 * nside, 4 * nside - ring, ring)
    capr = north < nside
    tmp = north * north * fact2
    costh = np.where(capr, 1.0 - tmp, (2 * nside - north) * fact1)
    sinth = np.sqrt(np.maximum((1.0 - costh) * (1.0 + costh), 0.0))
    theta = np.where(capr, np.arctan2(sinth, costh), np.arccos(np.clip(costh, -1.0, 1.0)))
    ringpix = np.where(capr, 4 * north, 4 * nside)
    shifted = np.where(capr, True, ((north - nside) & 1) == 0)
    startpix = np.where(capr, 2 * north * (north - 1), ncap + (north - nside) * 4 * nside)
    south = north != ring
    theta = np.where(south, np.pi - theta, theta)
    startpix = np.where(south, npix - startpix - ringpix, startpix)
    return startpix, ringpix, theta, shifted


def interp_weights_ring(nside, theta, phi):
    """The four RING-scheme pixels and weights of HEALPix's bilinear interpolation at
    (theta, phi) (Healpix_Base::get_interpol -- what healpy.get_interp_weights and
    astropy-healpix's bilinear_interpolation_weights return, hence pyuvdata's
    ``healpix_simple`` beam interpolation).  Returns (pix (4, N) int64, wgt (4, N))."""
    nside = int(nside)
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    phi = np.mod(np.atleast_1d(np.asarray(phi, dtype=np.float64)), 2.0 * np.pi)
    npix = 12 * nside * nside
    z = np.cos(theta)
    az = np.abs(z)
    ir_eq = (nside * (2.0 - 1.5 * z)).astype(np.int64)
    ir_cap = (nside * np.sqrt(3.0 * (1.0 - az))).astype(np.int64)
    ir1 = np.where(az <= 2.0 / 3.0, ir_eq, np.where(z > 0, ir_cap, 4 * nside - ir_cap - 1))
    ir2 = ir1 + 1
    pix = np.zeros((4, theta.size), dtype=np.int64)
    wgt = np.zeros((4, theta.size), dtype=np.float64)
    th = [np.zeros(theta.size), np.zeros(theta.size)]
    for k, ir in enumerate((ir1, ir2)):
        ok = (ir >= 1) & (ir <= 4 * nside - 1)
        sp, nr, thr, shift = _ring_info(nside, np.clip(ir, 1, 4 * nside - 1))
        dphi = 2.0 * np.pi / nr
        tmp = phi / dphi - 0.5 * shift
        i1 = np.where(tmp < 0, tmp.astype(np.int64) - 1, tmp.astype(np.int64))
        w1 = (phi - (i1 + 0.5 * shift) * dphi) / dphi
        i2 = i1 + 1
        i1 = np.where(i1 < 0, i1 + nr, i1)
        i2 = np.where(i2 >= nr, i2 - nr, i2)
        pix[2 * k] = np.where(ok, sp + i1, 0)
        pix[2 * k + 1] = np.where(ok, sp + i2, 0)
        wgt[2 * k] = np.where(ok, 1.0 - w1, 0.0)
        wgt[2 * k + 1] = np.where(ok, w1, 0.0)
        th[k] = thr
    npole, spole = ir1 == 0, ir2 == 4 * nside
    mid = ~(npole | spole)
    wt = np.where(mid, (theta - th[0]) / np.where(mid, th[1] - th[0], 1.0), 0.0)
    wgt[0] = np.where(mid, wgt[0] * (1.0 - wt), wgt[0])
    wgt[1] = np.where(mid, wgt[1] * (1.0 - wt), wgt[1])
    wgt[2] = np.where(mid, wgt[2] * wt, wgt[2])
    wgt[3] = np.where(mid, wgt[3] * wt, wgt[3])
    # north polar cap: the four pixels of ring 1 stand in for the missing ring
    wn = theta / th[1]
    fac = (1.0 - wn) * 0.25
    wgt[2] = np.where(npole, wgt[2] * wn + fac, wgt[2])
    wgt[3] = np.where(npole, wgt[3] * wn + fac, wgt[3])
    wgt[0] = np.where(npole, fac, wgt[0])
    wgt[1] = np.where(npole, fac, wgt[1])
    pix[0] = np.where(npole, (pix[2] + 2) & 3, pix[0])
    pix[1] = np.where(npole, (pix[3] + 2) & 3, pix[1])
    # south polar cap
    ws = (theta - th[0]) / (np.pi - th[0])
    fac = ws * 0.25
    wgt[0] = np.where(spole, wgt[0] * (1.0 - ws) + fac, wgt[0])
    wgt[1] = np.where(spole, wgt[1] * (1.0 - ws) + fac, wgt[1])
    wgt[2] = np.where(spole, fac, wgt[2])
    wgt[3] = np.where(spole, fac, wgt[3])
    pix[2] = np.where(spole, ((pix[0] + 2) & 3) + npix - 4, pix[2])
    pix[3] = np.where(spole, ((pix[1] + 2) & 3) + npix - 4, pix[3])
    return pix, wgt
