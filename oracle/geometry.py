"""
Oracle (test infrastructure): coordinate helpers the reference takes from
``matvis.coordinates`` (third party, not installed -> PARITY UNPINNED, formulas
restated from the package's published definitions).

Reference call sites:
  * ``hydra/vis_simulator.py:467``  point_source_crd_eq(ra, dec)
  * ``hydra/vis_simulator.py:470``  eci_to_enu_matrix(lst, latitude)
  * ``hydra/vis_simulator.py:247``  enu_to_az_za(enu_e, enu_n, orientation="uvbeam")
  * ``hydra/vis_simulator.py:262``  astropy.constants.c
"""
import numpy as np

#: astropy.constants.c.value (CODATA, exact), used at vis_simulator.py:262
C_LIGHT = 299792458.0


def point_source_crd_eq(ra, dec):
    """Unit vectors of sources in the ECI frame, shape (3, Nsrc).

    (cos ra cos dec, sin ra cos dec, sin dec): the layout documented at
    ``hydra/vis_simulator.py:137-143``.
    """
    ra = np.asarray(ra, dtype=np.float64)
    dec = np.asarray(dec, dtype=np.float64)
    cd = np.cos(dec)
    return np.asarray([np.cos(ra) * cd, np.sin(ra) * cd, np.sin(dec)])


def eci_to_enu_matrix(ha, lat):
    """3x3 rotation ECI -> local East/North/Up for hour angle ``ha`` (the
    reference passes the LST) and observer latitude ``lat``.

    Checks (tests/test_oracle_geometry.py): a source at ra=lst, dec=lat maps to
    (0, 0, 1); a source with ra slightly larger than lst has positive East.
    """
    sh, ch = np.sin(ha), np.cos(ha)
    sl, cl = np.sin(lat), np.cos(lat)
    return np.array(
        [
            [-sh, ch, 0.0],
            [-sl * ch, -sl * sh, cl],
            [cl * ch, cl * sh, sl],
        ]
    )


def enu_to_az_za(enu_e, enu_n, orientation="uvbeam", periodic_azimuth=True):
    """Azimuth / zenith angle from the East and North direction cosines.

    Only (e, n) enter, so a source below the horizon is mirrored to
    za in [0, pi/2]; the reference zeroes those sources afterwards
    (``vis_simulator.py:237,272``).  "uvbeam" azimuth runs from East through
    North; "astropy" from North through East.
    """
    enu_e = np.asarray(enu_e, dtype=np.float64)
    enu_n = np.asarray(enu_n, dtype=np.float64)
    lsqr = enu_n**2 + enu_e**2
    inside = lsqr < 1
    zeta = np.zeros_like(lsqr)
    zeta[inside] = np.sqrt(1 - lsqr[inside])
    az = np.arctan2(enu_e, enu_n)
    za = 0.5 * np.pi - np.arcsin(zeta)
    if orientation == "uvbeam":
        az = 0.5 * np.pi - az
    elif orientation != "astropy":
        raise ValueError("orientation must be 'astropy' or 'uvbeam'")
    if periodic_azimuth:
        az = np.mod(az, 2 * np.pi)
    return az, za
