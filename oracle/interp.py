"""ORACLE (test infrastructure only): the reference's lidar interpolation restated -- data/data_utils.py interpolate_depth :231-275 and
interpolate_depth_delft :333-367: valid pixels -> scipy.interpolate.LinearNDInterpolator (Delaunay + barycentric weights, third-party
Qhull, scipy as installed in this image) evaluated at every pixel; fill 0 (log space: log(1e-3), then exp and values < 0.1 zeroed).
Pinned by tests/golden/g15_interpolation.npz, produced by the reference's own functions (tests/golden/make_golden.py g_interpolation)."""
import numpy as np
from scipy.interpolate import LinearNDInterpolator


def interpolate_depth(depth_map, validity_map=None, log_space=False):
    depth_map = np.asarray(depth_map)
    if validity_map is None:                                  # data_utils.py:346-348
        validity_map = depth_map > 0.0
    rows, cols = depth_map.shape
    r, c = np.where(validity_map)                             # :250-251
    v = depth_map[r, c]
    if log_space:                                             # :254-255
        v = np.log(v)
    f = LinearNDInterpolator(points=np.stack([r, c], axis=1), values=v, fill_value=0 if not log_space else np.log(1e-3))   # :256-260
    qr, qc = np.meshgrid(np.arange(rows), np.arange(cols), indexing='ij')                                                  # :262-266
    Z = f(np.stack([qr.ravel(), qc.ravel()], axis=1)).reshape([rows, cols])                                                # :268
    if log_space:                                             # :270-272
        Z = np.exp(Z)
        Z[Z < 1e-1] = 0.0
    return Z
