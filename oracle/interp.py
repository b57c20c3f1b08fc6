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


def interpolate_knots(map_size, knot_coords, knot_values, interpolate, fill_corners=False):
    """modules/interpolator.py:7-18: scipy.interpolate.griddata(points=knot_coords.T, values, xi=(grid_y, grid_x), method, fill_value=1.0)
    with grid_x, grid_y = np.mgrid[0:H, 0:W] -- i.e. points are (x = column, y = row) and the query of pixel (row, col) is (col, row).
    griddata's 'linear' is LinearNDInterpolator, its 'nearest' NearestNDInterpolator (third-party scipy, as installed in this image).
    Pinned by tests/golden/g17_interpolator.npz (the reference's Interpolator2D run here)."""
    from scipy.interpolate import NearestNDInterpolator
    H, W = int(map_size[0]), int(map_size[1])
    pts = np.asarray(knot_coords).T
    rows, cols = np.mgrid[0:H, 0:W]
    q = np.stack([cols.ravel(), rows.ravel()], axis=1)
    if interpolate == 'linear':
        f = LinearNDInterpolator(pts, np.asarray(knot_values), fill_value=1.0)
    elif interpolate == 'nearest':
        f = NearestNDInterpolator(pts, np.asarray(knot_values))
    else:
        raise NotImplementedError(interpolate)
    return f(q).reshape(H, W)


def interpolated_scale_map(pred_inv, sparse_depth_inv, valid, method):
    """Interpolator2D.__init__ + generate_interpolated_scale_map (modules/interpolator.py:21-49): knots = valid pixels, values = the ratio
    sparse / predicted inverse depth, float32 result."""
    ys, xs = np.nonzero(valid)
    scales = sparse_depth_inv[valid] / pred_inv[valid]
    return interpolate_knots(np.shape(pred_inv), np.stack((xs, ys)), scales, method).astype(np.float32)
