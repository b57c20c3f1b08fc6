"""Knot interpolation of the Scale Map Learner's scale scaffolding: modules/interpolator.py of the reference (interpolate_knots :7-18,
Interpolator2D :21-49), same names, arguments and attributes.

The reference evaluates scipy.interpolate.griddata at every pixel of the map (fill_value 1.0).  Here the H*W evaluations run on the
device: 'linear' = Delaunay triangulation of the knots on the host (Qhull, as inside griddata) + point location and barycentric weights
per pixel in rd_tri_raster (exact int64 edge functions); 'nearest' = rd_nearest_knot (exact integer distances).  These two need a ROCm
device.  'cubic' (Clough-Tocher: iterative global gradient estimation on the triangulation, then a C1 piecewise cubic) has NO device
kernel: RIDERS never selects it (train_zju.py uses interp 'rcnet'; this class is not instantiated anywhere in the reference), so the method
value is passed through to the same scipy.interpolate.griddata call the reference makes (modules/interpolator.py:10-16) -- off the hot
path, host only, identical result by construction.
"""
import numpy as np


def interpolate_knots(map_size, knot_coords, knot_values, interpolate, fill_corners, device=None):
    """modules/interpolator.py:7-18.  knot_coords: (2, K) integer array (x = column, y = row); returns float64 (map_size)."""
    import torch
    from . import engine
    H, W = int(map_size[0]), int(map_size[1])
    knot_coords = np.asarray(knot_coords)
    if interpolate == 'cubic':      # host, as the reference (see the module docstring): no device kernel for a method RIDERS never selects
        from scipy.interpolate import griddata
        grid_x, grid_y = np.mgrid[0:H, 0:W]
        return griddata(points=knot_coords.T, values=knot_values, xi=(grid_y, grid_x), method='cubic', fill_value=1.0)
    pts = np.ascontiguousarray(knot_coords.T)                    # (K, 2) as (x, y): the `points` griddata triangulates
    vals = np.ascontiguousarray(np.asarray(knot_values), dtype=np.float64)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    prow = torch.from_numpy(np.ascontiguousarray(pts[:, 1], dtype=np.int32)).to(dev)
    pcol = torch.from_numpy(np.ascontiguousarray(pts[:, 0], dtype=np.int32)).to(dev)
    v = torch.from_numpy(vals).to(dev)
    out = torch.empty((H, W), dtype=torch.float64, device=dev)
    lib, st = engine.L(), engine._stream(out)
    if interpolate == 'linear':
        from scipy.spatial import Delaunay
        tri = Delaunay(pts.astype(np.float64))                   # raises on fewer than 3 / collinear knots, like griddata
        simp = torch.from_numpy(np.ascontiguousarray(tri.simplices, dtype=np.int32)).to(dev)
        owner = torch.empty((H, W), dtype=torch.int32, device=dev)
        engine._chk(lib.rd_tri_raster(engine._p(simp), engine._p(prow), engine._p(pcol), engine._p(v), int(simp.shape[0]), H, W, 1.0,
                                      engine._p(owner), engine._p(out), st), "rd_tri_raster")
    elif interpolate == 'nearest':
        engine._chk(lib.rd_nearest_knot(engine._p(prow), engine._p(pcol), engine._p(v), int(pts.shape[0]), H, W, 1.0, engine._p(out), st),
                    "rd_nearest_knot")
    else:
        raise ValueError("Unknown interpolation method %r for 2 dimensional data" % (interpolate,))      # griddata's own error
    return out.cpu().numpy()


class Interpolator2D(object):
    """modules/interpolator.py:21-49."""

    def __init__(self, pred_inv, sparse_depth_inv, valid):
        self.pred_inv = pred_inv
        self.sparse_depth_inv = sparse_depth_inv
        self.valid = valid

        self.map_size = np.shape(pred_inv)
        self.num_knots = np.sum(valid)
        nonzero_y_loc = np.nonzero(valid)[0]
        nonzero_x_loc = np.nonzero(valid)[1]
        self.knot_coords = np.stack((nonzero_x_loc, nonzero_y_loc))
        self.knot_scales = sparse_depth_inv[valid] / pred_inv[valid]
        self.knot_shifts = sparse_depth_inv[valid] - pred_inv[valid]

        self.knot_list = [(int(self.knot_coords[0, i]), int(self.knot_coords[1, i])) for i in range(int(self.num_knots))]

        # to be computed
        self.interpolated_map = None
        self.confidence_map = None
        self.output = None

    def generate_interpolated_scale_map(self, interpolate_method, fill_corners=False, device=None):
        self.interpolated_scale_map = interpolate_knots(
            map_size=self.map_size,
            knot_coords=self.knot_coords,
            knot_values=self.knot_scales,
            interpolate=interpolate_method,
            fill_corners=fill_corners,
            device=device
        ).astype(np.float32)
