"""Offline point-cloud projection on MI355X: counterpart of the reference's data/preprocess/project_transform.py
(project_pcl_to_image :67-97) + pointcloud_project_zju.py save_depth_map :57-76 / plot_radar_pcl :90-103 -- the sparse radar / lidar
scatter at the head of the RIDERS pipeline.  One launch projects every point in float64 (as numpy does with the float64 calibration
matrices), rounds half-to-even, applies canvas_crop and the depth range, and scatters max(depth, 1) with the NEAREST point winning a
pixel (atomic minimum on the float bit pattern instead of the reference's sort-by-depth + sequential overwrite: same map, bit for bit).
The barycentric lidar interpolation that follows it (data/data_utils.py:231-275, :333-367) is riders_amd.data_utils.interpolate_depth
(Delaunay on the host as in the reference, point location + barycentric evaluation of every pixel on the device, rd_tri_raster).
"""
import ctypes

import torch

from . import engine


def project_to_depth_map(point_cloud, t_camera_pcl, camera_projection_matrix, image_shape, max_distance_threshold=100.0,
                         min_distance_threshold=1.5, return_points=False):
    """point_cloud (N, >=3) float32 tensor on the device; 4x4 matrices (anything torch.as_tensor takes); image_shape (H, W[, C]).
    -> depth map (H, W) float32 [, points (M, 3) = (u, v, depth) sorted by depth, descending, as project_pcl_to_image returns them]."""
    pts = point_cloud if (point_cloud.dtype == torch.float32 and point_cloud.is_contiguous()) else point_cloud.float().contiguous()
    dev = pts.device
    H, W = int(image_shape[0]), int(image_shape[1])
    T = torch.as_tensor(t_camera_pcl, dtype=torch.float64).contiguous().to(dev)
    P = torch.as_tensor(camera_projection_matrix, dtype=torch.float64).contiguous().to(dev)
    if tuple(T.shape) != (4, 4) or tuple(P.shape) != (4, 4):
        raise ValueError("transform and projection must be 4x4")
    n = pts.shape[0]
    depth_map = torch.empty((H, W), dtype=torch.float32, device=dev)
    kept = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev) if return_points else None
    cnt = torch.empty(1, dtype=torch.int32, device=dev) if return_points else None
    engine._chk(engine.L().rd_project_scatter(engine._p(pts), n, pts.shape[1], engine._p(T), engine._p(P), H, W, ctypes.c_double(min_distance_threshold),
                                              ctypes.c_double(max_distance_threshold), engine._p(depth_map), engine._p(kept), engine._p(cnt),
                                              engine._stream(pts)), "rd_project_scatter")
    if not return_points:
        return depth_map
    k = kept[:int(cnt.item())]
    order = torch.argsort(k[:, 2], descending=True)       # host-side convenience (the .npy the reference saves is sorted by depth)
    return depth_map, k[order]
