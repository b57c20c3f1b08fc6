from oracle.rcnet import roi_pool as _roi_pool


def roi_pool(input, boxes, output_size, spatial_scale=1.0):
    return _roi_pool(input, boxes, spatial_scale, output_size)
