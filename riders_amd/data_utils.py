"""Depth-map disk format of the RIDERS pipeline: 16-bit grayscale PNG holding uint16(depth * 256), the hand-off between RC-Net
inference (RCNet/run_rcnet_zju.py:266) and the Scale Map Learner's dataset (data/UTV_dataset.py:14-17).

Same signatures and arithmetic as the reference's data/data_utils.py load_depth :94-125 and save_depth :128-143
(`np.uint32(z * multiplier)` -> PIL mode 'I' -> PNG, which PIL stores as 16-bit big-endian samples clamped to 0..65535).
The quantisation runs on the device (rd_depth_quantize_u16) when `z` is a ROCm tensor, so only two bytes per pixel cross PCIe;
the PNG container itself (zlib-compressed scanlines) is written and parsed here without PIL.
"""
import ctypes
import os
import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def quantize_depth(z, multiplier=256.0):
    """-> numpy uint16 (H,W): clamp(trunc(z * multiplier), 0, 65535), float32 arithmetic as np.uint32(z * multiplier)."""
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and torch.is_tensor(z):
        from . import engine
        zc = z.detach().float().contiguous()
        out = torch.empty(zc.shape, dtype=torch.int16, device=zc.device)
        engine._chk(engine.L().rd_depth_quantize_u16(engine._p(zc), engine._p(out), zc.numel(), float(multiplier), engine._stream(zc)),
                    "rd_depth_quantize_u16")
        return out.cpu().numpy().view(np.uint16)
    v = np.asarray(z, dtype=np.float32) * np.float32(multiplier)
    v = np.where(v >= 65536.0, 65535.0, np.where(v > 0, np.trunc(v), 0.0))
    return v.astype(np.uint16)


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def encode_png16(q, level=6):
    """uint16 (H,W) -> bytes of a 16-bit grayscale PNG (colour type 0, filter 0 on every scanline)."""
    q = np.ascontiguousarray(q, dtype=np.uint16)
    assert q.ndim == 2, "depth maps are H x W"
    h, w = q.shape
    rows = np.empty((h, 1 + 2 * w), dtype=np.uint8)
    rows[:, 0] = 0
    rows[:, 1:] = q.astype(">u2").view(np.uint8).reshape(h, 2 * w)
    ihdr = struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)
    return _SIG + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(rows.tobytes(), level)) + _chunk(b"IEND", b"")


_host = {"lib": None}


def _host_lib():
    """riders_amd/libriders_host.so (csrc/rd_host.cpp built by plain g++): host-only helpers bound WITHOUT importing torch or touching the GPU
    runtime -- this runs inside DataLoader workers (RCNetTrainingDataset.__getitem__ -> load_image / load_depth), which must not create a
    device context and cannot re-initialise one after a fork."""
    if _host["lib"] is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libriders_host.so")
        if not os.path.exists(path):
            raise RuntimeError("riders_amd: %s not found; build it with `python -m riders_amd.build`" % path)
        lib = ctypes.CDLL(path)
        lib.rd_png_unfilter_host.restype = ctypes.c_int
        lib.rd_png_unfilter_host.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        _host["lib"] = lib
    return _host["lib"]


def decode_png16(data):
    """bytes of an 8/16-bit grayscale (what PIL writes for modes 'I' / 'I;16' / 'L') or 8-bit RGB / RGBA, non-interlaced PNG -> integer
    (H,W) or (H,W,C) array."""
    if data[:8] != _SIG:
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    chans = {0: 1, 2: 3, 6: 4}.get(ctype)
    if chans is None or interlace != 0 or depth not in (8, 16) or (chans > 1 and depth != 8):
        raise ValueError("unsupported PNG layout (colour type %d, bit depth %d, interlace %d)" % (ctype, depth, interlace))
    bpp = chans * depth // 8
    raw = np.ascontiguousarray(np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * bpp))
    out = np.zeros((h, w * bpp), dtype=np.uint8)
    # PNG scanline filters (PIL picks them adaptively): the serial Average / Paeth recurrences run in the library's host helper
    rc = _host_lib().rd_png_unfilter_host(raw.ctypes.data_as(ctypes.c_void_p), h, w * bpp, bpp, out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise ValueError("bad PNG filter data (rd_png_unfilter_host returned %d)" % rc)
    if chans > 1:
        return out.reshape(h, w, chans)
    return out.view(">u2").astype(np.uint16) if depth == 16 else out


def load_image(path, normalize=False, data_format='HWC'):
    """data/data_utils.py:59-90: an RGB image as float32 (H,W,3) or (3,H,W); 8-bit grayscale files are replicated to three channels and an
    alpha channel is dropped, as PIL's convert('RGB') does."""
    px = decode_png16(open(path, "rb").read())
    if px.ndim == 2:
        if px.dtype != np.uint8:
            raise ValueError("16-bit grayscale images are depth maps: use load_depth")
        px = np.repeat(px[:, :, None], 3, axis=2)
    image = np.asarray(px[:, :, :3], np.float32)
    if data_format == 'HWC':
        pass
    elif data_format == 'CHW':
        image = np.transpose(image, (2, 0, 1))
    else:
        raise ValueError('Unsupported data format: {}'.format(data_format))
    return image / 255.0 if normalize else image


def load_depth(path, multiplier=256.0, data_format='HW'):
    """data/data_utils.py:94-125."""
    z = decode_png16(open(path, "rb").read()).astype(np.float32)
    z = z / multiplier
    z[z <= 0] = 0.0
    if data_format == 'HW':
        pass
    elif data_format == 'CHW':
        z = np.expand_dims(z, axis=0)
    elif data_format == 'HWC':
        z = np.expand_dims(z, axis=-1)
    else:
        raise ValueError('Unsupported data format: {}'.format(data_format))
    return z


def save_depth(z, path, multiplier=256.0):
    """data/data_utils.py:128-143; `z` may be a numpy array or a (ROCm) tensor."""
    with open(path, "wb") as f:
        f.write(encode_png16(quantize_depth(z, multiplier)))


# ---- stage hand-off layout (RC-Net inference -> Scale Map Learner) ------------------------------------------------------------------
def rcnet_output_paths(output_path, radar_path):
    """RCNet/run_rcnet_zju.py:159-171: where the quasi-dense depth / colour preview / response of one radar frame are written:
    <output_path>/<scene>/{depth_predicted, depth_predicted_colors, response_predicted}/<file id>.png, scene = third path component from
    the end of the radar file path."""
    import os
    file_id = os.path.basename(radar_path).split('.')[0]
    save_scene = radar_path.split('/')[-3]
    return (os.path.join(output_path, save_scene, 'depth_predicted', file_id + '.png'),
            os.path.join(output_path, save_scene, 'depth_predicted_colors', file_id + '.png'),
            os.path.join(output_path, save_scene, 'response_predicted', file_id + '.png'))


def sml_rcnet_input_paths(result_root, interp, scene):
    """train_zju.py:114-117 / val_zju.py: the SML reads the quasi-dense maps of a scene from <result_root>/<interp>/<scene>/depth_predicted,
    sorted by file name -- i.e. `output_path` of the RC-Net stage = <result_root>/<interp>."""
    import os
    root = os.path.join(result_root, interp, scene, 'depth_predicted')
    return [os.path.join(root, p) for p in sorted(os.listdir(root))]


def interpolate_depth(depth_map, validity_map=None, log_space=False, device=None):
    """Barycentric interpolation of a sparse depth map over the Delaunay triangulation of its valid pixels: same arguments and result
    (H x W float64 numpy array) as the reference's data/data_utils.py interpolate_depth :231-275 and interpolate_depth_delft :333-367
    (`validity_map=None` -> depth_map > 0, as the latter).  The reference hands the valid pixels to scipy's LinearNDInterpolator and
    evaluates it at all H*W pixels on the host; here Qhull (scipy.spatial.Delaunay -- the triangulation LinearNDInterpolator builds
    internally from the same points) stays on the host and the H*W evaluations run on the device (rd_tri_raster).  `device`: a ROCm
    device (default: the current one); there is no host evaluation path."""
    import torch
    from scipy.spatial import Delaunay
    from . import engine
    depth_map = np.asarray(depth_map)
    assert depth_map.ndim == 2
    if validity_map is None:
        validity_map = depth_map > 0.0
    validity_map = np.asarray(validity_map)
    assert validity_map.ndim == 2
    rows, cols = depth_map.shape
    data_row_idx, data_col_idx = np.where(validity_map)
    depth_values = depth_map[data_row_idx, data_col_idx]
    if log_space:
        depth_values = np.log(depth_values)
    fill = 0.0 if not log_space else float(np.log(1e-3))
    tri = Delaunay(np.stack([data_row_idx, data_col_idx], axis=1))      # raises on fewer than 3 / collinear points, like the reference
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    simp = torch.from_numpy(np.ascontiguousarray(tri.simplices, dtype=np.int32)).to(dev)
    prow = torch.from_numpy(data_row_idx.astype(np.int32)).to(dev)
    pcol = torch.from_numpy(data_col_idx.astype(np.int32)).to(dev)
    vals = torch.from_numpy(np.ascontiguousarray(depth_values, dtype=np.float64)).to(dev)
    owner = torch.empty((rows, cols), dtype=torch.int32, device=dev)
    out = torch.empty((rows, cols), dtype=torch.float64, device=dev)
    engine._chk(engine.L().rd_tri_raster(engine._p(simp), engine._p(prow), engine._p(pcol), engine._p(vals), int(simp.shape[0]), rows, cols, fill,
                                         engine._p(owner), engine._p(out), engine._stream(out)), "rd_tri_raster")
    Z = out.cpu().numpy()
    if log_space:
        Z = np.exp(Z)
        Z[Z < 1e-1] = 0.0
    return Z


interpolate_depth_delft = interpolate_depth
