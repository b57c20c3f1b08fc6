"""Depth PNG codec (riders_amd/data_utils.py) against fixture g12, produced by the REFERENCE's own save_depth / load_depth
(data/data_utils.py:94-143; generator tests/golden/make_golden.py g_depth_png).  Host logic only: no device work."""
import io
import os

import numpy as np

from riders_amd import data_utils

G = os.path.join(os.path.dirname(__file__), "golden", "g12_depth_png.npz")


def test_reads_the_reference_file_and_writes_an_equivalent_one(tmp_path):
    g = dict(np.load(G))
    # the file the reference wrote (PIL, adaptive scanline filters) decodes to the integers it stored and to its load_depth result
    assert np.array_equal(data_utils.decode_png16(g["png"].tobytes()), g["stored"])
    ref_path = tmp_path / "ref.png"
    ref_path.write_bytes(g["png"].tobytes())
    assert np.array_equal(data_utils.load_depth(str(ref_path)), g["loaded"])
    assert data_utils.load_depth(str(ref_path), data_format='CHW').shape == (1,) + g["loaded"].shape
    # quantisation = np.uint32(z * 256) as PIL stores it (0..65535), host path
    assert np.array_equal(data_utils.quantize_depth(g["z"]), g["stored"])
    # our file: same integers, same load_depth result, and PIL (the reference's reader) agrees
    ours = tmp_path / "ours.png"
    data_utils.save_depth(g["z"], str(ours))
    assert np.array_equal(data_utils.load_depth(str(ours)), g["loaded"])
    from PIL import Image
    im = Image.open(io.BytesIO(ours.read_bytes()))
    assert im.mode == "I;16" and np.array_equal(np.array(im).astype(np.uint16), g["stored"])
    assert np.array_equal(np.array(im, dtype=np.float32) / 256.0 * (np.array(im) > 0), g["loaded"])


def test_round_trip_properties():
    rs = np.random.RandomState(3)
    z = (rs.rand(64, 48) * 120).astype(np.float32)
    z[rs.rand(64, 48) < 0.5] = 0
    q = data_utils.quantize_depth(z)
    back = data_utils.decode_png16(data_utils.encode_png16(q))
    assert np.array_equal(back, q)
    d = back.astype(np.float32) / 256.0
    assert np.all(d <= z + 1e-6) and np.all(z - d < 1.0 / 256.0 + 1e-6)          # truncation: never above, less than one code below
    assert np.array_equal(d == 0, z < 1.0 / 256.0)
    assert data_utils.quantize_depth(np.array([[300.0, -2.0]], np.float32)).tolist() == [[65535, 0]]


def test_stage_handoff_layout(tmp_path):
    """run_rcnet_zju.py:159-171 writes where train_zju.py:114-117 reads: <result_root>/<interp>/<scene>/depth_predicted/<id>.png."""
    import os
    radar = ["/data/zju/scene_a/radar/000012.npy", "/data/zju/scene_a/radar/000003.npy"]
    out_root = os.path.join(str(tmp_path), "rcnet")
    z = np.full((4, 5), 2.5, np.float32)
    for r in radar:
        depth_p, color_p, resp_p = data_utils.rcnet_output_paths(out_root, r)
        assert depth_p == os.path.join(out_root, "scene_a", "depth_predicted", os.path.basename(r)[:-4] + ".png")
        assert os.path.dirname(color_p).endswith("depth_predicted_colors") and os.path.dirname(resp_p).endswith("response_predicted")
        os.makedirs(os.path.dirname(depth_p), exist_ok=True)
        data_utils.save_depth(z, depth_p)
    got = data_utils.sml_rcnet_input_paths(str(tmp_path), "rcnet", "scene_a")
    assert [os.path.basename(p) for p in got] == ["000003.png", "000012.png"]
    assert np.array_equal(data_utils.load_depth(got[0]), z)
