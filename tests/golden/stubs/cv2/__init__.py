"""Build-container stub so modules that `import cv2` at top level can be imported; no function is provided."""
INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_AREA = 3
INTER_CUBIC = 2
COLOR_BGR2RGB = 4
COLOR_GRAY2BGR = 8


def imread(path):
    """(tools/run_reference_sml_unchanged.py) 8-bit image file -> BGR uint8 array, as cv2.imread does."""
    import numpy as np
    from PIL import Image
    a = np.array(Image.open(path))
    if a.ndim == 3:
        a = a[:, :, ::-1]
    return np.ascontiguousarray(a)


def cvtColor(img, code):
    import numpy as np
    if code == COLOR_BGR2RGB:
        return np.ascontiguousarray(img[:, :, ::-1])
    if code == COLOR_GRAY2BGR:
        return np.stack([img] * 3, -1)
    raise NotImplementedError("cv2 stub: cvtColor code %r" % (code,))


def resize(src, dsize, interpolation=INTER_LINEAR):
    """INTER_NEAREST only (the one modules/midas/transforms.py:317-326 selects): src index = min(floor(dst * src_size / dst_size), src_size - 1)
    per axis (OpenCV's rule; cv2 itself is absent from this image -- SURVEY.md 8c item 3)."""
    import numpy as np
    if interpolation != INTER_NEAREST:
        raise NotImplementedError("cv2 stub: only INTER_NEAREST")
    w, h = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape[0], src.shape[1]
    ys = np.minimum(np.floor(np.arange(h) * (sh / h)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(w) * (sw / w)).astype(np.int64), sw - 1)
    return np.ascontiguousarray(src[ys][:, xs])
