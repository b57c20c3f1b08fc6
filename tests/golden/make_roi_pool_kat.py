"""Writes tests/golden/roi_pool_kat.json.  Every expected value below is a LITERAL worked out by hand from torchvision's published
roi_pool semantics; nothing here calls a roi_pool implementation."""
import json

def inc(H, W):        # in[h][w] = h*W + w  (strictly increasing in scan order: the max of a window is its bottom-right element)
    return [[float(h * W + w) for w in range(W)] for h in range(H)]

def dec(H, W):        # in[h][w] = 100 - (h*W + w)  (the max of a window is its top-left element)
    return [[float(100 - (h * W + w)) for w in range(W)] for h in range(H)]

cases = []
cases.append(dict(
    name="integer_box_scale1",
    why="sw=sh=1, ew=eh=4 -> rw=rh=4 (the +1), 2x2 bins of size 2: rows {1,2},{3,4}, cols {1,2},{3,4}; increasing map -> bottom-right of each bin",
    input=[[inc(6, 8)]], rois=[[0, 1, 1, 4, 4]], scale=1.0, output_size=[2, 2],
    out=[[[[18.0, 20.0], [34.0, 36.0]]]], argmax=[[[[18, 20], [34, 36]]]]))
cases.append(dict(
    name="half_rounds_away_from_zero_positive",
    why="scale 0.5: x1=3->1.5->2, y1=1->0.5->1, x2=9->4.5->5, y2=5->2.5->3 (C round; banker's rounding would give 2,0,4,2). rw=4, rh=3; "
        "PH=3,PW=2: bh=1 -> rows {1},{2},{3}; bw=2 -> cols {2,3},{4,5}",
    input=[[inc(6, 8)]], rois=[[0, 3, 1, 9, 5]], scale=0.5, output_size=[3, 2],
    out=[[[[11.0, 13.0], [19.0, 21.0], [27.0, 29.0]]]], argmax=[[[[11, 13], [19, 21], [27, 29]]]]))
cases.append(dict(
    name="half_rounds_away_from_zero_negative_and_empty_bins",
    why="scale 0.5: x1=-1->-0.5->-1, y1=-3->-1.5->-2 (floor(x+0.5) would give 0,-1), x2=5->2.5->3, y2=3->1.5->2. rw=rh=5, 5x5 bins of size 1: "
        "row bin ph covers map row ph-2, col bin pw covers map col pw-1; bins off the map are clamped to empty -> value 0, argmax -1",
    input=[[inc(6, 8)]], rois=[[0, -1, -3, 5, 3]], scale=0.5, output_size=[5, 5],
    out=[[[[0.0] * 5, [0.0] * 5, [0.0, 0.0, 1.0, 2.0, 3.0], [0.0, 8.0, 9.0, 10.0, 11.0], [0.0, 16.0, 17.0, 18.0, 19.0]]]],
    argmax=[[[[-1] * 5, [-1] * 5, [-1, 0, 1, 2, 3], [-1, 8, 9, 10, 11], [-1, 16, 17, 18, 19]]]]))
cases.append(dict(
    name="box_leaves_map_bottom_right",
    why="sw=6, sh=4, ew=11, eh=9 -> rw=rh=6, 3x3 bins of size 2: rows {4,5},{6,7},{8,9} and cols {6,7},{8,9},{10,11} clamped to H=6, W=8: only bin (0,0) "
        "is non-empty -> in[5][7]",
    input=[[inc(6, 8)]], rois=[[0, 6, 4, 11, 9]], scale=1.0, output_size=[3, 3],
    out=[[[[47.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]]], argmax=[[[[47, -1, -1], [-1, -1, -1], [-1, -1, -1]]]]))
cases.append(dict(
    name="ties_first_maximum_wins",
    why="constant map: strict '>' scan in row-major order keeps the FIRST element of each bin: rows {0,1},{2,3} x cols {0,1},{2,3} -> 0, 2, 8, 10",
    input=[[[[1.0] * 4 for _ in range(4)]]], rois=[[0, 0, 0, 3, 3]], scale=1.0, output_size=[2, 2],
    out=[[[[1.0, 1.0], [1.0, 1.0]]]], argmax=[[[[0, 2], [8, 10]]]]))
cases.append(dict(
    name="overlapping_fractional_bins",
    why="sw=1, sh=2, ew=5, eh=4 -> rw=5, rh=3; PW=2: bw=2.5 -> cols [1+floor(0), 1+ceil(2.5)) = {1,2,3} and [1+floor(2.5), 1+ceil(5)) = {3,4,5} "
        "(col 3 in both); PH=2: bh=1.5 -> rows {2,3} and {3,4}; decreasing map -> top-left of each bin",
    input=[[dec(6, 8)]], rois=[[0, 1, 2, 5, 4]], scale=1.0, output_size=[2, 2],
    out=[[[[83.0, 81.0], [75.0, 73.0]]]], argmax=[[[[17, 19], [25, 27]]]],
    grad_out=[[[[1.0, 2.0], [3.0, 4.0]]]], grad_in_nonzero=[[0, 0, 17, 1.0], [0, 0, 19, 2.0], [0, 0, 25, 3.0], [0, 0, 27, 4.0]]))
cases.append(dict(
    name="inverted_box_has_extent_one",
    why="x2<x1, y2<y1: rw=max(2-5+1,1)=1, rh=max(1-3+1,1)=1; 2x2 bins of size 0.5: [floor(0),ceil(.5))=[0,1) and [floor(.5),ceil(1))=[0,1) -> every bin "
        "is the single pixel (h=3,w=5); backward: the four output gradients add up on that pixel",
    input=[[inc(6, 8)]], rois=[[0, 5, 3, 2, 1]], scale=1.0, output_size=[2, 2],
    out=[[[[29.0, 29.0], [29.0, 29.0]]]], argmax=[[[[29, 29], [29, 29]]]],
    grad_out=[[[[1.0, 1.0], [1.0, 1.0]]]], grad_in_nonzero=[[0, 0, 29, 4.0]]))
img = lambda b: [[[1000.0 * b + 100.0 * c + v for v in row] for row in inc(6, 8)] for c in range(2)]
cases.append(dict(
    name="batch_index_and_channels",
    why="two images x two channels, the RoI names image 1; one bin over the whole map -> bottom-right pixel (index 47) of image 1, per channel",
    input=[img(0), img(1)], rois=[[1, 0, 0, 7, 5]], scale=1.0, output_size=[1, 1],
    out=[[[[1047.0]], [[1147.0]]]], argmax=[[[[47]], [[47]]]]))
cases.append(dict(
    name="zju_latent_geometry",
    why="RC-Net latent pooling (networks.py:418-422): 100x240 box centred on padded pixel (306,248): (256,128,356,368) * 1/32 = (8, 4, 11.125, 11.5) -> "
        "sw=8, sh=4, ew=11, eh=12 (11.5 rounds away from zero) -> rw=4, rh=9; PH=7: bh=9/7: row bins [4,6) [5,7) [6,8) [7,10) [9,11) [10,12) [11,13) "
        "(floor(k*9/7), ceil((k+1)*9/7)); PW=3: bw=4/3: col bins [8,10) [9,11) [10,12); 16x20 increasing map -> (he-1)*20 + (we-1)",
    input=[[inc(16, 20)]], rois=[[0, 256, 128, 356, 368]], scale=1.0 / 32.0, output_size=[7, 3],
    out=[[[[109.0, 110.0, 111.0], [129.0, 130.0, 131.0], [149.0, 150.0, 151.0], [189.0, 190.0, 191.0], [209.0, 210.0, 211.0],
           [229.0, 230.0, 231.0], [249.0, 250.0, 251.0]]]],
    argmax=[[[[109, 110, 111], [129, 130, 131], [149, 150, 151], [189, 190, 191], [209, 210, 211], [229, 230, 231], [249, 250, 251]]]]))
# the same two geometries over 32 channels (whole 16-byte channel vectors: the vectorised forward and the gather / LDS-tile backward kernels):
# channel c holds the plane + c, so every channel has the argmax derived above; its output gradient is the single-channel one times (c + 1)
cases.append(dict(
    name="overlapping_fractional_bins_32_channels",
    why="per channel identical to overlapping_fractional_bins (adding c to a plane does not move its maxima)",
    input=[[[[v + c for v in row] for row in dec(6, 8)] for c in range(32)]], rois=[[0, 1, 2, 5, 4]], scale=1.0, output_size=[2, 2],
    out=[[[[83.0 + c, 81.0 + c], [75.0 + c, 73.0 + c]] for c in range(32)]], argmax=[[[[17, 19], [25, 27]] for c in range(32)]],
    grad_out=[[[[1.0 * (c + 1), 2.0 * (c + 1)], [3.0 * (c + 1), 4.0 * (c + 1)]] for c in range(32)]],
    grad_in_nonzero=[[0, c, i, g * (c + 1)] for c in range(32) for i, g in ((17, 1.0), (19, 2.0), (25, 3.0), (27, 4.0))]))
cases.append(dict(
    name="inverted_box_32_channels_two_rois",
    why="per channel identical to inverted_box_has_extent_one; the RoI is listed twice, so eight unit gradients meet on pixel 29 of every channel",
    input=[[[[v + c for v in row] for row in inc(6, 8)] for c in range(32)]], rois=[[0, 5, 3, 2, 1], [0, 5, 3, 2, 1]], scale=1.0, output_size=[2, 2],
    out=[[[[29.0 + c] * 2] * 2 for c in range(32)]] * 2, argmax=[[[[29, 29], [29, 29]] for c in range(32)]] * 2,
    grad_out=[[[[1.0, 1.0], [1.0, 1.0]] for c in range(32)]] * 2, grad_in_nonzero=[[0, c, 29, 8.0] for c in range(32)]))
json.dump(dict(
    note="Hand-derived known-answer vectors for torchvision.ops.roi_pool (0.14 semantics; call sites RCNet/networks.py:418-433). "
         "Layout NCHW; rois rows (batch, x1, y1, x2, y2); argmax = h*W + w inside the image/channel plane, -1 for an empty bin.",
    cases=cases), open("/root/repo/tests/golden/roi_pool_kat.json", "w"), indent=1)
print(len(cases), "cases")
