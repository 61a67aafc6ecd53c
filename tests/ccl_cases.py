"""Hand-made mask cases shared by the oracle tests and the GPU parity tests."""
import numpy as np


def hand_cases():
    cases = {}
    cases["empty"] = np.zeros((6, 8), np.uint8)
    cases["full"] = np.ones((6, 8), np.uint8)
    m = np.zeros((6, 8), np.uint8); m[3, 5] = 1
    cases["single_px"] = m
    m = np.zeros((8, 8), np.uint8)
    for i in range(8):
        m[i, i] = 1
    cases["diagonal_chain"] = m                      # one 8-connected component
    m = np.zeros((8, 8), np.uint8)
    for i in range(8):
        m[i, 7 - i] = 1
    cases["anti_diagonal_chain"] = m
    yy, xx = np.mgrid[0:7, 0:9]
    cases["checkerboard"] = ((yy + xx) % 2).astype(np.uint8)   # 8-conn: a single component
    m = np.zeros((7, 9), np.uint8); m[::2, ::2] = 1
    cases["isolated_grid"] = m                                   # every pixel its own component
    # two blobs whose pixel-raster order and 2x2-block-raster order differ:
    # A first appears at (row 1, col 6); B at (row 0... no: B at (row 1, col 0) is later in
    # pixel raster than A at (row 0, col 7)?  Use: A = pixel (1, 1) [block row 0], B = pixel (0, 6)
    # [block row 0, later block col]: pixel-raster order = B, A ; block-raster order = A, B.
    m = np.zeros((4, 8), np.uint8); m[1, 1] = 1; m[0, 6] = 1
    cases["order_block_vs_pixel"] = m
    m = np.zeros((5, 7), np.uint8); m[4, :] = 1; m[:, 6] = 1; m[0, 0] = 1
    cases["odd_dims_border"] = m
    # U shape: two arms that only join at the bottom (forces a late merge of two provisional labels)
    m = np.zeros((8, 10), np.uint8); m[0:7, 1] = 1; m[0:7, 8] = 1; m[7, 1:9] = 1
    cases["u_shape"] = m
    # serpentine: long snake through the whole frame
    m = np.zeros((9, 12), np.uint8)
    for r in range(0, 9, 2):
        m[r, :] = 1
    for k, r in enumerate(range(1, 9, 2)):
        m[r, 11 if k % 2 == 0 else 0] = 1
    cases["serpentine"] = m
    # area threshold edge (== threshold must be kept): components of 2, 3, 4 pixels
    m = np.zeros((6, 12), np.uint8); m[0, 0:2] = 1; m[2, 4:7] = 1; m[4, 8:12] = 1
    cases["area_edge"] = m
    # diagonal touch between 2x2 blocks only via corners
    m = np.zeros((6, 6), np.uint8); m[1, 1] = 1; m[2, 2] = 1; m[3, 1] = 1; m[0, 2] = 1
    cases["corner_touch"] = m
    return cases
