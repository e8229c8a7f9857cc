"""CPU restatement (numpy, fp64 / fp32) of the algorithm behind dynavsr_amd/csrc/conv2d_wino.hip -- TEST INFRASTRUCTURE ONLY:
only tests/ may import this module (the product path never does).

Winograd's minimal filtering F(2x2, 3x3) for the reference's `nn.Conv2d(cin, cout, 3, 1, 1)` calls (EDVR_arch.py:254-313,
arch_util.py:36-52): Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 2x2 output block, summed over the input channels in the
transformed domain -- the same three transforms, in the same order of operations, as pack_weights_wino_kernel (U = G g G^T),
the kernel's tf_rows / tf_cols (V = B^T d B) and its epilogue (A^T M A).  The test checks it against a plain correlation."""
import numpy as np

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def weight_transform(w):
    """w [cout][cin][3][3] -> U [4][4][cout][cin] (pack_weights_wino_kernel: row xi of G on the columns of g, then row nu)."""
    g = w.astype(np.float64)
    return np.einsum("xa,oiab,nb->xnoi", G, g, G)


def conv3x3_winograd(x, w, b=None, dtype=np.float64):
    """x [n][cin][h][w] (h, w even), w [cout][cin][3][3]; stride 1, zero pad 1.  dtype = arithmetic of the transformed domain."""
    n, cin, h, wd = x.shape
    cout = w.shape[0]
    assert h % 2 == 0 and wd % 2 == 0
    xp = np.zeros((n, cin, h + 2, wd + 2), dtype=np.float64)
    xp[:, :, 1:-1, 1:-1] = x
    U = weight_transform(w).astype(dtype)
    y = np.zeros((n, cout, h, wd), dtype=np.float64)
    for ty in range(h // 2):
        for tx in range(wd // 2):
            d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                      # [n][cin][4][4]
            t = np.einsum("xa,ncab->ncxb", BT, d)                                  # rows:    B^T d   (tf_rows)
            V = np.einsum("ncxb,vb->ncxv", t, BT).astype(dtype)                    # columns: (B^T d) B (tf_cols)
            M = np.einsum("xvoc,ncxv->noxv", U, V).astype(dtype)                   # 16 GEMMs over cin
            s = np.einsum("ix,noxv->noiv", AT, M.astype(np.float64))               # rows of A^T M
            y[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("noiv,jv->noij", s, AT)
    if b is not None:
        y += b.reshape(1, -1, 1, 1)
    return y
