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


# ---- F(4x4, 3x3): the transforms of dynavsr_amd/csrc/conv2d_wino5.hip (Lavin & Gray's matrices for the points 0, +-1, +-2, inf).
# pack_weights_wino5_kernel computes U = G4 g G4^T in fp64 and rounds once; the producer waves apply B4^T row by row (rows 1, 2 and
# 3, 4 as e +- o), the consumer waves' epilogue A4^T along nu in registers and along xi after the exchange.
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
              dtype=np.float64)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def split3_bf16(x):
    """x (float32 array) -> its three exact bf16 pieces (as float32): round-to-nearest-even conversions, residuals by exact
    subtraction -- the split of conv2d_wino3/4/5.hip (v_cvt_pk_bf16_f32)."""
    def rne(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return u.astype(np.uint32).view(np.float32)
    h = rne(x)
    r = (x - h).astype(np.float32)
    m = rne(r)
    return h, m, rne((r - m).astype(np.float32))


def conv3x3_winograd_f4(x, w, b=None, split=False):
    """x [n][cin][h][w] (h, w multiples of 4), w [cout][cin][3][3]; stride 1, zero pad 1.  split = False: fp64 throughout (the
    algebra); True: the kernel's arithmetic -- V in fp32, both operands as three bf16 pieces, the six partial products above
    2^-24 (Uh Vh, Um Vh, Uh Vm, Um Vm, Uh Vl, Ul Vh) accumulated in fp32, output transform in fp32."""
    n, cin, h, wd = x.shape
    cout = w.shape[0]
    assert h % 4 == 0 and wd % 4 == 0
    xp = np.zeros((n, cin, h + 2, wd + 2), dtype=np.float64)
    xp[:, :, 1:-1, 1:-1] = x
    U = np.einsum("xa,oiab,nb->xnoi", G4, w.astype(np.float64), G4)
    if split:
        U = U.astype(np.float32)
        Uh, Um, Ul = (p.astype(np.float64) for p in split3_bf16(U))
    y = np.zeros((n, cout, h, wd), dtype=np.float64)
    for ty in range(h // 4):
        for tx in range(wd // 4):
            d = xp[:, :, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]                      # [n][cin][6][6]
            t = np.einsum("xa,ncab->ncxb", BT4, d)
            V = np.einsum("ncxb,vb->ncxv", t, BT4)
            if split:
                V = V.astype(np.float32)
                Vh, Vm, Vl = (p.astype(np.float64) for p in split3_bf16(V))
                M = sum(np.einsum("xvoc,ncxv->noxv", a_, b_) for a_, b_ in ((Uh, Vh), (Um, Vh), (Uh, Vm), (Um, Vm), (Uh, Vl), (Ul, Vh)))
                M = M.astype(np.float32).astype(np.float64)
            else:
                M = np.einsum("xvoc,ncxv->noxv", U, V)
            s = np.einsum("ix,noxv->noiv", AT4, M)
            y[:, :, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = np.einsum("noiv,jv->noij", s, AT4)
    if b is not None:
        y += b.reshape(1, -1, 1, 1)
    return y
