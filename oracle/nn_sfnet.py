"""fp32 PyTorch restatement of the face-feature network and face alignment behind the reference's face-realism loss term
(exp-1-debias-gender/1-main-debias.py:968-994 load, :292-312 ``image_pipeline``, :1176-1190 ``get_face_feats``,
:80-117 ``FaceFeatsModel.semantic_search``, :1917-1929 use).  TEST ORACLE.

* ``SFNet20`` -- opensphere ``model/backbone/sfnet.py:123-202`` (``sfnet20``: BasicBlock x [1,2,4,1], no norm layer, ReLU, 112x112
  input, fc 512*7*7 -> 512).  PINNED: opensphere is vendored under /root/reference and importable; tests/golden/make_golden.py runs the
  reference's own module on seeded weights/input and stores the output (tests/golden/reference_sfnet20.json).
* ``umeyama`` / ``warp_affine`` -- scikit-image==0.22.0 ``SimilarityTransform.estimate`` and kornia==0.7.1
  ``geometry.transform.warp_affine`` (environment.yml:148,183) are neither vendored nor installed: PARITY UNPINNED, restated from the
  published algorithms (Umeyama 1991; kornia normalises the pixel homography with (size-1) and samples with
  ``F.affine_grid`` / ``F.grid_sample`` at the caller's ``align_corners``).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

SRC_LANDMARKS = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]])  # :297-303


class _ConvBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=True)

    def forward(self, x):
        return F.relu(self.conv1(x))


class _BasicBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=True)

    def forward(self, x):
        return F.relu(self.conv2(F.relu(self.conv1(x))) + x)


class SFNet20(nn.Module):
    LAYERS = (1, 2, 4, 1)

    def __init__(self, channels=(64, 128, 256, 512), out_channel=512, in_size=112):
        super().__init__()
        cin = 3
        for i, (c, n) in enumerate(zip(channels, self.LAYERS)):
            setattr(self, f"layer{i + 1}", nn.Sequential(_ConvBlock(cin, c, 2), *[_BasicBlock(c) for _ in range(n)]))
            cin = c
        self.fc = nn.Linear(channels[3] * (in_size // 16) ** 2, out_channel)

    def forward(self, x):
        for i in range(4):
            x = getattr(self, f"layer{i + 1}")(x)
        return self.fc(torch.flatten(x, 1))


def get_face_feats(net, data, flip=True, normalize=True):
    """:1176-1190 -- features of the chip plus its horizontal mirror, fp32, L2-normalised."""
    feats = net(data)
    if flip:
        feats = feats + net(torch.flip(data, [3]))
    feats = feats.float()
    return F.normalize(feats, dim=-1) if normalize else feats


def umeyama(src, dst):
    """skimage ``_umeyama(src, dst, estimate_scale=True)``: least-squares similarity transform (3x3) mapping src -> dst points [n,2]."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    num, dim = src.shape
    src_mean, dst_mean = src.mean(axis=0), dst.mean(axis=0)
    sd, dd = src - src_mean, dst - dst_mean
    A = dd.T @ sd / num
    d = np.ones((dim,))
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.nan * T
    if rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / sd.var(axis=0).sum() * (S @ d)
    T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
    T[:dim, :dim] *= scale
    return T


def _normal_transform_pixel(h, w):
    return torch.tensor([[2.0 / (w - 1), 0.0, -1.0], [0.0, 2.0 / (h - 1), -1.0], [0.0, 0.0, 1.0]], dtype=torch.float64)


def warp_affine(src, M, dsize, align_corners=False):
    """kornia.geometry.transform.warp_affine(src [B,C,H,W], M [B,2,3], dsize=(h,w), 'bilinear', 'zeros', align_corners)."""
    B, C, H, W = src.shape
    M3 = torch.cat([M.double(), torch.tensor([[[0.0, 0.0, 1.0]]], dtype=torch.float64).expand(B, 1, 3)], dim=1)
    dst_norm_trans_src_norm = _normal_transform_pixel(dsize[0], dsize[1]) @ M3 @ torch.inverse(_normal_transform_pixel(H, W))
    src_norm_trans_dst_norm = torch.inverse(dst_norm_trans_src_norm).to(src.dtype)
    grid = F.affine_grid(src_norm_trans_dst_norm[:, :2, :], [B, C, dsize[0], dsize[1]], align_corners=align_corners)
    return F.grid_sample(src, grid, align_corners=align_corners, mode="bilinear", padding_mode="zeros")


def image_pipeline(img, tgz_landmark, crop=112):
    """:292-312 -- img [3,H,W] in [-1,1], landmarks [5,2] (x,y) in image pixels -> aligned chip [3,crop,crop] in [-1,1]."""
    x = (img + 1) / 2.0 * 255
    T = umeyama(tgz_landmark, SRC_LANDMARKS * (crop / 112.0))
    M = torch.tensor(T[0:2, :]).unsqueeze(0).to(img.dtype)
    face = warp_affine(x.unsqueeze(0), M, (crop, crop), align_corners=False).squeeze(0)
    return (face / 255.0) * 2 - 1


def semantic_search(face_feats_db, query):
    """FaceFeatsModel.semantic_search (:98-117): the database row with the largest dot product, detached."""
    with torch.no_grad():
        idx = (query.float() @ face_feats_db.float().t()).argmax(dim=-1)
        return face_feats_db[idx].detach().clone()
