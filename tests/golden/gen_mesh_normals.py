#!/usr/bin/env python3
"""Golden vectors for the geometric normal the reference shades with (a9 of SURVEY.md section 8a).

The reference triangulates the predicted depth into a mesh (`depth_file_to_mesh` -> `detect_boundary_points`,
myutils/mesh_recon.py:41-74,86-331: gap closing at grazing triangles, two triangles per 2x2 cell, duplicated vertices), rotates it
180 degrees about x (`rotate_mesh_around_x`, inverse_img_w_mi.py:721-727) and lets Mitsuba shade with the face normal at the hit
point.  Pixel centres coincide with mesh vertices, so the normal a pixel sees is a mixture of the faces around its vertex: recorded
here is the AREA-WEIGHTED mean of the adjacent face normals per pixel, plus the mesh itself (vertex / triangle counts).

Runs the reference's own function under an `open3d` stub that only stores what it is given.  Inputs + outputs only:

    python tests/golden/gen_mesh_normals.py        # writes tests/golden/mesh_normals.npz
"""
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MATPBR_REFERENCE", "/root/reference")


def install_stubs():
    o3d = types.ModuleType("open3d")

    class Intr:
        def __init__(self, width, height, fx, fy, cx, cy):
            self.width, self.height = width, height
            self.intrinsic_matrix = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])

    class IVec(list):
        pass

    class Mesh:
        def __init__(self, points, indices):
            self.vertices, self.triangles = np.asarray(points, dtype=np.float64), np.asarray(indices, dtype=np.int64)

        def rotate(self, R, center=(0, 0, 0)):
            self.vertices = self.vertices @ np.asarray(R).T

    o3d.camera = types.SimpleNamespace(PinholeCameraIntrinsic=Intr)
    o3d.utility = types.SimpleNamespace(Vector3iVector=IVec, Vector3dVector=lambda a: np.asarray(a))
    o3d.geometry = types.SimpleNamespace(PointCloud=lambda p: p, TriangleMesh=Mesh)
    mods = {"open3d": o3d, "mitsuba": types.ModuleType("mitsuba")}
    sk = types.ModuleType("skimage")
    skio = types.ModuleType("skimage.io")
    skio.imread = None
    skt = types.ModuleType("skimage.transform")
    skt.resize = None
    mods.update({"skimage": sk, "skimage.io": skio, "skimage.transform": skt})
    tq = types.ModuleType("tqdm")

    class _T:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def update(self, n=1):
            pass

    tq.tqdm = _T
    mods["tqdm"] = tq
    sys.modules.update(mods)
    sys.path.insert(0, REF)


def make_depth(H, W, seed=3):
    """MaterialNet-style (inverse) depth: smooth bumps with a raised foreground block (depth edges on all four sides)."""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d = np.full((H, W), 1.0)
    for _ in range(4):
        A, sig = rng.uniform(-0.15, 0.15), rng.uniform(6, 14)
        ci, cj = rng.uniform(0, H), rng.uniform(0, W)
        d += A * np.exp(-((ii - ci) ** 2 + (jj - cj) ** 2) / (2 * sig * sig))
    d[12:26, 15:30] += 0.6
    return d.astype(np.float32)


def main():
    install_stubs()
    import myutils.mesh_recon as MR

    H = W = 40
    fov = 35.0
    f = (W / 2) / math.tan(math.radians(fov) / 2)
    K = np.array([[f, 0, (W - 1) / 2], [0, f, (H - 1) / 2], [0, 0, 1.0]])
    pred = make_depth(H, W)
    depth = 2 * pred.max() - pred                                    # inverse_img_w_mi.py:722
    mesh, _ = MR.depth_file_to_mesh(depth.copy(), cameraMatrix=K, minAngle=6, sun3d=False, depthScale=1.0)   # :726
    mesh = MR.rotate_mesh_around_x(mesh, 180)                         # :727
    V, T = mesh.vertices, mesh.triangles
    fn = np.cross(V[T[:, 1]] - V[T[:, 0]], V[T[:, 2]] - V[T[:, 0]])   # 2 * area * unit normal
    acc = np.zeros((V.shape[0], 3))
    for k in range(3):
        np.add.at(acc, T[:, k], fn)
    acc = acc[: H * W].reshape(H, W, 3)                               # grid vertices only (duplicates belong to depth edges)
    # orient towards the camera at the origin (the winding of :190 is consistent, the sign is fixed here once)
    P = V[: H * W].reshape(H, W, 3)
    sign = np.where((acc * P).sum(-1, keepdims=True) > 0, -1.0, 1.0)
    ln = np.linalg.norm(acc, axis=-1, keepdims=True)
    nrm = np.where(ln > 0, sign * acc / np.maximum(ln, 1e-30), 0.0)
    np.savez_compressed(os.path.join(HERE, "mesh_normals.npz"), depth_pred=pred, depth_mesh_input=depth, fov_x_deg=fov,
                        n_vertices=V.shape[0], n_triangles=T.shape[0], grid_positions=P.astype(np.float32),
                        vertex_normal_area_weighted=nrm.astype(np.float32), has_faces=(ln[..., 0] > 0),
                        vertices=V.astype(np.float64), triangles=T.astype(np.int32))
    print("vertices", V.shape[0], "(grid", H * W, ") triangles", T.shape[0], "of", 2 * (H - 1) * (W - 1), "pixels with faces", int((ln > 0).sum()))


if __name__ == "__main__":
    main()
