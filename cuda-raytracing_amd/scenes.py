"""Deterministic synthetic OBJ scenes for the raycast hot path.

No bunny / Sponza asset exists offline, so the bench and parity workloads are
procedural meshes written as plain ``v`` / ``vt`` / ``f a/a b/b c/c`` OBJ text
(the subset the reference's OBJLoader.hpp:36-172 parses).  Every generator is
byte-deterministic: the same arguments always give the same file.
"""
import os
import numpy as np

# Reference demo camera (kernel.cu:158-164), quoted for 1920x1080.
K_1080 = (862.097835972576, 0.0, 998.1702383680802,
          0.0, 862.1368447300727, 569.6759403225842,
          0.0, 0.0, 1.0)
D_REF = (0.016233999489849514, -0.013875757716177956,
         0.03264329940126211, -0.019561619947134234)


def scaled_K(width, ref_width=1920):
    """K of kernel.cu:160-164 scaled by width/1920 (fx, fy, cx, cy all scale)."""
    s = width / float(ref_width)
    k = list(K_1080)
    for i in (0, 2, 4, 5):
        k[i] *= s
    return tuple(k)


def _write_grid_obj(path, comment, X, Y, Z, U, V, closed_poles):
    """Write a (rows+1)x(cols+1) vertex grid as v / vt / f lines.

    Faces per quad (a=row j col i, b=a+1, c=a+cols+1, d=c+1):
    ``f a c b`` and ``f b c d``; with closed_poles the first row keeps only
    ``b c d`` and the last row only ``a c b`` (the other one is degenerate).
    """
    rows, cols = X.shape[0] - 1, X.shape[1] - 1
    out = ["# %s\n" % comment]
    xs, ys, zs = X.ravel(), Y.ravel(), Z.ravel()
    out.extend("v %.6f %.6f %.6f\n" % (xs[n], ys[n], zs[n]) for n in range(xs.size))
    us, vs = U.ravel(), V.ravel()
    out.extend("vt %.6f %.6f\n" % (us[n], vs[n]) for n in range(us.size))
    for j in range(rows):
        for i in range(cols):
            a = j * (cols + 1) + i + 1
            b = a + 1
            c = a + cols + 1
            d = c + 1
            if not (closed_poles and j == 0):
                out.append("f %d/%d %d/%d %d/%d\n" % (a, a, c, c, b, b))
            if not (closed_poles and j == rows - 1):
                out.append("f %d/%d %d/%d %d/%d\n" % (b, b, c, c, d, d))
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.writelines(out)
    os.replace(tmp, path)


def write_blob_obj(path, nu=188, nv=187, seed=1234):
    """"Bunny-class" bumpy UV sphere; nu=188, nv=187 -> 69 936 triangles.

    r = 1 + 0.04 * sum_k sin(3(k+2)x+p_k0) sin(3(k+2)y+p_k1) sin(3(k+2)z+p_k2)
    on the unit-sphere point (x, y, z); outward winding.
    """
    p = np.random.default_rng(seed).uniform(0, 2 * np.pi, (6, 3))
    j = np.arange(nv + 1)[:, None]
    i = np.arange(nu + 1)[None, :]
    th = np.pi * j / nv
    ph = 2 * np.pi * i / nu
    x = np.sin(th) * np.cos(ph)
    y = np.sin(th) * np.sin(ph)
    z = np.cos(th) * np.ones_like(ph)
    r = np.ones_like(x)
    for k in range(6):
        f = (k + 2) * 3
        r = r + 0.04 * np.sin(f * x + p[k, 0]) * np.sin(f * y + p[k, 1]) * np.sin(f * z + p[k, 2])
    U = (i / nu) * np.ones_like(th)
    V = (1 - j / nv) * np.ones_like(ph)
    _write_grid_obj(path, "synthetic blob", x * r, y * r, z * r, U, V, closed_poles=True)
    return 2 * nu * (nv - 1)


def blob_dims_for(n_tris):
    """(nu, nv) for the named blob sizes."""
    table = {69936: (188, 187), 5000: (50, 51)}
    return table[n_tris]


def write_atrium_obj(path, seed=4321, cols_x=6, cols_y=10, col_seg=48, col_rings=26, wall_div=96):
    """"Sponza-class" procedural atrium: floor, ceiling, 4 walls (tessellated,
    inward-facing) plus cols_x*cols_y fluted columns; the camera sits inside, so
    depth complexity is >= 5 along most rays.  Defaults give 260 352 triangles.
    Written as independent grids in one OBJ (v/vt indices are global).
    """
    rng = np.random.default_rng(seed)
    verts, uvs, faces = [], [], []                                 # per grid: [n, 3] positions, [n, 2] uvs, [m, 3] 1-based indices

    def add_grid(P, U, V, flip):
        rows, cols = P.shape[0] - 1, P.shape[1] - 1
        base = sum(len(v) for v in verts)
        verts.append(P.reshape(-1, 3))
        uvs.append(np.stack([U, V], -1).reshape(-1, 2))
        jj, ii = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
        a = (base + jj * (cols + 1) + ii + 1).ravel()
        b, c = a + 1, a + cols + 1
        d = c + 1
        # two triangles per quad, quads row by row: (a b c)(b d c) when flipped, (a c b)(b c d) otherwise
        quad = np.stack([a, b, c, b, d, c] if flip else [a, c, b, b, c, d], -1)
        faces.append(quad.reshape(-1, 3))

    LX, LY, LZ = 12.0, 20.0, 8.0
    s = np.linspace(0, 1, wall_div + 1)
    S, T = np.meshgrid(s, s, indexing="xy")
    bump = lambda A, B: 0.03 * np.sin(17 * A + rng.uniform(0, 6.28)) * np.sin(13 * B + rng.uniform(0, 6.28))
    # floor (normal +z) and ceiling (normal -z)
    add_grid(np.stack([(S - .5) * LX, (T - .5) * LY, bump(S, T)], -1), S, T, flip=True)
    add_grid(np.stack([(S - .5) * LX, (T - .5) * LY, LZ + bump(S, T)], -1), S, T, flip=False)
    # walls x=-LX/2 (normal +x), x=+LX/2 (normal -x), y=-LY/2 (normal +y), y=+LY/2 (normal -y)
    add_grid(np.stack([-LX / 2 + bump(S, T), (S - .5) * LY, T * LZ], -1), S, T, flip=True)
    add_grid(np.stack([LX / 2 + bump(S, T), (S - .5) * LY, T * LZ], -1), S, T, flip=False)
    add_grid(np.stack([(S - .5) * LX, -LY / 2 + bump(S, T), T * LZ], -1), S, T, flip=False)
    add_grid(np.stack([(S - .5) * LX, LY / 2 + bump(S, T), T * LZ], -1), S, T, flip=True)
    # fluted columns (outward normals)
    a = np.linspace(0, 2 * np.pi, col_seg + 1)
    h = np.linspace(0, 1, col_rings + 1)
    A, H = np.meshgrid(a, h, indexing="xy")
    for cx in range(cols_x):
        for cy in range(cols_y):
            x0 = (cx + 0.5) / cols_x * LX * 0.8 - LX * 0.4
            y0 = (cy + 0.5) / cols_y * LY * 0.9 - LY * 0.45
            if abs(x0) < 1.2:
                x0 += 1.5 if x0 >= 0 else -1.5
            rad = 0.28 + 0.05 * rng.uniform() + 0.02 * np.cos(8 * A) + 0.06 * (H - .5) ** 2
            P = np.stack([x0 + rad * np.cos(A), y0 + rad * np.sin(A), H * LZ], -1)
            add_grid(P, A / (2 * np.pi), H, flip=True)

    verts, uvs, faces = np.concatenate(verts), np.concatenate(uvs), np.concatenate(faces)
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write("# synthetic atrium\n")
        # (text in pieces: the 4 M-triangle scale of workload c6 would otherwise hold every line as a Python string at once)
        for fmt, rows in (("v %.6f %.6f %.6f\n", verts.tolist()), ("vt %.6f %.6f\n", uvs.tolist())):
            for k in range(0, len(rows), 1 << 18):
                f.write("".join(fmt % tuple(r) for r in rows[k:k + (1 << 18)]))
        for k in range(0, len(faces), 1 << 18):
            f.write("".join("f %d/%d %d/%d %d/%d\n" % (q[0], q[0], q[1], q[1], q[2], q[2]) for q in faces[k:k + (1 << 18)].tolist()))
    os.replace(tmp, path)
    return len(faces)


def _write_grids_obj(path, comment, grids):
    """grids: list of (P[rows+1, cols+1, 3], U, V, flip) -- independent tessellated sheets in one OBJ; (a b c)(b d c) per quad when
    flipped, (a c b)(b c d) otherwise (a = row j col i, b = a + 1, c = a + cols + 1, d = c + 1)."""
    out = ["# %s\n" % comment]
    faces, base = [], 0
    for P, U, V, flip in grids:
        rows, cols = P.shape[0] - 1, P.shape[1] - 1
        out.extend("v %.6f %.6f %.6f\n" % tuple(r) for r in P.reshape(-1, 3).tolist())
        jj, ii = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
        a = (base + jj * (cols + 1) + ii + 1).ravel()
        b, c = a + 1, a + cols + 1
        d = c + 1
        faces.append(np.stack([a, b, c, b, d, c] if flip else [a, c, b, b, c, d], -1).reshape(-1, 3))
        base += P.shape[0] * P.shape[1]
    for P, U, V, flip in grids:
        out.extend("vt %.6f %.6f\n" % tuple(r) for r in np.stack([U, V], -1).reshape(-1, 2).tolist())
    faces = np.concatenate(faces)
    out.extend("f %d/%d %d/%d %d/%d\n" % (q[0], q[0], q[1], q[1], q[2], q[2]) for q in faces.tolist())
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.writelines(out)
    os.replace(tmp, path)
    return len(faces)


def write_demo_objs(area_path, board_path, offset=(0.0, 0.0, 0.0), seed=99):
    """Stand-ins for the two meshes of the reference's demo (kernel.cu:209-210: calibration_area.obj and calibration_board.obj are
    not in the repository): a "calibration area" -- a floor and three walls, open towards the demo camera at (-1, -4, 2), slightly
    uneven so that the tree is not degenerate -- of 32 768 triangles, and a "calibration board" -- an upright 1.2 m x 0.9 m sheet
    facing -y, 3 456 triangles -- in its own coordinates; the demo places it with MeshInstance::pose = (-0.6, 1.48, 0.73)
    (kernel.cu:229-232).  offset: added to the board's vertices (the baked twin of the scene: the same board as an identity instance)."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 6.28, 8)
    n = 64
    s = np.linspace(0, 1, n + 1)
    S, T = np.meshgrid(s, s, indexing="xy")
    bump = lambda A, B, k: 0.02 * np.sin(11 * A + ph[k]) * np.sin(9 * B + ph[k + 1])
    X0, X1, Y0, Y1, Z1 = -4.0, 4.0, -1.0, 7.0, 3.0
    grids = [
        (np.stack([X0 + S * (X1 - X0), Y0 + T * (Y1 - Y0), bump(S, T, 0)], -1), S, T, True),                      # floor, normal +z
        (np.stack([X0 + S * (X1 - X0), Y1 + bump(S, T, 2), T * Z1], -1), S, T, True),                             # back wall, normal -y
        (np.stack([X0 + bump(S, T, 4), Y0 + S * (Y1 - Y0), T * Z1], -1), S, T, True),                             # left wall, normal +x
        (np.stack([X1 + bump(S, T, 6), Y0 + S * (Y1 - Y0), T * Z1], -1), S, T, False)]                            # right wall, normal -x
    n_area = _write_grids_obj(area_path, "synthetic calibration area", grids)
    u = np.linspace(0, 1, 49)
    v = np.linspace(0, 1, 37)
    U, V = np.meshgrid(u, v, indexing="xy")
    board = np.stack([(U - 0.5) * 1.2 + offset[0], 0.0 * U + offset[1], (V - 0.5) * 0.9 + offset[2]], -1)
    n_board = _write_grids_obj(board_path, "synthetic calibration board", [(board, U, V, True)])
    return n_area, n_board


def demo_textures(seed=5):
    """BGR stand-ins for calibration_area.jpg / calibration_board.jpg (kernel.cu:192,204): a tiled, shaded pattern and a 9 x 7 checkerboard."""
    rng = np.random.default_rng(seed)
    h, w = 1024, 1024
    yy, xx = np.mgrid[0:h, 0:w]
    area = np.zeros((h, w, 3), np.uint8)
    tile = ((xx // 64 + yy // 64) % 2).astype(np.uint8)
    area[..., 0] = 60 + 120 * tile + (xx * 60 // w)
    area[..., 1] = 90 + 100 * (1 - tile) + (yy * 50 // h)
    area[..., 2] = 140 + rng.integers(0, 40, (h, w), dtype=np.uint8)
    bh, bw = 540, 720
    yy, xx = np.mgrid[0:bh, 0:bw]
    sq = (((xx * 9) // bw + (yy * 7) // bh) % 2).astype(np.uint8)
    board = np.repeat((30 + 210 * sq)[..., None], 3, -1).astype(np.uint8)
    board[:6, :, :] = 200; board[-6:, :, :] = 200; board[:, :6, :] = 200; board[:, -6:, :] = 200
    return area, board


def write_single_triangle_obj(path):
    """Config C1 geometry as an OBJ (the C1 fixture itself uses the 3-vertex ctor)."""
    with open(path, "w") as f:
        f.write("# single triangle\nv -1 0 -1\nv 1 0 -1\nv 0 0 1\nf 1 2 3\n")
    return 1


# Named workloads: geometry + camera(s) + material, as BASELINE.json configs[i].
C1 = dict(width=256, height=256, K=(120.0, 0, 128.0, 0, 120.0, 128.0, 0, 0, 1.0), D=D_REF,
          cam_pose=(0.0, -4.0, 0.0, 0.0, 0.0, 0.0), albedo=(0.1, 0.2, 0.9))
C2_CAMERAS = {"far": (0.0, -2.6, 0.2, 0.0, 0.0, 0.0),     # 24 % coverage
              "mid": (0.0, -1.6, 0.2, 0.0, 0.0, 0.0),     # 86 %
              "near": (0.0, -1.25, 0.2, 0.0, 0.0, 0.0)}   # 100 %
C2 = dict(width=1920, height=1080, D=D_REF, albedo=(0.9, 0.5, 0.2), n_tris=69936)
C4 = dict(width=3840, height=2160, D=D_REF, albedo=(0.8, 0.8, 0.7),
          cam_pose=(0.3, -8.5, 1.7, 0.15, 0.05, 0.0), spp=16, bounces=0, lighting=0)
# configs[2] / configs[4]: the C2 mesh with a half-mirror, slightly rough material, 64 samples per pixel, 8 specular
# bounces and the sun + shadow pass (the reference snapshot has none of these: semantics in DESIGN.md section 7)
C3 = dict(width=1920, height=1080, D=D_REF, albedo=(0.9, 0.5, 0.2), n_tris=69936, roughness=0.05, metallic=0.4,
          spp=64, bounces=8, lighting=1)
C5 = dict(C3, width=7680, height=4320)
# c6: the HBM regime.  The atrium generator at 16 x the triangle count of c4 (4 073 472 triangles: 395 MB of 64-B records, more
# than the 256 MiB Infinity Cache; most triangles are smaller than a pixel at 4K), same camera, 1 primary ray per pixel.
C6 = dict(C4, spp=1, atrium=dict(col_seg=192, col_rings=100, wall_div=384), n_tris=4073472)
# demo: the shape of the ONLY scene the reference itself renders (kernel.cu:155-240): two OBJ meshes, two textured materials, the
# second instance translated by (-0.6, 1.48, 0.73), camera at (-1, -4, 2) with K / D of kernel.cu:158-164, 1920x1080, 1 ray per pixel.
# The assets are not in the repository: write_demo_objs / demo_textures are deterministic stand-ins.  `baked` = the twin scene with the
# translation folded into the board's vertices and an identity instance (what the translated instance is compared with).
DEMO = dict(width=1920, height=1080, D=D_REF, albedo=(1.0, 1.0, 1.0), cam_pose=(-1.0, -4.0, 2.0, 0.0, 0.0, 0.0),
            board_pose=(-0.6, 1.48, 0.73, 0.0, 0.0, 0.0), n_tris=32768 + 3456)
WORKLOADS = {"c2": C2, "c3": C3, "c4": C4, "c5": C5, "c6": C6, "demo": DEMO}
