"""Scene definitions shared by the parity tests: each builds the SAME scene on the oracle
(tests/orc.py) and on the product (cuda-raytracing_amd) from one description."""
import numpy as np


def checker_texture(w=64, h=48, seed=7):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    t[::8, :, :] = 255
    return t


def random_triangles(n, seed, spread=1.0, size=0.25, uv=True):
    """n random triangles as [n,18] (normal via the 3-vertex constructor of the oracle)."""
    import orc
    o = orc.oracle()
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 1, 3)).astype(np.float32)
    v = (c + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32)
    out = np.zeros((n, 18), np.float32)
    for i in range(n):
        out[i] = o.tri_from_vertices(v[i].ravel())
    if uv:
        out[:, 12:18] = rng.uniform(0, 1, (n, 6)).astype(np.float32)
    return out


class SceneDesc:
    """materials: list of (albedo, texture or None[, dict(roughness=, metallic=, illumination=)]); meshes: list of ('obj', path) | ('tris', array[n,18]) |
    ('tri3', abc9); instances: list of (mesh, material, pose6, scale3)."""

    def __init__(self, materials, meshes, instances):
        self.materials, self.meshes, self.instances = materials, meshes, instances

    def build_oracle(self, orc):
        o = orc.oracle()
        s = orc.OracleScene(o)
        for mat in self.materials:
            albedo, tex = mat[0], mat[1]
            extra = mat[2] if len(mat) > 2 else {}
            s.add_material(albedo, tex, **extra)
        self.oracle_meshes = []
        for kind, arg in self.meshes:
            if kind == "obj":
                m = o.obj_load(arg)
            elif kind == "tris":
                m = o.mesh_from_triangles(arg)
            else:
                m = o.mesh_single_triangle(arg)
            assert m, "oracle mesh build failed"
            self.oracle_meshes.append(m)
            s.add_mesh(m)
        for mesh, mat, pose, scale in self.instances:
            s.add_instance(mesh, mat, pose, scale)
        return s

    def build_product(self, rt, gpu_build=False):
        """gpu_build: the meshes' trees come from rt_bvh_build instead of the host builder (same tree, so same planes)"""
        s = rt.Scene()
        for mat in self.materials:
            albedo, tex = mat[0], mat[1]
            extra = mat[2] if len(mat) > 2 else {}
            s.add_material(albedo, texture_bgr=tex, **extra)
        self.product_meshes = []
        for kind, arg in self.meshes:
            if kind == "obj":
                m = rt.Mesh.load_obj(arg, gpu_build=gpu_build)
            elif kind == "tris":
                m = rt.Mesh.from_triangles(arg, gpu_build=gpu_build)
            else:
                m = rt.Mesh.single_triangle(arg)
            self.product_meshes.append(m)
            s.add_mesh(m)
        for mesh, mat, pose, scale in self.instances:
            s.add_mesh_instance(mesh, mat, pose, scale)
        return s


def c1_scene(scenes):
    return SceneDesc([(scenes.C1["albedo"], None)], [("tri3", [-1, 0, -1, 1, 0, -1, 0, 0, 1])], [(0, 0, (0,) * 6, (1, 1, 1))])


def blob_scene(scenes, path, albedo=None):
    return SceneDesc([(albedo or scenes.C2["albedo"], None)], [("obj", path)], [(0, 0, (0,) * 6, (1, 1, 1))])


def multi_instance_scene(scenes, blob_path):
    """Two meshes, three instances with rotation + non-uniform scale, a textured material and an
    albedo material (exercises raycast.cu:33-51, :98-122, :224-245)."""
    tex = checker_texture()
    tris = random_triangles(300, seed=11, spread=0.8, size=0.3)
    return SceneDesc(
        [((0.9, 0.5, 0.2), None), ((1.0, 1.0, 1.0), tex), ((0.2, 0.7, 0.4), None)],
        [("obj", blob_path), ("tris", tris)],
        [(0, 1, (0.2, 0.5, -0.1, 0.7, 0.3, -0.4), (0.6, 1.1, 0.8)),
         (1, 0, (-1.5, 1.0, 0.3, -0.2, 0.1, 0.9), (1.3, 0.7, 1.0)),
         (0, 2, (1.6, 1.5, 0.4, 0.0, 0.0, 0.0), (0.5, 0.5, 0.5))])


MULTI_CAMERA = dict(width=320, height=200, pose=(0.1, -3.0, 0.4, 0.15, -0.05, 0.1))


def atrium_scene(scenes, path):
    """BASELINE.json configs[3] geometry ("Sponza-class", 260 352 triangles, camera inside) with a textured material so
    that RGB varies across the frame."""
    return SceneDesc([(scenes.C4["albedo"], checker_texture(128, 96, seed=21))], [("obj", path)], [(0, 0, (0,) * 6, (1, 1, 1))])


def deep_stack_scene(n=28):
    """Triangles at exponentially growing distance along the view axis (+y), each facing the camera and subtending the
    same angle: the BVH degenerates into a chain (about n levels, capped at 32 by the builder) and central rays keep
    one postponed far child per level on the traversal stack -- far more than the 16 entries the kernel keeps in LDS."""
    import orc
    o = orc.oracle()
    tris = np.zeros((n, 18), np.float32)
    for i in range(n):
        y = 1e-4 * (6.1 ** i)                 # each triangle 6.1x farther: every candidate plane splits off only the last one
                                              # (y^2 must stay finite in fp32, which caps the chain near 28 levels)
        h = 0.6 * (y + 1.0)
        # winding chosen so the normal points to -y (towards a camera at y = -1)
        tris[i] = o.tri_from_vertices(np.array([-h, y, -h, h, y, -h, 0.0, y, h], np.float32))
        tris[i, 12:18] = [0, 0, 0.5, 1, 1, 0]
    return SceneDesc([((0.2, 0.9, 0.4), None)], [("tris", tris)], [(0, 0, (0,) * 6, (1, 1, 1))])


def shiny_scene(scenes, blob_path):
    """Extension test scene: a half-mirror rough blob over a mirror-ish textured floor and a matte wall."""
    import orc
    o = orc.oracle()
    floor = np.stack([o.tri_from_vertices(np.array(v, np.float32)) for v in
                      ([-6, -6, -1.15, 6, -6, -1.15, 6, 8, -1.15], [-6, -6, -1.15, 6, 8, -1.15, -6, 8, -1.15])])
    floor[:, 12:18] = [[0, 0, 1, 0, 1, 1], [0, 0, 1, 1, 0, 1]]
    wall = np.stack([o.tri_from_vertices(np.array([-6, 3.5, -1.15, 6, 3.5, -1.15, 0, 3.5, 6], np.float32))])
    return SceneDesc(
        [((0.9, 0.5, 0.2), None, dict(roughness=0.08, metallic=0.5)),
         ((1.0, 1.0, 1.0), checker_texture(32, 32, seed=3), dict(roughness=0.0, metallic=0.35)),
         ((0.3, 0.8, 0.4), None, dict(roughness=0.4, metallic=0.0))],
        [("obj", blob_path), ("tris", floor), ("tris", wall)],
        [(0, 0, (0, 0, 0, 0.3, 0.0, 0.1), (0.9, 1.0, 1.1)), (1, 1, (0,) * 6, (1, 1, 1)), (2, 2, (0,) * 6, (1, 1, 1))])


SHINY_CAMERA = dict(width=240, height=136, pose=(0.3, -3.2, 0.9, 0.05, -0.2, 0.0))


def demo_scene(scenes, objs, baked=False):
    """The reference demo's scene shape (kernel.cu:166-240; bench.py --workload demo): two OBJ meshes, two textured materials, the
    second instance translated by (-0.6, 1.48, 0.73) -- or, baked, the same board as an identity instance."""
    area_tex, board_tex = scenes.demo_textures()
    w = scenes.DEMO
    return SceneDesc([(w["albedo"], area_tex), (w["albedo"], board_tex)], [("obj", objs[0]), ("obj", objs[2] if baked else objs[1])],
                     [(0, 0, (0,) * 6, (1, 1, 1)), (1, 1, (0,) * 6 if baked else w["board_pose"], (1, 1, 1))])
