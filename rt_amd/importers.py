"""Mesh importers: the callers on the input side of the hot path.

Host-side mirrors of `load_obj` (src/hittable.rs:497-554, over the `tobj` crate with
`GPU_LOAD_OPTIONS`) and `load_gltf` (src/hittable.rs:556-633, over the `gltf` crate), returning one
triangle list per model / primitive like the reference's `Vec<Vec<Triangle>>`.  A list is a
`TriangleSoup`: the same triangles as arrays (a, b, c, uv, stored normal), because a million `Triangle`
objects are not a data format; it iterates as `Triangle` objects and `SceneDesc.from_shapes` flattens
it without a Python loop.

tobj 4 / gltf 1 are dependencies that are not vendored in /root/reference (Cargo.lock pins them); what is
restated here is their documented behaviour that the reference's call sites rely on:
  * tobj GPU_LOAD_OPTIONS = triangulate (fan: (0, i, i+1)), single_index (one index per distinct
    v/vt/vn triple, re-indexed per model in first-use order), ignore points and lines; positions are
    f32; a new model starts at every `o` / `g` and whenever `usemtl` changes after faces were read.
  * gltf: meshes in document order, every primitive read as TRIANGLES with indices, POSITION (f32) and
    TEXCOORD_0 (f32, or normalised u8/u16); node transforms are NOT applied (the reference iterates
    `gltf.meshes()`, not the scene graph).
Both are "parity unpinned" against the crates themselves (no cargo here); the tests pin the quirks of the
reference's own code: `transform_vector` drops the translation, `centered` divides the sum over triangle
corners by the number of re-indexed vertices, a glTF material becomes Metal(base colour or image,
fuzz = roughness)."""
import base64
import io
import json
import os
import struct

import numpy as np

from .api import ImageTexture, Metal, Triangle, triangle_normals


class TriangleSoup:
    """n triangles sharing one material: a, b, c (n, 3) f64; uv (n, 3, 2) f64; normal (n, 3) f64 as
    Triangle::new_with_uv stores it (src/hittable.rs:195-204)."""

    def __init__(self, a, b, c, material, uv=None):
        self.a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3)
        self.b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 3)
        self.c = np.ascontiguousarray(c, dtype=np.float64).reshape(-1, 3)
        n = len(self.a)
        if uv is None:                                    # Triangle::new's defaults (:175-177)
            uv = np.broadcast_to(np.array([[0.0, 0.0], [1.0, 0.0], [0.5, 1.0]]), (n, 3, 2))
        self.uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(n, 3, 2)
        self.material = material
        self.normal = triangle_normals(self.a, self.b, self.c) if n else np.zeros((0, 3))

    def __len__(self):
        return len(self.a)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __getitem__(self, i):
        return Triangle.new_with_uv(self.a[i], self.b[i], self.c[i], self.uv[i, 0], self.uv[i, 1], self.uv[i, 2], self.material)

    def transform(self, matrix):
        """Triangle::transform (:214-227): Matrix4::transform_vector = the linear 3x3 part only; the
        translation column is ignored (vectors, not points)."""
        m = np.asarray(matrix, dtype=np.float64)[:3, :3]

        def tv(p):   # nalgebra: column-major accumulation  x*col0 + y*col1 + z*col2
            return (p[:, 0:1] * m[:, 0] + p[:, 1:2] * m[:, 1]) + p[:, 2:3] * m[:, 2]

        return TriangleSoup(tv(self.a), tv(self.b), tv(self.c), self.material, self.uv)

    def shift(self, shift):
        """Triangle::shift (:229-239)."""
        s = np.asarray(shift, dtype=np.float64)
        return TriangleSoup(self.a + s, self.b + s, self.c + s, self.material, self.uv)


# ---------------------------------------------------------------------------------------------
# OBJ
# ---------------------------------------------------------------------------------------------
def _obj_models(path):
    """tobj::load_obj with GPU_LOAD_OPTIONS, geometry only: [(name, positions (k, 3) f32, indices (m, 3) u32)]."""
    pos, n_vt, n_vn = [], 0, 0
    models = []
    name, faces, cur_mtl = "unnamed_object", [], None

    def flush():
        nonlocal faces
        if not faces:
            return
        remap, out_pos, idx = {}, [], []
        for tri in faces:
            for key in tri:
                j = remap.get(key)
                if j is None:
                    j = remap[key] = len(out_pos)
                    out_pos.append(pos[key[0]])
                idx.append(j)
        models.append((name, np.array(out_pos, dtype=np.float32).reshape(-1, 3), np.array(idx, dtype=np.uint32).reshape(-1, 3)))
        faces = []

    def resolve(tok, n):
        i = int(tok)
        return i - 1 if i > 0 else n + i               # negative = relative to the vertices read so far

    with open(path, "r", errors="replace") as f:
        for line in f:
            parts = line.split("#", 1)[0].split()
            if not parts:
                continue
            k = parts[0]
            if k == "v":
                pos.append((np.float32(parts[1]), np.float32(parts[2]), np.float32(parts[3])))
            elif k == "vt":
                n_vt += 1
            elif k == "vn":
                n_vn += 1
            elif k == "f":
                corners = []
                for tok in parts[1:]:
                    fields = tok.split("/")
                    v = resolve(fields[0], len(pos))
                    vt = resolve(fields[1], n_vt) if len(fields) > 1 and fields[1] else -1
                    vn = resolve(fields[2], n_vn) if len(fields) > 2 and fields[2] else -1
                    if not 0 <= v < len(pos):
                        raise ValueError(f"{path}: face references vertex {fields[0]} of {len(pos)}")
                    corners.append((v, vt, vn))
                if len(corners) < 3:
                    continue                              # ignore_points / ignore_lines
                for i in range(1, len(corners) - 1):      # triangulate: fan
                    faces.append((corners[0], corners[i], corners[i + 1]))
            elif k in ("o", "g"):
                flush()
                name = " ".join(parts[1:]) if len(parts) > 1 else "unnamed_object"
            elif k == "usemtl":
                mtl = " ".join(parts[1:])
                if faces and mtl != cur_mtl:
                    flush()
                cur_mtl = mtl
    flush()
    return models


def load_obj(file_path, mesh_material, transform=None, centered=False):
    """load_obj (src/hittable.rs:497-554) -> [TriangleSoup per model]."""
    if not os.path.isfile(file_path):
        raise FileNotFoundError(f"Failed to OBJ load file: {file_path}")
    out = []
    for _name, positions, indices in _obj_models(file_path):
        p = positions.astype(np.float64)                  # Float::from(f32)
        a, b, c = p[indices[:, 0]], p[indices[:, 1]], p[indices[:, 2]]
        soup = TriangleSoup(a, b, c, mesh_material)
        if transform is not None:
            soup = soup.transform(transform)
        if centered:
            # :520-542: the sum runs over the (untransformed) corners of every triangle in order and is
            # divided by the number of re-indexed vertices — not a centroid (the reference's own TODO)
            sum_pos = np.zeros(3)
            for tri_sum in (a + b) + c:
                sum_pos = sum_pos + tri_sum
            mean_pos = sum_pos / float(len(positions))
            soup = soup.shift(-mean_pos)
        out.append(soup)
    return out


# ---------------------------------------------------------------------------------------------
# glTF 2.0 (.gltf + external/embedded buffers, .glb)
# ---------------------------------------------------------------------------------------------
_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _read_uri(uri, base_dir):
    if uri.startswith("data:"):
        return base64.b64decode(uri.split(",", 1)[1])
    with open(os.path.join(base_dir, uri), "rb") as f:
        return f.read()


def _load_document(path):
    with open(path, "rb") as f:
        raw = f.read()
    glb_bin = None
    if raw[:4] == b"glTF":
        _magic, _ver, _length = struct.unpack_from("<4sII", raw, 0)
        off, doc = 12, None
        while off < len(raw):
            clen, ctype = struct.unpack_from("<II", raw, off)
            chunk = raw[off + 8: off + 8 + clen]
            if ctype == 0x4E4F534A:
                doc = json.loads(chunk.decode("utf-8"))
            elif ctype == 0x004E4942 and glb_bin is None:
                glb_bin = chunk
            off += 8 + clen
        if doc is None:
            raise ValueError(f"gltf loader failed to read {path}")
    else:
        doc = json.loads(raw.decode("utf-8"))
    base = os.path.dirname(os.path.abspath(path))
    buffers = []
    for b in doc.get("buffers", []):
        buffers.append(_read_uri(b["uri"], base) if "uri" in b else glb_bin)
    return doc, buffers, base


def _accessor(doc, buffers, index):
    acc = doc["accessors"][index]
    dt, nc = np.dtype(_COMPONENT[acc["componentType"]]), _NCOMP[acc["type"]]
    count = acc["count"]
    if "bufferView" not in acc:
        return np.zeros((count, nc), dtype=dt), acc
    bv = doc["bufferViews"][acc["bufferView"]]
    start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * nc
    buf = buffers[bv["buffer"]]
    arr = np.ndarray((count, nc), dtype=dt, buffer=buf, offset=start, strides=(stride, dt.itemsize))
    return np.array(arr), acc


def _gltf_image(doc, buffers, base, index):
    """gltf::import's decoded image -> (H, W, 3) uint8 as `impl From<&gltf::image::Data> for Image`
    reads it (src/camera.rs:83-125): channel/255, missing channels 0, alpha dropped."""
    from PIL import Image as PILImage
    im = doc["images"][index]
    if "uri" in im:
        data = _read_uri(im["uri"], base)
    else:
        bv = doc["bufferViews"][im["bufferView"]]
        o = bv.get("byteOffset", 0)
        data = buffers[bv["buffer"]][o: o + bv["byteLength"]]
    pil = PILImage.open(io.BytesIO(data))
    if pil.mode in ("I;16", "I;16B", "I;16L", "I", "F"):
        raise NotImplementedError("16-bit / float glTF images: the reference's conversion walks bytes, not texels "
                                  "(src/camera.rs:91-95, todo!() or wrong); refused here")
    if pil.mode == "P":
        pil = pil.convert("RGBA" if "transparency" in pil.info else "RGB")
    elif pil.mode == "1":
        pil = pil.convert("L")
    arr = np.asarray(pil)                                 # L -> R8, LA -> R8G8, RGB -> R8G8B8, RGBA -> R8G8B8A8
    if arr.ndim == 2:
        arr = arr[..., None]
    rgb = np.zeros(arr.shape[:2] + (3,), dtype=np.uint8)
    k = min(3, arr.shape[2])
    rgb[..., :k] = arr[..., :k]
    return rgb


def material_from_gltf(doc, buffers, base, material_index):
    """Material::from_gltf (src/material.rs:20-33): Metal with fuzz = roughnessFactor over the base colour
    texture if there is one, else over the solid baseColorFactor rgb."""
    mat = doc["materials"][material_index] if material_index is not None else {}
    pbr = mat.get("pbrMetallicRoughness", {})
    fuzz = float(np.float32(pbr.get("roughnessFactor", 1.0)))
    color = [float(np.float32(x)) for x in pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])]
    tex = pbr.get("baseColorTexture")
    if tex is not None:
        source = doc["textures"][tex["index"]]["source"]
        return Metal(ImageTexture(_gltf_image(doc, buffers, base, source)), fuzz)
    return Metal.new_solid((color[0], color[1], color[2]), fuzz)


def _node_world_matrices(doc):
    """mesh index -> [4x4 world matrix of every node of the default scene that instantiates it] (column-vector
    convention, parent * local; local = matrix or T * R * S)."""
    nodes = doc.get("nodes", [])

    def local(n):
        if "matrix" in n:
            return np.array(n["matrix"], dtype=np.float64).reshape(4, 4).T        # glTF stores column-major
        t = np.array(n.get("translation", [0, 0, 0]), dtype=np.float64)
        x, y, z, w = np.array(n.get("rotation", [0, 0, 0, 1]), dtype=np.float64)
        sc = np.array(n.get("scale", [1, 1, 1]), dtype=np.float64)
        r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        m = np.eye(4)
        m[:3, :3] = r * sc
        m[:3, 3] = t
        return m

    scenes_ = doc.get("scenes", [])
    roots = scenes_[doc.get("scene", 0)]["nodes"] if scenes_ else list(range(len(nodes)))
    out = {}
    stack = [(i, np.eye(4)) for i in roots]
    seen = 0
    while stack:
        i, parent = stack.pop()
        seen += 1
        if seen > 4 * len(nodes) + 4:
            raise ValueError("glTF node graph is not a tree")
        m = parent @ local(nodes[i])
        if "mesh" in nodes[i]:
            out.setdefault(nodes[i]["mesh"], []).append(m)
        stack.extend((c, m) for c in nodes[i].get("children", []))
    return out


def load_gltf(file_path, _mesh_material=None, apply_node_transforms=False):
    """load_gltf (src/hittable.rs:556-633) -> [TriangleSoup per primitive that has indices and positions].
    apply_node_transforms=True is an extension (SURVEY §8 f3 "honour node transforms"): every instance of a mesh in
    the default scene is emitted with its node's world matrix applied to the positions (as points); the default
    mirrors the reference, which reads `gltf.meshes()` and ignores the scene graph."""
    try:
        doc, buffers, base = _load_document(file_path)
    except (OSError, ValueError, KeyError) as e:
        raise ValueError(f"gltf loader failed to read {file_path}") from e
    meshes = []
    world = _node_world_matrices(doc) if apply_node_transforms else {}
    for mesh_index, mesh in enumerate(doc.get("meshes", [])):
        for prim in mesh.get("primitives", []):
            material = material_from_gltf(doc, buffers, base, prim.get("material"))
            attrs = prim.get("attributes", {})
            if "indices" not in prim or "POSITION" not in attrs:
                continue
            indices = _accessor(doc, buffers, prim["indices"])[0].reshape(-1).astype(np.uint32)
            positions = _accessor(doc, buffers, attrs["POSITION"])[0].astype(np.float32)
            if "TEXCOORD_0" not in attrs:
                raise ValueError("no tex coords")        # .expect("no tex coords") (:590)
            tc, acc = _accessor(doc, buffers, attrs["TEXCOORD_0"])
            if tc.dtype == np.uint8:
                tc = tc.astype(np.float32) / np.float32(255.0)          # into_f32 of normalised integers
            elif tc.dtype == np.uint16:
                tc = tc.astype(np.float32) / np.float32(65535.0)
            tc = tc.astype(np.float32)
            tri = indices[: len(indices) // 3 * 3].reshape(-1, 3)       # chunks_exact(3)
            p = positions.astype(np.float64)
            uv = tc.astype(np.float64)[tri]                             # (n, 3, 2)
            if not apply_node_transforms:
                meshes.append(TriangleSoup(p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]], material, uv))
                continue
            for m in world.get(mesh_index, []):
                pw = p @ m[:3, :3].T + m[:3, 3]
                meshes.append(TriangleSoup(pw[tri[:, 0]], pw[tri[:, 1]], pw[tri[:, 2]], material, uv))
    return meshes
