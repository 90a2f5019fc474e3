"""Importers: host mirrors of load_obj / load_gltf (src/hittable.rs:497-633).  The crates
behind them (tobj, gltf) are not in the reference tree, so their behaviour is restated (importers.py header);
these tests pin what the reference's own code does with their output."""
import base64
import io
import json
import struct

import numpy as np
import pytest

from rt_amd import importers
from rt_amd.api import ImageTexture, Lambertian, Metal, SceneDesc, SolidColor, Sphere, Triangle

OBJ = """# two objects, a quad, negative indices, v/vt/vn triples
o first
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vn 0 0 1
f 1/1/1 2/2/1 3/1/1 4/2/1
p 1
l 1 2
o second
v 0 0 1
v 2 0 1
v 0 2 1.5
f -3 -2 -1
f 1/1/1 5 6
"""


def write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_load_obj_models_triangulation_and_single_index(tmp_path):
    mat = Lambertian.new_rgb_solid(0.5, 0.5, 0.5)
    models = importers.load_obj(write(tmp_path, "m.obj", OBJ), mat)
    assert [len(m) for m in models] == [2, 2]
    q = models[0]
    # fan (0,1,2), (0,2,3)
    assert np.array_equal(q.a, [[0, 0, 0], [0, 0, 0]]) and np.array_equal(q.b, [[1, 0, 0], [1, 1, 0]])
    assert np.array_equal(q.c, [[1, 1, 0], [0, 1, 0]])
    assert np.array_equal(q.uv[0], [[0.0, 0.0], [1.0, 0.0], [0.5, 1.0]])       # Triangle::new's uvs, not the file's
    assert np.allclose(q.normal, [[0, 0, 1], [0, 0, 1]])
    s = models[1]
    assert np.array_equal(s.a[0], [0, 0, 1]) and np.array_equal(s.c[0], [0, 2, 1.5])
    assert np.array_equal(s.a[1], [0, 0, 0]) and np.array_equal(s.b[1], [0, 0, 1])
    t = s[1]
    assert isinstance(t, Triangle) and t.material is mat and t.c == (2.0, 0.0, 1.0)
    # positions are f32 in tobj: 1.1 is the f32 value, widened
    p = write(tmp_path, "f.obj", "v 1.1 0 0\nv 0 1.1 0\nv 0 0 1.1\nf 1 2 3\n")
    m = importers.load_obj(p, mat)[0]
    assert m.a[0, 0] == float(np.float32(1.1)) != 1.1


def test_load_obj_transform_drops_translation_and_centered_quirk(tmp_path):
    mat = Lambertian.new_rgb_solid(0.5, 0.5, 0.5)
    path = write(tmp_path, "m.obj", OBJ)
    M = np.array([[2.0, 0, 0, 10.0], [0, 3.0, 0, 20.0], [0, 0, 4.0, 30.0], [0, 0, 0, 1.0]])
    q = importers.load_obj(path, mat, transform=M)[0]
    assert np.array_equal(q.c[0], [2.0, 3.0, 0.0])                  # transform_vector: no translation
    # centered: sum over triangle corners / number of re-indexed vertices (4 distinct v/vt/vn triples)
    c = importers.load_obj(path, mat, centered=True)[0]
    plain = importers.load_obj(path, mat)[0]
    corner_sum = (plain.a + plain.b + plain.c).sum(0)
    assert np.allclose(plain.a - c.a, corner_sum / 4.0)
    with pytest.raises(FileNotFoundError):
        importers.load_obj(str(tmp_path / "missing.obj"), mat)


def test_usemtl_change_starts_a_new_model(tmp_path):
    text = "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nusemtl a\nf 1 2 3\nusemtl b\nf 2 4 3\nusemtl b\nf 1 2 4\n"
    models = importers.load_obj(write(tmp_path, "u.obj", text), Lambertian.new_rgb_solid(1, 1, 1))
    assert [len(m) for m in models] == [1, 2]


def make_gltf(tmp_path, with_texture, glb=False):
    from PIL import Image as PILImage
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.5]], dtype=np.float32)
    idx = np.array([0, 1, 2, 1, 3, 2], dtype=np.uint16)
    uv = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], dtype=np.float32)
    png = io.BytesIO()
    PILImage.fromarray(np.array([[[255, 0, 0, 255], [0, 128, 0, 255]], [[0, 0, 255, 255], [10, 20, 30, 0]]], dtype=np.uint8), "RGBA").save(png, "PNG")
    blob = pos.tobytes() + idx.tobytes() + uv.tobytes() + png.getvalue()
    o1, o2, o3 = len(pos.tobytes()), len(pos.tobytes()) + len(idx.tobytes()), len(pos.tobytes()) + len(idx.tobytes()) + len(uv.tobytes())
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": o1}, {"buffer": 0, "byteOffset": o1, "byteLength": o2 - o1},
                        {"buffer": 0, "byteOffset": o2, "byteLength": o3 - o2}, {"buffer": 0, "byteOffset": o3, "byteLength": len(blob) - o3}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"},
                      {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC2"}],
        "materials": [{"pbrMetallicRoughness": {"roughnessFactor": 0.25, "baseColorFactor": [0.2, 0.4, 0.6, 1.0]}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 2}, "indices": 1, "material": 0}]}],
        "nodes": [{"mesh": 0, "translation": [100, 0, 0]}],
    }
    if with_texture:
        doc["images"] = [{"bufferView": 3, "mimeType": "image/png"}]
        doc["textures"] = [{"source": 0}]
        doc["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    if glb:
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        bb = blob + b"\0" * (-len(blob) % 4)
        body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bb), 0x004E4942) + bb
        p = tmp_path / "m.glb"
        p.write_bytes(struct.pack("<4sII", b"glTF", 2, 12 + len(body)) + body)
        return str(p)
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    p = tmp_path / "m.gltf"
    p.write_text(json.dumps(doc))
    return str(p)


@pytest.mark.parametrize("glb", [False, True])
def test_load_gltf_geometry_uvs_and_material(tmp_path, glb):
    meshes = importers.load_gltf(make_gltf(tmp_path, with_texture=False, glb=glb))
    assert len(meshes) == 1 and len(meshes[0]) == 2
    m = meshes[0]
    assert np.array_equal(m.a, [[0, 0, 0], [1, 0, 0]]) and np.array_equal(m.b[1], [1, 1, 0.5])      # node translation ignored
    assert np.array_equal(m.uv[1], [[1, 0], [1, 1], [0, 1]])
    mat = m.material
    assert isinstance(mat, Metal) and mat.fuzz == 0.25 and isinstance(mat.texture, SolidColor)
    assert mat.texture.color == tuple(float(np.float32(x)) for x in (0.2, 0.4, 0.6))


def test_load_gltf_image_texture_becomes_metal_image(tmp_path):
    m = importers.load_gltf(make_gltf(tmp_path, with_texture=True))[0]
    assert isinstance(m.material.texture, ImageTexture)
    img = m.material.texture.image
    assert img.shape == (2, 2, 3) and img.dtype == np.uint8
    assert img[0, 1].tolist() == [0, 128, 0] and img[1, 1].tolist() == [10, 20, 30]       # alpha dropped
    bad = json.loads(open(make_gltf(tmp_path, with_texture=False)).read())
    del bad["meshes"][0]["primitives"][0]["attributes"]["TEXCOORD_0"]
    p = tmp_path / "bad.gltf"
    p.write_text(json.dumps(bad))
    with pytest.raises(ValueError, match="no tex coords"):
        importers.load_gltf(str(p))


def test_soups_flatten_like_triangle_lists(tmp_path):
    mat = Lambertian.new_rgb_solid(0.5, 0.5, 0.5)
    soups = importers.load_obj(write(tmp_path, "m.obj", OBJ), mat)
    extra = Triangle((5, 5, 5), (6, 5, 5), (5, 6, 5), mat)
    ball = Sphere((0, 0, 3), 1.0, Metal.new_solid((1, 1, 1), 0.1))
    fast = SceneDesc.from_shapes([extra, soups[0], ball, soups[1]])
    slow = SceneDesc.from_shapes([extra] + list(soups[0]) + [ball] + list(soups[1]))
    assert len(fast.triangles) == 5 and len(fast.spheres) == 1
    assert fast.triangles.tobytes() == slow.triangles.tobytes()
    assert fast.materials.tobytes() == slow.materials.tobytes()


def test_gltf_node_transforms_are_an_opt_in_extension(tmp_path):
    path = make_gltf(tmp_path, with_texture=False)
    doc = json.loads(open(path).read())
    doc["nodes"] = [{"children": [1], "translation": [100, 0, 0]}, {"mesh": 0, "scale": [2, 2, 2], "rotation": [0, 0, 0.7071067811865476, 0.7071067811865476]}]
    doc["scenes"] = [{"nodes": [0]}]
    p = tmp_path / "n.gltf"
    p.write_text(json.dumps(doc))
    plain = importers.load_gltf(str(p))[0]
    assert np.array_equal(plain.b[0], [1, 0, 0])
    moved = importers.load_gltf(str(p), apply_node_transforms=True)[0]
    assert np.allclose(moved.b[0], [100, 2, 0])            # (1,0,0): x2, rotated 90 degrees about z, then +100 in x
    assert np.allclose(moved.a[0], [100, 0, 0])


def test_write_png_round_trip(tmp_path):
    from PIL import Image as PILImage
    from rt_amd import write_png
    img = (np.arange(5 * 7 * 3) * 37 % 256).astype(np.uint8).reshape(5, 7, 3)
    write_png(str(tmp_path / "o.png"), img)
    assert np.array_equal(np.asarray(PILImage.open(tmp_path / "o.png").convert("RGB")), img)
