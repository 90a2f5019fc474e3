//! `impl From<&[Shape]> for FlatScene` — the flattening `World::build` does before `rt_scene_create`
//! (the Python mirror is `SceneDesc.from_shapes` in rt_amd/api.py, the C++ one `World::build` in
//! rt_amd/host/rt.hpp).  SOURCE ONLY, never compiled here.  It needs three getters the reference does not have yet
//! (its fields are private): `Triangle::normal()`, `Triangle::uvs() -> [Vec2; 3]`, `Sphere::front_direction()`.
use std::collections::HashMap;
use std::sync::Arc;

use rt::hittable::{Shape, World};
use rt::material::Material;
use rt::texture::TextureEnum;

use crate::*;

fn v3(v: &rt::vec3::Vec3) -> [f64; 3] { [v.x, v.y, v.z] }

struct Tables { materials: Vec<RtMaterial>, textures: Vec<RtTexture>, pool: Vec<u8>, mat_ids: HashMap<*const Material, u32> }

impl Tables {
    /// Textures become a node table with children before parents (`CheckerTexture.even/odd` are boxed `TextureEnum`s,
    /// src/texture.rs:47-48); images are repacked as RGBA8 (`Image.pixels[i].2` is channel / 255, src/camera.rs:68-70).
    fn texture(&mut self, t: &TextureEnum) -> u32 {
        let mut rec = RtTexture::default();
        match t {
            TextureEnum::SolidColor(s) => { rec.kind = RT_TEX_SOLID; rec.rgb = v3(&s.color); }
            TextureEnum::CheckerTexture(c) => {
                rec.even = self.texture(&c.even_texture);
                rec.odd = self.texture(&c.odd_texture);
                rec.kind = RT_TEX_CHECKER;
                rec.inv_scale = c.scale_inverted;
            }
            TextureEnum::ImageTexture(i) => {
                while self.pool.len() % 8 != 0 { self.pool.push(0); }
                rec.kind = RT_TEX_IMAGE;
                rec.width = i.image.width as u32;
                rec.height = i.image.height as u32;
                rec.texel_format = RT_TEXEL_RGBA8;
                rec.texel_offset = self.pool.len() as u64;
                for (_, _, c) in i.image.pixels.iter() {                 // row-major, no v-flip (src/texture.rs:108-117)
                    for ch in [c.x, c.y, c.z] { self.pool.push((ch * 255.0).round() as u8); }
                    self.pool.push(255);
                }
            }
        }
        self.textures.push(rec);
        (self.textures.len() - 1) as u32
    }
    /// Materials are shared `Arc<Material>`s (src/hittable.rs:161, src/main.rs:34-37): one record per allocation.
    fn material(&mut self, m: &Arc<Material>) -> u32 {
        let key = Arc::as_ptr(m);
        if let Some(&id) = self.mat_ids.get(&key) { return id; }
        let mut rec = RtMaterial::default();
        match m.as_ref() {
            Material::Lambertian(l) => { rec.kind = RT_MAT_LAMBERTIAN; rec.texture = self.texture(&l.texture); }
            Material::Metal(mt) => {
                rec.kind = RT_MAT_METAL;
                rec.texture = self.texture(&mt.texture);
                rec.has_fuzz = mt.fuzz.is_some() as u32;                 // Some(0.0) still draws random_unit
                rec.fuzz = mt.fuzz.unwrap_or(0.0);
            }
            Material::Dielectric(d) => {
                rec.kind = RT_MAT_DIELECTRIC;
                rec.ior = d.refractive_index;
                rec.has_fuzz = d.fuzz.is_some() as u32;
                rec.fuzz = d.fuzz.unwrap_or(0.0);
            }
        }
        self.materials.push(rec);
        let id = (self.materials.len() - 1) as u32;
        self.mat_ids.insert(key, id);
        id
    }
}

/// Flatten `World::shapes` (spheres first, then triangles, each kind in `Vec<Shape>` order) and the sky state.
pub fn flatten_world(world: &World) -> FlatScene {
    let mut t = Tables { materials: vec![], textures: vec![], pool: vec![], mat_ids: HashMap::new() };
    let mut flat = FlatScene::default();
    let (mut sph_pos, mut tri_pos) = (vec![], vec![]);
    for (i, s) in world.shapes.iter().enumerate() {
        match s {
            Shape::Sphere(sp) => {
                let material = t.material(&sp.material);
                flat.spheres.push(RtSphere { center: v3(&sp.center), radius: sp.radius, face: v3(&sp.front_direction()), material, _pad: 0 });
                sph_pos.push(i as u32);
            }
            Shape::Triangle(tr) => {
                let material = t.material(&tr.material);
                let uv = tr.uvs();
                flat.triangles.push(RtTriangle { a: v3(&tr.a), b: v3(&tr.b), c: v3(&tr.c),
                                                 uv: [[uv[0].x, uv[0].y], [uv[1].x, uv[1].y], [uv[2].x, uv[2].y]],
                                                 n: v3(&tr.normal()), material, _pad: 0 });
                tri_pos.push(i as u32);
            }
        }
    }
    let (params, radiances) = world.sky.raw();                           // hw_skymodel::rgb::SkyState::raw
    flat.sky = Some(RtSky { params, radiances, sun: v3(&world.sun_direction) });
    flat.materials = t.materials;
    flat.textures = t.textures;
    flat.texel_pool = t.pool;
    flat.prim_to_shape = sph_pos.into_iter().chain(tri_pos).collect();
    flat
}
